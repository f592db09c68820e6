"""GPU: the LiTS batch-assembly kernel (`unetk_lits_batch`) against the numpy restatement of the
reference's per-sample pre-processing (oracle/lits_ops.py <- DataLoader/Liver/input_pipeline.py:243-284), and the whole
device-resident pipeline feeding a training run from a synthetic dataset in the reference's on-disk format."""
import argparse
import json

import numpy as np
import pytest
import torch

from oracle import lits_ops

pytestmark = pytest.mark.gpu


def _store(rng, n_slices, h, w):
    im = rng.integers(0, 65535, size=(n_slices, h, w)).astype(np.uint16)
    # smooth-ish structure so that interpolation matters
    im = (im // 8 + (np.arange(h)[None, :, None] * 97 + np.arange(w)[None, None, :] * 53)).astype(np.uint16)
    lb = (rng.integers(0, 3, size=(n_slices, h, w)) * 64).astype(np.uint8)
    return im, lb


@pytest.mark.parametrize("out_hw,channels", [((32, 32), 3), ((24, 40), 1), ((17, 9), 3)])
def test_lits_batch_matches_numpy_restatement(out_hw, channels):
    from boxsegliver_amd import ops
    rng = np.random.default_rng(5)
    im, lb = _store(rng, 10, 48, 56)
    c = channels
    rows, clips, expect = [], [], []
    boxes = [(0, 0, 48, 56), (3, 5, 32, 32), (10, 2, 17, 50), (7, 10, out_hw[0], out_hw[1]), (1, 1, 40, 9)]   # inside 48x56
    for j, (oy, ox, ch, cw) in enumerate(boxes):
        center = 2 + j
        chans = [center] if c == 1 else [center - 1 if j != 2 else -1, center, center + 1 if j != 3 else -1]
        flip_lr, flip_ud = j % 2, (j // 2) % 2
        seg = center if j != 4 else -1
        rows.append(chans + [seg, oy, ox, ch, cw, flip_lr, flip_ud])
        clip = (10 * 64.0 + 100 * j, 500 * 64.0 + 50 * j)
        clips.append(clip)
        expect.append(lits_ops.process_sample([im[s] if s >= 0 else None for s in chans], lb[seg] if seg >= 0 else None,
                                              [oy, ox, ch, cw], clip, out_hw, 64, bool(flip_lr), bool(flip_ud)))
    tab = torch.tensor(rows, dtype=torch.int32).cuda()
    clip_t = torch.tensor(clips, dtype=torch.float32).cuda()
    im_t = torch.from_numpy(im.view(np.int16)).cuda()
    lb_t = torch.from_numpy(lb).cuda()
    images, labels = ops.lits_batch(im_t, lb_t, tab, clip_t, out_hw, c, 64, 0.0, 0)
    for j, (img, lab) in enumerate(expect):
        np.testing.assert_array_equal(labels[j].cpu().numpy(), lab)
        np.testing.assert_allclose(images[j].cpu().numpy(), img, atol=2e-6)
    # noise: bounded, zero-mean, exactly absent on zero-padding slices, reproducible per seed, different across seeds
    noisy, _ = ops.lits_batch(im_t, lb_t, tab, clip_t, out_hw, c, 64, 0.05, 123)
    d = (noisy - images).cpu().numpy()
    assert np.abs(d).max() <= 0.05 + 1e-6 and abs(d.mean()) < 2e-3 and d.std() > 0.02
    if c == 3:
        assert np.all(d[2, ..., 0] == 0) and np.all(d[3, ..., 2] == 0)
    again, _ = ops.lits_batch(im_t, lb_t, tab, clip_t, out_hw, c, 64, 0.05, 123)
    other, _ = ops.lits_batch(im_t, lb_t, tab, clip_t, out_hw, c, 64, 0.05, 124)
    assert torch.equal(noisy, again) and not torch.equal(noisy, other)


def _write_dataset(root, n_cases=3, depth=6, size=64):
    from boxsegliver_amd.data import lits
    rng = np.random.default_rng(9)
    meta = []
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    for pid in range(n_cases):
        d = root / "png" / "volume-{:d}".format(pid)
        d.mkdir(parents=True)
        liver = ((yy - 30) / 18.0) ** 2 + ((xx - 28) / 15.0) ** 2 <= 1
        tumor = (yy - 33) ** 2 + (xx - 26) ** 2 <= 16
        for z in range(depth):
            lab = np.zeros((size, size), np.uint8)
            if 1 <= z <= depth - 2:
                lab[liver] = 1
                if z in (2, 3):
                    lab[tumor] = 2
            hu = rng.normal(60, 30, size=(size, size)) + 80 * (lab > 0) - 40 * (lab == 2)
            im = ((np.clip(hu, -200, 250) + 200) * 64).astype(np.uint16)
            (d / "{:03d}_im.png".format(z)).write_bytes(lits.png_encode(im))
            (d / "{:03d}_lb.png".format(z)).write_bytes(lits.png_encode((lab * 64).astype(np.uint8)))
        meta.append({"PID": pid, "size": [depth, size, size], "spacing": [2.5, 0.8, 0.8], "bbox": [1, 12, 13, depth - 1, 49, 44],
                     "tumors": "[]", "tumor_areas": [], "tumor_centers": "[]", "tumor_stddevs": "[]",
                     "tumor_slices_from_to": [0, 1, 2], "tumor_slices": "[[29, 22, 38, 31], [29, 22, 38, 31]]",
                     "tumor_slices_index": [2, 3], "tumor_slices_centers": "[[33.0, 26.0], [33.0, 26.0]]",
                     "tumor_slices_stddevs": "[[2.0, 2.0], [2.0, 2.0]]", "tumor_slices_areas": [49, 49],
                     "tumor_slices_tid": [0, 0]})
    (root / "meta.json").write_text(json.dumps(meta))
    (root / "k_folds.txt").write_text("Fold 0:0\nFold 1:1\nFold 2:2\n")


def test_device_resident_pipeline_feeds_training(tmp_path):
    """PNG dataset on disk -> SliceStore in HBM -> sampler + gather kernel -> CustomEstimator.train with online eval."""
    import test_gpu_unet as t
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data import lits
    _write_dataset(tmp_path)
    args = t.make_args(batch_size=4, im_height=32, im_width=32, im_channel=3, test_fold=2, filter_size=0, noise_scale=0.05,
                       zoom_scale=(1.0, 1.4), random_flip=3, liver_percent=0.66, tumor_percent=0.5, eval_per_epoch=True,
                       eval_num_batches_per_epoch=2, model="UNet", log_step=1)
    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "solver": Solver(args),
              "solver_kwargs": {}, "lits_root": str(tmp_path)}
    batch = next(lits.input_fn("train", params))
    feats, labels = batch
    assert feats["images"].shape == (4, 32, 32, 3) and feats["images"].is_cuda and labels.dtype == torch.int32
    assert -0.05 - 1e-6 <= feats["images"].min().item() and feats["images"].max().item() <= 1.05 + 1e-6
    assert set(np.unique(labels.cpu().numpy())) <= {0, 1, 2} and int((labels[:2] == 2).sum()) > 0    # forced tumor share
    store = params[("lits_store", True)][0]
    assert store.im.shape == (2 * 6, 64, 64) and store.im.is_cuda                                      # folds 0 and 1 resident
    e = est.CustomEstimator(models.model_fn, str(tmp_path / "run"), est.RunConfig(model_dir=str(tmp_path / "run"),
                                                                                 save_checkpoints_steps=0), params)
    e.train(lits.input_fn, steps=3)
    evals = list(e.evaluate_online(None, ["Liver/Dice"], yield_single_examples=False))
    assert len(evals) == 2 and 0.0 <= float(evals[0]["Liver/Dice"]) <= 1.0


@pytest.mark.parametrize("use_global_dice", [False, True])
def test_online_3d_evaluation_from_the_training_session(tmp_path, use_global_dice):
    """--eval_3d (evaluators/evaluator_liver.py:171-282): every validation case served once as slice batches over its
    liver z range (the last batch padded), predictions stacked per case and scored in 3-D -- against the same loop written
    out here (own forward passes, numpy metrics)."""
    import test_gpu_unet as t
    from boxsegliver_amd import loss_metrics
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data import lits
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    _write_dataset(tmp_path, n_cases=4, depth=7)
    (tmp_path / "k_folds.txt").write_text("Fold 0:0 1\nFold 1:2 3\n")
    args = t.make_args(batch_size=4, im_height=32, im_width=32, im_channel=3, test_fold=1, filter_size=0, noise_scale=0.05,
                       zoom_scale=(1.0, 1.2), random_flip=3, liver_percent=0.66, tumor_percent=0.5, eval_per_epoch=True,
                       eval_num_batches_per_epoch=2, model="UNet", log_step=1, eval_3d=True, use_global_dice=use_global_dice,
                       metrics_eval=["Dice", "VOE"])
    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "solver": Solver(args),
              "solver_kwargs": {}, "lits_root": str(tmp_path)}
    e = est.CustomEstimator(models.model_fn, str(tmp_path / "run"), est.RunConfig(model_dir=str(tmp_path / "run"),
                                                                                 save_checkpoints_steps=0), params)
    e.train(lits.input_fn, steps=2)
    evaluator = ev.get_evaluator("Volume", estimator=e, model_dir=str(tmp_path / "run"), params=params)
    results = evaluator.run_with_session(None)
    # the batches: 2 validation cases, liver z range [1, 6) = 5 slices -> 2 batches of 4 each, 3 padding slices at the end
    store, cases = params[("lits_store", False)]
    batches = list(lits.batches_eval_3d(store, cases, args))
    assert len(batches) == 4 and [int(f["names"][0]) for f, _ in batches] == [2, 2, 3, 3]
    assert int(batches[1][1][1:].abs().sum()) == 0 and float(batches[1][0]["images"][1:].abs().sum()) == 0   # padding: zeros
    # the same evaluation written out
    model = params["model_instances"][0]
    per_case, tp_fp_fn = [], np.zeros((2, 3))
    for k in range(2):
        preds, labs = [], []
        for feats, labels in batches[2 * k:2 * k + 2]:
            model({"images": feats["images"], "labels": labels}, "eval", **t.YML)
            preds.append(torch.stack([model.predictions[c + "Pred"][..., 0] for c in ("Liver", "Tumor")]).cpu().numpy())
            labs.append(labels.cpu().numpy())
        pred, lab = np.concatenate(preds, axis=1)[:, :5], np.concatenate(labs, axis=0)[:5]
        row = {}
        for i, cls in enumerate(("Liver", "Tumor")):
            ref = (lab == i + 1)
            for met, v in loss_metrics.metric_3d(pred[i], ref, required=["Dice", "VOE"]).items():
                row["{}/{}".format(cls, met)] = v
            tp_fp_fn[i] += [np.sum((pred[i] > 0) & ref), np.sum((pred[i] > 0) & ~ref), np.sum(~(pred[i] > 0) & ref)]
        per_case.append(row)
    if use_global_dice:
        want = {cls + "/Dice": 2 * tp_fp_fn[i, 0] / max(2 * tp_fp_fn[i, 0] + tp_fp_fn[i, 1] + tp_fp_fn[i, 2], 1)
                for i, cls in enumerate(("Liver", "Tumor"))}
    else:
        want = {k: float(np.mean([r[k] for r in per_case])) for k in per_case[0]}
    assert set(results) == set(want)
    for k in want:
        assert results[k] == pytest.approx(want[k], abs=1e-12), k


def test_entry_main_trains_from_the_command_line(tmp_path):
    """`python -m boxsegliver_amd.entry.main liver --mode train ...` on a synthetic dataset in the reference's on-disk format:
    plateau policy + online evaluation + best checkpoint, the hook set entry/main.py:163-186 installs; and the GUNet entry
    (main_g, `nf_inter` on synthetic NF-shaped tensors) with EvaluatorHookV2."""
    import json
    import os
    from boxsegliver_amd.entry import main as entry
    from boxsegliver_amd.entry import main_g
    _write_dataset(tmp_path)
    run = tmp_path / "run"
    argv = ("liver --mode train --tag cli --model UNet --classes Liver Tumor --test_fold 2 --im_height 32 --im_width 32 "
            "--im_channel 3 --noise_scale 0.05 --random_flip 3 --eval_num_batches_per_epoch 2 --num_of_steps 5 "
            "--primary_metric Tumor/Dice --secondary_metric Liver/Dice --loss_weight_type numerical --loss_numeric_w 0.2 0.4 4.4 "
            "--batches_per_epoch 2 --batch_size 4 --weight_decay_rate 0.000001 --learning_policy plateau --learning_rate 0.001 "
            "--lr_end 0 --lr_decay_rate 0.2 --eval_per_epoch --evaluator Volume --save_best --log_step 1").split()
    argv += ["--lits_root", str(tmp_path), "--model_dir", str(run)]
    assert entry.main(argv) == 0
    status = json.load(open(str(run / "checkpoint")))
    assert status["global_step"] == 5 and os.path.exists(str(run / status["model_checkpoint_path"]))
    best = json.load(open(str(run / "best_result")))
    assert set(best) == {"Liver/Dice", "Tumor/Dice"} and os.path.exists(str(run / "checkpoint_best"))
    assert os.path.exists(str(run / "lr_schedule")) and os.path.exists(str(run / "logs" / "train_cli"))
    # resuming: max_steps already reached -> nothing to do; a larger budget continues from step 5
    argv2 = [a for a in argv]
    i = argv2.index("--num_of_steps")
    argv2[i:i + 2] = ["--num_of_total_steps", "7"]
    assert entry.main(argv2) == 0
    assert json.load(open(str(run / "checkpoint")))["global_step"] == 7
    # GUNet through main_g: synthetic NF-shaped tensors (the NF data is private), moving-average best checkpoint
    run_g = tmp_path / "run_g"
    argv_g = ("nf_inter --mode train --tag g --model GUNet --classes NF --im_height 32 --im_width 32 --im_channel 3 --noise_scale 0 "
              "--num_of_steps 5 --primary_metric NF/Dice --loss_weight_type numerical --loss_numeric_w 1 6 --batches_per_epoch 2 "
              "--batch_size 2 --normalizer instance_norm --eval_num_batches_per_epoch 2 --eval_per_epoch --evaluator Volume "
              "--save_best --summary_prefix nf --use_spatial --guide_channel 1 --log_step 1").split() + ["--model_dir", str(run_g)]
    assert main_g.main(argv_g) == 0
    best = json.load(open(str(run_g / "best_result")))
    assert set(best) == {"ma_results", "ma_best_result"} and "NF/Dice" in best["ma_results"]
    assert os.path.exists(str(run_g / "checkpoint_best"))
