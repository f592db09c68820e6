"""GPU: the volume evaluator (reference evaluators/evaluator_liver.py:616-766,906-996): device-side
mirror TTA == the reference's pipeline-side mirroring, and the whole case pipeline (slabs -> averaged probabilities
-> argmax -> post-processing -> metric_3d / global Dice) against a numpy + oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import unet2d

pytestmark = pytest.mark.gpu


def test_flip_axpy_exact():
    from boxsegliver_amd import ops
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((3, 6, 10, 5)).astype(np.float32)).cuda()
    for fh in (False, True):
        for fw in (False, True):
            dims = [d for d, f in ((1, fh), (2, fw)) if f]
            ref = torch.flip(x, dims) if dims else x
            assert torch.equal(ops.flip_axpy(x, None, fh, fw), ref)
    acc = torch.ones_like(x)
    ops.flip_axpy(x, acc, True, False, 0.25, accumulate=True)
    assert torch.allclose(acc, 1 + 0.25 * torch.flip(x, [1]), atol=1e-7)


def _setup(eval_mirror, random_flip, pipeline_mirror):
    import test_gpu_unet as t
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.data import synthetic
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    args = t.make_args(batch_size=2, im_height=32, im_width=32, eval_mirror=eval_mirror, random_flip=random_flip,
                       metrics_eval=["Dice", "VOE", "RVD", "ASSD"], use_global_dice=False, pred_type="pred", mode="eval",
                       eval_num=-1, save_path=None)
    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (),
              "eval_cases": [(7, 5), (8, 4)], "pipeline_mirror": pipeline_mirror}
    evaluator = ev.get_evaluator("Volume", estimator=None, model_dir=".", params=params)
    model = evaluator._model()
    # create the variables, then load known weights with non-trivial moving statistics
    feats = next(f for f, _ in synthetic.input_fn_eval_volumes("eval", params) if f)
    evaluator._forward(model, feats)
    net, oparams = t.oracle_for(args)
    g = torch.Generator().manual_seed(3)
    for name, _, kind in net.specs:
        if kind == "moving_mean":
            oparams[name] = 0.1 * torch.randn(oparams[name].shape, generator=g)
        elif kind == "moving_var":
            oparams[name] = 0.5 + torch.rand(oparams[name].shape, generator=g)
    model.params.load_state(oparams)
    return evaluator, params, net, oparams, args


def _reference_volumes(params, net, oparams, args, variants, div):
    """numpy / oracle restatement of evaluator_liver.py:616-678 for the synthetic cases."""
    from boxsegliver_amd.data import synthetic
    out = []
    for pid, depth in params["eval_cases"]:
        images, labels, _ = synthetic.make_batch(depth, 32, 32, 3, 3, int(args.seed) + int(pid))
        x = torch.from_numpy(images)
        logits, _ = net.forward(oparams, x, False)
        prob = torch.softmax(logits, -1).numpy() / div
        for m in variants:
            axes = {1: (2,), 2: (1,), 3: (2, 1)}[m]
            lg, _ = net.forward(oparams, torch.from_numpy(np.ascontiguousarray(np.flip(images, axis=axes))), False)
            prob = prob + np.flip(torch.softmax(lg, -1).numpy(), axis=axes) / div
        out.append((prob, labels))
    return out


@pytest.mark.parametrize("random_flip", [3, 1])
def test_volume_evaluator_device_mirror_equals_pipeline_mirror_and_oracle(random_flip):
    from boxsegliver_amd.data import synthetic
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    from boxsegliver_amd import loss_metrics as metric_ops
    res = {}
    for pipeline_mirror in (False, True):
        evaluator, params, net, oparams, args = _setup(True, random_flip, pipeline_mirror)
        res[pipeline_mirror] = evaluator.run(synthetic.input_fn_eval_volumes, checkpoint_path=None)
        assert evaluator.calls == 2
    assert set(res[False]) == {"Liver/Dice", "Liver/VOE", "Liver/RVD", "Liver/ASSD", "Tumor/Dice", "Tumor/VOE",
                               "Tumor/RVD", "Tumor/ASSD", "GLiverDice", "GTumorDice"}
    for k in res[False]:
        assert res[False][k] == pytest.approx(res[True][k], abs=1e-9), k     # same masks either way
    # against the restatement: same argmax volumes wherever the averaged probabilities are not tied at rounding level
    variants, div = ev.mirror_plan(args)
    assert (variants, div) == (([1, 2, 3], 4) if random_flip == 3 else ([1, 3], 2))
    evaluator, params, net, oparams, args = _setup(True, random_flip, False)
    got = []
    gen = evaluator._predict_case(_stream(evaluator, params), dtype="pred", resize=True)
    for (case, seg, volume, _), (prob, labels) in zip(gen, _reference_volumes(params, net, oparams, args, variants, div)):
        srt = np.sort(prob, -1)
        safe = (srt[..., -1] - srt[..., -2]) > 1e-5
        assert volume.dtype == np.uint8 and volume.shape == labels.shape
        assert (volume == prob.argmax(-1))[safe].all() and safe.mean() > 0.999
        np.testing.assert_array_equal(seg, labels)
        got.append(volume)
    # metrics of the evaluator == metric_3d on the restated post-processing of those volumes
    ref = {}
    for volume, (_, labels) in zip(got, _reference_volumes(params, net, oparams, args, variants, div)):
        v, l = evaluator._postprocess(volume), evaluator._postprocess(labels.astype(np.uint8), is_label=True)
        for cls in ("Liver", "Tumor"):
            for met, val in metric_ops.metric_3d(v[cls], l[cls], required=["Dice", "VOE", "RVD", "ASSD"]).items():
                ref.setdefault("{}/{}".format(cls, met), []).append(val)
    for k, vals in ref.items():
        assert res[False][k] == pytest.approx(float(np.mean(vals)), abs=1e-9), k


def _stream(evaluator, params):
    from boxsegliver_amd.data import synthetic
    model = evaluator._model()
    for features, labels in synthetic.input_fn_eval_volumes("eval", params):
        if features:
            out = {k: v for k, v in features.items() if not torch.is_tensor(v)}
            out["Prob"] = evaluator._slab_probability(model, features)
            yield out, None
        else:
            yield None, labels


def test_training_with_hooks_evaluates_online_and_saves_the_best_checkpoint(tmp_path):
    """entry/main.py:163-186's hook set on the estimator: learning-rate log, ReduceLROnPlateau, and an EvaluatorHook
    that evaluates the `eval_online` batches every 2 steps with the live variables (moving statistics) and keeps the
    best checkpoint under `checkpoint_best` + `best_result`."""
    import json
    import os
    import test_gpu_unet as t
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import hooks, models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data import synthetic
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    args = t.make_args(batch_size=2, im_height=32, im_width=32, eval_per_epoch=True, learning_policy="plateau",
                       use_global_dice=False, eval_3d=False, metrics_eval=["Dice"], model="UNet", noise_scale=0.05,
                       synthetic_batches=2, eval_num_batches_per_epoch=2, log_step=1)
    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "solver": Solver(args),
              "solver_kwargs": {}}
    e = est.CustomEstimator(models.model_fn, str(tmp_path), est.RunConfig(model_dir=str(tmp_path),
                                                                         save_checkpoints_steps=0), params)
    evaluator = ev.get_evaluator("Volume", estimator=e)
    lr_hook = hooks.LogLearningRateHook("Liver", every_n_steps=1, do_logging=False)
    plateau = hooks.ReduceLROnPlateauHook(str(tmp_path), lr_patience=0, tr_patience=50, every_n_steps=1, min_delta=10.0)
    eval_hook = hooks.EvaluatorHook(evaluator, checkpoint_dir=str(tmp_path), eval_n_steps=2, save_best=True,
                                    compare_fn=lambda c, o: ev._compare(c, o, primary_metric="Liver/Dice"))
    e.train(synthetic.input_fn, steps=5, hooks=[lr_hook, plateau, eval_hook])
    assert len(lr_hook.records) == 5
    assert lr_hook.records[-1][1] < lr_hook.records[0][1]            # min_delta 10: never "improves" -> lr decays
    assert os.path.exists(str(tmp_path / "lr_schedule")) and os.path.exists(str(tmp_path / "checkpoint_best"))
    best = json.load(open(str(tmp_path / "best_result")))
    assert set(best) == {"Liver/Dice", "Liver/VOE", "Liver/VD", "Tumor/Dice", "Tumor/VOE", "Tumor/VD"}
    assert len(eval_hook.summaries) >= 2 and all(0.0 <= s[1]["Eval/Liver/Dice"] <= 1.0 for s in eval_hook.summaries)
    status = json.load(open(str(tmp_path / "checkpoint_best")))
    assert os.path.exists(str(tmp_path / status["model_checkpoint_path"]))


def test_estimator_evaluate_writes_results(tmp_path):
    """CustomEstimator.evaluate (core/estimator.py:263-279) delegates to the evaluator and dumps eval_results_2d.txt."""
    import json
    import test_gpu_unet as t
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.core import estimator as est
    from boxsegliver_amd.core import models
    from boxsegliver_amd.data import synthetic
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    args = t.make_args(batch_size=2, im_height=32, im_width=32, eval_mirror=False, random_flip=0,
                       metrics_eval=["Dice"], use_global_dice=True, pred_type="pred", mode="eval", eval_num=1,
                       save_path=None, eval_3d=False, model="UNet", model_dir=str(tmp_path))
    params = {"args": args, "model": UNet, "model_kwargs": dict(t.YML), "model_args": (), "eval_cases": [(3, 4), (4, 4)]}
    e = est.CustomEstimator(models.model_fn, str(tmp_path), est.RunConfig(model_dir=str(tmp_path)), params)
    results = e.evaluate(ev.get_evaluator("Volume", estimator=e), synthetic.input_fn_eval_volumes)
    assert set(results) == {"LiverDice", "TumorDice"} and 0.0 <= results["LiverDice"] <= 1.0
    assert json.load(open(str(tmp_path / "eval_results_2d.txt"))) == results
