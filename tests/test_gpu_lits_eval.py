"""GPU: offline volume evaluation over NIfTI cases (data/lits.input_fn_eval -> evaluators.EvaluateVolume): the
reference's `entry/main.py --mode eval` path (DataLoader/Liver/input_pipeline.py:615-668 feeding
evaluators/evaluator_liver.py:616-766).  The evaluator's result must equal a plain re-statement of the same loop
(forward per slab, mirrored slabs un-flipped and averaged, argmax, zoom back to the crop, merge / largest component,
metric_3d) driven by the same model."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("eval_mirror", [False, True])
def test_lits_nifti_evaluation_matches_manual_loop(tmp_path, eval_mirror):
    import test_gpu_unet as t
    from test_lits_eval_host import _write_dataset
    from boxsegliver_amd import loss_metrics as metric_ops
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.data import lits
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    from boxsegliver_amd.utils import array_kits as arr_ops
    _write_dataset(tmp_path, depth=9, size=96)
    args = t.make_args(batch_size=4, im_height=64, im_width=64, eval_mirror=eval_mirror, random_flip=3,
                       metrics_eval=["Dice", "VOE", "RVD"], use_global_dice=False, pred_type="pred", mode="eval", eval_num=-1,
                       save_path=None, test_fold=2, filter_size=0, eval_skip_num=0, eval_in_patches=False, model="UNet")
    yml = dict(t.YML, num_down_samples=3)
    params = {"args": args, "model": UNet, "model_kwargs": yml, "model_args": (), "lits_root": tmp_path, "proj_root": tmp_path}
    evaluator = ev.get_evaluator("Volume", estimator=None, model_dir=str(tmp_path), params=params)
    results = evaluator.run(lits.input_fn_eval, checkpoint_path=None)
    assert evaluator.calls == 2                                            # fold 2: cases 2 and 5
    for key in ("Liver/Dice", "Liver/VOE", "Tumor/Dice", "GLiverDice", "GTumorDice"):
        assert key in results and np.isfinite(results[key]), key

    # the same loop, written out
    model = evaluator._model()
    div = 4.0 if eval_mirror else 1.0
    per_case, slabs = [], []
    for feats, labels in lits.input_fn_eval("eval", params):
        if feats is not None:
            x = torch.from_numpy(np.ascontiguousarray(feats["images"])).cuda()
            model({"images": x}, "eval", **yml)
            prob = model.probability.cpu().numpy() / div
            m = feats["mirror"]
            if m == 0:
                slabs.append(prob)
            else:
                axes = {1: (2,), 2: (1,), 3: (1, 2)}[m]
                slabs[-1] = slabs[-1] + np.flip(prob, axis=axes)
        else:
            seg, _, pads, bbox, resized = labels
            vol = np.concatenate(slabs)
            slabs = []
            if pads > 0:
                vol = vol[:-pads]
            vol = np.argmax(vol, -1).astype(np.uint8)
            ori = (vol.shape[0], bbox[4] - bbox[1] + 1, bbox[3] - bbox[0] + 1)
            if resized and ori != vol.shape:
                vol = ndi.zoom(vol, np.array(ori) / np.array(vol.shape), order=0)
            assert vol.shape == seg.shape
            pred = {"Liver": (vol == 1) | (vol == 2), "Tumor": vol == 2}
            pred["Liver"] = arr_ops.get_largest_component(pred["Liver"], rank=3)
            pred["Tumor"] = pred["Tumor"] * pred["Liver"].astype(pred["Tumor"].dtype)
            lab = {"Liver": (seg == 1) | (seg == 2), "Tumor": seg == 2}
            per_case.append({"{}/{}".format(c, k): v for c in ("Liver", "Tumor")
                             for k, v in metric_ops.metric_3d(pred[c], lab[c], required=["Dice", "VOE", "RVD"]).items()})
    assert len(per_case) == 2
    for key in per_case[0]:
        ref = float(np.mean([c[key] for c in per_case]))
        assert abs(results[key] - ref) < 1e-6 * max(1.0, abs(ref)), (key, results[key], ref)


def test_lits_eval_in_patches_matches_manual_loop(tmp_path):
    """--eval_in_patches: sliding 64 x 64 windows at native resolution over the liver box (input_pipeline.py:676-766,
    evaluator_liver.py:524-566): the evaluator equals the written-out loop (forward per window batch, windows written in
    order so the last one wins on overlaps, argmax, labels cropped to the box, merge / largest component, metric_3d)."""
    import test_gpu_unet as t
    from test_lits_eval_host import _write_dataset
    from boxsegliver_amd import loss_metrics as metric_ops
    from boxsegliver_amd.NetworksV2.UNet import UNet
    from boxsegliver_amd.data import lits
    from boxsegliver_amd.evaluators import evaluator_liver as ev
    from boxsegliver_amd.utils import array_kits as arr_ops
    _write_dataset(tmp_path, depth=9, size=128)
    args = t.make_args(batch_size=5, im_height=64, im_width=64, eval_mirror=False, random_flip=0,
                       metrics_eval=["Dice", "VOE"], use_global_dice=False, pred_type="pred", mode="eval", eval_num=-1,
                       save_path=None, test_fold=2, filter_size=0, eval_skip_num=0, eval_in_patches=True, model="UNet")
    yml = dict(t.YML, num_down_samples=3)
    params = {"args": args, "model": UNet, "model_kwargs": yml, "model_args": (), "lits_root": tmp_path, "proj_root": tmp_path}
    evaluator = ev.get_evaluator("Volume", estimator=None, model_dir=str(tmp_path), params=params)
    results = evaluator.run(lits.input_fn_eval, checkpoint_path=None)
    assert evaluator.calls == 2
    model = evaluator._model()
    per_case, result = [], None
    for feats, labels in lits.input_fn_eval("eval", params):
        bbox = feats["bbox"]
        if result is None:
            result = np.zeros(arr_ops.bbox_to_shape(bbox) + (3,), np.float32)
        model({"images": torch.from_numpy(feats["images"]).cuda()}, "eval", **yml)
        prob = model.probability.cpu().numpy()
        for i, (z, lb_y, ub_y, lb_x, ub_x) in enumerate(feats["position"][:5 - feats["pad"]]):
            result[z, lb_y:ub_y, lb_x:ub_x] = prob[i]
        if labels is None:
            continue
        vol = np.argmax(result, -1)
        seg = labels[bbox[2]:bbox[5] + 1, bbox[1]:bbox[4] + 1, bbox[0]:bbox[3] + 1]
        assert seg.shape == vol.shape
        result = None
        pred = {"Liver": (vol == 1) | (vol == 2), "Tumor": vol == 2}
        pred["Liver"] = arr_ops.get_largest_component(pred["Liver"], rank=3)
        pred["Tumor"] = pred["Tumor"] * pred["Liver"].astype(pred["Tumor"].dtype)
        lab = {"Liver": (seg == 1) | (seg == 2), "Tumor": seg == 2}
        per_case.append({"{}/{}".format(c, k): v for c in ("Liver", "Tumor")
                         for k, v in metric_ops.metric_3d(pred[c], lab[c], required=["Dice", "VOE"]).items()})
    assert len(per_case) == 2
    for key in per_case[0]:
        ref = float(np.mean([c[key] for c in per_case]))
        assert abs(results[key] - ref) < 1e-6 * max(1.0, abs(ref)), (key, results[key], ref)
