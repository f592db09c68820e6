"""GPU: the plugins (HIP kernels through the C ABI) against the committed whole-net fixtures of
tests/golden/ -- logits within the north-star 1e-3, loss, argmax / thresholded masks bit-exact outside rounding-level
margins, per-variable gradient norms, small gradients in full, moving statistics, a 3-step Adam trajectory."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import golden_common as gc                                            # noqa: E402

pytestmark = pytest.mark.gpu


def build(case):
    c = gc.CASES[case]
    g = gc.load(case)
    over = dict(normalizer=c["normalizer"], loss_type=c["loss_type"], loss_weight_type=c["w_type"],
                loss_numeric_w=gc.NUMERIC_W[c["kind"]], weight_decay_rate=gc.WD[c["kind"]], learning_rate=gc.LR)
    if c["kind"] == "UNet":
        import test_gpu_unet as t
        from boxsegliver_amd.NetworksV2.UNet import UNet as Net
    elif c["kind"] == "GUNet":
        import test_gpu_gunet as t
        from boxsegliver_amd.NetworksV2.GUNet import GUNet as Net
    elif c["kind"] == "LGNet":
        import test_gpu_gunet as t
        from boxsegliver_amd.NetworksV2.LGNet import LGNet as Net
        over.update(use_spatial=True, guide_channel=1)
        yml = dict(mod_layers=c["mod_layers"], ret_prob=False, ret_pred=True, build_metrics=True, build_summaries=False)
    elif c["kind"] == "SmallUNet":
        import test_gpu_gunet as t
        from boxsegliver_amd.NetworksV2.SmallUNet import SmallUNet as Net
        over.update(use_spatial=True, guide_channel=1, im_height=c["size"], im_width=c["size"])
        yml = dict(init_channel_factor=c["factor"], num_pool_layers=3, ret_prob=False, ret_pred=True, build_metrics=True,
                   build_summaries=False)
    else:
        import test_gpu_unet3d as t
        from boxsegliver_amd.NetworksV2.UNet3D import UNet3D as Net
    if c["kind"] not in ("LGNet", "SmallUNet"):
        yml = t.YML
    args = t.make_args(**over)
    model = Net(args)
    inputs = {k: torch.from_numpy(g[k]).cuda() for k in ("images", "labels", "sp_guide") if k in g.files}
    model(inputs, "eval", **yml)
    specs = getattr(model.params, "logical_specs", model.params.specs)
    params = gc.build_params(specs, seed=2024)
    np.testing.assert_allclose(gc.checksum(params), g["param_checksum"], rtol=1e-12)
    model.params.load_state({k: torch.from_numpy(v) for k, v in params.items()})
    return model, inputs, args, yml, g


@pytest.mark.parametrize("case", list(gc.CASES))
def test_plugin_matches_fixture(case):
    model, inputs, args, yml, g = build(case)
    model.params.zero_grad()
    loss = model(inputs, "train", **yml)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["total_loss"])) < 1e-4 * max(1.0, abs(float(g["total_loss"])))
    logits = model.layers["logits"].cpu().numpy()
    assert np.abs(logits - g["logits"]).max() < 1e-3
    safe = g["argmax_margin"] > 1e-3
    assert (logits.argmax(-1) == g["argmax"])[safe].all() and safe.mean() > 0.97
    logical = hasattr(model.params, "logical_grad")
    names = [str(n) for n in g["grad_names"]]
    assert names == list(model.params.trainable_names())
    grad = lambda n: (model.params.logical_grad(n) if logical else model.params[n].grad.cpu()).double().numpy()
    got = np.array([np.linalg.norm(grad(n)) for n in names])
    assert np.abs(got - g["grad_norms"]).max() / g["grad_norms"].max() < 1e-2
    assert abs(np.linalg.norm(got) - np.linalg.norm(g["grad_norms"])) / np.linalg.norm(g["grad_norms"]) < 2e-3
    for i, n in enumerate(g["small_grad_names"]):
        ref = g["small_grad_%d" % i]
        assert np.linalg.norm(grad(str(n)) - ref) / max(np.linalg.norm(ref), 1e-30) < 2e-2, n
    for i, n in enumerate(g["stat_names"]):
        np.testing.assert_allclose(model.params[str(n)].cpu().numpy(), g["stat_%d" % i], rtol=1e-4, atol=1e-6)
    # eval-mode thresholded masks of the in-graph predictions use moving statistics -> compare train-mode probabilities
    prob = torch.softmax(model.layers["logits"], -1).cpu().numpy()
    sure = np.abs(prob[..., 1:] - 0.5) > 1e-3
    assert ((prob[..., 1:] > 0.5).astype(np.uint8) == g["pred"])[sure].all()


@pytest.mark.parametrize("case", list(gc.CASES))
def test_adam_trajectory_matches_fixture(case):
    from boxsegliver_amd.core.solver import Solver
    model, inputs, args, yml, g = build(case)
    solver = Solver(args)
    traj = []
    for _ in range(3):
        loss = model(inputs, "train", **yml)
        traj.append(loss.item())
        solver(loss, model)
    np.testing.assert_allclose(traj, g["adam_traj"], rtol=2e-3)
    assert abs(traj[0] - g["adam_traj"][0]) < 1e-4 * max(1.0, abs(g["adam_traj"][0]))
