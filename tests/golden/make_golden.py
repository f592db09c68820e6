"""Generate tests/golden/*.npz from this repo's CPU oracle (float64).  Run from the repo root:

    python tests/golden/make_golden.py

The reference cannot be executed here (TensorFlow 1.13 is not installable), so these fixtures are
NOT reference outputs: they freeze the oracle's answers (parity stays "unpinned", oracle/__init__.py) so that
(1) the oracle cannot drift silently (tests/test_golden.py, CPU) and (2) the GPU box checks the HIP path against
committed numbers, not only against an oracle evaluated on that box (tests/test_gpu_golden.py).
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import golden_common as gc                                  # noqa: E402
from oracle import gunet2d, lgnet2d, losses, naive, smallunet2d, solver, tf_ops, unet2d, unet3d   # noqa: E402


def oracle_net(case):
    c = gc.CASES[case]
    if c["kind"] == "UNet":
        return unet2d.UNet2DOracle(3, 3, normalizer=c["normalizer"])
    if c["kind"] == "GUNet":
        return gunet2d.GUNet2DOracle(3, 3, guide_channel=1, normalizer=c["normalizer"])
    if c["kind"] == "LGNet":
        return lgnet2d.LGNetOracle(3, 3, guide_channel=1, mod_layers=c["mod_layers"], normalizer=c["normalizer"])
    if c["kind"] == "SmallUNet":
        return smallunet2d.SmallUNetOracle(4, 3, factor=c["factor"], normalizer=c["normalizer"])
    return unet3d.UNet3DOracle(1, 2, normalizer=c["normalizer"])


def run(net, case, params, inputs, dtype):
    c = gc.CASES[case]
    p = OrderedDict((k, torch.from_numpy(v).to(dtype)) for k, v in params.items())
    images = torch.from_numpy(inputs["images"]).to(dtype)
    labels = torch.from_numpy(inputs["labels"]).long()
    kw = dict(loss_type=c["loss_type"], loss_weight_type=c["w_type"], numeric_w=gc.NUMERIC_W[c["kind"]],
              weight_decay_rate=gc.WD[c["kind"]])
    if c["kind"] in ("GUNet", "LGNet"):
        return net.loss_and_grads(p, images, torch.from_numpy(inputs["sp_guide"]).to(dtype), labels, **kw)
    if c["kind"] == "SmallUNet":                                    # input = concat(images, sp_guide), SmallUNet.py:97
        return net.loss_and_grads(p, torch.cat((images, torch.from_numpy(inputs["sp_guide"]).to(dtype)), -1), labels, **kw)
    return net.loss_and_grads(p, images, labels, **kw)


def make_case(case):
    net = oracle_net(case)
    inputs = gc.make_inputs(case)
    params = gc.build_params(net.specs, seed=2024)
    total, data_loss, logits, grads, new_stats = run(net, case, params, inputs, torch.float64)
    probs = torch.softmax(logits, -1)
    out = dict(inputs)
    out["param_checksum"] = gc.checksum(params)
    out["total_loss"] = np.float64(total.item())
    out["data_loss"] = np.float64(data_loss.item())
    out["logits"] = logits.numpy().astype(np.float32)
    out["argmax"] = logits.numpy().argmax(-1).astype(np.uint8)
    srt = np.sort(logits.numpy(), -1)
    out["argmax_margin"] = (srt[..., -1] - srt[..., -2]).astype(np.float32)
    out["pred"] = (probs.numpy()[..., 1:] > 0.5).astype(np.uint8)
    names = list(grads.keys())
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([float(grads[n].norm()) for n in names])
    out["grad_sums"] = np.array([float(grads[n].sum()) for n in names])
    small = [n for n in names if grads[n].numel() <= 64]
    out["small_grad_names"] = np.array(small)
    for i, n in enumerate(small):
        out["small_grad_%d" % i] = grads[n].numpy()
    stat_names = list(new_stats.keys())[:4]
    out["stat_names"] = np.array(stat_names)
    for i, n in enumerate(stat_names):
        out["stat_%d" % i] = new_stats[n].numpy().astype(np.float64)
    # 3-step TF-Adam trajectory (float64 oracle, moving statistics irrelevant in train mode)
    adam = solver.TFAdam()
    p = OrderedDict((k, v.astype(np.float64)) for k, v in params.items())
    traj = []
    for _ in range(3):
        t, _, _, g, _ = run(net, case, p, inputs, torch.float64)
        traj.append(t.item())
        adam.step(p, {k: v.numpy() for k, v in g.items()}, gc.LR)
    out["adam_traj"] = np.array(traj)
    np.savez_compressed(os.path.join(gc.HERE, case + ".npz"), **out)
    print(case, "loss", out["total_loss"], "traj", traj, "vars", len(names))


def make_kat():
    """Op-level vectors: inputs and the answers of the pure-numpy loop restatement
    (oracle/naive.py) or closed forms -- independent of tf_ops/torch."""
    rng = np.random.default_rng(7)
    out = {}
    x = rng.standard_normal((2, 6, 5, 3))
    w = rng.standard_normal((3, 3, 3, 4))
    out["conv_x"], out["conv_w"] = x, w
    out["conv_s1"] = naive.conv2d_same(x, w)
    x2 = rng.standard_normal((1, 6, 8, 2))
    out["conv2_x"] = x2
    out["conv2_w"] = w2 = rng.standard_normal((3, 3, 2, 3))
    out["conv_s2_even"] = naive.conv2d_same(x2, w2, (2, 2))          # pad 0 before / 1 after
    dx = rng.standard_normal((2, 3, 2, 4))
    dw = rng.standard_normal((2, 2, 3, 4))                            # [kh,kw,Cout,Cin]
    db = rng.standard_normal(3)
    out["deconv_x"], out["deconv_w"], out["deconv_b"] = dx, dw, db
    out["deconv_relu"] = np.maximum(naive.conv2d_transpose_k2s2(dx, dw, db), 0.0)
    bx = rng.standard_normal((3, 4, 4, 5)) * 2 + 1
    g, b = rng.random(5) + 0.5, rng.standard_normal(5)
    out["bn_x"], out["bn_gamma"], out["bn_beta"] = bx, g, b
    out["bn_train"] = naive.batch_norm_train(bx, g, b, 1e-3)[0]
    m = bx.reshape(-1, 5).mean(0)
    v = bx.reshape(-1, 5).var(0)
    n = bx.size // 5
    out["bn_moving_mean"] = 0.999 * np.zeros(5) + 0.001 * m
    out["bn_moving_var"] = 0.999 * np.ones(5) + 0.001 * v * n / (n - 1)
    mi = bx.mean((1, 2), keepdims=True)
    vi = bx.var((1, 2), keepdims=True)
    out["in_out"] = (bx - mi) / np.sqrt(vi + 1e-6) * g + b
    px = rng.standard_normal((2, 4, 6, 3))
    out["pool_x"], out["pool_max"] = px, naive.max_pool2x2(px)
    z = rng.standard_normal((2, 4, 4, 3))
    lab = rng.integers(0, 3, size=(2, 4, 4))
    nw = np.array([0.0, 0.4, 4.4])                                   # a zero class weight exercises the NONZERO count
    wmap = nw[lab]
    wmap = wmap / wmap.sum((1, 2), keepdims=True) * 16.0               # loss_metrics.py:163-165
    out["xent_logits"], out["xent_labels"], out["xent_numeric_w"], out["xent_w"] = z, lab, nw, wmap
    out["xent_loss"] = np.float64(naive.weighted_xent(z, lab, wmap))
    pz = naive.softmax(z)
    onehot = np.eye(3)[lab][..., 1:]
    inter = (onehot * pz[..., 1:]).sum((1, 2, 3))
    union = (onehot + pz[..., 1:]).sum((1, 2, 3))
    out["dice_loss"] = np.float64(1.0 - np.mean(2 * inter / (union + 1e-8)))
    th, gr, lr = rng.standard_normal(6), rng.standard_normal(6), 1e-3
    out["adam_theta"], out["adam_grad"] = th, gr
    mm, vv = 0.1 * gr, 0.01 * gr * gr
    out["adam_step1"] = th - lr * np.sqrt(1 - 0.99) / (1 - 0.9) * mm / (np.sqrt(vv) + 1e-8)
    np.savez_compressed(os.path.join(gc.HERE, "kat_ops.npz"), **out)
    print("kat_ops", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]                                             # e.g. `make_golden.py lgnet_in_xent` adds one case
    if not only:
        make_kat()
    for case in only or gc.CASES:
        make_case(case)
