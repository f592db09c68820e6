"""GPU parity of the UNet3D plugin (BASELINE.json configs[4] at reduced size) against the CPU oracle: logits,
loss, gradients in the logical (TF-shaped, un-padded) variables, every conv3d / deconv3d backward on identical
operands, exactness of the channel padding, checkpoint round trip, training."""
import argparse

import numpy as np
import pytest
import torch

from oracle import tf_ops, unet3d
from test_gpu_unet import rel

pytestmark = pytest.mark.gpu

YML = dict(init_channels=30, max_channels=320, num_pool_layers=4, ret_prob=False, ret_pred=True, build_metrics=True,
           build_summaries=False)


def make_args(**over):
    a = argparse.Namespace(
        classes=["NF"], batch_size=2, num_gpus=1, im_depth=4, im_height=32, im_width=32, im_channel=1,
        normalizer="instance_norm", without_norm=False, weight_init="xavier", weight_decay_rate=3e-5, bias_decay=False,
        loss_type="xentropy", loss_weight_type="numerical", loss_numeric_w=[1.0, 1.0], loss_proportion_decay=1000,
        metrics_train=["Dice"], img_grad=False, tag="test3d", seed=1234, use_spatial=False, guide_channel=2,
        learning_rate=3e-4, learning_policy="period_step", lr_decay_step=100000, lr_decay_rate=0.1,
        num_of_total_steps=1000, lr_power=0.9, lr_end=1e-6, lr_decay_boundaries=None, lr_custom_values=None,
        optimizer="Adam", eval_per_epoch=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def setup(args):
    from boxsegliver_amd.NetworksV2.UNet3D import UNet3D
    from boxsegliver_amd.data.synthetic import make_batch_3d
    images, labels, _ = make_batch_3d(2, args.im_depth, args.im_height, args.im_width, 1, 2, 1234)
    model = UNet3D(args)
    inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda()}
    model(inputs, "eval", **YML)
    net = unet3d.UNet3DOracle(1, 2, normalizer=args.normalizer)
    assert [(n, tuple(s), k) for n, s, k in net.specs] == [(n, tuple(s), k) for n, s, k in model.params.logical_specs]
    params = unet3d.init_params(net.specs, seed=5)
    g = torch.Generator().manual_seed(9)
    for name, _, kind in net.specs:
        if kind == "gamma":
            params[name] = 0.5 + torch.rand(params[name].shape, generator=g)
        elif kind in ("beta", "bias"):
            params[name] = 0.1 * torch.randn(params[name].shape, generator=g)
    model.params.load_state(params)
    return model, inputs, net, params, (torch.from_numpy(images), torch.from_numpy(labels).long())


def kwargs_of(args):
    return dict(loss_type=args.loss_type, loss_weight_type=args.loss_weight_type, numeric_w=args.loss_numeric_w,
                weight_decay_rate=args.weight_decay_rate)


def _rel_t(got, ref):
    return ((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def check_conv3d_unit(c, dev="cpu"):
    """One conv3d + norm + ReLU unit's backward in float64 on the operands the device used (dev = "cuda": the float64
    evaluation itself runs on the GPU through the oracle's torch ops -- the checker for full-size layers)."""
    d64 = lambda t: None if t is None else t.detach().to(dev).double().requires_grad_(True)
    to = lambda t: t.detach().to(dev)
    y, g, b = d64(c["y"]), d64(c["gamma"]), d64(c["beta"])
    tol = 5e-4 if (c["per_sample"] and y.shape[1] * y.shape[2] * y.shape[3] <= 16) else 2e-5
    if c.get("plain"):                  # --without_norm: conv + bias + ReLU
        z = y + b
    elif c["per_sample"]:
        z = tf_ops.instance_norm(y, g, b, eps=1e-6)
    else:
        z, _, _ = tf_ops.batch_norm(y, g, b, torch.zeros(y.shape[-1], dtype=torch.float64, device=dev),
                                    torch.ones(y.shape[-1], dtype=torch.float64, device=dev), True)
    if c.get("z") is not None:          # the device's own ReLU mask (see test_gpu_unet.check_unit_backward)
        (z * (to(c["z"]) > 0).double()).backward(to(c["dz"]).double())
    else:
        torch.relu(z).backward(to(c["dz"]).double())
    assert _rel_t(to(c["dy"]), y.grad) < tol
    if g is not None:
        assert _rel_t(to(c["dgamma"]), g.grad) < tol
    assert _rel_t(to(c["dbeta"]), b.grad) < tol
    x = to(c["x"]).double().contiguous().requires_grad_(True)
    w = to(c["w"]).double().requires_grad_(True)
    tf_ops.conv_nd_same(x, w, stride=c["stride"]).backward(to(c["dy"]).double())
    assert _rel_t(to(c["dw"]), w.grad) < 2e-5
    if c["dx"] is not None:
        assert _rel_t(to(c["dx"]), x.grad) < 2e-5


def check_deconv3d(c, dev="cpu"):
    to = lambda t: t.detach().to(dev)
    x = to(c["x"]).double().requires_grad_(True)
    w = to(c["w"]).double().requires_grad_(True)
    coff, kd = c["coff"], c["kd"]
    pre = tf_ops.conv_transpose_ks(x, w, (kd, 2, 2))
    mask = (to(c["cat"][..., coff:]) > 0).double()
    (pre * mask).backward(to(c["dcat"][..., coff:]).double())
    assert _rel_t(to(c["dx"]), x.grad) < 2e-5
    assert _rel_t(to(c["dw"]), w.grad) < 2e-5


def test_unet3d_matches_oracle_and_padding_is_exact():
    from boxsegliver_amd import ops
    args = make_args()
    model, inputs, net, params, (images, labels) = setup(args)
    assert model.params.num_trainable() == 14535412                     # SURVEY.md 8a (a17), TF-shaped variables
    total, _, logits, grads, _ = net.loss_and_grads(params, images, labels, **kwargs_of(args))
    p64 = {k: v.double() for k, v in params.items()}
    _, _, _, grads64, _ = net.loss_and_grads(p64, images.double(), labels, **kwargs_of(args))
    ops.DEBUG_CAPTURE = []
    try:
        model.params.zero_grad()
        loss = model(inputs, "train", **YML)
        loss.backward()
        torch.cuda.synchronize()
        captured = ops.DEBUG_CAPTURE
    finally:
        ops.DEBUG_CAPTURE = None
    assert abs(loss.item() - total.item()) < 1e-4 * max(1.0, abs(total.item()))
    got = model.layers["logits"].cpu().numpy()
    assert np.abs(got - logits.numpy()).max() < 1e-3
    srt = np.sort(logits.numpy(), -1)
    safe = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert (got.argmax(-1) == logits.numpy().argmax(-1))[safe].all()
    convs = [c for c in captured if c["kind"] == "conv3d"]
    deconvs = [c for c in captured if c["kind"] == "deconv3d"]
    assert len(convs) == 18 and len(deconvs) == 4
    for c in convs:
        check_conv3d_unit(c)
    for c in deconvs:
        check_deconv3d(c)
    # gradients in logical shapes; the padded entries of every device gradient are exactly zero
    num = den = 0.0
    for name in model.params.trainable_names():
        g = model.params.logical_grad(name).numpy().astype(np.float64)
        ref = grads64[name].numpy()
        num += np.sum((g - ref) ** 2)
        den += np.sum(ref ** 2)
        phys = model.params[name].grad
        assert abs(float(phys.double().abs().sum()) - float(np.abs(g).sum())) <= 1e-6 * max(1.0, float(np.abs(g).sum()))
    assert (num / den) ** 0.5 < 1e-2
    assert model.metrics_dict["NF/Dice"].item() >= 0.0


def test_unet3d_step_is_bit_reproducible_at_a_many_tile_size():
    """A 16 x 64 x 64 patch (every level walks many tiles / linear blocks / parity classes): two runs of the same step
    give bit-identical loss and gradients."""
    from boxsegliver_amd.NetworksV2.UNet3D import UNet3D
    from boxsegliver_amd.data.synthetic import make_batch_3d
    args = make_args(batch_size=1, im_depth=16, im_height=64, im_width=64)
    images, labels, _ = make_batch_3d(1, 16, 64, 64, 1, 2, 77)
    inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda()}
    model = UNet3D(args)
    runs = []
    for _ in range(2):
        if model.params is not None:
            model.params.zero_grad()
        loss = model(inputs, "train", **YML)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.item(), model.params.grad["reg"].clone(), model.params.grad["noreg"].clone()))
    assert np.isfinite(runs[0][0]) and runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_unet3d_checkpoint_roundtrip_and_training():
    from boxsegliver_amd.core.solver import Solver
    args = make_args()
    model, inputs, net, params, _ = setup(args)
    sd = model.params.state_dict()
    assert sd["UNet3D/conv_e0/conv1/weights"].shape == (1, 3, 3, 1, 30)
    assert sd["UNet3D/conv_d0/conv1/weights"].shape == (1, 3, 3, 60, 30)        # concat(skip 30, up 30)
    assert sd["UNet3D/conv_d3/up/weights"].shape == (2, 2, 2, 240, 320) and "UNet3D/conv_d3/up/biases" not in sd
    assert model.params["UNet3D/conv_d0/conv1/weights"].shape == (1, 3, 3, 64, 32)   # device layout
    for k, v in params.items():
        assert torch.equal(sd[k], v), k
    solver = Solver(args)
    first = None
    for _ in range(4):
        loss = model(inputs, "train", **YML)
        first = loss.item() if first is None else first
        solver(loss, model)
    assert model(inputs, "train", **YML).item() < first
    # padded weights stay exactly zero under Adam + L2
    w = model.params["UNet3D/conv_d0/conv1/weights"]
    assert float(w[..., 30:32, :].abs().sum()) == 0.0 and float(w[..., 62:64, :].abs().sum()) == 0.0
    assert float(w[..., 30:32].abs().sum()) == 0.0
    model(inputs, "eval", **YML)
    assert model.probability.shape == (2, 4, 32, 32, 2)
    assert model.predictions["NFPred"].dtype == torch.uint8
