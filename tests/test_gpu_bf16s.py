"""GPU: UNETK_BF16S -- bf16 matrix cores AND bf16 storage of activations / activation gradients (BASELINE.json configs[2]:
"UNet 512x512x3 bs=64 bf16"; SURVEY.md 8d: "bf16 activations / weights-compute, fp32 master / accum / stats").

Op level: products of bf16 values are exact in fp32 and the accumulation is fp32, so against a float64 evaluation of the
SAME bf16 operands a kernel's fp32 accumulator agrees to ~1e-6; what is stored is that accumulator rounded to bf16, so the
stored value must be within half a bf16 ulp (2^-9 relative) of the float64 result -- plus the fp32 accumulation error,
which can move a value sitting on a rounding boundary to the neighbouring bf16 (counted, and bounded).
Whole net: against the oracle restating the same arithmetic AND the same storage roundings (oracle/tf_ops.py
store_bf16 / norm_relu_bf16s)."""
import math

import numpy as np
import pytest
import torch

from oracle import tf_ops

pytestmark = pytest.mark.gpu

ULP = 2.0 ** -8          # bf16: 8 significand bits -> spacing 2^-7 .. 2^-8 relative; half an ulp <= 2^-8 relative


def _r(t):
    """fp32 tensor -> the bf16 values it rounds to (RNE), as float64."""
    return t.float().bfloat16().double()


def _stored_ok(got_bf16, ref64, flips=2e-3):
    """got = round_bf16(fp32 accumulator), ref64 = the exact result.  Every element within one bf16 ulp; all but a
    `flips` share exactly the rounding of the exact result (the rest sat within fp32 accumulation error of a boundary)."""
    got = got_bf16.double()
    scale = ref64.abs().clamp_min(1e-30)
    err = (got - ref64).abs() / scale
    big = ref64.abs() > 1e-3 * ref64.abs().max()
    assert err[big].max().item() <= 1.01 * ULP, err[big].max().item()
    exact = (got == _r(ref64.float()))
    assert exact.double().mean().item() > 1.0 - flips, exact.double().mean().item()


CONV_SHAPES = [
    # N, H, W, Cin, Cout
    (1, 32, 32, 64, 128),      # 512x128 tile
    (2, 40, 20, 64, 128),      # masked edges in H and W
    (1, 8, 16, 64, 128),       # 128x128 (planes lower than 24 rows)
    (1, 64, 16, 64, 64),       # 256x64
    (3, 8, 48, 128, 64),       # 128x64
    (1, 4, 4, 64, 256),        # image smaller than a tile
    (1, 24, 16, 256, 256),     # 8 chunks, 2 N tiles
    (2, 16, 16, 1024, 512),    # bridge-sized channels
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_bf16s_conv_fwd_dgrad_wgrad(shape):
    import torch.nn.functional as F
    from boxsegliver_amd import _abi, ops
    n, h, w, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + cin + cout + h)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16()
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    dy = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16()
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = _r(wt).permute(3, 2, 0, 1).requires_grad_(True)
    y64 = F.conv2d(x64, w64, padding=1)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    wp_f, wp_d = ops.conv3x3_pack(wt, bf16=_abi.BF16S)
    y, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True, bf16=_abi.BF16S)
    assert y.dtype == torch.bfloat16
    ref = y64.detach().permute(0, 2, 3, 1)
    _stored_ok(y, ref)
    # statistics come from the fp32 accumulators (not from the rounded values)
    s = stats.double()
    assert s.shape == (2, rows, cout)
    tol = 3e-5 * ref.abs().sum((0, 1, 2)).max().item()
    assert (s[0].sum(0) - ref.sum((0, 1, 2))).abs().max().item() < tol
    assert ((s[1].sum(0) - (ref ** 2).sum((0, 1, 2))).abs() / (ref ** 2).sum((0, 1, 2))).max().item() < 3e-5
    dx = ops.conv3x3_dgrad(dy, wp_d, cin, bf16=_abi.BF16S)
    assert dx.dtype == torch.bfloat16
    _stored_ok(dx, x64.grad.permute(0, 2, 3, 1))
    dw = ops.conv3x3_wgrad(x, dy, bf16=_abi.BF16S)
    dw_ref = w64.grad.permute(2, 3, 1, 0)
    assert dw.dtype == torch.float32
    assert ((dw.double() - dw_ref).abs().max() / dw_ref.abs().max()).item() < 1e-5
    assert torch.equal(dw, ops.conv3x3_wgrad(x, dy, bf16=_abi.BF16S))          # bit-reproducible split-K


@pytest.mark.parametrize("shape", [(8, 128, 128, 64, 128), (4, 64, 64, 256, 256), (8, 256, 256, 64, 64), (2, 512, 512, 64, 64)])
def test_bf16s_conv_large_shapes(shape):
    """BASELINE-sized layers (many tiles per split-K block, several rounds of blocks)."""
    import torch.nn.functional as F
    from boxsegliver_amd import _abi, ops
    n, h, w, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(n + cin)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16()
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    dy = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16()
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = _r(wt).permute(3, 2, 0, 1).requires_grad_(True)
    y64 = F.conv2d(x64, w64, padding=1)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    wp_f, wp_d = ops.conv3x3_pack(wt, bf16=_abi.BF16S)
    y, _, _ = ops.conv3x3_fwd(x, wp_f, cout, want_stats=False, bf16=_abi.BF16S)
    _stored_ok(y, y64.detach().permute(0, 2, 3, 1))
    del y, y64
    dx = ops.conv3x3_dgrad(dy, wp_d, cin, bf16=_abi.BF16S)
    _stored_ok(dx, x64.grad.permute(0, 2, 3, 1))
    dw = ops.conv3x3_wgrad(x, dy, bf16=_abi.BF16S)
    dw_ref = w64.grad.permute(2, 3, 1, 0)
    assert ((dw.double() - dw_ref).abs().max() / dw_ref.abs().max()).item() < 2e-5
    assert torch.equal(dw, ops.conv3x3_wgrad(x, dy, bf16=_abi.BF16S))


def test_bf16s_first_layer_fp32_image_in_bf16_out():
    """Encode1/conv1 (Cin = 3): the direct kernel reads the fp32 image and stores bf16; its filter gradient contracts the
    fp32 image with the bf16 dy on the exact-fp32 matrix cores."""
    import torch.nn.functional as F
    from boxsegliver_amd import _abi, ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((2, 40, 52, 3), device="cuda", generator=g)
    wt = torch.randn((3, 3, 3, 64), device="cuda", generator=g) / math.sqrt(27)
    dy = torch.randn((2, 40, 52, 64), device="cuda", generator=g).bfloat16()
    x64 = x.double().permute(0, 3, 1, 2)
    w64 = wt.double().permute(3, 2, 0, 1).requires_grad_(True)
    y64 = F.conv2d(x64, w64, padding=1)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    y, stats, rows = ops.conv3x3_fwd(x, wt, 64, want_stats=True, bf16=_abi.BF16S)
    assert y.dtype == torch.bfloat16
    _stored_ok(y, y64.detach().permute(0, 2, 3, 1))
    ref = y64.detach().permute(0, 2, 3, 1)
    assert (stats.double()[0].sum(0) - ref.sum((0, 1, 2))).abs().max().item() < 1e-4 * ref.abs().sum((0, 1, 2)).max().item()
    dw = ops.conv3x3_wgrad(x, dy, bf16=_abi.BF16S)
    dw_ref = w64.grad.permute(2, 3, 1, 0)
    assert ((dw.double() - dw_ref).abs().max() / dw_ref.abs().max()).item() < 1e-5


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("C", [64, 256, 1024])
def test_bf16s_norm_apply_and_backward(per_sample, C):
    """norm.hip with bf16 tensors against the float64 restatement of the same formulas on the same bf16 inputs."""
    from boxsegliver_amd import ops
    n, h, w = 3, 12, 20
    g = torch.Generator(device="cuda").manual_seed(C + per_sample)
    yacc = torch.randn((n, h, w, C), device="cuda", generator=g) * 1.5 + 0.3
    y = yacc.bfloat16()
    gamma = 0.5 + torch.rand(C, device="cuda", generator=g)
    beta = 0.2 * torch.randn(C, device="cuda", generator=g)
    dz = torch.randn((n, h, w, C), device="cuda", generator=g).bfloat16()
    # statistics of the UNROUNDED accumulators, as the conv epilogue provides them
    axes = (1, 2) if per_sample else (0, 1, 2)
    mean = yacc.double().mean(dim=axes, keepdim=per_sample)
    var = yacc.double().var(dim=axes, unbiased=False, keepdim=per_sample)
    eps = 1e-6 if per_sample else 1e-3
    rstd = torch.rsqrt(var + eps)
    groups = n if per_sample else 1
    aff = torch.stack([mean.reshape(groups, C), rstd.reshape(groups, C), (rstd * gamma.double()).reshape(groups, C),
                       (beta.double() - mean * rstd * gamma.double()).reshape(groups, C)]).float().contiguous()
    d = ops.norm_desc(y.shape, per_sample, C)
    z = torch.empty_like(y)
    ops.norm_apply_relu(d, y, aff, z)
    m32, r32, sc32, sh32 = [t.double().reshape((n, 1, 1, C) if per_sample else (C,)) for t in aff]
    u = y.double() * sc32 + sh32
    _stored_ok(z, torch.relu(u), flips=5e-3)
    dy, dgamma, dbeta, _, _ = ops.norm_relu_bwd(d, y, dz, aff, True, True)
    assert dy.dtype == torch.bfloat16
    du = dz.double() * (u > 0)
    xhat = (y.double() - m32) * r32
    k1 = du.mean(dim=axes, keepdim=per_sample)
    k2 = (du * xhat).mean(dim=axes, keepdim=per_sample)
    dy_ref = sc32 * (du - k1 - xhat * k2)
    _stored_ok(dy, dy_ref, flips=5e-3)
    assert ((dbeta.double() - du.sum((0, 1, 2))).abs().max() / du.sum((0, 1, 2)).abs().max()).item() < 2e-5
    assert ((dgamma.double() - (du * xhat).sum((0, 1, 2))).abs().max() / (du * xhat).sum((0, 1, 2)).abs().max()).item() < 2e-5


def test_bf16s_maxpool_forward_backward_with_skip_gradient():
    from boxsegliver_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    cat = torch.randn((2, 16, 24, 128), device="cuda", generator=g).bfloat16()
    x = ops.alias(cat, 0, (2, 16, 24, 64), cat.stride())            # a channel slice of a concat buffer
    p = ops.maxpool2_fwd(x)
    ref = torch.nn.functional.max_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert p.dtype == torch.bfloat16 and torch.equal(p.float(), ref)
    dp = torch.randn(p.shape, device="cuda", generator=g).bfloat16()
    dcat = torch.randn(cat.shape, device="cuda", generator=g).bfloat16()
    dskip = dcat[..., :64]
    dx = ops.maxpool2_bwd(x, p, dp, dskip)
    xx = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pp, idx = torch.nn.functional.max_pool2d(xx, 2, return_indices=True)
    route = torch.nn.functional.max_unpool2d(dp.float().permute(0, 3, 1, 2), idx, 2).permute(0, 2, 3, 1)
    want = (route.double() + dskip.double())
    _stored_ok(dx, want, flips=1e-3)
    assert torch.equal(ops.maxpool2_bwd(x, p, dp).float(), route)   # without the skip: pure routing, exact


@pytest.mark.parametrize("shape", [(2, 4, 8, 128, 64), (1, 2, 2, 1024, 512), (2, 8, 8, 256, 128), (1, 5, 3, 128, 64), (4, 64, 64, 128, 64)])
def test_bf16s_deconv_concat_forward_backward(shape):
    """k = s transposed conv under UNETK_BF16S: bf16 x, bf16 concat buffer, bf16 dcat / dx; dw, db fp32."""
    from boxsegliver_amd import _abi, ops
    n, h, w, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(cin + h)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16()
    wt = torch.randn((2, 2, cout, cin), device="cuda", generator=g) / math.sqrt(cin)
    b = 0.1 * torch.randn(cout, device="cuda", generator=g)
    skip = torch.randn((n, 2 * h, 2 * w, cout), device="cuda", generator=g).bfloat16()
    x64 = x.double().requires_grad_(True)
    w64 = _r(wt).requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    pre = tf_ops.conv_transpose_ks(x64, w64, (2, 2), bias=b64)
    up = torch.relu(pre)
    cat = torch.zeros((n, 2 * h, 2 * w, 2 * cout), device="cuda", dtype=torch.bfloat16)
    cat[..., :cout] = skip
    wp_f, wp_d = ops.deconv2x2_pack(wt, bf16=_abi.BF16S)
    ops.deconv2x2_fwd(x, wp_f, b, cat, cout, cout, bf16=_abi.BF16S)
    assert torch.equal(cat[..., :cout], skip)                         # the skip half is untouched
    _stored_ok(cat[..., cout:], up.detach())
    dcat = torch.randn(cat.shape, device="cuda", generator=g).bfloat16()
    # backward on the STORED forward value's mask (what the kernel sees)
    mask = (cat[..., cout:].double() > 0)
    dpre = dcat[..., cout:].double() * mask
    pre.backward(dpre)
    dx, dw, db = ops.deconv2x2_bwd(x, wp_d, cat, dcat, cout, cout, bf16=_abi.BF16S)
    assert dx.dtype == torch.bfloat16 and dw.dtype == torch.float32
    _stored_ok(dx, x64.grad)
    assert ((dw.double() - w64.grad).abs().max() / w64.grad.abs().max()).item() < 1e-5
    assert ((db.double() - b64.grad).abs().max() / b64.grad.abs().max()).item() < 1e-5
    dx2, dw2, db2 = ops.deconv2x2_bwd(x, wp_d, cat, dcat, cout, cout, bf16=_abi.BF16S)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)


def test_bf16s_head_reads_bf16_features_and_writes_bf16_gradient():
    from boxsegliver_amd import ops
    from oracle import losses
    n, h, w, c, ncls = 2, 16, 24, 64, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    z = torch.rand((n, h, w, c), device="cuda", generator=g).bfloat16()
    wt = torch.randn((1, 1, c, ncls), device="cuda", generator=g) / 8
    b = 0.1 * torch.randn(ncls, device="cuda", generator=g)
    labels = torch.randint(0, ncls, (n, h, w), device="cuda", generator=g, dtype=torch.int32)
    desc = ops.head_desc(n, h * w, c, ncls, "numerical", [0.2, 0.4, 4.4])
    z_ = z.clone().requires_grad_(True)
    w_ = wt.clone().requires_grad_(True)
    b_ = b.clone().requires_grad_(True)
    xent, dice, logits, probs, result = ops.HeadLoss.apply(z_, w_, b_, labels, None, desc, True)
    xent.backward()
    z64 = z.double().cpu().requires_grad_(True)
    w64 = wt.double().cpu().requires_grad_(True)
    b64 = b.double().cpu().requires_grad_(True)
    lg = (z64.reshape(-1, c) @ w64.reshape(c, ncls) + b64).reshape(n, h, w, ncls)
    ref = losses.weighted_sparse_softmax_cross_entropy(lg, labels.long().cpu(), "numerical", numeric_w=[0.2, 0.4, 4.4])
    ref.backward()
    assert abs(xent.item() - ref.item()) < 1e-5
    assert (logits.reshape(n, h, w, ncls).double().cpu() - lg.detach()).abs().max().item() < 1e-5
    assert z_.grad.dtype == torch.bfloat16
    _stored_ok(z_.grad.cpu(), z64.grad, flips=5e-3)
    assert ((w_.grad.double().cpu() - w64.grad).abs().max() / w64.grad.abs().max()).item() < 1e-5
    assert ((b_.grad.double().cpu() - b64.grad).abs().max() / b64.grad.abs().max()).item() < 1e-5


# ----------------------------------------------------------------------------------------- whole network
def _grad_l2(model, grads):
    num = den = 0.0
    for name in model.params.trainable_names():
        gt = model.params[name].grad.double()
        r = grads[name].to(gt.device)
        num += float(((gt - r) ** 2).sum())
        den += float((r ** 2).sum())
    return (num / den) ** 0.5


def _run_pair(size, bs, **over):
    import test_gpu_unet as t
    args = t.make_args(batch_size=bs, im_height=size, im_width=size, compute_dtype="bf16", **over)
    images, labels = t.synth(bs, size, size, 3)
    model, inputs = t.build(args, images, labels)
    from oracle import unet2d
    net = unet2d.UNet2DOracle(3, 3, normalizer=args.normalizer, without_norm=args.without_norm)
    params = unet2d.init_params(net.specs, seed=77)
    gen = torch.Generator().manual_seed(5)
    for name, _, kind in net.specs:
        if kind == "gamma":
            params[name] = 0.5 + torch.rand(params[name].shape, generator=gen)
        elif kind == "beta":
            params[name] = 0.2 * torch.randn(params[name].shape, generator=gen)
        elif kind == "bias":
            params[name] = 0.1 * torch.randn(params[name].shape, generator=gen)
    model.params.load_state(params)
    net.bf16 = 2
    p64 = {k: v.double().cuda() for k, v in params.items()}
    total, _, logits, grads, stats = net.loss_and_grads(p64, inputs["images"].double(), inputs["labels"].long(),
                                                        **t.loss_kwargs(args))
    model.params.zero_grad()
    loss = model(inputs, "train", **t.YML)
    loss.backward()
    torch.cuda.synchronize()
    return t, args, model, inputs, net, loss, total, logits, grads, stats


@pytest.mark.parametrize("over", [dict(), dict(normalizer="instance_norm", loss_type="dice"), dict(without_norm=True)])
def test_unet_bf16s_step_against_the_same_arithmetic_oracle(over):
    """The whole UNet step in bf16-storage mode (reduced size) against the oracle restating the same roundings, in
    float64 on the device.  The kernels are pinned one by one on identical operands (tests above and below: stored values
    within half a bf16 ulp of the exact result); what this test measures is how far two correct executions of the WHOLE
    net drift apart when fp32-vs-fp64 accumulation moves ~0.1 % of the stored values per layer across a bf16 rounding
    boundary (a whole ulp = 0.4-0.8 % of the value).  Measured on MI355X: without normalisation (well conditioned) loss
    1e-5, logits 0.3 % of their range, whole-gradient L2 2.6e-3; with batch / instance norm at batch 2 (4 x 4 bridge:
    statistics over 32 / 16 values amplify every flip) logits 0.4 % of the range, gradient L2 0.18-0.20."""
    t, args, model, inputs, net, loss, total, logits, grads, stats = _run_pair(64, 2, **over)
    assert all(v.dtype == torch.bfloat16 for k, v in model.layers.items() if k.startswith("Encode"))
    got = model.layers["logits"].double()
    d = (got - logits).abs()
    rng_ = (logits.max() - logits.min()).item()
    print("bf16s small", over, "loss", abs(loss.item() - total.item()), "logits max", d.max().item(), "mean", d.mean().item(),
          "range", rng_, "argmax", (got.argmax(-1) == logits.argmax(-1)).double().mean().item(), "gradL2", _grad_l2(model, grads))
    assert abs(loss.item() - total.item()) < 5e-4 * max(1.0, abs(total.item()))
    assert d.max().item() < 1.5e-2 * rng_ and d.mean().item() < 2e-3 * rng_
    assert (got.argmax(-1) == logits.argmax(-1)).double().mean().item() > 0.99
    assert _grad_l2(model, grads) < (1e-2 if over.get("without_norm") else 0.5)
    if not over.get("without_norm") and args.normalizer == "batch_norm":     # moving statistics from the fp32 accumulators
        sd = model.params.state_dict()
        for k, v in stats.items():
            assert (sd[k].double() - v.cpu()).abs().max().item() < 1e-4, k


def test_unet_bf16s_every_backward_kernel_on_identical_operands():
    """Inside a real bf16-storage step every backward kernel is re-checked in float64 on the operands it actually saw
    (captured): the filter gradient (fp32) to 1e-5, the stored input gradient to half a bf16 ulp."""
    import torch.nn.functional as F
    from boxsegliver_amd import ops
    import test_gpu_unet as t
    args = t.make_args(batch_size=2, im_height=64, im_width=64, compute_dtype="bf16")
    images, labels = t.synth(2, 64, 64, 3)
    model, inputs = t.build(args, images, labels)
    ops.DEBUG_CAPTURE = []
    try:
        model.params.zero_grad()
        model(inputs, "train", **t.YML).backward()
        torch.cuda.synchronize()
        captured = ops.DEBUG_CAPTURE
    finally:
        ops.DEBUG_CAPTURE = None
    units = [c for c in captured if c.get("kind") != "deconv"]
    assert len(units) == 18 and sum(1 for c in units if c["x"].dtype == torch.bfloat16) == 17
    for c in units:
        x64 = c["x"].detach().double().permute(0, 3, 1, 2).requires_grad_(True)
        w = c["w"].detach()
        w64 = (_r(w) if c["x"].dtype == torch.bfloat16 else w.double()).permute(3, 2, 0, 1).requires_grad_(True)
        F.conv2d(x64, w64, padding=1).backward(c["dy"].detach().double().permute(0, 3, 1, 2))
        ref = w64.grad.permute(2, 3, 1, 0)
        assert ((c["dw"].double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
        if c["dx"] is not None:
            _stored_ok(c["dx"], x64.grad.permute(0, 2, 3, 1))
        assert c["dy"].dtype == torch.bfloat16 and c["dz"].dtype == torch.bfloat16


def test_unet_bf16s_step_is_bit_reproducible_and_trains():
    from boxsegliver_amd.core.solver import Solver
    import test_gpu_unet as t
    args = t.make_args(batch_size=4, im_height=128, im_width=128, compute_dtype="bf16")
    images, labels = t.synth(4, 128, 128, 3)
    model, inputs = t.build(args, images, labels)
    runs = []
    for _ in range(2):
        model.params.zero_grad()
        loss = model(inputs, "train", **t.YML)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.item(), model.params.grad["reg"].clone(), model.params.grad["noreg"].clone()))
    assert np.isfinite(runs[0][0]) and runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    solver = Solver(args)
    curve = []
    for _ in range(6):
        loss = model(inputs, "train", **t.YML)
        curve.append(loss.item())
        solver(loss, model)
    assert curve[-1] < curve[0]
    # eval mode (moving statistics) runs and yields probabilities
    model(inputs, "eval", **t.YML)
    assert torch.isfinite(model.probability).all() and model.probability.dtype == torch.float32
