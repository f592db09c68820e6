"""GPU: the inference epilogue -- conv3x3 + the normaliser's (scale, shift) + ReLU [+ the 2 x 2 max-pool the unit feeds] as ONE
kernel (unetk_conv3x3_fwd_affine), used when mode == EVAL and the affine is known before the conv
runs: slim.batch_norm with is_training False (NetworksV2/base.py:71-79,153-162: moving statistics) and --without_norm
(UNet.py:47-48).  Checked bitwise against the two-pass path (conv -> norm_apply_relu [-> max_pool]) -- same fmaf / fmaxf
expression on the same accumulators -- against float64, and through the whole net against the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import tf_ops, unet2d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from boxsegliver_amd import ops as o
    return o


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).float().cuda()


SHAPES = [
    # N, H, W, Cin, Cout, pool      -> tile configuration (conv_igemm.hip)
    (2, 16, 32, 16, 64, True),      # <4,1,1,2>  128 x 64
    (12, 32, 32, 64, 128, True),    # <2,2,1,2>  half-height tiles (small grid)
    (2, 24, 40, 32, 128, True),     # ragged last tile column, 24 rows = three 8-row tiles
    (1, 8, 16, 64, 128, False),     # small planes -> the linear-pixel kernel (affine epilogue, pool done separately)
    (2, 24, 20, 32, 128, False),
    (8, 16, 16, 128, 256, False),   # ... with the stream-K schedule: split tiles get the epilogue in the fix-up kernel
    (5, 12, 12, 64, 128, False),
    (1, 16, 16, 128, 64, True),
    (3, 8, 48, 64, 64, True),
    (2, 16, 32, 32, 32, True),      # <4,1,2,1>  256 x 32
    (16, 64, 64, 64, 128, True),    # <2,2,2,2>  128 x 128
    (32, 64, 64, 64, 128, True),    # <2,2,4,2>  16 x 16 pixel tiles (>= 512 blocks)
    (32, 64, 64, 64, 64, True),     # <4,1,2,2>
    (2, 9, 33, 32, 64, False),      # odd extents: no pool
    (2, 24, 40, 3, 64, False),      # first layer on the matrix pipe (Cin = 3)
    (1, 16, 16, 4, 64, False),      # guided nets' 4-channel input
]


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_conv_affine_relu_pool_equals_the_two_pass_path_bitwise(ops, shape):
    n, h, w, cin, cout, pool = shape
    assert ops.conv3x3_fwd_affine_ok(n, h, w, cin, cout, pool)
    rng = np.random.default_rng(abs(hash(shape)) % 2**31)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) / math.sqrt(9 * cin)).astype(np.float32)
    sc = (0.5 + rng.random(cout)).astype(np.float32) * np.where(rng.random(cout) < 0.2, -1.0, 1.0).astype(np.float32)
    sh = (0.3 * rng.standard_normal(cout)).astype(np.float32)
    xd, wd_ = dev(x), dev(wt)
    wp = ops.conv3x3_pack(wd_)[0] if ops.conv_uses_mfma(cin, cout) else wd_
    # two passes: raw conv output, then the norm-apply kernel (+ pool)
    y, _, _ = ops.conv3x3_fwd(xd, wp, cout, want_stats=False)
    aff = torch.stack([torch.zeros(cout), torch.ones(cout), torch.tensor(sc), torch.tensor(sh)]).reshape(4, 1, cout).cuda().contiguous()
    nd = ops.norm_desc(y.shape, False, cout, 0, 0, 0)
    z2 = ops.norm_apply_relu(nd, y, aff, torch.empty_like(y))
    # one pass, into a channel slice of a wider buffer (the decoder's concat buffer)
    buf = torch.full((n, h, w, cout + 32), -7.0, device="cuda")
    zv = buf[..., :cout]
    z1, p1 = ops.conv3x3_fwd_affine(xd, wp, cout, aff[2], aff[3], z=zv, pool=pool)
    torch.cuda.synchronize()
    assert torch.equal(zv, z2)
    assert float((buf[..., cout:] + 7.0).abs().max()) == 0.0                 # nothing outside the slice was touched
    if pool:
        assert torch.equal(p1, ops.maxpool2_fwd(z2))
    ref = torch.relu(tf_ops.conv_nd_same(torch.tensor(x, dtype=torch.float64), torch.tensor(wt, dtype=torch.float64)) *
                     torch.tensor(sc, dtype=torch.float64) + torch.tensor(sh, dtype=torch.float64))
    assert (zv.double().cpu() - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())


def test_shapes_without_a_fused_kernel_are_refused_not_approximated(ops):
    assert not ops.conv3x3_fwd_affine_ok(8, 16, 16, 512, 1024, pool=True)    # small planes: the linear-pixel kernel cannot pool
    assert not ops.conv3x3_fwd_affine_ok(2, 9, 33, 32, 64, pool=True)        # odd extents cannot pool
    assert not ops.conv3x3_fwd_affine_ok(2, 32, 32, 9, 64)                   # --img_grad first layer: generic direct kernel
    x, w = torch.zeros(8, 16, 16, 512, device="cuda"), torch.zeros(9 * 512 * 1024, device="cuda")
    with pytest.raises(ops._abi.UnetkError):
        ops.conv3x3_fwd_affine(x, w, 1024, torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda"), pool=True)


@pytest.mark.parametrize("variant", ["batch_norm", "without_norm"])
def test_unet_eval_takes_the_fused_path_and_matches_two_pass_and_oracle(variant):
    import test_gpu_unet as t
    from boxsegliver_amd import ops
    over = dict(without_norm=True) if variant == "without_norm" else {}
    args = t.make_args(**over)
    images, labels = t.synth(2, 32, 32, 3)
    model, inputs = t.build(args, images, labels)
    net = unet2d.UNet2DOracle(3, 3, normalizer=args.normalizer, without_norm=args.without_norm)
    params = unet2d.init_params(net.specs, seed=21)
    g = torch.Generator().manual_seed(6)
    for name, _, kind in net.specs:
        if kind == "gamma":
            params[name] = 0.5 + torch.rand(params[name].shape, generator=g)
        elif kind in ("beta", "bias", "moving_mean"):
            params[name] = 0.1 * torch.randn(params[name].shape, generator=g)
        elif kind == "moving_var":
            params[name] = 0.5 + torch.rand(params[name].shape, generator=g)
    model.params.load_state(params)
    lg_ref, _ = net.forward(params, torch.from_numpy(images), False)           # is_training False: moving statistics
    calls = []
    real = ops._abi.lib().unetk_conv3x3_fwd_affine
    runs = {}
    for fuse in (True, False):
        ops.FUSE_EVAL = fuse
        try:
            model(inputs, "eval", **t.YML)
            torch.cuda.synchronize()
            runs[fuse] = (model.layers["logits"].clone(), model.probability.clone(), model.predictions["LiverPred"].clone(),
                          model.predictions["TumorPred"].clone())
        finally:
            ops.FUSE_EVAL = True
    # the fused and the two-pass evaluation are the same arithmetic
    for a, b in zip(runs[True], runs[False]):
        assert torch.equal(a, b)
    logits, prob, liver, tumor = runs[True]
    assert (logits.cpu() - lg_ref).abs().max().item() < 1e-3
    # Pred = probability > 0.5 per class (UNet.py:108-113), argmax rule of the evaluator: exact on the device's own numbers
    assert torch.equal(liver[..., 0], (prob[..., 1] > 0.5).to(torch.uint8))
    assert torch.equal(tumor[..., 0], (prob[..., 2] > 0.5).to(torch.uint8))
    srt = torch.sort(lg_ref, -1).values
    safe = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert bool((logits.cpu().argmax(-1) == lg_ref.argmax(-1))[safe].all()) and safe.double().mean().item() > 0.99
    # and the fused kernels really ran: all 18 conv units (at 32 x 32 bs 2 the 16 x 16 and 8 x 8 levels take the small-plane
    # kernel, whose epilogue has the affine + ReLU but not the pool: two of the four pools ride on their conv)
    ops.profile_begin(0)
    rec = []
    ops.profile_on(rec)
    try:
        model(inputs, "eval", **t.YML)
    finally:
        ops.profile_on(None)
    torch.cuda.synchronize()
    fused = [r for r in rec if "+affine+relu" in r[0]]
    assert len(fused) == 18 and sum(1 for r in fused if r[0].endswith("+pool")) >= 2
    assert not [r for r in rec if r[0] in ("norm_apply_relu_pool", "norm_apply_relu")]


# ------------------------------------------------------------------------------------------------ bf16 storage mode
BF16_SHAPES = [
    # N, H, W, Cin, Cout, pool    (the persistent bf16-storage kernel: >= 200 tiles of 32 x 16 pixels)
    (8, 128, 128, 64, 128, True),     # 128-channel tiles
    (8, 128, 128, 128, 64, True),     # 64-channel tiles
    (8, 96, 80, 64, 256, True),       # two channel tiles, several chunks
    (20, 50, 70, 64, 128, False),     # ragged rows and columns: no pool
]


@pytest.mark.parametrize("shape", BF16_SHAPES)
def test_bf16_storage_fused_affine_relu_pool_against_float64_of_the_same_operands(ops, shape):
    """UNETK_BF16S: the activation is rounded to bf16 ONCE, from the fp32 accumulator after the affine and the ReLU (the
    two-pass path rounds the raw output first): checked against float64 on the rounded operands to half a bf16 ulp of the
    result, the pooled tensor bitwise against the maximum of the stored activations."""
    n, h, w, cin, cout, pool = shape
    B = ops._abi.BF16S
    assert ops.conv3x3_fwd_affine_ok(n, h, w, cin, cout, pool, bf16=B)
    rng = np.random.default_rng(abs(hash(shape)) % 2**31)
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).cuda().to(torch.bfloat16)
    wt = torch.from_numpy((rng.standard_normal((3, 3, cin, cout)) / math.sqrt(9 * cin)).astype(np.float32)).cuda()
    sc = torch.from_numpy(((0.5 + rng.random(cout)) * np.where(rng.random(cout) < 0.2, -1.0, 1.0)).astype(np.float32)).cuda()
    sh = torch.from_numpy((0.3 * rng.standard_normal(cout)).astype(np.float32)).cuda()
    wp = ops.conv3x3_pack(wt, want_dgrad=False, bf16=B)[0]
    buf = torch.full((n, h, w, cout + 64), -7.0, device="cuda", dtype=torch.bfloat16)
    zv = buf[..., :cout]
    z, p = ops.conv3x3_fwd_affine(x, wp, cout, sc, sh, z=zv, pool=pool, bf16=B)
    torch.cuda.synchronize()
    assert float((buf[..., cout:].float() + 7.0).abs().max()) == 0.0
    xr = x.double()
    wr = wt.to(torch.bfloat16).double()
    ref = torch.relu(torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1) * sc.double()
                     + sh.double())
    got = zv.double()
    ulp = torch.maximum(ref.abs(), torch.tensor(1e-30, device="cuda", dtype=torch.float64)) * 2.0 ** -8     # half a bf16 ulp <= |v| 2^-8
    assert bool(((got - ref).abs() <= ulp * 1.02 + 2e-5).all())
    if pool:
        assert torch.equal(p, torch.nn.functional.max_pool2d(zv.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).to(torch.bfloat16))


def test_unet_bf16_storage_eval_takes_the_fused_path_and_stays_within_the_mode_s_bars():
    """--compute_dtype bf16 evaluation at 256 x 256 bs 8 (the two finest levels are large enough for the persistent kernel): fused and two-pass forwards agree to bf16 rounding noise (the fused
    path rounds once per unit instead of twice), both match the oracle restating the bf16 arithmetic within the bars of
    tests/test_gpu_bf16s.py, and the persistent kernel's fused instantiations ran."""
    import test_gpu_unet as t
    from boxsegliver_amd import ops
    args = t.make_args(batch_size=8, im_height=256, im_width=256, compute_dtype="bf16")
    images, labels = t.synth(8, 256, 256, 3)
    model, inputs = t.build(args, images, labels)
    net, params = t.oracle_for(args)
    g = torch.Generator().manual_seed(3)
    for name, _, kind in net.specs:
        if kind == "moving_mean":
            params[name] = 0.1 * torch.randn(params[name].shape, generator=g)
        elif kind == "moving_var":
            params[name] = 0.5 + torch.rand(params[name].shape, generator=g)
    model.params.load_state(params)
    p64 = {k: v.double().cuda() for k, v in params.items()}
    net.bf16 = 2
    lg_ref, _ = net.forward(p64, inputs["images"].double(), False)
    runs = {}
    for fuse in (True, False):
        ops.FUSE_EVAL = fuse
        try:
            model(inputs, "eval", **t.YML)
            torch.cuda.synchronize()
            runs[fuse] = model.layers["logits"].double().clone()
        finally:
            ops.FUSE_EVAL = True
    for r in runs.values():
        d = (r - lg_ref).abs()
        assert d.mean().item() < 1e-2 and d.max().item() < 0.1
        assert (r.argmax(-1) == lg_ref.argmax(-1)).double().mean().item() > 0.99
    assert (runs[True] - runs[False]).abs().mean().item() < 1e-2
    ops.profile_begin(0)
    rec = []
    ops.profile_on(rec)
    try:
        model(inputs, "eval", **t.YML)
    finally:
        ops.profile_on(None)
    torch.cuda.synchronize()
    fused = [r for r in rec if "+affine+relu" in r[0]]
    assert len(fused) >= 8 and any("conv3x3_bf16s_kernel" in r[0] and r[0].endswith("+pool") for r in fused)
