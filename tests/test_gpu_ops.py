"""GPU parity tests, op by op: HIP kernels (through the libunetk C ABI) vs the CPU oracle on the
same seeded inputs.  Tolerances are fp32 accumulation-order bounds, far inside the north-star's
1e-3; integer outputs (argmax / thresholded masks) are compared bit-exactly."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as olosses
from oracle import solver as osolver
from oracle import tf_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from boxsegliver_amd import ops as _ops
    from boxsegliver_amd import _abi
    _abi.lib()
    return _ops


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def rel_err(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


CONV_SHAPES = [
    # N, H, W, Cin, Cout
    (2, 16, 32, 16, 64),       # 128x64 tile path
    (1, 8, 16, 64, 128),       # 128x128 tile path
    (2, 24, 20, 32, 128),      # H, W not multiples of the 8x16 tile (masked edges)
    (1, 16, 16, 128, 64),
    (3, 8, 48, 64, 64),
    (1, 4, 4, 32, 256),        # image smaller than a tile
    (2, 16, 32, 32, 32),       # 256x32 tile path (UNet3D's 30-channel levels padded to 32)
    (1, 24, 20, 64, 32),
    # small planes -> linear-pixel kernel (conv_igemm_lin.hip): blocks span rows and planes
    (5, 12, 12, 64, 128),      # UNet3D e3 shape; 720 pixels = 5.6 blocks, plane boundaries inside blocks
    (9, 6, 6, 32, 64),         # bridge shape: a block covers 3.6 planes
    (2, 24, 24, 48, 128),      # 24 wide: 75 % fill in the tiled kernel
    (3, 11, 13, 16, 64),       # odd sizes, last block partial
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_fwd_and_stats(ops, shape):
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(hash(shape) % 2**31)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) / math.sqrt(9 * cin)).astype(np.float32)
    ref = tf_ops.conv_nd_same(torch.tensor(x, dtype=torch.float64), torch.tensor(wt, dtype=torch.float64)).numpy()
    wp_f, _ = ops.conv3x3_pack(dev(wt))
    y, stats, rows = ops.conv3x3_fwd(dev(x), wp_f, cout, want_stats=True)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), ref) < 2e-6
    s = stats.cpu().numpy().astype(np.float64)
    assert s.shape == (2, rows, cout)
    np.testing.assert_allclose(s[0].sum(0), ref.sum((0, 1, 2)), atol=2e-4 * max(1, abs(ref).sum((0, 1, 2)).max()))
    np.testing.assert_allclose(s[1].sum(0), (ref ** 2).sum((0, 1, 2)), rtol=2e-5)


def test_conv3x3_fwd_strided_input_view(ops):
    # decoder conv1 reads the whole concat buffer; a channel-slice view exercises x_stride > Cin
    rng = np.random.default_rng(5)
    buf = rng.standard_normal((2, 8, 16, 96)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, 32, 64)) / 17).astype(np.float32)
    xb = dev(buf)
    xv = xb[..., 32:64]
    ref = tf_ops.conv_nd_same(torch.tensor(buf[..., 32:64], dtype=torch.float64), torch.tensor(wt, dtype=torch.float64)).numpy()
    wp_f, _ = ops.conv3x3_pack(dev(wt))
    y, _, _ = ops.conv3x3_fwd(xv, wp_f, 64, want_stats=False)
    assert rel_err(y.cpu().numpy(), ref) < 2e-6


def test_conv3x3_direct_first_layer(ops):
    # Encode1/conv1: Cin = 3 (K = 27) -> direct kernel on raw HWIO filters
    rng = np.random.default_rng(7)
    x = rng.random((2, 24, 40, 3)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, 3, 64)) / 5).astype(np.float32)
    ref = tf_ops.conv_nd_same(torch.tensor(x, dtype=torch.float64), torch.tensor(wt, dtype=torch.float64)).numpy()
    y, stats, rows = ops.conv3x3_fwd(dev(x), dev(wt), 64, want_stats=True)
    assert rel_err(y.cpu().numpy(), ref) < 2e-6
    s = stats.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(s[0].sum(0), ref.sum((0, 1, 2)), atol=1e-3)
    np.testing.assert_allclose(s[1].sum(0), (ref ** 2).sum((0, 1, 2)), rtol=2e-5)


@pytest.mark.parametrize("shape", [(2, 16, 32, 64, 64), (1, 8, 16, 128, 64), (2, 12, 20, 64, 128), (1, 16, 16, 64, 16 * 8),
                                   (2, 16, 32, 32, 32), (1, 8, 16, 32, 64), (2, 12, 20, 64, 32), (3, 9, 17, 96, 32),
                                   (5, 12, 12, 128, 64), (9, 6, 6, 64, 64), (2, 24, 24, 128, 64)])   # linear-pixel dgrad
def test_conv3x3_dgrad_wgrad(ops, shape):
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(11 + cin + cout)
    x = torch.tensor(rng.standard_normal((n, h, w, cin)), dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(rng.standard_normal((3, 3, cin, cout)) / math.sqrt(9 * cin), dtype=torch.float64, requires_grad=True)
    dy = rng.standard_normal((n, h, w, cout))
    y = tf_ops.conv_nd_same(x, wt)
    y.backward(torch.tensor(dy))
    _, wp_d = ops.conv3x3_pack(dev(wt.detach().numpy()))
    dx = ops.conv3x3_dgrad(dev(dy), wp_d, cin)
    dw = ops.conv3x3_wgrad(dev(x.detach().numpy()), dev(dy))
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 3e-6
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 3e-6
    # bit-reproducible (fixed-order split-K): run again, compare exactly
    dw2 = ops.conv3x3_wgrad(dev(x.detach().numpy()), dev(dy))
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("shape", [(8, 128, 128, 64, 128), (4, 64, 64, 256, 256), (16, 256, 256, 64, 64),
                                   (8, 128, 120, 64, 128)])      # 16 x 16 pixel tiles with a ragged last tile column
def test_conv3x3_large_shapes_against_on_device_float64(ops, shape):
    """BASELINE-sized layers (thousands of tiles, many tiles per split-K block): forward, input gradient and filter
    gradient against PyTorch's float64 convolution evaluated ON THE DEVICE (a checker the CPU oracle is too slow for;
    SURVEY.md 8c), plus run-to-run bit equality of the filter gradient."""
    n, h, w, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(n + cin)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g)
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    dy = torch.randn((n, h, w, cout), device="cuda", generator=g)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = wt.double().permute(3, 2, 0, 1).requires_grad_(True)
    y64 = F.conv2d(x64, w64, padding=1)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    wp_f, wp_d = ops.conv3x3_pack(wt)
    y, _, _ = ops.conv3x3_fwd(x, wp_f, cout, want_stats=False)
    dx = ops.conv3x3_dgrad(dy, wp_d, cin)
    dw = ops.conv3x3_wgrad(x, dy)

    def rel(a, b):
        return ((a.double() - b).abs().max() / b.abs().max()).item()

    assert rel(y, y64.detach().permute(0, 2, 3, 1)) < 3e-6
    assert rel(dx, x64.grad.permute(0, 2, 3, 1)) < 3e-6
    assert rel(dw, w64.grad.permute(2, 3, 1, 0)) < 1e-5
    assert torch.equal(dw, ops.conv3x3_wgrad(x, dy))


@pytest.mark.parametrize("n,off", [(1, 0), (3, 0), (4099, 0), (1 << 20, 0), (1000003, 1), (777, 3)])
def test_sumsq_sizes_and_alignment(ops, n, off):
    # float4 body + scalar tail; a view that starts off a 16-byte boundary takes the scalar path
    g = torch.Generator(device="cuda").manual_seed(n)
    buf = torch.randn(n + 8, device="cuda", generator=g)
    v = buf[off:off + n]
    ref = float((v.double() ** 2).sum())
    got = float(ops.sumsq(v).item())
    assert abs(got - ref) <= 1e-6 * max(ref, 1.0), (got, ref)
    assert got == float(ops.sumsq(v).item())                 # fixed summation order


def test_conv3x3_wgrad_first_layer(ops):
    rng = np.random.default_rng(13)
    x = torch.tensor(rng.random((2, 16, 32, 3)), dtype=torch.float64)
    wt = torch.tensor(rng.standard_normal((3, 3, 3, 64)), dtype=torch.float64, requires_grad=True)
    dy = rng.standard_normal((2, 16, 32, 64))
    tf_ops.conv_nd_same(x, wt).backward(torch.tensor(dy))
    dw = ops.conv3x3_wgrad(dev(x.numpy()), dev(dy))
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 3e-6


def _stats_like_conv_epilogue(yd, per_image_rows=True):
    """Statistic partials as the conv epilogue produces them: rows of 128 pixels, each image's rows contiguous."""
    n, c = yd.shape[0], yd.shape[-1]
    flat = yd.reshape(n, -1, c)
    rpi = (flat.shape[1] + 127) // 128
    pad = rpi * 128 - flat.shape[1]
    fp = torch.cat([flat, torch.zeros(n, pad, c, device="cuda")], 1).reshape(n * rpi, 128, c)
    return torch.stack([fp.sum(1), (fp * fp).sum(1)]).contiguous(), n * rpi


NORM_CASES = [
    # kind, C, (n,h,w), has_gamma, has_beta, guide channels
    ("batch_norm", 64, (2, 16, 16), True, True, 0),
    ("batch_norm", 128, (1, 8, 24), True, True, 0),
    ("batch_norm", 1024, (2, 2, 2), True, True, 0),
    ("batch_norm", 256, (3, 5, 7), True, True, 0),
    ("instance_norm", 64, (3, 16, 16), True, True, 0),
    ("instance_norm", 256, (2, 6, 10), True, True, 0),
    ("instance_norm", 128, (2, 16, 8), False, True, 1),     # GUNet.yml: centre only + 1-channel spatial guide
    ("batch_norm", 128, (2, 16, 8), False, True, 2),        # BN encoder with a 2-channel guide
    ("instance_norm", 64, (2, 8, 8), False, False, 0),
    # many pixels: the backward's partial rows reach the wide final reduction directly (576 and 1024 rows) and, per
    # sample, through a second level over the launch groups
    ("batch_norm", 64, (1, 96, 96), True, True, 0),
    ("batch_norm", 64, (4, 128, 128), True, True, 0),
    ("instance_norm", 32, (3, 64, 96), True, True, 0),
    ("batch_norm", 128, (2, 64, 64), False, True, 1),
]


@pytest.mark.parametrize("kind,c,shape,has_gamma,has_beta,g_ch", NORM_CASES)
def test_norm_relu_forward_backward(ops, kind, c, shape, has_gamma, has_beta, g_ch):
    rng = np.random.default_rng(c + g_ch)
    n, h, w = shape
    per_sample = kind == "instance_norm"
    eps = 1e-6 if per_sample else 1e-3
    y_np = (rng.standard_normal((n, h, w, c)) * 2 + 0.5).astype(np.float32)
    gamma = (rng.random(c) + 0.5).astype(np.float32) if has_gamma else None
    beta = (rng.standard_normal(c) * 0.3).astype(np.float32) if has_beta else None
    mm0 = rng.standard_normal(c).astype(np.float32)
    mv0 = (rng.random(c) + 0.5).astype(np.float32)
    dz_np = rng.standard_normal((n, h, w, c)).astype(np.float32)
    guide = rng.random((n, h, w, g_ch)).astype(np.float32) if g_ch else None
    gw_full = (rng.standard_normal((g_ch, 2 * c)) * 0.5).astype(np.float32) if g_ch else None
    gb_full = (rng.standard_normal(2 * c) * 0.2).astype(np.float32) if g_ch else None
    # oracle (float64)
    t64 = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    y64, g64, b64, gw64, gb64 = t64(y_np), t64(gamma), t64(beta), t64(gw_full), t64(gb_full)
    if per_sample:
        zn = tf_ops.instance_norm(y64, g64, b64, eps=eps)
        nmm = nmv = None
    else:
        zn, nmm, nmv = tf_ops.batch_norm(y64, g64, b64, torch.tensor(mm0, dtype=torch.float64),
                                         torch.tensor(mv0, dtype=torch.float64), True, eps=eps)
    if g_ch:   # GUNet.py:154-156,207-212: sp = conv1x1(guide) (+bias); net + sp[..., C:2C] (second conv of the block)
        sp = torch.tensor(guide, dtype=torch.float64) @ gw64 + gb64
        zn = zn + sp[..., c:2 * c]
    z_ref = torch.relu(zn)
    z_ref.backward(torch.tensor(dz_np, dtype=torch.float64))
    # HIP
    yd = dev(y_np)
    stats, rows = _stats_like_conv_epilogue(yd)
    mm, mv = dev(mm0), dev(mv0)
    gd, bd = (dev(gamma) if has_gamma else None), (dev(beta) if has_beta else None)
    guide_d = dev(guide) if g_ch else None
    gw_d = dev(gw_full[:, c:]) if g_ch else None            # this conv's column slice, contiguous [g][C]
    gb_d = dev(gb_full[c:]) if g_ch else None
    d = ops.norm_desc(yd.shape, per_sample, c, g_ch, c if g_ch else 0, 0)
    aff = ops.norm_finalize(d, stats, rows, gd, bd, eps, 0.999, True, mm, mv, yd.device)
    z = torch.empty_like(yd)
    ops.norm_apply_relu(d, yd, aff, z, guide_d, gw_d, gb_d)
    assert rel_err(z.cpu().numpy(), z_ref.detach().numpy()) < 1e-5
    if not per_sample:
        np.testing.assert_allclose(mm.cpu().numpy(), nmm.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mv.cpu().numpy(), nmv.numpy(), rtol=1e-5, atol=1e-6)
    else:
        assert torch.equal(mm.cpu(), torch.tensor(mm0))          # instance norm keeps no moving statistics
    dy, dgamma, dbeta, dgw, dgb = ops.norm_relu_bwd(d, yd, dev(dz_np), aff, has_gamma, has_beta, guide_d, gw_d, gb_d)
    assert rel_err(dy.cpu().numpy(), y64.grad.numpy()) < 2e-5
    if has_gamma:
        assert rel_err(dgamma.cpu().numpy(), g64.grad.numpy()) < 2e-5
    if has_beta:
        assert rel_err(dbeta.cpu().numpy(), b64.grad.numpy()) < 2e-5
    if g_ch:
        assert rel_err(dgw.cpu().numpy(), gw64.grad.numpy()[:, c:]) < 2e-5
        assert rel_err(dgb.cpu().numpy(), gb64.grad.numpy()[c:]) < 2e-5
        assert np.abs(gw64.grad.numpy()[:, :c]).max() == 0.0
    if not per_sample:
        # eval mode: moving statistics
        aff_e = ops.norm_finalize(d, None, 0, gd, bd, eps, 0.999, False, mm, mv, yd.device)
        ze = torch.empty_like(yd)
        ops.norm_apply_relu(d, yd, aff_e, ze, guide_d, gw_d, gb_d)
        ref_e, _, _ = tf_ops.batch_norm(torch.tensor(y_np, dtype=torch.float64), None if g64 is None else g64.detach(),
                                        None if b64 is None else b64.detach(), mm.cpu().double(), mv.cpu().double(),
                                        False, eps=eps)
        if g_ch:
            ref_e = ref_e + (torch.tensor(guide, dtype=torch.float64) @ gw64.detach() + gb64.detach())[..., c:2 * c]
        assert rel_err(ze.cpu().numpy(), torch.relu(ref_e).numpy()) < 1e-5


def test_norm_apply_into_concat_slice(ops):
    rng = np.random.default_rng(3)
    y = dev(rng.standard_normal((2, 4, 6, 64)))
    cat = torch.full((2, 4, 6, 128), -7.0, device="cuda")
    aff = torch.stack([torch.zeros(1, 64, device="cuda"), torch.ones(1, 64, device="cuda"),
                       dev(rng.random((1, 64)) + 0.5), dev(rng.standard_normal((1, 64)))]).contiguous()
    d = ops.norm_desc(y.shape, False, 128)
    ops.norm_apply_relu(d, y, aff, cat[..., :64])
    ref = torch.relu(y * aff[2, 0] + aff[3, 0])
    assert torch.allclose(cat[..., :64], ref, atol=1e-6)
    assert torch.all(cat[..., 64:] == -7.0)


def test_avgpool_guide_pyramid(ops):
    rng = np.random.default_rng(8)
    g = rng.random((2, 8, 12, 1)).astype(np.float32)
    p = ops.avgpool2_fwd(dev(g))
    ref = tf_ops.avg_pool2x2_same(torch.tensor(g))
    assert torch.allclose(p.cpu(), ref, atol=1e-7)


def test_maxpool_forward_backward_with_ties(ops):
    rng = np.random.default_rng(4)
    x_np = rng.standard_normal((2, 8, 12, 64)).astype(np.float32)
    x_np[0, :2, :2, :8] = 0.0                       # 4-way tie (post-ReLU zeros)
    x_np[1, 2, 4, :] = x_np[1, 3, 5, :] = 9.0       # 2-way tie, first in scan order wins
    cat = torch.zeros((2, 8, 12, 128), device="cuda")
    cat[..., :64] = dev(x_np)
    xv = cat[..., :64]                               # strided view, as the encoder skip is
    p = ops.maxpool2_fwd(xv)
    ref = tf_ops.max_pool2x2(torch.tensor(x_np))
    assert torch.equal(p.cpu(), ref)
    dp = rng.standard_normal(ref.shape).astype(np.float32)
    dx = ops.maxpool2_bwd(xv, p, dev(dp)).cpu().numpy()
    # reference: first max in window scan order
    exp = np.zeros_like(x_np)
    for n in range(2):
        for i in range(4):
            for j in range(6):
                win = x_np[n, 2 * i:2 * i + 2, 2 * j:2 * j + 2, :].reshape(4, 64)
                k = np.argmax(win, axis=0)          # np.argmax = first maximum
                for c in range(64):
                    exp[n, 2 * i + k[c] // 2, 2 * j + k[c] % 2, c] = dp[n, i, j, c]
    np.testing.assert_array_equal(dx, exp)


@pytest.mark.parametrize("shape", [(2, 4, 8, 128, 64), (1, 2, 2, 1024, 512), (2, 8, 8, 256, 128), (1, 5, 3, 128, 64)])
def test_deconv_concat_forward_backward(ops, shape):
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(cin)
    x = torch.tensor(rng.standard_normal((n, h, w, cin)), dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(rng.standard_normal((2, 2, cout, cin)) / math.sqrt(cin), dtype=torch.float64, requires_grad=True)
    b = torch.tensor(rng.standard_normal(cout) * 0.1, dtype=torch.float64, requires_grad=True)
    skip = rng.standard_normal((n, 2 * h, 2 * w, cout))
    up = torch.relu(tf_ops.conv_transpose_ks(x, wt, (2, 2), bias=b))
    cat_ref = torch.cat((torch.tensor(skip), up), dim=-1)
    dcat = rng.standard_normal(cat_ref.shape)
    cat_ref.backward(torch.tensor(dcat))
    cat = torch.zeros((n, 2 * h, 2 * w, 2 * cout), device="cuda")
    cat[..., :cout] = dev(skip)
    wp_f, wp_d = ops.deconv2x2_pack(dev(wt.detach().numpy()))
    ops.deconv2x2_fwd(dev(x.detach().numpy()), wp_f, dev(b.detach().numpy()), cat, cout, cout)
    assert rel_err(cat.cpu().numpy(), cat_ref.detach().numpy()) < 3e-6
    dx, dw, db = ops.deconv2x2_bwd(dev(x.detach().numpy()), wp_d, cat, dev(dcat), cout, cout)
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 5e-6
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < 5e-6


def test_deconv_backward_many_pixel_tiles_per_split_exact_and_reproducible(ops):
    """32768 input pixels = 256 pixel tiles over 128 splits: the filter-gradient kernel walks SEVERAL tiles per block
    (register-prefetched), which the small shapes above never do.  Checked against float64 and run-to-run bit equality
    (a mis-scheduled accumulator read-back once lost the last k-step of rows 27 / 31 of every 32-row panel here)."""
    n, h, w, cin, cout = 8, 64, 64, 128, 64
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g)
    wt = torch.randn((2, 2, cout, cin), device="cuda", generator=g) / math.sqrt(cin)
    b = torch.randn((cout,), device="cuda", generator=g) * 0.1
    cat = torch.zeros((n, 2 * h, 2 * w, 2 * cout), device="cuda")
    dcat = torch.randn(cat.shape, device="cuda", generator=g)
    for bf16 in (False, True):
        wp_f, wp_d = ops.deconv2x2_pack(wt, bf16)
        ops.deconv2x2_fwd(x, wp_f, b, cat, cout, cout, bf16)
        dx, dw, db = ops.deconv2x2_bwd(x, wp_d, cat, dcat, cout, cout, bf16)
        dx2, dw2, db2 = ops.deconv2x2_bwd(x, wp_d, cat, dcat, cout, cout, bf16)
        assert torch.equal(dw, dw2) and torch.equal(dx, dx2) and torch.equal(db, db2)
        dpre = dcat[..., cout:].double() * (cat[..., cout:] > 0).double()
        xr = x.double()
        if bf16:
            dpre, xr = dpre.float().bfloat16().double(), x.bfloat16().double()
        ref = torch.stack([torch.stack([torch.einsum("nhwo,nhwi->oi", dpre[:, a::2, c::2], xr) for c in range(2)])
                           for a in range(2)])
        assert rel_err(dw.cpu().numpy(), ref.cpu().numpy()) < 5e-6


HEAD_CASES = [
    ("none", None, 0.0, 3, "xentropy"),
    ("numerical", [0.2, 0.4, 4.4], 0.0, 3, "xentropy"),
    ("numerical", [0.0, 1.0], 0.0, 2, "xentropy"),       # zero weights leave the denominator
    ("proportion", None, 1000.0, 3, "xentropy"),
    ("proportion", None, 0.0, 2, "xentropy"),
    ("none", None, 0.0, 3, "dice"),
    ("none", None, 0.0, 2, "dice"),
]


@pytest.mark.parametrize("w_type,nw,decay,ncls,loss_type", HEAD_CASES)
def test_head_forward_backward(ops, w_type, nw, decay, ncls, loss_type):
    rng = np.random.default_rng(ncls * 7 + len(w_type))
    n, h, w, c = 3, 24, 20, 64
    z = torch.tensor(rng.standard_normal((n, h, w, c)), dtype=torch.float64, requires_grad=True)
    wl = torch.tensor(rng.standard_normal((1, 1, c, ncls)) * 0.3, dtype=torch.float64, requires_grad=True)
    bl = torch.tensor(rng.standard_normal(ncls) * 0.1, dtype=torch.float64, requires_grad=True)
    labels = torch.tensor(rng.integers(0, ncls, size=(n, h, w)))
    labels[0, :, :10] = 1
    logits = (z.reshape(-1, c) @ wl.reshape(c, ncls) + bl).reshape(n, h, w, ncls)
    kw = {}
    if w_type == "numerical":
        kw["numeric_w"] = nw
    elif w_type == "proportion" and decay > 0:
        kw["proportion_decay"] = decay
    if loss_type == "xentropy":
        lref = olosses.weighted_sparse_softmax_cross_entropy(logits.float(), labels, w_type, **kw)
    else:
        lref = olosses.sparse_dice_loss(torch.softmax(logits, -1).float(), labels)
    lref.backward()
    d = ops.head_desc(n, h * w, c, ncls, w_type, numeric_w=nw, proportion_decay=decay)
    zd, wd, bd = dev(z.detach().numpy()), dev(wl.detach().numpy().reshape(c, ncls)), dev(bl.detach().numpy())
    ld = labels.to(torch.int32).cuda()
    lg, probs, result, ws = ops.head_fwd(d, zd, wd, bd, ld, None, want_probs=True)
    res = result.cpu().numpy()
    assert rel_err(lg.cpu().numpy().reshape(n, h, w, ncls), logits.detach().numpy()) < 3e-6
    pref = torch.softmax(logits.detach(), -1).numpy()
    assert np.abs(probs.cpu().numpy().reshape(pref.shape) - pref).max() < 2e-6
    got_loss = res[0] if loss_type == "xentropy" else res[1]
    assert abs(got_loss - lref.item()) < 2e-5 * max(1.0, abs(lref.item()))
    # metrics on thresholded predictions (bit-exact counts where prob is not within 1e-6 of 0.5)
    preds = olosses.threshold_pred(torch.tensor(pref))
    for ci in range(1, ncls):
        lab = (labels == ci).unsqueeze(-1)
        sums = res[3:3 + n * (ncls - 1) * 4].reshape(n, ncls - 1, 4)[:, ci - 1]
        pr = preds[ci - 1].numpy().astype(np.float64)
        lb = lab.numpy().astype(np.float64)
        if np.abs(pref[..., ci] - 0.5).min() > 1e-5:
            np.testing.assert_array_equal(sums[:, 0], (pr * lb).sum((1, 2, 3)))
            np.testing.assert_array_equal(sums[:, 1], pr.sum((1, 2, 3)))
            np.testing.assert_array_equal(sums[:, 2], lb.sum((1, 2, 3)))
            np.testing.assert_array_equal(sums[:, 3], np.clip(pr + lb, 0, 1).sum((1, 2, 3)))
    xs, ds = (1.0, 0.0) if loss_type == "xentropy" else (0.0, 1.0)
    scales = torch.tensor([1.0, 1.0], device="cuda")
    dz, dw, db = ops.head_bwd(d, zd, wd, ld, None, lg, result, ws, xs, ds, scales)
    assert rel_err(dz.cpu().numpy(), z.grad.numpy()) < 2e-5
    assert rel_err(dw.cpu().numpy(), wl.grad.numpy().reshape(c, ncls)) < 2e-5
    assert rel_err(db.cpu().numpy(), bl.grad.numpy()) < 2e-5


def test_head_pixelmap_matches_numerical(ops):
    rng = np.random.default_rng(9)
    n, hw, c, ncls = 2, 96, 64, 3
    z, wl, bl = dev(rng.standard_normal((n * hw, c))), dev(rng.standard_normal((c, ncls)) * 0.2), dev(np.zeros(ncls))
    labels = torch.tensor(rng.integers(0, ncls, size=(n, hw)))
    wmap = olosses.compute_weights("numerical", labels, ncls, numeric_w=[0.2, 0.4, 4.4])
    d1 = ops.head_desc(n, hw, c, ncls, "numerical", numeric_w=[0.2, 0.4, 4.4])
    d2 = ops.head_desc(n, hw, c, ncls, "pixelmap")
    ld = labels.to(torch.int32).cuda()
    r1 = ops.head_fwd(d1, z, wl, bl, ld)[2].cpu().numpy()
    r2 = ops.head_fwd(d2, z, wl, bl, ld, wmap.float().cuda().contiguous())[2].cpu().numpy()
    assert abs(r1[0] - r2[0]) < 1e-5


def test_head_predict_bit_exact(ops):
    rng = np.random.default_rng(10)
    p = rng.random((1000, 3)).astype(np.float32)
    p[0] = [0.4, 0.4, 0.2]       # tie -> lowest index
    p[1] = [0.5, 0.5, 0.0]       # 0.5 is NOT > 0.5
    p[2] = [0.2, 0.3, 0.5000001]
    amax, preds = ops.head_predict(dev(p), 3)
    np.testing.assert_array_equal(amax.cpu().numpy(), np.argmax(p, -1).astype(np.uint8))
    np.testing.assert_array_equal(preds.cpu().numpy(), (p[:, 1:] > 0.5).T.astype(np.uint8))


def test_adam_and_momentum_match_tf_formulas(ops):
    rng = np.random.default_rng(12)
    n = 1003                                     # not a multiple of 4: exercises the tail
    p0 = rng.standard_normal(n).astype(np.float32)
    p = {"w": p0.astype(np.float64).copy()}
    opt = osolver.TFAdam(0.9, 0.99, 1e-8)
    buf = torch.zeros(1004, device="cuda")
    buf[:n] = dev(p0)
    pd, m, v = buf[:n], torch.zeros(1004, device="cuda")[:n], torch.zeros(1004, device="cuda")[:n]
    wd, lr = 1e-2, 1e-3
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        opt.step(p, {"w": g.astype(np.float64) + wd * p["w"]}, lr)
        lr_t = lr * math.sqrt(1 - 0.99 ** t) / (1 - 0.9 ** t)
        gb = torch.zeros(1004, device="cuda")
        gb[:n] = dev(g) * 2.0                    # gscale 0.5 undoes this (data-parallel mean)
        ops.adam_step(pd, gb[:n], m, v, lr_t, 0.9, 0.99, 1e-8, 0.5, wd)
    np.testing.assert_allclose(pd.cpu().numpy(), p["w"], rtol=2e-5, atol=2e-6)
    pm = {"w": p0.astype(np.float64).copy()}
    mom = osolver.TFMomentum(0.9, False)
    pd2, acc = dev(p0), torch.zeros(n, device="cuda")
    for t in range(3):
        g = rng.standard_normal(n).astype(np.float32)
        mom.step(pm, {"w": g.astype(np.float64)}, 0.1)
        ops.momentum_step(pd2, dev(g), acc, 0.1, 0.9, False)
    np.testing.assert_allclose(pd2.cpu().numpy(), pm["w"], rtol=2e-5, atol=2e-6)
    s = ops.sumsq(pd2).item()
    assert abs(s - float((pd2.double() ** 2).sum())) < 1e-3 * s
    # AdamW (tf.contrib.opt.AdamWOptimizer, solver.py:212-216): var <- var*(1 - wd), then the Adam update
    pw = rng.standard_normal(n).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    ref = {"w": pw.astype(np.float64).copy()}
    ref["w"] *= (1 - 0.05)
    opt2 = osolver.TFAdam(0.9, 0.99, 1e-8)
    dec = pw.astype(np.float64) * (1 - 0.05)
    tmp = {"w": pw.astype(np.float64).copy()}
    opt2.step(tmp, {"w": g.astype(np.float64)}, 1e-3)
    expect = dec + (tmp["w"] - pw.astype(np.float64))
    pd3, m3, v3 = dev(pw), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ops.adam_step(pd3, dev(g), m3, v3, 1e-3 * math.sqrt(1 - 0.99) / (1 - 0.9), 0.9, 0.99, 1e-8, 1.0, 0.0, 0.05)
    np.testing.assert_allclose(pd3.cpu().numpy(), expect, rtol=2e-5, atol=2e-6)


def test_boundary_weights_match_oracle_incl_single_label_slice():
    """unetk_boundary_weights vs the oracle's scipy restatement (loss_metrics.py:149-165): 3-class synthetic labels,
    a slice with one label only (no ring: scipy's no-background convention) and a non-square ragged size."""
    from boxsegliver_amd import ops
    from boxsegliver_amd.data.synthetic import make_batch
    from oracle import losses
    _, labels, _ = make_batch(3, 64, 48, 3, 3, 99)
    labels[1] = 0
    lab = torch.from_numpy(labels)
    got = ops.boundary_weights(lab.cuda()).cpu().numpy()
    ref = losses.compute_weights("boundary", lab.long(), 3).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-6)
    again = ops.boundary_weights(lab.cuda()).cpu().numpy()
    assert np.array_equal(got, again)
    rng = np.random.default_rng(3)
    lab = torch.from_numpy(rng.integers(0, 3, size=(2, 37, 301)).astype(np.int32))
    np.testing.assert_allclose(ops.boundary_weights(lab.cuda()).cpu().numpy(),
                               losses.compute_weights("boundary", lab.long(), 3).numpy(), rtol=2e-6)
