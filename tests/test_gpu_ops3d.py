"""GPU parity of the 3-D conv composition (unetk_conv3d_*) against the CPU oracle: every (kernel, stride)
pair of UNet3D's _ModelConfig (UNet3D.py:31-91), even and odd extents (TF SAME asymmetric padding), the
32-channel tile configurations, forward + norm-statistic partials + input gradient + filter gradient."""
import math

import numpy as np
import pytest
import torch

from oracle import tf_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from boxsegliver_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float32).cuda()


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


CASES = [
    # N, D, H, W, Cin, Cout, kd, (sd, sh, sw)
    (2, 3, 16, 16, 32, 32, 1, (1, 1, 1)),      # conv_e0/conv2, conv_d0: (1,3,3)
    (1, 4, 16, 32, 32, 64, 1, (1, 2, 2)),      # conv_e1/conv1: (1,3,3) stride (1,2,2), even sizes
    (1, 2, 15, 17, 32, 64, 1, (1, 2, 2)),      # odd sizes: pad 1 before
    (2, 4, 8, 16, 64, 64, 3, (1, 1, 1)),       # (3,3,3) stride 1
    (1, 5, 8, 8, 64, 128, 3, (1, 2, 2)),       # conv_e2/conv1: (3,3,3) stride (1,2,2)
    (1, 6, 8, 16, 64, 64, 3, (2, 2, 2)),       # bridge/conv1: (3,3,3) stride (2,2,2), even depth
    (1, 5, 6, 6, 64, 64, 3, (2, 2, 2)),        # odd depth / odd result sizes
    (1, 2, 8, 16, 128, 32, 1, (1, 1, 1)),      # 64 -> 30 style (conv_d0/conv1)
    # small planes -> linear-pixel kernel (conv_igemm_lin.hip): blocks span rows and planes, never samples
    (2, 8, 12, 12, 64, 128, 3, (1, 1, 1)),     # conv_e3 / conv_d3 shape: 1152 pixels per sample = 9 blocks
    (2, 7, 6, 6, 64, 64, 3, (1, 1, 1)),        # bridge shape: 252 pixels per sample (last block partial)
    (1, 5, 24, 24, 32, 64, 3, (1, 1, 1)),      # conv_e2 shape
    (3, 4, 11, 13, 32, 64, 1, (1, 1, 1)),      # odd plane, three samples, (1,3,3)
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_forward_backward(ops, case):
    n, dd, h, w, cin, cout, kd, stride = case
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    x = torch.tensor(rng.standard_normal((n, dd, h, w, cin)), dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(rng.standard_normal((kd, 3, 3, cin, cout)) / math.sqrt(9 * kd * cin), dtype=torch.float64,
                      requires_grad=True)
    y_ref = tf_ops.conv_nd_same(x, wt, stride=stride)
    dy = rng.standard_normal(tuple(y_ref.shape))
    y_ref.backward(torch.tensor(dy))
    d = ops.conv3d_desc(x.shape, cout, kd, stride)
    assert ops.conv3d_out_shape(d) == tuple(y_ref.shape)
    wp_f, wp_d = ops.conv3d_pack(dev(wt.detach().numpy()))
    y, stats, rows = ops.conv3d_fwd(dev(x.detach().numpy()), wp_f, d, want_stats=True)
    ref = y_ref.detach().numpy()
    assert rel_err(y.cpu().numpy(), ref) < 3e-6
    s = stats.cpu().numpy().astype(np.float64)
    assert rows % n == 0                                            # each sample's rows are contiguous
    per = s.reshape(2, n, rows // n, cout).sum(2)
    np.testing.assert_allclose(per[0], ref.sum((1, 2, 3)), atol=3e-4 * max(1.0, np.abs(ref).sum((1, 2, 3)).max()))
    np.testing.assert_allclose(per[1], (ref ** 2).sum((1, 2, 3)), rtol=3e-5)
    dx = ops.conv3d_dgrad(dev(dy), wp_d, d)
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 5e-6
    dw = ops.conv3d_wgrad(dev(x.detach().numpy()), dev(dy), d)
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6


def _mask8(segs):
    m = 0
    for lo, hi in segs:
        for g in range(lo // 8, (hi + 7) // 8):
            m |= 1 << g
    return m


LIVE_CASES = [
    # N, D, H, W, Cin, live Cin ranges, Cout, live Cout ranges, kd, stride -- UNet3D's channel-padded layers (NetworksV2/padded.py)
    (1, 8, 12, 12, 256, [(0, 240)], 256, [(0, 240)], 3, (1, 1, 1)),                 # 240 in 256: chunk 15 is skipped
    (1, 6, 24, 24, 128, [(0, 120)], 128, [(0, 120)], 3, (1, 1, 1)),                 # 120 in 128: chunk 7 runs half its MFMAs
    (1, 8, 12, 12, 512, [(0, 240), (256, 496)], 256, [(0, 240)], 3, (1, 1, 1)),     # concat(skip, up) of two padded halves
    (1, 4, 24, 24, 256, [(0, 120), (128, 248)], 128, [(0, 120)], 3, (1, 1, 1)),
    (2, 4, 24, 24, 128, [(0, 120)], 256, [(0, 240)], 3, (1, 2, 2)),                 # grouped-tap forward, four-class input gradient
    (1, 8, 12, 12, 256, [(0, 240)], 320, [(0, 320)], 3, (2, 2, 2)),                 # the bridge
    (3, 16, 12, 12, 256, [(0, 240)], 256, [(0, 240)], 3, (1, 1, 1)),                # whole tiles + stream-K remainder
]


@pytest.mark.parametrize("case", LIVE_CASES)
def test_conv3d_live_channel_masks(ops, case):
    """unetk_conv3d_desc.cin_live8 / cout_live8: skipping the padded groups of the contraction axis gives the result of
    contracting their zeros -- against the oracle on the padded tensors and against the same call without the masks."""
    n, dd, h, w, cin, cin_live, cout, cout_live, kd, stride = case
    rng = np.random.default_rng(cin * 7 + cout + h)
    ci = np.zeros(cin, bool)
    co = np.zeros(cout, bool)
    for lo, hi in cin_live:
        ci[lo:hi] = True
    for lo, hi in cout_live:
        co[lo:hi] = True
    x_np = rng.standard_normal((n, dd, h, w, cin)) * ci
    w_np = rng.standard_normal((kd, 3, 3, cin, cout)) / math.sqrt(9 * kd * cin) * ci[:, None] * co[None, :]
    x = torch.tensor(x_np, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w_np, dtype=torch.float64, requires_grad=True)
    y_ref = tf_ops.conv_nd_same(x, wt, stride=stride)
    dy = rng.standard_normal(tuple(y_ref.shape)) * co
    y_ref.backward(torch.tensor(dy))
    live8 = (_mask8(cin_live), _mask8(cout_live))
    d_live = ops.conv3d_desc(x.shape, cout, kd, stride, live8=live8)
    d_all = ops.conv3d_desc(x.shape, cout, kd, stride)
    assert tuple(d_live.cin_live8) == (live8[0] & 0xFFFFFFFF, live8[0] >> 32)
    wp_f, wp_d = ops.conv3d_pack(dev(w_np))
    xd, dyd = dev(x_np), dev(dy)
    y, stats, rows = ops.conv3d_fwd(xd, wp_f, d_live, want_stats=True)
    y0, stats0, rows0 = ops.conv3d_fwd(xd, wp_f, d_all, want_stats=True)
    ref = y_ref.detach().numpy()
    assert rel_err(y.cpu().numpy(), ref) < 3e-6
    assert rel_err(y.cpu().numpy(), y0.cpu().numpy()) < 1e-6
    assert rows == rows0
    np.testing.assert_allclose(stats.cpu().numpy().reshape(2, -1, cout).sum(1), stats0.cpu().numpy().reshape(2, -1, cout).sum(1),
                               rtol=2e-5, atol=2e-3)
    dx = ops.conv3d_dgrad(dyd, wp_d, d_live)
    dx0 = ops.conv3d_dgrad(dyd, wp_d, d_all)
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 5e-6
    assert rel_err(dx.cpu().numpy(), dx0.cpu().numpy()) < 1e-6
    assert float(dx[..., ~torch.as_tensor(ci)].abs().max()) == 0.0 if not ci.all() else True
    # bit-reproducible with the masks as without
    assert torch.equal(ops.conv3d_fwd(xd, wp_f, d_live, want_stats=False)[0], y)
    assert torch.equal(ops.conv3d_dgrad(dyd, wp_d, d_live), dx)
    dw = ops.conv3d_wgrad(xd, dyd, d_live)                          # the filter gradient contracts pixels: masks unused
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6


DECONV_CASES = [
    # N, D, H, W, Cin, Cout, kd
    (1, 3, 4, 8, 128, 64, 1),      # conv_d1/up: (1,2,2)
    (2, 2, 8, 8, 64, 32, 1),       # conv_d0/up: 60 -> 30 padded to 64 -> 32
    (1, 2, 3, 4, 320, 256, 2),     # conv_d3/up: (2,2,2)
    (2, 3, 2, 2, 64, 64, 2),
]


@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv3d_concat_forward_backward(ops, case):
    n, dd, h, w, cin, cout, kd = case
    rng = np.random.default_rng(cin + kd)
    x = torch.tensor(rng.standard_normal((n, dd, h, w, cin)), dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(rng.standard_normal((kd, 2, 2, cout, cin)) / math.sqrt(cin), dtype=torch.float64, requires_grad=True)
    skip = rng.standard_normal((n, kd * dd, 2 * h, 2 * w, cout))
    up = torch.relu(tf_ops.conv_transpose_ks(x, wt, (kd, 2, 2)))          # UNet3D: no bias, ReLU
    cat_ref = torch.cat((torch.tensor(skip), up), dim=-1)
    dcat = rng.standard_normal(tuple(cat_ref.shape))
    cat_ref.backward(torch.tensor(dcat))
    cat = torch.zeros((n, kd * dd, 2 * h, 2 * w, 2 * cout), device="cuda")
    cat[..., :cout] = dev(skip)
    wp_f, wp_d = ops.deconv3d_pack(dev(wt.detach().numpy()))
    ops.deconv3d_fwd(dev(x.detach().numpy()), wp_f, None, cat, cout, cout, kd)
    assert rel_err(cat.cpu().numpy(), cat_ref.detach().numpy()) < 3e-6
    dx, dw, db = ops.deconv3d_bwd(dev(x.detach().numpy()), wp_d, cat, dev(dcat), cout, cout, kd, want_dbias=False)
    assert db is None
    assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 5e-6
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6


def test_conv3d_first_layer_single_channel(ops):
    # conv_e0/conv1: Cin = 1 (direct kernel on raw DHWIO filters), Cout = 30 padded to 32
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.standard_normal((2, 3, 16, 32, 1)), dtype=torch.float64)
    wt = torch.tensor(rng.standard_normal((1, 3, 3, 1, 32)) / 3, dtype=torch.float64, requires_grad=True)
    y_ref = tf_ops.conv_nd_same(x, wt)
    dy = rng.standard_normal(tuple(y_ref.shape))
    y_ref.backward(torch.tensor(dy))
    d = ops.conv3d_desc(x.shape, 32, 1, (1, 1, 1))
    y, stats, rows = ops.conv3d_fwd(dev(x.numpy()), dev(wt.detach().numpy()), d, want_stats=True)
    assert rel_err(y.cpu().numpy(), y_ref.detach().numpy()) < 3e-6
    dw = ops.conv3d_wgrad(dev(x.numpy()), dev(dy), d)
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6
