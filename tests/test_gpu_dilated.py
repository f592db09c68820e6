"""Rate-2 atrous 3x3 conv (slim.conv2d(x, C, 3, rate=2): reference NetworksV2/SmallUNet.py:44-49, InterUNet.py) through
the C ABI: unetk_conv_desc.dilation = 2 -- forward (+ statistic partials), input gradient and filter gradient of the
tiled fp32 MFMA kernels against float64 convolutions on the device; the conv unit (conv + norm + ReLU) end to end."""
import numpy as np
import pytest
import torch

from oracle import tf_ops

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 32, 32, 64, 128), (3, 20, 24, 128, 64), (1, 16, 40, 64, 64), (2, 9, 7, 64, 128),
                                            (4, 32, 32, 512, 1024)])
def test_dilated_conv_forward_dgrad_wgrad(n, h, w, cin, cout):
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(h * w + cin)
    x = torch.randn(n, h, w, cin, generator=gen).cuda()
    wt = (torch.randn(3, 3, cin, cout, generator=gen) / (9 * cin) ** 0.5).cuda()
    dy = torch.randn(n, h, w, cout, generator=gen).cuda()
    wp_f, wp_d = ops.conv3x3_pack(wt)
    y, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True, dilation=2)
    x64 = x.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    ref = tf_ops.conv_nd_same(x64, w64, dilation=2)
    ref.backward(dy.double())
    assert rel(y.double(), ref.detach()) < 1e-5          # fp32 accumulation over K = 9 Cin (up to 9216 terms)
    # the epilogue's statistic partials: column sums of y and y^2 over all tiles
    assert stats.shape == (2, rows, cout)
    assert rel(stats[0].double().sum(0), ref.detach().sum((0, 1, 2))) < 1e-5
    assert rel(stats[1].double().sum(0), (ref.detach() ** 2).sum((0, 1, 2))) < 1e-5
    dx = ops.conv3x3_dgrad(dy, wp_d, cin, dilation=2)
    assert rel(dx.double(), x64.grad) < 1e-5
    dw = ops.conv3x3_wgrad(x, dy, dilation=2)
    assert rel(dw.double(), w64.grad) < 1e-5
    assert torch.equal(dw, ops.conv3x3_wgrad(x, dy, dilation=2)) and torch.equal(dx, ops.conv3x3_dgrad(dy, wp_d, cin, dilation=2))
    # and it really is not the dense conv
    y1, _, _ = ops.conv3x3_fwd(x, wp_f, cout, want_stats=False)
    assert not torch.allclose(y1, y)


def test_dilated_conv_on_channel_slices_and_unsupported_shapes():
    from boxsegliver_amd import _abi, ops
    gen = torch.Generator().manual_seed(2)
    buf = torch.randn(2, 16, 16, 192, generator=gen).cuda()
    x = buf[..., 64:128]
    wt = (torch.randn(3, 3, 64, 64, generator=gen) / 24).cuda()
    wp_f, _ = ops.conv3x3_pack(wt)
    y, _, _ = ops.conv3x3_fwd(x, wp_f, 64, want_stats=False, dilation=2)
    ref = tf_ops.conv_nd_same(x.double().contiguous(), wt.double(), dilation=2)
    assert rel(y.double(), ref) < 3e-6
    with pytest.raises(_abi.UnetkError):                               # Cout = 32: no atrous tile configuration
        w32 = torch.randn(3, 3, 64, 32).cuda()
        ops.conv3x3_fwd(x, ops.conv3x3_pack(w32)[0], 32, want_stats=False, dilation=2)
    with pytest.raises(_abi.UnetkError):
        ops.conv3x3_fwd(x, wp_f, 64, want_stats=False, dilation=3)


@pytest.mark.parametrize("kind", ["batch_norm", "instance_norm"])
def test_dilated_conv_unit_forward_backward(kind):
    """conv(rate 2) -> norm -> ReLU as one autograd node, against float64 autograd of the oracle ops."""
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(7)
    n, h, w, cin, cout = 2, 16, 16, 64, 128
    x = torch.randn(n, h, w, cin, generator=gen)
    wt = torch.randn(3, 3, cin, cout, generator=gen) / 24
    gamma = 0.5 + torch.rand(cout, generator=gen)
    beta = 0.2 * torch.randn(cout, generator=gen)
    dz = torch.randn(n, h, w, cout, generator=gen)
    xd, wd, gd, bd = (t.cuda().requires_grad_(True) for t in (x, wt, gamma, beta))
    spec = ops.NormSpec(kind, 1e-3 if kind == "batch_norm" else 1e-6, 0.99, True)
    mm, mv = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    z = ops.Conv3x3NormRelu.apply(xd, wd, gd, bd, mm, mv, spec, None, None, None, None, None, 2)
    z.backward(dz.cuda())
    x64, w64, g64, b64 = (t.double().requires_grad_(True) for t in (x, wt, gamma, beta))
    y = tf_ops.conv_nd_same(x64, w64, dilation=2)
    if kind == "batch_norm":
        t, _, _ = tf_ops.batch_norm(y, g64, b64, torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64), True)
    else:
        t = tf_ops.instance_norm(y, g64, b64, eps=1e-6)
    ref = torch.relu(t)
    ref.backward(dz.double())
    assert rel(z.detach().cpu().double(), ref.detach()) < 2e-5
    # gradients: a few ReLU mask flips at pre-activations within rounding of zero -> L2
    for got, want in ((xd.grad, x64.grad), (wd.grad, w64.grad), (gd.grad, g64.grad), (bd.grad, b64.grad)):
        l2 = float((got.cpu().double() - want).norm() / want.norm())
        assert l2 < 1e-3, l2
