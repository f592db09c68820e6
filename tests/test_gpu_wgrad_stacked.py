"""Filter gradient on small planes (csrc/conv_wgrad.hip, "stacked planes" variant): 10 x 12 pixel tiles walking the planes
stacked with one shared zero row -- chosen for widths that are multiples of 12 but not of 16 (UNet3D's 12^2 / 24^2
levels).  Checked against float64 autograd of the oracle's conv on the device, bit-reproducible, and through the 3-D
wrapper (depth taps = plane views with strides)."""
import numpy as np
import pytest
import torch

from oracle import tf_ops

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("n,h,w,cin,cout", [(6, 12, 12, 64, 128), (5, 24, 24, 128, 64), (3, 10, 36, 64, 64), (7, 7, 12, 64, 64),
                                            (1, 12, 12, 64, 64), (192, 12, 12, 256, 256), (2, 64, 24, 64, 64),
                                            (96, 6, 6, 320, 320), (5, 6, 18, 64, 64), (3, 9, 6, 64, 128)])   # 20 x 6 tiles
def test_stacked_wgrad_matches_float64(n, h, w, cin, cout):
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, h, w, cin, generator=gen).cuda()
    dy = torch.randn(n, h, w, cout, generator=gen).cuda()
    dw = ops.conv3x3_wgrad(x, dy)
    x64 = x.double()
    w64 = torch.zeros(3, 3, cin, cout, dtype=torch.float64, device="cuda", requires_grad=True)
    tf_ops.conv_nd_same(x64, w64).backward(dy.double())
    assert rel(dw.double(), w64.grad) < 5e-6
    assert torch.equal(dw, ops.conv3x3_wgrad(x, dy))                      # fixed summation order


def test_stacked_wgrad_on_channel_slices_of_wider_buffers():
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(4)
    n, h, w = 4, 12, 24
    xb = torch.randn(n, h, w, 192, generator=gen).cuda()
    dyb = torch.randn(n, h, w, 128, generator=gen).cuda()
    x, dy = xb[..., 64:192], dyb[..., :64]
    dw = ops.conv3x3_wgrad(x, dy)
    w64 = torch.zeros(3, 3, 128, 64, dtype=torch.float64, device="cuda", requires_grad=True)
    tf_ops.conv_nd_same(x.double().contiguous(), w64).backward(dy.double().contiguous())
    assert rel(dw.double(), w64.grad) < 5e-6


@pytest.mark.parametrize("kd,stride,shape", [
    (3, (1, 2, 2), (2, 5, 48, 48, 64, 128)),      # 4 x 12 output tiles (Wo = 24)
    (1, (1, 2, 2), (1, 4, 96, 96, 32, 64)),       # 4 x 16 output tiles, 32-channel input panel (UNet3D conv_e1/conv1)
    (3, (2, 2, 2), (2, 6, 24, 24, 256, 320)),     # bridge: depth stride 2 too, Cout = 5 x 64
    (3, (1, 2, 2), (1, 3, 23, 21, 64, 64)),       # odd extents: SAME pads 1 before
    (3, (2, 2, 2), (1, 5, 12, 12, 128, 64)),      # Wo = 6 < tile
])
def test_conv3d_strided_wgrad_native(kd, stride, shape):
    """Stride-2 filter gradient on tiles of output pixels (no zero-dilated dy): csrc/conv_wgrad.hip, S = 2."""
    from boxsegliver_amd import ops
    n, d, h, w, cin, cout = shape
    gen = torch.Generator().manual_seed(h * w + cin)
    x = torch.randn(n, d, h, w, cin, generator=gen).cuda()
    desc = ops.conv3d_desc(x.shape, cout, kd, stride)
    do, ho, wo = -(-d // stride[0]), -(-h // stride[1]), -(-w // stride[2])
    dy = torch.randn(n, do, ho, wo, cout, generator=gen).cuda()
    dw = ops.conv3d_wgrad(x, dy, desc)
    w64 = torch.zeros(kd, 3, 3, cin, cout, dtype=torch.float64, device="cuda", requires_grad=True)
    tf_ops.conv_nd_same(x.double(), w64, stride).backward(dy.double())
    assert rel(dw.double(), w64.grad) < 5e-6
    assert torch.equal(dw, ops.conv3d_wgrad(x, dy, desc))


@pytest.mark.parametrize("kd,stride", [(3, (1, 1, 1)), (3, (1, 2, 2)), (1, (1, 1, 1))])
def test_conv3d_wgrad_on_12x12_and_24x24_planes(kd, stride):
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(kd + stride[1])
    n, d, h, w, cin, cout = 2, 6, 24, 24, 64, 128
    x = torch.randn(n, d, h, w, cin, generator=gen).cuda()
    desc = ops.conv3d_desc(x.shape, cout, kd, stride)
    do, ho, wo = -(-d // stride[0]), -(-h // stride[1]), -(-w // stride[2])
    dy = torch.randn(n, do, ho, wo, cout, generator=gen).cuda()
    dw = ops.conv3d_wgrad(x, dy, desc)
    w64 = torch.zeros(kd, 3, 3, cin, cout, dtype=torch.float64, device="cuda", requires_grad=True)
    tf_ops.conv_nd_same(x.double(), w64, stride).backward(dy.double())
    assert rel(dw.double(), w64.grad) < 5e-6
    assert torch.equal(dw, ops.conv3d_wgrad(x, dy, desc))
