"""GPU: the UNETK_BF16 mode (BASELINE.json configs[2], "UNet 512x512x3 bs=64 bf16").

Operands of the 3x3 contractions are rounded to bf16 (RNE) on their way into v_mfma_f32_32x32x16_bf16; products of
bf16 values are exact in fp32 and the accumulation is fp32, so against an fp64 convolution of the SAME bf16-rounded
operands the kernels must agree to fp32 accumulation error (~1e-6) -- that is the op-level bar.  End to end the
bf16 run is compared with the fp32 oracle at mixed-precision tolerances (stated in the tests)."""
import math

import numpy as np
import pytest
import torch

from oracle import tf_ops

pytestmark = pytest.mark.gpu


def bf16_round(a):
    """Round-to-nearest-even to bf16, returned as float64 ndarray."""
    return torch.as_tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


SHAPES = [
    # N, H, W, Cin, Cout                       tile configuration
    (1, 32, 32, 64, 128),      # 512x128 (8 waves)
    (2, 40, 20, 32, 128),      # 512x128, masked edges in H and W
    (1, 8, 16, 64, 128),       # 128x128 (planes lower than 24 rows)
    (1, 64, 16, 64, 64),       # 512x64
    (3, 8, 48, 96, 64),        # 128x64, 3 chunks
    (2, 16, 32, 32, 32),       # 256x32 (UNet3D's padded 30-channel levels)
    (1, 4, 4, 32, 256),        # image smaller than a tile
    (1, 24, 16, 256, 256),     # 8 chunks, 2 N tiles
]


@pytest.mark.parametrize("shape", SHAPES)
def test_bf16_conv_fwd_dgrad_wgrad_match_fp64_on_rounded_operands(shape):
    from boxsegliver_amd import ops
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(abs(hash(shape)) % 2**31)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) / math.sqrt(9 * cin)).astype(np.float32)
    dy = rng.standard_normal((n, h, w, cout)).astype(np.float32)
    xr, wr, dyr = bf16_round(x), bf16_round(wt), bf16_round(dy)

    xt = torch.tensor(xr, requires_grad=True)
    wtt = torch.tensor(wr, requires_grad=True)
    ref = tf_ops.conv_nd_same(xt, wtt)
    wp_f, wp_d = ops.conv3x3_pack(dev(wt), bf16=True)
    assert wp_f.dtype == torch.bfloat16 and wp_f.numel() == 9 * cin * cout
    y, stats, rows = ops.conv3x3_fwd(dev(x), wp_f, cout, want_stats=True, bf16=True)
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), ref.detach().numpy()) < 3e-6
    s = stats.cpu().numpy().astype(np.float64)
    assert s.shape == (2, rows, cout)
    refn = ref.detach().numpy()
    np.testing.assert_allclose(s[0].sum(0), refn.sum((0, 1, 2)), atol=3e-4 * max(1, np.abs(refn).sum((0, 1, 2)).max()))
    np.testing.assert_allclose(s[1].sum(0), (refn ** 2).sum((0, 1, 2)), rtol=3e-5)

    # backward: dgrad uses bf16(dy) x bf16(w); wgrad uses bf16(x) x bf16(dy)
    ref.backward(torch.tensor(dyr))
    dx = ops.conv3x3_dgrad(dev(dy), wp_d, cin, bf16=True)
    dw = ops.conv3x3_wgrad(dev(x), dev(dy), bf16=True)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < 3e-6
    assert rel_err(dw.cpu().numpy(), wtt.grad.numpy()) < 5e-6
    dw2 = ops.conv3x3_wgrad(dev(x), dev(dy), bf16=True)
    assert torch.equal(dw, dw2)                                        # bit-reproducible split-K


def test_bf16_mode_is_close_to_fp32_and_rejects_unsupported_channels():
    from boxsegliver_amd import _abi, ops
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 32, 32, 64)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, 64, 64)) / 24).astype(np.float32)
    wp32, _ = ops.conv3x3_pack(dev(wt))
    y32, _, _ = ops.conv3x3_fwd(dev(x), wp32, 64, want_stats=False)
    wp16, _ = ops.conv3x3_pack(dev(wt), bf16=True)
    y16, _, _ = ops.conv3x3_fwd(dev(x), wp16, 64, want_stats=False, bf16=True)
    err = rel_err(y16.cpu().numpy(), y32.cpu().numpy())
    assert 1e-5 < err < 2e-2                                           # really bf16 operands, and no worse than that
    assert not ops.conv_uses_bf16(16, 64) and not ops.conv_uses_bf16(3, 64)
    with pytest.raises(_abi.UnetkError):
        w16 = torch.zeros(9 * 16 * 64, dtype=torch.bfloat16, device="cuda")
        ops.conv3x3_fwd(dev(rng.standard_normal((1, 8, 16, 16))), w16, 64, want_stats=False, bf16=True)
