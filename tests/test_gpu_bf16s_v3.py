"""GPU: the round-3 bf16-storage conv kernel (csrc/conv_igemm_bf16s.hip: persistent 512-pixel tiles, direct-to-LDS staging,
v_mfma_f32_16x16x32_bf16) -- slim.conv2d(x, C, 3) forward / input gradient of NetworksV2/UNet.py:79,85,94 in the mode of
BASELINE.json configs[2].  Same bar as tests/test_gpu_bf16s.py: against a float64 evaluation of the SAME bf16 operands the
stored bf16 value is within one bf16 ulp and equals the rounding of the exact result on all but a 2e-3 share; statistics come
from the fp32 accumulators.  Shapes chosen to walk every path of the persistent loop: several tiles per block, ragged tile
rows / columns, two to six 32-channel chunks per tile (the staging ring's phase carries over from tile to tile),
one and two output-channel tiles, 64-channel outputs, channel-slice views of wider buffers, the fused norm-backward
reduction, and the round-2 kernels on the shared filter pack (small planes)."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

ULP = 2.0 ** -8


def _r(t):
    return t.float().bfloat16().double()


def _stored_ok(got_bf16, ref64, flips=2e-3):
    got = got_bf16.double()
    scale = ref64.abs().clamp_min(1e-30)
    err = (got - ref64).abs() / scale
    big = ref64.abs() > 1e-3 * ref64.abs().max()
    assert err[big].max().item() <= 1.01 * ULP, err[big].max().item()
    exact = (got == _r(ref64.float()))
    assert exact.double().mean().item() > 1.0 - flips, exact.double().mean().item()


def _v3_rows(n, h, w):
    return n * ((h + 31) // 32) * ((w + 15) // 16)


SHAPES = [
    # N, H, W, Cin, Cout                       tiles (pixel tiles x couts tiles), chunks
    (8, 128, 128, 64, 128),    # 256 tiles, one per CU; 2 chunks
    (2, 256, 256, 128, 128),   # 512 tiles: two per block; 4 chunks
    (6, 200, 72, 192, 128),    # ragged rows (200 = 6 x 32 + 8) and columns (72 = 4 x 16 + 8); 6 chunks
    (16, 96, 80, 64, 128),     # two chunks per tile (the storage mode needs Cin % 64 == 0): look-aheads cross tile boundaries early
    (4, 128, 128, 64, 256),    # two couts tiles per pixel tile
    (8, 256, 128, 64, 64),     # 64 output channels (4 x 4 accumulator tiles per wave), 1024 tiles = four per block
    (14, 100, 50, 128, 64),    # ragged, 64 couts, 4 chunks
    (1, 512, 512, 64, 128),    # one image, 512 tiles
    # round 5, the drained epilogue of the 64-channel tiles (a tile's output is converted / stored / summed inside the NEXT tile's
    # K loop): four chunks with four tiles per block; ragged planes with up to two tiles per block (ragged tiles take the one-shot
    # epilogue between drained ones); three tiles per block in one image
    (8, 256, 128, 128, 64),
    (12, 200, 72, 64, 64),
    (3, 512, 256, 64, 64),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_v3_forward_and_input_gradient(shape):
    import torch.nn.functional as F
    from boxsegliver_amd import _abi, ops
    n, h, w, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(17 * n + cin + cout + h)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16()
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    x64 = x.double().permute(0, 3, 1, 2)
    w64 = _r(wt).permute(3, 2, 0, 1)
    ref = F.conv2d(x64, w64, padding=1).permute(0, 2, 3, 1)
    wp_f, wp_d = ops.conv3x3_pack(wt, bf16=_abi.BF16S)
    y, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True, bf16=_abi.BF16S)
    assert rows == _v3_rows(n, h, w), "the round-3 kernel was expected to take this shape"
    assert y.dtype == torch.bfloat16
    _stored_ok(y, ref)
    s = stats.double()
    assert s.shape == (2, rows, cout)
    tol = 3e-5 * ref.abs().sum((0, 1, 2)).max().item()
    assert (s[0].sum(0) - ref.sum((0, 1, 2))).abs().max().item() < tol
    assert ((s[1].sum(0) - (ref ** 2).sum((0, 1, 2))).abs() / (ref ** 2).sum((0, 1, 2))).max().item() < 3e-5
    # every tile's partial row on its own (a misplaced or stale row would still sum right only by accident)
    th, tw = (h + 31) // 32, (w + 15) // 16
    r4 = ref.float().double()
    for (ni, ti, tj) in [(0, 0, 0), (n - 1, th - 1, tw - 1), (n // 2, th // 2, tw - 1)]:
        blk = r4[ni, ti * 32:(ti + 1) * 32, tj * 16:(tj + 1) * 16, :]
        row = (ni * th + ti) * tw + tj
        assert (s[0, row] - blk.sum((0, 1))).abs().max().item() < 1e-4 * max(1.0, blk.abs().sum((0, 1)).max().item())
    assert torch.equal(y, ops.conv3x3_fwd(x, wp_f, cout, want_stats=False, bf16=_abi.BF16S)[0])       # bit-reproducible
    del ref
    # input gradient: the same kernel with Cin <-> Cout on the flipped pack (its output-channel count decides the tile)
    if cin % 128 == 0 or cin == 64:
        dy = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16()
        dref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w64, padding=1).permute(0, 2, 3, 1)
        dx = ops.conv3x3_dgrad(dy, wp_d, cin, bf16=_abi.BF16S)
        _stored_ok(dx, dref)


def test_v3_reads_and_writes_channel_slices_of_wider_buffers():
    """The zero-copy concat layout: x is channels [64, 128) of a 128-channel buffer, y goes to channels [0, 128) of a 256-wide one."""
    import torch.nn.functional as F
    from boxsegliver_amd import _abi, ops
    n, h, w, cin, cout = 8, 128, 128, 64, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    xbuf = torch.randn((n, h, w, 128), device="cuda", generator=g).bfloat16()
    x = xbuf[..., 64:]
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    ybuf = torch.full((n, h, w, 256), 7.0, device="cuda").bfloat16()
    wp_f, _ = ops.conv3x3_pack(wt, bf16=_abi.BF16S)
    y, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True, bf16=_abi.BF16S, y=ybuf[..., :128])
    assert rows == _v3_rows(n, h, w)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), _r(wt).permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    _stored_ok(ybuf[..., :128], ref)
    assert torch.all(ybuf[..., 128:] == 7.0)                   # the neighbouring channels were not touched


@pytest.mark.parametrize("kind", ["batch_norm", "instance_norm"])
def test_v3_fused_norm_backward_reduction(kind):
    """conv2's input gradient emitting conv1's norm-backward partials (unetk_conv3x3_dgrad_nbr) on the round-3 kernel ==
    the separate reduction pass (tests/test_gpu_fused_nbr.py's comparison, at a shape the persistent kernel takes)."""
    from boxsegliver_amd import ops
    from test_gpu_fused_nbr import _pair
    n, h, w, c0, c1, c2 = 8, 128, 128, 64, 128, 128
    d = ops.ConvDesc(n, h, w, c1, c2, c1, c2, 2, 1)
    assert ops._abi.lib().unetk_conv3x3_dgrad_nbr_rows(ctypes.byref(d)) == _v3_rows(n, h, w)
    g_f, left_f, _ = _pair(kind, 2, n, h, w, c0, c1, c2, True)
    g_s, left_s, _ = _pair(kind, 2, n, h, w, c0, c1, c2, False)
    assert left_f == 0 and left_s == 0
    for name, a, b in zip(("dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2"), g_f, g_s):
        err = float((a.double() - b.double()).norm() / b.double().norm())
        assert err < 3e-3, (name, err)


@pytest.mark.parametrize("shape", [(1, 8, 16, 64, 128), (2, 16, 16, 256, 256), (1, 64, 16, 64, 64), (3, 8, 48, 128, 64), (2, 40, 20, 64, 128)])
def test_round2_kernels_on_the_shared_pack(shape):
    """Planes too small for the persistent kernel keep the 32x32x16 kernels (conv_igemm_bf16.hip), which now read the SAME
    channel-permuted pack through conv_bf16s_pos: all four tile configurations."""
    import torch.nn.functional as F
    from boxsegliver_amd import _abi, ops
    n, h, w, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(n + h + cin)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16()
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    wp_f, wp_d = ops.conv3x3_pack(wt, bf16=_abi.BF16S)
    y, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True, bf16=_abi.BF16S)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), _r(wt).permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    _stored_ok(y, ref)
    assert (stats.double()[0].sum(0) - ref.sum((0, 1, 2))).abs().max().item() < 3e-5 * ref.abs().sum((0, 1, 2)).max().item()
    dy = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16()
    dref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), _r(wt).permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    _stored_ok(ops.conv3x3_dgrad(dy, wp_d, cin, bf16=_abi.BF16S), dref)
