"""Kernel-level error of the small-plane conv under the current UNETK_LIN_2D against float64 on the device (identical operands):
python tools/dbg/lin2d_check.py  -- prints max |err| / max |ref| of forward (+ statistic sums) and input gradient per shape."""
import math, sys, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from boxsegliver_amd import ops


def check(n, h, w, cin, cout):
    g = torch.Generator(device="cuda").manual_seed(1)
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    w64 = wt.double().permute(3, 2, 0, 1)
    wp_f, wp_d = ops.conv3x3_pack(wt)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g)
    dy = torch.randn((n, h, w, cout), device="cuda", generator=g)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w64, padding=1).permute(0, 2, 3, 1)
    got, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True)
    refd = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w64, padding=1).permute(0, 2, 3, 1)
    gotd = ops.conv3x3_dgrad(dy, wp_d, cin)
    ef = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    ed = ((gotd.double() - refd).abs().max() / refd.abs().max()).item()
    s = stats.double().reshape(2, -1, cout).sum(1)
    es = ((s[0] - ref.sum((0, 1, 2))).abs().max() / ref.abs().sum((0, 1, 2)).max()).item()
    eq = ((s[1] - (ref ** 2).sum((0, 1, 2))).abs().max() / (ref ** 2).sum((0, 1, 2)).max()).item()
    print((n, h, w, cin, cout), "fwd %.2e  dgrad %.2e  stat sum %.2e  stat sq %.2e  rows %d" % (ef, ed, es, eq, rows))


for shp in [(2, 16, 16, 512, 1024), (2, 16, 16, 1024, 1024), (8, 16, 16, 1024, 1024), (2, 16, 16, 256, 128), (3, 16, 16, 128, 256),
            (8, 16, 16, 1024, 512), (1, 16, 16, 64, 128)]:
    check(*shp)
