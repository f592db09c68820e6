import math, sys, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from boxsegliver_amd import _abi, ops

def r(t): return t.float().bfloat16().double()

def check(n, h, w, cin, cout, dgrad):
    g = torch.Generator(device="cuda").manual_seed(1)
    wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)
    w64 = r(wt).permute(3, 2, 0, 1)
    wp_f, wp_d = ops.conv3x3_pack(wt, bf16=_abi.BF16S)
    if dgrad:
        dy = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16()
        ref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w64, padding=1).permute(0, 2, 3, 1)
        got = ops.conv3x3_dgrad(dy, wp_d, cin, bf16=_abi.BF16S)
    else:
        x = torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w64, padding=1).permute(0, 2, 3, 1)
        got, stats, rows = ops.conv3x3_fwd(x, wp_f, cout, want_stats=True, bf16=_abi.BF16S)
    err = (got.double() - ref).abs()
    bad = err > 0.05
    print((n, h, w, cin, cout), 'dgrad' if dgrad else 'fwd', 'max err', err.max().item(), 'bad share', bad.double().mean().item())
    if bad.any():
        idx = bad.nonzero()
        print('  first bad', idx[:5].tolist(), ' last bad', idx[-3:].tolist())
        # pattern: by tile row (h % 32 // 4 = wave), h%4 (tm), w%16//4 (kq), w%4 (r), channel
        for name, key in (('n', idx[:,0]), ('tile_h', idx[:,1]//32), ('wave', idx[:,1]%32//4), ('tm', idx[:,1]%4), ('tile_w', idx[:,2]//16), ('kq', idx[:,2]%16//4), ('r', idx[:,2]%4), ('c', idx[:,3])):
            u, cnt = torch.unique(key, return_counts=True)
            print('  ', name, dict(zip(u.tolist()[:20], cnt.tolist()[:20])))
    if not dgrad:
        s = stats.double()
        e0 = (s[0].sum(0) - ref.sum((0,1,2))).abs().max().item()
        print('  stat sum err', e0)

check(8, 128, 128, 64, 128, True)
check(8, 256, 128, 64, 64, False)
check(8, 256, 128, 64, 64, True)
check(8, 256, 128, 128, 64, False)
