"""deconv2x2_bwd run-to-run reproducibility at a multi-tile-per-split size; prints where dw differs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from boxsegliver_amd import ops
    n, h, w, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 128, 128, 128, 64))]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g)
    wt = torch.randn((2, 2, cout, cin), device="cuda", generator=g) / cin ** 0.5
    b = torch.randn((cout,), device="cuda", generator=g) * 0.1
    cat = torch.zeros((n, 2 * h, 2 * w, 2 * cout), device="cuda")
    dcat = torch.randn(cat.shape, device="cuda", generator=g)
    wp_f, wp_d = ops.deconv2x2_pack(wt)
    ops.deconv2x2_fwd(x, wp_f, b, cat, cout, cout)
    outs = []
    for _ in range(4):
        dx, dw, db = ops.deconv2x2_bwd(x, wp_d, cat, dcat, cout, cout)
        torch.cuda.synchronize()
        outs.append((dx.clone(), dw.clone(), db.clone()))
    for i in range(1, 4):
        print("run", i, "dx equal", torch.equal(outs[0][0], outs[i][0]), "db equal", torch.equal(outs[0][2], outs[i][2]),
              "dw equal", torch.equal(outs[0][1], outs[i][1]))
        d = (outs[0][1] - outs[i][1]).abs()
        idx = torch.nonzero(d > 0)
        print("   differing dw elements:", idx.shape[0], "of", d.numel(), "max", d.max().item(), "ref max", outs[0][1].abs().max().item())
        if idx.shape[0]:
            print("   co values:", sorted(set(idx[:, 2].tolist())), " ci count:", len(set(idx[:, 3].tolist())))
            print("   first", idx[:6].tolist(), " per (a,b):", [int((d[a, bb] > 0).sum()) for a in range(2) for bb in range(2)])
    # correctness of run 0 against a float64 evaluation on the GPU
    mask = (cat[..., cout:] > 0).double()
    dpre = dcat[..., cout:].double() * mask
    x64 = x.double()
    ref = torch.empty((2, 2, cout, cin), dtype=torch.float64, device="cuda")
    for a in range(2):
        for bb in range(2):
            ref[a, bb] = torch.einsum("nhwo,nhwi->oi", dpre[:, a::2, bb::2], x64)
    print("dw rel err vs fp64:", ((outs[0][1].double() - ref).abs().max() / ref.abs().max()).item())


if __name__ == "__main__":
    main()
