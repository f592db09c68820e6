"""Run the same full-size step twice and report which gradient tensors are not bit-identical."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_unet as t
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    args = t.make_args(batch_size=bs, im_height=size, im_width=size)
    images, labels = t.synth(bs, size, size, 3)
    model, inputs = t.build(args, images, labels)
    runs = []
    for _ in range(3):
        model.params.zero_grad()
        loss = model(inputs, "train", **t.YML)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.item(), {n: model.params[n].grad.clone() for n in model.params.trainable_names()}))
    print("losses", [r[0] for r in runs])
    for i in (1, 2):
        bad = [(n, (runs[0][1][n] - runs[i][1][n]).abs().max().item(), runs[0][1][n].abs().max().item())
               for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[i][1][n])]
        print("run 0 vs", i, ":", len(bad), "tensors differ")
        for n, d, m in bad[:40]:
            print("   %-60s max|d| %.3e  max|g| %.3e" % (n, d, m))


if __name__ == "__main__":
    main()
