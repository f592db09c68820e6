"""Context number (not the product, not the CPU baseline): the oracle's plain-PyTorch restatement of the UNet step run
ON THE SAME MI355X in fp32 through PyTorch-ROCm's stock kernels (MIOpen convolutions, eager autograd), forward +
backward only (no optimiser), at BASELINE.json configs[1]'s shape.  Usage: python tools/torch_baseline.py [bs] [size]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    import test_gpu_unet as t
    args = t.make_args(batch_size=bs, im_height=size, im_width=size)
    images, labels = t.synth(bs, size, size, 3)
    net, params = t.oracle_for(args)
    p = {k: v.cuda() for k, v in params.items()}
    x, y = torch.from_numpy(images).cuda(), torch.from_numpy(labels).long().cuda()
    kw = t.loss_kwargs(args)
    for _ in range(3):
        net.loss_and_grads(p, x, y, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        net.loss_and_grads(p, x, y, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print('{"torch_rocm_eager_fp32_fwd_bwd": {"ms_per_step": %.2f, "slices_per_s": %.1f, "bs": %d, "size": %d}}' %
          (ms, bs * 1e3 / ms, bs, size))


if __name__ == "__main__":
    main()
