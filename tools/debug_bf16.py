"""How close is the HIP UNETK_BF16 step to (a) the fp32 oracle, (b) the oracle restating the same bf16 arithmetic?
Prints loss / logits / gradient distances (run on the GPU box: python tools/debug_bf16.py [size])."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import test_gpu_unet as t
    without_norm = len(sys.argv) > 2 and sys.argv[2] == "nonorm"
    args = t.make_args(im_height=size, im_width=size, compute_dtype="bf16", without_norm=without_norm)
    images, labels = t.synth(2, size, size, 3)
    model, inputs = t.build(args, images, labels)
    if without_norm:
        from oracle import unet2d
        net = unet2d.UNet2DOracle(3, 3, without_norm=True)
        params = unet2d.init_params(net.specs, seed=77)
        g = torch.Generator().manual_seed(5)
        for name, _, kind in net.specs:
            if kind == "bias":
                params[name] = 0.1 * torch.randn(params[name].shape, generator=g)
    else:
        net, params = t.oracle_for(args)
    model.params.load_state(params)
    model.params.zero_grad()
    loss = model(inputs, "train", **t.YML)
    loss.backward()
    torch.cuda.synchronize()
    got = model.layers["logits"].cpu().numpy()
    p64 = {k: v.double() for k, v in params.items()}
    for label, bf in (("fp32-arithmetic oracle", False), ("bf16-arithmetic oracle", True)):
        net.bf16 = bf
        total, _, logits, grads, _ = net.loss_and_grads(p64, torch.from_numpy(images).double(),
                                                        torch.from_numpy(labels).long(), **t.loss_kwargs(args))
        d = np.abs(got - logits.numpy())
        num = den = 0.0
        worst = ("", 0.0)
        for name in model.params.trainable_names():
            g = model.params[name].grad.cpu().numpy().astype(np.float64)
            r = grads[name].numpy()
            num += np.sum((g - r) ** 2)
            den += np.sum(r ** 2)
            l2 = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
            if l2 > worst[1]:
                worst = (name, l2)
        print("{}: loss {:.6f} vs {:.6f} (rel {:.2e}); logits max|d| {:.3e} mean|d| {:.3e} range {:.2f}; "
              "argmax agree {:.5f}; grad L2 {:.3e}; worst tensor {} {:.3e}".format(
                  label, loss.item(), total.item(), abs(loss.item() - total.item()) / abs(total.item()), d.max(), d.mean(),
                  logits.numpy().max() - logits.numpy().min(), (got.argmax(-1) == logits.numpy().argmax(-1)).mean(),
                  (num / den) ** 0.5, worst[0], worst[1]))


if __name__ == "__main__":
    main()
