"""Trained-model parity: the reference's training loop on the oracle, for comparison with the HIP path AFTER many optimiser steps.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).  Used by tests/test_gpu_train_dice.py and by bench.py's
`cpu_baseline` leg (its `dice_vs_oracle` field); never by the product.

What is restated: one iteration of core/estimator.py:523-560 for the UNet of NetworksV2/UNet.py:58-155 -- forward with batch
statistics, total loss = data loss + L2 (UNet.py:97-135), gradients, tf.train.AdamOptimizer(beta2 = 0.99) (core/solver.py:204-211),
moving statistics carried along (base.py:153-169) -- then the in-graph metrics `<class>/Dice` of the thresholded predictions
(loss_metrics.py:261-301) on held-out batches, and the Dice of the held-out slices stacked into one volume the way the volume
evaluator accumulates a case (evaluators/evaluator_liver.py:936-962: confusion counts summed over the case).

The oracle is a torch restatement, so it runs in float64 ON THE DEVICE when one is there (a few seconds for 100+ steps at
64 x 64); on the CPU the same code is the reference point, only slower."""
import math

import numpy as np
import torch


def stream(n, bs, size, seed):
    """Learnable LiTS-like batches: liver = an ellipse (brighter), tumor = a disk inside it (darker), random position / size per
    slice, three adjacent-slice-like channels (same anatomy, independent noise).  [(images f32 [bs,H,W,3], labels i32 [bs,H,W])]"""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    out = []
    for _ in range(n):
        img = np.zeros((bs, size, size, 3), np.float32)
        lab = np.zeros((bs, size, size), np.int32)
        for b in range(bs):
            cy, cx = (0.5 + rng.uniform(-0.15, 0.15, 2)) * size
            ry, rx = rng.uniform(0.18, 0.3) * size, rng.uniform(0.15, 0.28) * size
            ell = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            ty, tx = cy + rng.uniform(-0.4, 0.4) * ry, cx + rng.uniform(-0.4, 0.4) * rx
            tr = rng.uniform(0.06, 0.11) * size
            disk = ((yy - ty) ** 2 + (xx - tx) ** 2 <= tr ** 2) & ell
            lab[b][ell] = 1
            lab[b][disk] = 2
            base = 0.30 + 0.25 * ell - 0.22 * disk
            img[b] = base[..., None] + rng.normal(0.0, 0.08, (size, size, 3)).astype(np.float32)
        out.append((img, lab))
    return out


class TFAdamTorch(object):
    """oracle/solver.TFAdam on torch tensors (any device / dtype): tf.train.AdamOptimizer, epsilon OUTSIDE the bias correction."""

    def __init__(self, beta1=0.9, beta2=0.99, eps=1e-8):
        self.b1, self.b2, self.eps, self.t = beta1, beta2, eps, 0
        self.m, self.v = {}, {}

    def step(self, params, grads, lr):
        self.t += 1
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = torch.zeros_like(params[k])
                self.v[k] = torch.zeros_like(params[k])
            self.m[k] += (1 - self.b1) * (g - self.m[k])
            self.v[k] += (1 - self.b2) * (g * g - self.v[k])
            params[k] -= lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)


def train(net, params, batches, steps, lr, loss_kw, device="cpu", dtype=torch.float64):
    """`steps` iterations from `params` (not modified); returns (trained params on `device`, loss curve)."""
    p = {k: v.detach().to(device=device, dtype=dtype).clone() for k, v in params.items()}
    data = [(torch.from_numpy(i).to(device=device, dtype=dtype), torch.from_numpy(l).to(device=device).long()) for i, l in batches]
    opt, curve = TFAdamTorch(0.9, 0.99, 1e-8), []
    for s in range(steps):
        img, lab = data[s % len(data)]
        total, _, _, grads, new_stats = net.loss_and_grads(p, img, lab, **loss_kw)
        curve.append(float(total))
        opt.step(p, grads, lr)
        for k, v in new_stats.items():
            p[k] = v.detach()
    return p, curve


def confusion_dice(pred, label):
    """Dice from summed confusion counts (evaluator_liver.py:936-962): 2 tp / (2 tp + fn + fp); bool / {0,1} arrays of one case."""
    pred, label = np.asarray(pred).astype(bool), np.asarray(label).astype(bool)
    tp = int(np.count_nonzero(pred & label))
    fp = int(np.count_nonzero(pred & ~label))
    fn = int(np.count_nonzero(~pred & label))
    den = 2 * tp + fn + fp
    return 2.0 * tp / den if den else 0.0


def stream3d(n, bs, depth, size, seed):
    """3-D analogue for UNet3D (one image channel, two classes, UNet3D.py:31-91): an ellipsoid (brighter) at a random place in a
    noisy patch.  [(images f32 [bs,D,H,W,1], labels i32 [bs,D,H,W])]"""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(np.arange(depth, dtype=np.float32), np.arange(size, dtype=np.float32),
                             np.arange(size, dtype=np.float32), indexing="ij")
    out = []
    for _ in range(n):
        img = np.zeros((bs, depth, size, size, 1), np.float32)
        lab = np.zeros((bs, depth, size, size), np.int32)
        for b in range(bs):
            cz = (0.5 + rng.uniform(-0.2, 0.2)) * depth
            cy, cx = (0.5 + rng.uniform(-0.2, 0.2, 2)) * size
            rz, ry, rx = rng.uniform(0.25, 0.45) * depth, rng.uniform(0.15, 0.3) * size, rng.uniform(0.15, 0.3) * size
            ell = ((zz - cz) / rz) ** 2 + ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            lab[b][ell] = 1
            img[b, ..., 0] = 0.2 + 0.35 * ell + rng.normal(0.0, 0.08, ell.shape).astype(np.float32)
        out.append((img, lab))
    return out


def heldout(net, p, held, classes, device="cpu", dtype=torch.float64):
    """Forward with BATCH statistics on held-out batches (slim's moving averages, decay 0.999, are still near their initial values
    after a few hundred steps, in the reference as here): mean in-graph `<class>/Dice` (loss_metrics.py:261-301 on the thresholded
    predictions, UNet.py:112-118 / UNet3D.py:171-177), the thresholded predictions stacked into one volume per class (uint8, the
    labels' shape) and the logits.  Works for the 2-D and the 3-D oracle alike (both: forward(p, images, is_training))."""
    from . import losses
    dice = {c + "/Dice": [] for c in classes[1:]}
    vols = {c: [] for c in classes[1:]}
    logits_all = []
    with torch.no_grad():
        for img, lab in held:
            x = torch.from_numpy(img).to(device=device, dtype=dtype)
            y = torch.from_numpy(lab).to(device=device).long()
            logits = net.forward(p, x, True)[0]
            preds = losses.threshold_pred(torch.softmax(logits, -1))
            for i, c in enumerate(classes[1:], start=1):
                dice[c + "/Dice"].append(float(losses.metric_dice(preds[i - 1], (y == i).unsqueeze(-1))))
                vols[c].append(preds[i - 1].reshape(lab.shape).cpu().numpy())
            logits_all.append(logits.double().cpu().numpy())
    return ({k: float(np.mean(v)) for k, v in dice.items()}, {c: np.concatenate(v) for c, v in vols.items()},
            np.concatenate(logits_all))
