"""Loss head + in-graph metrics of the reference, restated on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/loss_metrics.py:
  _compute_weights                         :115-165
  weighted_sparse_softmax_cross_entropy    :172-177
  sparse_dice_loss / weighted_dice_loss    :180-231
  metric_dice / metric_voe / metric_vd     :261-339
and the TF-1.13 tf.losses semantics they call into:
  tf.losses.sparse_softmax_cross_entropy(labels, logits, weights) with
  reduction SUM_BY_NONZERO_WEIGHTS = sum(ce * w) / count(w != 0) after
  broadcasting w to ce's shape.
"""
import torch
import torch.nn.functional as F


def compute_weights(w_type, labels, num_classes, numeric_w=None, proportion_decay=None):
    """loss_metrics.py:115-165.  labels: int64 [bs, *spatial].  Returns a float
    tensor [bs, *spatial] or the python scalar 1.0 for w_type 'none'."""
    w_type = w_type.lower()
    if w_type == "none":
        return 1.0                                           # :123-124
    one_hot = F.one_hot(labels, num_classes).to(torch.float32)
    sp_axes = tuple(range(1, labels.dim()))
    if w_type == "numerical":                                # :125-133
        if numeric_w is None:
            raise KeyError("w_type `numerical` need keyword argument `numeric_w`")
        nw = torch.tensor(numeric_w, dtype=torch.float32, device=labels.device)
        w = (one_hot * nw).sum(-1)
    elif w_type == "proportion":                             # :134-143
        num_labels = one_hot.sum(dim=sp_axes)                # [bs, ncls]
        if proportion_decay is not None:
            num_labels = num_labels + proportion_decay
        proportions = 1.0 / num_labels
        pw = proportions / proportions.sum(dim=1, keepdim=True)
        shape = (labels.shape[0],) + (1,) * (labels.dim() - 1) + (num_classes,)
        w = (one_hot * pw.reshape(shape)).sum(-1)
    elif w_type == "boundary":                               # :149-159 (4-D one-hot only)
        import numpy as np
        from scipy.ndimage import distance_transform_edt
        if labels.dim() != 3:
            raise ValueError("boundary weights are 2-D only in the reference (loss_metrics.py:150-151)")
        # per class: clip(conv3x3_SAME(onehot), 0, 1) - onehot; summed over classes; EDT of its complement
        oh = one_hot.permute(0, 3, 1, 2).reshape(-1, 1, labels.shape[1], labels.shape[2])
        dil = F.conv2d(oh, torch.ones(1, 1, 3, 3), padding=1).clamp(0, 1) - oh
        ring = dil.reshape(labels.shape[0], num_classes, labels.shape[1], labels.shape[2]).sum(1) > 0
        dist = np.stack([distance_transform_edt(~r.numpy()).astype(np.float32) for r in ring])
        w = torch.exp(-torch.from_numpy(dist) / 25) + 1
    else:
        raise ValueError("Not supported weight type: " + w_type)
    # :163-165 per-sample renormalisation to mean 1
    size = 1
    for a in sp_axes:
        size *= labels.shape[a]
    w = w / w.sum(dim=sp_axes, keepdim=True) * float(size)
    return w


def weighted_sparse_softmax_cross_entropy(logits, labels, w_type="none", **kw):
    """loss_metrics.py:172-177 + TF SUM_BY_NONZERO_WEIGHTS."""
    ncls = logits.shape[-1]
    w = compute_weights(w_type, labels, ncls, **kw)
    ce = F.cross_entropy(logits.reshape(-1, ncls), labels.reshape(-1), reduction="none")
    ce = ce.reshape(labels.shape)
    if isinstance(w, float):
        num_present = float(ce.numel()) if w != 0.0 else 0.0
        total = (ce * w).sum()
    else:
        num_present = float((w != 0).sum().item())
        total = (ce * w).sum()
    if num_present == 0:
        return total * 0.0
    return total / num_present


def sparse_dice_loss(probs, labels, eps=1e-8):
    """loss_metrics.py:180-226 (with_bg=False): 1 - mean_b(2 I_b / (U_b + eps))."""
    ncls = probs.shape[-1]
    # the reference casts to float32 (:208-209); float64 is kept when the oracle is run in fp64 as
    # the high-precision yardstick of the parity tests
    dt = torch.float64 if probs.dtype == torch.float64 else torch.float32
    one_hot = F.one_hot(labels, ncls).to(dt)[..., 1:]
    p = probs.to(dt)[..., 1:]
    axes = tuple(range(1, probs.dim()))
    inter = (one_hot * p).sum(dim=axes)
    union = (one_hot + p).sum(dim=axes)
    return 1.0 - ((2.0 * inter) / (union + eps)).mean()


def threshold_pred(probs):
    """UNet.py:112-118: per foreground class, uint8 (prob > 0.5), shape [bs,H,W,1]."""
    return [(probs[..., i:i + 1] > 0.5).to(torch.uint8) for i in range(1, probs.shape[-1])]


def metric_dice(pred, label, eps=1e-5):
    """loss_metrics.py:261-301, reduce=True."""
    axes = tuple(range(1, pred.dim()))
    pred = pred.to(torch.float32)
    label = label.to(torch.float32)
    inter = (pred * label).sum(dim=axes)
    left = pred.sum(dim=axes)
    right = label.sum(dim=axes)
    return ((2 * inter + eps) / (left + right + eps)).mean()


def metric_voe(pred, label, eps=1e-5):
    """loss_metrics.py:304-320."""
    axes = tuple(range(1, pred.dim()))
    pred = pred.to(torch.float32)
    label = label.to(torch.float32)
    num = (pred * label).sum(dim=axes)
    den = torch.clamp(pred + label, 0.0, 1.0).sum(dim=axes)
    return (100 * (1.0 - num / (den + eps))).mean()


def metric_vd(pred, label, eps=1e-5):
    """loss_metrics.py:323-339."""
    axes = tuple(range(1, pred.dim()))
    pred = pred.to(torch.float32)
    label = label.to(torch.float32)
    a = pred.sum(dim=axes)
    b = label.sum(dim=axes)
    return (100 * ((a - b).abs() / (b + eps))).mean()


METRICS = {"Dice": metric_dice, "VOE": metric_voe, "VD": metric_vd}
