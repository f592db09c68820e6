"""Loss / metric flag surface and host glue -- mirror of the reference's loss_metrics.py.

The arithmetic of loss_metrics.py:115-339 (class-weight maps, weighted sparse softmax xent with
SUM_BY_NONZERO_WEIGHTS, soft Dice loss, Dice/VOE/VD on thresholded predictions) runs inside the
fused HIP head kernel (csrc/head.hip); this module only keeps the argparse group
(loss_metrics.py:26-67, flag names and defaults verbatim) and turns the kernel's per-sample sums
into the reference's scalars.
"""
import torch

from . import ops

METRICS = "metrics"


def add_arguments(parser):
    group = parser.add_argument_group(title="Loss Arguments")
    group.add_argument("--weight_decay_rate", type=float, default=1e-5, required=False,
                       help="Weight decay rate for variable regularizers (default: %(default)f)")
    group.add_argument("--bias_decay", action="store_true", required=False, help="Use bias decay or not")
    group.add_argument("--loss_type", type=str, default="xentropy",
                       choices=["xentropy", "dice", "xentropy+dice"], required=False,
                       help="Loss type (default %(default)s)")
    group.add_argument("--loss_weight_type", type=str, default="none",
                       choices=["none", "numerical", "proportion", "boundary"], required=False,
                       help="Weights used in loss function for alleviating class imbalance problem "
                            "(default %(default)s)")
    group.add_argument("--loss_numeric_w", type=float, nargs="+", required=False,
                       help="Numeric weights for loss_weight_type=\"numerical\". Notice that one value"
                            "for one class")
    group.add_argument("--loss_proportion_decay", type=float, default=1000, required=False,
                       help="Proportion decay for loss_weight_type=\"proportion\". Check source code"
                            "for details. (default: %(default)f)")
    group.add_argument("--metrics_train", type=str, default=["Dice"], choices=["Dice", "VOE", "VD"], nargs="+",
                       required=False, help="Evaluation metric names (default: %(default)s)")
    group.add_argument("--metrics_eval", type=str, default=["Dice"],
                       choices=["Dice", "VOE", "RVD", "ASSD", "RMSD", "MSD"], nargs="+", required=False,
                       help="Evaluation metric names (default: %(default)s)")


def build_head_desc(args, n, hw, c, ncls, explicit_map=False):
    """Translate --loss_weight_type & friends (loss_metrics.py:115-165) into the kernel descriptor."""
    w_type = (getattr(args, "loss_weight_type", "none") or "none").lower()
    if explicit_map:
        return ops.head_desc(n, hw, c, ncls, "pixelmap")
    if w_type == "none":
        return ops.head_desc(n, hw, c, ncls, "none")
    if w_type == "numerical":
        nw = getattr(args, "loss_numeric_w", None)
        if not nw:
            raise KeyError("w_type `numerical` need keyword argument `numeric_w`")
        return ops.head_desc(n, hw, c, ncls, "numerical", numeric_w=nw)
    if w_type == "proportion":
        decay = getattr(args, "loss_proportion_decay", 0) or 0
        return ops.head_desc(n, hw, c, ncls, "proportion", proportion_decay=decay if decay > 0 else 0.0)
    if w_type == "boundary":
        # resolved by pixel_weights() below into an explicit map (the reference round-trips to scipy on the host,
        # loss_metrics.py:149-159); reaching this line means 3-D labels, which the reference cannot handle either
        raise ValueError("loss_weight_type `boundary` is defined for 2-D labels only")
    raise ValueError("Not supported weight type: " + w_type)


def pixel_weights(args, inputs, labels):
    """Explicit per-pixel loss weights: a caller-provided inputs["pixel_weights"] map, or the `boundary` map
    (loss_metrics.py:149-165) computed on the device from the labels; None for the table-driven weight types."""
    explicit = inputs.get("pixel_weights")
    if explicit is not None or labels is None:
        return explicit
    w_type = (getattr(args, "loss_weight_type", "none") or "none").lower()
    if w_type == "boundary" and labels.dim() == 3:
        return ops.boundary_weights(labels)
    return None


def _class_sums(result, n, ncls, cls):
    """[n, 4] view of (sum pred*lab, sum pred, sum lab, sum clip(pred+lab)) for class `cls`."""
    body = result[3:3 + n * (ncls - 1) * 4].view(n, ncls - 1, 4)
    return body[:, cls - 1, :]


def metric_from_sums(result, n, ncls, cls, met):
    """loss_metrics.py:261-339 with reduce=True, from the head kernel's per-sample sums."""
    s = _class_sums(result, n, ncls, cls)
    inter, left, right, union = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    eps = 1e-5
    met = met.lower()
    if met == "dice":
        return ((2 * inter + eps) / (left + right + eps)).mean()
    if met == "voe":
        return (100 * (1.0 - inter / (union + eps))).mean()
    if met == "vd":
        return (100 * ((left - right).abs() / (right + eps))).mean()
    raise ValueError("Not supported metric: " + met)
