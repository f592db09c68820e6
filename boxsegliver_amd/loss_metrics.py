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


def _metric_of(inter, left, right, union, met):
    eps = 1e-5
    met = met.lower()
    if met == "dice":
        return (2 * inter + eps) / (left + right + eps)
    if met == "voe":
        return 100 * (1.0 - inter / (union + eps))
    if met == "vd":
        return 100 * ((left - right).abs() / (right + eps))
    raise ValueError("Not supported metric: " + met)


def metric_from_sums(result, n, ncls, cls, met):
    """loss_metrics.py:261-339 with reduce=True, from the head kernel's per-sample sums."""
    s = _class_sums(result, n, ncls, cls)
    return _metric_of(s[:, 0], s[:, 1], s[:, 2], s[:, 3], met).mean()


def metrics_from_sums(result, n, ncls, met):
    """The same metric for EVERY foreground class at once: a [ncls - 1] tensor whose element i - 1 is class i's value (the
    per-class calls cost six tiny launches per class and step; the elementwise arithmetic is identical)."""
    body = result[3:3 + n * (ncls - 1) * 4].view(n, ncls - 1, 4)
    return _metric_of(body[..., 0], body[..., 1], body[..., 2], body[..., 3], met).mean(0)


# --------------------------------------------------------------------------- volume metrics (host, per case)
def metric_3d(logits3d, labels3d, required=None, **kwargs):
    """loss_metrics.py:342-452: Dice / VOE / RVD (medpy's `dc`, `1 - jc`, `|ravd|`) and ASSD / RMSD / MSD
    (utils/surface.py) of a binary 3-D prediction against its label, on the host as in the reference.

    medpy 0.4 semantics restated: dc = 2|A&B| / (|A| + |B|), 0.0 when both are empty; jc = |A&B| / |A|B|
    (0/0 -> nan, like medpy's float division); ravd = (|A| - |B|) / |B|, RuntimeError on an empty reference.
    ASSD and MSD are 0 when either object is empty (:419-421); the reference leaves RMSD unset in that case."""
    import numpy as np
    metrics = ["Dice", "VOE", "RVD", "ASSD", "RMSD", "MSD"]
    if required is None:
        required = list(metrics)
    elif isinstance(required, str):
        required = [required]
    else:
        required = list(required)
    for req in required:
        if req not in metrics:
            raise ValueError("Not supported metric: %s" % req)
    need_dist_map = any(req in metrics[3:] for req in required)
    logits3d, labels3d = np.asarray(logits3d), np.asarray(labels3d)
    if logits3d.ndim > 3:
        logits3d = np.squeeze(logits3d)
    if labels3d.ndim > 3:
        labels3d = np.squeeze(labels3d)
    assert logits3d.shape == labels3d.shape, ("Shape mismatch of logits3D and labels3D. \n"
                                              "Logits3D has shape %r while labels3D has "
                                              "shape %r" % (logits3d.shape, labels3d.shape))
    a, b = logits3d.astype(bool), labels3d.astype(bool)
    out = {}
    sampling = kwargs.get("sampling", [1., 1., 1.])
    if need_dist_map:
        from .utils.surface import Surface
        if np.count_nonzero(a) == 0 or np.count_nonzero(b) == 0:
            out["ASSD"] = 0
            out["MSD"] = 0
        else:
            surf = Surface(a, b, physical_voxel_spacing=sampling, mask_offset=[0., 0., 0.], reference_offset=[0., 0., 0.])
            if "ASSD" in required:
                out["ASSD"] = surf.get_average_symmetric_surface_distance()
            if "MSD" in required:
                out["MSD"] = surf.get_maximum_symmetric_surface_distance()
            if "RMSD" in required:
                out["RMSD"] = surf.get_root_mean_square_symmetric_surface_distance()
    inter = int(np.count_nonzero(a & b))
    na, nb = int(np.count_nonzero(a)), int(np.count_nonzero(b))
    if "Dice" in required:
        out["Dice"] = 2.0 * inter / float(na + nb) if na + nb > 0 else 0.0
    if "VOE" in required:
        union = int(np.count_nonzero(a | b))
        out["VOE"] = 1.0 - (inter / float(union) if union > 0 else float("nan"))
    if "RVD" in required:
        if nb == 0:
            raise RuntimeError("The second supplied array does not contain any binary object.")
        out["RVD"] = abs((na - nb) / float(nb))
    return out


class ConfusionMatrix(object):
    """loss_metrics.py:506-580: tp / fp / tn / fn counts of a test volume against a reference volume."""

    def __init__(self, test=None, reference=None):
        self.test, self.reference = test, reference
        self.reset()

    def set_test(self, test):
        self.test = test
        self.reset()

    def set_reference(self, reference):
        self.reference = reference
        self.reset()

    def reset(self):
        self.tp = self.fp = self.tn = self.fn = self.size = None
        self.test_empty = self.test_full = self.reference_empty = self.reference_full = None

    def compute(self):
        import numpy as np
        if self.test is None or self.reference is None:
            raise ValueError("'test' and 'reference' must both be set to compute confusion matrix.")
        assert self.test.shape == self.reference.shape, "Shape mismatch: {} and {}".format(
            self.test.shape, self.reference.shape)
        t, r = self.test != 0, self.reference != 0
        self.tp = int((t & r).sum())
        self.fp = int((t & ~r).sum())
        self.tn = int((~t & ~r).sum())
        self.fn = int((~t & r).sum())
        self.size = self.reference.size
        self.test_empty, self.test_full = not np.any(self.test), bool(np.all(self.test))
        self.reference_empty, self.reference_full = not np.any(self.reference), bool(np.all(self.reference))

    def get_matrix(self):
        if None in (self.tp, self.fp, self.tn, self.fn):
            self.compute()
        return self.tp, self.fp, self.tn, self.fn

    def get_size(self):
        if self.size is None:
            self.compute()
        return self.size

    def get_existence(self):
        if None in (self.test_empty, self.test_full, self.reference_empty, self.reference_full):
            self.compute()
        return self.test_empty, self.test_full, self.reference_empty, self.reference_full
