"""Array helpers the volume evaluator needs -- host-side mirror of the reference's utils/array_kits.py
(`get_largest_component` :357-384, `merge_labels`, `bbox_to_shape`)."""
import numpy as np
from scipy import ndimage as ndi


def merge_labels(masks, merges):
    """utils/array_kits.py `merge_labels`: out = index (1-based) of the group of `merges` a label falls in; a group
    is an int, a list of ints, or -1 / [-1, ...] which collects "everything else" into 0."""
    out = np.zeros_like(masks, dtype=np.uint8)
    for i, group in enumerate(merges):
        if isinstance(group, (int, np.integer)):
            group = [int(group)]
        elif not isinstance(group, (list, tuple)):
            raise ValueError("Only integer or list is accepted, but got {}(type {}) in merges[{}]"
                             .format(group, type(group), i))
        for lab in group:
            out[masks == lab] = i
    return out


def get_largest_component(inputs, rank, connectivity=1):
    """Largest connected component (int8 0/1 array); a zero array stays zero (utils/array_kits.py:357-384).
    Ties resolve like np.argsort: the component with the larger label id."""
    struct = ndi.generate_binary_structure(rank, connectivity)
    res = np.asarray(inputs).astype(bool)
    if np.count_nonzero(res) == 0:
        return np.zeros_like(inputs, dtype=np.int8)
    labeled, _ = ndi.label(res, struct)
    areas = np.bincount(labeled.flat)[1:]
    order = np.argsort(areas)
    return merge_labels(labeled, [-1, int(order[-1]) + 1])


def bbox_to_shape(bbox):
    """(x1, y1, z1, x2, y2, z2) inclusive -> (d, h, w)  (utils/array_kits.py `bbox_to_shape`)."""
    ndim = len(bbox) // 2
    return tuple(int(bbox[i + ndim]) - int(bbox[i]) + 1 for i in range(ndim))[::-1]


def bbox_to_slices(bbox):
    """(x1, y1, [z1,] x2, y2[, z2]) inclusive -> index slices in array order ([z,] y, x) (utils/array_kits.py:177-195)."""
    if len(bbox) % 2 != 0:
        raise ValueError("`bbox` should have even number of elements, got {}".format(len(bbox)))
    ndim = len(bbox) // 2
    return tuple(slice(int(bbox[i]), int(bbox[i + ndim]) + 1) for i in reversed(range(ndim)))


def bbox_from_mask(mask, mask_values=1):
    """utils/array_kits.py:85-151 (without min_shape / padding): tight box of the voxels whose value is in mask_values,
    (x1, y1, x2, y2) or (x1, y1, z1, x2, y2, z2), both corners INSIDE the object; zeros for an empty mask."""
    mask = np.asarray(mask)
    values = np.atleast_1d(mask_values)
    sel = np.isin(mask, values)
    if not sel.any():
        return np.zeros(shape=(mask.ndim * 2,))
    lo, hi = [], []
    for ax in range(mask.ndim):
        proj = np.where(sel.any(axis=tuple(a for a in range(mask.ndim) if a != ax)))[0]
        lo.append(int(proj[0]))
        hi.append(int(proj[-1]))
    return np.array(lo[::-1] + hi[::-1])


def extract_region(mask):
    """utils/array_kits.py:263-340 with the defaults the exporter uses (align 1, padding 0): box of the non-zero voxels,
    (x1, y1, [z1,] x2, y2[, z2]) inclusive."""
    return bbox_from_mask(np.asarray(mask) != 0, 1)


def compute_robust_moments(binary_image, isotropic=False, indexing="ij", min_std=0.):
    """utils/array_kits.py:387-440: median of the foreground coordinates and 1.4826 x median absolute deviation per
    axis (or of the radial distance when isotropic); (-1, ...) for an empty image."""
    binary_image = np.asarray(binary_image)
    ndim = binary_image.ndim
    points = np.asarray(np.nonzero(binary_image)).astype(np.float32)
    if points.shape[1] == 0:
        return np.array([-1.0] * ndim, dtype=np.float32), np.array([-1.0] * ndim, dtype=np.float32)
    points = np.transpose(points)
    center = np.median(points, axis=0)
    if isotropic:
        mad = np.array([np.median(np.linalg.norm(points - center, axis=1))] * ndim)
    else:
        mad = np.median(np.absolute(points - center), axis=0)
    std_dev = np.maximum(1.4826 * mad, [min_std] * ndim)
    if not indexing or indexing == "xy":
        return center[::-1], std_dev[::-1]
    if indexing == "ij":
        return center, std_dev
    raise ValueError("Valid values for `indexing` are 'xy' and 'ij'.")
