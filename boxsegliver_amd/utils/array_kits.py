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
