"""Surface distance metrics between two binary 3-D objects -- host-side mirror of the reference's utils/surface.py
(`Surface`, used by loss_metrics.metric_3d :417-433 for ASSD / RMSD / MSD).

Same definitions (Heimann et al. 2009): the surface S(A) of an object is every object voxel with a background voxel
in its 18-neighbourhood (`A xor erode(A, 18-connectivity)`, the volume border counting as background,
utils/surface.py:255-285); d(v, S) = min over s in S of the Euclidean distance in millimetres (voxel index x spacing).

Restated, not copied: the reference finds nearest neighbours with a k-d tree over the surface point lists
(:226-249); here each direction is ONE exact Euclidean distance transform of the other surface's complement with the
voxel spacing as sampling -- the same minimum over the same point set, linear in the volume size.
"""
import math

import numpy as np
from scipy import ndimage as ndi


class Surface(object):
    def __init__(self, mask, reference, physical_voxel_spacing=(1, 1, 1), mask_offset=(0, 0, 0),
                 reference_offset=(0, 0, 0)):
        if tuple(mask_offset) != (0, 0, 0) or tuple(reference_offset) != (0, 0, 0):
            raise NotImplementedError("offsets are always zero at the reference's call site (loss_metrics.py:420-422)")
        mask = np.asarray(mask).astype(bool)
        reference = np.asarray(reference).astype(bool)
        if mask.shape != reference.shape or mask.ndim != 3:
            raise ValueError("two 3-D arrays of the same shape expected")
        self._mask_edge = self.compute_contour(mask)
        self._ref_edge = self.compute_contour(reference)
        if not self._mask_edge.any():
            raise Exception("The mask image does not seem to contain an object.")
        if not self._ref_edge.any():
            raise Exception("The reference image does not seem to contain an object.")
        self._spacing = tuple(float(s) for s in physical_voxel_spacing)
        self._m2r = None     # distance of every MASK surface voxel to the reference surface
        self._r2m = None

    @staticmethod
    def compute_contour(array):
        """Object voxels with background in their 18-neighbourhood (face-and-edge kernel)."""
        array = np.asarray(array).astype(bool)
        footprint = ndi.generate_binary_structure(3, 2)
        return array ^ ndi.binary_erosion(array, footprint)

    def _dist(self, src_edge, dst_edge):
        field = ndi.distance_transform_edt(~dst_edge, sampling=self._spacing)
        return field[src_edge]

    def get_mask_edge_points(self):
        return np.argwhere(self._mask_edge) * np.asarray(self._spacing)

    def get_reference_edge_points(self):
        return np.argwhere(self._ref_edge) * np.asarray(self._spacing)

    def get_reference_mask_nn(self):
        """Distances of the mask surface voxels to the reference surface (utils/surface.py:236-249)."""
        if self._m2r is None:
            self._m2r = self._dist(self._mask_edge, self._ref_edge)
        return self._m2r

    def get_mask_reference_nn(self):
        """Distances of the reference surface voxels to the mask surface (:226-234)."""
        if self._r2m is None:
            self._r2m = self._dist(self._ref_edge, self._mask_edge)
        return self._r2m

    def get_maximum_symmetric_surface_distance(self):
        return max(self.get_mask_reference_nn().max(), self.get_reference_mask_nn().max())

    def get_root_mean_square_symmetric_surface_distance(self):
        a, b = self.get_mask_reference_nn(), self.get_reference_mask_nn()
        n = int(self._mask_edge.sum()) + int(self._ref_edge.sum())
        return math.sqrt(1.0 / n) * math.sqrt(float((a * a).sum()) + float((b * b).sum()))

    def get_average_symmetric_surface_distance(self):
        a, b = self.get_mask_reference_nn(), self.get_reference_mask_nn()
        n = int(self._mask_edge.sum()) + int(self._ref_edge.sum())
        return 1.0 / n * (float(a.sum()) + float(b.sum()))
