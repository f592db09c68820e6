"""NIfTI-1 volumes of the LiTS dataset -- host-side mirror of the reference's DataLoader/Liver/nii_kits.py:21-75
(`read_lits`, `read_nii`, `write_nii`) without nibabel (not installed here): a minimal single-file NIfTI-1 codec
(.nii / .nii.gz, 348-byte header, the scalar datatypes LiTS uses) plus the reference's orientation rule.

Orientation (nii_kits.py:33-50): with A = best affine (sform if sform_code > 0, else qform if qform_code > 0, else the
pixdim scaling -- nibabel's `get_best_affine`), `trans[i]` = the data axis world axis i runs along; the array is
transposed to (z, y, x) and flipped so that x decreases, y decreases and z increases with the index; LiTS cases
28..47 (volumes) / 28..51 (labels) are additionally flipped in x ("special").
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


class Nifti1Header(object):
    """The header fields the pipeline uses."""

    def __init__(self, shape, dtype, pixdim=(1.0, 1.0, 1.0), sform=None, qform_code=0, quatern=(0.0, 0.0, 0.0),
                 qoffset=(0.0, 0.0, 0.0), qfac=1.0, scl_slope=0.0, scl_inter=0.0, vox_offset=352.0, endian="<"):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.pixdim = tuple(float(p) for p in pixdim)
        self.sform = None if sform is None else np.asarray(sform, np.float64).reshape(3, 4)
        self.qform_code = int(qform_code)
        self.quatern, self.qoffset, self.qfac = tuple(quatern), tuple(qoffset), float(qfac)
        self.scl_slope, self.scl_inter = float(scl_slope), float(scl_inter)
        self.vox_offset = float(vox_offset)
        self.endian = endian

    def get_zooms(self):
        return self.pixdim[:len(self.shape)]

    def get_data_shape(self):
        return self.shape

    def get_qform(self):
        b, c, d = self.quatern
        a2 = 1.0 - (b * b + c * c + d * d)
        a = np.sqrt(a2) if a2 > 0 else 0.0
        rot = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                        [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        zooms = np.array(self.pixdim[:3], np.float64).copy()
        zooms[2] *= -1.0 if self.qfac < 0 else 1.0
        aff = np.eye(4)
        aff[:3, :3] = rot * zooms[None, :]
        aff[:3, 3] = self.qoffset
        return aff

    def get_best_affine(self):
        """nibabel's rule: sform, else qform, else the pixdim scaling (centre shift omitted: only the axes are used)."""
        if self.sform is not None:
            aff = np.eye(4)
            aff[:3, :] = self.sform
            return aff
        if self.qform_code > 0:
            return self.get_qform()
        aff = np.eye(4)
        aff[0, 0], aff[1, 1], aff[2, 2] = self.pixdim[:3]
        aff[0, 0] *= -1.0          # nibabel's base affine is LAS+ -> RAS flips x
        return aff


def _open(path, mode):
    path = str(path)
    return gzip.open(path, mode) if path.endswith(".gz") else open(path, mode)


def load(file_name):
    """-> (Nifti1Header, raw array in file order [x, y, z], scaling applied as float64 like nibabel's get_fdata)."""
    with _open(file_name, "rb") as f:
        raw = f.read()
    endian = "<"
    if struct.unpack("<i", raw[:4])[0] != 348:
        endian = ">"
        if struct.unpack(">i", raw[:4])[0] != 348:
            raise ValueError("{}: not a NIfTI-1 file (sizeof_hdr != 348)".format(file_name))
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError("{}: bad NIfTI-1 magic {!r}".format(file_name, raw[344:348]))
    if raw[344:348] == b"ni1\0":
        raise ValueError("{}: header/image pairs (.hdr/.img) are not supported".format(file_name))
    dim = struct.unpack(endian + "8h", raw[40:56])
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError("{}: bad dim[0] = {}".format(file_name, ndim))
    shape = tuple(d for d in dim[1:1 + ndim])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    datatype, = struct.unpack(endian + "h", raw[70:72])
    if datatype not in _DTYPES:
        raise ValueError("{}: unsupported NIfTI datatype {}".format(file_name, datatype))
    pixdim = struct.unpack(endian + "8f", raw[76:108])
    vox_offset, scl_slope, scl_inter = struct.unpack(endian + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(endian + "2h", raw[252:256])
    quatern = struct.unpack(endian + "3f", raw[256:268])
    qoffset = struct.unpack(endian + "3f", raw[268:280])
    srow = struct.unpack(endian + "12f", raw[280:328])
    hdr = Nifti1Header(shape, _DTYPES[datatype], pixdim[1:4], srow if sform_code > 0 else None, qform_code, quatern,
                       qoffset, -1.0 if pixdim[0] < 0 else 1.0, scl_slope, scl_inter, vox_offset, endian)
    count = int(np.prod(shape))
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    data = np.frombuffer(raw, dt, count, int(vox_offset)).reshape(shape, order="F")
    data = data.astype(np.float64)
    if scl_slope not in (0.0,) and not np.isnan(scl_slope):
        data = data * scl_slope + scl_inter
    return hdr, data


def save(data_xyz, header, out_path):
    """Write `data_xyz` (file order [x, y, z]) with `header`'s geometry (dtype from the array)."""
    data_xyz = np.asarray(data_xyz)
    dt = np.dtype(data_xyz.dtype)
    if dt not in _CODES:
        raise ValueError("unsupported dtype {}".format(dt))
    h = bytearray(352)
    struct.pack_into("<i", h, 0, 348)
    dims = [data_xyz.ndim] + list(data_xyz.shape) + [1] * (7 - data_xyz.ndim)
    struct.pack_into("<8h", h, 40, *dims)
    struct.pack_into("<h", h, 70, _CODES[dt])
    struct.pack_into("<h", h, 72, dt.itemsize * 8)
    pix = [header.qfac] + list(header.pixdim[:3]) + [1.0] * 4
    struct.pack_into("<8f", h, 76, *pix)
    struct.pack_into("<3f", h, 108, 352.0, header.scl_slope, header.scl_inter)
    struct.pack_into("<2h", h, 252, header.qform_code, 1 if header.sform is not None else 0)
    struct.pack_into("<3f", h, 256, *header.quatern)
    struct.pack_into("<3f", h, 268, *header.qoffset)
    if header.sform is not None:
        struct.pack_into("<12f", h, 280, *np.asarray(header.sform, np.float32).reshape(-1))
    h[344:348] = b"n+1\0"
    with _open(out_path, "wb") as f:
        f.write(bytes(h))
        f.write(np.asfortranarray(data_xyz).astype(dt.newbyteorder("<")).tobytes(order="F"))


def _orient(affine):
    trans = np.argmax(np.abs(affine[:3, :3]), axis=1)
    return trans, (affine[0, trans[0]] > 0, affine[1, trans[1]] > 0, affine[2, trans[2]] < 0)


def read_nii(file_name, out_dtype=np.int16, special=False, only_header=False):
    """nii_kits.py:33-50: -> (header, data [z, y, x])."""
    vh, data = load(file_name)
    if only_header:
        return vh
    trans, (flip_x, flip_y, flip_z) = _orient(vh.get_best_affine())
    data = data.astype(out_dtype).transpose(*trans[::-1])
    if special:
        data = np.flip(data, axis=2)
    if flip_x:                                  # Increase x from Right to Left
        data = np.flip(data, axis=2)
    if flip_y:                                  # Increase y from Anterior to Posterior
        data = np.flip(data, axis=1)
    if flip_z:                                  # Increase z from Interior to Superior
        data = np.flip(data, axis=0)
    return vh, data


def read_lits(num, obj, file_name, only_header=False):
    """nii_kits.py:21-30: the x-flipped LiTS cases."""
    if obj == "vol":
        return read_nii(file_name, out_dtype=np.int16, special=28 <= int(num) < 48, only_header=only_header)
    if obj == "lab":
        return read_nii(file_name, out_dtype=np.uint8, special=28 <= int(num) < 52, only_header=only_header)
    raise ValueError("obj must be 'vol' or 'lab'")


def write_nii(data, header, out_path, out_dtype=np.int16, special=False, affine=None):
    """nii_kits.py:53-75: the inverse of read_nii (data [z, y, x] -> file order)."""
    if header is not None:
        affine = header.get_best_affine()
    affine = np.asarray(affine, np.float64)
    assert len(np.where(affine[:3, :3].reshape(-1) != 0)[0]) == 3, affine
    trans, (flip_x, flip_y, flip_z) = _orient(affine)
    trans_bk = [int(np.argwhere(np.array(trans[::-1]) == i)[0][0]) for i in range(3)]
    if special:
        data = np.flip(data, axis=2)
    if flip_x:
        data = np.flip(data, axis=2)
    if flip_y:
        data = np.flip(data, axis=1)
    if flip_z:
        data = np.flip(data, axis=0)
    out_image = np.transpose(data, trans_bk).astype(out_dtype)
    if header is None:
        zooms = np.abs(affine[:3, :3]).max(axis=0)
        header = Nifti1Header(out_image.shape, out_dtype, zooms, sform=affine[:3, :])
    save(out_image, header, out_path)
