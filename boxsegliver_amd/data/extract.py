"""NIfTI -> training-set export of LiTS -- host-side mirror of the reference's DataLoader/Liver/extract.py:61-213
(`process_case`, `nii_3d_to_png`): per case the HU volume is clipped to [-200, 250], shifted and scaled by 64 into 16-bit
PNG slices `<out>/volume-<PID>/<z:03d>_im.png`, the labels x 64 into `<z:03d>_lb.png`, and `meta.json` receives the
record `data/lits.py` consumes (size, spacing, liver box, 3-D tumour components with robust centre / spread, and the same
per slice).  Pure host code on this package's NIfTI and PNG codecs (no nibabel / SimpleITK in this image)."""
import json
from pathlib import Path

import numpy as np
import scipy.ndimage as ndi

from ..utils import array_kits
from . import nii_kits
from .lits import GRAY_MAX, GRAY_MIN, IM_SCALE, LB_SCALE, png_encode


def process_case(vol_case, dst_path, only_meta=False):
    """extract.py:61-189 for one `volume-<PID>.nii[.gz]` (its labels next to it as `segmentation-<PID>`)."""
    vol_case, dst_path = Path(vol_case), Path(dst_path)
    stem = vol_case.name.split(".")[0]
    pid = int(stem.split("-")[-1])
    vh, volume = nii_kits.read_nii(vol_case, out_dtype=np.int16, special=28 <= pid < 48)
    volume = ((np.clip(volume, GRAY_MIN, GRAY_MAX) - GRAY_MIN) * IM_SCALE).astype(np.uint16)
    lab_case = vol_case.parent / vol_case.name.replace("volume", "segmentation")
    _, labels = nii_kits.read_nii(lab_case, out_dtype=np.uint8, special=28 <= pid < 52)
    assert volume.shape == labels.shape, "Vol{} vs Lab{}".format(volume.shape, labels.shape)

    b = array_kits.extract_region(labels).tolist()                   # (x1, y1, z1, x2, y2, z2) inclusive
    bbox = [int(b[2]), int(b[1]), int(b[0]), int(b[5]) + 1, int(b[4]) + 1, int(b[3]) + 1]

    disc3 = ndi.generate_binary_structure(3, connectivity=2)
    tumors, _ = ndi.label(labels == 2, disc3)
    slices = ndi.find_objects(tumors)
    objects = [[z.start, y.start, x.start, z.stop, y.stop, x.stop] for z, y, x in slices]
    all_centers, all_stddevs, tumor_areas = [], [], []
    per_tumor = {i: {"centers": [], "stddevs": [], "areas": [], "slices": []} for i in range(len(slices))}
    z_rev = {i: {"tid": [], "rid": []} for i in range(volume.shape[0])}
    for j, sli in enumerate(slices):
        # NB labels[sli] == 2 of the bounding box, as the reference: voxels of OTHER tumours inside the box count too
        region = labels[sli] == 2
        center, stddev = array_kits.compute_robust_moments(region, indexing="ij", min_std=0.)
        center = center + np.array(objects[j][:3], np.float32)
        all_centers.append(center.tolist())
        all_stddevs.append([round(float(x), 3) for x in stddev])
        tumor_areas.append(int(np.count_nonzero(region)))
        for k in range(region.shape[0]):
            patch = region[k]
            c2, s2 = array_kits.compute_robust_moments(patch, indexing="ij", min_std=0.)
            c2 = c2 + np.array(objects[j][1:3], np.float32)
            per_tumor[j]["centers"].append(c2.tolist())
            per_tumor[j]["stddevs"].append([round(float(x), 3) for x in s2])
            per_tumor[j]["areas"].append(int(np.count_nonzero(patch)))
            x1, y1, x2, y2 = [int(v) for v in array_kits.bbox_from_mask(patch, mask_values=1).tolist()]
            per_tumor[j]["slices"].append([y1 + objects[j][1], x1 + objects[j][2], y2 + 1 + objects[j][1], x2 + 1 + objects[j][2]])
            z_rev[objects[j][0] + k]["tid"].append(j)
            z_rev[objects[j][0] + k]["rid"].append(k)

    index = [j for j in z_rev if len(z_rev[j]["tid"]) > 0]
    from_to, centers, stddevs, areas, boxes, tids = [0], [], [], [], [], []
    for j in index:
        from_to.append(from_to[-1] + len(z_rev[j]["tid"]))
        for tid, rid in zip(z_rev[j]["tid"], z_rev[j]["rid"]):
            centers.append(per_tumor[tid]["centers"][rid])
            stddevs.append(per_tumor[tid]["stddevs"][rid])
            areas.append(per_tumor[tid]["areas"][rid])
            boxes.append(per_tumor[tid]["slices"][rid])
            tids.append(tid)

    meta = {"PID": pid, "vol_case": str(vol_case), "lab_case": str(lab_case),
            "size": [int(x) for x in vh.get_data_shape()[::-1]], "spacing": [float(x) for x in vh.get_zooms()[::-1]],
            "bbox": bbox, "tumors": objects, "tumor_areas": tumor_areas, "tumor_centers": all_centers,
            "tumor_stddevs": all_stddevs, "tumor_slices_from_to": from_to, "tumor_slices": boxes, "tumor_slices_index": index,
            "tumor_slices_centers": centers, "tumor_slices_stddevs": stddevs, "tumor_slices_areas": areas,
            "tumor_slices_tid": tids}
    if not only_meta:
        dst_dir = dst_path / stem
        dst_dir.mkdir(parents=True, exist_ok=True)
        for j, (img, lab) in enumerate(zip(volume, labels * LB_SCALE)):
            (dst_dir / "{:03d}_im.png".format(j)).write_bytes(png_encode(np.ascontiguousarray(img)))
            (dst_dir / "{:03d}_lb.png".format(j)).write_bytes(png_encode(np.ascontiguousarray(lab.astype(np.uint8))))
    return meta


def nii_3d_to_png(in_path, out_path, only_meta=False):
    """extract.py:192-213: every volume-*.nii[.gz] of in_path -> out_path/volume-<PID>/*.png + out_path/meta.json."""
    src, dst = Path(in_path), Path(out_path)
    dst.mkdir(parents=True, exist_ok=True)
    files = sorted(list(src.glob("volume-*.nii")) + list(src.glob("volume-*.nii.gz")),
                   key=lambda x: int(x.name.split(".")[0].split("-")[-1]))
    metas = sorted((process_case(f, dst, only_meta) for f in files), key=lambda m: m["PID"])
    with (dst / "meta.json").open("w") as f:
        json.dump(metas, f)
    return metas
