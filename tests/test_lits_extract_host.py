"""NIfTI -> PNG / meta.json export (CPU): data/extract.py against brute-force definitions on a synthetic case, and the
exported set read back through the training pipeline's own parsers (reference DataLoader/Liver/extract.py:61-213)."""
import json

import numpy as np

from oracle import lits_ops

from boxsegliver_amd.data import extract, lits, nii_kits
from boxsegliver_amd.utils import array_kits


def test_array_kits_helpers():
    m = np.zeros((6, 7), np.uint8)
    m[2:5, 1:4] = 1
    assert array_kits.bbox_from_mask(m, 1).tolist() == [1, 2, 3, 4]                     # (x1, y1, x2, y2) inclusive
    assert array_kits.bbox_from_mask(m, 2).tolist() == [0, 0, 0, 0]
    v = np.zeros((4, 5, 6), np.uint8)
    v[1:3, 2:4, 0:5] = 2
    assert array_kits.extract_region(v).tolist() == [0, 2, 1, 4, 3, 2]                  # (x1, y1, z1, x2, y2, z2)
    pts = np.zeros((9, 9), np.uint8)
    pts[4, 2:7] = 1                                                                     # 5 points on a row
    c, s = array_kits.compute_robust_moments(pts, indexing="ij")
    assert c.tolist() == [4.0, 4.0] and np.allclose(s, [0.0, 1.4826])                   # MAD of {-2..2} = 1
    c, s = array_kits.compute_robust_moments(pts, indexing="xy")
    assert c.tolist() == [4.0, 4.0] and np.allclose(s, [1.4826, 0.0])
    c, s = array_kits.compute_robust_moments(np.zeros((3, 3)))
    assert c.tolist() == [-1.0, -1.0] and s.tolist() == [-1.0, -1.0]


def _case(tmp_path, pid=7, depth=8, size=40):
    rng = np.random.RandomState(pid)
    vol = rng.randint(-400, 500, size=(depth, size, size)).astype(np.int16)             # (z, y, x)
    lab = np.zeros((depth, size, size), np.uint8)
    lab[1:7, 5:30, 8:33] = 1
    lab[2:4, 10:14, 12:18] = 2                                                          # tumour A: 2 slices
    lab[5, 20:23, 25:27] = 2                                                            # tumour B: 1 slice
    aff = np.array([[-0.7, 0, 0, 0], [0, -0.7, 0, 0], [0, 0, 2.0, 0], [0, 0, 0, 1.0]])
    src = tmp_path / "Training_Batch"
    src.mkdir(exist_ok=True)
    nii_kits.write_nii(vol, None, src / "volume-{}.nii".format(pid), np.int16, affine=aff)
    nii_kits.write_nii(lab, None, src / "segmentation-{}.nii".format(pid), np.uint8, affine=aff)
    return src, vol, lab


def test_export_meta_and_png(tmp_path):
    src, vol, lab = _case(tmp_path)
    _case(tmp_path, pid=3)
    out = tmp_path / "png"
    metas = extract.nii_3d_to_png(src, out)
    assert [m["PID"] for m in metas] == [3, 7] and json.load((out / "meta.json").open())[1]["PID"] == 7
    m = metas[1]
    assert m["size"] == [8, 40, 40] and np.allclose(m["spacing"], [2.0, 0.7, 0.7])
    assert m["bbox"] == [1, 5, 8, 7, 30, 33]                                            # z1, y1, x1, z2+1, y2+1, x2+1 of labels > 0
    assert m["tumors"] == [[2, 10, 12, 4, 14, 18], [5, 20, 25, 6, 23, 27]] and m["tumor_areas"] == [48, 6]
    assert m["tumor_slices_index"] == [2, 3, 5] and m["tumor_slices_from_to"] == [0, 1, 2, 3] and m["tumor_slices_tid"] == [0, 0, 1]
    assert m["tumor_slices"] == [[10, 12, 14, 18], [10, 12, 14, 18], [20, 25, 23, 27]] and m["tumor_slices_areas"] == [24, 24, 6]
    assert np.allclose(m["tumor_centers"][0], [2.5, 11.5, 14.5]) and np.allclose(m["tumor_slices_centers"][2], [21.0, 25.5])
    # slices: 16-bit (clip(HU) + 200) * 64, labels * 64
    for z in (0, 3, 7):
        im = lits_ops.png_decode((out / "volume-7" / "{:03d}_im.png".format(z)).read_bytes())
        lb = lits_ops.png_decode((out / "volume-7" / "{:03d}_lb.png".format(z)).read_bytes())
        assert im.dtype == np.uint16 and lb.dtype == np.uint8
        np.testing.assert_array_equal(im, ((np.clip(vol[z], -200, 250) + 200) * 64).astype(np.uint16))
        np.testing.assert_array_equal(lb, lab[z] * 64)
    # only_meta writes no slices
    metas2 = extract.nii_3d_to_png(src, tmp_path / "meta_only", only_meta=True)
    assert metas2 == metas and not (tmp_path / "meta_only" / "volume-7").exists()


def test_exported_set_feeds_the_training_pipeline_parsers(tmp_path):
    src, _, _ = _case(tmp_path)
    for pid in (0, 1, 2, 3, 4):
        _case(tmp_path, pid=pid)
    root = tmp_path / "LiTS"
    extract.nii_3d_to_png(src, root / "png")
    (root / "meta.json").write_text((root / "png" / "meta.json").read_text())
    (root / "k_folds.txt").write_text("Fold 0:0 3\nFold 1:1 4\nFold 2:2 7\n")
    train = lits.collect_datasets(root, 2, "train")
    val = lits.collect_datasets(root, 2, "eval")
    assert [c["PID"] for c in train] == [0, 1, 3, 4] and [c["PID"] for c in val] == [2, 7]
    c = val[1]
    assert c["tumor_slices_index"] == [2, 3, 5] and len(c["slices"]) == 3 and c["slices"][2] == [[20, 25, 23, 27]]
