"""Offline-evaluation side of the LiTS pipeline (CPU): the NIfTI-1 codec and the reference's orientation rule
(DataLoader/Liver/nii_kits.py:21-75), OpenCV's bilinear resize restated, and the eval generators' contract
(DataLoader/Liver/input_pipeline.py:556-668, input_pipeline_li.py:398-456) on a small synthetic NIfTI dataset."""
import argparse
import json

import numpy as np
import pytest

from boxsegliver_amd.data import lits, nii_kits
from test_lits_host import _meta_case


# ------------------------------------------------------------------------------------------------- NIfTI codec
@pytest.mark.parametrize("dtype,ext", [(np.int16, ".nii"), (np.uint8, ".nii.gz"), (np.float32, ".nii"), (np.uint16, ".nii.gz")])
def test_nifti_roundtrip(tmp_path, dtype, ext):
    rng = np.random.RandomState(0)
    data = rng.randint(0, 200, size=(7, 5, 4)).astype(dtype)
    sform = np.array([[-0.8, 0, 0, 10.0], [0, -0.8, 0, 20.0], [0, 0, 2.5, -30.0]])
    hdr = nii_kits.Nifti1Header(data.shape, dtype, (0.8, 0.8, 2.5), sform=sform)
    path = tmp_path / ("vol" + ext)
    nii_kits.save(data, hdr, path)
    h2, d2 = nii_kits.load(path)
    assert h2.shape == (7, 5, 4) and h2.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(d2, data.astype(np.float64))
    np.testing.assert_allclose(h2.get_best_affine()[:3], sform, rtol=1e-6)
    np.testing.assert_allclose(h2.get_zooms(), (0.8, 0.8, 2.5), rtol=1e-6)


def test_nifti_scaling_and_bad_files(tmp_path):
    data = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    hdr = nii_kits.Nifti1Header(data.shape, np.int16, scl_slope=2.0, scl_inter=-1024.0, sform=np.eye(4)[:3])
    nii_kits.save(data, hdr, tmp_path / "s.nii")
    _, d = nii_kits.load(tmp_path / "s.nii")
    np.testing.assert_array_equal(d, data * 2.0 - 1024.0)
    (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
    with pytest.raises(ValueError):
        nii_kits.load(tmp_path / "bad.nii")


def test_qform_affine_and_base_affine():
    # identity quaternion -> diag(pixdim); qfac = -1 flips z
    h = nii_kits.Nifti1Header((2, 2, 2), np.int16, (0.7, 0.8, 3.0), qform_code=1, qoffset=(1, 2, 3), qfac=-1.0)
    np.testing.assert_allclose(h.get_best_affine(), [[0.7, 0, 0, 1], [0, 0.8, 0, 2], [0, 0, -3.0, 3], [0, 0, 0, 1]])
    # 180 degrees about z: (b, c, d) = (0, 0, 1) -> x and y negated
    h = nii_kits.Nifti1Header((2, 2, 2), np.int16, (1, 1, 1), qform_code=1, quatern=(0.0, 0.0, 1.0))
    np.testing.assert_allclose(np.diag(h.get_best_affine())[:3], [-1, -1, 1], atol=1e-12)
    h = nii_kits.Nifti1Header((2, 2, 2), np.int16, (2, 3, 4))           # neither form: pixdim scaling
    assert np.diag(h.get_best_affine())[:3].tolist() == [-2.0, 3.0, 4.0]


def _world_volume(shape_xyz):
    """data[i, j, k] encodes its own file index: value = i + 100 j + 10000 k."""
    i, j, k = np.meshgrid(*[np.arange(s) for s in shape_xyz], indexing="ij")
    return (i + 100 * j + 10000 * k).astype(np.int32)


@pytest.mark.parametrize("sx,sy,sz", [(-1, -1, 1), (1, -1, 1), (-1, 1, 1), (-1, -1, -1), (1, 1, -1)])
def test_read_nii_orientation_rule(tmp_path, sx, sy, sz):
    """nii_kits.py:33-50: output is (z, y, x); an axis is flipped iff x / y increases or z decreases with the index."""
    data = _world_volume((4, 5, 6))
    sform = np.array([[sx * 0.8, 0, 0, 0], [0, sy * 0.8, 0, 0], [0, 0, sz * 2.0, 0]], np.float64)
    nii_kits.save(data, nii_kits.Nifti1Header(data.shape, np.int32, (0.8, 0.8, 2.0), sform=sform), tmp_path / "o.nii")
    _, out = nii_kits.read_nii(tmp_path / "o.nii", out_dtype=np.int32)
    ref = data.transpose(2, 1, 0)
    if sx > 0:
        ref = ref[:, :, ::-1]
    if sy > 0:
        ref = ref[:, ::-1]
    if sz < 0:
        ref = ref[::-1]
    np.testing.assert_array_equal(out, ref)
    _, sp = nii_kits.read_nii(tmp_path / "o.nii", out_dtype=np.int32, special=True)
    np.testing.assert_array_equal(sp, out[:, :, ::-1])                       # the "special" LiTS cases: extra x flip


def test_read_nii_permuted_axes_and_write_inverse(tmp_path):
    # world x runs along data axis 1, world y along data axis 0 (a transposed acquisition)
    data = _world_volume((4, 5, 6))
    sform = np.array([[0, -1.0, 0, 0], [-1.0, 0, 0, 0], [0, 0, 1.0, 0]])
    hdr = nii_kits.Nifti1Header(data.shape, np.int32, (1, 1, 1), sform=sform)
    nii_kits.save(data, hdr, tmp_path / "p.nii")
    h2, out = nii_kits.read_nii(tmp_path / "p.nii", out_dtype=np.int32)
    np.testing.assert_array_equal(out, data.transpose(2, 0, 1))              # trans = [1, 0, 2] -> transpose(2, 0, 1)
    for special in (False, True):
        nii_kits.write_nii(nii_kits.read_nii(tmp_path / "p.nii", np.int32, special)[1], h2, tmp_path / "w.nii", np.int32, special)
        np.testing.assert_array_equal(nii_kits.load(tmp_path / "w.nii")[1], data)
    # read_lits: which PIDs are "special"
    nii_kits.save(data, hdr, tmp_path / "volume-30.nii")
    _, a = nii_kits.read_lits(30, "vol", tmp_path / "volume-30.nii")
    _, b = nii_kits.read_lits(50, "vol", tmp_path / "volume-30.nii")
    _, c = nii_kits.read_lits(50, "lab", tmp_path / "volume-30.nii")
    np.testing.assert_array_equal(a, b[:, :, ::-1])
    np.testing.assert_array_equal(c.astype(np.int16), a.astype(np.uint8).astype(np.int16))
    assert a.dtype == np.int16 and c.dtype == np.uint8


# ------------------------------------------------------------------------------------------------- cv2.resize restated
def _resize_bruteforce(img, dsize):
    dw, dh = dsize
    sh, sw = img.shape[:2]
    out = np.zeros((dh, dw) + img.shape[2:], np.float64)
    for y in range(dh):
        sy = (y + 0.5) * sh / dh - 0.5
        y0 = int(np.floor(sy))
        fy = sy - y0
        if y0 < 0:
            y0, fy = 0, 0.0
        if y0 >= sh - 1:
            y0, fy = sh - 1, 0.0
        y1 = min(y0 + 1, sh - 1)
        for x in range(dw):
            sx = (x + 0.5) * sw / dw - 0.5
            x0 = int(np.floor(sx))
            fx = sx - x0
            if x0 < 0:
                x0, fx = 0, 0.0
            if x0 >= sw - 1:
                x0, fx = sw - 1, 0.0
            x1 = min(x0 + 1, sw - 1)
            out[y, x] = (img[y0, x0] * (1 - fx) + img[y0, x1] * fx) * (1 - fy) + (img[y1, x0] * (1 - fx) + img[y1, x1] * fx) * fy
    return out


@pytest.mark.parametrize("shape,dsize", [((7, 9, 3), (5, 4)), ((6, 6), (12, 12)), ((10, 8, 2), (8, 10)), ((5, 5, 1), (13, 3))])
def test_cv2_resize_linear_rule(shape, dsize):
    rng = np.random.RandomState(1)
    img = rng.rand(*shape).astype(np.float32)
    got = lits.cv2_resize_linear(img, dsize)
    assert got.shape == (dsize[1], dsize[0]) + shape[2:] and got.dtype == np.float32
    np.testing.assert_allclose(got, _resize_bruteforce(img.astype(np.float64), dsize), atol=1e-6)
    np.testing.assert_array_equal(lits.cv2_resize_linear(img, (shape[1], shape[0])), img)
    # a linear ramp stays linear in the interior when up-sampling by 2
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, :], (4, 1))
    up = lits.cv2_resize_linear(ramp, (16, 8))
    np.testing.assert_allclose(up[0, 1:-1], np.arange(1, 15) * 0.5 - 0.25, atol=1e-6)


# ------------------------------------------------------------------------------------------------- eval generators
def _write_dataset(tmp_path, pids=(0, 1, 2, 3, 4, 5), depth=11, size=96):
    rng = np.random.RandomState(5)
    meta = []
    for pid in pids:
        case = _meta_case(pid, depth=depth, size=size)
        case["vol_case"] = "nii/volume-{}.nii".format(pid)
        case["lab_case"] = "nii/segmentation-{}.nii".format(pid)
        meta.append(case)
        vol = rng.randint(-400, 500, size=(depth, size, size)).astype(np.int16)           # (z, y, x)
        lab = np.zeros((depth, size, size), np.uint8)
        lab[2:depth - 2, 20:70, 24:72] = 1
        lab[4:6, 30:40, 36:42] = 2
        (tmp_path / "nii").mkdir(exist_ok=True)
        aff = np.array([[-0.8, 0, 0, 0], [0, -0.8, 0, 0], [0, 0, 2.5, 0], [0, 0, 0, 1.0]])
        nii_kits.write_nii(vol, None, tmp_path / case["vol_case"], np.int16, affine=aff)
        nii_kits.write_nii(lab, None, tmp_path / case["lab_case"], np.uint8, affine=aff)
    (tmp_path / "meta.json").write_text(json.dumps(meta))
    (tmp_path / "k_folds.txt").write_text("Fold 0:0 3\nFold 1:1 4\nFold 2:2 5\n")
    return meta


def _cfg(**over):
    a = argparse.Namespace(batch_size=4, im_channel=3, im_height=64, im_width=64, eval_skip_num=0, eval_mirror=False,
                           random_flip=1, mode="eval", model="UNet", test_fold=2, filter_size=0, eval_in_patches=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_eval_generator_v2_contract(tmp_path):
    _write_dataset(tmp_path)
    cfg = _cfg()
    # NB the slabs of a case share ONE images buffer (copy.copy is shallow, as in the reference): a consumer must use a
    # slab before advancing the generator -- so keep deep copies here
    items = [(None if f is None else dict(f, images=f["images"].copy()), l)
             for f, l in lits.input_fn_eval("eval", {"args": cfg, "lits_root": tmp_path, "proj_root": tmp_path})]
    ends = [i for i, (f, l) in enumerate(items) if f is None]
    assert len(ends) == 2                                                     # fold 2 = cases 2 and 5
    first = items[:ends[0]]
    seg, vol_path, pads, bbox, resize = items[ends[0]][1]
    # liver box z 2..9 (8 slices), y 20..70, x 24..72 padded by 25 and aligned to 16
    assert bbox[2] == 2 and bbox[5] == 8 and resize is True and vol_path == "nii/volume-2.nii"
    nslices = bbox[5] - bbox[2] + 1
    assert pads == (4 - nslices % 4) % 4 and len(first) == (nslices + pads) // 4
    assert (bbox[3] - bbox[0] + 1) % 16 == 0 and (bbox[4] - bbox[1] + 1) % 16 == 0
    assert seg.shape == (nslices, bbox[4] - bbox[1] + 1, bbox[3] - bbox[0] + 1) and seg.dtype == np.uint8
    for f, l in first:
        assert l is None and f["images"].shape == (4, 64, 64, 3) and f["images"].dtype == np.float32
        assert f["mirror"] == 0 and f["names"] == 2 and 0.0 <= f["images"].min() and f["images"].max() <= 1.0
    # slab content: slice z of the crop, windowed and resized, with its two neighbours as channels
    _, vol = nii_kits.read_lits(2, "vol", tmp_path / "nii/volume-2.nii")
    x1, y1, z1, x2, y2, z2 = bbox
    win = (np.clip(vol.astype(np.float64), -200, 250) + 200) / 450
    for j in range(4):
        z = z1 + j
        ref = lits.cv2_resize_linear(win[z - 1:z + 2, y1:y2 + 1, x1:x2 + 1].transpose(1, 2, 0).astype(np.float32), (64, 64))
        np.testing.assert_allclose(first[0][0]["images"][j], ref, atol=1e-6)
    # no resize: native crop size
    items = list(lits.get_dataset_for_eval_image_v2(lits.collect_datasets(tmp_path, 2, "eval"), _cfg(im_height=-1, im_width=-1),
                                                    tmp_path))
    assert items[0][0]["images"].shape[1:3] == (bbox[4] - bbox[1] + 1, bbox[3] - bbox[0] + 1) and items[ends[0]][1][4] is False


def test_eval_generator_mirror_and_whole_slices(tmp_path):
    _write_dataset(tmp_path, depth=10)
    cases = lits.collect_datasets(tmp_path, 2, "eval")
    cfg = _cfg(eval_mirror=True, random_flip=3)
    items = [(None if f is None else dict(f, images=np.array(f["images"])), l)
             for f, l in lits.get_dataset_for_eval_image_v2(cases[:1], cfg, tmp_path)]
    slabs = [f for f, _ in items if f is not None]
    assert [f["mirror"] for f in slabs[:4]] == [0, 1, 2, 3]
    np.testing.assert_array_equal(slabs[1]["images"], slabs[0]["images"][:, :, ::-1])
    np.testing.assert_array_equal(slabs[2]["images"], slabs[0]["images"][:, ::-1])
    np.testing.assert_array_equal(slabs[3]["images"], slabs[0]["images"][:, ::-1, ::-1])
    # the reference's literal `random_flip & 3 > 0`: random_flip = 1 also yields the double flip
    cfg1 = _cfg(eval_mirror=True, random_flip=1)
    slabs1 = [f["mirror"] for f, _ in lits.get_dataset_for_eval_image_v2(cases[:1], cfg1, tmp_path) if f is not None]
    assert slabs1[:2] == [0, 1] and slabs1[2] == 3
    # whole slices (input_pipeline_li.py): every slice of the volume, zero context at both ends
    items = [(None if f is None else dict(f, images=np.array(f["images"])), l)
             for f, l in lits.get_dataset_for_eval_image(cases[:1], _cfg(), tmp_path)]
    seg, seg_path, pads, bbox, resize = items[-1][1]
    assert seg.shape == (10, 96, 96) and pads == 2 and bbox == (0, 0, 0, 95, 95, 9) and seg_path == "nii/segmentation-2.nii"
    slabs = [f for f, _ in items if f is not None]
    assert len(slabs) == 3 and np.all(slabs[0]["images"][0, :, :, 0] == 0.0)          # channel 0 of slice 0 = zero pad
    assert np.all(slabs[2]["images"][3] == 0.0) and np.all(slabs[2]["images"][2][..., 1:] == 0.0)   # the two padded slices
    assert np.any(slabs[2]["images"][2][..., 0] != 0.0)                                # ... whose left context is slice 9


def test_eval_skip_num_and_errors(tmp_path):
    _write_dataset(tmp_path)
    cases = lits.collect_datasets(tmp_path, 2, "eval")
    ends = [l for f, l in lits.get_dataset_for_eval_image_v2(cases, _cfg(eval_skip_num=1), tmp_path) if f is None]
    assert len(ends) == 1 and ends[0][1] == "nii/volume-5.nii"


def test_patch_centres_tile_the_extent():
    # reference rule (input_pipeline.py:725-731): first / last window flush with the borders, stride <= patch / step
    for extent, psize in ((96, 64), (64, 64), (112, 32), (208, 64), (100, 32)):
        c = lits.patch_centres(extent, psize, 2)
        assert c[0] == psize // 2 and c[-1] == extent - psize // 2
        assert np.all(np.diff(c) <= psize // 2 + 1) and np.all(np.diff(c) > 0) or len(c) == 1
        cover = np.zeros(extent, bool)
        for v in c:
            cover[v - psize // 2:v + psize // 2] = True
        assert cover.all()
    assert list(lits.patch_centres(64, 64)) == [32] and list(lits.patch_centres(96, 64)) == [32, 64]


def test_eval_patches_generator_contract(tmp_path):
    """--eval_in_patches (input_pipeline.py:676-766): native-resolution windows over every slice of the liver box."""
    _write_dataset(tmp_path, size=128)
    cfg = _cfg(eval_in_patches=True, batch_size=5, im_height=64, im_width=64)
    items = list(lits.input_fn_eval("eval", {"args": cfg, "lits_root": tmp_path, "proj_root": tmp_path}))
    last = [i for i, (f, l) in enumerate(items) if l is not None]
    assert len(last) == 2 and last[-1] == len(items) - 1                     # fold 2 = two cases, labels with the last batch
    case = items[:last[0] + 1]
    bbox = case[0][0]["bbox"]
    x1, y1, z1, x2, y2, z2 = bbox
    ch, cw, cd = y2 - y1 + 1, x2 - x1 + 1, z2 - z1 + 1
    assert ch % 16 == 0 and cw % 16 == 0 and ch >= 64 and cw >= 64 and (z1, z2) == (2, 8)
    ny, nx = len(lits.patch_centres(ch, 64)), len(lits.patch_centres(cw, 64))
    total = ny * nx * cd
    assert len(case) == -(-total // 5) and case[-1][0]["pad"] == (5 - total % 5) % 5
    assert all(f["pad"] == 0 and l is None for f, l in case[:-1])
    labels = case[-1][1]
    assert labels.shape == (11, 128, 128)                                    # the WHOLE label volume (z, y, x)
    _, vol = nii_kits.read_lits(2, "vol", tmp_path / "nii/volume-2.nii")
    win = ((np.clip(vol.astype(np.float64), -200, 250) + 200) / 450).astype(np.float32)
    cover = np.zeros((cd, ch, cw), np.int32)
    seen = 0
    for f, _ in case:
        assert f["images"].shape == (5, 64, 64, 3) and f["name"] == 2
        n = 5 - f["pad"]
        assert all(p is None for p in f["position"][n:]) and np.all(f["images"][n:] == 0)
        for i, (z, lb_y, ub_y, lb_x, ub_x) in enumerate(f["position"][:n]):
            assert ub_y - lb_y == 64 and ub_x - lb_x == 64 and 0 <= lb_y and ub_y <= ch and 0 <= lb_x and ub_x <= cw
            ref = win[z1 + z - 1:z1 + z + 2, y1 + lb_y:y1 + ub_y, x1 + lb_x:x1 + ub_x].transpose(1, 2, 0)
            np.testing.assert_array_equal(f["images"][i], ref)
            cover[z, lb_y:ub_y, lb_x:ub_x] += 1
            seen += 1
    assert seen == total and cover.min() >= 1
    # a box smaller than the patch grows to it (parse_case min_shape), clamped inside the slice
    small = dict(lits.collect_datasets(tmp_path, 2, "eval")[0])
    small["bbox"] = [3, 100, 100, 6, 110, 110]
    pid, d, h, w, z1, y1, x1, z2, y2, x2 = lits.parse_case_patches(small, 16, 0, 0, min_shape=(64, 64))
    assert (y2 - y1, x2 - x1) == (64, 64) and 0 <= y1 and y2 <= h and 0 <= x1 and x2 <= w
    with pytest.raises(ValueError):
        lits.parse_case_patches(small, 16, 0, 0, min_shape=(256, 64))
