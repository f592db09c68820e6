"""CPU: host side of the LiTS input pipeline: PNG codec, k-folds file, meta.json case parsing, the
training sampler's invariants (reference DataLoader/Liver/input_pipeline.py:73-198,285-378; DataLoader/misc.py:45-74)."""
import argparse
import json
import struct
import zlib

import numpy as np

from oracle import lits_ops
import pytest

from boxsegliver_amd.data import lits


def _png_with_filter(arr, ftype):
    """Encode with ONE filter type on every row (reference PNGs come from cv2, which uses adaptive filtering)."""
    h, w = arr.shape
    depth = 16 if arr.dtype == np.uint16 else 8
    bpp = depth // 8
    rows = np.frombuffer(arr.astype(">u2").tobytes() if depth == 16 else arr.tobytes(), dtype=np.uint8).reshape(h, w * bpp)
    raw = bytearray()
    prev = np.zeros(w * bpp, np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        out = np.zeros_like(cur)
        for i in range(w * bpp):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ftype == 0:
                pred = 0
            elif ftype == 1:
                pred = a
            elif ftype == 2:
                pred = b
            elif ftype == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - pred) & 255
        raw += bytes([ftype]) + out.astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, p):
        return struct.pack(">I", len(p)) + t + p + struct.pack(">I", zlib.crc32(t + p) & 0xffffffff)

    idat = zlib.compress(bytes(raw))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)) + \
        chunk(b"IDAT", idat[:len(idat) // 2]) + chunk(b"IDAT", idat[len(idat) // 2:]) + chunk(b"IEND", b"")


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_png_codec_roundtrip_all_filters(dtype):
    rng = np.random.default_rng(1)
    arr = rng.integers(0, np.iinfo(dtype).max, size=(9, 13)).astype(dtype)
    np.testing.assert_array_equal(lits_ops.png_decode(lits.png_encode(arr)), arr)
    for ft in range(5):
        np.testing.assert_array_equal(lits_ops.png_decode(_png_with_filter(arr, ft)), arr)
    with pytest.raises(ValueError):
        lits_ops.png_decode(b"not a png")


def test_k_folds_file_format(tmp_path):
    f = tmp_path / "k_folds.txt"
    f.write_text("Fold 0:0 11 18\nFold 1:2 19 29\nFold 2:3 6\n")          # the shipped data/LiTS/k_folds.txt format
    assert lits.read_or_create_k_folds(f, []) == [["0", "11", "18"], ["2", "19", "29"], ["3", "6"]]
    g = tmp_path / "new.txt"
    folds = lits.read_or_create_k_folds(g, list(range(10)), k_split=3, seed=1357)
    assert sorted(int(x) for fold in folds for x in fold) == list(range(10))
    assert lits.read_or_create_k_folds(g, []) == [[str(x) for x in f] for f in folds]     # second call reads the file (strings)
    with pytest.raises(ValueError):
        lits.read_or_create_k_folds(tmp_path / "bad.txt", [1, 2], k_split=0)


def _meta_case(pid, depth=12, size=96):
    """A case in the shipped meta.json schema (nested lists stored as JSON strings)."""
    return {"PID": pid, "size": [depth, size, size], "spacing": [2.5, 0.8, 0.8], "bbox": [2, 20, 24, depth - 2, 70, 72],
            "tumors": "[]", "tumor_areas": [], "tumor_centers": "[]", "tumor_stddevs": "[]",
            "tumor_slices_from_to": [0, 1, 3], "tumor_slices": "[[30, 34, 38, 40], [31, 33, 39, 41], [50, 52, 55, 58]]",
            "tumor_slices_index": [4, 5], "tumor_slices_centers": "[[34.0, 37.0], [35.0, 37.0], [52.5, 55.0]]",
            "tumor_slices_stddevs": "[[2.0, 2.0], [2.0, 2.0], [1.0, 1.0]]", "tumor_slices_areas": [40, 50, 9],
            "tumor_slices_tid": [0, 0, 1]}


def test_parse_case_groups_and_filters_tumor_slices():
    c = lits.parse_case(_meta_case(3))
    assert c["tumor_slices_index"] == [4, 5] and c["slices"] == [[[30, 34, 38, 40]], [[31, 33, 39, 41], [50, 52, 55, 58]]]
    c = lits.parse_case(_meta_case(3), filter_size=45)                       # areas 40 | 50, 9 -> only the 50 survives
    assert c["tumor_slices_index"] == [5] and c["slices"] == [[[31, 33, 39, 41]]]
    assert "tumors" not in c and "tumor_slices" not in c


def test_collect_datasets_splits_by_fold(tmp_path):
    (tmp_path / "meta.json").write_text(json.dumps([_meta_case(i) for i in range(6)]))
    (tmp_path / "k_folds.txt").write_text("Fold 0:0 3\nFold 1:1 4\nFold 2:2 5\n")
    train = lits.collect_datasets(tmp_path, 2, "train")
    val = lits.collect_datasets(tmp_path, 2, "eval_online")
    assert [c["PID"] for c in train] == [0, 1, 3, 4] and [c["PID"] for c in val] == [2, 5]
    with pytest.raises(ValueError):
        lits.collect_datasets(tmp_path, 3, "train")


def test_sampler_policy_forced_shares_and_crop_invariants():
    """TrainSampler.draw emits a whole batch at once; the POLICY is the reference's (input_pipeline.py:285-378): forced
    tumor / liver shares in batch order, crops inside the slice and placed around the object, neighbour channels with -1
    padding, window ranges.  (The random stream is this package's own.)"""
    cfg = argparse.Namespace(im_height=48, im_width=48, im_channel=3)
    cases = [lits.parse_case(_meta_case(i)) for i in range(4)]
    smp = lits.TrainSampler(cases, 8, cfg, liver_percent=0.66, tumor_percent=0.5, random_scale=(1.0, 1.4),
                            random_window_level=True, random_flip=3, seed=7)
    seen_flips = np.zeros(2)
    inside = total_tumor = 0
    for _ in range(40):
        b = smp.draw()
        assert b["kind"].tolist() == [0, 0, 0, 0, 1, 1, 2, 2]             # ceil(8 * .5) tumor, up to ceil(8 * .66) = 6 liver
        for j in range(8):
            case = cases[int(b["case"][j])]
            depth, h, w = case["size"]
            off_y, off_x, ch, cw = b["box"][j]
            z = int(b["z"][j])
            assert case["PID"] == b["pid"][j]
            assert 48 <= ch <= 67 and 48 <= cw <= 67 and 0 <= off_y and off_y + ch <= h and 0 <= off_x and off_x + cw <= w
            assert b["chans"][j].tolist() == [z - 1 if z > 0 else -1, z, z + 1 if z + 1 < depth else -1]
            lo, hi = b["clip"][j]
            assert 10 * 64 <= lo <= 50 * 64 and 500 * 64 <= hi <= 540 * 64
            if b["kind"][j] == 0:
                k = case["tumor_slices_index"].index(z)                   # a tumor slice of a case that has tumors
                total_tumor += 1
                # the crop covers at least one tumor box of that slice whenever it is placed by the "inside" rule
                inside += any(off_y <= bb[0] and bb[2] <= off_y + ch and off_x <= bb[1] and bb[3] <= off_x + cw
                              for bb in case["slices"][k])
            elif b["kind"][j] == 1:
                assert case["bbox"][0] <= z <= case["bbox"][3] - 1
        seen_flips += b["flips"].sum(0)
    assert total_tumor == 160 and inside > 0.9 * total_tumor
    assert 100 < seen_flips[0] < 220 and 100 < seen_flips[1] < 220         # fair coins on 320 samples
    # the table for the gather kernel: store offsets applied, -1 kept, label slice = centre channel
    tab, clip, names = smp.table({c["PID"]: 1000 * c["PID"] for c in cases})
    assert tab.shape == (8, 10) and tab.dtype == np.int32 and clip.shape == (8, 2) and clip.dtype == np.float32
    assert (tab[:, 3] == tab[:, 1]).all() and (tab[:, 1] // 1000 == names).all()
    assert ((tab[:, 0] == -1) | (tab[:, 0] == tab[:, 1] - 1)).all() and set(np.unique(tab[:, 8:])) <= {0, 1}
    # fixed window and no flips when not asked for; same seed -> same stream; eval-style sampler has no forced shares
    s1 = lits.TrainSampler(cases, 4, cfg, seed=3)
    s2 = lits.TrainSampler(cases, 4, cfg, seed=3)
    a, b2 = s1.draw(), s2.draw()
    assert all(np.array_equal(a[k], b2[k]) for k in a) and (a["clip"] == [50 * 64.0, 500 * 64.0]).all()
    assert a["kind"].tolist() == [2, 2, 2, 2] and not a["flips"].any()
    with pytest.raises(ValueError):
        lits.TrainSampler(cases, 2, argparse.Namespace(im_height=200, im_width=48, im_channel=1), seed=1).draw()   # crop > slice


def test_sampler_distribution_matches_the_policy():
    """Over many draws: tumor samples hit every tumor slice of the chosen case with equal probability, liver samples are
    uniform over the liver's z range, unconstrained samples over the whole depth."""
    cfg = argparse.Namespace(im_height=32, im_width=32, im_channel=1)
    cases = [lits.parse_case(_meta_case(i)) for i in range(2)]
    smp = lits.TrainSampler(cases, 6, cfg, liver_percent=0.6, tumor_percent=0.3, seed=11)
    zs = {0: [], 1: [], 2: []}
    for _ in range(600):
        b = smp.draw()
        for k, z in zip(b["kind"], b["z"]):
            zs[int(k)].append(int(z))
    t = np.bincount(zs[0], minlength=12)
    tumor_slices = cases[0]["tumor_slices_index"]
    assert set(np.flatnonzero(t)) == set(tumor_slices)
    assert t[tumor_slices].min() > 0.8 * t[tumor_slices].mean()
    lz = np.bincount(zs[1], minlength=12)
    z0, z1 = cases[0]["bbox"][0], cases[0]["bbox"][3]
    assert lz[:z0].sum() == 0 and lz[z1:].sum() == 0 and lz[z0:z1].min() > 0.7 * lz[z0:z1].mean()
    az = np.bincount(zs[2], minlength=12)
    assert az.min() > 0.6 * az.mean()


def test_real_meta_json_excerpt_is_digested_by_the_dataset_collection_and_the_sampler(tmp_path):
    """Four cases of the reference's shipped DataLoader/Liver/prepare/meta.json (tests/golden/ref_meta_excerpt.json: data
    copied verbatim by make_ref_fixtures.py) through collect_datasets -> parse_case -> TrainSampler: the real schema (real
    lists, not strings; 500-800-slice volumes; a case without tumours; 244 tumour slices in one case)."""
    import argparse
    import os
    ex = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_meta_excerpt.json")))
    assert ex["n_cases_in_file"] == 131 and [c["PID"] for c in ex["cases"]] == [3, 4, 5, 32]
    (tmp_path / "meta.json").write_text(json.dumps(ex["cases"]))
    (tmp_path / "k_folds.txt").write_text("Fold 0:3 4\nFold 1:5\nFold 2:32\n")
    train = lits.collect_datasets(tmp_path, 2, "train")
    assert [c["PID"] for c in train] == [3, 4, 5]
    raw = {c["PID"]: c for c in ex["cases"]}
    for c in train:
        r = raw[c["PID"]]
        ft = r["tumor_slices_from_to"]
        assert c["size"] == r["size"] and c["bbox"] == r["bbox"] and c["tumor_slices_index"] == r["tumor_slices_index"]
        assert [len(b) for b in c["slices"]] == [ft[i + 1] - ft[i] for i in range(len(ft) - 1)]       # filter_size 0 keeps all
        assert [b for per in c["slices"] for b in per] == r["tumor_slices"]
        z0, y0, x0, z1, y1, x1 = c["bbox"]
        assert all(z0 <= z < z1 for z in c["tumor_slices_index"])                                      # tumours lie inside the liver's z range
        assert all(y0 <= b[0] and b[2] <= y1 and x0 <= b[1] and b[3] <= x1 for per in c["slices"] for b in per)
    # the case without tumours survives as a liver-only validation case unless it has no liver slices either
    val = lits.collect_datasets(tmp_path, 2, "eval_online", filter_only_liver_in_val=False)
    assert [c["PID"] for c in val] == [32] and val[0]["slices"] == [] and val[0]["tumor_slices_index"] == []
    assert lits.collect_datasets(tmp_path, 2, "eval_online") == []                                     # ... and is dropped by default
    # filter_size drops small tumours and the slices that hold nothing else
    big = lits.collect_datasets(tmp_path, 2, "train", filter_tumor_size=200)
    for c, full in zip(big, train):
        areas, ft = raw[c["PID"]]["tumor_slices_areas"], raw[c["PID"]]["tumor_slices_from_to"]
        keep = [z for i, z in enumerate(full["tumor_slices_index"]) if any(a > 200 for a in areas[ft[i]:ft[i + 1]])]
        assert c["tumor_slices_index"] == keep and len(c["slices"]) == len(keep)
    # the sampler's tables on the real extents: 512 x 512 slices, crops inside the slice, forced tumour / liver shares
    cfg = argparse.Namespace(im_height=256, im_width=256, im_channel=3)
    sm = lits.TrainSampler(train, 32, cfg, liver_percent=0.66, tumor_percent=0.5, random_scale=(1.0, 1.4),
                           random_window_level=True, random_flip=3, seed=5)
    offs, n = {}, 0
    for c in train:
        offs[c["PID"]] = n
        n += c["size"][0]
    for _ in range(20):
        b = sm.draw()
        assert (b["kind"][:16] == 0).all() and (b["kind"][16:22] == 1).all() and (b["kind"][22:] == 2).all()
        oy, ox, ch, cw = b["box"].T
        assert (oy >= 0).all() and (ox >= 0).all() and (oy + ch <= 512).all() and (ox + cw <= 512).all()
        for k in range(16):                                      # tumour samples sit on a tumour slice of their case
            assert b["z"][k] in raw[int(b["pid"][k])]["tumor_slices_index"]
        tab, clip, pids = sm.table(offs)
        assert tab.shape == (32, 10) and tab[:, :4].max() < n and (tab[:, 3] >= 0).all()
