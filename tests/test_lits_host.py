"""CPU: host side of the LiTS input pipeline: PNG codec, k-folds file, meta.json case parsing, the
training sampler's invariants (reference DataLoader/Liver/input_pipeline.py:73-198,285-378; DataLoader/misc.py:45-74)."""
import argparse
import json
import struct
import zlib

import numpy as np
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
    np.testing.assert_array_equal(lits.png_decode(lits.png_encode(arr)), arr)
    for ft in range(5):
        np.testing.assert_array_equal(lits.png_decode(_png_with_filter(arr, ft)), arr)
    with pytest.raises(ValueError):
        lits.png_decode(b"not a png")


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


def test_sampler_invariants():
    cfg = argparse.Namespace(im_height=48, im_width=48, im_channel=3)
    cases = [lits.parse_case(_meta_case(i)) for i in range(4)]
    gen = lits.gen_train_batch(cases, 8, liver_percent=0.66, tumor_percent=0.5, random_scale=(1.0, 1.4),
                               random_window_level=True, config=cfg, seed=7)
    n_tumor = 0
    for k in range(8 * 20):
        chans, lab, (off_y, off_x, ch, cw), pid, (lo, hi) = next(gen)
        case = cases[pid]
        depth, h, w = case["size"]
        assert 48 <= ch <= 67 and 48 <= cw <= 67 and 0 <= off_y and off_y + ch <= h and 0 <= off_x and off_x + cw <= w
        assert len(chans) == 3 and chans[1] == lab and 0 <= lab < depth
        assert chans[0] == (lab - 1 if lab > 0 else -1) and chans[2] == (lab + 1 if lab + 1 < depth else -1)
        assert 10 * 64 <= lo <= 50 * 64 and 500 * 64 <= hi <= 540 * 64
        if k % 8 < 4:                                                       # the first ceil(8 * 0.5) of a batch: tumor slices
            assert lab in case["tumor_slices_index"]
            n_tumor += 1
        elif k % 8 < 6:                                                     # up to ceil(8 * 0.66) = 6: liver slices
            assert case["bbox"][0] <= lab <= case["bbox"][3] - 1
    assert n_tumor == 80
    # fixed window without random_window_level; same seed -> same stream
    g1 = lits.gen_train_batch(cases, 4, config=cfg, seed=3)
    g2 = lits.gen_train_batch(cases, 4, config=cfg, seed=3)
    a, b = [next(g1) for _ in range(8)], [next(g2) for _ in range(8)]
    assert a == b and a[0][4] == (50 * 64.0, 500 * 64.0)
