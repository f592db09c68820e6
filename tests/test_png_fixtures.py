"""CPU: the PNG files under tests/golden/png/ were written by a third-party encoder (Pillow: zlib + adaptive per-row filters;
tests/golden/make_png_fixtures.py) -- the host half of the loader (`data.lits.png_inflate`) and the host checker
(`oracle.lits_ops.png_decode`) must read them back to the committed pixel arrays.  The reference's slices come from SimpleITK /
libpng (DataLoader/Liver/extract.py:176-187) and are read by cv2 (input_pipeline.py:243-284); the device half of the loader is
held to the same files in tests/test_gpu_lits_loader.py."""
import glob
import os
import struct

import numpy as np
import pytest

from boxsegliver_amd.data import lits
from oracle import lits_ops

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "*.png")))


def test_fixture_set_is_complete():
    px = np.load(os.path.join(HERE, "pixels.npz"))
    assert NAMES and sorted(px.files) == NAMES
    assert any(px[n].dtype == np.uint16 for n in NAMES) and any(px[n].dtype == np.uint8 for n in NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_third_party_png_decodes_to_the_committed_pixels(name):
    px = np.load(os.path.join(HERE, "pixels.npz"))[name]
    data = open(os.path.join(HERE, name + ".png"), "rb").read()
    w, h, depth, raw = lits.png_inflate(data)
    assert (h, w) == px.shape and depth == 8 * px.dtype.itemsize
    assert raw.dtype == np.uint8 and raw.size == h * (1 + w * depth // 8)
    assert int(raw.reshape(h, -1)[:, 0].max()) <= 4
    got = lits_ops.png_decode(data)
    assert got.dtype == px.dtype
    np.testing.assert_array_equal(got, px)


def test_fixtures_exercise_adaptive_filters_and_split_idat_streams():
    """What makes these files different from png_encode's: the encoder picked the row filters itself (several types inside one
    file) and one file's zlib stream is cut into several IDAT chunks."""
    kinds, idat_counts = set(), {}
    for name in NAMES:
        data = open(os.path.join(HERE, name + ".png"), "rb").read()
        w, h, depth, raw = lits.png_inflate(data)
        kinds |= set(int(v) for v in raw.reshape(h, -1)[:, 0])
        pos, n_idat = 8, 0
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            n_idat += typ == b"IDAT"
            pos += 12 + n
        idat_counts[name] = n_idat
    assert {1, 2, 4} <= kinds                                   # Sub, Up and Paeth rows (Pillow never picked Average here)
    assert max(idat_counts.values()) >= 3
