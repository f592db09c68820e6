"""Write tests/golden/png/*.png with a THIRD-PARTY encoder (Pillow's PNG writer: zlib + per-row adaptive filter selection, the
way libpng / SimpleITK wrote the reference's slices, DataLoader/Liver/extract.py:176-187) and tests/golden/png/pixels.npz with
the pixel arrays they were made from.  Run once in the build container (Pillow is importable here; it is NOT needed to run the
tests -- the files and arrays are committed data):

    python tests/golden/make_png_fixtures.py

Every PNG the decode path had seen before round 5 came from this package's own encoder (data/lits.png_encode); these files pin
`png_inflate` + `unetk_png_unfilter` (tests/test_gpu_lits_loader.py) and the host checker `oracle.lits_ops.png_decode`
(tests/test_png_fixtures.py) against an encoder the package did not write."""
import os
import struct

import numpy as np
from PIL import Image, ImageFile

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "png")


def ct_like(rng, h, w, noise):
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    body = ((yy - h / 2) / (0.42 * h)) ** 2 + ((xx - w / 2) / (0.46 * w)) ** 2 <= 1
    hu = rng.normal(40, noise, size=(h, w)) * body + (-200) * (~body) + 60 * np.sin(yy / 9.0) * body
    return ((np.clip(hu, -200, 250) + 200) * 64).astype(np.uint16)          # (HU + 200) * IM_SCALE, extract.py:47


def chunks(data):
    pos, out = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        out.append((typ, n))
        pos += 12 + n
    return out


def main():
    os.makedirs(HERE, exist_ok=True)
    rng = np.random.default_rng(20261005)
    arrays = {}
    # 16-bit slices the way extract.py writes them (sitk.WriteImage of a uint16 image -> 16-bit grayscale)
    arrays["im16_96x80"] = ct_like(rng, 96, 80, 25)
    arrays["im16_33x130"] = ct_like(rng, 33, 130, 5)                        # ragged extents, wider than 64 px
    arrays["im16_128x128_multi_idat"] = ct_like(rng, 128, 128, 60)          # noisy: compresses badly, several IDAT chunks
    # 8-bit label slices: label * LB_SCALE
    lab = (np.abs(arrays["im16_96x80"].astype(np.int32) - 16000) < 900).astype(np.uint8)
    lab[30:50, 20:40] *= 2
    arrays["lb8_96x80"] = (lab * 64).astype(np.uint8)
    arrays["lb8_64x48_gradient"] = ((np.add.outer(np.arange(64) * 3, np.arange(48) * 5) + rng.integers(0, 3, (64, 48))) & 255).astype(np.uint8)
    for name, a in arrays.items():
        path = os.path.join(HERE, name + ".png")
        ImageFile.MAXBLOCK = 8192 if "multi_idat" in name else 65536        # Pillow cuts IDAT chunks at its encoder block size
        img = Image.fromarray(a)                                             # uint16 -> mode I;16, uint8 -> L
        assert img.mode == ("I;16" if a.dtype == np.uint16 else "L")
        img.save(path, format="PNG", optimize=False, compress_level=6)
        back = np.array(Image.open(path))
        assert back.dtype == a.dtype and np.array_equal(back, a), name      # Pillow reads its own file back to the same pixels
        data = open(path, "rb").read()
        print(name, a.shape, a.dtype, len(data), "bytes,", sum(1 for t, _ in chunks(data) if t == b"IDAT"), "IDAT chunk(s)")
    np.savez_compressed(os.path.join(HERE, "pixels.npz"), **arrays)


if __name__ == "__main__":
    main()
