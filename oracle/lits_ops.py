"""TEST INFRASTRUCTURE (see oracle/__init__.py): numpy restatement of the reference's per-sample training
pre-processing, DataLoader/Liver/input_pipeline.py:243-284 (`data_processing_train`), with the TF-1.13 semantics of the
ops it calls: tf.image.crop_to_bounding_box, tf.image.resize_bilinear(align_corners=True) (float32 lerp, order
top/bottom then vertical), tf.image.resize_nearest_neighbor(align_corners=True) (roundf), clip + normalise,
`seg / lab_scale` cast to int32, flips.  Noise is left out (its generator cannot be matched); the GPU test checks its
statistics instead."""
import numpy as np


def resize_bilinear_align_corners(img, out_h, out_w):
    """img [h, w] float32 -> [out_h, out_w] float32 (tensorflow/core/kernels/resize_bilinear_op.cc)."""
    h, w = img.shape
    hs = np.float32((h - 1) / (out_h - 1)) if out_h > 1 else np.float32(0)
    ws = np.float32((w - 1) / (out_w - 1)) if out_w > 1 else np.float32(0)
    ys = (np.arange(out_h, dtype=np.float32) * hs).astype(np.float32)
    xs = (np.arange(out_w, dtype=np.float32) * ws).astype(np.float32)
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    ly = (ys - y0).astype(np.float32)[:, None]
    lx = (xs - x0).astype(np.float32)[None, :]
    img = img.astype(np.float32)
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = (tl + (tr - tl) * lx).astype(np.float32)
    bot = (bl + (br - bl) * lx).astype(np.float32)
    return (top + (bot - top) * ly).astype(np.float32)


def resize_nearest_align_corners(img, out_h, out_w):
    h, w = img.shape
    hs = np.float32((h - 1) / (out_h - 1)) if out_h > 1 else np.float32(0)
    ws = np.float32((w - 1) / (out_w - 1)) if out_w > 1 else np.float32(0)
    # roundf = round half away from zero (arguments are non-negative here)
    ys = np.minimum(np.floor((np.arange(out_h, dtype=np.float32) * hs).astype(np.float64) + 0.5).astype(np.int64), h - 1)
    xs = np.minimum(np.floor((np.arange(out_w, dtype=np.float32) * ws).astype(np.float64) + 0.5).astype(np.int64), w - 1)
    return img[ys][:, xs]


def process_sample(slices, seg, box, clip, out_hw, lab_scale=64, flip_lr=False, flip_ud=False):
    """slices: list of uint16 [h, w] arrays or None (zero padding slice); seg uint8 [h, w] or None;
    box = [off_y, off_x, crop_h, crop_w]; returns (image f32 [H, W, C], label int32 [H, W])."""
    off_y, off_x, ch, cw = box
    oh, ow = out_hw
    lo, hi = np.float32(clip[0]), np.float32(clip[1])
    chans = []
    for s in slices:
        if s is None:
            chans.append(np.zeros((oh, ow), np.float32))
            continue
        r = resize_bilinear_align_corners(s[off_y:off_y + ch, off_x:off_x + cw].astype(np.float32), oh, ow)
        chans.append(((np.clip(r, lo, hi) - lo) / (hi - lo)).astype(np.float32))
    img = np.stack(chans, axis=-1)
    if seg is None:
        lab = np.zeros((oh, ow), np.int32)
    else:
        lab = (resize_nearest_align_corners(seg[off_y:off_y + ch, off_x:off_x + cw], oh, ow) // lab_scale).astype(np.int32)
    if flip_lr:
        img, lab = img[:, ::-1], lab[:, ::-1]
    if flip_ud:
        img, lab = img[::-1], lab[::-1]
    return np.ascontiguousarray(img), np.ascontiguousarray(lab)


def png_decode(data):
    """8- or 16-bit grayscale, non-interlaced PNG -> ndarray (uint8 / uint16), entirely on the HOST: the checker of the product's
    loader (boxsegliver_amd.data.lits.SliceStore: zlib inflate on host threads + unetk_png_unfilter on the device).  The five
    row filters as the PNG specification defines them (what cv2.imread undoes in the reference's pipeline,
    DataLoader/Liver/input_pipeline.py:243-284); Average / Paeth rows run through a per-byte Python loop -- small images only."""
    import struct
    import zlib
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        chunk = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", chunk)
        elif typ == b"IDAT":
            idat.append(chunk)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 0 or interlace != 0 or depth not in (8, 16):
        raise ValueError("only non-interlaced 8/16-bit grayscale PNGs are supported")
    flat = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    bpp = depth // 8
    stride = w * bpp
    raw = flat.reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:                                       # Sub: byte lanes are independent running sums mod 256
            cur = line.copy()
            for k in range(bpp):
                cur[k::bpp] = np.cumsum(line[k::bpp]) & 255
        elif ft in (3, 4):                                  # Average / Paeth: sequential
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        else:
            raise ValueError("invalid PNG filter type {}".format(ft))
        out[y] = cur
        prev = cur
    if depth == 8:
        return out.copy()
    return (out[:, 0::2].astype(np.uint16) << 8 | out[:, 1::2].astype(np.uint16)).copy()
