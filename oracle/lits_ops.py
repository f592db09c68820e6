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
