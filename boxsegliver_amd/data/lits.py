"""LiTS training input pipeline -- MI355X-first restatement of the reference's
DataLoader/Liver/input_pipeline.py (dataset collection :73-198, sampler `gen_train_batch` :285-378, per-sample processing
`data_processing_train` :243-284) and DataLoader/misc.py (`read_or_create_k_folds` :45-74).

On-disk format (the reference's own): `<root>/png/volume-<PID>/<slice:03d>_im.png` 16-bit grayscale = (HU clipped to
[-200, 250] + 200) * IM_SCALE, `<slice:03d>_lb.png` 8-bit = label * LB_SCALE (IM_SCALE = LB_SCALE = 64, :47-48);
`meta.json` (shipped with the reference: 131 cases) and `k_folds.txt` ("Fold i:pid pid ...").

Design: the reference feeds the GPU from tf.data CPU threads (PNG decode + crop + resize per sample per step).  An
MI355X has 288 GB of HBM and the whole decoded training set is ~35 GB as uint16, so `SliceStore` decodes every slice ONCE
and keeps it on the device; a step's batch is then the host-side sampler (a few hundred integer operations) + one gather
kernel (`unetk_lits_batch`).  Decoding: zlib inflate on host threads, the five PNG row filters undone on the device
(`unetk_png_unfilter`), pixels written straight into the resident store (the product needs neither cv2 nor PIL; the host-side
checker of that path, a pure-numpy PNG decoder, lives with the test infrastructure: `oracle/lits_ops.png_decode`, and both
halves are pinned by files a third-party encoder wrote: tests/golden/png/, made with Pillow).

The sampler restates the reference's selection logic literally (forced tumor / liver shares, crop placement around the
object box, random zoom and window level) on a `random.Random(seed)` / `numpy.random.RandomState(seed)` pair instead
of the reference's unseeded global generators.
"""
import copy
import json
import math
import random
import struct
import zlib
from pathlib import Path

import numpy as np
import torch

from ..utils import hostcpu

from .. import ops
from ..utils import distribution_utils

IM_SCALE = 64
LB_SCALE = 64
PNG_MAX_WIDTH = 4096           # csrc/lits.hip: unetk_png_unfilter keeps a band's carried row in LDS
LIVER_PERCENT = 0.66
TUMOR_PERCENT = 0.5
RND_SCALE = (1.0, 1.4)


def add_arguments(parser):
    """DataLoader/Liver/input_pipeline.py:54-70 (names / defaults verbatim) + --lits_root (where png/, meta.json and
    k_folds.txt live; the reference hard-codes <project>/data/LiTS)."""
    group = parser.add_argument_group(title="Input Pipeline Arguments")
    group.add_argument("--test_fold", type=int, default=2)
    group.add_argument("--im_height", type=int, default=256)
    group.add_argument("--im_width", type=int, default=256)
    group.add_argument("--im_channel", type=int, default=3)
    group.add_argument("--filter_size", type=int, default=0, help="Filter tumors small than the given size")
    group.add_argument("--noise_scale", type=float, default=0.1)
    group.add_argument("--zoom_scale", type=float, nargs=2, default=RND_SCALE)
    group.add_argument("--random_flip", type=int, default=1,
                       help="Random flip while training. 0 for no flip, 1 for flip only left/right, "
                            "2 for only up/down, 3 for left/right and up/down")
    group.add_argument("--eval_in_patches", action="store_true")
    group.add_argument("--eval_num_batches_per_epoch", type=int, default=100)
    group.add_argument("--eval_mirror", action="store_true")
    group.add_argument("--liver_percent", type=float, default=LIVER_PERCENT)
    group.add_argument("--tumor_percent", type=float, default=TUMOR_PERCENT)
    group.add_argument("--lits_root", type=str, default="data/LiTS")


# ------------------------------------------------------------------------------------------------- PNG codec
def png_inflate(data):
    """Host half of the loader: parse the chunks of a non-interlaced 8- / 16-bit grayscale PNG and inflate its IDAT stream.
    Returns (width, height, bit_depth, filtered) with `filtered` = uint8 [height * (1 + width * bit_depth / 8)]: per row one
    filter-type byte + the filtered scanline -- what `unetk_png_unfilter` turns into pixels on the device.  zlib.decompress
    releases the GIL, so a thread pool of these scales with the cores."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError("PNG without an IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 0 or interlace != 0 or depth not in (8, 16):
        raise ValueError("only non-interlaced 8/16-bit grayscale PNGs are supported")
    raw = np.frombuffer(zlib.decompress(b"".join(idat) if len(idat) != 1 else idat[0]), dtype=np.uint8)
    if raw.size != h * (w * depth // 8 + 1):
        raise ValueError("PNG data stream has {} bytes, expected {}".format(raw.size, h * (w * depth // 8 + 1)))
    if w > PNG_MAX_WIDTH:
        raise ValueError("PNG is {} pixels wide; unetk_png_unfilter takes rows of at most {}".format(w, PNG_MAX_WIDTH))
    if h and int(raw[::w * depth // 8 + 1].max()) > 4:       # the device kernel's status word stays as the backstop
        raise ValueError("PNG row {} carries filter type {} (corrupt file)".format(
            int(np.argmax(raw[::w * depth // 8 + 1] > 4)), int(raw[::w * depth // 8 + 1].max())))
    return w, h, depth, raw


def png_filter_rows(arr, filters):
    """The FILTERED scanlines of `arr` (uint8 / uint16 [h, w]) with filter type filters[y % len(filters)] on row y -- the
    inverse of the five PNG un-filters, vectorised (filtering only needs the unfiltered neighbours).  uint8 [h, 1 + stride]."""
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape
    bpp = 2 if arr.dtype == np.uint16 else 1
    rows = np.frombuffer(arr.astype(">u2").tobytes() if bpp == 2 else arr.astype(np.uint8).tobytes(), dtype=np.uint8)
    rows = rows.reshape(h, w * bpp).astype(np.int32)
    left = np.concatenate([np.zeros((h, bpp), np.int32), rows[:, :-bpp]], axis=1)                      # a
    up = np.concatenate([np.zeros((1, w * bpp), np.int32), rows[:-1]], axis=0)                         # b
    upleft = np.concatenate([np.zeros((h, bpp), np.int32), up[:, :-bpp]], axis=1)                      # c
    p = left + up - upleft
    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
    preds = [np.zeros_like(rows), left, up, (left + up) >> 1, paeth]
    ft = np.array([int(filters[y % len(filters)]) for y in range(h)], dtype=np.int64)
    out = np.empty((h, 1 + w * bpp), dtype=np.uint8)
    out[:, 0] = ft
    for t in range(5):
        sel = ft == t
        out[sel, 1:] = ((rows[sel] - preds[t][sel]) & 255).astype(np.uint8)
    return out


def png_encode(arr, filters=None, level=6):
    """ndarray uint8 / uint16 [h, w] -> PNG bytes.  filters: None = type 0 on every row; else a sequence of filter types
    (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth) applied cyclically per row -- libpng picks them adaptively, so real files mix
    all five (used to write synthetic datasets in tests / tools and by data/extract.py)."""
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape
    depth = 16 if arr.dtype == np.uint16 else 8
    raw = png_filter_rows(arr, filters if filters is not None else (0,)).tobytes()

    def chunk(typ, payload):
        return struct.pack(">I", len(payload)) + typ + payload + struct.pack(">I", zlib.crc32(typ + payload) & 0xffffffff)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b"")


# ------------------------------------------------------------------------------------------------- dataset lists
def random_split_k_fold(items, k, seed=None):
    """DataLoader/misc.py:24-42: shuffle with NumPy's legacy generator seeded by `seed`, cut into k consecutive blocks of
    floor(n / k) items and deal the remainder one by one to the first folds (pinned on the reference's own output by
    tests/golden/ref_k_folds.json)."""
    items = list(items)
    np.random.RandomState(seed).shuffle(items)       # == np.random.seed(seed); np.random.shuffle(items), state untouched
    size = len(items) // k
    folds = [items[i * size:(i + 1) * size] for i in range(k)]
    for i, extra in enumerate(items[k * size:]):
        folds[i].append(extra)
    return folds


def read_or_create_k_folds(path, list_, k_split=None, seed=None):
    """DataLoader/misc.py:45-74: parse "Fold i:pid pid ..." lines, or create the split and write that file.  (The shipped
    data/LiTS/k_folds.txt is read as is; it was not produced by seed 1357 under today's NumPy.)"""
    path = Path(path)
    if path.exists():
        with path.open() as f:
            return [line[line.find(":") + 1:].strip().split(" ") for line in f.readlines()]
    if not isinstance(k_split, int) or k_split <= 0:
        raise ValueError("Wrong `k_split` value. Need a positive integer, got {}".format(k_split))
    k_folds = random_split_k_fold(list_, k_split, seed) if k_split > 1 else [list(list_)]
    with path.open("w") as f:
        for i, fold in enumerate(k_folds):
            f.write("Fold %d:" % i + " ".join(str(x) for x in fold) + "\n")
    return k_folds


def parse_case(case, filter_size=0):
    """input_pipeline.py:92-119: group the per-slice tumor boxes by slice, drop tumors not larger than filter_size."""
    case = copy.deepcopy(case)
    for k in ("tumors", "tumor_areas", "tumor_centers", "tumor_stddevs"):
        case.pop(k, None)
    ft = case.pop("tumor_slices_from_to")
    assert len(ft) == len(case["tumor_slices_index"]) + 1
    centers, stddevs = _maybe_json(case.pop("tumor_slices_centers")), _maybe_json(case.pop("tumor_slices_stddevs"))
    areas, coords = _maybe_json(case.pop("tumor_slices_areas")), _maybe_json(case.pop("tumor_slices"))
    case["centers"], case["stddevs"], case["slices"] = [], [], []
    slices = copy.deepcopy(case["tumor_slices_index"])
    for ii in range(len(ft) - 1):
        sel = [j for j, a in enumerate(areas[ft[ii]:ft[ii + 1]]) if a > filter_size]
        if not sel:
            case["tumor_slices_index"].remove(slices[ii])
        else:
            case["centers"].append([centers[ft[ii] + j] for j in sel])
            case["stddevs"].append([stddevs[ft[ii] + j] for j in sel])
            case["slices"].append([coords[ft[ii] + j] for j in sel])
    return case


def _maybe_json(v):
    return json.loads(v) if isinstance(v, str) else v      # the shipped meta.json stores the nested lists as strings


def collect_datasets(root, test_fold, mode, filter_tumor_size=0, filter_only_liver_in_val=True):
    """input_pipeline.py:73-198: meta.json + k_folds.txt -> the list of cases of `mode` ("train" / anything else = val)."""
    root = Path(root)
    with (root / "meta.json").open() as f:
        meta = {int(c["PID"]): c for c in json.load(f)}
    k_folds = read_or_create_k_folds(root / "k_folds.txt", sorted(meta), k_split=5, seed=1357)
    if test_fold + 1 > len(k_folds):
        raise ValueError("test_fold too large")
    test = k_folds[test_fold] if test_fold >= 0 else []
    train = [x for i, fold in enumerate(k_folds) if i != test_fold for x in fold]
    if mode == "train":
        return [parse_case(meta[i], filter_tumor_size) for i in sorted(int(x) for x in train)]
    val = [parse_case(meta[i], filter_tumor_size) for i in sorted(int(x) for x in test)]
    return [c for c in val if len(c["slices"]) > 0] if filter_only_liver_in_val else val


# ------------------------------------------------------------------------------------------------- resident slices
class SliceStore(object):
    """Every slice of the given cases decoded once and kept in device memory: `im` int16-typed storage of the uint16
    pixels [n_slices, h, w], `lb` uint8 [n_slices, h, w]; `offset[pid] + z` indexes slice z of case pid.

    Loading (round 4; the reference decodes per sample per step with cv2 on tf.data threads, input_pipeline.py:243-284):
    the store is allocated up front from meta.json's `size` fields; a thread pool reads the files and inflates their zlib
    streams (the GIL is released in both), `chunk` slices at a time, into one of two pinned staging buffers; each chunk is
    uploaded as FILTERED scanlines and `unetk_png_unfilter` writes the pixels straight into the store's slots while the
    pool already inflates the next chunk.  Peak host memory = the two staging buffers (2 x chunk x 0.77 MB at 512 x 512),
    whatever the dataset's size.  Under data parallelism (`strategy` with N > 1 replicas) each rank decodes its contiguous
    1/N share of the slices and the shares are broadcast rank by rank over the process group (RCCL / xGMI), so the node reads
    and inflates the dataset once, not N times."""

    def __init__(self, root, cases, device, strategy=None, chunk=256, threads=None):
        import concurrent.futures
        import os
        root = Path(root)
        self.device = device
        self.offset, files, n = {}, [], 0
        hw = None
        for case in cases:
            pid, depth = int(case["PID"]), int(case["size"][0])
            self.offset[pid] = n
            cur = (int(case["size"][1]), int(case["size"][2]))
            if hw is not None and cur != hw:
                raise ValueError("the resident store needs slices of one size, got {} and {}".format(hw, cur))
            hw = cur
            d = root / "png" / "volume-{:d}".format(pid)
            files.extend((d / "{:03d}_im.png".format(z), d / "{:03d}_lb.png".format(z)) for z in range(depth))
            n += depth
        if n == 0:
            raise ValueError("no slices to load")
        h, w = hw
        self.im = torch.empty((n, h, w), dtype=torch.int16, device=device)        # the uint16 pixels' bit patterns
        self.lb = torch.empty((n, h, w), dtype=torch.uint8, device=device)
        world = strategy.num_replicas_in_sync if strategy is not None else 1
        rank = strategy.rank if strategy is not None else 0
        share = [(n * r) // world for r in range(world + 1)]                       # rank r decodes slices [share[r], share[r + 1])
        lo, hi = share[rank], share[rank + 1]
        im_row, lb_row = h * (2 * w + 1), h * (w + 1)
        chunk = max(1, min(int(chunk), hi - lo))
        stage = [(torch.empty((chunk, im_row), dtype=torch.uint8).pin_memory() if device.type == "cuda" else torch.empty((chunk, im_row), dtype=torch.uint8),
                  torch.empty((chunk, lb_row), dtype=torch.uint8).pin_memory() if device.type == "cuda" else torch.empty((chunk, lb_row), dtype=torch.uint8))
                 for _ in range(2)]
        dev_stage = [(torch.empty((chunk, im_row), dtype=torch.uint8, device=device),
                      torch.empty((chunk, lb_row), dtype=torch.uint8, device=device)) for _ in range(2)]
        done = [None, None]                                                        # event: the device has consumed stage k
        status = torch.zeros(1, dtype=torch.int32, device=device)

        def inflate(slot, j, pair):
            for path, dst, depth_bits in ((pair[0], stage[slot][0], 16), (pair[1], stage[slot][1], 8)):
                try:
                    pw, ph, pd, raw = png_inflate(path.read_bytes())
                except (ValueError, zlib.error) as e:
                    raise ValueError("{}: {}".format(path, e))
                if (ph, pw, pd) != (h, w, depth_bits):
                    raise ValueError("{}: {}x{} {}-bit, expected {}x{} {}-bit".format(path, ph, pw, pd, h, w, depth_bits))
                dst[j].numpy()[:] = raw

        workers = int(threads or min(32, hostcpu.usable_cpus()))      # the cgroup's quota, not the host's count
        failure = None                     # a rank that cannot decode its share must not leave the others in the exchange below
        try:
            with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
                k = 0
                for c0 in range(lo, hi, chunk):
                    slot, cnt = k & 1, min(chunk, hi - c0)
                    if done[slot] is not None:
                        done[slot].synchronize()                                   # its previous upload has left the pinned buffer
                    for f in [pool.submit(inflate, slot, j, files[c0 + j]) for j in range(cnt)]:
                        f.result()
                    for which, depth_bits, dst in ((0, 16, self.im), (1, 8, self.lb)):
                        dev_stage[slot][which][:cnt].copy_(stage[slot][which][:cnt], non_blocking=True)
                        ops.png_unfilter(dev_stage[slot][which][:cnt], h, w, depth_bits, dst[c0:c0 + cnt], status)
                    if device.type == "cuda":
                        done[slot] = torch.cuda.Event()
                        done[slot].record()
                    k += 1
            if int(status.item()) != 0:
                raise ValueError("a PNG row carries an invalid filter type (corrupt file under {})".format(root / "png"))
        except (ValueError, OSError) as e:
            failure = e
        if world > 1:
            import torch.distributed as dist
            bad = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=device)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)                             # everybody learns of a failure BEFORE the exchange
            if int(bad.item()) and failure is None:
                failure = ValueError("another rank could not decode its share of the slices under {}".format(root / "png"))
        if failure is not None:
            raise failure
        if world > 1:
            for r in range(world):                                                 # every rank's share to everyone (uneven shares: one broadcast each)
                if share[r + 1] > share[r]:
                    dist.broadcast(self.im[share[r]:share[r + 1]].view(torch.uint8), src=r)    # bytes: every backend moves uint8
                    dist.broadcast(self.lb[share[r]:share[r + 1]], src=r)
        self.load_stats = {"slices": n, "decoded_here": hi - lo, "chunk": chunk, "threads": workers,
                           "staging_bytes": 2 * chunk * (im_row + lb_row)}


# ------------------------------------------------------------------------------------------------- sampler
class TrainSampler(object):
    """Which slices, crops and windows make up a training batch -- the sampling POLICY of the reference's python generator
    (input_pipeline.py:285-378) as one vectorised draw per batch.

    The reference yields one sample at a time from nested Python loops and feeds it through tf.data's per-sample map.  Here
    the cases are flattened ONCE into numpy tables (volume extents, liver boxes, a CSR list of tumor slices and of the tumor
    boxes on each of them) and `draw()` emits the whole batch at once: the `[bs, C + 7]` int32 table + `[bs, 2]` windows
    that `unetk_lits_batch` consumes, with no per-sample Python.  The policy (not the random stream) is the reference's:
      * the first ceil(bs * tumor_percent) samples come from cases WITH tumors: a uniformly chosen tumor slice, the crop
        placed around a uniformly chosen tumor box of that slice; they also count as liver samples;
      * then, up to ceil(bs * liver_percent) samples in all, a uniformly chosen slice of the liver's z range, the crop
        placed around the liver box; the rest are uniformly chosen slices, crop anywhere;
      * crop extent = target size x U(random_scale) per axis; the crop's origin is uniform in the range that keeps the
        object box (shrunk by 5 px) inside the crop when that range is more than 20 px wide, else in
        [box start - 20, a point between the box's start and its centre] clipped to the image (:337-352);
      * neighbouring slices fill the other channels, -1 (zeros) outside the volume; window level uniform in
        [10, 50] / [500, 540] HU-units x IM_SCALE when random_window_level else fixed 50 / 500; independent flip coins."""

    def __init__(self, data_list, batch_size, config, liver_percent=0., tumor_percent=0., random_scale=(1., 1.),
                 random_window_level=False, random_flip=0, seed=None):
        self.bs, self.c = int(batch_size), int(config.im_channel)
        self.target = np.array([config.im_height, config.im_width], dtype=np.float64)
        self.scale = (float(random_scale[0]), float(random_scale[1]))
        self.window, self.flip = bool(random_window_level), int(random_flip or 0)
        self.rng = np.random.default_rng(seed)
        d = list(data_list)
        self.pid = np.array([int(c["PID"]) for c in d], dtype=np.int64)
        self.size = np.array([c["size"] for c in d], dtype=np.int64)                       # [n, (depth, H, W)]
        self.liver = np.array([c["bbox"] for c in d], dtype=np.int64)                      # [n, (z0, y0, x0, z1, y1, x1)]
        # CSR: case -> its tumor slices -> the tumor boxes (y0, x0, y1, x1) on each
        n_slices = np.array([len(c["slices"]) for c in d], dtype=np.int64)
        self.slice_ptr = np.concatenate(([0], np.cumsum(n_slices)))
        self.slice_z = np.array([z for c in d for z in c["tumor_slices_index"]], dtype=np.int64)
        n_boxes = np.array([len(bx) for c in d for bx in c["slices"]], dtype=np.int64)
        self.box_ptr = np.concatenate(([0], np.cumsum(n_boxes)))
        self.boxes = np.array([b for c in d for bx in c["slices"] for b in bx], dtype=np.int64).reshape(-1, 4)
        self.tumor_cases = np.flatnonzero(n_slices > 0)
        self.n_tumor = int(math.ceil(self.bs * tumor_percent))
        self.n_liver = max(int(math.ceil(self.bs * liver_percent)), self.n_tumor)          # tumor samples count as liver
        if self.n_tumor and len(self.tumor_cases) == 0:
            raise ValueError("tumor_percent > 0 needs at least one case with tumors")

    def _randint(self, lo, hi):
        """Uniform integers in [lo, hi] (inclusive, element-wise)."""
        if np.any(hi < lo):
            raise ValueError("empty crop range: the crop is larger than the slice")      # random.randint raises there
        return lo + np.floor(self.rng.random(lo.shape) * (hi - lo + 1)).astype(np.int64)

    def draw(self):
        """One batch: dict(case [bs] index into data_list, pid, z, chans [bs, C] (-1 = zeros), box [bs, 4] = (off_y, off_x,
        crop_h, crop_w), clip [bs, 2], flips [bs, 2], kind [bs] 0 tumor / 1 liver / 2 any)."""
        bs, rng = self.bs, self.rng
        j = np.arange(bs)
        kind = np.where(j < self.n_tumor, 0, np.where(j < self.n_liver, 1, 2))
        case = rng.integers(0, len(self.pid), bs)
        if self.n_tumor:
            case[:self.n_tumor] = self.tumor_cases[rng.integers(0, len(self.tumor_cases), self.n_tumor)]
        depth, h, w = self.size[case, 0], self.size[case, 1], self.size[case, 2]
        crop = np.floor(self.target[None, :] * rng.uniform(self.scale[0], self.scale[1], (bs, 2))).astype(np.int64)
        # the slice and the object box (y0, x0, y1, x1) the crop is placed around; "no object" = an inverted box
        z = np.floor(rng.random(bs) * depth).astype(np.int64)
        obj = np.stack([h, w, np.zeros_like(h), np.zeros_like(w)], axis=1)
        lv = kind == 1
        if lv.any():
            lb = self.liver[case[lv]]
            z[lv] = lb[:, 0] + np.floor(rng.random(int(lv.sum())) * (lb[:, 3] - lb[:, 0])).astype(np.int64)
            obj[lv] = lb[:, [1, 2, 4, 5]]
        tm = kind == 0
        if tm.any():
            ct = case[tm]
            s = self.slice_ptr[ct] + np.floor(rng.random(len(ct)) * (self.slice_ptr[ct + 1] - self.slice_ptr[ct])).astype(np.int64)
            z[tm] = self.slice_z[s]
            obj[tm] = self.boxes[self.box_ptr[s] + np.floor(rng.random(len(ct)) * (self.box_ptr[s + 1] - self.box_ptr[s])).astype(np.int64)]
        off = np.empty((bs, 2), dtype=np.int64)
        for ax, extent in ((0, h), (1, w)):
            start, stop, cr = obj[:, ax], obj[:, ax + 2], crop[:, ax]
            lo_in, hi_in = np.maximum(stop + 5 - cr, 0), np.minimum(start - 5, extent - cr)      # box (shrunk by 5) inside the crop
            wide = lo_in + 20 < hi_in
            anchor = np.floor(start * .75 + stop * .25).astype(np.int64) if ax == 0 else (start + stop) // 2
            lo = np.where(wide, lo_in, np.maximum(start - 20, 0))
            hi = np.where(wide, hi_in, np.minimum(anchor, extent - cr))
            off[:, ax] = self._randint(lo, hi)
        left = (self.c - 1) // 2
        chans = z[:, None] + np.arange(-left, self.c - left)[None, :]
        chans = np.where((chans >= 0) & (chans < depth[:, None]), chans, -1)
        if self.window:
            clip = np.stack([rng.integers(10, 51, bs), rng.integers(500, 541, bs)], axis=1).astype(np.float32) * IM_SCALE
        else:
            clip = np.tile(np.array([[50., 500.]], dtype=np.float32) * IM_SCALE, (bs, 1))
        flips = np.stack([(rng.random(bs) < 0.5) & bool(self.flip & 1), (rng.random(bs) < 0.5) & bool(self.flip & 2)], axis=1)
        return dict(case=case, pid=self.pid[case], z=z, chans=chans, box=np.concatenate([off, crop], axis=1), clip=clip,
                    flips=flips.astype(np.int32), kind=kind)

    def table(self, slice_offset):
        """The batch as `unetk_lits_batch` takes it: int32 [bs, C + 7] = (resident-store indices of the C channel slices, of
        the label slice, off_y, off_x, crop_h, crop_w, flip_lr, flip_ud), float32 [bs, 2] windows, int64 [bs] case ids.
        slice_offset: {pid: index of the case's first slice in the resident store}."""
        b = self.draw()
        base = np.array([slice_offset[int(p)] for p in b["pid"]], dtype=np.int64)
        tab = np.empty((self.bs, self.c + 7), dtype=np.int32)
        tab[:, :self.c] = np.where(b["chans"] >= 0, base[:, None] + b["chans"], -1)
        tab[:, self.c] = base + b["z"]
        tab[:, self.c + 1:self.c + 5] = b["box"]
        tab[:, self.c + 5:] = b["flips"]
        return tab, b["clip"], b["pid"]


def batches(store, data_list, config, training, seed=1234, liver_percent=0., tumor_percent=0., random_scale=(1., 1.)):
    """The tf.data pipelines get_dataset_for_train / get_dataset_for_eval_online (:381-430) as a generator of
    (features, labels) device batches: one vectorised sampler draw on the host, everything else in `unetk_lits_batch`."""
    bs = distribution_utils.per_device_batch_size(config.batch_size, config.num_gpus)
    c = config.im_channel
    sampler = TrainSampler(data_list, bs, config, liver_percent, tumor_percent, random_scale if training else (1., 1.),
                           random_window_level=training, random_flip=(getattr(config, "random_flip", 0) if training else 0),
                           seed=seed)
    step = 0
    while True:
        tab, clip, names = sampler.table(store.offset)
        images, labels = ops.lits_batch(store.im, store.lb, torch.from_numpy(tab).to(store.device),
                                        torch.from_numpy(clip).to(store.device), (config.im_height, config.im_width), c,
                                        LB_SCALE, float(config.noise_scale) if training else 0.0, seed * 7919 + step)
        step += 1
        yield {"images": images, "names": torch.from_numpy(names)}, labels


def batches_eval_3d(store, data_list, config):
    """--eval_3d online evaluation (input_pipeline_g.py:602-700 `gen_eval_3d_online_batch` + :796-833): every validation
    case is served ONCE, as consecutive batch_size-slice batches over the liver's z range [z1, z2) -- the last batch padded
    with empty slices (index -1: zero image, zero label) -- each slice cropped to the liver box (y1, x1, y2 - y1, x2 - x1) and
    resized to the network size, fixed window, no noise, no flips; `names` carries the case id of the batch.
    The whole table of a case is laid out in one vectorised step; the gather kernel does the rest."""
    bs = distribution_utils.per_device_batch_size(config.batch_size, config.num_gpus)
    c = config.im_channel
    left = (c - 1) // 2
    ctx = np.arange(-left, c - left)[None, :]
    clip_row = np.array([[50., 500.]], dtype=np.float32) * IM_SCALE
    for case in data_list:
        z1, y1, x1, z2, y2, x2 = case["bbox"]
        depth, pid = case["size"][0], int(case["PID"])
        z = np.concatenate((np.arange(z1, z2), np.full((-(z2 - z1)) % bs, -100)))          # -100: padding (as the reference)
        chans = z[:, None] + ctx
        ok = (z[:, None] >= 0) & (chans >= 0) & (chans < depth)
        off = store.offset[pid]
        tab = np.zeros((len(z), c + 7), dtype=np.int32)
        tab[:, :c] = np.where(ok, off + chans, -1)
        tab[:, c] = np.where(z >= 0, off + z, -1)
        tab[:, c + 1:c + 5] = [y1, x1, y2 - y1, x2 - x1]
        for b0 in range(0, len(z), bs):
            t = torch.from_numpy(np.ascontiguousarray(tab[b0:b0 + bs])).to(store.device)
            clip = torch.from_numpy(np.repeat(clip_row, bs, axis=0)).to(store.device)
            images, labels = ops.lits_batch(store.im, store.lb, t, clip, (config.im_height, config.im_width), c, LB_SCALE, 0.0, 0)
            yield {"images": images, "names": torch.full((bs,), pid, dtype=torch.int64)}, labels


def input_fn(mode, params):
    """input_pipeline.py:199-241 for the modes train / eval_online; params["lits_root"] holds png/, meta.json,
    k_folds.txt.  Offline volume evaluation: `input_fn_eval` below."""
    args = params["args"]
    root = params["lits_root"]
    device = params.get("device", torch.device("cuda", torch.cuda.current_device()))
    key = ("lits_store", mode == "train")
    if key not in params:
        cases = collect_datasets(root, args.test_fold, "train" if mode == "train" else "val",
                                 filter_tumor_size=getattr(args, "filter_size", 0))
        params[key] = (SliceStore(root, cases, device, strategy=params.get("strategy")), cases)
    store, cases = params[key]
    if len(cases) == 0:
        raise ValueError("No valid dataset found!")
    rs = tuple(getattr(args, "zoom_scale", (1., 1.)))                        # --zoom_scale (input_pipeline.py:62)
    base_seed = int(getattr(args, "seed", 1234) or 1234)
    seed = base_seed + 1000 * int(params.get("rank", 0))                     # training: every replica its own shard of the stream
    if mode == "train":
        return batches(store, cases, args, True, seed, getattr(args, "liver_percent", 0.), getattr(args, "tumor_percent", 0.), rs)
    if mode == "eval_online" and getattr(args, "eval_3d", False):
        return batches_eval_3d(store, cases, args)
    if mode == "eval_online":
        # rank-INDEPENDENT stream: the reference (one process) evaluates one sample, and the evaluator hooks let rank 0 alone
        # decide / save -- every replica must see the same validation data so their best-result state stays identical
        gen = batches(store, cases, args, False, base_seed + 500, getattr(args, "liver_percent", 0.),
                      getattr(args, "tumor_percent", 0.))
        n = int(getattr(args, "eval_num_batches_per_epoch", 100))
        return (next(gen) for _ in range(n))
    raise ValueError("lits.input_fn handles the modes `train` and `eval_online`, got {}".format(mode))


def input_fn_eval(mode, params):
    """input_pipeline.py:228-234 for ModeKeys.EVAL / PREDICT: a python generator over the NIfTI volumes of the validation
    fold (paths in meta.json's vol_case / lab_case, relative to params["proj_root"]).  No device work here: the
    evaluator uploads each slab and does mirroring / accumulation / argmax on the GPU (evaluators/evaluator_liver.py)."""
    args = params["args"]
    cases = collect_datasets(params["lits_root"], args.test_fold, mode, filter_tumor_size=getattr(args, "filter_size", 0),
                             filter_only_liver_in_val=params.get("filter_only_liver_in_val", True))
    if len(cases) == 0:
        raise ValueError("No valid dataset found!")
    proj_root = params.get("proj_root", ".")
    if getattr(args, "eval_in_patches", False):
        return get_dataset_for_eval_patches(cases, config=args, proj_root=proj_root)
    if params.get("whole_slices", False):
        return get_dataset_for_eval_image(cases, args, proj_root)
    return get_dataset_for_eval_image_v2(cases, args, proj_root)


# ------------------------------------------------------------------------------------------------- offline evaluation
GRAY_MIN, GRAY_MAX = -200, 250       # input_pipeline.py:45-46


def cv2_resize_linear(img, dsize):
    """cv2.resize(img, dsize, interpolation=cv2.INTER_LINEAR) for float arrays [H, W] or [H, W, C]; dsize = (width, height).
    OpenCV's rule: source coordinate (dst + 0.5) * (src / dst) - 0.5, taps clamped to the image (border replicate)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    img = np.asarray(img, np.float32)
    sh, sw = img.shape[:2]

    def taps(dst_n, src_n):
        s = (np.arange(dst_n, dtype=np.float64) + 0.5) * (src_n / float(dst_n)) - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = (s - i0).astype(np.float32)
        lo = i0 < 0
        i0[lo], f[lo] = 0, 0.0
        hi = i0 >= src_n - 1
        i0[hi], f[hi] = src_n - 1, 0.0
        return i0, np.minimum(i0 + 1, src_n - 1), f

    if (sh, sw) == (dh, dw):
        return img.copy()
    y0, y1, fy = taps(dh, sh)
    x0, x1, fx = taps(dw, sw)
    ex = (Ellipsis,) if img.ndim == 2 else (Ellipsis, None)
    rows = img[y0] * (1.0 - fy)[(slice(None), None) + ((None,) if img.ndim == 3 else ())] + \
        img[y1] * fy[(slice(None), None) + ((None,) if img.ndim == 3 else ())]
    fxb = fx[(None, slice(None)) + ((None,) if img.ndim == 3 else ())]
    del ex
    return (rows[:, x0] * (1.0 - fxb) + rows[:, x1] * fxb).astype(np.float32)


def _grown_span(lo, hi, pad, extent, align, at_least=0):
    """One axis of the evaluation window: the object's half-open span [lo, hi) grown by `pad`, clipped to [0, extent),
    widened to `at_least` pixels if shorter, then its length rounded UP to a multiple of `align` about the same centre.
    Returns (start, stop, wanted length); stop - start < wanted length when the far border clipped it."""
    a, b = max(lo - pad, 0), min(hi + pad, extent)
    if at_least:
        if at_least > extent:
            raise ValueError("Cannot satisfied conditions!")
        if b - a < at_least:
            a = min(max(a - (at_least - (b - a)) // 2, 0), extent - at_least)
            b = a + at_least
    want = -(-(b - a) // align) * align
    start = max(int((a + b - 1) / 2 - (want - 1) / 2), 0)
    return start, min(start + want, extent), want


def aligned_window(case, align, padding, min_shape=None):
    """The (y, x) window of a case that the evaluators cut from every slice: the liver box + `padding`, at least `min_shape`,
    sides rounded up to `align` (input_pipeline.py:444-479,556-580).  When the far border clips EITHER axis so that its
    side is no longer a multiple of `align`, BOTH axes are re-anchored at their far ends (the reference's rule; it warns
    when that pushes a start below zero).  Returns (y1, y2, x1, x2)."""
    _, h, w = case["size"]
    bb = case["bbox"]
    spans = [_grown_span(bb[1], bb[4], padding, h, align, min_shape[0] if min_shape else 0),
             _grown_span(bb[2], bb[5], padding, w, align, min_shape[1] if min_shape else 0)]
    if any((stop - start) % align for start, stop, _ in spans):
        spans = [(stop - want, stop, want) for _, stop, want in spans]
        if any(start < 0 for start, _, _ in spans):
            print("\nWarning: bbox aligns with {} failed! point1 ({}, {}) point2 ({}, {})\n".format(
                align, spans[1][0], spans[0][0], spans[1][1], spans[0][1]))
    return spans[0][0], spans[0][1], spans[1][0], spans[1][1]


def _load_lits_volume(case, root, test_data=False):
    from . import nii_kits
    obj_num = int(case["vol_case"][:-4].split("-")[-1])
    if test_data:
        return obj_num, nii_kits.read_nii(root / case["vol_case"])[1]
    return obj_num, nii_kits.read_lits(obj_num, "vol", root / case["vol_case"])[1]


def _window_normalise(volume):
    """HU window [GRAY_MIN, GRAY_MAX] -> [0, 1], (z, y, x) -> (y, x, z) float32 (input_pipeline.py:595-596)."""
    volume = (np.clip(volume, GRAY_MIN, GRAY_MAX) - GRAY_MIN) / (GRAY_MAX - GRAY_MIN)
    return volume.transpose((1, 2, 0)).astype(np.float32)


def parse_case_eval(case, align, padding, padding_z, im_channel, parse_label=True, test_data=False, proj_root="."):
    """input_pipeline.py:556-612: the liver box (+ padding, sides rounded up to `align`) of one NIfTI case ->
    normalised float32 volume (y, x, z) with the half-channel context slices, cropped uint8 segmentation (z, y, x)."""
    from . import nii_kits
    d, h, w = case["size"]
    z1, z2 = max(case["bbox"][0] - padding_z, 0), min(case["bbox"][3] + padding_z, d)
    y1, y2, x1, x2 = aligned_window(case, align, padding)
    root = Path(proj_root)
    obj_num, volume = _load_lits_volume(case, root, test_data)
    lhc = (im_channel - 1) // 2                       # context slices before / after the centre slice of a sample
    rhc = im_channel - 1 - lhc
    lo, hi = z1 - lhc, z2 + rhc                       # zero slices stand in for context beyond the volume
    volume = np.pad(volume[max(lo, 0):min(hi, d), y1:y2, x1:x2], ((max(-lo, 0), max(hi - d, 0)), (0, 0), (0, 0)))
    cshape = list(volume.shape)
    volume = _window_normalise(volume)
    segmentation, lab_case = None, None
    if parse_label:
        _, segmentation = nii_kits.read_lits(obj_num, "lab", root / case["lab_case"])
        segmentation = segmentation.astype(np.uint8)[z1:z2, y1:y2, x1:x2]
        lab_case = case["lab_case"]
    bbox = [x1, y1, z1, x2 - 1, y2 - 1, z2 - 1]
    return case["PID"], case["vol_case"], lab_case, bbox, [d, h, w], cshape, lhc, rhc, volume, segmentation


def _mirrored(eval_batch, config):
    """The mirrored copies of a slab under --eval_mirror (input_pipeline.py:538-553): variant 1 = flipped along W when
    random_flip has bit 0, 2 = along H with bit 1, 3 = both -- the last one whenever `random_flip & 3 > 0`, i.e. also for
    random_flip 1 and 2 (the reference's test, kept literally; evaluators.mirror_plan averages accordingly)."""
    if not getattr(config, "eval_mirror", False):
        return
    rf = int(config.random_flip)
    for variant, wanted, axes in ((1, rf & 1, (2,)), (2, rf & 2, (1,)), (3, rf & 3, (1, 2))):
        if wanted > 0:
            yield dict(eval_batch, images=np.flip(eval_batch["images"], axis=axes), mirror=variant), None


def _slabs(volume, batch_size, lhc, rhc, head, config):
    """Serve a (y, x, z) volume as batch_size-slice slabs with (lhc, rhc) context slices per sample, each followed by its
    mirrored copies.  `head` = the constant entries of every slab's feature dict."""
    n = volume.shape[-1] - lhc - rhc
    assert n % batch_size == 0, "Wrong padding"
    win = np.lib.stride_tricks.sliding_window_view(volume, lhc + rhc + 1, axis=-1)      # [y, x, n, c] without copying
    for idx in range(0, n, batch_size):
        slab = dict(head, images=np.ascontiguousarray(np.moveaxis(win[:, :, idx:idx + batch_size], 2, 0)), mirror=0)
        yield slab, None
        for item in _mirrored(slab, config):
            yield item


def get_dataset_for_eval_image_v2(data_list, config, proj_root="."):
    """input_pipeline.py:615-668: per case the liver box is cut from the NIfTI volume, padded in z to whole batches,
    resized to the network size, and served as batch_size-slice slabs (each followed by its mirrored copies under
    --eval_mirror); a case ends with (None, (segmentation, vol_path, pads, bbox, resize))."""
    align = 16 if getattr(config, "model", "UNet") != "DenseUNet" else 32
    padding, padding_z = 25, 0
    batch_size = config.batch_size
    c = config.im_channel
    pshape = config.im_height, config.im_width
    resize = not (config.im_height <= 0 or config.im_width <= 0)
    for case in data_list[getattr(config, "eval_skip_num", 0):]:
        pid, vol_path, _, bbox, _, cshape, lhc, rhc, volume, segmentation = parse_case_eval(
            case, align, padding, padding_z, c, parse_label=getattr(config, "mode", "eval") != "infer", proj_root=proj_root)
        if not resize:
            pshape = tuple(cshape[1:])
        pads = -(bbox[5] - bbox[2] + 1) % batch_size          # zero slices that complete the last slab
        volume = np.pad(volume, ((0, 0), (0, 0), (0, pads)))
        if resize:
            volume = cv2_resize_linear(volume, pshape)      # dsize = (im_height, im_width), as the reference passes it
        for item in _slabs(volume, batch_size, lhc, rhc, {"names": pid}, config):
            yield item
        yield None, (segmentation, vol_path, pads, bbox, resize)


def get_dataset_for_eval_image(data_list, config, proj_root=".", test_data=False):
    """input_pipeline_li.py:398-456: whole slices (no liver crop)."""
    from . import nii_kits
    batch_size = config.batch_size
    c = config.im_channel
    pshape = config.im_height, config.im_width
    resize = not (config.im_height <= 0 or config.im_width <= 0)
    root = Path(proj_root)
    parse_label = getattr(config, "mode", "eval") != "infer"
    lhc = (c - 1) // 2
    rhc = c - 1 - lhc
    for case in data_list[getattr(config, "eval_skip_num", 0):]:
        obj_num, volume = _load_lits_volume(case, root, test_data)
        volume = _window_normalise(volume)
        segmentation, seg_path = None, None
        if parse_label:
            _, segmentation = nii_kits.read_lits(obj_num, "lab", root / case["lab_case"])
            segmentation, seg_path = segmentation.astype(np.uint8), case["lab_case"]
        h, w, ori_d = volume.shape
        pads = -ori_d % batch_size
        volume = np.pad(volume, ((0, 0), (0, 0), (lhc, pads + rhc)))       # context + the zero slices of the last slab
        if resize:
            volume = cv2_resize_linear(volume, pshape)
        for item in _slabs(volume, batch_size, lhc, rhc, {"names": case["PID"]}, config):
            yield item
        yield None, (segmentation, seg_path, pads, (0, 0, 0, w - 1, h - 1, ori_d - 1), resize)


def parse_case_patches(case, align, padding, padding_z, min_shape=None):
    """input_pipeline.py:444-479: the liver box (+ padding) grown to at least `min_shape` (the patch) inside the slice,
    then its sides rounded up to `align` around the centre."""
    d, h, w = case["size"]
    z1, z2 = max(case["bbox"][0] - padding_z, 0), min(case["bbox"][3] + padding_z, d)
    y1, y2, x1, x2 = aligned_window(case, align, padding, min_shape)
    return case["PID"], d, h, w, z1, y1, x1, z2, y2, x2


def patch_centres(extent, psize, step=2):
    """input_pipeline.py:725-731 along one axis: window centres from psize/2 to extent - psize/2 in equal steps of at most
    psize/step, rounded -- the windows tile [0, extent) with overlap, first and last flush with the borders."""
    start, end = psize // 2, extent - psize // 2
    num = math.ceil((end - start) / (psize / step))
    size = (end - start) / (num + 1e-8)
    if size == 0:
        size = 9999999
    return np.round(np.arange(start, end + 1e-8, size)).astype(np.int32)


def get_dataset_for_eval_patches(data_list, step=2, config=None, proj_root="."):
    """--eval_in_patches, input_pipeline.py:676-766: the liver box of each case is evaluated at NATIVE resolution by
    sliding an (im_height, im_width) window over every slice (window stride <= patch / step), batched batch_size windows
    at a time.  Yields (eval_batch, None) ... and, with the case's last batch, (eval_batch, labels) -- the whole label
    volume (z, y, x); eval_batch = {images [bs, ph, pw, c], name, pad, bbox, position[(z, lb_y, ub_y, lb_x, ub_x)]}.
    The reference indexes x windows with the patch HEIGHT (:743-744); identical for the square patches it is used with,
    restated here with the width."""
    import itertools
    from . import nii_kits
    align, padding, padding_z = 16, 25, 0
    batch_size = config.batch_size
    c = config.im_channel
    psize = int(config.im_height), int(config.im_width)
    root = Path(proj_root)
    for case in data_list[getattr(config, "eval_skip_num", 0):]:
        pid, d, h, w, z1, y1, x1, z2, y2, x2 = parse_case_patches(case, align, padding, padding_z, min_shape=psize)
        obj_num = int(case["vol_case"][:-4].split("-")[-1])
        _, volume = nii_kits.read_lits(obj_num, "vol", root / case["vol_case"])
        lhc = (c - 1) // 2
        rhc = c - 1 - lhc
        left_pad = lhc - z1 if z1 < lhc else 0
        right_pad = z2 + rhc - d if z2 + rhc > d else 0
        volume = volume[max(0, z1 - lhc):min(d, z2 + rhc), y1:y2, x1:x2]
        cd, ch, cw = volume.shape
        if left_pad > 0 or right_pad > 0:
            volume = np.concatenate((np.zeros((left_pad, ch, cw), dtype=volume.dtype), volume,
                                     np.zeros((right_pad, ch, cw), dtype=volume.dtype)), axis=0)
            cd, ch, cw = volume.shape
        volume = (np.clip(volume, GRAY_MIN, GRAY_MAX) - GRAY_MIN) / (GRAY_MAX - GRAY_MIN)
        volume = volume.transpose((1, 2, 0)).astype(np.float32)     # (y, x, z)
        ysteps, xsteps = patch_centres(ch, psize[0], step), patch_centres(cw, psize[1], step)
        all_patches = list(itertools.product(xsteps, ysteps, range(lhc, cd - rhc)))
        num_of_batches = (len(all_patches) + (batch_size - 1)) // batch_size
        for batch in range(num_of_batches):
            eval_batch = {"images": np.zeros((batch_size,) + psize + (c,), dtype=np.float32), "name": pid, "pad": 0,
                          "bbox": [x1, y1, z1, x2 - 1, y2 - 1, z2 - 1], "position": [None] * batch_size}
            chunk = all_patches[batch * batch_size:(batch + 1) * batch_size]
            for i, (x, y, z) in enumerate(chunk):
                lb_y, ub_y = y - psize[0] // 2, y + psize[0] // 2
                lb_x, ub_x = x - psize[1] // 2, x + psize[1] // 2
                eval_batch["images"][i] = volume[lb_y:ub_y, lb_x:ub_x, z - lhc:z + rhc + 1]
                eval_batch["position"][i] = (z - lhc, lb_y, ub_y, lb_x, ub_x)
            if batch < num_of_batches - 1:
                yield eval_batch, None
            else:
                eval_batch["pad"] = batch_size - len(chunk)
                _, labels = nii_kits.read_lits(obj_num, "lab", root / case["lab_case"])
                yield eval_batch, labels
