"""Synthetic LiTS-shaped input_fn: same tensor contract as the reference's
DataLoader/Liver/input_pipeline.py:243-284 -- features["images"] f32 [bs,H,W,C] (window-normalised CT in
[0,1] plus U(-noise, noise)), features["names"] int PIDs, labels int32 [bs,H,W] in {0..ncls-1}.

images ~ U[0,1) + U(-0.05, 0.05); labels: class 1 = filled ellipse centred (0.45H, 0.4W), radii
(0.3H, 0.25W) (~24 % area, liver-like); class 2 = disk radius 0.05H inside it (~0.8 %, tumor-like);
per-sample random shift of +-0.1H.  Generated once with numpy default_rng(seed) and kept resident on
the device; no file I/O.
"""
import numpy as np
import torch


def add_arguments(parser):
    """The liver pipeline flags the model reads (DataLoader/Liver/input_pipeline.py:54-70)."""
    group = parser.add_argument_group(title="Input Pipeline Arguments")
    group.add_argument("--test_fold", type=int, default=2)
    group.add_argument("--im_depth", type=int, default=10, help="UNet3D patch depth (DataLoader/NF/input_pipeline_3d.py:56)")
    group.add_argument("--im_height", type=int, default=256)
    group.add_argument("--im_width", type=int, default=256)
    group.add_argument("--im_channel", type=int, default=3)
    group.add_argument("--noise_scale", type=float, default=0.1)
    group.add_argument("--random_flip", type=int, default=1)
    group.add_argument("--eval_num_batches_per_epoch", type=int, default=100)
    # guided pipeline flags the GUNet plugin reads (DataLoader/Liver/input_pipeline_g.py:71-125; --guide_channel
    # comes from the NF pipelines, DataLoader/NF/input_pipeline_g_simply.py:106 -- the liver pipeline emits 1 channel)
    group.add_argument("--use_spatial", action="store_true")
    group.add_argument("--use_context", action="store_true")
    group.add_argument("--side_dropout", type=float, default=0.5)
    group.add_argument("--use_se", action="store_true")
    group.add_argument("--guide_channel", type=int, default=1)
    group.add_argument("--synthetic_batches", type=int, default=2, help="distinct synthetic batches kept on device")
    group.add_argument("--seed", type=int, default=1234)


def make_batch(bs, height, width, channel, num_classes, seed=1234, noise_scale=0.05):
    rng = np.random.default_rng(seed)
    images = rng.random((bs, height, width, channel), dtype=np.float32)
    images += rng.uniform(-noise_scale, noise_scale, size=images.shape).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    labels = np.zeros((bs, height, width), dtype=np.int32)
    for b in range(bs):
        dy, dx = rng.uniform(-0.1, 0.1, size=2) * height
        cy, cx = 0.45 * height + dy, 0.4 * width + dx
        ell = ((yy - cy) / (0.3 * height)) ** 2 + ((xx - cx) / (0.25 * width)) ** 2 <= 1.0
        labels[b][ell] = 1
        if num_classes > 2:
            ty, tx = cy + 0.1 * height, cx - 0.05 * width
            disk = (yy - ty) ** 2 + (xx - tx) ** 2 <= (0.05 * height) ** 2
            labels[b][disk & ell] = 2
    names = np.arange(bs, dtype=np.int64) + seed
    return images, labels, names


def make_batch_3d(bs, depth, height, width, channel, num_classes, seed=1234):
    """NF-style 3-D patches (DataLoader/NF/input_pipeline_3d.py): z-scored images ~ N(0,1) (the NF pipeline
    z-scores, input_pipeline_g_simply.py:436-442), labels = ellipsoid of ~3 % of the voxels (class 1)."""
    rng = np.random.default_rng(seed)
    images = rng.standard_normal((bs, depth, height, width, channel)).astype(np.float32)
    zz, yy, xx = np.meshgrid(np.arange(depth, dtype=np.float32), np.arange(height, dtype=np.float32),
                             np.arange(width, dtype=np.float32), indexing="ij")
    labels = np.zeros((bs, depth, height, width), dtype=np.int32)
    for b in range(bs):
        cz, cy, cx = (rng.uniform(0.35, 0.65, size=3) * np.array([depth, height, width])).astype(np.float32)
        ell = ((zz - cz) / max(0.3 * depth, 1.0)) ** 2 + ((yy - cy) / (0.2 * height)) ** 2 + \
            ((xx - cx) / (0.2 * width)) ** 2 <= 1.0
        labels[b][ell] = 1 if num_classes == 2 else int(rng.integers(1, num_classes))
    names = np.arange(bs, dtype=np.int64) + seed
    return images, labels, names


def input_fn_3d(mode, params):
    """input_fn for --model UNet3D (`nf_3d` sub-command of entry/main.py:53-77): features["images"] f32
    [bs,D,H,W,C], labels int32 [bs,D,H,W]."""
    args = params["args"]
    num_gpus = max(getattr(args, "num_gpus", 1), 1)
    bs = args.batch_size // num_gpus if num_gpus > 1 else args.batch_size
    ncls = len(args.classes) + 1
    rank = int(params.get("rank", 0))
    device = params.get("device", torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                        else torch.device("cpu"))
    nb = max(int(getattr(args, "synthetic_batches", 2)), 1)
    base_seed = int(getattr(args, "seed", 1234) or 1234)
    pool = []
    for i in range(nb):
        seed = base_seed + 1000 * rank + i + (0 if mode == "train" else 500)
        images, labels, names = make_batch_3d(bs, args.im_depth, args.im_height, args.im_width, args.im_channel, ncls, seed)
        pool.append(({"images": torch.from_numpy(images).to(device), "names": torch.from_numpy(names)},
                     torch.from_numpy(labels).to(device)))

    def gen():
        i = 0
        n = None if mode == "train" else int(getattr(args, "eval_num_batches_per_epoch", nb) or nb)
        while n is None or i < n:
            yield pool[i % nb]
            i += 1

    return gen()


def input_fn_eval_volumes(mode, params):
    """Eval generator with the contract of DataLoader/Liver/input_pipeline_li.py:398-456 on synthetic cases:
    `(features, None)` slabs of batch_size slices -- features["images"] f32 [bs,H,W,C] on the device,
    features["names"] PID (+ "mirror" and mirrored copies iff params["pipeline_mirror"], the reference's host-side
    TTA) -- then `(None, (segmentation [D,H,W] uint8, vol_path, pads, bbox, resize))` per case.  The last slab of a
    case is zero-padded to the batch size (pads).  params["eval_cases"] = [(pid, depth), ...]."""
    args = params["args"]
    bs, h, w, c = args.batch_size, args.im_height, args.im_width, args.im_channel
    ncls = len(args.classes) + 1
    device = params.get("device", torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                        else torch.device("cpu"))
    cases = params.get("eval_cases") or [(1000 + i, 2 * bs + 3) for i in range(2)]
    rf = int(getattr(args, "random_flip", 0) or 0)
    for pid, depth in cases:
        images, labels, _ = make_batch(depth, h, w, c, ncls, int(getattr(args, "seed", 1234) or 1234) + int(pid))
        pads = (bs - depth % bs) % bs
        if pads:
            images = np.concatenate((images, np.zeros((pads, h, w, c), np.float32)))
        for i in range(0, depth + pads, bs):
            feats = {"images": torch.from_numpy(images[i:i + bs]).to(device), "names": pid}
            if params.get("pipeline_mirror"):
                feats["mirror"] = 0
                yield feats, None
                if getattr(args, "eval_mirror", False):
                    for m, axes in ((1, (2,)), (2, (1,)), (3, (2, 1))):
                        if rf & m > 0:                      # m = 3: rf & 3 (input_pipeline_li.py:451)
                            flipped = np.ascontiguousarray(np.flip(images[i:i + bs], axis=axes))
                            yield {"images": torch.from_numpy(flipped).to(device), "names": pid, "mirror": m}, None
            else:
                yield feats, None
        yield None, (labels.astype(np.uint8), "synthetic-{}".format(pid), pads, (0, 0, 0, w - 1, h - 1, depth - 1), True)


def make_guide(labels, guide_channel=1, seed=1234):
    """Spatial guide like DataLoader/Liver/input_pipeline_g.py:382-394 / utils/image_ops.py:431-434:
    g/2 + 0.5 with g = max_k exp(-|p - c_k|^2 / (2 sigma^2)), 1-3 centres inside the foreground, sigma ~ U(2, 8)
    (pixels, scaled with the image).  Returns float32 [bs, H, W, guide_channel]."""
    rng = np.random.default_rng(seed + 77)
    bs, height, width = labels.shape
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    out = np.zeros((bs, height, width, guide_channel), dtype=np.float32)
    fg_cls = labels.max()
    for b in range(bs):
        ys, xs = np.nonzero(labels[b] == fg_cls) if fg_cls > 0 else (np.array([height // 2]), np.array([width // 2]))
        if len(ys) == 0:
            ys, xs = np.array([height // 2]), np.array([width // 2])
        for ch in range(guide_channel):
            g = np.zeros((height, width), dtype=np.float32)
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, len(ys)))
                sigma = rng.uniform(2.0, 8.0) * height / 256.0 + 0.5
                g = np.maximum(g, np.exp(-((yy - ys[k]) ** 2 + (xx - xs[k]) ** 2) / (2 * sigma ** 2)))
            out[b, ..., ch] = g / 2 + 0.5
    return out


def input_fn(mode, params):
    """input_fn(mode, params) -> iterator of (features, labels), modes train / eval_online / eval
    (reference contract: DataLoader/Liver/input_pipeline.py:199-203)."""
    args = params["args"]
    num_gpus = max(getattr(args, "num_gpus", 1), 1)
    bs = args.batch_size // num_gpus if num_gpus > 1 else args.batch_size
    ncls = len(args.classes) + 1
    rank = int(params.get("rank", 0))
    device = params.get("device", torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                        else torch.device("cpu"))
    nb = max(int(getattr(args, "synthetic_batches", 2)), 1)
    base_seed = int(getattr(args, "seed", 1234) or 1234)
    pool = []
    for i in range(nb):
        seed = base_seed + 1000 * rank + i + (0 if mode == "train" else 500)
        images, labels, names = make_batch(bs, args.im_height, args.im_width, args.im_channel, ncls, seed,
                                           getattr(args, "noise_scale", 0.05))
        feats = {"images": torch.from_numpy(images).to(device), "names": torch.from_numpy(names)}
        if getattr(args, "use_spatial", False):
            feats["sp_guide"] = torch.from_numpy(make_guide(labels, int(getattr(args, "guide_channel", 1)), seed)).to(device)
        pool.append((feats, torch.from_numpy(labels).to(device)))

    def gen():
        if mode == "train":
            i = 0
            while True:
                yield pool[i % nb]
                i += 1
        else:
            n = int(getattr(args, "eval_num_batches_per_epoch", nb) or nb)
            for i in range(min(n, nb) if mode == "eval" else n):
                yield pool[i % nb]

    return gen()
