"""Debug helper: per-unit backward check of a SmallUNet step (which unit, how many elements differ)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_gunet import make_args
from boxsegliver_amd import ops
from boxsegliver_amd.core import models
from boxsegliver_amd.data.synthetic import make_batch, make_guide
from oracle import tf_ops

YML = dict(init_channel_factor=1, num_pool_layers=3, ret_prob=False, ret_pred=True, build_metrics=True, build_summaries=False)
args = make_args(normalizer="batch_norm", loss_type="xentropy", use_spatial=True, guide_channel=1, im_height=64, im_width=64)
images, labels, _ = make_batch(2, 64, 64, 3, 3, 1234)
guide = make_guide(labels, 1, 1234)
zoo = {c.__name__: c for c in models.MODEL_ZOO}
model = zoo["SmallUNet"](args)
inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda(), "sp_guide": torch.from_numpy(guide).cuda()}
model(inputs, "eval", **YML)
gen = torch.Generator().manual_seed(6)
params = {}
for name, t in model.params.state_dict().items():
    kind = model.params.where[name][4]
    if kind == "gamma":
        params[name] = 0.5 + torch.rand(t.shape, generator=gen)
    elif kind in ("beta", "bias"):
        params[name] = 0.2 * torch.randn(t.shape, generator=gen)
    else:
        params[name] = t.clone()
model.params.load_state(params)
ops.DEBUG_CAPTURE = []
model.params.zero_grad()
model(inputs, "train", **YML).backward()
torch.cuda.synchronize()
for i, c in enumerate(ops.DEBUG_CAPTURE):
    if c.get("kind") in ("deconv", "conv3d"):
        print(i, c.get("kind"), tuple(c["x"].shape))
        continue
    y = c["y"].detach().cpu().double().requires_grad_(True)
    g = c["gamma"].detach().cpu().double()
    b = c["beta"].detach().cpu().double()
    z, _, _ = tf_ops.batch_norm(y, g, b, torch.zeros(y.shape[-1], dtype=torch.float64), torch.ones(y.shape[-1], dtype=torch.float64), True)
    torch.relu(z).backward(c["dz"].detach().cpu().double())
    err = (c["dy"].cpu().double() - y.grad).abs()
    scale = y.grad.abs().max()
    print(i, "unit", tuple(c["x"].shape), "->", y.shape[-1], "dil", c.get("dilation"), "dz contiguous", c["dz"].is_contiguous(), tuple(c["dz"].stride()),
          "max rel %.3e" % float(err.max() / scale), "bad elems", int((err > 1e-4 * scale).sum()), "of", err.numel(),
          "min |u| at bad:", float(z.detach().abs()[err > 1e-4 * scale].min()) if (err > 1e-4 * scale).any() else None)
