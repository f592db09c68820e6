"""Shared by make_golden.py (generator) and tests/test_golden*.py (consumers): deterministic parameter
construction from the oracle's variable specs and the three whole-net cases of SURVEY.md 8c item (8).

A fixture holds DATA only: seeded inputs, and the outputs of this repo's CPU oracle evaluated in float64
(logits, losses, per-variable gradient norms, a few small gradients in full, thresholded / argmax masks,
moving statistics after one step, a 3-step TF-Adam loss trajectory).  Parameters are rebuilt from
`build_params` (numpy default_rng, independent of torch's RNG stream) and pinned in the fixture by a
per-variable checksum (sum, sum of squares) so a drift of the generator is detected rather than hidden.
"""
import os
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = OrderedDict([
    # name -> (net kind, normalizer, loss_type, weight type, size)
    ("unet_bn_xent", dict(kind="UNet", normalizer="batch_norm", loss_type="xentropy", w_type="numerical", size=32)),
    ("unet_in_dice", dict(kind="UNet", normalizer="instance_norm", loss_type="dice", w_type="none", size=32)),
    ("gunet_in_xent", dict(kind="GUNet", normalizer="instance_norm", loss_type="xentropy", w_type="numerical", size=32)),
    ("unet3d_in_xent", dict(kind="UNet3D", normalizer="instance_norm", loss_type="xentropy", w_type="numerical", size=32)),
    # LGNet.yml (leaky-ReLU guide on encoder / decoder levels 0, 1) and SmallUNet.yml (stride-2 + atrous convs, guide concatenated)
    ("lgnet_in_xent", dict(kind="LGNet", normalizer="instance_norm", loss_type="xentropy", w_type="numerical", size=32,
                           mod_layers=[[0, 1], [0, 1]])),
    ("smallunet_bn_xent", dict(kind="SmallUNet", normalizer="batch_norm", loss_type="xentropy", w_type="numerical", size=64,
                               factor=1)),
])
GUIDED = ("GUNet", "LGNet", "SmallUNet")
NUMERIC_W = {"UNet": [0.2, 0.4, 4.4], "GUNet": [0.2, 0.4, 4.4], "UNet3D": [1.0, 1.0], "LGNet": [0.2, 0.4, 4.4],
             "SmallUNet": [0.2, 0.4, 4.4]}
WD = {"UNet": 1e-5, "GUNet": 1e-5, "UNet3D": 3e-5, "LGNet": 1e-5, "SmallUNet": 1e-5}
LR = 1e-3


def build_params(specs, seed):
    """specs: [(tf_name, shape, kind)] -> OrderedDict name -> float32 ndarray.  Weights ~ U(-l, l) with the
    Glorot limit on the TF shape (receptive field x in / out), gamma ~ 0.5 + U(0,1), beta/bias ~ 0.2 N(0,1),
    moving_mean 0, moving_var 1.  One child generator per variable (order-independent)."""
    out = OrderedDict()
    for i, (name, shape, kind) in enumerate(specs):
        rng = np.random.default_rng([seed, i])
        shape = tuple(int(s) for s in shape)
        if "/spatial/" in name and kind == "conv_w":
            v = 0.5 * rng.standard_normal(shape)            # guide 1x1 weights: large enough to matter
        elif kind in ("conv_w", "deconv_w"):
            rf = int(np.prod(shape[:-2]))
            lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
            v = rng.uniform(-lim, lim, size=shape)
        elif kind == "gamma":
            v = 0.5 + rng.random(shape)
        elif kind in ("beta", "bias"):
            v = 0.2 * rng.standard_normal(shape)
        elif kind == "moving_var":
            v = np.ones(shape)
        elif kind == "moving_mean":
            v = np.zeros(shape)
        else:
            v = 0.5 * rng.standard_normal(shape)            # any other trainable (e.g. guide 1x1 weights)
        out[name] = v.astype(np.float32)
    return out


def checksum(params):
    return np.array([[float(v.astype(np.float64).sum()), float((v.astype(np.float64) ** 2).sum())]
                     for v in params.values()], dtype=np.float64)


def make_inputs(case):
    """Seeded inputs of a case: numpy arrays (images, labels[, sp_guide])."""
    from boxsegliver_amd.data import synthetic
    c = CASES[case]
    s = c["size"]
    if c["kind"] == "UNet3D":
        images, labels, _ = synthetic.make_batch_3d(2, 4, s, s, 1, 2, 1234)
        return dict(images=images, labels=labels)
    images, labels, _ = synthetic.make_batch(2, s, s, 3, 3, 1234)
    out = dict(images=images, labels=labels)
    if c["kind"] in GUIDED:
        out["sp_guide"] = synthetic.make_guide(labels, 1, 1234)
    return out


def load(case):
    return np.load(os.path.join(HERE, case + ".npz"))
