"""ParamStore whose device tensors are channel-padded for the MFMA tiles while the checkpoint / TF view keeps
the reference's logical shapes.

UNet3D's channel counts (30, 60, 120, 240, 320 -- NetworksV2/UNet3D.py:155) are not multiples of the 32-wide
fp32-MFMA tile, so the kernels run on variables padded to 32 / 64 / 128 / 256 / 320 channels.  The padding is
exact, not an approximation: padded filter rows / columns are zero, so padded channels carry exactly 0 through
conv -> norm (beta_pad = 0) -> ReLU, their gradients are exactly 0 (du = dz * (u > 0) with u = 0), and Adam /
L2 leave a zero parameter with zero gradient at zero.  `state_dict` / `load_state` / `num_trainable` speak the
logical shapes (TF variable shapes), so checkpoints are interchangeable with the reference's.

A concat input (skip, up) pads each half separately: logical input channel c < C maps to c, c >= C maps to
P + (c - C) -- described per axis as segments (logical start, length, physical start).
"""
import itertools
from collections import OrderedDict

import torch

from .base import ParamStore


def pad_to(c, mult=32):
    return (c + mult - 1) // mult * mult


class PaddedParamStore(ParamStore):
    def __init__(self, specs, pads, device, bias_decay=False):
        """specs: logical (name, shape, kind); pads: {name: (phys_shape, {axis: [(lstart, length, pstart), ...]})}."""
        self.logical_specs = list(specs)
        self.pads = dict(pads)
        phys = [(n, tuple(self.pads[n][0]) if n in self.pads else tuple(s), k) for n, s, k in specs]
        super(PaddedParamStore, self).__init__(phys, device, bias_decay)
        self.logical_shape = {n: tuple(s) for n, s, _ in specs}
        # the conv kernels skip the padded groups of their contraction axis (ops.conv3d_desc live8; include/unetk.h
        # unetk_conv3d_desc.cin_live8): the masks ride on the filter tensors handed out by p[name]
        for n, _, kind in specs:
            if kind == "conv_w" and n in self.pads and len(self.pads[n][0]) == 5:
                self.tensors[n].unetk_live8 = self.live8_masks(n)

    def live8_masks(self, name):
        """(Cin mask, Cout mask) of a padded [kd, kh, kw, Cin, Cout] filter: bit i = physical channels [8 i, 8 i + 8) hold a
        logical channel; 0 where the axis does not fit a 64-bit mask of 8-channel groups (the kernels then contract it all)."""
        phys, axis_maps = self.pads[name]
        lshape = self.logical_shape[name]
        out = []
        for ax in (3, 4):
            m = 0
            if phys[ax] % 16 == 0 and phys[ax] <= 512:
                for _, ln, ps in axis_maps.get(ax, [(0, lshape[ax], 0)]):
                    for g in range(ps // 8, (ps + ln + 7) // 8):
                        m |= 1 << g
            out.append(m)
        return tuple(out)

    def _blocks(self, name):
        """Yield (logical index tuple, physical index tuple) of every dense block of the variable."""
        lshape = self.logical_shape[name]
        axis_maps = self.pads[name][1] if name in self.pads else {}
        per_axis = []
        for ax, n in enumerate(lshape):
            segs = axis_maps.get(ax, [(0, n, 0)])
            per_axis.append([(slice(ls, ls + ln), slice(ps, ps + ln)) for ls, ln, ps in segs])
        for combo in itertools.product(*per_axis):
            yield tuple(c[0] for c in combo), tuple(c[1] for c in combo)

    def num_trainable(self):
        total = 0
        for n in self.trainable_names():
            k = 1
            for s in self.logical_shape[n]:
                k *= s
            total += k
        return total

    @torch.no_grad()
    def load_state(self, state, strict=True):
        for name in self.tensors:
            if name in state:
                v = state[name]
                v = (v if torch.is_tensor(v) else torch.as_tensor(v)).to(torch.float32).reshape(self.logical_shape[name])
                t = self.tensors[name]
                t.zero_()
                v = v.to(t.device)
                for lidx, pidx in self._blocks(name):
                    t[pidx] = v[lidx]
            elif strict:
                raise KeyError("missing variable " + name)

    def state_dict(self):
        out = OrderedDict()
        for name, t in self.tensors.items():
            v = torch.zeros(self.logical_shape[name], dtype=torch.float32)
            tc = t.detach().cpu()
            for lidx, pidx in self._blocks(name):
                v[lidx] = tc[pidx]
            out[name] = v
        return out

    def read_slot(self, flat, name):
        """Optimiser slot of `name` in its logical (TF) shape: the padded entries (exactly zero) are dropped."""
        _, off, n, shp, _ = self.where[name]
        phys = flat[off:off + n].view(shp).detach().cpu()
        v = torch.zeros(self.logical_shape[name], dtype=torch.float32)
        for lidx, pidx in self._blocks(name):
            v[lidx] = phys[pidx]
        return v

    @torch.no_grad()
    def write_slot(self, flat, name, value):
        _, off, n, shp, _ = self.where[name]
        v = (value if torch.is_tensor(value) else torch.as_tensor(value)).to(torch.float32).reshape(self.logical_shape[name])
        t = flat[off:off + n].view(shp)
        t.zero_()
        v = v.to(t.device)
        for lidx, pidx in self._blocks(name):
            t[pidx] = v[lidx]

    def logical_grad(self, name):
        """Gradient of a variable in its logical (TF) shape."""
        g = self.tensors[name].grad.detach().cpu()
        v = torch.zeros(self.logical_shape[name], dtype=torch.float32)
        for lidx, pidx in self._blocks(name):
            v[lidx] = g[pidx]
        return v

    @torch.no_grad()
    def initialize(self, weight_init="xavier", seed=None):
        """Initialise the LOGICAL variables exactly as ParamStore would (fans from the TF shapes), then scatter."""
        tmp = ParamStore(self.logical_specs, torch.device("cpu"))
        tmp.initialize(weight_init, seed)
        self.load_state(tmp.state_dict())
