"""UNetInter plugin -- host-side mirror of the reference's NetworksV2/UNetInter.py:31-200 on the libunetk HIP kernels.

A U-Net whose INPUT is concat(images, sp_guide) (UNetInter.py:87-88): the interaction guide joins the image channels
instead of modulating the encoder.  Topology, scopes and variable names are GUNet's (`<name>/Encode/down_conv{i}/
mod_conv{j}`, `<name>/Decode/up{i}`, `up_conv{i}`, `AdjustChannels`) with no modulation; every encoder unit is
conv -> norm(centre, scale; BN decay .99, UNetInter.py:98-113) -> ReLU, the decoder uses the `_get_normalization`
defaults.  So this class is GUNet with the guide routed to the input -- same kernels, nothing new on the device.

--mid_cat (scripts/106_unetinter_v1.sh; UNetInter.py:87-90,124-129): the guide is concatenated to the level-0 output
before the first pool instead of to the input, so Encode2's first conv sees 64 + g channels; on the device that filter is
padded with zero rows to 96 input channels (NetworksV2/padded.py) and the pooled tensor with zero channels.

`use_2d` (a static-shape hint in the reference) needs nothing here; --without_norm = conv + bias units.  --img_grad computes dy / dx in the reference but never feeds them to the net (:82-85): ignored.
"""
from .GUNet import GUNet


class UNetInter(GUNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference UNetInter.py:32-43)."""
        super(UNetInter, self).__init__(args, name or "UNetInter")
        self.use_context_guide = False
        self.use_se = False
        self.dropout = None
        self._concat_guide = True
        self._mid_cat = bool(getattr(args, "mid_cat", False))
        self._encoder_decay = 0.99

    def _net_arg_scope(self, *args, **kwargs):
        # use_2d (UNetInter.py:76-78) only pins the static graph shape to [1, H, W, 3]; shapes are dynamic here
        self._norm = ("none", {}) if getattr(self.args, "without_norm", False) else self._get_normalization()
        return self._norm

    def _build_network(self, *args, **kwargs):
        kwargs = dict(kwargs, mod_layers=[], after_affine=False)
        return super(UNetInter, self)._build_network(*args, **kwargs)
