"""LGNet plugin -- host-side mirror of the reference's NetworksV2/LGNet.py:30-304 on the libunetk HIP kernels.

A U-Net (4 pools, 64..1024 channels) whose conv pairs are split around the guide: per level conv1 = conv + norm + ReLU,
conv2 = conv + norm, then `merge_guide_act` adds the level's spatial parameters -- a 1x1 conv of the average-pooled guide
with bias and a LEAKY ReLU (`_spatial_subnets`, :30-55) -- on the levels of `mod_layers[0]` (encoder, bridge = level 4)
and, after the decoder's conv1, on those of `mod_layers[1]`; then ReLU.  On MI355X the merged unit is the same fused
conv -> norm -> (+ guide term) -> ReLU kernel sequence GUNet uses, with `unetk_norm_desc.guide_leaky`: the guide conv is
evaluated on the fly in the norm-apply / norm-backward kernels (never materialised) and passed through the leaky ReLU
there.  Scopes: <name>/conv_e{i}/conv{1,2}, <name>/ED-Bridge/conv{1,2}, <name>/conv_d{i}/{up,conv1,conv2},
<name>/spatial/conv_{e,d}{l+1}, <name>/logits.

mod_layers must be ascending per branch (all shipped LGNet*.yml are): the reference pools the guide cumulatively.
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, pixel_weights
from .GUNet import GUNet
from .base import ModeKeys, ParamStore

LAYER_C = [64, 128, 256, 512, 1024]


def param_specs(in_channels, num_classes, guide_channel, mod_layers, normalizer, use_spatial, name):
    specs = []
    bn = normalizer == "batch_norm"
    ns = "BatchNorm" if bn else "InstanceNorm"

    def unit(scope, cin, cout):
        specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
        if bn:
            for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"),
                               ("moving_variance", "moving_var")):
                specs.append(("{}/{}/{}".format(scope, ns, leaf), (cout,), kind))
        else:
            specs.append(("{}/{}/gamma".format(scope, ns), (cout,), "gamma"))
            specs.append(("{}/{}/beta".format(scope, ns), (cout,), "beta"))

    if use_spatial:
        for branch, tag in ((0, "e"), (1, "d")):
            for l in mod_layers[branch]:
                specs.append(("{}/spatial/conv_{}{}/weights".format(name, tag, l + 1), (1, 1, guide_channel, LAYER_C[l]), "conv_w"))
                specs.append(("{}/spatial/conv_{}{}/biases".format(name, tag, l + 1), (LAYER_C[l],), "bias"))
    cin = in_channels
    for i in range(4):
        unit("{}/conv_e{}/conv1".format(name, i), cin, LAYER_C[i])
        unit("{}/conv_e{}/conv2".format(name, i), LAYER_C[i], LAYER_C[i])
        cin = LAYER_C[i]
    unit(name + "/ED-Bridge/conv1", 512, 1024)
    unit(name + "/ED-Bridge/conv2", 1024, 1024)
    for i in (3, 2, 1, 0):
        c = LAYER_C[i]
        specs.append(("{}/conv_d{}/up/weights".format(name, i), (2, 2, c, 2 * c), "deconv_w"))
        specs.append(("{}/conv_d{}/up/biases".format(name, i), (c,), "bias"))
        unit("{}/conv_d{}/conv1".format(name, i), 2 * c, c)
        unit("{}/conv_d{}/conv2".format(name, i), c, c)
    specs.append((name + "/logits/weights", (1, 1, 64, num_classes), "conv_w"))
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs


class LGNet(GUNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference LGNet.py:95-106)."""
        super(LGNet, self).__init__(args, name or "LGNet")
        self.use_context_guide = False
        self.use_se = False
        self.dropout = None            # stored by the reference (:106) but never used by its _build_network

    def _net_arg_scope(self, *args, **kwargs):
        """LGNet.py:108-130: every slim.conv2d = 3x3, normaliser from _get_normalization(), no activation (the ReLUs are
        explicit); pools SAME."""
        # --without_norm is INERT for LGNet, as in the reference: its _net_arg_scope (LGNet.py:108-130) never reads the flag
        # (only the module-level modulated_conv_block at :60-92 does, and _build_network does not call it), so every
        # slim.conv2d keeps its normaliser.  The flag is accepted and changes nothing.
        self._norm = self._get_normalization()
        return self._norm

    def _lg_unit(self, x, scope, out=None, guide=None, gw=None, gb=None, pool=False):
        spec = self._spec()
        spec.guide_leaky = guide is not None
        return self._unit(x, scope, spec, out, guide, gw, gb, pool=pool)

    def _build_network(self, *args, **kwargs):
        mod_layers = kwargs.get("mod_layers", [[0, 1], [1, 0]])
        mod_layers = [list(mod_layers[0]), list(mod_layers[1])]
        images = self._inputs["images"]
        if not images.is_cuda:
            raise ops._abi.UnetkError("LGNet runs on the GPU only: move `images` to cuda (no CPU path)")
        n, h, w, _ = images.shape
        if h % 16 or w % 16:
            raise ValueError("H and W must be divisible by 16")
        dev, nm = images.device, self.name
        g_ch = int(getattr(self.args, "guide_channel", 1)) if self.use_spatial_guide else 0
        if g_ch:
            for br in mod_layers:
                if br != sorted(br) or any(l < 0 or l > 4 for l in br):
                    raise ValueError("LGNet mod_layers must be ascending levels in 0..4 (the guide is pooled cumulatively, "
                                     "LGNet.py:39-52), got {}".format(mod_layers))
            if any(l > 3 for l in mod_layers[1]):
                raise ValueError("the decoder has levels 0..3")
        if self.params is None:
            in_ch = self.channel * (3 if getattr(self.args, "img_grad", False) else 1)
            specs = param_specs(in_ch, self.num_classes, g_ch, mod_layers, self.args.normalizer, g_ch > 0, nm)
            self.params = ParamStore(specs, dev, bias_decay=getattr(self.args, "bias_decay", False))
            self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))
        p = self.params

        with torch.set_grad_enabled(self.mode == ModeKeys.TRAIN):
            guides = {}
            if g_ch:
                gs = self._inputs["sp_guide"].to(torch.float32).contiguous()
                if gs.shape != (n, h, w, g_ch):
                    raise ValueError("sp_guide must be [bs, H, W, {}], got {}".format(g_ch, tuple(gs.shape)))
                need = set(mod_layers[0]) | set(mod_layers[1])
                for l in range(5):
                    if l in need:
                        guides[l] = gs
                    if l < 4:
                        gs = ops.avgpool2_fwd(gs)

            def sp(tag, l):
                wgt = p["{}/spatial/conv_{}{}/weights".format(nm, tag, l + 1)].view(g_ch, LAYER_C[l])
                return guides[l], wgt, p["{}/spatial/conv_{}{}/biases".format(nm, tag, l + 1)]

            x = ops.image_gradients(images.to(torch.float32)) if getattr(self.args, "img_grad", False) else images.contiguous()
            cats, skips = {}, {}
            hh, ww = h, w
            for i in range(5):
                c = LAYER_C[i]
                scope = "{}/conv_e{}".format(nm, i) if i < 4 else nm + "/ED-Bridge"
                x = self._lg_unit(x, scope + "/conv1")
                out = None
                if i < 4:
                    cat = torch.empty((n, hh, ww, 2 * c), dtype=torch.float32, device=dev)
                    out = ops.alias(cat, 0, (n, hh, ww, c), cat.stride())
                    cats[i] = cat
                guided = sp("e", i) if (g_ch and i in mod_layers[0]) else ()
                if i < 4:      # the pool rides on the unit's norm passes when the unit is plain (GUNet._unit, pool=True)
                    x, skips[i] = self._lg_unit(x, scope + "/conv2", out, *guided, pool=True)
                    hh //= 2
                    ww //= 2
                else:
                    x = self._lg_unit(x, scope + "/conv2", out, *guided)
            for i in (3, 2, 1, 0):
                d = "{}/conv_d{}".format(nm, i)
                x = ops.DeconvConcat.apply(x, p[d + "/up/weights"], p[d + "/up/biases"], skips[i], cats[i], False)
                if g_ch and i in mod_layers[1]:
                    x = self._lg_unit(x, d + "/conv1", None, *sp("d", i))
                else:
                    x = self._lg_unit(x, d + "/conv1")
                x = self._lg_unit(x, d + "/conv2")

            self.ret_prob = kwargs.get("ret_prob", False)
            self.ret_pred = kwargs.get("ret_pred", False)
            labels = self._inputs.get("labels")
            if labels is not None:
                labels = labels.to(torch.int32).contiguous()
            pixel_w = pixel_weights(self.args, self._inputs, labels)
            desc = build_head_desc(self.args, n, h * w, 64, self.num_classes, explicit_map=pixel_w is not None) \
                if labels is not None else ops.head_desc(n, h * w, 64, self.num_classes)
            want_probs = bool(self.ret_prob or self.ret_pred or self.mode != ModeKeys.TRAIN)
            xent, dice, logits, probs, result = ops.HeadLoss.apply(x, p[nm + "/logits/weights"], p[nm + "/logits/biases"],
                                                                   labels, pixel_w, desc, want_probs)
            self._head = (xent, dice, result)
            self._layers["logits"] = logits.view(n, h, w, self.num_classes)
            if want_probs:
                self.probability = probs.view(n, h, w, self.num_classes)
                if self.ret_prob:
                    for i in range(1, self.num_classes):
                        self.predictions[self.classes[i] + "Prob"] = self.probability[..., i:i + 1]
                if self.ret_pred:
                    _, preds = ops.head_predict(probs, self.num_classes, want_preds=True)
                    for i in range(1, self.num_classes):
                        obj = self.classes[i] + "Pred"
                        self.predictions[obj] = preds[i - 1].view(n, h, w, 1)
                        self._image_summaries[obj] = self.predictions[obj]
