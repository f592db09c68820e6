"""InterUNet plugin -- host-side mirror of the reference's NetworksV2/InterUNet.py:28-241 on the libunetk HIP kernels.

Two encoders of three blocks (32 / 64 / 128 channels; stride-2 first convs from the second block on): `image_e*` on
concat(images, sp_guide) and `inter_e*` on the images (+ the Sobel edges of the middle channel under --img_grad,
:105-109); `merge_e3` on concat(image_e2, inter_e2) -- stride-2 conv, conv, two rate-2 atrous convs; `conv_d3`; three
decoder blocks whose concat puts the UP-SAMPLED tensor first, then the two encoders' skips (:150-155); bias-free
transposed convs; logits in scope "logits".  The default variable scope is "SmallUNet" -- the reference's own
(`self.name = name or "SmallUNet"`, :74), kept so its checkpoints load by name.

All concats are zero-copy: each encoder block writes its output straight into its slice of the decoder's concat
buffer (channel offsets C_up and C_up + C_skip), the two e2 blocks into the halves of merge_e3's input, the transposed
conv into channels [0, C_up) (ops.DeconvConcatFront).  Kernels: SmallUNet's (dense / stride-2 / rate-2 conv units), the
direct small-Cin kernels for the 4- and 5-channel inputs, unetk_sobel_concat.

--without_norm: conv + bias + ReLU units (SmallUNet._unit).  init_channel_factor != 1 (`round(layer["out"] * c)`, InterUNet.py:121;
e.g. --model_config SmallUNet_V2.yml) runs on variables channel-padded to multiples of 64 (NetworksV2/padded.py).
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, pixel_weights
from .SmallUNet import SmallUNet
from .base import ModeKeys, ParamStore
from .padded import PaddedParamStore, pad_to

PAD = 64     # device channel granularity when init_channel_factor != 1 (as SmallUNet_V2: stride-2 / atrous / 64 x 64 tiles)


def model_channels(factor=1):
    """InterUNet.py:28-66 with `round(layer["out"] * c)` (:121 ...): encoder blocks [(tag, channels, first stride)],
    merge_e3's four convs, conv_d3's three, decoder levels [(level, up channels, skip channels of EACH encoder)]."""
    c = lambda v: int(round(v * factor))
    enc = [("e0", c(32), 1), ("e1", c(64), 2), ("e2", c(128), 2)]
    merge = [c(512), c(512), c(1024), c(1024)]
    d3 = [c(512), c(512), c(512)]
    dec = [(2, c(256), enc[2][1]), (1, c(128), enc[1][1]), (0, c(64), enc[0][1])]
    return enc, merge, d3, dec


ENC, _MERGE, _D3, DEC = model_channels(1)


def phys(c, factor):
    """Device channel count of a logical one: unpadded at factor 1 (the shipped InterUNet.yml), multiples of 64 otherwise."""
    return c if factor == 1 else pad_to(c, PAD)


def param_specs(x_channels, y_channels, num_classes, normalizer, name, without_norm=False, factor=1):
    """Logical specs (the reference's TF shapes) and the device padding of each variable (empty at factor 1).  A concat
    input pads each part separately: logical channel ranges map to the padded slots of the concat buffer."""
    enc, merge, d3, dec = model_channels(factor)
    specs, pads = [], {}
    P = lambda c: phys(c, factor)

    def vec(vname, c, kind):
        specs.append((vname, (c,), kind))
        if P(c) != c:
            pads[vname] = ((P(c),), {})

    def unit(scope, cin_parts, cout, raw_input=False):
        cin = sum(cin_parts)
        specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
        if factor != 1:
            segs, lpos, ppos = [], 0, 0
            for c in cin_parts:
                segs.append((lpos, c, ppos))
                lpos += c
                ppos += c if raw_input else P(c)
            pads[scope + "/weights"] = ((3, 3, ppos, P(cout)), {2: segs})
        if without_norm:
            vec(scope + "/biases", cout, "bias")
        elif normalizer == "batch_norm":
            for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"),
                               ("moving_variance", "moving_var")):
                vec("{}/BatchNorm/{}".format(scope, leaf), cout, kind)
        else:
            vec(scope + "/InstanceNorm/gamma", cout, "gamma")
            vec(scope + "/InstanceNorm/beta", cout, "beta")

    for stream, cin0 in (("image", x_channels), ("inter", y_channels)):
        cin, raw = cin0, True
        for tag, c, _ in enc:
            unit("{}/{}_{}/conv1".format(name, stream, tag), [cin], c, raw)
            unit("{}/{}_{}/conv2".format(name, stream, tag), [c], c)
            cin, raw = c, False
    e2 = enc[2][1]
    unit(name + "/merge_e3/conv1", [e2, e2], merge[0])
    unit(name + "/merge_e3/conv2", [merge[0]], merge[1])
    unit(name + "/merge_e3/conv3", [merge[1]], merge[2])
    unit(name + "/merge_e3/conv4", [merge[2]], merge[3])
    unit(name + "/conv_d3/conv1", [merge[3]], d3[0])
    unit(name + "/conv_d3/conv2", [d3[0]], d3[1])
    unit(name + "/conv_d3/conv3", [d3[1]], d3[2])
    cin = d3[2]
    for i, c, skip in dec:
        wname = "{}/conv_d{}/up/weights".format(name, i)
        specs.append((wname, (2, 2, c, cin), "deconv_w"))
        if factor != 1:
            pads[wname] = ((2, 2, P(c), P(cin)), {2: [(0, c, 0)], 3: [(0, cin, 0)]})
        unit("{}/conv_d{}/conv1".format(name, i), [c, skip, skip], c)
        unit("{}/conv_d{}/conv2".format(name, i), [c], c)
        cin = c
    specs.append((name + "/logits/weights", (1, 1, cin, num_classes), "conv_w"))
    if factor != 1:
        pads[name + "/logits/weights"] = ((1, 1, P(cin), num_classes), {2: [(0, cin, 0)]})
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs, pads


class InterUNet(SmallUNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference InterUNet.py:70-79; default scope "SmallUNet", :74)."""
        super(InterUNet, self).__init__(args, name or "SmallUNet")

    def _build_network(self, *args, **kwargs):
        factor = kwargs.get("init_channel_factor", 1)
        enc, merge, d3, dec = model_channels(factor)
        P = lambda c: phys(c, factor)
        if kwargs.get("num_pool_layers", 3) != 3:
            raise KeyError(kwargs.get("num_pool_layers"))
        images = self._inputs["images"]
        if not images.is_cuda:
            raise ops._abi.UnetkError("InterUNet runs on the GPU only: move `images` to cuda (no CPU path)")
        guide = self._inputs["sp_guide"].to(torch.float32)
        n, h, w, ch = images.shape
        if guide.shape[:3] != images.shape[:3]:
            raise ValueError("sp_guide must be [bs, H, W, g], got {}".format(tuple(guide.shape)))
        if h % 8 or w % 8:
            raise ValueError("H and W must be divisible by 8")
        dev, nm = images.device, self.name
        img_grad = bool(getattr(self.args, "img_grad", False))
        xc, yc = ch + guide.shape[3], ch + (2 if img_grad else 0)
        if self.params is None:
            specs, pads = param_specs(xc, yc, self.num_classes, self.args.normalizer, nm,
                                      bool(getattr(self.args, "without_norm", False)), factor)
            bd = getattr(self.args, "bias_decay", False)
            # factor != 1 (e.g. --model_config SmallUNet_V2.yml: 0.75 -> 24 / 48 / 96 / 384 ... channels): variables padded to
            # multiples of 64 on the device, exactly zero in the padding (NetworksV2/padded.py); factor 1 stays unpadded
            self.params = PaddedParamStore(specs, pads, dev, bias_decay=bd) if pads else ParamStore(specs, dev, bias_decay=bd)
            self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))
        p = self.params

        with torch.set_grad_enabled(self.mode == ModeKeys.TRAIN):
            images = images.to(torch.float32)
            x_in = torch.cat((images, guide), dim=-1).contiguous()                                   # InterUNet.py:103
            y_in = ops.sobel_concat(images, self.args.im_channel // 2) if img_grad else images.contiguous()   # :104-109
            # concat buffers: decoder level i = [up C_i | image_e{i} | inter_e{i}]; merge_e3's input concat(image_e2, inter_e2)
            # is the channel slice [256, 512) of the level-2 buffer -- no second copy
            # (all channel counts below are DEVICE counts: P(c) = c at factor 1, padded to 64 otherwise)
            sizes = {0: (h, w), 1: (h // 2, w // 2), 2: (h // 4, w // 4)}
            cats = {i: torch.empty((n,) + sizes[i] + (P(c) + 2 * P(sk),), dtype=torch.float32, device=dev) for i, c, sk in dec}
            skips = {}
            for si, (stream, t) in enumerate((("image", x_in), ("inter", y_in))):
                for li, (tag, c, stride) in enumerate(enc):
                    t = self._unit(t, "{}/{}_{}/conv1".format(nm, stream, tag), stride, 1)
                    cat = cats[li]
                    up_c = P(dec[2 - li][1])
                    out = ops.alias(cat, up_c + si * P(c), (n,) + sizes[li] + (P(c),), cat.stride())
                    t = self._unit(t, "{}/{}_{}/conv2".format(nm, stream, tag), 1, 1, out)
                    skips[(stream, li)] = t
            merge_view = ops.alias(cats[2], P(dec[0][1]), (n,) + sizes[2] + (2 * P(enc[2][1]),), cats[2].stride())
            z = _Join.apply(merge_view, skips[("image", 2)], skips[("inter", 2)])
            z = self._unit(z, nm + "/merge_e3/conv1", 2, 1)
            z = self._unit(z, nm + "/merge_e3/conv2", 1, 1)
            z = self._unit(z, nm + "/merge_e3/conv3", 1, 2)
            z = self._unit(z, nm + "/merge_e3/conv4", 1, 2)
            z = self._unit(z, nm + "/conv_d3/conv1", 1, 2)
            z = self._unit(z, nm + "/conv_d3/conv2", 1, 1)
            z = self._unit(z, nm + "/conv_d3/conv3", 1, 1)
            for i, c, sk in dec:
                z = ops.DeconvConcatFront.apply(z, p["{}/conv_d{}/up/weights".format(nm, i)], None, skips[("image", i)],
                                                skips[("inter", i)], cats[i])
                z = self._unit(z, "{}/conv_d{}/conv1".format(nm, i), 1, 1)
                z = self._unit(z, "{}/conv_d{}/conv2".format(nm, i), 1, 1)

            self.ret_prob = kwargs.get("ret_prob", False)
            self.ret_pred = kwargs.get("ret_pred", False)
            labels = self._inputs.get("labels")
            if labels is not None:
                labels = labels.to(torch.int32).contiguous()
            pixel_w = pixel_weights(self.args, self._inputs, labels)
            c_last = P(dec[2][1])
            desc = build_head_desc(self.args, n, h * w, c_last, self.num_classes, explicit_map=pixel_w is not None) \
                if labels is not None else ops.head_desc(n, h * w, c_last, self.num_classes)
            want_probs = bool(self.ret_prob or self.ret_pred or self.mode != ModeKeys.TRAIN)
            xent, dice, logits, probs, result = ops.HeadLoss.apply(z, p[nm + "/logits/weights"], p[nm + "/logits/biases"],
                                                                   labels, pixel_w, desc, want_probs)
            self._head = (xent, dice, result)
            self._layers["logits"] = logits.view(n, h, w, self.num_classes)
            if want_probs:
                self.probability = probs.view(n, h, w, self.num_classes)
                if self.ret_pred:
                    _, preds = ops.head_predict(probs, self.num_classes, want_preds=True)
                    for i in range(1, self.num_classes):
                        obj = self.classes[i] + "Pred"
                        self.predictions[obj] = preds[i - 1].view(n, h, w, 1)
                        self._image_summaries[obj] = self.predictions[obj]


class _Join(torch.autograd.Function):
    """Two adjacent channel slices of a concat buffer, already filled by their producers, as ONE tensor (a strided view);
    the gradient splits back onto the producers."""

    @staticmethod
    def forward(ctx, buf, a, b):
        ctx.ca = a.shape[3]
        return ops.alias(buf)

    @staticmethod
    def backward(ctx, g):
        return None, g[..., :ctx.ca], g[..., ctx.ca:]
