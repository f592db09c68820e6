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

--without_norm: conv + bias + ReLU units (SmallUNet._unit).  Not built: init_channel_factor != 1.
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, pixel_weights
from .SmallUNet import SmallUNet
from .base import ModeKeys, ParamStore

ENC = [("e0", 32, 1), ("e1", 64, 2), ("e2", 128, 2)]
DEC = [(2, 256, 128), (1, 128, 64), (0, 64, 32)]


def param_specs(x_channels, y_channels, num_classes, normalizer, name, without_norm=False):
    specs = []

    def unit(scope, cin, cout):
        specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
        if without_norm:
            specs.append((scope + "/biases", (cout,), "bias"))
        elif normalizer == "batch_norm":
            for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"),
                               ("moving_variance", "moving_var")):
                specs.append(("{}/BatchNorm/{}".format(scope, leaf), (cout,), kind))
        else:
            specs.append((scope + "/InstanceNorm/gamma", (cout,), "gamma"))
            specs.append((scope + "/InstanceNorm/beta", (cout,), "beta"))

    for stream, cin0 in (("image", x_channels), ("inter", y_channels)):
        cin = cin0
        for tag, c, _ in ENC:
            unit("{}/{}_{}/conv1".format(name, stream, tag), cin, c)
            unit("{}/{}_{}/conv2".format(name, stream, tag), c, c)
            cin = c
    unit(name + "/merge_e3/conv1", 256, 512)
    unit(name + "/merge_e3/conv2", 512, 512)
    unit(name + "/merge_e3/conv3", 512, 1024)
    unit(name + "/merge_e3/conv4", 1024, 1024)
    unit(name + "/conv_d3/conv1", 1024, 512)
    unit(name + "/conv_d3/conv2", 512, 512)
    unit(name + "/conv_d3/conv3", 512, 512)
    cin = 512
    for i, c, skip in DEC:
        specs.append(("{}/conv_d{}/up/weights".format(name, i), (2, 2, c, cin), "deconv_w"))
        unit("{}/conv_d{}/conv1".format(name, i), c + 2 * skip, c)
        unit("{}/conv_d{}/conv2".format(name, i), c, c)
        cin = c
    specs.append((name + "/logits/weights", (1, 1, 64, num_classes), "conv_w"))
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs


class InterUNet(SmallUNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference InterUNet.py:70-79; default scope "SmallUNet", :74)."""
        super(InterUNet, self).__init__(args, name or "SmallUNet")

    def _build_network(self, *args, **kwargs):
        if kwargs.get("init_channel_factor", 1) != 1:
            raise NotImplementedError("InterUNet init_channel_factor != 1 is not built")
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
            self.params = ParamStore(param_specs(xc, yc, self.num_classes, self.args.normalizer, nm,
                                                 bool(getattr(self.args, "without_norm", False))), dev,
                                     bias_decay=getattr(self.args, "bias_decay", False))
            self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))
        p = self.params

        with torch.set_grad_enabled(self.mode == ModeKeys.TRAIN):
            images = images.to(torch.float32)
            x_in = torch.cat((images, guide), dim=-1).contiguous()                                   # InterUNet.py:103
            y_in = ops.sobel_concat(images, self.args.im_channel // 2) if img_grad else images.contiguous()   # :104-109
            # concat buffers: decoder level i = [up C_i | image_e{i} | inter_e{i}]; merge_e3's input concat(image_e2, inter_e2)
            # is the channel slice [256, 512) of the level-2 buffer -- no second copy
            sizes = {0: (h, w), 1: (h // 2, w // 2), 2: (h // 4, w // 4)}
            cats = {i: torch.empty((n,) + sizes[i] + (c + 2 * sk,), dtype=torch.float32, device=dev) for i, c, sk in DEC}
            skips = {}
            for si, (stream, t) in enumerate((("image", x_in), ("inter", y_in))):
                for li, (tag, c, stride) in enumerate(ENC):
                    t = self._unit(t, "{}/{}_{}/conv1".format(nm, stream, tag), stride, 1)
                    cat = cats[li]
                    up_c = DEC[2 - li][1]
                    out = ops.alias(cat, up_c + si * c, (n,) + sizes[li] + (c,), cat.stride())
                    t = self._unit(t, "{}/{}_{}/conv2".format(nm, stream, tag), 1, 1, out)
                    skips[(stream, li)] = t
            merge_view = ops.alias(cats[2], 256, (n,) + sizes[2] + (256,), cats[2].stride())
            z = _Join.apply(merge_view, skips[("image", 2)], skips[("inter", 2)])
            z = self._unit(z, nm + "/merge_e3/conv1", 2, 1)
            z = self._unit(z, nm + "/merge_e3/conv2", 1, 1)
            z = self._unit(z, nm + "/merge_e3/conv3", 1, 2)
            z = self._unit(z, nm + "/merge_e3/conv4", 1, 2)
            z = self._unit(z, nm + "/conv_d3/conv1", 1, 2)
            z = self._unit(z, nm + "/conv_d3/conv2", 1, 1)
            z = self._unit(z, nm + "/conv_d3/conv3", 1, 1)
            for i, c, sk in DEC:
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
            desc = build_head_desc(self.args, n, h * w, 64, self.num_classes, explicit_map=pixel_w is not None) \
                if labels is not None else ops.head_desc(n, h * w, 64, self.num_classes)
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
