"""SmallUNet plugin -- host-side mirror of the reference's NetworksV2/SmallUNet.py:28-207 on the libunetk HIP kernels.

A four-level U-Net for the interactive-segmentation tasks (scripts/104_small_*.sh): input = concat(images, sp_guide)
(:97), down-sampling by STRIDE-2 3x3 convs instead of pooling, two RATE-2 atrous convs in the bridge and one in conv_d3
at the 1/8 resolution, transposed convs WITHOUT bias (:116-123), logits in scope "logits".  Topology table
`_ModelConfig.config[3]` (:32-59) restated in `model_config`.

Kernels: stride-1 units = ops.Conv3x3NormRelu (dilation 2 for the atrous ones: unetk_conv_desc.dilation); stride-2 units
= ops.Conv3dNormRelu on [N, 1, H, W, C] views (the natively strided (1,3,3)/(1,2,2) path of UNet3D); decoder =
ops.DeconvConcat with a NULL bias writing into the zero-copy concat buffers the encoder filled.

`init_channel_factor` (SmallUNet_V2.yml: 0.75 -> 48 / 96 / 192 / 384 / 768 channels): the device variables are
channel-padded to multiples of 64 (48 -> 64, 96 -> 128) exactly as UNet3D's are (NetworksV2/padded.py: padded filter rows /
columns are zero and provably stay zero; checkpoints speak the TF shapes).

--without_norm (:81-82): every unit = conv + bias + ReLU (variables <scope>/biases), through the same kernels with the
norm stage reduced to the per-channel shift (unetk_norm_desc.affine_only).
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, pixel_weights
from .UNet import UNet
from .base import ModeKeys
from .padded import PaddedParamStore, pad_to


def model_config(factor=1.0):
    """[(block, [(layer, out channels, stride, dilation)])] in graph order (SmallUNet.py:32-59); "up" = conv2d_transpose."""
    c = lambda v: int(round(v * factor))
    return [("conv_e0", [("conv1", c(64), 1, 1), ("conv2", c(64), 1, 1)]),
            ("conv_e1", [("conv1", c(128), 2, 1), ("conv2", c(128), 1, 1)]),
            ("conv_e2", [("conv1", c(256), 2, 1), ("conv2", c(256), 1, 1)]),
            ("conv_e3", [("conv1", c(512), 2, 1), ("conv2", c(512), 1, 1)]),
            ("bridge", [("conv1", c(1024), 1, 2), ("conv2", c(1024), 1, 2)]),
            ("conv_d3", [("conv1", c(512), 1, 2), ("conv2", c(512), 1, 1), ("conv3", c(512), 1, 1)]),
            ("conv_d2", [("up", c(256), 2, 1), ("conv1", c(256), 1, 1), ("conv2", c(256), 1, 1)]),
            ("conv_d1", [("up", c(128), 2, 1), ("conv1", c(128), 1, 1), ("conv2", c(128), 1, 1)]),
            ("conv_d0", [("up", c(64), 2, 1), ("conv1", c(64), 1, 1), ("conv2", c(64), 1, 1)])]


PAD = 64     # device channel granularity (the stride-2 / atrous / 64 x 64 filter-gradient tiles)


def param_specs(in_channels, num_classes, factor, normalizer, name, without_norm=False):
    """Logical specs <name>/<block>/<layer>/{weights, BatchNorm|InstanceNorm/...}, <name>/<block>/up/weights,
    <name>/logits/{weights,biases} and the device padding of each variable (empty at factor 1)."""
    specs, pads = [], {}
    enc_out = {}
    cin_parts = [in_channels]

    def vec(vname, c, kind):
        specs.append((vname, (c,), kind))
        if pad_to(c, PAD) != c:
            pads[vname] = ((pad_to(c, PAD),), {})

    def in_layout(parts):
        segs, lpos, ppos = [], 0, 0
        for c in parts:
            segs.append((lpos, c, ppos))
            lpos += c
            ppos += pad_to(c, PAD) if c >= 16 else c          # the raw image + guide channels are not padded
        return ppos, segs

    for block, layers in model_config(factor):
        for layer, cout, _, _ in layers:
            scope = "{}/{}/{}".format(name, block, layer)
            pcin, segs = in_layout(cin_parts)
            if layer == "up":
                specs.append((scope + "/weights", (2, 2, cout, sum(cin_parts)), "deconv_w"))
                pads[scope + "/weights"] = ((2, 2, pad_to(cout, PAD), pcin), {2: [(0, cout, 0)], 3: segs})
                cin_parts = [enc_out[block.replace("d", "e")], cout]
                continue
            specs.append((scope + "/weights", (3, 3, sum(cin_parts), cout), "conv_w"))
            pads[scope + "/weights"] = ((3, 3, pcin, pad_to(cout, PAD)), {2: segs})
            if without_norm:
                vec(scope + "/biases", cout, "bias")
            elif normalizer == "batch_norm":
                for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"),
                                   ("moving_variance", "moving_var")):
                    vec("{}/BatchNorm/{}".format(scope, leaf), cout, kind)
            else:
                vec(scope + "/InstanceNorm/gamma", cout, "gamma")
                vec(scope + "/InstanceNorm/beta", cout, "beta")
            cin_parts = [cout]
        if block.startswith("conv_e"):
            enc_out[block] = cin_parts[0]
    pcin, segs = in_layout(cin_parts)
    specs.append((name + "/logits/weights", (1, 1, sum(cin_parts), num_classes), "conv_w"))
    pads[name + "/logits/weights"] = ((1, 1, pcin, num_classes), {2: segs})
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs, pads


class SmallUNet(UNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference SmallUNet.py:63-72)."""
        super(SmallUNet, self).__init__(args, name or "SmallUNet")

    def _net_arg_scope(self, *args, **kwargs):
        self._norm = ("none", {}) if getattr(self.args, "without_norm", False) else self._get_normalization()
        return self._norm

    def _unit(self, x, scope, stride, dilation, out=None):
        """slim.conv2d(x, C, 3, stride, rate) = conv (no bias) + norm + ReLU (SmallUNet.py:104-110, 127-131)."""
        p = self.params
        kind, np_ = self._norm
        extra = (None, None)
        if kind == "none":
            spec = ops.NormSpec("none", 0.0, 0.0, self.is_training, False)
            gamma, beta = None, p[scope + "/biases"]
        elif kind == "batch_norm":
            ns = scope + "/BatchNorm"
            spec = ops.NormSpec("batch_norm", np_["eps"], np_["decay"], bool(np_["is_training"]), False)
            extra = (p[ns + "/moving_mean"], p[ns + "/moving_variance"])
            gamma, beta = p[ns + "/gamma"], p[ns + "/beta"]
        else:
            ns = scope + "/InstanceNorm"
            spec = ops.NormSpec("instance_norm", np_["eps"], 0.0, self.is_training, False)
            gamma, beta = p[ns + "/gamma"], p[ns + "/beta"]
        w = p[scope + "/weights"]
        if stride == 1:
            return ops.Conv3x3NormRelu.apply(x, w, gamma, beta, extra[0], extra[1], spec, out, None, None, None, None,
                                             dilation)
        assert dilation == 1 and out is None
        z = ops.Conv3dNormRelu.apply(x.unsqueeze(1), w.unsqueeze(0), gamma, beta, extra[0], extra[1], spec,
                                     (1, stride, stride), None)
        return z.squeeze(1)

    def _build_network(self, *args, **kwargs):
        factor = kwargs.get("init_channel_factor", 1)
        if kwargs.get("num_pool_layers", 3) != 3:
            raise KeyError(kwargs.get("num_pool_layers"))               # the reference only defines config[3]
        if any(cout % 16 for _, layers in model_config(factor) for _, cout, _, _ in layers):
            raise NotImplementedError("SmallUNet init_channel_factor {} gives channel counts that are not multiples of 16"
                                      .format(factor))
        images = self._inputs["images"]
        if not images.is_cuda:
            raise ops._abi.UnetkError("SmallUNet runs on the GPU only: move `images` to cuda (no CPU path)")
        guide = self._inputs["sp_guide"].to(torch.float32)
        n, h, w, _ = images.shape
        if guide.shape[:3] != images.shape[:3]:
            raise ValueError("sp_guide must be [bs, H, W, g], got {}".format(tuple(guide.shape)))
        if h % 8 or w % 8:
            raise ValueError("H and W must be divisible by 8")
        dev, nm = images.device, self.name
        if self.params is None:
            specs, pads = param_specs(images.shape[3] + guide.shape[3], self.num_classes, factor, self.args.normalizer, nm,
                                      bool(getattr(self.args, "without_norm", False)))
            self.params = PaddedParamStore(specs, pads, dev, bias_decay=getattr(self.args, "bias_decay", False))
            self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))
        p = self.params

        with torch.set_grad_enabled(self.mode == ModeKeys.TRAIN):
            x = torch.cat((images.to(torch.float32), guide), dim=-1).contiguous()       # SmallUNet.py:97
            cats, skips = {}, {}
            hh, ww = h, w
            c = 0
            for block, layers in model_config(factor):
                for li, (layer, cout, stride, dilation) in enumerate(layers):
                    scope = "{}/{}/{}".format(nm, block, layer)
                    if layer == "up":
                        enc = block.replace("d", "e")
                        x = ops.DeconvConcat.apply(x, p[scope + "/weights"], None, skips[enc], cats[enc], False)
                        hh, ww = hh * 2, ww * 2
                        continue
                    if stride == 2:
                        hh, ww = hh // 2, ww // 2
                    cout = pad_to(cout, PAD)                     # physical channels
                    out = None
                    if block in ("conv_e0", "conv_e1", "conv_e2") and li == len(layers) - 1:
                        # the block's output is a skip connection: write it straight into its concat buffer
                        cat = torch.empty((n, hh, ww, 2 * cout), dtype=torch.float32, device=dev)
                        out = ops.alias(cat, 0, (n, hh, ww, cout), cat.stride())
                        cats[block] = cat
                    x = self._unit(x, scope, stride, dilation, out)
                    c = cout
                    if out is not None:
                        skips[block] = x
            self.ret_prob = kwargs.get("ret_prob", False)
            self.ret_pred = kwargs.get("ret_pred", False)
            labels = self._inputs.get("labels")
            if labels is not None:
                labels = labels.to(torch.int32).contiguous()
            pixel_w = pixel_weights(self.args, self._inputs, labels)
            desc = build_head_desc(self.args, n, h * w, c, self.num_classes, explicit_map=pixel_w is not None) \
                if labels is not None else ops.head_desc(n, h * w, c, self.num_classes)
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
