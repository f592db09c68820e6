"""2-D U-Net plugin -- host-side mirror of the reference's NetworksV2/UNet.py:29-176, executing
on hand-written HIP kernels (boxsegliver_amd.ops -> libunetk.so).

Topology (UNet.py:58-118): num_down_samples x [2 x (conv3x3 -> BN -> ReLU) -> maxpool 2x2],
bridge 2 x conv, num_down_samples x [deconv 2x2 s2 (+bias, ReLU) -> concat(skip, up) ->
2 x conv3x3-BN-ReLU], 1x1 logits (+bias, linear), softmax, Pred = prob > 0.5.

MI355X layout decisions: NHWC fp32 everywhere; each decoder concat buffer [N,H,W,2C] is allocated
up front, the encoder's second conv writes its activation straight into channels [0,C) and the
transposed conv writes channels [C,2C) -- tf.concat (UNet.py:93) costs zero bytes.
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, metrics_from_sums, pixel_weights
from ..utils import distribution_utils
from . import base
from .base import ModeKeys, ParamStore


def param_specs(in_channels, num_classes, init_channels, num_down_samples, normalizer, without_norm, name):
    """Variables in graph-construction order with the reference's TF names (UNet.py:203-205):
    <name>/Encode{i}/Repeat/convolution2d_{1,2}/{weights,BatchNorm/{gamma,beta,moving_mean,moving_variance}},
    <name>/ED-Bridge/ED-Bridge_{1,2}/..., <name>/Decode{i}/Conv2d_transpose/{weights,biases},
    <name>/Decode{i}/Repeat/convolution2d_{1,2}/..., <name>/AdjustChannels/{weights,biases}."""
    specs = []

    def conv_unit(scope, cin, cout):
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

    c, cin = init_channels, in_channels
    for i in range(num_down_samples):
        s = "{}/Encode{}/Repeat/convolution2d_".format(name, i + 1)
        conv_unit(s + "1", cin, c)
        conv_unit(s + "2", c, c)
        cin = c
        c *= 2
    conv_unit(name + "/ED-Bridge/ED-Bridge_1", cin, c)
    conv_unit(name + "/ED-Bridge/ED-Bridge_2", c, c)
    for i in reversed(range(num_down_samples)):
        c //= 2
        d = "{}/Decode{}".format(name, i + 1)
        specs.append((d + "/Conv2d_transpose/weights", (2, 2, c, 2 * c), "deconv_w"))
        specs.append((d + "/Conv2d_transpose/biases", (c,), "bias"))
        conv_unit(d + "/Repeat/convolution2d_1", 2 * c, c)
        conv_unit(d + "/Repeat/convolution2d_2", c, c)
    specs.append((name + "/AdjustChannels/weights", (1, 1, c, num_classes), "conv_w"))
    specs.append((name + "/AdjustChannels/biases", (num_classes,), "bias"))
    return specs


class UNet(base.BaseNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference UNet.py:30-39)."""
        super(UNet, self).__init__(args)
        self.name = name or "UNet"
        self.classes.extend(self.args.classes)
        self.bs = distribution_utils.per_device_batch_size(args.batch_size, args.num_gpus)
        self.height = args.im_height
        self.width = args.im_width
        self.channel = args.im_channel
        self._norm = None
        self._taps = None           # debug: set to a dict to collect named activations

    # ------------------------------------------------------------------ variables
    def _ensure_params(self, device, init_channels, num_down_samples):
        if self.params is not None:
            return
        in_ch = self.channel * (3 if getattr(self.args, "img_grad", False) else 1)
        specs = param_specs(in_ch, self.num_classes, init_channels, num_down_samples,
                            self.args.normalizer, getattr(self.args, "without_norm", False), self.name)
        self.params = ParamStore(specs, device, bias_decay=getattr(self.args, "bias_decay", False))
        self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))

    def _net_arg_scope(self, *args, **kwargs):
        """UNet.py:41-56: conv2d -> normaliser (no bias) + ReLU; conv2d_transpose -> bias + ReLU."""
        if getattr(self.args, "without_norm", False):
            self._norm = ("none", {})
        else:
            self._norm = self._get_normalization()
        return self._norm

    def _conv_unit(self, x, scope, out=None, pool=False):
        """One slim.conv2d(x, C, 3) unit.  pool=True: the unit whose activation feeds slim.max_pool2d and the skip connection
        (UNet.py:80-81,93) -- returns (pooled, activation) from ONE autograd node (ops.Conv3x3NormReluPool)."""
        p = self.params
        kind, nparams = self._norm
        if kind == "none":           # UNet.py:47-48: conv + bias + ReLU
            spec = ops.NormSpec("none", 0.0, 0.0, self.is_training, self.compute_bf16)
            var = (None, p[scope + "/biases"], None, None)
        elif kind == "batch_norm":
            bn = scope + "/BatchNorm"
            spec = ops.NormSpec("batch_norm", nparams["eps"], nparams["decay"], bool(nparams["is_training"]),
                                self.compute_bf16)
            var = (p[bn + "/gamma"], p[bn + "/beta"], p[bn + "/moving_mean"], p[bn + "/moving_variance"])
        else:                        # slim.instance_norm defaults: centre + scale, eps 1e-6
            inn = scope + "/InstanceNorm"
            spec = ops.NormSpec("instance_norm", nparams["eps"], 0.0, self.is_training, self.compute_bf16)
            var = (p[inn + "/gamma"], p[inn + "/beta"], None, None)
        if pool:
            pooled, z = ops.Conv3x3NormReluPool.apply(x, p[scope + "/weights"], var[0], var[1], var[2], var[3], spec, out)
        else:
            z = ops.Conv3x3NormRelu.apply(x, p[scope + "/weights"], var[0], var[1], var[2], var[3], spec, out, None, None, None)
        if self._taps is not None:
            self._taps[scope] = z
        return (pooled, z) if pool else z

    # ------------------------------------------------------------------ network
    def _build_network(self, *args, **kwargs):
        out_channels = kwargs.get("init_channels", 64)
        num_down_samples = kwargs.get("num_down_samples", 4)
        images = self._inputs["images"]
        if not images.is_cuda:
            raise ops._abi.UnetkError("UNet runs on the GPU only: move `images` to cuda (no CPU path)")
        if images.dim() != 4 or images.shape[3] != self.channel:
            raise ValueError("images must be [bs, H, W, {}], got {}".format(self.channel, tuple(images.shape)))
        self._ensure_params(images.device, out_channels, num_down_samples)
        n, h, w, _ = images.shape
        if h % (1 << num_down_samples) or w % (1 << num_down_samples):
            raise ValueError("H and W must be divisible by 2**num_down_samples")
        dev = images.device
        nm = self.name

        if getattr(self.args, "img_grad", False):
            tensor_out = ops.image_gradients(images.to(torch.float32))        # UNet.py:69-71
        else:
            tensor_out = images.contiguous()

        grad_mode = self.mode == ModeKeys.TRAIN
        with torch.set_grad_enabled(grad_mode):
            cats = {}
            skips = {}
            c = out_channels
            hh, ww = h, w
            for i in range(num_down_samples):
                s = "{}/Encode{}/Repeat/convolution2d_".format(nm, i + 1)
                tensor_out = self._conv_unit(tensor_out, s + "1")
                cat = torch.empty((n, hh, ww, 2 * c), dtype=self.storage_dtype, device=dev)
                skip_view = ops.alias(cat, 0, (n, hh, ww, c), cat.stride())
                tensor_out, skips[i] = self._conv_unit(tensor_out, s + "2", out=skip_view, pool=True)
                self._layers["Encode{:d}".format(i + 1)] = skips[i]
                cats[i] = cat
                c *= 2
                hh //= 2
                ww //= 2

            tensor_out = self._conv_unit(tensor_out, nm + "/ED-Bridge/ED-Bridge_1")
            tensor_out = self._conv_unit(tensor_out, nm + "/ED-Bridge/ED-Bridge_2")

            for i in reversed(range(num_down_samples)):
                c //= 2
                d = "{}/Decode{}".format(nm, i + 1)
                tensor_out = ops.DeconvConcat.apply(tensor_out, self.params[d + "/Conv2d_transpose/weights"],
                                                    self.params[d + "/Conv2d_transpose/biases"], skips[i], cats[i],
                                                    self.compute_bf16)
                tensor_out = self._conv_unit(tensor_out, d + "/Repeat/convolution2d_1")
                tensor_out = self._conv_unit(tensor_out, d + "/Repeat/convolution2d_2")

            # final 1x1 conv + loss head: one fused kernel (UNet.py:97-135)
            self.ret_prob = kwargs.get("ret_prob", False)
            self.ret_pred = kwargs.get("ret_pred", False)
            labels = self._inputs.get("labels")
            if labels is not None:
                labels = labels.to(torch.int32).contiguous()
            pixel_w = pixel_weights(self.args, self._inputs, labels)
            desc = build_head_desc(self.args, n, h * w, c, self.num_classes,
                                   explicit_map=pixel_w is not None) if labels is not None else \
                ops.head_desc(n, h * w, c, self.num_classes)
            want_probs = bool(self.ret_prob or self.ret_pred or self.mode != ModeKeys.TRAIN)
            xent, dice, logits, probs, result = ops.HeadLoss.apply(
                tensor_out, self.params[nm + "/AdjustChannels/weights"], self.params[nm + "/AdjustChannels/biases"],
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

    # ------------------------------------------------------------------ loss
    def _build_loss(self):
        """UNet.py:120-135 (+ tf.losses.get_total_loss(): data loss + L2 regularisers)."""
        xent, dice, _ = self._head
        if self.args.loss_type == "xentropy":
            data_loss = xent
        elif self.args.loss_type == "dice":
            data_loss = dice
        else:
            raise ValueError("Not supported loss_type: {}".format(self.args.loss_type))
        w_reg, _ = self._get_regularizer()
        reg = None
        if w_reg is not None:
            # slim.l2_regularizer(wd)(w) = wd * sum(w^2) / 2 over the regularised buffer; its gradient
            # (wd * w) is applied inside the optimiser kernel, so it is detached here.
            reg = ops.sumsq(self.params.flat["reg"])[0] * (0.5 * w_reg)
        self.loss_terms = {"data": data_loss.detach(), "regularization": reg}
        total = data_loss if reg is None else data_loss + reg
        return total

    # ------------------------------------------------------------------ metrics
    def _build_metrics(self):
        """UNet.py:137-155: per foreground class, metric on thresholded Pred vs one-hot label."""
        if not self.ret_pred:
            return
        if self._inputs.get("labels") is None:
            return
        _, _, result = self._head
        n = self._inputs["images"].shape[0]
        # every foreground class of a metric in one pass; keys in the reference's order (class-major)
        per_class = {met: metrics_from_sums(result, n, self.num_classes, met) for met in self.args.metrics_train}
        for i in range(1, self.num_classes):
            for met in self.args.metrics_train:
                self.metrics_dict["{}/{}".format(self.classes[i], met)] = per_class[met][i - 1]

    def _build_summaries(self):
        """UNet.py:157-176 writes TensorBoard image summaries; out of scope."""
        return
