"""3-D U-Net plugin -- host-side mirror of the reference's NetworksV2/UNet3D.py:94-202 on the libunetk HIP kernels.

nnU-Net-like topology (UNet3D.py:31-91,123-186): anisotropic kernels (1,3,3) at the two finest levels, (3,3,3)
below, STRIDED convs instead of pooling (TF SAME: 0 before / 1 after at stride 2 on even sizes), channels
30 -> 60 -> 120 -> 240 -> 320 (cap), decoder conv3d_transpose(kernel == stride, no bias) + ReLU, concat(skip, up),
1x1x1 logits + bias; weighted cross-entropy only (:188-202).

MI355X decisions: NDHWC fp32; every conv3d is a composition of the 2-D fp32-MFMA conv kernel over depth-tap
plane views (csrc/conv3d.hip); variables are channel-padded to the 32-wide MFMA tile on the device
(30->32, 60->64, 120->128, 240->256) while checkpoints keep the TF shapes (NetworksV2/padded.py);
zero-copy concat as in the 2-D nets.
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, metrics_from_sums
from ..utils import distribution_utils
from . import base
from .base import ModeKeys
from .padded import PaddedParamStore, pad_to


def model_config(num_pool_layers=4):
    """UNet3D.py:31-91 `_ModelConfig.config[4|5]`: ordered (block, [(layer, kernel, stride)])."""
    if num_pool_layers not in (4, 5):
        raise KeyError(num_pool_layers)
    k133, k333 = (1, 3, 3), (3, 3, 3)
    enc = [("conv_e0", [("conv1", k133, (1, 1, 1)), ("conv2", k133, (1, 1, 1))]),
           ("conv_e1", [("conv1", k133, (1, 2, 2)), ("conv2", k133, (1, 1, 1))])]
    for i in range(2, num_pool_layers):
        enc.append(("conv_e%d" % i, [("conv1", k333, (1, 2, 2)), ("conv2", k333, (1, 1, 1))]))
    cfg = enc + [("bridge", [("conv1", k333, (2, 2, 2)), ("conv2", k333, (1, 1, 1))])]
    for i in reversed(range(num_pool_layers)):
        up = (2, 2, 2) if i == num_pool_layers - 1 else (1, 2, 2)
        kk = k333 if i >= 2 else k133
        cfg.append(("conv_d%d" % i, [("up", up, up), ("conv1", kk, (1, 1, 1)), ("conv2", kk, (1, 1, 1))]))
    return cfg


def param_specs(in_channels, num_classes, init_channels, num_pool_layers, max_channels, normalizer, name):
    """Logical specs (TF names / shapes: <name>/<block>/<layer>/{weights,<Norm>/{beta,gamma[,moving_*]}},
    <name>/logits/{weights,biases}) and the device padding of each variable."""
    specs, pads = [], {}
    bn = normalizer == "batch_norm"
    ns = "BatchNorm" if bn else "InstanceNorm"

    def vec(scope_name, c, kind):
        specs.append((scope_name, (c,), kind))
        if pad_to(c) != c:
            pads[scope_name] = ((pad_to(c),), {})

    def norm_vars(scope, c):
        vec("{}/{}/beta".format(scope, ns), c, "beta")
        vec("{}/{}/gamma".format(scope, ns), c, "gamma")
        if bn:
            vec("{}/{}/moving_mean".format(scope, ns), c, "moving_mean")
            vec("{}/{}/moving_variance".format(scope, ns), c, "moving_var")

    def in_layout(parts):
        """parts: logical channel counts of the concatenated inputs -> (physical total, axis segments)."""
        segs, lpos, ppos = [], 0, 0
        for c in parts:
            segs.append((lpos, c, ppos))
            lpos += c
            ppos += pad_to(c) if c >= 16 else c          # the raw image / guide channels are not padded
        return ppos, segs

    c = init_channels
    cin_parts = [in_channels]
    enc_c = {}
    for block, layers in model_config(num_pool_layers):
        if block.startswith("conv_e") or block == "bridge":
            for lname, k, _ in layers:
                scope = "{}/{}/{}".format(name, block, lname)
                pcin, segs = in_layout(cin_parts)
                specs.append((scope + "/weights", k + (sum(cin_parts), c), "conv_w"))
                pads[scope + "/weights"] = (k + (pcin, pad_to(c)), {3: segs})
                norm_vars(scope, c)
                cin_parts = [c]
            enc_c[block] = c
            c = min(c * 2, max_channels)
        else:
            c = enc_c[block.replace("d", "e")]
            for lname, k, _ in layers:
                scope = "{}/{}/{}".format(name, block, lname)
                pcin, segs = in_layout(cin_parts)
                if lname == "up":
                    specs.append((scope + "/weights", k + (c, sum(cin_parts)), "deconv_w"))
                    pads[scope + "/weights"] = (k + (pad_to(c), pcin), {4: segs})
                    cin_parts = [c, c]                 # concat(skip, up)
                else:
                    specs.append((scope + "/weights", k + (sum(cin_parts), c), "conv_w"))
                    pads[scope + "/weights"] = (k + (pcin, pad_to(c)), {3: segs})
                    norm_vars(scope, c)
                    cin_parts = [c]
    pcin, segs = in_layout(cin_parts)
    specs.append((name + "/logits/weights", (1, 1, 1, sum(cin_parts), num_classes), "conv_w"))
    pads[name + "/logits/weights"] = ((1, 1, 1, pcin, num_classes), {3: segs})
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs, pads


class UNet3D(base.BaseNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference UNet3D.py:95-106)."""
        super(UNet3D, self).__init__(args)
        self.name = name or "UNet3D"
        self.classes.extend(self.args.classes)
        self.bs = distribution_utils.per_device_batch_size(args.batch_size, args.num_gpus)
        self.depth = args.im_depth
        self.height = args.im_height
        self.width = args.im_width
        self.channel = args.im_channel
        self.use_spatial = getattr(args, "use_spatial", False)
        self._taps = None

    def _net_arg_scope(self, *args, **kwargs):
        """UNet3D.py:108-121: conv3d -> normaliser (no bias) -> ReLU."""
        self._norm = self._get_normalization()
        return self._norm

    def _spec(self):
        kind, np_ = self._norm
        if kind == "batch_norm":
            return ops.NormSpec("batch_norm", np_["eps"], np_["decay"], bool(np_["is_training"]))
        return ops.NormSpec("instance_norm", np_["eps"], 0.0, self.is_training)

    def _unit(self, x, scope, stride, out=None):
        p = self.params
        spec = self._spec()
        ns = scope + ("/BatchNorm" if spec.kind == "batch_norm" else "/InstanceNorm")
        z = ops.Conv3dNormRelu.apply(x, p[scope + "/weights"], p[ns + "/gamma"], p[ns + "/beta"],
                                     p.get(ns + "/moving_mean"), p.get(ns + "/moving_variance"), spec, stride, out)
        if self._taps is not None:
            self._taps[scope] = z
        return z

    def _build_network(self, *args, **kwargs):
        init_channels = kwargs.get("init_channels", 30)
        npl = kwargs.get("num_pool_layers", 4)
        max_channels = kwargs.get("max_channels", 320)
        cfg = model_config(npl)
        images = self._inputs["images"]
        if not images.is_cuda:
            raise ops._abi.UnetkError("UNet3D runs on the GPU only: move `images` to cuda (no CPU path)")
        if images.dim() != 5 or images.shape[4] != self.channel:
            raise ValueError("images must be [bs, D, H, W, {}], got {}".format(self.channel, tuple(images.shape)))
        if self.compute_bf16:
            raise NotImplementedError("--compute_dtype bf16 is built for the 2-D nets (UNet, GUNet); UNet3D runs fp32")
        if getattr(self.args, "img_grad", False):
            # reference UNet3D.py:138-140 unpacks three values from tf.image.image_gradients on a 5-D tensor; that op
            # takes 4-D input and returns (dy, dx), so the reference's own path raises at graph build
            raise ValueError("--img_grad is not defined for UNet3D (tf.image.image_gradients is 4-D only)")
        n, dd, h, w, _ = images.shape
        if h % (1 << npl) or w % (1 << npl) or dd % 2:
            raise ValueError("H, W must be divisible by 2**num_pool_layers and D by 2")
        dev = images.device
        nm = self.name
        x = images.to(torch.float32)
        if self.use_spatial:                                   # UNet3D.py:143-144: guide joins the input channels
            x = torch.cat((x, self._inputs["sp_guide"].to(torch.float32)), dim=-1)
        x = x.contiguous()
        if self.params is None:
            specs, pads = param_specs(x.shape[-1], self.num_classes, init_channels, npl, max_channels,
                                      self.args.normalizer, nm)
            self.params = PaddedParamStore(specs, pads, dev, bias_decay=getattr(self.args, "bias_decay", False))
            self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))
        p = self.params

        with torch.set_grad_enabled(self.mode == ModeKeys.TRAIN):
            c = init_channels
            end_pts = {}
            for block, layers in cfg:
                if block.startswith("conv_e") or block == "bridge":
                    pc = pad_to(c)
                    for li, (lname, _, stride) in enumerate(layers):
                        scope = "{}/{}/{}".format(nm, block, lname)
                        out = None
                        if li == len(layers) - 1 and block != "bridge":
                            # the block's output is a skip: write it straight into its decoder concat buffer
                            shp = ops.conv3d_out_shape(ops.conv3d_desc(x.shape, pc, 1, stride))
                            cat = torch.empty(shp[:4] + (2 * pc,), dtype=torch.float32, device=dev)
                            out = ops.alias(cat, 0, shp[:4] + (pc,), cat.stride())
                            end_pts[block] = {"cat": cat, "c": c}
                        x = self._unit(x, scope, stride, out)
                    if block != "bridge":
                        end_pts[block]["x"] = x
                    c = min(c * 2, max_channels)
                else:
                    enc = end_pts[block.replace("d", "e")]
                    c = enc["c"]
                    for lname, _, stride in layers:
                        scope = "{}/{}/{}".format(nm, block, lname)
                        if lname == "up":
                            x = ops.Deconv3dConcat.apply(x, p[scope + "/weights"], enc["x"], enc["cat"])
                        else:
                            x = self._unit(x, scope, stride)

            # logits + loss head: one fused kernel over the D*H*W voxels (UNet3D.py:167, 188-202)
            self.ret_prob = kwargs.get("ret_prob", False)
            self.ret_pred = kwargs.get("ret_pred", False)
            labels = self._inputs.get("labels")
            if labels is not None:
                labels = labels.to(torch.int32).contiguous()
            nvox = dd * h * w
            pcl = x.shape[-1]
            desc = build_head_desc(self.args, n, nvox, pcl, self.num_classes) if labels is not None else \
                ops.head_desc(n, nvox, pcl, self.num_classes)
            want_probs = bool(self.ret_prob or self.ret_pred or self.mode != ModeKeys.TRAIN)
            xent, dice, logits, probs, result = ops.HeadLoss.apply(
                x.reshape(n, nvox, 1, pcl), p[nm + "/logits/weights"], p[nm + "/logits/biases"], labels, None, desc,
                want_probs)
            self._head = (xent, dice, result)
            self._layers["logits"] = logits.view(n, dd, h, w, self.num_classes)
            if want_probs:
                self.probability = probs.view(n, dd, h, w, self.num_classes)
                if self.ret_prob:
                    for i in range(1, self.num_classes):
                        self.predictions[self.classes[i] + "Prob"] = self.probability[..., i:i + 1]
                if self.ret_pred:
                    _, preds = ops.head_predict(probs, self.num_classes, want_preds=True)
                    for i in range(1, self.num_classes):
                        obj = self.classes[i] + "Pred"
                        self.predictions[obj] = preds[i - 1].view(n, dd, h, w, 1)
                        self._image_summaries[obj] = self.predictions[obj]

    def _build_loss(self):
        """UNet3D.py:188-202: weighted xentropy only, + L2 regularisers."""
        if "xentropy" not in self.args.loss_type:
            raise ValueError("Not supported loss_type: {}".format(self.args.loss_type))
        xent, _, _ = self._head
        w_reg, _ = self._get_regularizer()
        reg = None
        if w_reg is not None:
            reg = ops.sumsq(self.params.flat["reg"])[0] * (0.5 * w_reg)
        self.loss_terms = {"data": xent.detach(), "regularization": reg}
        return xent if reg is None else xent + reg

    def _build_metrics(self):
        if not self.ret_pred or self._inputs.get("labels") is None:
            return
        _, _, result = self._head
        n = self._inputs["images"].shape[0]
        # every foreground class of a metric in one pass; keys in the reference's order (class-major)
        per_class = {met: metrics_from_sums(result, n, self.num_classes, met) for met in self.args.metrics_train}
        for i in range(1, self.num_classes):
            for met in self.args.metrics_train:
                self.metrics_dict["{}/{}".format(self.classes[i], met)] = per_class[met][i - 1]

    def _build_summaries(self):
        return
