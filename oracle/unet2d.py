"""2-D U-Net forward / loss / gradients with the reference's TF semantics, on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/NetworksV2/UNet.py:
  _net_arg_scope   :41-56   conv2d -> (no bias) norm -> ReLU; conv2d_transpose -> bias -> ReLU
  _build_network   :58-118  4x[2x conv3x3, maxpool] -> bridge -> 4x[deconv2x2s2, concat(skip, up), 2x conv3x3]
                            -> 1x1 logits (+bias, linear) -> softmax -> Pred = prob > 0.5
  _build_loss      :120-135 xentropy | dice, + tf.losses.get_total_loss() (adds L2 regularisers)
  _build_metrics   :137-155 per class metric on thresholded Pred
and NetworksV2/base.py:128-178 (regulariser / initialiser / normaliser selection).

TF variable names (UNet.py:203-205, slim scoping rules) are used as parameter keys so a
TF checkpoint name map is the identity.
"""
from collections import OrderedDict

import torch

from . import tf_ops, losses


def param_specs(in_channels, num_classes, init_channels=64, num_down_samples=4,
                normalizer="batch_norm", without_norm=False, name="UNet"):
    """Ordered (name, shape, kind) list in graph-construction order.
    kind in {conv_w, deconv_w, bias, gamma, beta, moving_mean, moving_var}."""
    specs = []

    def conv_block(scope, cin, cout):
        specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
        if without_norm:
            specs.append((scope + "/biases", (cout,), "bias"))
        elif normalizer == "batch_norm":
            bn = scope + "/BatchNorm"
            specs.append((bn + "/gamma", (cout,), "gamma"))
            specs.append((bn + "/beta", (cout,), "beta"))
            specs.append((bn + "/moving_mean", (cout,), "moving_mean"))
            specs.append((bn + "/moving_variance", (cout,), "moving_var"))
        else:   # instance_norm: center + scale
            inn = scope + "/InstanceNorm"
            specs.append((inn + "/gamma", (cout,), "gamma"))
            specs.append((inn + "/beta", (cout,), "beta"))

    c = init_channels
    cin = in_channels
    for i in range(num_down_samples):
        s = "{}/Encode{}/Repeat/convolution2d_".format(name, i + 1)
        conv_block(s + "1", cin, c)
        conv_block(s + "2", c, c)
        cin = c
        c *= 2
    conv_block("{}/ED-Bridge/ED-Bridge_1".format(name), cin, c)
    conv_block("{}/ED-Bridge/ED-Bridge_2".format(name), c, c)
    for i in reversed(range(num_down_samples)):
        c //= 2
        d = "{}/Decode{}".format(name, i + 1)
        specs.append((d + "/Conv2d_transpose/weights", (2, 2, c, 2 * c), "deconv_w"))
        specs.append((d + "/Conv2d_transpose/biases", (c,), "bias"))
        conv_block(d + "/Repeat/convolution2d_1", 2 * c, c)
        conv_block(d + "/Repeat/convolution2d_2", c, c)
    specs.append(("{}/AdjustChannels/weights".format(name), (1, 1, c, num_classes), "conv_w"))
    specs.append(("{}/AdjustChannels/biases".format(name), (num_classes,), "bias"))
    return specs


def init_params(specs, seed=1234, dtype=torch.float32):
    """Glorot-uniform weights (base.py:141), zero biases, BN gamma=1 beta=0 mm=0 mv=1."""
    gen = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape, kind in specs:
        if kind == "conv_w":
            rf = shape[0] * shape[1]
            params[name] = tf_ops.xavier_uniform_(shape, rf * shape[2], rf * shape[3], gen, dtype)
        elif kind == "deconv_w":
            # slim conv2d_transpose: fan computed on [kh,kw,Cout,Cin] as fan_in=k*k*Cout, fan_out=k*k*Cin
            rf = shape[0] * shape[1]
            params[name] = tf_ops.xavier_uniform_(shape, rf * shape[2], rf * shape[3], gen, dtype)
        elif kind == "fc_w":          # slim.fully_connected default (slim_nets.py:45): Glorot uniform on [in, out]
            params[name] = tf_ops.xavier_uniform_(shape, shape[0], shape[1], gen, dtype)
        elif kind == "conv1d_w":      # slim.conv1d default initialiser: Glorot uniform, fans k*Cin / k*Cout
            params[name] = tf_ops.xavier_uniform_(shape, shape[0] * shape[1], shape[0] * shape[2], gen, dtype)
        elif kind == "fc_b_one":      # final_biases_initializer=tf.ones_initializer() (GUNet.py:74)
            params[name] = torch.ones(shape, dtype=dtype)
        elif kind == "fc_w_he":       # tf.keras.initializers.he_normal (GUNet.py:59): truncated normal, var 2/fan_in
            std = (2.0 / shape[0]) ** 0.5 / 0.87962566103423978
            v = torch.empty(shape, dtype=torch.float64)
            torch.nn.init.trunc_normal_(v, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
            params[name] = v.to(dtype)
        elif kind in ("gamma", "moving_var"):
            params[name] = torch.ones(shape, dtype=dtype)
        else:
            params[name] = torch.zeros(shape, dtype=dtype)
    return params


TRAINABLE_KINDS = ("conv_w", "deconv_w", "bias", "gamma", "beta", "fc_w", "fc_w_he", "fc_b", "conv1d_w", "fc_w_zero", "fc_b_one")


class UNet2DOracle(object):
    def __init__(self, in_channels, num_classes, init_channels=64, num_down_samples=4,
                 normalizer="batch_norm", without_norm=False, name="UNet",
                 bn_decay=0.999, bn_eps=1e-3, in_eps=1e-6, img_grad=False):
        """img_grad (UNet.py:69-71): the net sees concat(images, dy, dx); in_channels is then 3 x image channels."""
        self.name = name
        self.img_grad = img_grad
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.init_channels = init_channels
        self.num_down_samples = num_down_samples
        self.normalizer = normalizer
        self.without_norm = without_norm
        self.bn_decay, self.bn_eps, self.in_eps = bn_decay, bn_eps, in_eps
        self.specs = param_specs(in_channels, num_classes, init_channels, num_down_samples,
                                 normalizer, without_norm, name)
        self.kinds = {n: k for n, _, k in self.specs}

    # ------------------------------------------------------------------ layers
    def _conv_norm_relu(self, x, p, scope, is_training, new_stats, taps):
        w = p[scope + "/weights"]
        mode = int(getattr(self, "bf16", 0) or 0)     # 0 fp32 | 1 UNETK_BF16 (operands) | 2 UNETK_BF16S (+ bf16 storage)
        if mode and w.shape[2] % 32 == 0 and w.shape[3] % 32 == 0:
            y = tf_ops.conv_same_bf16_operands(x, w)      # 3x3 units with Cin, Cout % 32 == 0 on the bf16 matrix cores
        else:
            y = tf_ops.conv_nd_same(x, w)
        if taps is not None:
            taps[scope + "/conv"] = y
        if mode == 2:
            return self._norm_relu_bf16s(y, p, scope, is_training, new_stats, taps)
        if self.without_norm:
            y = y + p[scope + "/biases"]
        elif self.normalizer == "batch_norm":
            bn = scope + "/BatchNorm"
            y, mm, mv = tf_ops.batch_norm(y, p[bn + "/gamma"], p[bn + "/beta"],
                                          p[bn + "/moving_mean"], p[bn + "/moving_variance"],
                                          is_training, eps=self.bn_eps, decay=self.bn_decay)
            new_stats[bn + "/moving_mean"] = mm
            new_stats[bn + "/moving_variance"] = mv
        else:
            inn = scope + "/InstanceNorm"
            y = tf_ops.instance_norm(y, p[inn + "/gamma"], p[inn + "/beta"], eps=self.in_eps)
        y = torch.relu(y)
        if taps is not None:
            taps[scope] = y
        return y

    def _norm_relu_bf16s(self, y, p, scope, is_training, new_stats, taps):
        """The unit's norm + ReLU with bf16 tensors (tf_ops.norm_relu_bf16s): y is the fp32 accumulator of the conv; the
        moving statistics are updated from it exactly as in the fp32 mode."""
        if self.without_norm:
            z = tf_ops.norm_relu_bf16s(y, None, p[scope + "/biases"], "none")
        elif self.normalizer == "batch_norm":
            bn = scope + "/BatchNorm"
            mm, mv = p[bn + "/moving_mean"], p[bn + "/moving_variance"]
            if is_training:
                axes = tuple(range(y.dim() - 1))
                mean, var = y.detach().mean(dim=axes), y.detach().var(dim=axes, unbiased=False)
                m = y.numel() // y.shape[-1]
                new_stats[bn + "/moving_mean"] = mm * self.bn_decay + mean * (1.0 - self.bn_decay)
                new_stats[bn + "/moving_variance"] = mv * self.bn_decay + var * (m / max(m - 1, 1)) * (1.0 - self.bn_decay)
                z = tf_ops.norm_relu_bf16s(y, p[bn + "/gamma"], p[bn + "/beta"], "batch_norm", self.bn_eps)
            else:
                new_stats[bn + "/moving_mean"], new_stats[bn + "/moving_variance"] = mm, mv
                z = tf_ops.norm_relu_bf16s(y, p[bn + "/gamma"], p[bn + "/beta"], "batch_norm", self.bn_eps, mm, mv)
        else:
            inn = scope + "/InstanceNorm"
            z = tf_ops.norm_relu_bf16s(y, p[inn + "/gamma"], p[inn + "/beta"], "instance_norm", self.in_eps)
        if taps is not None:
            taps[scope] = z
        return z

    def forward(self, p, images, is_training, taps=None):
        """Returns (logits, new_moving_stats).  images: [bs,H,W,C] float."""
        new_stats = OrderedDict()
        n = self.name
        x = torch.cat((images,) + tf_ops.image_gradients(images), dim=-1) if self.img_grad else images
        skips = []
        for i in range(self.num_down_samples):
            s = "{}/Encode{}/Repeat/convolution2d_".format(n, i + 1)
            x = self._conv_norm_relu(x, p, s + "1", is_training, new_stats, taps)
            x = self._conv_norm_relu(x, p, s + "2", is_training, new_stats, taps)
            skips.append(x)
            x = tf_ops.max_pool2x2(x)
            if int(getattr(self, "bf16", 0) or 0) == 2:
                x = tf_ops.store_bf16(x)          # the pooled tensor and its gradient are bf16 in memory
            if taps is not None:
                taps["{}/Encode{}/pool".format(n, i + 1)] = x
        x = self._conv_norm_relu(x, p, n + "/ED-Bridge/ED-Bridge_1", is_training, new_stats, taps)
        x = self._conv_norm_relu(x, p, n + "/ED-Bridge/ED-Bridge_2", is_training, new_stats, taps)
        for i in reversed(range(self.num_down_samples)):
            d = "{}/Decode{}".format(n, i + 1)
            wt = p[d + "/Conv2d_transpose/weights"]
            convt = tf_ops.conv_transpose_bf16_operands if (getattr(self, "bf16", False) and wt.shape[2] % 32 == 0
                                                            and wt.shape[3] % 32 == 0) else tf_ops.conv_transpose_ks
            up = convt(x, wt, (2, 2), bias=p[d + "/Conv2d_transpose/biases"])
            up = torch.relu(up)
            bf16s = int(getattr(self, "bf16", 0) or 0) == 2
            if bf16s:
                up = tf_ops.store_bf16(up)        # written into the bf16 concat buffer
            if taps is not None:
                taps[d + "/up"] = up
            x = torch.cat((skips[i], up), dim=-1)                   # UNet.py:93 skip first
            if bf16s:
                x = tf_ops.store_bf16(x)          # forward no-op; the concat buffer's GRADIENT is stored as bf16
            x = self._conv_norm_relu(x, p, d + "/Repeat/convolution2d_1", is_training, new_stats, taps)
            x = self._conv_norm_relu(x, p, d + "/Repeat/convolution2d_2", is_training, new_stats, taps)
        logits = tf_ops.conv_nd_same(x, p[n + "/AdjustChannels/weights"]) + p[n + "/AdjustChannels/biases"]
        return logits, new_stats

    # ------------------------------------------------------------------ loss
    def regularization_loss(self, p, weight_decay_rate, bias_decay=False):
        """base.py:128-135: l2_regularizer(wd)(w) = wd * sum(w^2)/2 on conv/deconv weights;
        biases get the SAME regulariser unless --bias_decay is given (literal reading)."""
        if not weight_decay_rate or weight_decay_rate <= 0:
            return torch.zeros((), dtype=torch.float32)
        total = 0.0
        for name, _, kind in self.specs:
            if kind in ("conv_w", "deconv_w") or (kind == "bias" and not bias_decay):
                total = total + weight_decay_rate * 0.5 * (p[name].to(torch.float32) ** 2).sum()
        return total

    def loss(self, p, images, labels, loss_type="xentropy", loss_weight_type="none",
             numeric_w=None, proportion_decay=None, weight_decay_rate=0.0, bias_decay=False,
             is_training=True, taps=None):
        logits, new_stats = self.forward(p, images, is_training, taps)
        kw = {}
        if loss_weight_type == "numerical":
            kw["numeric_w"] = numeric_w
        elif loss_weight_type == "proportion" and proportion_decay and proportion_decay > 0:
            kw["proportion_decay"] = proportion_decay
        if loss_type == "xentropy":
            data_loss = losses.weighted_sparse_softmax_cross_entropy(logits, labels, loss_weight_type, **kw)
        elif loss_type == "dice":
            data_loss = losses.sparse_dice_loss(torch.softmax(logits, -1), labels)
        else:
            raise ValueError("Not supported loss_type: {}".format(loss_type))
        reg = self.regularization_loss(p, weight_decay_rate, bias_decay)
        return data_loss + reg, data_loss, logits, new_stats

    def loss_and_grads(self, p, images, labels, **kw):
        """Autograd of the restated forward = gradient golden."""
        q = OrderedDict()
        for name, t in p.items():
            t = t.detach().clone()
            if self.kinds[name] in TRAINABLE_KINDS:
                t.requires_grad_(True)
            q[name] = t
        taps = kw.pop("taps", None)
        total, data_loss, logits, new_stats = self.loss(q, images, labels, taps=taps, **kw)
        total.backward()
        grads = OrderedDict((n, t.grad.detach()) for n, t in q.items() if t.requires_grad)
        return total.detach(), data_loss.detach(), logits.detach(), grads, new_stats

    # ------------------------------------------------------------------ metrics
    def predictions_and_metrics(self, logits, labels, classes, metrics_train=("Dice",)):
        probs = torch.softmax(logits, -1)
        preds = losses.threshold_pred(probs)
        out_pred, out_metric = OrderedDict(), OrderedDict()
        for i in range(1, self.num_classes):
            obj = classes[i]
            out_pred[obj + "Pred"] = preds[i - 1]
            if labels is not None:
                lab = (labels == i).unsqueeze(-1)
                for met in metrics_train:
                    out_metric["{}/{}".format(obj, met)] = losses.METRICS[met](preds[i - 1], lab)
        return probs, out_pred, out_metric
