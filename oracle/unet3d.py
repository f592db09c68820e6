"""3-D U-Net forward / loss / gradients with the reference's TF semantics, on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/NetworksV2/UNet3D.py:
  _ModelConfig.config[4|5] :31-91   per-layer kernels / strides (anisotropic (1,3,3) first, strided convs, no pooling)
  _net_arg_scope           :108-121 conv3d -> normaliser (no bias) -> ReLU
  _build_network           :123-186 channels c = init_channels, doubled per block, capped at max_channels (:155);
                                    decoder: conv3d_transpose(kernel == stride, biases_initializer=None) -> ReLU (no norm,
                                    no bias) (:161-162), concat(skip, up) (:163), two conv3d; 1x1x1 logits + bias (:167);
                                    optional sp_guide concatenated to the input channels (:143-144)
  _build_loss              :188-202 weighted xentropy only (5-D branch of _compute_weights)
TF SAME padding at stride 2 on even sizes pads 0 before / 1 after (tf_ops.conv_nd_same restates it).
"""
from collections import OrderedDict

import torch

from . import losses, tf_ops
from .unet2d import TRAINABLE_KINDS


def model_config(num_pool_layers=4):
    """UNet3D.py:31-91 as an ordered list of (block, [(layer, kernel, stride)])."""
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


def param_specs(in_channels, num_classes, init_channels=30, num_pool_layers=4, max_channels=320,
                normalizer="instance_norm", name="UNet3D"):
    specs = []
    bn = normalizer == "batch_norm"
    ns = "BatchNorm" if bn else "InstanceNorm"

    def norm_vars(scope, c):
        specs.append(("{}/{}/beta".format(scope, ns), (c,), "beta"))
        specs.append(("{}/{}/gamma".format(scope, ns), (c,), "gamma"))
        if bn:
            specs.append(("{}/{}/moving_mean".format(scope, ns), (c,), "moving_mean"))
            specs.append(("{}/{}/moving_variance".format(scope, ns), (c,), "moving_var"))

    c, cin = init_channels, in_channels
    enc_c = {}
    for block, layers in model_config(num_pool_layers):
        if block.startswith("conv_e") or block == "bridge":
            for lname, k, _ in layers:
                scope = "{}/{}/{}".format(name, block, lname)
                specs.append((scope + "/weights", k + (cin, c), "conv_w"))
                norm_vars(scope, c)
                cin = c
            enc_c[block] = c
            c = min(c * 2, max_channels)
        else:
            c = enc_c[block.replace("d", "e")]
            for lname, k, _ in layers:
                scope = "{}/{}/{}".format(name, block, lname)
                if lname == "up":
                    specs.append((scope + "/weights", k + (c, cin), "deconv_w"))     # [kd,kh,kw,Cout,Cin], no bias
                    cin = 2 * c
                else:
                    specs.append((scope + "/weights", k + (cin, c), "conv_w"))
                    norm_vars(scope, c)
                    cin = c
    specs.append((name + "/logits/weights", (1, 1, 1, cin, num_classes), "conv_w"))
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs


def init_params(specs, seed=1234, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape, kind in specs:
        if kind in ("conv_w", "deconv_w"):
            rf = 1
            for s in shape[:-2]:
                rf *= s
            params[name] = tf_ops.xavier_uniform_(shape, rf * shape[-2], rf * shape[-1], gen, dtype)
        elif kind in ("gamma", "moving_var"):
            params[name] = torch.ones(shape, dtype=dtype)
        else:
            params[name] = torch.zeros(shape, dtype=dtype)
    return params


class UNet3DOracle(object):
    def __init__(self, in_channels, num_classes, init_channels=30, num_pool_layers=4, max_channels=320,
                 normalizer="instance_norm", name="UNet3D"):
        self.name, self.num_classes, self.normalizer = name, num_classes, normalizer
        self.init_channels, self.npl, self.max_channels = init_channels, num_pool_layers, max_channels
        self.specs = param_specs(in_channels, num_classes, init_channels, num_pool_layers, max_channels, normalizer, name)
        self.kinds = {n: k for n, _, k in self.specs}

    def _unit(self, x, p, scope, stride, is_training, new_stats, taps):
        y = tf_ops.conv_nd_same(x, p[scope + "/weights"], stride=stride)
        if self.normalizer == "batch_norm":
            ns = scope + "/BatchNorm"
            y, mm, mv = tf_ops.batch_norm(y, p[ns + "/gamma"], p[ns + "/beta"], p[ns + "/moving_mean"],
                                          p[ns + "/moving_variance"], is_training, eps=1e-3, decay=0.999)
            new_stats[ns + "/moving_mean"], new_stats[ns + "/moving_variance"] = mm, mv
        else:
            ns = scope + "/InstanceNorm"
            y = tf_ops.instance_norm(y, p[ns + "/gamma"], p[ns + "/beta"], eps=1e-6)
        y = torch.relu(y)
        if taps is not None:
            taps[scope] = y
        return y

    def forward(self, p, images, is_training, sp_guide=None, taps=None):
        n = self.name
        new_stats = OrderedDict()
        x = images if sp_guide is None else torch.cat((images, sp_guide), dim=-1)      # UNet3D.py:143-144
        end_pts = {}
        for block, layers in model_config(self.npl):
            if block.startswith("conv_e") or block == "bridge":
                for lname, _, stride in layers:
                    x = self._unit(x, p, "{}/{}/{}".format(n, block, lname), stride, is_training, new_stats, taps)
                end_pts[block] = x
            else:
                for lname, k, stride in layers:
                    scope = "{}/{}/{}".format(n, block, lname)
                    if lname == "up":
                        up = torch.relu(tf_ops.conv_transpose_ks(x, p[scope + "/weights"], stride))
                        x = torch.cat((end_pts[block.replace("d", "e")], up), dim=-1)
                    else:
                        x = self._unit(x, p, scope, stride, is_training, new_stats, taps)
        w = p[n + "/logits/weights"]
        logits = x @ w.reshape(w.shape[-2], w.shape[-1]) + p[n + "/logits/biases"]
        return logits, new_stats

    def regularization_loss(self, p, wd, bias_decay=False):
        total = torch.zeros((), dtype=torch.float32)
        if not wd or wd <= 0:
            return total
        for name, _, kind in self.specs:
            if kind in ("conv_w", "deconv_w") or (kind == "bias" and not bias_decay):
                total = total + wd * 0.5 * (p[name].to(torch.float32) ** 2).sum()
        return total

    def loss(self, p, images, labels, sp_guide=None, loss_type="xentropy", loss_weight_type="none", numeric_w=None,
             proportion_decay=None, weight_decay_rate=0.0, bias_decay=False, is_training=True, taps=None):
        logits, new_stats = self.forward(p, images, is_training, sp_guide, taps)
        kw = {}
        if loss_weight_type == "numerical":
            kw["numeric_w"] = numeric_w
        elif loss_weight_type == "proportion" and proportion_decay and proportion_decay > 0:
            kw["proportion_decay"] = proportion_decay
        if "xentropy" not in loss_type:                               # UNet3D.py:193-200
            raise ValueError("Not supported loss_type: {}".format(loss_type))
        data_loss = losses.weighted_sparse_softmax_cross_entropy(logits, labels, loss_weight_type, **kw)
        return data_loss + self.regularization_loss(p, weight_decay_rate, bias_decay), data_loss, logits, new_stats

    def loss_and_grads(self, p, images, labels, **kw):
        q = OrderedDict()
        for name, t in p.items():
            t = t.detach().clone()
            if self.kinds[name] in TRAINABLE_KINDS:
                t.requires_grad_(True)
            q[name] = t
        total, data_loss, logits, new_stats = self.loss(q, images, labels, **kw)
        total.backward()
        grads = OrderedDict((n, t.grad.detach()) for n, t in q.items() if t.requires_grad)
        return total.detach(), data_loss.detach(), logits.detach(), grads, new_stats
