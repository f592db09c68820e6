"""SmallUNet forward / loss / gradients with the reference's TF semantics.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/NetworksV2/SmallUNet.py:
  _ModelConfig.config[3] :32-59   encoder blocks conv_e0..conv_e3 (two 3x3 convs, the first of e1..e3 with stride 2), a
                                  bridge and conv_d3 with rate-2 atrous convs at the 1/8 resolution, decoder blocks
                                  conv_d2..conv_d0 = conv2d_transpose(2, 2, biases_initializer=None) + concat(skip, up) +
                                  two 3x3 convs; channels 64 .. 1024 times `init_channel_factor`
  _build_network         :93-139  input = concat(images, sp_guide); slim.conv2d units = conv (no bias) + norm + ReLU;
                                  logits = 1x1 conv + bias in scope "logits"
  _build_loss            :150-166 xentropy | dice (+ regularisers)
`images` of forward() is the ALREADY CONCATENATED network input (images, sp_guide).
"""
from collections import OrderedDict

import torch

from . import tf_ops
from .unet2d import UNet2DOracle


def model_config(factor=1.0):
    """[(block, [(layer, out channels, stride, dilation)])] in graph order; layer "up" = the transposed conv."""
    c = lambda v: int(round(v * factor))
    enc = [("conv_e0", [("conv1", c(64), 1, 1), ("conv2", c(64), 1, 1)]),
           ("conv_e1", [("conv1", c(128), 2, 1), ("conv2", c(128), 1, 1)]),
           ("conv_e2", [("conv1", c(256), 2, 1), ("conv2", c(256), 1, 1)]),
           ("conv_e3", [("conv1", c(512), 2, 1), ("conv2", c(512), 1, 1)]),
           ("bridge", [("conv1", c(1024), 1, 2), ("conv2", c(1024), 1, 2)])]
    dec = [("conv_d3", [("conv1", c(512), 1, 2), ("conv2", c(512), 1, 1), ("conv3", c(512), 1, 1)]),
           ("conv_d2", [("up", c(256), 2, 1), ("conv1", c(256), 1, 1), ("conv2", c(256), 1, 1)]),
           ("conv_d1", [("up", c(128), 2, 1), ("conv1", c(128), 1, 1), ("conv2", c(128), 1, 1)]),
           ("conv_d0", [("up", c(64), 2, 1), ("conv1", c(64), 1, 1), ("conv2", c(64), 1, 1)])]
    return enc + dec


def param_specs(in_channels, num_classes, factor=1.0, normalizer="batch_norm", name="SmallUNet", without_norm=False):
    specs = []
    enc_out = {}
    cin = in_channels
    for block, layers in model_config(factor):
        for layer, cout, _, _ in layers:
            scope = "{}/{}/{}".format(name, block, layer)
            if layer == "up":
                specs.append((scope + "/weights", (2, 2, cout, cin), "deconv_w"))          # no bias (:121)
                cin = enc_out[block.replace("d", "e")] + cout
                continue
            specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
            if without_norm:                                                               # :81-82: conv + bias + ReLU
                specs.append((scope + "/biases", (cout,), "bias"))
            elif normalizer == "batch_norm":
                for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"),
                                   ("moving_variance", "moving_var")):
                    specs.append(("{}/BatchNorm/{}".format(scope, leaf), (cout,), kind))
            else:
                specs.append((scope + "/InstanceNorm/gamma", (cout,), "gamma"))
                specs.append((scope + "/InstanceNorm/beta", (cout,), "beta"))
            cin = cout
        if block.startswith("conv_e"):
            enc_out[block] = cin
    specs.append((name + "/logits/weights", (1, 1, cin, num_classes), "conv_w"))
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs


class SmallUNetOracle(UNet2DOracle):
    def __init__(self, in_channels, num_classes, factor=1.0, normalizer="batch_norm", name="SmallUNet", bn_decay=0.999,
                 bn_eps=1e-3, in_eps=1e-6, without_norm=False):
        self.name, self.img_grad = name, False
        self.in_channels, self.num_classes = in_channels, num_classes
        self.factor, self.normalizer, self.without_norm = factor, normalizer, without_norm
        self.bn_decay, self.bn_eps, self.in_eps = bn_decay, bn_eps, in_eps
        self.specs = param_specs(in_channels, num_classes, factor, normalizer, name, without_norm)
        self.kinds = {n: k for n, _, k in self.specs}

    def _unit(self, x, p, scope, stride, dilation, is_training, new_stats):
        y = tf_ops.conv_nd_same(x, p[scope + "/weights"], stride=(stride, stride), dilation=dilation)
        if self.without_norm:
            return torch.relu(y + p[scope + "/biases"])
        if self.normalizer == "batch_norm":
            bn = scope + "/BatchNorm"
            y, mm, mv = tf_ops.batch_norm(y, p[bn + "/gamma"], p[bn + "/beta"], p[bn + "/moving_mean"],
                                          p[bn + "/moving_variance"], is_training, eps=self.bn_eps, decay=self.bn_decay)
            new_stats[bn + "/moving_mean"], new_stats[bn + "/moving_variance"] = mm, mv
        else:
            inn = scope + "/InstanceNorm"
            y = tf_ops.instance_norm(y, p[inn + "/gamma"], p[inn + "/beta"], eps=self.in_eps)
        return torch.relu(y)

    def forward(self, p, images, is_training, taps=None):
        new_stats = OrderedDict()
        x = images
        end_pts = {}
        for block, layers in model_config(self.factor):
            for layer, _, stride, dilation in layers:
                scope = "{}/{}/{}".format(self.name, block, layer)
                if layer == "up":
                    up = torch.relu(tf_ops.conv_transpose_ks(x, p[scope + "/weights"], (2, 2), bias=None))
                    x = torch.cat((end_pts[block.replace("d", "e")], up), dim=-1)           # :124 skip first
                else:
                    x = self._unit(x, p, scope, stride, dilation, is_training, new_stats)
            if block.startswith("conv_e"):
                end_pts[block] = x
        logits = tf_ops.conv_nd_same(x, p[self.name + "/logits/weights"]) + p[self.name + "/logits/biases"]
        return logits, new_stats
