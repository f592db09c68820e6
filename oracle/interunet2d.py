"""InterUNet forward / loss / gradients with the reference's TF semantics.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/NetworksV2/InterUNet.py:
  _ModelConfig.config[3] :32-66   two encoders of three blocks (32 / 64 / 128 channels, stride-2 first convs from the
                                  second block on): "image_e*" on concat(images, sp_guide), "inter_e*" on the images
                                  (+ Sobel edges of the middle channel under --img_grad); "merge_e3" on
                                  concat(image_e2, inter_e2): stride-2 conv, conv, two rate-2 convs (512, 512, 1024, 1024);
                                  "conv_d3" three convs (the first rate 2); "conv_d2..0": conv2d_transpose (no bias) then
                                  concat(UP, image skip, inter skip) -- the up-sampled tensor FIRST -- and two convs
  _build_network         :99-170  slim.conv2d = conv (no bias) + norm + ReLU; logits 1x1 + bias in scope "logits"
  name                   :74      the default scope is "SmallUNet" (sic)
`images` of forward() is the pair (x_input, y_input): the already assembled inputs of the two encoders.
"""
from collections import OrderedDict

import torch

from . import tf_ops
from .unet2d import UNet2DOracle

ENC = [("e0", 32, 1), ("e1", 64, 2), ("e2", 128, 2)]


def param_specs(x_channels, y_channels, num_classes, normalizer="batch_norm", name="SmallUNet", without_norm=False, factor=1):
    """factor = init_channel_factor: every layer has round(out * factor) channels (InterUNet.py:121,129,137,148,157)."""
    specs = []
    r = lambda v: int(round(v * factor))

    def unit(scope, cin, cout):
        specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
        if without_norm:                                                                   # :91-92: conv + bias + ReLU
            specs.append((scope + "/biases", (cout,), "bias"))
        elif normalizer == "batch_norm":
            for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"), ("moving_variance", "moving_var")):
                specs.append(("{}/BatchNorm/{}".format(scope, leaf), (cout,), kind))
        else:
            specs.append((scope + "/InstanceNorm/gamma", (cout,), "gamma"))
            specs.append((scope + "/InstanceNorm/beta", (cout,), "beta"))

    for stream, cin0 in (("image", x_channels), ("inter", y_channels)):
        cin = cin0
        for tag, c, _ in ENC:
            unit("{}/{}_{}/conv1".format(name, stream, tag), cin, r(c))
            unit("{}/{}_{}/conv2".format(name, stream, tag), r(c), r(c))
            cin = r(c)
    unit(name + "/merge_e3/conv1", 2 * r(128), r(512))
    unit(name + "/merge_e3/conv2", r(512), r(512))
    unit(name + "/merge_e3/conv3", r(512), r(1024))
    unit(name + "/merge_e3/conv4", r(1024), r(1024))
    unit(name + "/conv_d3/conv1", r(1024), r(512))
    unit(name + "/conv_d3/conv2", r(512), r(512))
    unit(name + "/conv_d3/conv3", r(512), r(512))
    cin = r(512)
    for i, c, skip in ((2, 256, 128), (1, 128, 64), (0, 64, 32)):
        specs.append(("{}/conv_d{}/up/weights".format(name, i), (2, 2, r(c), cin), "deconv_w"))
        unit("{}/conv_d{}/conv1".format(name, i), r(c) + 2 * r(skip), r(c))
        unit("{}/conv_d{}/conv2".format(name, i), r(c), r(c))
        cin = r(c)
    specs.append((name + "/logits/weights", (1, 1, cin, num_classes), "conv_w"))
    specs.append((name + "/logits/biases", (num_classes,), "bias"))
    return specs


class InterUNetOracle(UNet2DOracle):
    def __init__(self, x_channels, y_channels, num_classes, normalizer="batch_norm", name="SmallUNet", without_norm=False,
                 factor=1):
        self.name, self.img_grad, self.num_classes = name, False, num_classes
        self.normalizer, self.without_norm = normalizer, without_norm
        self.bn_decay, self.bn_eps, self.in_eps = 0.999, 1e-3, 1e-6
        self.specs = param_specs(x_channels, y_channels, num_classes, normalizer, name, without_norm, factor)
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
        x, y = images
        n = self.name
        st = OrderedDict()
        ends = {}
        for stream, t in (("image", x), ("inter", y)):
            for tag, _, stride in ENC:
                t = self._unit(t, p, "{}/{}_{}/conv1".format(n, stream, tag), stride, 1, is_training, st)
                t = self._unit(t, p, "{}/{}_{}/conv2".format(n, stream, tag), 1, 1, is_training, st)
                ends[stream + tag] = t
        z = torch.cat((ends["imagee2"], ends["intere2"]), -1)
        z = self._unit(z, p, n + "/merge_e3/conv1", 2, 1, is_training, st)
        z = self._unit(z, p, n + "/merge_e3/conv2", 1, 1, is_training, st)
        z = self._unit(z, p, n + "/merge_e3/conv3", 1, 2, is_training, st)
        z = self._unit(z, p, n + "/merge_e3/conv4", 1, 2, is_training, st)
        z = self._unit(z, p, n + "/conv_d3/conv1", 1, 2, is_training, st)
        z = self._unit(z, p, n + "/conv_d3/conv2", 1, 1, is_training, st)
        z = self._unit(z, p, n + "/conv_d3/conv3", 1, 1, is_training, st)
        for i in (2, 1, 0):
            up = torch.relu(tf_ops.conv_transpose_ks(z, p["{}/conv_d{}/up/weights".format(n, i)], (2, 2), bias=None))
            z = torch.cat((up, ends["imagee%d" % i], ends["intere%d" % i]), -1)          # :155 the up-sampled tensor first
            z = self._unit(z, p, "{}/conv_d{}/conv1".format(n, i), 1, 1, is_training, st)
            z = self._unit(z, p, "{}/conv_d{}/conv2".format(n, i), 1, 1, is_training, st)
        logits = tf_ops.conv_nd_same(z, p[n + "/logits/weights"]) + p[n + "/logits/biases"]
        return logits, st

    def loss(self, p, images, labels, **kw):
        return super(InterUNetOracle, self).loss(p, images, labels, **kw)

    def loss_and_grads(self, p, images, labels, **kw):
        return super(InterUNetOracle, self).loss_and_grads(p, images, labels, **kw)


def sobel_concat(images, ch):
    """tf.image.sobel_edges of channel ch (REFLECT padding, cross-correlation) appended as (dy, dx): InterUNet.py:105-109."""
    import torch.nn.functional as F
    x = images[..., ch:ch + 1].permute(0, 3, 1, 2)
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]], dtype=images.dtype)
    k = torch.stack((ky, ky.t()))[:, None]
    e = F.conv2d(xp, k).permute(0, 2, 3, 1)
    return torch.cat((images, e), -1)
