"""LGNet forward / loss / gradients with the reference's TF semantics.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/NetworksV2/LGNet.py:
  _spatial_subnets  :30-55    per modulated level l (ascending lists, encoder / decoder branch): the guide, average-pooled
                              to the level's resolution (cumulative 2^(l - previous) pools, SAME), through a 1x1 conv with
                              bias and tf.nn.leaky_relu (alpha 0.2), layer_c[l] = 64 * 2^l channels, no normaliser
  merge_guide_act   :132-135  x + sp_params (when the level is modulated), then ReLU
  _build_network    :137-213  encoder level: conv1 = conv + norm + ReLU, conv2 = conv + norm, merge, 2x2 max-pool; bridge the
                              same without the pool; decoder level: conv2d_transpose(2, 2) + bias + ReLU, concat(skip, up),
                              conv1 = conv + norm, merge (decoder branch), conv2 = conv + norm + ReLU; logits 1x1 + bias
  _build_loss       :233-251  'xentropy' and / or 'dice' by substring
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import losses, tf_ops
from .unet2d import TRAINABLE_KINDS

LAYER_C = [64, 128, 256, 512, 1024]


def param_specs(in_channels, num_classes, guide_channel=1, mod_layers=((0, 1), (0, 1)), normalizer="instance_norm",
                use_spatial=True, name="LGNet"):
    specs = []
    bn = normalizer == "batch_norm"
    ns = "BatchNorm" if bn else "InstanceNorm"

    def unit(scope, cin, cout):
        specs.append((scope + "/weights", (3, 3, cin, cout), "conv_w"))
        if bn:
            for leaf, kind in (("gamma", "gamma"), ("beta", "beta"), ("moving_mean", "moving_mean"), ("moving_variance", "moving_var")):
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


class LGNetOracle(object):
    def __init__(self, in_channels, num_classes, guide_channel=1, mod_layers=((0, 1), (0, 1)), normalizer="instance_norm",
                 use_spatial=True, name="LGNet", img_grad=False):
        self.name, self.num_classes, self.normalizer = name, num_classes, normalizer
        self.mod_layers = (tuple(mod_layers[0]), tuple(mod_layers[1]))
        self.use_spatial, self.img_grad = use_spatial, img_grad
        self.specs = param_specs(in_channels, num_classes, guide_channel, self.mod_layers, normalizer, use_spatial, name)
        self.kinds = {n: k for n, _, k in self.specs}

    def _conv_norm(self, x, p, scope, is_training, new_stats):
        y = tf_ops.conv_nd_same(x, p[scope + "/weights"])
        if self.normalizer == "batch_norm":
            ns = scope + "/BatchNorm"
            y, mm, mv = tf_ops.batch_norm(y, p[ns + "/gamma"], p[ns + "/beta"], p[ns + "/moving_mean"],
                                          p[ns + "/moving_variance"], is_training, eps=1e-3, decay=0.999)
            new_stats[ns + "/moving_mean"], new_stats[ns + "/moving_variance"] = mm, mv
        else:
            ns = scope + "/InstanceNorm"
            y = tf_ops.instance_norm(y, p[ns + "/gamma"], p[ns + "/beta"], eps=1e-6)
        return y

    def _sp(self, p, sp_guide, branch, tag):
        """LGNet.py:30-55 for one branch: {level: leaky_relu(conv1x1(pooled guide))}."""
        out, sg, prev = {}, sp_guide, 0
        for l in self.mod_layers[branch]:
            for _ in range(l - prev):
                sg = tf_ops.avg_pool2x2_same(sg)
            prev = l
            w = p["{}/spatial/conv_{}{}/weights".format(self.name, tag, l + 1)]
            b = p["{}/spatial/conv_{}{}/biases".format(self.name, tag, l + 1)]
            out[l] = F.leaky_relu(sg @ w.reshape(w.shape[2], w.shape[3]) + b, 0.2)
        return out

    def forward(self, p, images, sp_guide, is_training):
        n = self.name
        new_stats = OrderedDict()
        sp_e = self._sp(p, sp_guide, 0, "e") if self.use_spatial else {}
        sp_d = self._sp(p, sp_guide, 1, "d") if self.use_spatial else {}
        x = torch.cat((images,) + tf_ops.image_gradients(images), dim=-1) if self.img_grad else images
        skips = []
        for i in range(5):
            scope = "{}/conv_e{}".format(n, i) if i < 4 else n + "/ED-Bridge"
            x = torch.relu(self._conv_norm(x, p, scope + "/conv1", is_training, new_stats))
            x = self._conv_norm(x, p, scope + "/conv2", is_training, new_stats)
            if i in sp_e:
                x = x + sp_e[i]
            x = torch.relu(x)
            if i < 4:
                skips.append(x)
                x = tf_ops.max_pool2x2(x)
        for i in (3, 2, 1, 0):
            d = "{}/conv_d{}".format(n, i)
            up = torch.relu(tf_ops.conv_transpose_ks(x, p[d + "/up/weights"], (2, 2), bias=p[d + "/up/biases"]))
            x = torch.cat((skips[i], up), dim=-1)
            x = self._conv_norm(x, p, d + "/conv1", is_training, new_stats)
            if i in sp_d:
                x = x + sp_d[i]
            x = torch.relu(x)
            x = torch.relu(self._conv_norm(x, p, d + "/conv2", is_training, new_stats))
        logits = tf_ops.conv_nd_same(x, p[n + "/logits/weights"]) + p[n + "/logits/biases"]
        return logits, new_stats

    def regularization_loss(self, p, wd, bias_decay=False):
        total = torch.zeros((), dtype=torch.float32)
        if not wd or wd <= 0:
            return total
        for name, _, kind in self.specs:
            if kind in ("conv_w", "deconv_w") or (kind == "bias" and not bias_decay):
                total = total + wd * 0.5 * (p[name].to(torch.float32) ** 2).sum()
        return total

    def loss(self, p, images, sp_guide, labels, loss_type="xentropy", loss_weight_type="none", numeric_w=None,
             proportion_decay=None, weight_decay_rate=0.0, bias_decay=False, is_training=True):
        logits, new_stats = self.forward(p, images, sp_guide, is_training)
        kw = {}
        if loss_weight_type == "numerical":
            kw["numeric_w"] = numeric_w
        elif loss_weight_type == "proportion" and proportion_decay and proportion_decay > 0:
            kw["proportion_decay"] = proportion_decay
        data_loss, has = 0.0, False
        if "xentropy" in loss_type:
            data_loss = data_loss + losses.weighted_sparse_softmax_cross_entropy(logits, labels, loss_weight_type, **kw)
            has = True
        if "dice" in loss_type:
            data_loss = data_loss + losses.sparse_dice_loss(torch.softmax(logits, -1), labels)
            has = True
        if not has:
            raise ValueError("Not supported loss_type: {}".format(loss_type))
        return data_loss + self.regularization_loss(p, weight_decay_rate, bias_decay), data_loss, logits, new_stats

    def loss_and_grads(self, p, images, sp_guide, labels, **kw):
        q = OrderedDict()
        for name, t in p.items():
            t = t.detach().clone()
            if self.kinds[name] in TRAINABLE_KINDS:
                t.requires_grad_(True)
            q[name] = t
        total, data_loss, logits, new_stats = self.loss(q, images, sp_guide, labels, **kw)
        total.backward()
        grads = OrderedDict((n, t.grad.detach()) for n, t in q.items() if t.requires_grad)
        return total.detach(), data_loss.detach(), logits.detach(), grads, new_stats
