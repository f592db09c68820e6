"""Guided U-Net (spatial-guide and context paths) forward / loss / gradients with the reference's TF semantics.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Follows /root/reference/NetworksV2/GUNet.py:
  _spatial_subnets      :136-159  guide pyramid: per modulated level a 1x1 conv (bias, linear) of the guide,
                                  2*C_i channels, avg_pool2d(2, SAME) between levels
  modulated_conv_block  :162-217  conv3x3 (no activation) -> norm -> [+ spatial params slice] -> ReLU, twice
  _net_arg_scope        :240-257  default norm for plain conv2d (decoder): BN scale=True / IN defaults
  _build_network        :259-392  encoder norm params: center = norm_with_center, scale = norm_with_scale,
                                  BN decay .99 (:313-330); level 0 un-modulated with (scale=True | IN defaults)
                                  (:183-188); decoder = deconv(+bias, ReLU) -> concat(skip, up) -> 2 conv units
  _build_loss           :394-413  'xentropy' and/or 'dice' by substring, + regularisers
  _context_subnets      :31-60    context_model "fc" = slim_nets.mlp (Backbone/slim_nets.py:34-57): ReLU fully_connected
                                  (+ dropout) per hidden width, linear he_normal last layer -> [bs, n_modulator_param]
  conditional_normalization :119-133,203-206  net * den[:, None, None, slice] after the norm, before the spatial add
The dropout masks are an INPUT here (`drop_masks`: one [bs, width] tensor of 0 / 1/keep_prob per hidden layer):
TF's own mask stream is not reproducible, so parity is on the arithmetic given the mask.
--dropout (:189-190): slim.dropout on the normalised output of the first conv of every encoder block; the unit masks are an
INPUT as well (`unit_masks`: {scope: [bs, H, W, C] of 0 / 1/keep_prob}).
--fix (:299-304): the guide's 1x1 convs are conv (no bias) -> norm(scale, centre, eps 1e-3, BN decay .99) -> ReLU, here
computed literally on the materialised 2C-channel tensor.
--use_se (:192-201): gains = sigmoid(fc(relu(fc(concat(mean_hw(net), context slice))))), the context MLP then emits
context_fc_channels[-1] values per modulated conv unit (:44-46).
context_model vgg16B / C / D (:62-75; Backbone/slim_nets.py:60-144) restated in `context_params`; ct_conv is not.
"""
from collections import OrderedDict

import torch

from . import losses, tf_ops
from .unet2d import TRAINABLE_KINDS  # noqa: F401


# Backbone/slim_nets.py:60-144 -- vgg16B / vgg16C / vgg16D on the 1-D context: per group (conv repeats with kernel 3, an
# optional extra kernel-1 conv named conv{g}_3, then max_pooling1d(2, 2, "same")); channels first_layer_channel x
# (1, 2, 4, 8, 8).  slim.repeat(net, r, conv_op, C, 3, scope="conv{g}") names the layers conv{g}/conv{g}_{1..r}.
_VGG1D = {
    "vgg16B": [(2, False), (2, False), (2, False), (2, False), (2, False)],      # slim_nets.py:60-86
    "vgg16C": [(2, False), (2, False), (2, True), (2, True), (2, True)],         # :89-118
    "vgg16D": [(2, False), (2, False), (3, False), (3, False), (3, False)],      # :121-146
}


def vgg1d_scopes(model, first_layer_channel):
    """[(scope, kernel, cin, cout) ... | None for a pool] in graph order."""
    out, cin = [], 1
    for g, ((rep, extra), mult) in enumerate(zip(_VGG1D[model], (1, 2, 4, 8, 8)), start=1):
        cout = first_layer_channel * mult
        for j in range(rep):
            out.append(("conv%d/conv%d_%d" % (g, g, j + 1), 3, cin, cout))
            cin = cout
        if extra:
            out.append(("conv%d_3" % g, 1, cin, cout))
        out.append(None)
    return out


def param_specs(in_channels, num_classes, guide_channel=1, init_channels=64, num_down_samples=4,
                mod_layers=(1, 2, 3, 4), normalizer="instance_norm", norm_with_center=True, norm_with_scale=False,
                name="GUNet", use_spatial=True, context_dims=None, after_affine=False, mid_cat_g=0, without_norm=False,
                fix=False, se_length=0, context_model="fc", context_conv_init_channels=16):
    """context_dims = [context length, fc widths ..., n_modulator_param] enables the context branch.
    without_norm (GUNet.py:251-252,314-315): every conv unit has a bias and no normaliser."""
    specs = []
    bn = normalizer == "batch_norm"
    norm_scope = "BatchNorm" if bn else "InstanceNorm"

    def norm_vars(scope, center, scale):
        if without_norm:
            specs.append((scope + "/biases", None, "bias"))
            return
        if center:
            specs.append(("{}/{}/beta".format(scope, norm_scope), None, "beta"))
        if scale:
            specs.append(("{}/{}/gamma".format(scope, norm_scope), None, "gamma"))
        if bn:
            specs.append(("{}/{}/moving_mean".format(scope, norm_scope), None, "moving_mean"))
            specs.append(("{}/{}/moving_variance".format(scope, norm_scope), None, "moving_var"))

    def fix_shapes(c, start):
        for k in range(start, len(specs)):
            if specs[k][1] is None:
                specs[k] = (specs[k][0], (c,), specs[k][2])

    if context_dims and context_model == "ct_conv":
        # _context_subnets_conv (GUNet.py:83-116): slim.conv2d x 3 under the model's arg_scope (default scopes Conv, Conv_1,
        # Conv_2), reduce_mean over (1, 2), fully_connected(200), fully_connected(n_mod) -- both he_normal
        cin = context_dims[0]
        for scope, cout in (("Conv", 64), ("Conv_1", 64), ("Conv_2", 128)):
            start = len(specs)
            specs.append(("{}/context/{}/weights".format(name, scope), (3, 3, cin, cout), "conv_w"))
            norm_vars("{}/context/{}".format(name, scope), True, True)
            fix_shapes(cout, start)
            cin = cout
        specs.append((name + "/context/fully_connected/weights", (128, 200), "fc_w_he"))
        specs.append((name + "/context/fully_connected/biases", (200,), "fc_b"))
        specs.append((name + "/context/fully_connected_1/weights", (200, context_dims[-1]), "fc_w_he"))
        specs.append((name + "/context/fully_connected_1/biases", (context_dims[-1],), "fc_b"))
    elif context_dims and context_model != "fc":
        # GUNet.py:62-75: slim_nets.vgg(...) then mlp(..., num_base=5) -> fc6, fc7, ..; final layer zeros / ones initialised
        length, feat_c = context_dims[0], 1
        for lay in vgg1d_scopes(context_model, context_conv_init_channels):
            if lay is None:
                length = -(-length // 2)                           # "same" pooling: ceil
                continue
            scope, k, ci, co = lay
            specs.append(("{}/context/{}/weights".format(name, scope), (k, ci, co), "conv1d_w"))
            specs.append(("{}/context/{}/biases".format(name, scope), (co,), "fc_b"))
            feat_c = co
        dims = [length * feat_c] + list(context_dims[1:])
        for i in range(1, len(dims)):
            last = i == len(dims) - 1
            specs.append(("{}/context/fc{}/weights".format(name, 5 + i), (dims[i - 1], dims[i]), "fc_w_zero" if last else "fc_w"))
            specs.append(("{}/context/fc{}/biases".format(name, 5 + i), (dims[i],), "fc_b_one" if last else "fc_b"))
    elif context_dims:
        for i in range(1, len(context_dims)):
            last = i == len(context_dims) - 1
            specs.append(("{}/context/fc{}/weights".format(name, i), (context_dims[i - 1], context_dims[i]),
                          "fc_w_he" if last else "fc_w"))
            specs.append(("{}/context/fc{}/biases".format(name, i), (context_dims[i],), "fc_b"))
    for i in range(num_down_samples + 1):
        if use_spatial and i in mod_layers:
            c2 = 2 * init_channels * 2 ** i
            specs.append(("{}/spatial/conv{}/weights".format(name, i + 1), (1, 1, guide_channel, c2), "conv_w"))
            if fix:                                                    # normalizer_fn set: no bias (GUNet.py:302-304)
                gsc = "{}/spatial/conv{}/{}".format(name, i + 1, norm_scope)
                specs.append((gsc + "/beta", (c2,), "beta"))
                specs.append((gsc + "/gamma", (c2,), "gamma"))
                if bn:
                    specs.append((gsc + "/moving_mean", (c2,), "moving_mean"))
                    specs.append((gsc + "/moving_variance", (c2,), "moving_var"))
            else:
                specs.append(("{}/spatial/conv{}/biases".format(name, i + 1), (c2,), "bias"))
    cin = in_channels
    for i in range(num_down_samples + 1):
        c = init_channels * 2 ** i
        for j in (1, 2):
            scope = "{}/Encode/down_conv{}/mod_conv{}".format(name, i + 1, j)
            specs.append((scope + "/weights", (3, 3, cin, c), "conv_w"))
            start = len(specs)
            if (use_spatial or context_dims) and i in mod_layers:
                norm_vars(scope, norm_with_center and not after_affine, norm_with_scale and not after_affine)   # :317-320
            else:
                norm_vars(scope, True, True)
            if after_affine:                                           # slim_nets.py:152-212 via GUNet.py:213-214
                specs.append((scope + "/ChannelWiseAffine/beta", None, "beta"))
                specs.append((scope + "/ChannelWiseAffine/gamma", None, "gamma"))
            fix_shapes(c, start)
            if se_length and context_dims and i in mod_layers:         # the SE gate's two slim.fully_connected (:196-199)
                hid = (c + se_length) // 4
                specs.append((scope + "/fully_connected/weights", (c + se_length, hid), "fc_w"))
                specs.append((scope + "/fully_connected/biases", (hid,), "fc_b"))
                specs.append((scope + "/fully_connected_1/weights", (hid, c), "fc_w"))
                specs.append((scope + "/fully_connected_1/biases", (c,), "fc_b"))
            cin = c
        if i == 0 and mid_cat_g:                                        # UNetInter --mid_cat (UNetInter.py:124-127)
            cin = c + mid_cat_g
    c = init_channels * 2 ** num_down_samples
    for i in reversed(range(num_down_samples)):
        c //= 2
        d = "{}/Decode/up{}".format(name, i + 1)
        specs.append((d + "/weights", (2, 2, c, 2 * c), "deconv_w"))
        specs.append((d + "/biases", (c,), "bias"))
        for j in (1, 2):
            scope = "{0}/Decode/up_conv{1}/up_conv{1}_{2}".format(name, i + 1, j)
            specs.append((scope + "/weights", (3, 3, 2 * c if j == 1 else c, c), "conv_w"))
            start = len(specs)
            norm_vars(scope, True, True)
            fix_shapes(c, start)
    specs.append((name + "/AdjustChannels/weights", (1, 1, c, num_classes), "conv_w"))
    specs.append((name + "/AdjustChannels/biases", (num_classes,), "bias"))
    return specs


class GUNet2DOracle(object):
    def __init__(self, in_channels, num_classes, guide_channel=1, init_channels=64, num_down_samples=4,
                 mod_layers=(1, 2, 3, 4), normalizer="instance_norm", norm_with_center=True, norm_with_scale=False,
                 name="GUNet", img_grad=False, use_spatial=True, context_length=None, context_fc_channels=(256, 256),
                 after_affine=False, concat_guide=False, encoder_decay=0.999, mid_cat=False, without_norm=False,
                 fix=False, use_se=False, context_model="fc", context_conv_init_channels=16):
        """concat_guide + mod_layers=() + encoder_decay=.99 + name="UNetInter" is the reference's UNetInter
        (NetworksV2/UNetInter.py:76-141): the guide joins the input channels, encoder BN decay .99 (:98-113)."""
        self.name, self.num_classes = name, num_classes
        self.after_affine, self.concat_guide, self.encoder_decay = after_affine, concat_guide, encoder_decay
        self.mid_cat = bool(mid_cat and concat_guide)
        if concat_guide:
            use_spatial, mod_layers = False, ()
        self.use_spatial = use_spatial
        self.context_dims = None
        self.fix = bool(fix and use_spatial)
        self.use_se = bool(use_se and context_length)
        self.se_length = int(list(context_fc_channels)[-1]) if self.use_se else 0
        if context_length:                                             # GUNet.py:44-48
            if self.use_se:
                n_mod = self.se_length * sum(1 for i in range(num_down_samples + 1) if i in mod_layers) * 2
            else:
                n_mod = init_channels * sum(2 ** i for i in range(num_down_samples + 1) if i in mod_layers) * 2
            self.context_dims = [context_length] + list(context_fc_channels) + [n_mod]
            if context_model == "ct_conv":                             # context_length = the context image's channels
                # GUNet.py:95-97: `_context_subnets_conv` sizes its last layer for the PLAIN gains whatever `use_se` says
                # (its use_se argument is unused); --use_se then slices context_fc_channels[-1] columns per unit off it
                n_mod = init_channels * sum(2 ** i for i in range(num_down_samples + 1) if i in mod_layers) * 2
                self.context_dims = [context_length, n_mod]
        self.img_grad = img_grad                                       # GUNet.py:335-338
        self.init_channels, self.nds = init_channels, num_down_samples
        self.mod_layers = tuple(mod_layers)
        self.normalizer = normalizer
        self.without_norm = without_norm
        self.specs = param_specs(in_channels, num_classes, guide_channel, init_channels, num_down_samples, mod_layers,
                                 normalizer, norm_with_center, norm_with_scale, name, use_spatial, self.context_dims,
                                 after_affine, guide_channel if self.mid_cat else 0, without_norm, self.fix, self.se_length,
                                 context_model, context_conv_init_channels)
        self.context_model, self.context_c0 = context_model, context_conv_init_channels
        self.kinds = {n: k for n, _, k in self.specs}

    def _unit(self, x, p, scope, is_training, new_stats, decay, sp=None, den=None, mask=None, se_feat=None):
        y = tf_ops.conv_nd_same(x, p[scope + "/weights"])
        if self.without_norm:
            y = y + p[scope + "/biases"]
        elif self.normalizer == "batch_norm":
            ns = scope + "/BatchNorm"
            y, mm, mv = tf_ops.batch_norm(y, p.get(ns + "/gamma"), p.get(ns + "/beta"), p[ns + "/moving_mean"],
                                          p[ns + "/moving_variance"], is_training, eps=1e-3, decay=decay)
            new_stats[ns + "/moving_mean"], new_stats[ns + "/moving_variance"] = mm, mv
        else:
            ns = scope + "/InstanceNorm"
            y = tf_ops.instance_norm(y, p.get(ns + "/gamma"), p.get(ns + "/beta"), eps=1e-6)
        if mask is not None:                                           # slim.dropout, GUNet.py:189-190 (mask given)
            y = y * mask
        if se_feat is not None:                                        # GUNet.py:192-201
            out = torch.cat((y.mean(dim=(1, 2)), se_feat), dim=-1)
            out = torch.relu(out @ p[scope + "/fully_connected/weights"] + p[scope + "/fully_connected/biases"])
            den = torch.sigmoid(out @ p[scope + "/fully_connected_1/weights"] + p[scope + "/fully_connected_1/biases"])
        if den is not None:                                            # conditional_normalization, GUNet.py:129-133
            y = y * den[:, None, None, :]
        if sp is not None:
            y = y + sp
        if self.after_affine and "/Encode/" in scope:                  # GUNet.py:213-214
            y = y * p[scope + "/ChannelWiseAffine/gamma"] + p[scope + "/ChannelWiseAffine/beta"]
        return torch.relu(y)

    def context_params(self, p, context, drop_masks=None):
        """slim_nets.mlp (slim_nets.py:43-56) as called from GUNet.py:51-60."""
        net = context
        dims = self.context_dims
        base = 0
        if self.context_model == "ct_conv":
            raise RuntimeError("ct_conv runs inside forward() (its convs carry norm statistics)")
        if self.context_model != "fc":
            # slim_nets.vgg16{B,C,D} on tf.expand_dims(context, -1) (GUNet.py:62-75): conv1d SAME + bias + ReLU, "same" pools
            t = context[:, None, :]                                                          # [B, C = 1, L] for torch
            for lay in vgg1d_scopes(self.context_model, self.context_c0):
                if lay is None:
                    t = torch.nn.functional.max_pool1d(t, 2, 2, ceil_mode=True)
                    continue
                scope, k, _, _ = lay
                w = p["{}/context/{}/weights".format(self.name, scope)]                      # TF [k, Cin, Cout]
                t = torch.nn.functional.conv1d(t, w.permute(2, 1, 0), p["{}/context/{}/biases".format(self.name, scope)],
                                               padding=(k - 1) // 2)
                t = torch.relu(t)
            net = t.permute(0, 2, 1).reshape(t.shape[0], -1)                                 # slim.flatten of [B, L, C]
            base = 5
        for li in range(1, len(dims)):
            net = net @ p["{}/context/fc{}/weights".format(self.name, base + li)] + \
                p["{}/context/fc{}/biases".format(self.name, base + li)]
            if li < len(dims) - 1:
                net = torch.relu(net)
                if drop_masks is not None:
                    net = net * drop_masks[li - 1]
        return net

    def forward(self, p, images, sp_guide, is_training, context=None, drop_masks=None, unit_masks=None):
        n = self.name
        new_stats = OrderedDict()
        den_all, den_off = None, 0
        if self.context_dims and self.context_model == "ct_conv":
            t = context                                                # GUNet.py:95-111
            for cs in ("Conv", "Conv_1", "Conv_2"):
                t = self._unit(t, p, "{}/context/{}".format(n, cs), is_training, new_stats, 0.999)
            t = t.mean(dim=(1, 2))
            t = torch.relu(t @ p[n + "/context/fully_connected/weights"] + p[n + "/context/fully_connected/biases"])
            den_all = t @ p[n + "/context/fully_connected_1/weights"] + p[n + "/context/fully_connected_1/biases"]
            self.last_context_params = den_all
        elif self.context_dims:
            den_all = self.context_params(p, context, drop_masks)
        # spatial subnets (GUNet.py:136-159)
        sp_params = {}
        gs = sp_guide
        for i in range(self.nds + 1):
            if self.use_spatial and i in self.mod_layers:
                w = p["{}/spatial/conv{}/weights".format(n, i + 1)]
                sp_i = gs @ w.reshape(w.shape[2], w.shape[3])
                if self.fix:                                           # conv -> norm (eps 1e-3, BN decay .99) -> ReLU
                    sc = "{}/spatial/conv{}".format(n, i + 1)
                    if self.normalizer == "batch_norm":
                        ns = sc + "/BatchNorm"
                        sp_i, mm, mv = tf_ops.batch_norm(sp_i, p[ns + "/gamma"], p[ns + "/beta"], p[ns + "/moving_mean"],
                                                         p[ns + "/moving_variance"], is_training, eps=1e-3, decay=0.99)
                        new_stats[ns + "/moving_mean"], new_stats[ns + "/moving_variance"] = mm, mv
                    else:
                        ns = sc + "/InstanceNorm"
                        sp_i = tf_ops.instance_norm(sp_i, p[ns + "/gamma"], p[ns + "/beta"], eps=1e-3)
                    sp_params[i] = torch.relu(sp_i)
                else:
                    sp_params[i] = sp_i + p["{}/spatial/conv{}/biases".format(n, i + 1)]
            if i < self.nds:
                gs = tf_ops.avg_pool2x2_same(gs)
        x = torch.cat((images,) + tf_ops.image_gradients(images), dim=-1) if self.img_grad else images
        if self.concat_guide and not self.mid_cat:                     # UNetInter.py:87-88
            x = torch.cat((images, sp_guide), dim=-1)
        skips = []
        for i in range(self.nds + 1):
            c = self.init_channels * 2 ** i
            mod = i in self.mod_layers and (self.use_spatial or den_all is not None)
            for j in (1, 2):
                scope = "{}/Encode/down_conv{}/mod_conv{}".format(n, i + 1, j)
                sp = sp_params[i][..., (j - 1) * c:j * c] if (mod and self.use_spatial) else None
                den = se_feat = None
                if mod and den_all is not None and self.use_se:
                    se_feat = den_all[:, den_off:den_off + self.se_length]
                    den_off += self.se_length
                elif mod and den_all is not None:
                    den = den_all[:, den_off:den_off + c]
                    den_off += c
                mask = unit_masks.get(scope) if (unit_masks and j == 1) else None
                x = self._unit(x, p, scope, is_training, new_stats, 0.99 if mod else self.encoder_decay, sp, den, mask,
                               se_feat)
            if i < self.nds:
                skips.append(x)
                if i == 0 and self.mid_cat:                             # UNetInter.py:124-127
                    x = torch.cat((x, sp_guide), dim=-1)
                x = tf_ops.max_pool2x2(x)
        for i in reversed(range(self.nds)):
            d = "{}/Decode/up{}".format(n, i + 1)
            up = torch.relu(tf_ops.conv_transpose_ks(x, p[d + "/weights"], (2, 2), bias=p[d + "/biases"]))
            x = torch.cat((skips[i], up), dim=-1)
            for j in (1, 2):
                x = self._unit(x, p, "{0}/Decode/up_conv{1}/up_conv{1}_{2}".format(n, i + 1, j), is_training,
                               new_stats, 0.999)
        logits = tf_ops.conv_nd_same(x, p[n + "/AdjustChannels/weights"]) + p[n + "/AdjustChannels/biases"]
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
             proportion_decay=None, weight_decay_rate=0.0, bias_decay=False, is_training=True, context=None,
             drop_masks=None, unit_masks=None):
        logits, new_stats = self.forward(p, images, sp_guide, is_training, context, drop_masks, unit_masks)
        kw = {}
        if loss_weight_type == "numerical":
            kw["numeric_w"] = numeric_w
        elif loss_weight_type == "proportion" and proportion_decay and proportion_decay > 0:
            kw["proportion_decay"] = proportion_decay
        data_loss, has = 0.0, False
        if "xentropy" in loss_type:                                   # GUNet.py:399-403 (substring match)
            data_loss = data_loss + losses.weighted_sparse_softmax_cross_entropy(logits, labels, loss_weight_type, **kw)
            has = True
        if "dice" in loss_type:                                       # :404-408
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
