"""Guided U-Net plugin (spatial-guide path) -- host-side mirror of the reference's NetworksV2/GUNet.py:220-415
executing on the libunetk HIP kernels.

U-Net backbone whose ENCODER conv units are modulated: conv3x3 -> norm (centre/scale per GUNet.yml) ->
+ spatial params slice -> ReLU (`modulated_conv_block`, GUNet.py:162-217), where the spatial params of level
i are a 1x1 conv of the avg-pooled guide with 2*C_i channels (`_spatial_subnets`, :136-159).  On MI355X the
1x1 conv is never materialised: the norm-apply kernel adds guide[n,pix,:] . gw[:, c] + gb[c] on the fly and
the norm-backward kernel reduces the gradients of gw / gb in the same pass.

The context (density) branch --use_context (`_context_subnets` with context_model "fc", GUNet.py:31-60, and
`conditional_normalization`, :119-133,203-206): a 3-layer MLP on the context vector (ops.FullyConnected) yields
per-sample channel gains; each modulated conv unit multiplies its normalised output by its slice of them inside
the same norm-apply / norm-backward kernels (no extra pass over the activations).

`after_affine` (GUNetV2.yml and the *_AA.yml configs; GUNet.py:213-214, slim_nets.channel_wise_affine): the per-channel
gamma' / beta' after the modulation fold into the gains / guide weights / post-shift of the same kernels.

`--dropout` (GUNet.py:189-190): slim.dropout on the normalised output of the FIRST conv of every encoder block, before
the gains / guide term -- a counter-RNG mask regenerated inside the norm kernels (unetk_norm_desc.dropout_keep).
`--fix` (GUNet.py:299-304): the guide's 1x1 convs get norm(scale, eps 1e-3, BN decay .99) + ReLU instead of a bias.  The
conv is linear in the guide, so its batch / instance statistics follow EXACTLY from the guide's first and second moments
(unetk_guide_moments): the norm folds into the guide weights on the host (tiny [g, C] products under autograd) and the
kernels only switch the guide branch's activation to ReLU (guide_leaky with slope 0); nothing is materialised.
`--use_se` (GUNet.py:192-201): the gains of a unit are sigmoid(fc(relu(fc(concat(mean_hw(net), context slice))))) --
ops.Conv3x3NormRelu forms mean_hw(net) from the conv's per-sample statistics and back-propagates through it.
context_model "vgg16B" / "vgg16C" / "vgg16D" (GUNet.py:62-75, slim_nets.py:60-144; ext_config/GUNet_DE_VGG16{B,D}.yml): the
context vector as [bs, L, 1] through 1-D conv + ReLU stacks and "same" max-pools (ops.Conv1d / ops.MaxPool1d, csrc/conv1d.hip),
flattened into mlp(num_base=5) = fc6 ..; the last layer starts at weights 0 / biases 1.
`ct_conv` (`_context_subnets_conv`, GUNet.py:83-116; the nf2 pipeline's [bs, 32, 32, 3] context): three conv units of the model's
own arg_scope, spatial mean (ops.SpatialMean), fully_connected(200) and fully_connected(n_mod), he_normal.
Built as flag combinations of the same kernels (round 3, tests/test_gpu_gunet_combos.py): --fix with --use_context (the guide
branch's ReLU together with the density gains: norm kernels <G, D, L>), after_affine with --without_norm.
Round 3, late: ct_conv with --use_se (the conv subnet emits the plain gain vector, the gate slices it) and after_affine with
--use_se (the affine's gamma multiplies the gate's output inside the op's autograd graph).
--use_se with --dropout: the gate pools the dropped-out values -- two per-sample sums of their own in the forward
(unetk_norm_drop_pool) and a masked extra term in the backward (unetk_norm_se_bwd_add_drop).
after_affine with --fix: the affine's gamma folds into the guide weights THROUGH the guide branch's ReLU by its sign (per-channel
slopes of the activation) and its beta follows behind the activation: unetk_norm_desc.guide_leaky == 3 (the gb block).
--without_norm (GUNet.py:251-252,314-315): every unit = conv + bias (* density gain + guide term) + ReLU, the norm stage
of the fused kernels reduced to the per-channel shift (unetk_norm_desc.affine_only).
"""
import torch

from .. import ops
from ..loss_metrics import build_head_desc, metrics_from_sums, pixel_weights
from ..utils import distribution_utils
from . import base
from .base import ModeKeys, ParamStore
from .padded import PaddedParamStore, pad_to


def n_modulator_params(init_channels, num_down_samples, mod_layers):
    """GUNet.py:47-48: two conv units per modulated level, C_i gains each."""
    return init_channels * sum(2 ** i for i in range(num_down_samples + 1) if i in mod_layers) * 2


def n_modulator_params_se(context_feature_length, num_down_samples, mod_layers):
    """GUNet.py:44-46 (--use_se): every modulated conv unit takes a context_fc_channels[-1]-long slice."""
    return context_feature_length * sum(1 for i in range(num_down_samples + 1) if i in mod_layers) * 2


def vgg_context_layout(model, length, c0):
    """Layers of slim_nets.vgg16B / C / D (slim_nets.py:60-144) on a [bs, length, 1] context, in graph order:
    ("conv", scope under <name>/context, kernel, cin, cout) | ("pool",); and the flattened feature length."""
    reps = {"vgg16B": (2, 2, 2, 2, 2), "vgg16C": (2, 2, 2, 2, 2), "vgg16D": (2, 2, 3, 3, 3)}[model]
    layers, cin, l = [], 1, int(length)
    for g, (rep, mult) in enumerate(zip(reps, (1, 2, 4, 8, 8)), start=1):
        cout = int(c0) * mult
        for j in range(1, rep + 1):                                   # slim.repeat(net, rep, conv_op, cout, 3, scope="conv{g}")
            layers.append(("conv", "conv{0}/conv{0}_{1}".format(g, j), 3, cin, cout))
            cin = cout
        if model == "vgg16C" and g >= 3:                              # conv_op(net, cout, 1, scope="conv{g}_3")
            layers.append(("conv", "conv{}_3".format(g), 1, cin, cout))
        layers.append(("pool",))
        l = (l + 1) // 2                                              # max_pooling1d(2, 2, padding="same")
    return layers, l * cin


VGG_CONTEXT_MODELS = ("vgg16B", "vgg16C", "vgg16D")


def param_specs(in_channels, num_classes, guide_channel, init_channels, num_down_samples, mod_layers, normalizer,
                norm_with_center, norm_with_scale, use_spatial, name, context_dims=None, after_affine=False, mid_cat_g=0,
                without_norm=False, fix=False, se_length=0, context_model="fc", context_conv_init_channels=16):
    """Variables with the reference's TF names: <name>/spatial/conv{i}/{weights,biases},
    <name>/Encode/down_conv{i}/mod_conv{j}/{weights,<Norm>/...}, <name>/Decode/up{i}/{weights,biases},
    <name>/Decode/up_conv{i}/up_conv{i}_{j}/..., <name>/AdjustChannels/{weights,biases}."""
    specs = []
    bn = normalizer == "batch_norm"
    ns = "BatchNorm" if bn else "InstanceNorm"

    def norm_vars(scope, c, center, scale):
        if without_norm:                              # GUNet.py:251-252,314-315: conv + bias, no normaliser
            specs.append((scope + "/biases", (c,), "bias"))
            return
        if center:
            specs.append(("{}/{}/beta".format(scope, ns), (c,), "beta"))
        if scale:
            specs.append(("{}/{}/gamma".format(scope, ns), (c,), "gamma"))
        if bn:
            specs.append(("{}/{}/moving_mean".format(scope, ns), (c,), "moving_mean"))
            specs.append(("{}/{}/moving_variance".format(scope, ns), (c,), "moving_var"))

    if context_dims and context_model == "ct_conv":
        # `_context_subnets_conv` (GUNet.py:83-116): three slim.conv2d(3x3) of the model's arg_scope (regulariser, normaliser,
        # ReLU) on the [bs, 32, 32, 3] context, mean over H x W, fully_connected(200) + fully_connected(n_mod), he_normal
        cin = context_dims[0]
        for scope, cout in (("Conv", 64), ("Conv_1", 64), ("Conv_2", 128)):
            specs.append(("{}/context/{}/weights".format(name, scope), (3, 3, cin, cout), "conv_w"))
            norm_vars("{}/context/{}".format(name, scope), cout, True, True)
            cin = cout
        specs.append((name + "/context/fully_connected/weights", (128, 200), "fc_w_he"))
        specs.append((name + "/context/fully_connected/biases", (200,), "fc_b"))
        specs.append((name + "/context/fully_connected_1/weights", (200, context_dims[-1]), "fc_w_he"))
        specs.append((name + "/context/fully_connected_1/biases", (context_dims[-1],), "fc_b"))
    elif context_dims and context_model in VGG_CONTEXT_MODELS:
        # GUNet.py:62-75: 1-D VGG trunk (slim.conv1d: bias, no regulariser) + mlp(num_base=5): fc6.. ; the last layer starts at
        # weights 0 / biases 1 (every gain = 1)
        layers, feat = vgg_context_layout(context_model, context_dims[0], context_conv_init_channels)
        for lay in layers:
            if lay[0] == "conv":
                _, scope, k, ci, co = lay
                specs.append(("{}/context/{}/weights".format(name, scope), (k, ci, co), "conv1d_w"))
                specs.append(("{}/context/{}/biases".format(name, scope), (co,), "fc_b"))
        dims = [feat] + list(context_dims[1:])
        for i in range(1, len(dims)):
            last = i == len(dims) - 1
            specs.append(("{}/context/fc{}/weights".format(name, 5 + i), (dims[i - 1], dims[i]), "fc_w_zero" if last else "fc_w"))
            specs.append(("{}/context/fc{}/biases".format(name, 5 + i), (dims[i],), "fc_b_one" if last else "fc_b"))
    elif context_dims:    # [context length, fc channels ..., n_modulator_param]: <name>/context/fc{i}/{weights,biases}
        for i in range(1, len(context_dims)):
            last = i == len(context_dims) - 1
            specs.append(("{}/context/fc{}/weights".format(name, i), (context_dims[i - 1], context_dims[i]),
                          "fc_w_he" if last else "fc_w"))
            specs.append(("{}/context/fc{}/biases".format(name, i), (context_dims[i],), "fc_b"))
    if use_spatial:
        for i in range(num_down_samples + 1):
            if i in mod_layers:
                c2 = 2 * init_channels * 2 ** i
                specs.append(("{}/spatial/conv{}/weights".format(name, i + 1), (1, 1, guide_channel, c2), "conv_w"))
                if fix:                     # GUNet.py:299-304: normalizer_fn set -> no bias; centre + scale
                    gs = "{}/spatial/conv{}/{}".format(name, i + 1, ns)
                    specs.append((gs + "/beta", (c2,), "beta"))
                    specs.append((gs + "/gamma", (c2,), "gamma"))
                    if bn:
                        specs.append((gs + "/moving_mean", (c2,), "moving_mean"))
                        specs.append((gs + "/moving_variance", (c2,), "moving_var"))
                else:
                    specs.append(("{}/spatial/conv{}/biases".format(name, i + 1), (c2,), "bias"))
    cin = in_channels
    for i in range(num_down_samples + 1):
        c = init_channels * 2 ** i
        mod = (use_spatial or bool(context_dims)) and i in mod_layers
        for j in (1, 2):
            scope = "{}/Encode/down_conv{}/mod_conv{}".format(name, i + 1, j)
            specs.append((scope + "/weights", (3, 3, cin, c), "conv_w"))
            if mod:                                   # GUNet.py:317-320: no centre / scale under after_affine
                norm_vars(scope, c, norm_with_center and not after_affine, norm_with_scale and not after_affine)
            else:
                norm_vars(scope, c, True, True)       # GUNet.py:183-188: scale=True (BN) / IN defaults
            if after_affine:                          # slim_nets.channel_wise_affine (slim_nets.py:152-212), GUNet.py:213-214
                specs.append((scope + "/ChannelWiseAffine/beta", (c,), "beta"))
                specs.append((scope + "/ChannelWiseAffine/gamma", (c,), "gamma"))
            if se_length and bool(context_dims) and i in mod_layers:      # GUNet.py:196-199: the SE gate's two slim.fully_connected
                hid = (c + se_length) // 4
                specs.append((scope + "/fully_connected/weights", (c + se_length, hid), "fc_w"))
                specs.append((scope + "/fully_connected/biases", (hid,), "fc_b"))
                specs.append((scope + "/fully_connected_1/weights", (hid, c), "fc_w"))
                specs.append((scope + "/fully_connected_1/biases", (c,), "fc_b"))
            cin = c
        if i == 0 and mid_cat_g:            # UNetInter --mid_cat: the guide joins the pooled level-0 output (UNetInter.py:124-127)
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
            norm_vars(scope, c, True, True)
    specs.append((name + "/AdjustChannels/weights", (1, 1, c, num_classes), "conv_w"))
    specs.append((name + "/AdjustChannels/biases", (num_classes,), "bias"))
    return specs


class GUNet(base.BaseNet):
    def __init__(self, args, name=None):
        """Don't create tensors in __init__() (reference GUNet.py:221-238)."""
        super(GUNet, self).__init__(args)
        self.name = name or "GUNet"
        self.classes.extend(self.args.classes)
        self.bs = distribution_utils.per_device_batch_size(args.batch_size, args.num_gpus)
        self.height = args.im_height
        self.width = args.im_width
        self.channel = args.im_channel
        self.use_context_guide = getattr(args, "use_context", False)
        self.use_spatial_guide = getattr(args, "use_spatial", False)
        self.side_dropout = getattr(args, "side_dropout", 0.5)
        self.dropout = getattr(args, "dropout", None)
        self.use_se = getattr(args, "use_se", False)
        self.ct_conv = hasattr(args, "ct_conv")                      # GUNet.py:236: the flag's presence, not its value
        self._taps = None
        self._concat_guide = False          # UNetInter: the guide joins the input channels instead of modulating
        self._mid_cat = False               # UNetInter --mid_cat: ... or the pooled level-0 output
        self._encoder_decay = None          # UNetInter: BN decay of every encoder unit

    def _net_arg_scope(self, *args, **kwargs):
        """GUNet.py:240-257: as UNet (decoder norm = _get_normalization defaults), pools with SAME."""
        self._norm = ("none", {}) if getattr(self.args, "without_norm", False) else self._get_normalization()
        return self._norm

    def _spec(self, decay=None):
        kind, np_ = self._norm
        if kind == "none":              # --without_norm: the norm stage is the conv bias (unetk_norm_desc.affine_only)
            return ops.NormSpec("none", 0.0, 0.0, self.is_training, self.compute_bf16)
        if kind == "batch_norm":
            return ops.NormSpec("batch_norm", np_["eps"], decay if decay is not None else np_["decay"],
                                bool(np_["is_training"]), self.compute_bf16)
        return ops.NormSpec("instance_norm", np_["eps"], 0.0, self.is_training, self.compute_bf16)

    def _unit(self, x, scope, spec, out=None, guide=None, gw=None, gb=None, den=None, se_feat=None, pool=False):
        """One conv unit.  pool=True (the block's second conv, whose activation feeds the pool and the skip): a PLAIN unit -- no
        guide, gains, gate or dropout, e.g. level 0 of the shipped GUNet configs (mod_layers [1, 2, 3, 4]) -- returns
        (pooled, activation) from ONE node (ops.Conv3x3NormReluPool: the pool rides on the norm passes); any other unit takes
        ops.MaxPoolSkip behind it as before."""
        p = self.params
        if pool:
            plain_unit = guide is None and gb is None and den is None and se_feat is None and getattr(spec, "se", None) is None \
                and getattr(spec, "dropout", None) is None
            if not plain_unit:
                z = self._unit(x, scope, spec, out, guide, gw, gb, den, se_feat)
                pooled, skip = ops.MaxPoolSkip.apply(z)
                return pooled, skip
            ns = scope + ("/BatchNorm" if spec.kind == "batch_norm" else "/InstanceNorm")
            var = (None, p[scope + "/biases"], None, None) if spec.kind == "none" else \
                (p.get(ns + "/gamma"), p.get(ns + "/beta"), p.get(ns + "/moving_mean"), p.get(ns + "/moving_variance"))
            pooled, z = ops.Conv3x3NormReluPool.apply(x, p[scope + "/weights"], var[0], var[1], var[2], var[3], spec, out)
            if self._taps is not None:
                self._taps[scope] = z
            return pooled, z
        if spec.kind == "none":
            z = ops.Conv3x3NormRelu.apply(x, p[scope + "/weights"], None, p[scope + "/biases"], None, None, spec, out,
                                          guide, gw, gb, den, 1, se_feat)
        else:
            ns = scope + ("/BatchNorm" if spec.kind == "batch_norm" else "/InstanceNorm")
            z = ops.Conv3x3NormRelu.apply(x, p[scope + "/weights"], p.get(ns + "/gamma"), p.get(ns + "/beta"),
                                          p.get(ns + "/moving_mean"), p.get(ns + "/moving_variance"), spec, out, guide,
                                          gw, gb, den, 1, se_feat)
        if self._taps is not None:
            self._taps[scope] = z
        return z

    def _se_gate(self, scope):
        """GUNet.py:192-201: (pooled [N, C], context slice [N, L]) -> sigmoid(fc(relu(fc(concat(...))))); the two
        slim.fully_connected live under the conv unit's scope."""
        p = self.params

        def gate(pooled, ctx_feat):
            h = torch.cat((pooled, ctx_feat), dim=1)
            h = ops.FullyConnected.apply(h, p[scope + "/fully_connected/weights"], p[scope + "/fully_connected/biases"], 1, None, 0)
            return ops.FullyConnected.apply(h, p[scope + "/fully_connected_1/weights"], p[scope + "/fully_connected_1/biases"],
                                            2, None, 0)
        return gate

    def _fixed_guide(self, level, j, c, guide, gw, spec):
        """--fix (GUNet.py:299-304): sp = relu(norm(conv1x1(guide))) with scale + centre, eps 1e-3 (BN decay .99) on the
        2C-channel conv of the level.  The conv has no bias and is linear in the guide, so per statistics group
            mean_c = E[g] . gw[:, c],   var_c = gw[:, c]^T Cov(g) gw[:, c]
        exactly; the norm folds into the weights: gw' = gw * A, gb' = beta - mean * A with A = gamma / sqrt(var + eps), and
        the kernels apply the ReLU (guide_leaky, slope 0).  Returns (gw', gb') -- [g, C] / [C], or [N, g, C] / [N, C] under
        instance norm.  Batch norm keeps moving statistics of the conv output as slim.batch_norm would."""
        p, nm = self.params, self.name
        kind, _ = self._norm
        bn = kind == "batch_norm"
        ns = "{}/spatial/conv{}/{}".format(nm, level + 1, "BatchNorm" if bn else "InstanceNorm")
        sl = slice((j - 1) * c, j * c)
        gamma, beta = p[ns + "/gamma"][sl], p[ns + "/beta"][sl]
        g_ch = guide.shape[-1]
        eps = 1e-3
        training = self.mode == ModeKeys.TRAIN
        if bn and not training:
            mean_c, var_c = p[ns + "/moving_mean"][sl], p[ns + "/moving_variance"][sl]
        else:
            mom = ops.guide_moments(guide, not bn)                                   # [groups, g + g*g]
            mg = mom[:, :g_ch]
            cov = mom[:, g_ch:].reshape(-1, g_ch, g_ch) - mg[:, :, None] * mg[:, None, :]
            mean_c = mg @ gw                                                          # [groups, C]
            var_c = torch.einsum("gc,kgh,hc->kc", gw, cov, gw).clamp_min(0.0)
            if bn:
                with torch.no_grad():
                    cnt = float(guide.numel() // g_ch)
                    mm, mv = p[ns + "/moving_mean"][sl], p[ns + "/moving_variance"][sl]
                    mm.mul_(0.99).add_(mean_c[0].detach(), alpha=0.01)
                    mv.mul_(0.99).add_(var_c[0].detach() * (cnt / max(cnt - 1.0, 1.0)), alpha=0.01)
                mean_c, var_c = mean_c[0], var_c[0]
        a = gamma * torch.rsqrt(var_c + eps)
        spec.guide_leaky, spec.guide_alpha = True, 0.0
        if bn:
            return (gw * a).contiguous(), (beta - mean_c * a).contiguous()
        spec.guide_per_sample = True
        return (gw[None, :, :] * a[:, None, :]).contiguous(), (beta - mean_c * a).contiguous()

    def _build_network(self, *args, **kwargs):
        base_channels = kwargs.get("init_channels", 64)
        nds = kwargs.get("num_down_samples", 4)
        mod_layers = list(kwargs.get("mod_layers", []))
        norm_with_center = kwargs.get("norm_with_center", False)
        norm_with_scale = kwargs.get("norm_with_scale", False)
        after_affine = bool(kwargs.get("after_affine", False))
        images = self._inputs["images"]
        if not images.is_cuda:
            raise ops._abi.UnetkError("GUNet runs on the GPU only: move `images` to cuda (no CPU path)")
        n, h, w, _ = images.shape
        if h % (1 << nds) or w % (1 << nds):
            raise ValueError("H and W must be divisible by 2**num_down_samples")
        dev = images.device
        nm = self.name
        g_ch = int(getattr(self.args, "guide_channel", 1)) if (self.use_spatial_guide and not self._concat_guide) else 0
        context_dims = None
        if self.use_context_guide:
            context_model = kwargs.get("context_model", "fc")
            if self.ct_conv:
                context_model = "ct_conv"
            if context_model == "resnet":
                raise NotImplementedError                                                   # GUNet.py:76-77, literally
            if context_model not in ("fc", "ct_conv") and context_model not in VGG_CONTEXT_MODELS:
                raise ValueError("Not supported context model")                             # GUNet.py:78-79
            context = self._inputs["context"]
            if self.ct_conv:
                if context.dim() != 4 or context.shape[0] != n or not context.is_cuda:      # GUNet.py:279: [bs, 32, 32, 3]
                    raise ValueError("ct_conv: context must be a [bs, h, w, c] device tensor, got {}".format(tuple(context.shape)))
            elif context.dim() != 2 or context.shape[0] != n or not context.is_cuda:
                raise ValueError("context must be a [bs, L] device tensor, got {}".format(tuple(context.shape)))
            fc_ch = [] if self.ct_conv else list(kwargs.get("context_fc_channels", [256]))
            # GUNet.py:95-97: the conv subnet always emits the plain gain count (its `use_se` argument is unused); --use_se
            # then cuts context_fc_channels[-1] columns per unit off that vector (GUNet.py:193-194)
            context_dims = [int(context.shape[-1])] + fc_ch + \
                [n_modulator_params_se(fc_ch[-1], nds, mod_layers) if (self.use_se and not self.ct_conv) else
                 n_modulator_params(base_channels, nds, mod_layers)]
            if self.use_se and self.ct_conv:
                need = n_modulator_params_se(list(kwargs.get("context_fc_channels", [256]))[-1], nds, mod_layers)
                if need > context_dims[-1]:      # tf.slice past the end of the gain vector
                    raise ValueError("--use_se slices {} columns off a context vector of {}".format(need, context_dims[-1]))
        if self.params is None:
            in_ch = self.channel * (3 if getattr(self.args, "img_grad", False) else 1)      # GUNet.py:335-338
            gc = int(getattr(self.args, "guide_channel", 1))
            if self._concat_guide and not self._mid_cat:                                    # UNetInter.py:87-88
                in_ch = self.channel + gc
            mid_g = gc if (self._concat_guide and self._mid_cat) else 0
            specs = param_specs(in_ch, self.num_classes, g_ch, base_channels, nds, mod_layers,
                                self.args.normalizer, norm_with_center, norm_with_scale, g_ch > 0, nm,
                                context_dims, after_affine, mid_g, bool(getattr(self.args, "without_norm", False)),
                                fix=bool(getattr(self.args, "fix", False)) and g_ch > 0,
                                se_length=(list(kwargs.get("context_fc_channels", [256]))[-1]
                                           if (self.use_se and self.use_context_guide) else 0),
                                context_model="ct_conv" if self.ct_conv else kwargs.get("context_model", "fc"),
                                context_conv_init_channels=int(kwargs.get("context_conv_init_channels", 16)))
            if mid_g:
                # Encode2's first conv sees 64 + g channels: padded with zero filter rows to the filter-gradient tile (32)
                wname = "{}/Encode/down_conv2/mod_conv1/weights".format(nm)
                cin_l = base_channels + mid_g
                self._mid_pad = pad_to(cin_l, 32)
                pads = {wname: ((3, 3, self._mid_pad, 2 * base_channels), {2: [(0, cin_l, 0)]})}
                self.params = PaddedParamStore(specs, pads, dev, bias_decay=getattr(self.args, "bias_decay", False))
            else:
                self.params = ParamStore(specs, dev, bias_decay=getattr(self.args, "bias_decay", False))
            self.params.initialize(self._get_initializer()[0], seed=getattr(self.args, "seed", None))
        p = self.params

        with torch.set_grad_enabled(self.mode == ModeKeys.TRAIN):
            # spatial guide pyramid (GUNet.py:136-159): avg-pool between levels; the 1x1 conv is fused downstream
            guides = {}
            if g_ch:
                gs = self._inputs["sp_guide"].to(torch.float32).contiguous()
                if gs.shape != (n, h, w, g_ch):
                    raise ValueError("sp_guide must be [bs, H, W, {}], got {}".format(g_ch, tuple(gs.shape)))
                for i in range(nds + 1):
                    if i in mod_layers:
                        guides[i] = gs
                    if i < nds:
                        gs = ops.avgpool2_fwd(gs)

            # context MLP (GUNet.py:31-60; slim_nets.py:34-57): ReLU fc + dropout per hidden layer, linear last layer
            den_all, den_off = None, 0
            if context_dims and context_dims[-1] > 0:
                den_all = self._inputs["context"].to(torch.float32).contiguous()
                training = self.mode == ModeKeys.TRAIN
                keep = 1.0 - self.side_dropout if (self.side_dropout and training) else None
                self._dropout_calls = getattr(self, "_dropout_calls", 0) + 1
                fc_base, n_fc = 0, len(context_dims) - 1
                cmodel = "ct_conv" if self.ct_conv else kwargs.get("context_model", "fc")
                if cmodel == "ct_conv":
                    # `_context_subnets_conv` (GUNet.py:83-116): conv units of the model's own arg_scope, spatial mean, two FCs
                    t = den_all
                    for cs in ("Conv", "Conv_1", "Conv_2"):
                        t = self._unit(t, "{}/context/{}".format(nm, cs), self._spec())
                    t = ops.SpatialMean.apply(t)
                    t = ops.FullyConnected.apply(t, p[nm + "/context/fully_connected/weights"],
                                                 p[nm + "/context/fully_connected/biases"], 1, None, 0)
                    den_all = ops.FullyConnected.apply(t, p[nm + "/context/fully_connected_1/weights"],
                                                       p[nm + "/context/fully_connected_1/biases"], 0, None, 0)
                    n_fc = 0
                elif cmodel in VGG_CONTEXT_MODELS:
                    # slim_nets.vgg (slim_nets.py:60-144) on tf.expand_dims(context, -1): conv1d + ReLU stacks, "same" pools,
                    # flatten; then mlp(num_base=5) = fc6 .. (GUNet.py:62-75)
                    layers, _ = vgg_context_layout(cmodel, context_dims[0], int(kwargs.get("context_conv_init_channels", 16)))
                    t = den_all.unsqueeze(-1)
                    for lay in layers:
                        if lay[0] == "conv":
                            cs = "{}/context/{}".format(nm, lay[1])
                            t = ops.Conv1d.apply(t, p[cs + "/weights"], p[cs + "/biases"], True)
                        else:
                            t = ops.MaxPool1d.apply(t)
                    den_all = t.reshape(t.shape[0], -1)                                       # slim.flatten
                    fc_base = 5
                for li in range(1, n_fc + 1):
                    last = li == n_fc
                    seed = (int(getattr(self.args, "seed", None) or 1234) * 1000003 + self._dropout_calls * 101 + li)
                    den_all = ops.FullyConnected.apply(
                        den_all, p["{}/context/fc{}/weights".format(nm, fc_base + li)],
                        p["{}/context/fc{}/biases".format(nm, fc_base + li)], not last, None if last else keep, seed)
                self._layers["context_params"] = den_all
            se_len = 0
            if context_dims and self.use_se:      # context_feature_length = context_fc_channels[-1] (GUNet.py:343-345)
                se_len = int(list(kwargs.get("context_fc_channels", [256]))[-1])

            if self._concat_guide:
                gs = self._inputs["sp_guide"].to(torch.float32)
                if gs.shape[:3] != images.shape[:3]:
                    raise ValueError("sp_guide must be [bs, H, W, g], got {}".format(tuple(gs.shape)))
                x = images.to(torch.float32).contiguous() if self._mid_cat else \
                    torch.cat((images.to(torch.float32), gs), dim=-1).contiguous()
            elif getattr(self.args, "img_grad", False):
                x = ops.image_gradients(images.to(torch.float32))
            else:
                x = images.contiguous()
            cats, skips = {}, {}
            hh, ww = h, w
            for i in range(nds + 1):
                c = base_channels * 2 ** i
                mod = i in guides
                dens = den_all is not None and i in mod_layers
                fix = bool(getattr(self.args, "fix", False)) and mod
                training = self.mode == ModeKeys.TRAIN
                for j in (1, 2):
                    spec = self._spec(0.99) if (mod or dens) else self._spec(self._encoder_decay)  # GUNet.py:313-330 decay .99
                    scope = "{}/Encode/down_conv{}/mod_conv{}".format(nm, i + 1, j)
                    out = None
                    if j == 2 and i < nds:
                        cat = torch.empty((n, hh, ww, 2 * c), dtype=self.storage_dtype, device=dev)
                        out = ops.alias(cat, 0, (n, hh, ww, c), cat.stride())
                        cats[i] = cat
                    if j == 1 and self.dropout and training:          # GUNet.py:189-190: between the two convs of a block
                        self._dropout_calls = getattr(self, "_dropout_calls", 0) + 1
                        spec.dropout = (1.0 - float(self.dropout),
                                        int(getattr(self.args, "seed", None) or 1234) * 7919 + self._dropout_calls * 131 + i)
                    den = se_feat = None
                    if dens and self.use_se:                                 # GUNet.py:192-201
                        se_feat = den_all[:, den_off:den_off + se_len].contiguous()
                        den_off += se_len
                        spec.se = self._se_gate(scope)
                    elif dens:                                               # GUNet.py:203-206
                        den = den_all[:, den_off:den_off + c]
                        den_off += c
                    guide = gw = gb = None
                    if mod:
                        guide = guides[i]
                        gw = p["{}/spatial/conv{}/weights".format(nm, i + 1)].view(g_ch, 2 * c)[:, (j - 1) * c:j * c]
                        if fix:
                            gw, gb = self._fixed_guide(i, j, c, guide, gw, spec)
                        else:
                            gb = p["{}/spatial/conv{}/biases".format(nm, i + 1)][(j - 1) * c:j * c]
                    if after_affine and spec.se is not None and not fix:
                        # the gate's gains are formed inside the op: the affine's gamma joins them THERE (a factor behind the
                        # sigmoid, inside the gate's autograd graph, so gamma gets its gradient with the gate's own variables);
                        # guide weights and post-shift fold as below
                        ga, ba = p[scope + "/ChannelWiseAffine/gamma"], p[scope + "/ChannelWiseAffine/beta"]
                        inner = spec.se
                        spec.se = (lambda pooled, feat, _g=inner, _ga=ga: _g(pooled, feat) * _ga)
                        if mod:
                            gw, gb = gw * ga, gb * ga + ba
                        else:
                            gb = ba
                    elif after_affine and fix:
                        # (t * den + relu(sg)) * ga + ba = t * (den ga) + ga relu(sg) + ba with sg = guide . gw + gb the folded
                        # guide branch.  ga joins gw / gb (s = ga sg): ga relu(sg) = relu(s) where ga >= 0 and min(s, 0) where
                        # ga < 0 -- per-channel slopes (1, 0) / (0, 1) of the guide activation -- and ba follows BEHIND it: the
                        # gb block [bias, slope+, slope-, post-shift] of unetk_norm_desc.guide_leaky == 3.  ga, ba and the
                        # guide's own variables get their gradients through these host-side products.
                        ga, ba = p[scope + "/ChannelWiseAffine/gamma"], p[scope + "/ChannelWiseAffine/beta"]
                        if spec.se is not None:
                            # --use_se as well: the gains are the gate's output, formed inside the op -- ga joins them there
                            # (as in the branch above), the guide block is the same
                            inner = spec.se
                            spec.se = (lambda pooled, feat, _g=inner, _ga=ga: _g(pooled, feat) * _ga)
                        else:
                            den = ga.expand(n, c) if den is None else den * ga
                        gw, gb = gw * ga, gb * ga
                        pos = (ga.detach() >= 0).to(torch.float32)
                        gb = torch.stack((gb, pos.expand_as(gb), (1.0 - pos).expand_as(gb), ba.expand_as(gb)), dim=-2).contiguous()
                        spec.guide_post = True
                    elif after_affine:
                        # (t * den + sp) * ga + ba == t * (den ga) + guide . (gw ga) + (gb ga + ba): the channel-wise affine
                        # folds into the gains / guide weights the kernel already takes (tiny [bs, C] / [g, C] products)
                        ga, ba = p[scope + "/ChannelWiseAffine/gamma"], p[scope + "/ChannelWiseAffine/beta"]
                        den = ga.expand(n, c) if den is None else den * ga
                        if mod:
                            gw, gb = gw * ga, gb * ga + ba
                        else:
                            gb = ba
                    fuse_pool = j == 2 and i < nds and not (i == 0 and self._concat_guide and self._mid_cat)
                    if fuse_pool:
                        x, skips[i] = self._unit(x, scope, spec, out, guide, gw, gb, den, se_feat, pool=True)
                    else:
                        x = self._unit(x, scope, spec, out, guide, gw, gb, den, se_feat)
                if i < nds:
                    if i == 0 and self._concat_guide and self._mid_cat:
                        # UNetInter.py:124-129: pool(concat(level-0 output, guide)); zero channels up to the padded width
                        skips[i] = x
                        gs = self._inputs["sp_guide"].to(torch.float32)
                        zpad = torch.zeros((n, hh, ww, self._mid_pad - c - gs.shape[3]), dtype=torch.float32, device=dev)
                        x = ops.MaxPool2x2.apply(torch.cat((x, gs, zpad), dim=-1))
                    # (else: the block's second unit already returned the pooled tensor and the skip, see _unit(pool=True))
                    hh //= 2
                    ww //= 2

            for i in reversed(range(nds)):
                d = "{}/Decode/up{}".format(nm, i + 1)
                x = ops.DeconvConcat.apply(x, p[d + "/weights"], p[d + "/biases"], skips[i], cats[i], self.compute_bf16)
                for j in (1, 2):
                    x = self._unit(x, "{0}/Decode/up_conv{1}/up_conv{1}_{2}".format(nm, i + 1, j), self._spec())

            c = base_channels
            self.ret_prob = kwargs.get("ret_prob", False)
            self.ret_pred = kwargs.get("ret_pred", False)
            labels = self._inputs.get("labels")
            if labels is not None:
                labels = labels.to(torch.int32).contiguous()
            pixel_w = pixel_weights(self.args, self._inputs, labels)
            desc = build_head_desc(self.args, n, h * w, c, self.num_classes, explicit_map=pixel_w is not None) \
                if labels is not None else ops.head_desc(n, h * w, c, self.num_classes)
            want_probs = bool(self.ret_prob or self.ret_pred or self.mode != ModeKeys.TRAIN)
            xent, dice, logits, probs, result = ops.HeadLoss.apply(
                x, p[nm + "/AdjustChannels/weights"], p[nm + "/AdjustChannels/biases"], labels, pixel_w, desc,
                want_probs)
            self._head = (xent, dice, result)
            self._layers["logits"] = logits.view(n, h, w, self.num_classes)
            if want_probs:
                self.probability = probs.view(n, h, w, self.num_classes)
                if self.ret_prob:                                           # GUNet.py:382-384
                    for i in range(1, self.num_classes):
                        self.predictions[self.classes[i] + "Prob"] = self.probability[..., i:i + 1]
                if self.ret_pred:
                    _, preds = ops.head_predict(probs, self.num_classes, want_preds=True)
                    for i in range(1, self.num_classes):
                        obj = self.classes[i] + "Pred"
                        self.predictions[obj] = preds[i - 1].view(n, h, w, 1)
                        self._image_summaries[obj] = self.predictions[obj]

    def _build_loss(self):
        """GUNet.py:394-413: xentropy and/or dice by substring, + L2 regularisers."""
        xent, dice, _ = self._head
        data_loss = None
        if "xentropy" in self.args.loss_type:
            data_loss = xent
        if "dice" in self.args.loss_type:
            data_loss = dice if data_loss is None else data_loss + dice
        if data_loss is None:
            raise ValueError("Not supported loss_type: {}".format(self.args.loss_type))
        w_reg, _ = self._get_regularizer()
        reg = None
        if w_reg is not None:
            reg = ops.sumsq(self.params.flat["reg"])[0] * (0.5 * w_reg)
        self.loss_terms = {"data": data_loss.detach(), "regularization": reg}
        return data_loss if reg is None else data_loss + reg

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
