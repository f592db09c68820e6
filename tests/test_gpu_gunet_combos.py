"""GPU: GUNet flag combinations the reference accepts (NetworksV2/GUNet.py:162-217,299-330) that no shipped script uses
(VERDICT r2 #7): --fix together with --use_context, after_affine together with --without_norm.  Every ingredient is a flag of
the same fused norm kernels (guide branch with ReLU from folded per-sample weights, density gains, affine_only), so these are
parity cases against the oracle, not new kernels.  Late round 3: after_affine with --use_se (the affine's gamma joins the gate's
output inside the op's autograd graph) and ct_conv with --use_se (GUNet.py:95-97: the conv subnet emits the plain gain vector,
the gate slices context_fc_channels[-1] columns per unit off it), --use_se with --dropout (the gate pools the dropped-out values:
unetk_norm_drop_pool / unetk_norm_se_bwd_add_drop), after_affine with --fix (the affine's gamma folds into the guide weights
through the guide branch's ReLU by its SIGN -- per-channel slopes of the activation -- and its beta follows behind the
activation: unetk_norm_desc.guide_leaky == 3)."""
import numpy as np
import pytest
import torch

from oracle import gunet2d
from test_gpu_gunet import YML, _setup_variant, _whole_net_check, kwargs_of, make_args, unit_mask_host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("normalizer,g_ch", [("instance_norm", 1), ("batch_norm", 2)])
def test_gunet_fix_with_context_matches_oracle(normalizer, g_ch):
    """relu(t * den + relu(norm(conv1x1(guide)))): the context gains multiply the normalised output, the guide branch carries
    its own norm + ReLU (folded into per-sample weights from the guide's moments)."""
    yml = dict(YML, context_fc_channels=[32, 16])
    args = make_args(normalizer=normalizer, guide_channel=g_ch, fix=True, use_context=True, side_dropout=0.0)
    model, inputs, net, params, tensors = _setup_variant(args, yml, dict(fix=True), ctx_len=10)
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"context": tensors[3]})
    p64 = {k: v.double() for k, v in model.params.state_dict().items()}
    ref, _ = net.forward(p64, tensors[0].double(), tensors[1].double(), False, context=tensors[3].double())
    model(inputs, "eval", **yml)
    assert np.abs(model.layers["logits"].cpu().numpy() - ref.numpy()).max() < 1e-3


@pytest.mark.parametrize("use_context", [False, True])
def test_gunet_after_affine_without_norm_matches_oracle(use_context):
    """(conv + bias) * den + guide term, then the channel-wise affine: folded into gains / guide weights / post-shift exactly as
    with a norm, on the affine_only kernels."""
    yml = dict(YML, after_affine=True, context_fc_channels=[32, 16])
    args = make_args(without_norm=True, use_context=use_context, side_dropout=0.0)
    model, inputs, net, params, tensors = _setup_variant(args, yml, dict(after_affine=True, without_norm=True),
                                                         ctx_len=10 if use_context else 0)
    assert "GUNet/Encode/down_conv1/mod_conv1/ChannelWiseAffine/gamma" in model.params.state_dict()
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"context": tensors[3]} if use_context else {})


@pytest.mark.parametrize("normalizer", ["instance_norm", "batch_norm"])
def test_gunet_after_affine_with_use_se_matches_oracle(normalizer):
    """relu((t * sigmoid(gate) + sp) * gamma' + beta'): gamma' multiplies the gate's output inside the op (and gets its gradient
    from the gate's graph), the guide weights and the post-shift fold on the host."""
    yml = dict(YML, after_affine=True, context_fc_channels=[32, 16])
    args = make_args(normalizer=normalizer, use_context=True, use_se=True, side_dropout=0.0)
    model, inputs, net, params, tensors = _setup_variant(args, yml, dict(after_affine=True, use_se=True), ctx_len=10)
    names = list(model.params.state_dict())
    assert "GUNet/Encode/down_conv2/mod_conv1/ChannelWiseAffine/gamma" in names
    assert "GUNet/Encode/down_conv2/mod_conv1/fully_connected/weights" in names
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"context": tensors[3]})
    g = model.params["GUNet/Encode/down_conv3/mod_conv2/ChannelWiseAffine/gamma"].grad
    assert g is not None and float(g.abs().sum()) > 0


@pytest.mark.parametrize("normalizer", ["instance_norm", "batch_norm"])
def test_gunet_after_affine_with_fix_and_use_se_matches_oracle(normalizer):
    """The triple --fix + --use_se under after_affine: relu((t * sigmoid(gate) + relu(norm(conv1x1(guide)))) * gamma' + beta') with
    gamma' of both signs -- gamma' multiplies the gate's output inside the op, the guide branch takes the per-channel slopes and
    the post-shift (unetk_norm_desc.guide_leaky == 3) as without the gate."""
    yml = dict(YML, after_affine=True, context_fc_channels=[32, 16])
    args = make_args(normalizer=normalizer, fix=True, use_context=True, use_se=True, side_dropout=0.0, im_height=64, im_width=64)
    model, inputs, net, params, tensors = _setup_variant(args, yml, dict(after_affine=True, fix=True, use_se=True), ctx_len=10, size=64)
    for name in list(params):
        if name.endswith("ChannelWiseAffine/gamma"):
            params[name] = params[name].clone()
            params[name][1::2] *= -1.0                     # every other channel: a negative affine scale
    model.params.load_state(params)
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"context": tensors[3]}, grad_tol=5e-2)
    for nm_ in ("GUNet/Encode/down_conv2/mod_conv1/ChannelWiseAffine/gamma", "GUNet/Encode/down_conv2/mod_conv1/ChannelWiseAffine/beta",
                "GUNet/spatial/conv2/weights", "GUNet/Encode/down_conv2/mod_conv1/fully_connected/weights"):
        g = model.params[nm_].grad
        assert g is not None and float(g.abs().sum()) > 0, nm_


def test_gunet_conv_context_subnet_with_use_se_matches_oracle():
    """ct_conv + --use_se: the conv context subnet's last layer keeps the plain gain count (GUNet.py:95-97), every modulated unit
    takes the next context_fc_channels[-1] columns of it as its gate's context feature (GUNet.py:193-194)."""
    from boxsegliver_amd.NetworksV2.GUNet import GUNet
    from boxsegliver_amd.data.synthetic import make_batch, make_guide
    yml = dict(YML, context_fc_channels=[16])
    args = make_args(normalizer="instance_norm", use_context=True, use_spatial=True, use_se=True, side_dropout=0.0)
    args.ct_conv = 1
    images, labels, _ = make_batch(2, 32, 32, 3, 3, 1234)
    guide = make_guide(labels, args.guide_channel, 1234)
    gen = torch.Generator().manual_seed(33)
    context = torch.rand(2, 32, 32, 3, generator=gen)
    model = GUNet(args)
    inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda(),
              "sp_guide": torch.from_numpy(guide).cuda(), "context": context.cuda()}
    model(inputs, "eval", **yml)
    net = gunet2d.GUNet2DOracle(3, 3, guide_channel=1, normalizer="instance_norm", context_length=3, context_model="ct_conv",
                                use_se=True, context_fc_channels=[16])
    assert [(n, tuple(s), k) for n, s, k in net.specs] == [(n, tuple(s), k) for n, s, k in model.params.specs]
    assert model.params["GUNet/context/fully_connected_1/weights"].shape == (200, 64 * (2 + 4 + 8 + 16) * 2)
    assert model.params["GUNet/Encode/down_conv2/mod_conv1/fully_connected/weights"].shape == (128 + 16, (128 + 16) // 4)
    params = {}
    for name, t in model.params.state_dict().items():
        kind = net.kinds[name]
        if kind == "gamma":
            params[name] = 0.5 + torch.rand(t.shape, generator=gen)
        elif kind in ("beta", "bias", "fc_b"):
            params[name] = 0.2 * torch.randn(t.shape, generator=gen)
        elif "spatial" in name:
            params[name] = 0.5 * torch.randn(t.shape, generator=gen)
        else:
            params[name] = t.clone()
    model.params.load_state(params)
    tensors = (torch.from_numpy(images), torch.from_numpy(guide), torch.from_numpy(labels).long(), context)
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"context": context, "drop_masks": None})


@pytest.mark.parametrize("kind", ["instance_norm", "batch_norm"])
def test_se_gate_on_dropped_out_values_against_float64_autograd(kind):
    """The unit itself, where nothing is ill-conditioned: conv -> norm -> dropout mask -> gate(mean_hw(masked)) -> gains -> ReLU
    against float64 autograd with the mask the kernels regenerate.  Every gradient (input, filter, gamma, beta, the gate's own
    weights) to rounding: this is the proof of unetk_norm_drop_pool / unetk_norm_se_bwd_add_drop; the whole-net test below only
    adds the wiring."""
    import math
    import torch.nn.functional as F
    from boxsegliver_amd import ops
    torch.manual_seed(0)
    n, h, w, cin, c = 2, 16, 16, 64, 64
    x = torch.randn(n, h, w, cin, device="cuda", requires_grad=True)
    wt = (torch.randn(3, 3, cin, c, device="cuda") / math.sqrt(9 * cin)).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(c, device="cuda")).requires_grad_(True)
    beta = (0.2 * torch.randn(c, device="cuda")).requires_grad_(True)
    wg = (0.3 * torch.randn(c, c, device="cuda")).requires_grad_(True)
    r = torch.randn(n, h, w, c, device="cuda")
    seed, keep = 12345, 0.7
    spec = ops.NormSpec(kind, 1e-5, 0.9, True, 0)
    spec.dropout = (keep, seed)
    spec.se = lambda pooled, feat: torch.sigmoid(pooled @ wg)
    bn = kind == "batch_norm"
    mm, mv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    z = ops.Conv3x3NormRelu.apply(x, wt, gamma, beta, mm if bn else None, mv if bn else None, spec, None, None, None, None, None, 1, None)
    (z * r).sum().backward()
    got = [t.grad.double().cpu() for t in (x, wt, gamma, beta, wg)]
    m = torch.from_numpy(unit_mask_host(seed, (n, h, w, c), keep)).double()
    xd, wd = x.detach().double().cpu().requires_grad_(True), wt.detach().double().cpu().requires_grad_(True)
    gd, bd = gamma.detach().double().cpu().requires_grad_(True), beta.detach().double().cpu().requires_grad_(True)
    wgd = wg.detach().double().cpu().requires_grad_(True)
    y = F.conv2d(xd.permute(0, 3, 1, 2), wd.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    dims = (0, 1, 2) if bn else (1, 2)
    mu, var = y.mean(dims, keepdim=True), y.var(dims, unbiased=False, keepdim=True)
    v = ((y - mu) / torch.sqrt(var + 1e-5) * gd + bd) * m
    zr = torch.relu(v * torch.sigmoid(v.mean((1, 2)) @ wgd)[:, None, None, :])
    assert float((z.detach().double().cpu() - zr.detach()).abs().max()) < 2e-5
    (zr * r.double().cpu()).sum().backward()
    for name, a, b in zip(("dx", "dw", "dgamma", "dbeta", "dgate"), got, (xd.grad, wd.grad, gd.grad, bd.grad, wgd.grad)):
        assert float((a - b).norm() / b.norm()) < 1e-5, name


@pytest.mark.parametrize("normalizer", ["instance_norm", "batch_norm"])
def test_gunet_use_se_with_dropout_matches_oracle(normalizer):
    """--use_se + --dropout 0.3 through the whole net: the first conv unit of every encoder block is masked after its norm and the
    gate pools the MASKED value (GUNet.py:189-201); the oracle gets the masks the kernels regenerate.  At 64 x 64 (a 4 x 4
    bridge): at 32 x 32 the 2 x 2 instance-norm level turns single ReLU flips into 3-6 % of the whole gradient vector (see
    test_gpu_gunet.py), which says nothing about this combination."""
    yml = dict(YML, context_fc_channels=[32, 16])
    args = make_args(normalizer=normalizer, dropout=0.3, use_context=True, use_se=True, side_dropout=0.0, im_height=64, im_width=64)
    model, inputs, net, params, tensors = _setup_variant(args, yml, dict(use_se=True), ctx_len=10, size=64)
    calls = getattr(model, "_dropout_calls", 0)
    masks = {}
    for i in range(5):                                   # the seeds GUNet._build_network will use in the next training call
        c = 64 * 2 ** i
        seed = int(args.seed) * 7919 + (calls + 1 + i + 1) * 131 + i       # (the context MLP draws one call number first)
        masks["GUNet/Encode/down_conv{}/mod_conv1".format(i + 1)] = torch.from_numpy(
            unit_mask_host(seed, (2, 64 >> i, 64 >> i, c), 0.7))
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"unit_masks": masks, "context": tensors[3]}, grad_tol=5e-2)
    model(inputs, "eval", **yml)                          # no dropout outside training: the plain --use_se path
    p64 = {k: v.double() for k, v in model.params.state_dict().items()}
    ref, _ = net.forward(p64, tensors[0].double(), tensors[1].double(), False, context=tensors[3].double())
    assert np.abs(model.layers["logits"].cpu().numpy() - ref.numpy()).max() < 1e-3


@pytest.mark.parametrize("normalizer,use_context", [("instance_norm", False), ("batch_norm", True), ("instance_norm", True)])
def test_gunet_after_affine_with_fix_matches_oracle(normalizer, use_context):
    """relu((t * den + relu(norm(conv1x1(guide)))) * gamma' + beta') with gamma' of BOTH signs: where gamma' < 0 the folded guide
    branch is min(s, 0) instead of relu(s) (slopes (0, 1)), and beta' must not pass through the guide's ReLU."""
    yml = dict(YML, after_affine=True, context_fc_channels=[32, 16])
    args = make_args(normalizer=normalizer, fix=True, use_context=use_context, side_dropout=0.0, im_height=64, im_width=64)
    model, inputs, net, params, tensors = _setup_variant(args, yml, dict(after_affine=True, fix=True),
                                                         ctx_len=10 if use_context else 0, size=64)
    flipped = 0
    for name in list(params):
        if name.endswith("ChannelWiseAffine/gamma"):
            params[name] = params[name].clone()
            params[name][1::2] *= -1.0                     # every other channel: a negative affine scale
            flipped += 1
    assert flipped == 10
    model.params.load_state(params)
    _whole_net_check(model, inputs, net, params, tensors, args, yml, {"context": tensors[3]} if use_context else {}, grad_tol=5e-2)
    for nm_ in ("GUNet/Encode/down_conv2/mod_conv1/ChannelWiseAffine/gamma", "GUNet/Encode/down_conv2/mod_conv1/ChannelWiseAffine/beta",
                "GUNet/spatial/conv2/weights"):
        g = model.params[nm_].grad
        assert g is not None and float(g.abs().sum()) > 0, nm_
