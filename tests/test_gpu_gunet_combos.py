"""GPU: GUNet flag combinations the reference accepts (NetworksV2/GUNet.py:162-217,299-330) that no shipped script uses
(VERDICT r2 #7): --fix together with --use_context, after_affine together with --without_norm.  Every ingredient is a flag of
the same fused norm kernels (guide branch with ReLU from folded per-sample weights, density gains, affine_only), so these are
parity cases against the oracle, not new kernels.  Late round 3: after_affine with --use_se (the affine's gamma joins the gate's
output inside the op's autograd graph) and ct_conv with --use_se (GUNet.py:95-97: the conv subnet emits the plain gain vector,
the gate slices context_fc_channels[-1] columns per unit off it).  Still refused (NotImplementedError): --use_se with --dropout
(the gate pools the dropped-out values), after_affine with --fix (a ReLU stands between the affine and the guide weights)."""
import numpy as np
import pytest
import torch

from oracle import gunet2d
from test_gpu_gunet import YML, _setup_variant, _whole_net_check, kwargs_of, make_args

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
