"""GPU: GUNet flag combinations the reference accepts (NetworksV2/GUNet.py:162-217,299-330) that no shipped script uses
(VERDICT r2 #7): --fix together with --use_context, after_affine together with --without_norm.  Every ingredient is a flag of
the same fused norm kernels (guide branch with ReLU from folded per-sample weights, density gains, affine_only), so these are
parity cases against the oracle, not new kernels.  Still refused (NotImplementedError): --use_se with --dropout (the gate
pools the dropped-out values), after_affine with --fix / --use_se (a ReLU / a gate computed inside the op stands between the
affine and the weights it would fold into), ct_conv with --use_se."""
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
