"""GPU parity of the InterUNet plugin (reference NetworksV2/InterUNet.py: two encoders, merge block with stride-2 and
rate-2 convs, decoder concats with the up-sampled tensor first, optional Sobel input) against the oracle; the Sobel input
kernel against tf.image.sobel_edges restated with torch."""
import numpy as np
import pytest
import torch

from oracle import interunet2d
from test_gpu_gunet import kwargs_of, make_args
from test_gpu_unet import check_deconv_backward, check_unit_backward, rel
from test_gpu_unet3d import check_conv3d_unit

pytestmark = pytest.mark.gpu

YML = dict(init_channel_factor=1, num_pool_layers=3, ret_prob=False, ret_pred=True, build_metrics=True, build_summaries=False)


def test_sobel_concat_kernel():
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(2, 9, 11, 3, generator=gen)
    got = ops.sobel_concat(x.cuda(), 1).cpu()
    ref = interunet2d.sobel_concat(x.double(), 1)
    assert got.shape == (2, 9, 11, 5) and rel(got.numpy(), ref.numpy()) < 1e-6
    # a vertical ramp: dy = 8 (the Sobel weights sum to 4, central difference 2), dx = 0, also on the reflected border rows
    ramp = torch.arange(6, dtype=torch.float32)[None, :, None, None].expand(1, 6, 5, 1).contiguous()
    e = ops.sobel_concat(ramp.cuda(), 0).cpu()
    assert torch.allclose(e[0, 1:-1, :, 1], torch.full((4, 5), 8.0)) and torch.all(e[..., 2] == 0) and torch.all(e[0, 0, :, 1] == 0)


@pytest.mark.parametrize("normalizer,loss_type,img_grad,without_norm", [("batch_norm", "xentropy", True, False),
                                                                          ("instance_norm", "dice", False, False),
                                                                          ("batch_norm", "xentropy", False, True)])
def test_interunet_matches_oracle_and_trains(normalizer, loss_type, img_grad, without_norm):
    from boxsegliver_amd import ops
    from boxsegliver_amd.core import models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data.synthetic import make_batch, make_guide
    zoo = {cls.__name__: cls for cls in models.MODEL_ZOO}
    args = make_args(normalizer=normalizer, loss_type=loss_type, use_spatial=True, guide_channel=1, im_height=64, im_width=64,
                     img_grad=img_grad, without_norm=without_norm)
    images, labels, _ = make_batch(2, 64, 64, 3, 3, 1234)
    guide = make_guide(labels, 1, 1234)
    model = zoo["InterUNet"](args)
    inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda(),
              "sp_guide": torch.from_numpy(guide).cuda()}
    model(inputs, "eval", **YML)
    yc = 5 if img_grad else 3
    net = interunet2d.InterUNetOracle(4, yc, 3, normalizer=normalizer, without_norm=without_norm)
    assert model.name == "SmallUNet"                                       # the reference's default scope (InterUNet.py:74)
    assert [(n, tuple(s), k) for n, s, k in net.specs] == [(n, tuple(s), k) for n, s, k in model.params.specs]
    names = list(model.params.state_dict())
    assert "SmallUNet/inter_e0/conv1/weights" in names and "SmallUNet/merge_e3/conv4/weights" in names
    assert model.params["SmallUNet/inter_e0/conv1/weights"].shape == (3, 3, yc, 32)
    assert ("SmallUNet/merge_e3/conv1/biases" in names) == without_norm
    assert ("SmallUNet/merge_e3/conv1/BatchNorm/gamma" in names) == (not without_norm and normalizer == "batch_norm")
    assert model.params["SmallUNet/conv_d1/conv1/weights"].shape == (3, 3, 256, 128) and "SmallUNet/conv_d2/up/biases" not in names
    gen = torch.Generator().manual_seed(16)
    params = {}
    for name, t in model.params.state_dict().items():
        kind = net.kinds[name]
        if kind == "gamma":
            params[name] = 0.5 + torch.rand(t.shape, generator=gen)
        elif kind in ("beta", "bias"):
            params[name] = 0.2 * torch.randn(t.shape, generator=gen)
        else:
            params[name] = t.clone()
    model.params.load_state(params)
    img, gd, lab = torch.from_numpy(images), torch.from_numpy(guide), torch.from_numpy(labels).long()
    pair = lambda t: (torch.cat((t(img), t(gd)), -1), interunet2d.sobel_concat(t(img), 1) if img_grad else t(img))
    total, _, logits, _, new_stats = net.loss_and_grads(params, pair(lambda v: v), lab, **kwargs_of(args))
    p64 = {k: v.double() for k, v in params.items()}
    _, _, _, grads64, _ = net.loss_and_grads(p64, pair(lambda v: v.double()), lab, **kwargs_of(args))
    ops.DEBUG_CAPTURE = []
    try:
        model.params.zero_grad()
        loss = model(inputs, "train", **YML)
        loss.backward()
        torch.cuda.synchronize()
        captured = ops.DEBUG_CAPTURE
    finally:
        ops.DEBUG_CAPTURE = None
    assert abs(loss.item() - total.item()) < 1e-4 * max(1.0, abs(total.item()))
    assert np.abs(model.layers["logits"].cpu().numpy() - logits.numpy()).max() < 1e-3
    units2d = [c for c in captured if c.get("kind") not in ("deconv", "conv3d")]
    strided = [c for c in captured if c.get("kind") == "conv3d"]
    deconvs = [c for c in captured if c.get("kind") == "deconv"]
    assert len(units2d) == 20 and len(strided) == 5 and len(deconvs) == 3
    assert sum(1 for c in units2d if c.get("dilation") == 2) == 3
    assert all(bool(c.get("plain")) == without_norm for c in units2d + strided)
    for c in units2d:
        check_unit_backward(c)
    for c in strided:
        check_conv3d_unit(c)
    for c in deconvs:
        assert c["coff"] == 0 and c["b"] is None
        check_deconv_backward(c)
    num = den = 0.0
    for name in model.params.trainable_names():
        g = model.params[name].grad.cpu().numpy().astype(np.float64)
        ref = grads64[name].numpy()
        num += np.sum((g - ref) ** 2)
        den += np.sum(ref ** 2)
    assert (num / den) ** 0.5 < 1e-2
    for name, ref in new_stats.items():
        np.testing.assert_allclose(model.params[name].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-6)
    solver = Solver(args)
    first = None
    for _ in range(4):
        loss = model(inputs, "train", **YML)
        first = loss.item() if first is None else first
        solver(loss, model)
    assert model(inputs, "train", **YML).item() < first


@pytest.mark.parametrize("factor,normalizer", [(0.75, "instance_norm"), (0.5, "batch_norm")])
def test_interunet_channel_factor_on_padded_variables(factor, normalizer):
    """init_channel_factor != 1 (`round(layer["out"] * c)`, InterUNet.py:121; e.g. --model_config SmallUNet_V2.yml): the
    variables keep the reference's logical shapes (24 / 48 / 96 / 384 ... at 0.75) in checkpoints and are channel-padded
    to multiples of 64 on the device; the padding is exactly zero and stays zero through training."""
    from boxsegliver_amd.core import models
    from boxsegliver_amd.core.solver import Solver
    from boxsegliver_amd.data.synthetic import make_batch, make_guide
    yml = dict(YML, init_channel_factor=factor)
    zoo = {cls.__name__: cls for cls in models.MODEL_ZOO}
    args = make_args(normalizer=normalizer, loss_type="xentropy", use_spatial=True, guide_channel=1, im_height=64, im_width=64,
                     img_grad=False, without_norm=False)
    images, labels, _ = make_batch(2, 64, 64, 3, 3, 1234)
    guide = make_guide(labels, 1, 1234)
    model = zoo["InterUNet"](args)
    inputs = {"images": torch.from_numpy(images).cuda(), "labels": torch.from_numpy(labels).cuda(),
              "sp_guide": torch.from_numpy(guide).cuda()}
    model(inputs, "eval", **yml)
    net = interunet2d.InterUNetOracle(4, 3, 3, normalizer=normalizer, factor=factor)
    r = lambda v: int(round(v * factor))
    logical = {n: tuple(s) for n, s, _ in net.specs}
    assert [(n, tuple(s), k) for n, s, k in net.specs] == [(n, tuple(s), k) for n, s, k in model.params.logical_specs]
    assert logical["SmallUNet/inter_e0/conv1/weights"] == (3, 3, 3, r(32))
    assert logical["SmallUNet/conv_d1/conv1/weights"] == (3, 3, r(128) + 2 * r(64), r(128))
    p64_ = lambda c: (c + 63) // 64 * 64
    assert model.params["SmallUNet/conv_d1/conv1/weights"].shape == (3, 3, p64_(r(128)) + 2 * p64_(r(64)), p64_(r(128)))   # device
    sd = model.params.state_dict()
    assert {n: tuple(t.shape) for n, t in sd.items()} == logical and model.params.num_trainable() == \
        sum(int(np.prod(s)) for n, s, k in net.specs if k not in ("moving_mean", "moving_var"))
    gen = torch.Generator().manual_seed(26)
    params = {}
    for name, t in sd.items():
        kind = net.kinds[name]
        if kind == "gamma":
            params[name] = 0.5 + torch.rand(t.shape, generator=gen)
        elif kind in ("beta", "bias"):
            params[name] = 0.2 * torch.randn(t.shape, generator=gen)
        else:
            params[name] = t.clone()
    model.params.load_state(params)
    img, gd, lab = torch.from_numpy(images), torch.from_numpy(guide), torch.from_numpy(labels).long()
    p64 = {k: v.double() for k, v in params.items()}
    total, _, logits, grads64, new_stats = net.loss_and_grads(p64, (torch.cat((img, gd), -1).double(), img.double()), lab,
                                                              **kwargs_of(args))
    model.params.zero_grad()
    loss = model(inputs, "train", **yml)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - total.item()) < 1e-4 * max(1.0, abs(total.item()))
    assert np.abs(model.layers["logits"].cpu().numpy() - logits.numpy()).max() < 1e-3
    num = den = 0.0
    for name in model.params.trainable_names():
        g = model.params.logical_grad(name).numpy().astype(np.float64)
        ref = grads64[name].numpy()
        num += np.sum((g - ref) ** 2)
        den += np.sum(ref ** 2)
    assert (num / den) ** 0.5 < 1e-2

    def padding_is_zero():
        for name, t in model.params.tensors.items():
            if name in model.params.pads and model.params.where[name][0] != "stats":
                full = float(t.detach().abs().sum())
                inner = sum(float(t.detach()[pidx].abs().sum()) for _, pidx in model.params._blocks(name))
                assert abs(full - inner) <= 1e-6 * max(full, 1.0), name
    padding_is_zero()
    solver = Solver(args)
    first = None
    for _ in range(4):
        loss = model(inputs, "train", **yml)
        first = loss.item() if first is None else first
        solver(loss, model)
    assert model(inputs, "train", **yml).item() < first
    padding_is_zero()
