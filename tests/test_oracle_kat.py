"""Known-answer tests that pin the CPU oracle (oracle/) -- the reference ships no tests or golden
vectors for this path, so every TF-1.13 semantic the oracle restates gets an
analytic case here, plus a cross-check against the independent numpy loop restatement."""
import math

import numpy as np
import pytest
import torch

from oracle import losses, naive, solver, tf_ops, unet2d


def test_same_padding_rule():
    # B2: out=ceil(in/s); extra pixel goes at the END
    assert tf_ops.same_pad(8, 3, 1) == (8, 1, 1)
    assert tf_ops.same_pad(8, 3, 2) == (4, 0, 1)
    assert tf_ops.same_pad(7, 3, 2) == (4, 1, 1)
    assert tf_ops.same_pad(8, 2, 2) == (4, 0, 0)


def test_conv_same_matches_loop_restatement():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 6, 3))
    w = rng.standard_normal((3, 3, 3, 4))
    got = tf_ops.conv_nd_same(torch.tensor(x), torch.tensor(w)).numpy()
    np.testing.assert_allclose(got, naive.conv2d_same(x, w), atol=1e-12)


def test_conv_same_stride2_even_is_asymmetric():
    # UNet3D e1/bridge case: k=3, s=2, even size -> pad 0 before, 1 after
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    w = np.ones((3, 3, 1, 1))
    got = tf_ops.conv_nd_same(torch.tensor(x), torch.tensor(w), stride=(2, 2)).numpy()
    np.testing.assert_allclose(got, naive.conv2d_same(x, w, (2, 2)), atol=1e-12)
    # top-left output sees rows/cols 0..2 (no leading pad)
    assert got[0, 0, 0, 0] == x[0, 0:3, 0:3, 0].sum()


def test_conv_identity_kernel_kat():
    x = np.random.default_rng(1).standard_normal((1, 4, 4, 2))
    w = np.zeros((3, 3, 2, 2))
    w[1, 1] = np.eye(2)
    np.testing.assert_allclose(tf_ops.conv_nd_same(torch.tensor(x), torch.tensor(w)).numpy(), x, atol=1e-14)


def test_deconv_k2s2_layout_and_bias_relu():
    # B5: filter [kh,kw,Cout,Cin]; out[2y+a,2x+b,co] = sum_ci x[y,x,ci] w[a,b,co,ci]
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 2, 3, 4))
    w = rng.standard_normal((2, 2, 5, 4))
    b = rng.standard_normal(5)
    got = tf_ops.conv_transpose_ks(torch.tensor(x), torch.tensor(w), (2, 2), bias=torch.tensor(b)).numpy()
    np.testing.assert_allclose(got, naive.conv2d_transpose_k2s2(x, w, b), atol=1e-12)
    y, xx, a, bb, co = 1, 2, 1, 0, 3
    assert math.isclose(got[0, 2 * y + a, 2 * xx + bb, co], x[0, y, xx] @ w[a, bb, co] + b[co], rel_tol=1e-12)


def test_batch_norm_train_eval_kat():
    # B3: biased variance normalises, unbiased goes to moving_variance, decay .999, eps 1e-3
    x = torch.tensor([[[[1.0], [3.0]]]], dtype=torch.float64)          # M = 2, mean 2, biased var 1
    g, b = torch.tensor([2.0], dtype=torch.float64), torch.tensor([0.5], dtype=torch.float64)
    mm, mv = torch.zeros(1, dtype=torch.float64), torch.ones(1, dtype=torch.float64)
    y, nmm, nmv = tf_ops.batch_norm(x, g, b, mm, mv, True)
    exp = (np.array([1.0, 3.0]) - 2.0) / math.sqrt(1.0 + 1e-3) * 2.0 + 0.5
    np.testing.assert_allclose(y.numpy().ravel(), exp, atol=1e-12)
    assert math.isclose(nmm.item(), 0.0 * 0.999 + 2.0 * 0.001, rel_tol=1e-12)
    assert math.isclose(nmv.item(), 1.0 * 0.999 + 2.0 * 0.001, rel_tol=1e-12)   # unbiased var = 2
    ye, _, _ = tf_ops.batch_norm(x, g, b, mm, mv, False)
    np.testing.assert_allclose(ye.numpy().ravel(), np.array([1.0, 3.0]) / math.sqrt(1.0 + 1e-3) * 2 + 0.5, atol=1e-12)
    yn, mean, var = naive.batch_norm_train(x.numpy(), 2.0, 0.5)
    np.testing.assert_allclose(y.numpy(), yn, atol=1e-12)


def test_instance_norm_eps():
    x = torch.tensor([[[[1.0], [3.0]]], [[[10.0], [10.0]]]], dtype=torch.float64)
    y = tf_ops.instance_norm(x, None, None)
    np.testing.assert_allclose(y[0].numpy().ravel(), np.array([-1.0, 1.0]) / math.sqrt(1 + 1e-6), atol=1e-12)
    np.testing.assert_allclose(y[1].numpy().ravel(), [0.0, 0.0], atol=1e-12)


def test_maxpool_valid():
    x = np.arange(2 * 4 * 4 * 1, dtype=np.float64).reshape(2, 4, 4, 1)
    np.testing.assert_allclose(tf_ops.max_pool2x2(torch.tensor(x)).numpy(), naive.max_pool2x2(x))


def test_compute_weights_and_sum_by_nonzero():
    labels = torch.tensor([[[0, 1], [1, 2]], [[0, 0], [0, 2]]])
    # numerical: w = nw[label], then per-sample renormalised to mean 1
    w = losses.compute_weights("numerical", labels, 3, numeric_w=[0.2, 0.4, 4.4])
    raw0 = np.array([0.2, 0.4, 0.4, 4.4])
    np.testing.assert_allclose(w[0].numpy().ravel(), raw0 / raw0.sum() * 4, rtol=1e-6)
    assert abs(w[1].mean().item() - 1.0) < 1e-6
    # proportion with decay: freq^-1 normalised
    wp = losses.compute_weights("proportion", labels, 3, proportion_decay=1.0)
    cnt = np.array([1.0, 2.0, 1.0]) + 1.0
    pw = (1 / cnt) / (1 / cnt).sum()
    raw = pw[[0, 1, 1, 2]]
    np.testing.assert_allclose(wp[0].numpy().ravel(), raw / raw.sum() * 4, rtol=1e-6)
    # `none` returns the scalar 1.0 and the loss divides by ALL elements
    logits = torch.randn(2, 2, 2, 3, generator=torch.Generator().manual_seed(0))
    l_none = losses.weighted_sparse_softmax_cross_entropy(logits, labels, "none").item()
    assert math.isclose(l_none, naive.weighted_xent(logits.numpy(), labels.numpy(), 1.0), rel_tol=1e-6)
    l_num = losses.weighted_sparse_softmax_cross_entropy(logits, labels, "numerical", numeric_w=[0.0, 1.0, 1.0]).item()
    wz = losses.compute_weights("numerical", labels, 3, numeric_w=[0.0, 1.0, 1.0]).numpy()
    assert math.isclose(l_num, naive.weighted_xent(logits.numpy(), labels.numpy(), wz), rel_tol=1e-6)
    assert np.count_nonzero(wz) == 4     # zero weights do not count in the denominator


def test_dice_loss_and_metrics_closed_form():
    # perfect prediction -> dice loss 0 (up to eps), metric 1
    labels = torch.tensor([[[1, 1], [0, 0]]])
    probs = torch.nn.functional.one_hot(labels, 2).to(torch.float32)
    assert abs(losses.sparse_dice_loss(probs, labels).item()) < 1e-6
    pred = (probs[..., 1:] > 0.5).to(torch.uint8)
    lab = (labels == 1).unsqueeze(-1)
    assert abs(losses.metric_dice(pred, lab).item() - 1.0) < 1e-6
    # half overlap: pred {a,b}, label {b,c}: k=1, |p|=2, |l|=2
    pred = torch.tensor([[[[1], [1]], [[0], [0]]]], dtype=torch.uint8)
    lab = torch.tensor([[[[0], [1]], [[1], [0]]]], dtype=torch.uint8)
    assert math.isclose(losses.metric_dice(pred, lab).item(), (2 * 1 + 1e-5) / (2 + 2 + 1e-5), rel_tol=1e-6)
    assert math.isclose(losses.metric_voe(pred, lab).item(), 100 * (1 - 1 / (3 + 1e-5)), rel_tol=1e-6)
    assert math.isclose(losses.metric_vd(pred, lab).item(), 0.0, abs_tol=1e-6)
    # empty prediction and label -> dice = eps/eps = 1
    z = torch.zeros(1, 2, 2, 1, dtype=torch.uint8)
    assert abs(losses.metric_dice(z, z).item() - 1.0) < 1e-6


def test_threshold_is_strict_and_argmax_lowest_index():
    probs = torch.tensor([[[[0.5, 0.5]]]])
    assert losses.threshold_pred(probs)[0].item() == 0           # prob > 0.5 is strict
    assert int(np.argmax(np.array([0.4, 0.4, 0.2]))) == 0       # evaluator_liver.py:663 tie rule


def test_tf_adam_single_step():
    # B12: t=1: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2; eps OUTSIDE the correction
    opt = solver.TFAdam(0.9, 0.99, 1e-8)
    p = {"w": np.array([1.0])}
    g = {"w": np.array([0.5])}
    opt.step(p, g, 1e-3)
    lr_t = 1e-3 * math.sqrt(1 - 0.99) / (1 - 0.9)
    exp = 1.0 - lr_t * (0.1 * 0.5) / (math.sqrt(0.01 * 0.25) + 1e-8)
    assert math.isclose(p["w"][0], exp, rel_tol=1e-12)


def test_lr_policies():
    assert solver.learning_rate("period_step", 250000, 1e-3, 100000, 0.1) == pytest.approx(1e-5)
    assert solver.learning_rate("custom_step", 10, boundaries=[10, 20], values=[1.0, 0.5, 0.1]) == 1.0
    assert solver.learning_rate("custom_step", 11, boundaries=[10, 20], values=[1.0, 0.5, 0.1]) == 0.5
    assert solver.learning_rate("custom_step", 21, boundaries=[10, 20], values=[1.0, 0.5, 0.1]) == 0.1
    assert solver.learning_rate("poly", 500, 1e-3, total_steps=1000, end_lr=1e-6, power=0.9) == \
        pytest.approx((1e-3 - 1e-6) * 0.5 ** 0.9 + 1e-6)
    assert solver.learning_rate("poly", 5000, 1e-3, total_steps=1000, end_lr=1e-6) == pytest.approx(1e-6)
    assert solver.learning_rate("period_step", 5, 1e-3, slow_start_step=10, slow_start_lr=1e-4) == 1e-4
    assert solver.plateau_update(1e-3, 0.2, 0) == pytest.approx(2e-4)


def test_unet_param_count_matches_survey():
    # 31 037 763 trainable parameters for 3 classes, 31 037 698 for 2
    for ncls, want in ((3, 31037763), (2, 31037698)):
        specs = unet2d.param_specs(3, ncls)
        n = sum(int(np.prod(s)) for _, s, k in specs if k in unet2d.TRAINABLE_KINDS)
        assert n == want


def test_unet_oracle_concat_order_and_shapes():
    net = unet2d.UNet2DOracle(3, 3, init_channels=4, num_down_samples=2)
    p = unet2d.init_params(net.specs, seed=3)
    x = torch.rand(1, 8, 8, 3, generator=torch.Generator().manual_seed(0))
    taps = {}
    logits, stats = net.forward(p, x, True, taps)
    assert logits.shape == (1, 8, 8, 3)
    # Decode1 conv1 weight has 2*C input channels: skip first, then up (UNet.py:93)
    assert p["UNet/Decode1/Repeat/convolution2d_1/weights"].shape == (3, 3, 8, 4)
    assert len(stats) == 2 * 10


def test_unet_regularisation_literal_bias_rule():
    net = unet2d.UNet2DOracle(3, 2, init_channels=4, num_down_samples=1)
    p = unet2d.init_params(net.specs, seed=1)
    for k in p:
        if k.endswith("biases"):
            p[k] = torch.ones_like(p[k])
    with_b = net.regularization_loss(p, 1e-2, bias_decay=False).item()
    without_b = net.regularization_loss(p, 1e-2, bias_decay=True).item()
    nb = sum(v.numel() for k, v in p.items() if k.endswith("biases"))
    assert math.isclose(with_b - without_b, 1e-2 * 0.5 * nb, rel_tol=1e-5)


def test_boundary_weights_against_brute_force():
    """loss_metrics.py:149-165: ring = pixels with a differently-labelled 3x3 neighbour; w = exp(-EDT/25) + 1,
    normalised to mean 1 per sample.  Brute-force distances on a hand-made map."""
    import numpy as np
    import torch
    from oracle import losses
    lab = np.zeros((1, 7, 9), dtype=np.int64)
    lab[0, 2:5, 3:6] = 1
    lab[0, 3, 4] = 2
    w = losses.compute_weights("boundary", torch.from_numpy(lab), 3).numpy()[0]
    ring = np.zeros((7, 9), bool)
    for h in range(7):
        for x in range(9):
            for dh in (-1, 0, 1):
                for dw in (-1, 0, 1):
                    hh, ww = h + dh, x + dw
                    if 0 <= hh < 7 and 0 <= ww < 9 and lab[0, hh, ww] != lab[0, h, x]:
                        ring[h, x] = True
    ys, xs = np.nonzero(ring)
    ref = np.zeros((7, 9))
    for h in range(7):
        for x in range(9):
            ref[h, x] = np.exp(-np.sqrt(((ys - h) ** 2 + (xs - x) ** 2).min()) / 25) + 1
    ref = ref / ref.sum() * 63
    np.testing.assert_allclose(w, ref, rtol=1e-6)
    assert abs(w.mean() - 1.0) < 1e-6
