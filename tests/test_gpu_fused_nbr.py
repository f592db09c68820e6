"""conv2's input gradient fused with conv1's norm-backward reduction (unetk_conv3x3_dgrad_nbr +
unetk_norm_relu_bwd_pre): the slim.repeat(x, 2, slim.conv2d, C, 3) pair of NetworksV2/UNet.py:79,85,94.
Checked against the two separate calls on the same operands (identical arithmetic per element, only the order of the
partial sums differs) and, for fp32, against float64 autograd of the oracle ops."""
import pytest
import torch

from oracle import tf_ops

pytestmark = pytest.mark.gpu


def _pair(kind, prec, n, h, w, c0, c1, c2, fuse, seed=3):
    from boxsegliver_amd import ops
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, c0, generator=gen)
    w1 = torch.randn(3, 3, c0, c1, generator=gen) / (3 * c0 ** 0.5)
    w2 = torch.randn(3, 3, c1, c2, generator=gen) / (3 * c1 ** 0.5)
    g1, g2 = 0.5 + torch.rand(c1, generator=gen), 0.5 + torch.rand(c2, generator=gen)
    g1[::7] *= -1.0                                          # negative scales: the ReLU mask must use scale, not |scale|
    b1, b2 = 0.2 * torch.randn(c1, generator=gen), 0.2 * torch.randn(c2, generator=gen)
    dz = torch.randn(n, h, w, c2, generator=gen)
    st = ops.storage_dtype(prec)
    xd = x.cuda().to(st).requires_grad_(True)
    ps = [t.cuda().requires_grad_(True) for t in (w1, g1, b1, w2, g2, b2)]
    spec = ops.NormSpec(kind, 1e-3 if kind == "batch_norm" else 1e-6, 0.99, True, bf16=prec)
    mm = [torch.zeros(c).cuda() for c in (c1, c2)]
    mv = [torch.ones(c).cuda() for c in (c1, c2)]
    ops.FUSE_NBR = fuse
    ops.new_step()
    try:
        z1 = ops.Conv3x3NormRelu.apply(xd, ps[0], ps[1], ps[2], mm[0], mv[0], spec, None, None, None, None)
        z2 = ops.Conv3x3NormRelu.apply(z1, ps[3], ps[4], ps[5], mm[1], mv[1], spec, None, None, None, None)
        z2.backward(dz.cuda().to(st))
        fused = len(ops.FUSED_NBR)                            # entries are popped by the producer's backward
    finally:
        ops.FUSE_NBR = True
    grads = [xd.grad.float().cpu()] + [p.grad.cpu() for p in ps]
    return grads, fused, (x, w1, g1, b1, w2, g2, b2, dz)


@pytest.mark.parametrize("kind", ["batch_norm", "instance_norm"])
@pytest.mark.parametrize("prec,shape", [(0, (2, 24, 40, 64, 128, 128)), (0, (3, 16, 32, 64, 256, 128)),
                                        (2, (2, 32, 48, 64, 128, 128)), (2, (2, 24, 16, 128, 128, 256)),
                                        (0, (16, 64, 64, 64, 256, 256))])    # 16 x 16 pixel tiles (>= 512 blocks, K = 256)
def test_fused_reduction_equals_separate_passes(kind, prec, shape):
    from boxsegliver_amd import ops
    n, h, w, c0, c1, c2 = shape
    g_f, left_f, ops_in = _pair(kind, prec, n, h, w, c0, c1, c2, True)
    g_s, left_s, _ = _pair(kind, prec, n, h, w, c0, c1, c2, False)
    assert left_f == 0 and left_s == 0                       # produced and consumed
    # did the fused path run?  its partial rows exist only then
    d = ops.ConvDesc(n, h, w, c1, c2, c1, c2, prec, 1)
    assert ops._abi.lib().unetk_conv3x3_dgrad_nbr_rows(__import__("ctypes").byref(d)) > 0
    names = ("dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2")
    for name, a, b in zip(names, g_f, g_s):
        err = float((a.double() - b.double()).norm() / b.double().norm())
        # same per-element arithmetic; the sums are grouped per conv tile instead of per reduction block.  bf16 storage
        # rounds dy once more downstream, so a last-bit change of a sum can move isolated dy values by one bf16 ulp
        assert err < (2e-6 if prec == 0 else 3e-3), (name, err)
    if prec == 0 and n * h * w <= 8192:                       # the float64 CPU oracle of the small shapes only
        x, w1, g1, b1, w2, g2, b2, dz = ops_in
        v = [t.double().requires_grad_(True) for t in (x, w1, g1, b1, w2, g2, b2)]

        def unit(t, wt, g, b):
            y = tf_ops.conv_nd_same(t, wt)
            if kind == "batch_norm":
                c = wt.shape[-1]
                u, _, _ = tf_ops.batch_norm(y, g, b, torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64), True)
            else:
                u = tf_ops.instance_norm(y, g, b, eps=1e-6)
            return torch.relu(u)
        unit(unit(v[0], v[1], v[2], v[3]), v[4], v[5], v[6]).backward(dz.double())
        for name, a, t in zip(names, g_f, v):
            l2 = float((a.double() - t.grad).norm() / t.grad.norm())
            assert l2 < 1e-3, (name, l2)                      # a few ReLU mask flips within rounding of zero


def test_fusion_is_taken_inside_the_unet():
    """The whole-net parity tests run with the fusion on: make sure it is actually exercised there."""
    from boxsegliver_amd import ops
    seen = []
    real = ops.conv3x3_dgrad

    def spy(*a, **k):
        out = real(*a, **k)
        seen.append(out.data_ptr() in ops.FUSED_NBR)
        return out
    ops.conv3x3_dgrad = spy
    try:
        _pair("batch_norm", 0, 2, 16, 32, 64, 128, 128, True)
    finally:
        ops.conv3x3_dgrad = real
    assert seen == [True, False]                             # conv2 -> conv1 fused; conv1's own dx (x needs a gradient here) has no producer


@pytest.mark.parametrize("prec", [0, 2])
def test_second_consumer_invalidates_the_fused_partials(prec):
    """z1 feeds conv2 AND a second branch: autograd adds the branch's gradient into conv2's dx in place (same data_ptr and
    shape), so the partials conv2's input gradient emitted are stale -- the producer must fall back to its own reduction
    pass (the entry's recorded version counter no longer matches).  Parity with FUSE_NBR = False."""
    from boxsegliver_amd import ops

    def run(fuse):
        gen = torch.Generator().manual_seed(11)
        n, h, w, c0, c1, c2 = 2, 24, 32, 64, 128, 128
        x = torch.randn(n, h, w, c0, generator=gen)
        w1 = torch.randn(3, 3, c0, c1, generator=gen) / (3 * c0 ** 0.5)
        w2 = torch.randn(3, 3, c1, c2, generator=gen) / (3 * c1 ** 0.5)
        g1, g2 = 0.5 + torch.rand(c1, generator=gen), 0.5 + torch.rand(c2, generator=gen)
        b1, b2 = 0.2 * torch.randn(c1, generator=gen), 0.2 * torch.randn(c2, generator=gen)
        dz = torch.randn(n, h, w, c2, generator=gen)
        side = torch.randn(n, h, w, c1, generator=gen)
        st = ops.storage_dtype(prec)
        xd = x.cuda().to(st).requires_grad_(True)
        ps = [t.cuda().requires_grad_(True) for t in (w1, g1, b1, w2, g2, b2)]
        spec = ops.NormSpec("batch_norm", 1e-3, 0.99, True, bf16=prec)
        mm = [torch.zeros(c).cuda() for c in (c1, c2)]
        mv = [torch.ones(c).cuda() for c in (c1, c2)]
        ops.FUSE_NBR = fuse
        ops.new_step()
        try:
            z1 = ops.Conv3x3NormRelu.apply(xd, ps[0], ps[1], ps[2], mm[0], mv[0], spec, None, None, None, None)
            z2 = ops.Conv3x3NormRelu.apply(z1, ps[3], ps[4], ps[5], mm[1], mv[1], spec, None, None, None, None)
            # the second consumer of z1: a plain torch branch with a large gradient of its own
            loss = (z2.float() * dz.cuda()).sum() + (z1.float() * side.cuda()).sum() * 3.0
            loss.backward()
        finally:
            ops.FUSE_NBR = True
            ops.FUSED_NBR.clear()
        return [xd.grad.float().cpu()] + [p.grad.cpu() for p in ps]

    g_f, g_s = run(True), run(False)
    for name, a, b in zip(("dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2"), g_f, g_s):
        err = float((a.double() - b.double()).norm() / b.double().norm())
        assert err < (2e-6 if prec == 0 else 3e-3), (name, err)
