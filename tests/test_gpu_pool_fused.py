"""GPU: the max-pool backward folded into the norm backward of the unit that feeds it (ops.Conv3x3NormReluPool,
unetk_norm_relu_bwd_pool) -- slim.conv2d + slim.max_pool2d + the skip connection of NetworksV2/UNet.py:79-81,93.
The fused route must give what the separate passes give (MaxPoolGrad's first-maximum rule included: ReLU produces whole
windows of equal zeros, and bf16 storage produces equal maxima all the time): the input gradient is compared BITWISE
where the statistics sums allow it (the per-pixel routed gradient is identical; only the two channel sums of the norm
backward are accumulated in another order), and the whole thing against a float64 autograd evaluation."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(kind, dtype, fused, n, h, w, cin, cout, seed=3):
    from boxsegliver_amd import _abi, ops
    old = ops.POOL_FUSED
    ops.POOL_FUSED = fused
    try:
        g = torch.Generator(device="cuda").manual_seed(seed)
        bf = dtype == "bf16"
        x = torch.randn((n, h, w, cin), device="cuda", generator=g)
        if bf:
            x = x.bfloat16()
        x.requires_grad_(True)
        wt = (torch.randn((3, 3, cin, cout), device="cuda", generator=g) / math.sqrt(9 * cin)).requires_grad_(True)
        gamma = (1.0 + 0.1 * torch.randn((cout,), device="cuda", generator=g)).requires_grad_(True)
        beta = (0.1 * torch.randn((cout,), device="cuda", generator=g)).requires_grad_(True)
        mm, mv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        spec = ops.NormSpec(kind, 1e-5, 0.9, True, _abi.BF16S if bf else 0)
        cat = torch.empty((n, h, w, 2 * cout), device="cuda", dtype=x.dtype)
        out = ops.alias(cat, 0, (n, h, w, cout), cat.stride())
        p, z = ops.Conv3x3NormReluPool.apply(x, wt, gamma, beta, mm if kind == "batch_norm" else None,
                                             mv if kind == "batch_norm" else None, spec, out)
        dp = torch.randn(p.shape, device="cuda", generator=g).to(p.dtype)
        dcat = torch.randn(cat.shape, device="cuda", generator=g).to(p.dtype)       # the skip's gradient = a channel slice
        dskip = dcat[..., :cout]
        torch.autograd.backward([p, z], [dp, dskip])
        torch.cuda.synchronize()
        return dict(x=x.detach(), w=wt.detach(), gamma=gamma.detach(), beta=beta.detach(), p=p.detach(), z=z.detach().clone(),
                    dp=dp, dskip=dskip.clone(), dx=x.grad.clone(), dw=wt.grad.clone(), dgamma=gamma.grad.clone(),
                    dbeta=beta.grad.clone())
    finally:
        ops.POOL_FUSED = old


CASES = [("batch_norm", "fp32", 2, 32, 48, 64, 64), ("instance_norm", "fp32", 3, 16, 32, 64, 128),
         ("batch_norm", "bf16", 4, 64, 64, 64, 64), ("instance_norm", "bf16", 2, 32, 32, 64, 128),
         ("batch_norm", "bf16", 8, 128, 128, 64, 128)]


@pytest.mark.parametrize("kind,dtype,n,h,w,cin,cout", CASES)
def test_fused_pool_backward_equals_the_separate_passes(kind, dtype, n, h, w, cin, cout):
    a = _run(kind, dtype, True, n, h, w, cin, cout)
    b = _run(kind, dtype, False, n, h, w, cin, cout)
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["z"], b["z"])

    def rel(u, v):
        return float((u.double() - v.double()).norm() / v.double().norm())
    # the routed gradient is bitwise the same; the two channel sums are accumulated window by window instead of row by row
    assert rel(a["dgamma"], b["dgamma"]) < 2e-6 and rel(a["dbeta"], b["dbeta"]) < 2e-6
    tol = 2e-6 if dtype == "fp32" else 2e-3         # bf16: a last-bit change of a sum flips roundings of stored dy values
    assert rel(a["dx"], b["dx"]) < tol, rel(a["dx"], b["dx"])
    assert rel(a["dw"], b["dw"]) < tol, rel(a["dw"], b["dw"])


@pytest.mark.parametrize("kind", ["batch_norm", "instance_norm"])
def test_fused_pool_backward_against_float64_autograd(kind):
    import torch.nn.functional as F
    n, h, w, cin, cout = 2, 16, 32, 64, 64
    a = _run(kind, "fp32", True, n, h, w, cin, cout)
    x = a["x"].double().permute(0, 3, 1, 2).requires_grad_(True)
    wt = a["w"].double().permute(3, 2, 0, 1).requires_grad_(True)
    gamma, beta = a["gamma"].double().requires_grad_(True), a["beta"].double().requires_grad_(True)
    y = F.conv2d(x, wt, padding=1)
    dims = (0, 2, 3) if kind == "batch_norm" else (2, 3)
    mu, var = y.mean(dims, keepdim=True), y.var(dims, unbiased=False, keepdim=True)
    z = torch.relu((y - mu) / torch.sqrt(var + 1e-5) * gamma[None, :, None, None] + beta[None, :, None, None])
    # the device's own window choice (first maximum of the fp32 z): route dp with it, so that near-ties do not decide the test
    zd = a["z"].double().permute(0, 3, 1, 2)
    win = zd.reshape(n, cout, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, cout, h // 2, w // 2, 4)
    first = torch.zeros_like(win)
    first.scatter_(-1, win.argmax(-1, keepdim=True), 1.0)          # argmax returns the first maximal index
    route = first.reshape(n, cout, h // 2, w // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, cout, h, w)
    dp_full = a["dp"].double().permute(0, 3, 1, 2).repeat_interleave(2, 2).repeat_interleave(2, 3)
    dz = a["dskip"].double().permute(0, 3, 1, 2) + route * dp_full
    z.backward(dz)

    def rel(u, v):
        return float((u.double() - v).norm() / v.norm())
    assert rel(a["dx"].permute(0, 3, 1, 2), x.grad) < 2e-5
    assert rel(a["dw"].permute(3, 2, 0, 1), wt.grad) < 2e-5
    assert rel(a["dgamma"], gamma.grad) < 2e-5 and rel(a["dbeta"], beta.grad) < 2e-5


def test_first_maximum_takes_the_pooled_gradient_in_windows_of_equal_values():
    """All-zero windows (ReLU) and exact ties: dz = dskip + dp at the FIRST maximum in scan order, as unetk_maxpool2_bwd routes."""
    from boxsegliver_amd import ops
    n, h, w, c = 1, 4, 4, 64
    # gamma = 0, beta = -1 => z == 0 everywhere: every window is a four-way tie and every ReLU mask is off => dy == 0;
    # beta = +1 => z == 1 everywhere: four-way ties with the mask ON: dz has dp at (0, 0) of each window
    for b0 in (-1.0, 1.0):
        x = torch.randn((n, h, w, 64), device="cuda").requires_grad_(True)
        wt = (torch.randn((3, 3, 64, c), device="cuda") * 0.05).requires_grad_(True)
        gamma = torch.zeros(c, device="cuda", requires_grad=True)
        beta = torch.full((c,), b0, device="cuda", requires_grad=True)
        spec = ops.NormSpec("instance_norm", 1e-6, 0.0, True, 0)
        cat = torch.empty((n, h, w, 2 * c), device="cuda")
        out = ops.alias(cat, 0, (n, h, w, c), cat.stride())
        p, z = ops.Conv3x3NormReluPool.apply(x, wt, gamma, beta, None, None, spec, out)
        dp = torch.ones_like(p)
        dskip = torch.zeros((n, h, w, c), device="cuda")
        torch.autograd.backward([p, z], [dp, dskip])
        # d beta = sum of du = (mask on) * number of windows (dp == 1 lands on exactly one pixel per window)
        want = 0.0 if b0 < 0 else float((h // 2) * (w // 2))
        assert torch.allclose(beta.grad, torch.full((c,), want, device="cuda")), (b0, beta.grad[:4])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fused_apply_and_pool_forward_equals_the_separate_passes(dtype):
    from boxsegliver_amd import ops
    a = _run("batch_norm", dtype, True, 4, 64, 32, 64, 64)
    assert torch.equal(a["p"], ops.maxpool2_fwd(a["z"].contiguous()))
    b = _run("batch_norm", dtype, False, 4, 64, 32, 64, 64)           # POOL_FUSED off: norm_apply_relu + maxpool2_fwd
    assert torch.equal(a["z"], b["z"]) and torch.equal(a["p"], b["p"])
