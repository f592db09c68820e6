"""TF-1.13 / tf.contrib.slim op semantics restated on PyTorch-CPU (NHWC API).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Every function takes/returns NHWC (or NDHWC) tensors and TF filter layouts so
that callers read like the reference's slim calls.  torch is used as a CPU
array library with autograd; nothing here touches a GPU.

Semantics restated:
  B1  slim.conv2d: stride 1, SAME, bias dropped when a normalizer is set
      (reference call sites NetworksV2/UNet.py:79,85,94,100)
  B2  TF SAME padding: out=ceil(in/s); total=max((out-1)*s+k-in,0);
      before=total//2, after=total-before
  B3  slim.batch_norm: decay .999, eps 1e-3, biased var to normalise,
      unbiased var into moving_variance (NetworksV2/base.py:153-162)
  B4  slim.instance_norm: eps 1e-6, biased var over spatial axes (base.py:163-165)
  B5  slim.conv2d_transpose: filter [kh,kw,Cout,Cin], SAME, out=in*s for k=s
      (NetworksV2/UNet.py:91-92)
  B6  slim.max_pool2d: 2x2 stride 2 VALID (UNet.py:81)
"""
import math

import torch
import torch.nn.functional as F


def same_pad(in_size, k, s):
    """TF SAME padding (B2): returns (out_size, pad_before, pad_after)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv_nd_same(x, w, stride=None, bias=None, dilation=1):
    """slim.conv2d / slim.conv3d with padding='SAME' (and `rate=dilation` on the spatial axes: tf.nn.convolution pads
    SAME for the EFFECTIVE kernel (k - 1) * rate + 1, e.g. 2 on each side for a 3x3 rate-2 conv at stride 1).

    x: [N, (D,) H, W, Cin]; w: TF filter [(kd,) kh, kw, Cin, Cout].
    """
    nsp = x.dim() - 2
    if stride is None:
        stride = (1,) * nsp
    ks = tuple(w.shape[:nsp])
    # NHWC -> NCHW
    perm_in = (0, nsp + 1) + tuple(range(1, nsp + 1))
    xc = x.permute(*perm_in)
    # [k.., Cin, Cout] -> [Cout, Cin, k..]
    wc = w.permute(nsp + 1, nsp, *range(nsp))
    pads = []
    for d in reversed(range(nsp)):   # F.pad wants last dim first
        _, pb, pa = same_pad(x.shape[1 + d], (ks[d] - 1) * dilation + 1, stride[d])
        pads += [pb, pa]
    xc = F.pad(xc, pads)
    conv = F.conv2d if nsp == 2 else F.conv3d
    y = conv(xc, wc, bias=bias, stride=stride, dilation=dilation)
    perm_out = (0,) + tuple(range(2, nsp + 2)) + (1,)
    return y.permute(*perm_out).contiguous()


def bf16_round(t):
    """Round-to-nearest-even to bfloat16, kept in t's dtype (bf16 values are exact in fp32 / fp64)."""
    return t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _ConvSameBf16Operands(torch.autograd.Function):
    """Restatement of the UNETK_BF16 arithmetic (include/unetk.h; not a reference mode -- BASELINE.json configs[2]):
    every contraction rounds BOTH operands to bf16 and accumulates exactly; forward y = conv(r(x), r(w)), backward
    dx = conv^T(r(dy), r(w)) and dw = corr(r(x), r(dy)) -- i.e. the gradient is NOT the derivative of the rounded
    forward but what mixed-precision kernels compute."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return conv_nd_same(bf16_round(x), bf16_round(w))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xr = bf16_round(x).requires_grad_(True)
            wr = bf16_round(w).requires_grad_(True)
            y = conv_nd_same(xr, wr)
            dx, dw = torch.autograd.grad(y, (xr, wr), bf16_round(dy))
        return dx, dw


def conv_same_bf16_operands(x, w):
    return _ConvSameBf16Operands.apply(x, w)


class _StoreBf16(torch.autograd.Function):
    """UNETK_BF16S ("bf16 storage", include/unetk.h; BASELINE.json configs[2]): a tensor that lives in HBM as bf16.  The
    forward value is rounded when it is written; so is its gradient (the activation gradient is stored as bf16 too, and
    when a tensor has two consumers their gradients are summed in fp32 and rounded once -- autograd hands the sum here)."""

    @staticmethod
    def forward(ctx, x):
        return bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return bf16_round(g)


def store_bf16(x):
    return _StoreBf16.apply(x)


class _NormReluBf16S(torch.autograd.Function):
    """Normalisation + ReLU of a conv unit under UNETK_BF16S, restating csrc/norm.hip with bf16 tensors:
    statistics come from the conv's fp32 accumulators (the UNROUNDED y), the raw output is stored rounded (y_r), the
    affine + ReLU reads y_r and its result is stored rounded; the backward recomputes u from y_r, uses xhat of y_r in
    the batch-norm gradient formula (rounding is passed straight through) and stores dy rounded.
    kind: "batch_norm" (statistics over all but the channel axis), "instance_norm" (per sample) or "none" (y + bias)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, kind, eps, mean_in, var_in):
        axes = tuple(range(y.dim() - 1)) if kind == "batch_norm" else tuple(range(1, y.dim() - 1))
        keep = kind == "instance_norm"
        yr = bf16_round(y)
        if kind == "none":
            scale = torch.ones_like(beta)
            mean = torch.zeros_like(beta)
            rstd = torch.ones_like(beta)
        else:
            if mean_in is None:
                mean = y.mean(dim=axes, keepdim=keep)
                var = y.var(dim=axes, unbiased=False, keepdim=keep)
            else:
                mean, var = mean_in, var_in
            rstd = torch.rsqrt(var + eps)
            scale = rstd * gamma if gamma is not None else rstd
        shift = (beta if beta is not None else 0.0) - mean * scale
        u = yr * scale + shift
        z = bf16_round(torch.relu(u))
        ctx.save_for_backward(yr, u, mean, rstd, scale)
        ctx.cfg = (kind, axes, keep, gamma is not None, beta is not None, mean_in is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        yr, u, mean, rstd, scale = ctx.saved_tensors
        kind, axes, keep, has_g, has_b, frozen = ctx.cfg
        du = bf16_round(dz) * (u > 0).to(dz.dtype)
        par_axes = tuple(range(yr.dim() - 1))
        dbeta = du.sum(dim=par_axes) if has_b else None
        if kind == "none":
            return bf16_round(du), None, dbeta, None, None, None, None
        xhat = (yr - mean) * rstd
        dgamma = (du * xhat).sum(dim=par_axes) if has_g else None
        if frozen:                       # eval-mode statistics: no dependence of the statistics on y
            dy = scale * du
        else:
            k1 = du.mean(dim=axes, keepdim=keep)
            k2 = (du * xhat).mean(dim=axes, keepdim=keep)
            dy = scale * (du - k1 - xhat * k2)
        return bf16_round(dy), dgamma, dbeta, None, None, None, None


def norm_relu_bf16s(y, gamma, beta, kind, eps=0.0, mean=None, var=None):
    return _NormReluBf16S.apply(y, gamma, beta, kind, eps, mean, var)


def conv_transpose_ks(x, w, stride, bias=None):
    """slim.conv2d_transpose / conv3d_transpose with kernel == stride, SAME (B5).

    x: [N, (D,) H, W, Cin]; w: TF filter [(kd,) kh, kw, Cout, Cin].
    out[.., s*y+a, s*x+b, co] = sum_ci x[.., y, x, ci] * w[a, b, co, ci] (+ bias)
    """
    nsp = x.dim() - 2
    assert tuple(w.shape[:nsp]) == tuple(stride), "oracle covers kernel==stride only"
    perm_in = (0, nsp + 1) + tuple(range(1, nsp + 1))
    xc = x.permute(*perm_in)
    # torch conv_transpose weight: [Cin, Cout, k..]
    wc = w.permute(nsp + 1, nsp, *range(nsp))
    convt = F.conv_transpose2d if nsp == 2 else F.conv_transpose3d
    y = convt(xc, wc, bias=bias, stride=stride)
    perm_out = (0,) + tuple(range(2, nsp + 2)) + (1,)
    return y.permute(*perm_out).contiguous()


class _ConvTransposeBf16Operands(torch.autograd.Function):
    """UNETK_BF16 arithmetic of the k = s transposed conv (bias is added outside, in full precision)."""

    @staticmethod
    def forward(ctx, x, w, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return conv_transpose_ks(bf16_round(x), bf16_round(w), stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xr = bf16_round(x).requires_grad_(True)
            wr = bf16_round(w).requires_grad_(True)
            y = conv_transpose_ks(xr, wr, ctx.stride)
            dx, dw = torch.autograd.grad(y, (xr, wr), bf16_round(dy))
        return dx, dw, None


def conv_transpose_bf16_operands(x, w, stride, bias=None):
    y = _ConvTransposeBf16Operands.apply(x, w, tuple(stride))
    return y if bias is None else y + bias


def max_pool2x2(x):
    """slim.max_pool2d(x, [2, 2]) : stride 2, VALID (B6)."""
    xc = x.permute(0, 3, 1, 2)
    y = F.max_pool2d(xc, kernel_size=2, stride=2)
    return y.permute(0, 2, 3, 1).contiguous()


def avg_pool2x2_same(x):
    """slim.avg_pool2d(x, 2) with SAME (GUNet.py:158); even sizes only here."""
    xc = x.permute(0, 3, 1, 2)
    y = F.avg_pool2d(xc, kernel_size=2, stride=2, ceil_mode=True, count_include_pad=False)
    return y.permute(0, 2, 3, 1).contiguous()


def batch_norm(x, gamma, beta, moving_mean, moving_var, is_training,
               eps=1e-3, decay=0.999):
    """slim.batch_norm(scale=True, fused) (B3).

    Returns (y, new_moving_mean, new_moving_var).  Training normalises with the
    batch mean and the *biased* batch variance; the moving variance receives the
    *unbiased* estimate (TF fused batch norm); moving <- moving*decay + batch*(1-decay).
    """
    axes = tuple(range(x.dim() - 1))
    if is_training:
        mean = x.mean(dim=axes)
        var = x.var(dim=axes, unbiased=False)
        m = x.numel() // x.shape[-1]
        var_unbiased = var * (m / max(m - 1, 1))
        new_mm = moving_mean * decay + mean.detach() * (1.0 - decay)
        new_mv = moving_var * decay + var_unbiased.detach() * (1.0 - decay)
    else:
        mean, var = moving_mean, moving_var
        new_mm, new_mv = moving_mean, moving_var
    y = (x - mean) * torch.rsqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y, new_mm, new_mv


def instance_norm(x, gamma, beta, eps=1e-6):
    """slim.instance_norm (B4): moments over spatial axes per (n, c), biased var."""
    axes = tuple(range(1, x.dim() - 1))
    mean = x.mean(dim=axes, keepdim=True)
    var = x.var(dim=axes, unbiased=False, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y


def xavier_uniform_(shape, fan_in, fan_out, gen, dtype=torch.float32):
    """slim.xavier_initializer() = Glorot uniform (B8, base.py:141)."""
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(limit).to(dtype)


def image_gradients(x):
    """tf.image.image_gradients (B16, UNet.py:70): forward difference, last row/col 0."""
    dy = torch.zeros_like(x)
    dx = torch.zeros_like(x)
    dy[:, :-1] = x[:, 1:] - x[:, :-1]
    dx[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1]
    return dy, dx
