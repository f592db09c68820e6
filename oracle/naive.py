"""Independent pure-numpy loop restatement of the slim ops (small cases only).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/__init__.py).

Used to cross-check oracle/tf_ops.py (which leans on torch's conv routines) so the
oracle does not rest on a single implementation.  Written directly from the TF-1.13
op definitions listed in SURVEY.md appendix B, in float64.
"""
import numpy as np


def same_pad(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, stride=(1, 1)):
    """x [N,H,W,Cin], w [kh,kw,Cin,Cout] (HWIO)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    Ho, pt, pb = same_pad(H, kh, stride[0])
    Wo, pl, pr = same_pad(W, kw, stride[1])
    y = np.zeros((N, Ho, Wo, Cout))
    for n in range(N):
        for ho in range(Ho):
            for wo in range(Wo):
                for a in range(kh):
                    hi = ho * stride[0] + a - pt
                    if hi < 0 or hi >= H:
                        continue
                    for b in range(kw):
                        wi = wo * stride[1] + b - pl
                        if wi < 0 or wi >= W:
                            continue
                        y[n, ho, wo] += x[n, hi, wi] @ w[a, b]
    return y


def conv2d_transpose_k2s2(x, w, bias=None):
    """x [N,H,W,Cin], w [2,2,Cout,Cin]: out[n,2y+a,2x+b,co] = sum_ci x[n,y,x,ci] w[a,b,co,ci]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    N, H, W, Cin = x.shape
    Cout = w.shape[2]
    y = np.zeros((N, 2 * H, 2 * W, Cout))
    for a in range(2):
        for b in range(2):
            y[:, a::2, b::2, :] = np.einsum("nhwi,oi->nhwo", x, w[a, b])
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y


def max_pool2x2(x):
    x = np.asarray(x)
    N, H, W, C = x.shape
    return x[:, :H // 2 * 2, :W // 2 * 2].reshape(N, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))


def batch_norm_train(x, gamma, beta, eps=1e-3):
    x = np.asarray(x, np.float64)
    mean = x.mean(axis=(0, 1, 2))
    var = ((x - mean) ** 2).mean(axis=(0, 1, 2))
    return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var


def softmax(z):
    z = np.asarray(z, np.float64)
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def weighted_xent(logits, labels, w):
    """sum(ce*w)/count(w!=0) with w scalar or per-pixel (tf.losses SUM_BY_NONZERO_WEIGHTS)."""
    p = softmax(logits)
    lab = np.asarray(labels)
    ce = -np.log(np.take_along_axis(p, lab[..., None], axis=-1)[..., 0])
    w = np.broadcast_to(np.asarray(w, np.float64), ce.shape)
    cnt = np.count_nonzero(w)
    return float((ce * w).sum() / cnt) if cnt else 0.0
