// GUNet's 1-D VGG context models (context_model "vgg16B" / "vgg16C" / "vgg16D", ext_config/GUNet_DE_VGG16{B,D}.yml):
// slim.conv1d (kernel 3 or 1, stride 1, SAME, bias, ReLU -- slim's defaults: the model's arg_scope names slim.conv2d, not
// conv1d) and tf.layers.max_pooling1d(2, 2, padding="same") on the context vector seen as [bs, L, 1].
//
// Replaces NetworksV2/Backbone/slim_nets.py:60-144 (vgg16B / C / D) as called by GUNet.py:62-75.  bs <= a few dozen rows,
// L <= a few hundred, <= 128 channels: latency-sized work -- one thread per output element, fixed summation orders
// (bit-reproducible), no MFMA; the flattened result feeds the context MLP of fc.hip.
#include "common.h"

namespace {

// y[b][l][co] = act(b[co] + sum_{t, ci} x[b][l + t - pad][ci] * w[t][ci][co]),  w = TF [k, Cin, Cout]
__global__ __launch_bounds__(256) void conv1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int L, int Cin, int Cout, int k, int relu) {
  const int64_t total = (int64_t)B * L * Cout;
  const int pad = (k - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % Cout);
    const int64_t r = i / Cout;
    const int l = (int)(r % L), b = (int)(r / L);
    float acc = bias ? bias[co] : 0.f;
    for (int t = 0; t < k; ++t) {
      const int ls = l + t - pad;
      if (ls < 0 || ls >= L) continue;
      const float* xr = x + ((int64_t)b * L + ls) * Cin;
      const float* wr = w + (int64_t)t * Cin * Cout + co;
      for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xr[ci], wr[(int64_t)ci * Cout], acc);
    }
    y[i] = relu ? fmaxf(acc, 0.f) : acc;
  }
}

// dpre = dy gated by the ReLU
__global__ __launch_bounds__(256) void conv1d_bwd_pre_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                             float* __restrict__ dpre, int64_t total, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    dpre[i] = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
}

// dw[t][ci][co] = sum_{b, l} x[b][l + t - pad][ci] * dpre[b][l][co];  db[co] = sum dpre (by the t = pad, ci = 0 threads)
__global__ __launch_bounds__(256) void conv1d_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dpre,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int L,
                                                           int Cin, int Cout, int k) {
  const int64_t total = (int64_t)k * Cin * Cout;
  const int pad = (k - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % Cout);
    const int64_t r = i / Cout;
    const int ci = (int)(r % Cin), t = (int)(r / Cin);
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b)
      for (int l = 0; l < L; ++l) {
        const float d = dpre[((int64_t)b * L + l) * Cout + co];
        sb += d;
        const int ls = l + t - pad;
        if (ls >= 0 && ls < L) s = fmaf(x[((int64_t)b * L + ls) * Cin + ci], d, s);
      }
    dw[i] = s;
    if (db && t == 0 && ci == 0) db[co] = sb;
  }
}

// dx[b][l][ci] = sum_{t, co} dpre[b][l - t + pad][co] * w[t][ci][co]
__global__ __launch_bounds__(256) void conv1d_bwd_x_kernel(const float* __restrict__ dpre, const float* __restrict__ w,
                                                           float* __restrict__ dx, int B, int L, int Cin, int Cout, int k) {
  const int64_t total = (int64_t)B * L * Cin;
  const int pad = (k - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int l = (int)(r % L), b = (int)(r / L);
    float s = 0.f;
    for (int t = 0; t < k; ++t) {
      const int lo = l - t + pad;
      if (lo < 0 || lo >= L) continue;
      const float* dr = dpre + ((int64_t)b * L + lo) * Cout;
      const float* wr = w + ((int64_t)t * Cin + ci) * Cout;
      for (int co = 0; co < Cout; ++co) s = fmaf(dr[co], wr[co], s);
    }
    dx[i] = s;
  }
}

// max_pooling1d(2, 2, "same"): Lo = ceil(L / 2), the last window of an odd L holds one element
__global__ __launch_bounds__(256) void maxpool1d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int L,
                                                            int C) {
  const int Lo = (L + 1) / 2;
  const int64_t total = (int64_t)B * Lo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int lo = (int)(r % Lo), b = (int)(r / Lo);
    const float a = x[((int64_t)b * L + 2 * lo) * C + c];
    y[i] = (2 * lo + 1 < L) ? fmaxf(a, x[((int64_t)b * L + 2 * lo + 1) * C + c]) : a;
  }
}

// the first maximum of the window gets the gradient (TF MaxPoolGrad)
__global__ __launch_bounds__(256) void maxpool1d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int B, int L, int C) {
  const int Lo = (L + 1) / 2;
  const int64_t total = (int64_t)B * Lo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int lo = (int)(r % Lo), b = (int)(r / Lo);
    const int64_t i0 = ((int64_t)b * L + 2 * lo) * C + c;
    const float g = dy[i];
    if (2 * lo + 1 < L) {
      const bool first = x[i0] >= x[i0 + C];
      dx[i0] = first ? g : 0.f;
      dx[i0 + C] = first ? 0.f : g;
    } else {
      dx[i0] = g;
    }
  }
}

inline int ew_blocks(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int unetk_conv1d_fwd(const float* x, const float* w, const float* b, float* y, int B, int L, int Cin, int Cout,
                                int k, int relu, void* stream) {
  UNETK_REQUIRE(x && w && y && B > 0 && L > 0 && Cin > 0 && Cout > 0);
  if (k != 1 && k != 3) return UNETK_E_UNSUPPORTED;
  UNETK_LAUNCH(conv1d_fwd_kernel, dim3(ew_blocks((int64_t)B * L * Cout)), dim3(256), 0, (hipStream_t)stream, x, w, b,
                     y, B, L, Cin, Cout, k, relu);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_conv1d_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                                float* db, float* dpre_ws, int B, int L, int Cin, int Cout, int k, int relu, void* stream) {
  UNETK_REQUIRE(x && w && y && dy && dw && dpre_ws && B > 0 && L > 0 && Cin > 0 && Cout > 0);
  if (k != 1 && k != 3) return UNETK_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * L * Cout;
  UNETK_LAUNCH(conv1d_bwd_pre_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, y, dy, dpre_ws, n, relu);
  UNETK_LAUNCH_CHECK();
  UNETK_LAUNCH(conv1d_bwd_w_kernel, dim3(ew_blocks((int64_t)k * Cin * Cout)), dim3(256), 0, st, x, dpre_ws, dw, db, B,
                     L, Cin, Cout, k);
  UNETK_LAUNCH_CHECK();
  if (dx) {
    UNETK_LAUNCH(conv1d_bwd_x_kernel, dim3(ew_blocks((int64_t)B * L * Cin)), dim3(256), 0, st, dpre_ws, w, dx, B, L,
                       Cin, Cout, k);
    UNETK_LAUNCH_CHECK();
  }
  return UNETK_OK;
}

extern "C" int unetk_maxpool1d_fwd(const float* x, float* y, int B, int L, int C, void* stream) {
  UNETK_REQUIRE(x && y && B > 0 && L > 0 && C > 0);
  UNETK_LAUNCH(maxpool1d_fwd_kernel, dim3(ew_blocks((int64_t)B * ((L + 1) / 2) * C)), dim3(256), 0,
                     (hipStream_t)stream, x, y, B, L, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool1d_bwd(const float* x, const float* dy, float* dx, int B, int L, int C, void* stream) {
  UNETK_REQUIRE(x && dy && dx && B > 0 && L > 0 && C > 0);
  UNETK_LAUNCH(maxpool1d_bwd_kernel, dim3(ew_blocks((int64_t)B * ((L + 1) / 2) * C)), dim3(256), 0,
                     (hipStream_t)stream, x, dy, dx, B, L, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
