// GUNet's context MLP: slim.fully_connected (+ slim.dropout) on [bs, k] feature vectors.
//
// Replaces NetworksV2/Backbone/slim_nets.py:34-57 `mlp` as called by GUNet.py:31-60 `_context_subnets`:
//   fc1 (ReLU) -> dropout -> fc2 (ReLU) -> dropout -> fc3 (linear, he_normal) = the density gains den[bs][n_mod].
// bs <= a few dozen rows, k / n <= a few thousand: latency-sized work, so plain FMA loops with coalesced weight
// reads (one lane per output column) and fixed summation orders (bit-reproducible); no MFMA.
#include "common.h"

namespace {

__device__ __forceinline__ float fc_uniform(uint32_t seed, uint32_t idx) { return unetk_uniform(seed, idx); }

__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     float* __restrict__ mask, int k, int n, int relu, float keep_prob,
                                                     uint32_t seed) {
  const int o = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (o >= n) return;
  const float* xr = x + (int64_t)r * k;
  float acc = b ? b[o] : 0.f;
  for (int i = 0; i < k; ++i) acc = fmaf(xr[i], w[(int64_t)i * n + o], acc);
  if (relu == 1) acc = fmaxf(acc, 0.f);
  else if (relu == 2) acc = 1.0f / (1.0f + expf(-acc));        // tf.nn.sigmoid (GUNet's SE gate, GUNet.py:199)
  if (mask) {
    const float m = fc_uniform(seed, (uint32_t)(r * n + o)) < keep_prob ? 1.0f / keep_prob : 0.f;
    mask[(int64_t)r * n + o] = m;
    acc *= m;
  }
  y[(int64_t)r * n + o] = acc;
}

// dpre = dy * mask gated by the ReLU; db[o] = sum_r dpre[r][o]
__global__ __launch_bounds__(256) void fc_bwd_pre_kernel(const float* __restrict__ y, const float* __restrict__ mask,
                                                         const float* __restrict__ dy, float* __restrict__ dpre,
                                                         float* __restrict__ db, int B, int n, int relu) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  float s = 0.f;
  for (int r = 0; r < B; ++r) {
    const int64_t i = (int64_t)r * n + o;
    float d = dy[i];
    if (mask) d *= mask[i];
    if (relu == 1 && !(y[i] > 0.f)) d = 0.f;
    else if (relu == 2) d *= y[i] * (1.0f - y[i]);
    dpre[i] = d;
    s += d;
  }
  if (db) db[o] = s;
}

// dw[i][o] = sum_r x[r][i] * dpre[r][o]
__global__ __launch_bounds__(256) void fc_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dpre,
                                                       float* __restrict__ dw, int B, int k, int n) {
  const int o = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (o >= n) return;
  float s = 0.f;
  for (int r = 0; r < B; ++r) s = fmaf(x[(int64_t)r * k + i], dpre[(int64_t)r * n + o], s);
  dw[(int64_t)i * n + o] = s;
}

// dx[r][i] = sum_o dpre[r][o] * w[i][o]: one wave per (r, i), lanes stride over o
__global__ __launch_bounds__(64) void fc_bwd_x_kernel(const float* __restrict__ dpre, const float* __restrict__ w,
                                                      float* __restrict__ dx, int k, int n) {
  const int i = blockIdx.x, r = blockIdx.y;
  float s = 0.f;
  for (int o = threadIdx.x; o < n; o += 64) s = fmaf(dpre[(int64_t)r * n + o], w[(int64_t)i * n + o], s);
  s = wave_sum(s);
  if (threadIdx.x == 0) dx[(int64_t)r * k + i] = s;
}

}  // namespace

extern "C" int unetk_fc_fwd(const float* x, const float* w, const float* b, float* y, float* mask, int B, int k, int n,
                            int relu, float keep_prob, uint32_t seed, void* stream) {
  UNETK_REQUIRE(x && w && y && B > 0 && k > 0 && n > 0 && B <= 65535);
  UNETK_REQUIRE(!mask || (keep_prob > 0.f && keep_prob <= 1.f));
  UNETK_LAUNCH(fc_fwd_kernel, dim3((n + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, w, b, y, mask, k, n,
                     relu, keep_prob, seed);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_fc_bwd(const float* x, const float* w, const float* y, const float* mask, const float* dy, float* dx,
                            float* dw, float* db, float* dpre_ws, int B, int k, int n, int relu, void* stream) {
  UNETK_REQUIRE(x && w && y && dy && dw && dpre_ws && B > 0 && k > 0 && n > 0 && B <= 65535 && k <= 65535);
  hipStream_t st = (hipStream_t)stream;
  UNETK_LAUNCH(fc_bwd_pre_kernel, dim3((n + 255) / 256), dim3(256), 0, st, y, mask, dy, dpre_ws, db, B, n, relu);
  UNETK_LAUNCH_CHECK();
  UNETK_LAUNCH(fc_bwd_w_kernel, dim3((n + 255) / 256, k), dim3(256), 0, st, x, dpre_ws, dw, B, k, n);
  UNETK_LAUNCH_CHECK();
  if (dx) {
    UNETK_LAUNCH(fc_bwd_x_kernel, dim3(k, B), dim3(64), 0, st, dpre_ws, w, dx, k, n);
    UNETK_LAUNCH_CHECK();
  }
  return UNETK_OK;
}
