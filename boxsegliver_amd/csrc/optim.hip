// Optimiser step on a flat fp32 parameter buffer (HBM-bound: 4 reads + 3 writes per element).
//
// Replaces tf.train.AdamOptimizer / MomentumOptimizer built at core/solver.py:204-219 and applied by
// optimizer.minimize at :236-241, plus the gradient of slim.l2_regularizer (NetworksV2/base.py:128-135),
// with TF's formulas (epsilon OUTSIDE the bias correction -- differs from torch.optim.Adam).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                                   float eps, float gscale, float l2, float dwd) {
  const int64_t n4 = n >> 2;
  const float keep = 1.f - dwd;   // tf.contrib.opt.AdamWOptimizer: var -= weight_decay * var, then the Adam update
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = ldg4(p + i * 4), mv = ldg4(m + i * 4), vv = ldg4(v + i * 4);
    const float4 gv = ldg4(g + i * 4);
#define ADAM1(f)                                  \
  {                                               \
    const float gg = fmaf(l2, pv.f, gv.f * gscale); \
    mv.f += c1 * (gg - mv.f);                     \
    vv.f += c2 * (gg * gg - vv.f);                \
    pv.f = pv.f * keep - lr_t * mv.f / (sqrtf(vv.f) + eps); \
  }
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    stg4(p + i * 4, pv);
    stg4(m + i * 4, mv);
    stg4(v + i * 4, vv);
  }
  // tail (n % 4)
  const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float gg = fmaf(l2, p[t], g[t] * gscale);
    m[t] += c1 * (gg - m[t]);
    v[t] += c2 * (gg * gg - v[t]);
    p[t] = p[t] * keep - lr_t * m[t] / (sqrtf(v[t]) + eps);
  }
}

__global__ __launch_bounds__(256) void momentum_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ acc, int64_t n, float lr, float mom,
                                                       int nesterov, float gscale, float l2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gg = fmaf(l2, p[i], g[i] * gscale);
    const float a = mom * acc[i] + gg;
    acc[i] = a;
    p[i] -= nesterov ? lr * (gg + mom * a) : lr * a;
  }
}

// Sum of squares with float64 accumulation in a fixed order (per-thread strided chains -> wave -> block -> one block over the
// partials): float4 loads with four independent chains (the scalar version ran at 2.2 TB/s), and the last level is a
// 256-thread tree (one thread adding 1024 partials took as long as the pass over the 124 MB itself).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ p, int64_t n, double* __restrict__ part, int vec) {
  __shared__ double red[4];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const int64_t n4 = vec ? n >> 2 : 0, stride = (int64_t)gridDim.x * blockDim.x;
  const float4* p4 = reinterpret_cast<const float4*>(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = p4[i];
    s0 += (double)v.x * (double)v.x;
    s1 += (double)v.y * (double)v.y;
    s2 += (double)v.z * (double)v.z;
    s3 += (double)v.w * (double)v.w;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {   // n % 4 tail (all of
    const double v = (double)p[i];                                                                      // a misaligned p)
    s0 += v * v;
  }
  double s = wave_sum_d((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

inline int flat_grid(int64_t n4) {
  int64_t g = (n4 + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int unetk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                               float beta2, float eps, float gscale, float l2, float decoupled_wd, void* stream) {
  UNETK_REQUIRE(p && g && m && v && n > 0);
  UNETK_REQUIRE(unetk_aligned16(p) && unetk_aligned16(g) && unetk_aligned16(m) && unetk_aligned16(v));
  UNETK_LAUNCH(adam_kernel, dim3(flat_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, beta1,
                     beta2, eps, gscale, l2, decoupled_wd);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_momentum_step(float* p, const float* g, float* acc, int64_t n, float lr, float mom, int nesterov,
                                   float gscale, float l2, void* stream) {
  UNETK_REQUIRE(p && g && acc && n > 0);
  UNETK_LAUNCH(momentum_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, acc, n, lr, mom,
                     nesterov, gscale, l2);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_sumsq(const float* p, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(p && out && ws && n > 0);
  if (ws_bytes < 1024 * sizeof(double)) return UNETK_E_WORKSPACE;
  int nb = flat_grid(n);
  if (nb > 1024) nb = 1024;
  UNETK_LAUNCH(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, n, (double*)ws,
                     unetk_aligned16(p) ? 1 : 0);
  UNETK_LAUNCH_CHECK();
  UNETK_LAUNCH(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, nb, out);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// NanTensorHook(loss) of the reference (core/estimator.py:676) without a host sync per step: a sticky device flag.
__global__ void nan_watch_kernel(const float* __restrict__ value, int32_t* __restrict__ flag, int32_t step) {
  const float v = value[0];
  if (v != v && flag[0] == 0) {
    flag[0] = 1;
    flag[1] = step;
  }
}

extern "C" int unetk_nan_watch(const float* value, int32_t* flag, int32_t step, void* stream) {
  UNETK_REQUIRE(value && flag);
  UNETK_LAUNCH(nan_watch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, value, flag, step);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_abi_version(void) { return 10; }   // 2: unetk_conv_desc.precision, UNETK_BF16; 3: density modulation, unetk_fc_*; 4: unetk_conv_desc.dilation; 5: UNETK_BF16S (bf16 storage), unetk_norm_desc.storage, unetk_head_desc.storage; 6: norm dropout / guide_alpha / guide_per_sample, unetk_norm_se_bwd_add, fc sigmoid; 7: unetk_conv3x3_dgrad_nbr, unetk_norm_relu_bwd_pre, unetk_conv1d_*, unetk_maxpool1d_*, unetk_spatial_mean_*, unetk_conv3x3_*_ws, unetk_pack_many; 8: unetk_prof_* (kernel trace), unetk_nan_watch; 9: unetk_deconv{2x2,3d}_bwd_parts; 10: unetk_conv3d_desc.cin_live8 / cout_live8

extern "C" const char* unetk_error_string(int code) {
  switch (code) {
    case UNETK_OK: return "ok";
    case UNETK_E_BADARG: return "bad argument (null pointer, non-positive size or misaligned buffer)";
    case UNETK_E_UNSUPPORTED: return "shape not supported by this build of libunetk";
    case UNETK_E_WORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
