// Normalisation, activation and pooling passes around the conv kernels (HBM-bound, float4 NHWC).
//
// Replaces slim.batch_norm (NetworksV2/base.py:153-162: eps 1e-3, decay .999, fused) + ReLU that
// slim.conv2d applies after every 3x3 conv (UNet.py:79,85,94), and slim.max_pool2d (UNet.py:81).
// The conv epilogue already produced per-tile sum / sum-of-squares partials, so the forward here is
// one tiny finalise + ONE read-modify-write pass; the backward is two passes over (y, dz).
#include "common.h"

namespace {

// ---------------------------------------------------------------- fixed-order row reductions
// level 1: src[k][rows][C] -> tmp[k][RB][C]; block = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void rows_reduce_l1_kernel(const float* __restrict__ src, int rows, int C,
                                                             int RB, float* __restrict__ tmp) {
  __shared__ double red[4][64];
  const int cblocks = (C + 63) / 64;
  int bid = blockIdx.x;
  const int cb = bid % cblocks; bid /= cblocks;
  const int rb = bid % RB;
  const int k = bid / RB;
  const int c = cb * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int chunk = (rows + RB - 1) / RB;
  const int r0 = rb * chunk, r1 = min(r0 + chunk, rows);
  double s = 0.0;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) s += (double)src[((int64_t)k * rows + r) * C + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C)
    tmp[((int64_t)k * RB + rb) * C + c] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// final: src[k][rows][C] -> dst[k][C]; block = 16 channels x 16 row lanes
__global__ __launch_bounds__(256) void rows_reduce_final_kernel(const float* __restrict__ src, int rows, int C,
                                                                float* __restrict__ dst) {
  __shared__ double red[16][17];
  const int cblocks = (C + 15) / 16;
  const int cb = blockIdx.x % cblocks, k = blockIdx.x / cblocks;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = cb * 16 + cl;
  double s = 0.0;
  if (c < C)
    for (int r = rl; r < rows; r += 16) s += (double)src[((int64_t)k * rows + r) * C + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][cl];
    dst[(int64_t)k * C + c] = (float)t;
  }
}

}  // namespace

size_t unetk_rows_reduce_tmp_floats(int K, int rows, int C) { return rows > 256 ? (size_t)K * 64 * C : 0; }

int unetk_rows_reduce(const float* src, int K, int rows, int C, float* dst, float* tmp, hipStream_t st) {
  if (rows > 256) {
    const int RB = 64;
    const int cblocks = (C + 63) / 64;
    hipLaunchKernelGGL(rows_reduce_l1_kernel, dim3(cblocks * RB * K), dim3(256), 0, st, src, rows, C, RB, tmp);
    UNETK_LAUNCH_CHECK();
    src = tmp;
    rows = RB;
  }
  const int cblocks = (C + 15) / 16;
  hipLaunchKernelGGL(rows_reduce_final_kernel, dim3(cblocks * K), dim3(256), 0, st, src, rows, C, dst);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

namespace {

// ---------------------------------------------------------------- BN finalise
// sums[0][c] = sum y, sums[1][c] = sum y^2 over `count` elements
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float decay, int training,
                                   float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    const double m = (double)sums[c] / count;
    double v = (double)sums[C + c] / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (moving_mean != nullptr) {
      const double unbiased = count > 1.0 ? v * (count / (count - 1.0)) : v;
      moving_mean[c] = moving_mean[c] * decay + mean * (1.f - decay);
      moving_var[c] = moving_var[c] * decay + (float)unbiased * (1.f - decay);
    }
  } else {
    mean = moving_mean[c];
    var = moving_var[c];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  const float scale = g * rstd;
  if (mean_out) mean_out[c] = mean;
  if (rstd_out) rstd_out[c] = rstd;
  scale_out[c] = scale;
  shift_out[c] = b - mean * scale;
}

// ---------------------------------------------------------------- z = relu(y*scale + shift)
__global__ __launch_bounds__(256) void affine_relu_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ z,
                                                          int64_t npix, int C, int zs, int cq_n, int rpi) {
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  if (rl >= rpi) return;
  const float4 sc = ldg4(scale + cq * 4), sh = ldg4(shift + cq * 4);
  for (int64_t pix = (int64_t)blockIdx.x * rpi + rl; pix < npix; pix += (int64_t)gridDim.x * rpi) {
    const float4 v = ldg4(y + pix * C + cq * 4);
    float4 o;
    o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
    o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
    o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    stg4(z + pix * zs + cq * 4, o);
  }
}

// ---------------------------------------------------------------- BN+ReLU backward
// pass 1: partial[0][blk][c] = sum du, partial[1][blk][c] = sum du * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                            int dzs, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ beta, float* __restrict__ partial,
                                                            int64_t npix, int C, int cq_n, int rpi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][rpi][C]
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rl < rpi) {
    const float4 mu = ldg4(mean + cq * 4), rs = ldg4(rstd + cq * 4), g = ldg4(gamma + cq * 4), b = ldg4(beta + cq * 4);
    const float4 sc = make_float4(g.x * rs.x, g.y * rs.y, g.z * rs.z, g.w * rs.w);
    const float4 sh = make_float4(b.x - mu.x * sc.x, b.y - mu.y * sc.y, b.z - mu.z * sc.z, b.w - mu.w * sc.w);
    for (int64_t pix = (int64_t)blockIdx.x * rpi + rl; pix < npix; pix += (int64_t)gridDim.x * rpi) {
      const float4 v = ldg4(y + pix * C + cq * 4);
      const float4 d = ldg4(dz + pix * dzs + cq * 4);
#define BNR(f)                                                  \
  {                                                             \
    const float du = fmaf(v.f, sc.f, sh.f) > 0.f ? d.f : 0.f;   \
    s1.f += du;                                                 \
    s2.f += du * ((v.f - mu.f) * rs.f);                         \
  }
      BNR(x) BNR(y) BNR(z) BNR(w)
#undef BNR
    }
    stg4(&smem[(0 * rpi + rl) * C + cq * 4], s1);
    stg4(&smem[(1 * rpi + rl) * C + cq * 4], s2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float s = 0.f;
    for (int j = 0; j < rpi; ++j) s += smem[(k * rpi + j) * C + c];
    partial[((int64_t)k * gridDim.x + blockIdx.x) * C + c] = s;
  }
}

// pass 2: dy = gamma*rstd * (du - sum_du/M - xhat * sum_du_xhat/M); sums = [2][C]
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                           int dzs, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ beta, const float* __restrict__ sums,
                                                           float* __restrict__ dy, int64_t npix, int C, int cq_n, int rpi) {
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  if (rl >= rpi) return;
  const float inv_m = 1.0f / (float)npix;
  const float4 mu = ldg4(mean + cq * 4), rs = ldg4(rstd + cq * 4), g = ldg4(gamma + cq * 4), b = ldg4(beta + cq * 4);
  const float4 sc = make_float4(g.x * rs.x, g.y * rs.y, g.z * rs.z, g.w * rs.w);
  const float4 sh = make_float4(b.x - mu.x * sc.x, b.y - mu.y * sc.y, b.z - mu.z * sc.z, b.w - mu.w * sc.w);
  float4 k1 = ldg4(sums + cq * 4), k2 = ldg4(sums + C + cq * 4);
  k1.x *= inv_m; k1.y *= inv_m; k1.z *= inv_m; k1.w *= inv_m;
  k2.x *= inv_m; k2.y *= inv_m; k2.z *= inv_m; k2.w *= inv_m;
  for (int64_t pix = (int64_t)blockIdx.x * rpi + rl; pix < npix; pix += (int64_t)gridDim.x * rpi) {
    const float4 v = ldg4(y + pix * C + cq * 4);
    const float4 d = ldg4(dz + pix * dzs + cq * 4);
    float4 o;
#define BNA(f)                                                  \
  {                                                             \
    const float du = fmaf(v.f, sc.f, sh.f) > 0.f ? d.f : 0.f;   \
    const float xh = (v.f - mu.f) * rs.f;                       \
    o.f = sc.f * (du - k1.f - xh * k2.f);                       \
  }
    BNA(x) BNA(y) BNA(z) BNA(w)
#undef BNA
    stg4(dy + pix * C + cq * 4, o);
  }
}

__global__ void bn_bwd_params_kernel(const float* __restrict__ sums, int C, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = sums[c];
  dgamma[c] = sums[C + c];
}

// ---------------------------------------------------------------- max pool 2x2 s2 VALID
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int xs, float* __restrict__ p,
                                                           int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, cq_n = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * cq_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    int64_t r = i / cq_n;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float* b = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * xs + cq * 4;
    const float4 a0 = ldg4(b), a1 = ldg4(b + xs), a2 = ldg4(b + (int64_t)W * xs), a3 = ldg4(b + (int64_t)W * xs + xs);
    float4 o;
    o.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
    o.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
    o.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z));
    o.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
    stg4(p + i * 4, o);
  }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, int xs, const float* __restrict__ p,
                                                           const float* __restrict__ dp, float* __restrict__ dx, int N,
                                                           int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, cq_n = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * cq_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    int64_t r = i / cq_n;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int64_t pix = ((int64_t)n * H + 2 * ho) * W + 2 * wo;
    const float* b = x + pix * xs + cq * 4;
    const float4 a0 = ldg4(b), a1 = ldg4(b + xs), a2 = ldg4(b + (int64_t)W * xs);  // 4th is implied
    const float4 m = ldg4(p + i * 4), g = ldg4(dp + i * 4);
    float4 o0, o1, o2, o3;
    // first maximum in window scan order gets the gradient (TF MaxPoolGrad)
#define MPB(f)                                              \
  {                                                         \
    const bool e0 = a0.f == m.f;                            \
    const bool e1 = !e0 && a1.f == m.f;                     \
    const bool e2 = !e0 && !e1 && a2.f == m.f;              \
    const bool e3 = !e0 && !e1 && !e2;                      \
    o0.f = e0 ? g.f : 0.f;                                  \
    o1.f = e1 ? g.f : 0.f;                                  \
    o2.f = e2 ? g.f : 0.f;                                  \
    o3.f = e3 ? g.f : 0.f;                                  \
  }
    MPB(x) MPB(y) MPB(z) MPB(w)
#undef MPB
    float* d = dx + pix * C + cq * 4;
    stg4(d, o0);
    stg4(d + C, o1);
    stg4(d + (int64_t)W * C, o2);
    stg4(d + (int64_t)W * C + C, o3);
  }
}

inline int ew_grid(int64_t total_threads) {
  int64_t g = (total_threads + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" size_t unetk_bn_finalize_ws_bytes(int stat_rows, int C) {
  return (2 * (size_t)C + unetk_rows_reduce_tmp_floats(2, stat_rows, C)) * sizeof(float);
}

extern "C" int unetk_bn_finalize(const float* stat_partials, int stat_rows, int C, int64_t count,
                                 const float* gamma, const float* beta, float eps, float decay, int training,
                                 float* moving_mean, float* moving_var, float* mean_out, float* rstd_out,
                                 float* scale_out, float* shift_out, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(C > 0 && scale_out && shift_out);
  hipStream_t st = (hipStream_t)stream;
  float* sums = nullptr;
  if (training) {
    UNETK_REQUIRE(stat_partials && stat_rows > 0 && count > 0 && ws);
    if (ws_bytes < unetk_bn_finalize_ws_bytes(stat_rows, C)) return UNETK_E_WORKSPACE;
    sums = (float*)ws;
    int rc = unetk_rows_reduce(stat_partials, 2, stat_rows, C, sums, sums + 2 * C, st);
    if (rc != UNETK_OK) return rc;
  } else {
    UNETK_REQUIRE(moving_mean && moving_var);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, (double)count, gamma,
                     beta, eps, decay, training, moving_mean, moving_var, mean_out, rstd_out, scale_out, shift_out);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_affine_relu(const float* y, const float* scale, const float* shift, float* z, int64_t npix,
                                 int C, int z_stride, void* stream) {
  UNETK_REQUIRE(y && scale && shift && z && npix > 0 && C > 0 && z_stride >= C);
  if (C % 4 != 0 || C > 1024 || z_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(y) && unetk_aligned16(z) && unetk_aligned16(scale) && unetk_aligned16(shift));
  const ColMap m = unetk_colmap(C);
  int64_t g = (npix + m.rows_per_iter - 1) / m.rows_per_iter;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(affine_relu_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, y, scale, shift, z, npix, C,
                     z_stride, m.cq_n, m.rows_per_iter);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

static int bn_bwd_blocks(int64_t npix, int C) {
  const ColMap m = unetk_colmap(C);
  int64_t g = (npix + m.rows_per_iter - 1) / m.rows_per_iter;
  if (g > UNETK_COL_BLOCKS) g = UNETK_COL_BLOCKS;
  return (int)g;
}

extern "C" size_t unetk_bn_bwd_ws_bytes(int64_t npix, int C) {
  if (npix <= 0 || C <= 0 || C % 4 != 0 || C > 1024) return 0;
  const int nblk = bn_bwd_blocks(npix, C);
  return ((size_t)2 * nblk * C + 2 * (size_t)C + unetk_rows_reduce_tmp_floats(2, nblk, C)) * sizeof(float);
}

extern "C" int unetk_bn_relu_bwd(const float* y, const float* dz, int dz_stride, const float* gamma,
                                 const float* mean, const float* rstd, const float* beta, float* dy, float* dgamma,
                                 float* dbeta, int64_t npix, int C, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(y && dz && gamma && mean && rstd && beta && dy && dgamma && dbeta && ws);
  UNETK_REQUIRE(npix > 0 && C > 0 && dz_stride >= C);
  if (C % 4 != 0 || C > 1024 || dz_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(y) && unetk_aligned16(dz) && unetk_aligned16(dy) && unetk_aligned16(ws));
  if (ws_bytes < unetk_bn_bwd_ws_bytes(npix, C)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const ColMap m = unetk_colmap(C);
  const int nblk = bn_bwd_blocks(npix, C);
  float* partial = (float*)ws;
  float* sums = partial + (size_t)2 * nblk * C;
  float* tmp = sums + 2 * C;
  const size_t lds = (size_t)2 * m.rows_per_iter * C * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), lds, st, y, dz, dz_stride, gamma, mean, rstd, beta,
                     partial, npix, C, m.cq_n, m.rows_per_iter);
  UNETK_LAUNCH_CHECK();
  int rc = unetk_rows_reduce(partial, 2, nblk, C, sums, tmp, st);
  if (rc != UNETK_OK) return rc;
  hipLaunchKernelGGL(bn_bwd_params_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, dgamma, dbeta);
  UNETK_LAUNCH_CHECK();
  int64_t g = (npix + m.rows_per_iter - 1) / m.rows_per_iter;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((int)g), dim3(256), 0, st, y, dz, dz_stride, gamma, mean, rstd, beta,
                     sums, dy, npix, C, m.cq_n, m.rows_per_iter);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool2_fwd(const float* x, int x_stride, float* p, int N, int H, int W, int C,
                                  void* stream) {
  UNETK_REQUIRE(x && p && N > 0 && H > 1 && W > 1 && C > 0 && x_stride >= C);
  if (C % 4 != 0 || x_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(p));
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_stride, p, N,
                     H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool2_bwd(const float* x, int x_stride, const float* p, const float* dp, float* dx, int N,
                                  int H, int W, int C, void* stream) {
  UNETK_REQUIRE(x && p && dp && dx && N > 0 && H > 1 && W > 1 && C > 0 && x_stride >= C);
  if (C % 4 != 0 || x_stride % 4 != 0 || (H & 1) || (W & 1)) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(p) && unetk_aligned16(dp) && unetk_aligned16(dx));
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_stride, p, dp,
                     dx, N, H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
