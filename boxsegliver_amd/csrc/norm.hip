// Normalisation + activation passes around the conv kernels (HBM-bound, float4 NHWC).
//
// Replaces, after every slim.conv2d(x, C, 3) (NetworksV2/UNet.py:79,85,94; GUNet.py:181-188; UNet3D.py:153,165):
//   slim.batch_norm    (NetworksV2/base.py:153-162: eps 1e-3, decay .999 / .99 in GUNet's encoder, fused;
//                       training: batch mean + biased variance, unbiased variance into the moving average)
//   slim.instance_norm (base.py:163-165: eps 1e-6, moments over the spatial axes per (n, c))
//   optional centre / scale (GUNet.yml: norm_with_center true, norm_with_scale false, GUNet.py:313-330)
//   GUNet's spatial modulation  net + sp_params[..., slice]  (GUNet.py:207-212) where sp_params is the
//     1x1 conv of the pooled guide (GUNet.py:154-156): computed on the fly, never materialised
//   ReLU
// The conv epilogue already produced per-tile sum / sum-of-squares partials, so the forward is one tiny
// finalise + ONE read-modify-write pass; the backward is two passes over (y, dz).
//
// One descriptor drives all of them: statistics groups Ns = per_sample ? N : 1, each over P pixels.
#include "common.h"

namespace {

constexpr int MAXG = 4;   // guide channels

struct NormGeom {
  int Ns;        // statistic groups (1 for batch norm, N for instance norm)
  int64_t P;     // pixels per group
  int C, cq_n, rpi;
};

NormGeom geom(const unetk_norm_desc* d) {
  NormGeom g;
  g.Ns = d->per_sample ? d->N : 1;
  g.P = d->per_sample ? (int64_t)d->HW : (int64_t)d->N * d->HW;
  g.C = d->C;
  const ColMap m = unetk_colmap(d->C);
  g.cq_n = m.cq_n;
  g.rpi = m.rows_per_iter;
  return g;
}

// sums[0][g][c] = sum y, sums[1][g][c] = sum y^2 over `count` elements of group g
__global__ void norm_finalize_kernel(const float* __restrict__ sums, int Ns, int C, double count,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     float decay, int use_moving, int update_moving, float* __restrict__ moving_mean,
                                     float* __restrict__ moving_var, float* __restrict__ mean_out,
                                     float* __restrict__ rstd_out, float* __restrict__ scale_out,
                                     float* __restrict__ shift_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ns * C) return;
  const int c = i % C;
  float mean, var;
  if (!use_moving) {
    const double m = (double)sums[i] / count;
    double v = (double)sums[Ns * C + i] / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (update_moving) {   // batch norm only (Ns == 1)
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
  mean_out[i] = mean;
  rstd_out[i] = rstd;
  scale_out[i] = scale;
  shift_out[i] = b - mean * scale;
}

struct ApplyArgs {
  const float* y;
  const float* scale;   // [Ns][C]
  const float* shift;
  const float* guide;   // [N*HW][G] or null
  const float* gw;      // [G][gw_stride], columns gw_coff ..
  const float* gb;      // [gw_stride]
  float* z;
  int64_t P;
  int C, zs, cq_n, rpi, gw_stride, gw_coff;
};

// z = relu(y*scale + shift [+ guide . gw + gb])
template <int G>
__global__ __launch_bounds__(256) void norm_apply_relu_kernel(ApplyArgs a) {
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  if (rl >= a.rpi) return;
  const int n = blockIdx.y;
  const float4 sc = ldg4(a.scale + (int64_t)n * a.C + cq * 4);
  float4 sh = ldg4(a.shift + (int64_t)n * a.C + cq * 4);
  float4 gwv[G > 0 ? G : 1];
  if (G > 0) {
    const float4 gbv = ldg4(a.gb + a.gw_coff + cq * 4);
    sh.x += gbv.x; sh.y += gbv.y; sh.z += gbv.z; sh.w += gbv.w;
#pragma unroll
    for (int g = 0; g < G; ++g) gwv[g] = ldg4(a.gw + (int64_t)g * a.gw_stride + a.gw_coff + cq * 4);
  }
  const int64_t base = (int64_t)n * a.P;
  for (int64_t pix = (int64_t)blockIdx.x * a.rpi + rl; pix < a.P; pix += (int64_t)gridDim.x * a.rpi) {
    const float4 v = ldg4(a.y + (base + pix) * a.C + cq * 4);
    float4 u;
    u.x = fmaf(v.x, sc.x, sh.x); u.y = fmaf(v.y, sc.y, sh.y); u.z = fmaf(v.z, sc.z, sh.z); u.w = fmaf(v.w, sc.w, sh.w);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float gg = a.guide[(base + pix) * G + g];
      u.x = fmaf(gg, gwv[g].x, u.x); u.y = fmaf(gg, gwv[g].y, u.y); u.z = fmaf(gg, gwv[g].z, u.z); u.w = fmaf(gg, gwv[g].w, u.w);
    }
    u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f);
    stg4(a.z + (base + pix) * a.zs + cq * 4, u);
  }
}

struct BwdArgs {
  const float* y;
  const float* dz;
  const float* scale;   // [Ns][C]
  const float* shift;
  const float* mean;
  const float* rstd;
  const float* guide;
  const float* gw;
  const float* gb;
  const float* sums;    // [K][Ns][C]   (apply pass)
  float* partial;       // [K][Ns][nblk][C] (reduce pass)
  float* dy;
  int64_t P;
  int C, dzs, cq_n, rpi, gw_stride, gw_coff, Ns, plain;
};

// pass 1: partial[0] = sum du, partial[1] = sum du*xhat, partial[2+g] = sum du*guide_g   (du = dz * (u > 0))
template <int G>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(BwdArgs a) {
  constexpr int K = 2 + G;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [K][rpi][C]
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  const int n = blockIdx.y;
  float4 s[K];
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < a.rpi) {
    const int64_t so = (int64_t)n * a.C + cq * 4;
    const float4 mu = ldg4(a.mean + so), rs = ldg4(a.rstd + so), sc = ldg4(a.scale + so);
    float4 sh = ldg4(a.shift + so);
    float4 gwv[G > 0 ? G : 1];
    if (G > 0) {
      const float4 gbv = ldg4(a.gb + a.gw_coff + cq * 4);
      sh.x += gbv.x; sh.y += gbv.y; sh.z += gbv.z; sh.w += gbv.w;
#pragma unroll
      for (int g = 0; g < G; ++g) gwv[g] = ldg4(a.gw + (int64_t)g * a.gw_stride + a.gw_coff + cq * 4);
    }
    const int64_t base = (int64_t)n * a.P;
    for (int64_t pix = (int64_t)blockIdx.x * a.rpi + rl; pix < a.P; pix += (int64_t)gridDim.x * a.rpi) {
      const float4 v = ldg4(a.y + (base + pix) * a.C + cq * 4);
      const float4 d = ldg4(a.dz + (base + pix) * a.dzs + cq * 4);
      float4 u;
      u.x = fmaf(v.x, sc.x, sh.x); u.y = fmaf(v.y, sc.y, sh.y); u.z = fmaf(v.z, sc.z, sh.z); u.w = fmaf(v.w, sc.w, sh.w);
      float gg[G > 0 ? G : 1];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        gg[g] = a.guide[(base + pix) * G + g];
        u.x = fmaf(gg[g], gwv[g].x, u.x); u.y = fmaf(gg[g], gwv[g].y, u.y); u.z = fmaf(gg[g], gwv[g].z, u.z); u.w = fmaf(gg[g], gwv[g].w, u.w);
      }
#define NBR(f)                                       \
  {                                                  \
    const float du = u.f > 0.f ? d.f : 0.f;          \
    s[0].f += du;                                    \
    s[1].f += du * ((v.f - mu.f) * rs.f);            \
    _Pragma("unroll") for (int g = 0; g < G; ++g) s[2 + g].f += du * gg[g]; \
  }
      NBR(x) NBR(y) NBR(z) NBR(w)
#undef NBR
    }
#pragma unroll
    for (int k = 0; k < K; ++k) stg4(&smem[(k * a.rpi + rl) * a.C + cq * 4], s[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * a.C; i += 256) {
    const int k = i / a.C, c = i - k * a.C;
    float t = 0.f;
    for (int j = 0; j < a.rpi; ++j) t += smem[(k * a.rpi + j) * a.C + c];
    a.partial[(((int64_t)k * a.Ns + n) * gridDim.x + blockIdx.x) * a.C + c] = t;
  }
}

// pass 2: dy = scale * (du - sum_du/P - xhat * sum_du_xhat/P)
template <int G>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(BwdArgs a) {
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  if (rl >= a.rpi) return;
  const int n = blockIdx.y;
  const float inv_p = 1.0f / (float)a.P;
  const int64_t so = (int64_t)n * a.C + cq * 4;
  const float4 mu = ldg4(a.mean + so), rs = ldg4(a.rstd + so), sc = ldg4(a.scale + so);
  float4 sh = ldg4(a.shift + so);
  float4 gwv[G > 0 ? G : 1];
  if (G > 0) {
    const float4 gbv = ldg4(a.gb + a.gw_coff + cq * 4);
    sh.x += gbv.x; sh.y += gbv.y; sh.z += gbv.z; sh.w += gbv.w;
#pragma unroll
    for (int g = 0; g < G; ++g) gwv[g] = ldg4(a.gw + (int64_t)g * a.gw_stride + a.gw_coff + cq * 4);
  }
  float4 k1 = ldg4(a.sums + so), k2 = ldg4(a.sums + (int64_t)a.Ns * a.C + so);
  k1.x *= inv_p; k1.y *= inv_p; k1.z *= inv_p; k1.w *= inv_p;
  k2.x *= inv_p; k2.y *= inv_p; k2.z *= inv_p; k2.w *= inv_p;
  if (a.plain) {   // no normalisation (--without_norm): dy = du * scale
    k1 = make_float4(0.f, 0.f, 0.f, 0.f);
    k2 = k1;
  }
  const int64_t base = (int64_t)n * a.P;
  for (int64_t pix = (int64_t)blockIdx.x * a.rpi + rl; pix < a.P; pix += (int64_t)gridDim.x * a.rpi) {
    const float4 v = ldg4(a.y + (base + pix) * a.C + cq * 4);
    const float4 d = ldg4(a.dz + (base + pix) * a.dzs + cq * 4);
    float4 u, o;
    u.x = fmaf(v.x, sc.x, sh.x); u.y = fmaf(v.y, sc.y, sh.y); u.z = fmaf(v.z, sc.z, sh.z); u.w = fmaf(v.w, sc.w, sh.w);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float gg = a.guide[(base + pix) * G + g];
      u.x = fmaf(gg, gwv[g].x, u.x); u.y = fmaf(gg, gwv[g].y, u.y); u.z = fmaf(gg, gwv[g].z, u.z); u.w = fmaf(gg, gwv[g].w, u.w);
    }
#define NBA(f)                                          \
  {                                                     \
    const float du = u.f > 0.f ? d.f : 0.f;             \
    const float xh = (v.f - mu.f) * rs.f;               \
    o.f = sc.f * (du - k1.f - xh * k2.f);               \
  }
    NBA(x) NBA(y) NBA(z) NBA(w)
#undef NBA
    stg4(a.dy + (base + pix) * a.C + cq * 4, o);
  }
}

// parameter gradients from psum[k][c] = sum over statistic groups of sums[k][g][c]
__global__ void norm_bwd_params_kernel(const float* __restrict__ psum, int C, int G, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ dgw, float* __restrict__ dgb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = psum[c];
  if (dgamma) dgamma[c] = psum[C + c];
  if (dgb) dgb[c] = psum[c];
  for (int g = 0; g < G; ++g) dgw[(int64_t)g * C + c] = psum[(int64_t)(2 + g) * C + c];
}

bool norm_desc_ok(const unetk_norm_desc* d) {
  return d && d->N > 0 && d->HW > 0 && d->C > 0 && d->guide_ch >= 0 && d->guide_ch <= MAXG;
}
bool norm_supported(const unetk_norm_desc* d) { return d->C % 4 == 0 && d->C <= 1024; }

int bwd_blocks(const NormGeom& g) {
  int64_t b = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = g.Ns > 1 ? 64 : UNETK_COL_BLOCKS;
  if (b > cap) b = cap;
  return (int)b;
}

#define G_DISPATCH(G_, CALL) \
  switch (G_) {              \
    case 0: { constexpr int GG = 0; CALL; } break; \
    case 1: { constexpr int GG = 1; CALL; } break; \
    case 2: { constexpr int GG = 2; CALL; } break; \
    case 3: { constexpr int GG = 3; CALL; } break; \
    default: { constexpr int GG = 4; CALL; } break; \
  }

}  // namespace

extern "C" size_t unetk_norm_finalize_ws_bytes(const unetk_norm_desc* d, int stat_rows) {
  if (!norm_desc_ok(d) || stat_rows <= 0) return 0;
  const NormGeom g = geom(d);
  const int rows_per_group = stat_rows / g.Ns;
  return (2 * (size_t)g.Ns * d->C + unetk_rows_reduce_tmp_floats(2 * g.Ns, rows_per_group, d->C)) * sizeof(float);
}

extern "C" int unetk_norm_finalize(const unetk_norm_desc* d, const float* stat_partials, int stat_rows,
                                   const float* gamma, const float* beta, float eps, float decay, int training,
                                   float* moving_mean, float* moving_var, float* mean_out, float* rstd_out,
                                   float* scale_out, float* shift_out, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && mean_out && rstd_out && scale_out && shift_out);
  hipStream_t st = (hipStream_t)stream;
  const NormGeom g = geom(d);
  const int use_moving = (!d->per_sample && !training) ? 1 : 0;
  float* sums = nullptr;
  if (!use_moving) {
    UNETK_REQUIRE(stat_partials && stat_rows > 0 && ws && stat_rows % g.Ns == 0);
    if (ws_bytes < unetk_norm_finalize_ws_bytes(d, stat_rows)) return UNETK_E_WORKSPACE;
    sums = (float*)ws;
    // partials are [2][stat_rows][C] with each image's tiles contiguous -> [2*Ns][rows_per_group][C]
    int rc = unetk_rows_reduce(stat_partials, 2 * g.Ns, stat_rows / g.Ns, d->C, sums, sums + 2 * g.Ns * d->C, st);
    if (rc != UNETK_OK) return rc;
  } else {
    UNETK_REQUIRE(moving_mean && moving_var);
  }
  const int update_moving = (!d->per_sample && training && moving_mean && moving_var) ? 1 : 0;
  const int total = g.Ns * d->C;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, st, sums, g.Ns, d->C, (double)g.P,
                     gamma, beta, eps, decay, use_moving, update_moving, moving_mean, moving_var, mean_out, rstd_out,
                     scale_out, shift_out);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_norm_apply_relu(const unetk_norm_desc* d, const float* y, const float* scale, const float* shift,
                                     const float* guide, const float* gw, const float* gb, float* z, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && scale && shift && z && d->z_stride >= d->C);
  if (!norm_supported(d) || d->z_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(y) && unetk_aligned16(z) && unetk_aligned16(scale) && unetk_aligned16(shift));
  if (d->guide_ch > 0) {
    UNETK_REQUIRE(guide && gw && gb && d->gw_stride >= d->gw_coff + d->C);
    UNETK_REQUIRE(d->gw_stride % 4 == 0 && d->gw_coff % 4 == 0 && unetk_aligned16(gw) && unetk_aligned16(gb));
  }
  const NormGeom g = geom(d);
  ApplyArgs a{y, scale, shift, guide, gw, gb, z, g.P, d->C, d->z_stride, g.cq_n, g.rpi, d->gw_stride, d->gw_coff};
  int64_t gx = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = g.Ns > 1 ? (4096 + g.Ns - 1) / g.Ns : 4096;
  if (gx > cap) gx = cap;
  G_DISPATCH(d->guide_ch, hipLaunchKernelGGL(norm_apply_relu_kernel<GG>, dim3((int)gx, g.Ns), dim3(256), 0,
                                             (hipStream_t)stream, a));
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" size_t unetk_norm_bwd_ws_bytes(const unetk_norm_desc* d) {
  if (!norm_desc_ok(d) || !norm_supported(d)) return 0;
  const NormGeom g = geom(d);
  const int K = 2 + d->guide_ch;
  const int nblk = bwd_blocks(g);
  size_t f = (size_t)K * g.Ns * nblk * d->C;                         // partials
  f += (size_t)K * g.Ns * d->C;                                      // sums per group
  f += (size_t)K * d->C;                                             // sums over groups
  f += unetk_rows_reduce_tmp_floats(K * g.Ns, nblk, d->C);
  f += unetk_rows_reduce_tmp_floats(K, g.Ns, d->C);
  return f * sizeof(float);
}

extern "C" int unetk_norm_relu_bwd(const unetk_norm_desc* d, const float* y, const float* dz, int dz_stride,
                                   const float* scale, const float* shift, const float* mean, const float* rstd,
                                   const float* guide, const float* gw, const float* gb, float* dy, float* dgamma,
                                   float* dbeta, float* dgw, float* dgb, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && dz && scale && shift && mean && rstd && dy && ws && dz_stride >= d->C);
  if (!norm_supported(d) || dz_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(y) && unetk_aligned16(dz) && unetk_aligned16(dy) && unetk_aligned16(ws));
  const int G = d->guide_ch;
  if (G > 0) {
    UNETK_REQUIRE(guide && gw && gb && dgw && dgb && d->gw_stride >= d->gw_coff + d->C);
    UNETK_REQUIRE(d->gw_stride % 4 == 0 && d->gw_coff % 4 == 0 && unetk_aligned16(gw) && unetk_aligned16(gb));
  }
  if (ws_bytes < unetk_norm_bwd_ws_bytes(d)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const NormGeom g = geom(d);
  const int K = 2 + G;
  const int nblk = bwd_blocks(g);
  float* partial = (float*)ws;
  float* sums = partial + (size_t)K * g.Ns * nblk * d->C;
  float* psum = sums + (size_t)K * g.Ns * d->C;
  float* tmp1 = psum + (size_t)K * d->C;
  float* tmp2 = tmp1 + unetk_rows_reduce_tmp_floats(K * g.Ns, nblk, d->C);
  BwdArgs a{y, dz, scale, shift, mean, rstd, guide, gw, gb, sums, partial, dy, g.P, d->C, dz_stride, g.cq_n, g.rpi,
            d->gw_stride, d->gw_coff, g.Ns, d->affine_only};
  const size_t lds = (size_t)K * g.rpi * d->C * sizeof(float);
  G_DISPATCH(G, hipLaunchKernelGGL(norm_bwd_reduce_kernel<GG>, dim3(nblk, g.Ns), dim3(256), lds, st, a));
  UNETK_LAUNCH_CHECK();
  int rc = unetk_rows_reduce(partial, K * g.Ns, nblk, d->C, sums, tmp1, st);   // -> sums[K][Ns][C]
  if (rc != UNETK_OK) return rc;
  rc = unetk_rows_reduce(sums, K, g.Ns, d->C, psum, tmp2, st);                 // -> psum[K][C]
  if (rc != UNETK_OK) return rc;
  hipLaunchKernelGGL(norm_bwd_params_kernel, dim3((d->C + 255) / 256), dim3(256), 0, st, psum, d->C, G, dgamma, dbeta,
                     dgw, dgb);
  UNETK_LAUNCH_CHECK();
  int64_t gx = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = g.Ns > 1 ? (4096 + g.Ns - 1) / g.Ns : 4096;
  if (gx > cap) gx = cap;
  G_DISPATCH(G, hipLaunchKernelGGL(norm_bwd_apply_kernel<GG>, dim3((int)gx, g.Ns), dim3(256), 0, st, a));
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
