// Normalisation + activation passes around the conv kernels (HBM-bound, float4 NHWC).
//
// Replaces, after every slim.conv2d(x, C, 3) (NetworksV2/UNet.py:79,85,94; GUNet.py:181-188; UNet3D.py:153,165):
//   slim.batch_norm    (NetworksV2/base.py:153-162: eps 1e-3, decay .999 / .99 in GUNet's encoder, fused;
//                       training: batch mean + biased variance, unbiased variance into the moving average)
//   slim.instance_norm (base.py:163-165: eps 1e-6, moments over the spatial axes per (n, c))
//   optional centre / scale (GUNet.yml: norm_with_center true, norm_with_scale false, GUNet.py:313-330)
//   GUNet's density modulation  net * den[b, c]  (`conditional_normalization`, GUNet.py:119-133,203-206; den = a slice of
//     the context MLP's output) -- template flag D
//   GUNet's spatial modulation  net + sp_params[..., slice]  (GUNet.py:207-212) where sp_params is the
//     1x1 conv of the pooled guide (GUNet.py:154-156): computed on the fly, never materialised
//   ReLU
// i.e.  t = (y - mean) rstd gamma + beta,  u = t * den[b,c] + guide . gw + gb,  z = relu(u).
// gb without a guide (guide_ch == 0) is a bare per-channel shift after the gain: with den = gain * affine-gamma that is
// `after_affine` (slim_nets.channel_wise_affine, GUNet.py:213-214) folded into the same pass.
// The conv epilogue already produced per-tile sum / sum-of-squares partials, so the forward is one tiny
// finalise + ONE read-modify-write pass; the backward is two passes over (y, dz).
//
// One descriptor drives all of them.  STATISTICS groups: Ns = per_sample ? N : 1, each over P_stat pixels.  LAUNCH
// groups (blockIdx.y): L = N when the pass needs the sample index (instance norm, or density modulation under batch
// norm), else 1; a launch group indexes the statistics with stride sst (C or 0).
#include "common.h"

namespace {

constexpr int MAXG = 4;   // guide channels

// adjacent row groups per pass in the plain norm kernels (measured: see norm_apply_relu_kernel); 1 = the plain loop
#ifndef UNETK_NORM_GROUPS
#define UNETK_NORM_GROUPS 4
#endif
// the same for bf16 storage (8-byte loads per thread: twice the groups keep the same bytes in flight) and for the reduction pass
#ifndef UNETK_NORM_GROUPS_BF
#define UNETK_NORM_GROUPS_BF 4
#endif
#ifndef UNETK_NORM_RGROUPS
#define UNETK_NORM_RGROUPS 2
#endif
#ifndef UNETK_NORM_RGROUPS_BF
#define UNETK_NORM_RGROUPS_BF 2
#endif

struct NormGeom {
  int Ns;        // statistic groups (1 for batch norm, N for instance norm)
  int64_t Ps;    // pixels per statistic group
  int L;         // launch groups
  int64_t P;     // pixels per launch group
  int sst;       // statistics stride of a launch group (C or 0)
  int C, cq_n, rpi;
};

NormGeom geom(const unetk_norm_desc* d, bool density) {
  NormGeom g;
  g.Ns = d->per_sample ? d->N : 1;
  g.Ps = d->per_sample ? (int64_t)d->HW : (int64_t)d->N * d->HW;
  g.L = (d->per_sample || density) ? d->N : 1;
  g.P = g.L > 1 ? (int64_t)d->HW : (int64_t)d->N * d->HW;
  g.sst = d->per_sample ? d->C : 0;
  g.C = d->C;
  const ColMap m = unetk_colmap(d->C);
  g.cq_n = m.cq_n;
  g.rpi = m.rows_per_iter;
  return g;
}

// Last level of the statistics reduction fused with the finalisation: src[2][Ns][rows][C] (rows <= 256: the conv
// epilogue's partials, or their first-level sums) -> mean / rstd / scale / shift.  Block = 16 channels x 16 row lanes of
// one group; fp64 row sums in a fixed order, rounded to fp32 exactly as unetk_rows_reduce + norm_finalize_kernel did.
__global__ __launch_bounds__(256) void norm_reduce_finalize_kernel(
    const float* __restrict__ src, int rows, int Ns, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float decay, int update_moving, float* __restrict__ moving_mean,
    float* __restrict__ moving_var, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    float* __restrict__ scale_out, float* __restrict__ shift_out) {
  __shared__ double red[2][16][17];
  const int cblocks = (C + 15) / 16;
  const int cb = blockIdx.x % cblocks, gi = blockIdx.x / cblocks;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = cb * 16 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C)
    for (int r = rl; r < rows; r += 16) {
      s0 += (double)src[((int64_t)gi * rows + r) * C + c];
      s1 += (double)src[((int64_t)(Ns + gi) * rows + r) * C + c];
    }
  red[0][rl][cl] = s0;
  red[1][rl][cl] = s1;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  double t0 = 0.0, t1 = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    t0 += red[0][j][cl];
    t1 += red[1][j][cl];
  }
  const double m = (double)(float)t0 / count;
  double v = (double)(float)t1 / count - m * m;
  if (v < 0.0) v = 0.0;
  const float mean = (float)m, var = (float)v;
  if (update_moving) {   // batch norm only (Ns == 1)
    const double unbiased = count > 1.0 ? v * (count / (count - 1.0)) : v;
    moving_mean[c] = moving_mean[c] * decay + mean * (1.f - decay);
    moving_var[c] = moving_var[c] * decay + (float)unbiased * (1.f - decay);
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  const float scale = g * rstd;
  const int i = gi * C + c;
  mean_out[i] = mean;
  rstd_out[i] = rstd;
  scale_out[i] = scale;
  shift_out[i] = b - mean * scale;
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
  const void* y;        // T = float, or bf16_t under d->storage == UNETK_BF16S (then z is bf16 too)
  const float* scale;   // [Ns][C]
  const float* shift;
  const float* den;     // [N][C] or null
  const float* guide;   // [N*HW][G] or null
  const float* gw;      // [G][gw_stride], columns gw_coff ..
  const float* gb;      // [gw_stride]
  void* z;
  int64_t P;
  int C, zs, cq_n, rpi, gw_stride, gw_coff, sst;
  int gw_ns, gb_ns;     // per-sample strides of gw / gb (0 = shared by the batch)
  int post;             // guide_leaky == 3: per-channel slopes + post-shift in the gb block (see post_act)
  float alpha;          // leaky slope of the guide branch (L)
  float keep;           // dropout keep probability, 0 = no dropout
  uint32_t seed;
};

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// LGNet's guide branch (LGNet.py:30-55): the 1x1 guide conv has a leaky-ReLU activation (tf.nn.leaky_relu, alpha 0.2)
// before it is added: u = t + lrelu(guide . gw + gb) -- template flag L (without density modulation).
// alpha = d->guide_alpha: 0.2 for LGNet, 0 (plain ReLU) for GUNet --fix
__device__ __forceinline__ float lrelu(float s, float al) { return s > 0.f ? s : al * s; }
__device__ __forceinline__ float lrelu_grad(float s, float al) { return s > 0.f ? 1.f : al; }
// guide_leaky == 3 ("post"): the guide branch's activation has PER-CHANNEL slopes for s > 0 / s <= 0 and a per-channel shift is
// added behind it: u = t * den + (s > 0 ? ap : an) * s + ps, s = guide . gw + gb.  This is GUNet after_affine with --fix,
//   (t * den + relu(sg)) * gamma' + beta' = t * (den gamma') + gamma' relu(sg) + beta',
// with gamma' folded into gw / gb on the host (s = gamma' sg): gamma' relu(sg) = relu(s) where gamma' >= 0 (slopes 1, 0) and
// = min(s, 0) where gamma' < 0 (slopes 0, 1).  The gb block is then [4][gw_stride]: bias, ap, an, ps.
struct PostV { float4 ap, an, ps; };
__device__ __forceinline__ float4 post_act(const float4& s, const PostV& q) {
  return make_float4((s.x > 0.f ? q.ap.x : q.an.x) * s.x + q.ps.x, (s.y > 0.f ? q.ap.y : q.an.y) * s.y + q.ps.y,
                     (s.z > 0.f ? q.ap.z : q.an.z) * s.z + q.ps.z, (s.w > 0.f ? q.ap.w : q.an.w) * s.w + q.ps.w);
}
__device__ __forceinline__ float4 post_slope(const float4& s, const PostV& q) {
  return make_float4(s.x > 0.f ? q.ap.x : q.an.x, s.y > 0.f ? q.ap.y : q.an.y, s.z > 0.f ? q.ap.z : q.an.z, s.w > 0.f ? q.ap.w : q.an.w);
}
__device__ __forceinline__ PostV post_load(const float* gb_chan, int gw_stride) {
  PostV q;
  q.ap = ldg4(gb_chan + gw_stride); q.an = ldg4(gb_chan + 2 * gw_stride); q.ps = ldg4(gb_chan + 3 * gw_stride);
  return q;
}

// slim.dropout between the two convs of a modulated block (GUNet.py:189-190): the mask multiplies the NORMALISED value
// before the density gain / guide term; kept entries are scaled by 1 / keep.  Regenerated from (seed, element) in every pass.
__device__ __forceinline__ float4 drop4(uint32_t seed, uint32_t idx, float keep, float inv_keep) {
  return make_float4(unetk_uniform(seed, idx) < keep ? inv_keep : 0.f, unetk_uniform(seed, idx + 1) < keep ? inv_keep : 0.f,
                     unetk_uniform(seed, idx + 2) < keep ? inv_keep : 0.f, unetk_uniform(seed, idx + 3) < keep ? inv_keep : 0.f);
}

// z = relu((y*scale + shift) [* den] [+ guide . gw + gb])
template <int G, bool D, bool L = false, typename T = float>
__global__ __launch_bounds__(256) void norm_apply_relu_kernel(ApplyArgs a) {
  static_assert(!(L && G == 0), "the leaky guide needs a guide");
  const T* ay = static_cast<const T*>(a.y);
  T* az = static_cast<T*>(a.z);
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  if (rl >= a.rpi) return;
  const int n = blockIdx.y;
  float4 sc = ldg4(a.scale + (int64_t)n * a.sst + cq * 4);
  float4 sh = ldg4(a.shift + (int64_t)n * a.sst + cq * 4);
  if (D) {
    const float4 dn = ldg4(a.den + (int64_t)n * a.C + cq * 4);
    sc = mul4(sc, dn);
    sh = mul4(sh, dn);
  }
  float4 gwv[G > 0 ? G : 1];
  float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.gb) gbv = ldg4(a.gb + (int64_t)n * a.gb_ns + a.gw_coff + cq * 4);   // guide bias, or a bare post-shift when G == 0
  PostV pq{};
  if (L && a.post) pq = post_load(a.gb + (int64_t)n * a.gb_ns + a.gw_coff + cq * 4, a.gw_stride);
  const bool drop = a.keep > 0.f;
  const float inv_keep = drop ? 1.0f / a.keep : 1.f;
  if (!L && !drop) sh = add4(sh, gbv);                 // with dropout the post-shift is added after the mask
  if (G > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) gwv[g] = ldg4(a.gw + (int64_t)n * a.gw_ns + (int64_t)g * a.gw_stride + a.gw_coff + cq * 4);
  }
  const int64_t base = (int64_t)n * a.P;
  const int64_t pstep = (int64_t)gridDim.x * a.rpi;
  int64_t pix = (int64_t)blockIdx.x * a.rpi + rl;
  if (G == 0 && !L && !drop && UNETK_NORM_GROUPS > 1) {
    // plain units (every unit of UNet / UNet3D): NP ADJACENT row groups per pass, all loads issued before any is used.
    // Measured on the headline step: NP = 2 -> 84.0 to 79.4 us per launch; pairing pix with pix + grid stride instead
    // (two far-apart streams per block) was 7 % SLOWER than no pairing.
    constexpr int NP = sizeof(T) == 2 ? UNETK_NORM_GROUPS_BF : UNETK_NORM_GROUPS;
    for (pix = (int64_t)blockIdx.x * NP * a.rpi + rl; pix < a.P; pix += NP * pstep) {
      float4 v[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (pix + j * a.rpi < a.P) v[j] = ld4(ay + (base + pix + j * a.rpi) * a.C + cq * 4);
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (pix + j * a.rpi < a.P) {
          float4 u;
          u.x = fmaxf(fmaf(v[j].x, sc.x, sh.x), 0.f); u.y = fmaxf(fmaf(v[j].y, sc.y, sh.y), 0.f);
          u.z = fmaxf(fmaf(v[j].z, sc.z, sh.z), 0.f); u.w = fmaxf(fmaf(v[j].w, sc.w, sh.w), 0.f);
          st4(az + (base + pix + j * a.rpi) * a.zs + cq * 4, u);
        }
    }
    return;
  }
  for (; pix < a.P; pix += pstep) {
    const float4 v = ld4(ay + (base + pix) * a.C + cq * 4);
    float4 u;
    u.x = fmaf(v.x, sc.x, sh.x); u.y = fmaf(v.y, sc.y, sh.y); u.z = fmaf(v.z, sc.z, sh.z); u.w = fmaf(v.w, sc.w, sh.w);
    if (drop) {
      const float4 m = drop4(a.seed, (uint32_t)((base + pix) * a.C + cq * 4), a.keep, inv_keep);
      u.x *= m.x; u.y *= m.y; u.z *= m.z; u.w *= m.w;
      if (!L) u = add4(u, gbv);
    }
    if (L) {
      float4 s = gbv;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float gg = a.guide[(base + pix) * G + g];
        s.x = fmaf(gg, gwv[g].x, s.x); s.y = fmaf(gg, gwv[g].y, s.y); s.z = fmaf(gg, gwv[g].z, s.z); s.w = fmaf(gg, gwv[g].w, s.w);
      }
      if (a.post) u = add4(u, post_act(s, pq));
      else { u.x += lrelu(s.x, a.alpha); u.y += lrelu(s.y, a.alpha); u.z += lrelu(s.z, a.alpha); u.w += lrelu(s.w, a.alpha); }
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float gg = a.guide[(base + pix) * G + g];
        u.x = fmaf(gg, gwv[g].x, u.x); u.y = fmaf(gg, gwv[g].y, u.y); u.z = fmaf(gg, gwv[g].z, u.z); u.w = fmaf(gg, gwv[g].w, u.w);
      }
    }
    u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f);
    st4(az + (base + pix) * a.zs + cq * 4, u);
  }
}

// z = relu(y * scale + shift) AND p = max_pool2d(z, 2, 2) in one pass (plain units whose activation feeds the pool:
// ops.Conv3x3NormReluPool): a thread owns a window, writes its four z and their maximum -- the pool's read of z disappears.
template <typename T>
__global__ __launch_bounds__(256) void norm_apply_relu_pool_kernel(ApplyArgs a, void* pooled, int H, int W, int L) {
  const T* ay = static_cast<const T*>(a.y);
  T* az = static_cast<T*>(a.z);
  T* ap = static_cast<T*>(pooled);
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  if (rl >= a.rpi) return;
  const int n = blockIdx.y;
  const float4 sc = ldg4(a.scale + (int64_t)n * a.sst + cq * 4);
  const float4 sh = ldg4(a.shift + (int64_t)n * a.sst + cq * 4);
  const int Wo = W >> 1;
  const int64_t per_img = (int64_t)(H >> 1) * Wo;
  const int64_t Q = a.P >> 2;
  for (int64_t q = (int64_t)blockIdx.x * a.rpi + rl; q < Q; q += (int64_t)gridDim.x * a.rpi) {
    const int64_t img = L > 1 ? n : q / per_img;
    const int64_t r = L > 1 ? q : q - img * per_img;
    const int ho = (int)(r / Wo), wo = (int)(r - (int64_t)ho * Wo);
    const int64_t pix = (img * H + 2 * ho) * W + 2 * wo;
    const int64_t off[4] = {0, 1, W, (int64_t)W + 1};
    float4 v[4], u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ld4(ay + (pix + off[i]) * a.C + cq * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i].x = fmaxf(fmaf(v[i].x, sc.x, sh.x), 0.f); u[i].y = fmaxf(fmaf(v[i].y, sc.y, sh.y), 0.f);
      u[i].z = fmaxf(fmaf(v[i].z, sc.z, sh.z), 0.f); u[i].w = fmaxf(fmaf(v[i].w, sc.w, sh.w), 0.f);
      st4(az + (pix + off[i]) * a.zs + cq * 4, u[i]);
    }
    float4 m;          // rounding to the storage type is monotonic: max of the stored values = stored max
    m.x = fmaxf(fmaxf(u[0].x, u[1].x), fmaxf(u[2].x, u[3].x)); m.y = fmaxf(fmaxf(u[0].y, u[1].y), fmaxf(u[2].y, u[3].y));
    m.z = fmaxf(fmaxf(u[0].z, u[1].z), fmaxf(u[2].z, u[3].z)); m.w = fmaxf(fmaxf(u[0].w, u[1].w), fmaxf(u[2].w, u[3].w));
    st4(ap + (img * per_img + r) * a.C + cq * 4, m);
  }
}

struct BwdArgs {
  const void* y;        // T = float or bf16_t (d->storage): y, dz and dy share it
  const void* dz;
  const float* scale;   // [Ns][C]
  const float* shift;
  const float* mean;
  const float* rstd;
  const float* den;     // [N][C] or null
  const float* guide;
  const float* gw;
  const float* gb;
  const float* ksum;    // apply pass: k = 0 row at ksum + n*kst, k = 1 row at ksum + krow + n*kst
  float* partial;       // [K][L][nblk][C] (reduce pass)
  void* dy;
  int64_t P;
  float inv_ps;         // 1 / pixels per STATISTICS group
  int C, dzs, cq_n, rpi, gw_stride, gw_coff, L, plain, sst, kst, krow;
  int gw_ns, gb_ns;     // per-sample strides of gw / gb (0 = shared)
  float alpha, keep;
  uint32_t seed;
  int post;             // guide_leaky == 3 (see post_act)
};

// pass 1.  With dt = du * den (dt = du without density), du = dz * (u > 0), xhat = (y - mean) rstd, t = y*scale + shift:
//   partial[0] = sum dt, partial[1] = sum dt*xhat, partial[2+g] = sum du*guide_g,
//   D only: partial[2+G] = sum du (guide bias gradient), partial[3+G] = sum du*t (density gradient, per sample)
//   L: partial[2+g] = sum du*lrelu'(s)*guide_g, partial[2+G] = sum du*lrelu'(s)  (s = guide . gw + gb; with D as well the
//   guide-bias row 2+G holds this sum instead of sum du, and 3+G the density gradient: GUNet --fix with --use_context)
//   P (guide_leaky == 3, with D and L): one more row, partial[4+G] = sum du -- the gradient of the post-shift
template <int G, bool D, bool L = false, typename T = float, bool P = false>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(BwdArgs a) {
  static_assert(!P || (D && L), "the post vectors come with the leaky guide and the density gains");
  constexpr int K = 2 + G + (D ? 2 : (L ? 1 : 0)) + (P ? 1 : 0);
  const T* ay = static_cast<const T*>(a.y);
  const T* adz = static_cast<const T*>(a.dz);
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [K][rpi][C]
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  const int n = blockIdx.y;
  float4 s[K];
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < a.rpi) {
    const int64_t so = (int64_t)n * a.sst + cq * 4;
    const float4 mu = ldg4(a.mean + so), rs = ldg4(a.rstd + so), sc0 = ldg4(a.scale + so), sh0 = ldg4(a.shift + so);
    float4 dn = make_float4(1.f, 1.f, 1.f, 1.f);
    if (D) dn = ldg4(a.den + (int64_t)n * a.C + cq * 4);
    const float4 sc = D ? mul4(sc0, dn) : sc0;
    float4 sh = D ? mul4(sh0, dn) : sh0;
    float4 gwv[G > 0 ? G : 1];
    float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.gb) gbv = ldg4(a.gb + (int64_t)n * a.gb_ns + a.gw_coff + cq * 4);
    PostV pq{};
    if (P) pq = post_load(a.gb + (int64_t)n * a.gb_ns + a.gw_coff + cq * 4, a.gw_stride);
    const bool drop = a.keep > 0.f;
    const float inv_keep = drop ? 1.0f / a.keep : 1.f;
    if (!L && !drop) sh = add4(sh, gbv);
    if (G > 0) {
#pragma unroll
      for (int g = 0; g < G; ++g) gwv[g] = ldg4(a.gw + (int64_t)n * a.gw_ns + (int64_t)g * a.gw_stride + a.gw_coff + cq * 4);
    }
    const int64_t base = (int64_t)n * a.P;
    const int64_t pstep = (int64_t)gridDim.x * a.rpi;
    int64_t pix = (int64_t)blockIdx.x * a.rpi + rl;
    if (G == 0 && !D && !L && !drop && UNETK_NORM_GROUPS > 1) {
      // plain units: two adjacent row groups per pass (four loads in flight), summed in pixel order (four groups measured
      // slower here -- 101 vs 105 us -- while they help the apply passes)
      constexpr int NP = sizeof(T) == 2 ? UNETK_NORM_RGROUPS_BF : UNETK_NORM_RGROUPS;
      for (pix = (int64_t)blockIdx.x * NP * a.rpi + rl; pix < a.P; pix += NP * pstep) {
        float4 v[NP], d[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          v[j] = d[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pix + j * a.rpi < a.P) {
            v[j] = ld4(ay + (base + pix + j * a.rpi) * a.C + cq * 4);
            d[j] = ld4(adz + (base + pix + j * a.rpi) * a.dzs + cq * 4);
          }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
#define NBR2(f)                                                              \
  {                                                                          \
    const float du = fmaf(v[j].f, sc.f, sh.f) > 0.f ? d[j].f : 0.f;          \
    s[0].f += du;                                                            \
    s[1].f += du * ((v[j].f - mu.f) * rs.f);                                 \
  }
          NBR2(x) NBR2(y) NBR2(z) NBR2(w)
#undef NBR2
        }
      }
      pix = a.P;     // done: skip the single-group loop
    }
    for (; pix < a.P; pix += pstep) {
      const float4 v = ld4(ay + (base + pix) * a.C + cq * 4);
      const float4 d = ld4(adz + (base + pix) * a.dzs + cq * 4);
      float4 u;
      u.x = fmaf(v.x, sc.x, sh.x); u.y = fmaf(v.y, sc.y, sh.y); u.z = fmaf(v.z, sc.z, sh.z); u.w = fmaf(v.w, sc.w, sh.w);
      float4 m = make_float4(1.f, 1.f, 1.f, 1.f);    // dropout mask (0 or 1 / keep) on the normalised value
      if (drop) {
        m = drop4(a.seed, (uint32_t)((base + pix) * a.C + cq * 4), a.keep, inv_keep);
        u.x *= m.x; u.y *= m.y; u.z *= m.z; u.w *= m.w;
        if (!L) u = add4(u, gbv);
      }
      float gg[G > 0 ? G : 1];
      float4 ls = make_float4(1.f, 1.f, 1.f, 1.f);   // lrelu'(s) of the leaky guide branch
      if (L) {
        float4 sg = gbv;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          gg[g] = a.guide[(base + pix) * G + g];
          sg.x = fmaf(gg[g], gwv[g].x, sg.x); sg.y = fmaf(gg[g], gwv[g].y, sg.y); sg.z = fmaf(gg[g], gwv[g].z, sg.z); sg.w = fmaf(gg[g], gwv[g].w, sg.w);
        }
        if (P) {
          u = add4(u, post_act(sg, pq));
          ls = post_slope(sg, pq);
        } else {
          u.x += lrelu(sg.x, a.alpha); u.y += lrelu(sg.y, a.alpha); u.z += lrelu(sg.z, a.alpha); u.w += lrelu(sg.w, a.alpha);
          ls = make_float4(lrelu_grad(sg.x, a.alpha), lrelu_grad(sg.y, a.alpha), lrelu_grad(sg.z, a.alpha), lrelu_grad(sg.w, a.alpha));
        }
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          gg[g] = a.guide[(base + pix) * G + g];
          u.x = fmaf(gg[g], gwv[g].x, u.x); u.y = fmaf(gg[g], gwv[g].y, u.y); u.z = fmaf(gg[g], gwv[g].z, u.z); u.w = fmaf(gg[g], gwv[g].w, u.w);
        }
      }
#define NBR(f)                                                                \
  {                                                                           \
    const float du = u.f > 0.f ? d.f : 0.f;                                   \
    const float dt = (D ? du * dn.f : du) * m.f;                              \
    const float dg = L ? du * ls.f : du;                                      \
    s[0].f += dt;                                                             \
    s[1].f += dt * ((v.f - mu.f) * rs.f);                                     \
    _Pragma("unroll") for (int g = 0; g < G; ++g) s[2 + g].f += dg * gg[g];  \
    if (D) {                                                                  \
      s[2 + G].f += dg;                                                       \
      s[3 + G].f += du * m.f * fmaf(v.f, sc0.f, sh0.f);                       \
      if (P) s[4 + G].f += du;                                                \
    } else if (L) {                                                           \
      s[2 + G].f += dg;                                                       \
    }                                                                         \
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
    a.partial[(((int64_t)k * a.L + n) * gridDim.x + blockIdx.x) * a.C + c] = t;
  }
}

// pass 2: dy = scale * (dt - sum_dt/Ps - xhat * sum_dt_xhat/Ps)   (sums over the STATISTICS group)
template <int G, bool D, bool L = false, typename T = float>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(BwdArgs a) {
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  if (rl >= a.rpi) return;
  const T* ay = static_cast<const T*>(a.y);
  const T* adz = static_cast<const T*>(a.dz);
  T* ady = static_cast<T*>(a.dy);
  const int n = blockIdx.y;
  const int64_t so = (int64_t)n * a.sst + cq * 4;
  const float4 mu = ldg4(a.mean + so), rs = ldg4(a.rstd + so), sc0 = ldg4(a.scale + so), sh0 = ldg4(a.shift + so);
  float4 dn = make_float4(1.f, 1.f, 1.f, 1.f);
  if (D) dn = ldg4(a.den + (int64_t)n * a.C + cq * 4);
  const float4 sc = D ? mul4(sc0, dn) : sc0;
  float4 sh = D ? mul4(sh0, dn) : sh0;
  float4 gwv[G > 0 ? G : 1];
  float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.gb) gbv = ldg4(a.gb + (int64_t)n * a.gb_ns + a.gw_coff + cq * 4);   // guide bias, or a bare post-shift when G == 0
  PostV pq{};
  if (L && a.post) pq = post_load(a.gb + (int64_t)n * a.gb_ns + a.gw_coff + cq * 4, a.gw_stride);
  const bool drop = a.keep > 0.f;
  const float inv_keep = drop ? 1.0f / a.keep : 1.f;
  if (!L && !drop) sh = add4(sh, gbv);
  if (G > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) gwv[g] = ldg4(a.gw + (int64_t)n * a.gw_ns + (int64_t)g * a.gw_stride + a.gw_coff + cq * 4);
  }
  float4 k1 = ldg4(a.ksum + (int64_t)n * a.kst + cq * 4), k2 = ldg4(a.ksum + a.krow + (int64_t)n * a.kst + cq * 4);
  k1.x *= a.inv_ps; k1.y *= a.inv_ps; k1.z *= a.inv_ps; k1.w *= a.inv_ps;
  k2.x *= a.inv_ps; k2.y *= a.inv_ps; k2.z *= a.inv_ps; k2.w *= a.inv_ps;
  if (a.plain) {   // no normalisation (--without_norm): dy = dt * scale
    k1 = make_float4(0.f, 0.f, 0.f, 0.f);
    k2 = k1;
  }
  const int64_t base = (int64_t)n * a.P;
  const int64_t pstep = (int64_t)gridDim.x * a.rpi;
  int64_t pix = (int64_t)blockIdx.x * a.rpi + rl;
  if (G == 0 && !D && !L && !drop && UNETK_NORM_GROUPS > 1) {
    // plain units: NP adjacent row groups per pass, 2 NP loads in flight (see norm_apply_relu_kernel): 130.1 -> 116.8 us
    constexpr int NP = sizeof(T) == 2 ? UNETK_NORM_GROUPS_BF : UNETK_NORM_GROUPS;
    for (pix = (int64_t)blockIdx.x * NP * a.rpi + rl; pix < a.P; pix += NP * pstep) {
      float4 v[NP], d[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (pix + j * a.rpi < a.P) {
          v[j] = ld4(ay + (base + pix + j * a.rpi) * a.C + cq * 4);
          d[j] = ld4(adz + (base + pix + j * a.rpi) * a.dzs + cq * 4);
        }
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (pix + j * a.rpi < a.P) {
          float4 o;
#define NBA2(f)                                                                     \
  {                                                                                 \
    const float du = fmaf(v[j].f, sc.f, sh.f) > 0.f ? d[j].f : 0.f;                 \
    o.f = sc0.f * (du - k1.f - ((v[j].f - mu.f) * rs.f) * k2.f);                    \
  }
          NBA2(x) NBA2(y) NBA2(z) NBA2(w)
#undef NBA2
          st4(ady + (base + pix + j * a.rpi) * a.C + cq * 4, o);
        }
    }
    return;
  }
  for (; pix < a.P; pix += pstep) {
    const float4 v = ld4(ay + (base + pix) * a.C + cq * 4);
    const float4 d = ld4(adz + (base + pix) * a.dzs + cq * 4);
    float4 u, o;
    u.x = fmaf(v.x, sc.x, sh.x); u.y = fmaf(v.y, sc.y, sh.y); u.z = fmaf(v.z, sc.z, sh.z); u.w = fmaf(v.w, sc.w, sh.w);
    float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
    if (drop) {
      m = drop4(a.seed, (uint32_t)((base + pix) * a.C + cq * 4), a.keep, inv_keep);
      u.x *= m.x; u.y *= m.y; u.z *= m.z; u.w *= m.w;
      if (!L) u = add4(u, gbv);
    }
    if (L) {
      float4 sg = gbv;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float gg = a.guide[(base + pix) * G + g];
        sg.x = fmaf(gg, gwv[g].x, sg.x); sg.y = fmaf(gg, gwv[g].y, sg.y); sg.z = fmaf(gg, gwv[g].z, sg.z); sg.w = fmaf(gg, gwv[g].w, sg.w);
      }
      if (a.post) u = add4(u, post_act(sg, pq));
      else { u.x += lrelu(sg.x, a.alpha); u.y += lrelu(sg.y, a.alpha); u.z += lrelu(sg.z, a.alpha); u.w += lrelu(sg.w, a.alpha); }
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float gg = a.guide[(base + pix) * G + g];
        u.x = fmaf(gg, gwv[g].x, u.x); u.y = fmaf(gg, gwv[g].y, u.y); u.z = fmaf(gg, gwv[g].z, u.z); u.w = fmaf(gg, gwv[g].w, u.w);
      }
    }
#define NBA(f)                                          \
  {                                                     \
    const float du = u.f > 0.f ? d.f : 0.f;             \
    const float dt = (D ? du * dn.f : du) * m.f;        \
    const float xh = (v.f - mu.f) * rs.f;               \
    o.f = sc0.f * (dt - k1.f - xh * k2.f);              \
  }
    NBA(x) NBA(y) NBA(z) NBA(w)
#undef NBA
    st4(ady + (base + pix) * a.C + cq * 4, o);
  }
}

// parameter gradients from psum[k][c] = sum over launch groups of sums[k][l][c]
__global__ void norm_bwd_params_kernel(const float* __restrict__ psum, int C, int G, int density, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ dgw, float* __restrict__ dgb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = psum[c];
  if (dgamma) dgamma[c] = psum[C + c];
  if (dgb) dgb[c] = density ? psum[(int64_t)(2 + G) * C + c] : psum[c];   // density | leaky: the separate sum du [* lrelu']
  for (int g = 0; g < G; ++g) dgw[(int64_t)g * C + c] = psum[(int64_t)(2 + g) * C + c];
}

// per-sample guide weights (GUNet --fix under instance norm): dgw[n][g][c] / dgb[n][c] from sums[k][l = n][c]
__global__ void norm_bwd_guide_ps_kernel(const float* __restrict__ sums, int N, int C, int G, int kb, float* __restrict__ dgw,
                                         float* __restrict__ dgb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  if (dgb) dgb[i] = sums[((int64_t)kb * N + n) * C + c];
  for (int g = 0; g < G; ++g) dgw[((int64_t)n * G + g) * C + c] = sums[((int64_t)(2 + g) * N + n) * C + c];
}

// guide_leaky == 3: dgb block [groups][4][C] from src[k][groups][C] (the per-sample sums, or their total with groups = 1)
__global__ void norm_bwd_post_block_kernel(const float* __restrict__ src, int groups, int C, int G, float* __restrict__ dgb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * C) return;
  const int n = i / C, c = i - n * C;
  float* o = dgb + (int64_t)n * 4 * C + c;
  o[0] = src[((int64_t)(2 + G) * groups + n) * C + c];
  o[C] = 0.f;
  o[2 * C] = 0.f;
  o[3 * C] = src[((int64_t)(4 + G) * groups + n) * C + c];
}

// GUNet --use_se: the SE gate reads pooled[b][c] = mean over the sample's pixels of the normalised conv output, so the
// loss reaches y a second time through it: dt gets the per-(sample, channel) constant g[b][c] / HW.  The norm backward is
// linear in dt, so that part is added in place afterwards:  dy += scale * (A[b][c] - xhat * k2[c]) with
// A = g / HW - mean over the statistics group, k2 = mean of (g / HW) * xhat over the group (both tiny, from the host).
template <typename T>
__global__ __launch_bounds__(256) void norm_se_bwd_add_kernel(const T* __restrict__ y, T* __restrict__ dy,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ scale, const float* __restrict__ A,
                                                              const float* __restrict__ k2, int64_t P, int C, int cq_n, int rpi,
                                                              int sst) {
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  if (rl >= rpi) return;
  const int n = blockIdx.y;
  const int64_t so = (int64_t)n * sst + cq * 4;
  const float4 mu = ldg4(mean + so), rs = ldg4(rstd + so), sc = ldg4(scale + so);
  const float4 av = ldg4(A + (int64_t)n * C + cq * 4), kv = ldg4(k2 + so);
  const int64_t base = (int64_t)n * P;
  for (int64_t pix = (int64_t)blockIdx.x * rpi + rl; pix < P; pix += (int64_t)gridDim.x * rpi) {
    const float4 v = ld4(y + (base + pix) * C + cq * 4);
    float4 o = ld4(dy + (base + pix) * C + cq * 4);
    o.x += sc.x * (av.x - (v.x - mu.x) * rs.x * kv.x);
    o.y += sc.y * (av.y - (v.y - mu.y) * rs.y * kv.y);
    o.z += sc.z * (av.z - (v.z - mu.z) * rs.z * kv.z);
    o.w += sc.w * (av.w - (v.w - mu.w) * rs.w * kv.w);
    st4(dy + (base + pix) * C + cq * 4, o);
  }
}

// GUNet --use_se with --dropout (GUNet.py:189-201): the gate pools the DROPPED-OUT normalised output,
//   pooled[b][c] = mean_p m_p (gamma xhat_p + beta) = gamma * mean_p(m_p xhat_p) + beta * mean_p(m_p),
// which does not follow from the conv's statistic partials: sums[0][b][c] = sum_p m_p xhat_p, sums[1][b][c] = sum_p m_p, the mask
// regenerated from (seed, element index) as in the apply / backward passes.  One block per sample, fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void norm_drop_pool_kernel(const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ sums, int64_t P,
                                                             int C, int cq_n, int rpi, int sst, int N, float keep, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][rpi][C]
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  const int n = blockIdx.x;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (rl < rpi) {
    const int64_t so = (int64_t)n * sst + cq * 4;
    const float4 mu = ldg4(mean + so), rs = ldg4(rstd + so);
    const float inv_keep = 1.0f / keep;
    const int64_t base = (int64_t)n * P;
    for (int64_t pix = rl; pix < P; pix += rpi) {
      const float4 v = ld4(y + (base + pix) * C + cq * 4);
      const float4 m = drop4(seed, (uint32_t)((base + pix) * C + cq * 4), keep, inv_keep);
      s0.x += m.x * ((v.x - mu.x) * rs.x); s0.y += m.y * ((v.y - mu.y) * rs.y);
      s0.z += m.z * ((v.z - mu.z) * rs.z); s0.w += m.w * ((v.w - mu.w) * rs.w);
      s1.x += m.x; s1.y += m.y; s1.z += m.z; s1.w += m.w;
    }
    stg4(&smem[(0 * rpi + rl) * C + cq * 4], s0);
    stg4(&smem[(1 * rpi + rl) * C + cq * 4], s1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float t = 0.f;
    for (int j = 0; j < rpi; ++j) t += smem[(k * rpi + j) * C + c];
    sums[((int64_t)k * N + n) * C + c] = t;
  }
}

// ... and its way back: dt gets m_p * E[b][c] (E = d loss / d pooled / HW), and the norm backward being linear in dt,
//   dy += scale * (m_p E[b][c] - k1 - xhat k2),  k1 / k2 = the statistics group's means of m E and m E xhat (from the host:
// they follow from the two sums above).
template <typename T>
__global__ __launch_bounds__(256) void norm_se_bwd_add_drop_kernel(const T* __restrict__ y, T* __restrict__ dy,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ scale, const float* __restrict__ E,
                                                                   const float* __restrict__ k1, const float* __restrict__ k2,
                                                                   int64_t P, int C, int cq_n, int rpi, int sst, float keep,
                                                                   uint32_t seed) {
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  if (rl >= rpi) return;
  const int n = blockIdx.y;
  const int64_t so = (int64_t)n * sst + cq * 4;
  const float4 mu = ldg4(mean + so), rs = ldg4(rstd + so), sc = ldg4(scale + so);
  const float4 ev = ldg4(E + (int64_t)n * C + cq * 4), k1v = ldg4(k1 + so), k2v = ldg4(k2 + so);
  const float inv_keep = 1.0f / keep;
  const int64_t base = (int64_t)n * P;
  for (int64_t pix = (int64_t)blockIdx.x * rpi + rl; pix < P; pix += (int64_t)gridDim.x * rpi) {
    const float4 v = ld4(y + (base + pix) * C + cq * 4);
    const float4 m = drop4(seed, (uint32_t)((base + pix) * C + cq * 4), keep, inv_keep);
    float4 o = ld4(dy + (base + pix) * C + cq * 4);
    o.x += sc.x * (m.x * ev.x - k1v.x - (v.x - mu.x) * rs.x * k2v.x);
    o.y += sc.y * (m.y * ev.y - k1v.y - (v.y - mu.y) * rs.y * k2v.y);
    o.z += sc.z * (m.z * ev.z - k1v.z - (v.z - mu.z) * rs.z * k2v.z);
    o.w += sc.w * (m.w * ev.w - k1v.w - (v.w - mu.w) * rs.w * k2v.w);
    st4(dy + (base + pix) * C + cq * 4, o);
  }
}

bool norm_desc_ok(const unetk_norm_desc* d) {
  return d && d->N > 0 && d->HW > 0 && d->C > 0 && d->guide_ch >= 0 && d->guide_ch <= MAXG;
}
bool norm_supported(const unetk_norm_desc* d) { return d->C % 4 == 0 && d->C <= 1024; }

int bwd_blocks(const NormGeom& g) {
  int64_t b = (g.P + g.rpi - 1) / g.rpi;
  // per launch group; few groups (instance norm at batch 1-2: UNet3D's 96^3 tensors) need more blocks each to fill the GPU
  const int64_t cap = g.L > 1 ? (2048 / g.L > 64 ? 2048 / g.L : 64) : UNETK_COL_BLOCKS;
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
#define GL_DISPATCH(G_, CALL) \
  switch (G_) {              \
    case 1: { constexpr int GG = 1; CALL; } break; \
    case 2: { constexpr int GG = 2; CALL; } break; \
    case 3: { constexpr int GG = 3; CALL; } break; \
    default: { constexpr int GG = 4; CALL; } break; \
  }
#define GD_DISPATCH_T(TT, G_, D_, L_, KERN, ...)                                               \
  if (L_ && D_) { GL_DISPATCH(G_, UNETK_LAUNCH((KERN<GG, true, true, TT>), __VA_ARGS__)); } \
  else if (L_) { GL_DISPATCH(G_, UNETK_LAUNCH((KERN<GG, false, true, TT>), __VA_ARGS__)); }   \
  else if (D_) { G_DISPATCH(G_, UNETK_LAUNCH((KERN<GG, true, false, TT>), __VA_ARGS__)); } \
  else { G_DISPATCH(G_, UNETK_LAUNCH((KERN<GG, false, false, TT>), __VA_ARGS__)); }
// S_: tensors in HBM are bf16 (UNETK_BF16S) instead of fp32
#define GD_DISPATCH(S_, G_, D_, L_, KERN, ...)                      \
  if (S_) { GD_DISPATCH_T(bf16_t, G_, D_, L_, KERN, __VA_ARGS__) } \
  else { GD_DISPATCH_T(float, G_, D_, L_, KERN, __VA_ARGS__) }

// ---- plain unit whose activation feeds a 2 x 2 max-pool AND the skip connection (UNet.py:80-81,93: every encoder level's
// second conv): the pool's backward is folded into both passes.  dz of pixel i of a window = dskip_i + (i is the window's
// first maximum of z ? dp : 0) (TF MaxPoolGrad's rule, pool.hip), with z = relu(y * scale + shift) re-evaluated from y by the
// forward's own expression and rounded as the forward stored it; the sum is rounded to the storage type as the separate
// pool-backward pass stored it.  Saves that pass: 3.25 tensor transits of the 8.25 the three passes moved.
struct PoolArgs {
  const void* dp;       // [N][H/2][W/2][C], dense
  int H, W;
};
template <typename T>
__device__ __forceinline__ float round_as(float v);
template <>
__device__ __forceinline__ float round_as<float>(float v) { return v; }
template <>
__device__ __forceinline__ float round_as<bf16_t>(float v) { return unetk_round_bf16(v); }

// window q of launch group n -> its four pixels' y, routed dz (d) and masked gradient du; f(pixel, y, du) per pixel
template <typename T, class F>
__device__ __forceinline__ void pool_window(const BwdArgs& a, const PoolArgs& pa, int n, int64_t q, int cq, const float4& sc,
                                            const float4& sh, F&& f) {
  const int Wo = pa.W >> 1;
  const int64_t per_img = (int64_t)(pa.H >> 1) * Wo;
  const int64_t img = a.L > 1 ? n : q / per_img;
  const int64_t r = a.L > 1 ? q : q - img * per_img;
  const int ho = (int)(r / Wo), wo = (int)(r - (int64_t)ho * Wo);
  const int64_t pix = (img * pa.H + 2 * ho) * pa.W + 2 * wo;               // global pixel index of the window's corner
  const int64_t off[4] = {0, 1, pa.W, (int64_t)pa.W + 1};
  const T* ay = static_cast<const T*>(a.y);
  const T* ask = static_cast<const T*>(a.dz);
  float4 v[4], d[4], u[4], z[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = ld4(ay + (pix + off[i]) * a.C + cq * 4);
    d[i] = ld4(ask + (pix + off[i]) * a.dzs + cq * 4);
  }
  const float4 g = ld4(static_cast<const T*>(pa.dp) + ((img * per_img + r) * a.C + cq * 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i].x = fmaf(v[i].x, sc.x, sh.x); u[i].y = fmaf(v[i].y, sc.y, sh.y); u[i].z = fmaf(v[i].z, sc.z, sh.z); u[i].w = fmaf(v[i].w, sc.w, sh.w);
    z[i].x = round_as<T>(fmaxf(u[i].x, 0.f)); z[i].y = round_as<T>(fmaxf(u[i].y, 0.f));
    z[i].z = round_as<T>(fmaxf(u[i].z, 0.f)); z[i].w = round_as<T>(fmaxf(u[i].w, 0.f));
  }
#define PWIN(c)                                                                   \
  {                                                                               \
    const float m = fmaxf(fmaxf(z[0].c, z[1].c), fmaxf(z[2].c, z[3].c));          \
    const bool e0 = z[0].c == m;                                                  \
    const bool e1 = !e0 && z[1].c == m;                                           \
    const bool e2 = !e0 && !e1 && z[2].c == m;                                    \
    const bool e3 = !e0 && !e1 && !e2;                                            \
    d[0].c = round_as<T>(d[0].c + (e0 ? g.c : 0.f));                              \
    d[1].c = round_as<T>(d[1].c + (e1 ? g.c : 0.f));                              \
    d[2].c = round_as<T>(d[2].c + (e2 ? g.c : 0.f));                              \
    d[3].c = round_as<T>(d[3].c + (e3 ? g.c : 0.f));                              \
  }
  PWIN(x) PWIN(y) PWIN(z) PWIN(w)
#undef PWIN
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 du;
    du.x = u[i].x > 0.f ? d[i].x : 0.f; du.y = u[i].y > 0.f ? d[i].y : 0.f;
    du.z = u[i].z > 0.f ? d[i].z : 0.f; du.w = u[i].w > 0.f ? d[i].w : 0.f;
    f(pix + off[i], v[i], du);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_reduce_pool_kernel(BwdArgs a, PoolArgs pa) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][rpi][C]
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  const int n = blockIdx.y;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (rl < a.rpi) {
    const int64_t so = (int64_t)n * a.sst + cq * 4;
    const float4 mu = ldg4(a.mean + so), rs = ldg4(a.rstd + so), sc = ldg4(a.scale + so), sh = ldg4(a.shift + so);
    const int64_t Q = a.P >> 2;                                    // windows of this launch group
    for (int64_t q = (int64_t)blockIdx.x * a.rpi + rl; q < Q; q += (int64_t)gridDim.x * a.rpi)
      pool_window<T>(a, pa, n, q, cq, sc, sh, [&](int64_t, const float4& v, const float4& du) {
        s0.x += du.x; s0.y += du.y; s0.z += du.z; s0.w += du.w;
        s1.x += du.x * ((v.x - mu.x) * rs.x); s1.y += du.y * ((v.y - mu.y) * rs.y);
        s1.z += du.z * ((v.z - mu.z) * rs.z); s1.w += du.w * ((v.w - mu.w) * rs.w);
      });
    stg4(&smem[(0 * a.rpi + rl) * a.C + cq * 4], s0);
    stg4(&smem[(1 * a.rpi + rl) * a.C + cq * 4], s1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
    const int k = i / a.C, c = i - k * a.C;
    float t = 0.f;
    for (int j = 0; j < a.rpi; ++j) t += smem[(k * a.rpi + j) * a.C + c];
    a.partial[(((int64_t)k * a.L + n) * gridDim.x + blockIdx.x) * a.C + c] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_apply_pool_kernel(BwdArgs a, PoolArgs pa) {
  const int cq = threadIdx.x % a.cq_n, rl = threadIdx.x / a.cq_n;
  if (rl >= a.rpi) return;
  T* ady = static_cast<T*>(a.dy);
  const int n = blockIdx.y;
  const int64_t so = (int64_t)n * a.sst + cq * 4;
  const float4 mu = ldg4(a.mean + so), rs = ldg4(a.rstd + so), sc = ldg4(a.scale + so), sh = ldg4(a.shift + so);
  float4 k1 = ldg4(a.ksum + (int64_t)n * a.kst + cq * 4), k2 = ldg4(a.ksum + a.krow + (int64_t)n * a.kst + cq * 4);
  k1.x *= a.inv_ps; k1.y *= a.inv_ps; k1.z *= a.inv_ps; k1.w *= a.inv_ps;
  k2.x *= a.inv_ps; k2.y *= a.inv_ps; k2.z *= a.inv_ps; k2.w *= a.inv_ps;
  if (a.plain) {
    k1 = make_float4(0.f, 0.f, 0.f, 0.f);
    k2 = k1;
  }
  const int64_t Q = a.P >> 2;
  for (int64_t q = (int64_t)blockIdx.x * a.rpi + rl; q < Q; q += (int64_t)gridDim.x * a.rpi)
    pool_window<T>(a, pa, n, q, cq, sc, sh, [&](int64_t pix, const float4& v, const float4& du) {
      float4 o;
      o.x = sc.x * (du.x - k1.x - ((v.x - mu.x) * rs.x) * k2.x);
      o.y = sc.y * (du.y - k1.y - ((v.y - mu.y) * rs.y) * k2.y);
      o.z = sc.z * (du.z - k1.z - ((v.z - mu.z) * rs.z) * k2.z);
      o.w = sc.w * (du.w - k1.w - ((v.w - mu.w) * rs.w) * k2.w);
      st4(ady + pix * a.C + cq * 4, o);
    });
}

}  // namespace

extern "C" size_t unetk_norm_finalize_ws_bytes(const unetk_norm_desc* d, int stat_rows) {
  if (!norm_desc_ok(d) || stat_rows <= 0) return 0;
  const NormGeom g = geom(d, false);
  const int rows_per_group = stat_rows / g.Ns;
  return (2 * (size_t)g.Ns * d->C + unetk_rows_reduce_tmp_floats(2 * g.Ns, rows_per_group, d->C)) * sizeof(float);
}

extern "C" int unetk_norm_finalize(const unetk_norm_desc* d, const float* stat_partials, int stat_rows,
                                   const float* gamma, const float* beta, float eps, float decay, int training,
                                   float* moving_mean, float* moving_var, float* mean_out, float* rstd_out,
                                   float* scale_out, float* shift_out, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && mean_out && rstd_out && scale_out && shift_out);
  hipStream_t st = (hipStream_t)stream;
  const NormGeom g = geom(d, false);
  const int use_moving = (!d->per_sample && !training) ? 1 : 0;
  float* sums = nullptr;
  if (!use_moving) {
    UNETK_REQUIRE(stat_partials && stat_rows > 0 && ws && stat_rows % g.Ns == 0);
    if (ws_bytes < unetk_norm_finalize_ws_bytes(d, stat_rows)) return UNETK_E_WORKSPACE;
    sums = (float*)ws;
    // partials are [2][stat_rows][C] with each image's tiles contiguous -> [2*Ns][rows_per_group][C]
    const float* src = stat_partials;
    int rows = stat_rows / g.Ns;
    if (rows > UNETK_RR_DIRECT_ROWS) {   // first level: 64 row blocks per (statistic, group)
      int rc = unetk_rows_reduce_l1(stat_partials, 2 * g.Ns, rows, d->C, sums + 2 * g.Ns * d->C, st);
      if (rc != UNETK_OK) return rc;
      src = sums + 2 * g.Ns * d->C;
      rows = 64;
    }
    const int update_moving = (!d->per_sample && training && moving_mean && moving_var) ? 1 : 0;
    UNETK_LAUNCH(norm_reduce_finalize_kernel, dim3(((d->C + 15) / 16) * g.Ns), dim3(256), 0, st, src, rows, g.Ns, d->C,
                       (double)g.Ps, gamma, beta, eps, decay, update_moving, moving_mean, moving_var, mean_out, rstd_out,
                       scale_out, shift_out);
    UNETK_LAUNCH_CHECK();
    return UNETK_OK;
  } else {
    UNETK_REQUIRE(moving_mean && moving_var);
  }
  const int update_moving = (!d->per_sample && training && moving_mean && moving_var) ? 1 : 0;
  const int total = g.Ns * d->C;
  UNETK_LAUNCH(norm_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, st, sums, g.Ns, d->C, (double)g.Ps,
                     gamma, beta, eps, decay, use_moving, update_moving, moving_mean, moving_var, mean_out, rstd_out,
                     scale_out, shift_out);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_norm_apply_relu(const unetk_norm_desc* d, const void* y, const float* scale, const float* shift,
                                     const float* den, const float* guide, const float* gw, const float* gb, void* z,
                                     void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && scale && shift && z && d->z_stride >= d->C);
  if (!norm_supported(d) || d->z_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? (unetk_aligned8(y) && unetk_aligned8(z)) : (unetk_aligned16(y) && unetk_aligned16(z)));
  UNETK_REQUIRE(unetk_aligned16(scale) && unetk_aligned16(shift));
  UNETK_REQUIRE(!den || unetk_aligned16(den));
  if (d->guide_ch > 0) {
    UNETK_REQUIRE(guide && gw && gb && d->gw_stride >= d->gw_coff + d->C);
    UNETK_REQUIRE(d->gw_stride % 4 == 0 && unetk_aligned16(gw));
  }
  UNETK_REQUIRE(!gb || (d->gw_coff % 4 == 0 && d->gw_coff >= 0 && unetk_aligned16(gb)));
  const NormGeom g = geom(d, den != nullptr);
  if (d->dropout_keep < 0.f || d->dropout_keep > 1.f) return UNETK_E_BADARG;
  if (d->guide_per_sample && g.L != d->N) return UNETK_E_UNSUPPORTED;      // needs one launch group per sample
  ApplyArgs a{y, scale, shift, den, guide, gw, gb, z, g.P, d->C, d->z_stride, g.cq_n, g.rpi, d->gw_stride, d->gw_coff, g.sst,
              d->guide_per_sample ? d->guide_ch * d->gw_stride : 0,
              d->guide_per_sample ? (d->guide_leaky == 3 ? 4 : 1) * d->gw_stride : 0, d->guide_leaky == 3 ? 1 : 0,
              d->guide_leaky == 2 ? d->guide_alpha : 0.2f,
              d->dropout_keep, d->dropout_seed};
  int64_t gx = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = g.L > 1 ? (4096 + g.L - 1) / g.L : 4096;
  if (gx > cap) gx = cap;
  const bool leaky = d->guide_leaky != 0;
  if (leaky && d->guide_ch < 1) return UNETK_E_UNSUPPORTED;
  if (d->guide_leaky == 3 && (den == nullptr || gb == nullptr)) return UNETK_E_UNSUPPORTED;     // post vectors: with density gains only
  GD_DISPATCH(bs, d->guide_ch, den != nullptr, leaky, norm_apply_relu_kernel, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, a);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// unetk_norm_apply_relu of a plain unit (no guide / density / dropout / post-shift) + unetk_maxpool2_fwd of its activation in one
// pass: z as there (pixel stride d->z_stride), pooled [N, H/2, W/2, C] dense.  d->HW = H * W, W given here, both even.
extern "C" int unetk_norm_apply_relu_pool(const unetk_norm_desc* d, int W, const void* y, const float* scale, const float* shift,
                                          void* z, void* pooled, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && scale && shift && z && pooled && d->z_stride >= d->C);
  if (!norm_supported(d) || d->z_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  if (d->guide_ch > 0 || d->guide_leaky || d->dropout_keep > 0.f) return UNETK_E_UNSUPPORTED;
  if (W < 2 || (W & 1) || d->HW % W != 0 || ((d->HW / W) & 1)) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? (unetk_aligned8(y) && unetk_aligned8(z) && unetk_aligned8(pooled))
                   : (unetk_aligned16(y) && unetk_aligned16(z) && unetk_aligned16(pooled)));
  UNETK_REQUIRE(unetk_aligned16(scale) && unetk_aligned16(shift));
  const NormGeom g = geom(d, false);
  ApplyArgs a{};
  a.y = y; a.scale = scale; a.shift = shift; a.z = z; a.P = g.P; a.C = d->C; a.zs = d->z_stride; a.cq_n = g.cq_n; a.rpi = g.rpi;
  a.sst = g.sst;
  int64_t gx = ((g.P >> 2) + g.rpi - 1) / g.rpi;
  const int64_t cap = g.L > 1 ? (4096 + g.L - 1) / g.L : 4096;
  if (gx > cap) gx = cap;
  if (bs) UNETK_LAUNCH(norm_apply_relu_pool_kernel<bf16_t>, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, a, pooled,
                             d->HW / W, W, g.L);
  else UNETK_LAUNCH(norm_apply_relu_pool_kernel<float>, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, a, pooled,
                          d->HW / W, W, g.L);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" size_t unetk_norm_bwd_ws_bytes(const unetk_norm_desc* d) {
  if (!norm_desc_ok(d) || !norm_supported(d)) return 0;
  // sized for the density variant (K + 2 rows, N launch groups): a superset of every other case
  const NormGeom g = geom(d, true);
  const int K = 5 + d->guide_ch;          // density + leaky guide + post-shift row: the largest variant
  NormGeom g1 = geom(d, false);
  int nblk = bwd_blocks(g);
  const int nblk1 = bwd_blocks(g1);
  size_t f = (size_t)K * ((size_t)g.L * nblk > (size_t)g1.L * nblk1 ? (size_t)g.L * nblk : (size_t)g1.L * nblk1) * d->C;
  f += (size_t)K * g.L * d->C;                                       // sums per launch group
  f += (size_t)K * d->C;                                             // sums over groups
  if (nblk1 > nblk) nblk = nblk1;
  f += unetk_rows_reduce_tmp_floats(K * g.L, UNETK_RR_DIRECT_ROWS + 1, d->C);              // always: pre-computed partials may have any row count
  f += unetk_rows_reduce_tmp_floats(K, g.L, d->C);
  return f * sizeof(float);
}

extern "C" int unetk_norm_relu_bwd(const unetk_norm_desc* d, const void* y, const void* dz, int dz_stride,
                                   const float* scale, const float* shift, const float* mean, const float* rstd,
                                   const float* den, const float* guide, const float* gw, const float* gb, void* dy,
                                   float* dgamma, float* dbeta, float* dden, float* dgw, float* dgb, void* ws,
                                   size_t ws_bytes, void* stream) {
  return unetk_norm_relu_bwd_pre(d, y, dz, dz_stride, scale, shift, mean, rstd, den, guide, gw, gb, dy, dgamma, dbeta, dden,
                                 dgw, dgb, nullptr, 0, ws, ws_bytes, stream);
}

// pre_partials [2][pre_rows][C] (nullable): the reduction pass's sums (sum du, sum du * xhat), per tile of the kernel that
// PRODUCED dz (unetk_conv3x3_dgrad_nbr) -- then only the apply pass runs here.  Plain units only (no guide, density,
// dropout, bias-only).
extern "C" int unetk_norm_relu_bwd_pre(const unetk_norm_desc* d, const void* y, const void* dz, int dz_stride,
                                       const float* scale, const float* shift, const float* mean, const float* rstd,
                                       const float* den, const float* guide, const float* gw, const float* gb, void* dy,
                                       float* dgamma, float* dbeta, float* dden, float* dgw, float* dgb,
                                       const float* pre_partials, int pre_rows, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && dz && scale && shift && mean && rstd && dy && ws && dz_stride >= d->C);
  if (!norm_supported(d) || dz_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? (unetk_aligned8(y) && unetk_aligned8(dz) && unetk_aligned8(dy))
                   : (unetk_aligned16(y) && unetk_aligned16(dz) && unetk_aligned16(dy)));
  UNETK_REQUIRE(unetk_aligned16(ws));
  const int G = d->guide_ch;
  const bool D = den != nullptr;
  UNETK_REQUIRE(!D || (dden && unetk_aligned16(den)));
  if (G > 0) {
    UNETK_REQUIRE(guide && gw && gb && dgw && d->gw_stride >= d->gw_coff + d->C);
    UNETK_REQUIRE(d->gw_stride % 4 == 0 && unetk_aligned16(gw));
  }
  UNETK_REQUIRE(!gb || (dgb && d->gw_coff % 4 == 0 && d->gw_coff >= 0 && unetk_aligned16(gb)));
  if (ws_bytes < unetk_norm_bwd_ws_bytes(d)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const NormGeom g = geom(d, D);
  const bool leaky = d->guide_leaky != 0;
  if (leaky && G < 1) return UNETK_E_UNSUPPORTED;
  const bool post = d->guide_leaky == 3;     // dgb is then the gradient of the whole gb block: [4][C] ([N][4][C] per sample)
  if (post && (!D || !gb || pre_partials != nullptr)) return UNETK_E_UNSUPPORTED;
  const int K = 2 + G + (D ? 2 : (leaky ? 1 : 0)) + (post ? 1 : 0);
  const int nblk = bwd_blocks(g);
  float* partial = (float*)ws;
  float* sums = partial + (size_t)K * g.L * nblk * d->C;
  float* psum = sums + (size_t)K * g.L * d->C;
  float* tmp1 = psum + (size_t)K * d->C;
  float* tmp2 = tmp1 + unetk_rows_reduce_tmp_floats(K * g.L, UNETK_RR_DIRECT_ROWS + 1, d->C);
  BwdArgs a{};
  a.y = y; a.dz = dz; a.scale = scale; a.shift = shift; a.mean = mean; a.rstd = rstd; a.den = den;
  a.guide = guide; a.gw = gw; a.gb = gb; a.partial = partial; a.dy = dy;
  a.P = g.P; a.inv_ps = 1.0f / (float)g.Ps; a.C = d->C; a.dzs = dz_stride; a.cq_n = g.cq_n; a.rpi = g.rpi;
  a.gw_stride = d->gw_stride; a.gw_coff = d->gw_coff; a.L = g.L; a.plain = d->affine_only; a.sst = g.sst;
  if (d->dropout_keep < 0.f || d->dropout_keep > 1.f) return UNETK_E_BADARG;
  if (d->dropout_keep > 0.f && (gb || G > 0) && !D && !leaky) return UNETK_E_UNSUPPORTED;   // the guide-bias sum needs the density variant (pass den = 1)
  if (d->guide_per_sample && g.L != d->N) return UNETK_E_UNSUPPORTED;
  a.gw_ns = d->guide_per_sample ? G * d->gw_stride : 0; a.gb_ns = d->guide_per_sample ? (post ? 4 : 1) * d->gw_stride : 0;
  a.alpha = d->guide_leaky == 2 ? d->guide_alpha : 0.2f; a.keep = d->dropout_keep; a.seed = d->dropout_seed;
  a.post = post ? 1 : 0;
  // the statistics sums of the dy formula: per launch group when the statistics are per sample, else the batch totals
  if (d->per_sample) { a.ksum = sums; a.kst = d->C; a.krow = g.L * d->C; }
  else if (g.L == 1) { a.ksum = sums; a.kst = 0; a.krow = d->C; }
  else { a.ksum = psum; a.kst = 0; a.krow = d->C; }
  const size_t lds = (size_t)K * g.rpi * d->C * sizeof(float);
  int rc;
  // plain unit (K = 2: sum du, sum du * xhat): d beta / d gamma ARE the two totals, written by the last reduction launch
  const bool simple = G == 0 && !D && !leaky && dgb == nullptr;
  float* const al0 = simple ? dbeta : nullptr;
  float* const al1 = simple ? dgamma : nullptr;
  const bool last1 = g.L == 1;
  if (pre_partials != nullptr) {
    if (G > 0 || D || leaky || gb || d->dropout_keep > 0.f || d->affine_only) return UNETK_E_UNSUPPORTED;
    UNETK_REQUIRE(pre_rows > 0 && pre_rows % g.L == 0);
    rc = unetk_rows_reduce_alias(pre_partials, K * g.L, pre_rows / g.L, d->C, sums, tmp1, last1 ? al0 : nullptr,
                                 last1 ? al1 : nullptr, st);   // -> sums[K][L][C]
  } else {
    if (post) {
      if (bs) { GL_DISPATCH(G, UNETK_LAUNCH((norm_bwd_reduce_kernel<GG, true, true, bf16_t, true>), dim3(nblk, g.L), dim3(256), lds, st, a)); }
      else { GL_DISPATCH(G, UNETK_LAUNCH((norm_bwd_reduce_kernel<GG, true, true, float, true>), dim3(nblk, g.L), dim3(256), lds, st, a)); }
    } else {
      GD_DISPATCH(bs, G, D, leaky, norm_bwd_reduce_kernel, dim3(nblk, g.L), dim3(256), lds, st, a);
    }
    UNETK_LAUNCH_CHECK();
    rc = unetk_rows_reduce_alias(partial, K * g.L, nblk, d->C, sums, tmp1, last1 ? al0 : nullptr, last1 ? al1 : nullptr,
                                 st);                          // -> sums[K][L][C]
  }
  if (rc != UNETK_OK) return rc;
  if (g.L == 1) {
    psum = sums;                                                               // one launch group: nothing to add up
  } else {
    rc = unetk_rows_reduce_alias(sums, K, g.L, d->C, psum, tmp2, al0, al1, st);   // -> psum[K][C]
    if (rc != UNETK_OK) return rc;
  }
  const bool gps = d->guide_per_sample != 0;
  if (!simple) {
    UNETK_LAUNCH(norm_bwd_params_kernel, dim3((d->C + 255) / 256), dim3(256), 0, st, psum, d->C, gps ? 0 : G,
                       (D || leaky) ? 1 : 0, dgamma, dbeta, gps ? nullptr : dgw, (gps || post) ? nullptr : dgb);
    UNETK_LAUNCH_CHECK();
  }
  if (gps) {   // dgw [N][G][C], dgb [N][C]: the per-launch-group sums, not their total
    const int kb = (D || leaky) ? 2 + G : 0;
    UNETK_LAUNCH(norm_bwd_guide_ps_kernel, dim3((d->N * d->C + 255) / 256), dim3(256), 0, st, sums, d->N, d->C, G, kb, dgw,
                       post ? nullptr : dgb);
    UNETK_LAUNCH_CHECK();
  }
  if (post) {  // gradient of the gb block: bias row = sum dg, slope rows = 0 (constants of the fold's sign), post-shift row = sum du
    const int groups = gps ? d->N : 1;
    UNETK_LAUNCH(norm_bwd_post_block_kernel, dim3((groups * d->C + 255) / 256), dim3(256), 0, st, gps ? sums : psum, groups,
                       d->C, G, dgb);
    UNETK_LAUNCH_CHECK();
  }
  if (D) {   // density gradient: the per-sample row sum du * t
    hipError_t e = hipMemcpyAsync(dden, sums + (size_t)(3 + G) * g.L * d->C, (size_t)d->N * d->C * sizeof(float),
                                  hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  int64_t gx = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = g.L > 1 ? (4096 + g.L - 1) / g.L : 4096;
  if (gx > cap) gx = cap;
  GD_DISPATCH(bs, G, D, leaky, norm_bwd_apply_kernel, dim3((int)gx, g.L), dim3(256), 0, st, a);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// The backward of a plain unit (no guide / density / dropout) whose output z feeds max_pool2d AND the skip connection:
// dz is never materialised.  dskip = the skip's gradient (pixel stride dskip_stride: a channel slice of the concat
// buffer's gradient), dp = the pooled tensor's gradient [N, H/2, W/2, C]; d->HW = H * W with W given here.
extern "C" int unetk_norm_relu_bwd_pool(const unetk_norm_desc* d, int W, const void* y, const void* dskip, int dskip_stride,
                                        const void* dp, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, void* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                        void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && dskip && dp && scale && shift && mean && rstd && dy && ws && dskip_stride >= d->C);
  if (!norm_supported(d) || dskip_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  if (d->guide_ch > 0 || d->guide_leaky || d->dropout_keep > 0.f) return UNETK_E_UNSUPPORTED;
  if (W < 2 || (W & 1) || d->HW % W != 0 || ((d->HW / W) & 1)) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? (unetk_aligned8(y) && unetk_aligned8(dskip) && unetk_aligned8(dp) && unetk_aligned8(dy))
                   : (unetk_aligned16(y) && unetk_aligned16(dskip) && unetk_aligned16(dp) && unetk_aligned16(dy)));
  UNETK_REQUIRE(unetk_aligned16(ws));
  if (ws_bytes < unetk_norm_bwd_ws_bytes(d)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const NormGeom g = geom(d, false);
  const int K = 2;
  const int nblk = bwd_blocks(g);
  float* partial = (float*)ws;
  float* sums = partial + (size_t)K * g.L * nblk * d->C;
  float* psum = sums + (size_t)K * g.L * d->C;
  float* tmp1 = psum + (size_t)K * d->C;
  float* tmp2 = tmp1 + unetk_rows_reduce_tmp_floats(K * g.L, UNETK_RR_DIRECT_ROWS + 1, d->C);
  BwdArgs a{};
  a.y = y; a.dz = dskip; a.scale = scale; a.shift = shift; a.mean = mean; a.rstd = rstd; a.partial = partial; a.dy = dy;
  a.P = g.P; a.inv_ps = 1.0f / (float)g.Ps; a.C = d->C; a.dzs = dskip_stride; a.cq_n = g.cq_n; a.rpi = g.rpi;
  a.L = g.L; a.plain = d->affine_only; a.sst = g.sst;
  if (d->per_sample) { a.ksum = sums; a.kst = d->C; a.krow = g.L * d->C; }
  else { a.ksum = sums; a.kst = 0; a.krow = d->C; }                 // batch statistics: one launch group
  PoolArgs pa{dp, d->HW / W, W};
  const size_t lds = (size_t)K * g.rpi * d->C * sizeof(float);
  const bool last1 = g.L == 1;
  if (bs) UNETK_LAUNCH(norm_bwd_reduce_pool_kernel<bf16_t>, dim3(nblk, g.L), dim3(256), lds, st, a, pa);
  else UNETK_LAUNCH(norm_bwd_reduce_pool_kernel<float>, dim3(nblk, g.L), dim3(256), lds, st, a, pa);
  UNETK_LAUNCH_CHECK();
  int rc = unetk_rows_reduce_alias(partial, K * g.L, nblk, d->C, sums, tmp1, last1 ? dbeta : nullptr, last1 ? dgamma : nullptr, st);
  if (rc != UNETK_OK) return rc;
  if (!last1) {
    rc = unetk_rows_reduce_alias(sums, K, g.L, d->C, psum, tmp2, dbeta, dgamma, st);
    if (rc != UNETK_OK) return rc;
  }
  int64_t gx = ((g.P >> 2) + g.rpi - 1) / g.rpi;
  const int64_t cap = g.L > 1 ? (4096 + g.L - 1) / g.L : 4096;
  if (gx > cap) gx = cap;
  if (bs) UNETK_LAUNCH(norm_bwd_apply_pool_kernel<bf16_t>, dim3((int)gx, g.L), dim3(256), 0, st, a, pa);
  else UNETK_LAUNCH(norm_bwd_apply_pool_kernel<float>, dim3((int)gx, g.L), dim3(256), 0, st, a, pa);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// --use_se with --dropout, forward: sums [2][N][C] = per (sample, channel) sum of m * xhat and of m (see norm_drop_pool_kernel);
// d->dropout_keep / dropout_seed as in unetk_norm_apply_relu.  mean / rstd [groups][C] from unetk_norm_finalize.
extern "C" int unetk_norm_drop_pool(const unetk_norm_desc* d, const void* y, const float* mean, const float* rstd, float* sums,
                                    void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && mean && rstd && sums);
  if (!norm_supported(d) || d->affine_only) return UNETK_E_UNSUPPORTED;
  if (!(d->dropout_keep > 0.f && d->dropout_keep <= 1.f)) return UNETK_E_BADARG;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? unetk_aligned8(y) : unetk_aligned16(y));
  const NormGeom g = geom(d, true);            // one launch group per sample
  const size_t lds = (size_t)2 * g.rpi * d->C * sizeof(float);
  if (bs)
    UNETK_LAUNCH(norm_drop_pool_kernel<bf16_t>, dim3(d->N), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)y, mean, rstd,
                       sums, g.P, d->C, g.cq_n, g.rpi, g.sst, d->N, d->dropout_keep, d->dropout_seed);
  else
    UNETK_LAUNCH(norm_drop_pool_kernel<float>, dim3(d->N), dim3(256), lds, (hipStream_t)stream, (const float*)y, mean, rstd,
                       sums, g.P, d->C, g.cq_n, g.rpi, g.sst, d->N, d->dropout_keep, d->dropout_seed);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// ... backward: dy += scale * (m E[b][c] - k1 - xhat k2); E [N][C], k1 / k2 [groups][C] (groups = N under instance norm, else 1).
extern "C" int unetk_norm_se_bwd_add_drop(const unetk_norm_desc* d, const void* y, void* dy, const float* mean, const float* rstd,
                                          const float* scale, const float* E, const float* k1, const float* k2, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && dy && mean && rstd && scale && E && k1 && k2);
  if (!norm_supported(d)) return UNETK_E_UNSUPPORTED;
  if (!(d->dropout_keep > 0.f && d->dropout_keep <= 1.f)) return UNETK_E_BADARG;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? (unetk_aligned8(y) && unetk_aligned8(dy)) : (unetk_aligned16(y) && unetk_aligned16(dy)));
  UNETK_REQUIRE(unetk_aligned16(E) && unetk_aligned16(k1) && unetk_aligned16(k2) && unetk_aligned16(mean) && unetk_aligned16(rstd) &&
                unetk_aligned16(scale));
  NormGeom g = geom(d, true);
  int64_t gx = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = (4096 + g.L - 1) / g.L;
  if (gx > cap) gx = cap;
  if (bs)
    UNETK_LAUNCH(norm_se_bwd_add_drop_kernel<bf16_t>, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y,
                       (bf16_t*)dy, mean, rstd, scale, E, k1, k2, g.P, d->C, g.cq_n, g.rpi, g.sst, d->dropout_keep, d->dropout_seed);
  else
    UNETK_LAUNCH(norm_se_bwd_add_drop_kernel<float>, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, (const float*)y,
                       (float*)dy, mean, rstd, scale, E, k1, k2, g.P, d->C, g.cq_n, g.rpi, g.sst, d->dropout_keep, d->dropout_seed);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_norm_se_bwd_add(const unetk_norm_desc* d, const void* y, void* dy, const float* mean, const float* rstd,
                                     const float* scale, const float* A, const float* k2, void* stream) {
  UNETK_REQUIRE(norm_desc_ok(d) && y && dy && mean && rstd && scale && A && k2);
  if (!norm_supported(d)) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE(d->storage == UNETK_FP32 || bs);
  UNETK_REQUIRE(bs ? (unetk_aligned8(y) && unetk_aligned8(dy)) : (unetk_aligned16(y) && unetk_aligned16(dy)));
  UNETK_REQUIRE(unetk_aligned16(A) && unetk_aligned16(k2) && unetk_aligned16(mean) && unetk_aligned16(rstd) && unetk_aligned16(scale));
  NormGeom g = geom(d, true);            // one launch group per sample: A is per (sample, channel)
  int64_t gx = (g.P + g.rpi - 1) / g.rpi;
  const int64_t cap = (4096 + g.L - 1) / g.L;
  if (gx > cap) gx = cap;
  if (bs)
    UNETK_LAUNCH(norm_se_bwd_add_kernel<bf16_t>, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y,
                       (bf16_t*)dy, mean, rstd, scale, A, k2, g.P, d->C, g.cq_n, g.rpi, g.sst);
  else
    UNETK_LAUNCH(norm_se_bwd_add_kernel<float>, dim3((int)gx, g.L), dim3(256), 0, (hipStream_t)stream, (const float*)y,
                       (float*)dy, mean, rstd, scale, A, k2, g.P, d->C, g.cq_n, g.rpi, g.sst);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
