// Logits (1x1 conv + bias) fused with the loss head and the in-graph metrics (HBM-bound: one read of the
// last feature map forward, one read + one write backward).
//
// Replaces, in ONE pass each way:
//   slim.conv2d(x, num_classes, 1, activation_fn=None, normalizer_fn=None)     NetworksV2/UNet.py:97-101
//   slim.softmax + Pred = prob > 0.5                                            UNet.py:104-118
//   _compute_weights / weighted_sparse_softmax_cross_entropy                    loss_metrics.py:115-177
//   sparse_dice_loss                                                            loss_metrics.py:180-226
//   metric_dice / metric_voe / metric_vd (their per-sample sums)                loss_metrics.py:261-339
// Thread mapping: C/4 lanes per pixel (float4 each, coalesced), xor-shuffle reduction of the class dot
// products; per-block partial sums in fixed order (bit-reproducible).
#include "common.h"

namespace {

constexpr int MAXC = UNETK_MAX_CLASSES;

struct HeadWs {        // layout of the caller-provided workspace (floats)
  int hist_off;        // int32 [N][ncls]
  int wn_off;          // float [N][ncls]   per-sample, per-class normalised weight
  int part_off;        // float [N][BPS][NQ]
  int bps, nq;
  int bwd_nblk;
  int64_t pw_off, pb_off, tmp_off, total;
};

inline int head_nq(int ncls) { return 2 + (ncls - 1) * 6; }

HeadWs head_ws(const unetk_head_desc* d) {
  HeadWs w{};
  const int N = d->N, ncls = d->ncls;
  int bps = (d->HW + 1023) / 1024;      // streaming, latency-bound kernels: enough blocks for ~8 waves per SIMD
  const int cap = N >= 32 ? 64 : (2048 + N - 1) / N;      // ~2048 blocks in all (a batch of 8 got only 512 with a flat cap of 64)
  if (bps > cap) bps = cap;
  if (bps < 1) bps = 1;
  w.bps = bps;
  w.nq = head_nq(ncls);
  int64_t off = 0;
  w.hist_off = (int)off; off += (int64_t)N * ncls;
  w.wn_off = (int)off; off += (int64_t)N * ncls;
  off = (off + 3) & ~(int64_t)3;
  w.part_off = (int)off; off += (int64_t)N * bps * w.nq;
  off = (off + 3) & ~(int64_t)3;
  int64_t npix = (int64_t)N * d->HW;
  int64_t nb = (npix + 255) / 256;
  if (nb > 2048) nb = 2048;
  w.bwd_nblk = (int)nb;
  w.pw_off = off; off += (int64_t)w.bwd_nblk * d->C * ncls;
  w.pb_off = off; off += (int64_t)w.bwd_nblk * ncls;
  off = (off + 3) & ~(int64_t)3;
  w.tmp_off = off; off += unetk_rows_reduce_tmp_floats(1, w.bwd_nblk, d->C * ncls);
  w.total = off;
  return w;
}

__global__ void label_hist_kernel(const int32_t* __restrict__ labels, int HW, int ncls, int bps, int* __restrict__ hist) {
  __shared__ int sh[MAXC];
  const int b = blockIdx.x / bps, blk = blockIdx.x % bps;
  if (threadIdx.x < MAXC) sh[threadIdx.x] = 0;
  __syncthreads();
  int cnt[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) cnt[k] = 0;
  for (int i = blk * blockDim.x + threadIdx.x; i < HW; i += bps * blockDim.x) {
    const int l = labels[(int64_t)b * HW + i];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) cnt[k] += (l == k);
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    int v = cnt[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[k], v);
  }
  __syncthreads();
  if (threadIdx.x < ncls && sh[threadIdx.x]) atomicAdd(&hist[b * ncls + threadIdx.x], sh[threadIdx.x]);
}

// loss_metrics.py:125-143 + :163-165: class weight table, normalised so the per-sample mean weight is 1
__global__ void weight_table_kernel(unetk_head_desc d, const int* __restrict__ hist, float* __restrict__ wn) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.N) return;
  float cw[MAXC];
  if (d.weight_mode == UNETK_W_NUMERICAL) {
    for (int k = 0; k < d.ncls; ++k) cw[k] = d.numeric_w[k];
  } else {  // proportion
    float tot = 0.f;
    for (int k = 0; k < d.ncls; ++k) {
      float n = (float)hist[b * d.ncls + k];
      if (d.proportion_decay > 0.f) n += d.proportion_decay;
      cw[k] = 1.0f / n;
      tot += cw[k];
    }
    for (int k = 0; k < d.ncls; ++k) cw[k] = cw[k] / tot;
  }
  float wsum = 0.f;
  for (int k = 0; k < d.ncls; ++k) wsum += cw[k] * (float)hist[b * d.ncls + k];
  for (int k = 0; k < d.ncls; ++k) wn[b * d.ncls + k] = cw[k] / wsum * (float)d.HW;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int NCLS>
__device__ __forceinline__ void pixel_logits(const float4 zv, const float (&wr)[4][NCLS], const float (&bias)[NCLS],
                                             int lpp, float (&lg)[NCLS]) {
#pragma unroll
  for (int k = 0; k < NCLS; ++k) {
    float s = zv.x * wr[0][k];
    s = fmaf(zv.y, wr[1][k], s);
    s = fmaf(zv.z, wr[2][k], s);
    s = fmaf(zv.w, wr[3][k], s);
    if (lpp == 16) {
      // C = 64 (every 2-D net): the 16 lanes of a pixel are one DPP row -- quad swaps, then the half-row and row mirrors, sum
      // it on the vector pipe.  The generic loop below compiles to ds_bpermute (an LDS crossbar instruction with a computed
      // address and its own wait) four times per class and pixel: that, not the softmax tail, was what held this kernel at
      // 0.2 ms whatever the storage type.
      s += dpp_f<0xB1>(s);       // quad_perm [1, 0, 3, 2]
      s += dpp_f<0x4E>(s);       // quad_perm [2, 3, 0, 1]
      s += dpp_f<0x141>(s);      // row_half_mirror
      s += dpp_f<0x140>(s);      // row_mirror
    } else {
      for (int o = lpp >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
    }
    lg[k] = s + bias[k];
  }
}

template <int NCLS, typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(unetk_head_desc d, const T* __restrict__ z,
                                                       const float* __restrict__ w, const float* __restrict__ bvec,
                                                       const int32_t* __restrict__ labels, const float* __restrict__ pixel_w,
                                                       const float* __restrict__ wn, float* __restrict__ logits,
                                                       float* __restrict__ probs, float* __restrict__ part, int bps) {
  constexpr int NQ = 2 + (NCLS - 1) * 6;
  __shared__ float red[4][NQ];
  const int lpp = d.C >> 2;            // lanes per pixel
  const int gpb = 256 / lpp;           // pixel groups per block
  const int gl = threadIdx.x % lpp, grp = threadIdx.x / lpp;
  const int b = blockIdx.x / bps, blk = blockIdx.x % bps;

  float wr[4][NCLS], bias[NCLS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < NCLS; ++k) wr[j][k] = w[(gl * 4 + j) * NCLS + k];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) bias[k] = bvec[k];

  float q[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) q[i] = 0.f;

  // (Round 4, measured and not kept: ONE LANE PER PIXEL for C = 64 -- the wave's 64 x 64 tile staged through LDS with a
  // conflict-free transposed read, a 64-term fma chain per class against the filter read back from LDS as broadcasts, no
  // cross-lane traffic, 98 VGPRs -- was no faster in fp32 (0.177 vs 0.171 ms at 256^2 bs 32) and twice as slow on bf16 input
  // (0.245 vs 0.128 ms at 512^2 bs 8): with 70 KB of LDS per block only two waves per SIMD are resident, and this kernel lives
  // on loads in flight, not on vector issue slots.)
  // A group of lpp lanes shares a pixel's channel dot products (xor-shuffle sums leave the logits in every lane).  The
  // softmax / loss / metric arithmetic that follows is ~100 instructions per pixel: done by ONE lane of the group it made
  // the kernel VALU-bound at 1.3-2.7 TB/s.  So a group takes lpp consecutive pixels per pass and lane j keeps pixel j's
  // logits: the per-pixel tail then runs in all lanes at once.
  for (int i0 = (blk * gpb + grp) * lpp; i0 < d.HW; i0 += bps * gpb * lpp) {
    float lg[NCLS];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) lg[k] = 0.f;
    const int step = (lpp & 3) == 0 ? 4 : 1;                 // four loads in flight per shuffle chain
    for (int pp = 0; pp < lpp; pp += step) {
      float4 zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ip = i0 + pp + u;
        zv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < step && ip < d.HW) zv[u] = ld4(z + ((int64_t)b * d.HW + ip) * d.C + gl * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < step) {
          float t[NCLS];
          pixel_logits<NCLS>(zv[u], wr, bias, lpp, t);
#pragma unroll
          for (int k = 0; k < NCLS; ++k) lg[k] = (gl == pp + u) ? t[k] : lg[k];
        }
      }
    }
    const int i = i0 + gl;
    const int64_t pix = (int64_t)b * d.HW + i;
    if (i < d.HW) {
      float mx = lg[0];
#pragma unroll
      for (int k = 1; k < NCLS; ++k) mx = fmaxf(mx, lg[k]);
      float e[NCLS], se = 0.f;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) { e[k] = expf(lg[k] - mx); se += e[k]; }
      const float inv = 1.0f / se;
      const int lab = labels ? labels[pix] : 0;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) {
        logits[pix * NCLS + k] = lg[k];
        if (probs) probs[pix * NCLS + k] = e[k] * inv;
      }
      if (labels) {
        float lsel = lg[0];
#pragma unroll
        for (int k = 1; k < NCLS; ++k) lsel = (lab == k) ? lg[k] : lsel;
        const float ce = (mx - lsel) + logf(se);
        float wt = 1.f;
        if (d.weight_mode == UNETK_W_PIXELMAP) wt = pixel_w[pix];
        else if (d.weight_mode != UNETK_W_NONE) wt = wn[b * NCLS + lab];
        q[0] += ce * wt;
        q[1] += (wt != 0.f) ? 1.f : 0.f;
#pragma unroll
        for (int c = 1; c < NCLS; ++c) {
          const float pc = e[c] * inv;
          const float lc = (lab == c) ? 1.f : 0.f;
          const float pr = pc > 0.5f ? 1.f : 0.f;
          float* qq = &q[2 + (c - 1) * 6];
          qq[0] += pr * lc;
          qq[1] += pr;
          qq[2] += lc;
          qq[3] += fminf(pr + lc, 1.f);
          qq[4] += lc * pc;
          qq[5] += lc + pc;
        }
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const float v = wave_sum(q[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NQ)
    part[((int64_t)b * bps + blk) * NQ + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one thread per sample; then thread 0 combines in sample order
__global__ void head_finalize_kernel(unetk_head_desc d, const float* __restrict__ part, int bps, int nq,
                                     float* __restrict__ result) {
  extern __shared__ double sh[];  // [N][3]: ce_sum, present, dice_term; [N][nq] column sums; [N * nq][L] slice sums
  const int ncls = d.ncls;
  double* colsum = sh + (int64_t)d.N * 3;
  double* slice = colsum + (int64_t)d.N * nq;
  // L threads per (sample, quantity), each a fixed-order sum over every L-th partial row, combined in slice order: one
  // thread walking all bps rows (864 of them for a 96^3 patch) made this single-block kernel 90 us of serial loads
  const int P = d.N * nq;
  int L = 1;
  while (L < 64 && 2 * L * P <= (int)blockDim.x) L *= 2;
  for (int t = threadIdx.x; t < P * L; t += blockDim.x) {
    const int pq = t / L, sl = t - pq * L;
    const int b = pq / nq, i = pq - b * nq;
    double a = 0.0;
    for (int j = sl; j < bps; j += L) a += (double)part[((int64_t)b * bps + j) * nq + i];
    slice[t] = a;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < P; t += blockDim.x) {
    double a = 0.0;
    for (int k = 0; k < L; ++k) a += slice[t * L + k];
    colsum[t] = a;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < d.N; b += blockDim.x) {
    const double* acc = colsum + (int64_t)b * nq;
    double I = 0.0, U = 0.0;
    for (int c = 1; c < ncls; ++c) {
      const double* qq = &acc[2 + (c - 1) * 6];
      float* r = result + 3 + ((int64_t)b * (ncls - 1) + (c - 1)) * 4;
      r[0] = (float)qq[0]; r[1] = (float)qq[1]; r[2] = (float)qq[2]; r[3] = (float)qq[3];
      I += qq[4];
      U += qq[5];
    }
    float* iu = result + 3 + (int64_t)d.N * (ncls - 1) * 4 + b * 2;
    iu[0] = (float)I;
    iu[1] = (float)U;
    sh[b * 3 + 0] = acc[0];
    sh[b * 3 + 1] = acc[1];
    sh[b * 3 + 2] = 2.0 * (double)(float)I / ((double)(float)U + 1e-8);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ce = 0.0, np = 0.0, dm = 0.0;
    for (int b = 0; b < d.N; ++b) { ce += sh[b * 3]; np += sh[b * 3 + 1]; dm += sh[b * 3 + 2]; }
    result[0] = np > 0.0 ? (float)(ce / np) : 0.f;
    result[1] = (float)(1.0 - dm / (double)d.N);
    result[2] = (float)np;
  }
}

template <int NCLS, typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(unetk_head_desc d, const T* __restrict__ z,
                                                       const float* __restrict__ w, const int32_t* __restrict__ labels,
                                                       const float* __restrict__ pixel_w, const float* __restrict__ wn,
                                                       const float* __restrict__ logits, const float* __restrict__ result,
                                                       float xs, float ds, const float* __restrict__ dev_scales,
                                                       T* __restrict__ dz, float* __restrict__ pw,
                                                       float* __restrict__ pb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [gpb][C*NCLS] then [4][NCLS]
  const int lpp = d.C >> 2, gpb = 256 / lpp;
  const int gl = threadIdx.x % lpp, grp = threadIdx.x / lpp;
  const int64_t npix = (int64_t)d.N * d.HW;
  float wr[4][NCLS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < NCLS; ++k) wr[j][k] = w[(gl * 4 + j) * NCLS + k];
  float aw[4][NCLS], abias[NCLS];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) {
    abias[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) aw[j][k] = 0.f;
  }
  if (dev_scales != nullptr) {
    xs *= dev_scales[0];
    ds *= dev_scales[1];
  }
  const float num_present = result[2];
  const float xscale = (xs != 0.f && num_present > 0.f) ? xs / num_present : 0.f;
  const float* iu = result + 3 + (int64_t)d.N * (NCLS - 1) * 4;

  for (int64_t pix = (int64_t)blockIdx.x * gpb + grp; pix < npix; pix += (int64_t)gridDim.x * gpb) {
    const float4 zv = ld4(z + pix * d.C + gl * 4);       // issued first: its latency overlaps the softmax arithmetic below
    const int b = (int)(pix / d.HW);
    const int lab = labels[pix];
    float lg[NCLS], p[NCLS], dl[NCLS];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < NCLS; ++k) { lg[k] = logits[pix * NCLS + k]; mx = fmaxf(mx, lg[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NCLS; ++k) { p[k] = expf(lg[k] - mx); se += p[k]; }
    const float inv = 1.0f / se;
#pragma unroll
    for (int k = 0; k < NCLS; ++k) { p[k] *= inv; dl[k] = 0.f; }
    if (xscale != 0.f) {
      float wt = 1.f;
      if (d.weight_mode == UNETK_W_PIXELMAP) wt = pixel_w[pix];
      else if (d.weight_mode != UNETK_W_NONE) wt = wn[b * NCLS + lab];
      const float f = xscale * wt;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) dl[k] = f * (p[k] - ((lab == k) ? 1.f : 0.f));
    }
    if (ds != 0.f) {
      const float I = iu[b * 2], U = iu[b * 2 + 1] + 1e-8f;
      const float coef = -2.0f * ds / ((float)d.N * U * U);
      float g[NCLS], pg = 0.f;
      g[0] = 0.f;
#pragma unroll
      for (int c = 1; c < NCLS; ++c) {
        g[c] = coef * (((lab == c) ? U : 0.f) - I);
        pg += p[c] * g[c];
      }
#pragma unroll
      for (int k = 0; k < NCLS; ++k) dl[k] += p[k] * (g[k] - pg);
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
      o.x = fmaf(dl[k], wr[0][k], o.x);
      o.y = fmaf(dl[k], wr[1][k], o.y);
      o.z = fmaf(dl[k], wr[2][k], o.z);
      o.w = fmaf(dl[k], wr[3][k], o.w);
      aw[0][k] = fmaf(zv.x, dl[k], aw[0][k]);
      aw[1][k] = fmaf(zv.y, dl[k], aw[1][k]);
      aw[2][k] = fmaf(zv.z, dl[k], aw[2][k]);
      aw[3][k] = fmaf(zv.w, dl[k], aw[3][k]);
      if (gl == 0) abias[k] += dl[k];
    }
    st4(dz + pix * d.C + gl * 4, o);
  }
  // block reduction over pixel groups (fixed order)
  const int CW = d.C * NCLS;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < NCLS; ++k) smem[grp * CW + (gl * 4 + j) * NCLS + k] = aw[j][k];
  __syncthreads();
  for (int i = threadIdx.x; i < CW; i += 256) {
    float s = 0.f;
    for (int g2 = 0; g2 < gpb; ++g2) s += smem[g2 * CW + i];
    pw[(int64_t)blockIdx.x * CW + i] = s;
  }
  __syncthreads();
  float* redb = smem;  // [4][NCLS]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NCLS; ++k) {
    const float v = wave_sum(abias[k]);
    if (lane == 0) redb[wave * NCLS + k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NCLS)
    pb[(int64_t)blockIdx.x * NCLS + threadIdx.x] =
        redb[threadIdx.x] + redb[NCLS + threadIdx.x] + redb[2 * NCLS + threadIdx.x] + redb[3 * NCLS + threadIdx.x];
}

__global__ void head_predict_kernel(const float* __restrict__ probs, int64_t npix, int ncls, uint8_t* __restrict__ amax,
                                    uint8_t* __restrict__ preds) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    float best = probs[i * ncls];
    int bi = 0;
    for (int k = 1; k < ncls; ++k) {
      const float v = probs[i * ncls + k];
      if (v > best) { best = v; bi = k; }   // strict: lowest index wins ties (np.argmax)
      if (preds) preds[(int64_t)(k - 1) * npix + i] = v > 0.5f ? 1 : 0;
    }
    if (amax) amax[i] = (uint8_t)bi;
  }
}

bool head_desc_ok(const unetk_head_desc* d) {
  if (!d || d->N <= 0 || d->HW <= 0 || d->C <= 0) return false;
  if (d->ncls < 2 || d->ncls > MAXC) return false;
  if (d->weight_mode < UNETK_W_NONE || d->weight_mode > UNETK_W_PIXELMAP) return false;
  return true;
}
bool head_shape_supported(const unetk_head_desc* d) {
  const int lpp = d->C / 4;
  return d->C % 4 == 0 && lpp >= 1 && lpp <= 64 && (lpp & (lpp - 1)) == 0 && d->N <= 2048;
}

}  // namespace

extern "C" size_t unetk_head_result_floats(const unetk_head_desc* d) {
  if (!head_desc_ok(d)) return 0;
  return 3 + (size_t)d->N * (d->ncls - 1) * 4 + (size_t)d->N * 2;
}

extern "C" size_t unetk_head_ws_bytes(const unetk_head_desc* d) {
  if (!head_desc_ok(d)) return 0;
  return (size_t)head_ws(d).total * sizeof(float);
}

#define HEAD_DISPATCH(NC, CALL) \
  switch (NC) {                 \
    case 2: { constexpr int K_ = 2; CALL; } break; \
    case 3: { constexpr int K_ = 3; CALL; } break; \
    case 4: { constexpr int K_ = 4; CALL; } break; \
    case 5: { constexpr int K_ = 5; CALL; } break; \
    case 6: { constexpr int K_ = 6; CALL; } break; \
    case 7: { constexpr int K_ = 7; CALL; } break; \
    default: { constexpr int K_ = 8; CALL; } break; \
  }

extern "C" int unetk_head_fwd(const unetk_head_desc* d, const void* z, const float* w, const float* b,
                              const int32_t* labels, const float* pixel_w, float* logits, float* probs,
                              float* result, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(head_desc_ok(d) && z && w && b && logits);
  if (!head_shape_supported(d)) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE((d->storage == UNETK_FP32 && unetk_aligned16(z)) || (bs && unetk_aligned8(z)));
  hipStream_t st = (hipStream_t)stream;
  const HeadWs L = head_ws(d);
  float* wsf = (float*)ws;
  float* wn = nullptr;
  float* part = nullptr;
  if (labels) {
    UNETK_REQUIRE(result && ws);
    if (ws_bytes < (size_t)L.total * sizeof(float)) return UNETK_E_WORKSPACE;
    UNETK_REQUIRE(d->weight_mode != UNETK_W_PIXELMAP || pixel_w);
    wn = wsf + L.wn_off;
    part = wsf + L.part_off;
    if (d->weight_mode == UNETK_W_NUMERICAL || d->weight_mode == UNETK_W_PROPORTION) {
      int* hist = (int*)(wsf + L.hist_off);
      hipError_t e = hipMemsetAsync(hist, 0, (size_t)d->N * d->ncls * sizeof(int), st);
      if (e != hipSuccess) return (int)e;
      UNETK_LAUNCH(label_hist_kernel, dim3(d->N * L.bps), dim3(256), 0, st, labels, d->HW, d->ncls, L.bps, hist);
      UNETK_LAUNCH_CHECK();
      UNETK_LAUNCH(weight_table_kernel, dim3((d->N + 63) / 64), dim3(64), 0, st, *d, hist, wn);
      UNETK_LAUNCH_CHECK();
    }
  }
  int bps = L.bps;
  float* part_eff = part;
  if (!labels) {
    // inference: partials are not produced; reuse a dummy row in logits-free space is not possible ->
    // require ws for the partial rows as well (tiny)
    UNETK_REQUIRE(ws && ws_bytes >= (size_t)L.total * sizeof(float));
    part_eff = wsf + L.part_off;
  }
  if (bs) {
    HEAD_DISPATCH(d->ncls, UNETK_LAUNCH((head_fwd_kernel<K_, bf16_t>), dim3(d->N * bps), dim3(256), 0, st, *d,
                                              (const bf16_t*)z, w, b, labels, pixel_w, wn, logits, probs, part_eff, bps));
  } else {
    HEAD_DISPATCH(d->ncls, UNETK_LAUNCH((head_fwd_kernel<K_, float>), dim3(d->N * bps), dim3(256), 0, st, *d,
                                              (const float*)z, w, b, labels, pixel_w, wn, logits, probs, part_eff, bps));
  }
  UNETK_LAUNCH_CHECK();
  if (labels) {
    {
      const int P = d->N * L.nq;
      int Ls = 1;
      while (Ls < 64 && 2 * Ls * P <= 1024) Ls *= 2;        // the kernel's own choice for 1024 threads
      UNETK_LAUNCH(head_finalize_kernel, dim3(1), dim3(1024),
                         ((size_t)d->N * (3 + L.nq) + (size_t)P * Ls) * sizeof(double), st, *d, part, bps, L.nq, result);
    }
    UNETK_LAUNCH_CHECK();
  }
  return UNETK_OK;
}

extern "C" int unetk_head_bwd(const unetk_head_desc* d, const void* z, const float* w, const int32_t* labels,
                              const float* pixel_w, const float* logits, const float* result, float xent_scale,
                              float dice_scale, const float* dev_scales, void* dz, float* dw, float* db, void* ws,
                              size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(head_desc_ok(d) && z && w && labels && logits && result && dz && dw && db && ws);
  if (!head_shape_supported(d)) return UNETK_E_UNSUPPORTED;
  const bool bs = d->storage == UNETK_BF16S;
  UNETK_REQUIRE((d->storage == UNETK_FP32 && unetk_aligned16(z) && unetk_aligned16(dz)) ||
                (bs && unetk_aligned8(z) && unetk_aligned8(dz)));
  UNETK_REQUIRE(d->weight_mode != UNETK_W_PIXELMAP || pixel_w);
  hipStream_t st = (hipStream_t)stream;
  const HeadWs L = head_ws(d);
  if (ws_bytes < (size_t)L.total * sizeof(float)) return UNETK_E_WORKSPACE;
  float* wsf = (float*)ws;
  const float* wn = wsf + L.wn_off;
  float* pw = wsf + L.pw_off;
  float* pb = wsf + L.pb_off;
  float* tmp = wsf + L.tmp_off;
  const int gpb = 256 / (d->C / 4);
  const size_t lds = (size_t)gpb * d->C * d->ncls * sizeof(float);
  if (lds > 64 * 1024) return UNETK_E_UNSUPPORTED;
  if (bs) {
    HEAD_DISPATCH(d->ncls, UNETK_LAUNCH((head_bwd_kernel<K_, bf16_t>), dim3(L.bwd_nblk), dim3(256), lds, st, *d,
                                              (const bf16_t*)z, w, labels, pixel_w, wn, logits, result, xent_scale,
                                              dice_scale, dev_scales, (bf16_t*)dz, pw, pb));
  } else {
    HEAD_DISPATCH(d->ncls, UNETK_LAUNCH((head_bwd_kernel<K_, float>), dim3(L.bwd_nblk), dim3(256), lds, st, *d,
                                              (const float*)z, w, labels, pixel_w, wn, logits, result, xent_scale,
                                              dice_scale, dev_scales, (float*)dz, pw, pb));
  }
  UNETK_LAUNCH_CHECK();
  int rc = unetk_rows_reduce(pw, 1, L.bwd_nblk, d->C * d->ncls, dw, tmp, st);
  if (rc != UNETK_OK) return rc;
  return unetk_rows_reduce(pb, 1, L.bwd_nblk, d->ncls, db, tmp, st);
}

extern "C" int unetk_head_predict(const float* probs, int64_t npix, int ncls, uint8_t* argmax, uint8_t* preds,
                                  void* stream) {
  UNETK_REQUIRE(probs && npix > 0 && ncls >= 2 && ncls <= MAXC && (argmax || preds));
  int64_t g = (npix + 255) / 256;
  if (g > 4096) g = 4096;
  UNETK_LAUNCH(head_predict_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, probs, npix, ncls, argmax, preds);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
