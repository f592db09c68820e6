// conv3x3 filter gradient (Conv2DBackpropFilter of slim.conv2d(x, C, 3), NetworksV2/UNet.py:79,85,94)
// as a split-K GEMM on the fp32 matrix cores: dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co].
//
// GEMM view: M = ci, N = co, K = pixels.  A block owns a 64(ci) x 64(co) x 9(taps) output panel and
// walks a contiguous range of 8x16 pixel tiles; per tile it stages the 10x18 x-halo and the 8x16 dy
// tile in LDS once and ALL NINE taps accumulate from them (each wave: a 32x32 panel for each of the
// 9 taps = 144 accumulator registers).  Partial panels go to a workspace slab per split; a fixed-order
// reduction sums the slabs -> bit-reproducible, no atomics (SURVEY.md 7 "wgrad").
#include "common.h"

namespace {

constexpr int TW = 16, TH = 8, HWD = TW + 2, HH = TH + 2;
constexpr int CT = 64;  // channel tile (both ci and co)

struct WgParams {
  const float* x;
  const float* dy;
  float* slab;
  int N, H, W, Cin, Cout, xs, ys;
  int tiles_h, tiles_w, total_tiles, tiles_per_split, n_ci_tiles, n_co_tiles;
};

__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xh = smem;                    // [HH*HWD][CT]
  float* dyt = smem + HH * HWD * CT;   // [TH*TW][CT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave >> 1, wco = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int split = bid;
  const int ci0 = ci_t * CT, co0 = co_t * CT;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.total_tiles);
  const int a_lane = wci * 32 + l31, b_lane = wco * 32 + l31;

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tw_i = tile % p.tiles_w;
    const int th_i = (tile / p.tiles_w) % p.tiles_h;
    const int n_img = tile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * TH, w0 = tw_i * TW;

    __syncthreads();  // previous tile's fragment reads are done
    // stage x halo: HH*HWD pixels x 16 float4
    for (int idx = tid; idx < HH * HWD * (CT / 4); idx += 256) {
      const int pix = idx >> 4, q = idx & 15;
      const int hh = pix / HWD, ww = pix - hh * HWD;
      const int gh = h0 - 1 + hh, gw = w0 - 1 + ww;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W)
        v = ldg4(p.x + (((int64_t)n_img * p.H + gh) * p.W + gw) * p.xs + ci0 + q * 4);
      *reinterpret_cast<float4*>(&xh[pix * CT + q * 4]) = v;
    }
    // stage dy tile: TH*TW pixels x 16 float4
    for (int idx = tid; idx < TH * TW * (CT / 4); idx += 256) {
      const int pix = idx >> 4, q = idx & 15;
      const int gh = h0 + (pix >> 4), gw = w0 + (pix & 15);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh < p.H && gw < p.W)
        v = ldg4(p.dy + (((int64_t)n_img * p.H + gh) * p.W + gw) * p.ys + co0 + q * 4);
      *reinterpret_cast<float4*>(&dyt[pix * CT + q * 4]) = v;
    }
    __syncthreads();

    // 64 k-steps; lane half h takes the odd/even column of each pixel pair
    for (int r = 0; r < TH; ++r) {
#pragma unroll 4
      for (int c2 = 0; c2 < TW / 2; ++c2) {
        const int col = 2 * c2 + h;
        const float b = dyt[(r * TW + col) * CT + b_lane];
        const float* xa = &xh[(r * HWD + col) * CT + a_lane];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float a = xa[(kh * HWD + kw) * CT];
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kh * 3 + kw], 0, 0, 0);
          }
      }
    }
  }

  // partial panel -> slab[split][tap][ci][co]
  float* out = p.slab + (int64_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wci * 32 + mfma32_row(r, h);
      out[((int64_t)t * p.Cin + ci) * p.Cout + co0 + b_lane] = acc[t][r];
    }
}

// Small-Cin filter gradient (Encode1/conv1, Cin = 3): HBM-bound on reading dy once.
// thread = (co, pixel lane); 9*Cin accumulators per thread; Cin <= 4.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_wgrad_smallc_kernel(WgParams p) {
  const int tid = threadIdx.x;
  const int PL = 256 / p.Cout;  // Cout <= 256, divides 256
  const int co = tid % p.Cout, pl = tid / p.Cout;
  float acc[9 * CIN];
#pragma unroll
  for (int i = 0; i < 9 * CIN; ++i) acc[i] = 0.f;
  const int split = blockIdx.x;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.total_tiles);
  if (pl < PL) {
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int tw_i = tile % p.tiles_w;
      const int th_i = (tile / p.tiles_w) % p.tiles_h;
      const int n_img = tile / (p.tiles_w * p.tiles_h);
      const int h0 = th_i * TH, w0 = tw_i * TW;
      for (int pix = pl; pix < TH * TW; pix += PL) {
        const int gh = h0 + (pix >> 4), gw = w0 + (pix & 15);
        if (gh >= p.H || gw >= p.W) continue;
        const float g = p.dy[(((int64_t)n_img * p.H + gh) * p.W + gw) * p.ys + co];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = gh + kh - 1;
          if (ih < 0 || ih >= p.H) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = gw + kw - 1;
            if (iw < 0 || iw >= p.W) continue;
            const float* xp = p.x + (((int64_t)n_img * p.H + ih) * p.W + iw) * p.xs;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc[(kh * 3 + kw) * CIN + ci] = fmaf(xp[ci], g, acc[(kh * 3 + kw) * CIN + ci]);
          }
        }
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [PL][9*CIN][Cout]
  if (pl < PL) {
#pragma unroll
    for (int i = 0; i < 9 * CIN; ++i) smem[(pl * 9 * CIN + i) * p.Cout + co] = acc[i];
  }
  __syncthreads();
  float* out = p.slab + (int64_t)split * 9 * CIN * p.Cout;
  for (int i = tid; i < 9 * CIN * p.Cout; i += 256) {
    float s = 0.f;
    for (int j = 0; j < PL; ++j) s += smem[(int64_t)j * 9 * CIN * p.Cout + i];
    out[i] = s;
  }
}

}  // namespace

// dst[i] = sum_s slab[s][i], fixed order.  n % 4 == 0.
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int S, int64_t n, float* __restrict__ dst) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float4 v = ldg4(slab + (int64_t)s * n + i * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    stg4(dst + i * 4, a);
  }
}

int unetk_launch_slab_reduce(const float* slab, int S, int64_t n, float* dst, hipStream_t st) {
  const int64_t n4 = n >> 2;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, st, slab, S, n, dst);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

namespace {

struct WgPlan {
  int mode;  // 0 mfma, 1 small-Cin, -1 unsupported
  int tiles_h, tiles_w, total_tiles, S, tiles_per_split, n_ci_tiles, n_co_tiles;
};

WgPlan wg_plan(const unetk_conv_desc* d) {
  WgPlan pl{};
  pl.tiles_h = (d->H + TH - 1) / TH;
  pl.tiles_w = (d->W + TW - 1) / TW;
  pl.total_tiles = d->N * pl.tiles_h * pl.tiles_w;
  if (d->Cin % CT == 0 && d->Cout % CT == 0) {
    pl.mode = 0;
    pl.n_ci_tiles = d->Cin / CT;
    pl.n_co_tiles = d->Cout / CT;
    const int panels = pl.n_ci_tiles * pl.n_co_tiles;
    int S = (1024 + panels - 1) / panels;  // ~2 blocks/CU x 256 CUs x 2 rounds
    if (S > pl.total_tiles) S = pl.total_tiles;
    if (S < 1) S = 1;
    pl.tiles_per_split = (pl.total_tiles + S - 1) / S;
    pl.S = (pl.total_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
  } else if (d->Cin <= 4 && d->Cout <= 256 && 256 % d->Cout == 0) {
    pl.mode = 1;
    int S = 2048;
    if (S > pl.total_tiles) S = pl.total_tiles;
    pl.tiles_per_split = (pl.total_tiles + S - 1) / S;
    pl.S = (pl.total_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
  } else {
    pl.mode = -1;
  }
  return pl;
}

}  // namespace

extern "C" size_t unetk_conv3x3_wgrad_ws_bytes(const unetk_conv_desc* d) {
  if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
  const WgPlan pl = wg_plan(d);
  if (pl.mode < 0) return 0;
  return (size_t)pl.S * 9 * d->Cin * d->Cout * sizeof(float);
}

extern "C" int unetk_conv3x3_wgrad(const unetk_conv_desc* d, const float* x, const float* dy, float* dw,
                                   void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(d && x && dy && dw && ws);
  UNETK_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0);
  UNETK_REQUIRE(d->x_stride >= d->Cin && d->y_stride >= d->Cout);
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(dy) && unetk_aligned16(dw) && unetk_aligned16(ws));
  const WgPlan pl = wg_plan(d);
  if (pl.mode < 0) return UNETK_E_UNSUPPORTED;
  if (ws_bytes < unetk_conv3x3_wgrad_ws_bytes(d)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  WgParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.xs = d->x_stride; p.ys = d->y_stride;
  p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.total_tiles = pl.total_tiles;
  p.tiles_per_split = pl.tiles_per_split; p.n_ci_tiles = pl.n_ci_tiles; p.n_co_tiles = pl.n_co_tiles;
  if (pl.mode == 0) {
    UNETK_REQUIRE(d->x_stride % 4 == 0 && d->y_stride % 4 == 0);
    constexpr size_t lds = (size_t)(HH * HWD + TH * TW) * CT * sizeof(float);  // 78848 B
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    const int grid = pl.S * pl.n_ci_tiles * pl.n_co_tiles;
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(grid), dim3(256), lds, st, p);
    UNETK_LAUNCH_CHECK();
  } else {
    const int PL = 256 / d->Cout;
    const size_t lds = (size_t)PL * 9 * d->Cin * d->Cout * sizeof(float);
    if (lds > 64 * 1024) return UNETK_E_UNSUPPORTED;
    switch (d->Cin) {
      case 1: hipLaunchKernelGGL(conv3x3_wgrad_smallc_kernel<1>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 2: hipLaunchKernelGGL(conv3x3_wgrad_smallc_kernel<2>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 3: hipLaunchKernelGGL(conv3x3_wgrad_smallc_kernel<3>, dim3(pl.S), dim3(256), lds, st, p); break;
      default: hipLaunchKernelGGL(conv3x3_wgrad_smallc_kernel<4>, dim3(pl.S), dim3(256), lds, st, p); break;
    }
    UNETK_LAUNCH_CHECK();
  }
  return unetk_launch_slab_reduce((const float*)ws, pl.S, (int64_t)9 * d->Cin * d->Cout, dw, st);
}
