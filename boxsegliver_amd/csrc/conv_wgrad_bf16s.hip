// conv3x3 filter gradient with bf16 STORAGE (UNETK_BF16S): x and dy are bf16 in HBM, dW accumulates in fp32.
//   dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co]        (Conv2DBackpropFilter of
//   slim.conv2d(x, C, 3), NetworksV2/UNet.py:79,85,94, in the mixed-precision mode of BASELINE.json configs[2])
//
// GEMM view: M = ci, N = co, K = pixels, on v_mfma_f32_32x32x16_bf16.  Both operands are K-STRIDED in memory (NHWC:
// the 8 pixels a lane needs for its channel are 8 different rows), which the fp32-storage kernel paid for with 38
// ds_read_b32 + 19 conversions per 9 MFMAs (LDS-read bound at ~520 TFLOP/s).  Here the tiles stay bf16 in LDS exactly
// as they lie in HBM ([pixel][64 channels], 128-byte rows) and are read through the LDS TRANSPOSE path:
// ds_read_b64_tr_b16 hands each lane four consecutive pixels of ONE channel.
//   * per 16-pixel k-step a wave issues 2 transposed reads for its dy fragment and 3 per filter row for x (12 consecutive
//     halo pixels of its channel); the three kw taps of a filter row are the windows [kw, kw + 8) of those 12 values --
//     kw = 0 and kw = 2 are register sub-ranges, kw = 1 costs four v_alignbit.  11 reads per 9 MFMAs.
//   * tiles arrive by async direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per wave instruction) into a 2-stage
//     ring, as in the fp32 kernel; the image is lane-linear, so the bank swizzle is applied on the GLOBAL side: lane L of
//     a piece fetches 16-byte chunk (L & 7) ^ 4 * bit1(row) of its row.  With it any four consecutive rows x 32 channels
//     (one transposed read of a 32-lane half) cover four distinct 64-byte bank groups: conflict-free for every tap shift.
//   * 512 threads = 2 (ci) x 2 (co) x 2 (pixel-row halves) waves own a 64 x 64 x 9-tap panel; split-K over pixel tiles
//     into slabs + the fixed-order slab reduction of conv_wgrad.hip => bit-reproducible, no atomics.
#include "common.h"

namespace {

constexpr int TW = 16, TH = 8, HWD = TW + 2;
constexpr int HALO_PIX = (TH + 2) * HWD;          // 180
constexpr int CT = 64;                            // channels per panel side = one 128-byte LDS row
constexpr int NP_X = (HALO_PIX + 7) / 8;          // 23 one-KiB pieces (8 rows each) of the x halo
constexpr int NP_Y = TH * TW / 8;                 // 16 pieces of the dy tile
constexpr int NP = NP_X + NP_Y;                   // 39
constexpr int IPW = (NP + 7) / 8;                 // pieces per wave
constexpr int XH_B = NP_X * 1024, STAGE_B = NP * 1024;
constexpr int RED_B = 4 * 144 * 64 * 4;           // pixel-row-half reduction scratch: [4 waves][9 x 16][64 lanes] floats
constexpr int LDS_B = 2 * STAGE_B > RED_B ? 2 * STAGE_B : RED_B;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// four consecutive rows (pixels) of one channel column: lane 4q + p of a 16-lane group addresses row q, channels 4p..4p+3
// of the group's 16; lane i receives channel i, row q in element q.
__device__ __forceinline__ uint2 tr_read(const char* base, int byte_off) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + byte_off));
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_bf16s_kernel(WgParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][STAGE_B]; reused as the reduction scratch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wci = (wave >> 1) & 1, ks = wave >> 2;
  const int l31 = lane & 31, h = lane >> 5;

  int bid = xcd_remap(blockIdx.x, gridDim.x);      // the panels of one pixel split run together on one XCD: its L2 serves
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;   // the x / dy tiles they share
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int split = bid;
  const int ci0 = ci_t * CT, co0 = co_t * CT;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy);

  // ---- staging: wave w issues pieces j = w + 8 i; lane = (row of the piece, 16-byte slot); the chunk it fetches is
  // the slot XOR-swizzled by bit 1 of the row (row = 8 j + (lane >> 3): bit 1 = lane bit 4)
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (((lane >> 4) & 1) << 2);
  int rel_h[IPW], rel_w[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int j = wave + 8 * i;
    if (j < NP_X) {
      const int pix = 8 * j + lrow;
      rel_h[i] = pix < HALO_PIX ? pix / HWD - 1 : (1 << 20);      // relative to the tile origin; 1 << 20 = dummy row
      rel_w[i] = pix % HWD - 1;
    } else {
      const int pix = 8 * (j - NP_X) + lrow;
      rel_h[i] = pix / TW;
      rel_w[i] = pix % TW;
    }
  }
  auto issue_tile = [&](int tile, int stage) {
    const int tw_i = tile % p.tiles_w;
    const int th_i = (tile / p.tiles_w) % p.tiles_h;
    const int n_tile = tile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int64_t ximg = p.xa.off(n_tile), yimg = p.ya.off(n_tile);
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int j = wave + 8 * i;
      if (j < NP) {   // wave-uniform
        const bool is_x = j < NP_X;
        const int gh = h0 + rel_h[i], gw = w0 + rel_w[i];
        const bool ok = gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
        const int64_t pixoff = (int64_t)gh * p.W + gw;
        const bf16_t* src = is_x ? xb + ximg + pixoff * p.xs + ci0 + lchunk * 8 : dyb + yimg + pixoff * p.ys + co0 + lchunk * 8;
        if (!ok) src = reinterpret_cast<const bf16_t*>(p.zeros) + (lane & 7) * 8;
        char* dst = smem + stage * STAGE_B + j * 1024;   // wave-uniform; lanes land at dst + lane * 16 B
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // ---- fragment addressing.  16-lane group g = lane >> 4: channel 16-block (g & 1) of the wave's 32, pixel half h = g >> 1.
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, g16 = (lane >> 4) & 1;
  const int chunk_a = 2 * (wci * 2 + g16) + (p4 >> 1), chunk_b = 2 * (wco * 2 + g16) + (p4 >> 1);
  // byte offset of this lane's 8 bytes inside its row, for rows with bit 1 == (q4 >> 1) ^ par
  const int a_off0 = q4 * 128 + ((chunk_a ^ (((q4 >> 1) ^ 0) << 2)) << 4) + 8 * (p4 & 1) + 8 * h * 128;
  const int a_off1 = q4 * 128 + ((chunk_a ^ (((q4 >> 1) ^ 1) << 2)) << 4) + 8 * (p4 & 1) + 8 * h * 128;
  const int b_off = q4 * 128 + ((chunk_b ^ ((q4 >> 1) << 2)) << 4) + 8 * (p4 & 1) + 8 * h * 128;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.total_tiles);

  if (t_begin < t_end) issue_tile(t_begin, 0);
  int stage = 0;
  for (int tile = t_begin; tile < t_end; ++tile, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tile have landed ...
    __syncthreads();                                   // ... everybody's have, and the other stage is free
    if (tile + 1 < t_end) issue_tile(tile + 1, stage ^ 1);

    const char* xh = smem + stage * STAGE_B;
    const char* dyt = xh + XH_B;
    // this wave's four tile rows; (ks * 4) is even, so the swizzle parity of halo row (r + kh) is that of (rr + kh)
#pragma unroll
    for (int rr = 0; rr < TH / 2; ++rr) {
      const int r = ks * (TH / 2) + rr;
      const int brow = r * TW * 128 + b_off;
      const uint2 b0 = tr_read(dyt, brow), b1 = tr_read(dyt, brow + 4 * 128);
      const bf16x8 b = frag(b0.x, b0.y, b1.x, b1.y);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int arow = (r + kh) * HWD * 128 + (((rr + kh) & 1) ? a_off1 : a_off0);
        const uint2 a0 = tr_read(xh, arow), a1 = tr_read(xh, arow + 4 * 128), a2 = tr_read(xh, arow + 8 * 128);
        // 12 consecutive halo pixels of this lane's channel: (a0.x a0.y a1.x a1.y a2.x a2.y), two pixels per word
        acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a0.x, a0.y, a1.x, a1.y), b, acc[kh * 3 + 0], 0, 0, 0);
        acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            frag(__builtin_amdgcn_alignbit(a0.y, a0.x, 16), __builtin_amdgcn_alignbit(a1.x, a0.y, 16),
                 __builtin_amdgcn_alignbit(a1.y, a1.x, 16), __builtin_amdgcn_alignbit(a2.x, a1.y, 16)),
            b, acc[kh * 3 + 1], 0, 0, 0);
        acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a0.y, a1.x, a1.y, a2.x), b, acc[kh * 3 + 2], 0, 0, 0);
      }
    }
  }

  // ---- sum the two pixel-row halves through LDS (fixed order), then the ks == 0 waves write the split's slab
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][144][64 lanes]
  const int pidx = wci * 2 + wco;
  __syncthreads();
  if (ks == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(pidx * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (ks == 0) {
    float* out = p.slab + (int64_t)split * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wci * 32 + mfma32_row(r, h);
        out[((int64_t)t * p.Cin + ci) * p.Cout + co0 + wco * 32 + l31] = acc[t][r] + red[(pidx * 144 + t * 16 + r) * 64 + lane];
      }
  }
}

// First layer (9 * Cin <= 32 rows, Cout = 64; Encode1/conv1): x is the fp32 image, dy is bf16.  HBM-bound on reading dy
// once; same MFMA contraction as conv3x3_wgrad_c3_kernel (conv_wgrad.hip) with dy widened to fp32 on its way into LDS.
__global__ __launch_bounds__(256) void conv3x3_wgrad_c3_bf16s_kernel(WgParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dyt = reinterpret_cast<float*>(smem);          // [128 pixels][64]
  float* xh = dyt + TH * TW * CT;                        // [180 pixels][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = wave >> 1, wco = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int cin = p.Cin, m_rows = 9 * cin;
  const int split = blockIdx.x;
  const bool a_on = l31 < m_rows;
  const int a_tap = a_on ? l31 / cin : 0, a_ci = a_on ? l31 % cin : 0;
  const int a_off = ((a_tap / 3) * HWD + (a_tap % 3)) * 4 + a_ci;
  const bf16_t* dyb = reinterpret_cast<const bf16_t*>(p.dy);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.total_tiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tw_i = tile % p.tiles_w;
    const int th_i = (tile / p.tiles_w) % p.tiles_h;
    const int n_img = tile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int64_t ximg = p.xa.off(n_img), yimg = p.ya.off(n_img);
    __syncthreads();   // previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < TH * TW * (CT / 4) / 256; ++i) {     // 8 x (pixel, channel quad) per thread
      const int idx = tid + i * 256;
      const int pix = idx >> 4, cq = idx & 15;
      const int gh = h0 + (pix >> 4), gw = w0 + (pix & 15);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh < p.H && gw < p.W) v = ld4(dyb + yimg + ((int64_t)gh * p.W + gw) * p.ys + cq * 4);
      *reinterpret_cast<float4*>(&dyt[pix * CT + cq * 4]) = v;
    }
    for (int idx = tid; idx < HALO_PIX * 4; idx += 256) {
      const int pix = idx >> 2, c = idx & 3;
      const int gh = h0 - 1 + pix / HWD, gw = w0 - 1 + pix % HWD;
      float v = 0.f;
      if (c < cin && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) v = p.x[ximg + ((int64_t)gh * p.W + gw) * p.xs + c];
      xh[idx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < TH / 2; ++rr) {
      const int r = khalf * (TH / 2) + rr;
#pragma unroll
      for (int c2 = 0; c2 < TW / 2; ++c2) {
        const int col = 2 * c2 + h;
        const float b = dyt[(r * TW + col) * CT + wco * 32 + l31];
        float a = xh[(r * HWD + col) * 4 + a_off];
        a = a_on ? a : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);   // [2][16][64]
  if (khalf == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wco * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (khalf == 0) {
    float* out = p.slab + (int64_t)split * m_rows * p.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = mfma32_row(r, h);
      if (i < m_rows) out[(int64_t)i * p.Cout + wco * 32 + l31] = acc[r] + red[(wco * 16 + r) * 64 + lane];
    }
  }
}

struct Plan {
  int tiles_h, tiles_w, total_tiles, S, tiles_per_split, n_ci_tiles, n_co_tiles;
  bool small;   // first-layer kernel
};

bool plan(int N, int H, int W, int Cin, int Cout, Plan* pl) {
  pl->tiles_h = (H + TH - 1) / TH;
  pl->tiles_w = (W + TW - 1) / TW;
  pl->total_tiles = N * pl->tiles_h * pl->tiles_w;
  int S;
  if (Cin % CT == 0 && Cout % CT == 0) {
    pl->small = false;
    pl->n_ci_tiles = Cin / CT;
    pl->n_co_tiles = Cout / CT;
    const int panels = pl->n_ci_tiles * pl->n_co_tiles;
    S = (512 + panels - 1) / panels;     // one 512-thread block per CU (144 KB of LDS) x 256 CUs x 2 rounds
    if (pl->total_tiles / S < 16 && panels <= 256) S = (256 + panels - 1) / panels;
  } else if (9 * Cin <= 32 && Cout == CT) {
    pl->small = true;
    pl->n_ci_tiles = pl->n_co_tiles = 1;
    S = 2048;
  } else {
    return false;
  }
  if (S > pl->total_tiles) S = pl->total_tiles;
  if (S < 1) S = 1;
  pl->tiles_per_split = (pl->total_tiles + S - 1) / S;
  pl->S = (pl->total_tiles + pl->tiles_per_split - 1) / pl->tiles_per_split;
  return true;
}

}  // namespace

size_t unetk_wgrad_bf16s_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  Plan pl;
  if (!plan(N, H, W, Cin, Cout, &pl)) return 0;
  return 256 + (size_t)pl.S * 9 * Cin * Cout * sizeof(float);    // 256 B zero page + slabs
}

int unetk_wgrad_bf16s_run(WgParams p, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
  Plan pl;
  if (!plan(p.N, p.H, p.W, p.Cin, p.Cout, &pl)) return UNETK_E_UNSUPPORTED;
  if (ws_bytes < unetk_wgrad_bf16s_ws_bytes(p.N, p.H, p.W, p.Cin, p.Cout)) return UNETK_E_WORKSPACE;
  if (p.ys % 8 != 0 || (!pl.small && p.xs % 8 != 0)) return UNETK_E_BADARG;       // 16-byte chunks of 8 bf16
  p.zeros = (const float*)ws;
  p.slab = (float*)ws + 64;
  p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.total_tiles = pl.total_tiles;
  p.tiles_per_split = pl.tiles_per_split; p.n_ci_tiles = pl.n_ci_tiles; p.n_co_tiles = pl.n_co_tiles;
  hipError_t ez = hipMemsetAsync(ws, 0, 256, st);
  if (ez != hipSuccess) return (int)ez;
  if (pl.small) {
    const size_t lds3 = (size_t)(TH * TW * CT + HALO_PIX * 4) * sizeof(float);
    hipLaunchKernelGGL(conv3x3_wgrad_c3_bf16s_kernel, dim3(pl.S), dim3(256), lds3, st, p);
    UNETK_LAUNCH_CHECK();
  } else {
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(conv3x3_wgrad_bf16s_kernel, dim3(pl.S * pl.n_ci_tiles * pl.n_co_tiles), dim3(512), LDS_B, st, p);
    UNETK_LAUNCH_CHECK();
  }
  return unetk_launch_slab_reduce((const float*)ws + 64, pl.S, (int64_t)9 * p.Cin * p.Cout, dw, st);
}
