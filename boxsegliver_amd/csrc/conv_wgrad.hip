// conv3x3 filter gradient (Conv2DBackpropFilter of slim.conv2d(x, C, 3), NetworksV2/UNet.py:79,85,94; one
// depth tap of slim.conv3d, UNet3D.py:153,165) as a split-K GEMM on the fp32 matrix cores:
//   dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co].
//
// GEMM view: M = ci, N = co, K = pixels.  A 512-thread block owns a CIT(ci) x COT(co) x 9(taps) output panel
// (CIT, COT in {32, 64}) and walks a contiguous range of 8x16 pixel tiles.  Per tile the 10x18 x-halo and the
// 8x16 dy tile are brought in by ASYNC direct-to-LDS loads (global_load_lds_dwordx4: no VGPRs, 1 KiB per
// wave-instruction) into the OTHER half of a 2-stage LDS ring while all eight waves run MFMAs on the current
// half: one barrier per tile, global latency fully behind the MFMAs.  ALL NINE taps accumulate from the one
// staged halo (9 x 16 accumulator registers per wave); the eight waves split (pixel rows) x (ci half) x (co half)
// and the pixel-row slices are summed through LDS in a fixed-order tree at the end.  Partial panels go to a
// workspace slab per split; a fixed-order reduction sums the slabs -> bit-reproducible, no atomics.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int TW = 16, TH = 8, HWD = TW + 2, HH = TH + 2;
constexpr int HALO_PIX = HH * HWD;   // 180
constexpr size_t WG_LDS_BYTES = 157696;   // 2 stages of the 64x64 panel; also holds the 147 KB reduction scratch

// 256 bytes of zeros in global memory: the source of out-of-image halo pixels for the direct-to-LDS loads (constant data in
// the code object -- no per-call memset of a workspace page)
__device__ const float kZeroPage[64] = {};

// Pixel tile TH_ x TW_: 8 x 16 in general; 10 x 12 (STK) for small planes whose width is a multiple of 12 but not of 16
// (UNet3D's 12^2 / 24^2 levels), where 8 x 16 tiles would be 56 % / 75 % full.
// S = 2: stride-2 conv (UNet3D's down-sampling layers): the tile is TH_ x TW_ OUTPUT pixels, its input halo
// ((TH_-1) 2 + 3) x ((TW_-1) 2 + 3) pixels, and tap (kh, kw) of output pixel (r, c) reads halo pixel (2r + kh, 2c + kw).
// DIL = 2 (S = 1): rate-2 atrous conv: halo (TH_ + 4) x (TW_ + 4), tap (kh, kw) reads halo pixel (r + 2 kh, c + 2 kw).
template <int CIT, int COT, int TH_ = TH, int TW_ = TW, int S = 1, int DIL = 1>
struct WgGeom {
  static constexpr int WCI = CIT / 32, WCO = COT / 32;
  static constexpr int KS = 8 / (WCI * WCO);            // pixel-row slices
  static constexpr int RPW = TH_ / KS;                  // tile rows per wave
  static constexpr int HWD_ = (TW_ - 1) * S + 2 * DIL + 1, HALO_PIX_ = ((TH_ - 1) * S + 2 * DIL + 1) * HWD_;
  static constexpr int PPX = 256 / CIT;                 // pixels per 1-KiB piece of the x halo
  static constexpr int PPY = 256 / COT;
  static constexpr int NI_X = (HALO_PIX_ + PPX - 1) / PPX;  // 45 (CIT 64) / 23 (CIT 32, last piece half dummy)
  static constexpr int NI_Y = TH_ * TW_ / PPY;
  static_assert(TH_ % KS == 0 && (TH_ * TW_) % PPY == 0 && TW_ % 2 == 0, "tile shape");
  static constexpr int NI = NI_X + NI_Y;
  static constexpr int IPW = (NI + 7) / 8;
  static constexpr int XH_F = NI_X * 256;
  static constexpr int STAGE_F = XH_F + NI_Y * 256;
  static_assert(2 * STAGE_F * sizeof(float) <= WG_LDS_BYTES, "ring must fit");
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  uint4 q;
  q.x = pk_bf16(v[0], v[1]);
  q.y = pk_bf16(v[2], v[3]);
  q.z = pk_bf16(v[4], v[5]);
  q.w = pk_bf16(v[6], v[7]);
  return __builtin_bit_cast(bf16x8, q);
}

// BF = UNETK_BF16: same staging (fp32 tiles by direct-to-LDS loads); a k-step is one 16-pixel tile row, each lane
// gathers its 8 pixels (lane half h -> columns 8h..8h+7) of x (10 per filter row: the three kw taps share them) and
// dy from LDS, rounds them to bf16 and issues v_mfma_f32_32x32x16_bf16: 9 MFMAs of 32 cycles per 38 ds_read_b32.
//
// STK ("stacked planes", fp32 only): the planes of the batch are stacked vertically with ONE shared zero row between
// neighbours -- plane i occupies virtual rows i (H + 1) .. i (H + 1) + H - 1 -- and the tiles walk the virtual rows, so
// a tile may span two planes and no tile rows are wasted on a 12- or 24-row plane (the shared row is the bottom padding
// of one plane and the top padding of the next; dy is zero there).
template <int CIT, int COT, bool BF, int TH_ = TH, int TW_ = TW, bool STK = false, int S = 1, int DIL = 1>
__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(WgParams p) {
  using G = WgGeom<CIT, COT, TH_, TW_, S, DIL>;
  static_assert(DIL == 1 || (!BF && !STK && S == 1), "atrous tiles are plain fp32 tiles");
  constexpr int HWD_ = G::HWD_;
  static_assert(!(BF && (STK || S != 1)) && !(STK && S != 1), "stacked / strided tiles are fp32 only");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][STAGE_F]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave % G::WCO, wci = (wave / G::WCO) % G::WCI, ks = wave / (G::WCO * G::WCI);
  const int l31 = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int KD = p.kd > 1 ? p.kd : 1;                       // fused depth taps (WgParams::kd)
  const int dt = bid % KD;
  const int split = bid / KD;
  const int ci0 = ci_t * CIT, co0 = co_t * COT;
  const int64_t x_dt = KD > 1 ? (int64_t)(p.dshift0 + dt) * p.dplane : 0;

  // ---- staging geometry: wave w issues pieces j = w + 8 i; lane = (pixel of the piece, float4 of the pixel)
  const int lpx = lane / (CIT / 4), qx = lane % (CIT / 4);
  const int lpy = lane / (COT / 4), qy = lane % (COT / 4);
  int rel_h[G::IPW], rel_w[G::IPW];   // pixel position relative to the tile origin; rel_h = 1<<20 marks a dummy
#pragma unroll
  for (int i = 0; i < G::IPW; ++i) {
    const int j = wave + 8 * i;
    if (j < G::NI_X) {
      const int pix = G::PPX * j + lpx;
      rel_h[i] = pix < G::HALO_PIX_ ? pix / HWD_ : (1 << 20);   // halo coordinates; the tile origin is subtracted below
      rel_w[i] = pix % HWD_;
    } else {
      const int pix = G::PPY * (j - G::NI_X) + lpy;
      rel_h[i] = pix / TW_;
      rel_w[i] = pix % TW_;
    }
  }

  auto issue_tile = [&](int tile, int stage) {
    const int tw_i = tile % p.tiles_w;
    const int th_i = STK ? tile / p.tiles_w : (tile / p.tiles_w) % p.tiles_h;
    const int n_tile = STK ? 0 : tile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * TH_, w0 = tw_i * TW_;
    int64_t ximg = 0, yimg = 0;
    bool xplane_ok = true;                                  // fused depth taps: the tap's input plane exists
    if (!STK) {
      ximg = p.xa.off(n_tile) + x_dt;
      yimg = p.ya.off(n_tile);
      if (KD > 1) {
        const int din_i = (n_tile % p.spg) * p.dsd + p.dshift0 + dt;
        xplane_ok = din_i >= 0 && din_i < p.din;
      }
    }
#pragma unroll
    for (int i = 0; i < G::IPW; ++i) {
      const int j = wave + 8 * i;
      if (j < G::NI) {   // wave-uniform
        const bool is_x = j < G::NI_X;
        int gh = is_x ? S * h0 - p.pbh + rel_h[i] : h0 + rel_h[i];
        const int gw = is_x ? S * w0 - p.pbw + rel_w[i] : w0 + rel_w[i];
        const int ph = (S != 1 && is_x) ? p.Hin : p.H, pw = (S != 1 && is_x) ? p.Win : p.W;
        bool ok;
        if (STK) {   // virtual row -> (plane, row); row == H is the shared zero row
          // (divisions by reciprocal, common.h unetk_fdiv: v < 2^20 by the plan's guard; dummy pieces carry rel_h = 1 << 20)
          const int v = (gh < 0 || gh >= (1 << 20)) ? 0 : gh;
          const int n_img = unetk_fdiv(v, p.H + 1, p.rcp_h1);
          const int r = v - n_img * (p.H + 1);
          ok = gh >= 0 && gh < (1 << 20) && n_img < p.N && r < p.H && gw >= 0 && gw < p.W;
          gh = r;
          if (is_x) {
            const int g = unetk_fdiv(n_img, p.xa.group, p.rcp_xg);
            ximg = (int64_t)g * p.xa.group_stride + (int64_t)(n_img - g * p.xa.group) * p.xa.img_stride + x_dt;
            if (KD > 1) {
              const int din_i = (n_img - unetk_fdiv(n_img, p.spg, p.rcp_spg) * p.spg) * p.dsd + p.dshift0 + dt;
              ok = ok && din_i >= 0 && din_i < p.din;
            }
          } else {
            const int g = unetk_fdiv(n_img, p.ya.group, p.rcp_yg);
            yimg = (int64_t)g * p.ya.group_stride + (int64_t)(n_img - g * p.ya.group) * p.ya.img_stride;
          }
        } else {
          ok = gh >= 0 && gh < ph && gw >= 0 && gw < pw && (!is_x || xplane_ok);
        }
        const int64_t pixoff = (int64_t)gh * pw + gw;
        const float* src = is_x ? p.x + ximg + pixoff * p.xs + ci0 + qx * 4 : p.dy + yimg + pixoff * p.ys + co0 + qy * 4;
        if (!ok) src = kZeroPage + (lane & 15) * 4;
        float* dst = smem + stage * G::STAGE_F + j * 256;   // wave-uniform; lanes land at dst + lane*16 B
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.total_tiles);
  const int a_lane = wci * 32 + l31, b_lane = wco * 32 + l31;

  if (t_begin < t_end) issue_tile(t_begin, 0);
  int stage = 0;
  for (int tile = t_begin; tile < t_end; ++tile, stage ^= 1) {
    // every wave: its own pieces of this tile have landed; barrier: everybody's have, and all reads of the
    // other stage (previous tile) are done, so it may be refilled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The next tile's loads (a few hundred address instructions per wave) are issued by the even waves now and by the odd
    // waves half-way through their rows, from ONE call site inside the row loop: 131.3 -> 133.8 TF.  (A second, inlined
    // call site ahead of the loop cost 7 % by itself -- measured with the stagger switched off.)
    const int issue_rr = ((wave & 1) != 0 && G::RPW > 1) ? G::RPW / 2 : 0;

    const float* xh = smem + stage * G::STAGE_F;
    const float* dyt = xh + G::XH_F;
    // this wave's k-steps (rows RPW*ks .. +RPW); lane half h takes the odd/even column of a pixel pair.  (Measured and
    // dropped: two rows per iteration with the shared x rows carried in registers and every LDS read requested a filter row
    // ahead -- 250 VGPRs, 131 instead of 133 TF: the LDS latency at the top of a row is not what this kernel waits for.)
#pragma unroll 1
    for (int rr = 0; rr < G::RPW; ++rr) {
      if (rr == issue_rr && tile + 1 < t_end) issue_tile(tile + 1, stage ^ 1);
      const int r = ks * G::RPW + rr;
      if constexpr (BF) {
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = dyt[(r * TW + 8 * h + j) * COT + b_lane];
        const bf16x8 b = pack8(bv);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          float av[10];
#pragma unroll
          for (int j = 0; j < 10; ++j) av[j] = xh[((r + kh) * HWD + 8 * h + j) * CIT + a_lane];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(av + kw), b, acc[kh * 3 + kw], 0, 0, 0);
        }
        continue;
      }
      if constexpr (S == 1 && DIL == 1) {
        // lane half h takes the columns h TW/2 .. h TW/2 + TW/2 - 1 of the row: the three kw taps of consecutive columns
        // share their x values, so a filter row needs TW/2 + 2 LDS reads per lane for 3 TW/2 MFMAs (the even / odd column
        // split below reads one value per MFMA: 1.1 LDS instructions per MFMA against 0.3 here)
        constexpr int TWH = TW_ / 2;
        float bv[TWH];
#pragma unroll
        for (int j = 0; j < TWH; ++j) bv[j] = dyt[(r * TW_ + h * TWH + j) * COT + b_lane];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          float av[TWH + 2];
#pragma unroll
          for (int j = 0; j < TWH + 2; ++j) av[j] = xh[((r + kh) * HWD_ + h * TWH + j) * CIT + a_lane];
#pragma unroll
          for (int c2 = 0; c2 < TWH; ++c2)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c2 + kw], bv[c2], acc[kh * 3 + kw], 0, 0, 0);
        }
        continue;
      }
#pragma unroll
      for (int c2 = 0; c2 < TW_ / 2; ++c2) {
        const int col = 2 * c2 + h;
        const float b = dyt[(r * TW_ + col) * COT + b_lane];
        const float* xa = &xh[(S * r * HWD_ + S * col) * CIT + a_lane];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float a = xa[(kh * DIL * HWD_ + kw * DIL) * CIT];
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kh * 3 + kw], 0, 0, 0);
          }
      }
    }
  }

  // ---- sum the pixel-row slices through LDS: fixed-order binary tree over ks, <= 4 writer waves per round
  float* red = smem;  // [<=4 waves][144][64 lanes] = 147456 B
  const int pidx = wci * G::WCO + wco;
#pragma unroll
  for (int step = G::KS / 2; step >= 1; step >>= 1) {
    __syncthreads();
    if (ks >= step && ks < 2 * step) {
      const int slot = (ks - step) * (G::WCI * G::WCO) + pidx;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(slot * 144 + t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (ks < step) {
      const int slot = ks * (G::WCI * G::WCO) + pidx;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += red[(slot * 144 + t * 16 + r) * 64 + lane];
    }
  }
  if (ks == 0) {
    float* out = p.slab + ((int64_t)split * KD + dt) * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wci * 32 + mfma32_row(r, h);
        out[((int64_t)t * p.Cin + ci) * p.Cout + co0 + b_lane] = acc[t][r];
      }
  }
}

// Small-Cin filter gradient for Cout = 64 and 9*Cin <= 32 (Encode1/conv1: Cin = 3; UNet3D conv_e0/conv1:
// Cin = 1).  HBM-bound on reading dy once (537 MB at cfg1); the 27 x 64 contraction over pixels still goes
// through the matrix cores (M = (tap, ci) padded to 32, N = co, K = pixels) because the scalar version is
// VALU-issue bound (27 FMAs + 27 loads per pixel per lane), ~15x off the HBM roofline.
// 4 waves = (pixel-row half) x (co half); dy tile by direct-to-LDS loads, x halo (Cin floats / pixel).
template <int COT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_c3_kernel(WgParams p) {
  // COT = 64: waves = (pixel-row half) x (co half); COT = 32 (UNet3D conv_e0/conv1, 30 channels padded): four row quarters
  constexpr int NCO = COT / 32, KS = 4 / NCO, RPW = TH / KS, PP = 256 / COT, NPIECE = TH * TW / PP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // two buffers of { dy tile [128 pixels][COT] (NPIECE pieces of 1 KiB), x halo [180 pixels][4] }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = wave / NCO, wco = wave % NCO;
  const int l31 = lane & 31, h = lane >> 5;
  const int cin = p.Cin, m_rows = 9 * cin;
  const int split = blockIdx.x;
  // A-operand gather: lane row i = (tap, ci)
  const bool a_on = l31 < m_rows;
  const int a_tap = a_on ? l31 / cin : 0, a_ci = a_on ? l31 % cin : 0;
  const int a_off = ((a_tap / 3) * HWD + (a_tap % 3)) * 4 + a_ci;
  const int lp = lane / (COT / 4), q = lane % (COT / 4);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.total_tiles);
  // Two tile buffers (round 3): the next tile's dy streams into the other buffer (direct-to-LDS) and its halo values wait in
  // registers while this tile is contracted -- the loop used to be load -> wait -> barrier -> MFMAs with nothing in flight
  // during the MFMAs.
  constexpr int BUF_F = TH * TW * COT + HALO_PIX * 4;          // floats per buffer: dy tile + x halo
  constexpr int NXH = (HALO_PIX * 4 + 255) / 256;
  float xr[NXH];
  auto issue = [&](int tile, int buf) {
    const int tw_i = tile % p.tiles_w;
    const int th_i = (tile / p.tiles_w) % p.tiles_h;
    const int n_img = tile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int64_t ximg = p.xa.off(n_img), yimg = p.ya.off(n_img);
    float* dst = smem + buf * BUF_F;
#pragma unroll
    for (int i = 0; i < NPIECE / 4; ++i) {          // dy: piece j = wave + 4 i covers pixels PP j .. PP j + PP - 1
      const int j = wave + 4 * i;
      const int pix = PP * j + lp;
      const int gh = h0 + (pix >> 4), gw = w0 + (pix & 15);
      const float* src = (gh < p.H && gw < p.W) ? p.dy + yimg + ((int64_t)gh * p.W + gw) * p.ys + q * 4
                                                : kZeroPage + (q & 15) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NXH; ++i) {
      const int idx = tid + i * 256;
      const int pix = idx >> 2, c = idx & 3;
      const int gh = h0 - 1 + pix / HWD, gw = w0 - 1 + pix % HWD;
      xr[i] = 0.f;
      if (idx < HALO_PIX * 4 && c < cin && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W)
        xr[i] = p.x[ximg + ((int64_t)gh * p.W + gw) * p.xs + c];
    }
  };
  if (t_begin < t_end) issue(t_begin, 0);
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile, buf ^= 1) {
    const float* dyt = smem + buf * BUF_F;
    float* xh = smem + buf * BUF_F + TH * TW * COT;
#pragma unroll
    for (int i = 0; i < NXH; ++i) {                  // (the compiler waits for the halo loads here)
      const int idx = tid + i * 256;
      if (idx < HALO_PIX * 4) xh[idx] = xr[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's dy has landed (nothing younger is in flight)
    __syncthreads();                                   // ... everybody's; and the other buffer's readers are done
    if (tile + 1 < t_end) issue(tile + 1, buf ^ 1);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = kq * RPW + rr;
#pragma unroll
      for (int c2 = 0; c2 < TW / 2; ++c2) {
        const int col = 2 * c2 + h;
        const float b = dyt[(r * TW + col) * COT + wco * 32 + l31];
        float a = xh[(r * HWD + col) * 4 + a_off];
        a = a_on ? a : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
  }
  __syncthreads();
  float* red = smem;   // [KS - 1][NCO][16][64]
  if (kq > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(((kq - 1) * NCO + wco) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (kq == 0) {
    float* out = p.slab + (int64_t)split * m_rows * p.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = mfma32_row(r, h);
      float v = acc[r];
#pragma unroll
      for (int k = 0; k < KS - 1; ++k) v += red[((k * NCO + wco) * 16 + r) * 64 + lane];   // fixed order
      if (i < m_rows) out[(int64_t)i * p.Cout + wco * 32 + l31] = v;
    }
  }
}

// Generic small-Cin filter gradient (fallback for Cin = 4 or Cout != 64): thread = (co, pixel lane).
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
      const int64_t ximg = p.xa.off(n_img), yimg = p.ya.off(n_img);
      for (int pix = pl; pix < TH * TW; pix += PL) {
        const int gh = h0 + (pix >> 4), gw = w0 + (pix & 15);
        if (gh >= p.H || gw >= p.W) continue;
        const float g = p.dy[yimg + ((int64_t)gh * p.W + gw) * p.ys + co];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = gh + kh - 1;
          if (ih < 0 || ih >= p.H) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = gw + kw - 1;
            if (iw < 0 || iw >= p.W) continue;
            const float* xp = p.x + ximg + ((int64_t)ih * p.W + iw) * p.xs;
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

// dst[i] = sum_s slab[s][i] in a fixed order, ONE launch for any S (n % 4 == 0): a block is (256 / RL) float4 columns x
// RL row lanes; row lane r sums slabs r, r + RL, ... and the RL partial sums are added in lane order through LDS.
template <int RL>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int S, int64_t n,
                                                          float* __restrict__ dst) {
  constexpr int CL = 256 / RL;
  __shared__ float4 red[RL > 1 ? RL : 1][CL];
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int64_t n4 = n >> 2;
  const int64_t i = (int64_t)blockIdx.x * CL + cl;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int s = rl; s < S; s += RL) {
      const float4 v = ldg4(slab + (int64_t)s * n + i * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  if (RL > 1) {
    red[rl][cl] = a;
    __syncthreads();
    if (rl != 0 || i >= n4) return;
#pragma unroll
    for (int r = 1; r < RL; ++r) {
      const float4 v = red[r][cl];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  } else if (i >= n4) {
    return;
  }
  stg4(dst + i * 4, a);
}

int unetk_launch_slab_reduce(const float* slab, int S, int64_t n, float* dst, hipStream_t st) {
  const int64_t n4 = n >> 2;
  if (S >= 64) {
    UNETK_LAUNCH(slab_reduce_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, slab, S, n, dst);
  } else if (S >= 8) {
    UNETK_LAUNCH(slab_reduce_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, slab, S, n, dst);
  } else {
    UNETK_LAUNCH(slab_reduce_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, slab, S, n, dst);
  }
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

namespace {

constexpr int STH = 10, STW = 12;   // stacked-plane tile

struct WgPlan {
  int mode;  // 0 mfma, 1 small-Cin, 2 mfma on stacked planes, -1 unsupported
  int cit, cot;
  int tiles_h, tiles_w, total_tiles, S, tiles_per_split, n_ci_tiles, n_co_tiles;
};

constexpr int S2TH = 4;             // stride-2 tile: 4 x 16 (or 4 x 12) output pixels, 32-channel ci panels

bool wg_strided_ok(int Cin, int Cout) { return Cin % 32 == 0 && Cout % 64 == 0; }

int wg_s2w6() {        // UNETK_WG_S2W6=0: measurement switch for the 6-wide stride-2 tile (read once)
  static int v = -1;
  if (v < 0) { const char* e = getenv("UNETK_WG_S2W6"); v = e ? atoi(e) : 1; }
  return v;
}

WgPlan wg_plan_strided(int N, int H, int W, int Cin, int Cout, int kd = 1) {   // H, W = output plane
  WgPlan pl{};
  // 6-wide output planes (UNet3D's (2,2,2) bridge, round 5): 4 x 6 tiles -- a 4 x 16 tile is 37 % full there (25 TFLOP/s)
  const int tw = (W % TW != 0 && W % STW == 0) ? STW : ((W == 6 && wg_s2w6()) ? 6 : TW);
  pl.mode = tw == STW ? 4 : (tw == 6 ? 6 : 3);
  pl.tiles_h = (H + S2TH - 1) / S2TH;
  pl.tiles_w = (W + tw - 1) / tw;
  pl.total_tiles = N * pl.tiles_h * pl.tiles_w;
  pl.cit = 32; pl.cot = 64;
  pl.n_ci_tiles = Cin / 32;
  pl.n_co_tiles = Cout / 64;
  const int panels = pl.n_ci_tiles * pl.n_co_tiles * (kd > 1 ? kd : 1);      // fused depth taps: kd x the blocks per split
  int S = panels >= 512 ? 1 : 512 / panels;   // floor: 513 blocks would be three rounds of the one-block-per-CU kernel
  if (S > pl.total_tiles) S = pl.total_tiles;
  if (S < 1) S = 1;
  pl.tiles_per_split = (pl.total_tiles + S - 1) / S;
  pl.S = (pl.total_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
  return pl;
}

WgPlan wg_plan(int N, int H, int W, int Cin, int Cout, bool bf16 = false, int kd = 1) {
  WgPlan pl{};
  pl.tiles_h = (H + TH - 1) / TH;
  pl.tiles_w = (W + TW - 1) / TW;
  pl.total_tiles = N * pl.tiles_h * pl.tiles_w;
  // small planes whose width 8 x 16 tiles fill badly: 10 x 12 tiles over the stacked planes (fp32, 64 x 64 panels)
  const bool stk_ok = !bf16 && Cin % 64 == 0 && Cout % 64 == 0 && W % TW != 0 && H <= 64 && (int64_t)N * (H + 1) < (1 << 20);
  const bool stacked12 = stk_ok && W % STW == 0;
  const bool stacked6 = stk_ok && !stacked12 && W % 6 == 0;       // 20 x 6 tiles: UNet3D's 6 x 6 bridge planes
  const bool stacked = stacked12 || stacked6;
  if (stacked) {
    const int sth = stacked12 ? STH : 20, stw = stacked12 ? STW : 6;
    pl.tiles_w = W / stw;
    pl.tiles_h = (int)(((int64_t)N * (H + 1) + sth - 1) / sth);   // over ALL planes
    pl.total_tiles = pl.tiles_h * pl.tiles_w;
  }
  if (Cin % 32 == 0 && Cout % 32 == 0) {
    pl.mode = stacked12 ? 2 : (stacked6 ? 5 : 0);
    pl.cit = Cin % 64 == 0 ? 64 : 32;
    pl.cot = Cout % 64 == 0 ? 64 : 32;
    pl.n_ci_tiles = Cin / pl.cit;
    pl.n_co_tiles = Cout / pl.cot;
    const int panels = pl.n_ci_tiles * pl.n_co_tiles * (kd > 1 ? kd : 1);
    int S = panels >= 512 ? 1 : 512 / panels;   // one 512-thread block per CU (154 KB of LDS) x 256 CUs x 2 rounds; floor: 513 blocks would be three
    // few tiles per block (small layers): one round of 256 blocks halves the per-block epilogue (LDS tree + slab) per tile
    // (floor, not ceil: 154 KB of LDS = one block per CU, so 257 blocks take two rounds)
    if (pl.total_tiles / S < 16 && panels <= 256) S = 256 / panels;
    if (S > pl.total_tiles) S = pl.total_tiles;
    if (S < 1) S = 1;
    pl.tiles_per_split = (pl.total_tiles + S - 1) / S;
    pl.S = (pl.total_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
  } else if ((Cin <= 5 || Cin == 9) && Cout <= 256 && 256 % Cout == 0) {  // 9 = 3 channels x (image, dy, dx): --img_grad; 5 = image + Sobel
    pl.mode = 1;
    int S = 512;      // = the resident blocks (two 71 KB blocks per CU): measured 2048 / 1024 / 512 splits -> 0.201 / 0.173 / 0.153 ms
    if (S > pl.total_tiles) S = pl.total_tiles;
    pl.tiles_per_split = (pl.total_tiles + S - 1) / S;
    pl.S = (pl.total_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
  } else {
    pl.mode = -1;
  }
  return pl;
}

template <int CIT, int COT, bool BF, int TH_ = TH, int TW_ = TW, bool STK = false, int S = 1, int DIL = 1>
int launch_wgrad(const WgParams& p, int grid, hipStream_t st) {
  auto kern = conv3x3_wgrad_kernel<CIT, COT, BF, TH_, TW_, STK, S, DIL>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  UNETK_LAUNCH(kern, dim3(grid), dim3(512), WG_LDS_BYTES, st, p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

}  // namespace

size_t unetk_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int kd) {
  if (kd > 1) {
    const WgPlan pk = wg_plan(N, H, W, Cin, Cout, false, kd);
    if (pk.mode != 0 && pk.mode != 2 && pk.mode != 5) return 0;
    const size_t fused = 256 + (size_t)pk.S * kd * 9 * Cin * Cout * sizeof(float);
    const size_t plain = unetk_wgrad_ws_bytes(N, H, W, Cin, Cout, 1);
    return fused > plain ? fused : plain;
  }
  const WgPlan pl = wg_plan(N, H, W, Cin, Cout, false), pb = wg_plan(N, H, W, Cin, Cout, true);
  if (pl.mode < 0) return 0;
  int S = pl.S > pb.S ? pl.S : pb.S;                           // either precision
  if (Cin % 64 == 0 && Cout % 64 == 0) {                       // ... or the atrous variant (6-row tiles: more tiles to split)
    const int panels = (Cin / 64) * (Cout / 64);
    int sd = (512 + panels - 1) / panels;
    const int64_t tiles = (int64_t)N * ((H + 5) / 6) * ((W + TW - 1) / TW);
    if (sd > tiles) sd = (int)tiles;
    if (sd > S) S = sd;
  }
  return 256 + (size_t)S * 9 * Cin * Cout * sizeof(float);    // 256 B zero page + slabs
}

bool unetk_wgrad_strided_ok(int Cin, int Cout) { return wg_strided_ok(Cin, Cout); }

size_t unetk_wgrad_strided_ws_bytes(int N, int Ho, int Wo, int Cin, int Cout, int kd) {
  if (!wg_strided_ok(Cin, Cout)) return 0;
  const WgPlan pl = wg_plan_strided(N, Ho, Wo, Cin, Cout, kd);
  return 256 + (size_t)pl.S * (kd > 1 ? kd : 1) * 9 * Cin * Cout * sizeof(float);
}

int unetk_wgrad_run(WgParams p, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
  p.rcp_h1 = 1.0f / (float)(p.H + 1);
  p.rcp_xg = 1.0f / (float)(p.xa.group > 0 ? p.xa.group : 1);
  p.rcp_yg = 1.0f / (float)(p.ya.group > 0 ? p.ya.group : 1);
  p.rcp_spg = 1.0f / (float)(p.spg > 0 ? p.spg : 1);
  if (p.stride == 2) {   // fp32 only: H, W = output plane
    if (!wg_strided_ok(p.Cin, p.Cout) || p.bf16) return UNETK_E_UNSUPPORTED;
    if (p.xs % 4 != 0 || p.ys % 4 != 0 || p.pbh < 0 || p.pbh > 1 || p.pbw < 0 || p.pbw > 1) return UNETK_E_BADARG;
    if (p.Hin < 2 * p.H - 1 || p.Hin > 2 * p.H || p.Win < 2 * p.W - 1 || p.Win > 2 * p.W) return UNETK_E_BADARG;
    const int KD = p.kd > 1 ? p.kd : 1;
    const WgPlan pl = wg_plan_strided(p.N, p.H, p.W, p.Cin, p.Cout, KD);
    if (ws_bytes < unetk_wgrad_strided_ws_bytes(p.N, p.H, p.W, p.Cin, p.Cout, KD)) return UNETK_E_WORKSPACE;
    p.slab = pl.S == 1 ? dw : (float*)ws + 64;           // a single split writes the gradient in place
    p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.total_tiles = pl.total_tiles;
    p.tiles_per_split = pl.tiles_per_split; p.n_ci_tiles = pl.n_ci_tiles; p.n_co_tiles = pl.n_co_tiles;
    const int grid = pl.S * KD * pl.n_ci_tiles * pl.n_co_tiles;
    const int rc = pl.mode == 4 ? launch_wgrad<32, 64, false, S2TH, STW, false, 2>(p, grid, st)
                 : pl.mode == 6 ? launch_wgrad<32, 64, false, S2TH, 6, false, 2>(p, grid, st)
                                : launch_wgrad<32, 64, false, S2TH, TW, false, 2>(p, grid, st);
    if (rc != UNETK_OK) return rc;
    if (pl.S == 1) return UNETK_OK;
    return unetk_launch_slab_reduce((const float*)ws + 64, pl.S, (int64_t)KD * 9 * p.Cin * p.Cout, dw, st);
  }
  if (p.dil == 2) {      // atrous: 6 x 16 tiles (the 10 x 20 halo of two stages fits LDS), 64 x 64 panels, fp32
    if (p.bf16 || p.Cin % 64 != 0 || p.Cout % 64 != 0 || p.kd > 1) return UNETK_E_UNSUPPORTED;
    if (p.xs % 4 != 0 || p.ys % 4 != 0) return UNETK_E_BADARG;
    p.stride = 1; p.Hin = p.H; p.Win = p.W; p.pbh = p.pbw = 2;
    WgPlan pl{};
    pl.tiles_h = (p.H + 5) / 6;
    pl.tiles_w = (p.W + TW - 1) / TW;
    pl.total_tiles = p.N * pl.tiles_h * pl.tiles_w;
    pl.n_ci_tiles = p.Cin / 64; pl.n_co_tiles = p.Cout / 64;
    const int panels = pl.n_ci_tiles * pl.n_co_tiles;
    int S = panels >= 512 ? 1 : 512 / panels;   // floor: 513 blocks would be three rounds of the one-block-per-CU kernel
    if (S > pl.total_tiles) S = pl.total_tiles;
    if (S < 1) S = 1;
    pl.tiles_per_split = (pl.total_tiles + S - 1) / S;
    pl.S = (pl.total_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    if (ws_bytes < 256 + (size_t)pl.S * 9 * p.Cin * p.Cout * sizeof(float)) return UNETK_E_WORKSPACE;
    p.slab = pl.S == 1 ? dw : (float*)ws + 64;           // a single split writes the gradient in place
    p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.total_tiles = pl.total_tiles;
    p.tiles_per_split = pl.tiles_per_split; p.n_ci_tiles = pl.n_ci_tiles; p.n_co_tiles = pl.n_co_tiles;
    const int rc = launch_wgrad<64, 64, false, 6, TW, false, 1, 2>(p, pl.S * panels, st);
    if (rc != UNETK_OK) return rc;
    if (pl.S == 1) return UNETK_OK;
    return unetk_launch_slab_reduce((const float*)ws + 64, pl.S, (int64_t)9 * p.Cin * p.Cout, dw, st);
  }
  p.stride = 1; p.Hin = p.H; p.Win = p.W; p.pbh = p.pbw = 1;
  const int KD = p.kd > 1 ? p.kd : 1;
  const WgPlan pl = wg_plan(p.N, p.H, p.W, p.Cin, p.Cout, p.bf16 != 0, KD);
  if (pl.mode < 0) return UNETK_E_UNSUPPORTED;
  if (KD > 1 && (p.bf16 || (pl.mode != 0 && pl.mode != 2 && pl.mode != 5))) return UNETK_E_UNSUPPORTED;   // MFMA fp32 kernels only
  if (ws_bytes < unetk_wgrad_ws_bytes(p.N, p.H, p.W, p.Cin, p.Cout, KD)) return UNETK_E_WORKSPACE;
  p.slab = pl.S == 1 ? dw : (float*)ws + 64;           // a single split writes the gradient in place
  p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.total_tiles = pl.total_tiles;
  p.tiles_per_split = pl.tiles_per_split; p.n_ci_tiles = pl.n_ci_tiles; p.n_co_tiles = pl.n_co_tiles;
  int rc = UNETK_OK;
  if (pl.mode == 2 || pl.mode == 5) {
    if (p.xs % 4 != 0 || p.ys % 4 != 0) return UNETK_E_BADARG;
    const int grid = pl.S * KD * pl.n_ci_tiles * pl.n_co_tiles;
    rc = pl.mode == 2 ? launch_wgrad<64, 64, false, STH, STW, true>(p, grid, st)
                      : launch_wgrad<64, 64, false, 20, 6, true>(p, grid, st);
    if (rc != UNETK_OK) return rc;
  } else if (pl.mode == 0) {
    if (p.xs % 4 != 0 || p.ys % 4 != 0) return UNETK_E_BADARG;
    const int grid = pl.S * KD * pl.n_ci_tiles * pl.n_co_tiles;
    if (p.bf16) {
      if (pl.cit == 64 && pl.cot == 64) rc = launch_wgrad<64, 64, true>(p, grid, st);
      else if (pl.cit == 64) rc = launch_wgrad<64, 32, true>(p, grid, st);
      else if (pl.cot == 64) rc = launch_wgrad<32, 64, true>(p, grid, st);
      else rc = launch_wgrad<32, 32, true>(p, grid, st);
    } else if (pl.cit == 64 && pl.cot == 64) rc = launch_wgrad<64, 64, false>(p, grid, st);
    else if (pl.cit == 64) rc = launch_wgrad<64, 32, false>(p, grid, st);
    else if (pl.cot == 64) rc = launch_wgrad<32, 64, false>(p, grid, st);
    else rc = launch_wgrad<32, 32, false>(p, grid, st);
    if (rc != UNETK_OK) return rc;
  } else if (9 * p.Cin <= 32 && (p.Cout == 64 || p.Cout == 32) && p.ys % 4 == 0) {
    const size_t lds3 = (size_t)2 * (TH * TW * p.Cout + HALO_PIX * 4) * sizeof(float);      // two tile buffers: 71 KB at Cout = 64
    static bool attr3 = false;
    if (!attr3) {
      hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_c3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         2 * (TH * TW * 64 + HALO_PIX * 4) * (int)sizeof(float));
      if (e != hipSuccess) return (int)e;
      attr3 = true;
    }
    if (p.Cout == 64) UNETK_LAUNCH(conv3x3_wgrad_c3_kernel<64>, dim3(pl.S), dim3(256), lds3, st, p);
    else UNETK_LAUNCH(conv3x3_wgrad_c3_kernel<32>, dim3(pl.S), dim3(256), lds3, st, p);
    UNETK_LAUNCH_CHECK();
  } else {
    const int PL = 256 / p.Cout;
    const size_t lds = (size_t)PL * 9 * p.Cin * p.Cout * sizeof(float);
    if (lds > (p.Cin == 9 ? 150 : 64) * 1024) return UNETK_E_UNSUPPORTED;
    switch (p.Cin) {
      case 1: UNETK_LAUNCH(conv3x3_wgrad_smallc_kernel<1>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 2: UNETK_LAUNCH(conv3x3_wgrad_smallc_kernel<2>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 3: UNETK_LAUNCH(conv3x3_wgrad_smallc_kernel<3>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 4: UNETK_LAUNCH(conv3x3_wgrad_smallc_kernel<4>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 5: UNETK_LAUNCH(conv3x3_wgrad_smallc_kernel<5>, dim3(pl.S), dim3(256), lds, st, p); break;
      case 9: {
        static bool attr_done = false;
        if (!attr_done) {
          hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wgrad_smallc_kernel<9>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
          if (e != hipSuccess) return (int)e;
          attr_done = true;
        }
        UNETK_LAUNCH(conv3x3_wgrad_smallc_kernel<9>, dim3(pl.S), dim3(256), lds, st, p);
        break;
      }
      default: return UNETK_E_UNSUPPORTED;
    }
    UNETK_LAUNCH_CHECK();
  }
  if (pl.S == 1) return UNETK_OK;
  return unetk_launch_slab_reduce((const float*)ws + 64, pl.S, (int64_t)KD * 9 * p.Cin * p.Cout, dw, st);
}

extern "C" size_t unetk_conv3x3_wgrad_ws_bytes(const unetk_conv_desc* d) {
  if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
  if (d->precision == UNETK_BF16S) return unetk_wgrad_bf16s_ws_bytes(d->N, d->H, d->W, d->Cin, d->Cout);
  return unetk_wgrad_ws_bytes(d->N, d->H, d->W, d->Cin, d->Cout);
}

extern "C" int unetk_conv3x3_wgrad(const unetk_conv_desc* d, const void* xv, const void* dyv, float* dw,
                                   void* ws, size_t ws_bytes, void* stream) {
  const float* x = (const float*)xv;
  const float* dy = (const float*)dyv;
  UNETK_REQUIRE(d && x && dy && dw && ws);
  UNETK_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0);
  UNETK_REQUIRE(d->x_stride >= d->Cin && d->y_stride >= d->Cout);
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(dy) && unetk_aligned16(dw) && unetk_aligned16(ws));
  if (d->precision == UNETK_BF16S) {        // x and dy are bf16 (x fp32 for the first layer); stride 1, dense taps only
    if (d->dilation > 1) return UNETK_E_UNSUPPORTED;
    WgParams q{};
    q.x = x; q.dy = dy;
    q.N = d->N; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.Cout = d->Cout; q.xs = d->x_stride; q.ys = d->y_stride;
    q.xa = unetk_dense_addr(q.H, q.W, q.xs);
    q.ya = unetk_dense_addr(q.H, q.W, q.ys);
    return unetk_wgrad_bf16s_run(q, dw, ws, ws_bytes, (hipStream_t)stream);
  }
  WgParams p{};
  p.x = x; p.dy = dy;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.xs = d->x_stride; p.ys = d->y_stride;
  p.bf16 = d->precision == UNETK_BF16;     // small-Cin layers (first conv) ignore it: they run the fp32 kernels
  p.dil = d->dilation;
  p.xa = unetk_dense_addr(p.H, p.W, p.xs);
  p.ya = unetk_dense_addr(p.H, p.W, p.ys);
  return unetk_wgrad_run(p, dw, ws, ws_bytes, (hipStream_t)stream);
}
