// conv3x3 (stride 1, SAME) forward / input-gradient as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), NHWC, for gfx950.
//
// Replaces the TF op behind slim.conv2d(x, C, 3) at NetworksV2/UNet.py:79,85,94 (forward) and its
// Conv2DBackpropInput (dgrad = the same kernel on flipped/transposed filters).
//
// GEMM view: M = output pixels (a TH x 16 spatial tile of one image per block), N = Cout,
// K = 9 taps x Cin.  Per 16-channel chunk the (TH+2) x 18 input halo is staged ONCE in LDS and all
// nine taps read their shifted A fragments from it (9x fewer global reads than im2col); the
// per-(chunk, tap) [16 x BN] filter panel is staged K4-interleaved so both A and B fragments are
// single ds_read_b128.  Global->LDS staging is register-prefetched one step ahead (issue before the
// MFMA block, write after it), LDS double-buffered, one barrier per step.
//
// Algorithmic bytes per block-step are tiny next to the 64-cycle fp32 MFMA, so the kernel is
// MFMA-issue bound by construction; the roofline that bounds it is the fp32 matrix peak.
//
// Where the last 10 % go (round 2, PMC SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, profiles/r02_pmc_mfma_busy.txt): the
// clock holds ~2.38 GHz under this kernel, the MFMA pipe is 87-89 % busy.  With the in-loop global loads and LDS writes
// compiled out (results wrong, timing only) the same launches run at 148-151 TF = 95-96 %, which is also what the
// stand-alone loop of tools/mfma_mix.hip reaches with a barrier per 32 MFMAs: the staging instructions themselves, not
// their latency, take MFMA issue time (s_setprio around the MFMA block, either way: no change).  Hence the 16 x 16 pixel
// tiles below -- fewer staged bytes per MFMA -- wherever the grid still gives two blocks per CU.  (Also measured, no gain: an
// epilogue fast path for interior tiles -- wave-uniform base, scalar row offsets, no per-row bounds branch, a quarter of
// the ~600 instructions per wave and tile -- 415.5 vs 416.7 slices/s: the other waves of the SIMD already cover it.)
#include "common.h"

#include <type_traits>
#include "pack.h"

namespace {

__device__ float kZeroF4[4] = {};        // NOT const: a const array lives in the constant address space and selecting between it and a global pointer makes the load a FLAT load (lgkmcnt: every LDS read then waits for it)      // what an out-of-image halo pixel reads (see load_halo)

constexpr int CK = 16;   // input channels per K-chunk
// tap steps at which the next chunk's halo is requested from HBM / written to LDS (registers in between)
#ifndef HALO_LD_T
#define HALO_LD_T 2
#endif
#ifndef HALO_ST_T
#define HALO_ST_T 7
#endif
constexpr int PS = 20;   // LDS pixel stride (floats): 16 + 4 pad -> conflict-free ds_read_b128
constexpr int TW = 16;   // spatial tile width

// S = 1: the SAME stride-1 conv (p.H x p.W = input = output extent, pad 1).  S = 2: TF SAME stride-2 conv (UNet3D's
// (1,2,2) / (2,2,2) layers, UNet3D.py:31-91): tiles walk the p.H x p.W OUTPUT, the staged halo is the
// (2 TH + 2) x 34 input window starting at 2 h0 - pbh (pad-before 0 on even, 1 on odd input extents), and pixel (r, c)
// reads tap (kh, kw) at halo (2 r + kh, 2 c + kw): same inner loop, A reads 2-way bank-conflicted (LDS has the slack).
// DIL = 2 (with S = 1): rate-2 atrous conv (slim.conv2d(x, C, 3, rate=2): SmallUNet.py:44-49 bridge / conv_d3): the staged
// halo grows to (TH + 4) x (TW + 4) and tap (kh, kw) reads it at (2 kh, 2 kw); everything else is unchanged.
// MODE 2 (NBR): the epilogue also emits the norm-backward reduction of the unit that produced the input (ConvParams::ny).  A
// separate instantiation, not a run-time branch: the branch kept 70 more VGPRs live in EVERY launch (114 -> 187 for the
// 128 x 128 tile, 81 -> 152 for the 256 x 64 one, whose occupancy halved: +9 % on the 64-channel layers).
// MODE 1 = the epilogue accumulates (y += acc: depth taps of a 3-D conv) -- also its own instantiation: the sixteen old
// values per fragment it keeps in flight are 32 registers the plain kernel does not need (114 -> 71 VGPRs, i.e. three
// waves per SIMD instead of two next to the 64 accumulator registers).
// MODE 3 / 4 (inference, unetk_conv3x3_fwd_affine): the epilogue applies the normaliser's (scale, shift) + ReLU to the
// accumulators and stores the ACTIVATION (no raw output, no statistics, no second pass); MODE 4 also stores max_pool2d(z, 2, 2):
// a 32-row fragment is two tile rows x 16 columns and a lane holds columns {0-3, 8-11} + 4 h of BOTH rows, i.e. four whole
// 2 x 2 windows -- the pool is four register maxima, no shuffles.
// LDS row stride of the halo in floats (round 5).  A `ds_read_b128` is served in four fixed 16-lane groups ({0-3,12-15,20-27},
// {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md LDS): an A fragment's group therefore reads columns {0-3,12-15} of one tile row and
// {4-11} of the next, and its sixteen 4-bank windows 20 c mod 64 (PS = 20) only tile the 64 banks when the row stride is a multiple
// of 64 floats.  18 x 20 = 360 is not: two 2-way conflicts per group, 35 % of this kernel's LDS cycles (SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE, profiles/r05_pmc_lds_*.txt).  Plain 3x3 tiles pad each halo row to 384 floats (+ 1.7 KB per buffer).
#ifndef UNETK_HALO_ROWPAD
#define UNETK_HALO_ROWPAD 1
#endif
constexpr int halo_row_f(int S, int DIL) {
  return (S == 1 && DIL == 1 && UNETK_HALO_ROWPAD) ? ((S * TW + 2 * DIL) * PS + 63) / 64 * 64 : (S * TW + 2 * DIL) * PS;
}

template <int WM, int WN, int TM, int TN, int S = 1, int DIL = 1, int MODE = 0>
__global__ __launch_bounds__(WM* WN * 64, (TM * TN == 8 ? 2 : 1)) void conv3x3_igemm_kernel(ConvParams p) {
  constexpr bool NBR = MODE == 2, ACC = MODE == 1, AFF = MODE >= 3, POOL = MODE == 4;
  static_assert(!AFF || (S == 1 && DIL == 1), "the fused inference epilogue is for plain 3x3 convs");
  static_assert(S == 1 || DIL == 1, "strided atrous convs are not needed");
  static_assert(!NBR || (S == 1 && DIL == 1), "the fused reduction is for plain 3x3 convs");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int TH = BM / TW, HH = S * TH + 2 * DIL;
  constexpr int HWD = S * TW + 2 * DIL;
  constexpr int HALO_PIX = HH * HWD;
  constexpr int HROW = halo_row_f(S, DIL);           // floats between the LDS images of two halo rows
  constexpr int HALO_F = HH * HROW;
  constexpr int WB_F = CK * BN;
  constexpr int HR = (HALO_PIX * 4 + NT - 1) / NT;   // float4 halo loads per thread
  constexpr int WR = (CK / 4 * BN + NT - 1) / NT;    // float4 weight loads per thread
  constexpr int WF4 = CK / 4 * BN;                   // float4s in one weight panel
  constexpr bool W_ALL = WF4 == WR * NT;             // every thread owns exactly WR of them: no per-thread condition around the requests

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;               // [2][HALO_F]
  float* wbuf = smem + 2 * HALO_F;  // [2][WB_F]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = bid % p.n_ntiles;
  const int mtile = bid / p.n_ntiles;
  const int tw_i = mtile % p.tiles_w;
  const int th_i = (mtile / p.tiles_w) % p.tiles_h;
  const int n_img = mtile / (p.tiles_w * p.tiles_h);
  const int h0 = th_i * TH, w0 = tw_i * TW, n0 = ntile * BN;

  const int64_t ximg = p.xa.off(n_img), yimg = p.ya.off(n_img);

  // ---- per-thread staging geometry (fixed across the K loop)
  int64_t hoff[HR];
  bool hok[HR];
  int hlds[HR];
#pragma unroll
  for (int r = 0; r < HR; ++r) {
    const int idx = tid + r * NT;
    const int pix = idx >> 2, q = idx & 3;
    const int hh = pix / HWD, ww = pix - hh * HWD;
    const int gh = S * h0 - (S == 1 ? DIL : p.pbh) + hh, gw = S * w0 - (S == 1 ? DIL : p.pbw) + ww;
    const int Hin = S == 1 ? p.H : p.Hin, Win = S == 1 ? p.W : p.Win;
    hok[r] = (idx < HALO_PIX * 4) && gh >= 0 && gh < Hin && gw >= 0 && gw < Win;
    hoff[r] = ximg + ((int64_t)gh * Win + gw) * p.xs + q * 4;
    hlds[r] = (idx < HALO_PIX * 4) ? hh * HROW + ww * PS + q * 4 : -1;
  }
  const int cin4 = p.Cin >> 2;
  int64_t woff[WR];
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    const int idx = min(tid + r * NT, WF4 - 1);      // threads beyond a small panel (WF4 < NT) request its last float4 again: no branch
    const int q = idx / BN, n = idx - q * BN;
    woff[r] = ((int64_t)q * p.Cout + n0 + n) * 4;
  }

  // fused depth taps (p.kd > 1): the chunk index runs over (valid depth tap, 16-channel chunk); a tap whose input plane
  // lies outside the sample is skipped outright (block-uniform: one block = one output plane)
  const int nchunks = p.Cin / CK;
  int c_begin = 0, c_end = nchunks;
  if (p.kd > 1) {
    const int dep = n_img % p.spg;
    int dt_lo = 0, dt_hi = p.kd;
    while (dt_lo < dt_hi && dep * p.dsd + p.dshift0 + dt_lo < 0) ++dt_lo;
    while (dt_hi > dt_lo && dep * p.dsd + p.dshift0 + dt_hi - 1 >= p.din) --dt_hi;
    c_begin = dt_lo * nchunks;
    c_end = dt_hi * nchunks;
  }
  float4 hreg[HR], wreg[WR];
  auto load_halo = [&](int cc) {
    const int dt = p.kd > 1 ? cc / nchunks : 0, c = cc - dt * nchunks;
    const int64_t soff = (p.kd > 1 ? (int64_t)(p.dshift0 + dt) * p.dplane : 0) + c * CK;
    // an out-of-image halo pixel reads a page of zeros: the request itself is UNCONDITIONAL.  (Rounds 1-4 branched around it,
    // `ok ? ldg4(..) : 0`: exec-masked blocks, and the compiler closed each with an s_waitcnt vmcnt(0) -- at the halo tap the wave
    // waited out the filter panel it had requested two instructions earlier, in front of the tap's MFMAs; same finding as in
    // conv_igemm_lin.hip, where it cost 7-8 %.  tests/test_asm_lint.py checks the placement.)
#pragma unroll
    for (int r = 0; r < HR; ++r) hreg[r] = ldg4(hok[r] ? p.x + hoff[r] + soff : kZeroF4);
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int r = 0; r < HR; ++r)
      if (hlds[r] >= 0) *reinterpret_cast<float4*>(&halo[buf * HALO_F + hlds[r]]) = hreg[r];
  };
  auto load_w = [&](int cc, int t) {
    const int dt = p.kd > 1 ? cc / nchunks : 0, c = cc - dt * nchunks;
    const float* base = p.wp + ((int64_t)(dt * 9 + t) * cin4 + c * (CK / 4)) * p.Cout * 4;
#pragma unroll
    for (int r = 0; r < WR; ++r)
      wreg[r] = ldg4(base + woff[r]);                 // unconditional (see woff)
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int r = 0; r < WR; ++r)
      if (W_ALL || tid + r * NT < WF4) *reinterpret_cast<float4*>(&wbuf[buf * WB_F + (tid + r * NT) * 4]) = wreg[r];
  };

  // ---- fragment addresses
  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int sub = wm * TM + tm;
    abase[tm] = S * (2 * sub + (l31 >> 4)) * HROW + S * (l31 & 15) * PS + 4 * h;
  }
  const int bbase = (h * BN + wn * TN * 32 + l31) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  load_halo(c_begin);
  load_w(c_begin, 0);
  store_halo(0);
  store_w(0);
  __syncthreads();

  int step = 0;
  for (int c = c_begin; c < c_end; ++c) {
    const float* hb = halo + ((c - c_begin) & 1) * HALO_F;
    const bool more_chunks = (c + 1 < c_end);
#pragma unroll
    for (int t = 0; t < 9; ++t, ++step) {
      const bool has_next = (t < 8) || more_chunks;
      if (has_next) load_w(t < 8 ? c : c + 1, t < 8 ? t + 1 : 0);
      if (t == HALO_LD_T && more_chunks) load_halo(c + 1);
      __builtin_amdgcn_sched_barrier(0);       // the requests stay in front of the tap's MFMAs (the scheduler sinks them to their LDS write otherwise)

      const float* wb = wbuf + (step & 1) * WB_F;
      const int toff = (t / 3) * DIL * HROW + (t % 3) * DIL * PS;
#pragma unroll
      for (int g = 0; g < CK / 8; ++g) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          a[tm] = *reinterpret_cast<const float4*>(&hb[abase[tm] + toff + 8 * g]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          b[tn] = *reinterpret_cast<const float4*>(&wb[bbase + (2 * g * BN + tn * 32) * 4]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
          }
      }

      if (has_next) store_w((step + 1) & 1);
      if (t == HALO_ST_T && more_chunks) store_halo((c + 1 - c_begin) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: store raw conv output, accumulate per-channel statistics
  float ssum[TN], ssq[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ssum[tn] = ssq[tn] = 0.f;
  float nsc[TN], nsh[TN], nmu[TN], nrs[TN];
  if constexpr (NBR) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int64_t o = (int64_t)n_img * p.nsst + n0 + (wn * TN + tn) * 32 + l31;
      nsc[tn] = p.nsc[o]; nsh[tn] = p.nsh[o]; nmu[tn] = p.nmu[o]; nrs[tn] = p.nrs[o];
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int sub = wm * TM + tm;
    // y += acc (depth taps of a 3-D conv): ALL old values of the fragment are loaded first, so their latencies overlap
    // (a load-add-store per row serialises 16 round trips to memory)
    float prior[16][TN];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) prior[r][tn] = 0.f;
    if constexpr (ACC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = mfma32_row(r, h);
        const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
        if (gh < p.H && gw < p.W) {
          const float* yp = p.y + yimg + ((int64_t)gh * p.W + gw) * p.ys + n0 + wn * TN * 32 + l31;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) prior[r][tn] = yp[tn * 32];
        }
      }
    }
    if constexpr (AFF) {
      float sc[TN], sh[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        sc[tn] = p.asc[n0 + (wn * TN + tn) * 32 + l31];
        sh[tn] = p.ash[n0 + (wn * TN + tn) * 32 + l31];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = mfma32_row(r, h);
        const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = fmaxf(fmaf(acc[tm][tn][r], sc[tn], sh[tn]), 0.f);
        if (gh < p.H && gw < p.W) {
          float* yp = p.y + yimg + ((int64_t)gh * p.W + gw) * p.ys + n0 + wn * TN * 32 + l31;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) yp[tn * 32] = acc[tm][tn][r];
        }
      }
      if constexpr (POOL) {
        // window q of the lane: registers 2 q, 2 q + 1 (tile row 2 sub) and 8 + 2 q, 9 + 2 q (tile row 2 sub + 1)
        const int Hp = p.H >> 1, Wp = p.W >> 1;
        float* pimg = static_cast<float*>(p.pool) + (int64_t)n_img * Hp * Wp * p.pool_s + n0 + wn * TN * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = mfma32_row(2 * q, h);
          const int gh = h0 + 2 * sub, gw = w0 + (i & 15);
          if (gh < p.H && gw < p.W) {
            float* pp = pimg + ((int64_t)(gh >> 1) * Wp + (gw >> 1)) * p.pool_s;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              pp[tn * 32] = fmaxf(fmaxf(acc[tm][tn][2 * q], acc[tm][tn][2 * q + 1]),
                                  fmaxf(acc[tm][tn][8 + 2 * q], acc[tm][tn][9 + 2 * q]));
          }
        }
      }
      continue;
    }
    if constexpr (NBR) {
      // fused norm-backward reduction of the producing unit (see ConvParams::ny): its raw output at this fragment's
      // pixels, eight rows' loads issued together (all sixteen cost 32 more live registers; the 2-D input gradient never
      // accumulates, so `prior` is dead here)
      const float* nyb = static_cast<const float*>(p.ny) + (int64_t)n_img * p.H * p.W * p.nys + n0 + wn * TN * 32 + l31;
#pragma unroll
      for (int rb = 0; rb < 16; rb += 8) {
        float qv[8][TN];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int i = mfma32_row(rb + r, h);
          const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
          const bool ok = gh < p.H && gw < p.W;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) qv[r][tn] = ok ? nyb[((int64_t)gh * p.W + gw) * p.nys + tn * 32] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int i = mfma32_row(rb + r, h);
          const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
          if (gh < p.H && gw < p.W) {
            float* yp = p.y + yimg + ((int64_t)gh * p.W + gw) * p.ys + n0 + wn * TN * 32 + l31;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              const float v = acc[tm][tn][rb + r];
              yp[tn * 32] = v;
              const float q = qv[r][tn];
              const float du = fmaf(q, nsc[tn], nsh[tn]) > 0.f ? v : 0.f;
              ssum[tn] += du;
              ssq[tn] += du * ((q - nmu[tn]) * nrs[tn]);
            }
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = mfma32_row(r, h);
      const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
      if (gh < p.H && gw < p.W) {
        float* yp = p.y + yimg + ((int64_t)gh * p.W + gw) * p.ys + n0 + wn * TN * 32 + l31;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const float v = acc[tm][tn][r] + prior[r][tn];
          yp[tn * 32] = v;
          ssum[tn] += v;
          ssq[tn] += v * v;
        }
      }
    }
  }
  if constexpr (AFF) return;
  if (p.stat != nullptr) {
    float* red = smem;  // [2][WM][BN]; all LDS reads of the main loop are behind its last barrier
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      ssum[tn] += __shfl_xor(ssum[tn], 32);
      ssq[tn] += __shfl_xor(ssq[tn], 32);
      if (h == 0) {
        red[(0 * WM + wm) * BN + (wn * TN + tn) * 32 + l31] = ssum[tn];
        red[(1 * WM + wm) * BN + (wn * TN + tn) * 32 + l31] = ssq[tn];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int k = tid / BN, n = tid - k * BN;
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < WM; ++m) s += red[(k * WM + m) * BN + n];
      p.stat[((int64_t)k * p.stat_rows + mtile) * p.Cout + n0 + n] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Direct kernel for layers the MFMA path does not cover (Encode1/conv1: Cin = 3, K = 27; UNet3D conv_e0/conv1: Cin = 1;
// the guided nets' 4- and 5-channel inputs; --img_grad: 9).  HBM-bound by design: it writes Cout floats per pixel against
// Cin read.  One block per 8x16 pixel tile; thread = (pixel lane, 4 output channels).  The thread's 9 x CIN x 4 filter
// taps live in registers and the 10x18xCIN input halo in LDS (read as broadcasts by the Cout/4 threads of a pixel): the
// first version re-read both through L1 per pixel and ran at 0.9 TB/s of output, VALU / L1-issue bound.
// CIN = 0: generic fallback (any Cin, filters through L1).
// TY = bf16_t: UNETK_BF16S -- the output is stored as bf16 (statistics still from the fp32 sums); the input stays fp32.
template <int CIN, typename TY = float>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [max(180 * cin, 2 * PL * Cout)]
  const int cin = CIN > 0 ? CIN : p.Cin;
  const int cq_n = p.Cout >> 2;
  const int PL = 256 / cq_n;
  const int tid = threadIdx.x;
  const int cq = tid % cq_n, pl = tid / cq_n;
  const int mtile = blockIdx.x;
  const int tw_i = mtile % p.tiles_w;
  const int th_i = (mtile / p.tiles_w) % p.tiles_h;
  const int n_img = mtile / (p.tiles_w * p.tiles_h);
  const int h0 = th_i * 8, w0 = tw_i * TW;
  const int64_t ximg = p.xa.off(n_img);
  // input halo (10 x 18 pixels x cin) -> LDS, zero outside the image
  for (int i = tid; i < 180 * cin; i += 256) {
    const int pix = i / cin, ci = i - pix * cin;
    const int ih = h0 + pix / 18 - 1, iw = w0 + pix % 18 - 1;
    smem[i] = (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? p.x[ximg + ((int64_t)ih * p.W + iw) * p.xs + ci] : 0.f;
  }
  float4 wreg[CIN > 0 ? 9 * CIN : 1];
  if (CIN > 0 && pl < PL) {
#pragma unroll
    for (int k = 0; k < 9 * CIN; ++k) wreg[k] = ldg4(p.wp + (int64_t)k * p.Cout + cq * 4);   // HWIO: k = tap * CIN + ci
  }
  __syncthreads();
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  if (pl < PL) {
    for (int pix = pl; pix < 128; pix += PL) {
      const int r = pix >> 4, c = pix & 15;
      const int gh = h0 + r, gw = w0 + c;
      if (gh >= p.H || gw >= p.W) continue;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (CIN > 0) {
        // two-wide vectors so that the compiler emits v_pk_fma_f32 (two IEEE fmas per instruction: bit-identical to the
        // scalar chain, half the vector issue slots -- this kernel is VALU-bound: 27 x 4 fmas per pixel and thread)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float* xp = &smem[((r + t / 3) * 18 + c + t % 3) * CIN];
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const float xv = xp[ci];
            const float4 wv = wreg[t * CIN + ci];
            const f32x2 x2 = {xv, xv};
            a01 = __builtin_elementwise_fma(x2, f32x2{wv.x, wv.y}, a01);
            a23 = __builtin_elementwise_fma(x2, f32x2{wv.z, wv.w}, a23);
          }
        }
        a = make_float4(a01.x, a01.y, a23.x, a23.y);
      } else {
        for (int t = 0; t < 9; ++t) {
          const float* xp = &smem[((r + t / 3) * 18 + c + t % 3) * cin];
          const float* wq = p.wp + ((int64_t)t * cin) * p.Cout + cq * 4;
          for (int ci = 0; ci < cin; ++ci) {
            const float xv = xp[ci];
            const float4 wv = ldg4(wq + (int64_t)ci * p.Cout);
            a.x = fmaf(xv, wv.x, a.x); a.y = fmaf(xv, wv.y, a.y); a.z = fmaf(xv, wv.z, a.z); a.w = fmaf(xv, wv.w, a.w);
          }
        }
      }
      TY* yp = reinterpret_cast<TY*>(p.y) + p.ya.off(n_img) + ((int64_t)gh * p.W + gw) * p.ys + cq * 4;
      if (p.accumulate) {
        const float4 o = ld4(yp);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      st4(yp, a);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      sq.x += a.x * a.x; sq.y += a.y * a.y; sq.z += a.z * a.z; sq.w += a.w * a.w;
    }
  }
  if (p.stat != nullptr) {
    __syncthreads();                                             // the halo is dead: reuse LDS as [2][PL][Cout]
    if (pl < PL) {
      stg4(&smem[(0 * PL + pl) * p.Cout + cq * 4], s);
      stg4(&smem[(1 * PL + pl) * p.Cout + cq * 4], sq);
    }
    __syncthreads();
    for (int i = tid; i < 2 * p.Cout; i += 256) {
      const int k = i / p.Cout, c = i - k * p.Cout;
      float v = 0.f;
      for (int j = 0; j < PL; ++j) v += smem[(k * PL + j) * p.Cout + c];
      p.stat[((int64_t)k * p.stat_rows + mtile) * p.Cout + c] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same layers on the matrix pipe when Cout = 64 and 1 <= Cin <= 5 (Encode1/conv1 of every 2-D net: Cin = 3, or 4 / 5 with a
// guide joined): the direct kernel above issues 27 x 64 fmas per pixel on the vector pipe (7.2 GFLOP at 256^2 x 32: 0.2 ms,
// three times its write time).  Here the 8 x 16 tile is a [128 px] x [K = 9 Cin, padded to even] x [64] GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32): wave w owns tile rows 2w, 2w + 1 (32 pixels) x two 32-column accumulators, lane
// (l31, h) feeds pixel l31's input value of k = 2 s + h from the LDS halo (one ds_read_b32 per step; the halo offset of k is a
// per-lane constant) and filter values from registers.  Accumulator column l31 of tile t is output channel 2 l31 + t, so a
// lane stores a channel PAIR per pixel (8 bytes fp32, 4 bytes bf16: 256- / 128-byte rows per store instruction) straight
// from the accumulators; statistics = per-lane column sums, the two row halves by a shuffle, the four waves through LDS.
// AFF: the inference epilogue (see conv3x3_igemm_kernel MODE 3): the activation relu(acc * scale + shift) is stored, no statistics.
template <int CIN, typename TY = float, bool AFF = false>
__global__ __launch_bounds__(256) void conv3x3_c3_mfma_kernel(ConvParams p) {
  static_assert(CIN >= 1 && CIN <= 5, "K = 9 Cin <= 46");
  constexpr int K = 9 * CIN, NS = (K + 1) / 2, COUT = 64;
  constexpr int NH = (180 * CIN + 255) / 256;                    // halo values per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];  // halo [180 * CIN] (+ pad), then the waves' statistic rows [2][4][64]
  float* halo = smem;
  float* red = smem + ((180 * CIN + 3) & ~3);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  // filter values of this lane: rows k = 2 s + h of the HWIO filter [9 Cin][64], channels 2 l31 and 2 l31 + 1 -- loaded ONCE: the
  // block is persistent (a block per tile spent most of its life on this load and the first halo's round trip)
  float b0[NS], b1[NS];
  int aoff[NS];
  const int pr = 2 * wave + (l31 >> 4), pc = l31 & 15;            // this lane's pixel of the tile (the A operand's row)
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2) {
    const int k = 2 * s2 + h;
    const bool on = k < K;
    const int kk = on ? k : 0;
    const float2 wv = *reinterpret_cast<const float2*>(p.wp + (int64_t)kk * COUT + 2 * l31);
    b0[s2] = on ? wv.x : 0.f;
    b1[s2] = on ? wv.y : 0.f;
    const int t = kk / CIN, ci = kk - t * CIN;
    aoff[s2] = ((pr + t / 3) * 18 + pc + t % 3) * CIN + ci;
  }
  float asc0 = 1.f, asc1 = 1.f, ash0 = 0.f, ash1 = 0.f;
  if constexpr (AFF) {
    asc0 = p.asc[2 * l31]; asc1 = p.asc[2 * l31 + 1];
    ash0 = p.ash[2 * l31]; ash1 = p.ash[2 * l31 + 1];
  }
  const int n_tiles = p.N * p.tiles_h * p.tiles_w;
  // the next tile's halo waits in registers while this tile is contracted and stored
  float hreg[NH];
  auto fetch = [&](int mtile) {
    const int tw_i = mtile % p.tiles_w;
    const int th_i = (mtile / p.tiles_w) % p.tiles_h;
    const int n_img = mtile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * 8, w0 = tw_i * TW;
    const int64_t ximg = p.xa.off(n_img);
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int i = tid + j * 256;
      const int pix = i / CIN, ci = i - pix * CIN;
      const int ih = h0 + pix / 18 - 1, iw = w0 + pix % 18 - 1;
      hreg[j] = (i < 180 * CIN && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? p.x[ximg + ((int64_t)ih * p.W + iw) * p.xs + ci] : 0.f;
    }
  };
  if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int mtile = blockIdx.x; mtile < n_tiles; mtile += gridDim.x) {
    const int tw_i = mtile % p.tiles_w;
    const int th_i = (mtile / p.tiles_w) % p.tiles_h;
    const int n_img = mtile / (p.tiles_w * p.tiles_h);
    const int h0 = th_i * 8, w0 = tw_i * TW;
    __syncthreads();                         // the previous tile's halo and statistic rows have been read
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int i = tid + j * 256;
      if (i < 180 * CIN) halo[i] = hreg[j];
    }
    __syncthreads();
    if (mtile + (int)gridDim.x < n_tiles) fetch(mtile + gridDim.x);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const float a = halo[aoff[s2]];          // k >= K: b is zero, any finite a will do
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[s2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[s2], acc1, 0, 0, 0);
    }
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    TY* yimg = reinterpret_cast<TY*>(p.y) + p.ya.off(n_img) + 2 * l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pix = wave * 32 + mfma32_row(r, h);
      const int gh = h0 + (pix >> 4), gw = w0 + (pix & 15);
      if (gh < p.H && gw < p.W) {
        float v0 = acc0[r], v1 = acc1[r];
        if constexpr (AFF) {
          v0 = fmaxf(fmaf(v0, asc0, ash0), 0.f);
          v1 = fmaxf(fmaf(v1, asc1, ash1), 0.f);
        }
        TY* yp = yimg + ((int64_t)gh * p.W + gw) * p.ys;
        if constexpr (std::is_same<TY, float>::value) {
          if (!AFF && p.accumulate) {
            const float2 o = *reinterpret_cast<const float2*>(yp);
            v0 += o.x; v1 += o.y;
          }
          *reinterpret_cast<float2*>(yp) = make_float2(v0, v1);
        } else {
          *reinterpret_cast<uint32_t*>(yp) = unetk_pk_bf16(v0, v1);
        }
        s0 += v0; s1 += v1; q0 += v0 * v0; q1 += v1 * v1;
      }
    }
    if (!AFF && p.stat != nullptr) {
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
      if (h == 0) {
        red[(0 * 4 + wave) * COUT + 2 * l31] = s0; red[(0 * 4 + wave) * COUT + 2 * l31 + 1] = s1;
        red[(1 * 4 + wave) * COUT + 2 * l31] = q0; red[(1 * 4 + wave) * COUT + 2 * l31 + 1] = q1;
      }
      __syncthreads();
      if (tid < 2 * COUT) {
        const int k = tid / COUT, c = tid - k * COUT;
        p.stat[((int64_t)k * p.stat_rows + mtile) * COUT + c] =
            red[(k * 4 + 0) * COUT + c] + red[(k * 4 + 1) * COUT + c] + red[(k * 4 + 2) * COUT + c] + red[(k * 4 + 3) * COUT + c];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Filter re-layout.  K4-interleaved panel: wp[t][q][n][j] = B_t[k = 4q + j][n].
//   forward : B_t[k = ci][n = co] = w[t][ci][co]
//   dgrad   : B_t[k = co][n = ci] = w[8 - t][ci][co]   (taps flipped, channels swapped)
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int Cin, int Cout,
                                    float* __restrict__ wp_fwd, float* __restrict__ wp_dgrad) {
  const int64_t total = (int64_t)9 * Cin * Cout / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    unetk_pack::conv3x3_f32(w, Cin, Cout, wp_fwd, wp_dgrad, i);      // csrc/pack.h
}

struct ConvCfg {
  int id;  // 0: 128x128 tile, 1: 128x64 tile, 2: 256x32 tile (UNet3D's 30-channel levels, padded to 32), -1: direct
  int th;
};

// Small grids (deep levels at small batch: 8 x 32^2 pixels x 512 couts = 256 blocks of 128 x 128) leave CUs idle or
// at one block each: halve the pixel tile (4-row tiles, <2,2,1,2>) when the 128 x 128 grid has fewer than 1.5 blocks
// per CU.  The statistic rows follow the tile height, so both users go through this one predicate.
inline bool small_grid(int N, int H, int W, int Cout) {
  if (Cout % 128 != 0) return false;
  const int64_t blocks = (int64_t)N * ((H + 7) / 8) * ((W + TW - 1) / TW) * (Cout / 128);
  return blocks < 384;
}

// Large grids: a 16 x 16 pixel tile (256 x 128, <2,2,4,2>, 128 accumulator registers per wave, two blocks per CU).  The
// staging traffic per MFMA is what holds the 128 x 128 tile at 87-89 % MFMA-busy (tools/mfma_mix.hip: the same loop
// without its global loads and LDS writes runs at 95-96 %): per 16-channel chunk a block stages the halo once and NINE
// filter panels, so doubling the pixels per block nearly halves the loads, LDS writes and barriers per MFMA.
// The 64-wide configuration has its 16 x 16 pixel tile too (256 x 64, <4,1,2,2>).  `nbr`: the launch also emits the producing
// unit's norm-backward reduction; with K < 256 channels the loop is too short to carry the four-fragment epilogue (measured
// on 128 -> 128 at 128^2: 1.19 -> 1.20 ms) and the 8-row tile stays.
inline bool big_grid(int N, int H, int W, int Cin, int Cout, int spg, bool nbr) {
  if (Cout % 64 != 0 || spg != 1 || H % 16 != 0 || (nbr && Cin < 256)) return false;
  const int bn = Cout % 128 == 0 ? 128 : 64;
  const int64_t blocks = (int64_t)N * (H / 16) * ((W + TW - 1) / TW) * (Cout / bn);
  return blocks >= 512;
}

inline ConvCfg pick_cfg(int Cin, int Cout) {
  if (Cin % CK == 0 && Cout % 128 == 0) return {0, 8};
  if (Cin % CK == 0 && Cout % 64 == 0) return {1, 8};
  if (Cin % CK == 0 && Cout % 32 == 0) return {2, 16};
  return {-1, 8};
}

// rows of the pixel tile the stride-1 tiled kernel takes for this shape (one statistic row per tile)
inline int tile_rows(int N, int H, int W, int Cin, int Cout, int spg, bool nbr) {
  const ConvCfg cfg = pick_cfg(Cin, Cout);
  if (cfg.id == 0 && small_grid(N, H, W, Cout)) return 4;
  if ((cfg.id == 0 || cfg.id == 1) && big_grid(N, H, W, Cin, Cout, spg, nbr)) return 16;
  return cfg.th;
}

template <int WM, int WN, int TM, int TN, int S = 1, int DIL = 1, int MODE = 0>
int launch_igemm_mode(const ConvParams& p, int n_mtiles, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int TH = BM / TW;
  constexpr size_t lds = (2 * (S * TH + 2 * DIL) * halo_row_f(S, DIL) + 2 * CK * BN) * sizeof(float);
  static_assert(lds >= 2 * WM * BN * sizeof(float), "stat scratch must fit");
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3x3_igemm_kernel<WM, WN, TM, TN, S, DIL, MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int grid = n_mtiles * p.n_ntiles;
  UNETK_LAUNCH(kern, dim3(grid), dim3(WM * WN * 64), lds, st, p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

template <int WM, int WN, int TM, int TN, int S = 1, int DIL = 1, bool NBR = false>
int launch_igemm(const ConvParams& p, int n_mtiles, hipStream_t st) {
  if constexpr (NBR) return launch_igemm_mode<WM, WN, TM, TN, S, DIL, 2>(p, n_mtiles, st);
  if (p.asc != nullptr) {       // inference epilogue: (scale, shift) + ReLU [+ 2 x 2 max-pool]
    if constexpr (S == 1 && DIL == 1) {
      if (p.accumulate || p.stat != nullptr) return UNETK_E_UNSUPPORTED;
      return p.pool != nullptr ? launch_igemm_mode<WM, WN, TM, TN, 1, 1, 4>(p, n_mtiles, st)
                               : launch_igemm_mode<WM, WN, TM, TN, 1, 1, 3>(p, n_mtiles, st);
    } else {
      return UNETK_E_UNSUPPORTED;
    }
  }
  if (p.accumulate) return launch_igemm_mode<WM, WN, TM, TN, S, DIL, 1>(p, n_mtiles, st);
  return launch_igemm_mode<WM, WN, TM, TN, S, DIL, 0>(p, n_mtiles, st);
}

}  // namespace

// H, W = OUTPUT extent; stride 2 always takes the tiled kernel.  Cout % 128 == 0: 4-row tiles -- the stride-2 halo of an
// 8-row tile (18 x 34 pixels, 98 KB double-buffered) leaves one block per CU, the 10 x 34 halo of a 4-row tile two
// (53 -> 69 TFLOP/s on UNet3D's (1,2,2) layers); the 64-wide configuration is better off with 8 rows (measured).
static int s2_th(int Cout) { return Cout % 128 == 0 ? 4 : 8; }
int unetk_conv_stat_rows(int N, int H, int W, int Cin, int Cout, int spg, int stride, int dil) {
  if (dil == 2) return N * ((H + 7) / 8) * ((W + TW - 1) / TW);
  if (stride == 2) return N * ((H + s2_th(Cout) - 1) / s2_th(Cout)) * ((W + TW - 1) / TW);
  if (unetk_conv_lin_ok(N, H, W, Cin, Cout, spg)) return unetk_conv_stat_rows_lin(N, H, W, spg, Cout);
  const int th = tile_rows(N, H, W, Cin, Cout, spg, false);
  return N * ((H + th - 1) / th) * ((W + TW - 1) / TW);
}

bool unetk_conv_stride2_ok(int Cin, int Cout) { return Cin % CK == 0 && Cout % 64 == 0; }

int unetk_conv_run(ConvParams p, hipStream_t st) {
  if ((p.ny != nullptr || p.asc != nullptr) && (p.stride == 2 || p.dil == 2)) return UNETK_E_UNSUPPORTED;
  if (p.stride == 2) {   // p.H x p.W = output extent, p.Hin x p.Win = input extent
    if (p.bf16 || !unetk_conv_stride2_ok(p.Cin, p.Cout) || p.xs % 4 != 0) return UNETK_E_UNSUPPORTED;
    p.tiles_h = (p.H + s2_th(p.Cout) - 1) / s2_th(p.Cout);
    p.tiles_w = (p.W + TW - 1) / TW;
    const int n_mt = p.N * p.tiles_h * p.tiles_w;
    p.stat_rows = n_mt;
    if (p.Cout % 128 == 0) {
      p.n_ntiles = p.Cout / 128;
      return launch_igemm<2, 2, 1, 2, 2>(p, n_mt, st);
    }
    p.n_ntiles = p.Cout / 64;
    return launch_igemm<4, 1, 1, 2, 2>(p, n_mt, st);
  }
  if (p.dil == 2) {      // atrous: tiled fp32 kernel only (SmallUNet's two 32 x 32 levels)
    if (p.bf16 || p.Cin % CK != 0 || p.Cout % 64 != 0 || p.xs % 4 != 0) return UNETK_E_UNSUPPORTED;
    p.tiles_h = (p.H + 7) / 8;
    p.tiles_w = (p.W + TW - 1) / TW;
    const int n_mt = p.N * p.tiles_h * p.tiles_w;
    p.stat_rows = n_mt;
    if (p.Cout % 128 == 0) {
      p.n_ntiles = p.Cout / 128;
      return launch_igemm<2, 2, 2, 2, 1, 2>(p, n_mt, st);
    }
    p.n_ntiles = p.Cout / 64;
    return launch_igemm<4, 1, 1, 2, 1, 2>(p, n_mt, st);
  }
  if (p.dil > 2 || p.dil < 0) return UNETK_E_UNSUPPORTED;
  if (p.bf16) return (p.asc != nullptr && p.bf16 != UNETK_BF16S) ? UNETK_E_UNSUPPORTED : unetk_conv_run_bf16(p, st);
  if (p.spg < 1) p.spg = 1;
  if (unetk_conv_lin_ok(p.N, p.H, p.W, p.Cin, p.Cout, p.spg))                                     // small planes: linear M
    return (p.asc != nullptr && (p.pool != nullptr || p.accumulate || p.kd > 1)) ? UNETK_E_UNSUPPORTED : unetk_conv_run_lin(p, st);
  const ConvCfg cfg = pick_cfg(p.Cin, p.Cout);
  const int th = tile_rows(p.N, p.H, p.W, p.Cin, p.Cout, p.spg, p.ny != nullptr);
  const bool small = cfg.id == 0 && th == 4, big = cfg.id <= 1 && th == 16;
  p.tiles_h = (p.H + th - 1) / th;
  p.tiles_w = (p.W + TW - 1) / TW;
  const int n_mtiles = p.N * p.tiles_h * p.tiles_w;
  p.stat_rows = n_mtiles;
  if (p.ny != nullptr && cfg.id != 0) return UNETK_E_UNSUPPORTED;   // the fused reduction exists for the 128-wide tiles
  if (small) {
    p.n_ntiles = p.Cout / 128;
    if (p.ny != nullptr) return launch_igemm<2, 2, 1, 2, 1, 1, true>(p, n_mtiles, st);
    return launch_igemm<2, 2, 1, 2>(p, n_mtiles, st);
  }
  if (big && cfg.id == 1) {
    p.n_ntiles = p.Cout / 64;
    return launch_igemm<4, 1, 2, 2>(p, n_mtiles, st);
  }
  if (big) {
    p.n_ntiles = p.Cout / 128;
    if (p.ny != nullptr) return launch_igemm<2, 2, 4, 2, 1, 1, true>(p, n_mtiles, st);
    return launch_igemm<2, 2, 4, 2>(p, n_mtiles, st);
  }
  if (cfg.id == 0) {
    p.n_ntiles = p.Cout / 128;
    if (p.ny != nullptr) return launch_igemm<2, 2, 2, 2, 1, 1, true>(p, n_mtiles, st);
    return launch_igemm<2, 2, 2, 2>(p, n_mtiles, st);
  }
  if (cfg.id == 1) {
    p.n_ntiles = p.Cout / 64;
    return launch_igemm<4, 1, 1, 2>(p, n_mtiles, st);
  }
  if (cfg.id == 2) {
    p.n_ntiles = p.Cout / 32;
    return launch_igemm<4, 1, 2, 1>(p, n_mtiles, st);
  }
  if (p.Cout % 4 != 0 || p.Cout > 1024) return UNETK_E_UNSUPPORTED;
  const int PL = 256 / (p.Cout / 4);
  size_t lds = p.stat ? (size_t)2 * PL * p.Cout * sizeof(float) : 0;
  const size_t halo = (size_t)180 * p.Cin * sizeof(float);
  if (halo > lds) lds = halo;
  if (lds > 64 * 1024) return UNETK_E_UNSUPPORTED;
  if (p.Cout == 64 && p.Cin >= 1 && p.Cin <= 5) {      // first layers: the matrix-pipe variant
    const size_t l3 = ((size_t)((180 * p.Cin + 3) & ~3) + 2 * 4 * 64) * sizeof(float);
    const int c3_grid = n_mtiles < 256 * 6 ? n_mtiles : 256 * 6;      // persistent: six resident blocks per CU walk the tiles
#define C3_LAUNCH(CI)                                                                                              \
  case CI:                                                                                                         \
    if (p.asc != nullptr) {                                                                                        \
      if (p.ybf16) UNETK_LAUNCH((conv3x3_c3_mfma_kernel<CI, bf16_t, true>), dim3(c3_grid), dim3(256), l3, st, p);  \
      else UNETK_LAUNCH((conv3x3_c3_mfma_kernel<CI, float, true>), dim3(c3_grid), dim3(256), l3, st, p);           \
    } else if (p.ybf16) UNETK_LAUNCH((conv3x3_c3_mfma_kernel<CI, bf16_t>), dim3(c3_grid), dim3(256), l3, st, p);   \
    else UNETK_LAUNCH((conv3x3_c3_mfma_kernel<CI, float>), dim3(c3_grid), dim3(256), l3, st, p);                   \
    break;
    if (p.ybf16 && p.accumulate) return UNETK_E_UNSUPPORTED;
    if (p.asc != nullptr && (p.pool != nullptr || p.accumulate || p.stat != nullptr)) return UNETK_E_UNSUPPORTED;
    switch (p.Cin) { C3_LAUNCH(1) C3_LAUNCH(2) C3_LAUNCH(3) C3_LAUNCH(4) C3_LAUNCH(5) default: break; }
#undef C3_LAUNCH
    UNETK_LAUNCH_CHECK();
    return UNETK_OK;
  }
  if (p.asc != nullptr) return UNETK_E_UNSUPPORTED;      // the generic direct kernel has no inference epilogue
  if (p.ybf16) {      // UNETK_BF16S first layer: fp32 image in, bf16 out
    if (p.accumulate) return UNETK_E_UNSUPPORTED;
    switch (p.Cin) {
      case 1: UNETK_LAUNCH((conv3x3_direct_kernel<1, bf16_t>), dim3(n_mtiles), dim3(256), lds, st, p); break;
      case 2: UNETK_LAUNCH((conv3x3_direct_kernel<2, bf16_t>), dim3(n_mtiles), dim3(256), lds, st, p); break;
      case 3: UNETK_LAUNCH((conv3x3_direct_kernel<3, bf16_t>), dim3(n_mtiles), dim3(256), lds, st, p); break;
      case 4: UNETK_LAUNCH((conv3x3_direct_kernel<4, bf16_t>), dim3(n_mtiles), dim3(256), lds, st, p); break;
      case 5: UNETK_LAUNCH((conv3x3_direct_kernel<5, bf16_t>), dim3(n_mtiles), dim3(256), lds, st, p); break;
      default: UNETK_LAUNCH((conv3x3_direct_kernel<0, bf16_t>), dim3(n_mtiles), dim3(256), lds, st, p); break;
    }
    UNETK_LAUNCH_CHECK();
    return UNETK_OK;
  }
  switch (p.Cin) {
    case 1: UNETK_LAUNCH(conv3x3_direct_kernel<1>, dim3(n_mtiles), dim3(256), lds, st, p); break;
    case 2: UNETK_LAUNCH(conv3x3_direct_kernel<2>, dim3(n_mtiles), dim3(256), lds, st, p); break;
    case 3: UNETK_LAUNCH(conv3x3_direct_kernel<3>, dim3(n_mtiles), dim3(256), lds, st, p); break;
    case 4: UNETK_LAUNCH(conv3x3_direct_kernel<4>, dim3(n_mtiles), dim3(256), lds, st, p); break;
    case 5: UNETK_LAUNCH(conv3x3_direct_kernel<5>, dim3(n_mtiles), dim3(256), lds, st, p); break;
    default: UNETK_LAUNCH(conv3x3_direct_kernel<0>, dim3(n_mtiles), dim3(256), lds, st, p); break;
  }
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

namespace {

bool conv_desc_ok(const unetk_conv_desc* d) {
  return d && d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->x_stride >= d->Cin &&
         d->y_stride >= d->Cout;
}

}  // namespace

extern "C" int unetk_conv3x3_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                                  void* stream) {
  UNETK_REQUIRE(w && Cin > 0 && Cout > 0);
  if (Cin % 4 != 0 || Cout % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(w) && unetk_aligned16(wp_fwd) && unetk_aligned16(wp_dgrad));
  const int64_t total = (int64_t)9 * Cin * Cout / 4;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  UNETK_LAUNCH(pack_conv3x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, wp_fwd,
                     wp_dgrad);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_conv3x3_stat_rows(const unetk_conv_desc* d) {
  if (!conv_desc_ok(d)) return UNETK_E_BADARG;
  if (d->dilation == 2) return unetk_conv_stat_rows(d->N, d->H, d->W, d->Cin, d->Cout, 1, 1, 2);
  if (d->precision == UNETK_BF16) return unetk_conv_stat_rows_bf16(d->N, d->H, d->W, d->Cin, d->Cout);
  if (d->precision == UNETK_BF16S && unetk_conv_bf16_ok(d->Cin, d->Cout))
    return unetk_conv_stat_rows_bf16s(d->N, d->H, d->W, d->Cin, d->Cout, d->x_stride, d->y_stride);
  return unetk_conv_stat_rows(d->N, d->H, d->W, d->Cin, d->Cout);
}

extern "C" size_t unetk_conv3x3_ws_bytes(const unetk_conv_desc* d) {
  if (!conv_desc_ok(d) || d->precision != UNETK_FP32 || d->dilation > 1) return 0;
  const size_t f = unetk_conv_lin_sk_bytes(d->N, d->H, d->W, d->Cin, d->Cout, 1, 1);
  const size_t b = unetk_conv_lin_sk_bytes(d->N, d->H, d->W, d->Cout, d->Cin, 1, 1);
  return f > b ? f : b;
}

extern "C" int unetk_conv3x3_fwd(const unetk_conv_desc* d, const void* x, const void* w, void* y,
                                 float* stat_partials, void* stream) {
  return unetk_conv3x3_fwd_ws(d, x, w, y, stat_partials, nullptr, 0, stream);
}

extern "C" int unetk_conv3x3_fwd_ws(const unetk_conv_desc* d, const void* x, const void* w, void* y,
                                    float* stat_partials, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(conv_desc_ok(d) && x && w && y);
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(w) && unetk_aligned16(y));
  UNETK_REQUIRE(d->y_stride % 4 == 0);
  if (pick_cfg(d->Cin, d->Cout).id >= 0) UNETK_REQUIRE(d->x_stride % 4 == 0);
  if (d->precision == UNETK_BF16 && !unetk_conv_bf16_ok(d->Cin, d->Cout)) return UNETK_E_UNSUPPORTED;
  ConvParams p{};
  p.bf16 = d->precision == UNETK_BF16 ? 1 : 0;
  if (d->precision == UNETK_BF16S) {
    if (d->dilation > 1) return UNETK_E_UNSUPPORTED;
    if (unetk_conv_bf16_ok(d->Cin, d->Cout)) p.bf16 = UNETK_BF16S;         // bf16 in, bf16 out (igemm on the bf16 matrix cores)
    else if (pick_cfg(d->Cin, d->Cout).id < 0) p.ybf16 = 1;                // first layer: fp32 image in, bf16 out (direct kernel)
    else return UNETK_E_UNSUPPORTED;
  }
  p.x = (const float*)x; p.wp = (const float*)w; p.y = (float*)y; p.stat = stat_partials;
  p.dil = d->dilation;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.xs = d->x_stride; p.ys = d->y_stride;
  p.xa = unetk_dense_addr(p.H, p.W, p.xs);
  p.ya = unetk_dense_addr(p.H, p.W, p.ys);
  if (ws && unetk_aligned16(ws) && ws_bytes > 0) { p.sk_slab = (float*)ws; p.sk_slab_bytes = ws_bytes; }
  return unetk_conv_run(p, (hipStream_t)stream);
}

// ---- inference: conv + (scale, shift) + ReLU [+ 2 x 2 max-pool] in one pass
extern "C" int unetk_conv3x3_fwd_affine_ok(const unetk_conv_desc* d, int with_pool) {
  if (!conv_desc_ok(d) || d->dilation > 1 || d->precision == UNETK_BF16) return 0;
  if (with_pool && ((d->H | d->W) & 1)) return 0;
  const bool first = pick_cfg(d->Cin, d->Cout).id < 0 && d->Cout == 64 && d->Cin >= 1 && d->Cin <= 5;   // Encode1/conv1 on the matrix pipe
  if (d->precision == UNETK_BF16S) {     // the persistent bf16-storage kernel (and the first layer: fp32 image in, bf16 out)
    if (unetk_conv_bf16_ok(d->Cin, d->Cout))
      return unetk_conv_bf16s_v3_ok(d->N, d->H, d->W, d->Cin, d->Cout, d->x_stride, d->y_stride) ? 1 : 0;
    return (first && !with_pool) ? 1 : 0;
  }
  if (pick_cfg(d->Cin, d->Cout).id >= 0)       // the small-plane (linear-pixel) kernel has the affine epilogue but no pool
    return (unetk_conv_lin_ok(d->N, d->H, d->W, d->Cin, d->Cout, 1) && with_pool) ? 0 : 1;
  return (first && !with_pool) ? 1 : 0;
}

extern "C" int unetk_conv3x3_fwd_affine(const unetk_conv_desc* d, const void* x, const void* w, const float* scale,
                                        const float* shift, void* z, void* pooled, int pooled_stride, void* ws,
                                        size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(conv_desc_ok(d) && x && w && z && scale && shift);
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(w) && unetk_aligned16(z));
  UNETK_REQUIRE(d->y_stride % 4 == 0 && (pooled == nullptr || (pooled_stride >= d->Cout && unetk_aligned16(pooled))));
  if (!unetk_conv3x3_fwd_affine_ok(d, pooled != nullptr)) return UNETK_E_UNSUPPORTED;
  if (pick_cfg(d->Cin, d->Cout).id >= 0) UNETK_REQUIRE(d->x_stride % 4 == 0);
  ConvParams p{};
  if (d->precision == UNETK_BF16S) {
    if (unetk_conv_bf16_ok(d->Cin, d->Cout)) p.bf16 = UNETK_BF16S;      // bf16 in, bf16 out
    else p.ybf16 = 1;                                                   // first layer: fp32 image in, bf16 out
  }
  p.x = (const float*)x; p.wp = (const float*)w; p.y = (float*)z; p.stat = nullptr;
  p.asc = scale; p.ash = shift; p.pool = pooled; p.pool_s = pooled_stride;
  if (ws && unetk_aligned16(ws) && ws_bytes > 0) { p.sk_slab = (float*)ws; p.sk_slab_bytes = ws_bytes; }   // stream-K scratch
  p.dil = d->dilation;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.xs = d->x_stride; p.ys = d->y_stride;
  p.xa = unetk_dense_addr(p.H, p.W, p.xs);
  p.ya = unetk_dense_addr(p.H, p.W, p.ys);
  return unetk_conv_run(p, (hipStream_t)stream);
}

extern "C" int unetk_conv3x3_dgrad(const unetk_conv_desc* d, const void* dy, const void* w, void* dx,
                                   void* stream) {
  return unetk_conv3x3_dgrad_ws(d, dy, w, dx, nullptr, 0, stream);
}

extern "C" int unetk_conv3x3_dgrad_ws(const unetk_conv_desc* d, const void* dy, const void* w, void* dx, void* ws,
                                      size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(conv_desc_ok(d) && dy && w && dx);
  UNETK_REQUIRE(unetk_aligned16(dy) && unetk_aligned16(w) && unetk_aligned16(dx));
  // dgrad = conv3x3 with Cin <-> Cout on the packed, tap-flipped filters
  if (pick_cfg(d->Cout, d->Cin).id < 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(d->x_stride % 4 == 0 && d->y_stride % 4 == 0);
  if (d->precision != UNETK_FP32 && !unetk_conv_bf16_ok(d->Cout, d->Cin)) return UNETK_E_UNSUPPORTED;
  if (d->precision == UNETK_BF16S && d->dilation > 1) return UNETK_E_UNSUPPORTED;
  ConvParams p{};
  p.bf16 = d->precision;                    // UNETK_FP32 / UNETK_BF16 / UNETK_BF16S (dy and dx are bf16)
  p.x = (const float*)dy; p.wp = (const float*)w; p.y = (float*)dx; p.stat = nullptr;
  p.dil = d->dilation;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cout; p.Cout = d->Cin; p.xs = d->y_stride; p.ys = d->x_stride;
  p.xa = unetk_dense_addr(p.H, p.W, p.xs);
  p.ya = unetk_dense_addr(p.H, p.W, p.ys);
  if (ws && unetk_aligned16(ws) && ws_bytes > 0) { p.sk_slab = (float*)ws; p.sk_slab_bytes = ws_bytes; }
  return unetk_conv_run(p, (hipStream_t)stream);
}

// ---- input gradient fused with the producing unit's norm-backward reduction (ConvParams::ny)
extern "C" int unetk_conv3x3_dgrad_nbr_rows(const unetk_conv_desc* d) {
  if (!conv_desc_ok(d) || d->dilation > 1) return 0;
  const int K = d->Cout, Nc = d->Cin;                       // the dgrad conv contracts Cout, produces Cin channels
  // measured: the extra epilogue work (16 x TN strided loads of prod_y per fragment) costs ~0.03 ms on the deep layers and
  // 0.38 ms on the 64-channel 256^2 level, whose K loop is only 36 steps -- more than the separate reduction pass there
  if (K < 128 || Nc < 128 || Nc % 128 != 0) return 0;
  if (d->precision == UNETK_BF16S) {
    if (!unetk_conv_bf16_ok(K, Nc) || K % 64 != 0 || Nc % 64 != 0) return 0;
    return unetk_conv_stat_rows_bf16s(d->N, d->H, d->W, K, Nc, d->y_stride, d->x_stride);
  }
  if (d->precision != UNETK_FP32) return 0;
  if (pick_cfg(K, Nc).id < 0 || unetk_conv_lin_ok(d->N, d->H, d->W, K, Nc, 1)) return 0;   // tiled fp32 kernel only
  const int th = tile_rows(d->N, d->H, d->W, K, Nc, 1, true);
  return d->N * ((d->H + th - 1) / th) * ((d->W + TW - 1) / TW);
}

extern "C" int unetk_conv3x3_dgrad_nbr(const unetk_conv_desc* d, const void* dy, const void* w, void* dx,
                                       const void* prod_y, int prod_y_stride, const float* scale, const float* shift,
                                       const float* mean, const float* rstd, int per_sample, float* partials,
                                       void* stream) {
  UNETK_REQUIRE(conv_desc_ok(d) && dy && w && dx && prod_y && scale && shift && mean && rstd && partials);
  UNETK_REQUIRE(unetk_aligned16(dy) && unetk_aligned16(w) && unetk_aligned16(dx));
  UNETK_REQUIRE(d->x_stride % 4 == 0 && d->y_stride % 4 == 0 && prod_y_stride >= d->Cin);
  if (unetk_conv3x3_dgrad_nbr_rows(d) <= 0) return UNETK_E_UNSUPPORTED;
  ConvParams p{};
  p.bf16 = d->precision;
  p.x = (const float*)dy; p.wp = (const float*)w; p.y = (float*)dx; p.stat = partials;
  p.dil = d->dilation;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cout; p.Cout = d->Cin; p.xs = d->y_stride; p.ys = d->x_stride;
  p.xa = unetk_dense_addr(p.H, p.W, p.xs);
  p.ya = unetk_dense_addr(p.H, p.W, p.ys);
  p.ny = prod_y; p.nys = prod_y_stride;
  p.nsc = scale; p.nsh = shift; p.nmu = mean; p.nrs = rstd;
  p.nsst = per_sample ? d->Cin : 0;
  return unetk_conv_run(p, (hipStream_t)stream);
}
