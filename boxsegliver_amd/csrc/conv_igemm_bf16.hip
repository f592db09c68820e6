// conv3x3 (stride 1, SAME) forward / input-gradient on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) with
// fp32 accumulation -- the mixed-precision mode of BASELINE.json configs[2] ("UNet 512x512x3 bs=64 bf16").
//
// Same implicit-GEMM structure as conv_igemm.hip (M = a TH x 16 pixel tile, N = Cout tile, K = 9 taps x Cin; the
// input halo staged once per channel chunk, all nine taps read shifted A fragments from it), re-proportioned for a
// matrix pipe that is 16x faster than the fp32 one:
//   * activations stay fp32 in HBM (round 1); the staging pass rounds them to bf16 (RNE, v_cvt_pk_bf16_f32) on
//     the way into LDS, so an LDS pixel row holds 32 channels in the 64 B the fp32 kernel needs for 16;
//   * filters are pre-packed to bf16 "K8-interleaved" [tap][Cin/8][Cout][8], so that -- exactly like the fp32
//     kernel's K4 layout -- every A and every B fragment of a 32x32x16 MFMA is ONE conflict-free ds_read_b128;
//   * per (chunk, tap) a wave issues TM + TN ds_read_b128 for TM x TN MFMAs of 32 cycles, and what bounds the
//     kernel is the L2 -> LDS stream (halo + the nine filter panels of the chunk), which scales as 1/BM for the
//     filters and 1/BN for the halo: hence the tall 512 x 128 block tile (8 waves, 4 x 2 MFMA tiles per wave).
// Statistics for the following norm come from the fp32 accumulators, as in the fp32 kernel.
//
// BS = true is UNETK_BF16S ("bf16 storage"): x is already bf16 in HBM (the halo is copied, not converted: half the
// staging bytes) and y is written as bf16.  So that a lane can store TWO adjacent output channels as one 4-byte word
// (a half-wave then writes 128 contiguous bytes per pixel instead of two 64-byte pieces), the filter pack of this mode
// (unetk_conv3x3_pack_bf16s) permutes the output channels inside every 64-channel block: MFMA column l of tile tn holds
// channel 2 l + tn.
#include "common.h"
#include "pack.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// 16-byte register staging uses the NATIVE vector type: an array of HIP's uint4 (a struct of unions) is not always
// promoted to registers by SROA -- two-element arrays ended up in scratch (found by tools/asm_lint.py)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int CKB = 32;  // input channels per K-chunk
// depth of the filter-panel register ring (see the kernel): 9 = a whole chunk of taps ahead.  A depth of 3 keeps the 4-wave
// configurations at three waves per SIMD instead of two, and is slower all the same (128 x 128: 834 vs 948 TF, 256 x 64:
// 723 vs 741 TF at 512 x 512 bs 8).
#ifndef UNETK_BF16_RING
#define UNETK_BF16_RING(WM, WN, TM, TN) 9
#endif
constexpr int PSQ = 5;   // LDS pixel stride in 16-B units: 64 B of bf16 + 16 B pad -> conflict-free ds_read_b128
constexpr int TW = 16;
constexpr int HWD = TW + 2;

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// NBR: input gradient fused with the producing unit's norm-backward reduction (ConvParams::ny) -- its own instantiation so
// the plain kernels keep their register budget (see conv_igemm.hip).
template <int WM, int WN, int TM, int TN, bool BS = false, bool NBR = false>
__global__ __launch_bounds__(WM* WN * 64) void conv3x3_igemm_bf16_kernel(ConvParams p) {
  constexpr int RING = UNETK_BF16_RING(WM, WN, TM, TN);   // depth of the filter-panel register ring (divides 9)
  static_assert(!BS || (TN == 2 && (WN * TN * 32 == 128 || WN * TN * 32 == 64)), "bf16 storage packs channel pairs (tile 0 / tile 1) into one word");
  static_assert(!NBR || BS, "the fused reduction is built for bf16 storage");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int TH = BM / TW, HH = TH + 2;
  constexpr int HALO_PIX = HH * HWD;
  constexpr int HALO_Q = HALO_PIX * PSQ;          // 16-B units per halo buffer
  constexpr int WB_Q = CKB / 8 * BN;              // 16-B units per filter panel
  constexpr int HR = (HALO_PIX * 4 + NT - 1) / NT;  // (pixel, 8-channel group) items per thread
  constexpr int WR = (WB_Q + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) uint4 smem_q[];
  uint4* halo = smem_q;               // [2][HALO_Q]
  uint4* wbuf = smem_q + 2 * HALO_Q;  // [2][WB_Q]

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

  int64_t hoff[HR];
  bool hok[HR];
  int hlds[HR];
#pragma unroll
  for (int r = 0; r < HR; ++r) {
    const int idx = tid + r * NT;
    const int pix = idx >> 2, q = idx & 3;
    const int hh = pix / HWD, ww = pix - hh * HWD;
    const int gh = h0 - 1 + hh, gw = w0 - 1 + ww;
    hok[r] = (idx < HALO_PIX * 4) && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
    hoff[r] = ximg + ((int64_t)gh * p.W + gw) * p.xs + q * 8;
    hlds[r] = (idx < HALO_PIX * 4) ? pix * PSQ + q : -1;
  }
  const int cin8 = p.Cin >> 3;
  const uint4* wq = reinterpret_cast<const uint4*>(p.wp);
  int woff[WR];
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    const int idx = tid + r * NT;
    const int q = idx / BN, n = idx - q * BN;
    woff[r] = q * p.Cout + n0 + n;
  }

  float4 hreg[BS ? 1 : HR][2];
  u32x4 hq[BS ? HR : 1];
  // Filter panels travel global -> registers -> LDS through a NINE-deep register ring (slot = tap): the panel of step s + 9
  // is requested when the panel of step s + 1 has been handed to LDS.  One step is only 0.1-0.2 us of MFMA work and VMEM
  // loads return in order, so a panel requested one step ahead (round 1-2) was waited for at every step for an L2 round
  // trip -- and for a whole HBM round trip at the step after the next chunk's halo was requested.
  u32x4 wring[RING][WR];
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  if constexpr (BS) {   // out-of-image items read the tensor's first bytes (always mapped) and are zeroed by a select: no
#pragma unroll          // per-thread branch around the loads for the compiler to unswitch the main loop on
    for (int r = 0; r < HR; ++r) hoff[r] = hok[r] ? hoff[r] : 0;
  }
  auto load_halo = [&](int c) {
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      if constexpr (BS) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(xb + hoff[r] + c * CKB);
        hq[r] = hok[r] ? v : u32x4{0u, 0u, 0u, 0u};
      } else if (hok[r]) {
        hreg[r][0] = ldg4(p.x + hoff[r] + c * CKB);
        hreg[r][1] = ldg4(p.x + hoff[r] + c * CKB + 4);
      } else {
        hreg[r][0] = hreg[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int r = 0; r < HR; ++r)
      if (hlds[r] >= 0) {
        u32x4 v;
        if constexpr (BS) {
          v = hq[r];
        } else {
          v.x = pk_bf16(hreg[r][0].x, hreg[r][0].y);
          v.y = pk_bf16(hreg[r][0].z, hreg[r][0].w);
          v.z = pk_bf16(hreg[r][1].x, hreg[r][1].y);
          v.w = pk_bf16(hreg[r][1].z, hreg[r][1].w);
        }
        *reinterpret_cast<u32x4*>(&halo[buf * HALO_Q + hlds[r]]) = v;
      }
  };
  auto load_w = [&](int c, int t) {        // t: compile-time under the unrolled tap loop (register-array index t % RING)
    const uint4* base = wq + ((int64_t)t * cin8 + c * (CKB / 8)) * p.Cout;
#pragma unroll
    for (int r = 0; r < WR; ++r)
      if (tid + r * NT < WB_Q) wring[t % RING][r] = *reinterpret_cast<const u32x4*>(base + woff[r]);
  };
  auto store_w = [&](int t, int buf) {
#pragma unroll
    for (int r = 0; r < WR; ++r)
      if (tid + r * NT < WB_Q) *reinterpret_cast<u32x4*>(&wbuf[buf * WB_Q + tid + r * NT]) = wring[t % RING][r];
  };

  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int sub = wm * TM + tm;
    abase[tm] = ((2 * sub + (l31 >> 4)) * HWD + (l31 & 15)) * PSQ + h;
  }
  // B fragment of tile tn: 16-byte unit of the panel.  fp32 storage: panel column = output channel.  bf16 storage: the lane
  // wants channels (2 l31, 2 l31 + 1) of the wave's 64-block for tiles 0 / 1 (one 4-byte store); the pack shared with the
  // round-3 kernel keeps channel c at column conv_bf16s_pos(c) (pack.h), conflict-free for these reads too.
  int bpos[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
    bpos[tn] = h * BN + (BS ? unetk_pack::conv_bf16s_pos(wn * 64 + 2 * l31 + tn, BN == 128 ? 128 : 64) : wn * TN * 32 + tn * 32 + l31);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int nchunks = p.Cin / CKB;

  load_halo(0);
#pragma unroll
  for (int t = 0; t < RING; ++t) load_w(0, t);
  store_halo(0);
  store_w(0, 0);
  load_w(min(RING / 9, nchunks - 1), RING % 9);            // step RING's panel into the slot just emptied
  __syncthreads();

  // The prefetches are UNCONDITIONAL (the last chunk re-fetches itself / tap 0 into the free buffers, never read): with
  // `if (more_chunks)` guards the compiler peeled the last chunk out of the loop and shuttled accumulators between
  // AGPRs and VGPRs around the peeled copy (tools/asm_lint.py: accumulator reads before the last MFMA).
  int step = 0;
  for (int c = 0; c < nchunks; ++c) {
    const uint4* hb = halo + (c & 1) * HALO_Q;
    const int cn = min(c + 1, nchunks - 1);
#pragma unroll
    for (int t = 0; t < 9; ++t, ++step) {
      if (t == 0) load_halo(cn);                     // HBM latency >> one tap step: a whole chunk of slack

      const uint4* wb = wbuf + (step & 1) * WB_Q;
      const int toff = ((t / 3) * HWD + (t % 3)) * PSQ;
#pragma unroll
      for (int g = 0; g < CKB / 16; ++g) {
        uint4 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = hb[abase[tm] + toff + 2 * g];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = wb[bpos[tn] + 2 * g * BN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[tm]),
                                                                  __builtin_bit_cast(bf16x8, b[tn]), acc[tm][tn], 0, 0, 0);
      }

      store_w((t + 1) % 9, (step + 1) & 1);                            // the next step's panel (requested RING - 1 steps ago)
      load_w(min(c + (t + 1 + RING) / 9, nchunks - 1), (t + 1 + RING) % 9);   // that slot's next use, RING steps from now
      if (t == 8) store_halo((c + 1) & 1);
      // NOT __syncthreads(): its fence waits for every outstanding global load (vmcnt(0)), i.e. it would force the halo
      // prefetch of the next chunk -- issued at tap 0 for use at tap 8 -- to land within ONE tap step.  The LDS writes
      // above must be complete (lgkmcnt(0)) before the barrier; the register-staged loads are tracked by the compiler.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // ---- epilogue: raw conv output (fp32) + per-channel statistics from the fp32 accumulators
  float ssum[TN], ssq[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ssum[tn] = ssq[tn] = 0.f;
  // input gradient fused with the producing unit's norm-backward reduction (ConvParams::ny; bf16 storage only): the two
  // channels of a lane are (2 l31, 2 l31 + 1) of the wave's 64-channel block
  float nsc[2], nsh[2], nmu[2], nrs[2];
  constexpr bool fuse = NBR;
  if constexpr (fuse) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t o = (int64_t)n_img * p.nsst + n0 + wn * 64 + 2 * l31 + j;
      nsc[j] = p.nsc[o]; nsh[j] = p.nsh[o]; nmu[j] = p.nmu[o]; nrs[j] = p.nrs[o];
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int sub = wm * TM + tm;
    uint32_t qv[16];
    if constexpr (fuse) {   // the producer's raw output at this fragment's pixels, loads issued together
      const bf16_t* nyb = static_cast<const bf16_t*>(p.ny) + (int64_t)n_img * p.H * p.W * p.nys + n0 + wn * 64 + 2 * l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = mfma32_row(r, h);
        const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
        qv[r] = (gh < p.H && gw < p.W) ? *reinterpret_cast<const uint32_t*>(nyb + ((int64_t)gh * p.W + gw) * p.nys) : 0u;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = mfma32_row(r, h);
      const int gh = h0 + 2 * sub + (i >> 4), gw = w0 + (i & 15);
      if (gh < p.H && gw < p.W) {
        if constexpr (BS) {   // channels (2 l31, 2 l31 + 1) of this wave's 64-channel block: one 4-byte store
          bf16_t* yb = reinterpret_cast<bf16_t*>(p.y) + yimg + ((int64_t)gh * p.W + gw) * p.ys + n0 + wn * 64 + 2 * l31;
          const float v0 = acc[tm][0][r], v1 = acc[tm][TN - 1][r];
          const uint32_t pk = pk_bf16(v0, v1);
          *reinterpret_cast<uint32_t*>(yb) = pk;
          if constexpr (fuse) {   // on the values memory holds (what a separate reduction pass would read back)
            const float d0 = unetk_bf16_lo(pk), d1 = unetk_bf16_hi(pk);
            const float q0 = unetk_bf16_lo(qv[r]), q1 = unetk_bf16_hi(qv[r]);
            const float du0 = fmaf(q0, nsc[0], nsh[0]) > 0.f ? d0 : 0.f, du1 = fmaf(q1, nsc[1], nsh[1]) > 0.f ? d1 : 0.f;
            ssum[0] += du0; ssq[0] += du0 * ((q0 - nmu[0]) * nrs[0]);
            ssum[TN - 1] += du1; ssq[TN - 1] += du1 * ((q1 - nmu[1]) * nrs[1]);
            continue;
          }
          ssum[0] += v0; ssq[0] += v0 * v0;
          ssum[TN - 1] += v1; ssq[TN - 1] += v1 * v1;
          continue;
        }
        float* yp = p.y + yimg + ((int64_t)gh * p.W + gw) * p.ys + n0 + wn * TN * 32 + l31;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          float v = acc[tm][tn][r];
          if (p.accumulate) v += yp[tn * 32];
          yp[tn * 32] = v;
          ssum[tn] += v;
          ssq[tn] += v * v;
        }
      }
    }
  }
  if (p.stat != nullptr) {
    float* red = reinterpret_cast<float*>(smem_q);  // [2][WM][BN]; behind the main loop's last barrier
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
    for (int i = tid; i < 2 * BN; i += NT) {
      const int k = i / BN, n = i - k * BN;
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < WM; ++m) s += red[(k * WM + m) * BN + n];
      const int nc = BS ? (n & ~63) + 2 * (n & 31) + ((n >> 5) & 1) : n;    // tile position -> channel
      p.stat[((int64_t)k * p.stat_rows + mtile) * p.Cout + n0 + nc] = s;
    }
  }
}

// Filter re-layout + rounding.  K8-interleaved panel: wp[t][q][n][j] = bf16(B_t[k = 8q + j][n]).
//   forward : B_t[k = ci][n = co] = w[t][ci][co]
//   dgrad   : B_t[k = co][n = ci] = w[8 - t][ci][co]
// perm (UNETK_BF16S): column position n' of every 64-column block holds output channel 2 (n' & 31) + (n' >> 5 & 1).
__global__ void pack_conv3x3_bf16_kernel(const float* __restrict__ w, int Cin, int Cout, uint4* __restrict__ wp_fwd,
                                         uint4* __restrict__ wp_dgrad, int perm) {
  const int64_t total = (int64_t)9 * Cin * Cout / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    unetk_pack::conv3x3_bf16(w, Cin, Cout, wp_fwd, wp_dgrad, perm, i);   // csrc/pack.h
}

struct BfCfg {
  int id, th;
};

// 0: 512x128 (8 waves)  1: 128x128  2: 256x64 (60 KB of LDS: two blocks per CU -- the Cout = 64 layers are
// HBM-bound, and a second resident block overlaps one block's halo load / output store with the other's MFMAs)
// 3: 128x64  4: 256x32
// (Measured: the tall 512 x 128 tile beats the 128 x 128 one even on grids it under-fills -- 8 x 32^2 pixels x 1024
// couts = 128 blocks on 256 CUs: 315 vs 277 slices/s at 512^2 bs 8 -- the bf16 kernels are bound by the L2 -> LDS stream,
// which the tall tile halves per MFMA; so the choice depends on H only.)
inline BfCfg pick_bf16(int H, int Cin, int Cout, int N = 1 << 20, int W = 1 << 10) {
  if (Cin % CKB != 0 || Cout % 32 != 0) return {-1, 8};
  if (Cout % 128 == 0) {
    if (H < 24) return BfCfg{1, 8};
    // the 32 x 32 level at batch 8: 512-pixel tiles give 64-128 blocks for 256 CUs; with bf16 storage (half the staging
    // bytes per MFMA) the 128-pixel tile's 4x larger grid wins there (measured in round 2; with fp32 storage it did not)
    const int64_t blocks = (int64_t)N * ((H + 31) / 32) * ((W + TW - 1) / TW) * (Cout / 128);
    if (blocks < 200) return BfCfg{1, 8};
    // (a 256 x 128 tile -- two blocks per CU instead of one -- was measured in round 2: 15.80 vs 15.65 ms per step, no gain)
    return BfCfg{0, 32};
  }
  if (Cout % 64 == 0) return H >= 12 ? BfCfg{2, 16} : BfCfg{3, 8};
  return {4, 16};
}

template <int WM, int WN, int TM, int TN, bool BS = false, bool NBR = false>
int launch_bf16(const ConvParams& p, int n_mtiles, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int TH = BM / TW;
  constexpr size_t lds = (size_t)(2 * (TH + 2) * HWD * PSQ + 2 * (CKB / 8) * BN) * 16;
  static_assert(lds >= 2 * WM * BN * sizeof(float), "stat scratch must fit");
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv3x3_igemm_bf16_kernel<WM, WN, TM, TN, BS, NBR>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  UNETK_LAUNCH(kern, dim3(n_mtiles * p.n_ntiles), dim3(WM * WN * 64), lds, st, p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

}  // namespace

bool unetk_conv_bf16_ok(int Cin, int Cout) { return Cin % CKB == 0 && Cout % 32 == 0; }

int unetk_conv_stat_rows_bf16(int N, int H, int W, int Cin, int Cout) {
  const BfCfg cfg = pick_bf16(H, Cin, Cout, N, W);
  if (cfg.id < 0) return UNETK_E_UNSUPPORTED;
  return N * ((H + cfg.th - 1) / cfg.th) * ((W + TW - 1) / TW);
}

// UNETK_BF16S: the round-3 kernel (conv_igemm_bf16s.hip) where its tile grid fills the chip, else the kernels of this file
int unetk_conv_stat_rows_bf16s(int N, int H, int W, int Cin, int Cout, int xs, int ys) {
  if (unetk_conv_bf16s_v3_ok(N, H, W, Cin, Cout, xs, ys)) return unetk_conv_bf16s_v3_stat_rows(N, H, W);
  return unetk_conv_stat_rows_bf16(N, H, W, Cin, Cout);
}

int unetk_conv_run_bf16(ConvParams p, hipStream_t st) {
  const BfCfg cfg = pick_bf16(p.H, p.Cin, p.Cout, p.N, p.W);
  if (cfg.id < 0) return UNETK_E_UNSUPPORTED;
  if (p.bf16 == UNETK_BF16S) {   // bf16 storage: 16-B halo pieces of 8 channels, 4-B output words of 2 channels
    if (cfg.id == 4 || p.accumulate) return UNETK_E_UNSUPPORTED;     // Cout % 64 != 0 / 3-D depth taps: not in this mode
    if (unetk_conv_bf16s_v3_ok(p.N, p.H, p.W, p.Cin, p.Cout, p.xs, p.ys) && (p.ny == nullptr || (p.Cout % 128 == 0 && p.nys % 8 == 0)))
      return unetk_conv_bf16s_v3_run(p, st);
    if (p.asc != nullptr) return UNETK_E_UNSUPPORTED;      // the inference epilogue lives in the persistent kernel only
    if (p.xs % 8 != 0 || p.ys % 2 != 0) return UNETK_E_BADARG;
    p.tiles_h = (p.H + cfg.th - 1) / cfg.th;
    p.tiles_w = (p.W + TW - 1) / TW;
    const int n_mt = p.N * p.tiles_h * p.tiles_w;
    p.stat_rows = n_mt;
    switch (cfg.id) {
      case 0:
        p.n_ntiles = p.Cout / 128;
        if (p.ny != nullptr) return launch_bf16<4, 2, 4, 2, true, true>(p, n_mt, st);
        return launch_bf16<4, 2, 4, 2, true>(p, n_mt, st);
      case 1:
        p.n_ntiles = p.Cout / 128;
        if (p.ny != nullptr) return launch_bf16<2, 2, 2, 2, true, true>(p, n_mt, st);
        return launch_bf16<2, 2, 2, 2, true>(p, n_mt, st);
      case 2:
        if (p.ny != nullptr) return UNETK_E_UNSUPPORTED;
        p.n_ntiles = p.Cout / 64;
        return launch_bf16<4, 1, 2, 2, true>(p, n_mt, st);
      default:
        if (p.ny != nullptr) return UNETK_E_UNSUPPORTED;
        p.n_ntiles = p.Cout / 64;
        return launch_bf16<4, 1, 1, 2, true>(p, n_mt, st);
    }
  }
  if (p.ny != nullptr) return UNETK_E_UNSUPPORTED;
  if (p.xs % 4 != 0) return UNETK_E_BADARG;
  p.tiles_h = (p.H + cfg.th - 1) / cfg.th;
  p.tiles_w = (p.W + TW - 1) / TW;
  const int n_mtiles = p.N * p.tiles_h * p.tiles_w;
  p.stat_rows = n_mtiles;
  switch (cfg.id) {
    case 0: p.n_ntiles = p.Cout / 128; return launch_bf16<4, 2, 4, 2>(p, n_mtiles, st);
    case 1: p.n_ntiles = p.Cout / 128; return launch_bf16<2, 2, 2, 2>(p, n_mtiles, st);
    case 2: p.n_ntiles = p.Cout / 64; return launch_bf16<4, 1, 2, 2>(p, n_mtiles, st);
    case 3: p.n_ntiles = p.Cout / 64; return launch_bf16<4, 1, 1, 2>(p, n_mtiles, st);
    default: p.n_ntiles = p.Cout / 32; return launch_bf16<4, 1, 2, 1>(p, n_mtiles, st);
  }
}

extern "C" int unetk_conv3x3_pack_bf16(const float* w_hwio, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                                       void* stream) {
  UNETK_REQUIRE(w_hwio && Cin > 0 && Cout > 0 && (wp_fwd || wp_dgrad));
  if (Cin % 8 != 0 || Cout % 8 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE((!wp_fwd || unetk_aligned16(wp_fwd)) && (!wp_dgrad || unetk_aligned16(wp_dgrad)));
  const int64_t total = (int64_t)9 * Cin * Cout / 8;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  UNETK_LAUNCH(pack_conv3x3_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_hwio, Cin, Cout,
                     (uint4*)wp_fwd, (uint4*)wp_dgrad, 0);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_conv3x3_pack_bf16s(const float* w_hwio, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                                        void* stream) {
  UNETK_REQUIRE(w_hwio && Cin > 0 && Cout > 0 && (wp_fwd || wp_dgrad));
  if (Cin % 64 != 0 || Cout % 64 != 0) return UNETK_E_UNSUPPORTED;     // the channel-pair permutation works on 64-blocks
  UNETK_REQUIRE((!wp_fwd || unetk_aligned16(wp_fwd)) && (!wp_dgrad || unetk_aligned16(wp_dgrad)));
  const int64_t total = (int64_t)9 * Cin * Cout / 8;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  UNETK_LAUNCH(pack_conv3x3_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_hwio, Cin, Cout,
                     (uint4*)wp_fwd, (uint4*)wp_dgrad, 1);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
