// conv3x3 (stride 1, SAME) forward / input-gradient for SMALL planes (W < 32): UNet3D's 24x24, 12x12 and 6x6
// levels (NetworksV2/UNet3D.py:153,165 via csrc/conv3d.hip) and any 2-D level below 32 pixels wide.
//
// The tiled kernel of conv_igemm.hip gives every block an 8 x 16 pixel tile of ONE plane, so a 12 x 12 plane fills
// 56 % of its MFMA rows and a 6 x 6 plane 28 %.  Here the M dimension is the LINEAR pixel index over the whole list of
// planes: block b owns pixels [128 b, 128 b + 128) whatever planes and rows they fall in, so every MFMA row is a real
// pixel (except in the last block).  The trick that keeps the inner loop identical to the tiled kernel -- one shifted
// ds_read_b128 per tap with a tap offset that is uniform across lanes -- is a VIRTUAL PADDED ROW SPACE: plane i, row r
// lives in padded row G = i (H + 2) + r + 1, each plane carrying one zero row above and below and one zero column left
// and right.  The block stages padded rows G0 .. G1 (all rows its pixels touch, +-1) into LDS, zero rows/columns
// included, and pixel (i, r, c) reads tap (kh, kw) at LDS slot ((G - 1 - G0 + kh) (W + 2) + c + kw): plane boundaries
// need no special case.  Same fp32 MFMA (v_mfma_f32_32x32x2_f32, exact), K4-interleaved filters, register prefetch,
// double-buffered LDS and statistics epilogue as the tiled kernel; same packed filters.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

__device__ float kZeroF4[4] = {};        // NOT const: a const array lives in the constant address space and selecting between it and a global pointer makes the load a FLAT load (lgkmcnt: every LDS read then waits for it)      // what a pad pixel of the halo reads

constexpr int CK = 16;
constexpr int PS = 20;
constexpr int LIN_MAXPIX = 416;   // staged padded pixels per block (host guarantees the bound)
// LIN_HALO_LD (round 5): the tap at which the NEXT chunk's halo is requested from global memory; it is written to LDS at tap 5.
// Rounds 2-4 requested it at tap 5 too, so every chunk's tap 5 ended in a full s_waitcnt on a load issued one tap earlier --
// and under stream-K these loads (and the filter panels: blocks start at unrelated K offsets, so the panels they stream do not
// meet in the XCD's L2; rocprofv3 FETCH_SIZE reads 820 MB per launch on UNet3D's 256-channel 12 x 12 layers against 35 MB of
// operands) come from the Infinity Cache / HBM.  Requested at tap 2 (as the tiled kernel has done since round 2): the 12 x 12
// and 24 x 24 layers run 7-8 % faster, UNet3D 96^3 at one patch 20.04 -> 19.51 ms; taps 0 / 1 / 3 the same within noise; the
// filter panel requested two steps ahead instead of one gives the same gain alone and nothing on top
// (profiles/r05_probe_lin_prefetch.txt, .patch).
#ifndef UNETK_LIN_HALO_LD
#define UNETK_LIN_HALO_LD 2
#endif
constexpr int LIN_HALO_LD = UNETK_LIN_HALO_LD;
static_assert(LIN_HALO_LD >= 0 && LIN_HALO_LD <= 5, "the halo is stored at tap 5");

// GEN = a SUBSET of the taps with an output scatter: the input gradient of a stride-2 conv, one output-parity class per
// launch.  dx[hi, wi] only receives taps with kh = (hi + pbh) mod 2 (+2), so class (ph, pw) is a stride-1 contraction of
// the UNDILATED dy over 4 / 2 / 2 / 1 taps (p.ntaps entries: halo offset + filter panel) whose result lands on every
// second dx pixel (p.os, p.ooh, p.oow; p.Hd x p.Wd = dx plane).  Exactly the 9 tap-products of the layer in total,
// instead of 36 on a zero-dilated dy.
// FUSED (with GEN): ONE block computes all four parity classes of its phase pixels -- four accumulator sets, the dy halo
// of a 16-channel chunk staged once and contracted with all 9 taps (class of tap (kh, kw) = (kh & 1, kw & 1), i.e.
// 4 + 2 + 2 + 1 taps in a fixed order) -- instead of four blocks that each re-stage the halo for 1..4 taps (the 1- and
// 2-tap classes were staging-bound).
//
// SK ("stream-K", plain variant only): the launch is a FIXED number of blocks (2 per CU) and the work is the linear
// sequence of (tile, K-chunk) pairs, tile-major; block b owns the contiguous range [b TOT / G, (b + 1) TOT / G) whatever
// tiles it crosses.  UNet3D at one or two patches per GPU has layers of 70 .. 900 tiles -- on 256 CUs a whole number of
// tiles per CU wastes up to a third of the machine (432 tiles: 176 CUs get two, 80 get one), and the 6 x 6 bridge
// (70 tiles, K = 8640) leaves most CUs idle.  A block that owns a whole tile runs the normal epilogue; the pieces of a
// split tile go to a slab (p.sk_slab, [tile][piece][BM][BN]) and lin_sk_fixup_kernel adds them in block order, writes y
// and the tile's statistic partials: fixed order, bit-reproducible, no atomics.
__device__ __forceinline__ int sk_block_of(int64_t pos, int64_t tot, int G) {   // the block b with s_b <= pos < s_{b+1}
  int b = (int)((pos * G) / tot);
  while (b + 1 < G && ((int64_t)(b + 1) * tot) / G <= pos) ++b;
  while (b > 0 && ((int64_t)b * tot) / G > pos) --b;
  return b;
}

// (Measured and dropped in round 2: fetching the filter panels two steps ahead instead of one -- no gain, the small tiles are
// not waiting for their weights -- and 8-wave variants of the same tiles.)
// ACC: the plain variant's epilogue accumulates (y += acc, one depth tap of an unfused 3-D conv).  Its own instantiation:
// the row pointers and old values it keeps in flight cost 60+ registers (the 128 x 128 stream-K variant ran at ONE wave
// per SIMD with them: 210 + 64 registers).
// GRP (plain variant, with or without SK; round 5): GROUPED TAPS -- the K sequence is p.ng groups x Cin / 16 chunks, every group
// with its own channel offset inside an input pixel, plane shift and list of <= 4 taps (ConvParams::ng ...): a strided conv
// on small output planes runs as a stride-1 contraction over a space-to-depth copy of its input (conv3d.hip).  UNet3D's
// (2,2,2) bridge (6 x 6 output planes) took the TILED stride-2 kernel at 28 % tile fill: 0.31 ms = 25 TFLOP/s.
// KHALF (round 5; plain variant): chunks flagged in p.khalf contract their lower 8 channels only (a 120-in-128 padded K axis).
template <int WM, int WN, int TM, int TN, bool GEN = false, bool FUSED = false, bool SK = false, bool ACC = false, bool GRP = false,
          bool KHALF = false>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(2))) void conv3x3_igemm_lin_kernel(ConvParams p) {
  static_assert(!(KHALF && (GEN || GRP)), "half chunks are a flag of the plain variant");
  static_assert(!(GRP && (GEN || ACC)), "grouped taps are a flag of the plain variant");
  static_assert(!(ACC && (GEN || SK)), "ACC is a flag of the plain variant (the GEN variants test p.accumulate)");
  static_assert(GEN || !FUSED, "FUSED is a GEN variant");
  static_assert(!(SK && GEN), "stream-K is for the plain variant");
  constexpr int NQ = FUSED ? 4 : 1;
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int HALO_F = p.lin_pix * PS;   // LDS is sized per launch for the plane shape: 2-3 resident blocks per CU
  constexpr int WB_F = CK * BN;
  constexpr int HR = (LIN_MAXPIX * 4 + NT - 1) / NT;
  constexpr int WR = (CK / 4 * BN + NT - 1) / NT;
  constexpr int WF4 = CK / 4 * BN;
  constexpr bool W_ALL = WF4 == WR * NT;      // every thread owns exactly WR float4 of a filter panel: no per-thread condition

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;               // [2][HALO_F]
  float* wbuf = smem + 2 * HALO_F;  // [2][WB_F]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int bid0 = SK ? 0 : xcd_remap(blockIdx.x, gridDim.x);
  // stream-K: the first p.sk_whole blocks take one whole tile each (as many full rounds of 256 as the layer has); the other
  // blocks share the REMAINING tiles' (tile, chunk) sequence evenly -- one pass of the loop below per tile a block touches
  const int sk_G = SK ? (int)gridDim.x - p.sk_whole : 1;
  const int64_t sk_tot = SK ? (int64_t)(p.sk_tiles - p.sk_whole) * p.sk_nc : 0;
  const int64_t sk_base = SK ? (int64_t)p.sk_whole * p.sk_nc : 0;
  int64_t sk_s = 0, sk_s1 = 0;
  int sk_b = 0;
  if (SK) {
    if ((int)blockIdx.x < p.sk_whole) {
      sk_s = (int64_t)xcd_remap(blockIdx.x, p.sk_whole) * p.sk_nc;
      sk_s1 = sk_s + p.sk_nc;
    } else {
      sk_b = xcd_remap(blockIdx.x - p.sk_whole, sk_G);
      sk_s = sk_base + ((int64_t)sk_b * sk_tot) / sk_G;
      sk_s1 = sk_base + ((int64_t)(sk_b + 1) * sk_tot) / sk_G;
    }
    if (sk_s >= sk_s1) return;
  }
  do {
  const int bid = SK ? (int)(sk_s / p.sk_nc) : bid0;
  const int c_lo = SK ? (int)(sk_s - (int64_t)bid * p.sk_nc) : 0;
  const int ntile = bid % p.n_ntiles;
  const int mtile_all = bid / p.n_ntiles;
  const int q = (GEN && !FUSED) ? (mtile_all & 3) : 0;          // parity class; its 4 blocks of a tile are neighbours (L2)
  // FUSED with p.dpar (depth stride 2, both depth parities of dx in ONE launch: twice the blocks for a layer whose 6 x 6 dy planes
  // give 108): pixel tiles [0, stat_rows) write the even dx planes from two fused depth taps (panels 0 and 2), tiles
  // [stat_rows, 2 stat_rows) the odd planes from the tap 1 alone
  const int dpar = (FUSED && p.dpar && mtile_all >= p.stat_rows) ? 1 : 0;
  const int mtile = (GEN && !FUSED) ? (mtile_all >> 2) : mtile_all - dpar * p.stat_rows;
  float* const ybase = p.y + (dpar ? p.dpar_yoff : 0);
  const int n0 = ntile * BN;

  const int HW = p.H * p.W, WP = p.W + 2, HP = p.H + 2;
  // blocks never straddle a statistics group (= the spg planes of one sample): per-sample statistics stay exact
  const int gpix = p.spg * HW;
  const int bpg = (gpix + BM - 1) / BM;
  const int grp = mtile / bpg, lb = mtile - grp * bpg;
  const int P0 = grp * gpix + lb * BM;
  const int P1 = min(P0 + BM, (grp + 1) * gpix);   // exclusive
  const int pl0 = P0 / HW, r0 = (P0 - pl0 * HW) / p.W;
  const int pl1 = (P1 - 1) / HW, r1 = (P1 - 1 - pl1 * HW) / p.W;
  const int G0 = pl0 * HP + r0;
  const int npix = (pl1 * HP + r1 + 2 - G0 + 1) * WP;

  int64_t hoff[HR];
  bool hok[HR];
  int hlds[HR];
  int hdep[HR];   // depth index of the staged plane inside its sample (fused depth taps)
#pragma unroll
  for (int r = 0; r < HR; ++r) {
    const int idx = tid + r * NT;
    const int slot = idx >> 2, q = idx & 3;
    const int gi = slot / WP, c = slot - gi * WP;
    const int g = G0 + gi;
    const int plane = g / HP, rr = g - plane * HP - 1, col = c - 1;
    hok[r] = slot < npix && plane < p.N && rr >= 0 && rr < p.H && col >= 0 && col < p.W;
    hoff[r] = p.xa.off(plane) + ((int64_t)rr * p.W + col) * p.xs + q * 4;
    hlds[r] = slot < npix ? slot * PS + q * 4 : -1;
    hdep[r] = plane % p.spg;
  }
  const int KD = GRP ? 1 : ((FUSED && p.dpar) ? (dpar ? 1 : 2) : (((!GEN || FUSED) && p.kd > 1) ? p.kd : 1));   // fused depth taps
  const int cin4 = p.Cin >> 2;
  int64_t woff[WR];
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    const int idx = min(tid + r * NT, WF4 - 1);      // threads beyond a small panel (WF4 < NT) request its last float4 again: no branch
    const int q = idx / BN, n = idx - q * BN;
    woff[r] = ((int64_t)q * p.Cout + n0 + n) * 4;
  }

  float4 hreg[HR], wreg[WR];
  const int nchunks = p.nlive ? p.nlive : p.Cin / CK;          // live chunks of a channel-padded K axis (common.h)
  // chunk index cc runs over (depth tap, 16-channel chunk); one depth tap = a plane shift inside the sample
  auto load_halo = [&](int cc) {
    const int dt = (GRP || KD > 1) ? cc / nchunks : 0, cl = cc - dt * nchunks;     // GRP: dt = the group
    const int c = p.nlive ? p.klive[cl] : cl;
    const int shift = GRP ? p.g_dz[dt] : (KD > 1 ? p.dshift0 + dt * p.dstep : 0);
    const int64_t soff = (int64_t)shift * p.xa.img_stride + c * CK + (GRP ? p.g_chan[dt] : 0);
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const bool ok = hok[r] && ((!GRP && KD == 1) || (hdep[r] + shift >= 0 && hdep[r] + shift < p.spg));
      hreg[r] = ldg4(ok ? p.x + hoff[r] + soff : kZeroF4);      // unconditional request; a pad pixel reads a page of zeros (see load_w)
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int r = 0; r < HR; ++r)
      if (hlds[r] >= 0) *reinterpret_cast<float4*>(&halo[buf * HALO_F + hlds[r]]) = hreg[r];
  };
  auto load_w = [&](int cc, int t) {      // GRP: t = the tap's position in its group's list
    const int dt = (GRP || KD > 1) ? cc / nchunks : 0, cl = cc - dt * nchunks;
#ifdef UNETK_LIN_PROBE_W0      // timing probe (wrong results): every block reads chunk 0 of the filter -- L2 hits whatever the K offsets
    const int c = 0 * cl;
#else
    const int c = p.nlive ? p.klive[cl] : cl;
#endif
    const int panel = GRP ? p.g_panel[dt][t] : ((FUSED && p.dpar) ? (dpar ? 9 + t : dt * 18 + t) : dt * 9 + t);
    const float* base = p.wp + ((int64_t)panel * cin4 + c * (CK / 4)) * p.Cout * 4;
    // Where every thread owns WR float4 of the panel (W_ALL: all but the 32-cout variant): NO per-thread condition around these
    // loads.  With one (`if (tid + r * NT < WF4)`, rounds 2-4) the loads sat in an exec-masked block and the compiler put the
    // s_waitcnt vmcnt(0) for them at the END OF THAT BLOCK -- in front of the tap's 32 MFMAs instead of in front of the LDS
    // write behind them: every tap of every chunk waited out the full latency of its filter panel (7-8 % of the 12 x 12 /
    // 24 x 24 layers of UNet3D, whose panels stream from beyond L2 under stream-K; found through rocprofv3's FETCH_SIZE and
    // a same-panel timing probe, profiles/r05_probe_lin_prefetch.txt; tests/test_asm_lint.py now checks the placement).
#pragma unroll
    for (int r = 0; r < WR; ++r)
      wreg[r] = ldg4(base + woff[r]);                 // unconditional (see woff)
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int r = 0; r < WR; ++r)
      if (W_ALL || tid + r * NT < WF4) *reinterpret_cast<float4*>(&wbuf[buf * WB_F + (tid + r * NT) * 4]) = wreg[r];
  };

  // A fragment base of this lane's pixel (clamped into the block's range; out-of-range rows are masked in the epilogue)
  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int pix = min(P0 + (wm * TM + tm) * 32 + l31, P1 - 1);
    const int plane = pix / HW, rem = pix - plane * HW;
    const int hh = rem / p.W, ww = rem - hh * p.W;
    abase[tm] = ((plane * HP + hh - G0) * WP + ww) * PS + 4 * h;
  }
  const int bbase = (h * BN + wn * TN * 32 + l31) * 4;

  f32x16 accq[NQ][TM][TN];
  f32x16 (&acc)[TM][TN] = accq[0];
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) accq[qq][tm][tn][r] = 0.f;

  const int nchunks_all = (GRP ? p.ng : KD) * nchunks;
  const int c_hi = SK ? min(nchunks_all, c_lo + (int)(sk_s1 - sk_s)) : nchunks_all;
  load_halo(c_lo);
  load_w(c_lo, GEN ? p.tap_panel[q][0] : 0);
  store_halo(0);
  store_w(0);
  __syncthreads();

  if constexpr (GEN && FUSED) {
    constexpr int NTAPS[4] = {4, 2, 2, 1}, KBASE[4] = {0, 4, 6, 8};
    int step = 0;
    for (int c = 0; c < KD * nchunks; ++c) {                  // (depth tap, 16-channel chunk) pairs
      const float* hb = halo + (c & 1) * HALO_F;
      const bool more_chunks = c + 1 < KD * nchunks;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
        for (int ti = 0; ti < NTAPS[qq]; ++ti, ++step) {
          const int k = KBASE[qq] + ti;                      // position in the chunk's 9-step schedule (static)
          const bool last = k == 8;
          const int nq = ti + 1 < NTAPS[qq] ? qq : (qq + 1) & 3, nti = ti + 1 < NTAPS[qq] ? ti + 1 : 0;
          const bool has_next = !last || more_chunks;
          if (has_next) load_w(last ? c + 1 : c, p.tap_panel[nq][nti]);
          if (k == LIN_HALO_LD && more_chunks) load_halo(c + 1);
          __builtin_amdgcn_sched_barrier(0);
          const float* wb = wbuf + (step & 1) * WB_F;
          const int toff = ((p.tap_off[qq][ti] >> 2) * WP + (p.tap_off[qq][ti] & 3)) * PS;
#pragma unroll
          for (int g = 0; g < CK / 8; ++g) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const float4*>(&hb[abase[tm] + toff + 8 * g]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const float4*>(&wb[bbase + (2 * g * BN + tn * 32) * 4]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) {
                accq[qq][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, accq[qq][tm][tn], 0, 0, 0);
                accq[qq][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, accq[qq][tm][tn], 0, 0, 0);
                accq[qq][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, accq[qq][tm][tn], 0, 0, 0);
                accq[qq][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, accq[qq][tm][tn], 0, 0, 0);
              }
          }
          if (has_next) store_w((step + 1) & 1);
          if (k == 5 && more_chunks) store_halo((c + 1) & 1);
          __syncthreads();
        }
      }
    }
    // scatter epilogue, one class after the other: phase pixel (a, b) -> dx pixel (os a + ooh[q], os b + oow[q])
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int pb = P0 + (wm * TM + tm) * 32 + 4 * h;
        int plane = pb / HW;
        const int rem0 = pb - plane * HW;
        int aa = rem0 / p.W, bb = rem0 - aa * p.W, prev = 0;
        int64_t img = p.ya.off(plane);
        // four rows at a time: their old values (y += acc) are loaded together, then added and stored.  All sixteen at once
        // (pointers + values = 64 more registers next to the 4 x 32 accumulators) held this kernel at ONE wave per SIMD.
#pragma unroll
        for (int rb = 0; rb < 16; rb += 4) {
          float* rowp[4];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int r = rb + r4;
            const int delta = (r & 3) + 8 * (r >> 2);
            bb += delta - prev;
            prev = delta;
            while (bb >= p.W) {
              bb -= p.W;
              if (++aa == p.H) {
                aa = 0;
                img = p.ya.off(++plane);
              }
            }
            const int hi = p.os * aa + p.ooh[qq], wi = p.os * bb + p.oow[qq];
            const bool ok = pb + delta < P1 && hi >= 0 && hi < p.Hd && wi >= 0 && wi < p.Wd;
            rowp[r4] = ok ? ybase + img + ((int64_t)hi * p.Wd + wi) * p.ys + n0 + wn * TN * 32 + l31 : nullptr;
          }
          float prior[4][TN];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) prior[r4][tn] = (p.accumulate && rowp[r4]) ? rowp[r4][tn * 32] : 0.f;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            if (rowp[r4]) {
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) rowp[r4][tn * 32] = accq[qq][tm][tn][rb + r4] + prior[r4][tn];
            }
        }
      }
    }
    return;
  } else if constexpr (GEN) {
    const int ntaps = p.ntaps[q], nsteps = nchunks * ntaps;
    int c = 0, ti = 0;
    for (int s = 0; s < nsteps; ++s) {
      const bool has_next = s + 1 < nsteps;
      const int ti2 = ti + 1 < ntaps ? ti + 1 : 0, c2 = ti + 1 < ntaps ? c : c + 1;
      if (has_next) load_w(c2, p.tap_panel[q][ti2]);
      const bool more_chunks = c + 1 < nchunks;
      if (ti == 0 && more_chunks) load_halo(c + 1);
      __builtin_amdgcn_sched_barrier(0);
      const float* hb = halo + (c & 1) * HALO_F;
      const float* wb = wbuf + (s & 1) * WB_F;
      const int toff = ((p.tap_off[q][ti] >> 2) * WP + (p.tap_off[q][ti] & 3)) * PS;
#pragma unroll
      for (int g = 0; g < CK / 8; ++g) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const float4*>(&hb[abase[tm] + toff + 8 * g]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const float4*>(&wb[bbase + (2 * g * BN + tn * 32) * 4]);
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
      if (has_next) store_w((s + 1) & 1);
      if (ti == ntaps - 1 && more_chunks) store_halo((c + 1) & 1);
      __syncthreads();
      ti = ti2;
      c = c2;
    }
    // scatter epilogue: phase pixel (a, b) -> dx pixel (os a + ooh, os b + oow)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      // (plane, row, column) of the fragment's first row by division, then advanced incrementally: a division per
      // stored row costs as much VALU time as a good part of the main loop on the small-K layers
      const int pb = P0 + (wm * TM + tm) * 32 + 4 * h;
      int plane = pb / HW;
      const int rem0 = pb - plane * HW;
      int aa = rem0 / p.W, bb = rem0 - aa * p.W, prev = 0;
      int64_t img = p.ya.off(plane);
      float* rowp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int delta = (r & 3) + 8 * (r >> 2);   // mfma32_row(r, h) - 4 h
        bb += delta - prev;
        prev = delta;
        while (bb >= p.W) {
          bb -= p.W;
          if (++aa == p.H) {
            aa = 0;
            img = p.ya.off(++plane);
          }
        }
        const int hi = p.os * aa + p.ooh[q], wi = p.os * bb + p.oow[q];
        const bool ok = pb + delta < P1 && hi >= 0 && hi < p.Hd && wi >= 0 && wi < p.Wd;
        rowp[r] = ok ? p.y + img + ((int64_t)hi * p.Wd + wi) * p.ys + n0 + wn * TN * 32 + l31 : nullptr;
      }
      float prior[16][TN];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) prior[r][tn] = (p.accumulate && rowp[r]) ? rowp[r][tn * 32] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rowp[r]) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) rowp[r][tn * 32] = acc[tm][tn][r] + prior[r][tn];
        }
    }
    return;
  }

  int step = 0;
  if constexpr (GRP) {
    for (int c = c_lo; c < c_hi; ++c) {
      const float* hb = halo + ((c - c_lo) & 1) * HALO_F;
      const bool more_chunks = (c + 1 < c_hi);
      const int g = c / nchunks, nt = p.g_ntaps[g];
      for (int ti = 0; ti < nt; ++ti, ++step) {
        const bool last = ti == nt - 1;
        const bool has_next = !last || more_chunks;
        if (has_next) load_w(last ? c + 1 : c, last ? 0 : ti + 1);
        if (ti == 0 && more_chunks) load_halo(c + 1);          // at the group's first tap, written at its last
        __builtin_amdgcn_sched_barrier(0);
        const float* wb = wbuf + (step & 1) * WB_F;
        const int toff = ((p.g_off[g][ti] >> 2) * WP + (p.g_off[g][ti] & 3)) * PS;
#pragma unroll
        for (int gg = 0; gg < CK / 8; ++gg) {
          float4 a[TM], b[TN];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const float4*>(&hb[abase[tm] + toff + 8 * gg]);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const float4*>(&wb[bbase + (2 * gg * BN + tn * 32) * 4]);
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
        if (last && more_chunks) store_halo((c + 1 - c_lo) & 1);
        __syncthreads();
      }
    }
  } else {
  unsigned hmask = 0;        // p.khalf rotated to the current chunk's place in the live list (KHALF)
  int hleft = 0;
  if constexpr (KHALF) {
    const int cl0 = c_lo % nchunks;
    hmask = p.khalf >> cl0;
    hleft = nchunks - cl0;
  }
  for (int c = c_lo; c < c_hi; ++c) {
    const float* hb = halo + ((c - c_lo) & 1) * HALO_F;
    const bool more_chunks = (c + 1 < c_hi);
    // a chunk whose upper 8 channels are padding (p.khalf) skips the second half of every tap's MFMAs
    const bool half = KHALF && (hmask & 1);
    if constexpr (KHALF) {
      hmask >>= 1;
      if (--hleft == 0) { hmask = p.khalf; hleft = nchunks; }      // next depth tap: the list starts over
    }
#pragma unroll
    for (int t = 0; t < 9; ++t, ++step) {
      const bool has_next = (t < 8) || more_chunks;
      if (has_next) load_w(t < 8 ? c : c + 1, t < 8 ? t + 1 : 0);
      if (t == LIN_HALO_LD && more_chunks) load_halo(c + 1);
      __builtin_amdgcn_sched_barrier(0);       // the requests stay HERE, in front of the tap's MFMAs (the scheduler sinks them to their LDS write otherwise)

      const float* wb = wbuf + (step & 1) * WB_F;
      const int toff = ((t / 3) * WP + (t % 3)) * PS;
#pragma unroll
      for (int g = 0; g < CK / 8; ++g) {
        if (KHALF && g == 1 && half) continue;
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
      if (t == 5 && more_chunks) store_halo((c + 1 - c_lo) & 1);
      __syncthreads();
    }
  }
  }

  if (SK && (c_lo != 0 || c_hi != nchunks_all)) {
    // a piece of a split tile: raw accumulators to the slab, [tile][piece = block - first block of the tile][BM][BN]
    const int piece = sk_b - sk_block_of((int64_t)bid * p.sk_nc - sk_base, sk_tot, sk_G);
    float* sl = p.sk_slab + ((int64_t)(bid - p.sk_whole) * p.sk_maxp + piece) * (BM * BN);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* rp = sl + ((wm * TM + tm) * 32 + mfma32_row(r, h)) * BN + wn * TN * 32 + l31;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) rp[tn * 32] = acc[tm][tn][r];
      }
    sk_s += c_hi - c_lo;
    continue;
  }

  float ssum[TN], ssq[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ssum[tn] = ssq[tn] = 0.f;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int pb = P0 + (wm * TM + tm) * 32 + 4 * h;   // first row of the fragment; advanced incrementally (see above)
    int plane = pb / HW;
    int rem = pb - plane * HW, prev = 0;
    int64_t img = p.ya.off(plane);
    if constexpr (!ACC) {
      // plain store: nothing to keep in flight -- each row's address is formed and used at once
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int delta = (r & 3) + 8 * (r >> 2);
        rem += delta - prev;
        prev = delta;
        while (rem >= HW) {
          rem -= HW;
          img = p.ya.off(++plane);
        }
        if (pb + delta < P1) {
          float* rp = p.y + img + (int64_t)rem * p.ys + n0 + wn * TN * 32 + l31;
          if (p.asc != nullptr) {      // inference epilogue (unetk_conv3x3_fwd_affine): the activation, no statistics
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              const int ch = n0 + (wn * TN + tn) * 32 + l31;
              rp[tn * 32] = fmaxf(fmaf(acc[tm][tn][r], p.asc[ch], p.ash[ch]), 0.f);
            }
          } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
              const float v = acc[tm][tn][r];
              rp[tn * 32] = v;
              ssum[tn] += v;
              ssq[tn] += v * v;
            }
          }
        }
      }
    } else {
      float* rowp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int delta = (r & 3) + 8 * (r >> 2);
        rem += delta - prev;
        prev = delta;
        while (rem >= HW) {
          rem -= HW;
          img = p.ya.off(++plane);
        }
        rowp[r] = pb + delta < P1 ? p.y + img + (int64_t)rem * p.ys + n0 + wn * TN * 32 + l31 : nullptr;
      }
      // y += acc: all old values first, so the load latencies overlap (see conv_igemm.hip)
      float prior[16][TN];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) prior[r][tn] = rowp[r] ? rowp[r][tn * 32] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (rowp[r]) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const float v = acc[tm][tn][r] + prior[r][tn];
            rowp[r][tn * 32] = v;
            ssum[tn] += v;
            ssq[tn] += v * v;
          }
        }
      }
    }
  }
  if (p.stat != nullptr) {
    float* red = smem;  // [2][WM][BN]; behind the main loop's last barrier
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
      p.stat[((int64_t)k * p.stat_rows + mtile) * p.Cout + n0 + n] = s;
    }
  }
  if (SK) {
    sk_s += c_hi - c_lo;
    __syncthreads();   // the statistic scratch aliases the next tile's staging buffers
  }
  } while (SK && sk_s < sk_s1);
}

// Stream-K fix-up: one block per remainder tile; a tile that one block computed whole is already written.  Otherwise y = the sum of
// the tile's pieces in block order (+ the statistic partials of the sums).  Thread = (channel quad, row lane).
template <int BM, int BN>
__global__ __launch_bounds__(256) void lin_sk_fixup_kernel(ConvParams p, int G) {
  constexpr int CQ = BN / 4, RL = 256 / CQ;
  __shared__ float4 red[2][RL][CQ];
  const int tr = blockIdx.x, tt = p.sk_whole + tr;          // the tr-th remainder tile
  const int64_t tot = (int64_t)(p.sk_tiles - p.sk_whole) * p.sk_nc;
  const int bf = sk_block_of((int64_t)tr * p.sk_nc, tot, G), bl = sk_block_of((int64_t)(tr + 1) * p.sk_nc - 1, tot, G);
  const int np = bl - bf + 1;
  if (np == 1) return;
  const int ntile = tt % p.n_ntiles, mtile = tt / p.n_ntiles;
  const int n0 = ntile * BN;
  const int HW = p.H * p.W;
  const int gpix = p.spg * HW;
  const int bpg = (gpix + BM - 1) / BM;
  const int grp = mtile / bpg, lb = mtile - grp * bpg;
  const int P0 = grp * gpix + lb * BM;
  const int P1 = min(P0 + BM, (grp + 1) * gpix);
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const float* sl = p.sk_slab + (int64_t)tr * p.sk_maxp * (BM * BN);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  for (int row = rl; row < BM && P0 + row < P1; row += RL) {
    float4 v = ldg4(sl + row * BN + cq * 4);
    for (int pc = 1; pc < np; ++pc) {
      const float4 u = ldg4(sl + (int64_t)pc * (BM * BN) + row * BN + cq * 4);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int pix = P0 + row;
    const int plane = pix / HW, rem = pix - plane * HW;
    if (p.asc != nullptr) {            // inference epilogue on the summed pieces
      const float4 sc = ldg4(p.asc + n0 + cq * 4), sh = ldg4(p.ash + n0 + cq * 4);
      v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
      v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    }
    stg4(p.y + p.ya.off(plane) + (int64_t)rem * p.ys + n0 + cq * 4, v);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
  }
  if (p.stat == nullptr) return;
  red[0][rl][cq] = s;
  red[1][rl][cq] = sq;
  __syncthreads();
  if (threadIdx.x < 2 * CQ) {
    const int k = threadIdx.x / CQ, c = threadIdx.x % CQ;
    float4 a = red[k][0][c];
#pragma unroll
    for (int j = 1; j < RL; ++j) {
      const float4 u = red[k][j][c];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
    stg4(p.stat + ((int64_t)k * p.stat_rows + mtile) * p.Cout + n0 + c * 4, a);
  }
}

template <int WM, int WN, int TM, int TN, bool GEN = false, bool FUSED = false, bool ACC = false, bool GRP = false, bool KHALF = false>
int launch_lin(const ConvParams& p, int n_mtiles, hipStream_t st) {
  if constexpr (!GEN && !ACC && !GRP) {
    if (p.accumulate) return launch_lin<WM, WN, TM, TN, false, false, true>(p, n_mtiles, st);   // (half chunks run whole there)
  }
  constexpr int BN = WN * TN * 32;
  constexpr size_t lds_max = (size_t)(2 * LIN_MAXPIX * PS + 2 * CK * BN) * sizeof(float);
  const size_t lds = (size_t)(2 * p.lin_pix * PS + 2 * CK * BN) * sizeof(float);
  auto kern = conv3x3_igemm_lin_kernel<WM, WN, TM, TN, GEN, FUSED, false, ACC, GRP, KHALF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  UNETK_LAUNCH(kern, dim3(n_mtiles * p.n_ntiles), dim3(WM * WN * 64), lds, st, p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

constexpr int LIN_BM = 128;

}  // namespace

// Use the linear-pixel kernel when the tiled one would leave > 10 % of its MFMA rows empty and the padded rows a
// 128-pixel block can touch fit the LDS budget.
// spg = planes per statistics group (one sample of a 3-D tensor; 1 for 2-D images).
// UNETK_LIN_2D (round 5; read once): bit 0 = 32-wide planes whose tiled grid cannot fill the chip take this kernel too
// (the 32 x 32 level of a 2-D net at 8 slices per GPU: 512 blocks of 64 pixels x 128 couts at ~120 TFLOP/s in the tiled kernel),
// bits 1..3 = lin_tune's bits 0..2 for 2-D planes (128-pixel blocks, stream-K over all tiles, stream-K for short K).
// Default 6 = 128-pixel blocks + stream-K over all tiles: the 16 x 16 bridge of GUNet at 8 slices (128 tiles of 128 x 128 cut into
// 512 K pieces instead of 256 whole tiles of 64 x 128) 93 -> 107 TFLOP/s, step 370.3 -> 373.3 slices/s in two interleaved runs;
// 2 alone 372.3, 14 373.2, bit 0 on top (15) 366.6 (profiles/r05_gunet_lin2d_ab.txt).
static int lin_2d() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("UNETK_LIN_2D");
    v = e ? atoi(e) : 6;
  }
  return v;
}

bool unetk_conv_lin_ok(int N, int H, int W, int Cin, int Cout, int spg) {
  const int wmax = (lin_2d() & 1) && spg == 1 ? 33 : 32;
  if (Cin % CK != 0 || Cout % 64 != 0 || W >= wmax || spg < 1 || N % spg != 0) return false;
  const int64_t gpix = (int64_t)spg * H * W;
  const int64_t lin = (gpix + LIN_BM - 1) / LIN_BM * LIN_BM;                                   // MFMA rows issued per group
  const int64_t tiled = (int64_t)spg * ((H + 7) / 8) * 8 * ((W + 15) / 16) * 16;
  // ... or when the tiled kernel's grid cannot fill the chip (the 16 x 16 bridge of a 2-D net at 8 slices per GPU: 128 or
  // 256 blocks of 64 pixels x 128 couts for 256 CUs, 47-100 TFLOP/s): this kernel has the stream-K schedule
  const int64_t tiled_blocks = (int64_t)(N / spg) * tiled / 64 * (Cout / (Cout % 128 == 0 ? 128 : 64));
  const bool starved = Cout % 128 == 0 && tiled_blocks <= (((lin_2d() & 1) && spg == 1) ? 512 : 256) && gpix % 64 == 0;
  if (lin * 10 > tiled * 9 && !starved) return false;
  const int rows = (LIN_BM + W - 1) / W + 1 + 2 + 2 * ((LIN_BM + H * W - 1) / (H * W));
  return rows * (W + 2) <= LIN_MAXPIX;
}

// 64-pixel blocks when 128-pixel blocks x 128 couts would give fewer than 1.5 blocks per CU (deep levels at small batch)
// Scheduling switches of the plain variant (UNETK_LIN_TUNE overrides, measurement only, read once): bit 0 = always 128-pixel
// blocks, bit 1 = stream-K over ALL tiles (no whole-tile rounds) when at most 1024 tiles do not fill whole rounds, bit 2 =
// stream-K also for short K (24 chunks).  Round 4, UNet3D 96^3 in one call: all off 20.81 / 38.32 ms (one / two patches), bit 0
// alone 20.70 / 38.78, bits 0 + 1 20.71 / 38.46, all three 20.51 / 38.02 -- the default.  (64-pixel blocks stage twice the filter
// panel per MFMA -- 72 % matrix-pipe busy against 82 % for the 128-pixel stream-K variant, profiles/r04_pmc_mfma_busy_unet3d.txt --
// and with every tile cut into K pieces the block count no longer has to fit the CU count.)  Applied to 3-D layers (spg > 1)
// first; round 5 gave the 2-D nets' starved small planes bits 0 + 1 as well (lin_2d() above).
static int lin_tune(int N, int H, int W, int spg) {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("UNETK_LIN_TUNE");
    v = e ? atoi(e) : 7;
  }
  if (spg > 1) return v;
  // 2-D planes: only those the tiled kernel would FILL (they are here because its grid starves the chip, unetk_conv_lin_ok) --
  // with at least 16 blocks of 128 pixels (8 slices of 16 x 16), where the change was measured.  The badly filled small planes
  // (8 x 8, 4 x 4 ...) and the two-slice batches of the end-to-end tests keep the round-2 schedule: same kernel-level error
  // either way (tools/dbg/lin2d_check.py: 0.5-2e-6 of the output's range against float64), but another fp32 summation order
  // flips other ReLU masks at a 4 x 4 bridge, and those tests' budgets were set on the old one
  const int64_t lin = ((int64_t)H * W + LIN_BM - 1) / LIN_BM * LIN_BM, tiled = (int64_t)((H + 7) / 8) * 8 * ((W + 15) / 16) * 16;
  return (lin * 10 > tiled * 9 && (int64_t)N * H * W >= 16 * LIN_BM) ? (lin_2d() >> 1) : 0;
}

static int lin_bm(int N, int H, int W, int Cout, int spg) {
  if (Cout % 128 != 0) return LIN_BM;
  if (lin_tune(N, H, W, spg) & 1) return LIN_BM;
  const int64_t nt = Cout / 128, groups = N / spg, gpix = (int64_t)spg * H * W;
  const int64_t b128 = groups * ((gpix + 127) / 128) * nt, b64 = groups * ((gpix + 63) / 64) * nt;
  if (b128 < 384) return 64;
  // ... and when the grid quantises badly over the 256 CUs: 864 blocks of 128 pixels are 3.4 per CU = 84 % of four full
  // rounds, 1728 blocks of 64 pixels 96 % of seven (measured on UNet3D's 24^2 levels: 108 -> 117 TFLOP/s; the 5 % margin
  // is what the smaller block costs in filter-panel reuse)
  auto eff = [](int64_t b) { return (double)b / (double)(((b + 255) / 256) * 256); };
  return eff(b64) * 0.95 > eff(b128) ? 64 : LIN_BM;
}

int unetk_conv_stat_rows_lin(int N, int H, int W, int spg, int Cout) {
  const int bm = Cout > 0 ? lin_bm(N, H, W, Cout, spg) : LIN_BM;
  return (N / spg) * (int)(((int64_t)spg * H * W + bm - 1) / bm);
}

// Padded rows a block of bm linear pixels may touch.  General case: ceil(bm / W) + 1 pixel rows (a block may start mid-row),
// the halo row above and below, and an extra pair of zero rows per plane boundary it crosses.  ALIGNED case (round 4: the
// 64- / 128-wide dy planes of UNet3D at the reference's 10 x 256 x 256 training shape): when W divides bm and bm divides the
// plane, every block is whole rows of ONE plane -- bm / W rows + the two halo rows, which are the plane's own pad rows at its
// edges.  (Those shapes used to fail the LDS bound and fell back to a zero-dilated dy: four times the MFMA work.)
static int lin_rows_bound(int H, int W, int bm = LIN_BM) {
  if (bm % W == 0 && (H * W) % bm == 0) return bm / W + 2;
  return (bm + W - 1) / W + 1 + 2 + 2 * ((bm + H * W - 1) / (H * W));
}

// Tap-subset / scatter variant (input gradient of a stride-2 conv): p.H x p.W = dy plane, p.Cin = dy channels,
// p.Cout = dx channels, p.ntaps / tap_off / tap_panel / os / ooh / oow / Hd / Wd set by the caller.
bool unetk_conv_lin_gen_ok(int H, int W, int Cin, int Cout) {
  const int bm = Cout % 128 == 0 ? 64 : LIN_BM;      // the four-class kernel's pixel block (unetk_conv_run_lin_gen)
  return Cin % CK == 0 && Cout % 32 == 0 && lin_rows_bound(H, W, bm) * (W + 2) <= LIN_MAXPIX;
}

int unetk_conv_run_lin_gen(ConvParams p, hipStream_t st) {
  if (!unetk_conv_lin_gen_ok(p.H, p.W, p.Cin, p.Cout) || p.xs % 4 != 0) return UNETK_E_UNSUPPORTED;
  for (int q = 0; q < 4; ++q)
    if (p.ntaps[q] < 1 || p.ntaps[q] > 4) return UNETK_E_BADARG;
  if (p.spg < 1) p.spg = 1;
  p.stat = nullptr;
  if (p.ntaps[0] == 4 && p.ntaps[1] == 2 && p.ntaps[2] == 2 && p.ntaps[3] == 1) {
    // all four classes in one block: 64-pixel blocks x 128 couts, or 128-pixel blocks x 64 couts (128 accumulator
    // registers either way)
    const int bm = p.Cout % 128 == 0 ? 64 : 128;
    const int n_mt = (p.N / p.spg) * (int)(((int64_t)p.spg * p.H * p.W + bm - 1) / bm);
    p.stat_rows = n_mt;
    p.lin_pix = lin_rows_bound(p.H, p.W, bm) * (p.W + 2);
    if (p.Cout % 128 == 0) {
      if ((int64_t)n_mt * (p.dpar ? 2 : 1) * (p.Cout / 128) < 128) {   // UNet3D's (2,2,2) layer: 6 x 6 dy planes = 27 pixel tiles -- 64 x 64 tiles
        p.n_ntiles = p.Cout / 64;                   // double the blocks (54 -> 108 per depth tap: 0.49 -> 0.42 ms; the short
        return launch_lin<2, 2, 1, 1, true, true>(p, n_mt * (p.dpar ? 2 : 1), st);   // steps of such tiles are latency-bound, not MFMA-bound)
      }
      p.n_ntiles = p.Cout / 128;
      return launch_lin<2, 2, 1, 2, true, true>(p, n_mt * (p.dpar ? 2 : 1), st);
    }
    if (p.Cout % 64 == 0) {
      p.n_ntiles = p.Cout / 64;
      return launch_lin<4, 1, 1, 2, true, true>(p, n_mt * (p.dpar ? 2 : 1), st);
    }
    p.n_ntiles = p.Cout / 32;                                     // UNet3D conv_e1/conv1: dx has 32 channels
    return launch_lin<4, 1, 1, 1, true, true>(p, n_mt * (p.dpar ? 2 : 1), st);
  }
  if (p.Cout % 64 != 0 || p.kd > 1) return UNETK_E_UNSUPPORTED;   // fused depth taps exist in the four-class variant only
  if (lin_rows_bound(p.H, p.W) * (p.W + 2) > LIN_MAXPIX) return UNETK_E_UNSUPPORTED;
  const int n_mtiles = unetk_conv_stat_rows_lin(p.N, p.H, p.W, p.spg, 0);
  p.stat_rows = n_mtiles;
  p.lin_pix = lin_rows_bound(p.H, p.W) * (p.W + 2);
  if (p.Cout % 128 == 0) {
    p.n_ntiles = p.Cout / 128;
    return launch_lin<2, 2, 2, 2, true>(p, 4 * n_mtiles, st);     // x4: the parity classes
  }
  p.n_ntiles = p.Cout / 64;
  return launch_lin<4, 1, 1, 2, true>(p, 4 * n_mtiles, st);
}

namespace {

// Stream-K plan: used when whole tiles quantise badly over the CUs (or do not fill them) and the K loop is long enough to
// cut.  G = 2 blocks per CU (fewer when the layer has fewer than 4 chunks per block).
struct SkPlan {
  bool on;
  int G, whole, tiles, nc, maxp, bm, bn;     // G = blocks sharing the remainder tiles
  size_t bytes;
};

SkPlan sk_plan(int N, int H, int W, int Cin, int Cout, int spg, int kd) {
  SkPlan s{};
  const int bm = lin_bm(N, H, W, Cout, spg);
  const int bn = (bm == 64 || Cout % 128 == 0) ? 128 : 64;
  const int n_mt = (N / spg) * (int)(((int64_t)spg * H * W + bm - 1) / bm);
  s.bm = bm; s.bn = bn;
  s.tiles = n_mt * (Cout / bn);
  s.nc = (kd > 1 ? kd : 1) * (Cin / CK);
  const double eff = (double)s.tiles / (double)(((s.tiles + 255) / 256) * 256);
  // (up to one tile per CU the K range is always split: a lone 4-wave block per CU runs at ~100 TFLOP/s, two half-K blocks
  // at more)
  if (s.tiles >= 2048 || (s.tiles > 256 && eff >= 0.92)) return s;
  s.whole = s.tiles <= 256 ? 0 : s.tiles / 256 * 256;    // full rounds run one whole tile per block
  if ((lin_tune(N, H, W, spg) & 2) && s.tiles <= 1024) s.whole = 0;
  const int rem = s.tiles - s.whole;
  // measured (UNet3D, one patch): a remainder of 176 tiles with K >= 48 chunks gains 10-12 %, 96 tiles or K = 24 chunks
  // do not pay for the slab round trip and the fix-up launch
  if (s.whole > 0 && (rem < 128 || s.nc < 48)) return s;
  if (s.whole == 0 && s.tiles > 256 && s.nc < 48 && !(lin_tune(N, H, W, spg) & 4)) return s;
  const int64_t tot = (int64_t)rem * s.nc;
  int G = 256;
  if (s.whole == 0 && tot >= 512 * 4) G = 512;           // nothing else resident: two blocks per CU
  while (G > 32 && tot < (int64_t)G * 4) G >>= 1;
  if (tot < (int64_t)G * 4) return s;
  s.G = G;
  const int64_t minlen = tot / G;
  s.maxp = (int)((s.nc + minlen - 1) / minlen) + 1;
  s.bytes = (size_t)rem * s.maxp * bm * bn * sizeof(float);
  s.on = s.bytes <= ((size_t)1 << 30);
  return s;
}

template <int WM, int WN, int TM, int TN, bool GRP = false, bool KHALF = false>
int launch_lin_sk(const ConvParams& p, const SkPlan& sk, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds_max = (size_t)(2 * LIN_MAXPIX * PS + 2 * CK * BN) * sizeof(float);
  const size_t lds = (size_t)(2 * p.lin_pix * PS + 2 * CK * BN) * sizeof(float);
  auto kern = conv3x3_igemm_lin_kernel<WM, WN, TM, TN, false, false, true, false, GRP, KHALF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  UNETK_LAUNCH(kern, dim3(sk.whole + sk.G), dim3(WM * WN * 64), lds, st, p);
  UNETK_LAUNCH_CHECK();
  UNETK_LAUNCH((lin_sk_fixup_kernel<BM, BN>), dim3(sk.tiles - sk.whole), dim3(256), 0, st, p, sk.G);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

}  // namespace

size_t unetk_conv_lin_sk_bytes(int N, int H, int W, int Cin, int Cout, int spg, int kd) {
  if (!unetk_conv_lin_ok(N, H, W, Cin, Cout, spg)) return 0;
  const SkPlan sk = sk_plan(N, H, W, Cin, Cout, spg, kd);
  return sk.on ? sk.bytes : 0;
}

int unetk_conv_run_lin(ConvParams p, hipStream_t st) {
  const int n_mtiles = unetk_conv_stat_rows_lin(p.N, p.H, p.W, p.spg, p.Cout);
  const int bm = lin_bm(p.N, p.H, p.W, p.Cout, p.spg);
  p.stat_rows = n_mtiles;
  p.tiles_h = p.tiles_w = 0;
  p.lin_pix = lin_rows_bound(p.H, p.W, bm) * (p.W + 2);
  if (p.xs % 4 != 0) return UNETK_E_BADARG;
  if (p.ng > 0) {      // grouped taps (a strided conv over its space-to-depth input, conv3d.hip): K sequence = ng x Cin / 16 chunks
    if (p.ng > 12 || p.accumulate || p.kd > 1 || p.asc != nullptr) return UNETK_E_UNSUPPORTED;
    for (int g = 0; g < p.ng; ++g)
      if (p.g_ntaps[g] < 1 || p.g_ntaps[g] > 4) return UNETK_E_BADARG;
    if (p.sk_slab != nullptr && p.ys % 4 == 0) {
      const SkPlan sk = sk_plan(p.N, p.H, p.W, p.nlive ? p.nlive * CK : p.Cin, p.Cout, p.spg, p.ng);
      if (sk.on && p.sk_slab_bytes >= sk.bytes && unetk_aligned16(p.sk_slab)) {
        p.sk_tiles = sk.tiles; p.sk_nc = sk.nc; p.sk_maxp = sk.maxp; p.sk_whole = sk.whole;
        p.n_ntiles = p.Cout / sk.bn;
        if (bm == 64) return launch_lin_sk<2, 2, 1, 2, true>(p, sk, st);
        if (p.Cout % 128 == 0) return launch_lin_sk<2, 2, 2, 2, true>(p, sk, st);
        return launch_lin_sk<4, 1, 1, 2, true>(p, sk, st);
      }
    }
    p.sk_slab = nullptr;
    if (bm == 64) {
      p.n_ntiles = p.Cout / 128;
      return launch_lin<2, 2, 1, 2, false, false, false, true>(p, n_mtiles, st);
    }
    if (p.Cout % 128 == 0) {
      p.n_ntiles = p.Cout / 128;
      return launch_lin<2, 2, 2, 2, false, false, false, true>(p, n_mtiles, st);
    }
    p.n_ntiles = p.Cout / 64;
    return launch_lin<4, 1, 1, 2, false, false, false, true>(p, n_mtiles, st);
  }
  if (p.sk_slab != nullptr && !p.accumulate && p.ys % 4 == 0) {
    const SkPlan sk = sk_plan(p.N, p.H, p.W, p.nlive ? p.nlive * CK : p.Cin, p.Cout, p.spg, p.kd);
    if (sk.on && p.sk_slab_bytes >= sk.bytes && unetk_aligned16(p.sk_slab)) {
      p.sk_tiles = sk.tiles; p.sk_nc = sk.nc; p.sk_maxp = sk.maxp; p.sk_whole = sk.whole;
      p.n_ntiles = p.Cout / sk.bn;
      if (p.khalf != 0 && bm == 64) return launch_lin_sk<2, 2, 1, 2, false, true>(p, sk, st);
      if (p.khalf != 0 && p.Cout % 128 == 0) return launch_lin_sk<2, 2, 2, 2, false, true>(p, sk, st);
      if (bm == 64) return launch_lin_sk<2, 2, 1, 2>(p, sk, st);
      if (p.Cout % 128 == 0) return launch_lin_sk<2, 2, 2, 2>(p, sk, st);
      return launch_lin_sk<4, 1, 1, 2>(p, sk, st);
    }
  }
  p.sk_slab = nullptr;
  if (bm == 64) {
    p.n_ntiles = p.Cout / 128;
    if (p.khalf != 0 && !p.accumulate) return launch_lin<2, 2, 1, 2, false, false, false, false, true>(p, n_mtiles, st);
    return launch_lin<2, 2, 1, 2>(p, n_mtiles, st);
  }
  if (p.Cout % 128 == 0) {
    p.n_ntiles = p.Cout / 128;
    if (p.khalf != 0 && !p.accumulate) return launch_lin<2, 2, 2, 2, false, false, false, false, true>(p, n_mtiles, st);
    return launch_lin<2, 2, 2, 2>(p, n_mtiles, st);
  }
  p.n_ntiles = p.Cout / 64;
  return launch_lin<4, 1, 1, 2>(p, n_mtiles, st);
}
