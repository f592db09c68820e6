// conv3x3 (stride 1, SAME) forward / input gradient with bf16 STORAGE (UNETK_BF16S), round-3 kernel: PERSISTENT blocks,
// direct-to-LDS staging, v_mfma_f32_16x16x32_bf16.  slim.conv2d(x, C, 3) of NetworksV2/UNet.py:79,85,94 and its
// Conv2DBackpropInput in the mixed-precision mode of BASELINE.json configs[2] ("UNet 512x512x3 bs=64 bf16").
//
// What tools/mfma_mix_bf16.hip measured about the round-2 kernel (conv_igemm_bf16.hip; profiles/r03_mfma_mix_bf16_*):
// its loop -- compiler-scheduled LDS operand reads right behind each barrier, global -> register ring -> ds_write_b128
// staging -- reproduces at 1103 TFLOP/s, the same loop with the operand reads software-pipelined and the staging done by
// direct-to-LDS loads at 1396 (32x32x16) / 1478-1493 (16x16x32), and NO variant exceeds 1800 / 1985 TFLOP/s on random data
// even with operands in registers: under a dense bf16 MFMA stream the chip holds 1.6-2.0 GHz, not 2.4 (DVFS), and
// 16x16x32 holds a higher clock for the same FLOPs.  On top of the loop the round-2 kernel lost the head and tail of every
// tile (first halo from HBM, 128 KB of output stores) -- 5 us against a 10-40 us K loop on the 64/128-channel levels.
//
// Design:
//   * GEMM view as before: M = a 32 x 16 pixel tile, N = 128 (or 64) output channels, K = 9 taps x Cin in 32-channel chunks;
//     8 waves, wave w owns pixel rows 4w..4w+3 x all N: 4 x 8 (4 x 4) accumulator tiles of 16 x 16.
//   * PERSISTENT: one 512-thread block per CU walks its share of the tiles (neighbouring tiles per XCD); the staging stream
//     simply continues into the next tile while the current one finishes and stores, so only the very first tile of a
//     block waits for HBM, and the output stores drain under the next tile's MFMAs.
//   * staging = global_load_lds_dwordx4 only (bf16 halo and bf16 K8-packed filter panels are pure copies): no staging
//     VGPRs, no ds_write.  Halo image [34 rows][20-pixel pitch][4 x 16 B] double-buffered per chunk; the 16-byte channel
//     units of a pixel are XOR-swizzled by its column (unit ^ (col & 4 ? 0 : 3)) ON THE GLOBAL SIDE (each lane fetches the
//     unit its LDS slot must hold), which makes every ds_read_b128 A fragment conflict-free for all nine tap shifts.
//     Filter panel of one (chunk, tap) = 8 KiB, ring of six: the panel of step s + 5 is requested at step s.
//   * every wave issues the same number of loads per step (one panel piece; two halo pieces on taps 0..2), so "my piece of
//     the next step's panel has landed" is a compile-time s_waitcnt vmcnt(N_tap) in front of the ONE barrier per step.
//   * operand reads are inline-asm ds_read_b128 with counted lgkmcnt waits: B fragments of the second half-step and all
//     fragments of the NEXT step are requested while the current MFMAs run; the A registers are refilled in place, row by
//     row, as the last MFMA group using a row has been issued.  No LDS latency is exposed behind a barrier.
//   * output: C tile column j of tile tn holds channel 8 j + tn (the filter pack permutes the panel columns accordingly,
//     pack.h: conv_bf16s_chan), so a lane owns 8 consecutive channels of a pixel: ONE 16-byte store, 16 per wave and tile
//     (round 2: 64 four-byte stores).  Per-channel statistics (sum y, sum y^2 from the fp32 accumulators, or the fused
//     norm-backward partials) per lane over its 16 pixels, two shuffles, one LDS round per tile.
#include "common.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 32, TW = 16;
constexpr int PITCH = 20;                       // halo pixels per LDS row (18 used): row stride = 0 mod 4 pixels
constexpr int ROWB = PITCH * 64;                // bytes per halo row
constexpr int HROWS = TH + 2;
constexpr int HALO_UNITS = HROWS * PITCH * 4;   // 2720 16-byte units = 40 one-KiB pieces (rows 0..31) + 160 units (rows 32, 33)
constexpr int HALO_B = HALO_UNITS * 16;
constexpr int HPW = 6;                          // halo requests per wave and chunk: five whole pieces + 20 lanes of the last two rows
constexpr int HTAIL = (HALO_UNITS - 40 * 64) / 8;               // 20 units per wave
static_assert(HALO_UNITS == 40 * 64 + 8 * HTAIL && HTAIL <= 64 && HPW == 6, "halo image = 5 pieces per wave + a partial sixth");
#ifndef UNETK_V3_RING
#define UNETK_V3_RING 6
#endif
constexpr int RING = UNETK_V3_RING;          // filter-panel slots; the panel of step s + RING - 1 is requested at step s
constexpr int FLIGHT = RING - 2;             // steps between a panel's request and the barrier that needs it landed
constexpr int SLOT_B = 8192;
constexpr int OFF_RING = 2 * HALO_B;
constexpr int OFF_RED = OFF_RING + RING * SLOT_B;               // [2][8 waves][128] floats
constexpr int LDS_B = OFF_RED + 2 * 8 * 128 * 4;
static_assert(LDS_B <= 160 * 1024, "LDS budget");
static_assert(OFF_RING % 256 == 0 && SLOT_B % 256 == 0, "B addresses: the rep XOR acts on bits 4..7 only");
static_assert(RING >= 4 && RING <= 9 && 2 * HALO_B + RING * SLOT_B + 8192 <= 160 * 1024, "ring depth vs LDS");

__device__ const uint32_t kZeros[16] = {};      // source of out-of-image halo units

// The MFMAs are inline asm too, accumulating IN PLACE ("+v"): through the builtin the register allocator rotated the 128
// accumulator registers from step to step (dst != srcC) and, at 2 waves per SIMD, spilled -- and a scratch reload behind the
// barrier is an s_waitcnt vmcnt(0) on the whole staging stream.  The compiler therefore does not see matrix instructions:
// the one software-managed hazard this kernel has (VALU read of an MFMA result, in the tile epilogue) gets explicit s_nops.
// (A function, not a macro: asm operands inside a generic lambda do not capture the enclosing function's variables.)
__device__ __forceinline__ void mfma16(float __attribute__((ext_vector_type(4))) & acc, const uint32_t __attribute__((ext_vector_type(4))) & a,
                                       const uint32_t __attribute__((ext_vector_type(4))) & b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// N ds_read_b128 at byte offsets OFF, OFF + STRIDE, ... into S[0..N)
template <int OFF, int STRIDE, int N>
__device__ __forceinline__ void read_n(u32x4* S, uint32_t addr) {
  DS_READ128(S[0], addr, OFF);
  if constexpr (N > 1) read_n<OFF + STRIDE, STRIDE, N - 1>(S + 1, addr);
}
// wait until at most N LDS reads are outstanding; the "+v" operands keep the MFMAs that consume them behind the wait
template <int N>
__device__ __forceinline__ void wait_lgkm1(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm3(u32x4& a, u32x4& b, u32x4& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lgkm5(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// loads a wave issues at tap t: its piece of a filter panel, then -- at taps 0..2 -- two of its HPW pieces of the next chunk's
// halo (five or more steps of flight for data that comes from HBM, and never more than three requests in one step: a
// direct-to-LDS load stalls the issuing wave for 60-180 cycles)
constexpr int HTAPS = 3, HPT = HPW / HTAPS;
static_assert(HTAPS * HPT == HPW, "halo pieces per wave");
constexpr int tap9(int t) { return ((t % 9) + 9) % 9; }
constexpr int ops_at(int t) { return 1 + (tap9(t) < HTAPS ? HPT : 0); }
// s_waitcnt vmcnt at the top of tap t: this wave's piece of the NEXT step's panel was requested FLIGHT = RING - 2 steps ago as the
// first load of that step; younger are that step's halo pieces and the steps since.  At tap 8 the next chunk's halo
// (last requested at tap HTAPS - 1) must have landed as well: the loads of the taps since are younger.
constexpr int vm_need(int t) {
  int n = ops_at(t - FLIGHT) - 1;
  for (int k = 1; k < FLIGHT; ++k) n += ops_at(t - k);
  if (t == 8) {
    int h = 0;
    for (int k = HTAPS; k < 8; ++k) h += ops_at(k);
    n = n < h ? n : h;
  }
  return n;
}
constexpr int NSTORE = 16;        // output stores per wave and (full) tile

struct TileAt {
  int h0, w0, n0, mt, n_img;
};

// AFF (inference, unetk_conv3x3_fwd_affine): 1 = the epilogue stores the ACTIVATION relu(acc * scale[c] + shift[c]) (rounded to
// bf16 once, from the fp32 accumulator) instead of the raw output, no statistics; 2 = and max_pool2d(z, 2, 2): a lane holds
// rows wave * 4 + tm and columns 4 kq + r of 8 (4) channels, i.e. four whole 2 x 2 windows -- register maxima, 4 more stores.
template <int NT8, bool NBR, int AFF = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16s_kernel(ConvParams p) {
  static_assert(!(NBR && AFF), "one epilogue variant per instantiation");
  constexpr int NST = NSTORE + (AFF == 2 ? 4 : 0);      // output stores per wave and full tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int BN = NT8 * 16;                  // output channels per tile
  constexpr int HB = NT8 / 2;                   // B fragments per half-step
  constexpr int PLN = BN / 2;                   // lanes per wave that fetch the filter panel (4 BN units over 8 waves): 64 (32)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const char* wq = reinterpret_cast<const char*>(p.wp);

  const int late_from = (p.dbg & 1) ? 8 : 4;     // UNETK_V3_FLAGS bit 0 (measurement only): every wave requests behind the barrier
  // ---- this block's tiles: XCD x (blocks x, x + 8, ...: dealt round-robin, speed only) owns a contiguous range of the
  // logical tiles and its blocks walk it interleaved, so at any time an XCD's blocks work on neighbouring tiles (shared
  // halo columns and filter panels in its L2)
  const int T = p.ptiles;
  const int G = (int)gridDim.x;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int tq = T >> 3, tr = T & 7;
  const int xs = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int xe = xs + (xcd < tr ? tq + 1 : tq);
  const int gx = (G - xcd + 7) >> 3;            // blocks on this XCD
  int L = xs + jx;
  if (L >= xe) return;

  const int nn = p.n_ntiles, tiles_w = p.tiles_w, tiles_hw = p.tiles_w * p.tiles_h;
  auto locate = [&](int l) {
    TileAt t;
    const int ntile = l % nn;
    t.mt = l / nn;
    const int tw_i = t.mt % tiles_w;
    t.n_img = t.mt / tiles_hw;
    const int th_i = (t.mt - t.n_img * tiles_hw) / tiles_w;
    t.h0 = th_i * TH; t.w0 = tw_i * TW; t.n0 = ntile * BN;
    return t;
  };
  const int nchunks = p.Cin >> 5;
  const int cin8 = p.Cin >> 3;
  const uint32_t tap_b = (uint32_t)cin8 * p.Cout * 16u;        // bytes between the panels of two taps
  const uint32_t chunk_b = 4u * p.Cout * 16u;                  // ... of two chunks

  // ---- staging addresses
  // filter panel: wave w fetches units w PLN .. of the 4 BN; unit -> (q = 8-channel group of the chunk, n' = panel column)
  const uint32_t punit = (uint32_t)wave * PLN + (lane % PLN);
  const uint32_t plane = ((punit / BN) * p.Cout + (punit % BN)) * 16u;
  // halo: piece j = wave + 8 i, lane -> unit u = 64 j + lane of the image: pixel u >> 2 (row-major, 20 per row), slot u & 3
  // holding channel unit (u & 3) ^ swz(col)
  // Sources are computed at the request (a dozen VALU operations per piece, six pieces per chunk and wave) from packed per-lane
  // constants -- holding ready-made addresses cost 12 registers the accumulators need: field i (16 bits) = halo row | column
  // << 6 | channel unit << 11 | (row and column exist) << 13.
  uint32_t hpk[HPW / 2];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int u = i < HPW - 1 ? (wave + 8 * i) * 64 + lane : 40 * 64 + wave * HTAIL + lane;     // the sixth request: HTAIL lanes
    const int pix = u >> 2, qs = u & 3;
    const int row = pix / PITCH, col = pix - row * PITCH;
    const int qq = qs ^ (((col >> 2) & 1) ? 0 : 3);
    const uint32_t ok = row < HROWS && col < TW + 2;
    const uint32_t f = (uint32_t)(row & 63) | ((uint32_t)col << 6) | ((uint32_t)qq << 11) | (ok << 13);
    if (i & 1) hpk[i / 2] |= f << 16;
    else hpk[i / 2] = f;
  }
  // wave-uniform: the halo the NEXT requests read = 32-channel chunk hcb / 64 of the tile at (hh0 + 1, hw0 + 1) of image himg
  int hh0 = 0, hw0 = 0;
  uint32_t hcb = 0;
  bool hlive = true;
  const char* himg = reinterpret_cast<const char*>(xb);
  auto aim_halo = [&](int l) {
    hlive = l >= 0;
    if (hlive) {
      const TileAt t = locate(l);
      hh0 = t.h0 - 1; hw0 = t.w0 - 1;
      himg = reinterpret_cast<const char*>(xb + p.xa.off(t.n_img));
#ifdef UNETK_V3_PROBE
      if (p.dbg & 8) { hh0 = -1; hw0 = -1; himg = reinterpret_cast<const char*>(xb); }
#endif
    }
    hcb = 0;
  };
  auto issue_halo = [&](auto ic, int buf) {
    constexpr int I = decltype(ic)::value;
    char* dst = smem + buf * HALO_B + (I < HPW - 1 ? (wave + 8 * I) * 1024 : 40 * 1024 + wave * HTAIL * 16);
    uint32_t f = hpk[I / 2] >> (16 * (I & 1));
    asm volatile("" : "+v"(f));      // opaque: the field extractions below are loop-invariant and would be hoisted into ~18 registers
    const int gh = hh0 + (int)(f & 63u), gw = hw0 + (int)((f >> 6) & 31u);
    const bool ok = hlive && (f & 0x2000u) && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
    const uint32_t off = (uint32_t)((gh * p.W + gw) * p.xs) * 2u + ((f >> 11) & 3u) * 16u + hcb;
    const char* src = ok ? himg + off : reinterpret_cast<const char*>(kZeros);
    if (I < HPW - 1 || lane < HTAIL)       // one (exec-masked) instruction in every wave: the vmcnt bookkeeping counts it
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue_panel = [&](const char* base, uint32_t off, int slot) {
    const char* src = base + off + plane;
    char* dst = smem + OFF_RING + slot * SLOT_B + wave * (PLN * 16);
    if (PLN == 64 || lane < PLN)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- fragment addressing (LDS byte addresses in registers; tap / tile offsets are immediates)
  // A (16 pixels of one tile row x 32 channels): lane (l15, kq) reads channel unit kq of pixel column l15 + kw.  acur = the
  // current chunk's halo buffer; it flips at tap 8, when the last reads of the chunk have been requested.
  uint32_t acur[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = l15 + kw;
    acur[kw] = lds0 + wave * 4 * ROWB + col * 64 + ((kq ^ (((col >> 2) & 1) ? 0 : 3)) << 4);
  }
  // B (16 panel columns x 32 channels): lane (l15, kq) reads unit kq * BN + column; C-tile column j of tile tn is panel
  // column 16 tn + (j ^ rep[tn >> 1]), rep = {0, 1, 8, 9} ({0, 2} for 64 couts) -- the XOR keeps the round-2 kernel's reads of
  // the same pack conflict-free (pack.h).  ONE base register (rep = 0) of the slot being read, advanced (mod 6 slots) once a
  // step between the two half-steps; the other reps are base ^ (rep << 4): every other term of the address is a multiple
  // of 256 (smem is 1 KiB-aligned).
  uint32_t bcur = lds0 + OFF_RING + kq * (BN * 16) + (l15 << 4);

  f32x4 acc[4][NT8];
  u32x4 A[4], B0[HB], B1[HB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < NT8; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // B fragments of couts tiles HALF * HB .. + HB of the slot bcur points at (two neighbouring tiles share a base register)
  auto read_b = [&](auto halfc, u32x4* S) {
    constexpr int HALF = decltype(halfc)::value;
    if constexpr (HB == 4) {
      read_n<(HALF * 4 + 0) * 256, 256, 2>(S, bcur ^ (HALF ? 0x80u : 0u));
      read_n<(HALF * 4 + 2) * 256, 256, 2>(S + 2, bcur ^ (HALF ? 0x90u : 0x10u));
    } else {
      read_n<(HALF * 2 + 0) * 256, 256, 2>(S, bcur ^ (HALF ? 0x20u : 0u));
    }
  };
  auto mfma_row = [&](auto tmc, auto halfc, const u32x4* Bf) {
    constexpr int tm = decltype(tmc)::value;
    constexpr int HALF = decltype(halfc)::value;
#pragma unroll
    for (int j = 0; j < HB; ++j) mfma16(acc[tm][HALF * HB + j], A[tm], Bf[j]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- per-tile state
  TileAt cur = locate(L);
  int Ln = L + gx < xe ? L + gx : -1;
  TileAt nxt = Ln >= 0 ? locate(Ln) : cur;
  const char* pan_cur = wq + (int64_t)cur.n0 * 16;
  const char* pan_nxt = wq + (int64_t)nxt.n0 * 16;
  bool after_full = false;          // the previous tile of this block left exactly NSTORE stores behind (see vm_need use)
  int slot = 0;                     // panel slot of the current step (wave-uniform), = step index mod 6
  int hb = 0;                       // halo buffer of the current chunk

  // ---- prologue: the first tile's halo chunk 0 and the panels of steps 0..4, all landed
  aim_halo(L);
  static_for<HPW>([&](auto ic) { issue_halo(ic, 0); });
  static_for<RING - 1>([&](auto sc) {
    constexpr int S = decltype(sc)::value;
    issue_panel(pan_cur, (uint32_t)S * tap_b, S);       // chunk 0, taps 0..RING-2 (< 9; nchunks >= 1)
  });
  zero_acc();
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  // requests of the first step: B half 0, then the four A rows -- the order the steady state leaves them in
  read_b(I0{}, B0);
  read_n<0, ROWB, 4>(A, acur[0]);

  for (;;) {
    for (int c = 0; c < nchunks; ++c) {
      const bool last_chunk = c == nchunks - 1;
      // the halo requested during this chunk is chunk c + 1 of this tile, or chunk 0 of the next tile
      if (last_chunk) aim_halo(Ln);
      else hcb = (uint32_t)(c + 1) * 64u;
      static_for<9>([&](auto tc) {
        constexpr int T9 = decltype(tc)::value;
        constexpr int NT9 = (T9 + 1) % 9, NKH = NT9 / 3, NKW = NT9 % 3;
        // ---- top: my piece of the next step's panel (and, at tap 8, of the next chunk's halo) has landed; everybody's
#ifdef UNETK_V3_PROBE
        if ((p.dbg & 64) && c == 0 && after_full) wait_vm<vm_need(T9) + NST>();      // never wait for the stores in the first chunk
        else
#endif
        if (T9 < FLIGHT && c == 0 && after_full) wait_vm<vm_need(T9) + NST>();   // the stores of the last tile are younger
        else wait_vm<vm_need(T9)>();
#ifdef UNETK_V3_PROBE
        if (!(p.dbg & 16))
#endif
        __builtin_amdgcn_s_barrier();
        // The requests of step + 5 / of the next chunk's halo: waves 0..3 issue them here, waves 4..7 between the two half-steps.
        // Wave w and wave w + 4 share a SIMD (waves are dealt to the SIMDs cyclically), and a direct-to-LDS load holds its wave
        // at issue: this way one partner requests while the other feeds the matrix pipe.  (Both requesting behind the barrier left
        // the pipe idle for that time in every step: 56 % busy where the loop without staging measures 90 %.)
        const int fslot = slot == 0 ? RING - 1 : slot - 1;
        const int hb_req = hb ^ 1;
        auto requests = [&]() {
#ifdef UNETK_V3_PROBE
          if (p.dbg & 4) return;
#endif
          // panel of step + RING - 1 into the slot of step - 1 (free behind the barrier): tap (T9 + RING - 1) % 9 of chunk
          // c + (T9 + RING - 1) / 9 -- of the next tile behind the last chunk
          constexpr int PT = (T9 + RING - 1) % 9, DC = (T9 + RING - 1) / 9;
          const int pc = c + DC;
          const bool over = pc >= nchunks;
          const char* base = over ? pan_nxt : pan_cur;
          const uint32_t off = (uint32_t)PT * tap_b + (over ? 0u : (uint32_t)pc * chunk_b);
#ifdef UNETK_V3_PROBE
          if (!(p.dbg & 512))
#endif
          issue_panel(base, off, fslot);
#ifdef UNETK_V3_PROBE
          if (!(p.dbg & 256))
#endif
          if constexpr (T9 < HTAPS) {
            issue_halo(std::integral_constant<int, T9 * HPT>{}, hb_req);
            issue_halo(std::integral_constant<int, T9 * HPT + 1>{}, hb_req);
          }
        };
        if (wave < late_from) requests();
        // ---- first half: couts tiles 0..HB-1.  Outstanding reads, oldest first: B0 x HB, A0..A3 (requested during the
        // previous step), now B1 x HB
        read_b(I1{}, B1);
        if constexpr (HB == 4) wait_lgkm5<3 + HB>(B0[0], B0[1], B0[2], B0[3], A[0]);
        else wait_lgkm3<3 + HB>(B0[0], B0[1], A[0]);
        mfma_row(I0{}, I0{}, B0);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm1<2 + HB>(A[1]);
        mfma_row(I1{}, I0{}, B0);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm1<1 + HB>(A[2]);
        mfma_row(I2{}, I0{}, B0);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm1<HB>(A[3]);
        mfma_row(I3{}, I0{}, B0);
        __builtin_amdgcn_sched_barrier(0);
        if (wave >= late_from) requests();
        // ---- second half: couts tiles HB..; the NEXT step's fragments are requested as registers fall free (behind the
        // last step of a tile they belong to the next tile's first step and wait in registers through the epilogue)
        {
          const int ns = slot == RING - 1 ? 0 : slot + 1;
          const uint32_t adv = (uint32_t)(ns - slot) * SLOT_B;      // + 8 KiB, or back to slot 0
          bcur += adv;
          slot = ns;
          if constexpr (T9 == 8) {
            const uint32_t flip = hb ? (uint32_t)-HALO_B : (uint32_t)HALO_B;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acur[kw] += flip;
            hb ^= 1;
          }
        }
        read_b(I0{}, B0);
        if constexpr (HB == 4) wait_lgkm5<HB>(B1[0], B1[1], B1[2], B1[3], A[0]);
        else wait_lgkm3<HB>(B1[0], B1[1], A[0]);
        mfma_row(I0{}, I1{}, B1);
        DS_READ128(A[0], acur[NKW], (0 + NKH) * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(I1{}, I1{}, B1);
        DS_READ128(A[1], acur[NKW], (1 + NKH) * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(I2{}, I1{}, B1);
        DS_READ128(A[2], acur[NKW], (2 + NKH) * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(I3{}, I1{}, B1);
        DS_READ128(A[3], acur[NKW], (3 + NKH) * ROWB);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- tile epilogue: bf16 output (one 16-byte store per pixel and lane), statistics
#ifdef UNETK_V3_PROBE
    if (p.dbg & 32) { zero_acc(); } else
#endif
    {
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA results -> VALU reads (see MFMA16)
      float ssum[NT8], ssq[NT8];
#pragma unroll
      for (int j = 0; j < NT8; ++j) ssum[j] = ssq[j] = 0.f;
      int ln = lane;                       // opaque: keeps the epilogue's per-lane address terms out of the main loop's registers
      asm volatile("" : "+v"(ln));
      const int l15 = ln & 15, kq = ln >> 4;
      const int64_t yimg = p.ya.off(cur.n_img);
      bf16_t* yb = reinterpret_cast<bf16_t*>(p.y) + yimg + cur.n0 + l15 * NT8;
      float nsc[NBR ? NT8 : 1], nsh[NBR ? NT8 : 1], nmu[NBR ? NT8 : 1], nrs[NBR ? NT8 : 1];
      const bf16_t* nyb = nullptr;
      if constexpr (NBR) {
        const int64_t o = (int64_t)cur.n_img * p.nsst + cur.n0 + l15 * NT8;
#pragma unroll
        for (int j = 0; j < NT8; ++j) { nsc[j] = p.nsc[o + j]; nsh[j] = p.nsh[o + j]; nmu[j] = p.nmu[o + j]; nrs[j] = p.nrs[o + j]; }
        nyb = static_cast<const bf16_t*>(p.ny) + (int64_t)cur.n_img * p.H * p.W * p.nys + cur.n0 + l15 * NT8;
      }
      if constexpr (AFF != 0) {
        // the normaliser's affine + ReLU on the fp32 accumulators, in place (they are zeroed behind the epilogue anyway)
        float asc[NT8], ash[NT8];
#pragma unroll
        for (int j = 0; j < NT8; ++j) { asc[j] = p.asc[cur.n0 + l15 * NT8 + j]; ash[j] = p.ash[cur.n0 + l15 * NT8 + j]; }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int j = 0; j < NT8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tm][j][r] = fmaxf(fmaf(acc[tm][j][r], asc[j], ash[j]), 0.f);
        if constexpr (AFF == 2) {
          const int Hp = p.H >> 1, Wp = p.W >> 1;
          bf16_t* pb = reinterpret_cast<bf16_t*>(p.pool) + (int64_t)cur.n_img * Hp * Wp * p.pool_s + cur.n0 + l15 * NT8;
#pragma unroll
          for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
              const int gh = cur.h0 + wave * 4 + 2 * tp, gw = cur.w0 + 4 * kq + 2 * rp;
              if (gh < p.H && gw < p.W) {
                uint32_t pk[NT8 / 2];
#pragma unroll
                for (int j = 0; j < NT8 / 2; ++j) {
                  // the maximum of the four ROUNDED activations = the rounded maximum (rounding is monotone)
                  const float m0 = fmaxf(fmaxf(acc[2 * tp][2 * j][2 * rp], acc[2 * tp][2 * j][2 * rp + 1]),
                                         fmaxf(acc[2 * tp + 1][2 * j][2 * rp], acc[2 * tp + 1][2 * j][2 * rp + 1]));
                  const float m1 = fmaxf(fmaxf(acc[2 * tp][2 * j + 1][2 * rp], acc[2 * tp][2 * j + 1][2 * rp + 1]),
                                         fmaxf(acc[2 * tp + 1][2 * j + 1][2 * rp], acc[2 * tp + 1][2 * j + 1][2 * rp + 1]));
                  pk[j] = unetk_pk_bf16(m0, m1);
                }
                bf16_t* pp = pb + ((int64_t)(gh >> 1) * Wp + (gw >> 1)) * p.pool_s;
                if constexpr (NT8 == 8) *reinterpret_cast<u32x4*>(pp) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                else *reinterpret_cast<uint2*>(pp) = make_uint2(pk[0], pk[1]);
              }
            }
        }
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const int gh = cur.h0 + wave * 4 + tm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gw = cur.w0 + 4 * kq + r;
          if (gh < p.H && gw < p.W) {
            const int64_t po = (int64_t)gh * p.W + gw;
            uint32_t pk[NT8 / 2];
#pragma unroll
            for (int j = 0; j < NT8 / 2; ++j) pk[j] = unetk_pk_bf16(acc[tm][2 * j][r], acc[tm][2 * j + 1][r]);
#ifdef UNETK_V3_PROBE
            if (!(p.dbg & 128)) {
#endif
            if constexpr (NT8 == 8) *reinterpret_cast<u32x4*>(yb + po * p.ys) = u32x4{pk[0], pk[1], pk[2], pk[3]};
            else *reinterpret_cast<uint2*>(yb + po * p.ys) = make_uint2(pk[0], pk[1]);
#ifdef UNETK_V3_PROBE
            }
#endif
            if constexpr (NBR) {      // on the values memory holds (what a separate reduction pass would read back)
              uint32_t qv[NT8 / 2];
              if constexpr (NT8 == 8) {
                const u32x4 q4 = *reinterpret_cast<const u32x4*>(nyb + po * p.nys);
                qv[0] = q4.x; qv[1] = q4.y; qv[2] = q4.z; qv[3] = q4.w;
              } else {
                const uint2 q2 = *reinterpret_cast<const uint2*>(nyb + po * p.nys);
                qv[0] = q2.x; qv[1] = q2.y;
              }
#pragma unroll
              for (int j = 0; j < NT8 / 2; ++j) {
                const float d0 = unetk_bf16_lo(pk[j]), d1 = unetk_bf16_hi(pk[j]);
                const float q0 = unetk_bf16_lo(qv[j]), q1 = unetk_bf16_hi(qv[j]);
                const float du0 = fmaf(q0, nsc[2 * j], nsh[2 * j]) > 0.f ? d0 : 0.f;
                const float du1 = fmaf(q1, nsc[2 * j + 1], nsh[2 * j + 1]) > 0.f ? d1 : 0.f;
                ssum[2 * j] += du0; ssq[2 * j] += du0 * ((q0 - nmu[2 * j]) * nrs[2 * j]);
                ssum[2 * j + 1] += du1; ssq[2 * j + 1] += du1 * ((q1 - nmu[2 * j + 1]) * nrs[2 * j + 1]);
              }
            } else if constexpr (AFF == 0) {
#pragma unroll
              for (int j = 0; j < NT8; ++j) {
                const float v = acc[tm][j][r];
                ssum[j] += v; ssq[j] += v * v;
              }
            }
          }
        }
      }
      after_full = !NBR && cur.h0 + TH <= p.H && cur.w0 + TW <= p.W;      // every store instruction was issued
      zero_acc();
      if (AFF == 0 && p.stat != nullptr) {
        float* red = reinterpret_cast<float*>(smem + OFF_RED);      // [2][8][BN]
#pragma unroll
        for (int j = 0; j < NT8; ++j) {
          ssum[j] += __shfl_xor(ssum[j], 16); ssum[j] += __shfl_xor(ssum[j], 32);
          ssq[j] += __shfl_xor(ssq[j], 16); ssq[j] += __shfl_xor(ssq[j], 32);
        }
        if (kq == 0) {
#pragma unroll
          for (int j = 0; j < NT8; ++j) {
            red[(0 * 8 + wave) * BN + l15 * NT8 + j] = ssum[j];
            red[(1 * 8 + wave) * BN + l15 * NT8 + j] = ssq[j];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int te = wave * 64 + ln;
        if (te < 2 * BN) {
          const int k = te / BN, n = te - k * BN;
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) s += red[(k * 8 + w) * BN + n];
          p.stat[((int64_t)k * p.stat_rows + cur.mt) * p.Cout + cur.n0 + n] = s;
        }
        // red is written again one whole tile (>= 9 barriers) later
      }
    }
    if (Ln < 0) break;
    L = Ln; cur = nxt; pan_cur = pan_nxt;
    Ln = L + gx < xe ? L + gx : -1;
    nxt = Ln >= 0 ? locate(Ln) : cur;
    pan_nxt = wq + (int64_t)nxt.n0 * 16;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragment requests behind the last step
  wait_vm<0>();       // the look-ahead requests past the last tile must land before the block's LDS is released
}

}  // namespace

// ---- host side (called from conv_igemm_bf16.hip's dispatch)
// UNETK_V3_FLAGS (measurement only; read once): bit 0 = no staggered requests, bit 1 = do not use this kernel at all.
// A build with -DUNETK_V3_PROBE (tools/probe_v3.sh; never the shipped library) adds switches that give WRONG RESULTS and
// exist to time the parts of the loop: 4 = no staging requests in the main loop, 8 = every halo read hits tile 0 of image 0
// (cache-resident), 16 = no barriers in the main loop, 32 = no tile epilogue, 64 = the waits of a tile's first chunk never
// cover the previous tile's stores, 128 = no output stores, 256 = no halo requests, 512 = no filter-panel requests.
static int v3_flags() {
  static int f = -1;
  if (f < 0) {
    const char* e = getenv("UNETK_V3_FLAGS");
    f = e ? atoi(e) : 0;
  }
  return f;
}

bool unetk_conv_bf16s_v3_ok(int N, int H, int W, int Cin, int Cout, int xs, int ys) {
  if (v3_flags() & 2) return false;
  if (Cin % 32 != 0 || (Cout % 128 != 0 && Cout != 64)) return false;
  if (xs % 8 != 0 || ys % 8 != 0) return false;              // 16-byte halo units / output stores
  if (H < 24 || (int64_t)H * W * xs * 2 >= (int64_t(1) << 31)) return false;      // 32-bit byte offsets inside an image
  const int bn = Cout % 128 == 0 ? 128 : 64;
  const int64_t tiles = (int64_t)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * (Cout / bn);
  return tiles >= 200 && tiles < (1 << 30);
}

int unetk_conv_bf16s_v3_stat_rows(int N, int H, int W) { return N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW); }

template <int NT8, bool NBR, int AFF = 0>
static int launch_v3(const ConvParams& p, hipStream_t st) {
  auto kern = conv3x3_bf16s_kernel<NT8, NBR, AFF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return UNETK_E_BADARG;
    n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount : 8;
  }
  const int grid = p.ptiles < n_cu ? p.ptiles : n_cu;       // one resident block per CU (146 KB of LDS), >= 8 blocks
  UNETK_LAUNCH(kern, dim3(grid), dim3(512), LDS_B, st, p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

int unetk_conv_bf16s_v3_run(ConvParams p, hipStream_t st) {
  if (!unetk_conv_bf16s_v3_ok(p.N, p.H, p.W, p.Cin, p.Cout, p.xs, p.ys)) return UNETK_E_UNSUPPORTED;
  if (p.accumulate) return UNETK_E_UNSUPPORTED;
  const int bn = p.Cout % 128 == 0 ? 128 : 64;
  p.tiles_h = (p.H + TH - 1) / TH;
  p.tiles_w = (p.W + TW - 1) / TW;
  p.n_ntiles = p.Cout / bn;
  p.stat_rows = p.N * p.tiles_h * p.tiles_w;
  p.ptiles = p.stat_rows * p.n_ntiles;
  p.dbg = v3_flags();
  if (p.ny != nullptr) {
    if (bn != 128 || p.nys % 8 != 0 || p.asc != nullptr) return UNETK_E_UNSUPPORTED;
    return launch_v3<8, true>(p, st);
  }
  if (p.asc != nullptr) {         // inference epilogue: (scale, shift) + ReLU [+ 2 x 2 max-pool]
    if (p.stat != nullptr) return UNETK_E_UNSUPPORTED;
    if (p.pool != nullptr) {
      if (((p.H | p.W) & 1) || p.pool_s % (bn == 128 ? 8 : 4) != 0) return UNETK_E_UNSUPPORTED;
      return bn == 128 ? launch_v3<8, false, 2>(p, st) : launch_v3<4, false, 2>(p, st);
    }
    return bn == 128 ? launch_v3<8, false, 1>(p, st) : launch_v3<4, false, 1>(p, st);
  }
  return bn == 128 ? launch_v3<8, false>(p, st) : launch_v3<4, false>(p, st);
}
