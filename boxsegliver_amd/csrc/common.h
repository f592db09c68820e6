// Shared device/host helpers for libunetk (gfx950 only).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "unetk.h"

// Every kernel launch of the library.  With the kernel trace on (prof.hip, unetk_prof_enable: bench.py's roofline block) the
// dispatch carries a start / stop event pair bound to the kernel itself, so its duration is the GPU's own begin -> end
// interval, independent of what the host does around the launch; off (always, outside bench.py) it is hipLaunchKernelGGL.
extern int g_unetk_prof_on;
void unetk_prof_pair(const void* fn, hipEvent_t* a, hipEvent_t* b);
#define UNETK_LAUNCH(kern, grid, block, lds, st, ...)                                                   \
  do {                                                                                                  \
    if (__builtin_expect(g_unetk_prof_on, 0)) {                                                         \
      hipEvent_t pa_, pb_;                                                                              \
      unetk_prof_pair(reinterpret_cast<const void*>(kern), &pa_, &pb_);                                 \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, pa_, pb_, 0, __VA_ARGS__);                      \
    } else {                                                                                            \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                      \
    }                                                                                                   \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define UNETK_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return (int)e_;         \
  } while (0)

#define UNETK_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return UNETK_E_BADARG; \
  } while (0)

static inline bool unetk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2): ids b and b+8 share
// one.  Remap so that each XCD walks a CONTIGUOUS range of logical tiles (neighbouring tiles
// share input halos / weight panels -> L2 hits).  Bijective for any grid size.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// C/D fragment row of v_mfma_f32_32x32x2_f32: reg r (0..15), lane half h -> row index.
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// floor(n / d) for 0 <= n < 2^22 and d >= 1 from a host-computed reciprocal: a float estimate (within 1 of the quotient) and one
// correction -- 6 VALU instructions where the integer division by a run-time divisor expands to ~25.  (Round 5: the stacked-plane
// filter gradient spent 5.1 VALU instructions per MFMA, the plain one 2.0: four such divisions per staged piece.)
__device__ __forceinline__ int unetk_fdiv(int n, int d, float rcp) {
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- bf16 storage (UNETK_BF16S): activations / activation gradients live in HBM as bf16 (uint16_t bit patterns),
// arithmetic stays fp32.  ld4 / st4 move FOUR consecutive channels: a float4 (16 B) or four bf16 (8 B).
typedef uint16_t bf16_t;
__device__ __forceinline__ uint32_t unetk_pk_bf16(float lo, float hi) {   // RNE, lo in bits 0..15
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float unetk_bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float unetk_bf16_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(unetk_bf16_lo(v.x), unetk_bf16_hi(v.x), unetk_bf16_lo(v.y), unetk_bf16_hi(v.y));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(unetk_pk_bf16(v.x, v.y), unetk_pk_bf16(v.z, v.w));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
// values a bf16 store would keep (for statistics / column sums that must see what memory holds)
__device__ __forceinline__ float unetk_round_bf16(float v) { return unetk_bf16_lo(unetk_pk_bf16(v, 0.f)); }
static inline bool unetk_aligned8(const void* p) { return (((uintptr_t)p) & 7u) == 0; }

// counter RNG of slim.dropout masks: murmur3 finaliser of (seed, element index) -> uniform [0, 1); the forward and the
// backward of an op regenerate the same mask from it (nothing is stored for the big activations)
__device__ __forceinline__ float unetk_uniform(uint32_t seed, uint32_t idx) {
  uint32_t h = idx * 0x9E3779B1u + seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- internal cross-file helpers (not part of the C ABI)
// Operand of the 2-D conv kernels.  "Images" are (n, d) planes of an N(D)HWC tensor: image i lives at
//   (i / group) * group_stride + (i % group) * img_stride   floats,
// so a depth-sliced / depth-strided view of a 5-D tensor (the taps of a 3-D conv) is addressed without copies.
struct ImgAddr {
  int64_t img_stride, group_stride;
  int group;
  __host__ __device__ int64_t off(int i) const { return (int64_t)(i / group) * group_stride + (int64_t)(i % group) * img_stride; }
};
static inline ImgAddr unetk_dense_addr(int H, int W, int pix_stride) {
  ImgAddr a;
  a.img_stride = (int64_t)H * W * pix_stride;
  a.group_stride = 0;
  a.group = 1 << 30;
  return a;
}

struct ConvParams {
  const float* x;
  const float* wp;
  float* y;
  float* stat;
  int N, H, W, Cin, Cout, xs, ys;   // N = number of images (planes)
  int tiles_h, tiles_w, n_ntiles, stat_rows;
  int ptiles;                        // conv_igemm_bf16s.hip (persistent blocks): pixel tiles x output-channel tiles
  int dbg;                           // conv_igemm_bf16s.hip: UNETK_V3_FLAGS (measurement switches)
  ImgAddr xa, ya;
  int accumulate;                    // epilogue: y += acc (depth taps of a 3-D conv), statistics on the sum
  int bf16;                          // UNETK_BF16: operands rounded to bf16 for v_mfma_f32_32x32x16_bf16 (wp = bf16 K8 pack);
                                     // UNETK_BF16S: x and y ARE bf16 in memory (wp = the channel-pair-permuted pack)
  int ybf16;                         // direct kernel only: y is stored as bf16 (UNETK_BF16S first layer, fp32 input)
  int spg;                           // planes per statistics group (a 3-D sample's depth planes); 0/1 = every plane
  int lin_pix;                       // conv_igemm_lin.hip: padded pixels a block may stage (sizes its LDS)
  int stride;                        // 2: TF SAME stride-2 conv; H x W = OUTPUT extent and the four fields below are set
  int Hin, Win, pbh, pbw;            // input extent and SAME pad-before (0 on even, 1 on odd input extents)
  // conv_igemm_lin.hip GEN variant (stride-2 input gradient): the four output-parity classes in one launch,
  // class q = blockIdx-derived; tap_off = 4 * (row offset 0..2) + (column offset 0..2) in the padded halo
  int ntaps[4], tap_off[4][4], tap_panel[4][4];
  int ooh[4], oow[4];                    // class q: output pixel (a, b) -> (os a + ooh[q], os b + oow[q])
  int os, Hd, Wd;                        // of an Hd x Wd plane
  int dil;                               // 2 = rate-2 atrous conv (taps at (2 kh, 2 kw), SAME pad 2); 0 / 1 = dense
  // fused depth taps (linear-pixel kernel, depth stride 1): kd > 1 contracts kd depth taps in ONE launch -- tap dt reads
  // the plane (depth index + dshift0 + dt * dstep) of the same sample (zero outside [0, spg)) with the filter panel
  // wp + dt * 9 * Cin * Cout; no per-tap read-modify-write of the output.  0 / 1 = a plain 2-D conv per plane.
  int kd, dshift0, dstep;
  // fused depth taps in the TILED kernel (conv_igemm.hip; the natively strided (3,3,3) layers): image i is output plane
  // dep = i % spg of its sample, tap dt reads input depth dep * dsd + dshift0 + dt (skipped outside [0, din)), i.e. the plane
  // at xa.off(i) + (dshift0 + dt) * dplane floats; filter panel wp + dt * 9 * Cin * Cout
  int dsd, din;
  int64_t dplane;
  // input gradient fused with the PRODUCER's norm-backward reduction (tiled fp32 / bf16-storage kernels): the output dx
  // is the dz of the unit whose raw conv output is ny (same pixels, p.Cout channels, pixel stride nys); the epilogue
  // emits, instead of (sum y, sum y^2), the partials  sum du  and  sum du * xhat  with du = dz * (ny*nsc + nsh > 0),
  // xhat = (ny - nmu) * nrs, into p.stat.  nsst = statistics stride per image (C under instance norm, 0 under batch norm).
  const void* ny;
  const float *nsc, *nsh, *nmu, *nrs;
  int nys, nsst;
  // inference epilogue (unetk_conv3x3_fwd_affine): the normaliser's per-channel (scale, shift) are known before the conv runs
  // (moving statistics / --without_norm), so the epilogue writes z = relu(acc * asc[c] + ash[c]) instead of the raw output,
  // and -- pool != nullptr -- also max_pool2d(z, 2, 2) into pool[N, H/2, W/2, .] (pixel stride pool_s; H and W even)
  const float *asc, *ash;
  void* pool;
  int pool_s;
  // stream-K scheduling of the linear-pixel kernel (conv_igemm_lin.hip): slab for the pieces of split tiles
  // ([sk_tiles][sk_maxp][BM][BN] floats, caller's workspace; nullptr = one block per tile), chunks per tile, tiles
  float* sk_slab;
  size_t sk_slab_bytes;
  int sk_tiles, sk_nc, sk_maxp, sk_whole;
  // grouped taps (conv_igemm_lin.hip GRP, plain variant; round 5): a STRIDED conv on small output planes as a stride-1
  // contraction over a space-to-depth copy of its input (conv3d.hip s2d_kernel: x[2a + p] -> xs[a][class p], the parity
  // classes side by side on the channel axis).  Output o reads input 2 o + k (SAME padding of an even extent: 0 before, 1
  // after): class k & 1 at offset k >> 1.  The K sequence is ng groups x Cin / 16 chunks; group g = one (depth tap, in-plane
  // class): its 16-channel chunks start at channel g_chan[g] of an xs pixel, are read from the plane shifted by g_dz[g], and
  // are contracted with g_ntaps[g] <= 4 taps -- halo offset g_off (4 x row + column, 0..2 each, in the padded plane) and
  // filter panel g_panel (0..26 in wp).  Exactly the layer's kd x 9 tap-products; every MFMA row is an output pixel.
  int ng;
  int g_chan[12], g_dz[12], g_ntaps[12], g_off[12][4], g_panel[12][4];
  // four-class input gradient with depth STRIDE 2 (the (2,2,2) layer; conv_igemm_lin.hip FUSED): both depth parities of dx in
  // one launch -- the second half of the pixel tiles writes the planes at p.y + dpar_yoff
  int dpar;
  int64_t dpar_yoff;
  // live K chunks (conv_igemm_lin.hip; round 5): the caller promises that only the listed 16-channel chunks of the contraction
  // axis carry non-zero filter rows (a channel-padded net: 240 real channels in 256, a concat of two such halves in 512), so
  // the K sequence visits klive[0 .. nlive) only; bit i of khalf = of live chunk i only channels 0..7 are real (120 in 128):
  // the plain variant runs half its MFMAs there.  nlive = 0: every chunk.
  int nlive;
  unsigned khalf;
  int klive[32];
};
// live8: bit i = channels [8 i, 8 i + 8) of a K-channel contraction axis hold a real channel (0 = no information: all live)
inline int unetk_live_chunks(const uint32_t live8[2], int K) {
  const uint64_t m = (uint64_t)live8[0] | ((uint64_t)live8[1] << 32);
  if (m == 0 || K % 16 != 0 || K > 512) return K / 16;
  int n = 0;
  for (int c = 0; c < K / 16; ++c) n += ((m >> (2 * c)) & 3) != 0;
  return n > 0 ? n : K / 16;
}
inline void unetk_set_klive(ConvParams& p, const uint32_t live8[2], int K, bool halves) {
  const uint64_t m = (uint64_t)live8[0] | ((uint64_t)live8[1] << 32);
  p.nlive = 0; p.khalf = 0;
  if (m == 0 || K % 16 != 0 || K > 512) return;
  int n = 0;
  unsigned half = 0;
  for (int c = 0; c < K / 16; ++c) {
    const int b = (int)((m >> (2 * c)) & 3);
    if (b == 0) continue;
    if (b == 1 && halves) half |= 1u << n;
    p.klive[n++] = c;
  }
  if (n == 0 || (n == K / 16 && half == 0)) return;
  p.nlive = n; p.khalf = half;
}
bool unetk_conv_lin_gen_ok(int H, int W, int Cin, int Cout);
int unetk_conv_run_lin_gen(ConvParams p, hipStream_t st);
int unetk_conv_run(ConvParams p, hipStream_t st);          // conv_igemm.hip: picks the tile configuration
int unetk_conv_stat_rows(int N, int H, int W, int Cin, int Cout, int spg = 1, int stride = 1, int dil = 1);
bool unetk_conv_stride2_ok(int Cin, int Cout);
// conv_igemm_lin.hip: linear-pixel variant for planes narrower than 32 pixels (same packed filters)
bool unetk_conv_lin_ok(int N, int H, int W, int Cin, int Cout, int spg);
int unetk_conv_stat_rows_lin(int N, int H, int W, int spg, int Cout = 0);   // Cout picks the block height (64 / 128 pixels)
int unetk_conv_run_lin(ConvParams p, hipStream_t st);
size_t unetk_conv_lin_sk_bytes(int N, int H, int W, int Cin, int Cout, int spg, int kd);   // 0 = stream-K not used for this shape
// conv_igemm_bf16.hip
bool unetk_conv_bf16_ok(int Cin, int Cout);
int unetk_conv_run_bf16(ConvParams p, hipStream_t st);
int unetk_conv_stat_rows_bf16(int N, int H, int W, int Cin, int Cout);
int unetk_conv_stat_rows_bf16s(int N, int H, int W, int Cin, int Cout, int xs, int ys);   // UNETK_BF16S: round-3 kernel where it applies
// conv_igemm_bf16s.hip: UNETK_BF16S round-3 kernel (persistent 512 x 128 / 512 x 64 tiles, LDS-DMA staging, 16x16x32 MFMA)
bool unetk_conv_bf16s_v3_ok(int N, int H, int W, int Cin, int Cout, int xs, int ys);
int unetk_conv_bf16s_v3_stat_rows(int N, int H, int W);
int unetk_conv_bf16s_v3_run(ConvParams p, hipStream_t st);

struct WgParams {
  const float* x;
  const float* dy;
  const float* zeros;  // >= 256 B of zeros in global memory: source of out-of-image halo pixels
  float* slab;
  int N, H, W, Cin, Cout, xs, ys;
  int tiles_h, tiles_w, total_tiles, tiles_per_split, n_ci_tiles, n_co_tiles;
  ImgAddr xa, ya;
  int bf16;            // x and dy rounded to bf16 on their way out of LDS, v_mfma_f32_32x32x16_bf16
  // strided convs (slim.conv3d stride (.,2,2)): H, W are the OUTPUT plane (dy), the input plane is Hin x Win and output
  // pixel (oh, ow) reads input rows stride*oh - pbh + kh.  stride 0 / 1 = plain conv (Hin = H, Win = W, pb = 1).
  int stride, Hin, Win, pbh, pbw;
  int dil;             // 2 = rate-2 atrous conv (slim.conv2d(..., rate=2), SmallUNet's bridge): taps at (2 kh, 2 kw), pad 2
  // fused depth taps (kd > 1; the filter gradient of a (3,3,3) conv in ONE launch): block (.., dt, split) contracts dy plane i
  // (output plane dep = i % spg of its sample) with the x plane at xa.off(i) + (dshift0 + dt) * dplane floats, i.e. input
  // depth dep * dsd + dshift0 + dt (zeros outside [0, din)); its panel goes to slab [split][dt][9 Cin Cout], so one slab
  // reduction over kd * 9 * Cin * Cout elements yields dw[kd][3][3][Cin][Cout].  With kd x the blocks per split, a third of
  // the splits fills the chip: three times the tiles per block, a third of the slab traffic.
  int kd, dshift0, dsd, din, spg;
  int64_t dplane;
  int dbg;             // conv_wgrad_bf16s.hip probe build (-DUNETK_V3_PROBE): UNETK_V3_FLAGS
  // reciprocals for the stacked-plane variant's per-lane index arithmetic (set by unetk_wgrad_run): 1 / (H + 1), 1 / xa.group,
  // 1 / ya.group, 1 / spg -- unetk_fdiv below
  float rcp_h1, rcp_xg, rcp_yg, rcp_spg;
};
// conv_wgrad.hip: dw[9][Cin][Cout] = filter gradient of one 2-D tap plane; ws layout as unetk_conv3x3_wgrad
size_t unetk_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int kd = 1);
int unetk_wgrad_run(WgParams p, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// conv_wgrad_bf16s.hip: UNETK_BF16S -- x and dy are bf16 in memory (x fp32 for the first layer, 9 * Cin <= 32)
size_t unetk_wgrad_bf16s_ws_bytes(int N, int H, int W, int Cin, int Cout);
int unetk_wgrad_bf16s_run(WgParams p, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// stride-2 variant (p.stride == 2; p.H, p.W = output plane): fp32, Cin % 32 == 0 and Cout % 64 == 0
bool unetk_wgrad_strided_ok(int Cin, int Cout);
size_t unetk_wgrad_strided_ws_bytes(int N, int Ho, int Wo, int Cin, int Cout, int kd = 1);
// dst[i] = sum_s slab[s*n + i] in fixed order (n % 4 == 0).
int unetk_launch_slab_reduce(const float* slab, int S, int64_t n, float* dst, hipStream_t st);
// dst[k][c] = sum_rows src[k][row][c] (fp64 accumulate).  tmp: K*64*C floats when rows > 256.
int unetk_rows_reduce(const float* src, int K, int rows, int C, float* dst, float* tmp, hipStream_t st);
int unetk_rows_reduce_alias(const float* src, int K, int rows, int C, float* dst, float* tmp, float* alias0, float* alias1,
                            hipStream_t st);
// row counts up to UNETK_RR_DIRECT_ROWS go straight to the final kernel; more rows take a first level of 64 row blocks
// (1024 was measured: the final kernel has only C / 16 x K blocks and took 15 us on 1024 rows, more than the two levels)
constexpr int UNETK_RR_DIRECT_ROWS = 256;
constexpr int UNETK_RR_WIDE_ROWS = 1024;    // float4-capable inputs: the wide final kernel takes this many rows directly
size_t unetk_rows_reduce_tmp_floats(int K, int rows, int C);
int unetk_rows_reduce_l1(const float* src, int K, int rows, int C, float* tmp, hipStream_t st);   // -> tmp[K][64][C]

// Thread mapping for [npix, C] column-wise kernels: thread = (channel quad, row lane).
struct ColMap {
  int cq_n, rows_per_iter;
};
static inline ColMap unetk_colmap(int C) {
  ColMap m;
  m.cq_n = C / 4;
  m.rows_per_iter = 256 / m.cq_n;
  return m;
}
constexpr int UNETK_COL_BLOCKS = 1024;
