// Filter re-layouts ("packs") of the matrix kernels as per-item device functions, shared by the single-filter pack
// kernels (conv_igemm.hip, conv_igemm_bf16.hip, deconv.hip) and the batched unetk_pack_many (pack.hip).
// Item i of a filter = one 16-byte unit of the packed result.
#pragma once
#include "common.h"

namespace unetk_pack {

__device__ __forceinline__ uint32_t pk2(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// UNETK_BF16S transposed-conv panels: column position n' of every 64-column block holds output channel 2 (n' & 31) + (n' >> 5 & 1)
__device__ __forceinline__ int perm64(int n) { return (n & ~63) + 2 * (n & 31) + ((n >> 5) & 1); }

// UNETK_BF16S conv3x3 panels: column position n' of the filter panel -> output channel.  Blocks of 128 columns when the
// output dimension is a multiple of 128, else blocks of 64.  Position n' = 16 m + r of a block holds channel
//   (r ^ rep[m >> 1]) * NT + m      (NT = 8 tiles per 128-block, rep = {0, 1, 8, 9};  NT = 4 per 64-block, rep = {0, 2})
// so that (a) the 16x16x32 kernel (conv_igemm_bf16s.hip), whose accumulator-tile column j of tile m reads position
// 16 m + (j ^ rep), owns channels NT j .. NT j + NT - 1 per lane (one 16- / 8-byte store), and (b) the 32x32x16 kernels
// (conv_igemm_bf16.hip), whose lane l of tile tn wants channel 2 l + tn (one 4-byte store), find those channels at positions
// that differ in the low four bits across every 16-lane ds_read_b128 group: {0,3,5,6} and {0,1,6,7,10,..13} -- the values of
// l >> 2 resp. l >> 1 inside a group -- are subgroups under XOR and the rep values are representatives of distinct cosets.
__device__ __host__ __forceinline__ int conv_bf16s_chan(int n, int Nout) {
  if (Nout % 128 == 0) {
    const int m = (n >> 4) & 7, r = n & 15;
    const int rep = ((m >> 1) & 1) | (((m >> 1) & 2) << 2);
    return (n & ~127) + (((r ^ rep) << 3) | m);
  }
  const int m = (n >> 4) & 3, r = n & 15;
  return (n & ~63) + (((r ^ ((m >> 1) << 1)) << 2) | m);
}
// inverse: the position (inside its 128- / 64-block) of block-relative channel c
__device__ __host__ __forceinline__ int conv_bf16s_pos(int c, int Nout) {
  if (Nout % 128 == 0) {
    const int m = c & 7;
    const int rep = ((m >> 1) & 1) | (((m >> 1) & 2) << 2);
    return m * 16 + ((c >> 3) ^ rep);
  }
  const int m = c & 3;
  return m * 16 + ((c >> 2) ^ ((m >> 1) << 1));
}

// Input-gradient packs transpose the filter: item i = (t, q, n) takes a few consecutive floats of filter ROW n, and with i
// running along n the 64 lanes of a wave touch 64 different rows -- 16 or 32 bytes of each 128-byte line, the rest of the line
// being fetched again by other blocks on other XCDs (measured: the pack of a step read 8x its bytes, 0.2 ms).  The items
// are therefore dealt to the threads in tiles of QL consecutive q x 64 / QL consecutive n: QL lanes share a row's line.  A
// bijection of [0, T x Q x Cin) when Q % QL == 0 and Cin % (64 / QL) == 0, else the identity.
__device__ __forceinline__ int64_t dgrad_item(int64_t i, int Cin, int Q, int QL) {
  const int NL = 64 / QL;
  if (Q % QL != 0 || Cin % NL != 0) return i;
  const int l = (int)(i & 63);
  const int64_t blk = i >> 6;
  const int q_l = l % QL, n_l = l / QL;
  const int nt_n = Cin / NL;
  const int nt = (int)(blk % nt_n);
  const int64_t rest = blk / nt_n;
  const int qt = (int)(rest % (Q / QL));
  const int64_t t = rest / (Q / QL);
  return (t * Q + qt * QL + q_l) * Cin + nt * NL + n_l;
}

// conv3x3 fp32, K4-interleaved: wp[t][q][n][j] = B_t[k = 4q + j][n]; forward B_t[ci][co] = w[t][ci][co],
// dgrad B_t[co][ci] = w[8 - t][ci][co].  Items: 9 * Cin * Cout / 4.
__device__ __forceinline__ void conv3x3_f32(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wp_fwd,
                                            float* __restrict__ wp_dgrad, int64_t i) {
  if (wp_fwd != nullptr) {
    const int n = (int)(i % Cout);
    const int64_t r = i / Cout;
    const int q = (int)(r % (Cin / 4));
    const int t = (int)(r / (Cin / 4));
    const float* s = w + ((int64_t)t * Cin + 4 * q) * Cout + n;
    stg4(wp_fwd + i * 4, make_float4(s[0], s[Cout], s[2 * (int64_t)Cout], s[3 * (int64_t)Cout]));
  }
  if (wp_dgrad != nullptr) {
    i = dgrad_item(i, Cin, Cout / 4, 8);
    const int n = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int q = (int)(r % (Cout / 4));
    const int t = (int)(r / (Cout / 4));
    const float* s = w + ((int64_t)(8 - t) * Cin + n) * Cout + 4 * q;
    stg4(wp_dgrad + i * 4, make_float4(s[0], s[1], s[2], s[3]));
  }
}

// conv3x3 bf16, K8-interleaved: wp[t][q][n][j] = bf16(B_t[k = 8q + j][n]).  Items: 9 * Cin * Cout / 8.
__device__ __forceinline__ void conv3x3_bf16(const float* __restrict__ w, int Cin, int Cout, uint4* __restrict__ wp_fwd,
                                             uint4* __restrict__ wp_dgrad, int perm, int64_t i) {
  if (wp_fwd != nullptr) {
    int n = (int)(i % Cout);
    if (perm) n = conv_bf16s_chan(n, Cout);
    const int64_t r = i / Cout;
    const int q = (int)(r % (Cin / 8));
    const int t = (int)(r / (Cin / 8));
    const float* s = w + ((int64_t)t * Cin + 8 * q) * Cout + n;
    const int64_t cs = Cout;
    uint4 v;
    v.x = pk2(s[0], s[cs]);
    v.y = pk2(s[2 * cs], s[3 * cs]);
    v.z = pk2(s[4 * cs], s[5 * cs]);
    v.w = pk2(s[6 * cs], s[7 * cs]);
    wp_fwd[i] = v;
  }
  if (wp_dgrad != nullptr) {
    i = dgrad_item(i, Cin, Cout / 8, 4);
    int n = (int)(i % Cin);
    if (perm) n = conv_bf16s_chan(n, Cin);
    const int64_t r = i / Cin;
    const int q = (int)(r % (Cout / 8));
    const int t = (int)(r / (Cout / 8));
    const float* s = w + ((int64_t)(8 - t) * Cin + n) * Cout + 8 * q;
    uint4 v;
    v.x = pk2(s[0], s[1]);
    v.y = pk2(s[2], s[3]);
    v.z = pk2(s[4], s[5]);
    v.w = pk2(s[6], s[7]);
    wp_dgrad[i] = v;
  }
}

// k = s transposed conv (one depth tap), w = TF [2][2][Cout][Cin].  fp32 items: Cin * Cout; bf16 items: Cin * Cout / 2.
__device__ __forceinline__ void deconv_f32(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wp_fwd,
                                           float* __restrict__ wp_dgrad, int64_t i) {
  const int Nf = 4 * Cout;
  if (wp_fwd != nullptr) {
    const int n = (int)(i % Nf);
    const int q = (int)(i / Nf);
    stg4(wp_fwd + i * 4, ldg4(w + (int64_t)n * Cin + 4 * q));
  }
  if (wp_dgrad != nullptr) {
    const int n = (int)(i % Cin);
    const int q = (int)(i / Cin);
    const float* s = w + (int64_t)(4 * q) * Cin + n;
    stg4(wp_dgrad + i * 4, make_float4(s[0], s[Cin], s[2 * (int64_t)Cin], s[3 * (int64_t)Cin]));
  }
}

__device__ __forceinline__ void deconv_bf16(const float* __restrict__ w, int Cin, int Cout, uint4* __restrict__ wp_fwd,
                                            uint4* __restrict__ wp_dgrad, int perm, int64_t i) {
  const int Nf = 4 * Cout;
  if (wp_fwd != nullptr) {
    int n = (int)(i % Nf);
    if (perm) n = perm64(n);
    const int q = (int)(i / Nf);
    const float* s = w + (int64_t)n * Cin + 8 * q;
    uint4 v;
    v.x = pk2(s[0], s[1]); v.y = pk2(s[2], s[3]); v.z = pk2(s[4], s[5]); v.w = pk2(s[6], s[7]);
    wp_fwd[i] = v;
  }
  if (wp_dgrad != nullptr) {
    int n = (int)(i % Cin);
    if (perm) n = perm64(n);
    const int q = (int)(i / Cin);
    const float* s = w + (int64_t)(8 * q) * Cin + n;
    const int64_t cs = Cin;
    uint4 v;
    v.x = pk2(s[0], s[cs]); v.y = pk2(s[2 * cs], s[3 * cs]);
    v.z = pk2(s[4 * cs], s[5 * cs]); v.w = pk2(s[6 * cs], s[7 * cs]);
    wp_dgrad[i] = v;
  }
}

}  // namespace unetk_pack
