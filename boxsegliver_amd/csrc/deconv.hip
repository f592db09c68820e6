// Transposed conv with kernel == stride (+bias, ReLU) written straight into the skip-concat buffer, and its
// backward.  2-D: slim.conv2d_transpose(x, C/2, 2, 2) + tf.concat((skip, up), -1) at NetworksV2/UNet.py:91-93.
// 3-D: slim.conv3d_transpose(x, c, (1,2,2)|(2,2,2), stride = kernel, biases_initializer=None) + tf.concat at
// NetworksV2/UNet3D.py:161-163.
//
// With kernel == stride there is no overlap: out[n,(kd z+a),2y+b,2x+c,co] = sum_ci x[n,z,y,x,ci] * w[a,b,c,co,ci],
// i.e. per depth tap a ONE dense GEMM  [pixels x Cin] . [Cin x (4*Cout)]  whose epilogue scatters each (b,c)
// column group to its output pixel -- genuinely dense, so it runs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32).  Depth taps address their output planes in place (ImgAddr), no copies.
//   forward : M = input pixels, N = (b,c,co), K = ci      epilogue: +bias, ReLU, scatter into the concat buffer
//   dgrad   : M = input pixels, N = ci,       K = (b,c,co) A rows gathered from the 4 output pixels
//   wgrad   : per (a,b,c): [co x ci] = sum_pixels dpre[pix(a,b,c)][co] * x[pix][ci]  (split-K slabs, fixed order)
#include "common.h"
#include "pack.h"

namespace {

__device__ float kZeroF4[4] = {};        // what a row past M reads in the bf16c pw_gemm (NOT const: see conv_igemm.hip)

constexpr int CK = 16, PS = 20;

struct PwParams {
  const float* a;     // fwd: x [M][Cin]; dgrad: dpre planes [.,2H,2W,Cout]
  const float* wp;    // K4-interleaved [K/4][Ncols][4]
  const float* bias;  // fwd only
  float* out;
  int M, K, Ncols;
  int H, W, Cout;     // input spatial dims (per plane), deconv output channels
  int out_stride, out_coff;
  int n_ntiles;
  ImgAddr oa;         // fwd: output plane of input plane nn;  dgrad: dpre plane of input plane nn
  int accumulate;     // dgrad: dx += (second depth tap)
  int bf16;           // UNETK_BF16: wp is the bf16 K8 pack
};

// Epilogue shared by the fp32 and bf16 kernels.  Forward: bias + ReLU + scatter of GEMM row m = input pixel (nn, yy, xx),
// column n = (tap ab, co) to output pixel (2yy + ab/2, 2xx + ab%2).  The (nn, yy, xx) decomposition is done ONCE per
// 32-row fragment and advanced incrementally along the fragment's rows (a division per stored element used to cost as
// much VALU time as the whole K = 128 main loop).
template <int MODE, int TM, int TN>
__device__ __forceinline__ void pw_epilogue(const PwParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int l31,
                                            int h) {
  if (MODE == 0) {
    int64_t tapoff[TN];
    float bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + l31;
      const int ab = n / p.Cout, co = n - ab * p.Cout;
      bv[tn] = p.bias ? p.bias[co] : 0.f;
      tapoff[tn] = ((int64_t)(ab >> 1) * 2 * p.W + (ab & 1)) * p.out_stride + p.out_coff + co;
    }
    const int HW = p.H * p.W;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int mb = m0 + (wm * TM + tm) * 32 + 4 * h;      // fragment row of register 0
      int nn = mb / HW;
      const int rem = mb - nn * HW;
      int yy = rem / p.W, xx = rem - yy * p.W;
      int64_t img = p.oa.off(nn);
      int prev = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int delta = (r & 3) + 8 * (r >> 2);           // mfma32_row(r, h) - 4 h
        xx += delta - prev;
        prev = delta;
        while (xx >= p.W) {
          xx -= p.W;
          if (++yy == p.H) {
            yy = 0;
            img = p.oa.off(++nn);
          }
        }
        if (mb + delta >= p.M) break;
        const int64_t ob = img + ((int64_t)(2 * yy) * 2 * p.W + 2 * xx) * p.out_stride;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) p.out[ob + tapoff[tn]] = fmaxf(acc[tm][tn][r] + bv[tn], 0.f);
      }
    }
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 32 + l31;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + tm) * 32 + mfma32_row(r, h);
          if (m >= p.M) continue;
          float* o = p.out + (int64_t)m * p.Ncols + n;
          *o = p.accumulate ? *o + acc[tm][tn][r] : acc[tm][tn][r];
        }
    }
  }
}

// UNETK_BF16S epilogue: the output is bf16 and the filter pack of this mode permutes the GEMM columns inside every
// 64-column block (MFMA column l of tile tn = column 2 l + tn), so a lane owns two ADJACENT columns and stores them as
// one 4-byte word: a half-wave writes 128 contiguous bytes per row.  Forward: bias + ReLU + scatter; dgrad: plain rows.
template <int MODE, int TM>
__device__ __forceinline__ void pw_epilogue_bf16s(const PwParams& p, f32x16 (&acc)[TM][2], int m0, int n0, int wm, int wn,
                                                  int l31, int h) {
  bf16_t* outb = reinterpret_cast<bf16_t*>(p.out);
  const int n = n0 + wn * 64 + 2 * l31;                      // this lane's column pair (n, n + 1)
  if (MODE == 0) {
    const int ab = n / p.Cout, co = n - ab * p.Cout;         // Cout % 64 == 0: the pair shares a tap
    const float b0 = p.bias ? p.bias[co] : 0.f, b1 = p.bias ? p.bias[co + 1] : 0.f;
    const int64_t tapoff = ((int64_t)(ab >> 1) * 2 * p.W + (ab & 1)) * p.out_stride + p.out_coff + co;
    const int HW = p.H * p.W;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int mb = m0 + (wm * TM + tm) * 32 + 4 * h;
      int nn = mb / HW;
      const int rem = mb - nn * HW;
      int yy = rem / p.W, xx = rem - yy * p.W;
      int64_t img = p.oa.off(nn);
      int prev = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int delta = (r & 3) + 8 * (r >> 2);
        xx += delta - prev;
        prev = delta;
        while (xx >= p.W) {
          xx -= p.W;
          if (++yy == p.H) {
            yy = 0;
            img = p.oa.off(++nn);
          }
        }
        if (mb + delta >= p.M) break;
        const int64_t ob = img + ((int64_t)(2 * yy) * 2 * p.W + 2 * xx) * p.out_stride;
        *reinterpret_cast<uint32_t*>(outb + ob + tapoff) =
            unetk_pk_bf16(fmaxf(acc[tm][0][r] + b0, 0.f), fmaxf(acc[tm][1][r] + b1, 0.f));
      }
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + tm) * 32 + mfma32_row(r, h);
        if (m >= p.M) continue;
        *reinterpret_cast<uint32_t*>(outb + (int64_t)m * p.Ncols + n) = unetk_pk_bf16(acc[tm][0][r], acc[tm][1][r]);
      }
  }
}

// MODE 0: forward (scatter epilogue), MODE 1: dgrad (gather prologue)
template <int MODE, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64, (TM * TN == 8 ? 2 : 1)) void pw_gemm_kernel(PwParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_F = BM * PS, WB_F = CK * BN;
  constexpr int AR = (BM * 4) / NT, WR = (CK / 4 * BN) / NT;
  static_assert((BM * 4) % NT == 0 && (CK / 4 * BN) % NT == 0, "tile must split evenly");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* abuf = smem;              // [2][A_F]
  float* wbuf = smem + 2 * A_F;    // [2][WB_F]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = bid % p.n_ntiles, mtile = bid / p.n_ntiles;
  const int m0 = mtile * BM, n0 = ntile * BN;

  int64_t aoff[AR];
  bool aok[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int idx = tid + r * NT;
    const int row = idx >> 2, q = idx & 3;
    const int m = m0 + row;
    aok[r] = m < p.M;
    if (MODE == 0) {
      aoff[r] = (int64_t)m * p.K + q * 4;
    } else {
      const int xx = m % p.W;
      const int yy = (m / p.W) % p.H;
      const int nn = m / (p.W * p.H);
      aoff[r] = p.oa.off(nn) + ((int64_t)(2 * yy) * 2 * p.W + 2 * xx) * p.Cout + q * 4;
    }
  }
  int64_t woff[WR];
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    const int idx = tid + r * NT;
    const int q = idx / BN, n = idx - q * BN;
    woff[r] = ((int64_t)q * p.Ncols + n0 + n) * 4;
  }

  float4 areg[AR], wreg[WR];
  auto load_a = [&](int s) {
    int64_t koff;
    if (MODE == 0) {
      koff = (int64_t)s * CK;
    } else {
      const int k0 = s * CK;
      const int ab = k0 / p.Cout, co0 = k0 - ab * p.Cout;
      koff = ((int64_t)(ab >> 1) * 2 * p.W + (ab & 1)) * p.Cout + co0;
    }
    // (Round 5 tried the zero-page pointer select of the conv kernels here too: on the 128 x 128 configuration in use the compiler
    // already waits for these loads behind the MFMAs, and the 64-bit selects cost this short-K kernel 1-2 %: 0.3340 -> 0.3417 ms
    // at the 256^2 level.  Kept as it was; the 256-row configuration does carry the early wait, tools/asm_lint.py.)
#pragma unroll
    for (int r = 0; r < AR; ++r) areg[r] = aok[r] ? ldg4(p.a + aoff[r] + koff) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const int idx = tid + r * NT;
      *reinterpret_cast<float4*>(&abuf[buf * A_F + (idx >> 2) * PS + (idx & 3) * 4]) = areg[r];
    }
  };
  auto load_w = [&](int s) {
    const float* base = p.wp + (int64_t)s * (CK / 4) * p.Ncols * 4;
#pragma unroll
    for (int r = 0; r < WR; ++r) wreg[r] = ldg4(base + woff[r]);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int r = 0; r < WR; ++r) *reinterpret_cast<float4*>(&wbuf[buf * WB_F + (tid + r * NT) * 4]) = wreg[r];
  };

  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) abase[tm] = ((wm * TM + tm) * 32 + l31) * PS + 4 * h;
  const int bbase = (h * BN + wn * TN * 32 + l31) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int nsteps = p.K / CK;
  load_a(0);
  load_w(0);
  store_a(0);
  store_w(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool has_next = s + 1 < nsteps;
    if (has_next) {
      load_a(s + 1);
      load_w(s + 1);
    }
    __builtin_amdgcn_sched_barrier(0);       // the requests stay in front of the step's MFMAs
    const float* ab_ = abuf + (s & 1) * A_F;
    const float* wb = wbuf + (s & 1) * WB_F;
#pragma unroll
    for (int g = 0; g < CK / 8; ++g) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const float4*>(&ab_[abase[tm] + 8 * g]);
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
    if (has_next) {
      store_a((s + 1) & 1);
      store_w((s + 1) & 1);
    }
    __syncthreads();
  }

  // epilogue
  pw_epilogue<MODE, TM, TN>(p, acc, m0, n0, wm, wn, l31, h);
}

// ---- UNETK_BF16 variant: same GEMM, operands rounded to bf16 on their way into LDS (A) / pre-packed bf16
// K8-interleaved [K/8][Ncols][8] (B), v_mfma_f32_32x32x16_bf16, fp32 accumulate; K chunk = 32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int CKB = 32, PSQ = 5;   // LDS row stride in 16-B units: 64 B of bf16 + 16 B pad
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native 16-B register type (see conv_igemm_bf16.hip)

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// BS = UNETK_BF16S: the A operand is already bf16 in memory (copied, not converted) and the output is bf16.
template <int MODE, int WM, int WN, int TM, int TN, bool BS = false>
__global__ __launch_bounds__(WM* WN * 64) void pw_gemm_bf16_kernel(PwParams p) {
  static_assert(!BS || TN == 2, "bf16 storage stores column pairs");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_Q = BM * PSQ, WB_Q = CKB / 8 * BN;
  constexpr int AR = (BM * 4) / NT, WR = WB_Q / NT;
  static_assert((BM * 4) % NT == 0 && WB_Q % NT == 0, "tile must split evenly");

  extern __shared__ __attribute__((aligned(16))) uint4 smem_q[];
  uint4* abuf = smem_q;              // [2][A_Q]
  uint4* wbuf = smem_q + 2 * A_Q;    // [2][WB_Q]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = bid % p.n_ntiles, mtile = bid / p.n_ntiles;
  const int m0 = mtile * BM, n0 = ntile * BN;

  int64_t aoff[AR];
  bool aok[AR];
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int idx = tid + r * NT;
    const int row = idx >> 2, q = idx & 3;
    const int m = m0 + row;
    aok[r] = m < p.M;
    if (MODE == 0) {
      aoff[r] = (int64_t)m * p.K + q * 8;
    } else {
      const int xx = m % p.W;
      const int yy = (m / p.W) % p.H;
      const int nn = m / (p.W * p.H);
      aoff[r] = p.oa.off(nn) + ((int64_t)(2 * yy) * 2 * p.W + 2 * xx) * p.Cout + q * 8;
    }
  }
  const uint4* wq = reinterpret_cast<const uint4*>(p.wp);
  int woff[WR];
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    const int idx = tid + r * NT;
    const int q = idx / BN, n = idx - q * BN;
    woff[r] = q * p.Ncols + n0 + n;
  }

  float4 areg[BS ? 1 : AR][2];
  u32x4 aq[BS ? AR : 1];
  u32x4 wreg[WR];
  if constexpr (BS) {   // rows past M read the tensor's first bytes and are zeroed by a select (no branch around the loads)
#pragma unroll
    for (int r = 0; r < AR; ++r) aoff[r] = aok[r] ? aoff[r] : 0;
  }
  const bf16_t* ab16 = reinterpret_cast<const bf16_t*>(p.a);
  auto load_a = [&](int s) {
    int64_t koff;
    if (MODE == 0) {
      koff = (int64_t)s * CKB;
    } else {
      const int k0 = s * CKB;
      const int ab = k0 / p.Cout, co0 = k0 - ab * p.Cout;
      koff = ((int64_t)(ab >> 1) * 2 * p.W + (ab & 1)) * p.Cout + co0;
    }
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      if constexpr (BS) {
        // the request only; rows past M are zeroed where the value is USED (store_a): a select on the loaded value here made the
        // wave wait for the load it had just issued, in front of the step's MFMAs (round 5)
        aq[r] = *reinterpret_cast<const u32x4*>(ab16 + aoff[r] + koff);
      } else {
        const float* src = aok[r] ? p.a + aoff[r] + koff : kZeroF4;      // rows past M read zeros: no branch around the request
        areg[r][0] = ldg4(src);
        areg[r][1] = ldg4(aok[r] ? src + 4 : kZeroF4);
      }
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const int idx = tid + r * NT;
      u32x4 v;
      if constexpr (BS) {
        v = aok[r] ? aq[r] : u32x4{0u, 0u, 0u, 0u};
      } else {
        v.x = pk_bf16(areg[r][0].x, areg[r][0].y);
        v.y = pk_bf16(areg[r][0].z, areg[r][0].w);
        v.z = pk_bf16(areg[r][1].x, areg[r][1].y);
        v.w = pk_bf16(areg[r][1].z, areg[r][1].w);
      }
      *reinterpret_cast<u32x4*>(&abuf[buf * A_Q + (idx >> 2) * PSQ + (idx & 3)]) = v;
    }
  };
  auto load_w = [&](int s) {
    const uint4* base = wq + (int64_t)s * (CKB / 8) * p.Ncols;
#pragma unroll
    for (int r = 0; r < WR; ++r) wreg[r] = *reinterpret_cast<const u32x4*>(base + woff[r]);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int r = 0; r < WR; ++r) *reinterpret_cast<u32x4*>(&wbuf[buf * WB_Q + tid + r * NT]) = wreg[r];
  };

  int abase[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) abase[tm] = ((wm * TM + tm) * 32 + l31) * PSQ + h;
  const int bbase = h * BN + wn * TN * 32 + l31;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int nsteps = p.K / CKB;
  load_a(0);
  load_w(0);
  store_a(0);
  store_w(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool has_next = s + 1 < nsteps;
    if (has_next) {
      load_a(s + 1);
      load_w(s + 1);
    }
    __builtin_amdgcn_sched_barrier(0);       // the requests stay in front of the step's MFMAs
    const uint4* ab_ = abuf + (s & 1) * A_Q;
    const uint4* wb = wbuf + (s & 1) * WB_Q;
#pragma unroll
    for (int g = 0; g < CKB / 16; ++g) {
      uint4 a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = ab_[abase[tm] + 2 * g];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = wb[bbase + 2 * g * BN + tn * 32];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[tm]),
                                                                __builtin_bit_cast(bf16x8, b[tn]), acc[tm][tn], 0, 0, 0);
    }
    if (has_next) {
      store_a((s + 1) & 1);
      store_w((s + 1) & 1);
    }
    __syncthreads();
  }

  if constexpr (BS) pw_epilogue_bf16s<MODE, TM>(p, acc, m0, n0, wm, wn, l31, h);
  else pw_epilogue<MODE, TM, TN>(p, acc, m0, n0, wm, wn, l31, h);
}

// dpre[pix][co] = dcat[pix][coff+co] * (cat[pix][coff+co] > 0); partial[blk][co] = column sums (bias grad)
// T = bf16_t (UNETK_BF16S): cat, dcat and dpre are bf16; dcat * mask is exact in bf16, the column sums are fp32.
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(const T* __restrict__ cat, const T* __restrict__ dcat,
                                                            int stride, int coff, T* __restrict__ dpre,
                                                            float* __restrict__ partial, int64_t npix, int C, int cq_n,
                                                            int rpi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [rpi][C]
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < rpi) {
    for (int64_t pix = (int64_t)blockIdx.x * rpi + rl; pix < npix; pix += (int64_t)gridDim.x * rpi) {
      const float4 v = ld4(cat + pix * stride + coff + cq * 4);
      const float4 d = ld4(dcat + pix * stride + coff + cq * 4);
      float4 o;
      o.x = v.x > 0.f ? d.x : 0.f;
      o.y = v.y > 0.f ? d.y : 0.f;
      o.z = v.z > 0.f ? d.z : 0.f;
      o.w = v.w > 0.f ? d.w : 0.f;
      st4(dpre + pix * C + cq * 4, o);
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    stg4(&smem[rl * C + cq * 4], s);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int j = 0; j < rpi; ++j) t += smem[j * C + c];
    partial[(int64_t)blockIdx.x * C + c] = t;
  }
}

// wgrad: slab[split][bc][co][ci] = sum over the split's pixels of dpre[pix(m,bc)][co] * x[m][ci]
struct DwParams {
  const float* x;
  const float* dpre;
  float* slab;
  int M, H, W, Cin, Cout;
  int m_per_split, n_co_tiles, n_ci_tiles;
  ImgAddr da;   // dpre plane of input plane nn
  int bf16;     // UNETK_BF16: a k-step is 16 pixels, each lane rounds its 8 rows of dpre / x to bf16
};

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  uint4 q;
  q.x = pk_bf16(v[0], v[1]);
  q.y = pk_bf16(v[2], v[3]);
  q.z = pk_bf16(v[4], v[5]);
  q.w = pk_bf16(v[6], v[7]);
  return __builtin_bit_cast(bf16x8, q);
}

// BF is a template parameter, not a run-time branch: with both MFMA loops in one kernel the compiler shuttled the
// accumulator between VGPRs and AGPRs once per tile and read a15 back ten wait-states after the last 16-pass MFMA of
// the loop -- too early (rows 27 / 31 of each 32-row panel lost the final k-step, non-reproducibly).  One loop per
// instantiation keeps the accumulator in place for the whole split.
template <bool BF>
__global__ __launch_bounds__(256) void deconv_wgrad_kernel(DwParams p) {
  constexpr int KT = 128, CT = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* at = smem;             // [KT][CT]  dpre rows (co)
  float* bt = smem + KT * CT;   // [KT][CT]  x rows (ci)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;
  const int ab = bid & 3;
  const int split = bid >> 2;
  const int co0 = co_t * CT, ci0 = ci_t * CT;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int mb = split * p.m_per_split, me = min(mb + p.m_per_split, p.M);
  // register prefetch: the next 128-pixel tile is in flight while the MFMAs of the current one run
  constexpr int LR = KT * (CT / 4) / 256;   // 8 float4 of each operand per thread
  float4 va[LR], vb[LR];
  // thread -> rows (tid >> 4) + 16 i of the tile, float4 q = tid & 15.  The pixel decomposition is done once per tile
  // and advanced by 16 per row: with a division per row the kernel was VALU-bound (5 divisions x 8 rows per 64 MFMAs).
  const int HWp = p.H * p.W;
  auto load_tile = [&](int mt) {
    const int m_first = mt + (tid >> 4), q = tid & 15;
    int nn = m_first / HWp;
    const int rem = m_first - nn * HWp;
    int yy = rem / p.W, xx = rem - yy * p.W;
    int64_t img = p.da.off(nn);
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      const int m = m_first + 16 * i;
      va[i] = vb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < me) {
        const int64_t o = img + ((int64_t)(2 * yy + (ab >> 1)) * 2 * p.W + 2 * xx + (ab & 1)) * p.Cout;
        if (co0 + q * 4 < p.Cout) va[i] = ldg4(p.dpre + o + co0 + q * 4);      // Cout may be 32: half a co tile
        vb[i] = ldg4(p.x + (int64_t)m * p.Cin + ci0 + q * 4);
      }
      xx += 16;
      while (xx >= p.W) {
        xx -= p.W;
        if (++yy == p.H) {
          yy = 0;
          img = p.da.off(++nn);
        }
      }
    }
  };
  if (mb < me) load_tile(mb);
  for (int mt = mb; mt < me; mt += KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      const int idx = tid + i * 256;
      *reinterpret_cast<float4*>(&at[(idx >> 4) * CT + (idx & 15) * 4]) = va[i];
      *reinterpret_cast<float4*>(&bt[(idx >> 4) * CT + (idx & 15) * 4]) = vb[i];
    }
    __syncthreads();
    if (mt + KT < me) load_tile(mt + KT);
    if constexpr (BF) {
#pragma unroll 2
      for (int s = 0; s < KT / 16; ++s) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          av[j] = at[(16 * s + 8 * h + j) * CT + wco * 32 + l31];
          bv[j] = bt[(16 * s + 8 * h + j) * CT + wci * 32 + l31];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(av), pack8(bv), acc, 0, 0, 0);
      }
      continue;
    }
#pragma unroll 8
    for (int s = 0; s < KT / 2; ++s) {
      const float a = at[(2 * s + h) * CT + wco * 32 + l31];
      const float b = bt[(2 * s + h) * CT + wci * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  float* out = p.slab + ((int64_t)split * 4 + ab) * p.Cout * p.Cin;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wco * 32 + mfma32_row(r, h);
    if (co < p.Cout) out[(int64_t)co * p.Cin + ci0 + wci * 32 + l31] = acc[r];
  }
}

// UNETK_BF16S filter gradient: dpre and x are bf16 in memory; their 128-pixel x 64-channel tiles stay bf16 in LDS
// ([pixel][64 channels], 128-byte rows, 16-byte chunks XOR-swizzled by bit 1 of the row as in conv_wgrad_bf16s.hip) and
// both k-strided MFMA operands come out of the LDS transpose path: 4 ds_read_b64_tr_b16 per 32x32x16 MFMA instead of the
// 16 ds_read_b32 + 8 conversions of the fp32-storage variant.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ uint2 tr_read(const char* base, int byte_off) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + byte_off)));
}

__global__ __launch_bounds__(256) void deconv_wgrad_bf16s_kernel(DwParams p) {
  constexpr int KT = 128, CT = 64;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  char* at = smem_c;                 // [KT][128 B]  dpre rows (co)
  char* bt = smem_c + KT * 128;      // [KT][128 B]  x rows (ci)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;
  const int ab = bid & 3;
  const int split = bid >> 2;
  const int co0 = co_t * CT, ci0 = ci_t * CT;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* db = reinterpret_cast<const bf16_t*>(p.dpre);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int mb = split * p.m_per_split, me = min(mb + p.m_per_split, p.M);
  constexpr int LR = KT * 8 / 256;      // 4 sixteen-byte chunks of each operand per thread
  uint4 va[LR], vb[LR];
  const int HWp = p.H * p.W;
  // thread -> rows (tid >> 3) + 32 i of the tile, chunk q = tid & 7
  auto load_tile = [&](int mt) {
    const int m_first = mt + (tid >> 3), q = tid & 7;
    int nn = m_first / HWp;
    const int rem = m_first - nn * HWp;
    int yy = rem / p.W, xx = rem - yy * p.W;
    int64_t img = p.da.off(nn);
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      const int m = m_first + 32 * i;
      va[i] = vb[i] = make_uint4(0u, 0u, 0u, 0u);
      if (m < me) {
        const int64_t o = img + ((int64_t)(2 * yy + (ab >> 1)) * 2 * p.W + 2 * xx + (ab & 1)) * p.Cout;
        if (co0 + q * 8 < p.Cout) va[i] = *reinterpret_cast<const uint4*>(db + o + co0 + q * 8);
        vb[i] = *reinterpret_cast<const uint4*>(xb + (int64_t)m * p.Cin + ci0 + q * 8);
      }
      xx += 32;
      while (xx >= p.W) {
        xx -= p.W;
        if (++yy == p.H) {
          yy = 0;
          img = p.da.off(++nn);
        }
      }
    }
  };
  // fragment addressing (see conv_wgrad_bf16s.hip): 16-lane group -> channel 16-block (g & 1), pixel half h
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, g16 = (lane >> 4) & 1;
  const int sw = (q4 >> 1) << 2;
  const int a_off = (8 * h + q4) * 128 + (((2 * (wco * 2 + g16) + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1);
  const int b_off = (8 * h + q4) * 128 + (((2 * (wci * 2 + g16) + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1);
  if (mb < me) load_tile(mb);
  for (int mt = mb; mt < me; mt += KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      const int row = (tid >> 3) + 32 * i, q = tid & 7;
      const int o = row * 128 + ((q ^ (((row >> 1) & 1) << 2)) << 4);
      *reinterpret_cast<uint4*>(at + o) = va[i];
      *reinterpret_cast<uint4*>(bt + o) = vb[i];
    }
    __syncthreads();
    if (mt + KT < me) load_tile(mt + KT);
#pragma unroll
    for (int s = 0; s < KT / 16; ++s) {
      const uint2 a0 = tr_read(at, s * 16 * 128 + a_off), a1 = tr_read(at, s * 16 * 128 + 4 * 128 + a_off);
      const uint2 b0 = tr_read(bt, s * 16 * 128 + b_off), b1 = tr_read(bt, s * 16 * 128 + 4 * 128 + b_off);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y)),
                                                    __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y)), acc, 0, 0, 0);
    }
  }
  float* out = p.slab + ((int64_t)split * 4 + ab) * p.Cout * p.Cin;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wco * 32 + mfma32_row(r, h);
    if (co < p.Cout) out[(int64_t)co * p.Cin + ci0 + wci * 32 + l31] = acc[r];
  }
}

// Round 3: the same contraction with every byte read ONCE per (co, ci) panel and four times the MFMAs per LDS read.  The kernel
// above is HBM-bound at a quarter of the stream rate: a block owns ONE sub-pixel of a 64 x 64 panel, so x is fetched 4 x Cout/64
// times and dpre Cin/64 times, and a wave issues 4 transposed reads per MFMA.  Here a 512-thread block owns all FOUR sub-pixels
// of a 64 (co) x 128 (ci) panel: a 64-pixel tile = four dpre images [64 px][64 co] + two x images [64 px][64 ci] (48 KB of LDS,
// same swizzle), wave (ab, ci half) holds 2 x 2 accumulators (8 transposed reads per 4 MFMAs), the next tile waits in
// registers (6 x 16 B per thread).  Needs Cin % 128 == 0 and Cout % 64 == 0 (every transposed conv of UNet / GUNet).
__global__ __launch_bounds__(512) void deconv_wgrad_bf16s4_kernel(DwParams p) {
  constexpr int KT = 64;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];   // [4 ab][KT][128 B] dpre, [2 halves][KT][128 B] x
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ab_w = wave >> 1, cih = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;
  const int split = bid;
  const int co0 = co_t * 64, ci0 = ci_t * 128;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* db = reinterpret_cast<const bf16_t*>(p.dpre);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int mb = split * p.m_per_split, me = min(mb + p.m_per_split, p.M);
  // thread -> pixel row tid >> 3 of the tile, 16-byte chunk q = tid & 7 of each of the six images
  const int row = tid >> 3, q = tid & 7;
  const int HWp = p.H * p.W;
  int nn, yy, xx;
  {
    const int m0 = mb + row;
    nn = m0 / HWp;
    const int rem = m0 - nn * HWp;
    yy = rem / p.W; xx = rem - yy * p.W;
  }
  uint4 va[4], vb[2];
  auto load_tile = [&](int mt) {      // (nn, yy, xx) = pixel mt + row; advanced by KT afterwards
    const int m = mt + row;
#pragma unroll
    for (int i = 0; i < 4; ++i) va[i] = make_uint4(0u, 0u, 0u, 0u);
    vb[0] = vb[1] = make_uint4(0u, 0u, 0u, 0u);
    if (m < me) {
      const int64_t o = p.da.off(nn) + ((int64_t)(2 * yy) * 2 * p.W + 2 * xx) * p.Cout + co0 + q * 8;
      va[0] = *reinterpret_cast<const uint4*>(db + o);
      va[1] = *reinterpret_cast<const uint4*>(db + o + p.Cout);
      va[2] = *reinterpret_cast<const uint4*>(db + o + (int64_t)2 * p.W * p.Cout);
      va[3] = *reinterpret_cast<const uint4*>(db + o + (int64_t)2 * p.W * p.Cout + p.Cout);
      const bf16_t* xr = xb + (int64_t)m * p.Cin + ci0 + q * 8;
      vb[0] = *reinterpret_cast<const uint4*>(xr);
      vb[1] = *reinterpret_cast<const uint4*>(xr + 64);
    }
    xx += KT;
    while (xx >= p.W) {
      xx -= p.W;
      if (++yy == p.H) { yy = 0; ++nn; }
    }
  };
  const int st_off = row * 128 + ((q ^ (((row >> 1) & 1) << 2)) << 4);
  // fragment addressing (deconv_wgrad_bf16s_kernel): 32-channel block b of a 64-channel image
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, g16 = (lane >> 4) & 1;
  const int sw = (q4 >> 1) << 2;
  int foff[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) foff[b] = (8 * h + q4) * 128 + (((2 * (b * 2 + g16) + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1);
  const char* at = smem_c + ab_w * (KT * 128);
  const char* bt = smem_c + (4 + cih) * (KT * 128);
  if (mb < me) load_tile(mb);
  for (int mt = mb; mt < me; mt += KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(smem_c + i * (KT * 128) + st_off) = va[i];
    *reinterpret_cast<uint4*>(smem_c + 4 * (KT * 128) + st_off) = vb[0];
    *reinterpret_cast<uint4*>(smem_c + 5 * (KT * 128) + st_off) = vb[1];
    __syncthreads();
    if (mt + KT < me) load_tile(mt + KT);
#pragma unroll
    for (int s = 0; s < KT / 16; ++s) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint2 a0 = tr_read(at, s * 16 * 128 + foff[b]), a1 = tr_read(at, s * 16 * 128 + 4 * 128 + foff[b]);
        const uint2 b0 = tr_read(bt, s * 16 * 128 + foff[b]), b1 = tr_read(bt, s * 16 * 128 + 4 * 128 + foff[b]);
        af[b] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
        bf[b] = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  float* out = p.slab + ((int64_t)split * 4 + ab_w) * p.Cout * p.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + i * 32 + mfma32_row(r, h);
        out[(int64_t)co * p.Cin + ci0 + cih * 64 + j * 32 + l31] = acc[i][j][r];
      }
}

// wp_fwd[q][n=(bc,co)][j] = w[bc][co][4q+j]   (K = Cin)
// wp_dgrad[q][n=ci][j]    = w_flat[(4q+j)][ci] with w_flat = [(bc,co)][ci]   (K = 4*Cout)
__global__ void pack_deconv_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wp_fwd,
                                   float* __restrict__ wp_dgrad) {
  const int64_t total = (int64_t)Cin * Cout;  // float4 count = 4*Cin*Cout/4
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    unetk_pack::deconv_f32(w, Cin, Cout, wp_fwd, wp_dgrad, i);        // csrc/pack.h
}

__global__ void pack_deconv_bf16_kernel(const float* __restrict__ w, int Cin, int Cout, uint4* __restrict__ wp_fwd,
                                        uint4* __restrict__ wp_dgrad, int perm) {
  const int64_t total = (int64_t)Cin * Cout / 2;  // 16-B units = 4*Cin*Cout/8
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    unetk_pack::deconv_bf16(w, Cin, Cout, wp_fwd, wp_dgrad, perm, i);  // csrc/pack.h
}

template <int MODE, int WM, int WN, int TM, int TN, bool BS = false>
int launch_pw_bf16(const PwParams& p, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (size_t)(2 * BM * PSQ + 2 * (CKB / 8) * BN) * 16;
  const int n_mtiles = (p.M + BM - 1) / BM;
  UNETK_LAUNCH((pw_gemm_bf16_kernel<MODE, WM, WN, TM, TN, BS>), dim3(n_mtiles * p.n_ntiles), dim3(WM * WN * 64), lds, st,
                     p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

template <int MODE, int WM, int WN, int TM, int TN>
int launch_pw(const PwParams& p, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (size_t)(2 * BM * PS + 2 * CK * BN) * sizeof(float);
  const int n_mtiles = (p.M + BM - 1) / BM;
  UNETK_LAUNCH((pw_gemm_kernel<MODE, WM, WN, TM, TN>), dim3(n_mtiles * p.n_ntiles), dim3(WM * WN * 64), lds, st, p);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

template <int MODE>
int run_pw(PwParams& p, hipStream_t st) {
  if (p.bf16 == UNETK_BF16S) {
    if (p.accumulate || p.Ncols % 64 != 0) return UNETK_E_UNSUPPORTED;
    if (p.Ncols % 128 == 0) {
      p.n_ntiles = p.Ncols / 128;
      return launch_pw_bf16<MODE, 2, 2, 2, 2, true>(p, st);
    }
    p.n_ntiles = p.Ncols / 64;
    return launch_pw_bf16<MODE, 4, 1, 1, 2, true>(p, st);
  }
  if (p.Ncols % 128 == 0) {
    p.n_ntiles = p.Ncols / 128;
    // fp32, long K: 256-row tiles while they still give two blocks per CU -- half the filter-panel staging per MFMA
    // (conv_igemm.hip).  Measured at bs 32: K = 1024 forward 0.335 -> 0.277 ms, K = 1024 input gradient -0.045 ms; K <= 512
    // is 5-10 % SLOWER (eight to thirty-two steps do not amortise the four-fragment prologue / epilogue).
    if (!p.bf16 && p.K >= 1024 && (int64_t)((p.M + 255) / 256) * p.n_ntiles >= 512) return launch_pw<MODE, 2, 2, 4, 2>(p, st);
    return p.bf16 ? launch_pw_bf16<MODE, 2, 2, 2, 2>(p, st) : launch_pw<MODE, 2, 2, 2, 2>(p, st);
  }
  p.n_ntiles = p.Ncols / 64;
  return p.bf16 ? launch_pw_bf16<MODE, 4, 1, 1, 2>(p, st) : launch_pw<MODE, 4, 1, 1, 2>(p, st);
}

// fp32 filter gradient of ALL FOUR output sub-pixels (a, b) of an input pixel in one block: the x tile is staged once and
// contracted with the four dpre tiles (four accumulators per wave, the x fragment shared by their MFMAs).  The one-
// sub-pixel kernel above re-read x four times (1.3 GB per launch against 0.8 GB algorithmic at the 128^2 level) and fed a
// single dependent MFMA chain per wave from two scalar LDS reads each: 88 TFLOP/s.
//   dw[a][b][co][ci] = sum_m dpre[2y + a][2x + b][co] * x[m][ci]
// Block = 64 co x 64 ci x 4 sub-pixels over a range of input pixels (split-K slabs as before); 32-pixel stages with a
// register prefetch of the next stage.
__global__ __launch_bounds__(256) void deconv_wgrad4_kernel(DwParams p) {
  constexpr int KT = 32, CT = 64;
  __shared__ __attribute__((aligned(16))) float bt[KT * CT];        // x rows (ci)
  __shared__ __attribute__((aligned(16))) float at[4][KT * CT];     // dpre rows (co) of the four sub-pixels
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wci = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int ci_t = bid % p.n_ci_tiles; bid /= p.n_ci_tiles;
  const int co_t = bid % p.n_co_tiles; bid /= p.n_co_tiles;
  const int split = bid;
  const int co0 = co_t * CT, ci0 = ci_t * CT;
  f32x16 acc[4];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ab][r] = 0.f;
  const int mb = split * p.m_per_split, me = min(mb + p.m_per_split, p.M);
  constexpr int LR = KT * (CT / 4) / 256;   // 2 float4 of x, 4 x 2 of dpre per thread and stage
  float4 va[4][LR], vb[LR];
  const int HWp = p.H * p.W;
  const bool co_ok = co0 + (tid & 15) * 4 < p.Cout;                  // Cout may be 32: half a co tile
  auto load_tile = [&](int mt) {
    const int m_first = mt + (tid >> 4), q = tid & 15;
    int nn = m_first / HWp;
    const int rem = m_first - nn * HWp;
    int yy = rem / p.W, xx = rem - yy * p.W;
    int64_t img = p.da.off(nn);
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      const int m = m_first + 16 * i;
      vb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) va[ab][i] = vb[i];
      if (m < me) {
        vb[i] = ldg4(p.x + (int64_t)m * p.Cin + ci0 + q * 4);
        if (co_ok) {
          const float* d0 = p.dpre + img + ((int64_t)(2 * yy) * 2 * p.W + 2 * xx) * p.Cout + co0 + q * 4;
          va[0][i] = ldg4(d0);
          va[1][i] = ldg4(d0 + p.Cout);
          va[2][i] = ldg4(d0 + (int64_t)2 * p.W * p.Cout);
          va[3][i] = ldg4(d0 + (int64_t)2 * p.W * p.Cout + p.Cout);
        }
      }
      xx += 16;
      while (xx >= p.W) {
        xx -= p.W;
        if (++yy == p.H) {
          yy = 0;
          img = p.da.off(++nn);
        }
      }
    }
  };
  if (mb < me) load_tile(mb);
  for (int mt = mb; mt < me; mt += KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      const int idx = tid + i * 256;
      const int o = (idx >> 4) * CT + (idx & 15) * 4;
      *reinterpret_cast<float4*>(&bt[o]) = vb[i];
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) *reinterpret_cast<float4*>(&at[ab][o]) = va[ab][i];
    }
    __syncthreads();
    if (mt + KT < me) load_tile(mt + KT);
#pragma unroll 4
    for (int s = 0; s < KT / 2; ++s) {
      const float b = bt[(2 * s + h) * CT + wci * 32 + l31];
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const float a = at[ab][(2 * s + h) * CT + wco * 32 + l31];
        acc[ab] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ab], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) {
    float* out = p.slab + ((int64_t)split * 4 + ab) * p.Cout * p.Cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + mfma32_row(r, h);
      if (co < p.Cout) out[(int64_t)co * p.Cin + ci0 + wci * 32 + l31] = acc[ab][r];
    }
  }
}

bool deconv_desc_ok(const unetk_deconv3d_desc* d) {
  return d && d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && (d->kd == 1 || d->kd == 2) &&
         d->out_coff >= 0 && d->out_stride >= d->out_coff + d->Cout;
}

struct DwPlan {
  int S, m_per_split, nblk_bias;
};
inline bool dw_bf16s4(const unetk_deconv3d_desc* d) {
  return d->precision == UNETK_BF16S && d->kd == 1 && d->Cin % 128 == 0 && d->Cout % 64 == 0;
}
DwPlan dw_plan(const unetk_deconv3d_desc* d) {
  DwPlan pl{};
  const int M = d->N * d->D * d->H * d->W;
  // fp32: one block takes all four sub-pixels of its (co, ci) panel (deconv_wgrad4_kernel) -- ~768 blocks of 40 KB LDS;
  // the bf16 kernels take one sub-pixel per block -- ~1024 blocks
  const bool four = d->precision == UNETK_FP32;
  if (dw_bf16s4(d)) {     // deconv_wgrad_bf16s4_kernel: 64-pixel tiles, (Cout / 64) x (Cin / 128) panels, one block per CU and more
    const int panels = (d->Cin / 128) * (d->Cout / 64);
    int S = (256 + panels - 1) / panels;
    const int mtiles = (M + 63) / 64;
    if (S > mtiles) S = mtiles;
    const int tiles_per = (mtiles + S - 1) / S;
    pl.m_per_split = tiles_per * 64;
    pl.S = (mtiles + tiles_per - 1) / tiles_per;
    const ColMap m = unetk_colmap(d->Cout);
    int64_t g = ((int64_t)4 * d->kd * M + m.rows_per_iter - 1) / m.rows_per_iter;
    if (g > UNETK_COL_BLOCKS) g = UNETK_COL_BLOCKS;
    pl.nblk_bias = (int)g;
    return pl;
  }
  const int panels = (four ? 1 : 4) * (d->Cin / 64) * ((d->Cout + 63) / 64);
  int S = ((four ? 768 : 1024) + panels - 1) / panels;
  const int mtiles = (M + 127) / 128;
  if (S > mtiles) S = mtiles;
  if (S < 1) S = 1;
  const int tiles_per = (mtiles + S - 1) / S;
  pl.m_per_split = tiles_per * 128;
  pl.S = (mtiles + tiles_per - 1) / tiles_per;
  const ColMap m = unetk_colmap(d->Cout);
  int64_t g = ((int64_t)4 * d->kd * M + m.rows_per_iter - 1) / m.rows_per_iter;
  if (g > UNETK_COL_BLOCKS) g = UNETK_COL_BLOCKS;
  pl.nblk_bias = (int)g;
  return pl;
}

unetk_deconv3d_desc from2d(const unetk_deconv_desc* d) {
  unetk_deconv3d_desc e;
  e.N = d->N; e.D = 1; e.H = d->H; e.W = d->W; e.Cin = d->Cin; e.Cout = d->Cout; e.kd = 1;
  e.out_stride = d->out_stride; e.out_coff = d->out_coff; e.precision = d->precision;
  return e;
}

// filter panel of depth tap a: 4*Cin*Cout elements per tap, 4 bytes (fp32 K4 pack) or 2 bytes (bf16 K8 pack) each
inline const float* tap_panel(const float* wp, int a, const unetk_deconv3d_desc* d) {
  const int64_t elems = (int64_t)a * 4 * d->Cin * d->Cout;
  return d->precision != UNETK_FP32 ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(wp) + elems)
                                    : wp + elems;
}

}  // namespace

static int deconv_pack_bf16_impl(const float* w, int kd, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, void* stream,
                                 int perm) {
  UNETK_REQUIRE(w && (kd == 1 || kd == 2) && Cin > 0 && Cout > 0 && (wp_fwd || wp_dgrad));
  if (Cin % 8 != 0 || Cout % 8 != 0) return UNETK_E_UNSUPPORTED;
  if (perm && (Cin % 64 != 0 || Cout % 64 != 0)) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE((!wp_fwd || unetk_aligned16(wp_fwd)) && (!wp_dgrad || unetk_aligned16(wp_dgrad)));
  const int64_t total = (int64_t)Cin * Cout / 2;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  for (int a = 0; a < kd; ++a) {
    const int64_t o = (int64_t)a * Cin * Cout / 2;        // 16-B units per tap
    UNETK_LAUNCH(pack_deconv_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       w + (int64_t)a * 4 * Cin * Cout, Cin, Cout, wp_fwd ? (uint4*)wp_fwd + o : nullptr,
                       wp_dgrad ? (uint4*)wp_dgrad + o : nullptr, perm);
    UNETK_LAUNCH_CHECK();
  }
  return UNETK_OK;
}

extern "C" int unetk_deconv3d_pack_bf16(const float* w, int kd, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                                        void* stream) {
  return deconv_pack_bf16_impl(w, kd, Cin, Cout, wp_fwd, wp_dgrad, stream, 0);
}

extern "C" int unetk_deconv2x2_pack_bf16s(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, void* stream) {
  return deconv_pack_bf16_impl(w, 1, Cin, Cout, wp_fwd, wp_dgrad, stream, 1);
}

extern "C" int unetk_deconv2x2_pack_bf16(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, void* stream) {
  return unetk_deconv3d_pack_bf16(w, 1, Cin, Cout, wp_fwd, wp_dgrad, stream);
}

extern "C" int unetk_deconv3d_pack(const float* w, int kd, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                                   void* stream) {
  UNETK_REQUIRE(w && (kd == 1 || kd == 2) && Cin > 0 && Cout > 0);
  if (Cin % 4 != 0 || Cout % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(w) && unetk_aligned16(wp_fwd) && unetk_aligned16(wp_dgrad));
  const int64_t total = (int64_t)Cin * Cout;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  for (int a = 0; a < kd; ++a) {
    const int64_t o = (int64_t)a * 4 * Cin * Cout;
    UNETK_LAUNCH(pack_deconv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w + o, Cin, Cout,
                       wp_fwd ? wp_fwd + o : nullptr, wp_dgrad ? wp_dgrad + o : nullptr);
    UNETK_LAUNCH_CHECK();
  }
  return UNETK_OK;
}

extern "C" int unetk_deconv3d_fwd(const unetk_deconv3d_desc* d, const void* xv, const void* wpv,
                                  const float* bias, void* outv, void* stream) {
  const float* x = (const float*)xv;
  const float* wp_fwd = (const float*)wpv;
  float* out = (float*)outv;
  UNETK_REQUIRE(deconv_desc_ok(d) && x && wp_fwd && out);
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(wp_fwd));
  if (d->Cin % CK != 0 || d->Cout % 16 != 0) return UNETK_E_UNSUPPORTED;
  if (d->precision == UNETK_BF16 && (d->Cin % CKB != 0 || d->Cout % 32 != 0)) return UNETK_E_UNSUPPORTED;
  if (d->precision == UNETK_BF16S && (d->kd != 1 || d->Cin % 64 != 0 || d->Cout % 64 != 0 || d->out_stride % 2 != 0 ||
                                      d->out_coff % 2 != 0))
    return UNETK_E_UNSUPPORTED;
  const int64_t plane = (int64_t)4 * d->H * d->W * d->out_stride;      // one output depth plane
  for (int a = 0; a < d->kd; ++a) {
    PwParams p{};
    p.bf16 = d->precision;
    p.a = x; p.wp = tap_panel(wp_fwd, a, d); p.bias = bias; p.out = out + a * plane;
    p.M = d->N * d->D * d->H * d->W; p.K = d->Cin; p.Ncols = 4 * d->Cout;
    p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.out_stride = d->out_stride; p.out_coff = d->out_coff;
    p.oa.group = d->D; p.oa.img_stride = d->kd * plane; p.oa.group_stride = (int64_t)d->kd * d->D * plane;
    int rc = run_pw<0>(p, (hipStream_t)stream);
    if (rc != UNETK_OK) return rc;
  }
  return UNETK_OK;
}

extern "C" size_t unetk_deconv3d_bwd_ws_bytes(const unetk_deconv3d_desc* d) {
  if (!deconv_desc_ok(d) || d->Cin % 64 != 0 || d->Cout % 32 != 0) return 0;
  const DwPlan pl = dw_plan(d);
  const size_t M = (size_t)d->N * d->D * d->H * d->W;
  size_t f = 4 * d->kd * M * d->Cout;                       // dpre
  f += (size_t)pl.nblk_bias * d->Cout;                      // bias partials
  f += unetk_rows_reduce_tmp_floats(1, pl.nblk_bias, d->Cout);
  f += (size_t)d->Cout;                                     // dbias sink when the caller passes NULL
  f = (f + 3) & ~(size_t)3;
  f += (size_t)pl.S * 4 * d->Cin * d->Cout;                 // wgrad slabs
  return f * sizeof(float);
}

extern "C" int unetk_deconv3d_bwd(const unetk_deconv3d_desc* d, const void* xv, const void* wpv,
                                  const void* catv, const void* dcatv, void* dxv, float* dw, float* dbias, void* ws,
                                  size_t ws_bytes, void* stream) {
  return unetk_deconv3d_bwd_parts(d, xv, wpv, catv, dcatv, dxv, dw, dbias, ws, ws_bytes, 3, stream);
}

// parts: bit 0 = ReLU backward + bias gradient + input gradient (writes the masked, re-laid gradient `dpre` into ws),
// bit 1 = filter gradient (reads dpre from the SAME ws: after a bit-0 call, on any stream ordered behind it).  The filter
// gradient is off backward's critical chain, so the host may run it on a second stream beside the next unit's kernels
// (ops.deconv_bwd; round 5).
extern "C" int unetk_deconv3d_bwd_parts(const unetk_deconv3d_desc* d, const void* xv, const void* wpv,
                                        const void* catv, const void* dcatv, void* dxv, float* dw, float* dbias, void* ws,
                                        size_t ws_bytes, int parts, void* stream) {
  const float* x = (const float*)xv;
  const float* wp_dgrad = (const float*)wpv;
  const float* cat = (const float*)catv;
  const float* dcat = (const float*)dcatv;
  float* dx = (float*)dxv;
  UNETK_REQUIRE(deconv_desc_ok(d) && x && wp_dgrad && cat && dcat && dx && dw && ws && parts >= 1 && parts <= 3);
  const bool bs = d->precision == UNETK_BF16S;
  if (bs && (d->kd != 1 || d->Cout % 64 != 0 || d->out_stride % 4 != 0)) return UNETK_E_UNSUPPORTED;
  if (d->Cin % 64 != 0 || d->Cout % 32 != 0 || d->out_stride % 4 != 0 || d->out_coff % 4 != 0)
    return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(wp_dgrad) && unetk_aligned16(cat) && unetk_aligned16(dcat) &&
                unetk_aligned16(dx) && unetk_aligned16(dw) && unetk_aligned16(ws));
  if (ws_bytes < unetk_deconv3d_bwd_ws_bytes(d)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const DwPlan pl = dw_plan(d);
  const int M = d->N * d->D * d->H * d->W;
  float* dpre = (float*)ws;
  float* bpart = dpre + (size_t)4 * d->kd * M * d->Cout;
  float* btmp = bpart + (size_t)pl.nblk_bias * d->Cout;
  float* bsink = btmp + unetk_rows_reduce_tmp_floats(1, pl.nblk_bias, d->Cout);
  size_t off = (size_t)(bsink + d->Cout - dpre);
  off = (off + 3) & ~(size_t)3;
  float* slab = dpre + off;

  int rc = UNETK_OK;
  if (parts & 1) {
  // 1. ReLU backward over the whole up half + bias-grad partials
  const ColMap cm = unetk_colmap(d->Cout);
  if (bs)     // dpre is bf16 in the same workspace region (half of it used)
    UNETK_LAUNCH(relu_bwd_bias_kernel<bf16_t>, dim3(pl.nblk_bias), dim3(256),
                       (size_t)cm.rows_per_iter * d->Cout * sizeof(float), st, (const bf16_t*)catv, (const bf16_t*)dcatv,
                       d->out_stride, d->out_coff, (bf16_t*)dpre, bpart, (int64_t)4 * d->kd * M, d->Cout, cm.cq_n,
                       cm.rows_per_iter);
  else
    UNETK_LAUNCH(relu_bwd_bias_kernel<float>, dim3(pl.nblk_bias), dim3(256),
                       (size_t)cm.rows_per_iter * d->Cout * sizeof(float), st, cat, dcat, d->out_stride, d->out_coff, dpre,
                       bpart, (int64_t)4 * d->kd * M, d->Cout, cm.cq_n, cm.rows_per_iter);
  UNETK_LAUNCH_CHECK();
  rc = unetk_rows_reduce(bpart, 1, pl.nblk_bias, d->Cout, dbias ? dbias : bsink, btmp, st);
  if (rc != UNETK_OK) return rc;
  }

  const int64_t plane = (int64_t)4 * d->H * d->W * d->Cout;            // one dpre depth plane
  ImgAddr da;
  da.group = d->D; da.img_stride = d->kd * plane; da.group_stride = (int64_t)d->kd * d->D * plane;
  for (int a = 0; a < d->kd; ++a) {
    // 2. input gradient: [M x 4Cout] . [4Cout x Cin], accumulated over depth taps
    if (parts & 1) {
    PwParams p{};
    p.bf16 = d->precision;
    p.a = dpre + a * plane; p.wp = tap_panel(wp_dgrad, a, d); p.bias = nullptr; p.out = dx;
    p.M = M; p.K = 4 * d->Cout; p.Ncols = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
    p.oa = da; p.accumulate = a > 0 ? 1 : 0;
    rc = run_pw<1>(p, st);
    if (rc != UNETK_OK) return rc;
    }
    if (!(parts & 2)) continue;

    // 3. filter gradient of this depth tap
    DwParams q{};
    q.bf16 = d->precision == UNETK_BF16;
    q.x = x; q.dpre = dpre + a * plane; q.slab = slab; q.M = M; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.Cout = d->Cout;
    q.m_per_split = pl.m_per_split; q.n_co_tiles = (d->Cout + 63) / 64; q.n_ci_tiles = d->Cin / 64; q.da = da;
    const int grid = pl.S * 4 * q.n_co_tiles * q.n_ci_tiles;
    if (d->precision == UNETK_FP32) {
      UNETK_LAUNCH(deconv_wgrad4_kernel, dim3(pl.S * q.n_co_tiles * q.n_ci_tiles), dim3(256), 0, st, q);
      UNETK_LAUNCH_CHECK();
      rc = unetk_launch_slab_reduce(slab, pl.S, (int64_t)4 * d->Cin * d->Cout, dw + (int64_t)a * 4 * d->Cin * d->Cout, st);
      if (rc != UNETK_OK) return rc;
      continue;
    }
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute((const void*)deconv_wgrad_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 64 * (int)sizeof(float));
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)deconv_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * 128 * 64 * (int)sizeof(float));
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    if (dw_bf16s4(d)) {
      static bool attr4 = false;
      if (!attr4) {
        hipError_t e = hipFuncSetAttribute((const void*)deconv_wgrad_bf16s4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 128);
        if (e != hipSuccess) return (int)e;
        attr4 = true;
      }
      q.n_ci_tiles = d->Cin / 128;
      UNETK_LAUNCH(deconv_wgrad_bf16s4_kernel, dim3(pl.S * q.n_co_tiles * q.n_ci_tiles), dim3(512), (size_t)6 * 64 * 128, st, q);
    } else if (bs)
      UNETK_LAUNCH(deconv_wgrad_bf16s_kernel, dim3(grid), dim3(256), (size_t)2 * 128 * 128, st, q);
    else if (q.bf16)
      UNETK_LAUNCH(deconv_wgrad_kernel<true>, dim3(grid), dim3(256), (size_t)2 * 128 * 64 * sizeof(float), st, q);
    else
      UNETK_LAUNCH(deconv_wgrad_kernel<false>, dim3(grid), dim3(256), (size_t)2 * 128 * 64 * sizeof(float), st, q);
    UNETK_LAUNCH_CHECK();
    rc = unetk_launch_slab_reduce(slab, pl.S, (int64_t)4 * d->Cin * d->Cout, dw + (int64_t)a * 4 * d->Cin * d->Cout, st);
    if (rc != UNETK_OK) return rc;
  }
  return UNETK_OK;
}

// ---- 2-D entry points (UNet / GUNet): depth 1, one depth tap
extern "C" int unetk_deconv2x2_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_dgrad, void* stream) {
  return unetk_deconv3d_pack(w, 1, Cin, Cout, wp_fwd, wp_dgrad, stream);
}

extern "C" int unetk_deconv2x2_fwd(const unetk_deconv_desc* d, const void* x, const void* wp_fwd, const float* bias,
                                   void* out, void* stream) {
  UNETK_REQUIRE(d);
  const unetk_deconv3d_desc e = from2d(d);
  return unetk_deconv3d_fwd(&e, x, wp_fwd, bias, out, stream);
}

extern "C" size_t unetk_deconv2x2_bwd_ws_bytes(const unetk_deconv_desc* d) {
  if (!d) return 0;
  const unetk_deconv3d_desc e = from2d(d);
  return unetk_deconv3d_bwd_ws_bytes(&e);
}

extern "C" int unetk_deconv2x2_bwd_parts(const unetk_deconv_desc* d, const void* x, const void* wp_dgrad, const void* cat,
                                         const void* dcat, void* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                                         int parts, void* stream) {
  UNETK_REQUIRE(d && dbias);
  const unetk_deconv3d_desc e = from2d(d);
  return unetk_deconv3d_bwd_parts(&e, x, wp_dgrad, cat, dcat, dx, dw, dbias, ws, ws_bytes, parts, stream);
}

extern "C" int unetk_deconv2x2_bwd(const unetk_deconv_desc* d, const void* x, const void* wp_dgrad, const void* cat,
                                   const void* dcat, void* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                                   void* stream) {
  UNETK_REQUIRE(d && dbias);
  const unetk_deconv3d_desc e = from2d(d);
  return unetk_deconv3d_bwd(&e, x, wp_dgrad, cat, dcat, dx, dw, dbias, ws, ws_bytes, stream);
}
