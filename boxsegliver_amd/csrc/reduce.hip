// Fixed-order row reductions shared by the normalisation / deconv / head kernels (fp64 accumulate).
#include "common.h"

namespace {

// ---------------------------------------------------------------- fixed-order row reductions
// level 1: src[k][rows][C] -> tmp[k][RB][C]; block = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void rows_reduce_l1_kernel(const float* __restrict__ src, int rows, int C,
                                                             int RB, float* __restrict__ tmp) {
  __shared__ double red[4][64];
  const int cblocks = (C + 63) / 64;
  int bid = blockIdx.x;
  const int cb = bid % cblocks; bid /= cblocks;
  const int rb = bid % RB;
  const int k = bid / RB;
  const int c = cb * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int chunk = (rows + RB - 1) / RB;
  const int r0 = rb * chunk, r1 = min(r0 + chunk, rows);
  double s = 0.0;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) s += (double)src[((int64_t)k * rows + r) * C + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C)
    tmp[((int64_t)k * RB + rb) * C + c] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// final: src[k][rows][C] -> dst[k][C]; block = 16 channels x 16 row lanes
__global__ __launch_bounds__(256) void rows_reduce_final_kernel(const float* __restrict__ src, int rows, int C,
                                                                float* __restrict__ dst) {
  __shared__ double red[16][17];
  const int cblocks = (C + 15) / 16;
  const int cb = blockIdx.x % cblocks, k = blockIdx.x / cblocks;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = cb * 16 + cl;
  double s = 0.0;
  if (c < C)
    for (int r = rl; r < rows; r += 16) s += (double)src[((int64_t)k * rows + r) * C + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][cl];
    dst[(int64_t)k * C + c] = (float)t;
  }
}

// final, wide: src[k][rows][C] -> dst[k][C] for up to UNETK_RR_WIDE_ROWS rows without a first level.  Block = 16 channels
// (four float4 lanes) x 64 row lanes: every row lane streams 64 contiguous bytes per row, sixteen independent loads per
// thread at 1024 rows (the 16 x 16 layout above reads 4 bytes per lane and 64 rows per thread: 15 us at 1024 rows).  Fixed
// order: row lane (strided rows) -> groups of four row lanes -> sixteen groups.  k == 0 / 1 may also be written to
// alias0 / alias1 (the norm backward's d beta / d gamma: saves its copy kernel).
__global__ __launch_bounds__(256) void rows_reduce_final_wide_kernel(const float* __restrict__ src, int rows, int C,
                                                                     float* __restrict__ dst, float* __restrict__ alias0,
                                                                     float* __restrict__ alias1) {
  __shared__ double red[64][17];
  __shared__ double red2[16][17];
  const int cblocks = (C + 15) / 16;
  const int cb = blockIdx.x % cblocks, k = blockIdx.x / cblocks;
  const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c0 = cb * 16 + q * 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c0 < C)       // C % 4 == 0
    for (int r = rl; r < rows; r += 64) {
      const float4 v = ldg4(src + ((int64_t)k * rows + r) * C + c0);
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
  red[rl][q * 4 + 0] = s0; red[rl][q * 4 + 1] = s1; red[rl][q * 4 + 2] = s2; red[rl][q * 4 + 3] = s3;
  __syncthreads();
  {
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    red2[j][cl] = (red[4 * j][cl] + red[4 * j + 1][cl]) + (red[4 * j + 2][cl] + red[4 * j + 3][cl]);
  }
  __syncthreads();
  const int c = cb * 16 + threadIdx.x;
  if (threadIdx.x < 16 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red2[j][threadIdx.x];
    dst[(int64_t)k * C + c] = (float)t;
    if (k == 0 && alias0) alias0[c] = (float)t;
    if (k == 1 && alias1) alias1[c] = (float)t;
  }
}

}  // namespace

size_t unetk_rows_reduce_tmp_floats(int K, int rows, int C) { return rows > UNETK_RR_DIRECT_ROWS ? (size_t)K * 64 * C : 0; }

// first level only: src[K][rows][C] -> tmp[K][64][C]
int unetk_rows_reduce_l1(const float* src, int K, int rows, int C, float* tmp, hipStream_t st) {
  const int RB = 64;
  const int cblocks = (C + 63) / 64;
  UNETK_LAUNCH(rows_reduce_l1_kernel, dim3(cblocks * RB * K), dim3(256), 0, st, src, rows, C, RB, tmp);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

int unetk_rows_reduce(const float* src, int K, int rows, int C, float* dst, float* tmp, hipStream_t st) {
  return unetk_rows_reduce_alias(src, K, rows, C, dst, tmp, nullptr, nullptr, st);
}

// alias0 / alias1: extra destinations of the k = 0 / k = 1 sums (may be null)
int unetk_rows_reduce_alias(const float* src, int K, int rows, int C, float* dst, float* tmp, float* alias0, float* alias1,
                            hipStream_t st) {
  const bool wide = C % 4 == 0 && unetk_aligned16(src);
  if (rows > (wide ? UNETK_RR_WIDE_ROWS : UNETK_RR_DIRECT_ROWS)) {
    const int rc = unetk_rows_reduce_l1(src, K, rows, C, tmp, st);
    if (rc != UNETK_OK) return rc;
    src = tmp;
    rows = 64;
  }
  const int cblocks = (C + 15) / 16;
  if (wide && unetk_aligned16(src)) {
    UNETK_LAUNCH(rows_reduce_final_wide_kernel, dim3(cblocks * K), dim3(256), 0, st, src, rows, C, dst, alias0, alias1);
  } else {
    UNETK_LAUNCH(rows_reduce_final_kernel, dim3(cblocks * K), dim3(256), 0, st, src, rows, C, dst);
    hipError_t e = hipSuccess;
    if (alias0) e = hipMemcpyAsync(alias0, dst, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && alias1 && K > 1)
      e = hipMemcpyAsync(alias1, dst + C, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
