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

}  // namespace

size_t unetk_rows_reduce_tmp_floats(int K, int rows, int C) { return rows > UNETK_RR_DIRECT_ROWS ? (size_t)K * 64 * C : 0; }

// first level only: src[K][rows][C] -> tmp[K][64][C]
int unetk_rows_reduce_l1(const float* src, int K, int rows, int C, float* tmp, hipStream_t st) {
  const int RB = 64;
  const int cblocks = (C + 63) / 64;
  hipLaunchKernelGGL(rows_reduce_l1_kernel, dim3(cblocks * RB * K), dim3(256), 0, st, src, rows, C, RB, tmp);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

int unetk_rows_reduce(const float* src, int K, int rows, int C, float* dst, float* tmp, hipStream_t st) {
  if (rows > UNETK_RR_DIRECT_ROWS) {
    const int rc = unetk_rows_reduce_l1(src, K, rows, C, tmp, st);
    if (rc != UNETK_OK) return rc;
    src = tmp;
    rows = 64;
  }
  const int cblocks = (C + 15) / 16;
  hipLaunchKernelGGL(rows_reduce_final_kernel, dim3(cblocks * K), dim3(256), 0, st, src, rows, C, dst);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

