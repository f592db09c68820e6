// Micro-benchmark: sustained fp32 MFMA ceiling on this device (v_mfma_f32_32x32x2_f32, back-to-back,
// 4 independent accumulators per wave, 1 and 2 waves per SIMD).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    const int grid = 256 * blocks_per_cu, iters = 200000;
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 1000, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 /*waves*/ * iters * 4.0 * 4096.0;
    printf("fp32 MFMA 32x32x2: %d block(s)/CU, %.1f ms, %.1f TFLOP/s\n", blocks_per_cu, ms, flop / ms / 1e9);
  }
  return 0;
}
