// Micro-benchmark: what the fp32 MFMA loop of the conv kernels loses to its surroundings.  The same 2 x 2 fragment loop as
// conv_igemm.hip (32 MFMAs per step, operands from four ds_read_b128 per 16 MFMAs), with the features of the real kernel
// switched on one at a time:  V0 register-only operands, V1 + LDS operand reads, V2 + one barrier and two ds_write_b128 per
// step, V3 + five global prefetch loads per step written to LDS at its end, V4 filter fragments straight from global memory
// to registers (barrier + LDS writes every ninth step), V5 = V3 + s_setprio, V6 = V3 with a barrier every second step,
// V7 = V3 through direct-to-LDS loads.  1-3 blocks (4 waves) per CU.  Output of one run: profiles/r02_mfma_mix.txt.
// V3..V7 overwrite their own operands (timing probes, not a correct GEMM); every global index is masked into the source.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void mix(const float* __restrict__ src, float* out, int steps, int en) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 8192; i += 256) smem[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int abase = ((wave >> 1) * 64 + l31) * 20 + 4 * h;      // 20-float pixel stride as in the conv kernel
  const int bbase = 4096 + (h * 128 + (wave & 1) * 64 + l31) * 4;
  float4 ra = *reinterpret_cast<const float4*>(&smem[abase]), rb = *reinterpret_cast<const float4*>(&smem[bbase]);
  const unsigned gi = (blockIdx.x * 256u + tid) * 4u;       // every global index is masked into the 1 Mi-float source
  float4 pre0 = make_float4(0, 0, 0, 0), pre1 = pre0;
  float4 bc[4], bn[4];
  if (V == 4)
    for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const float4*>(src + ((gi + j * 1024u) & 0xffffcu));
  for (int s = 0; s < steps; ++s) {
    if (V == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bn[j] = *reinterpret_cast<const float4*>(src + ((gi + (unsigned)(s & 63) * 16384u + j * 1024u) & 0xffffcu));
      if (s % 9 == 5) pre0 = *reinterpret_cast<const float4*>(src + ((gi + (unsigned)(s & 63) * 16384u + 8192u) & 0xffffcu));
    }
    float4 pf[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) pf[j] = make_float4(0, 0, 0, 0);
    // the run-time predicate (always true) keeps the loads in their own basic block at the top of the step, as the halo
    // predicates do in the real kernel; straight-line code lets the compiler sink them to their use at the end of it
    if (V == 7 && tid < en) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float* g = src + ((gi + (unsigned)(s & 63) * 16384u + j * 1024u) & 0xffffcu);
        // wave-uniform LDS target; lanes land at dst + lane * 16 B.  Issued as inline asm: through the builtin the compiler
        // knows it is an LDS write and puts an s_waitcnt vmcnt(0) before every operand ds_read (it cannot tell the buffers apart)
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + 3072 + j * 1024 + wave * 256);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g) : "memory", "m0");
      }
    }
    if ((V == 3 || V == 5 || V == 6) && tid < en) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
        pf[j] = *reinterpret_cast<const float4*>(src + ((gi + (unsigned)(s & 63) * 16384u + j * 1024u) & 0xffffcu));
    }
    const float* hb = smem + (V >= 2 && V != 4 ? (s & 1) * 16 : 0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float4 a[2], b[2];
      if (V >= 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const float4*>(&hb[abase + t * 640 + 8 * g]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          b[t] = V == 4 ? bc[g * 2 + t] : *reinterpret_cast<const float4*>(&hb[bbase + (2 * g * 128 + t * 32) * 4]);
      } else {
        a[0] = ra; a[1] = rb; b[0] = rb; b[1] = ra;
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
        }
    }
    if (V == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bc[j] = bn[j];
      if (s % 9 == 8) {
        float4 w0 = *reinterpret_cast<const float4*>(&smem[6144 + tid * 4]);
        w0.x += pre0.x * 0.f;
        *reinterpret_cast<float4*>(&smem[6144 + tid * 4]) = w0;
        *reinterpret_cast<float4*>(&smem[7168 + tid * 4]) = w0;
        *reinterpret_cast<float4*>(&smem[5120 + tid * 4]) = w0;
        __syncthreads();
      }
    } else if (V == 2) {
      float4 w0 = *reinterpret_cast<const float4*>(&smem[6144 + tid * 4]);
      *reinterpret_cast<float4*>(&smem[6144 + tid * 4]) = w0;
      *reinterpret_cast<float4*>(&smem[7168 + tid * 4]) = w0;
      __syncthreads();
    } else if (V == 3 || V == 5) {
      if (V == 5) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int j = 0; j < 5; ++j) *reinterpret_cast<float4*>(&smem[(3072 + j * 1024 + tid * 4) & 8191]) = pf[j];
      __syncthreads();
      if (V == 5) __builtin_amdgcn_s_setprio(1);
    } else if (V == 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else if (V == 6) {       // barrier every second step
#pragma unroll
      for (int j = 0; j < 5; ++j) *reinterpret_cast<float4*>(&smem[(3072 + j * 1024 + tid * 4) & 8191]) = pf[j];
      if (s & 1) __syncthreads();
    }
  }
  float sum = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int V>
void run(const float* src, float* out, int bpc) {
  // LDS per block chosen so that exactly `bpc` blocks fit a CU (160 KB)
  const int lds = (bpc == 1 ? 96 * 1024 : bpc == 2 ? 64 * 1024 : bpc == 3 ? 45 * 1024 : 36 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256 * bpc, steps = 6000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mix<V>, dim3(grid), dim3(256), lds, 0, src, out, 200, 256);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mix<V>, dim3(grid), dim3(256), lds, 0, src, out, steps, 256);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * steps * 32.0 * 4096.0;
  printf("V%d  %d block(s)/CU  %.2f ms  %.1f TFLOP/s\n", V, bpc, ms, flop / ms / 1e9);
}

int main() {
  float *src, *out;
  const size_t n = 1 << 20;
  float* hsrc = (float*)malloc(n * 4);
  srand(1);
  for (size_t i = 0; i < n; ++i) hsrc[i] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
  hipMalloc(&src, n * 4 + 4096); hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipMemcpy(src, hsrc, n * 4, hipMemcpyHostToDevice);
  for (int bpc = 1; bpc <= 3; ++bpc) {
    run<0>(src, out, bpc); run<1>(src, out, bpc); run<2>(src, out, bpc); run<3>(src, out, bpc); run<4>(src, out, bpc); run<5>(src, out, bpc); run<6>(src, out, bpc); run<7>(src, out, bpc);
  }
  return 0;
}
