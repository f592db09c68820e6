// Probe for the "128 x 128 accumulator tile per wave, ONE wave per SIMD" bf16 conv loop that DESIGN.md 4.1 / the round-3 review
// name as the structural alternative to the shipped kernel (csrc/conv_igemm_bf16s.hip: 512-thread blocks, 2 waves per SIMD,
// 64 px x 128 co = 4 x 8 fragments of v_mfma_f32_16x16x32_bf16 per wave, 12 ds_read_b128 per 32 MFMAs).
// Same block tile (512 px x 128 co x one (tap, 32-channel chunk) per step), same LDS image (halo [34 rows][20-px pitch][64 B],
// ring of six 8 KiB panels), same staging volume per block and step (16 direct-to-LDS requests of 1 KiB), one barrier per step:
//   W8: 8 waves, wave tile  64 x 128 (4 x 8 fragments, 128 accumulator registers), 12 LDS reads per 32 MFMAs
//   W4: 4 waves, wave tile 128 x 128 (8 x 8 fragments, 256 accumulator registers), 16 LDS reads per 64 MFMAs
// Both software-pipelined the same way (the reads of half-step g + 1 are requested before the MFMAs of half-step g).  Timing
// probe, not a convolution: staged bytes overwrite operands at random.  Flags: 'z' = zeros (no DVFS give-back).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_tile_bf16.hip -o tools/mfma_tile_bf16.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 20 * 64, SLOT_B = 8192, HALO_B = 43 * 1024, RING = 6, LDS_B = 2 * HALO_B + RING * SLOT_B + 1024;

#define DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// MFMAs as inline asm accumulating IN PLACE (through the builtin the register allocator rotates the accumulators: 1228
// v_accvgpr moves for 576 MFMAs in the 128 x 128 variant); AG = accumulators in AGPRs (the one-wave-per-SIMD variant: 256 of them)
template <bool AG>
__device__ __forceinline__ void mfma_ip(f32x4& c, const u32x4& a, const u32x4& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int N>
__device__ __forceinline__ void tie(u32x4* S) {      // makes the consuming MFMAs depend on the wait that precedes this
  if constexpr (N == 4) asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]));
  if constexpr (N == 8) asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(S[6]), "+v"(S[7]));
}

// NW = waves per block; TMF = A fragments (16-pixel rows) per wave; STG: direct-to-LDS requests on; BAR: barrier per step
template <int NW, int TMF, bool STG, bool BAR>
__global__ __launch_bounds__(NW * 64, 1) void tile_probe(const uint4* __restrict__ src, float* out, int nsteps9) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  for (int i = tid; i < LDS_B / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t abase[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = l15 + kw;
    const int s = ((col >> 2) & 1) ? 0 : 3;
    abase[kw] = lds0 + wave * TMF * ROWB + col * 64 + ((kq ^ s) << 4);
  }
  const uint32_t bbase0 = lds0 + 2 * HALO_B + kq * 2048 + l15 * 16;
  f32x4 acc[TMF][8];
#pragma unroll
  for (int i = 0; i < TMF; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 A[TMF], B0[4], B1[4];
  const unsigned gi = blockIdx.x * 4096u + tid;
  uint32_t hoff = 0, slot = 0;
  unsigned step = 0;
  // prologue: A fragments and the first four B fragments of step 0
#pragma unroll
  for (int i = 0; i < TMF; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(A[i]) : "v"(abase[0] + i * ROWB));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(B0[j]) : "v"(bbase0 + j * 256));
  for (int c = 0; c < nsteps9; ++c) {
#pragma unroll
    for (int T = 0; T < 9; ++T) {
      if constexpr (STG) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NW == 8 ? 7 : 14) : "memory");
      if constexpr (BAR) __builtin_amdgcn_s_barrier();
      if constexpr (STG) {
        // 16 KiB per block and step: 8 panel pieces + 8 halo pieces of 1 KiB (the shipped kernel's volume), dealt over the waves
        const uint32_t pslot = (slot + RING - 1 >= RING ? slot - 1 : slot + RING - 1) * SLOT_B;
#pragma unroll
        for (int j = 0; j < 8 / NW; ++j) {
          const uint4* g = src + (((step * 512u) & 0xffff) + (wave * (8 / NW) + j) * 64 + lane);
          char* dst = smem + 2 * HALO_B + pslot + (wave * (8 / NW) + j) * 1024;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
          const uint4* g2 = src + ((gi * 16u + step * 65536u + j * 4096u + (lane >> 2) * 16u + (lane & 3)) & 0xfffff);
          char* dst2 = smem + (hoff ? 0 : HALO_B) + ((T * 8 + wave * (8 / NW) + j) % (HALO_B / 1024)) * 1024;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g2, (__attribute__((address_space(3))) void*)dst2, 16, 0, 0);
        }
      }
      const uint32_t bslot = slot * SLOT_B;
      const uint32_t nslot = (slot + 1 == RING ? 0 : slot + 1);
      // half-step 1's B fragments are requested, then half-step 0 runs (A + B0 were requested a half-step ago)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(B1[j]) : "v"(bbase0 + bslot + (4 + j) * 256));
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      tie<TMF>(A); tie<4>(B0);
#pragma unroll
      for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          mfma_ip<(TMF > 4)>(acc[i][j], A[i], B0[j]);
      // next step's B0 (its panel became visible at this step's barrier); the A fragments are still needed by half-step 1
      const int NT9 = (T + 1) % 9;
      const uint32_t nh = (T == 8) ? (hoff ? 0u : (uint32_t)HALO_B) : hoff;
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(B0[j]) : "v"(bbase0 + nslot * SLOT_B + j * 256));
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      tie<4>(B1);
#pragma unroll
      for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          mfma_ip<(TMF > 4)>(acc[i][4 + j], A[i], B1[j]);
      // next step's A fragments: refilled behind the last MFMAs that read them (the shipped kernel refills row by row)
#pragma unroll
      for (int i = 0; i < TMF; ++i)
        asm volatile("ds_read_b128 %0, %1" : "=v"(A[i]) : "v"(abase[NT9 % 3] + nh + (i + NT9 / 3) * ROWB));
      slot = nslot;
      ++step;
    }
    hoff = hoff ? 0u : (uint32_t)HALO_B;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < TMF; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  sum += __uint_as_float(A[0].x & 0xff) + __uint_as_float(B0[0].x & 0xff) + __uint_as_float(B1[0].x & 0xff);
  out[blockIdx.x * NT + tid] = sum;
}

template <int NW, int TMF, bool STG, bool BAR>
void run(const uint4* src, float* out, const char* what) {
  auto k = tile_probe<NW, TMF, STG, BAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  const int grid = 256, n9 = 240;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), LDS_B, 0, src, out, 20);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), LDS_B, 0, src, out, n9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)grid * n9 * 9 * 512.0 * 128.0 * 32.0 * 2.0;     // a step = 512 px x 128 co x 32 channels per block
  printf("W%d tile %3dx128 %s%s %-44s %7.3f ms %7.1f TFLOP/s (%4.1f %% of 2516.6)\n", NW, TMF * 16, STG ? "S" : "-", BAR ? "B" : "-", what, best,
         flop / best / 1e9, flop / best / 1e9 / 25.166);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("  HIP error %d\n", (int)e);
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && argv[1][0] == 'z';
  uint4* src; float* out;
  const size_t n = 1 << 20;
  uint32_t* hs = (uint32_t*)malloc(n * 16);
  srand(1);
  for (size_t i = 0; i < n * 4; ++i) {
    const uint32_t lo = 0x3f00u | (rand() & 0x80ffu), hi = 0x3f00u | (rand() & 0x80ffu);
    hs[i] = zeros ? 0u : (lo | (hi << 16));
  }
  hipMalloc(&src, n * 16 + 65536); hipMalloc(&out, 1 << 22);
  hipMemcpy(src, hs, n * 16, hipMemcpyHostToDevice);
  printf("data: %s\n", zeros ? "zeros" : "random bf16");
  run<8, 4, false, false>(src, out, "2 waves/SIMD, LDS reads only");
  run<4, 8, false, false>(src, out, "1 wave/SIMD,  LDS reads only");
  run<8, 4, false, true>(src, out, "2 waves/SIMD, + barrier per step");
  run<4, 8, false, true>(src, out, "1 wave/SIMD,  + barrier per step");
  run<8, 4, true, true>(src, out, "2 waves/SIMD, + 16 KiB staged per step");
  run<4, 8, true, true>(src, out, "1 wave/SIMD,  + 16 KiB staged per step");
  return 0;
}
