// What do the start / stop events of hipExtLaunchKernelGGL measure on this ROCm?  A kernel spins for a known time on the
// 100 MHz wall clock and writes its own first / last tick; it is timed (a) by hipEventRecord brackets with the host asleep
// between the first record and the launch (an idle stream: the bracket swallows the host gap), (b) by the same brackets
// behind a long kernel (a deep queue), (c) by the start / stop events of hipExtLaunchKernelGGL in both situations.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_ext_events.hip -o tools/probe_ext_events.bin && tools/probe_ext_events.bin
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(long long ticks, long long* out) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) t = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = t; }
}

int main() {
  long long* d;
  CK(hipMalloc(&d, 64));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1, s0, s1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1));
  const long long ms1 = 100000;     // 1 ms of the 100 MHz clock
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    // (a) idle stream, host sleeps 3 ms between the record and the launch
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    usleep(3000);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, ms1, d);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("record brackets, idle stream, host asleep 3 ms inside : %.3f ms (kernel spins 1.000)\n", ms);
    // (b) deep queue: a 5 ms kernel in front, same sleep
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 5 * ms1, d + 2);
    CK(hipEventRecord(e0, st));
    usleep(3000);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, ms1, d);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("record brackets, 5 ms kernel in front, host asleep 3 ms: %.3f ms\n", ms);
    // (c) ext launch events, idle stream
    CK(hipStreamSynchronize(st));
    usleep(3000);
    hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, s0, s1, 0, ms1, d);
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, s0, s1));
    printf("ext-launch start/stop, idle stream                     : %.3f ms\n", ms);
    // (d) ext launch events behind a 5 ms kernel
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 5 * ms1, d + 2);
    hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, s0, s1, 0, ms1, d);
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, s0, s1));
    printf("ext-launch start/stop, 5 ms kernel in front            : %.3f ms\n", ms);
    // (e) stop event only, measured from a recorded event in front of the long kernel: 5 + 1 expected
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 5 * ms1, d + 2);
    hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, nullptr, s1, 0, ms1, d);
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, s1));
    printf("record .. 5 ms kernel .. ext stop event                : %.3f ms (6.0 expected)\n", ms);
  }
  // cost on the host: 2000 launches of an empty-ish kernel with and without the events
  hipEvent_t ev[4000];
  for (int i = 0; i < 4000; ++i) CK(hipEventCreate(&ev[i]));
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipStreamSynchronize(st));
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < 2000; ++i) {
      if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 0ll, d);
      else if (mode == 1) hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, ev[2 * i], ev[2 * i + 1], 0, 0ll, d);
      else { CK(hipEventRecord(ev[2 * i], st)); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 0ll, d); CK(hipEventRecord(ev[2 * i + 1], st)); }
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    CK(hipStreamSynchronize(st));
    struct timespec c;
    clock_gettime(CLOCK_MONOTONIC, &c);
    printf("%s: %.2f us host per launch, %.2f us per launch end to end\n", mode == 0 ? "plain launch      " : mode == 1 ? "ext start/stop    " : "record + launch + record",
           ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) / 2000 / 1e3, ((c.tv_sec - a.tv_sec) * 1e9 + (c.tv_nsec - a.tv_nsec)) / 2000 / 1e3);
  }
  return 0;
}
