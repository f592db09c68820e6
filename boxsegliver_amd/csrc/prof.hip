// Per-launch kernel timing inside the library (bench.py's `roofline` / `kernels` blocks).
//
// Every kernel of libunetk is launched through UNETK_LAUNCH (common.h).  While the trace is on, the launch goes through
// hipExtLaunchKernelGGL with a start and a stop event from the pool below: both events are bound to the DISPATCH itself, so
// hipEventElapsedTime(start, stop) is the kernel's own begin -> end interval on the GPU clock -- the interval rocprofv3
// --kernel-trace reports -- whatever the host was doing around the launch (tools/probe_ext_events.hip: a 1.000 ms kernel
// reads 1.001 ms behind a 3 ms host stall on an idle stream, where a hipEventRecord bracket reads 4.07-4.4 ms).  Nothing
// here synchronises; durations are read after the caller has synchronised the stream.
//
// Diagnostics only: with the trace off (the default) a launch costs one predicted-not-taken branch and the library keeps its
// "no global mutable state" contract.  The trace is process-wide and serialised by a mutex (autograd's backward thread and
// the main thread both launch, never concurrently on one stream).
#include <cxxabi.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

int g_unetk_prof_on = 0;

namespace {
struct Rec {
  const void* fn;
  hipEvent_t a, b;
};
std::mutex g_mu;
std::vector<Rec> g_recs;                 // launches since the last reset, in launch order
std::vector<hipEvent_t> g_pool;          // events created so far (two per record), reused after a reset
size_t g_pool_used = 0;

hipEvent_t pool_get() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}
}  // namespace

void unetk_prof_pair(const void* fn, hipEvent_t* a, hipEvent_t* b) {
  std::lock_guard<std::mutex> lk(g_mu);
  *a = pool_get();
  *b = pool_get();
  g_recs.push_back(Rec{fn, *a, *b});
}

extern "C" int unetk_prof_reset(int reserve_launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.clear();
  g_pool_used = 0;
  if (reserve_launches > 0) {
    g_recs.reserve((size_t)reserve_launches);
    while (g_pool.size() < 2 * (size_t)reserve_launches) {      // event creation is host work: do it outside the timed region
      hipEvent_t e = nullptr;
      hipError_t err = hipEventCreate(&e);
      if (err != hipSuccess) return (int)err;
      g_pool.push_back(e);
    }
  }
  return UNETK_OK;
}

extern "C" int unetk_prof_enable(int on) {
  g_unetk_prof_on = on ? 1 : 0;
  return UNETK_OK;
}

extern "C" int unetk_prof_mark(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_recs.size();
}

extern "C" int unetk_prof_read(int first, int count, float* ms_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (first < 0 || count < 0 || (size_t)first + (size_t)count > g_recs.size() || (count > 0 && ms_out == nullptr)) return UNETK_E_BADARG;
  for (int i = 0; i < count; ++i) {
    const Rec& r = g_recs[(size_t)first + i];
    if (r.a == nullptr || r.b == nullptr) return UNETK_E_BADARG;
    hipError_t e = hipEventElapsedTime(&ms_out[i], r.a, r.b);     // hipErrorNotReady if the caller did not synchronise
    if (e != hipSuccess) return (int)e;
  }
  return UNETK_OK;
}

extern "C" int unetk_prof_name(int i, char* buf, int cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (i < 0 || (size_t)i >= g_recs.size() || buf == nullptr || cap < 2) return UNETK_E_BADARG;
  const char* mangled = hipKernelNameRefByPtr(g_recs[(size_t)i].fn, nullptr);
  if (mangled == nullptr) mangled = "?";
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  const char* s = (status == 0 && dem != nullptr) ? dem : mangled;
  strncpy(buf, s, (size_t)cap - 1);
  buf[cap - 1] = 0;
  free(dem);
  return UNETK_OK;
}
