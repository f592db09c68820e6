// `boundary` loss-weight map (loss_metrics.py:149-165), entirely on the device.
//
// The reference builds it per step through a HOST round trip: one-hot -> 3x3 dilation ring per class -> tf.py_func
// (scipy.ndimage.distance_transform_edt) -> exp(-d/25) + 1 -> per-sample normalisation to mean 1.  Here:
//   ring(p)   = some in-bounds 3x3 neighbour carries a label different from p's   (== sum_c (dilate(onehot_c) - onehot_c) > 0)
//   g(h, w)   = row distance from (h, w) to the nearest ring pixel of column w     (two scans per column)
//   d2(h, w)  = min_w' (w - w')^2 + g(h, w')^2                                     (exact squared Euclidean distance)
//   d         = (float) sqrt((double) d2)      -- what scipy returns after .astype(float32): bit-exact
//   w         = exp(-d / 25) + 1, then w * H*W / sum_hw(w) with a fixed-order sum (bit-reproducible)
// An image without any ring pixel (a single label) has no zero for the EDT to measure to; scipy (1.15) then returns
// sqrt((h+1)^2 + w^2), which is reproduced so that such slices weigh as they do in the reference.
// HBM-bound and tiny next to the network (one int32 read + a few float passes over [N,H,W]).
#include "common.h"

namespace {

constexpr int G_INF = 1 << 20;

// thread = (n, w) column: ring detection + downward / upward scans
__global__ __launch_bounds__(256) void edt_columns_kernel(const int32_t* __restrict__ lab, int N, int H, int W,
                                                          int32_t* __restrict__ g, int32_t* __restrict__ has_ring) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * W) return;
  const int n = i / W, w = i - n * W;
  const int32_t* L = lab + (int64_t)n * H * W;
  int32_t* G = g + (int64_t)n * H * W;
  int last = -G_INF;
  bool any = false;
  for (int h = 0; h < H; ++h) {
    const int32_t own = L[(int64_t)h * W + w];
    bool ring = false;
    for (int dh = -1; dh <= 1; ++dh) {
      const int hh = h + dh;
      if (hh < 0 || hh >= H) continue;
      for (int dw = -1; dw <= 1; ++dw) {
        const int ww = w + dw;
        if (ww < 0 || ww >= W) continue;
        ring |= L[(int64_t)hh * W + ww] != own;
      }
    }
    if (ring) { last = h; any = true; }
    G[(int64_t)h * W + w] = min(h - last, G_INF);
  }
  last = G_INF * 2;
  for (int h = H - 1; h >= 0; --h) {
    const int64_t o = (int64_t)h * W + w;
    if (G[o] == 0) last = h;
    G[o] = min(G[o], min(last - h, G_INF));
  }
  if (any) atomicOr(&has_ring[n], 1);
}

// block = one image row (n, h); the row of g sits in LDS; thread = output column(s)
__global__ __launch_bounds__(256) void edt_rows_kernel(const int32_t* __restrict__ g, const int32_t* __restrict__ has_ring,
                                                       int N, int H, int W, float* __restrict__ wmap,
                                                       float* __restrict__ row_sums) {
  extern __shared__ int32_t grow[];
  __shared__ float red[256];
  const int row = blockIdx.x;
  const int n = row / H, h = row - n * H;
  const int32_t* G = g + (int64_t)row * W;
  for (int w = threadIdx.x; w < W; w += 256) grow[w] = G[w];
  __syncthreads();
  const bool ring = has_ring[n] != 0;
  float part = 0.f;
  for (int w = threadIdx.x; w < W; w += 256) {
    int64_t d2;
    if (ring) {
      d2 = (int64_t)1 << 60;
      for (int k = 0; k < W; ++k) {
        const int64_t gv = grow[k];
        if (gv >= G_INF) continue;
        const int64_t dx = w - k;
        d2 = min(d2, dx * dx + gv * gv);
      }
    } else {
      d2 = (int64_t)(h + 1) * (h + 1) + (int64_t)w * w;
    }
    const float d = (float)sqrt((double)d2);
    const float v = expf(-d / 25.f) + 1.f;
    wmap[(int64_t)row * W + w] = v;
    part += v;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) row_sums[row] = red[0];
}

__global__ __launch_bounds__(256) void edt_normalise_kernel(float* __restrict__ wmap, const float* __restrict__ row_sums,
                                                            int N, int H, int W) {
  const int n = blockIdx.y;
  double s = 0.0;
  for (int h = 0; h < H; ++h) s += (double)row_sums[n * H + h];          // fixed order, every thread the same value
  const float scale = (float)((double)H * (double)W / s);
  float* M = wmap + (int64_t)n * H * W;
  const int64_t total = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    M[i] *= scale;
}

}  // namespace

extern "C" size_t unetk_boundary_weights_ws_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return ((size_t)N * H * W + (size_t)N * H + (size_t)N + 64) * 4;
}

extern "C" int unetk_boundary_weights(const int32_t* labels, int N, int H, int W, float* wmap, void* ws, size_t ws_bytes,
                                      void* stream) {
  UNETK_REQUIRE(labels && wmap && ws && N > 0 && H > 0 && W > 0);
  if (W > 12 * 1024 || H >= G_INF) return UNETK_E_UNSUPPORTED;
  if (ws_bytes < unetk_boundary_weights_ws_bytes(N, H, W)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int32_t* g = (int32_t*)ws;
  float* row_sums = (float*)(g + (size_t)N * H * W);
  int32_t* has_ring = (int32_t*)(row_sums + (size_t)N * H);
  hipError_t e = hipMemsetAsync(has_ring, 0, (size_t)N * 4, st);
  if (e != hipSuccess) return (int)e;
  UNETK_LAUNCH(edt_columns_kernel, dim3((N * W + 255) / 256), dim3(256), 0, st, labels, N, H, W, g, has_ring);
  UNETK_LAUNCH_CHECK();
  UNETK_LAUNCH(edt_rows_kernel, dim3(N * H), dim3(256), (size_t)W * 4, st, g, has_ring, N, H, W, wmap, row_sums);
  UNETK_LAUNCH_CHECK();
  const int gx = (int)min((int64_t)1024, ((int64_t)H * W + 255) / 256);
  UNETK_LAUNCH(edt_normalise_kernel, dim3(gx, N), dim3(256), 0, st, wmap, row_sums, N, H, W);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
