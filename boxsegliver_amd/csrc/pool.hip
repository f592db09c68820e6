// 2x2 pooling (HBM-bound, float4 NHWC).  slim.max_pool2d(x, [2,2]) at NetworksV2/UNet.py:81 (VALID; GUNet
// sets SAME, identical for even sizes, GUNet.py:249-250) and slim.avg_pool2d(gs, 2) of GUNet's spatial-guide
// pyramid (GUNet.py:155-158).
#include "common.h"

namespace {

// ---------------------------------------------------------------- max pool 2x2 s2 VALID
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, int xs, T* __restrict__ p,
                                                           int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, cq_n = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * cq_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    int64_t r = i / cq_n;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const T* b = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * xs + cq * 4;
    const float4 a0 = ld4(b), a1 = ld4(b + xs), a2 = ld4(b + (int64_t)W * xs), a3 = ld4(b + (int64_t)W * xs + xs);
    float4 o;
    o.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
    o.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
    o.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z));
    o.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
    st4(p + i * 4, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ x, int xs, const T* __restrict__ p,
                                                           const T* __restrict__ dp, const T* __restrict__ add,
                                                           int as, T* __restrict__ dx, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, cq_n = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * cq_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    int64_t r = i / cq_n;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int64_t pix = ((int64_t)n * H + 2 * ho) * W + 2 * wo;
    const T* b = x + pix * xs + cq * 4;
    const float4 a0 = ld4(b), a1 = ld4(b + xs), a2 = ld4(b + (int64_t)W * xs);  // 4th is implied
    const float4 m = ld4(p + i * 4), g = ld4(dp + i * 4);
    float4 o0, o1, o2, o3;
    // first maximum in window scan order gets the gradient (TF MaxPoolGrad)
#define MPB(f)                                              \
  {                                                         \
    const bool e0 = a0.f == m.f;                            \
    const bool e1 = !e0 && a1.f == m.f;                     \
    const bool e2 = !e0 && !e1 && a2.f == m.f;              \
    const bool e3 = !e0 && !e1 && !e2;                      \
    o0.f = e0 ? g.f : 0.f;                                  \
    o1.f = e1 ? g.f : 0.f;                                  \
    o2.f = e2 ? g.f : 0.f;                                  \
    o3.f = e3 ? g.f : 0.f;                                  \
  }
    MPB(x) MPB(y) MPB(z) MPB(w)
#undef MPB
    if (add) {   // the skip connection's gradient (the other consumer of x), summed here instead of in a separate pass
      const T* s = add + pix * as + cq * 4;
      const float4 s0 = ld4(s), s1 = ld4(s + as), s2 = ld4(s + (int64_t)W * as), s3 = ld4(s + (int64_t)W * as + as);
      o0.x += s0.x; o0.y += s0.y; o0.z += s0.z; o0.w += s0.w;
      o1.x += s1.x; o1.y += s1.y; o1.z += s1.z; o1.w += s1.w;
      o2.x += s2.x; o2.y += s2.y; o2.z += s2.z; o2.w += s2.w;
      o3.x += s3.x; o3.y += s3.y; o3.z += s3.z; o3.w += s3.w;
    }
    T* d = dx + pix * C + cq * 4;
    st4(d, o0);
    st4(d + C, o1);
    st4(d + (int64_t)W * C, o2);
    st4(d + (int64_t)W * C + C, o3);
  }
}

inline int ew_grid(int64_t total_threads) {
  int64_t g = (total_threads + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}


// guide pyramid: few channels (g = 1..2), not float4-able -> scalar kernel; SAME == VALID for even sizes
__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ p, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float* b = x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c;
    p[i] = 0.25f * (b[0] + b[C] + b[(int64_t)W * C] + b[(int64_t)W * C + C]);
  }
}

// First and second moments of the (pooled) spatial guide, per statistics group: out[grp][0..G) = mean of channel g,
// out[grp][G + i*G + j] = mean of g_i * g_j.  GUNet --fix (GUNet.py:299-304) normalises the guide's 1x1 conv; because that
// conv is linear in the guide, its batch / instance statistics follow exactly from these G + G^2 numbers and the conv
// never has to be materialised.  One block per group, fp64 accumulation, fixed order.
__global__ __launch_bounds__(256) void guide_moments_kernel(const float* __restrict__ guide, int64_t P, int G,
                                                            float* __restrict__ out) {
  __shared__ double red[256];
  const int grp = blockIdx.x;
  const float* gp = guide + (int64_t)grp * P * G;
  const int nq = G + G * G;
  for (int q = 0; q < nq; ++q) {
    const int i = q < G ? q : (q - G) / G, j = q < G ? -1 : (q - G) % G;
    double s = 0.0;
    for (int64_t p = threadIdx.x; p < P; p += 256) {
      const double a = gp[p * G + i];
      s += j < 0 ? a : a * (double)gp[p * G + j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[(int64_t)grp * nq + q] = (float)(red[0] / (double)P);
    __syncthreads();
  }
}

// tf.image.image_gradients + concat (UNet.py:69-71): out[..., 0:C] = x, [C:2C] = x[h+1] - x[h] (last row 0),
// [2C:3C] = x[w+1] - x[w] (last column 0).  Few channels (C = 1..3) -> scalar, HBM-bound on the 3C write.
__global__ void image_gradients_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const float v = x[i];
    float* o = out + pix * 3 * C + c;
    o[0] = v;
    o[C] = h + 1 < H ? x[i + (int64_t)W * C] - v : 0.f;
    o[2 * C] = w + 1 < W ? x[i + C] - v : 0.f;
  }
}

// tf.image.sobel_edges of ONE channel + concat (InterUNet.py:105-109): out[..., 0:C] = x, out[..., C] = dy, out[..., C+1] = dx
// with the 3x3 Sobel kernels [[-1,-2,-1],[0,0,0],[1,2,1]] (dy) and its transpose (dx) as cross-correlations over the
// REFLECT-padded image (index -1 -> 1, H -> H-2).
__global__ void sobel_concat_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C, int ch) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int64_t img = (pix / ((int64_t)W * H)) * H * W;
    float v[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        int hh = h + a - 1, ww = w + b - 1;
        hh = hh < 0 ? -hh : (hh >= H ? 2 * H - 2 - hh : hh);
        ww = ww < 0 ? -ww : (ww >= W ? 2 * W - 2 - ww : ww);
        v[a][b] = x[(img + (int64_t)hh * W + ww) * C + ch];
      }
    float* o = out + pix * (C + 2);
    for (int c = 0; c < C; ++c) o[c] = x[pix * C + c];
    o[C] = (v[2][0] + 2.f * v[2][1] + v[2][2]) - (v[0][0] + 2.f * v[0][1] + v[0][2]);
    o[C + 1] = (v[0][2] + 2.f * v[1][2] + v[2][2]) - (v[0][0] + 2.f * v[1][0] + v[2][0]);
  }
}

}  // namespace

extern "C" int unetk_sobel_concat(const float* x, float* out, int N, int H, int W, int C, int ch, void* stream) {
  UNETK_REQUIRE(x && out && N > 0 && H > 1 && W > 1 && C > 0 && ch >= 0 && ch < C);
  UNETK_LAUNCH(sobel_concat_kernel, dim3(ew_grid((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, x, out, N, H,
                     W, C, ch);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// out[n,h,w,:] (+)= scale * x[n, fh ? H-1-h : h, fw ? W-1-w : w, :] -- mirror test-time augmentation on the device:
// flips the slab fed to the net and un-flips + accumulates the class probabilities (np.flip(...) / mirror_div at
// evaluators/evaluator_liver.py:648-655).
__global__ void flip_axpy_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C, int fh,
                                 int fw, float scale, int accumulate) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    const int sh = fh ? H - 1 - h : h, sw = fw ? W - 1 - w : w;
    const float v = scale * x[(((int64_t)n * H + sh) * W + sw) * C + c];
    out[i] = accumulate ? out[i] + v : v;
  }
}

extern "C" int unetk_flip_axpy(const float* x, float* out, int N, int H, int W, int C, int flip_h, int flip_w, float scale,
                               int accumulate, void* stream) {
  UNETK_REQUIRE(x && out && x != out && N > 0 && H > 0 && W > 0 && C > 0);
  const int64_t total = (int64_t)N * H * W * C;
  UNETK_LAUNCH(flip_axpy_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, H, W, C, flip_h,
                     flip_w, scale, accumulate);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_image_gradients(const float* x, float* out, int N, int H, int W, int C, void* stream) {
  UNETK_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0);
  const int64_t total = (int64_t)N * H * W * C;
  UNETK_LAUNCH(image_gradients_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool2_fwd(const float* x, int x_stride, float* p, int N, int H, int W, int C,
                                  void* stream) {
  UNETK_REQUIRE(x && p && N > 0 && H > 1 && W > 1 && C > 0 && x_stride >= C);
  if (C % 4 != 0 || x_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(p));
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  UNETK_LAUNCH(maxpool2_fwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_stride, p, N,
                     H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool2_fwd_bf16(const void* x, int x_stride, void* p, int N, int H, int W, int C, void* stream) {
  UNETK_REQUIRE(x && p && N > 0 && H > 1 && W > 1 && C > 0 && x_stride >= C);
  if (C % 4 != 0 || x_stride % 4 != 0) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned8(x) && unetk_aligned8(p));
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  UNETK_LAUNCH(maxpool2_fwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, x_stride, (bf16_t*)p, N, H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool2_bwd(const float* x, int x_stride, const float* p, const float* dp, const float* add,
                                  int add_stride, float* dx, int N, int H, int W, int C, void* stream) {
  UNETK_REQUIRE(x && p && dp && dx && N > 0 && H > 1 && W > 1 && C > 0 && x_stride >= C);
  if (C % 4 != 0 || x_stride % 4 != 0 || (H & 1) || (W & 1)) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(p) && unetk_aligned16(dp) && unetk_aligned16(dx));
  UNETK_REQUIRE(!add || (add_stride >= C && add_stride % 4 == 0 && unetk_aligned16(add)));
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  UNETK_LAUNCH(maxpool2_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, x_stride, p, dp,
                     add, add_stride, dx, N, H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_maxpool2_bwd_bf16(const void* x, int x_stride, const void* p, const void* dp, const void* add,
                                       int add_stride, void* dx, int N, int H, int W, int C, void* stream) {
  UNETK_REQUIRE(x && p && dp && dx && N > 0 && H > 1 && W > 1 && C > 0 && x_stride >= C);
  if (C % 4 != 0 || x_stride % 4 != 0 || (H & 1) || (W & 1)) return UNETK_E_UNSUPPORTED;
  UNETK_REQUIRE(unetk_aligned8(x) && unetk_aligned8(p) && unetk_aligned8(dp) && unetk_aligned8(dx));
  UNETK_REQUIRE(!add || (add_stride >= C && add_stride % 4 == 0 && unetk_aligned8(add)));
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  UNETK_LAUNCH(maxpool2_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, x_stride, (const bf16_t*)p, (const bf16_t*)dp, (const bf16_t*)add, add_stride,
                     (bf16_t*)dx, N, H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_avgpool2_fwd(const float* x, float* p, int N, int H, int W, int C, void* stream) {
  UNETK_REQUIRE(x && p && N > 0 && H > 1 && W > 1 && C > 0);
  if ((H & 1) || (W & 1)) return UNETK_E_UNSUPPORTED;
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
  UNETK_LAUNCH(avgpool2_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, p, N, H, W, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

// tf.reduce_mean(x, axis=(1, 2)) of the conv context subnet (GUNet.py:108): y[n][c] = mean over the sample's HW pixels.
// Latency-sized (32 x 32 x 128 per sample): block = (sample, 64 channels) x 4 pixel lanes, fixed-order sums.
__global__ __launch_bounds__(256) void spatial_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t HW,
                                                               int C) {
  __shared__ float red[4][64];
  const int cb = blockIdx.x, n = blockIdx.y;
  const int c = cb * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int64_t p = pl; p < HW; p += 4) s += x[((int64_t)n * HW + p) * C + c];
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < C)
    y[(int64_t)n * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW;
}

__global__ void spatial_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t HW, int C, int64_t total) {
  const float inv = 1.0f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t n = i / C / HW;
    dx[i] = dy[n * C + c] * inv;
  }
}

extern "C" int unetk_spatial_mean_fwd(const float* x, float* y, int N, int64_t HW, int C, void* stream) {
  UNETK_REQUIRE(x && y && N > 0 && N <= 65535 && HW > 0 && C > 0);
  UNETK_LAUNCH(spatial_mean_fwd_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, x, y, HW, C);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_spatial_mean_bwd(const float* dy, float* dx, int N, int64_t HW, int C, void* stream) {
  UNETK_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0);
  const int64_t total = (int64_t)N * HW * C;
  UNETK_LAUNCH(spatial_mean_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, HW, C, total);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_guide_moments(const float* guide, int N, int64_t HW, int G, int per_sample, float* out, void* stream) {
  UNETK_REQUIRE(guide && out && N > 0 && HW > 0 && G > 0 && G <= 4);
  const int groups = per_sample ? N : 1;
  const int64_t P = per_sample ? HW : (int64_t)N * HW;
  UNETK_LAUNCH(guide_moments_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, guide, P, G, out);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
