// slim.conv3d (NetworksV2/UNet3D.py:153,165) with kernels (1,3,3) / (3,3,3), strides 1 / (1,2,2) / (2,2,2),
// TF `SAME` padding (asymmetric for even sizes at stride 2: 0 before, 1 after -- SURVEY.md B2), no bias.
//
// Composition over the 2-D fp32-MFMA kernels instead of a separate 3-D kernel family:
//   * a (kd,3,3) conv is the sum over depth taps dt of (1,3,3) convs on depth-shifted planes; each tap is ONE
//     launch of conv3x3_igemm over the (n, d_out) planes, addressed in place through ImgAddr (depth slices /
//     depth-strided views, no copies), accumulating into the output (epilogue y += acc); the tap that covers
//     every output plane runs last and produces the norm statistics.  Depth stride 2 is just a plane stride.
//   * H/W stride 2 (10.7 % of UNet3D's FLOPs): forward = the igemm kernel's native stride-2 tile (conv_igemm.hip,
//     S = 2) when Cin % 16 == 0 and Cout % 64 == 0, else stride-1 conv into a scratch tensor + subsample (which also
//     emits the statistic partials); input gradient = four output-parity classes contracted from the UNDILATED dy
//     (conv_igemm_lin.hip, GEN variant) when Cin % 64 == 0, else dilate dy + the stride-1 kernel; filter gradient =
//     dilate dy (zero insertion) + the stride-1 kernel, dW = wgrad_s1(x, dilate(dy)) exactly, at 4x MFMA work (a
//     parity-plane variant was measured and dropped: it stages the same bytes, which is what bounds that kernel).
#include "common.h"

#include <stdlib.h>

namespace {

struct Geo3 {
  int Do, Ho, Wo;       // output extents
  int pb_d;             // SAME pad-before in depth
  int off_h, off_w;     // position of output (ho, wo) in the stride-1 result: 2*ho + off  (stride 2 only)
};

inline int same_pb(int in, int k, int s) {
  const int out = (in + s - 1) / s;
  int total = (out - 1) * s + k - in;
  if (total < 0) total = 0;
  return total / 2;
}

Geo3 geo3(const unetk_conv3d_desc* d) {
  Geo3 g;
  g.Do = (d->D + d->sd - 1) / d->sd;
  g.Ho = (d->H + d->shw - 1) / d->shw;
  g.Wo = (d->W + d->shw - 1) / d->shw;
  g.pb_d = same_pb(d->D, d->kd, d->sd);
  // stride-1 conv pads 1 before; stride-2 SAME pads pb before => y_s2[o] = y_s1[2*o + 1 - pb]
  g.off_h = 1 - same_pb(d->H, 3, 2);
  g.off_w = 1 - same_pb(d->W, 3, 2);
  return g;
}

// valid output-plane range of depth tap dt: 0 <= do*sd - pb + dt < D
inline void tap_range(const unetk_conv3d_desc* d, const Geo3& g, int dt, int* lo, int* hi) {
  int l = g.pb_d - dt;
  l = l <= 0 ? 0 : (l + d->sd - 1) / d->sd;
  int h = (d->D - 1 + g.pb_d - dt);
  h = h < 0 ? -1 : h / d->sd;
  if (h > g.Do - 1) h = g.Do - 1;
  *lo = l;
  *hi = h;
}

bool desc_ok(const unetk_conv3d_desc* d) {
  return d && d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 &&
         (d->kd == 1 || d->kd == 3) && (d->sd == 1 || d->sd == 2) && (d->shw == 1 || d->shw == 2) &&
         !(d->kd == 1 && d->sd != 1) && d->x_stride >= d->Cin && d->y_stride >= d->Cout;
}

// y[n, do, ho, wo, :] = t[n, do, 2ho+off_h, 2wo+off_w, :]; statistic partials per block, blocks grouped per sample
__global__ __launch_bounds__(256) void subsample2_stats_kernel(const float* __restrict__ t, float* __restrict__ y,
                                                               float* __restrict__ stat, int planes_per_sample, int H,
                                                               int W, int Ho, int Wo, int C, int ys, int off_h,
                                                               int off_w, int cq_n, int rpi, int bps, int stat_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][rpi][C]
  const int cq = threadIdx.x % cq_n, rl = threadIdx.x / cq_n;
  const int n = blockIdx.x / bps, blk = blockIdx.x % bps;
  const int64_t P = (int64_t)planes_per_sample * Ho * Wo;       // output pixels of this sample
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  if (rl < rpi) {
    for (int64_t pix = (int64_t)blk * rpi + rl; pix < P; pix += (int64_t)bps * rpi) {
      const int wo = (int)(pix % Wo);
      const int64_t r = pix / Wo;
      const int ho = (int)(r % Ho);
      const int64_t plane = (int64_t)n * planes_per_sample + r / Ho;
      const float4 v = ldg4(t + ((plane * H + 2 * ho + off_h) * W + 2 * wo + off_w) * C + cq * 4);
      stg4(y + ((int64_t)n * P + pix) * ys + cq * 4, v);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
    }
    if (stat != nullptr) {
      stg4(&smem[(0 * rpi + rl) * C + cq * 4], s);
      stg4(&smem[(1 * rpi + rl) * C + cq * 4], sq);
    }
  }
  if (stat == nullptr) return;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float v = 0.f;
    for (int j = 0; j < rpi; ++j) v += smem[(k * rpi + j) * C + c];
    stat[((int64_t)k * stat_rows + blockIdx.x) * C + c] = v;
  }
}

// z (pre-zeroed) [planes, H, W, C]: z[p, 2ho+off_h, 2wo+off_w, :] = dy[p, ho, wo, :]
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ dy, int dys, float* __restrict__ z,
                                                      int64_t npix_out, int H, int W, int Ho, int Wo, int C, int off_h,
                                                      int off_w) {
  const int cq_n = C >> 2;
  const int64_t total = npix_out * cq_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    int64_t r = i / cq_n;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int64_t plane = r / Ho;
    const float4 v = ldg4(dy + ((plane * Ho + ho) * Wo + wo) * dys + cq * 4);
    stg4(z + ((plane * H + 2 * ho + off_h) * W + 2 * wo + off_w) * C + cq * 4, v);
  }
}

// space-to-depth (round 5): xs[po][ho][wo][cls * C + c] = x[po * sd + cd][2 ho + ph][2 wo + pw][c], cls = (cd * 2 + ph) * 2 + pw
// (cd only with depth stride 2): the parity classes of a strided conv's input side by side on the channel axis, so that the
// conv is a stride-1 contraction over xs with tap subsets (conv_igemm_lin.hip GRP)
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, int xs_in, float* __restrict__ out, int64_t npix_out,
                                                  int Do, int D, int H, int W, int Ho, int Wo, int C, int sd) {
  const int cq_n = C >> 2, ncls = sd == 2 ? 8 : 4;
  const int64_t total = npix_out * ncls * cq_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    int64_t r = i / cq_n;
    const int cls = (int)(r % ncls); r /= ncls;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int po = (int)(r % Do);
    const int64_t n = r / Do;
    const int cd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int64_t plane = n * D + (int64_t)po * sd + cd;
    const float4 v = ldg4(x + ((plane * H + 2 * ho + ph) * W + 2 * wo + pw) * xs_in + cq * 4);
    stg4(out + ((((n * Do + po) * Ho + ho) * Wo + wo) * ncls + cls) * (int64_t)C + cq * 4, v);
  }
}

inline ImgAddr planes(int HW_floats, int group, int step, int planes_total) {
  ImgAddr a;
  a.img_stride = (int64_t)step * HW_floats;
  a.group_stride = (int64_t)planes_total * HW_floats;
  a.group = group;
  return a;
}

const int SUB_BPS = 64;   // subsample blocks per sample

// A strided conv whose OUTPUT planes are small (W < 32: the linear-pixel kernel's range) and whose input extents are even
// (SAME padding then puts the one pad row / column / plane behind the data): forward = space-to-depth + grouped taps.
// UNETK_S2LIN=0 (measurement): the tiled stride-2 kernel as before.
bool s2lin_ok(const unetk_conv3d_desc* d, const Geo3& g) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("UNETK_S2LIN"); on = e ? atoi(e) : 1; }
  if (!on || d->shw != 2 || (d->H & 1) || (d->W & 1) || d->Cin % 16 != 0) return false;
  if (d->sd == 2 && ((d->D & 1) || d->kd != 3)) return false;
  return unetk_conv_lin_ok(d->N * g.Do, g.Ho, g.Wo, d->Cin, d->Cout, g.Do);
}
inline size_t s2lin_xs_floats(const unetk_conv3d_desc* d, const Geo3& g) {
  return ((size_t)d->N * g.Do * g.Ho * g.Wo * (d->sd == 2 ? 8 : 4) * d->Cin + 63) & ~(size_t)63;
}

}  // namespace

extern "C" int unetk_conv3d_pack(const float* w, int kd, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                                 void* stream) {
  UNETK_REQUIRE(w && (kd == 1 || kd == 3) && Cin > 0 && Cout > 0);
  for (int dt = 0; dt < kd; ++dt) {
    const int64_t o = (int64_t)dt * 9 * Cin * Cout;
    int rc = unetk_conv3x3_pack(w + o, Cin, Cout, wp_fwd ? wp_fwd + o : nullptr, wp_dgrad ? wp_dgrad + o : nullptr, stream);
    if (rc != UNETK_OK) return rc;
  }
  return UNETK_OK;
}

extern "C" int unetk_conv3d_out_dims(const unetk_conv3d_desc* d, int* Do, int* Ho, int* Wo) {
  if (!desc_ok(d)) return UNETK_E_BADARG;
  const Geo3 g = geo3(d);
  if (Do) *Do = g.Do;
  if (Ho) *Ho = g.Ho;
  if (Wo) *Wo = g.Wo;
  return UNETK_OK;
}

extern "C" int unetk_conv3d_stat_rows(const unetk_conv3d_desc* d) {
  if (!desc_ok(d)) return UNETK_E_BADARG;
  const Geo3 g = geo3(d);
  if (d->shw == 2) {
    if (s2lin_ok(d, g)) return unetk_conv_stat_rows_lin(d->N * g.Do, g.Ho, g.Wo, g.Do, d->Cout);
    if (unetk_conv_stride2_ok(d->Cin, d->Cout)) return unetk_conv_stat_rows(d->N * g.Do, g.Ho, g.Wo, d->Cin, d->Cout, g.Do, 2);
    return d->N * SUB_BPS;
  }
  return unetk_conv_stat_rows(d->N * g.Do, d->H, d->W, d->Cin, d->Cout, g.Do);
}

// UNETK_KSKIP (measurement; read once): bit 0 = skip the dead 16-channel chunks of a channel-padded contraction axis
// (unetk_conv3d_desc.cin_live8 / cout_live8), bit 1 = half the MFMAs on a chunk whose upper 8 channels are padding.  Default 3.
static int kskip() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("UNETK_KSKIP"); v = e ? atoi(e) : 3; }
  return v;
}
static int live_k(const uint32_t live8[2], int K) { return (kskip() & 1) ? 16 * unetk_live_chunks(live8, K) : K; }
static void set_live(ConvParams& p, const uint32_t live8[2], int K) {
  if (kskip() & 1) unetk_set_klive(p, live8, K, (kskip() & 2) != 0);
}

extern "C" size_t unetk_conv3d_ws_bytes(const unetk_conv3d_desc* d) {
  if (!desc_ok(d)) return 0;
  const Geo3 g = geo3(d);
  size_t f = 0;
  if (d->shw == 2) f += (size_t)d->N * g.Do * d->H * d->W * d->Cout;          // stride-1 result / dilated dy
  f = (f + 63) & ~(size_t)63;
  size_t bytes = f * sizeof(float) + unetk_wgrad_ws_bytes(d->N * g.Do, d->H, d->W, d->Cin, d->Cout);
  if (d->shw == 1 && d->kd > 1) {   // fused depth taps (one launch for the whole (kd,3,3) filter gradient)
    const size_t fb = unetk_wgrad_ws_bytes(d->N * g.Do, d->H, d->W, d->Cin, d->Cout, d->kd);
    if (fb > bytes) bytes = fb;
  }
  if (d->shw == 2) {
    size_t sb = unetk_wgrad_strided_ws_bytes(d->N * g.Do, g.Ho, g.Wo, d->Cin, d->Cout);
    if (sb > bytes) bytes = sb;
    sb = unetk_wgrad_strided_ws_bytes(d->N * g.Do, g.Ho, g.Wo, d->Cin, d->Cout, d->kd);
    if (sb > bytes) bytes = sb;
  }
  if (s2lin_ok(d, g)) {              // space-to-depth copy of x + the stream-K slab of the grouped-tap forward
    const size_t sb = s2lin_xs_floats(d, g) * sizeof(float) +
                      unetk_conv_lin_sk_bytes(d->N * g.Do, g.Ho, g.Wo, live_k(d->cin_live8, d->Cin), d->Cout, g.Do, d->kd * 4);
    if (sb > bytes) bytes = sb;
  }
  if (d->shw == 1 && d->sd == 1) {   // stream-K slabs of the small-plane kernel, forward and input gradient
    const size_t kf = unetk_conv_lin_sk_bytes(d->N * d->D, d->H, d->W, live_k(d->cin_live8, d->Cin), d->Cout, d->D, d->kd);
    const size_t kb = unetk_conv_lin_sk_bytes(d->N * d->D, d->H, d->W, live_k(d->cout_live8, d->Cout), d->Cin, d->D, d->kd);
    if (kf > bytes) bytes = kf;
    if (kb > bytes) bytes = kb;
  }
  return bytes;
}

extern "C" int unetk_conv3d_fwd(const unetk_conv3d_desc* d, const float* x, const float* wp, float* y,
                                float* stat_partials, void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(desc_ok(d) && x && wp && y);
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(wp) && unetk_aligned16(y) && d->y_stride % 4 == 0);
  hipStream_t st = (hipStream_t)stream;
  const Geo3 g = geo3(d);
  // stride (.,2,2): natively strided tiles when the MFMA kernel covers the channel counts, else stride-1 + subsample
  const bool native = d->shw == 2 && unetk_conv_stride2_ok(d->Cin, d->Cout);
  const bool strided = d->shw == 2 && !native;
  float* T = y;
  int ts = d->y_stride;
  if (strided) {
    UNETK_REQUIRE(ws && unetk_aligned16(ws) && d->Cout % 4 == 0);
    if (ws_bytes < unetk_conv3d_ws_bytes(d)) return UNETK_E_WORKSPACE;
    T = (float*)ws;
    ts = d->Cout;
  }
  const int HWx = d->H * d->W * d->x_stride, HWt = (native ? g.Ho * g.Wo : d->H * d->W) * ts;
  if (d->kd == 3 && d->sd == 1 && d->shw == 1 && unetk_conv_lin_ok(d->N * d->D, d->H, d->W, d->Cin, d->Cout, d->D)) {
    // small planes, depth stride 1 (UNet3D's 24^2 / 12^2 / 6^2 levels): the three depth taps are contracted inside ONE
    // launch of the linear-pixel kernel (K = 27 Cin) -- no memset, no read-modify-write of y per tap
    ConvParams p{};
    p.x = x; p.wp = wp; p.y = y; p.stat = stat_partials;
    p.N = d->N * d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.xs = d->x_stride; p.ys = d->y_stride;
    p.xa = planes(HWx, d->D, 1, d->D);
    p.ya = planes(d->H * d->W * d->y_stride, d->D, 1, d->D);
    p.spg = d->D;
    p.kd = 3; p.dshift0 = -g.pb_d; p.dstep = 1;
    set_live(p, d->cin_live8, d->Cin);
    if (ws && unetk_aligned16(ws)) { p.sk_slab = (float*)ws; p.sk_slab_bytes = ws_bytes; }
    return unetk_conv_run(p, st);
  }
  if (s2lin_ok(d, g) && ws && unetk_aligned16(ws) && ws_bytes >= unetk_conv3d_ws_bytes(d)) {
    // small OUTPUT planes (UNet3D's 24^2 / 12^2 / 6^2 levels behind a stride-2 layer): space-to-depth copy of x, then ONE
    // launch of the linear-pixel kernel over kd x 4 tap groups (K = kd x 9 x Cin, every MFMA row an output pixel, stream-K)
    float* XS = (float*)ws;
    const int ncls = d->sd == 2 ? 8 : 4;
    const int64_t npix_out = (int64_t)d->N * g.Do * g.Ho * g.Wo;
    int64_t grid = (npix_out * ncls * (d->Cin / 4) + 255) / 256;
    if (grid > 8192) grid = 8192;
    UNETK_LAUNCH(s2d_kernel, dim3((int)grid), dim3(256), 0, st, x, d->x_stride, XS, npix_out, g.Do, d->D, d->H, d->W, g.Ho, g.Wo,
                 d->Cin, d->sd);
    UNETK_LAUNCH_CHECK();
    ConvParams p{};
    p.x = XS; p.wp = wp; p.y = y; p.stat = stat_partials;
    p.N = d->N * g.Do; p.H = g.Ho; p.W = g.Wo; p.Cin = d->Cin; p.Cout = d->Cout;
    p.xs = ncls * d->Cin; p.ys = d->y_stride;
    p.xa = planes(g.Ho * g.Wo * ncls * d->Cin, g.Do, 1, g.Do);
    p.ya = planes(g.Ho * g.Wo * d->y_stride, g.Do, 1, g.Do);
    p.spg = g.Do;
    int ng = 0;
    for (int kd_ = 0; kd_ < d->kd; ++kd_) {
      const int cd = d->sd == 2 ? (kd_ & 1) : 0;
      const int dz = d->sd == 2 ? (kd_ >> 1) : kd_ - g.pb_d;      // depth stride 1: input plane = do + kd - pb
      for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw, ++ng) {
          p.g_chan[ng] = ((cd * 2 + ph) * 2 + pw) * d->Cin;
          p.g_dz[ng] = dz;
          int nt = 0;
          for (int kh = ph; kh < 3; kh += 2)
            for (int kw = pw; kw < 3; kw += 2, ++nt) {
              p.g_off[ng][nt] = 4 * ((kh >> 1) + 1) + ((kw >> 1) + 1);      // output o reads class k & 1 at o + (k >> 1)
              p.g_panel[ng][nt] = kd_ * 9 + kh * 3 + kw;
            }
          p.g_ntaps[ng] = nt;
        }
    }
    p.ng = ng;
    set_live(p, d->cin_live8, d->Cin);
    const size_t xsb = s2lin_xs_floats(d, g) * sizeof(float);
    p.sk_slab = (float*)((char*)ws + xsb); p.sk_slab_bytes = ws_bytes - xsb;
    return unetk_conv_run_lin(p, st);
  }
  if (native && d->kd == 3) {
    // natively strided (3,3,3) conv (UNet3D's (1,2,2) and (2,2,2) down-sampling layers): the three depth taps are contracted
    // inside ONE launch of the tiled stride-2 kernel (K = 27 Cin; a tap whose input plane falls outside the sample is
    // skipped per block) -- no memset, no read-modify-write of y per tap, one prologue / epilogue instead of three
    ConvParams p{};
    p.x = x; p.wp = wp; p.y = y; p.stat = stat_partials;
    p.N = d->N * g.Do; p.H = g.Ho; p.W = g.Wo; p.Cin = d->Cin; p.Cout = d->Cout;
    p.xs = d->x_stride; p.ys = d->y_stride;
    p.xa = planes(HWx, g.Do, d->sd, d->D);
    p.ya = planes(g.Ho * g.Wo * d->y_stride, g.Do, 1, g.Do);
    p.spg = g.Do;
    p.stride = 2; p.Hin = d->H; p.Win = d->W; p.pbh = 1 - g.off_h; p.pbw = 1 - g.off_w;
    p.kd = 3; p.dshift0 = -g.pb_d; p.dstep = 1; p.dsd = d->sd; p.din = d->D; p.dplane = HWx;
    return unetk_conv_run(p, st);
  }
  // order: partial-coverage taps first, a full-coverage tap last (it emits the statistics)
  int order[3], n_taps = 0, full = -1;
  for (int dt = 0; dt < d->kd; ++dt) {
    int lo, hi;
    tap_range(d, g, dt, &lo, &hi);
    if (lo == 0 && hi == g.Do - 1 && full < 0) full = dt;
  }
  if (full < 0) return UNETK_E_UNSUPPORTED;
  for (int dt = 0; dt < d->kd; ++dt)
    if (dt != full) order[n_taps++] = dt;
  order[n_taps++] = full;
  if (d->kd > 1) {
    hipError_t e = hipMemsetAsync(T, 0, (size_t)d->N * g.Do * HWt * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  for (int i = 0; i < n_taps; ++i) {
    const int dt = order[i];
    int lo, hi;
    tap_range(d, g, dt, &lo, &hi);
    if (hi < lo) continue;
    ConvParams p{};
    p.x = x + (int64_t)(lo * d->sd - g.pb_d + dt) * HWx;
    p.wp = wp + (int64_t)dt * 9 * d->Cin * d->Cout;
    p.y = T + (int64_t)lo * HWt;
    p.stat = (i == n_taps - 1 && !strided) ? stat_partials : nullptr;
    p.N = d->N * (hi - lo + 1); p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.xs = d->x_stride; p.ys = ts;
    p.xa = planes(HWx, hi - lo + 1, d->sd, d->D);
    p.ya = planes(HWt, hi - lo + 1, 1, g.Do);
    p.accumulate = d->kd > 1 ? 1 : 0;
    p.spg = hi - lo + 1;
    if (native) {
      p.stride = 2;
      p.H = g.Ho; p.W = g.Wo; p.Hin = d->H; p.Win = d->W;
      p.pbh = 1 - g.off_h; p.pbw = 1 - g.off_w;
    }
    int rc = unetk_conv_run(p, st);
    if (rc != UNETK_OK) return rc;
  }
  if (strided) {
    const ColMap m = unetk_colmap(d->Cout);
    if (d->Cout > 1024) return UNETK_E_UNSUPPORTED;
    const size_t lds = stat_partials ? (size_t)2 * m.rows_per_iter * d->Cout * sizeof(float) : 0;
    UNETK_LAUNCH(subsample2_stats_kernel, dim3(d->N * SUB_BPS), dim3(256), lds, st, T, y, stat_partials, g.Do,
                       d->H, d->W, g.Ho, g.Wo, d->Cout, d->y_stride, g.off_h, g.off_w, m.cq_n, m.rows_per_iter, SUB_BPS,
                       d->N * SUB_BPS);
    UNETK_LAUNCH_CHECK();
  }
  return UNETK_OK;
}

namespace {
// dy -> (Z, stride) usable by the stride-1 backward kernels
int dilated_dy(const unetk_conv3d_desc* d, const Geo3& g, const float* dy, void* ws, hipStream_t st, const float** z,
               int* zs) {
  if (d->shw == 1) {
    *z = dy;
    *zs = d->y_stride;
    return UNETK_OK;
  }
  float* Z = (float*)ws;
  const size_t n = (size_t)d->N * g.Do * d->H * d->W * d->Cout;
  hipError_t e = hipMemsetAsync(Z, 0, n * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  const int64_t npix_out = (int64_t)d->N * g.Do * g.Ho * g.Wo;
  int64_t grid = (npix_out * (d->Cout / 4) + 255) / 256;
  if (grid > 8192) grid = 8192;
  UNETK_LAUNCH(dilate2_kernel, dim3((int)grid), dim3(256), 0, st, dy, d->y_stride, Z, npix_out, d->H, d->W, g.Ho,
                     g.Wo, d->Cout, g.off_h, g.off_w);
  UNETK_LAUNCH_CHECK();
  *z = Z;
  *zs = d->Cout;
  return UNETK_OK;
}
}  // namespace

extern "C" int unetk_conv3d_dgrad(const unetk_conv3d_desc* d, const float* dy, const float* wp_dgrad, float* dx,
                                  void* ws, size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(desc_ok(d) && dy && wp_dgrad && dx);
  UNETK_REQUIRE(unetk_aligned16(dy) && unetk_aligned16(wp_dgrad) && unetk_aligned16(dx));
  UNETK_REQUIRE(d->x_stride % 4 == 0 && d->y_stride % 4 == 0 && d->Cout % 4 == 0);
  if (d->Cout % 16 != 0 || d->Cin % 32 != 0) return UNETK_E_UNSUPPORTED;   // the MFMA kernels' K and N granularity
  hipStream_t st = (hipStream_t)stream;
  const Geo3 g = geo3(d);
  if (d->shw == 2 && unetk_conv_lin_gen_ok(g.Ho, g.Wo, d->Cout, d->Cin)) {
    // native: one launch per output-parity class of dx, each contracting the undilated dy over its 4 / 2 / 2 / 1 taps
    //   dx[hi] = sum_{kh = ph, ph+2} dy[a - (kh - ph)/2] w[kh],  ph = (hi + pbh) & 1,  a = (hi + pbh - ph) / 2
    const int pbh = 1 - g.off_h, pbw = 1 - g.off_w;
    const int HWx = d->H * d->W * d->x_stride, HWy = g.Ho * g.Wo * d->y_stride;
    // depth stride 1 with three depth taps (UNet3D's (1,2,2) layers): every dx plane receives all three taps, so they are
    // contracted in ONE launch (K = 3 x taps x Cout) -- no memset, no accumulation passes
    const bool fuse_d = d->kd == 3 && d->sd == 1 && d->Cin % 32 == 0;
    // depth stride 2 (the (2,2,2) bridge; round 5): dx plane di = 2 do + dt (pad-before 0 for an even depth), so the EVEN dx
    // planes receive the taps dt = 0 (do = di / 2) and dt = 2 (do = di / 2 - 1), the ODD ones dt = 1 alone -- two launches that
    // each write their planes once (fused taps; no memset, no read-modify-write of dx: three accumulating launches before)
    static int s2d_on = -1;
    if (s2d_on < 0) { const char* e = getenv("UNETK_S2LIN"); s2d_on = e ? atoi(e) : 1; }
    if (s2d_on && d->kd == 3 && d->sd == 2 && (d->D & 1) == 0 && g.pb_d == 0 && d->Cin % 32 == 0) {
      ConvParams p{};
      p.x = dy; p.wp = wp_dgrad; p.y = dx;
      p.N = d->N * g.Do; p.H = g.Ho; p.W = g.Wo; p.Cin = d->Cout; p.Cout = d->Cin;
      p.xs = d->y_stride; p.ys = d->x_stride;
      p.xa = planes(HWy, g.Do, 1, g.Do);
      p.ya = planes(HWx, g.Do, 2, d->D);
      p.spg = g.Do;
      p.dshift0 = 0; p.dstep = -1;           // even planes: dy plane do = di / 2 - dt' for the fused taps dt' = 0, 1 (dt = 0, 2)
      p.dpar = 1; p.dpar_yoff = HWx;          // odd planes: one plane further
      set_live(p, d->cout_live8, d->Cout);
      p.os = 2; p.Hd = d->H; p.Wd = d->W;
      for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
          const int q = ph * 2 + pw;
          p.ooh[q] = ph - pbh;
          p.oow[q] = pw - pbw;
          for (int kh = ph; kh < 3; kh += 2)
            for (int kw = pw; kw < 3; kw += 2) {
              p.tap_off[q][p.ntaps[q]] = 4 * ((ph - kh) / 2 + 1) + ((pw - kw) / 2 + 1);
              p.tap_panel[q][p.ntaps[q]] = 8 - (kh * 3 + kw);
              ++p.ntaps[q];
            }
        }
      return unetk_conv_run_lin_gen(p, st);
    }
    const bool multi = !fuse_d && (d->kd > 1 || d->sd > 1);
    if (multi) {
      hipError_t e = hipMemsetAsync(dx, 0, (size_t)d->N * d->D * HWx * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
    for (int dt = 0; dt < (fuse_d ? 1 : d->kd); ++dt) {
      int lo, hi;
      tap_range(d, g, dt, &lo, &hi);
      if (fuse_d) { lo = 0; hi = g.Do - 1; }
      if (hi < lo) continue;
      ConvParams p{};
      p.x = dy + (int64_t)lo * HWy;
      p.wp = wp_dgrad + (int64_t)dt * 9 * d->Cin * d->Cout;
      p.y = dx + (int64_t)(fuse_d ? 0 : lo * d->sd - g.pb_d + dt) * HWx;
      p.N = d->N * (hi - lo + 1); p.H = g.Ho; p.W = g.Wo; p.Cin = d->Cout; p.Cout = d->Cin;
      p.xs = d->y_stride; p.ys = d->x_stride;
      p.xa = planes(HWy, hi - lo + 1, 1, g.Do);
      p.ya = planes(HWx, hi - lo + 1, d->sd, d->D);
      p.accumulate = multi ? 1 : 0;
      p.spg = hi - lo + 1;
      if (fuse_d) { p.kd = 3; p.dshift0 = g.pb_d; p.dstep = -1; }   // dx[di] = sum_dt dy[di + pb - dt] * w[dt]
      set_live(p, d->cout_live8, d->Cout);
      p.os = 2; p.Hd = d->H; p.Wd = d->W;
      for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
          const int q = ph * 2 + pw;
          p.ooh[q] = ph - pbh;
          p.oow[q] = pw - pbw;
          for (int kh = ph; kh < 3; kh += 2)
            for (int kw = pw; kw < 3; kw += 2) {
              p.tap_off[q][p.ntaps[q]] = 4 * ((ph - kh) / 2 + 1) + ((pw - kw) / 2 + 1);
              p.tap_panel[q][p.ntaps[q]] = 8 - (kh * 3 + kw);
              ++p.ntaps[q];
            }
        }
      int rc = unetk_conv_run_lin_gen(p, st);
      if (rc != UNETK_OK) return rc;
    }
    return UNETK_OK;
  }
  if (d->shw == 2) {
    UNETK_REQUIRE(ws && unetk_aligned16(ws));
    if (ws_bytes < unetk_conv3d_ws_bytes(d)) return UNETK_E_WORKSPACE;
  }
  if (d->kd == 3 && d->sd == 1 && d->shw == 1 && unetk_conv_lin_ok(d->N * d->D, d->H, d->W, d->Cout, d->Cin, d->D)) {
    // fused depth taps (see unetk_conv3d_fwd): dx[di] = sum_dt dy[di + pb - dt] * w[dt]
    ConvParams p{};
    p.x = dy; p.wp = wp_dgrad; p.y = dx; p.stat = nullptr;
    p.N = d->N * d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cout; p.Cout = d->Cin;
    p.xs = d->y_stride; p.ys = d->x_stride;
    p.xa = planes(d->H * d->W * d->y_stride, d->D, 1, d->D);
    p.ya = planes(d->H * d->W * d->x_stride, d->D, 1, d->D);
    p.spg = d->D;
    p.kd = 3; p.dshift0 = g.pb_d; p.dstep = -1;
    set_live(p, d->cout_live8, d->Cout);
    if (ws && unetk_aligned16(ws)) { p.sk_slab = (float*)ws; p.sk_slab_bytes = ws_bytes; }
    return unetk_conv_run(p, st);
  }
  const float* Z;
  int zs;
  int rc = dilated_dy(d, g, dy, ws, st, &Z, &zs);
  if (rc != UNETK_OK) return rc;
  const int HWx = d->H * d->W * d->x_stride, HWz = d->H * d->W * zs;
  const bool multi = d->kd > 1 || d->sd > 1;
  if (multi) {
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)d->N * d->D * HWx * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  for (int dt = 0; dt < d->kd; ++dt) {
    int lo, hi;
    tap_range(d, g, dt, &lo, &hi);
    if (hi < lo) continue;
    ConvParams p{};
    p.x = Z + (int64_t)lo * HWz;                                              // input of the dgrad conv = dy planes
    p.wp = wp_dgrad + (int64_t)dt * 9 * d->Cin * d->Cout;
    p.y = dx + (int64_t)(lo * d->sd - g.pb_d + dt) * HWx;                      // output = dx planes di = do*sd - pb + dt
    p.stat = nullptr;
    p.N = d->N * (hi - lo + 1); p.H = d->H; p.W = d->W; p.Cin = d->Cout; p.Cout = d->Cin;
    p.xs = zs; p.ys = d->x_stride;
    p.xa = planes(HWz, hi - lo + 1, 1, g.Do);
    p.ya = planes(HWx, hi - lo + 1, d->sd, d->D);
    p.accumulate = multi ? 1 : 0;
    p.spg = hi - lo + 1;
    rc = unetk_conv_run(p, st);
    if (rc != UNETK_OK) return rc;
  }
  return UNETK_OK;
}

extern "C" int unetk_conv3d_wgrad(const unetk_conv3d_desc* d, const float* x, const float* dy, float* dw, void* ws,
                                  size_t ws_bytes, void* stream) {
  UNETK_REQUIRE(desc_ok(d) && x && dy && dw && ws && unetk_aligned16(ws));
  UNETK_REQUIRE(unetk_aligned16(x) && unetk_aligned16(dy) && unetk_aligned16(dw) && d->y_stride % 4 == 0);
  if (ws_bytes < unetk_conv3d_ws_bytes(d)) return UNETK_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const Geo3 g = geo3(d);
  if (d->shw == 2 && unetk_wgrad_strided_ok(d->Cin, d->Cout)) {
    // natively strided filter gradient: tiles of OUTPUT pixels, tap (kh, kw) of output (oh, ow) reads input
    // (2 oh - pb + kh, 2 ow - pb + kw); no zero-dilated copy of dy, a quarter of the MFMA work
    const int HWx = d->H * d->W * d->x_stride, HWy = g.Ho * g.Wo * d->y_stride;
    if (d->kd == 3) {
      // the three depth taps in ONE launch (WgParams::kd): every dy plane against the x plane each tap reads
      WgParams p{};
      p.x = x; p.dy = dy;
      p.N = d->N * g.Do; p.H = g.Ho; p.W = g.Wo; p.Cin = d->Cin; p.Cout = d->Cout;
      p.xs = d->x_stride; p.ys = d->y_stride;
      p.stride = 2; p.Hin = d->H; p.Win = d->W; p.pbh = same_pb(d->H, 3, 2); p.pbw = same_pb(d->W, 3, 2);
      p.xa = planes(HWx, g.Do, d->sd, d->D);
      p.ya = planes(HWy, g.Do, 1, g.Do);
      p.kd = 3; p.dshift0 = -g.pb_d; p.dsd = d->sd; p.din = d->D; p.spg = g.Do; p.dplane = HWx;
      return unetk_wgrad_run(p, dw, ws, ws_bytes, st);
    }
    for (int dt = 0; dt < d->kd; ++dt) {
      int lo, hi;
      tap_range(d, g, dt, &lo, &hi);
      float* dw_t = dw + (int64_t)dt * 9 * d->Cin * d->Cout;
      if (hi < lo) {
        hipError_t e = hipMemsetAsync(dw_t, 0, (size_t)9 * d->Cin * d->Cout * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
        continue;
      }
      WgParams p{};
      p.x = x + (int64_t)(lo * d->sd - g.pb_d + dt) * HWx;
      p.dy = dy + (int64_t)lo * HWy;
      p.N = d->N * (hi - lo + 1); p.H = g.Ho; p.W = g.Wo; p.Cin = d->Cin; p.Cout = d->Cout;
      p.xs = d->x_stride; p.ys = d->y_stride;
      p.stride = 2; p.Hin = d->H; p.Win = d->W; p.pbh = same_pb(d->H, 3, 2); p.pbw = same_pb(d->W, 3, 2);
      p.xa = planes(HWx, hi - lo + 1, d->sd, d->D);
      p.ya = planes(HWy, hi - lo + 1, 1, g.Do);
      const int rc = unetk_wgrad_run(p, dw_t, ws, ws_bytes, st);
      if (rc != UNETK_OK) return rc;
    }
    return UNETK_OK;
  }
  const float* Z;
  int zs;
  int rc = dilated_dy(d, g, dy, ws, st, &Z, &zs);
  if (rc != UNETK_OK) return rc;
  size_t zf = d->shw == 2 ? (size_t)d->N * g.Do * d->H * d->W * d->Cout : 0;
  zf = (zf + 63) & ~(size_t)63;
  float* wws = (float*)ws + zf;
  const size_t wws_bytes = ws_bytes - zf * sizeof(float);
  const int HWx = d->H * d->W * d->x_stride, HWz = d->H * d->W * zs;
  if (d->kd == 3 && d->shw == 1 && unetk_wgrad_ws_bytes(d->N * g.Do, d->H, d->W, d->Cin, d->Cout, 3) > 0) {
    // the three depth taps in ONE launch (WgParams::kd): a third of the splits fills the chip, so every block walks three
    // times the tiles and the slab round trip shrinks accordingly
    WgParams p{};
    p.x = x; p.dy = Z;
    p.N = d->N * g.Do; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.xs = d->x_stride; p.ys = zs;
    p.xa = planes(HWx, g.Do, d->sd, d->D);
    p.ya = planes(HWz, g.Do, 1, g.Do);
    p.kd = 3; p.dshift0 = -g.pb_d; p.dsd = d->sd; p.din = d->D; p.spg = g.Do; p.dplane = HWx;
    return unetk_wgrad_run(p, dw, wws, wws_bytes, st);
  }
  for (int dt = 0; dt < d->kd; ++dt) {
    int lo, hi;
    tap_range(d, g, dt, &lo, &hi);
    float* dw_t = dw + (int64_t)dt * 9 * d->Cin * d->Cout;
    if (hi < lo) {
      hipError_t e = hipMemsetAsync(dw_t, 0, (size_t)9 * d->Cin * d->Cout * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
      continue;
    }
    WgParams p{};
    p.x = x + (int64_t)(lo * d->sd - g.pb_d + dt) * HWx;
    p.dy = Z + (int64_t)lo * HWz;
    p.N = d->N * (hi - lo + 1); p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.xs = d->x_stride; p.ys = zs;
    p.xa = planes(HWx, hi - lo + 1, d->sd, d->D);
    p.ya = planes(HWz, hi - lo + 1, 1, g.Do);
    rc = unetk_wgrad_run(p, dw_t, wws, wws_bytes, st);
    if (rc != UNETK_OK) return rc;
  }
  return UNETK_OK;
}
