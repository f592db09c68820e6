// Training-batch assembly of the LiTS input pipeline on the device.
//
// The reference decodes three 16-bit PNG slices per sample on CPU threads and runs, per sample, inside tf.data:
// crop_to_bounding_box -> resize_bilinear(align_corners=True) -> clip to the window -> (x - lo) / (hi - lo), the label
// through crop -> resize_nearest_neighbor(align_corners=True) -> / LB_SCALE, then uniform noise (zero on padded
// slices) and random flips (DataLoader/Liver/input_pipeline.py:243-284; utils/image_ops.py).  On MI355X the decoded
// dataset (131 cases ~ 35 GB as uint16) simply LIVES in HBM; a step's batch is one kernel over the output pixels that
// gathers from the resident slices -- no host work, no host->device copy per step.
//   sample table row (int32): [slice index x C (-1 = zero padding slice), label slice index (-1 = zeros),
//                              off_y, off_x, crop_h, crop_w, flip_left_right, flip_up_down]
//   clip row (float):         [window lo, window hi]          (IM_SCALE units, input_pipeline.py:370-373)
// HBM-bound and tiny: reads <= 4 source pixels per output value.
#include "common.h"

namespace {

__device__ __forceinline__ float u01(uint32_t seed, uint64_t i) {   // counter-based: splitmix64 finaliser
  uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull + ((uint64_t)seed << 32 | seed);
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void lits_batch_kernel(unetk_lits_desc d, const uint16_t* __restrict__ slices,
                                                         const uint8_t* __restrict__ segs, const int32_t* __restrict__ tab,
                                                         const float* __restrict__ clip, float* __restrict__ images,
                                                         int32_t* __restrict__ labels) {
  const int64_t total = (int64_t)d.N * d.H * d.W;
  const int TW = d.C + 7;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % d.W);
    const int y = (int)((i / d.W) % d.H);
    const int n = (int)(i / ((int64_t)d.W * d.H));
    const int32_t* t = tab + (int64_t)n * TW;
    // a crop box outside the slice is a caller error (tf.image.crop_to_bounding_box raises); it is clamped into the
    // slice here so that it can never become an out-of-bounds read
    const int off_y = min(max(t[d.C + 1], 0), d.src_h - 1), off_x = min(max(t[d.C + 2], 0), d.src_w - 1);
    const int ch = min(max(t[d.C + 3], 1), d.src_h - off_y), cw = min(max(t[d.C + 4], 1), d.src_w - off_x);
    // random flips act on the finished sample: output (y, x) shows source position (sy, sx)
    const int sx = t[d.C + 5] ? d.W - 1 - x : x;
    const int sy = t[d.C + 6] ? d.H - 1 - y : y;
    // align_corners: in = out * (in_size - 1) / (out_size - 1)
    const float hs = d.H > 1 ? (float)(ch - 1) / (float)(d.H - 1) : 0.f;
    const float ws = d.W > 1 ? (float)(cw - 1) / (float)(d.W - 1) : 0.f;
    const float in_y = sy * hs, in_x = sx * ws;
    const int y0 = (int)floorf(in_y), x0 = (int)floorf(in_x);
    const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
    const float ly = in_y - y0, lx = in_x - x0;
    const float lo = clip[2 * n], hi = clip[2 * n + 1];
    const int64_t plane = (int64_t)d.src_h * d.src_w;
    for (int c = 0; c < d.C; ++c) {
      float v = 0.f;
      const int s = t[c];
      if (s >= 0 && s < d.n_slices) {
        const uint16_t* p = slices + s * plane + (int64_t)off_y * d.src_w + off_x;
        const float tl = p[(int64_t)y0 * d.src_w + x0], tr = p[(int64_t)y0 * d.src_w + x1];
        const float bl = p[(int64_t)y1 * d.src_w + x0], br = p[(int64_t)y1 * d.src_w + x1];
        const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;     // tf resize_bilinear's lerp order
        v = top + (bot - top) * ly;
        v = (fminf(fmaxf(v, lo), hi) - lo) / (hi - lo);
        if (d.noise_scale > 0.f)   // noise is drawn per FINAL pixel (after the flip in the reference too)
          v += (2.f * u01(d.seed, ((uint64_t)i * d.C + c)) - 1.f) * d.noise_scale;
      }
      images[i * d.C + c] = v;
    }
    int lab = 0;
    const int ls = t[d.C];
    if (ls >= 0 && ls < d.n_slices) {
      const int ny = min((int)roundf(in_y), ch - 1), nx = min((int)roundf(in_x), cw - 1);   // nearest, align_corners
      lab = segs[ls * plane + (int64_t)(off_y + ny) * d.src_w + off_x + nx] / d.lab_scale;
    }
    labels[i] = lab;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// PNG scanline un-filtering on the device (the loader of the resident slice store, data/lits.py).
//
// The reference decodes its PNG slices with cv2 on tf.data threads (DataLoader/Liver/input_pipeline.py:243-284; they were
// written by SimpleITK / libpng with ADAPTIVE row filters, DataLoader/Liver/extract.py:176-187).  Here the host only inflates
// the zlib stream (zlib releases the GIL: a thread pool scales) and uploads the FILTERED scanlines; this kernel undoes the
// five PNG filter types and writes native-endian pixels straight into the resident store.
//
// Every filter is  cur[i] = line[i] + pred(a, b, c)  mod 256 per byte lane, a = left pixel, b = the pixel above, c = above-left
// (None: 0, Sub: a, Up: b, Average: (a + b) >> 1, Paeth).  Average and Paeth chain through BOTH a (this row) and b (the row
// above), so the dependency is a wavefront: one wave per image, lane r owns row y0 + r of a 64-row band and runs ONE pixel
// behind lane r - 1 -- at step t it takes pixel i = t - r, its `b` is the value lane r - 1 produced at step t - 1 (a DPP
// shuffle), `c` its previous `b`, `a` its own previous output.  A band is w + 63 steps whatever mix of filter types its rows
// use; the band's last row goes through LDS to lane 0 of the next band (written at step i + 63, read at step i of the NEXT
// band).  Bytes in, bytes out: the host's inflate rate (a few thousand slices/s) is the pipeline's limit, not this kernel.
constexpr int PNG_MAX_W = 4096;

template <int BPP>
__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t* __restrict__ filt, int64_t in_stride, int h, int w,
                                                          void* __restrict__ out, int64_t out_stride, int32_t* __restrict__ status) {
  __shared__ uint8_t lastrow[PNG_MAX_W * BPP];
  const int lane = threadIdx.x;
  const uint8_t* src = filt + (int64_t)blockIdx.x * in_stride;
  const int rowb = w * BPP + 1;
  for (int i = lane; i < w * BPP; i += 64) lastrow[i] = 0;           // the row above row 0 is zero
  __syncthreads();
  for (int y0 = 0; y0 < h; y0 += 64) {
    const int y = y0 + lane;
    const bool active = y < h;
    const uint8_t* line = src + (int64_t)(active ? y : 0) * rowb;
    const int ft = active ? line[0] : 0;
    if (active && ft > 4) atomicOr(status, 1);                         // not a PNG filter type: the caller raises
    line += 1;
    const bool last_of_band = lane == 63 || y == h - 1;
    int a0 = 0, a1 = 0, c0 = 0, c1 = 0, o0 = 0, o1 = 0;
    for (int t = 0; t < w + 63; ++t) {
      const int i = t - lane;
      int b0 = __shfl_up(o0, 1), b1 = BPP == 2 ? __shfl_up(o1, 1) : 0;
      const bool on = active && i >= 0 && i < w;
      if (lane == 0 && on) {
        b0 = lastrow[i * BPP];
        if (BPP == 2) b1 = lastrow[i * BPP + 1];
      }
      if (on) {
        const int x0 = line[i * BPP], x1 = BPP == 2 ? line[i * BPP + 1] : 0;
        int p0 = 0, p1 = 0;
        if (ft == 1) { p0 = a0; p1 = a1; }
        else if (ft == 2) { p0 = b0; p1 = b1; }
        else if (ft == 3) { p0 = (a0 + b0) >> 1; p1 = (a1 + b1) >> 1; }
        else if (ft == 4) {
          const int q0 = a0 + b0 - c0, pa0 = abs(q0 - a0), pb0 = abs(q0 - b0), pc0 = abs(q0 - c0);
          p0 = (pa0 <= pb0 && pa0 <= pc0) ? a0 : (pb0 <= pc0 ? b0 : c0);
          const int q1 = a1 + b1 - c1, pa1 = abs(q1 - a1), pb1 = abs(q1 - b1), pc1 = abs(q1 - c1);
          p1 = (pa1 <= pb1 && pa1 <= pc1) ? a1 : (pb1 <= pc1 ? b1 : c1);
        }
        o0 = (x0 + p0) & 255;
        o1 = (x1 + p1) & 255;
        c0 = b0; c1 = b1; a0 = o0; a1 = o1;
        if (BPP == 2)      // PNG samples are big-endian
          static_cast<uint16_t*>(out)[(int64_t)blockIdx.x * out_stride + (int64_t)y * w + i] = (uint16_t)((o0 << 8) | o1);
        else
          static_cast<uint8_t*>(out)[(int64_t)blockIdx.x * out_stride + (int64_t)y * w + i] = (uint8_t)o0;
        if (last_of_band) {
          lastrow[i * BPP] = (uint8_t)o0;
          if (BPP == 2) lastrow[i * BPP + 1] = (uint8_t)o1;
        }
      }
    }
    __syncthreads();       // the band's last row is complete before the next band's lane 0 reads it
  }
}

extern "C" int unetk_png_unfilter(const uint8_t* filtered, int64_t image_stride_bytes, int n_images, int h, int w, int bit_depth,
                                  void* out, int64_t out_image_stride, int32_t* status, void* stream) {
  UNETK_REQUIRE(filtered && out && status && n_images > 0 && h > 0 && w > 0 && w <= PNG_MAX_W);
  UNETK_REQUIRE(bit_depth == 8 || bit_depth == 16);
  const int bpp = bit_depth / 8;
  UNETK_REQUIRE(image_stride_bytes >= (int64_t)h * (w * bpp + 1) && out_image_stride >= (int64_t)h * w);
  if (bpp == 2) {
    UNETK_REQUIRE((((uintptr_t)out) & 1u) == 0);
    UNETK_LAUNCH(png_unfilter_kernel<2>, dim3(n_images), dim3(64), 0, (hipStream_t)stream, filtered, image_stride_bytes, h, w, out,
                 out_image_stride, status);
  } else {
    UNETK_LAUNCH(png_unfilter_kernel<1>, dim3(n_images), dim3(64), 0, (hipStream_t)stream, filtered, image_stride_bytes, h, w, out,
                 out_image_stride, status);
  }
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}

extern "C" int unetk_lits_batch(const unetk_lits_desc* d, const uint16_t* slices, const uint8_t* seg_slices,
                                const int32_t* sample_tab, const float* clip, float* images, int32_t* labels,
                                void* stream) {
  UNETK_REQUIRE(d && slices && seg_slices && sample_tab && clip && images && labels);
  UNETK_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C <= 8 && d->src_h > 0 && d->src_w > 0 && d->lab_scale > 0);
  const int64_t total = (int64_t)d->N * d->H * d->W;
  int64_t grid = (total + 255) / 256;
  if (grid > 65536) grid = 65536;
  UNETK_LAUNCH(lits_batch_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, *d, slices, seg_slices,
                     sample_tab, clip, images, labels);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
