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
