// unetk_pack_many: every filter re-layout of a step in ONE launch.
//
// The matrix kernels read filters in MFMA-friendly layouts (K4 / K8 interleaved, tap-flipped and transposed for the input
// gradient, channel-pair-permuted under bf16 storage) that are rebuilt from the TF-layout master weights after every
// optimiser step: 17-37 launches of a few microseconds per step, one per layer.  The host keeps a table of the layers it
// has seen (ops.py: _PackCache) and this kernel walks it: block b belongs to the item whose block range holds b.
#include "pack.h"

namespace {

__global__ __launch_bounds__(256) void pack_many_kernel(const unetk_pack_item* __restrict__ items, int n_items) {
  // the item of this block: block ranges are ascending; a scalar walk over <= a few hundred entries
  int e = 0;
  const int b = blockIdx.x;
  while (e + 1 < n_items && b >= items[e + 1].block0) ++e;
  const unetk_pack_item it = items[e];
  const int lb = b - it.block0;
  const float* w = static_cast<const float*>(it.w);
  const int64_t step = (int64_t)it.nblocks * 256;
  int64_t total;
  switch (it.kind) {
    case UNETK_PACK_CONV3X3_F32:
      total = (int64_t)9 * it.Cin * it.Cout / 4;
      for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < total; i += step)
        unetk_pack::conv3x3_f32(w, it.Cin, it.Cout, static_cast<float*>(it.wp_fwd), static_cast<float*>(it.wp_dgrad), i);
      break;
    case UNETK_PACK_CONV3X3_BF16:
      total = (int64_t)9 * it.Cin * it.Cout / 8;
      for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < total; i += step)
        unetk_pack::conv3x3_bf16(w, it.Cin, it.Cout, static_cast<uint4*>(it.wp_fwd), static_cast<uint4*>(it.wp_dgrad), it.perm, i);
      break;
    case UNETK_PACK_DECONV_F32:
      total = (int64_t)it.Cin * it.Cout;
      for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < total; i += step)
        unetk_pack::deconv_f32(w, it.Cin, it.Cout, static_cast<float*>(it.wp_fwd), static_cast<float*>(it.wp_dgrad), i);
      break;
    case UNETK_PACK_DECONV_BF16:
      total = (int64_t)it.Cin * it.Cout / 2;
      for (int64_t i = (int64_t)lb * 256 + threadIdx.x; i < total; i += step)
        unetk_pack::deconv_bf16(w, it.Cin, it.Cout, static_cast<uint4*>(it.wp_fwd), static_cast<uint4*>(it.wp_dgrad), it.perm, i);
      break;
    default:
      break;
  }
}

}  // namespace

extern "C" int unetk_pack_item_blocks(int kind, int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return UNETK_E_BADARG;
  int64_t total;
  switch (kind) {
    case UNETK_PACK_CONV3X3_F32: if (Cin % 4 || Cout % 4) return UNETK_E_UNSUPPORTED; total = (int64_t)9 * Cin * Cout / 4; break;
    case UNETK_PACK_CONV3X3_BF16: if (Cin % 8 || Cout % 8) return UNETK_E_UNSUPPORTED; total = (int64_t)9 * Cin * Cout / 8; break;
    case UNETK_PACK_DECONV_F32: if (Cin % 4 || Cout % 4) return UNETK_E_UNSUPPORTED; total = (int64_t)Cin * Cout; break;
    case UNETK_PACK_DECONV_BF16: if (Cin % 8 || Cout % 8) return UNETK_E_UNSUPPORTED; total = (int64_t)Cin * Cout / 2; break;
    default: return UNETK_E_BADARG;
  }
  const int64_t g = (total + 255) / 256;
  return (int)(g > 512 ? 512 : (g < 1 ? 1 : g));
}

extern "C" int unetk_pack_many(const unetk_pack_item* items_dev, int n_items, int total_blocks, void* stream) {
  UNETK_REQUIRE(items_dev && n_items > 0 && total_blocks > 0 && unetk_aligned16(items_dev));
  UNETK_LAUNCH(pack_many_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
  UNETK_LAUNCH_CHECK();
  return UNETK_OK;
}
