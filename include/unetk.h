/*
 * unetk.h -- C ABI of libunetk.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * U-Net conv encoder/decoder + loss-head hot path of Jarvis73/BoxSegLiver.
 *
 * The reference has NO native boundary for this path (it is pure Python on
 * tensorflow-gpu 1.13).  Each entry point below therefore cites the
 * reference *call site* whose TensorFlow op it replaces (paths relative to the
 * reference repo root).
 *
 * Conventions
 *   - all tensors are device pointers, fp32, NHWC, densely packed unless a
 *     "*_stride" (pixel stride, in ELEMENTS) argument says otherwise; labels int32.
 *     Activation tensors passed as `void*` are fp32, or bf16 (2 bytes / element) when the
 *     descriptor selects UNETK_BF16S ("bf16 storage", see below)
 *   - conv filters are TF HWIO [kh,kw,Cin,Cout]; transposed-conv filters are TF
 *     [kh,kw,Cout,Cin]  (slim.conv2d / slim.conv2d_transpose variable layouts)
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it
 *   - no allocation, no host sync, no global mutable state: caller owns all buffers,
 *     including workspaces sized by the *_ws_bytes() queries  (graph-capturable)
 *   - return 0 on success, else a negative UNETK_E_* or a positive hipError_t
 */
#ifndef UNETK_H_
#define UNETK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNETK_OK 0
#define UNETK_E_BADARG (-1)       /* null pointer / non-positive size / misaligned */
#define UNETK_E_UNSUPPORTED (-2)  /* valid request this build has no kernel for */
#define UNETK_E_WORKSPACE (-3)    /* workspace too small */

#define UNETK_MAX_CLASSES 8

/* loss head weight modes: loss_metrics.py:115-165 `_compute_weights` w_type */
#define UNETK_W_NONE 0
#define UNETK_W_NUMERICAL 1
#define UNETK_W_PROPORTION 2
#define UNETK_W_PIXELMAP 3 /* caller-provided per-pixel map (covers `boundary`, which the
                              reference computes on the host via py_func, :149-159) */

/* loss types: loss_metrics.py:42-45 --loss_type */
#define UNETK_LOSS_XENTROPY 1
#define UNETK_LOSS_DICE 2

int unetk_abi_version(void);
const char* unetk_error_string(int code);

/* ---------------------------------------------------------------- conv 3x3 (slim.conv2d(x, C, 3))
 * NetworksV2/UNet.py:79,85,94 -- stride 1, SAME, no bias (a normaliser follows).
 * Geometry: x [N,H,W,Cin] (pixel stride x_stride >= Cin), y [N,H,W,Cout] (pixel stride y_stride). */
typedef struct unetk_conv_desc {
  int32_t N, H, W, Cin, Cout;
  int32_t x_stride, y_stride;
  int32_t precision; /* UNETK_FP32 (exact fp32 MFMA), UNETK_BF16, or UNETK_BF16S (x / y / dy / dx are bf16) */
  int32_t dilation;  /* 0 / 1 = dense 3x3; 2 = slim.conv2d(..., rate=2) (SmallUNet.py:44-49: bridge, conv_d3/conv1):
                        taps at (2 kh, 2 kw), SAME pads 2.  fp32 only, Cin % 16 == 0 and Cout % 64 == 0 (wgrad: both % 64) */
} unetk_conv_desc;

/* Arithmetic of the dense contractions.  UNETK_FP32: v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain.
 * UNETK_BF16 (BASELINE.json configs[2], "bf16"): both operands rounded to bf16 (RNE) on their way into the
 * matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulation, fp32 tensors in memory, fp32 statistics, master
 * weights and optimiser.  Needs Cin % 32 == 0 and Cout % 32 == 0 (else UNETK_E_UNSUPPORTED: use UNETK_FP32 for
 * that layer) and filters packed by unetk_conv3x3_pack_bf16. */
#define UNETK_FP32 0
#define UNETK_BF16 1
/* UNETK_BF16S = UNETK_BF16 arithmetic + bf16 STORAGE of activations and activation gradients (BASELINE.json
 * configs[2]: "bf16 activations / weights-compute, fp32 master / accum / stats"): every tensor passed as `void*`
 * (conv / deconv inputs and outputs, the norm's y / z / dz / dy, pooling, the head's feature map and its gradient)
 * is bf16 in memory, rounded RNE from the fp32 accumulator by the kernel that produces it; statistics come from the
 * fp32 accumulators, master weights, weight gradients, norm parameters and the optimiser stay fp32.  Halves the
 * bytes of every HBM-bound pass and lets the filter gradient read its k-strided operands with ds_read_b64_tr_b16.
 * The first conv (Cin < 16, direct kernel) reads fp32 images and writes bf16.  2-D, stride 1, dilation 1 only. */
#define UNETK_BF16S 2

/* Re-layout HWIO filters for the MFMA kernels ("K4-interleaved": [tap][Cin/4][Cout][4]).
 * wp_fwd feeds unetk_conv3x3_fwd; wp_dgrad (taps flipped, Cin<->Cout swapped) feeds
 * unetk_conv3x3_dgrad.  Either output may be NULL.  Each holds 9*Cin*Cout floats. */
int unetk_conv3x3_pack(const float* w_hwio, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                       void* stream);

/* Every filter re-layout of a step in ONE launch.  An item = one filter tap-set: kind (which of the four packs above /
 * below), perm (1 = the UNETK_BF16S channel-pair permutation), Cin, Cout, the TF-layout source w and the two outputs
 * (either may be NULL), and the item's block range [block0, block0 + nblocks) inside the launch (nblocks from
 * unetk_pack_item_blocks, block0 = running sum; items ascending).  `items_dev` is the table in DEVICE memory (16-byte
 * aligned), total_blocks the sum of nblocks.  A 3-D filter is kd items (one per depth tap). */
#define UNETK_PACK_CONV3X3_F32 0   /* unetk_conv3x3_pack */
#define UNETK_PACK_CONV3X3_BF16 1  /* unetk_conv3x3_pack_bf16 (perm 0) / _bf16s (perm 1) */
#define UNETK_PACK_DECONV_F32 2    /* one depth tap of unetk_deconv3d_pack / unetk_deconv2x2_pack */
#define UNETK_PACK_DECONV_BF16 3   /* ... of the bf16 transposed-conv packs */
typedef struct {
  int32_t kind, perm, Cin, Cout, block0, nblocks, reserved0, reserved1;
  const void* w;
  void* wp_fwd;
  void* wp_dgrad;
  void* reserved2;
} unetk_pack_item;
int unetk_pack_item_blocks(int kind, int Cin, int Cout);
int unetk_pack_many(const unetk_pack_item* items_dev, int n_items, int total_blocks, void* stream);

/* UNETK_BF16 filters: bf16 "K8-interleaved" [tap][Cin/8][Cout][8]; each output holds 9*Cin*Cout bf16
 * (2 bytes each), 16-byte aligned.  Cin % 8 == 0 and Cout % 8 == 0. */
int unetk_conv3x3_pack_bf16(const float* w_hwio, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                            void* stream);

/* UNETK_BF16S filters: as unetk_conv3x3_pack_bf16, with the OUTPUT channels of each pack (Cout for wp_fwd, Cin for
 * wp_dgrad) permuted inside every 64-channel block -- column position n' holds channel 2 (n' & 31) + (n' >> 5 & 1) -- so
 * that a lane of the bf16-storage kernels owns two adjacent channels and stores them as one word.  Cin % 64 == 0 and
 * Cout % 64 == 0. */
int unetk_conv3x3_pack_bf16s(const float* w_hwio, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                             void* stream);

/* Number of per-channel statistic partial rows unetk_conv3x3_fwd writes (one per pixel tile). */
int unetk_conv3x3_stat_rows(const unetk_conv_desc* d);

/* y = conv3x3(x, w).  If stat_partials != NULL also writes per-pixel-tile partial sums for the
 * following norm: stat_partials[0][row][c] = sum y, stat_partials[1][row][c] = sum y^2
 * (2 * stat_rows * Cout floats; deterministic, no atomics).
 * `w` is wp_fwd from unetk_conv3x3_pack when Cin % 16 == 0 && Cout % 32 == 0, else the raw
 * HWIO filter (direct kernel; used by Encode1/conv1 where Cin = 3). */
int unetk_conv3x3_fwd(const unetk_conv_desc* d, const void* x, const void* w, void* y,
                      float* stat_partials, void* stream);

/* dx = conv3x3_input_grad(dy, w): d describes the FORWARD conv (dx has Cin channels,
 * pixel stride x_stride; dy has Cout channels, pixel stride y_stride).  `w` = wp_dgrad. */
int unetk_conv3x3_dgrad(const unetk_conv_desc* d, const void* dy, const void* w, void* dx,
                        void* stream);

/* unetk_conv3x3_fwd / _dgrad with a scratch buffer: small planes whose tile grid cannot fill the 256 CUs (the 16 x 16 bridge
 * of a 2-D net at 8 slices per GPU) are scheduled stream-K -- the (tile, K-chunk) sequence split evenly over the blocks,
 * pieces of split tiles summed in a fixed order from ws (deterministic).  unetk_conv3x3_ws_bytes = bytes the two calls may
 * use for this shape (0 = never); ws = NULL behaves as the plain entry points. */
size_t unetk_conv3x3_ws_bytes(const unetk_conv_desc* d);
int unetk_conv3x3_fwd_ws(const unetk_conv_desc* d, const void* x, const void* w, void* y, float* stat_partials,
                         void* ws, size_t ws_bytes, void* stream);
int unetk_conv3x3_dgrad_ws(const unetk_conv_desc* d, const void* dy, const void* w, void* dx, void* ws,
                           size_t ws_bytes, void* stream);

/* Inference (mode == EVAL, NetworksV2/base.py:71-79: is_training False -> slim.batch_norm uses the moving statistics;
 * --without_norm: conv + bias + ReLU, UNet.py:47-48): the normaliser's per-channel affine is known BEFORE the conv runs, so
 * the conv, slim.batch_norm and ReLU of one slim.conv2d(x, C, 3) -- and, when the unit feeds one, slim.max_pool2d(z, 2, 2)
 * (UNet.py:81) -- are ONE pass: z = relu(conv(x, w) * scale[c] + shift[c]) is written straight to z (pixel stride
 * d->y_stride: a channel slice of the decoder's concat buffer), the raw conv output never reaches memory.  SURVEY.md 7
 * step 2 / 8b: the conv forward taking (scale, shift).  scale / shift: [Cout] floats (unetk_norm_finalize rows 2, 3).
 * pooled != NULL: also pooled[N, H/2, W/2, .] (pixel stride pooled_stride) = max over each 2 x 2 window of z; H, W even.
 * ws / ws_bytes: the stream-K scratch of unetk_conv3x3_ws_bytes (small planes), may be NULL / 0.
 * unetk_conv3x3_fwd_affine_ok = 1 when the shape has the fused kernel (fp32: the tiled kernels, the Cout = 64 first layers
 * and -- without the pool -- the small-plane linear-pixel kernel; not strided or atrous convs, not the generic direct
 * kernel) -- else the caller runs conv + unetk_norm_apply_relu. */
int unetk_conv3x3_fwd_affine_ok(const unetk_conv_desc* d, int with_pool);
int unetk_conv3x3_fwd_affine(const unetk_conv_desc* d, const void* x, const void* w, const float* scale,
                             const float* shift, void* z, void* pooled, int pooled_stride, void* ws, size_t ws_bytes,
                             void* stream);

/* The same input gradient, fused with the norm-backward REDUCTION of the unit that produced the conv's input (the
 * slim.repeat(x, 2, slim.conv2d, ...) pairs of UNet.py:79,85,94: conv2's dx is the dz of conv1's norm + ReLU): while the
 * dx tile is in registers the epilogue reads prod_y (conv1's raw output, same [N,H,W,Cin], pixel stride prod_y_stride;
 * bf16 under UNETK_BF16S) and writes, per tile, the partial sums  sum du  and  sum du * xhat  (du = dx * (prod_y * scale +
 * shift > 0), xhat = (prod_y - mean) * rstd; scale / shift / mean / rstd are conv1's unetk_norm_finalize outputs,
 * [N][Cin] when per_sample else [Cin]) into partials [2][rows][Cin] -- the input unetk_norm_relu_bwd_pre takes instead of
 * running its own pass over (dz, y).  unetk_conv3x3_dgrad_nbr_rows = rows, or 0 when the shape has no fused variant
 * (tiled fp32 kernel and bf16-storage kernel only; the caller then uses the two separate calls). */
int unetk_conv3x3_dgrad_nbr_rows(const unetk_conv_desc* d);
int unetk_conv3x3_dgrad_nbr(const unetk_conv_desc* d, const void* dy, const void* w, void* dx, const void* prod_y,
                            int prod_y_stride, const float* scale, const float* shift, const float* mean,
                            const float* rstd, int per_sample, float* partials, void* stream);

/* dw[HWIO] = conv3x3_filter_grad(x, dy).  Split-K over pixel tiles through a workspace of
 * fixed-order partial slabs (bit-reproducible). */
size_t unetk_conv3x3_wgrad_ws_bytes(const unetk_conv_desc* d);
int unetk_conv3x3_wgrad(const unetk_conv_desc* d, const void* x, const void* dy, float* dw,
                        void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- slim.conv3d  (NetworksV2/UNet3D.py:153,165)
 * Kernels (1,3,3) / (3,3,3) from _ModelConfig (UNet3D.py:31-91), strides 1 / (1,2,2) / (2,2,2), TF SAME
 * padding (asymmetric at stride 2 on even sizes: 0 before, 1 after), no bias.
 * x [N,D,H,W,Cin] (pixel stride x_stride), y [N,Do,Ho,Wo,Cout] (pixel stride y_stride), Do = ceil(D/sd), ...
 * Filters: TF DHWIO [kd,3,3,Cin,Cout]; packed per depth tap like unetk_conv3x3_pack. */
typedef struct unetk_conv3d_desc {
  int32_t N, D, H, W, Cin, Cout;
  int32_t kd;   /* 1 or 3 */
  int32_t sd;   /* depth stride 1 or 2 (kd == 3 only) */
  int32_t shw;  /* H and W stride, 1 or 2 */
  int32_t x_stride, y_stride;
  /* Channel-padded nets (UNet3D's 30/60/120/240 real channels inside 32/64/128/256; ABI 10): bit i of the 64-bit mask
   * {lo, hi} = channels [8 i, 8 i + 8) of x (cin_live8) / of y (cout_live8) hold a real channel.  A promise by the caller that
   * the filter is ZERO in every other input row / output column: the forward then skips the dead groups of its contraction over
   * Cin, the input gradient those of its contraction over Cout (small-plane kernel; results equal to contracting the zeros).
   * 0 = no information, every channel is contracted. */
  uint32_t cin_live8[2], cout_live8[2];
} unetk_conv3d_desc;

int unetk_conv3d_pack(const float* w, int kd, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                      void* stream);
int unetk_conv3d_out_dims(const unetk_conv3d_desc* d, int* Do, int* Ho, int* Wo);
int unetk_conv3d_stat_rows(const unetk_conv3d_desc* d);      /* rows of the statistic partials */
size_t unetk_conv3d_ws_bytes(const unetk_conv3d_desc* d);    /* one size serves fwd / dgrad / wgrad */
int unetk_conv3d_fwd(const unetk_conv3d_desc* d, const float* x, const float* wp_fwd, float* y,
                     float* stat_partials, void* ws, size_t ws_bytes, void* stream);
int unetk_conv3d_dgrad(const unetk_conv3d_desc* d, const float* dy, const float* wp_dgrad, float* dx,
                       void* ws, size_t ws_bytes, void* stream);
int unetk_conv3d_wgrad(const unetk_conv3d_desc* d, const float* x, const float* dy, float* dw,
                       void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- normalisation + ReLU after each 3x3 conv
 * slim.batch_norm   NetworksV2/base.py:153-162 -- TF defaults eps 1e-3, decay .999 (GUNet encoder .99,
 *                   GUNet.py:321-325); training: batch mean + biased variance, unbiased variance into
 *                   the moving average; eval: moving statistics                        
 * slim.instance_norm base.py:163-165 -- eps 1e-6, moments over the spatial axes per (n, c)   (B4)
 * centre / scale are optional (gamma / beta may be NULL): GUNet.yml norm_with_center / norm_with_scale.
 * guide_ch > 0 adds GUNet's spatial modulation before the ReLU (GUNet.py:154-156,207-212):
 *   u += sum_g guide[n,pix,g] * gw[g][gw_coff + c] + gb[gw_coff + c]   (the 1x1 conv of the pooled guide,
 *   computed on the fly; gw is [guide_ch][gw_stride], gb is [gw_stride]). */
typedef struct unetk_norm_desc {
  int32_t N, HW, C;            /* activations [N, HW, C], dense */
  int32_t per_sample;          /* 0 = batch norm (one statistic group), 1 = instance norm (N groups) */
  int32_t z_stride;            /* pixel stride of the activated output (concat placement) */
  int32_t guide_ch, gw_stride, gw_coff;
  int32_t affine_only;         /* 1 = no normalisation (--without_norm, UNet.py:47-48: conv + bias + ReLU):
                                  backward skips the statistics terms, dbeta is the bias gradient */
  int32_t guide_leaky;         /* 3 = per-channel activation + post-shift on the guide branch (GUNet after_affine with --fix,
                                  GUNet.py:213-214,299-304): gb is a block of four rows of gw_stride floats -- bias, slope for
                                  s > 0, slope for s <= 0, post-shift -- and u = t * den + slope(s) * s + post-shift with
                                  s = guide . gw + bias; needs den; dgb comes back as the same block (slope rows zero);
                                  1 = LGNet's guide branch (LGNet.py:30-55): the 1x1 guide conv is followed by
                                  tf.nn.leaky_relu (alpha 0.2) before the add: u = t + lrelu(guide . gw + gb);
                                  needs guide_ch > 0, no density gains */
  int32_t storage;             /* UNETK_FP32: y / z / dz / dy are fp32; UNETK_BF16S: they are bf16 in memory (strides in
                                  elements), arithmetic, statistics and parameter gradients stay fp32 */
  float guide_alpha;           /* slope of the guide branch's activation when guide_leaky == 2 (guide_leaky == 1 is
                                  tf.nn.leaky_relu's default 0.2, LGNet): 0 = the ReLU of GUNet --fix (GUNet.py:299-304: the guide convs get norm + ReLU; the host
                                  folds that norm into gw / gb, exactly, from the guide's first and second moments) */
  float dropout_keep;          /* > 0: slim.dropout(keep_prob) on the NORMALISED value before the gains / guide term
                                  (GUNet --dropout, GUNet.py:189-190; training only); the 0 | 1/keep mask is regenerated
                                  from (dropout_seed, element index) in the forward and both backward passes.  With a guide
                                  or post-shift the backward needs the density variant (pass den = ones) */
  uint32_t dropout_seed;
  int32_t guide_per_sample;    /* 1: gw is [N][guide_ch][gw_stride], gb [N][gw_stride] (per-sample folded weights: --fix under
                                  instance norm) and dgw / dgb come back per sample; needs per_sample or den */
} unetk_norm_desc;

/* Finalise the conv's statistic partials ([2][stat_rows][C], each image's tiles contiguous) into
 * mean / rstd and the fused affine scale = gamma*rstd, shift = beta - mean*scale, each [groups][C].
 * Batch norm in training also updates the moving statistics; batch norm with training == 0 builds the
 * affine from the moving statistics and ignores the partials. */
size_t unetk_norm_finalize_ws_bytes(const unetk_norm_desc* d, int stat_rows);
int unetk_norm_finalize(const unetk_norm_desc* d, const float* stat_partials, int stat_rows,
                        const float* gamma, const float* beta, float eps, float decay, int training,
                        float* moving_mean, float* moving_var, float* mean_out, float* rstd_out,
                        float* scale_out, float* shift_out, void* ws, size_t ws_bytes, void* stream);

/* z = relu((y*scale + shift) [* den[n][c]] [+ guide modulation]); z has pixel stride d->z_stride.
 * den (nullable, [N][C]) is GUNet's density modulation `conditional_normalization` (GUNet.py:119-133,203-206):
 * the per-sample channel gains the context MLP (unetk_fc_*) produced.  gb may be given with guide_ch == 0: a bare
 * per-channel shift after the gain -- with den = gain * gamma' this is `after_affine` (slim_nets.channel_wise_affine,
 * GUNet.py:213-214: (net) * gamma' + beta') in the same pass. */
int unetk_norm_apply_relu(const unetk_norm_desc* d, const void* y, const float* scale,
                          const float* shift, const float* den, const float* guide, const float* gw,
                          const float* gb, void* z, void* stream);

/* Backward of z = relu(norm(y) [* den] [+ modulation]).  Pass 1: per-group column sums of dt and dt*xhat
 * (du = dz * (z > 0), dt = du * den) [and du*guide_g, du, du*t]; pass 2: dy.  dz has pixel stride dz_stride.
 * Outputs (nullable when the parameter does not exist): dgamma[C], dbeta[C], dden[N][C] (required with den),
 * dgw[guide_ch][C], dgb[C]. */
size_t unetk_norm_bwd_ws_bytes(const unetk_norm_desc* d);
int unetk_norm_relu_bwd(const unetk_norm_desc* d, const void* y, const void* dz, int dz_stride,
                        const float* scale, const float* shift, const float* mean, const float* rstd,
                        const float* den, const float* guide, const float* gw, const float* gb, void* dy,
                        float* dgamma, float* dbeta, float* dden, float* dgw, float* dgb, void* ws,
                        size_t ws_bytes, void* stream);

/* unetk_norm_relu_bwd with the reduction pass already done by the kernel that produced dz (unetk_conv3x3_dgrad_nbr):
 * pre_partials [2][pre_rows][C], each statistics group's rows contiguous; NULL = unetk_norm_relu_bwd.  Plain units only
 * (no guide, density, dropout or bias-only mode: UNETK_E_UNSUPPORTED). */
int unetk_norm_relu_bwd_pre(const unetk_norm_desc* d, const void* y, const void* dz, int dz_stride,
                            const float* scale, const float* shift, const float* mean, const float* rstd,
                            const float* den, const float* guide, const float* gw, const float* gb, void* dy,
                            float* dgamma, float* dbeta, float* dden, float* dgw, float* dgb,
                            const float* pre_partials, int pre_rows, void* ws, size_t ws_bytes, void* stream);

/* unetk_norm_apply_relu of a plain unit (no guide / density / dropout) + unetk_maxpool2_fwd of its activation in ONE pass
 * (NetworksV2/UNet.py:79-80: slim.conv2d ... slim.max_pool2d): z as unetk_norm_apply_relu writes it (pixel stride
 * d->z_stride), pooled [N, H/2, W/2, C] dense.  d->HW = H * W, W given here, both even. */
int unetk_norm_apply_relu_pool(const unetk_norm_desc* d, int W, const void* y, const float* scale, const float* shift,
                               void* z, void* pooled, void* stream);

/* unetk_norm_relu_bwd for a plain unit (no guide / density / dropout) whose activation z feeds max_pool2d(2, 2) AND the skip
 * connection (NetworksV2/UNet.py:80-81,93: the second conv of every encoder level): replaces unetk_maxpool2_bwd (with its
 * `add` operand) + unetk_norm_relu_bwd -- the gradient of z, dskip + route(dp), is formed on the fly in both passes from
 * dskip (the skip's gradient: pixel stride dskip_stride, a channel slice of the concat buffer's gradient), dp (the pooled
 * tensor's gradient, [N, H/2, W/2, C] dense) and z re-evaluated from y; first maximum in window scan order takes dp
 * (TF MaxPoolGrad).  d->HW = H * W, W given here, both even.  Workspace: unetk_norm_bwd_ws_bytes(d). */
int unetk_norm_relu_bwd_pool(const unetk_norm_desc* d, int W, const void* y, const void* dskip, int dskip_stride,
                             const void* dp, const float* scale, const float* shift, const float* mean,
                             const float* rstd, void* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                             void* stream);

/* GUNet --use_se (GUNet.py:192-201): the SE gate's input is pooled[b][c] = mean over the sample's pixels of the normalised
 * conv output, so the loss reaches y once more through it.  The norm backward is linear in dt, and this part of dt is the
 * per-(sample, channel) constant g[b][c] / HW: after unetk_norm_relu_bwd,  dy += scale * (A[b][c] - xhat * k2[group][c])
 * with A = g / HW - its mean over the statistics group and k2 = the group mean of (g / HW) * xhat (both [.][C], formed by
 * the caller from the per-sample means; zero under instance norm). */
int unetk_norm_se_bwd_add(const unetk_norm_desc* d, const void* y, void* dy, const float* mean,
                          const float* rstd, const float* scale, const float* A, const float* k2,
                          void* stream);

/* GUNet --use_se together with --dropout (NetworksV2/GUNet.py:189-201: the gate pools the DROPPED-OUT normalised output).
 * Forward: sums [2][N][C] = per (sample, channel) sum over the pixels of mask * xhat and of mask (xhat = (y - mean) rstd, the
 * 0 | 1/keep mask of d->dropout_keep / d->dropout_seed as the norm kernels regenerate it); pooled = gamma * sums[0] / HW +
 * beta * sums[1] / HW.  Backward: dy += scale * (mask * E[n][c] - k1 - xhat * k2) with E = d loss / d pooled / HW [N][C] and
 * k1, k2 [groups][C] the statistics group's means of mask E and mask E xhat (groups = N under instance norm, else 1). */
int unetk_norm_drop_pool(const unetk_norm_desc* d, const void* y, const float* mean, const float* rstd, float* sums, void* stream);
int unetk_norm_se_bwd_add_drop(const unetk_norm_desc* d, const void* y, void* dy, const float* mean, const float* rstd,
                               const float* scale, const float* E, const float* k1, const float* k2, void* stream);

/* ---------------------------------------------------------------- GUNet's context MLP (GUNet.py:136-150 `_context_subnets`)
 * slim.fully_connected(x, n): y[B][n] = act(x[B][k] . w[k][n] + b[n]), TF weight layout [in, out]; relu = 1 for the
 * hidden layers, 0 for the last (activation_fn=None), 2 = tf.nn.sigmoid (the SE gate of GUNet --use_se, GUNet.py:199).  slim.dropout(keep_prob) in training: mask from a counter
 * RNG keyed by (seed, element), kept entries scaled by 1/keep_prob; `mask` ([B][n] floats, 0 or 1/keep_prob,
 * nullable = no dropout) is written by the forward and read by the backward.
 * Backward: dx[B][k] (nullable), dw[k][n], db[n] from dy[B][n] (gated by y > 0 when relu, times the mask). */
int unetk_fc_fwd(const float* x, const float* w, const float* b, float* y, float* mask, int B, int k, int n,
                 int relu, float keep_prob, uint32_t seed, void* stream);
int unetk_fc_bwd(const float* x, const float* w, const float* y, const float* mask, const float* dy, float* dx,
                 float* dw, float* db, float* dpre_ws, int B, int k, int n, int relu, void* stream);

/* ---------------------------------------------------------------- GUNet's 1-D VGG context models (slim_nets.py:60-144)
 * context_model "vgg16B" / "vgg16C" / "vgg16D" (GUNet.py:62-75; ext_config/GUNet_DE_VGG16{B,D}.yml): the context vector as
 * [B][L][1] through slim.conv1d (kernel k = 3 or 1, stride 1, SAME, bias, ReLU; w = TF [k][Cin][Cout]) and
 * tf.layers.max_pooling1d(2, 2, "same") (Lo = ceil(L / 2)); flattened [B][Lo * C] it feeds unetk_fc_*.
 * Backward: dx (nullable) / dw / db from dy gated by y > 0 when relu; dpre_ws = B * L * Cout floats of scratch.  The pool
 * backward routes to the first maximum of the window. */
int unetk_conv1d_fwd(const float* x, const float* w, const float* b, float* y, int B, int L, int Cin, int Cout, int k,
                     int relu, void* stream);
int unetk_conv1d_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db,
                     float* dpre_ws, int B, int L, int Cin, int Cout, int k, int relu, void* stream);
int unetk_maxpool1d_fwd(const float* x, float* y, int B, int L, int C, void* stream);
int unetk_maxpool1d_bwd(const float* x, const float* dy, float* dx, int B, int L, int C, void* stream);

/* tf.reduce_mean(x, axis=(1, 2)) of GUNet's conv context subnet (`_context_subnets_conv`, GUNet.py:83-116): x [N][HW][C] ->
 * y [N][C]; backward dx = dy / HW broadcast over the pixels. */
int unetk_spatial_mean_fwd(const float* x, float* y, int N, int64_t HW, int C, void* stream);
int unetk_spatial_mean_bwd(const float* dy, float* dx, int N, int64_t HW, int C, void* stream);

/* ---------------------------------------------------------------- slim.max_pool2d(x, [2,2])  UNet.py:81
 * VALID, stride 2.  x [N,H,W,C] with pixel stride x_stride; p dense [N,H/2,W/2,C].
 * Backward routes dp to the first maximum in window scan order (TF MaxPoolGrad); `add` (nullable, [N,H,W,C] with
 * pixel stride add_stride) is summed into dx: x's other consumer is the skip connection (UNet.py:93), whose gradient
 * is the first half of the concat buffer's gradient -- one pass instead of a pool backward plus an add. */
int unetk_maxpool2_fwd(const float* x, int x_stride, float* p, int N, int H, int W, int C,
                       void* stream);
int unetk_maxpool2_bwd(const float* x, int x_stride, const float* p, const float* dp, const float* add,
                       int add_stride, float* dx, int N, int H, int W, int C, void* stream);
/* UNETK_BF16S variants: x, p, dp, add, dx are bf16 (strides in elements); the skip gradient is summed in fp32 and
 * rounded once. */
int unetk_maxpool2_fwd_bf16(const void* x, int x_stride, void* p, int N, int H, int W, int C, void* stream);
int unetk_maxpool2_bwd_bf16(const void* x, int x_stride, const void* p, const void* dp, const void* add,
                            int add_stride, void* dx, int N, int H, int W, int C, void* stream);
/* slim.avg_pool2d(gs, 2) of GUNet's spatial-guide pyramid (GUNet.py:157-158); x, p dense, any C. */
int unetk_avgpool2_fwd(const float* x, float* p, int N, int H, int W, int C, void* stream);
/* Moments of the (pooled) spatial guide [N, HW, G] per statistics group (1 group, or N with per_sample):
 * out[grp][0..G) = E[g_i], out[grp][G + i G + j] = E[g_i g_j].  GUNet --fix normalises the guide's 1x1 conv
 * (GUNet.py:299-304); that conv being linear in the guide, its statistics follow exactly from these numbers. */
int unetk_guide_moments(const float* guide, int N, int64_t HW, int G, int per_sample, float* out, void* stream);
/* --img_grad (UNet.py:69-71, GUNet.py:335-338): out [N,H,W,3C] = concat(x, dy, dx) with
 * tf.image.image_gradients' forward differences (dy[h] = x[h+1] - x[h], last row 0; dx likewise). */
/* InterUNet's --img_grad input (InterUNet.py:105-109): out [N,H,W,C+2] = concat(x, sobel_dy(x[..., ch]), sobel_dx(x[..., ch]))
 * with tf.image.sobel_edges' kernels over the REFLECT-padded channel. */
int unetk_sobel_concat(const float* x, float* out, int N, int H, int W, int C, int ch, void* stream);
int unetk_image_gradients(const float* x, float* out, int N, int H, int W, int C, void* stream);
/* Mirror test-time augmentation (evaluators/evaluator_liver.py:648-655; the reference flips on the host with
 * np.flip): out[n,h,w,:] (+)= scale * x[n, flip_h ? H-1-h : h, flip_w ? W-1-w : w, :].  x != out. */
int unetk_flip_axpy(const float* x, float* out, int N, int H, int W, int C, int flip_h, int flip_w,
                    float scale, int accumulate, void* stream);

/* ---------------------------------------------------------------- slim.conv2d_transpose(x, C, 2, 2)
 * UNet.py:91-93: kernel 2 stride 2, bias, ReLU, then tf.concat((skip, up), -1).
 * out[n,2y+a,2x+b, out_coff+co] = relu(sum_ci x[n,y,x,ci]*w[a,b,co,ci] + bias[co]),
 * written straight into the concat buffer (pixel stride out_stride). */
typedef struct unetk_deconv_desc {
  int32_t N, H, W, Cin, Cout; /* input geometry; output is [N,2H,2W,Cout] */
  int32_t out_stride, out_coff;
  int32_t precision; /* UNETK_FP32 / UNETK_BF16 (Cin % 32 == 0 and Cout % 32 == 0; filters from *_pack_bf16) /
                        UNETK_BF16S (x, out / cat, dcat, dx are bf16; Cin % 64 == 0 and Cout % 32 == 0) */
} unetk_deconv_desc;

/* wp_fwd: [Cin/4][4*Cout][4]; wp_dgrad: [4*Cout/4][Cin][4]; each 4*Cin*Cout floats. */
int unetk_deconv2x2_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                         void* stream);
/* UNETK_BF16: wp_fwd [Cin/8][4*Cout][8], wp_dgrad [4*Cout/8][Cin][8], each 4*Cin*Cout bf16. */
int unetk_deconv2x2_pack_bf16(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                              void* stream);
/* UNETK_BF16S: the bf16 packs with the GEMM columns permuted inside every 64-column block (see
 * unetk_conv3x3_pack_bf16s).  Cin % 64 == 0 and Cout % 64 == 0. */
int unetk_deconv2x2_pack_bf16s(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                               void* stream);
int unetk_deconv2x2_fwd(const unetk_deconv_desc* d, const void* x, const void* wp_fwd,
                        const float* bias, void* out, void* stream);
/* Backward.  dcat/cat: gradient and forward value of the concat buffer (same strides/offset as
 * `out`).  Produces dx [N,H,W,Cin], dw [2,2,Cout,Cin], dbias [Cout]. */
size_t unetk_deconv2x2_bwd_ws_bytes(const unetk_deconv_desc* d);
int unetk_deconv2x2_bwd(const unetk_deconv_desc* d, const void* x, const void* wp_dgrad,
                        const void* cat, const void* dcat, void* dx, float* dw, float* dbias,
                        void* ws, size_t ws_bytes, void* stream);

/* slim.conv3d_transpose(x, c, kernel == stride in {(1,2,2), (2,2,2)}, biases_initializer=None) + tf.concat,
 * NetworksV2/UNet3D.py:161-163.  x [N,D,H,W,Cin]; out [N,kd*D,2H,2W,...]; filter TF [kd,2,2,Cout,Cin];
 * bias / dbias may be NULL (UNet3D has none).  The 2-D functions above are the D = 1, kd = 1 case. */
typedef struct unetk_deconv3d_desc {
  int32_t N, D, H, W, Cin, Cout;
  int32_t kd; /* depth kernel == depth stride: 1 or 2 */
  int32_t out_stride, out_coff;
  int32_t precision; /* UNETK_FP32 / UNETK_BF16 / UNETK_BF16S (as unetk_deconv_desc) */
} unetk_deconv3d_desc;
int unetk_deconv3d_pack(const float* w, int kd, int Cin, int Cout, float* wp_fwd, float* wp_dgrad,
                        void* stream);
int unetk_deconv3d_pack_bf16(const float* w, int kd, int Cin, int Cout, void* wp_fwd, void* wp_dgrad,
                             void* stream);
int unetk_deconv3d_fwd(const unetk_deconv3d_desc* d, const void* x, const void* wp_fwd,
                       const float* bias, void* out, void* stream);
size_t unetk_deconv3d_bwd_ws_bytes(const unetk_deconv3d_desc* d);
int unetk_deconv3d_bwd(const unetk_deconv3d_desc* d, const void* x, const void* wp_dgrad,
                       const void* cat, const void* dcat, void* dx, float* dw, float* dbias,
                       void* ws, size_t ws_bytes, void* stream);
/* The same in two parts (ABI 9): parts bit 0 = ReLU backward + dbias + dx (leaves the masked, re-laid gradient in ws), bit 1 = dw
 * (reads it from the SAME ws; any stream ordered behind the bit-0 call).  The filter gradient is off the critical chain of the
 * backward pass (TF schedules Conv2DBackpropFilter beside the rest of the graph the same way); parts = 3 is unetk_deconv3d_bwd. */
int unetk_deconv3d_bwd_parts(const unetk_deconv3d_desc* d, const void* x, const void* wp_dgrad,
                             const void* cat, const void* dcat, void* dx, float* dw, float* dbias,
                             void* ws, size_t ws_bytes, int parts, void* stream);
int unetk_deconv2x2_bwd_parts(const unetk_deconv_desc* d, const void* x, const void* wp_dgrad,
                              const void* cat, const void* dcat, void* dx, float* dw, float* dbias,
                              void* ws, size_t ws_bytes, int parts, void* stream);

/* ---------------------------------------------------------------- logits + loss head
 * UNet.py:97-135 + loss_metrics.py:115-231,261-339.
 * logits = z @ w[C,ncls] + b  (slim.conv2d(x, ncls, 1), linear);  softmax;  weighted sparse
 * softmax cross-entropy with SUM_BY_NONZERO_WEIGHTS and/or soft Dice loss;  thresholded
 * (prob > 0.5) per-class counts for metric_dice / metric_voe / metric_vd. */
typedef struct unetk_head_desc {
  int32_t N, HW, C, ncls;
  int32_t weight_mode;                    /* UNETK_W_* */
  float numeric_w[UNETK_MAX_CLASSES];     /* --loss_numeric_w */
  float proportion_decay;                 /* --loss_proportion_decay (<=0: none) */
  int32_t storage;                        /* UNETK_FP32, or UNETK_BF16S: the feature map z and its gradient dz are bf16
                                             (logits, probabilities, losses and dw / db stay fp32) */
} unetk_head_desc;

/* Reduced outputs of the forward pass (device, floats), layout:
 *   [0] xent loss  [1] dice loss  [2] num_present
 *   then per sample b, per foreground class c=1..ncls-1 (index 3 + (b*(ncls-1)+(c-1))*4 + k):
 *     k=0 sum(pred*lab) k=1 sum(pred) k=2 sum(lab) k=3 sum(clip(pred+lab,0,1))
 *   then per sample b: I_b, U_b  (soft dice intersection / union, loss_metrics.py:217-218) */
size_t unetk_head_result_floats(const unetk_head_desc* d);
size_t unetk_head_ws_bytes(const unetk_head_desc* d);
/* logits [N*HW, ncls] always written; probs (same shape) optional; pixel_w only for PIXELMAP. */
int unetk_head_fwd(const unetk_head_desc* d, const void* z, const float* w, const float* b,
                   const int32_t* labels, const float* pixel_w, float* logits, float* probs,
                   float* result, void* ws, size_t ws_bytes, void* stream);
/* Backward of (xent_scale * xent + dice_scale * dice) w.r.t. z, w, b.  `result` and `ws` are
 * the buffers the forward filled (ws keeps the per-sample weight tables).  dev_scales (nullable)
 * points at two device floats multiplied into xent_scale / dice_scale (upstream gradients that
 * live on the device -- avoids a host sync). */
int unetk_head_bwd(const unetk_head_desc* d, const void* z, const float* w,
                   const int32_t* labels, const float* pixel_w, const float* logits,
                   const float* result, float xent_scale, float dice_scale,
                   const float* dev_scales, void* dz, float* dw, float* db, void* ws,
                   size_t ws_bytes, void* stream);
/* Inference helpers: evaluators/evaluator_liver.py:663 np.argmax(prob, -1) (lowest index on ties)
 * and UNet.py:112-118 Pred_c = prob_c > 0.5 (uint8).  preds is [ncls-1][npix] or NULL. */
int unetk_head_predict(const float* probs, int64_t npix, int ncls, uint8_t* argmax,
                       uint8_t* preds, void* stream);

/* --loss_weight_type boundary (loss_metrics.py:149-165; 2-D only, as in the reference): labels int32
 * [N,H,W] -> wmap f32 [N,H,W] = normalised exp(-EDT/25) + 1, EDT = exact Euclidean distance to the nearest
 * pixel of the 3x3 dilation ring of any class (the reference round-trips to scipy on the host through
 * tf.py_func; this runs on the device).  Feed wmap to unetk_head_fwd/bwd as UNETK_W_PIXELMAP. */
size_t unetk_boundary_weights_ws_bytes(int N, int H, int W);
int unetk_boundary_weights(const int32_t* labels, int N, int H, int W, float* wmap, void* ws,
                           size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- LiTS training batch  
 * DataLoader/Liver/input_pipeline.py:243-284 `data_processing_train` for a whole batch, gathering from decoded slices
 * that are RESIDENT in device memory: per sample crop_to_bounding_box -> resize_bilinear(align_corners) -> window clip
 * -> normalise (images), crop -> resize_nearest_neighbor(align_corners) -> / lab_scale (labels), uniform noise
 * U(-noise_scale, noise_scale) on non-padding slices (counter-based generator keyed by `seed`), random flips.
 * slices: uint16 [n_slices, src_h, src_w]; seg_slices: uint8 [n_slices, src_h, src_w];
 * sample_tab int32 [N][C + 7] = {slice index x C (-1 = zero padding), label slice index (-1 = zeros), off_y, off_x,
 * crop_h, crop_w, flip_left_right, flip_up_down}; clip float [N][2] = {lo, hi};
 * images f32 [N,H,W,C]; labels int32 [N,H,W].  The caller guarantees crops inside the source slice. */
typedef struct unetk_lits_desc {
  int32_t N, H, W, C;
  int32_t n_slices, src_h, src_w; /* extent of the resident store; indices outside [0, n_slices) read as zeros */
  int32_t lab_scale; /* LB_SCALE = 64 */
  uint32_t seed;
  float noise_scale; /* 0 = no noise (eval_online) */
} unetk_lits_desc;
/* PNG row un-filtering for the resident slice store (the reference decodes with cv2 on tf.data threads,
 * DataLoader/Liver/input_pipeline.py:243-284; its PNGs come from SimpleITK / libpng with adaptive row filters,
 * DataLoader/Liver/extract.py:176-187).  filtered: n_images inflated IDAT streams of non-interlaced grayscale PNGs, image k at
 * filtered + k * image_stride_bytes, each h rows of (1 filter-type byte + w * bit_depth / 8 bytes).  out: image k at element
 * k * out_image_stride, h * w samples, uint8 (bit_depth 8) or native-endian uint16 (bit_depth 16).  All five filter types
 * (None, Sub, Up, Average, Paeth).  status[0] |= 1 if a row carries a filter type > 4 (the caller zeroes it and raises). */
int unetk_png_unfilter(const uint8_t* filtered, int64_t image_stride_bytes, int n_images, int h, int w, int bit_depth,
                       void* out, int64_t out_image_stride, int32_t* status, void* stream);
int unetk_lits_batch(const unetk_lits_desc* d, const uint16_t* slices, const uint8_t* seg_slices,
                     const int32_t* sample_tab, const float* clip, float* images, int32_t* labels,
                     void* stream);

/* ---------------------------------------------------------------- optimiser  core/solver.py:204-243
 * tf.train.AdamOptimizer on a flat parameter buffer.  g' = g*gscale + l2*p  (slim.l2_regularizer
 * gradient, base.py:128-135);  m += (1-b1)(g'-m);  v += (1-b2)(g'^2-v);
 * p -= lr_t * m / (sqrt(v) + eps)  with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller.
 * decoupled_wd > 0 = tf.contrib.opt.AdamWOptimizer (solver.py:212-216): p <- p*(1 - wd) first. */
int unetk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t,
                    float beta1, float beta2, float eps, float gscale, float l2, float decoupled_wd,
                    void* stream);
/* tf.train.MomentumOptimizer: acc = mom*acc + g'; p -= lr*acc (nesterov: lr*(g' + mom*acc)). */
int unetk_momentum_step(float* p, const float* g, float* acc, int64_t n, float lr, float mom,
                        int nesterov, float gscale, float l2, void* stream);
/* out[0] = sum(p^2) (fp64 accumulate) -- for the reported regularisation loss. ws >= 8 KiB. */
int unetk_sumsq(const float* p, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream);

/* tf.train.NanTensorHook(loss) (core/estimator.py:676: every step, raises NanLossDuringTrainingError) without a host sync
 * per step: if value[0] is NaN and flag[0] == 0, set flag[0] = 1 and flag[1] = step (sticky; the caller zeroes flag[0..1]
 * once and reads it whenever it synchronises anyway: log steps, and before every checkpoint it writes). */
int unetk_nan_watch(const float* value, int32_t* flag, int32_t step, void* stream);

/* ---------------------------------------------------------------- kernel trace (measurement only; bench.py)
 * SURVEY.md 8d asks for the dominant kernel's launch duration "measured live inside bench.py with HIP events ... on the
 * stream the kernel is launched on".  The reference has no counterpart (its profiling is tf.train.ProfilerHook,
 * core/estimator.py:688-690).  While enabled, EVERY kernel this library launches carries a start / stop event pair bound to
 * the dispatch (hipExtLaunchKernelGGL), i.e. the GPU's own begin -> end interval of that kernel -- what rocprofv3
 * --kernel-trace reports -- independent of host gaps around the launch.  Process-wide diagnostic state, off by default; the
 * only exception to the "no global mutable state" rule above.  No call here synchronises.
 *   unetk_prof_reset(n)     forget all records; pre-create events for n launches (outside any timed region)
 *   unetk_prof_enable(on)   start / stop recording (records accumulate across enable periods until the next reset)
 *   unetk_prof_mark()       number of launches recorded so far: a caller brackets one ABI call with two marks
 *   unetk_prof_read(...)    durations in ms of records [first, first + count); the stream must have been synchronised
 *                           (else hipErrorNotReady is returned)
 *   unetk_prof_name(i,...)  demangled kernel name of record i, as rocprofv3 prints it */
int unetk_prof_reset(int reserve_launches);
int unetk_prof_enable(int on);
int unetk_prof_mark(void);
int unetk_prof_read(int first, int count, float* ms_out);
int unetk_prof_name(int i, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* UNETK_H_ */
