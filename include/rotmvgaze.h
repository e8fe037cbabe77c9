/*
 * rotmvgaze.h - C ABI of librotmvgaze_hip.so, the MI355X (gfx950) compute library behind the
 * Rot-MVGaze hot path (FeatRotationSymm forward / loss / backward).
 *
 * The reference has NO native layer and no FFI: every FLOP is a stock PyTorch ATen op
 * (SURVEY.md §2).  Each entry point below therefore replaces the ATen op(s) that the cited
 * reference lines dispatch; INTEGRATION.md shows the ctypes stub that binds them and how the
 * drop-in nn.Module calls them.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - activations are NHWC, images ordered [group][n][h][w][c]; a "group" is one view's batch:
 *     BatchNorm statistics are reduced per group because the reference runs the backbone once
 *     per view (rot_mv.py:196-197).  Storage is fp32 by default; the *_split entry points read
 *     their operands in "sp" (every fp32 value as two fp16 pieces and a per-tensor power-of-two
 *     scale: 8-channel chunks, the two pieces of a chunk adjacent, 4 bytes per element - section
 *     "split operands") and the
 *     *_bf16 entry points keep activations and activation gradients in bf16 (config C5);
 *   - conv weights are KRSC fp32 ([cout][r][s][cin]) = the physical layout of a PyTorch
 *     [O,I,H,W] tensor in torch.channels_last; linear weights are [out][in] (PyTorch native);
 *   - the caller owns every buffer, including workspaces; the library never allocates device
 *     memory and never synchronises (except mvg_prof_collect).  Kernel sequences that want
 *     scratch of their own (stream-K pieces, two-level statistics) take it from the workspace
 *     the caller registered for the stream with mvg_set_scratch, and run their scratch-free
 *     form when there is none;
 *   - every launch goes to the caller's hipStream_t (passed as void*; NULL = default stream);
 *   - return 0 on success, non-zero on error; mvg_last_error() gives the message. No exceptions
 *     cross the boundary.
 */
#ifndef ROTMVGAZE_H
#define ROTMVGAZE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVG_ABI_VERSION 10

/* ---------------------------------------------------------------- library */
int mvg_abi_version(void);
const char *mvg_last_error(void);
/* number of CUs of the current device (used by the host to size split-K). */
int mvg_device_cus(void);
/* Register (ptr != NULL) or drop (ptr == NULL) the caller-owned workspace that launches on `stream` of the
 * current device may use as scratch; 16-byte aligned, valid until replaced or dropped.  Uses are ordered by the
 * stream, so one workspace per stream suffices.  mvg_scratch_bytes(): a size that covers every entry point on
 * the current device.  Without a workspace (or with a smaller one) the affected launches run their
 * scratch-free forms: correct, a few percent slower on the stem / fusion-block GEMMs. */
int mvg_set_scratch(void *ptr, size_t bytes, void *stream);
size_t mvg_scratch_bytes(void);
/* A HIP stream of the device's LOWEST priority (hipStreamNonBlocking) for work that should only fill
 * what the caller's stream leaves idle - the backward-weight kernels, which are off the critical
 * path of backward.  NULL on failure.  The stream lives until process exit. */
void *mvg_stream_create_low_priority(void);
/* Data-parallel runs: leave n CUs out of the stream-K grids / wgrad and split-K plans so that the
 * RCCL kernels of the gradient all-reduce (side stream) find room next to the persistent conv
 * kernels instead of pushing part of their grid into a second round.  0 = use every CU. */
int mvg_set_reserved_cus(int n);

/* ---------------------------------------------------------------- profiling (bench.py)
 * When enabled, every launch is bracketed by hipEvents on its own stream; mvg_prof_collect
 * synchronises those events and returns, per kernel family, launches, summed milliseconds,
 * algorithmic FLOPs and algorithmic bytes since mvg_prof_reset. */
enum {
  MVG_K_CONV_FPROP = 0, MVG_K_CONV_DGRAD = 1, MVG_K_CONV_WGRAD = 2, MVG_K_WGRAD_REDUCE = 3,
  MVG_K_BN_FINALIZE = 4, MVG_K_BN_APPLY = 5, MVG_K_BN_BWD_REDUCE = 6, MVG_K_BN_BWD_APPLY = 7,
  MVG_K_POOL = 8, MVG_K_LAYOUT = 9, MVG_K_LINEAR_FPROP = 10, MVG_K_LINEAR_DGRAD = 11,
  MVG_K_LINEAR_WGRAD = 12, MVG_K_ROTCAT = 13, MVG_K_COLSUM = 14, MVG_K_LOSS = 15,
  MVG_K_GEOMETRY = 16, MVG_K_ELEMENTWISE = 17, MVG_K_FAMILIES = 18
};
typedef struct {
  int64_t launches;
  double ms;
  double flops;
  double bytes;
} mvg_prof_entry;
int mvg_prof_enable(int on);
int mvg_prof_reset(void);
int mvg_prof_collect(mvg_prof_entry *host_out /* [MVG_K_FAMILIES] */);
const char *mvg_prof_family_name(int family);

/* ---------------------------------------------------------------- convolution
 * Replaces nn.Conv2d(bias=False) forward/backward: resnet.py:31-47 (conv3x3/conv1x1), :184
 * (7x7 stem) as used by BasicBlock.forward :80-96 and Bottleneck.forward :128-148, and - with
 * h = w = r = s = 1 - nn.Linear inside Mlp (backbones/blocks.py:41-47,57-60). */
typedef struct {
  int32_t groups;      /* view groups; total images = groups * n */
  int32_t n;           /* images per group */
  int32_t h, w;        /* input spatial size */
  int32_t cin, cout;   /* cin % 4 == 0, cout % 4 == 0 (the stem's 3 channels are padded to 4) */
  int32_t r, s;        /* filter size */
  int32_t stride, pad;
  int32_t ho, wo;      /* output spatial size */
} mvg_conv_desc;

/* y[g][n][ho][wo][cout] = conv(x[g][n][h][w][cin], wgt[cout][r][s][cin]).
 * Epilogue (all optional, NULL = off):
 *   bias[cout]   added to every row, then relu != 0 clamps at 0            (Linear [+ReLU]);
 *   stats        per-(group, 32|64-row partial, channel) {sum, centred sum of squares} of y,
 *                the input of mvg_bn_finalize; needs mvg_conv_stats_partials() * cout * 2 floats
 *                per group.                                                  (conv -> BatchNorm) */
int mvg_conv_fprop(const mvg_conv_desc *d, const float *x, const float *wgt, float *y,
                   const float *bias, int relu, float *stats, void *stream);
/* Inference (model.eval(), trainer.py:164-199): BatchNorm uses its running statistics, so it folds
 * into the conv epilogue: out = [relu]( conv(x, wgt) * scale[cout] + shift[cout] [+ residual] ) in one
 * launch (scale/shift from mvg_bn_eval_affine) - no separate normalisation pass, no raw conv output. */
int mvg_conv_fprop_affine(const mvg_conv_desc *d, const float *x, const float *wgt, float *out,
                          const float *scale, const float *shift, const float *residual, int relu,
                          void *stream);
/* number of row-partials per group that mvg_conv_fprop writes into `stats`, and the rows per
 * partial (out_rows_per_partial, may be NULL). */
int mvg_conv_stats_partials(const mvg_conv_desc *d, int32_t *out_rows_per_partial);

/* dx[g][n][h][w][cin] = conv_transpose(dy[g][n][ho][wo][cout], wgt).  Epilogue (optional):
 *   mask   (same shape as dx): result multiplied by (mask > 0)   (ReLU backward of the producer);
 *   addend (same shape as dx, may alias dx): added after masking (residual / fan-in accumulation). */
int mvg_conv_dgrad(const mvg_conv_desc *d, const float *dy, const float *wgt, float *dx,
                   const float *mask, const float *addend, void *stream);

/* dw[cout][r][s][cin] (+)= sum over all groups/images/pixels of dy (x) x.  Split over the pixel
 * axis into `splits` slabs in `workspace` (splits * cout*r*s*cin floats, ignored when
 * splits == 1) that a second kernel sums in a fixed order (bitwise reproducible).
 * accumulate != 0 adds to the existing dw. */
int mvg_conv_wgrad(const mvg_conv_desc *d, const float *x, const float *dy, float *dw,
                   float *workspace, int splits, int accumulate, void *stream);
/* a split count that fills the device for this shape (host helper, no launch). */
int mvg_conv_wgrad_splits(const mvg_conv_desc *d);
/* The cross-view fusion GEMM with its input generated inside the kernel (rot_mv.py:44-50 ImageFeatFuser
 * first layer on cat([img_feat_i, (R_ij @ F_j).flatten]), :234-239; and the gaze head's first layer on
 * cat([img_feat_i, F_i.flatten]), :249-254, with rel = NULL):
 *   y[m] = [relu]( W @ [ img_feat[row_img[m]] (cf) | rel[m] (3x3) @ feat[row_src[m]] (3 x nvec, axis-major) ] + bias )
 * for m < rows.  The concatenated / rotated row is built by the GEMM's operand loader (three loads and
 * three fmas per 16 bytes of the rotated part) - it is never written to memory; mvg_fuser_wgrad is the
 * backward-weight twin (dw += dy^T X, db += column sums of dy) with X generated the same way.
 * img_feat [img_rows][cf], feat [feat_rows][3*nvec], rel [rows][9] or NULL (identity), W [fout][cf + 3*nvec].
 * Workspaces like mvg_linear_fprop / mvg_linear_wgrad (fin = cf + 3*nvec). */
int mvg_fuser_fprop(const float *img_feat, const float *feat, const float *rel, const int32_t *row_img,
                    const int32_t *row_src, const float *w, const float *bias, int relu, float *y, int rows, int cf,
                    int nvec, int fout, int img_rows, int feat_rows, float *workspace, size_t ws_floats, void *stream);
int mvg_fuser_wgrad(const float *img_feat, const float *feat, const float *rel, const int32_t *row_img,
                    const int32_t *row_src, const float *dy, float *dw, float *db, int rows, int cf, int nvec, int fout,
                    int img_rows, int feat_rows, float *workspace, int splits, int accumulate, void *stream);

/* nn.Linear backward-weight (blocks.py:41-47 under autograd): dw[fout][fin] (+)= dy^T x and, in the SAME
 * launch, the bias gradient db[fout] (+)= column sums of dy (the kernel streams dy anyway; db may be NULL).
 * splits = mvg_conv_wgrad_splits(linear descriptor); workspace: splits * (fout*fin + fout) floats when
 * splits > 1. */
int mvg_linear_wgrad(const float *x, const float *dy, float *dw, float *db, int rows, int fin, int fout,
                     float *workspace, int splits, int accumulate, void *stream);

/* nn.Linear forward / backward-data (backbones/blocks.py:41-47 inside Mlp): the conv kernels with
 * h = w = r = s = 1 plus split-K - with a few hundred rows the tile grid cannot fill 256 CUs, so
 * K is cut into slices whose partial tiles land in `workspace` (mvg_linear_workspace_floats()
 * floats) and are summed, in fixed order, by a reduce kernel that applies the epilogue.
 *   fprop: y = [relu](x @ w^T + bias);   dgrad: dx = (dy @ w) * (mask > 0) + addend. */
size_t mvg_linear_workspace_floats(int rows, int fin, int fout);
int mvg_linear_fprop(const float *x, const float *w, const float *bias, int relu, float *y, int rows,
                     int fin, int fout, float *workspace, size_t ws_floats, void *stream);
int mvg_linear_dgrad(const float *dy, const float *w, const float *mask, const float *addend, float *dx,
                     int rows, int fin, int fout, float *workspace, size_t ws_floats, void *stream);

/* ---------------------------------------------------------------- BatchNorm2d (train + eval)
 * Replaces nn.BatchNorm2d (eps 1e-5, momentum 0.1, affine, track_running_stats) as constructed
 * at resnet.py:185,72-76,119-125 and the ReLU / residual add around it (:74,93-94,140-146). */

/* Merge the conv epilogue's partials into per-(group, channel) batch statistics, produce the
 * fused affine (scale = gamma*invstd, shift = beta - mean*scale) and update the running stats
 * once per group IN GROUP ORDER (view 0 first: rot_mv.py:196-197) with the unbiased variance.
 *   stats        [groups][partials][2][c]     rows_per_partial, rows_per_group as given
 *   mean,invstd  [groups][c] (saved for backward), scale, shift [groups][c]
 *   running_mean/var [c] updated in place unless NULL; num_batches_tracked is host-side. */
int mvg_bn_finalize(const float *stats, int groups, int partials, int rows_per_partial,
                    int64_t rows_per_group, int c, const float *gamma, const float *beta,
                    float *running_mean, float *running_var, float momentum, float eps,
                    float *mean, float *invstd, float *scale, float *shift, void *stream);
/* eval mode: scale/shift from the running statistics (same for every group). */
int mvg_bn_eval_affine(int groups, int c, const float *gamma, const float *beta,
                       const float *running_mean, const float *running_var, float eps,
                       float *scale, float *shift, void *stream);
/* out = [relu]( y*scale[g] + shift[g] [+ residual] );  y,out,residual: [groups][rows][c].
 * res_scale/res_shift ([groups][c], both or neither): the residual is the RAW output of the block's
 * downsample conv and its BatchNorm (residual*res_scale + res_shift) is applied here, so the normalised
 * downsample map is never written (resnet.py:88-93,137-145). */
int mvg_bn_apply(const float *y, const float *scale, const float *shift, const float *residual,
                 const float *res_scale, const float *res_shift, int relu, float *out, int groups,
                 int64_t rows_per_group, int c, void *stream);
/* mvg_bn_apply for a residual unit (relu = 1) that also records the ReLU mask: relu_bits holds one byte per
 * 16-byte access of `out` (bit k = element k > 0; groups*rows*c/4 bytes in fp32, /8 in bf16).
 * mvg_bn_bwd_reduce_bits takes the mask from those bytes instead of `act`: 1/16 of the bytes of reading the
 * activation again (resnet.py:93-94,145-146 under autograd: the ReLU after the residual add). */
int mvg_bn_apply_bits(const float *y, const float *scale, const float *shift, const float *residual,
                      const float *res_scale, const float *res_shift, float *out, uint8_t *relu_bits, int groups,
                      int64_t rows_per_group, int c, void *stream);
int mvg_bn_bwd_reduce_bits(const float *g, const uint8_t *relu_bits, const float *y, const float *mean,
                           const float *invstd, int groups, int64_t rows_per_group, int c, float *s1, float *s2,
                           float *dgamma, float *dbeta, int accumulate, float *workspace, float *dz_out, void *stream);
int mvg_bn_apply_bits_bf16(const uint16_t *y, const float *scale, const float *shift, const uint16_t *residual,
                           const float *res_scale, const float *res_shift, uint16_t *out, uint8_t *relu_bits,
                           int groups, int64_t rows_per_group, int c, void *stream);
int mvg_bn_bwd_reduce_bits_bf16(const uint16_t *g, const uint8_t *relu_bits, const uint16_t *y, const float *mean,
                                const float *invstd, int groups, int64_t rows_per_group, int c, float *s1,
                                float *s2, float *dgamma, float *dbeta, int accumulate, float *workspace,
                                uint16_t *dz_out, void *stream);
/* Backward, step 1: per (group, channel) s1 = sum(dz), s2 = sum(dz * xhat), xhat = (y - mean) * invstd,
 * dz = g masked by the unit's ReLU.  The mask comes from `act` (the unit's output: act > 0) or, for
 * a ReLU without residual, from (relu_scale, relu_shift) = the scale/shift mvg_bn_apply used:
 * fma(y, scale, shift) > 0 is the same bit pattern and saves reading act.  Both NULL: no ReLU.
 * Also dgamma[c] (+)= sum_g s2, dbeta[c] (+)= sum_g s1 (accumulate flag).
 * dz_out (may be NULL, may alias g): the masked gradient dz is written out, so that mvg_bn_bwd_apply can
 * run on it without reading the mask again and the residual branch gets its gradient from the same buffer.
 * workspace: mvg_bn_bwd_workspace_floats() floats. */
int mvg_bn_bwd_reduce(const float *g, const float *act, const float *y, const float *mean,
                      const float *invstd, const float *relu_scale, const float *relu_shift,
                      int groups, int64_t rows_per_group, int c, float *s1, float *s2, float *dgamma,
                      float *dbeta, int accumulate, float *workspace, float *dz_out, void *stream);
size_t mvg_bn_bwd_workspace_floats(int groups, int64_t rows_per_group, int c);
/* Backward, step 2: dy = gamma*invstd*(dz - s1/n - xhat*s2/n); dz_out (optional, may alias g)
 * receives the masked gradient for the residual branch. */
int mvg_bn_bwd_apply(const float *g, const float *act, const float *y, const float *mean,
                     const float *invstd, const float *gamma, const float *s1, const float *s2,
                     const float *relu_scale, const float *relu_shift, int groups,
                     int64_t rows_per_group, int c, float *dy, float *dz_out, void *stream);

/* ---------------------------------------------------------------- pooling / layout
 * nn.MaxPool2d(3,2,1) resnet.py:189; nn.AdaptiveAvgPool2d((1,1)) resnet.py:200 + rot_mv.py:126;
 * NCHW float input of FeatRotationSymm.forward (rot_mv.py:188-189) -> NHWC4. */
int mvg_maxpool3x3s2_fwd(const float *x, float *y, uint8_t *argmax, int n, int h, int w, int c,
                         int ho, int wo, void *stream);
int mvg_maxpool3x3s2_bwd(const float *dy, const uint8_t *argmax, float *dx, int n, int h, int w,
                         int c, int ho, int wo, void *stream);
/* Stem tail fused: pooled = MaxPool2d(3,2,1)(relu(y*scale + shift)) without materialising the
 * normalised activation (torchvision resnet.py:187-189 conv1 -> bn1 -> relu -> maxpool, called from
 * /root/reference/models/rot_mv.py:204-205).  y [groups][n_per_group][h][w][c]; scale/shift
 * [groups][c] from mvg_bn_finalize / mvg_bn_eval_affine.  The backward pair rebuilds the gradient
 * of the BN output from (g_pooled, argmax) and the ReLU mask from y: same results as
 * mvg_maxpool3x3s2_bwd -> mvg_bn_bwd_reduce -> mvg_bn_bwd_apply, 2.3 GB less traffic at C2. */
int mvg_bn_relu_maxpool_fwd(const float *y, const float *scale, const float *shift, float *pooled,
                            uint8_t *argmax, int groups, int n_per_group, int h, int w, int c,
                            int ho, int wo, void *stream);
int mvg_bn_relu_maxpool_bwd_reduce(const float *g_pooled, const uint8_t *argmax, const float *y,
                                   const float *mean, const float *invstd, const float *scale,
                                   const float *shift, int groups, int n_per_group, int h, int w,
                                   int c, int ho, int wo, float *s1, float *s2, float *dgamma,
                                   float *dbeta, int accumulate, float *workspace, void *stream);
int mvg_bn_relu_maxpool_bwd_apply(const float *g_pooled, const uint8_t *argmax, const float *y,
                                  const float *mean, const float *invstd, const float *gamma,
                                  const float *scale, const float *shift, const float *s1,
                                  const float *s2, int groups, int n_per_group, int h, int w, int c,
                                  int ho, int wo, float *dy, void *stream);
int mvg_avgpool_fwd(const float *x, float *y, int n, int hw, int c, void *stream);
int mvg_avgpool_bwd(const float *dy, float *dx, int n, int hw, int c, void *stream);
int mvg_nchw_to_nhwc4(const float *src, float *dst, int n, int c, int h, int w, void *stream);
int mvg_nhwc4_to_nchw(const float *src, float *dst, int n, int c, int h, int w, void *stream);
/* GPU input pipeline: raw uint8 [n][h][w][3] face patches (the HDF5
 * `face_patch` layout, dataset/gaze.py:122) -> optional BGR->RGB (:108-109) -> /255 (ToTensor) ->
 * (x - mean)/std (Normalize, main.py:38-39,54) -> NHWC4 fp32, the backbone's input layout. */
int mvg_preprocess_u8hwc(const uint8_t *src, float *dst, int n, int h, int w, float mean0, float mean1,
                         float mean2, float std0, float std1, float std2, int swap_rb, void *stream);
/* The same with the Resize((oh, ow), antialias=True) of main.py:46,53 between ToTensor and Normalize, for
 * patches that are not already oh x ow (h == oh && w == ow is mvg_preprocess_u8hwc: torchvision returns
 * the input unchanged).  Resize on a float tensor is torch.nn.functional.interpolate(mode="bilinear",
 * antialias=True, align_corners=False) (ATen _upsample_bilinear2d_aa): width pass, then height.
 * dst [n][oh][ow][4]. */
int mvg_preprocess_u8hwc_resize(const uint8_t *src, float *dst, int n, int h, int w, int oh, int ow, float mean0,
                                float mean1, float mean2, float std0, float std1, float std2, int swap_rb,
                                void *stream);
/* RandomMultiErasing.__call__ utils/augment.py:38-47 on a device batch: img [n][c][h][w] *= the
 * nearest-neighbour upsampling (F.interpolate default, augment.py:21) of masks[n] (grid[n] x grid[n]
 * floats in a gmax*gmax slot); grid[n] == 0 leaves image n untouched.  The random draws stay on the
 * host (rot_mvgaze_amd/augment.py replays the reference's RNG calls). */
int mvg_multi_erase_nchw(float *img, const float *masks, const int32_t *grid, int gmax, int n, int c,
                         int h, int w, void *stream);

/* ---------------------------------------------------------------- geometry
 * rotation_matrix_2d utils/math.py:188-219; relative rotations rot_mv.py:193-194. */
int mvg_rotation_matrix_2d(const float *pitch_yaw /*[n][2]*/, float *rot /*[n][3][3]*/, int n,
                           int inverse, void *stream);
/* rel[d][b] = rot[b][vi[d]] @ rot[b][vj[d]]^T for d < dirs; rot [b][views][3][3]. */
int mvg_relative_rotation(const float *rot, const int32_t *vi, const int32_t *vj, float *rel,
                          int batch, int views, int dirs, void *stream);

/* ---------------------------------------------------------------- cross-view fusion operands
 * ImageFeatFuser.forward's cat([img_feat, (R @ F).flatten]) rot_mv.py:44-50,234-239 and the head's
 * cat([img_feat, F.flatten]) :249-254.  Row (d, b) of x = [ img_feat[view_of[d]][b] (cf floats),
 * rel[d][b] @ feat[src_of[d]][b] (3*nvec floats, axis-major) ]; rel == NULL means identity. */
int mvg_rotcat_fwd(const float *img_feat /*[views][batch][cf]*/, const float *feat /*[dirs][batch][3][nvec]*/,
                   const float *rel /*[dirs][batch][3][3] or NULL*/, const int32_t *view_of,
                   const int32_t *src_of, float *x /*[dirs][batch][cf+3*nvec]*/, int batch, int dirs,
                   int cf, int nvec, void *stream);
/* dfeat[src_of[d]][b] = rel[d][b]^T @ dx_rot  (src_of must be injective: every slot written once). */
int mvg_rotcat_bwd(const float *dx, const float *rel, const int32_t *src_of, float *dfeat, int batch,
                   int dirs, int cf, int nvec, void *stream);
/* ---------------------------------------------------------------- ablation variants
 * encode_rotmat, ImageRotmatFeatFuser.forward rot_mv.py:62-69: row (d, b) of x (row length ld,
 * zero-padded) = [ img_feat[view_of[d]][b] | (rel_apply[d][b] @) feat[src_of[d]][b] | rel_append[d][b]
 * flattened (9) ]; either rel may be NULL (no rotation applied / nothing appended). */
int mvg_rotcat_ext_fwd(const float *img_feat, const float *feat, const float *rel_apply,
                       const float *rel_append, const int32_t *view_of, const int32_t *src_of,
                       float *x, int ld, int batch, int dirs, int cf, int nvec, void *stream);
int mvg_rotcat_ext_bwd(const float *dx, int ld, const float *rel, const int32_t *src_of, float *dfeat,
                       int batch, int dirs, int cf, int nvec, void *stream);
/* share_feature: IntensityBatchNorm rot_mv.py:13-32 as used by RotFeatFuser.forward :82-86.  The
 * 2*dirs calls of one iteration (direction d: a[view_of[d]] then feat[src_of[d]]) run in order on
 * the fuser's running_mean [nvec] (updated in place when training); scales [2*dirs][nvec] =
 * 1 / (running_mean + eps) as each call saw it. */
int mvg_ibn_scales(const float *a /*[views][batch][3][nvec]*/, const float *feat /*[srcs][batch][3][nvec]*/,
                   const int32_t *view_of, const int32_t *src_of, float *running_mean, int training,
                   float momentum, float eps, float *scales, int batch, int dirs, int nvec,
                   void *stream);
/* x[(d,b)][axis][0:nvec] = scales[2d] * a[view_of[d]][b][axis], [nvec:2nvec] = scales[2d+1] * (rel[d][b] @
 * feat[src_of[d]][b])[axis]: cat([.,.], dim=-1).flatten(-2,-1) of rot_mv.py:85 (scales) and :241-247
 * (gaze head input, scales == NULL, rel == NULL).  Backward: da_dir [(d,b)][3][nvec] (sum it over
 * directions with mvg_segment_sum) and dfeat[src_of[d]][b] = rel^T @ (...). */
int mvg_paircat_fwd(const float *a, const float *feat, const float *rel, const float *scales,
                    const int32_t *view_of, const int32_t *src_of, float *x, int batch, int dirs,
                    int nvec, void *stream);
int mvg_paircat_bwd(const float *dx, const float *rel, const float *scales, const int32_t *src_of,
                    float *da_dir, float *dfeat, int batch, int dirs, int nvec, void *stream);
/* out[v][b][0:width] (+)= sum over d with seg_of[d] == v, ascending d (reproducible), of
 * x[(d*batch + b)*row_stride + 0:width].  Gradient fan-in of the per-view features that several
 * directed pairs read (img_feat in every fuser/head input, the lifted feature at iteration 0). */
int mvg_segment_sum(const float *x, int64_t row_stride, int width, const int32_t *seg_of, float *out,
                    int batch, int dirs, int segments, int accumulate, void *stream);

/* y = a*x + b*y (n floats) - gradient fan-in accumulation. */
int mvg_axpby(const float *x, float *y, float a, float b, int64_t n, void *stream);
/* out = x * scale[0] with the scale read on the device (upstream gradient of a scalar loss). */
int mvg_scale_by(const float *x, const float *scale, float *out, int64_t n, void *stream);

/* Linear with out_features <= 4 (the gaze head's Linear(512 -> 2), rot_mv.py:179-184), for which a
 * 32-wide MFMA tile would be >90% padding.  bwd: dx = (dy @ w) * (mask > 0) [mask optional],
 * dw (+)= dy^T @ x, db (+)= colsum(dy); dx / dw may be NULL to skip. */
int mvg_linear_skinny_fwd(const float *x, const float *w, const float *bias, float *y, int rows, int k,
                          int nout, void *stream);
/* dx_absmax (may be NULL): a device float slot that receives max |dx| (its bits, by atomicMax: the caller clears it) */
int mvg_linear_skinny_bwd(const float *dy, const float *x, const float *w, const float *mask, float *dx,
                          float *dw, float *db, int rows, int k, int nout, int accumulate, float *dx_absmax, void *stream);

/* ---------------------------------------------------------------- optimizer
 * torch.optim.Adam step (trainer.py:54,141-143: betas (0.9, 0.999), eps 1e-8, coupled L2
 * weight_decay 1e-6, bias correction) over flat fp32 arenas - the model keeps parameters and
 * gradients in two arenas with identical offsets, so one launch updates every parameter. */
int mvg_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, void *stream);
/* The same step with nothing from the host but the launch itself (a step captured in a hipGraph replays it): lr_dev = one
 * device float (CyclicLR writes it when it steps, trainer.py:58-62,147), state3 = three device floats {step, 1 - beta1^step,
 * sqrt(1 - beta2^step)} - zero-initialised, advanced by a one-thread launch in front of the update. */
int mvg_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *lr_dev,
                      float *state3, float beta1, float beta2, float eps, float weight_decay, void *stream);

/* ---------------------------------------------------------------- loss
 * gaze_angular_loss losses/gaze_loss.py:42-52 over pitchyaw_to_vector utils/math.py:52-60:
 * theta_i = acos(clamp(cos_sim(v(gt_i), v(pred_i), eps 1e-6), -1, 1)) * 180/pi;
 * loss[0] (+)= sum_i row_weight * theta_i ; dpred_i = row_weight * dtheta_i/dpred_i (optional).
 * (row_weight = 1/batch gives the reference's batch mean.)  theta_out optional [n]. */
int mvg_gaze_angular_loss(const float *pred, const float *gt, int n, float row_weight,
                          float *loss, int accumulate, float *dpred, float *theta_out, void *stream);

/* gaze_l1_loss / gaze_l2_loss losses/gaze_loss.py:56-64 (GazeLoss loss_type 'l1' / 'l2', :21-29; not used by
 * StereoL1Loss, which fixes 'angular'): loss[0] = mean over the n elements of |pred - label|^p, p in {1, 2};
 * dpred (optional, n floats) = d loss / d pred (zero where pred == label, like torch.abs). */
int mvg_gaze_lp_loss(const float *pred, const float *label, int n, int p, float *loss, float *dpred, void *stream);
/* IterationLoss(StereoL1Loss) (stereo_loss.py:46-54,65-84) over the head's stacked predictions in ONE launch: pred
 * [iters][dirs * batch][2], gt [dirs * batch][2]; host_weights [iters * dirs] (HOST array, copied into the launch): row r of
 * iteration i weighs host_weights[i * dirs + r / batch] / batch.  loss[0] = the weighted sum of the angular errors (degrees),
 * dpred (may be NULL) = its gradient. */
int mvg_gaze_angular_loss_multi(const float *pred, const float *gt, int iters, int dirs, int batch, const float *host_weights,
                                float *loss, float *dpred, void *stream);

/* ---------------------------------------------------------------- stereo pair index (HOST)
 * GazeDataset.__init__'s idx_to_kv build, dataset/gaze.py:39-73, driven by CPython's
 * random.seed(int)/random.choice stream (MT19937 + getrandbits rejection).  Pure host integer
 * code; state (625 words: mt[624] + index) persists across calls like Python's global RNG.
 *   camera_tag: 0 = all, 1 = novel_train, 2 = novel_test.
 * Returns the number of tuples written (<= capacity) or -1 on error. */
int mvg_mt19937_seed(uint32_t *host_state /*[625]*/, uint64_t seed);
int64_t mvg_pair_index_build(uint32_t *host_state, const int64_t *host_file_rows, int n_files,
                             int camera_tag, int64_t *host_out /*[capacity][3]*/, int64_t capacity);


/* ---------------------------------------------------------------------------------------------
 * bf16 storage path (BASELINE.json configs[4]: "bf16 MFMA path").  Nothing in the reference
 * corresponds to it (the reference is fp32 everywhere: models/resnet.py:31-47, models/backbones/
 * blocks.py:41-47); these entry points are the same operators as above with activations, their
 * gradients and the conv weights held as bf16 (uint16_t storage) in HBM.  Matrix products run on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; BatchNorm statistics, scale/shift, biases, weight
 * gradients and the loss stay fp32.  Channels (cin, cout) must be multiples of 8 (16-byte vectors);
 * the 3-channel stem input is padded to 8 by mvg_nchw_to_nhwc8_bf16.  The parity bar of this path is
 * declared in tests/test_bf16_gpu.py (it cannot meet the fp32 path's 1e-4).
 * --------------------------------------------------------------------------------------------- */
/* fp32 KRSC weights (cin_src <= d->cin channels per tap, the rest zero) -> bf16 KRSC `w_krsc`
 * [cout][r][s][d->cin] and, when w_crsk != NULL, the transposed copy [d->cin][r][s][cout] that
 * mvg_conv_dgrad_bf16 reads (one cast per weight per step: the master weights stay fp32). */
int mvg_cast_weights_bf16(const mvg_conv_desc *d, const float *w, int cin_src, void *w_krsc, void *w_crsk,
                          void *stream);
/* like mvg_conv_fprop / mvg_conv_stats_partials; x, wgt (KRSC), y are bf16, bias/stats fp32 */
int mvg_conv_fprop_bf16(const mvg_conv_desc *d, const void *x, const void *wgt, void *y, const float *bias,
                        int relu, float *stats, void *stream);
int mvg_conv_stats_partials_bf16(const mvg_conv_desc *d, int32_t *out_rows_per_partial);
/* like mvg_conv_dgrad; wgt_crsk = the transposed bf16 weights; mask/addend (bf16) like dx */
int mvg_conv_dgrad_bf16(const mvg_conv_desc *d, const void *dy, const void *wgt_crsk, void *dx,
                        const void *mask, const void *addend, void *stream);
/* mvg_conv_dgrad_bf16 fused with the BatchNorm-backward reduce pass of the unit whose output gradient dx is (the bf16
 * form of mvg_conv_dgrad_split_bnreduce below; resnet.py:113-133 backward): dx is stored masked by that unit's ReLU
 * (bn_bits from mvg_bn_apply_bits_bf16 - one byte per 8 channels -, or fma(bn_y, relu_scale, relu_shift) > 0, or no mask)
 * and s1 / s2 / dgamma / dbeta - sums of the ROUNDED gradient, what mvg_bn_bwd_reduce_bf16 would read back - come
 * out of the same launch + a finalize.  Stride 1 or 2; cin, cout multiples of 64; bn_y bf16 shaped like dx.
 * partials: groups * mvg_conv_dgrad_bn_partials_bf16(d) * 2 * cin floats. */
int mvg_conv_dgrad_bn_partials_bf16(const mvg_conv_desc *d);
int mvg_conv_dgrad_bf16_bnreduce(const mvg_conv_desc *d, const void *dy, const void *wgt_crsk, void *dx, const void *addend,
                                 const void *bn_y, const uint8_t *bn_bits, const float *bn_mean, const float *bn_invstd,
                                 const float *relu_scale, const float *relu_shift, float *partials, float *s1, float *s2,
                                 float *dgamma, float *dbeta, int accumulate, void *stream);
/* The 7x7 stride-2 stem (resnet.py:184) on the LDS-DMA bf16 kernels as FOLDED row windows: xw [images][h][w/4][16][4]
 * bf16 - window m holds image columns 4 m - 4 .. 4 m + 11 x (R, G, B, 0), zero outside the image, made straight from
 * the NCHW fp32 input - serves output columns 2 m and 2 m + 1 as 2 * cout GEMM columns, so the kernel's output
 * [.., wo/2, 2 cout] is y [.., wo, cout] in memory.  d = the stem's descriptor (r = s = 7, stride 2, pad 3; w % 4 == 0;
 * n * ho * wo / 2 a multiple of 64).  w_fold: bf16 [2 cout][7][16][4] with w_fold[par * cout + o][r][j][c] =
 * w[o][r][j - 1 - 2 par][c] (zero elsewhere).  stats: BatchNorm partials of y, groups * 2 P * 2 * cout floats with
 * P = mvg_conv_stats_partials_bf16 of a descriptor with n, ho, wo / 2 (64 outputs per partial, like every bf16 forward).
 * mvg_stem_wgrad_bf16: dw_fold fp32 in the same layout (the caller adds the two parities' taps);
 * workspace = splits * 2 cout * 448 floats, splits = mvg_stem_wgrad_splits_bf16(d). */
int mvg_stem_rowwindow_bf16(const float *x_nchw, void *xw, int64_t images, int h, int w, void *stream);
int mvg_stem_fprop_bf16(const mvg_conv_desc *d, const void *xw, const void *w_fold, void *y, float *stats, void *stream);
int mvg_stem_wgrad_splits_bf16(const mvg_conv_desc *d);
int mvg_stem_wgrad_bf16(const mvg_conv_desc *d, const void *xw, const void *dy, float *dw_fold, float *workspace, int splits,
                        int accumulate, void *stream);
/* like mvg_conv_wgrad / mvg_conv_wgrad_splits; x, dy bf16, dw (and the slabs in workspace) fp32 */
int mvg_conv_wgrad_bf16(const mvg_conv_desc *d, const void *x, const void *dy, float *dw, float *workspace,
                        int splits, int accumulate, void *stream);
int mvg_conv_wgrad_splits_bf16(const mvg_conv_desc *d);
/* the launch without its slab reduce (splits > 1), for mvg_wgrad_reduce_batch - see mvg_conv_wgrad_split_slabs */
int mvg_conv_wgrad_bf16_slabs(const mvg_conv_desc *d, const void *x, const void *dy, float *workspace, int splits, void *stream);
/* nn.Linear of the fusion block in the bf16 path ("mixed"): activations, gradients, biases and outputs stay
 * fp32 in memory, the operand loaders round to bf16 on the way into LDS and the product runs on the bf16
 * matrix cores against the bf16 weight copies (w_bf16 [fout][fin]; wt_bf16 [fin][fout] for backward-data).
 * No split-K: meant for the large row counts of the bf16 configuration (C5: 3584 rows per GPU).
 *   fprop: y = [relu](x @ w^T + bias);  dgrad: dx = (dy @ w) * (mask > 0) + addend;
 *   wgrad: dw (+)= dy^T x, db (+)= column sums of the fp32 dy; splits = mvg_conv_wgrad_splits_bf16(linear
 *   descriptor), workspace splits * (fout*fin + fout) floats when splits > 1. */
int mvg_linear_fprop_mixed(const float *x, const void *w_bf16, const float *bias, int relu, float *y, int rows, int fin,
                           int fout, void *stream);
int mvg_linear_dgrad_mixed(const float *dy, const void *wt_bf16, const float *mask, const float *addend, float *dx,
                           int rows, int fin, int fout, void *stream);
int mvg_linear_wgrad_mixed(const float *x, const float *dy, float *dw, float *db, int rows, int fin, int fout,
                           float *workspace, int splits, int accumulate, void *stream);
/* the BatchNorm / pooling passes with bf16 activations (same arguments as the fp32 entry points) */
int mvg_bn_apply_bf16(const uint16_t *y, const float *scale, const float *shift, const uint16_t *residual,
                      const float *res_scale, const float *res_shift, int relu, uint16_t *out, int groups,
                      int64_t rows_per_group, int c, void *stream);
int mvg_bn_bwd_reduce_bf16(const uint16_t *g, const uint16_t *act, const uint16_t *y, const float *mean,
                           const float *invstd, const float *relu_scale, const float *relu_shift, int groups,
                           int64_t rows_per_group, int c, float *s1, float *s2, float *dgamma, float *dbeta,
                           int accumulate, float *workspace, uint16_t *dz_out, void *stream);
int mvg_bn_bwd_apply_bf16(const uint16_t *g, const uint16_t *act, const uint16_t *y, const float *mean,
                          const float *invstd, const float *gamma, const float *s1, const float *s2,
                          const float *relu_scale, const float *relu_shift, int groups, int64_t rows_per_group,
                          int c, uint16_t *dy, uint16_t *dz_out, void *stream);
int mvg_bn_relu_maxpool_fwd_bf16(const uint16_t *y, const float *scale, const float *shift, uint16_t *pooled,
                                 uint8_t *argmax, int groups, int n_per_group, int h, int w, int c, int ho,
                                 int wo, void *stream);
int mvg_bn_relu_maxpool_bwd_reduce_bf16(const uint16_t *g_pooled, const uint8_t *argmax, const uint16_t *y,
                                        const float *mean, const float *invstd, const float *scale,
                                        const float *shift, int groups, int n_per_group, int h, int w, int c,
                                        int ho, int wo, float *s1, float *s2, float *dgamma, float *dbeta,
                                        int accumulate, float *workspace, void *stream);
int mvg_bn_relu_maxpool_bwd_apply_bf16(const uint16_t *g_pooled, const uint8_t *argmax, const uint16_t *y,
                                       const float *mean, const float *invstd, const float *gamma,
                                       const float *scale, const float *shift, const float *s1,
                                       const float *s2, int groups, int n_per_group, int h, int w, int c,
                                       int ho, int wo, uint16_t *dy, void *stream);
/* bf16 feature map -> fp32 pooled features and back (the fusion block stays fp32) */
int mvg_avgpool_fwd_bf16(const uint16_t *x, float *y, int n, int hw, int c, void *stream);
int mvg_avgpool_bwd_bf16(const float *dy, uint16_t *dx, int n, int hw, int c, void *stream);
/* fp32 NCHW images (rot_mv.py:188-189) -> bf16 NHWC with the channels zero-padded to 8 */
int mvg_nchw_to_nhwc8_bf16(const float *src, uint16_t *dst, int n, int c, int h, int w, void *stream);

/* ---- "split" operands: fp32-accurate convolutions on the fp16 matrix cores (csrc/conv_split.hip) ----------------
 * An fp32 tensor in "sp" format holds every value v - times a per-tensor power-of-two scale 2^k chosen by its producer -
 * as TWO fp16 pieces, h1 = fp16(v 2^k), h2 = fp16(v 2^k - h1): |v - (h1 + h2) 2^-k| <= 2^-23 |v| (at most the last of the
 * 24 significand bits is lost; three values in four are exact).  Channels go in chunks of 8 with the two pieces of a chunk adjacent (4 bytes per element).
 * Three fp16 MFMAs per 32 k (h1 g1 + h1 g2 + h2 g1, fp32 accumulation) reproduce the fp32 product to fp32-accumulation
 * accuracy (conv_split.hip header; tests/test_split_gpu.py holds every kernel to 2e-6 against fp64 and to the
 * fp32-MFMA kernel's own error), so these entries serve the same 1e-4 parity path as mvg_conv_fprop / _dgrad / _wgrad
 * (resnet.py:31-47: F.conv2d forward and its autograd backward) - outputs (y, dx, dw) are plain fp32.
 * Scales travel as DEVICE scalars "sinv" = 2^-k next to their tensor (NULL = 1): activations are stored unscaled,
 * gradients dy get theirs from mvg_bn_bwd_apply_split, weight copies from the weight prep; a consumer multiplies its
 * accumulators by the sinv of both operands.  Shapes: cin, cout multiples of 32, r*s <= 32 (everything in the
 * backbone but the 3-channel stem). */
int mvg_split_f32(const float *x, void *out_sp, int64_t n, float scale, void *stream);        /* n % 8 == 0; out = sp(x * scale) */
int mvg_merge_sp(const void *x_sp, float *out, int64_t n, float inv_scale, void *stream);     /* out = (h1 + h2) * inv_scale */
/* fp32 KRSC weights -> sp KRSC (fprop) and, when w_crsk_sp != NULL, the sp transposed copy CRSK (dgrad), both scaled by the
 * 2^k that puts max |w| just below 2^15.  stat2: two device floats, receive {max |w| (bits), 2^-k}: pass stat2 + 1 as the
 * w_sinv of the consumers.  items_dev: 64 bytes of device memory (staging of the one-record table). */
int mvg_split_weights(const mvg_conv_desc *d, const float *w, void *w_krsc_sp, void *w_crsk_sp, float *stat2, void *items_dev,
                      void *stream);
/* All weight copies of a training step in two launches (max |w| per conv, then the copies).  items_dev: n records in
 * DEVICE memory of
 *   { const float *w (fp32 KRSC); void *wk; void *wt (NULL: no transposed copy); int32 cout, rs, cin, cin_pad;
 *     float *stat (mode 1: two device floats per conv, stat[0] CLEARED by the caller, receive {max |w| bits, 2^-k}); }  (48 bytes)
 * mode 1: wk / wt = the sp KRSC / CRSK copies of mvg_split_weights (cin_pad unused); mode 0: the bf16 KRSC (cin
 * zero-padded to cin_pad) / CRSK copies of mvg_cast_weights_bf16 (stat unused). */
int mvg_weights_prep_batch(const void *items_dev, int n, int mode, int blocks_per_item /* 0: 64 */, void *stream);
/* partial-statistics geometry of mvg_conv_fprop_split (like mvg_conv_stats_partials) */
int mvg_conv_stats_partials_split(const mvg_conv_desc *d, int32_t *rows_per_partial);
int mvg_conv_fprop_split(const mvg_conv_desc *d, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv,
                         float *y, float *stats, void *stream);
/* inference forward with BatchNorm folded into the epilogue (like mvg_conv_fprop_affine): out = relu?(conv * scale +
 * shift (+ residual)); residual fp32 or sp (residual_sp; unscaled), the result fp32 or sp (out_sp; unscaled) - in sp the
 * next conv reads it directly (resnet.py:60-75,113-133 in eval mode) */
int mvg_conv_fprop_split_affine(const mvg_conv_desc *d, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv,
                                void *out, int out_sp, const float *scale, const float *shift, const void *residual,
                                int residual_sp, int relu, void *stream);
/* relu_mask_sp (may be NULL; stride-1 launches): an sp tensor shaped like dx - dx is zeroed where it is <= 0 (the ReLU of a
 * Linear's hidden layer, whose activation the forward wrote in sp) */
int mvg_conv_dgrad_split(const mvg_conv_desc *d, const void *dy_sp, const float *dy_sinv, const void *w_crsk_sp,
                         const float *w_sinv, float *dx, const float *addend, const void *relu_mask_sp, void *stream);
/* The BatchNorm passes on the split path: the conv output y and every gradient g stay fp32; what the next conv
 * reads is written in sp - the normalised activation (mvg_bn_apply_split, residual = the previous block's sp output
 * when residual_sp != 0, else the raw fp32 downsample output with its res_scale / res_shift), the stem's pooled map
 * (mvg_bn_relu_maxpool_fwd_split) and dy (mvg_bn_bwd_apply_split: g already masked, or the mask from relu_scale /
 * relu_shift).  relu_bits as in mvg_bn_apply_bits (1 byte per 4 channels).  mvg_avgpool_fwd_split pools an sp map.
 * dy is stored times 2^k, k from the bound  max over (group, channel) of |gamma invstd| (mx + |s1|/n + sqrt(n) |s2|/n) >= |dy|,
 * mx [groups][c] = max |masked gradient| per (group, channel), left by mvg_bn_bwd_reduce_split or
 * mvg_conv_dgrad_split_bnreduce; *dy_sinv receives 2^-k for mvg_conv_dgrad_split / _wgrad_split.  The reduce entries
 * (mvg_bn_bwd_reduce_split, mvg_conv_dgrad_split_bnreduce, mvg_bn_relu_maxpool_bwd_reduce_split) take the unit's gamma and
 * dy_sinv (both NULL, or both given): their finalize launch then leaves *dy_sinv itself - partials, dgamma / dbeta and the
 * bound in ONE launch, folded into the last-arriving workgroups - and the apply entry is told so (dy_sinv_ready != 0). */
int mvg_bn_apply_split(const float *y, const float *scale, const float *shift, const void *residual, int residual_sp,
                       const float *res_scale, const float *res_shift, int relu, void *out_sp, uint8_t *relu_bits,
                       int groups, int64_t rows_per_group, int c, void *stream);
int mvg_bn_bwd_reduce_split(const float *g, const uint8_t *relu_bits, const float *y, const float *mean, const float *invstd,
                            const float *relu_scale, const float *relu_shift, int groups, int64_t rows_per_group, int c,
                            float *s1, float *s2, float *dgamma, float *dbeta, int accumulate, float *workspace,
                            float *dz_out, float *mx, const float *gamma, float *dy_sinv, void *stream);
int mvg_bn_bwd_apply_split(const float *g, const float *y, const float *mean, const float *invstd, const float *gamma,
                           const float *s1, const float *s2, const float *relu_scale, const float *relu_shift,
                           int groups, int64_t rows_per_group, int c, void *dy_sp, const float *mx, float *dy_sinv,
                           int dy_sinv_ready, void *stream);
int mvg_bn_relu_maxpool_fwd_split(const float *y, const float *scale, const float *shift, void *pooled_sp,
                                  uint8_t *argmax, int groups, int n_per_group, int h, int w, int c, int ho, int wo,
                                  void *stream);
int mvg_avgpool_fwd_split(const void *x_sp, float *y, int n, int hw, int c, void *stream);
/* mvg_conv_dgrad_split fused with the BatchNorm-backward reduce pass of the unit whose output gradient dx is (stride 1
 * or 2: a stride-2 launch's parity classes - those a 1x1 filter never touches included - each bring their partials): dx is stored masked by that unit's ReLU (bn_bits from mvg_bn_apply_split, or fma(bn_y, relu_scale,
 * relu_shift) > 0, or no mask) and s1 / s2 / dgamma / dbeta come out of the same launch + a finalize; mx [groups][cin] (may be
 * NULL) receives max |dx| per (group, channel).  partials: groups * mvg_conv_dgrad_bn_partials_split(d) * 3 * cin floats. */
int mvg_conv_dgrad_bn_partials_split(const mvg_conv_desc *d);
int mvg_conv_dgrad_split_bnreduce(const mvg_conv_desc *d, const void *dy_sp, const float *dy_sinv, const void *w_crsk_sp,
                                  const float *w_sinv, float *dx, const float *addend, const float *bn_y, const uint8_t *bn_bits,
                                  const float *bn_mean, const float *bn_invstd, const float *relu_scale, const float *relu_shift,
                                  float *partials, float *s1, float *s2, float *dgamma, float *dbeta, int accumulate,
                                  float *mx, const float *bn_gamma, float *dx_dy_sinv, void *stream);
/* The 7x7 stride-2 stem (resnet.py:184, ResNet.forward :262) on the split kernels, "row-window" form: the 3-channel image
 * (stored NHWC with 4 channels) is rewritten as xw [images][h][w/2][8][4] in sp - window ox holds image columns 2 ox - 4 ..
 * 2 ox + 3, zero outside the image - and the stem becomes a 7 x 1 filter over 32 "channels" (vertical stride 2 / pad 3,
 * horizontal stride 1 / pad 0): K = 224.  d = the stem's own descriptor (r = s = 7, stride 2, pad 3, cin = 4, even w).
 *   mvg_stem_rowwindow_split: x_nhwc4 fp32 -> xw_sp (32 * images * h * w/2 sp elements);
 *   w_sp: the sp copy (mvg_split_weights on a [cout][7][1][32] descriptor) of w'[o][r][j][c] = w[o][r][j - 1][c], zero for
 *         j = 0 and c = 3;   mvg_stem_fprop_split: y fp32 [groups][n][ho][wo][cout] + BatchNorm partials like
 *         mvg_conv_fprop_split (mvg_conv_stats_partials_split of a descriptor with the same n, ho, wo);
 *   mvg_stem_wgrad_split: dw_rw [cout][7][8][4] fp32 in the same tap layout (the caller drops j = 0 and c = 3);
 *         workspace = splits * cout * 224 floats, splits = mvg_stem_wgrad_splits_split(d).
 * The stem tail's backward with dy in sp: mvg_bn_relu_maxpool_bwd_reduce_split also leaves mx [groups][c], a bound on a
 * pixel's gradient (4 x the largest masked window gradient); mvg_bn_relu_maxpool_bwd_apply_split scales dy by the 2^k that
 * bound allows and writes 2^-k to *dy_sinv. */
int mvg_stem_rowwindow_split(const float *x_nhwc4, void *xw_sp, int64_t images, int h, int w, void *stream);
int mvg_stem_rowwindow_split_nchw(const float *x_nchw, void *xw_sp, int64_t images, int h, int w, void *stream);   /* from [images][3][h][w] */
int mvg_stem_fprop_split(const mvg_conv_desc *d, const void *xw_sp, const void *w_sp, const float *w_sinv, float *y, float *stats,
                         void *stream);
int mvg_stem_wgrad_splits_split(const mvg_conv_desc *d);
int mvg_stem_wgrad_split(const mvg_conv_desc *d, const void *xw_sp, const void *dy_sp, const float *dy_sinv, float *dw_rw,
                         float *workspace, int splits, int accumulate, void *stream);
int mvg_bn_relu_maxpool_bwd_reduce_split(const float *g_pooled, const uint8_t *argmax, const float *y, const float *mean,
                                         const float *invstd, const float *scale, const float *shift, int groups,
                                         int n_per_group, int h, int w, int c, int ho, int wo, float *s1, float *s2,
                                         float *dgamma, float *dbeta, int accumulate, float *workspace, float *mx, const float *gamma,
                                         float *dy_sinv, void *stream);
int mvg_bn_relu_maxpool_bwd_apply_split(const float *g_pooled, const uint8_t *argmax, const float *y, const float *mean,
                                        const float *invstd, const float *gamma, const float *scale, const float *shift,
                                        const float *s1, const float *s2, int groups, int n_per_group, int h, int w, int c,
                                        int ho, int wo, void *dy_sp, const float *mx, float *dy_sinv, int dy_sinv_ready, void *stream);
int mvg_conv_wgrad_splits_split(const mvg_conv_desc *d);   /* pixel-split count; workspace = splits * cout*r*s*cin floats */
int mvg_conv_wgrad_split(const mvg_conv_desc *d, const void *x_sp, const void *dy_sp, const float *dy_sinv, float *dw,
                         float *workspace, int splits, int accumulate, void *stream);
/* The same launch WITHOUT its slab reduce (splits > 1: workspace receives the per-split partial gradients), and the reduces
 * of up to 8 such gradients in ONE launch (a residual block's convs): dw[i] (+)= the sum over host_splits[i] slabs of
 * host_n[i] floats, fixed order.  Host arrays; the launch copies them. */
int mvg_conv_wgrad_split_slabs(const mvg_conv_desc *d, const void *x_sp, const void *dy_sp, const float *dy_sinv, float *workspace,
                               int splits, void *stream);
int mvg_wgrad_reduce_batch(const float *const *host_slabs, float *const *host_dw, const int64_t *host_n, const int32_t *host_splits,
                           const int32_t *host_accumulate, int n, void *stream);

/* ---------------------------------------------------------------- the fusion block on the split kernels
 * ImageFeatFuser / gaze head Linears (rot_mv.py:35-50,179-184,234-254; blocks.py:41-47) with >= 1024 rows: every tensor a
 * split kernel reads carries a per-tensor power-of-two scale (the reference computes these in plain fp32 and has no range
 * limit; fp16 pieces do), found WITHOUT extra passes: fp32 results leave max |.| in a device slot from the producing
 * launch's epilogue (float bits, atomicMax: order-independent; the caller clears the slots once per step), sp results are
 * stored times the 2^k a bound allows.
 *   mvg_absmax_multi: max |x| of up to 8 small tensors (HOST arrays of device pointers / counts / slots) in one launch.
 *   mvg_fuse_build_split: the operands built from a feature tensor F [.][3][nvec], straight into sp:
 *       xf[m] = [ img_feat[row_img[m]] | rel[m] @ F[row_src_f[m]] ]  (next iteration's fuser input; NULL = not wanted)
 *       xh[m] = [ img_feat[row_img[m]] |          F[row_src_h[m]] ]  (this iteration's head input;  NULL = not wanted)
 *     scaled from am_img / am_feat (max |img_feat|, max |F|: |rel @ f| <= sqrt(3) max |f|); *xf_sinv / *xh_sinv = 2^-k.
 *   mvg_linear_fprop_split: out = relu?(x W^T + bias); out fp32 (out_absmax, may be NULL, receives max |out|) or sp
 *     (out_sp: stored times 2^k from the bound fin * 2^30 * x_sinv * w_sinv + *bias_absmax >= |out|; *out_sinv = 2^-k).
 *   mvg_linear_dgrad_split: dx = (dy W) [* (relu_mask_sp > 0)] [+ addend], out_absmax as above.
 *   mvg_linear_wgrad_split: dw (+)= dy^T x with BOTH operands scaled (x_sinv may be NULL).
 *   mvg_split_colsum: g [rows][cols] fp32 -> sp times the 2^k that *absmax (max |g| bits) allows, *out_sinv = 2^-k, and
 *     db (may be NULL) (+)= the column sums of g - the Linear's bias gradient - in one pass (cols % 32 == 0).
 *   mvg_fuse_unbuild: the backward of mvg_fuse_build_split for one iteration: dfeat [segments][batch][3][nvec] =
 *     dxh[.][cf:] + sum over d with seg[d] == s of rel[d]^T @ dxn[d][cf:]; da [views][batch][cf] (+)= sum over d with
 *     vi[d] == v of (dxh + dxn)[d][:cf]; max |dfeat| into *absmax (may be NULL).  dxh / dxn: [dirs][batch][cf + 3 nvec]. */
int mvg_absmax_multi(const float *const *host_ptrs, const int64_t *host_counts, float *const *host_out_slots, int n, void *stream);
int mvg_fuse_build_split(const float *img_feat, const float *feat, const float *rel, const int32_t *row_img, const int32_t *row_src_f,
                         const int32_t *row_src_h, void *xf_sp, void *xh_sp, const float *am_img, const float *am_feat,
                         float *xf_sinv, float *xh_sinv, int rows, int cf, int nvec, void *stream);
int mvg_fuse_unbuild(const float *dxh, const float *dxn, const float *rel, const int32_t *seg, const int32_t *vi, float *dfeat, float *da,
                     int da_accumulate, int segments, int views, int dirs, int batch, int cf, int nvec, float *absmax, void *stream);
int mvg_linear_fprop_split(int rows, int fin, int fout, const void *x_sp, const float *x_sinv, const void *w_sp, const float *w_sinv,
                           const float *bias, int relu, void *out, int out_sp, float *out_sinv, const float *bias_absmax,
                           float *out_absmax, void *stream);
int mvg_linear_dgrad_split(int rows, int fin, int fout, const void *dy_sp, const float *dy_sinv, const void *wt_sp, const float *w_sinv,
                           float *dx, const float *addend, const void *relu_mask_sp, float *out_absmax, void *stream);
int mvg_linear_wgrad_split(int rows, int fin, int fout, const void *x_sp, const float *x_sinv, const void *dy_sp, const float *dy_sinv,
                           float *dw, float *workspace, int splits, int accumulate, void *stream);
int mvg_split_colsum(const float *g, int rows, int cols, const float *absmax, void *out_sp, float *out_sinv, float *db, int accumulate,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROTMVGAZE_H */
