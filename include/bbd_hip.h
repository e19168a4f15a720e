/*
 * bbd_hip.h - C ABI of the MI355X-native photometric-reprojection hot path.
 *
 * The upstream reference (kieran514/baseboostdepth) has no FFI / plugin interface: its
 * seam is Python (`Trainer.process_batch`, trainer.py:286-308).  This header is the drop-in
 * boundary the build adds underneath that seam: every entry point replaces a group of
 * eager PyTorch calls in the reference and cites them.  Rules for every function:
 *
 *   - plain pointers and sizes only (no torch types); all pointers are DEVICE pointers
 *     unless the parameter is documented "host";
 *   - no allocation, no synchronisation: work is enqueued on `stream` (a hipStream_t passed
 *     as void*; NULL = the null stream) and the call returns immediately;
 *   - return value 0 = enqueued, >0 = hipError_t from the launch, <0 = BBD_E_* argument error;
 *   - tensors are fp32, contiguous, NCHW like the reference's batch dict.
 *
 * Python binding: baseboostdepth_amd/_lib.py (ctypes).  See INTEGRATION.md for the stub a
 * reference maintainer would add.
 */
#ifndef BBD_HIP_H
#define BBD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY incompatible change of a prototype or of a scratch-size contract; the Python binding refuses a
 * library whose number differs (a stale build selected through BBD_HIP_LIB, or a C caller compiled against an older
 * header, would otherwise mis-marshal pointers into a kernel).  1 = round 1; 2 = bbd_bn_act_bwd gained `beta`, BN
 * scratch sizing changed (round 2, was not bumped then); 3 = round 3 (backward tiling / scratch contract); 4 = the
 * depth-wise token weight gradient's scratch contract (one partial row per workgroup) and `add_input` of
 * bbd_dwconv_tokens_fwd became a bit field (round 3, with the bbd_token_ln_* / bbd_colsum additions); 5 = round 4 (the
 * work-item table of the fused launches, per-row `invert` of the pose matrices, the multi-scale smoothness launches);
 * 6 = round 5: bbd_bn_act_grouped_fwd gained `untracked_groups` (the padding group of the batched pose pass);
 * 7 = round 6: bbd_bn_act_grouped_dev_fwd / _bwd (group table resident on the device: launches whose arguments do not
 * depend on the batch signature). */
#define BBD_ABI_VERSION 7

/* Source frames live in separate tensors, one per frame id (inputs[("color", f, 0)],
 * trainer.py:428).  A "slot" indexes a host array of their base pointers. */
#define BBD_MAX_FRAME_SLOTS 16   /* f = -7..7 and 's' */
#define BBD_MAX_CAND 20          /* reference maximum is 18 (trainer.py:1025-1042) */

#define BBD_KIND_WARP 0          /* reprojection or error-induced candidate */
#define BBD_KIND_IDENT 1         /* identity candidate (+ noise) */
#define BBD_FLAG_NO_POSE_GRAD 0x100 /* warp row whose pose is a constant (stereo_T, T_error) */
/* Optional pairing hint of a WARP candidate, bits 16-23 of `kind`: 1 + the index (arg-min id) of another warp
 * candidate of the same sample that samples the SAME source image (the error-induced warp of a frame and its true-pose
 * warp, trainer.py:439-442).  The backward takes two candidates per pass and prefers the hinted partner (their gathers
 * share cache lines); 0 = no hint (the next candidate in id order is taken).  A speed hint: min-loss maps, arg-min ids,
 * depth and warped images are bit-for-bit the same with any hint (the running minimum is order-free); the backward adds
 * a pixel's per-candidate depth gradients in visiting order, so disparity gradients are deterministic for a given table
 * and equal across tables up to fp32 summation order (a pixel collects gradient from every candidate that won one of
 * its 3x3 neighbours). */
#define BBD_PAIR_SHIFT 16

#define BBD_E_BADARG (-1)
#define BBD_E_TOOMANY (-2)

/* One arg-min candidate of one target sample; candidates are stored [B][BBD_MAX_CAND] in
 * the reference's arg-min id order (trainer.py:549-555, 983-1100).
 *   WARP : slot/row select the source image frames[slot] + row*3*H*W; `pose` is the row of
 *          the pose table; kind may be OR-ed with BBD_FLAG_NO_POSE_GRAD.
 *   IDENT: `row` is the row of the identity-loss tensor [NI,H,W]. */
typedef struct bbd_cand {
  int32_t kind;
  int32_t slot;
  int32_t row;
  int32_t pose;
} bbd_cand_t;

/* Row of the pose table (40 floats) of one (warp job, sample):  K[:3,:] row-major (12),
 * T row-major (16), inv_K[:3,:3] row-major (9), 3 pad.  bbd_pose_expand turns it into the
 * projection table with the reference's CPU rounding order; replaces the per-warp matmul chain
 * of layers.py:163-165 and :182-185. */
#define BBD_POSE_STRIDE 40

/* Row of the projection table (24 floats) the fused kernels read: P = (K@T)[:3,:] row-major (12),
 * inv_K[:3,:3] row-major (9), 3 pad.  Produced from the pose table by bbd_pose_expand. */
#define BBD_PROJ_STRIDE 24

/* Geometry of the launch tiling, so callers can size scratch buffers. */
int bbd_abi_version(void);
int bbd_tile_w(void);
int bbd_tile_h(void);
int bbd_num_tiles(int H, int W);
/* tiles of the fused FORWARD launch: sizes partial [S, B, bbd_num_tiles_fwd(H,W)]; bbd_num_tiles = the 64x16 tiling
 * of the identity / SSIM-map kernels */
int bbd_num_tiles_fwd(int H, int W);
/* tiles of the BACKWARD launch: sizes grad_proj [S, NP, bbd_num_tiles_bwd(H,W), 12] */
int bbd_num_tiles_bwd(int H, int W);

/* Pose composition of a step in one launch each way (replaces the per-frame Python loops of
 * trainer.py:359-388 (incremental chain), :376-377 / :403-405 (T_error) and :415-418 (partial swap)).
 *   steps  [R,4,4]   every pose-network result of the step (transformation_from_parameters output), row-major
 *   table  int32 [NO][BBD_COMPOSE_STRIDE]: {n, i0..i6, direct, flags, 0, 0} per composed matrix:
 *            out = steps[i0] @ steps[i1] @ ... @ steps[i(n-1)]   (n = 0: identity; products rounded like torch.matmul
 *            on the reference's CPU path); flags & BBD_COMPOSE_REPLACE: the 4th column is then taken from
 *            steps[direct]; flags & BBD_COMPOSE_ERROR: out[:3,3] /= pose_error and the row carries no gradient
 *   out    [NO,4,4]
 * Backward: refs_off int32 [R+1], refs int32 [.][2] = for each step row the (output row, chain position | -1 = direct)
 * pairs that read it; grad_steps [R,4,4] is written whole (zeros for unreferenced rows).  Deterministic. */
#define BBD_COMPOSE_STRIDE 12
#define BBD_COMPOSE_ERROR 1
#define BBD_COMPOSE_REPLACE 2
int bbd_pose_compose_fwd(const float* steps, const int32_t* table, float* out, int NO, double pose_error, void* stream);
int bbd_pose_compose_bwd(const float* steps, const int32_t* table, const int32_t* refs_off, const int32_t* refs,
                         const float* grad_out, float* grad_steps, int R, void* stream);

/* Pose table [NP,40] -> projection table [NP,24]: P = (K@T)[:3,:] formed with the rounding order of
 * the reference's CPU torch.matmul (layers.py:182), inv_K[:3,:3] copied. */
int bbd_pose_expand(const float* pose, float* proj, int NP, void* stream);

/* Identity photometric loss  0.85*mean_c SSIM(src, tgt) + 0.15*mean_c |tgt - src|
 * for NI (target sample, source image) pairs.  Replaces trainer.py:501-508
 * (compute_reprojection_loss on un-warped sources; layers.py:219-249).
 *   frames  host array[BBD_MAX_FRAME_SLOTS] of device pointers to [n_f,3,H,W] tensors
 *   target  [B,3,H,W]
 *   items   device int32 [NI][4] = {target sample, slot, row, 0}
 *   ident   out [NI,H,W]                                                            */
int bbd_identity_loss_fwd(const void* const* frames, const float* target,
                          const int32_t* items, int NI, float* ident,
                          int H, int W, int no_ssim, void* stream);
/* Grouped form of the same (what the trainer calls): the items of group g are items[group_off[g] .. group_off[g+1]) and
 * share ONE target sample (items[first].target): the target tile and its window statistics are set up once per group
 * and tile instead of once per item, the next item's source travels while the current one's SSIM runs.
 *   group_off  device int32 [G+1]        Same results, bit for bit. */
int bbd_identity_loss_grouped_fwd(const void* const* frames, const float* target, const int32_t* items,
                                  const int32_t* group_off, int G, float* ident, int H, int W, int no_ssim,
                                  void* stream);

/* Fused forward:  back-project -> project -> bilinear border sample -> SSIM+L1 ->
 * per-pixel min/arg-min over the sample's candidate list, for S scales x B samples.
 * Replaces trainer.py:421-442 (warping_block), :477-486, :525-557 and x_min_opt :983-1100,
 * i.e. layers.BackprojectDepth/Project3D/SSIM + F.grid_sample + cat + torch.min.
 *   depth      [S,B,H,W]   full-resolution depth per scale (outputs[("depth",0,s)])
 *   proj       [NP,24]     projection table from bbd_pose_expand (see BBD_PROJ_STRIDE)
 *   ident      [NI,H,W]    identity losses from bbd_identity_loss_fwd
 *   noise      [B,H,W]     identity noise per sample (trainer.py:518-523), may be NULL
 *   cand/ncand [B][BBD_MAX_CAND] / [B]
 *   min_loss   out [S,B,H,W]   value of the winning candidate (to_optimise)
 *   argmin     out [S,B,H,W]   u8 id of the winning candidate (ident, trainer.py:546)
 *   partial    out [S,B,ntiles] per-tile sums of min_loss (deterministic 2-stage mean)
 *   warped     out [S,NP,3,H,W] or NULL: materialise outputs[("color"/"color_D",f,s)]      */
int bbd_warp_ssim_min_fwd(const void* const* frames, const float* target, const float* depth,
                          const float* proj, const float* ident, const float* noise,
                          const bbd_cand_t* cand, const int32_t* ncand,
                          float* min_loss, uint8_t* argmin, float* partial, float* warped,
                          int S, int B, int NP, int H, int W, int no_ssim, void* stream);

/* Fused backward of the above w.r.t. depth and P = (K@T)[:3,:] of every pose-table row.
 * Replaces the autograd of min.dim, avg_pool2d, reflection_pad2d, grid_sampler_2d and bmm
 * (SURVEY.md Appendix A4-A6).  Gradient flows only to the arg-min candidate of each pixel.
 *   gscale       [S]           d loss / d min_loss[s,...] (one scalar per scale, device)
 *   grad_depth   out [S,B,H,W]
 *   grad_proj    out [S,NP,ntiles,12]  per-tile partial sums of dL/dP (caller reduces over
 *                               tiles and applies dL/dT = K[:3,:]^T dL/dP; rows never visited
 *                               are written as zeros)                                        */
int bbd_warp_ssim_min_bwd(const void* const* frames, const float* target, const float* depth,
                          const float* proj, const bbd_cand_t* cand, const int32_t* ncand,
                          const uint8_t* argmin, const float* gscale,
                          float* grad_depth, float* grad_proj,
                          int S, int B, int NP, int H, int W, int no_ssim, void* stream);

/* Disparity-mode forms of the two launches: instead of depth planes they take the decoder's
 * disparity maps themselves - disp = host array of S device pointers ([B,1,h_s,w_s] each), disp_hw = host array
 * {h_0,w_0,h_1,w_1,...} - and evaluate F.interpolate(bilinear, align_corners=False) + layers.disp_to_depth
 * (trainer.py:455-461, layers.py:13-22) per staged pixel, bit-identical to bbd_disp_to_depth_fwd.  No depth
 * buffer is read; depth_out (optional, [S,B,H,W]) receives outputs[("depth",0,s)] as a by-product.
 * The backward takes `depth` = the forward's depth_out (or NULL: it then re-evaluates the up-sampling per staged
 * pixel, +5 % instructions) and hands back grad_up [S,B,H,W] = d loss / d (up-sampled disparity); for a scale at full
 * resolution that IS the disparity gradient, the reduced scales go through bbd_disp_upsample_adjoint
 * (ONE launch for all of them; n <= 4 entries: grad_up[i] -> grad_disp[i] [B,h_i,w_i]; deterministic).
 * work_items (device [S*B*ntiles][2] int32 from bbd_fused_work_items, or NULL = grid order decoded in the kernel): which
 * (sample, scale, tile) each workgroup takes.  A speed choice: results do not depend on it.                 */
int bbd_warp_ssim_min_disp_fwd(const void* const* frames, const float* target, const void* const* disp,
                               const int32_t* disp_hw, double min_depth, double max_depth, const float* proj,
                               const float* ident, const float* noise, const bbd_cand_t* cand, const int32_t* ncand,
                               const int32_t* work_items, float* min_loss, uint8_t* argmin, float* partial, float* warped,
                               float* depth_out, int S, int B, int NP, int H, int W, int no_ssim, void* stream);
int bbd_warp_ssim_min_disp_bwd(const void* const* frames, const float* target, const void* const* disp,
                               const int32_t* disp_hw, double min_depth, double max_depth, const float* depth,
                               const float* proj, const bbd_cand_t* cand, const int32_t* ncand,
                               const int32_t* work_items, const uint8_t* argmin, const float* gscale, float* grad_up,
                               float* grad_proj, int S, int B, int NP, int H, int W, int no_ssim, void* stream);
/* Work order of the fused launches (host function, fills a HOST buffer the caller uploads once per batch signature):
 * out [S*B*ntiles][2] for the forward (backward = 0, ntiles = bbd_num_tiles_fwd) or the backward (1, bbd_num_tiles_bwd)
 * launch.  Slab order: every XCD walks one slab of consecutive tiles - for every sample in `sample_order` (host [B], a
 * permutation of 0..B-1, or NULL = 0..B-1; list the samples with the most candidates first so that a batch mixing 8-,
 * 14- and 18-candidate samples, mono_dataset.py:87-109, ends on its cheap workgroups), for every scale, the slab's
 * tiles - so the scales of a sample share one XCD's L2 and every XCD gets the same work.                    */
int bbd_fused_work_items(int B, int S, int H, int W, int backward, const int32_t* sample_order, int32_t* out);
int bbd_disp_upsample_adjoint(const void* const* grad_up, const int32_t* disp_hw, void* const* grad_disp, int n, int B,
                              int H, int W, void* stream);

/* disp -> full-resolution depth:  bilinear upsample (align_corners=False) then
 * depth = 1 / (1/max + (1/min - 1/max) * disp).  Replaces trainer.py:455-461 and
 * layers.disp_to_depth (layers.py:13-22).  disp [B,h,w] -> depth [B,H,W].           */
int bbd_disp_to_depth_fwd(const float* disp, float* depth, int B, int h, int w, int H, int W,
                          double min_depth, double max_depth, void* stream);
/* grad_disp [B,h,w] is OVERWRITTEN with the adjoint (gather form, deterministic).  `depth` is the
 * forward's output [B,H,W] (d depth/d disp_up = -span*depth^2) or NULL to recompute it from disp. */
int bbd_disp_to_depth_bwd(const float* disp, const float* depth, const float* grad_depth, float* grad_disp,
                          int B, int h, int w, int H, int W,
                          double min_depth, double max_depth, void* stream);

/* Pose matrix from the pose head's output: layers.transformation_from_parameters
 * (layers.py:25-100: rot_from_axisangle, get_translation_matrix, T@R or R^T@T(-t)).
 *   axisangle, translation [n,3] -> M [n,4,4];  bwd: grad_M [n,4,4] -> grads [n,3] each.
 *   invert_rows (device [n] int32, or NULL): per-row `invert` flag that replaces the scalar one, so that the poses of
 *   BOTH signs of a step (trainer.py:360,384,402: invert for negative frame ids) are one launch each way.         */
int bbd_pose_matrix_fwd(const float* axisangle, const float* translation, float* M, int n, int invert,
                        const int32_t* invert_rows, void* stream);
int bbd_pose_matrix_bwd(const float* axisangle, const float* translation, const float* grad_M,
                        float* grad_axisangle, float* grad_translation, int n, int invert,
                        const int32_t* invert_rows, void* stream);

/* Edge-aware smoothness of the mean-normalised disparity: layers.get_smooth_loss
 * (layers.py:203-216) applied to disp / (mean_{H,W}(disp) + 1e-7) as in trainer.py:560-563.
 *   disp [B,h,w], img [B,3,h,w]
 *   fwd : mean_disp out [B, bbd_smooth_chunks()] partial sums of disp (mean = sum / (h*w));
 *         sums out [B, bbd_smooth_chunks(), 2] partial sums of the x- and
 *         y-terms; smooth = sum(x-terms)/(B*h*(w-1)) + sum(y-terms)/(B*(h-1)*w)
 *   bwd : gscale [1] device scalar dL/d(smooth); dots scratch [B, bbd_smooth_chunks()];
 *         grad_disp out [B,h,w] (overwritten).  Deterministic (fixed reduction order).        */
int bbd_smooth_chunks(void);
int bbd_smooth_loss_fwd(const float* disp, const float* img, float* mean_disp, float* sums,
                        int B, int h, int w, void* stream);
int bbd_smooth_loss_bwd(const float* disp, const float* img, const float* mean_disp,
                        const float* gscale, float* grad_disp, float* dots,
                        int B, int h, int w, void* stream);
/* The same for ALL scales of a step (trainer.py:527-564 loops over opt.scales) in one launch pair each way: disp / img /
 * grad_disp = host arrays of S (<= 4) device pointers, hw = host {h_0,w_0,h_1,w_1,...}; mean_disp [S,B,chunks],
 * sums [S,B,chunks,2], dots [S,B,chunks], gscale [S] (device).  Same kernels, same reduction order as the single-scale
 * form (which is the S = 1 case).                                                                              */
int bbd_smooth_loss_multi_fwd(const void* const* disp, const void* const* img, const int32_t* hw, float* mean_disp,
                              float* sums, int S, int B, void* stream);
int bbd_smooth_loss_multi_bwd(const void* const* disp, const void* const* img, const int32_t* hw,
                              const float* mean_disp, const float* gscale, void* const* grad_disp, float* dots,
                              int S, int B, void* stream);

/* Stand-alone kernels behind the reference's layer classes (the training step uses the fused entry
 * points above and never materialises these tensors; callers written against the reference's
 * `layers` module - including its own trainer - get differentiable modules through these).
 *   bbd_backproject_fwd : layers.BackprojectDepth.forward (layers.py:160-167)
 *                         depth [n,H,W], inv_K [n,4,4] -> points [n,4,H*W]
 *   bbd_project3d_fwd   : layers.Project3D.forward (layers.py:181-195)
 *                         points [n,4,H*W], K [n,4,4], T [n,4,4] -> grid [n,H,W,2] in [-1,1]
 *   bbd_ssim_fwd        : layers.SSIM.forward (layers.py:235-249)  x,y [n,3,H,W] -> [n,3,H,W] */
int bbd_backproject_fwd(const float* depth, const float* inv_K, float* points, int n, int H, int W,
                        void* stream);
int bbd_project3d_fwd(const float* points, const float* K, const float* T, float* grid,
                      int n, int H, int W, double eps, void* stream);
int bbd_ssim_fwd(const float* x, const float* y, float* out, int n, int H, int W, void* stream);

/* Their backward passes (the reference's layers are plain autograd nn.Modules, layers.py:136-249, and its
 * trainer back-propagates through them, trainer.py:434-442, 477-486).  Closed forms of SURVEY Appendix A.
 *   bbd_backproject_bwd : grad_points [n,4,H*W] -> grad_depth [n,H,W]          (inv_K carries no gradient)
 *   bbd_project3d_bwd   : grad_grid [n,H,W,2] -> grad_points [n,4,H*W] and gp_partial
 *                         [n, bbd_project3d_bwd_blocks(), 12]: per-workgroup partial sums of dL/dP,
 *                         P = (K@T)[:3,:]; the caller sums them (fixed order => deterministic) and
 *                         forms dL/dT = K[:3,:]^T dP, dL/dK[:3,:] = dP T^T
 *   bbd_ssim_bwd        : grad_out [n,3,H,W] -> grad_x [n,3,H,W]; the SSIM expression is symmetric in
 *                         its two arguments, so d/dy is the same call with x and y exchanged          */
int bbd_backproject_bwd(const float* grad_points, const float* inv_K, float* grad_depth, int n, int H, int W,
                        void* stream);
int bbd_project3d_bwd_blocks(void);
int bbd_project3d_bwd(const float* points, const float* K, const float* T, const float* grad_grid,
                      float* grad_points, float* gp_partial, int n, int H, int W, double eps, void* stream);
int bbd_ssim_bwd(const float* x, const float* y, const float* grad_out, float* grad_x, int n, int H, int W,
                 void* stream);

/* Validation metrics on the device: one workgroup per image.
 *   flags 0                       : Trainer.compute_depth_losses, KITTI branch (trainer.py:594-617):
 *                                   pred = depth [n,h,w]; F.interpolate(bilinear, align_corners=False)
 *                                   to the ground-truth size, clamp to [clamp_lo, clamp_hi], mask
 *                                   (min_depth < gt < max_depth inside the window), torch.median
 *                                   scaling (lower median), clamp, layers.compute_depth_errors.
 *   BBD_EVAL_PRED_IS_DISP         : evaluate_depth.py:244-297: pred = disparity, cv2.resize (linear)
 *                                   then 1/disp, times scale_factor (opt.pred_depth_scale_factor).
 *   BBD_EVAL_MEDIAN_MIDPOINT      : np.median (mean of the two middle values) instead of torch.median.
 *   BBD_EVAL_NO_MEDIAN_SCALING    : opt.disable_median_scaling (stereo evaluation).
 * gt is a ragged buffer; desc[i] = {offset_lo, offset_hi (elements), GH, GW, r0, r1, c0, c1} with
 * [r0,r1) x [c0,c1) the crop window (Garg crop, trainer.py:603-606; whole image = 0,GH,0,GW).
 * out[i] = {abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, ratio, median_gt, median_pred, count, 0}. */
#define BBD_EVAL_DESC 8
#define BBD_EVAL_OUT 12
#define BBD_EVAL_PRED_IS_DISP 1
#define BBD_EVAL_MEDIAN_MIDPOINT 2
#define BBD_EVAL_NO_MEDIAN_SCALING 4
int bbd_depth_metrics(const float* pred, const float* gt, const int32_t* desc, float* out, int n, int h,
                      int w, double min_depth, double max_depth, double clamp_lo, double clamp_hi,
                      double scale_factor, int flags, void* stream);

/* ---- Loader image pipeline: replaces the per-item Pillow/torchvision work of
 * datasets/mono_dataset.py:186-205 (Resize(LANCZOS) chain, ColorJitter, ToTensor) and the stacking of
 * Trainer.custom_collate (trainer.py:867-886).  Images are uint8 HWC (as PIL decodes them) inside one
 * device buffer; every job addresses its source and destination by offset, so results land directly
 * in the rows of the collated batch tensors.  Bit-exact against Pillow (bbd_image_math.h).
 *
 * Resample job (int32[BBD_RESAMPLE_JOB]): src byte offset lo,hi | dst byte offset lo,hi | in_h, in_w,
 * out_size, ksize | coef_off, bounds_off (int32 elements into coef / bounds), flags, pad.
 *   bbd_resample_h_u8: [in_h,in_w,C] -> [in_h,out_size,C]   (flags & BBD_RESAMPLE_FLIP mirrors the source
 *                      columns = PIL transpose(FLIP_LEFT_RIGHT) before the resize, kitti_dataset.py:58-59)
 *   bbd_resample_v_u8: [in_h,in_w,C] -> [out_size,in_w,C]
 * coef[out][ksize] are Pillow's 22-bit fixed-point filter taps, bounds[out] = (first source index, count);
 * the host builds them (baseboostdepth_amd/imageops.py:resample_table, Resample.c precompute_coeffs). */
#define BBD_RESAMPLE_JOB 12
#define BBD_RESAMPLE_FLIP 1
int bbd_resample_h_u8(const uint8_t* src, uint8_t* dst, const int32_t* jobs, int n_jobs, int max_in_h,
                      const int32_t* coef, const int32_t* bounds, int channels, void* stream);
int bbd_resample_v_u8(const uint8_t* src, uint8_t* dst, const int32_t* jobs, int n_jobs, int max_out_h,
                      int max_row_bytes, const int32_t* coef, const int32_t* bounds, int channels, void* stream);

/* torchvision ColorJitter (functional_pil, order and factors drawn by the host per image) followed by
 * ToTensor, for n_jobs uint8 [H,W,3] images -> fp32 [3,H,W] at dst + offset (floats).
 * Jitter job (int32[BBD_JITTER_JOB]): src byte offset lo,hi | dst float offset lo,hi | op[4] in
 * application order (0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none) | param[4]: float bits of
 * the factor, or for hue the uint8 offset int(hue_factor*255) & 255.  lsum_scratch: uint32[n_jobs]. */
#define BBD_JITTER_JOB 12
int bbd_color_jitter_u8(const uint8_t* src, float* dst, const int32_t* jobs, int n_jobs, int H, int W,
                        uint32_t* lsum_scratch, void* stream);

/* ToTensor only. Convert job (int32[BBD_CONVERT_JOB]): src byte offset lo,hi | dst float offset lo,hi. */
#define BBD_CONVERT_JOB 4
int bbd_u8_to_float_chw(const uint8_t* src, float* dst, const int32_t* jobs, int n_jobs, int H, int W,
                        void* stream);

/* ---- Encoder glue: training-mode BatchNorm2d (+ residual) (+ ReLU) of the torchvision-layout ResNet
 * blocks (networks/resnet_encoder.py:12-91; `bn(conv(x))`, `relu(bn(conv(x)))`, `relu(bn(conv(x)) + identity)`),
 * same definition as F.batch_norm(training=True): biased variance to normalise, unbiased for running_var.
 * x, y, residual, grads are [N,C,HW] fp32 (NCHW contiguous); scratch holds bbd_bn_scratch_doubles(N,C,HW)
 * doubles; running_mean/var may both be NULL (no tracking); relu != 0 applies max(0, .).
 *   fwd: writes y, save_mean[C], save_invstd[C]; updates running stats with `momentum` and increments
 *        *num_batches_tracked (int64 device scalar, may be NULL)
 *   bwd: the ReLU mask (relu != 0) is y > 0; a forward WITHOUT a residual may pass y = NULL and `beta`: the mask is
 *        then re-derived from x with the forward's own expression (identical bits), which saves one read of the
 *        activation in each of the two backward launches; grad_residual may be NULL */
int bbd_bn_scratch_doubles(int N, int C, int HW);
/* Grouped form: the N samples are G consecutive call groups, rows [group_rows[g], group_rows[g+1]) (HOST array of
 * G+1 ints, group_rows[0] = 0, group_rows[G] = N, G <= BBD_BN_MAX_GROUPS), each normalised with its OWN batch
 * statistics - bit for bit what G separate calls of the layer on the G sub-batches compute, in one pass.  This is how
 * the pose network's per-frame calls (trainer.py:348-418: up to 26 calls per step on <= 12 samples each) run as one
 * batched pass.  save_mean / save_invstd are [G,C]; the running statistics receive G momentum updates in group order
 * and num_batches_tracked += G; grad_gamma / grad_beta are summed over the groups.  scratch:
 * bbd_bn_grouped_scratch_doubles(largest group, G, C, HW) doubles.
 * untracked_groups (ABI 6; 0 <= . < G): the LAST that many groups are normalised like the others but take no part in the
 * running statistics or num_batches_tracked - padding rows.  Boosted batches change the batched pose pass's row count
 * almost every step (mono_dataset.py:87-109), and every new row count is a new convolution problem for MIOpen (tens of
 * seconds of solver compilation at first sight); the trainer rounds the pass up to a row count with shipped find results (else a multiple of 32) with one
 * trailing group of zero rows whose outputs nobody reads (their gradients are exact zeros). */
#define BBD_BN_MAX_GROUPS 32
int bbd_bn_grouped_scratch_doubles(int max_group_rows, int G, int C, int HW);
int bbd_bn_act_grouped_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                           float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                           long long* num_batches_tracked, double* scratch, const int32_t* group_rows, int G,
                           int untracked_groups, int N, int C, int HW, double eps, double momentum, int relu, void* stream);
int bbd_bn_act_grouped_bwd(const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual,
                           float* grad_gamma, float* grad_beta, double* scratch, const int32_t* group_rows, int G, int N,
                           int C, int HW, int relu, void* stream);
/* The same two launches with the group table RESIDENT ON THE DEVICE (ABI 7): group_table = device int32
 * [BBD_BN_MAX_GROUPS + 2] = { rows[0] = 0, rows[1], ..., rows[G] = N, (unused up to index BBD_BN_MAX_GROUPS),
 * [BBD_BN_MAX_GROUPS + 1] = number of TRACKED groups }.  Groups may be EMPTY (rows[g+1] == rows[g]: nothing is read or
 * written for them); the groups from the tracked count on are the padding of bbd_bn_act_grouped_fwd's `untracked_groups`.
 * max_group_rows (host) bounds every group's row count (it sizes the grid and the scratch:
 * bbd_bn_grouped_scratch_doubles(max_group_rows, G, C, HW)); a group holds its own statistics exactly as in the
 * host-table form - bit for bit the same numbers.  Why: the boosted recipe redraws every sample's frame set per item
 * (mono_dataset.py:87-109), so the pose pass's call groups change every step; with the table on the device NOTHING in the
 * launch arguments depends on the batch signature, and one captured step graph serves every ordering that shares the
 * padded row count (the table is a section of the step's one table upload). */
int bbd_bn_act_grouped_dev_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                               float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                               long long* num_batches_tracked, double* scratch, const int32_t* group_table, int G,
                               int max_group_rows, int N, int C, int HW, double eps, double momentum, int relu,
                               void* stream);
int bbd_bn_act_grouped_dev_bwd(const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual,
                               float* grad_gamma, float* grad_beta, double* scratch, const int32_t* group_table, int G,
                               int max_group_rows, int N, int C, int HW, int relu, void* stream);
int bbd_bn_act_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                   float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                   long long* num_batches_tracked, double* scratch, int N, int C, int HW, double eps,
                   double momentum, int relu, void* stream);
int bbd_bn_act_bwd(const float* x, const float* y, const float* grad_y, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd, float* grad_x, float* grad_residual,
                   float* grad_gamma, float* grad_beta, double* scratch, int N, int C, int HW, int relu,
                   void* stream);

/* ReflectionPad2d(1) in front of every decoder convolution (layers.py:118-133) and the stem's
 * MaxPool2d(kernel 3, stride 2, padding 1) (networks/resnet_encoder.py:28), forward and gather-form
 * (atomic-free, deterministic) backward; `planes` = N*C, tensors NCHW fp32.
 *   pad : in [planes,H,W] -> out [planes,H+2,W+2]
 *   pool: in [planes,H,W] -> out [planes,OH,OW], OH = (H-1)/2+1; `code` u8 [planes,OH,OW] = position
 *         0..8 of the maximum inside its window (ATen's tie / NaN rule), consumed by the backward */
int bbd_reflect_pad1_fwd(const float* in, float* out, int planes, int H, int W, void* stream);
int bbd_reflect_pad1_bwd(const float* grad_out, float* grad_in, int planes, int H, int W, void* stream);
int bbd_maxpool3s2_fwd(const float* in, float* out, uint8_t* code, int planes, int H, int W, void* stream);
int bbd_maxpool3s2_bwd(const float* grad_out, const uint8_t* code, float* grad_in, int planes, int H, int W,
                       void* stream);

/* Disparity head of the depth decoder: layers.Conv3x3(C, 1) = ReflectionPad2d(1) + Conv2d(C, 1, 3) + bias
 * (layers.py:118-133, networks/depth_decoder.py:38-39), forward and backward without a padded copy.
 * x [N,C,H,W], weight [C,3,3] (= conv.weight[0]), bias [1] or NULL, y / grad_y [N,1,H,W];
 * grad_x or grad_weight(+grad_bias) may be NULL; scratch holds bbd_dispconv_scratch_doubles(C) doubles. */
int bbd_dispconv_scratch_doubles(int C);
int bbd_dispconv_fwd(const float* x, const float* weight, const float* bias, float* y, int N, int C, int H, int W,
                     void* stream);
int bbd_dispconv_bwd(const float* x, const float* weight, const float* grad_y, float* grad_x, float* grad_weight,
                     float* grad_bias, double* scratch, int N, int C, int H, int W, void* stream);

/* Input of the batched pose pass in the pooled form of the step: out [R, 2, chw] row r = the image pair
 * (pool[idx_a[r]], pool[idx_b[r]]) of the frame pool [F, chw] (chw = 3*H*W floats, multiple of 4), every texel (v - sub) * mul.
 * Replaces the reference's per-call `torch.cat([frame_a, frame_b], 1)` (trainer.py:352-358, 394-400) over masked sub-batches
 * plus the pose encoder's input normalisation `(x - 0.45) / 0.225` (networks/resnet_encoder.py:83; PyTorch-ROCm divides by a
 * Python scalar as a multiplication by its float reciprocal: pass mul = (float)1 / (float)0.225): same bits, one pass.
 * idx_a / idx_b: device int32 [R] (sections of the step's static table buffer).  sub = 0, mul = 1: a plain gather. */
int bbd_gather_pairs(const float* pool, const int32_t* idx_a, const int32_t* idx_b, float* out, int R, long chw, double sub,
                     double mul, void* stream);

/* Measurement aid (bench.py, SURVEY 8d "on-box measured stream-copy ceiling"): dst[i] = src[i] as a float4 grid-stride
 * copy of n_floats floats (multiple of 4, both pointers 16-byte aligned) with `unroll` (1, 2, 4, 8) independent 16-byte
 * loads per thread in flight; 2 * 4 * n_floats bytes of HBM traffic. */
int bbd_stream_copy(const float* src, float* dst, long n_floats, int unroll, void* stream);

/* Device self-test: the kernels replace hipcc's IEEE division sequence by a cheaper one that is
 * exact for moderate exponents (bbd_math.h).  Runs blocks*256*iters random operand tuples through
 * both and adds the number of bit mismatches to *mismatches (device int32, caller zeroes it). */
int bbd_selftest_div(int blocks, int iters, unsigned seed, int32_t* mismatches, void* stream);

/* Decoder glue (reference networks/depth_decoder.py:44-50, layers.py:103-133), fp32 NCHW:
 *   bbd_upcat_pad1_fwd : out [N, C1+C2, 2h+2, 2w+2] = ReflectionPad2d(1)(cat(nearest_x2(x [N,C1,h,w]), skip [N,C2,2h,2w]))
 *                        in one pass (skip may be NULL with C2 = 0); _bwd: grad_out -> grad_x, grad_skip (gather form,
 *                        no atomics, deterministic).  N*(C1+C2) <= 65535.
 *   bbd_bias_elu_fwd   : y <- ELU(y + bias[c]) in place on a convolution's output [N,C,HW] (HW % 4 == 0)
 *   bbd_bias_elu_bwd   : grad_x = grad_y * ELU'(from the saved output y), grad_bias[c] = sum grad_x (fp64 partial sums,
 *                        fixed order); scratch = bbd_bias_elu_scratch_doubles(N,C,HW) doubles.                       */
int bbd_upcat_pad1_fwd(const float* x, const float* skip, float* out, int N, int C1, int C2, int h, int w, void* stream);
int bbd_upcat_pad1_bwd(const float* grad_out, float* grad_x, float* grad_skip, int N, int C1, int C2, int h, int w,
                       void* stream);
int bbd_bias_elu_scratch_doubles(int N, int C, int HW);
int bbd_bias_elu_fwd(float* y, const float* bias, int N, int C, int HW, void* stream);
int bbd_bias_elu_bwd(const float* y, const float* grad_y, float* grad_x, float* grad_bias, double* scratch, int N, int C,
                     int HW, void* stream);

/* ---- MonoViT encoder (BASELINE configs[4]): depth-wise convolution on token-layout activations ----------
 * The position encodings of the reference's MPViT (networksvit/mpvit.py:240-330: ConvPosEnc, ConvRelPosEnc)
 * are depth-wise k x k convolutions (k in {3,5,7}, stride 1, zero padding k/2) over the token matrix viewed
 * as an image.  Tokens are [B, H*W, C] row-major (= NHWC); x / y / grad_y may be channel slices of wider
 * rows: `*_row` is the number of floats between consecutive tokens and the pointer addresses the slice's
 * first channel, C is the number of channels of the slice.  weight [C,k,k] (nn.Conv2d's [C,1,k,k]), bias [C]
 * or NULL.
 *   fwd   : y = bias + conv(x) (+ x when add_input & 1; added to y's previous contents when add_input & 2).
 *           flip = 1 mirrors the taps: with x := grad_y, bias NULL this is the data gradient (bit 0 carries the
 *           residual's gradient, bit 1 lets it land in a slice that already holds another contribution).
 *   wgrad : grad_weight [C,k,k], grad_bias [C] (or NULL); partial = scratch of
 *           bbd_dwconv_tokens_wgrad_scratch_floats(B,H,W,C,k) floats.  Deterministic (one partial row per
 *           workgroup, a fixed-order column sum in fp64).  accumulate != 0: added to grad_weight / grad_bias
 *           (a parameter shared by several layers - MPViT's MHCAEncoder shares its position encodings - whose
 *           launches follow each other on one stream).                                                   */
long bbd_dwconv_tokens_wgrad_scratch_floats(int B, int H, int W, int C, int k);
int bbd_dwconv_tokens_fwd(const float* x, int x_row, const float* weight, const float* bias, float* y, int y_row,
                          int B, int H, int W, int C, int k, int add_input, int flip, void* stream);
int bbd_dwconv_tokens_wgrad(const float* x, int x_row, const float* grad_y, int gy_row, float* partial,
                            float* grad_weight, float* grad_bias, int B, int H, int W, int C, int k, int accumulate,
                            void* stream);
/* The same for n_groups <= 4 consecutive-or-not channel groups with their own window sizes in ONE launch each (the head
 * groups of MPViT's ConvRelPosEnc, networksvit/mpvit.py:262-330): group g covers channels [c0[g], c0[g] + cn[g]) of the
 * x / y / grad_y rows with window k[g]; weights / biases / grad_weights / grad_biases are HOST arrays of device pointers
 * (a NULL biases array or entry: no bias).  c0 / cn / k are host arrays.                                              */
int bbd_dwconv_tokens_groups_fwd(const float* x, int x_row, float* y, int y_row, int n_groups, const int32_t* c0,
                                 const int32_t* cn, const int32_t* k, const void* const* weights, const void* const* biases,
                                 int B, int H, int W, int add_input, int flip, void* stream);
long bbd_dwconv_tokens_groups_wgrad_scratch_floats(int B, int H, int W, int n_groups, const int32_t* cn, const int32_t* k);
int bbd_dwconv_tokens_groups_wgrad(const float* x, int x_row, const float* grad_y, int gy_row, float* partial, int n_groups,
                                   const int32_t* c0, const int32_t* cn, const int32_t* k, const void* const* grad_weights,
                                   const void* const* grad_biases, int B, int H, int W, int accumulate, void* stream);

/* Residual + stochastic depth + LayerNorm on token-layout activations [rows = B*N, C] (csrc/bbd_tokens.hip), the glue of
 * the reference's MHCABlock (networksvit/mpvit.py:397-440: x = x + drop_path(branch); z = norm(x)) as one pass each way.
 *   fwd: y = x + branch * mask[row / N];  z = LayerNorm(y) * weight + bias;  stats[row] = (mean, rstd).
 *        branch == NULL: plain LayerNorm of x (y is not written);  mask == NULL: no stochastic depth;
 *        weight == NULL: residual only (z, stats, bias unused).
 *   bwd: g = grad_y + dLayerNorm(grad_z);  grad_x = g;  grad_branch = g * mask[row / N] (NULL: not wanted);
 *        grad_weight / grad_bias through `partial` (bbd_token_ln_scratch_floats(rows, C) floats): one partial row per
 *        workgroup, summed in a fixed order by a second launch (deterministic).  grad_y may be NULL.
 *   C % 4 == 0 and C <= 1024 (bbd_token_ln_supported).                                                             */
int bbd_token_ln_supported(int C);
long bbd_token_ln_scratch_floats(int rows, int C);
int bbd_token_ln_fwd(const float* x, const float* branch, const float* mask, const float* weight, const float* bias,
                     float* y, float* z, float* stats, int rows, int N, int C, double eps, void* stream);
int bbd_token_ln_bwd(const float* grad_z, const float* grad_y, const float* y, const float* stats, const float* weight,
                     const float* mask, float* grad_x, float* grad_branch, float* partial, float* grad_weight,
                     float* grad_bias, int rows, int N, int C, void* stream);

/* out[c] = sum over rows of x[row, c] for a [rows, C] matrix, C % 4 == 0: the bias gradient of the token-parallel
 * nn.Linear layers (qkv, proj, fc1, fc2 of networksvit/mpvit.py:51-78, 333-394) on [B*N, C] token activations, in two
 * launches (one partial row per workgroup, fixed-order sum of the partial rows: deterministic).
 * partial: bbd_colsum_scratch_floats(rows, C) floats.                                                                */
long bbd_colsum_scratch_floats(long rows, int C);
int bbd_colsum(const float* x, float* partial, float* out, long rows, int C, void* stream);

/* Factorised attention of MPViT (networksvit/mpvit.py:333-394) on the packed qkv activation.
 *   qkv [B, N, 3, h, Ch] = the qkv Linear's output (C = h*Ch); convv [B,N,C] = ConvRelPosEnc's conv(v)
 *   fwd : out[b,n,h,vc] = sum_kc q[b,n,h,kc] ctxs[b,h,kc,vc] + q[b,n,h,vc] convv[b,n,h,vc],
 *         ctxs = scale * softmax_N(k)^T v per head; also returns kmax / krsum [B,C] (softmax statistics of
 *         k over the tokens) and ctxs [B, h, Ch, Ch] for the backward.
 *   bwd : grad_out [B,N,C] -> grad_qkv [B,N,3C] (dq | dk | dv), grad_convv [B,N,C]; dctx = work buffer
 *         [B, h, Ch, Ch].  scratch: bbd_factor_att_scratch_floats(B,N,C,Ch) floats for both calls.
 *   Deterministic (per-token-segment partial sums combined in fixed order).  Supported shapes:
 *   bbd_factor_att_supported(C, Ch) != 0 (C*Ch <= 12288: MPViT tiny / xsmall / small).                    */
int bbd_factor_att_supported(int C, int Ch);
int bbd_factor_att_segments(int B, int N);
long bbd_factor_att_scratch_floats(int B, int N, int C, int Ch);
int bbd_factor_att_fwd(const float* qkv, const float* convv, float* kmax, float* krsum, float* ctxs, float* scratch,
                       float* out, int B, int N, int C, int Ch, double scale, void* stream);
int bbd_factor_att_bwd(const float* qkv, const float* convv, const float* kmax, const float* krsum, const float* ctxs,
                       const float* grad_out, float* dctx, float* scratch, float* grad_qkv, float* grad_convv, int B,
                       int N, int C, int Ch, double scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BBD_HIP_H */
