/*
 * mvster_hip.h -- C ABI of libmvster_hip.so, the MI355X (gfx950) implementation of the MVSTER
 * cost-volume / epipolar-attention hot path.
 *
 * The upstream reference has no native layer: the path is a sequence of PyTorch operators inside
 * models/mvs4net_utils.py.  Each entry point below replaces one such operator sequence; the
 * citation after "replaces:" is the reference code a maintainer would swap for the call
 * (see INTEGRATION.md for the Python/ctypes binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise; tensors are dense,
 *     row-major in the order written in the comment ([B,C,H,W] == NCHW contiguous);
 *   - inputs are borrowed and never written; outputs are fully overwritten (no accumulate-into);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and the
 *     call returns without synchronising; nothing is allocated, no global mutable state is kept,
 *     every function is re-entrant and may be called concurrently from several host threads;
 *   - the active HIP device must be the one that owns the pointers (the Python host sets it);
 *   - return value: MVSTER_OK (0) or an MVSTER_ERR_* code; mvster_error_string() explains it.
 *     Nothing aborts; on error no kernel has been launched.
 */
#ifndef MVSTER_HIP_H
#define MVSTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSTER_ABI_VERSION 3

#define MVSTER_OK 0
#define MVSTER_ERR_NULL_POINTER 1   /* a required pointer argument is NULL                     */
#define MVSTER_ERR_BAD_SHAPE 2      /* a dimension is <= 0, inconsistent, or too large          */
#define MVSTER_ERR_UNSUPPORTED 3    /* valid request outside what the fused kernels instantiate */
#define MVSTER_ERR_LAUNCH 4         /* the HIP runtime rejected a launch / memset               */

#define MVSTER_LAYOUT_NCHW 0        /* feature maps are [B,C,H,W] planes (the reference's layout)  */
#define MVSTER_LAYOUT_NHWC 1        /* feature maps are [B,H,W,C] (torch channels_last storage)    */
#define MVSTER_GRAD_PLANAR 16       /* flag, OR-ed into `layout` of mvster_cost_volume_bwd_ws with MVSTER_LAYOUT_NHWC: features are
                                       read channels-last but gref / gsrcs are WRITTEN as [B,C,H,W] planes (for callers whose
                                       leaves are NCHW: no gradient conversion pass).  Binned route only, else MVSTER_ERR_UNSUPPORTED */

#define MVSTER_MAX_SRC_VIEWS 32     /* source views per call of the fused kernels               */
#define MVSTER_MAX_DEPTHS 64        /* depth hypotheses per pixel (generic kernels)             */

typedef void *mvster_stream_t;

int mvster_abi_version(void);
const char *mvster_error_string(int code);

/* --------------------------------------------------------------------------------------------
 * Relative pose of a source view w.r.t. the reference view.
 *   pose[b] = 12 floats: rot (3x3 row-major) then trans (3) of  src_proj[b] @ inverse(ref_proj[b]).
 * The 4x4 inverse and product are evaluated in fp64 on the device and rounded once to fp32.
 * replaces: models/mvs4net_utils.py:32-34 (torch.inverse + matmul inside homo_warping).
 *   src_proj, ref_proj : [B,4,4]        pose : [B,12]
 */
int mvster_relative_pose(const float *src_proj, const float *ref_proj, float *pose, int B,
                         mvster_stream_t stream);

/* Same, starting from the dataset's matrices: proj_matrices [B,N,2,4,4] with [:, v, 0] = extrinsic
 * and [:, v, 1, :3, :3] = intrinsic; view 0 is the reference.  P_v = E_v with rows 0..2 replaced by
 * K_v[:3,:3] @ E_v[:3,:4]; pose[b, v-1] = P_v @ inverse(P_0) for v = 1..N-1.
 * replaces: models/mvs4net_utils.py:1030,1047-1050 + :32-34.
 *   proj_matrices : [B,N,2,4,4]         pose : [B,N-1,12]
 */
int mvster_stage_poses(const float *proj_matrices, float *pose, int B, int N, mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Feature-map layout conversion, [B,C,H,W] <-> [B,H,W,C] (C % 4 == 0, else MVSTER_ERR_UNSUPPORTED).
 * The fused cost-volume kernels gather C contiguous floats per bilinear tap from the channels-last
 * layout; a map that already is a torch channels_last tensor needs no conversion.
 * replaces: nothing in the reference (its FPN emits NCHW, models/mvs4net_utils.py:426-509).
 */
int mvster_nchw_to_nhwc(const float *src, float *dst, int B, int C, int H, int W, mvster_stream_t stream);
int mvster_nhwc_to_nchw(const float *src, float *dst, int B, int C, int H, int W, mvster_stream_t stream);
/* N maps of one shape in ONE launch (N <= MVSTER_MAX_SRC_VIEWS + 1): srcs / dsts are host arrays of N device pointers.
 * A stage converts its reference + source feature maps, and their gradients on the way back, with one launch each. */
int mvster_nchw_to_nhwc_many(const float *const *srcs, float *const *dsts, int N, int B, int C, int H, int W,
                             mvster_stream_t stream);
int mvster_nhwc_to_nchw_many(const float *const *srcs, float *const *dsts, int N, int B, int C, int H, int W,
                             mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A1  homo_warping: bilinear (zeros padding, align_corners=True) gather of src features at the
 * projection of every (reference pixel, depth hypothesis).
 * replaces: models/mvs4net_utils.py:21-67 (4-D src_fea branch).
 *   src : [B,C,Hs,Ws]   pose : [B,12]   depth : [B,D,Hr,Wr]   out : [B,C,D,Hr,Wr]
 */
int mvster_warp_fwd(const float *src, const float *pose, const float *depth, float *out, int B, int C,
                    int D, int Hr, int Wr, int Hs, int Ws, mvster_stream_t stream);

/* Gradient of mvster_warp_fwd w.r.t. src (the sampling grid carries no gradient, :31).
 *   gout : [B,C,D,Hr,Wr]   gsrc : [B,C,Hs,Ws] (overwritten; fp32 atomics, summation order varies)
 */
int mvster_warp_bwd(const float *gout, const float *pose, const float *depth, float *gsrc, int B, int C,
                    int D, int Hr, int Wr, int Hs, int Ws, mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A2  fused warp + group-wise correlation + softmax-over-depth view weighting + fusion over the
 * S source views (group_cor=True, attn_fuse_d=True -- the configuration of every reference script).
 * Never materialises the [B,C,D,H,W] warped volume.
 * replaces: models/mvs4net_utils.py:1036-1102 (the src-view loop and the normalisation).
 *   ref   : [B,C,H,W]            srcs : host array of S device pointers, each [B,C,Hs,Ws]
 *           (layout = MVSTER_LAYOUT_NCHW) or [B,H,W,C] / [B,Hs,Ws,C] (layout = MVSTER_LAYOUT_NHWC)
 *   pose  : [B,S,12]             hypo : [B,D,H,W]
 *   vol   : [B,G,D,H,W]  fused correlation volume (regnet input)
 *   wsum  : [B,D,H,W]    1e-8 + sum_s w_s   (saved for the backward pass; may be NULL)
 * Supported: C in {8,16,32,64} with G in {4,8}, C % G == 0, D in {4,8,16}; otherwise
 * MVSTER_ERR_UNSUPPORTED (the host then composes mvster_warp_fwd with device tensor ops).
 */
int mvster_cost_volume_fwd(const float *ref, const float *const *srcs, int S, const float *pose,
                           const float *hypo, float *vol, float *wsum, int B, int C, int G, int D, int H,
                           int W, int Hs, int Ws, float attn_temp, int layout, mvster_stream_t stream);

/* A2 in training mode: the same forward, which additionally takes the FIRST pass of the binned backward's counting sort
 * while it has every sample's projection at hand: bin_counts[mvster_cost_volume_bin_counts(...)] (int32, zeroed by the
 * call) receives, per (batch, view, tile of the source map, category), the number of (pixel, hypothesis) samples whose
 * upper-left tap lies in the tile; category = interior / bottom row / right column / corner of the tile, i.e. which
 * neighbouring tiles the sample's 2x2 footprint also reaches (every sample is counted exactly once).  Hand it to
 * mvster_cost_volume_bwd_ws as fwd_bin_counts.  bin_counts == NULL: plain mvster_cost_volume_fwd.  Channels-last only
 * (MVSTER_ERR_UNSUPPORTED otherwise).
 *   mvster_cost_volume_bin_counts : number of int32 counters for this shape (B * S * tiles * 4), 0 when the binned route
 *       does not exist.
 */
long long mvster_cost_volume_bin_counts(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, int layout);
int mvster_cost_volume_fwd_counted(const float *ref, const float *const *srcs, int S, const float *pose,
                                   const float *hypo, float *vol, float *wsum, int B, int C, int G, int D, int H,
                                   int W, int Hs, int Ws, float attn_temp, int layout, int32_t *bin_counts,
                                   mvster_stream_t stream);

/* A2 / A3 for everything the specialised kernels above do not instantiate: any channel count C, group-wise correlation
 * with any G dividing C (group_cor != 0) or the per-channel squared difference (ref - warped)^2 of group_cor=False
 * (group_cor == 0: G is ignored, the volume has C channels), both view-weight modes (attn_fuse_d != 0: softmax over depth
 * / sqrt(C) per hypothesis; == 0: its maximum over depth per pixel, no temperature), any D <= MVSTER_MAX_DEPTHS.
 * Planar (NCHW) feature maps and gradients; gsrcs accumulate with fp32 atomics.  Written for coverage, not speed.
 * replaces: models/mvs4net_utils.py:1066-1100 in the modes no reference script uses, and odd shapes (C = 24, D = 6 ...).
 *   vol, gvol : [B, G or C, D, H, W]     wsum : [B,D,H,W] (the weight sum; per pixel in the attn_fuse_d == 0 mode, stored per d)
 */
int mvster_cost_volume_generic_fwd(const float *ref, const float *const *srcs, int S, const float *pose,
                                   const float *hypo, float *vol, float *wsum, int B, int C, int G, int D, int H, int W,
                                   int Hs, int Ws, float attn_temp, int group_cor, int attn_fuse_d, mvster_stream_t stream);
int mvster_cost_volume_generic_bwd(const float *gvol, const float *vol, const float *wsum, const float *ref,
                                   const float *const *srcs, int S, const float *pose, const float *hypo, float *gref,
                                   float *const *gsrcs, int B, int C, int G, int D, int H, int W, int Hs, int Ws,
                                   float attn_temp, int group_cor, int attn_fuse_d, mvster_stream_t stream);

/* A3  gradient of mvster_cost_volume_fwd w.r.t. every feature map (recompute-based).
 *   gvol : [B,G,D,H,W]   vol, wsum : the forward outputs
 *   gref : [B,C,H,W] (overwritten)   gsrcs : host array of S device pointers, each [B,C,Hs,Ws]
 *   (overwritten; fp32 atomics).  Feature maps AND their gradients use `layout`.
 */
int mvster_cost_volume_bwd(const float *gvol, const float *vol, const float *wsum, const float *ref,
                           const float *const *srcs, int S, const float *pose, const float *hypo,
                           float *gref, float *const *gsrcs, int B, int C, int G, int D, int H, int W,
                           int Hs, int Ws, float attn_temp, int layout, mvster_stream_t stream);

/* A3 with caller-provided scratch memory.  Same operation and results (to summation order).  With a workspace of at
 * least mvster_cost_volume_bwd_workspace(...) 4-byte words the channels-last backward takes the BINNED route, which
 * issues no global float atomic: a coefficient kernel (everything that needs the feature gathers: correlations,
 * softmax, reference gradient, and G coefficients per (view, pixel, hypothesis)), a counting sort of the
 * (pixel, hypothesis) samples by destination tile of the source map, and one workgroup per tile that sorts its samples
 * by source pixel in LDS, sums them in registers (one owner lane per pixel) and writes the tile once with plain stores.
 * The workspace holds the
 * coefficients [B,S,H*W,D,G] and the sort's arrays (counters, scan partials, offsets, cursors: one int32 per bin each, and
 * ONE 4-byte item per (sample, view): B*S*H*W*D words): see DESIGN.md section 3.
 * A workspace of only B*S*H*W*D*G words selects the round-1 TILE_SCATTER route for D = 8, C >= 16;
 * workspace == NULL or anything shorter: identical to mvster_cost_volume_bwd.  The library never allocates.
 * fwd_bin_counts: the counters mvster_cost_volume_fwd_counted filled for the same inputs, or NULL.  With them the binned
 * route skips its own count pass and the coefficient kernel also does the sort's fill pass (no extra projection passes).
 *   mvster_cost_volume_bwd_workspace : words the RECOMMENDED (fastest measured) route of this shape wants, 0 when it
 *       wants none.  The entry point takes the binned route whenever the workspace is large enough for it, the
 *       recommendation only decides what the query asks for (C = 8: nothing, the lane-split window kernel is faster).
 *   mvster_cost_volume_bwd_route_workspace : words a given MVSTER_BWD_ROUTE_* needs for this shape (0: none),
 *       -1 when that route does not exist for the shape / layout.
 */
long long mvster_cost_volume_bwd_workspace(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws,
                                           int layout);
long long mvster_cost_volume_bwd_route_workspace(int route, int S, int B, int C, int G, int D, int H, int W, int Hs,
                                                 int Ws, int layout);
int mvster_cost_volume_bwd_ws(const float *gvol, const float *vol, const float *wsum, const float *ref,
                              const float *const *srcs, int S, const float *pose, const float *hypo,
                              float *gref, float *const *gsrcs, int B, int C, int G, int D, int H, int W,
                              int Hs, int Ws, float attn_temp, int layout, float *workspace,
                              long long workspace_floats, const int32_t *fwd_bin_counts, mvster_stream_t stream);

/* Which kernel family mvster_cost_volume_bwd_ws runs for a shape and workspace size (so that tests and the
 * benchmark can state -- and assert -- the route instead of mirroring the dispatch rules).
 * mvster_cost_volume_bwd_route_name: the route's kernels as rocprofv3 prints them, dominant first, ';'-separated, e.g.
 * "tile_owner_c8_kernel<4, 4>;cost_volume_bwd_coef_kernel<8, 4, 4>"; writes at most `cap` bytes including the
 * terminator and returns the length it wanted to write (0 for MVSTER_BWD_ROUTE_NONE). */
#define MVSTER_BWD_ROUTE_NONE 0          /* shape not instantiated: the entry point returns MVSTER_ERR_UNSUPPORTED   */
#define MVSTER_BWD_ROUTE_PLANAR 1        /* NCHW planes, tap-by-tap atomics                                           */
#define MVSTER_BWD_ROUTE_DENSE 2         /* channels-last, one LDS-transposed dense atomic strip per hypothesis       */
#define MVSTER_BWD_ROUTE_SPLIT_WINDOW 3  /* channels-last, D = 4, C <= 16: per-pixel hypothesis window, 2/4 lanes per pixel */
#define MVSTER_BWD_ROUTE_TILE_SCATTER 4  /* channels-last, D = 8: coefficient kernel + tiled LDS write-combining scatter  */
#define MVSTER_BWD_ROUTE_BINNED 5        /* channels-last: coefficient kernel + counting sort by tile + owner lanes, no float atomics */
int mvster_cost_volume_bwd_route(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, int layout,
                                 long long workspace_floats);
int mvster_cost_volume_bwd_route_name(int route, int C, int G, int D, char *buf, int cap);

/* --------------------------------------------------------------------------------------------
 * A4  post-regnet half: softmax over depth, FIRST arg-max, depth gather, confidence, next range.
 * replaces: models/mvs4net_utils.py:1109-1113,1126-1138,1151-1156.
 *   logits, hypo : [B,D,H,W]
 *   attn   : [B,D,H,W]  softmax_d(logits)
 *   depth  : [B,H,W]    hypo gathered at the first arg-max of attn
 *   index  : [B,H,W]    int32 arg-max index (may be NULL)
 *   conf   : [B,H,W]    max_d(logits) / sum_d(logits) on the RAW logits (eval mode; NULL in training)
 *   inv_min, inv_max : [B,H,W]  1/depth +- split_itv * (1/hypo[:,2] - 1/hypo[:,1])
 *                      (both NULL when inverse_depth is off; needs D >= 3 otherwise)
 */
int mvster_depth_regress_fwd(const float *logits, const float *hypo, float *attn, float *depth,
                             int32_t *index, float *conf, float *inv_min, float *inv_max, int B, int D,
                             int H, int W, float split_itv, mvster_stream_t stream);

/* Softmax backward: glogits = attn * (gattn - sum_d(gattn * attn)).   all [B,D,H,W] */
int mvster_softmax_bwd(const float *attn, const float *gattn, float *glogits, int B, int D, int H, int W,
                       mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A5  first-stage hypotheses from depth_values[:,0] (min) and depth_values[:,K-1] (max).
 *   inverse != 0 : 1 / (1/dmax + (1/dmin - 1/dmax) * k/(D-1))   (plane 0 = far plane)
 *   inverse == 0 : dmin + k * (dmax - dmin)/(D-1)
 * replaces: models/mvs4net_utils.py:79-85 (init_inverse_range) and :69-77 (init_range).
 *   depth_values : [B,K]   out : [B,D,H,W]
 */
int mvster_init_range(const float *depth_values, int K, float *out, int B, int D, int H, int W, int inverse,
                      mvster_stream_t stream);

/* A6  next-stage hypotheses: linear ramp in inverse depth between inv_max and inv_min at the
 * previous resolution, x2 bilinear (align_corners=True) upsample, reciprocal.
 * replaces: models/mvs4net_utils.py:87-94 (schedule_inverse_range).
 *   inv_min, inv_max : [B,Hp,Wp]   out : [B,D,H,W]
 */
int mvster_schedule_inverse_range(const float *inv_min, const float *inv_max, float *out, int B, int D,
                                  int Hp, int Wp, int H, int W, mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F1  Sinkhorn optimal-transport loss between softmax-over-depth `attn` and the ground-truth depth bin,
 * one fused per-pixel kernel.  pix_loss[b,h,w] = sum_{r,c} T[r,c] * cost[r,c] after `iters` log-domain
 * sweeps (0 where mask == 0); the stage loss of the reference is mean(pix_loss[mask]).
 * replaces: models/mvs4net_utils.py:1164-1210 (sinkhorn; `continuous` selects its D x (D+1) variant).
 *   gt_depth : [B,H,W]   hypo, attn : [B,D,H,W]   mask : [B,H,W] uint8 (0 = ignore)
 *   pix_loss : [B,H,W]   tmap : [B,H*W,D,D(+1)] transport plan (may be NULL; NaN rows where mask == 0)
 * Supported: D in {4,8,16}, iters <= 16; otherwise MVSTER_ERR_UNSUPPORTED.
 */
int mvster_sinkhorn_fwd(const float *gt_depth, const float *hypo, const float *attn, const uint8_t *mask,
                        float *pix_loss, float *tmap, int B, int D, int H, int W, int iters, float eps,
                        int continuous, mvster_stream_t stream);

/* Gradient w.r.t. attn: gattn[b,d,h,w] = gpix[b,h,w] * d pix_loss / d attn (0 where mask == 0). */
int mvster_sinkhorn_bwd(const float *gt_depth, const float *hypo, const float *attn, const uint8_t *mask,
                        const float *gpix, float *gattn, int B, int D, int H, int W, int iters, float eps,
                        int continuous, mvster_stream_t stream);

/* The same loss for the shapes the register-resident kernels do not instantiate: any D <= 64 (stage_splits such as 5, 6, 12,
 * 24, 48), any number of sweeps -- plain loops, the pixel's vectors and (backward) its per-sweep history in a workspace
 * laid out [slot][pixel]; coverage, not speed.  D > 64 -> MVSTER_ERR_UNSUPPORTED.
 *   workspace: mvster_sinkhorn_generic_workspace(B, D, H, W, iters, continuous, backward) floats (-1: unsupported). */
long long mvster_sinkhorn_generic_workspace(int B, int D, int H, int W, int iters, int continuous, int backward);
int mvster_sinkhorn_generic_fwd(const float *gt_depth, const float *hypo, const float *attn, const uint8_t *mask,
                                float *pix_loss, float *tmap, float *workspace, long long workspace_floats, int B, int D,
                                int H, int W, int iters, float eps, int continuous, mvster_stream_t stream);
int mvster_sinkhorn_generic_bwd(const float *gt_depth, const float *hypo, const float *attn, const uint8_t *mask,
                                const float *gpix, float *gattn, float *workspace, long long workspace_floats, int B, int D,
                                int H, int W, int iters, float eps, int continuous, mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F2/F3  training-mode BatchNorm (+ optional fused ReLU) over [N,C,HW] planes (NCHW, or [B*D,C,H,W] depth slices).
 * replaces: the BatchNorm2d/3d + ReLU after every convolution of FPN4 / reg2d / the mono head
 * (models/mvs4net_utils.py:123-130,230-256) when the module is in training mode.
 *   forward : y = relu?(gamma * (x - mean) / sqrt(var + eps) + beta), batch statistics over N*HW per channel
 *             (biased variance), running_mean/var updated with `momentum` (unbiased variance), PyTorch semantics.
 *             save_mean, save_invstd : [C]; affine : [C,2] (a_c, beta_c with y = (x-mean_c)*a_c + beta_c); sums : [C,2] double scratch.
 *             gamma/beta/running_* may be NULL.
 *   backward: dx, dgamma, dbeta from dy and the SAME x, affine, save_mean, save_invstd (the ReLU mask is recomputed).
 */
int mvster_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y, float *save_mean,
                        float *save_invstd, float *running_mean, float *running_var, double *sums, float *affine,
                        int N, int C, int HW, float momentum, float eps, int relu, mvster_stream_t stream);
int mvster_bn_train_bwd(const float *dy, const float *x, const float *affine, const float *save_mean,
                        const float *save_invstd, float *dx, float *dgamma, float *dbeta, double *sums, int N, int C,
                        int HW, int relu, mvster_stream_t stream);

/* The same with `groups` statistic groups: the N samples are `groups` runs of N/groups consecutive samples, each run
 * normalised with its own batch statistics, gamma/beta shared -- one launch for what the reference does as one
 * FPN4 pass per view (models/MVS4Net.py:88-93: every view is a batch of its own in training mode).
 *   save_mean, save_invstd : [groups,C]; affine : [groups,C,2]; sums : [groups,C,2] double scratch.
 *   running_mean/var [C] receive the groups' updates one after the other, in group order (as the reference's successive
 *   calls do); dgamma/dbeta [C] are summed over the groups.  groups = 1 is mvster_bn_train_fwd/bwd.
 *   residual (may be NULL): y = relu?(bn(x)) + residual, same shape as x -- reg2d's skip connections
 *   (`x = conv2 + self.conv9(x)`, models/mvs4net_utils.py:921-923) without a separate add pass; its gradient is dy itself.
 *   N % groups != 0 -> MVSTER_ERR_SHAPE. */
int mvster_bn_train_fwd_grouped(const float *x, const float *gamma, const float *beta, const float *residual, float *y,
                                float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                                double *sums, float *affine, int N, int C, int HW, int groups, float momentum,
                                float eps, int relu, mvster_stream_t stream);
int mvster_bn_train_bwd_grouped(const float *dy, const float *x, const float *affine, const float *save_mean,
                                const float *save_invstd, float *dx, float *dgamma, float *dbeta, double *sums, int N,
                                int C, int HW, int groups, int relu, mvster_stream_t stream);

/* The same two operations on channels-last storage: x, y, dy, dx are [M = N*H*W, C] row-major (torch.channels_last),
 * C in {4, 8, 16, 32, 64} (else MVSTER_ERR_UNSUPPORTED).  All other arguments as above. */
int mvster_bn_train_fwd_nhwc(const float *x, const float *gamma, const float *beta, float *y, float *save_mean,
                             float *save_invstd, float *running_mean, float *running_var, double *sums,
                             float *affine, long long M, int C, float momentum, float eps, int relu,
                             mvster_stream_t stream);
int mvster_bn_train_bwd_nhwc(const float *dy, const float *x, const float *affine, const float *save_mean,
                             const float *save_invstd, float *dx, float *dgamma, float *dbeta, double *sums,
                             long long M, int C, int relu, mvster_stream_t stream);

/* Inference mode (model.eval(): the whole of test_mvs4.py:393-411): the statistics are the module's running ones,
 *   y = relu?((x - running_mean) / sqrt(running_var + eps) * gamma + beta) [+ residual],
 * ONE streaming pass instead of F.batch_norm(training=False) followed by F.relu (and the skip-connection add).
 *   scratch : 3*C floats (the per-channel affine and mean, prepared by a C-thread launch on the same stream).
 *   residual (may be NULL) as in mvster_bn_train_fwd_grouped.  _nhwc: channels-last storage, C in {4, 8, 16, 32, 64}.
 *   mvster_bn_eval_prepare : only the coefficient launch -- norm[mvster_bn_eval_norm_floats(C)] = three arrays a | b | mean,
 *       each padded with zeros to a multiple of 32 channels, a = gamma / sqrt(running_var + eps), b = beta: what the *_ep
 *       convolution entry points take to apply the layer's BatchNorm in their epilogue (valid until the module's
 *       parameters / buffers change). */
long long mvster_bn_eval_norm_floats(int C);
int mvster_bn_eval_fwd(const float *x, const float *gamma, const float *beta, const float *running_mean,
                       const float *running_var, const float *residual, float *y, float *scratch, int N, int C, int HW,
                       float eps, int relu, mvster_stream_t stream);
int mvster_bn_eval_prepare(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                           float *norm, int C, float eps, mvster_stream_t stream);
int mvster_bn_eval_fwd_nhwc(const float *x, const float *gamma, const float *beta, const float *running_mean,
                            const float *running_var, float *y, float *scratch, long long M, int C, float eps, int relu,
                            mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F3  top pyramid level without its 64-channel full-resolution map.
 * replaces: `intra_feat = F.interpolate(intra_feat, x2, bilinear, align_corners=True) + self.inner3(conv0)` followed by
 * `self.out4(intra_feat)` (models/mvs4net_utils.py:500-505) when nothing else reads intra_feat: the 3x3 layer's channel
 * contraction is done at LOW resolution first (G = a 1x1 layer with 8*9 output channels, channel index co*9 + ky*3 + kx),
 * and this operator gathers, per output pixel, the nine shifted bilinear samples:
 *   out[n,co,Y,X] = base[n,co,Y,X] + sum_{ky,kx : (Y+ky-1, X+kx-1) inside} bilinear_x2(g[n, co*9+ky*3+kx])(Y+ky-1, X+kx-1)
 *   g [N,72,h,w]   base, out, dy [N,8,2h,2w] (base = the 8 -> 8 layer on conv0 with the composed weights; out may alias base)
 *   _fwd_layout: out_layout = MVSTER_LAYOUT_NHWC writes out as [N,2h,2w,8] (then out must NOT alias base) -- the stage-4
 *         feature map in the layout the cost-volume kernels gather from; MVSTER_LAYOUT_NCHW = mvster_fpn_head_gather_fwd.
 *   _bwd: dg [N,72,h,w] from dy (d base = dy). */
int mvster_fpn_head_gather_fwd(const float *g, const float *base, float *out, int N, int h, int w, mvster_stream_t stream);
int mvster_fpn_head_gather_fwd_layout(const float *g, const float *base, float *out, int N, int h, int w, int out_layout,
                                      mvster_stream_t stream);
int mvster_fpn_head_gather_bwd(const float *dy, float *dg, int N, int h, int w, mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F2/F3  many-channel convolutions on the fp32 matrix cores (csrc/conv_mfma.hip), NCHW fp32, padding K/2:
 * (K, stride) in {(1,1), (3,1), (3,2), (5,2)}.  replaces: nn.Conv2d / the nn.Conv3d layers run as conv2d over depth
 * slices in FPN4 (models/mvs4net_utils.py:441-455,466-470) and reg2d (models/mvs4net_utils.py:891-899).
 *   x [N,Cin,H,W]   y, dy [N,Cout,Ho,Wo] with Ho = (H + 2*(K/2) - K) / stride + 1 (same for Wo)
 *   mvster_conv_mfma_fwd : y = conv(x, weight [Cout,Cin,K,K]) (+ bias [Cout], may be NULL).
 *       flags & MVSTER_CONV_FLAG_FLIP (stride 1 only): `weight` is [Cin,Cout,K,K] and its taps are mirrored -- with
 *       x := dy of a stride-1 layer and Cin/Cout swapped this is that layer's input gradient.
 *       flags & MVSTER_CONV_FLAG_OUT_NHWC (Cout % 4 == 0): y is written channels-last, [N,Ho,Wo,Cout] -- FPN4's out1 / out2
 *       hand their maps to the cost-volume kernels in the layout those gather from, no conversion pass in between.
 *   mvster_conv_mfma_wgrad : dweight [Cout,Cin,K,K] = sum over n, pixels of dy (x) x-window.  (For a transposed
 *       convolution layer pass x := the layer's dy, dy := the layer's x, Cin/Cout swapped: the result is its
 *       [Cin,Cout,K,K] weight gradient.)  No atomics: bitwise reproducible.
 *   workspace: mvster_conv_mfma_workspace(..., op) floats (packed weights / per-workgroup partial sums).
 *   Unsupported (K, stride) -> MVSTER_ERR_UNSUPPORTED; mvster_conv_mfma_supported() says so beforehand. */
#define MVSTER_CONV_OP_FWD 0
#define MVSTER_CONV_OP_WGRAD 1
#define MVSTER_CONV_OP_TRANSPOSED 2     /* mvster_conv_mfma_transposed (pass stride = 2): its packing differs from the forward's */
#define MVSTER_CONV_FLAG_FLIP 1
#define MVSTER_CONV_FLAG_OUT_NHWC 2
int mvster_conv_mfma_supported(int Cin, int Cout, int K, int stride);
long long mvster_conv_mfma_workspace(int N, int Cin, int Cout, int H, int W, int K, int stride, int op);
int mvster_conv_mfma_fwd(const float *x, const float *weight, const float *bias, float *y, float *workspace,
                         long long workspace_floats, int N, int Cin, int Cout, int H, int W, int K, int stride, int flags,
                         mvster_stream_t stream);
int mvster_conv_mfma_wgrad(const float *x, const float *dy, float *dweight, float *workspace, long long workspace_floats,
                           int N, int Cin, int Cout, int H, int W, int K, int stride, mvster_stream_t stream);
/* The (3,3,3) layers of reg2d (models/mvs4net_utils.py:892,895,898: Conv3d kernel 3, stride 1, padding 1) over depth slices
 * stacked into the batch, WITHOUT materialising the three depth-shifted copies of the input: sample n = b*depth + d; input
 * channel dz*C + c of sample n is channel c of sample n + dz - 1, zero where d + dz - 1 leaves [0, depth).
 *   x [N,C,H,W] (N % depth == 0), weight [Cout, 3*C, K, K] (the Conv3d weight [Cout,C,3,K,K] with the depth taps folded
 *   into the channel axis, depth tap major), y, dy [N,Cout,H,W], stride 1, padding K/2.
 *   _fwd, input_gradient = 0 : y = conv(x, weight) (+ bias)
 *   _fwd, input_gradient = 1 : pass x := dy, y := dx [N,C,H,W]: the layer's input gradient (bias ignored)
 *   _wgrad                   : dweight [Cout, 3*C, K, K]
 *   workspace: mvster_conv_mfma_workspace(N, 3*C, Cout, ...) (forward, weight gradient) / (N, 3*Cout, C, ...) (input gradient).
 *   _wgrad: K = 3 and more than 8 output channels only, else MVSTER_ERR_UNSUPPORTED. */
int mvster_conv_mfma_depth3_fwd(const float *x, const float *weight, const float *bias, float *y, float *workspace,
                                long long workspace_floats, int N, int C, int Cout, int H, int W, int K, int depth,
                                int input_gradient, mvster_stream_t stream);
int mvster_conv_mfma_depth3_wgrad(const float *x, const float *dy, float *dweight, float *workspace,
                                  long long workspace_floats, int N, int C, int Cout, int H, int W, int K, int depth,
                                  mvster_stream_t stream);
/* Transposed convolution, stride 2, K in {3, 5}, padding K/2, as a gather (no zero insertion):
 *   y[co][oy][ox] = sum_{ci, taps of matching parity} weight[ci][co][ky][kx] * x[ci][(oy + K/2 - ky)/2][(ox + K/2 - kx)/2]
 * replaces: reg2d's nn.ConvTranspose3d (1,3,3) up-convolutions (models/mvs4net_utils.py:901-912, as conv_transpose2d over
 * depth slices) and -- with x := dy, weight := the layer's [Cout,Cin,K,K] weights, Cin/Cout swapped -- the input gradient of
 * the stride-2 layers.  x [N,Cin,H,W], weight [Cin,Cout,K,K], y [N,Cout,Ho,Wo]; [Ho,Wo] must map back onto [H,W] under the
 * stride-2 convolution ((Ho + 2*(K/2) - K)/2 + 1 == H: Ho = 2H or 2H-1), else MVSTER_ERR_BAD_SHAPE.
 * workspace: mvster_conv_mfma_workspace(N, Cin, Cout, H, W, K, 2, MVSTER_CONV_OP_FWD) floats. */
int mvster_conv_mfma_transposed(const float *x, const float *weight, float *y, float *workspace, long long workspace_floats,
                                int N, int Cin, int Cout, int H, int W, int Ho, int Wo, int K, mvster_stream_t stream);

/* Inference (model.eval(), no autograd -- the whole of test_mvs4.py:393-411): convolution -> BatchNorm(running statistics)
 * -> ReLU (-> + skip connection) as ONE kernel, and the weights packed once instead of on every call.
 * replaces: ConvBnReLU / ConvBnReLU3D / the ConvTranspose3d + BatchNorm3d + ReLU blocks and `conv_k + self.conv_j(x)` of reg2d
 * in eval mode (models/mvs4net_utils.py:123-130,230-256,901-912,921-923).
 *   mvster_conv_mfma_packed_floats / mvster_conv_mfma_pack : the kernels' weight layout [CinPad][K*K][CoutPad], written once per
 *       parameter version.  op = MVSTER_CONV_OP_FWD (mode 0: weight [Cout,Cin,K,K]) or MVSTER_CONV_OP_TRANSPOSED (mode 2: weight
 *       [Cin,Cout,K,K]); for a depth-stacked (3,k,k) layer pass its folded [Cout,3*C,k,k] weight with Cin = 3*C.
 *   *_ep : `packed` from mvster_conv_mfma_pack (mvster_conv_direct_fwd_ep: wt as in mvster_conv_direct_fwd), no workspace;
 *       y = relu?(fmaf(conv + bias - mean[co], a[co], b[co])) + residual   with norm = mvster_bn_eval_prepare's array
 *       (NULL: no normalisation), residual (NULL: none) of y's shape and layout.  Bit-identical to the convolution entry
 *       point followed by mvster_bn_eval_fwd.  depth > 0 (mvster_conv_mfma_fwd_ep): the Cin = 3*C input channels are the
 *       depth-shifted slices of mvster_conv_mfma_depth3_fwd. */
long long mvster_conv_mfma_packed_floats(int Cin, int Cout, int K, int stride, int op);
int mvster_conv_mfma_pack(const float *weight, float *packed, long long packed_floats, int Cin, int Cout, int K, int stride,
                          int op, int mode, mvster_stream_t stream);
int mvster_conv_mfma_fwd_ep(const float *x, const float *packed, const float *bias, float *y, int N, int Cin, int Cout, int H,
                            int W, int K, int stride, int flags, int depth, const float *norm, const float *residual, int relu,
                            mvster_stream_t stream);
int mvster_conv_mfma_transposed_ep(const float *x, const float *packed, float *y, int N, int Cin, int Cout, int H, int W,
                                   int Ho, int Wo, int K, const float *norm, const float *residual, int relu,
                                   mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F3  top-down merge of the feature pyramid, fused:  y = bilinear_x2(feat, align_corners=True) + conv1x1(skip; weight, bias).
 * replaces: models/mvs4net_utils.py:490-505 (F.interpolate + innerK 1x1 convolution + add in FPN4.forward) and, in the
 * backward, ATen's atomic upsample backward, two convolution-gradient kernels with their transposes and a bias reduction.
 *   feat [N,Ct,h,w]   skip [N,Cs,2h,2w]   weight [Ct,Cs]   bias [Ct] (may be NULL)   y, dy [N,Ct,2h,2w]
 *   dfeat [N,Ct,h,w]  dskip [N,Cs,2h,2w]  dweight [Ct,Cs]  dbias [Ct] (may be NULL)   -- all overwritten
 * NCHW planes.  Ct <= 64, Cs in {8,16,32}; otherwise MVSTER_ERR_UNSUPPORTED.
 */
int mvster_fpn_merge_fwd(const float *feat, const float *skip, const float *weight, const float *bias, float *y, int N,
                         int Ct, int Cs, int h, int w, mvster_stream_t stream);
int mvster_fpn_merge_bwd(const float *dy, const float *skip, const float *weight, float *dfeat, float *dskip,
                         float *dweight, float *dbias, int N, int Ct, int Cs, int h, int w, mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F2 / F3  direct convolution for the small-channel, full-resolution layers (stride 1, "same" padding, K = 1 or 3; NCHW).
 * replaces: nn.Conv2d / (1,3,3) nn.Conv3d slices of FPN4 conv0 / out3 / out4, reg2d conv0 / prob, the mono head's c->1
 * layers (models/mvs4net_utils.py:436-440,452-455,860-870,890,914), which MIOpen runs 4-10x off their memory floor.
 *   mvster_conv_direct_fwd   : y[n,co] = bias[co] + sum_{ci,tap} wt[ci][tap][co] * x[n,ci, . + tap]
 *        x [N,CIN,H,W]   wt [CIN,K*K,COUT] (the weight [COUT,CIN,K,K] permuted; for the INPUT gradient pass dy as x and the
 *        flipped weight permuted to [COUT,K*K,CIN])   bias [COUT] or NULL   y [N,COUT,H,W].  COUT in {1,3,4,8,16,32,64}.
 *   mvster_conv_direct_fwd_layout : the same with y in out_layout; MVSTER_LAYOUT_NHWC (COUT % 4 == 0) writes [N,H,W,COUT]
 *        (FPN4's out3 feeding the cost-volume kernels), MVSTER_LAYOUT_NCHW = mvster_conv_direct_fwd.
 *   mvster_conv_direct_wgrad : dweight [COUT,CIN,K,K] (+ dbias [COUT] unless NULL) from x and dy; workspace of
 *        mvster_conv_direct_wgrad_workspace(...) floats holds per-tile partial sums (no atomics, reproducible).
 *        COUT*K <= 256 and COUT <= 32, else MVSTER_ERR_UNSUPPORTED.
 */
int mvster_conv_direct_supported(int CIN, int COUT, int K);
int mvster_conv_direct_fwd(const float *x, const float *wt, const float *bias, float *y, int N, int CIN, int COUT, int H,
                           int W, int K, mvster_stream_t stream);
int mvster_conv_direct_fwd_layout(const float *x, const float *wt, const float *bias, float *y, int N, int CIN, int COUT,
                                  int H, int W, int K, int out_layout, mvster_stream_t stream);
/* Inference: the layer's eval-mode BatchNorm + ReLU (+ skip connection) in the convolution's epilogue, see F2/F3 below. */
int mvster_conv_direct_fwd_ep(const float *x, const float *wt, const float *bias, float *y, int N, int CIN, int COUT, int H,
                              int W, int K, int out_layout, const float *norm, const float *residual, int relu,
                              mvster_stream_t stream);
long long mvster_conv_direct_wgrad_workspace(int N, int CIN, int COUT, int H, int W, int K);
int mvster_conv_direct_wgrad(const float *x, const float *dy, float *dweight, float *dbias, float *workspace,
                             long long workspace_floats, int N, int CIN, int COUT, int H, int W, int K,
                             mvster_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * F4  geometric consistency of a (reference, source) depth-map pair: project each reference pixel into the source
 * view with its depth, sample the source depth bilinearly (constant border 0), project back, keep the pixel when the
 * round trip lands within `pix_thresh` pixels and `depth_thresh` relative depth.  float64 per-pixel algebra like numpy.
 * replaces: test_mvs4.py:612-670 (reproject_with_depth + check_geometric_consistency).  PARITY UNPINNED (cv2.remap).
 *   depth_ref [H,W], depth_src [Hs,Ws] : device, fp32
 *   mats : HOST pointer to 68 doubles, row-major: inv(K_ref)[3x3], E_src@inv(E_ref)[4x4], K_src, inv(K_src),
 *          E_ref@inv(E_src)[4x4], K_ref
 *   mask [H,W] uint8, depth_reproj [H,W] (0 where rejected), x_src, y_src [H,W] (both may be NULL) : device
 */
int mvster_geometric_consistency(const float *depth_ref, const float *depth_src, const double *mats, uint8_t *mask,
                                 float *depth_reproj, float *x_src, float *y_src, int H, int W, int Hs, int Ws,
                                 float pix_thresh, float depth_thresh, mvster_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVSTER_HIP_H */
