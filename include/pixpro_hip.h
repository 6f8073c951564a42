/* pixpro_hip.h — C ABI of libpixpro_hip.so (MI355X / gfx950).
 *
 * The reference (rioyokotalab/PixPro-with-OpticalFlow) is pure Python/PyTorch and has no
 * native ABI of its own; every entry point below replaces the chain of stock torch kernels
 * that one reference *function* launches, and cites that function (file:line relative to
 * the reference root).  The Python host in pixpro-with-opticalflow_amd/contrast/ binds
 * these with ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous fp32 (or uint8 for masks) array,
 *     unless the parameter name ends in _host;
 *   - the caller owns all memory, including workspaces (query *_workspace_bytes first);
 *     the library never allocates, frees or synchronises;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - return value 0 = success; otherwise an error code, text via pixpro_last_error()
 *     (thread-local).  Functions are re-entrant.
 */
#ifndef PIXPRO_HIP_H
#define PIXPRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIXPRO_OK 0
#define PIXPRO_ERR_ARG 1      /* bad shape / null pointer / unsupported combination */
#define PIXPRO_ERR_WORKSPACE 2 /* workspace too small */
#define PIXPRO_ERR_LAUNCH 3   /* hipLaunch failure (text carries hipGetErrorString) */

#define PIXPRO_MAX_DIRS 2

/* Geometry + hyper-parameters of one PPM/loss problem.
 * N samples, C channels, H x W feature grid (P = H*W), flow map Hf x Wf (0 = no flow),
 * original frame Ho x Wo (used only when size_dev == NULL and Ho > 0; when both are
 * absent the frame size is read on the device from coord_q[0][9], coord_q[0][8] exactly as
 * contrast/models/PixPro.py:124 does on the host). */
typedef struct pixpro_problem {
  int32_t N, C, H, W;
  int32_t Hf, Wf;
  int32_t Ho, Wo;
  float p;          /* --pixpro-p            (PixPro.py:358) */
  float clamp;      /* --pixpro-clamp-value  (PixPro.py:355) */
  float pos_ratio;  /* --pixpro-pos-ratio    (PixPro.py:219) */
  int32_t normalize_k; /* 1: k is the raw projector_k output, L2-normalise it over C inside
                          (PixPro.py:403,407); 0: k is already unit-norm */
  int32_t ndir;     /* 1 or 2 loss directions handled by one call */
  float grad_scale; /* upstream d(loss): grads are those of grad_scale * loss (normally 1) */
  int32_t layout;   /* memory order of x, v, k, pred, dx, dv: 0 = [N,C,P] (NCHW-contiguous), 1 = [N,P,C] (NHWC, i.e.
                       torch channels_last).  1 is accepted by pixpro_ppm_loss on the single-launch path only. */
  /* Lazy flow mode (--flow_up): lo_h > 0 means the composed flow [N,2,8*lo_h,8*lo_w] and the consistency mask are
   * NOT passed as dense tensors (pixpro_dir.flow / .mask are NULL); they are evaluated inside the kernel, at the P
   * sample points only, from the loader's 1/8-resolution stacks pixpro_dir.lo_own / .lo_opp
   * ([N,lo_frames,2,lo_h,lo_w], lo_frames = nframe-1 flows chained as concat_flow does) with the arithmetic of
   * util.apply_optical_flow (util.py:174-248) - identical results, ~10^3 x fewer bytes than preparing the dense
   * tensors.  lazy_mask = 0 reproduces alpha1/alpha2 = None. */
  int32_t lo_h, lo_w;
  int32_t lazy_mask;
  float alpha1;
  int32_t lo_frames;  /* flows per chain (nframe-1); 0 is read as 1 */
  double alpha2;
} pixpro_problem;

/* Per-direction operands.  Direction d: q-side view (x,v), k-side momentum projection k.
 * mask: uint8 [N,Hf,Wf] (torch.bool storage) or NULL.  flow: [N,2,Hf,Wf] or NULL.
 * size_dev: int32[2] = {Ho, Wo} on the device or NULL.
 * Outputs: pred [N,C,P] = normalize(featprop(x)) (nullable); dx, dv [N,C,P] (nullable, both
 * or neither). */
typedef struct pixpro_dir {
  const float* x;
  const float* v;
  const float* k;
  const float* coord_q;   /* [N,10] */
  const float* coord_k;   /* [N,10] */
  const float* flow;
  const uint8_t* mask;
  const int32_t* size_dev;
  const float* lo_own;    /* lazy flow mode: low-res flows of this direction, [N,lo_frames,2,lo_h,lo_w] */
  const float* lo_opp;    /* lazy flow mode: low-res flows of the opposite direction (for the mask) */
  float* pred;
  float* dx;
  float* dv;
} pixpro_dir;

int pixpro_version(void);            /* 100 * major + minor: 110 = round 2 (flow_prepare_ws added), 120 = round 4 (BatchNorm
                                         ReLU bit record added); nothing removed */
const char* pixpro_last_error(void);

/* ---- K1+K2+K3+K4 fused: PixPro.featprop (PixPro.py:339-363) + F.normalize (:380) +
 * regression_loss (:92-247, incl. add_optical_flow :46-89) and their backward.
 * loss_n, pos_num, pos_mean: [ndir*N]; loss: [1] = sum over directions of -2*mean_n(loss_n)
 * (PixPro.py:247,431).  Gradients (if dx != NULL) are d(grad_scale*loss)/dx, /dv.
 * Uses the single-launch MFMA kernel when P<=64 and C%128==0 (C<=256); larger grids (the 14x14
 * grid of 448^2 crops) with C in {64,128,256} and P%4==0 take the tiled 5-launch MFMA path,
 * anything else the staged generic path (both need the workspace). */
size_t pixpro_ppm_loss_workspace_bytes(const pixpro_problem* pb);          /* 0 when the MFMA kernel applies */
/* 1 when pixpro_ppm_loss consumes tiles of pb->layout in place for this shape (layout 1 = channels_last: the single-launch
 * kernels and the tiled path do, the staged path does not); 0: hand it NCHW tiles.  The reference's featprop is
 * layout-agnostic (PixPro.py:339-363); this lets the host wrapper avoid layout copies where the kernels can. */
int pixpro_ppm_loss_layout_supported(const pixpro_problem* pb);
size_t pixpro_ppm_loss_generic_workspace_bytes(const pixpro_problem* pb);  /* staged path, any shape */
int pixpro_ppm_loss(const pixpro_problem* pb, const pixpro_dir* dirs, float* loss_n,
                    float* pos_num, float* pos_mean, float* loss, void* workspace,
                    size_t workspace_bytes, void* stream);
/* Same, forcing the staged generic path (used by tests as an on-device cross-check). */
int pixpro_ppm_loss_generic(const pixpro_problem* pb, const pixpro_dir* dirs, float* loss_n,
                            float* pos_num, float* pos_mean, float* loss, void* workspace,
                            size_t workspace_bytes, void* stream);

/* buf[i] *= *scale_dev for i < n; returns immediately on the device when *scale_dev == 1.
 * Lets autograd apply a non-unit upstream gradient to the saved dx/dv without a host sync. */
int pixpro_scale_by_device_scalar(float* buf, size_t n, const float* scale_dev, void* stream);

/* ---- PixPro.featprop alone (public method, PixPro.py:339-363): Y = v_hat * A^T.
 * x, v, Y, dY, dx, dv: [N,C,P].  pb->N,C,H,W,p,clamp are used. */
size_t pixpro_featprop_workspace_bytes(const pixpro_problem* pb);
int pixpro_featprop_fwd(const pixpro_problem* pb, const float* x, const float* v, float* Y,
                        void* workspace, size_t workspace_bytes, void* stream);
int pixpro_featprop_bwd(const pixpro_problem* pb, const float* x, const float* v,
                        const float* dY, float* dx, float* dv, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- regression_loss alone (module function, PixPro.py:92-247): q (differentiable) and k
 * are given as-is ([N,C,P]); d->x = q, d->k = k, d->dx = dq (nullable); d->v, d->pred,
 * d->dv ignored.  loss = -2*mean_n(loss_n). */
size_t pixpro_regression_loss_workspace_bytes(const pixpro_problem* pb);
int pixpro_regression_loss(const pixpro_problem* pb, const pixpro_dir* d, float* loss_n,
                           float* pos_num, float* pos_mean, float* loss, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- add_optical_flow (PixPro.py:46-89): x_grid, y_grid, out_x, out_y [N,H,W];
 * mask_grid uint8 [N,H,W] (nullable iff mask NULL).  Needs pb->N,H,W,Hf,Wf,Ho,Wo. */
int pixpro_add_optical_flow(const pixpro_problem* pb, const float* flow, const float* x_grid,
                            const float* y_grid, const uint8_t* mask, float* out_x,
                            float* out_y, uint8_t* mask_grid, void* stream);

/* ---- flow preparation (contrast/util.py) ------------------------------------------- */
/* upflow8 (contrast/flow/utils/utils.py:87-89): in [M,h,w] planes -> out [M,8h,8w],
 * 8 * bilinear(align_corners=True). */
int pixpro_upflow8(const float* in, float* out, int64_t planes, int32_t h, int32_t w,
                   void* stream);
/* concat_flow (util.py:300-330): flows [F1,N,2,Hf,Wf] -> out [N,2,Hf,Wf]; is_norm as in the
 * reference (--flow_cat_norm). */
int pixpro_concat_flow(const float* flows, float* out, int32_t F1, int32_t N, int32_t Hf,
                       int32_t Wf, int32_t is_norm, void* stream);
/* forward_backward_consistency (util.py:252-297): mask uint8 [N,Hf,Wf] (1 = valid);
 * cycle [N,2,Hf,Wf] nullable (debug output of the reference). */
int pixpro_fbc_mask(const float* flow_fwd, const float* flow_bwd, uint8_t* mask, float* cycle,
                    int32_t N, int32_t Hf, int32_t Wf, float alpha1, double alpha2,
                    int32_t is_norm, void* stream);
/* Fused file-mode apply_optical_flow (util.py:174-248, use_flow_frames=False,
 * flow_cat_norm=False): low-res flows lo_fwd/lo_bwd [N,F1,2,h,w] (loader layout, before the
 * permute of util.py:183) -> composed flows [N,2,Hf,Wf] and masks uint8 [N,Hf,Wf]
 * (Hf,Wf = 8h,8w if flow_up else h,w).  masks nullable (alpha None).  invalid_count
 * (nullable) int32 [2*N]: number of invalid pixels per sample (fwd then bwd), the numerator
 * of calc_mask_ratio (util.py:360-366).  Any alignment is accepted; the fast F1 == 1 route wants the flows 8-B and the
 * masks 2-B aligned (torch allocations are) and falls back to a one-thread-per-pixel kernel otherwise (same bits).
 * F1 == 1 (nframe 2): two launches (flows + certified masks; the tiles the
 * certificate cannot settle), the only HBM traffic is the low-res read and the output write; F1 > 1: one chain launch
 * (up-sampling evaluated lazily) + one mask launch (see pixpro_flow_prepare_ws for the certified mask route).  Both of ATen's CPU bilinear up-sampling orders are reproduced
 * (switch at 8h + 8w = 128).  No workspace.  The environment variable PIXPRO_FLOW_PATH = auto | certified | staged
 * selects between two bit-identical ways of producing the F1 == 1 masks (a testing / timing knob). */
int pixpro_flow_prepare(const float* lo_fwd, const float* lo_bwd, float* flow_fwd,
                        float* flow_bwd, uint8_t* mask_fwd, uint8_t* mask_bwd,
                        int32_t* invalid_count, int32_t N, int32_t F1, int32_t h, int32_t w,
                        int32_t flow_up, float alpha1, double alpha2, void* stream);
/* The same with a caller-owned scratch buffer (16-B aligned, pixpro_flow_prepare_workspace_bytes; 0 for F1 == 1 and for
 * the shapes the generic kernels take).  With it the masks of F1 > 1 (the production recipe is nframe = 6) take the
 * certified route of F1 == 1: the chain launch leaves the bounding box of every 8 x 8-pixel block of both composed
 * flows in the workspace, one launch settles every pixel those boxes can settle and a third one the tiles they cannot
 * (bit-identical to the route without workspace; replaces the same reference lines, util.py:174-248). */
size_t pixpro_flow_prepare_workspace_bytes(int32_t N, int32_t F1, int32_t h, int32_t w, int32_t flow_up);
int pixpro_flow_prepare_ws(const float* lo_fwd, const float* lo_bwd, float* flow_fwd,
                           float* flow_bwd, uint8_t* mask_fwd, uint8_t* mask_bwd,
                           int32_t* invalid_count, int32_t N, int32_t F1, int32_t h, int32_t w,
                           int32_t flow_up, float alpha1, double alpha2, void* workspace,
                           size_t workspace_bytes, void* stream);

/* calc_mask_ratio (util.py:360-366): ratio[n] = mean(!mask[n]). */
int pixpro_mask_ratio(const uint8_t* mask, float* ratio, int32_t N, int64_t pixels,
                      void* stream);

/* ---- multi-tensor updates around the path ("next" rows of SURVEY.md §8f) ------------- */
/* Momentum (EMA) update, PixPro.py:321-337: pk = pk*m + pq*(1-m) for `count` tensors.
 * pq_ptrs / pk_ptrs / sizes: DEVICE arrays of count pointers / element counts; m is the
 * python double of PixPro.py:326 (the kernel uses fl32(m) and fl32(1-m) like the reference).
 * count <= 2048 per call (PIXPRO_ERR_ARG beyond: split the list); tensors need not be 16-B aligned (a scalar path takes
 * those that are not).  The same limits hold for pixpro_lars_sgd_step. */
int pixpro_ema_update(const float* const* pq_ptrs, float* const* pk_ptrs,
                      const int64_t* sizes, int32_t count, double m, void* stream);
/* LARS + SGD-momentum step without host syncs (contrast/lars.py:109-151 followed by
 * torch.optim.SGD with momentum, dampening 0, no nesterov).  Per tensor t:
 *   g = grad + wd[t]*p;  if adapt[t] and |p|>0 and |g|>0: g *= trust*|p|/(|g|+eps)
 *   buf = first_step ? g : momentum*buf + g;  p -= lr*buf.
 * norms2: DEVICE scratch float[2*count] (zeroed by the call). */
int pixpro_lars_sgd_step(float* const* p_ptrs, const float* const* g_ptrs,
                         float* const* buf_ptrs, const int64_t* sizes, const float* wd,
                         const int32_t* adapt, int32_t count, float lr, float momentum,
                         float trust_coef, float eps, int32_t first_step, float* norms2,
                         void* stream);

/* ---- RAFT correlation lookup (reference contrast/flow/corr.py:29-50 CorrBlock.__call__) -------
 * level_l: [B*h*w, h2>>l, w2>>l] correlation maps (level 0 = all-pairs volume, then 2x2 average pools); coords
 * [B,2,h,w] = current match position (x, y) of every frame-1 pixel in level-0 pixels.
 * out [B, num_levels*(2r+1)^2, h, w]; channel (l, i, j) = bilinear sample (zeros padding, align_corners) of the
 * pixel's level-l map at (x/2^l + i - r, y/2^l + j - r).  One launch for all levels and taps. */
int pixpro_corr_lookup(const float* level0, const float* level1, const float* level2,
                       const float* level3, int32_t num_levels, const float* coords, float* out,
                       int32_t B, int32_t h, int32_t w, int32_t h2, int32_t w2, int32_t radius,
                       void* stream);

/* ---- (Sync)BatchNorm with ReLU / residual add folded in ----------
 * Replaces what torch.nn.SyncBatchNorm (installed by contrast/models/PixPro.py:289-292) runs per layer in training mode
 * (torch/nn/modules/_functions.py: batch_norm_stats, batch_norm_gather_stats_with_counts, batch_norm_elemt,
 * batch_norm_backward_reduce, batch_norm_backward_elemt) plus the F.relu / `out += identity` that follow it in the
 * ResNet blocks (contrast/resnet.py).  Activations are NHWC fp32 seen as [rows = N*H*W][C], C % 4 == 0; the collectives
 * between the calls stay with the caller (torch.distributed over RCCL):
 *   forward : pixpro_bn_stats -> all_gather(packed [2C+1]) -> pixpro_bn_apply
 *   backward: pixpro_bn_backward_reduce -> all_reduce(sums [2C]) -> pixpro_bn_backward_apply                         */
/* Workspace of pixpro_bn_stats / pixpro_bn_backward_reduce.  Its first 4096 bytes are int32 tickets (the work-group that
 * finishes last folds the partial sums): they must be ZERO when a call starts and are zero again when its kernel has
 * finished, so one zero-initialised buffer serves every later call that is ordered after it (same stream).  Two calls
 * that may run concurrently (different streams) need a workspace each.  0 = unsupported shape. */
size_t pixpro_bn_workspace_bytes(int64_t rows, int32_t C);
/* packed[0..C) = mean, packed[C..2C) = biased variance, packed[2C] = rows: this rank's statistics. */
int pixpro_bn_stats(const float* x, int64_t rows, int32_t C, void* workspace, size_t workspace_bytes,
                    float* packed, void* stream);
/* gathered [world][2C+1]: the packed statistics of all ranks.  Writes mean, invstd [C] (saved for backward), updates
 * running_mean / running_var (nullable) with `momentum` (unbiased variance), and
 * y = relu?( (x - mean) * invstd * weight + bias (+ residual) ).  weight/bias/residual nullable.
 * scale_shift: scratch float[2C]. */
int pixpro_bn_apply(const float* gathered, int32_t world, const float* x, const float* residual,
                    const float* weight, const float* bias, float* running_mean, float* running_var,
                    float momentum, float eps, int32_t relu, int64_t rows, int32_t C, float* y,
                    float* mean, float* invstd, float* scale_shift, void* stream);
/* dz = dy * (y > 0) if relu else dy.  sums[0..C) = sum dz, sums[C..2C) = sum dz*(x-mean) of this rank;
 * grad_weight = sums[C..2C) * invstd, grad_bias = sums[0..C) (nullable).
 * The ReLU decision is read from the saved output y, or - when the forward had no residual - recomputed from x with the
 * forward's scale_shift (pass it and y may be NULL: one tensor read less). */
int pixpro_bn_backward_reduce(const float* dy, const float* x, const float* y, const float* scale_shift,
                              const float* mean, const float* invstd, int32_t relu, int64_t rows, int32_t C,
                              void* workspace, size_t workspace_bytes, float* sums,
                              float* grad_weight, float* grad_bias, void* stream);
/* sums: totals over all ranks.  counts[r * count_stride], r < world: the rows of every rank as exchanged in the forward
 * pass (pass gathered + 2C with count_stride = 2C+1); count = their sum, ranks may hold different batch sizes.
 * dx = (dz - sum_dz/count - (x-mean)*invstd^2*sum_dz_xmu/count) * invstd * weight;  dresidual = dz (nullable). */
int pixpro_bn_backward_apply(const float* dy, const float* x, const float* y, const float* scale_shift,
                             const float* mean, const float* invstd, const float* weight, const float* sums,
                             const float* counts, int32_t world, int64_t count_stride, int32_t relu, int64_t rows,
                             int32_t C, float* dx, float* dresidual, void* stream);

/* ---- ReLU decisions as a bit record (1.2) ----
 * Behind a residual add (the last BatchNorm of a ResNet unit, contrast/resnet.py: `out += identity; out = relu(out)`) the
 * ReLU decision cannot be recomputed from x, and reading the saved output y for it costs the backward pass two more
 * sweeps over the widest tensors of the network.  pixpro_bn_apply_bits is pixpro_bn_apply with relu = 1 that also
 * writes the decisions, one bit per element (rows * C / 8 bytes, caller-owned, 16-byte aligned); the two _bits backward
 * entry points read that record where the plain ones read y.  Same arithmetic, same results bit for bit.
 * pixpro_bn_relu_bits_bytes: size of the record, 0 = this shape has none (C % 256 != 0): use the plain entry points. */
size_t pixpro_bn_relu_bits_bytes(int64_t rows, int32_t C);
int pixpro_bn_apply_bits(const float* gathered, int32_t world, const float* x, const float* residual,
                         const float* weight, const float* bias, float* running_mean, float* running_var,
                         float momentum, float eps, int64_t rows, int32_t C, float* y, float* mean,
                         float* invstd, float* scale_shift, void* relu_bits, void* stream);
int pixpro_bn_backward_reduce_bits(const float* dy, const float* x, const void* relu_bits, const float* mean,
                                   const float* invstd, int64_t rows, int32_t C, void* workspace,
                                   size_t workspace_bytes, float* sums, float* grad_weight, float* grad_bias,
                                   void* stream);
int pixpro_bn_backward_apply_bits(const float* dy, const float* x, const void* relu_bits, const float* mean,
                                  const float* invstd, const float* weight, const float* sums,
                                  const float* counts, int32_t world, int64_t count_stride, int64_t rows,
                                  int32_t C, float* dx, float* dresidual, void* stream);
/* out = a + dy * bit (a nullable).  The shortcut gradient of a residual unit, dresidual = dy masked by the ReLU decisions,
 * need not be materialised by pixpro_bn_backward_apply_bits (pass dresidual = NULL): where autograd would add it to the
 * convolution branch's gradient `a` (contrast/resnet.py: `out += identity`), this reads dy and the record instead; a
 * BatchNorm on the shortcut (downsample.1) takes (dy, relu_bits) straight into its own _bits backward calls. */
int pixpro_bn_masked_add(const float* a, const float* dy, const void* relu_bits, float* out, int64_t rows,
                         int32_t C, void* stream);

/* ---- the stem: BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max pooling in one pass (1.2) ----
 * contrast/resnet.py: `x = maxpool(relu(bn1(conv1(x))))`.  The normalised full-resolution tensor is never written: the
 * forward reads x [N,H,W,C] (after pixpro_bn_stats + the collective, as for pixpro_bn_apply) and writes the pooled
 * tensor [N,(H-1)/2+1,(W-1)/2+1,C]; the backward calls take the POOLED gradient, rebuild which position won each window
 * (first maximum in row-major window order, as ATen's indices; windows summed in ascending order, as ATen's backward)
 * and continue as pixpro_bn_backward_reduce / pixpro_bn_backward_apply do.  C % 32 == 0.
 * Workspace as for pixpro_bn_backward_reduce with rows = N*H*W. */
int pixpro_bn_pool_supported(int32_t N, int32_t H, int32_t W, int32_t C);
int pixpro_bn_apply_pool(const float* gathered, int32_t world, const float* x, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float momentum, float eps, int32_t N, int32_t H,
                         int32_t W, int32_t C, float* pooled, float* mean, float* invstd, float* scale_shift,
                         void* stream);
int pixpro_bn_pool_backward_reduce(const float* dpooled, const float* x, const float* scale_shift, const float* mean,
                                   const float* invstd, int32_t N, int32_t H, int32_t W, int32_t C, void* workspace,
                                   size_t workspace_bytes, float* sums, float* grad_weight, float* grad_bias,
                                   void* stream);
int pixpro_bn_pool_backward_apply(const float* dpooled, const float* x, const float* scale_shift, const float* mean,
                                  const float* invstd, const float* weight, const float* sums, const float* counts,
                                  int32_t world, int64_t count_stride, int32_t N, int32_t H, int32_t W, int32_t C,
                                  float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXPRO_HIP_H */
