// extern "C" boundary of libpixpro_hip.so: argument validation, path selection, launches.
// No torch types, no allocation, no synchronisation (include/pixpro_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace pixpro {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PIXPRO_ERR_LAUNCH;
  }
  return PIXPRO_OK;
}

// implemented in the kernel translation units
bool fused_supported(const pixpro_problem& pb);
int run_fused(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
              hipStream_t st);
bool fused_nhwc_supported(const pixpro_problem& pb);
int run_fused_nhwc(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                   float* loss, int* folded, hipStream_t st);
size_t generic_workspace_bytes(const pixpro_problem& pb);
bool tiled_supported(const pixpro_problem& pb);
size_t tiled_workspace_bytes(const pixpro_problem& pb);
int run_tiled(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean, float* loss,
              void* workspace, size_t ws_bytes, hipStream_t st);
int run_generic(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                float* loss, void* workspace, size_t ws_bytes, hipStream_t st, int mode);
int run_finalize(const float* loss_n, float* loss, int B, int N, hipStream_t st);
int run_featprop(const pixpro_problem& pb, const float* x, const float* v, const float* dY, float* Y, float* dx,
                 float* dv, void* workspace, size_t ws_bytes, hipStream_t st);
int run_upflow8(const float* in, float* out, int64_t planes, int h, int w, hipStream_t st);
int run_concat_flow(const float* flows, float* out, int F1, int N, int Hf, int Wf, int is_norm, hipStream_t st);
int run_fbc(const float* f, const float* b, uint8_t* mask, float* cycle, int N, int Hf, int Wf, float alpha1,
            double alpha2, int is_norm, hipStream_t st);
int run_flow_prepare(const float* lo_fwd, const float* lo_bwd, float* flow_fwd, float* flow_bwd, uint8_t* mask_fwd,
                     uint8_t* mask_bwd, int32_t* invalid, int N, int F1, int h, int w, int flow_up, float alpha1,
                     double alpha2, void* workspace, size_t workspace_bytes, hipStream_t st);
size_t flow_prepare_workspace_bytes(int N, int F1, int h, int w, int flow_up);
int run_mask_ratio(const uint8_t* mask, float* ratio, int N, int64_t pixels, hipStream_t st);
int run_add_optical_flow(const pixpro_problem& pb, const float* flow, const float* xg, const float* yg,
                         const uint8_t* mask, float* ox, float* oy, uint8_t* mg, hipStream_t st);
int run_ema(const float* const* pq, float* const* pk, const int64_t* sizes, int count, double m, hipStream_t st);
int run_lars(float* const* pp, const float* const* gp, float* const* bp, const int64_t* sizes, const float* wd,
             const int32_t* adapt, int count, float lr, float momentum, float trust, float eps, int first_step,
             float* norms2, hipStream_t st);
int run_scale(float* buf, size_t n, const float* scale, hipStream_t st);
int run_corr_lookup(const float* const* levels, int L, const float* coords, float* out, int B, int h, int w, int h2, int w2,
                    int radius, hipStream_t st);
size_t bn_workspace_floats(int64_t M, int C);
size_t bn_relu_bits_bytes(int64_t M, int C);
int run_bn_masked_add(const float* a, const float* dy, const void* relu_bits, float* out, int64_t M, int C, hipStream_t st);
int bn_pool_supported(int N, int H, int W, int C);
int run_bn_apply_pool(const float* gathered, int Wr, const float* x, const float* weight, const float* bias, float* running_mean,
                      float* running_var, float momentum, float eps, int N, int H, int W, int C, float* pooled, float* mean,
                      float* invstd, float* scale_shift, hipStream_t st);
int run_bn_pool_bwd_reduce(const float* dp, const float* x, const float* scale_shift, const float* mean, const float* invstd,
                           int N, int H, int W, int C, float* ws, float* sums, float* grad_weight, float* grad_bias,
                           hipStream_t st);
int run_bn_pool_bwd_apply(const float* dp, const float* x, const float* scale_shift, const float* mean, const float* invstd,
                          const float* weight, const float* sums, const float* counts, int world, int64_t count_stride, int N,
                          int H, int W, int C, float* dx, hipStream_t st);
int run_bn_stats(const float* x, int64_t M, int C, float* ws, float* packed, hipStream_t st);
int run_bn_apply(const float* gathered, int W, const float* x, const float* res, const float* weight, const float* bias,
                 float* running_mean, float* running_var, float momentum, float eps, int relu, int64_t M, int C,
                 float* y, float* mean, float* invstd, float* scale_shift, void* relu_bits, hipStream_t st);
int run_bn_bwd_reduce(const float* dy, const float* x, const float* y, const float* scale_shift, const void* relu_bits,
                      const float* mean, const float* invstd, int relu, int64_t M, int C, float* ws, float* sums,
                      float* grad_weight, float* grad_bias, hipStream_t st);
int run_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* scale_shift, const void* relu_bits,
                     const float* mean, const float* invstd, const float* weight, const float* sums, const float* counts,
                     int world, int64_t count_stride, int relu, int64_t M, int C, float* dx, float* dres, hipStream_t st);

namespace {

int bad(const char* msg) {
  set_error("%s", msg);
  return PIXPRO_ERR_ARG;
}

int validate_problem(const pixpro_problem* pb, const pixpro_dir* dirs, bool need_v) {
  if (!pb || !dirs) return bad("null problem / dirs");
  if (pb->N < 1 || pb->C < 1 || pb->H < 1 || pb->W < 1) return bad("N, C, H, W must be positive");
  if (pb->ndir < 1 || pb->ndir > PIXPRO_MAX_DIRS) return bad("ndir must be 1 or 2");
  if (pb->layout != 0 && pb->layout != 1) return bad("layout must be 0 (NCHW) or 1 (NHWC)");
  for (int d = 0; d < pb->ndir; ++d) {
    const pixpro_dir& q = dirs[d];
    if (!q.x || !q.k || !q.coord_q || !q.coord_k) return bad("x, k, coord_q, coord_k are required");
    if (need_v && !q.v) return bad("v is required");
    if (need_v && ((q.dx == nullptr) != (q.dv == nullptr))) return bad("dx and dv must be given together");
    if ((q.dx == nullptr) != (dirs[0].dx == nullptr)) return bad("all directions must agree on wanting gradients");
    if (q.flow && (pb->Hf < 1 || pb->Wf < 1)) return bad("flow given but Hf/Wf not set");
    if (q.mask && !q.flow) return bad("mask given without flow");
    if (q.flow && !q.size_dev && pb->Ho < 1) return bad("flow given without a frame size");
    if ((q.lo_own == nullptr) != (q.lo_opp == nullptr)) return bad("lo_own and lo_opp go together");
    if (q.lo_own) {
      if (pb->lo_h < 2 || pb->lo_w < 2) return bad("lazy flow mode needs lo_h, lo_w >= 2");
      if (pb->lo_frames < 0 || pb->lo_frames > 64) return bad("lazy flow mode: lo_frames must be in 0..64");
      if (q.flow || q.mask) return bad("lazy flow mode excludes dense flow / mask tensors");
      if (!q.size_dev && pb->Ho < 1) return bad("lazy flow mode needs a frame size");
    }
  }
  return PIXPRO_OK;
}

}  // namespace
}  // namespace pixpro

using namespace pixpro;

extern "C" {

int pixpro_version(void) { return 120; }   // 1.1: pixpro_flow_prepare_ws, pixpro_flow_prepare_workspace_bytes; 1.2: the BatchNorm ReLU bit record

const char* pixpro_last_error(void) { return g_err; }

size_t pixpro_ppm_loss_workspace_bytes(const pixpro_problem* pb) {
  if (!pb || fused_supported(*pb)) return 0;
  const size_t g = generic_workspace_bytes(*pb), t = tiled_supported(*pb) ? tiled_workspace_bytes(*pb) : 0;
  return g > t ? g : t;
}

int pixpro_ppm_loss_layout_supported(const pixpro_problem* pb) {
  if (!pb) return 0;
  if (pb->layout == 0) return 1;
  return pb->layout == 1 && (fused_supported(*pb) || (tiled_supported(*pb) && !getenv("PIXPRO_NO_TILED"))) ? 1 : 0;
}

size_t pixpro_ppm_loss_generic_workspace_bytes(const pixpro_problem* pb) {
  if (!pb) return 0;
  return generic_workspace_bytes(*pb);
}

int pixpro_ppm_loss_generic(const pixpro_problem* pb, const pixpro_dir* dirs, float* loss_n, float* pos_num,
                            float* pos_mean, float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate_problem(pb, dirs, true)) return e;
  if (pb->layout != 0) return bad("the staged path takes NCHW tiles");
  if (!loss_n || !pos_num || !pos_mean || !loss || !workspace) return bad("null output / workspace");
  return run_generic(*pb, dirs, loss_n, pos_num, pos_mean, loss, workspace, workspace_bytes, (hipStream_t)stream, 0);
}

int pixpro_ppm_loss(const pixpro_problem* pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                    float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = validate_problem(pb, dirs, true)) return e;
  if (!loss_n || !pos_num || !pos_mean || !loss) return bad("null output");
  bool aligned = true;
  for (int d = 0; d < pb->ndir; ++d) {
    const pixpro_dir& q = dirs[d];
    const uintptr_t bits = (uintptr_t)q.x | (uintptr_t)q.v | (uintptr_t)q.k | (uintptr_t)q.dx | (uintptr_t)q.dv |
                           (uintptr_t)q.pred;      // 16 B per lane loads / stores / LDS-DMA in the MFMA paths
    if (bits & 15) aligned = false;
  }
  if (pb->layout == 1 && !((fused_supported(*pb) || tiled_supported(*pb)) && aligned))
    return bad("NHWC tiles are handled by the single-launch kernel (P <= 49, C in {128, 256}) and the tiled path "
               "(C in {64, 128, 256}, P % 4 == 0) only, 16-B aligned: see pixpro_ppm_loss_layout_supported");
  if (fused_supported(*pb) && aligned) {
    // channels_last tiles of the 7 x 7 grid at C = 256 (what the pre-training step runs): the NHWC-native persistent kernel
    // (that kernel also folds the per-sample losses into `loss`: one launch for the whole op)
    int folded = 0;
    if (int e = fused_nhwc_supported(*pb)
                    ? run_fused_nhwc(*pb, dirs, loss_n, pos_num, pos_mean, loss, &folded, (hipStream_t)stream)
                    : run_fused(*pb, dirs, loss_n, pos_num, pos_mean, (hipStream_t)stream))
      return e;
    if (folded) return PIXPRO_OK;
    return run_finalize(loss_n, loss, pb->N * pb->ndir, pb->N, (hipStream_t)stream);
  }
  if (!workspace) return bad("this shape takes a multi-launch path and needs a workspace");
  // grids beyond 64 positions (14 x 14 at 448^2 crops): the tiled MFMA path, 5 launches; anything else: the staged path
  if (tiled_supported(*pb) && aligned && !getenv("PIXPRO_NO_TILED"))
    return run_tiled(*pb, dirs, loss_n, pos_num, pos_mean, loss, workspace, workspace_bytes, (hipStream_t)stream);
  return run_generic(*pb, dirs, loss_n, pos_num, pos_mean, loss, workspace, workspace_bytes, (hipStream_t)stream, 0);
}

int pixpro_scale_by_device_scalar(float* buf, size_t n, const float* scale_dev, void* stream) {
  if (!buf || !scale_dev) return bad("null pointer");
  return run_scale(buf, n, scale_dev, (hipStream_t)stream);
}

size_t pixpro_featprop_workspace_bytes(const pixpro_problem* pb) {
  if (!pb) return 0;
  pixpro_problem q = *pb;
  q.ndir = 1;
  return generic_workspace_bytes(q);
}

int pixpro_featprop_fwd(const pixpro_problem* pb, const float* x, const float* v, float* Y, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (!pb || !x || !v || !Y || !workspace) return bad("null pointer");
  return run_featprop(*pb, x, v, nullptr, Y, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

int pixpro_featprop_bwd(const pixpro_problem* pb, const float* x, const float* v, const float* dY, float* dx,
                        float* dv, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pb || !x || !v || !dY || !dx || !dv || !workspace) return bad("null pointer");
  return run_featprop(*pb, x, v, dY, nullptr, dx, dv, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t pixpro_regression_loss_workspace_bytes(const pixpro_problem* pb) {
  return pixpro_featprop_workspace_bytes(pb);
}

int pixpro_regression_loss(const pixpro_problem* pb, const pixpro_dir* d, float* loss_n, float* pos_num,
                           float* pos_mean, float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  if (!pb || !d) return bad("null problem / dir");
  pixpro_problem q = *pb;
  q.ndir = 1;
  if (int e = validate_problem(&q, d, false)) return e;
  if (!loss_n || !pos_num || !pos_mean || !loss || !workspace) return bad("null output / workspace");
  return run_generic(q, d, loss_n, pos_num, pos_mean, loss, workspace, workspace_bytes, (hipStream_t)stream, 1);
}

int pixpro_add_optical_flow(const pixpro_problem* pb, const float* flow, const float* x_grid, const float* y_grid,
                            const uint8_t* mask, float* out_x, float* out_y, uint8_t* mask_grid, void* stream) {
  if (!pb || !flow || !x_grid || !y_grid || !out_x || !out_y) return bad("null pointer");
  if (pb->Hf < 1 || pb->Wf < 1 || pb->Ho < 2 || pb->Wo < 2) return bad("Hf, Wf, Ho, Wo must be set");
  if ((mask == nullptr) != (mask_grid == nullptr)) return bad("mask and mask_grid go together");
  return run_add_optical_flow(*pb, flow, x_grid, y_grid, mask, out_x, out_y, mask_grid, (hipStream_t)stream);
}

int pixpro_upflow8(const float* in, float* out, int64_t planes, int32_t h, int32_t w, void* stream) {
  if (!in || !out) return bad("null pointer");
  if (planes < 1 || h < 1 || w < 1) return bad("planes, h, w must be positive");
  return run_upflow8(in, out, planes, h, w, (hipStream_t)stream);
}

int pixpro_concat_flow(const float* flows, float* out, int32_t F1, int32_t N, int32_t Hf, int32_t Wf,
                       int32_t is_norm, void* stream) {
  if (!flows || !out) return bad("null pointer");
  if (F1 < 1 || N < 1 || Hf < 2 || Wf < 2) return bad("F1, N >= 1 and Hf, Wf >= 2 required");
  return run_concat_flow(flows, out, F1, N, Hf, Wf, is_norm, (hipStream_t)stream);
}

int pixpro_fbc_mask(const float* flow_fwd, const float* flow_bwd, uint8_t* mask, float* cycle, int32_t N, int32_t Hf,
                    int32_t Wf, float alpha1, double alpha2, int32_t is_norm, void* stream) {
  if (!flow_fwd || !flow_bwd || !mask) return bad("null pointer");
  if (N < 1 || Hf < 2 || Wf < 2) return bad("N >= 1 and Hf, Wf >= 2 required");
  return run_fbc(flow_fwd, flow_bwd, mask, cycle, N, Hf, Wf, alpha1, alpha2, is_norm, (hipStream_t)stream);
}

int pixpro_flow_prepare(const float* lo_fwd, const float* lo_bwd, float* flow_fwd, float* flow_bwd, uint8_t* mask_fwd,
                        uint8_t* mask_bwd, int32_t* invalid_count, int32_t N, int32_t F1, int32_t h, int32_t w,
                        int32_t flow_up, float alpha1, double alpha2, void* stream) {
  if (!lo_fwd || !lo_bwd || !flow_fwd || !flow_bwd) return bad("null pointer");
  if ((mask_fwd == nullptr) != (mask_bwd == nullptr)) return bad("mask_fwd and mask_bwd go together");
  if (N < 1 || F1 < 1 || h < 2 || w < 2) return bad("N, F1 >= 1 and h, w >= 2 required");
  return run_flow_prepare(lo_fwd, lo_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, invalid_count, N, F1, h, w,
                          flow_up, alpha1, alpha2, nullptr, 0, (hipStream_t)stream);
}

size_t pixpro_flow_prepare_workspace_bytes(int32_t N, int32_t F1, int32_t h, int32_t w, int32_t flow_up) {
  if (N < 1 || F1 < 1 || h < 2 || w < 2) return 0;
  return flow_prepare_workspace_bytes(N, F1, h, w, flow_up);
}

int pixpro_flow_prepare_ws(const float* lo_fwd, const float* lo_bwd, float* flow_fwd, float* flow_bwd,
                           uint8_t* mask_fwd, uint8_t* mask_bwd, int32_t* invalid_count, int32_t N, int32_t F1,
                           int32_t h, int32_t w, int32_t flow_up, float alpha1, double alpha2, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (!lo_fwd || !lo_bwd || !flow_fwd || !flow_bwd) return bad("null pointer");
  if ((mask_fwd == nullptr) != (mask_bwd == nullptr)) return bad("mask_fwd and mask_bwd go together");
  if (N < 1 || F1 < 1 || h < 2 || w < 2) return bad("N, F1 >= 1 and h, w >= 2 required");
  if (workspace && workspace_bytes < flow_prepare_workspace_bytes(N, F1, h, w, flow_up))
    return bad("workspace smaller than pixpro_flow_prepare_workspace_bytes");
  if (workspace && ((uintptr_t)workspace & 15) != 0) return bad("workspace must be 16-byte aligned");
  return run_flow_prepare(lo_fwd, lo_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, invalid_count, N, F1, h, w,
                          flow_up, alpha1, alpha2, workspace, workspace_bytes, (hipStream_t)stream);
}

int pixpro_mask_ratio(const uint8_t* mask, float* ratio, int32_t N, int64_t pixels, void* stream) {
  if (!mask || !ratio) return bad("null pointer");
  if (N < 1 || pixels < 1) return bad("N, pixels must be positive");
  return run_mask_ratio(mask, ratio, N, pixels, (hipStream_t)stream);
}

int pixpro_ema_update(const float* const* pq_ptrs, float* const* pk_ptrs, const int64_t* sizes, int32_t count,
                      double m, void* stream) {
  if (!pq_ptrs || !pk_ptrs || !sizes) return bad("null pointer");
  if (count < 1 || count > 65535) return bad("count out of range");
  return run_ema(pq_ptrs, pk_ptrs, sizes, count, m, (hipStream_t)stream);
}

int pixpro_lars_sgd_step(float* const* p_ptrs, const float* const* g_ptrs, float* const* buf_ptrs,
                         const int64_t* sizes, const float* wd, const int32_t* adapt, int32_t count, float lr,
                         float momentum, float trust_coef, float eps, int32_t first_step, float* norms2,
                         void* stream) {
  if (!p_ptrs || !g_ptrs || !buf_ptrs || !sizes || !wd || !adapt || !norms2) return bad("null pointer");
  if (count < 1 || count > 65535) return bad("count out of range");
  return run_lars(p_ptrs, g_ptrs, buf_ptrs, sizes, wd, adapt, count, lr, momentum, trust_coef, eps, first_step,
                  norms2, (hipStream_t)stream);
}

int pixpro_corr_lookup(const float* level0, const float* level1, const float* level2, const float* level3,
                       int32_t num_levels, const float* coords, float* out, int32_t B, int32_t h, int32_t w, int32_t h2,
                       int32_t w2, int32_t radius, void* stream) {
  const float* levels[4] = {level0, level1, level2, level3};
  if (num_levels < 1 || num_levels > 4) return bad("num_levels must be 1..4");
  for (int l = 0; l < num_levels; ++l)
    if (!levels[l]) return bad("null pyramid level");
  if (!coords || !out) return bad("null pointer");
  if (B < 1 || h < 1 || w < 1 || radius < 0 || radius > 8) return bad("B, h, w must be positive, radius in 0..8");
  if ((h2 >> (num_levels - 1)) < 1 || (w2 >> (num_levels - 1)) < 1) return bad("pyramid level smaller than one pixel");
  return run_corr_lookup(levels, num_levels, coords, out, B, h, w, h2, w2, radius, (hipStream_t)stream);
}

static int bn_shape_ok(int64_t rows, int32_t C) {
  if (rows < 1 || C < 4 || (C & 3)) return bad("batch norm: rows >= 1 and C a positive multiple of 4 required");
  if (rows * (int64_t)C / 4 > 0x7fffffffffLL) return bad("batch norm: tensor too large");
  return PIXPRO_OK;
}

size_t pixpro_bn_workspace_bytes(int64_t rows, int32_t C) {
  if (rows < 1 || C < 4 || (C & 3)) return 0;
  return bn_workspace_floats(rows, C) * sizeof(float);
}

int pixpro_bn_stats(const float* x, int64_t rows, int32_t C, void* workspace, size_t workspace_bytes, float* packed,
                    void* stream) {
  if (!x || !workspace || !packed) return bad("null pointer");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (workspace_bytes < pixpro_bn_workspace_bytes(rows, C)) return bad("workspace too small");
  if ((uintptr_t)x & 15) return bad("x must be 16-byte aligned");
  return run_bn_stats(x, rows, C, (float*)workspace, packed, (hipStream_t)stream);
}

int pixpro_bn_apply(const float* gathered, int32_t world, const float* x, const float* residual, const float* weight,
                    const float* bias, float* running_mean, float* running_var, float momentum, float eps, int32_t relu,
                    int64_t rows, int32_t C, float* y, float* mean, float* invstd, float* scale_shift, void* stream) {
  if (!gathered || !x || !y || !mean || !invstd || !scale_shift) return bad("null pointer");
  if (world < 1) return bad("world must be >= 1");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)scale_shift) & 15)
    return bad("x, y, residual, scale_shift must be 16-byte aligned");
  return run_bn_apply(gathered, world, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu, rows, C,
                      y, mean, invstd, scale_shift, nullptr, (hipStream_t)stream);
}

size_t pixpro_bn_relu_bits_bytes(int64_t rows, int32_t C) {
  if (rows < 1 || C < 4 || (C & 3)) return 0;
  return bn_relu_bits_bytes(rows, C);
}

int pixpro_bn_apply_bits(const float* gathered, int32_t world, const float* x, const float* residual, const float* weight,
                         const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                         int64_t rows, int32_t C, float* y, float* mean, float* invstd, float* scale_shift,
                         void* relu_bits, void* stream) {
  if (!gathered || !x || !y || !mean || !invstd || !scale_shift || !relu_bits) return bad("null pointer");
  if (world < 1) return bad("world must be >= 1");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (!bn_relu_bits_bytes(rows, C)) return bad("batch norm: no ReLU bit record for this shape (pixpro_bn_relu_bits_bytes == 0)");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)scale_shift | (uintptr_t)relu_bits) & 15)
    return bad("x, y, residual, scale_shift, relu_bits must be 16-byte aligned");
  return run_bn_apply(gathered, world, x, residual, weight, bias, running_mean, running_var, momentum, eps, 1, rows, C,
                      y, mean, invstd, scale_shift, relu_bits, (hipStream_t)stream);
}

int pixpro_bn_backward_reduce(const float* dy, const float* x, const float* y, const float* scale_shift,
                              const float* mean, const float* invstd, int32_t relu, int64_t rows, int32_t C,
                              void* workspace, size_t workspace_bytes, float* sums, float* grad_weight,
                              float* grad_bias, void* stream) {
  if (!dy || !x || !mean || !invstd || !workspace || !sums) return bad("null pointer");
  if (relu && !y && !scale_shift) return bad("relu backward needs the forward output y or the forward's scale_shift");
  if ((uintptr_t)scale_shift & 15) return bad("scale_shift must be 16-byte aligned");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (workspace_bytes < pixpro_bn_workspace_bytes(rows, C)) return bad("workspace too small");
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)mean) & 15) return bad("dy, x, y, mean must be 16-byte aligned");
  return run_bn_bwd_reduce(dy, x, y, scale_shift, nullptr, mean, invstd, relu, rows, C, (float*)workspace, sums, grad_weight,
                           grad_bias, (hipStream_t)stream);
}

int pixpro_bn_backward_reduce_bits(const float* dy, const float* x, const void* relu_bits, const float* mean,
                                   const float* invstd, int64_t rows, int32_t C, void* workspace, size_t workspace_bytes,
                                   float* sums, float* grad_weight, float* grad_bias, void* stream) {
  if (!dy || !x || !relu_bits || !mean || !invstd || !workspace || !sums) return bad("null pointer");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (!bn_relu_bits_bytes(rows, C)) return bad("batch norm: no ReLU bit record for this shape (pixpro_bn_relu_bits_bytes == 0)");
  if (workspace_bytes < pixpro_bn_workspace_bytes(rows, C)) return bad("workspace too small");
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)relu_bits | (uintptr_t)mean) & 15)
    return bad("dy, x, relu_bits, mean must be 16-byte aligned");
  return run_bn_bwd_reduce(dy, x, nullptr, nullptr, relu_bits, mean, invstd, 1, rows, C, (float*)workspace, sums, grad_weight,
                           grad_bias, (hipStream_t)stream);
}

int pixpro_bn_backward_apply(const float* dy, const float* x, const float* y, const float* scale_shift,
                             const float* mean, const float* invstd, const float* weight, const float* sums,
                             const float* counts, int32_t world, int64_t count_stride, int32_t relu, int64_t rows,
                             int32_t C, float* dx, float* dresidual, void* stream) {
  if (!dy || !x || !mean || !invstd || !sums || !dx) return bad("null pointer");
  if (relu && !y && !scale_shift) return bad("relu backward needs the forward output y or the forward's scale_shift");
  if ((uintptr_t)scale_shift & 15) return bad("scale_shift must be 16-byte aligned");
  if (!counts || world < 1 || count_stride < 1) return bad("counts / world / count_stride: the ranks' row counts are required");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)dresidual | (uintptr_t)mean |
       (uintptr_t)invstd | (uintptr_t)weight | (uintptr_t)sums) & 15)
    return bad("batch norm backward operands must be 16-byte aligned");
  return run_bn_bwd_apply(dy, x, y, scale_shift, nullptr, mean, invstd, weight, sums, counts, world, count_stride, relu, rows,
                          C, dx, dresidual, (hipStream_t)stream);
}

int pixpro_bn_backward_apply_bits(const float* dy, const float* x, const void* relu_bits, const float* mean,
                                  const float* invstd, const float* weight, const float* sums, const float* counts,
                                  int32_t world, int64_t count_stride, int64_t rows, int32_t C, float* dx,
                                  float* dresidual, void* stream) {
  if (!dy || !x || !relu_bits || !mean || !invstd || !sums || !dx) return bad("null pointer");
  if (!counts || world < 1 || count_stride < 1) return bad("counts / world / count_stride: the ranks' row counts are required");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (!bn_relu_bits_bytes(rows, C)) return bad("batch norm: no ReLU bit record for this shape (pixpro_bn_relu_bits_bytes == 0)");
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)relu_bits | (uintptr_t)dx | (uintptr_t)dresidual | (uintptr_t)mean |
       (uintptr_t)invstd | (uintptr_t)weight | (uintptr_t)sums) & 15)
    return bad("batch norm backward operands must be 16-byte aligned");
  return run_bn_bwd_apply(dy, x, nullptr, nullptr, relu_bits, mean, invstd, weight, sums, counts, world, count_stride, 1, rows,
                          C, dx, dresidual, (hipStream_t)stream);
}

int pixpro_bn_masked_add(const float* a, const float* dy, const void* relu_bits, float* out, int64_t rows, int32_t C,
                         void* stream) {
  if (!dy || !relu_bits || !out) return bad("null pointer");
  if (int e = bn_shape_ok(rows, C)) return e;
  if (!bn_relu_bits_bytes(rows, C)) return bad("batch norm: no ReLU bit record for this shape (pixpro_bn_relu_bits_bytes == 0)");
  if (((uintptr_t)a | (uintptr_t)dy | (uintptr_t)relu_bits | (uintptr_t)out) & 15) return bad("operands must be 16-byte aligned");
  return run_bn_masked_add(a, dy, relu_bits, out, rows, C, (hipStream_t)stream);
}

int pixpro_bn_pool_supported(int32_t N, int32_t H, int32_t W, int32_t C) {
  if (N < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return 0;
  return bn_pool_supported(N, H, W, C) && pixpro_bn_workspace_bytes((int64_t)N * H * W, C) != 0;
}

int pixpro_bn_apply_pool(const float* gathered, int32_t world, const float* x, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float momentum, float eps, int32_t N, int32_t H, int32_t W,
                         int32_t C, float* pooled, float* mean, float* invstd, float* scale_shift, void* stream) {
  if (!gathered || !x || !pooled || !mean || !invstd || !scale_shift) return bad("null pointer");
  if (world < 1) return bad("world must be >= 1");
  if (!pixpro_bn_pool_supported(N, H, W, C)) return bad("batch norm + pooling: unsupported shape (pixpro_bn_pool_supported == 0)");
  if (((uintptr_t)x | (uintptr_t)pooled | (uintptr_t)scale_shift) & 15) return bad("x, pooled, scale_shift must be 16-byte aligned");
  return run_bn_apply_pool(gathered, world, x, weight, bias, running_mean, running_var, momentum, eps, N, H, W, C, pooled, mean,
                           invstd, scale_shift, (hipStream_t)stream);
}

int pixpro_bn_pool_backward_reduce(const float* dpooled, const float* x, const float* scale_shift, const float* mean,
                                   const float* invstd, int32_t N, int32_t H, int32_t W, int32_t C, void* workspace,
                                   size_t workspace_bytes, float* sums, float* grad_weight, float* grad_bias, void* stream) {
  if (!dpooled || !x || !scale_shift || !mean || !invstd || !workspace || !sums) return bad("null pointer");
  if (!pixpro_bn_pool_supported(N, H, W, C)) return bad("batch norm + pooling: unsupported shape (pixpro_bn_pool_supported == 0)");
  if (workspace_bytes < pixpro_bn_workspace_bytes((int64_t)N * H * W, C)) return bad("workspace too small");
  if (((uintptr_t)dpooled | (uintptr_t)x | (uintptr_t)scale_shift | (uintptr_t)mean) & 15)
    return bad("dpooled, x, scale_shift, mean must be 16-byte aligned");
  return run_bn_pool_bwd_reduce(dpooled, x, scale_shift, mean, invstd, N, H, W, C, (float*)workspace, sums, grad_weight, grad_bias,
                                (hipStream_t)stream);
}

int pixpro_bn_pool_backward_apply(const float* dpooled, const float* x, const float* scale_shift, const float* mean,
                                  const float* invstd, const float* weight, const float* sums, const float* counts,
                                  int32_t world, int64_t count_stride, int32_t N, int32_t H, int32_t W, int32_t C, float* dx,
                                  void* stream) {
  if (!dpooled || !x || !scale_shift || !mean || !invstd || !sums || !dx) return bad("null pointer");
  if (!counts || world < 1 || count_stride < 1) return bad("counts / world / count_stride: the ranks' row counts are required");
  if (!pixpro_bn_pool_supported(N, H, W, C)) return bad("batch norm + pooling: unsupported shape (pixpro_bn_pool_supported == 0)");
  if (((uintptr_t)dpooled | (uintptr_t)x | (uintptr_t)scale_shift | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)weight |
       (uintptr_t)sums | (uintptr_t)dx) & 15)
    return bad("batch norm backward operands must be 16-byte aligned");
  return run_bn_pool_bwd_apply(dpooled, x, scale_shift, mean, invstd, weight, sums, counts, world, count_stride, N, H, W, C, dx,
                               (hipStream_t)stream);
}

}  // extern "C"

#ifdef PIXPRO_STAMPS
namespace pixpro { extern __device__ unsigned long long g_stamps[64]; }
extern "C" int pixpro_debug_stamps(unsigned long long* host64) {
  return hipMemcpyFromSymbol(host64, HIP_SYMBOL(pixpro::g_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : 3;
}
#endif
