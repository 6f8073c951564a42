// Shared device helpers for libpixpro_hip (gfx950 only).  Compiled with -ffp-contract=off:
// every fused multiply-add below is written explicitly (__builtin_fmaf) where the reference
// arithmetic (ATen CPU kernels on an FMA-capable x86 build) fuses, and nowhere else, so the
// coordinate / flow pipeline can be reproduced bit for bit (see oracle/pixpro_oracle.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pixpro_hip.h"

#define WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace pixpro {

struct Dirs {
  pixpro_dir d[PIXPRO_MAX_DIRS];
};

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 (k-ordered fma chain).
// lane l supplies a = A[l&15][l>>4], b = B[l>>4][l&15]; reg r of the result is
// D[(l>>4)*4 + r][l&15].  PIXPRO_EMULATE_MFMA swaps in a shuffle-based VALU version with the
// identical contract (used only to bisect layout bugs on hardware).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef PIXPRO_EMULATE_MFMA
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float bk = __shfl(b, k * 16 + (l & 15), 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float ak = __shfl(a, k * 16 + (l >> 4) * 4 + r, 64);
      c[r] = __builtin_fmaf(ak, bk, c[r]);
    }
  }
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Cross-lane sums on the vector ALU (no ds_bpermute round trips through the LDS pipe, which is what __shfl_xor compiles
// to): gfx950's row swaps for the sum over the four 16-lane rows, DPP for the sum inside a row.
//   rows_sum(v): lanes l, l + 16, l + 32, l + 48 all receive (v[l] + v[l+16]) + (v[l+32] + v[l+48])  (l < 16)
//   wave_sum_valu(v): every lane receives the sum over the wave (row sums by quad / half-row / row mirrors, then rows_sum)
// (Inline assembly on purpose: __builtin_amdgcn_permlane16/32_swap handed two copies of one value comes back from hipcc 7.2
// with both results in ONE register - `v_permlane16_swap v7, v5 ; v_add_f32 v5, v7, v7` - i.e. 2 v instead of the row sums;
// tools/ubench/lane_sums.hip checks these helpers against the shuffles on the device.  s_nop 1 = the two wait states a VALU
// write of a swap operand needs before the swap reads it.)
__device__ __forceinline__ float rows_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // rows 1, 3 of a <-> rows 0, 2 of b
  a = a + b;
  b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // lanes 32..63 of a <-> lanes 0..31 of b
  return a + b;
}
#define PIXPRO_DPP_ADD(v, ctrl) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false)))
__device__ __forceinline__ float wave_sum_valu(float v) {
  v = PIXPRO_DPP_ADD(v, 0xB1);      // quad_perm [1,0,3,2]
  v = PIXPRO_DPP_ADD(v, 0x4E);      // quad_perm [2,3,0,1]
  v = PIXPRO_DPP_ADD(v, 0x141);     // row_half_mirror
  v = PIXPRO_DPP_ADD(v, 0x140);     // row_mirror
  return rows_sum(v);
}

// ---------------------------------------------------------------------------------------
// Crop geometry, contrast/models/PixPro.py:140-157,168-175.  c = coord row [10].
// ---------------------------------------------------------------------------------------
struct Crop {
  float bw, bh, x0, y0;
};

__device__ __forceinline__ Crop crop_of(const float* c, int H, int W) {
  Crop g;
  g.bw = (c[2] - c[0]) / (float)W;
  g.bh = (c[3] - c[1]) / (float)H;
  g.x0 = c[0];
  g.y0 = c[1];
  return g;
}

// The same from a work-group-uniform address, read through the scalar cache (constant address space => s_load): a
// vector load here queues behind whatever the CU's other waves have already requested - in the fused loss kernel the
// whole x tile, ~10 k cycles.
typedef const float __attribute__((address_space(4))) * sc_float_ptr;
typedef const int __attribute__((address_space(4))) * sc_int_ptr;
__device__ __forceinline__ Crop crop_of_uniform(const float* c, int H, int W) {
  sc_float_ptr s = (sc_float_ptr)(uintptr_t)c;
  const float c0 = s[0], c1 = s[1], c2 = s[2], c3 = s[3];
  Crop g;
  g.bw = (c2 - c0) / (float)W;
  g.bh = (c3 - c1) / (float)H;
  g.x0 = c0;
  g.y0 = c1;
  return g;
}

__device__ __forceinline__ float bin_diag(const Crop& g, float wo1, float ho1) {
  float a = g.bw * wo1, b = g.bh * ho1;
  return sqrtf(a * a + b * b);
}

// centre of bin (row i, col j) in original-frame pixels
__device__ __forceinline__ float centre_x(const Crop& g, int j, float wo1) {
  return (((float)j + 0.5f) * g.bw + g.x0) * wo1;
}
__device__ __forceinline__ float centre_y(const Crop& g, int i, float ho1) {
  return (((float)i + 0.5f) * g.bh + g.y0) * ho1;
}

// ---------------------------------------------------------------------------------------
// grid_sample(bilinear, zeros, align_corners=True) on one 2-channel image (ATen CPU order,
// oracle.bilinear_sample).  gx, gy normalised.  `pre` is applied to every tap of channel
// 0/1 before interpolation (used to sample normalize_flow(f) without materialising it).
// ---------------------------------------------------------------------------------------
struct TapNone {
  __device__ __forceinline__ float x(float v) const { return v; }
  __device__ __forceinline__ float y(float v) const { return v; }
};
struct TapNormFlow {  // util.py:342-348: 2*f/(wd-1), 2*f/(ht-1)
  float wd1, ht1;
  __device__ __forceinline__ float x(float v) const { return 2.f * v / wd1; }
  __device__ __forceinline__ float y(float v) const { return 2.f * v / ht1; }
};

// Tap sources: a dense 2-channel map, or the x8 bilinear up-sampling (upflow8,
// contrast/flow/utils/utils.py:87-89) of a low-resolution map evaluated on the fly with
// exactly the arithmetic of the materialised version (oracle.upflow8).
struct FetchDense {
  const float* __restrict__ cx;
  const float* __restrict__ cy;
  int Wf;
  __device__ __forceinline__ void operator()(int y, int x, float& vx, float& vy) const {
    const int o = y * Wf + x;
    vx = cx[o]; vy = cy[o];
  }
};

struct Up8Axis {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Up8Axis up8_axis(int dst, int n_in, float scale) {
  Up8Axis a;
  const float src = (float)dst * scale;
  a.i0 = min((int)src, n_in - 1);
  a.i1 = min(a.i0 + 1, n_in - 1);
  a.l1 = fminf(fmaxf(src - (float)a.i0, 0.f), 1.f);
  a.l0 = 1.f - a.l1;
  return a;
}
__device__ __forceinline__ float up8_scale(int n_in) {
  return n_in > 1 ? (float)(n_in - 1) / (float)(8 * n_in - 1) : 0.f;
}
// ATen's CPU bilinear up-sampling is two kernels with different operation orders (oracle.upflow8): outputs with
// out_H + out_W <= 128 take the product-weight form (cpu_upsample_linear_channels_last), everything larger - every
// production size - the separable form (upsample_generic_Nd_kernel_impl).  a,b = row y0 (x0,x1); c,d = row y1.
__host__ __device__ inline bool up8_separable(int h, int w) { return 8 * h + 8 * w > 128; }
// separable form, shared pieces: the x pass of one source row and the y pass
__device__ __forceinline__ float up8_row(float l, float r, float lx0, float lx1) { return __builtin_fmaf(l, lx0, r * lx1); }
__device__ __forceinline__ float up8_col(float t0, float t1, float ly0, float ly1) { return __builtin_fmaf(t0, ly0, t1 * ly1); }
__device__ __forceinline__ float up8_combine(float a, float b, float c, float d, const Up8Axis& ay,
                                             const Up8Axis& ax, bool sep) {
  if (sep) return 8.f * up8_col(up8_row(a, b, ax.l0, ax.l1), up8_row(c, d, ax.l0, ax.l1), ay.l0, ay.l1);
  float acc = b * (ay.l0 * ax.l1);
  acc = __builtin_fmaf(a, ay.l0 * ax.l0, acc);
  acc = __builtin_fmaf(c, ay.l1 * ax.l0, acc);
  acc = __builtin_fmaf(d, ay.l1 * ax.l1, acc);
  return 8.f * acc;
}
struct FetchUp8 {
  const float* __restrict__ lx;   // low-res channel 0 [h][w]
  const float* __restrict__ ly;   // low-res channel 1
  int h, w;
  float sy, sx;
  bool sep;
  __device__ __forceinline__ void operator()(int y, int x, float& vx, float& vy) const {
    const Up8Axis ay = up8_axis(y, h, sy), ax = up8_axis(x, w, sx);
    const int o00 = ay.i0 * w + ax.i0, o01 = ay.i0 * w + ax.i1, o10 = ay.i1 * w + ax.i0, o11 = ay.i1 * w + ax.i1;
    vx = up8_combine(lx[o00], lx[o01], lx[o10], lx[o11], ay, ax, sep);
    vy = up8_combine(ly[o00], ly[o01], ly[o10], ly[o11], ay, ax, sep);
  }
};

// Geometry of one bilinear sample: corner pixel, the four weights and which taps lie inside the map
// (bit k: tap k of (y0,x0), (y0,x1), (y1,x0), (y1,x1)); taps outside contribute zero (padding_mode='zeros').
struct BilinGeom {
  int x0, y0, in;
  float nw, ne, sw, se;
};
__device__ __forceinline__ BilinGeom bilin_geom(int Hf, int Wf, float gx, float gy) {
  BilinGeom b;
  const float ix = (gx + 1.f) * ((float)(Wf - 1) / 2.f);
  const float iy = (gy + 1.f) * ((float)(Hf - 1) / 2.f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
  b.nw = s * e; b.ne = s * w; b.sw = n * e; b.se = n * w;
  // float -> int keeping far-away (or NaN) coordinates out of range
  b.x0 = (xw > -2.f && xw < (float)(Wf + 1)) ? (int)xw : -2;
  b.y0 = (yn > -2.f && yn < (float)(Hf + 1)) ? (int)yn : -2;
  const int x1 = b.x0 + 1, y1 = b.y0 + 1;
  const bool bx0 = b.x0 >= 0 && b.x0 < Wf, bx1 = x1 >= 0 && x1 < Wf;
  const bool by0 = b.y0 >= 0 && b.y0 < Hf, by1 = y1 >= 0 && y1 < Hf;
  b.in = (by0 && bx0 ? 1 : 0) | (by0 && bx1 ? 2 : 0) | (by1 && bx0 ? 4 : 0) | (by1 && bx1 ? 8 : 0);
  return b;
}
__device__ __forceinline__ float bilin_combine(const BilinGeom& b, float v0, float v1, float v2, float v3) {
  return __builtin_fmaf(v3, b.se, __builtin_fmaf(v2, b.sw, __builtin_fmaf(v1, b.ne, v0 * b.nw)));
}

template <class Fetch, class Tap>
__device__ __forceinline__ void bilinear2(const Fetch& fetch, int Hf, int Wf, float gx, float gy, const Tap& tap,
                                          float& ox, float& oy) {
  const BilinGeom b = bilin_geom(Hf, Wf, gx, gy);
  const int x1 = b.x0 + 1, y1 = b.y0 + 1;
  float ax = 0.f, ay = 0.f, bxv = 0.f, byv = 0.f, cxv = 0.f, cyv = 0.f, dxv = 0.f, dyv = 0.f;
  if (b.in & 1) { fetch(b.y0, b.x0, ax, ay); ax = tap.x(ax); ay = tap.y(ay); }
  if (b.in & 2) { fetch(b.y0, x1, bxv, byv); bxv = tap.x(bxv); byv = tap.y(byv); }
  if (b.in & 4) { fetch(y1, b.x0, cxv, cyv); cxv = tap.x(cxv); cyv = tap.y(cyv); }
  if (b.in & 8) { fetch(y1, x1, dxv, dyv); dxv = tap.x(dxv); dyv = tap.y(dyv); }
  ox = bilin_combine(b, ax, bxv, cxv, dxv);
  oy = bilin_combine(b, ay, byv, cyv, dyv);
}

// grid_sample(nearest, zeros, align_corners=True) of a uint8 mask -> bool (PixPro.py:68-70)
__device__ __forceinline__ bool nearest_mask(const uint8_t* __restrict__ m, int Hf, int Wf, float gx,
                                             float gy) {
  const float ix = (gx + 1.f) * ((float)(Wf - 1) / 2.f);
  const float iy = (gy + 1.f) * ((float)(Hf - 1) / 2.f);
  const float xr = rintf(ix), yr = rintf(iy);
  if (!(xr >= 0.f && xr < (float)Wf && yr >= 0.f && yr < (float)Hf)) return false;
  return m[(int)yr * Wf + (int)xr] != 0;
}

// add_optical_flow for one point (PixPro.py:46-89).  (x, y) original-frame pixels.
struct FlowGeom {
  int Hf, Wf;
  float wo1, ho1;     // Wo-1, Ho-1
  float rw, rh;       // Wf/Wo, Hf/Ho evaluated in double then rounded (python floats)
  bool diff;
};

__device__ __forceinline__ FlowGeom flow_geom(int Hf, int Wf, int Ho, int Wo) {
  FlowGeom g;
  g.Hf = Hf; g.Wf = Wf;
  g.wo1 = (float)(Wo - 1); g.ho1 = (float)(Ho - 1);
  g.rw = (float)((double)Wf / (double)Wo);
  g.rh = (float)((double)Hf / (double)Ho);
  g.diff = (Hf != Ho) || (Wf != Wo);
  return g;
}

__device__ __forceinline__ void displace_point(const float* __restrict__ flow_n, const uint8_t* __restrict__ mask_n,
                                               const FlowGeom& g, float x, float y, float& ox, float& oy,
                                               bool& valid) {
  const float gx = 2.f * (x / g.wo1) - 1.f;
  const float gy = 2.f * (y / g.ho1) - 1.f;
  float fx, fy;
  bilinear2(FetchDense{flow_n, flow_n + (size_t)g.Hf * g.Wf, g.Wf}, g.Hf, g.Wf, gx, gy, TapNone(), fx, fy);
  valid = mask_n ? nearest_mask(mask_n, g.Hf, g.Wf, gx, gy) : true;
  if (g.diff) {
    ox = (x * g.rw + fx) / g.rw;
    oy = (y * g.rh + fy) / g.rh;
  } else {
    ox = x + fx;
    oy = y + fy;
  }
}

// displace_point in two halves for callers that have something to do while the taps are in flight (the fused loss
// kernel): `displace_fetch` requests all eight flow taps and the mask byte up front on clamped indices - one memory
// round trip instead of five dependent ones (bilinear2 fetches each tap under its own branch) - and
// `displace_finish` selects / combines exactly as displace_point does (a tap outside the map contributes 0 * weight).
struct PointTaps {
  BilinGeom b;
  float ax, ay, bx, by, cx, cy, dx, dy;
  bool m_in, has_mask;
  uint8_t m;
};
__device__ __forceinline__ PointTaps displace_fetch(const float* __restrict__ flow_n, const uint8_t* __restrict__ mask_n,
                                                    const FlowGeom& g, float x, float y) {
  const float gx = 2.f * (x / g.wo1) - 1.f;
  const float gy = 2.f * (y / g.ho1) - 1.f;
  PointTaps t;
  t.b = bilin_geom(g.Hf, g.Wf, gx, gy);
  const int xa = min(max(t.b.x0, 0), g.Wf - 1), xb = min(max(t.b.x0 + 1, 0), g.Wf - 1);
  const int ya = min(max(t.b.y0, 0), g.Hf - 1), yb = min(max(t.b.y0 + 1, 0), g.Hf - 1);
  const float* __restrict__ px = flow_n;
  const float* __restrict__ py = flow_n + (size_t)g.Hf * g.Wf;
  t.ax = px[ya * g.Wf + xa]; t.ay = py[ya * g.Wf + xa];
  t.bx = px[ya * g.Wf + xb]; t.by = py[ya * g.Wf + xb];
  t.cx = px[yb * g.Wf + xa]; t.cy = py[yb * g.Wf + xa];
  t.dx = px[yb * g.Wf + xb]; t.dy = py[yb * g.Wf + xb];
  // nearest_mask's position
  const float ix = (gx + 1.f) * ((float)(g.Wf - 1) / 2.f);
  const float iy = (gy + 1.f) * ((float)(g.Hf - 1) / 2.f);
  const float xr = rintf(ix), yr = rintf(iy);
  t.m_in = xr >= 0.f && xr < (float)g.Wf && yr >= 0.f && yr < (float)g.Hf;
  // unconditional byte load (no branch => no wait on the loads above): without a mask, some readable byte of the flow
  const uint8_t* __restrict__ mp = mask_n ? mask_n : reinterpret_cast<const uint8_t*>(flow_n);
  t.m = mp[t.m_in ? (int)yr * g.Wf + (int)xr : 0];       // raw byte: any use of it here would wait for the load
  t.has_mask = mask_n != nullptr;
  return t;
}
__device__ __forceinline__ void displace_finish(const PointTaps& t, const FlowGeom& g, float x, float y, float& ox,
                                                float& oy, bool& valid) {
  const int in = t.b.in;
  const float fx = bilin_combine(t.b, (in & 1) ? t.ax : 0.f, (in & 2) ? t.bx : 0.f, (in & 4) ? t.cx : 0.f,
                                 (in & 8) ? t.dx : 0.f);
  const float fy = bilin_combine(t.b, (in & 1) ? t.ay : 0.f, (in & 2) ? t.by : 0.f, (in & 4) ? t.cy : 0.f,
                                 (in & 8) ? t.dy : 0.f);
  valid = t.has_mask ? (t.m_in && t.m != 0) : true;
  if (g.diff) {
    ox = (x * g.rw + fx) / g.rw;
    oy = (y * g.rh + fy) / g.rh;
  } else {
    ox = x + fx;
    oy = y + fy;
  }
}

// forward_backward_consistency for ONE pixel (contrast/util.py:252-297).  (fx, fy): this direction's flow at the
// pixel; `bwd` fetches the opposite flow (pixel units unless is_norm).  Returns validity; cycle flow in (cx, cy).
struct FbcHead {
  float fnx, fny, c1x, c1y;
  bool inside;
};
__device__ __forceinline__ FbcHead fbc_head(float fx, float fy, int Hf, int Wf, int y, int x, int is_norm) {
  const float wd1 = (float)(Wf - 1), ht1 = (float)(Hf - 1);
  FbcHead h;
  h.fnx = fx; h.fny = fy;
  if (!is_norm) { h.fnx = 2.f * fx / wd1; h.fny = 2.f * fy / ht1; }        // normalize_flow
  h.c1x = (2.f * (float)x / wd1 - 1.f) + h.fnx;                             // util.py:276
  h.c1y = (2.f * (float)y / ht1 - 1.f) + h.fny;
  h.inside = (fabsf(h.c1x) < 1.f) && (fabsf(h.c1y) < 1.f);                  // :277
  return h;
}
// (bx, by): the opposite flow, normalised, sampled at (c1x, c1y)                 // :279
__device__ __forceinline__ bool fbc_tail(const FbcHead& h, float bx, float by, float alpha1, float a2, float& cx,
                                         float& cy) {
  cx = h.fnx + bx; cy = h.fny + by;                                         // :280
  const float cyc = cx * cx + cy * cy;                                      // :294
  const float eps = alpha1 * ((h.fnx * h.fnx + h.fny * h.fny) + (bx * bx + by * by)) + a2;   // :295
  return h.inside && ((cyc - eps) <= 0.f);                                  // :297
}
template <class Fetch>
__device__ __forceinline__ bool fbc_pixel(float fx, float fy, const Fetch& bwd, int Hf, int Wf, int y, int x,
                                          float alpha1, float a2, int is_norm, float& cx, float& cy) {
  const FbcHead h = fbc_head(fx, fy, Hf, Wf, y, x, is_norm);
  float bx, by;
  if (is_norm) bilinear2(bwd, Hf, Wf, h.c1x, h.c1y, TapNone(), bx, by);
  else bilinear2(bwd, Hf, Wf, h.c1x, h.c1y, TapNormFlow{(float)(Wf - 1), (float)(Hf - 1)}, bx, by);
  return fbc_tail(h, bx, by, alpha1, a2, cx, cy);
}

// alpha_2 / sqrt(h^2+w^2): torch.sqrt of an int64 tensor -> fp32, .item() -> double (util.py:290-292)
__host__ __device__ inline float fbc_a2(double alpha2, int Hf, int Wf) {
  const float diag = sqrtf((float)((int64_t)Hf * Hf + (int64_t)Wf * Wf));
  return (float)(alpha2 / (double)diag);
}

// ---- K6: concat_flow for one pixel.  `Source` yields the t-th flow of sample n as a fetch functor. -------------
struct Up8Source {     // lo [N,F1,2,h,w] (loader layout), up-sampled lazily
  const float* lo;
  int F1, h, w;
  __device__ __forceinline__ FetchUp8 at(int t, int n) const {
    const float* p = lo + ((size_t)n * F1 + t) * 2 * h * w;
    return FetchUp8{p, p + (size_t)h * w, h, w, up8_scale(h), up8_scale(w), up8_separable(h, w)};
  }
};

template <class Source>
__device__ __forceinline__ void chain_pixel(const Source& src, int F1, int n, int Hf, int Wf, int y, int x,
                                            int is_norm, float& ox, float& oy) {
  const float wd1 = (float)(Wf - 1), ht1 = (float)(Hf - 1);
  if (F1 == 1) {                                   // util.py:303-308: a copy of the only flow
    src.at(0, n)(y, x, ox, oy);
    if (is_norm) { ox = 2.f * ox / wd1; oy = 2.f * oy / ht1; }
    return;
  }
  const float x0 = (float)x, y0 = (float)y;
  const float x0n = 2.f * x0 / wd1 - 1.f, y0n = 2.f * y0 / ht1 - 1.f;   // normalize_coord, util.py:333-339
  if (is_norm) {                                   // util.py:316-319
    float cx = x0n, cy = y0n;
    for (int t = 0; t < F1; ++t) {
      float fx, fy;
      bilinear2(src.at(t, n), Hf, Wf, cx, cy, TapNormFlow{wd1, ht1}, fx, fy);
      cx = cx + fx; cy = cy + fy;
    }
    ox = cx - x0n; oy = cy - y0n;
  } else {                                         // util.py:320-323
    float cx = x0, cy = y0;
    for (int t = 0; t < F1; ++t) {
      float fx, fy;
      bilinear2(src.at(t, n), Hf, Wf, 2.f * cx / wd1 - 1.f, 2.f * cy / ht1 - 1.f, TapNone(), fx, fy);
      cx = cx + fx; cy = cy + fy;
    }
    ox = cx - x0; oy = cy - y0;
  }
}

// ---- Lazy flow mode (flow_up): add_optical_flow for one point with the composed flow (upflow8 + concat_flow of the
// low-res stack, contrast/util.py:183-194) and the consistency mask (:206-222) evaluated ON DEMAND from the
// 1/8-resolution maps - the same arithmetic, hence the same bits, as sampling the dense tensors
// pixpro_flow_prepare would have written.
//   flow  : bilinear sample of the composed own flow at the point      (PixPro.py:64)  -> 4 pixels of it
//   valid : mask at the nearest pixel (PixPro.py:68-70) = forward_backward_consistency of that ONE pixel
//           -> the own flow there + the opposite flow at the 4 taps of its displaced position
// Nine composed-flow pixels per point, split in two stages so that a kernel can spread them over lanes:
//   stage A, s = 0..3: own flow at tap s of the point; s = 4: own flow at the nearest pixel
//   stage B, s = 0..3: opposite flow (normalised) at tap s of the nearest pixel's displaced position
struct LazySrc {
  const float* own;   // this sample's [F1][2][h][w] of this direction
  const float* opp;   // ... of the opposite direction
  int F1, h, w;
  bool with_mask;
  float alpha1, a2;
};
struct LazyPoint {
  float gx, gy;       // normalised sample position
  BilinGeom b;        // its bilinear taps in the composed map
  int xr, yr;         // nearest pixel
  bool near_in;       // ... lies inside the map
};
__device__ __forceinline__ LazyPoint lazy_point(const FlowGeom& g, float x, float y) {
  LazyPoint p;
  p.gx = 2.f * (x / g.wo1) - 1.f;
  p.gy = 2.f * (y / g.ho1) - 1.f;
  p.b = bilin_geom(g.Hf, g.Wf, p.gx, p.gy);
  const float ix = (p.gx + 1.f) * ((float)(g.Wf - 1) / 2.f);
  const float iy = (p.gy + 1.f) * ((float)(g.Hf - 1) / 2.f);
  const float xr = rintf(ix), yr = rintf(iy);
  p.near_in = xr >= 0.f && xr < (float)g.Wf && yr >= 0.f && yr < (float)g.Hf;
  p.xr = p.near_in ? (int)xr : 0;
  p.yr = p.near_in ? (int)yr : 0;
  return p;
}
__device__ __forceinline__ void lazy_flow_at(const float* lo, const LazySrc& s, const FlowGeom& g, int y, int x,
                                             float& vx, float& vy) {
  chain_pixel(Up8Source{lo, s.F1, s.h, s.w}, s.F1, 0, g.Hf, g.Wf, y, x, 0, vx, vy);
}
__device__ __forceinline__ void lazy_stage_a(const LazySrc& s, const FlowGeom& g, const LazyPoint& p, int k, float& vx,
                                             float& vy) {
  vx = 0.f; vy = 0.f;
  if (k < 4) {
    if ((p.b.in >> k) & 1) lazy_flow_at(s.own, s, g, p.b.y0 + (k >> 1), p.b.x0 + (k & 1), vx, vy);
  } else if (s.with_mask && p.near_in) {
    lazy_flow_at(s.own, s, g, p.yr, p.xr, vx, vy);
  }
}
__device__ __forceinline__ void lazy_stage_b(const LazySrc& s, const FlowGeom& g, const LazyPoint& p, float px, float py,
                                             int k, float& vx, float& vy) {
  vx = 0.f; vy = 0.f;
  if (!(s.with_mask && p.near_in)) return;
  const FbcHead h = fbc_head(px, py, g.Hf, g.Wf, p.yr, p.xr, 0);
  const BilinGeom b = bilin_geom(g.Hf, g.Wf, h.c1x, h.c1y);
  if ((b.in >> k) & 1) {
    lazy_flow_at(s.opp, s, g, b.y0 + (k >> 1), b.x0 + (k & 1), vx, vy);
    const TapNormFlow tap{(float)(g.Wf - 1), (float)(g.Hf - 1)};
    vx = tap.x(vx); vy = tap.y(vy);
  }
}
// A: stage-A results {x,y} x 5, B: stage-B results {x,y} x 4 (registers or LDS)
__device__ __forceinline__ void lazy_finish(const LazySrc& s, const FlowGeom& g, const LazyPoint& p, float x, float y,
                                            const float* A, const float* B, float& ox, float& oy, bool& valid) {
  const float fx = bilin_combine(p.b, A[0], A[2], A[4], A[6]);
  const float fy = bilin_combine(p.b, A[1], A[3], A[5], A[7]);
  valid = true;
  if (s.with_mask) {
    valid = false;
    if (p.near_in) {
      const FbcHead h = fbc_head(A[8], A[9], g.Hf, g.Wf, p.yr, p.xr, 0);
      const BilinGeom b = bilin_geom(g.Hf, g.Wf, h.c1x, h.c1y);
      float cx, cy;
      valid = fbc_tail(h, bilin_combine(b, B[0], B[2], B[4], B[6]), bilin_combine(b, B[1], B[3], B[5], B[7]), s.alpha1,
                       s.a2, cx, cy);
    }
  }
  if (g.diff) {
    ox = (x * g.rw + fx) / g.rw;
    oy = (y * g.rh + fy) / g.rh;
  } else {
    ox = x + fx;
    oy = y + fy;
  }
}
// all nine pixels on one lane (staged path, and the single-frame case of the fused kernel)
__device__ __forceinline__ void displace_point_lazy(const LazySrc& s, const FlowGeom& g, float x, float y, float& ox,
                                                    float& oy, bool& valid) {
  const LazyPoint p = lazy_point(g, x, y);
  float A[10], B[8];
#pragma unroll
  for (int k = 0; k < 5; ++k) lazy_stage_a(s, g, p, k, A[2 * k], A[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 4; ++k) lazy_stage_b(s, g, p, A[8], A[9], k, B[2 * k], B[2 * k + 1]);
  lazy_finish(s, g, p, x, y, A, B, ox, oy, valid);
}
__device__ __forceinline__ LazySrc lazy_src(const pixpro_problem& pb, const pixpro_dir& d, int n) {
  const int F1 = pb.lo_frames > 1 ? pb.lo_frames : 1;
  const size_t off = (size_t)n * F1 * 2 * pb.lo_h * pb.lo_w;
  return LazySrc{d.lo_own + off, d.lo_opp + off, F1, pb.lo_h, pb.lo_w, pb.lazy_mask != 0, pb.alpha1,
                 fbc_a2(pb.alpha2, 8 * pb.lo_h, 8 * pb.lo_w)};
}

// Frame size for one problem: explicit device tensor, explicit ints, or coord_q[0][9], [0][8].
__device__ __forceinline__ void frame_size(const pixpro_problem& pb, const pixpro_dir& d, int& Ho, int& Wo) {
  if (d.size_dev) { Ho = d.size_dev[0]; Wo = d.size_dev[1]; }
  else if (pb.Ho > 0) { Ho = pb.Ho; Wo = pb.Wo; }
  else { Ho = (int)d.coord_q[9]; Wo = (int)d.coord_q[8]; }
}

__device__ __forceinline__ void frame_size_uniform(const pixpro_problem& pb, const pixpro_dir& d, int& Ho, int& Wo) {
  if (d.size_dev) { sc_int_ptr s = (sc_int_ptr)(uintptr_t)d.size_dev; Ho = s[0]; Wo = s[1]; }
  else if (pb.Ho > 0) { Ho = pb.Ho; Wo = pb.Wo; }
  else { sc_float_ptr s = (sc_float_ptr)(uintptr_t)d.coord_q; Ho = (int)s[9]; Wo = (int)s[8]; }
}

// ---- asynchronous global -> LDS copies (global_load_lds_dwordx4) ------------------------------------------------------
// One wave instruction moves 64 x 16 B: every lane names its own SOURCE address, the destination is the wave-uniform LDS
// base + 16 * lane.  A slab image is therefore linear in LDS and any swizzle is applied to the source side and to the
// reads (the same involution on both).  The copies are counted in vmcnt; nothing orders them against LDS reads except the
// issuing wave's s_waitcnt vmcnt(n) followed by a barrier, so the loops below use raw s_barrier (a __syncthreads() would
// drain every copy in flight with vmcnt(0)).
// The copy is issued from inline assembly on purpose: hipcc tracks an LDS-DMA it knows about as a pending LDS write and,
// where it cannot prove that a later ds_read touches another buffer of the ring, puts s_waitcnt vmcnt(0) in front of the
// read - which drains the slabs that were requested AHEAD as well (found in the strip kernels' ISA in round 3: every slab
// waited for the copies issued in the same iteration).  The waits are placed by hand below, by count.  M0 (the LDS base of
// the copy) is saved and restored inside the statement.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {          // LDS traffic of this wave retired, then the work-group barrier
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// A = clamp(S,min=c) (+1e-6 if p<1) ** p  and  D = dA/dS  (PixPro.py:355-358).
__device__ __forceinline__ void pow_map(float S, float p, float clampv, float& A, float& D) {
  float B = fmaxf(S, clampv);
  if (p < 1.f) B = B + 1e-6f;
  const float pass = (S >= clampv) ? 1.f : 0.f;
  if (p == 2.f) { A = B * B; D = 2.f * B * pass; }
  else if (p == 1.f) { A = B; D = pass; }
  else { A = powf(B, p); D = p * powf(B, p - 1.f) * pass; }
}

}  // namespace pixpro
