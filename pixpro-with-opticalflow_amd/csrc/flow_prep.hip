// Flow preparation kernels (HBM-bound; lane = output column, coalesced along x).
//
// Map of the file:  generic one-thread-per-pixel forms (k_upflow8, k_chain, k_fbc, k_prepare_single: stand-alone API,
// flow_up = False, maps below ATen's small-output threshold)  ->  nframe = 2: k_prepare_cert (flows + masks settled by a
// certificate: per-cell box table, whole runs of rows at once) and k_prepare_staged (marked tiles, taps from LDS)  ->
// nframe > 2: k_chain_up8 (composed flows + bounding boxes of their 8 x 8 blocks), k_fbc_cert_dense (certificate from
// those boxes, whole blocks from their own box), k_fbc_pair / k_fbc_pair_marked (taps from a mean-shifted LDS region)  ->
// run_flow_prepare (routes, PIXPRO_FLOW_PATH).  DESIGN.md 4.3 / 4.4 has the reasoning and the measurements.
//
// Reference: contrast/util.py:174-248 (apply_optical_flow, file mode), :300-330 (concat_flow),
// :252-297 (forward_backward_consistency), :333-366 (normalise helpers, calc_mask_ratio),
// contrast/flow/utils/utils.py:87-89 (upflow8), contrast/models/PixPro.py:46-89
// (add_optical_flow).  Arithmetic order follows oracle/pixpro_oracle.py (bit-exact targets).
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace pixpro {
namespace {

// Dense outputs are written once and read back only at the loss kernels' sample points: k_prepare_cert writes them with
// streaming (non-temporal) stores (isolated launch at N = 256: 0.955 against 0.993 ms median; -DFLOW_PLAIN_STORE reverts).
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st2(float* p, float2 v) {
#ifndef FLOW_PLAIN_STORE
  v2f_t w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<v2f_t*>(p));
#else
  *reinterpret_cast<float2*>(p) = v;
#endif
}
__device__ __forceinline__ void st_u16(uint8_t* p, uint16_t v) {
#ifndef FLOW_PLAIN_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<uint16_t*>(p));
#else
  *reinterpret_cast<uint16_t*>(p) = v;
#endif
}

constexpr int TPB = 256;
inline unsigned blocks_for(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

// ---- K5: upflow8 ----------------------------------------------------------------------
__global__ void k_upflow8(const float* __restrict__ in, float* __restrict__ out, int64_t planes, int h, int w) {
  const int Wf = 8 * w, Hf = 8 * h;
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)planes * Hf * Wf) return;
  const int x = (int)(idx % Wf), y = (int)((idx / Wf) % Hf);
  const size_t pl = idx / ((size_t)Wf * Hf);
  const float* p = in + pl * h * w;
  const Up8Axis ay = up8_axis(y, h, up8_scale(h)), ax = up8_axis(x, w, up8_scale(w));
  out[idx] = up8_combine(p[ay.i0 * w + ax.i0], p[ay.i0 * w + ax.i1], p[ay.i1 * w + ax.i0], p[ay.i1 * w + ax.i1],
                         ay, ax, up8_separable(h, w));
}

// ---- K6: concat_flow ------------------------------------------------------------------
// One chain per pixel.  `Source` yields the t-th flow of sample n as a fetch functor.
struct DenseSource {   // flows [F1,N,2,Hf,Wf]
  const float* flows;
  int N, Hf, Wf;
  __device__ __forceinline__ FetchDense at(int t, int n) const {
    const float* p = flows + ((size_t)t * N + n) * 2 * Hf * Wf;
    return FetchDense{p, p + (size_t)Hf * Wf, Wf};
  }
};
struct LoSource {      // lo [N,F1,2,h,w] used at its own resolution (flow_up == False)
  const float* lo;
  int F1, h, w;
  __device__ __forceinline__ FetchDense at(int t, int n) const {
    const float* p = lo + ((size_t)n * F1 + t) * 2 * h * w;
    return FetchDense{p, p + (size_t)h * w, w};
  }
};

template <class Source>
__global__ void k_chain(Source s0, Source s1, float* __restrict__ out0, float* __restrict__ out1, int F1, int N,
                        int Hf, int Wf, int is_norm) {
  const size_t HW = (size_t)Hf * Wf;
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)N * HW) return;
  const int n = (int)(idx / HW);
  const size_t r = idx % HW;
  const int y = (int)(r / Wf), x = (int)(r % Wf);
  const Source& src = blockIdx.y ? s1 : s0;
  float* out = blockIdx.y ? out1 : out0;
  float ox, oy;
  chain_pixel(src, F1, n, Hf, Wf, y, x, is_norm, ox, oy);
  out[(size_t)n * 2 * HW + r] = ox;
  out[(size_t)n * 2 * HW + HW + r] = oy;
}

// ---- K7: forward_backward_consistency: fbc_pixel lives in common.h (shared with the lazy flow mode) ----
__global__ void k_fbc(const float* __restrict__ f, const float* __restrict__ bw, uint8_t* __restrict__ mask,
                      float* __restrict__ cycle, int32_t* __restrict__ invalid, int N, int Hf, int Wf, float alpha1,
                      float a2, int is_norm) {
  const size_t HW = (size_t)Hf * Wf;
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  bool bad = false;
  int n = 0;
  if (idx < (size_t)N * HW) {
    n = (int)(idx / HW);
    const size_t r = idx % HW;
    const int y = (int)(r / Wf), x = (int)(r % Wf);
    const float* fp = f + (size_t)n * 2 * HW;
    const float* bp = bw + (size_t)n * 2 * HW;
    float cx, cy;
    const bool ok = fbc_pixel(fp[r], fp[HW + r], FetchDense{bp, bp + HW, Wf}, Hf, Wf, y, x, alpha1, a2, is_norm,
                              cx, cy);
    mask[idx] = ok ? 1 : 0;
    if (cycle) { cycle[(size_t)n * 2 * HW + r] = cx; cycle[(size_t)n * 2 * HW + HW + r] = cy; }
    bad = !ok;
  }
  if (invalid) {
    // a block never straddles more than two samples when HW >= TPB; count per wave and add once
    const unsigned long long bits = __ballot(bad);
    const int n0 = __shfl(n, 0, 64);
    const bool uniform = __all(n == n0 || idx >= (size_t)N * HW);
    if (uniform) {
      if ((threadIdx.x & 63) == 0 && bits) atomicAdd(&invalid[n0], __popcll(bits));
    } else if (bad) {
      atomicAdd(&invalid[n], 1);
    }
  }
}

// ---- fused file-mode apply_optical_flow for F-1 == 1 (nframe = 2) -------------------------
// Per pixel, both directions: composed flow = up8(lo) (a copy, util.py:303-308) and the two
// consistency masks, with the opposite direction's taps recomputed from the low-res map.
// HBM traffic: read the low-res maps, write 2 flows + 2 masks; nothing else.
template <class Source>
__global__ void k_prepare_single(Source sf, Source sb, float* __restrict__ flow_fwd, float* __restrict__ flow_bwd,
                                 uint8_t* __restrict__ mask_fwd, uint8_t* __restrict__ mask_bwd,
                                 int32_t* __restrict__ invalid, int N, int Hf, int Wf, float alpha1, float a2) {
  const size_t HW = (size_t)Hf * Wf;
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  bool badf = false, badb = false;
  int n = 0;
  if (idx < (size_t)N * HW) {
    n = (int)(idx / HW);
    const size_t r = idx % HW;
    const int y = (int)(r / Wf), x = (int)(r % Wf);
    const auto ff = sf.at(0, n);
    const auto fb = sb.at(0, n);
    float fx, fy, bx, by;
    ff(y, x, fx, fy);
    fb(y, x, bx, by);
    flow_fwd[(size_t)n * 2 * HW + r] = fx; flow_fwd[(size_t)n * 2 * HW + HW + r] = fy;
    flow_bwd[(size_t)n * 2 * HW + r] = bx; flow_bwd[(size_t)n * 2 * HW + HW + r] = by;
    if (mask_fwd) {
      float cx, cy;
      const bool okf = fbc_pixel(fx, fy, fb, Hf, Wf, y, x, alpha1, a2, 0, cx, cy);
      const bool okb = fbc_pixel(bx, by, ff, Hf, Wf, y, x, alpha1, a2, 0, cx, cy);
      mask_fwd[idx] = okf ? 1 : 0;
      mask_bwd[idx] = okb ? 1 : 0;
      badf = !okf; badb = !okb;
    }
  }
  if (invalid && mask_fwd) {
    const unsigned long long bf = __ballot(badf), bb = __ballot(badb);
    const int n0 = __shfl(n, 0, 64);
    const bool uniform = __all(n == n0 || idx >= (size_t)N * HW);
    if (uniform) {
      if ((threadIdx.x & 63) == 0) {
        if (bf) atomicAdd(&invalid[n0], __popcll(bf));
        if (bb) atomicAdd(&invalid[N + n0], __popcll(bb));
      }
    } else {
      if (badf) atomicAdd(&invalid[n], 1);
      if (badb) atomicAdd(&invalid[N + n], 1);
    }
  }
}

// =====================================================================================================
// Production kernels (flow_up, output sizes on ATen's separable up-sampling path: every real frame size).
//
// Round 1 re-derived the x8 up-sampling of the OPPOSITE direction at each of the 2 x 4 consistency taps of every pixel
// (~350 VALU instructions and 36 ds_read_b128 per pixel).  Here every block first stages, once, what its taps will
// need - the up-sampled and already NORMALISED (2f/(Wf-1), 2f/(Hf-1)) flows of both directions over its tile plus a
// halo - in LDS, and a tap becomes one 8-byte LDS read.  The staging exploits that the reference's up-sampling is
// separable (common.h: t = fma(l, lx0, r*lx1) along x per source row, then fma(t0, ly0, t1*ly1) along y): a thread owns
// one column of the region, keeps the two x-interpolated source rows in registers and walks down its rows, so a
// staged pixel costs one fma+mul per channel plus the exact normalising division.  The un-normalised flows of the tile
// (the API's dense outputs) are stored to HBM from the same loop.  Taps that leave the staged region (displacements
// beyond the halo) take a slow, generic path with identical arithmetic (fbc_pixel over FetchUp8).
//
// Exact division by the per-launch constants (Wf-1), (Hf-1): q0 = a*r, q = fma(fma(-q0,b,a), r, q0) with r = RN(1/b)
// is the correctly rounded quotient (Markstein), 3 instructions instead of ~10.
// =====================================================================================================
struct DivC {
  float b, rb;
};
inline DivC make_divc(float b) { return DivC{b, (float)(1.0 / (double)b)}; }
__device__ __forceinline__ float div_c(float a, const DivC& d) {
  const float q0 = a * d.rb;
  const float rem = __builtin_fmaf(-q0, d.b, a);
  return __builtin_fmaf(rem, d.rb, q0);
}

struct Up8Geom {
  int h, w, Hf, Wf;
  float sy, sx;      // (h-1)/(8h-1), (w-1)/(8w-1)
  DivC wd, ht;       // Wf-1, Hf-1
  float hx, hy;      // (Wf-1)/2, (Hf-1)/2
  float alpha1, a2;
};

// Lean axis of the x8 up-sampling for positions inside the image: src = dst*scale < n_in always, so ATen's clamps of
// i0 and of the weight are no-ops; the +1 neighbour comes from a replicate-padded LDS window.
struct AxF {
  int i0;
  float l0, l1;
};
__device__ __forceinline__ AxF axf(int dst, float scale) {
  AxF a;
  const float src = (float)dst * scale;
  a.i0 = (int)src;
  a.l1 = src - (float)a.i0;
  a.l0 = 1.f - a.l1;
  return a;
}

// Tile / region geometry.  A block of 640 threads (10 waves) owns TX x TY output pixels; the staged region adds HL
// pixels on the left / top and HR on the right / bottom (one of them is the +1 neighbour of the last tap).
constexpr int TX = 64, TY = 40, HL = 8, HR = 8;
constexpr int RW = TX + HL + HR, RH = TY + HL + HR;          // 80 x 56
constexpr int NGROUP = 8;                                     // row groups of the staging loop
constexpr int NT = (NGROUP + NGROUP / 4) * 64, NWAVE = NT / 64;   // 8 waves stage the tile columns, 2 the halo columns
constexpr int ROWS_PER_GROUP = RH / NGROUP;                   // 7 region rows per staging thread
constexpr int LMARG = 3;                                      // extra cells around the region's cells: taps up to ~ +-32 px away
constexpr int LW_W = RW / 8 + 4 + 2 * LMARG, LW_H = RH / 8 + 4 + 2 * LMARG;   // low-res window (replicate-padded): 20 x 17 cells
static_assert(RH % NGROUP == 0 && TY % NWAVE == 0 && (RW - TX) * 4 == 64 && NGROUP % 4 == 0,
              "thread mapping of the staging loop");

// which region column / first region row a staging thread owns
__device__ __forceinline__ void stage_slot(int t, int& rc, int& r0) {
  if (t < NGROUP * 64) {                  // waves 0..NGROUP-1: the 64 tile columns, one row group per wave
    rc = HL + (t & 63);
    r0 = (t >> 6) * ROWS_PER_GROUP;
  } else {                                // the remaining waves: the 16 halo columns x 4 row groups each
    const int l = t - NGROUP * 64, j = l & 15;
    rc = j < HL ? j : TX + j;
    r0 = (l >> 4) * ROWS_PER_GROUP;
  }
}

// slow path of a tap set that leaves the staged region: the generic per-pixel routine over the lazily up-sampled maps
__device__ __attribute__((noinline)) bool fbc_generic(const float* own, const float* opp, int h, int w, int y, int x,
                                                      float alpha1, float a2) {
  const bool sep = up8_separable(h, w);
  const FetchUp8 fo{own, own + (size_t)h * w, h, w, up8_scale(h), up8_scale(w), sep};
  const FetchUp8 fp{opp, opp + (size_t)h * w, h, w, up8_scale(h), up8_scale(w), sep};
  float fx, fy, cx, cy;
  fo(y, x, fx, fy);
  return fbc_pixel(fx, fy, fp, 8 * h, 8 * w, y, x, alpha1, a2, 0, cx, cy);
}

// validity of one pixel from its own normalised flow fn and the staged normalised flows of the opposite direction
__device__ __forceinline__ bool fbc_staged(float2 fn, const float2* __restrict__ opp, const Up8Geom& g, float c0x,
                                           float c0y, int rx0, int ry0, bool& slow) {
  const float c1x = c0x + fn.x, c1y = c0y + fn.y;
  if (!((fabsf(c1x) < 1.f) && (fabsf(c1y) < 1.f))) return false;
  const float ix = (c1x + 1.f) * g.hx, iy = (c1y + 1.f) * g.hy;
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
  const int rx = (int)xw - rx0, ry = (int)yn - ry0;
  // x0+1 == Wf (y0+1 == Hf) only when the weight w (n) is exactly 0: the region's clamped duplicate is finite
  if (!((unsigned)rx < (unsigned)(RW - 1) && (unsigned)ry < (unsigned)(RH - 1))) { slow = true; return false; }
  const int o = ry * RW + rx;
  const float2 t0 = opp[o], t1 = opp[o + 1], t2 = opp[o + RW], t3 = opp[o + RW + 1];
  const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
  const float bx = __builtin_fmaf(t3.x, se, __builtin_fmaf(t2.x, sw, __builtin_fmaf(t1.x, ne, t0.x * nw)));
  const float by = __builtin_fmaf(t3.y, se, __builtin_fmaf(t2.y, sw, __builtin_fmaf(t1.y, ne, t0.y * nw)));
  const float cx = fn.x + bx, cy = fn.y + by;
  const float cyc = cx * cx + cy * cy;
  const float eps = g.alpha1 * ((fn.x * fn.x + fn.y * fn.y) + (bx * bx + by * by)) + g.a2;
  return (cyc - eps) <= 0.f;
}

// The same decision from the low-res CELLS in LDS (4 taps x separable up-sampling + exact normalisation, ~120 VALU): the
// middle path between the staged region / the certificate and the generic evaluation from global memory.  `cell(ry, rx)`
// returns the opposite flow's cell at window coordinates; the window holds cw_h x cw_w cells from (cwy0, cwx0).
template <class Cell>
__device__ __forceinline__ bool fbc_cells(float2 fn, const Up8Geom& g, float c0x, float c0y, int cwx0, int cwy0, int cw_w,
                                          int cw_h, Cell cell, bool& slow) {
  const float c1x = c0x + fn.x, c1y = c0y + fn.y;
  if (!((fabsf(c1x) < 1.f) && (fabsf(c1y) < 1.f))) return false;
  const float ix = (c1x + 1.f) * g.hx, iy = (c1y + 1.f) * g.hy;
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
  const int x0 = (int)xw, y0 = (int)yn;
  // x0 + 1 == Wf (y0 + 1 == Hf) only when the weight w (n) is exactly 0: the clamped duplicate is finite
  const AxF ax[2] = {axf(x0, g.sx), axf(min(x0 + 1, g.Wf - 1), g.sx)};
  const AxF ay[2] = {axf(y0, g.sy), axf(min(y0 + 1, g.Hf - 1), g.sy)};
  const int rx0 = ax[0].i0 - cwx0, rx1 = ax[1].i0 - cwx0, ry0 = ay[0].i0 - cwy0, ry1 = ay[1].i0 - cwy0;
  if (!(rx0 >= 0 && rx1 < cw_w - 1 && ry0 >= 0 && ry1 < cw_h - 1)) { slow = true; return false; }
  float tx[4], ty[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const AxF& qy = ay[k >> 1];
    const AxF& qx = ax[k & 1];
    const int ry = (k >> 1) ? ry1 : ry0, rx = (k & 1) ? rx1 : rx0;
    const float2 a = cell(ry, rx), b = cell(ry, rx + 1), c = cell(ry + 1, rx), d = cell(ry + 1, rx + 1);
    tx[k] = div_c(16.f * up8_col(up8_row(a.x, b.x, qx.l0, qx.l1), up8_row(c.x, d.x, qx.l0, qx.l1), qy.l0, qy.l1), g.wd);
    ty[k] = div_c(16.f * up8_col(up8_row(a.y, b.y, qx.l0, qx.l1), up8_row(c.y, d.y, qx.l0, qx.l1), qy.l0, qy.l1), g.ht);
  }
  const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
  const float bx = __builtin_fmaf(tx[3], se, __builtin_fmaf(tx[2], sw, __builtin_fmaf(tx[1], ne, tx[0] * nw)));
  const float by = __builtin_fmaf(ty[3], se, __builtin_fmaf(ty[2], sw, __builtin_fmaf(ty[1], ne, ty[0] * nw)));
  const float cx = fn.x + bx, cy = fn.y + by;
  const float cyc = cx * cx + cy * cy;
  const float eps = g.alpha1 * ((fn.x * fn.x + fn.y * fn.y) + (bx * bx + by * by)) + g.a2;
  return (cyc - eps) <= 0.f;
}

// ---- fused file-mode apply_optical_flow, nframe = 2: both composed flows + both masks ------------------------------
// k_prepare_cert writes the flows of every tile and the masks of every tile it can CERTIFY; the rest it marks for
// k_prepare_staged.  The certificate: a pixel of direction d is valid iff it stays inside the frame and
//     |fn + b'|^2 <= alpha1 (|fn|^2 + |b'|^2) + a2,          b' = bilinear sample of the opposite normalised flow,
// (util.py:276-297).  b' is a convex combination of four up-sampled values, each a convex combination of low-res cells,
// so it lies in the bounding box [L,U] of the cells its taps can touch.  Which cells those are follows from the pixel's
// own low-res cell and the range of the tile's own flow (a few cells around the cell the pixel is displaced to); the
// kernel tabulates that box once per low-res cell of the tile, in LDS, with two separable min/max passes.  Then
//     |fn + b'| <= max(|fn + L|, |fn + U|) per component (boxes inflated by 1e-5 relative for every rounding on the way;
//     a correctly rounded division is monotone), and since the right-hand side of the test is >= a2,
//     max(..)_x^2 + max(..)_y^2 <= a2 (1 - 2e-5)   ==>   valid, whatever the taps are.
// Pixels inside the frame that are not certified are evaluated exactly (generic path), so the masks are bit-identical
// either way; the certificate only decides who pays for the taps.  A tile in which the certificate cannot work (own flow
// range wider than the table reaches, or most boxes wider than the tolerance: motion boundaries, occlusions) is left
// to the staged kernel.  A certified pixel costs one fused up-sampling step per channel, the exact normalisation and
// ~12 instructions per direction - no taps, no halo.
constexpr int TYC = 2 * TY;                              // a certified tile = two staged tiles on top of each other ...
constexpr int TXC = 2 * TX;                              // ... and two side by side: 128 x 80 pixels, a lane owns TWO columns
constexpr int W0_W = TXC / 8 + 4, W0_H = TYC / 8 + 4;    // low-res cells under a tile: at most 20 x 14
constexpr int MARG = 4;                                  // cells of margin around them: own flow RANGE up to ~48 px
constexpr int WW = W0_W + 2 * MARG, WH = W0_H + 2 * MARG;   // cell window: 28 x 22
constexpr float BOX_SLOP = 1e-5f;
constexpr uint8_t TILE_MARK = 0xFF;                      // first mask byte of a tile left to k_prepare_staged
constexpr int NT_C = 320, NWAVE_C = NT_C / 64, ROWS_C = TYC / NWAVE_C;   // 5 waves x 16 rows, lane = column
static_assert(TYC % NWAVE_C == 0 && 4 * ROWS_C <= 64, "row mapping / deferred-pixel bit mask");

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// [lo, hi] of raw cell values -> box of the normalised up-sampled flow 16*acc/b, inflated for rounding
__device__ __forceinline__ void norm_box(float lo, float hi, const DivC& d, float& L, float& U) {
  const float s = BOX_SLOP * fmaxf(fabsf(lo), fabsf(hi));
  L = div_c(16.f * (lo - s), d);
  U = div_c(16.f * (hi + s), d);
  const float s2 = BOX_SLOP * fmaxf(fabsf(L), fabsf(U));
  L -= s2; U += s2;
}
// Cells (relative to the pixel's own cell j, j = int(x*scale)) that the taps of a pixel displaced by [dlo, dhi] pixels
// can touch.  Tap pixels are x0 = floor(x + f +- 2e-4) and x0 + 1, so x0 in [x + a, x + b] with a = floor(dlo) - 1,
// b = ceil(dhi) + 1; a pixel q interpolates cells int(q*scale) and that + 1, and x*scale is in [j, j+1).
__device__ __forceinline__ void cell_offsets(float dlo, float dhi, float scale, int& o_lo, int& o_hi) {
  const float a = floorf(dlo) - 1.f, b = ceilf(dhi) + 1.f;
  o_lo = (int)floorf(a * scale - 1e-4f);
  o_hi = 2 + (int)floorf(b * scale + 1e-4f);
}
__device__ __forceinline__ bool uniform(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

__global__ void __launch_bounds__(NT_C)
k_prepare_cert(const float* __restrict__ lo_fwd, const float* __restrict__ lo_bwd, float* __restrict__ flow_fwd,
               float* __restrict__ flow_bwd, uint8_t* __restrict__ mask_fwd, uint8_t* __restrict__ mask_bwd,
               int32_t* __restrict__ invalid, int N, Up8Geom g, int force) {
  __shared__ float4 cw[WH * WW];           // cells {fwd.x, fwd.y, bwd.x, bwd.y} under the tile + margin, replicate-padded
  // target cells of a DISPLACED window (large coherent motion), per direction: input of the row pass; the table the
  // column pass writes (two barriers later) takes its place
  __shared__ __attribute__((aligned(16))) float2 tw_store[2 * WH * WW];
  static_assert(sizeof(float2) * 2 * WH * WW >= sizeof(float4) * 2 * W0_H * W0_W, "the table fits where the window was");
  float2 (*tw)[WH * WW] = reinterpret_cast<float2 (*)[WH * WW]>(tw_store);
  float4 (*tab)[W0_H * W0_W] = reinterpret_cast<float4 (*)[W0_H * W0_W]>(tw_store);   // per own cell: normalised, inflated
                                           // box {Lx, Ux, Ly, Uy} of the opposite flow
  __shared__ float4 tmp[2][WH * W0_W];     // row pass {min x, max x, min y, max y} of the opposite flow
  __shared__ float part0[NWAVE_C][8];      // per-wave {min, max} x 4 channels of the cells under the tile
  __shared__ int odd_cells, wide_boxes[2]; // NaN / infinite cell seen (fminf would drop a NaN); boxes beyond the tolerance,
                                           // per 64-column half of the tile (cells on the seam count for both)
  __shared__ int uni[16];                  // wave 0's block-uniform results: certify, {ox, ex, oy, ey, shx, shy} x 2
  const int n = blockIdx.z, t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tx0 = blockIdx.x * TXC, ty0 = blockIdx.y * TYC;
  const int tx1 = min(tx0 + TXC, g.Wf) - 1, ty1 = min(ty0 + TYC, g.Hf) - 1;    // last pixel of the tile inside the frame
  const size_t hw = (size_t)g.h * g.w, HW = (size_t)g.Hf * g.Wf;
  const float* gf = lo_fwd + (size_t)n * 2 * hw;
  const float* gb = lo_bwd + (size_t)n * 2 * hw;
  const float INF = __builtin_huge_valf();
  // ---- the cell window and the range of the tile's own flows
  const int wx0 = axf(tx0, g.sx).i0, wy0 = axf(ty0, g.sy).i0;
  const int ucols = min(axf(tx1, g.sx).i0 + 1, g.w - 1) - wx0 + 1, urows = min(axf(ty1, g.sy).i0 + 1, g.h - 1) - wy0 + 1;
  if (t == 0) { odd_cells = 0; wide_boxes[0] = 0; wide_boxes[1] = 0; }
  {
    float mn[4] = {INF, INF, INF, INF}, mx[4] = {-INF, -INF, -INF, -INF};
    bool odd = false;
    for (int e = t; e < WH * WW; e += NT_C) {
      const int r = e / WW, c = e - r * WW;
      const unsigned o = min(max(wy0 - MARG + r, 0), g.h - 1) * g.w + min(max(wx0 - MARG + c, 0), g.w - 1);
      const float4 v = make_float4(gf[o], gf[hw + o], gb[o], gb[hw + o]);
      cw[e] = v;
      odd = odd || !(fabsf(v.x) < INF && fabsf(v.y) < INF && fabsf(v.z) < INF && fabsf(v.w) < INF);
      if ((unsigned)(r - MARG) < (unsigned)urows && (unsigned)(c - MARG) < (unsigned)ucols) {
        mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z); mn[3] = fminf(mn[3], v.w);
        mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z); mx[3] = fmaxf(mx[3], v.w);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = wave_min(mn[k]), b = wave_max(mx[k]);
      if (lane == 0) { part0[wave][2 * k] = a; part0[wave][2 * k + 1] = b; }
    }
    if (__any(odd) && lane == 0) odd_cells = 1;
  }
  __syncthreads();
  bool certify = mask_fwd != nullptr && force != 2;       // block-uniform throughout
  bool cert_h[2] = {false, false};                        // ... per 64-column half, once the tables are built
  const int c64 = axf(min(tx0 + TX, g.Wf - 1), g.sx).i0 - wx0;   // window column of the cell under the seam
  int shx[2] = {0, 0}, shy[2] = {0, 0};                   // displacement of the target window, in cells
  bool shifted = false;
  if (mask_fwd != nullptr) {                              // uniform: the barriers below are reached by every thread
    int ox[2] = {0, 0}, oy[2] = {0, 0}, ex[2] = {0, 0}, ey[2] = {0, 0};    // first / last cell offset per axis, per direction
    // The block-uniform arithmetic (range of the own flows -> cell offsets -> window displacement) is done by wave 0 alone
    // and handed over through LDS: five waves doing it redundantly was a third of the table cost.
    if (wave == 0) {
      certify = certify && odd_cells == 0;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        float lo[2], hi[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float l = part0[0][4 * d + 2 * k], h = part0[0][4 * d + 2 * k + 1];
#pragma unroll
          for (int wv = 1; wv < NWAVE_C; ++wv) { l = fminf(l, part0[wv][4 * d + 2 * k]); h = fmaxf(h, part0[wv][4 * d + 2 * k + 1]); }
          const float s = BOX_SLOP * fmaxf(fabsf(l), fabsf(h));
          lo[k] = 8.f * (l - s); hi[k] = 8.f * (h + s);      // own flow of direction d in pixels
          certify = certify && (hi[k] - lo[k] <= 64.f * MARG);                 // false for NaN too; bounds the casts below
        }
        if (certify) {
          cell_offsets(lo[0], hi[0], g.sx, ox[d], ex[d]);
          cell_offsets(lo[1], hi[1], g.sy, oy[d], ey[d]);
          // The neighbourhood of every cell under the tile has to lie inside a 20 x 22 cell window.  Around the tile itself
          // (the window that is already in LDS) that holds for own flows up to ~ +-24 px; a tile that moves further as a
          // whole (fast ego-motion) gets a second window displaced by (shx, shy) cells, centred on where it lands.
          const int span_x = MARG + (W0_W - ucols), span_y = MARG + (W0_H - urows);
          certify = ex[d] - ox[d] <= MARG + span_x && ey[d] - oy[d] <= MARG + span_y;
          shx[d] = min(max((ox[d] + ex[d]) / 2, ex[d] - span_x), ox[d] + MARG);
          shy[d] = min(max((oy[d] + ey[d]) / 2, ey[d] - span_y), oy[d] + MARG);
          if (ox[d] >= -MARG && ex[d] <= span_x && oy[d] >= -MARG && ey[d] <= span_y) shx[d] = shy[d] = 0;
        }
      }
      if (lane == 0) {
        uni[0] = certify ? 1 : 0;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          uni[1 + 6 * d] = ox[d]; uni[2 + 6 * d] = ex[d]; uni[3 + 6 * d] = oy[d]; uni[4 + 6 * d] = ey[d];
          uni[5 + 6 * d] = shx[d]; uni[6 + 6 * d] = shy[d];
        }
      }
    }
    __syncthreads();
    certify = uni[0] != 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      ox[d] = uni[1 + 6 * d]; ex[d] = uni[2 + 6 * d]; oy[d] = uni[3 + 6 * d]; ey[d] = uni[4 + 6 * d];
      shx[d] = uni[5 + 6 * d]; shy[d] = uni[6 + 6 * d];
    }
    certify = uniform(certify);
    shifted = uniform(certify && (shx[0] | shy[0] | shx[1] | shy[1]) != 0);
    if (shifted) {                                        // second, displaced window of the opposite map (one more round trip)
      for (int e = t; e < 2 * WH * WW; e += NT_C) {
        const int d = e >= WH * WW, q = e - d * WH * WW, r = q / WW, c = q - r * WW;
        const float* gm = d ? gf : gb;
        const unsigned o = min(max(wy0 - MARG + shy[d] + r, 0), g.h - 1) * g.w + min(max(wx0 - MARG + shx[d] + c, 0), g.w - 1);
        const float vx = gm[o], vy = gm[hw + o];
        tw[d][q] = make_float2(vx, vy);
        if (!(fabsf(vx) < INF && fabsf(vy) < INF)) odd_cells = 1;
      }
      __syncthreads();
      certify = uniform(certify && odd_cells == 0);
    }
    if (certify) {                                        // row pass: min / max over cells j + ox .. j + ex of each row
      for (int e = t; e < 2 * WH * W0_W; e += NT_C) {
        const int d = e >= WH * W0_W, q = e - d * WH * W0_W, r = q / W0_W, j = q - r * W0_W;
        // window row r holds cell row wy0 - MARG + shy + r; own cell row i needs rows MARG + i + oy - shy .. + ey - shy
        if (r >= MARG + oy[d] - shy[d] && r <= MARG + urows - 1 + ey[d] - shy[d] && j < ucols) {
          float4 m = make_float4(INF, -INF, INF, -INF);
          for (int k = ox[d] - shx[d]; k <= ex[d] - shx[d]; ++k) {
            float vx, vy;
            if (shifted) {
              const float2 v = tw[d][r * WW + MARG + j + k];
              vx = v.x; vy = v.y;
            } else {
              const float4 v = cw[r * WW + MARG + j + k];
              vx = d ? v.x : v.z; vy = d ? v.y : v.w;              // fwd pixels (d = 0) sample the bwd flow
            }
            m.x = fminf(m.x, vx); m.y = fmaxf(m.y, vx); m.z = fminf(m.z, vy); m.w = fmaxf(m.w, vy);
          }
          tmp[d][q] = m;
        }
      }
    }
    __syncthreads();
    if (certify) {                                        // column pass -> normalised boxes
      const int nd = urows * W0_W;                        // only the cell rows under the tile: one sweep of the block
      for (int e = t; e < 2 * nd; e += NT_C) {
        const int d = e >= nd, q = e - d * nd, i = q / W0_W, j = q - i * W0_W;
        if (j < ucols) {
          float4 m = make_float4(INF, -INF, INF, -INF);
          for (int k = oy[d] - shy[d]; k <= ey[d] - shy[d]; ++k) {
            const float4 v = tmp[d][(MARG + i + k) * W0_W + j];
            m.x = fminf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fminf(m.z, v.z); m.w = fmaxf(m.w, v.w);
          }
          float4 b;
          norm_box(m.x, m.y, g.wd, b.x, b.y);
          norm_box(m.z, m.w, g.ht, b.z, b.w);
          tab[d][q] = b;
          const float wxb = b.y - b.x, wyb = b.w - b.z;
          if (!(wxb * wxb + wyb * wyb <= g.a2)) {                            // cannot certify anything
            if (j <= c64) atomicAdd(&wide_boxes[0], 1);
            if (j >= c64) atomicAdd(&wide_boxes[1], 1);
          }
        }
      }
    }
    __syncthreads();
    // most boxes beyond the tolerance: the generic per-pixel fallback would dominate; the staged kernel is cheaper
    // The hand-over is decided per 64-column half (one column of two staged tiles): a motion boundary through one half
    // must not send the other half's pixels to the staged kernel, nor its own to the generic per-pixel path.
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int cols = hh ? ucols - c64 : min(c64 + 1, ucols);
      cert_h[hh] = uniform(certify && (force == 1 || 4 * wide_boxes[hh] <= 2 * urows * cols));
    }
    certify = cert_h[0] || cert_h[1];
    if (t < 4 && !cert_h[t & 1] && ty0 + (t >> 1) * TY < g.Hf && tx0 + (t & 1) * TX < g.Wf)   // the staged tiles of that half
      mask_fwd[(size_t)n * HW + (size_t)(ty0 + (t >> 1) * TY) * g.Wf + tx0 + (t & 1) * TX] = TILE_MARK;
  }
  const float a2c = g.a2 * (1.f - 2.f * BOX_SLOP);
  // ---- the pixels: a lane owns the columns x0, x0 + 1 (x0 even), each wave walks ROWS_C rows; separable up-sampling as in
  // the staged kernel.  Two columns per lane because of the masks: one byte per lane is a 64-B piece of a 128-B line per
  // wave instruction, and pieces of lines are what the memory side handles worst (tools/ubench/store_pattern.hip: the
  // same bytes stored as 4 B + 1 B per lane take 1.12 ms, as 8 B + 2 B per lane 0.76 ms); two bytes per lane are whole
  // lines, the flows leave as 8 B per lane, and a tile's tables serve twice the pixels.
  const int x0 = tx0 + 2 * lane;
  const bool act = x0 < g.Wf;                                  // Wf is even: both columns or none
  const bool cert_l = lane < 32 ? cert_h[0] : cert_h[1];       // this lane's half has its masks decided here
  AxF ax[2];
  int cxi[2];
  float c0x[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    ax[u] = axf(min(x0 + u, g.Wf - 1), g.sx);
    cxi[u] = ax[u].i0 - wx0;
    c0x[u] = div_c(2.f * (float)(x0 + u), g.wd) - 1.f;
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 t0[2] = {z4, z4}, t1[2] = {z4, z4}, bxf[2] = {z4, z4}, bxb[2] = {z4, z4};   // bxf: bwd-flow box for fwd pixels
  int have = -2, badf = 0, badb = 0;                           // bad counts are wave-uniform
  unsigned long long undecided = 0;                            // bit 4*rr + 2*u + d: inside the frame but not certified
  bool cellok[2][2] = {{false, false}, {false, false}};        // [column][direction] certified for the current run of rows
  bool allok = false;                                          // ... everything, in every lane of the wave
  float* of = flow_fwd + (size_t)n * 2 * HW + x0;
  float* ob = flow_bwd + (size_t)n * 2 * HW + x0;
  uint8_t* mf = mask_fwd + (size_t)n * HW + x0;
  uint8_t* mb = mask_bwd + (size_t)n * HW + x0;
  const int yw = ty0 + wave * ROWS_C;
  for (int rr = 0; rr < ROWS_C; ++rr) {
    const int y = yw + rr;
    if (y >= g.Hf) break;                                      // wave-uniform
    const AxF ay = axf(y, g.sy);
    if (ay.i0 != have) {
      const float cya = div_c(2.f * (float)y, g.ht) - 1.f, cyb = div_c(2.f * (float)min(y + 9, g.Hf - 1), g.ht) - 1.f;
      bool ok_all = true;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4* cell = cw + (MARG + ay.i0 - wy0) * WW + MARG + cxi[u];
        const AxF& a_ = ax[u];
        if (ay.i0 == have + 1) {
          t0[u] = t1[u];
        } else {
          const float4 a = cell[0], b = cell[1];
          t0[u] = make_float4(up8_row(a.x, b.x, a_.l0, a_.l1), up8_row(a.y, b.y, a_.l0, a_.l1),
                              up8_row(a.z, b.z, a_.l0, a_.l1), up8_row(a.w, b.w, a_.l0, a_.l1));
        }
        const float4 c = cell[WW], d = cell[WW + 1];
        t1[u] = make_float4(up8_row(c.x, d.x, a_.l0, a_.l1), up8_row(c.y, d.y, a_.l0, a_.l1),
                            up8_row(c.z, d.z, a_.l0, a_.l1), up8_row(c.w, d.w, a_.l0, a_.l1));
        if (cert_l) {
          const int ci = (ay.i0 - wy0) * W0_W + cxi[u];
          bxf[u] = tab[0][ci]; bxb[u] = tab[1][ci];
          // Certificate for ALL rows of this column that share the cell row (at most 9): between the two x-interpolated
          // source rows the column pass is a convex combination, so the column's own flow stays in [min(t0,t1), max(t0,t1)]
          // per channel; with the box of the opposite flow that bounds |fn + b'| for the whole run of rows, and the first /
          // last possible row bound the position.  Where it holds (smooth flow: almost everywhere) the rows below cost the
          // up-sampling and the stores only.
#pragma unroll
          for (int d2 = 0; d2 < 2; ++d2) {
            const float ax_ = d2 ? t0[u].z : t0[u].x, bx_ = d2 ? t1[u].z : t1[u].x;
            const float ay_ = d2 ? t0[u].w : t0[u].y, by_ = d2 ? t1[u].w : t1[u].y;
            float fLx, fUx, fLy, fUy;
            norm_box(fminf(ax_, bx_), fmaxf(ax_, bx_), g.wd, fLx, fUx);
            norm_box(fminf(ay_, by_), fmaxf(ay_, by_), g.ht, fLy, fUy);
            const float4 bx = d2 ? bxb[u] : bxf[u];
            const bool in_all = (c0x[u] + fLx > -1.f) && (c0x[u] + fUx < 1.f) && (cya + fLy > -1.f) && (cyb + fUy < 1.f);
            const float mx = fmaxf(fabsf(fLx + bx.x), fabsf(fUx + bx.y));
            const float my = fmaxf(fabsf(fLy + bx.z), fabsf(fUy + bx.w));
            cellok[u][d2] = in_all && (__builtin_fmaf(mx, mx, my * my) <= a2c);    // false for NaN
            ok_all = ok_all && cellok[u][d2];
          }
        }
      }
      if (certify) allok = __all(ok_all || !cert_l) != 0;      // the rows of a wave change cell rows together
      have = ay.i0;
    }
    const size_t q = (size_t)y * g.Wf;
    if (allok) {                                               // wave-uniform: flows and two constant mask words
      if (act) {
        st2(of + q, make_float2(8.f * up8_col(t0[0].x, t1[0].x, ay.l0, ay.l1), 8.f * up8_col(t0[1].x, t1[1].x, ay.l0, ay.l1)));
        st2(of + HW + q, make_float2(8.f * up8_col(t0[0].y, t1[0].y, ay.l0, ay.l1), 8.f * up8_col(t0[1].y, t1[1].y, ay.l0, ay.l1)));
        st2(ob + q, make_float2(8.f * up8_col(t0[0].z, t1[0].z, ay.l0, ay.l1), 8.f * up8_col(t0[1].z, t1[1].z, ay.l0, ay.l1)));
        st2(ob + HW + q, make_float2(8.f * up8_col(t0[0].w, t1[0].w, ay.l0, ay.l1), 8.f * up8_col(t0[1].w, t1[1].w, ay.l0, ay.l1)));
        if (cert_l) {
          st_u16(mf + q, 0x0101);
          st_u16(mb + q, 0x0101);
        }
      }
      continue;
    }
    // 16 * (interpolated value) = 2 * upflow8 value; both scalings are exact
    float fx16[2], fy16[2], bx16[2], by16[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fx16[u] = 16.f * up8_col(t0[u].x, t1[u].x, ay.l0, ay.l1); fy16[u] = 16.f * up8_col(t0[u].y, t1[u].y, ay.l0, ay.l1);
      bx16[u] = 16.f * up8_col(t0[u].z, t1[u].z, ay.l0, ay.l1); by16[u] = 16.f * up8_col(t0[u].w, t1[u].w, ay.l0, ay.l1);
    }
    if (act) {
      st2(of + q, make_float2(0.5f * fx16[0], 0.5f * fx16[1]));
      st2(of + HW + q, make_float2(0.5f * fy16[0], 0.5f * fy16[1]));
      st2(ob + q, make_float2(0.5f * bx16[0], 0.5f * bx16[1]));
      st2(ob + HW + q, make_float2(0.5f * by16[0], 0.5f * by16[1]));
    }
    if (!certify) continue;                                    // block-uniform; below, a lane of a handed-over half
                                                               // (!cert_l) takes part in the ballots and contributes nothing
    bool ok[2][2] = {{true, true}, {true, true}};              // [column][direction]
    const float c0y = div_c(2.f * (float)y, g.ht) - 1.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if ((cellok[u][0] && cellok[u][1]) || !cert_l) continue; // the per-pixel certificate, where the run of rows failed
      const float fn[2][2] = {{div_c(fx16[u], g.wd), div_c(fy16[u], g.ht)}, {div_c(bx16[u], g.wd), div_c(by16[u], g.ht)}};
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        if (cellok[u][d]) continue;
        const float4 bx = d ? bxb[u] : bxf[u];
        const float c1x = c0x[u] + fn[d][0], c1y = c0y + fn[d][1];
        const bool inside = (fabsf(c1x) < 1.f) && (fabsf(c1y) < 1.f);
        const float mx = fmaxf(fabsf(fn[d][0] + bx.x), fabsf(fn[d][0] + bx.y));
        const float my = fmaxf(fabsf(fn[d][1] + bx.z), fabsf(fn[d][1] + bx.w));
        const bool sure = __builtin_fmaf(mx, mx, my * my) <= a2c;
        ok[u][d] = inside && sure;
        if (inside && !sure) undecided |= 1ull << (4 * rr + 2 * u + d);   // settled exactly after the loop
      }
    }
    const bool mine = act && cert_l;
    if (mine) {
      st_u16(mf + q, (uint16_t)((ok[0][0] ? 1 : 0) | (ok[1][0] ? 0x100 : 0)));
      st_u16(mb + q, (uint16_t)((ok[0][1] ? 1 : 0) | (ok[1][1] ? 0x100 : 0)));
    }
    badf += __popcll(__ballot(mine && !ok[0][0])) + __popcll(__ballot(mine && !ok[1][0]));
    badb += __popcll(__ballot(mine && !ok[0][1])) + __popcll(__ballot(mine && !ok[1][1]));
  }
  // Pixels the certificate could not settle (rare in a tile that was not handed over): the exact generic evaluation, kept
  // out of the loop above so that the call does not dictate its register allocation.  (Handing such tiles to the staged
  // kernel instead makes this kernel leaner but re-does whole tiles for a few pixels: 0.96 vs 0.88 ms.)
  if (!act) undecided = 0;
  while (__any(undecided != 0)) {
    bool fixed_f = false, fixed_b = false;
    if (undecided) {
      const int bit = __builtin_ctzll(undecided);
      undecided &= undecided - 1;
      const int d = bit & 1, u = (bit >> 1) & 1, y = yw + (bit >> 2);
      if (fbc_generic(d ? gb : gf, d ? gf : gb, g.h, g.w, y, x0 + u, g.alpha1, g.a2)) {
        (d ? mb : mf)[(size_t)y * g.Wf + u] = 1;
        fixed_f = d == 0; fixed_b = d == 1;
      }
    }
    badf -= __popcll(__ballot(fixed_f));
    badb -= __popcll(__ballot(fixed_b));
  }
  if (invalid && certify && lane == 0) {
    if (badf) atomicAdd(&invalid[n], badf);
    if (badb) atomicAdd(&invalid[N + n], badb);
  }
}

// ---- second launch of the nframe = 2 preparation: the masks of the tiles the certified kernel could not settle -------
// (flows that vary by more than the certificate tolerates inside a tile: motion boundaries, occlusions).  Taps come from
// LDS; the flows themselves were already written by k_prepare_cert.
__global__ void __launch_bounds__(NT)
k_prepare_staged(const float* __restrict__ lo_fwd, const float* __restrict__ lo_bwd, uint8_t* __restrict__ mask_fwd,
                 uint8_t* __restrict__ mask_bwd, int32_t* __restrict__ invalid, int N, Up8Geom g, int tiles_x,
                 int tiles_y) {
  __shared__ float2 nf[RH * RW];           // normalised forward flow of the region
  __shared__ float2 nb[RH * RW];           // normalised backward flow
  __shared__ float4 lw[LW_H * LW_W];       // low-res cells {fwd.x, fwd.y, bwd.x, bwd.y}, replicate-padded
  // Few tiles are marked: a small grid scans the tile list - block b owns tiles b, b + G, b + 2G, ... (marked tiles spread
  // evenly over the blocks), checks NT of them per step (one mark byte per thread: one memory round trip per NT tiles, not
  // per tile) and works through the marked ones.
  __shared__ unsigned long long marks[NWAVE];
  const int t = threadIdx.x;
  const size_t hw = (size_t)g.h * g.w, HW = (size_t)g.Hf * g.Wf;
  const int ntiles = tiles_x * tiles_y * N, per_n = tiles_x * tiles_y;
  const int G = gridDim.x;
  for (int base = 0; blockIdx.x + (long)G * base < ntiles; base += NT) {
  {
    const long tl = blockIdx.x + (long)G * (base + t);
    bool marked = false;
    if (tl < ntiles) {
      const int n_ = tl / per_n, rem_ = tl - n_ * per_n, by_ = rem_ / tiles_x, bx_ = rem_ - by_ * tiles_x;
      marked = mask_fwd[(size_t)n_ * HW + (size_t)(by_ * TY) * g.Wf + bx_ * TX] == TILE_MARK;
    }
    const unsigned long long m = __ballot(marked);
    if ((t & 63) == 0) marks[t >> 6] = m;
  }
  __syncthreads();                                            // also orders these reads before any mask write below
  for (int wv = 0; wv < NWAVE; ++wv)
  for (unsigned long long todo = marks[wv]; todo; todo &= todo - 1) {
  const int tile = blockIdx.x + G * (base + 64 * wv + __builtin_ctzll(todo));
  const int n = tile / per_n, rem = tile - n * per_n, by = rem / tiles_x, bx = rem - by * tiles_x;
  const int tx0 = bx * TX, ty0 = by * TY;
  const int rx0 = tx0 - HL, ry0 = ty0 - HL;
  const float* gf = lo_fwd + (size_t)n * 2 * hw;
  const float* gb = lo_bwd + (size_t)n * 2 * hw;
  // low-res window: the cells of the region's (clamped) pixels plus LMARG cells all around (for taps beyond the region)
  const int wx0 = axf(max(rx0, 0), g.sx).i0 - LMARG, wy0 = axf(max(ry0, 0), g.sy).i0 - LMARG;
  for (int e = t; e < LW_H * LW_W; e += NT) {
    const int r = e / LW_W, c = e - r * LW_W;
    const unsigned o = min(max(wy0 + r, 0), g.h - 1) * g.w + min(max(wx0 + c, 0), g.w - 1);
    lw[e] = make_float4(gf[o], gf[hw + o], gb[o], gb[hw + o]);
  }
  __syncthreads();
  {
    int rc, r0;
    stage_slot(t, rc, r0);
    const int x = min(max(rx0 + rc, 0), g.Wf - 1);
    const AxF ax = axf(x, g.sx);
    const int cx = ax.i0 - wx0;
    float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;    // x-interpolated source rows i0y, i0y+1
    int have = -2;
#pragma unroll 2
    for (int r = r0; r < r0 + ROWS_PER_GROUP; ++r) {
      const int y = min(max(ry0 + r, 0), g.Hf - 1);
      const AxF ay = axf(y, g.sy);
      if (ay.i0 != have) {
        const float4* cell = lw + (ay.i0 - wy0) * LW_W + cx;
        if (ay.i0 == have + 1) {
          t0 = t1;
        } else {
          const float4 a = cell[0], b = cell[1];
          t0 = make_float4(up8_row(a.x, b.x, ax.l0, ax.l1), up8_row(a.y, b.y, ax.l0, ax.l1),
                           up8_row(a.z, b.z, ax.l0, ax.l1), up8_row(a.w, b.w, ax.l0, ax.l1));
        }
        const float4 c = cell[LW_W], d = cell[LW_W + 1];
        t1 = make_float4(up8_row(c.x, d.x, ax.l0, ax.l1), up8_row(c.y, d.y, ax.l0, ax.l1),
                         up8_row(c.z, d.z, ax.l0, ax.l1), up8_row(c.w, d.w, ax.l0, ax.l1));
        have = ay.i0;
      }
      // 16 * (interpolated value) = 2 * upflow8 value; both scalings are exact
      const float fx16 = 16.f * up8_col(t0.x, t1.x, ay.l0, ay.l1), fy16 = 16.f * up8_col(t0.y, t1.y, ay.l0, ay.l1);
      const float bx16 = 16.f * up8_col(t0.z, t1.z, ay.l0, ay.l1), by16 = 16.f * up8_col(t0.w, t1.w, ay.l0, ay.l1);
      nf[r * RW + rc] = make_float2(div_c(fx16, g.wd), div_c(fy16, g.ht));
      nb[r * RW + rc] = make_float2(div_c(bx16, g.wd), div_c(by16, g.ht));
    }
  }
  __syncthreads();
  const int wave = t >> 6, lane = t & 63;
  const int x = tx0 + lane;
  const float c0x = div_c(2.f * (float)x, g.wd) - 1.f;
  int badf = 0, badb = 0;
#pragma unroll 2
  for (int rr = 0; rr < TY / NWAVE; ++rr) {
    const int ty = wave * (TY / NWAVE) + rr, y = ty0 + ty;
    if (y >= g.Hf) break;                                   // wave-uniform
    bool okf = true, okb = true;
    if (x < g.Wf) {
      const int o = (ty + HL) * RW + (lane + HL);
      const float c0y = div_c(2.f * (float)y, g.ht) - 1.f;
      bool slowf = false, slowb = false;
      okf = fbc_staged(nf[o], nb, g, c0x, c0y, rx0, ry0, slowf);
      okb = fbc_staged(nb[o], nf, g, c0x, c0y, rx0, ry0, slowb);
      // taps beyond the staged region: from the cells in LDS (up to ~32 px away), beyond those from global memory
      if (__builtin_expect(slowf, 0)) {
        bool far = false;
        okf = fbc_cells(nf[o], g, c0x, c0y, wx0, wy0, LW_W, LW_H,
                        [&](int ry, int rx) { const float4 v = lw[ry * LW_W + rx]; return make_float2(v.z, v.w); }, far);
        if (far) okf = fbc_generic(gf, gb, g.h, g.w, y, x, g.alpha1, g.a2);
      }
      if (__builtin_expect(slowb, 0)) {
        bool far = false;
        okb = fbc_cells(nb[o], g, c0x, c0y, wx0, wy0, LW_W, LW_H,
                        [&](int ry, int rx) { const float4 v = lw[ry * LW_W + rx]; return make_float2(v.x, v.y); }, far);
        if (far) okb = fbc_generic(gb, gf, g.h, g.w, y, x, g.alpha1, g.a2);
      }
      const size_t q = (size_t)n * HW + (size_t)y * g.Wf + x;
      mask_fwd[q] = okf ? 1 : 0;
      mask_bwd[q] = okb ? 1 : 0;
    }
    badf += __popcll(__ballot(!okf));
    badb += __popcll(__ballot(!okb));
  }
  if (invalid && lane == 0) {
    if (badf) atomicAdd(&invalid[n], badf);
    if (badb) atomicAdd(&invalid[N + n], badb);
  }
  __syncthreads();                                          // the LDS images are reused by the next tile
  }
  __syncthreads();                                            // marks[] is rewritten by the next step
  }
}

// ---- multi-frame (F-1 > 1): concat_flow over lazily up-sampled maps, both directions in one launch -----------------
// One chain per pixel; every step takes the four bilinear taps of U_t (the x8 up-sampling of flow t) at the chain's
// current position straight from the flow's 1/8-resolution cells, which the block keeps in LDS for all flows of the
// chain (tile + a margin that covers +-24 pixels of drift; positions beyond it read the cells from global memory).
// No barrier after the load: chains are independent.  (A variant that staged U_t per step over a region following the
// block's mean drift had the same instruction count and two barriers per step: 3.9 ms against 2.x at N = 64, F = 6.)
constexpr int CH_MAXF = 6;                               // flows of a chain whose cell windows are kept in LDS
constexpr int CMARG = 4;                                 // cells of margin around the cells under the tile
constexpr int CWW = TX / 8 + 4 + 2 * CMARG, CWH = TYC / 8 + 4 + 2 * CMARG;   // 20 x 22 cells per flow

struct ChainWin {
  const float2* cells;     // this flow's LDS window
  const float* gx;         // ... and its global planes, for positions outside the window
  const float* gy;
  int wx0, wy0, h, w;
};
// U_t from the 2 x 2 cells at `cell` of an LDS window
__device__ __forceinline__ float2 up8_cell4(const float2* cell, const AxF& ay, const AxF& ax) {
  const float2 a = cell[0], b = cell[1], c = cell[CWW], d = cell[CWW + 1];
  return make_float2(8.f * up8_col(up8_row(a.x, b.x, ax.l0, ax.l1), up8_row(c.x, d.x, ax.l0, ax.l1), ay.l0, ay.l1),
                     8.f * up8_col(up8_row(a.y, b.y, ax.l0, ax.l1), up8_row(c.y, d.y, ax.l0, ax.l1), ay.l0, ay.l1));
}
// U_t at pixel (ty, tx) inside the frame: separable x8 up-sampling of the four cells around it
__device__ __forceinline__ float2 chain_tap(const ChainWin& cwn, const AxF& ay, const AxF& ax) {
  const int ry = ay.i0 - cwn.wy0, rx = ax.i0 - cwn.wx0;
  float2 a, b, c, d;
  if (__builtin_expect((unsigned)ry < (unsigned)(CWH - 1) && (unsigned)rx < (unsigned)(CWW - 1), 1)) {
    const float2* cell = cwn.cells + ry * CWW + rx;
    a = cell[0]; b = cell[1]; c = cell[CWW]; d = cell[CWW + 1];
  } else {
    const int x1 = min(ax.i0 + 1, cwn.w - 1), y1 = min(ay.i0 + 1, cwn.h - 1);
    const unsigned o00 = ay.i0 * cwn.w + ax.i0, o01 = ay.i0 * cwn.w + x1, o10 = y1 * cwn.w + ax.i0, o11 = y1 * cwn.w + x1;
    a = make_float2(cwn.gx[o00], cwn.gy[o00]); b = make_float2(cwn.gx[o01], cwn.gy[o01]);
    c = make_float2(cwn.gx[o10], cwn.gy[o10]); d = make_float2(cwn.gx[o11], cwn.gy[o11]);
  }
  return make_float2(8.f * up8_col(up8_row(a.x, b.x, ax.l0, ax.l1), up8_row(c.x, d.x, ax.l0, ax.l1), ay.l0, ay.l1),
                     8.f * up8_col(up8_row(a.y, b.y, ax.l0, ax.l1), up8_row(c.y, d.y, ax.l0, ax.l1), ay.l0, ay.l1));
}

// (6 waves per SIMD = 80 VGPRs: the block boxes must not cost the chain an occupancy step - 83 VGPRs without the hint,
// measured 2.2 -> 2.5 ms)
__global__ void __launch_bounds__(NT_C) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_chain_up8(const float* __restrict__ lo_fwd, const float* __restrict__ lo_bwd, float* __restrict__ flow_fwd,
            float* __restrict__ flow_bwd, float4* __restrict__ bbox, int N, int F1, Up8Geom g) {
  __shared__ float2 lwc[CH_MAXF][CWH * CWW];   // low-res cells of every flow under the tile + margin, replicate-padded
  const int n = blockIdx.z >> 1, dir = blockIdx.z & 1, t = threadIdx.x;
  const size_t hw = (size_t)g.h * g.w, HW = (size_t)g.Hf * g.Wf;
  const float* lo = (dir ? lo_bwd : lo_fwd) + (size_t)n * F1 * 2 * hw;          // [F1][2][h][w]
  float* out = (dir ? flow_bwd : flow_fwd) + (size_t)n * 2 * HW;
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TYC;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int wx0 = axf(tx0, g.sx).i0 - CMARG, wy0 = axf(ty0, g.sy).i0 - CMARG;
  for (int e = t; e < F1 * CWH * CWW; e += NT_C) {
    const int f = e / (CWH * CWW), q = e - f * CWH * CWW, r = q / CWW, c = q - r * CWW;
    const unsigned o = min(max(wy0 + r, 0), g.h - 1) * g.w + min(max(wx0 + c, 0), g.w - 1);
    lwc[f][q] = make_float2(lo[(size_t)f * 2 * hw + o], lo[(size_t)f * 2 * hw + hw + o]);
  }
  __syncthreads();
  const int x = tx0 + lane;
  if (x >= g.Wf) return;                 // Wf = 8 w: whole groups of eight lanes leave together
  const float x0f = (float)x;
  // by-product for the certified mask kernel: {min x, max x, min y, max y} of the composed flow over every 8 x 8 block
  // (tile origins and the 16 rows of a wave are multiples of 8; Hf = 8 h); +-inf when a block holds a non-finite value
  const float INF = __builtin_huge_valf();
  float4 bb = make_float4(INF, -INF, INF, -INF);
  bool odd = false;
  float4* bbn = bbox ? bbox + (size_t)blockIdx.z * (g.Hf / 8) * (g.Wf / 8) : nullptr;
#pragma unroll 2
  for (int i = 0; i < ROWS_C; ++i) {
    const int y = ty0 + wave * ROWS_C + i;
    if (y >= g.Hf) break;
    float cx = x0f, cy = (float)y;
    for (int f = 0; f < F1; ++f) {
      const ChainWin cwn{lwc[f], lo + (size_t)f * 2 * hw, lo + (size_t)f * 2 * hw + hw, wx0, wy0, g.h, g.w};
      const float nx = div_c(2.f * cx, g.wd) - 1.f, ny = div_c(2.f * cy, g.ht) - 1.f;   // normalize_coord
      const float ix = (nx + 1.f) * g.hx, iy = (ny + 1.f) * g.hy;
      const float xw = floorf(ix), yn = floorf(iy);
      const float w = ix - xw, e = 1.f - w, nn = iy - yn, s = 1.f - nn;
      const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
      float fx, fy;
      if (__builtin_expect(xw >= 0.f && xw < (float)(g.Wf - 1) && yn >= 0.f && yn < (float)(g.Hf - 1), 1)) {
        // all four taps inside the frame (the usual case): no padding logic, one window test for the 2 x 2 block of taps
        const int px0 = (int)xw, py0 = (int)yn;
        const AxF ax0 = axf(px0, g.sx), ax1 = axf(px0 + 1, g.sx), ay0 = axf(py0, g.sy), ay1 = axf(py0 + 1, g.sy);
        const int rx = ax0.i0 - wx0, ry = ay0.i0 - wy0;
        float2 a, b, c, d;
        if (__builtin_expect((unsigned)rx < (unsigned)(CWW - 2) && (unsigned)ry < (unsigned)(CWH - 2), 1)) {
          // ax1.i0 - ax0.i0 and ay1.i0 - ay0.i0 are 0 or 1: every cell below lies inside the window
          const float2* c00 = cwn.cells + ry * CWW + rx;
          const float2* c01 = c00 + (ax1.i0 - ax0.i0);
          const int dy = (ay1.i0 - ay0.i0) * CWW;
          a = up8_cell4(c00, ay0, ax0); b = up8_cell4(c01, ay0, ax1);
          c = up8_cell4(c00 + dy, ay1, ax0); d = up8_cell4(c01 + dy, ay1, ax1);
        } else {
          a = chain_tap(cwn, ay0, ax0); b = chain_tap(cwn, ay0, ax1); c = chain_tap(cwn, ay1, ax0); d = chain_tap(cwn, ay1, ax1);
        }
        fx = __builtin_fmaf(d.x, se, __builtin_fmaf(c.x, sw, __builtin_fmaf(b.x, ne, a.x * nw)));
        fy = __builtin_fmaf(d.y, se, __builtin_fmaf(c.y, sw, __builtin_fmaf(b.y, ne, a.y * nw)));
      } else {
        // float -> int keeping far-away (or NaN) coordinates out of range; taps outside the frame contribute zero
        const int px0 = (xw > -2.f && xw < (float)(g.Wf + 1)) ? (int)xw : -2;
        const int py0 = (yn > -2.f && yn < (float)(g.Hf + 1)) ? (int)yn : -2;
        const bool vx0 = px0 >= 0 && px0 < g.Wf, vx1 = px0 + 1 >= 0 && px0 + 1 < g.Wf;
        const bool vy0 = py0 >= 0 && py0 < g.Hf, vy1 = py0 + 1 >= 0 && py0 + 1 < g.Hf;
        const AxF ax0 = axf(min(max(px0, 0), g.Wf - 1), g.sx), ax1 = axf(min(max(px0 + 1, 0), g.Wf - 1), g.sx);
        const AxF ay0 = axf(min(max(py0, 0), g.Hf - 1), g.sy), ay1 = axf(min(max(py0 + 1, 0), g.Hf - 1), g.sy);
        float2 a = chain_tap(cwn, ay0, ax0), b = chain_tap(cwn, ay0, ax1), c = chain_tap(cwn, ay1, ax0),
               d = chain_tap(cwn, ay1, ax1);
        if (!(vy0 && vx0)) a = make_float2(0.f, 0.f);
        if (!(vy0 && vx1)) b = make_float2(0.f, 0.f);
        if (!(vy1 && vx0)) c = make_float2(0.f, 0.f);
        if (!(vy1 && vx1)) d = make_float2(0.f, 0.f);
        fx = __builtin_fmaf(d.x, se, __builtin_fmaf(c.x, sw, __builtin_fmaf(b.x, ne, a.x * nw)));
        fy = __builtin_fmaf(d.y, se, __builtin_fmaf(c.y, sw, __builtin_fmaf(b.y, ne, a.y * nw)));
      }
      cx = cx + fx; cy = cy + fy;
    }
    const size_t q = (size_t)y * g.Wf + x;
    const float ox = cx - x0f, oy = cy - (float)y;
    out[q] = ox;
    out[HW + q] = oy;
    if (bbn) {
      bb.x = fminf(bb.x, ox); bb.y = fmaxf(bb.y, ox); bb.z = fminf(bb.z, oy); bb.w = fmaxf(bb.w, oy);
      odd = odd || !(fabsf(ox) < INF && fabsf(oy) < INF);
      if ((i & 7) == 7) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
          bb.x = fminf(bb.x, __shfl_xor(bb.x, o, 64)); bb.y = fmaxf(bb.y, __shfl_xor(bb.y, o, 64));
          bb.z = fminf(bb.z, __shfl_xor(bb.z, o, 64)); bb.w = fmaxf(bb.w, __shfl_xor(bb.w, o, 64));
        }
        const unsigned long long oddm = __ballot(odd);
        if ((oddm >> (lane & ~7)) & 0xFFull) bb = make_float4(-INF, INF, -INF, INF);
        if ((lane & 7) == 0) bbn[(size_t)(y >> 3) * (g.Wf / 8) + (x >> 3)] = bb;
        bb = make_float4(INF, -INF, INF, -INF);
        odd = false;
      }
    }
  }
}

constexpr int CH_PX = TY / NWAVE;        // pixels per thread of the mask kernel: lane = x, CH_PX rows per wave

// k_fbc_pair: both consistency masks of dense composed flows in one launch (direction in blockIdx.z & 1).  The block
// reads its tile of the own flow (coalesced), displaces the region by the tile's mean flow, stages the normalised
// opposite flow over that region (coalesced row reads) and takes every tap from LDS.
__device__ __attribute__((noinline)) bool fbc_dense_generic(const float* f, const float* o, int Hf, int Wf, int y, int x,
                                                            float alpha1, float a2) {
  const size_t HW = (size_t)Hf * Wf, r = (size_t)y * Wf + x;
  float cx, cy;
  return fbc_pixel(f[r], f[HW + r], FetchDense{o, o + HW, Wf}, Hf, Wf, y, x, alpha1, a2, 0, cx, cy);
}

__device__ __forceinline__ void fbc_pair_tile(const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd,
                                              uint8_t* __restrict__ mask_fwd, uint8_t* __restrict__ mask_bwd,
                                              int32_t* __restrict__ invalid, int N, const Up8Geom& g, int bx, int by, int n,
                                              int dir, float2* __restrict__ no, float (*red)[3]) {
  const int t = threadIdx.x;
  const size_t HW = (size_t)g.Hf * g.Wf;
  const float* f = (dir ? flow_bwd : flow_fwd) + (size_t)n * 2 * HW;
  const float* o = (dir ? flow_fwd : flow_bwd) + (size_t)n * 2 * HW;
  uint8_t* mask = (dir ? mask_bwd : mask_fwd) + (size_t)n * HW;
  const int tx0 = bx * TX, ty0 = by * TY;
  const int wave = t >> 6, lane = t & 63;
  const int x = tx0 + lane;
  float2 fn[CH_PX];
  float sdx = 0.f, sdy = 0.f, cnt = 0.f;
#pragma unroll
  for (int i = 0; i < CH_PX; ++i) {
    const int y = ty0 + wave * CH_PX + i;
    fn[i] = make_float2(0.f, 0.f);
    if (y < g.Hf && x < g.Wf) {
      const size_t q = (size_t)y * g.Wf + x;
      const float fx = f[q], fy = f[HW + q];
      fn[i] = make_float2(div_c(2.f * fx, g.wd), div_c(2.f * fy, g.ht));       // normalize_flow
      sdx += fx; sdy += fy; cnt += 1.f;
    }
  }
  sdx = wave_sum(sdx); sdy = wave_sum(sdy); cnt = wave_sum(cnt);
  if (lane == 0) { red[wave][0] = sdx; red[wave][1] = sdy; red[wave][2] = cnt; }
  __syncthreads();
  float ax_ = 0.f, ay_ = 0.f, ac = 0.f;
#pragma unroll
  for (int k = 0; k < NWAVE; ++k) { ax_ += red[k][0]; ay_ += red[k][1]; ac += red[k][2]; }
  const float mx = ac > 0.f ? ax_ / ac : 0.f, my = ac > 0.f ? ay_ / ac : 0.f;
  const int shx = (int)rintf(fminf(fmaxf(mx, -4096.f), 4096.f)), shy = (int)rintf(fminf(fmaxf(my, -4096.f), 4096.f));
  const int rx0 = min(max(tx0 - HL + shx, -HL), g.Wf - 1 - HL), ry0 = min(max(ty0 - HL + shy, -HL), g.Hf - 1 - HL);
  for (int e = t; e < RH * RW; e += NT) {
    const int r = e / RW, c = e - r * RW;
    const size_t q = (size_t)min(max(ry0 + r, 0), g.Hf - 1) * g.Wf + min(max(rx0 + c, 0), g.Wf - 1);
    no[e] = make_float2(div_c(2.f * o[q], g.wd), div_c(2.f * o[HW + q], g.ht));
  }
  __syncthreads();
  const float c0x = div_c(2.f * (float)x, g.wd) - 1.f;
  int bad = 0;
#pragma unroll
  for (int i = 0; i < CH_PX; ++i) {
    const int y = ty0 + wave * CH_PX + i;
    if (y >= g.Hf) break;
    bool ok = true;
    if (x < g.Wf) {
      const float c0y = div_c(2.f * (float)y, g.ht) - 1.f;
      bool slow = false;
      ok = fbc_staged(fn[i], no, g, c0x, c0y, rx0, ry0, slow);
      if (__builtin_expect(slow, 0)) ok = fbc_dense_generic(f, o, g.Hf, g.Wf, y, x, g.alpha1, g.a2);
      mask[(size_t)y * g.Wf + x] = ok ? 1 : 0;
    }
    bad += __popcll(__ballot(!ok));
  }
  if (invalid && lane == 0 && bad) atomicAdd(&invalid[dir * N + n], bad);
}

__global__ void __launch_bounds__(NT)
k_fbc_pair(const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd, uint8_t* __restrict__ mask_fwd,
           uint8_t* __restrict__ mask_bwd, int32_t* __restrict__ invalid, int N, Up8Geom g) {
  __shared__ float2 no[RH * RW];           // normalised opposite flow over the region
  __shared__ float red[NWAVE][3];
  fbc_pair_tile(flow_fwd, flow_bwd, mask_fwd, mask_bwd, invalid, N, g, blockIdx.x, blockIdx.y, blockIdx.z >> 1, blockIdx.z & 1, no, red);
}

// Second launch behind k_fbc_cert_dense: only the tiles it marked.  Few are: a small grid scans the list of
// (direction, sample, tile) entries as k_prepare_staged does - block b owns entries b, b + G, ..., checks NT of them per
// step (one mark byte per thread) and works through the marked ones.
__global__ void __launch_bounds__(NT)
k_fbc_pair_marked(const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd, uint8_t* __restrict__ mask_fwd,
                  uint8_t* __restrict__ mask_bwd, int32_t* __restrict__ invalid, int N, Up8Geom g, int tiles_x, int tiles_y) {
  __shared__ float2 no[RH * RW];
  __shared__ float red[NWAVE][3];
  __shared__ unsigned long long marks[NWAVE];
  const int t = threadIdx.x, G = gridDim.x;
  const size_t HW = (size_t)g.Hf * g.Wf;
  const int per_n = tiles_x * tiles_y, nent = per_n * 2 * N;
  for (int base = 0; blockIdx.x + (long)G * base < nent; base += NT) {
    const long e = blockIdx.x + (long)G * (base + t);
    bool marked = false;
    if (e < nent) {
      const int z = e / per_n, rem = e - z * per_n, by = rem / tiles_x, bx = rem - by * tiles_x;
      const uint8_t* m = ((z & 1) ? mask_bwd : mask_fwd) + (size_t)(z >> 1) * HW;
      marked = m[(size_t)(by * TY) * g.Wf + bx * TX] == TILE_MARK;
    }
    const unsigned long long mm = __ballot(marked);
    if ((t & 63) == 0) marks[t >> 6] = mm;
    __syncthreads();                                          // also orders these reads before any mask write below
    for (int wv = 0; wv < NWAVE; ++wv)
      for (unsigned long long todo = marks[wv]; todo; todo &= todo - 1) {
        const int e2 = blockIdx.x + G * (base + 64 * wv + __builtin_ctzll(todo));
        const int z = e2 / per_n, rem = e2 - z * per_n, by = rem / tiles_x, bx = rem - by * tiles_x;
        fbc_pair_tile(flow_fwd, flow_bwd, mask_fwd, mask_bwd, invalid, N, g, bx, by, z >> 1, z & 1, no, red);
        __syncthreads();                                      // the tile is done with `no` and `red`
      }
    __syncthreads();                                          // ... and everybody with `marks`
  }
}

// ---- certified masks of DENSE composed flows (nframe > 2), both directions per block ------------------------------------
// The certificate of k_prepare_cert, with the 8 x 8-pixel bounding boxes k_chain_up8 leaves behind in place of the
// low-res cells: the taps of a pixel of block j displaced by a flow in [lo, hi] (the range of the tile's own flow) lie
// in blocks j + floor((floor(lo) - 1) / 8) .. j + floor((7 + ceil(hi) + 1) / 8); the union of their boxes bounds b'.
// A certified pixel costs its own two flow values (coalesced), the exact normalisation and ~12 instructions per
// direction; pixels the boxes cannot settle are evaluated exactly; tiles where they cannot work are marked in BOTH masks
// for k_fbc_pair (marked_only).  Bit-identical to k_fbc_pair either way.
constexpr int TBX = TX / 8, TBY = TYC / 8;               // 8 x 10 blocks per tile
constexpr int MB = 4;                                    // blocks of margin: own-flow RANGE inside a tile up to ~32 px
constexpr int WBW = TBX + 2 * MB, WBH = TBY + 2 * MB;    // block window: 16 x 18

__device__ __forceinline__ void block_offsets(float dlo, float dhi, int& o_lo, int& o_hi) {
  const float a = floorf(dlo) - 1.f, b = ceilf(dhi) + 1.f;
  o_lo = (int)floorf(a * 0.125f);
  o_hi = (int)floorf((7.f + b) * 0.125f);
}
// [lo, hi] of raw flow values (pixels) -> box of the normalised flow 2 f / (size - 1), inflated for rounding
__device__ __forceinline__ void norm_box_dense(float lo, float hi, const DivC& d, float& L, float& U) {
  const float s = BOX_SLOP * fmaxf(fabsf(lo), fabsf(hi));
  L = div_c(2.f * (lo - s), d);
  U = div_c(2.f * (hi + s), d);
  const float s2 = BOX_SLOP * fmaxf(fabsf(L), fabsf(U));
  L -= s2; U += s2;
}

__global__ void __launch_bounds__(NT_C)
k_fbc_cert_dense(const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd, const float4* __restrict__ bbox,
                 uint8_t* __restrict__ mask_fwd, uint8_t* __restrict__ mask_bwd, int32_t* __restrict__ invalid, int N,
                 Up8Geom g, int force) {
  __shared__ float4 ow[2][WBH * WBW];      // opposite-direction boxes around where the tile lands, per direction
  __shared__ float4 tmp[2][WBH * TBX];     // row pass
  __shared__ float4 tab[2][TBY * TBX];     // per own block: normalised, inflated box {Lx, Ux, Ly, Uy} of the opposite flow
  __shared__ float4 own[2][TBY * TBX];     // the tile's own boxes (raw pixels), per direction
  __shared__ float part0[2][4];            // {min x, max x, min y, max y} of the tile's own flow, per direction
  __shared__ int wide_boxes;
  const int n = blockIdx.z, t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TYC;
  const int Wb = g.Wf / 8, Hb = g.Hf / 8, bx0 = tx0 / 8, by0 = ty0 / 8;
  const int ucols = min(TBX, Wb - bx0), urows = min(TBY, Hb - by0);
  const size_t HW = (size_t)g.Hf * g.Wf, HWb = (size_t)Hb * Wb;
  const float INF = __builtin_huge_valf();
  const float4* bbd[2] = {bbox + (size_t)(2 * n) * HWb, bbox + (size_t)(2 * n + 1) * HWb};   // boxes of fwd, bwd
  if (t == 0) wide_boxes = 0;
  // ---- range of the tile's own flows: waves 0, 1 take direction 0, waves 2, 3 direction 1 (80 blocks each)
  if (wave < 4) {
    const int d = wave >> 1, e = t - d * 128;
    float4 m = make_float4(INF, -INF, INF, -INF);
    if (e < TBY * TBX) {
      const int r = e / TBX, c = e - r * TBX;
      if (r < urows && c < ucols) m = bbd[d][(size_t)(by0 + r) * Wb + bx0 + c];
      own[d][e] = m;
    }
    m.x = wave_min(m.x); m.y = wave_max(m.y); m.z = wave_min(m.z); m.w = wave_max(m.w);
    if (lane == 0) { tmp[0][wave] = m; }
  }
  __syncthreads();
  if (t < 2) {
    const float4 a = tmp[0][2 * t], b = tmp[0][2 * t + 1];
    part0[t][0] = fminf(a.x, b.x); part0[t][1] = fmaxf(a.y, b.y); part0[t][2] = fminf(a.z, b.z); part0[t][3] = fmaxf(a.w, b.w);
  }
  __syncthreads();
  bool certify = force != 2;                              // block-uniform throughout
  int ox[2] = {0, 0}, oy[2] = {0, 0}, ex[2] = {0, 0}, ey[2] = {0, 0}, shx[2] = {0, 0}, shy[2] = {0, 0};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    float lo[2], hi[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      lo[k] = part0[d][2 * k]; hi[k] = part0[d][2 * k + 1];
      // false for NaN / inf too; the magnitude bound keeps the casts below in range (such a tile is far outside anyway)
      certify = certify && (hi[k] - lo[k] <= 8.f * MB * 2.f) && fabsf(lo[k]) <= 1e6f && fabsf(hi[k]) <= 1e6f;
    }
    if (certify) {
      block_offsets(lo[0], hi[0], ox[d], ex[d]);
      block_offsets(lo[1], hi[1], oy[d], ey[d]);
      const int span_x = MB + (TBX - ucols), span_y = MB + (TBY - urows);
      certify = ex[d] - ox[d] <= MB + span_x && ey[d] - oy[d] <= MB + span_y;
      shx[d] = min(max((ox[d] + ex[d]) / 2, ex[d] - span_x), ox[d] + MB);
      shy[d] = min(max((oy[d] + ey[d]) / 2, ey[d] - span_y), oy[d] + MB);
    }
  }
  certify = uniform(certify);
  if (certify) {                                          // the windows: direction d samples the boxes of direction 1 - d
    for (int e = t; e < 2 * WBH * WBW; e += NT_C) {
      const int d = e >= WBH * WBW, q = e - d * WBH * WBW, r = q / WBW, c = q - r * WBW;
      const int br = min(max(by0 - MB + shy[d] + r, 0), Hb - 1), bc = min(max(bx0 - MB + shx[d] + c, 0), Wb - 1);
      ow[d][q] = bbd[1 - d][(size_t)br * Wb + bc];
    }
  }
  __syncthreads();
  if (certify) {                                          // row pass
    for (int e = t; e < 2 * WBH * TBX; e += NT_C) {
      const int d = e >= WBH * TBX, q = e - d * WBH * TBX, r = q / TBX, j = q - r * TBX;
      if (r >= MB + oy[d] - shy[d] && r <= MB + urows - 1 + ey[d] - shy[d] && j < ucols) {
        float4 m = make_float4(INF, -INF, INF, -INF);
        for (int k = ox[d] - shx[d]; k <= ex[d] - shx[d]; ++k) {
          const float4 v = ow[d][r * WBW + MB + j + k];
          m.x = fminf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fminf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        tmp[d][q] = m;
      }
    }
  }
  __syncthreads();
  if (certify) {                                          // column pass -> normalised boxes
    for (int e = t; e < 2 * TBY * TBX; e += NT_C) {
      const int d = e >= TBY * TBX, q = e - d * TBY * TBX, i = q / TBX, j = q - i * TBX;
      if (i < urows && j < ucols) {
        float4 m = make_float4(INF, -INF, INF, -INF);
        for (int k = oy[d] - shy[d]; k <= ey[d] - shy[d]; ++k) {
          const float4 v = tmp[d][(MB + i + k) * TBX + j];
          m.x = fminf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fminf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        float4 b;
        norm_box_dense(m.x, m.y, g.wd, b.x, b.y);
        norm_box_dense(m.z, m.w, g.ht, b.z, b.w);
        tab[d][q] = b;
        const float wxb = b.y - b.x, wyb = b.w - b.z;
        if (!(wxb * wxb + wyb * wyb <= g.a2)) atomicAdd(&wide_boxes, 1);     // cannot certify anything (NaN / inf too)
      }
    }
  }
  __syncthreads();
  if (force != 1) certify = certify && 4 * wide_boxes <= 2 * urows * ucols;
  certify = uniform(certify);
  if (!certify) {                                         // both staged tiles under this one, in both masks
    if (t < 4 && ty0 + (t & 1) * TY < g.Hf)
      ((t & 2) ? mask_bwd : mask_fwd)[(size_t)n * HW + (size_t)(ty0 + (t & 1) * TY) * g.Wf + tx0] = TILE_MARK;
    return;
  }
  const float a2c = g.a2 * (1.f - 2.f * BOX_SLOP);
  const int x = tx0 + lane;
  if (x >= g.Wf) return;                                  // whole waves' lane groups; no barrier below
  const float c0x = div_c(2.f * (float)x, g.wd) - 1.f;
  const float* ff = flow_fwd + (size_t)n * 2 * HW + x;
  const float* fb = flow_bwd + (size_t)n * 2 * HW + x;
  const int bj = lane >> 3;
  int badf = 0, badb = 0;
  unsigned undecided = 0;
  const int yw = ty0 + wave * ROWS_C;
  float4 bxf = make_float4(0.f, 0.f, 0.f, 0.f), bxb = bxf;
  // Certificate for a whole 8 x 8 block from its OWN box: the own flow of its 64 pixels lies in that box, their positions
  // between the block's first and last column / row; if the bound holds for the extremes, the block's mask bytes are 1
  // and its 16 B / pixel of flow are never read.
  const float cxa = div_c(2.f * (float)(tx0 + 8 * bj), g.wd) - 1.f, cxb = div_c(2.f * (float)(tx0 + 8 * bj + 7), g.wd) - 1.f;
  bool allok = false;
#pragma unroll 4
  for (int rr = 0; rr < ROWS_C; ++rr) {
    const int y = yw + rr;
    if (y >= g.Hf) break;                                 // wave-uniform
    if ((rr & 7) == 0) {
      const int ci = ((y - ty0) >> 3) * TBX + bj;
      bxf = tab[0][ci]; bxb = tab[1][ci];
      const float cya = div_c(2.f * (float)y, g.ht) - 1.f, cyb = div_c(2.f * (float)(y + 7), g.ht) - 1.f;
      bool bok = true;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float4 o = own[d][ci], bx = d ? bxb : bxf;
        float fLx, fUx, fLy, fUy;
        norm_box_dense(o.x, o.y, g.wd, fLx, fUx);
        norm_box_dense(o.z, o.w, g.ht, fLy, fUy);
        const bool in_all = (cxa + fLx > -1.f) && (cxb + fUx < 1.f) && (cya + fLy > -1.f) && (cyb + fUy < 1.f);
        const float mx = fmaxf(fabsf(fLx + bx.x), fabsf(fUx + bx.y));
        const float my = fmaxf(fabsf(fLy + bx.z), fabsf(fUy + bx.w));
        bok = bok && in_all && (__builtin_fmaf(mx, mx, my * my) <= a2c);       // false for NaN / inf
      }
      allok = __all(bok) != 0;
    }
    const size_t q = (size_t)y * g.Wf;
    if (allok) {                                          // wave-uniform
      mask_fwd[(size_t)n * HW + q + x] = 1;
      mask_bwd[(size_t)n * HW + q + x] = 1;
      continue;
    }
    const float c0y = div_c(2.f * (float)y, g.ht) - 1.f;
    const float fn[2][2] = {{div_c(2.f * ff[q], g.wd), div_c(2.f * ff[HW + q], g.ht)},
                            {div_c(2.f * fb[q], g.wd), div_c(2.f * fb[HW + q], g.ht)}};
    bool ok[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float4 bx = d ? bxb : bxf;
      const float c1x = c0x + fn[d][0], c1y = c0y + fn[d][1];
      const bool inside = (fabsf(c1x) < 1.f) && (fabsf(c1y) < 1.f);
      const float mx = fmaxf(fabsf(fn[d][0] + bx.x), fabsf(fn[d][0] + bx.y));
      const float my = fmaxf(fabsf(fn[d][1] + bx.z), fabsf(fn[d][1] + bx.w));
      const bool sure = __builtin_fmaf(mx, mx, my * my) <= a2c;
      ok[d] = inside && sure;
      if (inside && !sure) undecided |= 1u << (2 * rr + d);
    }
    mask_fwd[(size_t)n * HW + q + x] = ok[0] ? 1 : 0;
    mask_bwd[(size_t)n * HW + q + x] = ok[1] ? 1 : 0;
    badf += __popcll(__ballot(!ok[0]));
    badb += __popcll(__ballot(!ok[1]));
  }
  while (__any(undecided != 0)) {
    bool fixed_f = false, fixed_b = false;
    if (undecided) {
      const int bit = __builtin_ctz(undecided);
      undecided &= undecided - 1;
      const int d = bit & 1, y = yw + (bit >> 1);
      const float* fo = flow_fwd + (size_t)n * 2 * HW;
      const float* bo = flow_bwd + (size_t)n * 2 * HW;
      if (fbc_dense_generic(d ? bo : fo, d ? fo : bo, g.Hf, g.Wf, y, x, g.alpha1, g.a2)) {
        (d ? mask_bwd : mask_fwd)[(size_t)n * HW + (size_t)y * g.Wf + x] = 1;
        fixed_f = d == 0; fixed_b = d == 1;
      }
    }
    badf -= __popcll(__ballot(fixed_f));
    badb -= __popcll(__ballot(fixed_b));
  }
  if (invalid && (lane & 63) == __builtin_ctzll(__ballot(true))) {
    if (badf) atomicAdd(&invalid[n], badf);
    if (badb) atomicAdd(&invalid[N + n], badb);
  }
}

// ---- K8: calc_mask_ratio -----------------------------------------------------------------
__global__ void k_mask_ratio(const uint8_t* __restrict__ mask, float* __restrict__ ratio, int64_t pixels) {
  __shared__ int red[TPB / WAVE];
  const int n = blockIdx.x;
  const uint8_t* m = mask + (size_t)n * pixels;
  int acc = 0;
  for (int64_t e = threadIdx.x; e < pixels; e += TPB) acc += (m[e] == 0);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < TPB / WAVE; ++w) t += red[w];
    ratio[n] = (float)((double)t / (double)pixels);
  }
}

// ---- K4 stand-alone: add_optical_flow ------------------------------------------------------
__global__ void k_add_optical_flow(pixpro_problem pb, const float* __restrict__ flow, const float* __restrict__ xg,
                                   const float* __restrict__ yg, const uint8_t* __restrict__ mask,
                                   float* __restrict__ ox, float* __restrict__ oy, uint8_t* __restrict__ mg) {
  const int P = pb.H * pb.W;
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)pb.N * P) return;
  const int n = (int)(idx / P);
  const FlowGeom fg = flow_geom(pb.Hf, pb.Wf, pb.Ho, pb.Wo);
  float x, y;
  bool valid;
  displace_point(flow + (size_t)n * 2 * pb.Hf * pb.Wf, mask ? mask + (size_t)n * pb.Hf * pb.Wf : nullptr, fg,
                 xg[idx], yg[idx], x, y, valid);
  ox[idx] = x; oy[idx] = y;
  if (mg) mg[idx] = valid ? 1 : 0;
}

#define CHECKED(what)                      \
  do {                                     \
    if (int e = check_launch(what)) return e; \
  } while (0)

}  // namespace

int run_upflow8(const float* in, float* out, int64_t planes, int h, int w, hipStream_t st) {
  const size_t n = (size_t)planes * 64 * h * w;
  hipLaunchKernelGGL(k_upflow8, dim3(blocks_for(n)), dim3(TPB), 0, st, in, out, planes, h, w);
  CHECKED("k_upflow8");
  return PIXPRO_OK;
}

int run_concat_flow(const float* flows, float* out, int F1, int N, int Hf, int Wf, int is_norm, hipStream_t st) {
  DenseSource s{flows, N, Hf, Wf};
  const size_t n = (size_t)N * Hf * Wf;
  hipLaunchKernelGGL(k_chain<DenseSource>, dim3(blocks_for(n), 1), dim3(TPB), 0, st, s, s, out, out, F1, N, Hf, Wf,
                     is_norm);
  CHECKED("k_chain");
  return PIXPRO_OK;
}

int run_fbc(const float* f, const float* b, uint8_t* mask, float* cycle, int N, int Hf, int Wf, float alpha1,
            double alpha2, int is_norm, hipStream_t st) {
  const size_t n = (size_t)N * Hf * Wf;
  hipLaunchKernelGGL(k_fbc, dim3(blocks_for(n)), dim3(TPB), 0, st, f, b, mask, cycle, (int32_t*)nullptr, N, Hf, Wf,
                     alpha1, fbc_a2(alpha2, Hf, Wf), is_norm);
  CHECKED("k_fbc");
  return PIXPRO_OK;
}

static Up8Geom make_geom(int h, int w, int Hf, int Wf, float alpha1, float a2) {
  Up8Geom g;
  g.h = h; g.w = w; g.Hf = Hf; g.Wf = Wf;
  g.sy = h > 1 ? (float)(h - 1) / (float)(Hf - 1) : 0.f;
  g.sx = w > 1 ? (float)(w - 1) / (float)(Wf - 1) : 0.f;
  g.wd = make_divc((float)(Wf - 1)); g.ht = make_divc((float)(Hf - 1));
  g.hx = (float)(Wf - 1) / 2.f; g.hy = (float)(Hf - 1) / 2.f;
  g.alpha1 = alpha1; g.a2 = a2;
  return g;
}

// Scratch of the certified mask path for F1 > 1: one float4 box per 8 x 8 block of both composed flows.
size_t flow_prepare_workspace_bytes(int N, int F1, int h, int w, int flow_up) {
  if (F1 <= 1 || !flow_up || !up8_separable(h, w) || F1 > CH_MAXF) return 0;
  return (size_t)2 * N * h * w * sizeof(float4);
}

int run_flow_prepare(const float* lo_fwd, const float* lo_bwd, float* flow_fwd, float* flow_bwd, uint8_t* mask_fwd,
                     uint8_t* mask_bwd, int32_t* invalid, int N, int F1, int h, int w, int flow_up, float alpha1,
                     double alpha2, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int Hf = flow_up ? 8 * h : h, Wf = flow_up ? 8 * w : w;
  const size_t n = (size_t)N * Hf * Wf;
  const float a2 = fbc_a2(alpha2, Hf, Wf);
  if (invalid) {
    if (hipMemsetAsync(invalid, 0, sizeof(int32_t) * 2 * N, st) != hipSuccess) {
      set_error("hipMemsetAsync failed");
      return PIXPRO_ERR_LAUNCH;
    }
  }
  // the tiled kernels implement ATen's separable up-sampling (out_H + out_W > 128: every real frame size); the few
  // smaller maps take the generic one-thread-per-pixel kernels, which know both forms
  const dim3 grid((Wf + TX - 1) / TX, (Hf + TY - 1) / TY, 1);
  const bool tiled = flow_up && up8_separable(h, w) && 2 * N <= 65535 && grid.y <= 65535;
  if (F1 == 1) {
    // k_prepare_cert stores two pixels per lane: 8-B flow stores, 2-B mask stores.  Outputs that are not aligned for that
    // (views at odd offsets) take the one-thread-per-pixel kernel, same bits.
    const bool pair_aligned = (((uintptr_t)flow_fwd | (uintptr_t)flow_bwd) & 7) == 0 &&
                              (!mask_fwd || (((uintptr_t)mask_fwd | (uintptr_t)mask_bwd) & 1) == 0);
    if (tiled && pair_aligned) {
      // PIXPRO_FLOW_PATH: auto (default) | certified (hand over only what the table cannot cover) | staged (hand over all):
      // identical results, a knob for the tests and for timing the two paths
      const char* path = getenv("PIXPRO_FLOW_PATH");
      int force = !path ? 0 : !strcmp(path, "certified") ? 1 : !strcmp(path, "staged") ? 2 : 0;
      if (!(alpha1 >= 0.f)) force = 2;      // the certificate needs eps >= a2, i.e. alpha1 >= 0
      const Up8Geom g = make_geom(h, w, Hf, Wf, alpha1, a2);
      hipLaunchKernelGGL(k_prepare_cert, dim3((Wf + TXC - 1) / TXC, (Hf + TYC - 1) / TYC, N), dim3(NT_C), 0, st, lo_fwd, lo_bwd, flow_fwd, flow_bwd,
                         mask_fwd, mask_bwd, invalid, N, g, force);
      if (mask_fwd) {
        CHECKED("k_prepare_cert");
        const long ntiles = (long)grid.x * grid.y * N;
        hipLaunchKernelGGL(k_prepare_staged, dim3((unsigned)(ntiles < 1024 ? ntiles : 1024)), dim3(NT), 0, st, lo_fwd,
                           lo_bwd, mask_fwd, mask_bwd, invalid, N, g, (int)grid.x, (int)grid.y);
      }
    } else if (flow_up) {
      Up8Source sf{lo_fwd, F1, h, w}, sb{lo_bwd, F1, h, w};
      hipLaunchKernelGGL(k_prepare_single<Up8Source>, dim3(blocks_for(n)), dim3(TPB), 0, st, sf, sb, flow_fwd,
                         flow_bwd, mask_fwd, mask_bwd, invalid, N, Hf, Wf, alpha1, a2);
    } else {
      LoSource sf{lo_fwd, F1, h, w}, sb{lo_bwd, F1, h, w};
      hipLaunchKernelGGL(k_prepare_single<LoSource>, dim3(blocks_for(n)), dim3(TPB), 0, st, sf, sb, flow_fwd,
                         flow_bwd, mask_fwd, mask_bwd, invalid, N, Hf, Wf, alpha1, a2);
    }
    CHECKED("k_prepare");
    return PIXPRO_OK;
  }
  // F-1 > 1: chains (both directions in one launch), then both masks in one launch
  if (tiled && F1 <= CH_MAXF) {
    const Up8Geom g = make_geom(h, w, Hf, Wf, alpha1, a2);
    // With a workspace (pixpro_flow_prepare_ws) the masks take the certified route: the chain leaves the bounding box of
    // every 8 x 8 block behind, k_fbc_cert_dense settles what those boxes can settle and marks the other tiles for
    // k_fbc_pair.  PIXPRO_FLOW_PATH as for F1 == 1.  Without one: k_fbc_pair on every tile (round 1's route).
    const char* path = getenv("PIXPRO_FLOW_PATH");
    int force = !path ? 0 : !strcmp(path, "certified") ? 1 : !strcmp(path, "staged") ? 2 : 0;
    if (!(alpha1 >= 0.f)) force = 2;        // the certificate needs eps >= a2, i.e. alpha1 >= 0
    const bool cert = mask_fwd && workspace && ((uintptr_t)workspace & 15) == 0 &&
                      workspace_bytes >= flow_prepare_workspace_bytes(N, F1, h, w, flow_up);
    float4* bbox = cert ? (float4*)workspace : nullptr;
    hipLaunchKernelGGL(k_chain_up8, dim3(grid.x, (Hf + TYC - 1) / TYC, 2 * N), dim3(NT_C), 0, st, lo_fwd, lo_bwd, flow_fwd,
                       flow_bwd, bbox, N, F1, g);
    CHECKED("k_chain_up8");
    if (mask_fwd) {
      if (cert) {
        hipLaunchKernelGGL(k_fbc_cert_dense, dim3(grid.x, (Hf + TYC - 1) / TYC, N), dim3(NT_C), 0, st, flow_fwd, flow_bwd,
                           bbox, mask_fwd, mask_bwd, invalid, N, g, force);
        CHECKED("k_fbc_cert_dense");
      }
      if (cert) {
        const long nent = (long)grid.x * grid.y * 2 * N;
        hipLaunchKernelGGL(k_fbc_pair_marked, dim3((unsigned)(nent < 1024 ? nent : 1024)), dim3(NT), 0, st, flow_fwd, flow_bwd,
                           mask_fwd, mask_bwd, invalid, N, g, (int)grid.x, (int)grid.y);
      } else {
        hipLaunchKernelGGL(k_fbc_pair, dim3(grid.x, grid.y, 2 * N), dim3(NT), 0, st, flow_fwd, flow_bwd, mask_fwd,
                           mask_bwd, invalid, N, g);
      }
      CHECKED("k_fbc_pair");
    }
    return PIXPRO_OK;
  }
  if (flow_up) {
    Up8Source sf{lo_fwd, F1, h, w}, sb{lo_bwd, F1, h, w};
    hipLaunchKernelGGL(k_chain<Up8Source>, dim3(blocks_for(n), 2), dim3(TPB), 0, st, sf, sb, flow_fwd, flow_bwd, F1,
                       N, Hf, Wf, 0);
  } else {
    LoSource sf{lo_fwd, F1, h, w}, sb{lo_bwd, F1, h, w};
    hipLaunchKernelGGL(k_chain<LoSource>, dim3(blocks_for(n), 2), dim3(TPB), 0, st, sf, sb, flow_fwd, flow_bwd, F1,
                       N, Hf, Wf, 0);
  }
  CHECKED("k_chain");
  if (mask_fwd) {
    if (2 * N <= 65535 && grid.y <= 65535) {       // dense inputs: the tiled mask kernel does not up-sample anything
      hipLaunchKernelGGL(k_fbc_pair, dim3(grid.x, grid.y, 2 * N), dim3(NT), 0, st, flow_fwd, flow_bwd, mask_fwd,
                         mask_bwd, invalid, N, make_geom(h, w, Hf, Wf, alpha1, a2));
      CHECKED("k_fbc_pair");
    } else {
      hipLaunchKernelGGL(k_fbc, dim3(blocks_for(n)), dim3(TPB), 0, st, flow_fwd, flow_bwd, mask_fwd, (float*)nullptr,
                         invalid, N, Hf, Wf, alpha1, a2, 0);
      CHECKED("k_fbc");
      hipLaunchKernelGGL(k_fbc, dim3(blocks_for(n)), dim3(TPB), 0, st, flow_bwd, flow_fwd, mask_bwd, (float*)nullptr,
                         invalid ? invalid + N : nullptr, N, Hf, Wf, alpha1, a2, 0);
      CHECKED("k_fbc");
    }
  }
  return PIXPRO_OK;
}

int run_mask_ratio(const uint8_t* mask, float* ratio, int N, int64_t pixels, hipStream_t st) {
  hipLaunchKernelGGL(k_mask_ratio, dim3(N), dim3(TPB), 0, st, mask, ratio, pixels);
  CHECKED("k_mask_ratio");
  return PIXPRO_OK;
}

int run_add_optical_flow(const pixpro_problem& pb, const float* flow, const float* xg, const float* yg,
                         const uint8_t* mask, float* ox, float* oy, uint8_t* mg, hipStream_t st) {
  const size_t n = (size_t)pb.N * pb.H * pb.W;
  hipLaunchKernelGGL(k_add_optical_flow, dim3(blocks_for(n)), dim3(TPB), 0, st, pb, flow, xg, yg, mask, ox, oy, mg);
  CHECKED("k_add_optical_flow");
  return PIXPRO_OK;
}

}  // namespace pixpro
