// Flow preparation kernels (HBM-bound, one thread per full-resolution pixel, coalesced along x).
//
// Reference: contrast/util.py:174-248 (apply_optical_flow, file mode), :300-330 (concat_flow),
// :252-297 (forward_backward_consistency), :333-366 (normalise helpers, calc_mask_ratio),
// contrast/flow/utils/utils.py:87-89 (upflow8), contrast/models/PixPro.py:46-89
// (add_optical_flow).  Arithmetic order follows oracle/pixpro_oracle.py (bit-exact targets).
#include "common.h"

namespace pixpro {
namespace {

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
                         ay, ax);
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

// ---- production form of the above: flow_up, F-1 == 1 ----------------------------------------
// Same arithmetic, restructured for instruction count (the generic kernel is VALU-bound):
//   * 2-D tiles of 256 x 4 pixels, 4 consecutive pixels per thread -> no integer division for
//     indexing, 16-B flow stores and 4-B mask stores;
//   * divisions by the per-launch constants (Wf-1), (Hf-1) as q0 = a*r, q = fma(fma(-q0,b,a), r, q0)
//     with r = RN(1/b): the correctly rounded quotient (Markstein), 3 instructions instead of ~10;
//   * a pixel whose displaced position leaves the frame is invalid whatever the cycle error, so its
//     taps are skipped; inside the frame all four taps are in range and share two x- and two y-axes
//     of the lazy x8 up-sampling.
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

// Block-local LDS window over the four low-res channels {fwd.x, fwd.y, bwd.x, bwd.y}, one float4 per
// low-res cell, replicate-padded (index clamping reproduces upflow8's edge clamps exactly).  It covers the
// block's own footprint plus a margin of >= 6 low-res cells (~48 full-res pixels of displacement); taps that
// fall outside it read global memory instead.
constexpr int PT_X = 64, PT_Y = 4, PPX = 4;   // 64 x 4 threads, 4 pixels per thread along x
constexpr int WIN_W = 48, WIN_H = 16, WIN_MX = 7, WIN_MY = 6;

__shared__ float4 win_cells[WIN_H * WIN_W];   // file scope keeps the LDS address space (ds_read, not flat)

struct Win {
  int wx0, wy0;          // low-res coordinates of cell (0,0)
  const float* g[4];     // global fallback: fwd.x, fwd.y, bwd.x, bwd.y
  int h, w;
};

// Lean axis of the x8 up-sampling for positions inside the image: src = dst*scale < n_in always, so ATen's
// clamps of i0 and of the weight are no-ops; the +1 neighbour is left to the caller (replicated LDS window,
// or an explicit clamp on the global path).
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

// the 2x2 low-res neighbourhood (i0y..i0y+1) x (i0x..i0x+1) from global memory, clamped at the borders
__device__ __forceinline__ void lo_quad_global(const Win& wn, int i0y, int i0x, float4& a, float4& b, float4& c,
                                               float4& d) {
  const int x1 = min(i0x + 1, wn.w - 1), y1 = min(i0y + 1, wn.h - 1);
  const unsigned o00 = i0y * wn.w + i0x, o01 = i0y * wn.w + x1, o10 = y1 * wn.w + i0x, o11 = y1 * wn.w + x1;
  a = make_float4(wn.g[0][o00], wn.g[1][o00], wn.g[2][o00], wn.g[3][o00]);
  b = make_float4(wn.g[0][o01], wn.g[1][o01], wn.g[2][o01], wn.g[3][o01]);
  c = make_float4(wn.g[0][o10], wn.g[1][o10], wn.g[2][o10], wn.g[3][o10]);
  d = make_float4(wn.g[0][o11], wn.g[1][o11], wn.g[2][o11], wn.g[3][o11]);
}

// 16 * (weighted sum) = 2 * upflow8 value: both scalings are exact
__device__ __forceinline__ float up8_x16(float a, float b, float c, float d, float w00, float w01, float w10, float w11) {
  float acc = b * w01;
  acc = __builtin_fmaf(a, w00, acc);
  acc = __builtin_fmaf(c, w10, acc);
  acc = __builtin_fmaf(d, w11, acc);
  return 16.f * acc;
}

// validity of one pixel given its own flow (fx, fy); OPP selects the opposite direction's channels
// (0: fwd = .x/.y of the cell, 1: bwd = .z/.w)
template <int OPP>
__device__ __forceinline__ bool fbc_up8(float fx, float fy, const Win& wn, const Up8Geom& g, float c0x, float c0y) {
  const float fnx = div_c(2.f * fx, g.wd), fny = div_c(2.f * fy, g.ht);
  const float c1x = c0x + fnx, c1y = c0y + fny;
  if (!((fabsf(c1x) < 1.f) && (fabsf(c1y) < 1.f))) return false;
  const float ix = (c1x + 1.f) * g.hx, iy = (c1y + 1.f) * g.hy;
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
  const int x0 = (int)xw, y0 = (int)yn;
  // x0+1 == Wf (y0+1 == Hf) only when the weight w (n) is exactly 0: the clamped tap is finite and contributes 0
  const AxF ax[2] = {axf(x0, g.sx), axf(min(x0 + 1, g.Wf - 1), g.sx)};
  const AxF ay[2] = {axf(y0, g.sy), axf(min(y0 + 1, g.Hf - 1), g.sy)};
  const int rx0 = ax[0].i0 - wn.wx0, rx1 = ax[1].i0 - wn.wx0, ry0 = ay[0].i0 - wn.wy0, ry1 = ay[1].i0 - wn.wy0;
  const bool in_win = (rx0 >= 0) && (rx1 < WIN_W - 1) && (ry0 >= 0) && (ry1 < WIN_H - 1);
  float tx[4], ty[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const AxF& vy_ = ay[t >> 1];
    const AxF& vx_ = ax[t & 1];
    const float w00 = vy_.l0 * vx_.l0, w01 = vy_.l0 * vx_.l1, w10 = vy_.l1 * vx_.l0, w11 = vy_.l1 * vx_.l1;
    float4 a, b, c, d;
    if (__builtin_expect(in_win, 1)) {
      const int o = ((t >> 1) ? ry1 : ry0) * WIN_W + ((t & 1) ? rx1 : rx0);
      a = win_cells[o]; b = win_cells[o + 1]; c = win_cells[o + WIN_W]; d = win_cells[o + WIN_W + 1];
    } else {
      lo_quad_global(wn, vy_.i0, vx_.i0, a, b, c, d);
    }
    tx[t] = div_c(OPP ? up8_x16(a.z, b.z, c.z, d.z, w00, w01, w10, w11)
                      : up8_x16(a.x, b.x, c.x, d.x, w00, w01, w10, w11), g.wd);
    ty[t] = div_c(OPP ? up8_x16(a.w, b.w, c.w, d.w, w00, w01, w10, w11)
                      : up8_x16(a.y, b.y, c.y, d.y, w00, w01, w10, w11), g.ht);
  }
  const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
  const float bx = __builtin_fmaf(tx[3], se, __builtin_fmaf(tx[2], sw, __builtin_fmaf(tx[1], ne, tx[0] * nw)));
  const float by = __builtin_fmaf(ty[3], se, __builtin_fmaf(ty[2], sw, __builtin_fmaf(ty[1], ne, ty[0] * nw)));
  const float cx = fnx + bx, cy = fny + by;
  const float cyc = cx * cx + cy * cy;
  const float eps = g.alpha1 * ((fnx * fnx + fny * fny) + (bx * bx + by * by)) + g.a2;
  return (cyc - eps) <= 0.f;
}

__global__ void __launch_bounds__(PT_X* PT_Y)
k_prepare_up8(const float* __restrict__ lo_fwd, const float* __restrict__ lo_bwd, float* __restrict__ flow_fwd,
              float* __restrict__ flow_bwd, uint8_t* __restrict__ mask_fwd, uint8_t* __restrict__ mask_bwd,
              int32_t* __restrict__ invalid, int N, Up8Geom g) {
  const int n = blockIdx.z;
  const int y = blockIdx.y * PT_Y + threadIdx.y;
  const int xb = (blockIdx.x * PT_X + threadIdx.x) * PPX;
  const size_t hw = (size_t)g.h * g.w, HW = (size_t)g.Hf * g.Wf;
  Win wn;
  wn.g[0] = lo_fwd + (size_t)n * 2 * hw; wn.g[1] = wn.g[0] + hw;
  wn.g[2] = lo_bwd + (size_t)n * 2 * hw; wn.g[3] = wn.g[2] + hw;
  wn.h = g.h; wn.w = g.w;
  wn.wx0 = up8_axis(blockIdx.x * PT_X * PPX, g.w, g.sx).i0 - WIN_MX;
  wn.wy0 = up8_axis(blockIdx.y * PT_Y, g.h, g.sy).i0 - WIN_MY;
  for (int e = threadIdx.y * PT_X + threadIdx.x; e < WIN_H * WIN_W; e += PT_X * PT_Y) {
    const int r = e / WIN_W, c = e % WIN_W;
    const unsigned o = min(max(wn.wy0 + r, 0), g.h - 1) * g.w + min(max(wn.wx0 + c, 0), g.w - 1);
    win_cells[e] = make_float4(wn.g[0][o], wn.g[1][o], wn.g[2][o], wn.g[3][o]);
  }
  __syncthreads();
  int badf = 0, badb = 0;
  if (y < g.Hf && xb < g.Wf) {
    const AxF ay = axf(y, g.sy);
    const int ory = (ay.i0 - wn.wy0) * WIN_W - wn.wx0;
    const float c0y = div_c(2.f * (float)y, g.ht) - 1.f;
    float ffx[PPX], ffy[PPX], fbx[PPX], fby[PPX];
    uint8_t mf[PPX], mb[PPX];
#pragma unroll
    for (int i = 0; i < PPX; ++i) {
      const int x = min(xb + i, g.Wf - 1);
      const AxF ax = axf(x, g.sx);
      const float w00 = ay.l0 * ax.l0, w01 = ay.l0 * ax.l1, w10 = ay.l1 * ax.l0, w11 = ay.l1 * ax.l1;
      const int o = ory + ax.i0;                       // the block's own footprint is inside the window
      const float4 a = win_cells[o], b = win_cells[o + 1], c = win_cells[o + WIN_W], d = win_cells[o + WIN_W + 1];
      const float fx = 0.5f * up8_x16(a.x, b.x, c.x, d.x, w00, w01, w10, w11);
      const float fy = 0.5f * up8_x16(a.y, b.y, c.y, d.y, w00, w01, w10, w11);
      const float bx = 0.5f * up8_x16(a.z, b.z, c.z, d.z, w00, w01, w10, w11);
      const float by = 0.5f * up8_x16(a.w, b.w, c.w, d.w, w00, w01, w10, w11);
      ffx[i] = fx; ffy[i] = fy; fbx[i] = bx; fby[i] = by;
      mf[i] = mb[i] = 0;
      if (mask_fwd) {
        const float c0x = div_c(2.f * (float)x, g.wd) - 1.f;
        mf[i] = fbc_up8<1>(fx, fy, wn, g, c0x, c0y) ? 1 : 0;
        mb[i] = fbc_up8<0>(bx, by, wn, g, c0x, c0y) ? 1 : 0;
        if (xb + i < g.Wf) { badf += 1 - mf[i]; badb += 1 - mb[i]; }
      }
    }
    const size_t r = (size_t)y * g.Wf + xb;
    float* pf = flow_fwd + (size_t)n * 2 * HW + r;
    float* pb = flow_bwd + (size_t)n * 2 * HW + r;
    if (xb + PPX <= g.Wf && (g.Wf & 3) == 0) {
      *reinterpret_cast<float4*>(pf) = make_float4(ffx[0], ffx[1], ffx[2], ffx[3]);
      *reinterpret_cast<float4*>(pf + HW) = make_float4(ffy[0], ffy[1], ffy[2], ffy[3]);
      *reinterpret_cast<float4*>(pb) = make_float4(fbx[0], fbx[1], fbx[2], fbx[3]);
      *reinterpret_cast<float4*>(pb + HW) = make_float4(fby[0], fby[1], fby[2], fby[3]);
      if (mask_fwd) {
        *reinterpret_cast<uchar4*>(mask_fwd + (size_t)n * HW + r) = make_uchar4(mf[0], mf[1], mf[2], mf[3]);
        *reinterpret_cast<uchar4*>(mask_bwd + (size_t)n * HW + r) = make_uchar4(mb[0], mb[1], mb[2], mb[3]);
      }
    } else {
      for (int i = 0; i < PPX && xb + i < g.Wf; ++i) {
        pf[i] = ffx[i]; pf[HW + i] = ffy[i]; pb[i] = fbx[i]; pb[HW + i] = fby[i];
        if (mask_fwd) { mask_fwd[(size_t)n * HW + r + i] = mf[i]; mask_bwd[(size_t)n * HW + r + i] = mb[i]; }
      }
    }
  }
  if (invalid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { badf += __shfl_xor(badf, o, 64); badb += __shfl_xor(badb, o, 64); }
    if (threadIdx.x == 0) {
      if (badf) atomicAdd(&invalid[n], badf);
      if (badb) atomicAdd(&invalid[N + n], badb);
    }
  }
}

// ---- multi-frame (F-1 > 1) production kernels -----------------------------------------------------
// k_chain_up8: concat_flow over lazily up-sampled 1/8-resolution maps.  Same tiling / LDS-window / exact
// constant-division scheme as k_prepare_up8; one window per flow of the chain (float2 cells), direction in
// blockIdx.z.  A chain that drifts out of its block's window reads global memory; a tap that leaves the frame
// contributes zero (grid_sample zeros padding).
constexpr int CH_MAXF = 6;       // windows kept in LDS: 6 * 48 * 16 * 8 B = 36 KB
__shared__ float2 chain_cells[CH_MAXF * WIN_H * WIN_W];

__device__ __forceinline__ void chain_quad(const float* __restrict__ gx, const float* __restrict__ gy, int t, int wx0,
                                           int wy0, int h, int w, int i0y, int i0x, float2& a, float2& b, float2& c,
                                           float2& d) {
  const int ry = i0y - wy0, rx = i0x - wx0;
  if (__builtin_expect((unsigned)ry < (unsigned)(WIN_H - 1) && (unsigned)rx < (unsigned)(WIN_W - 1), 1)) {
    const int o = (t * WIN_H + ry) * WIN_W + rx;
    a = chain_cells[o]; b = chain_cells[o + 1]; c = chain_cells[o + WIN_W]; d = chain_cells[o + WIN_W + 1];
  } else {
    const int x1 = min(i0x + 1, w - 1), y1 = min(i0y + 1, h - 1);
    const unsigned o00 = i0y * w + i0x, o01 = i0y * w + x1, o10 = y1 * w + i0x, o11 = y1 * w + x1;
    a = make_float2(gx[o00], gy[o00]); b = make_float2(gx[o01], gy[o01]);
    c = make_float2(gx[o10], gy[o10]); d = make_float2(gx[o11], gy[o11]);
  }
}

__global__ void __launch_bounds__(PT_X* PT_Y)
k_chain_up8(const float* __restrict__ lo_fwd, const float* __restrict__ lo_bwd, float* __restrict__ flow_fwd,
            float* __restrict__ flow_bwd, int N, int F1, Up8Geom g) {
  const int n = blockIdx.z >> 1, dir = blockIdx.z & 1;
  const float* lo = (dir ? lo_bwd : lo_fwd) + (size_t)n * F1 * 2 * g.h * g.w;   // [F1][2][h][w]
  float* out = (dir ? flow_bwd : flow_fwd) + (size_t)n * 2 * g.Hf * g.Wf;
  const size_t hw = (size_t)g.h * g.w, HW = (size_t)g.Hf * g.Wf;
  const int y = blockIdx.y * PT_Y + threadIdx.y;
  const int xb = (blockIdx.x * PT_X + threadIdx.x) * PPX;
  const int wx0 = axf(blockIdx.x * PT_X * PPX, g.sx).i0 - WIN_MX;
  const int wy0 = axf(blockIdx.y * PT_Y, g.sy).i0 - WIN_MY;
  for (int e = threadIdx.y * PT_X + threadIdx.x; e < F1 * WIN_H * WIN_W; e += PT_X * PT_Y) {
    const int t = e / (WIN_H * WIN_W), r = (e / WIN_W) % WIN_H, c = e % WIN_W;
    const unsigned o = min(max(wy0 + r, 0), g.h - 1) * g.w + min(max(wx0 + c, 0), g.w - 1);
    chain_cells[e] = make_float2(lo[(size_t)t * 2 * hw + o], lo[(size_t)t * 2 * hw + hw + o]);
  }
  __syncthreads();
  if (y >= g.Hf || xb >= g.Wf) return;
  float ox[PPX], oy[PPX];
#pragma unroll
  for (int i = 0; i < PPX; ++i) {
    const int x = min(xb + i, g.Wf - 1);
    const float x0f = (float)x, y0f = (float)y;
    float cx = x0f, cy = y0f;
    for (int t = 0; t < F1; ++t) {
      const float* gx = lo + (size_t)t * 2 * hw;
      const float* gy = gx + hw;
      const float nx = div_c(2.f * cx, g.wd) - 1.f, ny = div_c(2.f * cy, g.ht) - 1.f;   // normalize_coord
      const float ix = (nx + 1.f) * g.hx, iy = (ny + 1.f) * g.hy;
      const float xw = floorf(ix), yn = floorf(iy);
      const float w = ix - xw, e = 1.f - w, nn = iy - yn, s = 1.f - nn;
      const int tx0 = (xw > -2.f && xw < (float)(g.Wf + 1)) ? (int)xw : -2;
      const int ty0 = (yn > -2.f && yn < (float)(g.Hf + 1)) ? (int)yn : -2;
      const bool vx[2] = {tx0 >= 0 && tx0 < g.Wf, tx0 + 1 >= 0 && tx0 + 1 < g.Wf};
      const bool vy[2] = {ty0 >= 0 && ty0 < g.Hf, ty0 + 1 >= 0 && ty0 + 1 < g.Hf};
      const AxF ax[2] = {axf(min(max(tx0, 0), g.Wf - 1), g.sx), axf(min(max(tx0 + 1, 0), g.Wf - 1), g.sx)};
      const AxF ay[2] = {axf(min(max(ty0, 0), g.Hf - 1), g.sy), axf(min(max(ty0 + 1, 0), g.Hf - 1), g.sy)};
      float tvx[4], tvy[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const AxF& qy = ay[k >> 1];
        const AxF& qx = ax[k & 1];
        const float w00 = qy.l0 * qx.l0, w01 = qy.l0 * qx.l1, w10 = qy.l1 * qx.l0, w11 = qy.l1 * qx.l1;
        float2 a, b, c, d;
        chain_quad(gx, gy, t, wx0, wy0, g.h, g.w, qy.i0, qx.i0, a, b, c, d);
        const bool ok = vx[k & 1] && vy[k >> 1];
        tvx[k] = ok ? 0.5f * up8_x16(a.x, b.x, c.x, d.x, w00, w01, w10, w11) : 0.f;
        tvy[k] = ok ? 0.5f * up8_x16(a.y, b.y, c.y, d.y, w00, w01, w10, w11) : 0.f;
      }
      const float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
      const float fx = __builtin_fmaf(tvx[3], se, __builtin_fmaf(tvx[2], sw, __builtin_fmaf(tvx[1], ne, tvx[0] * nw)));
      const float fy = __builtin_fmaf(tvy[3], se, __builtin_fmaf(tvy[2], sw, __builtin_fmaf(tvy[1], ne, tvy[0] * nw)));
      cx = cx + fx; cy = cy + fy;
    }
    ox[i] = cx - x0f; oy[i] = cy - y0f;
  }
  float* p = out + (size_t)y * g.Wf + xb;
  if (xb + PPX <= g.Wf && (g.Wf & 3) == 0) {
    *reinterpret_cast<float4*>(p) = make_float4(ox[0], ox[1], ox[2], ox[3]);
    *reinterpret_cast<float4*>(p + HW) = make_float4(oy[0], oy[1], oy[2], oy[3]);
  } else {
    for (int i = 0; i < PPX && xb + i < g.Wf; ++i) { p[i] = ox[i]; p[HW + i] = oy[i]; }
  }
}

// k_fbc_pair: both consistency masks of dense composed flows in one launch (direction in blockIdx.z & 1), 4 pixels
// per thread, 16 B flow loads, exact constant division; the opposite flow is gathered from HBM/L2.
__device__ __forceinline__ bool fbc_dense(float fx, float fy, const float* __restrict__ bx_, const float* __restrict__ by_,
                                          const Up8Geom& g, float c0x, float c0y) {
  const float fnx = div_c(2.f * fx, g.wd), fny = div_c(2.f * fy, g.ht);
  const float c1x = c0x + fnx, c1y = c0y + fny;
  if (!((fabsf(c1x) < 1.f) && (fabsf(c1y) < 1.f))) return false;
  const float ix = (c1x + 1.f) * g.hx, iy = (c1y + 1.f) * g.hy;
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
  const int x0 = (int)xw, y0 = (int)yn;
  const int x1 = min(x0 + 1, g.Wf - 1), y1 = min(y0 + 1, g.Hf - 1);   // clamped taps carry weight 0 (see fbc_up8)
  const unsigned o00 = y0 * g.Wf + x0, o01 = y0 * g.Wf + x1, o10 = y1 * g.Wf + x0, o11 = y1 * g.Wf + x1;
  const float tx0 = div_c(2.f * bx_[o00], g.wd), tx1 = div_c(2.f * bx_[o01], g.wd), tx2 = div_c(2.f * bx_[o10], g.wd),
              tx3 = div_c(2.f * bx_[o11], g.wd);
  const float ty0 = div_c(2.f * by_[o00], g.ht), ty1 = div_c(2.f * by_[o01], g.ht), ty2 = div_c(2.f * by_[o10], g.ht),
              ty3 = div_c(2.f * by_[o11], g.ht);
  const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
  const float bx = __builtin_fmaf(tx3, se, __builtin_fmaf(tx2, sw, __builtin_fmaf(tx1, ne, tx0 * nw)));
  const float by = __builtin_fmaf(ty3, se, __builtin_fmaf(ty2, sw, __builtin_fmaf(ty1, ne, ty0 * nw)));
  const float cx = fnx + bx, cy = fny + by;
  const float cyc = cx * cx + cy * cy;
  const float eps = g.alpha1 * ((fnx * fnx + fny * fny) + (bx * bx + by * by)) + g.a2;
  return (cyc - eps) <= 0.f;
}

__global__ void __launch_bounds__(PT_X* PT_Y)
k_fbc_pair(const float* __restrict__ flow_fwd, const float* __restrict__ flow_bwd, uint8_t* __restrict__ mask_fwd,
           uint8_t* __restrict__ mask_bwd, int32_t* __restrict__ invalid, int N, Up8Geom g) {
  const int n = blockIdx.z >> 1, dir = blockIdx.z & 1;
  const size_t HW = (size_t)g.Hf * g.Wf;
  const float* f = (dir ? flow_bwd : flow_fwd) + (size_t)n * 2 * HW;
  const float* o = (dir ? flow_fwd : flow_bwd) + (size_t)n * 2 * HW;
  uint8_t* mask = (dir ? mask_bwd : mask_fwd) + (size_t)n * HW;
  const int y = blockIdx.y * PT_Y + threadIdx.y;
  const int xb = (blockIdx.x * PT_X + threadIdx.x) * PPX;
  int bad = 0;
  if (y < g.Hf && xb < g.Wf) {
    const size_t r = (size_t)y * g.Wf + xb;
    const bool vec = xb + PPX <= g.Wf && (g.Wf & 3) == 0;
    float fx[PPX], fy[PPX];
    if (vec) {
      const float4 a = *reinterpret_cast<const float4*>(f + r), b = *reinterpret_cast<const float4*>(f + HW + r);
      fx[0] = a.x; fx[1] = a.y; fx[2] = a.z; fx[3] = a.w; fy[0] = b.x; fy[1] = b.y; fy[2] = b.z; fy[3] = b.w;
    } else {
      for (int i = 0; i < PPX; ++i) { const size_t q = (size_t)y * g.Wf + min(xb + i, g.Wf - 1); fx[i] = f[q]; fy[i] = f[HW + q]; }
    }
    const float c0y = div_c(2.f * (float)y, g.ht) - 1.f;
    uint8_t m[PPX];
#pragma unroll
    for (int i = 0; i < PPX; ++i) {
      const int x = min(xb + i, g.Wf - 1);
      const float c0x = div_c(2.f * (float)x, g.wd) - 1.f;
      m[i] = fbc_dense(fx[i], fy[i], o, o + HW, g, c0x, c0y) ? 1 : 0;
      if (xb + i < g.Wf) bad += 1 - m[i];
    }
    if (vec) *reinterpret_cast<uchar4*>(mask + r) = make_uchar4(m[0], m[1], m[2], m[3]);
    else for (int i = 0; i < PPX && xb + i < g.Wf; ++i) mask[r + i] = m[i];
  }
  if (invalid) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) bad += __shfl_xor(bad, s, 64);
    if (threadIdx.x == 0 && bad) atomicAdd(&invalid[dir * N + n], bad);
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
  g.sy = h > 1 ? (float)(h - 1) / (float)(8 * h - 1) : 0.f;
  g.sx = w > 1 ? (float)(w - 1) / (float)(8 * w - 1) : 0.f;
  g.wd = make_divc((float)(Wf - 1)); g.ht = make_divc((float)(Hf - 1));
  g.hx = (float)(Wf - 1) / 2.f; g.hy = (float)(Hf - 1) / 2.f;
  g.alpha1 = alpha1; g.a2 = a2;
  return g;
}

int run_flow_prepare(const float* lo_fwd, const float* lo_bwd, float* flow_fwd, float* flow_bwd, uint8_t* mask_fwd,
                     uint8_t* mask_bwd, int32_t* invalid, int N, int F1, int h, int w, int flow_up, float alpha1,
                     double alpha2, hipStream_t st) {
  const int Hf = flow_up ? 8 * h : h, Wf = flow_up ? 8 * w : w;
  const size_t n = (size_t)N * Hf * Wf;
  const float a2 = fbc_a2(alpha2, Hf, Wf);
  if (invalid) {
    if (hipMemsetAsync(invalid, 0, sizeof(int32_t) * 2 * N, st) != hipSuccess) {
      set_error("hipMemsetAsync failed");
      return PIXPRO_ERR_LAUNCH;
    }
  }
  if (F1 == 1) {
    if (flow_up && N <= 65535 && (Hf + PT_Y - 1) / PT_Y <= 65535) {
      Up8Geom g;
      g.h = h; g.w = w; g.Hf = Hf; g.Wf = Wf;
      g.sy = h > 1 ? (float)(h - 1) / (float)(Hf - 1) : 0.f;
      g.sx = w > 1 ? (float)(w - 1) / (float)(Wf - 1) : 0.f;
      g.wd = make_divc((float)(Wf - 1)); g.ht = make_divc((float)(Hf - 1));
      g.hx = (float)(Wf - 1) / 2.f; g.hy = (float)(Hf - 1) / 2.f;
      g.alpha1 = alpha1; g.a2 = a2;
      dim3 grid((Wf + PT_X * PPX - 1) / (PT_X * PPX), (Hf + PT_Y - 1) / PT_Y, N);
      hipLaunchKernelGGL(k_prepare_up8, grid, dim3(PT_X, PT_Y), 0, st, lo_fwd, lo_bwd, flow_fwd, flow_bwd, mask_fwd,
                         mask_bwd, invalid, N, g);
    } else if (flow_up) {
      Up8Source sf{lo_fwd, F1, h, w}, sb{lo_bwd, F1, h, w};
      hipLaunchKernelGGL(k_prepare_single<Up8Source>, dim3(blocks_for(n)), dim3(TPB), 0, st, sf, sb, flow_fwd,
                         flow_bwd, mask_fwd, mask_bwd, invalid, N, Hf, Wf, alpha1, a2);
    } else {
      LoSource sf{lo_fwd, F1, h, w}, sb{lo_bwd, F1, h, w};
      hipLaunchKernelGGL(k_prepare_single<LoSource>, dim3(blocks_for(n)), dim3(TPB), 0, st, sf, sb, flow_fwd,
                         flow_bwd, mask_fwd, mask_bwd, invalid, N, Hf, Wf, alpha1, a2);
    }
    CHECKED("k_prepare_single");
    return PIXPRO_OK;
  }
  // F-1 > 1: chains (up-sampling evaluated lazily, both directions in one launch), then both masks in one launch
  const bool grid_ok = 2 * N <= 65535 && (Hf + PT_Y - 1) / PT_Y <= 65535;
  if (grid_ok && flow_up && F1 <= CH_MAXF) {
    const Up8Geom g = make_geom(h, w, Hf, Wf, alpha1, a2);
    dim3 grid((Wf + PT_X * PPX - 1) / (PT_X * PPX), (Hf + PT_Y - 1) / PT_Y, 2 * N);
    hipLaunchKernelGGL(k_chain_up8, grid, dim3(PT_X, PT_Y), 0, st, lo_fwd, lo_bwd, flow_fwd, flow_bwd, N, F1, g);
    CHECKED("k_chain_up8");
    if (mask_fwd) {
      hipLaunchKernelGGL(k_fbc_pair, grid, dim3(PT_X, PT_Y), 0, st, flow_fwd, flow_bwd, mask_fwd, mask_bwd, invalid, N, g);
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
    hipLaunchKernelGGL(k_fbc, dim3(blocks_for(n)), dim3(TPB), 0, st, flow_fwd, flow_bwd, mask_fwd, (float*)nullptr,
                       invalid, N, Hf, Wf, alpha1, a2, 0);
    CHECKED("k_fbc");
    hipLaunchKernelGGL(k_fbc, dim3(blocks_for(n)), dim3(TPB), 0, st, flow_bwd, flow_fwd, mask_bwd, (float*)nullptr,
                       invalid ? invalid + N : nullptr, N, Hf, Wf, alpha1, a2, 0);
    CHECKED("k_fbc");
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
