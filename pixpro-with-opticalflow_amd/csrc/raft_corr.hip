// RAFT correlation lookup (reference contrast/flow/corr.py:29-50, utils/utils.py:66-78): for every query pixel of frame 1
// and every pyramid level, the (2r+1)^2 window of bilinear samples of that query's correlation map around the current
// match position.  The reference does, per refinement iteration and level, linspace/meshgrid/add (3 temporaries) + one
// grid_sample launch + view/cat/permute/contiguous; here all levels and taps are one launch that writes the
// [B, L*(2r+1)^2, h, w] tensor the motion encoder reads.  Arithmetic = ATen's grid_sample
// (bilinear, zeros padding, align_corners=True) through the same helpers as the flow kernels: bit-identical to the
// reference's CPU path on the same volume.
#include "common.h"

namespace pixpro {
namespace {

struct CorrLevels {
  const float* p[4];
  int h[4], w[4];
};

// One sample of the window: level-l map `m` of one query, tap (i, j), match position (cx, cy) in level-0 pixels.
__device__ __forceinline__ float corr_tap(const float* __restrict__ m, int hl, int wl, float cx, float cy, float scale, int i,
                                          int j, int r) {
  // window index i offsets x, j offsets y: the reference adds meshgrid(dy, dx) to (x, y)   (corr.py:37-43)
  const float sx = cx / scale + (float)(i - r);
  const float sy = cy / scale + (float)(j - r);
  const float gx = 2.f * sx / (float)(wl - 1) - 1.f;        // utils.py:69-70
  const float gy = 2.f * sy / (float)(hl - 1) - 1.f;
  const BilinGeom g = bilin_geom(hl, wl, gx, gy);
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (g.in & 1) v0 = m[g.y0 * wl + g.x0];
  if (g.in & 2) v1 = m[g.y0 * wl + g.x0 + 1];
  if (g.in & 4) v2 = m[(g.y0 + 1) * wl + g.x0];
  if (g.in & 8) v3 = m[(g.y0 + 1) * wl + g.x0 + 1];
  return bilin_combine(g, v0, v1, v2, v3);
}

// Block = 64 consecutive queries of one image row, one pyramid level.  Gather phase: a wave takes one query at a time with
// its lanes across the window taps, so the 4 x (2r+1)^2 reads of a query fall into the few cache lines of its own
// (2r+2)^2 neighbourhood (a lane-per-output mapping touches one line per lane: every query owns a separate map).
// Store phase: through LDS, 64 consecutive x per tap -> 256 B coalesced stores into the [B, L*taps, h, w] tensor.
constexpr int CQ = 64;          // queries per block
__global__ void __launch_bounds__(256)
k_corr_lookup(CorrLevels lv, int L, const float* __restrict__ coords, float* __restrict__ out, int B, int h, int w, int r) {
  extern __shared__ float tile[];            // [taps][CQ + 1]
  const int win = 2 * r + 1, taps = win * win;
  const size_t hw = (size_t)h * w;
  const int x0 = blockIdx.x * CQ, y = blockIdx.y, b = blockIdx.z / L, l = blockIdx.z % L;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hl = lv.h[l], wl = lv.w[l];
  const float scale = (float)(1 << l);
  const float* cxp = coords + ((size_t)b * 2) * hw + (size_t)y * w;
  const float* cyp = cxp + hw;
  const size_t q0 = (size_t)b * hw + (size_t)y * w;
  for (int k = wave; k < CQ; k += 4) {
    const int x = x0 + k;
    if (x >= w) break;
    const float cx = cxp[x], cy = cyp[x];
    const float* m = lv.p[l] + (q0 + x) * ((size_t)hl * wl);
    for (int t = lane; t < taps; t += 64) tile[t * (CQ + 1) + k] = corr_tap(m, hl, wl, cx, cy, scale, t / win, t % win, r);
  }
  __syncthreads();
  const int x = x0 + lane;
  if (x < w)
    for (int t = wave; t < taps; t += 4)
      out[(((size_t)b * L + l) * taps + t) * hw + (size_t)y * w + x] = tile[t * (CQ + 1) + lane];
}

}  // namespace

int run_corr_lookup(const float* const* levels, int L, const float* coords, float* out, int B, int h, int w, int h2, int w2,
                    int radius, hipStream_t st) {
  CorrLevels lv;
  int hh = h2, ww = w2;
  for (int l = 0; l < 4; ++l) {
    lv.p[l] = l < L ? levels[l] : nullptr;
    lv.h[l] = hh; lv.w[l] = ww;
    hh /= 2; ww /= 2;
  }
  const int win = 2 * radius + 1;
  const size_t lds = (size_t)win * win * (CQ + 1) * sizeof(float);
  if (h > 65535 || (size_t)B * L > 65535) { set_error("corr lookup: grid too large"); return PIXPRO_ERR_ARG; }
  hipLaunchKernelGGL(k_corr_lookup, dim3((w + CQ - 1) / CQ, h, B * L), dim3(256), lds, st, lv, L, coords, out, B, h, w,
                     radius);
  return check_launch("k_corr_lookup");
}

}  // namespace pixpro
