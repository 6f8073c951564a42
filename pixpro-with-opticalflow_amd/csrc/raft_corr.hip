// RAFT correlation lookup (reference contrast/flow/corr.py:29-50, utils/utils.py:66-78): for every query pixel of frame 1
// and every pyramid level, the (2r+1)^2 window of bilinear samples of that query's correlation map around the current
// match position.  The reference does, per refinement iteration and level, linspace/meshgrid/add (3 temporaries) + one
// grid_sample launch + view/cat/permute/contiguous; here all levels and taps are one launch that writes the
// [B, L*(2r+1)^2, h, w] tensor the motion encoder reads.  One lane per output element with x fastest (coalesced 4 B
// stores); its four taps are a gather from the query's own [h_l, w_l] map.  Arithmetic = ATen's grid_sample
// (bilinear, zeros padding, align_corners=True) through the same helpers as the flow kernels: bit-identical to the
// reference's CPU path on the same volume.
#include "common.h"

namespace pixpro {
namespace {

struct CorrLevels {
  const float* p[4];
  int h[4], w[4];
};

__global__ void __launch_bounds__(256)
k_corr_lookup(CorrLevels lv, int L, const float* __restrict__ coords, float* __restrict__ out, int B, int h, int w, int r) {
  const int win = 2 * r + 1, taps = win * win;
  const size_t hw = (size_t)h * w;
  const size_t total = (size_t)B * L * taps * hw;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % w), y = (int)((idx / w) % h);
  const int c = (int)((idx / hw) % ((size_t)L * taps));
  const int b = (int)(idx / (hw * L * taps));
  const int l = c / taps, i = (c % taps) / win, j = c % win;
  const size_t q = (size_t)b * hw + (size_t)y * w + x;
  const float cx = coords[((size_t)b * 2) * hw + (size_t)y * w + x];
  const float cy = coords[((size_t)b * 2 + 1) * hw + (size_t)y * w + x];
  const float scale = (float)(1 << l);
  // window index i offsets x, j offsets y: the reference adds meshgrid(dy, dx) to (x, y)   (corr.py:37-43)
  const float sx = cx / scale + (float)(i - r);
  const float sy = cy / scale + (float)(j - r);
  const int hl = lv.h[l], wl = lv.w[l];
  const float gx = 2.f * sx / (float)(wl - 1) - 1.f;        // utils.py:69-70
  const float gy = 2.f * sy / (float)(hl - 1) - 1.f;
  const BilinGeom g = bilin_geom(hl, wl, gx, gy);
  const float* m = lv.p[l] + q * ((size_t)hl * wl);
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (g.in & 1) v0 = m[g.y0 * wl + g.x0];
  if (g.in & 2) v1 = m[g.y0 * wl + g.x0 + 1];
  if (g.in & 4) v2 = m[(g.y0 + 1) * wl + g.x0];
  if (g.in & 8) v3 = m[(g.y0 + 1) * wl + g.x0 + 1];
  out[idx] = bilin_combine(g, v0, v1, v2, v3);
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
  const size_t total = (size_t)B * L * win * win * h * w;
  const size_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffULL) { set_error("corr lookup: too many elements"); return PIXPRO_ERR_ARG; }
  hipLaunchKernelGGL(k_corr_lookup, dim3((unsigned)blocks), dim3(256), 0, st, lv, L, coords, out, B, h, w, radius);
  return check_launch("k_corr_lookup");
}

}  // namespace pixpro
