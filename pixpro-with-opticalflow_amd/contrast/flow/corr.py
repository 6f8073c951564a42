"""All-pairs correlation pyramid + windowed lookup of RAFT (reference: contrast/flow/corr.py:12-60).

The volume itself is one batched GEMM (hipBLASLt through torch.matmul) and three 2x2 average pools: with 288 GB of HBM the
[B*h*w, h, w] volume of a 720x1280 frame pair (829 MB + 1/3 for the pyramid) simply stays resident.  What the
reference does per refinement iteration - 4 levels x `grid_sample` over a (2r+1)^2 window per query pixel, i.e. 4
launches gathering through three temporaries - is ONE HIP launch here (`pixpro_corr_lookup`, csrc/raft_corr.hip): one
lane per (query, level, window tap), bilinear arithmetic of ATen's grid_sample (same fma order as the rest of this
library), output written directly in the [B, L*(2r+1)^2, h, w] layout the motion encoder consumes.
"""
import math

import torch
import torch.nn.functional as F

from .. import _native as nv


def build_pyramid(fmap1, fmap2, num_levels=4):
    """corr.py:17-27,52-60: corr[b, i, j] = <fmap1[b,:,i], fmap2[b,:,j]> / sqrt(dim), then 2x2 average pools over j."""
    B, D, H, W = fmap1.shape
    f1 = fmap1.reshape(B, D, H * W)
    f2 = fmap2.reshape(B, D, H * W)
    # the 1/sqrt(dim) of corr.py:60 rides on the GEMM's alpha (one pass over the 829 MB/image volume less); it scales the
    # fp32 accumulator, so the result differs from matmul-then-divide by at most the rounding of x*(1/s) against x/s
    alpha = 1.0 / float(torch.sqrt(torch.tensor(D).float()))
    corr = torch.baddbmm(torch.empty((1, 1, 1), dtype=f1.dtype, device=f1.device).expand(B, H * W, H * W),
                         f1.transpose(1, 2), f2, beta=0.0, alpha=alpha)
    level = corr.reshape(B * H * W, 1, H, W)
    pyramid = [level]
    for _ in range(num_levels - 1):
        level = F.avg_pool2d(level, 2, stride=2)
        pyramid.append(level)
    return pyramid


def lookup(pyramid, coords, radius):
    """corr.py:29-50 as one launch.  pyramid[l]: [B*h*w, 1, h_l, w_l]; coords [B,2,h,w] (x, y) in level-0 pixels.
    Returns [B, L*(2r+1)^2, h, w]; channel (l, i, j) samples level l at (x/2^l + d[i], y/2^l + d[j]), d = -r..r - the
    reference's window is transposed with respect to (dy, dx) because it stacks meshgrid(dy, dx) onto (x, y)."""
    if len(pyramid) < 1 or len(pyramid) > 4:
        raise RuntimeError("corr lookup supports 1..4 pyramid levels")
    B, two, h, w = coords.shape
    coords = nv.dev_f32(coords, "coords", (B, 2, h, w))
    levels = [nv.dev_f32(p, f"pyramid[{i}]") for i, p in enumerate(pyramid)]
    h2, w2 = int(levels[0].shape[-2]), int(levels[0].shape[-1])
    hh, ww = h2, w2
    for i, p in enumerate(levels):
        if tuple(p.shape) != (B * h * w, 1, hh, ww):
            raise RuntimeError(f"pyramid[{i}]: expected {(B * h * w, 1, hh, ww)}, got {tuple(p.shape)}")
        hh, ww = hh // 2, ww // 2
    win = 2 * radius + 1
    out = torch.empty((B, len(levels) * win * win, h, w), dtype=torch.float32, device=coords.device)
    ptrs = [nv.ptr(p) for p in levels] + [None] * (4 - len(levels))
    nv.call("pixpro_corr_lookup", *ptrs, len(levels), nv.ptr(coords), nv.ptr(out), B, h, w, h2, w2, int(radius), nv.stream())
    return out


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.corr_pyramid = build_pyramid(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        return lookup(self.corr_pyramid, coords, self.radius)
