"""TEST INFRASTRUCTURE ONLY - CPU restatement of the RAFT pieces that run as HIP kernels in the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  Pinned by golden vectors captured
from the reference's own contrast/flow package (tests/golden/make_golden_raft.py): `raft_corr_*.npz` for the lookup,
`raft_small_e2e.npz` / `raft_basic_e2e.npz` for whole-model inference with seeded weights.
"""
import torch
import torch.nn.functional as F


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """contrast/flow/corr.py:17-27 + :52-60."""
    B, D, H, W = fmap1.shape
    corr = torch.matmul(fmap1.view(B, D, H * W).transpose(1, 2), fmap2.view(B, D, H * W))
    corr = (corr.view(B, H, W, 1, H, W) / torch.sqrt(torch.tensor(D).float())).reshape(B * H * W, 1, H, W)
    pyr = [corr]
    for _ in range(num_levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def corr_lookup(pyramid, coords, radius):
    """contrast/flow/corr.py:29-50 with bilinear_sampler of contrast/flow/utils/utils.py:66-78."""
    r = radius
    coords = coords.permute(0, 2, 3, 1)
    B, h1, w1, _ = coords.shape
    out = []
    for i, corr in enumerate(pyramid):
        d = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).to(coords.device)   # :37-39: (dy, dx) stacked
        lvl = coords.reshape(B * h1 * w1, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)   # :41-43
        H, W = corr.shape[-2:]
        xg, yg = lvl.split([1, 1], dim=-1)
        grid = torch.cat([2 * xg / (W - 1) - 1, 2 * yg / (H - 1) - 1], dim=-1)           # utils.py:69-72
        out.append(F.grid_sample(corr, grid, align_corners=True).view(B, h1, w1, -1))    # :73, corr.py:46
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


class CorrBlockCPU:
    """Stand-in for contrast.flow.corr.CorrBlock so that the product's RAFT module can be evaluated on the host."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.radius = radius
        self.corr_pyramid = corr_pyramid(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        return corr_lookup(self.corr_pyramid, coords, self.radius)


def upflow8_cpu(flow, mode="bilinear"):
    """contrast/flow/utils/utils.py:87-89."""
    return 8 * F.interpolate(flow, size=(8 * flow.shape[2], 8 * flow.shape[3]), mode=mode, align_corners=True)
