"""RAFT(-small) inference at the production frame size on one MI355X: time per frame pair, and the windowed correlation
lookup as one HIP launch (contrast/flow/corr.py) against the reference's formulation in torch ops (4 x grid_sample +
temporaries per iteration, restated below with torch ops on the device).  usage: python tools/raft_bench.py [batch] [--basic]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pixpro-with-opticalflow_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from contrast.miopen_tuning import use_shipped_db  # noqa: E402
use_shipped_db()
from contrast.flow import RAFT, corr  # noqa: E402


def torch_ops_lookup(pyramid, coords, radius):
    """The reference's per-iteration formulation (contrast/flow/corr.py:29-50) in torch ops on the device: per level
    linspace/meshgrid/add, one grid_sample, then view/cat/permute/contiguous."""
    import torch.nn.functional as F
    r = radius
    coords = coords.permute(0, 2, 3, 1)
    B, h1, w1, _ = coords.shape
    out = []
    for i, vol in enumerate(pyramid):
        d = torch.linspace(-r, r, 2 * r + 1, device=coords.device)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
        lvl = coords.reshape(B * h1 * w1, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        H, W = vol.shape[-2:]
        xg, yg = lvl.split([1, 1], dim=-1)
        grid = torch.cat([2 * xg / (W - 1) - 1, 2 * yg / (H - 1) - 1], dim=-1)
        out.append(F.grid_sample(vol, grid, align_corners=True).view(B, h1, w1, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()

from recipes import raft_images, raft_state_dict  # noqa: E402


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
    small = "--basic" not in sys.argv
    dev = torch.device("cuda:0")
    model = RAFT(argparse.Namespace(small=small)).eval()
    model.load_state_dict(raft_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 1))
    model = model.to(dev)
    H, W = 720, 1280
    im1, im2 = raft_images(B, H, W, 3)
    im1, im2 = im1.to(dev), im2.to(dev)
    with torch.no_grad():
        t_model = timed(lambda: model(im1, im2, iters=12, test_mode=True))
        fmap1, fmap2 = model.fnet([2 * (im1 / 255.0) - 1.0, 2 * (im2 / 255.0) - 1.0])
        r = model.args.corr_radius
        t_pyr = timed(lambda: corr.build_pyramid(fmap1, fmap2))
        pyr = corr.build_pyramid(fmap1, fmap2)
        coords = torch.stack(torch.meshgrid(torch.arange(W // 8, device=dev), torch.arange(H // 8, device=dev), indexing="xy"), 0)
        coords = coords.float()[None].repeat(B, 1, 1, 1) + 1.7
        t_hip = timed(lambda: corr.lookup(pyr, coords, r), 10)
        t_ref = timed(lambda: torch_ops_lookup(pyr, coords, r), 10)
        diff = (corr.lookup(pyr, coords, r) - torch_ops_lookup(pyr, coords, r)).abs().max().item()
    vol_gb = sum(p.numel() for p in pyr) * 4 / 1e9
    out_bytes = B * 4 * (2 * r + 1) ** 2 * (H // 8) * (W // 8) * 4
    print(f"RAFT{'-small' if small else ''} {H}x{W}, batch {B}, 12 iterations, fp32")
    print(f"  whole model        {t_model:9.2f} ms = {t_model / B:.2f} ms per frame pair")
    print(f"  correlation pyramid {t_pyr:8.2f} ms ({vol_gb:.2f} GB resident)")
    print(f"  lookup, HIP launch  {t_hip:8.3f} ms per iteration ({out_bytes / t_hip / 1e6:.0f} GB/s of output)")
    print(f"  lookup, torch ops   {t_ref:8.3f} ms per iteration (reference formulation on the device; max abs difference {diff:.1e}: "
          f"ATen's device grid_sample orders its fma differently from the CPU path the HIP kernel reproduces)")


if __name__ == "__main__":
    main()
