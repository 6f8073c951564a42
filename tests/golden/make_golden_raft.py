"""Golden vectors for the RAFT pieces (build container only; needs /root/reference).

Imports the reference's contrast.flow package read-only (torch + scipy only) and records: the windowed correlation lookup
(CorrBlock) for two small cases, and whole-model inference (RAFT-small and RAFT, test_mode, 12 iterations) on one seeded
frame pair with seeded weights (tests/golden/recipes.py: raft_state_dict).  Only data is written.
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from recipes import raft_images, raft_state_dict, tag_seed  # noqa: E402
REF = "/root/reference"


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for mod in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    corr = importlib.import_module("contrast.flow.corr")
    raft = importlib.import_module("contrast.flow.raft")
    torch.set_num_threads(4)

    for tag, (B, D, h, w, levels, radius) in {"l3r2": (2, 32, 12, 12, 3, 2), "l4r3": (1, 64, 16, 16, 4, 3)}.items():
        g = torch.Generator().manual_seed(tag_seed("raftcorr" + tag))
        f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
        blk = corr.CorrBlock(f1, f2, num_levels=levels, radius=radius)
        from contrast.flow.utils.utils import coords_grid
        coords = coords_grid(B, h, w) + 2.5 * torch.randn(B, 2, h, w, generator=g)
        coords[:, :, 0, 0] = -30.0           # far outside: every tap is zero padding
        coords[:, :, 1, 1] = torch.tensor([w - 1.0, h - 1.0]).view(1, 2)   # exactly on the last pixel
        save(f"raft_corr_{tag}", fmap1=f1, fmap2=f2, coords=coords, out=blk(coords), levels=levels, radius=radius)

    for tag, small, hw in (("small", True, (128, 160)), ("basic", False, (128, 128))):
        args = argparse.Namespace(small=small, mixed_precision=False)
        model = raft.RAFT(args).eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(raft_state_dict(shapes, tag_seed("raft" + tag)))
        im1, im2 = raft_images(1, hw[0], hw[1], tag_seed("raftimg" + tag))
        with torch.no_grad():
            lo, up = model(im1, im2, iters=12, upsample=False, test_mode=True)
        print(tag, "low-res flow: max-abs", lo.abs().max().item(), "mean-abs", lo.abs().mean().item())
        save(f"raft_{tag}_e2e", flow_lo=lo, flow_up=up.half(), hw=np.asarray(hw),
             names=np.asarray(sorted(shapes)), shapes=np.asarray([str(shapes[k]) for k in sorted(shapes)]))


if __name__ == "__main__":
    main()
