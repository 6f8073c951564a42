"""Per-shape rates of the fused BatchNorm kernels over the ResNet-50 layer shapes at N=256 (HIP events around each C call,
median of 7): which shapes pull the per-step averages down?   usage: python tools/bn_shapes.py [--lib exp_libs/libX.so]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))
import torch
from contrast import _native
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from contrast.sync_bn import FusedSyncBatchNorm

SHAPES = [(64, 112, 1, 0), (64, 56, 6, 0), (256, 56, 4, 3), (128, 56, 1, 0), (128, 28, 7, 0), (512, 28, 5, 4), (256, 28, 1, 0),
          (256, 14, 11, 0), (1024, 14, 7, 6), (512, 14, 1, 0), (512, 7, 5, 0), (2048, 7, 4, 3)]   # C, H, count, of which residual
N = 256
rec = {}
real = _native.call


def call(name, *a):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = real(name, *a)
    e.record()
    rec.setdefault(name.replace("pixpro_bn_", "").replace("_bits", ""), []).append((s, e))
    return r


_native.call = call
tot = {}
print(f"{'C':>5} {'H':>4} {'res':>3} {'MB':>7} | " + " | ".join(f"{k:>22}" for k in ("stats", "apply", "backward_reduce", "backward_apply")))
for C, H, cnt, nres in SHAPES:
    x = torch.randn(N, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    bn = FusedSyncBatchNorm(C).cuda().train()
    # res = 2: the residual wants no gradient tensor - what the step runs behind a ResidualLink (k_bwd_apply<3, false>)
    for res in ([0, 1, 2] if nres else [0]):
        r = (torch.randn_like(x).requires_grad_(res == 1)) if res else None
        xg = x.clone().requires_grad_(True)
        rec.clear()
        for _ in range(7):
            y = bn(xg, r, True)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        mb = x.numel() * 4 / 1e6
        passes = {"stats": 1, "apply": 2 + bool(res), "backward_reduce": 2 + (bool(res) and not C % 256 == 0),
                  "backward_apply": 3 + (res == 1)}
        cells = []
        for k in ("stats", "apply", "backward_reduce", "backward_apply"):
            ms = statistics.median(s.elapsed_time(e) for s, e in rec[k])
            cells.append(f"{ms * 1e3:8.1f} us {passes[k] * mb / ms / 1e3:6.2f} TB/s")
            n = (nres if res else cnt - nres) * (4 if k in ("stats", "apply") else 2)
            if res != 1:                          # the per-step total counts the linked form of the residual layers
                tot[k] = tot.get(k, 0.0) + ms * n
        print(f"{C:>5} {H:>4} {int(res):>3} {mb:7.1f} | " + " | ".join(cells), flush=True)
print("per step (4 forward + 2 backward passes), ms:", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))
