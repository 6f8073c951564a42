"""What does BatchNorm cost on the SyncBN route (N > 1 GPUs) next to the plain route (N = 1)?

torch.nn.SyncBatchNorm with world_size > 1 runs ATen's native kernels (batch_norm_stats, batch_norm_elemt,
batch_norm_backward_reduce, batch_norm_backward_elemt); with one rank it falls through to F.batch_norm = MIOpen.
This times both on the ResNet-50 BN shapes (channels_last, fp32) in ONE process, collectives left out.
usage: python tools/syncbn_microbench.py [N]
"""
import sys
import time

import torch
import torch.nn.functional as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda"
# (C, H) x count for one ResNet-50 forward at 224^2 (stem, then per stage: conv1/conv2/conv3 (+downsample))
SHAPES = [(64, 112, 1), (64, 56, 6), (256, 56, 4), (128, 56, 1), (128, 28, 7), (512, 28, 5), (256, 28, 1), (256, 14, 11),
          (1024, 14, 7), (512, 14, 1), (512, 7, 5), (2048, 7, 4)]


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


tot = {"miopen_fwd": 0., "native_fwd": 0., "miopen_bwd": 0., "native_bwd": 0.}
print(f"N={N}  ms per call                 miopen_fwd native_fwd miopen_bwd native_bwd   GB(fwd 3 passes)")
for C, H, cnt in SHAPES:
    x = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    count = torch.full((1,), x.numel() // C, device=dev, dtype=torch.float32)

    def miopen_fwd():
        return F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)

    def native_fwd():
        mean, invstd = torch.batch_norm_stats(x, 1e-5)
        mean, invstd = torch.batch_norm_gather_stats_with_counts(x, mean.unsqueeze(0), invstd.unsqueeze(0), rm, rv, 0.1,
                                                                 1e-5, count)
        return torch.batch_norm_elemt(x, w, b, mean, invstd, 1e-5), mean, invstd

    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = F.batch_norm(xr, rm, rv, wr, br, True, 0.1, 1e-5)

    def miopen_bwd():
        return torch.autograd.grad(y, (xr, wr, br), dy, retain_graph=True)

    _, mean, invstd = native_fwd()

    def native_bwd():
        sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(dy, x, mean, invstd, w, True, True, True)
        return torch.batch_norm_backward_elemt(dy, x, mean, invstd, w, sum_dy, sum_dy_xmu, count.int())

    r = [timed(miopen_fwd), timed(native_fwd), timed(miopen_bwd), timed(native_bwd)]
    for k, v in zip(tot, r):
        tot[k] += v * cnt
    print(f"C={C:5d} H={H:4d} x{cnt:2d}              {r[0]:9.3f} {r[1]:10.3f} {r[2]:10.3f} {r[3]:10.3f}   {3 * x.numel() * 4 / 1e9:.2f}")
print("one encoder pass, all 53 BN layers (ms):", {k: round(v, 2) for k, v in tot.items()})
print("per step (4 fwd + 2 bwd passes):  miopen %.1f ms   native(SyncBN) %.1f ms" %
      (4 * tot["miopen_fwd"] + 2 * tot["miopen_bwd"], 4 * tot["native_fwd"] + 2 * tot["native_bwd"]))
