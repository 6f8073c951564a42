"""Work-group counts of the fused BN kernels (PIXPRO_BN_CAPS) swept over the ResNet-50 layer shapes at N=256: one child
process per setting (the library reads the variable once), per-step totals (4 forward + 2 backward encoder passes).
usage: python tools/bn_sweep.py [--lib exp_libs/libX.so] [--caps 256,512,1024,512 ...]   (parent; child = one setting)"""
import itertools
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))
SHAPES = [(64, 112, 1, 0), (64, 56, 6, 0), (256, 56, 4, 3), (128, 56, 1, 0), (128, 28, 7, 0), (512, 28, 5, 4), (256, 28, 1, 0),
          (256, 14, 11, 0), (1024, 14, 7, 6), (512, 14, 1, 0), (512, 7, 5, 0), (2048, 7, 4, 3)]   # C, H, count, of which residual


def child():
    import torch
    if os.environ.get("BN_SWEEP_LIB"):          # a kernel-variant build of the library (exp_libs/), see --lib
        from contrast import _native
        _native.LIB_PATH = os.path.abspath(os.environ["BN_SWEEP_LIB"])
    from contrast.sync_bn import FusedSyncBatchNorm
    dev, N = "cuda", 256
    tot = {"fwd": 0.0, "bwd": 0.0}
    for C, H, cnt, nres in SHAPES:
        x = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        bn = FusedSyncBatchNorm(C).to(dev).train()
        for res in ([False, True] if nres else [False]):
            k = nres if res else cnt - nres
            r = torch.randn_like(x) if res else None
            xg = x.clone().requires_grad_(True)

            def fwd():
                return bn(xg, r, True)

            y = fwd()
            dy = torch.randn_like(y)

            def bwd():
                return torch.autograd.grad(y, xg, dy, retain_graph=True)

            for name, fn in (("fwd", fwd), ("bwd", bwd)):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                tot[name] += (time.perf_counter() - t0) / 5 * 1e3 * k
    print(f"{os.path.basename(os.environ.get('BN_SWEEP_LIB', 'in-tree')):>20s} {os.environ.get('PIXPRO_BN_CAPS', 'default'):>20s}  fwd/pass {tot['fwd']:7.2f}  bwd/pass {tot['bwd']:7.2f}  "
          f"per step {4 * tot['fwd'] + 2 * tot['bwd']:7.2f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        # --lib PATH: that build instead of the in-tree one; --caps a,b,c,d [a,b,c,d ...]: these settings only
        if "--lib" in sys.argv:
            os.environ["BN_SWEEP_LIB"] = sys.argv[sys.argv.index("--lib") + 1]
        only = None
        if "--caps" in sys.argv:
            only = [tuple(int(v) for v in a.split(",")) for a in sys.argv[sys.argv.index("--caps") + 1:] if "," in a]
        for caps in only or [(256, 512, 1024, 512), (192, 512, 1024, 512), (384, 512, 1024, 512), (256, 384, 1024, 512),
                     (256, 768, 1024, 512), (256, 512, 768, 512), (256, 512, 1024, 384), (256, 512, 1024, 768),
                     (320, 640, 1024, 640), (256, 512, 512, 512)]:
            env = dict(os.environ, PIXPRO_BN_CAPS=",".join(map(str, caps)))
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
