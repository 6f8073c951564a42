#!/usr/bin/env python
"""Diagnostic: per-phase s_memtime stamps of ppm_fused_kernel (block 0, wave 0).  Builds a separate
library with -DPIXPRO_STAMPS (never the shipped one) and prints the cycle share of each phase."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pixpro-with-opticalflow_amd")
sys.path.insert(0, PKG)
torch = None

NAMES_NHWC = ["start", "wait x (top barrier)", "v dma, pts, S gemm", "barrier, k dma", "rx, maskbits", "S epilogue", "v wait, rv norms",
              "Y gemm + edge", "ny, rk", "G gemm + edge", "loss reduce", "dY, barrier", "dA gemm + epi", "dV gemm + edge",
              "dV reduce", "x dma, dv epi, barrier", "dv store, symm", "x wait, barrier", "dX gemm + edge", "dX reduce",
              "next-x dma, dx epi, barrier", "dx store"]
NAMES = ["start", "load x, tap issue", "S gemm, pts, v->LDS", "rx, maskbits", "rv norms, S epi", "Y gemm", "ny, rk, Yhat", "G gemm", "loss",
         "dY", "dA gemm+epi", "symm", "dV gemm+epi", "store dv", "dX gemm+epi", "store dx"]


def _has_symbol(lib, name):
    try:
        getattr(lib, name)
        return True
    except AttributeError:
        return False


def build(out):
    srcs = [os.path.join(PKG, "csrc", f) for f in ("capi.hip", "ppm_fused.hip", "ppm_fused_nhwc.hip", "ppm_generic.hip", "ppm_tiled.hip", "flow_prep.hip",
                                                   "multi_tensor.hip", "sync_bn.hip", "raft_corr.hip")]
    objs = []
    base = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
            "-DPIXPRO_STAMPS", "-fgpu-rdc"]
    for src in srcs:
        o = "/tmp/stamps_" + os.path.basename(src) + ".o"
        subprocess.run(base + ["-c", src, "-o", o], check=True)
        objs.append(o)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fgpu-rdc", "--hip-link", "-shared", "-fPIC", "-o", out]
                   + objs, check=True)


def main():
    # --build-only PATH: compile the stamped library (no GPU needed; e.g. into exp_libs/, which travels with gpurun);
    # --lib PATH: use a prebuilt one instead of compiling on the GPU box
    if "--build-only" in sys.argv:
        out = sys.argv[sys.argv.index("--build-only") + 1]
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        build(out)
        return
    if "--lib" in sys.argv:
        out = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    else:
        out = "/tmp/libpixpro_hip_stamps.so"
        build(out)
    global torch
    import torch
    from contrast import _native
    _native.LIB_PATH = out
    from contrast import ops
    dev = torch.device("cuda:0")
    N, C, G = 256, 256, 7
    gen = torch.Generator(device=dev).manual_seed(0)
    nhwc = "--nhwc" in sys.argv
    if "--batch" in sys.argv:
        N = int(sys.argv[sys.argv.index("--batch") + 1])
    mf = torch.channels_last if nhwc else torch.contiguous_format
    t = lambda: torch.randn(N, C, G, G, device=dev, generator=gen).contiguous(memory_format=mf)
    x1, v1, x2, v2, k1, k2 = (t() for _ in range(6))
    for z in (x1, v1, x2, v2):
        z.requires_grad_(True)
    from contrast.synthetic import crop_coords
    import random
    c1 = crop_coords(N, 1280, 720, random.Random(0)).to(dev)
    c2 = crop_coords(N, 1280, 720, random.Random(1)).to(dev)
    flow = torch.randn(N, 2, 90, 160, device=dev, generator=gen)
    mask = torch.rand(N, 90, 160, device=dev, generator=gen) > 0.1
    for _ in range(3):
        d1 = ops.Direction(x1, v1, k2, c1, c2, flow, mask, (720, 1280))
        d2 = ops.Direction(x2, v2, k1, c2, c1, flow, mask, (720, 1280))
        ops.ppm_loss([d1, d2], 2.0, 0.0, 0.7, True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert _native.load().pixpro_debug_stamps(buf) == 0
    if nhwc:
        first, last, t_start = list(buf)[0:22], list(buf)[32:54], list(buf)[30]
        print(f"NHWC kernel, last work-group (N={N}): kernel entry -> first item {first[0] - t_start} ticks, first item "
              f"{first[-1] - first[0]}, last item {last[-1] - last[0]}, entry -> end of last item {last[-1] - t_start}")
        print(f"{'phase':>28s} {'first':>8s} {'last':>8s}")
        for i in range(1, 22):
            print(f"{NAMES_NHWC[i]:>28s} {first[i] - first[i - 1]:8d} {last[i] - last[i - 1]:8d} "
                  f"{100.0 * (last[i] - last[i - 1]) / (last[-1] - last[0]):5.1f}%")
        for name, base in (("first", 0), ("last", 32)):
            b_ = list(buf)[base:base + 32]
            print(f"{name} item, S phase from the top barrier (stamp 1): wave 0 crop records + centres {b_[22] - b_[1]}, taps requested "
                  f"{b_[23] - b_[1]}, threshold {b_[24] - b_[1]}, points done {b_[2] - b_[1]}; wave 1 tile done {b_[27] - b_[1]}; "
                  f"wave 7 centres + threshold from {b_[28] - b_[1]} to {b_[29] - b_[1]}")
        lib = _native.load()
        if hasattr(lib, "pixpro_debug_wave_stamps") or _has_symbol(lib, "pixpro_debug_wave_stamps"):
            # experimental builds only: every wave's clock at 16 events of the last item (when each wave ARRIVES somewhere
            # tells which wave a barrier waits for)
            wb = (ctypes.c_ulonglong * 128)()
            assert lib.pixpro_debug_wave_stamps(wb) == 0
            ev = ["top barrier passed", "S phase done (arrival)", "S epilogue done", "v norms published", "Y gemm done (arrival)",
                  "Y norms published", "G gemm done (arrival)", "loss reduced", "dY published", "dA done (arrival)",
                  "dV gemm done (arrival)", "dV reduced", "dv epilogue done", "x landed, dSs published", "dX gemm done (arrival)",
                  "dX reduced"]
            t0 = min(wb[0:8])
            print("per-wave ticks since the top barrier, last item:")
            print(f"{'event':>28s} " + " ".join(f"{'w' + str(k):>7s}" for k in range(8)))
            for e in range(16):
                print(f"{ev[e]:>28s} " + " ".join(f"{wb[8 * e + k] - t0:7d}" for k in range(8)))
        return
    st = list(buf)[:16]
    tot = st[15] - st[0]
    print(f"total ticks {tot} (s_memtime, 100 MHz const clock? -> {tot / 100:.1f} us if so)")
    for i in range(1, 16):
        print(f"{NAMES[i]:>16s} {st[i] - st[i - 1]:8d} {100.0 * (st[i] - st[i - 1]) / tot:5.1f}%")
    extra = list(buf)[16:24]
    print("extra stamps relative to start:", [e - st[0] if e else None for e in extra])


if __name__ == "__main__":
    main()
