#!/usr/bin/env python
"""Isolated timing of the hot-path launches at BASELINE config sizes (HIP events on the launch stream).
Used while tuning kernels; bench.py measures the same launches inside the real step."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))
import torch  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--grid", type=int, default=7)
    ap.add_argument("--n-frames", type=int, default=2)
    ap.add_argument("--what", default="flow,ppm,ppm_generic,ema,lars")
    ap.add_argument("--flow-offset", type=float, default=0.0, help="constant added to the forward flow's x component (full-res "
                                                                    "pixels; the backward flow gets the negative): coherent motion")
    ap.add_argument("--flow-step", type=float, default=0.0, help="motion boundary: this many pixels added right of the middle")
    ap.add_argument("--no-mask", action="store_true", help="flow preparation without the consistency masks (alpha None)")
    ap.add_argument("--lib", default=None, help="load this build of libpixpro_hip.so instead of the in-tree one (kernel variants)")
    ap.add_argument("--channels-last", action="store_true", help="feature maps in channels_last memory format (bench default)")
    a = ap.parse_args()
    import types
    if a.lib:
        from contrast import _native
        _native.LIB_PATH = os.path.abspath(a.lib)
    from contrast import ops, util
    from contrast.synthetic import make_batch
    dev = torch.device("cuda:0")
    N, G = a.batch, a.grid
    data = make_batch(N, image_size=8, n_frames=a.n_frames, device=dev)
    if a.flow_offset or a.flow_step:
        _, lo_f, lo_b = data[5]
        lo_f[:, :, 0] += a.flow_offset / 8.0; lo_b[:, :, 0] -= a.flow_offset / 8.0
        half = lo_f.shape[-1] // 2
        lo_f[:, :, 0, :, half:] += a.flow_step / 8.0; lo_b[:, :, 0, :, half:] -= a.flow_step / 8.0
    args = types.SimpleNamespace(alpha1=None if a.no_mask else 0.01, alpha2=None if a.no_mask else 0.5, use_flow_frames=False, use_flow_file=True, flow_up=True,
                                 flow_cat_norm=False, debug=False)
    A = 720 * 1280 * 4
    F1 = a.n_frames - 1
    res = {}
    what = a.what.split(",")
    ff, fb = util.apply_optical_flow(data, None, args)
    if "flow" in what:
        med, mn = timeit(lambda: util.apply_optical_flow(data, None, args), iters=10)
        fused = (F1 * A / 16 + 4.5 * A) * N
        res["flow_prepare"] = dict(ms=med, ms_min=mn, compulsory_GBps=fused / med / 1e6,
                                   unfused_equiv_GBps=(2 * F1 * 2 * A / 64 + 8 * F1 * A + 4.5 * A) * N / med / 1e6)
    C, P = 256, G * G
    gen = torch.Generator(device=dev).manual_seed(0)
    mf = torch.channels_last if a.channels_last else torch.contiguous_format
    t = lambda: torch.randn(N, C, G, G, device=dev, generator=gen).contiguous(memory_format=mf)
    x1, v1, x2, v2, k1, k2 = (t() for _ in range(6))
    x1.requires_grad_(True); v1.requires_grad_(True); x2.requires_grad_(True); v2.requires_grad_(True)
    size = (720, 1280)

    def run(generic):
        d1 = ops.Direction(x1, v1, k2, data[2], data[3], ff[0], ff[2], size)
        d2 = ops.Direction(x2, v2, k1, data[3], data[2], fb[0], fb[2], size)
        return ops.ppm_loss([d1, d2], 2.0, 0.0, 0.7, True, generic=generic)
    for name, gflag in (("ppm", False), ("ppm_generic", True)):
        if name in what:
            med, mn = timeit(lambda: run(gflag))
            res[name] = dict(ms=med, ms_min=mn, TFLOPs=24 * C * P * P * N / med / 1e9,
                             frac_mfma=24 * C * P * P * N / med / 1e9 / 157.3, GBps=10 * C * P * 4 * N / med / 1e6)
    print(res)
    for k, v in res.items():
        print(k, {a: round(b, 4) for a, b in v.items()})


if __name__ == "__main__":
    main()
