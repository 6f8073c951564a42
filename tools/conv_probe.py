"""Inclusive time of every distinct encoder convolution of the step (forward, input gradient, weight gradient), HIP events
around the aten call - i.e. INCLUDING whatever the chosen MIOpen solver launches beside its main kernel (the split-K solvers
zero their output first with SubTensorOpWithScalar1d, which runs at ~1.3 TB/s).  Run once per solver setting (MIOpen reads its
debug switches once per process) and compare:

    python tools/conv_probe.py --tag default
    MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 \
        python tools/conv_probe.py --tag no_asm_fwd_bwd
    python tools/conv_probe.py --compare A.json B.json

Shapes come from the model itself (hooks on one forward pass); --batch is the convolution batch (256 per view)."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))


def problems(arch, image):
    import torch
    import torch.nn as nn
    from contrast import resnet
    net = getattr(resnet, arch)(head_type="early_return")
    seen = {}

    def hook(m, inp, out):
        key = (inp[0].shape[1], inp[0].shape[2], inp[0].shape[3], m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0],
               m.dilation[0], m.groups)
        seen[key] = seen.get(key, 0) + 1
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.register_forward_hook(hook)
    with torch.no_grad():
        net.eval()(torch.zeros(1, 3, image, image))
    return seen


def run(a):
    import torch
    from contrast import miopen_tuning
    miopen_tuning.use_shipped_db()
    dev = "cuda"
    out = {}
    for key, count in problems(a.arch, a.image_size).items():
        if a.only and "x".join(map(str, key)) not in a.only:
            continue
        cin, h, w, cout, k, stride, pad, dil, groups = key
        x = torch.randn(a.batch, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(cout, cin // groups, k, k, device=dev).contiguous(memory_format=torch.channels_last)
        y = torch.nn.functional.conv2d(x, wt, None, stride, pad, dil, groups)
        dy = torch.randn_like(y)

        def fwd():
            return torch.nn.functional.conv2d(x, wt, None, stride, pad, dil, groups)

        def bwd(mask):
            return torch.ops.aten.convolution_backward(dy, x, wt, None, [stride, stride], [pad, pad], [dil, dil], False, [0, 0],
                                                       groups, mask)
        rec = {}
        for name, fn in (("fwd", fwd), ("dgrad", lambda: bwd([True, False, False])), ("wgrad", lambda: bwd([False, True, False]))):
            if name == "dgrad" and cin == 3:
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            rec[name] = statistics.median(ts)
        out["x".join(map(str, key))] = {"count": count, **rec}
        print(key, count, {k2: round(v, 4) for k2, v in rec.items()}, flush=True)
    tot = {d: sum(v.get(d, 0.0) * v["count"] for v in out.values()) for d in ("fwd", "dgrad", "wgrad")}
    print("per encoder pass (ms):", {k2: round(v, 2) for k2, v in tot.items()},
          "per step (4 fwd + 2 bwd):", round(4 * tot["fwd"] + 2 * (tot["dgrad"] + tot["wgrad"]), 2))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


def compare(pa, pb):
    A, B = json.load(open(pa)), json.load(open(pb))
    gain = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for key in A:
        for d in ("fwd", "dgrad", "wgrad"):
            if d in A[key] and d in B.get(key, {}):
                ta, tb, n = A[key][d], B[key][d], A[key]["count"]
                mark = "  <-- B better" if tb < 0.97 * ta else ""
                print(f"{key:36s} {d:6s} x{n:2d}  A {ta:7.3f}  B {tb:7.3f} ms{mark}")
                if tb < ta:
                    gain[d] += (ta - tb) * n
    print("ms per encoder pass to gain by taking the better one per problem:", {k: round(v, 3) for k, v in gain.items()},
          "per step:", round(4 * gain["fwd"] + 2 * (gain["dgrad"] + gain["wgrad"]), 2))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--tag", default="default")
    ap.add_argument("--out", default=None)
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--only", nargs="*", default=None, help="problem keys (as printed) to time")
    a = ap.parse_args()
    if a.compare:
        compare(*a.compare)
    else:
        a.out = a.out or os.path.join(ROOT, "gpurun_out", "conv_probe", f"{a.tag}.json")
        run(a)
