#!/usr/bin/env python
"""Which torch operators launch the small element-wise kernels of a pre-training step (torch.profiler, one step after
warm-up): kernel name -> the operator that launched it, calls and time per step.  Usage: tools/op_profile.py [--batch 64]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    import bench
    import main_pretrain
    from contrast.lr_scheduler import get_scheduler
    from contrast.synthetic import SyntheticLoader
    argv, sys.argv = sys.argv, [sys.argv[0], "--batch", str(a.batch), "--steps", "1", "--warmup", "1"]
    ba = bench.parse()
    sys.argv = argv
    dev = torch.device("cuda", 0)
    args = bench.model_args(ba, 1)
    from contrast.miopen_tuning import use_shipped_db
    use_shipped_db()
    loader = SyntheticLoader(4, a.batch, pool=2, image_size=224, n_frames=2, frame_hw=(720, 1280), device=dev, seed=1234,
                             channels_last=True)
    model, optimizer = main_pretrain.build_model(args, dev)
    scheduler = get_scheduler(optimizer, 70000 // a.batch, args)
    model.train()
    it = iter(loader)
    for _ in range(2):
        main_pretrain.train_step(model, optimizer, scheduler, next(it), args)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        main_pretrain.train_step(model, optimizer, scheduler, next(it), args)
        torch.cuda.synchronize()
    ev = prof.events()
    # map every device kernel to the innermost CPU operator whose interval contains its launch
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in ev:
        if e.device_type.name != "CPU" or not e.kernels:
            continue
        for k in e.kernels:
            # (shapes only for the plain copies / adds: which tensors are they?)
            shp = str(e.input_shapes)[:60] if e.name in ("aten::copy_", "aten::add_", "aten::add") and k.duration > 20 else ""
            key = (k.name[:70], e.name + " " + shp)
            rows[key][0] += 1
            rows[key][1] += k.duration
    small = sorted(rows.items(), key=lambda kv: -kv[1][1])
    for (kn, op), (n, us) in small[:60]:
        if "igemm" in kn or "Cijk" in kn or "k_stats" in kn or "k_apply" in kn or "k_bwd" in kn:
            continue
        print(f"{us / 1e3:8.3f} ms  calls {n:4d}  {op:75s} {kn}")


if __name__ == "__main__":
    main()
