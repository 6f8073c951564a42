"""Workload + parser for the HBM traffic of the fused BN kernels (rocprofv3 PMC).

  workload:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/bn_pmc.py run
             (and again with --pmc WRITE_SIZE): every BatchNorm call of ONE ResNet-50 encoder pass at N=256
             (forward, then backward), each exactly once.
  parser:    python tools/bn_pmc.py parse OUT_FETCH OUT_WRITE  -> JSON with KiB per encoder pass and per step
             (4 forward + 2 backward passes), FETCH_SIZE doubled for these 16 B/lane streaming reads (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixpro-with-opticalflow_amd"))
SHAPES = [(64, 112, 1, 0), (64, 56, 6, 0), (256, 56, 4, 3), (128, 56, 1, 0), (128, 28, 7, 0), (512, 28, 5, 4), (256, 28, 1, 0),
          (256, 14, 11, 0), (1024, 14, 7, 6), (512, 14, 1, 0), (512, 7, 5, 0), (2048, 7, 4, 3)]   # C, H, count, of which residual


def run():
    import torch
    from contrast.sync_bn import FusedSyncBatchNorm
    dev, N = "cuda", 256
    for C, H, cnt, nres in SHAPES:
        x = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        bn = FusedSyncBatchNorm(C).to(dev).train()
        r = torch.randn_like(x) if nres else None
        dy = torch.randn_like(x)
        for k in range(cnt):
            res = r if k < nres else None
            xg = x.clone().requires_grad_(True)
            y = bn(xg, res, k != cnt - 1 or C != 64 or H != 112 or True)      # every BN of the encoder is followed by a ReLU
            torch.autograd.grad(y, xg, dy)
    torch.cuda.synchronize()


def total(outdir, counter, needle):
    tot = 0.0
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter and "pixpro" in row["Kernel_Name"] and needle(row["Kernel_Name"]):
                tot += float(row["Counter_Value"])
    return tot


def parse(fetch_dir, write_dir):
    fwd = lambda n: "k_stats" in n or "k_apply" in n
    bwd = lambda n: "k_bwd" in n
    out = {}
    for name, sel in (("forward_pass", fwd), ("backward_pass", bwd)):
        out[name] = {"fetch_kib_x2": 2 * total(fetch_dir, "FETCH_SIZE", sel), "write_kib": total(write_dir, "WRITE_SIZE", sel)}
    step = 1024 * (4 * (out["forward_pass"]["fetch_kib_x2"] + out["forward_pass"]["write_kib"]) +
                   2 * (out["backward_pass"]["fetch_kib_x2"] + out["backward_pass"]["write_kib"]))
    out["hbm_bytes_per_step"] = step
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3])
