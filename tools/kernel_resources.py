#!/usr/bin/env python
"""Per-kernel register / scratch / LDS / occupancy table of one csrc/*.hip file, from hipcc's own resource remarks
(cross-compiles for gfx950 without a GPU):  python tools/kernel_resources.py ppm_fused.hip [-DNAME=VALUE ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pixpro-with-opticalflow_amd", "csrc")


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    path = os.path.abspath(src) if os.path.exists(src) else os.path.join(CSRC, src)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
           "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(path)).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    try:
        names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True,
                               text=True).stdout.splitlines()
    except OSError:
        names = [r["name"] for r in rows]
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
    for r, n in zip(rows, names):
        n = re.sub(r"pixpro::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)
        print(f"{n[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")


if __name__ == "__main__":
    main()
