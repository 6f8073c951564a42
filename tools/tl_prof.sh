#!/bin/bash
# Per-kernel times of the tiled PPM/loss path at BASELINE config 5 (N = 64, 14 x 14, C = 256): rocprofv3 kernel trace of
# tools/hotpath_bench.py, average duration per kernel.  Usage: tools/tl_prof.sh OUTDIR [extra hotpath_bench args]
OUT=${1:-gpurun_out/tl_prof}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/hotpath_bench.py --what ppm --batch 64 --grid 14 "$@" > "$OUT/run.log" 2>&1
tail -2 "$OUT/run.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_tl_" in n or "ppm_" in n:
        import re
        short = re.search(r"(k_tl_\w+(<[^>]*>)?|ppm_\w+(<[^>]*>)?)", n).group(1)
        avg = float(r["AverageNs"]) / 1e3
        tot += avg
        print(f"{short:28s} calls {r['Calls']:>4s}  avg {avg:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
print(f"sum of averages {tot:.1f} us")
PY
