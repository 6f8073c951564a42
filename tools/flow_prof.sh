#!/bin/bash
# Kernel times of the flow preparation (rocprofv3 kernel trace of tools/hotpath_bench.py --what flow).
# Usage: tools/flow_prof.sh OUTDIR [hotpath_bench args, e.g. --batch 64 --n-frames 6 --lib exp_libs/x.so]
OUT=${1:-gpurun_out/flow_prof}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/hotpath_bench.py --what flow "$@" > "$OUT/run.log" 2>&1
grep "^flow_prepare" "$OUT/run.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_prepare\w+|k_chain\w+|k_fbc\w+)", r["Name"])
    if m:
        print(f"  {m.group(1):22s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
