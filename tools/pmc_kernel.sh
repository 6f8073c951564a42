#!/bin/bash
# Per-kernel SQ / TCC counters of one hot-path workload (rocprofv3 --pmc, counters only: no trace domains besides
# --kernel-trace, as the pool requires).  Usage: tools/pmc_kernel.sh OUTDIR "<python args of tools/hotpath_bench.py>"
# Writes OUTDIR/pmc_summary.txt: median counter value per kernel name over its dispatches.
set -e
OUT=$1; shift
ARGS="$@"
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$OUT/$1" -- python3 tools/hotpath_bench.py $ARGS > "$OUT/$1.log" 2>&1
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"
run grbm "GRBM_GUI_ACTIVE"
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES" || true
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys, statistics, collections
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        m = re.search(r"\b(k_\w+(<[^>]*>)?|ppm_\w+)", row["Kernel_Name"])
        if m:
            vals[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.txt"), "w") as fh:
    for k, cs in sorted(vals.items()):
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"  {c:28s} median {statistics.median(v):16.1f}  n={len(v)}\n")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            # rocprofv3's MfmaUtil: matrix-core busy cycles (summed over the 1024 SIMDs) over active cycles x 1024 SIMDs.  The
            # per-dispatch GRBM_GUI_ACTIVE of the csv is the SUM over the 8 XCDs (8 x the kernel's cycles), hence the / 8.
            util = statistics.median(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / (statistics.median(cs["GRBM_GUI_ACTIVE"]) / 8 * 1024) * 100
            fh.write(f"  {'MFMA utilisation (derived)':28s} {util:23.1f} %\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())
PY
