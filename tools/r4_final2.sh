# Round-4 end measurements on one GPU box: full GPU test suite, smoke, default bench line, the same under rocprofv3 kernel
# stats, PMC summaries of the PPM kernels (7x7 channels_last, 14x14 channels_last).  Outputs under gpurun_out/r4final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4final
mkdir -p $O
bash tools/pmc_kernel.sh $O/pmc_ppm --what ppm --channels-last > $O/pmc_ppm.log 2>&1; grep -A26 "^ppm_nhwc" $O/pmc_ppm/pmc_summary.txt | head -30
bash tools/pmc_kernel.sh $O/pmc_tiled --what ppm --grid 14 --batch 64 --channels-last > $O/pmc_tiled.log 2>&1; grep "MFMA utilisation\|^k_tl" $O/pmc_tiled/pmc_summary.txt
python tools/hotpath_bench.py --what ppm --channels-last > $O/hp_ppm_nhwc.log 2>&1; tail -1 $O/hp_ppm_nhwc.log
python tools/hotpath_bench.py --what ppm > $O/hp_ppm_nchw.log 2>&1; tail -1 $O/hp_ppm_nchw.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 64 --grid 14 > $O/hp_ppm14_nhwc.log 2>&1; tail -1 $O/hp_ppm14_nhwc.log
python tools/hotpath_bench.py --what ppm --batch 64 --grid 14 > $O/hp_ppm14_nchw.log 2>&1; tail -1 $O/hp_ppm14_nchw.log
echo done
# the strips' LDS bank-conflict counter with the copies (TL_ABL=4) or the MFMAs (TL_ABL=2) ablated: who produces the conflicts?
for v in abl4 abl2; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_tl_$v -- python3 tools/hotpath_bench.py --what ppm --grid 14 --batch 64 --lib exp_libs/libpixpro_$v.so > $O/pmc_tl_$v.log 2>&1
  python3 - $O/pmc_tl_$v $v <<'PY'
import csv, glob, os, re, sys, statistics, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        m = re.search(r"\b(k_tl_\w+(<[^>]*>)?)", row["Kernel_Name"])
        if m:
            vals[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(vals.items()):
    print(sys.argv[2], k, {c: statistics.median(v) for c, v in cs.items()})
PY
done
echo done2
