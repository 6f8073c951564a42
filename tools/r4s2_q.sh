cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2q
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_sync_bn.py -q -x -k "stem" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extra-configs --no-untuned-probe --no-overlap-momentum > $O/bench_prof.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/prof
python3 - <<PY
import csv
for r in csv.DictReader(open('$O/kernel_stats.csv')):
    if 'pool' in r['Name'] or 'publish' in r['Name']:
        print(r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3, 'us')
PY
