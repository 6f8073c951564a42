# Round-4 (second session) end measurements on one GPU box: full GPU test suite, smoke, default bench line, the same under
# rocprofv3 kernel stats, HBM traffic of the BatchNorm kernels (PMC, separate passes).  Outputs under gpurun_out/r4s2final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open('$O/bench_default.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['config'].get('miopen_untuned_ms_per_step'))
for k,v in d['hot_kernels'].items(): print(k, v.get('ms_per_launch', v.get('ms_per_step')), v['frac'])
print(d['hot_kernels']['batch_norm']['by_call'])
for k,v in d.get('extra_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), {a:(b.get('ms_per_launch'), b.get('frac')) for a,b in v.get('hot_kernels',{}).items()})
print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:200])
PY
rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench_prof.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/prof; head -8 $O/kernel_stats.csv | cut -c1-150
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bn_fetch -- python3 tools/bn_pmc.py run > $O/bn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bn_write -- python3 tools/bn_pmc.py run > $O/bn_write.log 2>&1
python3 tools/bn_pmc.py parse $O/bn_fetch $O/bn_write > $O/bn_traffic.json; cat $O/bn_traffic.json; rm -rf $O/bn_fetch $O/bn_write
echo done
