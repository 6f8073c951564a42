# after the BatchNorm ReLU bit record + shortcut link: whole GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2f
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/tests_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests_gpu.log
[ $rc -eq 0 ] || exit $rc
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
