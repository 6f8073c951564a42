# Session-2 opening run: GPU test suite, then the one-GPU A/B of the view-batched encoder passes at per-view batch 256.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2a
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/tests_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests_gpu.log
[ $rc -eq 0 ] || exit $rc
for v in "" "--bn-view-batch"; do
  timeout -k 10 400 python bench.py $v --steps 20 --warmup 10 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench_ab$v.log 2> $O/bench_ab$v.err || exit 1
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_ab$v.log') if l.startswith('{')][-1]); print('variant [$v]:', d['value'], d['ms_per_step'], d['config']['miopen'], {k:(x.get('ms_per_launch',x.get('ms_per_step')), x['frac']) for k,x in d['hot_kernels'].items()})"
done
