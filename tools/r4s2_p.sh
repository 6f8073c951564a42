# the stem in one pass: tests, then the step with / without
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2p
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sync_bn.py tests/test_gpu_model.py -q -x > $O/tests_bn.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests_bn.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  PIXPRO_BN_FUSED_POOL=$v timeout -k 10 400 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench_pool$v.log 2> $O/bench_pool$v.err || { tail -5 $O/bench_pool$v.err; exit 1; }
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_pool$v.log') if l.startswith('{')][-1]); print('fused pool=$v:', d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
