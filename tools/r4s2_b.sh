# ReLU bit record of the fused BatchNorm: tests, then the step with / without it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sync_bn.py tests/test_gpu_model.py -q -x > $O/tests_bn.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests_bn.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  PIXPRO_BN_RELU_BITS=$v timeout -k 10 400 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench_bits$v.log 2> $O/bench_bits$v.err || exit 1
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_bits$v.log') if l.startswith('{')][-1]); print('bits=$v:', d['value'], d['ms_per_step'], {k:(x.get('ms_per_launch',x.get('ms_per_step')), x['frac']) for k,x in d['hot_kernels'].items() if k=='batch_norm'}, d['hot_kernels']['batch_norm'])"
done
