# streaming-load policy of the BatchNorm kernels: per shape forced off / forced on / policy, tests, the step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2l
mkdir -p $O
rm -f $O/shapes.txt
for m in 0 1 policy; do
  echo "== PIXPRO_BN_STREAM=$m" >> $O/shapes.txt
  if [ $m = policy ]; then python tools/bn_shapes.py 2>/dev/null >> $O/shapes.txt; else PIXPRO_BN_STREAM=$m python tools/bn_shapes.py 2>/dev/null >> $O/shapes.txt; fi
done
grep "==\|per step" $O/shapes.txt
timeout -k 10 600 python -m pytest tests/test_gpu_sync_bn.py tests/test_gpu_model.py -q -x > $O/tests_bn.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests_bn.log
[ $rc -eq 0 ] || exit $rc
for m in policy 0; do
  if [ $m = policy ]; then unset PIXPRO_BN_STREAM; else export PIXPRO_BN_STREAM=$m; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench_$m.log 2> $O/bench_$m.err || exit 1
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_$m.log') if l.startswith('{')][-1]); print('stream=$m:', d['value'], d['ms_per_step'], d['hot_kernels']['batch_norm']['ms_per_step'], d['hot_kernels']['batch_norm']['by_call'])"
done
