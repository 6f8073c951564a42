# weight gradients on a second stream: the step with / without
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2i
mkdir -p $O
for v in on off; do
  timeout -k 10 400 python bench.py --wgrad-stream $v --steps 20 --warmup 10 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench_$v.log 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_$v.log') if l.startswith('{')][-1]); print('wgrad-stream $v:', d['value'], d['ms_per_step'], d['config']['weight_gradients'], d['config']['batch_norm'], d['config']['final_loss'])"
done
