# zero_grad to None (no fill / accumulate-into-zeros launches): model tests and the step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2g
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_bench_launcher.py -q -x -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra-configs --no-untuned-probe > $O/bench.log 2> $O/bench.err || exit 1
python -c "import sys,json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"
