cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "persistent_walk or channels_last_tiles or retain_graph or lazy_flow or full_size" > gpurun_out/r4/t_f0.log 2>&1
rc=$?; tail -5 gpurun_out/r4/t_f0.log
if [ $rc -ne 0 ]; then echo "new kernel tests failed rc=$rc"; tail -80 gpurun_out/r4/t_f0.log; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_f.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_f.log
PIXPRO_NO_NHWC_KERNEL=1 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_old_f.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_old_f.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 > gpurun_out/r4/hp_ppm_nhwc_1024_f.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_1024_f.log
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so --nhwc > gpurun_out/r4/stamps_nhwc_f.log 2>&1; tail -25 gpurun_out/r4/stamps_nhwc_f.log
timeout -k 10 900 python -m pytest tests/test_gpu_ppm.py tests/test_gpu_sync_bn.py -q -m gpu > gpurun_out/r4/t_f1.log 2>&1
echo "ppm+bn rc=$?"; tail -8 gpurun_out/r4/t_f1.log
for fold in 1 0; do
PIXPRO_BN_FOLD=$fold timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/r4/bench_f_fold$fold.log 2> gpurun_out/r4/bench_f_fold$fold.err
echo "bench fold=$fold rc=$?"; python - <<PY
import json
l=[x for x in open('gpurun_out/r4/bench_f_fold$fold.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'])
print(d['hot_kernels']['batch_norm']['ms_per_step'], d['hot_kernels']['batch_norm']['by_call'])
PY
done
