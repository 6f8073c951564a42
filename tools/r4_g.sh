cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so --nhwc > gpurun_out/r4/stamps_nhwc_g.log 2>&1; tail -28 gpurun_out/r4/stamps_nhwc_g.log
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so --nhwc --batch 1024 > gpurun_out/r4/stamps_nhwc_1024_g.log 2>&1; tail -5 gpurun_out/r4/stamps_nhwc_1024_g.log
timeout -k 10 300 python -m pytest tests/test_gpu_sync_bn.py -q -m gpu > gpurun_out/r4/t_g1.log 2>&1; echo "bn rc=$?"; tail -3 gpurun_out/r4/t_g1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > gpurun_out/r4/bench_g.log 2> gpurun_out/r4/bench_g.err
python - <<PY
import json
l=[x for x in open('gpurun_out/r4/bench_g.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'])
print(d['hot_kernels']['batch_norm']['ms_per_step'], d['hot_kernels']['batch_norm']['by_call'])
PY
