cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "persistent_walk or channels_last_tiles or retain_graph or lazy_flow" > gpurun_out/r4/t_e0.log 2>&1
rc=$?; tail -5 gpurun_out/r4/t_e0.log
if [ $rc -ne 0 ]; then echo "new kernel tests failed rc=$rc"; tail -80 gpurun_out/r4/t_e0.log; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_e.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_e.log
PIXPRO_NHWC_GRID=512 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_g512_e.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_g512_e.log
PIXPRO_NO_NHWC_KERNEL=1 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_old_e.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_old_e.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 > gpurun_out/r4/hp_ppm_nhwc_1024_e.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_1024_e.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 64 --grid 14 > gpurun_out/r4/hp_ppm_14_nhwc_e.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_14_nhwc_e.log
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so --nhwc > gpurun_out/r4/stamps_nhwc_e.log 2>&1; tail -24 gpurun_out/r4/stamps_nhwc_e.log
timeout -k 10 900 python -m pytest tests/test_gpu_ppm.py tests/test_gpu_sync_bn.py -q -m gpu > gpurun_out/r4/t_e1.log 2>&1
echo "ppm+bn rc=$?"; tail -8 gpurun_out/r4/t_e1.log
timeout -k 10 900 python -m pytest tests/test_bench_launcher.py tests/test_gpu_model.py -q -m gpu > gpurun_out/r4/t_e2.log 2>&1
echo "launcher+model rc=$?"; tail -8 gpurun_out/r4/t_e2.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_e.log 2> gpurun_out/r4/bench_e.err
echo "bench rc=$?"; grep -v "warm-up step" gpurun_out/r4/bench_e.err | tail -12; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4/bench_e.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['config'].get('miopen_untuned_ms_per_step'))
for k,v in d['hot_kernels'].items(): print(k, v.get('ms_per_launch', v.get('ms_per_step')), v['frac'])
print(d['hot_kernels']['batch_norm']['by_call'])
for k,v in d.get('extra_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), {a:b.get('ms_per_launch') for a,b in v.get('hot_kernels',{}).items()})
PY
