cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "persistent_walk or channels_last_tiles or retain_graph or lazy_flow or full_size or hotpath_golden or vs_oracle" > gpurun_out/r4/t_o0.log 2>&1
rc=$?; tail -5 gpurun_out/r4/t_o0.log
if [ $rc -ne 0 ]; then echo "new kernel tests failed rc=$rc"; tail -80 gpurun_out/r4/t_o0.log; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_o.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_o.log
PIXPRO_NHWC_NOWARM=1 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_nowarm_o.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_nowarm_o.log
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_o2.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_o2.log
PIXPRO_NHWC_NOWARM=1 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_nowarm_o2.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_nowarm_o2.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 > gpurun_out/r4/hp_ppm_nhwc_1024_o.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_1024_o.log
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so --nhwc > gpurun_out/r4/stamps_nhwc_o.log 2>&1; tail -28 gpurun_out/r4/stamps_nhwc_o.log
