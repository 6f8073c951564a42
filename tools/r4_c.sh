cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
# the new kernel first, alone and under a short timeout: a hang must not take the rest of the call with it
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "persistent_walk or channels_last_tiles" > gpurun_out/r4/t_c0.log 2>&1
rc=$?; tail -5 gpurun_out/r4/t_c0.log
if [ $rc -ne 0 ]; then echo "new kernel tests failed rc=$rc"; tail -60 gpurun_out/r4/t_c0.log; exit 1; fi
timeout -k 10 900 python -m pytest tests/test_gpu_ppm.py tests/test_gpu_sync_bn.py -q -m gpu --deselect "tests/test_gpu_ppm.py::test_full_size_vs_oracle[shape1-False-True]" > gpurun_out/r4/t_c1.log 2>&1
echo "ppm+bn rc=$?"; tail -15 gpurun_out/r4/t_c1.log
timeout -k 10 900 python -m pytest tests/test_bench_launcher.py tests/test_gpu_model.py -q -m gpu > gpurun_out/r4/t_c2.log 2>&1
echo "launcher+model rc=$?"; tail -15 gpurun_out/r4/t_c2.log
python tools/hotpath_bench.py --what ppm > gpurun_out/r4/hp_ppm_nchw_c.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nchw_c.log
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_c.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_c.log
PIXPRO_NO_NHWC_KERNEL=1 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_old_c.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_old_c.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 > gpurun_out/r4/hp_ppm_nhwc_1024_c.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_1024_c.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_c.log 2> gpurun_out/r4/bench_c.err
echo "bench rc=$?"; tail -5 gpurun_out/r4/bench_c.err; tail -c 3000 gpurun_out/r4/bench_c.log
