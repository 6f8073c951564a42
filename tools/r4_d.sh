cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
# the new kernel first, alone and under a short timeout: a hang must not take the rest of the call with it
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "persistent_walk or channels_last_tiles" > gpurun_out/r4/t_d0.log 2>&1
rc=$?; tail -5 gpurun_out/r4/t_d0.log
if [ $rc -ne 0 ]; then echo "new kernel tests failed rc=$rc"; tail -80 gpurun_out/r4/t_d0.log; fi
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_d.log
PIXPRO_NO_NHWC_KERNEL=1 python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_old_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_old_d.log
python tools/hotpath_bench.py --what ppm > gpurun_out/r4/hp_ppm_nchw_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nchw_d.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 > gpurun_out/r4/hp_ppm_nhwc_1024_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_1024_d.log
PIXPRO_NO_NHWC_KERNEL=1 python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 > gpurun_out/r4/hp_ppm_nhwc_old_1024_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_nhwc_old_1024_d.log
python tools/hotpath_bench.py --what ppm --channels-last --batch 64 --grid 14 > gpurun_out/r4/hp_ppm_14_nhwc_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_14_nhwc_d.log
python tools/hotpath_bench.py --what ppm --batch 64 --grid 14 > gpurun_out/r4/hp_ppm_14_nchw_d.log 2>&1; tail -1 gpurun_out/r4/hp_ppm_14_nchw_d.log
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so --nhwc > gpurun_out/r4/stamps_nhwc_d.log 2>&1; tail -24 gpurun_out/r4/stamps_nhwc_d.log
python tools/ppm_phase_stamps.py --lib exp_libs/libpixpro_hip_stamps.so > gpurun_out/r4/stamps_nchw_d.log 2>&1; tail -18 gpurun_out/r4/stamps_nchw_d.log
timeout -k 10 900 python -m pytest tests/test_gpu_ppm.py tests/test_gpu_sync_bn.py -q -m gpu > gpurun_out/r4/t_d1.log 2>&1
echo "ppm+bn rc=$?"; tail -15 gpurun_out/r4/t_d1.log
timeout -k 10 900 python -m pytest tests/test_bench_launcher.py tests/test_gpu_model.py tests/test_gpu_flow.py -q -m gpu > gpurun_out/r4/t_d2.log 2>&1
echo "launcher+model+flow rc=$?"; tail -15 gpurun_out/r4/t_d2.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_d.log 2> gpurun_out/r4/bench_d.err
echo "bench rc=$?"; tail -5 gpurun_out/r4/bench_d.err; tail -c 3000 gpurun_out/r4/bench_d.log
bash tools/pmc_kernel.sh gpurun_out/r4/pmc_ppm_nhwc --what ppm --channels-last > gpurun_out/r4/pmc_ppm_nhwc.log 2>&1; tail -30 gpurun_out/r4/pmc_ppm_nhwc/pmc_summary.txt
