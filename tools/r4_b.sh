set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_ppm.py tests/test_gpu_sync_bn.py -q -m gpu -x --deselect "tests/test_gpu_ppm.py::test_full_size_vs_oracle[shape1-False-True]" > gpurun_out/r4/t_b.log 2>&1 || (tail -60 gpurun_out/r4/t_b.log; exit 1)
tail -3 gpurun_out/r4/t_b.log
python -m pytest tests/test_bench_launcher.py tests/test_gpu_model.py -q -m gpu -x > gpurun_out/r4/t_b2.log 2>&1 || (tail -60 gpurun_out/r4/t_b2.log; exit 1)
tail -3 gpurun_out/r4/t_b2.log
python tools/hotpath_bench.py --what ppm > gpurun_out/r4/hp_ppm_nchw_a.log 2>&1
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_a.log 2>&1
tail -1 gpurun_out/r4/hp_ppm_nchw_a.log; tail -1 gpurun_out/r4/hp_ppm_nhwc_a.log
python tools/ppm_phase_stamps.py > gpurun_out/r4/stamps_a.log 2>&1
tail -20 gpurun_out/r4/stamps_a.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_b.log 2> gpurun_out/r4/bench_b.err || (tail -30 gpurun_out/r4/bench_b.err; exit 1)
tail -c 1500 gpurun_out/r4/bench_b.log
