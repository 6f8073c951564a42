set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_ppm.py -q -m gpu -x --deselect "tests/test_gpu_ppm.py::test_full_size_vs_oracle[shape1-False-True]" > gpurun_out/r4/t_ppm_a.log 2>&1 || (tail -40 gpurun_out/r4/t_ppm_a.log; exit 1)
tail -3 gpurun_out/r4/t_ppm_a.log
python tools/hotpath_bench.py --what ppm > gpurun_out/r4/hp_ppm_nchw_a.log 2>&1
python tools/hotpath_bench.py --what ppm --channels-last > gpurun_out/r4/hp_ppm_nhwc_a.log 2>&1
tail -1 gpurun_out/r4/hp_ppm_nchw_a.log; tail -1 gpurun_out/r4/hp_ppm_nhwc_a.log
python tools/ppm_phase_stamps.py > gpurun_out/r4/stamps_a.log 2>&1
cat gpurun_out/r4/stamps_a.log | tail -20
