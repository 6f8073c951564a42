cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x > gpurun_out/r4/t_x0.log 2>&1; rc=$?; echo "ppm tests rc=$rc"; tail -5 gpurun_out/r4/t_x0.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2 3; do
echo rsq; python tools/hotpath_bench.py --what ppm --channels-last --batch 256 2>/dev/null | tail -1
echo base; python tools/hotpath_bench.py --what ppm --channels-last --batch 256 --lib exp_libs/libpixpro_base.so 2>/dev/null | tail -1
done
echo rsq1024; python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 2>/dev/null | tail -1
