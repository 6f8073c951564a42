cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "nhwc" > gpurun_out/r4/t_s0.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r4/t_s0.log
if [ $rc -ne 0 ]; then exit 1; fi
for rep in 1 2; do
for s in "0,0" "2,8000" "2,16000" "2,32000" "4,4000" "4,8000" "4,16000" "8,2000" "8,4000"; do
echo "stagger $s"; PIXPRO_NHWC_STAGGER=$s python tools/hotpath_bench.py --what ppm --channels-last --batch 256 2>/dev/null | tail -1
done
done
