cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "tiled or full_size or 14 or channels_last" > gpurun_out/r4/t_v1.log 2>&1; rc=$?; echo "tiled rc=$rc"; tail -3 gpurun_out/r4/t_v1.log
if [ $rc -ne 0 ]; then exit 1; fi
bash tools/tl_prof.sh gpurun_out/r4/tlprof_nhwc2 --channels-last | tail -8
for i in 1 2 3; do python tools/hotpath_bench.py --what ppm --channels-last --batch 64 --grid 14 2>/dev/null | tail -1; done
python tools/hotpath_bench.py --what ppm --batch 64 --grid 14 2>/dev/null | tail -1
