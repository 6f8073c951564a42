cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "nhwc or channels_last or retain or full_size" > gpurun_out/r4/t_q0.log 2>&1; rc=$?; echo "nhwc tests rc=$rc"; tail -5 gpurun_out/r4/t_q0.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
echo fold; python tools/hotpath_bench.py --what ppm --channels-last --batch 256 2>/dev/null | tail -1
echo nofold; PIXPRO_NHWC_NO_FOLD=1 python tools/hotpath_bench.py --what ppm --channels-last --batch 256 2>/dev/null | tail -1
done
echo fold1024; python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 2>/dev/null | tail -1
timeout -k 10 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs --no-untuned-probe 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default:', d['value'], d['ms_per_step'], d['hot_kernels']['ppm_loss'])"
