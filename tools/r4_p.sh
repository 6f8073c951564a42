cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_ppm.py -q -m gpu -k "lazy" > gpurun_out/r4/t_p0.log 2>&1; echo "lazy rc=$?"; tail -4 gpurun_out/r4/t_p0.log
timeout -k 10 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs --no-untuned-probe 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default:', d['value'], d['ms_per_step'], d['config']['miopen'])"
timeout -k 10 300 python -m pytest tests/test_gpu_ppm.py -q -m gpu -k "tiled or full_size or 14" > gpurun_out/r4/t_p1.log 2>&1; echo "tiled rc=$?"; tail -3 gpurun_out/r4/t_p1.log
bash tools/tl_prof.sh gpurun_out/r4/tlprof_nhwc --channels-last | tail -8
python tools/hotpath_bench.py --what ppm --channels-last --batch 64 --grid 14 2>/dev/null | tail -1
