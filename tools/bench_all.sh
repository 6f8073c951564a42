# Round-end measurements on a GPU box: default bench line, the same under rocprofv3 --kernel-trace --stats, BASELINE configs 4 and 5.
# Outputs under gpurun_out/final/ (copied into profiles/ by hand).
set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.log 2>gpurun_out/final/bench_default.err
tail -c 300 gpurun_out/final/bench_default.log; echo
rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof --output-format csv -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_prof.log 2>&1
cp $(ls gpurun_out/final/prof/*/*kernel_stats.csv | head -1) gpurun_out/final/kernel_stats.csv
head -12 gpurun_out/final/kernel_stats.csv | cut -c1-160
python bench.py --n-frames 6 --batch 64 --no-cpu-baseline > gpurun_out/final/bench_f6.log 2>/dev/null
python bench.py --image-size 448 --batch 64 --no-cpu-baseline > gpurun_out/final/bench_448.log 2>/dev/null
echo done
