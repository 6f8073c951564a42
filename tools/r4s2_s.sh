cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2s
mkdir -p $O
rocprofv3 --kernel-trace -d $O/prof --output-format csv -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra-configs --no-untuned-probe --no-overlap-momentum > $O/bench.log 2>&1
F=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python3 tools/zero_fill_trace.py $F > $O/zero_fill.txt; head -30 $O/zero_fill.txt | cut -c1-230
head -2 $F | cut -c1-400
rm -rf $O/prof
