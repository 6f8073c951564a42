# kernel statistics of the step with / without the shortcut link (why is the link slower?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2d
mkdir -p $O
for v in 1 0; do
  export PIXPRO_BN_LAZY_SHORTCUT=$v
  rocprofv3 --kernel-trace --stats -d $O/prof$v --output-format csv -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-extra-configs --no-untuned-probe --no-overlap-momentum > $O/bench_prof$v.log 2>&1 || exit 1
  cp $(ls $O/prof$v/*/*kernel_stats.csv | head -1) $O/kernel_stats_lazy$v.csv
  rm -rf $O/prof$v
  grep -E "k_bwd_apply|k_masked_add|CUDAFunctor_add|k_bwd_reduce" $O/kernel_stats_lazy$v.csv | cut -c1-60,200-400 | head -20
done
