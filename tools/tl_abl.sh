for d in 0 1 3 7; do
  export PIXPRO_TL_DBG=$d
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_abl$d -- python3 tools/hotpath_bench.py --what ppm --batch 64 --grid 14 > gpurun_out/kt_abl$d.log 2>&1
  f=$(find gpurun_out/kt_abl$d -name "*kernel_stats.csv" | head -1)
  echo dbg=$d $(grep "k_tl_strip<4, 0>" $f | awk -F'",' '{print $2}' | cut -d, -f3)
done
