#!/bin/bash
# Record MIOpen's find-db / kernel cache for one bench configuration into pixpro-with-opticalflow_amd/miopen_db
# (MIOpen's own tuner, torch.backends.cudnn.benchmark = True), then copy the files to gpurun_out/miopen_db_new/ so they
# can be committed.  usage (on the GPU box): tools/miopen_tune.sh --batch 64 --n-frames 6
set -e
DB=pixpro-with-opticalflow_amd/miopen_db
(while true; do sleep 55; echo "[hb] $(date +%T) $(du -sh $DB | cut -f1)"; done) &
HB=$!
PIXPRO_MIOPEN_FIND=1 timeout -k 10 1000 python bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/miopen_tune.log 2>&1 || true
kill $HB
grep -E "warm-up|timed" gpurun_out/miopen_tune.log || tail -5 gpurun_out/miopen_tune.log
mkdir -p gpurun_out/miopen_db_new
cp $DB/*.ufdb.txt $DB/*.udb.txt $DB/*.ukdb gpurun_out/miopen_db_new/
ls -la gpurun_out/miopen_db_new
