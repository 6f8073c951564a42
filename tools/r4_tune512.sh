# Record MIOpen's find-db entries for the view-batched encoder passes at per-GPU batch 256 (convolutions at batch 512):
# what `--bn-view-batch` (and the rccl.syncbn_ab variant of a multi-GPU run) needs to be compared fairly.
cd $GRAFT_REPO_ROOT
DB=$PWD/pixpro-with-opticalflow_amd/miopen_db
mkdir -p gpurun_out/miopen_db_new
(while true; do sleep 50; echo "[hb] $(date +%T) $(du -sh $DB | cut -f1)"; done) &
HB=$!
MIOPEN_USER_DB_PATH=$DB MIOPEN_CUSTOM_CACHE_DIR=$DB PIXPRO_MIOPEN_FIND=1 timeout -k 10 1000 python bench.py --bn-view-batch --steps 2 --warmup 2 --no-cpu-baseline --no-extra-configs --no-untuned-probe > gpurun_out/miopen_tune512.log 2>&1
echo "tune rc=$?"
kill $HB
grep -E "warm-up|timed" gpurun_out/miopen_tune512.log | tail -6
cp $DB/*.ufdb.txt $DB/*.udb.txt $DB/*.ukdb gpurun_out/miopen_db_new/
ls -la gpurun_out/miopen_db_new
# and what it buys on one GPU
python bench.py --bn-view-batch --steps 10 --warmup 5 --no-cpu-baseline --no-extra-configs --no-untuned-probe 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('view-batch tuned:', d['value'], d['ms_per_step'], d['config']['miopen'])"
