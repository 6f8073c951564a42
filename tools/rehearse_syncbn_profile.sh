#!/bin/bash
# Two ranks on one GPU over gloo (PIXPRO_REHEARSAL=1), rank 0 under rocprofv3 --kernel-trace: shows which BatchNorm
# kernels the SyncBN path launches and what they cost next to the single-rank (MIOpen) ones.  Not a measurement of
# multi-GPU throughput.  usage: tools/rehearse_syncbn_profile.sh <batch> <outdir>
set -e
B=${1:-64}; OUT=${2:-gpurun_out/syncbn}
mkdir -p "$OUT"
export TMPDIR=/tmp PIXPRO_REHEARSAL=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2
RANK=1 LOCAL_RANK=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 3 --warmup 2 --batch "$B" > "$OUT/rank1.log" 2>&1 &
R1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof2" -o r0 -- python3 bench.py --gpus 2 --steps 3 --warmup 2 --batch "$B" > "$OUT/rank0.log" 2>&1
wait $R1
unset WORLD_SIZE PIXPRO_REHEARSAL
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1" -o r0 -- python3 bench.py --steps 3 --warmup 2 --batch "$B" --no-cpu-baseline > "$OUT/single.log" 2>&1
find "$OUT" -type f ! -name "*kernel_stats.csv" ! -name "*.log" -delete
