cd $GRAFT_REPO_ROOT
V=${1:-v7}
mkdir -p gpurun_out/r4
L=pixpro-with-opticalflow_amd/lib/libpixpro_hip.so
cp $L /tmp/base.so
cp exp_libs/libpixpro_$V.so $L
timeout -k 10 600 python -m pytest tests/test_gpu_ppm.py -q -m gpu -x -k "nhwc or channels_last or retain or full_size or lazy or arbiter" > gpurun_out/r4/t_r_$V.log 2>&1; rc=$?; echo "$V tests rc=$rc"; tail -5 gpurun_out/r4/t_r_$V.log
cp /tmp/base.so $L
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
echo base; python tools/hotpath_bench.py --what ppm --channels-last --batch 256 2>/dev/null | tail -1
echo $V; python tools/hotpath_bench.py --what ppm --channels-last --batch 256 --lib exp_libs/libpixpro_$V.so 2>/dev/null | tail -1
done
echo ${V}_1024; python tools/hotpath_bench.py --what ppm --channels-last --batch 1024 --lib exp_libs/libpixpro_$V.so 2>/dev/null | tail -1
python tools/ppm_phase_stamps.py --nhwc --lib exp_libs/libpixpro_${V}_stamps.so 2>/dev/null > gpurun_out/r4/stamps_$V.log; cat gpurun_out/r4/stamps_$V.log
