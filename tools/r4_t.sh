cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
L=pixpro-with-opticalflow_amd/lib/libpixpro_hip.so
cp $L /tmp/base.so
cp exp_libs/libpixpro_bnF.so $L
timeout -k 10 600 python -m pytest tests/test_gpu_sync_bn.py -q -m gpu -x > gpurun_out/r4/t_t0.log 2>&1; rc=$?; echo "bnF tests rc=$rc"; tail -3 gpurun_out/r4/t_t0.log
cp /tmp/base.so $L
if [ $rc -ne 0 ]; then exit 1; fi
python tools/bn_sweep.py --caps 256,512,1024,512
for v in A B C D E; do
python tools/bn_sweep.py --lib exp_libs/libpixpro_bn$v.so --caps 256,512,1024,512 256,512,512,512
done
echo "fold (fence-free) with pipelined loops (F) and with the old loops (G)"
PIXPRO_BN_FOLD=1 python tools/bn_sweep.py --lib exp_libs/libpixpro_bnF.so --caps 256,512,1024,512 256,512,512,512
PIXPRO_BN_FOLD=1 python tools/bn_sweep.py --lib exp_libs/libpixpro_bnG.so --caps 256,512,1024,512
python tools/bn_sweep.py --lib exp_libs/libpixpro_bnG.so --caps 256,512,1024,512
python tools/bn_sweep.py --caps 256,512,1024,512
