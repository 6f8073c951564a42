# BatchNorm row walk: four adjacent row blocks per thread (BN_ADJ) against four far-apart ones, over work-group counts
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2e
mkdir -p $O
python tools/bn_sweep.py --lib exp_libs/libpixpro_base.so --caps 256,512,1024,512 256,2048,1024,2048 > $O/sweep.txt 2>&1
python tools/bn_sweep.py --lib exp_libs/libpixpro_adj.so --caps 256,512,1024,512 256,1024,1024,1024 256,2048,1024,2048 256,8192,1024,8192 256,65535,1024,65535 512,2048,512,2048 128,2048,256,2048 1024,4096,1024,4096 >> $O/sweep.txt 2>&1
cat $O/sweep.txt
