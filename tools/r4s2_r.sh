cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s2r
mkdir -p $O
rm -f $O/rows8.txt
echo "== in-tree (4 rows everywhere)" >> $O/rows8.txt; python tools/bn_shapes.py 2>/dev/null | tail -1 >> $O/rows8.txt
run() { echo "== $1 caps $2" >> $O/rows8.txt; PIXPRO_BN_CAPS=$2 python tools/bn_shapes.py --lib exp_libs/libpixpro_$1.so 2>/dev/null | tail -1 >> $O/rows8.txt; }
run r8 384,512,256,512
run r8 384,512,192,512
run r8 384,512,320,512
run r8s8 384,512,256,512
run r8s8 256,512,256,512
run r8s8 192,512,256,512
run r8a8 384,512,256,512
run r8a8 384,256,256,512
run r8a8 384,384,256,512
run r8b8 384,512,256,512
run r8b8 384,512,256,256
run r8b8 384,512,256,384
cat $O/rows8.txt
