# non-temporal loads / stores in the BatchNorm kernels, per shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s2k
mkdir -p $O
rm -f $O/shapes2.txt
for v in base nt_a nt_ra nt_sa nt_sra; do
  echo "== $v" >> $O/shapes2.txt
  python tools/bn_shapes.py --lib exp_libs/libpixpro_$v.so 2>/dev/null >> $O/shapes2.txt
done
echo "== nt_sra caps 512,512,1024,512" >> $O/shapes2.txt
PIXPRO_BN_CAPS=512,512,1024,512 python tools/bn_shapes.py --lib exp_libs/libpixpro_nt_sra.so 2>/dev/null >> $O/shapes2.txt
echo "== nt_sra caps 384,512,1024,512" >> $O/shapes2.txt
PIXPRO_BN_CAPS=384,512,1024,512 python tools/bn_shapes.py --lib exp_libs/libpixpro_nt_sra.so 2>/dev/null >> $O/shapes2.txt
grep "==\|per step" $O/shapes2.txt
