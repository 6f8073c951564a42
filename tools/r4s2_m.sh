cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s2m
mkdir -p $O
rm -f $O/caps.txt
for c in 384,512,1024,512 384,384,1024,384 384,768,1024,768 384,1024,1024,1024 320,512,768,512 448,512,1280,512 384,512,512,512 384,256,1024,256; do
  echo "== caps $c" >> $O/caps.txt
  PIXPRO_BN_CAPS=$c python tools/bn_shapes.py 2>/dev/null | tail -1 >> $O/caps.txt
done
cat $O/caps.txt
