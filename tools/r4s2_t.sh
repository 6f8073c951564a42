cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s2t
mkdir -p $O
echo "== launches" > $O/fold.txt; python tools/bn_shapes.py 2>/dev/null >> $O/fold.txt
echo "== fold" >> $O/fold.txt; PIXPRO_BN_FOLD=1 python tools/bn_shapes.py 2>/dev/null >> $O/fold.txt
python3 - <<PY
t=open('$O/fold.txt').read().split('== ')
A=t[1].split('\n')[2:-2]; B=t[2].split('\n')[2:-2]
for a,b in zip(A,B):
    fa=[float(x.strip().split(' us')[0]) for x in a.split('|')[1:]]; fb=[float(x.strip().split(' us')[0]) for x in b.split('|')[1:]]
    print(a[:22], ' stats %7.1f -> %7.1f   bwd_reduce %7.1f -> %7.1f' % (fa[0], fb[0], fa[2], fb[2]))
print(t[1].split('\n')[-2]); print(t[2].split('\n')[-2])
PY
