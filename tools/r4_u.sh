cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
L=pixpro-with-opticalflow_amd/lib/libpixpro_hip.so
cp $L /tmp/base.so
run() {  # label, lib, env...
  label=$1; lib=$2; shift 2
  cp $lib $L
  env "$@" timeout -k 10 400 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-extra-configs --no-untuned-probe 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); bn=d['hot_kernels'].get('batch_norm',{})
print('$label', d['value'], d['ms_per_step'], 'bn ms', bn.get('ms_per_step'), 'frac', bn.get('frac'))" || echo "$label failed"
  cp /tmp/base.so $L
}
run base /tmp/base.so X=1
run fold_v3 exp_libs/libpixpro_bnG.so PIXPRO_BN_FOLD=1
run rev5 exp_libs/libpixpro_bnH.so PIXPRO_BN_REVERSE=5
run rev10 exp_libs/libpixpro_bnH.so PIXPRO_BN_REVERSE=10
run rev6 exp_libs/libpixpro_bnH.so PIXPRO_BN_REVERSE=6
run rev9 exp_libs/libpixpro_bnH.so PIXPRO_BN_REVERSE=9
run base2 /tmp/base.so X=1
