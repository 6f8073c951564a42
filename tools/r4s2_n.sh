# inclusive per-convolution times under three solver settings (does the split-K solvers' zero fill make another solver better?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/conv_probe
mkdir -p $O
python tools/conv_probe.py --tag default > $O/default.txt 2>&1; tail -1 $O/default.txt
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 python tools/conv_probe.py --tag no_asm > $O/no_asm.txt 2>&1; tail -1 $O/no_asm.txt
python tools/conv_probe.py --compare $O/default.json $O/no_asm.json > $O/compare.txt; tail -1 $O/compare.txt
