cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
L=${1:-wstamps}
python tools/ppm_phase_stamps.py --nhwc --lib exp_libs/libpixpro_$L.so 2>/dev/null > gpurun_out/r4/wstamps_$L.log; cat gpurun_out/r4/wstamps_$L.log
python tools/ppm_phase_stamps.py --nhwc --lib exp_libs/libpixpro_$L.so 2>/dev/null | tail -18
