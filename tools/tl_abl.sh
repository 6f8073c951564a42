#!/bin/bash
# Timing ablations of the tiled path: builds libpixpro_hip variants with -DTL_ABL=n into exp_libs/ (git-ignored) HERE,
# then on the GPU box: tools/tl_abl.sh run  ->  per-kernel averages per variant.  Results of the variants are wrong by design.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" != "run" ]; then
  mkdir -p "$ROOT/exp_libs"
  for n in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -DTL_ABL=$n -c "$ROOT/pixpro-with-opticalflow_amd/csrc/ppm_tiled.hip" -o "$ROOT/exp_libs/ppm_tiled_abl$n.o"
    objs=$(ls "$ROOT"/pixpro-with-opticalflow_amd/lib/*.o | grep -v ppm_tiled.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/exp_libs/libpixpro_abl$n.so" $objs "$ROOT/exp_libs/ppm_tiled_abl$n.o"
  done
  exit 0
fi
for lib in "$ROOT"/exp_libs/libpixpro_abl*.so; do
  n=$(basename "$lib" .so)
  echo "== $n"
  "$ROOT/tools/tl_prof.sh" "gpurun_out/abl_$n" --lib "$lib" | grep -E "k_tl_|sum"
done
