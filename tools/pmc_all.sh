# Round-end PMC summaries of the hot kernels (tools/pmc_kernel.sh per workload); outputs under gpurun_out/final/.
set -e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/final
bash tools/pmc_kernel.sh gpurun_out/final/pmc_ppm "--what ppm" > gpurun_out/final/pmc_ppm.log 2>&1
bash tools/pmc_kernel.sh gpurun_out/final/pmc_flow "--what flow" > gpurun_out/final/pmc_flow.log 2>&1
bash tools/pmc_kernel.sh gpurun_out/final/pmc_flow6 "--what flow --n-frames 6 --batch 64" > gpurun_out/final/pmc_flow6.log 2>&1
bash tools/pmc_kernel.sh gpurun_out/final/pmc_tiled "--what ppm --grid 14 --batch 64" > gpurun_out/final/pmc_tiled.log 2>&1
echo pmc done
