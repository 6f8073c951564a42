// Check of the VALU cross-lane sums of csrc/common.h (rows_sum: gfx950 row swaps; wave_sum_valu: DPP + rows_sum) against
// the ds_bpermute butterflies they replace.  Build: hipcc --offload-arch=gfx950 -O3 -I pixpro-with-opticalflow_amd/csrc
// -o tools/ubench/lane_sums tools/ubench/lane_sums.hip ; prints per-lane results of one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
using namespace pixpro;
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float v = (float)(l * l % 17) + 0.25f * (float)l;
  float a = v;
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  out[l] = a;
  out[64 + l] = rows_sum(v);
  out[128 + l] = wave_sum(v);
  out[192 + l] = wave_sum_valu(v);
  out[256 + l] = v;
}
int main() {
  float* d;
  hipMalloc(&d, 320 * sizeof(float));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[320];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad_r = 0, bad_w = 0;
  for (int l = 0; l < 64; ++l) {
    if (h[l] != h[64 + l]) ++bad_r;
    if (h[128 + l] != h[192 + l]) ++bad_w;
  }
  printf("rows_sum mismatching lanes: %d, wave_sum_valu mismatching lanes: %d\n", bad_r, bad_w);
  for (int l = 0; l < 64; l += 7) printf("lane %2d v %7.2f  xor16+32 %8.2f rows_sum %8.2f | wave_sum %9.2f valu %9.2f\n", l, h[256 + l], h[l], h[64 + l], h[128 + l], h[192 + l]);
  return bad_r + bad_w != 0;
}
