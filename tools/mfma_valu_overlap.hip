// Microbenchmark: does v_mfma_f32_16x16x4_f32 overlap with f32 VALU work issued by ANOTHER wave of the same SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: every wave MFMA; 1: every wave VALU; 2: even waves MFMA, odd waves VALU (2 waves per SIMD: 512 threads)
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const float av = 1.0f + threadIdx.x * 1e-6f, bv = 0.5f;
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 4) == 0);   // waves 0-3 / 4-7 share the four SIMDs pairwise
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a3, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {   // 16 independent-ish fmas per iteration (= 4 MFMAs' worth of issue time at 2 cyc each? no: measured)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        x0 = __builtin_fmaf(x0, av, bv); x1 = __builtin_fmaf(x1, av, bv); x2 = __builtin_fmaf(x2, av, bv); x3 = __builtin_fmaf(x3, av, bv);
        x4 = __builtin_fmaf(x4, av, bv); x5 = __builtin_fmaf(x5, av, bv); x6 = __builtin_fmaf(x6, av, bv); x7 = __builtin_fmaf(x7, av, bv);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  const int iters = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    printf("mode %d: %.3f ms  (%.1f ns per iteration)\n", mode, ms, ms * 1e6 / iters);
  }
  return 0;
}
