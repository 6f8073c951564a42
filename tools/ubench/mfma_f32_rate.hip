// Microbenchmark: what a loop of v_mfma_f32_16x16x4_f32 fed from LDS sustains on gfx950, by waves per SIMD and by the
// amount of plain VALU work mixed in (the fp32 MFMA shares the vector datapath: do they add?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_rate.hip -o tools/ubench/mfma_f32_rate
// One block per CU (256 blocks), 256*W threads -> W waves per SIMD.  Per "slab": 4 k-steps x (8 ds_read_b32 + 16 MFMA)
// + VALU_PER_SLAB dependent-free v_fma_f32, exactly the inner structure of ppm_tiled.hip's strip product.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VALU_PER_SLAB>
__global__ void k(float* out, int slabs) {
  __shared__ float As[16 * 272], Bs[16 * 80];
  const int t = threadIdx.x, lane = t & 63, lr = lane & 15, lq = lane >> 4, wave = (t >> 6) & 3;
  for (int e = t; e < 16 * 272; e += blockDim.x) As[e] = 1.0f + 1e-3f * e;
  for (int e = t; e < 16 * 80; e += blockDim.x) Bs[e] = 0.5f + 1e-3f * e;
  __syncthreads();
  f32x4 acc[4][4];
  for (int m = 0; m < 4; ++m)
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x0 = t, x1 = t + 1, x2 = t + 2, x3 = t + 3;
  const float a = 1.0f + 1e-6f * t, b = 0.5f;
  for (int s = 0; s < slabs; ++s) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float av[4], bv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = As[(4 * ks + lq) * 272 + 64 * wave + 16 * m + lr];
#pragma unroll
      for (int n = 0; n < 4; ++n) bv[n] = Bs[(4 * ks + lq) * 80 + 16 * n + lr];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < VALU_PER_SLAB / 4; ++v) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
    }
  }
  float r = x0 + x1 + x2 + x3;
  for (int m = 0; m < 4; ++m)
    for (int n = 0; n < 4; ++n) r += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * blockDim.x + t] = r;
}

template <int V>
void run(float* out) {
  hipEvent_t s, e;
  (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  const int slabs = 2000;
  for (int W = 1; W <= 4; W *= 2) {
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * W), 0, 0, out, slabs);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * W), 0, 0, out, slabs);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double mfma = (double)slabs * 64 * 4 * W * 256;      // wave-MFMAs on the chip
    printf("VALU/slab %3d  W=%d waves/SIMD: %.3f ms  %.1f TFLOP/s (MFMA only)  %.1f ns per MFMA per SIMD\n", V, W, ms,
           mfma * 2048 / ms / 1e9, ms * 1e6 / (slabs * 64.0 * W));
  }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  run<0>(out);
  run<64>(out);
  run<160>(out);
  return 0;
}
