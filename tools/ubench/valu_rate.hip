// Microbenchmark: issue rate of plain and packed f32 VALU instructions on gfx950, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// One block per CU (256 blocks), block = 256 * W threads -> W waves per SIMD.  Inline asm keeps the compiler from
// packing / eliminating anything.  Reports cycles per instruction per SIMD at the measured clock (s_memtime / wall).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* clk) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  const f2 pa = {a, a}, pb = {b, b};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 1) {
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
    } else if (MODE == 2) {
      asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                   "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 3) {   // int / convert mix typical of address + floor work
      asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0\n v_floor_f32 %1, %1\n v_add_u32 %2, %2, %2\n"
                   "v_mul_lo_u32 %3, %3, %3\n v_min_i32 %4, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_mad_u32_u24 %7, %7, %7, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* clk) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  const int iters = 20000;
  for (int W = 1; W <= 4; W *= 2) {
    const int threads = 256 * W;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    unsigned long long c;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)iters * 8 * W;
    printf("%-28s W=%d waves/SIMD: %.3f ms, %.2f shader cycles per instruction per SIMD (in-kernel clock %.2f GHz)\n", name, W, ms,
           (double)c / instr_per_simd, (double)c / (ms * 1e6));
  }
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&clk, 8);
  run<0>("v_fma_f32", out, clk);
  run<1>("v_pk_fma_f32", out, clk);
  run<2>("v_mul_f32 / v_add_f32", out, clk);
  run<3>("cvt/floor/int mix", out, clk);
  return 0;
}
