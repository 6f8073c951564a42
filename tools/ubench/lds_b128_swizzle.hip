// ds_read_b128 fragment reads of a [row][16 float] slab image (the strip products' operand layout, ppm_tiled.hip): cycles per
// read for several placements of the 16-B chunks.  One wave, reads issued back to back, timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_b128_swizzle tools/ubench/lds_b128_swizzle.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(unsigned long long* out, float* sink, int waves_active) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 24];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
  for (int i = threadIdx.x; i < 64 * 24; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  int off;
  if (MODE == 0) off = lr * 16 + 4 * lq;                                  // no swizzle: rows 64 B apart
  if (MODE == 1) off = lr * 16 + 4 * (lq ^ ((-(lr >> 2)) & 3));           // ppm_tiled's swizzle
  if (MODE == 2) off = lr * 16 + 4 * (lq ^ ((lr >> 2) & 3));              // plain xor with the row group
  if (MODE == 3) off = lr * 20 + 4 * lq;                                  // padded rows (80 B)
  if (MODE == 4) off = lr * 16 + 4 * (lq ^ (lr & 3));                     // xor with the row
  if (MODE == 5) off = lane * 4;                                          // linear: lane i reads chunk i
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned addr = (unsigned)(size_t)(lds_ptr_t)lds + 4u * (unsigned)off;      // LDS byte address of this lane's chunk
  f4 v0, v1, v2, v3, v4, v5, v6, v7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8 offset:1024\n\t"
                 "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:1024\n\tds_read_b128 %6, %8\n\tds_read_b128 %7, %8 offset:1024\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                 : "v"(addr)
                 : "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float4 acc = make_float4(v0.x + v1.x + v2.x + v3.x, v4.y + v5.y, v6.z, v7.w);
  if (lane == 0 && wave == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  unsigned long long* d; float* s;
  CK(hipMalloc(&d, 8)); CK(hipMalloc(&s, 4096));
  const char* names[] = {"no swizzle (64-B rows)", "q ^ (-(r>>2) & 3)  [ppm_tiled]", "q ^ ((r>>2) & 3)", "80-B rows, no swizzle", "q ^ (r & 3)", "linear (lane i -> chunk i)"};
  for (int waves = 1; waves <= 4; waves *= 4)
    for (int m = 0; m < 6; ++m) {
      unsigned long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        switch (m) {
          case 0: k<0><<<1, 64 * waves>>>(d, s, waves); break;
          case 1: k<1><<<1, 64 * waves>>>(d, s, waves); break;
          case 2: k<2><<<1, 64 * waves>>>(d, s, waves); break;
          case 3: k<3><<<1, 64 * waves>>>(d, s, waves); break;
          case 4: k<4><<<1, 64 * waves>>>(d, s, waves); break;
          default: k<5><<<1, 64 * waves>>>(d, s, waves); break;
        }
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
      printf("%d wave(s)  %-34s %6.2f memtime ticks per ds_read_b128\n", waves, names[m], (double)h / (256.0 * 8));
    }
  return 0;
}
