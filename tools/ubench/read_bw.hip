// Read-only streaming probe for the BatchNorm statistics / backward-reduction kernels: an 822 MB fp32
// tensor (N = 256, 56 x 56, C = 256: the widest activation of ResNet-50) summed per lane, in the shapes a kernel could use.
// The element-wise BatchNorm kernels (read + write) run at 5.8-6.0 TB/s, the read-only ones at 4.5-5.0: is that the
// kernels or the chip?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/read_bw tools/ubench/read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// V1: grid-stride, U loads of 16 B in flight per lane (k_stats is U = 4 with 256 work-groups)
template <int U>
__global__ void __launch_bounds__(256) k_stride(const float4* __restrict__ p, size_t n4, float4* __restrict__ out) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n4; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) s = add4(s, v[u]);
  }
  for (; i < n4; i += step) s = add4(s, p[i]);
  if (s.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V2: every work-group owns contiguous 16 KB pieces (U x 4 KB adjacent), pieces dealt round-robin
template <int U>
__global__ void __launch_bounds__(256) k_chunk(const float4* __restrict__ p, size_t n4, float4* __restrict__ out) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t jump = (size_t)gridDim.x * 256 * U;
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  for (; i + (U - 1) * 256 < n4; i += jump) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) s = add4(s, v[u]);
  }
  if (s.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V3: non-temporal loads (the data is read once)
template <int U>
__global__ void __launch_bounds__(256) k_stride_nt(const float4* __restrict__ p, size_t n4, float4* __restrict__ out) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n4; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* q = reinterpret_cast<const float*>(p + i + u * step);
      v[u] = make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2),
                         __builtin_nontemporal_load(q + 3));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s = add4(s, v[u]);
  }
  if (s.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V4: LDS-DMA: every wave keeps D copies of 1 KB in flight into its own ring, reads its 16 B back when a copy has landed
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(const float4* src, float4* lds_wave_base) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}
template <int D>
__global__ void __launch_bounds__(256) k_dma(const float4* __restrict__ p, size_t n4, float4* __restrict__ out) {
  __shared__ float4 ring[4][D][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  // prologue: D copies
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (i + d * step < n4) dma16(p + i + d * step, &ring[wave][d][0]);
  for (; i + (2 * D - 1) * step < n4; i += D * step) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");      // the oldest copy has landed
      const float4 v = ring[wave][d][lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      s = add4(s, v);
      dma16(p + i + (D + d) * step, &ring[wave][d][0]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int d = 0; d < D; ++d)
    if (i + d * step < n4) s = add4(s, ring[wave][d][lane]);
  if (s.x == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V5: what k_stats computes per element (shifted sums of x and x^2), U rows in flight, plain or non-temporal loads
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stats_like(const float4* __restrict__ p, size_t n4, float4* __restrict__ out) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  const float4 K = p[threadIdx.x & 63];
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n4; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        const float* a = reinterpret_cast<const float*>(p + i + u * step);
        v[u] = make_float4(__builtin_nontemporal_load(a), __builtin_nontemporal_load(a + 1), __builtin_nontemporal_load(a + 2),
                           __builtin_nontemporal_load(a + 3));
      } else {
        v[u] = p[i + u * step];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float dx = v[u].x - K.x, dy = v[u].y - K.y, dz = v[u].z - K.z, dw = v[u].w - K.w;
      s.x += dx; s.y += dy; s.z += dz; s.w += dw;
      q.x = __builtin_fmaf(dx, dx, q.x); q.y = __builtin_fmaf(dy, dy, q.y);
      q.z = __builtin_fmaf(dz, dz, q.z); q.w = __builtin_fmaf(dw, dw, q.w);
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  out[(size_t)blockIdx.x * 512 + 256 + threadIdx.x] = q;
}

__global__ void k_randfill(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) * (1.f / 32768.f) - 1.f;
  }
}

// for comparison: copy (1 read + 1 write) and the same as two reads + one write
template <int U>
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ p, float4* __restrict__ q, size_t n4) {
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * step < n4; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) q[i + u * step] = v[u];
  }
}

template <class F>
static void timeit(const char* name, double bytes, F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  printf("%-58s mean %7.1f us  best %7.1f us  %6.2f TB/s (best)\n", name, sum / reps * 1e3, best * 1e3, bytes / best / 1e9);
  fflush(stdout);
}

int main() {
  const size_t n4 = (size_t)256 * 56 * 56 * 256 / 4;       // 822 MB
  const double bytes = (double)n4 * 16;
  // several buffers in turn so that the Infinity Cache (256 MB) cannot help: each launch reads another one
  const int NB = 3;
  float4* buf[NB]; float4* out; float4* dst;
  for (int b = 0; b < NB; ++b) {
    CK(hipMalloc(&buf[b], n4 * 16));
    hipLaunchKernelGGL(k_randfill, dim3(2048), dim3(256), 0, 0, reinterpret_cast<float*>(buf[b]), n4 * 4);     // real data
  }
  CK(hipDeviceSynchronize());
  CK(hipMalloc(&out, (size_t)65536 * 256 * 16)); CK(hipMalloc(&dst, n4 * 16));
  int turn = 0;
  auto next = [&]() { turn = (turn + 1) % NB; return buf[turn]; };
  char name[128];
  for (int g : {256, 512, 1024, 2048, 4096}) {
    snprintf(name, sizeof name, "V1 stride, 4 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_stride<4>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 512, 1024, 2048}) {
    snprintf(name, sizeof name, "V1 stride, 8 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_stride<8>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 1024}) {
    snprintf(name, sizeof name, "V1 stride, 16 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_stride<16>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 1024, 4096, 50176}) {
    snprintf(name, sizeof name, "V2 16 KB chunks, 4 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_chunk<4>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 1024, 4096, 25088}) {
    snprintf(name, sizeof name, "V2 32 KB chunks, 8 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_chunk<8>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 1024}) {
    snprintf(name, sizeof name, "V3 stride, non-temporal, 4 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_stride_nt<4>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
    snprintf(name, sizeof name, "V3 stride, non-temporal, 8 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_stride_nt<8>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 512, 1024, 2048}) {
    snprintf(name, sizeof name, "V4 LDS-DMA, 4 copies in flight per wave, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_dma<4>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
    snprintf(name, sizeof name, "V4 LDS-DMA, 8 copies in flight per wave, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL(k_dma<8>, dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {256, 512}) {
    snprintf(name, sizeof name, "V5 k_stats arithmetic, 4 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL((k_stats_like<4, false>), dim3(g), dim3(256), 0, 0, next(), n4, out); });
    snprintf(name, sizeof name, "V5 k_stats arithmetic, 8 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL((k_stats_like<8, false>), dim3(g), dim3(256), 0, 0, next(), n4, out); });
    snprintf(name, sizeof name, "V5 k_stats arithmetic, non-temporal, 4 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL((k_stats_like<4, true>), dim3(g), dim3(256), 0, 0, next(), n4, out); });
    snprintf(name, sizeof name, "V5 k_stats arithmetic, non-temporal, 8 in flight, %d work-groups", g);
    timeit(name, bytes, [&] { hipLaunchKernelGGL((k_stats_like<8, true>), dim3(g), dim3(256), 0, 0, next(), n4, out); });
  }
  for (int g : {512, 1024, 2048}) {
    snprintf(name, sizeof name, "copy (read + write), 4 in flight, %d work-groups", g);
    timeit(name, 2 * bytes, [&] { hipLaunchKernelGGL(k_copy<4>, dim3(g), dim3(256), 0, 0, next(), dst, n4); });
  }
  return 0;
}
