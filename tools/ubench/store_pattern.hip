// Store-pattern probe for the dense outputs of the flow preparation: the same 4.31 GB (N = 256 frames of
// 720 x 1280: two [N,2,Hf,Wf] fp32 flows + two [N,Hf,Wf] byte masks) written with nothing computed, in the shapes a
// kernel could use.  Which shape reaches the plain-fill rate decides the thread mapping of k_prepare_cert.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/store_pattern tools/ubench/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int Hf = 720, Wf = 1280;
constexpr size_t HW = (size_t)Hf * Wf;

// V0: linear 16 B / lane fill of one buffer
__global__ void k_fill(float4* p, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// V0b: as torch's fill: every work-group writes one contiguous 16 KB piece, no loop
__global__ void __launch_bounds__(256) k_fill16k(float4* p, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
  const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (b + k * 256 < n4) p[b + k * 256] = v;
}

// Full-width row bands: a work-group of NW waves covers NW*256 pixels of ROWS rows.  PLANES = 1: one fp32 plane per
// work-group (blockIdx.y = plane); PLANES = 4: the four flow planes of a frame; MASKS: + the two mask planes.
template <int NW, int ROWS, int PLANES, bool MASKS>
__global__ void __launch_bounds__(NW * 64) k_band(float* ff, float* fb, uint8_t* mf, uint8_t* mb, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = blockIdx.x * ROWS;
  const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
  for (int xb = wave * 256; xb < Wf; xb += NW * 256) {
    const int x = xb + lane * 4;
    if (x >= Wf) break;
    if (PLANES == 1) {
      const int pl = blockIdx.y;                         // 0 .. 4N-1
      float* o = (pl < 2 * N ? ff + (size_t)pl * HW : fb + (size_t)(pl - 2 * N) * HW) + x;
      for (int r = 0; r < ROWS; ++r) {
        const int y = y0 + r;
        if (y >= Hf) break;
        *reinterpret_cast<float4*>(o + (size_t)y * Wf) = v;
      }
    } else {
      const int n = blockIdx.y;
      float* of = ff + (size_t)n * 2 * HW + x;
      float* ob = fb + (size_t)n * 2 * HW + x;
      for (int r = 0; r < ROWS; ++r) {
        const int y = y0 + r;
        if (y >= Hf) break;
        const size_t q = (size_t)y * Wf;
        *reinterpret_cast<float4*>(of + q) = v; *reinterpret_cast<float4*>(of + HW + q) = v;
        *reinterpret_cast<float4*>(ob + q) = v; *reinterpret_cast<float4*>(ob + HW + q) = v;
        if (MASKS) {
          *reinterpret_cast<uint32_t*>(mf + (size_t)n * HW + q + x) = 0x01010101u;
          *reinterpret_cast<uint32_t*>(mb + (size_t)n * HW + q + x) = 0x01010101u;
        }
      }
    }
  }
}
// one mask plane per work-group, full-width bands, 4 B per lane
template <int NW, int ROWS>
__global__ void __launch_bounds__(NW * 64) k_band_mask(uint8_t* mf, uint8_t* mb, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = blockIdx.x * ROWS, pl = blockIdx.y;
  uint8_t* o = pl < N ? mf + (size_t)pl * HW : mb + (size_t)(pl - N) * HW;
  for (int xb = wave * 256; xb < Wf; xb += NW * 256) {
    const int x = xb + lane * 4;
    if (x >= Wf) break;
    for (int r = 0; r < ROWS; ++r) {
      const int y = y0 + r;
      if (y >= Hf) break;
      *reinterpret_cast<uint32_t*>(o + (size_t)y * Wf + x) = 0x01010101u;
    }
  }
}
// 256-pixel tiles of ONE plane per work-group
template <int NW, int ROWS>
__global__ void __launch_bounds__(NW * 64) k_quad1(float* ff, float* fb, int N) {
  const int pl = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 256 + lane * 4, y0 = blockIdx.y * (NW * ROWS) + wave * ROWS;
  float* o = (pl < 2 * N ? ff + (size_t)pl * HW : fb + (size_t)(pl - 2 * N) * HW) + x;
  const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
  for (int r = 0; r < ROWS; ++r) {
    const int y = y0 + r;
    if (y >= Hf) break;
    *reinterpret_cast<float4*>(o + (size_t)y * Wf) = v;
  }
}

// dynamic LDS only limits how many work-groups share a CU (the real kernels hold 28-60 KB each)
extern __shared__ float occupancy_pad[];
__device__ __forceinline__ void touch_pad() { if (threadIdx.x == 9999) occupancy_pad[0] = 1.f; }

// V4: lane owns 2 consecutive pixels: tile 128 x (NW*ROWS), 8-byte flow stores, 2-byte mask stores
template <int NW, int ROWS>
__global__ void __launch_bounds__(NW * 64) k_pair(float* ff, float* fb, uint8_t* mf, uint8_t* mb) {
  touch_pad();
  const int n = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 128 + lane * 2, y0 = blockIdx.y * (NW * ROWS) + wave * ROWS;
  float* of = ff + (size_t)n * 2 * HW + x;
  float* ob = fb + (size_t)n * 2 * HW + x;
  const float2 v = make_float2((float)lane, 1.f);
  for (int r = 0; r < ROWS; ++r) {
    const int y = y0 + r;
    if (y >= Hf) break;
    const size_t q = (size_t)y * Wf;
    *reinterpret_cast<float2*>(of + q) = v; *reinterpret_cast<float2*>(of + HW + q) = v;
    *reinterpret_cast<float2*>(ob + q) = v; *reinterpret_cast<float2*>(ob + HW + q) = v;
    *reinterpret_cast<uint16_t*>(mf + (size_t)n * HW + q + x) = 0x0101u;
    *reinterpret_cast<uint16_t*>(mb + (size_t)n * HW + q + x) = 0x0101u;
  }
}

// V1: today's mapping: tile 64 x 80, 5 waves x 16 rows, lane = column, 4-byte flow stores, 1-byte mask stores
template <bool MASKS>
__global__ void __launch_bounds__(320) k_col1(float* ff, float* fb, uint8_t* mf, uint8_t* mb) {
  touch_pad();
  const int n = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane, y0 = blockIdx.y * 80 + wave * 16;
  float* of = ff + (size_t)n * 2 * HW + x;
  float* ob = fb + (size_t)n * 2 * HW + x;
  const float v = (float)lane;
  for (int r = 0; r < 16; ++r) {
    const int y = y0 + r;
    if (y >= Hf) break;
    const size_t q = (size_t)y * Wf;
    of[q] = v; of[HW + q] = v + 1.f; ob[q] = v + 2.f; ob[HW + q] = v + 3.f;
    if (MASKS) { mf[(size_t)n * HW + q + x] = 1; mb[(size_t)n * HW + q + x] = 1; }
  }
}

// V3: lane owns 4 consecutive pixels: tile (64*4) x (NW*ROWS), 16-byte flow stores, 4-byte mask stores
template <int NW, int ROWS, bool MASKS>
__global__ void __launch_bounds__(NW * 64) k_quad(float* ff, float* fb, uint8_t* mf, uint8_t* mb) {
  touch_pad();
  const int n = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 256 + lane * 4, y0 = blockIdx.y * (NW * ROWS) + wave * ROWS;
  float* of = ff + (size_t)n * 2 * HW + x;
  float* ob = fb + (size_t)n * 2 * HW + x;
  const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
  for (int r = 0; r < ROWS; ++r) {
    const int y = y0 + r;
    if (y >= Hf) break;
    const size_t q = (size_t)y * Wf;
    *reinterpret_cast<float4*>(of + q) = v; *reinterpret_cast<float4*>(of + HW + q) = v;
    *reinterpret_cast<float4*>(ob + q) = v; *reinterpret_cast<float4*>(ob + HW + q) = v;
    if (MASKS) {
      *reinterpret_cast<uint32_t*>(mf + (size_t)n * HW + q + x) = 0x01010101u;
      *reinterpret_cast<uint32_t*>(mb + (size_t)n * HW + q + x) = 0x01010101u;
    }
  }
}

// V5: 64-pixel-wide tile, 16 lanes x 4 pixels per row, a wave instruction covers 4 rows (ROWS apart)
template <int ROWS, bool MASKS>
__global__ void __launch_bounds__(320) k_quad64(float* ff, float* fb, uint8_t* mf, uint8_t* mb) {
  const int n = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + (lane & 15) * 4, y0 = blockIdx.y * (5 * 4 * ROWS) + (wave * 4 + (lane >> 4)) * ROWS;
  float* of = ff + (size_t)n * 2 * HW + x;
  float* ob = fb + (size_t)n * 2 * HW + x;
  const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
  for (int r = 0; r < ROWS; ++r) {
    const int y = y0 + r;
    if (y < Hf) {
      const size_t q = (size_t)y * Wf;
      *reinterpret_cast<float4*>(of + q) = v; *reinterpret_cast<float4*>(of + HW + q) = v;
      *reinterpret_cast<float4*>(ob + q) = v; *reinterpret_cast<float4*>(ob + HW + q) = v;
      if (MASKS) {
        *reinterpret_cast<uint32_t*>(mf + (size_t)n * HW + q + x) = 0x01010101u;
        *reinterpret_cast<uint32_t*>(mb + (size_t)n * HW + q + x) = 0x01010101u;
      }
    }
  }
}

template <class F>
static float timeit(F&& f, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 256;
  float *ff, *fb; uint8_t *mf, *mb;
  const size_t fbytes = (size_t)N * 2 * HW * 4, mbytes = (size_t)N * HW;
  CK(hipMalloc(&ff, fbytes)); CK(hipMalloc(&fb, fbytes)); CK(hipMalloc(&mf, mbytes)); CK(hipMalloc(&mb, mbytes));
  const double total = 2.0 * fbytes + 2.0 * mbytes, flows = 2.0 * fbytes;
  auto rep = [&](const char* name, float ms, double bytes) {
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6);
  };
  rep("V0 linear fill 16 B/lane (all bytes)", timeit([&] {
        k_fill<<<4096, 256>>>((float4*)ff, fbytes / 16); k_fill<<<4096, 256>>>((float4*)fb, fbytes / 16);
        k_fill<<<1024, 256>>>((float4*)mf, mbytes / 16); k_fill<<<1024, 256>>>((float4*)mb, mbytes / 16); }), total);
  rep("V0 linear fill 16 B/lane (flows only)", timeit([&] {
        k_fill<<<4096, 256>>>((float4*)ff, fbytes / 16); k_fill<<<4096, 256>>>((float4*)fb, fbytes / 16); }), flows);
  rep("V0b 16 KB contiguous per work-group (all bytes)", timeit([&] {
        k_fill16k<<<fbytes / 16384, 256>>>((float4*)ff, fbytes / 16); k_fill16k<<<fbytes / 16384, 256>>>((float4*)fb, fbytes / 16);
        k_fill16k<<<mbytes / 16384, 256>>>((float4*)mf, mbytes / 16); k_fill16k<<<mbytes / 16384, 256>>>((float4*)mb, mbytes / 16); }), total);
  rep("PA1 band 5w x 16 rows, one plane / WG (flows)", timeit([&] {
        k_band<5, 16, 1, false><<<dim3(Hf / 16, 4 * N), 320>>>(ff, fb, mf, mb, N); }), flows);
  rep("PA2 band 5w x 80 rows, one plane / WG (flows)", timeit([&] {
        k_band<5, 80, 1, false><<<dim3(Hf / 80, 4 * N), 320>>>(ff, fb, mf, mb, N); }), flows);
  rep("PA3 band 5w x 4 rows, one plane / WG (flows)", timeit([&] {
        k_band<5, 4, 1, false><<<dim3(Hf / 4, 4 * N), 320>>>(ff, fb, mf, mb, N); }), flows);
  rep("PB1 band 5w x 16 rows, 4 planes / WG (flows)", timeit([&] {
        k_band<5, 16, 4, false><<<dim3(Hf / 16, N), 320>>>(ff, fb, mf, mb, N); }), flows);
  rep("PB2 band 5w x 80 rows, 4 planes / WG (flows)", timeit([&] {
        k_band<5, 80, 4, false><<<dim3(Hf / 80, N), 320>>>(ff, fb, mf, mb, N); }), flows);
  rep("PC1 band 5w x 16 rows, 4 planes + masks", timeit([&] {
        k_band<5, 16, 4, true><<<dim3(Hf / 16, N), 320>>>(ff, fb, mf, mb, N); }), total);
  rep("PC2 band 5w x 80 rows, 4 planes + masks", timeit([&] {
        k_band<5, 80, 4, true><<<dim3(Hf / 80, N), 320>>>(ff, fb, mf, mb, N); }), total);
  rep("PC3 band 5w x 8 rows, 4 planes + masks", timeit([&] {
        k_band<5, 8, 4, true><<<dim3(Hf / 8, N), 320>>>(ff, fb, mf, mb, N); }), total);
  rep("PD1 256x80 tile, one plane / WG (flows)", timeit([&] {
        k_quad1<5, 16><<<dim3(Wf / 256, Hf / 80, 4 * N), 320>>>(ff, fb, N); }), flows);
  rep("PM1 band 5w x 16 rows, one mask plane / WG", timeit([&] {
        k_band_mask<5, 16><<<dim3(Hf / 16, 2 * N), 320>>>(mf, mb, N); }), 2.0 * mbytes);
  rep("PA1 + PM1 (two launches, all bytes)", timeit([&] {
        k_band<5, 16, 1, false><<<dim3(Hf / 16, 4 * N), 320>>>(ff, fb, mf, mb, N);
        k_band_mask<5, 16><<<dim3(Hf / 16, 2 * N), 320>>>(mf, mb, N); }), total);
  const dim3 g1(Wf / 64, (Hf + 79) / 80, N);
  rep("V1 64x80 tile, 4 B flows + 1 B masks", timeit([&] { k_col1<true><<<g1, 320>>>(ff, fb, mf, mb); }), total);
  rep("V2 64x80 tile, 4 B flows only", timeit([&] { k_col1<false><<<g1, 320>>>(ff, fb, mf, mb); }), flows);
  rep("V3a 256x80 tile (4w x 20r), 16 B + 4 B masks", timeit([&] {
        k_quad<4, 20, true><<<dim3(Wf / 256, (Hf + 79) / 80, N), 256>>>(ff, fb, mf, mb); }), total);
  rep("V3b 256x40 tile (4w x 10r), 16 B + 4 B masks", timeit([&] {
        k_quad<4, 10, true><<<dim3(Wf / 256, (Hf + 39) / 40, N), 256>>>(ff, fb, mf, mb); }), total);
  rep("V3c 256x16 tile (2w x 8r), 16 B + 4 B masks", timeit([&] {
        k_quad<2, 8, true><<<dim3(Wf / 256, (Hf + 15) / 16, N), 128>>>(ff, fb, mf, mb); }), total);
  rep("V3d 256x80 tile (5w x 16r), 16 B + 4 B masks", timeit([&] {
        k_quad<5, 16, true><<<dim3(Wf / 256, (Hf + 79) / 80, N), 320>>>(ff, fb, mf, mb); }), total);
  rep("V3e 256x80 tile, 16 B flows only", timeit([&] {
        k_quad<4, 20, false><<<dim3(Wf / 256, (Hf + 79) / 80, N), 256>>>(ff, fb, mf, mb); }), flows);
  rep("V5a 64x80 tile, 16 lanes x 4 px, 4 rows/instr", timeit([&] {
        k_quad64<4, true><<<dim3(Wf / 64, (Hf + 79) / 80, N), 320>>>(ff, fb, mf, mb); }), total);
  rep("V5b 64x160 tile, 16 lanes x 4 px, 8 rows/lane", timeit([&] {
        k_quad64<8, true><<<dim3(Wf / 64, (Hf + 159) / 160, N), 320>>>(ff, fb, mf, mb); }), total);
  for (int lds : {30 << 10}) {
    printf("-- %d KB of LDS per work-group\n", lds >> 10);
    rep("V1 64x80 tile, 4 B flows + 1 B masks", timeit([&] { k_col1<true><<<g1, 320, lds>>>(ff, fb, mf, mb); }), total);
    rep("V2 64x80 tile, 4 B flows only", timeit([&] { k_col1<false><<<g1, 320, lds>>>(ff, fb, mf, mb); }), flows);
    rep("V3a 256x80 tile (4w x 20r), 16 B + 4 B masks", timeit([&] {
          k_quad<4, 20, true><<<dim3(Wf / 256, (Hf + 79) / 80, N), 256, lds>>>(ff, fb, mf, mb); }), total);
    rep("V3d 256x80 tile (5w x 16r), 16 B + 4 B masks", timeit([&] {
          k_quad<5, 16, true><<<dim3(Wf / 256, (Hf + 79) / 80, N), 320, lds>>>(ff, fb, mf, mb); }), total);
    rep("V3f 256x40 tile (5w x 8r), 16 B + 4 B masks", timeit([&] {
          k_quad<5, 8, true><<<dim3(Wf / 256, (Hf + 39) / 40, N), 320, lds>>>(ff, fb, mf, mb); }), total);
    rep("V4a 128x80 tile (5w x 16r), 8 B + 2 B masks", timeit([&] {
          k_pair<5, 16><<<dim3(Wf / 128, (Hf + 79) / 80, N), 320, lds>>>(ff, fb, mf, mb); }), total);
  }
  return 0;
}
