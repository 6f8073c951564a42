// Multi-tensor updates that surround the hot path each step:
//   - momentum (EMA) update of the key encoder, contrast/models/PixPro.py:321-337
//   - LARS trust-ratio scaling + SGD-momentum step, contrast/lars.py:109-151 + torch.optim.SGD
// The reference issues ~3 tiny kernels per parameter for the EMA (165 parameters) and two
// norms plus a device->host sync per >=2-D parameter for LARS (56 syncs per step).  Here each
// is one or two launches over a device-side table of pointers: HBM-bound, 16 B per lane, the tensors cut into equal
// chunks so that every block streams the same amount (see ChunkWalk).
#include "common.h"

namespace pixpro {
namespace {

constexpr int TPB = 256;
constexpr int MAXT = 2048;                 // tensors per call the chunk table in LDS can hold
constexpr int UNR = 4;                     // float4 per thread and array in flight
constexpr int CHUNK = TPB * 4 * UNR * 4;   // elements per chunk: four batches of UNR float4 per thread
constexpr int GRID = 2048;                 // blocks: every block walks a contiguous range of chunks

// Work distribution (round 2).  Round 1 gave every tensor 64 blocks: the 2.4 M-element convolution weights then took 36
// dependent grid-stride iterations of one float4 per thread while the blocks of the 64-element BatchNorm vectors idled -
// latency-bound at 0.38-0.40 of the HBM rate.  Now the tensors are cut into chunks of 16 K elements, every block builds
// the prefix table of chunk counts in LDS (one coalesced read of `sizes`, a block scan), takes a contiguous range of the
// chunk list and streams it with UNR float4 per array in flight.  No host-side knowledge of the sizes is needed.
struct ChunkWalk {
  int t;          // current tensor
  int first, last;   // chunk range [first, last) of this block in the global chunk list
};

__device__ __forceinline__ ChunkWalk chunk_table(const int64_t* __restrict__ sizes, int count, int* starts /* [MAXT + 1] */) {
  __shared__ int wsum[TPB / WAVE];
  constexpr int PER = MAXT / TPB;          // entries per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c[PER], local = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int t = tid * PER + e;
    c[e] = t < count ? (int)((sizes[t] + CHUNK - 1) / CHUNK) : 0;
    local += c[e];
  }
  int incl = local;                        // inclusive scan over the block
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int run = base + incl - local;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int t = tid * PER + e;
    if (t <= count) starts[t] = run;       // starts[count] = total, for count < MAXT
    run += c[e];
  }
  if (tid == TPB - 1 && count == MAXT) starts[MAXT] = run;   // a full table: no thread's entries reach t == count
  __syncthreads();
  const int total = starts[count];
  ChunkWalk w;
  w.first = (int)((int64_t)total * blockIdx.x / gridDim.x);
  w.last = (int)((int64_t)total * (blockIdx.x + 1) / gridDim.x);
  int lo = 0, hi = count;                  // largest t with starts[t] <= first (empty tensors share a start: take the last)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= w.first) lo = mid; else hi = mid;
  }
  w.t = lo;
  return w;
}

// element range of chunk `ch` inside tensor t (advancing t as the walk crosses tensor boundaries)
__device__ __forceinline__ void chunk_at(ChunkWalk& w, const int* starts, int ch, int64_t& e0) {
  while (ch >= starts[w.t + 1]) ++w.t;
  e0 = (int64_t)(ch - starts[w.t]) * CHUNK;
}

__global__ void __launch_bounds__(TPB)
k_ema(const float* const* __restrict__ pq, float* const* __restrict__ pk, const int64_t* __restrict__ sizes, int count,
      float m, float om) {
  __shared__ int starts[MAXT + 1];
  ChunkWalk w = chunk_table(sizes, count, starts);
  for (int ch = w.first; ch < w.last; ++ch) {
    int64_t e0;
    chunk_at(w, starts, ch, e0);
    const int64_t n = sizes[w.t], e1 = min(e0 + CHUNK, n);
    const float* __restrict__ q = pq[w.t];
    float* __restrict__ k = pk[w.t];
    if (((((uintptr_t)q | (uintptr_t)k) & 15) == 0)) {
      const int64_t v1 = e1 / 4;             // whole float4 of this chunk (e0 is a multiple of 4)
      for (int64_t i = e0 / 4 + threadIdx.x; i < v1; i += (int64_t)TPB * UNR) {
        float4 a[UNR], b[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (i + u * TPB < v1) { a[u] = reinterpret_cast<const float4*>(q)[i + u * TPB]; b[u] = reinterpret_cast<float4*>(k)[i + u * TPB]; }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (i + u * TPB < v1) {
            b[u].x = b[u].x * m + a[u].x * om; b[u].y = b[u].y * m + a[u].y * om;
            b[u].z = b[u].z * m + a[u].z * om; b[u].w = b[u].w * m + a[u].w * om;
            reinterpret_cast<float4*>(k)[i + u * TPB] = b[u];
          }
      }
      for (int64_t i = v1 * 4 + threadIdx.x; i < e1; i += TPB) k[i] = k[i] * m + q[i] * om;
    } else {
      for (int64_t i = e0 + threadIdx.x; i < e1; i += TPB) k[i] = k[i] * m + q[i] * om;
    }
  }
}

// norms2[2t] += sum p^2 ; norms2[2t+1] += sum (g + wd p)^2     (only for adapt[t] != 0)
__global__ void __launch_bounds__(TPB)
k_lars_norms(float* const* __restrict__ pp, const float* const* __restrict__ gp, const int64_t* __restrict__ sizes,
             int count, const float* __restrict__ wd, const int32_t* __restrict__ adapt, float* __restrict__ norms2) {
  __shared__ int starts[MAXT + 1];
  __shared__ float red[2][TPB / WAVE];
  ChunkWalk wk = chunk_table(sizes, count, starts);
  float sp = 0.f, sg = 0.f;
  int cur = -1;
  auto flush = [&](int t) {                // block-uniform call: partial sums of tensor t -> global
    sp = wave_sum(sp); sg = wave_sum(sg);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sp; red[1][threadIdx.x >> 6] = sg; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f, b = 0.f;
      for (int i = 0; i < TPB / WAVE; ++i) { a += red[0][i]; b += red[1][i]; }
      if (a != 0.f) atomicAdd(&norms2[2 * t], a);
      if (b != 0.f) atomicAdd(&norms2[2 * t + 1], b);
    }
    sp = 0.f; sg = 0.f;
  };
  for (int ch = wk.first; ch < wk.last; ++ch) {
    int64_t e0;
    chunk_at(wk, starts, ch, e0);
    const int t = wk.t;
    if (t != cur) {
      if (cur >= 0 && adapt[cur]) flush(cur);
      cur = t;
    }
    if (!adapt[t]) continue;
    const int64_t n = sizes[t], e1 = min(e0 + CHUNK, n);
    const float* __restrict__ p = pp[t];
    const float* __restrict__ g = gp[t];
    const float w = wd[t];
    auto acc = [&](float pv, float gv0) {
      const float gv = w > 0.f ? __builtin_fmaf(w, pv, gv0) : gv0;
      sp = __builtin_fmaf(pv, pv, sp);
      sg = __builtin_fmaf(gv, gv, sg);
    };
    if (((((uintptr_t)p | (uintptr_t)g) & 15) == 0)) {
      const int64_t v1 = e1 / 4;
      for (int64_t i = e0 / 4 + threadIdx.x; i < v1; i += (int64_t)TPB * UNR) {
        float4 a[UNR], b[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (i + u * TPB < v1) { a[u] = reinterpret_cast<const float4*>(p)[i + u * TPB]; b[u] = reinterpret_cast<const float4*>(g)[i + u * TPB]; }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (i + u * TPB < v1) { acc(a[u].x, b[u].x); acc(a[u].y, b[u].y); acc(a[u].z, b[u].z); acc(a[u].w, b[u].w); }
      }
      for (int64_t i = v1 * 4 + threadIdx.x; i < e1; i += TPB) acc(p[i], g[i]);
    } else {
      for (int64_t i = e0 + threadIdx.x; i < e1; i += TPB) acc(p[i], g[i]);
    }
  }
  if (cur >= 0 && adapt[cur]) flush(cur);
}

__global__ void __launch_bounds__(TPB)
k_lars_update(float* const* __restrict__ pp, const float* const* __restrict__ gp, float* const* __restrict__ bp,
              const int64_t* __restrict__ sizes, int count, const float* __restrict__ wd,
              const int32_t* __restrict__ adapt, const float* __restrict__ norms2, float lr, float momentum, float trust,
              float eps, int first_step) {
  __shared__ int starts[MAXT + 1];
  ChunkWalk wk = chunk_table(sizes, count, starts);
  for (int ch = wk.first; ch < wk.last; ++ch) {
    int64_t e0;
    chunk_at(wk, starts, ch, e0);
    const int t = wk.t;
    const int64_t n = sizes[t], e1 = min(e0 + CHUNK, n);
    float* __restrict__ p = pp[t];
    const float* __restrict__ g = gp[t];
    float* __restrict__ buf = bp[t];
    const float w = wd[t];
    float ratio = 1.f;
    if (adapt[t]) {
      const float pn = sqrtf(norms2[2 * t]), gn = sqrtf(norms2[2 * t + 1]);
      if (pn > 0.f && gn > 0.f) ratio = trust * pn / (gn + eps);
    }
    auto upd = [&](float pv, float gv0, float bv0, float& pout, float& bout) {
      float gv = w > 0.f ? __builtin_fmaf(w, pv, gv0) : gv0;
      gv = gv * ratio;
      bout = first_step ? gv : bv0 * momentum + gv;
      pout = __builtin_fmaf(-lr, bout, pv);
    };
    if (((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0)) {
      const int64_t v1 = e1 / 4;
      for (int64_t i = e0 / 4 + threadIdx.x; i < v1; i += (int64_t)TPB * UNR) {
        float4 a[UNR], b[UNR], c[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (i + u * TPB < v1) {
            a[u] = reinterpret_cast<float4*>(p)[i + u * TPB];
            b[u] = reinterpret_cast<const float4*>(g)[i + u * TPB];
            c[u] = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i + u * TPB];
          }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (i + u * TPB < v1) {
            upd(a[u].x, b[u].x, c[u].x, a[u].x, c[u].x); upd(a[u].y, b[u].y, c[u].y, a[u].y, c[u].y);
            upd(a[u].z, b[u].z, c[u].z, a[u].z, c[u].z); upd(a[u].w, b[u].w, c[u].w, a[u].w, c[u].w);
            reinterpret_cast<float4*>(buf)[i + u * TPB] = c[u];
            reinterpret_cast<float4*>(p)[i + u * TPB] = a[u];
          }
      }
      for (int64_t i = v1 * 4 + threadIdx.x; i < e1; i += TPB) {
        float pv, bv;
        upd(p[i], g[i], first_step ? 0.f : buf[i], pv, bv);
        buf[i] = bv;
        p[i] = pv;
      }
    } else {
      for (int64_t i = e0 + threadIdx.x; i < e1; i += TPB) {
        float pv, bv;
        upd(p[i], g[i], first_step ? 0.f : buf[i], pv, bv);
        buf[i] = bv;
        p[i] = pv;
      }
    }
  }
}

__global__ void k_scale(float* __restrict__ buf, size_t n, const float* __restrict__ scale) {
  const float s = *scale;
  if (s == 1.f) return;
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) buf[i] = buf[i] * s;
}

}  // namespace

int run_ema(const float* const* pq, float* const* pk, const int64_t* sizes, int count, double m, hipStream_t st) {
  if (count > MAXT) { set_error("at most %d tensors per call", MAXT); return PIXPRO_ERR_ARG; }
  hipLaunchKernelGGL(k_ema, dim3(GRID), dim3(TPB), 0, st, pq, pk, sizes, count, (float)m, (float)(1.0 - m));
  return check_launch("k_ema");
}

int run_lars(float* const* pp, const float* const* gp, float* const* bp, const int64_t* sizes, const float* wd,
             const int32_t* adapt, int count, float lr, float momentum, float trust, float eps, int first_step,
             float* norms2, hipStream_t st) {
  if (count > MAXT) { set_error("at most %d tensors per call", MAXT); return PIXPRO_ERR_ARG; }
  if (hipMemsetAsync(norms2, 0, sizeof(float) * 2 * count, st) != hipSuccess) {
    set_error("hipMemsetAsync failed");
    return PIXPRO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_lars_norms, dim3(GRID), dim3(TPB), 0, st, pp, gp, sizes, count, wd, adapt, norms2);
  if (int e = check_launch("k_lars_norms")) return e;
  hipLaunchKernelGGL(k_lars_update, dim3(GRID), dim3(TPB), 0, st, pp, gp, bp, sizes, count, wd, adapt, norms2, lr,
                     momentum, trust, eps, first_step);
  return check_launch("k_lars_update");
}

int run_scale(float* buf, size_t n, const float* scale, hipStream_t st) {
  size_t blocks = (n + TPB - 1) / TPB;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) return PIXPRO_OK;
  hipLaunchKernelGGL(k_scale, dim3((unsigned)blocks), dim3(TPB), 0, st, buf, n, scale);
  return check_launch("k_scale");
}

}  // namespace pixpro
