// Multi-tensor updates that surround the hot path each step:
//   - momentum (EMA) update of the key encoder, contrast/models/PixPro.py:321-337
//   - LARS trust-ratio scaling + SGD-momentum step, contrast/lars.py:109-151 + torch.optim.SGD
// The reference issues ~3 tiny kernels per parameter for the EMA (165 parameters) and two
// norms plus a device->host sync per >=2-D parameter for LARS (56 syncs per step).  Here each
// is one or two launches over a device-side table of pointers: HBM-bound, 16 B per lane.
#include "common.h"

namespace pixpro {
namespace {

constexpr int TPB = 256;
constexpr int GX = 64;   // blocks per tensor (grid-stride inside)

__global__ void k_ema(const float* const* __restrict__ pq, float* const* __restrict__ pk,
                      const int64_t* __restrict__ sizes, float m, float om) {
  const int t = blockIdx.y;
  const int64_t n = sizes[t];
  const float* __restrict__ q = pq[t];
  float* __restrict__ k = pk[t];
  const int64_t n4 = (((uintptr_t)q | (uintptr_t)k) & 15) == 0 ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)GX * TPB) {
    const float4 a = reinterpret_cast<const float4*>(q)[i];
    float4 b = reinterpret_cast<float4*>(k)[i];
    b.x = b.x * m + a.x * om; b.y = b.y * m + a.y * om; b.z = b.z * m + a.z * om; b.w = b.w * m + a.w * om;
    reinterpret_cast<float4*>(k)[i] = b;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)GX * TPB)
    k[i] = k[i] * m + q[i] * om;
}

// norms2[2t] += sum p^2 ; norms2[2t+1] += sum (g + wd p)^2     (only for adapt[t] != 0)
__global__ void k_lars_norms(float* const* __restrict__ pp, const float* const* __restrict__ gp,
                             const int64_t* __restrict__ sizes, const float* __restrict__ wd,
                             const int32_t* __restrict__ adapt, float* __restrict__ norms2) {
  __shared__ float red[2][TPB / WAVE];
  const int t = blockIdx.y;
  if (!adapt[t]) return;
  const int64_t n = sizes[t];
  const float* __restrict__ p = pp[t];
  const float* __restrict__ g = gp[t];
  const float w = wd[t];
  float sp = 0.f, sg = 0.f;
  auto acc = [&](float pv, float gv0) {
    const float gv = w > 0.f ? __builtin_fmaf(w, pv, gv0) : gv0;
    sp = __builtin_fmaf(pv, pv, sp);
    sg = __builtin_fmaf(gv, gv, sg);
  };
  const int64_t n4 = (((uintptr_t)p | (uintptr_t)g) & 15) == 0 ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)GX * TPB) {
    const float4 a = reinterpret_cast<const float4*>(p)[i];
    const float4 b = reinterpret_cast<const float4*>(g)[i];
    acc(a.x, b.x); acc(a.y, b.y); acc(a.z, b.z); acc(a.w, b.w);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)GX * TPB) acc(p[i], g[i]);
  sp = wave_sum(sp); sg = wave_sum(sg);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sp; red[1][threadIdx.x >> 6] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < TPB / WAVE; ++i) { a += red[0][i]; b += red[1][i]; }
    if (a != 0.f) atomicAdd(&norms2[2 * t], a);
    if (b != 0.f) atomicAdd(&norms2[2 * t + 1], b);
  }
}

__global__ void k_lars_update(float* const* __restrict__ pp, const float* const* __restrict__ gp,
                              float* const* __restrict__ bp, const int64_t* __restrict__ sizes,
                              const float* __restrict__ wd, const int32_t* __restrict__ adapt,
                              const float* __restrict__ norms2, float lr, float momentum, float trust, float eps,
                              int first_step) {
  const int t = blockIdx.y;
  const int64_t n = sizes[t];
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
  const int64_t n4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0 ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)GX * TPB) {
    float4 a = reinterpret_cast<float4*>(p)[i];
    const float4 b = reinterpret_cast<const float4*>(g)[i];
    float4 c = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
    upd(a.x, b.x, c.x, a.x, c.x); upd(a.y, b.y, c.y, a.y, c.y); upd(a.z, b.z, c.z, a.z, c.z); upd(a.w, b.w, c.w, a.w, c.w);
    reinterpret_cast<float4*>(buf)[i] = c;
    reinterpret_cast<float4*>(p)[i] = a;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)GX * TPB) {
    float pv, bv;
    upd(p[i], g[i], first_step ? 0.f : buf[i], pv, bv);
    buf[i] = bv;
    p[i] = pv;
  }
}

__global__ void k_scale(float* __restrict__ buf, size_t n, const float* __restrict__ scale) {
  const float s = *scale;
  if (s == 1.f) return;
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) buf[i] = buf[i] * s;
}

}  // namespace

int run_ema(const float* const* pq, float* const* pk, const int64_t* sizes, int count, double m, hipStream_t st) {
  hipLaunchKernelGGL(k_ema, dim3(GX, count), dim3(TPB), 0, st, pq, pk, sizes, (float)m, (float)(1.0 - m));
  return check_launch("k_ema");
}

int run_lars(float* const* pp, const float* const* gp, float* const* bp, const int64_t* sizes, const float* wd,
             const int32_t* adapt, int count, float lr, float momentum, float trust, float eps, int first_step,
             float* norms2, hipStream_t st) {
  if (hipMemsetAsync(norms2, 0, sizeof(float) * 2 * count, st) != hipSuccess) {
    set_error("hipMemsetAsync failed");
    return PIXPRO_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_lars_norms, dim3(GX, count), dim3(TPB), 0, st, pp, gp, sizes, wd, adapt, norms2);
  if (int e = check_launch("k_lars_norms")) return e;
  hipLaunchKernelGGL(k_lars_update, dim3(GX, count), dim3(TPB), 0, st, pp, gp, bp, sizes, wd, adapt, norms2, lr,
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
