// Pieces shared by the two single-launch PPM / loss kernels (ppm_fused.hip: [C][P] LDS image, any P <= 64;
// ppm_fused_nhwc.hip: channels_last-native [P][C] image for the 7 x 7 grid at C = 256).
#pragma once
#include "common.h"

namespace pixpro {
namespace fused {

constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * WAVE;
constexpr int LS = 81;    // row stride of the small P x P matrices: odd, so rows AND columns (dS^T) spread over the banks

// Cross-wave column reduction with ONE barrier: on return every lane holds in v[n] the sum over all C rows of
// its own column 16n + (lane & 15).  Consecutive calls must alternate between two scratch buffers.
template <int NT>
__device__ __forceinline__ void col_reduce_reg(float (&v)[NT], float* part) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float s = v[n];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) part[wave * 64 + 16 * n + lane] = s;
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += part[w * 64 + 16 * n + lr];
    v[n] = t;
  }
}

// The same with one more column (the 49th of the 7 x 7 grid, kept outside the MFMA tile grid): `ve` is this lane's partial
// for it (0 in lanes that hold none) and comes back as its total in every lane.  Slot 48 of a wave's slice is free
// because NT = 3 column tiles use slots 0..47.
template <int NT>
__device__ __forceinline__ void col_reduce_edge(float (&v)[NT], float& ve, float* part, int wave, int lane) {
  static_assert(NT == 3, "slot 48 is the first free one");
  const int lr = lane & 15;
  const float se = wave_sum(ve);
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float s = v[n];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) part[wave * 64 + 16 * n + lane] = s;
  }
  if (lane == 0) part[wave * 64 + 48] = se;
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += part[w * 64 + 16 * n + lr];
    v[n] = t;
  }
  float te = 0.f;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) te += part[w * 64 + 48];
  ve = te;
}

template <int NT>
__device__ __forceinline__ void col_reduce_edge(float (&v)[NT], float& ve, float* part) {
  col_reduce_edge<NT>(v, ve, part, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
}

// C x P contraction: acc[m][n] += sum_k Aop(m, k) * Bop(n, k), k in steps of 4.  Software-pipelined by hand: the
// operands of step k+1 are requested from LDS before the MFMAs of step k issue (hipcc otherwise places every ds_read
// right in front of its consumer with an lgkmcnt(0) wait).
template <int MT, int NT, class FA, class FS, class FB>
__device__ __forceinline__ void gemm_cp(f32x4 (&acc)[MT][NT], int ksteps, FA fa, FS fs, FB fb) {
  // fa(m, ks): raw A element, fs(ks): its per-column scale (applied when the MFMA issues, not when the data is
  // requested), fb(n, ks): B element
  float a[MT], b[NT], sc;
#pragma unroll
  for (int m = 0; m < MT; ++m) a[m] = fa(m, 0);
#pragma unroll
  for (int n = 0; n < NT; ++n) b[n] = fb(n, 0);
  sc = fs(0);
  for (int ks = 0; ks < ksteps; ++ks) {
    float an[MT], bn[NT], scn;
    const int kn = ks + 1 < ksteps ? ks + 1 : ks;      // last step re-reads its own operands (never used)
#pragma unroll
    for (int m = 0; m < MT; ++m) an[m] = fa(m, kn);
#pragma unroll
    for (int n = 0; n < NT; ++n) bn[n] = fb(n, kn);
    scn = fs(kn);
    __builtin_amdgcn_sched_barrier(0);                 // keep the requests above ahead of the MFMAs below
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float am = a[m] * sc;
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = mfma16(am, b[n], acc[m][n]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = an[m];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = bn[n];
    sc = scn;
  }
}

// Positive-pair test `sqrtf(d2) / diag < r` (PixPro.py:217-219) as `d2 < T`: correctly rounded sqrt and division are
// monotone, so the set of d2 that pass is a down-set of the non-negative floats; T is its supremum, found by a 64-ary
// search over the bit patterns (6 rounds of one sqrt + one division per lane, run by one wave while the tiles are in
// flight).  The 49 x 49 tests then cost two multiplies, an add and a compare each - and decide exactly as the reference.
__device__ __forceinline__ float dist_threshold(float diag, float r, int lane) {
  if (!(r == r) || !(diag > 0.f)) return 0.f;          // NaN ratio, zero / NaN diagonal: nothing passes
  long long lo = -1, hi = 0x7f800000LL + 1;            // test(lo) passes (or lo = -1); test(hi) fails (or hi = sentinel)
  {
    // One round usually settles it: T is within a few ulps of (r diag)^2, so test the 64 neighbouring bit patterns and
    // keep the bracket they prove (pattern c0 passes and c0 + k fails  =>  lo = c0 + k - 1, hi = c0 + k).
    const float t0 = (r * diag) * (r * diag);
    if (t0 > 0.f && t0 < 3.0e38f) {
      const long long c0 = (long long)__float_as_uint(t0) - 32;
      const long long c = c0 + lane;
      const bool in = c >= 0 && c <= 0x7f800000LL;
      bool fails = true;
      if (in) fails = (sqrtf(__uint_as_float((unsigned)c)) / diag) >= r;
      const unsigned long long m = __ballot(fails);
      const unsigned long long inm = __ballot(in);
      if (inm == ~0ull && !(m & 1ull) && m != 0) {      // first pattern passes, a later one fails
        const int first = __ffsll((long long)m) - 1;
        lo = c0 + first - 1; hi = c0 + first;
      } else if (inm == ~0ull && m == 0) {
        lo = c0 + 63;                                   // all pass: the search below continues above the window
      } else if (inm == ~0ull && (m & 1ull)) {
        hi = c0;                                        // all fail (monotone): continues below the window
      }
    }
  }
  while (hi - lo > 1) {
    const long long step = (hi - lo - 1 + 63) / 64;
    const long long c = lo + (long long)(lane + 1) * step;
    const bool in = c < hi;
    bool fails = true;
    if (in) fails = (sqrtf(__uint_as_float((unsigned)c)) / diag) >= r;
    const unsigned long long m = __ballot(fails);
    const int first = m ? __ffsll((long long)m) - 1 : 64;
    const long long nlo = lo + (long long)first * step;
    const long long c1 = lo + (long long)(first + 1) * step;
    hi = first < 64 ? (c1 < hi ? c1 : hi) : hi;
    lo = nlo;
  }
  return __uint_as_float(hi > 0x7f800000LL ? 0x7f800000u : (unsigned)hi);
}

}  // namespace fused
}  // namespace pixpro
