// Single-launch PPM -> flow-matched pixel loss -> backward for grids with P = H*W <= 64 and C * P <= 12,544
// (7x7 at 224^2 crops, 2x2 at 64^2) and C in {128, 256}.
//
// One 512-thread work-group (8 wave64) per (loss direction, sample).  The sample's [C][P] fp32
// tiles are contiguous in HBM (NCHW: one linear 16 B/lane stream; channels_last: 4 pixels x 64 channels per
// wave-load, transposed through LDS); waves 1..7 stream them, ONE tile request at a time (x -> v -> k -> x again),
// wave 0 works out the sample points meanwhile (one round trip for all flow taps, consumed after the S contraction).
// All six P x P / C x P contractions run on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32 with both
// operands read from LDS; S/A/dS never leave the CU.  The tiles stay RAW in LDS: the three L2
// normalisations over C are applied as per-column reciprocal scales on the MFMA operands / folded into
// the epilogues' per-column factors (n_x^2 is the diagonal of the raw Gram matrix), and the two projection dot
// products are reduced on the accumulator registers (wave shuffles + one LDS hop across the 8 waves).
// 7 x 7 specialisation (PC = 49): S as six mirrored upper tiles + row 48 on the vector ALU; the four C x P products on
// three column tiles + column 48 as a dot product per row (edge_column).  DESIGN.md 4.1 has the measurements.
//
// Buffers over time (B0, B1 = two [C][P] LDS tiles):
//   B0: x (S)  -> k (G) -> dY (dA, dV) -> x (dX, dx out)         B1: v (Y, dA, dV epilogue, dv out)
//
// Math: SURVEY.md Appendix A; reference contrast/models/PixPro.py:339-363,380,403 (PPM + the three
// F.normalize), :92-247 (regression_loss), :46-89 (add_optical_flow).
#include "ppm_fused_shared.h"

namespace pixpro {

#ifdef PIXPRO_STAMPS   // diagnostic build only (tools/ppm_phase_stamps.py): wave 0 of block 0 stamps s_memtime
__device__ unsigned long long g_stamps[64];
#define STAMP(i)                                                                            \
  do {                                                                                      \
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime();    \
  } while (0)
#define STAMPW(i, t)                                                                        \
  do {                                                                                      \
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == (t)) g_stamps[i] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define STAMP(i)
#define STAMPW(i, t)
#endif

namespace {

using namespace fused;
constexpr int NLOAD = NTHREADS - WAVE;   // waves 1..7 stream the tiles; wave 0 works out the sample points meanwhile
constexpr int PF = 7;     // float4 prefetch registers per loader thread: PF * NLOAD * 4 = 256 * 49 exactly
constexpr int KU = 8;     // k-unroll of the K = C contractions (operand loads issued KU steps ahead)

struct Smem {
  float* B0;     // [C*P + 4]
  float* B1;     // [C*P + 4]
  float* AA;     // [LR][LS]   A = pow(clamp(S))
  float* Dm;     // [LR][LS]   dA/dS, later dS + dS^T
  float* DS;     // [LR][LS]   dS
  float* rx;     // [64] reciprocal column norms of x and v (those of k and Y live in registers / wave-private slices)
  float* rv;
  float* nx2;    // [64] diag of the raw Gram matrix
  float* part0;  // [NWAVES][64] x2: scratch of the column reductions, used alternately
  float* part1;
  float* qx;     // [64] each
  float* qy;
  float* kx;
  float* ky;
  float* misc;   // [32]: diag, cnt, -, ..., [8 + w] = per-wave positive counts
  unsigned long long* mbits;  // [64]
  int* valid;    // [64]
};

__host__ __device__ inline int lr_rows(int P) { return (P + 3) / 4 * 4; }

__host__ __device__ inline size_t smem_floats(int C, int P) {
  return 2 * ((size_t)C * P + 4) + 3 * (size_t)lr_rows(P) * LS + 3 * 64 + 2 * NWAVES * 64 + 4 * 64 + 32 + 2 * 64 + 64;
}

__device__ __forceinline__ Smem carve(float* base, int C, int P) {
  Smem s;
  const int LR = lr_rows(P);
  float* p = base;
  s.B0 = p; p += (size_t)C * P + 4;
  s.B1 = p; p += (size_t)C * P + 4;
  s.AA = p; p += LR * LS;
  s.Dm = p; p += LR * LS;
  s.DS = p; p += LR * LS;
  s.rx = p; p += 64; s.rv = p; p += 64;
  s.nx2 = p; p += 64;
  s.part0 = p; p += NWAVES * 64; s.part1 = p; p += NWAVES * 64;
  s.qx = p; p += 64; s.qy = p; p += 64; s.kx = p; p += 64; s.ky = p; p += 64;
  s.misc = p; p += 32;
  s.mbits = (unsigned long long*)p; p += 2 * 64;
  s.valid = (int*)p;
  return s;
}

// HBM -> registers (issue only) and registers -> LDS of one [C][P] tile (count % 4 == 0, 16-B aligned), executed by
// the 448 threads of waves 1..7.  Seven NAMED float4 per thread (an indexed array ends up in scratch memory, which turns
// the prefetch into a synchronous round trip through HBM).  Loads are unconditional on a clamped index; the commit is
// guarded.
#define PF_DECL(p) float4 p##0, p##1, p##2, p##3, p##4, p##5, p##6
#define PF_ISSUE(p, src, count)                                                    \
  do {                                                                             \
    const float4* s4_ = reinterpret_cast<const float4*>(src);                      \
    const int n4_ = (count) / 4 - 1, t_ = (int)threadIdx.x - WAVE;                 \
    p##0 = s4_[min(t_, n4_)];              p##1 = s4_[min(t_ + NLOAD, n4_)];       \
    p##2 = s4_[min(t_ + 2 * NLOAD, n4_)];  p##3 = s4_[min(t_ + 3 * NLOAD, n4_)];   \
    p##4 = s4_[min(t_ + 4 * NLOAD, n4_)];  p##5 = s4_[min(t_ + 5 * NLOAD, n4_)];   \
    p##6 = s4_[min(t_ + 6 * NLOAD, n4_)];                                          \
  } while (0)
#define PF_COMMIT(p, dst, count)                                                   \
  do {                                                                             \
    float4* d4_ = reinterpret_cast<float4*>(dst);                                  \
    const int n4_ = (count) / 4, t_ = (int)threadIdx.x - WAVE;                     \
    if (t_ < n4_) d4_[t_] = p##0;                                                  \
    if (t_ + NLOAD < n4_) d4_[t_ + NLOAD] = p##1;                                  \
    if (t_ + 2 * NLOAD < n4_) d4_[t_ + 2 * NLOAD] = p##2;                          \
    if (t_ + 3 * NLOAD < n4_) d4_[t_ + 3 * NLOAD] = p##3;                          \
    if (t_ + 4 * NLOAD < n4_) d4_[t_ + 4 * NLOAD] = p##4;                          \
    if (t_ + 5 * NLOAD < n4_) d4_[t_ + 5 * NLOAD] = p##5;                          \
    if (t_ + 6 * NLOAD < n4_) d4_[t_ + 6 * NLOAD] = p##6;                          \
  } while (0)
// NHWC (channels_last) tiles: the sample is [P][C] in HBM.  One wave-load covers 4 pixels x 16 groups of 4 channels
// (4 x 256-B segments); the 4 x 64 scalar LDS writes that transpose it into the [C][P] image the NCHW path uses fall
// into 64 different banks (word address (4 c4 + q) P + p, P odd).  Loads are numbered L = (wave - 1) + 7 i: first the
// (P / 4) x (C4 / 16) full ones, then one per left-over pixel (a whole row of C4 <= 64 groups); 49 loads cover
// 49 x 256 exactly.
struct NhwcSlot {
  int p, c4;
  bool ok;
};
__device__ __forceinline__ NhwcSlot nhwc_slot(int i, int C4, int P) {
  const int wave = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int L = (wave - 1) + (NWAVES - 1) * i;
  const int CG = C4 >> 4, full = (P >> 2) * CG;
  NhwcSlot s;
  if (L < full) {
    s.p = 4 * (L / CG) + (lane >> 4);
    s.c4 = 16 * (L % CG) + (lane & 15);
    s.ok = true;
  } else {
    s.p = (P & ~3) + (L - full);
    s.c4 = lane;
    s.ok = s.p < P && lane < C4;
  }
  return s;
}
#define PFT_GET_(v, i, s4, C4, P)                                                         \
  do {                                                                                    \
    const NhwcSlot s_ = nhwc_slot(i, C4, P);                                              \
    v = (s4)[min(s_.p, (P) - 1) * (C4) + min(s_.c4, (C4) - 1)];                           \
  } while (0)
#define PFT_ISSUE(p, src, C4, P)                                                          \
  do {                                                                                    \
    const float4* s4_ = reinterpret_cast<const float4*>(src);                             \
    PFT_GET_(p##0, 0, s4_, C4, P); PFT_GET_(p##1, 1, s4_, C4, P); PFT_GET_(p##2, 2, s4_, C4, P); \
    PFT_GET_(p##3, 3, s4_, C4, P); PFT_GET_(p##4, 4, s4_, C4, P); PFT_GET_(p##5, 5, s4_, C4, P); \
    PFT_GET_(p##6, 6, s4_, C4, P);                                                        \
  } while (0)
#define PFT_PUT_(v, i, dst, C4, P)                                                        \
  do {                                                                                    \
    const NhwcSlot s_ = nhwc_slot(i, C4, P);                                              \
    if (s_.ok) {                                                                          \
      float* d_ = (dst) + 4 * s_.c4 * (P) + s_.p;                                         \
      d_[0] = v.x; d_[P] = v.y; d_[2 * (P)] = v.z; d_[3 * (P)] = v.w;                     \
    }                                                                                     \
  } while (0)
#define PFT_COMMIT(p, dst, C4, P)                                                         \
  do {                                                                                    \
    PFT_PUT_(p##0, 0, dst, C4, P); PFT_PUT_(p##1, 1, dst, C4, P); PFT_PUT_(p##2, 2, dst, C4, P); \
    PFT_PUT_(p##3, 3, dst, C4, P); PFT_PUT_(p##4, 4, dst, C4, P); PFT_PUT_(p##5, 5, dst, C4, P); \
    PFT_PUT_(p##6, 6, dst, C4, P);                                                        \
  } while (0)
__device__ __forceinline__ void store_tile_nhwc(float* __restrict__ dst, const float* __restrict__ src, int C4, int P) {
  const int ps = threadIdx.x & 7, c4 = threadIdx.x >> 3, c = 4 * c4;
  if (c4 >= C4) return;
#pragma unroll
  for (int blk = 0; blk < 7; ++blk) {
    const int p = blk * 8 + ps;
    if (p < P)
      reinterpret_cast<float4*>(dst)[p * C4 + c4] =
          make_float4(src[c * P + p], src[(c + 1) * P + p], src[(c + 2) * P + p], src[(c + 3) * P + p]);
  }
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* __restrict__ src, int count) {
  for (int e = threadIdx.x * 4; e < count; e += NTHREADS * 4)
    *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(src + e);
}

// rn[j] = 1 / max(||buf[:,j]||, 1e-12) for j < P  (ends with a barrier)
__device__ __forceinline__ void col_rnorms(const float* buf, float* rn, float* part, int C, int P) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rows = C / NWAVES;
  float a0 = 0.f, a1 = 0.f;
  if (lane < P) {
    const float* p = buf + wave * rows * P + lane;
    for (int c = 0; c < rows; c += 2) {
      const float t0 = p[c * P], t1 = p[(c + 1) * P];
      a0 = __builtin_fmaf(t0, t0, a0);
      a1 = __builtin_fmaf(t1, t1, a1);
    }
  }
  part[wave * 64 + lane] = a0 + a1;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += part[w * 64 + threadIdx.x];
    rn[threadIdx.x] = 1.f / fmaxf(sqrtf(t), 1e-12f);
  }
  __syncthreads();
}

// Column 48 of a C x P product on the vector ALU: e(c) = sum_{k < 49} A[c][k] w[k] for the wave's 16 MT rows.  Lane l
// takes row (l & 31) of the wave's row list and half of k (l >> 5).  w[k] = scale[k] * B[k][48] is written by lane k
// into the wave's own 64-float LDS slice `ws` first (same wave writes and reads: no barrier), so one k costs two LDS
// reads (both pair up with their neighbours) and one FMA.  Returns e in lanes 0..31 (duplicated in 32..63).
template <int MT>
__device__ __forceinline__ float edge_column(const float* __restrict__ tile, int P, float wl, float* __restrict__ ws) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rl = lane & 31, hi = lane >> 5;
  const int row = 16 * (wave + NWAVES * min(rl >> 4, MT - 1)) + (rl & 15);
  ws[lane] = wl;                            // lanes >= 49 pass 0
  const float* a = tile + row * P + 25 * hi;
  const float* w = ws + 25 * hi;
  float e0 = 0.f, e1 = 0.f;
#pragma unroll
  for (int k = 0; k < 24; k += 2) {
    e0 = __builtin_fmaf(a[k], w[k], e0);
    e1 = __builtin_fmaf(a[k + 1], w[k + 1], e1);
  }
  e0 = __builtin_fmaf(hi ? 0.f : a[24], w[24], e0);     // k = 24: lower half only (the upper half ends at 48 = 25 + 23)
  float e = e0 + e1;
  e += __shfl_xor(e, 32, 64);
  if (MT == 1 && rl >= 16) e = 0.f;         // one row tile per wave: lanes 16..31 duplicate it
  return e;
}
// The same for G = k_hat M^T: w[k] = M[48][k] / ||k_k|| is zero except for the few positive pairs of row 48, so the
// loop runs over the set bits of the row (wave-uniform) - two LDS reads and one FMA per positive.
template <int MT>
__device__ __forceinline__ float edge_column_bits(const float* __restrict__ tile, int P, unsigned long long bits,
                                                  const float* __restrict__ rk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rl = lane & 31;
  const int row = 16 * (wave + NWAVES * min(rl >> 4, MT - 1)) + (rl & 15);
  const float* a = tile + row * P;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bits), hi = __builtin_amdgcn_readfirstlane((unsigned)(bits >> 32));
  float e0 = 0.f, e1 = 0.f;
  while (lo | hi) {
    int k;
    if (lo) { k = __builtin_ctz(lo); lo &= lo - 1; } else { k = 32 + __builtin_ctz(hi); hi &= hi - 1; }
    e0 = __builtin_fmaf(a[k], rk[k], e0);
    if (lo | hi) {
      if (lo) { k = __builtin_ctz(lo); lo &= lo - 1; } else { k = 32 + __builtin_ctz(hi); hi &= hi - 1; }
      e1 = __builtin_fmaf(a[k], rk[k], e1);
    }
  }
  float e = e0 + e1;
  if (MT == 1 && rl >= 16) e = 0.f;
  return e;
}

// P x P contraction over K = C for up to two output tiles per wave (tiles `wave` and `wave + NWAVES` of the
// NT x NT grid): out[i][j] = sum_c L[c][i] * R[c][j].  Double-buffered: the operands of the next KU k-steps are
// requested before the MFMAs of the current KU steps issue.
template <int NT>
__device__ __forceinline__ void gemm_pp(f32x4& acc0, f32x4& acc1, const float* __restrict__ L,
                                        const float* __restrict__ R, int C, int P) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15,
            lq = lane >> 4;
  const int t0 = wave, t1 = wave + NWAVES;
  const bool has0 = (NT * NT >= NWAVES) || t0 < NT * NT;
  const bool has1 = (NT * NT >= 2 * NWAVES) || t1 < NT * NT;    // compile-time true for the 4 x 4 tile grid
  if (!has0) return;
  const int i0 = min(16 * (t0 / NT) + lr, P - 1), j0 = min(16 * (t0 % NT) + lr, P - 1);
  const int i1 = has1 ? min(16 * (t1 / NT) + lr, P - 1) : i0, j1 = has1 ? min(16 * (t1 % NT) + lr, P - 1) : j0;
  // K order: the four k of one MFMA step are 16 rows apart (row = 64 (ks / 16) + ks % 16 + 16 lq) instead of adjacent:
  // with P = 49 the four lane groups then read bank windows 16 apart (49 * 16 = 16 mod 64) - no conflicts - where
  // adjacent rows overlap by one bank each (2-way conflicts on every operand read of the two P x P contractions).
  const float* la = L + 16 * lq * P + i0;
  const float* rb = R + 16 * lq * P + j0;
  const float* lc = L + 16 * lq * P + i1;
  const float* rd = R + 16 * lq * P + j1;
  const int step = P;
  const int nblk = C / 4 / KU;
  static_assert(KU == 8, "two blocks of KU steps walk one group of 64 rows");
  float a[KU], b[KU], c[KU], d[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    a[u] = la[u * step]; b[u] = rb[u * step];
    if (has1) { c[u] = lc[u * step]; d[u] = rd[u * step]; }
  }
  for (int blk = 0; blk < nblk; ++blk) {
    float an[KU], bn[KU], cn[KU], dn[KU];
    const int adv = blk + 1 < nblk ? ((blk & 1) ? 56 * step : 8 * step) : 0;     // the last block re-reads itself (never used)
    la += adv; rb += adv; lc += adv; rd += adv;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      an[u] = la[u * step]; bn[u] = rb[u * step];
      if (has1) { cn[u] = lc[u * step]; dn[u] = rd[u * step]; }
    }
    __builtin_amdgcn_sched_barrier(0);                 // keep the requests above ahead of the MFMAs below
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      acc0 = mfma16(a[u], b[u], acc0);
      if (has1) acc1 = mfma16(c[u], d[u], acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      a[u] = an[u]; b[u] = bn[u];
      if (has1) { c[u] = cn[u]; d[u] = dn[u]; }
    }
  }
}

// One 16 x 16 tile (ti, tj) of L^T R over K = C, two interleaved accumulators (even / odd k-steps) so that
// consecutive MFMAs never wait on each other's result; returns their sum.  Same double buffering as gemm_pp.
__device__ __forceinline__ f32x4 gemm_pp_tile(const float* __restrict__ L, const float* __restrict__ R, int C, int P,
                                              int ti, int tj) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const float* la = L + 16 * lq * P + min(16 * ti + lr, P - 1);       // K order as in gemm_pp
  const float* rb = R + 16 * lq * P + min(16 * tj + lr, P - 1);
  const int step = P;
  const int nblk = C / 4 / KU;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float a[KU], b[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) { a[u] = la[u * step]; b[u] = rb[u * step]; }
  for (int blk = 0; blk < nblk; ++blk) {
    float an[KU], bn[KU];
    const int adv = blk + 1 < nblk ? ((blk & 1) ? 56 * step : 8 * step) : 0;     // the last block re-reads itself (never used)
    la += adv; rb += adv;
#pragma unroll
    for (int u = 0; u < KU; ++u) { an[u] = la[u * step]; bn[u] = rb[u * step]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KU; u += 2) {
      acc0 = mfma16(a[u], b[u], acc0);
      acc1 = mfma16(a[u + 1], b[u + 1], acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KU; ++u) { a[u] = an[u]; b[u] = bn[u]; }
  }
  f32x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = acc0[r] + acc1[r];
  return acc;
}

// PC: compile-time P (49 for the 7x7 grid of 224^2 crops) so that every tile index folds into immediates;
// 0 = run-time P.
template <int MT, int NT, int PC>
__global__ void __launch_bounds__(NTHREADS)
ppm_fused_kernel(pixpro_problem pb, Dirs D, float* __restrict__ loss_n, float* __restrict__ pos_num,
                 float* __restrict__ pos_mean) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int C = 16 * NWAVES * MT, P = PC ? PC : pb.H * pb.W, N = pb.N;
  const int LR = lr_rows(P);
  const Smem sm = carve(smem_raw, C, P);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15,
            lq = lane >> 4;
  const int b = blockIdx.x, dir = b / N, n = b % N;
  const pixpro_dir& dd = D.d[dir];
  const int tile = C * P;
  const float* gx = dd.x + (size_t)n * tile;
  const float* gv = dd.v + (size_t)n * tile;
  const float* gk = dd.k + (size_t)n * tile;
  const int KS = LR / 4;          // k-steps over the pixel dimension
  const bool want_grad = dd.dx != nullptr;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // row of the wave's m-th C-tile owned by this lane as A operand / as accumulator register r
  auto arow = [&](int m) { return 16 * (wave + NWAVES * m) + lr; };
  auto crow = [&](int m, int r) { return 16 * (wave + NWAVES * m) + 4 * lq + r; };

  STAMP(0);
  // ---- waves 1..7: x -> B0 and v -> B1 in flight together; wave 0: bin centres, flow taps, threshold ----
  PF_DECL(pfa);
  PF_DECL(pfb);
  const bool nhwc = pb.layout == 1;
  const int C4 = C / 4;
#define LOAD_TILE(p, src)  do { if (wave != 0) { if (nhwc) PFT_ISSUE(p, src, C4, P); else PF_ISSUE(p, src, tile); } } while (0)
#define PUT_TILE(p, dst)   do { if (wave != 0) { if (nhwc) PFT_COMMIT(p, dst, C4, P); else PF_COMMIT(p, dst, tile); } } while (0)
#define OUT_TILE(gdst, src) do { if (nhwc) store_tile_nhwc(gdst, src, C4, P); else store_tile(gdst, src, tile); } while (0)
  LOAD_TILE(pfa, gx);               // x alone first: with v queued behind it as well, wave 0's nine tap loads below would
                                    // sit behind 100 KB in the CU's memory queue (measured: 9 k cycles to issue them)
  if (tid < 4) { sm.B0[tile + tid] = 0.f; sm.B1[tile + tid] = 0.f; }
  // lazy flow mode with a chain (nframe > 2): the nine composed-flow pixels each point needs (common.h) are spread over
  // the work-group, two dependent stages through LDS (the AA matrix is not live yet), so that a point costs two chain
  // latencies instead of nine
  const bool lazy_spread = dd.lo_own != nullptr && pb.lo_frames > 1;
  if (lazy_spread) {
    int Ho, Wo;
    frame_size(pb, dd, Ho, Wo);
    const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
    const Crop cq = crop_of(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
    const FlowGeom fg = flow_geom(8 * pb.lo_h, 8 * pb.lo_w, Ho, Wo);
    const LazySrc ls = lazy_src(pb, dd, n);
    float* LA = sm.AA;
    float* LB = sm.AA + 10 * P;
    for (int e = tid; e < 5 * P; e += NTHREADS) {
      const int pt = e / 5, k = e % 5;
      const LazyPoint lp = lazy_point(fg, centre_x(cq, pt % pb.W, wo1), centre_y(cq, pt / pb.W, ho1));
      float vx, vy;
      lazy_stage_a(ls, fg, lp, k, vx, vy);
      LA[2 * e] = vx; LA[2 * e + 1] = vy;
    }
    __syncthreads();
    for (int e = tid; e < 4 * P; e += NTHREADS) {
      const int pt = e / 4, k = e % 4;
      const LazyPoint lp = lazy_point(fg, centre_x(cq, pt % pb.W, wo1), centre_y(cq, pt / pb.W, ho1));
      float vx, vy;
      lazy_stage_b(ls, fg, lp, LA[10 * pt + 8], LA[10 * pt + 9], k, vx, vy);
      LB[2 * e] = vx; LB[2 * e + 1] = vy;
    }
    __syncthreads();
  }
  // Wave 0 has no tile loads of its own queued in front of these: the crop records come through the scalar cache, the
  // dense-flow taps are ONE vector round trip (displace_fetch) whose results are only consumed after the S contraction,
  // and the threshold search runs while they are in flight.  (Before: 12-16 k cycles in front of the first barrier -
  // five dependent round trips queued behind the wave's own tile loads.)
  float ptx = 0.f, pty = 0.f, pkx = 0.f, pky = 0.f;
  bool ptv = true, dense = false;
  PointTaps taps;      // written and read by wave 0 in dense mode only; NOT zero-initialised: merging a zero with the loaded
  FlowGeom fgd;        // values at the end of the branch would make the compiler wait for the loads right there
  if (wave == 0) {
    int Ho, Wo;
    frame_size_uniform(pb, dd, Ho, Wo);
    const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
    const Crop cq = crop_of_uniform(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
    const Crop ck = crop_of_uniform(dd.coord_k + (size_t)n * 10, pb.H, pb.W);
    const int pt = min(lane, P - 1), row = pt / pb.W, col = pt % pb.W;
    ptx = centre_x(cq, col, wo1); pty = centre_y(cq, row, ho1);
    pkx = centre_x(ck, col, wo1); pky = centre_y(ck, row, ho1);
    dense = dd.lo_own == nullptr && dd.flow != nullptr;
    if (dense) {
      fgd = flow_geom(pb.Hf, pb.Wf, Ho, Wo);
      taps = displace_fetch(dd.flow + (size_t)n * 2 * pb.Hf * pb.Wf,
                            dd.mask ? dd.mask + (size_t)n * pb.Hf * pb.Wf : nullptr, fgd, ptx, pty);
    }
    STAMP(17);
    // threshold of the positive-pair test (needs only the two crop records)
    const float T = dist_threshold(fmaxf(bin_diag(cq, wo1, ho1), bin_diag(ck, wo1, ho1)), pb.pos_ratio, lane);
    if (lane == 0) sm.misc[2] = T;
    if (dd.lo_own) {            // lazy flow mode: evaluate flow + validity at this point from the low-res maps
      const FlowGeom fg = flow_geom(8 * pb.lo_h, 8 * pb.lo_w, Ho, Wo);
      const LazySrc ls = lazy_src(pb, dd, n);
      float ox, oy;
      if (lazy_spread) lazy_finish(ls, fg, lazy_point(fg, ptx, pty), ptx, pty, sm.AA + 10 * pt, sm.AA + 10 * P + 8 * pt, ox, oy, ptv);
      else displace_point_lazy(ls, fg, ptx, pty, ox, oy, ptv);
      ptx = ox; pty = oy;
    }
  }
  STAMP(16);
  PUT_TILE(pfa, sm.B0);
  __syncthreads();
  STAMP(1);
  // One tile request at a time: the CU's memory queue holds about one tile's worth of wave-loads; with more, a wave
  // stalls at the *issue* of its next load until the queue drains (measured: 14 loads per wave took 5.9 k cycles to
  // issue), and a barrier behind such a stall costs every wave that time.
  LOAD_TILE(pfb, gv);               // v streams in under the S contraction

  // ---- S_raw = x^T x ; rx from its diagonal ; S, A, D ----
  // 7 x 7 grid (PC == 49): S is symmetric and 49 = 48 + 1, so only the six upper tiles of the 3 x 3 tile grid go to the
  // matrix cores - one per wave 0..5, 64 MFMAs instead of the 128 of two tiles of the padded 4 x 4 grid - and are
  // mirrored by the epilogue; waves 6 and 7, which share their SIMDs with one MFMA wave each, take row 48
  // (49 dot products over half of C each) on the vector ALU.
  constexpr bool SYM = PC == 49;
  {
    f32x4 acc0 = zero4, acc1 = zero4;
    const int sti = (wave >= 3) + (wave >= 5), stj = wave - (wave >= 3 ? 2 : 0) - (wave >= 5 ? 1 : 0);   // SYM: tile of waves 0..5
    if constexpr (SYM) {
      if (wave < 6) {
        acc0 = gemm_pp_tile(sm.B0, sm.B0, C, P, sti, stj);
        if (sti == stj) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * lq + r == lr) sm.nx2[16 * sti + lr] = acc0[r];
        }
      } else {
        const int half = C / 2, jj = min(lane, P - 1);
        const float* xe = sm.B0 + (wave - 6) * half * P;
        float e0 = 0.f, e1 = 0.f;
#pragma unroll 8
        for (int c = 0; c < half; c += 2) {
          e0 = __builtin_fmaf(xe[c * P + (P - 1)], xe[c * P + jj], e0);
          e1 = __builtin_fmaf(xe[(c + 1) * P + (P - 1)], xe[(c + 1) * P + jj], e1);
        }
        sm.part1[(wave - 6) * 64 + lane] = e0 + e1;       // part1 is next used by the Y-norm reduction
      }
    } else {
      gemm_pp<NT>(acc0, acc1, sm.B0, sm.B0, C, P);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tt = wave + NWAVES * t;
        if (tt < NT * NT && tt / NT == tt % NT) {
          const f32x4& acc = t ? acc1 : acc0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * lq + r == lr) sm.nx2[16 * (tt / NT) + lr] = acc[r];
        }
      }
    }
    if (wave == 0) {                 // the sample points: taps requested before the contraction
      if (dense) {
        float ox, oy;
        displace_finish(taps, fgd, ptx, pty, ox, oy, ptv);
        ptx = ox; pty = oy;
      }
      const bool live = lane < P;
      sm.qx[lane] = live ? ptx : 0.f; sm.qy[lane] = live ? pty : 0.f;
      sm.kx[lane] = live ? pkx : 0.f; sm.ky[lane] = live ? pky : 0.f;
      sm.valid[lane] = (live && ptv) ? 1 : 0;
    }
    PUT_TILE(pfb, sm.B1);            // v has had the S contraction to arrive
    LOAD_TILE(pfa, gk);              // k streams in under the mask / norm / S epilogue phases
    __syncthreads();
    STAMP(2);
    if (tid < 64) {
      float n2 = sm.nx2[tid < 16 * NT ? tid : 0];
      if (SYM && tid >= 48) n2 = sm.part1[48] + sm.part1[64 + 48];
      sm.rx[tid] = 1.f / fmaxf(sqrtf(n2), 1e-12f);
    }
    // ---- positive-pair bit mask: row i of M = ballot over j (PixPro.py:217-223); row i's point comes out of lane i ----
    {
      const float T = sm.misc[2];     // sqrtf(d2) / diag < pos_ratio  <=>  d2 < T  (dist_threshold)
      const float kxj = sm.kx[lane], kyj = sm.ky[lane], qxl = sm.qx[lane], qyl = sm.qy[lane];
      const unsigned long long vrows = __ballot(sm.valid[lane] != 0);
      int cnt_w = 0;
#pragma unroll
      for (int it = 0; it < 64 / NWAVES; ++it) {
        const int i = wave + NWAVES * it;
        const float qxi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qxl), i));
        const float qyi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qyl), i));
        const float dx = qxi - kxj, dy = qyi - kyj;
        const bool pred = i < P && lane < P && (dx * dx + dy * dy < T) && ((vrows >> i) & 1ull);
        const unsigned long long bits = __ballot(pred);
        if (lane == 0) sm.mbits[i] = bits;
        cnt_w += __popcll(bits);
      }
      if (lane == 0) sm.misc[8 + wave] = (float)cnt_w;
    }
    STAMP(3);
    col_rnorms(sm.B1, sm.rv, sm.part0, C, P);           // (barriers inside also publish rx, the mask rows and counts)
    if (tid == 0) {
      float c = 0.f;
      for (int w = 0; w < NWAVES; ++w) c += sm.misc[8 + w];
      sm.misc[1] = c;                                    // read in the loss phase, several barriers from here
    }
    if constexpr (SYM) {
      if (wave < 6) {
        const int j = 16 * stj + lr;
        const float rxj = sm.rx[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * sti + 4 * lq + r;
          float a, d;
          pow_map(acc0[r] * sm.rx[i] * rxj, pb.p, pb.clamp, a, d);
          sm.AA[i * LS + j] = a;
          sm.Dm[i * LS + j] = d;
          if (sti != stj) { sm.AA[j * LS + i] = a; sm.Dm[j * LS + i] = d; }
        }
      } else {
        // row / column 48, then the zero padding the contractions read: rows 49..51 and columns 49..63
        const int t2 = tid - 6 * WAVE;                    // 0..127
        if (t2 < P) {
          float a, d;
          pow_map((sm.part1[t2] + sm.part1[64 + t2]) * sm.rx[P - 1] * sm.rx[t2], pb.p, pb.clamp, a, d);
          sm.AA[(P - 1) * LS + t2] = a; sm.Dm[(P - 1) * LS + t2] = d;
          sm.AA[t2 * LS + (P - 1)] = a; sm.Dm[t2 * LS + (P - 1)] = d;
        }
        for (int e = t2; e < (LR - P) * 64; e += 2 * WAVE) {
          const int i = P + (e >> 6), jz = e & 63;
          sm.AA[i * LS + jz] = 0.f; sm.Dm[i * LS + jz] = 0.f;
        }
        for (int e = t2; e < P * (64 - P); e += 2 * WAVE) {
          const int i = e / (64 - P), jz = P + e % (64 - P);
          sm.AA[i * LS + jz] = 0.f; sm.Dm[i * LS + jz] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tt = wave + NWAVES * t;
        if (tt < NT * NT) {
          const f32x4& acc = t ? acc1 : acc0;
          const int mi = tt / NT, nj = tt % NT, j = 16 * nj + lr;
          const float rxj = sm.rx[j];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * mi + 4 * lq + r;
            if (i < LR) {
              float a = 0.f, d = 0.f;
              if (i < P && j < P) pow_map(acc[r] * sm.rx[i] * rxj, pb.p, pb.clamp, a, d);
              sm.AA[i * LS + j] = a;
              sm.Dm[i * LS + j] = d;
            }
          }
        }
      }
    }
  }
  __syncthreads();                 // every wave is done with x in B0
  PUT_TILE(pfa, sm.B0);            // k -> B0
  STAMP(4);

  // ---- Y = v_hat A ----
  // 7 x 7 grid: the four C x P products run on NTM = 3 column tiles (48 columns); column 48 is a vector-ALU dot product
  // per row (edge_column), held one value per lane (lanes 0..31 = the wave's rows) next to the accumulators.
  constexpr bool EDGE = PC == 49;
  constexpr int NTM = EDGE ? 3 : NT;
  const bool elive = lane < 16 * MT;                      // lanes that own an edge value
  const int erow = 16 * (wave + NWAVES * min((lane & 31) >> 4, MT - 1)) + (lane & 15);   // ... and its row
  f32x4 Y[MT][NTM];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) Y[m][nn] = zero4;
  // k index j >= P reads the next row / the zeroed pad: finite, times a zero B row
  gemm_cp<MT, NTM>(Y, KS,
                   [&](int m, int ks) { return sm.B1[arow(m) * P + 4 * ks + lq]; },
                   [&](int ks) { return sm.rv[4 * ks + lq]; },
                   [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; });
  float Ye = 0.f;
  if constexpr (EDGE) Ye = edge_column<MT>(sm.B1, P, lane < P ? sm.rv[lane] * sm.AA[lane * LS + (P - 1)] : 0.f, sm.part0 + wave * 64);
  STAMP(5);
  // Column norms of Y and of k with two barriers.  Y stays raw in its registers (its 1 / ||.|| enters the loss dot product
  // and dY as per-column factors); every wave keeps its own copy of 1 / ||k_j|| in its slice of part1, so that no third
  // barrier is needed to publish it.
  float ry[NTM];     // 1 / ||Y[:, i]|| of this lane's columns
  float rye = 0.f;   // ... and of column 48
  {
    if constexpr (EDGE) {
      const float qe = wave_sum(elive ? Ye * Ye : 0.f);
      if (lane == 0) sm.part1[wave * 64 + 48] = qe;
    }
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float q = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) q = __builtin_fmaf(Y[m][nn][r], Y[m][nn][r], q);
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 16) sm.part1[wave * 64 + 16 * nn + lane] = q;
    }
    __syncthreads();                        // also publishes k in B0
    if (pb.normalize_k) {
      const int rows = C / NWAVES;
      float a0 = 0.f, a1 = 0.f;
      if (lane < P) {
        const float* kp = sm.B0 + wave * rows * P + lane;
        for (int c = 0; c < rows; c += 2) {
          const float t0 = kp[c * P], t1 = kp[(c + 1) * P];
          a0 = __builtin_fmaf(t0, t0, a0);
          a1 = __builtin_fmaf(t1, t1, a1);
        }
      }
      sm.part0[wave * 64 + lane] = a0 + a1;
    }
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part1[w * 64 + 16 * nn + lr];
      ry[nn] = 1.f / fmaxf(sqrtf(t), 1e-12f);
    }
    if constexpr (EDGE) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part1[w * 64 + 48];
      rye = 1.f / fmaxf(sqrtf(t), 1e-12f);
    }
    __syncthreads();
    float rkj = 1.f;
    if (pb.normalize_k) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part0[w * 64 + lane];
      rkj = 1.f / fmaxf(sqrtf(t), 1e-12f);
    }
    sm.part1[wave * 64 + lane] = rkj;       // wave-private from here to the end of the G contraction
  }
  if (dd.pred) {
    float* gp = dd.pred + (size_t)n * tile;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * nn + lr;
          if (i < P) gp[nhwc ? (size_t)i * C + crow(m, r) : (size_t)crow(m, r) * P + i] = Y[m][nn][r] * ry[nn];
        }
    if (EDGE && elive) gp[nhwc ? (size_t)(P - 1) * C + erow : (size_t)erow * P + (P - 1)] = Ye * rye;
  }
  STAMP(6);

  // ---- G = k_hat M^T ----
  f32x4 G[MT][NTM];
  float Ge = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) G[m][nn] = zero4;
  {
    // row bits pre-shifted by the lane's k offset: bit 4*ks of the low / high word is M[i][4*ks + lq]
    unsigned lo[NTM], hi[NTM];
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const unsigned long long rb = sm.mbits[16 * nn + lr] >> lq;
      lo[nn] = (unsigned)rb; hi[nn] = (unsigned)(rb >> 32);
    }
    gemm_cp<MT, NTM>(G, KS,
                     [&](int m, int ks) { return sm.B0[arow(m) * P + 4 * ks + lq]; },
                     [&](int ks) { return sm.part1[wave * 64 + 4 * ks + lq]; },
                     [&](int nn, int ks) {
                       const unsigned w = ks < 8 ? lo[nn] : hi[nn];
                       return (float)((w >> ((4 * ks) & 31)) & 1u);
                     });
    if constexpr (EDGE)                     // row 48 of M against the wave's own copy of 1 / ||k||
      Ge = edge_column_bits<MT>(sm.B0, P, sm.mbits[P - 1] & ((1ull << P) - 1), sm.part1 + wave * 64);
  }
  STAMP(7);
  float tc[NTM];     // <Y_hat[:, i], G[:, i]> of this lane's columns (Y raw in registers: ry applied after the reduction)
  float tce = 0.f;
  {
    float v[NTM];
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s = __builtin_fmaf(Y[m][nn][r], G[m][nn][r], s);
      v[nn] = s;
    }
    // a wave overwrites only its own slice (its 1 / ||k|| copy, no longer needed); barrier: every wave is done reading k
    if constexpr (EDGE) {
      float ve = elive ? Ye * Ge : 0.f;
      col_reduce_edge<NTM>(v, ve, sm.part1);
      tce = ve * rye;
    } else {
      col_reduce_reg<NTM>(v, sm.part1);
    }
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) tc[nn] = v[nn] * ry[nn];
  }
  const float cnt = sm.misc[1];
  const float den = cnt + 1e-6f;
  if (wave == 0) {
    float t = 0.f;
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) t += (lane < 16 && 16 * nn + lr < P) ? tc[nn] : 0.f;
    if (EDGE && lane == 0) t += tce;
    t = wave_sum(t);
    if (lane == 0) {
      loss_n[b] = t / den;
      pos_num[b] = cnt;
      pos_mean[b] = cnt / (float)P / (float)P;
    }
  }
  STAMP(8);
  if (!want_grad) return;

  // ---- dY -> B0 ; second read of x starts ----
  LOAD_TILE(pfa, gx);
  {
    const float gs = pb.grad_scale * -2.f / (float)N / den;
    if (EDGE && elive) sm.B0[erow * P + (P - 1)] = __builtin_fmaf(-Ye, rye * rye * (gs * tce), Ge * (gs * rye));
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const int i = 16 * nn + lr;
      if (i < P) {
        // (gs G - Y_hat ti) ry with Y_hat = Y ry, ti = gs <Y_hat, G>:  G (gs ry) - Y (ry ry ti)
        const float ryv = ry[nn], ga = gs * ryv, yb = ryv * ryv * (gs * tc[nn]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) sm.B0[crow(m, r) * P + i] = __builtin_fmaf(-Y[m][nn][r], yb, G[m][nn][r] * ga);
      }
    }
  }
  __syncthreads();
  STAMP(9);

  // ---- dA = dY^T v_hat ; dS = dA o D ----
  {
    f32x4 acc0 = zero4, acc1 = zero4;
    gemm_pp<NT>(acc0, acc1, sm.B0, sm.B1, C, P);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tt = wave + NWAVES * t;
      if (tt < NT * NT) {
        const f32x4& acc = t ? acc1 : acc0;
        const int mi = tt / NT, nj = tt % NT, j = 16 * nj + lr;
        const float rvj = sm.rv[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * mi + 4 * lq + r;
          if (i < LR) sm.DS[i * LS + j] = acc[r] * rvj * sm.Dm[i * LS + j];
        }
      }
    }
  }
  __syncthreads();
  STAMP(10);
  // dSs = dS + dS^T into Dm (rows/cols >= P are zero because D is zero there); published by the barriers
  // of the dV epilogue, first read by the dX contraction
  for (int e = tid; e < LR * 64; e += NTHREADS) {
    const int i = e >> 6, j = e & 63;
    sm.Dm[i * LS + j] = (j < LR) ? sm.DS[i * LS + j] + sm.DS[j * LS + i] : 0.f;
  }
  STAMP(11);

  // ---- dV_hat = dY A ; dv = (dV_hat - v_hat s) / n_v ----
  {
    f32x4 T[MT][NTM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) T[m][nn] = zero4;
    gemm_cp<MT, NTM>(T, KS,
                     [&](int m, int ks) { return sm.B0[arow(m) * P + 4 * ks + lq]; },
                     [&](int) { return 1.f; },
                     [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; });
    float Te = 0.f, ve = 0.f, ves = 0.f;     // column 48: product, raw v, partial of the projection dot product
    if constexpr (EDGE) {
      Te = edge_column<MT>(sm.B0, P, lane < P ? sm.AA[lane * LS + (P - 1)] : 0.f, sm.part1 + wave * 64);
      ve = sm.B1[erow * P + (P - 1)];
      ves = elive ? ve * Te * sm.rv[P - 1] : 0.f;
    }
    float vh[MT][NTM][4];
    float v[NTM];
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const int j = 16 * nn + lr;
      float s = 0.f;
      const float rvj = sm.rv[min(j, P - 1)];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vh[m][nn][r] = (j < P) ? sm.B1[crow(m, r) * P + j] : 0.f;        // raw v: rv folded into the factors below
          s = __builtin_fmaf(vh[m][nn][r], T[m][nn][r], s);
        }
      v[nn] = s * rvj;
    }
    // barrier inside: every wave is done reading dY from B0
    if constexpr (EDGE) col_reduce_edge<NTM>(v, ves, sm.part0); else col_reduce_reg<NTM>(v, sm.part0);
    PUT_TILE(pfa, sm.B0);                    // x -> B0 (second read)
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const int j = 16 * nn + lr;
      if (j < P) {
        const float rvj = sm.rv[j], cj = rvj * rvj * v[nn];      // (T - v_hat s) rv = T rv - v (rv rv s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) sm.B1[crow(m, r) * P + j] = __builtin_fmaf(-vh[m][nn][r], cj, T[m][nn][r] * rvj);
      }
    }
    if (EDGE && elive) {
      const float rve = sm.rv[P - 1];
      sm.B1[erow * P + (P - 1)] = __builtin_fmaf(-ve, rve * rve * ves, Te * rve);
    }
  }
  __syncthreads();
  STAMP(12);
  OUT_TILE(dd.dv + (size_t)n * tile, sm.B1);
  STAMP(13);

  // ---- dX_hat = x_hat dSs ; dx = (dX_hat - x_hat s) / n_x ----
  {
    f32x4 T[MT][NTM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) T[m][nn] = zero4;
    gemm_cp<MT, NTM>(T, KS,
                     [&](int m, int ks) { return sm.B0[arow(m) * P + 4 * ks + lq]; },
                     [&](int ks) { return sm.rx[4 * ks + lq]; },
                     [&](int nn, int ks) { return sm.Dm[(4 * ks + lq) * LS + 16 * nn + lr]; });
    float Te = 0.f, xe = 0.f, xes = 0.f;
    if constexpr (EDGE) {
      Te = edge_column<MT>(sm.B0, P, lane < P ? sm.rx[lane] * sm.Dm[lane * LS + (P - 1)] : 0.f, sm.part0 + wave * 64);
      xe = sm.B0[erow * P + (P - 1)];
      xes = elive ? xe * Te * sm.rx[P - 1] : 0.f;
    }
    float xh[MT][NTM][4];
    float v[NTM];
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const int j = 16 * nn + lr;
      float s = 0.f;
      const float rxj = sm.rx[min(j, P - 1)];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xh[m][nn][r] = (j < P) ? sm.B0[crow(m, r) * P + j] : 0.f;        // raw x
          s = __builtin_fmaf(xh[m][nn][r], T[m][nn][r], s);
        }
      v[nn] = s * rxj;
    }
    // barrier inside: every wave is done reading x from B0
    if constexpr (EDGE) col_reduce_edge<NTM>(v, xes, sm.part1); else col_reduce_reg<NTM>(v, sm.part1);
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const int j = 16 * nn + lr;
      if (j < P) {
        const float rxj = sm.rx[j], cj = rxj * rxj * v[nn];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) sm.B0[crow(m, r) * P + j] = __builtin_fmaf(-xh[m][nn][r], cj, T[m][nn][r] * rxj);
      }
    }
    if (EDGE && elive) {
      const float rxe = sm.rx[P - 1];
      sm.B0[erow * P + (P - 1)] = __builtin_fmaf(-xe, rxe * rxe * xes, Te * rxe);
    }
  }
  __syncthreads();
  STAMP(14);
  OUT_TILE(dd.dx + (size_t)n * tile, sm.B0);
  STAMP(15);
}

template <int MT, int NT, int PC = 0>
int launch(const pixpro_problem& pb, const Dirs& D, float* loss_n, float* pos_num, float* pos_mean, hipStream_t st) {
  const int P = pb.H * pb.W;
  const size_t bytes = smem_floats(pb.C, P) * sizeof(float);
  auto kern = ppm_fused_kernel<MT, NT, PC>;
  static thread_local size_t configured = 0;
  if (bytes > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
      return PIXPRO_ERR_LAUNCH;
    }
    configured = bytes;
  }
  hipLaunchKernelGGL(kern, dim3(pb.N * pb.ndir), dim3(NTHREADS), bytes, st, pb, D, loss_n, pos_num, pos_mean);
  return check_launch("ppm_fused_kernel");
}

}  // namespace

bool fused_supported(const pixpro_problem& pb) {
  const int P = pb.H * pb.W;
  if (P > 64 || P < 1) return false;
  if (pb.C != 128 && pb.C != 256) return false;
  if ((size_t)pb.C * P > (size_t)PF * NLOAD * 4) return false;      // 7 loader waves x 7 float4: P <= 49 at C = 256
  if (pb.layout == 1 && (P / 4) * (pb.C / 64) + P % 4 > PF * (NWAVES - 1)) return false;   // NHWC wave-loads (nhwc_slot)
  return smem_floats(pb.C, P) * sizeof(float) <= 160 * 1024;
}

int run_fused(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                   hipStream_t st) {
  Dirs D;
  for (int d = 0; d < PIXPRO_MAX_DIRS; ++d) D.d[d] = dirs[d < pb.ndir ? d : 0];
  const int P = pb.H * pb.W;
  const int NT = (P + 15) / 16;
  if (pb.C == 32 * NWAVES) {   // two C-tiles per wave (C = 256)
    if (P == 49) return launch<2, 4, 49>(pb, D, loss_n, pos_num, pos_mean, st);
    switch (NT) {
      case 1: return launch<2, 1>(pb, D, loss_n, pos_num, pos_mean, st);
      case 2: return launch<2, 2>(pb, D, loss_n, pos_num, pos_mean, st);
      case 3: return launch<2, 3>(pb, D, loss_n, pos_num, pos_mean, st);
      default: return launch<2, 4>(pb, D, loss_n, pos_num, pos_mean, st);
    }
  }
  if (P == 49) return launch<1, 4, 49>(pb, D, loss_n, pos_num, pos_mean, st);
  switch (NT) {
    case 1: return launch<1, 1>(pb, D, loss_n, pos_num, pos_mean, st);
    case 2: return launch<1, 2>(pb, D, loss_n, pos_num, pos_mean, st);
    case 3: return launch<1, 3>(pb, D, loss_n, pos_num, pos_mean, st);
    default: return launch<1, 4>(pb, D, loss_n, pos_num, pos_mean, st);
  }
}

}  // namespace pixpro
