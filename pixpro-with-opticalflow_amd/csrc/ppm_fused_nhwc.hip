// Single-launch PPM -> flow-matched pixel loss -> backward for channels_last (NHWC) tiles of the 7 x 7 grid at C = 256 -
// the layout and shape the pre-training step runs in (bench default).  Same math, same phases and the same bit-exact
// coordinate part as ppm_fused.hip (whose kernel keeps serving NCHW tiles and the other shapes); what differs is the LDS
// image and everything that follows from it:
//
//   * A sample's tile is [P][C] in HBM and STAYS [P][C] in LDS (row stride 272 floats = 16 mod 64 banks): it arrives by
//     LDS-DMA, one 1-KB pixel row per wave instruction (global_load_lds_dwordx4), with no staging registers, no commit
//     instructions and no transposition - all eight waves are free to compute while it travels.
//   * The 16-B chunk that holds channels 4q..4q+3 of pixel p sits at chunk position q ^ ((p >> 2) & 3) (applied on the
//     SOURCE address of the copy, so the copy itself stays linear).  With it
//       - the P x P products over K = C (S = x^T x, dA = dY^T v) read BOTH operands k-contiguous: one ds_read_b128 per
//         four MFMA steps (lane = pixel lr x k group lq), bank-conflict free;
//       - the C x P products over K = P read their A operand as ds_read_b32 (16 consecutive channels per k, the four k
//         of a step 16 banks apart), conflict free;
//       - every epilogue touches the tile in the accumulator layout with 16-B accesses (four consecutive channels of one
//         pixel) instead of four scalar ones 49 floats apart.
//   * The walk is persistent: grid = min(items, CUs); while a work-group finishes item b (dX product, dx store) the x
//     tile of item b + grid is already travelling into the other tile buffer, so only the first item of a work-group
//     pays the tile latency in front of its first product.
//   * dS lives in place of D, and dS + dS^T goes where A was once the dV product has read it (two P x P matrices
//     instead of three), which is what pays for the wider tile rows.
//
// Buffers over time (T0, T1 swap roles after every item):
//   T0: x (S) -> k (G) -> dY (dA, dV) -> x (dX, dx out)         T1: v (Y, dA, dv out) -> x of the next item
//
// Math: SURVEY.md Appendix A; reference contrast/models/PixPro.py:339-363,380,403 (PPM + the three F.normalize),
// :92-247 (regression_loss), :46-89 (add_optical_flow).
#include "ppm_fused_shared.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace pixpro {

#ifdef PIXPRO_STAMPS   // diagnostic build only (tools/ppm_phase_stamps.py --nhwc): wave 0 of the last work-group stamps s_memtime
extern __device__ unsigned long long g_stamps[64];            // defined in ppm_fused.hip; this kernel uses entries 32..63
#define STAMPN(i)                                                                                  \
  do {                                                                                             \
    if (stamp_on >= 0 && threadIdx.x == 0) g_stamps[stamp_on + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define STAMPN(i)
#endif

namespace {

using namespace fused;

constexpr int P = 49, C = 256, MT = 2, NTM = 3;
constexpr int TS = 272;                  // tile row stride in floats: 256 channels + 16 of padding; 272 = 16 (mod 64)
constexpr int TROWS = 52;                // 49 pixels + 3 rows of zeros (k = 49..51 of the C x P products)
constexpr int TILE = TROWS * TS;         // 14,144 floats = 56,576 B
constexpr int LR = 52;                   // rows of the P x P matrices
constexpr int KS = LR / 4;               // k-steps over the pixel dimension

// Lane coordinates of a thread.  Inside the persistent loop they are re-derived from an OPAQUE copy of threadIdx.x once per
// item (lane_ctx): whatever is computed from them - every LDS address of the body - then cannot be hoisted out of the loop,
// where it would stay live across the whole item (the inlined loop compiled to 256 VGPRs plus spills that way).
struct Ctx {
  int tid, wave, lane, lr, lq;
};
__device__ __forceinline__ Ctx lane_ctx() {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  Ctx c;
  c.tid = tid; c.wave = __builtin_amdgcn_readfirstlane(tid >> 6); c.lane = tid & 63; c.lr = c.lane & 15; c.lq = c.lane >> 4;
  return c;
}
// 1 / max(sqrt(t), 1e-12) = min(rsqrt(t), 1e12) on v_rsq_f32 (1 ulp) - two instructions instead of the ~20 of a correctly
// rounded sqrt and a correctly rounded division.  A lane evaluates about a dozen of these per item (it re-derives the norms
// it needs instead of waiting a barrier for somebody else's), which was a tenth of the kernel's vector-ALU instructions.
// The contractions re-associate anyway (tolerances of DESIGN.md 2); t = 0, denormal or NaN give 1e12 either way.
__device__ __forceinline__ float rnorm(float t) { return fminf(__builtin_amdgcn_rsqf(t), 1e12f); }
__device__ __forceinline__ int swz(int p) { return (p >> 2) & 3; }
// element (pixel p, channel c) of a tile
__device__ __forceinline__ int tix(int p, int c) { return p * TS + ((((c >> 2) ^ swz(p))) << 2) + (c & 3); }

struct Smem {
  float* T;      // [2][TILE]
  float* AA;     // [LR][LS]   A = pow(clamp(S))
  float* DM;     // [LR][LS]   dA/dS, then dS (in place); dS + dS^T goes to AA once the dV contraction has read A
  float* rx;     // [64] reciprocal column norms of x and v
  float* rv;
  float* nx2;    // [64] diag of the raw Gram matrix
  float* part0;  // [NWAVES][64] x3: scratch of the column reductions (rv / k norms / dV; Y norms / dX; loss)
  float* part1;
  float* part2;
  float* rkw;    // [NWAVES][64] every wave's own copy of 1 / ||k_j||
  float* part3;  // [NWAVES][64] scratch of the dV reduction (part0..rkw hold the dA partials meanwhile)
  float* qx;     // [64] each
  float* qy;
  float* kx;
  float* ky;
  float* misc;   // [32]: [1] cnt, [2] threshold, [8 + w] per-wave positive counts
  unsigned long long* mbits;  // [64]
  int* valid;    // [64]
};
constexpr size_t SMEM_FLOATS = 2 * TILE + 2 * LR * LS + 3 * 64 + 5 * NWAVES * 64 + 4 * 64 + 32 + 2 * 64 + 64;

__device__ __forceinline__ Smem carve(float* base) {
  Smem s;
  float* p = base;
  s.T = p; p += 2 * TILE;
  s.AA = p; p += LR * LS;
  s.DM = p; p += LR * LS;
  s.rx = p; p += 64; s.rv = p; p += 64; s.nx2 = p; p += 64;
  s.part0 = p; p += NWAVES * 64; s.part1 = p; p += NWAVES * 64; s.part2 = p; p += NWAVES * 64; s.rkw = p; p += NWAVES * 64; s.part3 = p; p += NWAVES * 64;
  s.qx = p; p += 64; s.qy = p; p += 64; s.kx = p; p += 64; s.ky = p; p += 64;
  s.misc = p; p += 32;
  s.mbits = (unsigned long long*)p; p += 2 * 64;
  s.valid = (int*)p;
  return s;
}

// HBM [P][C] -> LDS tile by LDS-DMA: wave w copies pixel rows w, w + 8, ...; lane l fetches logical chunk l ^ swz(p) into
// chunk position l.  Nothing orders the copy against LDS reads except s_waitcnt vmcnt + a barrier (tile_landed).
__device__ __forceinline__ void tile_dma(const Ctx& cx, const float* __restrict__ g, float* T) {
  const int wave = cx.wave, lane = cx.lane;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int p = wave + NWAVES * i;
    if (p < P) dma16(g + p * C + 4 * (lane ^ swz(p)), T + p * TS);
  }
}
// The same issued by waves 1..6 only (the S phase: waves 0 and 7 run latency chains on the vector ALU there and every copy
// a wave issues stalls it for ~200 cycles while the texture path takes the 1-KB row)
__device__ __forceinline__ void tile_dma_mid(const Ctx& cx, const float* __restrict__ g, float* T) {
  const int wave = cx.wave, lane = cx.lane;
  if (wave < 1 || wave > 6) return;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int p = (wave - 1) + 6 * i;
    if (p < P) dma16(g + p * C + 4 * (lane ^ swz(p)), T + p * TS);
  }
}
// LDS tile -> HBM [P][C], 16 B per lane, one pixel row per wave instruction
__device__ __forceinline__ void tile_store(const Ctx& cx, float* __restrict__ g, const float* T) {
  const int wave = cx.wave, lane = cx.lane;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int p = wave + NWAVES * i;
    if (p < P)
      *reinterpret_cast<float4*>(g + p * C + 4 * (lane ^ swz(p))) = *reinterpret_cast<const float4*>(T + p * TS + 4 * lane);
  }
}

// 1 / max(||T[:, j]||, 1e-12)-style column sums of squares: lane = pixel, wave w covers channels 32 w .. 32 w + 31
__device__ __forceinline__ float col_sumsq_part(const Ctx& cx, const float* T) {
  const int wave = cx.wave, lane = cx.lane;
  const int p = min(lane, P - 1), s = swz(p);
  const float* row = T + p * TS;
  float a0 = 0.f, a1 = 0.f;
  float4 t[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) t[q] = *reinterpret_cast<const float4*>(row + 4 * ((8 * wave + q) ^ s));
  __builtin_amdgcn_sched_barrier(0);          // all eight requested before the first is consumed
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a0 = __builtin_fmaf(t[q].x, t[q].x, a0); a1 = __builtin_fmaf(t[q].y, t[q].y, a1);
    a0 = __builtin_fmaf(t[q].z, t[q].z, a0); a1 = __builtin_fmaf(t[q].w, t[q].w, a1);
  }
  return a0 + a1;
}

// One 16 x 16 tile (ti, tj) of L^T R over K = C, both operands k-contiguous: lane (pixel lr, k group lq) reads the four
// channels 16 g + 4 lq .. + 3 of its pixel with ONE ds_read_b128 per operand and feeds four MFMA steps (the order of the
// contraction index is free as long as both operands agree).  Two interleaved accumulators; groups requested four ahead.
__device__ __forceinline__ f32x4 gemm_pp_tile128(const Ctx& cx, const float* __restrict__ L, const float* __restrict__ R, int ti,
                                              int tj) {
  const int lr = cx.lr, lq = cx.lq;
  const int pa = min(16 * ti + lr, P - 1), pb = min(16 * tj + lr, P - 1);      // rows beyond P: clamped duplicates, masked later
  const float* la = L + pa * TS + 4 * (lq ^ swz(pa));
  const float* rb = R + pb * TS + 4 * (lq ^ swz(pb));
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int GB = 4, NG = C / 16 / GB;          // 4 blocks of 4 groups of 4 k-steps
  float4 a[GB], b[GB];
#pragma unroll
  for (int u = 0; u < GB; ++u) {
    a[u] = *reinterpret_cast<const float4*>(la + 16 * u);
    b[u] = *reinterpret_cast<const float4*>(rb + 16 * u);
  }
#pragma unroll
  for (int blk = 0; blk < NG; ++blk) {
    float4 an[GB], bn[GB];
    const int nb = blk + 1 < NG ? blk + 1 : blk;       // the last block re-reads itself (never used)
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      an[u] = *reinterpret_cast<const float4*>(la + 16 * (GB * nb + u));
      bn[u] = *reinterpret_cast<const float4*>(rb + 16 * (GB * nb + u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      acc0 = mfma16(a[u].x, b[u].x, acc0);
      acc1 = mfma16(a[u].y, b[u].y, acc1);
      acc0 = mfma16(a[u].z, b[u].z, acc0);
      acc1 = mfma16(a[u].w, b[u].w, acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GB; ++u) { a[u] = an[u]; b[u] = bn[u]; }
  }
  f32x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = acc0[r] + acc1[r];
  return acc;
}
// ng groups of 16 channels (4 MFMA steps each) of one tile, starting at group g0: a K slice of gemm_pp_tile128
__device__ __forceinline__ f32x4 gemm_pp_tile128_slice(const Ctx& cx, const float* __restrict__ L, const float* __restrict__ R,
                                                    int ti, int tj, int g0) {
  const int lr = cx.lr, lq = cx.lq;
  const int pa = 16 * ti + lr, pb = 16 * tj + lr;
  const float* la = L + pa * TS + 4 * (lq ^ swz(pa)) + 16 * g0;
  const float* rb = R + pb * TS + 4 * (lq ^ swz(pb)) + 16 * g0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float4 a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    a[u] = *reinterpret_cast<const float4*>(la + 16 * u);
    b[u] = *reinterpret_cast<const float4*>(rb + 16 * u);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    acc0 = mfma16(a[u].x, b[u].x, acc0);
    acc1 = mfma16(a[u].y, b[u].y, acc1);
    acc0 = mfma16(a[u].z, b[u].z, acc0);
    acc1 = mfma16(a[u].w, b[u].w, acc1);
  }
  f32x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = acc0[r] + acc1[r];
  return acc;
}

// Two tiles at once (dA on the padded 4 x 4 grid: tiles `wave` and `wave + 8`), one accumulator each
__device__ __forceinline__ void gemm_pp_pair128(const Ctx& cx, f32x4& acc0, f32x4& acc1, const float* __restrict__ L,
                                                const float* __restrict__ R, int t0, int t1) {
  const int lr = cx.lr, lq = cx.lq;
  const int pa0 = min(16 * (t0 >> 2) + lr, P - 1), pb0 = min(16 * (t0 & 3) + lr, P - 1);
  const int pa1 = min(16 * (t1 >> 2) + lr, P - 1), pb1 = min(16 * (t1 & 3) + lr, P - 1);
  const float* la = L + pa0 * TS + 4 * (lq ^ swz(pa0));
  const float* rb = R + pb0 * TS + 4 * (lq ^ swz(pb0));
  const float* lc = L + pa1 * TS + 4 * (lq ^ swz(pa1));
  const float* rd = R + pb1 * TS + 4 * (lq ^ swz(pb1));
  constexpr int GB = 2, NG = C / 16 / GB;          // 8 blocks of 2 groups
  float4 a[GB], b[GB], c[GB], d[GB];
#pragma unroll
  for (int u = 0; u < GB; ++u) {
    a[u] = *reinterpret_cast<const float4*>(la + 16 * u); b[u] = *reinterpret_cast<const float4*>(rb + 16 * u);
    c[u] = *reinterpret_cast<const float4*>(lc + 16 * u); d[u] = *reinterpret_cast<const float4*>(rd + 16 * u);
  }
#pragma unroll
  for (int blk = 0; blk < NG; ++blk) {
    float4 an[GB], bn[GB], cn[GB], dn[GB];
    const int nb = blk + 1 < NG ? blk + 1 : blk;
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      an[u] = *reinterpret_cast<const float4*>(la + 16 * (GB * nb + u)); bn[u] = *reinterpret_cast<const float4*>(rb + 16 * (GB * nb + u));
      cn[u] = *reinterpret_cast<const float4*>(lc + 16 * (GB * nb + u)); dn[u] = *reinterpret_cast<const float4*>(rd + 16 * (GB * nb + u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      acc0 = mfma16(a[u].x, b[u].x, acc0); acc1 = mfma16(c[u].x, d[u].x, acc1);
      acc0 = mfma16(a[u].y, b[u].y, acc0); acc1 = mfma16(c[u].y, d[u].y, acc1);
      acc0 = mfma16(a[u].z, b[u].z, acc0); acc1 = mfma16(c[u].z, d[u].z, acc1);
      acc0 = mfma16(a[u].w, b[u].w, acc0); acc1 = mfma16(c[u].w, d[u].w, acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < GB; ++u) { a[u] = an[u]; b[u] = bn[u]; c[u] = cn[u]; d[u] = dn[u]; }
  }
}

// col_reduce_edge of ppm_fused_shared.h with the cross-lane sums on the vector ALU (rows_sum / wave_sum_valu, common.h): on
// return every lane holds in v[n] the sum over all C rows of its column 16 n + lr and in ve the total of column 48.
__device__ __forceinline__ void col_reduce_e(float (&v)[NTM], float& ve, float* part, int wave, int lane) {
  const int lr = lane & 15;
  const float se = wave_sum_valu(ve);
#pragma unroll
  for (int n = 0; n < NTM; ++n) {
    const float sn = rows_sum(v[n]);
    if (lane < 16) part[wave * 64 + 16 * n + lane] = sn;
  }
  if (lane == 0) part[wave * 64 + 48] = se;
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NTM; ++n) {
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

// C x P contraction on three column tiles with column 48 riding along on the vector ALU: the A values a lane loads for the
// MFMAs - element (row 16 (wave + 8 m) + lr, k = 4 ks + lq), already scaled - are exactly what the 49th column needs, so it
// costs one more B value per k-step (fe(ks) = B[k][48]) and MT FMAs, hidden between the MFMAs, instead of a pass of its own
// over the tile.  e[m] comes back summed over the four k groups: every lane holds the value of row 16 (wave + 8 m) + lr.
template <class FA, class FS, class FB, class FE>
__device__ __forceinline__ void gemm_cp_e(f32x4 (&acc)[MT][NTM], float (&e)[MT], FA fa, FS fs, FB fb, FE fe) {
  float a[MT], b[NTM], sc, be;
#pragma unroll
  for (int m = 0; m < MT; ++m) { a[m] = fa(m, 0); e[m] = 0.f; }
#pragma unroll
  for (int n = 0; n < NTM; ++n) b[n] = fb(n, 0);
  sc = fs(0); be = fe(0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float an[MT], bn[NTM], scn, ben;
    const int kn = ks + 1 < KS ? ks + 1 : ks;          // last step re-reads its own operands (never used)
#pragma unroll
    for (int m = 0; m < MT; ++m) an[m] = fa(m, kn);
#pragma unroll
    for (int n = 0; n < NTM; ++n) bn[n] = fb(n, kn);
    scn = fs(kn); ben = fe(kn);
    __builtin_amdgcn_sched_barrier(0);                 // keep the requests above ahead of the MFMAs below
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float am = a[m] * sc;
      e[m] = __builtin_fmaf(am, be, e[m]);
#pragma unroll
      for (int n = 0; n < NTM; ++n) acc[m][n] = mfma16(am, b[n], acc[m][n]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = an[m];
#pragma unroll
    for (int n = 0; n < NTM; ++n) b[n] = bn[n];
    sc = scn; be = ben;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) e[m] = rows_sum(e[m]);
}

typedef const Dirs __attribute__((address_space(4))) * dirs_kernarg_t;
typedef const pixpro_dir __attribute__((address_space(4))) * dir_kernarg_t;
// scalar loads of a struct from the kernel-argument segment (the host pass of hipcc only needs this to parse)
template <class T>
__device__ __forceinline__ T kernarg_load(const T __attribute__((address_space(4))) * p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  return T{};
#endif
}

// One (direction, sample) item, inlined into the persistent loop of the kernel.  The lane coordinates come from an opaque
// copy of threadIdx.x (lane_ctx) so that the body's addresses are recomputed per item instead of being hoisted out of the
// loop (256 VGPRs + spills); a non-inlined function was the first cure and cost ~60 callee-saved VGPR spills per call
// (profiles/NOTES_r04.md).  `flip` selects which tile buffer plays T0 for this item.
__device__ __forceinline__ void nhwc_item(const pixpro_problem& pb, dirs_kernarg_t Dp, float* __restrict__ loss_n,
                                          float* __restrict__ pos_num, float* __restrict__ pos_mean, int b, int nitems,
                                          int flip) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const Smem sm = carve(smem_raw);
  const int N = pb.N;
  const Ctx cx = lane_ctx();
  const int tid = cx.tid, wave = cx.wave, lane = cx.lane, lr = cx.lr, lq = cx.lq;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int tile = C * P;
  // A wave owns the 32 contiguous channels 32 wave .. 32 wave + 31 (two row tiles m) - one 128-B line of every pixel's row,
  // which lets the gradients leave straight from the accumulator registers in full lines.
  // A operand of the C x P products: element (channel 32 wave + 16 m + lr, pixel k = 4 ks + lq).  swz(k) = ks & 3 is
  // uniform, and bits 2..3 of the lane offset hold lr >> 2 and nothing else (TS = 256 + 16), so the swizzle is one XOR.
  const int aoff = lq * TS + 32 * wave + 4 * (lr >> 2) + (lr & 3);
  auto a_at = [&](const float* T, int m, int ks) { return T[(aoff ^ (4 * (ks & 3))) + 4 * ks * TS + 16 * m]; };
  // accumulator layout: register r of tile (m, nn) is (channel 32 wave + 16 m + 4 lq + r, pixel 16 nn + lr): the four r are
  // one 16-B chunk of the pixel's row
  auto c_at = [&](int m, int nn) { return tix(16 * nn + lr, 32 * wave + 16 * m + 4 * lq); };
  auto crow = [&](int m, int r) { return 32 * wave + 16 * m + 4 * lq + r; };
  // column 48 lives outside the MFMA tile grid: value e[m] of row erow(m), the same in the four lanes lr, lr + 16, ...; the
  // lanes below 16 speak for it in sums and stores
  auto erow = [&](int m) { return 32 * wave + 16 * m + lr; };
  const bool elive = lane < 16;
  float* T0 = sm.T + (flip ? TILE : 0);
  float* T1 = sm.T + (flip ? 0 : TILE);
#ifdef PIXPRO_STAMPS
  // the last work-group stamps its first item into entries 0.. and its last item into entries 32..
  const int stamp_on = blockIdx.x != gridDim.x - 1 ? -1 : b + (int)gridDim.x >= nitems ? 32 : b == (int)blockIdx.x ? 0 : -1;
#endif
  STAMPN(0);

  const int dir = b / N, n = b % N;
  const pixpro_dir dd = kernarg_load<pixpro_dir>(&Dp->d[dir]);
  const float* gx = dd.x + (size_t)n * tile;
  const float* gv = dd.v + (size_t)n * tile;
  const float* gk = dd.k + (size_t)n * tile;
  const bool want_grad = dd.dx != nullptr;
  const int bn = b + gridDim.x;                 // this work-group's next item
  const float* gx_next = bn < nitems ? Dp->d[bn / N].x + (size_t)(bn % N) * tile : nullptr;

  // x has landed in T0 (requested before the loop / during the previous item); every wave is done with the previous
  // item's tiles
  wait_vm<0>();
  __syncthreads();
  STAMPN(1);
  // v streams in under the S contraction (one tile request at a time).  A wave is stalled for ~2 k cycles while it issues
  // its rows (all CUs ask the memory system for a tile at the same moment), so the two MFMA waves of a SIMD take turns:
  // waves 4..6 request before their tile product, waves 1..3 after it - one of each pair always has MFMAs to issue.
  if (wave >= 4) tile_dma_mid(cx, gv, T1);

  // lazy flow mode with a chain (nframe > 2): the nine composed-flow pixels each point needs (common.h) are spread over
  // the work-group, two dependent stages through LDS (the AA matrix is not live yet)
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
  // ---- wave 0: the q-side sample points - bin centres, flow taps (ONE round trip) - the longest latency chain of the phase;
  // the k-side centres and the threshold of the positive-pair test ride on wave 7 behind its row-48 dot products
  float ptx = 0.f, pty = 0.f;
  bool ptv = true, dense = false;
  PointTaps taps;      // written and read by wave 0 in dense mode only (not zero-initialised on purpose, see ppm_fused.hip)
  FlowGeom fgd;
  if (wave == 0) {
    // this chain and wave 7's work below share their SIMDs with one MFMA wave each and are the longer halves: they go first
    __builtin_amdgcn_s_setprio(3);
    int Ho, Wo;
    frame_size_uniform(pb, dd, Ho, Wo);
    const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
    const Crop cq = crop_of_uniform(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
    const int pt = min(lane, P - 1), row = pt / pb.W, col = pt % pb.W;
    ptx = centre_x(cq, col, wo1); pty = centre_y(cq, row, ho1);
    STAMPN(22);
    dense = dd.lo_own == nullptr && dd.flow != nullptr;
    if (dense) {
      fgd = flow_geom(pb.Hf, pb.Wf, Ho, Wo);
      taps = displace_fetch(dd.flow + (size_t)n * 2 * pb.Hf * pb.Wf,
                            dd.mask ? dd.mask + (size_t)n * pb.Hf * pb.Wf : nullptr, fgd, ptx, pty);
    }
    STAMPN(23);
    if (dd.lo_own) {            // lazy flow mode: evaluate flow + validity at this point from the low-res maps
      const FlowGeom fg = flow_geom(8 * pb.lo_h, 8 * pb.lo_w, Ho, Wo);
      const LazySrc ls = lazy_src(pb, dd, n);
      float ox, oy;
      if (lazy_spread) lazy_finish(ls, fg, lazy_point(fg, ptx, pty), ptx, pty, sm.AA + 10 * pt, sm.AA + 10 * P + 8 * pt, ox, oy, ptv);
      else displace_point_lazy(ls, fg, ptx, pty, ox, oy, ptv);
      ptx = ox; pty = oy;
    }
    STAMPN(24);
  }

  // ---- S_raw = x^T x ; rx from its diagonal ; S, A, D ----
  // S is symmetric and 49 = 48 + 1: the six upper tiles of the 3 x 3 tile grid go to the matrix cores, one per wave
  // 1..6, and are mirrored by the epilogue; row 48 (49 dot products over C) runs on the vector ALU, an eighth of C per wave;
  // wave 0 has the q-side sample points (a latency chain: crop records -> flow taps), wave 7 the k side and the threshold.
  const int sw_ = wave - 1;                                                            // tile of waves 1..6
  const int sti = (sw_ >= 3) + (sw_ >= 5), stj = sw_ - (sw_ >= 3 ? 2 : 0) - (sw_ >= 5 ? 1 : 0);
  const bool stile = wave >= 1 && wave <= 6;
  f32x4 accS = zero4;
  if (stile) {
    accS = gemm_pp_tile128(cx, T0, T0, sti, stj);
    if (wave <= 3) tile_dma_mid(cx, gv, T1);
#ifdef PIXPRO_STAMPS
    if (stamp_on >= 0 && threadIdx.x == 64) g_stamps[stamp_on + 27] = __builtin_amdgcn_s_memtime();
#endif
    if (sti == stj) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * lq + r == lr) sm.nx2[16 * sti + lr] = accS[r];
    }
  } else if (wave == 7) {
#ifdef PIXPRO_STAMPS
    if (stamp_on >= 0 && threadIdx.x == 448) g_stamps[stamp_on + 28] = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_setprio(3);
    // the k-side bin centres and the threshold of the positive-pair test (they need only the two crop records)
    int Ho, Wo;
    frame_size_uniform(pb, dd, Ho, Wo);
    const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
    const Crop cq = crop_of_uniform(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
    const Crop ck = crop_of_uniform(dd.coord_k + (size_t)n * 10, pb.H, pb.W);
    const int pt = min(lane, P - 1), row = pt / pb.W, col = pt % pb.W;
    const bool live = lane < P;
    sm.kx[lane] = live ? centre_x(ck, col, wo1) : 0.f;
    sm.ky[lane] = live ? centre_y(ck, row, ho1) : 0.f;
    const float T = dist_threshold(fmaxf(bin_diag(cq, wo1, ho1), bin_diag(ck, wo1, ho1)), pb.pos_ratio, lane);
    if (lane == 0) sm.misc[2] = T;
    __builtin_amdgcn_s_setprio(0);
#ifdef PIXPRO_STAMPS
    if (stamp_on >= 0 && threadIdx.x == 448) g_stamps[stamp_on + 29] = __builtin_amdgcn_s_memtime();
#endif
  }
  // Row 48 of the raw Gram matrix (49 dot products over C) on the vector ALU, split over the channels: every wave takes its
  // 32 (8 chunks requested at once; one wave alone needed 5-6 k cycles for the 64), the partials meet in `part2`
  {
    const int jj = min(lane, P - 1), sj = swz(jj);
    const float* xj = T0 + jj * TS;
    const float* xe = T0 + (P - 1) * TS;            // swz(48) == 0
    float4 u[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      u[q] = *reinterpret_cast<const float4*>(xe + 4 * (8 * wave + q));
      w[q] = *reinterpret_cast<const float4*>(xj + 4 * ((8 * wave + q) ^ sj));
    }
    __builtin_amdgcn_sched_barrier(0);
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      e0 = __builtin_fmaf(u[q].x, w[q].x, e0); e1 = __builtin_fmaf(u[q].y, w[q].y, e1);
      e2 = __builtin_fmaf(u[q].z, w[q].z, e2); e3 = __builtin_fmaf(u[q].w, w[q].w, e3);
    }
    sm.part2[wave * 64 + lane] = (e0 + e1) + (e2 + e3);
  }
  if (wave == 0) {                 // the sample points: the taps requested above have had the S phase to arrive
    if (dense) {
      float ox, oy;
      displace_finish(taps, fgd, ptx, pty, ox, oy, ptv);
      ptx = ox; pty = oy;
    }
    const bool live = lane < P;
    sm.qx[lane] = live ? ptx : 0.f; sm.qy[lane] = live ? pty : 0.f;
    sm.valid[lane] = (live && ptv) ? 1 : 0;
    __builtin_amdgcn_s_setprio(0);
  }
  STAMPN(2);
  __syncthreads();                 // every wave is done with x in T0; nx2 / row 48 / the sample points are published
  // k streams in under the mask / S epilogue / norm phases; the same turn-taking: waves 4..7 request now, waves 0..3 after
  // the mask rows and the S epilogue
  if (wave >= 4) tile_dma(cx, gk, T0);
  STAMPN(3);
  // 1 / ||x_j|| from the diagonal of the raw Gram matrix: every lane works out the few it needs (the S epilogue below
  // runs before the next barrier), lanes 0..63 also publish the vector for the dX contraction
  auto row48 = [&](int j) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += sm.part2[w * 64 + j];
    return t;
  };
  const float n48 = row48(P - 1);
  auto rxof = [&](int j) { return rnorm(j < 48 ? sm.nx2[j] : n48); };
  if (tid < 64) sm.rx[tid] = rxof(min(tid, P - 1));
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
  STAMPN(4);
  // ---- S epilogue: A = pow(clamp(S)), D = dA/dS (needs nothing of v: it runs while v is still landing)
  if (stile) {
    const int j = 16 * stj + lr;
    const float rxj = rxof(j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * sti + 4 * lq + r;
      float a, d;
      pow_map(accS[r] * rxof(i) * rxj, pb.p, pb.clamp, a, d);
      sm.AA[i * LS + j] = a;
      sm.DM[i * LS + j] = d;
      if (sti != stj) { sm.AA[j * LS + i] = a; sm.DM[j * LS + i] = d; }
    }
  } else {
    // row / column 48, then the zero padding the contractions read: rows 49..51 and columns 49..63
    const int t2 = (wave == 7 ? 0 : 64) + lane;      // 0..127
    if (t2 < P) {
      float a, d;
      pow_map(row48(t2) * rxof(P - 1) * rxof(t2), pb.p, pb.clamp, a, d);
      sm.AA[(P - 1) * LS + t2] = a; sm.DM[(P - 1) * LS + t2] = d;
      sm.AA[t2 * LS + (P - 1)] = a; sm.DM[t2 * LS + (P - 1)] = d;
    }
    for (int e = t2; e < (LR - P) * 64; e += 2 * WAVE) {
      const int i = P + (e >> 6), jz = e & 63;
      sm.AA[i * LS + jz] = 0.f; sm.DM[i * LS + jz] = 0.f;
    }
    for (int e = t2; e < P * (64 - P); e += 2 * WAVE) {
      const int i = e / (64 - P), jz = P + e % (64 - P);
      sm.AA[i * LS + jz] = 0.f; sm.DM[i * LS + jz] = 0.f;
    }
  }
  if (wave < 4) tile_dma(cx, gk, T0);
  STAMPN(5);
  // ---- rv = 1 / ||v_j||.  v was requested at the top of the item; its copies are older than the (6 or 7) copies of k this
  // wave has just issued and copies complete in order, so "at most six outstanding" means v has landed.
  wait_vm<6>();
  __syncthreads();                 // v complete in T1 (and A, D, the mask rows, the counts, rx published)
  sm.part0[wave * 64 + lane] = col_sumsq_part(cx, T1);
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += sm.part0[w * 64 + tid];
    sm.rv[tid] = rnorm(t);
  }
  if (tid == 0) {
    float c = 0.f;
    for (int w = 0; w < NWAVES; ++w) c += sm.misc[8 + w];
    sm.misc[1] = c;                                    // read in the loss phase, several barriers from here
  }
  wait_vm<0>();                    // this wave's rows of k have landed (requested four phases ago) ...
  __syncthreads();                 // ... everybody's; rv published
  STAMPN(6);
  // partial sums of squares of k's columns now (part0 is free again): they are published by the barrier of the Y norms,
  // which then needs no second barrier for 1 / ||k||
  if (pb.normalize_k) sm.part0[wave * 64 + lane] = col_sumsq_part(cx, T0);

  // ---- Y = v_hat A : three column tiles on the matrix cores, column 48 as a dot product per row (edge_column) ----
  f32x4 Y[MT][NTM];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) Y[m][nn] = zero4;
  float Ye[MT];
  gemm_cp_e(Y, Ye,
            [&](int m, int ks) { return a_at(T1, m, ks); },
            [&](int ks) { return sm.rv[4 * ks + lq]; },
            [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; },
            [&](int ks) { return sm.AA[(4 * ks + lq) * LS + (P - 1)]; });
  STAMPN(7);
  // Column norms of Y and of k behind ONE barrier.  Y stays raw in its registers; every wave keeps its own copy of
  // 1 / ||k_j|| (rkw).
  float ry[NTM];     // 1 / ||Y[:, i]|| of this lane's columns
  float rye = 0.f;   // ... and of column 48
  {
    const float qe = wave_sum_valu(elive ? __builtin_fmaf(Ye[0], Ye[0], Ye[1] * Ye[1]) : 0.f);
    if (lane == 0) sm.part1[wave * 64 + 48] = qe;
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float q = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) q = __builtin_fmaf(Y[m][nn][r], Y[m][nn][r], q);
      q = rows_sum(q);
      if (lane < 16) sm.part1[wave * 64 + 16 * nn + lane] = q;
    }
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part1[w * 64 + 16 * nn + lr];
      ry[nn] = rnorm(t);
    }
    {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part1[w * 64 + 48];
      rye = rnorm(t);
    }
    float rkj = 1.f;
    if (pb.normalize_k) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part0[w * 64 + lane];
      rkj = rnorm(t);
    }
    sm.rkw[wave * 64 + lane] = rkj;         // wave-private
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (dd.pred) {
    float* gp = dd.pred + (size_t)n * tile;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) {
        const float4 o = make_float4(Y[m][nn][0] * ry[nn], Y[m][nn][1] * ry[nn], Y[m][nn][2] * ry[nn], Y[m][nn][3] * ry[nn]);
        *reinterpret_cast<float4*>(gp + (size_t)(16 * nn + lr) * C + crow(m, 0)) = o;
      }
    if (elive) {
#pragma unroll
      for (int m = 0; m < MT; ++m) gp[(size_t)(P - 1) * C + erow(m)] = Ye[m] * rye;
    }
  }

  STAMPN(8);
  // ---- G = k_hat M^T ----
  f32x4 G[MT][NTM];
  float Ge[MT];
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
    const unsigned long long rbe = (sm.mbits[P - 1] & ((1ull << P) - 1)) >> lq;      // row 48, shifted the same way
    const unsigned loe = (unsigned)rbe, hie = (unsigned)(rbe >> 32);
    gemm_cp_e(G, Ge,
              [&](int m, int ks) { return a_at(T0, m, ks); },
              [&](int ks) { return sm.rkw[wave * 64 + 4 * ks + lq]; },
              [&](int nn, int ks) {
                const unsigned w = ks < 8 ? lo[nn] : hi[nn];
                return (float)((w >> ((4 * ks) & 31)) & 1u);
              },
              [&](int ks) { return (float)(((ks < 8 ? loe : hie) >> ((4 * ks) & 31)) & 1u); });
  }
  STAMPN(9);
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
    // (a scratch of its own: slower waves may still be reading the norm partials in part0 / part1); barrier: every wave
    // is done reading k
    float ve = elive ? __builtin_fmaf(Ye[0], Ge[0], Ye[1] * Ge[1]) : 0.f;
    col_reduce_e(v, ve, sm.part2, cx.wave, cx.lane);
    tce = ve * rye;
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) tc[nn] = v[nn] * ry[nn];
  }
  const float cnt = sm.misc[1];
  const float den = cnt + 1e-6f;
  if (wave == 0) {
    float t = 0.f;
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) t += lane < 16 ? tc[nn] : 0.f;
    if (lane == 0) t += tce;
    t = wave_sum_valu(t);
    if (lane == 0) {
      // (write-through at agent scope: the work-group that folds the per-sample losses reads it on another XCD, see the kernel tail)
      __hip_atomic_store(loss_n + b, t / den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pos_num[b] = cnt;
      pos_mean[b] = cnt / (float)P / (float)P;
    }
  }
  STAMPN(10);
  if (!want_grad) {
    // forward only: v (T1) and k (T0) are dead behind the barrier of the loss reduction; the next x goes to T1
    if (gx_next) tile_dma(cx, gx_next, T1);
    return;
  }

  // ---- dY -> T0 (accumulator layout: one 16-B chunk per tile and lane) ----
  {
    const float gs = pb.grad_scale * -2.f / (float)N / den;
    if (elive) {
#pragma unroll
      for (int m = 0; m < MT; ++m) T0[tix(P - 1, erow(m))] = __builtin_fmaf(-Ye[m], rye * rye * (gs * tce), Ge[m] * (gs * rye));
    }
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      // (gs G - Y_hat ti) ry with Y_hat = Y ry, ti = gs <Y_hat, G>:  G (gs ry) - Y (ry ry ti)
      const float ryv = ry[nn], ga = gs * ryv, yb = ryv * ryv * (gs * tc[nn]);
#pragma unroll
      for (int m = 0; m < MT; ++m)
        *reinterpret_cast<float4*>(T0 + c_at(m, nn)) =
            make_float4(__builtin_fmaf(-Y[m][nn][0], yb, G[m][nn][0] * ga), __builtin_fmaf(-Y[m][nn][1], yb, G[m][nn][1] * ga),
                        __builtin_fmaf(-Y[m][nn][2], yb, G[m][nn][2] * ga), __builtin_fmaf(-Y[m][nn][3], yb, G[m][nn][3] * ga));
    }
  }
  __syncthreads();
  STAMPN(11);

  // ---- dA = dY^T v_hat ; dS = dA o D, in place of D ----
  // 49 = 48 + 1 once more: the 3 x 3 tiles over i, j < 48 go to the matrix cores - tiles 0..7 one per wave, tile (2, 2) as four
  // K slices on waves 4..7 (a SIMD holds waves w and w + 4: 64 + 80 MFMAs each, against 256 on the padded 4 x 4 grid) - and
  // row 48 / column 48 are dot products on the vector ALU, every wave over its own 32 channels.  The slices and the edge
  // partials meet in LDS (the four reduction scratch arrays, idle here) and are folded behind the dV contraction's barrier.
  float* const PK = sm.part0;                    // [4][256] tile (2,2) slices | [8][64] row-48 partials | [8][64] column-48 partials
  {
    const int mi = wave / 3, nj = wave - 3 * mi;
    const f32x4 acc = gemm_pp_tile128(cx, T0, T1, mi, nj);
    const int j = 16 * nj + lr;
    const float rvj = sm.rv[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * mi + 4 * lq + r;
      sm.DM[i * LS + j] = acc[r] * rvj * sm.DM[i * LS + j];
    }
    if (wave >= 4) {
      const f32x4 part = gemm_pp_tile128_slice(cx, T0, T1, 2, 2, 4 * (wave - 4));
#pragma unroll
      for (int r = 0; r < 4; ++r) PK[(wave - 4) * 256 + (4 * lq + r) * 16 + lr] = part[r];
    }
    // edges: lane j < 49: dA[48][j] over this wave's channels; lane i < 48: dA[i][48]
    const int pj = min(lane, P - 1), sj = swz(pj);
    const float* yr = T0 + pj * TS;              // dY row of pixel pj (column edge), v row of pixel pj (row edge)
    const float* vr = T1 + pj * TS;
    const float* y48 = T0 + (P - 1) * TS;        // swz(48) == 0
    const float* v48 = T1 + (P - 1) * TS;
    float er0 = 0.f, er1 = 0.f, ec0 = 0.f, ec1 = 0.f;
#pragma unroll
    for (int q0 = 0; q0 < 8; q0 += 4) {           // four chunks of each of the four rows requested at a time
      float4 a48[4], b48[4], bj[4], ai[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ql = 8 * wave + q0 + q;
        a48[q] = *reinterpret_cast<const float4*>(y48 + 4 * ql); b48[q] = *reinterpret_cast<const float4*>(v48 + 4 * ql);
        bj[q] = *reinterpret_cast<const float4*>(vr + 4 * (ql ^ sj)); ai[q] = *reinterpret_cast<const float4*>(yr + 4 * (ql ^ sj));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        er0 = __builtin_fmaf(a48[q].x, bj[q].x, er0); er1 = __builtin_fmaf(a48[q].y, bj[q].y, er1);
        er0 = __builtin_fmaf(a48[q].z, bj[q].z, er0); er1 = __builtin_fmaf(a48[q].w, bj[q].w, er1);
        ec0 = __builtin_fmaf(ai[q].x, b48[q].x, ec0); ec1 = __builtin_fmaf(ai[q].y, b48[q].y, ec1);
        ec0 = __builtin_fmaf(ai[q].z, b48[q].z, ec0); ec1 = __builtin_fmaf(ai[q].w, b48[q].w, ec1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PK[1024 + wave * 64 + lane] = er0 + er1;
    PK[1536 + wave * 64 + lane] = ec0 + ec1;
  }
  STAMPN(12);
  // (no barrier: the dV contraction reads dY and A, not DM; dS is first read by the symmetrisation behind two barriers)

  // ---- dV_hat = dY A ; dv = (dV_hat - v_hat s) / n_v ----
  {
    f32x4 Tv[MT][NTM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) Tv[m][nn] = zero4;
    float Te[MT], ve[MT];
    gemm_cp_e(Tv, Te,
              [&](int m, int ks) { return a_at(T0, m, ks); },
              [&](int) { return 1.f; },
              [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; },
              [&](int ks) { return sm.AA[(4 * ks + lq) * LS + (P - 1)]; });
#pragma unroll
    for (int m = 0; m < MT; ++m) ve[m] = T1[tix(P - 1, erow(m))];
    float ves = elive ? __builtin_fmaf(ve[0], Te[0], ve[1] * Te[1]) * sm.rv[P - 1] : 0.f;     // column 48: partial of the projection dot product
    float4 vh[MT][NTM];
    float v[NTM];
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        vh[m][nn] = *reinterpret_cast<const float4*>(T1 + c_at(m, nn));        // raw v: rv folded into the factors below
        s = __builtin_fmaf(vh[m][nn].x, Tv[m][nn][0], s); s = __builtin_fmaf(vh[m][nn].y, Tv[m][nn][1], s);
        s = __builtin_fmaf(vh[m][nn].z, Tv[m][nn][2], s); s = __builtin_fmaf(vh[m][nn].w, Tv[m][nn][3], s);
      }
      v[nn] = s * sm.rv[16 * nn + lr];
    }
    STAMPN(13);
    col_reduce_e(v, ves, sm.part3, cx.wave, cx.lane);          // barrier inside: every wave is done reading dY from T0
    STAMPN(14);
    if (wave >= 4) tile_dma(cx, gx, T0);                 // x again (second read: L2 / MALL hit), under the dv epilogue ...
    // dv = (dV_hat - v_hat s) / n_v leaves straight from the accumulator registers: the two row tiles of a wave are the two
    // 64-B halves of one line per pixel, so no staging in LDS, no barrier and no second pass over the tile
    {
      float* gdv = dd.dv + (size_t)n * tile;
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) {
        const float rvj = sm.rv[16 * nn + lr], cj = rvj * rvj * v[nn];      // (T - v_hat s) rv = T rv - v (rv rv s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          *reinterpret_cast<float4*>(gdv + (size_t)(16 * nn + lr) * C + crow(m, 0)) =
              make_float4(__builtin_fmaf(-vh[m][nn].x, cj, Tv[m][nn][0] * rvj), __builtin_fmaf(-vh[m][nn].y, cj, Tv[m][nn][1] * rvj),
                          __builtin_fmaf(-vh[m][nn].z, cj, Tv[m][nn][2] * rvj), __builtin_fmaf(-vh[m][nn].w, cj, Tv[m][nn][3] * rvj));
      }
      if (elive) {
        const float rve = sm.rv[P - 1];
#pragma unroll
        for (int m = 0; m < MT; ++m) gdv[(size_t)(P - 1) * C + erow(m)] = __builtin_fmaf(-ve[m], rve * rve * ves, Te[m] * rve);
      }
    }
  }
  if (wave < 4) tile_dma(cx, gx, T0);                // ... the other half of its rows (turn-taking, see the S phase)
  STAMPN(15);
  // dSs = dS + dS^T -> AA (dead since the dV contraction), rows / columns 0..48 (the zero padding of the S epilogue stays).
  // dS of the 3 x 3 tiles is in DM, tile (2,2) excepted; that tile, row 48 and column 48 are folded here from the partials of
  // the dA phase (complete behind the barrier of the dV reduction), in a fixed order.  Three uniform passes on disjoint
  // waves (one pass with the fold behind per-element branches took 11 k cycles):
  if (wave < 6) {
    // plain region: i, j < 48 without the (2,2) tile - 8 of the 9 tile positions = 2048 elements on 384 threads
    for (int e = tid; e < 2048; e += 6 * WAVE) {
      const int t = e >> 8, r = (e >> 4) & 15, c = e & 15;          // t: tile 0..7 in row-major order of the 3 x 3 grid
      const int i = 16 * (t / 3) + r, j = 16 * (t - 3 * (t / 3)) + c;
      // a partner in the (2,2) block cannot occur: (i, j) outside it means (j, i) outside it
      sm.AA[i * LS + j] = sm.DM[i * LS + j] + sm.DM[j * LS + i];
    }
  } else {
    const int t2 = tid - 6 * WAVE;                                    // 0..127
    auto f22 = [&](int i, int j) {                                    // dS of tile (2,2): its four K slices
      const int e2 = (i - 32) * 16 + (j - 32);
      return ((PK[e2] + PK[256 + e2]) + (PK[512 + e2] + PK[768 + e2])) * sm.rv[j] * sm.DM[i * LS + j];
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = t2 + 128 * u, i = 32 + (e >> 4), j = 32 + (e & 15);
      sm.AA[i * LS + j] = f22(i, j) + f22(j, i);
    }
    if (t2 < P) {                                                     // row / column 48: dSs[48][j] = dS[48][j] + dS[j][48]
      const int j = t2;
      float fr = 0.f, fc = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) { fr += PK[1024 + w * 64 + j]; fc += PK[1536 + w * 64 + j]; }
      const float dr = fr * sm.rv[j] * sm.DM[(P - 1) * LS + j];                          // dS[48][j]
      const float dc = j < P - 1 ? fc * sm.rv[P - 1] * sm.DM[j * LS + (P - 1)] : dr;      // dS[j][48]  (j = 48: the same entry)
      sm.AA[(P - 1) * LS + j] = dr + dc;
      sm.AA[j * LS + (P - 1)] = dr + dc;
    }
  }
  STAMPN(16);
  wait_vm<0>();                                      // this wave's rows of x have landed (and its dv stores have left)
  __syncthreads();                                   // ... everybody's; dSs complete; every wave is done reading dv from T1
  STAMPN(17);

  // ---- dX_hat = x_hat dSs ; dx = (dX_hat - x_hat s) / n_x ----
  // T1 is free behind the barrier above (dv has left): the next item's x travels from here on, requested by waves 0..3 in
  // front of the contraction ...
  if (wave < 4 && gx_next) tile_dma(cx, gx_next, T1);
  {
    f32x4 Tx[MT][NTM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) Tx[m][nn] = zero4;
    float Te[MT], xe[MT];
    gemm_cp_e(Tx, Te,
              [&](int m, int ks) { return a_at(T0, m, ks); },
              [&](int ks) { return sm.rx[4 * ks + lq]; },
              [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; },
              [&](int ks) { return sm.AA[(4 * ks + lq) * LS + (P - 1)]; });
    if (wave >= 4 && gx_next) tile_dma(cx, gx_next, T1);     // ... by waves 4..7 behind it
#pragma unroll
    for (int m = 0; m < MT; ++m) xe[m] = T0[tix(P - 1, erow(m))];
    float xes = elive ? __builtin_fmaf(xe[0], Te[0], xe[1] * Te[1]) * sm.rx[P - 1] : 0.f;
    float4 xh[MT][NTM];
    float v[NTM];
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xh[m][nn] = *reinterpret_cast<const float4*>(T0 + c_at(m, nn));        // raw x
        s = __builtin_fmaf(xh[m][nn].x, Tx[m][nn][0], s); s = __builtin_fmaf(xh[m][nn].y, Tx[m][nn][1], s);
        s = __builtin_fmaf(xh[m][nn].z, Tx[m][nn][2], s); s = __builtin_fmaf(xh[m][nn].w, Tx[m][nn][3], s);
      }
      v[nn] = s * sm.rx[16 * nn + lr];
    }
    STAMPN(18);
    col_reduce_e(v, xes, sm.part1, cx.wave, cx.lane);          // barrier inside: every wave is done reading x from T0
    STAMPN(19);
    {
      float* gdx = dd.dx + (size_t)n * tile;
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) {
        const float rxj = sm.rx[16 * nn + lr], cj = rxj * rxj * v[nn];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          *reinterpret_cast<float4*>(gdx + (size_t)(16 * nn + lr) * C + crow(m, 0)) =
              make_float4(__builtin_fmaf(-xh[m][nn].x, cj, Tx[m][nn][0] * rxj), __builtin_fmaf(-xh[m][nn].y, cj, Tx[m][nn][1] * rxj),
                          __builtin_fmaf(-xh[m][nn].z, cj, Tx[m][nn][2] * rxj), __builtin_fmaf(-xh[m][nn].w, cj, Tx[m][nn][3] * rxj));
      }
      if (elive) {
        const float rxe = sm.rx[P - 1];
#pragma unroll
        for (int m = 0; m < MT; ++m) gdx[(size_t)(P - 1) * C + erow(m)] = __builtin_fmaf(-xe[m], rxe * rxe * xes, Te[m] * rxe);
      }
    }
  }
  STAMPN(20);
  STAMPN(21);
}

__global__ void __launch_bounds__(NTHREADS)
ppm_nhwc_kernel(pixpro_problem pb, Dirs D, float* __restrict__ loss_n, float* __restrict__ pos_num,
                float* __restrict__ pos_mean, int nitems, float* __restrict__ loss, unsigned* __restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const Smem sm = carve(smem_raw);
  // kernel-argument segment: pb at offset 0, D right behind it (both 8-byte aligned)
  const char __attribute__((address_space(4))) * ka =
      (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
  static_assert(sizeof(pixpro_problem) % 8 == 0, "D follows pb without padding");
  const dirs_kernarg_t Dp = (dirs_kernarg_t)(ka + sizeof(pixpro_problem));
  const int tid = threadIdx.x;
#ifdef PIXPRO_STAMPS
  if (blockIdx.x == gridDim.x - 1 && tid == 0) g_stamps[30] = __builtin_amdgcn_s_memtime();
#endif
  for (int e = tid; e < (TROWS - P) * TS; e += NTHREADS) { sm.T[P * TS + e] = 0.f; sm.T[TILE + P * TS + e] = 0.f; }
  int b = blockIdx.x;
  if (b < nitems) {
    tile_dma(lane_ctx(), D.d[b / pb.N].x + (size_t)(b % pb.N) * (C * P), sm.T);
    int flip = 0;
#pragma clang loop unroll(disable)
    for (; b < nitems; b += gridDim.x) {
      nhwc_item(pb, Dp, loss_n, pos_num, pos_mean, b, nitems, flip);
      flip ^= 1;
    }
  }
  // ---- loss = -2 * sum_d mean_n loss_n (PixPro.py:247,431) by the work-group that finishes last: no second launch ----
  // Wave 0 wrote this work-group's loss_n entries (lane 0, write-through stores at agent scope).  vmcnt(0) = those stores are
  // acknowledged at the device's coherence point; only then the ticket is drawn.  The work-group that draws the last ticket
  // reads all entries back with agent-scope loads (they bypass this XCD's L2, which may hold the neighbours of its own entries
  // in the same line), adds them in a fixed order and puts the ticket back to zero for the next launch on the stream.
  // No cache write-back or invalidate anywhere: the only data that crosses work-groups is read and written past the L2.
  if (ticket && tid < WAVE) {
    int last = 0;
    if (tid == 0) {
      wait_vm<0>();
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    if (__builtin_amdgcn_readfirstlane(last)) {
      float acc = 0.f;
      for (int e = tid; e < nitems; e += WAVE)
        acc += __hip_atomic_load(loss_n + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc = wave_sum_valu(acc);
      if (tid == 0) {
        *loss = -2.f * acc / (float)pb.N;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

bool fused_nhwc_supported(const pixpro_problem& pb) {
  return pb.layout == 1 && pb.H * pb.W == P && pb.C == C && !getenv("PIXPRO_NO_NHWC_KERNEL");
}

// One ticket word per (device, stream): launches on one stream run one after the other, so the word is back at zero when the
// next one starts; launches on different streams never share a word.  The pool is allocated on first use (never while the
// stream is being captured into a graph: nullptr then, and the caller falls back to the separate k_finalize launch).
namespace {
constexpr int TICKET_SLOTS = 4096;
struct TicketPool {
  unsigned* base = nullptr;
  std::unordered_map<hipStream_t, int> slot;
};
unsigned* ticket_for(hipStream_t st) {
  static std::mutex mu;
  static std::map<int, TicketPool> pools;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> g(mu);
  TicketPool& pool = pools[dev];
  if (!pool.base) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    unsigned* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), TICKET_SLOTS * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, TICKET_SLOTS * sizeof(unsigned)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    pool.base = p;
  }
  auto it = pool.slot.find(st);
  if (it == pool.slot.end()) {
    if ((int)pool.slot.size() >= TICKET_SLOTS) return nullptr;
    it = pool.slot.emplace(st, (int)pool.slot.size()).first;
  }
  return pool.base + it->second;
}
}  // namespace

// *folded = 1: `loss` has been written by the kernel itself (no k_finalize launch needed)
int run_fused_nhwc(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                   float* loss, int* folded, hipStream_t st) {
  Dirs D;
  for (int d = 0; d < PIXPRO_MAX_DIRS; ++d) D.d[d] = dirs[d < pb.ndir ? d : 0];
  const size_t bytes = SMEM_FLOATS * sizeof(float);
  static thread_local bool configured = false;
  static thread_local int cus = 0;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppm_nhwc_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
      return PIXPRO_ERR_LAUNCH;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("hipGetDeviceProperties failed");
      return PIXPRO_ERR_LAUNCH;
    }
    cus = prop.multiProcessorCount;
    configured = true;
  }
  const int items = pb.N * pb.ndir;
  int grid = cus > 0 ? cus : 256;
  if (const char* e = getenv("PIXPRO_NHWC_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;      // tuning knob
  if (grid > items) grid = items;
  unsigned* ticket = (loss && !getenv("PIXPRO_NHWC_NO_FOLD")) ? ticket_for(st) : nullptr;
  *folded = ticket != nullptr;
  hipLaunchKernelGGL(ppm_nhwc_kernel, dim3(grid), dim3(NTHREADS), bytes, st, pb, D, loss_n, pos_num, pos_mean, items, loss,
                     ticket);
  return check_launch("ppm_nhwc_kernel");
}

}  // namespace pixpro
