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
//   * dS lives in place of D, and dS + dS^T is formed in place too (two P x P matrices instead of three), which is what
//     pays for the wider tile rows.
//
// Buffers over time (T0, T1 swap roles after every item):
//   T0: x (S) -> k (G) -> dY (dA, dV) -> x (dX, dx out)         T1: v (Y, dA, dv out) -> x of the next item
//
// Math: SURVEY.md Appendix A; reference contrast/models/PixPro.py:339-363,380,403 (PPM + the three F.normalize),
// :92-247 (regression_loss), :46-89 (add_optical_flow).
#include "ppm_fused_shared.h"

namespace pixpro {

#ifdef PIXPRO_STAMPS   // diagnostic build only (tools/ppm_phase_stamps.py --nhwc): wave 0 of the last work-group stamps s_memtime
extern __device__ unsigned long long g_stamps[64];            // defined in ppm_fused.hip; this kernel uses entries 32..63
#define STAMPN(i)                                                                                  \
  do {                                                                                             \
    if (stamp_on && threadIdx.x == 0) g_stamps[32 + (i)] = __builtin_amdgcn_s_memtime();            \
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

__device__ __forceinline__ int swz(int p) { return (p >> 2) & 3; }
// element (pixel p, channel c) of a tile
__device__ __forceinline__ int tix(int p, int c) { return p * TS + ((((c >> 2) ^ swz(p))) << 2) + (c & 3); }

struct Smem {
  float* T;      // [2][TILE]
  float* AA;     // [LR][LS]   A = pow(clamp(S))
  float* DM;     // [LR][LS]   dA/dS, then dS (in place), then dS + dS^T (in place)
  float* rx;     // [64] reciprocal column norms of x and v
  float* rv;
  float* nx2;    // [64] diag of the raw Gram matrix
  float* part0;  // [NWAVES][64] x2: scratch of the column reductions, used alternately
  float* part1;
  float* qx;     // [64] each
  float* qy;
  float* kx;
  float* ky;
  float* misc;   // [32]: [1] cnt, [2] threshold, [8 + w] per-wave positive counts
  unsigned long long* mbits;  // [64]
  int* valid;    // [64]
};
constexpr size_t SMEM_FLOATS = 2 * TILE + 2 * LR * LS + 3 * 64 + 2 * NWAVES * 64 + 4 * 64 + 32 + 2 * 64 + 64;

__device__ __forceinline__ Smem carve(float* base) {
  Smem s;
  float* p = base;
  s.T = p; p += 2 * TILE;
  s.AA = p; p += LR * LS;
  s.DM = p; p += LR * LS;
  s.rx = p; p += 64; s.rv = p; p += 64; s.nx2 = p; p += 64;
  s.part0 = p; p += NWAVES * 64; s.part1 = p; p += NWAVES * 64;
  s.qx = p; p += 64; s.qy = p; p += 64; s.kx = p; p += 64; s.ky = p; p += 64;
  s.misc = p; p += 32;
  s.mbits = (unsigned long long*)p; p += 2 * 64;
  s.valid = (int*)p;
  return s;
}

// HBM [P][C] -> LDS tile by LDS-DMA: wave w copies pixel rows w, w + 8, ...; lane l fetches logical chunk l ^ swz(p) into
// chunk position l.  Nothing orders the copy against LDS reads except s_waitcnt vmcnt + a barrier (tile_landed).
__device__ __forceinline__ void tile_dma(const float* __restrict__ g, float* T) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int p = wave + NWAVES * i;
    if (p < P) dma16(g + p * C + 4 * (lane ^ swz(p)), T + p * TS);
  }
}
// LDS tile -> HBM [P][C], 16 B per lane, one pixel row per wave instruction
__device__ __forceinline__ void tile_store(float* __restrict__ g, const float* T) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int p = wave + NWAVES * i;
    if (p < P)
      *reinterpret_cast<float4*>(g + p * C + 4 * (lane ^ swz(p))) = *reinterpret_cast<const float4*>(T + p * TS + 4 * lane);
  }
}

// 1 / max(||T[:, j]||, 1e-12)-style column sums of squares: lane = pixel, wave w covers channels 32 w .. 32 w + 31
__device__ __forceinline__ float col_sumsq_part(const float* T) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int p = min(lane, P - 1), s = swz(p);
  const float* row = T + p * TS;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(row + 4 * ((8 * wave + q) ^ s));
    a0 = __builtin_fmaf(t.x, t.x, a0); a1 = __builtin_fmaf(t.y, t.y, a1);
    a0 = __builtin_fmaf(t.z, t.z, a0); a1 = __builtin_fmaf(t.w, t.w, a1);
  }
  return a0 + a1;
}

// One 16 x 16 tile (ti, tj) of L^T R over K = C, both operands k-contiguous: lane (pixel lr, k group lq) reads the four
// channels 16 g + 4 lq .. + 3 of its pixel with ONE ds_read_b128 per operand and feeds four MFMA steps (the order of the
// contraction index is free as long as both operands agree).  Two interleaved accumulators; groups requested four ahead.
__device__ __forceinline__ f32x4 gemm_pp_tile128(const float* __restrict__ L, const float* __restrict__ R, int ti, int tj) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
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
// Two tiles at once (dA on the padded 4 x 4 grid: tiles `wave` and `wave + 8`), one accumulator each
__device__ __forceinline__ void gemm_pp_pair128(f32x4& acc0, f32x4& acc1, const float* __restrict__ L,
                                                const float* __restrict__ R, int t0, int t1) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
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

// Column 48 of a C x P product on the vector ALU: e(c) = sum_{k < 49} T[k][c] w[k] for the wave's 32 rows.  Lane l takes
// row (l & 31) of the wave's row list and half of k (l >> 5); w[k] is written by lane k into the wave's own 64-float LDS
// slice first (same wave writes and reads: no barrier).  Returns e in lanes 0..31 (duplicated in 32..63).
__device__ __forceinline__ float edge_column(const float* __restrict__ T, float wl, float* __restrict__ ws) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rl = lane & 31, hi = lane >> 5;
  const int row = 16 * (wave + NWAVES * (rl >> 4)) + (rl & 15);
  ws[lane] = wl;                            // lanes >= 49 pass 0
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float* w = ws + 25 * hi;
  float e0 = 0.f, e1 = 0.f;
#pragma unroll
  for (int k = 0; k < 24; k += 2) {
    e0 = __builtin_fmaf(T[tix(25 * hi + k, row)], w[k], e0);
    e1 = __builtin_fmaf(T[tix(25 * hi + k + 1, row)], w[k + 1], e1);
  }
  e0 = __builtin_fmaf(hi ? 0.f : T[tix(24, row)], w[24], e0);     // k = 24: lower half only (the upper half ends at 48 = 25 + 23)
  float e = e0 + e1;
  e += __shfl_xor(e, 32, 64);
  return e;
}
// The same for G = k_hat M^T: w[k] = M[48][k] / ||k_k|| is zero except for the few positive pairs of row 48, so the
// loop runs over the set bits of the row (wave-uniform).
__device__ __forceinline__ float edge_column_bits(const float* __restrict__ T, unsigned long long bits,
                                                  const float* __restrict__ rk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rl = lane & 31;
  const int row = 16 * (wave + NWAVES * (rl >> 4)) + (rl & 15);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bits), hi = __builtin_amdgcn_readfirstlane((unsigned)(bits >> 32));
  float e0 = 0.f, e1 = 0.f;
  while (lo | hi) {
    int k;
    if (lo) { k = __builtin_ctz(lo); lo &= lo - 1; } else { k = 32 + __builtin_ctz(hi); hi &= hi - 1; }
    e0 = __builtin_fmaf(T[tix(k, row)], rk[k], e0);
    if (lo | hi) {
      if (lo) { k = __builtin_ctz(lo); lo &= lo - 1; } else { k = 32 + __builtin_ctz(hi); hi &= hi - 1; }
      e1 = __builtin_fmaf(T[tix(k, row)], rk[k], e1);
    }
  }
  return e0 + e1;
}

typedef const pixpro_problem __attribute__((address_space(4))) * pb_kernarg_t;
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

// One (direction, sample) item.  Deliberately NOT inlined into the persistent loop of the kernel: inlined, the compiler
// hoists every lane-derived address of the body out of the loop (hundreds of registers live across the whole item:
// 256 VGPRs plus spills); as a function the body is allocated like a straight-line kernel.  pb / D are read from the
// kernel-argument segment (scalar loads), the LDS carve is recomputed from the dynamic LDS base.  `flip` selects which
// tile buffer plays T0 for this item.
__device__ __attribute__((noinline)) void nhwc_item(pb_kernarg_t pbp, dirs_kernarg_t Dp, float* __restrict__ loss_n,
                                                     float* __restrict__ pos_num, float* __restrict__ pos_mean, int b,
                                                     int nitems, int flip) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const pixpro_problem pb = kernarg_load<pixpro_problem>(pbp);
  const Smem sm = carve(smem_raw);
  const int N = pb.N;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15,
            lq = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int tile = C * P;
  // A operand of the C x P products: element (channel 16 (wave + 8 m) + lr, pixel k = 4 ks + lq).  swz(k) = ks & 3 is
  // uniform, and bits 2..3 of the lane offset hold lr >> 2 and nothing else (TS = 256 + 16), so the swizzle is one XOR.
  const int aoff = lq * TS + 16 * wave + 4 * (lr >> 2) + (lr & 3);
  auto a_at = [&](const float* T, int m, int ks) { return T[(aoff ^ (4 * (ks & 3))) + 4 * ks * TS + 128 * m]; };
  // accumulator layout: register r of tile (m, nn) is (channel 16 (wave + 8 m) + 4 lq + r, pixel 16 nn + lr): the four r are
  // one 16-B chunk of the pixel's row
  auto c_at = [&](int m, int nn) { return tix(16 * nn + lr, 16 * (wave + NWAVES * m) + 4 * lq); };
  auto crow = [&](int m, int r) { return 16 * (wave + NWAVES * m) + 4 * lq + r; };
  const int erow = 16 * (wave + NWAVES * ((lane & 31) >> 4)) + (lane & 15);     // row of this lane's edge value (lanes 0..31)
  const bool elive = lane < 32;
  float* T0 = sm.T + (flip ? TILE : 0);
  float* T1 = sm.T + (flip ? 0 : TILE);
#ifdef PIXPRO_STAMPS
  const bool stamp_on = blockIdx.x == gridDim.x - 1 && b + (int)gridDim.x >= nitems;     // the work-group's last item
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
  tile_dma(gv, T1);                   // v streams in under the S contraction (one tile request at a time)

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
  // ---- wave 0: bin centres, flow taps (ONE round trip, consumed after its share of the S phase), threshold ----
  float ptx = 0.f, pty = 0.f, pkx = 0.f, pky = 0.f;
  bool ptv = true, dense = false;
  PointTaps taps;      // written and read by wave 0 in dense mode only (not zero-initialised on purpose, see ppm_fused.hip)
  FlowGeom fgd;
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

  // ---- S_raw = x^T x ; rx from its diagonal ; S, A, D ----
  // S is symmetric and 49 = 48 + 1: the six upper tiles of the 3 x 3 tile grid go to the matrix cores, one per wave
  // 1..6, and are mirrored by the epilogue; waves 7 and 0 (wave 0 once its sample points are requested) take row 48
  // (49 dot products over half of C each) on the vector ALU.
  const int sw_ = wave - 1;                                                            // tile of waves 1..6
  const int sti = (sw_ >= 3) + (sw_ >= 5), stj = sw_ - (sw_ >= 3 ? 2 : 0) - (sw_ >= 5 ? 1 : 0);
  const bool stile = wave >= 1 && wave <= 6;
  f32x4 accS = zero4;
  if (stile) {
    accS = gemm_pp_tile128(T0, T0, sti, stj);
    if (sti == stj) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * lq + r == lr) sm.nx2[16 * sti + lr] = accS[r];
    }
  } else {
    const int half = wave == 7 ? 0 : 1, jj = min(lane, P - 1), sj = swz(jj);
    const float* xj = T0 + jj * TS;
    const float* xe = T0 + (P - 1) * TS;            // swz(48) == 0
    float e0 = 0.f, e1 = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) {
      const int ql = 32 * half + q;
      const float4 u = *reinterpret_cast<const float4*>(xe + 4 * ql);
      const float4 w = *reinterpret_cast<const float4*>(xj + 4 * (ql ^ sj));
      e0 = __builtin_fmaf(u.x, w.x, e0); e1 = __builtin_fmaf(u.y, w.y, e1);
      e0 = __builtin_fmaf(u.z, w.z, e0); e1 = __builtin_fmaf(u.w, w.w, e1);
    }
    sm.part1[half * 64 + lane] = e0 + e1;               // part1 is next used by the Y-norm reduction
  }
  if (wave == 0) {                 // the sample points: taps requested before the row-48 work
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
  STAMPN(2);
  wait_vm<0>();                    // this wave's rows of v have landed ...
  __syncthreads();                 // ... everybody's; every wave is done with x in T0
  tile_dma(gk, T0);                // k streams in under the mask / norm / S epilogue / Y phases
  STAMPN(3);
  if (tid < 64) {
    float n2 = sm.nx2[tid < 48 ? tid : 0];
    if (tid >= 48) n2 = sm.part1[48] + sm.part1[64 + 48];
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
  STAMPN(4);
  // ---- rv = 1 / ||v_j|| (the two barriers also publish rx, the mask rows and counts)
  sm.part0[wave * 64 + lane] = col_sumsq_part(T1);
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += sm.part0[w * 64 + tid];
    sm.rv[tid] = 1.f / fmaxf(sqrtf(t), 1e-12f);
  }
  __syncthreads();
  STAMPN(5);
  if (tid == 0) {
    float c = 0.f;
    for (int w = 0; w < NWAVES; ++w) c += sm.misc[8 + w];
    sm.misc[1] = c;                                    // read in the loss phase, several barriers from here
  }
  if (stile) {
    const int j = 16 * stj + lr;
    const float rxj = sm.rx[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * sti + 4 * lq + r;
      float a, d;
      pow_map(accS[r] * sm.rx[i] * rxj, pb.p, pb.clamp, a, d);
      sm.AA[i * LS + j] = a;
      sm.DM[i * LS + j] = d;
      if (sti != stj) { sm.AA[j * LS + i] = a; sm.DM[j * LS + i] = d; }
    }
  } else {
    // row / column 48, then the zero padding the contractions read: rows 49..51 and columns 49..63
    const int t2 = (wave == 7 ? 0 : 64) + lane;      // 0..127
    if (t2 < P) {
      float a, d;
      pow_map((sm.part1[t2] + sm.part1[64 + t2]) * sm.rx[P - 1] * sm.rx[t2], pb.p, pb.clamp, a, d);
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
  __syncthreads();
  STAMPN(6);

  // ---- Y = v_hat A : three column tiles on the matrix cores, column 48 as a dot product per row (edge_column) ----
  f32x4 Y[MT][NTM];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) Y[m][nn] = zero4;
  gemm_cp<MT, NTM>(Y, KS,
                   [&](int m, int ks) { return a_at(T1, m, ks); },
                   [&](int ks) { return sm.rv[4 * ks + lq]; },
                   [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; });
  float Ye = edge_column(T1, lane < P ? sm.rv[lane] * sm.AA[lane * LS + (P - 1)] : 0.f, sm.part0 + wave * 64);
  STAMPN(7);
  // Column norms of Y and of k with two barriers.  Y stays raw in its registers; every wave keeps its own copy of
  // 1 / ||k_j|| in its slice of part1.
  float ry[NTM];     // 1 / ||Y[:, i]|| of this lane's columns
  float rye = 0.f;   // ... and of column 48
  {
    const float qe = wave_sum(elive ? Ye * Ye : 0.f);
    if (lane == 0) sm.part1[wave * 64 + 48] = qe;
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
    wait_vm<0>();                           // this wave's rows of k have landed
    __syncthreads();                        // ... everybody's
    if (pb.normalize_k) sm.part0[wave * 64 + lane] = col_sumsq_part(T0);
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t += sm.part1[w * 64 + 16 * nn + lr];
      ry[nn] = 1.f / fmaxf(sqrtf(t), 1e-12f);
    }
    {
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
    if (elive) gp[(size_t)(P - 1) * C + erow] = Ye * rye;
  }

  STAMPN(8);
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
                     [&](int m, int ks) { return a_at(T0, m, ks); },
                     [&](int ks) { return sm.part1[wave * 64 + 4 * ks + lq]; },
                     [&](int nn, int ks) {
                       const unsigned w = ks < 8 ? lo[nn] : hi[nn];
                       return (float)((w >> ((4 * ks) & 31)) & 1u);
                     });
    Ge = edge_column_bits(T0, sm.mbits[P - 1] & ((1ull << P) - 1), sm.part1 + wave * 64);
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
    // a wave overwrites only its own slice (its 1 / ||k|| copy, no longer needed); barrier: every wave is done reading k
    float ve = elive ? Ye * Ge : 0.f;
    col_reduce_edge<NTM>(v, ve, sm.part1);
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
    t = wave_sum(t);
    if (lane == 0) {
      loss_n[b] = t / den;
      pos_num[b] = cnt;
      pos_mean[b] = cnt / (float)P / (float)P;
    }
  }
  STAMPN(10);
  if (!want_grad) {
    // forward only: v (T1) and k (T0) are dead behind the barrier of the loss reduction; the next x goes to T1
    if (gx_next) tile_dma(gx_next, T1);
    return;
  }

  // ---- dY -> T0 (accumulator layout: one 16-B chunk per tile and lane) ----
  {
    const float gs = pb.grad_scale * -2.f / (float)N / den;
    if (elive) T0[tix(P - 1, erow)] = __builtin_fmaf(-Ye, rye * rye * (gs * tce), Ge * (gs * rye));
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
  {
    f32x4 acc0 = zero4, acc1 = zero4;
    gemm_pp_pair128(acc0, acc1, T0, T1, wave, wave + NWAVES);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tt = wave + NWAVES * t;
      const f32x4& acc = t ? acc1 : acc0;
      const int mi = tt >> 2, nj = tt & 3, j = 16 * nj + lr;
      const float rvj = sm.rv[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * mi + 4 * lq + r;
        if (i < LR) sm.DM[i * LS + j] = acc[r] * rvj * sm.DM[i * LS + j];      // D is zero beyond P: so is dS
      }
    }
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
    gemm_cp<MT, NTM>(Tv, KS,
                     [&](int m, int ks) { return a_at(T0, m, ks); },
                     [&](int) { return 1.f; },
                     [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; });
    float Te = edge_column(T0, lane < P ? sm.AA[lane * LS + (P - 1)] : 0.f, sm.part1 + wave * 64);
    const float ve = T1[tix(P - 1, erow)];
    float ves = elive ? ve * Te * sm.rv[P - 1] : 0.f;     // column 48: partial of the projection dot product
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
    col_reduce_edge<NTM>(v, ves, sm.part0);          // barrier inside: every wave is done reading dY from T0
    STAMPN(14);
    tile_dma(gx, T0);                                // x again (second read: L2 / MALL hit), under the dv epilogue
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const float rvj = sm.rv[16 * nn + lr], cj = rvj * rvj * v[nn];      // (T - v_hat s) rv = T rv - v (rv rv s)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        *reinterpret_cast<float4*>(T1 + c_at(m, nn)) =
            make_float4(__builtin_fmaf(-vh[m][nn].x, cj, Tv[m][nn][0] * rvj), __builtin_fmaf(-vh[m][nn].y, cj, Tv[m][nn][1] * rvj),
                        __builtin_fmaf(-vh[m][nn].z, cj, Tv[m][nn][2] * rvj), __builtin_fmaf(-vh[m][nn].w, cj, Tv[m][nn][3] * rvj));
    }
    if (elive) {
      const float rve = sm.rv[P - 1];
      T1[tix(P - 1, erow)] = __builtin_fmaf(-ve, rve * rve * ves, Te * rve);
    }
  }
  __syncthreads();                                   // dv complete in T1; dS complete in DM
  STAMPN(15);
  tile_store(dd.dv + (size_t)n * tile, T1);
  // dSs = dS + dS^T in place (rows / columns >= P are zero because D is zero there): thread e owns the pair (i, j), i < j
  for (int e = tid; e < LR * LR; e += NTHREADS) {
    const int i = e / LR, j = e - i * LR;
    if (i < j) {
      const float s2 = sm.DM[i * LS + j] + sm.DM[j * LS + i];
      sm.DM[i * LS + j] = s2; sm.DM[j * LS + i] = s2;
    } else if (i == j) {
      sm.DM[i * LS + i] = 2.f * sm.DM[i * LS + i];
    }
  }
  STAMPN(16);
  wait_vm<0>();                                      // this wave's rows of x have landed (and its dv stores have left)
  __syncthreads();                                   // ... everybody's; dSs complete; every wave is done reading dv from T1
  STAMPN(17);

  // ---- dX_hat = x_hat dSs ; dx = (dX_hat - x_hat s) / n_x ----
  {
    f32x4 Tx[MT][NTM];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NTM; ++nn) Tx[m][nn] = zero4;
    gemm_cp<MT, NTM>(Tx, KS,
                     [&](int m, int ks) { return a_at(T0, m, ks); },
                     [&](int ks) { return sm.rx[4 * ks + lq]; },
                     [&](int nn, int ks) { return sm.DM[(4 * ks + lq) * LS + 16 * nn + lr]; });
    float Te = edge_column(T0, lane < P ? sm.rx[lane] * sm.DM[lane * LS + (P - 1)] : 0.f, sm.part0 + wave * 64);
    const float xe = T0[tix(P - 1, erow)];
    float xes = elive ? xe * Te * sm.rx[P - 1] : 0.f;
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
    col_reduce_edge<NTM>(v, xes, sm.part1);          // barrier inside: every wave is done reading x from T0
    STAMPN(19);
    if (gx_next) tile_dma(gx_next, T1);              // the next item's x travels under this item's tail
#pragma unroll
    for (int nn = 0; nn < NTM; ++nn) {
      const float rxj = sm.rx[16 * nn + lr], cj = rxj * rxj * v[nn];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        *reinterpret_cast<float4*>(T0 + c_at(m, nn)) =
            make_float4(__builtin_fmaf(-xh[m][nn].x, cj, Tx[m][nn][0] * rxj), __builtin_fmaf(-xh[m][nn].y, cj, Tx[m][nn][1] * rxj),
                        __builtin_fmaf(-xh[m][nn].z, cj, Tx[m][nn][2] * rxj), __builtin_fmaf(-xh[m][nn].w, cj, Tx[m][nn][3] * rxj));
    }
    if (elive) {
      const float rxe = sm.rx[P - 1];
      T0[tix(P - 1, erow)] = __builtin_fmaf(-xe, rxe * rxe * xes, Te * rxe);
    }
  }
  __syncthreads();
  STAMPN(20);
  tile_store(dd.dx + (size_t)n * tile, T0);
  STAMPN(21);
}

__global__ void __launch_bounds__(NTHREADS)
ppm_nhwc_kernel(pixpro_problem pb, Dirs D, float* __restrict__ loss_n, float* __restrict__ pos_num,
                float* __restrict__ pos_mean, int nitems) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const Smem sm = carve(smem_raw);
  // kernel-argument segment: pb at offset 0, D right behind it (both 8-byte aligned)
  const char __attribute__((address_space(4))) * ka =
      (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
  static_assert(sizeof(pixpro_problem) % 8 == 0, "D follows pb without padding");
  const pb_kernarg_t pbp = (pb_kernarg_t)ka;
  const dirs_kernarg_t Dp = (dirs_kernarg_t)(ka + sizeof(pixpro_problem));
  const int tid = threadIdx.x;
  for (int e = tid; e < (TROWS - P) * TS; e += NTHREADS) { sm.T[P * TS + e] = 0.f; sm.T[TILE + P * TS + e] = 0.f; }
  int b = blockIdx.x;
  if (b >= nitems) return;
  tile_dma(D.d[b / pb.N].x + (size_t)(b % pb.N) * (C * P), sm.T);
  int flip = 0;
#pragma clang loop unroll(disable)
  for (; b < nitems; b += gridDim.x) {
    nhwc_item(pbp, Dp, loss_n, pos_num, pos_mean, b, nitems, flip);
    flip ^= 1;
  }
}

}  // namespace

bool fused_nhwc_supported(const pixpro_problem& pb) {
  return pb.layout == 1 && pb.H * pb.W == P && pb.C == C && !getenv("PIXPRO_NO_NHWC_KERNEL");
}

int run_fused_nhwc(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                   hipStream_t st) {
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
  hipLaunchKernelGGL(ppm_nhwc_kernel, dim3(grid), dim3(NTHREADS), bytes, st, pb, D, loss_n, pos_num, pos_mean, items);
  return check_launch("ppm_nhwc_kernel");
}

}  // namespace pixpro
