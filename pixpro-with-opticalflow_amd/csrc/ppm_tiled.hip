// Tiled multi-launch PPM -> flow-matched pixel loss -> backward for grids too large for the single-launch kernel
// (P = H*W > 64, e.g. the 14 x 14 grid of 448^2 crops: one [256][196] fp32 tile is 200 KB, one P x P matrix 154 KB, against
// 160 KB of LDS).  Five launches instead of the ~22 of ppm_generic.hip; the P x P matrices live in the caller's
// workspace (L2 / Infinity Cache resident at these sizes), everything between two matrix products is an epilogue:
//
//   k_tl_pre      1 / max(||.||, 1e-12) of the columns of x, v, k (PixPro.py:344,348,403)  |  sample points (flow-displaced
//                 bin centres), positive-pair BIT rows, counts (:132-223) - independent work-groups of one launch
//   k_tl_gram<0>  S = x^T x on 64 x 64 tiles (K = C), epilogue: normalise, clamp/pow -> A, dA/dS       :354-358
//   k_tl_strip<0> per strip of columns, all C rows in one work-group: Y = v^ A, column norms, Y^ (:361,380);
//                 G = k^ M^T with M expanded from the bit rows (:239); <Y^,G> -> loss partials (:241); dY
//   k_tl_gram<1>  dA = dY^T v^ on tile PAIRS (i,j),(j,i) in one sweep, epilogue: dS = dA o D, dSs = dS + dS^T; its first
//                 work-group also reduces the loss (:241,247)
//   k_tl_strip<1> dV^ = dY A -> dv ; dX^ = x^ dSs -> dx   (one launch, the item index selects)
//   (k_tl_finish  loss_n, loss: a launch of its own only in forward-only calls)
//
// All products run on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32.  Round 3: operands arrive by LDS-DMA
// (global_load_lds_dwordx4) in rings of three slabs with one barrier per slab; both operands of the strip products are
// k-contiguous (the P x P matrices are symmetric up to a column scaling the producer applies) and read as ds_read_b128
// fragments; the 16-column tiles are dealt evenly to the strips; every epilogue moves its global traffic in 256-B rows
// through LDS; the work-groups of one (direction, sample) are placed on one XCD.  Same math and tolerances as the other
// two paths; the bit-exact part (points, threshold, mask) uses the same helpers as ppm_fused.hip.
#include <type_traits>

#include "common.h"

namespace pixpro {
namespace {

#ifndef TL_ABL
#define TL_ABL 0                 // timing ablations (tools/tl_abl.sh): strips: 1 strip<1> epilogue traffic off, 2 no MFMA, 4 no copies;
                                 // Gram kernels: 8 no epilogue stores, 16 no MFMA, 32 no copies after the first two slabs;
                                 // 64: strip operands re-read from the first slab (fetch served from cache)
#endif
constexpr int TB = 256;          // threads per work-group: 4 waves

struct TlWs {                    // workspace carve (floats unless noted)
  float *rx, *rv, *rk;           // [B][P] reciprocal column norms
  float *cnt, *lpart;            // [B][PREP_S] partial counts, [B][strips]
  float *A, *Dm, *dSs, *Av;      // [B][P][P]; Av[n][k] = rv[k] A[n][k], dSs[n][k] = rx[k] (dS + dS^T)[n][k]: the scales of
                                 // the strip products sit on the COLUMNS, row n of the stored matrix is operand column n
  float *dY;                     // [B][C][P]
  unsigned long long* bits;      // [B][P][W64]
  float* zero;                   // 16 floats of zeros (written by k_tl_pre): what the copies fetch for k beyond P
  float *xT, *vT, *kT;           // channels_last inputs (layout 1): [B][C][P] images written by k_tl_pre, else null
};

inline size_t al256(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

size_t tl_bytes(const pixpro_problem& pb) {
  const size_t B = (size_t)pb.N * pb.ndir, P = (size_t)pb.H * pb.W, R = (size_t)pb.C * P, Q = P * P, W64 = (P + 63) / 64;
  const size_t strips = (P + 63) / 64;
  return 3 * al256(B * P * 4) + al256(B * 4 * 4) + al256(B * strips * 4) + 4 * al256(B * Q * 4) + al256(B * R * 4) +
         al256(B * P * W64 * 8) + 256 + 4096 + (pb.layout == 1 ? 3 * al256(B * R * 4) : 0);
}

TlWs tl_carve(void* ws, const pixpro_problem& pb) {
  const size_t B = (size_t)pb.N * pb.ndir, P = (size_t)pb.H * pb.W, R = (size_t)pb.C * P, Q = P * P, W64 = (P + 63) / 64;
  const size_t strips = (P + 63) / 64;
  char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](size_t bytes) { char* r = p; p += al256(bytes); return r; };
  TlWs w;
  w.rx = (float*)take(B * P * 4); w.rv = (float*)take(B * P * 4); w.rk = (float*)take(B * P * 4);
  w.cnt = (float*)take(B * 4 * 4); w.lpart = (float*)take(B * strips * 4);   // 4 = PREP_S
  w.A = (float*)take(B * Q * 4); w.Dm = (float*)take(B * Q * 4); w.dSs = (float*)take(B * Q * 4);
  w.Av = (float*)take(B * Q * 4);
  w.dY = (float*)take(B * R * 4);
  w.bits = (unsigned long long*)take(B * P * W64 * 8);
  w.zero = (float*)take(64);
  w.xT = w.vT = w.kT = nullptr;
  if (pb.layout == 1) {
    w.xT = (float*)take(B * R * 4); w.vT = (float*)take(B * R * 4); w.kT = (float*)take(B * R * 4);
  }
  return w;
}

// ---- positive-pair threshold: identical to ppm_fused.hip (sqrtf(d2)/diag < r  <=>  d2 < T) -------------------------
__device__ __forceinline__ float tl_dist_threshold(float diag, float r, int lane) {
  if (!(r == r) || !(diag > 0.f)) return 0.f;
  long long lo = -1, hi = 0x7f800000LL + 1;
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

constexpr int TL_MAXP = 1024, PREP_S = 4;
// positive pairs of (direction, sample) b: the PREP_S partial counts of k_tl_pre (integers below 2^24: exact in any order)
__device__ __forceinline__ float cnt_of(const float* __restrict__ cnt, int b) {
  float c = 0.f;
#pragma unroll
  for (int k = 0; k < PREP_S; ++k) c += cnt[b * PREP_S + k];
  return c;
}

// ---- k_tl_pre: ONE launch for what depends on the inputs alone (round 2: two launches, 19 + 15 us; the norms are a
// stream, the sample points a latency chain of flow taps - they overlap well) ----------------------------------------------
//   work-groups [0, 6 B):        1 / max(||.||, 1e-12) of the columns of x, v, k: one work-group per (pair, tensor, column
//                                half) - 16 B per lane along the rows (round 2 read 4 B per lane in 64-column strips,
//                                a quarter of them holding 4 of 196 columns), three equal streams per CU
//   work-groups [6 B, 6 B + PREP_S B):  sample points (flow-displaced bin centres), positive-pair BIT rows, counts; every
//                                work-group of a pair takes every PREP_S-th group of 4 rows
__global__ void __launch_bounds__(TB) k_tl_pre(pixpro_problem pb, Dirs D, TlWs w) {
  constexpr int PXB = 112, TRS = PXB + 1;       // channels_last transposition: pixels per block; odd row stride
  constexpr int PTS = 5 * TL_MAXP > 64 * TRS + PXB ? 5 * TL_MAXP : 64 * TRS + PXB;
  __shared__ __attribute__((aligned(16))) float pts[PTS];
  float *qx = pts, *qy = pts + TL_MAXP, *kx = pts + 2 * TL_MAXP, *ky = pts + 3 * TL_MAXP;
  int* valid = reinterpret_cast<int*>(pts + 4 * TL_MAXP);
  __shared__ float sT;
  __shared__ int wcnt[TB / 64];
  const int B = pb.N * pb.ndir, P = pb.H * pb.W, W64 = (P + 63) / 64, C = pb.C;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  if ((int)blockIdx.x < 6 * B) {
    // ---- column norms
    const int q = blockIdx.x, half = q & 1, which = (q >> 1) % 3, b = q / 6;
    if (q == 0 && t < 16) w.zero[t] = 0.f;
    const pixpro_dir& dd = D.d[b / pb.N];
    const float* s = (which == 0 ? dd.x : which == 1 ? dd.v : dd.k) + (size_t)(b % pb.N) * C * P;
    float* o = (which == 0 ? w.rx : which == 1 ? w.rv : w.rk) + (size_t)b * P;
    const int f4 = P / 4, h0 = (f4 + 1) / 2, c0 = half ? h0 : 0, fw = half ? f4 - h0 : h0;   // this half's float4 columns
    if (fw == 0) return;
    if (pb.layout == 1) {
      // channels_last input: the sample is [P][C].  This work-group owns the pixels [4 c0, 4 (c0 + fw)) with ALL channels:
      // the norms are sums over one contiguous 1-KB row per pixel, and since every element passes through here anyway the
      // [C][P] image the products read is written on the way (transposed through LDS, 64 channels x up to 112 pixels at a
      // time; rows of the image leave as 16 B per lane) - no layout copies on the host side.
      float* img = (which == 0 ? w.xT : which == 1 ? w.vT : w.kT) + (size_t)b * C * P;
      // 64 channels x up to PXB = 112 pixels at a time (a half of the 14 x 14 grid, 98 pixels, in one block: rows of the
      // image leave as 392 contiguous bytes; round 4's first version used 64-pixel blocks: 44 us for the launch at N = 64)
      float* tr = pts;                                        // the point arrays are not in use in these work-groups
      float* nrm = tr + 64 * TRS;                             // [PXB] sums of squares
      const int p0 = 4 * c0, npx = 4 * fw, pr = t >> 4, l4 = t & 15;
      const bool unit = which == 2 && !pb.normalize_k;
      for (int pb0 = 0; pb0 < npx; pb0 += PXB) {
        const int nb = min(PXB, npx - pb0);
        for (int j = t; j < nb; j += TB) nrm[j] = 0.f;
        for (int cc = 0; cc < C; cc += 64) {
          __syncthreads();                                    // tr free again (and nrm zeroed)
          for (int pp = pr; pp < nb; pp += TB / 16) {
            const float4 v = *reinterpret_cast<const float4*>(s + (size_t)(p0 + pb0 + pp) * C + cc + 4 * l4);
            tr[(4 * l4 + 0) * TRS + pp] = v.x; tr[(4 * l4 + 1) * TRS + pp] = v.y;
            tr[(4 * l4 + 2) * TRS + pp] = v.z; tr[(4 * l4 + 3) * TRS + pp] = v.w;
            float q = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, __builtin_fmaf(v.z, v.z, v.w * v.w)));
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 8, 64);
            if (l4 == 0) nrm[pp] += q;                         // the same thread owns pixel pp in every channel block
          }
          __syncthreads();
          const int n4 = nb / 4;                               // P % 4 == 0: whole float4s
          for (int e = t; e < 64 * n4; e += TB) {
            const int c = e / n4, j4 = e - c * n4;
            const float* r = tr + c * TRS + 4 * j4;
            *reinterpret_cast<float4*>(img + (size_t)(cc + c) * P + p0 + pb0 + 4 * j4) = make_float4(r[0], r[1], r[2], r[3]);
          }
        }
        __syncthreads();
        for (int j = t; j < nb; j += TB) o[p0 + pb0 + j] = unit ? 1.f : 1.f / fmaxf(sqrtf(nrm[j]), 1e-12f);
        __syncthreads();
      }
      return;
    }
    if (which == 2 && !pb.normalize_k) {
      for (int j = t; j < 4 * fw; j += TB) o[4 * c0 + j] = 1.f;
      return;
    }
    const int rp = TB / fw, r = t / fw, c4 = t - r * fw;             // rows per pass; this thread's row and column
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rp) {
      const float* col = s + 4 * (c0 + c4);
      int row = r;
      for (; row + 3 * rp < C; row += 4 * rp) {                      // four rows in flight
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(col + (size_t)(row + u * rp) * P);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a.x = __builtin_fmaf(v[u].x, v[u].x, a.x); a.y = __builtin_fmaf(v[u].y, v[u].y, a.y);
          a.z = __builtin_fmaf(v[u].z, v[u].z, a.z); a.w = __builtin_fmaf(v[u].w, v[u].w, a.w);
        }
      }
      for (; row < C; row += rp) {
        const float4 v = *reinterpret_cast<const float4*>(col + (size_t)row * P);
        a.x = __builtin_fmaf(v.x, v.x, a.x); a.y = __builtin_fmaf(v.y, v.y, a.y);
        a.z = __builtin_fmaf(v.z, v.z, a.z); a.w = __builtin_fmaf(v.w, v.w, a.w);
      }
      *reinterpret_cast<float4*>(&qx[4 * t]) = a;                    // qx as scratch: [row group][column] partial sums
    }
    __syncthreads();
    for (int j = t; j < 4 * fw; j += TB) {
      float sum = 0.f;
      for (int k = 0; k < rp; ++k) sum += qx[4 * fw * k + j];        // fixed order
      o[4 * c0 + j] = 1.f / fmaxf(sqrtf(sum), 1e-12f);
    }
    return;
  }
  // ---- sample points, bit rows, counts
  const int pq = blockIdx.x - 6 * B, b = pq / PREP_S, part = pq - b * PREP_S;
  const int d = b / pb.N, n = b % pb.N;
  const int row0 = 4 * part + wave, rstep = 4 * PREP_S;              // this wave's rows: row0, row0 + rstep, ...
  const pixpro_dir& dd = D.d[d];
  int Ho, Wo;
  frame_size(pb, dd, Ho, Wo);
  const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
  const Crop cq = crop_of(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
  const Crop ck = crop_of(dd.coord_k + (size_t)n * 10, pb.H, pb.W);
  for (int i = t; i < P; i += TB) {
    const int row = i / pb.W, col = i % pb.W;
    kx[i] = centre_x(ck, col, wo1); ky[i] = centre_y(ck, row, ho1);
    if ((i >> 2) % PREP_S != part) continue;                         // q points of the rows this work-group owns
    float x = centre_x(cq, col, wo1), y = centre_y(cq, row, ho1);
    bool v = true;
    if (dd.lo_own) {            // lazy flow mode (common.h)
      const FlowGeom fg = flow_geom(8 * pb.lo_h, 8 * pb.lo_w, Ho, Wo);
      float ox, oy;
      displace_point_lazy(lazy_src(pb, dd, n), fg, x, y, ox, oy, v);
      x = ox; y = oy;
    } else if (dd.flow) {
      const FlowGeom fg = flow_geom(pb.Hf, pb.Wf, Ho, Wo);
      float ox, oy;
      displace_point(dd.flow + (size_t)n * 2 * pb.Hf * pb.Wf, dd.mask ? dd.mask + (size_t)n * pb.Hf * pb.Wf : nullptr, fg,
                     x, y, ox, oy, v);
      x = ox; y = oy;
    }
    qx[i] = x; qy[i] = y; valid[i] = v ? 1 : 0;
  }
  if (wave == 0) {
    const float T = tl_dist_threshold(fmaxf(bin_diag(cq, wo1, ho1), bin_diag(ck, wo1, ho1)), pb.pos_ratio, lane);
    if (lane == 0) sT = T;
  }
  __syncthreads();
  const float T = sT;
  int c = 0;
  for (int i = row0; i < P; i += rstep) {
    const float xi = qx[i], yi = qy[i];
    const bool vi = valid[i] != 0;
    for (int wd = 0; wd < W64; ++wd) {
      const int j = 64 * wd + lane;
      bool pred = false;
      if (j < P) {
        const float dx = xi - kx[j], dy = yi - ky[j];
        pred = (dx * dx + dy * dy < T) && vi;
      }
      const unsigned long long m = __ballot(pred);
      if (lane == 0) w.bits[((size_t)b * P + i) * W64 + wd] = m;
      c += __popcll(m);
    }
  }
  if (lane == 0) wcnt[wave] = c;
  __syncthreads();
  if (t == 0) {
    int tot = 0;
    for (int k = 0; k < TB / 64; ++k) tot += wcnt[k];
    w.cnt[b * PREP_S + part] = (float)tot;
  }
}

// ---- XCD-aware placement ----------------------------------------------------------------------------------------------
// Consecutive work-groups of a launch go to the eight XCDs in turn, each with its own 4 MB L2.  The `per` work-groups that
// share one (direction, sample) - the ten tile pairs of a Gram product, the strips of a strip product - get linear ids 8
// apart: they run on ONE XCD at about the same time, the tiles they share are fetched into that L2 once, and what one
// launch writes for a sample (A, Av, dY, dSs) is still in the L2 of the XCD that reads it in the next launch.
// (Placement only changes speed; nothing depends on it.)  Launches are 1-D: B * per work-groups.
__device__ __forceinline__ void xcd_item(int per, int B, int& b, int& item) {
  const int L = blockIdx.x, full = (B / 8) * 8 * per;
  if (L < full) {
    const int slot = L >> 3;
    b = 8 * (slot / per) + (L & 7);
    item = slot % per;
  } else {
    const int r = L - full;
    b = (B / 8) * 8 + r / per;
    item = r % per;
  }
}

// (dma16, wait_vm, lds_barrier: asynchronous global -> LDS copies, common.h)

// ---- P x P products over K = C on 64 x 64 tiles --------------------------------------------------------------------------
// NP 64 x 64 x C products at once: acc[q][i][j] += sum_c L_q[c*P + m0_q + ..] * R_q[c*P + n0_q + ..]; wave w owns the
// 32 x 32 block (w >> 1, w & 1) of every product.  Round 3: the slabs arrive by LDS-DMA, GNB - 1 slabs ahead of the MFMAs in
// a ring of GNB buffers, one barrier per slab (round 2: 16-deep slabs through registers, two barriers each - the kernel
// waited for memory latency 16 times per tile with 16 MFMAs per wave in between).  One product runs on 32-deep slabs, the
// pair of the backward pass on 16-deep ones: four copies per wave and slab and the same LDS either way.
// Slab image [k][64], the 16-B chunks of odd k rows swapped in pairs 16 floats apart (column ^ 16): the MFMA operand reads
// (lanes 0-31 of a ds_read_b32 = two k rows x 16 columns) are conflict-free.  Columns beyond P fetch the row's last valid
// chunk instead (finite duplicates; their products land in rows / columns nobody stores).
#ifndef TL_GNB
#define TL_GNB 2
#endif
// Ring length of the Gram products.  Two buffers (the copy of slab s + 1 travels under the MFMAs of slab s) leave room for
// four work-groups per CU next to the epilogue tiles; three buffers (two slabs ahead) for three - measured at N = 64, 14 x 14:
// 41 + 61 us against 44 + 71 us: these launches hide latency with neighbours, not with depth.
constexpr int GNB = TL_GNB;
constexpr int EPS = 68;                         // row stride of the epilogue tiles (16-B aligned rows, conflict-free writes)
constexpr int GRING = GNB * 2 * 32 * 64 > 2 * 64 * EPS ? GNB * 2 * 32 * 64 : 2 * 64 * EPS;   // floats: 34 KB at GNB = 2
struct GramOp {
  const float* L;
  const float* Rm;
  int m0, n0;
};
// Waves per Gram work-group.  Eight (a 4 x 2 grid of 16 x 32 blocks, one accumulator pair per wave and product) instead
// of four (2 x 2 of 32 x 32): twice the waves per CU to hide the latency of a launch that is all prologues and epilogues.
#ifndef TL_GW
#define TL_GW 8
#endif
constexpr int GW = TL_GW, TBG = 64 * GW, GNI = GW == 8 ? 1 : 2, GNJ = 2;
__device__ __forceinline__ int gram_wm(int wave) { return GW == 8 ? (wave >> 1) * 16 : (wave >> 1) * 32; }
__device__ __forceinline__ int gram_wn(int wave) { return (wave & 1) * 32; }
template <int NP>
__device__ __forceinline__ void gram_core(f32x4 (&acc)[NP][GNI][GNJ], const GramOp (&op)[NP], int C, int P, float* ring) {
  constexpr int GKT = 32 / NP, SLAB = GKT * 64;                         // k rows per slab
  constexpr int OPP = GKT / 4;                                          // 1-KB pieces per operand and slab
  constexpr int PIECES = NP * 2 * OPP / GW;                             // copies per wave and slab (2 or 4)
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const int wm = gram_wm(wave), wn = gram_wn(wave);
  // 16 x 16 sub-tiles wholly beyond P (P = 196 = 12.25 sub-tiles per side) issue no MFMA (wave-uniform)
  bool live[NP][GNI][GNJ], full = true, any = false;
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int i = 0; i < GNI; ++i)
#pragma unroll
      for (int j = 0; j < GNJ; ++j) {
        live[q][i][j] = (op[q].m0 + wm + 16 * i < P) && (op[q].n0 + wn + 16 * j < P);
        full = full && live[q][i][j];
        any = any || live[q][i][j];
      }
  // The slab's NP * 2 operands x OPP pieces form one list (the buffer's layout: operand after operand, 256 floats per
  // piece); this wave copies pieces PIECES * wave .. + PIECES - 1 of it.  Piece f: operand f / OPP (product f / (2 OPP), L or
  // R by the parity), k rows 4 (f % OPP) .. + 3.
  const float* src[PIECES];
#pragma unroll
  for (int e = 0; e < PIECES; ++e) {
    const int f = PIECES * wave + e, o = f / OPP, kk = 4 * (f % OPP) + lq, ch = lr ^ ((kk & 1) << 2);
    // Every product of a launch reads the same two tensors (only the tile origins differ), and the origin is picked by
    // arithmetic: indexing `op` with the run-time product number - or selecting its fields - put the array into scratch.
    const int q1 = NP > 1 ? (o >> 1) : 0;
    const int m0s = op[0].m0 + q1 * (op[NP - 1].m0 - op[0].m0), n0s = op[0].n0 + q1 * (op[NP - 1].n0 - op[0].n0);
    src[e] = ((o & 1) ? op[0].Rm : op[0].L) + (size_t)kk * P + min(((o & 1) ? n0s : m0s) + 4 * ch, P - 4);
  }
  const size_t slab_step = (size_t)GKT * P;
  auto issue = [&](int s) {
    float* buf = ring + (s % GNB) * NP * 2 * SLAB + 256 * PIECES * wave;
#pragma unroll
    for (int e = 0; e < PIECES; ++e) dma16(src[e] + s * slab_step, buf + 256 * e);
  };
  constexpr int PER = PIECES;                   // copies per wave and slab
  const int S = C / GKT;
  issue(0);
  if (S > 1 && GNB > 2) issue(1);
  const int sw = 16 * (lq & 1);
  for (int s = 0; s < S; ++s) {
    if ((TL_ABL & 32) || GNB == 2) wait_vm<0>();
    else if (s + 1 < S) wait_vm<PER>(); else wait_vm<0>();   // this wave's pieces of slab s have landed
    lds_barrier();                                      // ... everybody's; and everybody is done reading slab s - 1
    if (s + GNB - 1 < S && !(TL_ABL & 32)) issue(s + GNB - 1);   // into the buffer slab s - 1 occupied
    const float* buf = ring + (s % GNB) * NP * 2 * SLAB;
    if (TL_ABL & 16) {
#pragma unroll
      for (int q = 0; q < NP; ++q) acc[q][0][0][0] += buf[(2 * q) * SLAB + lane + 64 * (s & 7)] + buf[(2 * q + 1) * SLAB + lane];
    } else if (full) {                                         // the common case: straight-line code
#pragma unroll
      for (int ks = 0; ks < GKT / 4; ++ks)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          float av[GNI], bv[GNJ];
#pragma unroll
          for (int i = 0; i < GNI; ++i) av[i] = buf[(2 * q) * SLAB + (4 * ks + lq) * 64 + ((wm + 16 * i + lr) ^ sw)];
#pragma unroll
          for (int j = 0; j < GNJ; ++j) bv[j] = buf[(2 * q + 1) * SLAB + (4 * ks + lq) * 64 + ((wn + 16 * j + lr) ^ sw)];
#pragma unroll
          for (int i = 0; i < GNI; ++i)
#pragma unroll
            for (int j = 0; j < GNJ; ++j) acc[q][i][j] = mfma16(av[i], bv[j], acc[q][i][j]);
        }
    } else if (any) {
#pragma unroll
      for (int ks = 0; ks < GKT / 4; ++ks)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          float av[GNI], bv[GNJ];
#pragma unroll
          for (int i = 0; i < GNI; ++i) av[i] = buf[(2 * q) * SLAB + (4 * ks + lq) * 64 + ((wm + 16 * i + lr) ^ sw)];
#pragma unroll
          for (int j = 0; j < GNJ; ++j) bv[j] = buf[(2 * q + 1) * SLAB + (4 * ks + lq) * 64 + ((wn + 16 * j + lr) ^ sw)];
#pragma unroll
          for (int i = 0; i < GNI; ++i)
#pragma unroll
            for (int j = 0; j < GNJ; ++j)
              if (live[q][i][j]) acc[q][i][j] = mfma16(av[i], bv[j], acc[q][i][j]);
        }
    }
  }
  lds_barrier();                                        // the ring may be reused (epilogue tiles)
}

// MODE 0: S -> A, D on the tiles ti <= tj, mirrored (grid: pairs, 1, B).  MODE 1: dSs on tile pairs (grid: pairs, 1, B).
// loss_n[b] = sum of the strip partials / (cnt + 1e-6) (PixPro.py:241); loss = -2 sum_d mean_n loss_n (:247,431); one block
__device__ __forceinline__ void tl_finish(const float* __restrict__ lpart, const float* __restrict__ cnt,
                                          float* __restrict__ loss_n, float* __restrict__ loss, float* __restrict__ pos_num,
                                          float* __restrict__ pos_mean, int B, int N, int strips, int P, float* red) {
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < strips; ++k) s += lpart[(size_t)b * strips + k];
    const float cb = cnt_of(cnt, b);
    const float ln = s / (cb + 1e-6f);
    loss_n[b] = ln;
    pos_num[b] = cb;
    pos_mean[b] = cb / (float)P / (float)P;         // reference: mean(-1).mean(-1) in fp32 (< 1 ulp apart)
    acc += ln;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tsum = 0.f;
    for (int k = 0; k < (int)blockDim.x / 64; ++k) tsum += red[k];
    *loss = -2.f * tsum / (float)N;
  }
  __syncthreads();
}
struct TlFin {                   // outputs of the loss reduction (MODE 1 of k_tl_gram does it in its first block)
  float *loss_n, *loss, *pos_num, *pos_mean;
  int B, strips;
};

// Row-major store of a 64 x 64 tile staged in LDS ([64][EPS]): wave w takes rows 16 w .. + 15, 16 lanes per row (256 B
// contiguous, 16 B per lane); rows / 4-column groups beyond P are skipped (P % 4 == 0).
__device__ __forceinline__ void tile_rows_out(const float* tile, float* __restrict__ dst, int r0, int c0, int P) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c4 = lane & 15;
#pragma unroll
  for (int i = 0; i < 16 / GW; ++i) {
    const int row = (64 / GW) * wave + 4 * i + (lane >> 4);
    if (r0 + row < P && c0 + 4 * c4 < P && (!(TL_ABL & 8) || tile[0] == 123.456f))
      *reinterpret_cast<float4*>(dst + (size_t)(r0 + row) * P + c0 + 4 * c4) =
          *reinterpret_cast<const float4*>(tile + row * EPS + 4 * c4);
  }
}

template <int MODE>
__global__ void __launch_bounds__(TBG) k_tl_gram(Dirs D, TlWs w, int N, int C, int P, float p, float clampv, TlFin fin) {
  __shared__ __attribute__((aligned(16))) float ring[GRING];      // slab ring, then the epilogue tiles
  static_assert(GRING >= 2 * 64 * EPS, "two epilogue tiles fit in the slab ring");
  // The loss reduction needs every strip of the launch before (k_tl_strip<0>) and nothing of this one: when gradients
  // are wanted it rides in the first block of this launch instead of a one-block launch of its own.
  if (MODE == 1 && blockIdx.x == 0)
    tl_finish(w.lpart, w.cnt, fin.loss_n, fin.loss, fin.pos_num, fin.pos_mean, fin.B, N, fin.strips, P, ring);
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const int wm = gram_wm(wave), wn = gram_wn(wave);
  const size_t R = (size_t)C * P, Q = (size_t)P * P;
  const int nt = (P + 63) / 64;
  int b, rem;
  xcd_item(nt * (nt + 1) / 2, fin.B, b, rem);
  const pixpro_dir& dd = D.d[b / N];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // tile pair index -> (ti <= tj)
  int ti = 0;
  while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
  const int tj = ti + rem, i0 = ti * 64, j0 = tj * 64;
  float* T0 = ring;                 // tile (ti, tj), [row][EPS]
  float* T1 = ring + 64 * EPS;      // tile (tj, ti)
  if (MODE == 0) {
    // S is symmetric: only the tiles (ti <= tj) are computed (10 of 16 at P = 196) and written twice.  Round 3: both
    // images leave through LDS - the accumulator layout (16 columns x 4 rows per lane group) stored 64-B pieces of four
    // rows per instruction, the mirror image 16-B pieces of 64 rows; now every store instruction writes four 256-B rows.
    const float* x = w.xT ? w.xT + (size_t)b * R : dd.x + (size_t)(b % N) * R;     // channels_last: k_tl_pre's [C][P] image
    f32x4 acc[1][GNI][GNJ];
#pragma unroll
    for (int i = 0; i < GNI; ++i)
#pragma unroll
      for (int j = 0; j < GNJ; ++j) acc[0][i][j] = z;
    const GramOp op[1] = {{x, x, i0, j0}};
    gram_core<1>(acc, op, C, P, ring);
    const float* rx = w.rx + (size_t)b * P;
    const float* rv = w.rv + (size_t)b * P;
    float a[GNI][GNJ][4], d[GNI][GNJ][4], rvm[GNI][4], rvn[GNJ];
#pragma unroll
    for (int j = 0; j < GNJ; ++j) rvn[j] = rv[min(j0 + wn + 16 * j + lr, P - 1)];
#pragma unroll
    for (int i = 0; i < GNI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) rvm[i][r] = rv[min(i0 + wm + 16 * i + 4 * lq + r, P - 1)];
#pragma unroll
    for (int i = 0; i < GNI; ++i)
#pragma unroll
      for (int j = 0; j < GNJ; ++j) {
        const float rxn = rx[min(j0 + wn + 16 * j + lr, P - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pow_map(acc[0][i][j][r] * rx[min(i0 + wm + 16 * i + 4 * lq + r, P - 1)] * rxn, p, clampv, a[i][j][r], d[i][j][r]);
      }
    // which = 0: A;  1: D = dA/dS;  2: Av[n][k] = rv[k] A[n][k] (the 1/||v|| of Y = v^ A on the COLUMNS: row n of the
    // stored matrix is column n of the operand of the strip product)
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
      for (int i = 0; i < GNI; ++i)
#pragma unroll
        for (int j = 0; j < GNJ; ++j) {
          float v[4], m[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = which == 0 ? a[i][j][r] : which == 1 ? d[i][j][r] : rvn[j] * a[i][j][r];
            m[r] = which == 0 ? a[i][j][r] : which == 1 ? d[i][j][r] : rvm[i][r] * a[i][j][r];
            T0[(wm + 16 * i + 4 * lq + r) * EPS + wn + 16 * j + lr] = v[r];
          }
          if (ti != tj)      // mirror image: four consecutive rows of the accumulator = four consecutive columns there
            *reinterpret_cast<float4*>(T1 + (wn + 16 * j + lr) * EPS + wm + 16 * i + 4 * lq) = make_float4(m[0], m[1], m[2], m[3]);
        }
      lds_barrier();
      float* dst = (which == 0 ? w.A : which == 1 ? w.Dm : w.Av) + (size_t)b * Q;
      tile_rows_out(T0, dst, i0, j0, P);
      if (ti != tj) tile_rows_out(T1, dst, j0, i0, P);
      lds_barrier();
    }
  } else {
    // dS1[i][j] = rv[j] * sum_c dY[c][i] v[c][j] * D[i][j] on tile (ti, tj), dS2 likewise on tile (tj, ti): both products in
    // ONE pass over C (round 2 ran them one after the other: two pipeline fills, a third of the occupancy for the partner
    // tile kept in LDS meanwhile); the diagonal tiles need one product only.
    const float* dY = w.dY + (size_t)b * R;
    const float* v = w.vT ? w.vT + (size_t)b * R : dd.v + (size_t)(b % N) * R;
    const float* rv = w.rv + (size_t)b * P;
    const float* Dm = w.Dm + (size_t)b * Q;
    f32x4 acc[2][GNI][GNJ];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < GNI; ++i)
#pragma unroll
        for (int j = 0; j < GNJ; ++j) acc[q][i][j] = z;
    if (ti != tj) {
      const GramOp op[2] = {{dY, v, i0, j0}, {dY, v, j0, i0}};
      gram_core<2>(acc, op, C, P, ring);
    } else {
      f32x4 one[1][GNI][GNJ];
#pragma unroll
      for (int i = 0; i < GNI; ++i)
#pragma unroll
        for (int j = 0; j < GNJ; ++j) one[0][i][j] = z;
      const GramOp op[1] = {{dY, v, i0, j0}};
      gram_core<1>(one, op, C, P, ring);
#pragma unroll
      for (int i = 0; i < GNI; ++i)
#pragma unroll
        for (int j = 0; j < GNJ; ++j) acc[0][i][j] = one[0][i][j];
    }
    // dSs[i][j] = dS[i][j] + dS[j][i] with dS = rv[col] dA o D and D symmetric:  dSs[i][j] = D[i][j] (U1[i][j] + U2[j][i]),
    // U = rv[col] * dA staged in LDS per tile.  D is read row-major where the result is written (256-B rows); round 2
    // multiplied by D in the accumulator layout: 32 loads of 64-B pieces of four rows per lane.
    // (two straight-line calls: a loop with `break` over `half` kept acc[half] dynamically indexed - 64 B of scratch per
    // thread in the only kernel of this file that had any)
    auto stage_u = [&](auto hc) {
      constexpr int half = decltype(hc)::value;
      const int n0 = half ? i0 : j0;
      float* Lt = half ? T1 : T0;
#pragma unroll
      for (int i = 0; i < GNI; ++i)
#pragma unroll
        for (int j = 0; j < GNJ; ++j) {
          const int ln = wn + 16 * j + lr;
          const float rvn = rv[min(n0 + ln, P - 1)];
#pragma unroll
          for (int r = 0; r < 4; ++r) Lt[(wm + 16 * i + 4 * lq + r) * EPS + ln] = acc[half][i][j][r] * rvn;
        }
    };
    stage_u(std::integral_constant<int, 0>{});
    if (ti != tj) stage_u(std::integral_constant<int, 1>{});
    lds_barrier();
    const float* T2 = ti == tj ? T0 : T1;
    float* out = w.dSs + (size_t)b * Q;
    const float* rx = w.rx + (size_t)b * P;     // the 1/||x|| of dX^ = x^ dSs folded into the COLUMNS of the stored dSs
    for (int e = t; e < 64 * 64; e += TBG) {
      const int r = e >> 6, c = e & 63;
      if ((TL_ABL & 8) && T0[0] != 123.456f) continue;
      if (i0 + r < P && j0 + c < P) {
        const size_t o = (size_t)(i0 + r) * P + j0 + c;
        out[o] = rx[j0 + c] * (Dm[o] * (T0[r * EPS + c] + T2[c * EPS + r]));
      }
      if (ti != tj && j0 + r < P && i0 + c < P) {
        const size_t o = (size_t)(j0 + r) * P + i0 + c;
        out[o] = rx[i0 + c] * (Dm[o] * (T1[r * EPS + c] + T0[c * EPS + r]));
      }
    }
  }
}

// ---- C x 64 strips over K = P: all C rows of a strip of columns in one work-group -----------------------------------------
// acc[mt][nt] += sum_k Aop(c, k) * Bop(k, n);  wave w owns rows [w*C/4, (w+1)*C/4) = MW 16-row tiles.
//
// Round 3.  BOTH operands are k-contiguous in memory: the A operand is a [C][P] tile (v, k, dY, x); the B operands are the
// P x P matrices, which are symmetric (A, dS + dS^T) up to a scaling the producer (k_tl_gram) applies to the COLUMNS of what
// it stores, so that row n of the stored matrix is column n of the operand.  Slabs of 16 k arrive by LDS-DMA two slabs
// ahead in a ring of three, one barrier per slab, no staging registers and no LDS commit instructions (round 2: registers,
// ds_write, two barriers per slab; the in-loop loads cost a quarter of the kernel).  Slab image [row][16 k]: a lane of the
// MFMA layout (row lr, k group lq) takes its four k-steps' values k = 4 lq .. 4 lq + 3 with ONE ds_read_b128 (the order of
// the contraction index is free as long as both operands agree), 8 LDS reads per 64 MFMAs instead of 32.  The 16-B chunk
// that holds k group q of row r sits at position q ^ g(r), g(r) = (-(r >> 2)) & 3: with that every ds_read_b128 lane group
// ({0-3,12-15,20-27}, ...) touches each bank once.  The expanded mask bits of G = k^ M^T are written in the same image.
constexpr int SK = 16, SNB = 3;
__device__ __forceinline__ int chunk_swz(int row) { return (-(row >> 2)) & 3; }

template <int MW, int NWS>                    // MW 16-row tiles per wave, NWS waves: C = 16 MW NWS rows
struct StripRing {                            // LDS carve of one strip work-group
  static constexpr int A_FLOATS = 16 * MW * NWS * SK, B_FLOATS = 64 * SK, SLAB = A_FLOATS + B_FLOATS;
};

// BMODE 0: B rows by DMA from Bsrc (row n0 + n of a [P][P] matrix);  BMODE 1: B generated by `gen(k0, slab B buffer)`.
template <int MW, int NWS, int BMODE, class Gen>
__device__ __forceinline__ void strip_gemm(f32x4 (&acc)[MW][4], int P, int n0, int nlive, float* ring,
                                           const float* __restrict__ Asrc, const float* __restrict__ Bsrc,
                                           const float* __restrict__ zero16, Gen gen) {
  using R = StripRing<MW, NWS>;
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  // DMA pieces of this wave: A piece e covers rows 16 (NWS e + wave) .. + 15 (lane -> row lane >> 2, position lane & 3);
  // the B slab has four pieces: wave w < 4 copies B rows 16 w .. + 15
  const float* pa[MW];
  int qa[MW];                                  // source k chunk of this lane (position ^ swizzle)
#pragma unroll
  for (int e = 0; e < MW; ++e) {
    const int row = 16 * (NWS * e + wave) + (lane >> 2);
    qa[e] = (lane & 3) ^ chunk_swz(row);
    pa[e] = Asrc + (size_t)row * P + 4 * qa[e];
  }
  const bool hasb = BMODE == 0 && wave < 4;    // uniform
  const int brow = 16 * (wave & 3) + (lane >> 2), qb = (lane & 3) ^ chunk_swz(brow);
  const float* pb = Bsrc + (size_t)min(n0 + brow, P - 1) * P + 4 * qb;   // rows beyond P: finite duplicates, never stored
  const int S = (P + SK - 1) / SK;
  auto issue = [&](int s) {
    float* buf = ring + (s % SNB) * R::SLAB;
    const int k0 = (TL_ABL & 64) ? 0 : s * SK;   // ablation 64: every slab re-reads the first one (cache hits): what does the fetch cost?
    const bool tail = k0 + SK > P;             // uniform: only the last slab of P = 196 has k chunks beyond P
#pragma unroll
    for (int e = 0; e < MW; ++e)
      dma16((tail && k0 + 4 * qa[e] >= P) ? zero16 : pa[e] + k0, buf + 16 * (NWS * e + wave) * SK);
    if (hasb) dma16((tail && k0 + 4 * qb >= P) ? zero16 : pb + k0, buf + R::A_FLOATS + 16 * (wave & 3) * SK);
  };
  issue(0);
  if (BMODE == 1) gen(0, ring + R::A_FLOATS);
  if (S > 1) {
    issue(1);
    if (BMODE == 1) gen(SK, ring + R::SLAB + R::A_FLOATS);
  }
  // fragment addresses inside a slab (floats): A row wave*16MW + 16m + lr, B row 16nn + lr, k group lq
  int fa[MW], fb[4];
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int row = wave * 16 * MW + 16 * m + lr;
    fa[m] = row * SK + 4 * (lq ^ chunk_swz(row));
  }
#pragma unroll
  for (int nn = 0; nn < 4; ++nn) {
    const int row = 16 * nn + lr;
    fb[nn] = R::A_FLOATS + row * SK + 4 * (lq ^ chunk_swz(row));
  }
  for (int s = 0; s < S; ++s) {
    if ((TL_ABL & 4) || s + 1 >= S) wait_vm<0>();        // this wave's copies of slab s have landed ...
    else if (hasb) wait_vm<MW + 1>();                     // ... (one slab's worth of copies may stay in flight:
    else wait_vm<MW>();                                   //      MW pieces of A, and one of B in the waves that copy B)
    lds_barrier();                                      // ... everybody's; and slab s - 1 has been read by everybody
    if (s + 2 < S && !(TL_ABL & 4)) {
      issue(s + 2);
      if (BMODE == 1) gen((s + 2) * SK, ring + ((s + 2) % SNB) * R::SLAB + R::A_FLOATS);
    }
    const float* buf = ring + (s % SNB) * R::SLAB;
    float4 av[MW], bv[4];
#pragma unroll
    for (int m = 0; m < MW; ++m) av[m] = *reinterpret_cast<const float4*>(buf + fa[m]);
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) bv[nn] = *reinterpret_cast<const float4*>(buf + fb[nn]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
        if (nn < nlive) {                // column tiles beyond this strip issue no MFMA (uniform)
#pragma unroll
          for (int m = 0; m < MW; ++m) {
            const float a = ks == 0 ? av[m].x : ks == 1 ? av[m].y : ks == 2 ? av[m].z : av[m].w;
            const float b = ks == 0 ? bv[nn].x : ks == 1 ? bv[nn].y : ks == 2 ? bv[nn].z : bv[nn].w;
            if (TL_ABL & 2) acc[m][nn][0] += a + b;
            else acc[m][nn] = mfma16(a, b, acc[m][nn]);
          }
        }
  }
  lds_barrier();                                        // the ring may be refilled by the next product
}

// column sums over all C rows of v[nn] (this lane's columns 16*nn + lr): shuffles over lq, one LDS hop over the NWS waves
template <int NWS>
__device__ __forceinline__ void strip_colsum(float (&v)[4], float* part) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15;
#pragma unroll
  for (int nn = 0; nn < 4; ++nn) {
    float s = v[nn];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) part[wave * 64 + 16 * nn + lane] = s;
  }
  __syncthreads();
#pragma unroll
  for (int nn = 0; nn < 4; ++nn) {
    float s = part[16 * nn + lr];
#pragma unroll
    for (int k = 1; k < NWS; ++k) s += part[64 * k + 16 * nn + lr];
    v[nn] = s;
  }
  __syncthreads();
}

// MODE 0: forward + loss + dY (grid: B * strips).  MODE 1: dv / dx (grid: B * 2 strips; the item index selects).
// MW 16-row tiles per wave, NWS waves per work-group (C = 16 MW NWS): 256 channels run as 8 waves x 32 rows - twice the
// waves per CU of the 4 x 64 split (two work-groups per CU either way: the LDS ring), each with half the accumulators
template <int MW, int MODE, int NWS>
__global__ void __launch_bounds__(64 * NWS) k_tl_strip(pixpro_problem pb, Dirs D, TlWs w, float g0, int want_grad, int ns) {
  constexpr int C = 16 * MW * NWS, TBS = 64 * NWS;
  __shared__ __attribute__((aligned(16))) float ring[SNB * StripRing<MW, NWS>::SLAB];      // 60 KB at C = 256
  __shared__ float part[NWS * 64];
  __shared__ unsigned long long bl[MODE == 0 ? 64 * (TL_MAXP / 64) : 1];             // the strip's mask bit rows
  __shared__ float rks[MODE == 0 ? TL_MAXP + SK : 1];                                // 1/||k_j||, zero beyond P
  const int N = pb.N, P = pb.H * pb.W, W64 = (P + 63) / 64;
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const size_t R = (size_t)C * P, Q = (size_t)P * P;
  int b, item;                                  // item = strip (+ ns for the dx half of MODE 1)
  xcd_item(MODE == 0 ? ns : 2 * ns, N * pb.ndir, b, item);
  const int sx = item % ns;
  const pixpro_dir& dd = D.d[b / N];
  const size_t so = (size_t)(b % N) * R;
  // The 16-column tiles of P are dealt evenly to the strips, larger strips first (P = 196: 13 tiles as 4 + 3 + 3 + 3;
  // round 2 cut 64-column strips, 4 + 4 + 4 + 1: the CUs holding the one-tile strips idled for most of the launch).
  int n0, nlive;
  {
    const int tt = (P + 15) / 16, base = tt / ns, rem = tt % ns;
    n0 = 16 * (sx * base + min(sx, rem));
    nlive = base + (sx < rem ? 1 : 0);
  }
  const int pend = min(P, n0 + 16 * nlive);     // this strip's columns are [n0, pend): the epilogues touch nothing beyond
  // Global traffic of the epilogues goes through a per-wave staging tile in the (idle) ring: the accumulator layout (16
  // columns x 4 rows per lane group) would load / store 64-B pieces of four rows per instruction - measured at 56 us of
  // the 141 us of k_tl_strip<1>; staged, every instruction moves four rows of up to 256 B, 16 B per lane.  Same wave
  // writes and reads its tile (LDS operations of a wave execute in order): no barrier.
  constexpr int MC = (MW >= 2 && NWS == 4) ? 2 : 1;           // 16-row tiles per staging pass
  static_assert(SNB * StripRing<MW, NWS>::SLAB >= NWS * 16 * MC * EPS, "staging tiles fit in the ring");
  float* scr = ring + wave * (16 * MC * EPS);
  // Lanes of ONE wave exchange values through `scr` (lane A writes, lane B reads another index).  The hardware runs a
  // wave's LDS operations in order; this keeps the compiler from reordering the two per-thread accesses it sees.
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto rows_out = [&](float* __restrict__ dst, auto value) {      // value(m, nn, r) in accumulator layout -> rows of dst
#pragma unroll
    for (int pp = 0; pp < MW / MC; ++pp) {
#pragma unroll
      for (int mm = 0; mm < MC; ++mm)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[(16 * mm + 4 * lq + r) * EPS + 16 * nn + lr] = value(MC * pp + mm, nn, r);
      wave_sync();
#pragma unroll
      for (int i = 0; i < 4 * MC; ++i) {
        const int row = 4 * i + (lane >> 4), c4 = lane & 15;
        if (n0 + 4 * c4 < pend)
          *reinterpret_cast<float4*>(dst + (size_t)(wave * 16 * MW + 16 * MC * pp + row) * P + n0 + 4 * c4) =
              *reinterpret_cast<const float4*>(scr + row * EPS + 4 * c4);
      }
      wave_sync();
    }
  };
  auto rows_in = [&](const float* __restrict__ src, auto sink) {   // rows of src -> sink(m, nn, r, value)
#pragma unroll
    for (int pp = 0; pp < MW / MC; ++pp) {
#pragma unroll
      for (int i = 0; i < 4 * MC; ++i) {
        const int row = 4 * i + (lane >> 4), c4 = lane & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + 4 * c4 < pend)
          v = *reinterpret_cast<const float4*>(src + (size_t)(wave * 16 * MW + 16 * MC * pp + row) * P + n0 + 4 * c4);
        *reinterpret_cast<float4*>(scr + row * EPS + 4 * c4) = v;
      }
      wave_sync();
#pragma unroll
      for (int mm = 0; mm < MC; ++mm)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) sink(MC * pp + mm, nn, r, scr[(16 * mm + 4 * lq + r) * EPS + 16 * nn + lr]);
      wave_sync();
    }
  };
  // channels_last outputs (pred, dv, dx are [P][C]): the wave's 16 MW channels of a pixel are contiguous there (64 / 128 B), so
  // the staging tile is transposed - [32 pixels][16 MW channels + 4] - filled with one 16-B write per accumulator tile and lane
  // (its four registers are four consecutive channels of one pixel) and emptied pixel row by pixel row, 16 B per lane.
  const bool nhwc = pb.layout == 1;
  constexpr int CW = 16 * MW, PXS = CW + 4, LPP = CW / 4;          // channels per wave; staging row stride; lanes per pixel
  static_assert(SNB * StripRing<MW, NWS>::SLAB >= NWS * 32 * PXS, "transposed staging tiles fit in the ring");
  float* scrT = ring + wave * (32 * PXS);
  auto pix_out = [&](float* __restrict__ dst, auto value) {        // value(m, nn, r) in accumulator layout -> pixel rows of dst
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          *reinterpret_cast<float4*>(scrT + (16 * q + lr) * PXS + 16 * m + 4 * lq) =
              make_float4(value(m, 2 * h + q, 0), value(m, 2 * h + q, 1), value(m, 2 * h + q, 2), value(m, 2 * h + q, 3));
      wave_sync();
#pragma unroll
      for (int i = 0; i < 32 * LPP / 64; ++i) {
        const int px = (64 / LPP) * i + lane / LPP, l4 = lane % LPP, col = n0 + 32 * h + px;
        if (col < pend)
          *reinterpret_cast<float4*>(dst + (size_t)col * C + wave * CW + 4 * l4) =
              *reinterpret_cast<const float4*>(scrT + px * PXS + 4 * l4);
      }
      wave_sync();
    }
  };
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[MW][4];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = z;

  if (MODE == 0) {
    const float* v = nhwc ? w.vT + (size_t)b * R : dd.v + so;
    const float* kk = nhwc ? w.kT + (size_t)b * R : dd.k + so;
    const float* rk = w.rk + (size_t)b * P;
    const float* Av = w.Av + (size_t)b * Q;
    // ---- Y = v^ A = v (diag(rv) A): the scale sits in the COLUMNS of the stored matrix AvT[n][k] = rv[k] A[n][k] (A is
    // symmetric), whose row n is column n of the operand: both operands stream k-contiguous, as they are
    // stage what the G product needs while the first copies travel: the strip's bit rows and 1/||k||
    for (int e = t; e < 64 * W64; e += TBS) {
      const int n = e / W64, wd = e - n * W64;
      bl[e] = n0 + n < P ? w.bits[((size_t)b * P + n0 + n) * W64 + wd] : 0ull;
    }
    for (int e = t; e < P + SK; e += TBS) rks[e] = e < P ? rk[e] : 0.f;
    auto none = [](int, float*) {};
    strip_gemm<MW, NWS, 0>(acc, P, n0, nlive, ring, v, Av, w.zero, none);
    float ry[4];
    {
      float s[4];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        float a = 0.f;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) a = __builtin_fmaf(acc[m][nn][r], acc[m][nn][r], a);
        s[nn] = a;
      }
      strip_colsum<NWS>(s, part);
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) ry[nn] = 1.f / fmaxf(sqrtf(s[nn]), 1e-12f);
    }
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][nn][r] = acc[m][nn][r] * ry[nn];
    if (dd.pred) {
      auto val = [&](int m, int nn, int r) { return acc[m][nn][r]; };
      if (nhwc) pix_out(dd.pred + so, val); else rows_out(dd.pred + so, val);
      lds_barrier();             // the staging tiles lie where the next product's first slabs land
    }
    // ---- G = k^ M^T, M expanded from the bit rows: B(k = j, n = i) = bit j of row i
    f32x4 G[MW][4];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) G[m][nn] = z;
    // G = k^ M^T = k (diag(rk) M^T): the column scale of k rides on the expanded mask bits.  Thread (n = t & 63, k group
    // t >> 6) writes the four values of its k group as one 16-B store into the slab image (same swizzle as the copies).
    const int gn = t & 63, gq = t >> 6, gofs = gn * SK + 4 * (gq ^ chunk_swz(gn));
    auto gen = [&](int k0, float* Bbuf) {
      if (NWS > 4 && t >= 256) return;                                 // 64 rows x 4 k groups: the first four waves
      const int k = k0 + 4 * gq;                                       // < P + SK: rks is zero there, the bits too
      const unsigned m4 = k < 64 * W64 ? (unsigned)(bl[gn * W64 + (k >> 6)] >> (k & 63)) & 15u : 0u;
      *reinterpret_cast<float4*>(Bbuf + gofs) = make_float4((m4 & 1u) ? rks[k] : 0.f, (m4 & 2u) ? rks[k + 1] : 0.f,
                                                            (m4 & 4u) ? rks[k + 2] : 0.f, (m4 & 8u) ? rks[k + 3] : 0.f);
    };
    strip_gemm<MW, NWS, 1>(G, P, n0, nlive, ring, kk, nullptr, w.zero, gen);
    // ---- <Y^, G> per column, loss partial, dY
    float tc[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      float a = 0.f;
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) a = __builtin_fmaf(acc[m][nn][r], G[m][nn][r], a);
      tc[nn] = a;
    }
    strip_colsum<NWS>(tc, part);
    if (wave == 0) {
      float s = 0.f;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) s += (lane < 16 && n0 + 16 * nn + lr < pend) ? tc[nn] : 0.f;
      s = wave_sum(s);
      if (lane == 0) w.lpart[(size_t)b * ns + sx] = s;
    }
    if (!want_grad) return;
    const float gs = g0 / (cnt_of(w.cnt, b) + 1e-6f);
    float tg[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) tg[nn] = gs * tc[nn];
    rows_out(w.dY + (size_t)b * R,
             [&](int m, int nn, int r) { return (gs * G[m][nn][r] - acc[m][nn][r] * tg[nn]) * ry[nn]; });
  } else {
    // item < ns: dV^ = dY A, dv = (dV^ - v^ s) / n_v ;  item >= ns: dX^ = x^ dSs', dx = (dX^ - x^ s) / n_x
    // (both products in one work-group - a loop over the two - compiles to 256 VGPRs and runs slower: 204 us vs 127 us)
    const bool isx = item >= ns;
    // the tensor being differentiated ([C][P]: the caller's, or k_tl_pre's image of a channels_last one)
    const float* src = nhwc ? (isx ? w.xT : w.vT) + (size_t)b * R : (isx ? dd.x : dd.v) + so;
    const float* rn = (isx ? w.rx : w.rv) + (size_t)b * P;
    const float* Mat = (isx ? w.dSs : w.A) + (size_t)b * Q;
    const float* Aop = isx ? src : w.dY + (size_t)b * R;
    auto none = [](int, float*) {};
    strip_gemm<MW, NWS, 0>(acc, P, n0, nlive, ring, Aop, Mat, w.zero, none);
    float hat[MW][4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f}, rj[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) rj[nn] = n0 + 16 * nn + lr < pend ? rn[n0 + 16 * nn + lr] : 0.f;
    if (!(TL_ABL & 1))
      rows_in(src, [&](int m, int nn, int r, float val) {
        hat[m][nn][r] = val * rj[nn];
        s[nn] = __builtin_fmaf(hat[m][nn][r], acc[m][nn][r], s[nn]);
      });
    strip_colsum<NWS>(s, part);
    if (!(TL_ABL & 1) || s[0] == 123.456f) {
      auto val = [&](int m, int nn, int r) { return (acc[m][nn][r] - hat[m][nn][r] * s[nn]) * rj[nn]; };
      if (nhwc) pix_out((isx ? dd.dx : dd.dv) + so, val); else rows_out((isx ? dd.dx : dd.dv) + so, val);
    }
  }
}

// forward-only launches: the same reduction as a launch of its own
__global__ void __launch_bounds__(TB) k_tl_finish(const float* __restrict__ lpart, const float* __restrict__ cnt,
                                                  float* __restrict__ loss_n, float* __restrict__ loss,
                                                  float* __restrict__ pos_num, float* __restrict__ pos_mean, int B, int N,
                                                  int strips, int P) {
  __shared__ float red[TB / 64];
  tl_finish(lpart, cnt, loss_n, loss, pos_num, pos_mean, B, N, strips, P, red);
}

#define TL_CHECK(what)                        \
  do {                                        \
    if (int e = check_launch(what)) return e; \
  } while (0)

template <int MW, int NWS>
int run_tiled_c(const pixpro_problem& pb, const Dirs& D, const TlWs& w, float* loss_n, float* pos_num, float* pos_mean,
                float* loss, hipStream_t st) {
  const int B = pb.N * pb.ndir, N = pb.N, C = pb.C, P = pb.H * pb.W, nt = (P + 63) / 64;
  const bool want_grad = D.d[0].dx != nullptr;
  const float g0 = pb.grad_scale * -2.f / (float)N;
  const TlFin fin{loss_n, loss, pos_num, pos_mean, B, nt};
  hipLaunchKernelGGL(k_tl_pre, dim3((6 + PREP_S) * B), dim3(TB), 0, st, pb, D, w);
  TL_CHECK("k_tl_pre");
  hipLaunchKernelGGL(k_tl_gram<0>, dim3(nt * (nt + 1) / 2 * B), dim3(TBG), 0, st, D, w, N, C, P, pb.p, pb.clamp, fin);
  TL_CHECK("k_tl_gram<0>");
  hipLaunchKernelGGL((k_tl_strip<MW, 0, NWS>), dim3(nt * B), dim3(64 * NWS), 0, st, pb, D, w, g0, (int)want_grad, nt);
  TL_CHECK("k_tl_strip<0>");
  if (!want_grad) {
    hipLaunchKernelGGL(k_tl_finish, dim3(1), dim3(TB), 0, st, w.lpart, w.cnt, loss_n, loss, pos_num, pos_mean, B, N, nt, P);
    TL_CHECK("k_tl_finish");
    return PIXPRO_OK;
  }
  hipLaunchKernelGGL(k_tl_gram<1>, dim3(nt * (nt + 1) / 2 * B), dim3(TBG), 0, st, D, w, N, C, P, pb.p, pb.clamp, fin);
  TL_CHECK("k_tl_gram<1>");
  hipLaunchKernelGGL((k_tl_strip<MW, 1, NWS>), dim3(2 * nt * B), dim3(64 * NWS), 0, st, pb, D, w, g0, 1, nt);
  TL_CHECK("k_tl_strip<1>");
  return PIXPRO_OK;
}

}  // namespace

bool tiled_supported(const pixpro_problem& pb) {
  const int P = pb.H * pb.W;
  return (pb.layout == 0 || pb.layout == 1) && (pb.C == 64 || pb.C == 128 || pb.C == 256) && P % 4 == 0 && P >= 4 && P <= TL_MAXP;
}

size_t tiled_workspace_bytes(const pixpro_problem& pb) { return tl_bytes(pb); }

int run_tiled(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean, float* loss,
              void* workspace, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < tl_bytes(pb)) {
    set_error("workspace too small: %zu < %zu", ws_bytes, tl_bytes(pb));
    return PIXPRO_ERR_WORKSPACE;
  }
  Dirs D;
  for (int d = 0; d < PIXPRO_MAX_DIRS; ++d) D.d[d] = dirs[d < pb.ndir ? d : 0];
  const TlWs w = tl_carve(workspace, pb);
  switch (pb.C / 64) {
    case 1: return run_tiled_c<1, 4>(pb, D, w, loss_n, pos_num, pos_mean, loss, st);
    case 2: return run_tiled_c<2, 4>(pb, D, w, loss_n, pos_num, pos_mean, loss, st);
    default: return run_tiled_c<2, 8>(pb, D, w, loss_n, pos_num, pos_mean, loss, st);      // C = 256: 8 waves x 32 rows
  }
}

}  // namespace pixpro
