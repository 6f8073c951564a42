// Tiled multi-launch PPM -> flow-matched pixel loss -> backward for grids too large for the single-launch kernel
// (P = H*W > 64, e.g. the 14 x 14 grid of 448^2 crops: one [256][196] fp32 tile is 200 KB, one P x P matrix 154 KB, against
// 160 KB of LDS).  Seven launches instead of the ~22 of ppm_generic.hip; the P x P matrices live in the caller's
// workspace (L2 / Infinity Cache resident at these sizes), everything between two matrix products is an epilogue:
//
//   k_tl_prep     sample points (flow-displaced bin centres), positive-pair BIT rows, counts          PixPro.py:132-223
//   k_tl_rnorms   1 / max(||.||, 1e-12) of the columns of x, v, k                                      :344,348,403
//   k_tl_gram<0>  S = x^T x on 64 x 64 tiles (K = C), epilogue: normalise, clamp/pow -> A, dA/dS       :354-358
//   k_tl_strip<0> per 64-column strip, all C rows in one work-group: Y = v^ A, column norms, Y^ (:361,380);
//                 G = k^ M^T with M expanded from the bit rows (:239); <Y^,G> -> loss partials (:241); dY
//   k_tl_gram<1>  dA = dY^T v^ on tile PAIRS (i,j),(j,i), epilogue: dS = dA o D, dSs = dS + dS^T
//   k_tl_strip<1|2>  dV^ = dY A -> dv ; dX^ = x^ dSs -> dx   (one launch, blockIdx.z selects)
//   k_tl_finish   loss_n, loss                                                                          :241,247
//
// All products run on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32 with operands staged through LDS in 16-deep slabs
// (next slab requested into registers before the MFMAs of the current one).  Same math and tolerances as the other two
// paths; the bit-exact part (points, threshold, mask) uses the same helpers as ppm_fused.hip.
#include "common.h"

namespace pixpro {
namespace {

constexpr int TB = 256;          // threads per work-group: 4 waves
constexpr int TSB = 80;          // LDS row stride of a 64-wide operand slab (conflict-free MFMA operand reads)

struct TlWs {                    // workspace carve (floats unless noted)
  float *rx, *rv, *rk;           // [B][P] reciprocal column norms
  float *cnt, *lpart;            // [B], [B][strips]
  float *A, *Dm, *dSs, *Av;      // [B][P][P]; Av[s][i] = rv[s] A[s][i], dSs rows pre-scaled by rx
  float *dY;                     // [B][C][P]
  unsigned long long* bits;      // [B][P][W64]
};

inline size_t al256(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

size_t tl_bytes(const pixpro_problem& pb) {
  const size_t B = (size_t)pb.N * pb.ndir, P = (size_t)pb.H * pb.W, R = (size_t)pb.C * P, Q = P * P, W64 = (P + 63) / 64;
  const size_t strips = (P + 63) / 64;
  return 3 * al256(B * P * 4) + al256(B * 4) + al256(B * strips * 4) + 4 * al256(B * Q * 4) + al256(B * R * 4) +
         al256(B * P * W64 * 8) + 4096;
}

TlWs tl_carve(void* ws, const pixpro_problem& pb) {
  const size_t B = (size_t)pb.N * pb.ndir, P = (size_t)pb.H * pb.W, R = (size_t)pb.C * P, Q = P * P, W64 = (P + 63) / 64;
  const size_t strips = (P + 63) / 64;
  char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](size_t bytes) { char* r = p; p += al256(bytes); return r; };
  TlWs w;
  w.rx = (float*)take(B * P * 4); w.rv = (float*)take(B * P * 4); w.rk = (float*)take(B * P * 4);
  w.cnt = (float*)take(B * 4); w.lpart = (float*)take(B * strips * 4);
  w.A = (float*)take(B * Q * 4); w.Dm = (float*)take(B * Q * 4); w.dSs = (float*)take(B * Q * 4);
  w.Av = (float*)take(B * Q * 4);
  w.dY = (float*)take(B * R * 4);
  w.bits = (unsigned long long*)take(B * P * W64 * 8);
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

// ---- k_tl_prep: PREP_S work-groups per (direction, sample), each takes every PREP_S-th group of 4 rows ------------------
constexpr int TL_MAXP = 1024, PREP_S = 4;
__global__ void __launch_bounds__(TB) k_tl_prep(pixpro_problem pb, Dirs D, unsigned long long* __restrict__ bits,
                                                float* __restrict__ cnt) {
  __shared__ float qx[TL_MAXP], qy[TL_MAXP], kx[TL_MAXP], ky[TL_MAXP];
  __shared__ int valid[TL_MAXP];
  __shared__ float sT;
  __shared__ int wcnt[TB / 64];
  const int b = blockIdx.x, d = b / pb.N, n = b % pb.N, P = pb.H * pb.W, W64 = (P + 63) / 64;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int row0 = 4 * blockIdx.y + wave, rstep = 4 * PREP_S;        // this wave's rows: row0, row0 + rstep, ...
  const pixpro_dir& dd = D.d[d];
  int Ho, Wo;
  frame_size(pb, dd, Ho, Wo);
  const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
  const Crop cq = crop_of(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
  const Crop ck = crop_of(dd.coord_k + (size_t)n * 10, pb.H, pb.W);
  for (int i = t; i < P; i += TB) {
    const int row = i / pb.W, col = i % pb.W;
    kx[i] = centre_x(ck, col, wo1); ky[i] = centre_y(ck, row, ho1);
    if ((i >> 2) % PREP_S != (int)blockIdx.y) continue;              // q points of the rows this work-group owns
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
    for (int w = 0; w < W64; ++w) {
      const int j = 64 * w + lane;
      bool pred = false;
      if (j < P) {
        const float dx = xi - kx[j], dy = yi - ky[j];
        pred = (dx * dx + dy * dy < T) && vi;
      }
      const unsigned long long m = __ballot(pred);
      if (lane == 0) bits[((size_t)b * P + i) * W64 + w] = m;
      c += __popcll(m);
    }
  }
  if (lane == 0) wcnt[wave] = c;
  __syncthreads();
  if (t == 0) {
    int tot = 0;
    for (int w = 0; w < TB / 64; ++w) tot += wcnt[w];
    if (tot) atomicAdd(&cnt[b], (float)tot);      // integers below 2^24: exact in any order (cnt is zeroed by the host side)
  }
}

// ---- k_tl_rnorms: grid (ceil(P/64), B, 3) -----------------------------------------------------------------------------
__global__ void __launch_bounds__(TB) k_tl_rnorms(Dirs D, float* __restrict__ rx, float* __restrict__ rv,
                                                  float* __restrict__ rk, float* __restrict__ cnt, int N, int C, int P,
                                                  int normalize_k) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, which = blockIdx.z, tx = threadIdx.x & 63, ty = threadIdx.x >> 6, j = blockIdx.x * 64 + tx;
  if (blockIdx.x == 0 && which == 0 && threadIdx.x == 0) cnt[b] = 0.f;   // k_tl_prep (next launch) accumulates into it
  const pixpro_dir& dd = D.d[b / N];
  const float* s = (which == 0 ? dd.x : which == 1 ? dd.v : dd.k) + (size_t)(b % N) * C * P;
  float* o = (which == 0 ? rx : which == 1 ? rv : rk) + (size_t)b * P;
  if (which == 2 && !normalize_k) {
    if (ty == 0 && j < P) o[j] = 1.f;
    return;
  }
  float a0 = 0.f, a1 = 0.f;
  if (j < P) {
    for (int c = ty; c < C; c += 8) {
      const float t0 = s[(size_t)c * P + j], t1 = (c + 4 < C) ? s[(size_t)(c + 4) * P + j] : 0.f;
      a0 = __builtin_fmaf(t0, t0, a0);
      a1 = __builtin_fmaf(t1, t1, a1);
    }
  }
  red[ty][tx] = a0 + a1;
  __syncthreads();
  if (ty == 0 && j < P) o[j] = 1.f / fmaxf(sqrtf(red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]), 1e-12f);
}

// ---- P x P products over K = C on 64 x 64 tiles --------------------------------------------------------------------------
// One 64 x 64 x C product: acc[i][j] += sum_c L[c*P + m0 + ..] * Rm[c*P + n0 + ..]; wave w owns the 32 x 32 block
// (w >> 1, w & 1).  Columns beyond P read as zero.
__device__ __forceinline__ void gram_tile(f32x4 (&acc)[2][2], const float* __restrict__ L, const float* __restrict__ Rm,
                                          int m0, int n0, int C, int P, float* As, float* Bs) {
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int col = t & 63, k4 = t >> 6;                 // loader: 4 k rows per thread, 64 consecutive columns per wave
  const bool mok = m0 + col < P, nok = n0 + col < P;
  // 16 x 16 sub-tiles wholly beyond P (P = 196 = 12.25 sub-tiles per side) issue no MFMA (wave-uniform)
  bool live[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) live[i][j] = (m0 + wm + 16 * i < P) && (n0 + wn + 16 * j < P);
  float ra[4], rb[4];
  auto request = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t o = (size_t)(k0 + k4 + 4 * e) * P;
      ra[e] = mok ? L[o + m0 + col] : 0.f;
      rb[e] = nok ? Rm[o + n0 + col] : 0.f;
    }
  };
  request(0);
  for (int k0 = 0; k0 < C; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { As[(k4 + 4 * e) * TSB + col] = ra[e]; Bs[(k4 + 4 * e) * TSB + col] = rb[e]; }
    __syncthreads();
    if (k0 + 16 < C) request(k0 + 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = As[(4 * ks + lq) * TSB + wm + 16 * i + lr];
        bv[i] = Bs[(4 * ks + lq) * TSB + wn + 16 * i + lr];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (live[i][j]) acc[i][j] = mfma16(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
}

// MODE 0: S -> A, D on the tiles ti <= tj, mirrored (grid: pairs, 1, B).  MODE 1: dSs on tile pairs (grid: pairs, 1, B).
// loss_n[b] = sum of the strip partials / (cnt + 1e-6) (PixPro.py:241); loss = -2 sum_d mean_n loss_n (:247,431); one block
__device__ __forceinline__ void tl_finish(const float* __restrict__ lpart, const float* __restrict__ cnt,
                                          float* __restrict__ loss_n, float* __restrict__ loss, float* __restrict__ pos_num,
                                          float* __restrict__ pos_mean, int B, int N, int strips, int P, float* red) {
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += TB) {
    float s = 0.f;
    for (int k = 0; k < strips; ++k) s += lpart[(size_t)b * strips + k];
    const float ln = s / (cnt[b] + 1e-6f);
    loss_n[b] = ln;
    pos_num[b] = cnt[b];
    pos_mean[b] = cnt[b] / (float)P / (float)P;     // reference: mean(-1).mean(-1) in fp32 (< 1 ulp apart)
    acc += ln;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tsum = 0.f;
    for (int k = 0; k < TB / 64; ++k) tsum += red[k];
    *loss = -2.f * tsum / (float)N;
  }
  __syncthreads();
}
struct TlFin {                   // outputs of the loss reduction (MODE 1 of k_tl_gram does it in its first block)
  float *loss_n, *loss, *pos_num, *pos_mean;
  int B, strips;
};

template <int MODE>
__global__ void __launch_bounds__(TB) k_tl_gram(Dirs D, TlWs w, int N, int C, int P, float p, float clampv, TlFin fin) {
  __shared__ float As[16 * TSB], Bs[16 * TSB];
  // The loss reduction needs every strip of the launch before (k_tl_strip<0>) and nothing of this one: when gradients
  // are wanted it rides in the first block of this launch instead of a one-block launch of its own.
  if (MODE == 1 && blockIdx.x == 0 && blockIdx.z == 0)
    tl_finish(w.lpart, w.cnt, fin.loss_n, fin.loss, fin.pos_num, fin.pos_mean, fin.B, N, fin.strips, P, As);
  __shared__ float L1[MODE ? 64 * 65 : 1], L2[MODE ? 64 * 65 : 1];
  const int b = blockIdx.z, t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15,
            lq = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const size_t R = (size_t)C * P, Q = (size_t)P * P;
  const pixpro_dir& dd = D.d[b / N];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    // S is symmetric: only the tiles (ti <= tj) are computed (10 of 16 at P = 196) and written twice, the mirror image
    // as one float4 per lane (four consecutive rows of the accumulator are four consecutive columns of the mirror; P % 4 == 0)
    const int nt = (P + 63) / 64;
    int ti = 0, rem = blockIdx.x;
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    const int tj = ti + rem, m0 = ti * 64, n0 = tj * 64;
    const float* x = dd.x + (size_t)(b % N) * R;
    f32x4 acc[2][2] = {{z, z}, {z, z}};
    gram_tile(acc, x, x, m0, n0, C, P, As, Bs);
    const float* rx = w.rx + (size_t)b * P;
    const float* rv = w.rv + (size_t)b * P;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gn = n0 + wn + 16 * j + lr, gm0 = m0 + wm + 16 * i + 4 * lq;
        if (gn >= P || gm0 >= P) continue;
        const float rxn = rx[gn], rvn = rv[gn];
        float a[4], d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gm = gm0 + r;
          pow_map(acc[i][j][r] * rx[gm] * rxn, p, clampv, a[r], d[r]);
          w.A[(size_t)b * Q + (size_t)gm * P + gn] = a[r];
          w.Dm[(size_t)b * Q + (size_t)gm * P + gn] = d[r];
          // the 1/||v|| of Y = v^ A folded into A's rows, so that the strip product stages v as it is
          w.Av[(size_t)b * Q + (size_t)gm * P + gn] = rv[gm] * a[r];
        }
        if (ti != tj) {
          const size_t o = (size_t)b * Q + (size_t)gn * P + gm0;
          *reinterpret_cast<float4*>(w.A + o) = make_float4(a[0], a[1], a[2], a[3]);
          *reinterpret_cast<float4*>(w.Dm + o) = make_float4(d[0], d[1], d[2], d[3]);
          *reinterpret_cast<float4*>(w.Av + o) = make_float4(rvn * a[0], rvn * a[1], rvn * a[2], rvn * a[3]);
        }
      }
  } else {
    // pair index -> (ti <= tj)
    const int nt = (P + 63) / 64;
    int ti = 0, rem = blockIdx.x;
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    const int tj = ti + rem, i0 = ti * 64, j0 = tj * 64;
    const float* dY = w.dY + (size_t)b * R;
    const float* v = dd.v + (size_t)(b % N) * R;
    const float* rv = w.rv + (size_t)b * P;
    const float* Dm = w.Dm + (size_t)b * Q;
    // dS1[i][j] = rv[j] * sum_c dY[c][i] v[c][j] * D[i][j] on tile (ti, tj); dS2 likewise on tile (tj, ti)
    for (int half = 0; half < 2; ++half) {
      const int m0 = half ? j0 : i0, n0 = half ? i0 : j0;
      float* Lt = half ? L2 : L1;
      if (half && ti == tj) break;
      f32x4 acc[2][2] = {{z, z}, {z, z}};
      gram_tile(acc, dY, v, m0, n0, C, P, As, Bs);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ln = wn + 16 * j + lr, gn = n0 + ln;
          const float rvn = gn < P ? rv[gn] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lm = wm + 16 * i + 4 * lq + r, gm = m0 + lm;
            Lt[lm * 65 + ln] = (gm < P && gn < P) ? acc[i][j][r] * rvn * Dm[(size_t)gm * P + gn] : 0.f;
          }
        }
    }
    __syncthreads();
    const float* T2 = ti == tj ? L1 : L2;
    float* out = w.dSs + (size_t)b * Q;
    const float* rx = w.rx + (size_t)b * P;     // the 1/||x|| of dX^ = x^ dSs folded into the rows of dSs
    for (int e = t; e < 64 * 64; e += TB) {
      const int r = e >> 6, c = e & 63;
      if (i0 + r < P && j0 + c < P) out[(size_t)(i0 + r) * P + j0 + c] = rx[i0 + r] * (L1[r * 65 + c] + T2[c * 65 + r]);
      if (ti != tj && j0 + r < P && i0 + c < P)
        out[(size_t)(j0 + r) * P + i0 + c] = rx[j0 + r] * (L2[r * 65 + c] + L1[c * 65 + r]);
    }
  }
}

// ---- C x 64 strips over K = P: all C rows of a strip of columns in one work-group -----------------------------------------
// acc[mt][nt] += sum_k Aop(c, k) * Bop(k, n);  wave w owns rows [w*C/4, (w+1)*C/4) = MW 16-row tiles.
// FA(c, k4): float4 of A for k = 4*k4 .. 4*k4+3 (k-fast source rows);  FB(k, n): scalar of B.
template <int MW, class FB>
__device__ __forceinline__ void strip_gemm(f32x4 (&acc)[MW][4], int P, int nlive, float* As, float* Bs,
                                           const float* __restrict__ Asrc, FB fb) {
  // A operand: rows of Asrc ([64*MW][P], k-contiguous) as they are (every column scale of the products is folded into the
  // P x P operand, see k_tl_gram).  A slab in LDS as [row][k], 20 floats per row: four lanes fetch one row's 64 B (16 full
  // segments per wave instruction) and store them as one 16-B LDS write; the MFMA operand reads are 2-way conflicted
  // (rows r and r + 8 share banks), which at 8 reads per 16 MFMAs does not show.
  constexpr int TSA = 20;
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const int arow = t >> 2, ak4 = t & 3;      // + 64 rows per pass, MW passes
  const int bcol = t & 63, bk = t >> 6;
  const float4* ap[MW];                      // this thread's float4 of the current slab, advanced by 4 float4 per slab
#pragma unroll
  for (int e = 0; e < MW; ++e) ap[e] = reinterpret_cast<const float4*>(Asrc + (size_t)(arow + 64 * e) * P) + ak4;
  float4 ra[MW];
  float rb[4];
  const int K4 = P / 4;
  auto request = [&](int k0) {
    const bool full = k0 + 16 <= P;          // uniform: only the last slab of P = 196 checks bounds
#pragma unroll
    for (int e = 0; e < MW; ++e) {
      ra[e] = (full || k0 / 4 + ak4 < K4) ? *ap[e] : make_float4(0.f, 0.f, 0.f, 0.f);
      ap[e] += 4;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + bk + 4 * e;
      rb[e] = (full || k < P) ? fb(k, bcol) : 0.f;
    }
  };
  request(0);
  for (int k0 = 0; k0 < P; k0 += 16) {
#pragma unroll
    for (int e = 0; e < MW; ++e) *reinterpret_cast<float4*>(&As[(arow + 64 * e) * TSA + 4 * ak4]) = ra[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) Bs[(bk + 4 * e) * TSB + bcol] = rb[e];
    __syncthreads();
    if (k0 + 16 < P) request(k0 + 16);     // the next slab travels under the MFMAs of this one
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float av[MW], bv[4];
#pragma unroll
      for (int m = 0; m < MW; ++m) av[m] = As[(wave * 16 * MW + 16 * m + lr) * TSA + 4 * ks + lq];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) bv[nn] = Bs[(4 * ks + lq) * TSB + 16 * nn + lr];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
        if (nn < nlive) {                // column tiles wholly beyond P issue no MFMA (uniform)
#pragma unroll
          for (int m = 0; m < MW; ++m) acc[m][nn] = mfma16(av[m], bv[nn], acc[m][nn]);
        }
    }
    __syncthreads();
  }
}

// column sums over all C rows of v[nn] (this lane's columns 16*nn + lr): shuffles over lq, one LDS hop over the 4 waves
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
  for (int nn = 0; nn < 4; ++nn) v[nn] = part[16 * nn + lr] + part[64 + 16 * nn + lr] + part[128 + 16 * nn + lr] + part[192 + 16 * nn + lr];
  __syncthreads();
}

// MODE 0: forward + loss + dY (grid: strips, B, 1).  MODE 1 / 2: dv / dx (grid: strips, B, 2; blockIdx.z selects).
template <int MW, int MODE>
__global__ void __launch_bounds__(TB) k_tl_strip(pixpro_problem pb, Dirs D, TlWs w, float g0, int want_grad) {
  constexpr int C = 64 * MW;
  __shared__ __attribute__((aligned(16))) float As[C * 20];
  __shared__ float Bs[16 * TSB];
  __shared__ float part[4 * 64];
  const int b = blockIdx.y, N = pb.N, P = pb.H * pb.W, n0 = blockIdx.x * 64, W64 = (P + 63) / 64;
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const size_t R = (size_t)C * P, Q = (size_t)P * P;
  const pixpro_dir& dd = D.d[b / N];
  const size_t so = (size_t)(b % N) * R;
  const int nlive = min(4, (P - n0 + 15) / 16);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  auto crow = [&](int m, int r) { return wave * 16 * MW + 16 * m + 4 * lq + r; };
  f32x4 acc[MW][4];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = z;

  if (MODE == 0) {
    const float* v = dd.v + so;
    const float* kk = dd.k + so;
    const float* rv = w.rv + (size_t)b * P;
    const float* rk = w.rk + (size_t)b * P;
    const float* Av = w.Av + (size_t)b * Q;
    (void)rv;
    // ---- Y = v^ A = v (diag(rv) A): the column scale of v sits in the rows of Av, the loader is a plain copy
    const bool cok = n0 + (t & 63) < P;        // this loader thread's column exists
    strip_gemm<MW>(acc, P, nlive, As, Bs, v, [&](int k, int nl) { return cok ? Av[(size_t)k * P + n0 + nl] : 0.f; });
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
      strip_colsum(s, part);
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
      float* gp = dd.pred + so;
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int i = n0 + 16 * nn + lr;
          if (i < P)
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[(size_t)crow(m, r) * P + i] = acc[m][nn][r];
        }
    }
    // ---- G = k^ M^T, M expanded from the bit rows: B(k = j, n = i) = bit j of row i
    f32x4 G[MW][4];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) G[m][nn] = z;
    const unsigned long long* brow = w.bits + ((size_t)b * P + min(n0 + (t & 63), P - 1)) * W64;
    const bool bok = n0 + (t & 63) < P;
    // G = k^ M^T = k (diag(rk) M^T): the column scale of k rides on the expanded mask bits
    strip_gemm<MW>(G, P, nlive, As, Bs, kk,
                   [&](int k, int) { return (bok && ((brow[k >> 6] >> (k & 63)) & 1ull)) ? rk[k] : 0.f; });
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
    strip_colsum(tc, part);
    if (wave == 0) {
      float s = 0.f;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) s += (lane < 16 && n0 + 16 * nn + lr < P) ? tc[nn] : 0.f;
      s = wave_sum(s);
      if (lane == 0) w.lpart[(size_t)b * gridDim.x + blockIdx.x] = s;
    }
    if (!want_grad) return;
    const float gs = g0 / (w.cnt[b] + 1e-6f);
    float* dY = w.dY + (size_t)b * R;
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int i = n0 + 16 * nn + lr;
      if (i < P) {
        const float ti = gs * tc[nn], ryv = ry[nn];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) dY[(size_t)crow(m, r) * P + i] = (gs * G[m][nn][r] - acc[m][nn][r] * ti) * ryv;
      }
    }
  } else {
    // blockIdx.z = 0: dV^ = dY A, dv = (dV^ - v^ s) / n_v ;  blockIdx.z = 1: dX^ = x^ dSs', dx = (dX^ - x^ s) / n_x
    // (both products in one work-group - a loop over the two - compiles to 256 VGPRs and runs slower: 204 us vs 127 us)
    const bool isx = blockIdx.z == 1;
    const float* src = (isx ? dd.x : dd.v) + so;               // the tensor being differentiated
    const float* rn = (isx ? w.rx : w.rv) + (size_t)b * P;
    const float* Mat = (isx ? w.dSs : w.A) + (size_t)b * Q;
    const float* Aop = isx ? src : w.dY + (size_t)b * R;
    const bool cok = n0 + (t & 63) < P;
    strip_gemm<MW>(acc, P, nlive, As, Bs, Aop, [&](int k, int nl) { return cok ? Mat[(size_t)k * P + n0 + nl] : 0.f; });
    float hat[MW][4][4];
    float s[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int j = n0 + 16 * nn + lr;
      const float rj = j < P ? rn[j] : 0.f;
      float a = 0.f;
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hat[m][nn][r] = j < P ? src[(size_t)crow(m, r) * P + j] * rj : 0.f;
          a = __builtin_fmaf(hat[m][nn][r], acc[m][nn][r], a);
        }
      s[nn] = a;
    }
    strip_colsum(s, part);
    float* out = (isx ? dd.dx : dd.dv) + so;
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
      const int j = n0 + 16 * nn + lr;
      if (j < P) {
        const float sj = s[nn], rj = rn[j];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(size_t)crow(m, r) * P + j] = (acc[m][nn][r] - hat[m][nn][r] * sj) * rj;
      }
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

template <int MW>
int run_tiled_c(const pixpro_problem& pb, const Dirs& D, const TlWs& w, float* loss_n, float* pos_num, float* pos_mean,
                float* loss, hipStream_t st) {
  const int B = pb.N * pb.ndir, N = pb.N, C = pb.C, P = pb.H * pb.W, nt = (P + 63) / 64;
  const bool want_grad = D.d[0].dx != nullptr;
  const float g0 = pb.grad_scale * -2.f / (float)N;
  const TlFin fin{loss_n, loss, pos_num, pos_mean, B, nt};
  hipLaunchKernelGGL(k_tl_rnorms, dim3(nt, B, 3), dim3(TB), 0, st, D, w.rx, w.rv, w.rk, w.cnt, N, C, P, pb.normalize_k);
  TL_CHECK("k_tl_rnorms");                 // also zeroes the counts k_tl_prep accumulates
  hipLaunchKernelGGL(k_tl_prep, dim3(B, PREP_S), dim3(TB), 0, st, pb, D, w.bits, w.cnt);
  TL_CHECK("k_tl_prep");
  hipLaunchKernelGGL(k_tl_gram<0>, dim3(nt * (nt + 1) / 2, 1, B), dim3(TB), 0, st, D, w, N, C, P, pb.p, pb.clamp, fin);
  TL_CHECK("k_tl_gram<0>");
  hipLaunchKernelGGL((k_tl_strip<MW, 0>), dim3(nt, B, 1), dim3(TB), 0, st, pb, D, w, g0, (int)want_grad);
  TL_CHECK("k_tl_strip<0>");
  if (!want_grad) {
    hipLaunchKernelGGL(k_tl_finish, dim3(1), dim3(TB), 0, st, w.lpart, w.cnt, loss_n, loss, pos_num, pos_mean, B, N, nt, P);
    TL_CHECK("k_tl_finish");
    return PIXPRO_OK;
  }
  hipLaunchKernelGGL(k_tl_gram<1>, dim3(nt * (nt + 1) / 2, 1, B), dim3(TB), 0, st, D, w, N, C, P, pb.p, pb.clamp, fin);
  TL_CHECK("k_tl_gram<1>");
  hipLaunchKernelGGL((k_tl_strip<MW, 1>), dim3(nt, B, 2), dim3(TB), 0, st, pb, D, w, g0, 1);
  TL_CHECK("k_tl_strip<1>");
  return PIXPRO_OK;
}

}  // namespace

bool tiled_supported(const pixpro_problem& pb) {
  const int P = pb.H * pb.W;
  return pb.layout == 0 && (pb.C == 64 || pb.C == 128 || pb.C == 256) && P % 4 == 0 && P >= 4 && P <= TL_MAXP;
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
    case 1: return run_tiled_c<1>(pb, D, w, loss_n, pos_num, pos_mean, loss, st);
    case 2: return run_tiled_c<2>(pb, D, w, loss_n, pos_num, pos_mean, loss, st);
    default: return run_tiled_c<4>(pb, D, w, loss_n, pos_num, pos_mean, loss, st);
  }
}

}  // namespace pixpro
