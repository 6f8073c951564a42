// Single-launch PPM -> flow-matched pixel loss -> backward for grids with P = H*W <= 64
// (7x7 at 224^2 crops, 2x2 at 64^2) and C in {128, 256}.
//
// One 512-thread work-group (8 wave64) per (loss direction, sample).  The sample's
// [C][P] fp32 tiles are contiguous in HBM (NCHW), so each is one linear, fully coalesced
// 16 B/lane stream into LDS.  All six P x P / C x P contractions run on the exact-fp32 MFMA
// v_mfma_f32_16x16x4_f32 with both operands read from LDS; S/A/dS never leave the CU.
// Row reductions over C (three L2 norms and the two projection dot products) are done on
// the accumulator registers with wave shuffles + one LDS hop across the 8 waves.
//
// Phases (reference contrast/models/PixPro.py:339-363,380,92-247):
//   pts   q/k bin centres, flow displacement (bilinear) + validity (nearest), bit-mask M
//   S     B0 <- x_hat ; S = x_hat^T x_hat ; A = pow(clamp(S)), D = dA/dS          (P x P, K=C)
//   Y     B1 <- v_hat ; Y = v_hat A ; n_y ; Y_hat                                  (C x P, K=P)
//   G     B0 <- k_hat ; G = k_hat M^T ; t_i = <Y_hat_i, G_i> ; loss_n             (C x P, K=P)
//   dY    B0 <- dY = (gs G - Y_hat gs t)/n_y
//   dA    dS = (dY^T v_hat) o D ; dSs = dS + dS^T                                 (P x P, K=C)
//   dV    dv = proj(dY A, v_hat)/n_v  -> HBM                                      (C x P, K=P)
//   dX    B1 <- x_hat ; dx = proj(x_hat dSs, x_hat)/n_x -> HBM                    (C x P, K=P)
#include "common.h"

namespace pixpro {

namespace {

constexpr int NTHREADS = 512;
constexpr int NWAVES = NTHREADS / WAVE;
constexpr int LS = 80;  // row stride of the small P x P matrices (80 % 32 == 16: conflict-free row pairs)

struct Smem {
  float* B0;     // [C*P + 4]
  float* B1;     // [C*P + 4]
  float* AA;     // [LR][LS]   A = pow(clamp(S))
  float* Dm;     // [LR][LS]   dA/dS, later dS + dS^T
  float* DS;     // [LR][LS]   dS
  float* nx;     // [64]
  float* nv;
  float* ny;
  float* tcol;
  float* scol;
  float* part;   // [NWAVES][64]
  float* qx;     // [64] each
  float* qy;
  float* kx;
  float* ky;
  float* misc;   // [8]: diag, cnt, ...
  unsigned long long* mbits;  // [64]
  int* valid;    // [64]
};

__host__ __device__ inline int lr_rows(int P) { return (P + 3) / 4 * 4; }

__host__ __device__ inline size_t smem_floats(int C, int P) {
  return 2 * ((size_t)C * P + 4) + 3 * (size_t)lr_rows(P) * LS + 5 * 64 + NWAVES * 64 + 4 * 64 + 8 + 2 * 64 + 64;
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
  s.nx = p; p += 64; s.nv = p; p += 64; s.ny = p; p += 64; s.tcol = p; p += 64; s.scol = p; p += 64;
  s.part = p; p += NWAVES * 64;
  s.qx = p; p += 64; s.qy = p; p += 64; s.kx = p; p += 64; s.ky = p; p += 64;
  s.misc = p; p += 8;
  s.mbits = (unsigned long long*)p; p += 2 * 64;
  s.valid = (int*)p;
  return s;
}

// linear HBM -> LDS copy of one [C][P] tile (count % 4 == 0, both 16-B aligned)
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src, int count) {
  for (int e = threadIdx.x * 4; e < count; e += NTHREADS * 4)
    *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(src + e);
}
__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* __restrict__ src, int count) {
  for (int e = threadIdx.x * 4; e < count; e += NTHREADS * 4)
    *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(src + e);
}

// norms[j] = max(||buf[:,j]||, 1e-12) for j < P  (ends with a barrier)
__device__ __forceinline__ void col_norms(const float* buf, float* norms, float* part, int C, int P) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows = C / NWAVES;
  float acc = 0.f;
  if (lane < P)
    for (int c = wave * rows; c < (wave + 1) * rows; ++c) { float t = buf[c * P + lane]; acc += t * t; }
  part[wave * 64 + lane] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += part[w * 64 + threadIdx.x];
    norms[threadIdx.x] = fmaxf(sqrtf(t), 1e-12f);
  }
  __syncthreads();
}

// buf[c][j] /= norms[j]   (no barrier)
__device__ __forceinline__ void scale_cols(float* buf, const float* norms, int C, int P) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < P) {
    const float n = norms[lane];
    for (int c = wave; c < C; c += NWAVES) buf[c * P + lane] = buf[c * P + lane] / n;
  }
}

// Cross-wave column reduction of per-lane partials v[n] (column 16n + (lane&15)).
// out[col] = sum over all rows; ends with a barrier.
template <int NT>
__device__ __forceinline__ void col_reduce(float (&v)[NT], float* out, float* part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float s = v[n];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) part[wave * 64 + 16 * n + lane] = s;
  }
  __syncthreads();
  if (threadIdx.x < 16 * NT) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += part[w * 64 + threadIdx.x];
    out[threadIdx.x] = t;
  }
  __syncthreads();
}

// acc[m][n] += sum_k Aop(row tile m, k) * Bop(k, col tile n) over `ksteps` steps of 4.
template <int MT, int NT, class FA, class FB>
__device__ __forceinline__ void gemm(f32x4 (&acc)[MT][NT], int ksteps, FA fa, FB fb) {
  for (int ks = 0; ks < ksteps; ++ks) {
    float a[MT], b[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = fa(m, ks);
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = fb(n, ks);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = mfma16(a[m], b[n], acc[m][n]);
  }
}

template <int MT, int NT>
__global__ void __launch_bounds__(NTHREADS)
ppm_fused_kernel(pixpro_problem pb, Dirs D, float* __restrict__ loss_n, float* __restrict__ pos_num,
                 float* __restrict__ pos_mean) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int C = pb.C, P = pb.H * pb.W, N = pb.N;
  const int LR = lr_rows(P);
  const Smem sm = carve(smem_raw, C, P);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x, dir = b / N, n = b % N;
  const pixpro_dir& dd = D.d[dir];
  const size_t tile = (size_t)C * P;
  const float* gx = dd.x + (size_t)n * tile;
  const float* gv = dd.v + (size_t)n * tile;
  const float* gk = dd.k + (size_t)n * tile;
  const int KS = LR / 4;          // k-steps over the pixel dimension
  const int KC = C / 4;           // k-steps over the channel dimension
  const bool want_grad = dd.dx != nullptr;

  // ---- stage x ; bin centres, flow displacement, validity (independent of the loads) ----
  load_tile(sm.B0, gx, (int)tile);
  if (tid < 4) { sm.B0[tile + tid] = 0.f; sm.B1[tile + tid] = 0.f; }
  if (tid < 64) {
    float qx = 0.f, qy = 0.f, kx = 0.f, ky = 0.f;
    int valid = 0;
    if (tid < P) {
      int Ho, Wo;
      frame_size(pb, dd, Ho, Wo);
      const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
      const Crop cq = crop_of(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
      const Crop ck = crop_of(dd.coord_k + (size_t)n * 10, pb.H, pb.W);
      const int row = tid / pb.W, col = tid % pb.W;
      qx = centre_x(cq, col, wo1); qy = centre_y(cq, row, ho1);
      kx = centre_x(ck, col, wo1); ky = centre_y(ck, row, ho1);
      bool v = true;
      if (dd.flow) {
        const FlowGeom fg = flow_geom(pb.Hf, pb.Wf, Ho, Wo);
        float ox, oy;
        displace_point(dd.flow + (size_t)n * 2 * pb.Hf * pb.Wf,
                       dd.mask ? dd.mask + (size_t)n * pb.Hf * pb.Wf : nullptr, fg, qx, qy, ox, oy, v);
        qx = ox; qy = oy;
      }
      valid = v ? 1 : 0;
      if (tid == 0) sm.misc[0] = fmaxf(bin_diag(cq, wo1, ho1), bin_diag(ck, wo1, ho1));
    }
    sm.qx[tid] = qx; sm.qy[tid] = qy; sm.kx[tid] = kx; sm.ky[tid] = ky; sm.valid[tid] = valid;
  }
  __syncthreads();

  // ---- positive-pair bit mask: row i of M = ballot over j (PixPro.py:217-223) ----
  {
    const float diag = sm.misc[0];
    const float kxj = sm.kx[lane], kyj = sm.ky[lane];
    int cnt_w = 0;
    for (int i = wave; i < 64; i += NWAVES) {
      bool pred = false;
      if (i < P && lane < P) {
        const float dx = sm.qx[i] - kxj, dy = sm.qy[i] - kyj;
        const float dist = sqrtf(dx * dx + dy * dy) / diag;
        pred = (dist < pb.pos_ratio) && (sm.valid[i] != 0);
      }
      const unsigned long long bits = __ballot(pred);
      if (lane == 0) sm.mbits[i] = bits;
      cnt_w += __popcll(bits);
    }
    if (lane == 0) sm.part[wave * 64] = (float)cnt_w;
  }
  __syncthreads();
  if (tid == 0) {
    float c = 0.f;
    for (int w = 0; w < NWAVES; ++w) c += sm.part[w * 64];
    sm.misc[1] = c;
  }
  // ---- x_hat in B0 ----
  col_norms(sm.B0, sm.nx, sm.part, C, P);
  scale_cols(sm.B0, sm.nx, C, P);
  // prefetch of v into B1 overlaps the S phase
  load_tile(sm.B1, gv, (int)tile);
  __syncthreads();

  // ---- S = x_hat^T x_hat ; A, D ----
  for (int t0 = wave; t0 < NT * NT; t0 += NWAVES) {
    const int mi = t0 / NT, nj = t0 % NT;
    const int ca = min(16 * mi + lr, P - 1), cb = min(16 * nj + lr, P - 1);
    f32x4 acc[1][1] = {{{0.f, 0.f, 0.f, 0.f}}};
    gemm<1, 1>(acc, KC,
               [&](int, int ks) { return sm.B0[(4 * ks + lq) * P + ca]; },
               [&](int, int ks) { return sm.B0[(4 * ks + lq) * P + cb]; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * mi + 4 * lq + r, j = 16 * nj + lr;
      if (i < LR) {
        float a = 0.f, d = 0.f;
        if (i < P && j < P) pow_map(acc[0][0][r], pb.p, pb.clamp, a, d);
        sm.AA[i * LS + j] = a;
        sm.Dm[i * LS + j] = d;
      }
    }
  }
  __syncthreads();
  // ---- v_hat in B1 ----
  col_norms(sm.B1, sm.nv, sm.part, C, P);
  scale_cols(sm.B1, sm.nv, C, P);
  __syncthreads();

  // ---- Y = v_hat A ----
  f32x4 Y[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) Y[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm<MT, NT>(Y, KS,
               [&](int m, int ks) { return sm.B1[(16 * (wave + NWAVES * m) + lr) * P + min(4 * ks + lq, P - 1)]; },
               [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; });
  // k tile can stream into B0 now (x_hat no longer needed until the dX phase)
  load_tile(sm.B0, gk, (int)tile);
  {
    float v[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += Y[m][nn][r] * Y[m][nn][r];
      v[nn] = s;
    }
    col_reduce<NT>(v, sm.tcol, sm.part);   // also publishes the k tile (barriers inside)
  }
  if (tid < 64) sm.ny[tid] = fmaxf(sqrtf(sm.tcol[tid < 16 * NT ? tid : 0]), 1e-12f);
  if (pb.normalize_k) {
    col_norms(sm.B0, sm.scol, sm.part, C, P);
    scale_cols(sm.B0, sm.scol, C, P);
  }
  __syncthreads();
#pragma unroll
  for (int nn = 0; nn < NT; ++nn) {
    const float nyv = sm.ny[16 * nn + lr];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[m][nn][r] = Y[m][nn][r] / nyv;
  }
  if (dd.pred) {
    float* gp = dd.pred + (size_t)n * tile;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NT; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * (wave + NWAVES * m) + 4 * lq + r, i = 16 * nn + lr;
          if (i < P) gp[(size_t)c * P + i] = Y[m][nn][r];
        }
  }

  // ---- G = k_hat M^T ----
  f32x4 G[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) G[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    unsigned long long rowbits[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) rowbits[nn] = sm.mbits[16 * nn + lr];
    gemm<MT, NT>(G, KS,
                 [&](int m, int ks) { return sm.B0[(16 * (wave + NWAVES * m) + lr) * P + min(4 * ks + lq, P - 1)]; },
                 [&](int nn, int ks) { return ((rowbits[nn] >> (4 * ks + lq)) & 1ull) ? 1.f : 0.f; });
  }
  {
    float v[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += Y[m][nn][r] * G[m][nn][r];
      v[nn] = s;
    }
    col_reduce<NT>(v, sm.tcol, sm.part);   // barrier: every wave is done reading k_hat from B0
  }
  const float cnt = sm.misc[1];
  const float den = cnt + 1e-6f;
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < P; ++i) t += sm.tcol[i];
    loss_n[b] = t / den;
    pos_num[b] = cnt;
    pos_mean[b] = cnt / (float)P / (float)P;
  }
  if (!want_grad) return;

  // ---- dY -> B0 ----
  {
    const float gs = pb.grad_scale * -2.f / (float)N / den;
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const int i = 16 * nn + lr;
      if (i < P) {
        const float ti = gs * sm.tcol[i], nyv = sm.ny[i];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * (wave + NWAVES * m) + 4 * lq + r;
            sm.B0[c * P + i] = (gs * G[m][nn][r] - Y[m][nn][r] * ti) / nyv;
          }
      }
    }
  }
  __syncthreads();

  // ---- dA = dY^T v_hat ; dS = dA o D ----
  for (int t0 = wave; t0 < NT * NT; t0 += NWAVES) {
    const int mi = t0 / NT, nj = t0 % NT;
    const int ca = min(16 * mi + lr, P - 1), cb = min(16 * nj + lr, P - 1);
    f32x4 acc[1][1] = {{{0.f, 0.f, 0.f, 0.f}}};
    gemm<1, 1>(acc, KC,
               [&](int, int ks) { return sm.B0[(4 * ks + lq) * P + ca]; },
               [&](int, int ks) { return sm.B1[(4 * ks + lq) * P + cb]; });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * mi + 4 * lq + r, j = 16 * nj + lr;
      if (i < LR) sm.DS[i * LS + j] = acc[0][0][r] * sm.Dm[i * LS + j];
    }
  }
  __syncthreads();
  // dSs = dS + dS^T into Dm (rows/cols >= P are zero because D is zero there)
  for (int e = tid; e < LR * 64; e += NTHREADS) {
    const int i = e >> 6, j = e & 63;
    sm.Dm[i * LS + j] = (j < LR) ? sm.DS[i * LS + j] + sm.DS[j * LS + i] : 0.f;
  }
  // (visible after the barriers inside the dV epilogue's col_reduce; the dV gemm below reads AA only)

  // ---- dV_hat = dY A ; dv = (dV_hat - v_hat s)/n_v ----
  {
    f32x4 T[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) T[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm<MT, NT>(T, KS,
                 [&](int m, int ks) { return sm.B0[(16 * (wave + NWAVES * m) + lr) * P + min(4 * ks + lq, P - 1)]; },
                 [&](int nn, int ks) { return sm.AA[(4 * ks + lq) * LS + 16 * nn + lr]; });
    float v[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const int j = 16 * nn + lr;
      float s = 0.f;
      if (j < P) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += sm.B1[(16 * (wave + NWAVES * m) + 4 * lq + r) * P + j] * T[m][nn][r];
      }
      v[nn] = s;
    }
    col_reduce<NT>(v, sm.scol, sm.part);
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const int j = 16 * nn + lr;
      if (j < P) {
        const float sj = sm.scol[j], nvj = sm.nv[j];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = (16 * (wave + NWAVES * m) + 4 * lq + r) * P + j;
            sm.B1[idx] = (T[m][nn][r] - sm.B1[idx] * sj) / nvj;
          }
      }
    }
  }
  __syncthreads();
  store_tile(dd.dv + (size_t)n * tile, sm.B1, (int)tile);
  __syncthreads();

  // ---- x_hat back into B1 ; dX_hat = x_hat dSs ; dx ----
  load_tile(sm.B1, gx, (int)tile);
  __syncthreads();
  scale_cols(sm.B1, sm.nx, C, P);
  __syncthreads();
  {
    f32x4 T[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) T[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm<MT, NT>(T, KS,
                 [&](int m, int ks) { return sm.B1[(16 * (wave + NWAVES * m) + lr) * P + min(4 * ks + lq, P - 1)]; },
                 [&](int nn, int ks) { return sm.Dm[(4 * ks + lq) * LS + 16 * nn + lr]; });
    float v[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const int j = 16 * nn + lr;
      float s = 0.f;
      if (j < P) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += sm.B1[(16 * (wave + NWAVES * m) + 4 * lq + r) * P + j] * T[m][nn][r];
      }
      v[nn] = s;
    }
    col_reduce<NT>(v, sm.scol, sm.part);
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const int j = 16 * nn + lr;
      if (j < P) {
        const float sj = sm.scol[j], nxj = sm.nx[j];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = (16 * (wave + NWAVES * m) + 4 * lq + r) * P + j;
            sm.B1[idx] = (T[m][nn][r] - sm.B1[idx] * sj) / nxj;
          }
      }
    }
  }
  __syncthreads();
  store_tile(dd.dx + (size_t)n * tile, sm.B1, (int)tile);
}

template <int MT, int NT>
int launch(const pixpro_problem& pb, const Dirs& D, float* loss_n, float* pos_num, float* pos_mean, hipStream_t st) {
  const int P = pb.H * pb.W;
  const size_t bytes = smem_floats(pb.C, P) * sizeof(float);
  auto kern = ppm_fused_kernel<MT, NT>;
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
  return smem_floats(pb.C, P) * sizeof(float) <= 160 * 1024;
}

int run_fused(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
              hipStream_t st) {
  Dirs D;
  for (int d = 0; d < PIXPRO_MAX_DIRS; ++d) D.d[d] = dirs[d < pb.ndir ? d : 0];
  const int P = pb.H * pb.W;
  const int NT = (P + 15) / 16;
  if (pb.C == 256) {
    switch (NT) {
      case 1: return launch<2, 1>(pb, D, loss_n, pos_num, pos_mean, st);
      case 2: return launch<2, 2>(pb, D, loss_n, pos_num, pos_mean, st);
      case 3: return launch<2, 3>(pb, D, loss_n, pos_num, pos_mean, st);
      default: return launch<2, 4>(pb, D, loss_n, pos_num, pos_mean, st);
    }
  }
  switch (NT) {
    case 1: return launch<1, 1>(pb, D, loss_n, pos_num, pos_mean, st);
    case 2: return launch<1, 2>(pb, D, loss_n, pos_num, pos_mean, st);
    case 3: return launch<1, 3>(pb, D, loss_n, pos_num, pos_mean, st);
    default: return launch<1, 4>(pb, D, loss_n, pos_num, pos_mean, st);
  }
}

}  // namespace pixpro
