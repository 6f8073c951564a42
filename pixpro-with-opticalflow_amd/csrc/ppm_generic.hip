// Staged (multi-launch) implementation of the PPM -> pixel-loss path for ANY grid size.
//
// One thread per output element, operands read through L1/L2; P x P temporaries live in a
// caller-provided workspace.  This is the shape-agnostic path (14x14 grids, small-C test
// cases, the stand-alone featprop / regression_loss entry points) and the on-device
// cross-check of the single-launch MFMA kernel in ppm_fused.hip.  Math: SURVEY.md
// Appendix A; reference: contrast/models/PixPro.py:339-363 (featprop), :92-247
// (regression_loss), :46-89 (add_optical_flow).
#include "common.h"

namespace pixpro {

namespace {

constexpr int TPB = 256;
inline unsigned blocks_for(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

enum Src { SRC_X = 0, SRC_V = 1, SRC_K = 2 };

__device__ __forceinline__ const float* src_of(const Dirs& D, int which, int d) {
  return which == SRC_X ? D.d[d].x : (which == SRC_V ? D.d[d].v : D.d[d].k);
}

// dst[b,c,j] = src[b,c,j] / max(||src[b,:,j]||, 1e-12) (or a copy when do_norm == 0);
// norm[b,j] = the clamped norm.  F.normalize(dim=1), PixPro.py:344,348,380,403.
// Block = 64 columns x 4 channel groups (coalesced 256-B rows), grid (ceil(P/64), B).
__global__ void __launch_bounds__(TPB) k_normalize(Dirs D, int which, const float* __restrict__ ws_src,
                                                   float* __restrict__ dst, float* __restrict__ norm, int B, int N,
                                                   int C, int P, int do_norm) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6, j = blockIdx.x * 64 + tx;
  const float* s = ws_src ? ws_src + (size_t)b * C * P : src_of(D, which, b / N) + (size_t)(b % N) * C * P;
  float* o = dst + (size_t)b * C * P;
  float nn = 1.f;
  if (do_norm) {
    float acc = 0.f;
    if (j < P) {
#pragma unroll 8
      for (int c = ty; c < C; c += 4) { const float t = s[(size_t)c * P + j]; acc = __builtin_fmaf(t, t, acc); }
    }
    red[ty][tx] = acc;
    __syncthreads();
    nn = fmaxf(sqrtf(red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]), 1e-12f);
  }
  if (j >= P) return;
  if (norm && ty == 0) norm[(size_t)b * P + j] = nn;
#pragma unroll 8
  for (int c = ty; c < C; c += 4) {
    const float t = s[(size_t)c * P + j];
    o[(size_t)c * P + j] = do_norm ? t / nn : t;
  }
}

// Batched strided matrix product on the exact-fp32 MFMA: Out[b][m][n] = sum_k A(b,m,k) * B(b,k,n) with
// A(b,m,k) = A[b*sA + m*am + k*ak], B(b,k,n) = Bm[b*sB + k*bk + n*bn], Out[b*sO + m*on_m + n].
// 64 x 64 output tile per 256-thread block (each wave 32 x 32 = 2 x 2 MFMA tiles), K in steps of 16 staged
// through LDS as [k][m] / [k][n] (row stride 80 floats: conflict-free MFMA operand reads); out-of-range
// rows/columns/k are zero-filled, so any M, N, K works.  Covers the P x P Gram products (K = C) and the
// C x P propagation products (K = P) of the staged path.
struct MM {
  int M, N, K;
  long sA, sB, sO;
  int am, ak, bk, bn, om;
};

__global__ void __launch_bounds__(256) k_mm(const float* __restrict__ A, const float* __restrict__ Bm,
                                             float* __restrict__ Out, MM q) {
  constexpr int TS = 80;
  __shared__ float As[16 * TS], Bs[16 * TS];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15,
            lq = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* a = A + (long)blockIdx.z * q.sA;
  const float* bb = Bm + (long)blockIdx.z * q.sB;
  float* o = Out + (long)blockIdx.z * q.sO;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader mapping: walk the unit-stride dimension with consecutive threads; the next 16-deep slab is requested into
  // registers before the MFMAs of the current one and committed to LDS after them (global latency under the matrix work)
  const bool a_m_fast = q.am == 1, b_n_fast = q.bn == 1;
  int la[4], lb[4];              // LDS slots of this thread's four A / B elements
  long ga[4], gb[4];             // their global offsets at k0 = 0 (-1: outside M / N)
  int ka[4], kb[4];              // their k index inside the slab
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + e * 256;                         // 0 .. 1023
    const int am_ = a_m_fast ? (idx & 63) : (idx >> 4), ak_ = a_m_fast ? (idx >> 6) : (idx & 15);
    const int bn_ = b_n_fast ? (idx & 63) : (idx >> 4), bk_ = b_n_fast ? (idx >> 6) : (idx & 15);
    la[e] = ak_ * TS + am_; lb[e] = bk_ * TS + bn_;
    ka[e] = ak_; kb[e] = bk_;
    ga[e] = (m0 + am_ < q.M) ? (long)(m0 + am_) * q.am + (long)ak_ * q.ak : -1;
    gb[e] = (n0 + bn_ < q.N) ? (long)bk_ * q.bk + (long)(n0 + bn_) * q.bn : -1;
  }
  // 16 x 16 sub-tiles that lie entirely outside M x N (P = 196 = 12.25 tiles) issue no MFMA (wave-uniform)
  bool live[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) live[i][j] = (m0 + wm + 16 * i < q.M) && (n0 + wn + 16 * j < q.N);
  float ra[4], rb[4];
  auto request = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ra[e] = (ga[e] >= 0 && k0 + ka[e] < q.K) ? a[ga[e] + (long)k0 * q.ak] : 0.f;
      rb[e] = (gb[e] >= 0 && k0 + kb[e] < q.K) ? bb[gb[e] + (long)k0 * q.bk] : 0.f;
    }
  };
  request(0);
  for (int k0 = 0; k0 < q.K; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { As[la[e]] = ra[e]; Bs[lb[e]] = rb[e]; }
    __syncthreads();
    if (k0 + 16 < q.K) request(k0 + 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = As[(4 * ks + lq) * TS + wm + 16 * i + lr];
        bv[i] = Bs[(4 * ks + lq) * TS + wn + 16 * i + lr];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (live[i][j]) acc[i][j] = mfma16(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + 16 * i + 4 * lq + r, gn = n0 + wn + 16 * j + lr;
        if (gm < q.M && gn < q.N) o[(long)gm * q.om + gn] = acc[i][j][r];
      }
}

__global__ void k_powmap(const float* __restrict__ S, float* __restrict__ A, float* __restrict__ Dm, size_t n,
                         float p, float clampv) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= n) return;
  float a, d;
  pow_map(S[idx], p, clampv, a, d);
  A[idx] = a;
  if (Dm) Dm[idx] = d;
}

// Per (b,i): flow-displaced q centre + validity; per (b,j): k centre; per b: max bin diagonal.
// pts[b] = {qx[P], qy[P], valid[P], kx[P], ky[P]}, diag[b].   PixPro.py:132-201.
__global__ void k_points(pixpro_problem pb, Dirs D, float* __restrict__ pts, float* __restrict__ diag, int B) {
  const int P = pb.H * pb.W;
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)B * P) return;
  const int b = (int)(idx / P), i = (int)(idx % P), d = b / pb.N, n = b % pb.N;
  const pixpro_dir& dd = D.d[d];
  int Ho, Wo;
  frame_size(pb, dd, Ho, Wo);
  const float wo1 = (float)(Wo - 1), ho1 = (float)(Ho - 1);
  const Crop cq = crop_of(dd.coord_q + (size_t)n * 10, pb.H, pb.W);
  const Crop ck = crop_of(dd.coord_k + (size_t)n * 10, pb.H, pb.W);
  const int row = i / pb.W, col = i % pb.W;
  float qx = centre_x(cq, col, wo1), qy = centre_y(cq, row, ho1);
  bool valid = true;
  if (dd.lo_own) {            // lazy flow mode (see common.h)
    const FlowGeom fg = flow_geom(8 * pb.lo_h, 8 * pb.lo_w, Ho, Wo);
    float ox, oy;
    displace_point_lazy(lazy_src(pb, dd, n), fg, qx, qy, ox, oy, valid);
    qx = ox; qy = oy;
  } else if (dd.flow) {
    const FlowGeom fg = flow_geom(pb.Hf, pb.Wf, Ho, Wo);
    const float* fl = dd.flow + (size_t)n * 2 * pb.Hf * pb.Wf;
    const uint8_t* mk = dd.mask ? dd.mask + (size_t)n * pb.Hf * pb.Wf : nullptr;
    float ox, oy;
    displace_point(fl, mk, fg, qx, qy, ox, oy, valid);
    qx = ox; qy = oy;
  }
  float* o = pts + (size_t)b * 5 * P;
  o[i] = qx; o[P + i] = qy; o[2 * P + i] = valid ? 1.f : 0.f;
  o[3 * P + i] = centre_x(ck, col, wo1); o[4 * P + i] = centre_y(ck, row, ho1);
  if (i == 0) diag[b] = fmaxf(bin_diag(cq, wo1, ho1), bin_diag(ck, wo1, ho1));
}

// M[b,i,j] = (dist(q_i, k_j)/diag < pos_ratio) & valid_i   (PixPro.py:217-223)
__global__ void k_mask(const float* __restrict__ pts, const float* __restrict__ diag, float* __restrict__ M, int B,
                       int P, float pos_ratio) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)B * P * P) return;
  const int b = (int)(idx / ((size_t)P * P));
  const int r = (int)(idx % ((size_t)P * P)), i = r / P, j = r % P;
  const float* o = pts + (size_t)b * 5 * P;
  const float dx = o[i] - o[3 * P + j], dy = o[P + i] - o[4 * P + j];
  const float dist = sqrtf(dx * dx + dy * dy) / diag[b];
  M[idx] = (dist < pos_ratio && o[2 * P + i] != 0.f) ? 1.f : 0.f;
}

// one block per b: cnt[b] = sum M[b], pos_num / pos_mean outputs (PixPro.py:243-245)
__global__ void k_count(const float* __restrict__ M, float* __restrict__ cnt, float* __restrict__ pos_num,
                        float* __restrict__ pos_mean, int P) {
  __shared__ float red[TPB / WAVE];
  const int b = blockIdx.x;
  const float* m = M + (size_t)b * P * P;
  float acc = 0.f;
#pragma unroll 8
  for (int e = threadIdx.x; e < P * P; e += TPB) acc += m[e];      // 0/1 values: the sum is exact in any order
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < TPB / WAVE; ++w) t += red[w];
    cnt[b] = t;
    pos_num[b] = t;
    // reference: mean(-1).mean(-1) in fp32; cnt/P/P differs by < 1 ulp
    pos_mean[b] = t / (float)P / (float)P;
  }
}

// tcol[b,i] = sum_c A[b,c,i]*Bm[b,c,i]
__global__ void __launch_bounds__(TPB) k_coldot(const float* __restrict__ A, const float* __restrict__ Bm,
                                                float* __restrict__ tcol, int B, int C, int P) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6, i = blockIdx.x * 64 + tx;
  const float* a = A + (size_t)b * C * P;
  const float* bb = Bm + (size_t)b * C * P;
  float acc = 0.f;
  if (i < P) {
#pragma unroll 8
    for (int c = ty; c < C; c += 4) acc = __builtin_fmaf(a[(size_t)c * P + i], bb[(size_t)c * P + i], acc);
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < P) tcol[(size_t)b * P + i] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// one block per b: loss_n[b] = sum_i tcol[b,i] / (cnt[b] + 1e-6)   (PixPro.py:241)
__global__ void k_loss_n(const float* __restrict__ tcol, const float* __restrict__ cnt, float* __restrict__ loss_n,
                         int P) {
  __shared__ float red[TPB / WAVE];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int e = threadIdx.x; e < P; e += TPB) acc += tcol[(size_t)b * P + e];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < TPB / WAVE; ++w) t += red[w];
    loss_n[b] = t / (cnt[b] + 1e-6f);
  }
}

// loss = -2 * sum_d mean_n loss_n   (PixPro.py:247,431); single block, fixed order.
__global__ void k_finalize(const float* __restrict__ loss_n, float* __restrict__ loss, int B, int N) {
  __shared__ float red[TPB / WAVE];
  float acc = 0.f;
  for (int e = threadIdx.x; e < B; e += TPB) acc += loss_n[e];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < TPB / WAVE; ++w) t += red[w];
    *loss = -2.f * t / (float)N;
  }
}

// G[b,c,i] <- dY = (gs*G - Yh*gs*t_i) / ny_i,  gs = grad_scale * (-2/N) / (cnt+1e-6)
__global__ void k_dy(float* __restrict__ G, const float* __restrict__ Yh, const float* __restrict__ tcol,
                     const float* __restrict__ ny, const float* __restrict__ cnt, int B, int C, int P, float g0) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)B * C * P) return;
  const int b = (int)(idx / ((size_t)C * P)), i = (int)(idx % P);
  const float gs = g0 / (cnt[b] + 1e-6f);
  const float gy = gs * G[idx];
  G[idx] = (gy - Yh[idx] * (gs * tcol[(size_t)b * P + i])) / ny[(size_t)b * P + i];
}

// dSs[b,i,j] = dA[i,j]*D[i,j] + dA[j,i]*D[j,i]
// 32 x 32 tiles: tile (ti, tj) of E = dA o D is added to the LDS-transposed tile (tj, ti), so both reads are row-wise.
__global__ void __launch_bounds__(256) k_symm(const float* __restrict__ dA, const float* __restrict__ Dm,
                                              float* __restrict__ out, int B, int P) {
  __shared__ float tr[32][33];
  const int b = blockIdx.z, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const size_t base = (size_t)b * P * P;
  for (int r = ty; r < 32; r += 8) {                                             // E[j0 + r][i0 + tx] -> tr[r][tx]
    const int jj = j0 + r, ii = i0 + tx;
    tr[r][tx] = (jj < P && ii < P) ? dA[base + (size_t)jj * P + ii] * Dm[base + (size_t)jj * P + ii] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    if (i < P && j < P) {
      const size_t o = base + (size_t)i * P + j;
      out[o] = dA[o] * Dm[o] + tr[tx][r];
    }
  }
}

// out[b,c,j] = (T - Xh*s_j)/n_j written to the per-direction gradient buffer `which_out`
// (0: dx, 1: dv) or to ws_out when given.
__global__ void k_project(Dirs D, int which_out, float* __restrict__ ws_out, const float* __restrict__ T,
                          const float* __restrict__ Xh, const float* __restrict__ s, const float* __restrict__ nrm,
                          int B, int N, int C, int P) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)B * C * P) return;
  const int b = (int)(idx / ((size_t)C * P)), j = (int)(idx % P);
  const size_t r = idx % ((size_t)C * P);
  const float val = (T[idx] - Xh[idx] * s[(size_t)b * P + j]) / nrm[(size_t)b * P + j];
  float* o = ws_out ? ws_out + (size_t)b * C * P
                    : (which_out == 0 ? D.d[b / N].dx : D.d[b / N].dv) + (size_t)(b % N) * C * P;
  o[r] = val;
}

__global__ void k_copy_out(Dirs D, const float* __restrict__ src, int B, int N, size_t R) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)B * R) return;
  const int b = (int)(idx / R);
  float* o = D.d[b / N].pred;
  if (o) o[(size_t)(b % N) * R + idx % R] = src[idx];
}

__global__ void k_scale_dq(Dirs D, const float* __restrict__ G, const float* __restrict__ cnt, int B, int N,
                           size_t R, float g0) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)B * R) return;
  const int b = (int)(idx / R);
  D.d[b / N].dx[(size_t)(b % N) * R + idx % R] = g0 / (cnt[b] + 1e-6f) * G[idx];
}

struct Carve {
  char* base;
  size_t off = 0, cap;
  Carve(void* p, size_t c) : base((char*)p), cap(c) {}
  float* f(size_t n) {
    size_t bytes = (n * sizeof(float) + 255) & ~(size_t)255;
    float* r = (float*)(base + off);
    off += bytes;
    return r;
  }
};

}  // namespace

size_t generic_workspace_bytes(const pixpro_problem& pb) {
  const size_t B = (size_t)pb.N * pb.ndir, P = (size_t)pb.H * pb.W, R = (size_t)pb.C * P, Q = P * P;
  auto al = [](size_t n) { return (n * sizeof(float) + 255) & ~(size_t)255; };
  return 6 * al(B * R) + 5 * al(B * P) + 4 * al(B * Q) + al(B * 5 * P) + 2 * al(B) + 4096;
}

// out[b,i,j] = sum_c A[b,c,i] * Bm[b,c,j]
#define GRAM(Ap, Bp, Op)                                                                          \
  do {                                                                                            \
    MM q_{P, P, C, (long)R, (long)R, (long)Q, 1, P, P, 1, P};                                       \
    hipLaunchKernelGGL(k_mm, dim3((P + 63) / 64, (P + 63) / 64, B), dim3(256), 0, st, Ap, Bp, Op, q_); \
    if (int e = check_launch("k_mm(gram)")) return e;                                             \
  } while (0)
// trans == 0: out[b,c,o] = sum_s In[b,c,s] * Mt[b,o,s];  trans == 1: ... * Mt[b,s,o]
#define APPLY(Inp, Mtp, Op, trans)                                                                \
  do {                                                                                            \
    MM q_{C, P, P, (long)R, (long)Q, (long)R, P, 1, (trans) ? P : 1, (trans) ? 1 : P, P};           \
    hipLaunchKernelGGL(k_mm, dim3((P + 63) / 64, (C + 63) / 64, B), dim3(256), 0, st, Inp, Mtp, Op, q_); \
    if (int e = check_launch("k_mm(apply)")) return e;                                            \
  } while (0)

#define LAUNCH(kern, n, ...)                                                          \
  do {                                                                                \
    hipLaunchKernelGGL(kern, dim3(blocks_for(n)), dim3(TPB), 0, st, __VA_ARGS__);      \
    if (int e = check_launch(#kern)) return e;                                        \
  } while (0)
#define LAUNCH_G(kern, ...)                                                                    \
  do {                                                                                         \
    hipLaunchKernelGGL(kern, dim3((P + 63) / 64, B), dim3(TPB), 0, st, __VA_ARGS__);            \
    if (int e = check_launch(#kern)) return e;                                                 \
  } while (0)
#define LAUNCH_B(kern, nb, ...)                                                       \
  do {                                                                                \
    hipLaunchKernelGGL(kern, dim3(nb), dim3(TPB), 0, st, __VA_ARGS__);                 \
    if (int e = check_launch(#kern)) return e;                                        \
  } while (0)

// mode: 0 = fused semantics (x,v,k -> loss, pred, dx, dv); 1 = regression_loss only
// (x is q, already the prediction; dx = dq)
int run_generic(const pixpro_problem& pb, const pixpro_dir* dirs, float* loss_n, float* pos_num, float* pos_mean,
                float* loss, void* workspace, size_t ws_bytes, hipStream_t st, int mode) {
  if (ws_bytes < generic_workspace_bytes(pb)) {
    set_error("workspace too small: %zu < %zu", ws_bytes, generic_workspace_bytes(pb));
    return PIXPRO_ERR_WORKSPACE;
  }
  Dirs D;
  for (int d = 0; d < PIXPRO_MAX_DIRS; ++d) D.d[d] = dirs[d < pb.ndir ? d : 0];
  const int B = pb.N * pb.ndir, N = pb.N, C = pb.C, P = pb.H * pb.W;
  const size_t R = (size_t)C * P, Q = (size_t)P * P;
  Carve w(workspace, ws_bytes);
  float *xh = w.f(B * R), *vh = w.f(B * R), *kh = w.f(B * R), *Yh = w.f(B * R), *G = w.f(B * R), *T = w.f(B * R);
  float *nx = w.f((size_t)B * P), *nv = w.f((size_t)B * P), *ny = w.f((size_t)B * P), *tcol = w.f((size_t)B * P),
        *scol = w.f((size_t)B * P);
  float *A = w.f(B * Q), *Dm = w.f(B * Q), *M = w.f(B * Q), *dS = w.f(B * Q);
  float* pts = w.f((size_t)B * 5 * P);
  float *cnt = w.f(B), *diag = w.f(B);
  const bool want_grad = dirs[0].dx != nullptr;
  const float g0 = pb.grad_scale * -2.f / (float)N;

  // positive-pair mask and counts
  LAUNCH(k_points, (size_t)B * P, pb, D, pts, diag, B);
  LAUNCH(k_mask, B * Q, pts, diag, M, B, P, pb.pos_ratio);
  LAUNCH_B(k_count, B, M, cnt, pos_num, pos_mean, P);
  LAUNCH_G(k_normalize, D, SRC_K, nullptr, kh, nullptr, B, N, C, P, mode == 0 ? pb.normalize_k : 0);

  if (mode == 1) {
    // q is given: loss_n = <q, K M^T>/den ; dq = g0/den * K M^T
    LAUNCH_G(k_normalize, D, SRC_X, nullptr, Yh, nullptr, B, N, C, P, 0);
    APPLY(kh, M, G, 0);
    LAUNCH_G(k_coldot, Yh, G, tcol, B, C, P);
    LAUNCH_B(k_loss_n, B, tcol, cnt, loss_n, P);
    LAUNCH_B(k_finalize, 1, loss_n, loss, B, N);
    if (want_grad) LAUNCH(k_scale_dq, B * R, D, G, cnt, B, N, R, g0);
    return PIXPRO_OK;
  }

  // forward: PPM
  LAUNCH_G(k_normalize, D, SRC_X, nullptr, xh, nx, B, N, C, P, 1);
  LAUNCH_G(k_normalize, D, SRC_V, nullptr, vh, nv, B, N, C, P, 1);
  GRAM(xh, xh, dS);                     // S (in dS buffer)
  LAUNCH(k_powmap, B * Q, dS, A, Dm, B * Q, pb.p, pb.clamp);
  APPLY(vh, A, T, 0);                   // Y = vh * A^T
  LAUNCH_G(k_normalize, D, SRC_X, T, Yh, ny, B, N, C, P, 1);  // Yh, ny
  LAUNCH(k_copy_out, B * R, D, Yh, B, N, R);
  // loss
  APPLY(kh, M, G, 0);                   // G = kh * M^T
  LAUNCH_G(k_coldot, Yh, G, tcol, B, C, P);
  LAUNCH_B(k_loss_n, B, tcol, cnt, loss_n, P);
  LAUNCH_B(k_finalize, 1, loss_n, loss, B, N);
  if (!want_grad) return PIXPRO_OK;
  // backward
  LAUNCH(k_dy, B * R, G, Yh, tcol, ny, cnt, B, C, P, g0);         // G <- dY
  APPLY(G, A, T, 1);                    // dVh = dY * A
  LAUNCH_G(k_coldot, vh, T, scol, B, C, P);
  LAUNCH(k_project, B * R, D, 1, nullptr, T, vh, scol, nv, B, N, C, P);   // dv
  GRAM(G, vh, M);                       // dA (in M buffer)
  hipLaunchKernelGGL(k_symm, dim3((P + 31) / 32, (P + 31) / 32, B), dim3(256), 0, st, M, Dm, dS, B, P);
  if (int e = check_launch("k_symm")) return e;
  APPLY(xh, dS, T, 1);                  // dXh = xh * dSs
  LAUNCH_G(k_coldot, xh, T, scol, B, C, P);
  LAUNCH(k_project, B * R, D, 0, nullptr, T, xh, scol, nx, B, N, C, P);   // dx
  return PIXPRO_OK;
}

// featprop alone.  fwd: Y = vh * A^T.  bwd: given dY -> dx, dv.
int run_featprop(const pixpro_problem& pb, const float* x, const float* v, const float* dY, float* Y, float* dx,
                 float* dv, void* workspace, size_t ws_bytes, hipStream_t st) {
  pixpro_problem q = pb;
  q.ndir = 1;
  if (ws_bytes < generic_workspace_bytes(q)) {
    set_error("workspace too small: %zu < %zu", ws_bytes, generic_workspace_bytes(q));
    return PIXPRO_ERR_WORKSPACE;
  }
  Dirs D = {};
  D.d[0].x = x; D.d[0].v = v; D.d[0].k = x; D.d[0].dx = dx; D.d[0].dv = dv; D.d[0].pred = Y;
  D.d[1] = D.d[0];
  const int B = pb.N, N = pb.N, C = pb.C, P = pb.H * pb.W;
  const size_t R = (size_t)C * P, Q = (size_t)P * P;
  Carve w(workspace, ws_bytes);
  float *xh = w.f(B * R), *vh = w.f(B * R), *kh = w.f(B * R), *Yh = w.f(B * R), *G = w.f(B * R), *T = w.f(B * R);
  float *nx = w.f((size_t)B * P), *nv = w.f((size_t)B * P), *ny = w.f((size_t)B * P), *tcol = w.f((size_t)B * P),
        *scol = w.f((size_t)B * P);
  float *A = w.f(B * Q), *Dm = w.f(B * Q), *M = w.f(B * Q), *dS = w.f(B * Q);
  (void)kh; (void)Yh; (void)ny; (void)tcol;
  LAUNCH_G(k_normalize, D, SRC_X, nullptr, xh, nx, B, N, C, P, 1);
  LAUNCH_G(k_normalize, D, SRC_V, nullptr, vh, nv, B, N, C, P, 1);
  GRAM(xh, xh, dS);
  LAUNCH(k_powmap, B * Q, dS, A, Dm, B * Q, pb.p, pb.clamp);
  if (!dY) {
    APPLY(vh, A, T, 0);
    LAUNCH(k_copy_out, B * R, D, T, B, N, R);
    return PIXPRO_OK;
  }
  if (hipMemcpyAsync(G, dY, B * R * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("hipMemcpyAsync(dY) failed");
    return PIXPRO_ERR_LAUNCH;
  }
  APPLY(G, A, T, 1);
  LAUNCH_G(k_coldot, vh, T, scol, B, C, P);
  LAUNCH(k_project, B * R, D, 1, nullptr, T, vh, scol, nv, B, N, C, P);
  GRAM(G, vh, M);
  hipLaunchKernelGGL(k_symm, dim3((P + 31) / 32, (P + 31) / 32, B), dim3(256), 0, st, M, Dm, dS, B, P);
  if (int e = check_launch("k_symm")) return e;
  APPLY(xh, dS, T, 1);
  LAUNCH_G(k_coldot, xh, T, scol, B, C, P);
  LAUNCH(k_project, B * R, D, 0, nullptr, T, xh, scol, nx, B, N, C, P);
  return PIXPRO_OK;
}

}  // namespace pixpro

namespace pixpro {
int run_finalize(const float* loss_n, float* loss, int B, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(TPB), 0, st, loss_n, loss, B, N);
  return check_launch("k_finalize");
}
}  // namespace pixpro
