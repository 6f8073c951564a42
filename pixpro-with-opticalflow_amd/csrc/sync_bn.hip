// Training-mode (Sync)BatchNorm over NHWC fp32 activations with the ReLU and the residual add of the ResNet blocks
// folded in.  The reference converts every BN of the encoders / projectors to
// torch.nn.SyncBatchNorm (contrast/models/PixPro.py:289-292); with more than one rank that module leaves MIOpen and
// runs ATen's native kernels (torch/nn/modules/_functions.py: batch_norm_stats -> all_gather ->
// batch_norm_gather_stats_with_counts -> batch_norm_elemt; backward_reduce -> all_reduce -> backward_elemt), which
// on MI355X cost 188 ms per N=256 step against 65 ms for MIOpen's single-rank kernels
// (profiles/r01_syncbn_microbench.txt).  These kernels keep the same split (local statistics | exchange | apply),
// so the collectives stay where torch has them, and stream x as [M = N*H*W rows][C] with 16 B per lane:
//
//   forward   k_stats (1 read of x) -> k_stats_finalize (or, PIXPRO_BN_FOLD=1, the last work-group per channel chunk)
//             -> [all_gather of 2C+1 floats] -> k_apply (merges the ranks' records in its prologue)
//             k_apply: y = relu?( (x-mean)*invstd*w + b (+ residual) )                      (read x [,res], write y)
//   backward  k_bwd_reduce: dz = dy * (y > 0)?; sum dz, sum dz*(x-mean) -> k_bwd_finalize      (read dy, x [,y]; same option)
//             -> [all_reduce of 2C floats] -> k_bwd_apply: dx [, dres = dz]                   (read dy, x [,y])
//   ReLU decisions behind a residual add (the last BatchNorm of a ResNet unit) cannot be recomputed from x alone; instead of
//   reading the saved output y twice in backward (two passes over the widest tensors of the net), k_apply can leave them
//   as a BIT RECORD - 1/32 of the tensor - which both backward kernels read (relu_bits, MODE 3).
//
// Statistics are accumulated as shifted sums (pivot = the channel's first element), so the variance does not suffer
// the E[x^2]-E[x]^2 cancellation; partial sums are combined in a fixed order (deterministic; the only atomic is the ticket
// that tells a work-group it is the last of its channel chunk to finish - which one that is does not enter the result).
// Round 4: folding the partials in the producing kernel (the work-group that draws the last ticket of its channel chunk
// does it before it exits) was built to remove the 324 small finalize launches of a step - and measured slower than they
// are (bn_fold_in_kernel below); it stays selectable, the launches stay the default and read eight rows per thread at once.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace pixpro {
namespace {

constexpr int BN_TPB = 256;

struct BnTile {
  int TX, TY;        // lanes along channels (float4 units) x rows per block iteration; TX * TY = 256
  int gx, gy;        // grid: channel chunks x row blocks
};

// C4 = C/4.  TX = largest power of two <= 64 dividing C4; `cap` = work-groups per launch.  Measured on MI355X over the
// ResNet-50 layers (N=256): the read-only statistics kernel streams best with ~1 group per CU, the read+write kernels with
// ~2; the backward reduction with ~1 and eight rows of both tensors in flight per thread (round 4; ~4 groups with four rows before).
constexpr int CAP_MAX = 1024;
struct BnCaps { int stats, apply, bwd_reduce, bwd_apply; };
// PIXPRO_BN_CAPS="stats,apply,bwd_reduce,bwd_apply" overrides the defaults (tuning knob, tools/bn_sweep.py)
inline BnCaps bn_caps() {
  static const BnCaps caps = [] {
    BnCaps c{384, 512, 256, 512};
    if (const char* e = getenv("PIXPRO_BN_CAPS")) {
      int a, b, d, f;
      if (sscanf(e, "%d,%d,%d,%d", &a, &b, &d, &f) == 4) {
        auto cl = [](int v) { return std::min(std::max(v, 1), CAP_MAX); };
        auto cw = [](int v) { return std::min(std::max(v, 1), 65535); };      // no workspace behind the element-wise kernels
        c = BnCaps{cl(a), cw(b), cl(d), cw(f)};
      }
    }
    return c;
  }();
  return caps;
}
// Measured on MI355X, ResNet-50 step at N = 256 (profiles/r04_bn_fold_ab.txt): in-kernel fold 67.3 ms of BatchNorm per step
// (bn_stats 13.1, bn_backward_reduce 18.8 ms) against 57.7 ms with the two small launches (10.5 / 11.8 ms), 1134 against
// 1173 images/s: the one agent-scope release every work-group needs before it may draw its ticket costs the streaming kernels
// more than the launches cost.  So the launches stay the default; PIXPRO_BN_FOLD=1 selects the in-kernel fold.
inline bool bn_fold_in_kernel() {
  const char* e = getenv("PIXPRO_BN_FOLD");      // read per call: tests switch it at run time
  return e && e[0] == '1';
}
inline BnTile bn_tile(int64_t M, int C4, int cap) {
  BnTile t;
  t.TX = 1;
  while (t.TX < 64 && C4 % (t.TX * 2) == 0) t.TX *= 2;
  t.TY = BN_TPB / t.TX;
  t.gx = C4 / t.TX;
  int64_t want = std::max<int64_t>(1, cap / t.gx);
  int64_t most = std::max<int64_t>(1, (M + (int64_t)t.TY * 4 - 1) / ((int64_t)t.TY * 4));   // >= 4 rows per thread
  t.gy = (int)std::min<int64_t>(std::min(want, most), 65535);
  return t;
}

__device__ __forceinline__ float4 ld4(const float* p, int64_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ void st4(float* p, int64_t i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
// streaming (non-temporal) forms for tensors that are swept once: global_load_dwordx4 / global_store_dwordx4 ... nt
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p, int64_t i4) {
  const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p) + i4);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4nt(float* p, int64_t i4, float4 v) {
  v4f_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<v4f_t*>(p) + i4);
}
// NT (template flag of the sweeping kernels): the activation loads use the streaming form.  Measured per ResNet-50 shape at
// N = 256 (tools/bn_shapes.py, profiles/r04_bn_nontemporal.txt): the read-only sweeps gain 25 % on tensors the Infinity Cache
// cannot hold (statistics of an 822 MB tensor 170 -> 123 us, 6.7 TB/s), the element-wise kernels must then stream too (a
// streaming statistics pass leaves nothing cached for them); tensors of <= 51 MB lose (they live in the 256 MB Infinity
// Cache between the convolution that wrote them and these kernels).  Stores stay plain: the next convolution reads them.
#define LDX(NT, p, i) ((NT) ? ld4nt(p, i) : ld4(p, i))
struct BnStream { bool stats, apply, bwd_reduce, bwd_apply; };
inline BnStream bn_stream_policy(int64_t M, int C, bool writes_dres) {
  const char* env = getenv("PIXPRO_BN_STREAM");                // 0 / 1 force the choice (read per call: tests switch it)
  const int mode = env && env[0] ? atoi(env) : -1;
  if (mode == 0) return BnStream{false, false, false, false};
  if (mode == 1) return BnStream{true, true, true, true};
  const double mb = (double)M * C * 4 / 1e6;
  BnStream p;
  p.stats = p.apply = mb >= 80.0;
  p.bwd_reduce = mb >= 150.0;
  // the input-gradient kernel: streaming loads pay up to ~600 MB and wherever a second tensor is written (measured)
  p.bwd_apply = mb >= 150.0 && (mb < 600.0 || writes_dres);
  return p;
}

// Row walk of a work-group: the four rows a thread has in flight are gridDim.y * TY rows apart.  (Four ADJACENT row blocks
// per thread - 16 KB contiguous per work-group and iteration at C = 256 - measure the same: 57.6 against 58.0 ms of
// BatchNorm per step at their best work-group counts, round 4.)
#define BN_WALK                                                                         \
  const int64_t step = (int64_t)gridDim.y * TY, jump = 4 * step;                        \
  int64_t r = (int64_t)blockIdx.y * TY + ty;
#define BN_ROWS(body)                                                                   \
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;                         \
  const int c4 = blockIdx.x * TX + tx;                                                  \
  BN_WALK                                                                               \
  for (; r + 3 * step < M; r += jump) {                                                 \
    body(r) body(r + step) body(r + 2 * step) body(r + 3 * step)                        \
  }                                                                                     \
  for (int k_ = 0; k_ < 4 && r + k_ * step < M; ++k_) { body(r + k_ * step) }

// The same walk with the loads of the four rows issued together, in front of a scheduling barrier: left to itself hipcc
// emits the backward kernels as load - wait - use - store per row (one row in flight per thread, and with the bit record a
// second dependent round trip per row).  LOAD(k, row) declares what row slot k needs, USE(k, row) consumes it.
#define BN_ROWS_LU(LOAD, USE)                                                           \
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;                         \
  const int c4 = blockIdx.x * TX + tx;                                                  \
  BN_WALK                                                                               \
  for (; r + 3 * step < M; r += jump) {                                                 \
    LOAD(0, r) LOAD(1, r + step) LOAD(2, r + 2 * step) LOAD(3, r + 3 * step)            \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    USE(0, r) USE(1, r + step) USE(2, r + 2 * step) USE(3, r + 3 * step)                \
  }                                                                                     \
  for (int k_ = 0; k_ < 4 && r + k_ * step < M; ++k_) { LOAD(0, r + k_ * step) USE(0, r + k_ * step) }

#define BN_ROWS_LU8(LOAD, USE)                                                          \
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;                         \
  const int c4 = blockIdx.x * TX + tx;                                                  \
  const int64_t step = (int64_t)gridDim.y * TY;                                         \
  int64_t r = (int64_t)blockIdx.y * TY + ty;                                            \
  for (; r + 7 * step < M; r += 8 * step) {                                             \
    LOAD(0, r) LOAD(1, r + step) LOAD(2, r + 2 * step) LOAD(3, r + 3 * step)            \
    LOAD(4, r + 4 * step) LOAD(5, r + 5 * step) LOAD(6, r + 6 * step) LOAD(7, r + 7 * step) \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    USE(0, r) USE(1, r + step) USE(2, r + 2 * step) USE(3, r + 3 * step)                \
    USE(4, r + 4 * step) USE(5, r + 5 * step) USE(6, r + 6 * step) USE(7, r + 7 * step) \
  }                                                                                     \
  for (; r < M; r += step) { LOAD(0, r) USE(0, r) }

// reduce two float4 per thread over the TY rows of the block (fixed order) -> out0/out1 of the ty == 0 lanes
__device__ __forceinline__ void block_rows_sum(float4& a, float4& b, int TX, int TY, float4* sh) {
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;
  sh[threadIdx.x] = a;
  sh[BN_TPB + threadIdx.x] = b;
  __syncthreads();
  if (ty == 0) {
    for (int k = 1; k < TY; ++k) {
      const float4 u = sh[k * TX + tx], v = sh[BN_TPB + k * TX + tx];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
  }
}

// ---- last-work-group fold -------------------------------------------------------------------------------------------
// The workspace starts with BN_HDR int32 tickets (one per channel chunk = blockIdx.x), zero before the launch; every
// work-group publishes its partial rows (release), draws a ticket, and the one that draws gridDim.y - 1 sums the gridDim.y
// partial rows of its chunk in double (the partials are long sums themselves), in a fixed order, and puts the ticket back
// to zero for the next launch on the stream.  Returns true in the threads (ty == 0) that hold the totals of float4 column
// c4: s = first partial plane, q = second.
constexpr int BN_HDR = 1024;           // tickets: channel chunks per launch (C / 4 / TX <= 1024)
struct D4 { double x, y, z, w; };
__device__ __forceinline__ void d4_add(D4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
__device__ __forceinline__ void d4_add(D4& a, const D4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
__device__ __forceinline__ bool last_block_fold(int* __restrict__ tickets, const float* __restrict__ partial, int C4, int TX,
                                                int TY, D4& s, D4& q) {
  __shared__ int is_last;
  __shared__ D4 red[2][BN_TPB];
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX, c4 = blockIdx.x * TX + tx;
  const int nblk = gridDim.y;
  const int64_t C = (int64_t)C4 * 4;
  if (!tickets) return false;          // the caller folds with a launch of its own
  if (nblk > 1) {
    // ONE agent-scope release per work-group, by one lane, once every wave's partial stores have reached L2 - and one
    // acquire in the work-group that turns out to be last.  (A __threadfence() in every thread, the first version of
    // this, made bn_stats twice and bn_backward_reduce three times slower: 86 / 289 us per call instead of 45 / 94.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = t == nblk - 1;
      if (t == nblk - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (!is_last) return false;
  }
  s = D4{0.0, 0.0, 0.0, 0.0}; q = s;
  for (int b = ty; b < nblk; b += TY) {
    d4_add(s, ld4(partial + (int64_t)b * 2 * C, c4));
    d4_add(q, ld4(partial + (int64_t)b * 2 * C + C, c4));
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x == 0 && nblk > 1) __hip_atomic_store(tickets + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ty != 0) return false;
  for (int k = 1; k < TY; ++k) { d4_add(s, red[0][k * TX + tx]); d4_add(q, red[1][k * TX + tx]); }
  return true;
}

// partial[by][0][C] = sum (x - K), partial[by][1][C] = sum (x - K)^2,  K = x[0][c]; the last work-group of a chunk:
// packed[0..C) = mean, packed[C..2C) = biased variance, packed[2C] = M     (what one rank contributes)
template <bool NT>
__global__ void __launch_bounds__(BN_TPB)
k_stats(const float* __restrict__ x, int* __restrict__ tickets, float* __restrict__ partial, float* __restrict__ packed,
        int64_t M, int C4, int TX, int TY) {
  __shared__ float4 sh[2 * BN_TPB];
  float4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  const float4 K = ld4(x, blockIdx.x * TX + (threadIdx.x & (TX - 1)));
#define BODY(row) { const float4 v = LDX(NT, x, (row) * C4 + c4);                                                 \
    const float dx_ = v.x - K.x, dy_ = v.y - K.y, dz_ = v.z - K.z, dw_ = v.w - K.w;                            \
    s.x += dx_; s.y += dy_; s.z += dz_; s.w += dw_;                                                           \
    q.x = __builtin_fmaf(dx_, dx_, q.x); q.y = __builtin_fmaf(dy_, dy_, q.y);                                  \
    q.z = __builtin_fmaf(dz_, dz_, q.z); q.w = __builtin_fmaf(dw_, dw_, q.w); }
  BN_ROWS(BODY)
#undef BODY
  block_rows_sum(s, q, TX, TY, sh);
  const int64_t C = (int64_t)C4 * 4;
  if (ty == 0) {
    float* p = partial + (int64_t)blockIdx.y * 2 * C;
    st4(p, c4, s);
    st4(p + C, c4, q);
  }
  D4 ds, dq;
  if (!last_block_fold(tickets, partial, C4, TX, TY, ds, dq)) return;
  const double n = (double)M;
  const double sv[4] = {ds.x, ds.y, ds.z, ds.w}, qv[4] = {dq.x, dq.y, dq.z, dq.w};
  const float kv[4] = {K.x, K.y, K.z, K.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {        // scalar stores: a view's record [2C + 1] need not start on a 16-B boundary
    const double ms = sv[k] / n;
    packed[4 * c4 + k] = (float)((double)kv[k] + ms);
    packed[C + 4 * c4 + k] = (float)fmax(qv[k] / n - ms * ms, 0.0);
  }
  if (c4 == 0) packed[2 * C] = (float)M;
}

// ---- the same fold as launches of their own: the default (see bn_fold_in_kernel) ----
// Column sums of the [nblk][2][C] partials: 8 channels x 32 row groups per block, double accumulation (the partials are
// long sums themselves), fixed order.  Returns the two totals of channel c in the g == 0 lanes.
constexpr int FIN_CH = 8, FIN_G = BN_TPB / FIN_CH;
__device__ __forceinline__ bool finalize_sums(const float* __restrict__ partial, int nblk, int C, int& c, double& s,
                                              double& q) {
  __shared__ double red[2][FIN_G][FIN_CH];
  const int lc = threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
  c = blockIdx.x * FIN_CH + lc;
  s = 0.0; q = 0.0;
  if (c < C) {
    // eight rows in flight per thread (round 4: one dependent round trip per row made this one-block-per-8-channels
    // launch cost 21 us per layer; the summation order - rows g, g + 32, ... in turn - is unchanged)
    constexpr int U = 8;
    int b = g;
    for (; b + (U - 1) * FIN_G < nblk; b += U * FIN_G) {
      float sv[U], qv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        sv[u] = partial[(int64_t)(b + u * FIN_G) * 2 * C + c];
        qv[u] = partial[(int64_t)(b + u * FIN_G) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { s += (double)sv[u]; q += (double)qv[u]; }
    }
    for (; b < nblk; b += FIN_G) {
      s += (double)partial[(int64_t)b * 2 * C + c];
      q += (double)partial[(int64_t)b * 2 * C + C + c];
    }
  }
  red[0][g][lc] = s;
  red[1][g][lc] = q;
  __syncthreads();
  if (g != 0 || c >= C) return false;
  for (int k = 1; k < FIN_G; ++k) { s += red[0][k][lc]; q += red[1][k][lc]; }
  return true;
}

// packed[0..C) = mean, packed[C..2C) = biased variance, packed[2C] = M     (what one rank contributes)
__global__ void __launch_bounds__(BN_TPB)
k_stats_finalize(const float* __restrict__ x, const float* __restrict__ partial, int nblk, int64_t M, int C,
                 float* __restrict__ packed) {
  int c;
  double s, q;
  if (!finalize_sums(partial, nblk, C, c, s, q)) return;
  const double n = (double)M;
  const double ms = s / n;
  packed[c] = (float)((double)x[c] + ms);
  packed[C + c] = (float)fmax(q / n - ms * ms, 0.0);
  if (c == 0) packed[2 * C] = (float)M;
}

// sums[0..C) = sum dz, sums[C..2C) = sum dz (x - mean)   (this rank); grad_weight = sums[C+c] * invstd, grad_bias
__global__ void __launch_bounds__(BN_TPB)
k_bwd_finalize(const float* __restrict__ partial, int nblk, int C, const float* __restrict__ invstd,
               float* __restrict__ sums, float* __restrict__ grad_weight, float* __restrict__ grad_bias) {
  int c;
  double sd, qd;
  if (!finalize_sums(partial, nblk, C, c, sd, qd)) return;
  const float s = (float)sd, q = (float)qd;
  sums[c] = s;
  sums[C + c] = q;
  if (grad_weight) grad_weight[c] = q * invstd[c];
  if (grad_bias) grad_bias[c] = s;
}

// Count-weighted merge of the ranks' statistics for one channel (batch_norm_gather_stats_with_counts), in double.
struct BnMerged {
  float mean, invstd, scale, shift, var_unbiased;
};
__device__ __forceinline__ BnMerged merge_channel(const float* __restrict__ gathered, int W, int C, int c,
                                                  const float* __restrict__ weight, const float* __restrict__ bias,
                                                  float eps) {
  const int64_t S = 2 * (int64_t)C + 1;
  double n = 0.0, md = 0.0;
  for (int r = 0; r < W; ++r) {
    const double nr = (double)gathered[r * S + 2 * C];
    n += nr;
    md += nr * (double)gathered[r * S + c];
  }
  md = md / n;
  double v = 0.0;
  for (int r = 0; r < W; ++r) {
    const double nr = (double)gathered[r * S + 2 * C], d = (double)gathered[r * S + c] - md;
    v += nr * ((double)gathered[r * S + C + c] + d * d);
  }
  BnMerged o;
  o.mean = (float)md;
  o.invstd = 1.f / sqrtf((float)(v / n) + eps);
  o.scale = (weight ? weight[c] : 1.f) * o.invstd;
  o.shift = (bias ? bias[c] : 0.f) - o.mean * o.scale;
  o.var_unbiased = (float)(v / fmax(n - 1.0, 1.0));
  return o;
}

// gathered [W][2C+1] -> y = relu?(x * scale + shift (+ residual)).  Every work-group merges the statistics of its own
// channels in its prologue (a few dozen flops per lane: cheaper than a separate launch per layer); the first row block
// also publishes mean, invstd (saved for backward), scale/shift, and updates the running statistics (unbiased variance).
// BITS (needs TX == 64 and C4 % 64 == 0, so that a wave holds the 64 consecutive float4 i0 .. i0 + 63 of one row, i0 % 64 == 0):
// the ReLU decisions of those 256 elements leave as one 32-byte record, bits[(i0 / 64) * 4 + k] bit l = component k of
// float4 i0 + l is positive.  v_cmp produces the four words as they are; lanes 0..3 store them.
template <bool RELU, bool RES, bool BITS, bool NT>
__global__ void __launch_bounds__(BN_TPB)
k_apply(const float* __restrict__ gathered, int W, const float* __restrict__ weight, const float* __restrict__ bias,
        float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
        float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale_shift,
        const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
        unsigned long long* __restrict__ bits, int64_t M, int C4, int TX, int TY) {
  const int cc = blockIdx.x * TX + (threadIdx.x & (TX - 1));
  const int C = C4 * 4;
  const bool publish = blockIdx.y == 0 && threadIdx.x / TX == 0;
  float scv[4], sfv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * cc + k;
    const BnMerged mg = merge_channel(gathered, W, C, c, weight, bias, eps);
    scv[k] = mg.scale; sfv[k] = mg.shift;
    if (publish) {
      mean_out[c] = mg.mean;
      invstd_out[c] = mg.invstd;
      scale_shift[c] = mg.scale;
      scale_shift[C + c] = mg.shift;
      if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mg.mean;
      if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * mg.var_unbiased;
    }
  }
  const float4 sc = make_float4(scv[0], scv[1], scv[2], scv[3]), sf = make_float4(sfv[0], sfv[1], sfv[2], sfv[3]);
#define BODY(row) { const int64_t i = (row) * C4 + c4; const float4 v = LDX(NT, x, i); float4 o;                 \
    o.x = __builtin_fmaf(v.x, sc.x, sf.x); o.y = __builtin_fmaf(v.y, sc.y, sf.y);                              \
    o.z = __builtin_fmaf(v.z, sc.z, sf.z); o.w = __builtin_fmaf(v.w, sc.w, sf.w);                              \
    if (RES) { const float4 e = LDX(NT, res, i); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }                 \
    if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }  \
    st4(y, i, o);                                                                                             \
    if (BITS) { const unsigned long long b0 = __ballot(o.x > 0.f), b1 = __ballot(o.y > 0.f),                   \
                                         b2 = __ballot(o.z > 0.f), b3 = __ballot(o.w > 0.f);                   \
                if (tx < 4) bits[(i >> 6) * 4 + tx] = tx == 0 ? b0 : tx == 1 ? b1 : tx == 2 ? b2 : b3; } }
  BN_ROWS(BODY)
#undef BODY
}

// ReLU decision of the forward pass for one float4: MODE 1 reads the saved output y, MODE 2 recomputes it from x with
// the forward's own scale / shift and the same fma (no residual in that case), MODE 3 reads k_apply's bit record (`y`
// then points at it; the record of a wave's 64 float4 is 32 bytes at one address: a broadcast load), MODE 0: no ReLU.
// BN_MASK_LOAD issues what the decision needs from memory (slot k), BN_MASK_APPLY masks g.
#define BN_MASK_LOAD(MODE, k, i)                                                                               \
    float4 o##k = {0.f, 0.f, 0.f, 0.f}; uint4 p##k = {0u, 0u, 0u, 0u}, q##k = p##k;                             \
    if (MODE == 1) o##k = ld4(y, i);                                                                           \
    if (MODE == 3) { const uint4* bp_ = reinterpret_cast<const uint4*>(y) + ((i) >> 6) * 2;                     \
                     p##k = bp_[0]; q##k = bp_[1]; }
#define BN_MASK_APPLY(MODE, k, g, v)                                                                           \
    if (MODE == 1) { g.x = o##k.x > 0.f ? g.x : 0.f; g.y = o##k.y > 0.f ? g.y : 0.f;                            \
                     g.z = o##k.z > 0.f ? g.z : 0.f; g.w = o##k.w > 0.f ? g.w : 0.f; }                         \
    if (MODE == 2) { g.x = __builtin_fmaf(v.x, sc.x, sf.x) > 0.f ? g.x : 0.f;                                  \
                     g.y = __builtin_fmaf(v.y, sc.y, sf.y) > 0.f ? g.y : 0.f;                                  \
                     g.z = __builtin_fmaf(v.z, sc.z, sf.z) > 0.f ? g.z : 0.f;                                  \
                     g.w = __builtin_fmaf(v.w, sc.w, sf.w) > 0.f ? g.w : 0.f; }                                \
    if (MODE == 3) { const int sh_ = tx & 31; const bool hi_ = tx >= 32;                                       \
                     g.x = (((hi_ ? p##k.y : p##k.x) >> sh_) & 1u) ? g.x : 0.f;                                 \
                     g.y = (((hi_ ? p##k.w : p##k.z) >> sh_) & 1u) ? g.y : 0.f;                                 \
                     g.z = (((hi_ ? q##k.y : q##k.x) >> sh_) & 1u) ? g.z : 0.f;                                 \
                     g.w = (((hi_ ? q##k.w : q##k.z) >> sh_) & 1u) ? g.w : 0.f; }

// partial[by][0][C] = sum dz, partial[by][1][C] = sum dz * (x - mean);  dz = dy masked by the forward's ReLU decision
template <int MODE, bool NT>
__global__ void __launch_bounds__(BN_TPB)
k_bwd_reduce(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
             const float* __restrict__ scale_shift, const float* __restrict__ mean, const float* __restrict__ invstd,
             int* __restrict__ tickets, float* __restrict__ partial, float* __restrict__ sums, float* __restrict__ grad_weight,
             float* __restrict__ grad_bias, int64_t M, int C4, int TX, int TY) {
  __shared__ float4 sh[2 * BN_TPB];
  float4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
  const int cc = blockIdx.x * TX + (threadIdx.x & (TX - 1));
  const float4 mu = ld4(mean, cc);
  float4 sc = mu, sf = mu;
  if (MODE == 2) { sc = ld4(scale_shift, cc); sf = ld4(scale_shift + (int64_t)C4 * 4, cc); }
#define LOAD(k, row) const int64_t i##k = (row) * C4 + c4; float4 g##k = LDX(NT, dy, i##k); const float4 v##k = LDX(NT, x, i##k); \
    BN_MASK_LOAD(MODE, k, i##k)
#define USE(k, row) { BN_MASK_APPLY(MODE, k, g##k, v##k)                                                        \
    s.x += g##k.x; s.y += g##k.y; s.z += g##k.z; s.w += g##k.w;                                                \
    q.x = __builtin_fmaf(g##k.x, v##k.x - mu.x, q.x); q.y = __builtin_fmaf(g##k.y, v##k.y - mu.y, q.y);        \
    q.z = __builtin_fmaf(g##k.z, v##k.z - mu.z, q.z); q.w = __builtin_fmaf(g##k.w, v##k.w - mu.w, q.w); }
  BN_ROWS_LU8(LOAD, USE)       // eight rows of both tensors in flight, 256 work-groups: 8.6 against 9.6 ms per step (4 rows, 1024)
#undef LOAD
#undef USE
  block_rows_sum(s, q, TX, TY, sh);
  const int64_t C = (int64_t)C4 * 4;
  if (ty == 0) {
    float* p = partial + (int64_t)blockIdx.y * 2 * C;
    st4(p, c4, s);
    st4(p + C, c4, q);
  }
  // the last work-group of the chunk: sums[0..C) = sum dz, sums[C..2C) = sum dz (x - mean)  (this rank);
  // grad_weight = sums[C + c] * invstd[c], grad_bias = sums[c]
  D4 ds, dq;
  if (!last_block_fold(tickets, partial, C4, TX, TY, ds, dq)) return;
  const float sv[4] = {(float)ds.x, (float)ds.y, (float)ds.z, (float)ds.w};
  const float qv[4] = {(float)dq.x, (float)dq.y, (float)dq.z, (float)dq.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * c4 + k;
    sums[c] = sv[k];
    sums[C + c] = qv[k];
    if (grad_weight) grad_weight[c] = qv[k] * invstd[c];
    if (grad_bias) grad_bias[c] = sv[k];
  }
}

// dx = (dz - sum_dz/n - (x - mean) * invstd^2 * sum_dz_xmu/n) * invstd * w        (batch_norm_backward_elemt)
template <int MODE, bool RES, bool NT>
__global__ void __launch_bounds__(BN_TPB)
k_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
            const float* __restrict__ scale_shift, const float* __restrict__ mean, const float* __restrict__ invstd,
            const float* __restrict__ weight,
            const float* __restrict__ sums, const float* __restrict__ counts, int W, int64_t cstride,
            float* __restrict__ dx, float* __restrict__ dres, int64_t M,
            int C4, int TX, int TY) {
  const int cc = blockIdx.x * TX + (threadIdx.x & (TX - 1));
  const int64_t C = (int64_t)C4 * 4;
  const float4 mu = ld4(mean, cc), is = ld4(invstd, cc);
  const float4 w = weight ? ld4(weight, cc) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sd = ld4(sums, cc), sx = ld4(sums + C, cc);
  float4 sc = mu, sf = mu;
  if (MODE == 2) { sc = ld4(scale_shift, cc); sf = ld4(scale_shift + C, cc); }
  // total rows over all ranks: the counts the ranks exchanged in the forward pass (they may differ between ranks)
  double n_all = 0.0;
  for (int r = 0; r < W; ++r) n_all += (double)counts[r * cstride];
  const float inv_n = (float)(1.0 / n_all);
  // dx = dz * a + (x - mean) * b + c  per channel
  float4 a, b, c;
  a.x = is.x * w.x; a.y = is.y * w.y; a.z = is.z * w.z; a.w = is.w * w.w;
  b.x = -(sx.x * inv_n) * is.x * is.x * a.x; b.y = -(sx.y * inv_n) * is.y * is.y * a.y;
  b.z = -(sx.z * inv_n) * is.z * is.z * a.z; b.w = -(sx.w * inv_n) * is.w * is.w * a.w;
  c.x = -(sd.x * inv_n) * a.x; c.y = -(sd.y * inv_n) * a.y; c.z = -(sd.z * inv_n) * a.z; c.w = -(sd.w * inv_n) * a.w;
#define LOAD(k, row) const int64_t i##k = (row) * C4 + c4; float4 g##k = LDX(NT, dy, i##k); const float4 v##k = LDX(NT, x, i##k); \
    BN_MASK_LOAD(MODE, k, i##k)
#define USE(k, row) { BN_MASK_APPLY(MODE, k, g##k, v##k)                                                        \
    if (RES) st4(dres, i##k, g##k);                                                                           \
    float4 o2;                                                                                                \
    o2.x = __builtin_fmaf(g##k.x, a.x, __builtin_fmaf(v##k.x - mu.x, b.x, c.x));                              \
    o2.y = __builtin_fmaf(g##k.y, a.y, __builtin_fmaf(v##k.y - mu.y, b.y, c.y));                              \
    o2.z = __builtin_fmaf(g##k.z, a.z, __builtin_fmaf(v##k.z - mu.z, b.z, c.z));                              \
    o2.w = __builtin_fmaf(g##k.w, a.w, __builtin_fmaf(v##k.w - mu.w, b.w, c.w));                              \
    st4(dx, i##k, o2); }
  BN_ROWS_LU(LOAD, USE)
#undef LOAD
#undef USE
}

// out = a + dy * bit   (a nullable: out = dy * bit).  The gradient that reaches a residual unit's input is the sum of its
// convolution branch's and of what the unit's last BatchNorm passes to the shortcut, dz = dy masked by the ReLU decisions:
// with the bit record that term need not be written by k_bwd_apply and read back by an add - this kernel reads dy and the
// bits instead (one pass over the widest tensors less per unit).  A wave holds 64 consecutive float4, i0 % 64 == 0.
template <bool NT>
__global__ void __launch_bounds__(BN_TPB)
k_masked_add(const float* __restrict__ a, const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
             int64_t n4) {
  const int tx = threadIdx.x & 63;
  const float4 sc = {0.f, 0.f, 0.f, 0.f}, sf = sc;       // (named by the other modes of BN_MASK_APPLY)
  const int64_t step = (int64_t)gridDim.x * BN_TPB, jump = 4 * step;
  int64_t i = (int64_t)blockIdx.x * BN_TPB + threadIdx.x;
#define LOAD(k, idx) const int64_t i##k = (idx); float4 g##k = LDX(NT, dy, i##k); float4 e##k = {0.f, 0.f, 0.f, 0.f};   \
    if (a) e##k = LDX(NT, a, i##k);                                                                               \
    BN_MASK_LOAD(3, k, i##k)
#define USE(k, idx) { BN_MASK_APPLY(3, k, g##k, g##k)                                                           \
    g##k.x += e##k.x; g##k.y += e##k.y; g##k.z += e##k.z; g##k.w += e##k.w;                                    \
    st4(out, i##k, g##k); }
  for (; i + 3 * step < n4; i += jump) {
    LOAD(0, i) LOAD(1, i + step) LOAD(2, i + 2 * step) LOAD(3, i + 3 * step)
    __builtin_amdgcn_sched_barrier(0);
    USE(0, i) USE(1, i + step) USE(2, i + 2 * step) USE(3, i + 3 * step)
  }
  for (int k_ = 0; k_ < 4 && i + k_ * step < n4; ++k_) { LOAD(0, i + k_ * step) USE(0, i + k_ * step) }
#undef LOAD
#undef USE
}

// ---- the stem: BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max pooling in one pass (contrast/resnet.py: maxpool(relu(bn1(conv1)))) --
// The normalised tensor of the stem (822 MB at N = 256) exists only to be pooled: written by k_apply, read by ATen's pooling
// kernel (which also writes 411 MB of int64 indices), and in backward the pooled gradient is scattered into a full-size
// tensor that the two BatchNorm backward kernels then read.  Here the forward reads x and writes the pooled tensor; the
// backward kernels rebuild, per 16 x 16 tile of positions, which position won each window from an LDS tile of x (first
// maximum in row-major window order, as ATen's index), gather the pooled gradient through that (windows in ascending
// order: the sum ATen's backward forms), mask it by the ReLU decision and go straight on with the BatchNorm sums / the
// input gradient.  C % 32 == 0; one work-group = one tile x 32 channels.
__global__ void __launch_bounds__(BN_TPB)
k_bn_publish(const float* __restrict__ gathered, int W, const float* __restrict__ weight, const float* __restrict__ bias,
             float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
             float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale_shift, int C) {
  const int c = blockIdx.x * BN_TPB + threadIdx.x;
  if (c >= C) return;
  const BnMerged mg = merge_channel(gathered, W, C, c, weight, bias, eps);
  mean_out[c] = mg.mean;
  invstd_out[c] = mg.invstd;
  scale_shift[c] = mg.scale;
  scale_shift[C + c] = mg.shift;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mg.mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * mg.var_unbiased;
}

__device__ __forceinline__ float4 bn_relu4(float4 v, float4 sc, float4 sf) {
  return make_float4(fmaxf(__builtin_fmaf(v.x, sc.x, sf.x), 0.f), fmaxf(__builtin_fmaf(v.y, sc.y, sf.y), 0.f),
                     fmaxf(__builtin_fmaf(v.z, sc.z, sf.z), 0.f), fmaxf(__builtin_fmaf(v.w, sc.w, sf.w), 0.f));
}

// forward: one float4 of channels of one pooled pixel per thread slot; a work-group = 8 x 8 pooled pixels x 16 float4
__global__ void __launch_bounds__(BN_TPB)
k_apply_pool(const float* __restrict__ x, const float* __restrict__ scale_shift, float* __restrict__ p, int H, int W, int PH,
             int PW, int C4) {
  const int c4 = blockIdx.z * 16 + (threadIdx.x & 15);
  if (c4 >= C4) return;
  const int n = blockIdx.y;
  const int tiles_x = (PW + 7) / 8;
  const int ph0 = (blockIdx.x / tiles_x) * 8, pw0 = (blockIdx.x % tiles_x) * 8;
  const float4 sc = ld4(scale_shift, c4), sf = ld4(scale_shift + (int64_t)C4 * 4, c4);
  const float* xn = x + (int64_t)n * H * W * C4 * 4;
  float* pn = p + (int64_t)n * PH * PW * C4 * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = (threadIdx.x >> 4) + 16 * k;
    const int ph = ph0 + (pix >> 3), pw = pw0 + (pix & 7);
    if (ph >= PH || pw >= PW) continue;
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);          // every window holds a valid pixel and relu(.) >= 0
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * ph - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = 2 * pw - 1 + kw;
        if (w < 0 || w >= W) continue;
        const float4 y = bn_relu4(ld4(xn, ((int64_t)h * W + w) * C4 + c4), sc, sf);
        best.x = fmaxf(best.x, y.x); best.y = fmaxf(best.y, y.y); best.z = fmaxf(best.z, y.z); best.w = fmaxf(best.w, y.w);
      }
    }
    st4(pn, ((int64_t)ph * PW + pw) * C4 + c4, best);
  }
}

// The gradient that reaches x's position through pooling and ReLU, for the 16 x 16 positions of one tile and 8 float4 of
// channels: fills `dz[8]` (slot k = position (tid >> 3) + 32 k of the tile) and the matching x values.  LDS: the tile of x
// with a rim (19 x 19 x 8 float4), the pooled gradient and the winners of the 9 x 9 windows that touch the tile.
constexpr int PT = 16, PR = PT + 3, PWN = PT / 2 + 1;        // tile positions, x rows/cols held, windows per side
constexpr int PXL = (PR * PR + BN_TPB / 8 - 1) / (BN_TPB / 8);   // x loads per thread and tile (12)
constexpr int PDL = (PWN * PWN + BN_TPB / 8 - 1) / (BN_TPB / 8); // pooled-gradient loads per thread and tile (3)
struct PoolTile {
  float4 yt[PR][PR][8];
  float4 dpt[PWN][PWN][8];
  uchar4 win[PWN][PWN][8];
};
struct PoolRegs {
  float4 xr[PXL], dr[PDL];
};
// Position (row, column in 0..15) of slot k of thread group g = tid >> 3.  The 256 positions of a tile are dealt by PARITY
// CLASS: slots 2c, 2c + 1 hold the 64 positions with (row & 1, column & 1) = (c >> 1, c & 1), so that in the unrolled
// position loop the number of windows a position belongs to (1, 2, 2 or 4) is a compile-time fact of the slot.
__device__ __forceinline__ void pool_pos(int g, int k, int& r, int& c) {
  const int idx = g + 32 * (k & 1), cls = k >> 1;
  r = 2 * (idx >> 3) + (cls >> 1);
  c = 2 * (idx & 7) + (cls & 1);
}
// tile cell of load slot u of thread group g: slots 0..7 are the group's own interior positions (pool_pos: the position
// phase then has its x in registers), slots 8..11 the 105 rim cells in turn
__device__ __forceinline__ void pool_cell(int g, int u, int& r, int& c, bool& live) {
  if (u < 8) {
    pool_pos(g, u, r, c);
    r += 1; c += 1; live = true;
    return;
  }
  const int j = g + 32 * (u - 8);
  live = j < PR * PR - PT * PT;
  if (j < PR) { r = 0; c = j; }
  else if (j < 3 * PR) { r = PT + 1 + (j - PR) / PR; c = (j - PR) % PR; }
  else { const int jj = j - 3 * PR, k3 = jj % 3; r = 1 + jj / 3; c = k3 == 0 ? 0 : PT + k3; }
}
// every global load of a tile (x with its rim, the pooled gradient of the windows that touch it) into registers first: one
// round trip per tile
__device__ __forceinline__ void pool_tile_load(PoolRegs& R, const float* __restrict__ xn, const float* __restrict__ dpn, int H, int W,
                                               int PH, int PW, int C4, int cbase, int h0, int w0) {
  const int tid = threadIdx.x, c8 = tid & 7;
  const int ph0 = h0 / 2, pw0 = w0 / 2;
#pragma unroll
  for (int u = 0; u < PXL; ++u) {
    int r, c; bool live;
    pool_cell(tid >> 3, u, r, c, live);
    const int h = h0 - 1 + r, w = w0 - 1 + c;
    R.xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && h >= 0 && h < H && w >= 0 && w < W) R.xr[u] = ld4(xn, ((int64_t)h * W + w) * C4 + cbase + c8);
  }
#pragma unroll
  for (int u = 0; u < PDL; ++u) {
    const int e = (tid >> 3) + u * (BN_TPB / 8);
    const int wr = e / PWN, wc = e - wr * PWN, ph = ph0 + wr, pw = pw0 + wc;
    R.dr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < PWN * PWN && ph < PH && pw < PW) R.dr[u] = ld4(dpn, ((int64_t)ph * PW + pw) * C4 + cbase + c8);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// LDS image of the tile: y = relu(x scale + shift) where the pixel exists, -1 elsewhere (below every y: never a winner)
__device__ __forceinline__ void pool_tile_commit(PoolTile& T, const PoolRegs& R, int H, int W, int h0, int w0, float4 sc, float4 sf) {
  const int tid = threadIdx.x, c8 = tid & 7;
  __syncthreads();                                            // the previous tile's readers are done
#pragma unroll
  for (int u = 0; u < PXL; ++u) {
    int r, c; bool live;
    pool_cell(tid >> 3, u, r, c, live);
    const int h = h0 - 1 + r, w = w0 - 1 + c;
    const bool in = h >= 0 && h < H && w >= 0 && w < W;
    if (live) T.yt[r][c][c8] = in ? bn_relu4(R.xr[u], sc, sf) : make_float4(-1.f, -1.f, -1.f, -1.f);
  }
#pragma unroll
  for (int u = 0; u < PDL; ++u) {
    const int e = (tid >> 3) + u * (BN_TPB / 8);
    if (e < PWN * PWN) T.dpt[e / PWN][e % PWN][c8] = R.dr[u];
  }
  __syncthreads();
}
// (the tile is in LDS: pool_tile_commit; R still holds the thread's own eight positions in slots 0..7)
template <class Emit>
__device__ __forceinline__ void pool_tile_dz(PoolTile& T, const PoolRegs& R, int H, int W, int PH, int PW, int h0, int w0, float4 sc,
                                             float4 sf, Emit emit) {
  const int tid = threadIdx.x, c8 = tid & 7;
  const int ph0 = h0 / 2, pw0 = w0 / 2;                       // first window of the tile (h0, w0 are multiples of 16)
  for (int e = tid >> 3; e < PWN * PWN; e += BN_TPB / 8) {
    const int wr = e / PWN, wc = e - wr * PWN;
    uchar4 id = make_uchar4(255, 255, 255, 255);
    if (ph0 + wr < PH && pw0 + wc < PW) {
      float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);     // the first pixel above it wins: first maximum in window order
#pragma unroll
      for (int k9 = 0; k9 < 9; ++k9) {
        const float4 y = T.yt[2 * wr + k9 / 3][2 * wc + k9 % 3][c8];
        if (y.x > best.x) { best.x = y.x; id.x = (unsigned char)k9; }
        if (y.y > best.y) { best.y = y.y; id.y = (unsigned char)k9; }
        if (y.z > best.z) { best.z = y.z; id.z = (unsigned char)k9; }
        if (y.w > best.w) { best.w = y.w; id.w = (unsigned char)k9; }
      }
    }
    T.win[wr][wc][c8] = id;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int r, c;
    pool_pos(tid >> 3, k, r, c);
    const int h = h0 + r, w = w0 + c;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h < H && w < W) {
      // windows holding (h, w): ph in {h/2} for even h, {(h-1)/2, (h+1)/2} for odd h; likewise pw - taken in ascending
      // order, as ATen's backward sums them.  A window outside the pooled map carries id 255 and a zero gradient.
      const int wa = r >> 1, wb = (r + 1) >> 1, va = c >> 1, vb = (c + 1) >> 1;      // local window indices (h0, w0 even)
      const int kha = r - (2 * wa - 1), khb = r - (2 * wb - 1), kwa = c - (2 * va - 1), kwb = c - (2 * vb - 1);
      auto take = [&](int wr, int wc, int k9, bool on) {
        const uchar4 id = T.win[wr][wc][c8];
        const float4 d = T.dpt[wr][wc][c8];
        const unsigned char want = on ? (unsigned char)k9 : (unsigned char)254;
        g.x += id.x == want ? d.x : 0.f; g.y += id.y == want ? d.y : 0.f;
        g.z += id.z == want ? d.z : 0.f; g.w += id.w == want ? d.w : 0.f;
      };
      // (row odd <=> two window rows, column odd <=> two window columns: known per slot, the absent ones compile away)
      const bool two_r = ((k >> 1) >> 1) != 0, two_c = ((k >> 1) & 1) != 0;
      take(wa, va, kha * 3 + kwa, true);
      if (two_c) take(wa, vb, kha * 3 + kwb, true);
      if (two_r) take(wb, va, khb * 3 + kwa, true);
      if (two_r && two_c) take(wb, vb, khb * 3 + kwb, true);
      // ReLU: the winner of a window is positive or the whole window is zero (then the gradient stops here)
      const float4 y = bn_relu4(R.xr[k], sc, sf);
      g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
      emit(h, w, g, R.xr[k]);
    }
    __builtin_amdgcn_sched_barrier(0);       // one position at a time: the eight unrolled bodies must not pile up their reads
  }
}

// partial[by][0][C] = sum dz, partial[by][1][C] = sum dz (x - mean) over the tiles this work-group walks
__global__ void __launch_bounds__(BN_TPB, 2)
k_pool_bwd_reduce(const float* __restrict__ dp, const float* __restrict__ x, const float* __restrict__ scale_shift,
                  const float* __restrict__ mean, float* __restrict__ partial, int N, int H, int W, int PH, int PW, int C4) {
  __shared__ PoolTile T;
  __shared__ float4 red[2][BN_TPB];
  const int c8 = threadIdx.x & 7, cbase = blockIdx.x * 8;
  const float4 sc = ld4(scale_shift, cbase + c8), sf = ld4(scale_shift + (int64_t)C4 * 4, cbase + c8), mu = ld4(mean, cbase + c8);
  const int tx = (W + PT - 1) / PT, ty = (H + PT - 1) / PT, per = tx * ty, total = N * per;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  PoolRegs R;
  auto load = [&](int t) {
    const int n = t / per, tt = t - n * per;
    pool_tile_load(R, x + (int64_t)n * H * W * C4 * 4, dp + (int64_t)n * PH * PW * C4 * 4, H, W, PH, PW, C4, cbase, (tt / tx) * PT,
                   (tt % tx) * PT);
  };
  for (int t = blockIdx.y; t < total; t += gridDim.y) {
    const int tt = t % per, h0 = (tt / tx) * PT, w0 = (tt % tx) * PT;
    load(t);
    pool_tile_commit(T, R, H, W, h0, w0, sc, sf);
    pool_tile_dz(T, R, H, W, PH, PW, h0, w0, sc, sf, [&](int, int, float4 g, float4 xo) {
      s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
      q.x = __builtin_fmaf(g.x, xo.x - mu.x, q.x); q.y = __builtin_fmaf(g.y, xo.y - mu.y, q.y);
      q.z = __builtin_fmaf(g.z, xo.z - mu.z, q.z); q.w = __builtin_fmaf(g.w, xo.w - mu.w, q.w);
    });
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < 8) {                                        // fixed order over the 32 position groups
    for (int g = 1; g < BN_TPB / 8; ++g) {
      const float4 u = red[0][g * 8 + c8], v = red[1][g * 8 + c8];
      s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      q.x += v.x; q.y += v.y; q.z += v.z; q.w += v.w;
    }
    const int64_t C = (int64_t)C4 * 4;
    float* pr = partial + (int64_t)blockIdx.y * 2 * C;
    st4(pr, cbase + c8, s);
    st4(pr + C, cbase + c8, q);
  }
}

// dx = dz a + (x - mean) b + c, as k_bwd_apply
__global__ void __launch_bounds__(BN_TPB, 2)
k_pool_bwd_apply(const float* __restrict__ dp, const float* __restrict__ x, const float* __restrict__ scale_shift,
                 const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ weight,
                 const float* __restrict__ sums, const float* __restrict__ counts, int Wr, int64_t cstride, float* __restrict__ dx,
                 int N, int H, int W, int PH, int PW, int C4) {
  __shared__ PoolTile T;
  const int c8 = threadIdx.x & 7, cbase = blockIdx.x * 8, cc = cbase + c8;
  const int64_t C = (int64_t)C4 * 4;
  const float4 sc = ld4(scale_shift, cc), sf = ld4(scale_shift + C, cc), mu = ld4(mean, cc), is = ld4(invstd, cc);
  const float4 w = weight ? ld4(weight, cc) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sd = ld4(sums, cc), sx = ld4(sums + C, cc);
  double n_all = 0.0;
  for (int r = 0; r < Wr; ++r) n_all += (double)counts[r * cstride];
  const float inv_n = (float)(1.0 / n_all);
  float4 a, b, c;
  a.x = is.x * w.x; a.y = is.y * w.y; a.z = is.z * w.z; a.w = is.w * w.w;
  b.x = -(sx.x * inv_n) * is.x * is.x * a.x; b.y = -(sx.y * inv_n) * is.y * is.y * a.y;
  b.z = -(sx.z * inv_n) * is.z * is.z * a.z; b.w = -(sx.w * inv_n) * is.w * is.w * a.w;
  c.x = -(sd.x * inv_n) * a.x; c.y = -(sd.y * inv_n) * a.y; c.z = -(sd.z * inv_n) * a.z; c.w = -(sd.w * inv_n) * a.w;
  const int tx = (W + PT - 1) / PT, ty = (H + PT - 1) / PT, per = tx * ty, total = N * per;
  PoolRegs R;
  auto load = [&](int t) {
    const int n = t / per, tt = t - n * per;
    pool_tile_load(R, x + (int64_t)n * H * W * C, dp + (int64_t)n * PH * PW * C, H, W, PH, PW, C4, cbase, (tt / tx) * PT, (tt % tx) * PT);
  };
  for (int t = blockIdx.y; t < total; t += gridDim.y) {
    const int n = t / per, tt = t - n * per, h0 = (tt / tx) * PT, w0 = (tt % tx) * PT;
    load(t);
    pool_tile_commit(T, R, H, W, h0, w0, sc, sf);
    float* dxn = dx + (int64_t)n * H * W * C;
    pool_tile_dz(T, R, H, W, PH, PW, h0, w0, sc, sf, [&](int hh, int ww, float4 g, float4 xo) {
      float4 o;
      o.x = __builtin_fmaf(g.x, a.x, __builtin_fmaf(xo.x - mu.x, b.x, c.x));
      o.y = __builtin_fmaf(g.y, a.y, __builtin_fmaf(xo.y - mu.y, b.y, c.y));
      o.z = __builtin_fmaf(g.z, a.z, __builtin_fmaf(xo.z - mu.z, b.z, c.z));
      o.w = __builtin_fmaf(g.w, a.w, __builtin_fmaf(xo.w - mu.w, b.w, c.w));
      st4(dxn, ((int64_t)hh * W + ww) * C4 + cc, o);
    });
  }
}

#define CHECKED(what)                         \
  do {                                        \
    if (int e = check_launch(what)) return e; \
  } while (0)

}  // namespace

// workspace: BN_HDR int32 tickets (zero on entry, left zero), then the [gy][2][C] partial rows
size_t bn_workspace_floats(int64_t M, int C) {
  const BnTile t = bn_tile(M, C / 4, CAP_MAX);
  if (t.gx > BN_HDR) return 0;         // more channel chunks than tickets (C > 4096 with an odd C / 4): unsupported
  return (size_t)BN_HDR + (size_t)t.gy * 2 * C;
}

int run_bn_stats(const float* x, int64_t M, int C, float* ws, float* packed, hipStream_t st) {
  const BnTile t = bn_tile(M, C / 4, bn_caps().stats);
  const bool fold = bn_fold_in_kernel();
  if (bn_stream_policy(M, C, false).stats)
    hipLaunchKernelGGL(k_stats<true>, dim3(t.gx, t.gy), dim3(BN_TPB), 0, st, x, fold ? reinterpret_cast<int*>(ws) : nullptr,
                       ws + BN_HDR, packed, M, C / 4, t.TX, t.TY);
  else
    hipLaunchKernelGGL(k_stats<false>, dim3(t.gx, t.gy), dim3(BN_TPB), 0, st, x, fold ? reinterpret_cast<int*>(ws) : nullptr,
                       ws + BN_HDR, packed, M, C / 4, t.TX, t.TY);
  CHECKED("k_stats");
  if (!fold) {
    hipLaunchKernelGGL(k_stats_finalize, dim3((C + FIN_CH - 1) / FIN_CH), dim3(BN_TPB), 0, st, x, ws + BN_HDR, t.gy, M, C, packed);
    CHECKED("k_stats_finalize");
  }
  return PIXPRO_OK;
}

// the bit record of the ReLU decisions: rows * C / 8 bytes; shapes whose waves do not hold 64 consecutive float4: 0
size_t bn_relu_bits_bytes(int64_t M, int C) {
  if (C % 256 || M * (int64_t)(C / 256) > 0x7fffffff) return 0;
  return (size_t)M * (size_t)C / 8;
}

int run_bn_apply(const float* gathered, int W, const float* x, const float* res, const float* weight, const float* bias,
                 float* running_mean, float* running_var, float momentum, float eps, int relu, int64_t M, int C,
                 float* y, float* mean, float* invstd, float* scale_shift, void* relu_bits, hipStream_t st) {
  const BnTile t = bn_tile(M, C / 4, bn_caps().apply);
  const dim3 g(t.gx, t.gy), b(BN_TPB);
  unsigned long long* bits = static_cast<unsigned long long*>(relu_bits);
  const bool nt = bn_stream_policy(M, C, false).apply;
#define APPLY_(R, S, B, N) hipLaunchKernelGGL((k_apply<R, S, B, N>), g, b, 0, st, gathered, W, weight, bias, running_mean,    \
                                              running_var, momentum, eps, mean, invstd, scale_shift, x, res, y, bits, M,     \
                                              C / 4, t.TX, t.TY)
#define APPLY(R, S, B) do { if (nt) APPLY_(R, S, B, true); else APPLY_(R, S, B, false); } while (0)
  if (bits) {
    if (!relu || t.TX != 64 || !bn_relu_bits_bytes(M, C)) return PIXPRO_ERR_ARG;
    if (res) APPLY(true, true, true); else APPLY(true, false, true);
  }
  else if (relu && res) APPLY(true, true, false);
  else if (relu) APPLY(true, false, false);
  else if (res) APPLY(false, true, false);
  else APPLY(false, false, false);
#undef APPLY
#undef APPLY_
  CHECKED("k_apply");
  return PIXPRO_OK;
}

// ReLU decision source: 0 none, 1 the saved output y, 2 recomputed from x (scale_shift given), 3 the bit record
static inline int mask_mode(int relu, const float* scale_shift, const void* bits) {
  return !relu ? 0 : (bits ? 3 : (scale_shift ? 2 : 1));
}

int run_bn_bwd_reduce(const float* dy, const float* x, const float* y, const float* scale_shift, const void* relu_bits,
                      const float* mean, const float* invstd, int relu, int64_t M, int C, float* ws, float* sums,
                      float* grad_weight, float* grad_bias, hipStream_t st) {
  const BnTile t = bn_tile(M, C / 4, bn_caps().bwd_reduce);
  if (relu_bits) {
    if (!relu || t.TX != 64 || !bn_relu_bits_bytes(M, C)) return PIXPRO_ERR_ARG;
    y = static_cast<const float*>(relu_bits);        // MODE 3 reads the record through the `y` argument
  }
  const dim3 g(t.gx, t.gy), b(BN_TPB);
  const bool fold = bn_fold_in_kernel();
  const bool nt = bn_stream_policy(M, C, false).bwd_reduce;
#define REDUCE_(MD, N) hipLaunchKernelGGL((k_bwd_reduce<MD, N>), g, b, 0, st, dy, x, y, scale_shift, mean, invstd,             \
                                          fold ? reinterpret_cast<int*>(ws) : nullptr, ws + BN_HDR, sums, grad_weight,       \
                                          grad_bias, M, C / 4, t.TX, t.TY)
#define REDUCE(MD) do { if (nt) REDUCE_(MD, true); else REDUCE_(MD, false); } while (0)
  switch (mask_mode(relu, scale_shift, relu_bits)) {
    case 0: REDUCE(0); break;
    case 1: REDUCE(1); break;
    case 2: REDUCE(2); break;
    default: REDUCE(3); break;
  }
#undef REDUCE
#undef REDUCE_
  CHECKED("k_bwd_reduce");
  if (!fold) {
    hipLaunchKernelGGL(k_bwd_finalize, dim3((C + FIN_CH - 1) / FIN_CH), dim3(BN_TPB), 0, st, ws + BN_HDR, t.gy, C, invstd, sums,
                       grad_weight, grad_bias);
    CHECKED("k_bwd_finalize");
  }
  return PIXPRO_OK;
}

int run_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* scale_shift, const void* relu_bits,
                     const float* mean, const float* invstd, const float* weight, const float* sums, const float* counts,
                     int world, int64_t count_stride, int relu, int64_t M, int C, float* dx, float* dres, hipStream_t st) {
  const BnTile t = bn_tile(M, C / 4, bn_caps().bwd_apply);
  if (relu_bits) {
    if (!relu || t.TX != 64 || !bn_relu_bits_bytes(M, C)) return PIXPRO_ERR_ARG;
    y = static_cast<const float*>(relu_bits);
  }
  const dim3 g(t.gx, t.gy), b(BN_TPB);
  const bool nt = bn_stream_policy(M, C, dres != nullptr).bwd_apply;
#define BAPPLY_(MD, S, N) hipLaunchKernelGGL((k_bwd_apply<MD, S, N>), g, b, 0, st, dy, x, y, scale_shift, mean, invstd, weight, \
                                             sums, counts, world, count_stride, dx, dres, M, C / 4, t.TX, t.TY)
#define BAPPLY(MD, S) do { if (nt) BAPPLY_(MD, S, true); else BAPPLY_(MD, S, false); } while (0)
  const int mode = mask_mode(relu, scale_shift, relu_bits);
  if (dres) { if (mode == 0) BAPPLY(0, true); else if (mode == 1) BAPPLY(1, true); else if (mode == 2) BAPPLY(2, true); else BAPPLY(3, true); }
  else { if (mode == 0) BAPPLY(0, false); else if (mode == 1) BAPPLY(1, false); else if (mode == 2) BAPPLY(2, false); else BAPPLY(3, false); }
#undef BAPPLY
#undef BAPPLY_
  CHECKED("k_bwd_apply");
  return PIXPRO_OK;
}

int run_bn_masked_add(const float* a, const float* dy, const void* relu_bits, float* out, int64_t M, int C, hipStream_t st) {
  if (!bn_relu_bits_bytes(M, C)) return PIXPRO_ERR_ARG;
  const int64_t n4 = M * (int64_t)(C / 4);
  const int64_t blocks = std::min<int64_t>((n4 + (int64_t)BN_TPB * 4 - 1) / ((int64_t)BN_TPB * 4), 2048);
  if (bn_stream_policy(M, C, true).bwd_apply)
    hipLaunchKernelGGL(k_masked_add<true>, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(BN_TPB), 0, st, a, dy,
                       static_cast<const float*>(relu_bits), out, n4);
  else
    hipLaunchKernelGGL(k_masked_add<false>, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(BN_TPB), 0, st, a, dy,
                       static_cast<const float*>(relu_bits), out, n4);
  CHECKED("k_masked_add");
  return PIXPRO_OK;
}

// ---- the stem (BatchNorm + ReLU + 3x3/2 max pooling) ----
static inline bool pool_shape_ok(int N, int H, int W, int C) {
  return N >= 1 && H >= 2 && W >= 2 && C >= 32 && C % 32 == 0 && (int64_t)N * ((H + PT - 1) / PT) * ((W + PT - 1) / PT) < 0x7fffffff;
}
int bn_pool_supported(int N, int H, int W, int C) { return pool_shape_ok(N, H, W, C); }

int run_bn_apply_pool(const float* gathered, int Wr, const float* x, const float* weight, const float* bias, float* running_mean,
                      float* running_var, float momentum, float eps, int N, int H, int W, int C, float* pooled, float* mean,
                      float* invstd, float* scale_shift, hipStream_t st) {
  if (!pool_shape_ok(N, H, W, C)) return PIXPRO_ERR_ARG;
  const int PH = (H - 1) / 2 + 1, PW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(k_bn_publish, dim3((C + BN_TPB - 1) / BN_TPB), dim3(BN_TPB), 0, st, gathered, Wr, weight, bias, running_mean,
                     running_var, momentum, eps, mean, invstd, scale_shift, C);
  CHECKED("k_bn_publish");
  hipLaunchKernelGGL(k_apply_pool, dim3(((PH + 7) / 8) * ((PW + 7) / 8), N, (C / 4 + 15) / 16), dim3(BN_TPB), 0, st, x,
                     scale_shift, pooled, H, W, PH, PW, C / 4);
  CHECKED("k_apply_pool");
  return PIXPRO_OK;
}

static inline int pool_rows(int N, int H, int W) {
  const int64_t tiles = (int64_t)N * ((H + PT - 1) / PT) * ((W + PT - 1) / PT);
  return (int)std::min<int64_t>(tiles, 1024);
}

int run_bn_pool_bwd_reduce(const float* dp, const float* x, const float* scale_shift, const float* mean, const float* invstd,
                           int N, int H, int W, int C, float* ws, float* sums, float* grad_weight, float* grad_bias,
                           hipStream_t st) {
  if (!pool_shape_ok(N, H, W, C)) return PIXPRO_ERR_ARG;
  const int PH = (H - 1) / 2 + 1, PW = (W - 1) / 2 + 1, gy = pool_rows(N, H, W);
  hipLaunchKernelGGL(k_pool_bwd_reduce, dim3(C / 32, gy), dim3(BN_TPB), 0, st, dp, x, scale_shift, mean, ws + BN_HDR, N, H, W, PH,
                     PW, C / 4);
  CHECKED("k_pool_bwd_reduce");
  hipLaunchKernelGGL(k_bwd_finalize, dim3((C + FIN_CH - 1) / FIN_CH), dim3(BN_TPB), 0, st, ws + BN_HDR, gy, C, invstd, sums,
                     grad_weight, grad_bias);
  CHECKED("k_bwd_finalize");
  return PIXPRO_OK;
}

int run_bn_pool_bwd_apply(const float* dp, const float* x, const float* scale_shift, const float* mean, const float* invstd,
                          const float* weight, const float* sums, const float* counts, int world, int64_t count_stride, int N,
                          int H, int W, int C, float* dx, hipStream_t st) {
  if (!pool_shape_ok(N, H, W, C)) return PIXPRO_ERR_ARG;
  const int PH = (H - 1) / 2 + 1, PW = (W - 1) / 2 + 1;
  const int64_t tiles = (int64_t)N * ((H + PT - 1) / PT) * ((W + PT - 1) / PT);
  hipLaunchKernelGGL(k_pool_bwd_apply, dim3(C / 32, (unsigned)std::min<int64_t>(tiles, 1024)), dim3(BN_TPB), 0, st, dp, x,
                     scale_shift, mean, invstd, weight, sums, counts, world, count_stride, dx, N, H, W, PH, PW, C / 4);
  CHECKED("k_pool_bwd_apply");
  return PIXPRO_OK;
}

}  // namespace pixpro
