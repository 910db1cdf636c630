// BatchNorm1d (+ fused ReLU) forward / backward over [N, D] fp32 node features (gfx950).
// Replaces the ATen/cuDNN batch_norm the reference reaches through chem/model.py:252,269-275 and
// bio/model.py:24.  Pure HBM-bound column reductions + one elementwise pass:
//   pass 1: per-block partial column sums (float4 loads, rows strided over 4 row-lanes per block)
//   pass 2: per-column finalize in double (fixed block order -> deterministic)
//   pass 3: elementwise normalise (+ReLU) / gradient, float4.
// Statistics use sums shifted by row 0 (K = x[0,:]) so that var = E[(x-K)^2] - E[x-K]^2 does not
// cancel catastrophically when |mean| >> std.
#include <atomic>

#include "bn_fold.h"
#include "common.h"

namespace pgnn {
namespace {

constexpr int kMaxBlocks = kBnMaxBlocks;
__device__ unsigned g_bn_fold_tickets[kFoldSlots][kFoldMaxGroups + 8];
__device__ unsigned g_bn_fwd_fold_tickets[kFoldSlots / 4][kFwdFoldPanels][kFoldMaxGroups + 8];

// Inverted dropout fused into the normalise pass (F.dropout after the ReLU, chem/model.py:271-275).
// Counter-based: the keep bits of float4 (row r, column group c4) come from one splitmix64 of
// (seed, r*d4 + c4), 16 bits per element, keep iff bits >= p*65536 -- so the backward regenerates the
// mask from the seed instead of storing it.  thresh == 0 disables (wave-uniform branch).
struct Drop {
  uint64_t seed;
  uint32_t thresh;  // round(p * 65536)
  float scale;      // 1 / (1 - p)
};
inline Drop make_drop(float p, uint64_t seed) {
  Drop d;
  d.seed = seed;
  d.thresh = p > 0.f ? (uint32_t)(p * 65536.f + 0.5f) : 0u;
  d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  return d;
}
__device__ __forceinline__ float4 drop_factors(const Drop& d, int64_t r, int d4, int c4) {
  uint64_t z = d.seed + (uint64_t)(r * d4 + c4) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  float4 f;
  f.x = ((uint32_t)(z) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
  f.y = ((uint32_t)(z >> 16) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
  f.z = ((uint32_t)(z >> 32) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
  f.w = ((uint32_t)(z >> 48) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
  return f;
}

// thread t of a (dim/4 x 4)-shaped block: column group c4 = t % d4, row lane rl = t / d4
// blockDim.x = 4 * d4 rounded up to a multiple of 64.
__device__ __forceinline__ void block_col_reduce2(float4 a, float4 b, int d4, float* lds, float* dst_a,
                                                  float* dst_b) {
  // lds: [2][4][d4*4] floats
  const int t = threadIdx.x, c4 = t % d4, rl = t / d4, dim = d4 * 4;
  if (rl < 4) {
    reinterpret_cast<float4*>(lds + (0 * 4 + rl) * dim)[c4] = a;
    reinterpret_cast<float4*>(lds + (1 * 4 + rl) * dim)[c4] = b;
  }
  __syncthreads();
  for (int q = t; q < dim; q += blockDim.x) {  // (agent-scope stores: another block of the same launch may read them, see k_bn_bwd_partial)
    publish(dst_a + q, (lds[q] + lds[dim + q]) + (lds[2 * dim + q] + lds[3 * dim + q]));
    publish(dst_b + q, (lds[4 * dim + q] + lds[5 * dim + q]) + (lds[6 * dim + q] + lds[7 * dim + q]));
  }
  publish_commit();  // acknowledged before whatever barrier / ticket of the caller announces them
}

__global__ void k_bn_stats_partial(const float* __restrict__ x, int64_t ldx, int n, int d4,
                                   float* __restrict__ partial /*[nblk][2][dim]*/) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, c4 = t % d4, rl = t / d4, dim = d4 * 4;
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  float4 s1 = f4_zero(), s2 = f4_zero();
  if (rl < 4) {
    const float4 k = reinterpret_cast<const float4*>(x)[c4];  // shift = row 0
    constexpr int U = 4;  // four rows in flight per thread, summed in row order (see k_bn_bwd_partial)
    for (int rb = r0 + rl; rb < r1; rb += 4 * U) {
      float4 vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) vv[u] = reinterpret_cast<const float4*>(x + (int64_t)min(rb + 4 * u, r1 - 1) * ldx)[c4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (rb + 4 * u >= r1) break;
        const float4 v = vv[u];
        const float dx = v.x - k.x, dy = v.y - k.y, dz = v.z - k.z, dw = v.w - k.w;
        s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
        s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
      }
    }
  }
  float* p = partial + (size_t)blockIdx.x * 2 * dim;
  block_col_reduce2(s1, s2, d4, lds, p, p + dim);
}

// coef[0][dim] = scale a, coef[1][dim] = shift b  (y = a*x + b)
__global__ void k_bn_stats_final(const float* __restrict__ partial, int nblk, const float* __restrict__ x,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                 float momentum, float eps, int training, int n, int dim,
                                 float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                 float* __restrict__ coef) {
  // 256 threads = 4 columns x one wave of 64 slices (16-lane slices left a [N ~ 7k, 300] layer with 19 blocks that each
  // walked their partials in 13 dependent rounds: 8 us for a 0.5 MB reduction)
  const int sl = threadIdx.x & 63;
  const int c = min(blockIdx.x * 4 + (threadIdx.x >> 6), dim - 1);
  const bool writer = sl == 0 && (blockIdx.x * 4 + (threadIdx.x >> 6)) < dim;
  float mean, invstd;
  if (training) {
    const double s1 = slice_sum64(partial + c, (size_t)2 * dim, nblk, sl);
    const double s2 = slice_sum64(partial + dim + c, (size_t)2 * dim, nblk, sl);
    if (!writer) return;
    const double m1 = s1 / n;
    double var = s2 / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    mean = (float)((double)x[c] + m1);
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  } else {
    if (!writer) return;
    mean = running_mean[c];
    invstd = 1.0f / sqrtf(running_var[c] + eps);
  }
  if (save_mean) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
  }
  const float a = invstd * gamma[c];
  coef[c] = a;
  coef[dim + c] = fmaf(-mean, a, beta[c]);  // same expression as the backward's recomputation
}

// The same finalize from per-16-row-block statistics (pgnn_linear_fwd_colstats: the product in front of the BatchNorm left, for
// every block t and column c, S_t = sum of the block's rows and Q_t = sum of squared deviations from the block's own mean):
// blocks are merged pairwise with the parallel-variance formula in double,
//     n = nA + nB ; d = mB - mA ; m = mA + d nB / n ; M2 = M2A + M2B + d d nA nB / n,
// lane s of the column's wave folding blocks s, s + 64, ... in order and a 64-lane butterfly on top (fixed tree; lane 0 writes).
// No shift is needed: every deviation is taken from a mean of 16 neighbouring values.
__global__ void k_bn_stats_final_blocks(const float* __restrict__ blocks /*[nblk][2][dim]*/, int nblk, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* __restrict__ running_mean,
                                        float* __restrict__ running_var, float momentum, float eps, int n, int dim,
                                        float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ coef) {
  const int sl = threadIdx.x & 63;
  const int c = min(blockIdx.x * 4 + (threadIdx.x >> 6), dim - 1);
  const bool writer = sl == 0 && (blockIdx.x * 4 + (threadIdx.x >> 6)) < dim;
  double cn = 0.0, cm = 0.0, c2 = 0.0;
  auto merge = [&](double nb, double mb, double qb) {
    if (nb == 0.0) return;
    const double nn = cn + nb, d = mb - cm;
    cm += d * (nb / nn);
    c2 += qb + d * d * (cn * nb / nn);
    cn = nn;
  };
  for (int t = sl; t < nblk; t += 64) {
    const double nb = (double)min(16, n - 16 * t);
    merge(nb, (double)blocks[(size_t)t * 2 * dim + c] / nb, (double)blocks[(size_t)t * 2 * dim + dim + c]);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double nb = __shfl_xor(cn, off), mb = __shfl_xor(cm, off), qb = __shfl_xor(c2, off);
    merge(nb, mb, qb);
  }
  if (!writer) return;
  double var = c2 / n;
  if (var < 0.0) var = 0.0;
  const float mean = (float)cm;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
  if (save_mean) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
  }
  const float a = invstd * gamma[c];
  coef[c] = a;
  coef[dim + c] = fmaf(-mean, a, beta[c]);  // same expression as the backward's recomputation
}

__global__ void k_bn_apply(const float* __restrict__ x, int64_t ldx, const float* __restrict__ coef,
                           int relu, float* __restrict__ y, int64_t ldy, int n, int d4, Drop drop) {
  const int t = threadIdx.x, c4 = t % d4, rl = t / d4, dim = d4 * 4;
  if (rl >= 4) return;
  const float4 a = reinterpret_cast<const float4*>(coef)[c4];
  const float4 b = reinterpret_cast<const float4*>(coef + dim)[c4];
  for (int64_t r = (int64_t)blockIdx.x * 4 + rl; r < n; r += (int64_t)gridDim.x * 4) {
    const float4 v = reinterpret_cast<const float4*>(x + r * ldx)[c4];
    float4 o = make_float4(fmaf(a.x, v.x, b.x), fmaf(a.y, v.y, b.y), fmaf(a.z, v.z, b.z), fmaf(a.w, v.w, b.w));
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (drop.thresh) {
      const float4 f = drop_factors(drop, r, d4, c4);
      o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
    }
    reinterpret_cast<float4*>(y + r * ldy)[c4] = o;
  }
}

// backward pass 1: partial sums of dyr and dyr*xhat (dyr = dy masked by the recomputed ReLU)
__global__ void k_bn_bwd_partial(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                 int64_t ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                 float* __restrict__ coef /*out (block 0): a,b,mean,invstd*/, int relu,
                                 int n, int d4, float* __restrict__ partial, Drop drop, double* __restrict__ gsum,
                                 unsigned* __restrict__ tickets, int training, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta, const int64_t* __restrict__ rows, int nrows) {
  // rows (round 6, optional): the ONLY rows where dy is not zero, ascending or not, none repeated (the masking head's gradient:
  // ~17 % of the rows) -- the sums run over rows[0 .. nrows) instead of 0 .. n; every other row would add an exact zero
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, c4 = t % d4, rl = t / d4, dim = d4 * 4;
  const int cnt = rows ? nrows : n;
  const int per = (cnt + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(cnt, r0 + per);
  float4 s1 = f4_zero(), s2 = f4_zero();
  if (rl < 4) {
    // forward coefficients y = a*x + b, recomputed exactly as k_bn_stats_final formed them
    const float4 gm = reinterpret_cast<const float4*>(gamma)[c4], bt = reinterpret_cast<const float4*>(beta)[c4];
    const float4 mu = reinterpret_cast<const float4*>(save_mean)[c4];
    const float4 is = reinterpret_cast<const float4*>(save_invstd)[c4];
    const float4 a = make_float4(is.x * gm.x, is.y * gm.y, is.z * gm.z, is.w * gm.w);
    const float4 b = make_float4(fmaf(-mu.x, a.x, bt.x), fmaf(-mu.y, a.y, bt.y), fmaf(-mu.z, a.z, bt.z), fmaf(-mu.w, a.w, bt.w));
    if (blockIdx.x == 0 && rl == 0) {
      reinterpret_cast<float4*>(coef)[c4] = a;
      reinterpret_cast<float4*>(coef + dim)[c4] = b;
      reinterpret_cast<float4*>(coef + 2 * dim)[c4] = mu;
      reinterpret_cast<float4*>(coef + 3 * dim)[c4] = is;
    }
    // four of the thread's rows in flight (clamped, unconditional loads), consumed in row order: the same sums as a loop that
    // pays a memory round trip per row (8 round trips per thread at 32 rows per block: ~16 us of latency at 6 747 rows)
    constexpr int U = 4;
    for (int rb = r0 + rl; rb < r1; rb += 4 * U) {
      float4 vv[U], gg[U];
      int64_t rr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = min(rb + 4 * u, r1 - 1);
        rr[u] = rows ? min(max(rows[r], (int64_t)0), (int64_t)n - 1) : (int64_t)r;
        vv[u] = reinterpret_cast<const float4*>(x + rr[u] * ldx)[c4];
        gg[u] = reinterpret_cast<const float4*>(dy + rr[u] * lddy)[c4];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (rb + 4 * u >= r1) break;
        const int64_t r = rr[u];
        const float4 v = vv[u];
        float4 g = gg[u];
        if (drop.thresh) {
          const float4 f = drop_factors(drop, r, d4, c4);
          g.x *= f.x; g.y *= f.y; g.z *= f.z; g.w *= f.w;
        }
        if (relu) {
          if (!(fmaf(a.x, v.x, b.x) > 0.f)) g.x = 0.f;
          if (!(fmaf(a.y, v.y, b.y) > 0.f)) g.y = 0.f;
          if (!(fmaf(a.z, v.z, b.z) > 0.f)) g.z = 0.f;
          if (!(fmaf(a.w, v.w, b.w) > 0.f)) g.w = 0.f;
        }
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, (v.x - mu.x) * is.x, s2.x);
        s2.y = fmaf(g.y, (v.y - mu.y) * is.y, s2.y);
        s2.z = fmaf(g.z, (v.z - mu.z) * is.z, s2.z);
        s2.w = fmaf(g.w, (v.w - mu.w) * is.w, s2.w);
      }
    }
  }
  float* p = partial + (size_t)blockIdx.x * 2 * dim;
  block_col_reduce2(s1, s2, d4, lds, p, p + dim);
  if (!tickets) return;  // (the fold is k_bn_bwd_final's: more than kFoldMaxGroups groups)
  // ---- the fold of the partials, inside this launch (a launch of its own cost 6-8 us + a kernel boundary per layer on the
  // backward's critical path): bn_fold.h
  const BnBwdFold f{gamma, save_invstd, partial, gsum, tickets, coef, dgamma, dbeta, training, n};
  bn_bwd_fold(f, dim, blockIdx.x, gridDim.x, t, blockDim.x);
}

// coef layout for backward: [a, b, mean, invstd, k1, k2, k3] each [dim]
//   dx = k1*dyr + k2*(x-mean) + k3   with k1 = gamma*invstd;
//   training: k2 = -k1*invstd*mean(dyr*xhat), k3 = -k1*mean(dyr); eval: k2 = k3 = 0
__global__ void k_bn_bwd_final(const float* __restrict__ partial, int nblk, int training, int n, int dim,
                               const float* __restrict__ gamma, float* __restrict__ coef,
                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int sl = threadIdx.x & 63;
  const int c = min(blockIdx.x * 4 + (threadIdx.x >> 6), dim - 1);
  const bool writer = sl == 0 && (blockIdx.x * 4 + (threadIdx.x >> 6)) < dim;
  const double s1 = slice_sum64(partial + c, (size_t)2 * dim, nblk, sl);
  const double s2 = slice_sum64(partial + dim + c, (size_t)2 * dim, nblk, sl);
  if (!writer) return;
  if (dgamma) dgamma[c] = (float)s2;
  if (dbeta) dbeta[c] = (float)s1;
  const float invstd = coef[3 * dim + c];
  const float k1 = gamma[c] * invstd;
  // dx = k1 * (dyr - s1/n - xhat * s2/n),  xhat = (x - mean)*invstd
  float k2 = 0.f, k3 = 0.f;
  if (training) {
    const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
    k2 = -k1 * invstd * m2;  // multiplies (x - mean)
    k3 = -k1 * m1;
  }
  coef[4 * dim + c] = k1;
  coef[5 * dim + c] = k2;
  coef[6 * dim + c] = k3;
}

// rowmax (optional): the bit patterns of max |dx[r, :]| per row -- a row's d4 threads fold theirs by an LDS atomic maximum, two
// barriers per trip of four rows (every trip count is block-uniform; the lanes beyond the fourth row lane stay in the loop, idle)
__global__ void k_bn_bwd_apply(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                               int64_t ldx, const float* __restrict__ coef, int relu, float* __restrict__ dx,
                               int64_t lddx, int n, int d4, Drop drop, uint32_t* __restrict__ rowmax) {
  const int t = threadIdx.x, c4 = t % d4, rl = t / d4, dim = d4 * 4;
  __shared__ unsigned rmax[4];
  if (rowmax) {
    const float4 a = reinterpret_cast<const float4*>(coef)[min(c4, d4 - 1)];
    const float4 b = reinterpret_cast<const float4*>(coef + dim)[min(c4, d4 - 1)];
    const float4 mu = reinterpret_cast<const float4*>(coef + 2 * dim)[min(c4, d4 - 1)];
    const float4 k1 = reinterpret_cast<const float4*>(coef + 4 * dim)[min(c4, d4 - 1)];
    const float4 k2 = reinterpret_cast<const float4*>(coef + 5 * dim)[min(c4, d4 - 1)];
    const float4 k3 = reinterpret_cast<const float4*>(coef + 6 * dim)[min(c4, d4 - 1)];
    for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < n; r0 += (int64_t)gridDim.x * 4) {
      if (t < 4) rmax[t] = 0u;
      __syncthreads();
      const int64_t r = r0 + rl;
      if (rl < 4 && r < n) {
        const float4 v = reinterpret_cast<const float4*>(x + r * ldx)[c4];
        float4 g = reinterpret_cast<const float4*>(dy + r * lddy)[c4];
        if (relu) {
          if (!(fmaf(a.x, v.x, b.x) > 0.f)) g.x = 0.f;
          if (!(fmaf(a.y, v.y, b.y) > 0.f)) g.y = 0.f;
          if (!(fmaf(a.z, v.z, b.z) > 0.f)) g.z = 0.f;
          if (!(fmaf(a.w, v.w, b.w) > 0.f)) g.w = 0.f;
        }
        float4 o;
        o.x = fmaf(k1.x, g.x, fmaf(k2.x, v.x - mu.x, k3.x));
        o.y = fmaf(k1.y, g.y, fmaf(k2.y, v.y - mu.y, k3.y));
        o.z = fmaf(k1.z, g.z, fmaf(k2.z, v.z - mu.z, k3.z));
        o.w = fmaf(k1.w, g.w, fmaf(k2.w, v.w - mu.w, k3.w));
        reinterpret_cast<float4*>(dx + r * lddx)[c4] = o;
        atomicMax(&rmax[rl], __float_as_uint(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)))));
      }
      __syncthreads();
      if (t < 4 && r0 + t < n) rowmax[r0 + t] = rmax[t];
    }
    return;
  }
  if (rl >= 4) return;
  const float4 a = reinterpret_cast<const float4*>(coef)[c4];
  const float4 b = reinterpret_cast<const float4*>(coef + dim)[c4];
  const float4 mu = reinterpret_cast<const float4*>(coef + 2 * dim)[c4];
  const float4 k1 = reinterpret_cast<const float4*>(coef + 4 * dim)[c4];
  const float4 k2 = reinterpret_cast<const float4*>(coef + 5 * dim)[c4];
  const float4 k3 = reinterpret_cast<const float4*>(coef + 6 * dim)[c4];
  for (int64_t r = (int64_t)blockIdx.x * 4 + rl; r < n; r += (int64_t)gridDim.x * 4) {
    const float4 v = reinterpret_cast<const float4*>(x + r * ldx)[c4];
    float4 g = reinterpret_cast<const float4*>(dy + r * lddy)[c4];
    if (drop.thresh) {
      const float4 f = drop_factors(drop, r, d4, c4);
      g.x *= f.x; g.y *= f.y; g.z *= f.z; g.w *= f.w;
    }
    if (relu) {
      if (!(fmaf(a.x, v.x, b.x) > 0.f)) g.x = 0.f;
      if (!(fmaf(a.y, v.y, b.y) > 0.f)) g.y = 0.f;
      if (!(fmaf(a.z, v.z, b.z) > 0.f)) g.z = 0.f;
      if (!(fmaf(a.w, v.w, b.w) > 0.f)) g.w = 0.f;
    }
    float4 o;
    o.x = fmaf(k1.x, g.x, fmaf(k2.x, v.x - mu.x, k3.x));
    o.y = fmaf(k1.y, g.y, fmaf(k2.y, v.y - mu.y, k3.y));
    o.z = fmaf(k1.z, g.z, fmaf(k2.z, v.z - mu.z, k3.z));
    o.w = fmaf(k1.w, g.w, fmaf(k2.w, v.w - mu.w, k3.w));
    reinterpret_cast<float4*>(dx + r * lddx)[c4] = o;
  }
}

// rows per block of the two partial-sum kernels.  Measured at one 256-graph batch (N = 6 747, tools/small_kernel_bench.py):
// 32 rows -> statistics 9.6 us, backward (3 launches) 17.2 us; 16 -> 10.8 / 17.6; 8 -> 14.3 / 20.9; 4 -> 14.7 / 20.4.  Each of
// these launches already sits at the ~5 us floor of a dependent kernel; more, smaller blocks only add partials to reduce.
inline int stat_blocks(int64_t n) {
  return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, env_knob("PGNN_BN_ROWS_PER_BLOCK", 32)), 1), kMaxBlocks);
}
inline int stat_threads(int64_t dim) { return (int)align_up((size_t)dim, 64); }  // 4 row lanes x dim/4

inline int check_args(int64_t n, int64_t dim) {
  if (n <= 0 || dim <= 0 || dim % 4 != 0 || dim > 1024) {
    set_error("batchnorm: need N > 0 and feature width a multiple of 4 in (0,1024] (N=%lld D=%lld)",
              (long long)n, (long long)dim);
    return PGNN_ERR_ARG;
  }
  return PGNN_OK;
}

unsigned* draw_fold_tickets() {
  static std::atomic<unsigned> next_slot{0};
  static thread_local unsigned* base = nullptr;  // (one device per process; the lookup takes the runtime's locks)
  static thread_local int base_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!base || base_dev != dev) {
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_bn_fold_tickets)) != hipSuccess) return nullptr;
    base_dev = dev;
  }
  return base + (size_t)(next_slot.fetch_add(1, std::memory_order_relaxed) % kFoldSlots) * (kFoldMaxGroups + 8);
}

}  // namespace

bool bn_fwd_fold_scratch(void* ws, size_t ws_bytes, int64_t n, int64_t dim, BnFwdFold* f) {
  const size_t tiles = (size_t)ceil_div(n, 64);
  if (n <= 1 || dim <= 0 || ceil_div((int64_t)tiles, kFoldGroup) > kFoldMaxGroups || ceil_div(dim, 160) > kFwdFoldPanels) return false;
  const size_t need = align_up(tiles * 2 * dim * sizeof(double), 256) + align_up((size_t)kFoldMaxGroups * 2 * dim * sizeof(double), 256);
  if (ws_bytes < need) return false;
  static std::atomic<unsigned> next_slot{0};
  static thread_local unsigned* base = nullptr;
  static thread_local int base_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!base || base_dev != dev) {
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_bn_fwd_fold_tickets)) != hipSuccess) return false;
    base_dev = dev;
  }
  Carver cv(ws);
  f->part = cv.take<double>(tiles * 2 * dim);
  f->gpart = cv.take<double>((size_t)kFoldMaxGroups * 2 * dim);
  f->tickets = base + (size_t)(next_slot.fetch_add(1, std::memory_order_relaxed) % (kFoldSlots / 4)) * kFwdFoldPanels * (kFoldMaxGroups + 8);
  f->n = (int)n;
  return true;
}

int bn_bwd_scratch(void* ws, size_t ws_bytes, int64_t n, int64_t dim, BnBwdScratch* s) {
  if (int rc = check_args(n, dim)) return rc;
  if (ws_bytes < pgnn_bn_workspace_bytes(n, dim)) {
    set_error("batchnorm workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  s->max_blocks = stat_blocks(n);
  s->partial = cv.take<float>((size_t)s->max_blocks * 2 * dim);
  s->coef = cv.take<float>((size_t)7 * dim);
  s->gsum = cv.take<double>((size_t)kFoldMaxGroups * 2 * dim);
  s->tickets = draw_fold_tickets();
  PGNN_REQUIRE(s->tickets != nullptr, "batchnorm: ticket words unavailable");
  return PGNN_OK;
}

int bn_bwd_apply_only(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef, int relu, float* dx, int64_t lddx,
                      int64_t n, int64_t dim, hipStream_t st, uint32_t* rowmax) {
  if (int rc = check_args(n, dim)) return rc;
  PGNN_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "batchnorm: leading dimensions must be multiples of 4");
  // (four blocks per CU walking the rows instead of sixteen with one or two rows each: 0.982-0.991 against 0.990-0.996 ms per chem
  // step in four A/B pairs, profiles/r04/step_unprofiled.txt -- beside a weight-gradient product fewer, longer-lived blocks win)
  const int grid = (int)std::min<int64_t>(ceil_div(n, 4), (int64_t)num_cu() * env_knob("PGNN_BN_APPLY_BPC", 4));
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(grid), dim3(stat_threads(dim)), 0, st, dy, lddy, x, ldx, coef, relu, dx, lddx, (int)n,
                     (int)(dim / 4), make_drop(0.f, 0), rowmax);
  return check_launch("bn_bwd_apply");
}

}  // namespace pgnn

using namespace pgnn;

extern "C" {

size_t pgnn_bn_workspace_bytes(int64_t n, int64_t dim) {
  return align_up((size_t)stat_blocks(n) * 2 * dim * sizeof(float), 256) + align_up((size_t)7 * dim * sizeof(float), 256) +
         align_up((size_t)kFoldMaxGroups * 2 * dim * sizeof(double), 256);
}

int pgnn_bn_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                float* running_var, float momentum, float eps, int training, int relu, float* y, int64_t ldy,
                float* save_mean, float* save_invstd, float drop_p, uint64_t drop_seed, int64_t n, int64_t dim,
                void* ws, size_t ws_bytes, pgnn_stream stream) {
  if (int rc = check_args(n, dim)) return rc;
  PGNN_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "batchnorm: leading dimensions must be multiples of 4");
  PGNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "batchnorm: dropout probability must be in [0, 1)");
  PGNN_REQUIRE(training || (running_mean && running_var), "batchnorm eval needs running statistics");
  if (ws_bytes < pgnn_bn_workspace_bytes(n, dim)) {
    set_error("batchnorm workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws);
  const int nblk = stat_blocks(n);
  float* partial = cv.take<float>((size_t)nblk * 2 * dim);
  float* coef = cv.take<float>((size_t)7 * dim);
  const int d4 = (int)(dim / 4);
  if (training) {
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(nblk), dim3(stat_threads(dim)), (size_t)8 * dim * sizeof(float),
                       st, x, ldx, (int)n, d4, partial);
  }
  hipLaunchKernelGGL(k_bn_stats_final, dim3((int)ceil_div(dim, 4)), dim3(256), 0, st, partial, nblk, x, gamma,
                     beta, running_mean, running_var, momentum, eps, training, (int)n, (int)dim, save_mean,
                     save_invstd, coef);
  const int grid = (int)std::min<int64_t>(ceil_div(n, 4), (int64_t)num_cu() * 16);
  hipLaunchKernelGGL(k_bn_apply, dim3(grid), dim3(stat_threads(dim)), 0, st, x, ldx, coef, relu, y, ldy, (int)n, d4,
                     make_drop(drop_p, drop_seed));
  return check_launch("bn_fwd");
}

int pgnn_bn_stats_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int training, float* save_mean,
                      float* save_invstd, float* coef, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                      pgnn_stream stream) {
  if (int rc = check_args(n, dim)) return rc;
  PGNN_REQUIRE(ldx % 4 == 0 && coef, "batchnorm: leading dimension must be a multiple of 4, coef must be given");
  PGNN_REQUIRE(training || (running_mean && running_var), "batchnorm eval needs running statistics");
  if (ws_bytes < pgnn_bn_workspace_bytes(n, dim)) {
    set_error("batchnorm workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws);
  const int nblk = stat_blocks(n);
  float* partial = cv.take<float>((size_t)nblk * 2 * dim);
  if (training)
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(nblk), dim3(stat_threads(dim)), (size_t)8 * dim * sizeof(float), st, x,
                       ldx, (int)n, (int)(dim / 4), partial);
  hipLaunchKernelGGL(k_bn_stats_final, dim3((int)ceil_div(dim, 4)), dim3(256), 0, st, partial, nblk, x, gamma, beta,
                     running_mean, running_var, momentum, eps, training, (int)n, (int)dim, save_mean, save_invstd, coef);
  return check_launch("bn_stats_fwd");
}

int pgnn_bn_stats_fwd_blocks(const float* blocks, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* save_mean, float* save_invstd, float* coef, int64_t n, int64_t dim,
                             pgnn_stream stream) {
  if (int rc = check_args(n, dim)) return rc;
  PGNN_REQUIRE(blocks && coef && n < (1ll << 31), "batchnorm: block statistics and coef must be given");
  hipLaunchKernelGGL(k_bn_stats_final_blocks, dim3((int)ceil_div(dim, 4)), dim3(256), 0, (hipStream_t)stream, blocks,
                     (int)ceil_div(n, 16), gamma, beta, running_mean, running_var, momentum, eps, (int)n, (int)dim, save_mean,
                     save_invstd, coef);
  return check_launch("bn_stats_fwd_blocks");
}

int pgnn_bn_apply_fwd(const float* x, int64_t ldx, const float* coef, int relu, float* y, int64_t ldy, float drop_p,
                      uint64_t drop_seed, int64_t n, int64_t dim, pgnn_stream stream) {
  if (int rc = check_args(n, dim)) return rc;
  PGNN_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && coef, "batchnorm: leading dimensions must be multiples of 4, coef must be given");
  PGNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "batchnorm: dropout probability must be in [0, 1)");
  const int grid = (int)std::min<int64_t>(ceil_div(n, 4), (int64_t)num_cu() * 16);
  hipLaunchKernelGGL(k_bn_apply, dim3(grid), dim3(stat_threads(dim)), 0, (hipStream_t)stream, x, ldx, coef, relu, y, ldy, (int)n,
                     (int)(dim / 4), make_drop(drop_p, drop_seed));
  return check_launch("bn_apply_fwd");
}

int pgnn_bn_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                const float* beta, const float* save_mean, const float* save_invstd, int training, int relu,
                float* dx, int64_t lddx, float* dgamma, float* dbeta, float drop_p, uint64_t drop_seed, int64_t n,
                int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream) {
  return pgnn::bn_bwd_rows(dy, lddy, x, ldx, gamma, beta, save_mean, save_invstd, training, relu, dx, lddx, dgamma, dbeta, drop_p, drop_seed, n, dim,
                           ws, ws_bytes, (hipStream_t)stream, nullptr, 0);
}

}  // extern "C"

// pgnn_bn_bwd whose column sums run over `rows` [nrows] only -- the rows outside them hold dy == 0 (the caller's promise: the
// gradient of the masking head, chem/pretrain_masking.py:51-52: loss over node_rep[masked_atom_indices]); rows == NULL: all rows
int pgnn::bn_bwd_rows(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_invstd, int training, int relu, float* dx, int64_t lddx, float* dgamma, float* dbeta, float drop_p,
                      uint64_t drop_seed, int64_t n, int64_t dim, void* ws, size_t ws_bytes, hipStream_t st, const int64_t* rows, int64_t nrows) {
  if (int rc = check_args(n, dim)) return rc;
  if (rows && (nrows <= 0 || nrows > n)) rows = nullptr;
  PGNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "batchnorm: dropout probability must be in [0, 1)");
  const Drop drop = make_drop(drop_p, drop_seed);
  PGNN_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "batchnorm: leading dimensions must be multiples of 4");
  if (ws_bytes < pgnn_bn_workspace_bytes(n, dim)) {
    set_error("batchnorm workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  const int nblk = stat_blocks(rows ? nrows : n);  // (the workspace is sized for n rows)
  float* partial = cv.take<float>((size_t)stat_blocks(n) * 2 * dim);
  float* coef = cv.take<float>((size_t)7 * dim);
  const int d4 = (int)(dim / 4);
  double* gsum = cv.take<double>((size_t)kFoldMaxGroups * 2 * dim);
  const bool fold = ceil_div(nblk, kFoldGroup) <= kFoldMaxGroups && env_knob("PGNN_BN_BWD_FOLD", 1) != 0;
  unsigned* tickets = nullptr;
  if (fold) {
    tickets = draw_fold_tickets();
    PGNN_REQUIRE(tickets != nullptr, "batchnorm: ticket words unavailable");
  }
  hipLaunchKernelGGL(k_bn_bwd_partial, dim3(nblk), dim3(stat_threads(dim)), (size_t)8 * dim * sizeof(float), st, dy,
                     lddy, x, ldx, gamma, beta, save_mean, save_invstd, coef, relu, (int)n, d4, partial, drop, gsum, tickets, training,
                     dgamma, dbeta, rows, (int)nrows);
  if (!fold)
    hipLaunchKernelGGL(k_bn_bwd_final, dim3((int)ceil_div(dim, 4)), dim3(256), 0, st, partial, nblk, training, (int)n, (int)dim, gamma,
                       coef, dgamma, dbeta);
  const int grid = (int)std::min<int64_t>(ceil_div(n, 4), (int64_t)num_cu() * 16);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(grid), dim3(stat_threads(dim)), 0, st, dy, lddy, x, ldx, coef, relu, dx, lddx,
                     (int)n, d4, drop, static_cast<uint32_t*>(nullptr));
  return check_launch("bn_bwd");
}

