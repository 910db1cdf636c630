// The in-launch fold of BatchNorm-backward partial column sums, shared by k_bn_bwd_partial (batchnorm.hip) and by the transposed
// aggregation that leaves the sums of the layer below in its epilogue (aggregate.hip, round 4).
#pragma once
#include "common.h"

namespace pgnn {

constexpr int kBnMaxBlocks = 1024;
constexpr int kFoldGroup = 16;       // blocks per group of the in-launch fold
constexpr int kFoldMaxGroups = 64;   // (kBnMaxBlocks / kFoldGroup)
constexpr int kFoldSlots = 256;      // ticket sets; a launch draws the next one, every launch leaves its set zeroed

// what the fold needs: partial [nblk][2][dim] (published by every block: column sums of dyr and of dyr * xhat), scratch gsum
// [kFoldMaxGroups][2][dim] doubles, one ticket set (kFoldMaxGroups + 8 words, zero, left zero), and where the results go
struct BnBwdFold {
  const float* gamma;
  const float* save_invstd;
  float* partial;
  double* gsum;
  unsigned* tickets;
  float* coef;    // [7][dim]: a, b, mean, invstd (written by the caller's block 0), k1, k2, k3 (written here)
  float* dgamma;  // may be NULL
  float* dbeta;   // may be NULL
  int training;
  int n;          // rows of the batch
};

// Called by EVERY thread of every block (t = its index among the `nthreads` calling threads of its block, blk of nblk blocks),
// behind the block's publish() of its partial row and the publish_commit() of the publishing threads.  Blocks are grouped by
// kFoldGroup consecutive ids; the LAST block of a group to arrive adds the group's partials in block order (float64) into
// gsum[group]; the last GROUP leader to arrive adds the groups in order and finishes: dgamma, dbeta and the coefficients of
//   dx = k1 * dyr + k2 * (x - mean) + k3,   k1 = gamma * invstd;  training: k2 = -k1 * invstd * mean(dyr * xhat), k3 = -k1 * mean(dyr).
// Which block does the adding varies, what is added in which order does not.  Partials cross blocks as agent-scope stores / loads
// and relaxed tickets (common.h: no L2 write-backs).
__device__ __forceinline__ void bn_bwd_fold(const BnBwdFold& f, int dim, int blk, int nblk, int t, int nthreads) {
  __shared__ int role;
  const int ngroups = (nblk + kFoldGroup - 1) / kFoldGroup;
  const int grp = blk / kFoldGroup, gsize = min(kFoldGroup, nblk - grp * kFoldGroup);
  __syncthreads();  // this block's partial row is written (agent-scope stores, committed by their threads)
  if (t == 0) role = __hip_atomic_fetch_add(f.tickets + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)gsize - 1u ? 1 : 0;
  __syncthreads();
  if (role == 0) return;
  for (int q = t; q < 2 * dim; q += nthreads) {
    float v[kFoldGroup];  // every load of the group in flight at once (clamped, unconditional), added in block order
#pragma unroll
    for (int b = 0; b < kFoldGroup; ++b) v[b] = fetch_published(f.partial + (size_t)(grp * kFoldGroup + min(b, gsize - 1)) * 2 * dim + q);
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < kFoldGroup; ++b)
      if (b < gsize) acc += (double)v[b];
    publish(f.gsum + (size_t)grp * 2 * dim + q, acc);
  }
  publish_commit();
  __syncthreads();
  if (t == 0) {
    publish(f.tickets + 1 + grp, 0u);
    publish_commit();
    role = __hip_atomic_fetch_add(f.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ngroups - 1u ? 2 : 0;
  }
  __syncthreads();
  if (role != 2) return;
  if (t == 0) publish(f.tickets, 0u);
  for (int c = t; c < dim; c += nthreads) {
    double t1 = 0.0, t2 = 0.0;
    for (int g0 = 0; g0 < ngroups; g0 += 8) {  // eight groups' sums in flight at once, added in group order
      double u1[8], u2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int gq = min(g0 + j, ngroups - 1);
        u1[j] = fetch_published(f.gsum + (size_t)gq * 2 * dim + c);
        u2[j] = fetch_published(f.gsum + (size_t)gq * 2 * dim + dim + c);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (g0 + j < ngroups) {
          t1 += u1[j];
          t2 += u2[j];
        }
    }
    if (f.dgamma) f.dgamma[c] = (float)t2;
    if (f.dbeta) f.dbeta[c] = (float)t1;
    const float invstd = f.save_invstd[c];
    const float k1 = f.gamma[c] * invstd;
    float k2 = 0.f, k3 = 0.f;
    if (f.training) {  // dx = k1 * (dyr - s1/n - xhat * s2/n),  xhat = (x - mean) * invstd
      const float m1 = (float)(t1 / f.n), m2 = (float)(t2 / f.n);
      k2 = -k1 * invstd * m2;
      k3 = -k1 * m1;
    }
    f.coef[4 * dim + c] = k1;
    f.coef[5 * dim + c] = k2;
    f.coef[6 * dim + c] = k3;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Forward: the training-mode statistics of a BatchNorm behind a product, folded inside the product's launch (round 4).  The
// epilogue of the product holds, per 16-row block and column, S = the block's column sum and Q = the sum of squared deviations
// from the block's own mean (pgnn_linear_fwd_colstats wrote them out for k_bn_stats_final_blocks to merge); here a row tile
// merges its own blocks through LDS, publishes (mean, M2) per column, and the tiles of a column panel are merged in tile order
// by the last of every kFoldGroup tiles to arrive and then by the last group -- all with the parallel-variance formula in double,
//     n = nA + nB ; d = mB - mA ; m = mA + d nB / n ; M2 = M2A + M2B + d d nA nB / n
// -- which leaves mean / invstd / the running statistics / the affine coefficients exactly as k_bn_stats_final_blocks leaves them
// up to the association order of the merges.
struct BnFwdFold {
  const float* gamma;
  const float* beta;
  float* running_mean;  // may be NULL
  float* running_var;
  float momentum, eps;
  float* save_mean;     // [dim]
  float* save_invstd;   // [dim]
  float* coef;          // [2][dim]: y = a z + b
  double* part;         // [row tiles][2][dim]  (>= ceil(n / 64) tiles)
  double* gpart;        // [kFoldMaxGroups][2][dim]
  unsigned* tickets;    // [column panels (<= kFwdFoldPanels)][kFoldMaxGroups + 8] zero words, left zero
  int n;                // rows of the batch; 0 = no fold (the struct is unused)
};
constexpr int kFwdFoldPanels = 8;

struct Moments {
  double n = 0.0, m = 0.0, q = 0.0;
  __device__ __forceinline__ void merge(double nb, double mb, double qb) {
    if (nb <= 0.0) return;
    // (one reciprocal instead of two float64 divisions per merge: the counts are small integers, and 1 / nn enters both terms with
    // a relative error of a few 1e-16 -- the folds run on the tail of a product's launch, on its critical path)
    const double nn = n + nb, d = mb - m, r0 = __builtin_amdgcn_rcp(nn), r = r0 * (2.0 - nn * r0);  // (+ one Newton step)
    m += d * (nb * r);
    q += qb + d * d * (n * nb * r);
    n = nn;
  }
};

// Called by EVERY thread of a product's workgroup behind its epilogue.  elds [BM / 16][2][BN] floats: (S, Q) of the tile's 16-row
// blocks for its BN columns (written by the epilogue, not yet synchronised).  Tile (tile_m, tile_n) of tiles_m x . tiles covers
// rows m0 .. and columns n0 .. n0 + BN of the [M, N] result.
template <int BM, int BN>
__device__ __forceinline__ void bn_fwd_fold_tile(const BnFwdFold& f, const float* elds, int tile_m, int tiles_m, int tile_n, int m0, int n0,
                                                 int M, int N, int tid, int nthreads) {
  __shared__ int role_f;
  const int c = n0 + tid;
  const bool col = tid < BN && c < N;
  __syncthreads();  // elds complete
  if (col) {
    Moments a;
#pragma unroll
    for (int rb = 0; rb < BM / 16; ++rb) {
      const int cnt = min(16, M - (m0 + 16 * rb));
      if (cnt > 0) a.merge((double)cnt, (double)elds[(rb * 2 + 0) * BN + tid] / (double)cnt, (double)elds[(rb * 2 + 1) * BN + tid]);
    }
    publish(f.part + ((size_t)tile_m * 2 + 0) * N + c, a.m);
    publish(f.part + ((size_t)tile_m * 2 + 1) * N + c, a.q);
  }
  publish_commit();
  unsigned* tk = f.tickets + (size_t)tile_n * (kFoldMaxGroups + 8);
  const int ngroups = (tiles_m + kFoldGroup - 1) / kFoldGroup;
  const int grp = tile_m / kFoldGroup, gsize = min(kFoldGroup, tiles_m - grp * kFoldGroup);
  __syncthreads();
  if (tid == 0) role_f = __hip_atomic_fetch_add(tk + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)gsize - 1u ? 1 : 0;
  __syncthreads();
  if (role_f == 0) return;
  auto rows_of_tile = [&](int tm) { return (double)max(0, min(BM, M - tm * BM)); };
  if (col) {
    double vm[kFoldGroup], vq[kFoldGroup];  // the group's tiles in flight at once (clamped, unconditional), merged in tile order
#pragma unroll
    for (int b = 0; b < kFoldGroup; ++b) {
      const int tm = grp * kFoldGroup + min(b, gsize - 1);
      vm[b] = fetch_published(f.part + ((size_t)tm * 2 + 0) * N + c);
      vq[b] = fetch_published(f.part + ((size_t)tm * 2 + 1) * N + c);
    }
    Moments a;
#pragma unroll
    for (int b = 0; b < kFoldGroup; ++b)
      if (b < gsize) a.merge(rows_of_tile(grp * kFoldGroup + b), vm[b], vq[b]);
    publish(f.gpart + ((size_t)grp * 2 + 0) * N + c, a.m);
    publish(f.gpart + ((size_t)grp * 2 + 1) * N + c, a.q);
  }
  publish_commit();
  __syncthreads();
  if (tid == 0) {
    publish(tk + 1 + grp, 0u);
    publish_commit();
    role_f = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ngroups - 1u ? 2 : 0;
  }
  __syncthreads();
  if (role_f != 2) return;
  if (tid == 0) publish(tk, 0u);
  if (!col) return;
  Moments a;
  for (int g0 = 0; g0 < ngroups; g0 += 8) {  // eight groups in flight at once, merged in group order
    double um[8], uq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gq = min(g0 + j, ngroups - 1);
      um[j] = fetch_published(f.gpart + ((size_t)gq * 2 + 0) * N + c);
      uq[j] = fetch_published(f.gpart + ((size_t)gq * 2 + 1) * N + c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (g0 + j < ngroups) {
        const int t0 = (g0 + j) * kFoldGroup;
        a.merge((double)max(0, min(kFoldGroup * BM, M - t0 * BM)), um[j], uq[j]);
      }
  }
  const int n = f.n;
  double var = a.q / n;
  if (var < 0.0) var = 0.0;
  const float mean = (float)a.m;
  const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  if (f.running_mean) {
    const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
    f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
  }
  f.save_mean[c] = mean;
  f.save_invstd[c] = invstd;
  const float ac = invstd * f.gamma[c];
  f.coef[c] = ac;
  f.coef[N + c] = fmaf(-mean, ac, f.beta[c]);  // same expression as the backward's recomputation
}
// (batchnorm.hip) scratch of one such fold inside an op workspace of pgnn_bn_workspace_bytes (false: no room), and fresh tickets
bool bn_fwd_fold_scratch(void* ws, size_t ws_bytes, int64_t n, int64_t dim, BnFwdFold* f);

// (batchnorm.hip) scratch of one BatchNorm backward carved from an op workspace of pgnn_bn_workspace_bytes, and a fresh ticket set
struct BnBwdScratch {
  float* partial;  // [max_blocks][2][dim]
  float* coef;     // [7][dim]
  double* gsum;    // [kFoldMaxGroups][2][dim]
  unsigned* tickets;
  int max_blocks;
};
int bn_bwd_scratch(void* ws, size_t ws_bytes, int64_t n, int64_t dim, BnBwdScratch* s);
// the elementwise pass alone (no dropout): dx = k1 * dyr + k2 * (x - mean) + k3 from coef [7][dim]
// rowmax (optional): [n] words that receive the bit patterns of max |dx[i, :]| (for the two-plane product that reads dx)
int bn_bwd_apply_only(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* coef, int relu, float* dx, int64_t lddx,
                      int64_t n, int64_t dim, hipStream_t st, uint32_t* rowmax = nullptr);

// (batchnorm.hip) pgnn_bn_bwd whose column sums visit only `rows` [nrows] (everywhere else dy is zero); rows == NULL: pgnn_bn_bwd
int bn_bwd_rows(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* beta, const float* save_mean,
                const float* save_invstd, int training, int relu, float* dx, int64_t lddx, float* dgamma, float* dbeta, float drop_p,
                uint64_t drop_seed, int64_t n, int64_t dim, void* ws, size_t ws_bytes, hipStream_t st, const int64_t* rows, int64_t nrows);

// (aggregate.hip) pgnn_neighbor_sum (unweighted) whose launch ALSO leaves, for the BatchNorm whose input gradient it computes
// (out = dL/dy of the layer below, y = relu?(BatchNorm(z))): the column sums of the backward, folded -- coef / dgamma / dbeta as
// pgnn_bn_bwd's first two launches leave them -- so that layer's BatchNorm backward is bn_bwd_apply_only.  *fused says whether
// this launch could do it (feature width 300, batches below the Infinity-Cache policy switch); if not it is the plain sum.
struct BnBwdTail {
  const float* z;  // [n, ldz] pre-activations of that BatchNorm
  int64_t ldz;
  const float* gamma;
  const float* beta;
  const float* save_mean;
  const float* save_invstd;
  int relu, training;
  BnBwdScratch scratch;
  float* dgamma;
  float* dbeta;
};
int chem_aggregate_fwd_amax(const float* x, int64_t ldx, const float* coef, int relu, const int32_t* in_ptr, const int32_t* in_src,
                            const uint8_t* in_code, const float* emb1, const float* emb2, float* out, int64_t ldo, int64_t n, int64_t dim,
                            uint32_t* amax, bool* done, hipStream_t st);
int neighbor_sum_bn_bwd(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, float* out, int64_t ldo, int64_t n,
                        int64_t dim, const BnBwdTail& tail, bool* fused, hipStream_t st);

}  // namespace pgnn
