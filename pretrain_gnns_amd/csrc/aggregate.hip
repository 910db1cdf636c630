// Aggregation kernels of the message-passing hot path (gfx950, wave64).
//
// Three implementations of out[i] = sum_e w_e (x[nbr_e] + T[code_e]) + w_ii (x[i] + T[self]) live here,
// selected in launch_aggregate (PGNN_AGG_VARIANT overrides for A/B measurements):
//   k_aggregate_dma  (3) production, unweighted, D <= 320: loader wave + LDS ring + consumer waves
//   k_aggregate_grp  (1) D/4 threads per node, neighbour rows from L2: GCN weights, wide rows
//   k_aggregate      (0) one wave per node: the first kernel, kept as the bit-exact A/B reference
// The small reductions that surround them (edge-feature matmuls, embedding, segment sums) follow.
//
// Work decomposition of the helper kernels: ONE WAVE OWNS ONE NODE ROW AT A TIME.  A row of
// D fp32 features is D/4 float4 chunks; lane l owns chunks l, l+64, ... (R = ceil(D/256) of them,
// D = 300 -> 64 + 11 lanes), so every neighbour-row read and every result-row write is a fully
// coalesced 16 B/lane access, partial sums live in registers, and the only scattered accesses are
// whole 1200-byte rows.  Neighbour indices of a row are fetched by the lanes in one coalesced load
// and broadcast with v_readlane, which makes the row base address wave-uniform (SGPR base + lane
// offset).  Waves are persistent and take contiguous node chunks; the block->chunk map is
// XCD-aware so that the rows a molecule's atoms gather from each other stay in one XCD's L2.
// All reductions are sequential in a fixed order: results are bitwise reproducible, and the chem
// aggregation reproduces the reference's CPU scatter_add order exactly.
#include <stdlib.h>

#include "bn_fold.h"
#include "common.h"

namespace pgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kNumCodes = 18;       // bond type (6) x bond direction (3)
constexpr int kSelfLoopCode = 4 * 3 + 0;  // chem/model.py:43: self loop = bond type 4, direction 0

template <int R>
struct Row {
  float4 v[R];
};

template <int R>
__device__ __forceinline__ void row_load(Row<R>& r, const float* __restrict__ base, int lane, int d4) {
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int c = lane + k * kWave;
    if (k + 1 < R || c < d4) r.v[k] = reinterpret_cast<const float4*>(base)[c];
  }
}

template <int R>
__device__ __forceinline__ void row_store(const Row<R>& r, float* __restrict__ base, int lane, int d4) {
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int c = lane + k * kWave;
    if (k + 1 < R || c < d4) reinterpret_cast<float4*>(base)[c] = r.v[k];
  }
}

template <int R>
__device__ __forceinline__ void row_zero(Row<R>& r) {
#pragma unroll
  for (int k = 0; k < R; ++k) r.v[k] = f4_zero();
}

// persistent-wave node range: wave w of the (XCD-remapped) grid owns chunks w, w+W, ...
struct WaveSched {
  int wave, nwaves, lane;
  __device__ __forceinline__ explicit WaveSched(int nxcd) {
    const int b = xcd_remap(blockIdx.x, gridDim.x, nxcd);
    wave = b * kWavesPerBlock + (threadIdx.x >> 6);
    nwaves = gridDim.x * kWavesPerBlock;
    lane = lane_id();
  }
};

// ---------------------------------------------------------------------------------------------
// out[i] = sum_e w_e*(x[nbr_e] (+ T[code_e])) + w_ii*(x[i] (+ T[self]))
// TABLE : chem bond-embedding table T = emb1[a0]+emb2[a1] (18 x D) built once per block in LDS
// WEIGHT: GCN symmetric normaliser w_e = dinv[i]*dinv[nbr_e]
// ---------------------------------------------------------------------------------------------
template <int R, bool TABLE, bool WEIGHT, int K>
__device__ __forceinline__ void gather_k(Row<R>& acc, const float* __restrict__ x, int64_t ldx,
                                         int nbrs, int codes, float ws, float di, int j0,
                                         const float* __restrict__ T, int dim, int lane, int d4) {
#pragma clang fp contract(off)
  Row<R> v[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const int s = bcast_i32(nbrs, j0 + u);
    row_load<R>(v[u], x + (int64_t)s * ldx, lane, d4);
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    float w = 1.f;
    if (WEIGHT) w = di * __int_as_float(bcast_i32(__float_as_int(ws), j0 + u));
    const float* trow = nullptr;
    if (TABLE) trow = T + bcast_i32(codes, j0 + u) * dim;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int c = lane + k * kWave;
      if (k + 1 < R || c < d4) {
        float4 m = v[u].v[k];
        if (TABLE) m = f4_add(m, reinterpret_cast<const float4*>(trow)[c]);
        if (WEIGHT) m = f4_scale(m, w);
        acc.v[k] = f4_add(acc.v[k], m);
      }
    }
  }
}

template <int R, bool TABLE, bool WEIGHT>
__global__ void __launch_bounds__(kBlock)
k_aggregate(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
            const int32_t* __restrict__ nbr, const uint8_t* __restrict__ code,
            const float* __restrict__ emb1, const float* __restrict__ emb2,
            const float* __restrict__ dinv, float* __restrict__ out, int64_t ldo, int n, int dim,
            int nodes_per_chunk, int nxcd) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float T[];  // [18][dim] when TABLE
  if (TABLE) {
    for (int q = threadIdx.x; q < kNumCodes * dim; q += kBlock) {
      const int c = q / dim, d = q - c * dim;
      T[q] = emb1[(c / 3) * dim + d] + emb2[(c % 3) * dim + d];
    }
    __syncthreads();
  }
  const WaveSched ws(nxcd);
  const int lane = ws.lane, d4 = dim >> 2;
  const int nchunks = (n + nodes_per_chunk - 1) / nodes_per_chunk;
  for (int chunk = ws.wave; chunk < nchunks; chunk += ws.nwaves) {
    const int n0 = chunk * nodes_per_chunk;
    const int cnt = min(nodes_per_chunk, n - n0);
    const int myptr = lane <= cnt ? ptr[n0 + lane] : 0;
    for (int k = 0; k < cnt; ++k) {
      const int i = n0 + k;
      const int beg = bcast_i32(myptr, k), end = bcast_i32(myptr, k + 1);
      Row<R> self;
      row_load<R>(self, x + (int64_t)i * ldx, lane, d4);
      float di = 1.f;
      if (WEIGHT) di = dinv[i];
      Row<R> acc;
      row_zero<R>(acc);
      for (int base = beg; base < end; base += kWave) {
        const int m = min(kWave, end - base);
        int nbrs = 0, codes = 0;
        float wsrc = 0.f;
        if (lane < m) {
          nbrs = nbr[base + lane];
          if (TABLE) codes = code[base + lane];
          if (WEIGHT) wsrc = dinv[nbrs];
        }
        int j = 0;
        for (; j + 4 <= m; j += 4)
          gather_k<R, TABLE, WEIGHT, 4>(acc, x, ldx, nbrs, codes, wsrc, di, j, T, dim, lane, d4);
        switch (m - j) {
          case 3: gather_k<R, TABLE, WEIGHT, 3>(acc, x, ldx, nbrs, codes, wsrc, di, j, T, dim, lane, d4); break;
          case 2: gather_k<R, TABLE, WEIGHT, 2>(acc, x, ldx, nbrs, codes, wsrc, di, j, T, dim, lane, d4); break;
          case 1: gather_k<R, TABLE, WEIGHT, 1>(acc, x, ldx, nbrs, codes, wsrc, di, j, T, dim, lane, d4); break;
          default: break;
        }
      }
      // self loop goes last: the reference appends self loops after the real edges
      const float wself = di * di;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int c = lane + r * kWave;
        if (r + 1 < R || c < d4) {
          float4 m = self.v[r];
          if (TABLE) m = f4_add(m, reinterpret_cast<const float4*>(T + kSelfLoopCode * dim)[c]);
          if (WEIGHT) m = f4_scale(m, wself);
          acc.v[r] = f4_add(acc.v[r], m);
        }
      }
      row_store<R>(acc, out + (int64_t)i * ldo, lane, d4);
    }
  }
}

inline int pick_grid(int64_t n, int blocks_per_cu, int* nodes_per_chunk) {
  // spread small inputs over many waves, give big inputs 16-node chunks
  const int64_t max_waves = (int64_t)num_cu() * blocks_per_cu * kWavesPerBlock;
  int64_t npc = ceil_div(n, max_waves);
  npc = std::min<int64_t>(std::max<int64_t>(npc, 1), 16);
  *nodes_per_chunk = (int)npc;
  const int64_t chunks = ceil_div(n, npc);
  const int64_t blocks = std::min<int64_t>(ceil_div(chunks, kWavesPerBlock), (int64_t)num_cu() * blocks_per_cu);
  return (int)std::max<int64_t>(blocks, 1);
}

// ---------------------------------------------------------------------------------------------
// Group-per-node variant: a node row is owned by dim/4 consecutive THREADS (75 for D = 300), so a
// 320-thread block works on 4 nodes at once with one float4 accumulator per thread (94 % of the
// lanes busy instead of 59 % for the wave-per-node mapping, ~40 VGPRs -> 8 waves/SIMD).  The
// groups of a block take interleaved consecutive nodes, so the rows neighbouring atoms gather from
// each other are in flight in the same CU.  Per-lane control flow (a wave straddles two nodes);
// neighbour loads of one node are issued four at a time before the (ordered) accumulation.
// ---------------------------------------------------------------------------------------------
constexpr int kGrpBlock = 320;

inline int env_int(const char* name, int dflt) { return env_knob(name, dflt); }

template <bool TABLE, bool WEIGHT>
__global__ void __launch_bounds__(kGrpBlock)
k_aggregate_grp(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                const int32_t* __restrict__ nbr, const uint8_t* __restrict__ code,
                const float* __restrict__ emb1, const float* __restrict__ emb2,
                const float* __restrict__ dinv, float* __restrict__ out, int64_t ldo, int n, int dim,
                int groups, int nodes_per_block, int front, int nxcd) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float T[];  // [18][dim] when TABLE
  if (TABLE) {
    for (int q = threadIdx.x; q < kNumCodes * dim; q += kGrpBlock) {
      const int c = q / dim, d = q - c * dim;
      T[q] = emb1[(c / 3) * dim + d] + emb2[(c % 3) * dim + d];
    }
    __syncthreads();
  }
  const int gs = dim >> 2;
  const int g = threadIdx.x / gs, c4 = threadIdx.x - g * gs;
  if (g >= groups) return;
  // Scheduling.  front == 0: block b owns one contiguous node range.  front == 1: the node array is
  // cut into one contiguous slab per XCD (blocks are dealt to XCDs round-robin: xcd = blockIdx % 8);
  // the blocks of an XCD sweep their slab together as ONE narrow front (`groups` nodes per block per
  // step), so neighbour rows are shared through that XCD's L2 while HBM sees 8 sequential streams.
  int i_first, i_end, i_step;
  if (front) {
    const int xcd = blockIdx.x % nxcd, j = blockIdx.x / nxcd, nbx = gridDim.x / nxcd;
    const int slab = ((n + nxcd - 1) / nxcd + groups - 1) / groups * groups;
    i_first = xcd * slab + j * groups + g;
    i_end = min(n, (xcd + 1) * slab);
    i_step = nbx * groups;
  } else {
    const int b = xcd_remap(blockIdx.x, gridDim.x, nxcd);
    i_first = b * nodes_per_block + g;
    i_end = min(n, b * nodes_per_block + nodes_per_block);
    i_step = groups;
  }
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  const float4* __restrict__ T4 = reinterpret_cast<const float4*>(T);
  const int64_t ldx4 = ldx >> 2, ldo4 = ldo >> 2;
  for (int i = i_first; i < i_end; i += i_step) {
    const int beg = ptr[i], end = ptr[i + 1];
    const float4 self = x4[(int64_t)i * ldx4 + c4];
    float di = 1.f;
    if (WEIGHT) di = dinv[i];
    float4 acc = f4_zero();
    for (int p = beg; p < end; p += 4) {
      int s[4], cd[4];
      float w[4];
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p + j < end) {
          s[j] = nbr[p + j];
          if (TABLE) cd[j] = code[p + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p + j < end) {
          v[j] = x4[(int64_t)s[j] * ldx4 + c4];
          if (WEIGHT) w[j] = di * dinv[s[j]];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p + j < end) {
          float4 m = v[j];
          if (TABLE) m = f4_add(m, T4[cd[j] * gs + c4]);
          if (WEIGHT) m = f4_scale(m, w[j]);
          acc = f4_add(acc, m);
        }
      }
    }
    float4 m = self;
    if (TABLE) m = f4_add(m, T4[kSelfLoopCode * gs + c4]);
    if (WEIGHT) m = f4_scale(m, di * di);
    acc = f4_add(acc, m);
    reinterpret_cast<float4*>(out)[(int64_t)i * ldo4 + c4] = acc;
  }
}

template <bool TABLE, bool WEIGHT>
int launch_aggregate_grp(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr,
                         const uint8_t* code, const float* emb1, const float* emb2, const float* dinv,
                         float* out, int64_t ldo, int64_t n, int64_t dim, hipStream_t st) {
  const int gs = (int)(dim / 4);
  if (gs > kGrpBlock) {
    set_error("feature width %lld > %d not supported", (long long)dim, 4 * kGrpBlock);
    return PGNN_ERR_ARG;
  }
  const int groups = kGrpBlock / gs;
  const size_t lds = TABLE ? (size_t)kNumCodes * dim * sizeof(float) : 0;
  const int bpc = env_int("PGNN_AGG_BLOCKS_PER_CU", TABLE ? 6 : 6);
  const int64_t max_blocks = (int64_t)num_cu() * bpc;
  int64_t npb = std::max<int64_t>(ceil_div(n, max_blocks), groups);
  npb = ceil_div(npb, groups) * groups;
  int grid = (int)ceil_div(n, npb);
  const int front = env_int("PGNN_AGG_FRONT", 1);
  const int nx = num_xcd();
  if (front) grid = (int)std::max<int64_t>(nx, std::min<int64_t>(max_blocks, ceil_div(ceil_div(n, groups), nx) * nx) / nx * nx);
  allow_big_lds((const void*)k_aggregate_grp<TABLE, WEIGHT>, lds);
  hipLaunchKernelGGL((k_aggregate_grp<TABLE, WEIGHT>), dim3(grid), dim3(kGrpBlock), lds, st, x, ldx, ptr, nbr, code,
                     emb1, emb2, dinv, out, ldo, (int)n, (int)dim, groups, (int)npb, front, nx);
  return check_launch("aggregate_grp");
}

// ---------------------------------------------------------------------------------------------
// Producer/consumer LDS-DMA variant (production path for unweighted aggregation, D <= 320).
//   * wave CW (the last wave) is a LOADER: each step it streams the next 8 feature rows and the next
//     step's edge indices / bond codes straight into LDS with global_load_lds (no VGPR round trip).
//     It never stores and issues nothing else, so "s_waitcnt vmcnt(0); barrier" is exact: every row of
//     x leaves HBM once per block, as contiguous 8*D*4-byte bursts, one step ahead of its use.
//   * waves 0..CW-1 are CONSUMERS: 75 threads per node (D = 300), 8 nodes per step.  Neighbour rows
//     and bond-table rows are ds_read_b128 from LDS, results go out as coalesced stores that are never
//     waited for.  A chunk of edges whose source row is outside the 24-row window (or beyond the 64
//     staged edge slots) takes a wave-uniform slow path that reads from global memory instead.
//   * ring = 4 regions of 8 rows: compute(s) reads regions s-1, s, s+1 while the loader fills s+2.
// One barrier per step.  Sums run in original edge order, self loop last: bit-exact vs the reference.
// ---------------------------------------------------------------------------------------------
constexpr int kDmaG = 8;                      // nodes per step
constexpr int kDmaEdges = 64;                 // staged edge slots per step
constexpr int kDmaMaxNodes = 1024;            // nodes per block

#define PGNN_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PGNN_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// s_waitcnt vmcnt(n) needs an immediate; n = (prefetch depth - 1) * DMA instructions per step
__device__ __forceinline__ void wait_vmcnt(int n) {
#define PGNN_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    PGNN_W(1) PGNN_W(2) PGNN_W(3) PGNN_W(4) PGNN_W(5) PGNN_W(6) PGNN_W(7) PGNN_W(8) PGNN_W(9) PGNN_W(10) PGNN_W(11)
    PGNN_W(12) PGNN_W(13) PGNN_W(14) PGNN_W(15) PGNN_W(16) PGNN_W(17) PGNN_W(18) PGNN_W(19) PGNN_W(20) PGNN_W(21)
    PGNN_W(22) PGNN_W(23) PGNN_W(24) PGNN_W(25) PGNN_W(26) PGNN_W(27) PGNN_W(28) PGNN_W(29) PGNN_W(30) PGNN_W(31)
    PGNN_W(32) PGNN_W(33) PGNN_W(34) PGNN_W(35) PGNN_W(36)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef PGNN_W
}

// P = prefetch depth in steps; the ring holds P+3 regions of 8 rows: compute(s) reads regions
// s-1, s, s+1, regions s+2..s+P are landed or in flight, region s+1+P is being issued.
// NROW = ceil(8*gs/64) row-DMA instructions per step when known at compile time (their per-lane
// byte offsets are then hoisted into registers: the loader issues a step with ~2 instructions per
// KiB); NROW = 0 selects a generic run-time loop for other feature widths.
// PRE: the rows of x are pre-activations z of the previous layer's BatchNorm; every row becomes relu?(a*z + b) with the SAME
// fmaf/fmaxf expression k_bn_apply uses, so the previous layer's output is never materialised and the sums are bit-identical to
// aggregating the materialised tensor.  Round 6: a landed ring region is activated ONCE, IN PLACE, BY THE LOADER WAVE between its
// vmcnt wait and the barrier that publishes the region (8 rows = 16 ds_read_b128 / ds_write_b128 of a wave that would otherwise sit at
// the barrier) -- the consumers read activated rows, their inner loop and register count are the plain instance's, and two blocks
// stay resident per CU (rounds 3-5 activated every gathered row in the consumers, 3.2 times per row: 86 VGPRs = 5 waves per SIMD = ONE
// 11-wave block per CU, 337.7 us against the plain instance's 215 us on the same bytes).  Rows that bypass the ring (far sources) are
// activated where they are used, with the coefficients kept in LDS.  POL bit 5 keeps the consumer-side activation for A/B builds.
// POL (cache / scheduling policy, A/B-measured through PGNN_DMA_POL; 0 = none):
//   bit 0: row DMAs carry the non-temporal hint (every row of x is read once per block)
//   bit 1: result rows are stored non-temporally (written once, never re-read by this kernel)
//   bit 2: the loader wave runs at s_setprio 3
//   bit 3: instrumentation -- lane 0 of the loader and thread 0 accumulate s_memtime deltas of their phases into
//          prof[block][8] = {loader vmcnt wait, loader barrier, loader issue, consumer barrier, consumer work,
//          block total, steps, 0} (pgnn_debug_aggregate_profile)
typedef float v4f_t __attribute__((ext_vector_type(4)));
// WEIGHT: GCN symmetric normaliser w_e = dinv[i] * dinv[src_e], w_ii = dinv[i]^2 (chem/model.py:73-82,104).  dinv of the
// block's rows and of the 8-row halo either side sits in LDS (every source inside the ring window is covered), the
// products are formed exactly as k_aggregate_grp forms them: the two kernels are bit-identical.
// TAIL (round 4; the transposed instance of the GIN stack's backward, out = dL/dy of the layer below): every consumer thread also
// reads its float4 of that layer's BatchNorm input z, forms dyr = the row it has just summed, masked by the recomputed ReLU, and
// keeps the column sums of dyr and dyr * xhat over its rows; the block folds its eight node slots through LDS, publishes one
// partial row and joins bn_bwd_fold (bn_fold.h) -- the launch leaves what pgnn_bn_bwd's partial-sum launch left (coef, dgamma,
// dbeta), and that launch (14-25 us on the backward's critical path, beside a weight-gradient product) is gone.  The rows are
// summed in another order than k_bn_bwd_partial sums them (per thread: rows g, g + 8, ...; then the eight slots; then the blocks
// in order): fixed, so still deterministic, equal to fp32 rounding.
struct AggTail {
  const float* z;
  int64_t ldz;
  const float* gamma;
  const float* beta;
  const float* save_mean;
  const float* save_invstd;
  int relu;
  BnBwdFold fold;
  uint32_t* amax;  // any instance, optional: [n] words that receive the bit patterns of max |out[i, :]| (the row maxima the two-plane
                   // product behind this aggregation would otherwise take in a pass of its own); plain stores, one per row
};

// The block's end of the tail, entered by EVERY thread of the block -- the loader wave too (with nothing to add): no wave of a block
// that still has a barrier ahead may have ended (a barrier counts the surviving waves only, but a workgroup that is saved and
// restored between two processes' time slices with one wave gone and the others parked at a barrier is not a state to rely on).
__device__ __forceinline__ void agg_tail_finish(const AggTail& tail, float* redL, bool active, int g, int c4, float4 s1, float4 s2, int dim) {
  const int t = threadIdx.x, nt = blockDim.x;
  if (active) {
    reinterpret_cast<float4*>(redL + (g * 2 + 0) * dim)[c4] = s1;
    reinterpret_cast<float4*>(redL + (g * 2 + 1) * dim)[c4] = s2;
  }
  __syncthreads();
  float* prow = tail.fold.partial + (size_t)blockIdx.x * 2 * dim;
  for (int q = t; q < 2 * dim; q += nt) {  // q < dim: sums of dyr, else of dyr * xhat; the eight node slots in order
    const int h = q >= dim ? 1 : 0, c = q - h * dim;
    float a0 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a0 += redL[(k * 2 + h) * dim + c];
    publish(prow + q, a0);
  }
  publish_commit();
  bn_bwd_fold(tail.fold, dim, blockIdx.x, gridDim.x, t, nt);
}

template <bool TABLE, int P, int NROW, bool PRE, int POL, bool WEIGHT, bool TAIL = false>
__global__ void __launch_bounds__(704, ((PRE && (POL & 32) == 0) || TAIL) ? 6 : 1)
k_aggregate_dma(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                const int32_t* __restrict__ nbr, const uint8_t* __restrict__ code,
                const float* __restrict__ emb1, const float* __restrict__ emb2, float* __restrict__ out,
                int64_t ldo, int n, int dim, int npb, const float* __restrict__ pre_coef, int pre_relu,
                unsigned long long* __restrict__ prof, const float* __restrict__ dinv, AggTail tail) {
#pragma clang fp contract(off)
  // POL bit 4 (PF): source rows OUTSIDE the ring window are fetched into registers ONE STEP AHEAD of their use (the step's edge
  // slots are then staged one step earlier), instead of on the spot behind a wave-uniform branch that waits a memory round trip
  // per batch: at 5.6 % such edges the on-the-spot path ran at 0.46 of the HBM roofline against 0.63 without any (bench.py's
  // aggregation_robustness leg).  Two rows per node and step; a third one, or a step with more than 64 edges, still takes the branch.
  // The loader publishes one ballot per step (which staged sources are far), so a batch without such bonds pays a two-word LDS read
  // per step for it and nothing else: 0.62 either way, 0.52 / 0.47 / 0.41 at 5.6 / 13 / 33 % far bonds.
  constexpr bool PF = (POL & 16) != 0;
  constexpr int EL = PF ? 1 : 0;  // steps by which the edge slots lead the rows
  static_assert(!(PF && WEIGHT), "the prefetch variant does not carry the GCN normaliser of a far row");
  constexpr int NREG = P + 3, NBUF = P + 1 + EL;
  constexpr int AUX = (POL & 1) ? 2 : 0;
  constexpr bool PROF = (POL & 8) != 0;
  constexpr bool LACT = PRE && (POL & 32) == 0;  // the loader activates landed regions in place (POL bit 5: the consumers do, per gathered row)
  static_assert(!(PRE && (WEIGHT || TAIL)), "the BatchNorm-on-read coefficients sit where dinv / the tail's column sums would");
  unsigned long long t_begin = 0;
  if (PROF) t_begin = __builtin_readcyclecounter();
  extern __shared__ __align__(16) float smem[];
  const int gs = dim >> 2;
  const int row_f4 = kDmaG * gs;  // float4 per ring region
  float* T = smem;                                                                   // [18][dim]
  float4* ring = reinterpret_cast<float4*>(smem + (TABLE ? kNumCodes * dim : 0));    // [NREG][8][gs]
  int* ptrL = reinterpret_cast<int*>(ring + NREG * row_f4);                          // [npb+1]
  int* idxL = ptrL + (kDmaMaxNodes + 4);                                             // [NBUF][64]
  int* codeL = idxL + NBUF * kDmaEdges;                                              // [NBUF][64] (byte DMA lands as dwords)
  float* dinvL = reinterpret_cast<float*>(codeL + NBUF * kDmaEdges);                 // [8 (nsteps + 2)] rows n0-8 .. (WEIGHT)
  int* farL = codeL + NBUF * kDmaEdges;  // (PF; never together with WEIGHT) [NBUF][2]: bit k = staged edge slot k of the step is far
  unsigned* amaxL = reinterpret_cast<unsigned*>(farL + 16);  // (tail.amax) [2][8]: the row maxima of the step in flight / being flushed
  float* redL = reinterpret_cast<float*>(farL + 32);  // (TAIL; never together with WEIGHT) [8][2][dim]: the node slots' column sums
  float* coefL = redL;                                // (PRE; never together with TAIL or WEIGHT) [2][dim]: a, b of y = a z + b
  float* tailL = redL + kDmaG * 2 * dim + 16;         // (TAIL) [4][dim]: a, b, mean, invstd of the BatchNorm below
  const float4* __restrict__ T4 = reinterpret_cast<const float4*>(T);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  const int64_t ldx4 = ldx >> 2, ldo4 = ldo >> 2;

  const int cthreads = (int)blockDim.x - kWave;  // consumer threads
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * npb, n1 = min(n, n0 + npb);
  const int cnt = n1 - n0;
  const int nsteps = (cnt + kDmaG - 1) / kDmaG;
  auto slot_of = [&](int r) { return (((r >> 3) % NREG) << 3) + (r & 7); };

  if (t >= cthreads) {
    // ------------------------------------------------------------------ loader wave
    if (POL & 4) __builtin_amdgcn_s_setprio(3);
    const int lane = t - cthreads;
    const int ne = ptr[n];  // total edge count (clamps the staged-edge reads at the array end)
    const int nrow = (row_f4 + kWave - 1) / kWave;
    const int K = nrow + (ne > 0 ? (TABLE ? 2 : 1) : 0);  // DMA instructions issued per step
    const int g_lane = lane / gs, c_lane = lane - g_lane * gs;
    constexpr int NR = NROW > 0 ? NROW : 1;
    unsigned off[NR];  // byte offset of this lane's float4 inside an 8-row step, per DMA instruction
    bool val[NR];
    if (NROW > 0) {
      int g = g_lane, c4 = c_lane;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        val[k] = k * kWave + lane < row_f4;
        off[k] = (unsigned)(g * (int)ldx * 4 + c4 * 16);
        c4 += kWave;
        while (c4 >= gs) { c4 -= gs; ++g; }
      }
    }
    auto issue_rows = [&](int q) {  // load-step q: rows n0 + 8q .. +7 -> ring region (always nrow instructions)
      const int r0 = n0 + q * kDmaG;
      float4* dst0 = ring + (((max(r0, 0)) >> 3) % NREG) * row_f4;
      if (NROW > 0 && r0 >= 0 && r0 + kDmaG <= n) {  // wave-uniform fast path: SGPR row base + hoisted lane offsets
        const char* base = reinterpret_cast<const char*>(x) + (int64_t)r0 * ldx * 4;
#pragma unroll
        for (int k = 0; k < NR; ++k)  // only the last instruction of a step can be partial
          if (k + 1 < NR || val[k])
            __builtin_amdgcn_global_load_lds(PGNN_GPTR(base + off[k]), PGNN_LPTR(dst0 + k * kWave), 16, 0, AUX);
        return;
      }
      int g = g_lane, c4 = c_lane;
      for (int k = 0; k < nrow; ++k) {
        if (k * kWave + lane < row_f4) {
          const int r = min(max(r0 + g, 0), n - 1);  // out-of-range rows: harmless duplicates, never read
          __builtin_amdgcn_global_load_lds(PGNN_GPTR(x4 + (int64_t)r * ldx4 + c4), PGNN_LPTR(dst0 + k * kWave), 16, 0, AUX);
        }
        c4 += kWave;
        while (c4 >= gs) { c4 -= gs; ++g; }
      }
    };
    auto issue_edges = [&](int s, int e0) {  // edge slots of step s -> idxL/codeL[s % NBUF]
      if (ne == 0) return;
      const int p = min(e0 + lane, ne - 1);
      __builtin_amdgcn_global_load_lds(PGNN_GPTR(nbr + p), PGNN_LPTR(idxL + (s % NBUF) * kDmaEdges), 4, 0, 0);
      if (TABLE)
        __builtin_amdgcn_global_load_lds(PGNN_GPTR(code + p), PGNN_LPTR(codeL + (s % NBUF) * kDmaEdges), 1, 0, 0);
    };
    // (LACT) lane l owns float4 column l of a region's eight rows (lanes < gs); the columns from 64 on -- 8 x (gs - 64) float4, 88 at
    // D = 300 -- are dealt flat over the lanes, two per lane at most (gs <= 80), so a region costs 10 ds_read_b128 + 10 ds_write_b128
    // instead of 16 + 16 with eleven lanes live in half of them.  Inline asm for the LDS traffic: behind a C++ ds_read hipcc drains
    // vmcnt to 0 on this path (every DMA in flight), and the waits below are the only ones needed.
    v4f_t ca0 = {0.f, 0.f, 0.f, 0.f}, cb0 = ca0;
    const int wide = max(gs - kWave, 0), nwide = kDmaG * wide;  // float4 columns past the first 64, and how many float4 that is per region
    const bool act0 = lane < gs, act1 = lane < nwide, act2 = lane + kWave < nwide;
    unsigned woff1 = 0, woff2 = 0;  // byte offsets of this lane's one or two wide float4 inside a region,
    unsigned wc1 = 0, wc2 = 0;      // and the LDS addresses of their columns' a (b: + dim * 4) in coefL (kept there: registers are the consumers')
    if (LACT) {
      const v4f_t* pc = reinterpret_cast<const v4f_t*>(pre_coef);
      const unsigned cl = (unsigned)(unsigned long long)PGNN_LPTR(coefL);
      if (act0) { ca0 = pc[lane]; cb0 = pc[gs + lane]; }
      if (act1) { const int r = lane / wide, c = kWave + lane - r * wide; wc1 = cl + (unsigned)c * 16u; woff1 = (unsigned)(r * gs + c) * 16u; }
      if (act2) { const int i = lane + kWave, r = i / wide, c = kWave + i - r * wide; wc2 = cl + (unsigned)c * 16u; woff2 = (unsigned)(r * gs + c) * 16u; }
    }
    auto act4 = [&](v4f_t& v, const v4f_t& a, const v4f_t& b) {  // the expression of k_bn_apply
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float y = fmaf(a[c], v[c], b[c]);
        if (pre_relu) y = fmaxf(y, 0.f);
        v[c] = y;
      }
    };
    auto activate = [&](int q) {  // the region of load-step q, landed and not yet published
      const int r0 = n0 + q * kDmaG;
      const unsigned region = (unsigned)(unsigned long long)PGNN_LPTR(ring + (((max(r0, 0)) >> 3) % NREG) * row_f4);
      const unsigned base = region + (unsigned)lane * 16u, rb = (unsigned)gs * 16u;
      v4f_t w1 = {0.f, 0.f, 0.f, 0.f}, w2 = w1, ca1 = w1, cb1 = w1, ca2 = w1, cb2 = w1;
      const unsigned cbo = (unsigned)dim * 4u;
      if (act1) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(w1) : "v"(region + woff1) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(ca1) : "v"(wc1) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(cb1) : "v"(wc1 + cbo) : "memory");
      }
      if (act2) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(w2) : "v"(region + woff2) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(ca2) : "v"(wc2) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(cb2) : "v"(wc2 + cbo) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w1), "+v"(w2), "+v"(ca1), "+v"(cb1), "+v"(ca2), "+v"(cb2)::"memory");
      if (act1) { act4(w1, ca1, cb1); asm volatile("ds_write_b128 %0, %1" ::"v"(region + woff1), "v"(w1) : "memory"); }
      if (act2) { act4(w2, ca2, cb2); asm volatile("ds_write_b128 %0, %1" ::"v"(region + woff2), "v"(w2) : "memory"); }
      if (act0) {
#pragma unroll
        for (int h = 0; h < kDmaG; h += 4) {  // four rows in flight (the register count of the kernel stays the consumers')
          v4f_t v[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(v[g]) : "v"(base + (h + g) * rb) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            act4(v[g], ca0, cb0);
            asm volatile("ds_write_b128 %0, %1" ::"v"(base + (h + g) * rb), "v"(v[g]) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the barrier that follows publishes the region)
    };
    if (n0 > 0) issue_rows(-1);
    for (int q = 0; q <= P; ++q) issue_rows(q);
    for (int q = 0; q < P + EL; ++q) issue_edges(q, ptr[min(n0 + q * kDmaG, n1)]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // prologue: T, ptrL, coefL (consumers) and the first window (loader) are in LDS
    if (LACT) {
      if (n0 > 0) activate(-1);
      activate(0);  // (region s + 1 is step s's, below)
      __syncthreads();
    }
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0;
    for (int s = 0; s < nsteps; ++s) {
      unsigned long long t0 = 0, t1 = 0, t2 = 0;
      if (PROF) t0 = __builtin_readcyclecounter();
      wait_vmcnt(min(s, P - 1) * K);  // everything issued >= P steps ago (rows <= s+1, edges(s)) has landed
      if (PF) {
        // which of step s + 1's staged sources lie outside ITS ring window: one ballot, published before B(s), so that a consumer
        // looks ahead only when there is something to fetch (the look-ahead cost 14 % of the rate on batches without any far bond).
        // The slot is read with inline asm: behind a ds_read in C++ hipcc waits for vmcnt(0) on this path, i.e. for every DMA in flight.
        const int sn = s + 1;
        const unsigned addr = (unsigned)(unsigned long long)PGNN_LPTR(idxL + (sn % NBUF) * kDmaEdges + lane);
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        const int b2 = n0 + sn * kDmaG;
        const int ne_step = ptrL[min((sn + 1) * kDmaG, cnt)] - ptrL[min(sn * kDmaG, cnt)];  // (slots past it hold later steps' edges)
        const unsigned long long m = __ballot(lane < ne_step && (v < max(b2 - kDmaG, 0) || v >= min(b2 + 2 * kDmaG, n)));
        if (lane == 0) {
          farL[(sn % NBUF) * 2] = (int)(unsigned)m;
          farL[(sn % NBUF) * 2 + 1] = (int)(unsigned)(m >> 32);
        }
      }
      if (LACT) activate(s + 1);      // rows <= s + 1 have landed; the consumers of step s are the first to read region s + 1
      if (PROF) t1 = __builtin_readcyclecounter();
      __syncthreads();                // B(s)
      if (PROF) t2 = __builtin_readcyclecounter();
      issue_rows(s + 1 + P);
      issue_edges(s + P + EL, ptrL[min((s + P + EL) * kDmaG, cnt)]);
      if (PROF) {
        c_wait += t1 - t0;
        c_bar += t2 - t1;
        c_issue += __builtin_readcyclecounter() - t2;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the block's LDS allocation
    if (PROF && prof && lane == 0) {
      unsigned long long* pr = prof + (size_t)blockIdx.x * 8;
      pr[0] = c_wait; pr[1] = c_bar; pr[2] = c_issue;
      pr[5] = __builtin_readcyclecounter() - t_begin; pr[6] = (unsigned long long)nsteps;
    }
    if (!WEIGHT && tail.amax) __syncthreads();  // (the consumers' barrier in front of their last flush of the row maxima)
    if (TAIL) agg_tail_finish(tail, redL, false, 0, 0, f4_zero(), f4_zero(), dim);
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const int g = t / gs, c4 = t - g * gs;
  const bool active = g < kDmaG;
  float4 pa = f4_zero(), pb = f4_zero();
  if (PRE && !LACT && active) {
    pa = reinterpret_cast<const float4*>(pre_coef)[c4];
    pb = reinterpret_cast<const float4*>(pre_coef + dim)[c4];
  }
  auto act = [&](float4 v) {  // a row that did not pass through the ring (LACT: coefficients from LDS, this is the rare path)
    if (PRE) {
      if (LACT) {
        pa = reinterpret_cast<const float4*>(coefL)[c4];
        pb = reinterpret_cast<const float4*>(coefL + dim)[c4];
      }
      v = make_float4(fmaf(pa.x, v.x, pb.x), fmaf(pa.y, v.y, pb.y), fmaf(pa.z, v.z, pb.z), fmaf(pa.w, v.w, pb.w));
      if (pre_relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    return v;
  };
  auto act_ring = [&](float4 v) { return LACT ? v : act(v); };  // a row read from the ring
  if (LACT)
    for (int q = t; q < 2 * dim; q += cthreads) coefL[q] = pre_coef[q];
  if (TABLE) {
    for (int q = t; q < kNumCodes * dim; q += cthreads) {
      const int c = q / dim, d = q - c * dim;
      T[q] = emb1[(c / 3) * dim + d] + emb2[(c % 3) * dim + d];
    }
  }
  if (!WEIGHT && tail.amax && t < 2 * kDmaG) amaxL[t] = 0u;
  for (int q = t; q <= cnt; q += cthreads) ptrL[q] = ptr[n0 + q];
  if (WEIGHT)
    for (int q = t; q < (nsteps + 2) * kDmaG; q += cthreads) {  // the ring window of the last step ends at n0 + 8 nsteps + 8
      const int r = n0 - kDmaG + q;
      dinvL[q] = (r >= 0 && r < n) ? dinv[r] : 0.f;
    }
  // (TAIL) forward coefficients y = a z + b of the BatchNorm below, recomputed exactly as k_bn_bwd_partial recomputes them.  They live
  // in LDS (tailL [4][dim]: a, b, mean, invstd), not in 16 registers per thread: with them the instance needed 93 VGPRs = 5 waves
  // per SIMD = ONE 11-wave block per CU (rounds 4-5: 325.8 us); four ds_read_b128 per node row buy the second resident block.
  float4 ts1 = f4_zero(), ts2 = f4_zero();
  if (TAIL && g == 0 && c4 < gs) {
    const float4 gm = reinterpret_cast<const float4*>(tail.gamma)[c4], bt = reinterpret_cast<const float4*>(tail.beta)[c4];
    const float4 tmu = reinterpret_cast<const float4*>(tail.save_mean)[c4];
    const float4 tis = reinterpret_cast<const float4*>(tail.save_invstd)[c4];
    const float4 ta = make_float4(tis.x * gm.x, tis.y * gm.y, tis.z * gm.z, tis.w * gm.w);
    const float4 tb = make_float4(fmaf(-tmu.x, ta.x, bt.x), fmaf(-tmu.y, ta.y, bt.y), fmaf(-tmu.z, ta.z, bt.z), fmaf(-tmu.w, ta.w, bt.w));
    reinterpret_cast<float4*>(tailL)[c4] = ta;
    reinterpret_cast<float4*>(tailL + dim)[c4] = tb;
    reinterpret_cast<float4*>(tailL + 2 * dim)[c4] = tmu;
    reinterpret_cast<float4*>(tailL + 3 * dim)[c4] = tis;
    if (blockIdx.x == 0) {
      float* coef = tail.fold.coef;
      reinterpret_cast<float4*>(coef)[c4] = ta;
      reinterpret_cast<float4*>(coef + dim)[c4] = tb;
      reinterpret_cast<float4*>(coef + 2 * dim)[c4] = tmu;
      reinterpret_cast<float4*>(coef + 3 * dim)[c4] = tis;
    }
  }
  __syncthreads();  // prologue
  if (LACT) __syncthreads();  // (the loader has activated the rows of regions -1 and 0 with the coefficients published above)

  // per-step row pointers are read one step ahead (they sit in LDS for the whole block), so the chain
  // after a barrier is only: edge indices -> rows -> adds -> store
  int nb_e0 = ptrL[0], nb_beg = 0, nb_end = 0;
  if (active && g < cnt) { nb_beg = ptrL[g]; nb_end = ptrL[g + 1]; }
  unsigned long long c_cbar = 0, c_work = 0, t_prev = 0;
  if (PROF) t_prev = __builtin_readcyclecounter();
  float4 pf0 = f4_zero(), pf1 = f4_zero();  // (PF) rows fetched for this step's node during the previous step: its first npf far
  int npf = 0;                              // sources, in edge order (the summation below meets them in the same order)
  for (int s = 0; s < nsteps; ++s) {
    if (PROF) {
      const unsigned long long tb = __builtin_readcyclecounter();
      c_work += tb - t_prev;
      __syncthreads();  // B(s)
      t_prev = __builtin_readcyclecounter();
      c_cbar += t_prev - tb;
    } else {
      __syncthreads();  // B(s)
    }
    const int li = s * kDmaG + g;
    if (!WEIGHT && tail.amax && s > 0 && active && c4 == 0 && li - kDmaG < cnt) {  // the row this slot finished in step s - 1
      tail.amax[n0 + li - kDmaG] = amaxL[((s - 1) & 1) * kDmaG + g];
      amaxL[((s - 1) & 1) * kDmaG + g] = 0u;  // (next written in step s + 1, behind B(s + 1))
    }
    const int e0 = nb_e0, beg = nb_beg, end = nb_end;
    {
      const int ln = li + kDmaG;
      nb_e0 = ptrL[min((s + 1) * kDmaG, cnt)];
      if (active && ln < cnt) { nb_beg = ptrL[ln]; nb_end = ptrL[ln + 1]; }
    }
    if (!(active && li < cnt)) continue;
    const int i = n0 + li;
    const int base = n0 + s * kDmaG;
    const int win_lo = max(base - kDmaG, 0), win_hi = min(base + 2 * kDmaG, n);
    const int* idxB = idxL + (s % NBUF) * kDmaEdges;
    const int* codeB = codeL + (s % NBUF) * kDmaEdges;
    float4 zrow = f4_zero();
    if (TAIL) zrow = reinterpret_cast<const float4*>(tail.z + (int64_t)i * tail.ldz)[c4];  // (lands under the edge loop)
    // the node's own row (self loop) does not depend on the edge list: fetch it first
    float4 self = act_ring(ring[slot_of(i) * gs + c4]);
    if (TABLE) self = f4_add(self, T4[kSelfLoopCode * gs + c4]);
    float di = 1.f;
    if (WEIGHT) {
      di = dinvL[li + kDmaG];
      self = f4_scale(self, di * di);
    }
    float4 acc = f4_zero();
    // edges gathered per batch.  Measured on the roofline batch: 2 -> 225-234 us, 1 -> 240, 3/4 -> 245;
    // keeping the common bond-table rows in registers (select chain) was a loss (320-360 us).
    constexpr int CH = 2;
    int nf = 0;  // (PF) far sources met so far
    for (int p = beg; p < end; p += CH) {
      int sidx[CH], cd[CH], psel[CH];
      bool slow = false;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        sidx[j] = 0;
        cd[j] = 0;
        psel[j] = -1;
        if (p + j < end) {
          const int k = p + j - e0;
          if (k < kDmaEdges) {
            sidx[j] = idxB[k];
            if (TABLE) cd[j] = codeB[k] & 0xff;
            const bool far = (sidx[j] < win_lo) | (sidx[j] >= win_hi);
            if (PF) {
              if (far) {  // (rare: the whole wave skips this)
                slow |= nf >= npf;
                psel[j] = nf++;
                sidx[j] = i;  // the LDS read below must stay inside the ring; its value is replaced
              }
            } else {
              slow |= far;
            }
          } else {
            slow = true;
          }
        }
      }
      if (__any(slow)) {
        // rare: a source row outside the LDS window, or more edges in this step than staged slots
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (p + j < end) {
            const int sj = nbr[p + j];
            float4 m = act(x4[(int64_t)sj * ldx4 + c4]);
            if (TABLE) m = f4_add(m, T4[(int)code[p + j] * gs + c4]);
            if (WEIGHT) m = f4_scale(m, di * dinv[sj]);
            acc = f4_add(acc, m);
          }
        }
      } else {
        float4 v[CH], tv[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (p + j < end) {
            v[j] = ring[slot_of(sidx[j]) * gs + c4];
            if (PF) {
              if (psel[j] >= 0) v[j] = LACT ? act(psel[j] == 0 ? pf0 : pf1) : (psel[j] == 0 ? pf0 : pf1);
            }
            if (TABLE) tv[j] = T4[cd[j] * gs + c4];
          }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (p + j < end) {
            float4 m = act_ring(v[j]);
            if (TABLE) m = f4_add(m, tv[j]);
            if (WEIGHT) m = f4_scale(m, di * dinvL[sidx[j] - n0 + kDmaG]);
            acc = f4_add(acc, m);
          }
        }
      }
    }
    acc = f4_add(acc, self);
    if (PF) {
      // the node this thread sums in step s + 1: its first four edges' sources are already staged; fetch the (at most two)
      // rows that the ring will not hold then.  Issued in front of this step's store: the loads are older than it, so the wait in
      // front of their use does not include the store.
      npf = 0;
      const int fm = __builtin_amdgcn_readfirstlane(farL[((s + 1) % NBUF) * 2] | farL[((s + 1) % NBUF) * 2 + 1]);
      if (fm != 0 && li + kDmaG < cnt) {  // (uniform: nothing far among the next step's staged sources -> nothing to look for)
        const int* idxN = idxL + ((s + 1) % NBUF) * kDmaEdges;
        const int nlo = max(base, 0), nhi = min(base + 3 * kDmaG, n);
        const int pe = min(nb_end, nb_beg + 4);
        for (int q = nb_beg; q < pe; ++q) {
          const int k = q - nb_e0;
          if (k < kDmaEdges) {
            const int sj = idxN[k];
            if ((sj < nlo) | (sj >= nhi)) {
              if (npf == 0) pf0 = x4[(int64_t)sj * ldx4 + c4];
              else if (npf == 1) pf1 = x4[(int64_t)sj * ldx4 + c4];
              if (npf < 2) ++npf;
              else break;  // (a third far source: it and whatever follows take the branch -- the counts must stay in step)
            }
          }
        }
      }
    }
    if (POL & 2) {
      v4f_t o = {acc.x, acc.y, acc.z, acc.w};
      __builtin_nontemporal_store(o, reinterpret_cast<v4f_t*>(out) + ((int64_t)i * ldo4 + c4));
    } else {
      reinterpret_cast<float4*>(out)[(int64_t)i * ldo4 + c4] = acc;
    }
    if (!WEIGHT && tail.amax)
      atomicMax(amaxL + (s & 1) * kDmaG + g, __float_as_uint(fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)))));
    if (TAIL) {  // the same expressions as k_bn_bwd_partial's
      float4 gq = acc;
      const float4 tmu = reinterpret_cast<const float4*>(tailL + 2 * dim)[c4], tis = reinterpret_cast<const float4*>(tailL + 3 * dim)[c4];
      if (tail.relu) {
        const float4 ta = reinterpret_cast<const float4*>(tailL)[c4], tb = reinterpret_cast<const float4*>(tailL + dim)[c4];
        if (!(fmaf(ta.x, zrow.x, tb.x) > 0.f)) gq.x = 0.f;
        if (!(fmaf(ta.y, zrow.y, tb.y) > 0.f)) gq.y = 0.f;
        if (!(fmaf(ta.z, zrow.z, tb.z) > 0.f)) gq.z = 0.f;
        if (!(fmaf(ta.w, zrow.w, tb.w) > 0.f)) gq.w = 0.f;
      }
      ts1.x += gq.x; ts1.y += gq.y; ts1.z += gq.z; ts1.w += gq.w;
      ts2.x = fmaf(gq.x, (zrow.x - tmu.x) * tis.x, ts2.x);
      ts2.y = fmaf(gq.y, (zrow.y - tmu.y) * tis.y, ts2.y);
      ts2.z = fmaf(gq.z, (zrow.z - tmu.z) * tis.z, ts2.z);
      ts2.w = fmaf(gq.w, (zrow.w - tmu.w) * tis.w, ts2.w);
    }
  }
  if (!WEIGHT && tail.amax) {
    __syncthreads();  // the last step's maxima are complete (the loader wave meets this barrier too)
    const int li = (nsteps - 1) * kDmaG + g;
    if (nsteps > 0 && active && c4 == 0 && li < cnt) tail.amax[n0 + li] = amaxL[((nsteps - 1) & 1) * kDmaG + g];
  }
  if (TAIL) agg_tail_finish(tail, redL, active, g, c4, ts1, ts2, dim);
  if (PROF && prof && t == 0) {
    unsigned long long* pr = prof + (size_t)blockIdx.x * 8;
    pr[3] = c_cbar; pr[4] = c_work;
  }
}

unsigned long long* g_agg_prof = nullptr;  // set by pgnn_debug_aggregate_profile
int64_t g_agg_prof_blocks = 0;

template <bool TABLE, int P, int NROW, bool PRE = false, int POL = 0, bool WEIGHT = false, bool TAIL = false>
int launch_aggregate_dma_p(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const uint8_t* code,
                           const float* emb1, const float* emb2, float* out, int64_t ldo, int64_t n, int64_t dim,
                           hipStream_t st, const float* pre_coef = nullptr, int pre_relu = 0, const float* dinv = nullptr,
                           const AggTail* tail = nullptr, int max_blocks = 0) {
  static_assert(!(TAIL && WEIGHT), "the tail's LDS sits where the GCN normalisers would");
  const int gs = (int)(dim / 4);
  const int cthreads = (int)align_up((size_t)kDmaG * gs, kWave);
  const int threads = cthreads + kWave;
  const size_t lds = (size_t)(TABLE ? kNumCodes * dim : 0) * 4 + (size_t)(P + 3) * kDmaG * dim * 4 +
                     (size_t)(kDmaMaxNodes + 4) * 4 + (size_t)2 * (P + 1 + ((POL & 16) ? 1 : 0)) * kDmaEdges * 4 +
                     (WEIGHT ? (size_t)(kDmaMaxNodes + 3 * kDmaG) * 4 : 0) + 128 + (TAIL ? (size_t)(kDmaG * 2 + 4) * dim * 4 + 64 : 0) +
                     (PRE ? (size_t)2 * dim * 4 : 0);
  const int resident = (int)std::max<size_t>(1, (160 * 1024) / lds);
  const int64_t target_blocks = (int64_t)num_cu() * std::min(resident, env_int("PGNN_DMA_BPC", 2));
  int64_t npb = ceil_div(n, target_blocks);
  npb = std::min<int64_t>(std::max<int64_t>(npb, 4 * kDmaG), kDmaMaxNodes);
  npb = ceil_div(npb, kDmaG) * kDmaG;
  if (npb > kDmaMaxNodes) npb = kDmaMaxNodes;
  const int grid = (int)ceil_div(n, npb);
  if (TAIL && (grid > max_blocks || ceil_div(grid, kFoldGroup) > kFoldMaxGroups)) {
    set_error("aggregate_dma: %d blocks exceed the BatchNorm-backward scratch (%d)", grid, max_blocks);
    return PGNN_ERR_WORKSPACE;
  }
  unsigned long long* prof = ((POL & 8) && g_agg_prof_blocks >= grid) ? g_agg_prof : nullptr;
  allow_big_lds((const void*)k_aggregate_dma<TABLE, P, NROW, PRE, POL, WEIGHT, TAIL>, lds);
  hipLaunchKernelGGL((k_aggregate_dma<TABLE, P, NROW, PRE, POL, WEIGHT, TAIL>), dim3(grid), dim3(threads), lds, st, x, ldx, ptr, nbr,
                     code, emb1, emb2, out, ldo, (int)n, (int)dim, (int)npb, pre_coef, pre_relu, prof, dinv, tail ? *tail : AggTail{});
  return check_launch("aggregate_dma");
}

// BatchNorm(+ReLU)-on-read variant: D = 300 gets the tuned instantiation, every other width the generic one
int launch_aggregate_dma_pre(const float* z, int64_t ldz, const float* coef, int relu, const int32_t* ptr,
                             const int32_t* nbr, const uint8_t* code, const float* emb1, const float* emb2, float* out,
                             int64_t ldo, int64_t n, int64_t dim, hipStream_t st, const AggTail* tail = nullptr) {
  const int nrow = (int)ceil_div(kDmaG * (dim / 4), kWave);
  const bool small_ld = ldz * 4 * kDmaG < (1ll << 31);
  if (nrow == 10 && small_ld) {
    const bool nt = env_int("PGNN_DMA_POL", (int64_t)n * dim * 4 >= (128ll << 20) ? 3 : 0) == 3;
#ifdef PGNN_AB  // rounds 3-5: every gathered row activated by the consumers (86 VGPRs: one workgroup per CU) -- A/B builds only
    if (env_int("PGNN_DMA_ACT_CONSUMER", 0) != 0 && nt && !tail)
      return launch_aggregate_dma_p<true, 2, 10, true, 19 | 32>(z, ldz, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, coef, relu);
#endif
    if (env_int("PGNN_DMA_PF", 1) != 0) {
      if (nt) return launch_aggregate_dma_p<true, 2, 10, true, 19>(z, ldz, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, coef, relu);
      return launch_aggregate_dma_p<true, 2, 10, true, 16>(z, ldz, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, coef, relu, nullptr, tail);
    }
    if (nt) return launch_aggregate_dma_p<true, 2, 10, true, 3>(z, ldz, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, coef, relu);
    return launch_aggregate_dma_p<true, 2, 10, true>(z, ldz, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, coef, relu);
  }
  return launch_aggregate_dma_p<true, 2, 0, true>(z, ldz, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, coef, relu);
}

template <bool TABLE>
int launch_aggregate_dma(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const uint8_t* code,
                         const float* emb1, const float* emb2, float* out, int64_t ldo, int64_t n, int64_t dim,
                         hipStream_t st) {
  const int nrow = (int)ceil_div(kDmaG * (dim / 4), kWave);
  const bool small_ld = ldx * 4 * kDmaG < (1ll << 31);
#define PGNN_DMA_ARGS x, ldx, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st
#ifdef PGNN_AB  // ring depths 1 and 3 (measured behind 2): A/B builds only
  if (env_int("PGNN_DMA_P", 2) == 1) return launch_aggregate_dma_p<TABLE, 1, 0>(PGNN_DMA_ARGS);
#endif
  if (!small_ld || env_int("PGNN_DMA_GENERIC", 0)) return launch_aggregate_dma_p<TABLE, 2, 0>(PGNN_DMA_ARGS);
#ifdef PGNN_AB
  if (nrow == 10 && env_int("PGNN_DMA_P", 2) == 3) return launch_aggregate_dma_p<TABLE, 3, 10>(PGNN_DMA_ARGS);
#endif
  // Streaming hint.  When x and out together exceed the 256 MB Infinity Cache nothing this launch touches can be
  // re-used from cache by a later one: rows are then loaded and stored non-temporally (POL 3), which measured
  // 229.8 -> 214.3 us on the roofline batch (tools/agg_sweep.py; nt loads alone 224.9, nt stores alone 217.9, loader
  // priority 222.1).  Smaller batches keep the default policy so the next kernel finds the rows in L2 / MALL.
  const int pol = env_int("PGNN_DMA_POL", (int64_t)n * dim * 4 >= (128ll << 20) ? 3 : 0);
  const bool pf = env_int("PGNN_DMA_PF", 1) != 0;  // far rows fetched a step ahead (POL bit 4; the default -- 0 = read on the spot)
  if (nrow == 10 && pf) {
    if (pol == 3) return launch_aggregate_dma_p<TABLE, 2, 10, false, 19>(PGNN_DMA_ARGS);
    if (pol == 0) return launch_aggregate_dma_p<TABLE, 2, 10, false, 16>(PGNN_DMA_ARGS);
  }
  if (nrow == 10) {
    if (pol == 3) return launch_aggregate_dma_p<TABLE, 2, 10, false, 3>(PGNN_DMA_ARGS);
#ifdef PGNN_AB
    if (TABLE) {  // further A/B variants and the instrumented build (A/B builds only): production instantiation only
      switch (pol) {
        case 1: return launch_aggregate_dma_p<TABLE, 2, 10, false, 1>(PGNN_DMA_ARGS);
        case 2: return launch_aggregate_dma_p<TABLE, 2, 10, false, 2>(PGNN_DMA_ARGS);
        case 4: return launch_aggregate_dma_p<TABLE, 2, 10, false, 4>(PGNN_DMA_ARGS);
        case 7: return launch_aggregate_dma_p<TABLE, 2, 10, false, 7>(PGNN_DMA_ARGS);
        case 8: return launch_aggregate_dma_p<TABLE, 2, 10, false, 8>(PGNN_DMA_ARGS);
        default: break;
      }
    }
#endif
  }
  switch (nrow) {
    case 10: return launch_aggregate_dma_p<TABLE, 2, 10>(PGNN_DMA_ARGS);  // D = 300 (the reference's emb_dim)
    case 8: return launch_aggregate_dma_p<TABLE, 2, 8>(PGNN_DMA_ARGS);    // D = 256
    case 4: return launch_aggregate_dma_p<TABLE, 2, 4>(PGNN_DMA_ARGS);    // D = 128
    case 2: return launch_aggregate_dma_p<TABLE, 2, 2>(PGNN_DMA_ARGS);    // D = 64
    case 1: return launch_aggregate_dma_p<TABLE, 2, 1>(PGNN_DMA_ARGS);    // D = 32
    default: return launch_aggregate_dma_p<TABLE, 2, 0>(PGNN_DMA_ARGS);
  }
#undef PGNN_DMA_ARGS
}

// GCN-weighted aggregation on the loader/consumer kernel (D = 300 gets the tuned row-DMA instantiation)
template <bool TABLE>
int launch_aggregate_dma_weighted(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const uint8_t* code,
                                  const float* emb1, const float* emb2, const float* dinv, float* out, int64_t ldo,
                                  int64_t n, int64_t dim, hipStream_t st) {
  const int nrow = (int)ceil_div(kDmaG * (dim / 4), kWave);
  const bool small_ld = ldx * 4 * kDmaG < (1ll << 31);
  const bool nt = env_int("PGNN_DMA_POL", (int64_t)n * dim * 4 >= (128ll << 20) ? 3 : 0) == 3;
  if (nrow == 10 && small_ld) {
    if (nt) return launch_aggregate_dma_p<TABLE, 2, 10, false, 3, true>(x, ldx, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, nullptr, 0, dinv);
    return launch_aggregate_dma_p<TABLE, 2, 10, false, 0, true>(x, ldx, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, nullptr, 0, dinv);
  }
  return launch_aggregate_dma_p<TABLE, 2, 0, false, 0, true>(x, ldx, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st, nullptr, 0, dinv);
}

template <bool TABLE, bool WEIGHT>
int launch_aggregate(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr,
                     const uint8_t* code, const float* emb1, const float* emb2, const float* dinv,
                     float* out, int64_t ldo, int64_t n, int64_t dim, hipStream_t st) {
  const int variant = env_int("PGNN_AGG_VARIANT", dim <= 320 ? 3 : 1);
  if (variant == 3 && dim <= 320 && WEIGHT)
    return launch_aggregate_dma_weighted<TABLE>(x, ldx, ptr, nbr, code, emb1, emb2, dinv, out, ldo, n, dim, st);
  if (variant == 3 && dim <= 320 && !WEIGHT)
    return launch_aggregate_dma<TABLE>(x, ldx, ptr, nbr, code, emb1, emb2, out, ldo, n, dim, st);
  if (variant >= 1)
    return launch_aggregate_grp<TABLE, WEIGHT>(x, ldx, ptr, nbr, code, emb1, emb2, dinv, out, ldo, n, dim, st);
  const int R = (int)ceil_div(dim / 4, kWave);
  const size_t lds = TABLE ? (size_t)kNumCodes * dim * sizeof(float) : 0;
  int npc;
  const int grid = pick_grid(n, TABLE ? 6 : 8, &npc);
#define PGNN_LAUNCH_AGG(RR)                                                                         \
  allow_big_lds((const void*)k_aggregate<RR, TABLE, WEIGHT>, lds);                                  \
  hipLaunchKernelGGL((k_aggregate<RR, TABLE, WEIGHT>), dim3(grid), dim3(kBlock), lds, st, x, ldx,  \
                     ptr, nbr, code, emb1, emb2, dinv, out, ldo, (int)n, (int)dim, npc, num_xcd())
  switch (R) {
    case 1: PGNN_LAUNCH_AGG(1); break;
    case 2: PGNN_LAUNCH_AGG(2); break;
    case 3: PGNN_LAUNCH_AGG(3); break;
    case 4: PGNN_LAUNCH_AGG(4); break;
    default: set_error("feature width %lld > 1024 not supported", (long long)dim); return PGNN_ERR_ARG;
  }
#undef PGNN_LAUNCH_AGG
  return check_launch("aggregate");
}

// ---------------------------------------------------------------------------------------------
// out[i,:] (+)= cfeat[i,0:KC] . table[0:KC,:]
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(kBlock)
k_rowfeat_fwd(const float* __restrict__ cfeat, int kc, const float* __restrict__ table, int64_t ldt,
              float* __restrict__ out, int64_t ldo, int n, int dim, int accumulate, int nxcd) {
  extern __shared__ __align__(16) float T[];  // [kc][dim]
  for (int q = threadIdx.x; q < kc * dim; q += kBlock) T[q] = table[(q / dim) * ldt + (q % dim)];
  __syncthreads();
  const WaveSched ws(nxcd);
  const int lane = ws.lane, d4 = dim >> 2;
  for (int i = ws.wave; i < n; i += ws.nwaves) {
    const float myc = lane < kc ? cfeat[(int64_t)i * kc + lane] : 0.f;
    Row<R> acc;
    if (accumulate) row_load<R>(acc, out + (int64_t)i * ldo, lane, d4);
    else row_zero<R>(acc);
    for (int t = 0; t < kc; ++t) {
      const float c = __int_as_float(bcast_i32(__float_as_int(myc), t));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + r * kWave;
        if (r + 1 < R || ch < d4) {
          const float4 tv = reinterpret_cast<const float4*>(T + t * dim)[ch];
          acc.v[r].x = fmaf(c, tv.x, acc.v[r].x);
          acc.v[r].y = fmaf(c, tv.y, acc.v[r].y);
          acc.v[r].z = fmaf(c, tv.z, acc.v[r].z);
          acc.v[r].w = fmaf(c, tv.w, acc.v[r].w);
        }
      }
    }
    row_store<R>(acc, out + (int64_t)i * ldo, lane, d4);
  }
}

// gtable[t,:] = sum_i cfeat[i,t]*g[i,:] : per-wave register accumulators over a contiguous node
// range -> per-block partial (fixed wave order) -> second pass over blocks (fixed order).
template <int R, int KC>
__global__ void __launch_bounds__(kBlock)
k_rowfeat_bwd_partial(const float* __restrict__ cfeat, const float* __restrict__ g, int64_t ldg,
                      float* __restrict__ partial, int n, int dim) {
  extern __shared__ __align__(16) float red[];  // [waves][KC][dim]
  const int lane = lane_id(), w = threadIdx.x >> 6, d4 = dim >> 2;
  const int gw = blockIdx.x * kWavesPerBlock + w, nw = gridDim.x * kWavesPerBlock;
  const int per = (n + nw - 1) / nw;
  const int i0 = gw * per, i1 = min(n, i0 + per);
  Row<R> acc[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) row_zero<R>(acc[t]);
  // four rows' loads in flight per wave (clamped, unconditional: a guarded load waits for itself), consumed in row order: the
  // same fmaf chains as a one-row-at-a-time loop, which paid a full memory round trip per row (8 rows per wave at 6 747 rows:
  // 18.6 us of pure latency in the 256-graph step)
  constexpr int U = 4;
  for (int i = i0; i < i1; i += U) {
    float myc[U];
    Row<R> gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = min(i + u, i1 - 1);
      myc[u] = lane < KC ? cfeat[(int64_t)ii * KC + lane] : 0.f;
      row_load<R>(gv[u], g + (int64_t)ii * ldg, lane, d4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u >= i1) break;
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        const float c = __int_as_float(bcast_i32(__float_as_int(myc[u]), t));
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ch = lane + r * kWave;
          if (r + 1 < R || ch < d4) {
            acc[t].v[r].x = fmaf(c, gv[u].v[r].x, acc[t].v[r].x);
            acc[t].v[r].y = fmaf(c, gv[u].v[r].y, acc[t].v[r].y);
            acc[t].v[r].z = fmaf(c, gv[u].v[r].z, acc[t].v[r].z);
            acc[t].v[r].w = fmaf(c, gv[u].v[r].w, acc[t].v[r].w);
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KC; ++t) row_store<R>(acc[t], red + ((size_t)w * KC + t) * dim, lane, d4);
  __syncthreads();
  for (int q = threadIdx.x; q < KC * dim; q += kBlock) {
    float s = red[q];
    for (int ww = 1; ww < kWavesPerBlock; ++ww) s += red[(size_t)ww * KC * dim + q];
    partial[(size_t)blockIdx.x * KC * dim + q] = s;
  }
}

__global__ void __launch_bounds__(kBlock)
k_rowfeat_bwd_final(const float* __restrict__ partial, int nblocks, int kc, int dim,
                    float* __restrict__ gtable, int64_t s_row, int64_t s_col, float* __restrict__ last_row_out) {
  const int sl = threadIdx.x & 15;
  const int qq = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int q = min(qq, kc * dim - 1);
  const double s = slice_sum16(partial + q, (size_t)kc * dim, nblocks, sl);
  if (sl == 0 && qq < kc * dim) {
    const int r = q / dim, c = q - r * dim;
    if (last_row_out && r == kc - 1) last_row_out[c] = (float)s;
    else gtable[(int64_t)r * s_row + (int64_t)c * s_col] = (float)s;
  }
}

// rows per block of the weighted column sums.  Measured at N = 6 747 (tools/small_kernel_bench.py): 64 rows -> 12.0 us,
// 32 -> 10.9, 16 -> 13.4, 8 -> 20.5 (both launches together).
inline int rowfeat_bwd_blocks(int64_t n) {
  return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, env_int("PGNN_ROWFEAT_ROWS_PER_BLOCK", 32)), 1), 4 * num_cu());
}

// ---------------------------------------------------------------------------------------------
// input embedding: out[i] = t1[idx[i,0]] + t2[idx[i,1]]
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(kBlock)
k_embed_fwd(const int64_t* __restrict__ idx, int64_t stride, const float* __restrict__ t1, int rows1,
            const float* __restrict__ t2, int rows2, float* __restrict__ out, int64_t ldo, int n,
            int dim, int32_t* status, int nxcd) {
#pragma clang fp contract(off)
  const WaveSched ws(nxcd);
  const int lane = ws.lane, d4 = dim >> 2;
  for (int i = ws.wave; i < n; i += ws.nwaves) {
    int64_t a = idx[(int64_t)i * stride];
    int64_t b = t2 ? idx[(int64_t)i * stride + 1] : 0;
    if (a < 0 || a >= rows1 || (t2 && (b < 0 || b >= rows2))) {
      if (lane == 0) atomicAdd(status, 1);
      a = 0;
      b = 0;
    }
    Row<R> va, vb;
    row_load<R>(va, t1 + a * dim, lane, d4);
    if (t2) {
      row_load<R>(vb, t2 + b * dim, lane, d4);
#pragma unroll
      for (int r = 0; r < R; ++r) va.v[r] = f4_add(va.v[r], vb.v[r]);
    }
    row_store<R>(va, out + (int64_t)i * ldo, lane, d4);
  }
}

// ---------------------------------------------------------------------------------------------
// two-level deterministic segment sum over a grouped item list (see pgnn.h)
// ---------------------------------------------------------------------------------------------
constexpr int kSegChunk = 8;   // items per wave of the first pass: the pass is latency-bound (one wave walks its rows serially), so small chunks = more waves

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segsum_chunks(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                const int32_t* __restrict__ perm, int n_items, int n_seg, int mean,
                float* __restrict__ out, int64_t ldo, float* __restrict__ partial, int dim, int ptr_in_lds) {
  const int lane = lane_id(), d4 = dim >> 2;
  const int chunk = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int p0 = min(chunk * kSegChunk, max(n_items - 1, 0));  // a wave past the end still takes part in the block's table load
  const bool idle = chunk * kSegChunk >= n_items;
  const int cnt = min(kSegChunk, n_items - p0);
  const int item = lane < cnt ? (perm ? perm[p0 + lane] : p0 + lane) : 0;
  // segment of every position of the chunk: largest s with ptr[s] <= position (with empty segments ptr repeats and the
  // largest such s is the non-empty one).  Each lane searches for its own position, so a chunk that crosses dozens of
  // empty segments (the pair-type segments of the atom-embedding gradient: 360 segments, ~40 populated) pays one search,
  // not one dependent load per empty segment; a short table is searched in LDS.
  extern __shared__ int32_t s_ptr[];
  const int32_t* P = ptr;
  if (ptr_in_lds) {
    for (int q = threadIdx.x; q <= n_seg; q += kBlock) s_ptr[q] = ptr[q];
    __syncthreads();
    P = s_ptr;
  }
  const int pos = p0 + (lane < cnt ? lane : 0);
  int lo = 0, hi = n_seg;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (P[mid] <= pos) lo = mid; else hi = mid;
  }
  if (idle) return;
  const int seg_of = lo;
  int seg = bcast_i32(seg_of, 0);
  int seg_beg = P[seg], seg_end = P[seg + 1];
  Row<R> acc;
  row_zero<R>(acc);
  int j = 0;
  while (j < cnt) {
    const int run_end = min(cnt, seg_end - p0);  // positions [j, run_end) belong to `seg`
    while (j < run_end) {  // eight row loads in flight (a chunk is one wave: latency, not bandwidth, sets its pace), added in position order
      const int m = min(8, run_end - j);
      Row<R> v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < m) row_load<R>(v[u], x + (int64_t)bcast_i32(item, j + u) * ldx, lane, d4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < m) {
#pragma unroll
          for (int r = 0; r < R; ++r) acc.v[r] = f4_add(acc.v[r], v[u].v[r]);
        }
      j += m;
    }
    const bool whole = seg_beg >= p0 && seg_end <= p0 + kSegChunk;  // segment inside this chunk
    if (whole) {
      const float sc = mean ? 1.f / (float)max(seg_end - seg_beg, 1) : 1.f;
#pragma unroll
      for (int r = 0; r < R; ++r) acc.v[r] = f4_scale(acc.v[r], sc);
      row_store<R>(acc, out + (int64_t)seg * ldo, lane, d4);
    } else {
      const int slot = seg_beg < p0 ? 0 : 1;  // 0: continues from the previous chunk, 1: continues into the next
      row_store<R>(acc, partial + ((size_t)chunk * 2 + slot) * dim, lane, d4);
    }
    row_zero<R>(acc);
    if (j < cnt) {
      seg = bcast_i32(seg_of, j);
      seg_beg = P[seg];
      seg_end = P[seg + 1];
    }
  }
}

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segsum_final(const int32_t* __restrict__ ptr, int n_seg, int mean, float* __restrict__ out,
               int64_t ldo, const float* __restrict__ partial, int dim) {
  const int lane = lane_id(), d4 = dim >> 2;
  const int seg = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (seg >= n_seg) return;
  const int s = ptr[seg], e = ptr[seg + 1];
  Row<R> acc;
  row_zero<R>(acc);
  if (e > s) {
    const int c0 = s / kSegChunk, c1 = (e - 1) / kSegChunk;
    if (c0 == c1) return;  // written directly by k_segsum_chunks
    for (int c = c0; c <= c1; c += 4) {  // four partial rows in flight, added in chunk order
      const int m = min(4, c1 - c + 1);
      Row<R> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < m) {
          const int slot = ((c + u) * kSegChunk > s) ? 0 : 1;
          row_load<R>(v[u], partial + ((size_t)(c + u) * 2 + slot) * dim, lane, d4);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < m) {
#pragma unroll
          for (int r = 0; r < R; ++r) acc.v[r] = f4_add(acc.v[r], v[u].v[r]);
        }
    }
    if (mean) {
      const float sc = 1.f / (float)(e - s);
#pragma unroll
      for (int r = 0; r < R; ++r) acc.v[r] = f4_scale(acc.v[r], sc);
    }
  }
  row_store<R>(acc, out + (int64_t)seg * ldo, lane, d4);
}

// Long segments (a few table rows that collect hundreds of thousands of items, e.g. the carbon row of
// the atom-embedding gradient at a 16k-graph batch): one wave walking ~10^4 partial rows serially took
// 4 ms.  Split every segment's partial range over kSegSplit waves, then add the <= kSegSplit results.
constexpr int kSegSplitMax = 64;
inline int seg_split(int64_t nchunks) { return nchunks <= 512 ? 8 : (nchunks <= 8192 ? 32 : kSegSplitMax); }

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segsum_mid(const int32_t* __restrict__ ptr, int n_seg, const float* __restrict__ partial,
             float* __restrict__ partial2, int dim, int nsplit) {
  const int lane = lane_id(), d4 = dim >> 2;
  const int w = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int seg = w / nsplit, split = w % nsplit;
  if (seg >= n_seg) return;
  const int s = ptr[seg], e = ptr[seg + 1];
  Row<R> acc;
  row_zero<R>(acc);
  if (e > s) {
    const int c0 = s / kSegChunk, c1 = (e - 1) / kSegChunk;
    if (c0 == c1) return;  // written directly by k_segsum_chunks; final2 skips it too
    const int per = (c1 - c0 + 1 + nsplit - 1) / nsplit;
    const int a = c0 + split * per, b = min(c1 + 1, a + per);
    for (int c = a; c < b; c += 4) {
      const int m = min(4, b - c);
      Row<R> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < m) {
          const int slot = ((c + u) * kSegChunk > s) ? 0 : 1;
          row_load<R>(v[u], partial + ((size_t)(c + u) * 2 + slot) * dim, lane, d4);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < m) {
#pragma unroll
          for (int r = 0; r < R; ++r) acc.v[r] = f4_add(acc.v[r], v[u].v[r]);
        }
    }
  }
  row_store<R>(acc, partial2 + (size_t)w * dim, lane, d4);
}

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segsum_final2(const int32_t* __restrict__ ptr, int n_seg, int mean, float* __restrict__ out, int64_t ldo,
                const float* __restrict__ partial2, int dim, int nsplit) {
  const int lane = lane_id(), d4 = dim >> 2;
  const int seg = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (seg >= n_seg) return;
  const int s = ptr[seg], e = ptr[seg + 1];
  Row<R> acc;
  row_zero<R>(acc);
  if (e > s) {
    if (s / kSegChunk == (e - 1) / kSegChunk) return;
    for (int q = 0; q < nsplit; q += 4) {  // nsplit is a multiple of 4
      Row<R> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) row_load<R>(v[u], partial2 + ((size_t)seg * nsplit + q + u) * dim, lane, d4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc.v[r] = f4_add(acc.v[r], v[u].v[r]);
      }
    }
    if (mean) {
      const float sc = 1.f / (float)(e - s);
#pragma unroll
      for (int r = 0; r < R; ++r) acc.v[r] = f4_scale(acc.v[r], sc);
    }
  }
  row_store<R>(acc, out + (int64_t)seg * ldo, lane, d4);
}

inline bool segsum_two_level(int64_t n_items, int64_t n_segments) {
  return n_segments <= 1024 && ceil_div(std::max<int64_t>(n_items, 1), kSegChunk) > 64;
}

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segment_broadcast(const float* __restrict__ g, int64_t ldg, const int64_t* __restrict__ key,
                    const int32_t* __restrict__ ptr, int mean, float* __restrict__ gx, int64_t ldgx,
                    int n, int dim, int nxcd) {
  const WaveSched ws(nxcd);
  const int lane = ws.lane, d4 = dim >> 2;
  for (int i = ws.wave; i < n; i += ws.nwaves) {
    const int64_t s = key[i];
    Row<R> v;
    row_load<R>(v, g + s * ldg, lane, d4);
    if (mean) {
      const float sc = 1.f / (float)max(ptr[s + 1] - ptr[s], 1);
#pragma unroll
      for (int r = 0; r < R; ++r) v.v[r] = f4_scale(v.v[r], sc);
    }
    row_store<R>(v, gx + (int64_t)i * ldgx, lane, d4);
  }
}

inline int stream_grid(int64_t n_rows) {
  return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n_rows, kWavesPerBlock), 1), (int64_t)num_cu() * 8);
}

#define PGNN_DISPATCH_R(R_, CALL)                                                     \
  switch (R_) {                                                                       \
    case 1: { constexpr int RR = 1; CALL; } break;                                    \
    case 2: { constexpr int RR = 2; CALL; } break;                                    \
    case 3: { constexpr int RR = 3; CALL; } break;                                    \
    case 4: { constexpr int RR = 4; CALL; } break;                                    \
    default: set_error("feature width > 1024 not supported"); return PGNN_ERR_ARG;    \
  }

inline int check_dim(int64_t dim) {
  if (dim <= 0 || dim % 4 != 0 || dim > 1024) {
    set_error("feature width must be a multiple of 4 in (0,1024], got %lld", (long long)dim);
    return PGNN_ERR_ARG;
  }
  return PGNN_OK;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

int pgnn_chem_aggregate_fwd(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src,
                            const uint8_t* in_code, const float* emb1, const float* emb2,
                            const float* dinv, float* out, int64_t ldo, int64_t n, int64_t dim,
                            pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldx % 4 == 0 && ldo % 4 == 0, "bad aggregate arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dinv) return launch_aggregate<true, true>(x, ldx, in_ptr, in_src, in_code, emb1, emb2, dinv, out, ldo, n, dim, st);
  return launch_aggregate<true, false>(x, ldx, in_ptr, in_src, in_code, emb1, emb2, dinv, out, ldo, n, dim, st);
}

int pgnn_chem_aggregate_bn_fwd(const float* z, int64_t ldz, const float* coef, int relu, const int32_t* in_ptr,
                               const int32_t* in_src, const uint8_t* in_code, const float* emb1, const float* emb2,
                               float* out, int64_t ldo, int64_t n, int64_t dim, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldz % 4 == 0 && ldo % 4 == 0 && coef, "bad aggregate_bn arguments");
  PGNN_REQUIRE(dim <= 320, "aggregate_bn: feature width above 320 is not supported (materialise BatchNorm's output instead)");
  return launch_aggregate_dma_pre(z, ldz, coef, relu, in_ptr, in_src, in_code, emb1, emb2, out, ldo, n, dim,
                                  (hipStream_t)stream);
}

// out_a[a] = sum_b S[a*n_b + b], out_b[b] = sum_a S[a*n_b + b], float4 columns.  out_a: one thread per (a, column), b ascending.
// out_b sums over the LONG axis (120 atom types): 8 lanes per (b, column) take a = part, part + 8, ... and are folded by a
// fixed xor tree -- the same association every run, and 15 dependent adds per lane instead of 120.
__global__ void __launch_bounds__(256) k_pair_fold(const float* __restrict__ S, int n_a, int n_b, int d4,
                                                   float* __restrict__ out_a, int64_t lda, float* __restrict__ out_b,
                                                   int64_t ldb) {
  const int na4 = n_a * d4, nb0 = (na4 + 7) & ~7, total = nb0 + n_b * d4 * 8;  // the 8-lane groups start on a multiple of 8
  const float4* __restrict__ S4 = reinterpret_cast<const float4*>(S);
  for (int q0 = blockIdx.x * 256; q0 < total; q0 += gridDim.x * 256) {  // whole blocks iterate together (shuffles below)
    const int q = q0 + threadIdx.x;
    if (q < na4) {
      const int row = q / d4, c = q - row * d4;
      float4 acc = f4_zero();
      for (int b = 0; b < n_b; ++b) acc = f4_add(acc, S4[(size_t)(row * n_b + b) * d4 + c]);
      if (out_a) reinterpret_cast<float4*>(out_a + (int64_t)row * lda)[c] = acc;
    }
    const int t = q - nb0;
    const bool live = q >= nb0 && q < total;
    const int part = t & 7, o = live ? t >> 3 : 0;
    const int b = o / d4, c = o - b * d4;
    float4 acc = f4_zero();
    if (live)
      for (int a = part; a < n_a; a += 8) acc = f4_add(acc, S4[(size_t)(a * n_b + b) * d4 + c]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      acc.x += __shfl_xor(acc.x, m);
      acc.y += __shfl_xor(acc.y, m);
      acc.z += __shfl_xor(acc.z, m);
      acc.w += __shfl_xor(acc.w, m);
    }
    if (live && part == 0 && out_b) reinterpret_cast<float4*>(out_b + (int64_t)b * ldb)[c] = acc;
  }
}

int pgnn_pair_fold(const float* sums, int64_t n_a, int64_t n_b, float* out_a, int64_t lda, float* out_b, int64_t ldb,
                   int64_t dim, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n_a > 0 && n_b > 0 && lda % 4 == 0 && ldb % 4 == 0, "bad pair_fold arguments");
  const int d4 = (int)(dim / 4);
  const int grid = (int)std::min<int64_t>(ceil_div(align_up((size_t)(n_a * d4), 8) + n_b * d4 * 8, 256), 1024);
  hipLaunchKernelGGL(k_pair_fold, dim3(grid), dim3(256), 0, (hipStream_t)stream, sums, (int)n_a, (int)n_b, d4, out_a, lda,
                     out_b, ldb);
  return check_launch("pair_fold");
}

// diagnostics: plain float4 grid-stride copy -- the streaming ceiling the aggregation is measured against
__global__ void __launch_bounds__(256) k_debug_copy(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) b[q] = a[q];
}
__global__ void __launch_bounds__(256) k_debug_copy_nt(const v4f_t* __restrict__ a, v4f_t* __restrict__ b, int64_t n4) {
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + q), b + q);
}
int pgnn_debug_stream_copy(const float* src, float* dst, int64_t n_floats, int64_t blocks, pgnn_stream stream) {
  if (blocks < 0) {  // negative block count: the same copy with non-temporal loads and stores
    hipLaunchKernelGGL(k_debug_copy_nt, dim3((int)-blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const v4f_t*>(src), reinterpret_cast<v4f_t*>(dst), n_floats / 4);
    return check_launch("debug_stream_copy_nt");
  }
  hipLaunchKernelGGL(k_debug_copy, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n_floats / 4);
  return check_launch("debug_stream_copy");
}

int pgnn_debug_aggregate_profile(uint64_t* buffer, int64_t blocks) {
  g_agg_prof = reinterpret_cast<unsigned long long*>(buffer);
  g_agg_prof_blocks = buffer ? blocks : 0;
  return PGNN_OK;
}

int pgnn_neighbor_sum(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr,
                      const float* dinv, float* out, int64_t ldo, int64_t n, int64_t dim,
                      pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldx % 4 == 0 && ldo % 4 == 0, "bad neighbor_sum arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dinv) return launch_aggregate<false, true>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, dinv, out, ldo, n, dim, st);
  return launch_aggregate<false, false>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, dinv, out, ldo, n, dim, st);
}

}  // extern "C"

// pgnn_chem_aggregate_fwd / pgnn_chem_aggregate_bn_fwd (coef != NULL) that also leave the row maxima of `out` in amax [n] (bit
// patterns) for the two-plane product behind them -- on the tuned instantiation only (feature width 300, default policies); else the
// plain aggregation and *done = false (the product then takes the maxima itself).
int pgnn::chem_aggregate_fwd_amax(const float* x, int64_t ldx, const float* coef, int relu, const int32_t* in_ptr, const int32_t* in_src,
                                  const uint8_t* in_code, const float* emb1, const float* emb2, float* out, int64_t ldo, int64_t n,
                                  int64_t dim, uint32_t* amax, bool* done, hipStream_t st) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldx % 4 == 0 && ldo % 4 == 0, "bad aggregate arguments");
  const int nrow = (int)ceil_div(kDmaG * (dim / 4), kWave);
  const bool small_ld = ldx * 4 * kDmaG < (1ll << 31);
  *done = amax && dim <= 320 && nrow == 10 && small_ld && env_int("PGNN_AGG_VARIANT", 3) == 3 && env_int("PGNN_DMA_P", 2) == 2 &&
          !env_int("PGNN_DMA_GENERIC", 0) && env_int("PGNN_DMA_PF", 1) != 0 &&
          env_int("PGNN_DMA_POL", (int64_t)n * dim * 4 >= (128ll << 20) ? 3 : 0) == 0;
  if (!*done) {
    if (coef) return launch_aggregate_dma_pre(x, ldx, coef, relu, in_ptr, in_src, in_code, emb1, emb2, out, ldo, n, dim, st);
    return launch_aggregate<true, false>(x, ldx, in_ptr, in_src, in_code, emb1, emb2, nullptr, out, ldo, n, dim, st);
  }
  AggTail t{};
  t.amax = amax;
  if (coef) return launch_aggregate_dma_p<true, 2, 10, true, 16>(x, ldx, in_ptr, in_src, in_code, emb1, emb2, out, ldo, n, dim, st, coef, relu, nullptr, &t);
  return launch_aggregate_dma_p<true, 2, 10, false, 16>(x, ldx, in_ptr, in_src, in_code, emb1, emb2, out, ldo, n, dim, st, nullptr, 0, nullptr, &t);
}

// pgnn_neighbor_sum + the BatchNorm-backward column sums of the layer below in the same launch (bn_fold.h).  Fused only on the
// tuned instantiation (feature width 300, far rows prefetched, the default cache policy of batches below 128 MB); anything else
// is the plain sum and *fused = false: the caller then runs pgnn_bn_bwd as before.
int pgnn::neighbor_sum_bn_bwd(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, float* out, int64_t ldo, int64_t n,
                              int64_t dim, const BnBwdTail& tail, bool* fused, hipStream_t st) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldx % 4 == 0 && ldo % 4 == 0 && tail.ldz % 4 == 0, "bad neighbor_sum arguments");
  const int nrow = (int)ceil_div(kDmaG * (dim / 4), kWave);
  const bool small_ld = ldx * 4 * kDmaG < (1ll << 31);
  const bool tuned = dim <= 320 && nrow == 10 && small_ld && env_int("PGNN_AGG_VARIANT", 3) == 3 && env_int("PGNN_DMA_P", 2) == 2 &&
                     !env_int("PGNN_DMA_GENERIC", 0) && env_int("PGNN_DMA_PF", 1) != 0 &&
                     env_int("PGNN_BN_BWD_IN_AGG", 1) != 0;
  const int pol = (int)env_int("PGNN_DMA_POL", (int64_t)n * dim * 4 >= (128ll << 20) ? 3 : 0);  // large: non-temporal row loads and stores
  // (pol 3: the instance of the large batches, since the end of round 4 -- 438 792 rows: 31.67 against 32.09 ms per 16 384-graph step)
  if (pol != 0 && pol != 3) {
    *fused = false;
    return launch_aggregate<false, false>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, nullptr, out, ldo, n, dim, st);
  }
  *fused = tuned && tail.z && tail.scratch.partial && tail.scratch.tickets;
  if (!*fused) return launch_aggregate<false, false>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, nullptr, out, ldo, n, dim, st);
  AggTail t{};
  t.z = tail.z; t.ldz = tail.ldz; t.gamma = tail.gamma; t.beta = tail.beta; t.save_mean = tail.save_mean; t.save_invstd = tail.save_invstd;
  t.relu = tail.relu;
  t.fold = BnBwdFold{tail.gamma, tail.save_invstd, tail.scratch.partial, tail.scratch.gsum, tail.scratch.tickets, tail.scratch.coef,
                     tail.dgamma, tail.dbeta, tail.training, (int)n};
  const int rc = pol == 3 ? launch_aggregate_dma_p<false, 2, 10, false, 19, false, true>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, out, ldo, n, dim,
                                                                                         st, nullptr, 0, nullptr, &t, tail.scratch.max_blocks)
                          : launch_aggregate_dma_p<false, 2, 10, false, 16, false, true>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, out, ldo, n, dim,
                                                                                         st, nullptr, 0, nullptr, &t, tail.scratch.max_blocks);
  if (rc != PGNN_ERR_WORKSPACE) return rc;
  // more blocks than the BatchNorm-backward scratch has partial rows for (beyond 1 048 576 nodes, or PGNN_BN_ROWS_PER_BLOCK above 32;
  // nothing was launched): the plain sum, and the caller's pgnn_bn_bwd takes the sums in a pass of its own
  *fused = false;
  return launch_aggregate<false, false>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, nullptr, out, ldo, n, dim, st);
}

extern "C" {

int pgnn_rowfeat_matmul_fwd(const float* cfeat, int64_t kc, const float* table, int64_t ldt, float* out,
                            int64_t ldo, int64_t n, int64_t dim, int accumulate, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && kc > 0 && kc <= 16 && ldo % 4 == 0, "bad rowfeat_matmul_fwd arguments");
  const int R = (int)ceil_div(dim / 4, kWave);
  const size_t lds = (size_t)kc * dim * sizeof(float);
  const int grid = stream_grid(n);
  PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_rowfeat_fwd<RR>), dim3(grid), dim3(kBlock), lds,
                                        (hipStream_t)stream, cfeat, (int)kc, table, ldt, out, ldo,
                                        (int)n, (int)dim, accumulate, num_xcd()));
  return check_launch("rowfeat_matmul_fwd");
}

size_t pgnn_rowfeat_matmul_bwd_workspace_bytes(int64_t n, int64_t kc, int64_t dim) {
  return (size_t)rowfeat_bwd_blocks(n) * kc * dim * sizeof(float) + 256;
}

int pgnn_rowfeat_matmul_bwd(const float* cfeat, int64_t kc, const float* g, int64_t ldg, float* gtable,
                            int64_t ldgt, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                            pgnn_stream stream) {
  return pgnn::rowfeat_matmul_bwd_strided(cfeat, kc, g, ldg, gtable, ldgt, 1, nullptr, n, dim, ws, ws_bytes, (hipStream_t)stream);
}
}  // extern "C"

int pgnn::rowfeat_matmul_bwd_strided(const float* cfeat, int64_t kc, const float* g, int64_t ldg, float* gtable, int64_t s_row,
                                     int64_t s_col, float* last_row_out, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                                     hipStream_t stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && (kc == 2 || kc == 4 || kc == 7 || kc == 9 || kc == 10) && ldg % 4 == 0, "rowfeat_matmul_bwd supports kc in {2,4,7,9,10}");
  if (ws_bytes < pgnn_rowfeat_matmul_bwd_workspace_bytes(n, kc, dim)) {
    set_error("rowfeat_matmul_bwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int R = (int)ceil_div(dim / 4, kWave);
  const int nb = rowfeat_bwd_blocks(n);
  float* partial = static_cast<float*>(ws);
  const size_t lds = (size_t)kWavesPerBlock * kc * dim * sizeof(float);
  if (lds > 160 * 1024) {
    set_error("rowfeat_matmul_bwd: feature width %lld too large for the LDS reduction", (long long)dim);
    return PGNN_ERR_ARG;
  }
  if (kc == 2) {
    PGNN_DISPATCH_R(R, {
      allow_big_lds((const void*)k_rowfeat_bwd_partial<RR, 2>, lds);
      hipLaunchKernelGGL((k_rowfeat_bwd_partial<RR, 2>), dim3(nb), dim3(kBlock), lds, st, cfeat, g, ldg, partial, (int)n, (int)dim);
    });
  } else if (kc == 4) {
    PGNN_DISPATCH_R(R, {
      allow_big_lds((const void*)k_rowfeat_bwd_partial<RR, 4>, lds);
      hipLaunchKernelGGL((k_rowfeat_bwd_partial<RR, 4>), dim3(nb), dim3(kBlock), lds, st, cfeat, g, ldg, partial, (int)n, (int)dim);
    });
  } else if (kc == 7) {
    PGNN_DISPATCH_R(R, {
      allow_big_lds((const void*)k_rowfeat_bwd_partial<RR, 7>, lds);
      hipLaunchKernelGGL((k_rowfeat_bwd_partial<RR, 7>), dim3(nb), dim3(kBlock), lds, st, cfeat, g, ldg, partial, (int)n, (int)dim);
    });
  } else if (kc == 9) {
    PGNN_DISPATCH_R(R, {
      allow_big_lds((const void*)k_rowfeat_bwd_partial<RR, 9>, lds);
      hipLaunchKernelGGL((k_rowfeat_bwd_partial<RR, 9>), dim3(nb), dim3(kBlock), lds, st, cfeat, g, ldg, partial, (int)n, (int)dim);
    });
  } else {
    PGNN_DISPATCH_R(R, {
      allow_big_lds((const void*)k_rowfeat_bwd_partial<RR, 10>, lds);
      hipLaunchKernelGGL((k_rowfeat_bwd_partial<RR, 10>), dim3(nb), dim3(kBlock), lds, st, cfeat, g, ldg, partial, (int)n, (int)dim);
    });
  }
  hipLaunchKernelGGL(k_rowfeat_bwd_final, dim3((int)ceil_div(kc * dim, 16)), dim3(kBlock), 0, st,
                     partial, nb, (int)kc, (int)dim, gtable, s_row, s_col, last_row_out);
  return check_launch("rowfeat_matmul_bwd");
}

extern "C" {

int pgnn_embed_fwd(const int64_t* idx, int64_t idx_stride, const float* table1, int64_t rows1,
                   const float* table2, int64_t rows2, float* out, int64_t ldo, int64_t n, int64_t dim,
                   int32_t* status, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldo % 4 == 0 && idx_stride >= (table2 ? 2 : 1), "bad embed_fwd arguments");
  const int R = (int)ceil_div(dim / 4, kWave);
  const int grid = stream_grid(n);
  PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_embed_fwd<RR>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                                        idx, idx_stride, table1, (int)rows1, table2, (int)rows2, out, ldo,
                                        (int)n, (int)dim, status, num_xcd()));
  return check_launch("embed_fwd");
}

size_t pgnn_segment_sum_workspace_bytes(int64_t n_items, int64_t n_segments, int64_t dim) {
  size_t b = align_up((size_t)ceil_div(std::max<int64_t>(n_items, 1), kSegChunk) * 2 * dim * sizeof(float), 256);
  if (segsum_two_level(n_items, n_segments)) b += align_up((size_t)n_segments * kSegSplitMax * dim * sizeof(float), 256);
  return b + 256;
}

int pgnn_segment_sum(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* perm, int64_t n_items,
                     int64_t n_segments, int mean, float* out, int64_t ldo, int64_t dim, void* ws,
                     size_t ws_bytes, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n_items >= 0 && n_segments > 0 && ldx % 4 == 0 && ldo % 4 == 0, "bad segment_sum arguments");
  if (ws_bytes < pgnn_segment_sum_workspace_bytes(n_items, n_segments, dim)) {
    set_error("segment_sum workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int R = (int)ceil_div(dim / 4, kWave);
  float* partial = static_cast<float*>(ws);
  const int nchunks = (int)ceil_div(n_items, kSegChunk);
  if (nchunks > 0) {
    const int in_lds = n_segments + 1 <= 4096;
    PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_segsum_chunks<RR>), dim3((int)ceil_div(nchunks, kWavesPerBlock)),
                                          dim3(kBlock), in_lds ? (size_t)(n_segments + 1) * sizeof(int32_t) : 0, st, x, ldx, ptr,
                                          perm, (int)n_items, (int)n_segments, mean, out, ldo, partial, (int)dim, in_lds));
  }
  if (segsum_two_level(n_items, n_segments)) {
    float* partial2 = partial + align_up((size_t)nchunks * 2 * dim * sizeof(float), 256) / sizeof(float);
    const int ns = seg_split(nchunks);
    PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_segsum_mid<RR>), dim3((int)ceil_div(n_segments * ns, kWavesPerBlock)),
                                          dim3(kBlock), 0, st, ptr, (int)n_segments, partial, partial2, (int)dim, ns));
    PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_segsum_final2<RR>), dim3((int)ceil_div(n_segments, kWavesPerBlock)),
                                          dim3(kBlock), 0, st, ptr, (int)n_segments, mean, out, ldo, partial2,
                                          (int)dim, ns));
    return check_launch("segment_sum");
  }
  PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_segsum_final<RR>), dim3((int)ceil_div(n_segments, kWavesPerBlock)),
                                        dim3(kBlock), 0, st, ptr, (int)n_segments, mean, out, ldo, partial,
                                        (int)dim));
  return check_launch("segment_sum");
}

int pgnn_segment_broadcast(const float* g, int64_t ldg, const int64_t* key, const int32_t* ptr, int mean,
                           float* gx, int64_t ldgx, int64_t n_items, int64_t dim, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n_items > 0 && ldg % 4 == 0 && ldgx % 4 == 0, "bad segment_broadcast arguments");
  const int R = (int)ceil_div(dim / 4, kWave);
  const int grid = stream_grid(n_items);
  PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_segment_broadcast<RR>), dim3(grid), dim3(kBlock), 0,
                                        (hipStream_t)stream, g, ldg, key, ptr, mean, gx, ldgx, (int)n_items,
                                        (int)dim, num_xcd()));
  return check_launch("segment_broadcast");
}

}  // extern "C"
