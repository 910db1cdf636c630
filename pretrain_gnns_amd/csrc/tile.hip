// Graph-resident aggregation for batches of small DENSE graphs (bio PPI ego nets: ~40 nodes, ~19 in-edges per node that
// reach anywhere inside the graph -- bio/model.py:52-55,112-114 through propagate).
//
// The molecule kernel of aggregate.hip keeps a 24-row window of x in LDS because a molecule's bonds join atoms a few
// rows apart; an ego net's edges do not, so there almost every neighbour row missed the window and was gathered from L2
// (228 MB of L2 reads per 36 MB of compulsory traffic at 256 graphs, 0.127 of the HBM roofline at 4 096).  Here the unit
// of work is a CLOSED node interval -- no edge crosses its boundaries; for a block-diagonal batch these are the graphs --
// whose rows are DMA'd into LDS once, after which the ~19 row gathers per node are LDS reads.
//
//   pgnn_graph_tiles        once per batch: closed intervals from the two CSRs (a boundary c is open iff some node j < c
//                           has a neighbour >= c: every node marks the boundaries (j, max neighbour] it covers), compacted
//                           by a one-block scan into tile_start[T + 1]; T stays on the device
//   pgnn_neighbor_sum_tiled out[i] = sum_{e in seg(i)} w_e x[nbr_e] + w_ii x[i]  per interval, in chunks of kRows rows
//                           (an interval longer than a chunk still works: sources outside the chunk are read from
//                           memory); sums sequential in CSR order, self loop last = bit-identical to pgnn_neighbor_sum.
// Algorithmic bytes per launch: N*D*4 (x) + N*D*4 (out) + 4 E + 4 N; HBM-bound.
#include <atomic>
#include <type_traits>

#include "common.h"

using namespace pgnn;

namespace {

constexpr int kRows = 112;       // rows of x resident per chunk: 134.4 KB at D = 300, one block per CU.  Smaller chunks with two
                                 // blocks per CU were measured (48, 60 rows): the ego nets that do not fit send a batch of edges per
                                 // wave to memory and ONE such graph sets the time of a 256-graph launch (45 us vs ~9 us for a resident one)
constexpr int kIdxCap = 2048;    // staged neighbour indices per chunk (8 KB); the rest is read from memory
constexpr int kThreads = 1024;   // 13 groups of 75 threads at D = 300
constexpr int kMaxFeat = 10;     // per-node edge-feature sums folded into the same pass (bio: 9 attribute columns + the count)
constexpr int kBatch = 4;        // edges gathered per round (8: 205 vs 190 us on the 4 096-graph batch)

#define PGNN_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PGNN_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef float v4f_tile __attribute__((ext_vector_type(4)));

// covered[c] = 1 iff some edge joins a node < c with a node >= c  (c = 1 .. N-1)
__global__ void __launch_bounds__(256) k_tile_cover(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src,
                                                    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                    int n, uint8_t* __restrict__ covered) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  int hi = j;
  for (int p = in_ptr[j]; p < in_ptr[j + 1]; ++p) hi = max(hi, in_src[p]);
  for (int p = out_ptr[j]; p < out_ptr[j + 1]; ++p) hi = max(hi, out_dst[p]);
  hi = min(hi, n - 1);
  for (int c = j + 1; c <= hi; ++c) covered[c] = 1;  // every writer stores the same value
}

// tile_start = [0] + {c : !covered[c]} + [N]; one block, each thread owns a contiguous slice (count, scan, write)
__global__ void __launch_bounds__(1024) k_tile_compact(const uint8_t* __restrict__ covered, int n, int32_t* __restrict__ tile_start,
                                                       int32_t* __restrict__ num_tiles) {
  __shared__ int cnt[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = max(1, t * per), hi = min(n, (t + 1) * per);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += covered[i] == 0;
  cnt[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? cnt[t - d] : 0;
    __syncthreads();
    cnt[t] += v;
    __syncthreads();
  }
  int w = 1 + cnt[t] - c;  // tile_start[0] = 0
  for (int i = lo; i < hi; ++i)
    if (covered[i] == 0) tile_start[w++] = i;
  if (t == 1023) {
    const int total = cnt[1023] + 1;  // tiles
    tile_start[total] = n;
    *num_tiles = total;
  }
  if (t == 0) tile_start[0] = 0;
}

// MASK: out[i, c] is zeroed where mask[i, c] <= 0 -- the ReLU between two layers in the backward (out = the gradient of the previous
// layer's output, mask = that output): the chunk's mask rows are folded to one bit per value while the rows are staged (the words
// take the LDS of the edge-feature sums, which a masked call does not have), and the store picks its four bits.  As a launch of its
// own the mask was a read-modify-write pass over the gradient, 10-12 us per layer of a 256-ego-net batch.
template <bool WEIGHT, int DBG = 0, bool MASK = false>  // DBG (PGNN_TILE_DEBUG, timing only): 1 = skip the gather loop, 2 = skip the row DMA
__global__ void __launch_bounds__(kThreads) k_neighbor_sum_tile(const float* __restrict__ x, int64_t ldx,
                                                                const int32_t* __restrict__ ptr, const int32_t* __restrict__ nbr,
                                                                const float* __restrict__ dinv,
                                                                const int32_t* __restrict__ tile_start,
                                                                const int32_t* __restrict__ num_tiles, float* __restrict__ out,
                                                                int64_t ldo, int n, int dim, const float* __restrict__ cfeat,
                                                                int kc, const float* __restrict__ table, int64_t ldt,
                                                                float* __restrict__ fout, int64_t ldf,
                                                                const float* __restrict__ mask = nullptr, int64_t ldm = 0) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float smem[];
  const int gs = dim >> 2;
  float4* rows = reinterpret_cast<float4*>(smem);                       // [kRows][gs]
  int* idxL = reinterpret_cast<int*>(rows + kRows * gs);                // [kIdxCap]
  int* ptrL = idxL + kIdxCap;                                           // [kRows + 1]
  float4* tabL = reinterpret_cast<float4*>(ptrL + kRows + 4);           // [kc][gs]: the edge-feature table (cfeat != NULL)
  float* cfL = reinterpret_cast<float*>(tabL + kMaxFeat * gs);          // [kRows][kc]: the chunk's per-node edge-feature sums
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  const int64_t ldx4 = ldx >> 2, ldo4 = ldo >> 2;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = kThreads / 64;
  const int groups = kThreads / gs;
  const int g = t / gs, c4 = t - g * gs;
  // the edge-feature table column of this thread in registers (its c4 never changes): read from LDS per node it was a third of the
  // fused launch's LDS traffic (10 float4 per thread and node against ~19 row gathers)
  float4 tv[kMaxFeat];
#pragma unroll
  for (int tt = 0; tt < kMaxFeat; ++tt)
    tv[tt] = (cfeat && tt < kc && g < groups) ? reinterpret_cast<const float4*>(table + (int64_t)tt * ldt)[c4] : f4_zero();
  (void)tabL;
  const int T = *num_tiles;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int a = tile_start[tile], b = tile_start[tile + 1];
    for (int c0 = a; c0 < b; c0 += kRows) {
      const int c1 = min(b, c0 + kRows), cnt = c1 - c0;
      __syncthreads();  // previous chunk fully consumed
      // rows [c0, c1) -> LDS, 1 KiB per wave instruction
      const int total4 = cnt * gs;
      for (int base = wave * 64; base < total4 && DBG != 2; base += nwaves * 64) {
        const int q = base + lane;
        if (q < total4) {
          const int r = q / gs, cc = q - r * gs;
          __builtin_amdgcn_global_load_lds(PGNN_GPTR(x4 + (int64_t)(c0 + r) * ldx4 + cc), PGNN_LPTR(rows + base), 16, 0, 0);
        }
      }
      const int e0 = ptr[c0];
      for (int q = t; q <= cnt; q += kThreads) ptrL[q] = ptr[c0 + q] - e0;
      const int ne = min(ptr[c1] - e0, kIdxCap);
      for (int q = t; q < ne; q += kThreads) idxL[q] = nbr[e0 + q];
      if (cfeat)  // contiguous: rows c0 .. c1 of cfeat [N, kc]
        for (int q = t; q < cnt * kc; q += kThreads) cfL[q] = cfeat[(int64_t)c0 * kc + q];
      const int wpr = (gs + 7) >> 3;  // mask words per row: 8 float4 = 32 values per word
      if (MASK) {
        unsigned* mL = reinterpret_cast<unsigned*>(cfL);  // [kRows][wpr], wpr <= kMaxFeat
        const float4* __restrict__ m4 = reinterpret_cast<const float4*>(mask);
        for (int q = t; q < cnt * wpr; q += kThreads) {
          const int r = q / wpr, wd = q - r * wpr;
          float4 yv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) yv[u] = wd * 8 + u < gs ? m4[(int64_t)(c0 + r) * (ldm >> 2) + wd * 8 + u] : f4_zero();
          unsigned bits = 0;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            bits |= ((yv[u].x > 0.f ? 1u : 0u) | (yv[u].y > 0.f ? 2u : 0u) | (yv[u].z > 0.f ? 4u : 0u) | (yv[u].w > 0.f ? 8u : 0u)) << (4 * u);
          mL[q] = bits;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (g < groups) {
        for (int li = g; li < cnt; li += groups) {
          const int i = c0 + li;
          const int beg = ptrL[li], end = ptrL[li + 1];
          float di = 1.f;
          if (WEIGHT) di = dinv[i];
          float4 acc = f4_zero();
          // batches of kBatch edges: indices first, then the rows, then the adds in edge order (one edge at a time is
          // ~250 cycles of LDS latency per edge with nothing to hide it).  A batch with a source outside the resident chunk,
          // or beyond the staged indices, takes a WAVE-UNIFORM branch that reads everything from memory: a per-lane choice
          // between an LDS and a global address compiles to flat loads, and per-lane global fall-backs make the compiler
          // wait for vmcnt(0) -- i.e. for the previous node's output store -- in front of every batch (measured: 44 us
          // per 256-graph launch either way, against 9 us for the load / store phases alone)
          // One batch of kBatch edges starting at position p, the first `nv` of them real.  Indices: unconditional LDS reads
          // at clamped positions (a position past the node's last edge re-reads that edge, one past the staged indices the
          // last staged one), so the reads issue back to back instead of read / wait / branch per edge; d = source - c0 as
          // unsigned doubles as the residency test (d < cnt) and the row offset; full batches carry no per-edge validity
          // work at all.  16 waves per CU make this loop VALU-issue-bound: ~15 -> ~8 vector instructions per edge took the
          // 4 096-graph launch from 299 to 190 us (fused with the edge-feature product: 378 -> 242 us).
          auto batch = [&](int p, int nv, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            unsigned d[kBatch];
            bool slow = p + kBatch > kIdxCap;
#pragma unroll
            for (int j = 0; j < kBatch; ++j)
              d[j] = (unsigned)(idxL[min(FULL ? p + j : min(p + j, p + nv - 1), kIdxCap - 1)] - c0);
#pragma unroll
            for (int j = 0; j < kBatch; ++j) slow |= (FULL || j < nv) & (d[j] >= (unsigned)cnt);
            if (__any(slow)) {
              // rare: resident rows still come from LDS (index clamped), only the lanes with a far source go to memory
#pragma unroll
              for (int j = 0; j < kBatch; ++j) {
                if (FULL || j < nv) {
                  const int sj = p + j < kIdxCap ? (int)d[j] + c0 : nbr[e0 + p + j];
                  const bool far = sj < c0 || sj >= c1;
                  float4 v = rows[(far ? li : sj - c0) * gs + c4];
                  if (far) v = x4[(int64_t)sj * ldx4 + c4];
                  if (WEIGHT) v = f4_scale(v, di * dinv[sj]);
                  acc = f4_add(acc, v);
                }
              }
            } else {
              float4 v[kBatch];
#pragma unroll
              for (int j = 0; j < kBatch; ++j) v[j] = rows[d[j] * gs + c4];  // (a padding j re-reads a resident row)
#pragma unroll
              for (int j = 0; j < kBatch; ++j) {
                if (FULL || j < nv) {  // (partial batches: nv is uniform across the lanes of a node's group)
                  if (WEIGHT) v[j] = f4_scale(v[j], di * dinv[(int)d[j] + c0]);
                  acc = f4_add(acc, v[j]);
                }
              }
            }
          };
          if (DBG != 1) {
            int p = beg;
            for (; p + kBatch <= end; p += kBatch) batch(p, kBatch, std::true_type{});
            if (p < end) batch(p, end - p, std::false_type{});
          }
          float4 self = rows[li * gs + c4];
          if (WEIGHT) self = f4_scale(self, di * di);
          acc = f4_add(acc, self);
          if (cfeat) {
            // + cfeat[i, :] . table: the edge-feature half of the bio message, the same fmaf chain as pgnn_rowfeat_matmul_fwd
            // (continuing from the neighbour sum when fout == out: the GCN form; from zero into its own columns: GIN)
            const bool same = fout == out;
            float4 f = same ? acc : f4_zero();
#pragma unroll
            for (int tt = 0; tt < kMaxFeat; ++tt) {
              if (tt < kc) {
                const float c = cfL[li * kc + tt];
                f.x = fmaf(c, tv[tt].x, f.x); f.y = fmaf(c, tv[tt].y, f.y); f.z = fmaf(c, tv[tt].z, f.z); f.w = fmaf(c, tv[tt].w, f.w);
              }
            }
            if (same) acc = f;
            else reinterpret_cast<float4*>(fout)[(int64_t)i * (ldf >> 2) + c4] = f;
          }
          if (MASK) {
            const unsigned bits = reinterpret_cast<const unsigned*>(cfL)[li * wpr + (c4 >> 3)] >> ((c4 & 7) * 4);
            if (!(bits & 1u)) acc.x = 0.f;
            if (!(bits & 2u)) acc.y = 0.f;
            if (!(bits & 4u)) acc.z = 0.f;
            if (!(bits & 8u)) acc.w = 0.f;
          }
          reinterpret_cast<float4*>(out)[(int64_t)i * ldo4 + c4] = acc;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same aggregation with the loads of tile j + 1 under the gathers of tile j (persistent blocks, one per CU).
// k_neighbor_sum_tile above loads, waits, gathers, stores: HBM idles during every gather and the LDS during every load (0.35 of
// the HBM roofline on the 4 096-graph batch at 1.007x the compulsory traffic).  Here, as in k_aggregate_dma (aggregate.hip):
//   * the LAST wave is a LOADER: it streams the next tile's rows, CSR slice, neighbour ids and edge-feature sums into LDS by DMA
//     (global_load_lds: no VGPR round trip), waits for them with vmcnt(0) -- it issues nothing else -- and publishes a small tile
//     descriptor; the other 15 waves are CONSUMERS (12 groups of 75 threads at D = 300) that gather out of LDS and store rows
//     they never wait for.  The split is by wave, not by phase, because hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS
//     read that follows an LDS-DMA on the same path -- on a path that only consumes there is none;
//   * rows live in one buffer of kCap rows that consecutive tiles fill from opposite ends: tile j + 1 is loaded under tile j's
//     gather whenever the two fit together, else after it (two large ego nets in a row: no overlap for that pair); neighbour
//     ids (kIdxSlot per tile, the rest is read from memory), CSR pointers, feature sums and descriptors have two slots each;
//   * a tile longer than kCap rows is walked in chunks of kCap rows (sources outside the chunk come from memory, as above).
// Barrier protocol per item k (a tile or a chunk): loader [desc k written, DMAs of k landed] A_k [stage k + 1, or, if it has to
// reuse item k's space, B_k first]; consumers A_k [gather k] (B_k).  Sums in CSR order, self loop last: bit-identical to
// k_neighbor_sum_tile and to pgnn_neighbor_sum.
constexpr int kCap = 104;        // 124.8 KB of rows at D = 300
constexpr int kIdxSlot = 1280;   // a 40-node ego net has ~760 in-edges
constexpr int kIdxPitch = kIdxSlot + 8;   // + padding a full batch may read past the staged ids
constexpr int kPtrSlot = kCap + 8;
constexpr int kConsumerThreads = kThreads - 64;

__device__ __forceinline__ void tile_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Tiles are handed out dynamically (ego nets differ 10x in size: round-robin left the busiest CU of the 4 096-graph batch with 860
// rows against a mean of 636): [slot][0] = next tile, [slot][1] = blocks that have left; the last one to leave zeroes both.  A launch
// takes the next of kTicketSlots slots (host counter), so launches in flight on different streams never share one.
constexpr int kTicketSlots = 1024;
__device__ unsigned g_tile_tickets[kTicketSlots][2];

template <bool WEIGHT, bool NT>
__global__ void __launch_bounds__(kThreads) k_neighbor_sum_tile_pipe(const float* __restrict__ x, int64_t ldx,
                                                                     const int32_t* __restrict__ ptr, const int32_t* __restrict__ nbr,
                                                                     const float* __restrict__ dinv,
                                                                     const int32_t* __restrict__ tile_start,
                                                                     const int32_t* __restrict__ num_tiles, float* __restrict__ out,
                                                                     int64_t ldo, int n, int dim, const float* __restrict__ cfeat,
                                                                     int kc, const float* __restrict__ table, int64_t ldt,
                                                                     float* __restrict__ fout, int64_t ldf, int ticket_slot) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float smem[];
  constexpr int AUX = NT ? 2 : 0;
  const int gs = dim >> 2;
  float4* rows = reinterpret_cast<float4*>(smem);                       // [kCap][gs]
  int* idxS = reinterpret_cast<int*>(rows + kCap * gs);                 // [2][kIdxPitch]
  int* ptrS = idxS + 2 * kIdxPitch;                                     // [2][kPtrSlot]
  float* cfS = reinterpret_cast<float*>(ptrS + 2 * kPtrSlot);           // [2][kCap * kMaxFeat]
  float4* tabL = reinterpret_cast<float4*>(cfS + 2 * kCap * kMaxFeat);  // [kc][gs]
  int* desc = reinterpret_cast<int*>(tabL + kMaxFeat * gs);  // [2][8]: c0, c1, e0, ring row, valid, late-next (plain LDS accesses: the
                                                            // barriers below are asm with a memory clobber; a volatile pointer compiles to flat loads + vmcnt(0))
  if (cfeat) {
    for (int q = threadIdx.x; q < kc * gs; q += kThreads)
      tabL[q] = reinterpret_cast<const float4*>(table + (int64_t)(q / gs) * ldt)[q % gs];
  }
  __syncthreads();
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  const int64_t ldx4 = ldx >> 2, ldo4 = ldo >> 2;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int T = *num_tiles, G = gridDim.x;

  if (wave == kThreads / 64 - 1) {
    // ------------------------------------------------------------------------------------------------ loader wave
    // everything of rows [c0, c1) -> LDS by DMA (one wave: 1 KiB of rows per instruction, lane offsets advanced without a division)
    __builtin_amdgcn_s_setprio(3);  // one wave feeds fifteen: it must not queue behind them for issue slots
    auto stage = [&](int c0, int c1, int e0, int e1, int off, int slot) {
      const int cnt = c1 - c0, total4 = cnt * gs;
      float4* dst = rows + off * gs;
      if (ldx4 == gs) {  // rows adjacent in memory: the tile is one flat copy
        const float4* src = x4 + (int64_t)c0 * gs + lane;
        for (int base = 0; base < total4; base += 64)
          if (base + lane < total4) __builtin_amdgcn_global_load_lds(PGNN_GPTR(src + base), PGNN_LPTR(dst + base), 16, 0, AUX);
      } else {
        int r = lane / gs, cc = lane - r * gs;
        const float4* src = x4 + (int64_t)c0 * ldx4;
        for (int base = 0; base < total4; base += 64) {
          if (base + lane < total4)
            __builtin_amdgcn_global_load_lds(PGNN_GPTR(src + (int64_t)r * ldx4 + cc), PGNN_LPTR(dst + base), 16, 0, AUX);
          cc += 64;
          if (gs >= 64) {  // (uniform) at most one row boundary per 64 float4
            const bool wrap = cc >= gs;
            cc -= wrap ? gs : 0;
            r += wrap ? 1 : 0;
          } else {
            while (cc >= gs) {
              cc -= gs;
              ++r;
            }
          }
        }
      }
      int* pd = ptrS + slot * kPtrSlot;
      for (int base = 0; base <= cnt; base += 64)
        if (base + lane <= cnt) __builtin_amdgcn_global_load_lds(PGNN_GPTR(ptr + c0 + base + lane), PGNN_LPTR(pd + base), 4, 0, 0);
      const int ne = min(e1 - e0, kIdxSlot);
      int* id = idxS + slot * kIdxPitch;
      for (int base = 0; base < ne; base += 64)
        if (base + lane < ne) __builtin_amdgcn_global_load_lds(PGNN_GPTR(nbr + e0 + base + lane), PGNN_LPTR(id + base), 4, 0, 0);
      if (cfeat) {
        float* cd = cfS + slot * kCap * kMaxFeat;
        const int nf = cnt * kc;
        for (int base = 0; base < nf; base += 64)
          if (base + lane < nf)
            __builtin_amdgcn_global_load_lds(PGNN_GPTR(cfeat + (int64_t)c0 * kc + base + lane), PGNN_LPTR(cd + base), 4, 0, 0);
      }
    };
    // item sequence of this block: the tiles its tickets buy, each in chunks of <= kCap rows
    unsigned* tickets = g_tile_tickets[ticket_slot];
    auto next_tile = [&]() {
      unsigned v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
      return tk < (unsigned)T ? (int)tk : T;  // (T = none left)
    };
    int tl = next_tile(), tb = 0;  // current tile and its end
    int c0 = 0, c1 = 0, e0 = 0, e1 = 0;
    bool valid = tl < T;
    if (valid) {
      c0 = tile_start[tl];
      tb = tile_start[tl + 1];
      c1 = min(tb, c0 + kCap);
      e0 = ptr[c0];
      e1 = ptr[c1];
      stage(c0, c1, e0, e1, 0, 0);
    }
    int off = 0, slot = 0;
    while (true) {
      // the item after this one (its scalar loads overlap the DMAs in flight)
      int n0 = 0, n1 = 0, ne0 = 0, ne1 = 0, ntl = tl, ntb = tb;
      bool nvalid = false;
      if (valid) {
        if (c1 < tb) {
          n0 = c1;
          nvalid = true;
        } else {
          ntl = next_tile();
          if (ntl < T) {
            n0 = tile_start[ntl];
            ntb = tile_start[ntl + 1];
            nvalid = true;
          }
        }
        if (nvalid) {
          n1 = min(ntb, n0 + kCap);
          ne0 = ptr[n0];
          ne1 = ptr[n1];
        }
      }
      // items alternate between the two ends of the row ring (even: from row 0 up, odd: from row kCap down), so two consecutive
      // items share it whenever their rows add up to at most kCap -- behind-or-in-front placement lost another 13 % of the pairs
      // to fragmentation
      int noff = -1;
      if (nvalid) {
        const int cnt0 = c1 - c0, cnt1 = n1 - n0;
        if (cnt0 + cnt1 <= kCap) noff = (slot ^ 1) ? kCap - cnt1 : 0;
      }
      const bool late = nvalid && noff < 0;
      if (lane == 0) {
        int* d = desc + slot * 8;
        d[0] = c0; d[1] = c1; d[2] = e0; d[3] = off; d[4] = valid ? 1 : 0; d[5] = late ? 1 : 0;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      tile_barrier();  // A_k: item k is in LDS, its descriptor too; the consumers are done with item k - 1
      if (!valid) break;
      if (nvalid) {
        if (late) {
          tile_barrier();  // B_k: the consumers are done with item k, whose space item k + 1 needs
          noff = (slot ^ 1) ? kCap - (n1 - n0) : 0;
        }
        stage(n0, n1, ne0, ne1, noff, slot ^ 1);
      }
      tl = ntl; tb = ntb; c0 = n0; c1 = n1; e0 = ne0; e1 = ne1; valid = nvalid;
      off = noff < 0 ? 0 : noff;
      slot ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the block's LDS allocation
    if (lane == 0 && __hip_atomic_fetch_add(tickets + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u) {
      publish(tickets, 0u);  // every block has taken its last ticket: the slot is clean for the launch that draws it next
      publish(tickets + 1, 0u);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumer waves
  const int groups = kConsumerThreads / gs;
  const int g = t / gs, c4 = t - g * gs;
  // the edge-feature table column of this thread in registers (its c4 never changes): read from LDS per node it was a third of the
  // fused launch's LDS traffic (10 float4 per thread and node against ~19 row gathers)
  float4 tv[kMaxFeat];
#pragma unroll
  for (int tt = 0; tt < kMaxFeat; ++tt) tv[tt] = (cfeat && tt < kc && g < groups) ? tabL[tt * gs + c4] : f4_zero();
  int slot = 0;
  while (true) {
    tile_barrier();  // A_k
    const int* d = desc + slot * 8;
    const int c0 = __builtin_amdgcn_readfirstlane(d[0]), c1 = __builtin_amdgcn_readfirstlane(d[1]);  // (uniform: scalar registers)
    const int e0 = __builtin_amdgcn_readfirstlane(d[2]), off = __builtin_amdgcn_readfirstlane(d[3]);
    const int valid = __builtin_amdgcn_readfirstlane(d[4]), late = __builtin_amdgcn_readfirstlane(d[5]);
    if (!valid) break;
    const int cnt = c1 - c0;
    const float4* rowsB = rows + off * gs;
    const int* ptrL = ptrS + slot * kPtrSlot;
    const int* idxL = idxS + slot * kIdxPitch;
    const float* cfL = cfS + slot * kCap * kMaxFeat;
    if (g < groups) {
      for (int li = g; li < cnt; li += groups) {
        const int i = c0 + li;
        const int beg = ptrL[li] - e0, end = ptrL[li + 1] - e0;
        float di = 1.f;
        if (WEIGHT) di = dinv[i];
        float4 acc = f4_zero();
        // (batches of kBatch edges; clamped unconditional index reads; wave-uniform slow branch: see k_neighbor_sum_tile)
        auto batch = [&](int p, int nv, auto full_tag) {
          constexpr bool FULL = decltype(full_tag)::value;
          unsigned dd[kBatch];
          bool slow = p + kBatch > kIdxSlot;
          // (a full batch reads kBatch adjacent ids from one clamped start -- two ds_read2_b32 instead of four clamped ds_read_b32;
          // ids past the staged ones are padding the slow branch below never uses)
          const int pb = min(p, kIdxSlot);
#pragma unroll
          for (int j = 0; j < kBatch; ++j)
            dd[j] = (unsigned)(idxL[FULL ? pb + j : min(min(p + j, p + nv - 1), kIdxSlot - 1)] - c0);
#pragma unroll
          for (int j = 0; j < kBatch; ++j) slow |= (FULL || j < nv) & (dd[j] >= (unsigned)cnt);
          if (__any(slow)) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
              if (FULL || j < nv) {
                const int sj = p + j < kIdxSlot ? (int)dd[j] + c0 : nbr[e0 + p + j];
                const bool far = sj < c0 || sj >= c1;
                float4 v = rowsB[(far ? li : sj - c0) * gs + c4];
                if (far) v = x4[(int64_t)sj * ldx4 + c4];
                if (WEIGHT) v = f4_scale(v, di * dinv[sj]);
                acc = f4_add(acc, v);
              }
            }
          } else {
            float4 v[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) v[j] = rowsB[dd[j] * gs + c4];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
              if (FULL || j < nv) {
                if (WEIGHT) v[j] = f4_scale(v[j], di * dinv[(int)dd[j] + c0]);
                acc = f4_add(acc, v[j]);
              }
            }
          }
        };
        int p = beg;
        for (; p + kBatch <= end; p += kBatch) batch(p, kBatch, std::true_type{});
        if (p < end) batch(p, end - p, std::false_type{});
        float4 self = rowsB[li * gs + c4];
        if (WEIGHT) self = f4_scale(self, di * di);
        acc = f4_add(acc, self);
        if (cfeat) {
          const bool same = fout == out;
          float4 f = same ? acc : f4_zero();
#pragma unroll
          for (int tt = 0; tt < kMaxFeat; ++tt) {
            if (tt < kc) {
              const float c = cfL[li * kc + tt];
              f.x = fmaf(c, tv[tt].x, f.x); f.y = fmaf(c, tv[tt].y, f.y); f.z = fmaf(c, tv[tt].z, f.z); f.w = fmaf(c, tv[tt].w, f.w);
            }
          }
          if (same) acc = f;
          else if (NT) __builtin_nontemporal_store(__builtin_bit_cast(v4f_tile, f), reinterpret_cast<v4f_tile*>(fout) + ((int64_t)i * (ldf >> 2) + c4));
          else reinterpret_cast<float4*>(fout)[(int64_t)i * (ldf >> 2) + c4] = f;
        }
        if (NT) __builtin_nontemporal_store(__builtin_bit_cast(v4f_tile, acc), reinterpret_cast<v4f_tile*>(out) + ((int64_t)i * ldo4 + c4));
        else reinterpret_cast<float4*>(out)[(int64_t)i * ldo4 + c4] = acc;
      }
    }
    if (late) tile_barrier();  // B_k
    slot ^= 1;
  }
}

}  // namespace

extern "C" {

size_t pgnn_graph_tiles_workspace_bytes(int64_t num_nodes) { return (size_t)num_nodes + 256; }

int pgnn_graph_tiles(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                     int64_t num_nodes, int32_t* tile_start, int32_t* num_tiles, void* ws, size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(num_nodes > 0 && num_nodes < (1ll << 31), "graph_tiles: bad size");
  if (ws_bytes < pgnn_graph_tiles_workspace_bytes(num_nodes)) {
    set_error("graph_tiles workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  uint8_t* covered = static_cast<uint8_t*>(ws);
  PGNN_HIP(hipMemsetAsync(covered, 0, (size_t)num_nodes, st));
  hipLaunchKernelGGL(k_tile_cover, dim3((int)ceil_div(num_nodes, 256)), dim3(256), 0, st, in_ptr, in_src, out_ptr, out_dst,
                     (int)num_nodes, covered);
  hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(1024), 0, st, covered, (int)num_nodes, tile_start, num_tiles);
  return check_launch("graph_tiles");
}

int pgnn_neighbor_sum_tiled(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const float* dinv,
                            const int32_t* tile_start, const int32_t* num_tiles, float* out, int64_t ldo, int64_t num_nodes,
                            int64_t dim, const float* cfeat, int64_t kc, const float* table, int64_t ldt, float* feat_out,
                            int64_t ld_feat_out, pgnn_stream stream) {
  return pgnn::neighbor_sum_tiled_masked(x, ldx, ptr, nbr, dinv, tile_start, num_tiles, out, ldo, num_nodes, dim, cfeat, kc, table, ldt,
                                         feat_out, ld_feat_out, nullptr, 0, nullptr, (hipStream_t)stream);
}

}  // extern "C"

int pgnn::neighbor_sum_tiled_masked(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const float* dinv,
                                    const int32_t* tile_start, const int32_t* num_tiles, float* out, int64_t ldo, int64_t num_nodes,
                                    int64_t dim, const float* cfeat, int64_t kc, const float* table, int64_t ldt, float* feat_out,
                                    int64_t ld_feat_out, const float* mask, int64_t ldm, bool* mask_applied, hipStream_t stream) {
  if (mask_applied) *mask_applied = false;
  PGNN_REQUIRE(!mask || (mask_applied && !cfeat && !dinv && ldm % 4 == 0), "neighbor_sum_tiled: a mask goes with the plain sum only");
  PGNN_REQUIRE(num_nodes > 0 && dim > 0 && dim % 4 == 0 && dim / 4 <= kThreads && ldx % 4 == 0 && ldo % 4 == 0,
               "neighbor_sum_tiled: bad shape");
  PGNN_REQUIRE(cfeat == nullptr || (kc > 0 && kc <= kMaxFeat && table && feat_out && ldt % 4 == 0 && ld_feat_out % 4 == 0),
               "neighbor_sum_tiled: bad edge-feature arguments (kc <= %d)", kMaxFeat);
  const size_t lds = (size_t)kRows * dim * sizeof(float) + (size_t)(kIdxCap + kRows + 4) * sizeof(int) +
                     (size_t)kMaxFeat * dim * sizeof(float) + (size_t)kRows * kMaxFeat * sizeof(float) + 64;
  if (lds > 160 * 1024) {
    // the 112-row tile of a wider feature matrix (dim > 308) does not fit the LDS: the row-streaming kernel + the edge-feature
    // product as a launch of its own -- the path this entry replaced, same sums in the same order (a bio GNN with
    // --emb_dim 512 used to fail here)
    if (int rc = pgnn_neighbor_sum(x, ldx, ptr, nbr, dinv, out, ldo, num_nodes, dim, stream)) return rc;
    if (cfeat) return pgnn_rowfeat_matmul_fwd(cfeat, kc, table, ldt, feat_out, ld_feat_out, num_nodes, dim, feat_out == out ? 1 : 0, stream);
    return PGNN_OK;
  }
  hipStream_t st = (hipStream_t)stream;
#define PGNN_TILE_ARGS x, ldx, ptr, nbr, dinv, tile_start, num_tiles, out, ldo, (int)num_nodes, (int)dim, cfeat, (int)kc, table, ldt, feat_out, ld_feat_out
  const int dbg = env_knob("PGNN_TILE_DEBUG", 0);
  const size_t lds_pipe = (size_t)kCap * dim * sizeof(float) + (size_t)(2 * kIdxPitch + 2 * kPtrSlot) * sizeof(int) +
                          (size_t)2 * kCap * kMaxFeat * sizeof(float) + (size_t)kMaxFeat * dim * sizeof(float) + 64 + 64;
  // the pipelined kernel pays for its single loader wave and its tickets when a CU gets one or two tiles (256 ego nets: 25 us
  // against 17); from ~128 rows per CU on it wins (4 096 ego nets, neighbour sum + edge-feature product: 168 us against 247).
  // PGNN_TILE_PIPE: 0 = never, 2 = always
  const int pipe_knob = env_knob("PGNN_TILE_PIPE", 1);
  if (dbg == 0 && lds_pipe <= 160 * 1024 && (pipe_knob == 2 || (pipe_knob == 1 && num_nodes >= (int64_t)128 * num_cu()))) {
    // persistent blocks, one per CU; rows loaded / stored non-temporally once x + out exceed the Infinity Cache (aggregate.hip's rule)
    const int pblocks = (int)std::min<int64_t>(num_nodes, (int64_t)num_cu());
    const bool nt = (size_t)num_nodes * dim * 8 > ((size_t)200 << 20);
    static std::atomic<unsigned> next_slot{0};
    const int slot = (int)(next_slot.fetch_add(1, std::memory_order_relaxed) % kTicketSlots);
#define PGNN_TILE_PIPE_LAUNCH(W, N)                                                                              \
  do {                                                                                                            \
    allow_big_lds((const void*)k_neighbor_sum_tile_pipe<W, N>, lds_pipe);                                         \
    hipLaunchKernelGGL((k_neighbor_sum_tile_pipe<W, N>), dim3(pblocks), dim3(kThreads), lds_pipe, st, PGNN_TILE_ARGS, slot); \
  } while (0)
    if (dinv) {
      if (nt) PGNN_TILE_PIPE_LAUNCH(true, true); else PGNN_TILE_PIPE_LAUNCH(true, false);
    } else {
      if (nt) PGNN_TILE_PIPE_LAUNCH(false, true); else PGNN_TILE_PIPE_LAUNCH(false, false);
    }
#undef PGNN_TILE_PIPE_LAUNCH
    return check_launch("neighbor_sum_tiled");
  }
  const int blocks = (int)std::min<int64_t>(num_nodes, (int64_t)num_cu() * 4);
  if (dbg == 1) {
    allow_big_lds((const void*)k_neighbor_sum_tile<false, 1>, lds);
    hipLaunchKernelGGL((k_neighbor_sum_tile<false, 1>), dim3(blocks), dim3(kThreads), lds, st, PGNN_TILE_ARGS);
  } else if (dbg == 2) {
    allow_big_lds((const void*)k_neighbor_sum_tile<false, 2>, lds);
    hipLaunchKernelGGL((k_neighbor_sum_tile<false, 2>), dim3(blocks), dim3(kThreads), lds, st, PGNN_TILE_ARGS);
  } else if (dinv) {
    allow_big_lds((const void*)k_neighbor_sum_tile<true>, lds);
    hipLaunchKernelGGL(k_neighbor_sum_tile<true>, dim3(blocks), dim3(kThreads), lds, st, PGNN_TILE_ARGS);
  } else if (mask && (dim / 4 + 7) / 8 <= kMaxFeat && env_knob("PGNN_TILE_MASK", 1) != 0) {
    allow_big_lds((const void*)k_neighbor_sum_tile<false, 0, true>, lds);
    hipLaunchKernelGGL((k_neighbor_sum_tile<false, 0, true>), dim3(blocks), dim3(kThreads), lds, st, PGNN_TILE_ARGS, mask, ldm);
    *mask_applied = true;
  } else {
    allow_big_lds((const void*)k_neighbor_sum_tile<false>, lds);
    hipLaunchKernelGGL(k_neighbor_sum_tile<false>, dim3(blocks), dim3(kThreads), lds, st, PGNN_TILE_ARGS);
  }
#undef PGNN_TILE_ARGS
  return check_launch("neighbor_sum_tiled");
}
