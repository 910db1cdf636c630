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
#include "common.h"

using namespace pgnn;

namespace {

constexpr int kRows = 48;        // rows of x resident per chunk: 57.6 KB at D = 300 -> two blocks per CU
constexpr int kIdxCap = 2048;    // staged neighbour indices per chunk (8 KB); the rest is read from memory
constexpr int kThreads = 640;    // 8 groups of 75 threads at D = 300

#define PGNN_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PGNN_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

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

template <bool WEIGHT>
__global__ void __launch_bounds__(kThreads) k_neighbor_sum_tile(const float* __restrict__ x, int64_t ldx,
                                                                const int32_t* __restrict__ ptr, const int32_t* __restrict__ nbr,
                                                                const float* __restrict__ dinv,
                                                                const int32_t* __restrict__ tile_start,
                                                                const int32_t* __restrict__ num_tiles, float* __restrict__ out,
                                                                int64_t ldo, int n, int dim) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float smem[];
  const int gs = dim >> 2;
  float4* rows = reinterpret_cast<float4*>(smem);                       // [kRows][gs]
  int* idxL = reinterpret_cast<int*>(rows + kRows * gs);                // [kIdxCap]
  int* ptrL = idxL + kIdxCap;                                           // [kRows + 1]
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  const int64_t ldx4 = ldx >> 2, ldo4 = ldo >> 2;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = kThreads / 64;
  const int groups = kThreads / gs;
  const int g = t / gs, c4 = t - g * gs;
  const int T = *num_tiles;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int a = tile_start[tile], b = tile_start[tile + 1];
    for (int c0 = a; c0 < b; c0 += kRows) {
      const int c1 = min(b, c0 + kRows), cnt = c1 - c0;
      __syncthreads();  // previous chunk fully consumed
      // rows [c0, c1) -> LDS, 1 KiB per wave instruction
      const int total4 = cnt * gs;
      for (int base = wave * 64; base < total4; base += nwaves * 64) {
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (g < groups) {
        for (int li = g; li < cnt; li += groups) {
          const int i = c0 + li;
          const int beg = ptrL[li], end = ptrL[li + 1];
          float di = 1.f;
          if (WEIGHT) di = dinv[i];
          float4 acc = f4_zero();
          for (int p = beg; p < end; ++p) {
            const int s = p < kIdxCap ? idxL[p] : nbr[e0 + p];
            float4 v = (s >= c0 && s < c1) ? rows[(s - c0) * gs + c4] : x4[(int64_t)s * ldx4 + c4];
            if (WEIGHT) v = f4_scale(v, di * dinv[s]);
            acc = f4_add(acc, v);
          }
          float4 self = rows[li * gs + c4];
          if (WEIGHT) self = f4_scale(self, di * di);
          acc = f4_add(acc, self);
          reinterpret_cast<float4*>(out)[(int64_t)i * ldo4 + c4] = acc;
        }
      }
    }
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
                            int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(num_nodes > 0 && dim > 0 && dim % 4 == 0 && dim / 4 <= kThreads && ldx % 4 == 0 && ldo % 4 == 0,
               "neighbor_sum_tiled: bad shape");
  const size_t lds = (size_t)kRows * dim * sizeof(float) + (size_t)(kIdxCap + kRows + 1) * sizeof(int) + 64;
  PGNN_REQUIRE(lds <= 160 * 1024, "neighbor_sum_tiled: feature width %lld too wide for the LDS tile", (long long)dim);
  const int blocks = (int)std::min<int64_t>(num_nodes, (int64_t)num_cu() * 2 * 4);
  hipStream_t st = (hipStream_t)stream;
  if (dinv) {
    allow_big_lds((const void*)k_neighbor_sum_tile<true>, lds);
    hipLaunchKernelGGL(k_neighbor_sum_tile<true>, dim3(blocks), dim3(kThreads), lds, st, x, ldx, ptr, nbr, dinv, tile_start,
                       num_tiles, out, ldo, (int)num_nodes, (int)dim);
  } else {
    allow_big_lds((const void*)k_neighbor_sum_tile<false>, lds);
    hipLaunchKernelGGL(k_neighbor_sum_tile<false>, dim3(blocks), dim3(kThreads), lds, st, x, ldx, ptr, nbr, dinv, tile_start,
                       num_tiles, out, ldo, (int)num_nodes, (int)dim);
  }
  return check_launch("neighbor_sum_tiled");
}

}  // extern "C"
