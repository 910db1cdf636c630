// Device-side batching for a dataset that lives in HBM.
//
// The reference collates on the host: DataLoaderMasking workers run MaskAtom per graph
// (chem/util.py:225-277) and BatchMasking.from_data_list concatenates ~256 tiny tensors per batch
// (chem/batch.py:17-52), then the int64 COO batch crosses PCIe.  An MI355X holds the whole corpus
// (ZINC-2M in InMemoryDataset form is ~4.5 GB of 288 GB), so here the dataset stays resident in the
// same concatenated (data, slices) layout InMemoryDataset stores on disk (chem/loader.py), and a
// batch is built by a handful of kernels from a list of graph ids: offsets (one block scan), node and
// edge gathers (one thread per output row, graph found by binary search in the batch offsets), and
// MaskAtom as a per-graph random-key ranking.  Pure integer/byte work, HBM/latency-bound.
#include "common.h"

using namespace pgnn;

namespace {

constexpr int kBlock = 256;

// ---- counter-based random keys (splitmix64 finaliser over (seed, graph id, local atom)) ----------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t graph_stream(uint64_t seed, int64_t graph_id) {
  return mix64(seed ^ mix64((uint64_t)graph_id));
}
__device__ __forceinline__ uint64_t atom_key(uint64_t stream, int64_t local) { return mix64(stream + (uint64_t)local); }

__device__ __forceinline__ int64_t mask_count(int64_t n, double rate) {
  // int(num_atoms * mask_rate + 1) in double precision, as Python evaluates it (chem/util.py:232)
  return n > 0 ? min((int64_t)((double)n * rate + 1.0), n) : 0;
}

// One block: exclusive scans of the per-graph node / edge / masked-atom counts of the batch.
// off arrays are [B+1]; status bit 0: graph id out of range, bit 1: totals differ from the caller's.
__global__ void __launch_bounds__(1024) k_batch_offsets(const int64_t* __restrict__ ids, int64_t B, int64_t G,
                                                        const int64_t* __restrict__ node_slice,
                                                        const int64_t* __restrict__ edge_slice, double rate,
                                                        int unit, int64_t* __restrict__ node_off, int64_t* __restrict__ edge_off,
                                                        int64_t* __restrict__ mask_off, int64_t want_n, int64_t want_e,
                                                        int64_t want_m, int32_t* __restrict__ status) {
  __shared__ int64_t sh[3][1024];
  __shared__ int64_t carry[3];
  const int t = threadIdx.x;
  if (t < 3) carry[t] = 0;
  __syncthreads();
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t i = base + t;
    int64_t v[3] = {0, 0, 0};
    if (i < B) {
      int64_t g = ids[i];
      if (g < 0 || g >= G) {
        atomicOr(status, 1);
        g = min(max(g, (int64_t)0), G - 1);
      }
      v[0] = node_slice[g + 1] - node_slice[g];
      v[1] = edge_slice[g + 1] - edge_slice[g];
      v[2] = unit == 1 ? mask_count(v[0], rate) : unit == 2 ? mask_count(v[1] / 2, rate) : 0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) sh[c][t] = v[c];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan, three columns at once
      int64_t a[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) a[c] = t >= d ? sh[c][t - d] : 0;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) sh[c][t] += a[c];
      __syncthreads();
    }
    if (i < B) {
      node_off[i] = carry[0] + sh[0][t] - v[0];
      edge_off[i] = carry[1] + sh[1][t] - v[1];
      mask_off[i] = carry[2] + sh[2][t] - v[2];
    }
    __syncthreads();
    if (t < 3) carry[t] += sh[t][1023];
    __syncthreads();
  }
  if (t == 0) {
    node_off[B] = carry[0];
    edge_off[B] = carry[1];
    mask_off[B] = carry[2];
    if (carry[0] != want_n || carry[1] != want_e || (unit != 0 && carry[2] != want_m)) atomicOr(status, 2);
  }
}

// largest g with off[g] <= p   (off is [B+1], non-decreasing, off[0] = 0, p < off[B])
__device__ __forceinline__ int64_t find_graph(const int64_t* __restrict__ off, int64_t B, int64_t p) {
  int64_t lo = 0, hi = B;  // invariant: off[lo] <= p < off[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (off[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// x rows (row_words 32-bit words each) + the batch vector
__global__ void __launch_bounds__(kBlock) k_gather_nodes(const int64_t* __restrict__ ids, int64_t B,
                                                         const int64_t* __restrict__ node_slice,
                                                         const int64_t* __restrict__ node_off,
                                                         const uint32_t* __restrict__ x_all, int row_words,
                                                         uint32_t* __restrict__ x_out, int64_t* __restrict__ batch,
                                                         int64_t G) {
  const int64_t n = node_off[B];
  for (int64_t p = blockIdx.x * (int64_t)kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int64_t g = find_graph(node_off, B, p);
    const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
    const int64_t src = node_slice[id] + (p - node_off[g]);
    for (int w = 0; w < row_words; ++w) x_out[p * row_words + w] = x_all[src * row_words + w];
    batch[p] = g;
  }
}

// edge_index (shifted by the graph's node offset in the batch) + edge_attr rows
__global__ void __launch_bounds__(kBlock) k_gather_edges(const int64_t* __restrict__ ids, int64_t B,
                                                         const int64_t* __restrict__ edge_slice,
                                                         const int64_t* __restrict__ node_off,
                                                         const int64_t* __restrict__ edge_off,
                                                         const int64_t* __restrict__ ei_all, int64_t e_all,
                                                         const uint32_t* __restrict__ attr_all, int attr_words,
                                                         int64_t* __restrict__ ei_out, uint32_t* __restrict__ attr_out,
                                                         int64_t G) {
  const int64_t e = edge_off[B];
  for (int64_t q = blockIdx.x * (int64_t)kBlock + threadIdx.x; q < e; q += (int64_t)gridDim.x * kBlock) {
    const int64_t g = find_graph(edge_off, B, q);
    const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
    const int64_t src = edge_slice[id] + (q - edge_off[g]);
    const int64_t shift = node_off[g];
    ei_out[q] = ei_all[src] + shift;
    ei_out[e + q] = ei_all[e_all + src] + shift;
    for (int w = 0; w < attr_words; ++w) attr_out[q * attr_words + w] = attr_all[src * attr_words + w];
  }
}

// The batch's graph structure by OFFSET-ADD ("emitting int32 CSR + packed attrs directly removes K1/K2").  A batch
// is a block-diagonal union of whole graphs whose nodes and edges stay contiguous (chem/batch.py:31-52), so both CSRs of the batch
// are the concatenation of the per-graph CSRs: row pointers shifted by the graph's edge offset, node ids by its node offset, bond
// codes / per-node feature sums / normalisers copied.  The per-graph CSRs are slices of ONE structure built over the whole dataset
// when it went to HBM (ds_* arrays, dataset-global positions) -- no histogram, scan or sort per batch.  One thread per output node
// (p < n), per output edge (n <= p < n + e), and one for the two closing row pointers.
__global__ void __launch_bounds__(kBlock) k_collate_structure(
    const int64_t* __restrict__ ids, int64_t B, int64_t G, const int64_t* __restrict__ node_slice, const int64_t* __restrict__ edge_slice,
    const int64_t* __restrict__ node_off, const int64_t* __restrict__ edge_off, const int32_t* __restrict__ ds_in_ptr,
    const int32_t* __restrict__ ds_in_src, const uint8_t* __restrict__ ds_in_code, const int32_t* __restrict__ ds_out_ptr,
    const int32_t* __restrict__ ds_out_dst, const float* __restrict__ ds_dinv, const float* __restrict__ ds_cfeat, int kc,
    int32_t* __restrict__ in_ptr, int32_t* __restrict__ in_src, uint8_t* __restrict__ in_code, int32_t* __restrict__ out_ptr,
    int32_t* __restrict__ out_dst, float* __restrict__ dinv, float* __restrict__ cfeat) {
  const int64_t n = node_off[B], e = edge_off[B];
  for (int64_t p = blockIdx.x * (int64_t)kBlock + threadIdx.x; p <= n + e; p += (int64_t)gridDim.x * kBlock) {
    if (p < n) {
      const int64_t g = find_graph(node_off, B, p);
      const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
      const int64_t src = node_slice[id] + (p - node_off[g]);
      const int64_t shift = edge_off[g] - edge_slice[id];
      in_ptr[p] = (int32_t)(ds_in_ptr[src] + shift);
      out_ptr[p] = (int32_t)(ds_out_ptr[src] + shift);
      dinv[p] = ds_dinv[src];
      for (int w = 0; w < kc; ++w) cfeat[p * kc + w] = ds_cfeat[src * kc + w];
    } else if (p < n + e) {
      const int64_t q = p - n;
      const int64_t g = find_graph(edge_off, B, q);
      const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
      const int64_t src = edge_slice[id] + (q - edge_off[g]);
      const int64_t shift = node_off[g] - node_slice[id];
      in_src[q] = (int32_t)(ds_in_src[src] + shift);
      out_dst[q] = (int32_t)(ds_out_dst[src] + shift);
      if (in_code) in_code[q] = ds_in_code[src];
    } else {
      in_ptr[n] = (int32_t)e;
      out_ptr[n] = (int32_t)e;
    }
  }
}

// MaskAtom / MaskEdge selection.  Items are atoms (div = 1, unit_off = node offsets) or undirected
// edges (div = 2, unit_off = directed-edge offsets; the reference stores both directions adjacently and
// samples pairs, bio/util.py:77-83).  Item (graph g, local a) is masked iff fewer than k_g items of its
// graph have a smaller (key, local) pair; its position among the graph's masked items is that rank, so
// the k_g indices come out in random-key order (random.sample order is random too, chem/util.py:233).
// Output = batch position of the atom, or of the FIRST direction of the edge pair.
__global__ void __launch_bounds__(kBlock) k_mask_select(const int64_t* __restrict__ ids, int64_t B,
                                                        const int64_t* __restrict__ unit_off, int div,
                                                        const int64_t* __restrict__ mask_off, uint64_t seed,
                                                        int64_t* __restrict__ masked_idx) {
  const int64_t n = unit_off[B] / div;
  for (int64_t p = blockIdx.x * (int64_t)kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int64_t g = find_graph(unit_off, B, p * div);
    const int64_t n_g = (unit_off[g + 1] - unit_off[g]) / div, k_g = mask_off[g + 1] - mask_off[g];
    const int64_t a = p - unit_off[g] / div;
    const uint64_t stream = graph_stream(seed, ids[g]);
    const uint64_t mine = atom_key(stream, a);
    int64_t rank = 0;
    for (int64_t o = 0; o < n_g && rank < k_g; ++o) {
      const uint64_t other = atom_key(stream, o);
      rank += (other < mine || (other == mine && o < a)) ? 1 : 0;
    }
    if (rank < k_g) masked_idx[mask_off[g] + rank] = p * div;
  }
}

// bio MaskEdge (bio/util.py:85-102): label := attr row of the first direction; both directions :=
// [0,0,0,0,0,0,0,0,1] (generally: zeros with a one in the last column).  Pairs are distinct.
__global__ void __launch_bounds__(kBlock) k_mask_edges_apply(const int64_t* __restrict__ masked_idx, int64_t m,
                                                             float* __restrict__ attr, int cols, int64_t e,
                                                             float* __restrict__ label, int32_t* __restrict__ status) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const int64_t q = masked_idx[i];
    if (q < 0 || q + 1 >= e || (q & 1)) {
      atomicOr(status, 4);
      for (int c = 0; c < cols; ++c) label[i * cols + c] = 0.f;
      continue;
    }
    for (int c = 0; c < cols; ++c) {
      label[i * cols + c] = attr[q * cols + c];
      const float v = c == cols - 1 ? 1.f : 0.f;
      attr[q * cols + c] = v;
      attr[(q + 1) * cols + c] = v;
    }
  }
}

// label := original feature row; row := mask token (chem/util.py:236-244).  Indices are distinct.
__global__ void __launch_bounds__(kBlock) k_mask_apply(const int64_t* __restrict__ masked_idx, int64_t m,
                                                       int64_t* __restrict__ x, int64_t cols, int64_t n,
                                                       int64_t token0, int64_t* __restrict__ label,
                                                       int32_t* __restrict__ status) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const int64_t p = masked_idx[i];
    if (p < 0 || p >= n) {
      atomicOr(status, 4);
      for (int64_t c = 0; c < cols; ++c) label[i * cols + c] = 0;
      continue;
    }
    for (int64_t c = 0; c < cols; ++c) {
      label[i * cols + c] = x[p * cols + c];
      x[p * cols + c] = c == 0 ? token0 : 0;
    }
  }
}

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock), 8 * num_cu())); }

}  // namespace

extern "C" {

int pgnn_batch_offsets(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs, const int64_t* node_slice,
                       const int64_t* edge_slice, double mask_rate, int mask_unit, int64_t* node_off, int64_t* edge_off,
                       int64_t* mask_off, int64_t expect_nodes, int64_t expect_edges, int64_t expect_masked,
                       int32_t* status, pgnn_stream stream) {
  PGNN_REQUIRE(num_graphs > 0 && dataset_graphs > 0 && mask_rate >= 0.0 && mask_rate <= 1.0 && mask_unit >= 0 &&
                   mask_unit <= 2,
               "bad batch_offsets arguments");
  hipLaunchKernelGGL(k_batch_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, graph_ids, num_graphs, dataset_graphs,
                     node_slice, edge_slice, mask_rate, mask_unit, node_off, edge_off, mask_off, expect_nodes, expect_edges,
                     expect_masked, status);
  return check_launch("batch_offsets");
}

int pgnn_collate_graphs(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs, const int64_t* node_slice,
                        const int64_t* edge_slice, const int64_t* node_off, const int64_t* edge_off, const void* x_all,
                        int64_t x_row_bytes, const int64_t* edge_index_all, int64_t edges_all, const void* edge_attr_all,
                        int64_t attr_row_bytes, int64_t num_nodes, int64_t num_edges, void* x, int64_t* edge_index,
                        void* edge_attr, int64_t* batch, pgnn_stream stream) {
  PGNN_REQUIRE(num_graphs > 0 && num_nodes >= 0 && num_edges >= 0 && x_row_bytes > 0 && x_row_bytes % 4 == 0 &&
                   attr_row_bytes >= 0 && attr_row_bytes % 4 == 0,
               "collate_graphs: row sizes must be multiples of 4 bytes");
  hipStream_t st = (hipStream_t)stream;
  if (num_nodes > 0)
    hipLaunchKernelGGL(k_gather_nodes, dim3(grid_for(num_nodes)), dim3(kBlock), 0, st, graph_ids, num_graphs, node_slice,
                       node_off, (const uint32_t*)x_all, (int)(x_row_bytes / 4), (uint32_t*)x, batch, dataset_graphs);
  if (num_edges > 0)
    hipLaunchKernelGGL(k_gather_edges, dim3(grid_for(num_edges)), dim3(kBlock), 0, st, graph_ids, num_graphs, edge_slice,
                       node_off, edge_off, edge_index_all, edges_all, (const uint32_t*)edge_attr_all,
                       (int)(attr_row_bytes / 4), edge_index, (uint32_t*)edge_attr, dataset_graphs);
  return check_launch("collate_graphs");
}

int pgnn_collate_structure(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs, const int64_t* node_slice,
                           const int64_t* edge_slice, const int64_t* node_off, const int64_t* edge_off, const int32_t* ds_in_ptr,
                           const int32_t* ds_in_src, const uint8_t* ds_in_code, const int32_t* ds_out_ptr, const int32_t* ds_out_dst,
                           const float* ds_dinv, const float* ds_cfeat, int64_t cfeat_cols, int64_t num_nodes, int64_t num_edges,
                           int32_t* in_ptr, int32_t* in_src, uint8_t* in_code, int32_t* out_ptr, int32_t* out_dst, float* dinv,
                           float* cfeat, pgnn_stream stream) {
  PGNN_REQUIRE(num_graphs > 0 && num_nodes >= 0 && num_edges >= 0 && cfeat_cols > 0 && cfeat_cols <= 16 && ds_in_ptr && ds_in_src &&
                   ds_out_ptr && ds_out_dst && ds_dinv && ds_cfeat && in_ptr && in_src && out_ptr && out_dst && dinv && cfeat &&
                   (ds_in_code != nullptr) == (in_code != nullptr),
               "bad collate_structure arguments");
  PGNN_REQUIRE(num_nodes < (1ll << 31) && num_edges < (1ll << 31), "collate_structure: the batch exceeds int32 positions");
  hipLaunchKernelGGL(k_collate_structure, dim3(grid_for(num_nodes + num_edges + 1)), dim3(kBlock), 0, (hipStream_t)stream, graph_ids,
                     num_graphs, dataset_graphs, node_slice, edge_slice, node_off, edge_off, ds_in_ptr, ds_in_src, ds_in_code, ds_out_ptr,
                     ds_out_dst, ds_dinv, ds_cfeat, (int)cfeat_cols, in_ptr, in_src, in_code, out_ptr, out_dst, dinv, cfeat);
  return check_launch("collate_structure");
}

int pgnn_mask_select(const int64_t* graph_ids, int64_t num_graphs, const int64_t* unit_off, int unit_div,
                     const int64_t* mask_off, int64_t num_units, uint64_t seed, int64_t* masked_indices,
                     pgnn_stream stream) {
  PGNN_REQUIRE(num_graphs > 0 && num_units >= 0 && (unit_div == 1 || unit_div == 2) && num_units % unit_div == 0,
               "bad mask_select arguments");
  if (num_units == 0) return PGNN_OK;
  hipLaunchKernelGGL(k_mask_select, dim3(grid_for(num_units / unit_div)), dim3(kBlock), 0, (hipStream_t)stream, graph_ids,
                     num_graphs, unit_off, unit_div, mask_off, seed, masked_indices);
  return check_launch("mask_select");
}

int pgnn_mask_edges_apply(const int64_t* masked_edge_idx, int64_t num_masked, float* edge_attr, int64_t attr_cols,
                          int64_t num_edges, float* mask_edge_label, int32_t* status, pgnn_stream stream) {
  PGNN_REQUIRE(num_masked >= 0 && attr_cols > 0 && num_edges >= 0, "bad mask_edges_apply arguments");
  if (num_masked == 0) return PGNN_OK;
  hipLaunchKernelGGL(k_mask_edges_apply, dim3(grid_for(num_masked)), dim3(kBlock), 0, (hipStream_t)stream, masked_edge_idx,
                     num_masked, edge_attr, (int)attr_cols, num_edges, mask_edge_label, status);
  return check_launch("mask_edges_apply");
}

int pgnn_mask_atoms_apply(const int64_t* masked_atom_indices, int64_t num_masked, int64_t* x, int64_t x_cols,
                          int64_t num_nodes, int64_t mask_token, int64_t* mask_node_label, int32_t* status,
                          pgnn_stream stream) {
  PGNN_REQUIRE(num_masked >= 0 && x_cols > 0 && num_nodes > 0, "bad mask_atoms_apply arguments");
  if (num_masked == 0) return PGNN_OK;
  hipLaunchKernelGGL(k_mask_apply, dim3(grid_for(num_masked)), dim3(kBlock), 0, (hipStream_t)stream, masked_atom_indices,
                     num_masked, x, x_cols, num_nodes, mask_token, mask_node_label, status);
  return check_launch("mask_atoms_apply");
}

}  // extern "C"

// =================================================================================================
// ExtractSubstructureContextPair (chem/util.py:96-149) + BatchSubstructContext (chem/batch.py:141-210)
// for a list of graph ids, on the resident dataset.  Per graph: BFS distances from a root atom;
// substructure = atoms within k hops, context = atoms with l1 < dist <= l2, overlap = both.  Induced
// sub-graphs keep atoms in ascending order and bonds in their original order (what G.subgraph +
// reset_idxes give for the nodes; the host restatement in data/synthetic.py uses the same bond order).
// Graphs whose context or overlap is empty are dropped, as the reference's collate does (batch.py:169).
// One wave per graph: molecules have tens of atoms; larger graphs just loop the wave.
// =================================================================================================
namespace {

constexpr int kCtxCols = 6;  // per-graph counts: n_sub, e_sub, n_ctx, e_ctx, n_overlap, kept

// The lanes of one wave exchange per-atom values through global memory (graphs may not fit in LDS):
// agent-scope atomic accesses go to L2, so a value stored by one lane is what another lane loads after
// the fence -- no reliance on the CU's L1 for data written by a neighbouring lane.
__device__ __forceinline__ int ld_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }

__device__ __forceinline__ int wave_excl_rank(bool flag, int lane, int* total) {
  const uint64_t m = __ballot(flag);
  *total = __popcll(m);
  return __popcll(m & ((1ull << lane) - 1ull));
}

// (round 5) One wave per graph.  The BFS used to keep `dist` in global memory: seven levels of load edge -> load dist[u] -> load dist[v] ->
// store -> fence, every link a full memory round trip -- 47 us for 256 molecules, the longest kernel of the context-prediction
// step.  Now, for graphs of up to kPlanLdsNodes nodes (every molecule, every PPI ego net of the shapes here), the distances -- and after
// them the two membership flags the bond pass needs -- live in the wave's slice of LDS, and a lane's first two bonds stay in
// registers: a level is LDS traffic inside one wave.  Larger graphs take the global-memory path unchanged.  Same distances (a BFS
// level is unique), same ranks, same counts: tests/test_gpu_loader.py holds the batches to the host extraction bit for bit.
constexpr int kPlanLdsNodes = 1024;
__device__ __forceinline__ int lds_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void lds_st(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__global__ void __launch_bounds__(256) k_ctx_plan(const int64_t* __restrict__ ids, int64_t B, int64_t G,
                                                  const int64_t* __restrict__ node_slice,
                                                  const int64_t* __restrict__ edge_slice,
                                                  const int64_t* __restrict__ node_off, const int64_t* __restrict__ edge_off,
                                                  const int64_t* __restrict__ ei_all, int64_t e_all,
                                                  const int64_t* __restrict__ roots, uint64_t seed, int k, int l1, int l2,
                                                  int32_t* __restrict__ dist, int32_t* __restrict__ sub_rank,
                                                  int32_t* __restrict__ ctx_rank, int32_t* __restrict__ esub_rank,
                                                  int32_t* __restrict__ ectx_rank, int64_t* __restrict__ counts,
                                                  int64_t* __restrict__ root_out) {
  __shared__ int32_t s_all[4][kPlanLdsNodes];
  const int lane = threadIdx.x & 63;
  const int64_t g = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= B) return;
  int32_t* const sd = s_all[threadIdx.x >> 6];
  const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
  const int64_t n0 = node_off[g], n = node_off[g + 1] - n0;
  const int64_t e0 = edge_off[g], e = edge_off[g + 1] - e0;
  const int64_t es = edge_slice[id];
  const bool lds = n <= kPlanLdsNodes;  // (wave-uniform)
  int64_t root = roots ? roots[g] : (n > 0 ? (int64_t)(atom_key(graph_stream(seed, id), 0x5bd1e995) % (uint64_t)n) : 0);
  root = min(max(root, (int64_t)0), max(n - 1, (int64_t)0));
  if (lane == 0) root_out[g] = root;
  // this lane's first two bonds (graph-local ids): every molecule's, most of an ego net's
  int64_t cu[2] = {0, 0}, cv[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (lane + 64 * j < e) {
      cu[j] = ei_all[es + lane + 64 * j];
      cv[j] = ei_all[e_all + es + lane + 64 * j];
    }
  auto get = [&](int64_t a) -> int { return lds ? lds_ld(&sd[a]) : ld_i32(&dist[n0 + a]); };
  auto put = [&](int64_t a, int v) {
    if (lds) lds_st(&sd[a], v);
    else st_i32(&dist[n0 + a], v);
  };
  auto fence = [&]() {
    if (lds) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    else wave_fence();
  };
  for (int64_t a = lane; a < n; a += 64) put(a, a == root ? 0 : -1);
  // level-synchronous BFS by edge relaxation; writers of one level all store the same value.  The bio transform
  // (k < 0: substructure = the whole graph; l2 < 0: context = everything farther than l1, unreachable nodes included)
  // only has to know which nodes lie within l1 hops.
  const bool whole = k < 0, open_end = l2 < 0;
  const int depth = open_end ? (whole ? l1 : max(k, l1)) : max(k, l2);
  for (int level = 0; level < depth; ++level) {
    fence();
    bool grew = false;
    auto relax = [&](int64_t u, int64_t v) {
      if (get(u) == level && get(v) < 0) {
        put(v, level + 1);
        grew = true;
      }
    };
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (lane + 64 * j < e) relax(cu[j], cv[j]);
    for (int64_t q = lane + 128; q < e; q += 64) relax(ei_all[es + q], ei_all[e_all + es + q]);
    if (!__any(grew)) break;
  }
  fence();
  // ranks of the kept atoms / bonds (stable, by wave ballots), and the five counts
  int n_sub = 0, n_ctx = 0, n_ov = 0;
  for (int64_t b0 = 0; b0 < n; b0 += 64) {
    const int64_t a = b0 + lane;
    const int d = a < n ? get(a) : -1;
    const bool s = a < n && (whole || (d >= 0 && d <= k));
    const bool c = a < n && (open_end ? (d < 0 || d > l1) : (d > l1 && d <= l2));
    int ts, tc, to;
    const int rs = wave_excl_rank(s, lane, &ts), rc = wave_excl_rank(c, lane, &tc);
    (void)wave_excl_rank(s && c, lane, &to);
    if (a < n) {
      st_i32(&sub_rank[n0 + a], s ? n_sub + rs : -1);
      st_i32(&ctx_rank[n0 + a], c ? n_ctx + rc : -1);
      if (lds) {
        st_i32(&dist[n0 + a], d);                   // (the distances as the global path leaves them)
        lds_st(&sd[a], (s ? 1 : 0) | (c ? 2 : 0));  // from here on: membership flags (each lane rewrites the slot it has just read)
      }
    }
    n_sub += ts; n_ctx += tc; n_ov += to;
  }
  fence();
  int e_sub = 0, e_ctx = 0;
  auto member = [&](int64_t u, int64_t v, bool& s, bool& c) {
    if (lds) {
      const int fu = lds_ld(&sd[u]), fv = lds_ld(&sd[v]);
      s = (fu & fv & 1) != 0;
      c = (fu & fv & 2) != 0;
    } else {
      s = ld_i32(&sub_rank[n0 + u]) >= 0 && ld_i32(&sub_rank[n0 + v]) >= 0;
      c = ld_i32(&ctx_rank[n0 + u]) >= 0 && ld_i32(&ctx_rank[n0 + v]) >= 0;
    }
  };
  for (int64_t b0 = 0; b0 < e; b0 += 64) {
    const int64_t q = b0 + lane;
    bool s = false, c = false;
    if (q < e) {
      if (b0 == 0) member(cu[0], cv[0], s, c);
      else if (b0 == 64) member(cu[1], cv[1], s, c);
      else member(ei_all[es + q], ei_all[e_all + es + q], s, c);
    }
    int ts, tc;
    const int rs = wave_excl_rank(s, lane, &ts), rc = wave_excl_rank(c, lane, &tc);
    if (q < e) {
      esub_rank[e0 + q] = s ? e_sub + rs : -1;
      ectx_rank[e0 + q] = c ? e_ctx + rc : -1;
    }
    e_sub += ts; e_ctx += tc;
  }
  if (lane == 0) {
    const bool keep = n_ctx > 0 && n_ov > 0;
    int64_t* c = counts + g * kCtxCols;
    c[0] = keep ? n_sub : 0; c[1] = keep ? e_sub : 0; c[2] = keep ? n_ctx : 0; c[3] = keep ? e_ctx : 0;
    c[4] = keep ? n_ov : 0; c[5] = keep ? 1 : 0;
  }
}

// exclusive scans of the six count columns over the batch's graphs; offs is [6][B+1], totals = offs[c][B]
__global__ void __launch_bounds__(1024) k_ctx_offsets(const int64_t* __restrict__ counts, int64_t B, int64_t* __restrict__ offs) {
  __shared__ int64_t sh[kCtxCols][1024];
  __shared__ int64_t carry[kCtxCols];
  const int t = threadIdx.x;
  if (t < kCtxCols) carry[t] = 0;
  __syncthreads();
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t i = base + t;
    int64_t v[kCtxCols];
#pragma unroll
    for (int c = 0; c < kCtxCols; ++c) {
      v[c] = i < B ? counts[i * kCtxCols + c] : 0;
      sh[c][t] = v[c];
    }
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      int64_t a[kCtxCols];
#pragma unroll
      for (int c = 0; c < kCtxCols; ++c) a[c] = t >= d ? sh[c][t - d] : 0;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < kCtxCols; ++c) sh[c][t] += a[c];
      __syncthreads();
    }
    if (i < B) {
#pragma unroll
      for (int c = 0; c < kCtxCols; ++c) offs[c * (B + 1) + i] = carry[c] + sh[c][t] - v[c];
    }
    __syncthreads();
    if (t < kCtxCols) carry[t] += sh[t][1023];
    __syncthreads();
  }
  if (t < kCtxCols) offs[t * (B + 1) + B] = carry[t];
}

__global__ void __launch_bounds__(kBlock) k_ctx_fill_nodes(
    const int64_t* __restrict__ ids, int64_t B, int64_t G, const int64_t* __restrict__ node_slice,
    const int64_t* __restrict__ node_off, const int64_t* __restrict__ offs, const int64_t* __restrict__ counts,
    const int64_t* __restrict__ root, const int32_t* __restrict__ sub_rank, const int32_t* __restrict__ ctx_rank,
    const int32_t* __restrict__ x_all, int x_cols, int32_t* __restrict__ x_sub, int32_t* __restrict__ x_ctx,
    int64_t* __restrict__ center_idx, int64_t* __restrict__ overlap_idx, int64_t* __restrict__ overlap_batch,
    int64_t* __restrict__ overlap_size) {
  const int64_t n = node_off[B];
  const int64_t* o_nsub = offs;
  const int64_t* o_nctx = offs + 2 * (B + 1);
  const int64_t* o_nov = offs + 4 * (B + 1);
  const int64_t* o_keep = offs + 5 * (B + 1);
  for (int64_t p = blockIdx.x * (int64_t)kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
    const int64_t g = find_graph(node_off, B, p);
    if (!counts[g * kCtxCols + 5]) continue;
    const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
    const int64_t a = p - node_off[g], src = node_slice[id] + a;
    const int rs = sub_rank[p], rc = ctx_rank[p];
    if (rs >= 0) {
      for (int c = 0; c < x_cols; ++c) x_sub[(o_nsub[g] + rs) * x_cols + c] = x_all[src * x_cols + c];
      if (a == root[g]) center_idx[o_keep[g]] = o_nsub[g] + rs;
    }
    if (rc >= 0) {
      for (int c = 0; c < x_cols; ++c) x_ctx[(o_nctx[g] + rc) * x_cols + c] = x_all[src * x_cols + c];
    }
    if (a == 0) overlap_size[o_keep[g]] = counts[g * kCtxCols + 4];
  }
  // overlap lists, atoms in order: one WAVE per graph, 64 atoms a trip, positions by ballot ranks (round 5: one thread per graph
  // walking its atoms paid two dependent loads per atom -- 20 of the kernel's 27 us for 256 molecules)
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t g = wave; g < B; g += nwaves) {
    if (!counts[g * kCtxCols + 5]) continue;  // (wave-uniform)
    int64_t w = o_nov[g];
    const int64_t p1 = node_off[g + 1];
    for (int64_t b0 = node_off[g]; b0 < p1; b0 += 64) {
      const int64_t p = b0 + lane;
      int rc = -1;
      bool both = false;
      if (p < p1) {
        rc = ctx_rank[p];
        both = rc >= 0 && sub_rank[p] >= 0;
      }
      int total;
      const int r = wave_excl_rank(both, lane, &total);
      if (both) {
        overlap_idx[w + r] = o_nctx[g] + rc;
        overlap_batch[w + r] = o_keep[g];
      }
      w += total;
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_ctx_fill_edges(
    const int64_t* __restrict__ ids, int64_t B, int64_t G, const int64_t* __restrict__ edge_slice,
    const int64_t* __restrict__ node_off, const int64_t* __restrict__ edge_off, const int64_t* __restrict__ offs,
    const int64_t* __restrict__ counts, const int32_t* __restrict__ sub_rank, const int32_t* __restrict__ ctx_rank,
    const int32_t* __restrict__ esub_rank, const int32_t* __restrict__ ectx_rank, const int64_t* __restrict__ ei_all,
    int64_t e_all, const int32_t* __restrict__ attr_all, int attr_cols, int64_t* __restrict__ ei_sub, int32_t* __restrict__ ea_sub,
    int64_t* __restrict__ ei_ctx, int32_t* __restrict__ ea_ctx, int ctx_zero_from) {
  const int64_t e = edge_off[B];
  const int64_t* o_nsub = offs;
  const int64_t* o_esub = offs + 1 * (B + 1);
  const int64_t* o_nctx = offs + 2 * (B + 1);
  const int64_t* o_ectx = offs + 3 * (B + 1);
  const int64_t tot_esub = o_esub[B], tot_ectx = o_ectx[B];
  for (int64_t q = blockIdx.x * (int64_t)kBlock + threadIdx.x; q < e; q += (int64_t)gridDim.x * kBlock) {
    const int64_t g = find_graph(edge_off, B, q);
    if (!counts[g * kCtxCols + 5]) continue;
    const int64_t id = min(max(ids[g], (int64_t)0), G - 1);
    const int64_t src = edge_slice[id] + (q - edge_off[g]);
    const int64_t u = node_off[g] + ei_all[src], v = node_off[g] + ei_all[e_all + src];
    const int rs = esub_rank[q], rc = ectx_rank[q];
    if (rs >= 0) {
      const int64_t w = o_esub[g] + rs;
      ei_sub[w] = o_nsub[g] + sub_rank[u];
      ei_sub[tot_esub + w] = o_nsub[g] + sub_rank[v];
      for (int c = 0; c < attr_cols; ++c) ea_sub[w * attr_cols + c] = attr_all[src * attr_cols + c];
    }
    if (rc >= 0) {
      const int64_t w = o_ectx[g] + rc;
      ei_ctx[w] = o_nctx[g] + ctx_rank[u];
      ei_ctx[tot_ectx + w] = o_nctx[g] + ctx_rank[v];
      for (int c = 0; c < attr_cols; ++c) ea_ctx[w * attr_cols + c] = c >= ctx_zero_from ? 0 : attr_all[src * attr_cols + c];
    }
  }
}

}  // namespace

extern "C" {

int pgnn_substruct_context_plan(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs,
                                const int64_t* node_slice, const int64_t* edge_slice, const int64_t* node_off,
                                const int64_t* edge_off, const int64_t* edge_index_all, int64_t edges_all,
                                const int64_t* roots, uint64_t seed, int k, int l1, int l2, int32_t* dist,
                                int32_t* sub_rank, int32_t* ctx_rank, int32_t* esub_rank, int32_t* ectx_rank,
                                int64_t* counts, int64_t* root_out, int64_t* offsets, pgnn_stream stream) {
  PGNN_REQUIRE(num_graphs > 0 && l1 >= 0 && (k >= 0 || k == -1) && (l2 >= l1 || l2 == -1),
               "bad substruct_context_plan arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ctx_plan, dim3((int)ceil_div(num_graphs, 4)), dim3(256), 0, st, graph_ids, num_graphs, dataset_graphs,
                     node_slice, edge_slice, node_off, edge_off, edge_index_all, edges_all, roots, seed, k, l1, l2, dist,
                     sub_rank, ctx_rank, esub_rank, ectx_rank, counts, root_out);
  hipLaunchKernelGGL(k_ctx_offsets, dim3(1), dim3(1024), 0, st, counts, num_graphs, offsets);
  return check_launch("substruct_context_plan");
}

int pgnn_substruct_context_fill(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs,
                                const int64_t* node_slice, const int64_t* edge_slice, const int64_t* node_off,
                                const int64_t* edge_off, const int64_t* offsets, const int64_t* counts,
                                const int64_t* root, const int32_t* sub_rank, const int32_t* ctx_rank,
                                const int32_t* esub_rank, const int32_t* ectx_rank, const void* x_all, int64_t x_row_bytes,
                                const int64_t* edge_index_all, int64_t edges_all, const void* edge_attr_all,
                                int64_t attr_row_bytes, int64_t context_attr_zero_from_byte, int64_t num_nodes,
                                int64_t num_edges, void* x_substruct, int64_t* edge_index_substruct,
                                void* edge_attr_substruct, void* x_context, int64_t* edge_index_context,
                                void* edge_attr_context, int64_t* center_substruct_idx,
                                int64_t* overlap_context_substruct_idx, int64_t* batch_overlapped_context,
                                int64_t* overlapped_context_size, pgnn_stream stream) {
  PGNN_REQUIRE(num_graphs > 0 && x_row_bytes > 0 && attr_row_bytes > 0 && x_row_bytes % 4 == 0 && attr_row_bytes % 4 == 0 &&
                   (context_attr_zero_from_byte < 0 || context_attr_zero_from_byte % 4 == 0),
               "bad substruct_context_fill arguments (feature rows are copied as 32-bit words)");
  hipStream_t st = (hipStream_t)stream;
  const int xw = (int)(x_row_bytes / 4), aw = (int)(attr_row_bytes / 4);
  const int zero_from = context_attr_zero_from_byte < 0 ? aw : (int)(context_attr_zero_from_byte / 4);
  hipLaunchKernelGGL(k_ctx_fill_nodes, dim3(grid_for(std::max(num_nodes, num_graphs))), dim3(kBlock), 0, st, graph_ids,
                     num_graphs, dataset_graphs, node_slice, node_off, offsets, counts, root, sub_rank, ctx_rank,
                     static_cast<const int32_t*>(x_all), xw, static_cast<int32_t*>(x_substruct),
                     static_cast<int32_t*>(x_context), center_substruct_idx, overlap_context_substruct_idx,
                     batch_overlapped_context, overlapped_context_size);
  if (num_edges > 0)
    hipLaunchKernelGGL(k_ctx_fill_edges, dim3(grid_for(num_edges)), dim3(kBlock), 0, st, graph_ids, num_graphs,
                       dataset_graphs, edge_slice, node_off, edge_off, offsets, counts, sub_rank, ctx_rank, esub_rank,
                       ectx_rank, edge_index_all, edges_all, static_cast<const int32_t*>(edge_attr_all), aw,
                       edge_index_substruct, static_cast<int32_t*>(edge_attr_substruct), edge_index_context,
                       static_cast<int32_t*>(edge_attr_context), zero_from);
  return check_launch("substruct_context_fill");
}

}  // extern "C"
