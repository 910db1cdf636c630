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

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock), 8 * kNumCU)); }

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
