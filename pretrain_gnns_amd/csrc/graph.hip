// Graph-structure construction for the message-passing hot path (gfx950).
//
// Replaces, once per batch, what the reference redoes in every layer: add_self_loops + attr cat
// (chem/model.py:39-45, bio/model.py:39-45) and the COO gather/scatter of propagate.  Output is a
// stable CSR by destination (forward) and by source (backward) in int32, a packed bond code per
// edge, the GCN normaliser, and per-node edge-feature sums ("cfeat") that turn the edge-embedding
// gradients into a tall-skinny reduction.  All kernels are HBM/latency-bound integer work.
#include <stdarg.h>

#include <string.h>

#include "common.h"

extern char** environ;

namespace pgnn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

unsigned g_env_generation = 1;
static bool scan_env_for_knobs() {
  for (char** e = ::environ; e && *e; ++e)
    if (strncmp(*e, "PGNN_", 5) == 0) return true;
  return false;
}
bool g_env_any = scan_env_for_knobs();

DeviceInfo device_info() {
  constexpr int kMaxDev = 64;
  static DeviceInfo cache[kMaxDev];  // zero-initialised; a benign race re-queries the same values
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDev) dev = 0;
  if (cache[dev].num_cu == 0) {
    hipDeviceProp_t prop;
    int cu = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cu = prop.multiProcessorCount;
    int xcd = cu / 32;
    if (xcd < 1) xcd = 1;
    if (xcd > 8) xcd = 8;
    cache[dev].num_xcd = xcd;
    cache[dev].num_cu = cu;
  }
  return cache[dev];
}

namespace {

constexpr int kScanItems = 1024;  // per block: 256 threads x 4

struct GroupJob {
  const int64_t* key;
  int64_t stride;
  int32_t* ptr;     // [n_keys+1]
  int32_t* perm;    // [n_items]  final (sorted inside each segment)
  int32_t* cursor;  // [n_keys]   zeroed
  int32_t* tmp;     // [n_items]  unsorted fill
  int32_t* bsum;    // [ceil((n_keys+1)/kScanItems)]
};
struct GroupJobs {
  GroupJob j[2];
};

__global__ void k_hist(GroupJobs jobs, int64_t n_items, int64_t n_keys, int32_t* status) {
  const GroupJob& job = jobs.j[blockIdx.y];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_items;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = job.key[e * job.stride];
    if (k < 0 || k >= n_keys) {
      atomicAdd(status, 1);
      k = 0;  // clamp: keeps every later access in bounds; the caller inspects status
    }
    atomicAdd(&job.ptr[k + 1], 1);
  }
}

__device__ __forceinline__ int block_scan_256(int v, int* lds, int& block_total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  if (lane == 63) lds[w] = v;
  __syncthreads();
  int add = 0;
  for (int i = 0; i < w; ++i) add += lds[i];
  block_total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return v + add;
}

// in-place inclusive scan of each 1024-item chunk; chunk totals to bsum
__global__ void __launch_bounds__(256) k_scan_local(GroupJobs jobs, int64_t n) {
  __shared__ int lds[4];
  const GroupJob& job = jobs.j[blockIdx.y];
  int32_t* d = job.ptr;
  const int64_t base = (int64_t)blockIdx.x * kScanItems + threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (base + i < n) ? d[base + i] : 0;
  v[1] += v[0];
  v[2] += v[1];
  v[3] += v[2];
  int total;
  const int incl = block_scan_256(v[3], lds, total);
  const int excl = incl - v[3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (base + i < n) d[base + i] = v[i] + excl;
  if (threadIdx.x == 0) job.bsum[blockIdx.x] = total;
}

// exclusive scan of the chunk totals (single block per job, any count)
__global__ void __launch_bounds__(256) k_scan_bsum(GroupJobs jobs, int nb) {
  __shared__ int lds[4];
  int32_t* b = jobs.j[blockIdx.y].bsum;
  int carry = 0;
  for (int base = 0; base < nb; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? b[i] : 0;
    int total;
    const int incl = block_scan_256(v, lds, total);
    if (i < nb) b[i] = carry + incl - v;
    carry += total;
  }
}

__global__ void k_scan_add(GroupJobs jobs, int64_t n);

// whole scan in ONE block of 1 024 threads (any n, meant for n <= ~32k): chunks of 8 192 (8 consecutive items per thread) with a
// running carry.  (256 threads x 4 items took seven trips and 9.7 us for the 6 741 + 1 pointers of a 256-molecule batch.)
constexpr int kScanSingleThreads = 1024;
__global__ void __launch_bounds__(kScanSingleThreads) k_scan_single(GroupJobs jobs, int64_t n) {
  __shared__ int wsum[kScanSingleThreads / 64];
  int32_t* d = jobs.j[blockIdx.y].ptr;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int carry = 0;
  for (int64_t base0 = 0; base0 < n; base0 += kScanSingleThreads * 8) {
    const int64_t base = base0 + tid * 8;
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (base + i < n) ? d[base + i] : 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) v[i] += v[i - 1];
    int incl = v[7];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int add = 0, total = 0;
#pragma unroll
    for (int i = 0; i < kScanSingleThreads / 64; ++i) {
      const int t = wsum[i];
      add += i < w ? t : 0;
      total += t;
    }
    const int excl = carry + add + incl - v[7];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (base + i < n) d[base + i] = v[i] + excl;
    carry += total;
    __syncthreads();
  }
}

constexpr int64_t kScanSingleMax = 32768;

// in-place inclusive scan of jobs.j[0..njobs).ptr[0..n)
inline void launch_scan(GroupJobs jobs, int njobs, int64_t n, hipStream_t st) {
  if (n <= kScanSingleMax) {
    hipLaunchKernelGGL(k_scan_single, dim3(1, njobs), dim3(kScanSingleThreads), 0, st, jobs, n);
    return;
  }
  const int nb = (int)ceil_div(n, kScanItems);
  hipLaunchKernelGGL(k_scan_local, dim3(nb, njobs), dim3(256), 0, st, jobs, n);
  if (nb > 1) {
    hipLaunchKernelGGL(k_scan_bsum, dim3(1, njobs), dim3(256), 0, st, jobs, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb - 1, njobs), dim3(256), 0, st, jobs, n);
  }
}

__global__ void __launch_bounds__(256) k_scan_add(GroupJobs jobs, int64_t n) {
  const GroupJob& job = jobs.j[blockIdx.y];
  const int add = job.bsum[blockIdx.x + 1];
  const int64_t base = (int64_t)(blockIdx.x + 1) * kScanItems + threadIdx.x * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (base + i < n) job.ptr[base + i] += add;
}

__global__ void k_fill(GroupJobs jobs, int64_t n_items, int64_t n_keys) {
  const GroupJob& job = jobs.j[blockIdx.y];
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_items;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = job.key[e * job.stride];
    if (k < 0 || k >= n_keys) k = 0;
    const int pos = atomicAdd(&job.cursor[k], 1);
    job.tmp[job.ptr[k] + pos] = (int32_t)e;
  }
}

// The atomic fill is order-nondeterministic; ranking the (unique) item ids inside each segment
// restores the original order => stable grouping, bitwise-reproducible downstream sums.
// One thread per ITEM slot: its rank is the number of smaller ids in its segment (l loads, balanced over
// the grid whatever the degree distribution -- a thread per segment costs l^2 for its slowest lane, 91 us
// on a PPI batch).  Segments longer than kRankMax are ranked by whole waves instead (thread index = key).
constexpr int kRankMax = 512;
__global__ void __launch_bounds__(256) k_sort_segments(GroupJobs jobs, int64_t n_items, int64_t n_keys) {
  const GroupJob& job = jobs.j[blockIdx.y];
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n_items) {
    const int v = job.tmp[t];
    int64_t k = job.key[(int64_t)v * job.stride];
    if (k < 0 || k >= n_keys) k = 0;
    const int beg = job.ptr[k], len = job.ptr[k + 1] - beg;
    if (len <= kRankMax) {
      int rank = 0;
      for (int b = 0; b < len; ++b) rank += job.tmp[beg + b] < v;
      job.perm[beg + rank] = v;
    }
  }
  int beg = 0, len = 0;
  if (t < n_keys) {
    beg = job.ptr[t];
    len = job.ptr[t + 1] - beg;
  }
  unsigned long long m = __ballot(len > kRankMax);
  const int lane = lane_id();
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int b = __shfl(beg, src), l = __shfl(len, src);
    for (int p = lane; p < l; p += kWave) {
      const int v = job.tmp[b + p];
      int rank = 0;
      for (int q = 0; q < l; ++q) rank += job.tmp[b + q] < v;
      job.perm[b + rank] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Stable partition for FEW keys with LONG segments (atom types, chirality tags, graph ids of a
// small batch): one LSD-radix pass with the key as the digit.  Tile = 256 items; per-tile key
// histograms are laid out key-major, so one exclusive scan over [n_keys][n_tiles] yields, for every
// (key, tile), the output offset of that tile's first item with that key.  Inside a tile the rank of
// an item among equal keys comes from wave ballots (lanes are in item order) plus the counts of the
// earlier waves.  O(n), deterministic, no sort.
// ---------------------------------------------------------------------------------------------
constexpr int kPartTile = 256;
constexpr int kPartMaxKeys = 1024;

// key of item e: key[e*stride], or the pair (key[e*stride], key2[e*stride]) flattened as a*n2 + b when a second
// column is given (n_keys = n1*n2); out-of-range components are clamped to 0 (and counted by the histogram pass)
__device__ __forceinline__ int part_key(const int64_t* __restrict__ key, const int64_t* __restrict__ key2, int64_t stride,
                                        int64_t e, int n_keys, int n2, bool* bad) {
  int64_t a = key[e * stride];
  if (!key2) {
    if (a < 0 || a >= n_keys) { *bad = true; a = 0; }
    return (int)a;
  }
  int64_t b = key2[e * stride];
  const int n1 = n_keys / n2;
  if (a < 0 || a >= n1) { *bad = true; a = 0; }
  if (b < 0 || b >= n2) { *bad = true; b = 0; }
  return (int)(a * n2 + b);
}

__global__ void __launch_bounds__(kPartTile)
k_part_hist(const int64_t* __restrict__ key, const int64_t* __restrict__ key2, int n2, int64_t stride, int64_t n_items,
            int n_keys, int n_tiles, int32_t* __restrict__ offs /*[n_keys*n_tiles + 1], element 0 reserved*/,
            int32_t* status) {
  extern __shared__ int lh[];
  for (int k = threadIdx.x; k < n_keys; k += kPartTile) lh[k] = 0;
  __syncthreads();
  const int64_t e = (int64_t)blockIdx.x * kPartTile + threadIdx.x;
  if (e < n_items) {
    bool bad = false;
    const int k = part_key(key, key2, stride, e, n_keys, n2, &bad);
    if (bad) atomicAdd(status, 1);
    atomicAdd(&lh[k], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_keys; k += kPartTile) offs[(size_t)k * n_tiles + blockIdx.x + 1] = lh[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) offs[0] = 0;  // the reserved element (was a 4-byte memset launch of its own)
}

__global__ void __launch_bounds__(kPartTile)
k_part_scatter(const int64_t* __restrict__ key, const int64_t* __restrict__ key2, int n2, int64_t stride, int64_t n_items,
               int n_keys, int n_tiles, const int32_t* __restrict__ offs, int32_t* __restrict__ ptr,
               int32_t* __restrict__ perm) {
  extern __shared__ int wh[];  // [4][n_keys] per-wave key counts
  for (int q = threadIdx.x; q < 4 * n_keys; q += kPartTile) wh[q] = 0;
  __syncthreads();
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * kPartTile + threadIdx.x;
  const bool valid = e < n_items;
  int k = 0;
  if (valid) {
    bool bad = false;
    k = part_key(key, key2, stride, e, n_keys, n2, &bad);
  }
  int rank = 0;
  unsigned long long todo = __ballot(valid);
  const unsigned long long lt = (1ull << lane) - 1ull;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int kl = __shfl(k, leader);
    const unsigned long long m = __ballot(valid && k == kl);
    if (valid && k == kl) rank = __popcll(m & lt);
    if (lane == leader) wh[w * n_keys + kl] = __popcll(m);
    todo &= ~m;
  }
  __syncthreads();
  if (valid) {
    int base = offs[(size_t)k * n_tiles + blockIdx.x];
    for (int ww = 0; ww < w; ++ww) base += wh[ww * n_keys + k];
    perm[base + rank] = (int32_t)e;
  }
  if (blockIdx.x == 0) {
    for (int q = threadIdx.x; q < n_keys; q += kPartTile) ptr[q] = offs[(size_t)q * n_tiles];
    if (threadIdx.x == 0) ptr[n_keys] = (int32_t)n_items;
  }
}

inline bool use_partition(int64_t n_keys, int64_t n_items) {
  return n_keys <= kPartMaxKeys && n_keys * ceil_div(std::max<int64_t>(n_items, 1), kPartTile) <= (1ll << 24);
}

size_t partition_ws_bytes(int64_t n_keys, int64_t n_items) {
  const int64_t tiles = ceil_div(std::max<int64_t>(n_items, 1), kPartTile);
  const int64_t len = n_keys * tiles + 1;
  return align_up((size_t)len * 4, 256) + align_up((size_t)ceil_div(len, kScanItems) * 4, 256);
}

int run_partition(const int64_t* key, int64_t stride, int64_t n_items, int64_t n_keys, int32_t* ptr,
                  int32_t* perm, int32_t* status, void* ws, hipStream_t st, const int64_t* key2 = nullptr, int n2 = 1) {
  const int n_tiles = (int)ceil_div(std::max<int64_t>(n_items, 1), kPartTile);
  const int64_t len = n_keys * n_tiles + 1;
  Carver cv(ws);
  int32_t* offs = cv.take<int32_t>((size_t)len);
  int32_t* bsum = cv.take<int32_t>((size_t)ceil_div(len, kScanItems));
  hipLaunchKernelGGL(k_part_hist, dim3(n_tiles), dim3(kPartTile), (size_t)n_keys * 4, st, key, key2, n2, stride, n_items,
                     (int)n_keys, n_tiles, offs, status);
  GroupJobs jobs;
  jobs.j[0] = GroupJob{nullptr, 1, offs, nullptr, nullptr, nullptr, bsum};
  jobs.j[1] = jobs.j[0];
  launch_scan(jobs, 1, len, st);
  hipLaunchKernelGGL(k_part_scatter, dim3(n_tiles), dim3(kPartTile), (size_t)4 * n_keys * 4, st, key, key2, n2, stride,
                     n_items, (int)n_keys, n_tiles, offs, ptr, perm);
  return check_launch("group_by_key(partition)");
}

int run_group(GroupJobs jobs, int njobs, int64_t n_items, int64_t n_keys, int32_t* status,
              hipStream_t st) {
  const int64_t n = n_keys + 1;
  const int gi = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n_items, 256), 1), 4096);
  if (n_items > 0) {
    hipLaunchKernelGGL(k_hist, dim3(gi, njobs), dim3(256), 0, st, jobs, n_items, n_keys, status);
  }
  launch_scan(jobs, njobs, n, st);
  if (n_items > 0) {
    hipLaunchKernelGGL(k_fill, dim3(gi, njobs), dim3(256), 0, st, jobs, n_items, n_keys);
    hipLaunchKernelGGL(k_sort_segments, dim3((int)ceil_div(std::max(n_keys, n_items), 256), njobs), dim3(256), 0, st,
                       jobs, n_items, n_keys);
  }
  return check_launch("group_by_key");
}

size_t group_ws_bytes(int64_t n_keys, int64_t n_items) {
  return align_up((size_t)n_keys * 4, 256) + align_up((size_t)n_items * 4, 256) +
         align_up((size_t)ceil_div(n_keys + 1, kScanItems) * 4, 256);
}

// (in-degree + 1)^-1/2, straight from the CSR row pointer: the payload kernels evaluate it for the node and for
// each neighbour instead of waiting for a separate pass over the nodes
__device__ __forceinline__ float dinv_of(const int32_t* __restrict__ in_ptr, int64_t i) {
  return 1.0f / sqrtf((float)(in_ptr[i + 1] - in_ptr[i] + 1));
}

__global__ void __launch_bounds__(256)
k_chem_payload(const int64_t* __restrict__ ei, const int64_t* __restrict__ ea, int64_t E, int64_t N,
               int gcn, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ perm_in,
               int32_t* __restrict__ in_src, uint8_t* __restrict__ in_code,
               const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ perm_out,
               int32_t* __restrict__ out_dst, float* __restrict__ dinv,
               float* __restrict__ cfeat, int32_t* status) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float di = dinv_of(in_ptr, i);
  dinv[i] = di;
  float c[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) c[t] = 0.f;
  for (int p = in_ptr[i]; p < in_ptr[i + 1]; ++p) {
    const int e = perm_in[p];
    int64_t s = ei[E + e];
    if (s < 0 || s >= N) s = 0;  // already counted in status by the histogram of the source keys
    int64_t a0 = ea[2 * (int64_t)e], a1 = ea[2 * (int64_t)e + 1];
    if (a0 < 0 || a0 >= 6 || a1 < 0 || a1 >= 3) {
      atomicAdd(status, 1);
      a0 = 0;
      a1 = 0;
    }
    in_src[p] = (int32_t)s;
    in_code[p] = (uint8_t)(a0 * 3 + a1);
    const float w = gcn ? di * dinv_of(in_ptr, s) : 1.0f;
#pragma unroll
    for (int t = 0; t < 6; ++t) c[t] += (a0 == t) ? w : 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) c[6 + t] += (a1 == t) ? w : 0.f;
  }
  const float ws = gcn ? di * di : 1.0f;  // self loop: bond type 4, direction 0
  c[4] += ws;
  c[6] += ws;
#pragma unroll
  for (int t = 0; t < 9; ++t) cfeat[i * 9 + t] = c[t];
  for (int p = out_ptr[i]; p < out_ptr[i + 1]; ++p) {
    int64_t d = ei[perm_out[p]];
    if (d < 0 || d >= N) d = 0;
    out_dst[p] = (int32_t)d;
  }
}

__global__ void __launch_bounds__(256)
k_bio_payload(const int64_t* __restrict__ ei, const float* __restrict__ ea, int64_t E, int64_t N,
              int gcn, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ perm_in,
              int32_t* __restrict__ in_src, const int32_t* __restrict__ out_ptr,
              const int32_t* __restrict__ perm_out, int32_t* __restrict__ out_dst,
              float* __restrict__ dinv, float* __restrict__ cfeat) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float di = dinv_of(in_ptr, i);
  dinv[i] = di;
  float c[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) c[t] = 0.f;
  for (int p = in_ptr[i]; p < in_ptr[i + 1]; ++p) {
    const int e = perm_in[p];
    int64_t s = ei[E + e];
    if (s < 0 || s >= N) s = 0;
    in_src[p] = (int32_t)s;
    const float w = gcn ? di * dinv_of(in_ptr, s) : 1.0f;
    const float* a = ea + (int64_t)e * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) c[t] += w * a[t];
    c[9] += w;
  }
  const float ws = gcn ? di * di : 1.0f;  // self loop attr = one-hot(7), bio/model.py:42-43
  c[7] += ws;
  c[9] += ws;
#pragma unroll
  for (int t = 0; t < 10; ++t) cfeat[i * 10 + t] = c[t];
  for (int p = out_ptr[i]; p < out_ptr[i + 1]; ++p) {
    int64_t d = ei[perm_out[p]];
    if (d < 0 || d >= N) d = 0;
    out_dst[p] = (int32_t)d;
  }
}

// The same results from 16 lanes per node (4 nodes per wave): lane j of a group fetches edge j of a 16-edge chunk (its id, its
// source, the normaliser) and writes in_src coalesced, then lane t < 10 owns feature column t and adds the chunk's 16 attribute
// values -- all loaded before the first addition -- in edge order, exactly the order and the operations of k_bio_payload.
// One thread per node walked three dependent loads per edge one edge at a time, and a protein's hub nodes (hundreds of
// in-edges, 36-byte attribute rows at random offsets) set the kernel's time: 46 us on a 256-ego-net batch, the largest
// launch of the build.
__global__ void __launch_bounds__(256)
k_bio_payload16(const int64_t* __restrict__ ei, const float* __restrict__ ea, int64_t E, int64_t N,
                int gcn, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ perm_in,
                int32_t* __restrict__ in_src, const int32_t* __restrict__ out_ptr,
                const int32_t* __restrict__ perm_out, int32_t* __restrict__ out_dst,
                float* __restrict__ dinv, float* __restrict__ cfeat) {
  const int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) >> 4;
  if (i >= N) return;  // (a whole group of 16 lanes at a time)
  const int t = threadIdx.x & 15, gbase = lane_id() & ~15;
  const int beg = in_ptr[i], end = in_ptr[i + 1];
  const float di = 1.0f / sqrtf((float)(end - beg + 1));
  if (t == 0) dinv[i] = di;
  float acc = 0.f;
  for (int p0 = beg; p0 < end; p0 += 16) {
    const int p = p0 + t, cnt = min(16, end - p0);
    int e = 0;
    float w = 0.f;
    if (p < end) {
      e = perm_in[p];
      int64_t s = ei[E + e];
      if (s < 0 || s >= N) s = 0;
      in_src[p] = (int32_t)s;
      w = gcn ? di * dinv_of(in_ptr, s) : 1.0f;
    }
    float av[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ej = __shfl(e, gbase + j);
      av[j] = (j < cnt && t < 9) ? ea[(int64_t)ej * 9 + t] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float wj = __shfl(w, gbase + j);
      if (j < cnt) {
        if (t < 9) acc += wj * av[j];
        else if (t == 9) acc += wj;
      }
    }
  }
  const float ws = gcn ? di * di : 1.0f;  // self loop attr = one-hot(7), bio/model.py:42-43
  if (t == 7 || t == 9) acc += ws;
  if (t < 10) cfeat[i * 10 + t] = acc;
  for (int p = out_ptr[i] + t; p < out_ptr[i + 1]; p += 16) {
    int64_t d = ei[perm_out[p]];
    if (d < 0 || d >= N) d = 0;
    out_dst[p] = (int32_t)d;
  }
}

// the three zero-initialised arrays of a graph build in ONE launch (three hipMemsetAsync calls are three fill kernels)
struct ZeroJobs {
  int32_t* p[3];
  int64_t n[3];
};
__global__ void __launch_bounds__(256) k_zero_words(ZeroJobs z) {
  int32_t* __restrict__ p = z.p[blockIdx.y];
  const int64_t n = z.n[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

struct GraphWs {
  GroupJobs jobs;
  int32_t *perm_in, *perm_out;
};

int prepare_graph_jobs(const int64_t* ei, int64_t E, int64_t N, int32_t* in_ptr, int32_t* out_ptr,
                       void* ws, size_t ws_bytes, hipStream_t st, GraphWs& g) {
  if (ws_bytes < pgnn_graph_workspace_bytes(N, E)) {
    set_error("graph workspace too small: %zu < %zu", ws_bytes, pgnn_graph_workspace_bytes(N, E));
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  int32_t* cursors = cv.take<int32_t>(2 * (size_t)N);
  g.perm_in = cv.take<int32_t>((size_t)E);
  g.perm_out = cv.take<int32_t>((size_t)E);
  int32_t* tmp_in = cv.take<int32_t>((size_t)E);
  int32_t* tmp_out = cv.take<int32_t>((size_t)E);
  const size_t nb = (size_t)ceil_div(N + 1, kScanItems);
  int32_t* bs_in = cv.take<int32_t>(nb);
  int32_t* bs_out = cv.take<int32_t>(nb);
  ZeroJobs z;
  z.p[0] = cursors, z.n[0] = 2 * N;
  z.p[1] = in_ptr, z.n[1] = N + 1;
  z.p[2] = out_ptr, z.n[2] = N + 1;
  hipLaunchKernelGGL(k_zero_words, dim3((int)std::min<int64_t>(ceil_div(2 * N, 1024), 1024), 3), dim3(256), 0, st, z);
  g.jobs.j[0] = GroupJob{ei, 1, in_ptr, g.perm_in, cursors, tmp_in, bs_in};           // by destination
  g.jobs.j[1] = GroupJob{ei + E, 1, out_ptr, g.perm_out, cursors + N, tmp_out, bs_out};  // by source
  return PGNN_OK;
}


// (Measured and dropped, round 3: the whole chem build, and pgnn_group_by_key's generic path, as ONE workgroup of 1 024 threads
// with counters, packed edges and 16-bit edge ids in LDS -- bit-identical, one launch instead of six.  A 256-molecule batch took
// 59 us that way against ~30 us for the six launches: ~450 LDS operations per thread with random bank conflicts on one CU's LDS
// cost more than five launch gaps, batched global loads made no difference; 45 us against 28 + 10 us for the masked-bond
// endpoints of a bio batch.  profiles/r03/small_graph_ab.txt.)

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

int pgnn_abi_version(void) { return PGNN_ABI_VERSION; }
void pgnn_reload_env(void) {
  pgnn::g_env_any = pgnn::scan_env_for_knobs();
  ++pgnn::g_env_generation;
}
const char* pgnn_last_error(void) { return pgnn::g_err; }

size_t pgnn_graph_workspace_bytes(int64_t N, int64_t E) {
  return align_up(2 * (size_t)N * 4, 256) + 4 * align_up((size_t)E * 4, 256) +
         2 * align_up((size_t)ceil_div(N + 1, kScanItems) * 4, 256) + 256;
}

int pgnn_chem_graph_build(const int64_t* ei, const int64_t* ea, int64_t E, int64_t N, int gcn,
                          int32_t* in_ptr, int32_t* in_src, uint8_t* in_code, int32_t* out_ptr,
                          int32_t* out_dst, float* dinv, float* cfeat, int32_t* status, void* ws,
                          size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(N > 0 && E >= 0 && N < (1ll << 31) && E < (1ll << 31), "bad graph size N=%lld E=%lld",
               (long long)N, (long long)E);
  hipStream_t st = (hipStream_t)stream;
  GraphWs g;
  int rc = prepare_graph_jobs(ei, E, N, in_ptr, out_ptr, ws, ws_bytes, st, g);
  if (rc) return rc;
  rc = run_group(g.jobs, 2, E, N, status, st);
  if (rc) return rc;
  const int nbk = (int)ceil_div(N, 256);
  hipLaunchKernelGGL(k_chem_payload, dim3(nbk), dim3(256), 0, st, ei, ea, E, N, gcn, in_ptr,
                     g.perm_in, in_src, in_code, out_ptr, g.perm_out, out_dst, dinv, cfeat, status);
  return check_launch("chem_graph_build");
}

int pgnn_bio_graph_build(const int64_t* ei, const float* ea, int64_t E, int64_t N, int gcn,
                         int32_t* in_ptr, int32_t* in_src, int32_t* out_ptr, int32_t* out_dst,
                         float* dinv, float* cfeat, int32_t* status, void* ws, size_t ws_bytes,
                         pgnn_stream stream) {
  PGNN_REQUIRE(N > 0 && E >= 0 && N < (1ll << 31) && E < (1ll << 31), "bad graph size N=%lld E=%lld",
               (long long)N, (long long)E);
  hipStream_t st = (hipStream_t)stream;
  GraphWs g;
  int rc = prepare_graph_jobs(ei, E, N, in_ptr, out_ptr, ws, ws_bytes, st, g);
  if (rc) return rc;
  rc = run_group(g.jobs, 2, E, N, status, st);
  if (rc) return rc;
  if (env_knob("PGNN_BIO_PAYLOAD16", 1) != 0) {
    hipLaunchKernelGGL(k_bio_payload16, dim3((int)ceil_div(N * 16, 256)), dim3(256), 0, st, ei, ea, E, N, gcn, in_ptr,
                       g.perm_in, in_src, out_ptr, g.perm_out, out_dst, dinv, cfeat);
    return check_launch("bio_graph_build");
  }
  const int nbk = (int)ceil_div(N, 256);
  hipLaunchKernelGGL(k_bio_payload, dim3(nbk), dim3(256), 0, st, ei, ea, E, N, gcn, in_ptr,
                     g.perm_in, in_src, out_ptr, g.perm_out, out_dst, dinv, cfeat);
  return check_launch("bio_graph_build");
}

size_t pgnn_group_workspace_bytes(int64_t n_keys, int64_t n_items) {
  return (use_partition(n_keys, n_items) ? partition_ws_bytes(n_keys, n_items) : group_ws_bytes(n_keys, n_items)) + 256;
}

int pgnn_group_by_key(const int64_t* key, int64_t key_stride, int64_t n_items, int64_t n_keys,
                      int32_t* ptr, int32_t* perm, int32_t* status, void* ws, size_t ws_bytes,
                      pgnn_stream stream) {
  PGNN_REQUIRE(n_keys > 0 && n_items >= 0 && key_stride >= 1 && n_items < (1ll << 31), "bad group_by_key sizes");
  if (ws_bytes < pgnn_group_workspace_bytes(n_keys, n_items)) {
    set_error("group_by_key workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (use_partition(n_keys, n_items)) return run_partition(key, key_stride, n_items, n_keys, ptr, perm, status, ws, st);
  Carver cv(ws);
  GroupJobs jobs;
  int32_t* cursor = cv.take<int32_t>((size_t)n_keys);
  int32_t* tmp = cv.take<int32_t>((size_t)n_items);
  int32_t* bsum = cv.take<int32_t>((size_t)ceil_div(n_keys + 1, kScanItems));
  PGNN_HIP(hipMemsetAsync(cursor, 0, (size_t)n_keys * 4, st));
  PGNN_HIP(hipMemsetAsync(ptr, 0, (size_t)(n_keys + 1) * 4, st));
  jobs.j[0] = GroupJob{key, key_stride, ptr, perm, cursor, tmp, bsum};
  jobs.j[1] = jobs.j[0];
  return run_group(jobs, 1, n_items, n_keys, status, st);
}

int pgnn_group_by_key_pair(const int64_t* key_a, const int64_t* key_b, int64_t key_stride, int64_t n_items,
                           int64_t n_a, int64_t n_b, int32_t* ptr, int32_t* perm, int32_t* status, void* ws,
                           size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(n_a > 0 && n_b > 0 && n_items >= 0 && key_stride >= 1 && n_items < (1ll << 31), "bad group_by_key_pair sizes");
  PGNN_REQUIRE(use_partition(n_a * n_b, n_items), "group_by_key_pair: the product of the key ranges must stay within %d", kPartMaxKeys);
  if (ws_bytes < pgnn_group_workspace_bytes(n_a * n_b, n_items)) {
    set_error("group_by_key_pair workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  return run_partition(key_a, key_stride, n_items, n_a * n_b, ptr, perm, status, ws, (hipStream_t)stream, key_b, (int)n_b);
}

}  // extern "C"
