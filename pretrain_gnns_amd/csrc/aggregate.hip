// Aggregation kernels of the message-passing hot path (gfx950, wave64).
//
// Work decomposition shared by every kernel here: ONE WAVE OWNS ONE NODE ROW AT A TIME.  A row of
// D fp32 features is D/4 float4 chunks; lane l owns chunks l, l+64, ... (R = ceil(D/256) of them,
// D = 300 -> 64 + 11 lanes), so every neighbour-row read and every result-row write is a fully
// coalesced 16 B/lane access, partial sums live in registers, and the only scattered accesses are
// whole 1200-byte rows.  Neighbour indices of a row are fetched by the lanes in one coalesced load
// and broadcast with v_readlane, which makes the row base address wave-uniform (SGPR base + lane
// offset).  Waves are persistent and take contiguous node chunks; the block->chunk map is
// XCD-aware so that the rows a molecule's atoms gather from each other stay in one XCD's L2.
// All reductions are sequential in a fixed order: results are bitwise reproducible, and the chem
// aggregation reproduces the reference's CPU scatter_add order exactly.
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
  __device__ __forceinline__ WaveSched() {
    const int b = xcd_remap(blockIdx.x, gridDim.x);
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
            int nodes_per_chunk) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float T[];  // [18][dim] when TABLE
  if (TABLE) {
    for (int q = threadIdx.x; q < kNumCodes * dim; q += kBlock) {
      const int c = q / dim, d = q - c * dim;
      T[q] = emb1[(c / 3) * dim + d] + emb2[(c % 3) * dim + d];
    }
    __syncthreads();
  }
  const WaveSched ws;
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

inline int pick_grid(int64_t n, int waves_per_node_chunk, int blocks_per_cu, int* nodes_per_chunk) {
  // spread small inputs over many waves, give big inputs 16-node chunks
  const int64_t max_waves = (int64_t)kNumCU * blocks_per_cu * kWavesPerBlock;
  int64_t npc = ceil_div(n, max_waves);
  npc = std::min<int64_t>(std::max<int64_t>(npc, 1), 16);
  *nodes_per_chunk = (int)npc;
  const int64_t chunks = ceil_div(n, npc);
  const int64_t blocks = std::min<int64_t>(ceil_div(chunks, kWavesPerBlock), (int64_t)kNumCU * blocks_per_cu);
  (void)waves_per_node_chunk;
  return (int)std::max<int64_t>(blocks, 1);
}

template <bool TABLE, bool WEIGHT>
int launch_aggregate(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr,
                     const uint8_t* code, const float* emb1, const float* emb2, const float* dinv,
                     float* out, int64_t ldo, int64_t n, int64_t dim, hipStream_t st) {
  const int R = (int)ceil_div(dim / 4, kWave);
  const size_t lds = TABLE ? (size_t)kNumCodes * dim * sizeof(float) : 0;
  int npc;
  const int grid = pick_grid(n, 1, TABLE ? 6 : 8, &npc);
#define PGNN_LAUNCH_AGG(RR)                                                                         \
  allow_big_lds((const void*)k_aggregate<RR, TABLE, WEIGHT>, lds);                                  \
  hipLaunchKernelGGL((k_aggregate<RR, TABLE, WEIGHT>), dim3(grid), dim3(kBlock), lds, st, x, ldx,  \
                     ptr, nbr, code, emb1, emb2, dinv, out, ldo, (int)n, (int)dim, npc)
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
              float* __restrict__ out, int64_t ldo, int n, int dim, int accumulate) {
  extern __shared__ __align__(16) float T[];  // [kc][dim]
  for (int q = threadIdx.x; q < kc * dim; q += kBlock) T[q] = table[(q / dim) * ldt + (q % dim)];
  __syncthreads();
  const WaveSched ws;
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
  for (int i = i0; i < i1; ++i) {
    const float myc = lane < KC ? cfeat[(int64_t)i * KC + lane] : 0.f;
    Row<R> gv;
    row_load<R>(gv, g + (int64_t)i * ldg, lane, d4);
#pragma unroll
    for (int t = 0; t < KC; ++t) {
      const float c = __int_as_float(bcast_i32(__float_as_int(myc), t));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = lane + r * kWave;
        if (r + 1 < R || ch < d4) {
          acc[t].v[r].x = fmaf(c, gv.v[r].x, acc[t].v[r].x);
          acc[t].v[r].y = fmaf(c, gv.v[r].y, acc[t].v[r].y);
          acc[t].v[r].z = fmaf(c, gv.v[r].z, acc[t].v[r].z);
          acc[t].v[r].w = fmaf(c, gv.v[r].w, acc[t].v[r].w);
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
                    float* __restrict__ gtable, int64_t ldgt) {
  const int sl = threadIdx.x & 15;
  const int qq = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int q = min(qq, kc * dim - 1);
  const double s = slice_sum16(partial + q, (size_t)kc * dim, nblocks, sl);
  if (sl == 0 && qq < kc * dim) gtable[(int64_t)(q / dim) * ldgt + (q % dim)] = (float)s;
}

inline int rowfeat_bwd_blocks(int64_t n) {
  return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 64), 1), 2 * kNumCU);
}

// ---------------------------------------------------------------------------------------------
// input embedding: out[i] = t1[idx[i,0]] + t2[idx[i,1]]
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(kBlock)
k_embed_fwd(const int64_t* __restrict__ idx, int64_t stride, const float* __restrict__ t1, int rows1,
            const float* __restrict__ t2, int rows2, float* __restrict__ out, int64_t ldo, int n,
            int dim, int32_t* status) {
#pragma clang fp contract(off)
  const WaveSched ws;
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
constexpr int kSegChunk = 64;

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segsum_chunks(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                const int32_t* __restrict__ perm, int n_items, int n_seg, int mean,
                float* __restrict__ out, int64_t ldo, float* __restrict__ partial, int dim) {
  const int lane = lane_id(), d4 = dim >> 2;
  const int chunk = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int p0 = chunk * kSegChunk;
  if (p0 >= n_items) return;
  const int cnt = min(kSegChunk, n_items - p0);
  const int item = lane < cnt ? (perm ? perm[p0 + lane] : p0 + lane) : 0;
  // segment of the chunk's first position: largest s with ptr[s] <= p0 (binary search, uniform)
  int lo = 0, hi = n_seg;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= p0) lo = mid; else hi = mid;
  }
  int seg = lo;
  int seg_end = ptr[seg + 1];
  while (seg_end <= p0) { ++seg; seg_end = ptr[seg + 1]; }  // skip empty segments
  int seg_beg = ptr[seg];
  Row<R> acc;
  row_zero<R>(acc);
  int j = 0;
  while (j < cnt) {
    const int run_end = min(cnt, seg_end - p0);  // positions [j, run_end) belong to `seg`
    for (; j < run_end; ++j) {
      Row<R> v;
      row_load<R>(v, x + (int64_t)bcast_i32(item, j) * ldx, lane, d4);
#pragma unroll
      for (int r = 0; r < R; ++r) acc.v[r] = f4_add(acc.v[r], v.v[r]);
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
      do { ++seg; seg_beg = seg_end; seg_end = ptr[seg + 1]; } while (seg_end <= seg_beg);
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
    for (int c = c0; c <= c1; ++c) {
      const int slot = (c * kSegChunk > s) ? 0 : 1;
      Row<R> v;
      row_load<R>(v, partial + ((size_t)c * 2 + slot) * dim, lane, d4);
#pragma unroll
      for (int r = 0; r < R; ++r) acc.v[r] = f4_add(acc.v[r], v.v[r]);
    }
    if (mean) {
      const float sc = 1.f / (float)(e - s);
#pragma unroll
      for (int r = 0; r < R; ++r) acc.v[r] = f4_scale(acc.v[r], sc);
    }
  }
  row_store<R>(acc, out + (int64_t)seg * ldo, lane, d4);
}

template <int R>
__global__ void __launch_bounds__(kBlock)
k_segment_broadcast(const float* __restrict__ g, int64_t ldg, const int64_t* __restrict__ key,
                    const int32_t* __restrict__ ptr, int mean, float* __restrict__ gx, int64_t ldgx,
                    int n, int dim) {
  const WaveSched ws;
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
  return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n_rows, kWavesPerBlock), 1), (int64_t)kNumCU * 8);
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

int pgnn_neighbor_sum(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr,
                      const float* dinv, float* out, int64_t ldo, int64_t n, int64_t dim,
                      pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldx % 4 == 0 && ldo % 4 == 0, "bad neighbor_sum arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dinv) return launch_aggregate<false, true>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, dinv, out, ldo, n, dim, st);
  return launch_aggregate<false, false>(x, ldx, ptr, nbr, nullptr, nullptr, nullptr, dinv, out, ldo, n, dim, st);
}

int pgnn_rowfeat_matmul_fwd(const float* cfeat, int64_t kc, const float* table, int64_t ldt, float* out,
                            int64_t ldo, int64_t n, int64_t dim, int accumulate, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && kc > 0 && kc <= 16 && ldo % 4 == 0, "bad rowfeat_matmul_fwd arguments");
  const int R = (int)ceil_div(dim / 4, kWave);
  const size_t lds = (size_t)kc * dim * sizeof(float);
  const int grid = stream_grid(n);
  PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_rowfeat_fwd<RR>), dim3(grid), dim3(kBlock), lds,
                                        (hipStream_t)stream, cfeat, (int)kc, table, ldt, out, ldo,
                                        (int)n, (int)dim, accumulate));
  return check_launch("rowfeat_matmul_fwd");
}

size_t pgnn_rowfeat_matmul_bwd_workspace_bytes(int64_t n, int64_t kc, int64_t dim) {
  return (size_t)rowfeat_bwd_blocks(n) * kc * dim * sizeof(float) + 256;
}

int pgnn_rowfeat_matmul_bwd(const float* cfeat, int64_t kc, const float* g, int64_t ldg, float* gtable,
                            int64_t ldgt, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                            pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && (kc == 9 || kc == 10) && ldg % 4 == 0, "rowfeat_matmul_bwd supports kc in {9,10}");
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
  if (kc == 9) {
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
                     partial, nb, (int)kc, (int)dim, gtable, ldgt);
  return check_launch("rowfeat_matmul_bwd");
}

int pgnn_embed_fwd(const int64_t* idx, int64_t idx_stride, const float* table1, int64_t rows1,
                   const float* table2, int64_t rows2, float* out, int64_t ldo, int64_t n, int64_t dim,
                   int32_t* status, pgnn_stream stream) {
  if (int rc = check_dim(dim)) return rc;
  PGNN_REQUIRE(n > 0 && ldo % 4 == 0 && idx_stride >= (table2 ? 2 : 1), "bad embed_fwd arguments");
  const int R = (int)ceil_div(dim / 4, kWave);
  const int grid = stream_grid(n);
  PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_embed_fwd<RR>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                                        idx, idx_stride, table1, (int)rows1, table2, (int)rows2, out, ldo,
                                        (int)n, (int)dim, status));
  return check_launch("embed_fwd");
}

size_t pgnn_segment_sum_workspace_bytes(int64_t n_items, int64_t n_segments, int64_t dim) {
  (void)n_segments;
  return (size_t)ceil_div(std::max<int64_t>(n_items, 1), kSegChunk) * 2 * dim * sizeof(float) + 256;
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
    PGNN_DISPATCH_R(R, hipLaunchKernelGGL((k_segsum_chunks<RR>), dim3((int)ceil_div(nchunks, kWavesPerBlock)),
                                          dim3(kBlock), 0, st, x, ldx, ptr, perm, (int)n_items, (int)n_segments,
                                          mean, out, ldo, partial, (int)dim));
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
                                        (int)dim));
  return check_launch("segment_broadcast");
}

}  // extern "C"
