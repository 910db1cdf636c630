// The edge-prediction head of the masking pre-training steps (bio/pretrain_masking.py:45-58; chem/pretrain_masking.py:60-66):
//     edge_rep = node_rep[u] + node_rep[v];  pred = linear(edge_rep);  loss = CrossEntropyLoss()(pred, label);
//     acc = (argmax(pred, 1) == label).sum() / len(pred)
// A 256-graph bio batch masks ~57 000 edges: the reference materialises two [57k, 300] gathers (2 x 68 MB), their sum, a
// GEMM onto 7 classes, and in the backward a [57k, 300] gradient that is sorted and scattered back into ~41k node rows.
// The linear layer has at most 8 classes, so everything edge-sized can be 8 wide instead of 300 wide:
//     forward   P = node_rep . W^T                      [N, C]   one pass over node_rep
//               pred[r] = P[u_r] + P[v_r] + b           [m, C]   (the reference rounds h[u] + h[v] first: same value to fp32 rounding)
//     backward  dl[r] = (softmax(pred[r]) - onehot) g/m [m, C]
//               S[n]  = sum of dl[r] over the masked edges incident to n (fixed order: pgnn_group_by_key)   [N, C]
//               d node_rep = S . W   (pgnn_rowfeat_matmul_fwd),   dW = S^T . node_rep   (pgnn_rowfeat_matmul_bwd),   db = sum_r dl[r]
// because sum_r dl[r,c] (h[u_r] + h[v_r]) regrouped by node is sum_n S[n,c] h[n].  Two launches forward, ten backward (four of them the
// grouping), ~0.1 GB of traffic instead of ~0.6.  All sums in a fixed order.
#include "common.h"

namespace pgnn {
namespace {

constexpr int kEdgeMaxC = 8;
constexpr int kEdgeBlock = 256;

// P[n, c] = sum_k h[n, k] w[c, k]: one wave per node row (four rows' loads in flight), w in registers, 64-lane butterflies
template <int R>
__global__ void __launch_bounds__(256) k_node_logits(const float* __restrict__ h, int64_t ldh, int n, const float* __restrict__ w,
                                                     int classes, int dim, float* __restrict__ P) {
  const int lane = lane_id(), d4 = dim >> 2;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  float4 wr[kEdgeMaxC][R];
#pragma unroll
  for (int c = 0; c < kEdgeMaxC; ++c)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ch = lane + r * kWave;
      wr[c][r] = (c < classes && ch < d4) ? reinterpret_cast<const float4*>(w + (int64_t)c * dim)[ch] : f4_zero();
    }
  const int per = (n + nw - 1) / nw;
  const int i0 = gw * per, i1 = min(n, i0 + per);
  constexpr int U = 4;
  for (int i = i0; i < i1; i += U) {
    float4 hv[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = min(i + u, i1 - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = min(lane + r * kWave, d4 - 1);
        hv[u][r] = reinterpret_cast<const float4*>(h + (int64_t)ii * ldh)[ch];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u >= i1) break;
      float mine = 0.f;
#pragma unroll
      for (int c = 0; c < kEdgeMaxC; ++c) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {  // (a clamped duplicate of the last float4 meets a zero weight)
          s = fmaf(hv[u][r].x, wr[c][r].x, s);
          s = fmaf(hv[u][r].y, wr[c][r].y, s);
          s = fmaf(hv[u][r].z, wr[c][r].z, s);
          s = fmaf(hv[u][r].w, wr[c][r].w, s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == c) mine = s;
      }
      if (lane < classes) P[(int64_t)(i + u) * classes + lane] = mine;
    }
  }
}

struct EdgeLabel {
  const int64_t* label;  // [m] with a stride, or
  int64_t label_stride;
  const float* onehot;   // [m, cols] rows whose FIRST maximum is the label (torch.argmax's tie rule)
  int64_t ld_onehot;
  int cols;
};

__device__ __forceinline__ int edge_label(const EdgeLabel& L, int r) {
  if (L.label) return (int)L.label[(int64_t)r * L.label_stride];
  int arg = 0;
  float best = L.onehot[(int64_t)r * L.ld_onehot];
  for (int c = 1; c < L.cols; ++c) {
    const float v = L.onehot[(int64_t)r * L.ld_onehot + c];
    if (v > best) {
      best = v;
      arg = c;
    }
  }
  return arg;
}

// soft-max pieces of one row in the precision the reference's loss runs in (float: bio; double: chem's pred.double())
template <typename T>
__device__ __forceinline__ void row_softmax(const float (&z)[kEdgeMaxC], int classes, T& zmax, T& sum, int& arg) {
  zmax = (T)z[0];
  arg = 0;
#pragma unroll
  for (int c = 1; c < kEdgeMaxC; ++c)
    if (c < classes && (T)z[c] > zmax) {
      zmax = (T)z[c];
      arg = c;
    }
  sum = (T)0;
#pragma unroll
  for (int c = 0; c < kEdgeMaxC; ++c)
    if (c < classes) sum += (T)exp((T)z[c] - zmax);
}

// one thread per masked edge: logits, its nll and hit; the LAST block folds the rows in a fixed order (strided partials + tree)
__global__ void __launch_bounds__(kEdgeBlock) k_edge_ce(const float* __restrict__ P, int n_nodes, const int64_t* __restrict__ ends, int m,
                                                        const float* __restrict__ b, EdgeLabel L, int classes, int f64,
                                                        float* __restrict__ logits, double* __restrict__ row_nll, int* __restrict__ row_hit,
                                                        double* __restrict__ loss64, float* __restrict__ loss32, int64_t* __restrict__ correct,
                                                        double* __restrict__ metrics, double* __restrict__ accum, int accum_slot,
                                                        int accum_step, unsigned* __restrict__ counter, int* __restrict__ status) {
  __shared__ double red[kEdgeBlock];
  __shared__ int redi[kEdgeBlock];
  __shared__ bool last;
  const int r = blockIdx.x * kEdgeBlock + threadIdx.x;
  double my_nll = 0.0;
  int my_hit = 0;
  if (r < m) {
    int64_t u = ends[r], v = ends[(int64_t)m + r];
    if (u < 0 || u >= n_nodes || v < 0 || v >= n_nodes) {
      atomicAdd(status, 1);
      u = v = 0;
    }
    float z[kEdgeMaxC];
#pragma unroll
    for (int c = 0; c < kEdgeMaxC; ++c)
      z[c] = c < classes ? (P[u * classes + c] + P[v * classes + c]) + (b ? b[c] : 0.f) : -INFINITY;
    for (int c = 0; c < classes; ++c) logits[(int64_t)r * classes + c] = z[c];
    int y = edge_label(L, r);
    if (y < 0 || y >= classes) {
      atomicAdd(status, 1);
      y = 0;
    }
    float zy = z[0];
#pragma unroll
    for (int c = 1; c < kEdgeMaxC; ++c)
      if (c == y) zy = z[c];
    int arg;
    double nll;
    if (f64) {
      double zmax, sum;
      row_softmax<double>(z, classes, zmax, sum, arg);
      nll = log(sum) + zmax - (double)zy;
    } else {
      float zmax, sum;
      row_softmax<float>(z, classes, zmax, sum, arg);
      nll = (double)((logf(sum) + zmax) - zy);
    }
    my_nll = nll;
    my_hit = arg == y ? 1 : 0;
  }
  // the block's rows by a fixed tree, one partial per block (read by the last block: agent-scope, see common.h), the blocks'
  // partials by strided sums + the same tree: the same order every run
  red[threadIdx.x] = my_nll;
  redi[threadIdx.x] = my_hit;
  __syncthreads();
  for (int t = kEdgeBlock / 2; t > 0; t >>= 1) {
    if ((int)threadIdx.x < t) {
      red[threadIdx.x] += red[threadIdx.x + t];
      redi[threadIdx.x] += redi[threadIdx.x + t];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    publish(row_nll + blockIdx.x, red[0]);
    publish(row_hit + blockIdx.x, redi[0]);
    last = arrive_last(counter);
  }
  __syncthreads();
  if (!last) return;
  double s = 0.0;
  int hits = 0;
  for (int q = threadIdx.x; q < (int)gridDim.x; q += kEdgeBlock) {
    s += fetch_published(row_nll + q);
    hits += fetch_published(row_hit + q);
  }
  __syncthreads();
  red[threadIdx.x] = s;
  redi[threadIdx.x] = hits;
  __syncthreads();
  for (int t = kEdgeBlock / 2; t > 0; t >>= 1) {
    if ((int)threadIdx.x < t) {
      red[threadIdx.x] += red[threadIdx.x + t];
      redi[threadIdx.x] += redi[threadIdx.x + t];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double l = red[0] / (double)m;
    // The bio loop's F.cross_entropy reduces fp32 row losses in fp32 (bio/pretrain_masking.py:58); here the per-row fp32 losses are
    // summed in float64 in a fixed tree and the MEAN is rounded to fp32 once -- at least as close to the exact mean as any fp32
    // summation order, but not bit-comparable with the unfused torch fallback of readback="inline" (which sums in torch's order):
    // the two read-back modes may differ in the last fp32 digit of the loss and hence of the gradient scale.
    if (!f64) l = (double)(float)l;  // the reference's loss is an fp32 tensor there
    *loss64 = l;
    if (loss32) *loss32 = (float)l;
    *correct = redi[0];
    if (metrics) {
      metrics[0] = l;
      metrics[1] = (double)redi[0];
    }
    if (accum) {
      accum[0] += l;
      accum[accum_slot] += (double)redi[0] / (double)m;
      if (accum_step) accum[3] += 1.0;
    }
  }
}

// dl[r, c] = (softmax(logits[r])[c] - [c == y]) * gloss / m in the loss's precision, cast to fp32; per-block column sums (fixed tree),
// folded by the last block into db
__global__ void __launch_bounds__(kEdgeBlock) k_edge_dl(const float* __restrict__ logits, int m, EdgeLabel L, int classes, int f64,
                                                        const double* __restrict__ gloss64, const float* __restrict__ gloss32,
                                                        float* __restrict__ dl, float* __restrict__ partial, float* __restrict__ db,
                                                        unsigned* __restrict__ counter) {
  __shared__ float red[kEdgeMaxC][kEdgeBlock];
  __shared__ bool last;
  const int r = blockIdx.x * kEdgeBlock + threadIdx.x;
  float d[kEdgeMaxC];
#pragma unroll
  for (int c = 0; c < kEdgeMaxC; ++c) d[c] = 0.f;
  if (r < m) {
    float z[kEdgeMaxC];
#pragma unroll
    for (int c = 0; c < kEdgeMaxC; ++c) z[c] = c < classes ? logits[(int64_t)r * classes + c] : -INFINITY;
    int y = edge_label(L, r);
    if (y < 0 || y >= classes) y = 0;
    int arg;
    if (f64) {
      double zmax, sum;
      row_softmax<double>(z, classes, zmax, sum, arg);
      const double g = (gloss64 ? *gloss64 : (double)*gloss32) / (double)m;
#pragma unroll
      for (int c = 0; c < kEdgeMaxC; ++c)
        if (c < classes) d[c] = (float)((exp((double)z[c] - zmax) / sum - (c == y ? 1.0 : 0.0)) * g);
    } else {
      float zmax, sum;
      row_softmax<float>(z, classes, zmax, sum, arg);
      const float g = (gloss32 ? *gloss32 : (float)*gloss64) / (float)m;
#pragma unroll
      for (int c = 0; c < kEdgeMaxC; ++c)
        if (c < classes) d[c] = (expf(z[c] - zmax) / sum - (c == y ? 1.f : 0.f)) * g;
    }
    for (int c = 0; c < classes; ++c) dl[(int64_t)r * classes + c] = d[c];
  }
#pragma unroll
  for (int c = 0; c < kEdgeMaxC; ++c) red[c][threadIdx.x] = d[c];
  __syncthreads();
  for (int t = kEdgeBlock / 2; t > 0; t >>= 1) {
    if ((int)threadIdx.x < t) {
#pragma unroll
      for (int c = 0; c < kEdgeMaxC; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + t];
    }
    __syncthreads();
  }
  if (threadIdx.x < kEdgeMaxC) publish(partial + (size_t)blockIdx.x * kEdgeMaxC + threadIdx.x, red[threadIdx.x][0]);
  publish_commit();
  __syncthreads();
  if (threadIdx.x == 0) last = arrive_last(counter);
  __syncthreads();
  if (!last) return;
  // thread t: column t & 7, blocks t >> 3, t >> 3 + 32, ... in order; then a fixed tree over the 32 slices of each column
  const int c = threadIdx.x & (kEdgeMaxC - 1);
  float s = 0.f;
  for (int q = threadIdx.x >> 3; q < (int)gridDim.x; q += kEdgeBlock / kEdgeMaxC) s += fetch_published(partial + (size_t)q * kEdgeMaxC + c);
  red[0][threadIdx.x] = s;
  __syncthreads();
  for (int t = kEdgeBlock / 2; t >= kEdgeMaxC; t >>= 1) {
    if ((int)threadIdx.x < t) red[0][threadIdx.x] += red[0][threadIdx.x + t];
    __syncthreads();
  }
  if ((int)threadIdx.x < classes && db) db[threadIdx.x] = red[0][threadIdx.x];
}

// S[n, c] = sum over the endpoint items grouped under node n (perm order = (node, item id): u-endpoints by edge, then v-endpoints)
__global__ void __launch_bounds__(kEdgeBlock) k_node_gsum(const float* __restrict__ dl, int m, const int32_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ perm, int n_nodes, int classes,
                                                          float* __restrict__ S) {
  const int t = blockIdx.x * kEdgeBlock + threadIdx.x;
  const int n = t >> 3, c = t & 7;
  if (n >= n_nodes || c >= classes) return;
  float s = 0.f;
  const int p0 = ptr[n], p1 = ptr[n + 1];
  for (int p = p0; p < p1; ++p) {
    int e = perm[p];
    if (e >= m) e -= m;
    s += dl[(int64_t)e * classes + c];
  }
  S[(int64_t)n * classes + c] = s;
}

struct EdgeWs {
  float* P;
  double* row_nll;
  int* row_hit;
  float* dl;
  float* partial;
  float* S;
  int32_t* ptr;
  int32_t* perm;
  int32_t* gstatus;
  void* group_ws;
  size_t group_bytes;
  void* rf_ws;
  size_t rf_bytes;
  size_t total;
};

inline EdgeWs carve_edge_ws(void* ws, int64_t n_nodes, int64_t m, int64_t classes, int64_t dim) {
  Carver cv(ws);
  EdgeWs e;
  e.P = cv.take<float>((size_t)n_nodes * classes);
  e.row_nll = cv.take<double>((size_t)m);
  e.row_hit = cv.take<int>((size_t)m);
  e.dl = cv.take<float>((size_t)m * classes);
  e.partial = cv.take<float>((size_t)ceil_div(m, kEdgeBlock) * kEdgeMaxC);
  e.S = cv.take<float>((size_t)n_nodes * classes);
  e.ptr = cv.take<int32_t>((size_t)n_nodes + 1);
  e.perm = cv.take<int32_t>((size_t)2 * m);
  e.gstatus = cv.take<int32_t>(64);
  e.group_bytes = pgnn_group_workspace_bytes(n_nodes, 2 * m);
  e.group_ws = cv.take<char>(e.group_bytes);
  e.rf_bytes = pgnn_rowfeat_matmul_bwd_workspace_bytes(n_nodes, classes, dim);
  e.rf_ws = cv.take<char>(e.rf_bytes);
  e.total = cv.used;
  return e;
}

inline int check_edge_args(int64_t n_nodes, int64_t m, int64_t classes, int64_t dim, int64_t ldh) {
  PGNN_REQUIRE(n_nodes > 0 && m > 0 && 2 * m < (1ll << 31) && (classes == 4 || classes == 7) && dim > 0 && dim % 4 == 0 && dim <= 1024 &&
                   ldh % 4 == 0,
               "edge_head: 4 or 7 classes (the reference's bond / PPI edge-type heads), dim a multiple of 4 up to 1024");
  return PGNN_OK;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

size_t pgnn_edge_head_workspace_bytes(int64_t n_nodes, int64_t m, int64_t classes, int64_t dim) {
  return carve_edge_ws(nullptr, n_nodes, m, classes, dim).total + 256;
}

int pgnn_edge_head_fwd(const float* h, int64_t ldh, int64_t n_nodes, const int64_t* ends, int64_t m, const float* w, const float* b,
                       const int64_t* label, int64_t label_stride, const float* onehot, int64_t ld_onehot, int64_t onehot_cols,
                       int64_t classes, int64_t dim, int loss_float64, float* logits, double* loss64, float* loss32, int64_t* correct, double* metrics,
                       double* accum, int accum_slot, int accum_step, int32_t* status, uint32_t* counter, void* ws, size_t ws_bytes,
                       pgnn_stream stream) {
  if (int rc = check_edge_args(n_nodes, m, classes, dim, ldh)) return rc;
  PGNN_REQUIRE((label != nullptr) != (onehot != nullptr) && (!onehot || (onehot_cols >= 1 && ld_onehot >= onehot_cols)), "edge_head: exactly one of label / onehot [m, onehot_cols]");
  PGNN_REQUIRE(counter && status && loss64 && correct && accum_slot >= 1 && accum_slot <= 2, "edge_head: counter, status, loss64, correct are required; accum_slot 1 or 2");
  if (ws_bytes < pgnn_edge_head_workspace_bytes(n_nodes, m, classes, dim)) {
    set_error("edge_head workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  const EdgeWs e = carve_edge_ws(ws, n_nodes, m, classes, dim);
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)std::min<int64_t>(ceil_div(n_nodes, 16), (int64_t)num_cu() * 8);
  if (dim <= 256)
    hipLaunchKernelGGL((k_node_logits<1>), dim3(grid), dim3(256), 0, st, h, ldh, (int)n_nodes, w, (int)classes, (int)dim, e.P);
  else if (dim <= 512)
    hipLaunchKernelGGL((k_node_logits<2>), dim3(grid), dim3(256), 0, st, h, ldh, (int)n_nodes, w, (int)classes, (int)dim, e.P);
  else
    hipLaunchKernelGGL((k_node_logits<4>), dim3(grid), dim3(256), 0, st, h, ldh, (int)n_nodes, w, (int)classes, (int)dim, e.P);
  const EdgeLabel L{label, label_stride, onehot, ld_onehot, (int)onehot_cols};
  hipLaunchKernelGGL(k_edge_ce, dim3((int)ceil_div(m, kEdgeBlock)), dim3(kEdgeBlock), 0, st, e.P, (int)n_nodes, ends, (int)m, b, L,
                     (int)classes, loss_float64, logits, e.row_nll, e.row_hit, loss64, loss32, correct, metrics, accum, accum_slot,
                     accum_step, counter, status);
  return check_launch("edge_head_fwd");
}

int pgnn_edge_head_bwd(const float* h, int64_t ldh, int64_t n_nodes, const int64_t* ends, int64_t m, const float* w,
                       const int64_t* label, int64_t label_stride, const float* onehot, int64_t ld_onehot, int64_t onehot_cols, const float* logits,
                       const double* gloss64, const float* gloss32, int64_t classes, int64_t dim, int loss_float64, float* dnode,
                       int64_t ldd, float* dw, float* db, uint32_t* counter, void* ws, size_t ws_bytes, pgnn_stream stream) {
  if (int rc = check_edge_args(n_nodes, m, classes, dim, ldh)) return rc;
  PGNN_REQUIRE((label != nullptr) != (onehot != nullptr) && (gloss64 != nullptr) != (gloss32 != nullptr) && counter && ldd % 4 == 0 && ldd >= dim,
               "edge_head_bwd: exactly one of label / onehot and of gloss64 / gloss32; counter required");
  if (ws_bytes < pgnn_edge_head_workspace_bytes(n_nodes, m, classes, dim)) {
    set_error("edge_head workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  const EdgeWs e = carve_edge_ws(ws, n_nodes, m, classes, dim);
  hipStream_t st = (hipStream_t)stream;
  const EdgeLabel L{label, label_stride, onehot, ld_onehot, (int)onehot_cols};
  hipLaunchKernelGGL(k_edge_dl, dim3((int)ceil_div(m, kEdgeBlock)), dim3(kEdgeBlock), 0, st, logits, (int)m, L, (int)classes, loss_float64,
                     gloss64, gloss32, e.dl, e.partial, db, counter);
  // (endpoints were range-checked by the forward; the grouping's own count of bad keys is added to a scratch word nobody reads,
  // so nobody zeroes it either)
  if (int rc = pgnn_group_by_key(ends, 1, 2 * m, n_nodes, e.ptr, e.perm, e.gstatus, e.group_ws, e.group_bytes, stream)) return rc;
  hipLaunchKernelGGL(k_node_gsum, dim3((int)ceil_div(n_nodes * 8, kEdgeBlock)), dim3(kEdgeBlock), 0, st, e.dl, (int)m, e.ptr, e.perm,
                     (int)n_nodes, (int)classes, e.S);
  if (int rc = pgnn_rowfeat_matmul_fwd(e.S, classes, w, dim, dnode, ldd, n_nodes, dim, 0, stream)) return rc;
  if (int rc = pgnn_rowfeat_matmul_bwd(e.S, classes, h, ldh, dw, dim, n_nodes, dim, e.rf_ws, e.rf_bytes, stream)) return rc;
  return check_launch("edge_head_bwd");
}

}  // extern "C"
