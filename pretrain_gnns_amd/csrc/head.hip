// The prediction head of the masking pre-training step, fused (chem/pretrain_masking.py:52-57, bio/pretrain_masking.py):
//     pred = linear_pred(node_rep[masked_indices]);  loss = CrossEntropyLoss()(pred.double(), label);
//     acc  = (argmax(pred, 1) == label).sum() / len(pred)
// The reference spends ~25 tiny launches (gather, addmm, cast, soft-max, nll, arg-max, compare, their backward twins) on a
// few hundred rows; at 256 graphs the host cannot enqueue them as fast as the GPU retires them and the GPU idles between
// the two halves of the GNN.  Here: one launch forward, three (+ a memset) backward.
//
// Arithmetic follows the reference's dtypes: logits and the linear layer's gradients in fp32, soft-max / loss / their
// gradient in float64, the gradient cast back to fp32 where `.double()` sits in the autograd graph.  All sums run in a
// fixed order (bitwise reproducible).
#include "common.h"

namespace pgnn {
namespace {

constexpr int kHeadThreads = 128;   // one class per thread, up to 128 classes (119 atom types, 4 bond types, ...)
constexpr int kHeadMaxDim = 2048;
constexpr int kHeadRows = 4;        // selected rows per block: a weight value is loaded once and used for 4 rows (8 rows: half
                                    // the blocks, one per two CUs at 1007 rows -- slower)
constexpr int kHeadClassGroup = 8;  // classes per block of the weight-gradient kernel
constexpr int kHeadSlices = 4;      // threads per (class, row block) in the forward kernel: each walks a quarter of k (a thread's chain
                                    // of 300 dependent FMAs per row was the kernel's critical path), partials folded in slice order

// rows [4 blk, 4 blk + 4): logits[r, :] = h[idx[r], :] . W^T + b, then each row's float64 log-soft-max terms.
// The LAST block to finish (device counter, self-resetting) folds the rows in order: loss = mean_r nll_r, correct = sum_r.
__global__ void __launch_bounds__(kHeadThreads * kHeadSlices) k_head_fwd(const float* __restrict__ h, int64_t ldh, const int64_t* __restrict__ idx,
                                                           int m, const float* __restrict__ w, const float* __restrict__ b,
                                                           const int64_t* __restrict__ label, int64_t label_stride, int classes, int dim,
                                                           int64_t n_rows, float* __restrict__ logits, double* __restrict__ row_nll,
                                                           int* __restrict__ row_hit, double* __restrict__ loss,
                                                           int64_t* __restrict__ correct, double* __restrict__ metrics,
                                                           double* __restrict__ accum, unsigned* __restrict__ counter,
                                                           int* __restrict__ status) {
  extern __shared__ __align__(16) float hrows[];  // [kHeadRows][dim]
  __shared__ double red[kHeadThreads];
  __shared__ int redi[kHeadThreads];
  __shared__ float part[kHeadSlices][kHeadRows][kHeadThreads];
  __shared__ float zhi[kHeadRows][64];  // logits of classes 64 .. 127, handed to wave 0
  __shared__ bool last;
  const int r0 = blockIdx.x * kHeadRows, tid = threadIdx.x, c = tid & (kHeadThreads - 1), sl = tid / kHeadThreads;
  const int nr = min(kHeadRows, m - r0);
  for (int q = tid; q < kHeadRows * dim; q += kHeadThreads * kHeadSlices) {
    const int i = q / dim, k = q - i * dim;
    float v = 0.f;
    if (i < nr) {
      const int64_t node = idx[r0 + i];
      if (node >= 0 && node < n_rows) v = h[node * ldh + k];
      else if (k == 0) atomicAdd(status, 1);
    }
    hrows[q] = v;
  }
  __syncthreads();
  // (staging W through LDS in [classes][32 k] tiles to coalesce its loads measured SLOWER, 71 vs 49 us at 1007 rows: the
  // scalar LDS reads cost more than the strided global float4 loads, which hit L1 after the first touch of a line)
  {
    float acc[kHeadRows];
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) acc[i] = 0.f;
    if (c < classes) {
      const float4* wr = reinterpret_cast<const float4*>(w + (int64_t)c * dim);
#pragma unroll 4
      for (int k4 = sl; k4 < dim / 4; k4 += kHeadSlices) {  // slice sl takes every fourth float4 of its class row
        const float4 wv = wr[k4];
#pragma unroll
        for (int i = 0; i < kHeadRows; ++i) {
          const float4 hv = *reinterpret_cast<const float4*>(hrows + i * dim + 4 * k4);
          acc[i] = fmaf(hv.x, wv.x, acc[i]);
          acc[i] = fmaf(hv.y, wv.y, acc[i]);
          acc[i] = fmaf(hv.z, wv.z, acc[i]);
          acc[i] = fmaf(hv.w, wv.w, acc[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) part[sl][i][c] = acc[i];
  }
  __syncthreads();
  float z[kHeadRows];
#pragma unroll
  for (int i = 0; i < kHeadRows; ++i) z[i] = -INFINITY;
  if (sl == 0 && c < classes) {
    const float bias = b ? b[c] : 0.f;
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i)
      if (i < nr) {
        z[i] = ((part[0][i][c] + part[1][i][c]) + (part[2][i][c] + part[3][i][c])) + bias;
        logits[(int64_t)(r0 + i) * classes + c] = z[i];
      }
  }
  if (sl != 0) return;  // the soft-max below is the first 128 threads' (whole waves: the barriers that follow count them only)
  // Row maximum with its FIRST index (torch.max's tie rule) and the sum of exp in float64, all rows of the block at once by
  // wave 0: classes 64.. come over from wave 1 through LDS (the s = 64 step of the 128-wide tree this replaces), the steps
  // s = 32 .. 1 are lane shuffles with the tree's pairing -- the same operations on the same operands, bit for bit, without its
  // fourteen barriers per row (4 rows x 18 barriers were a third of the launch).
  if (c >= 64) {
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) zhi[i][c - 64] = z[i];
  }
  __syncthreads();
  double blk_nll = 0.0;
  int blk_hit = 0;
  if (c < 64) {
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) {
      if (i >= nr) continue;
      const double zlo = (double)z[i], zup = (double)zhi[i][c];
      double a = zlo;
      int ai = c;
      if (zup > a) {  // (an equal value at index c + 64 loses to index c)
        a = zup;
        ai = c + 64;
      }
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) {
        const double o = __shfl_down(a, sh);
        const int oi = __shfl_down(ai, sh);
        if (o > a || (o == a && oi < ai)) {
          a = o;
          ai = oi;
        }
      }
      const double zmax = __shfl(a, 0);
      const int arg = __shfl(ai, 0);
      double v = (c < classes ? exp(zlo - zmax) : 0.0) + (c + 64 < classes ? exp(zup - zmax) : 0.0);
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_down(v, sh);
      if (c == 0) {
        const int64_t y = label[(int64_t)(r0 + i) * label_stride];
        const bool yok = y >= 0 && y < classes;
        if (!yok) atomicAdd(status, 1);
        const double zy = yok ? (double)logits[(int64_t)(r0 + i) * classes + y] : 0.0;  // written by thread y of this block, before the barrier
        blk_nll += log(v) + zmax - zy;  // rows of a block in order, blocks in order below: the same sums every run
        blk_hit += (yok && arg == (int)y) ? 1 : 0;
      }
    }
  }
  if (c == 0) {  // one partial per block (read by the last block: agent-scope, see common.h)
    publish(row_nll + blockIdx.x, blk_nll);
    publish(row_hit + blockIdx.x, blk_hit);
    last = arrive_last(counter);
  }
  __syncthreads();
  if (!last) return;
  double s = 0.0;
  int hits = 0;
  for (int q = c; q < (int)gridDim.x; q += kHeadThreads) {  // per-thread strided partials, then a fixed tree: the same order every run
    s += fetch_published(row_nll + q);
    hits += fetch_published(row_hit + q);
  }
  red[c] = s;
  redi[c] = hits;
  __syncthreads();
  for (int t = kHeadThreads / 2; t > 0; t >>= 1) {
    if (c < t) {
      red[c] += red[c + t];
      redi[c] += redi[c + t];
    }
    __syncthreads();
  }
  if (c == 0) {
    *loss = red[0] / (double)m;
    *correct = redi[0];
    if (metrics) {  // the two numbers the train loop reads back, side by side: one device-to-host copy
      metrics[0] = red[0] / (double)m;
      metrics[1] = (double)redi[0];
    }
    if (accum) {  // the epoch sums the reference keeps on the host (loss_accum += loss; acc_node_accum += correct / n), left on
                  // the device so the train loop needs no per-step read-back: same float64 operations in the same order
      accum[0] += red[0] / (double)m;
      accum[1] += (double)redi[0] / (double)m;
      accum[3] += 1.0;
    }
  }
}

// rows [4 blk, 4 blk + 4): dl[r, c] = float((softmax64(logits[r])[c] - [c == y]) * gloss / m), then the rows of
// d node_rep: dnode[idx[r], :] = sum_c dl[r, c] W[c, :]   (dnode is zero elsewhere; idx must not repeat)
__global__ void __launch_bounds__(kHeadThreads * kHeadSlices) k_head_bwd_rows(const float* __restrict__ logits, const int64_t* __restrict__ idx, int m,
                                                                const float* __restrict__ w, const int64_t* __restrict__ label,
                                                                int64_t label_stride, const double* __restrict__ gloss, int classes, int dim,
                                                                int64_t n_rows, float* __restrict__ dl, float* __restrict__ dnode, int64_t ldd) {
  __shared__ float zhi[kHeadRows][64];
  __shared__ float dls[kHeadRows][kHeadThreads];
  const int r0 = blockIdx.x * kHeadRows, tid = threadIdx.x, c = tid;  // the soft-max is wave 0's work
  const int nr = min(kHeadRows, m - r0);
  const double g = *gloss / (double)m;
  // the float64 soft-max of the forward, recomputed the same way (see k_head_fwd): classes 64 .. 127 reach wave 0 through LDS,
  // the rest of the 128-wide max / sum trees are lane shuffles with the trees' pairing
  float zl[kHeadRows];
#pragma unroll
  for (int i = 0; i < kHeadRows; ++i)
    zl[i] = (tid < kHeadThreads && i < nr && c < classes) ? logits[(int64_t)(r0 + i) * classes + c] : -INFINITY;
  if (tid >= 64 && tid < kHeadThreads) {
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) zhi[i][c - 64] = zl[i];
  }
  __syncthreads();
  if (tid < 64) {
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) {
      float dlo = 0.f, dup = 0.f;
      if (i < nr) {  // (uniform)
        const double zlo = (double)zl[i], zup = (double)zhi[i][c];
        double zmax = fmax(zlo, zup);
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) zmax = fmax(zmax, __shfl_down(zmax, sh));
        zmax = __shfl(zmax, 0);
        const double elo = c < classes ? exp(zlo - zmax) : 0.0, eup = c + 64 < classes ? exp(zup - zmax) : 0.0;
        double tot = elo + eup;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) tot += __shfl_down(tot, sh);
        tot = __shfl(tot, 0);
        const int y = (int)label[(int64_t)(r0 + i) * label_stride];
        if (c < classes) {
          dlo = (float)((elo / tot - (c == y ? 1.0 : 0.0)) * g);
          dl[(int64_t)(r0 + i) * classes + c] = dlo;
        }
        if (c + 64 < classes) {
          dup = (float)((eup / tot - (c + 64 == y ? 1.0 : 0.0)) * g);
          dl[(int64_t)(r0 + i) * classes + c + 64] = dup;
        }
      }
      dls[i][c] = dlo;
      dls[i][c + 64] = dup;
    }
  }
  __syncthreads();
  int64_t node[kHeadRows];
#pragma unroll
  for (int i = 0; i < kHeadRows; ++i) {
    node[i] = i < nr ? idx[r0 + i] : -1;
    if (node[i] >= n_rows) node[i] = -1;
  }
  for (int k = tid; k < dim; k += kHeadThreads * kHeadSlices) {  // one column per thread at dim <= 512
    float acc[kHeadRows];
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int q = 0; q < classes; ++q) {  // (coalesced across the block: consecutive threads, consecutive k; eight loads in flight)
      const float wv = w[(int64_t)q * dim + k];
#pragma unroll
      for (int i = 0; i < kHeadRows; ++i) acc[i] = fmaf(dls[i][q], wv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < kHeadRows; ++i)
      if (node[i] >= 0) dnode[node[i] * ldd + k] = acc[i];
  }
}

// block (class group g, row chunk q): partial[q][c][:] = sum_{r in chunk} dl[r, c] h[idx[r], :] for the 8 classes of the group
// (+ the chunk's share of db); rows in order inside a chunk, chunks folded in order by k_head_fold
__global__ void __launch_bounds__(256) k_head_bwd_weight(const float* __restrict__ dl, const float* __restrict__ h, int64_t ldh,
                                                         const int64_t* __restrict__ idx, int m, int chunk, int classes, int dim,
                                                         int64_t n_rows, float* __restrict__ partial) {
  const int c0 = blockIdx.x * kHeadClassGroup, q = blockIdx.y;
  const int rbeg = q * chunk, rend = min(m, rbeg + chunk);
  const int ncl = min(kHeadClassGroup, classes - c0);
  float* out = partial + (int64_t)q * classes * (dim + 1);
  // the chunk's row ids and gradient values first (LDS, 64 rows at a time): the row loop below then has nothing but
  // independent, coalesced loads of h in it and the compiler keeps several in flight
  __shared__ int64_t nodes[64];
  __shared__ float dls[64][kHeadClassGroup];
  float acc[2][kHeadClassGroup];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < kHeadClassGroup; ++j) acc[u][j] = 0.f;
  for (int r0 = rbeg; r0 < rend; r0 += 64) {
    const int nr = min(64, rend - r0);
    __syncthreads();
    for (int t = threadIdx.x; t < nr * (1 + kHeadClassGroup); t += blockDim.x) {
      const int r = t / (1 + kHeadClassGroup), j = t - r * (1 + kHeadClassGroup);
      if (j == 0) {
        const int64_t node = idx[r0 + r];
        nodes[r] = (node >= 0 && node < n_rows) ? node : -1;
      } else {
        dls[r][j - 1] = j - 1 < ncl ? dl[(int64_t)(r0 + r) * classes + c0 + j - 1] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // column k = threadIdx.x + 256 u; column `dim` carries the bias gradient (h := 1)
      const int k = threadIdx.x + 256 * u;
      if (k > dim) continue;
#pragma unroll 4
      for (int r = 0; r < nr; ++r) {
        const float hv = k == dim ? 1.f : (nodes[r] >= 0 ? h[nodes[r] * ldh + k] : 0.f);
#pragma unroll
        for (int j = 0; j < kHeadClassGroup; ++j) acc[u][j] = fmaf(dls[r][j], hv, acc[u][j]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = threadIdx.x + 256 * u;
    if (k > dim) continue;
#pragma unroll
    for (int j = 0; j < kHeadClassGroup; ++j)
      if (j < ncl) out[(int64_t)(c0 + j) * (dim + 1) + k] = acc[u][j];
  }
  for (int k = threadIdx.x + 512; k < dim + 1; k += blockDim.x) {  // dim > 511: the plain loop
    float a2[kHeadClassGroup];
#pragma unroll
    for (int j = 0; j < kHeadClassGroup; ++j) a2[j] = 0.f;
    for (int r = rbeg; r < rend; ++r) {
      const int64_t node = idx[r];
      const float hv = k == dim ? 1.f : ((node >= 0 && node < n_rows) ? h[node * ldh + k] : 0.f);
#pragma unroll
      for (int j = 0; j < kHeadClassGroup; ++j)
        if (j < ncl) a2[j] = fmaf(dl[(int64_t)r * classes + c0 + j], hv, a2[j]);
    }
#pragma unroll
    for (int j = 0; j < kHeadClassGroup; ++j)
      if (j < ncl) out[(int64_t)(c0 + j) * (dim + 1) + k] = a2[j];
  }
}
__global__ void __launch_bounds__(256) k_head_fold(const float* __restrict__ partial, int nchunk, int classes, int dim, float* __restrict__ dw,
                                                   float* __restrict__ db) {
  const int64_t total = (int64_t)classes * (dim + 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int q = 0; q < nchunk; ++q) acc += partial[(int64_t)q * total + t];
    const int c = (int)(t / (dim + 1)), k = (int)(t - (int64_t)c * (dim + 1));
    if (k < dim) dw[(int64_t)c * dim + k] = acc;
    else if (db) db[c] = acc;
  }
}

inline int head_chunk(int64_t m) { return (int)std::max<int64_t>(32, ceil_div(m, 128)); }  // at most 128 row chunks

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

size_t pgnn_masked_head_workspace_bytes(int64_t m, int64_t classes, int64_t dim) {
  // row_nll [m] f64, row_hit [m] i32, counter, then (backward) dl [m, classes] f32 and the weight-gradient partials
  const int64_t nchunk = ceil_div(m, head_chunk(m));
  return align_up((size_t)m * 8, 256) + align_up((size_t)m * 4, 256) + 256 + align_up((size_t)m * classes * 4, 256) +
         align_up((size_t)nchunk * classes * (dim + 1) * 4, 256) + 256;
}

int pgnn_masked_head_fwd(const float* h, int64_t ldh, int64_t n_rows, const int64_t* idx, int64_t m, const float* w, const float* b,
                         const int64_t* label, int64_t label_stride, int64_t classes, int64_t dim, float* logits, double* loss,
                         int64_t* correct, double* metrics, double* accum, int32_t* status, uint32_t* counter, void* ws,
                         size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && classes > 0 && classes <= kHeadThreads && dim > 0 && dim % 4 == 0 && dim <= kHeadMaxDim && ldh % 4 == 0,
               "masked_head: 1..%d classes, dim and ldh multiples of 4, dim up to %d", kHeadThreads, kHeadMaxDim);
  if (ws_bytes < pgnn_masked_head_workspace_bytes(m, classes, dim)) {
    set_error("masked_head workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  double* row_nll = cv.take<double>((size_t)m);
  int* row_hit = cv.take<int>((size_t)m);
  cv.take<unsigned>(64);
  PGNN_REQUIRE(counter != nullptr, "masked_head: counter (PGNN_TICKET_WORDS zeroed uint32 that the call leaves zeroed) is required");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)kHeadRows * dim * sizeof(float);
  allow_big_lds((const void*)k_head_fwd, lds);
  hipLaunchKernelGGL(k_head_fwd, dim3((int)ceil_div(m, kHeadRows)), dim3(kHeadThreads * kHeadSlices), lds, st, h, ldh, idx, (int)m, w, b, label,
                     label_stride, (int)classes, (int)dim, n_rows, logits, row_nll, row_hit, loss, correct, metrics, accum, counter, status);
  return check_launch("masked_head_fwd");
}

int pgnn_masked_head_bwd(const float* h, int64_t ldh, int64_t n_rows, const int64_t* idx, int64_t m, const float* w,
                         const int64_t* label, int64_t label_stride, const float* logits, const double* gloss, int64_t classes,
                         int64_t dim, float* dnode, int64_t ldd, float* dw, float* db, void* ws, size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && classes > 0 && classes <= kHeadThreads && dim > 0 && dim % 4 == 0 && dim <= kHeadMaxDim && ldd >= dim,
               "masked_head: 1..%d classes, dim a multiple of 4 up to %d", kHeadThreads, kHeadMaxDim);
  if (ws_bytes < pgnn_masked_head_workspace_bytes(m, classes, dim)) {
    set_error("masked_head workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  cv.take<double>((size_t)m);
  cv.take<int>((size_t)m);
  cv.take<unsigned>(64);
  float* dl = cv.take<float>((size_t)m * classes);
  const int chunk = head_chunk(m), nchunk = (int)ceil_div(m, chunk);
  float* partial = cv.take<float>((size_t)nchunk * classes * (dim + 1));
  hipStream_t st = (hipStream_t)stream;
  PGNN_HIP(hipMemsetAsync(dnode, 0, (size_t)n_rows * ldd * sizeof(float), st));
  hipLaunchKernelGGL(k_head_bwd_rows, dim3((int)ceil_div(m, kHeadRows)), dim3(kHeadThreads * kHeadSlices), 0, st, logits, idx, (int)m, w, label,
                     label_stride, gloss, (int)classes, (int)dim, n_rows, dl, dnode, ldd);
  hipLaunchKernelGGL(k_head_bwd_weight, dim3((int)ceil_div(classes, kHeadClassGroup), nchunk), dim3(256), 0, st, dl, h, ldh, idx, (int)m,
                     chunk, (int)classes, (int)dim, n_rows, partial);
  hipLaunchKernelGGL(k_head_fold, dim3((int)std::min<int64_t>(ceil_div(classes * (dim + 1), 256), 1024)), dim3(256), 0, st, partial, nchunk,
                     (int)classes, (int)dim, dw, db);
  return check_launch("masked_head_bwd");
}

}  // extern "C"
