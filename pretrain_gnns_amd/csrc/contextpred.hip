// The negative-sampling dot-product loss of context prediction, fused (chem/pretrain_contextpred.py:54-67,86-97; the same body in
// bio/pretrain_contextpred.py:49-62,81-92), cbow mode with mean context pooling -- the reference's defaults:
//     substruct_rep = model_substruct(...)[center_substruct_idx]                                   [B, D]
//     context_rep   = global_mean_pool(model_context(...)[overlap_context_substruct_idx], batch_overlapped_context)   [B, D]
//     pred_pos[i]       = sum_d substruct_rep[i] * context_rep[i]
//     pred_neg[k B + i] = sum_d substruct_rep[i] * context_rep[(i + k + 1) % B]                    (cycle_index, k < neg_samples)
//     loss_pos = BCEWithLogits(pred_pos.double(), 1), loss_neg = BCEWithLogits(pred_neg.double(), 0)   (means)
//     acc      = 0.5 * (sum(pred_pos > 0) / len(pred_pos) + sum(pred_neg < 0) / len(pred_neg))
// In torch that is ~50 launches of a few hundred elements each (index, scatter-mean, cat, repeat, mul, sum, cast, log-sigmoid,
// mean, compare and their backward twins): at 256 molecules the host enqueues them slower than the GPU retires them and the
// GPU idles for ~0.5 ms between the two networks' forward and backward passes (profiles/r03/ctx_step_timeline.txt).
// Here: two launches forward (pooled context rows; scores + float64 loss terms, folded by the last block to finish), one
// backward (d node embeddings: zero everywhere but the centre rows / the overlap rows, written by the same launch).
// Dtypes as the reference: dot products and their gradients in fp32, the loss and its derivative in float64, cast back to
// fp32 where `.double()` sits in the autograd graph.  All sums in a fixed order (bitwise reproducible).
#include "common.h"

namespace pgnn {
namespace {

constexpr int kCtxThreads = 128;
constexpr int kCtxMaxDim = 4 * kCtxThreads;  // one float4 per thread
constexpr int kCtxMaxNeg = 8;

// first position p in [0, n) with seg[p] >= g (seg sorted ascending)
__device__ __forceinline__ int lower_bound_i64(const int64_t* __restrict__ seg, int n, int64_t g) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seg[mid] < g) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// block g: ctx[g] = mean of hc[overlap[p]] over the positions p whose segment id is g (sequential fp32 sum in position order,
// divided by max(count, 1): scatter_mean of torch_scatter 1.1.2); cnt[g] = count
__global__ void __launch_bounds__(kCtxThreads) k_ctx_pool(const float* __restrict__ hc, int64_t ldc, int64_t n_ctx, const int64_t* __restrict__ overlap,
                                                          const int64_t* __restrict__ seg, int n_ov, int dim, float* __restrict__ ctx,
                                                          int* __restrict__ cnt, int* __restrict__ status) {
  __shared__ int range[2];
  const int g = blockIdx.x, t = threadIdx.x;
  if (t < 2) range[t] = lower_bound_i64(seg, n_ov, (int64_t)g + t);
  __syncthreads();
  const int lo = range[0], hi = range[1];
  if (t * 4 >= dim) return;
  float4 q = f4_zero();
  for (int p = lo; p < hi; ++p) {
    int64_t r = overlap[p];
    if (r < 0 || r >= n_ctx) {
      if (t == 0) atomicAdd(status, 1);
      r = 0;
    }
    q = f4_add(q, *reinterpret_cast<const float4*>(hc + r * ldc + 4 * t));
  }
  const float c = (float)max(hi - lo, 1);
  const float4 s = make_float4(q.x / c, q.y / c, q.z / c, q.w / c);  // a true division, as sum / count in the reference
  *reinterpret_cast<float4*>(ctx + (int64_t)g * dim + 4 * t) = s;
  if (t == 0) cnt[g] = hi - lo;
}

__device__ __forceinline__ double bce_logits(double x, double target) {  // torch's formula: (1 - t) x + max(-x, 0) + log(exp(-max) + exp(-x - max))
  const double mx = fmax(-x, 0.0);
  return (1.0 - target) * x + mx + log(exp(-mx) + exp(-x - mx));
}

// block g: scores[g] = s_g . ctx[g], scores[(k + 1) B + g] = s_g . ctx[(g + k + 1) % B] with s_g = hs[center[g]]; the last block
// to finish folds the float64 loss terms and the hit counts in index order into out[4] = (loss_pos, loss_neg, frac_pos, frac_neg)
__global__ void __launch_bounds__(kCtxThreads) k_ctx_scores(const float* __restrict__ hs, int64_t lds_, int64_t n_sub, const int64_t* __restrict__ center,
                                                            const float* __restrict__ ctx, int B, int dim, int neg, float* __restrict__ scores,
                                                            double* __restrict__ out, double* __restrict__ loss, double* __restrict__ accum,
                                                            unsigned* __restrict__ counter, int* __restrict__ status) {
  __shared__ float red[kCtxMaxNeg + 1][kCtxThreads];
  __shared__ double dred[2][kCtxThreads];
  __shared__ int ired[2][kCtxThreads];
  __shared__ bool last;
  const int g = blockIdx.x, t = threadIdx.x;
  int64_t r = center[g];
  if (r < 0 || r >= n_sub) {
    if (t == 0) atomicAdd(status, 1);
    r = 0;
  }
  const bool on = t * 4 < dim;
  const float4 s = on ? *reinterpret_cast<const float4*>(hs + r * lds_ + 4 * t) : f4_zero();
  for (int k = 0; k <= neg; ++k) {
    const int j = k == 0 ? g : (g + k) % B;
    const float4 c = on ? *reinterpret_cast<const float4*>(ctx + (int64_t)j * dim + 4 * t) : f4_zero();
    red[k][t] = (s.x * c.x + s.y * c.y) + (s.z * c.z + s.w * c.w);
  }
  __syncthreads();
  for (int h = kCtxThreads / 2; h > 0; h >>= 1) {  // fixed tree: the same order every run
    if (t < h)
      for (int k = 0; k <= neg; ++k) red[k][t] += red[k][t + h];
    __syncthreads();
  }
  if (t <= neg) publish(scores + (int64_t)t * B + g, red[t][0]);  // (read by the last block: agent-scope, see common.h)
  publish_commit();
  __syncthreads();  // (the score stores of threads 1..neg are acknowledged before this block's arrival)
  if (t == 0) last = arrive_last(counter);
  __syncthreads();
  if (!last) return;
  double lp = 0.0, ln = 0.0;
  int hp = 0, hn = 0;
  for (int q = t; q < B; q += kCtxThreads) {
    const float x = fetch_published(scores + q);
    lp += bce_logits((double)x, 1.0);
    hp += x > 0.f ? 1 : 0;
  }
  for (int q = t; q < neg * B; q += kCtxThreads) {
    const float x = fetch_published(scores + B + q);
    ln += bce_logits((double)x, 0.0);
    hn += x < 0.f ? 1 : 0;
  }
  dred[0][t] = lp; dred[1][t] = ln; ired[0][t] = hp; ired[1][t] = hn;
  __syncthreads();
  for (int h = kCtxThreads / 2; h > 0; h >>= 1) {
    if (t < h) {
      dred[0][t] += dred[0][t + h]; dred[1][t] += dred[1][t + h];
      ired[0][t] += ired[0][t + h]; ired[1][t] += ired[1][t + h];
    }
    __syncthreads();
  }
  if (t == 0) {
    const double loss_pos = dred[0][0] / (double)B, loss_neg = dred[1][0] / (double)(neg * B);
    const double fp = (double)ired[0][0] / (double)B, fn = (double)ired[1][0] / (double)(neg * B);
    out[0] = loss_pos; out[1] = loss_neg; out[2] = fp; out[3] = fn;
    if (loss) *loss = loss_pos + (double)neg * loss_neg;  // what train() back-propagates (:89)
    if (accum) {  // the epoch sums train() keeps on the host (:99-100): balanced loss, accuracy, -, steps
      accum[0] += loss_pos + loss_neg;
      accum[1] += 0.5 * (fp + fn);
      accum[3] += 1.0;
    }
  }
}

// block g: d hs[center[g]] = a_g ctx[g] + sum_k b_{k,g} ctx[(g + k + 1) % B]
//          d ctx[g]        = a_g s_g   + sum_k b_{k,g'} s_{g'},  g' = (g - k - 1) mod B;   d hc[overlap[p]] = d ctx[g] / count_g for p in g's segment
// with a_g = float(g (sigmoid(pos_g) - 1) / B), b_{k,g} = float(g neg sigmoid(neg_{k,g}) / (neg B)), g = d / d (loss_pos + neg loss_neg) --
// the float64 derivative of the two BCE means, cast to fp32 where the reference's `.double()` sits.  dhs / dhc are zero elsewhere (written by the caller).
__global__ void __launch_bounds__(kCtxThreads) k_ctx_bwd(const float* __restrict__ hs, int64_t lds_, int64_t n_sub, const int64_t* __restrict__ center,
                                                         const float* __restrict__ ctx, const int* __restrict__ cnt, const int64_t* __restrict__ overlap,
                                                         const int64_t* __restrict__ seg, int n_ov, int64_t n_ctx, const float* __restrict__ scores,
                                                         const double* __restrict__ gout, int B, int dim, int neg, float* __restrict__ dhs,
                                                         int64_t lddhs, float* __restrict__ dhc, int64_t lddhc) {
  __shared__ int range[2];
  const int g = blockIdx.x, t = threadIdx.x;
  if (t < 2) range[t] = lower_bound_i64(seg, n_ov, (int64_t)g + t);
  __syncthreads();
  if (t * 4 >= dim) return;
  const double gpos = gout[0], gneg = (double)neg * gout[0];  // d (loss_pos + neg loss_neg)
  auto coef_pos = [&](int i) { return (float)(gpos * (1.0 / (1.0 + exp(-(double)scores[i])) - 1.0) / (double)B); };
  auto coef_neg = [&](int k, int i) { return (float)(gneg * (1.0 / (1.0 + exp(-(double)scores[(int64_t)(k + 1) * B + i]))) / (double)(neg * B)); };
  auto srow = [&](int i) {
    int64_t r = center[i];
    if (r < 0 || r >= n_sub) r = 0;
    return *reinterpret_cast<const float4*>(hs + r * lds_ + 4 * t);
  };
  // d substructure row
  {
    float4 d = f4_scale(*reinterpret_cast<const float4*>(ctx + (int64_t)g * dim + 4 * t), coef_pos(g));
    for (int k = 0; k < neg; ++k) {
      const int j = (g + k + 1) % B;
      d = f4_add(d, f4_scale(*reinterpret_cast<const float4*>(ctx + (int64_t)j * dim + 4 * t), coef_neg(k, g)));
    }
    int64_t r = center[g];
    if (r >= 0 && r < n_sub) *reinterpret_cast<float4*>(dhs + r * lddhs + 4 * t) = d;
  }
  // d pooled context row, spread over the segment's rows
  {
    float4 d = f4_scale(srow(g), coef_pos(g));
    for (int k = 0; k < neg; ++k) {
      const int i = ((g - k - 1) % B + B) % B;
      d = f4_add(d, f4_scale(srow(i), coef_neg(k, i)));
    }
    const float c = (float)max(cnt[g], 1);
    d = make_float4(d.x / c, d.y / c, d.z / c, d.w / c);
    for (int p = range[0]; p < range[1]; ++p) {
      const int64_t r = overlap[p];
      if (r >= 0 && r < n_ctx) *reinterpret_cast<float4*>(dhc + r * lddhc + 4 * t) = d;
    }
  }
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

size_t pgnn_contextpred_loss_workspace_bytes(int64_t graphs, int64_t dim, int64_t neg_samples) {
  // ctx [B, dim] f32, cnt [B] i32, scores [(1 + neg) B] f32
  return align_up((size_t)graphs * dim * 4, 256) + align_up((size_t)graphs * 4, 256) + align_up((size_t)(1 + neg_samples) * graphs * 4, 256) + 256;
}

int pgnn_contextpred_loss_fwd(const float* hs, int64_t ldhs, int64_t n_sub, const int64_t* center, const float* hc, int64_t ldhc, int64_t n_ctx,
                              const int64_t* overlap, const int64_t* seg, int64_t n_overlap, int64_t graphs, int64_t dim, int64_t neg_samples,
                              double* out, double* loss, double* accum, int32_t* status, uint32_t* counter, void* ws, size_t ws_bytes,
                              pgnn_stream stream) {
  PGNN_REQUIRE(graphs > 0 && graphs < (1 << 24) && dim > 0 && dim % 4 == 0 && dim <= kCtxMaxDim && neg_samples >= 1 && neg_samples <= kCtxMaxNeg &&
                   n_overlap >= 0 && n_overlap < (1ll << 31) && ldhs % 4 == 0 && ldhc % 4 == 0 && counter && status && out,
               "contextpred_loss_fwd: bad arguments (dim %% 4 == 0, dim <= %d, 1 <= neg_samples <= %d)", kCtxMaxDim, kCtxMaxNeg);
  if (ws_bytes < pgnn_contextpred_loss_workspace_bytes(graphs, dim, neg_samples)) {
    set_error("contextpred_loss_fwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  float* ctx = cv.take<float>((size_t)graphs * dim);
  int* cnt = cv.take<int>((size_t)graphs);
  float* scores = cv.take<float>((size_t)(1 + neg_samples) * graphs);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ctx_pool, dim3((int)graphs), dim3(kCtxThreads), 0, st, hc, ldhc, n_ctx, overlap, seg, (int)n_overlap, (int)dim, ctx, cnt, status);
  hipLaunchKernelGGL(k_ctx_scores, dim3((int)graphs), dim3(kCtxThreads), 0, st, hs, ldhs, n_sub, center, ctx, (int)graphs, (int)dim, (int)neg_samples,
                     scores, out, loss, accum, counter, status);
  return check_launch("contextpred_loss_fwd");
}

int pgnn_contextpred_loss_bwd(const float* hs, int64_t ldhs, int64_t n_sub, const int64_t* center, int64_t n_ctx, const int64_t* overlap,
                              const int64_t* seg, int64_t n_overlap, int64_t graphs, int64_t dim, int64_t neg_samples, const double* grad_loss,
                              float* dhs, int64_t lddhs, float* dhc, int64_t lddhc, const void* ws, size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(graphs > 0 && dim > 0 && dim % 4 == 0 && dim <= kCtxMaxDim && neg_samples >= 1 && neg_samples <= kCtxMaxNeg && lddhs % 4 == 0 &&
                   lddhc % 4 == 0 && grad_loss && dhs && dhc,
               "contextpred_loss_bwd: bad arguments");
  if (ws_bytes < pgnn_contextpred_loss_workspace_bytes(graphs, dim, neg_samples)) {
    set_error("contextpred_loss_bwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(const_cast<void*>(ws));
  const float* ctx = cv.take<float>((size_t)graphs * dim);
  const int* cnt = cv.take<int>((size_t)graphs);
  const float* scores = cv.take<float>((size_t)(1 + neg_samples) * graphs);
  hipStream_t st = (hipStream_t)stream;
  PGNN_HIP(hipMemsetAsync(dhs, 0, (size_t)n_sub * lddhs * sizeof(float), st));
  PGNN_HIP(hipMemsetAsync(dhc, 0, (size_t)n_ctx * lddhc * sizeof(float), st));
  hipLaunchKernelGGL(k_ctx_bwd, dim3((int)graphs), dim3(kCtxThreads), 0, st, hs, ldhs, n_sub, center, ctx, cnt, overlap, seg, (int)n_overlap, n_ctx, scores,
                     grad_loss, (int)graphs, (int)dim, (int)neg_samples, dhs, lddhs, dhc, lddhc);
  return check_launch("contextpred_loss_bwd");
}

}  // extern "C"
