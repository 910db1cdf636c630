// Layer-level entry points: one C call enqueues every kernel of a chem GIN layer (+ its outer
// BatchNorm), forward or backward.  Host-side composition of the public per-op entry points only --
// no new device code.  Exists because at the reference's batch size (256 graphs ~ 6.8k nodes) the step
// is launch-bound: ~35 kernels per layer round trip, and a Python/ctypes/allocator round trip per
// kernel costs more than the kernel.  (chem/model.py:37-55 + :269-275 under autograd.)
#include <stdlib.h>

#include "bn_fold.h"
#include "common.h"
#include <mutex>

using namespace pgnn;

namespace pgnn {
namespace {
thread_local hipEvent_t t_next_stop_event = nullptr;
// pgnn_stack_bwd_dy_rows: the row support of the NEXT one-call backward's dy on this host thread (consumed or dropped by that call)
struct DyRows {
  const float* dy = nullptr;
  const int64_t* rows = nullptr;
  int64_t count = 0;
};
thread_local DyRows t_dy_rows;
DyRows take_dy_rows(const float* dy) {
  DyRows r = t_dy_rows;
  t_dy_rows = DyRows{};
  if (r.dy != dy || env_knob("PGNN_SPARSE_TOP_GRAD", 1) == 0) r = DyRows{};
  return r;
}
}
void set_next_launch_stop_event(hipEvent_t ev) { t_next_stop_event = ev; }
hipEvent_t take_next_launch_stop_event() {
  hipEvent_t ev = t_next_stop_event;
  t_next_stop_event = nullptr;
  return ev;
}
}  // namespace pgnn

namespace {
// Side stream for the backward's independent branches.  At ~6.8k rows one GEMM only gives each CU ~1.7
// tiles, so the weight-gradient product (needs dz/dhid + saved activations) runs concurrently with the
// data-gradient product that the rest of the chain is waiting for: fork with an event after each
// producer, join once before returning.  Streams/events are created once per device and reused.
constexpr int64_t kSideMaxRows = 32768;
struct Side {
  hipStream_t stream = nullptr;
  hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr};  // [3]: chem stack backward, "embedding grouping done"
  hipEvent_t join = nullptr;
  hipEvent_t lag[2] = {nullptr, nullptr};  // stack backward: "aux finished with buffer set p"
  bool ok = false;
};
Side* side_for_current_device() {
  static Side sides[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Side& s = sides[dev];
  if (!s.ok) {
    // default priority on purpose: a low- (or high-) priority side stream changed nothing in the eager step
    // and made HIP-graph replay of the step 50 % slower (2.8 vs 1.8 ms, measured)
    // (round 4 re-measured it with the caller's stream the longer one: lowest priority 0.994-0.997 against 0.990-0.993 ms, no effect)
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    // (hipEventReleaseToDevice on the fork / lag events: measured level, profiles/r05/fork_via_launch_ab.txt)
    const unsigned fl = hipEventDisableTiming;
    for (auto& e : s.fork)
      if (hipEventCreateWithFlags(&e, fl) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (auto& e : s.lag)
      if (hipEventCreateWithFlags(&e, fl) != hipSuccess) return nullptr;
    s.ok = true;
  }
  return &s;
}
inline bool use_side_stream() { return env_knob("PGNN_SIDE_STREAM", 1) != 0; }
// Is `st` being captured into a HIP graph?  The completion event of a hipExtLaunchKernelGGL dispatch is NOT a capture edge: under
// capture a hipStreamWaitEvent on it orders nothing, and the side stream's weight gradients then read dz / dhid of an earlier replay
// (hipgraph_replay's loss went 0.37 -> 0.99 and stopped being bit-stable once the fork became the product's own
// dispatch).  While capturing, the fork is an ordinary hipEventRecord behind the product.
inline bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return true;  // (cannot tell: take the path that is correct either way)
  }
  return cs != hipStreamCaptureStatusNone;
}
// PGNN_SIDE_MIN_ROWS: rows below which the one-call chem GIN backward stays on the caller's stream.  0 (always fork) is the default: a
// fork / lag / join costs the host ~25 us per layer (four event calls), and whether the overlap pays below ~5 000 rows depends on the
// box -- the context-prediction step (4 1xx + 1 9xx rows) measures 1.30-1.38 ms forked against 1.46 on one stream on most boxes,
// 1.61-1.68 against 1.46-1.47 on the ones that run every small kernel ~20 % slower (profiles/r05/ctx_side_stream_ab.txt)
inline int64_t side_min_rows() { return env_knob("PGNN_SIDE_MIN_ROWS", 0); }

// Gradient milestone of a stack backward (pgnn_stack_bwd_milestone_arm / _wait): once layer `layer` has been enqueued, every
// parameter gradient of layers >= `layer` is behind one of two events -- the caller's stream, the side stream -- so a
// communication stream that waits for both can all-reduce the top of the network while the layers below are still running.
// One armed milestone per DEVICE, not per host thread: torch runs the backward on its autograd thread, the arming and the wait
// happen on the caller's.  A backward that never reaches the layer (other networks, fewer layers) leaves it unrecorded.
struct GradMilestone {
  int layer = -1;
  const void* network = nullptr;  // the armed network: the w1 of its layer `layer` (parameter storage is stable across steps)
  int backwards = 0;              // stack backwards of that network that reached the layer since it was armed
  bool recorded = false;
  hipEvent_t ev[2] = {nullptr, nullptr};
};
std::mutex g_milestone_mutex;
GradMilestone g_milestones[64];
GradMilestone* milestone_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return &g_milestones[dev];
}
// `network`: layers[l].w1 of the backward that calls -- another network's backward (context prediction runs two under one set of
// optimizers) must not record the armed one's milestone.  A SECOND backward of the armed network before the wait
// (gradient accumulation) adds into gradients the first one's events no longer cover: counted, and the wait then refuses.
int milestone_record(int l, const void* network, hipStream_t main, hipStream_t aux) {
  GradMilestone* m = milestone_of_current_device();
  if (!m || m->layer < 0) return PGNN_OK;  // (unlocked peek: the production path without data parallelism never takes the lock)
  std::lock_guard<std::mutex> lock(g_milestone_mutex);
  if (l != m->layer || network != m->network) return PGNN_OK;
  ++m->backwards;
  if (m->recorded) return PGNN_OK;
  for (auto& e : m->ev)
    if (!e) PGNN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  PGNN_HIP(hipEventRecord(m->ev[0], main));
  PGNN_HIP(hipEventRecord(m->ev[1], aux));
  m->recorded = true;
  return PGNN_OK;
}
inline bool milestone_armed() {
  GradMilestone* m = milestone_of_current_device();
  return m && m->layer >= 0;  // (unlocked peek, as in milestone_record)
}
// backward-data on pre-transposed weights (both operands k-contiguous): 0 = never, 2 = always, 1 (default) = from 16 384 rows,
// where the k-contiguous form pulls ahead of pgnn_linear_bwd_data's transpose-read form (44 / 45 us against 54 / 56 at 16 384
// rows, 145 / 135 against 204 / 244 at 65 536); at one 256-graph batch the two are level and the extra launch is not worth it
inline bool use_transposed_weights(int64_t n) {
  const int v = env_knob("PGNN_BWD_TRANSPOSED", 1);
  return v == 2 || (v == 1 && n >= 16384);
}
inline bool per_layer_buffers() { return env_knob("PGNN_STACK_PER_LAYER_BUFFERS", 0) != 0; }
// Products on pre-split weight planes (csrc/linear.hip, k_gemm3w): the weights of the whole stack are split (backward: transposed
// and split) into three bf16 planes by ONE launch per pass; every forward and backward-data product then streams them by DMA.
constexpr int kMaxPlaneLayers = 16;  // pgnn_split_weights takes 32 matrices per launch
inline size_t mlp_planes_bytes(int64_t d_in, int64_t d_hid, int64_t d_out) {  // W1 [d_hid, d_in] + W2 [d_out, d_hid], either orientation
  return std::max(pgnn_weight_planes_bytes(d_hid, d_in), pgnn_weight_planes_bytes(d_in, d_hid)) +
         std::max(pgnn_weight_planes_bytes(d_out, d_hid), pgnn_weight_planes_bytes(d_hid, d_out));
}
inline bool mlp_wp(int64_t n, int64_t d_in, int64_t d_hid, int64_t d_out, int num_layer) {
  return num_layer <= kMaxPlaneLayers && pgnn_linear_wp_preferred(n, d_in, d_hid) && pgnn_linear_wp_preferred(n, d_hid, d_out) &&
         pgnn_linear_wp_preferred(n, d_out, d_hid) && pgnn_linear_wp_preferred(n, d_hid, d_in);
}
// The stacks' products on planes run on TWO fp16 planes + a power-of-two scale per row (csrc/linear.hip, the block at its end;
// round 4: the default) -- PGNN_GEMM_2P=0: three bf16 planes (round 3; bit-identical to the per-layer calls' split-bf16 kernel).
// The split, the forward products and the backward-data products switch together.  x_amax / y_amax: row maxima handed from a
// product's epilogue to the product that consumes its result (two planes only; see pgnn_linear_fwd_2p).
inline bool two_planes() { return env_knob("PGNN_GEMM_2P", 1) != 0; }
inline int stack_fwd_wp(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m, int64_t k,
                        int64_t n, int relu, float* colstat, pgnn_stream stream, const uint32_t* x_amax = nullptr, uint32_t* y_amax = nullptr) {
  return two_planes() ? linear_fwd_wp_2p(x, ldx, wplanes, bias, y, ldy, m, k, n, relu, colstat, (hipStream_t)stream, x_amax, y_amax)
                      : pgnn_linear_fwd_wp(x, ldx, wplanes, bias, y, ldy, m, k, n, relu, colstat, stream);
}
inline int stack_bwd_data_wp(const float* dy, int64_t lddy, const void* wtplanes, const float* relu_out, int64_t ldr, float* dx,
                             int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream, const uint32_t* dy_amax = nullptr,
                             uint32_t* dx_amax = nullptr) {
  return two_planes() ? linear_bwd_data_wp_2p(dy, lddy, wtplanes, relu_out, ldr, dx, lddx, m, k, n, (hipStream_t)stream, dy_amax, dx_amax)
                      : pgnn_linear_bwd_data_wp(dy, lddy, wtplanes, relu_out, ldr, dx, lddx, m, k, n, stream);
}
// Both products of a chem GIN mlp in ONE launch (csrc/mlp_fused.hip: the hidden activation written once, never re-read; the
// activations fetched once): PGNN_MLP_FUSED = 0 never, 1 (default) from kFusedMinRows rows on -- below, a workgroup's 128 rows do not
// amortise the 1.5 MB of weight planes it streams from L2, and the tiled products win --, 2 wherever the shape is covered.
constexpr int64_t kFusedMinRows = 32768;
inline bool mlp_fused(int64_t n, int64_t d_in, int64_t d_hid, int64_t d_out) {
  const int v = env_knob("PGNN_MLP_FUSED", 1);
  return v != 0 && two_planes() && (v >= 2 || n >= kFusedMinRows) && mlp_fused_supported(n, d_in, d_hid, d_out);
}
// the chem GIN stack's bond-table gradients as twelve more columns of its dW1 products (linear.hip, linear_bwd_weight_pair_ext):
// cfeat [n][9] padded to [n][12] once per backward + G = dhid^T cfeat [2 dim][12] per layer (PGNN_BOND_IN_DW=0: a pass over dagg per layer)
inline size_t bond_ws_bytes(int64_t n, int64_t dim, int64_t num_layer) {
  return align_up((size_t)n * 12 * 4, 256) + (size_t)std::min<int64_t>(num_layer, kMaxBondJobs) * align_up((size_t)2 * dim * 12 * 4, 256) + 256;
}
inline size_t amax_words(int64_t n) { return align_up((size_t)n * 4, 256) / 4; }  // one row-maximum vector, in words
// planes of W1 / W2 (transpose = 0) or W1^T / W2^T (1) of every layer: p1[l], p2[l] carved from `base`
// (bump: also increment the layers' num_batches_tracked -- a training-mode forward -- in the same launch;
//  zero_ptr / zero_words: the row-maximum words of the pass, cleared by the same launch -- two planes only)
inline int split_mlp_weights(const pgnn_gin_layer* layers, int num_layer, int64_t d_in, int64_t d_hid, int64_t d_out, int transpose,
                             char* base, void** p1, void** p2, hipStream_t st, bool bump = false, const EncTables* tabs = nullptr,
                             uint32_t* zero_ptr = nullptr, int64_t zero_words = 0) {
  const float* src[2 * kMaxPlaneLayers];
  void* dst[2 * kMaxPlaneLayers];
  int64_t rows[2 * kMaxPlaneLayers], cols[2 * kMaxPlaneLayers];
  int32_t tr[2 * kMaxPlaneLayers];
  const size_t b1 = std::max(pgnn_weight_planes_bytes(d_hid, d_in), pgnn_weight_planes_bytes(d_in, d_hid));
  const size_t per = mlp_planes_bytes(d_in, d_hid, d_out);
  for (int l = 0; l < num_layer; ++l) {
    p1[l] = base + (size_t)l * per;
    p2[l] = base + (size_t)l * per + b1;
    src[2 * l] = layers[l].w1; dst[2 * l] = p1[l]; rows[2 * l] = d_hid; cols[2 * l] = d_in; tr[2 * l] = transpose;
    src[2 * l + 1] = layers[l].w2; dst[2 * l + 1] = p2[l]; rows[2 * l + 1] = d_out; cols[2 * l + 1] = d_hid; tr[2 * l + 1] = transpose;
  }
  int64_t* counters[kMaxPlaneLayers];
  int nb = 0;
  if (bump)
    for (int l = 0; l < num_layer; ++l)
      if (layers[l].num_batches_tracked) counters[nb++] = layers[l].num_batches_tracked;
  return two_planes() ? split_weights_2p(src, dst, rows, cols, tr, 2 * num_layer, counters, nb, st, tabs, zero_ptr, zero_words)
                      : split_weights_bump(src, dst, rows, cols, tr, 2 * num_layer, counters, nb, st, tabs);
}
// the same increment as a launch of its own, for the calls that split nothing (small batches on the fp32-MFMA products)
__global__ void k_bump_counters(long long* c0, long long* c1, long long* c2, long long* c3, long long* c4, long long* c5, long long* c6,
                                long long* c7) {
  long long* c[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
  if (threadIdx.x < 8 && c[threadIdx.x]) *c[threadIdx.x] += 1;
}
inline int bump_batches_tracked(const pgnn_gin_layer* layers, int num_layer, hipStream_t st) {
  long long* c[8];
  int nb = 0;
  for (int l = 0; l < num_layer; ++l) {
    if (!layers[l].num_batches_tracked) continue;
    c[nb++] = reinterpret_cast<long long*>(layers[l].num_batches_tracked);
    if (nb == 8 || l == num_layer - 1) {
      for (int q = nb; q < 8; ++q) c[q] = nullptr;
      hipLaunchKernelGGL(k_bump_counters, dim3(1), dim3(64), 0, st, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
      nb = 0;
    }
  }
  if (nb) {
    for (int q = nb; q < 8; ++q) c[q] = nullptr;
    hipLaunchKernelGGL(k_bump_counters, dim3(1), dim3(64), 0, st, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
  }
  return check_launch("bump_batches_tracked");
}
constexpr int64_t kStatsInGemmMaxRows = 32768;  // pgnn_chem_gin_stack_fwd: BatchNorm statistics from the GEMM epilogue up to here
constexpr int kMaxTransposed = 8;  // layers whose weights pgnn_chem_gin_stack_bwd transposes up front (one 16-job launch)
inline size_t op_ws_bytes(int64_t n, int64_t d) {
  size_t m = pgnn_bn_workspace_bytes(n, d);
  m = std::max(m, pgnn_bn_workspace_bytes(n, 2 * d));
  m = std::max(m, pgnn_linear_bwd_weight_workspace_bytes(n, d, 2 * d) + pgnn_linear_bwd_weight_workspace_bytes(n, 2 * d, d));  // as a pair
  m = std::max(m, pgnn_rowfeat_matmul_bwd_workspace_bytes(n, 9, d));
  return align_up(m, 256);
}
}  // namespace

extern "C" {

size_t pgnn_chem_gin_layer_workspace_bytes(int64_t n, int64_t dim) {
  // 2 x op scratch (main / side stream) + dhid [n, 2*dim] + dagg [n, dim] + dz [n, dim]
  return 2 * op_ws_bytes(n, dim) + align_up((size_t)n * 2 * dim * 4, 256) + 2 * align_up((size_t)n * dim * 4, 256) + 256;
}

int pgnn_chem_gin_layer_fwd(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src,
                            const uint8_t* in_code, const float* emb1, const float* emb2, const float* w1,
                            const float* b1, const float* w2, const float* b2, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                            int training, int relu, float* agg, float* hid, float* z, float* y, float* save_mean,
                            float* save_invstd, float drop_p, uint64_t drop_seed, int64_t n, int64_t dim, void* ws,
                            size_t ws_bytes, pgnn_stream stream) {
  if (ws_bytes < op_ws_bytes(n, dim)) {
    set_error("chem_gin_layer_fwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  int rc;
  if ((rc = pgnn_chem_aggregate_fwd(x, ldx, in_ptr, in_src, in_code, emb1, emb2, nullptr, agg, dim, n, dim, stream))) return rc;
  if ((rc = pgnn_linear_fwd(agg, dim, w1, b1, hid, 2 * dim, n, dim, 2 * dim, 1, stream))) return rc;
  if ((rc = pgnn_linear_fwd(hid, 2 * dim, w2, b2, z, dim, n, 2 * dim, dim, 0, stream))) return rc;
  return pgnn_bn_fwd(z, dim, gamma, beta, running_mean, running_var, momentum, eps, training, relu, y, dim, save_mean,
                     save_invstd, drop_p, drop_seed, n, dim, ws, ws_bytes, stream);
}

int pgnn_chem_gin_layer_bwd(const float* dy, int64_t lddy, const float* agg, const float* hid, const float* z,
                            const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat, const float* w1,
                            const float* w2, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, int training, int relu, float* dx, float* demb /*[9,dim]*/,
                            float* dw1, float* db1, float* dw2, float* db2, float* dgamma, float* dbeta, float drop_p,
                            uint64_t drop_seed, int64_t n, int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream) {
  if (ws_bytes < pgnn_chem_gin_layer_workspace_bytes(n, dim)) {
    set_error("chem_gin_layer_bwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  Carver cv(ws);
  const size_t opb = op_ws_bytes(n, dim);
  char* op = cv.take<char>(opb);
  char* op2 = cv.take<char>(opb);  // scratch of the side-stream ops
  float* dhid = cv.take<float>((size_t)n * 2 * dim);
  float* dagg = cv.take<float>((size_t)n * dim);
  float* dz = cv.take<float>((size_t)n * dim);
  hipStream_t main = (hipStream_t)stream;
  // concurrency only pays while one GEMM cannot fill the chip; at large n the branches just thrash L2
  Side* sd = (use_side_stream() && n <= kSideMaxRows) ? side_for_current_device() : nullptr;
  hipStream_t aux = sd ? sd->stream : main;
  char* aux_ws = sd ? op2 : op;
  auto fork = [&](int i) -> int {  // aux stream continues after everything enqueued on main so far
    if (!sd) return PGNN_OK;
    PGNN_HIP(hipEventRecord(sd->fork[i], main));
    PGNN_HIP(hipStreamWaitEvent(sd->stream, sd->fork[i], 0));
    return PGNN_OK;
  };
  int rc;
  // BatchNorm(+ReLU) backward -> dz
  if ((rc = pgnn_bn_bwd(dy, lddy, z, dim, gamma, beta, save_mean, save_invstd, training, relu, dz, dim, dgamma, dbeta,
                        drop_p, drop_seed, n, dim, op, opb, main))) return rc;
  // second Linear: dW2, db2 (aux) || dhid = (dz . W2) masked by hid > 0 (main)
  if ((rc = fork(0))) return rc;
  if ((rc = pgnn_linear_bwd_weight(dz, dim, hid, 2 * dim, dw2, db2, n, 2 * dim, dim, aux_ws, opb, aux))) return rc;
  if ((rc = pgnn_linear_bwd_data(dz, dim, w2, hid, 2 * dim, dhid, 2 * dim, n, 2 * dim, dim, main))) return rc;
  // first Linear: dW1, db1 (aux) || dagg = dhid . W1 (main)
  if ((rc = fork(1))) return rc;
  if ((rc = pgnn_linear_bwd_weight(dhid, 2 * dim, agg, dim, dw1, db1, n, dim, 2 * dim, aux_ws, opb, aux))) return rc;
  if ((rc = pgnn_linear_bwd_data(dhid, 2 * dim, w1, nullptr, 0, dagg, dim, n, dim, 2 * dim, main))) return rc;
  // aggregation: bond-embedding gradients (aux) || dx on the transposed CSR (main)
  if ((rc = fork(2))) return rc;
  if ((rc = pgnn_rowfeat_matmul_bwd(cfeat, 9, dagg, dim, demb, dim, n, dim, aux_ws, opb, aux))) return rc;
  if (dx && (rc = pgnn_neighbor_sum(dagg, dim, out_ptr, out_dst, nullptr, dx, dim, n, dim, main))) return rc;
  if (sd) {  // join: nothing of this call is in flight on the side stream once main passes this point
    PGNN_HIP(hipEventRecord(sd->join, sd->stream));
    PGNN_HIP(hipStreamWaitEvent(main, sd->join, 0));
  }
  return PGNN_OK;
}

/* ---------------------------------------------------------------------------------------------
 * The whole node-embedding network of chem/model.py:258-277 (JK="last", no dropout) as one call per
 * direction: atom embedding -> num_layer x (GIN conv, BatchNorm, ReLU except after the last layer).
 * Same kernels in the same order as the per-layer calls, hence bit-identical results; what changes
 * is the host side: one Python/ctypes round trip per direction instead of one per layer, and in the
 * backward ONE fork per layer -- the side stream runs a layer's three parameter-gradient products
 * (dW2, dW1, bond tables) back to back while the main stream is already in the next layer's chain.
 * dz/dhid/dagg are double-buffered by layer parity so that overlap is safe; `lag` events stop the
 * main stream from re-using a buffer set before the side stream is done with it.
 * --------------------------------------------------------------------------------------------- */
namespace {
// Both atom-embedding gradients come from ONE grouping by the (type, chirality) pair when the product of the
// table sizes fits the partition kernel (120 x 3 = 360 <= 1024): one stable partition, one segment sum over the
// node gradients, one fold -- instead of two groupings and two passes over the same gradients.
inline bool embed_pair_path(int64_t rows1, int64_t rows2) { return rows1 > 0 && rows2 > 0 && rows1 * rows2 <= 1024; }
inline size_t stack_group_ws(int64_t n, int64_t rows1, int64_t rows2) {
  size_t m = std::max(pgnn_group_workspace_bytes(std::max<int64_t>(rows1, 1), n),
                      pgnn_group_workspace_bytes(std::max<int64_t>(rows2, 1), n));
  if (embed_pair_path(rows1, rows2)) m = std::max(m, pgnn_group_workspace_bytes(rows1 * rows2, n));
  return align_up(m, 256);
}
inline size_t stack_segsum_ws(int64_t n, int64_t dim, int64_t rows1, int64_t rows2) {
  size_t m = std::max(pgnn_segment_sum_workspace_bytes(n, std::max<int64_t>(rows1, 1), dim),
                      pgnn_segment_sum_workspace_bytes(n, std::max<int64_t>(rows2, 1), dim));
  if (embed_pair_path(rows1, rows2)) m = std::max(m, pgnn_segment_sum_workspace_bytes(n, rows1 * rows2, dim));
  return align_up(m, 256);
}
inline size_t stack_keys(int64_t rows1, int64_t rows2) {
  return (size_t)(embed_pair_path(rows1, rows2) ? rows1 * rows2 : std::max(rows1, rows2));
}
inline size_t stack_pair_sums(int64_t dim, int64_t rows1, int64_t rows2) {
  return embed_pair_path(rows1, rows2) ? align_up((size_t)rows1 * rows2 * dim * 4, 256) : 0;
}

// the grouping step of embed_tables_bwd on its own (it depends on the atom columns only)
inline int embed_tables_group(const int64_t* x_idx, int64_t n, int64_t rows1, int64_t rows2, const float* dxemb1, const float* dxemb2,
                              int32_t* gptr0, int32_t* gperm0, int32_t* gptr1, int32_t* gperm1, int32_t* gstatus, char* grp_ws,
                              size_t grp_b, hipStream_t st) {
  if (embed_pair_path(rows1, rows2) && dxemb1 && dxemb2)
    return pgnn_group_by_key_pair(x_idx, x_idx + 1, 2, n, rows1, rows2, gptr0, gperm0, gstatus, grp_ws, grp_b, st);
  const int64_t rows[2] = {rows1, rows2};
  const float* dx[2] = {dxemb1, dxemb2};
  int32_t* ptrs[2] = {gptr0, gptr1};
  int32_t* perms[2] = {gperm0, gperm1};
  for (int c = 0; c < 2; ++c)
    if (dx[c])
      if (int rc = pgnn_group_by_key(x_idx + c, 2, n, rows[c], ptrs[c], perms[c], gstatus, grp_ws, grp_b, st)) return rc;
  return PGNN_OK;
}

// gradients of the two atom embedding tables from the gradient g [n, dim] of their sum (chem/model.py:264)
inline int embed_tables_bwd(const float* g, const int64_t* x_idx, int64_t n, int64_t dim, int64_t rows1, int64_t rows2,
                            float* dxemb1, float* dxemb2, int32_t* gptr0, int32_t* gperm0, int32_t* gptr1,
                            int32_t* gperm1, int32_t* gstatus, char* grp_ws, size_t grp_b, char* seg_ws, size_t seg_b,
                            float* pair_sums, bool grouped, hipStream_t st) {
  int rc;
  if (embed_pair_path(rows1, rows2) && dxemb1 && dxemb2) {
    if (!grouped &&
        (rc = pgnn_group_by_key_pair(x_idx, x_idx + 1, 2, n, rows1, rows2, gptr0, gperm0, gstatus, grp_ws, grp_b, st)))
      return rc;
    if ((rc = pgnn_segment_sum(g, dim, gptr0, gperm0, n, rows1 * rows2, 0, pair_sums, dim, dim, seg_ws, seg_b, st))) return rc;
    return pgnn_pair_fold(pair_sums, rows1, rows2, dxemb1, dim, dxemb2, dim, dim, st);
  }
  const int64_t rows[2] = {rows1, rows2};
  float* dx[2] = {dxemb1, dxemb2};
  int32_t* ptrs[2] = {gptr0, gptr1};
  int32_t* perms[2] = {gperm0, gperm1};
  for (int c = 0; c < 2; ++c) {
    if (!dx[c]) continue;
    if (!grouped && (rc = pgnn_group_by_key(x_idx + c, 2, n, rows[c], ptrs[c], perms[c], gstatus, grp_ws, grp_b, st))) return rc;
    if ((rc = pgnn_segment_sum(g, dim, ptrs[c], perms[c], n, rows[c], 0, dx[c], dim, dim, seg_ws, seg_b, st))) return rc;
  }
  return PGNN_OK;
}
}  // namespace

size_t pgnn_chem_gin_stack_workspace_bytes(int64_t n, int64_t dim, int64_t rows1, int64_t rows2, int64_t num_layer) {
  const size_t nd = align_up((size_t)n * dim * 4, 256);
  // 2 x op scratch + S x (dz, dagg, dx: nd each; dhid: 2 nd) + group-by-key of the two atom columns.
  // S = 2 (ping-pong by layer parity), or one set per layer under the PGNN_STACK_PER_LAYER_BUFFERS=1 A/B knob.
  const size_t sets = (per_layer_buffers() && n <= kSideMaxRows) ? (size_t)std::max<int64_t>(num_layer, 2) : 2;
  // W1^T, W2^T of the transposed-weights backward, or the bf16 planes of every layer's weights (forward: W1, W2; backward: their
  // transposes) -- whichever is larger
  const size_t wt = std::max((size_t)std::min<int64_t>(num_layer, kMaxTransposed) * 2 * align_up((size_t)2 * dim * dim * 4, 256),
                             (size_t)std::min<int64_t>(num_layer, kMaxPlaneLayers) * mlp_planes_bytes(dim, 2 * dim, dim));
  // + one row-maximum vector per layer (two-plane products: the maxima of hid / dhid from the epilogue of the product that writes them)
  return 2 * op_ws_bytes(n, dim) + sets * 5 * nd + wt + 2 * align_up((size_t)n * 4, 256) +
         2 * align_up((stack_keys(rows1, rows2) + 1) * 4, 256) + 256 + stack_group_ws(n, rows1, rows2) +
         stack_segsum_ws(n, dim, rows1, rows2) + stack_pair_sums(dim, rows1, rows2) + 256 +
         ((size_t)std::min<int64_t>(num_layer, kMaxPlaneLayers) + 2) * amax_words(n) * 4 +  // (+ the maxima of agg / dz: one vector each)
         bond_ws_bytes(n, dim, num_layer);  // the bond-table gradients out of the dW1 products: cfeat padded to 12 floats a row, G per layer
}

int pgnn_chem_gin_stack_fwd(const int64_t* x_idx, const float* xemb1, int64_t rows1, const float* xemb2,
                            int64_t rows2, const int32_t* in_ptr, const int32_t* in_src, const uint8_t* in_code,
                            const pgnn_gin_layer* layers, int num_layer, int training, float* h0, float* acts,
                            float* hid, float* stats, int32_t* status, float drop_p, uint64_t drop_seed, int64_t n,
                            int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream) {
  if (num_layer < 1 || !layers) {
    set_error("chem_gin_stack_fwd: no layers");
    return PGNN_ERR_ARG;
  }
  if (ws_bytes < op_ws_bytes(n, dim)) {
    set_error("chem_gin_stack_fwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  int rc;
  if ((rc = pgnn_embed_fwd(x_idx, 2, xemb1, rows1, xemb2, rows2, h0, dim, n, dim, status, stream))) return rc;
  const size_t nd = (size_t)n * dim;
  // Between two layers the BatchNorm(+ReLU) output is not written: the next layer's aggregation applies
  // relu(a*z + b) on read (pgnn_chem_aggregate_bn_fwd).  Needs the statistics only; dropout, wide
  // features and PGNN_FUSE_BN_AGG=0 take the materialising route.
  const bool fuse = drop_p == 0.f && dim <= 320 && env_knob("PGNN_FUSE_BN_AGG", 1) != 0;
  // Training-mode statistics from the epilogue of the product in front (PGNN_BN_STATS_IN_GEMM=0: the separate partial-sum
  // pass).  Up to kStatsInGemmMaxRows rows: beyond, the per-16-row blocks (150 B a row) cost more than the pass they replace.
  // (re-measured with the resident-plane products at 438 792 rows: 32.99 against 32.11 ms per step with the blocks -- still a loss)
  const bool stats_in_gemm = training && n > 1 && n <= kStatsInGemmMaxRows && env_knob("PGNN_BN_STATS_IN_GEMM", 1) != 0;
  // both products of every layer on pre-split weight planes when the caller's workspace has room for them behind the op scratch
  // (pgnn_chem_gin_stack_workspace_bytes does; the per-layer size of older callers does not: they keep the in-kernel split)
  const size_t opb = op_ws_bytes(n, dim);
  void *wp1[kMaxPlaneLayers], *wp2[kMaxPlaneLayers];
  const size_t planes_b = (size_t)num_layer * mlp_planes_bytes(dim, 2 * dim, dim);
  const bool wp = mlp_wp(n, dim, 2 * dim, dim, num_layer) && ws_bytes >= opb + planes_b;
  // two planes: the first product of a layer leaves the row maxima of hid for the second (one vector per layer behind the planes,
  // cleared by the split launch) when the workspace has the room
  uint32_t* hid_amax = nullptr;
  if (wp && two_planes() && ws_bytes >= opb + planes_b + ((size_t)num_layer + 1) * amax_words(n) * 4)
    hid_amax = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + opb + planes_b);
  // ... and the aggregation leaves the row maxima of agg for the first (one vector, rewritten by every layer: plain stores)
  // (PGNN_PRODUCER_AMAX=0: the products take those two kinds of maxima themselves -- the same maxima, the same bits)
  uint32_t* const agg_amax = (hid_amax && env_knob("PGNN_PRODUCER_AMAX", 1) != 0) ? hid_amax + (size_t)num_layer * amax_words(n) : nullptr;
  if (wp && (rc = split_mlp_weights(layers, num_layer, dim, 2 * dim, dim, 0, static_cast<char*>(ws) + opb, wp1, wp2, (hipStream_t)stream,
                                    training != 0, nullptr, hid_amax, hid_amax ? (int64_t)num_layer * (int64_t)amax_words(n) : 0)))
    return rc;
  if (!wp && training && (rc = bump_batches_tracked(layers, num_layer, (hipStream_t)stream))) return rc;
  const bool fused_mlp = wp && mlp_fused(n, dim, 2 * dim, dim);
  for (int l = 0; l < num_layer; ++l) {
    const pgnn_gin_layer& p = layers[l];
    float* a = acts + (size_t)l * 3 * nd;  // agg, z, y
    float* agg = a;
    float* z = a + nd;
    float* y = a + 2 * nd;
    float* hd = hid + (size_t)l * 2 * nd;
    float* st = stats + (size_t)l * 4 * dim;  // mean, invstd, a, b
    const bool last = l == num_layer - 1;
    bool agg_has_amax = false;
    if (l == 0) {
      if (agg_amax) rc = chem_aggregate_fwd_amax(h0, dim, nullptr, 0, in_ptr, in_src, in_code, p.emb1, p.emb2, agg, dim, n, dim, agg_amax, &agg_has_amax, (hipStream_t)stream);
      else rc = pgnn_chem_aggregate_fwd(h0, dim, in_ptr, in_src, in_code, p.emb1, p.emb2, nullptr, agg, dim, n, dim, stream);
    } else if (fuse) {
      const float* zprev = acts + (size_t)(l - 1) * 3 * nd + nd;
      const float* cprev = stats + (size_t)(l - 1) * 4 * dim + 2 * dim;
      if (agg_amax) rc = chem_aggregate_fwd_amax(zprev, dim, cprev, 1, in_ptr, in_src, in_code, p.emb1, p.emb2, agg, dim, n, dim, agg_amax, &agg_has_amax, (hipStream_t)stream);
      else rc = pgnn_chem_aggregate_bn_fwd(zprev, dim, cprev, 1, in_ptr, in_src, in_code, p.emb1, p.emb2, agg, dim, n, dim, stream);
    } else {
      const float* yprev = acts + (size_t)(l - 1) * 3 * nd + 2 * nd;
      if (agg_amax) rc = chem_aggregate_fwd_amax(yprev, dim, nullptr, 0, in_ptr, in_src, in_code, p.emb1, p.emb2, agg, dim, n, dim, agg_amax, &agg_has_amax, (hipStream_t)stream);
      else rc = pgnn_chem_aggregate_fwd(yprev, dim, in_ptr, in_src, in_code, p.emb1, p.emb2, nullptr, agg, dim, n, dim, stream);
    }
    if (rc) return rc;
    uint32_t* ham = hid_amax ? hid_amax + (size_t)l * amax_words(n) : nullptr;
    if (wp && fused_mlp) {
      // one launch for both products; the statistics of z as per-16-row blocks from its epilogue where the unfused path takes them
      // from the second product's (the in-launch fold of bn_fold.h belongs to the tiled kernel: blocks + their merge launch here)
      float* blocks = stats_in_gemm ? static_cast<float*>(ws) : nullptr;
      if ((rc = mlp_fwd_2p_fused(agg, dim, wp1[l], p.b1, wp2[l], p.b2, hd, 2 * dim, z, dim, n, dim, 2 * dim, dim, blocks, (hipStream_t)stream)))
        return rc;
      if (stats_in_gemm) {
        if ((rc = pgnn_bn_stats_fwd_blocks(blocks, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, st, st + dim,
                                           st + 2 * dim, n, dim, stream)))
          return rc;
        if (!(fuse && !last))
          rc = pgnn_bn_apply_fwd(z, dim, st + 2 * dim, !last, y, dim, drop_p, drop_seed + (uint64_t)l, n, dim, stream);
      } else if (fuse && !last) {
        rc = pgnn_bn_stats_fwd(z, dim, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, training, st, st + dim,
                               st + 2 * dim, n, dim, ws, opb, stream);
      } else {
        rc = pgnn_bn_fwd(z, dim, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, training, !last, y, dim, st, st + dim,
                         drop_p, drop_seed + (uint64_t)l, n, dim, ws, opb, stream);
      }
      if (rc) return rc;
      continue;
    }
    if (wp) rc = stack_fwd_wp(agg, dim, wp1[l], p.b1, hd, 2 * dim, n, dim, 2 * dim, 1, nullptr, stream, agg_has_amax ? agg_amax : nullptr, ham);
    else rc = pgnn_linear_fwd(agg, dim, p.w1, p.b1, hd, 2 * dim, n, dim, 2 * dim, 1, stream);
    if (rc) return rc;
    if (stats_in_gemm) {
      // the BatchNorm statistics of z fall out of the second product's epilogue: no pass over z for them, one launch less
      float* blocks = static_cast<float*>(ws);  // ceil(n/16) x 2 x dim floats <= the statistics partials of op_ws_bytes
      // two planes (round 4): the merge of those blocks happens inside the product's launch too (bn_fold.h: tile, group of
      // tiles, column panel) -- no k_bn_stats_final_blocks launch between the product and the next aggregation
      BnFwdFold ff{};
      const bool folded = wp && two_planes() && env_knob("PGNN_BN_STATS_FOLD", 1) != 0 && bn_fwd_fold_scratch(ws, opb, n, dim, &ff);
      if (folded) {
        ff.gamma = p.gamma; ff.beta = p.beta; ff.running_mean = p.running_mean; ff.running_var = p.running_var;
        ff.momentum = p.momentum; ff.eps = p.eps; ff.save_mean = st; ff.save_invstd = st + dim; ff.coef = st + 2 * dim;
        rc = linear_fwd_wp_2p(hd, 2 * dim, wp2[l], p.b2, z, dim, n, 2 * dim, dim, 0, nullptr, (hipStream_t)stream, ham, nullptr, &ff);
      } else if (wp) {
        rc = stack_fwd_wp(hd, 2 * dim, wp2[l], p.b2, z, dim, n, 2 * dim, dim, 0, blocks, stream, ham);
      } else {
        rc = pgnn_linear_fwd_colstats(hd, 2 * dim, p.w2, p.b2, z, dim, n, 2 * dim, dim, 0, blocks, stream);
      }
      if (rc) return rc;
      if (!folded && (rc = pgnn_bn_stats_fwd_blocks(blocks, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, st, st + dim,
                                                    st + 2 * dim, n, dim, stream)))
        return rc;
      if (!(fuse && !last))
        rc = pgnn_bn_apply_fwd(z, dim, st + 2 * dim, !last, y, dim, drop_p, drop_seed + (uint64_t)l, n, dim, stream);
      if (rc) return rc;
      continue;
    }
    if (wp) rc = stack_fwd_wp(hd, 2 * dim, wp2[l], p.b2, z, dim, n, 2 * dim, dim, 0, nullptr, stream, ham);
    else rc = pgnn_linear_fwd(hd, 2 * dim, p.w2, p.b2, z, dim, n, 2 * dim, dim, 0, stream);
    if (rc) return rc;
    if (fuse && !last)
      rc = pgnn_bn_stats_fwd(z, dim, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, training, st,
                             st + dim, st + 2 * dim, n, dim, ws, opb, stream);
    else
      rc = pgnn_bn_fwd(z, dim, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, training, !last, y, dim,
                       st, st + dim, drop_p, drop_seed + (uint64_t)l, n, dim, ws, opb, stream);
    if (rc) return rc;
  }
  return PGNN_OK;
}

int pgnn_stack_bwd_milestone_arm(int layer, const void* network) {
  GradMilestone* m = milestone_of_current_device();
  if (!m) {
    set_error("stack_bwd_milestone_arm: no current device");
    return PGNN_ERR_HIP;
  }
  std::lock_guard<std::mutex> lock(g_milestone_mutex);
  m->layer = layer;
  m->network = network;
  m->backwards = 0;
  m->recorded = false;
  return PGNN_OK;
}

int pgnn_stack_bwd_milestone_wait(pgnn_stream stream) {
  GradMilestone* m = milestone_of_current_device();
  if (!m) return 1;
  std::lock_guard<std::mutex> lock(g_milestone_mutex);
  // nothing behind an event, or more than one backward ran since the arming (the later ones ACCUMULATE into gradients the
  // recorded events do not cover): the caller orders its stream behind the whole backward
  if (m->layer < 0 || !m->recorded || m->backwards != 1) return 1;
  PGNN_HIP(hipStreamWaitEvent((hipStream_t)stream, m->ev[0], 0));
  PGNN_HIP(hipStreamWaitEvent((hipStream_t)stream, m->ev[1], 0));
  return PGNN_OK;
}

int pgnn_stack_bwd_dy_rows(const float* dy, const int64_t* rows, int64_t count) {
  t_dy_rows = DyRows{};
  if (dy && rows && count > 0) t_dy_rows = DyRows{dy, rows, count};
  return PGNN_OK;
}

int pgnn_chem_gin_stack_bwd(const float* dy, int64_t lddy, const int64_t* x_idx, int64_t rows1, int64_t rows2,
                            const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat,
                            const pgnn_gin_layer* layers, int num_layer, int training, const float* acts,
                            const float* hid, const float* stats, float* dxemb1, float* dxemb2, float drop_p,
                            uint64_t drop_seed, int64_t n, int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream) {
  const DyRows dy_rows = take_dy_rows(dy);  // (taken first: whatever this call does, the hint does not outlive it)
  if (num_layer < 1 || !layers) {
    set_error("chem_gin_stack_bwd: no layers");
    return PGNN_ERR_ARG;
  }
  if (ws_bytes < pgnn_chem_gin_stack_workspace_bytes(n, dim, rows1, rows2, num_layer)) {
    set_error("chem_gin_stack_bwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  const size_t nd = (size_t)n * dim;
  Carver cv(ws);
  const size_t opb = op_ws_bytes(n, dim);
  char* op = cv.take<char>(opb);
  char* op2 = cv.take<char>(opb);
  constexpr int kMaxSets = 64;
  const int sets = (per_layer_buffers() && n <= kSideMaxRows) ? std::min(std::max(num_layer, 2), kMaxSets) : 2;
  float *dz[kMaxSets], *dhid[kMaxSets], *dagg[kMaxSets], *dxb[kMaxSets];
  for (int p = 0; p < sets; ++p) {
    dz[p] = cv.take<float>(nd);
    dhid[p] = cv.take<float>(2 * nd);
    dagg[p] = cv.take<float>(nd);
    dxb[p] = cv.take<float>(nd);
  }
  // backward-data on the planes of W^T (every layer; one split launch up front), else on transposed fp32 weights from 16 384 rows
  const bool wp = mlp_wp(n, dim, 2 * dim, dim, num_layer);
  void *wp1[kMaxPlaneLayers], *wp2[kMaxPlaneLayers];  // planes of W1^T [dim, 2 dim], W2^T [2 dim, dim]
  const int ntr = wp ? 0 : std::min(num_layer, kMaxTransposed);  // layers whose backward-data runs on transposed fp32 weights
  float *w1t[kMaxTransposed], *w2t[kMaxTransposed];
  char* plane_base = cv.base + cv.used;  // one region, sized for the larger of the two uses (pgnn_chem_gin_stack_workspace_bytes)
  {
    const size_t region = std::max((size_t)std::min<int64_t>(num_layer, kMaxTransposed) * 2 * align_up((size_t)2 * dim * dim * 4, 256),
                                   (size_t)std::min<int64_t>(num_layer, kMaxPlaneLayers) * mlp_planes_bytes(dim, 2 * dim, dim));
    Carver tv(plane_base);
    for (int l = 0; l < ntr; ++l) {
      w1t[l] = tv.take<float>((size_t)2 * dim * dim);
      w2t[l] = tv.take<float>((size_t)2 * dim * dim);
    }
    cv.take<char>(region);
  }
  int32_t* gptr[2];
  int32_t* gperm[2];
  for (int c = 0; c < 2; ++c) {
    gptr[c] = cv.take<int32_t>(stack_keys(rows1, rows2) + 1);
    gperm[c] = cv.take<int32_t>((size_t)n);
  }
  int32_t* gstatus = cv.take<int32_t>(64);
  const size_t grp_b = stack_group_ws(n, rows1, rows2);
  char* grp_ws = cv.take<char>(grp_b);
  const size_t seg_b = stack_segsum_ws(n, dim, rows1, rows2);
  char* seg_ws = cv.take<char>(seg_b);
  float* pair_sums = reinterpret_cast<float*>(cv.take<char>(stack_pair_sums(dim, rows1, rows2)));
  // two planes: the row maxima of dhid, from the epilogue of the product that writes it to the product that reads it (cleared by the split launch)
  uint32_t* dhid_amax = (two_planes() && num_layer <= kMaxPlaneLayers) ? cv.take<uint32_t>((size_t)num_layer * amax_words(n)) : nullptr;
  // ... and the BatchNorm backward's elementwise pass those of dz (one vector, rewritten by every layer: plain stores)
  uint32_t* const dz_amax = (dhid_amax && env_knob("PGNN_PRODUCER_AMAX", 1) != 0) ? cv.take<uint32_t>(amax_words(n)) : nullptr;
  // bond-table gradients out of the dW1 products (see bond_ws_bytes).  The amax vectors above are carved conditionally, so these
  // come from the END of the workspace, whose size does not depend on any knob.
  const bool bond_in_dw = cfeat && num_layer <= kMaxBondJobs && linear_bwd_weight_pair_ext_ok(n, 2 * dim, dim, dim, 2 * dim);
  float* cfeat12 = nullptr;
  float* gbond[kMaxBondJobs];
  if (bond_in_dw) {
    Carver bv(static_cast<char*>(ws) + pgnn_chem_gin_stack_workspace_bytes(n, dim, rows1, rows2, num_layer) - bond_ws_bytes(n, dim, num_layer));
    cfeat12 = bv.take<float>((size_t)n * 12);
    for (int l = 0; l < num_layer; ++l) gbond[l] = bv.take<float>((size_t)2 * dim * 12);
  }
  BondTableJob bond_jobs[kMaxBondJobs];
  int n_bond_jobs = 0;

  hipStream_t main = (hipStream_t)stream;
  Side* sd = (use_side_stream() && n >= side_min_rows() && n <= kSideMaxRows && num_layer <= kMaxSets) ? side_for_current_device() : nullptr;
  // One buffer set per layer would save the lag events (8 event calls per step), but the ping-pong pair stays
  // resident in the 256 MB Infinity Cache and wins on the GPU side: 1.80 vs 1.85 ms per replayed step (measured).
  const bool per_layer = sd && per_layer_buffers();
  hipStream_t aux = sd ? sd->stream : main;
  char* aux_ws = sd ? op2 : op;
  int rc;
  // Every event record / wait costs ~7 us of host time, so forks are spent only where they buy overlap.
  // (gstatus is scratch: the forward's embedding lookup validated x_idx, nobody reads the grouping's count -- no memset launch)
  // W1^T / W2^T planes of the whole stack in one launch (without planes: transposes of the top `ntr` layers): backward-data
  // then has both operands contiguous along the contracted dimension and runs the forward (split-bf16) kernel.
  // The atom-type x chirality grouping of the embedding gradients needs nothing from the backward: with a side stream it
  // runs there, behind the split, while the caller's stream does the top layer's BatchNorm backward and first products
  // (without one it stays on the caller's stream after the layers, in front of the segment sums).
  const bool group_early = sd != nullptr;
  if (sd) {
    PGNN_HIP(hipEventRecord(sd->fork[0], main));  // (the workspace words it writes were the previous call's until here)
    PGNN_HIP(hipStreamWaitEvent(aux, sd->fork[0], 0));
  }
  // The caller's stream needs the planes (or transposes) for its first product only, i.e. after the top layer's BatchNorm
  // backward: the wait for fork[2] is enqueued there, not here, and the side stream does the split BEFORE the grouping
  // (fork[3], awaited in front of the embedding gradients) -- with the wait up here the step began with ~26 us of the
  // caller's stream idling behind four tiny side-stream launches.
  bool wait_fork2 = false;
  if (wp) {
    if ((rc = split_mlp_weights(layers, num_layer, dim, 2 * dim, dim, 1, plane_base, wp1, wp2, aux, false, nullptr, dhid_amax,
                                dhid_amax ? (int64_t)num_layer * (int64_t)amax_words(n) : 0)))  // (aux already waits on fork[0])
      return rc;
    if (sd) {
      PGNN_HIP(hipEventRecord(sd->fork[2], aux));
      wait_fork2 = true;
    }
  } else if (ntr > 0 && use_transposed_weights(n)) {
    const float* tsrc[2 * kMaxTransposed];
    float* tdst[2 * kMaxTransposed];
    int64_t trows[2 * kMaxTransposed], tcols[2 * kMaxTransposed];
    for (int q = 0; q < ntr; ++q) {
      const pgnn_gin_layer& p = layers[num_layer - 1 - q];
      tsrc[2 * q] = p.w1; tdst[2 * q] = w1t[q]; trows[2 * q] = 2 * dim; tcols[2 * q] = dim;          // W1 [2d, d]
      tsrc[2 * q + 1] = p.w2; tdst[2 * q + 1] = w2t[q]; trows[2 * q + 1] = dim; tcols[2 * q + 1] = 2 * dim;  // W2 [d, 2d]
    }
    if ((rc = pgnn_transpose_batch(tsrc, tdst, trows, tcols, 2 * ntr, aux))) return rc;  // (aux already waits on fork[0])
    if (sd) {
      PGNN_HIP(hipEventRecord(sd->fork[2], aux));
      wait_fork2 = true;
    }
  }
  if (sd) {
    if ((rc = embed_tables_group(x_idx, n, rows1, rows2, dxemb1, dxemb2, gptr[0], gperm[0], gptr[1], gperm[1], gstatus, grp_ws, grp_b, aux)))
      return rc;
    PGNN_HIP(hipEventRecord(sd->fork[3], aux));
  }
  // (behind everything the caller's stream waits for; first read by the top layer's weight gradients, on aux too)
  if (bond_in_dw && (rc = pad_rowfeat12(cfeat, 9, cfeat12, n, aux))) return rc;
  const bool tr = ntr > 0 && use_transposed_weights(n);
  const bool fused_mlp = wp && mlp_fused(n, dim, 2 * dim, dim);
  const bool fork_via_launch = env_knob("PGNN_FORK_VIA_LAUNCH", 1) != 0 && sd && !stream_is_capturing(main);

  const float* g = dy;
  int64_t ldg = lddy;
  bool sums_ready = false;   // the BatchNorm-backward sums of the layer about to run are already folded (in bn_scratch.coef)
  BnBwdScratch bn_scratch{};
  // The parameter gradients of layer l on `aux` behind fork[1]: both weight-gradient products in one launch + the fold of their
  // split-K partials (with bond_in_dw the dW1 product carries G = dhid^T cfeat along -- twelve columns of its tile padding -- and the
  // bond-table gradient demb = G^T W1 needs no pass over dagg: one launch for all layers behind the loop), else the bond tables' own pass
  // sampled ONCE per call: an arm issued by another thread half way through this backward must not split the layers
  // between the immediate and the deferred bond-table launches -- the milestone's events would then not cover the deferred ones
  const bool armed_at_entry = milestone_armed();
  auto side_work = [&](int l, bool demb_on_main) -> int {
    const pgnn_gin_layer& p = layers[l];
    const int b = per_layer ? l : (l & 1);
    const float* agg = acts + (size_t)l * 3 * nd;
    const float* hd = hid + (size_t)l * 2 * nd;
    int r;
    if (sd) PGNN_HIP(hipStreamWaitEvent(aux, sd->fork[1], 0));
    bool g_done = false;
    if ((r = linear_bwd_weight_pair_ext(dz[b], dim, hd, 2 * dim, p.dw2, p.db2, 2 * dim, dim, dhid[b], 2 * dim, agg, dim, p.dw1, p.db1, dim,
                                        2 * dim, n, aux_ws, opb, aux, bond_in_dw ? cfeat12 : nullptr, bond_in_dw ? gbond[l] : nullptr, &g_done)))
      return r;
    if (g_done) {
      bond_jobs[n_bond_jobs++] = BondTableJob{gbond[l], p.w1, dim, p.demb, dim};
      if (armed_at_entry) {  // a communication stream may be waiting for this layer's gradients: no deferral
        if ((r = bond_tables_from_g(bond_jobs + n_bond_jobs - 1, 1, 2 * dim, dim, 9, aux))) return r;
        --n_bond_jobs;
      }
    } else if (!demb_on_main || bond_in_dw) {  // (bond_in_dw without g_done cannot happen: linear_bwd_weight_pair_ext_ok said the one-launch path runs)
      if ((r = pgnn_rowfeat_matmul_bwd(cfeat, 9, dagg[b], dim, p.demb, dim, n, dim, aux_ws, opb, aux))) return r;
    }
    if (sd && !per_layer && l >= 2) PGNN_HIP(hipEventRecord(sd->lag[b], aux));  // awaited by layer l-2 only
    // every parameter gradient of layers >= l is enqueued now (weights, biases and edge tables on `aux`, the BatchNorm's on `main`;
    // layer l - 1's BatchNorm sums, which the transposed aggregation also leaves, only arrive early)
    return milestone_record(l, p.w1, main, aux);
  };
  // (Measured and NOT kept, profiles/r05/fork_late_ab.txt: forking layer l's parameter gradients behind layer l - 1's BatchNorm
  // elementwise pass instead -- that pass and the transposed aggregation then run alone (6 and 17 us instead of 26 and 19), but the
  // second backward-data product, whose 115 KB of LDS cannot share a CU with two 55 KB weight-gradient workgroups, waits for them to
  // retire: 49-58 us instead of 15-18, step 1.007-1.012 against 0.968-0.972 ms.)
  for (int l = num_layer - 1; l >= 0; --l) {
    const pgnn_gin_layer& p = layers[l];
    const int q = num_layer - 1 - l;  // index into the transposed weights
    const int b = per_layer ? l : (l & 1);  // own buffer set per layer, or ping-pong guarded by lag events
    if (sd && !per_layer && l + 2 <= num_layer - 1) PGNN_HIP(hipStreamWaitEvent(main, sd->lag[b], 0));
    const float* a = acts + (size_t)l * 3 * nd;  // agg, z, y
    const float* z = a + nd;
    const float* hd = hid + (size_t)l * 2 * nd;
    const float* mean = stats + (size_t)l * 4 * dim;
    // BatchNorm backward.  Below the top layer its column sums came out of the aggregation that produced g (the layer above's
    // transposed aggregation, neighbor_sum_bn_bwd below): only the elementwise pass is left.
    const bool dz_has_amax = sums_ready && wp && dz_amax != nullptr;
    if (sums_ready)
      rc = bn_bwd_apply_only(g, ldg, z, dim, bn_scratch.coef, l != num_layer - 1, dz[b], dim, n, dim, main, dz_has_amax ? dz_amax : nullptr);
    else  // (the top layer; given the rows outside which dy is zero, its column sums visit only those: pgnn_stack_bwd_dy_rows)
      rc = bn_bwd_rows(g, ldg, z, dim, p.gamma, p.beta, mean, mean + dim, training, l != num_layer - 1, dz[b], dim, p.dgamma, p.dbeta,
                       drop_p, drop_seed + (uint64_t)l, n, dim, op, opb, main, l == num_layer - 1 ? dy_rows.rows : nullptr, dy_rows.count);
    if (rc) return rc;
    sums_ready = false;
    bool fork_recorded = false;
    if (wait_fork2) {
      PGNN_HIP(hipStreamWaitEvent(main, sd->fork[2], 0));
      wait_fork2 = false;
    }
    if (wp && fused_mlp) {
      // dhid = (dz . W2) * (hid > 0) and dagg = dhid . W1 in one launch: dhid written once (the weight gradient reads it), never re-read here
      if ((rc = mlp_bwd_data_2p_fused(dz[b], dim, wp2[l], hd, 2 * dim, wp1[l], dhid[b], 2 * dim, dagg[b], dim, n, dim, 2 * dim, dim, main))) return rc;
    } else if (wp) {
      uint32_t* dam = dhid_amax ? dhid_amax + (size_t)l * amax_words(n) : nullptr;
      if ((rc = stack_bwd_data_wp(dz[b], dim, wp2[l], hd, 2 * dim, dhid[b], 2 * dim, n, 2 * dim, dim, main, dz_has_amax ? dz_amax : nullptr, dam))) return rc;
      // PGNN_FORK_VIA_LAUNCH=1: fork[1] is the completion of this product's own dispatch instead of a marker behind it
      if (sd && fork_via_launch) set_next_launch_stop_event(sd->fork[1]);
      rc = stack_bwd_data_wp(dhid[b], 2 * dim, wp1[l], nullptr, 0, dagg[b], dim, n, dim, 2 * dim, main, dam);
      const bool taken = take_next_launch_stop_event() == nullptr;  // (always cleared here: by the launch, or -- another kernel ran, an error -- now)
      if (rc) return rc;
      fork_recorded = sd && fork_via_launch && taken;
    } else if (tr && q < ntr) {
      if ((rc = pgnn_linear_bwd_data_t(dz[b], dim, w2t[q], hd, 2 * dim, dhid[b], 2 * dim, n, 2 * dim, dim, main))) return rc;
      if ((rc = pgnn_linear_bwd_data_t(dhid[b], 2 * dim, w1t[q], nullptr, 0, dagg[b], dim, n, dim, 2 * dim, main))) return rc;
    } else {
      if ((rc = pgnn_linear_bwd_data(dz[b], dim, p.w2, hd, 2 * dim, dhid[b], 2 * dim, n, 2 * dim, dim, main))) return rc;
      if ((rc = pgnn_linear_bwd_data(dhid[b], 2 * dim, p.w1, nullptr, 0, dagg[b], dim, n, dim, 2 * dim, main))) return rc;
    }
    // (recording fork[1] behind the transposed aggregation instead -- one idle gap less on this stream per layer -- starts the side
    // stream ~11 us later and loses: 1.107-1.111 against 1.072-1.078 ms per step, profiles/r03/fork_placement_ab.txt)
    // (round 4: recording it EARLIER, between the two backward-data products -- the weight gradients need dz and dhid, not dagg -- with
    // the edge-table gradient deferred to the head of the next layer's side work: bit-identical, 1.001-1.010 against 0.997-1.001 ms,
    // profiles/r04/fork_early_ab.txt; not kept)
    if (sd && !fork_recorded) PGNN_HIP(hipEventRecord(sd->fork[1], main));
    // the bottom layer's edge-table gradient stays on the caller's stream: the side stream is the longer of the two there
    // (two weight-gradient products behind the data products), and the caller's stream only has the embedding gradients left
    const bool demb_on_main = sd && l == 0;
    // The caller's stream is the critical path: its transposed aggregation is enqueued BEFORE the side stream's launches
    // (it reads dagg, which they only read, and writes dx, which they never touch), so that stream is never waiting for the host.
    // the transposed aggregation; for l > 0 its launch also leaves the BatchNorm-backward sums of layer l - 1 (whose output
    // gradient it is writing) in the caller's op scratch -- folded, as pgnn_bn_bwd's first launch would
    auto aggregate_t = [&]() -> int {
      if (l > 0 && drop_p == 0.f) {
        const pgnn_gin_layer& q1 = layers[l - 1];
        const float* st1 = stats + (size_t)(l - 1) * 4 * dim;
        BnBwdTail tail{acts + (size_t)(l - 1) * 3 * nd + nd, dim, q1.gamma, q1.beta, st1, st1 + dim, 1, training, BnBwdScratch{}, q1.dgamma, q1.dbeta};
        if (int r = bn_bwd_scratch(op, opb, n, dim, &tail.scratch)) return r;
        bn_scratch = tail.scratch;
        return neighbor_sum_bn_bwd(dagg[b], dim, out_ptr, out_dst, dxb[b], dim, n, dim, tail, &sums_ready, main);
      }
      return pgnn_neighbor_sum(dagg[b], dim, out_ptr, out_dst, nullptr, dxb[b], dim, n, dim, main);
    };
    if (sd) {
      if ((rc = aggregate_t())) return rc;
      if (demb_on_main && !bond_in_dw && (rc = pgnn_rowfeat_matmul_bwd(cfeat, 9, dagg[b], dim, p.demb, dim, n, dim, op, opb, main))) return rc;
    }
    if ((rc = side_work(l, demb_on_main))) return rc;
    if (!sd && (rc = aggregate_t())) return rc;
    g = dxb[b];
    ldg = dim;
  }
  // The embedding gradients need the last aggregation only, not the side stream's weight gradients: they run beside the
  // bottom layer's two weight-gradient products, and the join comes after them (it used to come before: ~70 us of one
  // stream idling per 256-graph step).
  if (sd) PGNN_HIP(hipStreamWaitEvent(main, sd->fork[3], 0));  // the grouping (finished long ago)
  rc = embed_tables_bwd(g, x_idx, n, dim, rows1, rows2, dxemb1, dxemb2, gptr[0], gperm[0], gptr[1], gperm[1], gstatus, grp_ws,
                        grp_b, seg_ws, seg_b, pair_sums, group_early, main);
  if (rc) return rc;
  if (n_bond_jobs > 0) rc = bond_tables_from_g(bond_jobs, n_bond_jobs, 2 * dim, dim, 9, aux);  // every layer's demb = G^T W1, one launch
  if (sd) {  // join: nothing of this call is in flight on the side stream once the caller's stream passes this point
    PGNN_HIP(hipEventRecord(sd->join, aux));
    PGNN_HIP(hipStreamWaitEvent(main, sd->join, 0));
  }
  return rc;
}

/* ---------------------------------------------------------------------------------------------
 * The same one-call network for the "linear, then aggregate" convolutions of chem/model.py:
 *   kind 1 = GCNConv  (:58-104)  : lin = h W^T + b ; z = sum_e dinv_i dinv_j (lin_j + e_ij)   (self loop incl.)
 *   kind 2 = GraphSAGE (:165-202): lin = h W^T + b ; s = sum_e (lin_j + e_ij) ; z = normalize(s / count)
 * followed by BatchNorm (+ReLU except after the last layer, + optional fused dropout), JK = "last".
 * acts [L][4][n][dim] = (lin, sum (GraphSAGE only), z, y); norms [L][n] (GraphSAGE only);
 * stats [L][4][dim] as for the GIN stack.  The per-layer struct uses w1/b1 (+ dw1/db1) for the Linear.
 * --------------------------------------------------------------------------------------------- */
size_t pgnn_chem_lin_stack_workspace_bytes(int64_t n, int64_t dim, int64_t rows1, int64_t rows2) {
  const size_t nd = align_up((size_t)n * dim * 4, 256);
  return op_ws_bytes(n, dim) + 2 * 4 * nd + 2 * align_up((size_t)n * 4, 256) +
         2 * align_up((stack_keys(rows1, rows2) + 1) * 4, 256) + 256 + stack_group_ws(n, rows1, rows2) +
         stack_segsum_ws(n, dim, rows1, rows2) + stack_pair_sums(dim, rows1, rows2) + 256;
}

int pgnn_chem_lin_stack_fwd(int kind, const int64_t* x_idx, const float* xemb1, int64_t rows1, const float* xemb2,
                            int64_t rows2, const int32_t* in_ptr, const int32_t* in_src, const uint8_t* in_code,
                            const float* dinv, const pgnn_gin_layer* layers, int num_layer, int training, float* h0,
                            float* acts, float* norms, float* stats, int32_t* status, float drop_p, uint64_t drop_seed,
                            int64_t n, int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream) {
  if (num_layer < 1 || !layers || (kind != 1 && kind != 2) || (kind == 1 && !dinv) || (kind == 2 && !norms)) {
    set_error("chem_lin_stack_fwd: bad arguments");
    return PGNN_ERR_ARG;
  }
  if (ws_bytes < op_ws_bytes(n, dim)) {
    set_error("chem_lin_stack_fwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  int rc;
  if ((rc = pgnn_embed_fwd(x_idx, 2, xemb1, rows1, xemb2, rows2, h0, dim, n, dim, status, stream))) return rc;
  const size_t nd = (size_t)n * dim;
  const float* h = h0;
  for (int l = 0; l < num_layer; ++l) {
    const pgnn_gin_layer& p = layers[l];
    float* a = acts + (size_t)l * 4 * nd;
    float *lin = a, *sum = a + nd, *z = a + 2 * nd, *y = a + 3 * nd;
    float* st = stats + (size_t)l * 4 * dim;
    if ((rc = pgnn_linear_fwd(h, dim, p.w1, p.b1, lin, dim, n, dim, dim, 0, stream))) return rc;
    if (kind == 1) {
      if ((rc = pgnn_chem_aggregate_fwd(lin, dim, in_ptr, in_src, in_code, p.emb1, p.emb2, dinv, z, dim, n, dim, stream))) return rc;
    } else {
      if ((rc = pgnn_chem_aggregate_fwd(lin, dim, in_ptr, in_src, in_code, p.emb1, p.emb2, nullptr, sum, dim, n, dim, stream))) return rc;
      if ((rc = pgnn_mean_l2norm_fwd(sum, dim, in_ptr, z, dim, norms + (size_t)l * n, n, dim, stream))) return rc;
    }
    if ((rc = pgnn_bn_fwd(z, dim, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, training,
                          l != num_layer - 1, y, dim, st, st + dim, drop_p, drop_seed + (uint64_t)l, n, dim, ws, ws_bytes,
                          stream))) return rc;
    h = y;
  }
  return PGNN_OK;
}

int pgnn_chem_lin_stack_bwd(int kind, const float* dy, int64_t lddy, const int64_t* x_idx, int64_t rows1, int64_t rows2,
                            const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_dst, const float* dinv,
                            const float* cfeat, const pgnn_gin_layer* layers, int num_layer, int training,
                            const float* h0, const float* acts, const float* norms, const float* stats, float* dxemb1,
                            float* dxemb2, float drop_p, uint64_t drop_seed, int64_t n, int64_t dim, void* ws,
                            size_t ws_bytes, pgnn_stream stream) {
  if (num_layer < 1 || !layers || (kind != 1 && kind != 2) || (kind == 1 && !dinv) || (kind == 2 && !norms)) {
    set_error("chem_lin_stack_bwd: bad arguments");
    return PGNN_ERR_ARG;
  }
  if (ws_bytes < pgnn_chem_lin_stack_workspace_bytes(n, dim, rows1, rows2)) {
    set_error("chem_lin_stack_bwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  // Everything on the caller's stream: with a third of the GIN stack's MFMA work per layer this path is
  // bound by the host's launch rate at the reference's batch size, and the ~20 event calls a forked
  // backward needs cost more (0.15 ms) than the overlap returns (measured: 1.49 vs 1.57 ms per GCN step).
  const size_t nd = (size_t)n * dim;
  Carver cv(ws);
  const size_t opb = op_ws_bytes(n, dim);
  char* op = cv.take<char>(opb);
  float *dz[2], *dsum[2], *dlin[2], *dxb[2];
  for (int p = 0; p < 2; ++p) {
    dz[p] = cv.take<float>(nd);
    dsum[p] = cv.take<float>(nd);
    dlin[p] = cv.take<float>(nd);
    dxb[p] = cv.take<float>(nd);
  }
  int32_t* gptr[2];
  int32_t* gperm[2];
  for (int c = 0; c < 2; ++c) {
    gptr[c] = cv.take<int32_t>(stack_keys(rows1, rows2) + 1);
    gperm[c] = cv.take<int32_t>((size_t)n);
  }
  int32_t* gstatus = cv.take<int32_t>(64);
  const size_t grp_b = stack_group_ws(n, rows1, rows2);
  char* grp_ws = cv.take<char>(grp_b);
  const size_t seg_b = stack_segsum_ws(n, dim, rows1, rows2);
  char* seg_ws = cv.take<char>(seg_b);
  float* pair_sums = reinterpret_cast<float*>(cv.take<char>(stack_pair_sums(dim, rows1, rows2)));

  hipStream_t main = (hipStream_t)stream;
  int rc;

  const float* g = dy;
  int64_t ldg = lddy;
  for (int l = num_layer - 1; l >= 0; --l) {
    const pgnn_gin_layer& p = layers[l];
    const int b = l & 1;
    const float* a = acts + (size_t)l * 4 * nd;  // lin, sum, z, y
    const float* z = a + 2 * nd;
    const float* hin = l == 0 ? h0 : acts + (size_t)(l - 1) * 4 * nd + 3 * nd;
    const float* mean = stats + (size_t)l * 4 * dim;
    if ((rc = pgnn_bn_bwd(g, ldg, z, dim, p.gamma, p.beta, mean, mean + dim, training, l != num_layer - 1, dz[b], dim,
                          p.dgamma, p.dbeta, drop_p, drop_seed + (uint64_t)l, n, dim, op, opb, main))) return rc;
    const float* dagg = dz[b];  // gradient of the aggregation's output
    if (kind == 2) {
      if ((rc = pgnn_mean_l2norm_bwd(dz[b], dim, z, dim, norms + (size_t)l * n, in_ptr, dsum[b], dim, n, dim, main))) return rc;
      dagg = dsum[b];
    }
    if ((rc = pgnn_neighbor_sum(dagg, dim, out_ptr, out_dst, kind == 1 ? dinv : nullptr, dlin[b], dim, n, dim, main))) return rc;
    if ((rc = pgnn_linear_bwd_data(dlin[b], dim, p.w1, nullptr, 0, dxb[b], dim, n, dim, dim, main))) return rc;
    if ((rc = pgnn_linear_bwd_weight(dlin[b], dim, hin, dim, p.dw1, p.db1, n, dim, dim, op, opb, main))) return rc;
    if ((rc = pgnn_rowfeat_matmul_bwd(cfeat, 9, dagg, dim, p.demb, dim, n, dim, op, opb, main))) return rc;
    g = dxb[b];
    ldg = dim;
  }
  return embed_tables_bwd(g, x_idx, n, dim, rows1, rows2, dxemb1, dxemb2, gptr[0], gperm[0], gptr[1], gperm[1], gstatus,
                          grp_ws, grp_b, seg_ws, seg_b, pair_sums, false, main);
}

}  // extern "C"

/* ---------------------------------------------------------------------------------------------
 * The bio GIN network in one call per direction (bio/model.py:11-58, 227-290, JK = "last", no dropout):
 *   agg = [sum_j h_j + h_i | cfeat . EncT]   (graph-resident aggregation, csrc/tile.hip, both halves in one launch)
 *   pre = agg W1^T + b1 ; hid = relu(BN_2D(pre)) ; y = hid W2^T + b2, ReLU'd between layers (fused in the product's epilogue)
 * pgnn_gin_layer is reused: emb1 = EncT [10, dim] = [W_enc^T; b_enc] (emb2 unused), w1 [2D,2D], b1 [2D], w2 [D,2D], b2 [D],
 * gamma / beta / running stats of the BatchNorm1d(2D) inside the mlp; gradients: demb = d EncT [10, dim], the rest as named.
 * acts [num_layer][7][n][dim] = (agg 2, pre 2, hid 2, y 1 slots of n*dim); stats [num_layer][2][2D] = (mean, 1/std).
 * --------------------------------------------------------------------------------------------- */
namespace {
__global__ void __launch_bounds__(256) k_relu_mask(float* __restrict__ g, const float* __restrict__ y, int64_t n4) {
  for (int64_t q = blockIdx.x * (int64_t)256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<float4*>(g)[q];
    const float4 m = reinterpret_cast<const float4*>(y)[q];
    if (!(m.x > 0.f)) v.x = 0.f;
    if (!(m.y > 0.f)) v.y = 0.f;
    if (!(m.z > 0.f)) v.z = 0.f;
    if (!(m.w > 0.f)) v.w = 0.f;
    reinterpret_cast<float4*>(g)[q] = v;
  }
}
inline size_t bio_op_ws_bytes(int64_t n, int64_t d) {
  size_t m = pgnn_bn_workspace_bytes(n, 2 * d);
  m = std::max(m, pgnn_linear_bwd_weight_workspace_bytes(n, 2 * d, 2 * d));
  m = std::max(m, pgnn_linear_bwd_weight_workspace_bytes(n, 2 * d, d));
  m = std::max(m, pgnn_rowfeat_matmul_bwd_workspace_bytes(n, 10, d));
  return align_up(m, 256);
}
}  // namespace

extern "C" {

size_t pgnn_bio_gin_stack_workspace_bytes(int64_t n, int64_t dim, int64_t num_layer) {
  const size_t nd = align_up((size_t)n * dim * 4, 256);
  // 2 x op scratch + 2 x (dhid, dpre, dagg: 2 nd each; dx: nd) + W1^T, W2^T per layer
  const size_t wt = std::max((size_t)std::min<int64_t>(num_layer, kMaxTransposed) * (align_up((size_t)4 * dim * dim * 4, 256) + align_up((size_t)2 * dim * dim * 4, 256)),
                             (size_t)std::min<int64_t>(num_layer, kMaxPlaneLayers) * mlp_planes_bytes(2 * dim, 2 * dim, dim));
  // + the per-16-row column statistics of the 2D-wide pre-activation and the BatchNorm coefficients (forward, statistics from the
  // product's epilogue)
  const size_t blocks = align_up((size_t)ceil_div(n, 16) * 2 * 2 * dim * 4, 256) + align_up((size_t)2 * 2 * dim * 4, 256);
  // + the [10, dim] edge-encoder tables the forward builds when it is handed edge_encoder.weight / .bias as they are
  const size_t tabs = align_up((size_t)std::min<int64_t>(num_layer, 16) * 10 * dim * 4, 256);
  return 2 * bio_op_ws_bytes(n, dim) + 2 * 7 * nd + wt + blocks + tabs + 512;
}

int pgnn_bio_gin_stack_fwd(const float* h0, int64_t ldh0, const int32_t* in_ptr, const int32_t* in_src, const float* cfeat,
                           const int32_t* tile_start, const int32_t* num_tiles, const pgnn_gin_layer* layers, int num_layer,
                           int training, float* acts, float* stats, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                           pgnn_stream stream) {
  if (num_layer < 1 || !layers || !h0 || !cfeat) {
    set_error("bio_gin_stack_fwd: bad arguments");
    return PGNN_ERR_ARG;
  }
  if (ws_bytes < bio_op_ws_bytes(n, dim)) {
    set_error("bio_gin_stack_fwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  const size_t nd = (size_t)n * dim;
  const float* h = h0;
  int64_t ldh = ldh0;
  int rc;
  // both products of every layer on pre-split weight planes (behind the op scratch of the workspace), as in the chem stack
  const size_t opb = bio_op_ws_bytes(n, dim);
  void *wp1[kMaxPlaneLayers], *wp2[kMaxPlaneLayers];
  const size_t planes_b = (size_t)num_layer * mlp_planes_bytes(2 * dim, 2 * dim, dim);
  const bool wp = mlp_wp(n, 2 * dim, 2 * dim, dim, num_layer) && ws_bytes >= opb + planes_b;
  // layers[l].emb2 set: emb1 / emb2 are edge_encoder.weight [dim, 9] / .bias [dim] as the module holds them, and the [10, dim] table
  // [W^T; b] of every layer is written by the launch that splits the weights (emb2 NULL: emb1 IS that table)
  const bool enc_raw = layers[0].emb2 != nullptr;
  const float* table[kMaxPlaneLayers];
  EncTables tabs{};
  const size_t coef_b = align_up((size_t)2 * 2 * dim * 4, 256);
  const size_t blocks_b = align_up((size_t)ceil_div(n, 16) * 2 * 2 * dim * 4, 256);
  if (enc_raw) {
    const size_t tabs_off = opb + planes_b + blocks_b + coef_b, tabs_b = align_up((size_t)num_layer * 10 * dim * 4, 256);
    PGNN_REQUIRE(num_layer <= 16 && ws_bytes >= tabs_off + tabs_b, "bio_gin_stack_fwd: workspace too small for %d encoder tables", num_layer);
    float* tb = reinterpret_cast<float*>(static_cast<char*>(ws) + tabs_off);
    tabs.count = num_layer, tabs.dim = (int)dim, tabs.k = 9;
    for (int l = 0; l < num_layer; ++l) {
      PGNN_REQUIRE(layers[l].emb1 && layers[l].emb2, "bio_gin_stack_fwd: edge encoder of layer %d missing", l);
      tabs.w[l] = layers[l].emb1, tabs.b[l] = layers[l].emb2, tabs.dst[l] = tb + (size_t)l * 10 * dim;
      table[l] = tabs.dst[l];
    }
  }
  if (wp && (rc = split_mlp_weights(layers, num_layer, 2 * dim, 2 * dim, dim, 0, static_cast<char*>(ws) + opb, wp1, wp2, (hipStream_t)stream,
                                    training != 0, enc_raw ? &tabs : nullptr)))
    return rc;
  if (!wp && training && (rc = bump_batches_tracked(layers, num_layer, (hipStream_t)stream))) return rc;
  if (!wp && enc_raw && (rc = split_weights_bump(nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, (hipStream_t)stream, &tabs))) return rc;
  // training-mode statistics of the mlp's BatchNorm1d(2D) from the epilogue of the product that writes its input (the chem form,
  // csrc/batchnorm.hip pgnn_bn_stats_fwd_blocks): no pass over `pre` for them, two launches less per layer
  const bool stats_in_gemm = wp && training && n > 1 && n <= kStatsInGemmMaxRows && env_knob("PGNN_BN_STATS_IN_GEMM", 1) != 0 &&
                             ws_bytes >= opb + planes_b + blocks_b + coef_b;
  float* blocks = reinterpret_cast<float*>(static_cast<char*>(ws) + opb + planes_b);
  float* coef = reinterpret_cast<float*>(static_cast<char*>(ws) + opb + planes_b + blocks_b);
  for (int l = 0; l < num_layer; ++l) {
    const pgnn_gin_layer& p = layers[l];
    float* a = acts + (size_t)l * 7 * nd;
    float *agg = a, *pre = a + 2 * nd, *hid = a + 4 * nd, *y = a + 6 * nd;
    float* st = stats + (size_t)l * 4 * dim;  // mean [2D], invstd [2D]
    if (tile_start && num_tiles) {
      rc = pgnn_neighbor_sum_tiled(h, ldh, in_ptr, in_src, nullptr, tile_start, num_tiles, agg, 2 * dim, n, dim, cfeat, 10,
                                   enc_raw ? table[l] : p.emb1, dim, agg + dim, 2 * dim, stream);
    } else {
      if ((rc = pgnn_neighbor_sum(h, ldh, in_ptr, in_src, nullptr, agg, 2 * dim, n, dim, stream))) return rc;
      rc = pgnn_rowfeat_matmul_fwd(cfeat, 10, enc_raw ? table[l] : p.emb1, dim, agg + dim, 2 * dim, n, dim, 0, stream);
    }
    if (rc) return rc;
    if (wp) rc = stack_fwd_wp(agg, 2 * dim, wp1[l], p.b1, pre, 2 * dim, n, 2 * dim, 2 * dim, 0, stats_in_gemm ? blocks : nullptr, stream);
    else rc = pgnn_linear_fwd(agg, 2 * dim, p.w1, p.b1, pre, 2 * dim, n, 2 * dim, 2 * dim, 0, stream);
    if (rc) return rc;
    if (stats_in_gemm) {
      if ((rc = pgnn_bn_stats_fwd_blocks(blocks, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, st, st + 2 * dim, coef, n,
                                         2 * dim, stream))) return rc;
      rc = pgnn_bn_apply_fwd(pre, 2 * dim, coef, 1, hid, 2 * dim, 0.f, 0, n, 2 * dim, stream);
    } else {
      rc = pgnn_bn_fwd(pre, 2 * dim, p.gamma, p.beta, p.running_mean, p.running_var, p.momentum, p.eps, training, 1, hid, 2 * dim, st,
                       st + 2 * dim, 0.f, 0, n, 2 * dim, ws, opb, stream);
    }
    if (rc) return rc;
    if (wp) rc = stack_fwd_wp(hid, 2 * dim, wp2[l], p.b2, y, dim, n, 2 * dim, dim, l != num_layer - 1, nullptr, stream);
    else rc = pgnn_linear_fwd(hid, 2 * dim, p.w2, p.b2, y, dim, n, 2 * dim, dim, l != num_layer - 1, stream);
    if (rc) return rc;
    h = y;
    ldh = dim;
  }
  return PGNN_OK;
}

int pgnn_bio_gin_stack_bwd(const float* dy, int64_t lddy, const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat,
                           const int32_t* tile_start, const int32_t* num_tiles, const pgnn_gin_layer* layers, int num_layer,
                           int training, const float* acts, const float* stats, float* dh0, int64_t n, int64_t dim, void* ws,
                           size_t ws_bytes, pgnn_stream stream) {
  if (num_layer < 1 || !layers || !dy || !cfeat) {
    set_error("bio_gin_stack_bwd: bad arguments");
    return PGNN_ERR_ARG;
  }
  if (ws_bytes < pgnn_bio_gin_stack_workspace_bytes(n, dim, num_layer)) {
    set_error("bio_gin_stack_bwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  const size_t nd = (size_t)n * dim;
  Carver cv(ws);
  const size_t opb = bio_op_ws_bytes(n, dim);
  char* op = cv.take<char>(opb);
  char* op2 = cv.take<char>(opb);
  float *dhid[2], *dpre[2], *dagg[2], *dxb[2];
  for (int q = 0; q < 2; ++q) {
    dhid[q] = cv.take<float>(2 * nd);
    dpre[q] = cv.take<float>(2 * nd);
    dagg[q] = cv.take<float>(2 * nd);
    dxb[q] = cv.take<float>(nd);
  }
  const bool wp = mlp_wp(n, 2 * dim, 2 * dim, dim, num_layer);  // backward-data on the planes of W1^T [2D, 2D], W2^T [2D, D]
  void *wp1[kMaxPlaneLayers], *wp2[kMaxPlaneLayers];
  const int ntr = wp ? 0 : std::min(num_layer, kMaxTransposed);
  float *w1t[kMaxTransposed], *w2t[kMaxTransposed];
  char* plane_base = cv.base + cv.used;
  {
    const size_t region = std::max((size_t)std::min<int64_t>(num_layer, kMaxTransposed) * (align_up((size_t)4 * dim * dim * 4, 256) + align_up((size_t)2 * dim * dim * 4, 256)),
                                   (size_t)std::min<int64_t>(num_layer, kMaxPlaneLayers) * mlp_planes_bytes(2 * dim, 2 * dim, dim));
    Carver tv(plane_base);
    for (int q = 0; q < ntr; ++q) {
      w1t[q] = tv.take<float>((size_t)4 * dim * dim);
      w2t[q] = tv.take<float>((size_t)2 * dim * dim);
    }
    cv.take<char>(region);
  }
  hipStream_t main = (hipStream_t)stream;
  Side* sd = (use_side_stream() && n <= kSideMaxRows) ? side_for_current_device() : nullptr;
  hipStream_t aux = sd ? sd->stream : main;
  char* aux_ws = sd ? op2 : op;
  int rc;
  const bool tr = ntr > 0 && use_transposed_weights(n);
  if (wp) {
    if (sd) {
      PGNN_HIP(hipEventRecord(sd->fork[0], main));
      PGNN_HIP(hipStreamWaitEvent(aux, sd->fork[0], 0));
    }
    if ((rc = split_mlp_weights(layers, num_layer, 2 * dim, 2 * dim, dim, 1, plane_base, wp1, wp2, aux))) return rc;
    if (sd) {
      PGNN_HIP(hipEventRecord(sd->fork[2], aux));
      PGNN_HIP(hipStreamWaitEvent(main, sd->fork[2], 0));
    }
  } else if (tr) {
    const float* tsrc[2 * kMaxTransposed];
    float* tdst[2 * kMaxTransposed];
    int64_t trows[2 * kMaxTransposed], tcols[2 * kMaxTransposed];
    for (int q = 0; q < ntr; ++q) {
      const pgnn_gin_layer& p = layers[num_layer - 1 - q];
      tsrc[2 * q] = p.w1; tdst[2 * q] = w1t[q]; trows[2 * q] = 2 * dim; tcols[2 * q] = 2 * dim;          // W1 [2D, 2D]
      tsrc[2 * q + 1] = p.w2; tdst[2 * q + 1] = w2t[q]; trows[2 * q + 1] = dim; tcols[2 * q + 1] = 2 * dim;  // W2 [D, 2D]
    }
    if (sd) {
      PGNN_HIP(hipEventRecord(sd->fork[0], main));
      PGNN_HIP(hipStreamWaitEvent(aux, sd->fork[0], 0));
    }
    if ((rc = pgnn_transpose_batch(tsrc, tdst, trows, tcols, 2 * ntr, aux))) return rc;
    if (sd) {
      PGNN_HIP(hipEventRecord(sd->fork[2], aux));
      PGNN_HIP(hipStreamWaitEvent(main, sd->fork[2], 0));
    }
  }
  const float* g = dy;
  int64_t ldg = lddy;
  for (int l = num_layer - 1; l >= 0; --l) {
    const pgnn_gin_layer& p = layers[l];
    const int q = num_layer - 1 - l, b = l & 1;
    const float* a = acts + (size_t)l * 7 * nd;
    const float *agg = a, *pre = a + 2 * nd, *hid = a + 4 * nd;
    const float* st = stats + (size_t)l * 4 * dim;
    // g = gradient of this layer's output (already masked by the ReLU that follows it, see the end of the loop body):
    // dy for the last layer, else dxb[l & 1], written by iteration l + 1
    if (wp) rc = stack_bwd_data_wp(g, ldg, wp2[l], nullptr, 0, dhid[b], 2 * dim, n, 2 * dim, dim, main);
    else if (tr && q < ntr) rc = pgnn_linear_bwd_data_t(g, ldg, w2t[q], nullptr, 0, dhid[b], 2 * dim, n, 2 * dim, dim, main);
    else rc = pgnn_linear_bwd_data(g, ldg, p.w2, nullptr, 0, dhid[b], 2 * dim, n, 2 * dim, dim, main);
    if (rc) return rc;
    // (Measured and NOT kept, profiles/r05/bio_bn_bwd_in_gemm_ab.txt: the BatchNorm backward's column sums per 16-row block out of this
    // product's epilogue -- k_gemm2pr with the ReLU mask recomputed from `pre` -- + a fold of the 641 blocks instead of
    // k_bn_bwd_partial: correct, 2.034-2.037 against 2.011-2.016 ms per step.  The pass it removes was waiting for CUs the side
    // stream's weight gradients hold; the elementwise pass behind it inherits the wait, 18 -> 40 us.)
    if ((rc = pgnn_bn_bwd(dhid[b], 2 * dim, pre, 2 * dim, p.gamma, p.beta, st, st + 2 * dim, training, 1, dpre[b], 2 * dim,
                          p.dgamma, p.dbeta, 0.f, 0, n, 2 * dim, op, opb, main))) return rc;
    // (fork[1] as the completion of this product's own dispatch where a two-plane kernel runs it: see pgnn_chem_gin_stack_bwd)
    const bool fork_via_launch = sd && wp && two_planes() && env_knob("PGNN_FORK_VIA_LAUNCH", 1) != 0 && !stream_is_capturing(main);
    if (fork_via_launch) set_next_launch_stop_event(sd->fork[1]);
    if (wp) rc = stack_bwd_data_wp(dpre[b], 2 * dim, wp1[l], nullptr, 0, dagg[b], 2 * dim, n, 2 * dim, 2 * dim, main);
    else if (tr && q < ntr) rc = pgnn_linear_bwd_data_t(dpre[b], 2 * dim, w1t[q], nullptr, 0, dagg[b], 2 * dim, n, 2 * dim, 2 * dim, main);
    else rc = pgnn_linear_bwd_data(dpre[b], 2 * dim, p.w1, nullptr, 0, dagg[b], 2 * dim, n, 2 * dim, 2 * dim, main);
    const bool taken = take_next_launch_stop_event() == nullptr;  // (always cleared here)
    if (rc) return rc;
    if (sd) {
      if (!fork_via_launch || !taken) PGNN_HIP(hipEventRecord(sd->fork[1], main));  // (not taken: another kernel ran)
      PGNN_HIP(hipStreamWaitEvent(aux, sd->fork[1], 0));
    }
    // parameter gradients (side stream when there is one): dW2 = g^T hid, dW1 = dpre^T agg, d EncT = cfeat^T dagg[:, D:]
    if ((rc = pgnn_linear_bwd_weight(g, ldg, hid, 2 * dim, p.dw2, p.db2, n, 2 * dim, dim, aux_ws, opb, aux))) return rc;
    if ((rc = pgnn_linear_bwd_weight(dpre[b], 2 * dim, agg, 2 * dim, p.dw1, p.db1, n, 2 * dim, 2 * dim, aux_ws, opb, aux))) return rc;
    // d EncT [10, dim]; with the raw encoder (emb2 set) laid out as the module's gradients: d weight [dim, 9], then d bias [dim]
    if (p.emb2) rc = rowfeat_matmul_bwd_strided(cfeat, 10, dagg[b] + dim, 2 * dim, p.demb, 1, 9, p.demb + 9 * dim, n, dim, aux_ws, opb, aux);
    else rc = pgnn_rowfeat_matmul_bwd(cfeat, 10, dagg[b] + dim, 2 * dim, p.demb, dim, n, dim, aux_ws, opb, aux);
    if (rc) return rc;
    if (sd && l >= 1) PGNN_HIP(hipEventRecord(sd->lag[b], aux));
    if (l == 0 && !dh0) break;
    // dx goes to the buffer set of its consumer, layer l - 1; that set (and its dx slot, which the side stream reads as the
    // `g` of layer l + 1's dW2) was last used by layer l + 1: wait for the side stream's layer l + 1 work
    if (sd && l + 1 <= num_layer - 1) PGNN_HIP(hipStreamWaitEvent(main, sd->lag[(l + 1) & 1], 0));
    float* dx = l == 0 ? dh0 : dxb[(l - 1) & 1];
    // the ReLU between layer l-1 and l (y_{l-1} = relu(.) was written by the forward product's epilogue) masks dx: inside the
    // tiled aggregation's store when that launch can, else as a pass of its own
    const float* yprev = l > 0 ? acts + (size_t)(l - 1) * 7 * nd + 6 * nd : nullptr;
    bool masked = false;
    if (tile_start && num_tiles)
      rc = neighbor_sum_tiled_masked(dagg[b], 2 * dim, out_ptr, out_dst, nullptr, tile_start, num_tiles, dx, dim, n, dim, nullptr, 0,
                                     nullptr, 0, nullptr, 0, yprev, dim, &masked, main);
    else
      rc = pgnn_neighbor_sum(dagg[b], 2 * dim, out_ptr, out_dst, nullptr, dx, dim, n, dim, main);
    if (rc) return rc;
    if (l > 0 && !masked) {
      hipLaunchKernelGGL(k_relu_mask, dim3((int)std::min<int64_t>(ceil_div(nd / 4, 256), 4096)), dim3(256), 0, main, dx, yprev,
                         (int64_t)(nd / 4));
    }
    g = dx;
    ldg = dim;
  }
  if (sd) {
    PGNN_HIP(hipEventRecord(sd->join, aux));
    PGNN_HIP(hipStreamWaitEvent(main, sd->join, 0));
  }
  return check_launch("bio_gin_stack_bwd");
}

}  // extern "C"

// pgnn_neighbor_sum whose launch also leaves the BatchNorm-backward column sums of the layer below, folded (aggregate.hip TAIL) --
// the step of pgnn_chem_gin_stack_bwd between two layers, as an entry of its own: `out` = dL/dy of y = relu?(BatchNorm(z)); after the
// call coef [7, dim] (inside ws: returned through *coef_out) holds what pgnn_bn_bwd's elementwise pass needs and dgamma / dbeta are
// final.  *fused = 0: shape / policy outside the tuned instance -- the plain sum ran, nothing else was written.
extern "C" int pgnn_neighbor_sum_bn_bwd(const float* x, int64_t ldx, const int32_t* out_ptr, const int32_t* out_dst, float* out, int64_t ldo,
                                        const float* z, int64_t ldz, const float* gamma, const float* beta, const float* save_mean,
                                        const float* save_invstd, int relu, int training, float* dgamma, float* dbeta, int64_t n,
                                        int64_t dim, void* ws, size_t ws_bytes, const float** coef_out, int* fused, pgnn_stream stream) {
  PGNN_REQUIRE(x && out_ptr && out_dst && out && z && gamma && beta && save_mean && save_invstd && dgamma && dbeta && ws && fused,
               "bad neighbor_sum_bn_bwd arguments");
  BnBwdTail tail{z, ldz, gamma, beta, save_mean, save_invstd, relu, training, BnBwdScratch{}, dgamma, dbeta};
  if (int r = bn_bwd_scratch(ws, ws_bytes, n, dim, &tail.scratch)) return r;
  bool f = false;
  const int rc = neighbor_sum_bn_bwd(x, ldx, out_ptr, out_dst, out, ldo, n, dim, tail, &f, (hipStream_t)stream);
  *fused = f ? 1 : 0;
  if (coef_out) *coef_out = tail.scratch.coef;
  return rc;
}
