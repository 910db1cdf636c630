// Layer-level entry points: one C call enqueues every kernel of a chem GIN layer (+ its outer
// BatchNorm), forward or backward.  Host-side composition of the public per-op entry points only --
// no new device code.  Exists because at the reference's batch size (256 graphs ~ 6.8k nodes) the step
// is launch-bound: ~35 kernels per layer round trip, and a Python/ctypes/allocator round trip per
// kernel costs more than the kernel.  (chem/model.py:37-55 + :269-275 under autograd.)
#include <stdlib.h>

#include "common.h"

using namespace pgnn;

namespace {
// Side stream for the backward's independent branches.  At ~6.8k rows one GEMM only gives each CU ~1.7
// tiles, so the weight-gradient product (needs dz/dhid + saved activations) runs concurrently with the
// data-gradient product that the rest of the chain is waiting for: fork with an event after each
// producer, join once before returning.  Streams/events are created once per device and reused.
struct Side {
  hipStream_t stream = nullptr;
  hipEvent_t fork[3] = {nullptr, nullptr, nullptr};
  hipEvent_t join = nullptr;
  bool ok = false;
};
Side* side_for_current_device() {
  static Side sides[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Side& s = sides[dev];
  if (!s.ok) {
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (auto& e : s.fork)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    s.ok = true;
  }
  return &s;
}
inline bool use_side_stream() {
  const char* v = getenv("PGNN_SIDE_STREAM");
  return !v || atoi(v) != 0;
}
inline size_t op_ws_bytes(int64_t n, int64_t d) {
  size_t m = pgnn_bn_workspace_bytes(n, d);
  m = std::max(m, pgnn_bn_workspace_bytes(n, 2 * d));
  m = std::max(m, pgnn_linear_bwd_weight_workspace_bytes(n, d, 2 * d));
  m = std::max(m, pgnn_linear_bwd_weight_workspace_bytes(n, 2 * d, d));
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
                            float* save_invstd, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                            pgnn_stream stream) {
  if (ws_bytes < op_ws_bytes(n, dim)) {
    set_error("chem_gin_layer_fwd workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  int rc;
  if ((rc = pgnn_chem_aggregate_fwd(x, ldx, in_ptr, in_src, in_code, emb1, emb2, nullptr, agg, dim, n, dim, stream))) return rc;
  if ((rc = pgnn_linear_fwd(agg, dim, w1, b1, hid, 2 * dim, n, dim, 2 * dim, 1, stream))) return rc;
  if ((rc = pgnn_linear_fwd(hid, 2 * dim, w2, b2, z, dim, n, 2 * dim, dim, 0, stream))) return rc;
  return pgnn_bn_fwd(z, dim, gamma, beta, running_mean, running_var, momentum, eps, training, relu, y, dim, save_mean,
                     save_invstd, n, dim, ws, ws_bytes, stream);
}

int pgnn_chem_gin_layer_bwd(const float* dy, int64_t lddy, const float* agg, const float* hid, const float* z,
                            const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat, const float* w1,
                            const float* w2, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, int training, int relu, float* dx, float* demb /*[9,dim]*/,
                            float* dw1, float* db1, float* dw2, float* db2, float* dgamma, float* dbeta, int64_t n,
                            int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream) {
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
  Side* sd = (use_side_stream() && n <= 32768) ? side_for_current_device() : nullptr;
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
  if ((rc = pgnn_bn_bwd(dy, lddy, z, dim, gamma, beta, save_mean, save_invstd, training, relu, dz, dim, dgamma, dbeta, n,
                        dim, op, opb, main))) return rc;
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

}  // extern "C"
