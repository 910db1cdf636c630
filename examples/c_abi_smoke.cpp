// A plain C++ host (no Python, no torch) driving the C ABI of include/pgnn.h: builds the CSR structure of a
// small molecule-shaped graph, runs the GIN aggregation and one Linear forward, and checks both against a
// host loop written from the reference's definitions (chem/model.py:39-52: self loops appended last with bond
// attr [4,0]; message = x_j + emb1[a0] + emb2[a1]; aggregate at edge_index[0]).
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_abi_smoke.cpp -Lpretrain_gnns_amd -lpgnn \
//         -Wl,-rpath,$PWD/pretrain_gnns_amd -o c_abi_smoke && ./c_abi_smoke
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "pgnn.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define PGNN_OK_(x) do { int rc_ = (x); if (rc_) { printf("pgnn error %d: %s (%s)\n", rc_, pgnn_last_error(), #x); return 3; } } while (0)

template <class T> static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
  hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

int main() {
  const int64_t N = 777, D = 300;
  srand(7);
  // ring of N atoms + a few chords; both directions of a bond adjacent, identical attributes (chem/loader.py:83-86)
  std::vector<int64_t> src, dst, a0, a1;
  auto bond = [&](int64_t u, int64_t v) {
    const int64_t t = rand() % 4, d = rand() % 3;
    src.push_back(v); dst.push_back(u); a0.push_back(t); a1.push_back(d);
    src.push_back(u); dst.push_back(v); a0.push_back(t); a1.push_back(d);
  };
  for (int64_t i = 0; i < N; ++i) bond(i, (i + 1) % N);
  for (int64_t i = 0; i < N; i += 9) bond(i, (i + 5) % N);
  const int64_t E = (int64_t)src.size();
  std::vector<int64_t> ei(2 * E), ea(2 * E);
  for (int64_t e = 0; e < E; ++e) { ei[e] = dst[e]; ei[E + e] = src[e]; ea[2 * e] = a0[e]; ea[2 * e + 1] = a1[e]; }
  std::vector<float> x(N * D), emb1(6 * D), emb2(3 * D), w(2 * D * D), b(2 * D);
  for (auto& v : x) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : emb1) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : emb2) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : b) v = (float)rand() / RAND_MAX - 0.5f;

  if (pgnn_abi_version() != PGNN_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int64_t *d_ei = to_device(ei), *d_ea = to_device(ea);
  float *d_x = to_device(x), *d_e1 = to_device(emb1), *d_e2 = to_device(emb2), *d_w = to_device(w), *d_b = to_device(b);
  int32_t *in_ptr, *in_src, *out_ptr, *out_dst, *status;
  uint8_t* in_code;
  float *dinv, *cfeat, *agg, *hid;
  void* ws;
  const size_t ws_bytes = pgnn_graph_workspace_bytes(N, E);
  HIP_OK(hipMalloc(&in_ptr, (N + 1) * 4)); HIP_OK(hipMalloc(&out_ptr, (N + 1) * 4));
  HIP_OK(hipMalloc(&in_src, E * 4)); HIP_OK(hipMalloc(&out_dst, E * 4)); HIP_OK(hipMalloc(&in_code, E));
  HIP_OK(hipMalloc(&dinv, N * 4)); HIP_OK(hipMalloc(&cfeat, N * 9 * 4)); HIP_OK(hipMalloc(&status, 4));
  HIP_OK(hipMalloc(&agg, N * D * 4)); HIP_OK(hipMalloc(&hid, N * 2 * D * 4)); HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMemsetAsync(status, 0, 4, st));

  PGNN_OK_(pgnn_chem_graph_build(d_ei, d_ea, E, N, 0, in_ptr, in_src, in_code, out_ptr, out_dst, dinv, cfeat, status, ws, ws_bytes, st));
  PGNN_OK_(pgnn_chem_aggregate_fwd(d_x, D, in_ptr, in_src, in_code, d_e1, d_e2, nullptr, agg, D, N, D, st));
  PGNN_OK_(pgnn_linear_fwd(agg, D, d_w, d_b, hid, 2 * D, N, D, 2 * D, 1, st));
  HIP_OK(hipStreamSynchronize(st));

  std::vector<float> h_agg(N * D), h_hid(N * 2 * D);
  int32_t h_status = -1;
  HIP_OK(hipMemcpy(h_agg.data(), agg, h_agg.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_hid.data(), hid, h_hid.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&h_status, status, 4, hipMemcpyDeviceToHost));

  // host reference: sequential scatter_add in edge order, self loops last
  std::vector<float> ref(N * D, 0.f);
  for (int64_t e = 0; e < E; ++e)
    for (int64_t d = 0; d < D; ++d)
      ref[dst[e] * D + d] += x[src[e] * D + d] + (emb1[a0[e] * D + d] + emb2[a1[e] * D + d]);
  for (int64_t i = 0; i < N; ++i)
    for (int64_t d = 0; d < D; ++d) ref[i * D + d] += x[i * D + d] + (emb1[4 * D + d] + emb2[0 * D + d]);
  int64_t bit_diff = 0;
  for (size_t q = 0; q < ref.size(); ++q) bit_diff += ref[q] != h_agg[q];
  double max_err = 0;
  for (int64_t i = 0; i < N; i += 37)
    for (int64_t o = 0; o < 2 * D; ++o) {
      double acc = b[o];
      for (int64_t d = 0; d < D; ++d) acc += (double)ref[i * D + d] * w[o * D + d];
      if (acc < 0) acc = 0;
      max_err = fmax(max_err, fabs(acc - h_hid[i * 2 * D + o]));
    }
  printf("c_abi_smoke: N=%lld E=%lld status=%d aggregation elements differing from the host scatter_add: %lld, "
         "linear max |err| = %.3g\n", (long long)N, (long long)E, h_status, (long long)bit_diff, max_err);
  return (h_status == 0 && bit_diff == 0 && max_err < 1e-4) ? 0 : 4;
}
