// tile-configuration sweep of pgnn_linear_fwd_wp (k_gemm3w) over the product shapes of the chem / bio / context-prediction steps:
// per (rows, k, n) the steady-state launch-to-launch time of every PGNN_GEMM3W_CFG (0 = 128x160, 1 = 64x160 / 4 stages, 2 = 64x160 / 3)
// and of the automatic choice.   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/gemm3w_shapes.cpp -Lpretrain_gnns_amd -lpgnn
//                                      -Wl,-rpath,'$ORIGIN/../../pretrain_gnns_amd' -o tools/bin/gemm3w_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "pgnn.h"
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define PG(x) do { int rc_ = (x); if (rc_) { printf("pgnn error %d: %s\n", rc_, pgnn_last_error()); exit(3); } } while (0)
static hipStream_t st;
template <class F> static double time_us(F fn, int iters, int warm) {
  for (int i = 0; i < warm; ++i) fn();
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipEventRecord(a, st));
  for (int i = 0; i < iters; ++i) fn();
  HIP_OK(hipEventRecord(b, st));
  HIP_OK(hipEventSynchronize(b));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / iters;
}
int main() {
  HIP_OK(hipStreamCreate(&st));
  const int64_t rows[] = {1600, 2560, 3600, 5100, 6747, 8000, 10249, 13000, 16384, 24000, 32768};
  const int64_t shapes[][2] = {{300, 600}, {600, 300}, {600, 600}};
  float *x, *w, *b, *y;
  HIP_OK(hipMalloc(&x, 32768ll * 600 * 4)); HIP_OK(hipMalloc(&y, 32768ll * 600 * 4));
  HIP_OK(hipMalloc(&w, 600 * 600 * 4)); HIP_OK(hipMalloc(&b, 600 * 4));
  HIP_OK(hipMemset(x, 0x3c, 32768ll * 600 * 4)); HIP_OK(hipMemset(w, 0x3b, 600 * 600 * 4)); HIP_OK(hipMemset(b, 0, 600 * 4));
  void* planes;
  HIP_OK(hipMalloc(&planes, pgnn_weight_planes_bytes(600, 608)));
  for (auto& s : shapes) {
    const int64_t k = s[0], n = s[1];
    const float* src[1] = {w};
    void* dst[1] = {planes};
    const int64_t r[1] = {n}, c[1] = {k};
    const int32_t tr[1] = {0};
    PG(pgnn_split_weights(src, dst, r, c, tr, 1, st));
    for (int64_t m : rows) {
      printf("k %3lld n %3lld rows %6lld :", (long long)k, (long long)n, (long long)m);
      for (int cfg : {0, 1, 2, -1}) {
        if (cfg < 0) unsetenv("PGNN_GEMM3W_CFG");
        else setenv("PGNN_GEMM3W_CFG", std::to_string(cfg).c_str(), 1);
        pgnn_reload_env();
        const double t = time_us([&] { PG(pgnn_linear_fwd_wp(x, k, planes, b, y, n, m, k, n, 1, nullptr, st)); }, 200, 30);
        printf("  cfg %2d %6.1f us", cfg, t);
      }
      printf("\n");
    }
  }
  return 0;
}
