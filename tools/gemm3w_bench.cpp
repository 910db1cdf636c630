// Plain C++ host (no torch: starts in milliseconds on a fresh GPU box) for the products of the GIN mlp on pre-split weight
// planes (k_gemm3w, csrc/linear.hip) against the split-bf16 kernel they replace (k_gemm3): bitwise equality of every
// output (forward + ReLU + bias, forward + column statistics, backward-data + ReLU mask) and HIP-event timings, per shape and
// per PGNN_GEMM3W_CFG variant, both as a repeated launch and as the layer's chain (300->600 then 600->300).
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/gemm3w_bench.cpp -Lpretrain_gnns_amd -lpgnn -Wl,-rpath,'$ORIGIN/../../pretrain_gnns_amd' -o tools/bin/gemm3w_bench
//   tools/bin/gemm3w_bench [rows ...]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pgnn.h"

extern "C" void pgnn_reload_env(void);

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define PG(x) do { int rc_ = (x); if (rc_) { printf("pgnn error %d: %s (%s)\n", rc_, pgnn_last_error(), #x); exit(3); } } while (0)

static hipStream_t st;

static float* dev_random(size_t n, float scale, unsigned seed, bool relu = false) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    float v = ((float)(s >> 8) / 8388608.0f - 1.0f) * scale;  // uniform [-scale, scale)
    h[i] = relu ? (v > 0 ? v : 0.f) : v;
  }
  float* d;
  HIP_OK(hipMalloc(&d, n * sizeof(float) + 256));
  HIP_OK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
template <class T> static T* dev_alloc(size_t n) {
  T* d;
  HIP_OK(hipMalloc(&d, n * sizeof(T) + 256));
  HIP_OK(hipMemset(d, 0xff, n * sizeof(T)));
  return d;
}
static bool same(const float* a, const float* b, size_t n, const char* what) {
  std::vector<float> ha(n), hb(n);
  HIP_OK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, first = 0;
  double maxd = 0;
  for (size_t i = 0; i < n; ++i)
    if (memcmp(&ha[i], &hb[i], 4)) {
      if (!bad) first = i;
      ++bad;
      maxd = std::max(maxd, (double)fabsf(ha[i] - hb[i]));
    }
  if (bad) printf("  MISMATCH %s: %zu of %zu words differ (first at %zu: %g vs %g, max |d| %g)\n", what, bad, n, first, ha[first], hb[first], maxd);
  return bad == 0;
}

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
  HIP_OK(hipEventDestroy(a)); HIP_OK(hipEventDestroy(b));
  return ms * 1e3 / iters;
}

static void set_cfg(int c) {
  if (c < 0) unsetenv("PGNN_GEMM3W_CFG");
  else setenv("PGNN_GEMM3W_CFG", std::to_string(c).c_str(), 1);
  pgnn_reload_env();
}

int main(int argc, char** argv) {
  std::vector<int64_t> rows;
  for (int i = 1; i < argc; ++i) rows.push_back(atoll(argv[i]));
  if (rows.empty()) rows = {6747, 2048, 16384, 65536};
  HIP_OK(hipStreamCreate(&st));
  const int64_t D = 300;
  bool all_ok = true;
  for (int64_t m : rows) {
    const int iters = m > 100000 ? 30 : 300, warm = m > 100000 ? 5 : 50;
    // the two Linears of a GIN mlp: W1 [2D, D], W2 [D, 2D]
    float* x = dev_random((size_t)m * D, 1.f, 1);
    float* w1 = dev_random((size_t)2 * D * D, 0.06f, 2);
    float* b1 = dev_random(2 * D, 0.5f, 3);
    float* w2 = dev_random((size_t)D * 2 * D, 0.04f, 4);
    float* b2 = dev_random(D, 0.5f, 5);
    float* dz = dev_random((size_t)m * D, 1e-3f, 6);
    float *hid_a = dev_alloc<float>((size_t)m * 2 * D), *hid_b = dev_alloc<float>((size_t)m * 2 * D);
    float *z_a = dev_alloc<float>((size_t)m * D), *z_b = dev_alloc<float>((size_t)m * D);
    float *dh_a = dev_alloc<float>((size_t)m * 2 * D), *dh_b = dev_alloc<float>((size_t)m * 2 * D);
    float *da_a = dev_alloc<float>((size_t)m * D), *da_b = dev_alloc<float>((size_t)m * D);
    const size_t nblk = (size_t)((m + 15) / 16) * 2 * D;
    float *cs_a = dev_alloc<float>(nblk), *cs_b = dev_alloc<float>(nblk);
    float *w1t = dev_alloc<float>((size_t)2 * D * D), *w2t = dev_alloc<float>((size_t)2 * D * D);
    // planes: W1 [2D, D], W2 [D, 2D], W1^T [D, 2D], W2^T [2D, D]
    void* planes[4];
    const int64_t prow[4] = {2 * D, D, D, 2 * D}, pcol[4] = {D, 2 * D, 2 * D, D};
    for (int j = 0; j < 4; ++j) HIP_OK(hipMalloc(&planes[j], pgnn_weight_planes_bytes(prow[j], pcol[j])));
    const float* ssrc[4] = {w1, w2, w1, w2};
    const int64_t srow[4] = {2 * D, D, 2 * D, D}, scol[4] = {D, 2 * D, D, 2 * D};
    const int32_t str_[4] = {0, 0, 1, 1};
    PG(pgnn_split_weights(ssrc, planes, srow, scol, str_, 4, st));
    {
      const float* tsrc[2] = {w1, w2};
      float* tdst[2] = {w1t, w2t};
      const int64_t tr[2] = {2 * D, D}, tc[2] = {D, 2 * D};
      PG(pgnn_transpose_batch(tsrc, tdst, tr, tc, 2, st));
    }
    const double t_split = time_us([&] { PG(pgnn_split_weights(ssrc, planes, srow, scol, str_, 4, st)); }, 100, 10);
    printf("rows %lld   (split of 4 matrices: %.1f us)\n", (long long)m, t_split);

    // ---- reference results on the kernel in use today
    PG(pgnn_linear_fwd(x, D, w1, b1, hid_a, 2 * D, m, D, 2 * D, 1, st));
    PG(pgnn_linear_fwd_colstats(hid_a, 2 * D, w2, b2, z_a, D, m, 2 * D, D, 0, cs_a, st));
    PG(pgnn_linear_bwd_data_t(dz, D, w2t, hid_a, 2 * D, dh_a, 2 * D, m, 2 * D, D, st));
    PG(pgnn_linear_bwd_data_t(dh_a, 2 * D, w1t, nullptr, 0, da_a, D, m, D, 2 * D, st));
    const double t_f1 = time_us([&] { PG(pgnn_linear_fwd(x, D, w1, b1, hid_a, 2 * D, m, D, 2 * D, 1, st)); }, iters, warm);
    const double t_f2 = time_us([&] { PG(pgnn_linear_fwd_colstats(hid_a, 2 * D, w2, b2, z_a, D, m, 2 * D, D, 0, cs_a, st)); }, iters, warm);
    const double t_b2 = time_us([&] { PG(pgnn_linear_bwd_data_t(dz, D, w2t, hid_a, 2 * D, dh_a, 2 * D, m, 2 * D, D, st)); }, iters, warm);
    const double t_b1 = time_us([&] { PG(pgnn_linear_bwd_data_t(dh_a, 2 * D, w1t, nullptr, 0, da_a, D, m, D, 2 * D, st)); }, iters, warm);
    const double t_chain = time_us([&] {
      PG(pgnn_linear_fwd(x, D, w1, b1, hid_a, 2 * D, m, D, 2 * D, 1, st));
      PG(pgnn_linear_fwd_colstats(hid_a, 2 * D, w2, b2, z_a, D, m, 2 * D, D, 0, cs_a, st));
    }, iters, warm);
    const double gf = 2.0 * m * D * 2 * D * 1e-9;
    printf("  k_gemm3 (today)       fwd1 %7.1f us  fwd2+stats %7.1f us  bwd2+mask %7.1f us  bwd1 %7.1f us  chain(fwd1,fwd2) %7.1f us   [%.0f / %.0f TF]\n",
           t_f1, t_f2, t_b2, t_b1, t_chain, gf / t_f1 * 1e3, gf / t_f2 * 1e3);

    for (int cfg = -1; cfg <= 5; ++cfg) {
      set_cfg(cfg);
      HIP_OK(hipMemsetAsync(hid_b, 0xff, (size_t)m * 2 * D * 4, st));
      HIP_OK(hipMemsetAsync(z_b, 0xff, (size_t)m * D * 4, st));
      HIP_OK(hipMemsetAsync(dh_b, 0xff, (size_t)m * 2 * D * 4, st));
      HIP_OK(hipMemsetAsync(da_b, 0xff, (size_t)m * D * 4, st));
      HIP_OK(hipMemsetAsync(cs_b, 0xff, nblk * 4, st));
      PG(pgnn_linear_fwd_wp(x, D, planes[0], b1, hid_b, 2 * D, m, D, 2 * D, 1, nullptr, st));
      PG(pgnn_linear_fwd_wp(hid_b, 2 * D, planes[1], b2, z_b, D, m, 2 * D, D, 0, cs_b, st));
      PG(pgnn_linear_bwd_data_wp(dz, D, planes[3], hid_b, 2 * D, dh_b, 2 * D, m, 2 * D, D, st));
      PG(pgnn_linear_bwd_data_wp(dh_b, 2 * D, planes[2], nullptr, 0, da_b, D, m, D, 2 * D, st));
      HIP_OK(hipStreamSynchronize(st));
      bool ok = same(hid_a, hid_b, (size_t)m * 2 * D, "fwd1") & same(z_a, z_b, (size_t)m * D, "fwd2") & same(cs_a, cs_b, nblk, "colstats") &
                same(dh_a, dh_b, (size_t)m * 2 * D, "bwd2") & same(da_a, da_b, (size_t)m * D, "bwd1");
      all_ok &= ok;
      const double u_f1 = time_us([&] { PG(pgnn_linear_fwd_wp(x, D, planes[0], b1, hid_b, 2 * D, m, D, 2 * D, 1, nullptr, st)); }, iters, warm);
      const double u_f2 = time_us([&] { PG(pgnn_linear_fwd_wp(hid_b, 2 * D, planes[1], b2, z_b, D, m, 2 * D, D, 0, cs_b, st)); }, iters, warm);
      const double u_b2 = time_us([&] { PG(pgnn_linear_bwd_data_wp(dz, D, planes[3], hid_b, 2 * D, dh_b, 2 * D, m, 2 * D, D, st)); }, iters, warm);
      const double u_b1 = time_us([&] { PG(pgnn_linear_bwd_data_wp(dh_b, 2 * D, planes[2], nullptr, 0, da_b, D, m, D, 2 * D, st)); }, iters, warm);
      const double u_chain = time_us([&] {
        PG(pgnn_linear_fwd_wp(x, D, planes[0], b1, hid_b, 2 * D, m, D, 2 * D, 1, nullptr, st));
        PG(pgnn_linear_fwd_wp(hid_b, 2 * D, planes[1], b2, z_b, D, m, 2 * D, D, 0, cs_b, st));
      }, iters, warm);
      printf("  k_gemm3w cfg %2d %s   fwd1 %7.1f us  fwd2+stats %7.1f us  bwd2+mask %7.1f us  bwd1 %7.1f us  chain(fwd1,fwd2) %7.1f us   [%.0f / %.0f TF]\n", cfg,
             ok ? "bit-equal" : "DIFFERS  ", u_f1, u_f2, u_b2, u_b1, u_chain, gf / u_f1 * 1e3, gf / u_f2 * 1e3);
    }
    set_cfg(-1);
    if (m <= 300000) {  // in-kernel phase cycles (instrumented build), first workgroup, every wave
      uint64_t* dbg = dev_alloc<uint64_t>(8 * 8 * 8);
      for (int cfg : {1, 0, 4}) {
        for (int which = 0; which < 2; ++which) {
          HIP_OK(hipMemsetAsync(dbg, 0, 8 * 8 * 8 * 8, st));
          if (which == 0) PG(pgnn_debug_gemm3w_profile(x, D, planes[0], b1, hid_b, 2 * D, m, D, 2 * D, cfg, dbg, st));
          else PG(pgnn_debug_gemm3w_profile(hid_a, 2 * D, planes[1], b2, z_b, D, m, 2 * D, D, cfg, dbg, st));
          HIP_OK(hipStreamSynchronize(st));
          std::vector<uint64_t> h(8 * 8 * 8);
          HIP_OK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
          for (int wg : {0}) {
            printf("  phases cfg %d %s wg %d (cycles per k-step; wait issue - step | total):", cfg, which ? "fwd2" : "fwd1", wg);
            for (int w = 0; w < 8; ++w) {
              const uint64_t* o = &h[((size_t)wg * 8 + w) * 8];
              const double nk = o[5] ? (double)o[5] : 1.0;
              printf("  w%d %4.0f %4.0f %4.0f %4.0f | %5.0f", w, o[0] / nk, o[1] / nk, o[2] / nk, o[3] / nk, o[4] / nk);
            }
            printf("\n");
          }
        }
      }
      HIP_OK(hipFree(dbg));
    }
    for (float* q : {x, w1, b1, w2, b2, dz, hid_a, hid_b, z_a, z_b, dh_a, dh_b, da_a, da_b, cs_a, cs_b, w1t, w2t}) HIP_OK(hipFree(q));
    for (void* q : planes) HIP_OK(hipFree(q));
  }
  printf(all_ok ? "ALL BIT-EQUAL\n" : "SOME DIFFER\n");
  return all_ok ? 0 : 1;
}
