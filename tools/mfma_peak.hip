// micro-benchmark: sustained fp32 MFMA (v_mfma_f32_16x16x4_f32) rate on this box, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 4; wpc *= 2) {   // blocks per CU (4 waves each)
    const int blocks = 256 * wpc, iters = 20000;
    k<<<blocks, 256>>>(d, 100, 0.5f); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); k<<<blocks, 256>>>(d, iters, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
      printf("blocks/CU %d: %.2f ms  %.1f TFLOP/s\n", wpc, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
