// v_permlane32_swap / v_permlane16_swap on gfx950: what lands where (the register shuffle k_gemm2pr uses to turn two row-contiguous
// 64-byte fetches per row into MFMA fragments).  hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned g = threadIdx.x >> 4;  // 16-lane group
  unsigned x = g, y = 4 + g;            // x: groups hold chunks 0 1 2 3; y: 4 5 6 7
  u2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
  u2 q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
  out[128 + threadIdx.x] = q[0];
  out[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d;
  unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"after permlane32_swap: x", "                      y", "then permlane16_swap:  x", "                      y"};
  for (int r = 0; r < 4; ++r) {
    printf("%s = groups", names[r]);
    for (int g = 0; g < 4; ++g) printf(" %u", h[64 * r + 16 * g]);
    bool uniform = true;
    for (int l = 0; l < 64; ++l) uniform &= h[64 * r + l] == h[64 * r + (l & ~15)];
    printf("  (%s within groups)\n", uniform ? "uniform" : "NOT uniform");
  }
  return 0;
}
