// what does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i (16-bit); every lane passes an address; the four 16-bit results
// per lane are printed for a few address patterns.  Build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int* da; short* dout;
  hipMalloc(&da, 64 * 4); hipMalloc(&dout, 256 * 2);
  for (int pat = 0; pat < 4; ++pat) {
    std::vector<int> a(64);
    const int S = pat == 3 ? 300 : 64;  // row stride in elements
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, g = l >> 4;
      if (pat == 0) a[l] = (i / 4) * S + (i % 4) * 4 + g * 4 * S;     // lane i -> row i/4, 4-column chunk i%4 of a 4x16 block; groups = k-blocks
      if (pat == 1) a[l] = i * 4 + g * 64;                               // the guide's contiguous [4][16] block per group
      if (pat == 2) a[l] = (i % 4) * S + (i / 4) * 4 + g * 4 * S;     // lane i -> row i%4, chunk i/4
      if (pat == 3) a[l] = (i / 4) * S + (i % 4) * 4 + g * 4 * S;     // pattern 0 with a 300-element row stride
    }
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
    std::vector<short> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (row stride %d)\n", pat, S);
    for (int l = 0; l < 64; ++l) printf("  lane %2d addr %5d -> %5d %5d %5d %5d\n", l, a[l], o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  }
  return 0;
}
