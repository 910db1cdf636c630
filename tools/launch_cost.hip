// host-side cost of the HIP runtime calls the library makes per step (build: hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
struct Big { char pad[120]; int* p; };
__global__ void k_bigargs(Big b) { if (b.p && threadIdx.x == 9999) *b.p = 1; }
template <class F> double us_per(F f, int n, hipStream_t st) {
  for (int i = 0; i < 20; ++i) f();
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
  int* d; hipMalloc(&d, 1 << 20);
  const int n = 2000;
  printf("kernel launch (1 arg)          %.2f us\n", us_per([&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, a, d); }, n, a));
  Big big{}; big.p = d;
  printf("kernel launch (128 B of args)  %.2f us\n", us_per([&] { hipLaunchKernelGGL(k_bigargs, dim3(64), dim3(256), 0, a, big); }, n, a));
  printf("kernel launch + 72 KB dyn LDS  %.2f us\n", us_per([&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 60000, a, d); }, n, a));
  printf("hipMemsetAsync 4 B             %.2f us\n", us_per([&] { hipMemsetAsync(d, 0, 4, a); }, n, a));
  printf("hipMemsetAsync 64 KB           %.2f us\n", us_per([&] { hipMemsetAsync(d, 0, 65536, a); }, n, a));
  printf("event record + stream wait     %.2f us\n", us_per([&] { hipEventRecord(e, a); hipStreamWaitEvent(b, e, 0); }, n, a));
  printf("hipFuncSetAttribute            %.2f us\n", us_per([&] { hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 100000); }, n, a));
  printf("hipGetLastError                %.2f us\n", us_per([&] { (void)hipGetLastError(); }, n, a));
  printf("hipGetDevice                   %.2f us\n", us_per([&] { int dv; hipGetDevice(&dv); }, n, a));
  printf("getenv (unset name)            %.2f us\n", us_per([&] { volatile const char* v = getenv("PGNN_NOT_SET_ANYWHERE"); (void)v; }, n, a));
  hipStreamSynchronize(b);
  return 0;
}
