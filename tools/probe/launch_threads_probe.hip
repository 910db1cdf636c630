// Does enqueueing from two host threads into two streams overlap?  (round 5: the side stream's launches of the stack backward from a
// helper thread of the library -- worth building only if the runtime does not serialise the two threads' API calls)
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_threads_probe tools/probe/launch_threads_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <atomic>
__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* buf;
  hipMalloc(&buf, 1024);
  hipStream_t s[2];
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i) { hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
  const int N = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[i & 1], buf + 64 * (i & 1));
    double t1 = now();
    hipDeviceSynchronize();
    double t2 = now();
    std::atomic<int> go{0};
    auto work = [&](int id) {
      while (!go.load()) {}
      for (int i = 0; i < N / 2; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[id], buf + 64 * id);
    };
    std::thread a(work, 0), b(work, 1);
    double t3 = now();
    go.store(1);
    a.join(); b.join();
    double t4 = now();
    hipDeviceSynchronize();
    double t5 = now();
    // events: record on s0 + wait on s1, one thread
    double t6 = now();
    for (int i = 0; i < N / 4; ++i) { hipEventRecord(ev[0], s[0]); hipStreamWaitEvent(s[1], ev[0], 0); }
    double t7 = now();
    hipDeviceSynchronize();
    printf("rep %d: one thread %d launches over two streams: %.2f us per launch enqueue (%.2f incl. drain); two threads: %.2f us per launch (wall / N; %.2f incl. drain); "
           "record+wait pair %.2f us\n", rep, N, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, (t4 - t3) / N * 1e6, (t5 - t3) / N * 1e6, (t7 - t6) / (N / 4) * 1e6);
  }
  return 0;
}
