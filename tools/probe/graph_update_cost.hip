// What does replaying a captured chain of small kernels with NEW launch parameters cost the host, against launching them?
// (the "unchanged script" question of VERDICT r04 item 3: an eager step pays ~80 hipLaunchKernel calls + ~13 event calls.)
// hipcc --offload-arch=gfx950 -O2 graph_update_cost.hip -o graph_update_cost && ./graph_update_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k(float* p, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.f;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int N = 80, n = 1 << 16;
  float* buf[2];
  CK(hipMalloc(&buf[0], n * 4)); CK(hipMalloc(&buf[1], n * 4));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev[2];
  CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  // eager: N launches, every 8th with a fork/join through a second stream
  auto eager = [&](int it) {
    for (int i = 0; i < N; ++i) {
      float* p = buf[(i + it) & 1];
      int nn = n - (it & 7);
      float a = 1.f;
      if (i % 8 == 7) {
        hipEventRecord(ev[0], s); hipStreamWaitEvent(s2, ev[0], 0);
        hipLaunchKernelGGL(k, dim3((nn + 255) / 256), dim3(256), 0, s2, p, nn, a);
        hipEventRecord(ev[1], s2); hipStreamWaitEvent(s, ev[1], 0);
      } else {
        hipLaunchKernelGGL(k, dim3((nn + 255) / 256), dim3(256), 0, s, p, nn, a);
      }
    }
  };
  for (int it = 0; it < 20; ++it) eager(it);
  CK(hipStreamSynchronize(s));
  double t0 = now();
  for (int it = 0; it < 200; ++it) eager(it);
  double t_enq = now() - t0;
  CK(hipStreamSynchronize(s));
  double t_all = now() - t0;
  printf("eager:  %.1f us host enqueue per %d-launch chain (%.2f us per launch incl. events), %.1f us wall\n", t_enq / 200, N, t_enq / 200 / N, t_all / 200);
  // capture once, remember the kernel nodes in launch order
  hipGraph_t g;
  std::vector<hipGraphNode_t> nodes;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) {
    float* p = buf[i & 1];
    int nn = n;
    float a = 1.f;
    hipStream_t st = s;
    if (i % 8 == 7) { hipEventRecord(ev[0], s); hipStreamWaitEvent(s2, ev[0], 0); st = s2; }
    hipLaunchKernelGGL(k, dim3((nn + 255) / 256), dim3(256), 0, st, p, nn, a);
    hipStreamCaptureStatus cs; unsigned long long id; hipGraph_t gg; const hipGraphNode_t* deps; size_t nd;
    CK(hipStreamGetCaptureInfo_v2(st, &cs, &id, &gg, &deps, &nd));
    if (nd != 1) { printf("launch %d: %zu dependencies after the launch\n", i, nd); return 1; }
    nodes.push_back(deps[0]);
    if (i % 8 == 7) { hipEventRecord(ev[1], s2); hipStreamWaitEvent(s, ev[1], 0); }
  }
  CK(hipStreamEndCapture(s, &g));
  hipGraphExec_t ex;
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  auto replay = [&](int it, bool update) {
    if (update)
      for (int i = 0; i < N; ++i) {
        float* p = buf[(i + it) & 1];
        int nn = n - (it & 7);
        float a = 1.f;
        void* args[3] = {&p, &nn, &a};
        hipKernelNodeParams kp{};
        kp.func = (void*)k; kp.gridDim = dim3((nn + 255) / 256); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
        hipGraphExecKernelNodeSetParams(ex, nodes[i], &kp);
      }
    hipGraphLaunch(ex, s);
  };
  for (int mode = 0; mode < 2; ++mode) {
    for (int it = 0; it < 20; ++it) replay(it, mode == 1);
    CK(hipStreamSynchronize(s));
    t0 = now();
    for (int it = 0; it < 200; ++it) replay(it, mode == 1);
    t_enq = now() - t0;
    CK(hipStreamSynchronize(s));
    t_all = now() - t0;
    printf("graph %s: %.1f us host per replay (%.2f us per node), %.1f us wall\n", mode ? "with every node's parameters updated" : "replayed as captured",
           t_enq / 200, t_enq / 200 / N, t_all / 200);
  }
  return 0;
}
