// Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient, no amsgrad) over a list of tensors in ONE
// launch.  The reference builds three `optim.Adam` objects (chem/pretrain_masking.py:134-136) and steps them one after the
// other; torch's fused path is one multi-tensor launch per optimizer plus ~0.3 ms of Python per step, more than a tenth
// of the 256-graph train step.  Here the tensors of all three are one job table in the kernel arguments, the step count
// lives on the device (so the launch can be captured in a HIP graph; the last block to arrive advances it), and the update is the reference formula in fp32:
//     g += wd p ; m += (g - m)(1 - b1) ; v = b2 v + (1 - b2) g g ; p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace pgnn {
namespace {

constexpr int kAdamMaxTensors = 96;
constexpr int kAdamChunk = 4096;  // elements per block

struct AdamJobs {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  int count[kAdamMaxTensors];     // elements of tensor j
  int first[kAdamMaxTensors + 1];  // first block of tensor j
  int64_t state_off[kAdamMaxTensors];  // offset of tensor j in the flat exp_avg / exp_avg_sq buffers
  int n;
};

__global__ void __launch_bounds__(256) k_adam(AdamJobs jobs, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                              int64_t* step /*[2]: updates applied, arrival ticket*/, float lr, float beta1,
                                              float beta2, float eps, float weight_decay) {
  // tensor of this block: the table is tiny, a linear scan by one lane is cheaper than anything clever
  __shared__ int js;
  if (threadIdx.x == 0) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first[j + 1]) ++j;
    js = j;
  }
  __syncthreads();
  const int j = js;
  const double t = (double)(*step + 1);
  const float step_size = (float)((double)lr / (1.0 - pow((double)beta1, t)));
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
  const int base = ((int)blockIdx.x - jobs.first[j]) * kAdamChunk;
  const int end = min(jobs.count[j], base + kAdamChunk);
  float* __restrict__ p = jobs.p[j];
  const float* __restrict__ g = jobs.g[j];
  float* __restrict__ m = exp_avg + jobs.state_off[j];
  float* __restrict__ v = exp_avg_sq + jobs.state_off[j];
  auto update = [&](float pi, float gi, float& mi, float& vi) -> float {
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
    mi = fmaf(gi - mi, 1.f - beta1, mi);
    vi = fmaf(1.f - beta2, gi * gi, beta2 * vi);
    return pi - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
  };
  // float4 when the tensor's four arrays allow it: the update is pure streaming
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0 && (jobs.count[j] & 3) == 0;
  if (vec) {
    for (int i = base + 4 * threadIdx.x; i < end; i += 4 * 256) {
      float4 pv = *reinterpret_cast<const float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
      pv.x = update(pv.x, gv.x, mv.x, vv.x);
      pv.y = update(pv.y, gv.y, mv.y, vv.y);
      pv.z = update(pv.z, gv.z, mv.z, vv.z);
      pv.w = update(pv.w, gv.w, mv.w, vv.w);
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
      *reinterpret_cast<float4*>(p + i) = pv;
    }
  } else {
    for (int i = base + threadIdx.x; i < end; i += 256) {
      float mi = m[i], vi = v[i];
      const float pn = update(p[i], g[i], mi, vi);
      m[i] = mi;
      v[i] = vi;
      p[i] = pn;
    }
  }
  // every thread of every block has read step[0] before its block takes a ticket, so the holder of the last ticket may advance it
  // (a separate one-thread launch for this cost 6 us of a 1.4 ms train step)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(step + 1);
    if (atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1ull) {
      step[0] += 1;
      *ticket = 0ull;
    }
  }
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

int pgnn_adam_max_tensors(void) { return kAdamMaxTensors; }

int pgnn_adam_step(float* const* params, const float* const* grads, const int64_t* counts, const int64_t* state_offsets, int64_t n,
                   float* exp_avg, float* exp_avg_sq, int64_t* step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   pgnn_stream stream) {
  PGNN_REQUIRE(n > 0 && n <= kAdamMaxTensors, "adam_step: 1..%d tensors per call", kAdamMaxTensors);
  AdamJobs jobs{};
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    PGNN_REQUIRE(params[j] && grads[j] && counts[j] > 0 && counts[j] < (1ll << 31) - kAdamChunk, "adam_step: bad tensor %d", j);
    jobs.p[j] = params[j];
    jobs.g[j] = grads[j];
    jobs.count[j] = (int)counts[j];
    jobs.state_off[j] = state_offsets[j];
    jobs.first[j] = blocks;
    blocks += (int)ceil_div(counts[j], kAdamChunk);
  }
  jobs.first[n] = blocks;
  jobs.n = (int)n;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, jobs, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay);
  return check_launch("adam_step");
}

}  // extern "C"
