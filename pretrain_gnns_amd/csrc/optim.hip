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
constexpr int kAdamChunk = 1024;  // elements per block: one float4 per thread, so a block is ONE round trip to memory (four serial ones at 4096 made the launch 34 us for 1.9 M parameters)

struct AdamJobs {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  int count[kAdamMaxTensors];     // elements of tensor j
  int first[kAdamMaxTensors + 1];  // first block of tensor j
  int64_t state_off[kAdamMaxTensors];  // offset of tensor j in the flat exp_avg / exp_avg_sq buffers
  int n;
};

__global__ void __launch_bounds__(256) k_adam(AdamJobs jobs, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                              int64_t* step /*[32]: updates applied, -, cached beta powers (see below), arrival tickets from word 8*/, float lr, float beta1,
                                              float beta2, float eps, float weight_decay) {
  // tensor of this block: largest j with first[j] <= blockIdx.x (uniform bisection over the kernel arguments: 7 dependent
  // scalar loads; the linear scan cost the blocks of the last tensors ~50)
  int j = 0;
  {
    int hi = jobs.n;
    while (hi - j > 1) {
      const int mid = (j + hi) >> 1;
      if (jobs.first[mid] <= (int)blockIdx.x) j = mid; else hi = mid;
    }
  }
  // beta^t: two float64 pow() per wave were most of this launch once a block was a single round trip (93 us for 7 280 waves).  The
  // powers of the PREVIOUS update are kept next to the step count (words 2..6: beta1^s, beta2^s, the betas they belong to, s) and
  // advanced by one multiplication; pow() runs only when they do not match (first update, changed betas, a restored step count).
  const int64_t s0 = *step;
  const double* pw = reinterpret_cast<const double*>(step + 2);
  double b1t, b2t;
  if (s0 > 0 && step[6] == s0 && pw[2] == (double)beta1 && pw[3] == (double)beta2) {
    b1t = pw[0] * (double)beta1;
    b2t = pw[1] * (double)beta2;
  } else {
    b1t = pow((double)beta1, (double)(s0 + 1));
    b2t = pow((double)beta2, (double)(s0 + 1));
  }
  const float step_size = (float)((double)lr / (1.0 - b1t));
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - b2t));
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
  // every thread of every block has read step[0] and the cached powers before its block arrives, so the last block to arrive may
  // advance them (a separate one-thread launch for this cost 6 us of a 1.4 ms train step)
  __syncthreads();
  if (threadIdx.x == 0 && arrive_last(reinterpret_cast<unsigned*>(step + 8))) {
    double* pwo = reinterpret_cast<double*>(step + 2);
    pwo[0] = b1t;
    pwo[1] = b2t;
    pwo[2] = (double)beta1;
    pwo[3] = (double)beta2;
    step[6] = s0 + 1;
    step[0] = s0 + 1;
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
