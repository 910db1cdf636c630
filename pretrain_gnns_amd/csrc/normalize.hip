// GraphSAGE update: mean over the incoming messages, then row-wise L2 normalisation
// (chem/model.py:165-202, bio/model.py:183-224: aggr="mean" + F.normalize(aggr_out, p=2, dim=-1)).
// The sum over (x_j + e_ij) incl. the self loop is the GIN aggregation kernel; this file is the row pass
// behind it: v = sum / (deg+1) (a true division, like scatter_mean), y = v / max(||v||, 1e-12).  One wave per row, float4 lanes, xor
// butterfly for the norm.  HBM-bound: forward 2*N*D*4 bytes, backward 4*N*D*4.
#include "common.h"

using namespace pgnn;

namespace {

constexpr int kBlock = 256;             // 4 rows per block
constexpr float kNormEps = 1e-12f;      // F.normalize default

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// count_i = in-degree + 1 (the self loop), read off the CSR row pointer
template <int R>  // float4 slots per lane: dim/4 <= 64*R
__global__ void __launch_bounds__(kBlock) k_mean_l2norm_fwd(const float* __restrict__ sum, int64_t lds_,
                                                            const int32_t* __restrict__ in_ptr, float* __restrict__ y,
                                                            int64_t ldy, float* __restrict__ norm, int n, int d4) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x * (int64_t)(kBlock / 64) + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * (kBlock / 64)) {
    const float cnt = (float)(in_ptr[r + 1] - in_ptr[r] + 1);
    float4 v[R];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int c = lane + 64 * q;
      v[q] = f4_zero();
      if (c < d4) {
        const float4 t = reinterpret_cast<const float4*>(sum + r * lds_)[c];
        v[q] = make_float4(t.x / cnt, t.y / cnt, t.z / cnt, t.w / cnt);
      }
      acc += (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
    }
    const float nr = sqrtf(wave_sum(acc));
    const float inv = 1.f / fmaxf(nr, kNormEps);
    if (lane == 0) norm[r] = nr;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int c = lane + 64 * q;
      if (c < d4) reinterpret_cast<float4*>(y + r * ldy)[c] = f4_scale(v[q], inv);
    }
  }
}

// dsum = dv / count ;  dv = (dy - y (y . dy)) / ||v||   (||v|| >= eps), dy / eps otherwise (the clamp's branch)
template <int R>
__global__ void __launch_bounds__(kBlock) k_mean_l2norm_bwd(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ norm, const int32_t* __restrict__ in_ptr,
                                                            float* __restrict__ dsum, int64_t ldd, int n, int d4) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x * (int64_t)(kBlock / 64) + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * (kBlock / 64)) {
    float4 g[R], o[R];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int c = lane + 64 * q;
      g[q] = c < d4 ? reinterpret_cast<const float4*>(dy + r * lddy)[c] : f4_zero();
      o[q] = c < d4 ? reinterpret_cast<const float4*>(y + r * ldy)[c] : f4_zero();
      dot += (g[q].x * o[q].x + g[q].y * o[q].y) + (g[q].z * o[q].z + g[q].w * o[q].w);
    }
    dot = wave_sum(dot);
    const float nr = norm[r];
    const bool clamped = nr < kNormEps;
    if (clamped) dot = 0.f;
    const float k = 1.f / ((float)(in_ptr[r + 1] - in_ptr[r] + 1) * fmaxf(nr, kNormEps));
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int c = lane + 64 * q;
      if (c < d4) {
        float4 t;
        t.x = (g[q].x - o[q].x * dot) * k; t.y = (g[q].y - o[q].y * dot) * k;
        t.z = (g[q].z - o[q].z * dot) * k; t.w = (g[q].w - o[q].w * dot) * k;
        reinterpret_cast<float4*>(dsum + r * ldd)[c] = t;
      }
    }
  }
}

inline int rows_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock / 64), 16 * num_cu())); }

}  // namespace

#define PGNN_NORM_DISPATCH(R, ...)            \
  switch (R) {                                \
    case 1: { constexpr int RR = 1; __VA_ARGS__; break; } \
    case 2: { constexpr int RR = 2; __VA_ARGS__; break; } \
    case 3: { constexpr int RR = 3; __VA_ARGS__; break; } \
    default: { constexpr int RR = 4; __VA_ARGS__; break; } \
  }

extern "C" {

int pgnn_mean_l2norm_fwd(const float* sum, int64_t ld_sum, const int32_t* in_ptr, float* y, int64_t ldy, float* norm,
                         int64_t n, int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(n > 0 && dim > 0 && dim % 4 == 0 && dim <= 1024 && ld_sum % 4 == 0 && ldy % 4 == 0,
               "mean_l2norm: feature width must be a multiple of 4 in (0,1024]");
  const int d4 = (int)(dim / 4), R = (int)ceil_div(d4, 64);
  PGNN_NORM_DISPATCH(R, hipLaunchKernelGGL((k_mean_l2norm_fwd<RR>), dim3(rows_grid(n)), dim3(kBlock), 0, (hipStream_t)stream,
                                           sum, ld_sum, in_ptr, y, ldy, norm, (int)n, d4));
  return check_launch("mean_l2norm_fwd");
}

int pgnn_mean_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* norm,
                         const int32_t* in_ptr, float* dsum, int64_t ld_dsum, int64_t n, int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(n > 0 && dim > 0 && dim % 4 == 0 && dim <= 1024 && lddy % 4 == 0 && ldy % 4 == 0 && ld_dsum % 4 == 0,
               "mean_l2norm: feature width must be a multiple of 4 in (0,1024]");
  const int d4 = (int)(dim / 4), R = (int)ceil_div(d4, 64);
  PGNN_NORM_DISPATCH(R, hipLaunchKernelGGL((k_mean_l2norm_bwd<RR>), dim3(rows_grid(n)), dim3(kBlock), 0, (hipStream_t)stream,
                                           dy, lddy, y, ldy, norm, in_ptr, dsum, ld_dsum, (int)n, d4));
  return check_launch("mean_l2norm_bwd");
}

}  // extern "C"
