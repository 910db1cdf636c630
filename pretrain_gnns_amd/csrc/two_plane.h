// fp32 operands on TWO fp16 planes under a power-of-two scale per row (gfx950): the pieces the products on planes share --
// csrc/linear.hip (k_gemm2pw, k_gemm2pr, k_split2p_jobs) and csrc/mlp_fused.hip (k_mlp2p_fused).  See the block comment in front of
// k_split2p_jobs for the format and DESIGN 3.4c for its error against float64.
#pragma once
#include "common.h"

namespace pgnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define PGNN_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PGNN_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt_imm() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {  // round to nearest even, a in the low half
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
}
// the low plane's share of (a, b).  Clamped to fp16's range: a - h is at most half an fp16 ulp for every finite h, so the clamp only
// catches h = +-inf, where a - h is NaN -- and the product would turn an fp32 +-inf into NaN.  (fminf / fmaxf drop a NaN operand.)
__device__ __forceinline__ float low_part(float a, _Float16 h) { return fminf(fmaxf(a - (float)h, -65504.f), 65504.f); }
__device__ __forceinline__ void split2(float a, float b, uint32_t& h, uint32_t& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  h = __builtin_bit_cast(uint32_t, hh);
  l = pack_f16(low_part(a, hh[0]), low_part(b, hh[1]));
}
// s = 2^(13 - floor(log2(amax))) and its inverse from amax's exponent field: s amax lands in [2^13, 2^14).  Rows below 2^-113
// (exponent field <= 13: the scale would leave fp32's range) take 2^127, whose inverse is the subnormal 2^-127 -- they keep
// 11 + 11 bits down to fp32's smallest subnormals; an all-zero row is one of them.  Exponent field 255 (the row holds an inf: fmaxf
// drops a NaN operand, so a NaN alone never reaches the maximum): unscaled, the non-finite value propagates through the high plane.
__device__ __forceinline__ void pow2_scales(float amax, float& s, float& inv) {
  const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;
  if (e == 255u) {
    s = 1.f;
    inv = 1.f;
  } else if (e <= 13u) {
    s = __uint_as_float(0x7F000000u);    // 2^127
    inv = __uint_as_float(0x00400000u);  // 2^-127
  } else {
    s = __uint_as_float((267u - e) << 23);
    inv = __uint_as_float((e - 13u) << 23);
  }
}

}  // namespace
}  // namespace pgnn
