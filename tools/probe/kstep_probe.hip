// What does one k-step of the split-bf16 product cost, ingredient by ingredient?  One workgroup per CU runs ITERS k-steps of
// the 64x160 tile's instruction mix per wave (30 v_mfma_f32_16x16x32_bf16 on 5 accumulators, term-major; 17 ds_read_b128; the
// 44-instruction three-term split of one A fragment; 5 global_load_lds_dwordx4 from an L2-resident buffer; one s_barrier),
// switched on one at a time, with 4 waves (one per SIMD) and 8 waves (two per SIMD).  Prints shader cycles per k-step
// (s_memtime, wave 0) and wall time.     hipcc --offload-arch=gfx950 -O3 tools/probe/kstep_probe.hip -o tools/bin/kstep_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ void split3(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pack_bf16(ra, rb);
  const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  l = pack_bf16(sa, sb);
}
__device__ __forceinline__ uint64_t clk() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// bits of MODE: 1 = ds reads, 2 = split VALU, 4 = DMA, 8 = barrier, 16 = block-major MFMA order (6 terms of an accumulator
// back to back), 32 = no MFMAs at all, 64 = the TWO-plane form (two fp16 planes per operand under a scale): three products per
// accumulator instead of six (same MFMA rate as bf16), two B planes read from LDS instead of three, four DMA pieces instead of five
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k(const unsigned char* src, float* out, uint64_t* cyc, int iters) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int STAGE = 38912, S = 4;
  for (int i = threadIdx.x; i < S * STAGE / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x4 acc[5];
  for (int j = 0; j < 5; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3], b[5][3];
  for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const bf16x8*>(lds + q * 1024 + lane * 16);
  for (int j = 0; j < 5; ++j)
    for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(lds + 8192 + (j * 3 + q) * 1024 + lane * 16);
  const unsigned char* g = src + (size_t)(blockIdx.x % 64) * 65536 + wave * 5 * 1024 + lane * 16;
  const uint64_t t0 = clk();
  for (int it = 0; it < iters; ++it) {
    const unsigned char* st = lds + (it & 3) * STAGE;
    if (MODE & 8) {
      if (MODE & 64) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (MODE & 4) {
      unsigned char* dst = lds + ((it + 3) & 3) * STAGE + wave * 5 * 1024;
#pragma unroll
      for (int j = 0; j < ((MODE & 64) ? 4 : 5); ++j) __builtin_amdgcn_global_load_lds(GPTR(g + j * 1024 + (it & 7) * 8192), LPTR(dst + j * 1024), 16, 0, 0);
    }
    f32x4 lo = f32x4{1.f, 2.f, 3.f, 4.f}, hi = f32x4{5.f, 6.f, 7.f, 8.f};
    if (MODE & 1) {
      lo = *reinterpret_cast<const f32x4*>(st + (wave * 16 + (lane & 15)) * 128 + (lane >> 4) * 32);
      hi = *reinterpret_cast<const f32x4*>(st + (wave * 16 + (lane & 15)) * 128 + (lane >> 4) * 32 + 16);
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int q = 0; q < ((MODE & 64) ? 2 : 3); ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(st + 8192 + q * 10240 + ((wave & 1) * 80 + j * 16 + (lane & 15)) * 64 + (((lane >> 4) ^ ((-((lane & 15) >> 2)) & 3)) * 16));
    }
    uint32_t pl[3][4] = {};
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NT = (MODE & 64) ? 3 : 6;
    constexpr int TB[6] = {0, 1, 0, 0, 1, 2}, TA[6] = {1, 0, 0, 2, 1, 0};  // (the first three: the terms of the two-plane form)
    if (MODE & 16) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (!(MODE & 32)) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][TB[t]], a[TA[t]], acc[j], 0, 0, 0);
        }
        if ((MODE & 2) && j < 4) {
          const f32x4 v = j < 2 ? lo : hi;
          split3(v[2 * (j & 1)], v[2 * (j & 1) + 1], pl[0][j], pl[1][j], pl[2][j]);
          asm volatile("" ::"v"(pl[0][j]), "v"(pl[1][j]), "v"(pl[2][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < (NT == 3 ? 4 : 6); ++t) {  // (four rounds either way: the split of the next A fragment rides on them)
        if (!(MODE & 32) && t < NT) {
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][TB[t]], a[TA[t]], acc[j], 0, 0, 0);
        }
        if ((MODE & 2) && t < 4) {
          const f32x4 v = t < 2 ? lo : hi;
          split3(v[2 * (t & 1)], v[2 * (t & 1) + 1], pl[0][t], pl[1][t], pl[2][t]);
          asm volatile("" ::"v"(pl[0][t]), "v"(pl[1][t]), "v"(pl[2][t]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE & 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(bf16x8, uint4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
    } else if (MODE & 1) {
      asm volatile("" ::"v"(lo), "v"(hi));
    }
  }
  const uint64_t t1 = clk();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int j = 0; j < 5; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)a[0][0];
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(const char* what, const unsigned char* src, float* out, uint64_t* cyc, int threads) {
  const int iters = 2000, blocks = 256;
  const size_t lds = 4 * 38912;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads, lds>>>(src, out, cyc, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, threads, lds>>>(src, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s %d waves/SIMD: %7.0f cycles per k-step (wave 0), %6.3f us per k-step wall  [clock %.2f GHz]\n", what, threads / 256,
         (double)h[0] / iters, ms * 1e3 / iters, (double)h[0] / (ms * 1e6));
}

int main() {
  unsigned char* src;
  float* out;
  uint64_t* cyc;
  hipMalloc(&src, 64 * 65536 + 65536);
  hipMemset(src, 0x3c, 64 * 65536 + 65536);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 64);
  for (int threads : {256, 512}) {
    run<0>("30 MFMA term-major", src, out, cyc, threads);
    run<16>("30 MFMA block-major (6 dependent in a row)", src, out, cyc, threads);
    run<1>("30 MFMA + 17 ds_read_b128", src, out, cyc, threads);
    run<2>("30 MFMA + split VALU", src, out, cyc, threads);
    run<3>("30 MFMA + ds_read + split", src, out, cyc, threads);
    run<4>("30 MFMA + 5 DMA", src, out, cyc, threads);
    run<7>("30 MFMA + ds_read + split + 5 DMA", src, out, cyc, threads);
    run<15>("30 MFMA + ds_read + split + 5 DMA + barrier", src, out, cyc, threads);
    run<11>("30 MFMA + ds_read + split + barrier (no DMA)", src, out, cyc, threads);
    run<32 + 4>("5 DMA only", src, out, cyc, threads);
    run<32 + 1>("17 ds_read only", src, out, cyc, threads);
    run<32 + 2>("split VALU only", src, out, cyc, threads);
    run<32 + 15>("everything but the MFMAs", src, out, cyc, threads);
    run<64>("two planes: 15 MFMA", src, out, cyc, threads);
    run<64 + 15>("two planes: 15 MFMA + 12 ds_read + split + 4 DMA + barrier", src, out, cyc, threads);
    run<64 + 32 + 15>("two planes: everything but the MFMAs", src, out, cyc, threads);
  }
  return 0;
}
