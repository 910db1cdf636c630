// PROTOTYPE, not part of libpgnn.so: the planes product of csrc/linear.hip (k_gemm3w) on TWO fp16 planes per operand
// instead of three bf16 planes -- three v_mfma_f32_16x16x32_f16 per accumulator and k-step instead of six -- with a power-of-two
// scale per ROW of each operand (a row's largest magnitude lands in [2^13, 2^14): the low plane stays out of fp16's subnormals
// for every element that matters, tools/two_plane_numerics.py).  C[m, n] = (1 / (sa[m] sb[n])) sum_k (sa[m] A[m, k]) (sb[n] W[n, k]),
// products h1 h1 + h1 h2 + h2 h1 in fp32 accumulators; the scales are exact, the epilogue's rescale too.
// This file includes the library's linear.hip for its kernel (the six-product baseline it is timed against) and helpers; the
// two-plane kernel below is k_gemm3w's one-tile-per-workgroup path with the plane count, the split and the epilogue changed.
// Checks every output against float64 on the host (componentwise error scale |a| |w|) and times both kernels with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipretrain_gnns_amd/csrc tools/probe/gemm2p_probe.hip \
//         -Lpretrain_gnns_amd -lpgnn -Wl,-rpath,'$ORIGIN/../../pretrain_gnns_amd' -o tools/bin/gemm2p_probe
//   tools/bin/gemm2p_probe [rows ...]
#include "../../pretrain_gnns_amd/csrc/linear.hip"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

namespace pgnn {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct Gemm2pArgs {
  const float* A;             // [M, lda] fp32
  int64_t lda;
  const unsigned short* Bp;   // two fp16 planes of (sb[n] W[n, :]): [2][N][ldbp], zero beyond K
  int64_t ldbp, bplane;
  float* C;
  int64_t ldc;
  int M, N, K;
  const float* a_scale;       // [M] sa: power of two per row of A
  const float* a_inv;         // [M] 1 / sa
  const float* b_inv;         // [N] 1 / sb
  int nxcd;
};

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {  // round to nearest even, a in the low half
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& h, uint32_t& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  h = __builtin_bit_cast(uint32_t, hh);
  l = pack_f16(a - (float)hh[0], b - (float)hh[1]);
}

// SELF: the row scales of A are not given: the workgroup takes the maxima of its BM rows itself, in a pass over them in front of
// the k-loop (the rows come out of L2 a second time in the k-loop; every workgroup of a row panel repeats the pass) -- no producer
// has to supply anything.  What that pass costs is what this variant measures.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool SELF = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) __attribute__((amdgpu_waves_per_eu(2, 2))) k_gemm2p(Gemm2pArgs p) {
  constexpr int BK = 32, NPL = 2;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int A_BYTES = BM * 128, B_PLANE = BN * 64, STAGE = A_BYTES + NPL * B_PLANE;
  constexpr int PA = BM / 8, PB = BN / 16;
  constexpr int NP = PA + NPL * PB, NJ = (NP + NW - 1) / NW;
  static_assert(NI >= 5 && STAGES >= 3, "tile shape");

  extern __shared__ __align__(16) unsigned char smem2p[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x, p.nxcd);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + BK - 1) / BK;

  const unsigned char* src[NJ];
  int koff[NJ], klast[NJ], kstep[NJ], ldsoff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int d = min(wave + j * NW, NP - 1);
    if (d < PA) {
      const int row = 8 * d + (lane >> 3);
      const int c = (lane & 7) ^ (((lane >> 3) & 6) | (d & 1));
      koff[j] = 16 * c;
      klast[j] = 4 * (p.K - 4);
      kstep[j] = BK * 4;
      ldsoff[j] = d * 1024;
      src[j] = reinterpret_cast<const unsigned char*>(p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda);
    } else {
      const int q = (d - PA) / PB, pb = (d - PA) % PB;
      const int row = 16 * pb + (lane >> 2);
      const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
      koff[j] = 16 * c;
      klast[j] = 2 * ((int)p.ldbp - 8);
      kstep[j] = BK * 2;
      ldsoff[j] = A_BYTES + (d - PA) * 1024;
      src[j] = reinterpret_cast<const unsigned char*>(p.Bp + q * p.bplane + (int64_t)min(n0 + row, p.N - 1) * p.ldbp);
    }
  }
  auto issue_piece = [&](int j, int stage) {
    __builtin_amdgcn_global_load_lds(PGNN_GPTR(src[j] + min(koff[j], klast[j])), PGNN_LPTR(smem2p + stage * STAGE + ldsoff[j]), 16, 0, 0);
    koff[j] += kstep[j];
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fk = lane >> 4;
  const int a_off = (wm0 + fr) * 128, a_lo = (((2 * fk) ^ ((fr & 6) | (fr >> 3))) * 16), a_hi = (((2 * fk + 1) ^ ((fr & 6) | (fr >> 3))) * 16);
  const int b_off = (wn0 + fr) * 64 + ((fk ^ ((-(fr >> 2)) & 3)) * 16);
  float sa[MI];  // the scale of this lane's A rows (the fragment's row is fr)
  __shared__ float rowmax[BM];
  constexpr int TPR = (64 * NW) / BM;  // threads per row of the pass (8 at 64 x 160 / 512 threads, 4 at 128 x 160)
  float mx = 0.f;
  if constexpr (SELF) {
    // thread t: row t / TPR, float4 columns t % TPR, + TPR, ...: eight loads in flight per round, all older than the first DMA
    const int r = min(m0 + tid / TPR, p.M - 1), k4 = p.K / 4;
    const float4* row = reinterpret_cast<const float4*>(p.A + (int64_t)r * p.lda);
    for (int c = tid % TPR; c < k4; c += 8 * TPR) {
      float4 pre[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pre[u] = row[min(c + u * TPR, k4 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(pre[u].x), fabsf(pre[u].y)), fmaxf(fabsf(pre[u].z), fabsf(pre[u].w))));
    }
#pragma unroll
    for (int off = 1; off < TPR; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (tid % TPR == 0) rowmax[tid / TPR] = mx;
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) sa[i] = p.a_scale[min(m0 + wm0 + i * 16 + fr, p.M - 1)];
  }

  auto aload = [&](int stage, f32x4 (&lo)[MI], f32x4 (&hi)[MI]) {
    const unsigned char* s = smem2p + stage * STAGE;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      lo[i] = *reinterpret_cast<const f32x4*>(s + a_off + i * 16 * 128 + a_lo);
      hi[i] = *reinterpret_cast<const f32x4*>(s + a_off + i * 16 * 128 + a_hi);
    }
  };
  f16x8 b[NI][NPL];
  auto bload = [&](int stage, int j) {
    const unsigned char* s = smem2p + stage * STAGE + A_BYTES;
#pragma unroll
    for (int q = 0; q < NPL; ++q) b[j][q] = *reinterpret_cast<const f16x8*>(s + q * B_PLANE + b_off + j * 16 * 64);
  };
  auto asplit_q = [&](int c, const f32x4 (&lo)[MI], const f32x4 (&hi)[MI], uint4 (&pl)[MI][NPL]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x4 v = c < 2 ? lo[i] : hi[i];
      uint32_t h, l;
      split2(v[2 * (c & 1)] * sa[i], v[2 * (c & 1) + 1] * sa[i], h, l);
      (&pl[i][0].x)[c] = h; (&pl[i][1].x)[c] = l;
    }
  };
  auto pin_q = [&](int c, const uint4 (&pl)[MI][NPL]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" ::"v"((&pl[i][0].x)[c]), "v"((&pl[i][1].x)[c]));
  };
  auto step = [&](auto do_issue, const f16x8 (&cur)[MI][NPL], f16x8 (&nxt)[MI][NPL], int stage, int next_stage, int issue_stage) {
    f32x4 lo[MI], hi[MI];
    uint4 pl[MI][NPL];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], cur[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][1], cur[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], cur[i][0], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j >= 1 && j <= 4) {
        asplit_q(j - 1, lo, hi, pl);
        pin_q(j - 1, pl);
      }
      if (j + 2 < NI) bload(stage, j + 2);
      else bload(next_stage, j + 2 - NI);
      if (j == 0) aload(next_stage, lo, hi);
      if constexpr (decltype(do_issue)::value) {
#pragma unroll
        for (int q = 0; q < NJ; ++q)
          if ((NJ <= NI ? q : q * NI / NJ) == j) issue_piece(q, issue_stage);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < NPL; ++q) nxt[i][q] = __builtin_bit_cast(f16x8, pl[i][q]);
  };

#pragma unroll
  for (int q = 0; q < STAGES - 1; ++q)
    if (q < nk) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) issue_piece(j, q);
    }
  auto sync = [&](int t) {
    const int infl = max(0, min(t + STAGES - 2, nk - 1) - (t + 1));
    if (STAGES >= 4 && infl == STAGES - 3) gemm_wait_vmcnt_imm<(STAGES >= 4 ? STAGES - 3 : 0) * NJ>();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
  };
  f16x8 a0[MI][NPL], a1[MI][NPL];
  float ainv[MI];
  if constexpr (SELF) {
    (void)mx;  // (the pass ran in front of the prologue's DMAs; rowmax is in LDS, the barrier of sync(0) publishes it)
  }
  sync(0);
  if constexpr (SELF) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // 2^(13 - floor(log2(max))) from the exponent field (max = 0 or subnormal: scale 1)
      const unsigned e = (__float_as_uint(rowmax[wm0 + i * 16 + fr]) >> 23) & 0xffu;
      sa[i] = e > 13u ? __uint_as_float((267u - e) << 23) : 1.f;   // 127 + 13 - (e - 127)   (tiny or zero rows: scale 1)
      ainv[i] = e > 13u ? __uint_as_float((e - 13u) << 23) : 1.f;  // 127 - 13 + (e - 127)
    }
  }
  {
    f32x4 lo[MI], hi[MI];
    aload(0, lo, hi);
    bload(0, 0);
    bload(0, 1);
    uint4 pl[MI][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c) asplit_q(c, lo, hi, pl);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < NPL; ++q) a0[i][q] = __builtin_bit_cast(f16x8, pl[i][q]);
  }
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  const int n_main = max(0, nk - (STAGES - 1));
  int it = 0;
  for (; it + 2 <= n_main; it += 2) {
    if (it > 0) sync(it);
    step(Yes{}, a0, a1, it % STAGES, (it + 1) % STAGES, (it + STAGES - 1) % STAGES);
    sync(it + 1);
    step(Yes{}, a1, a0, (it + 1) % STAGES, (it + 2) % STAGES, (it + STAGES) % STAGES);
  }
  if (it < n_main) {
    if (it > 0) sync(it);
    step(Yes{}, a0, a1, it % STAGES, (it + 1) % STAGES, (it + STAGES - 1) % STAGES);
    ++it;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < NPL; ++q) a0[i][q] = a1[i][q];
  }
  for (; it < nk; it += 2) {
    if (it > 0) sync(it);
    step(No{}, a0, a1, it % STAGES, (it + 1) % STAGES, 0);
    if (it + 1 < nk) {
      sync(it + 1);
      step(No{}, a1, a0, (it + 1) % STAGES, (it + 2) % STAGES, 0);
    }
  }
  // epilogue: lane holds C[m0 + wm0 + 16 i + fr][n0 + wn0 + 16 j + 4 fk + 0..3]; undo the two scales (exact: powers of two)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn0 + j * 16 + fk * 4;
    const float4 bi = *reinterpret_cast<const float4*>(p.b_inv + min(n, p.N - 4));
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm0 + i * 16 + fr;
      const float ai = SELF ? ainv[i] : p.a_inv[min(m, p.M - 1)];
      const float4 v = make_float4(acc[i][j][0] * (ai * bi.x), acc[i][j][1] * (ai * bi.y), acc[i][j][2] * (ai * bi.z), acc[i][j][3] * (ai * bi.w));
      if (m < p.M && n < p.N) *reinterpret_cast<float4*>(p.C + (int64_t)m * p.ldc + n) = v;
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool SELF = false>
void launch_gemm2p(const Gemm2pArgs& p, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * (BM * 128 + 2 * BN * 64);
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  (void)hipFuncSetAttribute((const void*)k_gemm2p<BM, BN, WAVES_M, WAVES_N, STAGES, SELF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_gemm2p<BM, BN, WAVES_M, WAVES_N, STAGES, SELF>), dim3(tiles), dim3(64 * WAVES_M * WAVES_N), lds, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The weight-gradient product dW [n, k] = dy^T x (+ db as the ones column) the same way: k_gemm3's row-contiguous / row-contiguous
// instantiation (operands staged through registers, transpose-read fragments, split over the contracted rows) with two fp16 planes.
// The contraction runs over the ROWS of dy and x, so the scales belong to their COLUMNS: sa[nout] for dy[:, nout], sb[kcol] for x[:, kcol].
struct Wgrad2pArgs {
  const float* A;  // dy [rows, lda]
  int64_t lda;
  const float* B;  // x [rows, ldb]
  int64_t ldb;
  float* C;        // dW [M = n_out, ldc] (or the split partials)
  int64_t ldc;
  int M, N, K;     // M = out features, N = in features, K = contracted rows
  int kchunk;
  int64_t split_stride;
  float* colsum;   // db [M] (or its partials)
  const float *a_scale, *a_inv;  // [M]
  const float *b_scale, *b_inv;  // [N + 4]: the ones column's entries are 1
  int nxcd;
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_wgrad2p(Wgrad2pArgs p) {
  constexpr int BK = 32;
  constexpr int NW = WAVES_M * WAVES_N, T = 64 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  using TA = RowMajorTile<BM>;
  using TB = RowMajorTile<BN>;
  constexpr int PA = TA::PLANE, PB = TB::PLANE;
  constexpr int UA = BM * 8, UB = BN * 8;
  constexpr int NA = (UA + T - 1) / T, NB = (UB + T - 1) / T;
  extern __shared__ __align__(16) unsigned char smemw[];
  unsigned char* const ldsA = smemw;
  unsigned char* const ldsB = smemw + 2 * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + 4 + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x, p.nxcd);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const float* srcA[NA];
  const float* srcB[NB];
  int offA[NA], offB[NB], kofA[NA], kofB[NB];
  bool okA[NA], okB[NB], oneB[NB];
  float4 scA[NA], scB[NB];  // the column scales of this thread's staging units (the same columns in every k-step)
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int u = min(tid + j * T, UA - 1);
    const int kr = u / (BM / 4), cq = u - kr * (BM / 4);
    okA[j] = m0 + 4 * cq < p.M;
    kofA[j] = kr;
    srcA[j] = p.A + (int64_t)(kbeg + kr) * p.lda + m0 + 4 * cq;
    offA[j] = TA::offset(kr, 4 * cq);
    scA[j] = okA[j] ? *reinterpret_cast<const float4*>(p.a_scale + m0 + 4 * cq) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int u = min(tid + j * T, UB - 1);
    const int kr = u / (BN / 4), cq = u - kr * (BN / 4);
    okB[j] = n0 + 4 * cq < p.N;
    oneB[j] = n0 + 4 * cq == p.N;
    kofB[j] = kr;
    srcB[j] = p.B + (int64_t)(kbeg + kr) * p.ldb + n0 + 4 * cq;
    offB[j] = TB::offset(kr, 4 * cq);
    scB[j] = (okB[j] || oneB[j]) ? *reinterpret_cast<const float4*>(p.b_scale + n0 + 4 * cq) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  float4 ra[NA], rb[NB];
  auto load_tile = [&](int it) {
    const int k0 = kbeg + it * BK;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool ok = okA[j] && k0 + kofA[j] < kend;
      const float* g = ok ? srcA[j] + (int64_t)it * BK * p.lda : reinterpret_cast<const float*>(g_zero_page);
      ra[j] = *reinterpret_cast<const float4*>(g);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool kok = k0 + kofB[j] < kend;
      const float* g = (okB[j] && kok) ? srcB[j] + (int64_t)it * BK * p.ldb
                                       : reinterpret_cast<const float*>((oneB[j] && kok) ? g_ones_page : g_zero_page);
      rb[j] = *reinterpret_cast<const float4*>(g);
    }
  };
  auto store_unit = [&](const float4& v, const float4& sc, unsigned char* base, int plane, int off) {
    uint32_t h0, l0, h1, l1;
    split2(v.x * sc.x, v.y * sc.y, h0, l0);
    split2(v.z * sc.z, v.w * sc.w, h1, l1);
    *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(base + plane + off) = make_uint2(l0, l1);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) store_unit(ra[j], scA[j], ldsA, PA, offA[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) store_unit(rb[j], scB[j], ldsB, PB, offB[j]);
  };
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fk = lane >> 4;
  auto frag = [&](auto cols_tag, const unsigned char* base, int plane, int c0, int q) -> f16x8 {
    using L = RowMajorTile<decltype(cols_tag)::value>;
    const int col = (c0 + 4 * (fr & 3) + 16 * (fk & 1)) % L::S;
    const unsigned char* a0 = base + q * plane + ((8 * fk + (fr >> 2)) * L::S + col) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * L::S * 2));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  using CA = std::integral_constant<int, BM>;
  using CB = std::integral_constant<int, BN>;
  auto compute = [&]() {
    f16x8 a[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) a[i][q] = frag(CA{}, ldsA, PA, wm0 + i * 16, q);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      f16x8 b[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) b[q] = frag(CB{}, ldsB, PB, wn0 + j * 16, q);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[0], a[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[1], a[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[0], a[i][0], acc[i][j], 0, 0, 0);
    }
  };
  if (nk > 0) {
    load_tile(0);
    store_tile();
    if (1 < nk) load_tile(1);
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    compute();
    __syncthreads();
    if (t + 1 < nk) store_tile();
    __syncthreads();
    if (t + 2 < nk) load_tile(t + 2);
  }
  float* C = p.C + (int64_t)blockIdx.y * p.split_stride;
  float* cs = p.colsum + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn0 + j * 16 + fk * 4;
    if (n > p.N) continue;
    const float4 bi = *reinterpret_cast<const float4*>(p.b_inv + n);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm0 + i * 16 + fr;
      if (m >= p.M) continue;
      const float ai = p.a_inv[m];
      if (n == p.N) cs[m] = acc[i][j][0] * ai;  // the ones column (scale 1): the bias gradient
      else *reinterpret_cast<float4*>(C + (int64_t)m * p.ldc + n) =
          make_float4(acc[i][j][0] * (ai * bi.x), acc[i][j][1] * (ai * bi.y), acc[i][j][2] * (ai * bi.z), acc[i][j][3] * (ai * bi.w));
    }
  }
}
}  // namespace
}  // namespace pgnn

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static float pow2_scale(float amax) {  // 2^k with amax * 2^k in [2^13, 2^14)
  if (!(amax > 0.f)) return 1.f;
  return exp2f(13.f - floorf(log2f(amax)));
}

template <class F>
static float time_us(F&& launch, hipStream_t st, int iters) {
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch();
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) launch();
  HIP_OK(hipEventRecord(e1, st));
  HIP_OK(hipEventSynchronize(e1));
  float ms;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

static void run_case(int M, int K, int N, float row_spread, float amp, hipStream_t st) {
  using namespace pgnn;
  std::vector<float> A((size_t)M * K), W((size_t)N * K);
  unsigned s = 12345u + M * 7 + K;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.0f - 1.0f; };
  for (int m = 0; m < M; ++m) {
    const float mag = amp * expf(row_spread * rnd());  // row magnitudes spread over exp(+-row_spread)
    for (int k = 0; k < K; ++k) A[(size_t)m * K + k] = mag * rnd();
  }
  for (auto& w : W) w = rnd() / sqrtf((float)K);
  // scales and the two fp16 planes of W on the host
  const int ldbp = (K + 31) / 32 * 32;
  std::vector<float> sa(M), sai(M), sbi(N);
  std::vector<_Float16> planes((size_t)2 * N * ldbp, (_Float16)0.f);
  for (int m = 0; m < M; ++m) {
    float amax = 0.f;
    for (int k = 0; k < K; ++k) amax = fmaxf(amax, fabsf(A[(size_t)m * K + k]));
    sa[m] = pow2_scale(amax);
    sai[m] = 1.f / sa[m];
  }
  for (int n = 0; n < N; ++n) {
    float amax = 0.f;
    for (int k = 0; k < K; ++k) amax = fmaxf(amax, fabsf(W[(size_t)n * K + k]));
    const float sb = pow2_scale(amax);
    sbi[n] = 1.f / sb;
    for (int k = 0; k < K; ++k) {
      const float y = W[(size_t)n * K + k] * sb;
      const _Float16 h = (_Float16)y;
      planes[(size_t)n * ldbp + k] = h;
      planes[(size_t)N * ldbp + (size_t)n * ldbp + k] = (_Float16)(y - (float)h);
    }
  }
  float *dA, *dW, *dC2, *dC3, *dsa, *dsai, *dsbi;
  unsigned short* dP2;
  void* dP3;
  HIP_OK(hipMalloc(&dA, A.size() * 4 + 256)); HIP_OK(hipMalloc(&dW, W.size() * 4 + 256));
  HIP_OK(hipMalloc(&dC2, (size_t)M * N * 4 + 256)); HIP_OK(hipMalloc(&dC3, (size_t)M * N * 4 + 256));
  HIP_OK(hipMalloc(&dsa, M * 4 + 256)); HIP_OK(hipMalloc(&dsai, M * 4 + 256)); HIP_OK(hipMalloc(&dsbi, N * 4 + 256));
  HIP_OK(hipMalloc(&dP2, planes.size() * 2 + 256));
  HIP_OK(hipMalloc(&dP3, pgnn_weight_planes_bytes(N, K) + 256));
  HIP_OK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsa, sa.data(), M * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsai, sai.data(), M * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsbi, sbi.data(), N * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dP2, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
  {
    const float* srcs[1] = {dW};
    void* dsts[1] = {dP3};
    int64_t rows[1] = {N}, cols[1] = {K};
    int32_t tr[1] = {0};
    if (pgnn_split_weights(srcs, dsts, rows, cols, tr, 1, st)) { printf("split failed: %s\n", pgnn_last_error()); exit(3); }
  }
  Gemm2pArgs p{};
  p.A = dA; p.lda = K; p.Bp = dP2; p.ldbp = ldbp; p.bplane = (int64_t)N * ldbp; p.C = dC2; p.ldc = N; p.M = M; p.N = N; p.K = K;
  p.a_scale = dsa; p.a_inv = dsai; p.b_inv = dsbi; p.nxcd = num_xcd();
  auto l2_64 = [&]() { launch_gemm2p<64, 160, 4, 2, 4>(p, st); };
  auto l2_128 = [&]() { launch_gemm2p<128, 160, 8, 1, 3>(p, st); };
  auto s2_64 = [&]() { launch_gemm2p<64, 160, 4, 2, 4, true>(p, st); };
  auto s2_128 = [&]() { launch_gemm2p<128, 160, 8, 1, 3, true>(p, st); };
  auto l3 = [&]() { if (pgnn_linear_fwd_wp(dA, K, dP3, nullptr, dC3, N, M, K, N, 0, nullptr, st)) { printf("fwd_wp failed: %s\n", pgnn_last_error()); exit(3); } };
  // float64 truth and the componentwise error scale on a sample of rows
  std::vector<float> C2((size_t)M * N), C3((size_t)M * N);
  auto err_of = [&](const std::vector<float>& C, double& emax, double& emean) {
    emax = 0, emean = 0;
    size_t cnt = 0;
    for (int m = 0; m < M; m += std::max(1, M / 257)) {
      for (int n = 0; n < N; ++n) {
        double t = 0, d = 0;
        for (int k = 0; k < K; ++k) {
          const double a = A[(size_t)m * K + k], w = W[(size_t)n * K + k];
          t += a * w;
          d += fabs(a * w);
        }
        const double e = fabs((double)C[(size_t)m * N + n] - t) / (d > 0 ? d : 1);
        emax = std::max(emax, e);
        emean += e;
        ++cnt;
      }
    }
    emean /= (double)cnt;
  };
  double e2max, e2mean, e3max, e3mean, e2bmax, e2bmean;
  l2_64();
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
  err_of(C2, e2max, e2mean);
  HIP_OK(hipMemsetAsync(dC2, 0xff, (size_t)M * N * 4, st));
  l2_128();
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
  err_of(C2, e2bmax, e2bmean);
  l3();
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(C3.data(), dC3, C3.size() * 4, hipMemcpyDeviceToHost));
  err_of(C3, e3max, e3mean);
  double esmax = 0, esmean = 0, esbmax = 0, esbmean = 0;
  {
    HIP_OK(hipMemsetAsync(dC2, 0xff, (size_t)M * N * 4, st));
    s2_64();
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
    err_of(C2, esmax, esmean);
    HIP_OK(hipMemsetAsync(dC2, 0xff, (size_t)M * N * 4, st));
    s2_128();
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
    err_of(C2, esbmax, esbmean);
    printf("   scales taken by the workgroup itself (a pass over its rows): 64x160 %6.1f us  128x160 %6.1f us   max err %.2e / %.2e\n",
           time_us(s2_64, st, 200), time_us(s2_128, st, 200), esmax, esbmax);
  }
  const float t2a = time_us(l2_64, st, 200), t2b = time_us(l2_128, st, 200), t3 = time_us(l3, st, 200);
  printf("M %6d K %3d N %3d rows x e^+-%.0f amp %.0e | two fp16 planes: 64x160 %6.1f us  128x160 %6.1f us   max err %.2e / %.2e mean %.2e | "
         "three bf16 planes (library's choice of tile): %6.1f us   max err %.2e mean %.2e\n",
         M, K, N, row_spread, amp, t2a, t2b, e2max, e2bmax, e2mean, t3, e3max, e3mean);
  hipFree(dA); hipFree(dW); hipFree(dC2); hipFree(dC3); hipFree(dsa); hipFree(dsai); hipFree(dsbi); hipFree(dP2); hipFree(dP3);
}

// dW [n_out, k_in] = dy^T x over `rows` rows, db = column sums of dy: the two-plane prototype (split over the rows as weight_product
// splits, folded by the library's k_splitk_reduce) against pgnn_linear_bwd_weight
static void run_wgrad(int rows, int n_out, int k_in, float col_spread, float amp, hipStream_t st) {
  using namespace pgnn;
  std::vector<float> DY((size_t)rows * n_out), X((size_t)rows * k_in);
  unsigned s = 777u + rows + n_out;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.0f - 1.0f; };
  std::vector<float> cmag(n_out);
  for (auto& c : cmag) c = amp * expf(col_spread * rnd());
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < n_out; ++c) DY[(size_t)r * n_out + c] = cmag[c] * rnd();
  for (auto& x : X) x = 2.f * rnd();
  std::vector<float> sa(n_out), sai(n_out), sb(k_in + 4, 1.f), sbi(k_in + 4, 1.f);
  for (int c = 0; c < n_out; ++c) {
    float amax = 0.f;
    for (int r = 0; r < rows; ++r) amax = fmaxf(amax, fabsf(DY[(size_t)r * n_out + c]));
    sa[c] = pow2_scale(amax);
    sai[c] = 1.f / sa[c];
  }
  for (int c = 0; c < k_in; ++c) {
    float amax = 0.f;
    for (int r = 0; r < rows; ++r) amax = fmaxf(amax, fabsf(X[(size_t)r * k_in + c]));
    sb[c] = pow2_scale(amax);
    sbi[c] = 1.f / sb[c];
  }
  const int nsplit = weight_splits(rows, k_in, n_out, 64, 160);
  int64_t chunk = ceil_div(ceil_div(rows, nsplit), 32) * 32;
  const int used = (int)ceil_div(rows, chunk);
  const int64_t stride = (int64_t)n_out * k_in + n_out;
  const size_t wsb = pgnn_linear_bwd_weight_workspace_bytes(rows, k_in, n_out);
  float *dDY, *dX, *dW2, *dB2, *dW3, *dB3, *dpart, *dsa, *dsai, *dsb, *dsbi;
  void* ws;
  HIP_OK(hipMalloc(&dDY, DY.size() * 4 + 256)); HIP_OK(hipMalloc(&dX, X.size() * 4 + 256));
  HIP_OK(hipMalloc(&dW2, (size_t)n_out * k_in * 4)); HIP_OK(hipMalloc(&dB2, n_out * 4)); HIP_OK(hipMalloc(&dW3, (size_t)n_out * k_in * 4)); HIP_OK(hipMalloc(&dB3, n_out * 4));
  HIP_OK(hipMalloc(&dpart, (size_t)used * stride * 4 + 256)); HIP_OK(hipMalloc(&ws, wsb + 256));
  HIP_OK(hipMalloc(&dsa, n_out * 4)); HIP_OK(hipMalloc(&dsai, n_out * 4)); HIP_OK(hipMalloc(&dsb, (k_in + 4) * 4)); HIP_OK(hipMalloc(&dsbi, (k_in + 4) * 4));
  HIP_OK(hipMemcpy(dDY, DY.data(), DY.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsa, sa.data(), n_out * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dsai, sai.data(), n_out * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsb, sb.data(), (k_in + 4) * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dsbi, sbi.data(), (k_in + 4) * 4, hipMemcpyHostToDevice));
  Wgrad2pArgs p{};
  p.A = dDY; p.lda = n_out; p.B = dX; p.ldb = k_in; p.M = n_out; p.N = k_in; p.K = rows; p.kchunk = (int)chunk;
  p.C = used == 1 ? dW2 : dpart; p.ldc = k_in; p.split_stride = used == 1 ? 0 : stride; p.colsum = used == 1 ? dB2 : dpart + (size_t)n_out * k_in;
  p.a_scale = dsa; p.a_inv = dsai; p.b_scale = dsb; p.b_inv = dsbi; p.nxcd = num_xcd();
  constexpr size_t lds = (size_t)2 * (RowMajorTile<64>::PLANE + RowMajorTile<160>::PLANE);
  const int tiles = (int)(ceil_div(n_out, 64) * ceil_div(k_in + 4, 160));
  auto l2 = [&]() {
    hipLaunchKernelGGL((k_wgrad2p<64, 160, 4, 2>), dim3(tiles, used), dim3(512), lds, st, p);
    if (used > 1)
      hipLaunchKernelGGL(k_splitk_reduce, dim3((int)std::min<int64_t>(ceil_div((int64_t)n_out * k_in / 4 + n_out / 4, 256), 1024)), dim3(256), 0, st, dpart,
                         used, stride, dW2, (int64_t)n_out * k_in / 4, dB2, (int64_t)n_out / 4);
  };
  auto l3 = [&]() { if (pgnn_linear_bwd_weight(dDY, n_out, dX, k_in, dW3, dB3, rows, k_in, n_out, ws, wsb + 256, st)) { printf("bwd_weight failed: %s\n", pgnn_last_error()); exit(3); } };
  l2();
  l3();
  HIP_OK(hipStreamSynchronize(st));
  std::vector<float> W2((size_t)n_out * k_in), W3((size_t)n_out * k_in), B2(n_out), B3(n_out);
  HIP_OK(hipMemcpy(W2.data(), dW2, W2.size() * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(W3.data(), dW3, W3.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(B2.data(), dB2, n_out * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(B3.data(), dB3, n_out * 4, hipMemcpyDeviceToHost));
  double e2 = 0, e3 = 0, m2 = 0, m3 = 0, eb2 = 0, eb3 = 0;
  size_t cnt = 0;
  for (int c = 0; c < n_out; c += std::max(1, n_out / 48)) {
    double tb = 0, db_ = 0;
    for (int r = 0; r < rows; ++r) { tb += DY[(size_t)r * n_out + c]; db_ += fabs(DY[(size_t)r * n_out + c]); }
    eb2 = std::max(eb2, fabs(B2[c] - tb) / db_);
    eb3 = std::max(eb3, fabs(B3[c] - tb) / db_);
    for (int k = 0; k < k_in; ++k) {
      double t = 0, d = 0;
      for (int r = 0; r < rows; ++r) {
        const double a = DY[(size_t)r * n_out + c], b = X[(size_t)r * k_in + k];
        t += a * b;
        d += fabs(a * b);
      }
      const double x2 = fabs(W2[(size_t)c * k_in + k] - t) / d, x3 = fabs(W3[(size_t)c * k_in + k] - t) / d;
      e2 = std::max(e2, x2); e3 = std::max(e3, x3); m2 += x2; m3 += x3;
      ++cnt;
    }
  }
  const float t2 = time_us(l2, st, 100), t3 = time_us(l3, st, 100);
  printf("weight gradient rows %6d dW [%3d, %3d] (%d splits) columns x e^+-%.0f amp %.0e | two fp16 planes %6.1f us  max err %.2e (db %.2e) mean %.2e | "
         "three bf16 planes %6.1f us  max err %.2e (db %.2e) mean %.2e\n",
         rows, n_out, k_in, used, col_spread, amp, t2, e2, eb2, m2 / cnt, t3, e3, eb3, m3 / cnt);
  hipFree(dDY); hipFree(dX); hipFree(dW2); hipFree(dB2); hipFree(dW3); hipFree(dB3); hipFree(dpart); hipFree(ws); hipFree(dsa); hipFree(dsai); hipFree(dsb); hipFree(dsbi);
}

int main(int argc, char** argv) {
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  std::vector<int> rows;
  for (int i = 1; i < argc; ++i) rows.push_back(atoi(argv[i]));
  if (rows.empty()) rows = {6740, 10249, 41269};
  for (int M : rows) {
    run_case(M, 300, 600, 3.f, 1.f, st);    // forward 300 -> 600, row magnitudes over e^+-3
    run_case(M, 600, 300, 0.f, 3.f, st);    // forward 600 -> 300
    run_case(M, 600, 300, 2.f, 1e-6f, st);  // backward-data shape, gradients ~1e-6
  }
  for (int M : rows) {
    if (M > 20000) continue;
    run_wgrad(M, 600, 300, 2.f, 1e-6f, st);  // dW1 of a chem layer: dhid^T agg
    run_wgrad(M, 300, 600, 2.f, 1e-6f, st);  // dW2: dz^T hid
  }
  run_wgrad(10249, 600, 600, 2.f, 1e-6f, st);  // bio dW1
  return 0;
}
