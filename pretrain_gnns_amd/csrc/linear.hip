// fp32 MFMA GEMMs for the GIN mlp / GCN linear (gfx950): forward, backward-data, backward-weight.
//
// The reference reaches cuBLAS sgemm through nn.Linear (chem/model.py:29,63; bio/model.py:24,67).
// fp32 parity (1e-4 on node embeddings after 5 BatchNorm'ed layers) rules out bf16/fp8, and gfx950
// has no xf32, so the contraction runs on v_mfma_f32_16x16x4_f32: exact fp32 FMA chains at the
// fp32-vector peak (157 TFLOP/s), with the VALU left free for staging and epilogues.
//
// One kernel template covers the three products; they differ only in which global dimension is
// contiguous for each operand:
//    C[m,n] = sum_k Aop(m,k) * Bop(n,k)
//    forward      y  = x . W^T   A = x  (k contiguous)   B = W  (k contiguous)
//    bwd data     dx = dy . W    A = dy (k contiguous)   B = W  (n contiguous)
//    bwd weight   dW = dy^T . x  A = dy (m contiguous)   B = x  (n contiguous), split over k (rows)
// LDS tiles are stored k-major ([BK][BM+pad]) so an MFMA A/B fragment (lane l: row l&15, k l>>4)
// is one conflict-light ds_read_b32; global->LDS staging goes through registers (float4 loads,
// transposing 4 x ds_write_b32 for k-contiguous operands, ds_write_b128 for the others) and is
// double-buffered: tile t+1's global loads are issued before tile t's MFMAs, one barrier per tile.
// The block->tile map is XCD-aware (tiles sharing an A row-panel run on one XCD's L2).
#include <stdlib.h>

#include "common.h"

namespace pgnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  int M, N, K;
  const float* bias;   // EPI_BIAS: [N]
  int relu;            // EPI_BIAS: apply max(.,0)
  const float* mask;   // EPI_MASK: [M, ldmask], C *= (mask > 0)
  int64_t ldmask;
  int kchunk;          // split-K: k range per blockIdx.y
  int64_t split_stride;  // split-K: floats between partial C matrices
  float* colsum;       // ONES: receives sum_k Aop(m,k) (one value per m), same split stride
  int nxcd;            // XCDs of the device (block -> tile remap)
};

enum { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_MASK = 2 };

__device__ float4 g_zero_page[4];  // zero-initialised: DMA source for lanes past the K / M / N edge
__device__ float4 g_ones_page[1] = {{1.f, 0.f, 0.f, 0.f}};  // DMA source of the "ones column" (bias gradient)

#define PGNN_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PGNN_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// C[m,n] = sum_k Aop(m,k) * Bop(n,k);  N and ldc must be multiples of 4 (float4 epilogue).
//
// Staging: each 16-deep k-step's A and B tiles go HBM -> LDS by global_load_lds (1 KiB per wave
// instruction, no VGPR round trip, no ds_write), double-buffered: the DMA of step t+1 is issued
// before the MFMAs of step t and is drained by the vmcnt(0)+barrier that ends the step.  LDS images
// are un-padded (a DMA writes lane-linear):  k-contiguous operands as [row][16], read back as ONE
// ds_read_b128 per fragment (4 k's per lane; MFMA r of a step then covers k = 4*(lane>>4)+r, the
// same permutation on both operands);  row-contiguous operands as [16][rows], read as ds_read_b32
// with that same k mapping.  Lanes beyond an edge fetch from a zero page instead of being masked,
// so stale LDS contents can never leak into the accumulators.
// The MFMA is issued with the operands swapped (D = Bfrag x Afrag), so a lane ends up holding FOUR
// CONSECUTIVE COLUMNS of one C row: the epilogue is one float4 store (+ float4 bias / mask load) per
// 16x16 block.
// ONES (weight-gradient product only): the B tile gets one extra column n == N whose entries are 1,
// fetched from a constant page, so C[m][N] = sum_k Aop(m,k) -- the bias gradient -- falls out of the
// same MFMAs and lands in p.colsum instead of needing its own column-sum kernels.
//
// STAGES-deep LDS ring: the DMA of step t+STAGES-1 is issued while step t is computed and each
// step waits only for ITS OWN pieces with a counted s_waitcnt vmcnt (every vector-memory instruction
// in the k-loop is one of this wave's DMA pieces, so the count is exact), then one raw s_barrier.
// With ~1-2 waves per SIMD (a 256-graph batch is only ~1.7 tiles per CU) this is what hides the HBM
// latency of a k-step; a plain __syncthreads() would drain vmcnt to 0 and serialise load and MFMA.
__device__ __forceinline__ void gemm_wait_vmcnt(int n) {
#define PGNN_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    PGNN_W(1) PGNN_W(2) PGNN_W(3) PGNN_W(4) PGNN_W(5) PGNN_W(6) PGNN_W(7) PGNN_W(8) PGNN_W(9) PGNN_W(10) PGNN_W(11)
    PGNN_W(12) PGNN_W(13) PGNN_W(14) PGNN_W(15) PGNN_W(16) PGNN_W(17) PGNN_W(18) PGNN_W(19) PGNN_W(20) PGNN_W(21)
    PGNN_W(22) PGNN_W(23) PGNN_W(24) PGNN_W(25) PGNN_W(26) PGNN_W(27) PGNN_W(28)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef PGNN_W
}

// KSS encodes the k-step depth and the ring depth: 1 / 2 = KS images per barrier with a 2-stage ring; 11 / 12 = one image per
// barrier with a 3- / 4-stage ring (PGNN_GEMM_KS selects; measured in tools/gemm_bench.py)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES, int KSS>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_gemm(GemmArgs p) {
  constexpr int BK = 16;       // depth of one LDS image (= 4 MFMA k-steps)
  constexpr int KS = KSS >= 10 ? 1 : KSS;
  constexpr int STAGES = KSS >= 10 ? KSS - 8 : 2;    // LDS ring; each stage holds KS images = 16*KS of k per barrier
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % 16 == 0 && BN % 16 == 0, "tile shape");
  constexpr int PA = BM / 16, PB = BN / 16;        // 1-KiB DMA pieces per image
  constexpr int NP = PA + PB, NJ = (KS * NP + NW - 1) / NW;  // pieces per wave and k-step
  constexpr int TILE = (BM + BN) * BK;              // floats per image

  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + (ONES ? 4 : 0) + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x, p.nxcd);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk16 = (kend - kbeg + BK - 1) / BK;  // images to consume
  const int nk = (nk16 + KS - 1) / KS;           // k-steps (barriers)

  // ---- per-wave DMA pieces: piece d < PA is float4 [64d, 64d+64) of the A tile, else of the B tile
  const float* src[NJ];   // this lane's source at k-step 0
  int kofs[NJ];           // k index (relative to the k-step start) this lane covers
  int64_t kstride[NJ];    // floats to advance per k-step
  bool rowok[NJ], ones[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int dd = wave + j * NW;
    src[j] = nullptr; kofs[j] = 0; kstride[j] = 0; rowok[j] = false; ones[j] = false;
    if (dd < KS * NP) {
      const int sub = dd / NP, d = dd % NP;  // image inside the k-step, piece inside the image
      const bool isA = d < PA;
      const int idx = (isA ? d : d - PA) * 64 + lane;  // float4 index inside the tile
      const float* base = isA ? p.A : p.B;
      const int64_t ld = isA ? p.lda : p.ldb;
      const int r0 = isA ? m0 : n0, rmax = isA ? p.M : p.N;
      const bool kmajor = isA ? A_KMAJOR : B_KMAJOR;
      const int rows = isA ? BM : BN;
      if (kmajor) {
        // XOR swizzle of the four 16-byte k-chunks of a row (chunk kq of row r lives at kq ^ ((-(r>>2))&3)):
        // the 16 lanes of every ds_read_b128 group then hit 16 distinct 16-byte bank groups (conflict-free)
        const int row = idx >> 2, kq = (idx & 3) ^ ((-(row >> 2)) & 3);
        rowok[j] = r0 + row < rmax;
        kofs[j] = sub * BK + 4 * kq;
        src[j] = base + (int64_t)(r0 + row) * ld + kbeg + kofs[j];
        kstride[j] = BK * KS;
      } else {
        // (rotating k-rows 4..7 / 12..15 by 16 columns makes these ds_read_b32 fragments conflict-free as well,
        // but measured 2 % slower on backward-data: the extra index arithmetic costs more than the conflicts)
        const int kr = idx / (rows / 4), rq = idx % (rows / 4);
        rowok[j] = r0 + 4 * rq < rmax;
        ones[j] = ONES && !isA && r0 + 4 * rq == rmax;
        kofs[j] = sub * BK + kr;
        src[j] = base + (int64_t)(kbeg + kofs[j]) * ld + r0 + 4 * rq;
        kstride[j] = BK * KS * ld;
      }
    }
  }
  auto issue = [&](int stage, int it) {
    float* st = smem + stage * (KS * TILE);
    const int k0 = kbeg + it * (BK * KS);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int d = wave + j * NW;
      if (d < KS * NP) {
        const bool kok = k0 + kofs[j] < kend;
        const float* g = (rowok[j] && kok) ? src[j] + (int64_t)it * kstride[j]
                                           : reinterpret_cast<const float*>((ONES && ones[j] && kok) ? g_ones_page : g_zero_page);
        __builtin_amdgcn_global_load_lds(PGNN_GPTR(g), PGNN_LPTR(st + d * 256), 16, 0, 0);
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fk = lane >> 4;

  int npw = 0;  // DMA pieces this wave issues per k-step
#pragma unroll
  for (int j = 0; j < NJ; ++j) npw += (wave + j * NW < KS * NP) ? 1 : 0;

  auto load_frags = [&](int stage, int sub, f32x4 (&a)[MI], f32x4 (&b)[NI]) {
    const float* At = smem + (stage * KS + sub) * TILE;
    const float* Bt = At + BM * BK;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = wm0 + i * 16 + fr;
      if (A_KMAJOR) {
        a[i] = *reinterpret_cast<const f32x4*>(At + m * BK + ((fk ^ ((-(fr >> 2)) & 3)) * 4));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[i][r] = At[(fk * 4 + r) * BM + m];
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = wn0 + j * 16 + fr;
      if (B_KMAJOR) {
        b[j] = *reinterpret_cast<const f32x4*>(Bt + n * BK + ((fk ^ ((-(fr >> 2)) & 3)) * 4));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) b[j][r] = Bt[(fk * 4 + r) * BN + n];
      }
    }
  };
  auto mfma_step = [&](const f32x4 (&a)[MI], const f32x4 (&b)[NI]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][r], a[i][r], acc[i][j], 0, 0, 0);
  };

  {
#pragma unroll
    for (int q = 0; q < STAGES - 1; ++q)
      if (q < nk) issue(q, q);
    for (int it = 0; it < nk; ++it) {
      const int stage = it % STAGES;
      gemm_wait_vmcnt(min(STAGES - 2, nk - 1 - it) * npw);  // step `it` has landed; younger steps stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave's pieces of step `it` are in LDS; buffer (it-1)%STAGES is free
      if (it + STAGES - 1 < nk) issue((it + STAGES - 1) % STAGES, it + STAGES - 1);
#pragma unroll
      for (int sub = 0; sub < KS; ++sub) {
        if (sub > 0 && it * KS + sub >= nk16) break;  // K tail: the last k-step may hold fewer images
        f32x4 a[MI], b[NI];
        load_frags(stage, sub, a, b);
        mfma_step(a, b);
      }
    }
  }

  // epilogue: lane holds C[m = .. + (lane & 15)][n = .. + (lane >> 4) * 4 + 0..3] of each 16x16 block
  float* C = p.C + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn0 + j * 16 + fk * 4;
    if (ONES && n == p.N) {  // the ones column: per-row sums of Aop
      float* cs = p.colsum + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm0 + i * 16 + fr;
        if (m < p.M) cs[m] = acc[i][j][0];
      }
      continue;
    }
    if (n >= p.N) continue;
    float4 bv = f4_zero();
    if (EPI == EPI_BIAS && p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm0 + i * 16 + fr;
      if (m >= p.M) continue;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (EPI == EPI_BIAS) {
        v = f4_add(v, bv);
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      if (EPI == EPI_MASK) {
        const float4 mk = *reinterpret_cast<const float4*>(p.mask + (int64_t)m * p.ldmask + n);
        if (!(mk.x > 0.f)) v.x = 0.f;
        if (!(mk.y > 0.f)) v.y = 0.f;
        if (!(mk.z > 0.f)) v.z = 0.f;
        if (!(mk.w > 0.f)) v.w = 0.f;
      }
      *reinterpret_cast<float4*>(C + (int64_t)m * p.ldc + n) = v;
    }
  }
}

// dst[i] = sum_z partial[z][i]  (fixed order), float4
// partial matrices are [nsplit][n4a + n4b] float4: the first n4a go to dst_a (dW), the rest to dst_b (db)
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ partial, int nsplit,
                                                       int64_t stride, float* __restrict__ dst_a, int64_t n4a,
                                                       float* __restrict__ dst_b, int64_t n4b) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4a + n4b;
       q += (int64_t)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(partial)[q];
    int z = 1;
    for (; z + 4 <= nsplit; z += 4) {  // four independent loads in flight, added in split order
      const float4 v0 = reinterpret_cast<const float4*>(partial + (z + 0) * stride)[q];
      const float4 v1 = reinterpret_cast<const float4*>(partial + (z + 1) * stride)[q];
      const float4 v2 = reinterpret_cast<const float4*>(partial + (z + 2) * stride)[q];
      const float4 v3 = reinterpret_cast<const float4*>(partial + (z + 3) * stride)[q];
      s = f4_add(f4_add(f4_add(f4_add(s, v0), v1), v2), v3);
    }
    for (; z < nsplit; ++z) s = f4_add(s, reinterpret_cast<const float4*>(partial + z * stride)[q]);
    if (q < n4a) reinterpret_cast<float4*>(dst_a)[q] = s;
    else reinterpret_cast<float4*>(dst_b)[q - n4a] = s;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES, int KS>
int launch_gemm_s(const GemmArgs& p, int nsplit, hipStream_t st) {
  constexpr size_t lds = (size_t)(KS >= 10 ? KS - 8 : 2 * KS) * 16 * (BM + BN) * sizeof(float);
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N + (ONES ? 4 : 0), BN));
  allow_big_lds((const void*)k_gemm<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, KS>, lds);
  hipLaunchKernelGGL((k_gemm<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, KS>), dim3(tiles, nsplit),
                     dim3(64 * WAVES_M * WAVES_N), lds, st, p);
  return check_launch("gemm");
}

constexpr int kDefaultKS = 1;
inline int env_ks(int dflt) { return env_knob("PGNN_GEMM_KS", dflt); }

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES = false>
int launch_gemm(const GemmArgs& p, int nsplit, hipStream_t st) {
  // KS = LDS images (16 of k each) per barrier.  History: a 3-stage ring and a software-pipelined loop
  // (fragments of step t+1 and the DMA of step t+2 in flight behind the MFMAs of step t) measured no
  // faster than the plain 2-stage loop -- DMA issue, fragment reads and the barrier are *issue-time*
  // costs that both waves of a SIMD pay in lockstep in front of their MFMA burst -- so the lever is
  // fewer barriers per FLOP, i.e. a deeper k-step.
  const int ks = env_ks(kDefaultKS);
  if (ks == 11) return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 11>(p, nsplit, st);
  if (ks == 12) return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 12>(p, nsplit, st);
  if (ks >= 2) return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 2>(p, nsplit, st);
  return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 1>(p, nsplit, st);
}

// Tile configurations (all 4 waves, BK = 16).  N = 300 / 600 are 18.75 / 37.5 MFMA blocks wide, so the
// 304-wide tiles (19 blocks) waste 1.3 % and the 160-wide ones (10 blocks) 6.7 %, against 22 % for a
// power-of-two 128.  Small M (one 256-graph batch is ~6.8k rows) is a quantisation problem -- the
// whole product is only ~8 MFMA blocks per SIMD -- so it gets the smallest wave tiles that still
// give every SIMD a wave; large M gets the widest tile (least re-reading of the A panel).
enum TileCfg { T128x304 = 0, T64x160 = 1, T128x160 = 2, T128x128 = 3, T64x64 = 4, T64x160w8 = 5, T320x160 = 6, T256x304 = 7, kNumCfg = 8 };
struct CfgInfo { int bm, bn, wave_blocks; };
static const CfgInfo kCfg[kNumCfg] = {{128, 304, 38}, {64, 160, 10}, {128, 160, 20}, {128, 128, 16}, {64, 64, 4},
                                      {64, 160, 5}, {320, 160, 25}, {256, 304, 38}};

inline int env_int_linear(const char* name, int dflt) { return env_knob(name, dflt); }
inline int env_cfg() { return env_knob("PGNN_GEMM_CFG", -1); }

// At M = 6747 every tiling tried (64x160 with 4 or 8 waves, 64x64, 32x160, 32x320, 128x128; 2 or 3
// stages; 16- or 32-deep k-steps) lands at 36-42 us per product, as does rocBLAS: 2.43 GFLOP is ~24 us at
// the large-M rate plus a fixed ~10 us of dispatch ramp and drain, so that regime is bound by kernel
// granularity, not by the tile shape.
// kind: 0 = forward (both operands k-contiguous), 1 = backward-data (weights row-contiguous).
// Measured on MI355X (tools/gemm_bench.py, M = 262144 / 6747, N,K in {300,600}): forward is best on
// 64x160 / 64x64 at every M; backward-data prefers 128x304 once M is large, 64x160 below.
inline TileCfg pick_cfg(int64_t m, int64_t n, int kind) {
  const int forced = env_cfg();
  if (forced >= 0 && forced < kNumCfg) return (TileCfg)forced;
  // A 256x304 tile (8 waves x 32x304; 1.3 % instead of 6.7 % padding on N = 300 / 600) is 5-8 % faster in the
  // isolated micro-benchmark at M = 262144 (forward 92 -> 97-103 TFLOP/s) but makes the 16384-graph train step
  // 3 % SLOWER (55.9 vs 54.1 ms, measured A/B in one process) -- opt-in only.
  if (m >= 65536 && env_int_linear("PGNN_GEMM_WIDE", 0)) return T256x304;
  double best = 1e30;
  int arg = T64x64;
  for (int c = 0; c < kNumCfg; ++c) {
    if (c == T128x304 && (kind == 0 || m < 32768)) continue;
    if (c == T128x160 || c == T64x160 || c == T320x160 || c == T256x304) continue;  // 8-wave 64x160 beats the 4-wave one everywhere measured
    const int64_t tiles = ceil_div(m, kCfg[c].bm) * ceil_div(n, kCfg[c].bn);
    const int64_t per_simd = ceil_div(tiles * 4, 4 * num_cu());             // waves each SIMD must run
    const double t = (double)per_simd * (kCfg[c].wave_blocks + 5.0);     // + fixed per-tile overhead
    if (t < best) { best = t; arg = c; }
  }
  return (TileCfg)arg;
}

template <bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES = false>
int launch_cfg(TileCfg c, const GemmArgs& p, int nsplit, hipStream_t st) {
  switch (c) {
    case T128x304: return launch_gemm<128, 304, 4, 1, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T64x160: return launch_gemm<64, 160, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T128x160: return launch_gemm<128, 160, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T128x128: return launch_gemm<128, 128, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T64x160w8: return launch_gemm<64, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T320x160: return launch_gemm<320, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T256x304: return launch_gemm<256, 304, 8, 1, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    default: return launch_gemm<64, 64, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
  }
}

constexpr int kWgtBK = 16;

// split count for dW = dy^T x: every SIMD should get ~2 waves, each split at least 4 k-tiles deep
inline int weight_splits(int64_t m, int64_t k, int64_t n, int bm, int bn) {
  const int64_t tiles = ceil_div(n, bm) * ceil_div(k, bn);
  int64_t s = ceil_div(2 * num_cu(), tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(m / (4 * kWgtBK), 1));
  return (int)std::max<int64_t>(std::min<int64_t>(s, 256), 1);
}
// Weight-gradient product dW[Nout,Kin] = dy^T x reduces over the M rows, so every tile re-streams its
// dy / x panels from memory: traffic = |dy| * ceil(Kin/BN) + |x| * ceil(Nout/BM).  With 64x160 tiles that
// is 7.4 GB per product at M = 438k (measured 69 TFLOP/s = L2/HBM-bound); 320x160 halves it to 3.2 GB.
// Small M keeps the small tile: there the split count, not the traffic, is what fills the chip.
inline TileCfg weight_cfg(int64_t m) {
  const int forced = env_cfg();
  if (forced == T64x64 || forced == T128x128 || forced == T128x160 || forced == T64x160 || forced == T64x160w8 ||
      forced == T320x160)
    return (TileCfg)forced;
  return m >= 32768 ? T320x160 : T64x160w8;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

int pgnn_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                    int64_t m, int64_t k, int64_t n, int relu, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0,
               "linear_fwd: K, N and the leading dimensions must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  p.A = x; p.lda = ldx; p.B = w; p.ldb = k; p.C = y; p.ldc = ldy;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.bias = bias; p.relu = relu; p.kchunk = (int)k; p.split_stride = 0;
  return launch_cfg<true, true, EPI_BIAS>(pick_cfg(m, n, 0), p, 1, (hipStream_t)stream);
}

int pgnn_linear_bwd_data(const float* dy, int64_t lddy, const float* w, const float* relu_out, int64_t ldr,
                         float* dx, int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0,
               "linear_bwd_data: K, N and lddy must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  // C = dx [m, k] ; reduction over n ; A = dy (n contiguous) ; B(kcol, nn) = w[nn*k + kcol]
  p.A = dy; p.lda = lddy; p.B = w; p.ldb = k; p.C = dx; p.ldc = lddx;
  p.M = (int)m; p.N = (int)k; p.K = (int)n; p.mask = relu_out; p.ldmask = ldr; p.kchunk = (int)n; p.split_stride = 0;
  const TileCfg c = pick_cfg(m, k, 1);
  if (relu_out) return launch_cfg<true, false, EPI_MASK>(c, p, 1, (hipStream_t)stream);
  return launch_cfg<true, false, EPI_PLAIN>(c, p, 1, (hipStream_t)stream);
}

size_t pgnn_linear_bwd_weight_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  const TileCfg c = weight_cfg(m);
  return align_up((size_t)weight_splits(m, k, n, kCfg[c].bm, kCfg[c].bn) * (n * k + n) * sizeof(float), 256) + 256;
}

int pgnn_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db,
                           int64_t m, int64_t k, int64_t n, void* ws, size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0,
               "linear_bwd_weight: K, N and leading dimensions must be multiples of 4");
  if (ws_bytes < pgnn_linear_bwd_weight_workspace_bytes(m, k, n)) {
    set_error("linear_bwd_weight workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws);
  const TileCfg cfg = weight_cfg(m);
  const int nsplit = weight_splits(m, k, n, kCfg[cfg].bm, kCfg[cfg].bn);
  float* partial = cv.take<float>((size_t)nsplit * (n * k + n));
  GemmArgs p{};
  p.nxcd = num_xcd();
  // C = dW [n, k] ; reduction over rows m ; A(nout, r) = dy[r*lddy + nout] ; B(kcol, r) = x[r*ldx + kcol]
  // db[nout] = sum_r dy[r, nout] rides along as the "ones column" of B.
  p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx;
  p.M = (int)n; p.N = (int)k; p.K = (int)m;
  int64_t chunk = ceil_div(m, nsplit);
  chunk = ceil_div(chunk, kWgtBK) * kWgtBK;
  p.kchunk = (int)chunk;
  const int used = (int)ceil_div(m, chunk);
  const bool direct = used == 1;
  p.C = direct ? dw : partial;
  p.ldc = k;
  p.split_stride = direct ? 0 : n * k + n;
  p.colsum = direct ? db : partial + n * k;
  int rc = db ? launch_cfg<false, false, EPI_PLAIN, true>(cfg, p, used, st)
              : launch_cfg<false, false, EPI_PLAIN, false>(cfg, p, used, st);
  if (rc) return rc;
  if (!direct) {
    const int64_t n4a = n * k / 4, n4b = db ? n / 4 : 0;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((int)std::min<int64_t>(ceil_div(n4a + n4b, 256), 1024)), dim3(256), 0, st,
                       partial, used, n * k + n, dw, n4a, db, n4b);
  }
  return check_launch("linear_bwd_weight");
}

}  // extern "C"
