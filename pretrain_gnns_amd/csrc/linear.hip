// GEMMs for the GIN mlp / GCN linear (gfx950): forward, backward-data, backward-weight -- an fp32-MFMA kernel (k_gemm) and a
// split-bf16 kernel (k_gemm3, further down; the default wherever it is faster, see use_split / weight_split).
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

#include <type_traits>

#include <hip/hip_ext.h>

#include "bn_fold.h"
#include "common.h"
#include "two_plane.h"

namespace pgnn {
namespace {

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
  float* colstat;      // EPI_BIAS, optional: [ceil(M/16)][2][N] per 16-row block: column sums of C and sums of squared deviations
                       // from the block's own column mean (what a BatchNorm behind this product needs; see pgnn_linear_fwd_colstats)
  int nxcd;            // XCDs of the device (block -> tile remap)
  unsigned long long* dbg;   // k_gemm3w<DBG>: phase cycle totals
  const unsigned short* Bp;  // k_gemm3w: B as three pre-split bf16 planes [3][rows][ldbp], zero beyond K up to ldbp
  int64_t ldbp, bplane;      //           row pitch and plane pitch in bf16 elements
  const uint32_t* a_amax;    // k_gemm2pw, optional: [M] bit patterns of max |A[m, :]| (from the producer of A); NULL = taken in the kernel
  uint32_t* c_amax;          // k_gemm2pw, optional: [M] words, zero before the launch: receives max |C[m, :]| (atomic max of the tiles)
  BnFwdFold bnf;             // k_gemm2pw, EPI_BIAS, bnf.n > 0: the statistics of the BatchNorm behind C, folded in this launch (bn_fold.h)
  const float* F;            // gemm3_body<EXTRA>, optional: [K][12] more rows of Bop behind the ones column (Bop columns N + 4 .. N + 15)
  float* extra;              //                    their products [M][12] (same split stride as C)
  const uint32_t* a_colmax;  // gemm3_body<TWO>: [M] bit patterns of max_k |Aop(m, k)| (the column maxima of the row-contiguous A)
  const uint32_t* b_colmax;  //                  [N] the same for B; the ones column and the twelve extra columns run unscaled
};

enum { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_MASK = 2 };

__device__ float4 g_zero_page[4];  // zero-initialised: DMA source for lanes past the K / M / N edge
__device__ float4 g_ones_page[1] = {{1.f, 0.f, 0.f, 0.f}};  // DMA source of the "ones column" (bias gradient)


// epilogue shared by the fp32-MFMA and the split-bf16 kernels (identical C/D register layout): lane holds
// C[m = mw + 16 i + (lane & 15)][n = nw + 16 j + (lane >> 4) * 4 + 0..3] of each 16x16 block
// the ReLU mask operand of an EPI_MASK product, fetched BEFORE the k-loop (k_gemm3): at the 256-graph batch every workgroup
// reaches its epilogue at the same moment, and a mask load issued there is a full memory round trip in front of the stores
template <int MI, int NI>
__device__ __forceinline__ void gemm_prefetch_mask(const GemmArgs& p, float4 (&mk)[MI][NI], int mw, int nw, int lane) {
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = nw + j * 16 + fk * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // clamped, unconditional loads (a guarded load compiles to a branch with its own s_waitcnt: five serial round trips);
      // the epilogue never looks at the values of rows / columns beyond the edge
      const int m = min(mw + i * 16 + fr, p.M - 1);
      mk[i][j] = *reinterpret_cast<const float4*>(p.mask + (int64_t)m * p.ldmask + min(n, p.N - 4));
    }
  }
}

template <int EPI, bool ONES, int MI, int NI, int PM>
__device__ __forceinline__ void gemm_epilogue_pre(const GemmArgs& p, f32x4 (&acc)[MI][NI], int mw, int nw, int lane,
                                                  const float4 (&pre)[PM][NI]) {  // PM == MI: pre holds the mask values
  const int fr = lane & 15, fk = lane >> 4;
  float* C = p.C + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = nw + j * 16 + fk * 4;
    if (ONES && n == p.N) {  // the ones column: per-row sums of Aop
      float* cs = p.colsum + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mw + i * 16 + fr;
        if (m < p.M) cs[m] = acc[i][j][0];
      }
      continue;
    }
    if (n >= p.N) continue;
    float4 bv = f4_zero();
    if (EPI == EPI_BIAS && p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mw + i * 16 + fr;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (EPI == EPI_BIAS) {
        v = f4_add(v, bv);
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (p.colstat) {  // (uniform) the 16 lanes of a DPP row hold the 16 rows of this block for the same four columns
          const int mb = mw + i * 16, cnt = min(16, p.M - mb);
          if (cnt > 0) {
            const bool ok = fr < cnt;
            float4 s = ok ? v : f4_zero();
            s.x = row16_sum(s.x); s.y = row16_sum(s.y); s.z = row16_sum(s.z); s.w = row16_sum(s.w);
            const float inv = 1.f / (float)cnt;
            float4 q;
            q.x = ok ? v.x - s.x * inv : 0.f; q.y = ok ? v.y - s.y * inv : 0.f;
            q.z = ok ? v.z - s.z * inv : 0.f; q.w = ok ? v.w - s.w * inv : 0.f;
            q.x = row16_sum(q.x * q.x); q.y = row16_sum(q.y * q.y); q.z = row16_sum(q.z * q.z); q.w = row16_sum(q.w * q.w);
            if (fr == 0) {
              float* cs = p.colstat + (int64_t)(mb >> 4) * 2 * p.N + n;
              *reinterpret_cast<float4*>(cs) = s;
              *reinterpret_cast<float4*>(cs + p.N) = q;
            }
          }
        }
      }
      if (m >= p.M) continue;
      if (EPI == EPI_MASK) {
        float4 mk;
        if constexpr (PM == MI) mk = pre[i][j];
        else mk = *reinterpret_cast<const float4*>(p.mask + (int64_t)m * p.ldmask + n);
        if (!(mk.x > 0.f)) v.x = 0.f;
        if (!(mk.y > 0.f)) v.y = 0.f;
        if (!(mk.z > 0.f)) v.z = 0.f;
        if (!(mk.w > 0.f)) v.w = 0.f;
      }
      *reinterpret_cast<float4*>(C + (int64_t)m * p.ldc + n) = v;
    }
  }
}

template <int EPI, bool ONES, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[MI][NI], int mw, int nw, int lane) {
  const float4 none[MI + 1][NI] = {};
  gemm_epilogue_pre<EPI, ONES, MI, NI, MI + 1>(p, acc, mw, nw, lane, none);
}

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

  gemm_epilogue<EPI, ONES, MI, NI>(p, acc, m0 + wm0, n0 + wn0, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// fp32 product on the bf16 matrix cores.  Every fp32 operand value is
// split EXACTLY into three bf16 terms
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (3 x 8 = 24 significand bits)
// and a.b is accumulated as the six products whose weight is above 2^-24 of |a.b|:
//     a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)            [dropped: a2b3 + a3b2 + a3b3 <= 2^-24 |a||b| (1 + 2^-7)]
// Each bf16 x bf16 product is exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in fp32, so the result carries
// the error of an fp32 FMA chain (measured against float64 in tests/test_gpu_ops.py: not larger than the fp32-MFMA
// kernel's) at 6/16 of its matrix-core time: 6 x 16 cycles per 16x16x32 block against 8 x 32 for v_mfma_f32_16x16x4_f32.
// (Values within a factor 2^-8 of FLT_MAX would round a1 to inf; activations / weights are nowhere near.)
//
// Staging goes through registers (the split is VALU work on the way in, done once per element per workgroup):
// global float4 -> v_cvt_pk_bf16_f32 / subtract, twice -> three bf16 planes in LDS, [row][32 k] = 64-byte rows whose
// four 16-byte chunks are XOR-swizzled by the row, so the ds_write_b64 of the staging pass and the ds_read_b128 of an
// MFMA fragment (lane: row l&15, chunk l>>4) are both conflict-free.  Tile t+1's global loads are in flight behind tile
// t's MFMAs; one LDS stage (two barriers per 32 of k) and 8 waves per workgroup, two workgroups per CU.
//
// What was measured on the way (tools/gemm_split_check.py, M = 262144 and 6747, profiles/r02/README.md):
//  - the phases of a k-step add up instead of overlapping (skip-one-phase runs: epilogue/prologue 149 us, MFMA 293,
//    LDS writes + barriers 73, split arithmetic 90, global loads 150 of 755 at M = 262144, K = 300, N = 600): the two
//    resident workgroups settle into lockstep, so the kernel sits at 1.45-1.5x the fp32-MFMA kernel rather than 2.7x;
//  - 4-wave workgroups, a two-stage LDS ring with the staging interleaved behind the MFMAs, a two-tile register ring,
//    256x160 / 128x320 tiles, weights pre-split (and pre-transposed) into bf16 planes by a side kernel, and wave
//    specialisation (8 loader waves that load / split / stage + 8 consumer waves that only issue MFMAs, two LDS stages,
//    1-4 tiles of loads in flight): all slower or equal.  The K sweep (tools/gemm_ksweep.py, M = 6747, N = 608) puts a
//    k-step at 1.5 us where its MFMAs need ~0.9, with 8.5 us fixed per launch, in the symmetric AND the specialised form;
//  - row-contiguous operands with a 4x4 register transposition in front of the split: slower than the fp32 MFMA.  What
//    replaced it -- untransposed staging + ds_read_b64_tr_b16 fragment reads, RowMajorTile below -- is faster than the fp32
//    MFMA up to ~32 k rows (backward-weight) / ~65 k rows (backward-data).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // round to nearest even, a in the low half
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// (a, b) -> packed (a1,b1), (a2,b2), (a3,b3)
__device__ __forceinline__ void split3(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pack_bf16(ra, rb);
  const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  l = pack_bf16(sa, sb);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS image of one operand tile of a 32-deep k-step, one bf16 plane:
//   k-contiguous operand  : [row][32 k], 64-byte rows, 16-byte chunks XOR-swizzled by the row; fragment = one ds_read_b128
//   row-contiguous operand: [32 k][S columns] as the data lies in memory (no transposition on the way in: a float4 is four
//                           columns of one k); the MFMA fragment -- 8 consecutive k of one column per lane -- comes out of
//                           two ds_read_b64_tr_b16 (each 16-lane group hands over a 4 k x 16 column block transposed;
//                           semantics probed in tools/probe/tr16_probe.hip).  Rows with bit 3 of k set are rotated by 16
//                           columns and S = columns (+ 32 unless columns % 64 == 32): conflict-free for the instruction's
//                           32-lane groups (brute-forced over its bank map, (byte / 4) % 64).
template <int COLS>
struct RowMajorTile {
  static constexpr int S = (COLS % 64 == 32) ? COLS : COLS + 32;
  static constexpr int PLANE = 32 * S * 2;  // bytes
  __device__ static __forceinline__ int offset(int k, int col) { return (k * S + (col + 16 * ((k >> 3) & 1)) % S) * 2; }
};
template <int ROWS>
struct KMajorTile {
  static constexpr int PLANE = ROWS * 64;
  __device__ static __forceinline__ int offset(int row, int kq) {
    return row * 64 + (((kq >> 1) ^ ((-(row >> 2)) & 3)) * 16) + (kq & 1) * 8;
  }
};

// (64-deep k-steps -- two 32-deep slice images per barrier pair, PGNN_GEMM3_KS=2 of a round-2 experiment -- measured SLOWER: the
// 256-graph step 1.313 against 1.189 ms.  With K = 300 / 600 a tile has 10 / 19 steps; halving them doubles the prologue (two
// slices fetched and split before the first MFMA) and the registers held across the loop, which costs more than the barriers saved.)
// One kernel, four operand-layout instantiations: forward / backward-data on transposed weights (both k-contiguous),
// backward-data (dy k-contiguous, W row-contiguous), backward-weight (both row-contiguous, split over k = rows, optional
// ones column for the bias gradient).
// EXTRA (row-contiguous B with ONES only): twelve more columns of Bop behind the ones column, from p.F [K][12] -- they ride in the
// column padding of the last tile (N + 16 <= tiles_n BN) and their products leave through p.extra [M][12]; see linear_bwd_weight_pair_ext
// TWO (round 6; weight gradients, both operands row-contiguous): the operands on TWO fp16 planes under a power-of-two scale per
// COLUMN instead of three bf16 planes -- dW = dy^T x contracts over the rows, so the scales that factor out of the sum belong to the
// columns of dy and of x: C[m][n] = (1 / sa_m)(1 / sb_n) sum_k (sa_m A[k][m]) (sb_n B[k][n]).  The staging thread, which keeps the
// same four columns for the whole k-loop, multiplies its float4 by their scales (registers) and splits it into two planes; three
// v_mfma_f32_16x16x32_f16 per accumulator (low.high, high.low, high.high) instead of six bf16 products, a third less LDS traffic
// either way, 36 instead of 55 KB per 64x160 workgroup; the epilogue's rescale is exact.  The column maxima come from the caller
// (p.a_colmax / p.b_colmax: k_colmax_jobs, or the producers of the operands).
// PFD (round 6): register stages of the staging loads.  1: the loads of k-step t + 2 are issued behind the stores of t + 1 and waited
// for one k-step later.  2: two register sets, k-step t + 3 is issued where t + 2 was -- the wait in front of a store covers loads that
// are two k-steps old.  Measured level at the paired weight gradients (0.943-0.944 against 0.925-0.937 ms per step,
// profiles/r06/dw_two_planes_ab.txt: the two resident workgroups of a CU already cover each other's waits): -DPGNN_AB builds only.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES, bool EXTRA = false, bool TWO = false, int PFD = 1>
__device__ __forceinline__ void gemm3_body(const GemmArgs& p, const int block_x, const int grid_x, const int split = -1, const int tile_direct = -1) {
  static_assert(PFD == 1 || PFD == 2, "one or two register stages");
  static_assert(!EXTRA || (ONES && !B_KMAJOR), "EXTRA rides behind the ones column of a row-contiguous B");
  static_assert(!TWO || (!A_KMAJOR && !B_KMAJOR && EPI == EPI_PLAIN), "column scales: both operands row-contiguous, plain epilogue");
  constexpr int PL = TWO ? 2 : 3;  // planes per operand
  constexpr int BK = 32;
  constexpr int NW = WAVES_M * WAVES_N, T = 64 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % 16 == 0 && BN % 16 == 0, "tile shape");
  using TA = typename std::conditional<A_KMAJOR, KMajorTile<BM>, RowMajorTile<BM>>::type;
  using TB = typename std::conditional<B_KMAJOR, KMajorTile<BN>, RowMajorTile<BN>>::type;
  constexpr int PA = TA::PLANE, PB = TB::PLANE;  // bytes per plane; a stage = three A planes, then three B planes
  constexpr int UA = BM * 8, UB = BN * 8;        // staging units (one float4 each) per operand
  constexpr int NA = (UA + T - 1) / T, NB = (UB + T - 1) / T;

  extern __shared__ __align__(16) unsigned char smem3[];
  unsigned char* const ldsA = smem3;
  unsigned char* const ldsB = smem3 + PL * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + (ONES ? 4 : 0) + BN - 1) / BN;
  const int tile = tile_direct >= 0 ? tile_direct : xcd_remap(block_x, grid_x, p.nxcd);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = (split >= 0 ? split : (int)blockIdx.y) * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- this thread's staging units.  k-contiguous: unit u = (row u >> 3, k-quad u & 7), source advances along k;
  //      row-contiguous: unit u = (k u / (cols/4), column quad u % (cols/4)), source advances by 32 rows per step.
  const float* srcA[NA];
  const float* srcB[NB];
  int offA[NA], offB[NB], kofA[NA], kofB[NB];
  bool okA[NA], okB[NB], oneB[NB], extB[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int u = min(tid + j * T, UA - 1);  // (a clamped duplicate rewrites the same LDS bytes with the same values)
    if (A_KMAJOR) {
      const int row = u >> 3, kq = u & 7;
      okA[j] = m0 + row < p.M;
      kofA[j] = 4 * kq;
      srcA[j] = p.A + (int64_t)(m0 + row) * p.lda + kbeg + 4 * kq;
      offA[j] = KMajorTile<BM>::offset(row, kq);
    } else {
      const int kr = u / (BM / 4), cq = u - kr * (BM / 4);
      okA[j] = m0 + 4 * cq < p.M;
      kofA[j] = kr;
      srcA[j] = p.A + (int64_t)(kbeg + kr) * p.lda + m0 + 4 * cq;
      offA[j] = RowMajorTile<BM>::offset(kr, 4 * cq);
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int u = min(tid + j * T, UB - 1);
    oneB[j] = false;
    extB[j] = false;
    if (B_KMAJOR) {
      const int row = u >> 3, kq = u & 7;
      okB[j] = n0 + row < p.N;
      kofB[j] = 4 * kq;
      srcB[j] = p.B + (int64_t)(n0 + row) * p.ldb + kbeg + 4 * kq;
      offB[j] = KMajorTile<BN>::offset(row, kq);
    } else {
      const int kr = u / (BN / 4), cq = u - kr * (BN / 4);
      okB[j] = n0 + 4 * cq < p.N;
      oneB[j] = ONES && n0 + 4 * cq == p.N;  // the ones column: C[m][N] = sum_k Aop(m, k) = the bias gradient
      kofB[j] = kr;
      srcB[j] = p.B + (int64_t)(kbeg + kr) * p.ldb + n0 + 4 * cq;
      if constexpr (EXTRA) {
        extB[j] = p.F != nullptr && n0 + 4 * cq >= p.N + 4 && n0 + 4 * cq < p.N + 16;
        if (extB[j]) srcB[j] = p.F + (int64_t)(kbeg + kr) * 12 + (n0 + 4 * cq - p.N - 4);
      }
      offB[j] = RowMajorTile<BN>::offset(kr, 4 * cq);
    }
  }
  // (TWO) the power-of-two scales of this thread's four columns per staging unit: s amax lands in [2^13, 2^14) (two_plane.h)
  float4 scA[TWO ? NA : 1], scB[TWO ? NB : 1];
  if constexpr (TWO) {
    auto scale_of = [&](const uint32_t* cm, int c, int lim) {
      float sc = 1.f, inv;
      if (cm != nullptr && c < lim) pow2_scales(__uint_as_float(cm[c]), sc, inv);
      return sc;
    };
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int u = min(tid + j * T, UA - 1);
      const int c = m0 + 4 * (u % (BM / 4));
      scA[j] = make_float4(scale_of(p.a_colmax, c, p.M), scale_of(p.a_colmax, c + 1, p.M), scale_of(p.a_colmax, c + 2, p.M), scale_of(p.a_colmax, c + 3, p.M));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int u = min(tid + j * T, UB - 1);
      const int c = n0 + 4 * (u % (BN / 4));  // (columns from N on -- the ones column, the extra columns, padding -- are not scaled)
      scB[j] = make_float4(scale_of(p.b_colmax, c, p.N), scale_of(p.b_colmax, c + 1, p.N), scale_of(p.b_colmax, c + 2, p.N), scale_of(p.b_colmax, c + 3, p.N));
    }
  }
  float4 rA[PFD][NA], rB[PFD][NB];
  auto load_tile_into = [&](int it, float4 (&ra)[NA], float4 (&rb)[NB]) {
    const int k0 = kbeg + it * BK;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool ok = okA[j] && k0 + kofA[j] < kend;
      const float* g = ok ? srcA[j] + (A_KMAJOR ? (int64_t)it * BK : (int64_t)it * BK * p.lda) : reinterpret_cast<const float*>(g_zero_page);
      ra[j] = *reinterpret_cast<const float4*>(g);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool kok = k0 + kofB[j] < kend;
      const float* g = (okB[j] && kok) ? srcB[j] + (B_KMAJOR ? (int64_t)it * BK : (int64_t)it * BK * p.ldb)
                                       : reinterpret_cast<const float*>((ONES && oneB[j] && kok) ? g_ones_page : g_zero_page);
      if constexpr (EXTRA) {
        if (extB[j] && kok) g = srcB[j] + (int64_t)it * BK * 12;
      }
      rb[j] = *reinterpret_cast<const float4*>(g);
    }
  };
  auto store_unit = [&](const float4& v, unsigned char* base, int plane, int off) {
    uint32_t h0, m0_, l0, h1, m1, l1;
    split3(v.x, v.y, h0, m0_, l0);
    split3(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(base + plane + off) = make_uint2(m0_, m1);
    *reinterpret_cast<uint2*>(base + 2 * plane + off) = make_uint2(l0, l1);
  };
  auto store_unit2 = [&](const float4& v, const float4& sc, unsigned char* base, int plane, int off) {
    uint32_t h0, l0, h1, l1;
    split2(v.x * sc.x, v.y * sc.y, h0, l0);
    split2(v.z * sc.z, v.w * sc.w, h1, l1);
    *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(base + plane + off) = make_uint2(l0, l1);
  };
  auto store_tile_from = [&](const float4 (&ra)[NA], const float4 (&rb)[NB]) {
    if constexpr (TWO) {
#pragma unroll
      for (int j = 0; j < NA; ++j) store_unit2(ra[j], scA[j], ldsA, PA, offA[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j) store_unit2(rb[j], scB[j], ldsB, PB, offB[j]);
      return;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) store_unit(ra[j], ldsA, PA, offA[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) store_unit(rb[j], ldsB, PB, offB[j]);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fk = lane >> 4;
  // fragment of the 16 tile rows / columns starting at `c0`, plane q.  The lane's byte offset inside a plane does not depend on the
  // k-step: it is formed once per 16-row / 16-column block in front of the loop (round 6: hipcc re-formed the wrapped column -- add,
  // compare, select -- for every block in every k-step; hoisted, the large weight gradient measures level in isolation and the
  // 16 384-graph step 27.56 -> 27.42 ms on one box, profiles/r06/dw_two_planes_ab.txt); inside the loop a fragment is
  // base + q * plane (an immediate) + that offset.
  auto frag_off = [&](auto kmajor, auto cols_tag, int c0) -> int {
    constexpr int COLS = decltype(cols_tag)::value;
    if constexpr (decltype(kmajor)::value) {
      return (c0 + fr) * 64 + ((fk ^ ((-(fr >> 2)) & 3)) * 16);
    } else {
      // lane (i = fr, g = fk): k-blocks 2g and 2g+1 (rows 8g .. 8g+7, all with the same bit 3 = g & 1), row 4 kb + i / 4,
      // its address = four consecutive columns of that row; the group's 16 lanes receive one column each
      using L = RowMajorTile<COLS>;
      const int col = (c0 + 4 * (fr & 3) + 16 * (fk & 1)) % L::S;
      return ((8 * fk + (fr >> 2)) * L::S + col) * 2;
    }
  };
  auto frag = [&](auto kmajor, auto cols_tag, const unsigned char* base, int plane, int off, int q) -> bf16x8 {
    constexpr int COLS = decltype(cols_tag)::value;
    if constexpr (decltype(kmajor)::value) {
      return *reinterpret_cast<const bf16x8*>(base + q * plane + off);
    } else {
      using L = RowMajorTile<COLS>;
      const unsigned char* a0 = base + q * plane + off;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * L::S * 2));
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  using KA = std::integral_constant<bool, A_KMAJOR>;
  using KB = std::integral_constant<bool, B_KMAJOR>;
  using CA = std::integral_constant<int, BM>;
  using CB = std::integral_constant<int, BN>;
  int foA[MI], foB[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) foA[i] = frag_off(KA{}, CA{}, wm0 + i * 16);
#pragma unroll
  for (int j = 0; j < NI; ++j) foB[j] = frag_off(KB{}, CB{}, wn0 + j * 16);

  auto compute = [&]() {
    if constexpr (TWO) {
      f16x8 a[MI][2];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) a[i][q] = __builtin_bit_cast(f16x8, frag(KA{}, CA{}, ldsA, PA, foA[i], q));
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f16x8 b[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) b[q] = __builtin_bit_cast(f16x8, frag(KB{}, CB{}, ldsB, PB, foB[j], q));
        // smallest terms first (the order of k_gemm2pw); operands swapped (D = B x A) as below
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[1], a[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[0], a[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[0], a[i][0], acc[i][j], 0, 0, 0);
      }
      return;
    }
    bf16x8 a[MI][3];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[i][q] = frag(KA{}, CA{}, ldsA, PA, foA[i], q);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      bf16x8 b[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) b[q] = frag(KB{}, CB{}, ldsB, PB, foB[j], q);
      // smallest terms first; operands swapped (D = B x A) so a lane ends up with 4 consecutive columns of one C row
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[i][2], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], a[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2], a[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], a[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], a[i][0], acc[i][j], 0, 0, 0);
    }
  };

  if (nk > 0) {  // (k-step j lives in register set j % PFD)
    load_tile_into(0, rA[0], rB[0]);
    store_tile_from(rA[0], rB[0]);
    if (1 < nk) load_tile_into(1, rA[PFD - 1], rB[PFD - 1]);
    if (PFD == 2 && 2 < nk) load_tile_into(2, rA[0], rB[0]);
  }
  float4 mk[MI][NI];
  if constexpr (EPI == EPI_MASK) gemm_prefetch_mask<MI, NI>(p, mk, m0 + wm0, n0 + wn0, lane);  // lands under the k-loop
  __syncthreads();
  if constexpr (PFD == 1) {
    for (int t = 0; t < nk; ++t) {
      compute();
      __syncthreads();  // every wave is done reading the stage
      if (t + 1 < nk) store_tile_from(rA[0], rB[0]);
      __syncthreads();
      if (t + 2 < nk) load_tile_into(t + 2, rA[0], rB[0]);
    }
  } else {
    auto kstep = [&](int t, auto set_tag) {  // stores k-step t + 1 (set (t + 1) % 2), refills that set with k-step t + 3
      constexpr int S = decltype(set_tag)::value;
      compute();
      __syncthreads();
      if (t + 1 < nk) store_tile_from(rA[S], rB[S]);
      __syncthreads();
      if (t + 3 < nk) load_tile_into(t + 3, rA[S], rB[S]);
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      kstep(t, std::integral_constant<int, PFD - 1>{});
      kstep(t + 1, std::integral_constant<int, 0>{});
    }
    if (t < nk) kstep(t, std::integral_constant<int, PFD - 1>{});
  }
  if constexpr (TWO) {  // back to the operands' own scale: exact (powers of two)
    auto inv_of = [&](const uint32_t* cm, int c, int lim) {
      float sc, inv = 1.f;
      if (cm != nullptr && c < lim) pow2_scales(__uint_as_float(cm[c]), sc, inv);
      return inv;
    };
    float ia[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) ia[i] = inv_of(p.a_colmax, m0 + wm0 + i * 16 + fr, p.M);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn0 + j * 16 + fk * 4;
      const float i0 = inv_of(p.b_colmax, n, p.N), i1 = inv_of(p.b_colmax, n + 1, p.N), i2 = inv_of(p.b_colmax, n + 2, p.N), i3 = inv_of(p.b_colmax, n + 3, p.N);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        acc[i][j][0] = acc[i][j][0] * ia[i] * i0;
        acc[i][j][1] = acc[i][j][1] * ia[i] * i1;
        acc[i][j][2] = acc[i][j][2] * ia[i] * i2;
        acc[i][j][3] = acc[i][j][3] * ia[i] * i3;
      }
    }
  }
  if constexpr (EPI == EPI_MASK) gemm_epilogue_pre<EPI, ONES, MI, NI, MI>(p, acc, m0 + wm0, n0 + wn0, lane, mk);
  else gemm_epilogue<EPI, ONES, MI, NI>(p, acc, m0 + wm0, n0 + wn0, lane);
  if constexpr (EXTRA) {  // Bop columns N + 4 .. N + 15 (the epilogue above skips everything beyond the ones column)
    if (p.F != nullptr) {
      float* gx = p.extra + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn0 + j * 16 + fk * 4;
        if (n < p.N + 4 || n >= p.N + 16) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int m = m0 + wm0 + i * 16 + fr;
          if (m < p.M) *reinterpret_cast<float4*>(gx + (int64_t)m * 12 + (n - p.N - 4)) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_gemm3(GemmArgs p) {
  gemm3_body<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, blockIdx.x, gridDim.x);
}
// The same product on a ONE-dimensional grid that keeps the tiles of a split on one XCD (round 6; the large weight gradients).  A tile
// of dW = dy^T x re-streams its two operand panels over ALL rows; the `tiles` workgroups of one split read the SAME rows -- at
// 438 792 rows and 320x160 tiles every byte of dy and x is wanted by two (dW1) or four / one (dW2) workgroups, 3.2 GB against
// 1.6 GB if each were fetched once.  Workgroups are dealt to the XCDs round-robin by linear id; on the (tiles, splits) grid the
// tiles of a split therefore land on `tiles` DIFFERENT XCDs, each with its own L2, and every one of them fetches the panels from
// HBM / MALL itself.  Here workgroup id = xcd + nxcd * j is tile j % tiles of split xcd + nxcd * (j / tiles): the tiles of a split are
// neighbours in dispatch order on ONE XCD, walk the rows at the same pace, and the second reader of a panel finds it in that L2.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_gemm3_xg(GemmArgs p, int tiles, int nsplit) {
  const int id = blockIdx.x, xcd = id % p.nxcd, j = id / p.nxcd;
  const int tile = j % tiles, split = xcd + p.nxcd * (j / tiles);
  if (split >= nsplit) return;  // (the grid is rounded up to whole XCD rows when the split count is not a multiple of the XCD count)
  GemmArgs q = p;  // (the epilogues index the partial matrices by blockIdx.y, which is 0 here)
  q.C = p.C + (int64_t)split * p.split_stride;
  if (ONES) q.colsum = p.colsum + (int64_t)split * p.split_stride;
  gemm3_body<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES>(q, 0, 1, split, tile);
}
// Two products of one shape family in ONE launch (blockIdx.z picks the product; round 4: the two weight gradients of a layer --
// twice the tiles per launch, so half the splits over the contracted rows: half the partial matrices to write and to fold, and a
// workgroup's pipeline fill paid once per 17 k-steps instead of once per 9)
struct GemmArgs2 {
  GemmArgs a[2];
  int tiles[2];
};
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES, bool EXTRA = false, bool TWO = false, int PFD = 1>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_gemm3_pair(GemmArgs2 q) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x >= q.tiles[z]) return;
  gemm3_body<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, EXTRA, TWO, PFD>(q.a[z], blockIdx.x, q.tiles[z]);
}

// ------------------------------------------------------------------------------------------------------------------
// The same arithmetic with every in-loop staging instruction gone (round 3): C = A . B^T, both operands contiguous along k,
//   A  fp32 activations, DMA'd (global_load_lds) as they lie into a [row][32 k] fp32 image (128-byte rows, 16-byte chunks
//      XOR-swizzled by (row & 6) | (row >> 3) through the DMA's per-lane SOURCE address: both ds_read_b128 of a fragment are
//      conflict-free over the instruction's 16-lane groups, brute-forced over the bank map) and split into its three bf16
//      terms by the CONSUMER wave, on its 16 x 32 fragment, between the LDS read and the MFMAs (44 VALU per fragment against
//      the 30 MFMAs = 480 matrix-pipe cycles it feeds at a 16 x 80 wave tile);
//   B  weights, split ONCE per step into three zero-padded bf16 planes by pgnn_split_weights (k_split_jobs below) and DMA'd
//      into the [row][32 k] bf16 images of k_gemm3 (same swizzle, same fragment reads).
// No VGPR staging, no v_cvt / ds_write in front of a barrier: a k-step is wait(counted vmcnt) -> s_barrier -> DMA issue of step
// t + STAGES - 1 -> fragment reads + split + MFMAs, over a 3- or 4-deep LDS ring (38 / 46 KB per stage at 64x160 / 128x160).
// Every output element sees the same terms in the same order as in k_gemm3: the two kernels are bit-identical.
__device__ __forceinline__ uint64_t gemm_clock() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// DBG (pgnn_debug_gemm3w_profile only): lane 0 of every wave of the first 8 workgroups accumulates s_memtime deltas of its
// phases -- wait + barrier, DMA issue, A split, multiply -- into p.dbg [8 workgroups][NW][8] = {wait, issue, split, mma, total}
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, bool PP, bool DBG = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) __attribute__((amdgpu_waves_per_eu(2, 2))) k_gemm3w(GemmArgs p) {
  constexpr int BK = 32;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % 16 == 0 && BN % 16 == 0, "tile shape");
  constexpr int A_BYTES = BM * 128, B_PLANE = BN * 64, STAGE = A_BYTES + 3 * B_PLANE;
  constexpr int PA = BM / 8, PB = BN / 16;       // 1-KiB DMA pieces: 8 fp32 rows / 16 bf16 rows each
  constexpr int NP = PA + 3 * PB, NJ = (NP + NW - 1) / NW;

  extern __shared__ __align__(16) unsigned char smem3w[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: piece selection compiles to scalar code
  const int tiles_n = (p.N + BN - 1) / BN;
  // PP (persistent): the workgroup walks tiles q = blockIdx.x, + gridDim.x, ... and keeps its DMA ring running ACROSS them -- the
  // last STAGES - 1 k-steps of a tile fetch the first stages of the next one, so only the first tile of a workgroup pays the
  // pipeline fill, and no tile pays a workgroup launch / retirement (together 6.6-8 us of a 20 us tile at 41 269 x 300 x 600,
  // where a CU holds ONE of these workgroups and nothing else overlaps them)
  const int ntiles = tiles_n * ((p.M + BM - 1) / BM);
  int tile = xcd_remap(blockIdx.x, gridDim.x, p.nxcd);
  int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + BK - 1) / BK;

  // ---- this wave's DMA pieces: piece d < PA = rows 8 d .. 8 d + 7 of the A image, else 16 rows of one B plane.  Every wave
  // issues exactly NJ pieces per k-step (the counted vmcnt below is then a compile-time immediate): a wave whose last slot falls
  // beyond NP repeats piece NP - 1 -- the same bytes to the same LDS address as its owner, harmless.
  // One code path for both kinds (a branch on the kind is a join inside the k-step, see step()): source = start of this lane's
  // row + min(offset of its chunk, last whole chunk of the row) -- beyond K the A chunks re-read the row's last whole chunk
  // instead of a zero page: finite values against the B planes' zero padding, which also bounds the B offsets -- ; the offset
  // advances by one k-step of bytes (128 for the fp32 A image, 64 for a bf16 plane); the LDS address is wave-uniform.
  const unsigned char* src[NJ];
  int koff[NJ], klast[NJ], kstep[NJ], ldsoff[NJ];
  auto set_pieces = [&](int pm0, int pn0) {  // the DMA sources of tile (pm0, pn0), offsets back at its first k-step
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int d = min(wave + j * NW, NP - 1);
      if (d < PA) {
        const int row = 8 * d + (lane >> 3);
        const int c = (lane & 7) ^ (((lane >> 3) & 6) | (d & 1));  // logical 16-byte chunk stored at position lane & 7
        koff[j] = 16 * c;
        klast[j] = 4 * (p.K - 4);
        kstep[j] = BK * 4;
        ldsoff[j] = d * 1024;
        src[j] = reinterpret_cast<const unsigned char*>(p.A + (int64_t)min(pm0 + row, p.M - 1) * p.lda);
      } else {
        const int q = (d - PA) / PB, pb = (d - PA) % PB;
        const int row = 16 * pb + (lane >> 2);
        const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
        koff[j] = 16 * c;
        klast[j] = 2 * ((int)p.ldbp - 8);
        kstep[j] = BK * 2;
        ldsoff[j] = A_BYTES + (d - PA) * 1024;
        src[j] = reinterpret_cast<const unsigned char*>(p.Bp + q * p.bplane + (int64_t)min(pn0 + row, p.N - 1) * p.ldbp);
      }
    }
  };
  set_pieces(m0, n0);
  auto issue_piece = [&](int j, int stage) {
    __builtin_amdgcn_global_load_lds(PGNN_GPTR(src[j] + min(koff[j], klast[j])), PGNN_LPTR(smem3w + stage * STAGE + ldsoff[j]), 16, 0, 0);
    koff[j] += kstep[j];
  };
  auto issue = [&](int stage, int it) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue_piece(j, stage);
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

  auto aload = [&](int stage, f32x4 (&lo)[MI], f32x4 (&hi)[MI]) {  // raw fp32 A fragment: two ds_read_b128 per 16 rows
    const unsigned char* sa = smem3w + stage * STAGE;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      lo[i] = *reinterpret_cast<const f32x4*>(sa + a_off + i * 16 * 128 + a_lo);
      hi[i] = *reinterpret_cast<const f32x4*>(sa + a_off + i * 16 * 128 + a_hi);
    }
  };
  bf16x8 b[NI][3];  // B fragments, read two column blocks ahead of their MFMAs (static indices: registers)
  auto bload = [&](int stage, int j) {
    const unsigned char* sb = smem3w + stage * STAGE + A_BYTES;
#pragma unroll
    for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(sb + q * B_PLANE + b_off + j * 16 * 64);
  };
  // quarter c of the three-term split: k pair c of every row block -> dword c of the three planes' fragments
  auto asplit_q = [&](int c, const f32x4 (&lo)[MI], const f32x4 (&hi)[MI], uint4 (&pl)[MI][3]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x4 v = c < 2 ? lo[i] : hi[i];
      uint32_t h, m, l;
      split3(v[2 * (c & 1)], v[2 * (c & 1) + 1], h, m, l);
      (&pl[i][0].x)[c] = h; (&pl[i][1].x)[c] = m; (&pl[i][2].x)[c] = l;
    }
  };
  // (pure VALU code is not ordered against sched_barrier by itself: an empty volatile asm that consumes the quarter's results is)
  auto pin_q = [&](int c, const uint4 (&pl)[MI][3]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" ::"v"((&pl[i][0].x)[c]), "v"((&pl[i][1].x)[c]), "v"((&pl[i][2].x)[c]));
  };
  // One k-step of a wave.  tools/probe/kstep_probe.hip prices its ingredients at a 64x160 tile, two waves per SIMD, per CU and
  // k-step: 60 MFMAs per SIMD 0.53 us (the matrix pipes saturate at ~1.85 PFLOP/s bf16: the chip clocks down under them;
  // six dependent MFMAs on one accumulator run as fast as six independent ones), 38 DMA pieces 0.28 us (64 B per clock and CU
  // out of L2), 136 ds_read_b128 0.11 us, the split 0.02 us -- and a k-step that reads its fragments, waits for them, splits and
  // then multiplies pays their SUM (0.94 us: what k_gemm3 and the first version of this kernel measured).  So nothing but MFMAs
  // may sit on a wave's critical path: column block j multiplies (6 MI MFMAs, k_gemm3's term order: bit-identical) while the B
  // fragments of block j + 2 are read -- the last two blocks read the first two of the NEXT step, whose stage barrier t has
  // already seen landed -- the DMAs of step t + STAGES - 1 and the raw A fragment of step t + 1 go out behind block 0, and a
  // quarter of its split rides behind each of blocks 1-4.  sched_barrier(0) after every block pins that interleaving (left alone,
  // hipcc hoists every read to the top of the step and sinks the split below it).  Operands swapped (D = B x A): a lane ends up
  // with four consecutive columns of a C row.
  static_assert(NI >= 5, "the split's quarters ride behind column blocks 1-4");
  auto step = [&](auto do_issue, const bf16x8 (&cur)[MI][3], bf16x8 (&nxt)[MI][3], int stage, int next_stage, int issue_stage) {
    f32x4 lo[MI], hi[MI];
    uint4 pl[MI][3];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][0], cur[i][2], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][1], cur[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][2], cur[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][0], cur[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][1], cur[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][0], cur[i][0], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // block j's fragment registers are dead from here: block j + 2 may land in them
      // order behind the MFMAs of block j: the split quarter first (its lgkmcnt wait then covers reads issued a block ago, not
      // the ones that follow), then the fragment reads of block j + 2, then this block's share of the wave's DMA pieces for step
      // t + STAGES - 1 (all at once they hold the wave -- and, barrier-aligned, every wave of the CU -- in the vector-memory
      // issue queue for ~370-700 cycles with the matrix pipe idle).  do_issue is a compile-time flag: a branch around the DMAs
      // is a join at which hipcc's s_waitcnt pass falls back to lgkmcnt(0).
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
          if ((NJ <= NI ? q : q * NI / NJ) == j) issue_piece(q, issue_stage);  // one piece behind each of the FIRST blocks (spread over ALL
          // blocks their last ones had a few hundred cycles to land: 41 269 rows 114 / 105 / 121 / 92 -> 110 / 96 / 112 / 84 us; two per block: worse)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) nxt[i][q] = __builtin_bit_cast(bf16x8, pl[i][q]);
  };

  // epilogue operands: the ReLU mask of a backward-data product, the bias of a forward one (clamped, unconditional loads).  One
  // tile per workgroup: fetched BEFORE the DMAs (older than every one of them: they land first, and no ordinary load sits in the
  // k-loop).  Persistent: fetched in front of a tile's LAST k-step, which is straight-line code up to the epilogue, so hipcc
  // counts the DMAs issued behind them and waits for exactly these loads.
  float4 mk[EPI == EPI_MASK ? MI : 1][NI];
  float4 bv[NI];
  auto prefetch_epi = [&]() {
    if constexpr (EPI == EPI_MASK) gemm_prefetch_mask<MI, NI>(p, mk, m0 + wm0, n0 + wn0, lane);
    if constexpr (EPI == EPI_BIAS) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        bv[j] = f4_zero();
        if (p.bias) bv[j] = *reinterpret_cast<const float4*>(p.bias + min(n0 + wn0 + j * 16 + fk * 4, p.N - 4));
      }
    }
  };
  if constexpr (!PP) prefetch_epi();
#pragma unroll
  for (int q = 0; q < STAGES - 1; ++q)
    if (q < nk) issue(q, q);
  uint64_t c_wait = 0, c_issue = 0, c_split = 0, c_mma = 0, c_t = 0, c_start = 0;
  auto tick = [&](uint64_t& acc_) {
    if constexpr (DBG) {
      __builtin_amdgcn_sched_barrier(0);
      const uint64_t now = gemm_clock();
      acc_ += now - c_t;
      c_t = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (DBG) c_start = c_t = gemm_clock();
  static_assert(STAGES >= 3, "k_gemm3w needs a 3-deep ring");
  // barrier t: every wave's pieces of steps <= t + 1 are in LDS (step t reads the first fragments of step t + 1), and every
  // wave is done with the buffer of step t - 1, which the DMAs of step t + STAGES - 1 refill; steps t + 2 .. t + STAGES - 2
  // stay in flight.  `cont`: the stream of steps continues into another tile (persistent), so the steady-state count holds to the
  // tile's end.  (Loads and stores that are not DMAs -- the epilogue's -- only make a counted wait stricter: they are older.)
  auto sync = [&](int t, bool cont) {
    const int infl = cont ? STAGES - 3 : max(0, min(t + STAGES - 2, nk - 1) - (t + 1));
    if (STAGES >= 4 && infl == STAGES - 3) gemm_wait_vmcnt_imm<(STAGES >= 4 ? STAGES - 3 : 0) * NJ>();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the builtin, not inline asm: hipcc's own s_waitcnt pass then KNOWS that no LDS read is outstanding behind the barrier --
    // with an asm wait it assumes the fragment reads of the previous step are still in flight and drains lgkmcnt(0), i.e. the
    // reads it has just issued, in front of the next block's MFMAs)
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
  };
  bf16x8 a0[MI][3], a1[MI][3];
  sync(0, false);
  {
    f32x4 lo[MI], hi[MI];
    aload(0, lo, hi);
    bload(0, 0);
    bload(0, 1);
    uint4 pl[MI][3];
#pragma unroll
    for (int c = 0; c < 4; ++c) asplit_q(c, lo, hi, pl);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) a0[i][q] = __builtin_bit_cast(bf16x8, pl[i][q]);
  }
  // (the first fragments of step t + 1 are read unconditionally: behind the last step that is a stale buffer, results dropped)
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  auto swap_back = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) a0[i][q] = a1[i][q];
  };
  // ---- epilogue (the C/D register layout of gemm_epilogue_pre): lane holds C[m0 + wm0 + 16 i + fr][n0 + wn0 + 16 j + 4 fk + 0..3]
  auto epilogue = [&]() {
    float* C = p.C;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn0 + j * 16 + fk * 4;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int mb = m0 + wm0 + i * 16, m = mb + fr;
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if constexpr (EPI == EPI_BIAS) {
          v = f4_add(v, bv[j]);
          if (p.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
          if (p.colstat) {  // (uniform) per-16-row-block column sums and squared deviations, see gemm_epilogue_pre
            const int cnt = min(16, p.M - mb);
            if (cnt > 0) {
              const bool ok = fr < cnt;
              float4 sm = ok ? v : f4_zero();
              sm.x = row16_sum(sm.x); sm.y = row16_sum(sm.y); sm.z = row16_sum(sm.z); sm.w = row16_sum(sm.w);
              const float inv = 1.f / (float)cnt;
              float4 q;
              q.x = ok ? v.x - sm.x * inv : 0.f; q.y = ok ? v.y - sm.y * inv : 0.f;
              q.z = ok ? v.z - sm.z * inv : 0.f; q.w = ok ? v.w - sm.w * inv : 0.f;
              q.x = row16_sum(q.x * q.x); q.y = row16_sum(q.y * q.y); q.z = row16_sum(q.z * q.z); q.w = row16_sum(q.w * q.w);
              if (fr == 0 && n < p.N) {
                float* cs = p.colstat + (int64_t)(mb >> 4) * 2 * p.N + n;
                *reinterpret_cast<float4*>(cs) = sm;
                *reinterpret_cast<float4*>(cs + p.N) = q;
              }
            }
          }
        }
        if constexpr (EPI == EPI_MASK) {
          const float4 k4 = mk[i][j];
          if (!(k4.x > 0.f)) v.x = 0.f;
          if (!(k4.y > 0.f)) v.y = 0.f;
          if (!(k4.z > 0.f)) v.z = 0.f;
          if (!(k4.w > 0.f)) v.w = 0.f;
        }
        if (m < p.M && n < p.N) *reinterpret_cast<float4*>(C + (int64_t)m * p.ldc + n) = v;
      }
    }
  };
  // the first nk - (STAGES - 1) steps fetch a later step, the last STAGES - 1 do not: two loops, each unrolled by two over the
  // two A-fragment register sets (one loop choosing per step makes hipcc shuffle accumulators between the variants)
  const int n_main = max(0, nk - (STAGES - 1));
  auto stamp = [&]() {
    if constexpr (DBG) asm volatile("" ::"v"(acc[MI - 1][NI - 1]), "v"(a0[0][2]), "v"(a1[0][2]), "v"(b[1][2]));
    tick(c_mma);
  };

  if constexpr (PP) {
    // ------------------------------------------------------------------------------------------------ persistent: a stream of tiles
    // (launched only with nk >= STAGES.)  Ring position of a tile's step t: (sb + t) % STAGES, sb = steps of the earlier tiles.
    int sb = 0;
    bool first = true;
    for (int q = blockIdx.x;;) {
      const int qn = q + gridDim.x;
      const int tn = (qn / gridDim.x) * gridDim.x + xcd_remap(blockIdx.x, gridDim.x, p.nxcd);  // (= qn with the block id remapped)
      const bool has_next = tn < ntiles;
      int it = 0;
      auto st = [&](int t) { return (sb + t) % STAGES; };
      for (; it + 2 <= n_main; it += 2) {
        if (!(first && it == 0)) sync(it, true);
        step(Yes{}, a0, a1, st(it), st(it + 1), st(it + STAGES - 1));
        sync(it + 1, true);
        step(Yes{}, a1, a0, st(it + 1), st(it + 2), st(it + STAGES));
      }
      if (it < n_main) {
        if (!(first && it == 0)) sync(it, true);
        step(Yes{}, a0, a1, st(it), st(it + 1), st(it + STAGES - 1));
        ++it;
        swap_back();
      }
      first = false;
      // the last STAGES - 1 steps: their DMAs belong to the next tile (its steps 0 .. STAGES - 2), the very last one is peeled so
      // that the epilogue's operands are fetched in straight-line code in front of it
      if (has_next) {
        set_pieces((tn / tiles_n) * BM, (tn % tiles_n) * BN);
        if constexpr (STAGES == 3) {
          sync(it, true);
          step(Yes{}, a0, a1, st(it), st(it + 1), st(it + 2));
          sync(it + 1, true);
          prefetch_epi();
          step(Yes{}, a1, a0, st(it + 1), st(it + 2), st(it + 3));
        } else {
          sync(it, true);
          step(Yes{}, a0, a1, st(it), st(it + 1), st(it + 3));
          sync(it + 1, true);
          step(Yes{}, a1, a0, st(it + 1), st(it + 2), st(it + 4));
          sync(it + 2, true);
          prefetch_epi();
          step(Yes{}, a0, a1, st(it + 2), st(it + 3), st(it + 5));
          swap_back();
        }
      } else {
        if constexpr (STAGES == 3) {
          sync(it, false);
          step(No{}, a0, a1, st(it), st(it + 1), 0);
          sync(it + 1, false);
          prefetch_epi();
          step(No{}, a1, a0, st(it + 1), st(it + 2), 0);
        } else {
          sync(it, false);
          step(No{}, a0, a1, st(it), st(it + 1), 0);
          sync(it + 1, false);
          step(No{}, a1, a0, st(it + 1), st(it + 2), 0);
          sync(it + 2, false);
          prefetch_epi();
          step(No{}, a0, a1, st(it + 2), st(it + 3), 0);
        }
      }
      epilogue();
      if (!has_next) break;
      // the next tile: accumulators back to zero (its fragments of step 0 are already in a0 / b[0], b[1])
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      sb = (sb + nk) % STAGES;
      q = qn;
      tile = tn;
      m0 = (tile / tiles_n) * BM;
      n0 = (tile % tiles_n) * BN;
    }
    return;
  }

  int it = 0;
  for (; it + 2 <= n_main; it += 2) {
    if (it > 0) sync(it, false);
    tick(c_wait);
    step(Yes{}, a0, a1, it % STAGES, (it + 1) % STAGES, (it + STAGES - 1) % STAGES);
    stamp();
    sync(it + 1, false);
    tick(c_wait);
    step(Yes{}, a1, a0, (it + 1) % STAGES, (it + 2) % STAGES, (it + STAGES) % STAGES);
    stamp();
  }
  if (it < n_main) {  // odd count: one more fetching step, then the sets swap back
    if (it > 0) sync(it, false);
    tick(c_wait);
    step(Yes{}, a0, a1, it % STAGES, (it + 1) % STAGES, (it + STAGES - 1) % STAGES);
    stamp();
    ++it;
    swap_back();
  }
  for (; it < nk; it += 2) {
    if (it > 0) sync(it, false);
    tick(c_wait);
    step(No{}, a0, a1, it % STAGES, (it + 1) % STAGES, 0);
    stamp();
    if (it + 1 < nk) {
      sync(it + 1, false);
      tick(c_wait);
      step(No{}, a1, a0, (it + 1) % STAGES, (it + 2) % STAGES, 0);
      stamp();
    }
  }

  if constexpr (DBG) {
    if (p.dbg && blockIdx.x < 8 && lane == 0) {
      unsigned long long* o = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
      o[0] = c_wait; o[1] = c_issue; o[2] = c_split; o[3] = c_mma; o[4] = gemm_clock() - c_start; o[5] = nk;
    }
  }
  epilogue();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, bool PP = false, bool DBG = false>
int launch_gemm3w_s(const GemmArgs& p, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * (BM * 128 + 3 * BN * 64);
  static_assert(lds <= 160 * 1024, "LDS ring too deep");
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  if constexpr (!PP && !DBG) {
    // two or more tiles per CU (a CU holds one of these workgroups): persistent workgroups that prefetch across tile boundaries.
    // Measured (tools/gemm3w_bench.cpp, 300 -> 600 / 600 -> 300 backward-data): 10 249 rows 40.7 -> 36.2 / 40.1 -> 35.4 us,
    // 41 269 rows 122 -> 114 / 135 -> 121 us; with 1.3-1.7 tiles per CU (6 747 rows, 64-row tiles) it is level or 3 % behind
    if (tiles >= 2 * num_cu() && ceil_div(p.K, 32) >= STAGES && env_knob("PGNN_GEMM3W_PERSIST", 1))
      return launch_gemm3w_s<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, true, false>(p, st);
  }
  const int grid = PP ? std::min(tiles, num_cu()) : tiles;
  allow_big_lds((const void*)k_gemm3w<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, PP, DBG>, lds);
  hipLaunchKernelGGL((k_gemm3w<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, PP, DBG>), dim3(grid), dim3(64 * WAVES_M * WAVES_N), lds, st, p);
  return check_launch("gemm3w");
}

// tile choice: the cheaper of 64x160 and 128x160 under a rounds x (k-steps + fixed) model, see below.
// PGNN_GEMM3W_CFG: 0 = 128x160 (8 x 1 waves of 16 x 160) / 3 stages, 1 = 64x160 (4 x 2 waves of 16 x 80) / 4 stages,
// 2 = 64x160 / 3 stages, 3 = 128x160 (4 x 2 waves of 32 x 80) / 3 stages
template <int EPI>
int launch_gemm3w(const GemmArgs& p, hipStream_t st) {
  int cfg = env_knob("PGNN_GEMM3W_CFG", -1);
  if (cfg < 0) {
    // one workgroup per CU (LDS), so a launch runs ceil(tiles / CUs) rounds of (k-steps x step time + per-tile prologue and
    // epilogue): 0.66 / 1.34 us per 32-deep k-step of a 64- / 128-row tile and ~3 us per tile (tools/gemm3w_bench.cpp, 2 048 rows).
    // 6 747 x 600: 212 tiles of 128 rows in one round (16 us) beat 424 of 64 in two (19); 10 249 x 600 (bio): 644 tiles of 64 in
    // three rounds (47 us) beat 324 of 128 in two (57, measured 57)
    const int64_t nk = ceil_div(p.K, 32), cus = num_cu();
    const double t64 = (double)ceil_div(ceil_div(p.M, 64) * ceil_div(p.N, 160), cus) * (0.66 * nk + 3.0);
    const double t128 = (double)ceil_div(ceil_div(p.M, 128) * ceil_div(p.N, 160), cus) * (1.34 * nk + 3.0);
    cfg = t128 <= t64 ? 0 : 1;
  }
  switch (cfg) {
    case 0: return launch_gemm3w_s<128, 160, 8, 1, 3, EPI>(p, st);
    case 2: return launch_gemm3w_s<64, 160, 4, 2, 3, EPI>(p, st);
    case 3: return launch_gemm3w_s<128, 160, 4, 2, 3, EPI>(p, st);
    case 4: return launch_gemm3w_s<64, 80, 4, 1, 3, EPI>(p, st);   // 71 KB of LDS, four waves: two workgroups share a CU
    case 5: return launch_gemm3w_s<128, 80, 8, 1, 3, EPI>(p, st);  // 94 KB: one per CU, 16 x 80 wave tiles
    default: return launch_gemm3w_s<64, 160, 4, 2, 4, EPI>(p, st);
  }
}

// fp32 matrices -> three zero-padded bf16 planes each, optionally transposed, for up to 32 matrices in one launch
// (the weights of a layer stack, once per forward / backward pass).  dst[q][r][c] = term q of src[r][c] (transposed:
// of src[c][r]); rows r < rows_out, columns c < ld with zeros from cols_out on.  32 x 32 tiles through LDS, blockIdx.y = job.
struct SplitJobs {
  const float* src[32];
  unsigned short* dst[32];
  int rows[32], cols[32], ld[32];  // rows / cols of the OUTPUT planes, their row pitch
  int transpose[32];
  long long* bump[16];  // device counters this launch increments by one (BatchNorm's num_batches_tracked), nbump of them
  int nbump;
  EncTables tabs;  // edge-encoder tables written by the blocks of job 0 (count = 0: none)
};
__global__ void __launch_bounds__(256) k_split_jobs(SplitJobs jobs) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < jobs.nbump) *jobs.bump[threadIdx.x] += 1;
  if (jobs.tabs.count && blockIdx.y == 0) {
    const int dim = jobs.tabs.dim, k = jobs.tabs.k, per = (k + 1) * dim;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < jobs.tabs.count * per; i += gridDim.x * 256) {
      const int l = i / per, q = i - l * per, r = q / dim, c = q - r * dim;
      jobs.tabs.dst[l][q] = r < k ? jobs.tabs.w[l][(int64_t)c * k + r] : jobs.tabs.b[l][c];
    }
  }
  __shared__ float tile[32][33];
  const int j = blockIdx.y, rows = jobs.rows[j], cols = jobs.cols[j], ld = jobs.ld[j];
  const bool tr = jobs.transpose[j] != 0;
  const float* __restrict__ src = jobs.src[j];
  unsigned short* __restrict__ dst = jobs.dst[j];
  const int64_t plane = (int64_t)rows * ld;
  const int tc = (ld + 31) / 32, tiles = ((rows + 31) / 32) * tc;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // element (r, c) of the output comes from src[r][c] (pitch cols) or, transposed, from src[c][r] (pitch rows)
      float v = 0.f;
      if (!tr) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) v = src[(int64_t)r * cols + c];
        tile[ty + 8 * i][tx] = v;
      } else {
        const int c = c0 + ty + 8 * i, r = r0 + tx;  // coalesced along the source's contiguous dimension (r)
        if (r < rows && c < cols) v = src[(int64_t)c * rows + r];
        tile[tx][ty + 8 * i] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      if (r < rows && c < ld) {
        const float a = tile[ty + 8 * i][tx];
        const __bf16 h = (__bf16)a;
        const float ra = a - (float)h;
        const __bf16 m = (__bf16)ra;
        const __bf16 l = (__bf16)(ra - (float)m);
        const int64_t o = (int64_t)r * ld + c;
        dst[o] = __builtin_bit_cast(unsigned short, h);
        dst[plane + o] = __builtin_bit_cast(unsigned short, m);
        dst[2 * plane + o] = __builtin_bit_cast(unsigned short, l);
      }
    }
    __syncthreads();
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES>
int launch_gemm3_s(const GemmArgs& p, int nsplit, hipStream_t st) {
  using TA = typename std::conditional<A_KMAJOR, KMajorTile<BM>, RowMajorTile<BM>>::type;
  using TB = typename std::conditional<B_KMAJOR, KMajorTile<BN>, RowMajorTile<BN>>::type;
  constexpr size_t lds = (size_t)3 * (TA::PLANE + TB::PLANE);
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N + (ONES ? 4 : 0), BN));
  if constexpr (!A_KMAJOR && !B_KMAJOR && EPI == EPI_PLAIN && BM >= 128) {
    if (nsplit > 1 && p.nxcd > 1 && env_knob("PGNN_DW_XCD_GROUP", 1) != 0) {
      allow_big_lds((const void*)k_gemm3_xg<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES>, lds);
      const int grid = tiles * (int)ceil_div(nsplit, p.nxcd) * p.nxcd;  // every XCD gets ceil(nsplit / nxcd) splits' worth of slots
      hipLaunchKernelGGL((k_gemm3_xg<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES>), dim3(grid), dim3(64 * WAVES_M * WAVES_N),
                         lds, st, p, tiles, nsplit);
      return check_launch("gemm3_xg");
    }
  }
  allow_big_lds((const void*)k_gemm3<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES>, lds);
  hipLaunchKernelGGL((k_gemm3<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES>), dim3(tiles, nsplit),
                     dim3(64 * WAVES_M * WAVES_N), lds, st, p);
  return check_launch("gemm3");
}

// 128x160 tiles (wave tile 32x80) when they give at least 3/4 of the CUs a workgroup, else 64x160 (16x80)
template <bool A_KMAJOR, bool B_KMAJOR, int EPI, bool ONES = false>
int launch_gemm3(const GemmArgs& p, int nsplit, hipStream_t st) {
  const int forced = env_knob("PGNN_GEMM3_CFG", -1);
  const bool big = forced >= 0 ? forced == 0 : ceil_div(p.M, 128) * ceil_div(p.N, 160) * nsplit * 4 >= 3 * num_cu();
  if (big) return launch_gemm3_s<128, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
  return launch_gemm3_s<64, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
}

// forward product: 1 = split-bf16 (v_mfma_f32_16x16x32_bf16 x 6), 0 = fp32 MFMA (v_mfma_f32_16x16x4_f32)
inline int gemm_mode() { return env_knob("PGNN_GEMM_SPLIT", 1); }
// Below ~160 tiles of 64x160 the split kernel's smallest tile leaves most CUs empty and the fp32-MFMA kernel's 64x64 tiles win
// (M = 1000: 12.1 / 20.4 us against 16.7 / 25.0; M = 6747: 37 / 36 against 25 / 27, tools/gemm_split_check.py)
inline bool use_split(int64_t m, int64_t n) {
  if (gemm_mode() != 1) return false;
  return env_knob("PGNN_GEMM3_CFG", -1) >= 0 || ceil_div(m, 64) * ceil_div(n, 160) >= env_knob("PGNN_GEMM3_MIN_TILES", 160);
}

// dst_j[c][r] = src_j[r][c] for up to 16 small matrices in one launch (the weights of a layer stack, transposed once
// per backward pass so that backward-data becomes a k-contiguous product): 32x32 tiles through LDS, blockIdx.y = job
struct TransposeJobs {
  const float* src[16];
  float* dst[16];
  int rows[16], cols[16];
};
__global__ void __launch_bounds__(256) k_transpose_jobs(TransposeJobs jobs) {
  __shared__ float tile[32][33];
  const int j = blockIdx.y, rows = jobs.rows[j], cols = jobs.cols[j];
  const int tc = (cols + 31) / 32, tiles = ((rows + 31) / 32) * tc;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      tile[ty + 8 * i][tx] = (r < rows && c < cols) ? jobs.src[j][(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (c < cols && r < rows) jobs.dst[j][(int64_t)c * rows + r] = tile[tx][ty + 8 * i];
    }
    __syncthreads();
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
  // (the deeper / ring variants -- PGNN_GEMM_KS = 2, 11, 12 -- measured level or behind and exist in A/B builds only: they were
  //  three quarters of this file's 224 k_gemm instances, 1.8 MB of device code)
#ifdef PGNN_AB
  const int ks = env_ks(kDefaultKS);
  if (ks == 11) return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 11>(p, nsplit, st);
  if (ks == 12) return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 12>(p, nsplit, st);
  if (ks >= 2) return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, 2>(p, nsplit, st);
#endif
  return launch_gemm_s<BM, BN, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI, ONES, kDefaultKS>(p, nsplit, st);
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
#ifdef PGNN_AB  // tiles no default path picks (PGNN_GEMM_CFG / PGNN_GEMM_WIDE reach them): A/B builds only -- the default build falls to 64x160 on 8 waves
    case T64x160: return launch_gemm<64, 160, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T128x160: return launch_gemm<128, 160, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T256x304: return launch_gemm<256, 304, 8, 1, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
#else
    case T64x160:
    case T128x160:
    case T256x304: return launch_gemm<64, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
#endif
    case T128x128: return launch_gemm<128, 128, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T64x160w8: return launch_gemm<64, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    case T320x160: return launch_gemm<320, 160, 4, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
    default: return launch_gemm<64, 64, 2, 2, A_KMAJOR, B_KMAJOR, EPI, ONES>(p, nsplit, st);
  }
}

constexpr int kWgtBK = 16;

// split count for dW = dy^T x: every SIMD should get ~2 waves, each split at least 4 k-tiles deep
// (round 5) 64x160 tiles of the split-bf16 kernel: two 55 KB workgroups are resident per CU, and a grid just above 2 x CUs -- 13
// splits x 40 tiles = 520 -- runs its last eight workgroups in a second round: as many splits as fit in ONE round instead (12 x 40 =
// 480: chem step 0.969-0.972 -> 0.953-0.960 ms, profiles/r05/dw_one_round_ab.txt; PGNN_DW_SPLIT_MODE=0 = rounded up)
inline bool one_round_splits() { return env_knob("PGNN_DW_SPLIT_MODE", 1) == 1; }
inline int weight_splits(int64_t m, int64_t k, int64_t n, int bm, int bn) {
  const int64_t tiles = ceil_div(n, bm) * ceil_div(k, bn);
  int64_t s = (bm == 64 && bn == 160 && one_round_splits()) ? std::max<int64_t>(2 * num_cu() / tiles, 1) : ceil_div(2 * num_cu(), tiles);
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
  if (use_split(m, n)) return launch_gemm3<true, true, EPI_BIAS>(p, 1, (hipStream_t)stream);
  return launch_cfg<true, true, EPI_BIAS>(pick_cfg(m, n, 0), p, 1, (hipStream_t)stream);
}

int pgnn_linear_fwd_colstats(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t m,
                             int64_t k, int64_t n, int relu, float* colstat, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && colstat,
               "linear_fwd_colstats: K, N and the leading dimensions must be multiples of 4, colstat must be given");
  GemmArgs p{};
  p.nxcd = num_xcd();
  p.A = x; p.lda = ldx; p.B = w; p.ldb = k; p.C = y; p.ldc = ldy;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.bias = bias; p.relu = relu; p.kchunk = (int)k; p.split_stride = 0;
  p.colstat = colstat;
  if (use_split(m, n)) return launch_gemm3<true, true, EPI_BIAS>(p, 1, (hipStream_t)stream);
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
  // W row-contiguous: transpose-read fragments (RowMajorTile).  28 / 27 us against 38 / 35 at 6 747 rows, 54 / 56 against
  // 74 / 67 at 16 384; level with the fp32-MFMA kernel from 65 536 rows on (the transposed-weights form keeps its 1.4x there)
  if (use_split(m, k) && m <= 65536 && env_knob("PGNN_GEMM3_BWD", 1)) {
    if (relu_out) return launch_gemm3<true, false, EPI_MASK>(p, 1, (hipStream_t)stream);
    return launch_gemm3<true, false, EPI_PLAIN>(p, 1, (hipStream_t)stream);
  }
  const TileCfg c = pick_cfg(m, k, 1);
  if (relu_out) return launch_cfg<true, false, EPI_MASK>(c, p, 1, (hipStream_t)stream);
  return launch_cfg<true, false, EPI_PLAIN>(c, p, 1, (hipStream_t)stream);
}

int pgnn_transpose_batch(const float* const* src, float* const* dst, const int64_t* rows, const int64_t* cols, int64_t count,
                         pgnn_stream stream) {
  PGNN_REQUIRE(count >= 0 && count <= 16, "transpose_batch: at most 16 matrices per call");
  if (count == 0) return PGNN_OK;
  TransposeJobs jobs{};
  int64_t most = 1;
  for (int j = 0; j < count; ++j) {
    PGNN_REQUIRE(src[j] && dst[j] && rows[j] > 0 && cols[j] > 0 && rows[j] < (1 << 30) && cols[j] < (1 << 30), "transpose_batch: bad job");
    jobs.src[j] = src[j]; jobs.dst[j] = dst[j]; jobs.rows[j] = (int)rows[j]; jobs.cols[j] = (int)cols[j];
    most = std::max(most, ceil_div(rows[j], 32) * ceil_div(cols[j], 32));
  }
  hipLaunchKernelGGL(k_transpose_jobs, dim3((int)std::min<int64_t>(most, 4096), (int)count), dim3(256), 0, (hipStream_t)stream, jobs);
  return check_launch("transpose_batch");
}

int pgnn_linear_bwd_data_t(const float* dy, int64_t lddy, const float* wt, const float* relu_out, int64_t ldr, float* dx,
                           int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0,
               "linear_bwd_data_t: K, N and the leading dimensions must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  // dx[m, kcol] = sum_nn dy[m, nn] wt[kcol, nn]: the forward product with the transposed weights as its (k-contiguous) B
  p.A = dy; p.lda = lddy; p.B = wt; p.ldb = n; p.C = dx; p.ldc = lddx;
  p.M = (int)m; p.N = (int)k; p.K = (int)n; p.mask = relu_out; p.ldmask = ldr; p.kchunk = (int)n; p.split_stride = 0;
  hipStream_t st = (hipStream_t)stream;
  if (use_split(m, k)) return relu_out ? launch_gemm3<true, true, EPI_MASK>(p, 1, st) : launch_gemm3<true, true, EPI_PLAIN>(p, 1, st);
  const TileCfg c = pick_cfg(m, k, 0);
  return relu_out ? launch_cfg<true, true, EPI_MASK>(c, p, 1, st) : launch_cfg<true, true, EPI_PLAIN>(c, p, 1, st);
}

// ---- products on pre-split weight planes (k_gemm3w) ----
size_t pgnn_weight_planes_bytes(int64_t rows, int64_t cols) {
  return align_up((size_t)3 * rows * ceil_div(cols, 32) * 32 * sizeof(unsigned short), 256);
}

// 1 where the products of a layer stack should run on pre-split planes: from 48 tiles of 64x160 (where pgnn_linear_fwd still
// takes the fp32-MFMA kernel's 64x64 tiles: 12.4 / 18.3 us against 20.3 / 20.6 at 2 048 rows) on -- 21.9 / 18.7 us
// against the split-bf16 kernel's 29.1 / 26.8 for the 600 -> 300 products at 6 747 rows, 23.6 against 26.1 for 300 -> 600; with
// persistent workgroups and early DMA issue also at 262 144 rows (forward 696 + 576 us against 666 + 599, backward-data
// 691 + 510 against 907 + 563: tools/gemm3w_bench.cpp; the first half of the round stopped at 65 536 rows, where it was level).  Bit-identical to pgnn_linear_fwd wherever THAT runs the split-bf16 kernel (from
// 160 tiles; PGNN_GEMM_WP_MIN_TILES=160 restricts the planes to that range), fp32-rounding-equal to its fp32-MFMA kernel below.
int pgnn_linear_wp_preferred(int64_t m, int64_t k, int64_t n) {
  if (env_knob("PGNN_GEMM_WP", 1) == 0 || gemm_mode() != 1 || k < 4 || k % 4 || n % 4) return 0;
  return ceil_div(m, 64) * ceil_div(n, 160) >= env_knob("PGNN_GEMM_WP_MIN_TILES", 48);
}

int pgnn_split_weights(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                       int64_t count, pgnn_stream stream) {
  return pgnn::split_weights_bump(src, dst, rows, cols, transpose, count, nullptr, 0, (hipStream_t)stream);
}
}  // extern "C"

int pgnn::split_weights_bump(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                             int64_t count, int64_t* const* bump, int nbump, hipStream_t stream, const EncTables* tabs) {
  PGNN_REQUIRE(count >= 0 && count <= 32 && nbump >= 0 && nbump <= 16, "split_weights: at most 32 matrices (and 16 counters) per call");
  PGNN_REQUIRE(!tabs || (tabs->count >= 0 && tabs->count <= 16 && tabs->dim > 0 && tabs->k > 0), "split_weights: at most 16 encoder tables");
  const bool with_tabs = tabs && tabs->count > 0;
  if (count == 0 && !with_tabs) return PGNN_OK;
  SplitJobs jobs{};
  if (with_tabs) jobs.tabs = *tabs;
  for (int j = 0; j < nbump; ++j) jobs.bump[j] = reinterpret_cast<long long*>(bump[j]);
  jobs.nbump = nbump;
  int64_t most = 1;
  for (int j = 0; j < count; ++j) {
    PGNN_REQUIRE(src[j] && dst[j] && rows[j] > 0 && cols[j] > 0 && rows[j] < (1 << 24) && cols[j] < (1 << 24), "split_weights: bad job");
    const bool tr = transpose && transpose[j];
    jobs.src[j] = src[j]; jobs.dst[j] = static_cast<unsigned short*>(dst[j]);
    jobs.rows[j] = (int)(tr ? cols[j] : rows[j]); jobs.cols[j] = (int)(tr ? rows[j] : cols[j]);
    jobs.ld[j] = (int)(ceil_div(jobs.cols[j], 32) * 32); jobs.transpose[j] = tr;
    most = std::max(most, ceil_div(jobs.rows[j], 32) * (jobs.ld[j] / 32));
  }
  if (with_tabs) most = std::max<int64_t>(most, std::min<int64_t>(ceil_div((int64_t)tabs->count * (tabs->k + 1) * tabs->dim, 256), 64));
  hipLaunchKernelGGL(k_split_jobs, dim3((int)std::min<int64_t>(most, 4096), (int)std::max<int64_t>(count, 1)), dim3(256), 0, stream, jobs);
  return check_launch("split_weights");
}

extern "C" {

int pgnn_linear_fwd_wp(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m, int64_t k,
                       int64_t n, int relu, float* colstat, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && wplanes,
               "linear_fwd_wp: K, N and the leading dimensions must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  p.A = x; p.lda = ldx; p.C = y; p.ldc = ldy;
  p.Bp = static_cast<const unsigned short*>(wplanes); p.ldbp = ceil_div(k, 32) * 32; p.bplane = n * p.ldbp;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.bias = bias; p.relu = relu; p.kchunk = (int)k; p.split_stride = 0;
  p.colstat = colstat;
  return launch_gemm3w<EPI_BIAS>(p, (hipStream_t)stream);
}

int pgnn_debug_gemm3w_profile(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m,
                              int64_t k, int64_t n, int cfg, uint64_t* buffer, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && wplanes && buffer, "debug_gemm3w_profile: bad arguments");
  GemmArgs p{};
  p.nxcd = num_xcd();
  p.A = x; p.lda = ldx; p.C = y; p.ldc = ldy;
  p.Bp = static_cast<const unsigned short*>(wplanes); p.ldbp = ceil_div(k, 32) * 32; p.bplane = n * p.ldbp;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.bias = bias; p.relu = 1; p.kchunk = (int)k;
  p.dbg = reinterpret_cast<unsigned long long*>(buffer);
#ifdef PGNN_AB  // the instrumented instances exist in A/B builds only (python -m pretrain_gnns_amd.build --ab)
  hipStream_t st = (hipStream_t)stream;
  switch (cfg) {
    case 0: return launch_gemm3w_s<128, 160, 8, 1, 3, EPI_BIAS, false, true>(p, st);
    case 4: return launch_gemm3w_s<64, 80, 4, 1, 3, EPI_BIAS, false, true>(p, st);
    default: return launch_gemm3w_s<64, 160, 4, 2, 4, EPI_BIAS, false, true>(p, st);
  }
#else
  (void)cfg;
  (void)stream;
  set_error("pgnn_debug_gemm3w_profile: instrumented kernels are compiled into A/B builds only (-DPGNN_AB)");
  return PGNN_ERR_ARG;
#endif
}

int pgnn_linear_bwd_data_wp(const float* dy, int64_t lddy, const void* wtplanes, const float* relu_out, int64_t ldr, float* dx,
                            int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && wtplanes,
               "linear_bwd_data_wp: K, N and the leading dimensions must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  // dx[m, kcol] = sum_nn dy[m, nn] W^T[kcol, nn]: the forward product with the planes of W^T [k, n] as its B
  p.A = dy; p.lda = lddy; p.C = dx; p.ldc = lddx;
  p.Bp = static_cast<const unsigned short*>(wtplanes); p.ldbp = ceil_div(n, 32) * 32; p.bplane = k * p.ldbp;
  p.M = (int)m; p.N = (int)k; p.K = (int)n; p.mask = relu_out; p.ldmask = ldr; p.kchunk = (int)n; p.split_stride = 0;
  hipStream_t st = (hipStream_t)stream;
  return relu_out ? launch_gemm3w<EPI_MASK>(p, st) : launch_gemm3w<EPI_PLAIN>(p, st);
}

// backward-weight on the split-bf16 kernel (both operands row-contiguous: transpose-read fragments) from 2 048 rows on:
// 64x160 tiles below kWeightBigRows (31 us against the fp32-MFMA kernel's 38 at 6 747 rows, 64 against 83 at 16 384), 320x160
// tiles above (the activation panels are re-read a fifth as often: 584 / 570 us against 900 at 262 144 rows, where the 64x160
// tile takes 975; the two tiles cross at ~24 k rows: 95 vs 90 us)
constexpr int64_t kWeightBigRows = 24576;
inline bool weight_split(int64_t m) { return gemm_mode() == 1 && env_knob("PGNN_GEMM3_BWD", 1) && m >= 2048; }
// ... the paired launch (both weight gradients + the bond-table columns in one launch, one fold) already from 1 024 rows: the context
// network of the context-prediction step (1 9xx rows) then takes 3 launches per layer instead of 7 (two fp32-MFMA products, fold, two
// bond-table passes): 1.495-1.549 -> 1.446-1.450 ms per step (profiles/r05/ctx_pair_min_rows_ab.txt; PGNN_DW_PAIR_MIN_ROWS)
inline bool pair_split(int64_t m) { return gemm_mode() == 1 && env_knob("PGNN_GEMM3_BWD", 1) && m >= env_knob("PGNN_DW_PAIR_MIN_ROWS", 1024); }

size_t pgnn_linear_bwd_weight_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  const TileCfg c = weight_cfg(m);
  const int64_t splits = std::max({weight_splits(m, k, n, kCfg[c].bm, kCfg[c].bn), weight_splits(m, k, n, 64, 160), weight_splits(m, k, n, 320, 160)});
  return align_up((size_t)splits * (n * k + n) * sizeof(float), 256) + 256 + align_up((size_t)(n + k) * sizeof(uint32_t), 256);  // (+ column maxima)
}

namespace {
// what is left to do after the split-K product of one weight gradient: dst = sum over `used` partial matrices
struct ReduceJob {
  const float* partial;
  int used;
  int64_t stride;
  float* dw;
  int64_t n4a;
  float* db;
  int64_t n4b;
  float* g;     // the [n][12] products of the twelve extra columns (linear_bwd_weight_pair_ext), behind db in every partial
  int64_t n4c;
};
struct ReduceJobs {
  ReduceJob j[2];
};
// the same fold as k_splitk_reduce for two weight gradients in one launch (blockIdx.y picks the job)
__global__ void __launch_bounds__(256) k_splitk_reduce_jobs(ReduceJobs jobs) {
  const ReduceJob r = jobs.j[blockIdx.y];
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < r.n4a + r.n4b + r.n4c; q += (int64_t)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(r.partial)[q];
    int z = 1;
    for (; z + 4 <= r.used; z += 4) {  // four independent loads in flight, added in split order
      const float4 v0 = reinterpret_cast<const float4*>(r.partial + (z + 0) * r.stride)[q];
      const float4 v1 = reinterpret_cast<const float4*>(r.partial + (z + 1) * r.stride)[q];
      const float4 v2 = reinterpret_cast<const float4*>(r.partial + (z + 2) * r.stride)[q];
      const float4 v3 = reinterpret_cast<const float4*>(r.partial + (z + 3) * r.stride)[q];
      s = f4_add(f4_add(f4_add(f4_add(s, v0), v1), v2), v3);
    }
    for (; z < r.used; ++z) s = f4_add(s, reinterpret_cast<const float4*>(r.partial + z * r.stride)[q]);
    if (q < r.n4a) reinterpret_cast<float4*>(r.dw)[q] = s;
    else if (q < r.n4a + r.n4b) reinterpret_cast<float4*>(r.db)[q - r.n4a] = s;
    else reinterpret_cast<float4*>(r.g)[q - r.n4a - r.n4b] = s;
  }
}

// ---- column maxima for the two-plane weight gradients (round 6) ----------------------------------------------------------------
// out[c] = bit pattern of max_r |x[r][c]| for up to four matrices of n rows in one launch: the power-of-two column scales of
// gemm3_body<TWO>.  A block OWNS four float4 columns (64 bytes of every row) of one matrix for ALL rows -- 256 row lanes, eight
// rows' loads in flight per lane, the lanes folded through LDS -- and stores its sixteen maxima: no atomics, nothing to clear.  (The
// first version split the rows over blocks and folded by atomic maximum: 211 blocks arriving together serialise on the 113 cache
// lines of the four vectors -- 87 us for a pass that reads 48 MB out of L2.)  Non-negative floats order like unsigned integers; a
// NaN's pattern is above every number's and leaves the column unscaled, as an inf does.  For the row counts where the operands sit
// in L2 / MALL (the paired launch: below 24 576 rows).
struct ColmaxJob {
  const float* x;
  int64_t ld;
  int cols;
  uint32_t* out;
};
struct ColmaxJobs {
  ColmaxJob j[4];
  int first_block[5];  // blocks first_block[i] .. first_block[i + 1] - 1 belong to matrix i
};
constexpr int kColmaxThreads = 1024;
__global__ void __launch_bounds__(kColmaxThreads) k_colmax_jobs(ColmaxJobs jobs, int n) {
  int z = 0;
  while (z < 3 && (int)blockIdx.x >= jobs.first_block[z + 1]) ++z;
  const ColmaxJob jb = jobs.j[z];
  const int d4 = jb.cols >> 2, t = threadIdx.x, q = t & 3, rl = t >> 2;  // a wave = 16 rows x 64 bytes
  const int c4 = ((int)blockIdx.x - jobs.first_block[z]) * 4 + q;
  __shared__ uint4 red[kColmaxThreads];
  uint4 m = make_uint4(0u, 0u, 0u, 0u);
  auto fold = [&](const float4& v) {
    m.x = max(m.x, __float_as_uint(fabsf(v.x))); m.y = max(m.y, __float_as_uint(fabsf(v.y)));
    m.z = max(m.z, __float_as_uint(fabsf(v.z))); m.w = max(m.w, __float_as_uint(fabsf(v.w)));
  };
  if (c4 < d4) {
    constexpr int L = kColmaxThreads / 4;  // row lanes
    const float* base = jb.x + 4 * c4;
    int r = rl;
    for (; r + 7 * L < n; r += 8 * L) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (int64_t)(r + u * L) * jb.ld);
#pragma unroll
      for (int u = 0; u < 8; ++u) fold(v[u]);
    }
    for (; r < n; r += L) fold(*reinterpret_cast<const float4*>(base + (int64_t)r * jb.ld));
  }
  red[t] = m;
  __syncthreads();
  for (int s = kColmaxThreads / 2; s >= 4; s >>= 1) {  // (lanes t and t + s own the same column quad: s is a multiple of 4)
    if (t < s) {
      const uint4 o = red[t + s];
      m.x = max(m.x, o.x); m.y = max(m.y, o.y); m.z = max(m.z, o.z); m.w = max(m.w, o.w);
      red[t] = m;
    }
    __syncthreads();
  }
  if (t < 4 && c4 < d4) *reinterpret_cast<uint4*>(jb.out + 4 * c4) = m;
}
int launch_colmax(const ColmaxJob* jobs, int count, int64_t n, hipStream_t st) {
  ColmaxJobs cj{};
  int blocks = 0;
  for (int i = 0; i < 4; ++i) {
    cj.first_block[i] = blocks;
    if (i < count) {
      PGNN_REQUIRE(jobs[i].cols % 4 == 0 && jobs[i].cols > 0 && jobs[i].ld % 4 == 0 && (reinterpret_cast<uintptr_t>(jobs[i].out) & 15) == 0,
                   "colmax: bad shape");
      cj.j[i] = jobs[i];
      blocks += (int)ceil_div(jobs[i].cols / 4, 4);
    }
  }
  cj.first_block[4] = blocks;
  hipLaunchKernelGGL(k_colmax_jobs, dim3(blocks), dim3(kColmaxThreads), 0, st, cj, (int)n);
  return check_launch("colmax");
}

// ---- the bond-table gradient of a chem GIN layer without a pass of its own (round 5) ------------------------------------------
// chem/model.py:37-52 under autograd: demb [9, D] = cfeat^T [9, n] . dagg [n, D] with dagg = dhid . W1, i.e. (cfeat^T . dhid) . W1.
// G = dhid^T . cfeat [2D, 9] is twelve more columns of the dW1 product (dhid^T . [agg | 1 | cfeat]): they sit in the column padding of
// its last tile (300 + 4 + 12 <= 320), so the launch that runs anyway computes them for nothing, the fold of the split-K partials
// folds them too, and what is left is [9, 2D] x [2D, D] -- 3 MFLOP, one small launch for every layer of a network together.
// cfeat comes padded to 12 floats a row (float4 staging units): k_pad_rowfeat12, once per backward.
__global__ void __launch_bounds__(256) k_pad_rowfeat12(const float* __restrict__ cfeat, int kc, float* __restrict__ out, int64_t n) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n * 3; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q / 3;
    const int c = (int)(q - i * 3) * 4;
    const float* src = cfeat + i * kc;
    float4 v;
    v.x = c + 0 < kc ? src[c + 0] : 0.f;
    v.y = c + 1 < kc ? src[c + 1] : 0.f;
    v.z = c + 2 < kc ? src[c + 2] : 0.f;
    v.w = c + 3 < kc ? src[c + 3] : 0.f;
    reinterpret_cast<float4*>(out)[q] = v;
  }
}
struct BondJobs {
  BondTableJob j[kMaxBondJobs];
};
// demb[t][c] = sum_r G[r][t] W1[r][c], float64 accumulators (the pass it replaces folds its block partials in float64 too).  A block:
// 64 columns x four of the twelve t; sixteen waves, each over every sixteenth row with eight rows' loads in flight (a loop of one
// load and its use per trip pays a full L2 round trip per row: 30 us on the backward's tail in the first version), folded through
// LDS in wave order.
constexpr int kBondWaves = 16;
__global__ void __launch_bounds__(64 * kBondWaves) k_bond_tables_from_g(BondJobs jobs, int rows, int dim, int kc) {
  const BondTableJob jb = jobs.j[blockIdx.y];
  __shared__ double red[kBondWaves][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t0 = 4 * blockIdx.z;
  const int c = blockIdx.x * 64 + lane, cc = min(c, dim - 1);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int U = 8;
  for (int r0 = w; r0 < rows; r0 += kBondWaves * U) {
    float wv[U];
    float4 gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(r0 + kBondWaves * u, rows - 1);
      wv[u] = jb.w[(int64_t)r * jb.ldw + cc];
      gv[u] = *reinterpret_cast<const float4*>(jb.g + (int64_t)r * 12 + t0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + kBondWaves * u < rows) {
        const double wd = (double)wv[u];
        acc[0] += (double)gv[u].x * wd;
        acc[1] += (double)gv[u].y * wd;
        acc[2] += (double)gv[u].z * wd;
        acc[3] += (double)gv[u].w * wd;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[w][k][lane] = acc[k];
  __syncthreads();
  if (threadIdx.x < 4 * 64) {
    const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
    double s = red[0][k][l];
#pragma unroll
    for (int ww = 1; ww < kBondWaves; ++ww) s += red[ww][k][l];
    const int col = blockIdx.x * 64 + l, t = t0 + k;
    if (col < dim && t < kc) jb.demb[(int64_t)t * jb.ldd + col] = (float)s;
  }
}

// the split-K product of dW [n, k] = dy^T x (+ db as the ones column); job.used == 1: written in place, nothing left to fold
int weight_product(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db, int64_t m, int64_t k, int64_t n,
                   void* ws, hipStream_t st, ReduceJob& job) {
  Carver cv(ws);
  const TileCfg cfg = weight_cfg(m);
  const bool split3 = weight_split(m);
  const bool big3 = split3 && m >= kWeightBigRows;
  // (round 6, A/B) PGNN_DW_BIG_TILE: the tile of the large weight gradients -- 320 (shipped: 104 KB of LDS, ONE workgroup per CU, every
  // operand byte wanted twice), 160 (10 waves, 60 KB: two workgroups per CU overlap each other's staging and product phases; operand
  // bytes wanted 2-4 times) or 128 (8 waves, 60 KB)
#ifdef PGNN_AB  // (measured level / slower, profiles/r06/dw_two_planes_ab.txt: A/B builds only)
  const int big_tile = big3 ? env_knob("PGNN_DW_BIG_TILE", 320) : 64;
#else
  const int big_tile = big3 ? 320 : 64;
#endif
  const int bm_sel = big3 ? (big_tile == 160 ? 160 : big_tile == 128 ? 128 : 320) : 64;
  const int nsplit = split3 ? weight_splits(m, k, n, bm_sel, 160) : weight_splits(m, k, n, kCfg[cfg].bm, kCfg[cfg].bn);
  float* partial = cv.take<float>((size_t)nsplit * (n * k + n));
  GemmArgs p{};
  p.nxcd = num_xcd();
  // C = dW [n, k] ; reduction over rows m ; A(nout, r) = dy[r*lddy + nout] ; B(kcol, r) = x[r*ldx + kcol]
  // db[nout] = sum_r dy[r, nout] rides along as the "ones column" of B.
  p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx;
  p.M = (int)n; p.N = (int)k; p.K = (int)m;
  int64_t chunk = ceil_div(m, nsplit);
  chunk = split3 ? ceil_div(chunk, 32) * 32 : ceil_div(chunk, kWgtBK) * kWgtBK;  // whole k-steps of the kernel in use
  p.kchunk = (int)chunk;
  const int used = (int)ceil_div(m, chunk);
  const bool direct = used == 1;
  p.C = direct ? dw : partial;
  p.ldc = k;
  p.split_stride = direct ? 0 : n * k + n;
  p.colsum = direct ? db : partial + n * k;
  job = ReduceJob{partial, used, n * k + n, dw, n * k / 4, db, db ? n / 4 : 0};
#ifdef PGNN_AB
  if (split3 && big3 && bm_sel == 160 && db) return launch_gemm3_s<160, 160, 5, 2, false, false, EPI_PLAIN, true>(p, used, st);
  if (split3 && big3 && bm_sel == 128 && db) return launch_gemm3_s<128, 160, 4, 2, false, false, EPI_PLAIN, true>(p, used, st);
#endif
  if (split3 && big3)
    return db ? launch_gemm3_s<320, 160, 4, 2, false, false, EPI_PLAIN, true>(p, used, st)
              : launch_gemm3_s<320, 160, 4, 2, false, false, EPI_PLAIN, false>(p, used, st);
  if (split3)
    return db ? launch_gemm3_s<64, 160, 4, 2, false, false, EPI_PLAIN, true>(p, used, st)
              : launch_gemm3_s<64, 160, 4, 2, false, false, EPI_PLAIN, false>(p, used, st);
  return db ? launch_cfg<false, false, EPI_PLAIN, true>(cfg, p, used, st) : launch_cfg<false, false, EPI_PLAIN, false>(cfg, p, used, st);
}
}  // namespace

int pgnn_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db,
                           int64_t m, int64_t k, int64_t n, void* ws, size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0,
               "linear_bwd_weight: K, N and leading dimensions must be multiples of 4");
  if (ws_bytes < pgnn_linear_bwd_weight_workspace_bytes(m, k, n)) {
    set_error("linear_bwd_weight workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  ReduceJob job;
  if (int rc = weight_product(dy, lddy, x, ldx, dw, db, m, k, n, ws, st, job)) return rc;
  if (job.used > 1)
    hipLaunchKernelGGL(k_splitk_reduce, dim3((int)std::min<int64_t>(ceil_div(job.n4a + job.n4b, 256), 1024)), dim3(256), 0, st,
                       job.partial, job.used, job.stride, dw, job.n4a, db, job.n4b);
  return check_launch("linear_bwd_weight");
}

int pgnn_linear_bwd_weight_pair(const float* dy_a, int64_t lddy_a, const float* x_a, int64_t ldx_a, float* dw_a, float* db_a, int64_t k_a,
                                int64_t n_a, const float* dy_b, int64_t lddy_b, const float* x_b, int64_t ldx_b, float* dw_b, float* db_b,
                                int64_t k_b, int64_t n_b, int64_t m, void* ws, size_t ws_bytes, pgnn_stream stream) {
  return pgnn::linear_bwd_weight_pair_ext(dy_a, lddy_a, x_a, ldx_a, dw_a, db_a, k_a, n_a, dy_b, lddy_b, x_b, ldx_b, dw_b, db_b, k_b, n_b, m, ws,
                                          ws_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

}  // extern "C"

// splits over the rows of the paired weight gradients: as many as fit in one round of two resident workgroups per CU (see weight_splits)
static inline int64_t pair_splits(int64_t tiles) {
  const int64_t slots = 2 * num_cu();
  return one_round_splits() ? std::max<int64_t>(slots / tiles, 1) : ceil_div(slots, tiles);
}
// the conditions under which linear_bwd_weight_pair_ext takes its one-launch path AND has room for twelve extra columns in product b
bool pgnn::linear_bwd_weight_pair_ext_ok(int64_t m, int64_t k_a, int64_t n_a, int64_t k_b, int64_t n_b) {
  if (!(pair_split(m) && m < kWeightBigRows && env_knob("PGNN_DW_PAIR", 1) != 0 && env_knob("PGNN_BOND_IN_DW", 1) != 0)) return false;
  if (k_b + 16 > ceil_div(k_b + 4, 160) * 160) return false;  // no column padding to ride in
  const int64_t tiles_a = ceil_div(n_a, 64) * ceil_div(k_a + 4, 160), tiles_b = ceil_div(n_b, 64) * ceil_div(k_b + 4, 160);
  int64_t splits = pair_splits(tiles_a + tiles_b);
  splits = std::max<int64_t>(std::min<int64_t>(splits, std::max<int64_t>(m / (4 * 32), 1)), 1);
  const int64_t chunk = ceil_div(ceil_div(m, splits), 32) * 32;
  const int64_t used = ceil_div(m, chunk);
  const size_t cma = align_up((size_t)(n_a + k_a) * sizeof(uint32_t), 256), cmb = align_up((size_t)(n_b + k_b) * sizeof(uint32_t), 256);
  return used > 1 && (size_t)used * (n_a * k_a + n_a) * sizeof(float) + cma <= pgnn_linear_bwd_weight_workspace_bytes(m, k_a, n_a) &&
         (size_t)used * (n_b * k_b + n_b + 12 * n_b) * sizeof(float) + cmb <= pgnn_linear_bwd_weight_workspace_bytes(m, k_b, n_b);
}

int pgnn::pad_rowfeat12(const float* cfeat, int64_t kc, float* out12, int64_t n, hipStream_t st) {
  PGNN_REQUIRE(cfeat && out12 && n > 0 && kc > 0 && kc <= 12, "pad_rowfeat12: bad arguments");
  hipLaunchKernelGGL(k_pad_rowfeat12, dim3((int)std::min<int64_t>(ceil_div(n * 3, 256), 2048)), dim3(256), 0, st, cfeat, (int)kc, out12, n);
  return check_launch("pad_rowfeat12");
}

int pgnn::bond_tables_from_g(const BondTableJob* jobs, int count, int64_t rows, int64_t dim, int64_t kc, hipStream_t st) {
  PGNN_REQUIRE(jobs && count > 0 && count <= kMaxBondJobs && rows > 0 && dim > 0 && kc > 0 && kc <= 12, "bond_tables_from_g: bad arguments");
  BondJobs bj{};
  for (int i = 0; i < count; ++i) bj.j[i] = jobs[i];
  hipLaunchKernelGGL(k_bond_tables_from_g, dim3((int)ceil_div(dim, 64), count, (int)ceil_div(kc, 4)), dim3(64 * kBondWaves), 0, st, bj, (int)rows, (int)dim,
                     (int)kc);
  return check_launch("bond_tables_from_g");
}

// pgnn_linear_bwd_weight_pair; with cfeat12 [m][12] and g_out [n_b][12] it also leaves g_out = dy_b^T . cfeat12 where the one-launch
// path runs (*g_done says whether it did -- the caller takes its own pass otherwise)
int pgnn::linear_bwd_weight_pair_ext(const float* dy_a, int64_t lddy_a, const float* x_a, int64_t ldx_a, float* dw_a, float* db_a, int64_t k_a,
                                     int64_t n_a, const float* dy_b, int64_t lddy_b, const float* x_b, int64_t ldx_b, float* dw_b, float* db_b,
                                     int64_t k_b, int64_t n_b, int64_t m, void* ws, size_t ws_bytes, hipStream_t stream, const float* cfeat12,
                                     float* g_out, bool* g_done) {
  if (g_done) *g_done = false;
  PGNN_REQUIRE(m > 0 && k_a > 0 && n_a > 0 && k_b > 0 && n_b > 0 && (k_a | n_a | k_b | n_b | lddy_a | ldx_a | lddy_b | ldx_b) % 4 == 0,
               "linear_bwd_weight_pair: K, N and leading dimensions must be multiples of 4");
  const size_t wa = pgnn_linear_bwd_weight_workspace_bytes(m, k_a, n_a), wb = pgnn_linear_bwd_weight_workspace_bytes(m, k_b, n_b);
  if (ws_bytes < wa + wb) {
    set_error("linear_bwd_weight_pair workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  ReduceJobs jobs{};
  int rc;
  // both products in one launch of 64x160 tiles (k_gemm3_pair) where the single products would take that tile and split:
  // PGNN_DW_PAIR=0 = two launches (round 3; bit-identical to two pgnn_linear_bwd_weight calls)
  if (pair_split(m) && m < kWeightBigRows && db_a && db_b && env_knob("PGNN_DW_PAIR", 1) != 0) {
    const int64_t tiles_a = ceil_div(n_a, 64) * ceil_div(k_a + 4, 160), tiles_b = ceil_div(n_b, 64) * ceil_div(k_b + 4, 160);
    int64_t splits = pair_splits(tiles_a + tiles_b);
    splits = std::max<int64_t>(std::min<int64_t>(splits, std::max<int64_t>(m / (4 * 32), 1)), 1);
    const int64_t chunk = ceil_div(ceil_div(m, splits), 32) * 32;
    const int used = (int)ceil_div(m, chunk);
    const bool ext = cfeat12 && g_out && g_done && linear_bwd_weight_pair_ext_ok(m, k_a, n_a, k_b, n_b);
    const int64_t gcols = ext ? 12 : 0;  // floats per row of product b's partials behind its bias gradient
    // (round 6) PGNN_DW_2P=1: the two products on two fp16 planes under power-of-two COLUMN scales (gemm3_body<TWO>); the four column
    // maxima vectors live at the end of the two workspace halves and are taken by one launch over the four operands (k_colmax_jobs)
    const size_t cma = align_up((size_t)(n_a + k_a) * sizeof(uint32_t), 256), cmb = align_up((size_t)(n_b + k_b) * sizeof(uint32_t), 256);
    // Built and measured in round 6, OFF by default (profiles/r06/dw_two_planes_ab.txt): at 6 740 rows the launch is bound by its L2
    // traffic (64-column tiles re-read each operand 4-10 times: 224 MB per layer) and its staging, not by the matrix pipes -- 52-60 us
    // against 48 for three bf16 planes, plus 26 us for the column maxima; and a column scale cannot survive a column whose large
    // entries meet zeros of the other operand (one row 2^30 times the rest under a ReLU'd partner: error 3.9e-4 of the |a|.|b| bound
    // against 6e-7 -- every other family is at or below the three-plane kernel's error).
    const bool two = env_knob("PGNN_DW_2P", 0) != 0;
    if (used > 1 && (size_t)used * (n_a * k_a + n_a) * sizeof(float) + cma <= wa &&
        (size_t)used * (n_b * k_b + n_b + gcols * n_b) * sizeof(float) + cmb <= wb) {
      GemmArgs2 q{};
      float* parts[2] = {static_cast<float*>(ws), reinterpret_cast<float*>(static_cast<char*>(ws) + wa)};
      uint32_t* cms[2] = {reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + wa - cma), reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + wa + wb - cmb)};
      const float* dys[2] = {dy_a, dy_b};
      const float* xs[2] = {x_a, x_b};
      const int64_t lddys[2] = {lddy_a, lddy_b}, ldxs[2] = {ldx_a, ldx_b}, ks[2] = {k_a, k_b}, ns[2] = {n_a, n_b};
      float* dws[2] = {dw_a, dw_b};
      float* dbs[2] = {db_a, db_b};
      for (int z = 0; z < 2; ++z) {
        GemmArgs& p = q.a[z];
        p.nxcd = num_xcd();
        p.A = dys[z]; p.lda = lddys[z]; p.B = xs[z]; p.ldb = ldxs[z];
        p.M = (int)ns[z]; p.N = (int)ks[z]; p.K = (int)m;
        p.kchunk = (int)chunk;
        p.C = parts[z]; p.ldc = ks[z];
        const int64_t gz = z == 1 ? gcols : 0;
        p.split_stride = ns[z] * ks[z] + ns[z] + gz * ns[z];
        p.colsum = parts[z] + ns[z] * ks[z];
        jobs.j[z] = ReduceJob{parts[z], used, p.split_stride, dws[z], ns[z] * ks[z] / 4, dbs[z], ns[z] / 4, nullptr, 0};
        if (gz) {
          p.F = cfeat12;
          p.extra = parts[z] + ns[z] * ks[z] + ns[z];
          jobs.j[z].g = g_out;
          jobs.j[z].n4c = gz * ns[z] / 4;
        }
        if (two) {  // A = dy [m rows][n columns], B = x [m rows][k columns]
          p.a_colmax = cms[z];
          p.b_colmax = cms[z] + ns[z];
        }
      }
      q.tiles[0] = (int)tiles_a; q.tiles[1] = (int)tiles_b;
#ifdef PGNN_AB  // (two register stages of the staging loads measured level: 0.943-0.944 against 0.925-0.937 ms per step; A/B builds only)
      const int pfd = env_knob("PGNN_DW_PFD", 1) == 2 ? 2 : 1;
#else
      const int pfd = 1;
#endif
      if (two) {
        const ColmaxJob cj[4] = {{dy_a, lddy_a, (int)n_a, cms[0]}, {x_a, ldx_a, (int)k_a, cms[0] + n_a},
                                 {dy_b, lddy_b, (int)n_b, cms[1]}, {x_b, ldx_b, (int)k_b, cms[1] + n_b}};
        if ((rc = launch_colmax(cj, 4, m, st))) return rc;
      }
      // (Round 4 built and removed a two-plane version whose workgroups took the column maxima of THEIR chunk of rows in a pass in front
      // of the k-loop: 78.7 us against 50.4, profiles/r04/wgrad2p_and_ctx_two_streams_ab.txt.)
      // (512 workgroups, two per CU: 384 / 768 / 256 aimed at, or one resident per CU, all measured slower -- profiles/r05/dw_pair_grid_ab.txt)
      const dim3 grid((int)std::max(tiles_a, tiles_b), used, 2);
      auto launch = [&](auto ext_tag, auto two_tag, auto pfd_tag) {
        constexpr bool E = decltype(ext_tag)::value, T2 = decltype(two_tag)::value;
        constexpr int PF = decltype(pfd_tag)::value;
        constexpr size_t lds = (size_t)(T2 ? 2 : 3) * (RowMajorTile<64>::PLANE + RowMajorTile<160>::PLANE);
        allow_big_lds((const void*)k_gemm3_pair<64, 160, 4, 2, false, false, EPI_PLAIN, true, E, T2, PF>, lds);
        hipLaunchKernelGGL((k_gemm3_pair<64, 160, 4, 2, false, false, EPI_PLAIN, true, E, T2, PF>), grid, dim3(512), lds, st, q);
      };
      using Tt = std::true_type;
      using Ff = std::false_type;
      using P1 = std::integral_constant<int, 1>;
#ifdef PGNN_AB
      using P2 = std::integral_constant<int, 2>;
#endif
      bool launched = false;
#ifdef PGNN_AB
      if (pfd == 2) {
        if (ext) { if (two) launch(Tt{}, Tt{}, P2{}); else launch(Tt{}, Ff{}, P2{}); }
        else     { if (two) launch(Ff{}, Tt{}, P2{}); else launch(Ff{}, Ff{}, P2{}); }
        launched = true;
      }
#endif
      (void)pfd;
      if (!launched) {
        if (ext) { if (two) launch(Tt{}, Tt{}, P1{}); else launch(Tt{}, Ff{}, P1{}); }
        else     { if (two) launch(Ff{}, Tt{}, P1{}); else launch(Ff{}, Ff{}, P1{}); }
      }
      if (ext) *g_done = true;
      const int64_t work = std::max(jobs.j[0].n4a + jobs.j[0].n4b + jobs.j[0].n4c, jobs.j[1].n4a + jobs.j[1].n4b + jobs.j[1].n4c);
      hipLaunchKernelGGL(k_splitk_reduce_jobs, dim3((int)std::min<int64_t>(ceil_div(work, 256), 1024), 2), dim3(256), 0, st, jobs);
      return check_launch("linear_bwd_weight_pair");
    }
  }
  if ((rc = weight_product(dy_a, lddy_a, x_a, ldx_a, dw_a, db_a, m, k_a, n_a, ws, st, jobs.j[0]))) return rc;
  if ((rc = weight_product(dy_b, lddy_b, x_b, ldx_b, dw_b, db_b, m, k_b, n_b, static_cast<char*>(ws) + wa, st, jobs.j[1]))) return rc;
  if (jobs.j[0].used > 1 && jobs.j[1].used > 1) {  // both split (the same m: they split together in practice)
    const int64_t work = std::max(jobs.j[0].n4a + jobs.j[0].n4b, jobs.j[1].n4a + jobs.j[1].n4b);
    hipLaunchKernelGGL(k_splitk_reduce_jobs, dim3((int)std::min<int64_t>(ceil_div(work, 256), 1024), 2), dim3(256), 0, st, jobs);
  } else {
    for (const ReduceJob& r : jobs.j)
      if (r.used > 1)
        hipLaunchKernelGGL(k_splitk_reduce, dim3((int)std::min<int64_t>(ceil_div(r.n4a + r.n4b, 256), 1024)), dim3(256), 0, st, r.partial,
                           r.used, r.stride, r.dw, r.n4a, r.db, r.n4b);
  }
  return check_launch("linear_bwd_weight_pair");
}

// ==================================================================================================================================
// The planes products on TWO fp16 planes (PGNN_GEMM_2P=1; measured as a prototype at the end of round 3, DESIGN 8.1 and
// profiles/r03/gemm2p_probe.txt): x = (h1 + h2) / s with a power-of-two scale s per ROW of each operand -- the row's largest
// magnitude lands in [2^13, 2^14), so h2 = fp16(s x - h1) stays out of fp16's subnormals for every element that matters -- needs
// three v_mfma_f32_16x16x32_f16 per accumulator and k-step (h1 h2, h2 h1, h1 h1) where three bf16 planes need six, at a plain fp32
// product's error (tools/two_plane_numerics.py).  Scales and the epilogue's rescale are exact.
//   weights: pgnn::split_weights_2p writes [2][rows][ld] fp16 planes of s W followed by 1/s per row (inside the room of three bf16 planes);
//   activations: every workgroup takes the maxima of its own rows in a pass in front of its k-loop (no producer has to supply them);
//   k_gemm2pw = k_gemm3w's one-tile-per-workgroup path with the plane count, the split and the epilogue's rescale changed.
// ==================================================================================================================================
namespace pgnn {
namespace {

// one wave per output row (k_split_jobs' jobs: dst row r = src row r, or src column r when transposed): the row's maximum, then
// the two planes of s W (zero from `cols` to `ld`) and 1/s behind the planes.  Rows that are contiguous in the source (the forward's
// planes: this launch opens the forward pass, on the caller's stream) are read ONCE, as float4, and held in registers (up to
// 1 024 columns); the transposed jobs (backward, on the side stream, beside the top layer's BatchNorm backward) re-read.
// zero_words: a region this launch clears (the row-maximum words the pass's products accumulate into)
__global__ void __launch_bounds__(256) k_split2p_jobs(SplitJobs jobs, uint32_t* __restrict__ zero_ptr, int64_t zero_words) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < jobs.nbump) *jobs.bump[threadIdx.x] += 1;
  for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_words; i += (int64_t)gridDim.x * gridDim.y * 256)
    zero_ptr[i] = 0u;
  if (jobs.tabs.count && blockIdx.y == 0) {
    const int dim = jobs.tabs.dim, k = jobs.tabs.k, per = (k + 1) * dim;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < jobs.tabs.count * per; i += gridDim.x * 256) {
      const int l = i / per, q = i - l * per, r = q / dim, c = q - r * dim;
      jobs.tabs.dst[l][q] = r < k ? jobs.tabs.w[l][(int64_t)c * k + r] : jobs.tabs.b[l][c];
    }
  }
  const int j = blockIdx.y, rows = jobs.rows[j], cols = jobs.cols[j], ld = jobs.ld[j];
  const bool tr = jobs.transpose[j] != 0;
  const float* __restrict__ src = jobs.src[j];
  unsigned short* __restrict__ dst = jobs.dst[j];
  if (!src) return;
  const int64_t plane = (int64_t)rows * ld;
  float* __restrict__ binv = reinterpret_cast<float*>(dst + 2 * plane);
  const int lane = threadIdx.x & 63;
  if (!tr && cols % 4 == 0 && cols <= 1024) {
    const int c4n = cols >> 2, ld4 = ld >> 2;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
      const float4* __restrict__ row = reinterpret_cast<const float4*>(src + (int64_t)r * cols);
      float4 v[4];
      float mx = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c4 = lane + 64 * u;
        v[u] = c4 < c4n ? row[c4] : f4_zero();
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      float sc, inv;
      pow2_scales(mx, sc, inv);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c4 = lane + 64 * u;
        if (c4 < ld4) {  // (zero from cols to ld: v[u] is zero there)
          uint32_t h0, l0, h1, l1;
          split2(v[u].x * sc, v[u].y * sc, h0, l0);
          split2(v[u].z * sc, v[u].w * sc, h1, l1);
          *reinterpret_cast<uint2*>(dst + (int64_t)r * ld + 4 * c4) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + plane + (int64_t)r * ld + 4 * c4) = make_uint2(l0, l1);
        }
      }
      if (lane == 0) binv[r] = inv;
    }
    return;
  }
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
    float mx = 0.f;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, fabsf(tr ? src[(int64_t)c * rows + r] : src[(int64_t)r * cols + c]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sc, inv;
    pow2_scales(mx, sc, inv);
    for (int c = lane; c < ld; c += 64) {
      const float v = c < cols ? (tr ? src[(int64_t)c * rows + r] : src[(int64_t)r * cols + c]) * sc : 0.f;
      const _Float16 h = (_Float16)v;
      const _Float16 l = (_Float16)low_part(v, h);
      dst[(int64_t)r * ld + c] = __builtin_bit_cast(unsigned short, h);
      dst[plane + (int64_t)r * ld + c] = __builtin_bit_cast(unsigned short, l);
    }
    if (lane == 0) binv[r] = inv;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) __attribute__((amdgpu_waves_per_eu(2, 2))) k_gemm2pw(GemmArgs p) {
  constexpr int BK = 32, NPL = 2;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int A_BYTES = BM * 128, B_PLANE = BN * 64, STAGE = A_BYTES + NPL * B_PLANE;
  constexpr int PA = BM / 8, PB = BN / 16;
  constexpr int NP = PA + NPL * PB, NJ = (NP + NW - 1) / NW;
  constexpr int TPR = (64 * NW) / BM;  // threads per row of the row-maximum pass
  static_assert(NI >= 5 && STAGES >= 3 && (TPR & (TPR - 1)) == 0 && TPR <= 64, "tile shape");

  extern __shared__ __align__(16) unsigned char smem2p[];
  __shared__ float rowmax[BM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x, p.nxcd);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + BK - 1) / BK;
  const float* __restrict__ b_inv = reinterpret_cast<const float*>(p.Bp + 2 * p.bplane);

  // the maxima of this tile's rows of A: from the producer of A when it left them (a_amax: the previous product's epilogue), else
  // thread t takes row t / TPR, float4 columns t % TPR, + TPR, ...; eight loads in flight per round, all of them older than the
  // first DMA (that pass costs 1.7 / 2.3 us of a 17 us product at K = 300 / 600: every workgroup of a row panel repeats it)
  if (p.a_amax) {
    if (tid < BM) rowmax[tid] = __uint_as_float(p.a_amax[min(m0 + tid, p.M - 1)]);
  } else {
    float mx = 0.f;
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
  }

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
  float sa[MI], ainv[MI];  // scale and inverse scale of this lane's A rows (fragment row fr = epilogue row fr)

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

  // epilogue operands, fetched before the DMAs as in k_gemm3w: the ReLU mask, the bias and the inverse scales of the B rows
  float4 mk[EPI == EPI_MASK ? MI : 1][NI];
  float4 bv[NI], bi[NI];
  if constexpr (EPI == EPI_MASK) gemm_prefetch_mask<MI, NI>(p, mk, m0 + wm0, n0 + wn0, lane);
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int nn = min(n0 + wn0 + j * 16 + fk * 4, p.N - 4);
    bi[j] = *reinterpret_cast<const float4*>(b_inv + nn);
    bv[j] = f4_zero();
    if constexpr (EPI == EPI_BIAS)
      if (p.bias) bv[j] = *reinterpret_cast<const float4*>(p.bias + nn);
  }
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
  sync(0);  // (its barrier also publishes rowmax)
#pragma unroll
  for (int i = 0; i < MI; ++i) pow2_scales(rowmax[wm0 + i * 16 + fr], sa[i], ainv[i]);
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
  // ---- epilogue (k_gemm3w's, behind the exact rescale): lane holds C[m0 + wm0 + 16 i + fr][n0 + wn0 + 16 j + 4 fk + 0..3]
  float* C = p.C;
  float cmax[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) cmax[i] = 0.f;
  const bool fold_stats = EPI == EPI_BIAS && p.bnf.n > 0;  // (uniform) the BatchNorm statistics of C are folded in this launch
  float* const elds = reinterpret_cast<float*>(smem2p);     // [BM / 16][2][BN]: the ring is free once every wave has left the k-loop
  if (fold_stats) __syncthreads();
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn0 + j * 16 + fk * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int mb = m0 + wm0 + i * 16, m = mb + fr;
      const float ai = ainv[i];
      float4 v = make_float4(acc[i][j][0] * (ai * bi[j].x), acc[i][j][1] * (ai * bi[j].y), acc[i][j][2] * (ai * bi[j].z),
                             acc[i][j][3] * (ai * bi[j].w));
      if constexpr (EPI == EPI_BIAS) {
        v = f4_add(v, bv[j]);
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (p.colstat || fold_stats) {  // (uniform) per-16-row-block column sums and squared deviations, as in k_gemm3w
          const int cnt = min(16, p.M - mb);
          if (cnt > 0) {
            const bool ok = fr < cnt;
            float4 sm = ok ? v : f4_zero();
            sm.x = row16_sum(sm.x); sm.y = row16_sum(sm.y); sm.z = row16_sum(sm.z); sm.w = row16_sum(sm.w);
            const float inv = 1.f / (float)cnt;
            float4 q;
            q.x = ok ? v.x - sm.x * inv : 0.f; q.y = ok ? v.y - sm.y * inv : 0.f;
            q.z = ok ? v.z - sm.z * inv : 0.f; q.w = ok ? v.w - sm.w * inv : 0.f;
            q.x = row16_sum(q.x * q.x); q.y = row16_sum(q.y * q.y); q.z = row16_sum(q.z * q.z); q.w = row16_sum(q.w * q.w);
            if (fr == 0 && n < p.N) {
              if (fold_stats) {  // into the tile's LDS image: merged below, inside this launch
                float* e = elds + ((wm0 / 16 + i) * 2) * BN + (wn0 + j * 16 + fk * 4);
                *reinterpret_cast<float4*>(e) = sm;
                *reinterpret_cast<float4*>(e + BN) = q;
              } else {
                float* cs = p.colstat + (int64_t)(mb >> 4) * 2 * p.N + n;
                *reinterpret_cast<float4*>(cs) = sm;
                *reinterpret_cast<float4*>(cs + p.N) = q;
              }
            }
          }
        }
      }
      if constexpr (EPI == EPI_MASK) {
        const float4 k4 = mk[i][j];
        if (!(k4.x > 0.f)) v.x = 0.f;
        if (!(k4.y > 0.f)) v.y = 0.f;
        if (!(k4.z > 0.f)) v.z = 0.f;
        if (!(k4.w > 0.f)) v.w = 0.f;
      }
      if (m < p.M && n < p.N) *reinterpret_cast<float4*>(C + (int64_t)m * p.ldc + n) = v;
      if (p.c_amax) {  // (uniform) the largest magnitude this lane wrote to row i's block
        const float lm = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (n < p.N) cmax[i] = fmaxf(cmax[i], lm);
      }
    }
  }
  // row maxima of C for the product that takes it as its A operand: the four lane groups (columns 4 fk ..) of a row fold by two
  // cross-lane maxima, the column tiles and the waves side by side by an atomic maximum on the bit patterns (non-negative floats
  // order like unsigned integers).  fmaxf drops a NaN operand: a row holding a NaN gets the maximum of its FINITE entries and is scaled
  // like any other row -- the NaN itself goes through the high plane (fp16(NaN s) = NaN) and poisons its row of the next product, as
  // it would in fp32; an infinite entry makes the maximum infinite and pow2_scales runs that row unscaled.
  if (p.c_amax) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v = cmax[i];
      v = fmaxf(v, __shfl_xor(v, 16));
      v = fmaxf(v, __shfl_xor(v, 32));
      const int m = m0 + wm0 + i * 16 + fr;
      if (fk == 0 && m < p.M) atomicMax(p.c_amax + m, __float_as_uint(v));
    }
  }
  if constexpr (EPI == EPI_BIAS) {
    if (fold_stats)
      bn_fwd_fold_tile<BM, BN>(p.bnf, elds, tile / tiles_n, (p.M + BM - 1) / BM, tile % tiles_n, m0, n0, p.M, p.N, tid, 64 * NW);
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI>
int launch_gemm2pw_s(const GemmArgs& p, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * (BM * 128 + 2 * BN * 64);
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  allow_big_lds((const void*)k_gemm2pw<BM, BN, WAVES_M, WAVES_N, STAGES, EPI>, lds);
  // a stop event handed over by the caller (set_next_launch_stop_event) becomes THIS dispatch's completion: a fork point for another
  // stream without a marker packet behind the kernel in this one (a hipEventRecord there idles the stream 7-8 us per layer of the
  // chem backward, profiles/r05/step_b256_timeline_bond.txt)
  if (hipEvent_t stop = take_next_launch_stop_event())
    hipExtLaunchKernelGGL((k_gemm2pw<BM, BN, WAVES_M, WAVES_N, STAGES, EPI>), dim3(tiles), dim3(64 * WAVES_M * WAVES_N), lds, st, nullptr, stop, 0, p);
  else
    hipLaunchKernelGGL((k_gemm2pw<BM, BN, WAVES_M, WAVES_N, STAGES, EPI>), dim3(tiles), dim3(64 * WAVES_M * WAVES_N), lds, st, p);
  return check_launch("gemm2pw");
}
// ==================================================================================================================================
// Large M: the weight planes RESIDENT in LDS, the activations streamed through registers, no barrier in the loop (k_gemm2pr).
// At 262 144 x 300 -> 600 the tiled kernel above spends its time waiting, not multiplying (profiles/r04/gemm2pr_ab.txt: the same
// 540-640 us with its MFMAs removed): every 64/128-row tile pays a workgroup launch, a pipeline fill and an epilogue, and every 32 of k a
// barrier, with one or two workgroups per CU to hide them behind.  Here one persistent workgroup per CU DMAs the two planes of ITS
// BN weight rows -- all NK k-steps of them, 150-152 KB in the ring's [row][32 k] image -- into LDS once; then its waves never
// synchronise again.  A wave owns 16-row blocks of A (block w, w + NW, ... of the workgroup's row range): each lane fetches its own
// MFMA fragments -- row fr, floats 32 s + 8 fk .. + 7 -- straight into registers, a WHOLE BLOCK ahead (fragment s of the next block
// goes into the registers fragment s of this block has just been split out of: 8 NK registers, static indices, the k-loop fully
// unrolled), splits them under the row's scale, multiplies against fragments read from the resident planes and writes its 16 x BN
// results.  The column workgroups that share a row range sit on one XCD and walk it at the same pace: A comes from HBM once.
// Loads are inline asm with counted s_waitcnt (hipcc's own pass would drain vmcnt(0) at every first use while LDS-DMA or younger
// loads are outstanding); the counts below hold with stores in flight (loads return in order among themselves: if a load is
// outstanding so is every younger load, so "at most <number of younger loads> outstanding" implies it has landed).
// Same fragments, same k order, same term order as k_gemm2pw: bit-identical (tests/test_gpu_ops.py).
// ==================================================================================================================================
// BN: the weight rows (output columns) a workgroup holds -- a multiple of 8: 120 of them at K = 300 fill the 160 KB (five workgroups
// cover N = 600, so A passes through five L1s, not eight: at ~20 bytes per clock and CU for a stream that misses the L1, THAT traffic
// is the kernel's bound, profiles/r04/gemm2pr_ab.txt); the last 16-column block of such a tile is half a block: eight weight rows in
// LDS, lanes 8-15 of the fragment read re-read rows 0-7, and the columns they produce are never stored.
template <int NK, int BN, int NW, int EPI>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) k_gemm2pr(GemmArgs p, int ranges_per_xcd) {
  constexpr int NI = (BN + 15) / 16, NFULL = BN / 16, B_PLANE = BN * 64, STAGE = 2 * B_PLANE;
  constexpr bool HALF = BN % 16 != 0;
  static_assert(BN % 8 == 0, "whole or half 16-column blocks");
  constexpr int NIM = EPI == EPI_MASK ? NI : 0;       // mask loads per block
  constexpr int Y_STEP = 2 * NK - 1 + NIM;            // loads younger than fragment s + 1 of this block when step s claims it
  constexpr int Y_BLOCK = 2 * NK - 2;                 // ... than fragment 0 / the row maximum at the top of a block, than the mask in its epilogue
  static_assert(NI >= 4 && NK >= 2 && Y_STEP < 64 && NW % 4 == 0, "shape");
  extern __shared__ __align__(16) unsigned char smem2r[];
  constexpr int EPN = 16 * NI;
  float* const epi = reinterpret_cast<float*>(smem2r + NK * STAGE);  // [2][EPN]: 1 / scale of the weight rows, bias
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int xcd = blockIdx.x % p.nxcd, slot = blockIdx.x / p.nxcd;  // (workgroups go round-robin over the XCDs)
  const int rloc = slot / tiles_n;
  if (rloc >= ranges_per_xcd) return;
  const int n0 = (slot % tiles_n) * BN;
  const int nranges = ranges_per_xcd * p.nxcd, range = xcd * ranges_per_xcd + rloc;
  const int nblocks = (p.M + 15) / 16;
  const int lo = (int)((int64_t)nblocks * range / nranges), hi = (int)((int64_t)nblocks * (range + 1) / nranges);
  const float* __restrict__ b_inv = reinterpret_cast<const float*>(p.Bp + 2 * p.bplane);

  // ---- the planes of rows n0 .. n0 + BN, every k-step: piece pb of plane q of stage s = rows 16 pb .. (a half piece: lanes 0-31)
  for (int d = wave; d < NK * 2 * NI; d += NW) {
    const int s = d / (2 * NI), e = d % (2 * NI), q = e / NI, pb = e % NI;
    const int row = 16 * pb + (lane >> 2);
    const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.Bp + q * p.bplane + (int64_t)min(n0 + row, p.N - 1) * p.ldbp);
    if (!HALF || pb < NFULL || lane < 32)
      __builtin_amdgcn_global_load_lds(PGNN_GPTR(src + min(64 * s + 16 * c, 2 * ((int)p.ldbp - 8))),
                                       PGNN_LPTR(smem2r + s * STAGE + q * B_PLANE + pb * 1024), 16, 0, 0);
  }
  for (int i = tid; i < EPN; i += 64 * NW) {
    const int nn = min(n0 + i, p.N - 1);
    epi[i] = b_inv[nn];
    epi[EPN + i] = (EPI == EPI_BIAS && p.bias) ? p.bias[nn] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int rb = lo + wave;
  if (rb >= hi) return;  // (no barrier below)

  const int b_off = fr * 64 + ((fk ^ ((-(fr >> 2)) & 3)) * 16);
  const int b_off_half = (fr & 7) * 64 + ((fk ^ ((-((fr & 7) >> 2)) & 3)) * 16);
  f16x8 b[NI][2];
  auto bload = [&](int stage, int j) {
    const unsigned char* s = smem2r + stage * STAGE + (HALF && j == NI - 1 ? b_off_half : b_off);
#pragma unroll
    for (int q = 0; q < 2; ++q) b[j][q] = *reinterpret_cast<const f16x8*>(s + q * B_PLANE + j * 16 * 64);
  };
  // this lane's fragments: raw[s] = floats 32 s + 8 fk .. + 7 of its row (fetching 64 contiguous bytes per row and load, with two
  // v_permlane swaps per dword back to this map, measured the same: tools/probe/permlane_probe.hip, profiles/r04/gemm2pr_ab.txt); the
  // last k-step re-reads the row's last four floats past the end (the planes are zero there), like the tiled kernel's DMA
  f32x4 raw[NK][2];
  float amx;
  f32x4 mk[EPI == EPI_MASK ? NI : 1];
  const int last0 = min(32 * (NK - 1) + 8 * fk, p.K - 4), last1 = min(32 * (NK - 1) + 8 * fk + 4, p.K - 4);
  auto row_of = [&](int blk) { return min(16 * blk + fr, p.M - 1); };
  auto issue_frag = [&](auto sc, const float* rowp) {
    constexpr int S = decltype(sc)::value;
    if constexpr (S < NK - 1) {
      const float* q = rowp + 8 * fk;
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(raw[S][0]) : "v"(q), "n"(S * 128) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(raw[S][1]) : "v"(q), "n"(S * 128 + 16) : "memory");
    } else {
      const float* q0 = rowp + last0;
      const float* q1 = rowp + last1;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[S][0]) : "v"(q0) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[S][1]) : "v"(q1) : "memory");
    }
  };
  auto issue_amax = [&](int blk) {  // (without maxima from the producer the slot is loaded all the same: the counts stay put)
    const uint32_t* q = p.a_amax ? p.a_amax + row_of(blk) : reinterpret_cast<const uint32_t*>(p.A);
    asm volatile("global_load_dword %0, %1, off" : "=v"(amx) : "v"(q) : "memory");
  };
#define PGNN_CLAIM(x) asm volatile("" : "+v"(x))  /* behind the wait that covers x's load: from here on x may be read */
  auto split_q = [&](int c, const f32x4& lo, const f32x4& hi, float sa, uint4 (&pl)[2]) {
    const f32x4 v = c < 2 ? lo : hi;
    uint32_t h, l;
    split2(v[2 * (c & 1)] * sa, v[2 * (c & 1) + 1] * sa, h, l);
    (&pl[0].x)[c] = h; (&pl[1].x)[c] = l;
  };

  {
    const float* rowp = p.A + (int64_t)row_of(rb) * p.lda;
    issue_amax(rb);
    auto all = [&](auto self, auto sc) {
      constexpr int S = decltype(sc)::value;
      if constexpr (S < NK) {
        issue_frag(sc, rowp);
        self(self, std::integral_constant<int, S + 1>{});
      }
    };
    all(all, std::integral_constant<int, 0>{});
  }
  constexpr int BD = NK == 10 ? 3 : 2;  // column blocks the plane fragments are read ahead (registers permitting)
#pragma unroll
  for (int j = 0; j < BD; ++j) bload(0, j);
  f32x4 acc[NI];
  f16x8 a0[2], a1[2];
  for (;;) {
    // the wave's last block fetches nothing ahead (a fetch nobody claims lands in registers hipcc believes free), and waits for
    // everything instead of counting: its fragments were issued a block ago
    const bool more = rb + NW < hi;
    const float* rowp = p.A + (int64_t)row_of(rb + NW) * p.lda;
    // ---- top of the block: its row maximum and fragment 0 have landed
    if (more) gemm_wait_vmcnt_imm<Y_BLOCK>();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(amx));
    PGNN_CLAIM(raw[0][0]);
    PGNN_CLAIM(raw[0][1]);
    if (!p.a_amax) {  // (uniform) no maxima from a producer: the wave holds its rows entirely -- fold them out of the fragments
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float mx = 0.f;
      auto fold = [&](auto self, auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < NK) {
          PGNN_CLAIM(raw[S][0]);
          PGNN_CLAIM(raw[S][1]);
#pragma unroll
          for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fabsf(raw[S][0][u]), fabsf(raw[S][1][u])));
          self(self, std::integral_constant<int, S + 1>{});
        }
      };
      fold(fold, std::integral_constant<int, 0>{});
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      amx = fmaxf(mx, __shfl_xor(mx, 32));
    }
    float sa, ainv;
    pow2_scales(amx, sa, ainv);
    {
      uint4 pl[2];
#pragma unroll
      for (int c = 0; c < 4; ++c) split_q(c, raw[0][0], raw[0][1], sa, pl);
      a0[0] = __builtin_bit_cast(f16x8, pl[0]);
      a0[1] = __builtin_bit_cast(f16x8, pl[1]);
    }
    if (more) {
      issue_amax(rb + NW);
      issue_frag(std::integral_constant<int, 0>{}, rowp);
    }
    const int mrow = row_of(rb);
    if constexpr (EPI == EPI_MASK) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float* q = p.mask + (int64_t)mrow * p.ldmask + min(n0 + j * 16 + fk * 4, p.N - 4);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(mk[j]) : "v"(q) : "memory");
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](auto sc, const f16x8 (&cur)[2], f16x8 (&nx)[2]) {
      constexpr int S = decltype(sc)::value;
      constexpr bool MORE = S + 1 < NK;          // a fragment of THIS block is still to be split
      constexpr int NS = MORE ? S + 1 : 0;
      uint4 pl[2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], cur[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][1], cur[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], cur[0], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) {
          if (j == 0) {
            gemm_wait_vmcnt_imm<Y_STEP>();
            PGNN_CLAIM(raw[NS][0]);
            PGNN_CLAIM(raw[NS][1]);
          }
          // the four quarters of the next fragment behind blocks 1 .. NI - 1
          constexpr int Q0 = 1;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if ((NI >= 5 ? Q0 + c : (c == 0 ? 1 : c)) == j) {
              split_q(c, raw[NS][0], raw[NS][1], sa, pl);
              asm volatile("" ::"v"((&pl[0].x)[c]), "v"((&pl[1].x)[c]));
            }
          if (j == (NI >= 5 ? 4 : 3) && more) issue_frag(std::integral_constant<int, NS>{}, rowp);  // the next block's, into the registers just split
        }
        if (j + BD < NI) bload(S, j + BD);
        else bload(MORE ? S + 1 : 0, j + BD - NI);  // (the last step wraps: the next block starts at stage 0)
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MORE) {
        nx[0] = __builtin_bit_cast(f16x8, pl[0]);
        nx[1] = __builtin_bit_cast(f16x8, pl[1]);
      }
    };
    auto ksteps = [&](auto self, auto sc) {
      constexpr int S = decltype(sc)::value;
      if constexpr (S < NK) {
        if constexpr (S % 2 == 0) kstep(sc, a0, a1);
        else kstep(sc, a1, a0);
        self(self, std::integral_constant<int, S + 1>{});
      }
    };
    ksteps(ksteps, std::integral_constant<int, 0>{});

    // ---- epilogue: lane holds C[16 rb + fr][n0 + 16 j + 4 fk + 0..3]
    if constexpr (EPI == EPI_MASK) {
      if (more) gemm_wait_vmcnt_imm<Y_BLOCK>();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(mk[j]));
    }
    const int mb = 16 * rb, m = mb + fr;
    float cmax = 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + j * 16 + fk * 4;
      const float4 bi = *reinterpret_cast<const float4*>(epi + j * 16 + fk * 4);
      float4 v = make_float4(acc[j][0] * (ainv * bi.x), acc[j][1] * (ainv * bi.y), acc[j][2] * (ainv * bi.z), acc[j][3] * (ainv * bi.w));
      if constexpr (EPI == EPI_BIAS) {
        v = f4_add(v, *reinterpret_cast<const float4*>(epi + EPN + j * 16 + fk * 4));
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (p.colstat) {  // (uniform) per-16-row-block column sums and squared deviations, as in k_gemm2pw
          const int cnt = min(16, p.M - mb);
          const bool ok = fr < cnt;
          float4 sm = ok ? v : f4_zero();
          sm.x = row16_sum(sm.x); sm.y = row16_sum(sm.y); sm.z = row16_sum(sm.z); sm.w = row16_sum(sm.w);
          const float inv = 1.f / (float)cnt;
          float4 q;
          q.x = ok ? v.x - sm.x * inv : 0.f; q.y = ok ? v.y - sm.y * inv : 0.f;
          q.z = ok ? v.z - sm.z * inv : 0.f; q.w = ok ? v.w - sm.w * inv : 0.f;
          q.x = row16_sum(q.x * q.x); q.y = row16_sum(q.y * q.y); q.z = row16_sum(q.z * q.z); q.w = row16_sum(q.w * q.w);
          if (fr == 0 && n < p.N && (!HALF || j < NI - 1 || fk < 2)) {
            float* cs = p.colstat + (int64_t)rb * 2 * p.N + n;
            *reinterpret_cast<float4*>(cs) = sm;
            *reinterpret_cast<float4*>(cs + p.N) = q;
          }
        }
      }
      if constexpr (EPI == EPI_MASK) {
        const f32x4 k4 = mk[j];
        if (!(k4[0] > 0.f)) v.x = 0.f;
        if (!(k4[1] > 0.f)) v.y = 0.f;
        if (!(k4[2] > 0.f)) v.z = 0.f;
        if (!(k4[3] > 0.f)) v.w = 0.f;
      }
      const bool mine = n < p.N && (!HALF || j < NI - 1 || fk < 2);  // (the second half of a half block belongs to the next workgroup)
      if (m < p.M && mine) *reinterpret_cast<float4*>(p.C + (int64_t)m * p.ldc + n) = v;
      if (mine) cmax = fmaxf(cmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (p.c_amax) {  // (uniform) as in k_gemm2pw: the row's maximum over this workgroup's columns, atomic over the column workgroups
      cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
      cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
      if (fk == 0 && m < p.M) atomicMax(p.c_amax + m, __float_as_uint(cmax));
    }
    if (!more) break;
    rb += NW;
  }
#undef PGNN_CLAIM
}

// can k_gemm2pr take this product?  K of 289-320 (10 k-steps, 120 columns per workgroup) or 577-608 (19, 64): the GIN mlp's two shapes;
// enough 16-row blocks for every wave of every workgroup to stream a few; one workgroup per CU, a whole number of row ranges per XCD
inline int gemm2pr_ranges(const GemmArgs& p, int bn) {
  const int per_xcd = num_cu() / std::max(1, p.nxcd);
  return per_xcd / (int)ceil_div(p.N, bn);
}
inline bool gemm2pr_eligible(const GemmArgs& p) {
  const int knob = env_knob("PGNN_GEMM2P_RES", 1);
  if (!knob || p.bnf.n > 0 || p.nxcd <= 0 || num_cu() % p.nxcd != 0) return false;
  const int64_t nk = ceil_div(p.K, 32);
  const int bn = nk == 10 ? 120 : nk == 19 ? 64 : 0;
  if (!bn || gemm2pr_ranges(p, bn) < 1) return false;
  // where it overtakes the tiled kernel (tools/gemm2p_sweep.py, profiles/r04/gemm2pr_sweep.txt): 300 -> 600 from ~8 k rows, 600 -> 600
  // (ten column workgroups) from ~9 k, 600 -> 300 from ~20 k
  const int64_t from = nk == 10 ? 8192 : p.N > 320 ? 9216 : 20480;
  return knob >= 2 || p.M >= from;
}
template <int NK, int BN, int NW, int EPI>
int launch_gemm2pr_s(const GemmArgs& p, hipStream_t st) {
  constexpr size_t lds = (size_t)NK * 2 * BN * 64 + 2 * (16 * ((BN + 15) / 16)) * 4;
  static_assert(lds <= 160 * 1024, "LDS");
  const int ranges = gemm2pr_ranges(p, BN);
  allow_big_lds((const void*)k_gemm2pr<NK, BN, NW, EPI>, lds);
  // (a stop event handed over by the caller becomes this dispatch's completion, as in launch_gemm2pw_s)
  if (hipEvent_t stop = take_next_launch_stop_event())
    hipExtLaunchKernelGGL((k_gemm2pr<NK, BN, NW, EPI>), dim3(num_cu()), dim3(64 * NW), lds, st, nullptr, stop, 0, p, ranges);
  else
    hipLaunchKernelGGL((k_gemm2pr<NK, BN, NW, EPI>), dim3(num_cu()), dim3(64 * NW), lds, st, p, ranges);
  return check_launch("gemm2pr");
}
template <int EPI>
int launch_gemm2pr(const GemmArgs& p, hipStream_t st) {
  if (ceil_div(p.K, 32) == 10) return launch_gemm2pr_s<10, 120, 8, EPI>(p, st);
  return launch_gemm2pr_s<19, 64, 8, EPI>(p, st);
}

// the two tiles of launch_gemm3w's default choice, under its rounds x (k-steps + fixed) model with the two-plane step times
template <int EPI>
int launch_gemm2pw(const GemmArgs& p, hipStream_t st) {
  if (gemm2pr_eligible(p)) return launch_gemm2pr<EPI>(p, st);
  const int64_t nk = ceil_div(p.K, 32), cus = num_cu();
  const double t64 = (double)ceil_div(ceil_div(p.M, 64) * ceil_div(p.N, 160), cus) * (0.5 * nk + 5.0);
  const double t128 = (double)ceil_div(ceil_div(p.M, 128) * ceil_div(p.N, 160), cus) * (0.95 * nk + 6.0);
  // 112 rows (seven waves of 16 x 160): where 128-row tiles leave a sixth of the CUs without one -- 6 7xx x 600: 212 tiles of 128 rows
  // on 256 CUs, 244 of 112 -- the same single round with an eighth less work per tile (PGNN_GEMM2P_T112=0: off)
  const double t112 = env_knob("PGNN_GEMM2P_T112", 1) ? (double)ceil_div(ceil_div(p.M, 112) * ceil_div(p.N, 160), cus) * (0.84 * nk + 5.8) : 1e30;
  if (t112 < t128 && t112 < t64) return launch_gemm2pw_s<112, 160, 7, 1, 3, EPI>(p, st);
  if (t128 <= t64) return launch_gemm2pw_s<128, 160, 8, 1, 3, EPI>(p, st);
  return launch_gemm2pw_s<64, 160, 4, 2, 4, EPI>(p, st);
}

}  // namespace

int split_weights_2p(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                     int64_t count, int64_t* const* bump, int nbump, hipStream_t stream, const EncTables* tabs, uint32_t* zero_ptr,
                     int64_t zero_words) {
  PGNN_REQUIRE(count >= 0 && count <= 32 && nbump >= 0 && nbump <= 16, "split_weights: at most 32 matrices (and 16 counters) per call");
  PGNN_REQUIRE(!tabs || (tabs->count >= 0 && tabs->count <= 16 && tabs->dim > 0 && tabs->k > 0), "split_weights: at most 16 encoder tables");
  PGNN_REQUIRE(zero_words == 0 || zero_ptr, "split_weights: zero_ptr");
  const bool with_tabs = tabs && tabs->count > 0;
  if (count == 0 && !with_tabs && zero_words == 0) return PGNN_OK;
  SplitJobs jobs{};
  if (with_tabs) jobs.tabs = *tabs;
  for (int j = 0; j < nbump; ++j) jobs.bump[j] = reinterpret_cast<long long*>(bump[j]);
  jobs.nbump = nbump;
  int64_t most = 1;
  for (int j = 0; j < count; ++j) {
    PGNN_REQUIRE(src[j] && dst[j] && rows[j] > 0 && cols[j] > 0 && rows[j] < (1 << 24) && cols[j] < (1 << 24), "split_weights: bad job");
    const bool tr = transpose && transpose[j];
    jobs.src[j] = src[j]; jobs.dst[j] = static_cast<unsigned short*>(dst[j]);
    jobs.rows[j] = (int)(tr ? cols[j] : rows[j]); jobs.cols[j] = (int)(tr ? rows[j] : cols[j]);
    jobs.ld[j] = (int)(ceil_div(jobs.cols[j], 32) * 32); jobs.transpose[j] = tr;
    most = std::max(most, ceil_div(jobs.rows[j], 4));
  }
  if (with_tabs) most = std::max<int64_t>(most, std::min<int64_t>(ceil_div((int64_t)tabs->count * (tabs->k + 1) * tabs->dim, 256), 64));
  hipLaunchKernelGGL(k_split2p_jobs, dim3((int)std::min<int64_t>(most, 4096), (int)std::max<int64_t>(count, 1)), dim3(256), 0, stream, jobs,
                     zero_ptr, zero_words);
  return check_launch("split_weights_2p");
}

int linear_fwd_wp_2p(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m, int64_t k,
                     int64_t n, int relu, float* colstat, hipStream_t st, const uint32_t* x_amax, uint32_t* y_amax, const BnFwdFold* bnf) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && wplanes,
               "linear_fwd_2p: K, N and the leading dimensions must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  p.A = x; p.lda = ldx; p.C = y; p.ldc = ldy;
  p.Bp = static_cast<const unsigned short*>(wplanes); p.ldbp = ceil_div(k, 32) * 32; p.bplane = n * p.ldbp;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.bias = bias; p.relu = relu; p.kchunk = (int)k; p.split_stride = 0;
  p.colstat = colstat; p.a_amax = x_amax; p.c_amax = y_amax;
  if (bnf) {
    PGNN_REQUIRE(bnf->n == m && bnf->part && bnf->gpart && bnf->tickets && ceil_div(n, 160) <= kFwdFoldPanels, "linear_fwd_2p: bad statistics fold");
    p.bnf = *bnf;
  }
  return launch_gemm2pw<EPI_BIAS>(p, st);
}

int linear_bwd_data_wp_2p(const float* dy, int64_t lddy, const void* wtplanes, const float* relu_out, int64_t ldr, float* dx, int64_t lddx,
                          int64_t m, int64_t k, int64_t n, hipStream_t st, const uint32_t* dy_amax, uint32_t* dx_amax) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && wtplanes,
               "linear_bwd_data_2p: K, N and the leading dimensions must be multiples of 4");
  GemmArgs p{};
  p.nxcd = num_xcd();
  p.A = dy; p.lda = lddy; p.C = dx; p.ldc = lddx;
  p.Bp = static_cast<const unsigned short*>(wtplanes); p.ldbp = ceil_div(n, 32) * 32; p.bplane = k * p.ldbp;
  p.M = (int)m; p.N = (int)k; p.K = (int)n; p.mask = relu_out; p.ldmask = ldr; p.kchunk = (int)n; p.split_stride = 0;
  p.a_amax = dy_amax; p.c_amax = dx_amax;
  return relu_out ? launch_gemm2pw<EPI_MASK>(p, st) : launch_gemm2pw<EPI_PLAIN>(p, st);
}

}  // namespace pgnn

extern "C" {

int pgnn_split_weights_2p(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                          int64_t count, pgnn_stream stream) {
  return pgnn::split_weights_2p(src, dst, rows, cols, transpose, count, nullptr, 0, (hipStream_t)stream, nullptr, nullptr, 0);
}
int pgnn_linear_fwd_2p(const float* x, int64_t ldx, const uint32_t* x_amax, const void* wplanes2, const float* bias, float* y, int64_t ldy,
                       int64_t m, int64_t k, int64_t n, int relu, float* colstat, uint32_t* y_amax, pgnn_stream stream) {
  return pgnn::linear_fwd_wp_2p(x, ldx, wplanes2, bias, y, ldy, m, k, n, relu, colstat, (hipStream_t)stream, x_amax, y_amax, nullptr);
}
int pgnn_linear_bwd_data_2p(const float* dy, int64_t lddy, const uint32_t* dy_amax, const void* wtplanes2, const float* relu_out, int64_t ldr,
                            float* dx, int64_t lddx, int64_t m, int64_t k, int64_t n, uint32_t* dx_amax, pgnn_stream stream) {
  return pgnn::linear_bwd_data_wp_2p(dy, lddy, wtplanes2, relu_out, ldr, dx, lddx, m, k, n, (hipStream_t)stream, dy_amax, dx_amax);
}

}  // extern "C"
