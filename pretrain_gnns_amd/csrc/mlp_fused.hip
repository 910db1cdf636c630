// The GIN mlp as ONE launch per direction at large row counts (gfx950): chem/model.py:29,54-55
//     forward        hid = relu(agg . W1^T + b1)        z    = hid . W2^T + b2
//     backward-data  dhid = (dz . W2) * (hid > 0)       dagg = dhid . W1
// Both are "X [M, 300] -> H [M, 600] -> Y [M, 300]" with an elementwise step on H in between.  The two products on planes
// (csrc/linear.hip, k_gemm2pr) each stream their activation operand through the L1s of several column workgroups and write / re-read
// H through HBM: profiles/r04/gemm2pr_ab.txt shows the large-M product costs the same with its MFMAs deleted.  Here a WAVE owns 16
// rows for both products: X is fetched once (into registers, as two fp16 planes under the row's power-of-two scale), H is produced
// 32 columns at a time in the first product's accumulators, written ONCE (the backward and the weight gradients need it), split
// into planes in registers and consumed at once as one 32-deep k-step of the second product, whose 16 x 304 accumulators stay in
// registers for the whole row block.  No shuffle in between: the ROWS of the first matrix are dealt to the accumulator positions
// (by the DMA that stages them) so that a lane's eight H values -- positions 4 fk + 0..3 of the chunk's two 16-column blocks --
// are the chunk's columns 8 fk .. 8 fk + 7, which IS the A fragment of v_mfma_f32_16x16x32_f16 for that 32-deep k-step.
//
// What streams instead of the activations is the weights: per 32 columns of H, the [32 rows][320 k] slice of the first matrix's
// planes (40 KB) and the [304 rows][32 k] slice of the second's (38 KB), by LDS-DMA from L2 into a three-slot ring shared by the
// workgroup's eight waves (8 x 16 = 128 rows per pass; one barrier per slice, the DMA two slices ahead under a counted vmcnt).
// That is 1.5 MB from L2 per 128 rows instead of 0.9 MB per 128 rows from HBM; the workgroups walk the same slices at the same
// pace, so they are L2 hits.
//
// The scale of H's planes: a row's maximum over all 600 columns is not known when its first 32 are split, so the scale RUNS: a
// chunk whose maximum would leave fp16's range under the current scale lowers it (to the chunk's own [2^13, 2^14) placement) and
// the second product's accumulators of that row are multiplied by the ratio -- a power of two, exact.  Every chunk is therefore
// split at a scale at least as fine as the whole-row scale of the unfused kernels; H has their bits, Y differs from theirs only
// in elements below 2^-24 of a row's maximum (tests/test_gpu_ops.py: float64 bar, and no worse than the two products).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "two_plane.h"

namespace pgnn {
namespace {

struct MlpFusedArgs {
  const float* X;             // [M, K1]
  int64_t ldx;
  const unsigned short* P1;   // planes of the first matrix [2][N1][ld1] fp16 + 1 / scale per row (pgnn_split_weights_2p)
  int64_t ld1, plane1;
  const unsigned short* P2;   // planes of the second [2][N2][ld2]
  int64_t ld2, plane2;
  const float* bias1;         // forward: [N1], [N2] (NULL: none)
  const float* bias2;
  const float* mask;          // backward: the forward's H; the new H is zeroed where mask <= 0
  int64_t ldmask;
  float* H;                   // [M, N1]
  int64_t ldh;
  float* Y;                   // [M, N2]
  int64_t ldy;
  float* colstat;             // forward, optional: [ceil(M/16)][2][N2] column sums / squared deviations of Y per 16-row block
  int M, K1, N1, N2;
  int groups;                 // ceil(M / 128): passes of the workgroups
};

constexpr int kSlot = 40 * 1024;  // one weight slice
constexpr int kRing = 3;
constexpr int kMaxChunks = 19;        // N1 <= 608


// NK1: 32-deep k-steps of the first product (K1 in (32 (NK1 - 1), 32 NK1]); NB2: 16-column blocks of Y.
// Eight waves per workgroup, two per SIMD, each owning ONE block of 16 rows: the planes of X (80 registers) and the accumulators
// of Y (76) leave room for double-buffered weight fragments inside 256 registers.  (Four waves of two blocks each -- half the LDS
// reads per MFMA -- want ~470 of the 512 registers a lone wave per SIMD may have, and hipcc spills 94-144 of them.)
constexpr int kFusedWaves = 8, kFusedMI = 1;
constexpr int kFusedRows = kFusedWaves * kFusedMI * 16;  // rows per pass of a workgroup
// ABL (builds with -DPGNN_AB only; results are WRONG, the time is what is asked): bit 0 no MFMAs, bit 1 no stores of H / Y, bit 2 no
// weight refill (the DMA stream), bit 3 no fragment reads from LDS, bit 4 no split of the chunk into planes (profiles/r05/mlp_fused_ablation.txt)
template <int NK1, int NB2, bool BWD, int ABL = 0>
__global__ void __launch_bounds__(64 * kFusedWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mlp2p_fused(MlpFusedArgs p) {
  constexpr int NW = kFusedWaves, MI = kFusedMI;
  constexpr int NP1 = NK1 * 4, NP2 = NB2 * 2;  // 1-KiB DMA pieces of a slice
  constexpr int NJ = (NP1 + NW - 1) / NW;      // ... per wave
  static_assert(NP1 * 1024 <= kSlot && NP2 * 1024 <= kSlot && (NP2 + NW - 1) / NW == NJ, "slice shape");
  extern __shared__ __align__(16) unsigned char smemf[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int NC = (p.N1 + 31) >> 5;  // 32-column chunks of H
  // epilogue constants in a STATIC array: hipcc orders every read of the dynamic array behind all LDS-DMA in flight (vmcnt(0)), and
  // a read of these in the middle of a slice would drain the prefetch of the slice after next
  __shared__ __align__(16) float epi_c[2 * kMaxChunks * 32 + 2 * NB2 * 16];
  float* const inv1 = epi_c;                   // [NC 32]: 1 / scale of the first matrix's rows
  float* const bia1 = inv1 + kMaxChunks * 32;
  float* const inv2 = bia1 + kMaxChunks * 32;  // [NB2 16]
  float* const bia2 = inv2 + NB2 * 16;
  {
    const float* g1 = reinterpret_cast<const float*>(p.P1 + 2 * p.plane1);
    const float* g2 = reinterpret_cast<const float*>(p.P2 + 2 * p.plane2);
    for (int i = tid; i < NC * 32; i += 64 * NW) {
      const int n = min(i, p.N1 - 1);
      inv1[i] = g1[n];
      bia1[i] = (!BWD && p.bias1) ? p.bias1[n] : 0.f;
    }
    for (int i = tid; i < NB2 * 16; i += 64 * NW) {
      const int n = min(i, p.N2 - 1);
      inv2[i] = g2[n];
      bia2[i] = (!BWD && p.bias2) ? p.bias2[n] : 0.f;
    }
  }
  const int npass = (p.groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = 2 * NC * npass;  // slices this workgroup consumes: first-matrix slice of chunk 0, second-matrix slice of chunk 0, ...

  // ---- the weight stream.  Piece d of a first-matrix slice: k-step d >> 2, plane (d >> 1) & 1, rows 16 (d & 1) ..; of a second-matrix
  // slice: plane d / NB2, rows 16 (d % NB2) ...  A piece is 16 rows x 64 bytes, the four 16-byte chunks of a row XOR-swizzled by the
  // row (k_gemm2pw's image: fragment reads are conflict-free).  Rows beyond the matrix re-read its last row (never stored).
  // (addresses as a wave-uniform 64-bit base + a 32-bit lane offset: ten hoisted 64-bit lane addresses cost the registers that
  // make the difference between no spill and a scratch reload -- with its vmcnt(0) -- in the middle of the stream)
  const int lrow = lane >> 2, c4 = (lane & 3) ^ ((-(lane >> 4)) & 3);
  const unsigned char* const g1 = reinterpret_cast<const unsigned char*>(p.P1);
  const unsigned char* const g2 = reinterpret_cast<const unsigned char*>(p.P2);
  const uint32_t pitch1 = (uint32_t)p.ld1 * 2u, pitch2 = (uint32_t)p.ld2 * 2u;  // bytes per row of a plane
  // piece i (of this wave's NJ) of the first- / second-matrix slice of chunk `chunk`, into ring slot `slot_idx`
  auto piece_w1 = [&](int i, int chunk, int slot_idx) {
    const int d = min(wave + NW * i, NP1 - 1);
    const int jj = d & 1, q = (d >> 1) & 1, s = d >> 2;
    // position i of block jj holds the row of H's column 8 (i >> 2) + 4 jj + (i & 3): the accumulators of lane group fk (positions
    // 4 fk + 0..3 of both blocks) are then the chunk's columns 8 fk .. 8 fk + 7 -- a second-product A fragment in the standard k order
    const uint32_t row = (uint32_t)min(32 * chunk + 8 * (lrow >> 2) + 4 * jj + (lrow & 3), p.N1 - 1);
    const unsigned char* ub = g1 + ((int64_t)q * p.plane1 * 2 + 64 * s);
    __builtin_amdgcn_global_load_lds(PGNN_GPTR(ub + (row * pitch1 + 16u * (uint32_t)c4)), PGNN_LPTR(smemf + slot_idx * kSlot + d * 1024), 16, 0, 0);
  };
  auto piece_w2 = [&](int i, int chunk, int slot_idx) {
    const int d = min(wave + NW * i, NP2 - 1);
    const int q = d >= NB2 ? 1 : 0, j = d - q * NB2;
    const uint32_t row = (uint32_t)min(16 * j + lrow, p.N2 - 1);
    const unsigned char* ub = g2 + ((int64_t)q * p.plane2 * 2 + 64 * chunk);
    __builtin_amdgcn_global_load_lds(PGNN_GPTR(ub + (row * pitch2 + 16u * (uint32_t)c4)), PGNN_LPTR(smemf + slot_idx * kSlot + d * 1024), 16, 0, 0);
  };
  // top of a slice: this wave's pieces of it have landed (the NJ pieces of the next slice may still be in flight: loads return in
  // order among themselves, stores only make the count conservative -- `drain` where no younger slice was issued: the last one, and
  // a pass's first, whose rows were fetched behind it), then every wave's have, and nobody reads the slot this slice's MFMA loop
  // refills any more.  The refill -- the slice after next, NJ pieces per wave -- is issued a piece at a time BETWEEN the units of
  // the loop: a piece costs its wave 60-180 cycles of issue, and at a slice's top both waves of a SIMD would pay for all of theirs
  // at the same moment with the matrix pipe idle (measured: 45 % MFMA-busy with the issue at the top).
  auto slice_top = [&](bool drain) {
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else gemm_wait_vmcnt_imm<NJ>();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
  };

  // fragment addresses inside a slice
  const int swz = (-(fr >> 2)) & 3;
  const int w1_off = fr * 64 + ((fk ^ swz) * 16);
  // one unit of either product: the two planes of a 16-row block of a slice -- unit u of a first-matrix slice is k-step u >> 1, block
  // u & 1; unit j of a second-matrix slice is block j -- and the three MFMAs it feeds (term order of k_gemm2pw: high.low, low.high,
  // high.high).  Fragments are fetched kAhead units ahead of their MFMAs into a rotating set of kAhead + 1 register pairs, and the
  // wait in front of a unit's MFMAs is COUNTED: the 2 kAhead younger fetches stay in flight, no LDS latency on a wave's critical
  // path.  Inline asm because hipcc, with LDS-DMA pending, drains lgkmcnt to 0 at every wait it places (it treats the DMA like a
  // FLAT access that may return out of order) -- one full LDS round trip per kAhead + 1 units: measured 43 % MFMA-busy.  The
  // fragment reads are the only LDS traffic of a wave inside the product loops, so the counts are exact.
  constexpr int kAhead = 2;
  const uint32_t lds_base = (uint32_t)(uintptr_t)PGNN_LPTR(smemf) + (uint32_t)w1_off;
#define PGNN_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
  auto ld_w1 = [&](uint32_t addr, auto uc, f16x8 (&w)[2]) {
    constexpr int u = decltype(uc)::value;
    if constexpr ((ABL & 8) != 0) return;
    PGNN_DS_READ(w[0], addr, (((u >> 1) * 2 + 0) * 2 + (u & 1)) * 1024);
    PGNN_DS_READ(w[1], addr, (((u >> 1) * 2 + 1) * 2 + (u & 1)) * 1024);
  };
  auto ld_w2 = [&](uint32_t addr, auto jc, f16x8 (&w)[2]) {
    constexpr int j = decltype(jc)::value;
    if constexpr ((ABL & 8) != 0) return;
    PGNN_DS_READ(w[0], addr, j * 1024);
    PGNN_DS_READ(w[1], addr, (NB2 + j) * 1024);
  };
  // behind the wait that covers them: from here on the pair may be read
  auto claim = [&](f16x8 (&w)[2]) { asm volatile("" : "+v"(w[0]), "+v"(w[1])); };
  auto mfma3 = [&](f32x4 (&c)[MI], const f16x8 (&w)[2], const f16x8 (&x)[MI][2]) {
    if constexpr ((ABL & 1) != 0) {  // (the operands stay live: one add each)
#pragma unroll
      for (int i = 0; i < MI; ++i) c[i][0] += (float)w[0][0] + (float)w[1][0] + (float)x[i][0][0] + (float)x[i][1][0];
      return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[i][1], c[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], x[i][0], c[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[i][0], c[i], 0, 0, 0);
  };

  // this lane's share of a 16-row block of X: floats 32 s + 8 fk .. + 7 of row fr; the last k-step re-reads the row's last floats
  // where it would run past the end (the planes are zero there)
  const int last0 = min(32 * (NK1 - 1) + 8 * fk, p.K1 - 4), last1 = min(32 * (NK1 - 1) + 8 * fk + 4, p.K1 - 4);
  f32x4 raw[MI][NK1][2];
  auto load_rows = [&](int g) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float* rowp = p.X + (int64_t)min(16 * ((g * NW + wave) * MI + i) + fr, p.M - 1) * p.ldx;
#pragma unroll
      for (int s = 0; s < NK1 - 1; ++s) {
        raw[i][s][0] = *reinterpret_cast<const f32x4*>(rowp + 32 * s + 8 * fk);
        raw[i][s][1] = *reinterpret_cast<const f32x4*>(rowp + 32 * s + 8 * fk + 4);
      }
      raw[i][NK1 - 1][0] = *reinterpret_cast<const f32x4*>(rowp + last0);
      raw[i][NK1 - 1][1] = *reinterpret_cast<const f32x4*>(rowp + last1);
    }
  };

#pragma unroll
  for (int i = 0; i < NJ; ++i) piece_w1(i, 0, 0);
#pragma unroll
  for (int i = 0; i < NJ; ++i) piece_w2(i, 0, 1);
  load_rows(blockIdx.x);
  int rslot = 0, t = 0;  // the slot and the index of the slice being consumed
  auto next_rslot = [&]() {
    rslot = rslot + 1 == kRing ? 0 : rslot + 1;
    ++t;
  };

  for (int g = blockIdx.x; g < p.groups; g += gridDim.x) {
    const int rb0 = (g * NW + wave) * MI;  // this wave's first 16-row block
    // ---- the blocks' rows as two planes under each row's scale
    f16x8 a[MI][NK1][2];
    float ainv[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float mx = 0.f;
#pragma unroll
      for (int s = 0; s < NK1; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fabsf(raw[i][s][0][u]), fabsf(raw[i][s][1][u])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sa;
      pow2_scales(mx, sa, ainv[i]);
#pragma unroll
      for (int s = 0; s < NK1; ++s) {
        uint4 ph, pl;
        split2(raw[i][s][0][0] * sa, raw[i][s][0][1] * sa, ph.x, pl.x);
        split2(raw[i][s][0][2] * sa, raw[i][s][0][3] * sa, ph.y, pl.y);
        split2(raw[i][s][1][0] * sa, raw[i][s][1][1] * sa, ph.z, pl.z);
        split2(raw[i][s][1][2] * sa, raw[i][s][1][3] * sa, ph.w, pl.w);
        a[i][s][0] = __builtin_bit_cast(f16x8, ph);
        a[i][s][1] = __builtin_bit_cast(f16x8, pl);
      }
    }
    f32x4 o[NB2][MI];
#pragma unroll
    for (int j = 0; j < NB2; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) o[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_run[MI], inv_run[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      s_run[i] = __uint_as_float(0x7F000000u);  // 2^127: no chunk has set a scale yet
      inv_run[i] = __uint_as_float(0x00400000u);
    }

    for (int c = 0; c < NC; ++c) {
      // ================= first product: H[:, 32 c .. 32 c + 31] of this wave's rows
      slice_top(c == 0);
      const int nbase = 32 * c + 8 * fk;  // this lane's eight columns of the chunk
      f32x4 mk[MI][2];
      if constexpr (BWD) {  // (in front of the DMA: the wait at the next slice's top then covers them)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            mk[i][jj] = *reinterpret_cast<const f32x4*>(p.mask + (int64_t)min(16 * (rb0 + i) + fr, p.M - 1) * p.ldmask +
                                                        min(nbase + 4 * jj, p.N1 - 4));
      }
      const int cnext = c + 1 == NC ? 0 : c + 1, wslot = rslot == 0 ? kRing - 1 : rslot - 1;  // the slice after next: chunk, ring slot
      f32x4 h[2][MI];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < MI; ++i) h[jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        const uint32_t addr = lds_base + (uint32_t)(rslot * kSlot);
        f16x8 w[kAhead + 1][2];
        auto units = [&](auto self, auto uc) {
          constexpr int u = decltype(uc)::value;
          if constexpr (u < 2 * NK1 + kAhead) {
            if constexpr (u < 2 * NK1) ld_w1(addr, uc, w[u % (kAhead + 1)]);
            if constexpr (u >= kAhead) {  // the MFMAs of unit v = u - kAhead: its pair has landed once at most 2 * (units issued behind it) reads are out
              constexpr int v = u - kAhead;
              constexpr int behind = (u < 2 * NK1 ? u : 2 * NK1 - 1) - v;
              asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * behind) : "memory");
              claim(w[v % (kAhead + 1)]);
              f16x8 x[MI][2];
#pragma unroll
              for (int i = 0; i < MI; ++i) { x[i][0] = a[i][v >> 1][0]; x[i][1] = a[i][v >> 1][1]; }
              mfma3(h[v & 1], w[v % (kAhead + 1)], x);
              if constexpr (v % 3 == 0 && v >= 3 && v / 3 <= NJ) {
                if constexpr ((ABL & 4) == 0) piece_w1(v / 3 - 1, cnext, wslot);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            self(self, std::integral_constant<int, u + 1>{});
          }
        };
        __builtin_amdgcn_sched_barrier(0);
        units(units, std::integral_constant<int, 0>{});
      }
      next_rslot();
      // ================= the chunk between the products, then its k-step of the second
      // (its constants: asm reads in front of the top's lgkmcnt(0) -- a read hipcc sees would come with a vmcnt(0) that drains the refill)
      f32x4 bi1[2], bb1[2];
      {
        const uint32_t ea = (uint32_t)(uintptr_t)PGNN_LPTR(inv1 + nbase);
        PGNN_DS_READ(bi1[0], ea, 0);
        PGNN_DS_READ(bi1[1], ea, 16);
        if constexpr (!BWD) {
          PGNN_DS_READ(bb1[0], ea, kMaxChunks * 32 * 4);
          PGNN_DS_READ(bb1[1], ea, kMaxChunks * 32 * 4 + 16);
        }
      }
      slice_top(t == total - 1);
      asm volatile("" : "+v"(bi1[0]), "+v"(bi1[1]));
      if constexpr (!BWD) asm volatile("" : "+v"(bb1[0]), "+v"(bb1[1]));
      f16x8 hp[MI][2];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = 16 * (rb0 + i) + fr;
        float vv[8];
        float cm = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const float4 bi = make_float4(bi1[jj][0], bi1[jj][1], bi1[jj][2], bi1[jj][3]);
          const float ai = ainv[i];
          float4 v = make_float4(h[jj][i][0] * (ai * bi.x), h[jj][i][1] * (ai * bi.y), h[jj][i][2] * (ai * bi.z), h[jj][i][3] * (ai * bi.w));
          if constexpr (!BWD) {
            v = f4_add(v, make_float4(bb1[jj][0], bb1[jj][1], bb1[jj][2], bb1[jj][3]));
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          } else {
            if (!(mk[i][jj][0] > 0.f)) v.x = 0.f;
            if (!(mk[i][jj][1] > 0.f)) v.y = 0.f;
            if (!(mk[i][jj][2] > 0.f)) v.z = 0.f;
            if (!(mk[i][jj][3] > 0.f)) v.w = 0.f;
          }
          vv[4 * jj] = v.x; vv[4 * jj + 1] = v.y; vv[4 * jj + 2] = v.z; vv[4 * jj + 3] = v.w;
          cm = fmaxf(cm, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        {
          // H goes out as whole 64-byte runs per row and instruction: lane group fk holds columns 8 fk .. + 7 of the chunk, so a store
          // of its first float4 would leave 16-byte holes (32 bytes apart) that the second store fills later -- 64 partial writes per
          // instruction at the L2.  Two cross-row swaps per register (v_permlane16_swap: rows 1 <-> 0' and 3 <-> 2'; v_permlane32_swap:
          // rows 2,3 <-> 0',1') hand group fk the chunk's columns 4 fk .. + 3 (first store) and 16 + 4 fk .. + 3 (second).
          typedef unsigned u2_t __attribute__((ext_vector_type(2)));
          float4 s0, s1;
          float* a0 = &s0.x;
          float* a1 = &s1.x;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            u2_t q = __builtin_amdgcn_permlane16_swap(__float_as_uint(vv[r]), __float_as_uint(vv[4 + r]), false, false);
            q = __builtin_amdgcn_permlane32_swap(q[0], q[1], false, false);
            a0[r] = __uint_as_float(q[0]);
            a1[r] = __uint_as_float(q[1]);
          }
          const int ns = 32 * c + 4 * fk;
          if (((ABL & 2) == 0 || p.M < 0) && m < p.M) {
            float* hrow = p.H + (int64_t)m * p.ldh + ns;
            if (ns < p.N1) *reinterpret_cast<float4*>(hrow) = s0;
            if (ns + 16 < p.N1) *reinterpret_cast<float4*>(hrow + 16) = s1;
          }
        }
        cm = fmaxf(cm, __shfl_xor(cm, 16));
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        // the running scale of this row's planes of H: lowered when the chunk would not fit under it
        const bool lower = cm * s_run[i] >= 32768.f;
        if (__builtin_amdgcn_ballot_w64(lower) != 0) {
          float ratio = 1.f;
          if (lower) {
            float sn, in;
            pow2_scales(cm, sn, in);
            ratio = sn * inv_run[i];
            s_run[i] = sn;
            inv_run[i] = in;
          }
#pragma unroll
          for (int j = 0; j < NB2; ++j) o[j][i] *= ratio;
        }
        const float sr = s_run[i];
        uint4 ph, pl;
        if constexpr ((ABL & 16) != 0) {
          ph = make_uint4(__float_as_uint(vv[0] * sr), __float_as_uint(vv[2]), __float_as_uint(vv[4]), __float_as_uint(vv[6]));
          pl = make_uint4(__float_as_uint(vv[1]), __float_as_uint(vv[3]), __float_as_uint(vv[5]), __float_as_uint(vv[7]));
        } else {
        split2(vv[0] * sr, vv[1] * sr, ph.x, pl.x);
        split2(vv[2] * sr, vv[3] * sr, ph.y, pl.y);
        split2(vv[4] * sr, vv[5] * sr, ph.z, pl.z);
        split2(vv[6] * sr, vv[7] * sr, ph.w, pl.w);
        }
        hp[i][0] = __builtin_bit_cast(f16x8, ph);
        hp[i][1] = __builtin_bit_cast(f16x8, pl);
      }
      {
        const int wslot2 = rslot == 0 ? kRing - 1 : rslot - 1;
        const uint32_t addr = lds_base + (uint32_t)(rslot * kSlot);
        f16x8 w[kAhead + 1][2];
        auto units = [&](auto self, auto uc) {
          constexpr int u = decltype(uc)::value;
          if constexpr (u < NB2 + kAhead) {
            if constexpr (u < NB2) ld_w2(addr, uc, w[u % (kAhead + 1)]);
            if constexpr (u >= kAhead) {
              constexpr int v = u - kAhead;
              constexpr int behind = (u < NB2 ? u : NB2 - 1) - v;
              asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * behind) : "memory");
              claim(w[v % (kAhead + 1)]);
              mfma3(o[v], w[v % (kAhead + 1)], hp);
              if constexpr (v % 3 == 0 && v >= 3 && v / 3 <= NJ) {
                if constexpr ((ABL & 4) == 0) piece_w2(v / 3 - 1, cnext, wslot2);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            self(self, std::integral_constant<int, u + 1>{});
          }
        };
        __builtin_amdgcn_sched_barrier(0);
        units(units, std::integral_constant<int, 0>{});
      }
      next_rslot();
    }
    // the next pass's rows, in flight under this pass's stores (unconditional -- the last pass re-reads rows nobody uses -- so that
    // the registers are dead from the split at the top of a pass to here: behind a condition they would be live throughout)
    load_rows(min(g + (int)gridDim.x, p.groups - 1));
    // ---- Y[16 rb + fr][16 j + 4 fk + 0..3]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int rb = rb0 + i, m = 16 * rb + fr;
#pragma unroll
      for (int j = 0; j < NB2; ++j) {
        const int n = 16 * j + 4 * fk;
        const float4 bi = *reinterpret_cast<const float4*>(inv2 + n);
        const float ir = inv_run[i];
        float4 v = make_float4(o[j][i][0] * (ir * bi.x), o[j][i][1] * (ir * bi.y), o[j][i][2] * (ir * bi.z), o[j][i][3] * (ir * bi.w));
        if constexpr (!BWD) {
          v = f4_add(v, *reinterpret_cast<const float4*>(bia2 + n));
          if (p.colstat) {  // (uniform) per-16-row-block column sums and squared deviations, as in k_gemm2pw
            const int cnt = min(16, p.M - 16 * rb);
            if (cnt > 0) {
              const bool ok = fr < cnt;
              float4 sm = ok ? v : f4_zero();
              sm.x = row16_sum(sm.x); sm.y = row16_sum(sm.y); sm.z = row16_sum(sm.z); sm.w = row16_sum(sm.w);
              const float inv = 1.f / (float)cnt;
              float4 q;
              q.x = ok ? v.x - sm.x * inv : 0.f; q.y = ok ? v.y - sm.y * inv : 0.f;
              q.z = ok ? v.z - sm.z * inv : 0.f; q.w = ok ? v.w - sm.w * inv : 0.f;
              q.x = row16_sum(q.x * q.x); q.y = row16_sum(q.y * q.y); q.z = row16_sum(q.z * q.z); q.w = row16_sum(q.w * q.w);
              if (fr == 0 && n < p.N2) {
                float* cs = p.colstat + (int64_t)rb * 2 * p.N2 + n;
                *reinterpret_cast<float4*>(cs) = sm;
                *reinterpret_cast<float4*>(cs + p.N2) = q;
              }
            }
          }
        }
        if (((ABL & 2) == 0 || p.M < 0) && m < p.M && n < p.N2) *reinterpret_cast<float4*>(p.Y + (int64_t)m * p.ldy + n) = v;
      }
    }
  }
  // the refill runs unconditionally (a branch around a piece inside the pinned loops costs ~90 spilled registers): the last two
  // slices' refills fetch weights nobody reads into slots nobody reads -- they must have landed before this workgroup's LDS is
  // handed to the next one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef PGNN_DS_READ
inline bool fused_shape_ok(int64_t k1, int64_t n1, int64_t n2) {
  return k1 > 288 && k1 <= 320 && n2 > 288 && n2 <= 304 && n1 >= 32 && n1 <= 608 && k1 % 4 == 0 && n1 % 4 == 0 && n2 % 4 == 0;
}

template <bool BWD, int ABL>
int launch_fused_abl(const MlpFusedArgs& p, hipStream_t st) {
  const size_t lds = (size_t)kRing * kSlot;
  const int grid = std::min(num_cu(), p.groups);
  allow_big_lds((const void*)k_mlp2p_fused<10, 19, BWD, ABL>, lds);
  hipLaunchKernelGGL((k_mlp2p_fused<10, 19, BWD, ABL>), dim3(grid), dim3(64 * kFusedWaves), lds, st, p);
  return check_launch(BWD ? "mlp_bwd_data_2p_fused" : "mlp_fwd_2p_fused");
}
template <bool BWD>
int launch_fused(const MlpFusedArgs& p, hipStream_t st) {
#ifdef PGNN_AB  // the ablation instances exist in A/B builds only (python -m pretrain_gnns_amd.build --ab)
  switch (env_knob("PGNN_FUSED_ABL", 0)) {
    case 1: return launch_fused_abl<BWD, 1>(p, st);
    case 2: return launch_fused_abl<BWD, 2>(p, st);
    case 3: return launch_fused_abl<BWD, 3>(p, st);
    case 4: return launch_fused_abl<BWD, 4>(p, st);
    case 7: return launch_fused_abl<BWD, 7>(p, st);
    case 8: return launch_fused_abl<BWD, 8>(p, st);
    case 15: return launch_fused_abl<BWD, 15>(p, st);
    case 16: return launch_fused_abl<BWD, 16>(p, st);
    case 31: return launch_fused_abl<BWD, 31>(p, st);
    default: break;
  }
#endif
  return launch_fused_abl<BWD, 0>(p, st);
}

}  // namespace

int mlp_fused_supported(int64_t m, int64_t k1, int64_t n1, int64_t n2) { return m > 0 && m < (int64_t(1) << 30) && fused_shape_ok(k1, n1, n2); }

int mlp_fwd_2p_fused(const float* x, int64_t ldx, const void* planes1, const float* b1, const void* planes2, const float* b2, float* hid,
                     int64_t ldh, float* y, int64_t ldy, int64_t m, int64_t k1, int64_t n1, int64_t n2, float* colstat, hipStream_t st) {
  PGNN_REQUIRE(mlp_fused_supported(m, k1, n1, n2), "mlp_fwd_2p_fused: shape [m, %lld] -> [m, %lld] -> [m, %lld] not covered", (long long)k1,
               (long long)n1, (long long)n2);
  PGNN_REQUIRE(x && planes1 && planes2 && hid && y && ldx % 4 == 0 && ldh % 4 == 0 && ldy % 4 == 0, "mlp_fwd_2p_fused: pointers / leading dimensions");
  MlpFusedArgs p{};
  p.X = x; p.ldx = ldx;
  p.P1 = static_cast<const unsigned short*>(planes1); p.ld1 = ceil_div(k1, 32) * 32; p.plane1 = n1 * p.ld1;
  p.P2 = static_cast<const unsigned short*>(planes2); p.ld2 = ceil_div(n1, 32) * 32; p.plane2 = n2 * p.ld2;
  p.bias1 = b1; p.bias2 = b2;
  p.H = hid; p.ldh = ldh; p.Y = y; p.ldy = ldy; p.colstat = colstat;
  p.M = (int)m; p.K1 = (int)k1; p.N1 = (int)n1; p.N2 = (int)n2;
  p.groups = (int)ceil_div(m, kFusedRows);
  return launch_fused<false>(p, st);
}

int mlp_bwd_data_2p_fused(const float* dy, int64_t lddy, const void* planes2t, const float* relu_out, int64_t ldr, const void* planes1t,
                          float* dhid, int64_t lddh, float* dx, int64_t lddx, int64_t m, int64_t k1, int64_t n1, int64_t n2, hipStream_t st) {
  PGNN_REQUIRE(mlp_fused_supported(m, k1, n1, n2), "mlp_bwd_data_2p_fused: shape [m, %lld] -> [m, %lld] -> [m, %lld] not covered", (long long)k1,
               (long long)n1, (long long)n2);
  PGNN_REQUIRE(dy && planes2t && planes1t && relu_out && dhid && dx && lddy % 4 == 0 && ldr % 4 == 0 && lddh % 4 == 0 && lddx % 4 == 0,
               "mlp_bwd_data_2p_fused: pointers / leading dimensions");
  MlpFusedArgs p{};
  p.X = dy; p.ldx = lddy;
  p.P1 = static_cast<const unsigned short*>(planes2t); p.ld1 = ceil_div(k1, 32) * 32; p.plane1 = n1 * p.ld1;
  p.P2 = static_cast<const unsigned short*>(planes1t); p.ld2 = ceil_div(n1, 32) * 32; p.plane2 = n2 * p.ld2;
  p.mask = relu_out; p.ldmask = ldr;
  p.H = dhid; p.ldh = lddh; p.Y = dx; p.ldy = lddx;
  p.M = (int)m; p.K1 = (int)k1; p.N1 = (int)n1; p.N2 = (int)n2;
  p.groups = (int)ceil_div(m, kFusedRows);
  return launch_fused<true>(p, st);
}

}  // namespace pgnn

extern "C" {

int pgnn_mlp_2p_fused_supported(int64_t m, int64_t k1, int64_t n1, int64_t n2) { return pgnn::mlp_fused_supported(m, k1, n1, n2); }

int pgnn_mlp_fwd_2p_fused(const float* x, int64_t ldx, const void* wplanes1, const float* b1, const void* wplanes2, const float* b2, float* hid,
                          int64_t ldh, float* y, int64_t ldy, int64_t m, int64_t k1, int64_t n1, int64_t n2, float* colstat, pgnn_stream stream) {
  return pgnn::mlp_fwd_2p_fused(x, ldx, wplanes1, b1, wplanes2, b2, hid, ldh, y, ldy, m, k1, n1, n2, colstat, (hipStream_t)stream);
}

int pgnn_mlp_bwd_data_2p_fused(const float* dy, int64_t lddy, const void* w2tplanes, const float* relu_out, int64_t ldr, const void* w1tplanes,
                               float* dhid, int64_t lddh, float* dx, int64_t lddx, int64_t m, int64_t k1, int64_t n1, int64_t n2,
                               pgnn_stream stream) {
  return pgnn::mlp_bwd_data_2p_fused(dy, lddy, w2tplanes, relu_out, ldr, w1tplanes, dhid, lddh, dx, lddx, m, k1, n1, n2, (hipStream_t)stream);
}

}  // extern "C"
