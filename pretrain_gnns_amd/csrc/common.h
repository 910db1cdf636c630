// Shared host/device helpers for the pgnn HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/pgnn.h"

namespace pgnn {

constexpr int kWave = 64;         // CDNA wavefront

// Compute units and XCDs of the CURRENT device, queried once per device (hipDeviceProp_t.multiProcessorCount; an
// MI355X in SPX mode reports 256 CUs = 8 XCDs x 32, a CPX partition 32 CUs = 1 XCD).  The XCD count is not an API
// field: gfx950 XCDs carry 32 active CUs each.  Used for grid sizing and for the XCD-aware block remap; both are
// speed heuristics -- any value gives correct results.
struct DeviceInfo { int num_cu, num_xcd; };
DeviceInfo device_info();
inline int num_cu() { return device_info().num_cu; }
inline int num_xcd() { return device_info().num_xcd; }

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PGNN_ERR_HIP;
  }
  return PGNN_OK;
}

#define PGNN_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      ::pgnn::set_error(__VA_ARGS__);  \
      return PGNN_ERR_ARG;             \
    }                                  \
  } while (0)

#define PGNN_HIP(call)                                                    \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      ::pgnn::set_error("%s: %s", #call, hipGetErrorString(e_));          \
      return PGNN_ERR_HIP;                                                \
    }                                                                     \
  } while (0)

// dynamic LDS above the 64 KiB default needs an explicit opt-in (gfx950 has 160 KiB per CU).  The attribute
// is sticky per function and device, so it is raised once to the largest size seen, not on every launch
// (hipFuncSetAttribute takes the runtime's locks: microseconds that a launch-bound step does not have).
inline void allow_big_lds(const void* func, size_t bytes) {
  if (bytes <= 64 * 1024) return;
  struct Seen { const void* f; int dev; size_t bytes; };
  constexpr int kSlots = 256;
  static thread_local Seen seen[kSlots];
  static thread_local int nseen = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  for (int i = 0; i < nseen; ++i)
    if (seen[i].f == func && seen[i].dev == dev) {
      if (seen[i].bytes >= bytes) return;
      seen[i].bytes = bytes;
      (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      return;
    }
  (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (nseen < kSlots) {
    seen[nseen++] = Seen{func, dev, bytes};
  } else {
    static thread_local bool warned = false;  // still correct (the attribute is set on every launch), just slower
    if (!warned) fprintf(stderr, "pgnn: allow_big_lds cache full (%d kernels): hipFuncSetAttribute now runs per launch\n", kSlots);
    warned = true;
  }
}

// A/B knobs come from the environment.  A process without any `PGNN_*` variable -- production -- takes every default at the
// cost of one flag test per call site.  With one set, getenv walks the whole environment block (~0.3 us) and a train step asks
// ~140 times, so values are cached per call site (keyed by the literal's address) and only re-read after pgnn_reload_env() --
// which is what a test that flips a knob mid-process calls.
extern unsigned g_env_generation;
extern bool g_env_any;  // is ANY `PGNN_*` variable set?  (scanned at load and by pgnn_reload_env)
inline int env_knob(const char* name, int dflt) {
  if (!g_env_any) return dflt;  // the production path: no A/B variable in the environment, no table to consult
  struct Slot { const char* name; unsigned gen; bool set; int value; };
  constexpr int kSlots = 128;
  static thread_local Slot slots[kSlots];
  static thread_local int nslots = 0;
  const unsigned gen = g_env_generation;
  Slot* s = nullptr;
  for (int i = 0; i < nslots && !s; ++i)
    if (slots[i].name == name) s = &slots[i];
  if (!s) {
    if (nslots == kSlots) {  // more call sites than slots: uncached (correct, ~0.3 us slower per query) -- say so once
      static thread_local bool warned = false;
      if (!warned) fprintf(stderr, "pgnn: env_knob cache full (%d call sites): '%s' is read with getenv on every call\n", kSlots, name);
      warned = true;
      const char* v = getenv(name);
      return v ? atoi(v) : dflt;
    }
    s = &slots[nslots++];
    s->name = name;
    s->gen = gen - 1;
  }
  if (s->gen != gen) {
    const char* v = getenv(name);
    s->set = v != nullptr;
    s->value = v ? atoi(v) : 0;
    s->gen = gen;
  }
  return s->set ? s->value : dflt;  // the default belongs to the call site, not to the cache
}

// (linear.hip) pgnn_split_weights + `nbump` device int64 counters incremented by the same launch (BatchNorm's num_batches_tracked)
// ... and, for the bio stack, the [k+1, dim] edge-encoder tables [W_enc^T; b_enc] of up to 16 layers (W_enc [dim, k], b_enc [dim]):
// the same launch writes them, instead of three torch.cat / stack launches per step
struct EncTables {
  const float* w[16];
  const float* b[16];
  float* dst[16];
  int count, dim, k;
};
int split_weights_bump(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                       int64_t count, int64_t* const* bump, int nbump, hipStream_t stream, const EncTables* tabs = nullptr);
// (linear.hip) the planes products on TWO fp16 planes + a power-of-two scale per row (PGNN_GEMM_2P): the split (same
// jobs as split_weights_bump; the planes and the rows' inverse scales fit the room of three bf16 planes) and the two products
int split_weights_2p(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                     int64_t count, int64_t* const* bump, int nbump, hipStream_t stream, const EncTables* tabs = nullptr,
                     uint32_t* zero_ptr = nullptr, int64_t zero_words = 0);  // + a region of words the launch clears
// x_amax / dy_amax: [m] bit patterns of the rows' largest magnitudes when the producer of the operand left them (else NULL: the
// kernel takes them); y_amax / dx_amax: [m] zeroed words that receive the result rows' largest magnitudes (NULL: not wanted)
struct BnFwdFold;  // (bn_fold.h) non-NULL: the statistics of the BatchNorm behind y are folded inside the product's launch
int linear_fwd_wp_2p(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m, int64_t k,
                     int64_t n, int relu, float* colstat, hipStream_t st, const uint32_t* x_amax = nullptr, uint32_t* y_amax = nullptr,
                     const BnFwdFold* bnf = nullptr);
int linear_bwd_data_wp_2p(const float* dy, int64_t lddy, const void* wtplanes, const float* relu_out, int64_t ldr, float* dx, int64_t lddx,
                          int64_t m, int64_t k, int64_t n, hipStream_t st, const uint32_t* dy_amax = nullptr, uint32_t* dx_amax = nullptr);
// (mlp_fused.hip) both products of a GIN mlp in one launch, H written once and never re-read (pgnn_mlp_fwd_2p_fused / _bwd_data_2p_fused)
int mlp_fused_supported(int64_t m, int64_t k1, int64_t n1, int64_t n2);
int mlp_fwd_2p_fused(const float* x, int64_t ldx, const void* planes1, const float* b1, const void* planes2, const float* b2, float* hid,
                     int64_t ldh, float* y, int64_t ldy, int64_t m, int64_t k1, int64_t n1, int64_t n2, float* colstat, hipStream_t st);
int mlp_bwd_data_2p_fused(const float* dy, int64_t lddy, const void* planes2t, const float* relu_out, int64_t ldr, const void* planes1t,
                          float* dhid, int64_t lddh, float* dx, int64_t lddx, int64_t m, int64_t k1, int64_t n1, int64_t n2, hipStream_t st);
// (linear.hip) pgnn_linear_bwd_weight_pair that can also leave g_out [n_b][12] = dy_b^T . cfeat12 ([m][12]: twelve more columns of
// product b's second operand, riding in the column padding of its last tile) where its one-launch path runs -- *g_done says whether it
// did; linear_bwd_weight_pair_ext_ok: whether it would.  The chem GIN stack's bond-table gradients: demb = (cfeat^T dhid) W1
// (bond_tables_from_g: one launch for up to kMaxBondJobs layers) instead of a pass over dagg per layer; pad_rowfeat12: cfeat [n][kc] ->
// [n][12], zero-filled.
bool linear_bwd_weight_pair_ext_ok(int64_t m, int64_t k_a, int64_t n_a, int64_t k_b, int64_t n_b);
int linear_bwd_weight_pair_ext(const float* dy_a, int64_t lddy_a, const float* x_a, int64_t ldx_a, float* dw_a, float* db_a, int64_t k_a,
                               int64_t n_a, const float* dy_b, int64_t lddy_b, const float* x_b, int64_t ldx_b, float* dw_b, float* db_b,
                               int64_t k_b, int64_t n_b, int64_t m, void* ws, size_t ws_bytes, hipStream_t stream, const float* cfeat12,
                               float* g_out, bool* g_done);
int pad_rowfeat12(const float* cfeat, int64_t kc, float* out12, int64_t n, hipStream_t st);
constexpr int kMaxBondJobs = 16;
struct BondTableJob {
  const float* g;  // [rows][12]
  const float* w;  // [rows][ldw]: W1
  int64_t ldw;
  float* demb;     // [kc][ldd]
  int64_t ldd;
};
int bond_tables_from_g(const BondTableJob* jobs, int count, int64_t rows, int64_t dim, int64_t kc, hipStream_t st);
// (layer.hip) the stop event of the NEXT tiled two-plane product launched by this host thread (hipExtLaunchKernelGGL): the launch
// takes it; whoever set it checks afterwards whether it is still there (another kernel ran: record the event the ordinary way)
void set_next_launch_stop_event(hipEvent_t ev);
hipEvent_t take_next_launch_stop_event();
// (tile.hip) pgnn_neighbor_sum_tiled whose result is zeroed where mask[i, c] <= 0 (the ReLU between two layers, backward), when
// the launch that runs can do it: *mask_applied says whether it did (the pipelined kernel of large batches and the untiled
// fall-back cannot -- the caller masks in a pass of its own then)
int neighbor_sum_tiled_masked(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const float* dinv,
                              const int32_t* tile_start, const int32_t* num_tiles, float* out, int64_t ldo, int64_t num_nodes,
                              int64_t dim, const float* cfeat, int64_t kc, const float* table, int64_t ldt, float* feat_out,
                              int64_t ld_feat_out, const float* mask, int64_t ldm, bool* mask_applied, hipStream_t stream);
// (aggregate.hip) pgnn_rowfeat_matmul_bwd with a strided result: row r < kc, column c of the product goes to out[r * s_row + c * s_col],
// except that with `last_row_out` the LAST row (the bias gradient of an edge encoder whose input carries a ones column) goes
// there, contiguous
int rowfeat_matmul_bwd_strided(const float* cfeat, int64_t kc, const float* g, int64_t ldg, float* out, int64_t s_row, int64_t s_col,
                               float* last_row_out, int64_t n, int64_t dim, void* ws, size_t ws_bytes, hipStream_t stream);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Carve sub-buffers out of a caller-provided workspace (256 B aligned).
struct Carver {
  char* base;
  size_t used = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    T* p = reinterpret_cast<T*>(base + used);
    used += align_up(n * sizeof(T), 256);
    return p;
  }
};

// XCD-aware block remap: consecutive logical blocks land on the same XCD (same L2) instead of
// being dealt round-robin over the `nxcd` XCDs (host: num_xcd()).  Bijective for any grid size and any nxcd >= 1.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks, int nxcd) {
  const int q = nblocks / nxcd, r = nblocks % nxcd;
  const int xcd = bid % nxcd, slot = bid / nxcd;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + slot;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// wave-uniform broadcast of lane `src`'s value into a scalar register
__device__ __forceinline__ int bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Results one block publishes and ANOTHER block of the same launch reads: agent-scope accesses -- write-through stores, loads that
// bypass the non-coherent caches (sc1).  The alternative, __threadfence() around the arrival ticket, is a write-back of the XCD's
// whole L2 (buffer_wbl2 sc1) per block: ~0.4 us each and serialized per XCD -- the 1 820 blocks of the Adam launch spent 88 of
// their 91 us in them, the 253 blocks of the masking head 12 of 27 us.
template <typename T>
__device__ __forceinline__ void publish(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// publish() only ISSUES the store.  On gfx950 stores and loads share `vmcnt`, a workgroup-scope release fence and __syncthreads()
// wait for `lgkmcnt` only, and the store and the ticket atomic that follows go to different L2 channels -- so every thread that
// has published must call publish_commit() (s_waitcnt vmcnt(0): its write-through stores are acknowledged) BEFORE the barrier /
// ticket that announces them (without it the last block could fetch a partial that had not landed).
__device__ __forceinline__ void publish_commit() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ T fetch_published(const T* p) { return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// "The last block to finish folds everybody's partial results": called by ONE thread of every block, behind a __syncthreads()
// that follows the block's publish() stores AND their publish_commit() in every publishing thread; true in exactly one block, which may
// then read every block's results with fetch_published().  No cache maintenance (see above).  Same-address device-scope atomics
// retire at ~50 ns each, so a block first takes a ticket in one of up to 32 group words and only the last of each group takes one
// in the top word.  words: PGNN_TICKET_WORDS uint32, zero before the launch, left zero.
constexpr unsigned kTicketGroups = 32;
__device__ __forceinline__ bool arrive_last(unsigned* words) {
  const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned G = nb < kTicketGroups ? nb : kTicketGroups, g = b % G;
  const unsigned size = nb / G + (g < nb % G ? 1u : 0u);
  publish_commit();  // this thread's own stores (the single-thread publishers call arrive_last right behind publish())
  if (__hip_atomic_fetch_add(words + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != size - 1u) return false;
  publish(words + 1 + g, 0u);
  publish_commit();
  if (__hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != G - 1u) return false;
  publish(words, 0u);
  asm volatile("" ::: "memory");  // the caller's fetch_published() loads stay below the ticket (its value was waited for above)
  return true;
}

// Second-pass reduction helper: 16 lanes share one output column; lane s sums the partials
// b = s, s+16, ... in double, then an xor-butterfly combines the 16 slices (every lane ends with the
// same bits: fixed association order => deterministic).  Call with all 16 lanes of the group active.
__device__ __forceinline__ double slice_sum16(const float* __restrict__ p, size_t stride, int count, int s) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // fixed interleave: 4 loads in flight per lane
  int b = s;
  for (; b + 48 < count; b += 64) {
    a0 += (double)p[(size_t)b * stride];
    a1 += (double)p[(size_t)(b + 16) * stride];
    a2 += (double)p[(size_t)(b + 32) * stride];
    a3 += (double)p[(size_t)(b + 48) * stride];
  }
  for (; b < count; b += 16) a0 += (double)p[(size_t)b * stride];
  double a = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) a += __shfl_xor(a, off);
  return a;
}


// Sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15); every lane of the row ends with the same bits (each step adds
// the two halves of a commutative pair): quad xor 1, quad xor 2, mirror within 8, mirror within 16.  No LDS, four VALU ops.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// The same with a whole wave per output column: lane s sums the partials b = s, s+64, ... in double (four loads in
// flight), then a 64-lane xor butterfly.  Every lane ends with the same bits; fixed association order => deterministic.
__device__ __forceinline__ double slice_sum64(const float* __restrict__ p, size_t stride, int count, int s) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = s;
  for (; b + 192 < count; b += 256) {
    a0 += (double)p[(size_t)b * stride];
    a1 += (double)p[(size_t)(b + 64) * stride];
    a2 += (double)p[(size_t)(b + 128) * stride];
    a3 += (double)p[(size_t)(b + 192) * stride];
  }
  for (; b < count; b += 64) a0 += (double)p[(size_t)b * stride];
  double a = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  return a;
}

}  // namespace pgnn
