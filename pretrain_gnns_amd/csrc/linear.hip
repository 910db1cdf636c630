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
#include "common.h"

namespace pgnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

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
};

enum { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_MASK = 2 };

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ void __launch_bounds__(kThreads) k_gemm(GemmArgs p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BK % 4 == 0, "tile shape");
  constexpr int LDA_S = BM + (A_KMAJOR ? 1 : 4);
  constexpr int LDB_S = BN + (B_KMAJOR ? 1 : 4);
  constexpr int PA = BM * BK / (4 * kThreads), PB = BN * BK / (4 * kThreads);
  static_assert(PA >= 1 && PB >= 1 && (BM * BK) % (4 * kThreads) == 0 && (BN * BK) % (4 * kThreads) == 0, "staging");

  extern __shared__ __align__(16) float smem[];
  auto As = [&](int buf) { return smem + buf * (BK * LDA_S); };
  auto Bs = [&](int buf) { return smem + 2 * BK * LDA_S + buf * (BK * LDB_S); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  float4 ra[PA], rb[PB];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int idx = tid + q * kThreads;
      float4 v = f4_zero();
      if (A_KMAJOR) {
        const int row = idx / (BK / 4), kq = idx % (BK / 4);
        const int m = m0 + row, k = k0 + 4 * kq;
        if (m < p.M && k < kend) v = *reinterpret_cast<const float4*>(p.A + (int64_t)m * p.lda + k);
      } else {
        const int kr = idx / (BM / 4), mq = idx % (BM / 4);
        const int m = m0 + 4 * mq, k = k0 + kr;
        if (m < p.M && k < kend) v = *reinterpret_cast<const float4*>(p.A + (int64_t)k * p.lda + m);
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int idx = tid + q * kThreads;
      float4 v = f4_zero();
      if (B_KMAJOR) {
        const int row = idx / (BK / 4), kq = idx % (BK / 4);
        const int n = n0 + row, k = k0 + 4 * kq;
        if (n < p.N && k < kend) v = *reinterpret_cast<const float4*>(p.B + (int64_t)n * p.ldb + k);
      } else {
        const int kr = idx / (BN / 4), nq = idx % (BN / 4);
        const int n = n0 + 4 * nq, k = k0 + kr;
        if (n < p.N && k < kend) v = *reinterpret_cast<const float4*>(p.B + (int64_t)k * p.ldb + n);
      }
      rb[q] = v;
    }
  };

  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int idx = tid + q * kThreads;
      if (A_KMAJOR) {
        const int row = idx / (BK / 4), kq = idx % (BK / 4);
        float* d = As(buf) + (4 * kq) * LDA_S + row;
        d[0] = ra[q].x; d[LDA_S] = ra[q].y; d[2 * LDA_S] = ra[q].z; d[3 * LDA_S] = ra[q].w;
      } else {
        const int kr = idx / (BM / 4), mq = idx % (BM / 4);
        *reinterpret_cast<float4*>(As(buf) + kr * LDA_S + 4 * mq) = ra[q];
      }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int idx = tid + q * kThreads;
      if (B_KMAJOR) {
        const int row = idx / (BK / 4), kq = idx % (BK / 4);
        float* d = Bs(buf) + (4 * kq) * LDB_S + row;
        d[0] = rb[q].x; d[LDB_S] = rb[q].y; d[2 * LDB_S] = rb[q].z; d[3 * LDB_S] = rb[q].w;
      } else {
        const int kr = idx / (BN / 4), nq = idx % (BN / 4);
        *reinterpret_cast<float4*>(Bs(buf) + kr * LDB_S + 4 * nq) = rb[q];
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

  if (nk > 0) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) load_tiles(kbeg + (it + 1) * BK);
    const float* a_base = As(buf) + fk * LDA_S + wm0 + fr;
    const float* b_base = Bs(buf) + fk * LDB_S + wn0 + fr;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = a_base[kk * LDA_S + i * 16];
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = b_base[kk * LDB_S + j * 16];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds C[row = (lane>>4)*4 + r][col = lane&15] of each 16x16 block
  float* C = p.C + (int64_t)blockIdx.y * p.split_stride;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn0 + j * 16 + fr;
    if (n >= p.N) continue;
    float bv = 0.f;
    if (EPI == EPI_BIAS && p.bias) bv = p.bias[n];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm0 + i * 16 + fk * 4 + r;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (EPI == EPI_BIAS) {
          v += bv;
          if (p.relu) v = fmaxf(v, 0.f);
        }
        if (EPI == EPI_MASK) {
          if (!(p.mask[(int64_t)m * p.ldmask + n] > 0.f)) v = 0.f;
        }
        C[(int64_t)m * p.ldc + n] = v;
      }
    }
  }
}

// dst[i] = sum_z partial[z][i]  (fixed order), float4
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ partial, int nsplit,
                                                       int64_t stride, float* __restrict__ dst, int64_t n4) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(partial)[q];
    for (int z = 1; z < nsplit; ++z) s = f4_add(s, reinterpret_cast<const float4*>(partial + z * stride)[q]);
    reinterpret_cast<float4*>(dst)[q] = s;
  }
}

// column sums of dy[M, N] -> partial[blk][N] (float4 columns x 4 row lanes per block); then final
__global__ void k_colsum_partial(const float* __restrict__ dy, int64_t ld, int m, int n4,
                                 float* __restrict__ partial) {
  extern __shared__ __align__(16) float lds[];  // [4][n]
  const int t = threadIdx.x, c4 = t % n4, rl = t / n4, n = n4 * 4;
  const int per = (m + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(m, r0 + per);
  float4 s = f4_zero();
  if (rl < 4) {
    for (int r = r0 + rl; r < r1; r += 4) s = f4_add(s, reinterpret_cast<const float4*>(dy + (int64_t)r * ld)[c4]);
    reinterpret_cast<float4*>(lds + rl * n)[c4] = s;
  }
  __syncthreads();
  for (int q = t; q < n; q += blockDim.x)
    partial[(size_t)blockIdx.x * n + q] = (lds[q] + lds[n + q]) + (lds[2 * n + q] + lds[3 * n + q]);
}
__global__ void __launch_bounds__(256) k_colsum_final(const float* __restrict__ partial, int nblk, int n,
                                                      float* __restrict__ out) {
  const int sl = threadIdx.x & 15;
  const int cc = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = min(cc, n - 1);
  const double s = slice_sum16(partial + c, (size_t)n, nblk, sl);
  if (sl == 0 && cc < n) out[c] = (float)s;
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool A_KMAJOR, bool B_KMAJOR, int EPI>
int launch_gemm(const GemmArgs& p, int nsplit, hipStream_t st) {
  constexpr int LDA_S = BM + (A_KMAJOR ? 1 : 4), LDB_S = BN + (B_KMAJOR ? 1 : 4);
  constexpr size_t lds = (size_t)2 * BK * (LDA_S + LDB_S) * sizeof(float);
  static_assert(lds <= 64 * 1024, "LDS budget");
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  hipLaunchKernelGGL((k_gemm<BM, BN, BK, WAVES_M, WAVES_N, A_KMAJOR, B_KMAJOR, EPI>), dim3(tiles, nsplit),
                     dim3(kThreads), lds, st, p);
  return check_launch("gemm");
}

inline int colsum_blocks(int64_t m) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(m, 32), 1), 1024); }

constexpr int kWgtBM = 64, kWgtBN = 64, kWgtBK = 16;

inline int weight_splits(int64_t m, int64_t k, int64_t n) {
  // enough (tile x split) blocks to fill the chip ~2x, at least 4 k-tiles per split
  const int64_t tiles = ceil_div(n, kWgtBM) * ceil_div(k, kWgtBN);
  int64_t s = ceil_div(2 * kNumCU, tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(m / (4 * kWgtBK), 1));
  return (int)std::max<int64_t>(std::min<int64_t>(s, 256), 1);
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" {

int pgnn_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                    int64_t m, int64_t k, int64_t n, int relu, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && ldx % 4 == 0, "linear_fwd: K and ldx must be multiples of 4");
  GemmArgs p{};
  p.A = x; p.lda = ldx; p.B = w; p.ldb = k; p.C = y; p.ldc = ldy;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.bias = bias; p.relu = relu; p.kchunk = (int)k; p.split_stride = 0;
  return launch_gemm<128, 128, 16, 2, 2, true, true, EPI_BIAS>(p, 1, (hipStream_t)stream);
}

int pgnn_linear_bwd_data(const float* dy, int64_t lddy, const float* w, const float* relu_out, int64_t ldr,
                         float* dx, int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0,
               "linear_bwd_data: K, N and lddy must be multiples of 4");
  GemmArgs p{};
  // C = dx [m, k] ; reduction over n ; A = dy (n contiguous) ; B(kcol, nn) = w[nn*k + kcol]
  p.A = dy; p.lda = lddy; p.B = w; p.ldb = k; p.C = dx; p.ldc = lddx;
  p.M = (int)m; p.N = (int)k; p.K = (int)n; p.mask = relu_out; p.ldmask = ldr; p.kchunk = (int)n; p.split_stride = 0;
  if (relu_out) return launch_gemm<128, 128, 16, 2, 2, true, false, EPI_MASK>(p, 1, (hipStream_t)stream);
  return launch_gemm<128, 128, 16, 2, 2, true, false, EPI_PLAIN>(p, 1, (hipStream_t)stream);
}

size_t pgnn_linear_bwd_weight_workspace_bytes(int64_t m, int64_t k, int64_t n) {
  return align_up((size_t)weight_splits(m, k, n) * n * k * sizeof(float), 256) +
         align_up((size_t)colsum_blocks(m) * n * sizeof(float), 256);
}

int pgnn_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw, float* db,
                           int64_t m, int64_t k, int64_t n, void* ws, size_t ws_bytes, pgnn_stream stream) {
  PGNN_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0,
               "linear_bwd_weight: K, N and leading dimensions must be multiples of 4");
  PGNN_REQUIRE(n <= 1024, "linear_bwd_weight: output width > 1024 not supported");
  if (ws_bytes < pgnn_linear_bwd_weight_workspace_bytes(m, k, n)) {
    set_error("linear_bwd_weight workspace too small");
    return PGNN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws);
  const int nsplit = weight_splits(m, k, n);
  float* partial = cv.take<float>((size_t)nsplit * n * k);
  float* colpart = cv.take<float>((size_t)colsum_blocks(m) * n);
  GemmArgs p{};
  // C = dW [n, k] ; reduction over rows m ; A(nout, r) = dy[r*lddy + nout] ; B(kcol, r) = x[r*ldx + kcol]
  p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx;
  p.M = (int)n; p.N = (int)k; p.K = (int)m;
  int64_t chunk = ceil_div(m, nsplit);
  chunk = ceil_div(chunk, kWgtBK) * kWgtBK;
  p.kchunk = (int)chunk;
  const int used = (int)ceil_div(m, chunk);
  if (used == 1) {
    p.C = dw; p.ldc = k; p.split_stride = 0;
  } else {
    p.C = partial; p.ldc = k; p.split_stride = n * k;
  }
  int rc = launch_gemm<kWgtBM, kWgtBN, kWgtBK, 2, 2, false, false, EPI_PLAIN>(p, used, st);
  if (rc) return rc;
  if (used > 1) {
    const int64_t n4 = n * k / 4;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((int)std::min<int64_t>(ceil_div(n4, 256), 1024)), dim3(256), 0, st,
                       partial, used, n * k, dw, n4);
  }
  if (db) {
    const int nb = colsum_blocks(m);
    hipLaunchKernelGGL(k_colsum_partial, dim3(nb), dim3((int)align_up((size_t)n, 64)), (size_t)4 * n * sizeof(float), st,
                       dy, lddy, (int)m, (int)(n / 4), colpart);
    hipLaunchKernelGGL(k_colsum_final, dim3((int)ceil_div(n, 16)), dim3(256), 0, st, colpart, nb, (int)n, db);
  }
  return check_launch("linear_bwd_weight");
}

}  // extern "C"
