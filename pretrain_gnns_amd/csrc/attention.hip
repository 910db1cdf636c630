// Attention-flavoured layers of the class surface: the 2-head GATConv of chem/model.py:107-162 and
// bio/model.py:117-180 as CSR kernels, and the segment soft-max / segment max that GlobalAttention, Set2Set and
// global_max_pool (chem/model.py:322-339) reduce to.  Off the north-star path (GIN / GCN), but native: no atomics, every
// sum sequential in a fixed order (bitwise reproducible), one thread per (segment, head) for the scalar passes -- molecule
// and ego-net segments are a handful to a few dozen items -- and D/4-thread groups / one wave per node for the row passes.
//
// GAT, per destination node i with in-edges e (CSR by destination, original order) and the self loop LAST, heads h:
//   m_eh    = xh[src_e, h, :] + T_h[code_e, :]                 (T_h = emb1[type] + emb2[dir], head h's columns)
//   z_eh    = leaky_relu(xh[i,h,:].att_i[h] + m_eh.att_j[h])   = leaky(sd[i,h] + ss[src_e,h] + c[code_e,h])
//   a_eh    = exp(z_eh - max(0, max_e z_eh)) / (sum_e exp(..) + 1e-16)      (torch_geometric 1.0.3 softmax on
//             torch_scatter 1.1.2, whose scatter_max output is pre-filled with 0)
//   out[i]  = mean_h sum_e a_eh m_eh + bias
// Logits / weights live in "extended slot" arrays [E + N, H]: slot p of node i at p + i, its self loop at in_ptr[i+1] + i.
//
// bio GATConv (bio/model.py:117-180): the edge term is edge_encoder(attr_e) = W_enc attr_e + b instead of a table row.  By
// linearity nothing of size [E, 2D] is ever formed: with per-slot features f_e = [attr_e (9), 1] (CSR order) and
// Tenc = [W_enc^T; b] ([10, H*D]),  m_eh.att_j[h] = xh[src].att_j + f_e . wv[:,h]  (wv = Tenc_h . att_j[h], parameter space),
// sum_e a_eh m_eh = sum_e a_eh xh[src_e,h] + (sum_e a_eh f_e) . Tenc_h, and the backward's per-edge rows are rebuilt on the
// fly from f_e and Tenc (10 x 600 floats in LDS).  The chem kernels take the same path with `code` replaced by `sfeat`.
#include "common.h"

using namespace pgnn;

namespace {

constexpr int kBlock = 256;
constexpr int kHeads = 2;          // the reference's GATConv default, the only value its GNN constructs
constexpr int kCodes = 18;         // bond type (6) x direction (3)
constexpr int kSelfCode = 4 * 3;   // self loop: type 4, direction 0
constexpr float kSoftEps = 1e-16f;
constexpr int kMaxKF = 10;         // per-slot feature columns of the bio form (9 attributes + the constant 1)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

inline int grid_rows(int64_t rows, int per_block) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(rows, per_block), (int64_t)num_cu() * 16));
}

// s[i, h] = xh[i,h,:] . att[h, 0:D]   (destination term),  s[i, H + h] = xh[i,h,:] . att[h, D:2D]   (source term)
__global__ void __launch_bounds__(kBlock) k_rowdot(const float* __restrict__ xh, int64_t ldx, const float* __restrict__ att,
                                                   float* __restrict__ s, int n, int d) {
  const int lane = threadIdx.x & 63, d4 = d >> 2;
  for (int64_t i = blockIdx.x * (int64_t)(kBlock / 64) + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * (kBlock / 64)) {
#pragma unroll
    for (int h = 0; h < kHeads; ++h) {
      float ai = 0.f, aj = 0.f;
      for (int c = lane; c < d4; c += 64) {
        const float4 v = reinterpret_cast<const float4*>(xh + i * ldx + h * d)[c];
        ai += dot4(v, reinterpret_cast<const float4*>(att + h * 2 * d)[c]);
        aj += dot4(v, reinterpret_cast<const float4*>(att + h * 2 * d + d)[c]);
      }
      ai = wave_sum(ai);
      aj = wave_sum(aj);
      if (lane == 0) {
        s[i * 2 * kHeads + h] = ai;
        s[i * 2 * kHeads + kHeads + h] = aj;
      }
    }
  }
}

// logits + soft-max of every node's in-segment (+ self), one thread per (node, head).  ctab [kCodes or 1][H] holds the
// bond term m.att_j's table part (chem) -- bio passes per-slot terms in `cslot` [E + N, H] instead (ctab == NULL).
// Also emits cfa[h][i][0:9] = sum of a_eh / H per bond type (0..5) and direction (6..8): the weights with which dout[i]
// enters the bond-embedding gradients (chem only, cfa may be NULL).
__global__ void __launch_bounds__(kBlock) k_gat_alpha_fwd(const float* __restrict__ s, const int32_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ src, const uint8_t* __restrict__ code,
                                                          const float* __restrict__ ctab, const float* __restrict__ sfeat,
                                                          const float* __restrict__ self_feat, const float* __restrict__ wv, int kf,
                                                          float slope, float* __restrict__ z, float* __restrict__ alpha,
                                                          float* __restrict__ cfa, int n) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)n * kHeads) return;
  const int i = (int)(t / kHeads), h = (int)(t % kHeads);
  const int beg = ptr[i], end = ptr[i + 1];
  const float sd = s[(int64_t)i * 2 * kHeads + h];
  float w[kMaxKF];
  if (sfeat)
    for (int k = 0; k < kf; ++k) w[k] = wv[k * kHeads + h];
  auto edge_term = [&](int p) {  // the bond / attribute part of the logit
    if (!sfeat) return ctab[(p == end ? kSelfCode : code[p]) * kHeads + h];
    const float* f = p == end ? self_feat : sfeat + (int64_t)p * kf;
    float c = 0.f;
    for (int k = 0; k < kf; ++k) c = fmaf(f[k], w[k], c);
    return c;
  };
  float mx = 0.f;  // scatter_max's fill value
  for (int p = beg; p <= end; ++p) {
    float v = sd + s[(int64_t)(p == end ? i : src[p]) * 2 * kHeads + kHeads + h] + edge_term(p);
    v = v > 0.f ? v : v * slope;
    z[(int64_t)(p + i) * kHeads + h] = v;
    mx = fmaxf(mx, v);
  }
  float sum = 0.f;
  for (int p = beg; p <= end; ++p) {
    const float u = expf(z[(int64_t)(p + i) * kHeads + h] - mx);
    alpha[(int64_t)(p + i) * kHeads + h] = u;
    sum += u;
  }
  const float inv = 1.f / (sum + kSoftEps);
  float cf[kMaxKF];
  const int ncf = sfeat ? kf : 9;
  for (int k = 0; k < ncf; ++k) cf[k] = 0.f;
  for (int p = beg; p <= end; ++p) {
    const float a = alpha[(int64_t)(p + i) * kHeads + h] * inv;
    alpha[(int64_t)(p + i) * kHeads + h] = a;
    const float ah = a * (1.f / kHeads);
    if (sfeat) {
      const float* f = p == end ? self_feat : sfeat + (int64_t)p * kf;
      for (int k = 0; k < kf; ++k) cf[k] = fmaf(ah, f[k], cf[k]);
    } else {
      const int cd = p == end ? kSelfCode : code[p];
      cf[cd / 3] += ah;
      cf[6 + cd % 3] += ah;
    }
  }
  for (int k = 0; k < ncf; ++k) cfa[((int64_t)h * n + i) * ncf + k] = cf[k];
}

// out[i, :] = mean_h sum_e a_eh (xh[src_e,h,:] + T_h[code_e,:]) + bias.  D/4 threads per node, each owning one float4
// column of BOTH heads; sums sequential in edge order, self loop last.  tab: chem bond table in LDS (emb1 / emb2 rows are
// H*D wide); bio: per-slot edge embeddings ee [E + N, H*D] read from memory (emb1 == NULL).
__global__ void __launch_bounds__(320) k_gat_aggregate_fwd(const float* __restrict__ xh, int64_t ldx,
                                                           const int32_t* __restrict__ ptr, const int32_t* __restrict__ src,
                                                           const uint8_t* __restrict__ code, const float* __restrict__ emb1,
                                                           const float* __restrict__ emb2, const float* __restrict__ ee,
                                                           const float* __restrict__ alpha, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t ldo, int n, int d, int groups) {
#pragma clang fp contract(off)
  extern __shared__ __align__(16) float T[];  // [kCodes][H*D] (chem)
  const int hd = kHeads * d;
  if (emb1) {
    for (int q = threadIdx.x; q < kCodes * hd; q += blockDim.x) {
      const int c = q / hd, k = q - c * hd;
      T[q] = emb1[(c / 3) * hd + k] + emb2[(c % 3) * hd + k];
    }
    __syncthreads();
  }
  const int gs = d >> 2, g = threadIdx.x / gs, c4 = threadIdx.x - g * gs;
  if (g >= groups) return;
  const float4 b4 = reinterpret_cast<const float4*>(bias)[c4];
  for (int64_t i = blockIdx.x * (int64_t)groups + g; i < n; i += (int64_t)gridDim.x * groups) {
    const int beg = ptr[i], end = ptr[i + 1];
    float4 acc[kHeads] = {f4_zero(), f4_zero()};
    for (int p = beg; p <= end; ++p) {
      const bool self = p == end;
      const int64_t j = self ? i : src[p];
#pragma unroll
      for (int h = 0; h < kHeads; ++h) {
        float4 m = reinterpret_cast<const float4*>(xh + j * ldx + h * d)[c4];
        if (emb1) m = f4_add(m, reinterpret_cast<const float4*>(T + (self ? kSelfCode : code[p]) * hd + h * d)[c4]);
        else if (ee) m = f4_add(m, reinterpret_cast<const float4*>(ee + (int64_t)(p + i) * hd + h * d)[c4]);
        acc[h] = f4_add(acc[h], f4_scale(m, alpha[(int64_t)(p + i) * kHeads + h]));
      }
    }
    float4 o = f4_scale(f4_add(acc[0], acc[1]), 1.f / kHeads);
    reinterpret_cast<float4*>(out + i * ldo)[c4] = f4_add(o, b4);  // (feature form: the caller adds cfa . Tenc afterwards)
  }
}

// dalpha[slot, h] = (g[i,:] / H) . (xh[src,h,:] + T_h[code,:]): one wave per node.  Feature form: the edge row is rebuilt as
// sum_k f[k] Tenc[k, h, :] from the 10 x H*D table in LDS.
__global__ void __launch_bounds__(kBlock) k_gat_edge_dot(const float* __restrict__ g, int64_t ldg, const float* __restrict__ xh,
                                                         int64_t ldx, const int32_t* __restrict__ ptr,
                                                         const int32_t* __restrict__ src, const uint8_t* __restrict__ code,
                                                         const float* __restrict__ emb1, const float* __restrict__ emb2,
                                                         const float* __restrict__ sfeat, const float* __restrict__ self_feat,
                                                         const float* __restrict__ tenc, int kf, float* __restrict__ dalpha, int n,
                                                         int d) {
  extern __shared__ __align__(16) float T[];
  const int hd = kHeads * d;
  if (emb1) {
    for (int q = threadIdx.x; q < kCodes * hd; q += blockDim.x) {
      const int c = q / hd, k = q - c * hd;
      T[q] = emb1[(c / 3) * hd + k] + emb2[(c % 3) * hd + k];
    }
    __syncthreads();
  } else if (sfeat) {
    for (int q = threadIdx.x; q < kf * hd; q += blockDim.x) T[q] = tenc[q];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, d4 = d >> 2;
  for (int64_t i = blockIdx.x * (int64_t)(kBlock / 64) + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * (kBlock / 64)) {
    const int beg = ptr[i], end = ptr[i + 1];
    for (int p = beg; p <= end; ++p) {
      const bool self = p == end;
      const int64_t j = self ? i : src[p];
      const float* f = sfeat ? (self ? self_feat : sfeat + (int64_t)p * kf) : nullptr;
      float acc[kHeads] = {0.f, 0.f};
      for (int c = lane; c < d4; c += 64) {
        const float4 gv = reinterpret_cast<const float4*>(g + i * ldg)[c];
#pragma unroll
        for (int h = 0; h < kHeads; ++h) {
          float4 m = reinterpret_cast<const float4*>(xh + j * ldx + h * d)[c];
          if (emb1) {
            m = f4_add(m, reinterpret_cast<const float4*>(T + (self ? kSelfCode : code[p]) * hd + h * d)[c]);
          } else if (sfeat) {
            for (int k = 0; k < kf; ++k) {
              const float4 tv = reinterpret_cast<const float4*>(T + k * hd + h * d)[c];
              m.x = fmaf(f[k], tv.x, m.x); m.y = fmaf(f[k], tv.y, m.y); m.z = fmaf(f[k], tv.z, m.z); m.w = fmaf(f[k], tv.w, m.w);
            }
          }
          acc[h] += dot4(gv, m);
        }
      }
#pragma unroll
      for (int h = 0; h < kHeads; ++h) {
        const float v = wave_sum(acc[h]);
        if (lane == 0) dalpha[(int64_t)(p + i) * kHeads + h] = v * (1.f / kHeads);
      }
    }
  }
}

// soft-max + leaky-relu backward per (node, head): dz = a (da - sum a da) * (z > 0 ? 1 : slope); dsd[h][i][0] = sum dz
// (destination term) ; czf[h][i][0:9] = sum of dz per bond type / direction (chem).  dz overwrites dalpha.
__global__ void __launch_bounds__(kBlock) k_gat_alpha_bwd(const float* __restrict__ alpha, const float* __restrict__ z,
                                                          float* __restrict__ dalpha, const int32_t* __restrict__ ptr,
                                                          const uint8_t* __restrict__ code, const float* __restrict__ sfeat,
                                                          const float* __restrict__ self_feat, int kf, float slope,
                                                          float* __restrict__ dsd, float* __restrict__ czf, int n) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)n * kHeads) return;
  const int i = (int)(t / kHeads), h = (int)(t % kHeads);
  const int beg = ptr[i], end = ptr[i + 1];
  float dotp = 0.f;
  for (int p = beg; p <= end; ++p) dotp += alpha[(int64_t)(p + i) * kHeads + h] * dalpha[(int64_t)(p + i) * kHeads + h];
  float tot = 0.f;
  float cf[kMaxKF];
  const int ncf = sfeat ? kf : 9;
  for (int k = 0; k < ncf; ++k) cf[k] = 0.f;
  for (int p = beg; p <= end; ++p) {
    const int64_t q = (int64_t)(p + i) * kHeads + h;
    float dz = alpha[q] * (dalpha[q] - dotp);
    if (!(z[q] > 0.f)) dz *= slope;
    dalpha[q] = dz;
    tot += dz;
    if (sfeat) {
      const float* f = p == end ? self_feat : sfeat + (int64_t)p * kf;
      for (int k = 0; k < kf; ++k) cf[k] = fmaf(dz, f[k], cf[k]);
    } else {
      const int cd = p == end ? kSelfCode : code[p];
      cf[cd / 3] += dz;
      cf[6 + cd % 3] += dz;
    }
  }
  dsd[((int64_t)h * n + i) * 2 + 0] = tot;
  for (int k = 0; k < ncf; ++k) czf[((int64_t)h * n + i) * ncf + k] = cf[k];
}

// per SOURCE node j (CSR by source): find, for each out-edge q -> dst, the matching slot in dst's in-list (the r-th edge
// from j there, r = number of earlier out-edges of j to the same dst: parallel bonds stay paired) and collect
//   dsd[h][j][1] = sum_q dz[slot] + dz[self slot of j]     (source term of the logits)
//   wout[q, h]   = alpha[slot]                             (the weight the transposed aggregation needs)
__global__ void __launch_bounds__(kBlock) k_gat_src_gather(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src,
                                                           const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                           const float* __restrict__ alpha, const float* __restrict__ dz,
                                                           float* __restrict__ dsd, float* __restrict__ wout, int n) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)n * kHeads) return;
  const int j = (int)(t / kHeads), h = (int)(t % kHeads);
  const int ob = out_ptr[j], oe = out_ptr[j + 1];
  float tot = dz[(int64_t)(in_ptr[j + 1] + j) * kHeads + h];
  for (int q = ob; q < oe; ++q) {
    const int dnode = out_dst[q];
    int r = 0;
    for (int q2 = ob; q2 < q; ++q2) r += out_dst[q2] == dnode;
    int slot = -1;
    for (int p = in_ptr[dnode]; p < in_ptr[dnode + 1]; ++p)
      if (in_src[p] == j && r-- == 0) { slot = p; break; }
    const int64_t s = (int64_t)(slot + dnode) * kHeads + h;
    tot += dz[s];
    wout[(int64_t)q * kHeads + h] = alpha[s];
  }
  dsd[((int64_t)h * n + j) * 2 + 1] = tot;
}

// dxh[j,h,:] = (1/H) (sum_q wout[q,h] g[dst_q,:] + a_self[j,h] g[j,:]) + dsd[h][j][0] att_i[h,:] + dsd[h][j][1] att_j[h,:]
__global__ void __launch_bounds__(320) k_gat_aggregate_bwd(const float* __restrict__ g, int64_t ldg,
                                                           const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr,
                                                           const int32_t* __restrict__ out_dst, const float* __restrict__ wout,
                                                           const float* __restrict__ alpha, const float* __restrict__ dsd,
                                                           const float* __restrict__ att, float* __restrict__ dxh, int64_t ldd,
                                                           int n, int d, int groups) {
#pragma clang fp contract(off)
  const int gs = d >> 2, gidx = threadIdx.x / gs, c4 = threadIdx.x - gidx * gs;
  if (gidx >= groups) return;
  float4 ai[kHeads], aj[kHeads];
#pragma unroll
  for (int h = 0; h < kHeads; ++h) {
    ai[h] = reinterpret_cast<const float4*>(att + h * 2 * d)[c4];
    aj[h] = reinterpret_cast<const float4*>(att + h * 2 * d + d)[c4];
  }
  for (int64_t j = blockIdx.x * (int64_t)groups + gidx; j < n; j += (int64_t)gridDim.x * groups) {
    float4 acc[kHeads] = {f4_zero(), f4_zero()};
    for (int q = out_ptr[j]; q < out_ptr[j + 1]; ++q) {
      const float4 gv = reinterpret_cast<const float4*>(g + (int64_t)out_dst[q] * ldg)[c4];
#pragma unroll
      for (int h = 0; h < kHeads; ++h) acc[h] = f4_add(acc[h], f4_scale(gv, wout[(int64_t)q * kHeads + h]));
    }
    const float4 gs_ = reinterpret_cast<const float4*>(g + j * ldg)[c4];
#pragma unroll
    for (int h = 0; h < kHeads; ++h) {
      acc[h] = f4_add(acc[h], f4_scale(gs_, alpha[(int64_t)(in_ptr[j + 1] + j) * kHeads + h]));
      float4 o = f4_scale(acc[h], 1.f / kHeads);
      o = f4_add(o, f4_scale(ai[h], dsd[((int64_t)h * n + j) * 2 + 0]));
      o = f4_add(o, f4_scale(aj[h], dsd[((int64_t)h * n + j) * 2 + 1]));
      reinterpret_cast<float4*>(dxh + j * ldd + h * d)[c4] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// generic segment soft-max (GlobalAttention gate, Set2Set attention): z [items, H] -> alpha, one thread per (segment, head)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_segment_softmax_fwd(const float* __restrict__ z, const int32_t* __restrict__ ptr,
                                                                const int32_t* __restrict__ perm, float* __restrict__ alpha,
                                                                int nseg, int heads) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)nseg * heads) return;
  const int s = (int)(t / heads), h = (int)(t % heads);
  auto at = [&](int p) { return (int64_t)(perm ? perm[p] : p) * heads + h; };  // items of a segment: perm[ptr[s] .. ptr[s+1])
  float mx = 0.f;
  for (int p = ptr[s]; p < ptr[s + 1]; ++p) mx = fmaxf(mx, z[at(p)]);
  float sum = 0.f;
  for (int p = ptr[s]; p < ptr[s + 1]; ++p) {
    const float u = expf(z[at(p)] - mx);
    alpha[at(p)] = u;
    sum += u;
  }
  const float inv = 1.f / (sum + kSoftEps);
  for (int p = ptr[s]; p < ptr[s + 1]; ++p) alpha[at(p)] *= inv;
}

__global__ void __launch_bounds__(kBlock) k_segment_softmax_bwd(const float* __restrict__ alpha, const float* __restrict__ dalpha,
                                                                const int32_t* __restrict__ ptr, const int32_t* __restrict__ perm,
                                                                float* __restrict__ dz, int nseg, int heads) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)nseg * heads) return;
  const int s = (int)(t / heads), h = (int)(t % heads);
  auto at = [&](int p) { return (int64_t)(perm ? perm[p] : p) * heads + h; };
  float dotp = 0.f;
  for (int p = ptr[s]; p < ptr[s + 1]; ++p) dotp += alpha[at(p)] * dalpha[at(p)];
  for (int p = ptr[s]; p < ptr[s + 1]; ++p) dz[at(p)] = alpha[at(p)] * (dalpha[at(p)] - dotp);
}

// segment max with arg (global_max_pool): one thread per (segment, float4 column); first maximum wins, empty segment -> 0
// (torch_geometric 1.0.3's scatter_('max') replaces the fill value by 0)
__global__ void __launch_bounds__(kBlock) k_segment_max_fwd(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ perm, float* __restrict__ out, int64_t ldo,
                                                            int32_t* __restrict__ arg, int nseg, int d4) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)nseg * d4) return;
  const int s = (int)(t / d4), c = (int)(t % d4);
  float4 best = make_float4(-1e38f, -1e38f, -1e38f, -1e38f);
  int a0 = -1, a1 = -1, a2 = -1, a3 = -1;
  for (int p = ptr[s]; p < ptr[s + 1]; ++p) {
    const int i = perm ? perm[p] : p;
    const float4 v = reinterpret_cast<const float4*>(x + (int64_t)i * ldx)[c];
    if (v.x > best.x) { best.x = v.x; a0 = i; }
    if (v.y > best.y) { best.y = v.y; a1 = i; }
    if (v.z > best.z) { best.z = v.z; a2 = i; }
    if (v.w > best.w) { best.w = v.w; a3 = i; }
  }
  if (a0 < 0) best.x = 0.f;
  if (a1 < 0) best.y = 0.f;
  if (a2 < 0) best.z = 0.f;
  if (a3 < 0) best.w = 0.f;
  reinterpret_cast<float4*>(out + (int64_t)s * ldo)[c] = best;
  reinterpret_cast<int4*>(arg + (int64_t)s * d4 * 4)[c] = make_int4(a0, a1, a2, a3);
}

// dx[i, :] = g[key[i], :] where arg[key[i]] == i, else 0: one thread per (item, float4 column)
__global__ void __launch_bounds__(kBlock) k_segment_max_bwd(const float* __restrict__ g, int64_t ldg, const int64_t* __restrict__ key,
                                                            const int32_t* __restrict__ arg, float* __restrict__ dx, int64_t ldd,
                                                            int nseg, int nitems, int d4) {
  const int64_t t = blockIdx.x * (int64_t)kBlock + threadIdx.x;
  if (t >= (int64_t)nitems * d4) return;
  const int i = (int)(t / d4), c = (int)(t % d4);
  const int64_t sk = key[i];
  float4 o = f4_zero();
  if (sk >= 0 && sk < nseg) {
    const int4 a = reinterpret_cast<const int4*>(arg + sk * d4 * 4)[c];
    const float4 gv = reinterpret_cast<const float4*>(g + sk * ldg)[c];
    o = make_float4(a.x == i ? gv.x : 0.f, a.y == i ? gv.y : 0.f, a.z == i ? gv.z : 0.f, a.w == i ? gv.w : 0.f);
  }
  reinterpret_cast<float4*>(dx + (int64_t)i * ldd)[c] = o;
}

}  // namespace

extern "C" {

int pgnn_gat_fwd(const float* xh, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const uint8_t* in_code,
                 const float* emb1, const float* emb2, const float* ctab, const float* slot_feat, const float* self_feat,
                 const float* wv, int64_t kf, const float* att, const float* bias, float negative_slope, float* scores,
                 float* z, float* alpha, float* cfa, float* out, int64_t ldo, int64_t num_nodes, int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(num_nodes > 0 && dim > 0 && dim % 4 == 0 && dim <= 1280 && ldx % 4 == 0 && ldo % 4 == 0, "gat_fwd: bad shape");
  const bool chem = emb1 != nullptr;
  PGNN_REQUIRE(chem ? (emb2 && ctab && in_code && !slot_feat) : (slot_feat && self_feat && wv && kf > 0 && kf <= kMaxKF),
               "gat_fwd: pass either the chem tables (emb1, emb2, ctab, in_code) or the per-slot features (slot_feat, self_feat, wv, kf <= %d)",
               kMaxKF);
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)num_nodes, d = (int)dim;
  hipLaunchKernelGGL(k_rowdot, dim3(grid_rows(n, kBlock / 64)), dim3(kBlock), 0, st, xh, ldx, att, scores, n, d);
  hipLaunchKernelGGL(k_gat_alpha_fwd, dim3((int)ceil_div((int64_t)n * kHeads, kBlock)), dim3(kBlock), 0, st, scores, in_ptr, in_src,
                     in_code, ctab, slot_feat, self_feat, wv, (int)kf, negative_slope, z, alpha, cfa, n);
  const int gs = d / 4, groups = std::max(1, 320 / gs);
  PGNN_REQUIRE(gs <= 320, "gat_fwd: dim too wide");
  const size_t lds = chem ? (size_t)kCodes * kHeads * d * sizeof(float) : 0;
  allow_big_lds((const void*)k_gat_aggregate_fwd, lds);
  hipLaunchKernelGGL(k_gat_aggregate_fwd, dim3(grid_rows(n, groups)), dim3(320), lds, st, xh, ldx, in_ptr, in_src, in_code, emb1,
                     emb2, (const float*)nullptr, alpha, bias, out, ldo, n, d, groups);
  return check_launch("gat_fwd");
}

int pgnn_gat_bwd(const float* g, int64_t ldg, const float* xh, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src,
                 const uint8_t* in_code, const int32_t* out_ptr, const int32_t* out_dst, const float* emb1, const float* emb2,
                 const float* slot_feat, const float* self_feat, const float* tenc, int64_t kf, const float* att,
                 float negative_slope, const float* z, const float* alpha, float* dalpha, float* dsd, float* czf, float* wout,
                 float* dxh, int64_t ldd, int64_t num_nodes, int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(num_nodes > 0 && dim > 0 && dim % 4 == 0 && dim <= 1280 && ldx % 4 == 0 && ldg % 4 == 0 && ldd % 4 == 0,
               "gat_bwd: bad shape");
  const bool chem = emb1 != nullptr;
  PGNN_REQUIRE(chem ? (emb2 && in_code && !slot_feat) : (slot_feat && self_feat && tenc && kf > 0 && kf <= kMaxKF),
               "gat_bwd: pass either the chem tables or the per-slot features");
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)num_nodes, d = (int)dim;
  const size_t lds = (size_t)(chem ? kCodes : kf) * kHeads * d * sizeof(float);
  allow_big_lds((const void*)k_gat_edge_dot, lds);
  hipLaunchKernelGGL(k_gat_edge_dot, dim3(grid_rows(n, kBlock / 64)), dim3(kBlock), lds, st, g, ldg, xh, ldx, in_ptr, in_src, in_code,
                     emb1, emb2, slot_feat, self_feat, tenc, (int)kf, dalpha, n, d);
  const int tgrid = (int)ceil_div((int64_t)n * kHeads, kBlock);
  hipLaunchKernelGGL(k_gat_alpha_bwd, dim3(tgrid), dim3(kBlock), 0, st, alpha, z, dalpha, in_ptr, in_code, slot_feat, self_feat, (int)kf,
                     negative_slope, dsd, czf, n);
  hipLaunchKernelGGL(k_gat_src_gather, dim3(tgrid), dim3(kBlock), 0, st, in_ptr, in_src, out_ptr, out_dst, alpha, dalpha, dsd, wout, n);
  const int gs = d / 4, groups = std::max(1, 320 / gs);
  PGNN_REQUIRE(gs <= 320, "gat_bwd: dim too wide");
  hipLaunchKernelGGL(k_gat_aggregate_bwd, dim3(grid_rows(n, groups)), dim3(320), 0, st, g, ldg, in_ptr, out_ptr, out_dst, wout, alpha,
                     dsd, att, dxh, ldd, n, d, groups);
  return check_launch("gat_bwd");
}

int pgnn_segment_softmax_fwd(const float* z, const int32_t* ptr, const int32_t* perm, float* alpha, int64_t num_segments,
                             int64_t heads, pgnn_stream stream) {
  PGNN_REQUIRE(num_segments > 0 && heads > 0, "segment_softmax_fwd: bad shape");
  hipLaunchKernelGGL(k_segment_softmax_fwd, dim3((int)ceil_div(num_segments * heads, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, z,
                     ptr, perm, alpha, (int)num_segments, (int)heads);
  return check_launch("segment_softmax_fwd");
}

int pgnn_segment_softmax_bwd(const float* alpha, const float* dalpha, const int32_t* ptr, const int32_t* perm, float* dz,
                             int64_t num_segments, int64_t heads, pgnn_stream stream) {
  PGNN_REQUIRE(num_segments > 0 && heads > 0, "segment_softmax_bwd: bad shape");
  hipLaunchKernelGGL(k_segment_softmax_bwd, dim3((int)ceil_div(num_segments * heads, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     alpha, dalpha, ptr, perm, dz, (int)num_segments, (int)heads);
  return check_launch("segment_softmax_bwd");
}

int pgnn_segment_max_fwd(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* perm, float* out, int64_t ldo,
                         int32_t* arg, int64_t num_segments, int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(num_segments > 0 && dim > 0 && dim % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "segment_max_fwd: bad shape");
  hipLaunchKernelGGL(k_segment_max_fwd, dim3((int)ceil_div(num_segments * (dim / 4), kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x,
                     ldx, ptr, perm, out, ldo, arg, (int)num_segments, (int)(dim / 4));
  return check_launch("segment_max_fwd");
}

int pgnn_segment_max_bwd(const float* g, int64_t ldg, const int64_t* key, const int32_t* arg, float* dx, int64_t ldd,
                         int64_t num_segments, int64_t num_items, int64_t dim, pgnn_stream stream) {
  PGNN_REQUIRE(num_segments > 0 && num_items > 0 && dim % 4 == 0 && ldg % 4 == 0 && ldd % 4 == 0, "segment_max_bwd: bad shape");
  hipLaunchKernelGGL(k_segment_max_bwd, dim3((int)ceil_div(num_items * (dim / 4), kBlock)), dim3(kBlock), 0, (hipStream_t)stream, g, ldg,
                     key, arg, dx, ldd, (int)num_segments, (int)num_items, (int)(dim / 4));
  return check_launch("segment_max_bwd");
}

}  // extern "C"
