/* pgnn.h -- C ABI of the MI355X (gfx950) message-passing hot path of pretrain-gnns.
 *
 * The reference (snap-stanford/pretrain-gnns) has no FFI layer: its hot path is Python calling
 * third-party wheels (torch_geometric 1.0.3 / torch_scatter 1.1.2 / torch 1.0.1).  This header is
 * the boundary a maintainer would bind instead; every entry point cites the reference lines whose
 * arithmetic it replaces (paths relative to the reference repo).  The ctypes binding that
 * `pretrain_gnns_amd/_lib.py` uses is the "reference-side stub" shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless marked host; the library never allocates: callers
 *    pass outputs and workspaces (sizes from the *_workspace_bytes functions);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - node-feature matrices are fp32 row-major with an explicit leading dimension (ld, in floats);
 *    feature width D must be a multiple of 4 (rows are read as float4);
 *  - integer graph inputs are int64 exactly as PyG stores them (edge_index [2,E], edge_attr [E,2]);
 *    the library's own structures are int32 / uint8;
 *  - return 0 on success, otherwise a PGNN_ERR_* code; pgnn_last_error() gives a message.
 *  - direction convention (torch_geometric 1.0.3 MessagePassing.propagate): messages flow from
 *    edge_index[1] ("j", source) to edge_index[0] ("i", destination / aggregation index).
 */
#ifndef PGNN_H
#define PGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGNN_ABI_VERSION 11
/* uint32 words behind every `counter` argument below: the arrival tickets of a launch whose last block folds the others'
 * results (one top word + up to 32 group words: same-address atomics retire at ~50 ns each, see csrc/common.h).  Zero before
 * the first call, left zero by every call; one buffer per device serves all calls of a stream. */
#define PGNN_TICKET_WORDS 40

#define PGNN_OK 0
#define PGNN_ERR_ARG 1
#define PGNN_ERR_HIP 2
#define PGNN_ERR_WORKSPACE 3

typedef void* pgnn_stream; /* hipStream_t */

int pgnn_abi_version(void);
const char* pgnn_last_error(void); /* host string, thread-local */
/* The PGNN_* environment knobs (A/B switches) are read once per call site and cached;
 * call this after changing one inside a running process. */
void pgnn_reload_env(void);
/* ------------------------------------------------------------------------------------------
 * Graph structure.  Replaces the per-layer add_self_loops + torch.cat of chem/model.py:39-45,
 * 84-93 and bio/model.py:39-45,94-100 and the COO gather/scatter inside propagate: built ONCE
 * per batch and shared by all layers and by the backward pass.
 *
 *  in_ptr[N+1], in_src[E]  : CSR by destination (edge_index[0]); inside a row the edges keep
 *                            their original order (stable), so sums run in the reference's order
 *  out_ptr[N+1], out_dst[E]: CSR by source (edge_index[1]) -- the transpose, for the backward
 *  in_code[E]              : chem only, bond code a0*3+a1 (a0 = bond type <6, a1 = direction <3)
 *  dinv[N]                 : (in_degree+1)^-1/2, the GCN normaliser of chem/model.py:73-82
 *  cfeat[N,KC]             : per-destination edge-feature sums (self loop included), see below
 *  status[1]               : incremented once per out-of-range index / attribute found
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_graph_workspace_bytes(int64_t num_nodes, int64_t num_edges);

/* chem: cfeat[N,9]; columns 0..5 count in-edges per bond type, 6..8 per bond direction, self loop
 * counted as type 4 / direction 0 (chem/model.py:42-45).  gcn!=0: each count is weighted by the
 * edge's symmetric normaliser dinv[i]*dinv[src] (chem/model.py:82).  Used for the bond-embedding
 * gradients: dE[t] = sum_i cfeat[i,t] * dAgg[i]. */
int pgnn_chem_graph_build(const int64_t* edge_index, const int64_t* edge_attr, int64_t num_edges,
                          int64_t num_nodes, int gcn, int32_t* in_ptr, int32_t* in_src,
                          uint8_t* in_code, int32_t* out_ptr, int32_t* out_dst, float* dinv,
                          float* cfeat, int32_t* status, void* ws, size_t ws_bytes,
                          pgnn_stream stream);

/* bio: edge_attr is fp32 [E,9]; cfeat[N,10] = [sum of in-edge attr rows + self-loop row e_7,
 * in_degree+1] (bio/model.py:42-47), so that sum_e edge_encoder(attr_e) = cfeat[i,:] . [W^T; b].
 * gcn!=0: rows weighted by dinv[i]*dinv[src] (bio/model.py:79-92). */
int pgnn_bio_graph_build(const int64_t* edge_index, const float* edge_attr, int64_t num_edges,
                         int64_t num_nodes, int gcn, int32_t* in_ptr, int32_t* in_src,
                         int32_t* out_ptr, int32_t* out_dst, float* dinv, float* cfeat,
                         int32_t* status, void* ws, size_t ws_bytes, pgnn_stream stream);

/* Stable grouping of n_items by key in [0,n_keys): ptr[n_keys+1], perm[n_items] (item ids ordered
 * by (key, id)).  Keys are read with a stride (in int64 elements) so a column of x[N,2] can be
 * grouped in place.  n_keys <= 1024: one stable radix pass (any segment length, O(n)); more keys:
 * CSR fill + in-segment ranking (meant for short segments such as node neighbourhoods).  Used for pooling (key = batch vector, chem/model.py:369) and for the input
 * embedding gradient (key = atom type, chem/model.py:264). */
size_t pgnn_group_workspace_bytes(int64_t n_keys, int64_t n_items);
int pgnn_group_by_key(const int64_t* key, int64_t key_stride, int64_t n_items, int64_t n_keys,
                      int32_t* ptr, int32_t* perm, int32_t* status, void* ws, size_t ws_bytes,
                      pgnn_stream stream);

/* The same for a PAIR of key columns in one pass: items grouped by (key_a, key_b), segment index
 * key_a*n_b + key_b, n_a*n_b <= 1024.  With pgnn_segment_sum and pgnn_pair_fold this yields the gradients of
 * BOTH atom embedding tables (chem/model.py:264) from one grouping and one pass over the node gradients. */
int pgnn_group_by_key_pair(const int64_t* key_a, const int64_t* key_b, int64_t key_stride, int64_t n_items,
                           int64_t n_a, int64_t n_b, int32_t* ptr, int32_t* perm, int32_t* status,
                           void* ws, size_t ws_bytes, pgnn_stream stream);
/* sums [n_a*n_b, dim] (row a*n_b + b) -> out_a[a] = sum_b, out_b[b] = sum_a, ascending order; either may be NULL */
int pgnn_pair_fold(const float* sums, int64_t n_a, int64_t n_b, float* out_a, int64_t lda, float* out_b,
                   int64_t ldb, int64_t dim, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Aggregation (the scatter_add of propagate, chem/model.py:49-52,101-104; bio/model.py:52-55).
 * ------------------------------------------------------------------------------------------ */

/* chem GIN (dinv == NULL) / chem GCN (dinv != NULL) forward:
 *   out[i] = sum_{e in in(i)} w_e * (x[src_e] + (emb1[a0_e] + emb2[a1_e])) + w_ii * (x[i] + (emb1[4] + emb2[0]))
 * with w = 1 (GIN) or dinv[i]*dinv[src] (GCN); additions in the reference's order (edges in
 * original order, self loop last), no fused multiply-add. */
int pgnn_chem_aggregate_fwd(const float* x, int64_t ldx, const int32_t* in_ptr,
                            const int32_t* in_src, const uint8_t* in_code, const float* emb1,
                            const float* emb2, const float* dinv, float* out, int64_t ldo,
                            int64_t num_nodes, int64_t dim, pgnn_stream stream);

/* chem GIN aggregation of h = relu?(coef[0]*z + coef[1]) computed on read: the BatchNorm(+ReLU) between
 * two GIN layers (chem/model.py:269-273) folded into the next layer's gather, so h is never written to
 * memory.  Bit-identical to pgnn_bn_fwd followed by pgnn_chem_aggregate_fwd.  dim <= 320. */
int pgnn_chem_aggregate_bn_fwd(const float* z, int64_t ldz, const float* coef /*[2,dim]*/, int relu,
                               const int32_t* in_ptr, const int32_t* in_src, const uint8_t* in_code,
                               const float* emb1, const float* emb2, float* out, int64_t ldo,
                               int64_t num_nodes, int64_t dim, pgnn_stream stream);

/* Plain neighbour sum over any CSR (ptr, nbr):  out[i] = sum_p w * x[nbr_p] + w_ii * x[i].
 * Serves: backward of every aggregation w.r.t. x (with the transposed CSR: chem/model.py:49-52
 * under autograd), and the x-half of the bio GIN message concat (bio/model.py:54-55). */
int pgnn_neighbor_sum(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr,
                      const float* dinv, float* out, int64_t ldo, int64_t num_nodes, int64_t dim,
                      pgnn_stream stream);

/* out[i,:] (+)= cfeat[i,0:kc] . table[0:kc,:]   (kc <= 16).  bio edge-encoder term of the
 * aggregation (bio/model.py:47,55) with table = [W^T; b]; accumulate!=0 adds to out. */
int pgnn_rowfeat_matmul_fwd(const float* cfeat, int64_t kc, const float* table, int64_t ldt,
                            float* out, int64_t ldo, int64_t num_nodes, int64_t dim,
                            int accumulate, pgnn_stream stream);

/* gtable[t,:] = sum_i cfeat[i,t] * g[i,:]   (deterministic two-pass reduction).  Gradient of the
 * bond embeddings (chem) / of the edge encoder (bio). */
size_t pgnn_rowfeat_matmul_bwd_workspace_bytes(int64_t num_nodes, int64_t kc, int64_t dim);
int pgnn_rowfeat_matmul_bwd(const float* cfeat, int64_t kc, const float* g, int64_t ldg,
                            float* gtable, int64_t ldgt, int64_t num_nodes, int64_t dim, void* ws,
                            size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Input embedding (chem/model.py:264; bio/model.py:49-50) and segment reductions
 * (global_mean_pool / global_add_pool, chem/model.py:324-326,369; chem/pretrain_contextpred.py:28-34).
 * ------------------------------------------------------------------------------------------ */

/* out[i] = table1[idx[i*idx_stride]] + (table2 ? table2[idx[i*idx_stride+1]] : 0) */
int pgnn_embed_fwd(const int64_t* idx, int64_t idx_stride, const float* table1, int64_t rows1,
                   const float* table2, int64_t rows2, float* out, int64_t ldo, int64_t num_nodes,
                   int64_t dim, int32_t* status, pgnn_stream stream);

/* out[s] = scale_s * sum_{p in [ptr[s],ptr[s+1])} x[perm ? perm[p] : p]; mean!=0: scale = 1/max(len,1).
 * Any segment length (two-level, deterministic).  With (ptr, perm) from pgnn_group_by_key this is
 * both the pooling forward and the embedding-table gradient. */
size_t pgnn_segment_sum_workspace_bytes(int64_t n_items, int64_t n_segments, int64_t dim);
int pgnn_segment_sum(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* perm,
                     int64_t n_items, int64_t n_segments, int mean, float* out, int64_t ldo,
                     int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream);

/* gx[i] = scale * g[key[i]]  (backward of the pooling; mean!=0 divides by the segment length) */
int pgnn_segment_broadcast(const float* g, int64_t ldg, const int64_t* key, const int32_t* ptr,
                           int mean, float* gx, int64_t ldgx, int64_t n_items, int64_t dim,
                           pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d (+ fused ReLU): chem/model.py:252,269-275 (outer BN), bio/model.py:24 (BN in the mlp).
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_bn_workspace_bytes(int64_t num_rows, int64_t dim);

/* training != 0: batch statistics (biased variance for normalisation, unbiased for the running
 * estimate, momentum as torch), running stats updated in place, save_mean/save_invstd written.
 * training == 0: running statistics. relu != 0 fuses max(.,0). */
int pgnn_bn_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float momentum, float eps, int training,
                int relu, float* y, int64_t ldy, float* save_mean, float* save_invstd, float drop_p,
                uint64_t drop_seed, int64_t num_rows, int64_t dim, void* ws, size_t ws_bytes,
                pgnn_stream stream);
/* drop_p > 0 fuses the F.dropout that follows (chem/model.py:271-275): inverted dropout with keep
 * bits from a counter-based generator over (drop_seed, element) -- 16 bits per element, p resolved to
 * 1/65536 -- so the backward regenerates the mask from the same seed instead of reading it. */

/* Statistics only: what pgnn_bn_fwd does before its normalise pass.  coef [2, dim] receives the
 * affine form of the layer, y = coef[0]*x + coef[1], for a consumer that applies it on read
 * (pgnn_chem_aggregate_bn_fwd). */
int pgnn_bn_stats_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, int training,
                      float* save_mean, float* save_invstd, float* coef, int64_t num_rows, int64_t dim,
                      void* ws, size_t ws_bytes, pgnn_stream stream);

/* The same statistics from what the product in front of the BatchNorm already had in registers (pgnn_linear_fwd_colstats):
 * blocks [ceil(num_rows / 16)][2][dim] = per 16-row block and column, the column sum and the sum of squared deviations from
 * the block's own column mean; merged pairwise in float64 with the parallel-variance formula (fixed order).  Training mode
 * only.  pgnn_bn_apply_fwd is the normalise pass on its own: y = coef[0]*x + coef[1] (+ReLU, + fused dropout).
 * Together they are pgnn_bn_fwd minus its read of x for the statistics (chem/model.py:269: batch_norms[layer](h)). */
int pgnn_bn_stats_fwd_blocks(const float* blocks, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* save_mean, float* save_invstd, float* coef, int64_t num_rows,
                             int64_t dim, pgnn_stream stream);
int pgnn_bn_apply_fwd(const float* x, int64_t ldx, const float* coef, int relu, float* y, int64_t ldy, float drop_p,
                      uint64_t drop_seed, int64_t num_rows, int64_t dim, pgnn_stream stream);

/* Backward of the above (ReLU and dropout masks are recomputed, nothing else is kept). */
int pgnn_bn_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                const float* beta, const float* save_mean, const float* save_invstd, int training,
                int relu, float* dx, int64_t lddx, float* dgamma, float* dbeta, float drop_p,
                uint64_t drop_seed, int64_t num_rows, int64_t dim, void* ws, size_t ws_bytes,
                pgnn_stream stream);

/* pgnn_neighbor_sum (unweighted, transposed CSR = the backward of the aggregation, chem/model.py:49-52 under autograd) whose launch
 * ALSO leaves the column sums of the BatchNorm backward of the layer below (chem/model.py:269-273: out = dL/dy, y = relu?(BN(z))),
 * folded: dgamma / dbeta final, and *coef_out -> [7, dim] floats inside ws from which the elementwise pass of pgnn_bn_bwd forms dz.
 * ws: pgnn_bn_workspace_bytes(n, dim).  *fused = 0: outside the tuned instance (feature width 300): plain sum, nothing else written. */
int pgnn_neighbor_sum_bn_bwd(const float* x, int64_t ldx, const int32_t* out_ptr, const int32_t* out_dst, float* out, int64_t ldo,
                             const float* z, int64_t ldz, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, int relu, int training, float* dgamma, float* dbeta, int64_t num_nodes,
                             int64_t dim, void* ws, size_t ws_bytes, const float** coef_out, int* fused, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * GraphSAGE update (chem/model.py:165-202, bio/model.py:183-224): aggr="mean" + F.normalize(p=2).
 * `sum` is the unweighted aggregation incl. the self loop (pgnn_chem_aggregate_fwd with dinv == NULL, or
 * pgnn_neighbor_sum + pgnn_rowfeat_matmul_fwd for bio); in_ptr is the CSR-by-destination row pointer.
 *   v = sum / (in_ptr[i+1] - in_ptr[i] + 1) ;  norm = ||v||_2 ;  y = v / max(norm, 1e-12)
 * ------------------------------------------------------------------------------------------ */
int pgnn_mean_l2norm_fwd(const float* sum, int64_t ld_sum, const int32_t* in_ptr, float* y, int64_t ldy,
                         float* norm /*[n]*/, int64_t num_nodes, int64_t dim, pgnn_stream stream);
/* dsum = gradient w.r.t. `sum` */
int pgnn_mean_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* norm,
                         const int32_t* in_ptr, float* dsum, int64_t ld_dsum, int64_t num_nodes, int64_t dim,
                         pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Context prediction: the negative-sampling dot-product loss (chem/pretrain_contextpred.py:54-67,86-97 and the same lines of
 * bio/pretrain_contextpred.py; cbow mode, mean context pooling -- the reference's defaults) in two launches forward, one backward.
 *   hs [n_sub, dim]: substructure node embeddings, center [graphs]: row of every graph's centre atom
 *   hc [n_ctx, dim]: context node embeddings, overlap [n_overlap]: rows of the overlap nodes, seg [n_overlap]: their graph ids,
 *                    ascending (batch_overlapped_context)
 *   out [4] float64 = (loss_pos, loss_neg, fraction of pred_pos > 0, fraction of pred_neg < 0); loss (may be NULL) float64 =
 *   loss_pos + neg_samples loss_neg, the quantity train() back-propagates (:89); accum (may be NULL) [4] float64:
 *   accum[0] += loss_pos + loss_neg, accum[1] += 0.5 (fraction + fraction), accum[3] += 1 (the epoch sums of train(), :99-100).
 *   counter: PGNN_TICKET_WORDS zeroed uint32 that the call leaves zeroed; status: incremented per out-of-range index (clamped).
 * The workspace carries the pooled context rows and the scores from forward to backward (same buffer, untouched in between).
 * backward: grad_loss [1] float64 = d / d loss; dhs [n_sub, lddhs], dhc [n_ctx, lddhc] are written in full
 * (zero except the centre rows / the overlap rows; neither index vector may repeat a row).
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_contextpred_loss_workspace_bytes(int64_t graphs, int64_t dim, int64_t neg_samples);
int pgnn_contextpred_loss_fwd(const float* hs, int64_t ldhs, int64_t n_sub, const int64_t* center, const float* hc, int64_t ldhc, int64_t n_ctx,
                              const int64_t* overlap, const int64_t* seg, int64_t n_overlap, int64_t graphs, int64_t dim, int64_t neg_samples,
                              double* out, double* loss, double* accum, int32_t* status, uint32_t* counter, void* ws, size_t ws_bytes,
                              pgnn_stream stream);
int pgnn_contextpred_loss_bwd(const float* hs, int64_t ldhs, int64_t n_sub, const int64_t* center, int64_t n_ctx, const int64_t* overlap,
                              const int64_t* seg, int64_t n_overlap, int64_t graphs, int64_t dim, int64_t neg_samples, const double* grad_loss,
                              float* dhs, int64_t lddhs, float* dhc, int64_t lddhc, const void* ws, size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Linear layers of the GIN mlp / GCN linear (chem/model.py:29,54-55,63,99; bio/model.py:24,67,109):
 * fp32 GEMMs on the matrix cores.  Forward: three-term bf16 split of every fp32 value, six v_mfma_f32_16x16x32_bf16
 * products per k-step, fp32 accumulate -- the rounding error of an fp32 FMA chain (csrc/linear.hip; PGNN_GEMM_SPLIT=0
 * selects the v_mfma_f32_16x16x4_f32 kernel instead).  Backward products: v_mfma_f32_16x16x4_f32, exact fp32 FMA chains.
 * ------------------------------------------------------------------------------------------ */

/* y[M,N] = act(x[M,K] . W[N,K]^T + b) ; relu != 0 -> act = max(.,0) */
int pgnn_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                    int64_t ldy, int64_t m, int64_t k, int64_t n, int relu, pgnn_stream stream);

/* The same product, also leaving the column statistics of y that a BatchNorm behind it needs (the mlp's second Linear in front
 * of batch_norms[layer], chem/model.py:54-55,269): colstat [ceil(m / 16)][2][n]; for the 16-row block t and column c,
 * colstat[t][0][c] = sum_r y[r, c] and colstat[t][1][c] = sum_r (y[r, c] - mean_t,c)^2 over the block's rows, taken in the
 * epilogue from the accumulators (pgnn_bn_stats_fwd_blocks consumes them). */
int pgnn_linear_fwd_colstats(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t m,
                             int64_t k, int64_t n, int relu, float* colstat, pgnn_stream stream);

/* dx[M,K] = dy[M,N] . W[N,K] ; if relu_out != NULL: dx *= (relu_out > 0)  (relu_out = the
 * activation this dx flows into, i.e. the forward output of the preceding Linear+ReLU) */
int pgnn_linear_bwd_data(const float* dy, int64_t lddy, const float* w, const float* relu_out,
                         int64_t ldr, float* dx, int64_t lddx, int64_t m, int64_t k, int64_t n,
                         pgnn_stream stream);

/* The same product from pre-transposed weights wt[K,N] = W^T: both operands are then contiguous along the contracted
 * dimension and the product runs the forward kernel (the weights of a layer stack are transposed once per backward pass
 * with pgnn_transpose_batch, off the critical path). */
int pgnn_linear_bwd_data_t(const float* dy, int64_t lddy, const float* wt, const float* relu_out, int64_t ldr, float* dx,
                           int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream);
/* dst[j][c][r] = src[j][r][c] for count <= 16 dense row-major matrices (host arrays of device pointers / sizes), one launch */
int pgnn_transpose_batch(const float* const* src, float* const* dst, const int64_t* rows, const int64_t* cols, int64_t count,
                         pgnn_stream stream);

/* Products on PRE-SPLIT weights (round 3; nn.Linear of chem/model.py:29,54-55, forward and backward-data).  The weights of a
 * layer stack are constant within a pass, so their three-term bf16 split (and, for the backward, their transposition) is done
 * once per pass by pgnn_split_weights instead of by every workgroup in every k-step: planes [3][rows][ld] of bf16, ld = cols
 * rounded up to 32, zero beyond cols (pgnn_weight_planes_bytes bytes per matrix).  transpose[j] != 0: the planes of src[j]^T.
 * count <= 32 matrices per launch (host arrays of device pointers / sizes). */
size_t pgnn_weight_planes_bytes(int64_t rows, int64_t cols);
/* 1 if a layer stack should run its [m, k] x [n, k]^T products on planes (same bits as pgnn_linear_fwd there, faster) */
int pgnn_linear_wp_preferred(int64_t m, int64_t k, int64_t n);
int pgnn_split_weights(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                       int64_t count, pgnn_stream stream);
/* pgnn_linear_fwd / pgnn_linear_fwd_colstats (colstat may be NULL) with wplanes = the planes of W [n, k]; bit-identical to them
 * wherever they run the split-bf16 kernel: the activations reach LDS by DMA as fp32 and are split by the consuming wave */
int pgnn_linear_fwd_wp(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m,
                       int64_t k, int64_t n, int relu, float* colstat, pgnn_stream stream);
/* pgnn_linear_bwd_data_t with wtplanes = the planes of W^T [k, n] */
int pgnn_linear_bwd_data_wp(const float* dy, int64_t lddy, const void* wtplanes, const float* relu_out, int64_t ldr, float* dx,
                            int64_t lddx, int64_t m, int64_t k, int64_t n, pgnn_stream stream);

/* The same products on TWO fp16 planes per operand under a power-of-two scale per ROW (round 4; the default of the one-call
 * networks): x = (h1 + h2) / s with s x's largest magnitude of the row in [2^13, 2^14) -- 11 + 11 significant bits, the low plane
 * clear of fp16's subnormals for every element that matters -- needs three v_mfma_f32_16x16x32_f16 per accumulator (h1 h2, h2 h1,
 * h1 h1) where three bf16 planes need six; scales, and the epilogue's rescale, are exact.  Error against float64 as for the
 * fp32-MFMA kernel (tests/test_gpu_ops.py); +-inf / NaN inputs give non-finite outputs in the rows fp32 gives them.
 * pgnn_split_weights_2p: planes [2][rows][ld] fp16 of s W, then 1 / s per row as fp32, inside pgnn_weight_planes_bytes.
 * x_amax / dy_amax: [m] uint32 = bit patterns of max |row| of the activation operand if its producer left them (a previous
 * product's y_amax / dx_amax), NULL = every workgroup takes the maxima of its own rows in a pass in front of its k-loop.
 * y_amax / dx_amax: NULL, or [m] words that are ZERO before the call and receive the bit patterns of the result rows' largest
 * magnitudes (atomic maximum over the column tiles; the maximum of a row's non-NaN entries -- a NaN entry does not change the row's
 * scale, it propagates through the planes themselves). */
int pgnn_split_weights_2p(const float* const* src, void* const* dst, const int64_t* rows, const int64_t* cols, const int32_t* transpose,
                          int64_t count, pgnn_stream stream);
int pgnn_linear_fwd_2p(const float* x, int64_t ldx, const uint32_t* x_amax, const void* wplanes2, const float* bias, float* y, int64_t ldy,
                       int64_t m, int64_t k, int64_t n, int relu, float* colstat, uint32_t* y_amax, pgnn_stream stream);
int pgnn_linear_bwd_data_2p(const float* dy, int64_t lddy, const uint32_t* dy_amax, const void* wtplanes2, const float* relu_out, int64_t ldr,
                            float* dx, int64_t lddx, int64_t m, int64_t k, int64_t n, uint32_t* dx_amax, pgnn_stream stream);

/* The GIN mlp (chem/model.py:29,54-55: Linear(D, 2D) -> ReLU -> Linear(2D, D)) as ONE launch per direction on two-plane weights
 * (ABI 10, csrc/mlp_fused.hip): a wave owns 16 rows for both products, the [m, n1] hidden activation is produced 32 columns at a
 * time in registers, WRITTEN ONCE (the backward needs it) and consumed on chip as a k-step of the second product -- it is never
 * re-read, and the activations are fetched once instead of once per column workgroup.  The planes of the hidden rows take a
 * running power-of-two scale (lowered, with an exact rescale of the accumulators, when a 32-column chunk would leave fp16's range).
 *   forward:        hid = relu(x . W1^T + b1) [m, n1],  y = hid . W2^T + b2 [m, n2]; wplanes1 / wplanes2 = pgnn_split_weights_2p of
 *                   W1 [n1, k1] / W2 [n2, n1]; colstat as in pgnn_linear_fwd_2p (per-16-row-block column statistics of y) or NULL
 *   backward-data:  dhid = (dy . W2) * (relu_out > 0) [m, n1],  dx = dhid . W1 [m, n2]; w2tplanes / w1tplanes = the planes of
 *                   W2^T [n1, k1] / W1^T [n2, n1] (transpose[j] = 1); here k1 = columns of dy (the mlp's output width)
 * Shapes covered (pgnn_mlp_2p_fused_supported): k1 in (288, 320], n2 in (288, 304], n1 in [32, 608], all multiples of 4 -- the
 * emb_dim = 300 mlp; everything else takes the two products above.  Error against float64 as for them (tests/test_gpu_ops.py). */
int pgnn_mlp_2p_fused_supported(int64_t m, int64_t k1, int64_t n1, int64_t n2);
int pgnn_mlp_fwd_2p_fused(const float* x, int64_t ldx, const void* wplanes1, const float* b1, const void* wplanes2, const float* b2, float* hid,
                          int64_t ldh, float* y, int64_t ldy, int64_t m, int64_t k1, int64_t n1, int64_t n2, float* colstat, pgnn_stream stream);
int pgnn_mlp_bwd_data_2p_fused(const float* dy, int64_t lddy, const void* w2tplanes, const float* relu_out, int64_t ldr, const void* w1tplanes,
                               float* dhid, int64_t lddh, float* dx, int64_t lddx, int64_t m, int64_t k1, int64_t n1, int64_t n2,
                               pgnn_stream stream);

/* dW[N,K] = dy[M,N]^T . x[M,K] ; db[N] = column sums of dy (db may be NULL).  Split over M with a
 * deterministic second-pass reduction. */
size_t pgnn_linear_bwd_weight_workspace_bytes(int64_t m, int64_t k, int64_t n);
int pgnn_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dw,
                           float* db, int64_t m, int64_t k, int64_t n, void* ws, size_t ws_bytes,
                           pgnn_stream stream);

/* Two weight gradients over the same m rows -- the two Linears of one GIN mlp (chem/model.py:29) -- with ONE fold of their
 * split-K partials; results identical to two pgnn_linear_bwd_weight calls.  Workspace: the sum of the two single-call sizes. */
int pgnn_linear_bwd_weight_pair(const float* dy_a, int64_t lddy_a, const float* x_a, int64_t ldx_a, float* dw_a, float* db_a, int64_t k_a,
                                int64_t n_a, const float* dy_b, int64_t lddy_b, const float* x_b, int64_t ldx_b, float* dw_b, float* db_b,
                                int64_t k_b, int64_t n_b, int64_t m, void* ws, size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * torch.optim.Adam's update (chem/pretrain_masking.py:134-136 builds three of them with the same hyper-parameters) over
 * n <= pgnn_adam_max_tensors() fp32 tensors in one launch: params[j] / grads[j] device pointers (host arrays), counts[j]
 * elements, state_offsets[j] = where tensor j's moments live in the flat exp_avg / exp_avg_sq buffers.  step = device
 * int64[32], zero before the first call: step[0], the number of updates already applied, is advanced by the call (so the call can
 * be captured in a HIP graph); words 2..6 are the call's own cache of beta1^step, beta2^step (float64), the betas and the step
 * they belong to -- a caller may overwrite step[0] (a restored checkpoint) without touching them, the call then recomputes the
 * powers; words 8.. hold the kernel's PGNN_TICKET_WORDS arrival tickets, zero on entry and left zero.
 * L2 weight decay is added to the gradient (Adam, not AdamW); amsgrad is not offered.
 * ------------------------------------------------------------------------------------------ */
int pgnn_adam_max_tensors(void);
int pgnn_adam_step(float* const* params, const float* const* grads, const int64_t* counts, const int64_t* state_offsets, int64_t n,
                   float* exp_avg, float* exp_avg_sq, int64_t* step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Prediction head of the masking pre-training step, fused (chem/pretrain_masking.py:52-57; bio/pretrain_masking.py:45-56):
 *   pred = linear_pred(node_rep[idx]) ; loss = CrossEntropyLoss()(pred.double(), label) ; correct = #(argmax(pred) == label)
 * h [n_rows, ldh] node representations, idx [m] int64 rows to predict (must not repeat: MaskAtom samples without
 * replacement), w [classes, dim], b [classes] or NULL, label[r * label_stride] int64 in [0, classes), classes <= 128.
 * Outputs: logits [m, classes] fp32 (kept for the backward), *loss float64 (mean over rows), *correct int64, optionally
 * metrics[2] = {loss, (double)correct} for a single read-back, and optionally accum[4] float64 = the running epoch sums the
 * reference's train() keeps on the host (chem/pretrain_masking.py:72-76): accum[0] += loss, accum[1] += correct / m,
 * accum[3] += 1 (accum[2], the bond-accuracy sum, is the caller's), so that a train loop reads back once per epoch;
 * status += bad indices / labels.  counter: PGNN_TICKET_WORDS uint32 that are
 * zero on entry and are left zero (the arrival tickets of the in-kernel final fold; a persistent per-device buffer).  fp32 linear algebra, float64 soft-max and loss, as in the reference; fixed summation order.
 * Backward (gloss = d objective / d loss, float64 on the device): dnode [n_rows, ldd] is overwritten (zero outside idx),
 * dw [classes, dim], db [classes] or NULL.  Both calls use the same workspace.
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_masked_head_workspace_bytes(int64_t m, int64_t classes, int64_t dim);
int pgnn_masked_head_fwd(const float* h, int64_t ldh, int64_t n_rows, const int64_t* idx, int64_t m, const float* w, const float* b,
                         const int64_t* label, int64_t label_stride, int64_t classes, int64_t dim, float* logits, double* loss,
                         int64_t* correct, double* metrics, double* accum, int32_t* status, uint32_t* counter, void* ws,
                         size_t ws_bytes, pgnn_stream stream);
int pgnn_masked_head_bwd(const float* h, int64_t ldh, int64_t n_rows, const int64_t* idx, int64_t m, const float* w,
                         const int64_t* label, int64_t label_stride, const float* logits, const double* gloss, int64_t classes,
                         int64_t dim, float* dnode, int64_t ldd, float* dw, float* db, void* ws, size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Edge-prediction head of the masking pre-training steps (bio/pretrain_masking.py:45-58; chem/pretrain_masking.py:60-66):
 *   edge_rep = node_rep[u] + node_rep[v] ; pred = linear(edge_rep) ; loss = CrossEntropyLoss()(pred, label) ;
 *   correct = #(argmax(pred, 1) == label)
 * h [n_nodes, ldh]; ends [2, m] int64 = the masked edges' end points (row 0: u, row 1: v -- edge_index[:, masked_idx],
 * contiguous; a node may appear in any number of edges); w [classes, dim], b [classes] or NULL, classes 4 (chem bond types)
 * or 7 (bio edge types).  Labels: EITHER label[r * label_stride] int64 OR onehot [m, onehot_cols] fp32 rows (pitch ld_onehot) whose
 * first maximum is the label (the bio script's torch.argmax(mask_edge_label, 1) over the 9 edge-attribute columns; a maximum at
 * a column >= classes counts in status, as a label out of range).  loss_float64 != 0: soft-max and loss in float64 (chem's
 * pred.double()); 0: in fp32 (bio) -- *loss64 then holds the fp32 value widened, loss32 (optional) the fp32 value.
 * logits [m, classes] is kept for the backward.  accum (optional float64 [4]): accum[0] += loss, accum[accum_slot] += correct / m
 * (slot 1 or 2), accum[3] += 1 iff accum_step; metrics / status / counter as in pgnn_masked_head_fwd.
 * The head is evaluated as P = h . w^T [n_nodes, classes], pred[r] = P[u] + P[v] + b, so nothing of size [m, dim] exists.
 * Backward: exactly one of gloss64 / gloss32 (device scalar, d objective / d loss).  dnode [n_nodes, ldd] OVERWRITTEN with
 * d objective / d h (S . w with S[n] = the sum of d pred over the masked edges at n, grouped in a fixed order), dw
 * [classes, dim] = S^T . h, db [classes] or NULL.  Both calls use the same workspace and the same zeroed counter words.
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_edge_head_workspace_bytes(int64_t n_nodes, int64_t m, int64_t classes, int64_t dim);
int pgnn_edge_head_fwd(const float* h, int64_t ldh, int64_t n_nodes, const int64_t* ends, int64_t m, const float* w, const float* b,
                       const int64_t* label, int64_t label_stride, const float* onehot, int64_t ld_onehot, int64_t onehot_cols,
                       int64_t classes, int64_t dim, int loss_float64, float* logits, double* loss64, float* loss32, int64_t* correct, double* metrics,
                       double* accum, int accum_slot, int accum_step, int32_t* status, uint32_t* counter, void* ws, size_t ws_bytes,
                       pgnn_stream stream);
int pgnn_edge_head_bwd(const float* h, int64_t ldh, int64_t n_nodes, const int64_t* ends, int64_t m, const float* w,
                       const int64_t* label, int64_t label_stride, const float* onehot, int64_t ld_onehot, int64_t onehot_cols, const float* logits,
                       const double* gloss64, const float* gloss32, int64_t classes, int64_t dim, int loss_float64, float* dnode,
                       int64_t ldd, float* dw, float* db, uint32_t* counter, void* ws, size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Layer-level composition (host-side only: each call enqueues the per-op kernels above in order).
 * One chem GIN layer + its outer BatchNorm (chem/model.py:37-55,269-275):
 *   agg = aggregate(x) ; hid = relu(agg W1^T + b1) ; z = hid W2^T + b2 ; y = BN(z) [ReLU]
 * agg [n,dim], hid [n,2dim], z [n,dim] are caller-owned and are what the backward needs.
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_chem_gin_layer_workspace_bytes(int64_t n, int64_t dim);
int pgnn_chem_gin_layer_fwd(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src,
                            const uint8_t* in_code, const float* emb1, const float* emb2, const float* w1,
                            const float* b1, const float* w2, const float* b2, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum,
                            float eps, int training, int relu, float* agg, float* hid, float* z, float* y,
                            float* save_mean, float* save_invstd, float drop_p, uint64_t drop_seed,
                            int64_t n, int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream);
/* backward: dx may be NULL (layer 0 input needs no gradient path other than the embedding's);
 * demb is [9, dim]: rows 0..5 = d edge_embedding1, rows 6..8 = d edge_embedding2. */
int pgnn_chem_gin_layer_bwd(const float* dy, int64_t lddy, const float* agg, const float* hid, const float* z,
                            const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat,
                            const float* w1, const float* w2, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, int training, int relu,
                            float* dx, float* demb, float* dw1, float* db1, float* dw2, float* db2,
                            float* dgamma, float* dbeta, float drop_p, uint64_t drop_seed, int64_t n,
                            int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * The whole node-embedding network as ONE call per direction: chem/model.py:258-277 with JK="last"
 * and no dropout -- x_embedding1(x[:,0]) + x_embedding2(x[:,1]), then num_layer x (GINConv,
 * BatchNorm1d, ReLU except after the last layer) -- and, backward, everything autograd would run for
 * it including the two atom-embedding table gradients.  Same kernels and order as the per-layer calls
 * (bit-identical results).  Activations are caller-owned and contiguous:
 *   h0 [n,dim]; acts [num_layer][3][n][dim] = (agg, z, y) per layer; hid [num_layer][n][2dim];
 *   stats [num_layer][4][dim] = (batch mean, 1/std, scale a, shift b).  The output is acts[num_layer-1][2];
 *   the y slot of earlier layers is only written when the BatchNorm output has to be materialised
 *   (dropout, dim > 320, PGNN_FUSE_BN_AGG=0) -- otherwise the next layer's aggregation applies
 *   relu(a*z+b) on read.
 * ------------------------------------------------------------------------------------------ */
typedef struct pgnn_gin_layer {
  const float *emb1, *emb2;           /* edge_embedding1 [6,dim], edge_embedding2 [3,dim]   (model.py:32-33) */
  const float *w1, *b1, *w2, *b2;     /* mlp: [2dim,dim], [2dim], [dim,2dim], [dim]         (model.py:29)    */
  const float *gamma, *beta;          /* batch_norms[l] affine                              (model.py:252)   */
  float *running_mean, *running_var;  /* NULL: not tracked */
  float momentum, eps;
  /* gradients, written by the backward only */
  float *demb /* [9,dim]: rows 0..5 d emb1, 6..8 d emb2 */, *dw1, *db1, *dw2, *db2, *dgamma, *dbeta;
  /* batch_norms[l].num_batches_tracked (device int64) or NULL: incremented by a stack FORWARD call in training mode, as
   * nn.BatchNorm1d.forward does -- inside a launch the call makes anyway (torch's own increment of the five counters is one more
   * launch and ~70 us of host time per train step) */
  int64_t* num_batches_tracked;
} pgnn_gin_layer;

size_t pgnn_chem_gin_stack_workspace_bytes(int64_t n, int64_t dim, int64_t rows1, int64_t rows2,
                                           int64_t num_layer);
int pgnn_chem_gin_stack_fwd(const int64_t* x_idx /* [n,2] atom type, chirality */, const float* xemb1,
                            int64_t rows1, const float* xemb2, int64_t rows2, const int32_t* in_ptr,
                            const int32_t* in_src, const uint8_t* in_code, const pgnn_gin_layer* layers,
                            int num_layer, int training, float* h0, float* acts, float* hid, float* stats,
                            int32_t* status, float drop_p, uint64_t drop_seed, int64_t n, int64_t dim,
                            void* ws, size_t ws_bytes, pgnn_stream stream);
/* drop_p > 0: dropout after every layer (layer l uses seed drop_seed + l), as GNN.forward does when
 * drop_ratio > 0 in training mode. */
/* dy: gradient of acts[num_layer-1][2].  dxemb1 [rows1,dim] / dxemb2 [rows2,dim] may be NULL. */
int pgnn_chem_gin_stack_bwd(const float* dy, int64_t lddy, const int64_t* x_idx, int64_t rows1, int64_t rows2,
                            const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat,
                            const pgnn_gin_layer* layers, int num_layer, int training, const float* acts,
                            const float* hid, const float* stats, float* dxemb1, float* dxemb2, float drop_p,
                            uint64_t drop_seed, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                            pgnn_stream stream);

/* Row support of the gradient the NEXT pgnn_chem_gin_stack_bwd of this host thread receives: `rows` [count] (int64, none repeated)
 * are the only rows where its dy is not zero -- the masking head's gradient touches node_rep[masked_atom_indices] only
 * (chem/pretrain_masking.py:51-52), ~17 % of the rows -- so the column sums of the top layer's BatchNorm backward visit those rows
 * instead of all of them (every other row would add an exact zero: same sums up to the order of the additions).  The hint is
 * consumed (or dropped: another dy pointer, PGNN_SPARSE_TOP_GRAD=0) by that call; pgnn_masked_head_bwd's caller sets it. */
int pgnn_stack_bwd_dy_rows(const float* dy, const int64_t* rows, int64_t count);


/* Gradient milestone of the next pgnn_chem_gin_stack_bwd of ONE network on the current device, whichever host thread runs it (data
 * parallelism: the reference is single-device, chem/pretrain_masking.py:114; this is what lets the gradient all-reduce of the top
 * layers start under the backward of the lower ones).  arm(layer, network): `network` = the w1 pointer of that network's layer
 * `layer` (pgnn_gin_layer.w1: parameter storage is stable across steps) -- the backwards of other networks leave the milestone
 * alone (ABI 10; context prediction runs two networks under one set of optimizers).  Once the armed network's backward has
 * enqueued layer `layer`, every parameter gradient of layers >= `layer` (weights, biases, BatchNorm, edge tables) is behind one
 * of two events, one per stream the backward uses; layer < 0 disarms.
 * wait(stream): makes `stream` wait for both events and returns 0; returns 1 -- and enqueues nothing -- when no backward of the
 * armed network has reached the layer since arm(), or when MORE THAN ONE has (gradient accumulation: the later backwards add
 * into gradients the recorded events do not cover): order the stream behind the whole backward instead. */
int pgnn_stack_bwd_milestone_arm(int layer, const void* network);
int pgnn_stack_bwd_milestone_wait(pgnn_stream stream);

/* The same one-call network for the "Linear, then aggregate" convolutions: kind 1 = GCNConv
 * (chem/model.py:58-104, needs dinv), kind 2 = GraphSAGEConv (chem/model.py:165-202, needs norms).
 * acts [num_layer][4][n][dim] = (lin, sum [GraphSAGE], z, y); norms [num_layer][n] (GraphSAGE);
 * stats [num_layer][4][dim].  pgnn_gin_layer: w1/b1 (+dw1/db1) hold the Linear, w2/b2 are unused.
 * Output = acts[num_layer-1][3]. */
size_t pgnn_chem_lin_stack_workspace_bytes(int64_t n, int64_t dim, int64_t rows1, int64_t rows2);
int pgnn_chem_lin_stack_fwd(int kind, const int64_t* x_idx, const float* xemb1, int64_t rows1, const float* xemb2,
                            int64_t rows2, const int32_t* in_ptr, const int32_t* in_src, const uint8_t* in_code,
                            const float* dinv, const pgnn_gin_layer* layers, int num_layer, int training,
                            float* h0, float* acts, float* norms, float* stats, int32_t* status, float drop_p,
                            uint64_t drop_seed, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                            pgnn_stream stream);
int pgnn_chem_lin_stack_bwd(int kind, const float* dy, int64_t lddy, const int64_t* x_idx, int64_t rows1,
                            int64_t rows2, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_dst,
                            const float* dinv, const float* cfeat, const pgnn_gin_layer* layers, int num_layer,
                            int training, const float* h0, const float* acts, const float* norms,
                            const float* stats, float* dxemb1, float* dxemb2, float drop_p, uint64_t drop_seed,
                            int64_t n, int64_t dim, void* ws, size_t ws_bytes, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * The bio GIN network (bio/model.py:11-58, 258-290; JK = "last", no dropout) in one call per direction:
 *   agg = [sum_j h_j + h_i | cfeat . EncT] ; pre = agg W1^T + b1 ; hid = relu(BatchNorm_2D(pre)) ; y = hid W2^T + b2,
 *   ReLU between layers.  pgnn_gin_layer is reused: w1 [2D,2D], b1 [2D], w2 [D,2D], b2 [D], gamma / beta / running
 *   statistics of the BatchNorm1d(2D) inside the mlp, and the edge encoder (bio/model.py:24) in one of two forms, the same
 *   for every layer of a call:
 *     emb2 == NULL: emb1 = EncT [10, dim] = [W_enc^T; b_enc], built by the caller; gradient demb = d EncT [10, dim];
 *     emb2 != NULL: emb1 = edge_encoder.weight [dim, 9], emb2 = edge_encoder.bias [dim] as the module holds them (the
 *       forward writes the tables into its workspace, in the launch that splits the weights; at most 16 layers);
 *       gradient demb = d weight [dim, 9] followed by d bias [dim] (10 * dim floats).
 *   The other gradients as named.  h0 [n, ldh0] = the first layer's (embedded) input; cfeat [n, 10] and
 *   the CSRs from pgnn_bio_graph_build; tile_start / num_tiles from pgnn_graph_tiles (NULL: untiled aggregation).
 *   acts [num_layer][7][n][dim] = (agg: 2 slots, pre: 2, hid: 2, y: 1); stats [num_layer][2][2*dim] = (mean, 1/std);
 *   the output is acts[num_layer-1][6].  Backward: dy = its gradient; dh0 [n, dim] may be NULL.
 * ------------------------------------------------------------------------------------------ */
size_t pgnn_bio_gin_stack_workspace_bytes(int64_t n, int64_t dim, int64_t num_layer);
int pgnn_bio_gin_stack_fwd(const float* h0, int64_t ldh0, const int32_t* in_ptr, const int32_t* in_src, const float* cfeat,
                           const int32_t* tile_start, const int32_t* num_tiles, const pgnn_gin_layer* layers, int num_layer,
                           int training, float* acts, float* stats, int64_t n, int64_t dim, void* ws, size_t ws_bytes,
                           pgnn_stream stream);
int pgnn_bio_gin_stack_bwd(const float* dy, int64_t lddy, const int32_t* out_ptr, const int32_t* out_dst, const float* cfeat,
                           const int32_t* tile_start, const int32_t* num_tiles, const pgnn_gin_layer* layers, int num_layer,
                           int training, const float* acts, const float* stats, float* dh0, int64_t n, int64_t dim, void* ws,
                           size_t ws_bytes, pgnn_stream stream);


/* ------------------------------------------------------------------------------------------
 * Device-side batching over a dataset resident in HBM.  The dataset is kept in
 * the concatenated (data, slices) form InMemoryDataset stores (chem/loader.py MoleculeDataset,
 * bio/loader.py BioDataset): x_all [sum n, *], edge_index_all [2, sum e] with graph-local node ids,
 * edge_attr_all [sum e, *], node_slice / edge_slice [G+1].  Replaces, for a list of graph ids,
 * BatchMasking.from_data_list (chem/batch.py:17-52; bio/batch.py:70-106), MaskAtom
 * (chem/util.py:225-244) and bio MaskEdge (bio/util.py:77-102).  All sizes are known to the host from its copy of the slices, so no call
 * synchronises; `status` collects bit 0 = graph id out of range, bit 1 = totals differ from the
 * caller's, bit 2 = masked index out of range.
 * ------------------------------------------------------------------------------------------ */
/* node_off / edge_off / mask_off [num_graphs+1]: exclusive sums of the batch's per-graph node, edge and
 * masked-item counts.  mask_unit 0: none; 1: atoms, int(n * mask_rate + 1) per graph (chem/util.py:232);
 * 2: undirected edges, int(e/2 * mask_rate + 1) per graph (bio/util.py:79-80) */
int pgnn_batch_offsets(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs,
                       const int64_t* node_slice, const int64_t* edge_slice, double mask_rate, int mask_unit,
                       int64_t* node_off, int64_t* edge_off, int64_t* mask_off, int64_t expect_nodes,
                       int64_t expect_edges, int64_t expect_masked, int32_t* status, pgnn_stream stream);
/* x [N, x_row_bytes], edge_index [2, E] = local ids + the graph's node offset (batch.py:38-39),
 * edge_attr [E, attr_row_bytes], batch [N] = position of the graph in graph_ids (batch.py:36) */
int pgnn_collate_graphs(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs,
                        const int64_t* node_slice, const int64_t* edge_slice, const int64_t* node_off,
                        const int64_t* edge_off, const void* x_all, int64_t x_row_bytes,
                        const int64_t* edge_index_all, int64_t edges_all, const void* edge_attr_all,
                        int64_t attr_row_bytes, int64_t num_nodes, int64_t num_edges, void* x,
                        int64_t* edge_index, void* edge_attr, int64_t* batch, pgnn_stream stream);
/* The batch's graph structure -- what pgnn_chem_graph_build / pgnn_bio_graph_build (gcn = 0) would compute from the collated
 * edge_index / edge_attr -- by offset-add from ONE structure built over the whole resident dataset (ds_* arrays: that build's outputs
 * on the dataset's graphs taken as one block-diagonal batch, i.e. dataset-global row pointers and node ids): a batch of whole graphs
 * is block diagonal (chem/batch.py:31-52, bio/batch.py:70-106), so its CSRs are the per-graph CSRs concatenated.  Bit-identical to
 * the per-batch build; replaces its histogram / scan / fill / sort launches (and the int64 COO as their input: SURVEY 8f rank 1) by
 * one gather.  ds_in_code / in_code: chem only (both NULL for bio).  node_off / edge_off: from pgnn_batch_offsets. */
int pgnn_collate_structure(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs, const int64_t* node_slice,
                           const int64_t* edge_slice, const int64_t* node_off, const int64_t* edge_off, const int32_t* ds_in_ptr,
                           const int32_t* ds_in_src, const uint8_t* ds_in_code, const int32_t* ds_out_ptr, const int32_t* ds_out_dst,
                           const float* ds_dinv, const float* ds_cfeat, int64_t cfeat_cols, int64_t num_nodes, int64_t num_edges,
                           int32_t* in_ptr, int32_t* in_src, uint8_t* in_code, int32_t* out_ptr, int32_t* out_dst, float* dinv,
                           float* cfeat, pgnn_stream stream);
/* masked_indices [M]: per graph, its share of distinct items drawn uniformly (counter-based keys from
 * (seed, graph id, item): the same graph gets the same draw wherever it sits in a batch).
 * unit_div 1: items = atoms, unit_off = node_off, output = batch node positions (batch.py:39-40);
 * unit_div 2: items = undirected edges, unit_off = edge_off, output = batch index of the pair's first
 * direction (bio/util.py:82-83, bio/batch.py masked_edge_idx + edge cumsum). */
int pgnn_mask_select(const int64_t* graph_ids, int64_t num_graphs, const int64_t* unit_off, int unit_div,
                     const int64_t* mask_off, int64_t num_units, uint64_t seed, int64_t* masked_indices,
                     pgnn_stream stream);
/* mask_node_label[i] = x[idx_i] ; x[idx_i] = [mask_token, 0, ...]   (chem/util.py:236-244) */
int pgnn_mask_atoms_apply(const int64_t* masked_atom_indices, int64_t num_masked, int64_t* x, int64_t x_cols,
                          int64_t num_nodes, int64_t mask_token, int64_t* mask_node_label, int32_t* status,
                          pgnn_stream stream);

/* mask_edge_label[i] = edge_attr[idx_i] ; edge_attr[idx_i] = edge_attr[idx_i + 1] = [0,...,0,1]
 * (bio/util.py:85-102) */
int pgnn_mask_edges_apply(const int64_t* masked_edge_idx, int64_t num_masked, float* edge_attr,
                          int64_t attr_cols, int64_t num_edges, float* mask_edge_label, int32_t* status,
                          pgnn_stream stream);

/* ExtractSubstructureContextPair (chem/util.py:96-149) + BatchSubstructContext.from_data_list
 * (chem/batch.py:141-210) over the resident dataset, for the graphs of one batch (node_off / edge_off from
 * pgnn_batch_offsets with mask_unit 0).  Per graph: BFS distances from a root atom (roots[g], graph-local,
 * or drawn from (seed, graph id) when roots == NULL); substructure = atoms within k hops; context = atoms
 * with l1 < dist <= l2; overlap = both.  Graphs whose context or overlap is empty are dropped (batch.py:169).
 *   plan : dist / sub_rank / ctx_rank [N], esub_rank / ectx_rank [E] (rank inside the induced sub-graph or
 *          -1), counts [B][6] = (n_sub, e_sub, n_ctx, e_ctx, n_overlap, kept), root_out [B], and
 *          offsets [6][B+1] = exclusive sums of the count columns over the batch (offsets[c][B] = totals,
 *          the only values the host has to read back to size the outputs).
 *   fill : the two induced, renumbered, concatenated graphs (atoms ascending, bonds in original order),
 *          center_substruct_idx [kept], overlap_context_substruct_idx / batch_overlapped_context
 *          [total overlap], overlapped_context_size [kept]. */
int pgnn_substruct_context_plan(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs,
                                const int64_t* node_slice, const int64_t* edge_slice, const int64_t* node_off,
                                const int64_t* edge_off, const int64_t* edge_index_all, int64_t edges_all,
                                const int64_t* roots, uint64_t seed, int k, int l1, int l2, int32_t* dist,
                                int32_t* sub_rank, int32_t* ctx_rank, int32_t* esub_rank, int32_t* ectx_rank,
                                int64_t* counts, int64_t* root_out, int64_t* offsets, pgnn_stream stream);
int pgnn_substruct_context_fill(const int64_t* graph_ids, int64_t num_graphs, int64_t dataset_graphs,
                                const int64_t* node_slice, const int64_t* edge_slice, const int64_t* node_off,
                                const int64_t* edge_off, const int64_t* offsets, const int64_t* counts,
                                const int64_t* root, const int32_t* sub_rank, const int32_t* ctx_rank,
                                const int32_t* esub_rank, const int32_t* ectx_rank, const void* x_all,
                                int64_t x_row_bytes, const int64_t* edge_index_all, int64_t edges_all,
                                const void* edge_attr_all, int64_t attr_row_bytes,
                                int64_t context_attr_zero_from_byte, int64_t num_nodes, int64_t num_edges,
                                void* x_substruct, int64_t* edge_index_substruct, void* edge_attr_substruct,
                                void* x_context, int64_t* edge_index_context, void* edge_attr_context,
                                int64_t* center_substruct_idx, int64_t* overlap_context_substruct_idx,
                                int64_t* batch_overlapped_context, int64_t* overlapped_context_size,
                                pgnn_stream stream);
/* The same two calls serve the bio transform, ExtractSubstructureContextPair(l1, center=True) (bio/util.py:123-209) +
 * bio BatchSubstructContext (bio/batch.py:127-232): pass k = -1 (the substructure is the WHOLE ego net, unreachable
 * nodes included), l2 = -1 (the context is every node farther than l1 hops from the root, unreachable nodes
 * included), roots = center_node_idx, and context_attr_zero_from_byte = 28 (the context graph is rebuilt from the
 * w1..w7 flags only: the self-loop and mask columns of its [E,9] float attributes read 0, bio/loader.py:56-60,134).
 * Feature rows (int64 [N,2] / [E,2] for chem, float [N,1] / [E,9] for bio) are copied as 32-bit words:
 * x_row_bytes / attr_row_bytes must be multiples of 4; context_attr_zero_from_byte < 0 copies whole rows. */

/* ------------------------------------------------------------------------------------------
 * Graph-resident aggregation for batches of small dense graphs (bio ego nets; csrc/tile.hip).
 * ------------------------------------------------------------------------------------------ */

/* closed node intervals of the batch (no edge crosses an interval boundary; for a block-diagonal batch: the graphs):
 * tile_start [num_nodes + 1] receives T + 1 ascending boundaries (0 ... num_nodes), *num_tiles = T, both on the device. */
size_t pgnn_graph_tiles_workspace_bytes(int64_t num_nodes);
int pgnn_graph_tiles(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                     int64_t num_nodes, int32_t* tile_start, int32_t* num_tiles, void* ws, size_t ws_bytes, pgnn_stream stream);
/* pgnn_neighbor_sum with the rows of every interval resident in LDS while its nodes are summed (bit-identical result).
 * cfeat != NULL folds the edge-feature half of the bio message into the same pass (bio/model.py:47,55,102-114):
 *   feat_out[i, :] (+)= cfeat[i, 0:kc] . table[0:kc, :]      -- the fmaf chain of pgnn_rowfeat_matmul_fwd;
 * feat_out == out continues from the neighbour sum (GCN: one [N, D] result), otherwise it starts from zero in its own
 * columns (GIN: the second half of the [N, 2D] concat message).  kc <= 10. */
int pgnn_neighbor_sum_tiled(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* nbr, const float* dinv,
                            const int32_t* tile_start, const int32_t* num_tiles, float* out, int64_t ldo, int64_t num_nodes,
                            int64_t dim, const float* cfeat, int64_t kc, const float* table, int64_t ldt, float* feat_out,
                            int64_t ld_feat_out, pgnn_stream stream);

/* ------------------------------------------------------------------------------------------
 * Attention layers (off the north-star path, native and deterministic): csrc/attention.hip.
 * ------------------------------------------------------------------------------------------ */

/* 2-head GATConv message / edge soft-max / aggregate / update (chem/model.py:133-162, bio/model.py:147-180) on the CSR of
 * the graph build.  Common: xh [N, 2*dim] = weight_linear(x), heads side by side; att [2, 2*dim] (the reference's `att`
 * parameter [1, heads, 2*emb_dim]); bias [dim]; scores [N, 4], z / alpha [E + N, 2] (slot p of node i at p + i, its self loop at
 * in_ptr[i+1] + i) and cfa are saved for the backward; out [N, dim].  The soft-max follows torch_geometric 1.0.3 on
 * torch_scatter 1.1.2 (shift = max(0, segment max), + 1e-16).
 *   chem form : emb1 [6, 2*dim], emb2 [3, 2*dim], in_code, ctab [18, 2] = (emb1[t] + emb2[d]) . att[h, dim:2dim] (parameter
 *               space, host side); cfa [2, N, 9]; slot_feat = NULL.
 *   bio form  : emb1 = NULL; slot_feat [E, kf] = the edge attributes in CSR-slot order with a trailing constant 1 (kf = 10),
 *               self_feat [kf] (one-hot 7 and the 1), wv [kf, 2] = Tenc_h . att[h, dim:2dim] with Tenc = [W_enc^T; b] [kf, 2*dim];
 *               cfa [2, N, kf].  The edge_encoder output [E, 2*dim] is never formed: out lacks the term
 *               sum_h cfa[h] . Tenc_h, which the caller adds with pgnn_rowfeat_matmul_fwd (accumulate). */
int pgnn_gat_fwd(const float* xh, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const uint8_t* in_code,
                 const float* emb1, const float* emb2, const float* ctab, const float* slot_feat, const float* self_feat,
                 const float* wv, int64_t kf, const float* att, const float* bias, float negative_slope, float* scores,
                 float* z, float* alpha, float* cfa, float* out, int64_t ldo, int64_t num_nodes, int64_t dim, pgnn_stream stream);
/* backward: dalpha [E + N, 2] scratch (receives dz), dsd [2, N, 2] = per head (sum dz over the in-segment, sum dz over the
 * out-edges + self), czf [2, N, 9 | kf] = dz per bond type / direction (chem) or dz-weighted feature sums (bio), wout [E, 2]
 * scratch, dxh [N, 2*dim]; tenc [kf, 2*dim] (bio).  The parameter gradients follow from these with pgnn_rowfeat_matmul_bwd
 * (kc = 2: att; kc = 9 / 10: bond tables / encoder) -- see ops.GATAggregate / ops.BioGATAggregate. */
int pgnn_gat_bwd(const float* g, int64_t ldg, const float* xh, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src,
                 const uint8_t* in_code, const int32_t* out_ptr, const int32_t* out_dst, const float* emb1, const float* emb2,
                 const float* slot_feat, const float* self_feat, const float* tenc, int64_t kf, const float* att,
                 float negative_slope, const float* z, const float* alpha, float* dalpha, float* dsd, float* czf, float* wout,
                 float* dxh, int64_t ldd, int64_t num_nodes, int64_t dim, pgnn_stream stream);

/* soft-max over segments (torch_geometric.utils.softmax 1.0.3): items perm[ptr[s] .. ptr[s+1]) (perm NULL = identity),
 * z / alpha [items, heads].  GlobalAttention gate and Set2Set attention (chem/model.py:329-339). */
int pgnn_segment_softmax_fwd(const float* z, const int32_t* ptr, const int32_t* perm, float* alpha, int64_t num_segments,
                             int64_t heads, pgnn_stream stream);
int pgnn_segment_softmax_bwd(const float* alpha, const float* dalpha, const int32_t* ptr, const int32_t* perm, float* dz,
                             int64_t num_segments, int64_t heads, pgnn_stream stream);

/* global_max_pool (chem/model.py:327-328): column-wise max over the items of a segment (empty segment -> 0) and the item
 * that attains it (arg [segments, dim], first maximum); backward routes g[key[i]] to the arg items. */
int pgnn_segment_max_fwd(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* perm, float* out, int64_t ldo,
                         int32_t* arg, int64_t num_segments, int64_t dim, pgnn_stream stream);
int pgnn_segment_max_bwd(const float* g, int64_t ldg, const int64_t* key, const int32_t* arg, float* dx, int64_t ldd,
                         int64_t num_segments, int64_t num_items, int64_t dim, pgnn_stream stream);

/* diagnostics: plain float4 grid-stride copy (the HBM streaming ceiling bench.py quotes next to the
 * aggregation kernel).  Not part of the hot path. */
int pgnn_debug_stream_copy(const float* src, float* dst, int64_t n_floats, int64_t blocks, pgnn_stream stream);

/* diagnostics: device buffer [blocks][8] of uint64 that the instrumented aggregation kernel (PGNN_DMA_POL=8) fills
 * with shader-cycle totals per block: {loader vmcnt wait, loader barrier, loader issue, consumer barrier, consumer
 * work, block total, steps, 0}.  NULL detaches.  Not part of the hot path. */
int pgnn_debug_aggregate_profile(uint64_t* buffer, int64_t blocks);
/* Instrumented build of the pre-split-weights product (pgnn_linear_fwd_wp with ReLU; cfg as PGNN_GEMM3W_CFG 0 / 1 / 5 / 6): lane 0
 * of every wave of the first 8 workgroups writes buffer[workgroup][wave][8] = shader-cycle totals {wait + barrier, DMA issue,
 * A split, multiply, loop total, k-steps, 0, 0}.  Not part of the hot path. */
int pgnn_debug_gemm3w_profile(const float* x, int64_t ldx, const void* wplanes, const float* bias, float* y, int64_t ldy, int64_t m,
                              int64_t k, int64_t n, int cfg, uint64_t* buffer, pgnn_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PGNN_H */
