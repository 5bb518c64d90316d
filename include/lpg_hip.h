/*
 * lpg_hip.h -- C ABI of liblpg_hip.so: MI355X (gfx950 / CDNA4) message-passing aggregation
 * kernels behind the torch_scatter / torch_sparse / PyG-GCNConv call signatures used by
 * NguyenTrieu903/Link-Prediction-GNN (reference file:line cited per entry point; paths are
 * relative to the reference repo root).
 *
 * The reference has NO FFI of its own for this path: its boundary is the Python operator API of
 * three third-party wheels.  This header is therefore the ABI those
 * Python signatures bind to in this build; INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor) unless marked "host";
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work,
 *    they never synchronise unless the comment says "syncs";
 *  - no allocation inside: scratch comes from the caller, sized by the matching *_bytes() query;
 *  - return value: 0 = ok, <0 = error (LPG_E_*); text via lpg_last_error() (thread-local);
 *  - index tensors are int64 at the API (PyG convention) and narrowed to int32 once per graph;
 *  - `err_flag` (int32 on device, caller-zeroed) is OR-ed with LPG_F_* bits by kernels that take
 *    user indices; out-of-range indices are SKIPPED (never dereferenced) and flagged.
 */
#ifndef LPG_HIP_H
#define LPG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPG_ABI_VERSION 1

#define LPG_OK 0
#define LPG_E_INVALID (-1)   /* bad argument (null pointer, negative size, misalignment) */
#define LPG_E_LAUNCH (-2)    /* hip launch / runtime error */
#define LPG_E_WORKSPACE (-3) /* workspace too small */
#define LPG_E_UNSUPPORTED (-4)

#define LPG_F_INDEX_RANGE 1 /* an index was outside [0, limit) */

#define LPG_ACT_NONE 0
#define LPG_ACT_RELU 1
#define LPG_ACT_TANH 2

const char* lpg_last_error(void);
int lpg_abi_version(void);
/* Kernel-selection options (csrc/options.hip; build-defined, no reference line): a process-wide table the launch
 * paths read with plain loads -- the library never calls getenv on a launch path, and a release build never reads
 * the environment at all.  Every value a release build accepts selects between variants that all return correct
 * results (the tests pin each to the oracle): "spmm_chunk", "spmm_mode" (0 / 1), "spmm_lds", "spmm_block" (0 automatic, 64 / 128 / 256),
 * "spmm_off32", "spmm_rows_max", "spmm_rows_ecap", "spmm_no_edge", "spmm_fill_overlap" (1: big graphs fill
 * their edge-less rows on a library-owned second stream beside the main kernel, joined before the call's work on the
 * caller's stream ends; never inside a stream capture), "seal_conv_fp32", "gn_small",
 * "seal_conv_variant", "linear_percu_delta", "linear_wres" (-1 automatic, 0 off, 1..3 force a W-resident variant),
 * "linear_x3" (-1 automatic, 0 off, 1 force the strip form / 2 the tiled form of the fp32 linear as exact three-plane bf16 products, csrc/linear_x3.hip), "seal_pair_small",
 * "spmm_phase_timing" (1: lpg_csr_spmm_* record events around their kernels, see lpg_csr_spmm_phase_ms).
 * `name` is a HOST string.  set: 0 / LPG_E_INVALID (unknown name, value out of range); get: the value, INT64_MIN
 * for an unknown name.  lpg_tuning_build() = 1 for a -DLPG_TUNING build (profiles/ sweeps only: seeds the table
 * once from LPG_<NAME> environment variables and accepts the measurement-only spmm_mode = 2), else 0. */
int lpg_set_option(const char* name, int64_t value);
int64_t lpg_get_option(const char* name);
int lpg_reset_options(void);
int lpg_tuning_build(void);
/* edges per work chunk the chunked SpMM uses for a graph of capacity e_cap (16..512; tests use it
 * to place rows on chunk borders) */
int lpg_spmm_chunk_edges(int64_t e_cap);
/* the association order of the chunked SpMM's fix-up, for a same-order reference loop (oracle/spmm_ref.c): a row whose
 * per-chunk pieces after the first number more than lpg_spmm_fixup_long_row() is summed as `groups` interleaved
 * left-to-right sub-sums (piece j goes to sub-sum j % groups) folded by a balanced binary tree, then added to the first
 * piece; shorter rows are summed left to right.  wide = d, ldx, ldy multiples of 16 bytes and x, y 16-byte aligned. */
int lpg_spmm_fixup_groups(int64_t d, int elem_bytes, int wide);
int lpg_spmm_fixup_long_row(void);

/* ---------------------------------------------------------------- scatter_add (K1, K2)
 * torch_scatter.scatter_add(src, index, dim=0, out=None, dim_size=N)   [TwoWL/utils.py:5,10]
 * out[index[e], :] += src[e, :].  `out` is accumulated into (caller zeroes it when out=None).
 * i64: src == NULL means "ones" (the degree() call).  Integer adds commute => bit-exact.
 * f32: global_atomic_add_f32, row-contiguous; order of adds is not fixed (use the CSR path for
 *      a reproducible sum). */
int lpg_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t E, int64_t d,
                        int64_t* out, int64_t N, int32_t* err_flag, void* stream);
int lpg_scatter_add_f32(const float* src, const int64_t* index, int64_t E, int64_t d,
                        float* out, int64_t N, int32_t* err_flag, void* stream);
/* out[index[e], :] += src[e, :] for DISTINCT index entries (a caller contract: the rows one peer contributes to
 * a shard in the sparse multi-GPU exchange, lpgnn_amd/dist.py): plain 16-byte read-modify-writes, no atomics. */
int lpg_rows_add_f32(const float* src, const int64_t* index, int64_t E, int64_t d, float* out,
                     int64_t N, int32_t* err_flag, void* stream);
/* the same for bf16 source rows (bf16 partials on the links, SURVEY 2b C1 "bf16 option"; build-defined): the owner's
 * shard stays fp32 while the peers' rows are added, so the sum over the peers is rounded once at the end. */
int lpg_rows_add_bf16(const uint16_t* src, const int64_t* index, int64_t E, int64_t d, float* out,
                      int64_t N, int32_t* err_flag, void* stream);

/* In-degree histogram, the reference's only direct torch_scatter call:
 * degree(ei, num_node) = scatter_add(ones_like(ei[1]), ei[1], dim_size=N)  [TwoWL/utils.py:8-10].
 * out[index[e]] += 1, int64, bit-exact.  Big lists (E >= 2 M, N > 16 K) are counted without global atomics: one
 * radix pass over the bits above the low 15 makes each 32 K-bin bucket contiguous, workgroups count 256 K keys at a
 * time in 128 KB of LDS and the partial histograms are folded with plain adds (|E| = 100 M: scattered 64-bit atomics
 * 10 ms, this 1.1 ms; a full sort + run lengths when the buckets would be tiny, N > 32 M or E < 2 N); everything
 * else, or a NULL workspace, takes the kernels behind lpg_scatter_add_i64(src = NULL). */
int64_t lpg_degree_i64_bytes(int64_t E, int64_t N);
int lpg_degree_i64(const int64_t* index, int64_t E, int64_t* out, int64_t N, void* workspace,
                   int64_t workspace_bytes, int32_t* err_flag, void* stream);
/* The reference's own spelling of that histogram, scatter_add(ones_like(index), index, dim_size=N) [TwoWL/utils.py:10]:
 * `ones` [E] is read beside the index inside the atomic-free paths (E >= 2^21, N > 16384) and *not_ones (device int32,
 * caller-zeroed) is set when any entry differs from 1 -- the caller then discards `out` and takes
 * lpg_scatter_add_i64 -- instead of a separate comparison pass over `ones` and its host sync. */
int lpg_degree_i64_ones(const int64_t* index, const int64_t* ones, int64_t E, int64_t* out, int64_t N,
                        void* workspace, int64_t workspace_bytes, int32_t* not_ones, int32_t* err_flag,
                        void* stream);

/* ---------------------------------------------------------------- COO -> CSR (once per graph)
 * Stable sort of the E edges by destination `dst` (row of the CSR): inside a row the original
 * edge order is kept, which is the order torch's CPU scatter_add_ accumulates in.
 * gcn_self_loops != 0 reproduces torch_geometric add_remaining_self_loops as gcn_norm calls it
 * [PyG 2.3.1; call sites TwoWL/model/model.py:37,73,77]: existing self loops are dropped and one
 * loop per node is appended after the row's other edges (weight = the node's last existing loop
 * weight if `val` is given, else fill 1).  Capacity of col_out/val_out/perm_out: E (+ N with
 * self loops).  The number of stored edges is rowptr[N] (device value).
 * perm_out[k] = original edge id of CSR slot k (>= E for appended loops); may be NULL.
 * flip: bit 0 -> source ids are read as id^1, bit 1 -> destination ids as id^1: the two tuple-graph
 * orientations of TwoWL reverse() (edge = (flip(e0), e1), edge_r = (e0, flip(e1)),
 * TwoWL/utils.py:71-78) are built straight from ei2 without materialising them.
 * src == NULL: the column of entry e is e itself -- the grouping a big scatter_add needs (row r lists, in
 * order, the entries whose index is r; n_cols >= E); lpg_csr_spmm_f32 over it with x = the [E, d] source rows
 * is scatter_add without atomics. */
int64_t lpg_coo_to_csr_bytes(int64_t E, int64_t N, int gcn_self_loops);
int lpg_coo_to_csr(const int64_t* src, const int64_t* dst, const float* val, int64_t E,
                   int64_t n_rows, int64_t n_cols, int gcn_self_loops, int flip,
                   int64_t* rowptr, int32_t* col_out, float* val_out, int32_t* perm_out,
                   void* workspace, int64_t workspace_bytes, int32_t* err_flag, void* stream);

/* gcn_norm on the dst-major CSR [PyG gcn_norm; TwoWL/model/model.py:37] (needs the plan: per-edge
 * row ids): deg[j] = sum of row j's weights; dinv = deg^-1/2 (inf -> 0);
 * val[k] = dinv[col[k]] * val[k] * dinv[row(k)], edge-parallel, in place.
 * unit_weights != 0: every weight is 1 (edge_weight=None): `val` is output only and deg is the row
 * length. */
int lpg_gcn_norm_csr(const void* plan, const int64_t* rowptr, const int32_t* col, float* val,
                     float* dinv, int64_t N, int64_t e_cap, int unit_weights, void* stream);

/* ---------------------------------------------------------------- SpMM plan (once per graph)
 * Expands rowptr into a per-edge row-id stream and lists the empty rows, so the SpMM kernels
 * need no dependent rowptr walks.  Checks 0 <= col < n_cols (err_flag).  e_cap = capacity of
 * the col array (>= rowptr[N]). */
int64_t lpg_csr_plan_bytes(int64_t N, int64_t e_cap);
int lpg_csr_plan_build(const int64_t* rowptr, const int32_t* col, int64_t N, int64_t n_cols,
                       int64_t e_cap, void* plan, int64_t plan_bytes, int32_t* err_flag,
                       void* stream);

/* ---------------------------------------------------------------- CSR SpMM (K4, K10) -- hot
 * Y[N,d] = act( row_scale[:,None] * (A @ X) + bias[None,:] ),  A = (rowptr, col, val) dst-major.
 * torch_sparse.spmm / matmul(reduce="sum") and the propagate step of GCNConv
 * [TwoWL/model/model.py:73,77]; with row_scale = D^-1, act = tanh it is SEAL's
 * tanh(D^-1 (A~ (X W)))  [SEAL/model/gnn.py:47-48,51-52,55-56,59-60].
 * val == NULL: all ones.  row_scale / bias may be NULL.  ldx/ldy = row strides in elements;
 * n_cols = rows of x (the bound the plan checked `col` against).
 * Atomic-free and bitwise reproducible: edges are cut into fixed chunks, rows that straddle
 * chunks are finished by a second kernel that adds the partials in chunk order.
 * bf16 variant: x, y are bfloat16 (uint16 bit patterns), accumulation in fp32.
 * n_empty_hint: number of rows without edges if the caller knows it (0 skips the empty-row
 * launch), -1 = unknown (read from the plan on the device).
 * max_row_hint: length of the longest row if the caller knows it (the plan stores it: int32 word 1
 * of the plan buffer, next to the empty-row count in word 0), -1 = unknown.  Short rows (<= 32) on
 * graphs up to 2 M edges run as ONE launch with a lane group per row, added left to right (no chunk
 * cuts, no fix-up, empty rows included); otherwise the chunked kernels.  d in {1,2,4,8,16} uses an
 * edge-parallel kernel whose in-row additions form a fixed tree (reproducible, not left-to-right). */
int64_t lpg_csr_spmm_bytes(int64_t e_cap, int64_t d);
int lpg_csr_spmm_f32(const void* plan, const int64_t* rowptr, const int32_t* col,
                     const float* val, const float* x, int64_t ldx, float* y, int64_t ldy,
                     int64_t N, int64_t n_cols, int64_t e_cap, int64_t d, const float* row_scale,
                     const float* bias, int act, int64_t n_empty_hint, int64_t max_row_hint,
                     void* workspace, int64_t workspace_bytes, void* stream);
int lpg_csr_spmm_bf16(const void* plan, const int64_t* rowptr, const int32_t* col,
                      const float* val, const uint16_t* x, int64_t ldx, uint16_t* y, int64_t ldy,
                      int64_t N, int64_t n_cols, int64_t e_cap, int64_t d, const float* row_scale,
                      const float* bias, int act, int64_t n_empty_hint, int64_t max_row_hint,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* Per-kernel split of the most recent lpg_csr_spmm_* call made with option "spmm_phase_timing" = 1: out_ms (HOST,
 * 3 floats) = {main kernel, fix-up kernel, empty-rows kernel} in milliseconds, from events the library recorded on
 * the launch stream.  Syncs on the last of them.  bench.py reports it next to the whole-call HIP-event time so the
 * rocprofv3 kernel statistics under profiles/ can be checked against the JSON line. */
int lpg_csr_spmm_phase_ms(float* out_ms);

/* ---------------------------------------------------------------- dense feature linear (K5)
 * Y[n,m] = X[n,k] @ W^T (+ bias), W is [m,k] (nn.Linear / GCNConv.lin layout) or, with
 * w_is_km != 0, [k,m] (SEAL's tf.matmul(X, W), SEAL/model/gnn.py:46,50,54,58).
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 products).
 * act: LPG_ACT_* applied after the bias (SEAL's dense layer relu(x W + b), SEAL/model/gnn.py:127). */
int lpg_linear_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias,
                   int act, float* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream);

/* bf16 storage variant (configs[2]: "d = 128, bf16"): x, w, y are bfloat16 bit patterns, products
 * on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bias fp32, one rounding to bf16 at the store. */
int lpg_linear_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias,
                    int act, uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream);

/* Row-wise LayerNorm: y = (x - mean) / sqrt(var + eps) [* weight + bias] [ReLU], biased variance over the d columns
 * of each row -- torch.nn.LayerNorm(d, elementwise_affine=False) of LocalWLNet's node-feature branch
 * (relu_lin, TwoWL/model/model.py:28-33,47-51).  weight / bias may be NULL.  d <= 1024. */
int lpg_layernorm_rows_f32(const float* x, int64_t ldx, int64_t n, int64_t d, float eps, const float* weight,
                           const float* bias, int relu, float* y, int64_t ldy, void* stream);

/* ---------------------------------------------------------------- GraphNorm (K6)
 * torch_geometric.nn.GraphNorm.forward(x, batch=None) [TwoWL/model/model.py:38,54]:
 * o = x - mean*mean_scale; y = weight*o/sqrt(mean(o^2)+eps) + bias, column-wise over all n rows;
 * optional fused ReLU and `y = add_to + (...)` (the conv2s + conv2s_r sum, model.py:77).
 * Column sums are accumulated in fp64 over a fixed block partition (reproducible). */
int64_t lpg_graphnorm_bytes(int64_t n, int64_t d);
int lpg_graphnorm_f32(const float* x, int64_t n, int64_t d, const float* weight,
                      const float* bias, const float* mean_scale, float eps, int relu,
                      const float* add_to, float* y, void* workspace, int64_t workspace_bytes,
                      void* stream);
/* The same normalisation of a TABLE whose row v stands for row_weight[v] identical rows of the real input (n_total =
 * their sum): statistics are the weighted column sums, only the n distinct rows are normalised.  LocalWLNet's first
 * layer input GraphNorm(Embedding(deg)) [TwoWL/model/model.py:53-55,71] has one distinct row per distinct degree;
 * row_weight = the degree histogram (lpg_degree_i64 of the index vector). */
int lpg_graphnorm_weighted_f32(const float* x, int64_t n, int64_t d, const int64_t* row_weight,
                               int64_t n_total, const float* weight, const float* bias,
                               const float* mean_scale, float eps, int relu, float* y,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- TwoWL pair ops (K7-K9)
 * embedding: out[i,:] = table[idx[i],:]                                  [model.py:71]
 * pair_gather_mul: h[t,:] = x[pos[t,0],:] * x[pos[t,1],:], pos is [P,2]    [model.py:75]
 * pair_readout: s[b] = w . (h[idx[2b]] * h[idx[2b+1]]) + b0               [model.py:78-83]
 * flip_index: edge = (flip(e0), e1), edge_r = (e0, flip(e1)), flip(k) = k^1 [TwoWL/utils.py:71-78] */
int lpg_embedding_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* idx,
                      int64_t n, float* out, int32_t* err_flag, void* stream);
/* the same with every gathered row multiplied by row_scale[i] (out[i] = row_scale[i] * table[idx[i]]): the D^-1/2 of a
 * unit-weight GCN normalisation folded into the gather, so that the SpMM after it runs unweighted with row_scale =
 * D^-1/2 (twowl._one_wl_stage).  Needs 16-byte aligned rows, d % 4 == 0, d / 4 a divisor of 256. */
int lpg_embedding_scaled_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* idx,
                             int64_t n, const float* row_scale, float* out, int32_t* err_flag,
                             void* stream);
int lpg_pair_gather_mul_f32(const float* x, int64_t n, int64_t d, const int64_t* pos, int64_t P,
                            float* h, int32_t* err_flag, void* stream);
int lpg_pair_readout_f32(const float* h, int64_t P, int64_t d, const int64_t* idx, int64_t B,
                         const float* w, const float* b0, float* out, int32_t* err_flag,
                         void* stream);
int lpg_flip_index_i64(const int64_t* ei2, int64_t E2, int64_t* edge, int64_t* edge_r,
                       void* stream);
/* ---------------------------------------------------------------- small-graph GCN blocks (a5-a8)
 * The reference's own dataset (fb-pages-food: 620 nodes, 4252 tuples) is launch-bound: a
 * GCNConv + GraphNorm + Dropout(eval) + ReLU block [TwoWL/model/model.py:36-41,73,77] runs as TWO
 * launches here, for graphs of at most lpg_small_max_rows() rows with unit edge weights (PyG
 * gcn_norm).  GraphNorm + ReLU are DEFERRED: the aggregation writes the un-normalised tensor y and
 * per-row-block column sums `part` [row_blocks][2][C] (fp64: sum y, sum y^2); the consumer (next
 * linear, readout) folds them in a fixed order and normalises while loading.  A deferred-norm
 * descriptor is (part, gn_weight, gn_bias, gn_mean_scale, row_blocks, relu); part == NULL means
 * "use the tensor as it is".
 *
 * lpg_embedding_graphnorm_f32: y = GraphNorm(table[idx])  [model.py:53-55,71], one launch.
 * lpg_small_pack_build: packs a CSR graph's edge stream once per (graph, row_blocks):
 *   (col | row << 16) words in [row block][strip step][thread] order; `plan` is the
 *   lpg_csr_plan_build result for the same graph.
 * lpg_small_linear_t_f32: hT[m, ldt] = (X W^T)^T (and h2T with W2), X by `mode`:
 *   0: N_a(xa) [n,k];  1: N_a(xa[pos[t,0],:]) * N_a(xa[pos[t,1],:]), xa [n_src,k], pos [n,2]
 *   [model.py:75];  2: N_a(xa) + N_b(xb) [model.py:77].  k <= 256; ldt % 4 == 0, >= n rounded up to 4.
 * lpg_small_gcn_f32: y[n, ldo] (column c) = dinv * (A (dinv * h[:, c])) + bias[c] and `part`, for one
 *   or two (graph, bias) sets in one launch (conv2s[i] / conv2s_r[i], model.py:77; both graphs packed
 *   with the same e_cap and row_blocks); second set NULL = one side.  Atomic-free, fixed order.
 * lpg_small_gcn_embed_f32: the same aggregation for conv1 of a model without node features
 *   [model.py:53-55,71-73: x = GraphNorm(Embedding(deg)); conv1 = GCNConv(x, edge1)]: the feature column
 *   h[:, c] = GraphNorm(table[idx]) W[c, :] is made inside the launch from the table (n_table, k <= 1024), the
 *   index histogram and the distinct rows' products -- lpg_embedding_graphnorm_f32 + lpg_small_linear_t_f32 +
 *   lpg_small_gcn_f32 in one launch; out-of-range idx sets err_flag (row 0 is used).
 * lpg_small_readout_f32: out[b] = sum_c x[idx[2b],c] * x[idx[2b+1],c] * w[c] + b0 with
 *   x = N_a(ya) + N_r(yr)  (yr NULL: x = N_a(ya))  [model.py:77-83]. */
int64_t lpg_small_max_rows(void);
int64_t lpg_small_max_row_blocks(void);
int lpg_embedding_graphnorm_f32(const float* table, int64_t n_table, int64_t d, const int64_t* idx,
                                int64_t n, const float* weight, const float* bias,
                                const float* mean_scale, float eps, int relu, float* y,
                                int32_t* err_flag, void* stream);
int64_t lpg_small_pack_bytes(int64_t e_cap, int64_t row_blocks);
int lpg_small_pack_build(const void* plan, const int64_t* rowptr, const int32_t* col, int64_t n,
                         int64_t e_cap, int64_t row_blocks, void* pack, int64_t pack_bytes,
                         void* stream);
int lpg_small_linear_t_f32(const float* xa, const float* xb, const int64_t* pos, int mode,
                           int64_t n_src, int64_t n, int64_t k, const float* w, const float* w2,
                           int64_t m, float* hT, float* h2T, int64_t ldt, const double* part_a,
                           const float* gn_weight_a, const float* gn_bias_a,
                           const float* gn_mean_scale_a, int64_t row_blocks_a, int relu_a,
                           const double* part_b, const float* gn_weight_b, const float* gn_bias_b,
                           const float* gn_mean_scale_b, int64_t row_blocks_b, int relu_b, float eps,
                           int32_t* err_flag, void* stream);
int lpg_small_gcn_f32(int64_t n, int64_t c, int64_t ldt, int64_t ldo, int64_t row_blocks,
                      int64_t e_cap, const float* hT, const void* pack, const int64_t* rowptr,
                      const float* dinv, const float* bias, float* y, double* part,
                      const float* hT2, const void* pack2, const int64_t* rowptr2,
                      const float* dinv2, const float* bias2, float* y2, double* part2,
                      void* stream);
int lpg_small_gcn_embed_f32(int64_t n, int64_t c, int64_t ldo, int64_t row_blocks, int64_t e_cap,
                            const float* table, int64_t n_table, int64_t k, const int64_t* idx,
                            const float* gn_weight, const float* gn_bias, const float* gn_mean_scale,
                            float eps, const float* lin_w, const void* pack, const int64_t* rowptr,
                            const float* dinv, const float* bias, float* y, double* part,
                            int32_t* err_flag, void* stream);
int lpg_small_readout_f32(const float* ya, const float* yr, int64_t P, int64_t C,
                          const double* part_a, const float* gn_weight_a, const float* gn_bias_a,
                          const float* gn_mean_scale_a, int64_t row_blocks_a, int relu_a,
                          const double* part_r, const float* gn_weight_r, const float* gn_bias_r,
                          const float* gn_mean_scale_r, int64_t row_blocks_r, int relu_r, float eps,
                          const int64_t* idx, int64_t B, const float* w, const float* b0,
                          float* out, int32_t* err_flag, void* stream);
/* The whole small-graph scoring forward of LocalWLNet (TwoWL/model/model.py:68-84; the timed call of
 * TwoWL/model/train.py:54-60) as ONE call: `block` (HOST memory, n_words int64 words) lists the forward's launches in
 * order as { op, a_0 .. a_{k-1} } -- op 1: lpg_embedding_graphnorm_f32 (k = 12), 2: lpg_small_linear_t_f32 (27),
 * 3: lpg_small_gcn_f32 (20), 4: lpg_small_readout_f32 (23), 5: lpg_small_gcn_embed_f32 (21), 6: lpg_tuple_layer_f32 (19),
 * 7: lpg_tuple_readout_f32 (31), 8: lpg_tuple_ax_small_f32 (15); a_i = that entry point's i-th argument (pointers and
 * integers as they are, `float eps` as the bit pattern of a double; the trailing stream argument is left out).  Every
 * launch goes through the same entry point as the one-by-one path and is enqueued on `stream`: same kernels, same
 * results, one host call per forward instead of six (the forward is launch-bound: 205 -> ~60 us eager).
 * The host mirror builds the block once per (model, graph, candidate set): lpgnn_amd.twowl.LocalWLNet. */
int lpg_twowl_forward_small(const int64_t* block, int64_t n_words, void* stream);


/* ---------------------------------------------------------------- 2-WL tuple graph (a3, a2)
 * get_ei2(n_node, pos_edge, pred_edge) [TwoWL/utils.py:36-45]: tuple ids = columns of
 * cat(pos_edge, pred_edge); for node i ascending, all (a, b) with pos_edge[1,a]==i, edge[0,b]==i,
 * a-major b-minor.  Two calls: _count fills node_off[n+1] (exclusive scan of in_i*out_i) and the
 * sorted id lists in `workspace`, and returns E2 through *e2_host (syncs); _fill writes [2,E2]. */
int64_t lpg_get_ei2_bytes(int64_t n_node, int64_t E1, int64_t Ep);
int lpg_get_ei2_count(const int64_t* pos_edge, int64_t E1, const int64_t* pred_edge, int64_t Ep,
                      int64_t n_node, void* workspace, int64_t workspace_bytes,
                      int64_t* e2_host, int32_t* err_flag, void* stream);
int lpg_get_ei2_fill(const void* workspace, int64_t n_node, int64_t E1, int64_t Ep, int64_t E2,
                     int64_t* ei2, void* stream);
/* stable compaction of the columns of a [rows, E] int64 matrix where keep[e] != 0
 * (the ei[:, mask] / blockei2 filters of sample_block, TwoWL/utils.py:48-50,61-68).
 * mark_blocked: mask[idx[i]] = 1.  keep_from_blocked: keep[e] = !blocked[key[e]].
 * Returns the kept count through *count_host (syncs). */
int lpg_mark_i64(const int64_t* idx, int64_t n_idx, uint8_t* mask, int64_t n_mask,
                 int32_t* err_flag, void* stream);
int lpg_keep_from_blocked(const int64_t* key, int64_t E, const uint8_t* blocked, int64_t n_mask,
                          int invert, uint8_t* keep, int32_t* err_flag, void* stream);
int64_t lpg_compact_cols_bytes(int64_t E);
int lpg_compact_cols_i64(const int64_t* in, int64_t rows, int64_t E, const uint8_t* keep,
                         int64_t* out, void* workspace, int64_t workspace_bytes,
                         int64_t* count_host, void* stream);

/* ---------------------------------------------------------------- 2-WL aggregation WITHOUT the tuple graph (a7, f1)
 * TwoWL/utils.py:36-45 (get_ei2) links every graph tuple ending at node i with every tuple starting at i, :48-50
 * (blockei2) drops the edges of blocked source tuples, :71-78 (reverse) makes the two orientations the model convolves
 * over (TwoWL/model/model.py:76-77).  Every per-node block is complete bipartite and -- with pos_edge in double()'s
 * layout (:81-90) and blocked ids in partner pairs (TwoWL/model/train.py:26) -- all its sources share one degree, so
 * GCNConv over either orientation is a per-node sum over the MEMBER tuples (graph tuples that are not blocked) and a
 * broadcast; E2 = sum deg^2 edges are never built (csrc/tuple_wl.hip; fp64 restatement and proof-by-test:
 * oracle/aggregation.py localwl_forward_collapsed).  Tuple b = column b of cat(pos_edge, pred_edge); node_a(b) =
 * start(b), node_r(b) = start(b ^ 1); Q = the non-member tuples (candidates + blocked), numbered in tuple order.
 *
 * lpg_tuple_plan_build (once per tuple graph; pos_edge [2,E1], pred_edge [2,Ep] int64 row-major, E1 and Ep even;
 *   blocked: uint8 [E1] or NULL; rowptr: the member adjacency's CSR row pointer, int64 [n_node+1]):
 *   cnt[i] = members starting at i; qpos[b] = position of b among the non-members or -1; qinfo[q] = (start, end,
 *   node_r, b) for q < q_cap; optional tnode[b] = (node_a, node_r) and seg_a / seg_r [E1] = the member's start / end
 *   node, n_node for a non-member (scatter indices of the operator form); head (int32[4], device): [0] = Q,
 *   [1] = bit 0: layout conditions violated (use the explicit tuple graph), bit 1: a node id out of range (also
 *   err_flag).  No host sync.
 * lpg_tuple_ax_small_f32 (graphs whose longest member row is short; a workgroup per node): x1n = GraphNorm(y1)
 *   (+ ReLU) with the column sums in_part [in_rb][2][C] the small-graph conv1 leaves (NULL: x1n = y1) and ax = A x1n
 *   over the member adjacency (rowptr, col).  Big graphs use lpg_csr_spmm_f32 for A x.
 * lpg_tuple_layer_f32: one 2-WL layer, both orientations (C <= 32).  mode 1 (first layer): x = the 1-WL output
 *   [n_node, C] (model.py:73), ax = A x [n_node, C]; the pair product (model.py:75) never exists for member tuples:
 *   their group sum is x[i] * (A x)[i].  mode 2:
 *   x = (F_a | F_r) [n_node, 2C] from lpg_tuple_apply_f32, ax = A x [n_node, 2C], hq_prev [Q, C].
 *   Writes t_node [n_node, 2C] = (T_a | T_r) (may be NULL), y_q [Q, 2C] = the non-member rows (bias added) and
 *   part [2 * n_blocks][2][64]: fp64 sums of y and y^2 (column = orientation * 32 + c) -- even blocks over the member rows
 *   (through the node rows), odd blocks over the non-member rows; all of them together cover ALL P rows.  n_blocks =
 *   lpg_tuple_layer_blocks(n_node, Q).  Atomic-free, fixed order.
 * lpg_tuple_fold_f64: part [n_in][2][64] -> out [n_out][2][64], out[j] = sum of blocks j, j + n_out, ... (consumers
 *   take at most 64 partial blocks).  n_out = 2 keeps the member rows' and the non-member rows' sums apart: with the
 *   candidate list sharded over GPUs the second block is summed across ranks (lpgnn_amd.dist.ShardedTwoWLScorer).
 * lpg_tuple_apply_f32: GraphNorm + ReLU between two layers: ff [n_node, 2C], hq [Q, C] = f_a(y_a) + f_r(y_r).
 * lpg_tuple_readout_f32: model.py:78-83 with the last layer's GraphNorm + ReLU applied on load; idx may name any
 *   tuple (member rows are F_a[start] + F_r[end]).  n_rows = the rows `part` covers (P; more when the candidate list is
 *   sharded over GPUs and `part` holds the sums over all ranks' candidates).
 * lpg_tuple_bcast_f32: the broadcast of the operator form -- GCNConv(x, orientation) for an arbitrary [P, C] input:
 *   y[b] = t_node[node(b)] / cnt + bias (member), t_node[node(b)] / sqrt(cnt (cnt + 1)) + hq[q] / (cnt + 1) + bias
 *   (otherwise); t_node = (group sums of x over the members) W^T, hq = x[non-members] W^T (required whenever any
 *   qpos >= 0; NULL is only valid for a plan without non-member tuples). */
int64_t lpg_tuple_plan_bytes(int64_t P);
int lpg_tuple_plan_build(const int64_t* pos_edge, int64_t E1, const int64_t* pred_edge, int64_t Ep,
                         int64_t n_node, const uint8_t* blocked, const int64_t* rowptr, int32_t* cnt,
                         int32_t* qpos, int32_t* qinfo, int64_t q_cap, int32_t* tnode, int64_t* seg_a,
                         int64_t* seg_r, int32_t* head, void* workspace, int64_t workspace_bytes,
                         int32_t* err_flag, void* stream);
int64_t lpg_tuple_layer_blocks(int64_t n_node, int64_t Q);
int lpg_tuple_ax_small_f32(int64_t n_node, int64_t C, const float* y1, int64_t ldy,
                           const int64_t* rowptr, const int32_t* col, const double* in_part,
                           const float* in_gw, const float* in_gb, const float* in_gms, int64_t in_rb,
                           int in_relu, float in_eps, float* x1n, float* ax, void* stream);
int lpg_tuple_layer_f32(int mode, int64_t n_node, int64_t Q, int64_t C, const float* x, int64_t ldx,
                        const float* ax, int64_t ldax, const float* hq_prev, const int32_t* cnt,
                        const int32_t* qinfo, const float* w_a, const float* w_r, const float* b_a,
                        const float* b_r, float* t_node, float* y_q, double* part, int64_t n_blocks,
                        void* stream);
int lpg_tuple_fold_f64(const double* part, int64_t n_in, double* out, int64_t n_out, void* stream);
int lpg_tuple_apply_f32(int64_t n_node, int64_t Q, int64_t C, const float* t_node, const float* y_q,
                        const int32_t* cnt, const double* part, int64_t n_part, int64_t n_rows,
                        const float* gw_a, const float* gb_a, const float* gms_a, float eps_a,
                        int relu_a, const float* gw_r, const float* gb_r, const float* gms_r,
                        float eps_r, int relu_r, const float* b_a, const float* b_r, float* ff,
                        float* hq, void* stream);
int lpg_tuple_readout_f32(int64_t n_node, int64_t Q, int64_t C, int64_t E1, int64_t P,
                          const int64_t* pos_edge, const float* t_node, const float* y_q,
                          const int32_t* cnt, const int32_t* qpos, const double* part, int64_t n_part,
                          int64_t n_rows, const float* gw_a, const float* gb_a, const float* gms_a,
                          float eps_a, int relu_a, const float* gw_r, const float* gb_r,
                          const float* gms_r, float eps_r, int relu_r, const float* b_a,
                          const float* b_r, const int64_t* idx, int64_t B, const float* w,
                          const float* b0, float* out, int32_t* err_flag, void* stream);
int lpg_tuple_bcast_f32(int64_t P, int64_t C, int orient, const int32_t* tnode, const int32_t* cnt,
                        const int32_t* qpos, const float* t_node, int64_t ldt, const float* hq,
                        int64_t ldh, const float* bias, float* y, int64_t ldy, void* stream);

/* The same graph for ANY size: both directions of every pair become a 64-bit key row * n_nodes + col, the keys are
 * radix-sorted over their significant bits and deduplicated (rocPRIM), and one kernel emits col (ascending, unique
 * per row) and rowptr.  col capacity col_cap >= 2 * n_pairs; rowptr[n_nodes] = nnz stays on the device; out-of-range
 * pairs are flagged and dropped.  No host sync. */
int64_t lpg_seal_train_csr_bytes(int64_t n_pairs, int64_t n_nodes);
int lpg_seal_train_csr(const int64_t* pairs, int64_t n_pairs, int64_t n_nodes, int directed,
                       int64_t* rowptr, int32_t* col, int64_t col_cap, void* workspace,
                       int64_t workspace_bytes, int32_t* err_flag, void* stream);

/* The train-positive graph `A` / `G` of SEAL (SEAL/config/subgraph.py:27-31,70-72) for SMALL graphs
 * (n_nodes <= lpg_seal_train_csr_small_max_nodes() = 8192; fb-pages-food has 620): symmetric, neighbours ascending
 * and unique per row, self loops kept.  pairs [n_pairs, 2] int64 (the train positives); `bits` = scratch of
 * n_nodes * ceil(n_nodes / 32) * 4 bytes (zeroed by the call); rowptr [n_nodes + 1]; col capacity col_cap >=
 * 2 * n_pairs.  Three launches, no host sync (rowptr[n_nodes] = nnz stays on the device).
 * directed != 0 (network_type = 1, SEAL/config/subgraph.py:29-31: A set in ONE orientation, nx.DiGraph): rows list
 * successors only. */
int64_t lpg_seal_train_csr_small_max_nodes(void);
int lpg_seal_train_csr_small(const int64_t* pairs, int64_t n_pairs, int64_t n_nodes, int directed,
                             void* bits, int64_t bits_bytes, int64_t* rowptr, int32_t* col,
                             int64_t col_cap, int32_t* err_flag, void* stream);

/* ---------------------------------------------------------------- SEAL structure path (a10-a12)
 * Batched 0/1-hop enclosing subgraphs + DRNL labels on the train graph given as a symmetric CSR
 * (adj_rowptr[n+1], adj_col sorted ascending per row, self loops allowed)
 * [SEAL/config/subgraph.py:105-141, SEAL/utils/utils.py:35-54].
 * Per candidate pair b (pairs is [B,2]): nodes[b, 0:n_b] = [u, v, rest ascending]; capped[b] = 1
 * when the reference's RNG-dependent 100-node truncation (np.random.shuffle, :122,128) would
 * trigger in either visiting order.  cap_policy 0: such pairs come back with n = 0 and the caller
 * supplies the node list (replay != 0 makes the kernel take nodes/n_nodes as INPUT -- how tests
 * replay the reference's own lists); cap_policy 1: the kernel follows the same branch with a
 * counter-based hash of (seed, u, v) in place of the RNG (visiting order, kept neighbours).
 * tags[b, i] = DRNL tag (raw, before the global compaction); adjacency is returned as a bit
 * matrix adj_bits[b, i, w] (uint32 words, MAXN/32 words per row), target entries cleared.
 * tag_present (optional, caller-zeroed, LPG_SEAL_TAG_BOUND + 1 bytes): byte [tag] is set for every tag the
 * batch uses -- the presence table of the global tag compaction [SEAL/config/subgraph.py:84-101].
 * hub_index / hub_bits (optional, both or neither): hub_index[node] (int32, n_graph_nodes entries) = the node's row in
 * the hub bit matrix or -1; hub_bits[row * hub_words + (c >> 5)] bit (c & 31) = hubs row and c are adjacent.  Nodes
 * with more than hub_degree neighbours should be hubs: their mutual adjacency is then one load instead of a binary search.
 * hub_degree (>= 1): neighbour lists longer than this are not streamed when the induced adjacency is built -- symmetry
 * fills those rows from the short lists' side, and the entries between two long rows come from the hub matrix (or a
 * binary search).  128 on a collab-shaped graph: 0.47 -> 0.34 ms per 100 K pairs against 256; 64 and 192 are slower.
 * One launch, two kinds of workgroup: a pair whose two neighbour lists hold <= 62 entries together (n <= 64, the
 * cap cannot fire: most of a batch) is handled by ONE 64-lane wave (two pairs per workgroup, no workgroup barrier); every
 * other pair by the 128 threads of a workgroup that owns 8 consecutive pairs.  Same outputs bit for bit either way;
 * option "seal_pair_small" = 0 (or replay) gives every pair a 128-thread workgroup.
 * directed != 0 (network_type = 1): adj_rowptr / adj_col list SUCCESSORS (lpg_seal_train_csr*(directed = 1)); the
 * neighbour sets are successor sets (nx.neighbors of a DiGraph), the induced adjacency is not symmetric and the DRNL
 * distances are directed shortest paths from either target (SEAL/utils/utils.py:38); always the workgroup form. */
#define LPG_SEAL_MAXN 128
#define LPG_SEAL_TAG_BOUND (1 + 127 + 127 * 127)
int lpg_seal_subgraphs(const int64_t* adj_rowptr, const int32_t* adj_col, int64_t n_graph_nodes,
                       const int64_t* pairs, int64_t B, int hop, int max_neighbors, int replay,
                       int cap_policy, uint64_t seed, int32_t* nodes, int32_t* n_nodes,
                       int32_t* tags, uint32_t* adj_bits, uint8_t* capped, uint8_t* tag_present,
                       const int32_t* hub_index, const uint32_t* hub_bits, int64_t hub_words,
                       int64_t hub_degree, int directed, int32_t* err_flag, void* stream);
/* Measurement only (profiles/seal_phases.py; a -DLPG_TUNING build of the library, LPG_E_UNSUPPORTED otherwise): ticks of
 * the 100 MHz wall clock the 128-thread teams of lpg_seal_subgraphs spent per phase since the last reset, summed over
 * teams -- out8[0..6] = node set, row descriptors + filter, streamed lists, hub bit matrix, searches, DRNL, outputs;
 * out8[7] = pairs handled by teams.  out8_host is HOST memory; synchronises the device. */
int lpg_seal_phase_ticks(uint64_t* out8_host, int reset);
/* the same for the SEAL conv stack's 128-row windows, EIGHT values: [1] layers, [2] SortPooling epilogue, [3] windows,
 * prologue steps [4] bounds + offsets, [5] row table + tags / nodes, [7] table gathers + planes ([0], [6] unused)
 * (-DLPG_TUNING builds only; profiles/seal_conv_phases.py) */
int lpg_seal_conv_phase_ticks(uint64_t* out8_host, int reset);

/* Block-diagonal CSR of A~ = A + I over a batch, D^-1 = 1/rowsum(A~), and
 * X = [onehot(tag_map[tag], T) || emb[node]]  [SEAL/config/data.py:94-140].
 * node_off[B+1] = exclusive scan of n_nodes (host-computed or torch.cumsum);
 * edge capacity e_cap >= sum_b (nnz_b + n_b).  rowptr gets sum(n)+1 entries.
 * fold_dinv != 0: the stored values are D^-1 A~ (each row's scale folded into its edges), so the
 * SpMM needs no per-row scale load: tanh((D^-1 A~) H) == tanh(D^-1 (A~ H)). */
int lpg_seal_block_csr(const int32_t* n_nodes, const int64_t* node_off, const uint32_t* adj_bits,
                       int64_t B, int64_t* rowptr, int32_t* col, float* val, float* d_inv,
                       int64_t e_cap, int fold_dinv, void* workspace, int64_t workspace_bytes,
                       void* stream);
int64_t lpg_seal_block_csr_bytes(int64_t total_nodes);
int lpg_seal_features_f32(const int32_t* nodes, const int32_t* n_nodes, const int64_t* node_off,
                          const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len,
                          int64_t T, const float* emb, int64_t emb_dim, int64_t B, float* x,
                          int64_t ldx, int zero_pad, int32_t* err_flag, void* stream);
/* same, X stored as bfloat16 (the embedding half rounded to nearest even; configs[2]'s bf16 storage) */
int lpg_seal_features_bf16(const int32_t* nodes, const int32_t* n_nodes, const int64_t* node_off,
                           const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len,
                           int64_t T, const float* emb, int64_t emb_dim, int64_t B, uint16_t* x,
                           int64_t ldx, int zero_pad, int32_t* err_flag, void* stream);
/* zero_pad != 0: columns [T + emb_dim, ldx) of every row are written with zeros as well.  Rows padded to a
 * 16-byte multiple are then written as whole cache lines (a row with an unwritten tail turns every store
 * of the kernel into a read-modify-write: 2x slower), and the pad is safe to feed to a GEMM. */
/* Layer 1 of the SEAL stack without materialising X: H_1 = X W_1 with X = [onehot(tag) || emb[node]]
 * [SEAL/config/data.py:129-140, SEAL/model/gnn.py:46] is  H_1[g] = W_1[tag_g, :] + (emb W_1[T:, :])[node_g, :].
 * w1_tag = the first T rows of W_1 ([T,32], input index major), p_table = emb @ W_1[T:] ([n_graph,32], one small
 * lpg_linear_f32); both 16-byte aligned, 32 columns (the reference's layer width).  h1 [total, ldh >= 32]. */
int lpg_seal_h1_tables_f32(const int32_t* nodes, const int64_t* node_off, const int32_t* tags,
                           const int32_t* tag_map, int64_t tag_map_len, int64_t T, const float* w1_tag,
                           const float* p_table, int64_t n_graph, int64_t B, float* h1, int64_t ldh,
                           int32_t* err_flag, void* stream);
/* bf16 storage (configs[2]): the same two fp32 table gathers and add, the sum rounded ONCE to bf16 -- what
 * bf16(X W_1) with fp32 accumulation gives when X = [onehot || emb] and W_1 hold bf16 values (p_table = the fp32
 * product of the bf16-valued emb and W_1[T:]).  h1 rows 8-byte aligned. */
int lpg_seal_h1_tables_bf16(const int32_t* nodes, const int64_t* node_off, const int32_t* tags,
                            const int32_t* tag_map, int64_t tag_map_len, int64_t T, const float* w1_tag,
                            const float* p_table, int64_t n_graph, int64_t B, uint16_t* h1, int64_t ldh,
                            int32_t* err_flag, void* stream);

/* Layers 1..4 of the reference's fixed SEAL architecture (graph weights [F,32], [32,32], [32,32], [32,1];
 * SEAL/model/gnn.py:32-42,46-63) for a whole batch as dense 32 x 32 block products on the fp32 matrix cores,
 * every block of D^-1 (A + I) built in registers from the bit adjacency (the reference feeds the same dense
 * operands to tf.matmul, one subgraph per run).  Given h1 = X W_1 [total, 32] (lpg_linear_f32) it writes
 * cat = [Z_1 | Z_2 | Z_3] [total, >= 96] and key = Z_4[:, 0] [total]; no block CSR, no plan, no intermediate
 * in HBM.  first[w] (int32, n_windows + 1 entries) = the first subgraph whose first row is >= 128 w, i.e.
 * lower_bound(node_off, 128 w) clamped to B, for w = 0 .. ceil(total / 128); a workgroup owns the subgraphs that
 * start in its 128-row window.  w2, w3 are [32,32] with the INPUT index major (TensorFlow's x @ W), w4 is [32].
 * Head fusion: with head_k1 [96,16] (conv1d_kernel_1 of the DGCNN head, SEAL/model/gnn.py:104-110, a per-vertex
 * linear on the concat) the kernel also writes head_a16 [total,16] = [Z_1|Z_2|Z_3] @ head_k1; cat may then be NULL
 * (the 96-wide concat is never written).  Every product runs on the bf16 matrix cores: the aggregation through an
 * exact three-way bf16 split of H (0/1/2 adjacency entries are exact in bf16), the dense products H' = Z W and
 * a16 += Z k1 through the six leading plane pairs of the two splits (what is dropped is <= 2^-24 |z||w| per term: one
 * fp32 rounding).  Options: "seal_conv_variant" = 1 keeps the dense products on the fp32 matrix cores (the round-1
 * kernel), "seal_conv_fp32" = 1 the first all-fp32-MFMA formulation. */
/* The index of a batch of subgraphs in one launch: node_off [B + 1] = exclusive scan of the sizes n_nodes [B]
 * (the row offsets of the block-diagonal batch, SEAL/config/data.py:94-117 stacks per-graph arrays); tag_map
 * [n_tags] = rank of every PRESENT tag among the present ones, -1 for the others -- the global tag compaction of
 * SEAL/config/subgraph.py:84-101 (ascending tags -> 0 .. T-1) from the presence bytes lpg_seal_subgraphs filled
 * (tag_present may be NULL: no map); head [4] = {total vertices, *err_flag, T, largest present tag (0 without tag_present)} for the caller's single host
 * read.  first (may be NULL; capacity ceil(B * 128 / window_rows) + 1 entries is always enough, the number written
 * is ceil(total / window_rows) + 1) receives the row windows of lpg_seal_conv_windows in the same launch.
 * Workspace only for B > 2^14 (lpg_seal_batch_index_bytes). */
int64_t lpg_seal_batch_index_bytes(int64_t B);
int lpg_seal_batch_index(const int32_t* n_nodes, int64_t B, int64_t* node_off, const uint8_t* tag_present,
                         int64_t n_tags, int32_t* tag_map, const int32_t* err_flag, int64_t* head,
                         int32_t* first, int64_t window_rows, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* first[w], w = 0 .. n_windows (n_windows + 1 entries) for lpg_seal_conv_stack_f32: one launch instead of a
 * searchsorted / arange / clamp chain; window_rows <= 128. */
int lpg_seal_conv_windows(const int64_t* node_off, int64_t B, int64_t n_windows, int64_t window_rows,
                          int32_t* first, void* stream);
int lpg_seal_conv_stack_f32(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                            const uint32_t* adj_bits, const float* h1, int64_t ldh, const float* w2,
                            const float* w3, const float* w4, float* cat, int64_t ldc, float* key,
                            const float* head_k1, float* head_a16, void* stream);
/* The same stack in bf16 STORAGE (BASELINE configs[2]: "d = 128, bf16"): h1, the weights and cat are bf16, H and Z are
 * rounded to bf16 (nearest even) where the layer-by-layer bf16 path stores them, accumulation is fp32, key and
 * head_a16 stay fp32 (head_k1 is the head's fp32 kernel: its product with the bf16 Z is exact to fp32 rounding).
 * One bf16 plane instead of three: 2 MFMAs per block product, 17 KB of LDS per workgroup instead of 51.
 * h1 / cat rows 8-byte aligned (ldh, ldc multiples of 4). */
int lpg_seal_conv_stack_bf16(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                             const uint32_t* adj_bits, const uint16_t* h1, int64_t ldh,
                             const uint16_t* w2, const uint16_t* w3, const uint16_t* w4, uint16_t* cat,
                             int64_t ldc, float* key, const float* head_k1, float* head_a16,
                             void* stream);
/* The scoring form of the same stack with both ends fused (round 2): the conv1d #1 fusion is mandatory
 * (head_k1 [96,16]) and
 *  - h1 == NULL: layer 1 comes from the two tables of lpg_seal_h1_tables_* gathered in the kernel's prologue (nodes,
 *    tags [B,128] from lpg_seal_subgraphs; tag_map / T / w1_tag [T,32] / p_table [n_graph,32] as there) -- H_1 is
 *    neither written nor read back; bit for bit what lpg_seal_h1_tables_* + the h1 form computes;
 *  - pooled != NULL: SortPooling (SEAL/model/gnn.py:82-91; the total order of lpg_seal_sortpool_f32) of the 16-wide
 *    rows runs in the epilogue and the kernel writes pooled [B, pool_k, 16]; key / head_a16 may then be NULL.
 * Either fusion can be left out (h1 given / pooled NULL: key and head_a16 are written as by lpg_seal_conv_stack_*).
 * n_windows == -1: the number of row windows is ceil(node_off[n_subgraphs] / 128), read ON THE DEVICE (first from
 * lpg_seal_conv_windows(n_windows = -1), n_subgraphs + 2 entries): with pooled output nothing between the
 * extraction and the scores needs the batch's vertex total on the host -- no sync in a scoring forward.
 * err_flag: OR-ed with LPG_F_INDEX_RANGE when a tag or node of a row is out of range (the row is then zero). */
int lpg_seal_score_stack_f32(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                             const uint32_t* adj_bits, const float* h1, int64_t ldh, const int32_t* nodes,
                             const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len, int64_t T,
                             const float* w1_tag, const float* p_table, int64_t n_graph, const float* w2,
                             const float* w3, const float* w4, const float* head_k1, float* key,
                             float* head_a16, float* pooled, int64_t pool_k, int64_t n_subgraphs,
                             int32_t* err_flag, void* stream);
int lpg_seal_score_stack_bf16(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                              const uint32_t* adj_bits, const uint16_t* h1, int64_t ldh,
                              const int32_t* nodes, const int32_t* tags, const int32_t* tag_map,
                              int64_t tag_map_len, int64_t T, const float* w1_tag, const float* p_table,
                              int64_t n_graph, const uint16_t* w2, const uint16_t* w3, const uint16_t* w4,
                              const float* head_k1, float* key, float* head_a16, float* pooled,
                              int64_t pool_k, int64_t n_subgraphs, int32_t* err_flag, void* stream);

/* The DGCNN head's last dense layer and its softmax in one launch (SEAL/model/gnn.py:127-137): pre_y [B, 2] =
 * h [B, H] @ w [H, 2] + bias [2] (w as TensorFlow stores it, input index major), score [B] = softmax(pre_y)[:, 1]. */
int lpg_seal_logits_f32(const float* h, int64_t ldh, int64_t B, int64_t H, const float* w, const float* bias,
                        float* pre_y, float* score, void* stream);

/* ---------------------------------------------------------------- SEAL head
 * SortPooling [SEAL/model/gnn.py:82-91]: per subgraph b, order its n_b vertices by key (= Z_4[:,0])
 * descending, ties -> lower vertex index first (tf.nn.top_k), take the first k rows of z
 * (= concat(Z1,Z2,Z3), C columns) and zero-pad to k rows:  out[b, t, :], out is [B, k, C].
 * The 1-D convolutions and dense layers that follow (:103-137) are plain matmuls on this layout
 * and run on lpg_linear_f32.  pos_score: softmax(pre_y)[:, 1] for the 2-class logits (:137). */
int lpg_seal_sortpool_f32(const float* z, int64_t ldz, const float* key, const int64_t* node_off,
                          int64_t B, int64_t k, int64_t C, float* out, void* stream);
int lpg_softmax2_pos_score_f32(const float* pre_y, int64_t B, float* score, void* stream);

/* ---------------------------------------------------------------- RCCL: sum of the per-rank partial aggregates
 * The north star's "RCCL all-reduce of partial node aggregates over xGMI" at the C boundary (build-defined: the
 * reference has no distributed code).  One process per GPU.  librccl is resolved with dlopen("librccl.so.1") at first
 * use -- inside a PyTorch process that is the copy torch already loaded -- so the library has no link-time dependency
 * on it; lpg_comm_available() = 1 when it resolved.  id128: HOST memory, 128 bytes (ncclUniqueId): rank 0 creates it,
 * the caller ships it to the other ranks by any host channel, every rank calls lpg_comm_init with the device it owns
 * current; *comm (HOST pointer to a handle) is an ncclComm_t.  Collectives enqueue on `stream` and return; buffers
 * are device memory, fp32 or (is_bf16 != 0) bf16; `send` may equal `recv` for the all-reduce.
 * lpg_reduce_scatter_sum: send holds world * recv_count elements, rank r keeps the sum of every rank's r-th block --
 * the shard the next layer on the same source partition reads (half the all-reduce's bytes per link).
 * Returns LPG_E_UNSUPPORTED when RCCL cannot be loaded, LPG_E_LAUNCH with RCCL's message on a collective error.
 * The host mirror (lpgnn_amd.dist) drives the same collectives through torch.distributed, whose "nccl" backend IS
 * RCCL; RcclComm there wraps these entry points for callers without torch.distributed. */
int lpg_comm_available(void);
int lpg_comm_unique_id(void* id128);
int lpg_comm_init(void** comm, int world, int rank, const void* id128);
int lpg_comm_destroy(void* comm);
int lpg_allreduce_sum(void* comm, const void* send, void* recv, int64_t count, int is_bf16, void* stream);
int lpg_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recv_count, int is_bf16,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LPG_HIP_H */
