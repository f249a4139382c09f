/*
 * tgnn.h -- C ABI of libtgnn.so: the MI355X (gfx950) implementation of TilinGNN's
 * graph-conv scoring forward.
 *
 * The reference (xuhaocuhk/TilinGNN) has no FFI layer: its boundary for this path is the
 * Python call
 *     probs, *_ = network(x=, adj_e_index=, adj_e_features=, col_e_idx=, col_e_features=)
 *     (solver/ml_solver/ml_solver.py:39-43, graph_networks/network_utils.py:11-15)
 * and, one level down, the two PyTorch-Geometric layer calls
 *     self.nnConv(x, edge_index, edge_features)    (graph_networks/layers/edge_conv.py:25)
 *     self.ginConv(x, edge_index)                  (graph_networks/layers/coll_conv.py:25)
 * This header is what a maintainer binds (ctypes) to replace those calls.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all matrices are row-major fp32, indices at the API are int64 [2,E] exactly as the
 *     reference passes them (row 0 = source, row 1 = destination; util/data_util.py:110-117),
 *     internal CSR arrays are int32;
 *   - functions enqueue work on `stream` (a hipStream_t) and return without synchronising,
 *     except where the comment says "synchronises";
 *   - nothing is allocated that outlives a call: scratch comes from the caller as (ws, ws_bytes),
 *     sized by the matching *_workspace_bytes(); caller memory is never freed or resized;
 *   - return value: 0 = ok, <0 = error (TGNN_ERR_*); tgnn_last_error() returns a thread-local
 *     message for the last failing call on this thread;
 *   - functions are re-entrant; every call runs on the device its stream belongs to (the calling thread's current
 *     device is switched for the duration of the call and restored); the only state kept between calls is idempotent
 *     per-device set-up (the one-time opt-in of a kernel to > 64 KB of LDS, the reusable events of the two-stream
 *     schedule per thread and device).
 */
#ifndef TGNN_H
#define TGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGNN_VERSION 100 /* 0.1.0 */

#define TGNN_OK 0
#define TGNN_ERR_INVALID_ARG (-1)
#define TGNN_ERR_WORKSPACE (-2)   /* workspace too small */
#define TGNN_ERR_LAUNCH (-3)      /* HIP launch / runtime failure */
#define TGNN_ERR_UNSUPPORTED (-4) /* shape outside what the kernels are built for */
#define TGNN_ERR_BAD_GRAPH (-5)   /* edge index out of [0, N) */
#define TGNN_ERR_UNVERIFIED (-7)   /* tgnn_graph.nn_mid_verdict given where the forward cannot honour it: nothing was queued */
#define TGNN_ERR_STALE_RESULT (-6) /* an EARLIER forward's persistent kernel gave up and nobody collected it: see tgnn_spin_error_poll */

/* activation codes (torch.nn.LeakyReLU() slope 0.01 / torch.nn.Sigmoid()) */
#define TGNN_ACT_NONE 0
#define TGNN_ACT_LEAKY_RELU 1
#define TGNN_ACT_SIGMOID 2

/* BatchNorm statistics record produced by tgnn_bn_finalize and consumed by the kernels that
 * apply the normalisation while loading:  y = ((v - mean_hi) - mean_lo) * ginv + beta,
 * ginv = gamma / sqrt(var_biased + eps).  Layout: float [4][F] = mean_hi, mean_lo, ginv, beta. */
#define TGNN_BN_STAT_ROWS 4
/* per-block partial sums: double [P][2][F] = sum(v), sum(v*v);  P <= TGNN_BN_MAX_PARTIALS */
#define TGNN_BN_MAX_PARTIALS 512

typedef void *tgnn_stream_t; /* hipStream_t */

int tgnn_version(void);
const char *tgnn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Graph preparation (once per layout; the graph is constant across the 20 layers of a forward)
 * ------------------------------------------------------------------------------------------ */

/* CSR-by-destination of an edge_index [2,E] (int64).  Rows keep the ORIGINAL edge order (the
 * order torch's CPU index_add_ accumulates in), so sums are reproducible run to run.
 *   rowptr   int32 [N+1]
 *   col_src  int32 [E]   source node of each CSR slot
 *   col_eid  int32 [E]   original edge number of each CSR slot
 * drop_self_loops != 0 removes (v,v) edges (GINConv semantics, PyG gin_conv.py); the number of
 * kept edges is rowptr[N].  Destinations must lie in [0, n_nodes), sources in [0, n_src_nodes)
 * (n_src_nodes = n_nodes for a whole graph; a node-range shard also gathers from halo rows stored
 * behind its own, so n_src_nodes = own + halo).  err_flag (int32 device word, may be NULL) is set
 * to 1 when an index is out of range; such edges are skipped. */
size_t tgnn_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int tgnn_csr_build(const int64_t *edge_index, int64_t n_edges, int64_t n_nodes, int64_t n_src_nodes,
                   int drop_self_loops, int32_t *rowptr, int32_t *col_src, int32_t *col_eid, int32_t *err_flag,
                   void *ws, size_t ws_bytes, tgnn_stream_t stream);

/* Exact de-duplication of edge-attribute rows (K1 of SURVEY.md: the per-edge NNConv weight depends
 * only on the attribute row; real layouts carry a small codebook -- 13 rows on the labyrinth set).
 *   edge_type      int32 [E]  type id of every edge, ids numbered by first occurrence
 *   type_rep_edge  int32 [E]  first n_types entries: an edge carrying that type's row
 *   n_types        int32 device word
 * Rows are compared bit-exactly after mapping -0.0 to +0.0. */
size_t tgnn_edge_dedup_workspace_bytes(int64_t n_edges, int32_t fe);
int tgnn_edge_type_dedup(const float *edge_attr, int64_t n_edges, int32_t fe, int32_t *edge_type,
                         int32_t *type_rep_edge, int32_t *n_types, void *ws, size_t ws_bytes,
                         tgnn_stream_t stream);

/* out[i] = src[idx[i]] for int32 arrays (used to bring edge types into CSR order); an index outside
 * [0, n_src) yields 0 (CSR slots beyond rowptr[N] are unspecified when edges were skipped). */
int tgnn_gather_i32(const int32_t *src, int64_t n_src, const int32_t *idx, int64_t n, int32_t *out,
                    tgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-op entry points (the layer seams of the reference)
 * ------------------------------------------------------------------------------------------ */

/* GraphConv's edge MLP on the T representative rows (edge_conv.py:17-18: Fe->32->64->C*C, Sigmoid
 * on all three layers):  wtab[t] = sigmoid(L3(sigmoid(L2(sigmoid(L1(edge_attr[type_rep_edge[t]])))))),
 * stored [T][C][C] with flat index i*C+o exactly as NNConv's .view(-1, C_in, C_out).
 * w1 [32,Fe] b1 [32] w2 [64,32] b2 [64] w3 [C*C,64] b3 [C*C]  (nn.Linear layout [out,in]). */
int tgnn_edge_weight_table(const float *edge_attr, const int32_t *type_rep_edge, int32_t n_types, int32_t fe,
                           const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *w3, const float *b3, int32_t c, float *wtab, tgnn_stream_t stream);

/* NNConv(aggr="mean") + optional LeakyReLU (edge_conv.py:25-27):
 *   out[v] = act( (1/max(deg_v,1)) * sum_{e: dst_e = v} h[src_e] . wtab[type_e]  +  h[v] . root + bias )
 * h [N, C] with row stride ldh (floats); root [C_in, C_out]; out [N, C] dense.
 * bn_partial (may be NULL): per-block column sums of the OUTPUT for the train-mode BatchNorm that
 * follows (edge_conv.py:28-29); *n_partials_host receives the number of partial rows written. */
int tgnn_nnconv_mean_fwd(const float *h, int64_t ldh, const int32_t *rowptr, const int32_t *col_src,
                         const int32_t *col_type, const float *wtab, int32_t n_types, const float *root,
                         const float *bias, int64_t n_nodes, int32_t c, int32_t act, float *out,
                         double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream);

/* The same NNConv on matrix cores, over the type-column structure built by tgnn_nnconv_cols_build
 * (network_width 32 and at most tgnn_nnconv_cols_max_types() edge types; otherwise use
 * tgnn_nnconv_mean_fwd, which handles any width / type count).  Production path of tgnn_forward.
 *
 * The sum over edges moves inside the product: sum_e h[src_e] . W_type(e) = sum_t (sum_{e of type t} h[src_e]) . W_t,
 * one dense [16 x 32(T+1)] x [32(T+1) x 32] product per 16 destination rows whose accumulator is the output tile.
 *
 * Column structure (adjacency set, once per layout): every 16 destination rows form a tile with a list of
 * columns sorted by edge type; column (t, r) holds, for each of the 16 rows, the source of the row's r-th
 * in-edge of type t in original edge order, or -1.  The last column of a tile is the root column (type T):
 *   tile_col_ptr int32 [ceil(N/16)+1]   column range of every tile
 *   col_meta     int32 [n_cols]         type | first-of-type << 8 | last-of-type << 9 | end-of-tile << 10
 *   col_src      int32 [16*n_cols]      source row per tile row, -1 = none; in a root column the float bits of
 *                                       max(in-degree, 1), or -1 for rows >= N
 * n_cols <= tgnn_nnconv_cols_max_columns(N, E) (allocate col_meta / col_src for that many: the kernel reads
 * index words a few columns past the end); the exact count is tile_col_ptr[ceil(N/16)].
 * Source rows must lie within 2 GB of h (buffer addressing): N_src * ldh * 4 < 2^31. */
int64_t tgnn_nnconv_cols_max_columns(int64_t n_nodes, int64_t n_edges);
size_t tgnn_nnconv_cols_workspace_bytes(int64_t n_nodes);
int tgnn_nnconv_cols_build(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type,
                           int64_t n_nodes, int32_t n_types, int32_t *tile_col_ptr, int32_t *col_meta,
                           int32_t *col_slot_src, void *ws, size_t ws_bytes, tgnn_stream_t stream);
int32_t tgnn_nnconv_cols_max_types(void);
/* wimg_scratch: tgnn_nnconv_weight_image_floats(T) floats of caller scratch (the [T][C][C] table and
 * the root matrix are re-laid out into MFMA operand order there before the main kernel runs). */
size_t tgnn_nnconv_weight_image_floats(int32_t n_types);
int tgnn_nnconv_mean_cols_fwd(const float *h, int64_t ldh, const int32_t *tile_col_ptr, const int32_t *col_meta,
                              const int32_t *col_src, const float *wtab, int32_t n_types, const float *root,
                              const float *bias, int64_t n_nodes, int32_t c, int32_t act, float *out,
                              float *wimg_scratch, double *bn_partial, int32_t *n_partials_host,
                              tgnn_stream_t stream);
/* The same kernel with the fp16 x 2 split (tgnn_set_split_precision: three matrix terms instead of six) as tgnn_forward runs
 * it, as one op for tests: the bounds tgnn_forward gets from the kernels that produce h are computed here (largest |h| over
 * the n_src_rows >= n_nodes rows of h, rows packed: ldh == 32; largest |root|) into bounds_scratch (2 words of device scratch);
 * max_in_degree >= the layout's largest in-degree (a larger value only lowers the scale: the result must not move).
 * width 32 only. */
int tgnn_nnconv_mean_cols_f16_fwd(const float *h, int64_t ldh, int64_t n_src_rows, const int32_t *tile_col_ptr,
                                  const int32_t *col_meta, const int32_t *col_src, const float *wtab, int32_t n_types,
                                  const float *root, const float *bias, int64_t n_nodes, int32_t max_in_degree, int32_t act,
                                  float *out, float *wimg_scratch, uint32_t *bounds_scratch, double *bn_partial,
                                  int32_t *n_partials_host, tgnn_stream_t stream);

/* The same NNConv (edge_conv.py:24-27) over EDGE GROUPS instead of type columns (csrc/nnconv_eg.hip): the in-edges of a 16-row
 * tile sorted by (type, destination row, original order) and cut into groups of up to 16 edges of one type, so that a gather
 * instruction of the kernel fetches 16 source rows whatever rows of the tile they belong to (a type column is ~30 % full on
 * real layouts, a group ~70 %); the rows of a group are multiplied by the type's matrix and folded into their destination
 * rows by a second matrix product with the group's 0 / 1 selection matrix.  The last group of a tile is the root group.
 *   tile_grp_ptr int32 [ceil(N/16)+1]   group range of every tile
 *   grp          int32 [2*16*n_groups]  (src, sm) pairs, 8-byte aligned.  src of slot k: source row, -1 = none; root group:
 *                                       float bits of max(in-degree, 1) of row k, -1 for rows >= N.  sm of word j of a
 *                                       group: mask of the slots that end in row j | (type | root << 8) << 16
 * n_groups <= tgnn_nnconv_eg_max_groups(N, E, T) (allocate for that many: the kernel reads index words past the end); workspace
 * of the build: tgnn_nnconv_cols_workspace_bytes(N).  tgnn_nnconv_mean_eg_fwd is the op for tests (bounds and the fp16-pair
 * weight image computed inside, as tgnn_nnconv_mean_cols_f16_fwd does); width 32, packed rows (ldh == 32). */
int64_t tgnn_nnconv_eg_max_groups(int64_t n_nodes, int64_t n_edges, int32_t n_types);
int tgnn_nnconv_eg_build(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type, int64_t n_nodes,
                         int32_t n_types, int32_t *tile_grp_ptr, int32_t *grp, void *ws, size_t ws_bytes, tgnn_stream_t stream);
int tgnn_nnconv_mean_eg_fwd(const float *h, int64_t ldh, int64_t n_src_rows, const int32_t *tile_grp_ptr,
                            const int32_t *grp, const float *wtab, int32_t n_types,
                            const float *root, const float *bias, int64_t n_nodes, int32_t act, float *out,
                            float *wimg_scratch, uint32_t *bounds_scratch, double *bn_partial, int32_t *n_partials_host,
                            tgnn_stream_t stream);

/* Measurement helper of bench.py (roofline.gather_bound; csrc/ubench.hip): rows of 128 bytes gathered per second by one launch
 * over every CU from an L2 / Infinity-Cache resident table of n_rows x 128 bytes (>= 8192 rows), band-local random rows, best of
 * `reps` launches of `iters` x 8 gather steps per wave.  shape 0 = the column NNConv's lane map (16 rows x 64 bytes per
 * instruction), 1 = whole rows (8 lanes x 16 bytes: the GIN aggregate's).  sink: 256 x 1024 floats.  Synchronises the stream. */
int tgnn_ubench_row_gather(int32_t shape, const float *table, int64_t n_rows, float *sink, int32_t iters, int32_t reps,
                           double *rows_per_s_out, tgnn_stream_t stream);

/* NNConv BATCHES of the mid-size persistent layer loop (csrc/forward_mid.hip; layouts of 4 097 .. tgnn_mid_layout_max_nodes()
 * nodes run TilinGNN.py:59-71 as ONE kernel).  Built from the type-column structure, once per layout:
 *   tile_nb int32  [ceil(N/16)]                                            batches of every 16-row tile (<= TGNN_MID_TILE_BATCHES)
 *   ent     uint32 [ceil(N/16)][TGNN_MID_TILE_BATCHES][TGNN_MID_BATCH_WORDS]  per tile its edge types in order; a batch = up to 32
 *           in-edges of ONE type: words 0-3 = {type | last-batch-of-the-type << 8, mask of the tile's rows that have an edge of
 *           the type, 0, 0}, word 4 + 4 o + g = slot o (0..7) of gather instruction g (0..3):
 *           source row (24 bits) | destination row (0..15) << 25 | add << 30; an empty slot is 0x21000000 (source 0x1000000 = none,
 *           destination 16 = the kernel's spare row).  Eight whole 128-byte source rows per gather
 *           instruction, never two edges of one destination row in one instruction: a row's first edge of a type STORES its
 *           type-sum slot, further ones (add) read-add-write it, in CSR (= original edge) order.
 * result (device int32 [2], zeroed by the caller): [0] most batches of a tile, [1] 1 = a tile does not fit (more than
 * TGNN_MID_TILE_BATCHES batches / 64 columns) or no column structure -- the layout then takes the general schedule.
 * cols_built_dev (may be NULL): device word, 0 = the column structure was not built (tgnn_graph_prep: result + 5). */
#define TGNN_MID_BATCH_WORDS 36
#define TGNN_MID_TILE_BATCHES 24
int64_t tgnn_mid_entries_words(int64_t n_nodes);
int tgnn_mid_entries_build(const int32_t *tile_col_ptr, const int32_t *col_meta, const int32_t *col_src, int64_t n_nodes,
                           const int32_t *cols_built_dev, int32_t *tile_nb, uint32_t *ent, int32_t *result, tgnn_stream_t stream);
/* Layouts of more than 4 096 and up to this many nodes (default 32 768, maximum 65 536; 0 = off) run the layer loop as the
 * persistent mid-size kernel when the graph carries the batches (nn_mid_*), width 32, train-mode BatchNorm, single device. */
/* Forwards queued so far by this process on {the general launch schedule, the small-layout kernel, the mid-size kernel}. */
void tgnn_forward_path_counts(int64_t *out3);
void tgnn_set_mid_layout_limit(int64_t n_nodes);
int64_t tgnn_get_mid_layout_limit(void);
int64_t tgnn_mid_layout_max_nodes(void);

/* GINConv + optional LeakyReLU (coll_conv.py:25-27):
 *   z[v]  = (1+eps) * f(a[v]) + sum_{e: dst_e = v} f(a[src_e]),   f = identity or the BatchNorm
 *           described by in_stat (lets layer i+1 read layer i's pre-BN activations directly);
 *   out[v] = act( sigmoid(L3(sigmoid(L2(sigmoid(L1(z[v])))))) )   L1 [32,C] L2 [64,32] L3 [C,64].
 * eps is the device buffer ginConv.eps [1].  z_scratch: [N, C] floats of caller scratch (the
 * aggregate z travels through it between the gather kernel and the MFMA MLP kernel). */
int tgnn_gin_fwd(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr,
                 const int32_t *col_src, const float *eps, const float *w1, const float *b1,
                 const float *w2, const float *b2, const float *w3, const float *b3, int64_t n_nodes,
                 int32_t c, int32_t act, float *out, float *z_scratch, double *bn_partial,
                 int32_t *n_partials_host, tgnn_stream_t stream);

/* Linear_trans (layers/util.py:31-37) without its BatchNorm:  out = act(f(A) . W^T + b).
 * A is given as ceil(in_dim/32) column blocks of 32: element (r, k) lives at
 * a[(k/32)*a_kblock_stride + r*lda + k%32]  (row-major A: lda = in_dim, a_kblock_stride = 32;
 * the [D+1][N][C] skip-connection buffer: lda = C = 32, a_kblock_stride = N*C).
 * in_stat (may be NULL): BatchNorm of the PREVIOUS Linear_trans applied to A while loading.
 * w [out_dim, in_dim]; out [N, out_dim] with row stride ldo. */
int tgnn_dense_act_fwd(const float *a, int64_t lda, int64_t a_kblock_stride, const float *in_stat,
                       const float *w, const float *b, int64_t n_rows, int32_t in_dim, int32_t out_dim,
                       int32_t act, float *out, int64_t ldo, double *bn_partial, int32_t *n_partials_host,
                       tgnn_stream_t stream);
/* The same Linear block over a SLOT-MAJOR input [S][N][slot_width] (the skip-connection buffer read as the
 * concatenation of its S slots, TilinGNN.py:74: in_dim = S * slot_width, feature k of row r lives at
 * a[(k / slot_width) * slot_stride + r * slot_width + k % slot_width]); slot_width a multiple of 32. */
int tgnn_dense_act_slots_fwd(const float *a, int32_t slot_width, int64_t slot_stride, const float *in_stat,
                             const float *w, const float *b, int64_t n_rows, int32_t in_dim, int32_t out_dim,
                             int32_t act, float *out, int64_t ldo, double *bn_partial, int32_t *n_partials_host,
                             tgnn_stream_t stream);
/* The same (no input BatchNorm, slots of 32 channels with packed rows, out_dim >= 64) with the fp16 x 2 split as tgnn_forward runs
 * it (tgnn_set_split_precision), as one op for tests: the bounds tgnn_forward gets from the kernels that fill the slots are
 * computed here into bounds_scratch (in_dim / 32 + 1 words of device scratch).  wimg_scratch (4 * in_dim * out_dim bytes of
 * device scratch, 16-byte aligned; out_dim 64 / 128 / 256): the rows-per-wave kernel tgnn_forward runs on the final MLP (W as
 * a pre-split operand image, every wave 32 rows x all columns); NULL: the block-tile kernel. */
int tgnn_dense_act_slots_f16_fwd(const float *a, int32_t slot_width, int64_t slot_stride, const float *w, const float *b,
                                 int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act, float *out, int64_t ldo,
                                 uint32_t *bounds_scratch, void *wimg_scratch, double *bn_partial, int32_t *n_partials_host,
                                 tgnn_stream_t stream);

/* Train-mode BatchNorm1d statistics (fact 2 of SURVEY.md: the reference never leaves train mode).
 * mode 0: partials -> stat (+ running stats)      single GPU
 * mode 1: partials -> sums  (double [2][F])        multi GPU, before the all-reduce
 * mode 2: sums     -> stat (+ running stats)      multi GPU, after the all-reduce
 * n_rows_total = number of rows the statistics cover (all ranks).  running_mean / running_var /
 * num_batches_tracked (int64) may be NULL; otherwise they are updated as torch does (momentum,
 * unbiased variance). */
int tgnn_bn_finalize(int32_t mode, const double *partials, int32_t n_partials, double *sums, int32_t f,
                     int64_t n_rows_total, const float *gamma, const float *beta, float eps, float momentum,
                     float *running_mean, float *running_var, int64_t *num_batches_tracked, float *stat,
                     tgnn_stream_t stream);

/* y = BatchNorm(v) for a dense [N, F] matrix given its stat record (used by the per-layer modules
 * and tests; the fused forward never materialises this). */
int tgnn_bn_apply(const float *v, int64_t ldv, const float *stat, int64_t n_rows, int32_t f, float *out,
                  int64_t ldo, tgnn_stream_t stream);

/* Branch merge (TilinGNN.py:64-71):  out = BN1(a1) * BN2(a2) [+ resid];  optionally also writes
 * h2 = BN2(a2) (NULL to skip).  All [N, C] dense. */
int tgnn_merge_fwd(const float *a1, const float *stat1, const float *a2, const float *stat2,
                   const float *resid, int64_t n_nodes, int32_t c, float *out, float *h2_out,
                   tgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole forward: TilinGNN.forward (graph_networks/networks/TilinGNN.py:51-78)
 * ------------------------------------------------------------------------------------------ */
typedef struct tgnn_model_dims {
    int32_t node_features_dim;     /* Fx = tile_count + 1            (TilinGNN.py:19) */
    int32_t adj_edge_features_dim; /* Fe                             (TilinGNN.py:15) */
    int32_t network_width;         /* C                              (inputs/config.py:38) */
    int32_t network_depth;         /* D                              (inputs/config.py:37) */
    int32_t output_dim;            /* 1                              (TilinGNN.py:18) */
} tgnn_model_dims;

/* Number of entries of the `params` pointer table for a model, and the key (state-dict name) of
 * entry i.  The table is a HOST array of DEVICE pointers in exactly this order. */
int32_t tgnn_param_count(const tgnn_model_dims *dims);
int tgnn_param_name(const tgnn_model_dims *dims, int32_t index, char *buf, size_t buf_len);

/* A prepared graph: device arrays owned by the caller (typically one torch buffer carved up). */
typedef struct tgnn_graph {
    int64_t n_nodes;
    int64_t n_adj_edges;        /* CSR slots of the adjacency set (= Ea) */
    int64_t n_col_edges;        /* CSR slots of the collision set after self-loop removal */
    int32_t n_types;            /* T */
    const int32_t *adj_rowptr;  /* [N+1] */
    const int32_t *adj_src;     /* [Ea]  */
    const int32_t *adj_type;    /* [Ea]  CSR order */
    const int32_t *type_rep_edge; /* [T] original edge numbers */
    const int32_t *col_rowptr;  /* [N+1] */
    const int32_t *col_src;     /* [Ec'] */
    /* NNConv type-column structure (all NULL => tgnn_forward uses the CSR kernel) */
    const int32_t *nn_tile_col_ptr;
    const int32_t *nn_col_meta;
    const int32_t *nn_col_src;
    /* largest adjacency in-degree of a node, if the caller knows it (tilingnn_amd.ops.prepare_graph reads it back together
     * with the type count); 0 = unknown.  The small-layout kernel (tgnn_set_small_layout_limit) keeps a row's gather list in
     * registers and runs only when 1 <= nn_max_in_degree <= 31; otherwise the general schedule does. */
    int32_t nn_max_in_degree;
    /* NNConv batches of the mid-size persistent layer loop (tgnn_mid_entries_build; both NULL => the general schedule). */
    const int32_t *nn_mid_tile_nb;
    const uint32_t *nn_mid_ent;
    /* NNConv edge-group structure (tgnn_nnconv_eg_build; both NULL => the type columns, or the CSR kernel, are used).  The
     * general schedule of the fp32 width-32 forward prefers it (tgnn_set_nnconv_eg; in-degrees up to 2048). */
    const int32_t *nn_tile_grp_ptr;
    const int32_t *nn_grp;
    /* [r6] NULL, or: a device word that is non-zero when nn_mid_tile_nb / nn_mid_ent turned out not to be usable (tgnn_graph_prep's
     * result word 9, final behind its last launch).  A forward queued BEHIND the preparation without waiting for that word passes
     * it: the mid-size persistent kernels read it first and leave without touching anything -- no output, no running-statistics
     * update -- when it is set; the caller, who reads the word later, then runs the forward again without the batches.  Where the
     * mid-size forward is not those two kernels alone (its init MLP or final MLP as separate launches), a forward given this
     * pointer returns TGNN_ERR_UNVERIFIED and queues nothing: wait for the word, pass NULL. */
    const int32_t *nn_mid_verdict;
} tgnn_graph;

size_t tgnn_forward_workspace_bytes(const tgnn_model_dims *dims, int64_t n_nodes, int32_t n_types);

/* probs [N, output_dim].  BatchNorm runs with batch statistics and updates the running buffers
 * in `params` when update_running != 0 (train mode); with use_running_stats != 0 it normalises
 * with the running buffers instead (eval mode; never used by the reference's solver).
 * update_running is a bit field: bit 0 the update; bit 1 (general schedule, batch statistics) "the init MLP's two BatchNorms have
 * this forward's update already" -- for the caller whose tgnn_forward_begin could not be resumed (a layout with more than 16 edge
 * types needs a larger workspace than begin carved) and who runs the whole forward again: ONE update per forward.
 * With stream2 != NULL (and != stream) the collision branch -- a chain of its own, CollConv_i reads only
 * CollConv_{i-1} (TilinGNN.py:63) -- is enqueued on stream2 and runs free beside the adjacency branch; the two
 * meet at the product of :64 through events the library records itself, so on return all work is ordered on
 * `stream` and the caller needs no synchronisation of its own with stream2. */
int tgnn_forward(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                 const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                 int32_t use_running_stats, float *probs, void *ws, size_t ws_bytes,
                 tgnn_stream_t stream, tgnn_stream_t stream2);

/* [r6] The forward in two calls around the layout's preparation (ML_Solver.predict on a NEW layout: util/data_util.py:110-117 then
 * networks/TilinGNN.py:51-78).  tgnn_forward_begin queues, on stream2 and ordered behind what `stream` holds so far, everything the
 * general schedule does in front of its first layer that needs nothing of the graph -- the operands' bounds, the init MLP
 * (middle[0]), the final MLP's bounds and operand images -- so that it runs BESIDE the preparation's launches (tgnn_graph_prep on
 * `stream`) instead of behind them; tgnn_forward_resume is tgnn_forward (train-mode BatchNorm) picking that work up.  Same
 * workspace (sized by tgnn_forward_workspace_bytes with any type count up to 16: the pieces begin fills do not move with it),
 * same thread, same node count; the result is bit-identical to tgnn_forward's.
 * stream == NULL in tgnn_forward_begin: stream2 is NOT ordered behind `stream` by the call -- the caller has done that at the point
 * where x and the parameters were ready (the host mirror queues the preparation's launches on `stream` first, then begin's).
 * tgnn_forward_begin returns TGNN_ERR_UNSUPPORTED (and queues nothing) where it does not apply -- layouts of the persistent
 * schedules, widths other than 32, more than 8 node features, no side stream -- : call tgnn_forward then.  A layout that
 * turns out not to take the fp16-pair path (more than 16 edge types, in-degree above 2 048) is handled by resume itself. */
int tgnn_forward_begin(const tgnn_model_dims *dims, const void *const *params_host, const float *x, int64_t n_nodes,
                       int32_t update_running, void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2);
int tgnn_forward_resume(const tgnn_model_dims *dims, const void *const *params_host, const float *x, const float *adj_edge_attr,
                        const tgnn_graph *graph, int32_t update_running, float *probs, void *ws, size_t ws_bytes,
                        tgnn_stream_t stream, tgnn_stream_t stream2);
/* [r6] Small layouts (<= tgnn_get_small_layout_limit() nodes: the forward is one persistent kernel behind a pre-pass): the pre-pass
 * -- edge weights with their operand images, parameter pack -- queued on `stream` BEHIND tgnn_graph_prep_small and before the host
 * has its result words; the type count is read on the device (n_types_dev = the preparation's `result`, word 0; type_rep_edge:
 * its output of that name).  ws as for tgnn_forward with n_types = 0 (the layout does not depend on the count up to 16).  The
 * next tgnn_forward of this thread with the same ws and node count uses it if it takes the small-layout path (otherwise it queues
 * its own work: nothing is lost but the launches).  TGNN_ERR_UNSUPPORTED (nothing queued) where it does not apply. */
int tgnn_forward_small_prepass(const tgnn_model_dims *dims, const void *const *params_host, const float *adj_edge_attr,
                               const int32_t *type_rep_edge, const int32_t *n_types_dev, int64_t n_nodes, void *ws, size_t ws_bytes,
                               tgnn_stream_t stream);

/* [r6] Optional, between the two: queued on `stream` BEHIND tgnn_graph_prep's launches and before the host has read the type count
 * -- the edge weights and the edge-group NNConv's operand images of all layers (the first launch tgnn_forward_resume would queue),
 * with the count read on the device from n_types_dev (= tgnn_graph_prep's `result`, word 0; more than 16 types: the launch writes
 * nothing and tgnn_forward_resume, which knows the count, queues its own).  type_rep_edge: tgnn_graph_prep's output of that name.
 * TGNN_ERR_UNSUPPORTED when there is no matching tgnn_forward_begin or the forward would not take this launch (nothing queued). */
int tgnn_forward_begin_weights(const tgnn_model_dims *dims, const void *const *params_host, const float *adj_edge_attr,
                               const int32_t *type_rep_edge, const int32_t *n_types_dev, int64_t n_nodes, void *ws, size_t ws_bytes,
                               tgnn_stream_t stream);

/* ---- the same forward for the TRAINING step (Trainer.train, solver/ml_solver/trainer.py:68-75: the network in train mode
 * with autograd recording): identical kernels and schedule, but what the backward reads is kept in the caller's buffers
 * instead of the rotating workspace ones.  C = network_width, D = network_depth, T = graph->n_types. */
typedef struct tgnn_train_save {
    float *init_a[2];    /* [N, C] x 2: pre-BatchNorm activations of the init MLP's two layers */
    float *init_stat[2]; /* [4, C] x 2: their BatchNorm records */
    float *a1, *a2;      /* [D][N][C]: LeakyReLU(conv) of the adjacency / collision branch, before BatchNorm */
    float *u;            /* [D][N][C]: the aggregate GINConv's MLP read */
    float *stat1, *stat2;/* [D][4][C]: the two BatchNorm records of every layer */
    float *fin_a[4];     /* [N, 256], [N, 128], [N, 64], [N, C]: pre-BatchNorm activations of the final MLP */
    float *fin_stat[4];  /* [4, 256], [4, 128], [4, 64], [4, C] */
    float *skip;         /* [D + 1][N][C]: the skip-connection buffer = every layer's input */
    float *wtab;         /* [D][max(T, 1)][C][C]: the per-edge-type NNConv matrices */
} tgnn_train_save;
int tgnn_forward_train(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                       const float *adj_edge_attr, const tgnn_graph *graph, const tgnn_train_save *keep, float *probs,
                       void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2);

/* The backward of the training step (loss.backward() of trainer.py:79) as ONE call: the schedule of tilingnn_amd/train.py
 * (backward_train: final MLP, D x (merge / BatchNorm / NNConv / GIN adjoints), init MLP) enqueued from the library.
 * tgraph: what only the backward reads; keep: the buffers tgnn_forward_train filled; grads_host: one device pointer per
 * entry of params_host (same indexing, tgnn_param_name); entries of buffers (running statistics, GIN's eps) are ignored.
 * Width 32, at most 63 edge types. */
typedef struct tgnn_train_graph {
    const int32_t *adjT_rowptr, *adjT_src, *adjT_type; /* CSR of the TRANSPOSED adjacency edges + the type of every slot */
    const int32_t *colT_rowptr, *colT_src;             /* CSR of the transposed collision edges, self loops dropped */
    const float *deg, *inv_deg;                        /* max(in-degree, 1) of the adjacency graph, and 1 / it */
} tgnn_train_graph;
size_t tgnn_backward_workspace_bytes(const tgnn_model_dims *dims, int64_t n_nodes, int32_t n_types);
int tgnn_backward(const tgnn_model_dims *dims, const void *const *params_host, void *const *grads_host, const float *x,
                  const float *adj_edge_attr, const tgnn_graph *graph, const tgnn_train_graph *tgraph,
                  const tgnn_train_save *keep, const float *probs, const float *dprobs, void *ws, size_t ws_bytes,
                  tgnn_stream_t stream);

/* ---- the same forward for ONE SHARD of a node-range partition (one process per GPU) -------------------------
 * This device owns rows [0, n_own) of a layout whose buffers carry n_rows - n_own halo rows of other shards behind
 * them: `graph` is built with n_nodes = n_own destinations and sources in [0, n_rows) (tgnn_csr_build's
 * n_src_nodes), `x` has n_own rows.  Per layer the library needs two collectives.  Either it issues them itself over RCCL
 * (rccl_comm below: ncclSend / ncclRecv groups and ncclAllReduce on the kernels' stream, no host code in the loop), or they
 * stay with the caller as callbacks (torch.distributed in tilingnn_amd/dist.py, thread-simulated ranks in the tests): the
 * library enqueues its kernels, calls back on the host, and continues; a callback must enqueue its collective on the
 * stream it is handed or order it with that stream.
 *   allreduce_f64(ctx, sum_buf, count, stream): sum_buf[0..count) <- sum over all shards      (BatchNorm sums)
 *   alltoall_rows(ctx, send_buf, recv_buf, row_floats, extra_rows, stream): send_buf holds n_send rows of row_floats floats
 *       (the owned rows listed in send_idx, grouped by destination shard); recv_buf must receive the n_rows - n_own
 *       halo rows in the order they sit behind the owned rows.  row_floats is C for the first exchange, 2 C after.
 * Both return 0 on success.  Train mode only (batch statistics over all n_total rows of the partition).
 *
 * Fused mode (world >= 1 and the two *_fused index arrays given; width 32): per message-passing layer ONE all-to-all
 * instead of an all-reduce plus an all-to-all.  Every shard sends to EVERY peer the raw rows of both branches the
 * peer keeps as halo, followed by 4 rows (256 floats = 128 doubles) with its local BatchNorm sums; the receiver adds
 * the sums of all shards in rank order (identical statistics everywhere) and merges its halo rows itself.
 *   send_idx_fused [n_send + 4 world]: per destination rank its rows of send_idx, then -1, -2, -3, -4
 *   recv_idx_fused [n_halo + 4 world]: per source rank the halo slots 0.. it fills, then -1 - (4 p + k), k = 0..3
 * alltoall_rows is then called with extra_rows = 4: the per-peer row counts are the plain ones + 4. */
typedef struct tgnn_shard {
    int64_t n_own, n_rows, n_total;
    const int32_t *send_idx;        /* device, [n_send] local row numbers */
    int64_t n_send;
    double *sum_buf;                /* device, >= max(1024, 128 (world + 2)) doubles */
    float *send_buf;                /* device, >= max(n_send + 4 world, 1) * 2 C floats */
    float *recv_buf;                /* device, >= max(n_rows - n_own + 4 world, 1) * 2 C floats */
    int (*allreduce_f64)(void *ctx, double *buf, int64_t count, tgnn_stream_t stream);
    int (*alltoall_rows)(void *ctx, const float *send, float *recv, int32_t row_floats, int32_t extra_rows,
                         tgnn_stream_t stream);
    void *ctx;
    int32_t world, rank;            /* fused mode */
    const int32_t *send_idx_fused;  /* device, may be NULL = plain mode */
    const int32_t *recv_idx_fused;  /* device */
    tgnn_stream_t side_stream;      /* may be NULL.  Not NULL (and != stream), fused mode: SPLIT exchange -- the collision branch
                                     * of a layer (GIN, then an all-to-all of its own: halo rows of that branch + its BatchNorm
                                     * sums, C floats per row) runs on it, ahead of the adjacency branch's chain (NNConv, its
                                     * all-to-all, merge) on `stream`: two all-to-alls per layer, each handed the stream it belongs
                                     * to and its halves of send_buf / recv_buf.  Plain mode: only the neighbourhood sum of the
                                     * collision branch runs there. */
    /* Collectives issued by the library (RCCL, csrc/rccl_comm.hip) instead of through the callbacks: comm from
     * tgnn_rccl_comm_create (NULL: the callbacks are used and must be given); comm_side (may be NULL = comm) carries the side
     * stream's all-to-alls -- a communicator of its own lets the two chains' collectives overlap; send_counts / recv_counts
     * (host, [world]): rows sent to / received from every peer in one exchange, without the extra rows. */
    void *rccl_comm, *rccl_comm_side;
    const int64_t *send_counts, *recv_counts;
    /* [r6] optional (NULL: a pack launch behind every NNConv), split exchange: the INVERSE of send_idx_fused -- for own row v the
     * message rows that carry it are send_row_slot[send_row_ptr[v] .. send_row_ptr[v + 1]) (device, int32, [n_own + 1] and
     * [n_send]).  With them the NNConv's epilogue writes the adjacency branch's halo message itself and its last block the
     * BatchNorm sums: one launch less on the step's critical chain per layer. */
    const int32_t *send_row_ptr, *send_row_slot;
} tgnn_shard;

/* RCCL communicators for tgnn_shard.rccl_comm.  RCCL is looked up at run time (librccl.so.1, the copy already in the process
 * if there is one): tgnn_rccl_available() = 0 when it cannot be found.  One rank calls tgnn_rccl_unique_id (id_out:
 * tgnn_rccl_unique_id_bytes() = 128 bytes) and hands the bytes to the others (any channel: torch.distributed broadcast,
 * MPI, a file); then EVERY rank calls tgnn_rccl_comm_create (a collective; the calling thread's current device is the
 * communicator's). */
int32_t tgnn_rccl_available(void);
size_t tgnn_rccl_unique_id_bytes(void);
int tgnn_rccl_unique_id(void *id_out);
int tgnn_rccl_comm_create(const void *unique_id, int32_t rank, int32_t world, void **comm_out);
int tgnn_rccl_comm_destroy(void *comm);
/* out2[0] / out2[1]: all-to-alls / all-reduces this process has issued through its communicators so far (reporting) */
void tgnn_rccl_counters(int64_t *out2);
size_t tgnn_forward_sharded_workspace_bytes(const tgnn_model_dims *dims, int64_t n_own, int64_t n_rows,
                                            int32_t n_types);
int tgnn_forward_sharded(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                         const float *adj_edge_attr, const tgnn_graph *graph, const tgnn_shard *shard,
                         int32_t update_running, float *probs, void *ws, size_t ws_bytes, tgnn_stream_t stream);

/* Small layouts (what the greedy loop of the reference scores: 1 254 nodes for the labyrinth example): tgnn_forward runs
 * the message-passing layers (TilinGNN.py:59-71) as ONE persistent kernel with grid barriers instead of ~5 dependent
 * launches per layer (csrc/forward_small.hip) when the layout has at most this many nodes, width 32, train-mode
 * BatchNorm and few enough edge types for the weights to sit in LDS.  Same formulas; the BatchNorm sums and the NNConv
 * tile products are associated differently than in the general schedule (fp64 / fp32 rounding), deterministic either way.
 * Process-wide setting, default and maximum 4096 (one 16-row tile per CU); 0 switches the path off (the general schedule then
 * runs at every size).  The persistent kernel and tgnn_graph_prep_small synchronise their blocks with spin barriers and need
 * all of them resident: the library serialises such launches per device (a process-wide mutex and one event per device, waited
 * on and re-recorded by every launch, whichever stream it is on) -- state beyond the "idempotent set-up" of the conventions
 * above, and the reason these two calls cannot be captured into a HIP graph. */
void tgnn_set_small_layout_limit(int64_t n_nodes);
/* No wait of the persistent kernels (small and mid-size layer loops) spins without bound: a block that has waited for the others
 * longer than the budget (default 250 000 us; 0 restores the default; returns the previous value) ORs a reason into a device
 * word and runs on without waiting, so that the launch terminates; the results of that forward are invalid.
 * tgnn_spin_error_poll synchronises `stream`, reads the word of the stream's device into *code_out (0 = every forward since the
 * last poll was sound; bits: 1 grid barrier, 2 partial rows, 4 edge weights) and clears it; with a non-zero code
 * tgnn_last_error() carries the explanation.  What gives up is a kernel whose blocks are not all resident -- another process
 * or tenant holds compute units: run the forward again (the poll has switched the persistent schedules off for a while:
 * tgnn_persist_fallback). */
int tgnn_spin_error_poll(tgnn_stream_t stream, uint32_t *code_out);
uint64_t tgnn_set_spin_budget_us(uint64_t us);
/* The persistent schedules (small and mid-size layer loops) are off for the next n_forwards forwards of this process, then come
 * back by themselves (0: back now).  tgnn_spin_error_poll opens such a window (256 forwards) when it finds a failure, and so does
 * any forward entry point that finds one nobody collected -- it then returns TGNN_ERR_STALE_RESULT (the word cleared, the message
 * set): the results of the EARLIER forward that failed, and of persistent forwards queued behind it, are invalid. */
void tgnn_persist_fallback(int64_t n_forwards);
/* what the host sees of the current device's spin-error word right now (its host-mapped mirror: no copy, no wait, not cleared) */
uint32_t tgnn_spin_error_peek(void);
int64_t tgnn_get_small_layout_limit(void);

/* Split precision of the general schedule's matrix-core kernels (NNConv, the final MLP's first Linear).  Both hold the fp32
 * operands to fp32-class accuracy on the low-precision matrix instructions of gfx950:
 *   0: bf16 x 3 -- every operand split exactly into three bf16 pieces, six cross terms;
 *   1: fp16 x 2 (default) -- operands scaled by a power of two, split into an fp16 pair (2^-22 relative), three cross terms:
 *      half the matrix cycles.  Needs a bound of every operand: the kernels that write the skip buffer leave the largest
 *      magnitude of each slot, one small launch per forward those of the weights; the layout's largest in-degree comes from
 *      the preparation (tgnn_graph.nn_max_in_degree; 0 = unknown: bf16 x 3 runs).  Train-mode BatchNorm, single device or
 *      the one-all-to-all sharded schedule (every shard scales by its own bounds); eval mode, the all-reduce + all-to-all
 *      sharded scheme and the per-op entry points stay on bf16 x 3.
 * Process-wide; returns the previous mode (any other argument: only queries). */
int32_t tgnn_set_split_precision(int32_t mode);
/* GINConv's neighbourhood sum and MLP as ONE kernel on single-device inference forwards (the aggregate never travels through
 * HBM; csrc/gin.hip: gin32_fused_kernel): 0 never (the two kernels of tgnn_gin_fwd), 1 (default) for layouts of 200 000 nodes and
 * more -- where it is faster --, 2 always.  Same arithmetic, the CollConv rows are bit-identical.  Returns the previous mode. */
int32_t tgnn_set_gin_fused(int32_t mode);
/* The collision branch's MLP inside tgnn_forward / tgnn_forward_sharded (inference): 1 = layers 2 and 3 on fp16 pairs
 * (three matrix terms; their inputs are sigmoids), 16 waves per block (csrc/gin.hip: gin32_mlp16_kernel); 0 (default: the
 * forward measured no faster with it, csrc/gin.hip) = the bf16 x 3 kernel the training forward and tgnn_gin_fwd run.  Returns the previous setting. */
int32_t tgnn_set_gin_mlp_f16(int32_t on);
/* Mid-size layouts (4 097 .. tgnn_get_mid_layout_limit() nodes): the two ends of the network inside the persistent kernels
 * instead of launch-per-layer (reference: graph_networks/networks/TilinGNN.py:54 and :74-76).  Bit 0: the final MLP behind the
 * persistent layer loop as one persistent kernel of its own (csrc/forward_tail.hip; layouts of up to 16 384 nodes) instead of 5
 * Linear + 4 bn_finalize launches; bit 1: the init MLP in the layer loop's prologue (csrc/forward_mid.hip; node_features_dim <= 8)
 * instead of 5 launches in front of it.  BatchNorm statistics are all-reduced over the grid in a fixed order (bit-reproducible).
 * Default 3; returns the previous setting (an argument outside 0 .. 3 only queries). */
int32_t tgnn_set_mid_tail(int32_t on);
/* General schedule, width 32, fp16-pair operands: the NNConv over the layout's edge groups (csrc/nnconv_eg.hip) when the graph
 * carries them, instead of over its type columns.  Default 1; returns the previous setting (an argument outside 0 .. 1 only
 * queries). */
int32_t tgnn_set_nnconv_eg(int32_t on);
/* The final MLP's fp16-pair Linears for many rows (csrc/dense.hip; reference: graph_networks/networks/TilinGNN.py:74-76).  Bit 1
 * (default on): the BatchNorm-on-load layers 256 -> 128 -> 64 on dense_f16_resident_kernel (W's operand image resident in LDS, no
 * barrier in the k loop, 16 KB requests per wave).  Bit 0 (default off; an experiment that measured slower): dense_f16_rows2_kernel
 * (one wave per SIMD, every operand two k-tiles in flight) instead of dense_f16_rows_kernel for what is left.  The same bits
 * in every setting.  Default 2; returns the previous setting (an argument outside 0 .. 3 only queries). */
int32_t tgnn_set_dense_rows_mode(int32_t mode);
/* The front of the general schedule (reference: graph_networks/networks/TilinGNN.py:54 and the operand preparation in front of the
 * first layer).  Bit 0: no memset in front of the first launch, the layer loop waits for the edge weights only (the final MLP's
 * bounds and operand images are one launch each behind them); bit 1: the init MLP as three launches that recompute from x
 * (csrc/init_mlp.hip; the same bits as the five launches it replaces); bit 2 (off by default: measured no faster): the final MLP's
 * BatchNorm records written by their producers (bn_fold_two_level) instead of a bn_finalize launch behind each -- the same bits;
 * bit 3 (off by default; A/B switch): tgnn_forward_resume keeps the edge weights on the side stream instead of queueing them on
 * `stream` straight behind the preparation's last launch (40 us slower per step, profiles/r06_resume_weights.txt).
 * Default 3; returns the previous setting (an argument outside 0 .. 15 only queries). */
int32_t tgnn_set_lean_head(int32_t bits);

/* Small layouts (<= tgnn_graph_prep_small_max_nodes() nodes, <= ..._max_edges() edges per set): everything above --
 * tgnn_csr_build of both edge sets (self loops dropped from the collision set), tgnn_edge_type_dedup, the types in CSR
 * order, tgnn_nnconv_cols_build -- in ONE launch (up to 16 resident blocks with grid barriers); every output bit-identical
 * to the separate calls.  tmp: tgnn_graph_prep_small_tmp_ints() ints.  counters: 2 words of scratch, zeroed by the call on
 * `stream` (one call at a time per counter pair).  result [32] (device; 8 used): n_types,
 * adjacency index error, collision index error, collision CSR slots, largest adjacency in-degree, 1 = column structure
 * built (n_types <= tgnn_nnconv_cols_max_types()), 1 = fall back to the separate calls (more than 1024 distinct attribute
 * rows).  Asynchronous.  [r6] result_host (NULL, or 32 words of pinned host memory the device can address): the kernel stores the
 * words there too, word 31 = 0x600D0001 last, and tgnn_graph_prep_wait(stream) polls it -- no copy, no stream synchronise. */
int64_t tgnn_graph_prep_small_max_nodes(void);
int64_t tgnn_graph_prep_small_max_edges(void);
size_t tgnn_graph_prep_small_tmp_ints(int64_t n_nodes, int64_t n_adj_edges, int64_t n_col_edges);
int tgnn_graph_prep_small(const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_attr, int32_t fe,
                          const int64_t *col_edge_index, int64_t n_col_edges, int64_t n_nodes, int32_t *adj_rowptr,
                          int32_t *adj_src, int32_t *adj_eid, int32_t *adj_type, int32_t *edge_type, int32_t *type_rep_edge,
                          int32_t *col_rowptr, int32_t *col_src, int32_t *col_eid, int32_t *tile_col_ptr, int32_t *col_meta,
                          int32_t *col_slot_src, int32_t *tmp, int32_t *result, uint32_t *counters, int32_t *result_host,
                          tgnn_stream_t stream);

/* The same at any size: one call that queues every launch of the preparation itself (no host round trip in the middle:
 * the column structure reads the type count from the device).  result [32] as above; word 6 = 1: the layout has more than
 * 4 096 distinct attribute rows -- more than the one-block numbering of the types takes: edge_type / adj_type / the column
 * structure hold nothing usable, the caller goes through the separate calls (tgnn_edge_type_dedup has no such limit).  With
 * mid_tile_nb / mid_ent (both or none; sized as for tgnn_mid_entries_build) the batches of the mid-size layer loop are built
 * too: result[8..9] = that call's result words.  n_src_nodes >= n_nodes: sources may index rows behind the n_nodes destinations
 * (the halo rows of a shard, as in tgnn_csr_build); n_nodes on a single device.
 * The NNConv structures: the type columns (tile_col_ptr, col_meta, col_slot_src: all three or none) and / or the edge groups
 * (tile_grp_ptr + grp, sized by tgnn_nnconv_eg_max_groups with T = tgnn_nnconv_cols_max_types(): both or none; result[10] = 1:
 * built); at least one of the two. */
size_t tgnn_graph_prep_workspace_bytes(int64_t n_nodes, int64_t n_adj_edges, int64_t n_col_edges, int32_t fe);
int tgnn_graph_prep(const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_attr, int32_t fe,
                    const int64_t *col_edge_index, int64_t n_col_edges, int64_t n_nodes, int64_t n_src_nodes, int32_t *adj_rowptr,
                    int32_t *adj_src, int32_t *adj_eid, int32_t *adj_type, int32_t *edge_type, int32_t *type_rep_edge, int32_t *col_rowptr,
                    int32_t *col_src, int32_t *col_eid, int32_t *tile_col_ptr, int32_t *col_meta, int32_t *col_slot_src,
                    int32_t *mid_tile_nb, uint32_t *mid_ent, int32_t *tile_grp_ptr, int32_t *grp,
                    void *ws, size_t ws_bytes, int32_t *result, int32_t *result_host, tgnn_stream_t stream);
/* result_host (pinned host memory, 32 words, or NULL): the library copies `result` there as soon as the words a caller reads --
 * [0] type count, [1] [2] index errors, [3] collision CSR slots, [4] largest in-degree, [6] fall-back -- are final, i.e. BEFORE the
 * launches that build the NNConv structure; tgnn_graph_prep_wait(stream) (same host thread, the stream tgnn_graph_prep was given:
 * the event belongs to that stream's device, whatever the thread's current device is) waits for that copy alone, so
 * the caller can queue its forward while the structure is still being built.  Words [5], [8], [9], [10] are not final in that
 * copy: [5] = [10] = ([0] <= tgnn_nnconv_cols_max_types() && ![6]) for the structures asked for; a caller that needs [8] / [9]
 * (the mid-size batches) reads `result` after the stream instead. */
int tgnn_graph_prep_wait(tgnn_stream_t stream);
/* [r6] With edge groups alone asked for (the general schedule's layouts) and result_host addressable by the device
 * (hipHostGetDevicePointer), the words are not copied: the first wave of the first launch behind the join of the preparation's two
 * chains stores them into result_host and sets word 31 to 0x600D0001 last (system scope), and tgnn_graph_prep_wait polls that word
 * (two seconds, then it synchronises the stream) -- no copy launch and no hand-over to a second queue between the CSR chain and
 * the host.  result_host[31] is the library's from tgnn_graph_prep to tgnn_graph_prep_wait.  on = 0: always the copy; returns the
 * previous setting (negative: query). */
int32_t tgnn_set_prep_words_poll(int32_t on);
#ifdef TGNN_DEBUG
/* ---- test / experiment hooks: only in libtgnn_debug.so (make -C tilingnn_amd/csrc debug: the same sources with -DTGNN_DEBUG); the
 *      production library does not export them ---- */
/* (tests) tgnn_graph_prep builds both CSRs through buckets of 512 destination rows sorted in LDS; a bucket with more than `cap`
 * edges (default and maximum 15 360) takes a slow in-place path.  Sets the threshold (negative: only queries); returns the
 * previous one. */
int32_t tgnn_debug_set_csr_bucket_cap(int32_t cap);
/* (experiments: scratch/block_caps.py, scratch/mid_trace.sh) upper bounds on the grids of the two kernels whose blocks need a CU to
 * themselves -- the column NNConv and the GIN MLP (at most the device's CU count); 0 = the built-in policy (device CUs minus 32 each). */
void tgnn_debug_set_block_caps(int32_t nnconv_blocks, int32_t gin_mlp_blocks);
/* the next n_launches persistent kernels run with their last block absent (it returns at once) -- what a block that never
 * becomes resident looks like to the others; they must give up after the budget and report through the spin-error word. */
void tgnn_debug_spin_fault(int32_t n_launches);
#endif

/* The same forward with a hipEvent pair around every launch (on `stream`, where the kernels run);
 * synchronises, then ADDS the elapsed milliseconds and launch counts per kernel class into the
 * two host arrays of TGNN_PROF_CLASSES entries.  Measurement aid for bench.py's roofline line. */
#define TGNN_PROF_CLASSES 8
#define TGNN_PROF_EDGE_WEIGHTS 0 /* edge MLP on T rows, all layers   */
#define TGNN_PROF_DENSE_INIT 1   /* init MLP GEMMs + BN apply        */
#define TGNN_PROF_NNCONV 2       /* NNConv mean (+LeakyReLU, BN sums) */
#define TGNN_PROF_GIN 3          /* GIN aggregate + MLP              */
#define TGNN_PROF_BN_FINALIZE 4
#define TGNN_PROF_MERGE 5
#define TGNN_PROF_DENSE_FINAL 6  /* final MLP GEMMs                  */
int tgnn_forward_profiled(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                          const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                          int32_t use_running_stats, float *probs, void *ws, size_t ws_bytes,
                          tgnn_stream_t stream, float *class_ms_host, int32_t *class_launches_host);

/* The same measurement INSIDE the two-stream forward (the schedule tgnn_forward runs): event pairs on `stream` around the
 * kernels of the adjacency chain only (TGNN_PROF_NNCONV, TGNN_PROF_MERGE) while the collision chain runs beside them on
 * stream2 -- the average launch duration the production forward actually sees, contention included.  Train mode. */
int tgnn_forward_profiled_two_stream(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                     const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                                     float *probs, void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2,
                                     float *class_ms_host, int32_t *class_launches_host);
/* K independent layouts -- every one a batch of its own, exactly what K tgnn_forward calls compute -- queued in ONE call:
 * layout k on streams[k % n_streams] (and the shared side stream stream2).  For small layouts (tgnn_set_small_layout_limit) the
 * persistent kernels of different streams run side by side when they fit the device together: the reference's crop loop
 * (Tiling-Shape.py:52-64) hands over layouts of ~1 000 nodes that fill a third of the chip each.  The caller orders `streams`
 * with its own work before and after.  update_running != 0 with layouts sharing one parameter set: the K updates of the
 * running statistics race (they do not enter train-mode outputs); pass 0 to leave them untouched.
 * (HIP binds streams to its hardware queues -- 4 by default, GPU_MAX_HW_QUEUES -- round robin, and two streams on one queue run
 *  their kernels one after the other: hand over streams that were seen to overlap; tilingnn_amd._lib.concurrent_streams
 *  measures that.) */
int tgnn_forward_many(const tgnn_model_dims *dims, const void *const *params_host, int32_t n_layouts, const float *const *x,
                      const float *const *adj_edge_attr, const tgnn_graph *graphs, int32_t update_running,
                      int32_t use_running_stats, float *const *probs, void *const *ws, const size_t *ws_bytes,
                      const tgnn_stream_t *streams, int32_t n_streams, tgnn_stream_t stream2);
/* The PRODUCTION forward (tgnn_forward, train mode, two chains when stream2 is given; no event, no profiler) with the column
 * NNConv launches stamped on the device's wall clock: nnconv_us_host [network_depth] = last block out - first block in of every
 * layer's launch, microseconds -- the duration a kernel trace reports, measured inside the schedule as it runs (0 where the
 * layer did not go through the column kernel: small layouts, no column structure).  Synchronises both streams. */
int tgnn_forward_stamped(const tgnn_model_dims *dims, const void *const *params_host, const float *x, const float *adj_edge_attr,
                         const tgnn_graph *graph, int32_t update_running, float *probs, void *ws, size_t ws_bytes,
                         tgnn_stream_t stream, tgnn_stream_t stream2, float *nnconv_us_host);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU helpers: node-range shards exchange boundary rows each layer (RCCL does the moving)
 * ------------------------------------------------------------------------------------------ */
/* out[i, :] = src[idx[i], :]  and  dst[idx[i], :] = in[i, :]   for [*, C] fp32 rows */
/* ---- sub-layout of the unlabelled nodes, for the greedy assembly loop -----------------
 * BrickLayout.compute_sub_layout (tiling/brick_layout.py:248-286) as a stream compaction on the device:
 * alive [N] (int32, != 0 = still unlabelled); the alive nodes in ascending order become nodes 0..N'-1; an edge
 * survives iff both ends are alive; survivors keep their order, carry re-indexed ends and their attribute rows.
 *   x_out [N'][Fx], inverse_out [N'] (new -> old node), adj_out [2][Ea'] (row 1 starts Ea' entries after row 0),
 *   adj_attr_out [Ea'][Fe], col_out [2][Ec'];  counts_out (device int64 [3]) = {N', Ea', Ec'}.
 * Output arrays are sized for the un-compacted counts.  (The collision attributes are not carried: the network
 * never reads them, TilinGNN.py:51.)  err_flag (device, may be NULL) is set if an edge end is outside [0, N). */
size_t tgnn_sublayout_workspace_bytes(int64_t n_nodes, int64_t n_adj_edges, int64_t n_col_edges);
int tgnn_sublayout_compact(const int32_t *alive, int64_t n_nodes, const float *x, int32_t fx,
                           const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_attr, int32_t fe,
                           const int64_t *col_edge_index, int64_t n_col_edges, float *x_out, int64_t *inverse_out,
                           int64_t *adj_out, float *adj_attr_out, int64_t *col_out, int64_t *counts_out,
                           int32_t *err_flag, void *ws, size_t ws_bytes, tgnn_stream_t stream);

/* A node-range SHARD's rows of the next greedy round (tilingnn_amd.dist.compact_shard_device): alive_local [n_rows] = 1 for the
 * owned rows (0 .. n_own - 1) that are still unlabelled and for the halo rows (n_own ..) that are unlabelled AND still the source
 * of an edge whose destination is unlabelled too; tgnn_sublayout_compact over the shard's local row space with that mask then IS
 * the shard of the sub-layout.  alive_global [N] over the current global numbering, gid [n_rows] the global number of every local
 * row, edge indices [2][E] in local numbering.  *err_flag is set on an edge end outside [0, n_rows). */
int tgnn_shard_alive_rows(const int32_t *alive_global, const int64_t *gid, int64_t n_own, int64_t n_rows,
                          const int64_t *adj_edge_index, int64_t n_adj_edges, const int64_t *col_edge_index, int64_t n_col_edges,
                          int32_t *alive_local, int32_t *err_flag, tgnn_stream_t stream);

/* ---- one round of the greedy assembly loop's acceptance, batched on the device (csrc/greedy.hip; the DOCUMENTED SUBSTITUTE of
 * the sequential sweep of util/algorithms.py:41-54 for large layouts -- not bit-compatible with the reference's RNG stream, see
 * the file's header; tilingnn_amd.util.algorithms keeps the reference's sweep on the host as the default).
 *   prob [n_sub] (stride ld_prob floats): this round's probabilities of the sub-layout's nodes; inverse [n_sub] (may be NULL =
 *   identity): sub-layout node -> original node; col_edge_index [2][n_col_edges]: the sub-layout's collision edges (both
 *   directions, sub-layout numbering); round >= 1; prob_saved [N] doubles over ORIGINAL nodes (1.0 before round 1): the running
 *   geometric mean of :33-34, updated; alive [N] int32: cleared for accepted nodes and their collision neighbours;
 *   selected_round [N] int32: set to `round` for accepted nodes; *n_selected (device int64) += accepted; *err_flag set on an edge
 *   end outside [0, n_sub).  A node is accepted when it precedes all its neighbours in the reference's visiting order (larger
 *   running mean first, smaller number on ties) and exp(p - 1) > u, u = uniform(seed, round, original node) -- counter based,
 *   reproducible.  No two accepted nodes collide. */
size_t tgnn_greedy_round_workspace_bytes(int64_t n_sub);
int tgnn_greedy_round(const float *prob, int64_t ld_prob, const int64_t *inverse, int64_t n_sub, const int64_t *col_edge_index,
                      int64_t n_col_edges, int32_t round, uint64_t seed, double *prob_saved, int32_t *alive,
                      int32_t *selected_round, int64_t *n_selected, int32_t *err_flag, void *ws, size_t ws_bytes,
                      tgnn_stream_t stream);
/* [r6] The END of a greedy solve as one launch: on a sub-layout without adjacency edges (or without collision edges) every node
 * gets probability 1 from ML_Solver.predict without the network (ml_solver.py:31-32), and so on every sub-layout of it -- the
 * remaining rounds are tgnn_greedy_round with prob = 1 on ever smaller sub-layouts.  One block runs them all on the given
 * sub-layout (n_sub <= tgnn_greedy_finish_max_nodes()): the same means, visiting order, draws and round numbers (first_round,
 * first_round + 1, ...), hence the same selection as round by round.  out [2] (device): rounds run, nodes still unlabelled
 * (0 unless max_rounds ran out). */
int64_t tgnn_greedy_finish_max_nodes(void);
int tgnn_greedy_finish(const int64_t *inverse, int64_t n_sub, const int64_t *col_edge_index, int64_t n_col_edges,
                       int32_t first_round, int32_t max_rounds, uint64_t seed, double *prob_saved, int32_t *alive,
                       int32_t *selected_round, int64_t *n_selected, int32_t *err_flag, int32_t *out, tgnn_stream_t stream);

/* ---- the loss on the predict path -------------------------------------------------
 * Losses.calculate_unsupervised_loss (solver/ml_solver/losses.py:48-116), evaluated by ML_Solver.predict through
 * get_best_prob_map (ml_solver.py:46,133-136): for every probability map m (column of probs [N, n_maps])
 *   loss[m] = (1 - Wa log max(mean_v area[v] p[v], 1e-7))
 *           * (1 - Wc mean_{collision edges} log(1 - clamp(p[i] p[j], 1e-7, 1 - 1e-7)))      (1 if there are none)
 *           * (1 - Wl mean_{adjacency edges} log10 max(p[i] p[j] len_e, 1e-7))                 (1 if there are none)
 * area_ratio = last column of the node features (x + Fx - 1, ld_area = Fx); adj_edge_len = column 1 of the
 * adjacency edge attributes (attr + 1, ld_len = Fe); edge indices int64 [2, E] as everywhere.  Elements in fp32,
 * sums in fp64 over a fixed tree.  losses: device [n_maps] doubles; terms (may be NULL): device [n_maps][3] =
 * the three logarithmic terms.  The caller picks argmin (the reference: np.argsort(losses)[0]).
 * An edge end outside [0, N) (torch.gather raises there) makes every loss NaN; such edges are skipped, nothing is
 * read or written out of range. */
size_t tgnn_unsupervised_loss_workspace_bytes(int32_t n_maps);
int tgnn_unsupervised_loss(const float *probs, int64_t ld_probs, int32_t n_maps, const float *area_ratio,
                           int64_t ld_area, int64_t n_nodes, const int64_t *col_edge_index, int64_t n_col_edges,
                           const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_len,
                           int64_t ld_len, float collision_weight, float align_length_weight, float avg_area_weight,
                           double *losses, double *terms, void *ws, size_t ws_bytes, tgnn_stream_t stream);

/* Losses.solution_score (solver/ml_solver/losses.py:120-148), the quality of a finished 0/1 selection that
 * create_solution (util/algorithms.py:210-220) attaches to every greedy solve: the three sums behind it,
 *   sums[0] = sum_v predict[v] area_ratio[v]                (:126, area_ratio = x[:, -1]: times max_area / contour area)
 *   sums[1] = sum_e predict[src_e] predict[dst_e] len_e     (:131-141, len_e = attr[:, 1]: times max_align_length)
 *   sums[2] = sum_{v: predict[v] == 1} perimeter[v]         (:143-144)
 * products in fp32 (the reference's type), sums in fp64 over a fixed tree; sums: device [3] doubles (NaN when an edge
 * end is outside [0, N)).  The caller forms  Wa sums[0] max_area / contour_area + Wl sums[1] max_align_length / sums[2].
 * ws: tgnn_unsupervised_loss_workspace_bytes(1). */
int tgnn_solution_score_sums(const float *predict, const float *area_ratio, int64_t ld_area, const float *perimeter,
                             int64_t n_nodes, const int64_t *adj_edge_index, int64_t n_adj_edges,
                             const float *adj_edge_len, int64_t ld_len, double *sums, void *ws, size_t ws_bytes,
                             tgnn_stream_t stream);

/* ---- the training step: adjoints of the forward kernels ---------------------------------
 * Trainer.train (solver/ml_solver/trainer.py:68-84) = forward in train mode, the unsupervised loss, loss.backward(),
 * optimizer.step().  torch.autograd derives the backward there; these are the same adjoints per forward kernel
 * (csrc/backward.hip; scheduled by tilingnn_amd/train.py).  Column sums run in fp64 over fixed trees. */
int tgnn_transpose(const float *w, int32_t rows, int32_t cols, float *out, tgnn_stream_t stream);
/* out[b][a][c] = in[a][b][c], with rows of out_da >= da entries in `out` (weight tables re-laid for the input-gradient
 * product of NNConv: the T type matrices and, behind them, the root matrix) */
int tgnn_swap_leading(const float *in, int32_t da, int32_t db, int32_t dc, float *out, int32_t out_da,
                      tgnn_stream_t stream);
/* z = (1 + eps) BN_in(a) + sum over the row's CSR slots of BN_in(a)[src] (width 32): the input of GINConv's MLP; on the
 * transposed collision graph, the adjoint of that aggregation. */
int tgnn_gin_aggregate(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr, const int32_t *col_src,
                       const float *eps, int64_t n_nodes, int32_t c, float *z, tgnn_stream_t stream);
/* out = d * t * (1 - t): derivative of torch.nn.Sigmoid given its output t (layers/util.py:34) */
int tgnn_sigmoid_bwd(const float *d, int64_t ld_d, const float *t, int64_t ld_t, int64_t n_rows, int32_t c, float *out,
                     int64_t ld_o, tgnn_stream_t stream);
int tgnn_add_into(const float *src, int64_t ld_s, int64_t n_rows, int32_t c, float *dst, int64_t ld_d,
                  tgnn_stream_t stream);
/* scratch for tgnn_colsum / tgnn_bn_bwd_reduce / tgnn_merge_bwd_reduce on matrices of `width` columns */
size_t tgnn_reduce_workspace_bytes(int32_t width);
/* out[c] = sum_r x[r][c]: the bias gradient of a Linear (layers/util.py:32) */
int tgnn_colsum(const float *x, int64_t ld, int64_t n_rows, int32_t c, float *out, void *ws, size_t ws_bytes,
                tgnn_stream_t stream);
/* Train-mode BatchNorm1d backward fused with the derivative of the activation in front of it
 * (Linear_trans: Linear -> activation -> BatchNorm, layers/util.py:31-37; GraphConv / CollConv: conv -> LeakyReLU ->
 * BatchNorm, edge_conv.py:24-30).  a = the activation's output (the BatchNorm's input), stat = the forward's record.
 *   reduce: coef [2][F] (scratch for apply), dgamma, dbeta (may be NULL)
 *   apply : dz = act'(a) gamma invstd (dy - mean(dy) - xhat mean(dy xhat));  scaled (may be NULL) = dz * row_scale[r] */
int tgnn_bn_bwd_reduce(const float *dy, int64_t ld_dy, const float *a, int64_t ld_a, const float *stat, int64_t n_rows,
                       int32_t f, float eps, float *coef, float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                       tgnn_stream_t stream);
int tgnn_bn_bwd_apply(const float *dy, int64_t ld_dy, const float *a, int64_t ld_a, const float *stat, const float *coef,
                      int64_t n_rows, int32_t f, int32_t act, float *dz, int64_t ld_dz, const float *row_scale,
                      float *scaled, int64_t ld_scaled, tgnn_stream_t stream);
/* Backward of the branch merge h = BN1(a1) * BN2(a2) (+ resid) (TilinGNN.py:64-71) and the reductions of both
 * BatchNorms behind it (width 32):  dy1 = dh * BN2(a2);  dy2 = dh * BN1(a1) (+ carry: the gradient arriving at
 * BN2(a2) from the next CollConv, TilinGNN.py:63);  resid_grad (may be NULL) += dh. */
int tgnn_merge_bwd_reduce(const float *dh, int64_t ld_dh, const float *a1, const float *stat1, const float *a2,
                          const float *stat2, const float *carry, int64_t n_rows, int32_t c, float eps1, float eps2,
                          float *dy1, float *dy2, float *resid_grad, int64_t ld_resid, float *coef1, float *dgamma1,
                          float *dbeta1, float *coef2, float *dgamma2, float *dbeta2, void *ws, size_t ws_bytes,
                          tgnn_stream_t stream);
/* Weight gradient of a Linear: out [cout, cin] = dz^T . x (fp32 matrix cores); dbias [cout] (may be NULL) = the column
 * sums of dz from the same pass.  x element (r, k) lives at
 * x[(k / 32) * x_kblock_stride + r * ld_x + k % 32] when x_kblock_stride != 0 (the slot-major skip buffer), else at
 * x[r * ld_x + k]. */
size_t tgnn_wgrad_workspace_bytes(int64_t n_rows, int32_t cout, int32_t cin);
int tgnn_wgrad(const float *dz, int64_t ld_dz, const float *x, int64_t ld_x, int64_t x_kblock_stride, int64_t n_rows,
               int32_t cout, int32_t cin, float *out, float *dbias, void *ws, size_t ws_bytes, tgnn_stream_t stream);
/* Backward of a 3-layer sigmoid MLP without BatchNorm (GraphConv's edge MLP, edge_conv.py:17-18; GINConv's MLP,
 * coll_conv.py:14-18) in one call: hidden activations re-derived from x [n, d0]; per layer dpre = d t (1 - t),
 * dW = dpre^T . in, db, d_in = dpre . W.  w_k [d_k, d_{k-1}]; t3 = the MLP's output [n, d3]; d_out: gradient at t3;
 * dx [n, d0] may be NULL.  (b3 is not needed: t3 is given.) */
size_t tgnn_sigmoid_mlp_bwd_workspace_bytes(int64_t n_rows, int32_t d0, int32_t d1, int32_t d2, int32_t d3);
int tgnn_sigmoid_mlp_bwd(const float *x, int64_t n_rows, int32_t d0, int32_t d1, int32_t d2, int32_t d3, const float *w1,
                         const float *b1, const float *w2, const float *b2, const float *w3, const float *t3,
                         const float *d_out, int64_t ld_dout, float *dw1, float *db1, float *dw2, float *db2, float *dw3,
                         float *db3, float *dx, void *ws, size_t ws_bytes, tgnn_stream_t stream);
/* NNConv backward building block (edge_conv.py:25; PyG NNConv: message = x_j . W_e, mean, + x . root):
 * out [n_nodes][(n_types + 1) * 32]: slot t < n_types = sum over the row's CSR slots of type t of rows[src];
 * slot n_types = own[j] * root_scale[j] (NULL: 1).  Run over the TRANSPOSED adjacency CSR on g = dz / deg (root slot = dz) it
 * makes both adjoints dense products:  d x = slots . [W_t^T; root^T]  and  [d W_t; d root] = x^T . slots. */
int tgnn_nnconv_type_sum(const float *rows, int64_t ld_rows, const float *own, int64_t ld_own, const float *root_scale,
                         const int32_t *rowptr, const int32_t *src, const int32_t *type, int64_t n_nodes, int32_t n_types,
                         int32_t c, float *out, tgnn_stream_t stream);
/* deg[j] = max(in-degree, 1) (scatter_mean's clamp) and 1 / deg[j] from a CSR row pointer */
int tgnn_csr_degree(const int32_t *rowptr, int64_t n_nodes, float *deg, float *inv_deg, tgnn_stream_t stream);
/* d loss / d probs of tgnn_unsupervised_loss for ONE probability map (the arg-min map: the reference back-propagates
 * through torch.min, losses.py:108).  probs / dprobs point at that map's column; terms = the three doubles the forward
 * wrote for it; grad_out (device float, NULL = 1) = d objective / d loss.  ws: n_nodes doubles. */
int tgnn_unsupervised_loss_bwd(const float *probs, int64_t ld_probs, const float *area_ratio, int64_t ld_area,
                               int64_t n_nodes, const int64_t *col_edge_index, int64_t n_col_edges,
                               const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_len, int64_t ld_len,
                               float collision_weight, float align_length_weight, float avg_area_weight,
                               const double *terms, const float *grad_out, float *dprobs, int64_t ld_dprobs, void *ws,
                               size_t ws_bytes, tgnn_stream_t stream);

/* ---- BASELINE config 3: network_width 64, bf16 STORAGE of the activations that cross HBM between kernels (skip buffer,
 * pre-BatchNorm branch outputs, GIN aggregate), fp32 accumulation, fp64 BatchNorm sums, fp32 stat records and final-MLP
 * activations (inputs/config.py:17,37-38).  csrc/bf16_path.hip.  bf16 buffers are `void *` here
 * (2 bytes per element, row-major, 64 per row).  Same graph structure (tgnn_graph incl. the NNConv columns), same parameter
 * table and BatchNorm semantics as tgnn_forward; train mode only.  Per-op entries (the layer seams, for parity tests): */
int tgnn_f32_to_bf16(const float *src, int64_t count, void *dst_bf16, tgnn_stream_t stream);     /* round to nearest even */
size_t tgnn_nnconv64_image_elems(int32_t n_types);   /* bf16 elements of wimg_scratch */
/* NNConv mean + root + bias (+ LeakyReLU) over the type-column structure: every column feeds 8 MFMAs against the bf16
 * image of W_type; out_bf16 [N][64]; BatchNorm partial rows [blocks][2][64] of the ROUNDED output */
int tgnn_nnconv64_bf16_fwd(const void *h_bf16, int64_t n_src_rows, const int32_t *tile_col_ptr, const int32_t *col_meta,
                           const int32_t *col_src, const float *wtab, int32_t n_types, const float *root,
                           const float *bias, int64_t n_nodes, int32_t act, void *out_bf16, void *wimg_scratch,
                           double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream);
/* The same op over the layout's edge groups (tgnn_nnconv_eg_build) instead of its type columns: csrc/bf16_path.hip,
 * nnconv64_bf16_eg_kernel -- what tgnn_forward_bf16 runs on a graph that carries groups (tgnn_set_nnconv_eg).  The messages
 * are rounded to bf16 once before they are folded into their rows (one more rounding of 2^-9 beside those of the operands and
 * of the result: inside the path's stated 2^-7). */
int tgnn_nnconv64_bf16_eg_fwd(const void *h_bf16, int64_t n_src_rows, const int32_t *tile_grp_ptr, const int32_t *grp,
                              const float *wtab, int32_t n_types, const float *root, const float *bias, int64_t n_nodes,
                              int32_t act, void *out_bf16, void *wimg_scratch, double *bn_partial, int32_t *n_partials_host,
                              tgnn_stream_t stream);
/* GINConv (MLP 64 -> 32 -> 64 -> 64, sigmoids) + optional LeakyReLU; in_stat as in tgnn_gin_fwd; z_scratch_bf16 [N][64] */
int tgnn_gin64_bf16_fwd(const void *a_bf16, const float *in_stat, const int32_t *rowptr, const int32_t *col_src,
                        const float *eps, const float *w1, const float *b1, const float *w2, const float *b2,
                        const float *w3, const float *b3, int64_t n_nodes, int32_t act, void *out_bf16,
                        void *z_scratch_bf16, double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream);
/* CollConv.forward (coll_conv.py:24-30) incl. its train-mode BatchNorm, the OUTPUT stored as bf16 (the pre-BatchNorm sigmoid
 * columns vary by ~5e-3: 16-bit storage in front of the BatchNorm would destroy them).  One pass over the MLP leaves the
 * statistics and the fp32 rows in pre_scratch_f32 [N][64]; an element-wise pass normalises, rounds and stores.  stat_scratch:
 * 4 x 64 floats (the record); bn_partial: TGNN_BN_MAX_PARTIALS x 128 doubles. */
int tgnn_collconv64_bf16_fwd(const void *h2_in_bf16, const int32_t *rowptr, const int32_t *col_src, const float *eps,
                             const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                             const float *b3, const float *gamma, const float *beta, float *running_mean,
                             float *running_var, int64_t *num_batches_tracked, int64_t n_nodes, void *out_bf16,
                             void *z_scratch_bf16, float *pre_scratch_f32, float *stat_scratch, double *bn_partial,
                             tgnn_stream_t stream);
/* stat2 == NULL: a2 holds BN2's output already (tgnn_collconv64_bf16_fwd) */
int tgnn_merge_bf16_fwd(const void *a1_bf16, const float *stat1, const void *a2_bf16, const float *stat2,
                        const void *resid_bf16, int64_t n_nodes, int32_t c, void *out_bf16, tgnn_stream_t stream);
/* act(cat . w^T + b) with cat = the slot-major bf16 skip buffer [n_slots][n_rows][64] read in place; w fp32
 * [out_dim][64 n_slots] is rounded to bf16 into wb_scratch (out_dim * 64 * n_slots bf16 elements); out fp32 */
int tgnn_dense_bf16_slots_fwd(const void *a_bf16, int64_t slot_stride, int32_t n_slots, const float *w, const float *b,
                              int64_t n_rows, int32_t out_dim, int32_t act, float *out, void *wb_scratch,
                              double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream);
size_t tgnn_forward_bf16_workspace_bytes(const tgnn_model_dims *dims, int64_t n_nodes, int32_t n_types);
int tgnn_forward_bf16(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                      const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running, float *probs,
                      void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2);
/* [r6] A NEW layout: the init MLP (TilinGNN.py:54 -- it needs nothing of the graph) queued on stream2 BEFORE the layout's
 * preparation, so that it runs beside it; the caller has ordered stream2 behind x and the parameters.  ws: as for
 * tgnn_forward_bf16 (the workspace's size and layout do not depend on the type count).  The next tgnn_forward_bf16 of this thread
 * with the same ws and node count waits for it and skips its own init MLP (the running-statistics update was applied here). */
int tgnn_forward_bf16_begin(const tgnn_model_dims *dims, const void *const *params_host, const float *x, int64_t n_nodes,
                            int32_t update_running, void *ws, size_t ws_bytes, tgnn_stream_t stream2);

int tgnn_rows_gather(const float *src, int64_t ld_src, const int32_t *idx, int64_t n_idx, int32_t c,
                     float *out, int64_t ld_out, tgnn_stream_t stream);
int tgnn_rows_scatter(const float *in, const int32_t *idx, int64_t n_idx, int32_t c, float *dst,
                      int64_t ld_dst, tgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TGNN_H */
