// Whole-network orchestration and C-ABI glue of libtgnn.so.
//
// tgnn_forward == TilinGNN.forward (/root/reference/graph_networks/networks/TilinGNN.py:51-78):
//   :54     init MLP (2 x Linear -> LeakyReLU -> BN)                     dense_mfma + bn_finalize
//   :59-71  D x { GraphConv(h1) || CollConv(h2); h1 = g1*h2 (+ middle[i-2]) }
//                                                                       nnconv32 / gin32 / bn_finalize(x2) / merge
//   :74-76  cat(middle) -> final MLP -> sigmoid                          dense_mfma reading the slot-major
//                                                                       [D+1][N][C] buffer as K blocks
// Data layout in HBM (all fp32, caller-provided workspace):
//   mid   [D+1][N][C]   skip-connection maps; slot i+1 is written by merge_i, gathered by NNConv_{i+1}
//                       and read as K-block i+1 by the final GEMM -- torch.cat never happens;
//   a1    [N][C]        pre-BN GraphConv output of the current layer;
//   a2    2 x [N][C]    pre-BN CollConv outputs, ping-pong: GIN_{i+1} reads a2[i] + stat2[i] and applies
//                       the BatchNorm inside its neighbourhood sum (affine => commutes with the sum);
//   wtab  [D][T][C*C]   per-layer per-edge-type NNConv matrices;
//   BN partials (fp64) and stat records.
#include <stdarg.h>
#include <string.h>

#include <string>
#include <vector>

#include "tgnn_common.h"

namespace tgnn {
int spin_error_collect_stale(hipStream_t s);   // forward_small.hip (forward_persist.h)
}

namespace tgnn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- parameter table ---------------------------------------------------------------------
// Order = registration order of the reference modules, minus the aliased
// '<p>.nnConv.nn.*' entries (same tensors as '<p>.mlp.*', edge_conv.py:17-18).
static void bn_names(std::vector<std::string> &v, const std::string &p) {
    v.push_back(p + ".weight");
    v.push_back(p + ".bias");
    v.push_back(p + ".running_mean");
    v.push_back(p + ".running_var");
    v.push_back(p + ".num_batches_tracked");
}
static void lt_names(std::vector<std::string> &v, const std::string &p, bool bn) {
    v.push_back(p + ".linear.weight");
    v.push_back(p + ".linear.bias");
    if (bn) bn_names(v, p + ".batch_norm");
}

static std::vector<std::string> param_names(const tgnn_model_dims &d) {
    std::vector<std::string> v;
    for (int l = 0; l < 2; ++l) lt_names(v, "init_node_feature_trans.mlp." + std::to_string(l), true);
    for (int i = 0; i < d.network_depth; ++i) {
        const std::string p1 = "brch_1_graph_conv_layers." + std::to_string(i);
        for (int l = 0; l < 3; ++l) lt_names(v, p1 + ".mlp.mlp." + std::to_string(l), false);
        v.push_back(p1 + ".nnConv.root");
        v.push_back(p1 + ".nnConv.bias");
        bn_names(v, p1 + ".batch_norm");
        const std::string p2 = "brch_2_coll_conv_layers." + std::to_string(i);
        v.push_back(p2 + ".ginConv.eps");
        for (int l = 0; l < 3; ++l) lt_names(v, p2 + ".ginConv.nn.mlp." + std::to_string(l), false);
        bn_names(v, p2 + ".batch_norm");
    }
    for (int l = 0; l < 4; ++l) lt_names(v, "final_mlp.0.mlp." + std::to_string(l), true);
    lt_names(v, "final_mlp.1", false);
    return v;
}

static bool dims_ok(const tgnn_model_dims *d) {
    return d && d->node_features_dim >= 1 && d->adj_edge_features_dim >= 1 && d->adj_edge_features_dim <= 1024 &&
           d->network_width >= 4 && d->network_width % 4 == 0 && d->network_width <= 256 && d->network_depth >= 1 &&
           d->network_depth <= kMaxDepth && d->output_dim >= 1 && d->output_dim <= 256;
}

struct Workspace {
    float *mid, *a1, *a2[2], *t0, *f1, *f2, *f3, *f4, *wtab, *wimg;
    double *part1, *part2, *partf;
    float *small_pack;            // small-layout kernel: per-layer parameter packs, its partial rows, its barrier counter
    double *small_part, *small_part_wide, *small_runstat;
    double *mid_part;             // mid-size persistent layer loop: tagged partial rows + group sums (forward_mid.hip)
    unsigned *small_ctr;
    void *dimg[4];                 // fp16-pair operand images of the final MLP's four Linears (dense_f16_image_build; [r6] the fourth, 64 -> 32)
    unsigned *bounds;            // [0, D]: max |middle[k]| as float bits; [D + 1, 2 D]: max |root_i|; [2 D + 1]: max |final W_0|
    float *stat1, *stat2[2], *stat_i[2], *stat_f[4];
    size_t bytes;
};

// ---- optional per-kernel-class timing (tgnn_forward_profiled) -----------------------------
struct Prof {
    hipStream_t s;
    std::vector<hipEvent_t> ev;      // pairs
    std::vector<int> slot;
    bool on = false;
    unsigned class_mask = ~0u;       // two-stream mode: only these classes (main-stream kernels) are bracketed
    bool two_stream = false, open = false;
    unsigned long long *stamps = nullptr;   // tgnn_forward_stamped: [depth][2] device words for the NNConv launches' wall-clock stamps
    void begin(int sl) {
        open = on && ((class_mask >> sl) & 1u);
        if (!open) return;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, s);
        ev.push_back(a);
        ev.push_back(b);
        slot.push_back(sl);
    }
    void end() {
        if (!open) return;
        (void)hipEventRecord(ev.back(), s);
        open = false;
    }
    void collect(float *ms_out, int32_t *count_out) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        for (size_t i = 0; i < slot.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            ms_out[slot[i]] += ms;
            count_out[slot[i]] += 1;
            (void)hipEventDestroy(ev[2 * i]);
            (void)hipEventDestroy(ev[2 * i + 1]);
        }
    }
};

static const int kFinalDims[4] = {256, 128, 64, 0};  // TilinGNN.py:46 hidden_layer_dims; [3] = C
constexpr int kCarveTypes = 16;                      // (= tgnn_nnconv_cols_max_types(): what the matrix-core NNConv kernels take)
static std::atomic<int> g_split_f16{1};              // tgnn_set_split_precision
static std::atomic<int> g_nnconv_eg{1};              // tgnn_set_nnconv_eg
static std::atomic<int> g_lean_head{3};              // tgnn_set_lean_head: bit 0 the head without memsets / early edge-weight event, bit 1 the fused init MLP
static std::atomic<int64_t> g_path_count[3];          // forwards queued on the general schedule / small-layout kernel / mid-size kernel

// n = rows this device computes; nr >= n = rows of the buffers that are GATHERED from (owned rows, then halo rows
// of other shards; nr == n on a single device)
static Workspace carve(const tgnn_model_dims &d, int64_t n, int64_t nr, int32_t n_types, void *ws, size_t ws_bytes) {
    Carver cv(ws, ws_bytes);
    const int c = d.network_width, D = d.network_depth;
    Workspace w{};
    w.mid = cv.take<float>((size_t)(D + 1) * nr * c);
    w.a1 = cv.take<float>((size_t)nr * c);
    w.a2[0] = cv.take<float>((size_t)nr * c);
    w.a2[1] = cv.take<float>((size_t)nr * c);
    w.t0 = cv.take<float>((size_t)n * c);
    w.f1 = cv.take<float>((size_t)n * 256);
    w.f2 = cv.take<float>((size_t)n * 128);
    w.f3 = cv.take<float>((size_t)n * 64);
    w.f4 = cv.take<float>((size_t)n * c);
    // [r6] the two type-dependent pieces are sized for at least kCarveTypes types: every other piece then sits at the same offset
    // whatever the layout's type count turns out to be -- tgnn_forward_begin fills its pieces before the preparation has counted
    const int tc = n_types < kCarveTypes ? kCarveTypes : n_types;
    w.wtab = cv.take<float>((size_t)D * tc * c * c);
    w.wimg = cv.take<float>((size_t)D * (tc + 1) * kWtType);  // MFMA operand images of the column NNConv
    w.part1 = cv.take<double>((size_t)TGNN_BN_MAX_PARTIALS * 2 * c);
    w.part2 = cv.take<double>((size_t)TGNN_BN_MAX_PARTIALS * 2 * c);
    w.partf = cv.take<double>((size_t)TGNN_BN_MAX_PARTIALS * 2 * 256);
    w.small_pack = cv.take<float>(small_pack_floats(D));
    w.small_part = cv.take<double>((size_t)256 * 128);
    w.small_part_wide = cv.take<double>((size_t)2 * 256 * 512);
    w.small_runstat = cv.take<double>((size_t)D * 128);
    w.mid_part = cv.take<double>(mid_part_doubles());
    w.small_ctr = cv.take<unsigned>(64);
    w.bounds = cv.take<unsigned>(2 * kMaxDepth + 100);
    {
        const int fd[4] = {c * (D + 1), kFinalDims[0], kFinalDims[1], kFinalDims[2]};
        for (int l = 0; l < 3; ++l) w.dimg[l] = cv.take<unsigned char>(dense_f16_image_size(fd[l], fd[l + 1]));
        w.dimg[3] = cv.take<unsigned char>(dense_f16_image_size(fd[3], c));
    }
    w.stat1 = cv.take<float>(4 * c);
    w.stat2[0] = cv.take<float>(4 * c);
    w.stat2[1] = cv.take<float>(4 * c);
    w.stat_i[0] = cv.take<float>(4 * c);
    w.stat_i[1] = cv.take<float>(4 * c);
    for (int l = 0; l < 4; ++l) w.stat_f[l] = cv.take<float>(4 * 256);
    w.bytes = cv.off + 256;
    return w;
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int tgnn_version(void) { return TGNN_VERSION; }
extern "C" int32_t tgnn_set_split_precision(int32_t mode) {
    if (mode != 0 && mode != 1) return g_split_f16.load();
    return g_split_f16.exchange(mode);
}

extern "C" int32_t tgnn_set_nnconv_eg(int32_t on) {
    if (on != 0 && on != 1) return g_nnconv_eg.load();
    return g_nnconv_eg.exchange(on);
}
extern "C" int32_t tgnn_set_lean_head(int32_t bits) {
    if (bits < 0 || bits > 31) return g_lean_head.load();
    return g_lean_head.exchange(bits);
}
extern "C" const char *tgnn_last_error(void) { return g_err; }
extern "C" void tgnn_forward_path_counts(int64_t *out3) {
    if (!out3) return;
    for (int k = 0; k < 3; ++k) out3[k] = g_path_count[k].load(std::memory_order_relaxed);
}

extern "C" int32_t tgnn_param_count(const tgnn_model_dims *dims) {
    if (!dims_ok(dims)) return -1;
    return 2 * kInitStride + dims->network_depth * kLayerStride + 4 * kFinalStride + 2;
}

extern "C" int tgnn_param_name(const tgnn_model_dims *dims, int32_t index, char *buf, size_t buf_len) {
    TGNN_CHECK_ARG(dims_ok(dims), "model dims");
    TGNN_CHECK_ARG(buf && buf_len > 0, "buffer");
    const std::vector<std::string> names = param_names(*dims);
    TGNN_CHECK_ARG(index >= 0 && index < (int)names.size(), "index");
    TGNN_CHECK_ARG(names[index].size() + 1 <= buf_len, "buffer too small");
    memcpy(buf, names[index].c_str(), names[index].size() + 1);
    return TGNN_OK;
}

extern "C" size_t tgnn_forward_workspace_bytes(const tgnn_model_dims *dims, int64_t n_nodes, int32_t n_types) {
    if (!dims_ok(dims) || n_nodes < 0) return 0;
    return carve(*dims, n_nodes, n_nodes, n_types, nullptr, 0).bytes;
}

extern "C" size_t tgnn_forward_sharded_workspace_bytes(const tgnn_model_dims *dims, int64_t n_own, int64_t n_rows,
                                                       int32_t n_types) {
    if (!dims_ok(dims) || n_own < 0 || n_rows < n_own) return 0;
    return carve(*dims, n_own, n_rows, n_types, nullptr, 0).bytes;
}

#define TGNN_TRY(expr)               \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != TGNN_OK) return rc__; \
    } while (0)

// [r6] What the general schedule does in front of its first layer that needs NOTHING of the graph: the operands' bounds, the init
// MLP (middle[0] and its bound), the final MLP's bounds and operand images.  Either inside forward_impl or -- tgnn_forward_begin --
// before the layout is prepared, on the side stream, beside the preparation's launches.
struct HeadEvent {
    hipEvent_t ev = nullptr;
    const void *ws = nullptr;        // what the last tgnn_forward_begin of this thread and device filled
    int64_t n = 0;
    bool weights = false;            // tgnn_forward_begin_weights has queued the edge weights (edge-group images, device-side type count)
};
static thread_local HeadEvent g_head[64];
// [r6] tgnn_forward_small_prepass: the workspace whose edge weights (bf16 x 3 images, type count read on the device) and parameter
// pack are queued on the caller's stream already -- what a small layout's forward does in front of its one persistent kernel
struct SmallPre {
    const void *ws = nullptr;
    int64_t n = 0;
};
static thread_local SmallPre g_small_pre[64];
// words of w.bounds holding (max |W_l|, bound of |BN(input of l)|) of the final MLP's layer l = 1 .. 3 (3: lean head only)
static inline unsigned *final_bound_word(const Workspace &w, int D, int l) { return w.bounds + (l <= 2 ? 2 * D + 2 + 2 * (l - 1) : 2 * D + 8); }
static int forward_head_bounds_images(const tgnn_model_dims *dims, const Params &P, const Workspace &w, int64_t n, int64_t n_total,
                                      hipStream_t st, bool *images_ok) {
    const int c = dims->network_width, D = dims->network_depth;
    const int fin_dims[5] = {c * (D + 1), kFinalDims[0], kFinalDims[1], kFinalDims[2], c};
    unsigned *dense_max = w.bounds + 2 * D + 1;
    // layers 1 .. 3 (256 -> 128 -> 64 -> 32): the weights' bound and, from the producer's BatchNorm parameters, the input's; layer 3
    // ([r6]: it ran the exact-fp32 kernel, 17.6 us for 38 MB) only where the row count takes the resident kernel
    const int nl = (c == 32 && n >= kDenseRowsKernelMin) ? 4 : 3;
    const float *bw[4], *bg[4], *bb[4];
    int64_t bwn[4];
    int bf[4];
    unsigned *bwm[4], *bam[4];
    bw[0] = P.f(P.fin(0)); bwn[0] = (int64_t)fin_dims[0] * fin_dims[1]; bg[0] = nullptr; bb[0] = nullptr; bf[0] = 0; bwm[0] = dense_max; bam[0] = nullptr;
    for (int l = 1; l < nl; ++l) {
        const BnPtrs bp = P.bn(P.fin(l - 1) + 2);
        bw[l] = P.f(P.fin(l));
        bwn[l] = (int64_t)fin_dims[l] * fin_dims[l + 1];
        bg[l] = bp.gamma; bb[l] = bp.beta; bf[l] = fin_dims[l];
        bwm[l] = final_bound_word(w, D, l);
        bam[l] = final_bound_word(w, D, l) + 1;
    }
    launch_dense_bounds(nl, bw, bwn, bg, bb, bf, bwm, bam, n_total, st);
    *images_ok = false;
    if (nl == 4) {
        const float *iw[4];
        int iin[4], iout[4];
        const unsigned *iwm[4];
        void *iimg[4];
        for (int l = 0; l < 4; ++l) {
            iw[l] = P.f(P.fin(l)); iin[l] = fin_dims[l]; iout[l] = fin_dims[l + 1];
            iwm[l] = l == 0 ? dense_max : final_bound_word(w, D, l);
            iimg[l] = w.dimg[l];
        }
        *images_ok = dense_f16_images_build(4, iw, iin, iout, iwm, iimg, st) == TGNN_OK;
    }
    return TGNN_OK;
}
static int forward_head_init(const tgnn_model_dims *dims, const Params &P, const float *x, const Workspace &w, int64_t n,
                             int32_t update_running, unsigned *slot_max, hipStream_t st) {
    const int fx = dims->node_features_dim;
    const int ib = init_mlp_fused_blocks(n);
    BnJob j0 = BnJob{w.partf, ib, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, w.stat_i[0]};
    BnJob j1 = BnJob{w.partf + (size_t)TGNN_BN_MAX_PARTIALS * 64, ib, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, w.stat_i[1]};
    const BnPtrs b0 = P.bn(P.init(0) + 2), b1 = P.bn(P.init(1) + 2);
    j0.gamma = b0.gamma; j0.beta = b0.beta; j1.gamma = b1.gamma; j1.beta = b1.beta;
    if (update_running) {
        j0.running_mean = b0.rm; j0.running_var = b0.rv; j0.num_batches_tracked = b0.nbt;
        j1.running_mean = b1.rm; j1.running_var = b1.rv; j1.num_batches_tracked = b1.nbt;
    }
    return launch_init_mlp_fused(x, fx, fx, P.f(P.init(0)), P.f(P.init(0) + 1), P.f(P.init(1)), P.f(P.init(1) + 1), j0, j1, n, 1e-5f, 0.1f,
                                 w.mid, slot_max, st);
}

static int forward_impl(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                        const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                        int32_t use_running_stats, float *probs, void *ws, size_t ws_bytes, tgnn_stream_t stream,
                        tgnn_stream_t stream2, Prof &prof, const tgnn_shard *sh = nullptr,
                        const tgnn_train_save *keep = nullptr, bool head_done = false) {
    TGNN_CHECK_ARG(dims_ok(dims), "model dims");
    TGNN_CHECK_ARG(params_host && graph && probs && x, "null pointer");
    // (bit 1 of update_running: the init MLP's running statistics have this forward's update already -- a tgnn_forward_begin whose
    //  work could not be picked up, tgnn.h)
    const bool init_running_done = (update_running & 2) != 0;
    update_running &= 1;
    const int64_t n = graph->n_nodes;
    TGNN_CHECK_ARG(n >= 1, "empty graph");
    TGNN_CHECK_ARG(use_running_stats || sh || n >= 2, "train-mode BatchNorm needs more than one row");
    TGNN_CHECK_ARG(graph->adj_rowptr && graph->col_rowptr, "graph pointers");
    TGNN_CHECK_ARG(graph->n_types == 0 || (adj_edge_attr && graph->type_rep_edge && graph->adj_src && graph->adj_type),
                   "adjacency pointers");
    const int np = tgnn_param_count(dims);
    for (int i = 0; i < np; ++i)
        if (!params_host[i]) {
            set_error("tgnn_forward: params_host[%d] is null", i);
            return TGNN_ERR_INVALID_ARG;
        }
    // ---- sharded mode (tgnn_forward_sharded): this device owns rows [0, n) of buffers that carry nr - n halo rows
    // of other shards behind them; BatchNorm statistics are summed over all shards, halo rows are exchanged once
    // per layer.  The collectives are the caller's (callbacks, enqueued on / ordered with `stream`).
    const int64_t nr = sh ? sh->n_rows : n, n_total = sh ? sh->n_total : n, n_halo = nr - n;
    if (sh) {
        TGNN_CHECK_ARG(sh->n_own == n && nr >= n && n_total >= n, "shard row counts");
        TGNN_CHECK_ARG(!use_running_stats, "sharded forward runs in train mode");
        TGNN_CHECK_ARG(sh->sum_buf && sh->send_buf && sh->recv_buf, "shard buffers");
        TGNN_CHECK_ARG(sh->rccl_comm ? (sh->send_counts && sh->recv_counts && sh->world >= 1 && sh->world <= 64)
                                     : (sh->allreduce_f64 && sh->alltoall_rows), "shard communicator / callbacks");
        TGNN_CHECK_ARG(sh->n_send == 0 || sh->send_idx, "send_idx");
    }
    // a persistent kernel of an earlier call that gave up and whose failure nobody collected (forward_persist.h): loud, here
    TGNN_TRY(spin_error_collect_stale(static_cast<hipStream_t>(stream)));
    Workspace w = carve(*dims, n, nr, graph->n_types, ws, ws_bytes);
    if (!ws || w.bytes > ws_bytes) {
        set_error("tgnn_forward: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return TGNN_ERR_WORKSPACE;
    }
    if (keep) {
        // training forward (tgnn_forward_train): what the backward reads lives in the caller's buffers instead of the
        // rotating workspace ones -- the same kernels, the same schedule, only the destinations differ
        TGNN_CHECK_ARG(!sh && !use_running_stats, "the training forward is single-device, train mode");
        TGNN_CHECK_ARG(keep->skip && keep->wtab && keep->a1 && keep->a2 && keep->u && keep->stat1 && keep->stat2, "keep buffers");
        for (int l = 0; l < 2; ++l) TGNN_CHECK_ARG(keep->init_a[l] && keep->init_stat[l], "keep buffers (init)");
        for (int l = 0; l < 4; ++l) TGNN_CHECK_ARG(keep->fin_a[l] && keep->fin_stat[l], "keep buffers (final)");
        w.mid = keep->skip;
        w.wtab = keep->wtab;
        w.t0 = keep->init_a[0];
        w.a1 = keep->init_a[1];
        w.stat_i[0] = keep->init_stat[0];
        w.stat_i[1] = keep->init_stat[1];
        w.f1 = keep->fin_a[0]; w.f2 = keep->fin_a[1]; w.f3 = keep->fin_a[2]; w.f4 = keep->fin_a[3];
        for (int l = 0; l < 4; ++l) w.stat_f[l] = keep->fin_stat[l];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool weights_early = false;
    bool small_pre = false;
    {
        int devp = 0;
        if (hipGetDevice(&devp) == hipSuccess && devp >= 0 && devp < 64 && g_small_pre[devp].ws) {
            small_pre = g_small_pre[devp].ws == ws && g_small_pre[devp].n == n && !head_done && !sh && !keep;
            g_small_pre[devp].ws = nullptr;
        }
    }
    if (head_done) {
        int dev = 0;
        TGNN_CHECK_HIP(hipGetDevice(&dev));
        TGNN_CHECK_ARG(dev >= 0 && dev < 64 && g_head[dev].ev && g_head[dev].ws == ws && g_head[dev].n == n,
                       "tgnn_forward_resume without a matching tgnn_forward_begin (same thread, device, workspace, node count)");
        g_head[dev].ws = nullptr;
        weights_early = g_head[dev].weights;
        g_head[dev].weights = false;
        TGNN_CHECK_HIP(hipStreamWaitEvent(s, g_head[dev].ev, 0));   // middle[0], the bounds, the final MLP's images: done long ago (the preparation ran meanwhile)
    }
    // Two-chain schedule: the collision branch is a chain of its own -- CollConv_i reads only CollConv_{i-1}
    // (TilinGNN.py:63); the branches meet in the product of :64 only.  With a side stream the whole GIN chain runs
    // free beside the NNConv chain and fills the GPU wherever the latter leaves it idle (1-block BN finalizes, the
    // HBM-bound merge, kernel tails); it is held back only by the two-deep buffers it shares with merge.
    hipStream_t s2 = (prof.on && !prof.two_stream) ? nullptr : static_cast<hipStream_t>(sh ? sh->side_stream : stream2);
    if (s2 == s || (sh && dims->network_width != 32)) s2 = nullptr;
    // Events of the two-chain schedule, per calling thread and device; created once, never destroyed.
    // [0] init done, [1 + i] GIN_i done, [1 + kMaxDepth + i] merge_i done, then: fork at entry, edge weights done
    constexpr int kEvPerDev = 3 + 2 * kMaxDepth, kEvFork = 1 + 2 * kMaxDepth, kEvWeights = 2 + 2 * kMaxDepth;
    static thread_local hipEvent_t ev_cache[64][kEvPerDev] = {};
    hipEvent_t *ev = nullptr;
    if (s2) {
        int dev = 0;
        TGNN_CHECK_HIP(hipGetDevice(&dev));
        TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
        if (!ev_cache[dev][0])
            for (int k = 0; k < kEvPerDev; ++k)
                TGNN_CHECK_HIP(hipEventCreateWithFlags(&ev_cache[dev][k], hipEventDisableTiming));
        ev = ev_cache[dev];
    }
    prof.s = s;
    const int c = dims->network_width, D = dims->network_depth, fx = dims->node_features_dim,
              fe = dims->adj_edge_features_dim, T = graph->n_types;
    const Params P{params_host, D};
    const float eps = 1e-5f, momentum = 0.1f;  // torch.nn.BatchNorm1d defaults
    const int fin_mode = use_running_stats ? 3 : 0;
    int32_t np1 = 0, np2 = 0;

    // BatchNorm statistics from the producers' partial rows.  Sharded: partials -> local sums (mode 1) -> all-reduce
    // over the shards (caller's callback; both BatchNorms of a layer travel in one message) -> stat record (mode 2).
    // the two collectives: RCCL calls of the library's own when the shard carries a communicator, else the caller's callbacks
    auto allreduce = [&](double *buf, int64_t count, hipStream_t st) -> int {
        if (sh->rccl_comm) return rccl_allreduce_f64(sh->rccl_comm, buf, count, st);
        if (sh->allreduce_f64(sh->ctx, buf, count, st) != 0) {
            set_error("tgnn_forward_sharded: the all-reduce callback failed");
            return TGNN_ERR_INVALID_ARG;
        }
        return TGNN_OK;
    };
    auto alltoall = [&](const float *send, float *recv, int row_floats, int extra_rows, hipStream_t st) -> int {
        if (sh->rccl_comm) {
            void *comm = (st != s && sh->rccl_comm_side) ? sh->rccl_comm_side : sh->rccl_comm;
            return rccl_alltoall_rows(comm, send, recv, sh->send_counts, sh->recv_counts, sh->world, row_floats, extra_rows, st);
        }
        if (sh->alltoall_rows(sh->ctx, send, recv, row_floats, extra_rows, st) != 0) {
            set_error("tgnn_forward_sharded: the all-to-all callback failed");
            return TGNN_ERR_INVALID_ARG;
        }
        return TGNN_OK;
    };
    auto finalize_jobs = [&](BnJobs jobs, int nj, int f) -> int {
        if (nj == 0) return TGNN_OK;
        if (!sh) {
            prof.begin(4);
            launch_bn_finalize(jobs, nj, fin_mode, f, n, eps, momentum, s);
            prof.end();
            return TGNN_OK;
        }
        for (int j = 0; j < nj; ++j) jobs.job[j].sums = sh->sum_buf + (size_t)j * 2 * f;
        launch_bn_finalize(jobs, nj, 1, f, n_total, eps, momentum, s);
        TGNN_TRY(allreduce(sh->sum_buf, (int64_t)nj * 2 * f, s));
        launch_bn_finalize(jobs, nj, 2, f, n_total, eps, momentum, s);
        return TGNN_OK;
    };
    auto finalize1 = [&](double *part, int nparts, int f, const BnPtrs &b, float *stat) -> int {
        BnJobs jobs{};
        jobs.job[0] = BnJob{part, nparts, nullptr, b.gamma, b.beta, (update_running || use_running_stats) ? b.rm : nullptr,
                            (update_running || use_running_stats) ? b.rv : nullptr,
                            (update_running && !use_running_stats) ? b.nbt : nullptr, stat};
        return finalize_jobs(jobs, 1, f);
    };
    // Halo exchange: the rows other shards need (send_idx) of slot `slot` of the skip buffer and, from layer 1 on,
    // of the collision branch's pre-BN activations travel in ONE all-to-all (64 floats per row) and land behind the
    // owned rows.
    auto exchange = [&](int slot, const float *a2_own, float *a2_halo_dst) -> int {
        if (!sh) return TGNN_OK;
        const int rf = a2_own ? 2 * c : c;                     // floats per exchanged row
        float *slot_rows = w.mid + (size_t)slot * nr * c;
        if (sh->n_send > 0) {
            TGNN_TRY(tgnn_rows_gather(slot_rows, c, sh->send_idx, sh->n_send, c, sh->send_buf, rf, s));
            if (a2_own) TGNN_TRY(tgnn_rows_gather(a2_own, c, sh->send_idx, sh->n_send, c, sh->send_buf + c, rf, s));
        }
        TGNN_TRY(alltoall(sh->send_buf, sh->recv_buf, rf, 0, s));
        if (n_halo > 0) {
            TGNN_CHECK_HIP(hipMemcpy2DAsync(slot_rows + (size_t)n * c, (size_t)c * 4, sh->recv_buf, (size_t)rf * 4,
                                            (size_t)c * 4, (size_t)n_halo, hipMemcpyDeviceToDevice, s));
            if (a2_own)
                TGNN_CHECK_HIP(hipMemcpy2DAsync(a2_halo_dst + (size_t)n * c, (size_t)c * 4, sh->recv_buf + c, (size_t)rf * 4,
                                                (size_t)c * 4, (size_t)n_halo, hipMemcpyDeviceToDevice, s));
        }
        return TGNN_OK;
    };

    // ---- K1: per-type NNConv matrices of all layers, one launch -- on the side stream when there is one: it idles until
    //      the init MLP is through, and these ~30 us (serial 3-layer MLP per (layer, type) + the bf16 image) then leave the
    //      critical chain; the first NNConv waits for them.
    hipStream_t sw = s;
    constexpr bool weights_on_side = true;
    const bool addr_ok = c == 32 && (int64_t)nr * c * 4 < (int64_t(1) << 31);   // buffer-addressed gathers
    const bool cols_ok = graph->nn_tile_col_ptr && addr_ok;
    const bool groups_ok = graph->nn_tile_grp_ptr && graph->nn_grp && addr_ok && graph->nn_max_in_degree <= 2048 && g_nnconv_eg;
    // Small layouts: the layer loop below is replaced by one persistent kernel (forward_small.hip)
    const int small_teams = (cols_ok && !sh && !keep && !use_running_stats && !prof.on && nr == n) ? small_layout_teams(dims, n, T, graph->nn_max_in_degree) : 0;
    const int64_t cat_w_floats = (int64_t)c * (D + 1) * kFinalDims[0];
    const int fin_dims[5] = {c * (D + 1), kFinalDims[0], kFinalDims[1], kFinalDims[2], c};   // in / out widths of the final MLP's layers
    // fp16-pair operands (3 matrix terms instead of the 6 of bf16 x 3) wherever a bound of the operand is at hand: the kernels
    // that write a slot of the skip buffer leave its largest magnitude (w.bounds), one launch up front those of the root
    // matrices and of the final MLP's first Linear.  General schedule, train-mode BatchNorm; needs the layout's largest
    // in-degree.  Sharded: with one all-to-all per layer every shard merges its own AND its halo rows itself, so the bound it
    // leaves covers every row its NNConv gathers (the shards' scales may differ -- powers of two, taken off again: each
    // shard's results are as accurate as a single device's); the all-reduce + all-to-all scheme stays on bf16 x 3.
    const bool fused_shard = sh && sh->send_idx_fused && sh->recv_idx_fused && sh->world >= 1 && c == 32;
    const bool f16 = g_split_f16 && (cols_ok || groups_ok) && (!sh || fused_shard) && !small_teams && !use_running_stats &&
                     graph->nn_max_in_degree >= 1 && D <= kMaxDepth && (cat_w_floats % 4) == 0;
    unsigned *slot_max = f16 ? w.bounds : nullptr, *root_max = f16 ? w.bounds + D + 1 : nullptr,
             *dense_max = f16 ? w.bounds + 2 * D + 1 : nullptr;
    // Mid-size layouts (above the small-layout limit, up to 65 536 nodes): the D layers between the init and the final MLP are
    // ONE persistent kernel carrying both chains (forward_mid.hip) instead of ~5 dependent launches per layer on two streams
    int mid_blocks = 0;
    const int mid_k = (f16 && !sh && !keep && !prof.on && nr == n) ? mid_layout_tiles_per_block(dims, graph, n, &mid_blocks) : 0;
    // The persistent kernels, with CUs to spare for the edge-weight kernel's blocks, do not wait for the edge weights on the host's
    // event (~12 us of cross-queue latency on a launch-bound forward) but on a counter of that kernel's finished blocks: the small
    // kernel's init MLP, the mid kernel's first collision layer run beside them
    // (the mid path's counter is a word of w.bounds that launch_forward_scales below zeroes anyway: no launch of its own)
    // ... and the final MLP behind it as one more persistent kernel (forward_tail.hip) instead of 5 + 4 launches
    int tail_blocks = 0;
    const int tail_k = mid_k ? mid_tail_tiles_per_block(dims, n, &tail_blocks) : 0;
    // the NNConv of the general schedule over the layout's edge groups (nnconv_eg.hip) where the graph carries them; its
    // operand images are the fp16-pair ones with one more power of two on the weights
    const bool eg = f16 && groups_ok && !mid_k;
    const bool tiled = cols_ok || eg;
    const bool mid_counter = mid_k && mid_blocks + 16 <= device_cus() && s2 && weights_on_side;
    // the init MLP in that kernel's prologue instead of 5 launches -- where the kernel starts beside the edge-weight kernel (with
    // a block on every CU it starts BEHIND it, a cross-queue event later, and the launches, which run beside it, win: measured at
    // 16 384 and 32 768 nodes, profiles/r05_mid_tail.txt)
    const bool mid_init = mid_counter && mid_init_in_kernel() && fx <= 8;
    // [r6] tgnn_graph.nn_mid_verdict: a forward queued behind a preparation whose batches are not verified yet -- honoured where the
    // mid-size forward is the two persistent kernels alone (they read the word and leave); anywhere else nothing is queued
    if (mid_k && graph->nn_mid_verdict && !(mid_init && tail_k)) {
        set_error("tgnn_forward: nn_mid_verdict given, but the mid-size forward of this layout is not the two persistent kernels alone");
        return TGNN_ERR_UNVERIFIED;
    }
    // [r6] a small layout whose pre-pass is on `stream` already (tgnn_forward_small_prepass): nothing to wait for, nothing to queue
    const bool small_pre_used = small_pre && small_teams && c == 32 && T <= kCarveTypes && cols_ok && edge_weight_table_device_count_ok(fe, c);
    unsigned *weights_done = mid_counter ? w.bounds + 2 * D + 6 : (small_teams == 2 && s2 && weights_on_side && !small_pre_used) ? w.small_ctr + 16 : nullptr;
    if (weights_done && !mid_counter) TGNN_CHECK_HIP(hipMemsetAsync(weights_done, 0, 4, s));
    const unsigned weights_target = edge_weight_table_blocks(T, fe, D, c, tiled);
    // [r6] the general schedule's head: no memset anywhere (a hipMemsetAsync is two fill kernels and ~10 us in front of the first
    // launch) -- the scales kernel clears the words, among them the collision branch's fold counter; the first Linear's bound is
    // taken on the side stream with the other two (nobody needs it before the final MLP)
    const bool lean_head = f16 && !mid_k && !small_teams && (g_lean_head.load(std::memory_order_relaxed) & 1);
    // (what the init MLP's fused form needs is known here already: tgnn_forward_begin's work is used if and only if both hold)
    const bool init_fused_early = c == 32 && fx <= 8 && !sh && !keep && !use_running_stats && (g_lean_head.load(std::memory_order_relaxed) & 2);
    const bool head_used = head_done && lean_head && init_fused_early;
    // (17 words: gin32_mlp_kernel folds the collision branch's BatchNorm in two levels -- a ticket per row group and one over the groups)
    unsigned *fold_ctr = lean_head ? w.bounds + 2 * D + 10 : w.small_ctr + 32;
    if (f16) {
        const float *roots[kMaxDepth];
        for (int i = 0; i < D; ++i) roots[i] = P.f(P.layer(i) + 6);
        if (lean_head && head_done && init_fused_early)
            ;                                                 // (tgnn_forward_begin's, on the side stream)
        else if (lean_head)
            launch_forward_scales(w.bounds, 2 * D + 95, roots, D, root_max, nullptr, 0, nullptr, s);
        else
            launch_forward_scales(w.bounds, 2 * D + 7, roots, D, root_max, P.f(P.fin(0)), cat_w_floats, dense_max, s);   // (before the fork: both chains see the zeroed words)
    }
    // [r6] tgnn_forward_resume picking up tgnn_forward_begin's work: `stream` has nothing to do in front of the first NNConv but wait
    // for the preparation's last launches, so the edge weights go on IT, straight behind them (no cross-queue hand-over in front of
    // them and none in front of the first NNConv: ~12 us each), and the collision chain is released first -- its first aggregate
    // runs beside them instead of behind them
    const bool weights_on_main = head_used && s2 && !sh && !keep && !mid_k && !small_teams && (g_lean_head.load(std::memory_order_relaxed) & 8) == 0;
    if (small_pre_used) {
        // (everything in front of the persistent kernel is on `stream`)
    } else if (weights_on_main) {
        TGNN_CHECK_HIP(hipEventRecord(ev[0], s));            // middle[0] (the head's event, waited for above) and the layout are complete
        TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[0], 0));
    } else if (s2 && weights_on_side) {
        TGNN_CHECK_HIP(hipEventRecord(ev[kEvFork], s));     // everything the caller queued on `stream` so far
        TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[kEvFork], 0));
        sw = s2;
    }
    // (first on the side stream: the layer loop waits for these, the final MLP's bounds and images have the whole loop's time)
    // [r6] tgnn_forward_begin_weights has queued exactly this launch behind the preparation already (type count read on the device)
    const bool weights_queued = weights_early && weights_on_main && eg && T <= kCarveTypes && !weights_done && edge_weight_table_device_count_ok(fe, c);
    if ((T > 0 || tiled) && !weights_queued && !small_pre_used) {
        // edge MLP of every (layer, type) and, for the matrix-core NNConv, its operand images (root = pseudo-type T): one launch
        EdgeMlpLayers layers{};
        const float *roots[kMaxDepth];
        for (int i = 0; i < D; ++i) {
            const int b = P.layer(i);
            layers.l[i] = EdgeMlpLayer{P.f(b), P.f(b + 1), P.f(b + 2), P.f(b + 3), P.f(b + 4), P.f(b + 5)};
            roots[i] = P.f(b + 6);
        }
        prof.begin(0);
        launch_edge_weight_table_batched(adj_edge_attr, graph->type_rep_edge, T, fe, layers, D, c, w.wtab, tiled ? roots : nullptr,
                                         tiled ? w.wimg : nullptr, sw, weights_done, root_max, eg ? kEgImageScale : 1.0f);
        prof.end();
    }
    // [r6] the layer loop waits for the edge weights alone: the event sits in front of the final MLP's bounds and images (they
    // are joined with the collision chain, which the last merge waits for) -- 12 us of idle main stream in front of the first NNConv
    bool weights_recorded = false;
    if (lean_head && sw != s) {
        TGNN_CHECK_HIP(hipEventRecord(ev[kEvWeights], s2));
        weights_recorded = true;
    }
    bool dimg_ok[4] = {false, false, false, false};
    if (lean_head && head_done && init_fused_early) {
        dimg_ok[0] = dimg_ok[1] = dimg_ok[2] = dimg_ok[3] = c == 32 && n >= kDenseRowsKernelMin;   // (built by tgnn_forward_begin)
    } else if (lean_head) {
        // bounds of the three Linears' weights (+ of layers 1, 2's inputs from their BatchNorm parameters), then the three operand
        // images in ONE launch
        bool ok = false;
        TGNN_TRY(forward_head_bounds_images(dims, P, w, n, n_total, sw, &ok));
        dimg_ok[0] = dimg_ok[1] = dimg_ok[2] = dimg_ok[3] = ok;
    } else if (f16 && !tail_k) {
        // the final MLP's layers 1 and 2 (256 -> 128 -> 64): weights' bounds and, from the BatchNorm parameters alone, their inputs'
        const float *bw[2], *bg[2], *bb[2];
        int64_t bwn[2];
        int bf[2];
        unsigned *bwm[2], *bam[2];
        for (int l = 1; l <= 2; ++l) {
            const BnPtrs bp = P.bn(P.fin(l - 1) + 2);
            bw[l - 1] = P.f(P.fin(l));
            bwn[l - 1] = (int64_t)fin_dims[l] * fin_dims[l + 1];
            bg[l - 1] = bp.gamma; bb[l - 1] = bp.beta; bf[l - 1] = fin_dims[l];
            bwm[l - 1] = w.bounds + 2 * D + 2 + 2 * (l - 1);
            bam[l - 1] = w.bounds + 2 * D + 3 + 2 * (l - 1);
        }
        launch_dense_bounds(2, bw, bwn, bg, bb, bf, bwm, bam, n_total, sw);   // (side stream: off the critical chain; the final MLP is behind every join)
        // the three Linears' operand images for the rows-per-wave kernel (dense.hip), behind their weights' bounds on the same stream
        for (int l = 0; l < 3; ++l) {
            const unsigned *wm = l == 0 ? dense_max : w.bounds + 2 * D + 2 + 2 * (l - 1);
            dimg_ok[l] = c == 32 && n >= kDenseRowsKernelMin && dense_f16_image_build(P.f(P.fin(l)), fin_dims[l], fin_dims[l + 1], wm, w.dimg[l], sw) == TGNN_OK;
        }
    }
    if (small_teams && !small_pre_used) launch_small_pack(P, D, w.small_pack, w.small_ctr, s);   // on the main stream: it has nothing else to do yet
    // (parameter vectors + GIN images of the layers; the same launch clears the barrier counter and the tagged partial rows)
    if (mid_k) launch_small_pack(P, D, w.small_pack, w.small_ctr, s, tail_k > 0 || mid_init, w.mid_part, mid_part_doubles() * sizeof(double), tail_k > 0 ? dense_max : nullptr,
                                 sw != s ? sw : nullptr);   // (the final MLP's images: side stream, joined behind the layer loop)
    if (sw != s && !weights_recorded) TGNN_CHECK_HIP(hipEventRecord(ev[kEvWeights], s2));
    g_path_count[small_teams ? 1 : mid_k ? 2 : 0].fetch_add(1, std::memory_order_relaxed);
    if (small_teams) {
        // init MLP, the layers and the final MLP: one persistent kernel behind the pre-pass
        if (sw != s && !weights_done) TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[kEvWeights], 0));
        TGNN_TRY(launch_forward_small(dims, P, x, probs, w.mid, w.a2[0], w.a2[1], w.wimg, w.small_pack, graph, w.small_part,
                                      w.small_part_wide, w.small_runstat, w.small_ctr, weights_done, weights_target, n,
                                      update_running, eps, momentum, s));
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }

    // ---- K10: init MLP  (TilinGNN.py:54)
    // [r6] three launches that recompute from x instead of five that store (init_mlp.hip); the launch-per-op form stays for what
    // keeps the activations (training), all-reduces the statistics (shards) or normalises with running statistics
    const bool init_fused = !mid_init && c == 32 && fx <= 8 && !sh && !keep && !use_running_stats && (g_lean_head.load(std::memory_order_relaxed) & 2);
    // (a layout that turns out not to take the fp16-pair path -- more than 16 edge types, an in-degree above 2 048 -- runs its head
    //  again, the launch-per-op way; the init MLP's running statistics were updated by tgnn_forward_begin already)
    BnPtrs ibn0 = P.bn(P.init(0) + 2), ibn1 = P.bn(P.init(1) + 2);
    const bool init_stats_written = (head_done && !head_used) || (init_running_done && !use_running_stats);
    if (init_stats_written) {
        ibn0.rm = ibn0.rv = ibn1.rm = ibn1.rv = nullptr;
        ibn0.nbt = ibn1.nbt = nullptr;
    }
    if (head_used) {
    } else if (init_fused) {
        prof.begin(1);
        TGNN_TRY(forward_head_init(dims, P, x, w, n, init_stats_written ? 0 : update_running, slot_max, s));
        prof.end();
    } else if (!mid_init) {
        prof.begin(1);
        TGNN_TRY(tgnn_dense_act_fwd(x, fx, 32, nullptr, P.f(P.init(0)), P.f(P.init(0) + 1), n, fx, c, TGNN_ACT_LEAKY_RELU,
                                    w.t0, c, w.partf, &np1, s));
        prof.end();
        TGNN_TRY(finalize1(w.partf, np1, c, ibn0, w.stat_i[0]));
        prof.begin(1);
        TGNN_TRY(tgnn_dense_act_fwd(w.t0, c, 32, w.stat_i[0], P.f(P.init(1)), P.f(P.init(1) + 1), n, c, c,
                                    TGNN_ACT_LEAKY_RELU, w.a1, c, w.partf, &np1, s));
        prof.end();
        TGNN_TRY(finalize1(w.partf, np1, c, ibn1, w.stat_i[1]));
        prof.begin(1);
        launch_bn_apply(w.a1, c, w.stat_i[1], n, c, w.mid, c, slot_max, s);   // middle[0] = brch_1 = brch_2 (:55,58)
        prof.end();
    }
    TGNN_TRY(exchange(0, nullptr, nullptr));
    if (f16 && n_halo > 0) launch_absmax(w.mid + (size_t)n * c, n_halo * c, slot_max, s);   // (the halo rows of middle[0])

    // ---- main loop (TilinGNN.py:59-71)
    const bool run_stats = update_running || use_running_stats;
    auto bn_job = [&](double *part, int nparts, const BnPtrs &bp, float *stat) {
        return BnJob{part, nparts, nullptr, bp.gamma, bp.beta, run_stats ? bp.rm : nullptr, run_stats ? bp.rv : nullptr,
                     (update_running && !use_running_stats) ? bp.nbt : nullptr, stat};
    };
    // CollConv (:63): input = BN_{i-1}(a2_{i-1}) folded into the gather; layer 0 reads middle[0]
    // one all-to-all per layer instead of all-reduce + all-to-all (see tgnn_shard in tgnn.h)
    if (fused_shard) TGNN_CHECK_ARG(sh->rank >= 0 && sh->rank < sh->world && sh->world <= 64, "shard rank / world (<= 64)");
    const bool split = fused_shard && s2 != nullptr;       // one exchange per branch and layer, the collision branch's on the side stream
    // The collision branch's BatchNorm record by the LAST block of the GIN MLP kernel (gin.hip: GinFin) instead of a 1-block
    // finalize launch behind it -- 20 launches less on that chain (measured by leaving them out: 0.07 ms of 1.95).  Single
    // device, width 32, batch statistics.
#ifdef TGNN_ABL_NOFOLD
    const bool fold_fin2 = false;
#else
    const bool fold_fin2 = c == 32 && !sh && !use_running_stats;
#endif
    if (fold_fin2 && !mid_k && !lean_head) TGNN_CHECK_HIP(hipMemsetAsync(fold_ctr, 0, 17 * sizeof(unsigned), s));   // (before ev[0]: the side chain sees it)
    auto gin_layer = [&](int i, hipStream_t gs) -> int {
        const int b = P.layer(i);
        const float *gin_in = i == 0 ? w.mid : w.a2[(i - 1) & 1];
        const float *gin_stat = i == 0 ? nullptr : w.stat2[(i - 1) & 1];
        if (fold_fin2) {
            GinFin fin{};
            fin.counter = fold_ctr;
            fin.job = bn_job(nullptr, 0, P.bn(b + 20), w.stat2[i & 1]);
            fin.n_total = n;
            fin.eps = eps;
            fin.momentum = momentum;
            prof.begin(3);
            const int rc = gin32_fwd_folded(gin_in, c, gin_stat, graph->col_rowptr, graph->col_src, P.f(b + 13), P.f(b + 14), P.f(b + 15),
                                            P.f(b + 16), P.f(b + 17), P.f(b + 18), P.f(b + 19), n, TGNN_ACT_LEAKY_RELU, w.a2[i & 1],
                                            w.t0, w.part2, &np2, fin, gs, keep != nullptr);
            prof.end();
            if (rc != TGNN_ERR_UNSUPPORTED) return rc;
            return TGNN_ERR_UNSUPPORTED;                       // (the workspace is aligned: cannot happen)
        }
        prof.begin(3);
        TGNN_TRY(tgnn_gin_fwd(gin_in, c, gin_stat, graph->col_rowptr, graph->col_src, P.f(b + 13), P.f(b + 14),
                              P.f(b + 15), P.f(b + 16), P.f(b + 17), P.f(b + 18), P.f(b + 19), n, c,
                              TGNN_ACT_LEAKY_RELU, w.a2[i & 1], w.t0, w.part2, &np2, gs));
        prof.end();
        return TGNN_OK;
    };
    if (mid_k) {
        double *const tail_zero[2] = {w.partf, w.small_part_wide};   // the tail kernel's tagged rows: cleared by the layer loop's blocks
        if (sw != s && !weights_done) TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[kEvWeights], 0));   // (the images of the side stream)
        TGNN_TRY(launch_forward_mid(dims, P, w.mid, w.a1, w.a2[0], w.a2[1], w.wimg, w.small_pack, graph, w.mid_part, w.small_runstat,
                                    w.small_ctr, w.bounds, n, mid_k, mid_blocks, update_running, eps, momentum, s, weights_done,
                                    weights_target, tail_k ? tail_zero : nullptr, mid_tail_part_doubles(), mid_init ? x : nullptr));
        // (the final MLP reads the side stream's bounds and operand images: behind the layer loop, where the wait costs nothing)
        if (sw != s && weights_done) TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[kEvWeights], 0));
        if (tail_k) {
            TGNN_TRY(launch_forward_tail(dims, P, w.mid, w.small_pack, probs, w.partf, w.small_part_wide, slot_max, dense_max, n, tail_k,
                                         tail_blocks, update_running, eps, momentum, s, graph->nn_mid_verdict));
            return TGNN_OK;
        }
    } else if (s2 && !weights_on_main) {
        TGNN_CHECK_HIP(hipEventRecord(ev[0], s));            // middle[0] is complete
        TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[0], 0));
        if (sw != s) TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[kEvWeights], 0));
    }
    for (int i = 0; i < (mid_k ? 0 : D); ++i) {
        const int b = P.layer(i);
        if (keep) {                                          // this layer's own buffers (kernels already queued keep theirs)
            w.a1 = keep->a1 + (size_t)i * n * c;
            w.a2[i & 1] = keep->a2 + (size_t)i * n * c;
            w.t0 = keep->u + (size_t)i * n * c;
            w.stat1 = keep->stat1 + (size_t)i * 4 * c;
            w.stat2[i & 1] = keep->stat2 + (size_t)i * 4 * c;
        }
        const float *h1 = w.mid + (size_t)i * nr * c;
        // [r6] tgnn_forward_resume, layer 0: the NNConv's launch goes out BEFORE the collision chain's -- the host is what the first
        // layer waits for behind a just-prepared layout, and the adjacency chain is the longer one
        const bool nn_first = weights_on_main && i == 0;
        if (s2 && !nn_first) {
            // ---- collision chain, layer i, on the side stream: a2[i & 1] / stat2[i & 1] were last read by merge_{i-2}
            // (sharded: the halo rows and the statistics GIN_i reads arrive with the exchange of layer i-1, so the chain cannot run
            //  ahead; only the HBM-bound neighbourhood sum goes beside the merge / NNConv -- the MLP, which finds no CU beside an
            //  NNConv block, follows on the main stream: 52 us beside the NNConv against 19 us behind it, measured)
            if (sh && split) {
                // split exchange: the whole collision branch of layer i -- GIN, then ITS OWN all-to-all (halo rows of a2 + the
                // BatchNorm sums, 32 floats per row) and the statistics -- on the side stream: it needs nothing of the adjacency
                // branch, so the chain runs ahead of the NNConv / merge chain as it does on a single device, held back only by
                // the two-deep buffers (a2[i & 1] / stat2[i & 1] were last read by merge_{i-2})
                if (i >= 2) TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[1 + kMaxDepth + i - 2], 0));
                TGNN_TRY(gin_layer(i, s2));
                if (i + 1 < D) {
                    BnJob j2 = bn_job(w.part2, np2, P.bn(b + 20), w.stat2[i & 1]);
                    j2.sums = sh->sum_buf + 64;
                    const int64_t n_out = sh->n_send + 4 * (int64_t)sh->world, n_in = n_halo + 4 * (int64_t)sh->world;
                    float *sb2 = sh->send_buf + (size_t)n_out * c, *rb2 = sh->recv_buf + (size_t)n_in * c;
                    launch_shard_pack1(w.a2[i & 1], sh->send_idx_fused, n_out, j2, sb2, s2);
                    TGNN_TRY(alltoall(sb2, rb2, c, 4, s2));
                    launch_shard_unpack1(rb2, sh->recv_idx_fused, n_in, n, w.a2[i & 1], j2, sh->world, sh->rank, n_total, eps,
                                         momentum, s2);
                }
            } else if (sh) {
                if (i >= 1) TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[1 + kMaxDepth + i - 1], 0));
                const float *gin_in = i == 0 ? w.mid : w.a2[(i - 1) & 1];
                TGNN_TRY(tgnn_gin_aggregate(gin_in, c, i == 0 ? nullptr : w.stat2[(i - 1) & 1], graph->col_rowptr, graph->col_src,
                                            P.f(b + 13), n, c, w.t0, s2));
            } else {
                if (i >= 2) TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[1 + kMaxDepth + i - 2], 0));
                TGNN_TRY(gin_layer(i, s2));
                if (!fold_fin2) {
                    BnJobs j2{};
                    j2.job[0] = bn_job(w.part2, np2, P.bn(b + 20), w.stat2[i & 1]);
                    launch_bn_finalize(j2, 1, fin_mode, c, n, eps, momentum, s2);
                }
            }
            TGNN_CHECK_HIP(hipEventRecord(ev[1 + i], s2));
        }
        // GraphConv (:62): NNConv mean + LeakyReLU; BN statistics emitted as partials
        prof.begin(2);
        // [r6] sharded, split exchange: the pack of the adjacency branch's message inside the NNConv (nnconv_eg.hip: SHARD)
        const bool pack_in_nnconv = eg && split && i + 1 < D && sh->send_row_ptr && sh->send_row_slot && lean_head;
        if (eg) {
            EgShardPack pk{};
            if (pack_in_nnconv)
                pk = EgShardPack{sh->send_row_ptr, sh->send_row_slot, sh->send_buf, sh->send_idx_fused, sh->n_send + 4 * (int64_t)sh->world,
                                 sh->sum_buf, w.bounds + 2 * D + 27, w.small_part_wide + (size_t)4 * 16 * 512};
            TGNN_TRY(launch_nnconv_eg(h1, graph->nn_tile_grp_ptr, graph->nn_grp, w.wimg + (size_t)i * (T + 1) * kWtTypeF16, T,
                                      P.f(b + 7), n, TGNN_ACT_LEAKY_RELU, w.a1, w.part1, &np1, s, slot_max + i, root_max + i,
                                      prof.stamps ? prof.stamps + 2 * i : nullptr, pack_in_nnconv ? &pk : nullptr));
        } else if (tiled) {
            TGNN_TRY(launch_nnconv_cols(h1, c, graph->nn_tile_col_ptr, graph->nn_col_meta, graph->nn_col_src,
                                        w.wimg + (size_t)i * (T + 1) * (f16 ? kWtTypeF16 : kWtType), T, P.f(b + 7), n,
                                        TGNN_ACT_LEAKY_RELU, w.a1, w.part1, &np1, s, f16 ? slot_max + i : nullptr,
                                        f16 ? root_max + i : nullptr, graph->nn_max_in_degree,
                                        prof.stamps ? prof.stamps + 2 * i : nullptr));
        } else {
            TGNN_TRY(tgnn_nnconv_mean_fwd(h1, c, graph->adj_rowptr, graph->adj_src, graph->adj_type,
                                      w.wtab + (size_t)i * T * c * c, T, P.f(b + 6), P.f(b + 7), n, c,
                                      TGNN_ACT_LEAKY_RELU, w.a1, w.part1, &np1, s));
        }
        prof.end();
        if (nn_first) {                                      // (single device, i == 0: see the branch above)
            TGNN_TRY(gin_layer(i, s2));
            if (!fold_fin2) {
                BnJobs j2{};
                j2.job[0] = bn_job(w.part2, np2, P.bn(b + 20), w.stat2[i & 1]);
                launch_bn_finalize(j2, 1, fin_mode, c, n, eps, momentum, s2);
            }
            TGNN_CHECK_HIP(hipEventRecord(ev[1 + i], s2));
        }
        if (split && i + 1 < D) {
            // ---- sharded, split exchange: this chain carries the adjacency branch only (rows of a1 + its BatchNorm sums); the
            //      collision branch's half arrived (or is arriving) on the side stream
            BnJob j1 = bn_job(w.part1, np1, P.bn(b + 8), w.stat1);
            j1.sums = sh->sum_buf;
            const int64_t n_out = sh->n_send + 4 * (int64_t)sh->world, n_in = n_halo + 4 * (int64_t)sh->world;
            if (!pack_in_nnconv) launch_shard_pack1(w.a1, sh->send_idx_fused, n_out, j1, sh->send_buf, s);
            TGNN_TRY(alltoall(sh->send_buf, sh->recv_buf, c, 4, s));
            const float *resid_f = i >= 2 ? w.mid + (size_t)(i - 2) * nr * c : nullptr;
            TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[1 + i], 0));
            // [r6] unpack + merge as one launch: the record from the shards' sums in every block, the halo rows of a1 out of the message
            launch_shard_unpack1_merge(sh->recv_buf, sh->recv_idx_fused, n_in, n, w.a1, j1, sh->world, sh->rank, n_total, eps, momentum,
                                       w.a2[i & 1], w.stat2[i & 1], resid_f, nr, w.mid + (size_t)(i + 1) * nr * c,
                                       f16 ? slot_max + i + 1 : nullptr, s);
            TGNN_CHECK_HIP(hipEventRecord(ev[1 + kMaxDepth + i], s));       // (a2[i & 1] / stat2[i & 1] are free for GIN_{i+2})
            continue;
        }
        if (fused_shard && i + 1 < D) {
            // ---- sharded, fused: local sums -> ONE all-to-all (raw halo rows of both branches + the sums) -> the sums of
            //      all shards added in rank order -> statistics -> merge of the own AND the halo rows
            if (s2) {
                TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[1 + i], 0));
                TGNN_TRY(launch_gin32_mlp(w.t0, P.f(b + 14), P.f(b + 15), P.f(b + 16), P.f(b + 17), P.f(b + 18), P.f(b + 19), n,
                                          TGNN_ACT_LEAKY_RELU, w.a2[i & 1], w.part2, &np2, s, nullptr, true));
            } else {
                TGNN_TRY(gin_layer(i, s));
            }
            BnJobs jobs{};
            jobs.job[0] = bn_job(w.part1, np1, P.bn(b + 8), w.stat1);
            jobs.job[1] = bn_job(w.part2, np2, P.bn(b + 20), w.stat2[i & 1]);
            double *own = sh->sum_buf;
            jobs.job[0].sums = own;
            jobs.job[1].sums = own + 64;
            const int64_t n_out = sh->n_send + 4 * (int64_t)sh->world, n_in = n_halo + 4 * (int64_t)sh->world;
            launch_shard_pack_sums(w.a1, w.a2[i & 1], sh->send_idx_fused, n_out, jobs, sh->send_buf, s);
            TGNN_TRY(alltoall(sh->send_buf, sh->recv_buf, 2 * c, 4, s));
            launch_shard_unpack_finalize(sh->recv_buf, sh->recv_idx_fused, n_in, n, w.a1, w.a2[i & 1], jobs, sh->world,
                                         sh->rank, n_total, eps, momentum, s);
            // (what the next GIN reads -- the collision rows incl. halo and their statistics -- is complete here: it starts beside
            //  the merge, not behind it)
            if (s2) TGNN_CHECK_HIP(hipEventRecord(ev[1 + kMaxDepth + i], s));
            const float *resid_f = i >= 2 ? w.mid + (size_t)(i - 2) * nr * c : nullptr;
            if (f16)
                launch_merge(w.a1, w.stat1, w.a2[i & 1], w.stat2[i & 1], resid_f, nr, c, w.mid + (size_t)(i + 1) * nr * c, nullptr,
                             slot_max + i + 1, s);
            else
                TGNN_TRY(tgnn_merge_fwd(w.a1, w.stat1, w.a2[i & 1], w.stat2[i & 1], resid_f, nr, c,
                                        w.mid + (size_t)(i + 1) * nr * c, nullptr, s));
            continue;
        }
        // Few partial rows (small layouts): merge derives the first BatchNorm's record from them itself -- one launch
        // less on the critical chain of a launch-latency-bound forward.  (With the 256 rows of a 100k-node layout the
        // repeated reduction costs every merge block more than the separate 1-block finalize: measured.)
        // (round 2: with 1024-thread blocks -- one batch of independent loads per thread -- the repeated reduction pays at every
        //  size: the launch it replaces sits on the critical NNConv -> merge chain.  TGNN_BN_MAX_PARTIALS rows at most.)
        constexpr int fuse_rows = TGNN_BN_MAX_PARTIALS;
        const bool fused_bn1 = c == 32 && !use_running_stats && !sh && np1 <= fuse_rows;
        if (s2 && sh) {
            TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[1 + i], 0));
            if (!split)
                TGNN_TRY(launch_gin32_mlp(w.t0, P.f(b + 14), P.f(b + 15), P.f(b + 16), P.f(b + 17), P.f(b + 18), P.f(b + 19), n,
                                          TGNN_ACT_LEAKY_RELU, w.a2[i & 1], w.part2, &np2, s, nullptr, true));
            BnJobs jobs{};
            jobs.job[0] = bn_job(w.part1, np1, P.bn(b + 8), w.stat1);
            jobs.job[1] = bn_job(w.part2, np2, P.bn(b + 20), w.stat2[i & 1]);
            TGNN_TRY(finalize_jobs(jobs, 2, c));
        } else if (s2) {
            if (!fused_bn1) {
                BnJobs j1{};
                j1.job[0] = bn_job(w.part1, np1, P.bn(b + 8), w.stat1);
                launch_bn_finalize(j1, 1, fin_mode, c, n, eps, momentum, s);
            }
            TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[1 + i], 0));
        } else {
            TGNN_TRY(gin_layer(i, s));
            BnJobs jobs{};
            int nj = 0;
            if (!fused_bn1) jobs.job[nj++] = bn_job(w.part1, np1, P.bn(b + 8), w.stat1);
            if (!fold_fin2) jobs.job[nj++] = bn_job(w.part2, np2, P.bn(b + 20), w.stat2[i & 1]);
            TGNN_TRY(finalize_jobs(jobs, nj, c));
        }
        // merge (:64-71): middle[i+1] = BN1(a1) * BN2(a2) (+ middle[i-2])
        const float *resid = i >= 2 ? w.mid + (size_t)(i - 2) * nr * c : nullptr;
        prof.begin(5);
        if (fused_bn1) {
            launch_merge_bn1(w.a1, bn_job(w.part1, np1, P.bn(b + 8), w.stat1), n, eps, momentum, w.a2[i & 1],
                             w.stat2[i & 1], resid, n, w.mid + (size_t)(i + 1) * nr * c, s, f16 ? slot_max + i + 1 : nullptr);
        } else if (f16) {
            launch_merge(w.a1, w.stat1, w.a2[i & 1], w.stat2[i & 1], resid, n, c, w.mid + (size_t)(i + 1) * nr * c, nullptr,
                         slot_max + i + 1, s);
        } else {
            TGNN_TRY(tgnn_merge_fwd(w.a1, w.stat1, w.a2[i & 1], w.stat2[i & 1], resid, n, c,
                                    w.mid + (size_t)(i + 1) * nr * c, nullptr, s));
        }
        prof.end();
        if (s2 && !sh) TGNN_CHECK_HIP(hipEventRecord(ev[1 + kMaxDepth + i], s));
        if (i + 1 < D) TGNN_TRY(exchange(i + 1, w.a2[i & 1], w.a2[i & 1]));
        if (s2 && sh) TGNN_CHECK_HIP(hipEventRecord(ev[1 + kMaxDepth + i], s));     // (the next GIN also reads the halo rows)
    }

    // ---- K11: final MLP over the concatenation (TilinGNN.py:74-76); K block kb = middle[kb]
    const int cat_dim = c * (D + 1);
    float *fbuf[4] = {w.f1, w.f2, w.f3, w.f4};
    int fdim[5] = {cat_dim, kFinalDims[0], kFinalDims[1], kFinalDims[2], c};
    // [r6] the final MLP's BatchNorm records by their producers (bn_fold_two_level in the rows / resident kernels: the same bits) instead
    // of a 7 us bn_finalize launch behind each; 17 counter words per layer behind the collision branch's, cleared by the scales kernel
    // (bit 2 of tgnn_set_lean_head, OFF by default: measured, every producer grew by the 6 - 8 us its finalize launch took -- all of
    //  a dense kernel's blocks finish together, so both levels of the fold are serial latency behind the last one, unlike in the
    //  collision MLP, whose row groups finish at different times: profiles/r06_tail_fold.txt)
    const bool fold_final = lean_head && !sh && !use_running_stats && c == 32 && (g_lean_head.load(std::memory_order_relaxed) & 4);
    for (int l = 0; l < 4; ++l) {
        const int pi = P.fin(l);
        GinFin ff{};
        bool folded = false;
        if (fold_final) {
            ff.counter = w.bounds + 2 * D + 27 + 17 * l;
            ff.job = bn_job(nullptr, 0, P.bn(pi + 2), w.stat_f[l]);
            ff.n_total = n;
            ff.eps = eps;
            ff.momentum = momentum;
        }
        double *fold_rows = w.small_part_wide + (size_t)l * 16 * 512;
        if (l == 0) {
            TGNN_CHECK_ARG(c % 32 == 0, "final MLP over the slot-major buffer needs a network_width that is a multiple of 32");
            prof.begin(6);
            if (f16)
                TGNN_TRY(dense_act_slots_bounded(w.mid, c, (int64_t)nr * c, P.f(pi), P.f(pi + 1), n, cat_dim, fdim[1],
                                                 TGNN_ACT_LEAKY_RELU, fbuf[0], fdim[1], w.partf, &np1, slot_max, D + 1, dense_max, s,
                                                 dimg_ok[0] ? w.dimg[0] : nullptr, fold_final ? &ff : nullptr, fold_rows, &folded));
            else
                TGNN_TRY(tgnn_dense_act_slots_fwd(w.mid, c, (int64_t)nr * c, nullptr, P.f(pi), P.f(pi + 1), n, cat_dim, fdim[1],
                                                  TGNN_ACT_LEAKY_RELU, fbuf[0], fdim[1], w.partf, &np1, s));
            prof.end();
        } else {
            prof.begin(6);
            if (f16 && (l <= 2 || dimg_ok[l]))   // fp16 pairs: the input's bound follows from the producer's BatchNorm parameters (dense_bounds_kernel)
                TGNN_TRY(dense_act_bounded(fbuf[l - 1], fdim[l], 32, w.stat_f[l - 1], P.f(pi), P.f(pi + 1), n, fdim[l], fdim[l + 1],
                                           TGNN_ACT_LEAKY_RELU, fbuf[l], fdim[l + 1], w.partf, &np1, final_bound_word(w, D, l) + 1, 1,
                                           final_bound_word(w, D, l), s, dimg_ok[l] ? w.dimg[l] : nullptr, fold_final ? &ff : nullptr, fold_rows,
                                           &folded));
            else
                TGNN_TRY(tgnn_dense_act_fwd(fbuf[l - 1], fdim[l], 32, w.stat_f[l - 1], P.f(pi), P.f(pi + 1), n, fdim[l],
                                            fdim[l + 1], TGNN_ACT_LEAKY_RELU, fbuf[l], fdim[l + 1], w.partf, &np1, s));
            prof.end();
        }
        if (!folded) TGNN_TRY(finalize1(w.partf, np1, fdim[l + 1], P.bn(pi + 2), w.stat_f[l]));
    }
    prof.begin(6);
    TGNN_TRY(tgnn_dense_act_fwd(fbuf[3], c, 32, w.stat_f[3], P.f(P.last()), P.f(P.last() + 1), n, c, dims->output_dim,
                                TGNN_ACT_SIGMOID, probs, dims->output_dim, nullptr, nullptr, s));
    prof.end();
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_forward(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                            const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                            int32_t use_running_stats, float *probs, void *ws, size_t ws_bytes, tgnn_stream_t stream,
                            tgnn_stream_t stream2) {
    DeviceGuard guard__(stream);
    Prof prof;
    return forward_impl(dims, params_host, x, adj_edge_attr, graph, update_running, use_running_stats, probs, ws,
                        ws_bytes, stream, stream2, prof);
}

extern "C" int tgnn_forward_begin(const tgnn_model_dims *dims, const void *const *params_host, const float *x, int64_t n_nodes,
                                  int32_t update_running, void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2) {
    DeviceGuard guard__(stream ? stream : stream2);
    TGNN_CHECK_ARG(dims_ok(dims) && params_host && x && n_nodes >= 2, "arguments");
    hipStream_t s = static_cast<hipStream_t>(stream), s2 = static_cast<hipStream_t>(stream2);
    const int c = dims->network_width, D = dims->network_depth, fx = dims->node_features_dim;
    // what forward_impl's lean head + fused init MLP need, as far as it can be known without the graph; the rest (fp16-pair
    // operands: edge groups / columns, largest in-degree) is checked by tgnn_forward_resume
    if (!s2 || s2 == s || c != 32 || fx > 8 || D > kMaxDepth || ((int64_t)c * (D + 1) * kFinalDims[0]) % 4 != 0 || !g_split_f16 ||
        (g_lean_head.load(std::memory_order_relaxed) & 3) != 3 || n_nodes <= tgnn_get_mid_layout_limit() || n_nodes <= tgnn_get_small_layout_limit())
        return TGNN_ERR_UNSUPPORTED;
    const int np = tgnn_param_count(dims);
    for (int i = 0; i < np; ++i)
        if (!params_host[i]) {
            set_error("tgnn_forward_begin: params_host[%d] is null", i);
            return TGNN_ERR_INVALID_ARG;
        }
    Workspace w = carve(*dims, n_nodes, n_nodes, 0, ws, ws_bytes);
    if (!ws || w.bytes > ws_bytes) {
        set_error("tgnn_forward_begin: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return TGNN_ERR_WORKSPACE;
    }
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
    HeadEvent &he = g_head[dev];
    static thread_local hipEvent_t fork_ev[64] = {};
    if (!he.ev) TGNN_CHECK_HIP(hipEventCreateWithFlags(&he.ev, hipEventDisableTiming));
    if (!fork_ev[dev]) TGNN_CHECK_HIP(hipEventCreateWithFlags(&fork_ev[dev], hipEventDisableTiming));
    if (s) {                                                 // (stream == NULL: the caller has ordered stream2 behind x and the parameters itself)
        TGNN_CHECK_HIP(hipEventRecord(fork_ev[dev], s));       // x and the parameters are the caller's, ordered on `stream`
        TGNN_CHECK_HIP(hipStreamWaitEvent(s2, fork_ev[dev], 0));
    }
    const Params P{params_host, D};
    const float *roots[kMaxDepth];
    for (int i = 0; i < D; ++i) roots[i] = P.f(P.layer(i) + 6);
    launch_forward_scales(w.bounds, 2 * D + 95, roots, D, w.bounds + D + 1, nullptr, 0, nullptr, s2);
    TGNN_TRY(forward_head_init(dims, P, x, w, n_nodes, update_running, w.bounds, s2));
    bool ok = false;
    TGNN_TRY(forward_head_bounds_images(dims, P, w, n_nodes, n_nodes, s2, &ok));
    TGNN_CHECK_HIP(hipEventRecord(he.ev, s2));
    he.ws = ws;
    he.n = n_nodes;
    he.weights = false;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// [r6] Between tgnn_forward_begin and tgnn_forward_resume, BEHIND the preparation's launches on `stream` and before the host has the
// type count: the edge weights and the edge-group NNConv's operand images of all layers, with the count read on the device
// (result word 0 of tgnn_graph_prep) -- the launch tgnn_forward_resume would queue first, ~60 us of host round trip earlier.
// [r6] Small layouts (the one persistent kernel, forward_small.hip): what their forward queues in front of that kernel -- the edge
// weights with the column kernel's bf16 x 3 operand images and the parameter pack -- BEHIND the one-launch preparation on `stream`
// and before the host has read its result words: the type count is read on the device (`n_types_dev` = the preparation's result
// word 0).  The next tgnn_forward of this thread with the same workspace and node count picks it up if it takes the small-layout
// path; any other forward queues its own.
extern "C" int tgnn_forward_small_prepass(const tgnn_model_dims *dims, const void *const *params_host, const float *adj_edge_attr,
                                          const int32_t *type_rep_edge, const int32_t *n_types_dev, int64_t n_nodes, void *ws,
                                          size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(dims_ok(dims) && params_host && n_types_dev, "arguments");
    if (!adj_edge_attr || !type_rep_edge) return TGNN_ERR_UNSUPPORTED;   // (a layout without adjacency edges: nothing to queue early)
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
    const int c = dims->network_width, D = dims->network_depth, fe = dims->adj_edge_features_dim;
    if (c != 32 || n_nodes < 2 || n_nodes > tgnn_get_small_layout_limit() || !edge_weight_table_device_count_ok(fe, c) ||
        (g_lean_head.load(std::memory_order_relaxed) & 8))
        return TGNN_ERR_UNSUPPORTED;
    Workspace w = carve(*dims, n_nodes, n_nodes, 0, ws, ws_bytes);
    if (!ws || w.bytes > ws_bytes) return TGNN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Params P{params_host, D};
    EdgeMlpLayers layers{};
    const float *roots[kMaxDepth];
    for (int i = 0; i < D; ++i) {
        const int b = P.layer(i);
        layers.l[i] = EdgeMlpLayer{P.f(b), P.f(b + 1), P.f(b + 2), P.f(b + 3), P.f(b + 4), P.f(b + 5)};
        roots[i] = P.f(b + 6);
    }
    launch_edge_weight_table_batched(adj_edge_attr, type_rep_edge, 0, fe, layers, D, c, w.wtab, roots, w.wimg, s, nullptr, nullptr, 1.0f,
                                     n_types_dev, kCarveTypes);
    launch_small_pack(P, D, w.small_pack, w.small_ctr, s);
    g_small_pre[dev].ws = ws;
    g_small_pre[dev].n = n_nodes;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_forward_begin_weights(const tgnn_model_dims *dims, const void *const *params_host, const float *adj_edge_attr,
                                          const int32_t *type_rep_edge, const int32_t *n_types_dev, int64_t n_nodes, void *ws,
                                          size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(dims_ok(dims) && params_host && n_types_dev, "arguments");
    if (!adj_edge_attr || !type_rep_edge) return TGNN_ERR_UNSUPPORTED;   // (a layout without adjacency edges: nothing to queue early)
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
    HeadEvent &he = g_head[dev];
    const int c = dims->network_width, D = dims->network_depth, fe = dims->adj_edge_features_dim;
    if (!he.ev || he.ws != ws || he.n != n_nodes || c != 32 || !edge_weight_table_device_count_ok(fe, c) || !g_nnconv_eg.load() ||
        !g_split_f16.load() || (g_lean_head.load(std::memory_order_relaxed) & 24))
        return TGNN_ERR_UNSUPPORTED;                           // (no matching tgnn_forward_begin, or a forward that will not take this launch)
    Workspace w = carve(*dims, n_nodes, n_nodes, 0, ws, ws_bytes);
    if (!ws || w.bytes > ws_bytes) return TGNN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Params P{params_host, D};
    EdgeMlpLayers layers{};
    const float *roots[kMaxDepth];
    for (int i = 0; i < D; ++i) {
        const int b = P.layer(i);
        layers.l[i] = EdgeMlpLayer{P.f(b), P.f(b + 1), P.f(b + 2), P.f(b + 3), P.f(b + 4), P.f(b + 5)};
        roots[i] = P.f(b + 6);
    }
    // (no wait for tgnn_forward_begin's stream: the kernel takes the roots' bounds itself when it reads the type count itself)
    launch_edge_weight_table_batched(adj_edge_attr, type_rep_edge, 0, fe, layers, D, c, w.wtab, roots, w.wimg, s, nullptr, w.bounds + D + 1,
                                     kEgImageScale, n_types_dev, kCarveTypes);
    he.weights = true;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_forward_resume(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                   const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running, float *probs, void *ws,
                                   size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2) {
    DeviceGuard guard__(stream);
    Prof prof;
    return forward_impl(dims, params_host, x, adj_edge_attr, graph, update_running, 0, probs, ws, ws_bytes, stream, stream2, prof,
                        nullptr, nullptr, true);
}

extern "C" int tgnn_forward_many(const tgnn_model_dims *dims, const void *const *params_host, int32_t n_layouts,
                                 const float *const *x, const float *const *adj_edge_attr, const tgnn_graph *graphs,
                                 int32_t update_running, int32_t use_running_stats, float *const *probs, void *const *ws,
                                 const size_t *ws_bytes, const tgnn_stream_t *streams, int32_t n_streams, tgnn_stream_t stream2) {
    TGNN_CHECK_ARG(n_layouts >= 0 && n_streams >= 1 && x && adj_edge_attr && graphs && probs && ws && ws_bytes && streams,
                   "arguments");
    for (int k = 0; k < n_layouts; ++k) {
        tgnn_stream_t st = streams[k % n_streams];
        DeviceGuard guard__(st);
        Prof prof;
        const int rc = forward_impl(dims, params_host, x[k], adj_edge_attr[k], graphs + k, update_running, use_running_stats,
                                    probs[k], ws[k], ws_bytes[k], st, stream2, prof);
        if (rc != TGNN_OK) return rc;
    }
    return TGNN_OK;
}

extern "C" int tgnn_forward_train(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                  const float *adj_edge_attr, const tgnn_graph *graph, const tgnn_train_save *keep,
                                  float *probs, void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(keep, "null keep struct");
    Prof prof;
    return forward_impl(dims, params_host, x, adj_edge_attr, graph, 1, 0, probs, ws, ws_bytes, stream, stream2, prof, nullptr,
                        keep);
}

extern "C" int tgnn_forward_sharded(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                    const float *adj_edge_attr, const tgnn_graph *graph, const tgnn_shard *shard,
                                    int32_t update_running, float *probs, void *ws, size_t ws_bytes,
                                    tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(shard, "null shard");
    Prof prof;
    return forward_impl(dims, params_host, x, adj_edge_attr, graph, update_running, 0, probs, ws, ws_bytes, stream,
                        nullptr, prof, shard);
}

extern "C" int tgnn_forward_profiled(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                     const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                                     int32_t use_running_stats, float *probs, void *ws, size_t ws_bytes,
                                     tgnn_stream_t stream, float *class_ms_host, int32_t *class_launches_host) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(class_ms_host && class_launches_host, "null profile arrays");
    Prof prof;
    prof.on = true;
    const int rc = forward_impl(dims, params_host, x, adj_edge_attr, graph, update_running, use_running_stats, probs,
                                ws, ws_bytes, stream, nullptr, prof);
    prof.collect(class_ms_host, class_launches_host);
    return rc;
}

extern "C" int tgnn_forward_stamped(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                    const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running, float *probs,
                                    void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2,
                                    float *nnconv_us_host) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(nnconv_us_host && dims_ok(dims), "arguments");
    const int D = dims->network_depth;
    std::vector<unsigned long long> host(2 * (size_t)D);
    for (int i = 0; i < D; ++i) { host[2 * i] = ~0ull; host[2 * i + 1] = 0ull; }
    unsigned long long *dev_stamps = nullptr;
    TGNN_CHECK_HIP(hipMalloc(&dev_stamps, host.size() * sizeof(unsigned long long)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = (int)hipMemcpyAsync(dev_stamps, host.data(), host.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s);
    if (rc == 0) {
        Prof prof;
        prof.stamps = dev_stamps;
        rc = forward_impl(dims, params_host, x, adj_edge_attr, graph, update_running, 0, probs, ws, ws_bytes, stream, stream2, prof);
    }
    (void)hipStreamSynchronize(s);
    if (stream2) (void)hipStreamSynchronize(static_cast<hipStream_t>(stream2));
    if (rc == 0) rc = (int)hipMemcpy(host.data(), dev_stamps, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(dev_stamps);
    if (rc != 0) return rc < 0 ? rc : TGNN_ERR_LAUNCH;
    int dev = 0, khz = 100000;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    for (int i = 0; i < D; ++i)
        nnconv_us_host[i] = host[2 * i + 1] > host[2 * i] ? (float)((double)(host[2 * i + 1] - host[2 * i]) * 1e3 / (double)khz) : 0.f;
    return TGNN_OK;
}

extern "C" int tgnn_forward_profiled_two_stream(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                                const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running,
                                                float *probs, void *ws, size_t ws_bytes, tgnn_stream_t stream,
                                                tgnn_stream_t stream2, float *class_ms_host, int32_t *class_launches_host) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(class_ms_host && class_launches_host, "null profile arrays");
    Prof prof;
    prof.on = true;
    prof.two_stream = true;
    prof.class_mask = (1u << TGNN_PROF_NNCONV) | (1u << TGNN_PROF_MERGE);   // the adjacency chain: launched on `stream`
    const int rc = forward_impl(dims, params_host, x, adj_edge_attr, graph, update_running, 0, probs, ws, ws_bytes, stream,
                                stream2, prof);
    prof.collect(class_ms_host, class_launches_host);
    return rc;
}
