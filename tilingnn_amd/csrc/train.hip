// The backward of the training step as ONE library call.
//
// Reference: Trainer.train, /root/reference/solver/ml_solver/trainer.py:76-80 -- loss.backward() through the network
// TilinGNN.forward built (graph_networks/networks/TilinGNN.py:51-78).  tilingnn_amd/train.py:backward_train is the
// readable, op-by-op statement of this schedule (and stays the checker: tests compare the two bit for bit); here the same
// adjoint kernels (backward.hip) are enqueued from C++ so that a step on the 1-3 k-node layouts the reference trains on is
// not bound by ~600 Python-level calls.
//
// Walk, given d loss / d probs:
//   final Linear_trans (sigmoid, no BN)  ->  final MLP (4 x Linear/LeakyReLU/BatchNorm)  ->  d cat [N, (D+1) C]
//   for i = D-1 .. 0:  merge backward (+ both BatchNorm reductions)  ->  BatchNorm/LeakyReLU apply, twice
//                      ->  NNConv adjoint (one type-sum pass over the transposed graph; dense products; edge MLP)
//                      ->  GIN adjoint (sigmoid MLP, aggregation over the transposed collision graph)
//   init MLP (2 x Linear/LeakyReLU/BatchNorm)
// Gradients go to grads_host[k], indexed like params_host (entries of buffers -- running statistics, GIN's eps -- are
// not touched).
#include "tgnn_common.h"

namespace tgnn {

struct Arena {
    char *base;
    size_t off = 0, cap;
    Arena(void *p, size_t bytes) : base(static_cast<char *>(p)), cap(bytes) {}
    float *f(size_t count) {
        off = align_up(off, 256);
        float *p = reinterpret_cast<float *>(base + off);
        off += count * sizeof(float);
        return p;
    }
    void *bytes(size_t n) {
        off = align_up(off, 256);
        void *p = base + off;
        off += n;
        return p;
    }
};

static const int kFinDims[3] = {256, 128, 64};             // TilinGNN.py:46 hidden_layer_dims

struct BwdBuffers {
    float *dcat, *buf[3], *dy1, *dy2, *dz1, *gsc, *dz2, *sbwd, *wd, *dh1, *dw_in, *dwcat, *rows, *carry[2], *coef, *wt, *zero;
    void *red, *wg, *mlp;
    size_t red_bytes, wg_bytes, mlp_bytes, total;
};

static BwdBuffers carve_bwd(const tgnn_model_dims &d, int64_t n, int32_t T, void *ws, size_t ws_bytes) {
    Arena a(ws, ws_bytes);
    const int c = d.network_width, D = d.network_depth, fe = d.adj_edge_features_dim;
    const int cat = c * (D + 1);
    BwdBuffers b{};
    const int64_t nn = n > 0 ? n : 1;
    b.dcat = a.f((size_t)nn * cat);
    for (int k = 0; k < 3; ++k) b.buf[k] = a.f((size_t)nn * 256);
    b.dy1 = a.f((size_t)nn * c); b.dy2 = a.f((size_t)nn * c); b.dz1 = a.f((size_t)nn * c); b.gsc = a.f((size_t)nn * c);
    b.dz2 = a.f((size_t)nn * c); b.dh1 = a.f((size_t)nn * c);
    b.carry[0] = a.f((size_t)nn * c); b.carry[1] = a.f((size_t)nn * c);
    b.sbwd = a.f((size_t)nn * (T + 1) * c);
    b.wd = a.f((size_t)c * (T + 1) * c);
    b.dw_in = a.f((size_t)(T + 1) * c * c);
    b.dwcat = a.f((size_t)(T + 1) * c * c);
    b.rows = a.f((size_t)(T > 0 ? T : 1) * fe);
    b.coef = a.f(4 * 256);
    size_t wt = (size_t)cat * 256;                           // the largest transposed weight: final MLP layer 0 ...
    if ((size_t)256 * 128 > wt) wt = (size_t)256 * 128;      // ... or layer 1 when the network is shallow
    b.wt = a.f(wt);
    b.zero = a.f((size_t)(cat > 1024 ? cat : 1024));
    b.red_bytes = tgnn_reduce_workspace_bytes(256);
    b.red = a.bytes(b.red_bytes);
    size_t wg = tgnn_wgrad_workspace_bytes(nn, 256, cat);
    const size_t cand[] = {tgnn_wgrad_workspace_bytes(nn, 128, 256), tgnn_wgrad_workspace_bytes(nn, c, (T + 1) * c),
                           tgnn_wgrad_workspace_bytes(nn, c, d.node_features_dim), tgnn_wgrad_workspace_bytes(nn, c, c),
                           tgnn_wgrad_workspace_bytes(nn, 64, 128), tgnn_wgrad_workspace_bytes(nn, c, 64),
                           tgnn_wgrad_workspace_bytes(nn, d.output_dim, c)};
    for (size_t v : cand)
        if (v > wg) wg = v;
    b.wg_bytes = wg;
    b.wg = a.bytes(wg);
    size_t mlp = tgnn_sigmoid_mlp_bwd_workspace_bytes(nn, c, 32, 64, c);
    const size_t mlp_e = tgnn_sigmoid_mlp_bwd_workspace_bytes(T > 0 ? T : 1, fe, 32, 64, c * c);
    if (mlp_e > mlp) mlp = mlp_e;
    b.mlp_bytes = mlp;
    b.mlp = a.bytes(mlp);
    b.total = a.off + 256;
    return b;
}

#define TGNN_TRYB(expr)                  \
    do {                                 \
        const int rc__ = (expr);         \
        if (rc__ != TGNN_OK) return rc__; \
    } while (0)

}  // namespace tgnn

using namespace tgnn;

extern "C" size_t tgnn_backward_workspace_bytes(const tgnn_model_dims *dims, int64_t n_nodes, int32_t n_types) {
    if (!dims || n_nodes < 0 || n_types < 0) return 0;
    return carve_bwd(*dims, n_nodes, n_types, nullptr, 0).total;
}

extern "C" int tgnn_backward(const tgnn_model_dims *dims, const void *const *params_host, void *const *grads_host,
                             const float *x, const float *adj_edge_attr, const tgnn_graph *graph,
                             const tgnn_train_graph *tgraph, const tgnn_train_save *keep, const float *probs,
                             const float *dprobs, void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(dims && params_host && grads_host && x && graph && tgraph && keep && probs && dprobs, "null pointer");
    const int c = dims->network_width, D = dims->network_depth, fx = dims->node_features_dim,
              fe = dims->adj_edge_features_dim, od = dims->output_dim, T = graph->n_types;
    const int64_t n = graph->n_nodes;
    if (c != 32 || T > 63) {
        set_error("tgnn_backward: network_width 32 and at most 63 edge types");
        return TGNN_ERR_UNSUPPORTED;
    }
    TGNN_CHECK_ARG(n >= 2 && D >= 1 && D <= kMaxDepth, "shape");
    TGNN_CHECK_ARG(tgraph->adjT_rowptr && tgraph->colT_rowptr && tgraph->deg && tgraph->inv_deg, "transposed graph");
    BwdBuffers b = carve_bwd(*dims, n, T, ws, ws_bytes);
    if (!ws || b.total > ws_bytes) {
        set_error("tgnn_backward: workspace too small (%zu < %zu)", ws_bytes, b.total);
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Params P{params_host, D};
    auto G = [&](int i) { return static_cast<float *>(grads_host[i]); };
    const int np = 2 * kInitStride + D * kLayerStride + 4 * kFinalStride + 2;
    for (int i = 0; i < np; ++i) TGNN_CHECK_ARG(params_host[i], "params_host entry");
    const float eps = 1e-5f;
    const int cat = c * (D + 1);
    TGNN_CHECK_HIP(hipMemsetAsync(b.zero, 0, sizeof(float) * (cat > 1024 ? cat : 1024), s));

    // dz [n, out] . W [out, in] -> dx [n, in]: the forward dense kernel on W^T
    auto dense_dx = [&](const float *dz, const float *w, int out_dim, int in_dim, float *dx) -> int {
        TGNN_TRYB(tgnn_transpose(w, out_dim, in_dim, b.wt, stream));
        return tgnn_dense_act_fwd(dz, out_dim, 32, nullptr, b.wt, b.zero, n, out_dim, in_dim, TGNN_ACT_NONE, dx, in_dim, nullptr,
                                  nullptr, stream);
    };
    // BatchNorm (train) + LeakyReLU backward of a Linear_trans: dy -> dz; gamma / beta gradients
    auto bn_leaky_bwd = [&](const float *dy, int64_t ld_dy, const float *a, const float *stat, int f, int bn_idx,
                            float *dz) -> int {
        TGNN_TRYB(tgnn_bn_bwd_reduce(dy, ld_dy, a, f, stat, n, f, eps, b.coef, G(bn_idx), G(bn_idx + 1), b.red, b.red_bytes,
                                     stream));
        return tgnn_bn_bwd_apply(dy, ld_dy, a, f, stat, b.coef, n, f, TGNN_ACT_LEAKY_RELU, dz, f, nullptr, nullptr, 0, stream);
    };

    // ---- final Linear_trans (C -> out, Sigmoid, no BatchNorm)
    {
        float *dl = b.buf[0], *y = b.buf[1], *dy = b.buf[2];
        TGNN_TRYB(tgnn_sigmoid_bwd(dprobs, od, probs, od, n, od, dl, od, stream));
        TGNN_TRYB(tgnn_bn_apply(keep->fin_a[3], c, keep->fin_stat[3], n, c, y, c, stream));
        TGNN_TRYB(tgnn_wgrad(dl, od, y, c, 0, n, od, c, G(P.last()), G(P.last() + 1), b.wg, b.wg_bytes, stream));
        TGNN_TRYB(dense_dx(dl, P.f(P.last()), od, c, dy));
    }
    // ---- final MLP, layers 3 .. 0 (widths C <- 64 <- 128 <- 256 <- cat)
    {
        const int fdim[5] = {cat, kFinDims[0], kFinDims[1], kFinDims[2], c};
        float *dy = b.buf[2];                                 // gradient at the (normalised) output of layer l
        for (int l = 3; l >= 0; --l) {
            const int pi = P.fin(l), f = fdim[l + 1], fin = fdim[l];
            float *dz = b.buf[0];
            TGNN_TRYB(bn_leaky_bwd(dy, f, keep->fin_a[l], keep->fin_stat[l], f, pi + 2, dz));
            if (l > 0) {
                float *inp = b.buf[1];
                TGNN_TRYB(tgnn_bn_apply(keep->fin_a[l - 1], fin, keep->fin_stat[l - 1], n, fin, inp, fin, stream));
                TGNN_TRYB(tgnn_wgrad(dz, f, inp, fin, 0, n, f, fin, G(pi), G(pi + 1), b.wg, b.wg_bytes, stream));
                TGNN_TRYB(dense_dx(dz, P.f(pi), f, fin, dy));   // dy is free again: dz holds what it carried
            } else {
                TGNN_TRYB(tgnn_wgrad(dz, f, keep->skip, c, n * c, n, f, fin, G(pi), G(pi + 1), b.wg, b.wg_bytes, stream));
                TGNN_TRYB(dense_dx(dz, P.f(pi), f, fin, b.dcat));
            }
        }
    }

    // ---- the message-passing layers, last to first
    auto slot = [&](int k) { return b.dcat + (size_t)k * c; };           // row stride cat
    const float *carry = nullptr;
    for (int i = D - 1; i >= 0; --i) {
        const int pb = P.layer(i);
        const float *a1 = keep->a1 + (size_t)i * n * c, *a2 = keep->a2 + (size_t)i * n * c;
        const float *st1 = keep->stat1 + (size_t)i * 4 * c, *st2 = keep->stat2 + (size_t)i * 4 * c;
        float *coef1 = b.coef, *coef2 = b.coef + 2 * c;
        TGNN_TRYB(tgnn_merge_bwd_reduce(slot(i + 1), cat, a1, st1, a2, st2, carry, n, c, eps, eps, b.dy1, b.dy2,
                                        i >= 2 ? slot(i - 2) : nullptr, cat, coef1, G(pb + 8), G(pb + 9), coef2, G(pb + 20),
                                        G(pb + 21), b.red, b.red_bytes, stream));
        TGNN_TRYB(tgnn_bn_bwd_apply(b.dy1, c, a1, c, st1, coef1, n, c, TGNN_ACT_LEAKY_RELU, b.dz1, c, tgraph->inv_deg, b.gsc, c,
                                    stream));
        TGNN_TRYB(tgnn_bn_bwd_apply(b.dy2, c, a2, c, st2, coef2, n, c, TGNN_ACT_LEAKY_RELU, b.dz2, c, nullptr, nullptr, 0,
                                    stream));
        // NNConv adjoint
        const float *h = keep->skip + (size_t)i * n * c;
        const float *wtab = keep->wtab + (size_t)i * (T > 0 ? T : 1) * c * c;
        TGNN_TRYB(tgnn_nnconv_type_sum(b.gsc, c, b.gsc, c, tgraph->deg, tgraph->adjT_rowptr, tgraph->adjT_src, tgraph->adjT_type,
                                       n, T, c, b.sbwd, stream));
        TGNN_TRYB(tgnn_swap_leading(wtab, T, c, c, b.wd, T + 1, stream));
        TGNN_TRYB(tgnn_swap_leading(P.f(pb + 6), 1, c, c, b.wd + (size_t)T * c, T + 1, stream));
        TGNN_TRYB(tgnn_dense_act_fwd(b.sbwd, (int64_t)(T + 1) * c, 32, nullptr, b.wd, b.zero, n, (T + 1) * c, c, TGNN_ACT_NONE,
                                     b.dh1, c, nullptr, nullptr, stream));
        TGNN_TRYB(tgnn_wgrad(h, c, b.sbwd, (int64_t)(T + 1) * c, 0, n, c, (T + 1) * c, b.dw_in, nullptr, b.wg, b.wg_bytes,
                             stream));
        TGNN_TRYB(tgnn_swap_leading(b.dw_in, c, T + 1, c, b.dwcat, c, stream));
        TGNN_CHECK_HIP(hipMemcpyAsync(G(pb + 6), b.dwcat + (size_t)T * c * c, sizeof(float) * c * c, hipMemcpyDeviceToDevice, s));
        TGNN_TRYB(tgnn_colsum(b.dz1, c, n, c, G(pb + 7), b.red, b.red_bytes, stream));
        TGNN_TRYB(tgnn_add_into(b.dh1, c, n, c, slot(i), cat, stream));
        if (T > 0) {
            TGNN_TRYB(tgnn_rows_gather(adj_edge_attr, fe, graph->type_rep_edge, T, fe, b.rows, fe, stream));
            TGNN_TRYB(tgnn_sigmoid_mlp_bwd(b.rows, T, fe, 32, 64, c * c, P.f(pb), P.f(pb + 1), P.f(pb + 2), P.f(pb + 3),
                                           P.f(pb + 4), wtab, b.dwcat, (int64_t)c * c, G(pb), G(pb + 1), G(pb + 2), G(pb + 3),
                                           G(pb + 4), G(pb + 5), nullptr, b.mlp, b.mlp_bytes, stream));
        } else {
            const int sz[6] = {32 * fe, 32, 64 * 32, 64, c * c * 64, c * c};
            for (int k = 0; k < 6; ++k) TGNN_CHECK_HIP(hipMemsetAsync(G(pb + k), 0, sizeof(float) * sz[k], s));
        }
        // GIN adjoint: u = the aggregate the MLP read, a2 = its output (LeakyReLU is the identity on a sigmoid)
        float *du = b.dy1;                                   // dy1 is dead by now
        TGNN_TRYB(tgnn_sigmoid_mlp_bwd(keep->u + (size_t)i * n * c, n, c, 32, 64, c, P.f(pb + 14), P.f(pb + 15), P.f(pb + 16),
                                       P.f(pb + 17), P.f(pb + 18), a2, b.dz2, c, G(pb + 14), G(pb + 15), G(pb + 16), G(pb + 17),
                                       G(pb + 18), G(pb + 19), du, b.mlp, b.mlp_bytes, stream));
        float *next_carry = b.carry[i & 1];
        TGNN_TRYB(tgnn_gin_aggregate(du, c, nullptr, tgraph->colT_rowptr, tgraph->colT_src, P.f(pb + 13), n, c, next_carry, stream));
        carry = next_carry;
    }
    TGNN_TRYB(tgnn_add_into(carry, c, n, c, slot(0), cat, stream));     // h2 of layer 0 is the init output (TilinGNN.py:55)

    // ---- init MLP, layers 1, 0 (C <- C <- Fx); the gradient at x is not needed
    {
        float *dz = b.buf[0], *inp = b.buf[1], *dy = b.buf[2];
        const int p1 = P.init(1), p0 = P.init(0);
        TGNN_TRYB(bn_leaky_bwd(slot(0), cat, keep->init_a[1], keep->init_stat[1], c, p1 + 2, dz));
        TGNN_TRYB(tgnn_bn_apply(keep->init_a[0], c, keep->init_stat[0], n, c, inp, c, stream));
        TGNN_TRYB(tgnn_wgrad(dz, c, inp, c, 0, n, c, c, G(p1), G(p1 + 1), b.wg, b.wg_bytes, stream));
        TGNN_TRYB(dense_dx(dz, P.f(p1), c, c, dy));
        TGNN_TRYB(bn_leaky_bwd(dy, c, keep->init_a[0], keep->init_stat[0], c, p0 + 2, dz));
        TGNN_TRYB(tgnn_wgrad(dz, c, x, fx, 0, n, c, fx, G(p0), G(p0 + 1), b.wg, b.wg_bytes, stream));
    }
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
