// The unsupervised loss on the predict path.
//
// Reference: Losses.calculate_unsupervised_loss, /root/reference/solver/ml_solver/losses.py:48-116, called from
// ML_Solver.predict through get_best_prob_map (ml_solver.py:46,133-136).  Per probability map m:
//     l_area  = log( max( mean_v area[v] p[v], eps ) )                                   (:65-67)
//     l_feas  = mean_{collision edges (i,j)} log( 1 - clamp(p[i] p[j], eps, 1 - eps) )     (:69-81)   0 without such edges
//     l_align = mean_{adjacency edges (i,j)} log10( max(p[i] p[j] len_e, eps) )            (:83-98)   0 without such edges
//     loss    = (1 - Wa l_area) (1 - Wc l_feas) (1 - Wl l_align)                           (:104-106)
// The reference evaluates every element in fp32 and sums in fp32; here the elements are fp32 (same clamp / log) and
// the three sums run in fp64 over a fixed tree: per-block partials, then one block per map.  Two launches, no atomics.
#include "tgnn_common.h"

namespace tgnn {

constexpr float kLossEps = 1e-7f;          // losses.py:10
constexpr int kLossThreads = 256;

// grid = (blocks, maps); partial[(m * gridDim.x + b) * 3 + {0,1,2}]
__global__ __launch_bounds__(kLossThreads) void loss_partial_kernel(
    const float *__restrict__ probs, int64_t ldp, const float *__restrict__ area, int64_t lda, int64_t n,
    const int64_t *__restrict__ col, int64_t ec, const int64_t *__restrict__ adj, int64_t ea,
    const float *__restrict__ len, int64_t ldl, double *__restrict__ partial, int *__restrict__ err_flag) {
    const int m = blockIdx.y;
    const float *p = probs + m;
    double s_area = 0.0, s_feas = 0.0, s_align = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kLossThreads, t0 = (int64_t)blockIdx.x * kLossThreads + threadIdx.x;
    for (int64_t v = t0; v < n; v += stride) s_area += (double)(area[v * lda] * p[v * ldp]);
    bool bad = false;                                      // an edge end outside [0, N): torch.gather raises there
    for (int64_t e = t0; e < ec; e += stride) {
        const int64_t i = col[e], j = col[ec + e];
        if (i < 0 || i >= n || j < 0 || j >= n) { bad = true; continue; }
        float pp = p[i * ldp] * p[j * ldp];
        pp = fminf(fmaxf(pp, kLossEps), 1.0f - kLossEps);
        s_feas += (double)logf(1.0f - pp);
    }
    for (int64_t e = t0; e < ea; e += stride) {
        const int64_t i = adj[e], j = adj[ea + e];
        if (i < 0 || i >= n || j < 0 || j >= n) { bad = true; continue; }
        float pp = p[i * ldp] * p[j * ldp] * len[e * ldl];
        pp = fmaxf(pp, kLossEps);
        s_align += (double)(logf(pp) / 2.302585092994046f);
    }
    if (bad) *err_flag = 1;
    __shared__ double red[3][kLossThreads];
    red[0][threadIdx.x] = s_area; red[1][threadIdx.x] = s_feas; red[2][threadIdx.x] = s_align;
    __syncthreads();
    for (int d = kLossThreads / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[((int64_t)m * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// one wavefront per map: the partial rows are summed over a fixed tree (lane l takes rows l, l + 64, ..; then a
// butterfly), then the product of the three factors.  (A first version let three threads walk the <= 512 rows one
// dependent load after the other: 50 us.)
__global__ __launch_bounds__(64) void loss_final_kernel(const double *__restrict__ partial, int n_blocks, int64_t n,
                                                        int64_t ec, int64_t ea, float wc, float wl, float wa,
                                                        double *__restrict__ losses, double *__restrict__ terms,
                                                        const int *__restrict__ err_flag) {
    const int m = blockIdx.x, lane = threadIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = lane; b < n_blocks; b += 64)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += partial[((int64_t)m * n_blocks + b) * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[k] += __shfl_xor(s[k], d, 64);
    if (lane == 0) {
        const double t0 = log(fmax(s[0] / (double)n, (double)kLossEps));
        const double t1 = ec > 0 ? s[1] / (double)ec : 0.0;
        const double t2 = ea > 0 ? s[2] / (double)ea : 0.0;
        if (terms) { terms[m * 3] = t0; terms[m * 3 + 1] = t1; terms[m * 3 + 2] = t2; }
        const double l = (1.0 - (double)wa * t0) * (1.0 - (double)wc * t1) * (1.0 - (double)wl * t2);
        losses[m] = *err_flag ? __longlong_as_double(0x7ff8000000000000ll) : l;   // NaN = "edge index out of range"
    }
}

// ---- Losses.solution_score (losses.py:120-148): the three sums of the score of a 0/1 selection
//   s0 = sum_v predict[v] area_ratio[v]                 (:126: predict . (x[:, -1] max_area) / contour area)
//   s1 = sum_e predict[i_e] predict[j_e] len_ratio[e]   (:131-141: (p_i p_j) . (len max_align_length))
//   s2 = sum_{v: predict[v] == 1} perimeter[v]          (:143-144)
// products in fp32 like the reference, sums in fp64 over the fixed tree of the loss kernels.
__global__ __launch_bounds__(kLossThreads) void score_partial_kernel(
    const float *__restrict__ predict, const float *__restrict__ area, int64_t lda, const float *__restrict__ perim,
    int64_t n, const int64_t *__restrict__ adj, int64_t ea, const float *__restrict__ len, int64_t ldl,
    double *__restrict__ partial, int *__restrict__ err_flag) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kLossThreads, t0 = (int64_t)blockIdx.x * kLossThreads + threadIdx.x;
    for (int64_t v = t0; v < n; v += stride) {
        const float p = predict[v];
        s0 += (double)(p * area[v * lda]);
        if (p == 1.0f) s2 += (double)perim[v];
    }
    bool bad = false;
    for (int64_t e = t0; e < ea; e += stride) {
        const int64_t i = adj[e], j = adj[ea + e];
        if (i < 0 || i >= n || j < 0 || j >= n) { bad = true; continue; }
        s1 += (double)(predict[i] * predict[j] * len[e * ldl]);
    }
    if (bad) *err_flag = 1;
    __shared__ double red[3][kLossThreads];
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int d = kLossThreads / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(64) void score_final_kernel(const double *__restrict__ partial, int n_blocks,
                                                         const int *__restrict__ err_flag, double *__restrict__ sums) {
    const int lane = threadIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = lane; b < n_blocks; b += 64)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += partial[(int64_t)b * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[k] += __shfl_xor(s[k], d, 64);
    if (lane < 3) sums[lane] = *err_flag ? __longlong_as_double(0x7ff8000000000000ll) : s[lane];
}

static int loss_blocks(int64_t n, int64_t ec, int64_t ea) {
    int64_t work = n > ec ? n : ec;
    if (ea > work) work = ea;
    int64_t b = (work + kLossThreads * 4 - 1) / (kLossThreads * 4);
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    return (int)b;
}

}  // namespace tgnn

using namespace tgnn;

extern "C" size_t tgnn_unsupervised_loss_workspace_bytes(int32_t n_maps) {
    return (size_t)(n_maps > 0 ? n_maps : 1) * 512 * 3 * sizeof(double) + 256;     // partial rows + the error flag
}

extern "C" int tgnn_unsupervised_loss(const float *probs, int64_t ld_probs, int32_t n_maps, const float *area_ratio,
                                      int64_t ld_area, int64_t n_nodes, const int64_t *col_edge_index,
                                      int64_t n_col_edges, const int64_t *adj_edge_index, int64_t n_adj_edges,
                                      const float *adj_edge_len, int64_t ld_len, float collision_weight,
                                      float align_length_weight, float avg_area_weight, double *losses,
                                      double *terms, void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_maps >= 1 && n_maps <= 65535 && n_nodes >= 1, "shape");
    TGNN_CHECK_ARG(probs && area_ratio && losses && ld_probs >= n_maps && ld_area >= 1, "null pointer / strides");
    TGNN_CHECK_ARG(n_col_edges >= 0 && (n_col_edges == 0 || col_edge_index), "collision edges");
    TGNN_CHECK_ARG(n_adj_edges >= 0 && (n_adj_edges == 0 || (adj_edge_index && adj_edge_len && ld_len >= 1)), "adjacency edges");
    if (!ws || ws_bytes < tgnn_unsupervised_loss_workspace_bytes(n_maps)) {
        set_error("tgnn_unsupervised_loss: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(ws);
    int *err_flag = reinterpret_cast<int *>(partial + (size_t)n_maps * 512 * 3);
    TGNN_CHECK_HIP(hipMemsetAsync(err_flag, 0, sizeof(int), s));
    const int blocks = loss_blocks(n_nodes, n_col_edges, n_adj_edges);
    loss_partial_kernel<<<dim3(blocks, n_maps), kLossThreads, 0, s>>>(probs, ld_probs, area_ratio, ld_area, n_nodes,
                                                                      col_edge_index, n_col_edges, adj_edge_index,
                                                                      n_adj_edges, adj_edge_len, ld_len, partial, err_flag);
    loss_final_kernel<<<n_maps, 64, 0, s>>>(partial, blocks, n_nodes, n_col_edges, n_adj_edges, collision_weight,
                                            align_length_weight, avg_area_weight, losses, terms, err_flag);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_solution_score_sums(const float *predict, const float *area_ratio, int64_t ld_area,
                                        const float *perimeter, int64_t n_nodes, const int64_t *adj_edge_index,
                                        int64_t n_adj_edges, const float *adj_edge_len, int64_t ld_len, double *sums,
                                        void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && predict && area_ratio && perimeter && sums && ld_area >= 1, "null pointer / shape");
    TGNN_CHECK_ARG(n_adj_edges >= 0 && (n_adj_edges == 0 || (adj_edge_index && adj_edge_len && ld_len >= 1)), "adjacency edges");
    if (!ws || ws_bytes < tgnn_unsupervised_loss_workspace_bytes(1)) {
        set_error("tgnn_solution_score_sums: workspace too small (tgnn_unsupervised_loss_workspace_bytes(1))");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(ws);
    int *err_flag = reinterpret_cast<int *>(partial + (size_t)512 * 3);
    TGNN_CHECK_HIP(hipMemsetAsync(err_flag, 0, sizeof(int), s));
    const int blocks = loss_blocks(n_nodes, 0, n_adj_edges);
    score_partial_kernel<<<blocks, kLossThreads, 0, s>>>(predict, area_ratio, ld_area, perimeter, n_nodes, adj_edge_index,
                                                         n_adj_edges, adj_edge_len, ld_len, partial, err_flag);
    score_final_kernel<<<1, 64, 0, s>>>(partial, blocks, err_flag, sums);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
