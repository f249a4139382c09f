// Batched acceptance of the greedy assembly loop on the device (BASELINE config 5 "batched greedy selection").
//
// Reference: solve_by_probablistic_greedy, /root/reference/util/algorithms.py:18-62.  Per round it (:33-34) folds the round's
// probabilities into the running geometric mean  p_v = (saved_v^(r-1) * prob_v)^(1/r),  then (:41-54) walks the unlabelled
// nodes in DESCENDING p, stops at the first node an earlier acceptance of this round has labelled, and accepts a node when
// exp(p_v - 1) > np.random.uniform(); an accepted node labels its collision neighbours (label_collision_neighbor, :196-207).
// That sweep is sequential by definition and consumes numpy's global RNG stream; tilingnn_amd.util.algorithms keeps it on the
// host for seeded parity with the reference.  It visits ~sqrt(N / mean collision degree) nodes per round, so a layout of
// 10^5..10^6 nodes takes hundreds to thousands of rounds, each a full forward.
//
// THE SUBSTITUTE (documented, not bit-compatible -- SURVEY 8f-1 allows one): every round accepts ALL nodes that
//   (1) beat every unlabelled collision neighbour in the reference's visiting order (larger p first, the smaller node number on
//       ties) -- the nodes the sequential sweep could reach before any of their neighbours --, and
//   (2) pass the reference's test  exp(p_v - 1) > u_v  with u_v uniform in [0, 1) from a counter-based generator keyed by
//       (seed, round, ORIGINAL node number): seeded, reproducible, independent of the launch geometry.
// No two accepted nodes collide (of two neighbours at most one beats the other), every accepted node labels its neighbours, so
// the result is a collision-free selection that is maximal when the loop has run dry -- the invariants the reference's loop
// guarantees.  A round labels a constant fraction of the remaining nodes: O(log N) rounds instead of O(sqrt N)..O(N).
// Four small launches per round, everything stays on the device; the caller reads back one count.
#include "tgnn_common.h"

namespace tgnn {

constexpr int kGrThreads = 256;

// splitmix64 of (seed, round, node) -> uniform double in [0, 1)
__device__ __forceinline__ double greedy_uniform(unsigned long long seed, unsigned round, unsigned long long node) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (node + 1ull) + 0xD1B54A32D192ED03ull * (unsigned long long)(round + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// p_v of the round (algorithms.py:33-34), saved for the next one; flags cleared
__global__ void greedy_mean_kernel(const float *__restrict__ prob, int64_t ldp, const int64_t *__restrict__ inverse, int64_t n_sub,
                                   int round, double *__restrict__ saved, double *__restrict__ p, int *__restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * kGrThreads + threadIdx.x; i < n_sub; i += (int64_t)gridDim.x * kGrThreads) {
        const int64_t o = inverse ? inverse[i] : i;
        const double v = pow(pow(saved[o], (double)(round - 1)) * (double)prob[i * ldp], 1.0 / (double)round);
        saved[o] = v;
        p[i] = v;
        flags[i] = 0;                                             // bit 0: beaten by a neighbour, bit 1: accepted
    }
}
// a directed collision edge (u, v) of the sub-layout: u is beaten when v comes first in the visiting order
__global__ void greedy_beaten_kernel(const int64_t *__restrict__ col, int64_t ec, int64_t n_sub, const double *__restrict__ p,
                                     int *__restrict__ flags, int *__restrict__ err) {
    for (int64_t e = (int64_t)blockIdx.x * kGrThreads + threadIdx.x; e < ec; e += (int64_t)gridDim.x * kGrThreads) {
        const int64_t u = col[e], v = col[ec + e];
        if (u < 0 || u >= n_sub || v < 0 || v >= n_sub) { *err = 1; continue; }
        if (u == v) continue;                                     // (GINConv drops self loops; so does the collision test)
        const double pu = p[u], pv = p[v];
        // the later of the two in the visiting order is beaten -- marked from EITHER direction of the pair, so that a collision
        // stored one way only (the reference stores both, tile_graph.py:206-207) still keeps the two apart
        if (pv > pu || (pv == pu && v < u)) flags[u] = 1;         // (every writer stores the same word)
        else flags[v] = 1;
    }
}
__global__ void greedy_accept_kernel(const int64_t *__restrict__ inverse, int64_t n_sub, const double *__restrict__ p, int round,
                                     unsigned long long seed, int *__restrict__ flags, int *__restrict__ alive,
                                     int *__restrict__ selected, long long *__restrict__ count) {
    int mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * kGrThreads + threadIdx.x; i < n_sub; i += (int64_t)gridDim.x * kGrThreads) {
        if (flags[i] & 1) continue;
        const int64_t o = inverse ? inverse[i] : i;
        if (exp((p[i] - 1.0) * 1.0) > greedy_uniform(seed, (unsigned)round, (unsigned long long)o)) {   // (algorithms.py:51)
            flags[i] = 2;
            alive[o] = 0;
            selected[o] = round;
            ++mine;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(reinterpret_cast<unsigned long long *>(count), (unsigned long long)mine);
}
// label_collision_neighbor (algorithms.py:196-207): the neighbours of an accepted node leave the layout
__global__ void greedy_label_kernel(const int64_t *__restrict__ col, int64_t ec, int64_t n_sub, const int64_t *__restrict__ inverse,
                                    const int *__restrict__ flags, int *__restrict__ alive) {
    for (int64_t e = (int64_t)blockIdx.x * kGrThreads + threadIdx.x; e < ec; e += (int64_t)gridDim.x * kGrThreads) {
        const int64_t u = col[e], v = col[ec + e];
        if (u < 0 || u >= n_sub || v < 0 || v >= n_sub) continue;
        if (flags[u] & 2) alive[inverse ? inverse[v] : v] = 0;
    }
}

// [r6] The END of a solve as ONE launch: a sub-layout without adjacency edges (or without collision edges) gets probability 1 for
// every node from ML_Solver.predict without the network (ml_solver.py:31-32), and so does every sub-layout of it -- the remaining
// rounds are the four kernels above with prob = 1 on ever smaller sub-layouts.  One block runs them all on the sub-layout it is
// given: nodes that have left (alive == 0) and the edges at them are skipped instead of compacted away; the geometric mean, the
// visiting order (ties by node number: compaction keeps the order), the draws (keyed by seed, round and ORIGINAL node number) and
// the round numbers are those of the round-by-round path, hence the same selection, order and round count.
constexpr int kFinThreads = 1024;
constexpr int kFinMaxNodes = 4096;
__global__ __launch_bounds__(kFinThreads) void greedy_finish_kernel(const int64_t *__restrict__ inverse, int n_sub,
                                                                    const int64_t *__restrict__ col, int64_t ec, int first_round,
                                                                    int max_rounds, unsigned long long seed, double *__restrict__ saved,
                                                                    int *__restrict__ alive, int *__restrict__ selected,
                                                                    long long *__restrict__ count, int *__restrict__ err,
                                                                    int *__restrict__ out) {   // out[0] rounds run, out[1] nodes left
    __shared__ double p[kFinMaxNodes];
    __shared__ __attribute__((aligned(16))) unsigned char flags[kFinMaxNodes];   // bit 0 beaten, bit 1 accepted, bit 2 gone, bit 3 leaving
    __shared__ int left, accepted;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_sub; i += kFinThreads) flags[i] = alive[inverse ? inverse[i] : i] ? 0 : 4;
    __syncthreads();
    int round = first_round, rounds_run = 0;
    for (;; ++round) {
        if (tid == 0) { left = 0; accepted = 0; }
        __syncthreads();
        int mine = 0;
        for (int i = tid; i < n_sub; i += kFinThreads) {
            if (flags[i] & 4) continue;
            ++mine;
            const int64_t o = inverse ? inverse[i] : i;
            const double v = pow(pow(saved[o], (double)(round - 1)) * 1.0, 1.0 / (double)round);     // (greedy_mean_kernel, prob = 1)
            saved[o] = v;
            p[i] = v;
            flags[i] = 0;
        }
        if (mine) atomicAdd(&left, mine);
        __syncthreads();
        if (left == 0 || rounds_run >= max_rounds) break;    // (uniform)
        ++rounds_run;
        for (int64_t e = tid; e < ec; e += kFinThreads) {      // greedy_beaten_kernel
            const int64_t u = col[e], v = col[ec + e];
            if (u < 0 || u >= n_sub || v < 0 || v >= n_sub) { *err = 1; continue; }
            if (u == v || ((flags[u] | flags[v]) & 4)) continue;
            const double pu = p[u], pv = p[v];
            if (pv > pu || (pv == pu && v < u)) atomicOr(reinterpret_cast<unsigned *>(flags) + (u >> 2), 1u << (8 * (u & 3)));
            else atomicOr(reinterpret_cast<unsigned *>(flags) + (v >> 2), 1u << (8 * (v & 3)));
        }
        __syncthreads();
        int acc = 0;
        for (int i = tid; i < n_sub; i += kFinThreads) {       // greedy_accept_kernel
            if (flags[i] & 5) continue;
            const int64_t o = inverse ? inverse[i] : i;
            if (exp((p[i] - 1.0) * 1.0) > greedy_uniform(seed, (unsigned)round, (unsigned long long)o)) {
                flags[i] |= 2;
                alive[o] = 0;
                selected[o] = round;
                ++acc;
            }
        }
        if (acc) atomicAdd(&accepted, acc);
        __syncthreads();
        for (int64_t e = tid; e < ec; e += kFinThreads) {      // greedy_label_kernel
            const int64_t u = col[e], v = col[ec + e];
            if (u < 0 || u >= n_sub || v < 0 || v >= n_sub) continue;
            if ((flags[u] & 2) && !(flags[v] & 4)) {
                alive[inverse ? inverse[v] : v] = 0;
                atomicOr(reinterpret_cast<unsigned *>(flags) + (v >> 2), 8u << (8 * (v & 3)));   // bit 3: leaves behind this round
            }
        }
        __syncthreads();
        for (int i = tid; i < n_sub; i += kFinThreads)
            if (flags[i] & (2 | 8)) flags[i] = 4;
        if (tid == 0 && accepted) atomicAdd(reinterpret_cast<unsigned long long *>(count), (unsigned long long)accepted);
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = rounds_run;
        out[1] = left;
    }
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int64_t tgnn_greedy_finish_max_nodes(void) { return kFinMaxNodes; }

/* The remaining rounds of a greedy solve on a sub-layout whose nodes all get probability 1 (no adjacency edge or no collision edge
 * left: ml_solver.py:31-32) as one launch; see greedy_finish_kernel.  n_sub <= tgnn_greedy_finish_max_nodes().  out [2] (device):
 * rounds run (first_round, first_round + 1, ...), nodes still unlabelled (0 unless max_rounds ran out). */
extern "C" int tgnn_greedy_finish(const int64_t *inverse, int64_t n_sub, const int64_t *col_edge_index, int64_t n_col_edges,
                                  int32_t first_round, int32_t max_rounds, uint64_t seed, double *prob_saved, int32_t *alive,
                                  int32_t *selected_round, int64_t *n_selected, int32_t *err_flag, int32_t *out, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_sub >= 1 && n_sub <= kFinMaxNodes && n_col_edges >= 0 && first_round >= 1 && max_rounds >= 1, "shape");
    TGNN_CHECK_ARG(prob_saved && alive && selected_round && n_selected && err_flag && out, "null pointer");
    TGNN_CHECK_ARG(n_col_edges == 0 || col_edge_index, "null edge index");
    greedy_finish_kernel<<<1, kFinThreads, 0, static_cast<hipStream_t>(stream)>>>(
        inverse, (int)n_sub, col_edge_index, n_col_edges, first_round, max_rounds, (unsigned long long)seed, prob_saved, alive,
        selected_round, reinterpret_cast<long long *>(n_selected), err_flag, out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" size_t tgnn_greedy_round_workspace_bytes(int64_t n_sub) {
    return align_up((size_t)(n_sub > 0 ? n_sub : 1) * sizeof(double), 256) + align_up((size_t)(n_sub > 0 ? n_sub : 1) * sizeof(int), 256) + 256;
}

extern "C" int tgnn_greedy_round(const float *prob, int64_t ld_prob, const int64_t *inverse, int64_t n_sub, const int64_t *col_edge_index,
                                 int64_t n_col_edges, int32_t round, uint64_t seed, double *prob_saved, int32_t *alive,
                                 int32_t *selected_round, int64_t *n_selected, int32_t *err_flag, void *ws, size_t ws_bytes,
                                 tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_sub >= 0 && n_col_edges >= 0 && round >= 1 && ld_prob >= 1, "shape");
    if (n_sub == 0) return TGNN_OK;
    TGNN_CHECK_ARG(prob && prob_saved && alive && selected_round && n_selected && err_flag, "null pointer");
    TGNN_CHECK_ARG(n_col_edges == 0 || col_edge_index, "null edge index");
    if (!ws || ws_bytes < tgnn_greedy_round_workspace_bytes(n_sub)) {
        set_error("tgnn_greedy_round: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    Carver cv(ws, ws_bytes);
    double *p = cv.take<double>(n_sub);
    int *flags = cv.take<int>(n_sub);
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto grid = [](int64_t n) {
        int64_t g = (n + kGrThreads - 1) / kGrThreads;
        return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
    };
    greedy_mean_kernel<<<grid(n_sub), kGrThreads, 0, s>>>(prob, ld_prob, inverse, n_sub, round, prob_saved, p, flags);
    if (n_col_edges > 0)
        greedy_beaten_kernel<<<grid(n_col_edges), kGrThreads, 0, s>>>(col_edge_index, n_col_edges, n_sub, p, flags, err_flag);
    greedy_accept_kernel<<<grid(n_sub), kGrThreads, 0, s>>>(inverse, n_sub, p, round, (unsigned long long)seed, flags, alive,
                                                           selected_round, reinterpret_cast<long long *>(n_selected));
    if (n_col_edges > 0)
        greedy_label_kernel<<<grid(n_col_edges), kGrThreads, 0, s>>>(col_edge_index, n_col_edges, n_sub, inverse, flags, alive);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
