// TilinGNN.forward (/root/reference/graph_networks/networks/TilinGNN.py:54-76) for SMALL layouts -- init MLP, the 20 message-passing
// layers, final MLP -- as ONE persistent kernel with grid-wide barriers.
//
// Why: the layouts the greedy solver actually scores (1 254 nodes for the labyrinth example, shrinking every round) have
// ~80 16-row tiles -- fewer than the chip has CUs.  The general schedule (forward.hip) spends such a forward in ~130
// dependent kernel launches at ~5 us of GPU-side dispatch latency each (measured: hipGraph replay does not help, the
// gap is between dependent dispatches, not in the host's launch path).  Here one block (8 waves) owns one 16-row tile
// from the first Linear to the sigmoid.  A message-passing layer is
//     phase A   NNConv: two waves gather the tile's source rows -- whole 128-byte rows, 8 per load instruction, a row's edges
//               as a list -- into per-type sum tiles in LDS, then six waves multiply them by the types' weight fragments
//               (straight from the image in global memory, used once per tile) and two waves finish the mean / bias / LeakyReLU.
//               Beside them one wave per half tile runs the whole collision branch: gather, neighbourhood sum, GIN MLP.
//               BatchNorm column sums of the tile -> one partial row
//     barrier   (all blocks)
//     phase B   every block folds all partial rows in the same fixed order -> both BatchNorm records; merge of its own
//               rows (BN1(a1) * BN2(a2) + residual) -> the next slot of the skip buffer
//     barrier
// i.e. two barriers (~1.5 us each at 80 blocks) instead of five dependent launches; the dense layers of the init / final MLP
// run on the tile's own rows with one barrier per BatchNorm.
//
// Cross-block data (skip-buffer rows, the collision branch's pre-BN rows, the partial rows) is written and read with sc1
// (agent-scope) buffer instructions: each XCD has its own L2, and the release/acquire fences that would make ordinary
// accesses visible across XCDs (buffer_wbl2 / buffer_inv) serialise per XCD -- 10.8 us per barrier at 256 blocks against
// 3.8 us without them (scratch/ubench/gridsync.hip).  The barrier itself is a monotonic counter.  All blocks must be
// resident at once: small_layout_teams() checks the device's capacity (see there why this is not a cooperative launch).
//
// Arithmetic: the same formulas as the general path's kernels (nnconv_cols.hip, gin.hip, dense.hip, bn_merge.hip); what differs
// is the association of sums (NNConv: six partial products per tile; BatchNorm: one partial row per tile; dense layers: K in
// steps of 32), i.e. fp32 / fp64 rounding only.  Deterministic: every order is fixed.
#include <mutex>

#include <vector>

#include "forward_persist.h"

namespace tgnn {

struct SmallPackLayer {
    const float *nn_bias, *g1, *b1, *g2, *b2, *eps, *w1, *gb1, *w2, *gb2, *w3, *gb3;
};
constexpr int kSmallPackChunk = 32;
struct SmallPackLayers {
    SmallPackLayer l[kSmallPackChunk];
};

__device__ __forceinline__ int small_kf(int q, int e) { return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4); }   // gin.hip: gin_kf

struct SmallImageJob {
    const float *w;              // [m][k] row-major (torch Linear.weight)
    float *img;
    int m, k;                    // multiples of 16 / 32
    const unsigned *f16_max;     // NULL: bf16 x 3 planes; else fp16 pairs of w * pow2_scale_for(*f16_max) (forward_tail.hip, layer 0)
};
struct SmallImageJobs {
    SmallImageJob j[5];
};
// The per-forward pre-pass of the persistent kernel, one launch: blockIdx.y < n_layers: the parameter pack of that layer
// (block x = 0; it also re-arms the barrier counter); above: the MFMA image of one dense layer, one thread per (mb, ks, lane)
__global__ __launch_bounds__(256) void small_pack_kernel(SmallPackLayers layers, int n_layers, float *__restrict__ pack,
                                                         unsigned *__restrict__ barrier_ctr, SmallImageJobs jobs,
                                                         u32x4 *__restrict__ zero, int64_t zero_n) {
    const int tid = threadIdx.x;
    // (the mid-size kernel's tagged partial rows must carry no valid tag before its launch: the layers' blocks clear them)
    if (zero && (int)blockIdx.y < n_layers && blockIdx.x == 0)
        for (int64_t i = (int64_t)blockIdx.y * 256 + tid; i < zero_n; i += (int64_t)n_layers * 256) zero[i] = u32x4{0u, 0u, 0u, 0u};
    if ((int)blockIdx.y >= n_layers) {
        const SmallImageJob J = jobs.j[blockIdx.y - n_layers];
        const int ksteps = J.k / 32, items = (J.m / 16) * ksteps * 64;
        for (int it = blockIdx.x * 256 + tid; it < items; it += gridDim.x * 256) {
            const int lane = it & 63, blk = it >> 6, ks = blk % ksteps, mb = blk / ksteps;
            const int i = lane & 15, q = lane >> 4;
            const float4 *src = reinterpret_cast<const float4 *>(J.w + (size_t)(16 * mb + i) * J.k + 32 * ks + 8 * q);
            const float4 a = src[0], b = src[1];
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (J.f16_max) {
                tgnn_f16x8 *dst = reinterpret_cast<tgnn_f16x8 *>(J.img) + (size_t)blk * 2 * 64 + lane;
                split2_f16(x, pow2_scale_for(*J.f16_max, 0), dst[0], dst[64]);
                continue;
            }
            bf16x8 *dst = reinterpret_cast<bf16x8 *>(J.img) + (size_t)blk * 3 * 64 + lane;
            split3_trunc(x, dst[0], dst[64], dst[128]);
        }
        return;
    }
    if (blockIdx.x != 0) return;
    const SmallPackLayer L = layers.l[blockIdx.y];
    float *out = pack + (size_t)blockIdx.y * kSpStride;
    if (barrier_ctr && blockIdx.y == 0 && tid == 0) *barrier_ctr = 0u;
    if (tid < 32) {
        out[kSpBias + tid] = L.nn_bias[tid];
        out[kSpG1 + tid] = L.g1[tid];
        out[kSpB1 + tid] = L.b1[tid];
        out[kSpG2 + tid] = L.g2[tid];
        out[kSpB2 + tid] = L.b2[tid];
        out[kSpEps + tid] = 1.0f + L.eps[0];
        out[kSpGinB + tid] = L.gb1[tid];
        out[kSpGinB + 96 + tid] = L.gb3[tid];
    }
    if (tid < 64) out[kSpGinB + 32 + tid] = L.gb2[tid];
    // the MFMA images of gin32_mlp_kernel (gin.hip): [plane 3][M block][i 16][q 4] x 8 bf16, K of layers 2 / 3 in kf order
    bf16x8 *W1s = reinterpret_cast<bf16x8 *>(out + kSpGinW), *W2s = W1s + 3 * 2 * 64, *W3s = W2s + 3 * 4 * 64;
    for (int i = tid; i < 2 * 64; i += 256) {
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = L.w1[(16 * mb + ii) * 32 + 8 * q + e];
        split3_trunc(x, W1s[(0 * 2 + mb) * 64 + ii * 4 + q], W1s[(1 * 2 + mb) * 64 + ii * 4 + q], W1s[(2 * 2 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += 256) {
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = L.w2[(16 * mb + ii) * 32 + small_kf(q, e)];
        split3_trunc(x, W2s[(0 * 4 + mb) * 64 + ii * 4 + q], W2s[(1 * 4 + mb) * 64 + ii * 4 + q], W2s[(2 * 4 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += 256) {
        const int mb = i >> 7, ks = (i >> 6) & 1, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = L.w3[(16 * mb + ii) * 64 + 32 * ks + small_kf(q, e)];
        const int o = (mb * 2 + ks) * 64 + ii * 4 + q;
        split3_trunc(x, W3s[0 * 256 + o], W3s[1 * 256 + o], W3s[2 * 256 + o]);
    }
}

// ---- the persistent kernel ------------------------------------------------------------------------------------------------
struct SmallArgs {
    float *mid;                  // skip buffer [depth + 1][n][32]; slot 0 filled by the init MLP
    float *a2[2];                // collision branch, pre-BatchNorm rows, two-deep
    const float *wimg;           // NNConv MFMA weight images [depth][(T + 1)][kWtType]
    const float *pack;           // [depth][kSpStride]
    const int *tile_col_ptr, *col_meta, *col_src;   // NNConv column structure (graph_prep.hip)
    const int *col_rowptr, *col_nbr;                // collision CSR by destination
    double *part;                // [blocks][128]: bn1 sum | bn1 sumsq | bn2 sum | bn2 sumsq
    double *runstat;             // [depth][2][mean 32 | unbiased variance 32]: batch statistics of the layers, for the running buffers
    unsigned *ctr;               // barrier counter (zeroed by small_pack_kernel)
    const unsigned *weights_done;   // NULL, or: blocks of the edge-weight kernel that have finished (the kernel may start before them)
    unsigned weights_target;
    unsigned *err;               // the device's spin-error word (forward_persist.h: no wait of this kernel spins without bound)
    unsigned *err_host;          // ... and its host-mapped mirror (may be NULL)
    unsigned long long spin_budget;
    int64_t n;
    int n_types, depth, update_running, fault;
    float eps, momentum;
};

// ---- the init MLP (TilinGNN.py:54) and the final MLP (:74-76) inside the same kernel -----------------------------------
// One BatchNorm'd Linear_trans: MFMA image of the weights [M / 16][K / 32][plane 3][lane 64] x 8 bf16 (the A fragment of
// lane (i, q) for output block mb, K step ks: W[16 mb + i][32 ks + 8 q .. + 7], split hi / mid / lo), bias, BatchNorm
struct SmallDense {
    const float *img, *bias, *gamma, *beta;
    float *rm, *rv;
    int64_t *nbt;
};
struct SmallEnds {
    const float *x;              // node features [n][fx]
    const float *w0;             // init Linear 0 [32][fx] (computed on the vector pipe: fx = tile_count + 1 is tiny)
    SmallDense i0, i1;           // init Linear_trans 0 (img unused), 1 (32 -> 32)
    SmallDense f[4];             // final MLP: 32 (depth + 1) -> 256 -> 128 -> 64 -> 32
    const float *w_last, *b_last;   // final_mlp.1: Linear(32, out_dim) + sigmoid, no BatchNorm
    float *probs;                // [n][out_dim]
    double *part_wide;           // [2][256][512]: column sums | sums of squares of one dense layer, two sets
    int fx, out_dim;
};
#ifdef TGNN_SMALL_TIMING
// phase timers of block 0 / the slowest arrival (scratch builds only): wall_clock64 ticks (100 MHz), summed over the layers
__device__ unsigned long long g_small_timing[32 * 260];
#define TGNN_ST(slot) { const unsigned long long now_ = wall_clock64(); if (tid == 0) tacc[slot] += now_ - tlast; tlast = now_; }
#define TGNN_ST2(slot) { const unsigned long long now_ = wall_clock64(); tacc2[slot] += now_ - tlast2; tlast2 = now_; }
#define TGNN_ST2_RESET { tlast2 = wall_clock64(); }
#define TGNN_ST3(slot) { const unsigned long long now_ = wall_clock64(); tacc3[slot] += now_ - tlast3; tlast3 = now_; }
#define TGNN_ST3_RESET { tlast3 = wall_clock64(); }
#else
#define TGNN_ST(slot)
#define TGNN_ST2(slot)
#define TGNN_ST2_RESET
#define TGNN_ST3(slot)
#define TGNN_ST3_RESET
#endif

// gathers of rows other blocks wrote in the previous phase
#ifdef TGNN_SMALL_CACHED
constexpr int kCpGather = 0;     // through L2: needs the invalidate after barrier 2
#else
constexpr int kCpGather = kCpSc1;
#endif
__device__ __forceinline__ float4 ld_gather_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kCpGather));
}
// All blocks are resident (small_layout_teams checks the device's capacity).  Stores of this block are acknowledged (vmcnt(0)) before its arrival
// is published; the data itself is sc1, so no cache maintenance is needed on either side.
__device__ __forceinline__ void small_grid_barrier(unsigned *ctr, unsigned &target, unsigned nblk, SpinCtx &sp) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    target += nblk;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until_ge(ctr, target, sp, kSpinErrBarrier);       // bounded: forward_persist.h
    }
    __syncthreads();
}

constexpr int kColFirst = 1 << 8;    // col_meta flag (graph_prep.hip): first column of a run of same-type columns; low byte = type
constexpr int kSmallThreads = 512;   // 8 waves per tile: 6 NNConv column chunks, 2 halves of the collision neighbourhoods
constexpr int kNnWaves = 6;
#ifndef TGNN_SMALL_RUN_WAVES
#define TGNN_SMALL_RUN_WAVES 6
#endif
// Stage 2 of the NNConv (matrix products of the tile's runs) on all six NNConv waves.  (4: only on waves 0, 1, 4, 5, so that
// waves 2, 3, which share their SIMDs with the two collision waves -- the longer chain --, only do the short epilogue:
// measured equal, 0.337 vs 0.337 ms on the labyrinth layout; six waves take up to 18 runs.)
constexpr int kRunWaves = TGNN_SMALL_RUN_WAVES, kRunsPerWave = kRunWaves == 4 ? 4 : 3;
// LDS, floats, after the weight images: parameter vectors of two layers | NNConv partial products [6][64][8] (phase B: the
// fp64 fold [8][128]) | second half of the collision sums [64][8] | a1 tile | a2 tile | records | root degrees
constexpr int kLdsSpv = 2 * kSpGinW, kLdsNnRed = kNnWaves * 64 * 8, kLdsGinRed = 64 * 8, kLdsTile = 512;
#ifndef TGNN_SMALL_NN_DELAY
#define TGNN_SMALL_NN_DELAY 16           // x 64 clocks
#endif
constexpr int kGinCached = 16;        // collision neighbours per row whose gather offsets stay in registers
constexpr int kPfG = 4;               // float4 per thread of the next layer's GIN weight image held across barrier 1
constexpr int kNnEntries = 32;        // gather entries per row (adjacency in-edges + the root row): in-degree <= 31

// acc (D^T tile pair) += W_t^T . (a * scale)^T for one run of same-type columns: bf16 x 3 split, six cross terms per M block
__device__ __forceinline__ void small_run_mma(const float *wl, int t, int lane, const float (&af)[8], float scale, f32x4 &d0,
                                              f32x4 &d1) {
    bf16x8 xh, xm, xl;
    {
        float as[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) as[k] = af[k] * scale;
        split3_trunc(as, xh, xm, xl);
    }
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wl + t * kWtType) + lane;
    constexpr int kPl = kWtPlane / 4;
    const bf16x8 h0 = wp[0], h1 = wp[64], m0 = wp[kPl], m1 = wp[kPl + 64], l0 = wp[2 * kPl], l1 = wp[2 * kPl + 64];
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, xh, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, xh, d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xl, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xl, d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m0, xm, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m1, xm, d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m0, xh, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m1, xh, d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xm, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xm, d1, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xh, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xh, d1, 0, 0, 0);
}

// ---- dense layers on one 16-row tile, block-wide (8 waves) ------------------------------------------------------------------
// LDS of these phases (the weight images of the message-passing layers are not live then), floats from the start:
//   B-operand planes of the layer's input [K / 32][3][64] x 16 B (capacity: max(depth + 1, 8) K steps) | activations of the
//   tile [16][kDActLd] fp32 | fp64 fold [512 / M][2 M] = 1024 doubles | BatchNorm record [4][256]
constexpr int kDActLd = 260, kDPlaneStep = 3 * 64 * 4;
constexpr int kSmallMaxDepth = 40;
constexpr size_t kPartWideSet = (size_t)256 * 512;   // doubles of one set of wide partial rows (two sets: see small_tile_bn)
__host__ __device__ constexpr int small_dense_ksteps(int depth) { return depth + 1 > 8 ? depth + 1 : 8; }
__host__ __device__ constexpr int small_dense_lds_floats(int depth) {
    return small_dense_ksteps(depth) * kDPlaneStep + 16 * kDActLd + 2048 + 1024;
}
// the type-sum tiles of the layer loop sit behind both layouts (they are zeroed once and must not be touched by anything else)
constexpr int kLoopFixedFloats = kSpGinFrags * 4 + kLdsSpv + kLdsNnRed + kLdsGinRed + 2 * kLdsTile + 256 + 16 + 32 + 16 * kNnEntries * 2 +
                                 16 * kGinCached;
__host__ __device__ constexpr int small_s_offset(int depth) {
    return small_dense_lds_floats(depth) > kLoopFixedFloats ? small_dense_lds_floats(depth) : kLoopFixedFloats;
}

// out[n][16 mb + 4 q + r] = LeakyReLU(sum_k W[.][k] x[n][k] + b): D^T = W . X^T, split precision (bf16 x 3, six cross terms).
// Output blocks go round the waves, MBW at a time (sharing the B fragments); A fragments stream from the image in global
// memory one K step ahead.
template <int MBW>
__device__ __forceinline__ void small_tile_dense(const float *img, int ksteps, int mblocks, const float *bias, const float *lds_x,
                                                 float *act, int tw, int lane) {
    const int fn = lane & 15, fq = lane >> 4;
    const bf16x8 *xp = reinterpret_cast<const bf16x8 *>(lds_x) + lane;
    for (int mb0 = tw * MBW; mb0 < mblocks; mb0 += 8 * MBW) {
        f32x4 acc[MBW];
        const bf16x8 *wp[MBW];
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
            const int mb = mb0 + m < mblocks ? mb0 + m : mblocks - 1;
            const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * mb + 4 * fq);
            acc[m] = f32x4{b.x, b.y, b.z, b.w};
            wp[m] = reinterpret_cast<const bf16x8 *>(img) + (size_t)mb * ksteps * 3 * 64 + lane;
        }
        bf16x8 cur[MBW][3], nxt[MBW][3];
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) cur[m][pl] = wp[m][pl * 64];
        for (int ks = 0; ks < ksteps; ++ks) {
            const int kn = ks + 1 < ksteps ? ks + 1 : ks;
#pragma unroll
            for (int m = 0; m < MBW; ++m)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) nxt[m][pl] = wp[m][(kn * 3 + pl) * 64];
            const bf16x8 x0 = xp[(ks * 3 + 0) * 64], x1 = xp[(ks * 3 + 1) * 64], x2 = xp[(ks * 3 + 2) * 64];
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][2], x0, acc[m], 0, 0, 0);   // lo . hi
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][0], x2, acc[m], 0, 0, 0);   // hi . lo
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][1], x1, acc[m], 0, 0, 0);   // mid . mid
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][1], x0, acc[m], 0, 0, 0);   // mid . hi
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][0], x1, acc[m], 0, 0, 0);   // hi . mid
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[m][0], x0, acc[m], 0, 0, 0);   // hi . hi
            }
#pragma unroll
            for (int m = 0; m < MBW; ++m)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) cur[m][pl] = nxt[m][pl];
        }
#pragma unroll
        for (int m = 0; m < MBW; ++m)
            if (mb0 + m < mblocks)
                *reinterpret_cast<float4 *>(act + fn * kDActLd + 16 * (mb0 + m) + 4 * fq) =
                    make_float4(leakyf_(acc[m][0]), leakyf_(acc[m][1]), leakyf_(acc[m][2]), leakyf_(acc[m][3]));
    }
}

// B-operand planes of BatchNorm(act) for the next dense layer: item (ks, lane (n, q)) = floats 32 ks + 8 q .. + 7 of row n
__device__ __forceinline__ void small_tile_planes_from_act(const float *act, const float *rec, int m_in, int valid_rows, float *lds_x,
                                                           int tid) {
    const int items = (m_in / 32) * 64;
    for (int it = tid; it < items; it += kSmallThreads) {
        const int lane = it & 63, ks = it >> 6, n = lane & 15, q = lane >> 4, k0 = 32 * ks + 8 * q;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            x[e] = n < valid_rows ? bn_apply1(act[n * kDActLd + k0 + e], rec[k0 + e], rec[256 + k0 + e], rec[512 + k0 + e], rec[768 + k0 + e]) : 0.f;
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(lds_x) + (size_t)ks * 3 * 64 + lane;
        split3_trunc(x, dst[0], dst[64], dst[128]);
    }
}

// Train-mode BatchNorm statistics of act [valid_rows][M] over ALL tiles: this block's column sums -> its partial row -> grid
// barrier -> every block folds all rows in the same order -> record [4][256] in LDS (block 0 updates the running buffers).
// Consecutive calls alternate between two sets of partial rows (`part_wide` = the caller's set for this call): nothing but this
// one barrier separates a fast block's next row from a slow block's reads of the current ones.
template <int M>
__device__ __forceinline__ void small_tile_bn(const float *act, int valid_rows, const SmallDense &L, double *part_wide, double *red,
                                              float *rec, int64_t n_total, float eps, float momentum, int update_running,
                                              unsigned *ctr, unsigned &target, unsigned nblk, SpinCtx &sp, int tid) {
    static_assert(M == 32 || M == 64 || M == 128 || M == 256, "width");
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(part_wide);
    if (tid < 2 * M) {
        const int ch = tid % M;
        const bool sq = tid >= M;
        double acc = 0.0;
        for (int r = 0; r < valid_rows; ++r) {
            const double v = (double)act[r * kDActLd + ch];
            acc += sq ? v * v : v;
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, acc), rs, ((uint32_t)blockIdx.x * 512u + (uint32_t)tid) * 8u, 0, kCpSc1);
    }
    small_grid_barrier(ctr, target, nblk, sp);
    {
        constexpr int groups = kSmallThreads / M;              // thread = (column pair, row group): 2 M doubles per row = M pairs
        const int jp = tid % M, g = tid / M;
        double acc0 = 0.0, acc1 = 0.0;
        for (int p0 = g; p0 < (int)nblk; p0 += 16 * groups) {
            u32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + u * groups;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, p < (int)nblk ? ((uint32_t)p * 512u + 2u * (uint32_t)jp) * 8u : kOob, 0, kCpSc1);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc0 += __builtin_bit_cast(double, u32x2{v[u][0], v[u][1]});
                acc1 += __builtin_bit_cast(double, u32x2{v[u][2], v[u][3]});
            }
        }
        red[g * 2 * M + 2 * jp] = acc0;
        red[g * 2 * M + 2 * jp + 1] = acc1;
        __syncthreads();
        if (tid < M) {
            double t_sum = 0.0, t_sq = 0.0;
#pragma unroll
            for (int gg = 0; gg < groups; ++gg) {
                t_sum += red[gg * 2 * M + tid];
                t_sq += red[gg * 2 * M + M + tid];
            }
            const double inv_n = 1.0 / (double)n_total;
            const double mean = t_sum * inv_n;
            double var = t_sq * inv_n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float mh = (float)mean;
            rec[tid] = mh;
            rec[256 + tid] = (float)(mean - (double)mh);
            rec[512 + tid] = (float)((double)L.gamma[tid] / sqrt(var + (double)eps));
            rec[768 + tid] = L.beta[tid];
            if (blockIdx.x == 0 && update_running) {
                const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
                L.rm[tid] = (float)((1.0 - (double)momentum) * (double)L.rm[tid] + (double)momentum * mean);
                L.rv[tid] = (float)((1.0 - (double)momentum) * (double)L.rv[tid] + (double)momentum * unbiased);
                if (tid == 0) *L.nbt += 1;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kSmallThreads) void forward_layers_small_kernel(SmallArgs A, SmallRunTab R, SmallEnds E) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = kSmallThreads;
    if (A.fault && blockIdx.x == gridDim.x - 1) return;         // (test hook: a block that never shows up)
    const int T = A.n_types, D = A.depth;
    const int64_t n = A.n;
    // LDS of the layer loop: GIN weight image | parameter vectors of two layers | NNConv partial products [6][64][8] (phase B:
    // the fp64 fold [8][128]; set-up: the entry lists) | collision z tiles [2][8][32] | a1 tile | a2 tile | records | root degrees |
    // run types; then, behind everything the init / final phases use, the type-sum tiles S [(T + 1)][16][32]
    float *gw = lds;                                              // GIN MLP weight image of the layer
    float *spv = gw + kSpGinFrags * 4;                            // [2][kSpGinW]: parameter vectors, by layer parity
    float *nnred = spv + kLdsSpv;
    float *ginred = nnred + kLdsNnRed;
    float *a1s = ginred + kLdsGinRed, *a2s = a1s + kLdsTile;
    float *st = a2s + kLdsTile;                                   // [2][4][32]: records of BN1, BN2
    float *rootdeg = st + 256;                                    // [16]
    int *run_type = reinterpret_cast<int *>(rootdeg + 16);        // [32] type of the tile's r-th run of same-type columns; [31] = count
    int2 *ent = reinterpret_cast<int2 *>(run_type + 32);          // [16 rows][kNnEntries] NNConv gather entries (see the set-up below)
    uint32_t *gnb = reinterpret_cast<uint32_t *>(ent + 16 * kNnEntries);   // [16 rows][kGinCached] collision neighbours' row offsets
    float *S = lds + small_s_offset(D);                           // type-sum tiles; empty slots stay zero for the whole kernel
    double *red = reinterpret_cast<double *>(nnred);              // phase B: [8][128]

    const int tid = threadIdx.x, lane = tid & 63, tw = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    const int64_t tile = blockIdx.x, my_row = tile * 16 + fj;
    const bool row_ok = my_row < n;
    const unsigned nblk = gridDim.x;
    unsigned target = 0;
    SpinCtx spin{A.err, A.spin_budget, false, A.err_host};
    const size_t slot = (size_t)n * 32;
    const __amdgpu_buffer_rsrc_t part_rs = rsrc_of(A.part);
#ifdef TGNN_SMALL_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    unsigned long long tacc2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast2 = tlast;
    unsigned long long tacc3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast3 = tlast;
#endif

    // ---- the GIN weight image and the parameter vectors of the next layer: global -> registers -> LDS (exact-size descriptor,
    // one offset register per thread; native vectors: a float4 struct is copied by memcpy, which pins arrays in scratch memory)
    u32x4 pfg[kPfG], pfs;
    const uint32_t pf_off = (uint32_t)tid * 16u;
#define TGNN_SMALL_PREFETCH(LAYER)                                                                                          \
    {                                                                                                                       \
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(                                             \
            const_cast<float *>(A.pack + (size_t)(LAYER) * kSpStride), 0, kSpStride * 4, 0x00020000);                      \
        _Pragma("unroll") for (int u = 0; u < kPfG; ++u)                                                                    \
            pfg[u] = __builtin_amdgcn_raw_buffer_load_b128(g_rs, pf_off, kSpGinW * 4 + u * NT * 16, 0);                     \
        pfs = __builtin_amdgcn_raw_buffer_load_b128(g_rs, tid < kSpGinW / 4 ? pf_off : 0u, 0, 0);                           \
    }
#define TGNN_SMALL_COMMIT(LAYER)                                                                                            \
    {                                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < kPfG; ++u)                                                                    \
            if (tid + u * NT < kSpGinFrags) reinterpret_cast<u32x4 *>(gw)[tid + u * NT] = pfg[u];                           \
        if (tid < kSpGinW / 4) reinterpret_cast<u32x4 *>(spv + ((LAYER) & 1) * kSpGinW)[tid] = pfs;                         \
    }

    // ====================================== init MLP (TilinGNN.py:54) ======================================
    // Linear(fx, 32) + LeakyReLU + BatchNorm, Linear(32, 32) + LeakyReLU + BatchNorm -> slot 0 of the skip buffer, own rows
    {
        float *dx = lds, *dact = lds + small_dense_ksteps(D) * kDPlaneStep;
        double *dred = reinterpret_cast<double *>(dact + 16 * kDActLd);
        float *drec = dact + 16 * kDActLd + 2048;
        const int valid_rows = n - tile * 16 < 16 ? (int)(n - tile * 16) : 16;
        {   // Linear 0 on the vector pipe: thread = (row, channel), k ascending
            const int r = tid >> 5, ch = tid & 31;
            float acc = E.i0.bias[ch];
            if (r < valid_rows)
                for (int k = 0; k < E.fx; ++k) acc = fmaf(E.x[(tile * 16 + r) * E.fx + k], E.w0[ch * E.fx + k], acc);
            dact[r * kDActLd + ch] = leakyf_(acc);
        }
        __syncthreads();
        small_tile_bn<32>(dact, valid_rows, E.i0, E.part_wide, dred, drec, n, A.eps, A.momentum, A.update_running, A.ctr, target, nblk, spin, tid);
        small_tile_planes_from_act(dact, drec, 32, valid_rows, dx, tid);
        __syncthreads();
        small_tile_dense<1>(E.i1.img, 1, 2, E.i1.bias, dx, dact, tw, lane);
        __syncthreads();
        small_tile_bn<32>(dact, valid_rows, E.i1, E.part_wide + kPartWideSet, dred, drec, n, A.eps, A.momentum, A.update_running, A.ctr, target, nblk, spin, tid);
        if (tid < 128) {
            const int row = tid >> 3, c4 = (tid & 7) * 4;
            if (row < valid_rows) {
                float4 o;
                o.x = bn_apply1(dact[row * kDActLd + c4 + 0], drec[c4 + 0], drec[256 + c4 + 0], drec[512 + c4 + 0], drec[768 + c4 + 0]);
                o.y = bn_apply1(dact[row * kDActLd + c4 + 1], drec[c4 + 1], drec[256 + c4 + 1], drec[512 + c4 + 1], drec[768 + c4 + 1]);
                o.z = bn_apply1(dact[row * kDActLd + c4 + 2], drec[c4 + 2], drec[256 + c4 + 2], drec[512 + c4 + 2], drec[768 + c4 + 2]);
                o.w = bn_apply1(dact[row * kDActLd + c4 + 3], drec[c4 + 3], drec[256 + c4 + 3], drec[512 + c4 + 3], drec[768 + c4 + 3]);
                st_sc1_f4(rsrc_of(A.mid), (uint32_t)(tile * 16 + row) * 128u + (uint32_t)c4 * 4u, o);
            }
        }
        small_grid_barrier(A.ctr, target, nblk, spin);                 // (also: everybody is done with the LDS of this phase)
    }
    TGNN_SMALL_PREFETCH(0)

    // ---- layer-invariant pieces of the tile, kept in registers for all layers
    // NNConv, stage 1 (waves 0, 1; gather mapping: lane = (row slot o = lane >> 3, 16-byte piece p = lane & 7), rows 8 w + o):
    //   the row's entries in column order -- source row offset and LDS address of the type-sum slot S[run][row] it is stored to
    //   (first column of its run) or added to (later columns: further edges of the same type)
    // NNConv, stage 2 (waves 0 .. 5): runs w, w + 6, w + 12 of the tile: type (weight image) of each
    int n_ent = 0;                                                // waves 0, 1: entries of the longest of the wave's 8 rows
    int my_run_t[kRunsPerWave];
#pragma unroll
    for (int j = 0; j < kRunsPerWave; ++j) my_run_t[j] = -1;
    // this wave's slot among the waves that multiply: its runs are slot, slot + kRunWaves, ...
    const int run_slot = kRunWaves == 6 ? tw : (tw < 2 ? tw : (tw == 4 || tw == 5) ? tw - 2 : -1);
    {
        const int c0 = __builtin_amdgcn_readfirstlane(A.tile_col_ptr[tile]);
        const int c1 = __builtin_amdgcn_readfirstlane(A.tile_col_ptr[tile + 1]);
        const int nc = c1 - c0;
        int *s_src = reinterpret_cast<int *>(S);                  // [nc][16] source words of the tile's columns (S is zeroed below)
        int *s_meta = reinterpret_cast<int *>(nnred);             // [nc] meta words, then run | rank << 8
        int *s_cnt = s_meta + 2048;                               // [16] entries per row
        for (int i = tid; i < nc * 16; i += NT) s_src[i] = A.col_src[(int64_t)c0 * 16 + i];
        for (int i = tid; i < nc; i += NT) s_meta[i] = A.col_meta[c0 + i];
        __syncthreads();
        if (tid == 0) {                                           // runs of same-type columns, in column order
            int run = -1, first = 0;
            for (int k = 0; k < nc; ++k) {
                const int m = s_meta[k];
                if (m & kColFirst) {
                    ++run;
                    first = k;
                    run_type[run & 31] = m & 0xff;
                }
                s_meta[k] = run | ((k - first) << 8);
            }
            run_type[31] = run + 1;
        }
        __syncthreads();
        if (tid < 16) {
            // one thread per row: its entries in column order = (byte offset of the source row, LDS byte offset of the type-sum
            // slot S[run][row] | 1: add to the slot (a further edge of the same type) | 2: valid)
            int cnt = 0;
            float rd = 0.f;
            for (int k = 0; k < nc; ++k) {
                const int sv = s_src[k * 16 + tid];
                if (sv < 0) continue;
                const int rr = s_meta[k], run = rr & 0xff;
                const bool root = run_type[run & 31] == T;
                if (root) rd = __int_as_float(sv);                // the root column carries max(deg, 1) in the source slot
                const int64_t srow = root ? tile * 16 + tid : (int64_t)sv;
                if (cnt < kNnEntries) ent[tid * kNnEntries + cnt] = make_int2((int)(srow * 128), (int)((run * 16 + tid) * 128) | ((rr >> 8) ? 1 : 0) | 2);
                ++cnt;
            }
            for (int i = cnt; i < kNnEntries; ++i) ent[tid * kNnEntries + i] = make_int2(0, 0);
            rootdeg[tid] = rd;
            s_cnt[tid] = cnt < kNnEntries ? cnt : kNnEntries;
        }
        __syncthreads();
        if (tw < 2) {
            int m = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) m = max(m, s_cnt[8 * tw + r]);
            n_ent = __builtin_amdgcn_readfirstlane(m);
        }
        const int nruns = run_type[31];
        if (tw < kNnWaves && run_slot >= 0) {
#pragma unroll
            for (int j = 0; j < kRunsPerWave; ++j)
                my_run_t[j] = run_slot + kRunWaves * j < nruns ? __builtin_amdgcn_readfirstlane(run_type[(run_slot + kRunWaves * j) & 31]) : -1;
        }
        __syncthreads();
        for (int i = tid; i < (T + 1) * 512 / 4; i += NT) reinterpret_cast<float4 *>(S)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // collision waves (one per half tile; gather mapping: lane = (row slot r = lane >> 3, 16-byte piece p = lane & 7) of the
    // rows 8 h + r): the neighbour list of row r and the first kGinCached neighbours' gather offsets
    const int gh = tw - kNnWaves, gr = lane >> 3, gp = lane & 7;
    const int64_t g_row = tile * 16 + 8 * gh + gr;
    int gbeg = 0, gend = 0;
    if (tw >= kNnWaves && g_row < n) {
        gbeg = A.col_rowptr[g_row];
        gend = A.col_rowptr[g_row + 1];
        if (gp == 0)
            for (int k = 0; k < kGinCached; ++k) gnb[(8 * gh + gr) * kGinCached + k] = gbeg + k < gend ? (uint32_t)A.col_nbr[gbeg + k] * 128u : kOob;
    } else if (tw >= kNnWaves && gp == 0) {
        for (int k = 0; k < kGinCached; ++k) gnb[(8 * gh + gr) * kGinCached + k] = kOob;
    }
    TGNN_SMALL_COMMIT(0)
    if (A.weights_done) {
        // the NNConv operand images come from a kernel on another stream that may still be running (the init MLP above did not
        // need them): wait for its last block, then drop what this CU / XCD caches of other XCDs' lines
        if (tid == 0) spin_until_ge(A.weights_done, A.weights_target, spin, kSpinErrWeights);
        __syncthreads();
        if (tw == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();

    for (int layer = 0; layer < D; ++layer) {
        const float *sp = spv + (layer & 1) * kSpGinW;
        TGNN_ST(0)
        // =========================================== phase A ===========================================
        if (tw < kNnWaves) {
            // ---- NNConv
            // Stage 1 (waves 0, 1): gather whole 128-byte source rows, 8 rows per instruction (8 lanes x 16 bytes each: a fraction
            //      of the address-path time of the 16-rows x 64-bytes pattern of the matrix layout), and store / add them to the
            //      type-sum tiles: S[run][row] = sum of the sources of the row's edges of that type, in edge order
            if (tw < 2) {
                TGNN_ST3_RESET
                const __amdgpu_buffer_rsrc_t h_rs = rsrc_of(A.mid + (size_t)layer * slot);
                char *lds_b = reinterpret_cast<char *>(S);
                const int2 *my_ent = ent + (8 * tw + (lane >> 3)) * kNnEntries;
                const uint32_t pp16 = (uint32_t)(lane & 7) * 16u;
                for (int b0 = 0; b0 < n_ent; b0 += 16) {          // (wave-uniform trip count: 1 up to 15 in-edges per row)
                    f32x4 x[16];
                    uint32_t dd[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int2 e = my_ent[b0 + i];
                        dd[i] = (uint32_t)e.y;
                        x[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rs, (e.y & 2) ? (uint32_t)e.x + pp16 : kOob, 0, kCpGather));
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (dd[i] & 2u) {
                            f32x4 *dst = reinterpret_cast<f32x4 *>(lds_b + (dd[i] & ~3u) + pp16);
                            if (dd[i] & 1u) *dst = *dst + x[i];
                            else *dst = x[i];
                        }
                    }
                }
                TGNN_ST3(1)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();                         // (all eight waves: the collision waves pass theirs after issuing their gathers)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // Stage 2: the weight fragments of this wave's runs straight from the image in global memory (used once per tile: no LDS
            //      staging).  Issued only now: ahead of the gathers, these 108 KB per tile held up the collision waves' gathers --
            //      the longer chain -- by ~2 us in the CU's memory pipeline; this chain has the slack.
            const bf16x8 *wimg_l = reinterpret_cast<const bf16x8 *>(A.wimg + (size_t)layer * (T + 1) * kWtType) + lane;
            constexpr int kPl = kWtPlane / 4, kTy = kWtType / 4;  // 16-byte fragments per plane / per type
            bf16x8 wf[kRunsPerWave][6];
#pragma unroll
            for (int j = 0; j < kRunsPerWave; ++j) {
                const bf16x8 *wp = wimg_l + (size_t)(my_run_t[j] >= 0 ? my_run_t[j] : 0) * kTy;
                wf[j][0] = wp[0]; wf[j][1] = wp[64]; wf[j][2] = wp[kPl]; wf[j][3] = wp[kPl + 64]; wf[j][4] = wp[2 * kPl]; wf[j][5] = wp[2 * kPl + 64];
            }
            //      D^T += W_t^T . S_t^T for this wave's runs (root run: operand pre-multiplied by max(deg, 1))
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;           // row fj, channels 4 fq + r and 16 + 4 fq + r
#pragma unroll
            for (int j = 0; j < kRunsPerWave; ++j) {
                if (my_run_t[j] >= 0) {                           // wave-uniform
                    const float *srow = S + ((run_slot + kRunWaves * j) * 16 + fj) * 32 + 8 * fq;
                    const float4 sa = *reinterpret_cast<const float4 *>(srow), sb = *reinterpret_cast<const float4 *>(srow + 4);
                    const float scale = my_run_t[j] == T ? rootdeg[fj] : 1.0f;
                    bf16x8 xh, xm, xl;
                    {
                        const float as[8] = {sa.x * scale, sa.y * scale, sa.z * scale, sa.w * scale, sb.x * scale, sb.y * scale, sb.z * scale, sb.w * scale};
                        split3_trunc(as, xh, xm, xl);
                    }
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][4], xh, d0, 0, 0, 0);   // lo . hi
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][5], xh, d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xl, d0, 0, 0, 0);   // hi . lo
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], xl, d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], xm, d0, 0, 0, 0);   // mid . mid
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][3], xm, d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], xh, d0, 0, 0, 0);   // mid . hi
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][3], xh, d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xm, d0, 0, 0, 0);   // hi . mid
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], xm, d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xh, d0, 0, 0, 0);   // hi . hi
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], xh, d1, 0, 0, 0);
                }
            }
            float *mine = nnred + (tw * 64 + lane) * 8;
            *reinterpret_cast<float4 *>(mine) = make_float4(d0[0], d0[1], d0[2], d0[3]);
            *reinterpret_cast<float4 *>(mine + 4) = make_float4(d1[0], d1[1], d1[2], d1[3]);
            TGNN_ST3(2)
        } else {
            // ---- CollConv of half a tile on ONE wave, no hand-over: gather whole 128-byte rows (8 lanes x 16 bytes per row: a
            //      sixteenth of the CU's address-path time per byte of the 16-rows x 64-bytes pattern of the matrix layout), the
            //      neighbourhood sum in registers, z through a wave-private LDS tile into the B-operand layout, then the
            //      32 -> 32 -> 64 -> 32 MLP of gin32_mlp_kernel on the half-filled 16-row tile.  Runs beside the NNConv waves.
            TGNN_ST2_RESET
            const float *src = layer == 0 ? A.mid : ((layer - 1) & 1 ? A.a2[1] : A.a2[0]);
            const __amdgpu_buffer_rsrc_t a_rs = rsrc_of(src);
            // (all kGinCached slots are loaded, used or not: with the loads of the unused ones behind a wave-uniform `k < longest row`
            //  branch hipcc drains the queue at every join -- 0.375 instead of 0.337 ms per forward)
            float4 xr[kGinCached];
            uint32_t noff[kGinCached];
#pragma unroll
            for (int k = 0; k < kGinCached; ++k) {
                const uint32_t o = gnb[(8 * gh + gr) * kGinCached + k];
                noff[k] = o == kOob ? kOob : o + (uint32_t)gp * 16u;
                xr[k] = ld_gather_f4(a_rs, noff[k]);
            }
            const float4 selfv = ld_gather_f4(a_rs, g_row < n ? (uint32_t)g_row * 128u + (uint32_t)gp * 16u : kOob);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();                         // (the NNConv waves' stage 1 -> stage 2 barrier: every wave of the block
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   //  passes one; here it costs nothing, the gathers are in flight)
            TGNN_ST2(5)
#ifdef TGNN_SMALL_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TGNN_ST2(6)
#endif
            const bool use_stat = layer > 0;
            // x = BatchNorm of the previous layer's pre-BN rows, folded into the sum as in gin32_aggregate_kernel
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 mhi = use_stat ? *reinterpret_cast<const float4 *>(st + 128 + 4 * gp) : zero4;
            const float4 mlo = use_stat ? *reinterpret_cast<const float4 *>(st + 160 + 4 * gp) : zero4;
            const float4 gv = use_stat ? *reinterpret_cast<const float4 *>(st + 192 + 4 * gp) : one4;
            const float4 bv = use_stat ? *reinterpret_cast<const float4 *>(st + 224 + 4 * gp) : zero4;
            float4 acc = zero4;
#pragma unroll
            for (int k = 0; k < kGinCached; ++k)
                if (noff[k] != kOob) {
                    acc.x += (xr[k].x - mhi.x) - mlo.x; acc.y += (xr[k].y - mhi.y) - mlo.y;
                    acc.z += (xr[k].z - mhi.z) - mlo.z; acc.w += (xr[k].w - mhi.w) - mlo.w;
                }
            for (int e = gbeg + kGinCached; __any(e < gend); e += 8) {   // longer neighbour lists: 8 rows in flight at a time
                float4 y[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    y[k] = ld_gather_f4(a_rs, e + k < gend ? (uint32_t)A.col_nbr[e + k] * 128u + (uint32_t)gp * 16u : kOob);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (e + k < gend) {
                        acc.x += (y[k].x - mhi.x) - mlo.x; acc.y += (y[k].y - mhi.y) - mlo.y;
                        acc.z += (y[k].z - mhi.z) - mlo.z; acc.w += (y[k].w - mhi.w) - mlo.w;
                    }
            }
            const float one_eps = sp[kSpEps];
            const float kb = one_eps + (float)(gend - gbeg);
            float4 z4;
            z4.x = fmaf(gv.x, fmaf(one_eps, (selfv.x - mhi.x) - mlo.x, acc.x), kb * bv.x);
            z4.y = fmaf(gv.y, fmaf(one_eps, (selfv.y - mhi.y) - mlo.y, acc.y), kb * bv.y);
            z4.z = fmaf(gv.z, fmaf(one_eps, (selfv.z - mhi.z) - mlo.z, acc.z), kb * bv.z);
            z4.w = fmaf(gv.w, fmaf(one_eps, (selfv.w - mhi.w) - mlo.w, acc.w), kb * bv.w);
            float *zs = ginred + gh * 256;                       // [8 rows][32]
            *reinterpret_cast<float4 *>(zs + gr * 32 + 4 * gp) = z4;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // matrix layout: lane (n = fj, q = fq) holds floats 8 q .. 8 q + 7 of tile row n; rows of the other half are zero
            const bool mine = (fj >> 3) == gh;
            float z[8];
            {
                const float4 za = *reinterpret_cast<const float4 *>(zs + (fj & 7) * 32 + 8 * fq);
                const float4 zb = *reinterpret_cast<const float4 *>(zs + (fj & 7) * 32 + 8 * fq + 4);
                z[0] = mine ? za.x : 0.f; z[1] = mine ? za.y : 0.f; z[2] = mine ? za.z : 0.f; z[3] = mine ? za.w : 0.f;
                z[4] = mine ? zb.x : 0.f; z[5] = mine ? zb.y : 0.f; z[6] = mine ? zb.z : 0.f; z[7] = mine ? zb.w : 0.f;
            }
            const bf16x8 *W1s = reinterpret_cast<const bf16x8 *>(gw), *W2s = W1s + 3 * 2 * 64, *W3s = W2s + 3 * 4 * 64;
            const float *Bs = sp + kSpGinB;
            auto bias4 = [&](int base, int mb) {
                const float4 t = *reinterpret_cast<const float4 *>(Bs + base + 16 * mb + 4 * fq);
                return f32x4{t.x, t.y, t.z, t.w};
            };
            const bf16x8 *w1p = W1s + fj * 4 + fq, *w2p = W2s + fj * 4 + fq, *w3p = W3s + fj * 4 + fq;
            bf16x8 xb[3];
            split3_trunc(z, xb[0], xb[1], xb[2]);
            TGNN_ST2(0)
            f32x4 h1a = small_mma6(w1p + 0 * 64, 2 * 64, xb, bias4(0, 0));
            f32x4 h1b = small_mma6(w1p + 1 * 64, 2 * 64, xb, bias4(0, 1));
            {
                const float x[8] = {sigmoidf_(h1a[0]), sigmoidf_(h1a[1]), sigmoidf_(h1a[2]), sigmoidf_(h1a[3]),
                                    sigmoidf_(h1b[0]), sigmoidf_(h1b[1]), sigmoidf_(h1b[2]), sigmoidf_(h1b[3])};
                split3_trunc(x, xb[0], xb[1], xb[2]);
            }
            TGNN_ST2(1)
            f32x4 h2[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) h2[mb] = small_mma6(w2p + mb * 64, 4 * 64, xb, bias4(32, mb));
            f32x4 o0 = bias4(96, 0), o1 = bias4(96, 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float x[8] = {sigmoidf_(h2[2 * ks][0]), sigmoidf_(h2[2 * ks][1]), sigmoidf_(h2[2 * ks][2]), sigmoidf_(h2[2 * ks][3]),
                                    sigmoidf_(h2[2 * ks + 1][0]), sigmoidf_(h2[2 * ks + 1][1]), sigmoidf_(h2[2 * ks + 1][2]), sigmoidf_(h2[2 * ks + 1][3])};
                split3_trunc(x, xb[0], xb[1], xb[2]);
                o0 = small_mma6(w3p + (0 * 2 + ks) * 64, 256, xb, o0);
                o1 = small_mma6(w3p + (1 * 2 + ks) * 64, 256, xb, o1);
            }
            TGNN_ST2(2)
            auto sig_out = [](float v) { return sigmoid_out_f32(v); };           // full accuracy, as gin32_mlp_kernel (LeakyReLU behind a sigmoid: the identity)
            float4 r0, r1;
            r0.x = sig_out(o0[0]); r0.y = sig_out(o0[1]); r0.z = sig_out(o0[2]); r0.w = sig_out(o0[3]);
            r1.x = sig_out(o1[0]); r1.y = sig_out(o1[1]); r1.z = sig_out(o1[2]); r1.w = sig_out(o1[3]);
            if (!row_ok) r0 = r1 = zero4;
            // row fj, channels 4 fq + r and 16 + 4 fq + r: to the LDS tile (merge, BatchNorm sums) and to HBM (next layer's gathers)
            if (mine) {
                *reinterpret_cast<float4 *>(a2s + fj * 32 + 4 * fq) = r0;
                *reinterpret_cast<float4 *>(a2s + fj * 32 + 16 + 4 * fq) = r1;
            }
            TGNN_ST2(3)
            const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(layer & 1 ? A.a2[1] : A.a2[0]);
            const uint32_t o_off = mine && row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 16u : kOob;
            st_sc1_f4(o_rs, o_off, r0);
            st_sc1_f4(o_rs, o_off == kOob ? kOob : o_off + 64u, r1);
            TGNN_ST2(4)
        }
        TGNN_ST(1)
        __syncthreads();
        // the next layer's weight images start their trip now -- behind this layer's gathers in the CU's memory pipeline (114 KB:
        // ahead of them they delayed every gather by ~1 us), in flight during the epilogue / MLP below, in registers until
        // this layer is through with the images in LDS
        if (layer + 1 < D) TGNN_SMALL_PREFETCH(layer + 1)
        TGNN_ST(2)
        constexpr int kEpiWave = kRunWaves == 4 ? 2 : kNnWaves - 2;
        if (tw == kEpiWave || tw == kEpiWave + 1) {
            // ---- NNConv epilogue, one wave per 16-channel half: the six partial products in fixed order, mean, bias, LeakyReLU
            const int half = tw - kEpiWave;                      // channels 16 half + 4 fq + r of row fj
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < kNnWaves; ++w) {
                const float4 a = *reinterpret_cast<const float4 *>(nnred + (w * 64 + lane) * 8 + 4 * half);
                if (w == 0) t4 = a;
                else t4 = make_float4(t4.x + a.x, t4.y + a.y, t4.z + a.z, t4.w + a.w);
            }
            const float rd = rootdeg[fj];
            const bool valid = rd > 0.f;
            const float inv = valid ? 1.0f / rd : 0.f;
            const float4 bias = *reinterpret_cast<const float4 *>(sp + kSpBias + 16 * half + 4 * fq);
            float4 o;
            o.x = leakyf_(fmaf(t4.x, inv, bias.x)); o.y = leakyf_(fmaf(t4.y, inv, bias.y));
            o.z = leakyf_(fmaf(t4.z, inv, bias.z)); o.w = leakyf_(fmaf(t4.w, inv, bias.w));
            if (!valid) o = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(a1s + fj * 32 + 16 * half + 4 * fq) = o;
        }
        __syncthreads();
        if (tid < 128) {
            // ---- BatchNorm column sums of the two tiles over their 16 rows (rows >= n hold zeros) = this block's partial row
            //      [bn1 sum | bn1 sumsq | bn2 sum | bn2 sumsq]: one column per lane, rows in order
            const float *tile_s = tid < 64 ? a1s : a2s;
            const int ch = tid & 31;
            const bool sq = (tid & 32) != 0;
            double acc2 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double v = (double)tile_s[r * 32 + ch];
                acc2 += sq ? v * v : v;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, acc2), part_rs, ((uint32_t)blockIdx.x * 128u + (uint32_t)tid) * 8u, 0,
                                                  kCpSc1);
        }
        TGNN_ST(3)
        small_grid_barrier(A.ctr, target, nblk, spin);
        TGNN_ST(4)
        // =========================================== phase B ===========================================
        {
            // every block folds all partial rows in the same order: thread = (column pair, row group), its rows in flight at once
            // (while they fly, the next layer's weight images go from the prefetch registers to LDS)
            const int jp = tid & 63, g = tid >> 6;
            double acc0 = 0.0, acc1 = 0.0;
            u32x4 v[16];
            auto fetch = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int p = p0 + 8 * u;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, p < (int)nblk ? ((uint32_t)p * 128u + 2u * (uint32_t)jp) * 8u : kOob, 0, kCpSc1);
                }
            };
            auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc0 += __builtin_bit_cast(double, u32x2{v[u][0], v[u][1]});     // (rows past the end were read as +0.0)
                    acc1 += __builtin_bit_cast(double, u32x2{v[u][2], v[u][3]});
                }
            };
            TGNN_ST3_RESET
            fetch(g);
            TGNN_ST3(4)
            if (layer + 1 < D) TGNN_SMALL_COMMIT(layer + 1)
            TGNN_ST3(5)
            fold();
            TGNN_ST3(6)
            if (nblk > 128) {                                     // (at most 256 blocks)
                fetch(g + 128);
                fold();
            }
            red[g * 128 + 2 * jp] = acc0;
            red[g * 128 + 2 * jp + 1] = acc1;
            __syncthreads();
            TGNN_ST3(7)
            if (tid < 64) {                                       // the two records, as bn_finalize_kernel writes them
                const int job = tid >> 5, ch = tid & 31;
                double t_sum = 0.0, t_sq = 0.0;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    t_sum += red[gg * 128 + job * 64 + ch];
                    t_sq += red[gg * 128 + job * 64 + 32 + ch];
                }
                const double inv_n = 1.0 / (double)n;
                const double mean = t_sum * inv_n;
                double var = t_sq * inv_n - mean * mean;
                if (var < 0.0) var = 0.0;
                const float gamma = sp[(job ? kSpG2 : kSpG1) + ch], beta = sp[(job ? kSpB2 : kSpB1) + ch];
                const float mh = (float)mean;
                float *rec = st + job * 128;
                rec[ch] = mh;
                rec[32 + ch] = (float)(mean - (double)mh);
                rec[64 + ch] = (float)((double)gamma / sqrt(var + (double)A.eps));
                rec[96 + ch] = beta;
                if (blockIdx.x == 0 && A.update_running) {
                    // the running buffers are updated after the last layer (their read-modify-write round trips would make
                    // block 0 the straggler of every barrier): park the batch statistics
                    double *rs = A.runstat + (size_t)layer * 128 + job * 64;
                    rs[ch] = mean;
                    rs[32 + ch] = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
                }
            }
            __syncthreads();
        }
        TGNN_ST(5)
        if (tid < 128) {
            // ---- merge (TilinGNN.py:64-71): slot layer + 1 = BN1(a1) * BN2(a2) (+ slot layer - 2), own rows
            const int row = tid >> 3, c4 = (tid & 7) * 4;
            const int64_t r = tile * 16 + row;
            if (r < n) {
                const float4 x1 = *reinterpret_cast<const float4 *>(a1s + row * 32 + c4);
                const float4 x2 = *reinterpret_cast<const float4 *>(a2s + row * 32 + c4);
                const float4 m1h = *reinterpret_cast<const float4 *>(st + c4), m1l = *reinterpret_cast<const float4 *>(st + 32 + c4);
                const float4 g1 = *reinterpret_cast<const float4 *>(st + 64 + c4), b1 = *reinterpret_cast<const float4 *>(st + 96 + c4);
                const float4 m2h = *reinterpret_cast<const float4 *>(st + 128 + c4), m2l = *reinterpret_cast<const float4 *>(st + 160 + c4);
                const float4 g2 = *reinterpret_cast<const float4 *>(st + 192 + c4), b2 = *reinterpret_cast<const float4 *>(st + 224 + c4);
                float4 o;
                o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x);
                o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y);
                o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z);
                o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w);
                const uint32_t off = (uint32_t)r * 128u + (uint32_t)c4 * 4u;
                if (layer >= 2) {
                    const float4 rs = ld_sc1_f4(rsrc_of(A.mid + (size_t)(layer - 2) * slot), off);
                    o.x += rs.x; o.y += rs.y; o.z += rs.z; o.w += rs.w;
                }
                st_sc1_f4(rsrc_of(A.mid + (size_t)(layer + 1) * slot), off, o);
            }
        }
        TGNN_ST(6)
        if (layer + 1 < D) {
            small_grid_barrier(A.ctr, target, nblk, spin);
#ifdef TGNN_SMALL_CACHED
            if (tw == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop what this CU's L1 / this XCD's L2 hold of other XCDs' rows
            __syncthreads();
#endif
        }
        TGNN_ST(7)
    }

    if (blockIdx.x == 0 && A.update_running && __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        // running statistics of the 2 x depth BatchNorms of the layers (momentum update, num_batches_tracked); not after a
        // wait that gave up: the parked statistics are garbage then, and the host repeats the forward
        __syncthreads();
        for (int idx = tid; idx < D * 64; idx += NT) {
            const int l = idx >> 6, job = (idx >> 5) & 1, ch = idx & 31;
            const SmallRun run = R.l[l];
            float *rm = job ? run.rm2 : run.rm1, *rv = job ? run.rv2 : run.rv1;
            const double *rs = A.runstat + (size_t)l * 128 + job * 64;
            rm[ch] = (float)((1.0 - (double)A.momentum) * (double)rm[ch] + (double)A.momentum * rs[ch]);
            rv[ch] = (float)((1.0 - (double)A.momentum) * (double)rv[ch] + (double)A.momentum * rs[32 + ch]);
            if (ch == 0) *(job ? run.nbt2 : run.nbt1) += 1;
        }
    }

    // ====================================== final MLP (TilinGNN.py:74-76) ======================================
    // over the concatenation of the depth + 1 slots (K step ks = slot ks), own rows; 4 x (Linear + LeakyReLU + BatchNorm), then
    // Linear(32, out_dim) + sigmoid
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own rows only: the last merge's stores are through, and
        __syncthreads();                                             // the LDS of the layer loop is free
        float *dx = lds, *dact = lds + small_dense_ksteps(D) * kDPlaneStep;
        double *dred = reinterpret_cast<double *>(dact + 16 * kDActLd);
        float *drec = dact + 16 * kDActLd + 2048;
        const int valid_rows = n - tile * 16 < 16 ? (int)(n - tile * 16) : 16;
        for (int it = tid; it < (D + 1) * 64; it += NT) {        // B-operand planes of the own rows of every slot
            const int ln = it & 63, ks = it >> 6, nn = ln & 15, q = ln >> 4;
            const uint32_t off = nn < valid_rows ? (uint32_t)(tile * 16 + nn) * 128u + (uint32_t)q * 32u : kOob;
            const __amdgpu_buffer_rsrc_t s_rs = rsrc_of(A.mid + (size_t)ks * slot);
            const float4 a = ld_sc1_f4(s_rs, off), b = ld_sc1_f4(s_rs, off == kOob ? kOob : off + 16u);
            const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            bf16x8 *dst = reinterpret_cast<bf16x8 *>(dx) + (size_t)ks * 3 * 64 + ln;
            split3_trunc(xv, dst[0], dst[64], dst[128]);
        }
        __syncthreads();
        small_tile_dense<2>(E.f[0].img, D + 1, 16, E.f[0].bias, dx, dact, tw, lane);
        __syncthreads();
        small_tile_bn<256>(dact, valid_rows, E.f[0], E.part_wide, dred, drec, n, A.eps, A.momentum, A.update_running, A.ctr, target, nblk, spin, tid);
        small_tile_planes_from_act(dact, drec, 256, valid_rows, dx, tid);
        __syncthreads();
        small_tile_dense<1>(E.f[1].img, 8, 8, E.f[1].bias, dx, dact, tw, lane);
        __syncthreads();
        small_tile_bn<128>(dact, valid_rows, E.f[1], E.part_wide + kPartWideSet, dred, drec, n, A.eps, A.momentum, A.update_running, A.ctr, target, nblk, spin, tid);
        small_tile_planes_from_act(dact, drec, 128, valid_rows, dx, tid);
        __syncthreads();
        small_tile_dense<1>(E.f[2].img, 4, 4, E.f[2].bias, dx, dact, tw, lane);
        __syncthreads();
        small_tile_bn<64>(dact, valid_rows, E.f[2], E.part_wide, dred, drec, n, A.eps, A.momentum, A.update_running, A.ctr, target, nblk, spin, tid);
        small_tile_planes_from_act(dact, drec, 64, valid_rows, dx, tid);
        __syncthreads();
        small_tile_dense<1>(E.f[3].img, 2, 2, E.f[3].bias, dx, dact, tw, lane);
        __syncthreads();
        small_tile_bn<32>(dact, valid_rows, E.f[3], E.part_wide + kPartWideSet, dred, drec, n, A.eps, A.momentum, A.update_running, A.ctr, target, nblk, spin, tid);
        for (int it = tid; it < 16 * E.out_dim; it += NT) {      // final_mlp.1
            const int r = it / E.out_dim, o = it - r * E.out_dim;
            if (r < valid_rows) {
                float acc = E.b_last[o];
                for (int k = 0; k < 32; ++k)
                    acc = fmaf(bn_apply1(dact[r * kDActLd + k], drec[k], drec[256 + k], drec[512 + k], drec[768 + k]), E.w_last[o * 32 + k], acc);
                E.probs[(tile * 16 + r) * E.out_dim + o] = sigmoidf_(acc);
            }
        }
    }
#ifdef TGNN_SMALL_TIMING
    if (tid == 0 && blockIdx.x < 260)
        for (int k = 0; k < 8; ++k) g_small_timing[blockIdx.x * 32 + k] = tacc[k];
    if (tid == kNnWaves * 64 && blockIdx.x < 260)
        for (int k = 0; k < 8; ++k) g_small_timing[blockIdx.x * 32 + 8 + k] = tacc2[k];
    if (tid == 0 && blockIdx.x < 260)
        for (int k = 0; k < 8; ++k) g_small_timing[blockIdx.x * 32 + 16 + k] = tacc3[k];
#endif
}

// ---- the device's spin-error word (forward_persist.h) ------------------------------------------------------------------
// 4 bytes of device memory per device, allocated on first use, zero; the spin kernels OR a reason into it when a wait runs out
// of its budget, tgnn_spin_error_poll reads and clears it.
static std::atomic<unsigned long long> g_spin_budget{kSpinBudgetTicksDefault};
unsigned long long spin_budget_ticks() { return g_spin_budget.load(std::memory_order_relaxed); }
static std::atomic<int> g_spin_fault{0};
int spin_take_fault() {
    int v = g_spin_fault.load(std::memory_order_relaxed);
    while (v > 0 && !g_spin_fault.compare_exchange_weak(v, v - 1)) {}
    return v > 0 ? 1 : 0;
}
unsigned *spin_error_word() {
    static std::mutex mu;
    static unsigned *word[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!word[dev]) {
        unsigned *p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        word[dev] = p;
    }
    return word[dev];
}

// The mirror: 64 bytes of host memory mapped into the device, one per device.  Only a thread that gives up writes it (a
// system-scope OR over the host link: rare by construction); the waiters poll the DEVICE word.
struct SpinMirror {
    unsigned *host = nullptr, *dev = nullptr;
};
static SpinMirror *spin_mirror_of_current_device() {
    static std::mutex mu;
    static SpinMirror m[64];
    static bool tried[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried[dev]) {
        tried[dev] = true;
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            for (int i = 0; i < 16; ++i) static_cast<volatile unsigned *>(h)[i] = 0u;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                m[dev].host = static_cast<unsigned *>(h);
                m[dev].dev = static_cast<unsigned *>(d);
            } else {
                (void)hipHostFree(h);
            }
        }
    }
    return m[dev].host ? &m[dev] : nullptr;
}
unsigned *spin_error_mirror() {
    SpinMirror *m = spin_mirror_of_current_device();
    return m ? m->dev : nullptr;
}
unsigned spin_error_pending() {
    SpinMirror *m = spin_mirror_of_current_device();
    return m ? __atomic_load_n(m->host, __ATOMIC_RELAXED) : 0u;
}
static void spin_error_clear(hipStream_t s) {                // (stream-ordered for the device word; the mirror at once)
    if (unsigned *w = spin_error_word()) (void)hipMemsetAsync(w, 0, 4, s);
    if (SpinMirror *m = spin_mirror_of_current_device()) __atomic_store_n(m->host, 0u, __ATOMIC_RELAXED);
}
static std::atomic<int64_t> g_persist_off{0};                // forwards left in the fallback window
void persist_fallback(int64_t n_forwards) { g_persist_off.store(n_forwards > 0 ? n_forwards : 0, std::memory_order_relaxed); }
bool persist_allowed() {
    int64_t v = g_persist_off.load(std::memory_order_relaxed);
    while (v > 0 && !g_persist_off.compare_exchange_weak(v, v - 1)) {}
    return v <= 0;
}
constexpr int64_t kPersistFallbackForwards = 256;            // how long a starved persistent kernel keeps its kind off
int spin_error_collect_stale(hipStream_t s) {
    const unsigned code = spin_error_pending();
    if (!code) return TGNN_OK;
    spin_error_clear(s);
    persist_fallback(kPersistFallbackForwards);
    set_error("a persistent forward kernel of an EARLIER call gave up waiting for its other blocks (reason bits %u: 1 grid barrier, 2 "
              "partial rows, 4 edge weights) and nobody collected the failure: the results of that forward -- and of any persistent "
              "forward queued behind it -- are invalid.  The word is cleared, the persistent schedules are off for the next %lld "
              "forwards; repeat the call (TilinGNN.forward_checked does all this by itself)", code, (long long)kPersistFallbackForwards);
    return TGNN_ERR_STALE_RESULT;
}

static size_t small_lds_bytes(int n_types, int depth) {
    return ((size_t)small_s_offset(depth) + (size_t)(n_types + 1) * 512) * sizeof(float);
}
constexpr size_t kSmallMaxLds = 160 * 1024 - 256;

static std::atomic<int64_t> g_small_limit{4096};

// 1 = eligible: one 16-row tile per block and at most one block per CU, the weight images of a layer fit LDS and the
// prefetch registers, the final MLP's input planes fit LDS
int small_layout_teams(const tgnn_model_dims *d, int64_t n_nodes, int n_types, int max_in_degree) {
    const int64_t limit = g_small_limit.load(std::memory_order_relaxed);
    if (n_nodes < 2 || n_nodes > limit || n_nodes > 4096) return 0;
    if (!persist_allowed()) return 0;                        // (a starved kernel of this kind a few forwards ago: general schedule for now)
    if (max_in_degree < 1 || max_in_degree + 1 > kNnEntries) return 0;   // a row's gather list (edges + the root row) in registers
    if (d->network_width != 32 || d->network_depth < 1 || d->network_depth > kSmallMaxDepth || d->output_dim > 256 ||
        d->node_features_dim > 256)
        return 0;
    if (small_lds_bytes(n_types, d->network_depth) > kSmallMaxLds) return 0;
    if (n_types + 1 > kRunWaves * kRunsPerWave || n_types + 1 > 31) return 0;      // runs per multiplying wave; run_type[32]
    // The grid barrier needs every block resident at the same time.  The kernel is launched as an ordinary kernel on the
    // caller's stream (a cooperative launch goes through a queue of its own: ~25 us of cross-queue dependency before and
    // after the kernel, measured), so the guarantee a cooperative launch gives is checked here instead: blocks <= CUs of
    // the device (of the partition, in CPX / NPS modes) x resident blocks per CU for this kernel's registers and LDS.
    static std::atomic<int> capacity[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int cap = capacity[dev].load(std::memory_order_acquire);
    if (cap == 0) {
        static LdsOptIn site;
        int per_cu = 0, cus = 0;
        if (opt_in_dynamic_lds(forward_layers_small_kernel, (int)kSmallMaxLds, site) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, forward_layers_small_kernel, kSmallThreads, kSmallMaxLds) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        cap = per_cu > 0 && cus > 0 ? (per_cu > 1 ? 1 : per_cu) * cus : -1;   // (counted as one block per CU: the LDS images fill it)
        capacity[dev].store(cap, std::memory_order_release);
    }
    // 2: at least 16 CUs stay free -- the pre-pass may then still be running when the kernel starts (it waits on a flag)
    return (n_nodes + 15) / 16 <= cap - 16 ? 2 : (n_nodes + 15) / 16 <= cap ? 1 : 0;
}

// workspace of the path, floats: per-layer packs | dense images (init 1, final 0..3)
static size_t small_image_floats(int depth, size_t (&off)[5]) {
    const size_t sz[5] = {32 * 32 * 3 / 2, (size_t)256 * 32 * (depth + 1) * 3 / 2, 128 * 256 * 3 / 2, 64 * 128 * 3 / 2, 32 * 64 * 3 / 2};
    size_t at = 0;
    for (int k = 0; k < 5; ++k) {
        off[k] = at;
        at += sz[k];
    }
    return at;
}
// image k (0: init Linear 1; 1 .. 4: the final MLP's Linears) inside a pack built with dense_images
const float *small_dense_image(const float *pack, int depth, int k) {
    size_t off[5];
    small_image_floats(depth, off);
    return pack + (size_t)depth * kSpStride + off[k];
}
size_t small_pack_floats(int depth) {
    size_t off[5];
    return (size_t)depth * kSpStride + small_image_floats(depth, off);
}

// Per-forward pre-pass (side stream): parameter packs + GIN images of the layers, MFMA images of the dense layers; re-arms
// the barrier counter
void launch_small_pack(const Params &P, int depth, float *pack, unsigned *barrier_ctr, hipStream_t s, bool dense_images, void *zero,
                       size_t zero_bytes, const unsigned *fin0_f16_max, hipStream_t final_images_stream) {
    size_t off[5];
    small_image_floats(depth, off);
    float *img = pack + (size_t)depth * kSpStride;
    SmallImageJobs J{};
    J.j[0] = SmallImageJob{P.f(P.init(1)), img + off[0], 32, 32, nullptr};
    J.j[1] = SmallImageJob{P.f(P.fin(0)), img + off[1], 256, 32 * (depth + 1), fin0_f16_max};
    J.j[2] = SmallImageJob{P.f(P.fin(1)), img + off[2], 128, 256, nullptr};
    J.j[3] = SmallImageJob{P.f(P.fin(2)), img + off[3], 64, 128, nullptr};
    J.j[4] = SmallImageJob{P.f(P.fin(3)), img + off[4], 32, 64, nullptr};
    for (int lo = 0; lo < depth; lo += kSmallPackChunk) {
        const int nl = depth - lo < kSmallPackChunk ? depth - lo : kSmallPackChunk;
        SmallPackLayers L{};
        for (int k = 0; k < nl; ++k) {
            const int b = P.layer(lo + k);
            L.l[k] = SmallPackLayer{P.f(b + 7), P.f(b + 8), P.f(b + 9), P.f(b + 20), P.f(b + 21), P.f(b + 13),
                                    P.f(b + 14), P.f(b + 15), P.f(b + 16), P.f(b + 17), P.f(b + 18), P.f(b + 19)};
        }
        const bool first = lo == 0 && dense_images;             // the dense images ride along with the first chunk
        // (final_images_stream: only the init Linear's there -- the final MLP's four, needed a whole layer loop later, go to a
        //  launch of their own on that stream: 64 x 4 blocks less in front of the persistent kernel)
        const int n_img = first ? (final_images_stream ? 1 : 5) : 0;
        small_pack_kernel<<<dim3(n_img > 1 ? 64 : 1, nl + n_img), 256, 0, s>>>(L, nl, pack + (size_t)lo * kSpStride,
                                                                               lo == 0 ? barrier_ctr : nullptr, J,
                                                                               lo == 0 ? static_cast<u32x4 *>(zero) : nullptr, (int64_t)(zero_bytes / 16));
    }
    if (dense_images && final_images_stream) {
        SmallImageJobs J4{};
        for (int k = 0; k < 4; ++k) J4.j[k] = J.j[1 + k];
        small_pack_kernel<<<dim3(64, 4), 256, 0, final_images_stream>>>(SmallPackLayers{}, 0, pack, nullptr, J4, nullptr, 0);
    }
}

#define TGNN_TRY_SMALL(expr) do { const int rc__ = (expr); if (rc__ != TGNN_OK) return rc__; } while (0)

// Spin-barrier kernels need all their blocks resident at once; two of them started side by side could each hold part of the CUs
// and wait for the rest for ever.  Every such launch passes through this per-device gate with the number of CUs it occupies:
//   * it joins the kernels still in flight when all of them TOGETHER fit the device (`capacity` CUs: small layouts of
//     different streams then run side by side -- three labyrinth-sized forwards, 79 CUs each, take the time of one);
//   * else it waits for every one of them (events, whichever streams they are on) and runs behind them.
// Invariant: the kernels that can be started-and-unfinished at any moment fit the device together -- the last one launched
// among them either waited for all the others or was admitted because the whole list + itself fits.
int spin_kernel_chain(hipStream_t s, void (*launch)(void *ctx, hipStream_t s), void *ctx, int cus_needed) {
    struct InFlight { hipEvent_t ev; int cus; };
    struct Gate { std::vector<InFlight> live; std::vector<hipEvent_t> pool; };
    static std::mutex mu;
    static Gate gates[64];
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
    const int capacity = device_cus() - 16;                   // (16 CUs stay free for the pre-pass kernels the forwards wait on)
    std::lock_guard<std::mutex> lock(mu);
    Gate &g = gates[dev];
    int used = 0;
    for (size_t i = 0; i < g.live.size();) {                   // forget what has finished
        if (hipEventQuery(g.live[i].ev) == hipSuccess) {
            g.pool.push_back(g.live[i].ev);
            g.live.erase(g.live.begin() + (long)i);
        } else {
            used += g.live[i].cus;
            ++i;
        }
    }
    (void)hipGetLastError();                                   // (hipEventQuery leaves hipErrorNotReady behind)
    if (used + cus_needed > capacity || g.live.size() >= 16)
        for (const InFlight &f : g.live) TGNN_CHECK_HIP(hipStreamWaitEvent(s, f.ev, 0));
    hipEvent_t ev;
    if (!g.pool.empty()) {
        ev = g.pool.back();
        g.pool.pop_back();
    } else {
        TGNN_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    launch(ctx, s);
    TGNN_CHECK_HIP(hipEventRecord(ev, s));
    g.live.push_back(InFlight{ev, cus_needed});
    return TGNN_OK;
}

// The whole forward behind the pre-pass: x -> probs (stream order)
int launch_forward_small(const tgnn_model_dims *d, const Params &P, const float *x, float *probs, float *mid, float *a2_0,
                         float *a2_1, const float *wimg, float *pack, const tgnn_graph *graph, double *part, double *part_wide,
                         double *runstat, unsigned *ctr, const unsigned *weights_done, unsigned weights_target, int64_t n,
                         int update_running, float eps, float momentum, hipStream_t s) {
    const int depth = d->network_depth;
    SmallArgs A{};
    A.mid = mid;
    A.a2[0] = a2_0;
    A.a2[1] = a2_1;
    A.wimg = wimg;
    A.pack = pack;
    A.tile_col_ptr = graph->nn_tile_col_ptr;
    A.col_meta = graph->nn_col_meta;
    A.col_src = graph->nn_col_src;
    A.col_rowptr = graph->col_rowptr;
    A.col_nbr = graph->col_src;
    A.part = part;
    A.runstat = runstat;
    A.ctr = ctr;
    A.weights_done = weights_done;
    A.weights_target = weights_target;
    A.n = n;
    A.n_types = graph->n_types;
    A.depth = depth;
    A.update_running = update_running;
    A.eps = eps;
    A.momentum = momentum;
    A.err = spin_error_word();
    A.err_host = spin_error_mirror();
    A.spin_budget = spin_budget_ticks();
    A.fault = spin_take_fault();
    if (!A.err) {
        set_error("tgnn_forward: the spin-error word of the device could not be allocated");
        return TGNN_ERR_LAUNCH;
    }
    SmallRunTab R{};
    for (int i = 0; i < depth; ++i) {
        const BnPtrs b1 = P.bn(P.layer(i) + 8), b2 = P.bn(P.layer(i) + 20);
        R.l[i] = SmallRun{b1.rm, b1.rv, b1.nbt, b2.rm, b2.rv, b2.nbt};
    }
    size_t off[5];
    small_image_floats(depth, off);
    const float *img = pack + (size_t)depth * kSpStride;
    auto dense = [&](int pi, const float *image) {
        const BnPtrs b = P.bn(pi + 2);
        return SmallDense{image, P.f(pi + 1), b.gamma, b.beta, b.rm, b.rv, b.nbt};
    };
    SmallEnds E{};
    E.x = x;
    E.w0 = P.f(P.init(0));
    E.i0 = dense(P.init(0), nullptr);
    E.i1 = dense(P.init(1), img + off[0]);
    for (int l = 0; l < 4; ++l) E.f[l] = dense(P.fin(l), img + off[1 + l]);
    E.w_last = P.f(P.last());
    E.b_last = P.f(P.last() + 1);
    E.probs = probs;
    E.part_wide = part_wide;
    E.fx = d->node_features_dim;
    E.out_dim = d->output_dim;
    const int blocks = (int)((n + 15) / 16);
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(forward_layers_small_kernel, (int)kSmallMaxLds, site));
    // One spin-barrier kernel at a time per device (spin_kernel_chain; the one-launch graph preparation shares the chain).
    // Other PROCESSES on the same GPU are outside this: they delay the kernel (blocks wait for a CU) but do not depend on it.
    struct Ctx { SmallArgs *A; SmallRunTab *R; SmallEnds *E; int blocks; size_t lds; } ctx{&A, &R, &E, blocks, small_lds_bytes(graph->n_types, depth)};
    TGNN_TRY_SMALL(spin_kernel_chain(s, [](void *c, hipStream_t st) {
        Ctx *x = static_cast<Ctx *>(c);
        forward_layers_small_kernel<<<dim3(x->blocks), dim3(kSmallThreads), x->lds, st>>>(*x->A, *x->R, *x->E);
    }, &ctx, blocks));                                         // (one block per CU: the LDS images fill it)
    return TGNN_OK;
}

}  // namespace tgnn

#ifdef TGNN_SMALL_TIMING
extern "C" int tgnn_debug_small_timing(unsigned long long *out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnn::g_small_timing), (size_t)n_blocks * 32 * sizeof(unsigned long long));
}
#endif
extern "C" int tgnn_spin_error_poll(tgnn_stream_t stream, uint32_t *code_out) {
    tgnn::DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(code_out, "null pointer");
    unsigned *w = tgnn::spin_error_word();
    if (!w) {
        tgnn::set_error("tgnn_spin_error_poll: no error word on this device");
        return TGNN_ERR_LAUNCH;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned host = 0;
    TGNN_CHECK_HIP(hipMemcpyAsync(&host, w, 4, hipMemcpyDeviceToHost, s));
    TGNN_CHECK_HIP(hipStreamSynchronize(s));
    host |= tgnn::spin_error_pending();
    if (host) {
        tgnn::spin_error_clear(s);
        tgnn::persist_fallback(tgnn::kPersistFallbackForwards);   // both persistent schedules off for a while, then re-armed
    }
    *code_out = host;
    if (host)
        tgnn::set_error("a persistent forward kernel gave up waiting for its other blocks (reason bits %u: 1 grid barrier, 2 partial "
                        "rows, 4 edge weights): another process or tenant holds compute units; the results of that forward are "
                        "invalid -- repeat it: the persistent schedules are off for the next %lld forwards", host,
                        (long long)tgnn::kPersistFallbackForwards);
    return TGNN_OK;
}
extern "C" void tgnn_persist_fallback(int64_t n_forwards) { tgnn::persist_fallback(n_forwards); }
extern "C" uint32_t tgnn_spin_error_peek(void) { return tgnn::spin_error_pending(); }
#ifdef TGNN_DEBUG
extern "C" void tgnn_debug_spin_fault(int32_t n_launches) { tgnn::g_spin_fault.store(n_launches > 0 ? n_launches : 0); }
#endif
extern "C" uint64_t tgnn_set_spin_budget_us(uint64_t us) {
    const unsigned long long ticks = us * 100ull;             // wall_clock64: 100 MHz
    return tgnn::g_spin_budget.exchange(ticks ? ticks : tgnn::kSpinBudgetTicksDefault) / 100ull;
}
extern "C" void tgnn_set_small_layout_limit(int64_t n_nodes) { tgnn::g_small_limit.store(n_nodes < 0 ? 0 : n_nodes); }
extern "C" int64_t tgnn_get_small_layout_limit(void) { return tgnn::g_small_limit.load(); }
