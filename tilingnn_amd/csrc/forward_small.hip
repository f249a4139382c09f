// The 20 message-passing layers of TilinGNN.forward (/root/reference/graph_networks/networks/TilinGNN.py:59-71) for SMALL
// layouts as ONE persistent kernel with grid-wide barriers.
//
// Why: the layouts the greedy solver actually scores (1 254 nodes for the labyrinth example, shrinking every round) have
// ~80 16-row tiles -- fewer than the chip has CUs.  The general schedule (forward.hip) spends such a forward in ~130
// dependent kernel launches at ~5 us of GPU-side dispatch latency each (measured: hipGraph replay does not help, the
// gap is between dependent dispatches, not in the host's launch path).  Here one block owns TEAMS tiles for all layers:
//     phase A   NNConv of the tile on three waves (column chunks, partial products summed through LDS) while the fourth
//               wave gathers the tile's collision neighbourhoods and runs the GIN MLP; BatchNorm column sums of the
//               block -> one partial row
//     barrier   (all blocks)
//     phase B   every block folds all partial rows in the same fixed order -> both BatchNorm records; merge of its own
//               rows (BN1(a1) * BN2(a2) + residual) -> the next slot of the skip buffer
//     barrier
// so a layer costs two barriers (~1.5 us each at 80 blocks) instead of five dependent launches.
//
// Cross-block data (skip-buffer rows, the collision branch's pre-BN rows, the partial rows) is written and read with sc1
// (agent-scope) buffer instructions: each XCD has its own L2, and the release/acquire fences that would make ordinary
// accesses visible across XCDs (buffer_wbl2 / buffer_inv) serialise per XCD -- 10.8 us per barrier at 256 blocks against
// 3.8 us without them (scratch/ubench/gridsync.hip).  The barrier itself is a monotonic counter.
//
// Arithmetic: the same formulas as the general path's kernels (nnconv_cols.hip, gin.hip, bn_merge.hip); what differs is the
// association of two sums (NNConv: three partial products per tile; BatchNorm: one partial row per block), i.e. fp32 / fp64
// rounding only.  Deterministic: every order is fixed.
#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

// ---- per-layer parameter pack (floats): small vectors, then the GIN MLP's MFMA weight image -----------------------------
constexpr int kSpBias = 0, kSpG1 = 32, kSpB1 = 64, kSpG2 = 96, kSpB2 = 128, kSpEps = 160, kSpGinB = 192, kSpGinW = 320;
constexpr int kSpGinFrags = 3 * 2 * 64 + 3 * 4 * 64 + 3 * 2 * 2 * 64;      // 1920 fragments of 16 bytes
constexpr int kSpStride = kSpGinW + kSpGinFrags * 4;                       // 8000 floats per layer

struct SmallPackLayer {
    const float *nn_bias, *g1, *b1, *g2, *b2, *eps, *w1, *gb1, *w2, *gb2, *w3, *gb3;
};
constexpr int kSmallPackChunk = 32;
struct SmallPackLayers {
    SmallPackLayer l[kSmallPackChunk];
};

__device__ __forceinline__ int small_kf(int q, int e) { return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4); }   // gin.hip: gin_kf

// one block per layer; block 0 of the first chunk re-arms the barrier counter of the persistent kernel
__global__ __launch_bounds__(256) void small_pack_kernel(SmallPackLayers layers, float *__restrict__ pack,
                                                         unsigned *__restrict__ barrier_ctr) {
    const SmallPackLayer L = layers.l[blockIdx.x];
    float *out = pack + (size_t)blockIdx.x * kSpStride;
    const int tid = threadIdx.x;
    if (barrier_ctr && blockIdx.x == 0 && tid == 0) *barrier_ctr = 0u;
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
struct SmallRun {
    float *rm1, *rv1;
    int64_t *nbt1;
    float *rm2, *rv2;
    int64_t *nbt2;
};
struct SmallRunTab {
    SmallRun l[kMaxDepth];
};
struct SmallArgs {
    float *mid;                  // skip buffer [depth + 1][n][32]; slot 0 filled by the init MLP
    float *a2[2];                // collision branch, pre-BatchNorm rows, two-deep
    const float *wimg;           // NNConv MFMA weight images [depth][(T + 1)][kWtType]
    const float *pack;           // [depth][kSpStride]
    const int *tile_col_ptr, *col_meta, *col_src;   // NNConv column structure (graph_prep.hip)
    const int *col_rowptr, *col_nbr;                // collision CSR by destination
    double *part;                // [blocks][128]: bn1 sum | bn1 sumsq | bn2 sum | bn2 sumsq
    unsigned *ctr;               // barrier counter (zeroed by small_pack_kernel)
    int64_t n;
    int n_types, depth, update_running;
    float eps, momentum;
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

constexpr int kCpSc1 = 16;       // cache-policy bit of the raw buffer builtins: sc1 = agent scope (coherent across the XCDs' L2s)
constexpr uint32_t kOob = 0x80000000u;   // offset outside the 2 GB window of every descriptor here: the load returns 0

// gathers of rows other blocks wrote in the previous phase
#ifdef TGNN_SMALL_CACHED
constexpr int kCpGather = 0;     // through L2: needs the invalidate after barrier 2
#else
constexpr int kCpGather = kCpSc1;
#endif
__device__ __forceinline__ float4 ld_gather_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kCpGather));
}
__device__ __forceinline__ float4 ld_sc1_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kCpSc1));
}
__device__ __forceinline__ void st_sc1_f4(__amdgpu_buffer_rsrc_t r, uint32_t off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, kCpSc1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)0x80000000u, 0x00020000);
}

// All blocks are resident (cooperative launch).  Stores of this block are acknowledged (vmcnt(0)) before its arrival
// is published; the data itself is sc1, so no cache maintenance is needed on either side.
__device__ __forceinline__ void small_grid_barrier(unsigned *ctr, unsigned &target, unsigned nblk) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    target += nblk;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

constexpr int kColFirst = 1 << 8, kColLast = 1 << 9, kColSkip = 1 << 11;   // col_meta flags (graph_prep.hip; bit 10 = end of tile: the root column)
constexpr int kSmallThreads = 512;   // 8 waves per tile: 6 NNConv column chunks, 2 halves of the collision neighbourhoods
constexpr int kNnWaves = 6;
// LDS, floats, after the weight images: parameter vectors of two layers | NNConv partial products [6][64][8] (phase B: the
// fp64 fold [8][128]) | second half of the collision sums [64][8] | a1 tile | a2 tile | records | root degrees
constexpr int kLdsSpv = 2 * kSpGinW, kLdsNnRed = kNnWaves * 64 * 8, kLdsGinRed = 64 * 8, kLdsTile = 512;
constexpr int kPfW = 14, kPfG = 4;   // float4 per thread of the next layer's weight images (NNConv, GIN) held across barrier 1

__device__ __forceinline__ f32x4 small_mma6(const bf16x8 *wpl, int plane_stride, const bf16x8 (&x)[3], f32x4 acc) {
    const bf16x8 w0 = wpl[0], w1 = wpl[plane_stride], w2 = wpl[2 * plane_stride];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x[0], acc, 0, 0, 0);   // lo . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[2], acc, 0, 0, 0);   // hi . lo
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[1], acc, 0, 0, 0);   // mid . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[0], acc, 0, 0, 0);   // mid . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[1], acc, 0, 0, 0);   // hi . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[0], acc, 0, 0, 0);   // hi . hi
    return acc;
}

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

__global__ __launch_bounds__(kSmallThreads) void forward_layers_small_kernel(SmallArgs A, SmallRunTab R) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = kSmallThreads;
    const int T = A.n_types, D = A.depth;
    const int64_t n = A.n;
    const int n4w = (T + 1) * kWtType / 4;                        // float4 count of the NNConv image
    float *wl = lds;                                              // NNConv weight image of the layer
    float *gw = wl + (size_t)(T + 1) * kWtType;                   // GIN MLP weight image (contiguous with wl)
    float *spv = gw + kSpGinFrags * 4;                            // [2][kSpGinW]: parameter vectors, by layer parity
    float *nnred = spv + kLdsSpv;
    float *ginred = nnred + kLdsNnRed;
    float *a1s = ginred + kLdsGinRed, *a2s = a1s + kLdsTile;
    float *st = a2s + kLdsTile;                                   // [2][4][32]: records of BN1, BN2
    float *rootdeg = st + 256;                                    // [16]
    double *red = reinterpret_cast<double *>(nnred);              // phase B: [8][128]

    const int tid = threadIdx.x, lane = tid & 63, tw = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    const int64_t tile = blockIdx.x, my_row = tile * 16 + fj;
    const bool row_ok = my_row < n;
    const unsigned nblk = gridDim.x;
    unsigned target = 0;
    const size_t slot = (size_t)n * 32;
    const __amdgpu_buffer_rsrc_t part_rs = rsrc_of(A.part);
#ifdef TGNN_SMALL_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    unsigned long long tacc2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast2 = tlast;
    unsigned long long tacc3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast3 = tlast;
#endif

    // ---- weight images: global -> registers -> LDS; the registers are in flight across barrier 1 of the previous layer.
    // Buffer loads with exact-size descriptors: one offset register per thread, reads past the end return zeros.
    // (native vectors: the float4 struct is copied by memcpy, which pins the array in scratch memory)
    u32x4 pfw[kPfW], pfg[kPfG], pfs;
    const uint32_t pf_off = (uint32_t)tid * 16u;
#define TGNN_SMALL_PREFETCH(LAYER)                                                                                          \
    {                                                                                                                       \
        const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(                                             \
            const_cast<float *>(A.wimg + (size_t)(LAYER) * (T + 1) * kWtType), 0, n4w * 16, 0x00020000);                   \
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(                                             \
            const_cast<float *>(A.pack + (size_t)(LAYER) * kSpStride), 0, kSpStride * 4, 0x00020000);                      \
        _Pragma("unroll") for (int u = 0; u < kPfW; ++u) pfw[u] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, pf_off, u * NT * 16, 0); \
        _Pragma("unroll") for (int u = 0; u < kPfG; ++u)                                                                    \
            pfg[u] = __builtin_amdgcn_raw_buffer_load_b128(g_rs, pf_off, kSpGinW * 4 + u * NT * 16, 0);                     \
        pfs = __builtin_amdgcn_raw_buffer_load_b128(g_rs, tid < kSpGinW / 4 ? pf_off : 0u, 0, 0);                           \
    }
#define TGNN_SMALL_COMMIT(LAYER)                                                                                            \
    {                                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < kPfW; ++u)                                                                    \
            if (tid + u * NT < n4w) reinterpret_cast<u32x4 *>(wl)[tid + u * NT] = pfw[u];                                   \
        _Pragma("unroll") for (int u = 0; u < kPfG; ++u)                                                                    \
            if (tid + u * NT < kSpGinFrags) reinterpret_cast<u32x4 *>(gw)[tid + u * NT] = pfg[u];                           \
        if (tid < kSpGinW / 4) reinterpret_cast<u32x4 *>(spv + ((LAYER) & 1) * kSpGinW)[tid] = pfs;                         \
    }
    TGNN_SMALL_PREFETCH(0)

    // ---- layer-invariant pieces of the tile, kept in registers for all layers
    // NNConv waves: chunk [cb, ce) of the tile's columns; the first 8 columns' gather offsets and meta words
    int cb = 0, ce = 0;
    uint32_t coff[8];
    int cmeta[8];
    float my_root_deg = 0.f;                                      // the wave that holds the root column: max(deg, 1) of row fj, 0 = row >= n
    // collision waves: half of row fj's neighbour list; the first 8 neighbours' gather offsets
    int gbeg = 0, gend = 0, deg_all = 0;
    uint32_t noff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        coff[k] = kOob;
        cmeta[k] = kColSkip;
        noff[k] = kOob;
    }
    if (tw < kNnWaves) {
        const int c0 = __builtin_amdgcn_readfirstlane(A.tile_col_ptr[tile]);
        const int c1 = __builtin_amdgcn_readfirstlane(A.tile_col_ptr[tile + 1]);
        const int nc = c1 - c0;
        cb = c0 + nc * tw / kNnWaves;
        ce = c0 + nc * (tw + 1) / kNnWaves;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (cb + k < ce) {                                    // wave-uniform
                const int s = A.col_src[(int64_t)(cb + k) * 16 + fj];
                const int m = __builtin_amdgcn_readfirstlane(A.col_meta[cb + k]);
                const bool root = (m & 0xff) == T;
                cmeta[k] = m;
                coff[k] = s >= 0 ? (root ? (uint32_t)my_row : (uint32_t)s) * 128u + (uint32_t)fq * 32u : kOob;
                if (root) my_root_deg = s >= 0 ? __int_as_float(s) : 0.f;
            }
        }
    } else if (row_ok) {
        const int b0 = A.col_rowptr[my_row], e0 = A.col_rowptr[my_row + 1];
        deg_all = e0 - b0;
        const int mid = b0 + (deg_all + 1) / 2;
        gbeg = tw == kNnWaves ? b0 : mid;
        gend = tw == kNnWaves ? mid : e0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (gbeg + k < gend) noff[k] = (uint32_t)A.col_nbr[gbeg + k] * 128u + (uint32_t)fq * 32u;
    }
    TGNN_SMALL_COMMIT(0)
    __syncthreads();

    for (int layer = 0; layer < D; ++layer) {
        const float *sp = spv + (layer & 1) * kSpGinW;
        TGNN_ST(0)
        // =========================================== phase A ===========================================
        float gacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // collision waves: this half's neighbourhood sum
        float4 self0 = make_float4(0.f, 0.f, 0.f, 0.f), self1 = self0;
        if (tw < kNnWaves) {
            // ---- NNConv: partial product over columns [cb, ce) -- every gather of the chunk in flight at once
            TGNN_ST3_RESET
            const __amdgpu_buffer_rsrc_t h_rs = rsrc_of(A.mid + (size_t)layer * slot);
            float4 x[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x[k][0] = ld_gather_f4(h_rs, coff[k]);
                x[k][1] = ld_gather_f4(h_rs, coff[k] == kOob ? kOob : coff[k] + 16u);
            }
            TGNN_ST3(0)
#ifdef TGNN_SMALL_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TGNN_ST3(1)
#endif
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;           // D^T: row fj, channels 4 fq + r and 16 + 4 fq + r
            float af[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int mu = cmeta[k];
                if (!(mu & kColSkip)) {                           // wave-uniform
                    const bool first = (mu & kColFirst) || k == 0, last = (mu & kColLast) || cb + k == ce - 1;
                    const float xv[8] = {x[k][0].x, x[k][0].y, x[k][0].z, x[k][0].w, x[k][1].x, x[k][1].y, x[k][1].z, x[k][1].w};
                    if (first) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) af[c] = xv[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) af[c] += xv[c];
                    }
                    if (last) small_run_mma(wl, mu & 0xff, lane, af, (mu & 0xff) == T ? my_root_deg : 1.0f, d0, d1);
                }
            }
            TGNN_ST3(2)
            // chunks longer than 8 columns (tiles with more than 48): one column at a time, index words from memory
            for (int p = cb + 8; p < ce; ++p) {
                const int s = A.col_src[(int64_t)p * 16 + fj];
                const int mu = __builtin_amdgcn_readfirstlane(A.col_meta[p]);
                const bool root = (mu & 0xff) == T;
                const uint32_t off = s >= 0 ? (root ? (uint32_t)my_row : (uint32_t)s) * 128u + (uint32_t)fq * 32u : kOob;
                const float4 y0 = ld_gather_f4(h_rs, off), y1 = ld_gather_f4(h_rs, off == kOob ? kOob : off + 16u);
                const float xv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
                if (mu & kColFirst) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) af[c] = xv[c];
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) af[c] += xv[c];
                }
                if (root) my_root_deg = s >= 0 ? __int_as_float(s) : 0.f;
                if ((mu & kColLast) || p == ce - 1) small_run_mma(wl, mu & 0xff, lane, af, root ? my_root_deg : 1.0f, d0, d1);
            }
            float *mine = nnred + (tw * 64 + lane) * 8;
            *reinterpret_cast<float4 *>(mine) = make_float4(d0[0], d0[1], d0[2], d0[3]);
            *reinterpret_cast<float4 *>(mine + 4) = make_float4(d1[0], d1[1], d1[2], d1[3]);
            if (tw == kNnWaves - 1 && fq == 0) rootdeg[fj] = my_root_deg;   // (the last chunk holds the tile's last column: the root)
            TGNN_ST3(3)
        } else {
            // ---- CollConv, gather: sum over this half of row fj's collision neighbours of (x - mean), x = the previous layer's
            //      pre-BatchNorm rows (the BatchNorm is folded into the sum as in gin32_aggregate_kernel), in the B-operand layout
            //      of the MLP: lane (n, q) holds floats 8 q .. 8 q + 7 of row n
            const float *src = layer == 0 ? A.mid : ((layer - 1) & 1 ? A.a2[1] : A.a2[0]);
            const __amdgpu_buffer_rsrc_t a_rs = rsrc_of(src);
            float4 xr[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xr[k][0] = ld_gather_f4(a_rs, noff[k]);
                xr[k][1] = ld_gather_f4(a_rs, noff[k] == kOob ? kOob : noff[k] + 16u);
            }
            if (tw == kNnWaves) {
                const uint32_t self_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 32u : kOob;
                self0 = ld_gather_f4(a_rs, self_off);
                self1 = ld_gather_f4(a_rs, self_off == kOob ? kOob : self_off + 16u);
            }
            const bool use_stat = layer > 0;
            float mhi[8], mlo[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                mhi[c] = use_stat ? st[128 + 8 * fq + c] : 0.f;
                mlo[c] = use_stat ? st[128 + 32 + 8 * fq + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xv[8] = {xr[k][0].x, xr[k][0].y, xr[k][0].z, xr[k][0].w, xr[k][1].x, xr[k][1].y, xr[k][1].z, xr[k][1].w};
                if (noff[k] != kOob) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) gacc[c] += (xv[c] - mhi[c]) - mlo[c];
                }
            }
            for (int e = gbeg + 8; __any(e < gend); ++e) {       // more than 16 collision neighbours: one at a time
                const uint32_t off = e < gend ? (uint32_t)A.col_nbr[e] * 128u + (uint32_t)fq * 32u : kOob;
                const float4 y0 = ld_gather_f4(a_rs, off), y1 = ld_gather_f4(a_rs, off == kOob ? kOob : off + 16u);
                const float xv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
                if (e < gend) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) gacc[c] += (xv[c] - mhi[c]) - mlo[c];
                }
            }
            if (tw == kNnWaves + 1) {
                *reinterpret_cast<float4 *>(ginred + lane * 8) = make_float4(gacc[0], gacc[1], gacc[2], gacc[3]);
                *reinterpret_cast<float4 *>(ginred + lane * 8 + 4) = make_float4(gacc[4], gacc[5], gacc[6], gacc[7]);
            }
        }
        TGNN_ST(1)
        __syncthreads();
        // the next layer's weight images start their trip now -- behind this layer's gathers in the CU's memory pipeline (114 KB:
        // ahead of them they delayed every gather by ~1 us), in flight during the epilogue / MLP below, in registers until
        // this layer is through with the images in LDS
        if (layer + 1 < D) TGNN_SMALL_PREFETCH(layer + 1)
        TGNN_ST(2)
        if (tw == kNnWaves - 1) {
            // ---- NNConv epilogue: the six partial products in fixed order, mean, bias, LeakyReLU
            float t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < kNnWaves; ++w) {
                const float *pw = nnred + (w * 64 + lane) * 8;
                const float4 a = *reinterpret_cast<const float4 *>(pw), b = *reinterpret_cast<const float4 *>(pw + 4);
                if (w == 0) {
                    t8[0] = a.x; t8[1] = a.y; t8[2] = a.z; t8[3] = a.w; t8[4] = b.x; t8[5] = b.y; t8[6] = b.z; t8[7] = b.w;
                } else {
                    t8[0] += a.x; t8[1] += a.y; t8[2] += a.z; t8[3] += a.w; t8[4] += b.x; t8[5] += b.y; t8[6] += b.z; t8[7] += b.w;
                }
            }
            const float rd = rootdeg[fj];
            const bool valid = rd > 0.f;
            const float inv = valid ? 1.0f / rd : 0.f;
            const float4 bias0 = *reinterpret_cast<const float4 *>(sp + kSpBias + 4 * fq);
            const float4 bias1 = *reinterpret_cast<const float4 *>(sp + kSpBias + 16 + 4 * fq);
            float4 o0, o1;
            o0.x = leakyf_(fmaf(t8[0], inv, bias0.x)); o0.y = leakyf_(fmaf(t8[1], inv, bias0.y));
            o0.z = leakyf_(fmaf(t8[2], inv, bias0.z)); o0.w = leakyf_(fmaf(t8[3], inv, bias0.w));
            o1.x = leakyf_(fmaf(t8[4], inv, bias1.x)); o1.y = leakyf_(fmaf(t8[5], inv, bias1.y));
            o1.z = leakyf_(fmaf(t8[6], inv, bias1.z)); o1.w = leakyf_(fmaf(t8[7], inv, bias1.w));
            if (!valid) o0 = o1 = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(a1s + fj * 32 + 4 * fq) = o0;
            *reinterpret_cast<float4 *>(a1s + fj * 32 + 16 + 4 * fq) = o1;
        } else if (tw == kNnWaves) {
            // ---- CollConv: z = gamma' ((1 + eps)(x[v] - mean) + sum) + (1 + eps + deg) beta, then the MLP of gin32_mlp_kernel
            TGNN_ST2_RESET
            const bool use_stat = layer > 0;
            const float one_eps = sp[kSpEps];
            const float kb = one_eps + (float)deg_all;
            float z[8];
            {
                const float4 h0 = *reinterpret_cast<const float4 *>(ginred + lane * 8), h1 = *reinterpret_cast<const float4 *>(ginred + lane * 8 + 4);
                const float other[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                const float sv[8] = {self0.x, self0.y, self0.z, self0.w, self1.x, self1.y, self1.z, self1.w};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float mh = use_stat ? st[128 + 8 * fq + c] : 0.f, ml = use_stat ? st[128 + 32 + 8 * fq + c] : 0.f;
                    const float gvc = use_stat ? st[128 + 64 + 8 * fq + c] : 1.f, bvc = use_stat ? st[128 + 96 + 8 * fq + c] : 0.f;
                    z[c] = fmaf(gvc, fmaf(one_eps, (sv[c] - mh) - ml, gacc[c] + other[c]), kb * bvc);
                }
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
            auto sig_out = [](float v) { return 1.0f / (1.0f + expf(-v)); };     // full precision, as gin32_mlp_kernel
            float4 r0, r1;
            r0.x = leakyf_(sig_out(o0[0])); r0.y = leakyf_(sig_out(o0[1])); r0.z = leakyf_(sig_out(o0[2])); r0.w = leakyf_(sig_out(o0[3]));
            r1.x = leakyf_(sig_out(o1[0])); r1.y = leakyf_(sig_out(o1[1])); r1.z = leakyf_(sig_out(o1[2])); r1.w = leakyf_(sig_out(o1[3]));
            if (!row_ok) r0 = r1 = make_float4(0.f, 0.f, 0.f, 0.f);
            // row fj, channels 4 fq + r and 16 + 4 fq + r: to the LDS tile (merge, BatchNorm sums) and to HBM (next layer's gathers)
            *reinterpret_cast<float4 *>(a2s + fj * 32 + 4 * fq) = r0;
            *reinterpret_cast<float4 *>(a2s + fj * 32 + 16 + 4 * fq) = r1;
            TGNN_ST2(3)
            const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(layer & 1 ? A.a2[1] : A.a2[0]);
            const uint32_t o_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 16u : kOob;
            st_sc1_f4(o_rs, o_off, r0);
            st_sc1_f4(o_rs, o_off == kOob ? kOob : o_off + 64u, r1);
            TGNN_ST2(4)
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
        small_grid_barrier(A.ctr, target, nblk);
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
            fetch(g);
            if (layer + 1 < D) TGNN_SMALL_COMMIT(layer + 1)
            fold();
            if (nblk > 128) {                                     // (at most 256 blocks)
                fetch(g + 128);
                fold();
            }
            red[g * 128 + 2 * jp] = acc0;
            red[g * 128 + 2 * jp + 1] = acc1;
            __syncthreads();
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
                    const SmallRun run = R.l[layer];
                    float *rm = job ? run.rm2 : run.rm1, *rv = job ? run.rv2 : run.rv1;
                    int64_t *nbt = job ? run.nbt2 : run.nbt1;
                    const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
                    rm[ch] = (float)((1.0 - (double)A.momentum) * (double)rm[ch] + (double)A.momentum * mean);
                    rv[ch] = (float)((1.0 - (double)A.momentum) * (double)rv[ch] + (double)A.momentum * unbiased);
                    if (ch == 0) *nbt += 1;
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
            small_grid_barrier(A.ctr, target, nblk);
#ifdef TGNN_SMALL_CACHED
            if (tw == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop what this CU's L1 / this XCD's L2 hold of other XCDs' rows
            __syncthreads();
#endif
        }
        TGNN_ST(7)
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

static size_t small_lds_bytes(int n_types) {
    return ((size_t)(n_types + 1) * kWtType + kSpGinFrags * 4 + kLdsSpv + kLdsNnRed + kLdsGinRed + 2 * kLdsTile) * sizeof(float) +
           (256 + 16) * sizeof(float);
}
constexpr size_t kSmallMaxLds = 160 * 1024 - 256;

static std::atomic<int64_t> g_small_limit{4096};

// 1 = eligible: one 16-row tile per block and at most one block per CU, the weight images of a layer fit LDS and the
// prefetch registers
int small_layout_teams(int64_t n_nodes, int n_types, int depth) {
    const int64_t limit = g_small_limit.load(std::memory_order_relaxed);
    if (n_nodes < 2 || n_nodes > limit || n_nodes > 4096 || depth < 1 || depth > kMaxDepth) return 0;
    if (small_lds_bytes(n_types) > kSmallMaxLds) return 0;
    if ((n_types + 1) * kWtType / 4 > kPfW * kSmallThreads) return 0;
    return 1;
}

size_t small_pack_floats(int depth) { return (size_t)depth * kSpStride; }

void launch_small_pack(const Params &P, int depth, float *pack, unsigned *barrier_ctr, hipStream_t s) {
    for (int lo = 0; lo < depth; lo += kSmallPackChunk) {
        const int nl = depth - lo < kSmallPackChunk ? depth - lo : kSmallPackChunk;
        SmallPackLayers L{};
        for (int k = 0; k < nl; ++k) {
            const int b = P.layer(lo + k);
            L.l[k] = SmallPackLayer{P.f(b + 7), P.f(b + 8), P.f(b + 9), P.f(b + 20), P.f(b + 21), P.f(b + 13),
                                    P.f(b + 14), P.f(b + 15), P.f(b + 16), P.f(b + 17), P.f(b + 18), P.f(b + 19)};
        }
        small_pack_kernel<<<nl, 256, 0, s>>>(L, pack + (size_t)lo * kSpStride, lo == 0 ? barrier_ctr : nullptr);
    }
}

// mid slot 0 holds the init MLP's output; on return (stream order) slots 1 .. depth are filled
int launch_forward_layers_small(int teams, const Params &P, float *mid, float *a2_0, float *a2_1, const float *wimg,
                                const float *pack, const tgnn_graph *graph, double *part, unsigned *ctr, int64_t n,
                                int depth, int update_running, float eps, float momentum, hipStream_t s) {
    (void)teams;
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
    A.ctr = ctr;
    A.n = n;
    A.n_types = graph->n_types;
    A.depth = depth;
    A.update_running = update_running;
    A.eps = eps;
    A.momentum = momentum;
    SmallRunTab R{};
    for (int i = 0; i < depth; ++i) {
        const BnPtrs b1 = P.bn(P.layer(i) + 8), b2 = P.bn(P.layer(i) + 20);
        R.l[i] = SmallRun{b1.rm, b1.rv, b1.nbt, b2.rm, b2.rv, b2.nbt};
    }
    const int blocks = (int)((n + 15) / 16);
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(forward_layers_small_kernel, (int)kSmallMaxLds, site));
    void *args[] = {&A, &R};
    TGNN_CHECK_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(forward_layers_small_kernel), dim3(blocks),
                                              dim3(kSmallThreads), args, small_lds_bytes(graph->n_types), s));
    return TGNN_OK;
}

}  // namespace tgnn

#ifdef TGNN_SMALL_TIMING
extern "C" int tgnn_debug_small_timing(unsigned long long *out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnn::g_small_timing), (size_t)n_blocks * 32 * sizeof(unsigned long long));
}
#endif
extern "C" void tgnn_set_small_layout_limit(int64_t n_nodes) { tgnn::g_small_limit.store(n_nodes < 0 ? 0 : n_nodes); }
extern "C" int64_t tgnn_get_small_layout_limit(void) { return tgnn::g_small_limit.load(); }
