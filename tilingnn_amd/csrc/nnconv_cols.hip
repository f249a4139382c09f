// NNConv(aggr="mean"), network_width 32, "type column" formulation for the MFMA pipe of gfx950.
//
// Reference semantics: GraphConv.forward (/root/reference/graph_networks/layers/edge_conv.py:24-27) over
// PyG 1.3.2 NNConv:   out[v] = mean_{e: dst_e = v} h[src_e] . W_{type_e}  +  h[v] . root + bias  (+ LeakyReLU)
//
// Linearity in the source row lets the sum over edges move INSIDE the matrix product:
//     sum_e h[src_e] . W_{type_e}  =  sum_t ( sum_{e of type t} h[src_e] ) . W_t  =  [S_0 | S_1 | .. | S_{T-1}] . [W_0; ..; W_{T-1}]
// i.e. one dense [16 rows x 32 T] x [32 T x 32] product per 16-row destination tile whose accumulator IS the
// output tile: no scatter, no segmented reduction, no LDS accumulators, no block barrier.
//
// Layout built once per graph (graph_prep.hip: nnconv_col_*_kernel), a STREAM of columns per wavefront ("part"):
//   * per 16-row tile the columns sorted by type; column (t, r) holds for each of the 16 rows the BYTE OFFSET of the source
//     row of its r-th in-edge of type t (0x80000000 = none: the buffer load of that lane fetches nothing and returns 0);
//     columns of a type are summed in registers, the last one of the run fires the MFMAs against W_t;
//   * then a "degree" column (not gathered from: its 16 words are the float bits of -max(deg, 1), which as offsets are
//     out of range too) and the root column (type T: the rows themselves), which closes the tile;
//   * parts are cut at tile boundaries with equal column counts (the builder balances what the kernel's time is
//     proportional to) and padded with skip columns to whole chunks of 8.
// What the second version of this kernel removed was the per-column bookkeeping the first one spent most of its
// instructions on (rocprof, round 1: 1 904 vector + 757 scalar instructions per tile, 57 per column): offsets arrive
// pre-multiplied, one LDS read with an immediate offset hands a column's word to its 4 lanes per row (index words travel
// global -> LDS in chunks of 8 columns, no cross-lane permutes), meta words come through the scalar cache eight at a
// time, there is no per-column end-of-range test (padding) and no root special case in the gather path.
//
// One wavefront walks its part as ONE stream through a 4-deep register pipeline (gather -> consume), across tile
// boundaries.  D^T = W^T . S^T is computed (operands swapped) so that a lane ends up with 4 consecutive output channels
// of ONE row: float4 stores.
#include <atomic>

#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kColStage = 16 * 20;         // floats of the per-wave BN staging tile: [16 rows][16 cols], row stride 20
constexpr int kIdxSlots = 4;               // chunks of index words a wave keeps in LDS (power of two; 3 are live)
constexpr int kIdxChunkWords = kColChunk * 16;

template <int WAVES, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void nnconv32_cols_kernel(
    const float *__restrict__ h, uint32_t h_bytes, const int *__restrict__ part_ptr, const uint32_t *__restrict__ col_meta,
    const uint32_t *__restrict__ col_off, const float *__restrict__ wimg, int n_types, const float *__restrict__ bias,
    int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;                                        // [(T+1)][3 planes][2 M blocks][16][4] x 8 bf16
    uint32_t *idx_all = reinterpret_cast<uint32_t *>(lds + (n_types + 1) * kWtType);     // [WAVES][kIdxSlots][8][16]
    float *stage = lds + (n_types + 1) * kWtType + WAVES * kIdxSlots * kIdxChunkWords;   // [WAVES][16][20]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    constexpr int kThreads = WAVES * 64;

    {   // weight image: straight copy, all loads of a thread issued before the first LDS store
        const int n4 = (n_types + 1) * kWtType / 4;
        for (int i = tid; i < n4; i += 4 * kThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + u * kThreads < n4 ? i + u * kThreads : n4 - 1;
                v[u] = reinterpret_cast<const float4 *>(wimg)[ii];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * kThreads < n4) reinterpret_cast<float4 *>(wl)[i + u * kThreads] = v[u];
        }
    }
    float *stg = stage + wave * kColStage;
    uint32_t *idx_w = idx_all + wave * (kIdxSlots * kIdxChunkWords);

    // ---- this wave's part.  Waves w, w+4, .. of a block share SIMD w & 3 and take consecutive parts; parts follow the
    //      XCD (block b runs on XCD b % 8), then the block, then the SIMD: neighbouring tiles share an L2.
    static_assert(WAVES % 4 == 0, "whole SIMD quads");
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int part = (blk * 4 + (wave & 3)) * (WAVES / 4) + (wave >> 2);
    const int tile0 = __builtin_amdgcn_readfirstlane(part_ptr[2 * part]);
    const int cbeg = __builtin_amdgcn_readfirstlane(part_ptr[2 * part + 1]);
    const int cend = __builtin_amdgcn_readfirstlane(part_ptr[2 * part + 3]);
#ifdef TGNN_ABL_EMPTY
    const int nchunks = 0 * (cend - cbeg);
#else
    const int nchunks = (cend - cbeg) / kColChunk;          // parts are whole chunks
#endif

    // bias of this lane's 8 output channels: 16 m + 4 q + r
    const float4 bias0 = *reinterpret_cast<const float4 *>(bias + 4 * fq);
    const float4 bias1 = *reinterpret_cast<const float4 *>(bias + 16 + 4 * fq);
    // BN partial sums of this lane: channel 16 m + fj over the rows 4 fq .. 4 fq + 3 of every tile
    double bs[2] = {0, 0}, bq[2] = {0, 0};   // (indexed with constants only)
    __syncthreads();

    // Gathers are BUFFER loads: a lane whose offset is out of range returns 0 without touching memory, so empty slots
    // cost nothing and need no select (scratch/ubench/vmem2.hip); past the end of the part the null descriptor makes
    // every lane of the look-ahead gathers such a lane.  Nothing is loaded inside a branch (hipcc would drain vmcnt(0)
    // at the join).
    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(h), 0, (int)h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(h), 0, 0, 0x00020000);
    const uint32_t lane_off = (uint32_t)fq * 32u;
    // index words of chunk k of this part: 8 columns x 16 words = 128 x 8 bytes, lane l moves words 2l, 2l+1
    auto load_chunk = [&](int k) -> uint2 {
        const int kk = k < nchunks ? k : (nchunks > 0 ? nchunks - 1 : 0);
        return *reinterpret_cast<const uint2 *>(col_off + ((int64_t)cbeg + (int64_t)kk * kColChunk) * 16 + 2 * lane);
    };
    auto put_chunk = [&](int k, uint2 v) { *reinterpret_cast<uint2 *>(idx_w + (k & (kIdxSlots - 1)) * kIdxChunkWords + 2 * lane) = v; };
    auto issue_gather = [&](uint32_t word, __amdgpu_buffer_rsrc_t rs, float4 (&x)[2]) {
#ifdef TGNN_ABL_NOGATHER
        rs = null_rsrc;
#endif
        const uint32_t off = word + lane_off;
        x[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        x[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, 0));
    };

    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;               // D^T tiles: channels 4 fq + r and 16 + 4 fq + r of row fj
    float af[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float ndeg = -1.0f;                                     // -max(deg, 1) of row fj of the tile being consumed
    int64_t ctile = tile0;
    auto consume = [&](uint32_t mu, uint32_t word, const float4 (&x)[2]) {
        if (mu & kColMetaSkip) return;                       // padding (wave-uniform, like every test on mu)
        if (mu & kColMetaDeg) { ndeg = __uint_as_float(word); return; }
        const float xv[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
        if (mu & kColMetaFirst) {
#pragma unroll
            for (int k = 0; k < 8; ++k) af[k] = xv[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) af[k] += xv[k];
        }
        if (!(mu & kColMetaLast)) return;
        // bf16 x 3 split precision (see dense.hip): the summed source rows are split exactly into hi + mid + lo, the
        // weights were split when the image was built; six cross terms per M block, smallest first, fp32 accumulation.
        // ONE accumulator pair: the root block's operand is multiplied by max(deg, 1), so that a single 1/deg at the end
        // turns the edge sum into the mean and leaves the root term as it is.
        const int t = mu & 0xff;
        bf16x8 xh, xm, xl;
        if (mu & kColMetaEnd) {
            float as[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) as[k] = af[k] * -ndeg;
            split3_trunc(as, xh, xm, xl);
        } else {
#ifdef TGNN_ABL_NOSPLIT
            for (int k = 0; k < 8; ++k) xh[k] = (__bf16)af[k];
            xm = xh; xl = xh;
#else
            split3_trunc(af, xh, xm, xl);
#endif
        }
#ifdef TGNN_ABL_NOWREAD
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wl + (t & 0) * kWtType) + fj * 4 + fq;
#else
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wl + t * kWtType) + fj * 4 + fq;
#endif
        constexpr int kPl = kWtPlane / 4;                    // 16-byte fragments per plane
        const bf16x8 h0 = wp[0], h1 = wp[64], m0 = wp[kPl], m1 = wp[kPl + 64], l0 = wp[2 * kPl], l1 = wp[2 * kPl + 64];
#ifdef TGNN_ABL_NOMFMA
        d0[0] += (float)h0[0] * af[0] + (float)m0[0] * af[1] + (float)l0[0] * af[2] + (float)xm[0] + (float)xl[0];
        d1[0] += (float)h1[0] * af[0] + (float)m1[0] * af[1] + (float)l1[0] * af[2] + (float)xh[0];
        if (!(mu & kColMetaEnd)) return;
#endif
#ifndef TGNN_ABL_NOMFMA
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, xh, d0, 0, 0, 0);   // lo . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, xh, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xl, d0, 0, 0, 0);   // hi . lo
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xl, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m0, xm, d0, 0, 0, 0);   // mid . mid
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m1, xm, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m0, xh, d0, 0, 0, 0);   // mid . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m1, xh, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xm, d0, 0, 0, 0);   // hi . mid
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xm, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xh, d0, 0, 0, 0);   // hi . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xh, d1, 0, 0, 0);
        if (!(mu & kColMetaEnd)) return;
#endif
        // ---- the tile is complete.  lane (fj, fq): row fj, channels 4 fq .. 4 fq + 3 (d0) and 16 + the same (d1)
        const int64_t v = ctile * 16 + fj;
        const bool valid = v < n;
        // 1 / max(deg, 1): v_rcp_f32 (1 ulp) + one Newton step on the exact residual = the correctly rounded reciprocal
        // of a small integer, in 3 instructions instead of the ~10 of an IEEE division
        const float inv0 = __builtin_amdgcn_rcpf(-ndeg);
        const float inv = fmaf(fmaf(ndeg, inv0, 1.0f), inv0, inv0);
        float4 o0, o1;
        o0.x = fmaf(d0[0], inv, bias0.x); o0.y = fmaf(d0[1], inv, bias0.y);
        o0.z = fmaf(d0[2], inv, bias0.z); o0.w = fmaf(d0[3], inv, bias0.w);
        o1.x = fmaf(d1[0], inv, bias1.x); o1.y = fmaf(d1[1], inv, bias1.y);
        o1.z = fmaf(d1[2], inv, bias1.z); o1.w = fmaf(d1[3], inv, bias1.w);
        if (act == TGNN_ACT_LEAKY_RELU) {
            o0.x = leakyf_(o0.x); o0.y = leakyf_(o0.y); o0.z = leakyf_(o0.z); o0.w = leakyf_(o0.w);
            o1.x = leakyf_(o1.x); o1.y = leakyf_(o1.y); o1.z = leakyf_(o1.z); o1.w = leakyf_(o1.w);
        }
        if (valid) {
            *reinterpret_cast<float4 *>(out + v * 32 + 4 * fq) = o0;
            *reinterpret_cast<float4 *>(out + v * 32 + 16 + 4 * fq) = o1;
        }
#ifdef TGNN_ABL_NOBN
        if (bn_partial && n < 0) {
#else
        if (bn_partial) {
#endif
            // column sums in fp64: transpose through the wave's own LDS tile, one 16-channel half at a time
            // (written out twice: indexing {o0, o1} with the loop variable would put them in scratch memory,
            //  and a scratch access drains the whole gather pipeline with vmcnt(0))
            auto half_sums = [&](float4 o, double &sum, double &sq) {
                o.x = valid ? o.x : 0.f; o.y = valid ? o.y : 0.f;   // (component selects: `valid ? o : z` on
                o.z = valid ? o.z : 0.f; o.w = valid ? o.w : 0.f;   //  float4 lvalues becomes a POINTER select)
                *reinterpret_cast<float4 *>(stg + fj * 20 + 4 * fq) = o;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double val = (double)stg[(4 * fq + r) * 20 + fj];
                    sum += val;
                    sq += val * val;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            half_sums(o0, bs[0], bq[0]);
            half_sums(o1, bs[1], bq[1]);
        }
        d0 = f32x4{0.f, 0.f, 0.f, 0.f};
        d1 = d0;
        ++ctile;
    };

    // ---- the column stream in chunks of 8: while chunk c is consumed, the gathers of columns 4 ahead are issued (into
    //      the registers the consumed column just freed), chunk c + 1's meta words arrive through the scalar cache,
    //      chunk c + 2's index words move from registers into LDS and chunk c + 3's are requested.
    float4 x[4][2];
    uint32_t xw[4];
    if (nchunks > 0) {
        put_chunk(0, load_chunk(0));
        put_chunk(1, load_chunk(1));
        uint2 pend = load_chunk(2);
        const uint32_t *idx_rd = idx_w + fj;                 // column u of chunk k: idx_rd[(k & 3) * 128 + u * 16]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xw[u] = idx_rd[u * 16];
            issue_gather(xw[u], h_rsrc, x[u]);
        }
        const uint32_t *meta_p = col_meta + cbeg;
        uint32_t mc[kColChunk];
#pragma unroll
        for (int u = 0; u < kColChunk; ++u) mc[u] = __builtin_amdgcn_readfirstlane(meta_p[u]);
        for (int c = 0; c < nchunks; ++c) {
            const int cn = c + 1 < nchunks ? c + 1 : c;      // (scalar)
            uint32_t mn[kColChunk];
#pragma unroll
            for (int u = 0; u < kColChunk; ++u) mn[u] = __builtin_amdgcn_readfirstlane(meta_p[cn * kColChunk + u]);
            put_chunk(c + 2, pend);
            pend = load_chunk(c + 3);
            const uint32_t *rd0 = idx_rd + (c & (kIdxSlots - 1)) * kIdxChunkWords;
            const uint32_t *rd1 = idx_rd + ((c + 1) & (kIdxSlots - 1)) * kIdxChunkWords;
            const __amdgpu_buffer_rsrc_t rs_next = c + 1 < nchunks ? h_rsrc : null_rsrc;
#pragma unroll
            for (int u = 0; u < kColChunk; ++u) {
                consume(mc[u], xw[u & 3], x[u & 3]);
                if (u < 4) {
                    xw[u & 3] = rd0[(u + 4) * 16];
                    issue_gather(xw[u & 3], h_rsrc, x[u & 3]);
                } else {
                    xw[u & 3] = rd1[(u - 4) * 16];
                    issue_gather(xw[u & 3], rs_next, x[u & 3]);
                }
            }
#pragma unroll
            for (int u = 0; u < kColChunk; ++u) mc[u] = mn[u];
        }
    }

    // ---- BN partials of the block: lanes (fj, fq) -> channel 16 m + fj; fold fq, then the waves, in fixed order
    if (bn_partial) {
        __syncthreads();                                     // everybody is done with the weight image
        double *red = reinterpret_cast<double *>(lds);       // [WAVES][64 lanes][4]
        double *mine = red + ((int64_t)wave * 64 + lane) * 4;
        mine[0] = bs[0]; mine[1] = bs[1]; mine[2] = bq[0]; mine[3] = bq[1];
        __syncthreads();
        if (tid < 64) {                                      // tid = which * 32 + channel
            const int which = tid >> 5, ch = tid & 31, m2 = ch >> 4, j = ch & 15;
            double acc = 0;
            for (int w = 0; w < WAVES; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += red[((int64_t)w * 64 + q * 16 + j) * 4 + which * 2 + m2];
            bn_partial[(int64_t)blockIdx.x * 64 + tid] = acc;
        }
    }
}

static size_t cols_lds_bytes(int n_types, int waves) {
    size_t a = ((size_t)(n_types + 1) * kWtType + (size_t)waves * (kIdxSlots * kIdxChunkWords + kColStage)) * sizeof(float);
    const size_t b = (size_t)waves * 64 * 4 * sizeof(double);
    return a > b ? a : b;
}

constexpr size_t kColsMaxLds = 160 * 1024 - 256;

// Launch shape of the column kernel for a layout: also the PARTITION the structure is built for (one part per wavefront).
ColsShape cols_shape(int64_t n_nodes, int n_types) {
    ColsShape sh;
    // 16 waves per CU either way: two 8-wave blocks when two weight images (6 KB per type) fit the LDS, else one 16-wave
    // block.  Measured with the first fp32 kernel at N = 100k, T = 13 (us): <depth 4, 16 waves/CU> 55.7 | <16, 8> 66.0 |
    // <8, 8> 64.9 | <16, 4> 85.2 | <32, 4> 90.2
    // (Many edge types: one 8-wave block per CU is what still fits.)
    const bool two_blocks = cols_lds_bytes(n_types, 8) * 2 <= 160 * 1024;
    sh.waves = two_blocks || cols_lds_bytes(n_types, 16) > kColsMaxLds ? 8 : 16;
    const int blocks_per_cu = two_blocks ? 2 : 1;
    const int64_t n_tiles = (n_nodes + 15) / 16;
    // One tile per SIMD before a second wave of a SIMD gets one: a small layout is bound by the latency of a tile, and
    // waves that share a SIMD stretch each other's (matrix and vector issue do not overlap).  Large layouts hit the cap.
    int64_t blocks = (n_tiles + 3) / 4;
    // Some CUs (4 per XCD) are left to the OTHER chain of the two-stream forward: this kernel's blocks own their CU's
    // whole register file for its whole duration, and the small launches of the collision chain would otherwise sit
    // in the queue until the first of them retires.  Measured (round 1), cached-layout forward, reserve 0 / 8 / 32 / 64:
    // 20 000 nodes 1.02 / 0.96 / 0.94 / 0.93 ms, 50 000 nodes 1.43 / 1.36 / 1.35 / 1.39, 100 000 nodes 2.29 / 2.27 / 2.23 / 2.25.
    const int64_t cap = (int64_t)(256 - kColsReserveCus) * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks &= ~(int64_t)7;
    if (blocks < 1) blocks = 1;
    sh.blocks = (int)blocks;
    return sh;
}

template <int WAVES>
static int launch_cols_t(const float *h, int64_t n_src_rows, const int32_t *part_ptr, const int32_t *col_meta,
                         const int32_t *col_off, const float *wimg, int32_t n_types, const float *bias,
                         int64_t n_nodes, int32_t act, float *out, double *bn_partial, int blocks, hipStream_t s) {
    auto kern = nnconv32_cols_kernel<WAVES, 4>;
    // the opt-in to > 64 KB of dynamic LDS is a per-device attribute of the function: set once per device
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        TGNN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColsMaxLds));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    kern<<<(unsigned)blocks, WAVES * 64, cols_lds_bytes(n_types, WAVES), s>>>(
        h, (uint32_t)(n_src_rows * 128), part_ptr, reinterpret_cast<const uint32_t *>(col_meta),
        reinterpret_cast<const uint32_t *>(col_off), wimg, n_types, bias, n_nodes, act, out, bn_partial);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int launch_nnconv_cols(const float *h, int64_t n_src_rows, const int32_t *part_ptr, const int32_t *col_meta,
                       const int32_t *col_off, const float *wimg, int32_t n_types, const float *bias,
                       int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                       hipStream_t s) {
    const ColsShape sh = cols_shape(n_nodes, n_types);
    if (n_partials_host) *n_partials_host = sh.blocks;
    if (sh.waves == 8)
        return launch_cols_t<8>(h, n_src_rows, part_ptr, col_meta, col_off, wimg, n_types, bias, n_nodes, act, out,
                                bn_partial, sh.blocks, s);
    return launch_cols_t<16>(h, n_src_rows, part_ptr, col_meta, col_off, wimg, n_types, bias, n_nodes, act, out,
                             bn_partial, sh.blocks, s);
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int32_t tgnn_nnconv_cols_max_types(void) {
    const size_t per_wave = (size_t)(kIdxSlots * kIdxChunkWords + kColStage);
    const int32_t t = (int32_t)((kColsMaxLds / sizeof(float) - 8 * per_wave) / kWtType) - 1;
    return t < 200 ? t : 200;              // (the type field of a column's meta word is 8 bits)
}

extern "C" int tgnn_nnconv_mean_cols_fwd(const float *h, int64_t ldh, int64_t n_src_rows, const int32_t *part_ptr,
                                         const int32_t *col_meta, const int32_t *col_off, const float *wtab,
                                         int32_t n_types, const float *root, const float *bias, int64_t n_nodes,
                                         int32_t c, int32_t act, float *out, float *wimg_scratch, double *bn_partial,
                                         int32_t *n_partials_host, tgnn_stream_t stream) {
    TGNN_CHECK_ARG(n_nodes >= 1 && c == 32, "the column NNConv kernel is built for network_width 32");
    TGNN_CHECK_ARG(ldh == 32, "the column structure addresses dense rows of 32 floats (use tgnn_nnconv_mean_fwd otherwise)");
    TGNN_CHECK_ARG(n_src_rows >= n_nodes && n_src_rows * 128 < (int64_t(1) << 31), "source rows must lie within 2 GB");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && part_ptr && col_meta && col_off && root && bias && out && wimg_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                       ((uintptr_t)wimg_scratch % 16) == 0 && ((uintptr_t)col_off % 8) == 0, "alignment");
    if (n_types > tgnn_nnconv_cols_max_types()) {
        set_error("tgnn_nnconv_mean_cols_fwd: %d edge types do not fit the LDS weight image (max %d)", n_types,
                  tgnn_nnconv_cols_max_types());
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_nnconv_weight_image(wtab, &root, n_types, 1, wimg_scratch, s);
    return launch_nnconv_cols(h, n_src_rows, part_ptr, col_meta, col_off, wimg_scratch, n_types, bias, n_nodes, act, out,
                              bn_partial, n_partials_host, s);
}
