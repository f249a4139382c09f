// NNConv(aggr="mean"), network_width 32, "type column" formulation for the MFMA pipe of gfx950.
//
// Reference semantics: GraphConv.forward (/root/reference/graph_networks/layers/edge_conv.py:24-27) over
// PyG 1.3.2 NNConv:   out[v] = mean_{e: dst_e = v} h[src_e] . W_{type_e}  +  h[v] . root + bias  (+ LeakyReLU)
//
// Linearity in the source row lets the sum over edges move INSIDE the matrix product:
//     sum_e h[src_e] . W_{type_e}  =  sum_t ( sum_{e of type t} h[src_e] ) . W_t  =  [S_0 | S_1 | .. | S_{T-1}] . [W_0; ..; W_{T-1}]
// i.e. one dense [16 rows x 32 T] x [32 T x 32] product per 16-row destination tile whose accumulator IS the
// output tile: no scatter, no segmented reduction, no LDS accumulators, no block barrier.  On real layouts a
// row has ~7-10 in-edges over 13 types, almost all of distinct type, so about half of the K-blocks of a row are
// zero -- MFMA work that costs less than the reduction machinery it replaces.
//
// Layout built once per graph (graph_prep.hip: nnconv_col_*_kernel): for every 16-row tile a list of COLUMNS,
// sorted by type; column (t, r) holds for each of the 16 rows the source of its r-th in-edge of type t or -1.
// Columns of the same type are summed in registers (A-operand pre-add, CSR order), the last one of the run
// triggers the MFMAs against W_t.  The last column of a tile is the root column (type T): the row itself,
// with max(deg, 1) in the slot where the others keep the source row.
//
// One wavefront owns a contiguous run of tiles and walks its columns as ONE stream through a DEPTH-deep
// register pipeline (index load -> gather -> consume), across tile boundaries.  D^T = W^T . S^T is computed
// (operands swapped) so that a lane ends up with 4 consecutive output channels of ONE row: float4 stores.
#include <atomic>
#include <type_traits>

#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = tgnn_f16x8;

#ifdef TGNN_TIMING
__device__ unsigned long long g_col_timing[512 * 8 * 8];
#define TGNN_CT(slot) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[slot] += now_ - tlast; tlast = now_; }
#else
#define TGNN_CT(slot)
#endif

constexpr int kColMetaFirst = 1 << 8, kColMetaLast = 1 << 9, kColMetaEnd = 1 << 10, kColMetaSkip = 1 << 11;
constexpr int kColStage = 16 * 20;         // floats of the per-wave BN staging tile: [16 rows][16 cols], row stride 20

// F16: the 3-term fp16-pair split instead of the 6-term bf16 x 3 one (tgnn_common.h: split2_f16).  The summed source rows
// are multiplied by the power of two sx that keeps max_in_degree * max |h| below 2^15 (h_max: the bound the producer of h
// left as float bits), the image's weights by nnconv_weight_scale(max |root|); both come off again with the 1 / deg.
// FAST (ldh == 32, packed 128-byte rows): the steady state of the column stream -- every group whose gathers lie before the
// end of the wave's share -- runs without the masks of the general path: a source word IS the gather offset but for a shift
// (-1 lands beyond the descriptor's range and loads zeros), the root column's own-row offset is kept in a register and moves
// on once per tile, the end-of-share tests are gone.  28 instructions per column become 10; same arithmetic, same bits.
#ifdef TGNN_ABL_BLOCKTIMES
__device__ unsigned long long g_blk_times[2][256];
#endif
template <int DEPTH, int WAVES, int OCC, bool F16, bool FAST>
__global__ __launch_bounds__(WAVES * 64, OCC) void nnconv32_cols_kernel(
    const float *__restrict__ h, int64_t ldh, const int *__restrict__ tile_col_ptr, const int *__restrict__ col_meta,
    const int *__restrict__ col_src, const float *__restrict__ wimg, int n_types, const float *__restrict__ bias,
    int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial, const unsigned *__restrict__ h_max,
    const unsigned *__restrict__ root_max, int deg_log2, unsigned long long *__restrict__ stamp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // stamp (tgnn_forward_stamped): first block in / last block out on the device's wall clock -- the launch's duration as a
    // kernel trace sees it, measured inside the production schedule without an event or a profiler around it
    if (stamp && threadIdx.x == 0) atomicMin(stamp, wall_clock64());
#ifdef TGNN_ABL_BLOCKTIMES
    if (threadIdx.x == 0 && blockIdx.x < 256) g_blk_times[0][blockIdx.x] = wall_clock64();   // (experiment: when do the blocks start?)
#endif
    constexpr int kTy = F16 ? kWtTypeF16 : kWtType;         // floats of one type's image
    float *wl = lds;                                        // [(T+1)][3 (2) planes][2 M blocks][16][4] x 8 bf16 (fp16)
    float *stage = lds + (n_types + 1) * kTy;               // [WAVES][16][20]
    float sx = 1.0f, unscale = 1.0f;
    if (F16) {
        sx = pow2_scale_for(*h_max, deg_log2);
        unscale = 1.0f / (sx * nnconv_weight_scale(*root_max));   // (powers of two: exact)
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    constexpr int kThreads = WAVES * 64;
#ifdef TGNN_ABL_EMPTY
    if (n > 0) return;                                     // (timing ablation: what the forward costs without this kernel)
#endif

#ifdef TGNN_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    {   // weight image: straight copy, all loads of a thread issued before the first LDS store
        const int n4 = (n_types + 1) * kTy / 4;
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

    // ---- this wave's run of 16-row tiles.  Balance is needed per SIMD, not per wave (a wave that finishes early
    // leaves its SIMD to its partners): waves w, w+4, .. of a block share SIMD w & 3 and split ONE contiguous
    // share; shares follow the XCD (block b runs on XCD b % 8), then the block, then the SIMD.
    static_assert(WAVES % 4 == 0, "whole SIMD quads");
    const int64_t n_tiles = (n + 15) / 16;
    const int nblk = gridDim.x;
    int64_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (int64_t)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int64_t slot = blk * 4 + (wave & 3), n_slots = (int64_t)nblk * 4;
    const int64_t q0 = n_tiles * slot / n_slots, q1 = n_tiles * (slot + 1) / n_slots;
    constexpr int kSubs = WAVES / 4;
    const int sub = wave >> 2;
    const int64_t t0 = q0 + (q1 - q0) * sub / kSubs, t1 = q0 + (q1 - q0) * (sub + 1) / kSubs;
    const int cbeg = __builtin_amdgcn_readfirstlane(tile_col_ptr[t0]);
    const int cend = __builtin_amdgcn_readfirstlane(tile_col_ptr[t1]);

    // bias of this lane's 8 output channels: 16 m + 4 q + r
    const float4 bias0 = *reinterpret_cast<const float4 *>(bias + 4 * fq);
    const float4 bias1 = *reinterpret_cast<const float4 *>(bias + 16 + 4 * fq);
    // BN partial sums of this lane: channel 16 m + fj over the rows 4 fq .. 4 fq + 3 of every tile
    double bs[2] = {0, 0}, bq[2] = {0, 0};   // (indexed with constants only)
    __syncthreads();

    // ---- pipeline helpers.  Vector-memory INSTRUCTIONS are the scarce resource (scratch/ubench/vmem2.hip: the
    // CU's L1 path moves ~16 B/clk and charges every wave-level load a floor of ~20 cycles, dummy lanes included):
    //   * index data comes per GROUP of 4 columns: one dword load fetches the 64 sources (lane (fj, fq) <- column
    //     fq, row fj), one the 4 meta words; ds_bpermute / v_readlane hand them out (LDS crossbar, scalar pipe: idle)
    //   * the gathers are BUFFER loads: a lane with an out-of-range offset returns 0 without touching memory, so
    //     empty slots cost nothing and need no select; and for this 16-rows x 64-B shape a buffer load costs the
    //     L1 path half of what a global load does (66 vs 129 cycles).
    // Nothing is loaded inside a branch (hipcc would drain vmcnt(0) at the join).
    const __amdgpu_buffer_rsrc_t h_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(h), 0, (int)0x80000000u, 0x00020000);   // 2 GB window
    const uint32_t row_bytes = (uint32_t)ldh * 4u;
    auto load_group = [&](int p, int &s4, int &m4) {        // columns p .. p+3 (reads past cend stay inside the slack)
#ifdef TGNN_ABL_SAMEIDX
        const int pc = cbeg + (p & 0);
#else
        const int pc = p < cend ? p : cbeg;
#endif
        s4 = col_src[(int64_t)pc * 16 + lane];
        m4 = col_meta[pc + (lane & 3)];
    };
    auto unpack = [&](int p, int u, int s4, int m4, int &s, int &m) {
        const bool ok = p + u < cend;                        // wave-uniform
        const int sv = __shfl(s4, u * 16 + fj, 64);
        const int mv = __builtin_amdgcn_readlane(m4, u);
        s = ok ? sv : -1;
        m = ok ? mv : kColMetaSkip;
    };
    int64_t gtile = t0;                                     // tile of the column the gather stage is at
    auto issue_gather = [&](int s, int mu, float4 (&x)[2]) {
        const bool root = (mu & 0xff) == n_types && !(mu & kColMetaSkip);
        const uint32_t row = root ? (uint32_t)(gtile * 16 + fj) : (uint32_t)s;
#ifdef TGNN_ABL_NOGATHER
        const uint32_t off = 0x80000000u + (row & 0);
#else
        const uint32_t off = s >= 0 ? row * row_bytes + (uint32_t)fq * 32u : 0x80000000u;   // s < 0: empty slot / row >= n
#endif
        x[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off, 0, 0));
        x[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off + 16u, 0, 0));
        if (root) ++gtile;                                   // wave-uniform
    };
    // steady state (FAST): see the kernel's header.  own_off: this lane's 32 bytes of its own row of the tile the gather
    // stage is at (rows past n, last tile: out of range)
    const uint32_t fq_bytes = (uint32_t)fq * 32u;
    auto own_off_of = [&](int64_t tile) -> uint32_t {
        const int64_t r = tile * 16 + fj;
        return r < n ? (uint32_t)r * 128u + fq_bytes : 0x80000000u;
    };
    uint32_t own_off = 0;
    auto unpack_steady = [&](int u, int s4, int m4, int &s, int &m) {
        s = __shfl(s4, u * 16 + fj, 64);
        m = __builtin_amdgcn_readlane(m4, u);
    };
    auto issue_gather_steady = [&](int s, int mu, float4 (&x)[2]) {
        const bool root = (mu & kColMetaEnd) != 0;           // wave-uniform; the root column is the one that ends a tile
        const uint32_t off = root ? own_off : ((uint32_t)s << 7) + fq_bytes;   // s = -1: 0xffffff80 + .. >= 2 GB, loads zeros
        x[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off, 0, 0));
        x[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off + 16u, 0, 0));
        if (root) {
            ++gtile;
            own_off = own_off_of(gtile);
        }
    };

    // ONE accumulator pair: the root block's operand is pre-multiplied by max(deg, 1) (the root column carries it), so
    // that a single 1/deg at the end turns the edge sum into the mean and leaves the root term as it is
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;               // D^T tiles: channels 4 fq + r and 16 + 4 fq + r of row fj
    float af[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t ctile = t0;
    auto consume = [&](auto steady, int s, int mu, const float4 (&x)[2]) {
        if constexpr (!decltype(steady)::value)
            if (mu & kColMetaSkip) return;                   // wave-uniform (a steady-state column is never a skip)
        const int t = mu & 0xff;
        const bool valid = s >= 0;                           // (empty slots were loaded as zeros)
        const float xv[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
        if (mu & kColMetaFirst) {
#pragma unroll
            for (int k = 0; k < 8; ++k) af[k] = xv[k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) af[k] += xv[k];
        }
        TGNN_CT(1)
        if (mu & kColMetaLast) {
            // bf16 x 3 split precision (see dense.hip): the summed source rows are split exactly into hi + mid + lo,
            // the weights were split when the image was built; six cross terms per M block, smallest first, fp32
            // accumulation -- 12 MFMAs of ~18 cycles with K = 32 in one instruction instead of 16 fp32 MFMAs of ~36.
            // Matrix and vector time ADD on this chip, so the matrix cycles saved pay for the 44 split instructions.
            const bool root = t == n_types;                  // wave-uniform
            const float scale = root ? (s >= 0 ? __int_as_float(s) : 0.f) : 1.0f;   // root column: max(deg, 1) in the source slot
            constexpr int kPl = kWtPlane / 4;                // 16-byte fragments per plane
            if constexpr (F16) {
                f16x8 xh, xl;
                split2_f16(af, scale * sx, xh, xl);
                const f16x8 *wp = reinterpret_cast<const f16x8 *>(wl + t * kTy) + lane;      // lane order: conflict-free
                const f16x8 h0 = wp[0], h1 = wp[64], l0 = wp[kPl], l1 = wp[kPl + 64];
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, xh, d0, 0, 0, 0);   // lo . hi
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, xh, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xl, d0, 0, 0, 0);   // hi . lo
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xl, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xh, d0, 0, 0, 0);   // hi . hi
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xh, d1, 0, 0, 0);
            } else {
            bf16x8 xh, xm, xl;
            {
                float as[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) as[k] = af[k] * scale;
                split3_trunc(as, xh, xm, xl);
            }
            const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wl + t * kWtType) + lane;   // lane order: conflict-free
            const bf16x8 h0 = wp[0], h1 = wp[64], m0 = wp[kPl], m1 = wp[kPl + 64], l0 = wp[2 * kPl], l1 = wp[2 * kPl + 64];
#ifdef TGNN_ABL_NOMFMA
            d0[0] += (float)h0[0] * af[0]; d1[0] += (float)h1[0] * af[0];
#else
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
#endif
            }
        }
        TGNN_CT(2)
        if (mu & kColMetaEnd) {
            // lane (fj, fq): row fj of the tile, channels 4 fq .. 4 fq + 3 (de0/dr0) and 16 + the same (de1/dr1)
            const float inv = valid ? unscale / __int_as_float(s) : 0.f;   // root column (always the last): max(deg, 1)
            const int64_t v = ctile * 16 + fj;
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
            if (bn_partial) {
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
        }
        TGNN_CT(3)
    };

    // ---- the column stream, in groups of 4 columns, G = DEPTH / 4 groups in flight: a group's index words are
    //      fetched two rounds (8 G columns) ahead, its gathers one round (4 G columns) ahead, each gather into the
    //      registers the consumed column just freed.  (Measured: deeper than one group does not pay -- at 4 waves
    //      per SIMD the waves cover each other's memory latency, and the extra registers cost occupancy.)
    static_assert(DEPTH % 4 == 0, "the pipeline moves groups of 4 columns");
    constexpr int G = DEPTH / 4;
    int s4n[G], m4n[G];
    int xs[G][4], xm[G][4];
    float4 x[G][4][2];
    {
        int s4[G], m4[G];
#pragma unroll
        for (int g = 0; g < G; ++g) load_group(cbeg + 4 * g, s4[g], m4[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) load_group(cbeg + 4 * (G + g), s4n[g], m4n[g]);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unpack(cbeg + 4 * g, u, s4[g], m4[g], xs[g][u], xm[g][u]);
                issue_gather(xs[g][u], xm[g][u], x[g][u]);
            }
    }
    TGNN_CT(0)
    int base = cbeg;
    if constexpr (FAST) {
        own_off = own_off_of(gtile);
        for (; base + 8 * G <= cend; base += 4 * G) {        // the groups gathered in this round lie before cend
#pragma unroll
            for (int g = 0; g < G; ++g) {
                int s4c, m4c;
                load_group(base + 4 * (2 * G + g), s4c, m4c);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    consume(std::true_type{}, xs[g][u], xm[g][u], x[g][u]);
                    unpack_steady(u, s4n[g], m4n[g], xs[g][u], xm[g][u]);
                    issue_gather_steady(xs[g][u], xm[g][u], x[g][u]);
                }
                s4n[g] = s4c;
                m4n[g] = m4c;
            }
        }
    }
    for (; base < cend; base += 4 * G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int s4c, m4c;
            load_group(base + 4 * (2 * G + g), s4c, m4c);
            TGNN_CT(6)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                consume(std::false_type{}, xs[g][u], xm[g][u], x[g][u]);
                unpack(base + 4 * (G + g), u, s4n[g], m4n[g], xs[g][u], xm[g][u]);
                TGNN_CT(4)
                issue_gather(xs[g][u], xm[g][u], x[g][u]);
                TGNN_CT(5)
            }
            s4n[g] = s4c;
            m4n[g] = m4c;
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
    if (stamp && tid == 0) atomicMax(stamp + 1, wall_clock64());
#ifdef TGNN_ABL_BLOCKTIMES
    if (tid == 0 && blockIdx.x < 256) g_blk_times[1][blockIdx.x] = wall_clock64();
#endif
#ifdef TGNN_TIMING
    TGNN_CT(7)
    if (lane == 0 && blockIdx.x < 512 && wave < 8)
        for (int k = 0; k < 8; ++k) g_col_timing[(blockIdx.x * 8 + wave) * 8 + k] = tacc[k];
#endif
}

static size_t cols_lds_bytes(int n_types, int waves, bool f16 = false) {
    size_t a = ((size_t)(n_types + 1) * (f16 ? kWtTypeF16 : kWtType) + (size_t)waves * kColStage) * sizeof(float);
    const size_t b = (size_t)waves * 64 * 4 * sizeof(double);
    return a > b ? a : b;
}

constexpr size_t kColsMaxLds = 160 * 1024 - 256;

template <int DEPTH, int WAVES, int OCC, bool F16 = false>
static int launch_cols_t(const float *h, int64_t ldh, const int32_t *tile_col_ptr, const int32_t *col_meta,
                         const int32_t *col_src, const float *wimg, int32_t n_types, const float *bias,
                         int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                         int blocks_per_cu, hipStream_t s, const unsigned *h_max = nullptr, const unsigned *root_max = nullptr,
                         int deg_log2 = 0, unsigned long long *stamp = nullptr) {
    const bool fast = ldh == 32;
    auto kern = fast ? nnconv32_cols_kernel<DEPTH, WAVES, OCC, F16, true> : nnconv32_cols_kernel<DEPTH, WAVES, OCC, F16, false>;
    // the opt-in to > 64 KB of dynamic LDS is a per-device attribute of the function: set once per device (idempotent)
    static LdsOptIn site[2];
    TGNN_CHECK_HIP(opt_in_dynamic_lds(kern, (int)kColsMaxLds, site[fast]));
    const int64_t n_tiles = (n_nodes + 15) / 16;
    // One tile per SIMD before a second wave of a SIMD gets one: a small layout is bound by the latency of a tile, and
    // waves that share a SIMD stretch each other's (matrix and vector issue do not overlap).  Large layouts hit the cap.
    constexpr int tiles_per_block = 4;
    int64_t blocks = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    // Some CUs (4 per XCD) are left to the OTHER chain of the two-stream forward: this kernel's blocks own their CU's
    // whole register file for its whole duration, and the 1-block BatchNorm finalize of the collision chain would
    // otherwise sit in the queue until the first of them retires (rocprof: 4 us alone, 17 us on average beside this one).
    // Measured, cached-layout forward, reserve 0 / 8 / 32 / 64: 20 000 nodes 1.02 / 0.96 / 0.94 / 0.93 ms, 50 000 nodes
    // 1.43 / 1.36 / 1.35 / 1.39, 100 000 nodes 2.29 / 2.27 / 2.23 / 2.25; the isolated kernel at 100k nodes: 45.7 us on
    // 256 CUs, 47.8 on 224 (6 250 tiles are 7 per SIMD either way), 56.5 on 192, 71.6 on 128.
    constexpr int reserve = 32;
    int64_t cap = cus_minus(reserve) * (int64_t)blocks_per_cu;
    if (const int dbg = g_debug_block_cap[0].load(); dbg > 0) cap = (dbg < device_cus() ? dbg : device_cus()) * (int64_t)blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks &= ~7;
    if (blocks < 1) blocks = 1;
    kern<<<(unsigned)blocks, WAVES * 64, cols_lds_bytes(n_types, WAVES, F16), s>>>(
        h, ldh, tile_col_ptr, col_meta, col_src, wimg, n_types, bias, n_nodes, act, out, bn_partial, h_max, root_max, deg_log2, stamp);
    if (n_partials_host) *n_partials_host = (int32_t)blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int launch_nnconv_cols(const float *h, int64_t ldh, const int32_t *tile_col_ptr, const int32_t *col_meta,
                       const int32_t *col_src, const float *wimg, int32_t n_types, const float *bias,
                       int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                       hipStream_t s, const unsigned *h_max, const unsigned *root_max, int max_in_degree,
                       unsigned long long *stamp) {
    if (h_max && root_max && max_in_degree >= 1) {
        int deg_log2 = 0;
        while ((1 << deg_log2) < max_in_degree) ++deg_log2;
        // (always one 16-wave block per CU: two 8-wave blocks per CU -- they would fit, the fp16 image is 4 KB per type -- spread
        //  over ALL CUs and leave none to the collision chain's 1-block kernels: BatchNorm finalize 7.8 -> 15.2 us, rocprof;
        //  [r4] measured again now that that chain has no 1-block kernel left: cached forward 1.89 -> 2.16 ms at 100 000 nodes)
        return launch_cols_t<4, 16, 4, true>(h, ldh, tile_col_ptr, col_meta, col_src, wimg, n_types, bias, n_nodes, act, out,
                                             bn_partial, n_partials_host, 1, s, h_max, root_max, deg_log2, stamp);
    }
    // 16 waves per CU either way: two 8-wave blocks when two 6 KB-per-type weight images fit the LDS (T <= 11), else
    // one 16-wave block.  Measured with the fp32 kernel at N = 100k, T = 13 (us): <depth 4, 16 waves/CU> 55.7 |
    // <16, 8> 66.0 | <8, 8> 64.9 | <16, 4> 85.2 | <32, 4> 90.2
    if (cols_lds_bytes(n_types, 8) * 2 <= 160 * 1024)
        return launch_cols_t<4, 8, 4>(h, ldh, tile_col_ptr, col_meta, col_src, wimg, n_types, bias, n_nodes, act, out,
                                      bn_partial, n_partials_host, 2, s, nullptr, nullptr, 0, stamp);
    return launch_cols_t<4, 16, 4>(h, ldh, tile_col_ptr, col_meta, col_src, wimg, n_types, bias, n_nodes, act, out,
                                   bn_partial, n_partials_host, 1, s, nullptr, nullptr, 0, stamp);
}

}  // namespace tgnn
#ifdef TGNN_TIMING
extern "C" int tgnn_debug_col_timing(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tgnn::g_col_timing), sizeof(unsigned long long) * 512 * 8 * 8);
}
#endif

using namespace tgnn;

extern "C" int32_t tgnn_nnconv_cols_max_types(void) {
    return (int32_t)((kColsMaxLds / sizeof(float) - 16 * kColStage) / kWtType) - 1;
}

extern "C" int tgnn_nnconv_mean_cols_fwd(const float *h, int64_t ldh, const int32_t *tile_col_ptr,
                                         const int32_t *col_meta, const int32_t *col_src, const float *wtab,
                                         int32_t n_types, const float *root, const float *bias, int64_t n_nodes,
                                         int32_t c, int32_t act, float *out, float *wimg_scratch, double *bn_partial,
                                         int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && c == 32, "the column NNConv kernel is built for network_width 32");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && tile_col_ptr && col_meta && col_src && root && bias && out && wimg_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(ldh >= 32 && ldh % 4 == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                       ((uintptr_t)bias % 16) == 0 && ((uintptr_t)wimg_scratch % 16) == 0, "alignment");
    if (n_types > tgnn_nnconv_cols_max_types()) {
        set_error("tgnn_nnconv_mean_cols_fwd: %d edge types do not fit the LDS weight image (max %d)", n_types,
                  tgnn_nnconv_cols_max_types());
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_nnconv_weight_image(wtab, &root, n_types, 1, wimg_scratch, s);
    return launch_nnconv_cols(h, ldh, tile_col_ptr, col_meta, col_src, wimg_scratch, n_types, bias, n_nodes, act, out,
                              bn_partial, n_partials_host, s);
}

extern "C" int tgnn_nnconv_mean_cols_f16_fwd(const float *h, int64_t ldh, int64_t n_src_rows, const int32_t *tile_col_ptr,
                                             const int32_t *col_meta, const int32_t *col_src, const float *wtab,
                                             int32_t n_types, const float *root, const float *bias, int64_t n_nodes,
                                             int32_t max_in_degree, int32_t act, float *out, float *wimg_scratch,
                                             uint32_t *bounds_scratch, double *bn_partial, int32_t *n_partials_host,
                                             tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_src_rows >= n_nodes && max_in_degree >= 1, "shape");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && tile_col_ptr && col_meta && col_src && root && bias && out && wimg_scratch && bounds_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(ldh == 32 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                       ((uintptr_t)wimg_scratch % 16) == 0 && ((uintptr_t)root % 16) == 0, "alignment / packed rows");
    if (n_types > tgnn_nnconv_cols_max_types()) {
        set_error("tgnn_nnconv_mean_cols_f16_fwd: %d edge types do not fit the LDS weight image (max %d)", n_types,
                  tgnn_nnconv_cols_max_types());
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // bounds: [0] = max |h| over every row that can be gathered, [1] = max |root|
    launch_forward_scales(bounds_scratch, 2, &root, 1, bounds_scratch + 1, nullptr, 0, nullptr, s);
    launch_absmax(h, n_src_rows * 32, bounds_scratch, s);
    launch_nnconv_weight_image(wtab, &root, n_types, 1, wimg_scratch, s, bounds_scratch + 1);
    return launch_nnconv_cols(h, ldh, tile_col_ptr, col_meta, col_src, wimg_scratch, n_types, bias, n_nodes, act, out,
                              bn_partial, n_partials_host, s, bounds_scratch, bounds_scratch + 1, max_in_degree);
}

#ifdef TGNN_ABL_BLOCKTIMES
extern "C" int tgnn_debug_block_times(unsigned long long *host512) {
    return (int)hipMemcpyFromSymbol(host512, HIP_SYMBOL(tgnn::g_blk_times), sizeof(unsigned long long) * 512);
}
#endif

