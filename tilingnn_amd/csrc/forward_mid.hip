// The 20 message-passing layers of TilinGNN.forward (/root/reference/graph_networks/networks/TilinGNN.py:59-71) for MID-SIZE
// layouts -- 4 097 .. 65 536 nodes: BASELINE config 2 (10 000 nodes), every greedy round of a large solve, a rank's share of
// a 100 000-node layout strong-scaled over 8 GPUs -- as ONE persistent kernel carrying BOTH chains (GraphConv and CollConv).
//
// Why: at these sizes the general schedule (forward.hip) is ~100 dependent launches of ~5 us each; neither chain fills the
// chip, the host cannot queue launches fast enough (0.56 of 0.89 ms at 10 000 nodes) and single-chain fusions lose the
// block-by-block sharing of the device between the chains.  The small-layout kernel (forward_small.hip:
// one 16-row tile per block, 8 waves co-operating on it) does not scale past one tile per CU.  Here one block of 16 waves per
// CU owns a contiguous range of K tiles (K = ceil(tiles / CUs) <= 16) and every tile is a WAVE-PRIVATE work item -- no block
// barrier inside a phase, latency hidden by the other waves of the SIMD:
//
//   NNConv item (waves 0-7), edge_conv.py:24-27 / PyG NNConv mean + root + bias + LeakyReLU:
//       sum_e h[src_e] W_type(e) = sum_t (sum_{e of type t} h[src_e]) W_t.  The tile's in-edges come as BATCHES packed by edge
//       type (tgnn_mid_entries_build): a batch = up to 32 (source row, destination row, store / add) entries of ONE type, 8
//       entries per gather instruction -- whole 128-byte rows on 8 consecutive lanes, the cheap shape on the CU's address path
//        -- no two entries of a destination row in one instruction.  Rows land in a wave-private 2 KB
//       type-sum tile in LDS (store, or read-add-write in edge order), which is then read in the matrix layout, split into a
//       scaled fp16 pair (tgnn_common.h) and multiplied by W_t's fragments from the block's LDS image: 6 matrix instructions
//       per type.  The root term is the tile's own rows times max(deg, 1), so one 1/deg at the end yields mean + root.
//   GIN item (waves 0-11), coll_conv.py:24-27 / PyG GINConv: neighbourhood sum with the previous layer's BatchNorm folded in
//       (whole-row gathers), z through the wave's LDS tile into the matrix layout, the 32 -> 32 -> 64 -> 32 sigmoid MLP on
//       bf16 x 3 fragments (gin.hip's arithmetic), all on one wave.
//
// One layer = NNConv_i items -> [R] all-reduce of the 128 BatchNorm column sums -> merge of the own rows (BN1(a1) * BN2(a2)
// + residual -> slot i + 1) -> [B] grid barrier -> next layer.  The collision chain hides in the barrier's shadow: GIN_{i+1}
// needs only a2_i and its statistics (complete at [R]), so its items run between a block's ARRIVAL at [B] and its wait.
//   [R]  no barrier: every block's partial row carries a 2-bit generation tag in the low mantissa bits of its doubles (2^-51
//        relative, far below the sums' own rounding); a reader polls the rows it needs until the tags match.  Two levels
//        (groups of 16 blocks; every block then reads one copy of every group's sum): 16 + 16 KB per block instead of the 256 KB
//        a flat fold of 256 rows would pull through every CU, fixed order, hence deterministic.
//   [B]  monotonic counter, sc1 data, as in forward_small.hip.  Every wait is bounded (forward_persist.h).
// The init and final MLP stay with the general schedule's kernels (forward.hip) in front of and behind this launch.
//
// Arithmetic: NNConv = nnconv_cols.hip's fp16-pair formulas (same scales, same image), GIN = gin.hip's, merge = bn_merge.hip's;
// what differs from the general schedule is the association of the BatchNorm sums (per tile fp64, per wave, per block, two
// fold levels) and of the same-type source sums (in LDS, edge order as there).  Every order is fixed: bit-reproducible.
#include <mutex>

#include "forward_persist.h"

namespace tgnn {

using f16x8 = tgnn_f16x8;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kMidThreads = 1024, kMidWaves = 16, kMidNnWaves = 8, kMidGinWaves = 12;
constexpr int kMidBatchWords = TGNN_MID_BATCH_WORDS, kMidTileBatches = TGNN_MID_TILE_BATCHES;
constexpr int kMidEntWords = kMidBatchWords * kMidTileBatches;     // 648 words per tile
constexpr int kMidMaxTilesPerBlock = 16;
constexpr int kMidColFirst = 1 << 8;

struct MidArgs {
    float *mid;                  // skip buffer [depth + 1][n][32]; slot 0 filled by the init MLP
    float *a1;                   // [n][32] GraphConv pre-BatchNorm rows of the layer
    float *a2[2];                // CollConv pre-BatchNorm rows, two-deep
    const float *wimg;           // NNConv fp16-pair weight images [depth][(T + 1)][kWtTypeF16]
    const float *pack;           // [depth][kSpStride] parameter vectors + GIN MFMA images (small_pack_kernel)
    const int *adj_rowptr;       // in-degrees of the adjacency set
    const int *col_rowptr, *col_nbr;
    const int *tile_nb;          // batches per tile
    const uint32_t *ent;         // [tiles][kMidEntWords]
    double *part, *gpart;        // [2][blocks][128] tagged partial rows / group sums (zeroed before the launch)
    double *runstat;             // [depth][128] parked batch statistics for the running buffers
    unsigned *ctr;               // barrier counter (zero before the launch)
    unsigned *bounds;            // [0, depth]: max |slot k| as float bits (this kernel fills 1 ..); [depth + 1 ..]: max |root_i|
    unsigned *err;
    unsigned long long spin_budget;
    int64_t n;
    int n_types, depth, update_running, tiles_per_block, deg_log2, fault;
    float eps, momentum;
};

// ---- wave tile in LDS: [16 rows][8 chunks of 16 bytes], chunk c of row r at position c ^ (r & 7): whole-row writes (8 lanes a row),
//      matrix-layout reads (lane (n, q): chunks 2 q, 2 q + 1 of row n) and column walks are all bank-conflict free
__device__ __forceinline__ int mid_chunk(int row, int c) { return row * 32 + ((c ^ (row & 7)) << 2); }   // float index

// ---- one tagged double: the low two mantissa bits carry the generation
__device__ __forceinline__ u32x2 mid_tag(double v, unsigned tag) {
    u32x2 b = __builtin_bit_cast(u32x2, v);
    b[0] = (b[0] & ~3u) | tag;
    return b;
}
__device__ __forceinline__ double mid_untag(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, u32x2{lo & ~3u, hi}); }

// thread (jp = tid & 63, r = tid >> 6): the two doubles 2 jp, 2 jp + 1 of row `row` of a tagged array, polled until both carry `tag`
// (row < 0: zeros)
__device__ __forceinline__ void mid_poll_pair(__amdgpu_buffer_rsrc_t rs, int64_t row, int jp, unsigned tag, double &x0, double &x1,
                                              SpinCtx &sp) {
    x0 = x1 = 0.0;
    if (row < 0) return;
    const uint32_t off = ((uint32_t)row * 128u + 2u * (uint32_t)jp) * 8u;
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0;; ++it) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kCpSc1);
        if ((v[0] & 3u) == tag && (v[2] & 3u) == tag) {
            x0 = mid_untag(v[0], v[1]);
            x1 = mid_untag(v[2], v[3]);
            return;
        }
        if (!spin_continue(sp, t0, it, kSpinErrRows)) return;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- NNConv of one 16-row tile on one wave -------------------------------------------------------------------------------------
// gather lane map: lane = (o = lane >> 3: entry slot of the instruction, p = lane & 7: 16-byte piece of the row)
// matrix lane map: lane = (fj = lane & 15: row of the tile, fq = lane >> 4: k group / output quarter)
struct MidNn {
    const float *wl;             // LDS: this layer's fp16-pair image
    float *tbuf;                 // LDS: the wave's tile
    const uint32_t *ebuf;        // LDS: the tile's batches
    __amdgpu_buffer_rsrc_t h_rs; // slot `layer` of the skip buffer
    float sx, unscale;
    int n_types;
};

__device__ __forceinline__ void mid_nnconv_tile(const MidNn &N, const MidArgs &A, int64_t tile, int nb, const float *bias,
                                                int lane, double &bn_acc) {
    const int fj = lane & 15, fq = lane >> 4, go = lane >> 3, gp = lane & 7;
    const int64_t n = A.n;
    const uint32_t gp16 = (uint32_t)gp * 16u;
    // the tile's own rows in the matrix layout (root run) and their in-degrees: requested first, used last
    const int64_t my_row = tile * 16 + fj;
    const bool row_ok = my_row < n;
    const uint32_t own_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 32u : kOob;
    const float4 own0 = ld_sc1_f4(N.h_rs, own_off), own1 = ld_sc1_f4(N.h_rs, own_off == kOob ? kOob : own_off + 16u);
    int deg_i = 0;
    if (row_ok) deg_i = A.adj_rowptr[my_row + 1] - A.adj_rowptr[my_row];
    const float degf = row_ok ? (float)(deg_i > 0 ? deg_i : 1) : 0.f;

    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;                 // D^T tiles: channels 4 fq + r and 16 + 4 fq + r of row fj
    constexpr int kPl = kWtPlane / 4;                            // 16-byte fragments per plane
    auto run_mma = [&](const float (&af)[8], float scale, int t) {
        f16x8 xh, xl;
        split2_f16(af, scale, xh, xl);
        const f16x8 *wp = reinterpret_cast<const f16x8 *>(N.wl + t * kWtTypeF16) + lane;
        const f16x8 h0 = wp[0], h1 = wp[64], l0 = wp[kPl], l1 = wp[kPl + 64];
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, xh, d0, 0, 0, 0);   // lo . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, xh, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xl, d0, 0, 0, 0);   // hi . lo
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xl, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xh, d0, 0, 0, 0);   // hi . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xh, d1, 0, 0, 0);
    };

    // batches: the gathers of batch b + 1 are in flight while batch b is scattered into the tile and multiplied
    u32x4 ew = {0u, 0u, 0u, 0u};                                 // this lane's four entry words of the batch in flight
    f32x4 x[4];
    auto issue = [&](int b) {
        const bool live = b < nb;                                // (wave-uniform; no load sits behind a branch)
        const int bb = live ? b : 0;
        ew = *reinterpret_cast<const u32x4 *>(N.ebuf + bb * kMidBatchWords + 4 + go * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w = live ? ew[g] : 0u;
            ew[g] = w;
            x[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 N.h_rs, (w >> 31) ? ((w & 0xfffffu) << 7) + gp16 : kOob, 0, kCpSc1));
        }
    };
    issue(0);
    for (int b = 0; b < nb; ++b) {
        const uint32_t hdr0 = __builtin_amdgcn_readfirstlane(N.ebuf[b * kMidBatchWords]);
        const uint32_t hdr1 = __builtin_amdgcn_readfirstlane(N.ebuf[b * kMidBatchWords + 1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w = ew[g];
            const bool valid = (w >> 31) != 0u, add = ((w >> 24) & 1u) != 0u;
            const int row = (int)((w >> 20) & 15u);
            f32x4 *dst = reinterpret_cast<f32x4 *>(N.tbuf + mid_chunk(row, gp));
            f32x4 v = x[g];
            if (__any(valid && add)) {                            // (wave-uniform) a further edge of the same type: read-add-write
                const f32x4 old = *dst;
                if (add) v = v + old;
            }
            if (valid) *dst = v;
        }
        issue(b + 1);
        if (hdr0 & 0x100u) {                                      // last batch of its type run: S_t is complete
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4 s0 = *reinterpret_cast<const f32x4 *>(N.tbuf + mid_chunk(fj, 2 * fq));
            const f32x4 s1 = *reinterpret_cast<const f32x4 *>(N.tbuf + mid_chunk(fj, 2 * fq + 1));
            const float af[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            // rows without an edge of this type hold whatever the run before left there: their operand is zeroed by the scale
            const float scale = ((hdr1 >> fj) & 1u) ? N.sx : 0.f;
            run_mma(af, scale, (int)(hdr0 & 0xffu));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                      // (the next run's stores come after these reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    {   // root run: the row itself, pre-multiplied by max(deg, 1)
        const float af[8] = {own0.x, own0.y, own0.z, own0.w, own1.x, own1.y, own1.z, own1.w};
        run_mma(af, degf * N.sx, N.n_types);
    }
    // epilogue: mean + root + bias, LeakyReLU; rows past n are zero
    const float inv = row_ok ? N.unscale / degf : 0.f;
    const float4 bias0 = *reinterpret_cast<const float4 *>(bias + 4 * fq), bias1 = *reinterpret_cast<const float4 *>(bias + 16 + 4 * fq);
    f32x4 o0, o1;
    o0[0] = leakyf_(fmaf(d0[0], inv, bias0.x)); o0[1] = leakyf_(fmaf(d0[1], inv, bias0.y));
    o0[2] = leakyf_(fmaf(d0[2], inv, bias0.z)); o0[3] = leakyf_(fmaf(d0[3], inv, bias0.w));
    o1[0] = leakyf_(fmaf(d1[0], inv, bias1.x)); o1[1] = leakyf_(fmaf(d1[1], inv, bias1.y));
    o1[2] = leakyf_(fmaf(d1[2], inv, bias1.z)); o1[3] = leakyf_(fmaf(d1[3], inv, bias1.w));
    if (!row_ok) o0 = o1 = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(N.tbuf + mid_chunk(fj, fq)) = o0;
    *reinterpret_cast<f32x4 *>(N.tbuf + mid_chunk(fj, 4 + fq)) = o1;
    const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(A.a1);
    const uint32_t o_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 16u : kOob;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), o_rs, o_off, 0, kCpSc1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), o_rs, o_off == kOob ? kOob : o_off + 64u, 0, kCpSc1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {   // BatchNorm column sums of the tile, fp64: lane = (channel lane & 31, sum / sum of squares lane >> 5), rows in order
        const int ch = lane & 31;
        const bool sq = lane >= 32;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double v = (double)N.tbuf[mid_chunk(r, ch >> 2) + (ch & 3)];
            acc += sq ? v * v : v;
        }
        bn_acc += acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- CollConv of one 16-row tile on one wave ------------------------------------------------------------------------------------
__device__ __forceinline__ void mid_gin_tile(const MidArgs &A, int64_t tile, int layer, const float *src, float *dst, const float *gw,
                                             const float *sp, const float *st2, float *tbuf, int lane, double &bn_acc) {
    const int fj = lane & 15, fq = lane >> 4, go = lane >> 3, gp = lane & 7;
    const int64_t n = A.n;
    const __amdgpu_buffer_rsrc_t a_rs = rsrc_of(src);
    const bool use_stat = layer > 0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    // x = BatchNorm of the previous layer's pre-BN rows, folded into the sum as in gin32_aggregate_kernel (gin.hip)
    const float4 mhi = use_stat ? *reinterpret_cast<const float4 *>(st2 + 4 * gp) : zero4;
    const float4 mlo = use_stat ? *reinterpret_cast<const float4 *>(st2 + 32 + 4 * gp) : zero4;
    const float4 gv = use_stat ? *reinterpret_cast<const float4 *>(st2 + 64 + 4 * gp) : one4;
    const float4 bv = use_stat ? *reinterpret_cast<const float4 *>(st2 + 96 + 4 * gp) : zero4;
    const float one_eps = sp[kSpEps];
    // neighbourhood sums: lane (go, gp) walks the rows 8 h + go (h = 0, 1), piece gp; 8 neighbours of both rows in flight
    int beg[2], deg[2];
    float4 selfv[2], acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t row = tile * 16 + 8 * h + go;
        const bool ok = row < n;
        beg[h] = ok ? A.col_rowptr[row] : 0;
        deg[h] = ok ? A.col_rowptr[row + 1] - beg[h] : 0;
        selfv[h] = ld_sc1_f4(a_rs, ok ? (uint32_t)row * 128u + (uint32_t)gp * 16u : kOob);
        acc[h] = zero4;
    }
    constexpr int kG = 4;                                         // neighbours of each of the two rows in flight (registers: 128 per lane)
    for (int k0 = 0; __any(k0 < deg[0] || k0 < deg[1]); k0 += kG) {
        uint32_t off[2][kG];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < kG; ++k) {
                const bool in = k0 + k < deg[h];
                const int nb = A.col_nbr[in ? beg[h] + k0 + k : 0];
                off[h][k] = in ? (uint32_t)nb * 128u + (uint32_t)gp * 16u : kOob;
            }
        float4 y[2][kG];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < kG; ++k) y[h][k] = ld_sc1_f4(a_rs, off[h][k]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < kG; ++k)
                if (off[h][k] != kOob) {
                    acc[h].x += (y[h][k].x - mhi.x) - mlo.x; acc[h].y += (y[h][k].y - mhi.y) - mlo.y;
                    acc[h].z += (y[h][k].z - mhi.z) - mlo.z; acc[h].w += (y[h][k].w - mhi.w) - mlo.w;
                }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float kb = one_eps + (float)deg[h];
        f32x4 z4;
        z4[0] = fmaf(gv.x, fmaf(one_eps, (selfv[h].x - mhi.x) - mlo.x, acc[h].x), kb * bv.x);
        z4[1] = fmaf(gv.y, fmaf(one_eps, (selfv[h].y - mhi.y) - mlo.y, acc[h].y), kb * bv.y);
        z4[2] = fmaf(gv.z, fmaf(one_eps, (selfv[h].z - mhi.z) - mlo.z, acc[h].z), kb * bv.z);
        z4[3] = fmaf(gv.w, fmaf(one_eps, (selfv[h].w - mhi.w) - mlo.w, acc[h].w), kb * bv.w);
        *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(8 * h + go, gp)) = z4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // matrix layout: lane (n = fj, q = fq) holds floats 8 q .. 8 q + 7 of tile row n
    float z[8];
    {
        const f32x4 za = *reinterpret_cast<const f32x4 *>(tbuf + mid_chunk(fj, 2 * fq));
        const f32x4 zb = *reinterpret_cast<const f32x4 *>(tbuf + mid_chunk(fj, 2 * fq + 1));
        z[0] = za[0]; z[1] = za[1]; z[2] = za[2]; z[3] = za[3]; z[4] = zb[0]; z[5] = zb[1]; z[6] = zb[2]; z[7] = zb[3];
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
    f32x4 h1a = small_mma6(w1p + 0 * 64, 2 * 64, xb, bias4(0, 0));
    f32x4 h1b = small_mma6(w1p + 1 * 64, 2 * 64, xb, bias4(0, 1));
    {
        const float x[8] = {sigmoidf_(h1a[0]), sigmoidf_(h1a[1]), sigmoidf_(h1a[2]), sigmoidf_(h1a[3]),
                            sigmoidf_(h1b[0]), sigmoidf_(h1b[1]), sigmoidf_(h1b[2]), sigmoidf_(h1b[3])};
        split3_trunc(x, xb[0], xb[1], xb[2]);
    }
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
    auto sig_out = [](float v) { return 1.0f / (1.0f + expf(-v)); };     // full precision, as gin32_mlp_kernel
    const int64_t my_row = tile * 16 + fj;
    const bool row_ok = my_row < n;
    f32x4 r0, r1;
    r0[0] = leakyf_(sig_out(o0[0])); r0[1] = leakyf_(sig_out(o0[1])); r0[2] = leakyf_(sig_out(o0[2])); r0[3] = leakyf_(sig_out(o0[3]));
    r1[0] = leakyf_(sig_out(o1[0])); r1[1] = leakyf_(sig_out(o1[1])); r1[2] = leakyf_(sig_out(o1[2])); r1[3] = leakyf_(sig_out(o1[3]));
    if (!row_ok) r0 = r1 = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                              // (everybody has read z)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, fq)) = r0;
    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, 4 + fq)) = r1;
    const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(dst);
    const uint32_t o_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 16u : kOob;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r0), o_rs, o_off, 0, kCpSc1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r1), o_rs, o_off == kOob ? kOob : o_off + 64u, 0, kCpSc1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const int ch = lane & 31;
        const bool sq = lane >= 32;
        double acc2 = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double v = (double)tbuf[mid_chunk(r, ch >> 2) + (ch & 3)];
            acc2 += sq ? v * v : v;
        }
        bn_acc += acc2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// global -> LDS by DMA: `chunks` KB starting at src, round the waves [w0, w0 + nw) of the block; 1 KB per wave-level instruction
__device__ __forceinline__ void mid_dma(const float *src, float *dst_lds, int bytes, int wave, int lane, int w0, int nw) {
    if (wave < w0 || wave >= w0 + nw) return;
    const int chunks = bytes >> 10;                               // (multiples of 1 KB)
    for (int c = wave - w0; c < chunks; c += nw)
        __builtin_amdgcn_global_load_lds(src + c * 256 + lane * 4, (lds_void_t *)(dst_lds + c * 256), 16, 0, 0);
}

__global__ __launch_bounds__(kMidThreads) void forward_layers_mid_kernel(MidArgs A, SmallRunTab R) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    if (A.fault && blockIdx.x == gridDim.x - 1) return;         // (test hook: a block that never shows up)
    const int T = A.n_types, D = A.depth;
    const int64_t n = A.n;
    // LDS: NNConv image | GIN image | 16 wave tiles [16][32] (the all-reduce's two fold arrays alias them) | batches of the 8
    //      NNConv waves' tiles | BatchNorm records [2][4][32] | all-reduce totals [128] doubles | scratch words
    float *wl = lds;
    float *gw = wl + (T + 1) * kWtTypeF16;
    float *tiles = gw + kSpGinFrags * 4;
    uint32_t *ents = reinterpret_cast<uint32_t *>(tiles + kMidWaves * 512);
    float *st = reinterpret_cast<float *>(ents + kMidNnWaves * kMidEntWords);
    double *tot = reinterpret_cast<double *>(st + 256);
    float *scr = reinterpret_cast<float *>(tot + 128);            // [32]
    double *red1 = reinterpret_cast<double *>(tiles), *red2 = red1 + 16 * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tbuf = tiles + wave * 512;
    uint32_t *ebuf = ents + (wave & (kMidNnWaves - 1)) * kMidEntWords;
    const int64_t n_tiles = (n + 15) / 16;
    const int K = A.tiles_per_block;
    const int64_t tile0 = (int64_t)blockIdx.x * K;
    const int kb = (int)(n_tiles - tile0 < K ? (n_tiles - tile0 > 0 ? n_tiles - tile0 : 0) : K);   // tiles of this block
    const unsigned nblk = gridDim.x, blk = blockIdx.x;
    const size_t slot = (size_t)n * 32;
    SpinCtx spin{A.err, A.spin_budget, false};
    unsigned b_target = 0;
    const bool ent_resident = K <= kMidNnWaves;                  // one tile per NNConv wave: its batches stay in LDS for all layers
    double bn1 = 0.0, bn2 = 0.0;                                  // this wave's BatchNorm sums of the layer: lane = (channel, sum | sumsq)

    auto load_entries = [&](int64_t tile, int nb) {
        const uint32_t *src = A.ent + (size_t)tile * kMidEntWords;
        for (int w4 = lane; w4 < nb * (kMidBatchWords / 4); w4 += 64)
            reinterpret_cast<u32x4 *>(ebuf)[w4] = reinterpret_cast<const u32x4 *>(src)[w4];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto gin_items = [&](int layer) {
        if (wave >= kMidGinWaves) return;
        const float *src = layer == 0 ? A.mid : A.a2[(layer - 1) & 1];
        const float *sp = A.pack + (size_t)layer * kSpStride;
        for (int k = wave; k < kb; k += kMidGinWaves)
            mid_gin_tile(A, tile0 + k, layer, src, A.a2[layer & 1], gw, sp, st + 128, tbuf, lane, bn2);
    };

    // ---- prologue: images of layer 0, the batches of one-tile waves, the zeroed tiles; then GIN_0 (reads slot 0, no statistics)
    mid_dma(A.wimg, wl, (T + 1) * kWtTypeF16 * 4, wave, lane, 0, 8);
    mid_dma(A.pack + kSpGinW, gw, kSpGinFrags * 16, wave, lane, 8, 8);
    for (int i = lane; i < 128; i += 64) reinterpret_cast<f32x4 *>(tbuf)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nb_res = 0;
    if (ent_resident && wave < kMidNnWaves && wave < kb) {
        nb_res = __builtin_amdgcn_readfirstlane(A.tile_nb[tile0 + wave]);
        load_entries(tile0 + wave, nb_res);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gin_items(0);

    for (int layer = 0; layer < D; ++layer) {
        const float *sp = A.pack + (size_t)layer * kSpStride;
        __syncthreads();                                          // GIN_layer is through with the GIN image
        // the GIN image of the next layer (used in this layer's barrier shadow): by the waves that have no NNConv item
        if (layer + 1 < D) mid_dma(A.pack + (size_t)(layer + 1) * kSpStride + kSpGinW, gw, kSpGinFrags * 16, wave, lane, 8, 8);
        // =========================================== NNConv_layer ===========================================
        if (wave < kMidNnWaves && wave < kb) {
            MidNn N;
            N.wl = wl;
            N.tbuf = tbuf;
            N.ebuf = ebuf;
            N.h_rs = rsrc_of(A.mid + (size_t)layer * slot);
            N.n_types = T;
            {
                const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(A.bounds, 0, (2 * kMaxDepth + 8) * 4, 0x00020000);
                const unsigned hmax = __builtin_amdgcn_raw_buffer_load_b32(b_rs, (uint32_t)layer * 4u, 0, kCpSc1);
                const unsigned rmax = __builtin_amdgcn_raw_buffer_load_b32(b_rs, (uint32_t)(D + 1 + layer) * 4u, 0, kCpSc1);
                N.sx = pow2_scale_for(hmax, A.deg_log2);
                N.unscale = 1.0f / (N.sx * nnconv_weight_scale(rmax));   // (powers of two: exact)
            }
            for (int k = wave; k < kb; k += kMidNnWaves) {
                int nb = nb_res;
                if (!ent_resident) {
                    nb = __builtin_amdgcn_readfirstlane(A.tile_nb[tile0 + k]);
                    load_entries(tile0 + k, nb);
                }
                mid_nnconv_tile(N, A, tile0 + k, nb, sp + kSpBias, lane, bn1);
            }
        }
        // =========================================== [R] BatchNorm sums of the layer ===========================================
        // the wave's sums -> its tile; the block's 128 sums (waves in order) -> tagged partial row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the a1 / a2 rows of this block are written)
        reinterpret_cast<double *>(tbuf)[lane] = bn1;
        reinterpret_cast<double *>(tbuf)[64 + lane] = bn2;
        bn1 = bn2 = 0.0;
        __syncthreads();
        const unsigned tag = 1u + ((unsigned)(layer >> 1) & 1u);
        const size_t par = (size_t)(layer & 1) * nblk;
        const __amdgpu_buffer_rsrc_t p_rs = rsrc_of(A.part + par * 128), g_rs = rsrc_of(A.gpart + par * 128);
        double blocksum = 0.0;
        if (tid < 128) {
#pragma unroll
            for (int w = 0; w < kMidWaves; ++w) blocksum += reinterpret_cast<const double *>(tiles + w * 512)[tid];
            __builtin_amdgcn_raw_buffer_store_b64(mid_tag(blocksum, tag), p_rs, (blk * 128u + (uint32_t)tid) * 8u, 0, kCpSc1);
        }
        __syncthreads();                                          // (the tiles are free: the fold arrays alias them)
        // the NNConv image of the next layer: the last waves bring it in while the sums travel (everybody is through with this one)
        if (layer + 1 < D) mid_dma(A.wimg + (size_t)(layer + 1) * (T + 1) * kWtTypeF16, wl, (T + 1) * kWtTypeF16 * 4, wave, lane, 12, 4);
        {
            const int jp = tid & 63, r = tid >> 6;
            // level 1: the 16 blocks of this block's group, rows in order
            const unsigned gbase = blk & ~15u;
            double x0, x1;
            mid_poll_pair(p_rs, gbase + r < nblk ? (int64_t)(gbase + r) : -1, jp, tag, x0, x1, spin);
            red1[r * 128 + 2 * jp] = x0;
            red1[r * 128 + 2 * jp + 1] = x1;
            __syncthreads();
            if (tid < 128) {
                double s = 0.0;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) s += red1[rr * 128 + tid];
                __builtin_amdgcn_raw_buffer_store_b64(mid_tag(s, tag), g_rs, (blk * 128u + (uint32_t)tid) * 8u, 0, kCpSc1);
            }
            // level 2: one copy of every group's sum (all copies carry the same bits), groups in order
            const unsigned n_groups = (nblk + 15u) >> 4;
            int64_t grow = -1;
            if ((unsigned)r < n_groups) {
                const unsigned gsize = nblk - 16u * (unsigned)r < 16u ? nblk - 16u * (unsigned)r : 16u;
                const unsigned member = (blk & 15u) < gsize ? (blk & 15u) : gsize - 1u;
                grow = (int64_t)(16u * (unsigned)r + member);
            }
            mid_poll_pair(g_rs, grow, jp, tag, x0, x1, spin);
            red2[r * 128 + 2 * jp] = x0;
            red2[r * 128 + 2 * jp + 1] = x1;
            __syncthreads();
            if (tid < 128) {
                double s = 0.0;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) s += red2[rr * 128 + tid];
                tot[tid] = s;
            }
            __syncthreads();
            if (tid < 64) {                                       // the two records, as bn_finalize_kernel writes them
                const int job = tid >> 5, ch = tid & 31;
                const double inv_n = 1.0 / (double)n;
                const double mean = tot[job * 64 + ch] * inv_n;
                double var = tot[job * 64 + 32 + ch] * inv_n - mean * mean;
                if (var < 0.0) var = 0.0;
                const float gamma = sp[(job ? kSpG2 : kSpG1) + ch], beta = sp[(job ? kSpB2 : kSpB1) + ch];
                const float mh = (float)mean;
                float *rec = st + job * 128;
                rec[ch] = mh;
                rec[32 + ch] = (float)(mean - (double)mh);
                rec[64 + ch] = (float)((double)gamma / sqrt(var + (double)A.eps));
                rec[96 + ch] = beta;
                if (blk == 0 && A.update_running) {
                    double *rs = A.runstat + (size_t)layer * 128 + job * 64;
                    rs[ch] = mean;
                    rs[32 + ch] = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
                }
            }
            __syncthreads();
        }
        // =========================================== merge (TilinGNN.py:64-71) ===========================================
        // slot layer + 1 = BN1(a1) * BN2(a2) (+ slot layer - 2), own rows; the slot's largest magnitude for the next NNConv's scale
        {
            const __amdgpu_buffer_rsrc_t a1_rs = rsrc_of(A.a1), a2_rs = rsrc_of(A.a2[layer & 1]);
            const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(A.mid + (size_t)(layer + 1) * slot);
            const __amdgpu_buffer_rsrc_t r_rs = rsrc_of(A.mid + (size_t)(layer >= 2 ? layer - 2 : 0) * slot);
            const int c4 = (tid & 7) * 4;
            const float4 m1h = *reinterpret_cast<const float4 *>(st + c4), m1l = *reinterpret_cast<const float4 *>(st + 32 + c4);
            const float4 g1 = *reinterpret_cast<const float4 *>(st + 64 + c4), b1 = *reinterpret_cast<const float4 *>(st + 96 + c4);
            const float4 m2h = *reinterpret_cast<const float4 *>(st + 128 + c4), m2l = *reinterpret_cast<const float4 *>(st + 160 + c4);
            const float4 g2 = *reinterpret_cast<const float4 *>(st + 192 + c4), b2 = *reinterpret_cast<const float4 *>(st + 224 + c4);
            float mx = 0.f;
            for (int idx = tid; idx < kb * 128; idx += kMidThreads) {
                const int64_t r = tile0 * 16 + (idx >> 3);
                const uint32_t off = r < n ? (uint32_t)r * 128u + (uint32_t)c4 * 4u : kOob;
                const float4 x1 = ld_sc1_f4(a1_rs, off), x2 = ld_sc1_f4(a2_rs, off);
                const float4 rs4 = ld_sc1_f4(r_rs, layer >= 2 ? off : kOob);
                float4 o;
                o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x) + rs4.x;
                o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y) + rs4.y;
                o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z) + rs4.z;
                o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w) + rs4.w;
                if (off != kOob) {
                    mx = absmax4(mx, o);
                    st_sc1_f4(o_rs, off, o);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
            if (lane == 0) scr[wave] = mx;
        }
        // =========================================== [B] arrive; GIN_{layer + 1} in the shadow; wait ===========================================
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        b_target += nblk;
        if (tid == 0) {
            float m = scr[0];
#pragma unroll
            for (int w = 1; w < kMidWaves; ++w) m = fmaxf(m, scr[w]);
            if (m > 0.f) atomicMax(A.bounds + layer + 1, __float_as_uint(m));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the bound is in before the arrival is counted)
            __hip_atomic_fetch_add(A.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (layer + 1 < D) {
            for (int i = lane; i < 128; i += 64) reinterpret_cast<f32x4 *>(tbuf)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // (the fold left doubles there)
            gin_items(layer + 1);
            if (tid == 0) spin_until_ge(A.ctr, b_target, spin, kSpinErrBarrier);
        }
    }
    // (no barrier behind the last merge: the final MLP is the next launch)

    if (blockIdx.x == 0 && A.update_running && __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        // running statistics of the 2 x depth BatchNorms of the layers (momentum update, num_batches_tracked); not after a
        // wait that gave up: the parked statistics are garbage then, and the host repeats the forward
        __syncthreads();
        for (int idx = tid; idx < D * 64; idx += kMidThreads) {
            const int l = idx >> 6, job = (idx >> 5) & 1, ch = idx & 31;
            const SmallRun run = R.l[l];
            float *rm = job ? run.rm2 : run.rm1, *rv = job ? run.rv2 : run.rv1;
            const double *rs = A.runstat + (size_t)l * 128 + job * 64;
            rm[ch] = (float)((1.0 - (double)A.momentum) * (double)rm[ch] + (double)A.momentum * rs[ch]);
            rv[ch] = (float)((1.0 - (double)A.momentum) * (double)rv[ch] + (double)A.momentum * rs[32 + ch]);
            if (ch == 0) *(job ? run.nbt2 : run.nbt1) += 1;
        }
    }
}

static size_t mid_lds_bytes(int n_types) {
    return ((size_t)(n_types + 1) * kWtTypeF16 + (size_t)kSpGinFrags * 4 + (size_t)kMidWaves * 512 + (size_t)kMidNnWaves * kMidEntWords + 256 +
            256 + 32) * sizeof(float);
}
constexpr size_t kMidMaxLds = 160 * 1024 - 256;

static std::atomic<int64_t> g_mid_limit{65536};
static std::atomic<int> g_mid_blocks_cap{0};                      // experiments: upper bound of the grid (0 = one block per CU)

// > 0: tiles per block of the persistent layer loop for this layout; 0: not eligible (the general schedule runs)
int mid_layout_tiles_per_block(const tgnn_model_dims *d, const tgnn_graph *g, int64_t n_nodes, int *blocks_out) {
    const int64_t limit = g_mid_limit.load(std::memory_order_relaxed);
    if (n_nodes <= 4096 || n_nodes > limit || n_nodes > 65536 || !g->nn_mid_tile_nb || !g->nn_mid_ent) return 0;   // (up to 4 096: forward_small.hip)
    if (d->network_width != 32 || d->network_depth < 1 || d->network_depth > kMaxDepth) return 0;
    if (g->nn_max_in_degree < 1 || g->n_types + 1 > kMidTileBatches) return 0;
    if (mid_lds_bytes(g->n_types) > kMidMaxLds) return 0;
    static std::atomic<int> capacity[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int cap = capacity[dev].load(std::memory_order_acquire);
    if (cap == 0) {
        static LdsOptIn site;
        int per_cu = 0;
        if (opt_in_dynamic_lds(forward_layers_mid_kernel, (int)kMidMaxLds, site) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, forward_layers_mid_kernel, kMidThreads, kMidMaxLds) != hipSuccess)
            return 0;
        cap = per_cu > 0 ? device_cus() : -1;                     // one 16-wave block with the whole LDS per CU
        capacity[dev].store(cap, std::memory_order_release);
    }
    if (cap <= 0) return 0;
    int max_blocks = cap;
    if (const int dbg = g_mid_blocks_cap.load(std::memory_order_relaxed); dbg > 0 && dbg < max_blocks) max_blocks = dbg;
    const int64_t n_tiles = (n_nodes + 15) / 16;
    const int64_t k = (n_tiles + max_blocks - 1) / max_blocks;
    if (k > kMidMaxTilesPerBlock) return 0;
    *blocks_out = (int)((n_tiles + k - 1) / k);
    return (int)k;
}

size_t mid_part_doubles() { return (size_t)2 * 2 * 256 * 128; }   // part + gpart: [2 parities][<= 256 blocks][128]

int launch_forward_mid(const tgnn_model_dims *d, const Params &P, float *mid, float *a1, float *a2_0, float *a2_1, const float *wimg,
                       const float *pack, const tgnn_graph *graph, double *part, double *runstat, unsigned *ctr, unsigned *bounds,
                       int64_t n, int tiles_per_block, int blocks, int update_running, float eps, float momentum, hipStream_t s) {
    const int depth = d->network_depth;
    MidArgs A{};
    A.mid = mid;
    A.a1 = a1;
    A.a2[0] = a2_0;
    A.a2[1] = a2_1;
    A.wimg = wimg;
    A.pack = pack;
    A.adj_rowptr = graph->adj_rowptr;
    A.col_rowptr = graph->col_rowptr;
    A.col_nbr = graph->col_src;
    A.tile_nb = graph->nn_mid_tile_nb;
    A.ent = graph->nn_mid_ent;
    A.part = part;
    A.gpart = part + (size_t)2 * 256 * 128;
    A.runstat = runstat;
    A.ctr = ctr;
    A.bounds = bounds;
    A.err = spin_error_word();
    A.spin_budget = spin_budget_ticks();
    A.fault = spin_take_fault();
    if (!A.err) {
        set_error("tgnn_forward: the spin-error word of the device could not be allocated");
        return TGNN_ERR_LAUNCH;
    }
    A.n = n;
    A.n_types = graph->n_types;
    A.depth = depth;
    A.update_running = update_running;
    A.tiles_per_block = tiles_per_block;
    int deg_log2 = 0;
    while ((1 << deg_log2) < graph->nn_max_in_degree) ++deg_log2;
    A.deg_log2 = deg_log2;
    A.eps = eps;
    A.momentum = momentum;
    SmallRunTab R{};
    for (int i = 0; i < depth; ++i) {
        const BnPtrs b1 = P.bn(P.layer(i) + 8), b2 = P.bn(P.layer(i) + 20);
        R.l[i] = SmallRun{b1.rm, b1.rv, b1.nbt, b2.rm, b2.rv, b2.nbt};
    }
    TGNN_CHECK_ARG(blocks >= 1 && blocks <= 256, "blocks of the persistent layer loop");
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(forward_layers_mid_kernel, (int)kMidMaxLds, site));
    // the tagged rows of both parities carry no valid tag before the launch (tag 0 is never used)
    TGNN_CHECK_HIP(hipMemsetAsync(part, 0, mid_part_doubles() * sizeof(double), s));
    TGNN_CHECK_HIP(hipMemsetAsync(ctr, 0, sizeof(unsigned), s));
    struct Ctx { MidArgs *A; SmallRunTab *R; int blocks; size_t lds; } ctx{&A, &R, blocks, mid_lds_bytes(graph->n_types)};
    const int rc = spin_kernel_chain(s, [](void *c, hipStream_t st) {
        Ctx *x = static_cast<Ctx *>(c);
        forward_layers_mid_kernel<<<dim3(x->blocks), dim3(kMidThreads), x->lds, st>>>(*x->A, *x->R);
    }, &ctx, blocks);
    if (rc != TGNN_OK) return rc;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The NNConv batches of the mid-size kernel, from the type-column structure (graph_prep.hip: column (t, r) = the r-th in-edge of
// type t of each of the tile's 16 rows).  Per tile and type run: the run's entries -- column by column, i.e. a row's edges in CSR
// order -- packed 8 to a gather instruction, 4 instructions to a batch, never two entries of one row in one instruction (a row's
// second edge of a type is a read-add-write of the slot its first edge stored); a run of more than 32 entries continues in
// further batches, the last one carries the `last` flag and the mask of rows that have an edge of the type.
//   batch = [type | last << 8, row mask, 0, 0 | 32 entry words: slot o (0..7) of instruction g (0..3) at 4 + 4 o + g]
//   entry = source row | destination row << 20 | add << 24 | valid << 31
// One block of one wave per tile: the columns come in coalesced, thread 0 packs in LDS, the batches go out coalesced.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kMidBuildMaxCols = 128;

__global__ __launch_bounds__(64) void mid_entries_kernel(const int *__restrict__ tile_col_ptr, const int *__restrict__ col_meta,
                                                         const int *__restrict__ col_src, int64_t n_tiles,
                                                         const int *__restrict__ cols_built, int *__restrict__ tile_nb,
                                                         uint32_t *__restrict__ ent, int *__restrict__ result) {
    __shared__ int s_src[kMidBuildMaxCols * 16];
    __shared__ int s_meta[kMidBuildMaxCols];
    __shared__ uint32_t s_out[kMidEntWords];
    __shared__ int s_nb;
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if (cols_built && *cols_built == 0) {                         // no column structure (too many edge types): nothing to pack
        if (tid == 0) {
            tile_nb[tile] = 0;
            if (tile == 0) result[1] = 1;
        }
        return;
    }
    const int c0 = tile_col_ptr[tile], c1 = tile_col_ptr[tile + 1];
    const int nc = c1 - c0 - 1;                                   // edge columns (the last column of a tile is the root column)
    if (nc > kMidBuildMaxCols) {
        if (tid == 0) {
            tile_nb[tile] = 0;
            result[1] = 1;                                        // overflow: the layout is not for this kernel
        }
        return;
    }
    for (int i = tid; i < nc * 16; i += 64) s_src[i] = col_src[(int64_t)c0 * 16 + i];
    for (int i = tid; i < nc; i += 64) s_meta[i] = col_meta[c0 + i];
    for (int i = tid; i < kMidEntWords; i += 64) s_out[i] = 0u;
    __syncthreads();
    if (tid == 0) {
        int nb = 0, pos = 0, type = -1;
        unsigned mask = 0;
        int rowgrp[16];
        bool open = false, overflow = false;
        auto close = [&](bool last) {
            if (nb < kMidTileBatches) {
                s_out[nb * kMidBatchWords] = (unsigned)type | (last ? 0x100u : 0u);
                s_out[nb * kMidBatchWords + 1] = mask;
            } else {
                overflow = true;
            }
            ++nb;
            pos = 0;
            for (int i = 0; i < 16; ++i) rowgrp[i] = -1;
        };
        for (int i = 0; i < 16; ++i) rowgrp[i] = -1;
        for (int k = 0; k < nc; ++k) {
            const int m = s_meta[k];
            if (m & kMidColFirst) {
                if (open) close(true);
                type = m & 0xff;
                mask = 0;
                open = true;
            }
            const bool add = !(m & kMidColFirst);
            for (int i = 0; i < 16; ++i) {
                const int sv = s_src[k * 16 + i];
                if (sv < 0) continue;
                if (rowgrp[i] == (pos >> 3)) pos = ((pos >> 3) + 1) * 8;      // this row already has an entry in the instruction
                if (pos >= 32) close(false);
                if (nb < kMidTileBatches)
                    s_out[nb * kMidBatchWords + 4 + 4 * (pos & 7) + (pos >> 3)] =
                        ((unsigned)sv & 0xfffffu) | ((unsigned)i << 20) | (add ? 1u << 24 : 0u) | 0x80000000u;
                rowgrp[i] = pos >> 3;
                ++pos;
                mask |= 1u << i;
            }
        }
        if (open) close(true);
        if (overflow) {
            nb = 0;
            result[1] = 1;
        }
        if (nb > result[0]) atomicMax(result, nb);
        s_nb = nb;
        tile_nb[tile] = nb;
    }
    __syncthreads();
    const int words = s_nb * kMidBatchWords;
    uint32_t *dst = ent + (size_t)tile * kMidEntWords;
    for (int i = tid; i < words; i += 64) dst[i] = s_out[i];
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int64_t tgnn_mid_entries_words(int64_t n_nodes) { return ((n_nodes + 15) / 16) * (int64_t)kMidEntWords; }

extern "C" int tgnn_mid_entries_build(const int32_t *tile_col_ptr, const int32_t *col_meta, const int32_t *col_src, int64_t n_nodes,
                                      const int32_t *cols_built_dev, int32_t *tile_nb, uint32_t *ent, int32_t *result,
                                      tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_nodes < (1 << 20), "n_nodes (source rows are 20-bit)");
    TGNN_CHECK_ARG(tile_col_ptr && col_meta && col_src && tile_nb && ent && result, "null pointer");
    const int64_t n_tiles = (n_nodes + 15) / 16;
    mid_entries_kernel<<<(unsigned)n_tiles, 64, 0, static_cast<hipStream_t>(stream)>>>(tile_col_ptr, col_meta, col_src, n_tiles,
                                                                                       cols_built_dev, tile_nb, ent, result);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" void tgnn_set_mid_layout_limit(int64_t n_nodes) { g_mid_limit.store(n_nodes < 0 ? 0 : n_nodes); }
extern "C" int64_t tgnn_get_mid_layout_limit(void) { return g_mid_limit.load(); }
extern "C" int64_t tgnn_mid_layout_max_nodes(void) { return 65536; }
extern "C" void tgnn_debug_set_mid_blocks(int32_t blocks) { g_mid_blocks_cap.store(blocks < 0 ? 0 : blocks); }
