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

constexpr int kMidThreads = 512, kMidWaves = 8;   // 2 waves per SIMD, 256 registers a lane: a wave hides its own latency (deep gather queues)
constexpr int kMidBatchWords = TGNN_MID_BATCH_WORDS, kMidTileBatches = TGNN_MID_TILE_BATCHES;
constexpr int kMidEntWords = kMidBatchWords * kMidTileBatches;     // 864 words per tile
constexpr uint32_t kMidEmptyEntry = 0x01000000u | (16u << 25);      // no source row, destination = the spare row
constexpr int kMidPf = 4;                                          // gather batches in flight per wave
constexpr int kMidEntLds = kMidEntWords + 2 * kMidPf * kMidBatchWords;   // a wave's batches in LDS + the empty ones behind them (the unrolled loop runs up to kPf - 1 past the end and issues kPf ahead)
constexpr int kMidMaxTilesPerBlock = 16;
constexpr int kMidColFirst = 1 << 8;

#ifdef TGNN_MID_TIMING
// phase timers (scratch builds only): wall_clock64 ticks (100 MHz) summed over the layers; per block [0..15] thread 0's phases,
// [16 + w] wave w's time inside its NNConv items, [32 + w] inside its GIN items
__device__ unsigned long long g_mid_timing[256 * 64];
#define TGNN_MT(slot) { const unsigned long long now_ = wall_clock64(); if (tid == 0) tacc[slot] += now_ - tlast; tlast = now_; }
#define TGNN_MI(slot) if (tid == 0 && blockIdx.x < 256) g_mid_timing[blockIdx.x * 64 + 48 + (slot)] = wall_clock64();
#else
#define TGNN_MT(slot)
#define TGNN_MI(slot)
#endif

struct MidArgs {
    float *mid;                  // skip buffer [depth + 1][n][32]; slot 0 filled by the init MLP
    float *a1;                   // [n][32] GraphConv pre-BatchNorm rows of the layer
    float *a2[2];                // CollConv pre-BatchNorm rows, two-deep
    const float *wimg;           // NNConv fp16-pair weight images [depth][(T + 1)][kWtTypeF16]
    const unsigned *weights_done;   // NULL, or: blocks of the edge-weight kernel that have finished (this kernel may start before them)
    unsigned weights_target;
    const float *pack;           // [depth][kSpStride] parameter vectors + GIN MFMA images (small_pack_kernel)
    const int *adj_rowptr;       // in-degrees of the adjacency set
    const int *col_rowptr, *col_nbr;
    const int *tile_nb;          // batches per tile
    const uint32_t *ent;         // [tiles][kMidEntWords]
    double *part, *gpart;        // [2][blocks][128] tagged partial rows / group sums (zeroed before the launch)
    double *runstat;             // [depth][128] parked batch statistics for the running buffers
    unsigned *ctr;               // barrier counter (zero before the launch)
    // the init MLP (TilinGNN.py:54) in this kernel's prologue (x != NULL), else slot 0 arrives filled by the launches in front
    const float *x;              // node features [n][fx], fx <= 8
    const float *iw0, *ib0, *ig0, *ibt0;        // Linear 0 [32][fx], bias; BatchNorm 0 weight, bias
    const float *i1img, *ib1, *ig1, *ibt1;      // Linear 1 as a bf16 x 3 MFMA image [2][1][3][64] x 16 B (small_pack_kernel), bias; BatchNorm 1
    float *irm0, *irv0, *irm1, *irv1;           // running buffers (update_running)
    int64_t *inbt0, *inbt1;
    int fx;
    double *tail_zero[2];        // NULL, or: the tagged rows of the final MLP's kernel behind this one (forward_tail.hip), cleared here
    unsigned tail_zero_vec;      // ... 16-byte pieces of each
    unsigned *bounds;            // [0, depth]: max |slot k| as float bits (this kernel fills 1 ..); [depth + 1 ..]: max |root_i|
    unsigned *err, *err_host;    // the device's spin-error word and its host-mapped mirror (forward_persist.h)
    unsigned long long spin_budget;
    int64_t n;
    int n_types, depth, update_running, tiles_per_block, deg_log2, fault, nn_split;
    float eps, momentum;
    const int *verdict;          // NULL, or: non-zero = the batches are not usable (tgnn_graph.nn_mid_verdict): every block leaves at once
};

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

// The wave's share of a tile's type runs (all of them, or every W-th when W waves share the tile; the root run belongs to share 0):
// d0 / d1 = its partial D^T tiles (channels 4 fq + r and 16 + 4 fq + r of row fj), before the 1 / deg.
__device__ __forceinline__ void mid_nnconv_partial(const MidNn &N, const MidArgs &A, int64_t tile, int nb, bool with_root, int lane,
                                                   f32x4 &d0, f32x4 &d1) {
    const int fj = lane & 15, fq = lane >> 4, go = lane >> 3, gp = lane & 7;
    const uint32_t gp16 = (uint32_t)gp * 16u;
    // the tile's own rows in the matrix layout (root run): requested first ...
    const int64_t my_row = tile * 16 + fj;
    const bool row_ok = my_row < A.n && with_root;
    const uint32_t own_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 32u : kOob;
    const float4 own0 = ld_sc1_f4(N.h_rs, own_off), own1 = ld_sc1_f4(N.h_rs, own_off == kOob ? kOob : own_off + 16u);
    int deg_i = 0;
    if (row_ok) deg_i = A.adj_rowptr[my_row + 1] - A.adj_rowptr[my_row];
    const float degf = row_ok ? (float)(deg_i > 0 ? deg_i : 1) : 0.f;

    d0 = f32x4{0.f, 0.f, 0.f, 0.f};
    d1 = d0;
    constexpr int kPl = kWtPlane / 4;                            // 16-byte fragments per plane
    auto run_mma = [&](const float (&af)[8], float scale, int t) {
        const f16x8 *wp = reinterpret_cast<const f16x8 *>(N.wl + t * kWtTypeF16) + lane;
        const f16x8 h0 = wp[0], h1 = wp[64], l0 = wp[kPl], l1 = wp[kPl + 64];
        f16x8 xh, xl;
        split2_f16(af, scale, xh, xl);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, xh, d0, 0, 0, 0);   // lo . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, xh, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xl, d0, 0, 0, 0);   // hi . lo
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xl, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xh, d0, 0, 0, 0);   // hi . hi
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xh, d1, 0, 0, 0);
    };

    // Batches: the gathers of kPf batches are in flight while one is scattered into the tile and multiplied -- a wave works on its
    // tile alone, so what hides the latency of a gather is its own queue.  Register slots are static (the loop is unrolled by
    // kPf); nothing is loaded behind a branch (hipcc would drain the queue at the join): past the last batch the offsets are out
    // of range and fetch nothing.  A batch's entry words and header travel in registers from the issue to the scatter.
    constexpr int kPf = kMidPf;
    f32x4 x[kPf][4];
    u32x4 ews[kPf];
    uint32_t hdr0[kPf], hdr1[kPf];
    // entry word: source row (bits 0-23; 0x1000000 = none: shifted by 7 that is an offset out of the window, nothing is fetched) |
    // destination row << 25 (16 = the tile's spare row, for empty slots) | add << 30 -- one instruction each for the gather offset,
    // the row and the LDS address.  The buffer holds kPf empty batches behind the tile's last one: no end test anywhere.
    auto issue = [&](f32x4 (&xs)[4], u32x4 &ew, uint32_t &h0, uint32_t &h1, int b) {
        const uint32_t *bp = N.ebuf + b * kMidBatchWords;
        ew = *reinterpret_cast<const u32x4 *>(bp + 4 + go * 4);
        const u32x2 hh = *reinterpret_cast<const u32x2 *>(bp);
        h0 = __builtin_amdgcn_readfirstlane(hh[0]);
        h1 = __builtin_amdgcn_readfirstlane(hh[1]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            xs[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(N.h_rs, (ew[g] << 7) | gp16, 0, kCpSc1));
    };
    auto consume = [&](const f32x4 (&xs)[4], const u32x4 &ew, uint32_t h0, uint32_t h1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w = ew[g];
            const bool add = ((w >> 30) & 1u) != 0u;
            f32x4 *dst = reinterpret_cast<f32x4 *>(N.tbuf + ((w >> 25) & 31u) * kMidRowFloats + 4 * gp);
            f32x4 v = xs[g];
            if (__any(add)) {                                     // (wave-uniform) a further edge of the same type: read-add-write
                const f32x4 old = *dst;
                if (add) v = v + old;
            }
            *dst = v;
        }
        if (h0 & 0x100u) {                                        // last batch of its type run: S_t is complete
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4 s0 = *reinterpret_cast<const f32x4 *>(N.tbuf + mid_chunk(fj, 2 * fq));
            const f32x4 s1 = *reinterpret_cast<const f32x4 *>(N.tbuf + mid_chunk(fj, 2 * fq + 1));
            const float af[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            // rows without an edge of this type hold whatever the run before left there: their operand is zeroed by the scale
            const float scale = ((h1 >> fj) & 1u) ? N.sx : 0.f;
            run_mma(af, scale, (int)(h0 & 0xffu));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                      // (the next run's stores come after these reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
#pragma unroll
    for (int u = 0; u < kPf; ++u) issue(x[u], ews[u], hdr0[u], hdr1[u], u);
    if (with_root) {   // ... root run first (its rows were requested ahead of the batches): the row itself times max(deg, 1)
        const float af[8] = {own0.x, own0.y, own0.z, own0.w, own1.x, own1.y, own1.z, own1.w};
        run_mma(af, degf * N.sx, N.n_types);
    }
    for (int b0 = 0; b0 < nb; b0 += kPf) {                        // (batches nb .. are empty ones: harmless to scatter)
#pragma unroll
        for (int u = 0; u < kPf; ++u) {
            consume(x[u], ews[u], hdr0[u], hdr1[u]);
            issue(x[u], ews[u], hdr0[u], hdr1[u], b0 + u + kPf);
        }
    }
}

// mean + root + bias, LeakyReLU of a finished tile (d0 / d1: the sums over all type runs and the root run); rows past n are zero;
// the rows go to a1, their BatchNorm column sums (fp64) into bn_acc
__device__ __forceinline__ void mid_nnconv_finish(const MidNn &N, const MidArgs &A, int64_t tile, const float *bias, int lane,
                                                  const f32x4 &d0, const f32x4 &d1, double &bn_acc) {
    const int fj = lane & 15, fq = lane >> 4;
    const int64_t my_row = tile * 16 + fj;
    const bool row_ok = my_row < A.n;
    int deg_i = 0;
    if (row_ok) deg_i = A.adj_rowptr[my_row + 1] - A.adj_rowptr[my_row];
    const float degf = (float)(deg_i > 0 ? deg_i : 1);
    const float inv = row_ok ? N.unscale / degf : 0.f;
    const float4 bias0 = *reinterpret_cast<const float4 *>(bias + 4 * fq), bias1 = *reinterpret_cast<const float4 *>(bias + 16 + 4 * fq);
    f32x4 o0, o1;
    o0[0] = leakyf_(fmaf(d0[0], inv, bias0.x)); o0[1] = leakyf_(fmaf(d0[1], inv, bias0.y));
    o0[2] = leakyf_(fmaf(d0[2], inv, bias0.z)); o0[3] = leakyf_(fmaf(d0[3], inv, bias0.w));
    o1[0] = leakyf_(fmaf(d1[0], inv, bias1.x)); o1[1] = leakyf_(fmaf(d1[1], inv, bias1.y));
    o1[2] = leakyf_(fmaf(d1[2], inv, bias1.z)); o1[3] = leakyf_(fmaf(d1[3], inv, bias1.w));
    if (!row_ok) o0 = o1 = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

// global -> LDS by DMA: `bytes` (a multiple of 1 KB) starting at src, shared out round the block's waves; 1 KB per wave-level
// instruction, no register in between
// Grid all-reduce of ONE BatchNorm's 64 column sums outside the layer loop (the init MLP in the prologue): the layer loop's scheme --
// tagged rows, two levels of 16, fixed order -- on the rows of parity `par` with a tag the layers never use.  bn: lane = (channel
// lane & 31, sum | sum of squares) of this wave; the totals land in tot [0, 64).  (red aliases the waves' tiles.)
constexpr unsigned kMidTagInit = 3u;
__device__ __forceinline__ void mid_allreduce_init(double *part, double *gpart, double bn, int par, double *bnred, double *red, double *tot,
                                                   SpinCtx &spin, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned nblk = gridDim.x, blk = blockIdx.x;
    bnred[wave * 128 + lane] = bn;
    bnred[wave * 128 + 64 + lane] = 0.0;
    __syncthreads();
    const size_t po = (size_t)par * nblk * 128;
    const __amdgpu_buffer_rsrc_t p_rs = rsrc_of(part + po), g_rs = rsrc_of(gpart + po);
    if (tid < 128) {
        double blocksum = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) blocksum += bnred[w * 128 + tid];
        __builtin_amdgcn_raw_buffer_store_b64(mid_tag(blocksum, kMidTagInit), p_rs, (blk * 128u + (uint32_t)tid) * 8u, 0, kCpSc1);
    }
    __syncthreads();
    const int jp = tid & 63, r = tid >> 6;
    const unsigned gbase = blk & ~15u;
    double x0, x1, y0, y1;
    mid_poll_pair(p_rs, gbase + r < nblk ? (int64_t)(gbase + r) : -1, jp, kMidTagInit, x0, x1, spin);
    mid_poll_pair(p_rs, gbase + r + 8 < nblk ? (int64_t)(gbase + r + 8) : -1, jp, kMidTagInit, y0, y1, spin);
    red[r * 128 + 2 * jp] = x0;
    red[r * 128 + 2 * jp + 1] = x1;
    red[(r + 8) * 128 + 2 * jp] = y0;
    red[(r + 8) * 128 + 2 * jp + 1] = y1;
    __syncthreads();
    if (tid < 128) {
        double s = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s += red[rr * 128 + tid];
        __builtin_amdgcn_raw_buffer_store_b64(mid_tag(s, kMidTagInit), g_rs, (blk * 128u + (uint32_t)tid) * 8u, 0, kCpSc1);
    }
    const unsigned n_groups = (nblk + 15u) >> 4;
    auto group_row = [&](unsigned g) -> int64_t {
        if (g >= n_groups) return -1;
        const unsigned gsize = nblk - 16u * g < 16u ? nblk - 16u * g : 16u;
        const unsigned member = (blk & 15u) < gsize ? (blk & 15u) : gsize - 1u;
        return (int64_t)(16u * g + member);
    };
    mid_poll_pair(g_rs, group_row((unsigned)r), jp, kMidTagInit, x0, x1, spin);
    mid_poll_pair(g_rs, group_row((unsigned)r + 8u), jp, kMidTagInit, y0, y1, spin);
    __syncthreads();                                              // (level 1's sums have been read)
    red[r * 128 + 2 * jp] = x0;
    red[r * 128 + 2 * jp + 1] = x1;
    red[(r + 8) * 128 + 2 * jp] = y0;
    red[(r + 8) * 128 + 2 * jp + 1] = y1;
    __syncthreads();
    if (tid < 128) {
        double s = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s += red[rr * 128 + tid];
        tot[tid] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ void mid_dma(const float *src, float *dst_lds, int bytes, int wave, int lane) {
    const int chunks = bytes >> 10;
    for (int c = wave; c < chunks; c += kMidWaves)
        __builtin_amdgcn_global_load_lds(src + c * 256 + lane * 4, (lds_void_t *)(dst_lds + c * 256), 16, 0, 0);
}

__global__ __launch_bounds__(kMidThreads) void forward_layers_mid_kernel(MidArgs A, SmallRunTab R) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    if (A.fault && blockIdx.x == gridDim.x - 1) return;         // (test hook: a block that never shows up)
    if (A.verdict && *A.verdict != 0) return;                   // (uniform over the grid: nobody waits at a barrier)
    const int T = A.n_types, D = A.depth;
    const int64_t n = A.n;
    // LDS: NNConv image | GIN image | 8 wave tiles [16][32] (the all-reduce's fold array [16][128] doubles aliases them) | batches of
    //      the waves' tiles | BatchNorm records [2][4][32] | all-reduce totals [128] doubles | scratch words | the waves' BatchNorm sums
    float *wl = lds;
    float *gw = wl + (T + 1) * kWtTypeF16;
    float *tiles = gw + kSpGinFrags * 4;
    uint32_t *ents = reinterpret_cast<uint32_t *>(tiles + kMidWaves * kMidTileFloats);
    float *st = reinterpret_cast<float *>(ents + kMidWaves * kMidEntLds);
    double *tot = reinterpret_cast<double *>(st + 256);
    float *scr = reinterpret_cast<float *>(tot + 128);            // [32]
    double *bnred = reinterpret_cast<double *>(scr + 32);         // [8 waves][128]: the waves' BatchNorm sums of the layer (NOT in the tiles:
                                                                  //  a share's partial product may still be read there by share 0)
    double *red = reinterpret_cast<double *>(tiles);
    static_assert(kMidWaves * kMidTileFloats * 4 >= 16 * 128 * 8, "the fold array fits the wave tiles");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tbuf = tiles + wave * kMidTileFloats;
    uint32_t *ebuf = ents + wave * kMidEntLds;
    const int64_t n_tiles = (n + 15) / 16;
    const int K = A.tiles_per_block;
    const int64_t tile0 = (int64_t)blockIdx.x * K;
    const int kb = (int)(n_tiles - tile0 < K ? (n_tiles - tile0 > 0 ? n_tiles - tile0 : 0) : K);   // tiles of this block
    const unsigned nblk = gridDim.x, blk = blockIdx.x;
    const size_t slot = (size_t)n * 32;
    SpinCtx spin{A.err, A.spin_budget, false, A.err_host};
    unsigned b_target = 0;
    const bool ent_resident = K <= kMidWaves;                    // one tile per wave: its batches stay in LDS for all layers
    double bn1 = 0.0, bn2 = 0.0;                                  // this wave's BatchNorm sums of the layer: lane = (channel, sum | sumsq)
#ifdef TGNN_MID_TIMING
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64(), t_nn = 0, t_gin = 0;
#endif

    // W waves share a tile's type runs when the block has fewer tiles than waves (W = 8 / K rounded down to a power of two): run r
    // belongs to share r % W; the shares' partial products meet through the tiles.  W == 1: a wave owns whole tiles (k, k + 8, ..)
    const int W = A.nn_split, my_part = wave & (W - 1);
    const int my_k = W > 1 ? wave / W : wave;                     // W > 1: THE tile of this wave (if < kb)
    // the batches of `tile` that belong to share `part` of W, compacted into this wave's buffer; returns their number
    auto load_entries = [&](int64_t tile, int part) -> int {
        const int nb_all = __builtin_amdgcn_readfirstlane(A.tile_nb[tile]);
        const uint32_t *src = A.ent + (size_t)tile * kMidEntWords;
        int cnt = nb_all;
        if (W == 1) {
            for (int w4 = lane; w4 < nb_all * (kMidBatchWords / 4); w4 += 64)
                reinterpret_cast<u32x4 *>(ebuf)[w4] = reinterpret_cast<const u32x4 *>(src)[w4];
        } else {
            // lane b looks at batch b's header: its run = the number of `last` flags in front of it
            const uint32_t h = lane < nb_all ? src[lane * kMidBatchWords] : 0u;
            const unsigned long long lastm = __ballot((h & 0x100u) != 0u);
            const int run = __popcll(lastm & ((1ull << lane) - 1ull));
            const unsigned long long minem = __ballot(lane < nb_all && (run & (W - 1)) == part);
            cnt = 0;
            for (unsigned long long m = minem; m; m &= m - 1ull, ++cnt) {
                const int b = __builtin_ctzll(m);
                if (lane < kMidBatchWords) ebuf[cnt * kMidBatchWords + lane] = src[b * kMidBatchWords + lane];
            }
        }
        // the empty batches behind the last one: header 0 (no `last` flag), every slot = no source, spare row
        for (int w = lane; w < 2 * kMidPf * kMidBatchWords; w += 64)
            ebuf[cnt * kMidBatchWords + w] = (w % kMidBatchWords) < 4 ? 0u : kMidEmptyEntry;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return cnt;
    };
    auto gin_items = [&](int layer) {
        const float *src = layer == 0 ? A.mid : A.a2[(layer - 1) & 1];
        const float *sp = A.pack + (size_t)layer * kSpStride;
#ifdef TGNN_MID_TIMING
        const unsigned long long tg0 = wall_clock64();
#endif
        for (int k = wave; k < kb; k += kMidWaves)
            gin_tile<kCpSc1>(GinGraph{A.col_rowptr, A.col_nbr, A.n}, tile0 + k, layer > 0, src, A.a2[layer & 1], gw, sp, st + 128, tbuf, lane, bn2);
#ifdef TGNN_MID_TIMING
        t_gin += wall_clock64() - tg0;
#endif
    };

    // ---- prologue: images of layer 0, the batches of one-tile waves, the zeroed tiles; then GIN_0 (reads slot 0, no statistics)
    if (A.tail_zero[0])
        for (unsigned i = blockIdx.x * kMidThreads + tid; i < 2u * A.tail_zero_vec; i += gridDim.x * kMidThreads)
            reinterpret_cast<u32x4 *>(A.tail_zero[i >= A.tail_zero_vec])[i >= A.tail_zero_vec ? i - A.tail_zero_vec : i] = u32x4{0u, 0u, 0u, 0u};
    if (A.x) {
        // =========================================== init MLP (TilinGNN.py:54) ===========================================
        // Linear(fx, 32) + LeakyReLU + BatchNorm, Linear(32, 32) + LeakyReLU + BatchNorm -> slot 0, own rows: 5 launches of the
        // general schedule (47 us in front of this kernel at 10 000 nodes: profiles/r05_mid_trace_10000.txt) as two grid all-reduces
        // and a barrier.  A wave's tiles are those it owns in the layer loop (W > 1: the first wave of a tile's shares); lane
        // (fj, fq) computes channels 8 fq .. + 7 of row fj of Linear 0 -- which IS its piece of Linear 1's matrix operand.
        constexpr int kInitTiles = kMidMaxTilesPerBlock / kMidWaves;
        const int fj = lane & 15, fq = lane >> 4, ch = lane & 31;
        const bool sq = lane >= 32;
        int64_t itile[kInitTiles];
        bool iok[kInitTiles];
#pragma unroll
        for (int u = 0; u < kInitTiles; ++u) {
            const int k = W == 1 ? wave + kMidWaves * u : (u == 0 && my_part == 0 ? my_k : K);
            iok[u] = k < kb;
            itile[u] = tile0 + k;
        }
        TGNN_MI(0)
        float a0[kInitTiles][8];
        double bn = 0.0;
#pragma unroll
        for (int u = 0; u < kInitTiles; ++u) {
            if (!iok[u]) continue;                                // (uniform per wave)
            const int64_t row = itile[u] * 16 + fj;
            float xin[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xin[k] = (k < A.fx && row < n) ? A.x[row * A.fx + k] : 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float acc = A.ib0[8 * fq + c];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < A.fx) acc = fmaf(xin[k], A.iw0[(8 * fq + c) * A.fx + k], acc);
                a0[u][c] = leakyf_(acc);
            }
            // column sums over the tile's valid rows, through the wave's tile: lane = (channel, sum | sum of squares)
            *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, 2 * fq)) = f32x4{a0[u][0], a0[u][1], a0[u][2], a0[u][3]};
            *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, 2 * fq + 1)) = f32x4{a0[u][4], a0[u][5], a0[u][6], a0[u][7]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int r = 0; r < 16; ++r)
                if (itile[u] * 16 + r < n) {
                    const double v = (double)tbuf[r * kMidRowFloats + ch];
                    bn += sq ? v * v : v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto init_record = [&](const float *gamma, const float *beta, float *rm, float *rv, int64_t *nbt) {
            if (tid < 32) {
                const double inv_n = 1.0 / (double)n;
                const double mean = tot[tid] * inv_n;
                double var = tot[32 + tid] * inv_n - mean * mean;
                if (var < 0.0) var = 0.0;
                const float mh = (float)mean;
                st[tid] = mh;
                st[32 + tid] = (float)(mean - (double)mh);
                st[64 + tid] = (float)((double)gamma[tid] / sqrt(var + (double)A.eps));
                st[96 + tid] = beta[tid];
                if (blk == 0 && A.update_running && __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
                    rm[tid] = (float)((1.0 - (double)A.momentum) * (double)rm[tid] + (double)A.momentum * mean);
                    rv[tid] = (float)((1.0 - (double)A.momentum) * (double)rv[tid] + (double)A.momentum * unbiased);
                    if (tid == 0) *nbt += 1;
                }
            }
            __syncthreads();
        };
        TGNN_MI(1)
        mid_allreduce_init(A.part, A.gpart, bn, 0, bnred, red, tot, spin, tid);
        TGNN_MI(2)
        init_record(A.ig0, A.ibt0, A.irm0, A.irv0, A.inbt0);
        TGNN_MI(3)
        // Linear 1 on the matrix pipe, bf16 x 3 (small_tile_dense's order): D^T = W . X^T, X = BatchNorm 0 of the lane's own piece
        f32x4 a1v[kInitTiles][2];
        {
            const bf16x8 *wi = reinterpret_cast<const bf16x8 *>(A.i1img) + lane;
            bf16x8 wf[2][3];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[mb][pl] = wi[(mb * 3 + pl) * 64];
            bn = 0.0;
#pragma unroll
            for (int u = 0; u < kInitTiles; ++u) {
                if (!iok[u]) continue;
                const bool row_ok = itile[u] * 16 + fj < n;
                float xs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    xs[e] = row_ok ? bn_apply1(a0[u][e], st[8 * fq + e], st[32 + 8 * fq + e], st[64 + 8 * fq + e], st[96 + 8 * fq + e]) : 0.f;
                bf16x8 x3[3];
                split3_trunc(xs, x3[0], x3[1], x3[2]);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const float4 b = *reinterpret_cast<const float4 *>(A.ib1 + 16 * mb + 4 * fq);
                    f32x4 acc = f32x4{b.x, b.y, b.z, b.w};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][2], x3[0], acc, 0, 0, 0);   // lo . hi
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][0], x3[2], acc, 0, 0, 0);   // hi . lo
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][1], x3[1], acc, 0, 0, 0);   // mid . mid
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][1], x3[0], acc, 0, 0, 0);   // mid . hi
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][0], x3[1], acc, 0, 0, 0);   // hi . mid
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mb][0], x3[0], acc, 0, 0, 0);   // hi . hi
                    a1v[u][mb] = f32x4{leakyf_(acc[0]), leakyf_(acc[1]), leakyf_(acc[2]), leakyf_(acc[3])};
                    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, 4 * mb + fq)) = a1v[u][mb];     // (row fj, channels 16 mb + 4 fq .. + 3)
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int r = 0; r < 16; ++r)
                    if (itile[u] * 16 + r < n) {
                        const double v = (double)tbuf[r * kMidRowFloats + ch];
                        bn += sq ? v * v : v;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        TGNN_MI(4)
        mid_allreduce_init(A.part, A.gpart, bn, 1, bnred, red, tot, spin, tid);
        TGNN_MI(5)
        init_record(A.ig1, A.ibt1, A.irm1, A.irv1, A.inbt1);
        TGNN_MI(6)
        // slot 0 = BatchNorm 1 of the own rows; its largest magnitude for the first NNConv's scale; everybody's rows before anybody gathers
        {
            const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(A.mid);
            float mx = 0.f;
#pragma unroll
            for (int u = 0; u < kInitTiles; ++u) {
                if (!iok[u]) continue;
                const int64_t row = itile[u] * 16 + fj;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int c0 = 16 * mb + 4 * fq;
                    float4 o;
                    o.x = bn_apply1(a1v[u][mb][0], st[c0 + 0], st[32 + c0 + 0], st[64 + c0 + 0], st[96 + c0 + 0]);
                    o.y = bn_apply1(a1v[u][mb][1], st[c0 + 1], st[32 + c0 + 1], st[64 + c0 + 1], st[96 + c0 + 1]);
                    o.z = bn_apply1(a1v[u][mb][2], st[c0 + 2], st[32 + c0 + 2], st[64 + c0 + 2], st[96 + c0 + 2]);
                    o.w = bn_apply1(a1v[u][mb][3], st[c0 + 3], st[32 + c0 + 3], st[64 + c0 + 3], st[96 + c0 + 3]);
                    if (row < n) {
                        mx = absmax4(mx, o);
                        st_sc1_f4(o_rs, (uint32_t)row * 128u + (uint32_t)c0 * 4u, o);
                    }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
            if (lane == 0) scr[wave] = mx;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            b_target += nblk;
            if (tid == kMidThreads - 64) {
                float m = scr[0];
#pragma unroll
                for (int w = 1; w < kMidWaves; ++w) m = fmaxf(m, scr[w]);
                if (m > 0.f) atomicMax(A.bounds, __float_as_uint(m));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(A.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_until_ge(A.ctr, b_target, spin, kSpinErrBarrier);
            }
            __syncthreads();
            TGNN_MI(7)
        }
    }
    if (!A.weights_done) mid_dma(A.wimg, wl, (T + 1) * kWtTypeF16 * 4, wave, lane);
    mid_dma(A.pack + kSpGinW, gw, kSpGinFrags * 16, wave, lane);
    for (int i = lane; i < kMidTileFloats / 4; i += 64) reinterpret_cast<f32x4 *>(tbuf)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nb_res = 0;
    if (ent_resident && my_k < kb) nb_res = load_entries(tile0 + my_k, my_part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gin_items(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // GIN_0 is through with the GIN image
    if (D > 1) mid_dma(A.pack + (size_t)kSpStride + kSpGinW, gw, kSpGinFrags * 16, wave, lane);
    if (A.weights_done) {
        // the NNConv operand images come from a kernel on another stream that may still be running (GIN_0 above did not need
        // them): wait for its last block, drop what this CU caches of other XCDs' lines, then bring layer 0's image in
        if (tid == 0) spin_until_ge(A.weights_done, A.weights_target, spin, kSpinErrWeights);
        __syncthreads();
        if (wave == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        mid_dma(A.wimg, wl, (T + 1) * kWtTypeF16 * 4, wave, lane);
    }

    for (int layer = 0; layer < D; ++layer) {
        const float *sp = A.pack + (size_t)layer * kSpStride;
        TGNN_MT(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the images brought in during the barrier's shadow have landed)
        __syncthreads();
        TGNN_MT(1)
        // =========================================== NNConv_layer ===========================================
        if (my_k < kb) {
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
#ifdef TGNN_MID_TIMING
            const unsigned long long tn0 = wall_clock64();
#endif
            f32x4 d0, d1;
            if (W == 1) {
                for (int k = wave; k < kb; k += kMidWaves) {
                    const int nb = ent_resident ? nb_res : load_entries(tile0 + k, 0);
                    mid_nnconv_partial(N, A, tile0 + k, nb, true, lane, d0, d1);
                    mid_nnconv_finish(N, A, tile0 + k, sp + kSpBias, lane, d0, d1, bn1);
                }
            } else {
                mid_nnconv_partial(N, A, tile0 + my_k, nb_res, my_part == 0, lane, d0, d1);
                if (my_part != 0) {                               // a share's partial product waits in its tile for share 0
                    const int fj = lane & 15, fq = lane >> 4;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, fq)) = d0;
                    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, 4 + fq)) = d1;
                }
            }
#ifdef TGNN_MID_TIMING
            t_nn += wall_clock64() - tn0;
#endif
            if (W > 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();                                  // (W is the same for every wave of every block)
                if (my_part == 0) {
                    const int fj = lane & 15, fq = lane >> 4;
                    for (int j = 1; j < W; ++j) {                 // shares in order
                        const float *pt = tiles + (wave + j) * kMidTileFloats;
                        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(pt + mid_chunk(fj, fq));
                        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(pt + mid_chunk(fj, 4 + fq));
                        d0 = d0 + p0;
                        d1 = d1 + p1;
                    }
                    mid_nnconv_finish(N, A, tile0 + my_k, sp + kSpBias, lane, d0, d1, bn1);
                }
            }
        } else if (W > 1) {
            __syncthreads();                                      // (the shares' hand-over above)
        }
        TGNN_MT(2)
        // =========================================== [R] BatchNorm sums of the layer ===========================================
        // the wave's sums -> its tile; the block's 128 sums (waves in order) -> tagged partial row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the a1 / a2 rows of this block are written)
        bnred[wave * 128 + lane] = bn1;
        bnred[wave * 128 + 64 + lane] = bn2;
        bn1 = bn2 = 0.0;
        __syncthreads();
        TGNN_MT(3)
        // the merge's operands (own rows of a1, a2 and the residual slot; complete: the stores above were waited for) are
        // requested NOW, so that their round trip runs beside the all-reduce's instead of behind it
        const __amdgpu_buffer_rsrc_t a1_rs = rsrc_of(A.a1), a2_rs = rsrc_of(A.a2[layer & 1]);
        const __amdgpu_buffer_rsrc_t r_rs = rsrc_of(A.mid + (size_t)(layer >= 2 ? layer - 2 : 0) * slot);
        constexpr int kMergeItems = (kMidMaxTilesPerBlock * 128 + kMidThreads - 1) / kMidThreads;   // 4
        float4 mg1[kMergeItems], mg2[kMergeItems], mgr[kMergeItems];
        uint32_t mgoff[kMergeItems];
#pragma unroll
        for (int j = 0; j < kMergeItems; ++j) {
            const int idx = tid + j * kMidThreads;
            const int64_t r = tile0 * 16 + (idx >> 3);
            mgoff[j] = (idx < kb * 128 && r < n) ? (uint32_t)r * 128u + (uint32_t)(tid & 7) * 16u : kOob;
            mg1[j] = ld_sc1_f4(a1_rs, mgoff[j]);
            mg2[j] = ld_sc1_f4(a2_rs, mgoff[j]);
            mgr[j] = ld_sc1_f4(r_rs, layer >= 2 ? mgoff[j] : kOob);
        }
        const unsigned tag = 1u + ((unsigned)(layer >> 1) & 1u);
        const size_t par = (size_t)(layer & 1) * nblk;
        const __amdgpu_buffer_rsrc_t p_rs = rsrc_of(A.part + par * 128), g_rs = rsrc_of(A.gpart + par * 128);
        if (tid < 128) {
            double blocksum = 0.0;
#pragma unroll
            for (int w = 0; w < kMidWaves; ++w) blocksum += bnred[w * 128 + tid];
            __builtin_amdgcn_raw_buffer_store_b64(mid_tag(blocksum, tag), p_rs, (blk * 128u + (uint32_t)tid) * 8u, 0, kCpSc1);
        }
        __syncthreads();                                          // (the tiles are free: the fold array aliases them)
        {
            // thread = (pair of doubles jp, rows r and r + 8): two polls in flight
            const int jp = tid & 63, r = tid >> 6;
            // level 1: the 16 blocks of this block's group, rows in order
            const unsigned gbase = blk & ~15u;
            double x0, x1, y0, y1;
            mid_poll_pair(p_rs, gbase + r < nblk ? (int64_t)(gbase + r) : -1, jp, tag, x0, x1, spin);
            mid_poll_pair(p_rs, gbase + r + 8 < nblk ? (int64_t)(gbase + r + 8) : -1, jp, tag, y0, y1, spin);
            red[r * 128 + 2 * jp] = x0;
            red[r * 128 + 2 * jp + 1] = x1;
            red[(r + 8) * 128 + 2 * jp] = y0;
            red[(r + 8) * 128 + 2 * jp + 1] = y1;
            __syncthreads();
            TGNN_MT(4)
            if (tid < 128) {
                double s = 0.0;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) s += red[rr * 128 + tid];
                __builtin_amdgcn_raw_buffer_store_b64(mid_tag(s, tag), g_rs, (blk * 128u + (uint32_t)tid) * 8u, 0, kCpSc1);
            }
            // level 2: one copy of every group's sum (all copies carry the same bits), groups in order
            const unsigned n_groups = (nblk + 15u) >> 4;
            auto group_row = [&](unsigned g) -> int64_t {
                if (g >= n_groups) return -1;
                const unsigned gsize = nblk - 16u * g < 16u ? nblk - 16u * g : 16u;
                const unsigned member = (blk & 15u) < gsize ? (blk & 15u) : gsize - 1u;
                return (int64_t)(16u * g + member);
            };
            mid_poll_pair(g_rs, group_row((unsigned)r), jp, tag, x0, x1, spin);
            mid_poll_pair(g_rs, group_row((unsigned)r + 8u), jp, tag, y0, y1, spin);
            __syncthreads();                                      // (level 1's sums have been read)
            red[r * 128 + 2 * jp] = x0;
            red[r * 128 + 2 * jp + 1] = x1;
            red[(r + 8) * 128 + 2 * jp] = y0;
            red[(r + 8) * 128 + 2 * jp + 1] = y1;
            __syncthreads();
            TGNN_MT(5)
            if (tid < 128) {
                double s = 0.0;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) s += red[rr * 128 + tid];
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
        TGNN_MT(6)
        // =========================================== merge (TilinGNN.py:64-71) ===========================================
        // slot layer + 1 = BN1(a1) * BN2(a2) (+ slot layer - 2), own rows; the slot's largest magnitude for the next NNConv's scale
        {
            const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(A.mid + (size_t)(layer + 1) * slot);
            const int c4 = (tid & 7) * 4;
            const float4 m1h = *reinterpret_cast<const float4 *>(st + c4), m1l = *reinterpret_cast<const float4 *>(st + 32 + c4);
            const float4 g1 = *reinterpret_cast<const float4 *>(st + 64 + c4), b1 = *reinterpret_cast<const float4 *>(st + 96 + c4);
            const float4 m2h = *reinterpret_cast<const float4 *>(st + 128 + c4), m2l = *reinterpret_cast<const float4 *>(st + 160 + c4);
            const float4 g2 = *reinterpret_cast<const float4 *>(st + 192 + c4), b2 = *reinterpret_cast<const float4 *>(st + 224 + c4);
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < kMergeItems; ++j) {
                const float4 x1 = mg1[j], x2 = mg2[j], rs4 = mgr[j];
                float4 o;
                o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x) + rs4.x;
                o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y) + rs4.y;
                o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z) + rs4.z;
                o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w) + rs4.w;
                if (mgoff[j] != kOob) {
                    mx = absmax4(mx, o);
                    st_sc1_f4(o_rs, mgoff[j], o);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
            if (lane == 0) scr[wave] = mx;
        }
        // =========================================== [B] arrive; GIN_{layer + 1} in the shadow; wait ===========================================
        TGNN_MT(7)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        TGNN_MT(8)
        b_target += nblk;
        if (tid == kMidThreads - 64) {                            // (the last wave: it is the one least likely to hold a GIN item)
            float m = scr[0];
#pragma unroll
            for (int w = 1; w < kMidWaves; ++w) m = fmaxf(m, scr[w]);
            if (m > 0.f) atomicMax(A.bounds + layer + 1, __float_as_uint(m));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the bound is in before the arrival is counted)
            __hip_atomic_fetch_add(A.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (layer + 1 < D) {
            // in the barrier's shadow: the collision branch of the next layer, then the images the next phases need (the next
            // layer's NNConv image -- everybody is through with this one -- and its successor's GIN image)
            for (int i = lane; i < kMidTileFloats / 4; i += 64) reinterpret_cast<f32x4 *>(tbuf)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // (the fold left doubles there)
            TGNN_MT(9)
            gin_items(layer + 1);
            TGNN_MT(10)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                      // GIN_{layer + 1} is through with the GIN image
            mid_dma(A.wimg + (size_t)(layer + 1) * (T + 1) * kWtTypeF16, wl, (T + 1) * kWtTypeF16 * 4, wave, lane);
            if (layer + 2 < D) mid_dma(A.pack + (size_t)(layer + 2) * kSpStride + kSpGinW, gw, kSpGinFrags * 16, wave, lane);
            if (tid == kMidThreads - 64) spin_until_ge(A.ctr, b_target, spin, kSpinErrBarrier);
            TGNN_MT(11)
        }
    }
    // (no barrier behind the last merge: the final MLP is the next launch)
#ifdef TGNN_MID_TIMING
    if (blockIdx.x < 256) {
        if (tid == 0)
            for (int k = 0; k < 16; ++k) g_mid_timing[blockIdx.x * 64 + k] = tacc[k];
        if (lane == 0) {
            g_mid_timing[blockIdx.x * 64 + 16 + wave] = t_nn;
            g_mid_timing[blockIdx.x * 64 + 32 + wave] = t_gin;
        }
    }
#endif

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
    return ((size_t)(n_types + 1) * kWtTypeF16 + (size_t)kSpGinFrags * 4 + (size_t)kMidWaves * kMidTileFloats + (size_t)kMidWaves * kMidEntLds + 256 +
            256 + 32 + (size_t)kMidWaves * 256) * sizeof(float);
}
constexpr size_t kMidMaxLds = 160 * 1024 - 256;

// default: up to 8 tiles per block on a 256-CU device (the waves of a block then hold one tile each); beyond that the kernel still
// runs (tiles in rounds, tgnn_set_mid_layout_limit up to 65 536) but the general schedule is faster (measured: 40 000 nodes 1.13 vs
// 1.05 ms, 50 000: 1.28 vs 1.20; 32 000: 0.82 vs 0.97, 10 000: 0.61 vs 0.77)
static std::atomic<int64_t> g_mid_limit{32768};
static std::atomic<int> g_mid_blocks_cap{0};                      // experiments: upper bound of the grid (0 = one block per CU)

// > 0: tiles per block of the persistent layer loop for this layout; 0: not eligible (the general schedule runs)
int mid_layout_tiles_per_block(const tgnn_model_dims *d, const tgnn_graph *g, int64_t n_nodes, int *blocks_out) {
    const int64_t limit = g_mid_limit.load(std::memory_order_relaxed);
    if (n_nodes <= 4096 || n_nodes > limit || n_nodes > 65536 || !g->nn_mid_tile_nb || !g->nn_mid_ent) return 0;   // (up to 4 096: forward_small.hip)
    if (d->network_width != 32 || d->network_depth < 1 || d->network_depth > kMaxDepth) return 0;
    if (g->nn_max_in_degree < 1 || g->n_types + 1 > kMidTileBatches) return 0;
    if (mid_lds_bytes(g->n_types) > kMidMaxLds) return 0;
    if (!persist_allowed()) return 0;                             // (a starved persistent kernel a few forwards ago: general schedule for now)
    static std::atomic<int> capacity[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int cap = capacity[dev].load(std::memory_order_acquire);
    if (cap == 0) {
        static LdsOptIn site;
        int per_cu = 0;
        const hipError_t e1 = opt_in_dynamic_lds(forward_layers_mid_kernel, (int)kMidMaxLds, site);
        const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, forward_layers_mid_kernel, kMidThreads, kMidMaxLds);
        if (e1 != hipSuccess || e2 != hipSuccess) return 0;
        cap = per_cu > 0 ? device_cus() : -1;                     // one 16-wave block with the whole LDS per CU
        capacity[dev].store(cap, std::memory_order_release);
    }
    if (cap <= 0) return 0;
    int max_blocks = cap < 256 ? cap : 256;                       // (mid_part / gpart are sized for 256 blocks: launch_forward_mid)
    if (const int dbg = g_mid_blocks_cap.load(std::memory_order_relaxed); dbg > 0 && dbg < max_blocks) max_blocks = dbg;
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t k = (n_tiles + max_blocks - 1) / max_blocks;
    // a power of two up to 8 tiles per block: the waves of a block then share its tiles evenly (8 / k waves per tile), and fewer
    // blocks make the all-reduce and the barrier cheaper; beyond 8 the waves take their tiles in rounds
    for (int64_t p2 = 1; p2 <= 8; p2 *= 2)
        if (k <= p2) {
            k = p2;
            break;
        }
    if (k > kMidMaxTilesPerBlock) return 0;
    *blocks_out = (int)((n_tiles + k - 1) / k);
    return (int)k;
}

size_t mid_part_doubles() { return (size_t)2 * 2 * 256 * 128; }   // part + gpart: [2 parities][<= 256 blocks][128]

int launch_forward_mid(const tgnn_model_dims *d, const Params &P, float *mid, float *a1, float *a2_0, float *a2_1, const float *wimg,
                       const float *pack, const tgnn_graph *graph, double *part, double *runstat, unsigned *ctr, unsigned *bounds,
                       int64_t n, int tiles_per_block, int blocks, int update_running, float eps, float momentum, hipStream_t s,
                       const unsigned *weights_done, unsigned weights_target, double *const *tail_zero, size_t tail_zero_doubles,
                       const float *x_init) {
    const int depth = d->network_depth;
    MidArgs A{};
    A.weights_done = weights_done;
    A.weights_target = weights_target;
    A.tail_zero[0] = tail_zero ? tail_zero[0] : nullptr;
    A.tail_zero[1] = tail_zero ? tail_zero[1] : nullptr;
    A.tail_zero_vec = (unsigned)(tail_zero_doubles / 2);
    if (x_init) {                                                 // the init MLP in the prologue (the pack carries Linear 1's image)
        const BnPtrs b0 = P.bn(P.init(0) + 2), b1 = P.bn(P.init(1) + 2);
        A.x = x_init;
        A.fx = d->node_features_dim;
        A.iw0 = P.f(P.init(0));
        A.ib0 = P.f(P.init(0) + 1);
        A.ig0 = b0.gamma; A.ibt0 = b0.beta; A.irm0 = b0.rm; A.irv0 = b0.rv; A.inbt0 = b0.nbt;
        A.i1img = small_dense_image(pack, depth, 0);
        A.ib1 = P.f(P.init(1) + 1);
        A.ig1 = b1.gamma; A.ibt1 = b1.beta; A.irm1 = b1.rm; A.irv1 = b1.rv; A.inbt1 = b1.nbt;
    }
    A.mid = mid;
    A.a1 = a1;
    A.a2[0] = a2_0;
    A.a2[1] = a2_1;
    A.wimg = wimg;
    A.pack = pack;
    A.verdict = graph->nn_mid_verdict;
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
    A.err_host = spin_error_mirror();
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
    A.nn_split = tiles_per_block <= 1 ? 8 : tiles_per_block <= 2 ? 4 : tiles_per_block <= 4 ? 2 : 1;
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
    // (the tagged rows of both parities carry no valid tag -- tag 0 is never used -- and the counter is zero: launch_small_pack,
    //  queued on this stream in front of the init MLP, cleared them)
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
// type t of each of the tile's 16 rows, i.e. a column holds at most one edge per row and a row's edges of a type sit in
// consecutive columns in CSR order).  Per tile and type run the columns are laid one behind the other into gather
// instructions of 8 entries; a column starts a NEW instruction when one of its rows already has an entry in the one being
// filled: no instruction then holds two entries of a row, and a row's r-th edge of the type comes in a later instruction than
// its (r - 1)-th -- it read-add-writes the type-sum slot the first one stored.  4 instructions make a batch; a run of more than 32 slots continues in further batches, the last one carries the
// `last` flag and the mask of rows that have an edge of the type (= the rows of the run's first column).
//   batch = [type | last << 8, row mask, 0, 0 | 32 entry words: slot o (0..7) of instruction g (0..3) at 4 + 4 o + g]
//   entry = source row (24 bits; 0x1000000: none) | destination row << 25 (16: none) | add << 30
// One wave packs FOUR tiles, 16 lanes (= the 16 rows) each: a column is one ballot + one prefix count.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kMidBuildMaxCols = 64;

__global__ __launch_bounds__(64) void mid_entries_kernel(const int *__restrict__ tile_col_ptr, const int *__restrict__ col_meta,
                                                         const int *__restrict__ col_src, int64_t n_tiles,
                                                         const int *__restrict__ cols_built, int *__restrict__ tile_nb,
                                                         uint32_t *__restrict__ ent, int *__restrict__ result) {
    __shared__ int s_src[4][kMidBuildMaxCols * 16];
    __shared__ int s_meta[4][kMidBuildMaxCols];
    __shared__ uint32_t s_out[4][kMidEntWords];
    const int tid = threadIdx.x, q = tid >> 4, i = tid & 15;
    const int64_t tile = (int64_t)blockIdx.x * 4 + q;
    const bool live = tile < n_tiles;
    if (cols_built && *cols_built == 0) {                         // no column structure (too many edge types): nothing to pack
        if (live && i == 0) tile_nb[tile] = 0;
        if (tid == 0 && blockIdx.x == 0) result[1] = 1;
        return;
    }
    int c0 = 0, nc = 0;
    if (live) {
        c0 = tile_col_ptr[tile];
        nc = tile_col_ptr[tile + 1] - c0 - 1;                     // edge columns (the last column of a tile is the root column)
    }
    bool over = nc > kMidBuildMaxCols;
    if (over) nc = 0;
    for (int w = i; w < nc * 16; w += 16) s_src[q][w] = col_src[(int64_t)c0 * 16 + w];
    for (int w = i; w < nc; w += 16) s_meta[q][w] = col_meta[c0 + w];
    for (int w = i; w < kMidEntWords; w += 16) s_out[q][w] = (w % kMidBatchWords) < 4 ? 0u : kMidEmptyEntry;
    __syncthreads();
    int ncmax = nc;
#pragma unroll
    for (int d = 32; d >= 16; d >>= 1) ncmax = max(ncmax, __shfl_xor(ncmax, d, 64));
    // state of this quarter's tile (the same in its 16 lanes): batches closed so far, slots used / type / rows of the open run,
    // rows that already have an entry in the gather instruction being filled
    int nb = 0, rpos = 0, type = 0;
    unsigned runmask = 0, gmask = 0;
    bool open = false;
    auto close_run = [&]() {
        const int nbr = rpos > 0 ? (rpos + 31) >> 5 : 1;
        if (i == 0)
            for (int b = 0; b < nbr; ++b)
                if (nb + b < kMidTileBatches) {
                    s_out[q][(nb + b) * kMidBatchWords] = (unsigned)type | (b == nbr - 1 ? 0x100u : 0u);
                    s_out[q][(nb + b) * kMidBatchWords + 1] = runmask;
                }
        nb += nbr;
        rpos = 0;
        gmask = 0;
    };
    for (int k = 0; k < ncmax; ++k) {
        const bool act = k < nc;
        const int sv = act ? s_src[q][k * 16 + i] : -1;
        const bool valid = sv >= 0;
        const unsigned bal = (unsigned)(__ballot(valid) >> (16 * q)) & 0xffffu;
        const int m = act ? s_meta[q][k] : 0;
        const bool first = (m & kMidColFirst) != 0;
        if (first) {
            if (open) close_run();
            type = m & 0xff;
            runmask = bal;
            open = true;
        }
        // a column joins the instruction being filled when none of its rows is in it yet and it fits; else it starts a new one (a
        // column of more than 8 entries runs on into the next instruction: its rows are all different)
        const int cnt = __popc(bal), fill = rpos & 7;
        if (fill != 0 && ((gmask & bal) != 0u || fill + cnt > 8)) {
            rpos = (rpos + 7) & ~7;
            gmask = 0;
        }
        const int p = rpos + __popc(bal & ((1u << i) - 1u));
        if (valid) {
            const int b = nb + (p >> 5), pp = p & 31;
            if (b < kMidTileBatches)
                s_out[q][b * kMidBatchWords + 4 + 4 * (pp & 7) + (pp >> 3)] =
                    ((unsigned)sv & 0xffffffu) | ((unsigned)i << 25) | (first ? 0u : 1u << 30);
        }
        const int end = rpos + cnt, lastg = (end - 1) >> 3;
        const unsigned in_last = (unsigned)(__ballot(valid && (p >> 3) == lastg) >> (16 * q)) & 0xffffu;
        if (cnt > 0) {
            gmask = (rpos >> 3) == lastg ? (gmask | in_last) : in_last;
            if ((end & 7) == 0) gmask = 0;
            rpos = end;
        }
    }
    if (open) close_run();
    if (nb > kMidTileBatches) over = true;
    if (over) nb = 0;
    __syncthreads();
    if (live) {
        if (i == 0) {
            tile_nb[tile] = nb;
            if (over) result[1] = 1;                              // the layout is not for this kernel
            else if (nb > result[0]) atomicMax(result, nb);
        }
        uint32_t *dst = ent + (size_t)tile * kMidEntWords;
        for (int w = i; w < nb * kMidBatchWords; w += 16) dst[w] = s_out[q][w];
    }
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
    mid_entries_kernel<<<(unsigned)((n_tiles + 3) / 4), 64, 0, static_cast<hipStream_t>(stream)>>>(tile_col_ptr, col_meta, col_src, n_tiles,
                                                                                                   cols_built_dev, tile_nb, ent, result);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

#ifdef TGNN_MID_TIMING
extern "C" int tgnn_debug_mid_timing(unsigned long long *out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnn::g_mid_timing), (size_t)n_blocks * 64 * sizeof(unsigned long long));
}
#endif
extern "C" void tgnn_set_mid_layout_limit(int64_t n_nodes) { g_mid_limit.store(n_nodes < 0 ? 0 : n_nodes); }
extern "C" int64_t tgnn_get_mid_layout_limit(void) { return g_mid_limit.load(); }
extern "C" int64_t tgnn_mid_layout_max_nodes(void) { return 65536; }
extern "C" void tgnn_debug_set_mid_blocks(int32_t blocks) { g_mid_blocks_cap.store(blocks < 0 ? 0 : blocks); }
