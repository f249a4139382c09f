// Shared pieces of the two persistent layer-loop kernels (forward_small.hip: layouts up to 4 096 nodes, one tile per block;
// forward_mid.hip: up to 65 536 nodes, several tiles per block): the per-layer parameter pack, the agent-scope (sc1) buffer
// accesses cross-block data travels through, and the BOUNDED spin every cross-block wait of these kernels goes through.
#pragma once
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

// running-statistics buffers of the two BatchNorms of every layer (updated once, after the last layer)
struct SmallRun {
    float *rm1, *rv1;
    int64_t *nbt1;
    float *rm2, *rv2;
    int64_t *nbt2;
};
struct SmallRunTab {
    SmallRun l[kMaxDepth];
};

constexpr int kCpSc1 = 16;       // cache-policy bit of the raw buffer builtins: sc1 = agent scope (coherent across the XCDs' L2s)
constexpr uint32_t kOob = 0x80000000u;   // offset outside the 2 GB window of every descriptor here: the load returns 0

__device__ __forceinline__ float4 ld_sc1_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kCpSc1));
}
__device__ __forceinline__ void st_sc1_f4(__amdgpu_buffer_rsrc_t r, uint32_t off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, kCpSc1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)0x80000000u, 0x00020000);
}

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

template <int CP>
__device__ __forceinline__ float4 ld_cp_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, CP));
}

// ---- wave tile in LDS: [17 rows][36 floats]: rows of 32 floats padded to 144 bytes (a matrix-layout read -- lane (n, q): floats
//      8 q .. 8 q + 7 of row n -- then meets at most 2-way bank conflicts, whole-row writes and column walks none), row 16 = where
//      the empty slots of a gather batch store.  An address is one multiply-add of the row number.
constexpr int kMidRowFloats = 36;
__device__ __forceinline__ int mid_chunk(int row, int c) { return row * kMidRowFloats + (c << 2); }   // float index of 16-byte chunk c

constexpr int kMidTileFloats = 17 * kMidRowFloats;               // a wave's tile incl. the spare row

// ---- CollConv of one 16-row tile on one wave ------------------------------------------------------------------------------------
// (coll_conv.py:24-27 / PyG GINConv): neighbourhood sum of whole 128-byte rows with the producer's BatchNorm folded in (st2: its
// record, used when use_stat), z through the wave's LDS tile into the matrix layout, the 32 -> 32 -> 64 -> 32 sigmoid MLP on bf16 x 3
// fragments (gw: the images of small_pack_kernel / gin_pack_image, sp: the layer's parameter vectors), LeakyReLU; the rows go to
// dst, their BatchNorm column sums (fp64; lane = (channel lane & 31, sum | sum of squares)) into bn_acc.
// CP: cache policy of the row gathers and stores -- kCpSc1 inside a persistent kernel (rows other blocks of the SAME launch wrote),
// 0 in a kernel of its own.
struct GinGraph {
    const int *rowptr, *nbr;     // collision CSR by destination
    int64_t n;
};
// first half: the neighbourhood sums z of the tile's 16 rows -> zt (an LDS tile in mid_chunk layout)
template <int CP>
__device__ __forceinline__ void gin_tile_gather(const GinGraph &A, int64_t tile, bool use_stat, const float *src, const float *sp,
                                                const float *st2, float *zt, int lane) {
    const int go = lane >> 3, gp = lane & 7;
    float *tbuf = zt;
    const int64_t n = A.n;
    const __amdgpu_buffer_rsrc_t a_rs = rsrc_of(src);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    // x = BatchNorm of the previous layer's pre-BN rows, folded into the sum as in gin32_aggregate_kernel (gin.hip)
    const float4 mhi = use_stat ? *reinterpret_cast<const float4 *>(st2 + 4 * gp) : zero4;
    const float4 mlo = use_stat ? *reinterpret_cast<const float4 *>(st2 + 32 + 4 * gp) : zero4;
    const float4 gv = use_stat ? *reinterpret_cast<const float4 *>(st2 + 64 + 4 * gp) : one4;
    const float4 bv = use_stat ? *reinterpret_cast<const float4 *>(st2 + 96 + 4 * gp) : zero4;
    const float one_eps = sp[kSpEps];
    // neighbourhood sums: lane (go, gp) walks the rows 8 h + go (h = 0, 1), piece gp.  A row's 8 lanes fetch 8 consecutive
    // neighbour indices with ONE load (the next 8 while the rows of these fly) and hand them round by lane permutes; 8 source
    // rows of both destination rows are in flight per step: ~3 memory round trips per 16 neighbours instead of one per 4
    int beg[2], deg[2];
    float4 selfv[2], acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t row = tile * 16 + 8 * h + go;
        const bool ok = row < n;
        beg[h] = ok ? A.rowptr[row] : 0;
        deg[h] = ok ? A.rowptr[row + 1] - beg[h] : 0;
        selfv[h] = ld_cp_f4<CP>(a_rs, ok ? (uint32_t)row * 128u + (uint32_t)gp * 16u : kOob);
        acc[h] = zero4;
    }
    int maxdeg = deg[0] > deg[1] ? deg[0] : deg[1];
#pragma unroll
    for (int d = 32; d >= 8; d >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, d, 64));
    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
    auto load_idx = [&](int k0, int (&idx)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool in = k0 + gp < deg[h];
            const int v = A.nbr[in ? beg[h] + k0 + gp : 0];
            idx[h] = in ? v : -1;
        }
    };
    int idx[2], idx_next[2];
    load_idx(0, idx);
    for (int k0 = 0; k0 < maxdeg; k0 += 8) {
        load_idx(k0 + 8, idx_next);                               // (past the longest row: nothing is fetched)
        float4 y[2][8];
        bool has[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int nbk = __shfl(idx[h], (lane & ~7) | k, 64);
                has[h][k] = nbk >= 0;
                y[h][k] = ld_cp_f4<CP>(a_rs, nbk >= 0 ? (uint32_t)nbk * 128u + (uint32_t)gp * 16u : kOob);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (has[h][k]) {
                    acc[h].x += (y[h][k].x - mhi.x) - mlo.x; acc[h].y += (y[h][k].y - mhi.y) - mlo.y;
                    acc[h].z += (y[h][k].z - mhi.z) - mlo.z; acc[h].w += (y[h][k].w - mhi.w) - mlo.w;
                }
        idx[0] = idx_next[0];
        idx[1] = idx_next[1];
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
}
// second half: z (zt, possibly another wave's hand-over slot: *release = release_val once it has been read) through the MLP; the
// output rows to dst and, through this wave's own tile tbuf, into the BatchNorm sums
template <int CP>
__device__ __forceinline__ void gin_tile_mlp(const GinGraph &A, int64_t tile, float *dst, const float *gw, const float *sp, const float *zt,
                                             float *tbuf, int lane, double &bn_acc, volatile int *release = nullptr, int release_val = 0) {
    const int fj = lane & 15, fq = lane >> 4;
    const int64_t n = A.n;
    // matrix layout: lane (n = fj, q = fq) holds floats 8 q .. 8 q + 7 of tile row n
    float z[8];
    {
        const f32x4 za = *reinterpret_cast<const f32x4 *>(zt + mid_chunk(fj, 2 * fq));
        const f32x4 zb = *reinterpret_cast<const f32x4 *>(zt + mid_chunk(fj, 2 * fq + 1));
        z[0] = za[0]; z[1] = za[1]; z[2] = za[2]; z[3] = za[3]; z[4] = zb[0]; z[5] = zb[1]; z[6] = zb[2]; z[7] = zb[3];
    }
    if (release) {                                                // (wave-uniform)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) *release = release_val;
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
    auto sig_out = [](float v) { return sigmoid_out_f32(v); };           // full accuracy, as gin32_mlp_kernel (LeakyReLU behind a sigmoid: the identity)
    const int64_t my_row = tile * 16 + fj;
    const bool row_ok = my_row < n;
    f32x4 r0, r1;
    r0[0] = sig_out(o0[0]); r0[1] = sig_out(o0[1]); r0[2] = sig_out(o0[2]); r0[3] = sig_out(o0[3]);
    r1[0] = sig_out(o1[0]); r1[1] = sig_out(o1[1]); r1[2] = sig_out(o1[2]); r1[3] = sig_out(o1[3]);
    if (!row_ok) r0 = r1 = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                              // (everybody has read z)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, fq)) = r0;
    *reinterpret_cast<f32x4 *>(tbuf + mid_chunk(fj, 4 + fq)) = r1;
    const __amdgpu_buffer_rsrc_t o_rs = rsrc_of(dst);
    const uint32_t o_off = row_ok ? (uint32_t)my_row * 128u + (uint32_t)fq * 16u : kOob;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r0), o_rs, o_off, 0, CP);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r1), o_rs, o_off == kOob ? kOob : o_off + 64u, 0, CP);
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
// both halves on one wave, z through the wave's own tile (the mid-size persistent kernel's work item)
template <int CP>
__device__ __forceinline__ void gin_tile(const GinGraph &A, int64_t tile, bool use_stat, const float *src, float *dst, const float *gw,
                                         const float *sp, const float *st2, float *tbuf, int lane, double &bn_acc) {
    gin_tile_gather<CP>(A, tile, use_stat, src, sp, st2, tbuf, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    gin_tile_mlp<CP>(A, tile, dst, gw, sp, tbuf, tbuf, lane, bn_acc);
}

// ---- bounded spins ------------------------------------------------------------------------------------------------------
// A persistent kernel's blocks wait for each other (grid barriers, tagged partial rows, a flag of the pre-pass kernel).  That is
// only sound while ALL of its blocks are resident, which the launch gate (spin_kernel_chain) guarantees inside ONE process --
// a second process on the same GPU, or any tenant holding CUs, is invisible to it.  So no wait spins without bound: after
// kSpinBudgetTicks of the device's constant 100 MHz clock (wall_clock64) a waiter gives up, ORs a reason code into the
// device's error word and lets its block run on WITHOUT waiting (so that the kernel, and every block still waiting for this
// one, terminates); blocks also leave their waits as soon as they see the word set by anybody.  The forward's results are
// then garbage: the host reads the word (tgnn_spin_error_poll), disables the persistent schedules and runs the general
// launch schedule instead (tilingnn_amd.TilinGNN does this for ML_Solver.predict).
constexpr unsigned long long kSpinBudgetTicksDefault = 25ull * 1000 * 1000;   // 0.25 s: ~500 x the longest forward these kernels run
constexpr unsigned kSpinErrBarrier = 1u, kSpinErrRows = 2u, kSpinErrWeights = 4u;

struct SpinCtx {
    unsigned *err;                       // device error word (never NULL)
    unsigned long long budget;           // ticks
    bool gave_up;                        // this thread has given up (it then stops waiting for good)
    unsigned *mirror = nullptr;          // the same word in host-mapped memory (may be NULL): written by whoever GIVES UP, so
                                         // that the host sees a failure at its next forward call without a copy or a wait
};
// true: go on waiting; false: stop (budget used up, or somebody reported an error)
__device__ __forceinline__ bool spin_continue(SpinCtx &sp, unsigned long long t0, unsigned iter, unsigned code) {
    if (sp.gave_up) return false;
    if ((iter & 63u) != 63u) return true;
    if (__hip_atomic_load(sp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        sp.gave_up = true;
        return false;
    }
    if (wall_clock64() - t0 > sp.budget) {
        __hip_atomic_fetch_or(sp.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (a STORE: of the read-modify-writes only add / swap / compare-and-swap exist on the host link; any non-zero value says it)
        if (sp.mirror) __hip_atomic_store(sp.mirror, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sp.gave_up = true;
        return false;
    }
    return true;
}
// one thread: wait until *ctr >= target
__device__ __forceinline__ void spin_until_ge(const unsigned *ctr, unsigned target, SpinCtx &sp, unsigned code) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++it) {
        if (!spin_continue(sp, t0, it, code)) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// the per-device error word + budget of the spin kernels (forward_small.hip); NULL when the allocation failed
unsigned *spin_error_word();
// its host-mapped mirror as the device sees it (NULL: none); spin_error_pending: what the host sees there right now
unsigned *spin_error_mirror();
unsigned spin_error_pending();
// A failure nobody has collected (tgnn_spin_error_poll was not called behind the forward that failed: plain forward(), the crop
// loop, a C caller): cleared here, the persistent schedules go off for a while (persist_fallback) and the caller is told.  Called
// at the entry of every forward of the library; returns TGNN_OK or TGNN_ERR_STALE_RESULT with the message set.
int spin_error_collect_stale(hipStream_t s);
// the persistent schedules are off for the next n forwards of this process, then come back (0: back now)
void persist_fallback(int64_t n_forwards);
bool persist_allowed();                  // (counts a forward of the window down)
unsigned long long spin_budget_ticks();
// test hook (tgnn_debug_spin_fault): 1 = the launch being queued runs with its last block absent (it returns at once), which
// is what a block that never becomes resident looks like to the others
int spin_take_fault();

}  // namespace tgnn
