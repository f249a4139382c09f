// The final MLP of a mid-size layout (4 097 .. 16 384 nodes) as ONE persistent kernel behind the persistent layer loop
// (forward_mid.hip) -- TilinGNN.py:74-76: cat(middle) -> 4 x (Linear -> LeakyReLU -> BatchNorm) -> Linear(32, out) -> sigmoid.
//
// The general schedule runs this as 5 Linear launches with 4 one-block bn_finalize launches between them: at 10 000 nodes
// 35 + 20 + 14 + 7 + 4 us of latency-bound block-tile kernels (a 128-row tile per block leaves 1.2 blocks per CU, every K step
// waits a full memory round trip) + 4 x 4.7 us + a 10 us gap behind the layer loop = 110 us of a 600 us forward
// (profiles/r05_mid_trace_10000.txt).  Here one block per CU owns G = 2 .. 4 16-row tiles for all five layers:
//   * rows = the B operand of v_mfma_f32_16x16x32, output channels = the A operand (weight fragments from L2-resident images
//     built once per forward by small_pack_kernel), D^T = W . X^T: a wave owns 16 or 32 output channels of every tile;
//   * layer 0 (K = 32 (depth + 1)) on fp16 PAIRS -- three cross terms; the slots' largest magnitudes are at hand (the layer
//     loop left them), the weights' too -- in chunks of 8 K steps: wave w brings slot 8 c + w's rows in (the next chunk's while
//     this one multiplies) and splits them into the operand planes, between two barriers every wave walks the 8 steps in
//     straight-line code with its weight fragments two steps ahead;
//   * layers 1 .. 3 (K = 256, 128, 64) on bf16 x 3 (six cross terms; no bound needed): the activations never leave LDS, a wave's
//     weight fragments of the WHOLE layer are requested one layer ahead and land during the all-reduce in front of it;
//   * the BatchNorm column sums of a layer are all-reduced over the grid through TAGGED partial rows in two levels of 16 (a
//     reader polls the data itself; fixed order: bit-reproducible), the record is applied while a row is split into the next
//     layer's operand planes.
// Measured: profiles/r05_mid_tail.txt, DESIGN.md section 15.
#include <hip/hip_runtime.h>

#include <atomic>

#include "forward_persist.h"

namespace tgnn {

constexpr int kTailThreads = 512;              // 8 waves
constexpr int kTailGroup = 4;                  // tiles per block at most: their activations sit in LDS
constexpr int kTailActLd = 260;                // floats per activation row in LDS (256 + 4: column walks and row writes conflict-free)
constexpr int kTailChunk = 8;                  // K steps whose operand planes sit in LDS at a time
constexpr int kTailPlaneVec = 3 * 64;          // 16-byte fragments of one tile's bf16 x 3 operand planes of one K step
constexpr int kTailPlaneVec16 = 2 * 64;        // ... of its fp16-pair planes (layer 0)
constexpr int kTailRowDoubles = 512 + 256 + 128 + 64;   // a block's partial rows of the four BatchNorms
__host__ __device__ constexpr int tail_row_off(int l) { return l == 0 ? 0 : l == 1 ? 512 : l == 2 ? 768 : 896; }
constexpr int kTailMaxBlocks = 256;
constexpr size_t kTailPlaneBytes = (size_t)kTailChunk * kTailGroup * kTailPlaneVec * 16;      // 96 KB; the activations (65 KB) alias it
static_assert(kTailPlaneBytes >= (size_t)kTailGroup * 16 * kTailActLd * 4, "the activations fit the planes");
constexpr size_t kTailLdsBytes = kTailPlaneBytes + 1024 * 4 + 512 * 8;

struct TailLayer {
    const float *img, *bias, *gamma, *beta;    // img: [M / 16][K / 32][plane][lane 64] x 16 B (small_pack_kernel; layer 0: 2 fp16 planes)
    float *rm, *rv;
    int64_t *nbt;
};
struct TailArgs {
    const float *mid;            // skip buffer [depth + 1][n][32]
    TailLayer f[4];
    const float *w_last, *b_last;
    float *probs;
    double *part, *gpart;        // [layer region: 256 x tail_row_off(l)][block][2 M]: the blocks' column sums, the groups' sums; tagged
                                 // doubles, ZERO before the launch (the layer loop's kernel clears them)
    const unsigned *slot_max;    // [depth + 1] largest |slot k| as float bits (the layer loop's merges, bn_apply for slot 0)
    const unsigned *w0_max;      // largest |W| of layer 0 as float bits (the scale its image was built with)
    unsigned *err, *err_host;
    unsigned long long spin_budget;
    int64_t n;
    int depth, tiles_per_block, out_dim, update_running, fault;
    float eps, momentum;
    const int *verdict;          // as MidArgs.verdict: the layer loop's kernel in front of this one left without filling `mid`
};

#ifdef TGNN_TAIL_TIMING
// phase stamps of every block (scratch builds only): wall_clock64 ticks (100 MHz) of thread 0 at [layer 4][dense done, column sums
// done, -, level 1 done, level 2 done, record done], [24] = kernel entry, [25] = exit, [26 ..] = layer 0's first two chunks
__device__ unsigned long long g_tail_timing[256 * 32];
#define TGNN_TT(k) if (tid == 0) g_tail_timing[blockIdx.x * 32 + (k)] = wall_clock64();
#else
#define TGNN_TT(k)
#endif

using f16x8 = tgnn_f16x8;
__host__ __device__ constexpr int tail_mbw(int m) { return m / 16 >= 8 ? m / 16 / 8 : 1; }

// One K step of the wave's MBW output blocks x TW tiles, TERM-major: between two matrix instructions on the same accumulator sit
// MBW x TW - 1 independent ones.  bf16 x 3: the six cross terms, smallest first.
template <int MBW, int TW>
__device__ __forceinline__ void tail_mma_step(const u32x4 (&w)[MBW][3], const bf16x8 (&x)[TW][3], f32x4 (&acc)[MBW][TW]) {
    constexpr int wp[6] = {2, 0, 1, 1, 0, 0}, xp[6] = {0, 2, 1, 0, 1, 0};   // lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int m = 0; m < MBW; ++m)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[m][wp[term]]), x[j][xp[term]], acc[m][j], 0, 0, 0);
}
// fp16 pairs: lo.hi, hi.lo, hi.hi (dense_split_kernel<.., true>'s order)
template <int MBW, int TW>
__device__ __forceinline__ void tail_mma_step16(const u32x4 (&w)[MBW][2], const f16x8 (&x)[TW][2], f32x4 (&acc)[MBW][TW]) {
    constexpr int wp[3] = {1, 0, 0}, xp[3] = {0, 1, 0};
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int m = 0; m < MBW; ++m)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[m][wp[term]]), x[j][xp[term]], acc[m][j], 0, 0, 0);
}
// all K steps of a narrow layer's weight fragments of this wave, into registers (issued a layer ahead)
template <int M, int KS>
__device__ __forceinline__ void tail_wload_all(const TailLayer &L, u32x4 (&w)[KS][tail_mbw(M)][3], int tid) {
    constexpr int MB = M / 16, MBW = tail_mbw(M);
    const int wave = tid >> 6, lane = tid & 63;
    const int mb0 = MB >= 8 ? wave * MBW : wave % MB;
    const u32x4 *img = reinterpret_cast<const u32x4 *>(L.img);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[ks][m][pl] = img[((size_t)(mb0 + m) * KS + ks) * kTailPlaneVec + pl * 64 + lane];
}

// the wave's accumulators -> LeakyReLU(scale * acc + bias) -> actl [tile][16][kTailActLd]
template <int M, int G, int MBW, int TW>
__device__ __forceinline__ void tail_store_act(const f32x4 (&acc)[MBW][TW], const float *bias, float scale, int ng, float *actl, int tid) {
    constexpr int MB = M / 16, WPM = MB >= 8 ? 1 : 8 / MB;
    const int wave = tid >> 6, lane = tid & 63, fq = lane >> 4, fn = lane & 15;
    const int mb0 = MB >= 8 ? wave * MBW : wave % MB;
    const int tsel = MB >= 8 ? 0 : wave / MB;
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int t = tsel + j * WPM;
        if (t < ng && t < G) {
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * (mb0 + m) + 4 * fq);
                *reinterpret_cast<float4 *>(actl + (t * 16 + fn) * kTailActLd + 16 * (mb0 + m) + 4 * fq) =
                    make_float4(leakyf_(fmaf(acc[m][j][0], scale, b.x)), leakyf_(fmaf(acc[m][j][1], scale, b.y)),
                                leakyf_(fmaf(acc[m][j][2], scale, b.z)), leakyf_(fmaf(acc[m][j][3], scale, b.w)));
            }
        }
    }
}

// Layer 0: Linear(32 (depth + 1) -> 256) + LeakyReLU over the block's G tiles, X = their rows of every slot of the skip buffer.
// (One barrier per K step with the loads in flight across it does not work: the fence of __syncthreads drains the vector-memory
// counter and every step waits a full round trip -- 1.3 us per step, measured.  Hence chunks, plain s_barrier and explicit waits.)
template <int G>
__device__ __forceinline__ void tail_first_layer(const TailArgs &A, int64_t tile0, int ng, u32x4 *planes, float *actl, int tid) {
    constexpr int M = 256, MBW = 2, TW = G;
    const TailLayer &L = A.f[0];
    const int ksteps = A.depth + 1;
    const int wave = tid >> 6, lane = tid & 63;
    const int mb0 = wave * MBW;
    const int64_t n = A.n;
    const int xn = lane & 15, xq = lane >> 4;
    const __amdgpu_buffer_rsrc_t in_rs = rsrc_of(A.mid);
    const uint32_t xstep = (uint32_t)n * 128u;                    // bytes from one slot to the next
    bool xok[G];
    uint32_t xbase[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int64_t xrow = (tile0 + j) * 16 + xn;
        xok[j] = j < ng && xrow < n;
        xbase[j] = (uint32_t)xrow * 128u + (uint32_t)xq * 32u;
    }
    const __amdgpu_buffer_rsrc_t img_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.img), 0, (M / 16) * ksteps * kTailPlaneVec16 * 16, 0x00020000);
    // the two operands' powers of two (their maxima just below 2^15), taken off the accumulators in the epilogue
    float sa;
    {
        unsigned mb = 0;
        for (int i = lane; i < ksteps; i += 64) mb = max(mb, A.slot_max[i]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d, 64));
        sa = pow2_scale_for(mb, 0);
    }
    const float unscale = 1.0f / (sa * pow2_scale_for(*A.w0_max, 0));

    f32x4 acc[MBW][TW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xr[G][2];
    auto load_rows = [&](int c) {
        const int ks = kTailChunk * c + wave;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint32_t off = (xok[j] && ks < ksteps) ? xbase[j] + (uint32_t)ks * xstep : kOob;
            xr[j][0] = ld_cp_f4<0>(in_rs, off);                   // (rows the previous launch wrote)
            xr[j][1] = ld_cp_f4<0>(in_rs, off == kOob ? kOob : off + 16u);
        }
    };
    u32x4 wq[3][MBW][2];
    auto wload = [&](int c, int ksl, u32x4 (&wf)[MBW][2]) {
        const int ks = kTailChunk * c + ksl;
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const uint32_t off = ks < ksteps ? ((uint32_t)((mb0 + m) * ksteps + ks) * kTailPlaneVec16 + (uint32_t)(pl * 64 + lane)) * 16u : kOob;
                wf[m][pl] = __builtin_amdgcn_raw_buffer_load_b128(img_rs, off, 0, 0);
            }
    };
    const int nchunks = (ksteps + kTailChunk - 1) / kTailChunk;
    load_rows(0);
    wload(0, 0, wq[0]);
    wload(0, 1, wq[1]);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const float x[8] = {xr[j][0].x, xr[j][0].y, xr[j][0].z, xr[j][0].w, xr[j][1].x, xr[j][1].y, xr[j][1].z, xr[j][1].w};
            f16x8 *dst = reinterpret_cast<f16x8 *>(planes) + (wave * G + j) * kTailPlaneVec16 + lane;
            split2_f16(x, sa, dst[0], dst[64]);
        }
        if (c < 2) { TGNN_TT(26 + 3 * c) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // (LDS only: the weight fragments of steps 0, 1 stay in flight)
        if (c < 2) { TGNN_TT(27 + 3 * c) }
        if (c + 1 < nchunks) load_rows(c + 1);
#pragma unroll
        for (int ksl = 0; ksl < kTailChunk; ++ksl) {
            if (ksl + 2 < kTailChunk) wload(c, ksl + 2, wq[(ksl + 2) % 3]);
            if (kTailChunk * c + ksl < ksteps) {                  // (uniform)
                f16x8 x2[TW][2];
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const f16x8 *xp = reinterpret_cast<const f16x8 *>(planes) + (ksl * G + j) * kTailPlaneVec16 + lane;
                    x2[j][0] = xp[0];
                    x2[j][1] = xp[64];
                }
                tail_mma_step16<MBW, TW>(wq[ksl % 3], x2, acc);
            }
        }
        if (c + 1 < nchunks) {                                    // the next chunk's first fragments: in flight across the barrier
            wload(c + 1, 0, wq[0]);
            wload(c + 1, 1, wq[1]);
        }
        if (c < 2) { TGNN_TT(28 + 3 * c) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // (the planes are free: the next chunk's, or the activations)
    }
    tail_store_act<M, G, MBW, TW>(acc, L.bias, unscale, ng, actl, tid);
    __syncthreads();
}

// Layers 1 .. 3: Linear(32 KS -> M) + LeakyReLU on X = BatchNorm(the activations in LDS, record rec [4][256]); the wave's weight
// fragments of all KS steps are in w already.  The operand planes ALIAS the activations: every wave takes its K step out of them
// first, and only behind a barrier are the planes written.
template <int M, int G, int KS>
__device__ __forceinline__ void tail_narrow_layer(const TailArgs &A, const TailLayer &L, int64_t tile0, int ng, const float *rec, u32x4 *planes,
                                                  float *actl, int tid, const u32x4 (&w)[KS][tail_mbw(M)][3]) {
    constexpr int MB = M / 16, MBW = tail_mbw(M), WPM = MB >= 8 ? 1 : 8 / MB, TW = (G + WPM - 1) / WPM;
    const int wave = tid >> 6, lane = tid & 63;
    const int tsel = MB >= 8 ? 0 : wave / MB;
    const int64_t n = A.n;
    const int xn = lane & 15, xq = lane >> 4;
    float4 xr[G][2];
    if (wave < KS) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const float *src = actl + (j * 16 + xn) * kTailActLd + 32 * wave + 8 * xq;
            xr[j][0] = *reinterpret_cast<const float4 *>(src);
            xr[j][1] = *reinterpret_cast<const float4 *>(src + 4);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave < KS) {
        const int k0 = 32 * wave + 8 * xq;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const bool ok = j < ng && (tile0 + j) * 16 + xn < n;
            float x[8] = {xr[j][0].x, xr[j][0].y, xr[j][0].z, xr[j][0].w, xr[j][1].x, xr[j][1].y, xr[j][1].z, xr[j][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                x[e] = ok ? bn_apply1(x[e], rec[k0 + e], rec[256 + k0 + e], rec[512 + k0 + e], rec[768 + k0 + e]) : 0.f;
            bf16x8 *dst = reinterpret_cast<bf16x8 *>(planes) + (wave * G + j) * kTailPlaneVec + lane;
            split3_trunc(x, dst[0], dst[64], dst[128]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 acc[MBW][TW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ksl = 0; ksl < KS; ++ksl) {
        bf16x8 x3[TW][3];
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            int t = tsel + j * WPM;
            t = t < G ? t : G - 1;                                // (a wave without a tile of its own left repeats the last one; not stored)
            const bf16x8 *xp = reinterpret_cast<const bf16x8 *>(planes) + (ksl * G + t) * kTailPlaneVec + lane;
            x3[j][0] = xp[0];
            x3[j][1] = xp[64];
            x3[j][2] = xp[128];
        }
        tail_mma_step<MBW, TW>(w[ksl], x3, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // (the planes are free: the activations go there)
    tail_store_act<M, G, MBW, TW>(acc, L.bias, 1.0f, ng, actl, tid);
    __syncthreads();
}

// ---- one tagged double: the low two mantissa bits say "written by this forward" (the rows are zeroed by the layer loop's kernel
//      in front of this one: forward_mid.hip) -- a reader polls the DATA, no flag word and no second round trip
constexpr unsigned kTailTag = 1u;
__device__ __forceinline__ u32x2 tail_tag(double v) {
    u32x2 b = __builtin_bit_cast(u32x2, v);
    b[0] = (b[0] & ~3u) | kTailTag;
    return b;
}
// column `col` of 16 rows of a tagged array [row][NV] (row_of(r) < 0: no such row), polled until every one carries the tag, summed
// in order
template <int NV, typename RowOf>
__device__ __forceinline__ double tail_poll_sum16(__amdgpu_buffer_rsrc_t rs, RowOf row_of, int col, SpinCtx &sp) {
    u32x2 v[16];
    uint32_t off[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_of(r);
        off[r] = row >= 0 ? ((uint32_t)row * (uint32_t)NV + (uint32_t)col) * 8u : kOob;
        v[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, off[r], 0, kCpSc1);
    }
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0;; ++it) {
        bool all = true;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (off[r] != kOob && (v[r][0] & 3u) != kTailTag) {
                all = false;
                v[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, off[r], 0, kCpSc1);
            }
        if (all || !spin_continue(sp, t0, it, kSpinErrRows)) break;
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += __builtin_bit_cast(double, u32x2{v[r][0] & ~3u, v[r][1]});
    return s;
}

// Grid all-reduce of a layer's NV = 2 M column sums: own row (tagged) -> the 16 rows of this block's group in order -> group sum
// (every member computes it: the same bits; tagged) -> one copy of every group's sum in order -> tot [NV] in LDS.
template <int NV>
__device__ __forceinline__ void tail_allreduce(double mine, double *part, double *gpart, double *tot, SpinCtx &spin, int tid, int tt) {
    const unsigned nblk = gridDim.x, blk = blockIdx.x;
    const __amdgpu_buffer_rsrc_t p_rs = rsrc_of(part), g_rs = rsrc_of(gpart);
    if (tid < NV) {
        __builtin_amdgcn_raw_buffer_store_b64(tail_tag(mine), p_rs, (blk * (uint32_t)NV + (uint32_t)tid) * 8u, 0, kCpSc1);
        const unsigned gbase = blk & ~15u;
        const double s = tail_poll_sum16<NV>(p_rs, [&](int r) { return gbase + r < nblk ? (int)(gbase + r) : -1; }, tid, spin);
        __builtin_amdgcn_raw_buffer_store_b64(tail_tag(s), g_rs, (blk * (uint32_t)NV + (uint32_t)tid) * 8u, 0, kCpSc1);
        TGNN_TT(tt + 3)
        const unsigned n_groups = (nblk + 15u) >> 4;
        tot[tid] = tail_poll_sum16<NV>(g_rs, [&](int g) {          // this block's counterpart in group g (the last group may be short)
            if ((unsigned)g >= n_groups) return -1;
            const unsigned gsize = nblk - 16u * g < 16u ? nblk - 16u * g : 16u;
            return (int)(16u * g + ((blk & 15u) < gsize ? (blk & 15u) : gsize - 1u));
        }, tid, spin);
    }
    __syncthreads();
    TGNN_TT(tt + 4)
}

// A layer's BatchNorm: column sums of the activations in LDS over the block's valid rows, all-reduce, the record [4][256] in LDS,
// the running statistics (block 0)
template <int M>
__device__ __forceinline__ void tail_stats(const TailArgs &A, int l, int rows, const float *actl, float *rec, double *tot, SpinCtx &spin, int tid) {
    const TailLayer &L = A.f[l];
    const int64_t n = A.n;
    double colsum = 0.0;                                          // thread < 2 M: column tid % M, sum | sum of squares
    if (tid < 2 * M) {
        const int ch = tid % M;
        const bool sq = tid >= M;
        for (int r = 0; r < rows; ++r) {
            const double v = (double)actl[r * kTailActLd + ch];
            colsum += sq ? v * v : v;
        }
    }
    TGNN_TT(l * 6 + 1)
    const size_t reg = (size_t)kTailMaxBlocks * tail_row_off(l);
    tail_allreduce<2 * M>(colsum, A.part + reg, A.gpart + reg, tot, spin, tid, l * 6);
    if (tid < M) {                                                // the record, as bn_finalize_kernel writes it
        const double inv_n = 1.0 / (double)n;
        const double mean = tot[tid] * inv_n;
        double var = tot[M + tid] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mh = (float)mean;
        rec[tid] = mh;
        rec[256 + tid] = (float)(mean - (double)mh);
        rec[512 + tid] = (float)((double)L.gamma[tid] / sqrt(var + (double)A.eps));
        rec[768 + tid] = L.beta[tid];
        if (blockIdx.x == 0 && A.update_running && __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
            L.rm[tid] = (float)((1.0 - (double)A.momentum) * (double)L.rm[tid] + (double)A.momentum * mean);
            L.rv[tid] = (float)((1.0 - (double)A.momentum) * (double)L.rv[tid] + (double)A.momentum * unbiased);
            if (tid == 0) *L.nbt += 1;
        }
    }
    __syncthreads();
    TGNN_TT(l * 6 + 5)
}

template <int G>
__device__ __forceinline__ void tail_body(const TailArgs &A, float *lds, SpinCtx &spin, int tid) {
    u32x4 *planes = reinterpret_cast<u32x4 *>(lds);               // [K step 8][G][planes][64] x 16 B
    float *actl = lds;                                            // [G * 16][kTailActLd]: behind a layer's last K step
    float *rec = lds + kTailPlaneBytes / 4;                       // [4][256]
    double *tot = reinterpret_cast<double *>(rec + 1024);         // [512]
    const int64_t n = A.n, n_tiles = (n + 15) / 16;
    const int64_t tile0 = (int64_t)blockIdx.x * A.tiles_per_block;
    const int ng = (int)(n_tiles - tile0 < A.tiles_per_block ? (n_tiles - tile0 > 0 ? n_tiles - tile0 : 0) : A.tiles_per_block);
    const int64_t row0 = tile0 * 16;
    const int rows = n - row0 < (int64_t)ng * 16 ? (int)(n - row0 > 0 ? n - row0 : 0) : ng * 16;     // valid rows (contiguous)
    // (the narrow layers' weight fragments are requested one layer ahead: they arrive during the all-reduce in front of them)
    u32x4 w1[8][1][3], w2[4][1][3], w3[2][1][3];
    tail_first_layer<G>(A, tile0, ng, planes, actl, tid);
    TGNN_TT(0)
    tail_wload_all<128, 8>(A.f[1], w1, tid);
    tail_stats<256>(A, 0, rows, actl, rec, tot, spin, tid);
    tail_narrow_layer<128, G, 8>(A, A.f[1], tile0, ng, rec, planes, actl, tid, w1);
    TGNN_TT(6)
    tail_wload_all<64, 4>(A.f[2], w2, tid);
    tail_stats<128>(A, 1, rows, actl, rec, tot, spin, tid);
    tail_narrow_layer<64, G, 4>(A, A.f[2], tile0, ng, rec, planes, actl, tid, w2);
    TGNN_TT(12)
    tail_wload_all<32, 2>(A.f[3], w3, tid);
    tail_stats<64>(A, 2, rows, actl, rec, tot, spin, tid);
    tail_narrow_layer<32, G, 2>(A, A.f[3], tile0, ng, rec, planes, actl, tid, w3);
    TGNN_TT(18)
    tail_stats<32>(A, 3, rows, actl, rec, tot, spin, tid);
    // final_mlp.1: Linear(32, out_dim) + sigmoid on the vector pipe, k ascending
    for (int it = tid; it < rows * A.out_dim; it += kTailThreads) {
        const int r = it / A.out_dim, o = it - r * A.out_dim;
        float acc = A.b_last[o];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            acc = fmaf(bn_apply1(actl[r * kTailActLd + k], rec[k], rec[256 + k], rec[512 + k], rec[768 + k]), A.w_last[o * 32 + k], acc);
        A.probs[(row0 + r) * A.out_dim + o] = sigmoidf_(acc);
    }
}

__global__ __launch_bounds__(kTailThreads) void forward_tail_mid_kernel(TailArgs A) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    if (A.fault && blockIdx.x == gridDim.x - 1) return;         // (test hook: a block that never shows up)
    if (A.verdict && *A.verdict != 0) return;
    const int tid = threadIdx.x;
    SpinCtx spin{A.err, A.spin_budget, false, A.err_host};
    TGNN_TT(24)
    if (A.tiles_per_block == 4) tail_body<4>(A, lds, spin, tid);
    else if (A.tiles_per_block == 3) tail_body<3>(A, lds, spin, tid);
    else tail_body<2>(A, lds, spin, tid);
    TGNN_TT(25)
}

static std::atomic<int> g_mid_tail{3};                           // bit 0: this kernel; bit 1: the init MLP in the layer loop's prologue
bool mid_init_in_kernel() { return (g_mid_tail.load(std::memory_order_relaxed) & 2) != 0; }

// > 0: tiles per block (2 .. 4) of the tail kernel for this layout (behind the persistent layer loop only); 0: the general final MLP
int mid_tail_tiles_per_block(const tgnn_model_dims *d, int64_t n_nodes, int *blocks_out) {
    if (!(g_mid_tail.load(std::memory_order_relaxed) & 1)) return 0;
    if (d->network_width != 32 || d->network_depth + 1 > 64 || n_nodes < 1 || n_nodes > 65536) return 0;
    static std::atomic<int> capacity[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int cap = capacity[dev].load(std::memory_order_acquire);
    if (cap == 0) {
        static LdsOptIn site;
        int per_cu = 0;
        const hipError_t e1 = opt_in_dynamic_lds(forward_tail_mid_kernel, (int)kTailLdsBytes, site);
        const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, forward_tail_mid_kernel, kTailThreads, kTailLdsBytes);
        if (e1 != hipSuccess || e2 != hipSuccess) return 0;
        cap = per_cu > 0 ? device_cus() : -1;
        capacity[dev].store(cap, std::memory_order_release);
    }
    if (cap <= 0) return 0;
    const int max_blocks = cap < kTailMaxBlocks ? cap : kTailMaxBlocks;
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t k = (n_tiles + max_blocks - 1) / max_blocks;
    if (k < 2) k = 2;
    // one group of <= kTailGroup tiles per block: the activations never leave LDS.  (Larger layouts -- several groups per block, the
    // rows parked in HBM between the layers -- were built and measured (scratch/mid_tail): 20 000 and 32 768 nodes came out level
    // with the general final MLP, whose block-tile kernels fill the device from there on: profiles/r05_mid_tail.txt.)
    if (k > kTailGroup) return 0;
    *blocks_out = (int)((n_tiles + k - 1) / k);
    return (int)k;
}

size_t mid_tail_part_doubles() { return (size_t)kTailMaxBlocks * kTailRowDoubles; }

int launch_forward_tail(const tgnn_model_dims *d, const Params &P, const float *mid, const float *pack, float *probs, double *part,
                        double *gpart, const unsigned *slot_max, const unsigned *w0_max, int64_t n, int tiles_per_block, int blocks,
                        int update_running, float eps, float momentum, hipStream_t s, const int *verdict) {
    TailArgs A{};
    A.verdict = verdict;
    A.mid = mid;
    for (int l = 0; l < 4; ++l) {
        const int pi = P.fin(l);
        const BnPtrs b = P.bn(pi + 2);
        A.f[l] = TailLayer{small_dense_image(pack, d->network_depth, 1 + l), P.f(pi + 1), b.gamma, b.beta, b.rm, b.rv, b.nbt};
    }
    A.w_last = P.f(P.last());
    A.b_last = P.f(P.last() + 1);
    A.probs = probs;
    A.part = part;
    A.gpart = gpart;
    A.slot_max = slot_max;
    A.w0_max = w0_max;
    A.err = spin_error_word();
    A.err_host = spin_error_mirror();
    A.spin_budget = spin_budget_ticks();
    A.fault = 0;
    if (!A.err) {
        set_error("tgnn_forward: the spin-error word of the device could not be allocated");
        return TGNN_ERR_LAUNCH;
    }
    A.n = n;
    A.depth = d->network_depth;
    A.tiles_per_block = tiles_per_block;
    A.out_dim = d->output_dim;
    A.update_running = update_running;
    A.eps = eps;
    A.momentum = momentum;
    TGNN_CHECK_ARG(blocks >= 1 && blocks <= kTailMaxBlocks && tiles_per_block >= 2 && tiles_per_block <= kTailGroup,
                   "blocks / tiles per block of the final MLP's persistent kernel");
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(forward_tail_mid_kernel, (int)kTailLdsBytes, site));
    struct Ctx { TailArgs *A; int blocks; } ctx{&A, blocks};
    const int rc = spin_kernel_chain(s, [](void *c, hipStream_t st) {
        Ctx *x = static_cast<Ctx *>(c);
        forward_tail_mid_kernel<<<dim3(x->blocks), dim3(kTailThreads), kTailLdsBytes, st>>>(*x->A);
    }, &ctx, blocks);
    if (rc != TGNN_OK) return rc;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

}  // namespace tgnn

#ifdef TGNN_TAIL_TIMING
extern "C" int tgnn_debug_tail_timing(unsigned long long *out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgnn::g_tail_timing), (size_t)n_blocks * 32 * sizeof(unsigned long long));
}
#endif
extern "C" int32_t tgnn_set_mid_tail(int32_t on) {
    if (on < 0 || on > 3) return tgnn::g_mid_tail.load();
    return tgnn::g_mid_tail.exchange(on);
}
