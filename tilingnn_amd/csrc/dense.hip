// Dense Linear_trans blocks of TilinGNN on gfx950 matrix cores (K10 / K11 of SURVEY.md section 2b).
//
// Reference: Linear_trans.forward (/root/reference/graph_networks/layers/util.py:31-37) is
// Linear -> activation -> BatchNorm1d(train); MLP (util.py:4-17) stacks them.  On the hot path
// these are the init MLP [Fx->32->32] (TilinGNN.py:31), and the final MLP
// [672->256->128->64->32] + Linear_trans(32->1, Sigmoid) (TilinGNN.py:45-48) -- the only
// genuinely dense GEMMs of the forward (N x 0.43 MFLOP), hence MFMA.
//
// out = act( f(A) . W^T + b )   with f = the previous layer's BatchNorm applied while the A tile is
// staged into LDS (so normalised activations are never written to HBM), and fp64 column sums of
// `out` emitted for the BatchNorm that follows.  fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32: exact f32 (a k-ordered fma chain), 1e-5 parity needs no split tricks.
//
// Tiling: block = 4 waves, tile 128 rows x (32*NT) cols x 32 k, NT in {1,2,4,8} chosen so that one
// block covers the whole output width up to 256 (A is then read from HBM exactly once); wave w owns
// rows [32w, 32w+32) x all columns = NT accumulators.  The next k-tile is prefetched into registers
// while the current one feeds the MFMAs.  LDS rows are padded to 33 floats: the A/B fragment reads
// (lane&31 -> row, lane>>5 -> k) are conflict free.  Blocks are persistent over row tiles so that
// one block = one BN partial row.
#include <stdlib.h>

#include "tgnn_common.h"

namespace tgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kBM = 128, kBK = 32, kLD = 33;

// Where the 32-column K block `kt` of the A operand starts.  A is a sequence of SLOTS of kps K blocks each: block kt
// lives in slot kt / kps at column 32 (kt % kps); slots are `slot_stride` floats apart.  Row-major [N, K]: one slot
// (or kps = 1, stride 32); the slot-major skip buffer [D+1][N][C]: kps = C / 32, stride N C.
__device__ __forceinline__ int64_t a_kblock_offset(int kt, int kps, int64_t slot_stride) {
    return kps == 1 ? (int64_t)kt * slot_stride : (int64_t)(kt / kps) * slot_stride + (int64_t)(kt % kps) * kBK;
}

struct Frag4 {
    float v[4];
};

// FAST = in_dim % 32 == 0 and 16-byte aligned operands: every staging load is an unconditional float4 with
// the row / column index clamped (out-of-range rows are masked at the store).  A predicated load compiles
// to a branch and hipcc drains vmcnt(0) at every join -- the "prefetch" then pays 12 serial HBM round trips
// per k-tile (measured: 47 TF; the same kernel branch-free: see profiles/).
template <int NT, bool FAST>
__global__ __launch_bounds__(256) void dense_mfma_kernel(
    const float *__restrict__ a, int64_t lda, int64_t a_kb_stride, int kps, const float *__restrict__ in_stat,
    const float *__restrict__ w, const float *__restrict__ bias, int64_t n, int in_dim, int out_dim, int act,
    float *__restrict__ out, int64_t ldo, double *__restrict__ bn_partial, int vec_a, int vec_w) {
    constexpr int BN = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [128][33]
    float *Bs = smem + kBM * kLD;      // [BN][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * BN;
    const int ktiles = (in_dim + kBK - 1) / kBK;
    const int64_t row_tiles = (n + kBM - 1) / kBM;
    const int frag_r = lane & 31, frag_k = lane >> 5;
    const int ld_r = tid >> 3, ld_q = tid & 7;   // staging: thread -> (row ld_r + 32 j, float4 column ld_q)

    double csum[NT], csq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) csum[nt] = csq[nt] = 0.0;

    auto load_a = [&](int64_t m0, int kt, Frag4 (&ra)[4]) {
        if (FAST) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int64_t row = m0 + ld_r + 32 * j;
                row = row < n ? row : n - 1;
                const float4 t4 = *reinterpret_cast<const float4 *>(a + a_kblock_offset(kt, kps, a_kb_stride) + row * lda + 4 * ld_q);
                ra[j].v[0] = t4.x; ra[j].v[1] = t4.y; ra[j].v[2] = t4.z; ra[j].v[3] = t4.w;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row = m0 + ld_r + 32 * j;
            const int k = kt * kBK + 4 * ld_q;
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[j].v[e] = 0.f;
            if (row < n) {
                const float *src = a + a_kblock_offset(kt, kps, a_kb_stride) + row * lda + 4 * ld_q;
                if (vec_a && k + 3 < in_dim) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(src);
                    ra[j].v[0] = t4.x; ra[j].v[1] = t4.y; ra[j].v[2] = t4.z; ra[j].v[3] = t4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < in_dim) ra[j].v[e] = src[e];
                }
            }
        }
    };
    auto load_b = [&](int kt, Frag4 (&rb)[NT]) {
        if (FAST) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int col = n0 + ld_r + 32 * j;
                col = col < out_dim ? col : out_dim - 1;
                const float4 t4 = *reinterpret_cast<const float4 *>(w + (int64_t)col * in_dim + kt * kBK + 4 * ld_q);
                rb[j].v[0] = t4.x; rb[j].v[1] = t4.y; rb[j].v[2] = t4.z; rb[j].v[3] = t4.w;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + ld_r + 32 * j, k = kt * kBK + 4 * ld_q;
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[j].v[e] = 0.f;
            if (col < out_dim) {
                const float *src = w + (int64_t)col * in_dim + k;
                if (vec_w && k + 3 < in_dim) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(src);
                    rb[j].v[0] = t4.x; rb[j].v[1] = t4.y; rb[j].v[2] = t4.z; rb[j].v[3] = t4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < in_dim) rb[j].v[e] = src[e];
                }
            }
        }
    };
    auto stage = [&](int64_t m0, int kt, const Frag4 (&ra)[4], const Frag4 (&rb)[NT]) {
        const int k = kt * kBK + 4 * ld_q;
        if (FAST) {
            float mh[4] = {0, 0, 0, 0}, ml[4] = {0, 0, 0, 0}, gg[4] = {1, 1, 1, 1}, bb[4] = {0, 0, 0, 0};
            if (in_stat) {                               // uniform; BatchNorm of the producer, applied on the fly
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mh[e] = in_stat[k + e]; ml[e] = in_stat[in_dim + k + e];
                    gg[e] = in_stat[2 * in_dim + k + e]; bb[e] = in_stat[3 * in_dim + k + e];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    As[(ld_r + 32 * j) * kLD + 4 * ld_q + e] = in_stat ? bn_apply1(ra[j].v[e], mh[e], ml[e], gg[e], bb[e]) : ra[j].v[e];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[(ld_r + 32 * j) * kLD + 4 * ld_q + e] = rb[j].v[e];
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool row_ok = m0 + ld_r + 32 * j < n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = ra[j].v[e];
                if (in_stat && row_ok && k + e < in_dim)     // BatchNorm of the producer, applied on the fly
                    v = bn_apply1(v, in_stat[k + e], in_stat[in_dim + k + e], in_stat[2 * in_dim + k + e],
                                  in_stat[3 * in_dim + k + e]);
                As[(ld_r + 32 * j) * kLD + 4 * ld_q + e] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[(ld_r + 32 * j) * kLD + 4 * ld_q + e] = rb[j].v[e];
    };

    for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
        const int64_t m0 = rt * kBM;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        Frag4 ra[4], rb[NT];
        load_a(m0, 0, ra);
        load_b(0, rb);
        __syncthreads();                       // previous row tile's fragment reads are done
        stage(m0, 0, ra, rb);
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            if (kt + 1 < ktiles) {             // prefetch the next k-tile; it lands while the MFMAs run
                load_a(m0, kt + 1, ra);
                load_b(kt + 1, rb);
            }
            const float *ap = As + (wave * 32 + frag_r) * kLD + frag_k;
            const float *bp = Bs + frag_r * kLD + frag_k;
#pragma unroll
            for (int kk = 0; kk < kBK; kk += 2) {
                const float av = ap[kk];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[nt * 32 * kLD + kk], acc[nt], 0, 0, 0);
            }
            if (kt + 1 < ktiles) {
                __syncthreads();
                stage(m0, kt + 1, ra, rb);
                __syncthreads();
            }
        }
        // ---- epilogue: bias, activation, store, BN partial sums.
        // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + frag_r;
            const bool col_ok = col < out_dim;
            const float b = col_ok ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * frag_k;
                if (col_ok && row < n) {
                    const float v = act_apply(acc[nt][r] + b, act);
                    out[row * ldo + col] = v;
                    csum[nt] += (double)v;
                    csq[nt] += (double)v * (double)v;
                }
            }
        }
    }

    if (bn_partial) {
        __syncthreads();                       // LDS tiles are dead: reuse them for the cross-wave reduction
        double *red = reinterpret_cast<double *>(smem);     // [4 waves][2][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const double s = csum[nt] + __shfl_xor(csum[nt], 32, 64);
            const double q = csq[nt] + __shfl_xor(csq[nt], 32, 64);
            if (lane < 32) {
                red[(wave * 2 + 0) * BN + nt * 32 + lane] = s;
                red[(wave * 2 + 1) * BN + nt * 32 + lane] = q;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, cl = i % BN;
            const int col = n0 + cl;
            if (col < out_dim) {
                const double tot = red[(0 * 2 + which) * BN + cl] + red[(1 * 2 + which) * BN + cl] +
                                   red[(2 * 2 + which) * BN + cl] + red[(3 * 2 + which) * BN + cl];
                bn_partial[(int64_t)blockIdx.x * 2 * out_dim + (int64_t)which * out_dim + col] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 x 3 split-precision variant for the big Linear blocks (in_dim % 32 == 0, out_dim >= 64).
//
// gfx950 runs bf16 MFMA at 16x the fp32 MFMA rate.  Every fp32 operand is split exactly into three bf16
// pieces x = hi + mid + lo (by truncation, 8 + 8 + 8 significant bits: no rounding at all) while its tile is staged
// into LDS, and the product
// is accumulated in fp32 from the six leading cross terms
//        hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi          (dropped terms <= 2^-24 |a||b|)
// on v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 32x32x16 block instead of 8 x 64 for the fp32 MFMA --
// 2.67x fewer matrix-pipe cycles at fp32-class accuracy (measured max-norm error vs the fp64 oracle
// ~2e-7, same as the exact-fp32 kernel; tolerance 1e-5).  Tile 128 x (64*WN) x 32, 4 waves as (4/WN) x WN,
// wave tile (32*TM) x 64; LDS rows are 32 bf16 + 8 pad = 80 B so that the 16-B fragment reads of a
// 16-lane group hit 16 distinct slots.
// ------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int kSplitLd = 40;   // bf16 elements per LDS row (32 used)

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
    split3_trunc(x, hi, mid, lo);           // exact three-way split by truncation, tgnn_common.h
}

// F16: fp16 x 2 instead (tgnn_common.h: split2_f16) -- three cross terms  hi.hi + hi.lo + lo.hi  on v_mfma_f32_32x32x16_f16,
// half the matrix cycles and 24 instead of 44 split instructions per 8 elements, where bounds of both operands are at hand:
// a_max[0 .. n_a_max) / w_max hold max |a| (per slot of the skip buffer, left by the kernels that wrote them) and max |w| as
// float bits; the operands are scaled by the powers of two that bring the maxima just below 2^15 and the accumulators
// un-scaled in the epilogue.  Holds a and w to 2^-22 relative (elements below 2^-39 of the maximum: to 2^-39 of it).
using f16x8 = tgnn_f16x8;
template <int TM, int WN, bool F16>
__global__ __launch_bounds__(256, 2) void dense_split_kernel(
    const float *__restrict__ a, int64_t lda, int64_t a_kb_stride, int kps, const float *__restrict__ in_stat,
    const float *__restrict__ w, const float *__restrict__ bias, int64_t n, int in_dim, int out_dim, int act,
    float *__restrict__ out, int64_t ldo, double *__restrict__ bn_partial, const unsigned *__restrict__ a_max, int n_a_max,
    const unsigned *__restrict__ w_max, int nbx, int nby) {
    constexpr int WM = 4 / WN;                 // waves along M
    constexpr int BM = WM * TM * 32, BN = WN * 64;
    constexpr int RA = BM * 4 / 256, RB = BN * 4 / 256;   // (row, k-octet) items per thread when staging
    constexpr int NP = F16 ? 2 : 3;                       // planes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    __bf16 *As = reinterpret_cast<__bf16 *>(smem_b);                 // [NP][BM][40]  (2-byte elements: bf16 or fp16)
    __bf16 *Bs = As + NP * BM * kSplitLd;                           // [NP][BN][40]
    float sa = 1.0f, sw = 1.0f, unscale = 1.0f;
    if constexpr (F16) {
        unsigned mb = 0;
        for (int i = 0; i < n_a_max; ++i) mb = max(mb, a_max[i]);
        sa = pow2_scale_for(mb, 0);
        sw = pow2_scale_for(*w_max, 0);
        unscale = 1.0f / (sa * sw);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 31, fg = lane >> 5;
    // 1-D grid, nbx row walkers x nby column blocks: the column blocks of one row tile get work-group ids 8 apart -- the same
    // XCD (ids go round the 8 XCDs), dispatched back to back -- so that the A tile comes out of HBM once and out of that XCD's
    // L2 for the others (with the plain (x, y) grid every column block re-read it from HBM: 2 x 269 MB at 100k nodes)
    int bx, by;
    {
        const int id = blockIdx.x;
        if ((nbx & 7) == 0) {
            const int g = id / (8 * nby), r = id % (8 * nby);
            bx = g * 8 + (r & 7);
            by = r >> 3;
        } else {
            bx = id % nbx;
            by = id / nbx;
        }
    }
    const int n0 = by * BN;
    const int ktiles = in_dim / kBK;
    const int64_t row_tiles = (n + BM - 1) / BM;

    double csum[2], csq[2];
    csum[0] = csum[1] = csq[0] = csq[1] = 0.0;

    float4 ra[RA][2], rb[RB][2];
    auto load_tiles = [&](int64_t m0, int kt) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int item = tid + 256 * j, r = item >> 2, o = item & 3;       // row r, k-octet o
            int64_t row = m0 + r;
            row = row < n ? row : n - 1;
            const float4 *p = reinterpret_cast<const float4 *>(a + a_kblock_offset(kt, kps, a_kb_stride) + row * lda + 8 * o);
            ra[j][0] = p[0];
            ra[j][1] = p[1];
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int item = tid + 256 * j, r = item >> 2, o = item & 3;
            int col = n0 + r;
            col = col < out_dim ? col : out_dim - 1;
            const float4 *p = reinterpret_cast<const float4 *>(w + (int64_t)col * in_dim + kt * kBK + 8 * o);
            rb[j][0] = p[0];
            rb[j][1] = p[1];
        }
    };
    auto stage = [&](int kt) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int item = tid + 256 * j, r = item >> 2, o = item & 3;
            float x[8] = {ra[j][0].x, ra[j][0].y, ra[j][0].z, ra[j][0].w, ra[j][1].x, ra[j][1].y, ra[j][1].z, ra[j][1].w};
            if (in_stat) {                           // BatchNorm of the producer, applied on the fly (uniform branch)
                const int k = kt * kBK + 8 * o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    x[e] = bn_apply1(x[e], in_stat[k + e], in_stat[in_dim + k + e], in_stat[2 * in_dim + k + e],
                                     in_stat[3 * in_dim + k + e]);
            }
            if constexpr (F16) {
                f16x8 hi, lo;
#ifdef TGNN_ABL_ASPLIT
                hi = __builtin_bit_cast(f16x8, ra[j][0]); lo = __builtin_bit_cast(f16x8, ra[j][1]);   // (timing ablation: pre-split activations)
#else
                split2_f16(x, sa, hi, lo);
#endif
                *reinterpret_cast<f16x8 *>(As + (0 * BM + r) * kSplitLd + 8 * o) = hi;
                *reinterpret_cast<f16x8 *>(As + (1 * BM + r) * kSplitLd + 8 * o) = lo;
            } else {
                bf16x8 hi, mid, lo;
                split3(x, hi, mid, lo);
                *reinterpret_cast<bf16x8 *>(As + (0 * BM + r) * kSplitLd + 8 * o) = hi;
                *reinterpret_cast<bf16x8 *>(As + (1 * BM + r) * kSplitLd + 8 * o) = mid;
                *reinterpret_cast<bf16x8 *>(As + (2 * BM + r) * kSplitLd + 8 * o) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int item = tid + 256 * j, r = item >> 2, o = item & 3;
            const float x[8] = {rb[j][0].x, rb[j][0].y, rb[j][0].z, rb[j][0].w, rb[j][1].x, rb[j][1].y, rb[j][1].z, rb[j][1].w};
            if constexpr (F16) {
                f16x8 hi, lo;
#ifdef TGNN_ABL_BSPLIT
                hi = __builtin_bit_cast(f16x8, rb[j][0]); lo = __builtin_bit_cast(f16x8, rb[j][1]);   // (timing ablation: a pre-split image)
#else
                split2_f16(x, sw, hi, lo);
#endif
                *reinterpret_cast<f16x8 *>(Bs + (0 * BN + r) * kSplitLd + 8 * o) = hi;
                *reinterpret_cast<f16x8 *>(Bs + (1 * BN + r) * kSplitLd + 8 * o) = lo;
            } else {
                bf16x8 hi, mid, lo;
                split3(x, hi, mid, lo);
                *reinterpret_cast<bf16x8 *>(Bs + (0 * BN + r) * kSplitLd + 8 * o) = hi;
                *reinterpret_cast<bf16x8 *>(Bs + (1 * BN + r) * kSplitLd + 8 * o) = mid;
                *reinterpret_cast<bf16x8 *>(Bs + (2 * BN + r) * kSplitLd + 8 * o) = lo;
            }
        }
    };

    for (int64_t rt = bx; rt < row_tiles; rt += nbx) {
        const int64_t m0 = rt * BM;
        f32x16 acc[TM][2];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
        load_tiles(m0, 0);
        __syncthreads();
        stage(0);
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            if (kt + 1 < ktiles) load_tiles(m0, kt + 1);          // lands while the MFMAs run
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {                      // two 16-wide k blocks per tile
                bf16x8 af[TM][NP], bfr[2][NP];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        af[tm][pl] = *reinterpret_cast<const bf16x8 *>(
                            As + (pl * BM + (wm * TM + tm) * 32 + fi) * kSplitLd + kb * 16 + 8 * fg);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        bfr[tn][pl] = *reinterpret_cast<const bf16x8 *>(
                            Bs + (pl * BN + wn * 64 + tn * 32 + fi) * kSplitLd + kb * 16 + 8 * fg);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        f32x16 c = acc[tm][tn];
                        if constexpr (F16) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, af[tm][0]), al = __builtin_bit_cast(f16x8, af[tm][NP - 1]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, bfr[tn][0]), bl = __builtin_bit_cast(f16x8, bfr[tn][NP - 1]);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);   // lo . hi
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);   // hi . lo
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);   // hi . hi
                            acc[tm][tn] = c;
                            continue;
                        }
                        // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][NP - 1], bfr[tn][0], c, 0, 0, 0);   // lo . hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bfr[tn][NP - 1], c, 0, 0, 0);   // hi . lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][1], bfr[tn][1], c, 0, 0, 0);   // mid . mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][1], bfr[tn][0], c, 0, 0, 0);   // mid . hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bfr[tn][1], c, 0, 0, 0);   // hi . mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][0], bfr[tn][0], c, 0, 0, 0);   // hi . hi
                        acc[tm][tn] = c;
                    }
            }
            if (kt + 1 < ktiles) {
                __syncthreads();
                stage(kt + 1);
                __syncthreads();
            }
        }
        // ---- epilogue (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + tn * 32 + fi;
            const bool col_ok = col < out_dim;
            const float b = col_ok ? bias[col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
                    if (col_ok && row < n) {
                        const float v = act_apply(F16 ? fmaf(acc[tm][tn][r], unscale, b) : acc[tm][tn][r] + b, act);
                        out[row * ldo + col] = v;
                        csum[tn] += (double)v;
                        csq[tn] += (double)v * (double)v;
                    }
                }
        }
    }

    if (bn_partial) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(smem_b);   // [WM][2][BN]
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const double s_ = csum[tn] + __shfl_xor(csum[tn], 32, 64);
            const double q_ = csq[tn] + __shfl_xor(csq[tn], 32, 64);
            if (lane < 32) {
                red[(wm * 2 + 0) * BN + wn * 64 + tn * 32 + lane] = s_;
                red[(wm * 2 + 1) * BN + wn * 64 + tn * 32 + lane] = q_;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, cl = i % BN, col = n0 + cl;
            if (col < out_dim) {
                double tot = 0.0;
#pragma unroll
                for (int m = 0; m < WM; ++m) tot += red[(m * 2 + which) * BN + cl];
                bn_partial[(int64_t)bx * 2 * out_dim + (int64_t)which * out_dim + col] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// [r4] fp16-pair Linear for MANY rows and out_dim = 32 TN <= 256 (the final MLP's 672 -> 256 -> 128 -> 64): every wave owns
// 32 rows x ALL output columns.
//  * A never goes through LDS: lane (i, g) of the v_mfma_f32_32x32x16_f16 A operand needs k = 8 g .. 8 g + 7 of row i -- 32
//    bytes of the fp32 row, read straight from memory two k-tiles ahead and split ONCE (dense_split_kernel splits every A
//    tile once per column block and once more per barrier-separated stage; its two barriers per k-tile with the loads of a
//    single k-tile in flight left the matrix pipe idle 3/4 of the time: 188 us against 41 us of fp16 matrix work).
//  * W comes as a pre-split operand image in fragment order (dense_f16_image_kernel, once per forward: [k-tile][kb][tn][plane]
//    [lane] x 16 B), copied through registers into a double-buffered LDS tile -- ONE barrier per k-tile, conflict-free
//    ds_read_b128, no split work on it.
//  * MFMA order per accumulator as in dense_split_kernel<.., true> (lo.hi, hi.lo, hi.hi over k ascending): the same bits.
// Persistent over 128-row tiles, one BatchNorm partial row per block (column sums kept in 2 x TN / 2 doubles per lane).
// ------------------------------------------------------------------------------------------
using u32x4_ = __attribute__((ext_vector_type(4))) unsigned int;
using f32x4_ = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int kRowsThreads = 256;
__global__ __launch_bounds__(256) void dense_f16_image_kernel(const float *__restrict__ w, int in_dim, int tn_count,
                                                              const unsigned *__restrict__ w_max, u32x4_ *__restrict__ wimg) {
    const int ktiles = in_dim / kBK;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= ktiles * 2 * tn_count * 64) return;
    const int lane = item & 63, t = item >> 6, tn = t % tn_count, kb = (t / tn_count) & 1, kt = t / (2 * tn_count);
    const int col = tn * 32 + (lane & 31), k0 = kt * kBK + kb * 16 + 8 * (lane >> 5);
    const float sw = pow2_scale_for(*w_max, 0);
    const float4 *p = reinterpret_cast<const float4 *>(w + (int64_t)col * in_dim + k0);
    const float4 x0 = p[0], x1 = p[1];
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    f16x8 hi, lo;
    split2_f16(x, sw, hi, lo);
    const int64_t o = ((int64_t)(kt * 2 + kb) * tn_count + tn) * 128 + lane;
    wimg[o] = __builtin_bit_cast(u32x4_, hi);
    wimg[o + 64] = __builtin_bit_cast(u32x4_, lo);
}

template <int TN, bool STAT>
__global__ __launch_bounds__(kRowsThreads, 2) void dense_f16_rows_kernel(
    const float *__restrict__ a, int64_t lda, int64_t a_kb_stride, int kps, const float *__restrict__ in_stat,
    const u32x4_ *__restrict__ wimg, const float *__restrict__ bias, int64_t n, int in_dim, int act, float *__restrict__ out,
    int64_t ldo, double *__restrict__ bn_partial, const unsigned *__restrict__ a_max, int n_a_max,
    const unsigned *__restrict__ w_max, GinFin fin, double *__restrict__ fold_rows) {
    constexpr int N = 32 * TN, TNH = TN / 2;
    constexpr int kTileVec = 2 * TN * 2 * 64;                 // 16-byte pieces of one k-tile of the image
    constexpr int RB = kTileVec / kRowsThreads;               // ... per thread
    __shared__ __attribute__((aligned(1024))) u32x4_ Bs0[kTileVec];
    __shared__ __attribute__((aligned(1024))) u32x4_ Bs1[kTileVec];
    extern __shared__ __attribute__((aligned(16))) float st[];    // [4][in_dim] (STAT)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fg = lane >> 5;
    const int ktiles = in_dim / kBK;
    float sa, unscale;
    {
        unsigned mb = 0;
        for (int i = lane; i < n_a_max; i += 64) mb = max(mb, a_max[i]);   // (one round trip, not one per slot)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d, 64));
        sa = pow2_scale_for(mb, 0);
        unscale = 1.0f / (sa * pow2_scale_for(*w_max, 0));
    }
    if (STAT) {
        for (int i = tid; i < 4 * in_dim; i += kRowsThreads) st[i] = in_stat[i];
    }
    const int64_t row_tiles = (n + 127) / 128;
    // BatchNorm sums of the block: thread t keeps entries t and t + 256 of the row [sum N | sum of squares N], added up tile by
    // tile from the waves' column sums (through the LDS tile the last k-step does not read)
    double bsum[2] = {0.0, 0.0};
    double *red = reinterpret_cast<double *>((ktiles & 1) ? Bs1 : Bs0);   // [wave][2][N]
    static_assert((size_t)4 * 2 * N * sizeof(double) <= sizeof(Bs0), "the reduction array lives in one tile");
    const bool leaky = act == TGNN_ACT_LEAKY_RELU;

    for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
        const int64_t m0 = rt * 128 + wave * 32;
        int64_t row = m0 + fi;
        row = row < n ? row : n - 1;
        const float *arow = a + row * lda + 8 * fg;
        auto load_a = [&](int kt, float4 (&r)[4]) {
            const float4 *p = reinterpret_cast<const float4 *>(arow + a_kblock_offset(kt, kps, a_kb_stride));
            r[0] = p[0]; r[1] = p[1]; r[2] = p[4]; r[3] = p[5];        // k = 8 g .. + 7 and 16 + 8 g .. + 7
        };
        // image k-tile -> LDS by DMA, 1 KB per wave-level instruction, no register in between
        auto dma_b = [&](int kt, u32x4_ *dst) {
            const u32x4_ *p = wimg + (int64_t)kt * kTileVec + tid;
#pragma unroll
            for (int j = 0; j < RB; ++j)
                __builtin_amdgcn_global_load_lds(p + kRowsThreads * j, (lds_void_t *)(dst + kRowsThreads * j + wave * 64), 16, 0, 0);
        };
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
        float4 abuf0[4], abuf1[4];                            // A(kt) of even / odd k-tiles: no copies between them
        load_a(0, abuf0);
        __syncthreads();                                      // the previous row tile's last reads of the tiles (and st's fill) are over
        dma_b(0, Bs0);
        // One k-tile: A(kt) -> fp16 pairs; tile kt has landed (the fence of __syncthreads waits for the copy); the next tile's copy
        // and A rows are issued and land while this tile is multiplied.  (Everything compiler-visible: hand-counted vmcnt /
        // lgkmcnt with inline-asm loads -- A two tiles ahead, B fragments four steps ahead, accumulators interleaved -- measured
        // the same 155-160 us and hipcc copies asm-loaded registers at control-flow merges before the data has landed.)
        auto step = [&](int kt, const u32x4_ *bcur, u32x4_ *bnxt, const float4 (&acur)[4], float4 (&anxt)[4]) {
            f16x8 ah[2], al[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float x[8] = {acur[2 * kb].x, acur[2 * kb].y, acur[2 * kb].z, acur[2 * kb].w,
                              acur[2 * kb + 1].x, acur[2 * kb + 1].y, acur[2 * kb + 1].z, acur[2 * kb + 1].w};
                if (STAT) {
                    const int k = kt * kBK + kb * 16 + 8 * fg;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = bn_apply1(x[e], st[k + e], st[in_dim + k + e], st[2 * in_dim + k + e], st[3 * in_dim + k + e]);
                }
                split2_f16(x, sa, ah[kb], al[kb]);
            }
            __syncthreads();                                  // tile kt is complete, the other one is free
            if (kt + 1 < ktiles) {
                dma_b(kt + 1, bnxt);
                load_a(kt + 1, anxt);
            }
            const u32x4_ *bt = bcur + lane;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const f16x8 bh = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128]);
                    const f16x8 bl = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128 + 64]);
                    f32x16 c = acc[tn];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb], bh, c, 0, 0, 0);   // lo . hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], bl, c, 0, 0, 0);   // hi . lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], bh, c, 0, 0, 0);   // hi . hi
                    acc[tn] = c;
                }
        };
        for (int kt = 0; kt < ktiles; kt += 2) {
            step(kt, Bs0, Bs1, abuf0, abuf1);
            if (kt + 1 < ktiles) step(kt + 1, Bs1, Bs0, abuf1, abuf0);
        }
        // ---- epilogue (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = tn * 32 + fi;
            const float b = bias[col];
            double s_ = 0.0, q_ = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * fg;
                if (orow < n) {
                    const float u = fmaf(acc[tn][r], unscale, b);
                    const float v = leaky ? (u >= 0.f ? u : u * kLeakySlope) : act_apply(u, act);
                    out[orow * ldo + col] = v;
                    s_ += (double)v;
                    q_ += (double)v * (double)v;
                }
            }
            s_ += __shfl_xor(s_, 32, 64);
            q_ += __shfl_xor(q_, 32, 64);
            if (bn_partial && (tn / TNH) == fg) {             // (lanes < 32 write the columns of tn < TN / 2, the others the rest)
                red[(wave * 2 + 0) * N + col] = s_;
                red[(wave * 2 + 1) * N + col] = q_;
            }
        }
        if (bn_partial) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = tid + kRowsThreads * h;
                if (i < 2 * N) {
                    const int which = i / N, cl = i % N;
                    double tot = 0.0;
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) tot += red[(wv * 2 + which) * N + cl];
                    bsum[h] += tot;
                }
            }
        }
    }

    if (bn_partial) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + kRowsThreads * h;
            if (i < 2 * N) {
                if (fin.counter) st_partial_sc1(bn_partial, (int64_t)blockIdx.x * 2 * N + i, bsum[h]);   // (another block of this launch reads it)
                else bn_partial[(int64_t)blockIdx.x * 2 * N + i] = bsum[h];
            }
        }
        // [r6] the BatchNorm's record by the producer itself, in two levels (no bn_finalize launch behind the kernel); the dead image
        // tiles are the fold's LDS
        if (fin.counter) bn_fold_two_level<N>(fin, bn_partial, fold_rows, reinterpret_cast<double *>(Bs0), reinterpret_cast<unsigned *>(Bs1));
    }
}

// ------------------------------------------------------------------------------------------
// [r6] The same Linear, ONE wave per SIMD with the whole register file (one 4-wave block per CU, up to 512 registers per lane):
// dense_f16_rows_kernel above runs 2 waves per SIMD on 253 registers and hipcc has no room to move a load: per accumulator it
// emits "2 ds_read_b128, s_waitcnt lgkmcnt(0), 3 dependent matrix instructions", and per k-tile one "s_waitcnt vmcnt(0)" on
// loads issued a single k-tile (~0.7 us) earlier -- 166 us at 100 000 rows against 41 us of matrix work.  Here
//  * everything that comes from memory is in flight for TWO k-tiles: the A rows and the W image tile travel to registers (plain,
//    compiler-counted loads; the image tile is stored to LDS one k-tile before it is read) -- no LDS-DMA, hence no vmcnt(0) in
//    front of a barrier;
//  * the matrix instructions of a k-half go term-major (lo.hi over all TN accumulators, then hi.lo, then hi.hi): eight
//    independent instructions between two on the same accumulator, all operand fragments of the half in registers before
//    the first one (the next half's reads are issued behind them);
//  * per accumulator the order of the terms is the one of the kernels above: the same bits.
// ------------------------------------------------------------------------------------------
template <int TN, bool STAT>
__global__ __launch_bounds__(kRowsThreads, 1) void dense_f16_rows2_kernel(
    const float *__restrict__ a, int64_t lda, int64_t a_kb_stride, int kps, const float *__restrict__ in_stat,
    const u32x4_ *__restrict__ wimg, const float *__restrict__ bias, int64_t n, int in_dim, int act, float *__restrict__ out,
    int64_t ldo, double *__restrict__ bn_partial, const unsigned *__restrict__ a_max, int n_a_max,
    const unsigned *__restrict__ w_max) {
    constexpr int N = 32 * TN, TNH = TN / 2;
    constexpr int kTileVec = 2 * TN * 2 * 64;                 // 16-byte pieces of one k-tile of the image
    constexpr int RB = kTileVec / kRowsThreads;               // ... per thread
    // dynamic LDS (more than the 64 KB a kernel may declare statically): two image tiles, the column sums of the four waves, the
    // BatchNorm record of the input (STAT)
    extern __shared__ __attribute__((aligned(1024))) unsigned char rows2_lds[];
    u32x4_ (*Bs)[kTileVec] = reinterpret_cast<u32x4_ (*)[kTileVec]>(rows2_lds);
    double *red = reinterpret_cast<double *>(rows2_lds + sizeof(u32x4_) * 2 * kTileVec);                       // [wave][2][N]
    float *st = reinterpret_cast<float *>(rows2_lds + sizeof(u32x4_) * 2 * kTileVec + sizeof(double) * 8 * N);   // [4][in_dim]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fg = lane >> 5;
    const int ktiles = in_dim / kBK;
    float sa, unscale;
    {
        unsigned mb = 0;
        for (int i = lane; i < n_a_max; i += 64) mb = max(mb, a_max[i]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d, 64));
        sa = pow2_scale_for(mb, 0);
        unscale = 1.0f / (sa * pow2_scale_for(*w_max, 0));
    }
    if (STAT) {
        for (int i = tid; i < 4 * in_dim; i += kRowsThreads) st[i] = in_stat[i];
    }
    const int64_t row_tiles = (n + 127) / 128;
    double bsum[2] = {0.0, 0.0};
    const bool leaky = act == TGNN_ACT_LEAKY_RELU;

    for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
        const int64_t m0 = rt * 128 + wave * 32;
        int64_t row = m0 + fi;
        row = row < n ? row : n - 1;
        const float *arow = a + row * lda + 8 * fg;
        auto load_a = [&](int kt, float4 (&r)[4]) {
            const float4 *p = reinterpret_cast<const float4 *>(arow + (int64_t)kt * a_kb_stride);   // (kps == 1: the launcher's condition)
            r[0] = p[0]; r[1] = p[1]; r[2] = p[4]; r[3] = p[5];        // k = 8 g .. + 7 and 16 + 8 g .. + 7
        };
        auto load_b = [&](int kt, u32x4_ (&r)[RB]) {
            const u32x4_ *p = wimg + (int64_t)kt * kTileVec + tid;
#pragma unroll
            for (int j = 0; j < RB; ++j) r[j] = p[kRowsThreads * j];
        };
        auto store_b = [&](const u32x4_ (&r)[RB], u32x4_ *dst) {
#pragma unroll
            for (int j = 0; j < RB; ++j) dst[tid + kRowsThreads * j] = r[j];
        };
        auto split_a = [&](int kt, const float4 (&r)[4], f16x8 (&hi)[2], f16x8 (&lo)[2]) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float x[8] = {r[2 * kb].x, r[2 * kb].y, r[2 * kb].z, r[2 * kb].w,
                              r[2 * kb + 1].x, r[2 * kb + 1].y, r[2 * kb + 1].z, r[2 * kb + 1].w};
                if (STAT) {
                    const int k = kt * kBK + kb * 16 + 8 * fg;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = bn_apply1(x[e], st[k + e], st[in_dim + k + e], st[2 * in_dim + k + e], st[3 * in_dim + k + e]);
                }
                split2_f16(x, sa, hi[kb], lo[kb]);
            }
        };
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
        // registers of the pipeline, by the parity of the k-tile they hold
        float4 araw0[4], araw1[4];
        u32x4_ breg0[RB], breg1[RB];
        f16x8 ah0[2], al0[2], ah1[2], al1[2];
        const int last = ktiles - 1;
        auto clampk = [&](int kt) { return kt < last ? kt : last; };
        load_a(0, araw0);
        load_b(0, breg0);
        load_a(clampk(1), araw1);
        load_b(clampk(1), breg1);
        __syncthreads();                                      // the previous row tile's reads of Bs (and st's fill) are over
        store_b(breg0, Bs[0]);
        split_a(0, araw0, ah0, al0);
        load_a(clampk(2), araw0);
        load_b(clampk(2), breg0);
        // Branch-free inside the k loop (an accumulator modified on two paths of a loop costs hipcc a register copy per element and
        // iteration): loads past the last tile re-read the last tile, the store / split of a tile nobody multiplies are harmless.
        auto mma = [&](const u32x4_ *bcur, const f16x8 (&ahc)[2], const f16x8 (&alc)[2]) {
            const u32x4_ *bt = bcur + lane;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f16x8 bh[TN], bl[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    bh[tn] = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128]);
                    bl[tn] = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128 + 64]);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alc[kb], bh[tn], acc[tn], 0, 0, 0);   // lo . hi
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc[kb], bl[tn], acc[tn], 0, 0, 0);   // hi . lo
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc[kb], bh[tn], acc[tn], 0, 0, 0);   // hi . hi
            }
        };
        const int pairs = ktiles >> 1;
        for (int kp = 0; kp < pairs; ++kp) {
            const int kt = 2 * kp;
            // step kt: tile kt is in Bs[0] behind the barrier, A(kt) split in (ah0, al0); tile kt + 1 goes registers -> Bs[1] and is
            // split, its registers take tile kt + 3; then the 6 TN matrix instructions of tile kt
            __syncthreads();
            store_b(breg1, Bs[1]);
            split_a(clampk(kt + 1), araw1, ah1, al1);
            load_a(clampk(kt + 3), araw1);
            load_b(clampk(kt + 3), breg1);
            mma(Bs[0], ah0, al0);
            // step kt + 1 the other way round
            __syncthreads();
            store_b(breg0, Bs[0]);
            split_a(clampk(kt + 2), araw0, ah0, al0);
            load_a(clampk(kt + 4), araw0);
            load_b(clampk(kt + 4), breg0);
            mma(Bs[1], ah1, al1);
        }
        if (ktiles & 1) {                                     // (uniform) the last tile of an odd count: in Bs[0], split in (ah0, al0)
            __syncthreads();
            mma(Bs[0], ah0, al0);
        }
        // ---- epilogue (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = tn * 32 + fi;
            const float b = bias[col];
            double s_ = 0.0, q_ = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * fg;
                if (orow < n) {
                    const float u = fmaf(acc[tn][r], unscale, b);
                    const float v = leaky ? (u >= 0.f ? u : u * kLeakySlope) : act_apply(u, act);
                    out[orow * ldo + col] = v;
                    s_ += (double)v;
                    q_ += (double)v * (double)v;
                }
            }
            s_ += __shfl_xor(s_, 32, 64);
            q_ += __shfl_xor(q_, 32, 64);
            if (bn_partial && (tn / TNH) == fg) {             // (lanes < 32 write the columns of tn < TN / 2, the others the rest)
                red[(wave * 2 + 0) * N + col] = s_;
                red[(wave * 2 + 1) * N + col] = q_;
            }
        }
        if (bn_partial) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = tid + kRowsThreads * h;
                if (i < 2 * N) {
                    const int which = i / N, cl = i % N;
                    double tot = 0.0;
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) tot += red[(wv * 2 + which) * N + cl];
                    bsum[h] += tot;
                }
            }
        }
    }

    if (bn_partial) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + kRowsThreads * h;
            if (i < 2 * N) bn_partial[(int64_t)blockIdx.x * 2 * N + i] = bsum[h];
        }
    }
}

// ------------------------------------------------------------------------------------------
// [r6] dense_f16_rows2_kernel's pipeline on HALF the output columns per work item (item = (128-row tile, column half): a wave's
// accumulators are 64 registers, the block's image tile 16 KB), two or three blocks per CU: the one-block-per-CU form loses to
// the bubbles at every row tile's start and end (nothing else runs on the CU meanwhile).  The A rows of a tile are read and
// split by both of its items (the second read comes out of L2: the two run on neighbouring blocks at the same time).
// ------------------------------------------------------------------------------------------
template <int TNI, int TN>                                    // TNI 32-column blocks in the image, TN of them per work item
__global__ __launch_bounds__(kRowsThreads, 2) void dense_f16_halves_kernel(
    const float *__restrict__ a, int64_t lda, int64_t a_kb_stride, int kps, const float *__restrict__ in_stat,
    const u32x4_ *__restrict__ wimg, const float *__restrict__ bias, int64_t n, int in_dim, int act, float *__restrict__ out,
    int64_t ldo, double *__restrict__ bn_partial, const unsigned *__restrict__ a_max, int n_a_max,
    const unsigned *__restrict__ w_max) {
    constexpr bool STAT = false;
    constexpr int N = 32 * TN, TNH = TN / 2, NI = 32 * TNI, PARTS = TNI / TN;
    constexpr int kTileVec = 2 * TN * 2 * 64;                 // 16-byte pieces of one k-tile of the image, this item's columns
    constexpr int kImgTileVec = 2 * TNI * 2 * 64;             // ... all columns
    constexpr int RB = kTileVec / kRowsThreads;               // ... per thread
    // dynamic LDS (more than the 64 KB a kernel may declare statically): two image tiles, the column sums of the four waves, the
    // BatchNorm record of the input (STAT)
    extern __shared__ __attribute__((aligned(1024))) unsigned char halves_lds[];
    u32x4_ (*Bs)[kTileVec] = reinterpret_cast<u32x4_ (*)[kTileVec]>(halves_lds);
    double *red = reinterpret_cast<double *>(halves_lds + sizeof(u32x4_) * 2 * kTileVec);                       // [wave][2][N]
    float *st = reinterpret_cast<float *>(halves_lds + sizeof(u32x4_) * 2 * kTileVec + sizeof(double) * 8 * N);   // [4][in_dim]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fg = lane >> 5;
    const int ktiles = in_dim / kBK;
    float sa, unscale;
    {
        unsigned mb = 0;
        for (int i = lane; i < n_a_max; i += 64) mb = max(mb, a_max[i]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d, 64));
        sa = pow2_scale_for(mb, 0);
        unscale = 1.0f / (sa * pow2_scale_for(*w_max, 0));
    }
    if (STAT) {
        for (int i = tid; i < 4 * in_dim; i += kRowsThreads) st[i] = in_stat[i];
    }
    const int64_t row_tiles = (n + 127) / 128, items = row_tiles * PARTS;
    // BatchNorm sums of the block: thread t keeps entries t, t + 256, .. of the row [sum NI | sum of squares NI]
    constexpr int BS = 2 * NI / kRowsThreads;
    double bsum[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) bsum[e] = 0.0;
    const bool leaky = act == TGNN_ACT_LEAKY_RELU;

    for (int64_t it = blockIdx.x; it < items; it += gridDim.x) {
        const int64_t rt = it / PARTS;
        const int part = (int)(it % PARTS);                   // columns 32 TN part .. + 32 TN
        const int64_t m0 = rt * 128 + wave * 32;
        int64_t row = m0 + fi;
        row = row < n ? row : n - 1;
        const float *arow = a + row * lda + 8 * fg;
        auto load_a = [&](int kt, float4 (&r)[4]) {
            const float4 *p = reinterpret_cast<const float4 *>(arow + (int64_t)kt * a_kb_stride);   // (kps == 1: the launcher's condition)
            r[0] = p[0]; r[1] = p[1]; r[2] = p[4]; r[3] = p[5];        // k = 8 g .. + 7 and 16 + 8 g .. + 7
        };
        auto load_b = [&](int kt, u32x4_ (&r)[RB]) {
            // the item's columns of image tile kt: per k-half a run of TN * 128 pieces at (kb TNI + part TN) * 128
            const u32x4_ *p = wimg + (int64_t)kt * kImgTileVec;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int i = tid + kRowsThreads * j, kb = i / (TN * 128), rem = i % (TN * 128);
                r[j] = p[(kb * TNI + part * TN) * 128 + rem];
            }
        };
        auto store_b = [&](const u32x4_ (&r)[RB], u32x4_ *dst) {
#pragma unroll
            for (int j = 0; j < RB; ++j) dst[tid + kRowsThreads * j] = r[j];
        };
        auto split_a = [&](int kt, const float4 (&r)[4], f16x8 (&hi)[2], f16x8 (&lo)[2]) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float x[8] = {r[2 * kb].x, r[2 * kb].y, r[2 * kb].z, r[2 * kb].w,
                              r[2 * kb + 1].x, r[2 * kb + 1].y, r[2 * kb + 1].z, r[2 * kb + 1].w};
                if (STAT) {
                    const int k = kt * kBK + kb * 16 + 8 * fg;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = bn_apply1(x[e], st[k + e], st[in_dim + k + e], st[2 * in_dim + k + e], st[3 * in_dim + k + e]);
                }
                split2_f16(x, sa, hi[kb], lo[kb]);
            }
        };
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
        // registers of the pipeline, by the parity of the k-tile they hold
        float4 araw0[4], araw1[4];
        u32x4_ breg0[RB], breg1[RB];
        f16x8 ah0[2], al0[2], ah1[2], al1[2];
        const int last = ktiles - 1;
        auto clampk = [&](int kt) { return kt < last ? kt : last; };
        load_a(0, araw0);
        load_b(0, breg0);
        load_a(clampk(1), araw1);
        load_b(clampk(1), breg1);
        __syncthreads();                                      // the previous row tile's reads of Bs (and st's fill) are over
        store_b(breg0, Bs[0]);
        split_a(0, araw0, ah0, al0);
        load_a(clampk(2), araw0);
        load_b(clampk(2), breg0);
        // Branch-free inside the k loop (an accumulator modified on two paths of a loop costs hipcc a register copy per element and
        // iteration): loads past the last tile re-read the last tile, the store / split of a tile nobody multiplies are harmless.
        auto mma = [&](const u32x4_ *bcur, const f16x8 (&ahc)[2], const f16x8 (&alc)[2]) {
            const u32x4_ *bt = bcur + lane;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f16x8 bh[TN], bl[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    bh[tn] = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128]);
                    bl[tn] = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128 + 64]);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alc[kb], bh[tn], acc[tn], 0, 0, 0);   // lo . hi
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc[kb], bl[tn], acc[tn], 0, 0, 0);   // hi . lo
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahc[kb], bh[tn], acc[tn], 0, 0, 0);   // hi . hi
            }
        };
        const int pairs = ktiles >> 1;
        for (int kp = 0; kp < pairs; ++kp) {
            const int kt = 2 * kp;
            // step kt: tile kt is in Bs[0] behind the barrier, A(kt) split in (ah0, al0); tile kt + 1 goes registers -> Bs[1] and is
            // split, its registers take tile kt + 3; then the 6 TN matrix instructions of tile kt
            __syncthreads();
            store_b(breg1, Bs[1]);
            split_a(clampk(kt + 1), araw1, ah1, al1);
            load_a(clampk(kt + 3), araw1);
            load_b(clampk(kt + 3), breg1);
            mma(Bs[0], ah0, al0);
            // step kt + 1 the other way round
            __syncthreads();
            store_b(breg0, Bs[0]);
            split_a(clampk(kt + 2), araw0, ah0, al0);
            load_a(clampk(kt + 4), araw0);
            load_b(clampk(kt + 4), breg0);
            mma(Bs[1], ah1, al1);
        }
        if (ktiles & 1) {                                     // (uniform) the last tile of an odd count: in Bs[0], split in (ah0, al0)
            __syncthreads();
            mma(Bs[0], ah0, al0);
        }
        // ---- epilogue (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = (part * TN + tn) * 32 + fi;
            const float b = bias[col];
            double s_ = 0.0, q_ = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * fg;
                if (orow < n) {
                    const float u = fmaf(acc[tn][r], unscale, b);
                    const float v = leaky ? (u >= 0.f ? u : u * kLeakySlope) : act_apply(u, act);
                    out[orow * ldo + col] = v;
                    s_ += (double)v;
                    q_ += (double)v * (double)v;
                }
            }
            s_ += __shfl_xor(s_, 32, 64);
            q_ += __shfl_xor(q_, 32, 64);
            if (bn_partial && (tn / TNH) == fg) {             // (lanes < 32 write the columns of tn < TN / 2, the others the rest)
                red[(wave * 2 + 0) * N + tn * 32 + fi] = s_;
                red[(wave * 2 + 1) * N + tn * 32 + fi] = q_;
            }
        }
        if (bn_partial) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < BS; ++e) {
                const int i = tid + kRowsThreads * e, which = i / NI, cl = i % NI;
                if (cl / N == part) {
                    double tot = 0.0;
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) tot += red[(wv * 2 + which) * N + cl % N];
                    bsum[e] += tot;
                }
            }
        }
    }

    if (bn_partial) {
#pragma unroll
        for (int e = 0; e < BS; ++e) bn_partial[(int64_t)blockIdx.x * 2 * NI + tid + kRowsThreads * e] = bsum[e];
    }
}

// ------------------------------------------------------------------------------------------
// [r6] The final MLP's inner Linears (256 -> 128 -> 64, BatchNorm on load) with the WHOLE operand image of W resident in LDS
// (128 KB / 32 KB: one 8-wave block per CU, the image copied once per block) -- nothing is shared between the waves after
// that, so the k loop has NO barrier: every wave streams its own 32-row tiles, two k-tiles (8 KB) per request, two requests in
// flight (128 KB per CU).  These layers are HBM-bound (154 + 77 MB at 100 000 rows); dense_f16_rows_kernel waits for every k-tile's rows one
// tile after asking for them -- 8 / 4 round trips per row tile with 4 KB per wave in flight: 55.6 + 27.1 us, 2.8 TB/s.
// Per accumulator the matrix terms come in the rows kernels' order (k ascending; lo.hi, hi.lo, hi.hi): the same bits.
// One BatchNorm partial row per block (column sums kept per lane in fp64 over all of a wave's tiles).
// ------------------------------------------------------------------------------------------
constexpr int kResThreads = 512;
template <int TN, int CPT, int kResChunk = 2>                 // kResChunk k-tiles per request (2: 8 KB per wave; two requests in flight), CPT
                                                              // requests per row tile (even): in_dim = 32 kResChunk CPT
__global__ __launch_bounds__(kResThreads, 1) void dense_f16_resident_kernel(
    const float *__restrict__ a, int64_t lda, const float *__restrict__ in_stat, const u32x4_ *__restrict__ wimg,
    const float *__restrict__ bias, int64_t n, int act, float *__restrict__ out, int64_t ldo, double *__restrict__ bn_partial,
    const unsigned *__restrict__ a_max, const unsigned *__restrict__ w_max, GinFin fin, double *__restrict__ fold_rows) {
    constexpr int N = 32 * TN, K = 32 * kResChunk * CPT, KT = K / kBK;
    constexpr int kTileVec = 2 * TN * 2 * 64;                 // 16-byte pieces of one k-tile of the image
    extern __shared__ __attribute__((aligned(1024))) unsigned char res_lds[];
    u32x4_ *Bs = reinterpret_cast<u32x4_ *>(res_lds);                                   // [KT][kTileVec]
    float *st = reinterpret_cast<float *>(res_lds + sizeof(u32x4_) * KT * kTileVec);    // [4][K]
    double *red = reinterpret_cast<double *>(res_lds + sizeof(u32x4_) * KT * kTileVec + sizeof(float) * 4 * K);   // [8 waves][2][N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fg = lane >> 5;
    const int64_t tiles = (n + 31) / 32, stride = (int64_t)gridDim.x * 8;
    int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    auto arow_of = [&](int64_t t) {
        int64_t row = (t < tiles ? t : tiles - 1) * 32 + fi;
        row = row < n ? row : n - 1;
        return a + row * lda + 8 * fg;
    };
    // one request: two k-tiles of the wave's 32 rows, 8 x 16 bytes per lane
    auto load_chunk = [&](const float *arow, int c, float4 (&r)[4 * kResChunk]) {
#pragma unroll
        for (int t = 0; t < kResChunk; ++t) {
            const float4 *p = reinterpret_cast<const float4 *>(arow + (c * kResChunk + t) * kBK);
            r[4 * t] = p[0]; r[4 * t + 1] = p[1]; r[4 * t + 2] = p[4]; r[4 * t + 3] = p[5];   // k = 8 g .. + 7 and 16 + 8 g .. + 7
        }
    };
    float4 ra0[4 * kResChunk], ra1[4 * kResChunk];
    // the first two requests go out before anything else: they land while the image is copied
    load_chunk(arow_of(tile), 0, ra0);
    load_chunk(arow_of(tile), 1, ra1);
    for (int i = tid; i < KT * kTileVec; i += kResThreads) Bs[i] = wimg[i];
    for (int i = tid; i < 4 * K; i += kResThreads) st[i] = in_stat[i];
    const float sa = pow2_scale_for(*a_max, 0);
    const float unscale = 1.0f / (sa * pow2_scale_for(*w_max, 0));
    const bool leaky = act == TGNN_ACT_LEAKY_RELU;
    float bcol[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bcol[tn] = bias[tn * 32 + fi];
    double cs[TN], cq[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) cs[tn] = cq[tn] = 0.0;
    __syncthreads();

    // the k-tiles c * 4 .. + 3 of a row tile out of one request's registers
    auto mma_chunk = [&](int c, const float4 (&r)[4 * kResChunk], f32x16 (&acc)[TN]) {
#pragma unroll
        for (int t = 0; t < kResChunk; ++t) {
            const int kt = c * kResChunk + t;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float x[8] = {r[4 * t + 2 * kb].x, r[4 * t + 2 * kb].y, r[4 * t + 2 * kb].z, r[4 * t + 2 * kb].w,
                              r[4 * t + 2 * kb + 1].x, r[4 * t + 2 * kb + 1].y, r[4 * t + 2 * kb + 1].z, r[4 * t + 2 * kb + 1].w};
                const int k = kt * kBK + kb * 16 + 8 * fg;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = bn_apply1(x[e], st[k + e], st[K + k + e], st[2 * K + k + e], st[3 * K + k + e]);
                f16x8 ah, al;
                split2_f16(x, sa, ah, al);
                // term-major: TN independent matrix instructions between two on the same accumulator, every fragment of the k-half
                // requested before the first one (per accumulator the order stays lo.hi, hi.lo, hi.hi)
                const u32x4_ *bt = Bs + (int64_t)kt * kTileVec + lane;
                f16x8 bh[TN], bl[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    bh[tn] = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128]);
                    bl[tn] = __builtin_bit_cast(f16x8, bt[(kb * TN + tn) * 128 + 64]);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[tn], acc[tn], 0, 0, 0);   // lo . hi
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[tn], acc[tn], 0, 0, 0);   // hi . lo
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[tn], acc[tn], 0, 0, 0);   // hi . hi
            }
        }
    };
    auto epilogue = [&](int64_t t, const f32x16 (&acc)[TN]) {
        const int64_t m0 = t * 32;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = tn * 32 + fi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * fg;
                if (orow < n) {
                    const float u = fmaf(acc[tn][r], unscale, bcol[tn]);
                    const float v = leaky ? (u >= 0.f ? u : u * kLeakySlope) : act_apply(u, act);
                    out[orow * ldo + col] = v;
                    cs[tn] += (double)v;
                    cq[tn] += (double)v * (double)v;
                }
            }
        }
    };
    static_assert(CPT % 2 == 0, "requests alternate between the two register sets");
    for (; tile < tiles; tile += stride) {
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
        const float *acur = arow_of(tile), *anext = arow_of(tile + stride);
#pragma unroll
        for (int c = 0; c < CPT; c += 2) {
            // request c + 2 (of this row tile or, behind its last two, of the wave's next one) into the registers request c leaves
            mma_chunk(c, ra0, acc);
            if (c + 2 < CPT) load_chunk(acur, c + 2, ra0);
            else load_chunk(anext, c + 2 - CPT, ra0);
            mma_chunk(c + 1, ra1, acc);
            if (c + 3 < CPT) load_chunk(acur, c + 3, ra1);
            else load_chunk(anext, c + 3 - CPT, ra1);
        }
        epilogue(tile, acc);
    }
    if (bn_partial) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const double s_ = cs[tn] + __shfl_xor(cs[tn], 32, 64);
            const double q_ = cq[tn] + __shfl_xor(cq[tn], 32, 64);
            if (fg == 0) {
                red[(wave * 2 + 0) * N + tn * 32 + fi] = s_;
                red[(wave * 2 + 1) * N + tn * 32 + fi] = q_;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * N; i += kResThreads) {
            const int which = i / N, cl = i % N;
            double tot = 0.0;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) tot += red[(wv * 2 + which) * N + cl];
            if (fin.counter) st_partial_sc1(bn_partial, (int64_t)blockIdx.x * 2 * N + i, tot);
            else bn_partial[(int64_t)blockIdx.x * 2 * N + i] = tot;
        }
        // [r6] the record by the producer itself (see dense_f16_rows_kernel); `red` is dead behind the helper's first barrier
        if (fin.counter) bn_fold_two_level<N>(fin, bn_partial, fold_rows, red, reinterpret_cast<unsigned *>(red + 2 * N));
    }
}

static size_t dense_f16_image_bytes(int in_dim, int out_dim) { return (size_t)in_dim * out_dim * 4; }
static std::atomic<int> g_dense_resident{1};                 // tgnn_set_dense_rows_mode bit 1 (0 = off): dense_f16_resident_kernel for the inner layers
static std::atomic<int> g_dense_halves{0};                   // tgnn_set_dense_rows_mode bits 2-3: blocks per CU of dense_f16_halves_kernel (0 = off)
static std::atomic<int> g_dense_rows_mode{0};                // tgnn_set_dense_rows_mode: 0 = dense_f16_rows_kernel, 1 = dense_f16_rows2_kernel

template <int TN>
static int launch_dense_f16_rows2(hipStream_t s, const float *a, int64_t lda, int64_t akb, int kps, const float *in_stat,
                                  const void *wimg, const float *b, int64_t n, int in_dim, int act, float *out, int64_t ldo,
                                  double *bn_partial, const unsigned *a_max, int n_a_max, const unsigned *w_max) {
    constexpr size_t fixed = sizeof(u32x4_) * 2 * (2 * TN * 2 * 64) + sizeof(double) * 8 * (32 * TN);
    const size_t lds = fixed + (in_stat ? (size_t)4 * in_dim * sizeof(float) : 0);
    int blocks = producer_blocks(n, 128);
    const int cap = device_cus();                             // one block per CU: each wave has a SIMD's registers to itself
    if (blocks > cap) blocks = cap;
    const u32x4_ *img = static_cast<const u32x4_ *>(wimg);
    if (in_stat) {
        static LdsOptIn site;
        (void)opt_in_dynamic_lds(dense_f16_rows2_kernel<TN, true>, 160 * 1024 - 256, site);
        dense_f16_rows2_kernel<TN, true><<<blocks, kRowsThreads, lds, s>>>(a, lda, akb, kps, in_stat, img, b, n, in_dim, act, out, ldo,
                                                                          bn_partial, a_max, n_a_max, w_max);
    } else {
        static LdsOptIn site;
        (void)opt_in_dynamic_lds(dense_f16_rows2_kernel<TN, false>, 160 * 1024 - 256, site);
        dense_f16_rows2_kernel<TN, false><<<blocks, kRowsThreads, lds, s>>>(a, lda, akb, kps, nullptr, img, b, n, in_dim, act, out, ldo,
                                                                           bn_partial, a_max, n_a_max, w_max);
    }
    return blocks;
}

// 64 -> 32 (the final MLP's fourth Linear) on the resident kernel, one k-tile per request; -1 = not applicable
static int launch_dense_f16_resident1(hipStream_t s, const float *a, int64_t lda, int64_t akb, int kps, const float *in_stat,
                                      const void *wimg, const float *b, int64_t n, int in_dim, int act, float *out, int64_t ldo,
                                      double *bn_partial, const unsigned *a_max, int n_a_max, const unsigned *w_max,
                                      const GinFin &fin = GinFin{}, double *fold_rows = nullptr) {
    if (!(g_dense_resident.load(std::memory_order_relaxed) && in_stat && n_a_max == 1 && kps == 1 && akb == kBK && in_dim == 64 && lda % 4 == 0))
        return -1;
    const size_t lds1 = (size_t)in_dim * 32 * 4 + (size_t)4 * in_dim * sizeof(float) + (size_t)8 * 2 * 32 * sizeof(double);
    int blocks1 = (int)((n + 255) / 256);
    if (blocks1 > device_cus()) blocks1 = device_cus();
    if (blocks1 < 1) blocks1 = 1;
    dense_f16_resident_kernel<1, 2, 1><<<blocks1, kResThreads, lds1, s>>>(a, lda, in_stat, static_cast<const u32x4_ *>(wimg), b, n, act, out,
                                                                        ldo, bn_partial, a_max, w_max, fin, fold_rows);
    return blocks1;
}

// fin.counter != NULL: the kernel writes the BatchNorm's record itself (bn_fold_two_level) and *folded is set -- the rows and the
// resident kernels; the experimental ones leave *folded alone (the caller's finalize launch follows)
template <int TN>
static int launch_dense_f16_rows(hipStream_t s, const float *a, int64_t lda, int64_t akb, int kps, const float *in_stat,
                                 const void *wimg, const float *b, int64_t n, int in_dim, int act, float *out, int64_t ldo,
                                 double *bn_partial, const unsigned *a_max, int n_a_max, const unsigned *w_max,
                                 const GinFin &fin_in = GinFin{}, double *fold_rows = nullptr, bool *folded = nullptr) {
    GinFin fin = (bn_partial && fold_rows && folded) ? fin_in : GinFin{};
    // [r6] BatchNorm-on-load layers whose whole image fits LDS: the barrier-free resident kernel
    if constexpr (TN == 4 || TN == 2)
    if (g_dense_resident.load(std::memory_order_relaxed) && in_stat && n_a_max == 1 && kps == 1 && akb == kBK &&
        (in_dim == 128 || in_dim == 256 || (in_dim == 64 && TN == 2)) && lda % 4 == 0) {
        constexpr int N = 32 * TN;
        const size_t lds = (size_t)in_dim * N * 4 + (size_t)4 * in_dim * sizeof(float) + (size_t)8 * 2 * N * sizeof(double);
        int blocks = (int)((n + 255) / 256);
        const int cap = device_cus();
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        const u32x4_ *img = static_cast<const u32x4_ *>(wimg);
        if constexpr (TN == 2)
        if (in_dim == 64) {                                   // (config 3's fourth Linear, 64 -> 64: one k-tile per request)
            dense_f16_resident_kernel<2, 2, 1><<<blocks, kResThreads, lds, s>>>(a, lda, in_stat, img, b, n, act, out, ldo, bn_partial, a_max, w_max, fin,
                                                                                fold_rows);
            if (fin.counter) *folded = true;
            return blocks;
        }
        if (in_dim == 256) {
            static LdsOptIn site;
            (void)opt_in_dynamic_lds(dense_f16_resident_kernel<TN, 4>, 160 * 1024 - 256, site);
            dense_f16_resident_kernel<TN, 4><<<blocks, kResThreads, lds, s>>>(a, lda, in_stat, img, b, n, act, out, ldo, bn_partial, a_max, w_max, fin,
                                                                              fold_rows);
        } else {
            static LdsOptIn site;
            (void)opt_in_dynamic_lds(dense_f16_resident_kernel<TN, 2>, 160 * 1024 - 256, site);
            dense_f16_resident_kernel<TN, 2><<<blocks, kResThreads, lds, s>>>(a, lda, in_stat, img, b, n, act, out, ldo, bn_partial, a_max, w_max, fin,
                                                                              fold_rows);
        }
        if (fin.counter) *folded = true;
        return blocks;
    }
    if constexpr (TN == 8)
    if (g_dense_halves.load(std::memory_order_relaxed) && !in_stat && kps == 1) {
        constexpr size_t lds = sizeof(u32x4_) * 2 * (2 * 4 * 2 * 64) + sizeof(double) * 8 * 128;
        const int64_t items = ((n + 127) / 128) * 2;
        int blocks = (int)(items < TGNN_BN_MAX_PARTIALS ? items : TGNN_BN_MAX_PARTIALS);
        const int cap = g_dense_halves.load(std::memory_order_relaxed) * device_cus();
        if (blocks > cap) blocks = cap;
        if (blocks > 1) blocks &= ~1;                         // (even: a block keeps its column half over its items)
        dense_f16_halves_kernel<8, 4><<<blocks, kRowsThreads, lds, s>>>(a, lda, akb, kps, nullptr, static_cast<const u32x4_ *>(wimg), b, n,
                                                                       in_dim, act, out, ldo, bn_partial, a_max, n_a_max, w_max);
        return blocks;
    }
    if (g_dense_rows_mode.load(std::memory_order_relaxed) == 1 && kps == 1)
        return launch_dense_f16_rows2<TN>(s, a, lda, akb, kps, in_stat, wimg, b, n, in_dim, act, out, ldo, bn_partial, a_max, n_a_max,
                                          w_max);
    const size_t lds = in_stat ? (size_t)4 * in_dim * sizeof(float) : 0;     // (dynamic part: the BatchNorm record)
    int blocks = producer_blocks(n, 128);
    const int cap = 2 * device_cus();
    if (blocks > cap) blocks = cap;
    const u32x4_ *img = static_cast<const u32x4_ *>(wimg);
    if (in_stat) {
        static LdsOptIn site;
        if (TN == 8) (void)opt_in_dynamic_lds(dense_f16_rows_kernel<TN, true>, (int)lds, site);
        dense_f16_rows_kernel<TN, true><<<blocks, kRowsThreads, lds, s>>>(a, lda, akb, kps, in_stat, img, b, n, in_dim, act, out, ldo,
                                                                         bn_partial, a_max, n_a_max, w_max, fin, fold_rows);
    } else {
        dense_f16_rows_kernel<TN, false><<<blocks, kRowsThreads, lds, s>>>(a, lda, akb, kps, nullptr, img, b, n, in_dim, act, out, ldo,
                                                                          bn_partial, a_max, n_a_max, w_max, fin, fold_rows);
    }
    if (fin.counter) *folded = true;
    return blocks;
}

template <int TM, int WN, bool F16 = false>
static void launch_dense_split(int blocks_x, hipStream_t s, const float *a, int64_t lda, int64_t akb, int kps,
                               const float *in_stat, const float *w, const float *b, int64_t n, int in_dim, int out_dim,
                               int act, float *out, int64_t ldo, double *bn_partial, const unsigned *a_max = nullptr,
                               int n_a_max = 0, const unsigned *w_max = nullptr) {
    constexpr int BM = (4 / WN) * TM * 32, BN = WN * 64;
    size_t lds = (size_t)(F16 ? 2 : 3) * (BM + BN) * kSplitLd * 2;
    const size_t red = (size_t)(4 / WN) * 2 * BN * sizeof(double);
    if (red > lds) lds = red;
    static LdsOptIn site;
    if (lds > 64 * 1024) (void)opt_in_dynamic_lds(dense_split_kernel<TM, WN, F16>, 160 * 1024 - 256, site);
    const int nby = (out_dim + BN - 1) / BN;
    dense_split_kernel<TM, WN, F16><<<blocks_x * nby, 256, lds, s>>>(
        a, lda, akb, kps, in_stat, w, b, n, in_dim, out_dim, act, out, ldo, bn_partial, a_max, n_a_max, w_max, blocks_x, nby);
}

template <int NT, bool FAST>
static void launch_dense(dim3 grid, hipStream_t s, const float *a, int64_t lda, int64_t akb, int kps, const float *in_stat,
                         const float *w, const float *b, int64_t n, int in_dim, int out_dim, int act, float *out,
                         int64_t ldo, double *bn_partial, int vec_a, int vec_w) {
    constexpr int BN = 32 * NT;
    size_t lds = (size_t)(kBM + BN) * kLD * sizeof(float);
    const size_t red = (size_t)4 * 2 * BN * sizeof(double);
    if (red > lds) lds = red;
    dense_mfma_kernel<NT, FAST><<<grid, 256, lds, s>>>(a, lda, akb, kps, in_stat, w, b, n, in_dim, out_dim, act, out, ldo,
                                                       bn_partial, vec_a, vec_w);
}

}  // namespace tgnn

using namespace tgnn;

// ------------------------------------------------------------------------------------------
// The two ends of the network are too narrow for matrix tiles: the first init layer (Fx = 3 .. 8 inputs -> 32) and the read-out
// (32 -> 1, sigmoid).  Through the 128-row MFMA kernels they cost 20 us each for 13 MB of traffic (one k-tile, mostly padding,
// a block-level prologue and epilogue per 128 rows); as plain element-wise kernels they run at memory speed.
// ------------------------------------------------------------------------------------------
// out[r] = act( BN(a[r][:]) . w + b ), out_dim == 1: L = in_dim / 4 lanes per row (a power of two <= 64), float4 each, shuffle fold
template <int L>
__global__ __launch_bounds__(256) void dense_out1_kernel(const float *__restrict__ a, int64_t lda, const float *__restrict__ in_stat,
                                                         const float *__restrict__ w, const float *__restrict__ bias, int64_t n,
                                                         int in_dim, int act, float *__restrict__ out, int64_t ldo) {
    const int q = threadIdx.x % L;
    const float4 wv = reinterpret_cast<const float4 *>(w)[q];
    float4 mh = make_float4(0, 0, 0, 0), ml = mh, gi = make_float4(1, 1, 1, 1), be = mh;
    if (in_stat) {
        mh = reinterpret_cast<const float4 *>(in_stat)[q];
        ml = reinterpret_cast<const float4 *>(in_stat + in_dim)[q];
        gi = reinterpret_cast<const float4 *>(in_stat + 2 * in_dim)[q];
        be = reinterpret_cast<const float4 *>(in_stat + 3 * in_dim)[q];
    }
    const float b0 = bias[0];
    constexpr int kRows = 256 / L;
    for (int64_t r = (int64_t)blockIdx.x * kRows + threadIdx.x / L; r < n; r += (int64_t)gridDim.x * kRows) {
        const float4 x = *reinterpret_cast<const float4 *>(a + r * lda + 4 * q);
        float s = bn_apply1(x.x, mh.x, ml.x, gi.x, be.x) * wv.x;
        s = fmaf(bn_apply1(x.y, mh.y, ml.y, gi.y, be.y), wv.y, s);
        s = fmaf(bn_apply1(x.z, mh.z, ml.z, gi.z, be.z), wv.z, s);
        s = fmaf(bn_apply1(x.w, mh.w, ml.w, gi.w, be.w), wv.w, s);
#pragma unroll
        for (int d = L / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (q == 0) out[r * ldo] = act_apply(s + b0, act);
    }
}

// out[r][0..32) = act( a[r][0..in_dim) . W^T + b ), in_dim <= 8, no input BatchNorm; 8 lanes per row, 4 outputs each; fp64 column
// sums per block (one partial row per block, fixed order: rows of a thread in order, then the threads of a column group in order)
__global__ __launch_bounds__(256) void dense_in8_kernel(const float *__restrict__ a, int64_t lda, const float *__restrict__ w,
                                                        const float *__restrict__ bias, int64_t n, int in_dim, int act,
                                                        float *__restrict__ out, int64_t ldo, double *__restrict__ bn_partial) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x, q = tid & 7;
    float wr[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[o][k] = k < in_dim ? w[(4 * q + o) * in_dim + k] : 0.f;
    const float4 bv = reinterpret_cast<const float4 *>(bias)[q];
    double cs[4] = {0, 0, 0, 0}, cq[4] = {0, 0, 0, 0};
    for (int64_t r = (int64_t)blockIdx.x * 32 + (tid >> 3); r < n; r += (int64_t)gridDim.x * 32) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = k < in_dim ? a[r * lda + k] : 0.f;
        float o4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o4[o] = fmaf(x[k], wr[o][k], o4[o]);
            o4[o] = act_apply(o4[o], act);
            cs[o] += (double)o4[o];
            cq[o] += (double)o4[o] * (double)o4[o];
        }
        *reinterpret_cast<float4 *>(out + r * ldo + 4 * q) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
    if (bn_partial) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            red[tid * 8 + o] = cs[o];
            red[tid * 8 + 4 + o] = cq[o];
        }
        __syncthreads();
        if (tid < 64) {                                       // entry = which * 32 + column; column = 4 q + o
            const int which = tid >> 5, colm = tid & 31, qq = colm >> 2, o = colm & 3;
            double t = 0.0;
            for (int g = 0; g < 32; ++g) t += red[(g * 8 + qq) * 8 + which * 4 + o];
            bn_partial[(int64_t)blockIdx.x * 64 + tid] = t;
        }
    }
}

static int dense_act_impl(const float *a, int64_t lda, int64_t a_kblock_stride, int kps, const float *in_stat,
                          const float *w, const float *b, int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act,
                          float *out, int64_t ldo, double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream,
                          const unsigned *a_max = nullptr, int n_a_max = 0, const unsigned *w_max = nullptr,
                          const void *wimg = nullptr, const GinFin *fold = nullptr, double *fold_rows = nullptr, bool *folded = nullptr) {
    const GinFin fin = (fold && fold_rows && folded && fold->counter) ? *fold : GinFin{};
    if (folded) *folded = false;
    TGNN_CHECK_ARG(n_rows >= 0 && in_dim >= 1 && out_dim >= 1, "shape");
    TGNN_CHECK_ARG(act >= TGNN_ACT_NONE && act <= TGNN_ACT_SIGMOID, "activation");
    if (n_rows == 0) {
        if (n_partials_host) *n_partials_host = 0;
        return TGNN_OK;
    }
    TGNN_CHECK_ARG(a && w && b && out, "null pointer");
    TGNN_CHECK_ARG(ldo >= out_dim, "ldo");
    TGNN_CHECK_ARG(in_dim <= kBK || lda >= kBK || a_kblock_stride == kBK, "A layout");
    TGNN_CHECK_ARG(kps >= 1, "K blocks per slot");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int vec_a = (lda % 4 == 0) && (a_kblock_stride % 4 == 0) && ((uintptr_t)a % 16 == 0);
    // ---- the network's two narrow ends as element-wise kernels (see dense_out1_kernel / dense_in8_kernel)
    if (out_dim == 1 && !bn_partial && kps == 1 && a_kblock_stride == kBK && vec_a && in_dim >= 4 && in_dim <= 256 &&
        (in_dim & (in_dim - 1)) == 0 && ((uintptr_t)w % 16 == 0) && (!in_stat || (uintptr_t)in_stat % 16 == 0)) {
        int64_t nb = (n_rows * (in_dim / 4) + 255) / 256;
        if (nb > 2048) nb = 2048;
        const unsigned g = (unsigned)nb;
        switch (in_dim / 4) {
            case 1: dense_out1_kernel<1><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
            case 2: dense_out1_kernel<2><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
            case 4: dense_out1_kernel<4><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
            case 8: dense_out1_kernel<8><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
            case 16: dense_out1_kernel<16><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
            case 32: dense_out1_kernel<32><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
            default: dense_out1_kernel<64><<<g, 256, 0, s>>>(a, lda, in_stat, w, b, n_rows, in_dim, act, out, ldo); break;
        }
        if (n_partials_host) *n_partials_host = 0;
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }
    if (in_dim <= 8 && out_dim == 32 && !in_stat && kps == 1 && ldo % 4 == 0 && ((uintptr_t)out % 16 == 0) && ((uintptr_t)b % 16 == 0)) {
        const int nb = producer_blocks(n_rows, 32 * 8);       // 32 rows per sweep, ~8 sweeps per block; <= TGNN_BN_MAX_PARTIALS
        dense_in8_kernel<<<nb, 256, 0, s>>>(a, lda, w, b, n_rows, in_dim, act, out, ldo, bn_partial);
        if (n_partials_host) *n_partials_host = nb;
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }
    const int vec_w = (in_dim % 4 == 0) && ((uintptr_t)w % 16 == 0);
    // one block = one BatchNorm partial row, hence the cap of producer_blocks; a product nobody takes statistics of (the
    // backward's dx = dz . W, the GIN hidden layers it re-derives) gets one block per row tile instead of walking two
    // tiles per block behind a full chip: 23 -> ~13 us for [100k, 32] x [32, 32]
    auto row_blocks = [&](int rows_per_block) {
        if (bn_partial) return producer_blocks(n_rows, rows_per_block);
        int64_t nb = (n_rows + rows_per_block - 1) / rows_per_block;
        if (nb > 8192) nb = 8192;
        return (int)(nb < 1 ? 1 : nb);
    };
    const int blocks_x = row_blocks(kBM);
    const bool fast = vec_a && vec_w && in_dim % kBK == 0;
    constexpr int exact_only = 0;     // 1: exact-fp32 MFMA kernels everywhere (the split-precision ones measured equal to 4e-7)
    // [r6] 64 -> 32 with bounds and an operand image at hand (the final MLP's fourth Linear at benchmark sizes): the resident kernel's
    // smallest form on fp16 pairs instead of the exact-fp32 block-tile kernel (17.6 us for 38 MB at 100 000 rows)
    if (fast && !exact_only && out_dim == 32 && in_dim == 64 && a_max && w_max && n_a_max == 1 && wimg && in_stat &&
        n_rows >= kDenseRowsKernelMin && ((uintptr_t)wimg % 16) == 0) {
        const int nb = launch_dense_f16_resident1(s, a, lda, a_kblock_stride, kps, in_stat, wimg, b, n_rows, in_dim, act, out, ldo, bn_partial,
                                                  a_max, n_a_max, w_max, bn_partial ? fin : GinFin{}, fold_rows);
        if (nb > 0) {
            if (fin.counter && bn_partial) *folded = true;
            if (n_partials_host) *n_partials_host = nb;
            TGNN_CHECK_LAUNCH();
            return TGNN_OK;
        }
    }
    if (fast && out_dim >= 64 && !exact_only && lda % 8 == 0 && a_kblock_stride % 8 == 0 && in_dim % 8 == 0) {
        // bf16 x 3 split-precision path (see dense_split_kernel)
        // few row tiles (small layouts): the 128 x 64 block tile puts twice as many blocks on the chip and halves the
        // matrix work per k-step of each -- the kernel is then bound by the latency of its serial k loop
        // (round 2: the weights pre-split into bf16 planes once per forward -- three 16-byte copies per item instead of a split per
        //  block and k-tile -- changed nothing measurable, 2.17-2.24 vs 2.20-2.22 ms per forward; a 128 x 256 tile -- A split once per row tile instead of once per column tile, but 92 KB of LDS = one block
        //  per CU -- runs 672 -> 256 in 449 us against 281 us: the kernel lives on a second block covering the first one's
        //  two barriers per k-tile)
        constexpr int small_rows = 16384;
        const bool f16 = a_max && w_max && n_a_max >= 1;     // (a_max bounds the input AFTER the BatchNorm applied while staging)
        // the operand image of W is at hand (dense_f16_image_build) and every output column fits one wave's accumulators:
        // rows-per-wave kernel (dense_f16_rows_kernel)
        // (measured, 672 -> 256: 54 / 59 / 158 / 393 us at 10 000 / 32 000 / 100 000 / 300 000 rows against 34 / 56 / 177 / 446 for the
        //  block-tile kernels; inside the forward at 100 000 rows 166 + 58 + 28 us for the three layers against 197 + 72 + 36)
        if (f16 && wimg && n_rows >= kDenseRowsKernelMin && (out_dim == 64 || out_dim == 128 || out_dim == 256) &&
            ((uintptr_t)wimg % 16) == 0 && (!in_stat || in_dim <= 1024)) {
            int nb;
            if (out_dim == 256)
                nb = launch_dense_f16_rows<8>(s, a, lda, a_kblock_stride, kps, in_stat, wimg, b, n_rows, in_dim, act, out, ldo, bn_partial,
                                              a_max, n_a_max, w_max, fin, fold_rows, folded);
            else if (out_dim == 128)
                nb = launch_dense_f16_rows<4>(s, a, lda, a_kblock_stride, kps, in_stat, wimg, b, n_rows, in_dim, act, out, ldo, bn_partial,
                                              a_max, n_a_max, w_max, fin, fold_rows, folded);
            else
                nb = launch_dense_f16_rows<2>(s, a, lda, a_kblock_stride, kps, in_stat, wimg, b, n_rows, in_dim, act, out, ldo, bn_partial,
                                              a_max, n_a_max, w_max, fin, fold_rows, folded);
            if (n_partials_host) *n_partials_host = nb;
            TGNN_CHECK_LAUNCH();
            return TGNN_OK;
        }
        if (out_dim > 64 && n_rows > small_rows) {
            const int bx = row_blocks(128);
            if (f16)
                launch_dense_split<2, 2, true>(bx, s, a, lda, a_kblock_stride, kps, in_stat, w, b, n_rows, in_dim, out_dim, act,
                                               out, ldo, bn_partial, a_max, n_a_max, w_max);
            else
                launch_dense_split<2, 2>(bx, s, a, lda, a_kblock_stride, kps, in_stat, w, b, n_rows, in_dim, out_dim, act, out, ldo,
                                         bn_partial);
            if (n_partials_host) *n_partials_host = bx;
        } else if (f16) {
            const int bx = row_blocks(128);
            launch_dense_split<1, 1, true>(bx, s, a, lda, a_kblock_stride, kps, in_stat, w, b, n_rows, in_dim, out_dim, act, out,
                                           ldo, bn_partial, a_max, n_a_max, w_max);
            if (n_partials_host) *n_partials_host = bx;
        } else {
            const int bx = row_blocks(128);
            launch_dense_split<1, 1>(bx, s, a, lda, a_kblock_stride, kps, in_stat, w, b, n_rows, in_dim, out_dim, act, out, ldo,
                                     bn_partial);
            if (n_partials_host) *n_partials_host = bx;
        }
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }
#define TGNN_DENSE(NT_)                                                                                            \
    do {                                                                                                           \
        const dim3 grid_(blocks_x, (out_dim + 32 * NT_ - 1) / (32 * NT_));                                         \
        if (fast)                                                                                                  \
            launch_dense<NT_, true>(grid_, s, a, lda, a_kblock_stride, kps, in_stat, w, b, n_rows, in_dim, out_dim, act, \
                                    out, ldo, bn_partial, vec_a, vec_w);                                           \
        else                                                                                                       \
            launch_dense<NT_, false>(grid_, s, a, lda, a_kblock_stride, kps, in_stat, w, b, n_rows, in_dim, out_dim, act, \
                                     out, ldo, bn_partial, vec_a, vec_w);                                          \
    } while (0)
    constexpr int force_nt = 0;
    if (force_nt == 4 && out_dim > 32) {
        TGNN_DENSE(4);
    } else if (force_nt == 2 && out_dim > 32) {
        TGNN_DENSE(2);
    } else if (out_dim > 128) {
        TGNN_DENSE(8);
    } else if (out_dim > 64) {
        TGNN_DENSE(4);
    } else if (out_dim > 32) {
        TGNN_DENSE(2);
    } else {
        TGNN_DENSE(1);
    }
#undef TGNN_DENSE
    if (n_partials_host) *n_partials_host = blocks_x;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int32_t tgnn_set_dense_rows_mode(int32_t mode) {
    const int prev = g_dense_rows_mode.load() | (g_dense_resident.load() ? 2 : 0) | (g_dense_halves.load() << 2);
    if (mode < 0 || mode > 15) return prev;
    g_dense_rows_mode.store(mode & 1);
    g_dense_resident.store((mode >> 1) & 1);
    g_dense_halves.store((mode >> 2) & 3);
    return prev;
}

extern "C" int tgnn_dense_act_fwd(const float *a, int64_t lda, int64_t a_kblock_stride, const float *in_stat,
                                  const float *w, const float *b, int64_t n_rows, int32_t in_dim, int32_t out_dim,
                                  int32_t act, float *out, int64_t ldo, double *bn_partial,
                                  int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    return dense_act_impl(a, lda, a_kblock_stride, 1, in_stat, w, b, n_rows, in_dim, out_dim, act, out, ldo, bn_partial,
                          n_partials_host, stream);
}

extern "C" int tgnn_dense_act_slots_fwd(const float *a, int32_t slot_width, int64_t slot_stride, const float *in_stat,
                                        const float *w, const float *b, int64_t n_rows, int32_t in_dim, int32_t out_dim,
                                        int32_t act, float *out, int64_t ldo, double *bn_partial,
                                        int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(slot_width >= 32 && slot_width % 32 == 0 && in_dim % slot_width == 0,
                   "slot-major input: slot width must be a multiple of 32 that divides in_dim");
    return dense_act_impl(a, slot_width, slot_stride, slot_width / 32, in_stat, w, b, n_rows, in_dim, out_dim, act, out, ldo,
                          bn_partial, n_partials_host, stream);
}

namespace tgnn {
int dense_act_bounded(const float *a, int64_t lda, int64_t a_kblock_stride, const float *in_stat, const float *w, const float *b,
                      int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act, float *out, int64_t ldo, double *bn_partial,
                      int32_t *n_partials_host, const unsigned *a_max, int n_a_max, const unsigned *w_max, hipStream_t s,
                      const void *wimg, const GinFin *fold, double *fold_group_rows, bool *folded) {
    return dense_act_impl(a, lda, a_kblock_stride, 1, in_stat, w, b, n_rows, in_dim, out_dim, act, out, ldo, bn_partial,
                          n_partials_host, s, a_max, n_a_max, w_max, wimg, fold, fold_group_rows, folded);
}
size_t dense_f16_image_size(int in_dim, int out_dim) { return align_up(dense_f16_image_bytes(in_dim, out_dim), 256); }
// W [out_dim][in_dim] -> the fp16-pair operand image dense_f16_rows_kernel reads (scaled by w_max's power of two)
int dense_f16_image_build(const float *w, int in_dim, int out_dim, const unsigned *w_max, void *wimg, hipStream_t s) {
    if (in_dim % kBK || out_dim % 32 || ((uintptr_t)w % 16) || ((uintptr_t)wimg % 16)) return TGNN_ERR_UNSUPPORTED;
    const int items = (in_dim / kBK) * 2 * (out_dim / 32) * 64;
    dense_f16_image_kernel<<<(items + 255) / 256, 256, 0, s>>>(w, in_dim, out_dim / 32, w_max, static_cast<u32x4_ *>(wimg));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
// [r6] several images in one launch: block ranges per job
struct DenseImageJobs {
    const float *w[4];
    u32x4_ *img[4];
    const unsigned *w_max[4];
    int in_dim[4], tn_count[4], block0[5];
};
__global__ __launch_bounds__(256) void dense_f16_images_kernel(DenseImageJobs J, int n_jobs) {
    int k = 0;
    while (k + 1 < n_jobs && (int)blockIdx.x >= J.block0[k + 1]) ++k;
    const int in_dim = J.in_dim[k], tn_count = J.tn_count[k], ktiles = in_dim / kBK;
    const int item = ((int)blockIdx.x - J.block0[k]) * 256 + threadIdx.x;
    if (item >= ktiles * 2 * tn_count * 64) return;
    const int lane = item & 63, t = item >> 6, tn = t % tn_count, kb = (t / tn_count) & 1, kt = t / (2 * tn_count);
    const int col = tn * 32 + (lane & 31), k0 = kt * kBK + kb * 16 + 8 * (lane >> 5);
    const float sw = pow2_scale_for(*J.w_max[k], 0);
    const float4 *p = reinterpret_cast<const float4 *>(J.w[k] + (int64_t)col * in_dim + k0);
    const float4 x0 = p[0], x1 = p[1];
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    f16x8 hi, lo;
    split2_f16(x, sw, hi, lo);
    const int64_t o = ((int64_t)(kt * 2 + kb) * tn_count + tn) * 128 + lane;
    J.img[k][o] = __builtin_bit_cast(u32x4_, hi);
    J.img[k][o + 64] = __builtin_bit_cast(u32x4_, lo);
}
int dense_f16_images_build(int n_jobs, const float *const *w, const int *in_dim, const int *out_dim, const unsigned *const *w_max,
                           void *const *wimg, hipStream_t s) {
    if (n_jobs < 1 || n_jobs > 4) return TGNN_ERR_UNSUPPORTED;
    DenseImageJobs J{};
    int blocks = 0;
    for (int k = 0; k < n_jobs; ++k) {
        if (in_dim[k] % kBK || out_dim[k] % 32 || ((uintptr_t)w[k] % 16) || ((uintptr_t)wimg[k] % 16)) return TGNN_ERR_UNSUPPORTED;
        J.w[k] = w[k];
        J.img[k] = static_cast<u32x4_ *>(wimg[k]);
        J.w_max[k] = w_max[k];
        J.in_dim[k] = in_dim[k];
        J.tn_count[k] = out_dim[k] / 32;
        J.block0[k] = blocks;
        blocks += ((in_dim[k] / kBK) * 2 * (out_dim[k] / 32) * 64 + 255) / 256;
    }
    J.block0[n_jobs] = blocks;
    dense_f16_images_kernel<<<blocks, 256, 0, s>>>(J, n_jobs);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
// tgnn_dense_act_slots_fwd with the operands' bounds (forward.hip: the first Linear of the final MLP over the skip buffer)
int dense_act_slots_bounded(const float *a, int32_t slot_width, int64_t slot_stride, const float *w, const float *b,
                            int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act, float *out, int64_t ldo,
                            double *bn_partial, int32_t *n_partials_host, const unsigned *a_max, int n_a_max,
                            const unsigned *w_max, hipStream_t s, const void *wimg, const GinFin *fold, double *fold_group_rows,
                            bool *folded) {
    return dense_act_impl(a, slot_width, slot_stride, slot_width / 32, nullptr, w, b, n_rows, in_dim, out_dim, act, out, ldo,
                          bn_partial, n_partials_host, s, a_max, n_a_max, w_max, wimg, fold, fold_group_rows, folded);
}
}  // namespace tgnn

extern "C" int tgnn_dense_act_slots_f16_fwd(const float *a, int32_t slot_width, int64_t slot_stride, const float *w, const float *b,
                                            int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act, float *out, int64_t ldo,
                                            uint32_t *bounds_scratch, void *wimg_scratch, double *bn_partial,
                                            int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(slot_width == 32 && in_dim % 32 == 0 && in_dim >= 32 && out_dim >= 64 && n_rows >= 1, "shape");
    TGNN_CHECK_ARG(a && w && b && out && bounds_scratch, "null pointer");
    TGNN_CHECK_ARG(slot_stride % 8 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0, "alignment");
    const int n_slots = in_dim / 32;
    hipStream_t s = static_cast<hipStream_t>(stream);
    TGNN_CHECK_HIP(hipMemsetAsync(bounds_scratch, 0, (size_t)(n_slots + 1) * sizeof(uint32_t), s));
    for (int k = 0; k < n_slots; ++k) launch_absmax(a + (int64_t)k * slot_stride, n_rows * 32, bounds_scratch + k, s);
    launch_absmax(w, (int64_t)in_dim * out_dim, bounds_scratch + n_slots, s);
    if (wimg_scratch) {
        TGNN_CHECK_ARG(out_dim == 64 || out_dim == 128 || out_dim == 256, "the rows kernel takes out_dim 64 / 128 / 256");
        const int rc = dense_f16_image_build(w, in_dim, out_dim, bounds_scratch + n_slots, wimg_scratch, s);
        if (rc != TGNN_OK) return rc;
    }
    return dense_act_slots_bounded(a, slot_width, slot_stride, w, b, n_rows, in_dim, out_dim, act, out, ldo, bn_partial,
                                   n_partials_host, bounds_scratch, n_slots, bounds_scratch + n_slots, s, wimg_scratch);
}
