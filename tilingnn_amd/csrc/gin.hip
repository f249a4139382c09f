// GINConv for TilinGNN's collision branch on gfx950 (K6-K7 of SURVEY.md section 2b).
//
// Reference semantics (CollConv.forward, /root/reference/graph_networks/layers/coll_conv.py:24-27,
// calling PyG 1.3.2 GINConv with nn = MLP(C->32->64->C, Sigmoid x3, no BN), eps = 0 buffer):
//   z[v]   = (1 + eps) * x[v] + sum_{e: dst_e = v, src_e != v} x[src_e]
//   out[v] = LeakyReLU( sigmoid(L3 sigmoid(L2 sigmoid(L1 z[v]))) )
// In the network x = BatchNorm(a) of the previous CollConv (coll_conv.py:28-29).  BatchNorm is
// affine per column, so it commutes with the neighbourhood sum:
//   z[v] = ginv * [ (1+eps) (a[v]-mu) + sum (a[src]-mu) ] + (1 + eps + deg_v) * beta
// and the kernel reads the PRE-BN activations `a` of layer i-1 directly (in_stat != NULL); the
// normalised collision features are never written to HBM.
//
// Two launches per layer:
//   gin32_aggregate_kernel (HBM-bound gather): 8 lanes x float4 per destination row, CSR by
//            destination, sums in original edge order, XCD-contiguous row ranges; z -> HBM scratch.
//   gin32_mlp_kernel (MFMA-bound): 32 -> 32 -> 64 -> 32 with sigmoids on v_mfma_f32_32x32x2_f32,
//            hidden activations wave-private in LDS, LeakyReLU + fp64 BN column sums in the epilogue.
// (A first version kept the MLP per-thread with wave-uniform weights through the scalar cache:
//  297 us per layer at N = 100k -- every s_load batch paid an L2 round trip.  A later variant kept the
//  activations in the MFMA accumulator registers via the transposed product H^T = W . Z^T, no LDS
//  round trips at all: correct, but 29 us vs 23.7 us for this one -- 64 VGPRs of per-lane fp64 BN sums
//  cap occupancy, the 16-B-strided row loads/stores are TA-expensive and the final 32-lane fp64
//  butterfly is serial.  See DESIGN.md section 5.)
#include <stdlib.h>

#include "forward_persist.h"

namespace tgnn {


// ------------------------------------------------------------------------------------------
// K6: neighbourhood sum (HBM-bound gather).  8 lanes x float4 per destination row, rows of one
// XCD contiguous so that its private L2 holds the activations its gathers touch; no LDS, few
// registers -> full occupancy hides the rowptr -> col_src -> a[src] dependent-load chain.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gin32_aggregate_kernel(
    const float *__restrict__ a, int64_t lda, const float *__restrict__ in_stat, const int *__restrict__ rowptr,
    const int *__restrict__ col_src, const float *__restrict__ eps_p, int64_t n, float *__restrict__ z) {
    constexpr int C = 32;
#ifdef TGNN_ABL_EMPTYGIN
    if (n > 0) return;                                     // (timing ablation)
#endif
    const int tid = threadIdx.x, g = tid >> 3, q = tid & 7;
    // block -> (xcd, chunk): XCD x owns rows [n*x/8, n*(x+1)/8), 32 rows per block
    const int xcd = blockIdx.x & 7, chunk = blockIdx.x >> 3;
    const int64_t r_beg = n * xcd / 8, r_end = n * (xcd + 1) / 8;
    const int64_t v = r_beg + (int64_t)chunk * 32 + g;
    if (v >= r_end) return;
    const float one_eps = 1.0f + eps_p[0];
    float4 mhi = make_float4(0, 0, 0, 0), mlo = mhi, gv = make_float4(1, 1, 1, 1), bv = mhi;
    if (in_stat) {
        mhi = reinterpret_cast<const float4 *>(in_stat)[q];
        mlo = reinterpret_cast<const float4 *>(in_stat + C)[q];
        gv = reinterpret_cast<const float4 *>(in_stat + 2 * C)[q];
        bv = reinterpret_cast<const float4 *>(in_stat + 3 * C)[q];
    }
    const int beg = rowptr[v], end = rowptr[v + 1];
    const float4 self = *reinterpret_cast<const float4 *>(a + v * lda + 4 * q);
    float4 acc = make_float4(0, 0, 0, 0);
    int e = beg;
    for (; e + 4 <= end; e += 4) {  // 4 independent gathers in flight, summed in edge order
        const int s0 = col_src[e], s1 = col_src[e + 1], s2 = col_src[e + 2], s3 = col_src[e + 3];
        const float4 x0 = *reinterpret_cast<const float4 *>(a + (int64_t)s0 * lda + 4 * q);
        const float4 x1 = *reinterpret_cast<const float4 *>(a + (int64_t)s1 * lda + 4 * q);
        const float4 x2 = *reinterpret_cast<const float4 *>(a + (int64_t)s2 * lda + 4 * q);
        const float4 x3 = *reinterpret_cast<const float4 *>(a + (int64_t)s3 * lda + 4 * q);
#define TGNN_ACC(X)                          \
    acc.x += ((X).x - mhi.x) - mlo.x;        \
    acc.y += ((X).y - mhi.y) - mlo.y;        \
    acc.z += ((X).z - mhi.z) - mlo.z;        \
    acc.w += ((X).w - mhi.w) - mlo.w;
        TGNN_ACC(x0) TGNN_ACC(x1) TGNN_ACC(x2) TGNN_ACC(x3)
    }
    for (; e < end; ++e) {
        const float4 x0 = *reinterpret_cast<const float4 *>(a + (int64_t)col_src[e] * lda + 4 * q);
        TGNN_ACC(x0)
    }
#undef TGNN_ACC
    const float kb = one_eps + (float)(end - beg);
    float4 o;
    o.x = fmaf(gv.x, fmaf(one_eps, (self.x - mhi.x) - mlo.x, acc.x), kb * bv.x);
    o.y = fmaf(gv.y, fmaf(one_eps, (self.y - mhi.y) - mlo.y, acc.y), kb * bv.y);
    o.z = fmaf(gv.z, fmaf(one_eps, (self.z - mhi.z) - mlo.z, acc.z), kb * bv.z);
    o.w = fmaf(gv.w, fmaf(one_eps, (self.w - mhi.w) - mlo.w, acc.w), kb * bv.w);
    *reinterpret_cast<float4 *>(z + v * C + 4 * q) = o;
}

// ------------------------------------------------------------------------------------------
// K7: the GIN MLP (32 -> 32 -> 64 -> 32, sigmoid after every Linear) on matrix cores, activations resident in
// registers from the first layer to the last, fp32-class accuracy on the bf16 matrix pipe.
//
// (1) The TRANSPOSED product H^T = W . Z^T is computed (A operand = weights, B operand = activations): the
//     accumulators of v_mfma_f32_16x16x32_bf16 then hold, in lane (n = lane & 15, q = lane >> 4), features
//     16 mb + 4 q + r (r = 0..3) of batch row n for M block mb -- two M blocks are exactly the 8 values per lane
//     the B operand of the next layer takes, if the K order of that layer is DEFINED as
//         kf(q, e) = 4 q + e  (e < 4),  16 + 4 q + (e - 4)  (e >= 4)        [+ 32 for the second K step]
//     The sum over K does not care about the order; the weights are stored in LDS in that order.  No LDS round
//     trip and no barrier between the layers.
// (2) Every fp32 operand is split exactly into three bf16 pieces (hi + mid + lo) and the product is accumulated
//     in fp32 from the six leading cross terms (see dense.hip): 60 MFMAs of ~18 cycles per 16-row tile instead
//     of 40 fp32 MFMAs of ~72.  On this chip MFMA and VALU time ADD (scratch/ubench/coissue.hip), so matrix
//     cycles saved are wall-clock saved.
// History: per-thread MLP through the scalar cache 297 us/layer; fp32 MFMA with H1/H2 through LDS 23.7 us;
// the same chain register-resident in fp32 24.6 us (not the LDS traffic but matrix-pipe time and the 4-vs-3.05
// tiles-per-SIMD quantisation of 32-row tiles were the cost); this one: see profiles/.
// ------------------------------------------------------------------------------------------
constexpr int kMlpWaves = 8, kMlpThreads = kMlpWaves * 64;
constexpr int kGinGroupRow0 = TGNN_BN_MAX_PARTIALS - 16;      // [r6] the 16 group rows of gin32_mlp_kernel's two-level BatchNorm fold
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ int gin_kf(int q, int e) { return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4); }

__device__ __forceinline__ void gin_split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
    split3_trunc(x, hi, mid, lo);           // exact three-way split by truncation, tgnn_common.h
}

// acc += W . X over one K step of 32: six cross terms, smallest first
__device__ __forceinline__ f32x4 gin_mma6(const bf16x8 *wpl, int plane_stride, const bf16x8 (&x)[3], f32x4 acc) {
    const bf16x8 w0 = wpl[0], w1 = wpl[plane_stride], w2 = wpl[2 * plane_stride];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x[0], acc, 0, 0, 0);   // lo . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[2], acc, 0, 0, 0);   // hi . lo
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[1], acc, 0, 0, 0);   // mid . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[0], acc, 0, 0, 0);   // mid . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[1], acc, 0, 0, 0);   // hi . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[0], acc, 0, 0, 0);   // hi . hi
    return acc;
}

// (2 waves per SIMD on purpose: the compiler then keeps all 30 weight fragments in registers, 232 VGPRs; with
//  16 waves per block and 128 VGPRs the same code spills: 59.6 us)
__global__ __launch_bounds__(kMlpThreads, 2) void gin32_mlp_kernel(
    const float *__restrict__ z, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial, GinFin fin) {
    // weight images: [plane 3][M block][i 16][q 4] x bf16x8 -- the A fragment of lane (i, q) is one ds_read_b128
    __shared__ bf16x8 W1s[3 * 2 * 64];          // K = 32 (natural order 8 q + e: Z comes straight from memory)
    __shared__ bf16x8 W2s[3 * 4 * 64];          // K = 32 in kf order
    __shared__ bf16x8 W3s[3 * 2 * 2 * 64];      // [plane][M block][K step][..], K = 64 in kf order
    __shared__ __attribute__((aligned(16))) float Bs[128];   // b1 | b2 | b3
    __shared__ double red[kMlpWaves * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fn = lane & 15, fq = lane >> 4;
#ifdef TGNN_ABL_EMPTYGIN
    if (n > 0) return;
#endif

    // ---- tile share of this wave (16-row tiles): balanced per SIMD (waves w, w + 4 of a block run on SIMD w & 3)
    const int64_t n_tiles = (n + 15) / 16;
    const int nblk = gridDim.x;
    int64_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (int64_t)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int64_t slot = blk * 4 + (wave & 3), n_slots = (int64_t)nblk * 4;
    const int64_t q0 = n_tiles * slot / n_slots, q1 = n_tiles * (slot + 1) / n_slots;
    constexpr int kSubs = kMlpWaves / 4;
    const int sub = wave >> 2;
    const int64_t t0 = q0 + (q1 - q0) * sub / kSubs, t1 = q0 + (q1 - q0) * (sub + 1) / kSubs;

    // Z rows of a tile: lane (n, q) reads floats 8 q .. 8 q + 7 of row n; one tile ahead
    auto load_z = [&](int64_t tile, float4 (&zin)[2]) {
        int64_t zr = tile * 16 + fn;                        // clamped, unconditional: rows >= n are masked at the store
        zr = zr < n ? zr : n - 1;
        const float4 *pz = reinterpret_cast<const float4 *>(z + zr * 32 + 8 * fq);
        zin[0] = pz[0];
        zin[1] = pz[1];
    };
    float4 zin[2];
    load_z(t0 < t1 ? t0 : 0, zin);                          // in flight while the weight images are built

#ifndef TGNN_ABL_GINNOPRO
    for (int i = tid; i < 2 * 64; i += kMlpThreads) {       // item = (M block, i, q): 8 weights
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w1[(16 * mb + ii) * 32 + 8 * q + e];
        gin_split3(x, W1s[(0 * 2 + mb) * 64 + ii * 4 + q], W1s[(1 * 2 + mb) * 64 + ii * 4 + q], W1s[(2 * 2 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += kMlpThreads) {
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w2[(16 * mb + ii) * 32 + gin_kf(q, e)];
        gin_split3(x, W2s[(0 * 4 + mb) * 64 + ii * 4 + q], W2s[(1 * 4 + mb) * 64 + ii * 4 + q], W2s[(2 * 4 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += kMlpThreads) {       // item = (M block, K step, i, q)
        const int mb = i >> 7, ks = (i >> 6) & 1, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w3[(16 * mb + ii) * 64 + 32 * ks + gin_kf(q, e)];
        const int o = (mb * 2 + ks) * 64 + ii * 4 + q;
        gin_split3(x, W3s[0 * 256 + o], W3s[1 * 256 + o], W3s[2 * 256 + o]);
    }
#endif
    if (tid < 32) Bs[tid] = b1[tid];
    else if (tid < 96) Bs[tid] = b2[tid - 32];
    else if (tid < 128) Bs[tid] = b3[tid - 96];
    __syncthreads();

    // this lane's accumulator registers hold features 16 mb + 4 q + r: bias vectors in that order
    auto bias4 = [&](int base, int mb) {
        const float4 t = *reinterpret_cast<const float4 *>(Bs + base + 16 * mb + 4 * fq);
        return f32x4{t.x, t.y, t.z, t.w};
    };
    const bf16x8 *w1p = W1s + fn * 4 + fq, *w2p = W2s + fn * 4 + fq, *w3p = W3s + fn * 4 + fq;
    double cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cq[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // BN sums of this lane's 8 output features

    for (int64_t tile = t0; tile < t1; ++tile) {
        bf16x8 xb[3];
        {
            const float x[8] = {zin[0].x, zin[0].y, zin[0].z, zin[0].w, zin[1].x, zin[1].y, zin[1].z, zin[1].w};
            gin_split3(x, xb[0], xb[1], xb[2]);
        }
        load_z(tile + 1 < t1 ? tile + 1 : tile, zin);
        // ---- layer 1: 2 M blocks
        f32x4 h1a = gin_mma6(w1p + 0 * 64, 2 * 64, xb, bias4(0, 0));
        f32x4 h1b = gin_mma6(w1p + 1 * 64, 2 * 64, xb, bias4(0, 1));
        {
            const float x[8] = {sigmoidf_(h1a[0]), sigmoidf_(h1a[1]), sigmoidf_(h1a[2]), sigmoidf_(h1a[3]),
                                sigmoidf_(h1b[0]), sigmoidf_(h1b[1]), sigmoidf_(h1b[2]), sigmoidf_(h1b[3])};
            gin_split3(x, xb[0], xb[1], xb[2]);
        }
        // ---- layer 2: 4 M blocks
        f32x4 h2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) h2[mb] = gin_mma6(w2p + mb * 64, 4 * 64, xb, bias4(32, mb));
        // ---- layer 3: 2 M blocks x 2 K steps
        f32x4 o0 = bias4(96, 0), o1 = bias4(96, 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float x[8] = {sigmoidf_(h2[2 * ks][0]), sigmoidf_(h2[2 * ks][1]), sigmoidf_(h2[2 * ks][2]), sigmoidf_(h2[2 * ks][3]),
                                sigmoidf_(h2[2 * ks + 1][0]), sigmoidf_(h2[2 * ks + 1][1]), sigmoidf_(h2[2 * ks + 1][2]), sigmoidf_(h2[2 * ks + 1][3])};
            gin_split3(x, xb[0], xb[1], xb[2]);
            o0 = gin_mma6(w3p + (0 * 2 + ks) * 64, 256, xb, o0);
            o1 = gin_mma6(w3p + (1 * 2 + ks) * 64, 256, xb, o1);
        }
        // ---- epilogue: row fn, features 4 q + r and 16 + 4 q + r
        float4 r0, r1;
        // the OUTPUT sigmoid at the accuracy of expf + a division (tgnn_common.h: sigmoid_out_f32; the hardware transcendentals
        // alone lose an order of magnitude at |v| ~ 20): the BatchNorm behind this kernel divides columns that vary by ~1 % of their value, so every ulp here is
        // ~100 ulp there (CollConv incl. BatchNorm vs fp64: 1.4e-5 with sigmoidf_)
        auto sig_out = [](float v) { return sigmoid_out_f32(v); };
        r0.x = sig_out(o0[0]); r0.y = sig_out(o0[1]); r0.z = sig_out(o0[2]); r0.w = sig_out(o0[3]);
        r1.x = sig_out(o1[0]); r1.y = sig_out(o1[1]); r1.z = sig_out(o1[2]); r1.w = sig_out(o1[3]);
        if (act == TGNN_ACT_LEAKY_RELU) {
            r0.x = leakyf_(r0.x); r0.y = leakyf_(r0.y); r0.z = leakyf_(r0.z); r0.w = leakyf_(r0.w);
            r1.x = leakyf_(r1.x); r1.y = leakyf_(r1.y); r1.z = leakyf_(r1.z); r1.w = leakyf_(r1.w);
        }
        const int64_t row = tile * 16 + fn;
        if (row < n) {
            *reinterpret_cast<float4 *>(out + row * 32 + 4 * fq) = r0;
            *reinterpret_cast<float4 *>(out + row * 32 + 16 + 4 * fq) = r1;
            const float v[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] += (double)v[e];
                cq[e] += (double)v[e] * (double)v[e];
            }
        }
    }
    if (bn_partial) {
        // lanes of one 16-lane row hold the same 8 features: fold the 16 batch rows (fixed butterfly), then the waves
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int d = 1; d <= 8; d <<= 1) {
                cs[e] += __shfl_xor(cs[e], d, 64);
                cq[e] += __shfl_xor(cq[e], d, 64);
            }
        if (fn == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int feat = (e < 4 ? 0 : 16) + 4 * fq + (e & 3);
                red[wave * 64 + feat] = cs[e];
                red[wave * 64 + 32 + feat] = cq[e];
            }
        }
        __syncthreads();
        using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(bn_partial, 0, (int)0x80000000u, 0x00020000);
        constexpr int kSc1 = 16;                              // cache policy sc1: agent scope, coherent across the XCDs' L2s
        if (tid < 64) {
            double tot = 0.0;
            for (int w = 0; w < kMlpWaves; ++w) tot += red[w * 64 + tid];
            // (a write-through store when another block of THIS launch reads the row; an agent-scope release fence instead would
            //  write the XCD's whole L2 back -- this kernel's 12.8 MB of output -- once per block: 19 -> 84 us, measured)
            if (fin.counter)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, tot), prs, ((uint32_t)blockIdx.x * 64u + (uint32_t)tid) * 8u, 0, kSc1);
            else
                bn_partial[(int64_t)blockIdx.x * 64 + tid] = tot;
        }
        if (fin.counter) {
            // [r6] The BatchNorm's record in TWO levels (bn_finalize_kernel's tree -- 16 row groups p = g, g + 16, .. summed in
            // ascending p, then the groups in ascending g: the same bits -- no launch behind).  One last block folding all ~224 rows
            // was 7 us of serial tail per launch (ticket -> rows -> sums -> record with 255 CUs idle; without it the forward is 60 us
            // shorter: profiles/r06_gin_fold_ablation.txt).  Now the last block OF A GROUP (ticket fin.counter[1 + g]) folds its
            // group's ~14 rows into a group row as soon as they are there -- the groups finish at different times, most of this runs
            // under the other blocks' tiles -- and only the last group-finisher (ticket fin.counter[0]) adds 16 group rows and
            // writes the record.  Group rows: rows kGinGroupRow0 .. + 15 of bn_partial.  fin.counter: 17 zeroed words, left zeroed.
            __shared__ unsigned ticket;
            __shared__ double ftot[64];
            const int g = (int)(blockIdx.x & 15u), np = (int)gridDim.x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this block's row has been written through before its ticket
            __syncthreads();
            if (tid == 0) ticket = __hip_atomic_fetch_add(fin.counter + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned group_size = (unsigned)((np - g + 15) / 16);
#ifdef TGNN_ABL_NOFOLDWORK
            if (ticket == group_size - 1 && tid == 0) __hip_atomic_store(fin.counter + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (false) {                                      // (timing ablation: see bn_fold_finish)
#else
            if (ticket == group_size - 1) {                   // (uniform) the group's rows are all written
#endif
                if (tid < 64) {
                    // rows g, g + 16, ..: sixteen at a time (one batch up to 256 partial rows), added in ascending order
                    double acc = 0.0;
                    for (int u0 = 0; g + u0 * 16 < np; u0 += 16) {
                        u32x2 v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int pp = g + (u0 + u) * 16;
                            v[u] = __builtin_amdgcn_raw_buffer_load_b64(prs, pp < np ? ((uint32_t)pp * 64u + (uint32_t)tid) * 8u : 0x80000000u, 0, kSc1);
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            if (g + (u0 + u) * 16 < np) acc += __builtin_bit_cast(double, v[u]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, acc), prs, ((uint32_t)(kGinGroupRow0 + g) * 64u + (uint32_t)tid) * 8u, 0, kSc1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    __hip_atomic_store(fin.counter + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                const unsigned n_groups = (unsigned)(np < 16 ? np : 16);
                if (ticket == n_groups - 1) {                 // (uniform) all group rows are written
                    if (tid < 64) {
                        u32x2 v[16];
#pragma unroll
                        for (int gg = 0; gg < 16; ++gg)
                            v[gg] = __builtin_amdgcn_raw_buffer_load_b64(prs, gg < (int)n_groups ? ((uint32_t)(kGinGroupRow0 + gg) * 64u + (uint32_t)tid) * 8u : 0x80000000u, 0, kSc1);
                        double t = 0.0;
#pragma unroll
                        for (int gg = 0; gg < 16; ++gg)
                            if (gg < (int)n_groups) t += __builtin_bit_cast(double, v[gg]);
                        ftot[tid] = t;
                    }
                    __syncthreads();
                    bn_record_from_sums(fin.job, ftot, 32, fin.n_total, fin.eps, fin.momentum);
                    if (tid == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same MLP for the inference forward ([r5]; the kernel above stays for the training forward, whose adjoint tests pin its
// bits, and for layouts the checks below exclude).  What its counters said (profiles/r05_pmc_gin_mlp.txt): 675 vector
// instructions per 16-row tile against 60 matrix instructions, two waves per SIMD parked 40 % of their cycles -- it runs at the
// speed of its vector stream, and 40 % of that stream was the libm sigmoid of the output.  Here:
//   * layers 2 and 3 take their inputs -- sigmoids, in (0, 1) -- as fp16 PAIRS (tgnn_common.h: split2_f16) against weight images
//     scaled by a power of two (max |W| from the block's own prologue): 3 matrix terms instead of 6, 24 instead of 36 split
//     instructions per 8 values; the scale comes off inside the sigmoid's exponent (acc = s (W x + b); sigma(acc / s));
//     layer 1 (its input, the neighbourhood sum, has no bound at hand) stays on bf16 x 3.  36 matrix instructions per tile.
//   * the output sigmoid keeps its accuracy at a third of the instructions: exp2 on a two-part product (the rounding of
//     v log2 e is what costs accuracy at |v| ~ 10), one Newton step on the hardware reciprocal; LeakyReLU behind a sigmoid is the
//     identity and is not executed.
//   * the output sigmoid (tgnn_common.h: sigmoid_out_f32) is the one the kernel above has taken over since.
//   * 22 weight fragments in 88 registers (the kernel above: 30 in 120).  A first form re-read them from the LDS image every tile
//     to run 16 waves per block: 1 - 1.5 % slower in the forward.  As it stands the tile loop is 313 vector + 36 matrix
//     instructions (above: 377 + 60) and the forward takes exactly as long with either kernel (same-box pairs: 1.864 / 1.862 and
//     1.893 / 1.891 ms) -- which is why this one stays opt-in.
// ------------------------------------------------------------------------------------------
extern std::atomic<int> g_gin_mlp16;     // tgnn_set_gin_mlp_f16 (default off: see the note below)
constexpr int kMlp16Waves = 8, kMlp16Threads = kMlp16Waves * 64;      // (two waves per SIMD: the fragments' 206 VGPRs)
using f16x8g = tgnn_f16x8;

__global__ __launch_bounds__(kMlp16Threads) void gin32_mlp16_kernel(
    const float *__restrict__ z, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, int64_t n, float *__restrict__ out, double *__restrict__ bn_partial, GinFin fin) {
    // weight images: W1 [plane 3][M block 2][i 16][q 4] x bf16x8; W2 [plane 2][M block 4][..] x f16x8 (K in kf order);
    // W3 [plane 2][M block 2][K step 2][..] x f16x8
    __shared__ bf16x8 W1s[3 * 2 * 64];
    __shared__ f16x8g W2s[2 * 4 * 64];
    __shared__ f16x8g W3s[2 * 2 * 2 * 64];
    __shared__ __attribute__((aligned(16))) float Bs[128];   // b1 | s2 b2 | s3 b3
    __shared__ float wmax[2][kMlp16Waves];
    __shared__ double red[kMlp16Waves * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fn = lane & 15, fq = lane >> 4;

    const int64_t n_tiles = (n + 15) / 16;
    const int nblk = gridDim.x;
    int64_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (int64_t)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int64_t slot = blk * 4 + (wave & 3), n_slots = (int64_t)nblk * 4;
    const int64_t q0 = n_tiles * slot / n_slots, q1 = n_tiles * (slot + 1) / n_slots;
    constexpr int kSubs = kMlp16Waves / 4;
    const int sub = wave >> 2;
    const int64_t t0 = q0 + (q1 - q0) * sub / kSubs, t1 = q0 + (q1 - q0) * (sub + 1) / kSubs;

    auto load_z = [&](int64_t tile, float4 (&zin)[2]) {
        int64_t zr = tile * 16 + fn;                        // clamped, unconditional: rows >= n are masked at the store
        zr = zr < n ? zr : n - 1;
        const float4 *pz = reinterpret_cast<const float4 *>(z + zr * 32 + 8 * fq);
        zin[0] = pz[0];
        zin[1] = pz[1];
    };
    float4 zin[2];
    load_z(t0 < t1 ? t0 : 0, zin);                          // in flight while the weight images are built

    // ---- the scales of the fp16 images: max |W2|, max |W3| over the block (2 048 weights each)
    {
        float m2 = 0.f, m3 = 0.f;
        for (int i = tid; i < 2048; i += kMlp16Threads) {
            m2 = fmaxf(m2, fabsf(w2[i]));
            m3 = fmaxf(m3, fabsf(w3[i]));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            m2 = fmaxf(m2, __shfl_xor(m2, d, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, d, 64));
        }
        if (lane == 0) { wmax[0][wave] = m2; wmax[1][wave] = m3; }
    }
    __syncthreads();
    float s2, s3;
    {
        float m2 = 0.f, m3 = 0.f;
#pragma unroll
        for (int w = 0; w < kMlp16Waves; ++w) { m2 = fmaxf(m2, wmax[0][w]); m3 = fmaxf(m3, wmax[1][w]); }
        s2 = pow2_scale_for(__float_as_uint(m2), 0);         // s |w| < 2^15 (0, inf, nan: 1)
        s3 = pow2_scale_for(__float_as_uint(m3), 0);
    }
    for (int i = tid; i < 2 * 64; i += kMlp16Threads) {     // item = (M block, i, q): 8 weights
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w1[(16 * mb + ii) * 32 + 8 * q + e];
        gin_split3(x, W1s[(0 * 2 + mb) * 64 + ii * 4 + q], W1s[(1 * 2 + mb) * 64 + ii * 4 + q], W1s[(2 * 2 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += kMlp16Threads) {
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w2[(16 * mb + ii) * 32 + gin_kf(q, e)];
        split2_f16(x, s2, W2s[(0 * 4 + mb) * 64 + ii * 4 + q], W2s[(1 * 4 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += kMlp16Threads) {     // item = (M block, K step, i, q)
        const int mb = i >> 7, ks = (i >> 6) & 1, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w3[(16 * mb + ii) * 64 + 32 * ks + gin_kf(q, e)];
        const int o = (mb * 2 + ks) * 64 + ii * 4 + q;
        split2_f16(x, s3, W3s[0 * 256 + o], W3s[1 * 256 + o]);
    }
    if (tid < 32) Bs[tid] = b1[tid];
    else if (tid < 96) Bs[tid] = s2 * b2[tid - 32];
    else if (tid < 128) Bs[tid] = s3 * b3[tid - 96];
    __syncthreads();

    auto bias4 = [&](int base, int mb) {
        const float4 t = *reinterpret_cast<const float4 *>(Bs + base + 16 * mb + 4 * fq);
        return f32x4{t.x, t.y, t.z, t.w};
    };
    const bf16x8 *w1p = W1s + fn * 4 + fq;
    const f16x8g *w2p = W2s + fn * 4 + fq, *w3p = W3s + fn * 4 + fq;
    // sigma(acc / s) = 1 / (1 + 2^(acc * c)),  c = -log2 e / s  (s a power of two: exact)
    const float c2 = -1.44269504088896340736f / s2, inv3 = 1.0f / s3;
    auto sig2 = [&](float a) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * c2)); };
    double cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cq[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // BN sums of this lane's 8 output features
    // the 22 weight fragments and the biases live in registers (206 VGPRs, two waves per SIMD), as gin32_mlp_kernel keeps its 30:
    // re-read from the LDS image every tile (106 VGPRs, up to 16 waves per block) the forward was 1 - 1.5 % slower, same bits
    bf16x8 rw1[2][3];
    f16x8g rw2[4][2], rw3[4][2];
    f32x4 rb1[2], rb2[4], rb3[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) rw1[mb][pl] = w1p[mb * 64 + pl * 2 * 64];
        rb1[mb] = bias4(0, mb);
        rb3[mb] = bias4(96, mb);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            rw2[mb][pl] = w2p[mb * 64 + pl * 4 * 64];
            rw3[mb][pl] = w3p[mb * 64 + pl * 256];           // mb = M block * 2 + K step
        }
        rb2[mb] = bias4(32, mb);
    }
    auto mma6r = [&](const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], acc, 0, 0, 0);
        return acc;
    };
    // acc += W . X over one K step of 32, fp16 pairs: three cross terms, smallest first
    auto mma3r = [&](const f16x8g (&w)[2], const f16x8g &xh, const f16x8g &xl, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], xh, acc, 0, 0, 0);   // lo . hi
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], xl, acc, 0, 0, 0);   // hi . lo
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], xh, acc, 0, 0, 0);   // hi . hi
        return acc;
    };

    for (int64_t tile = t0; tile < t1; ++tile) {
        bf16x8 xb[3];
        {
            const float x[8] = {zin[0].x, zin[0].y, zin[0].z, zin[0].w, zin[1].x, zin[1].y, zin[1].z, zin[1].w};
            gin_split3(x, xb[0], xb[1], xb[2]);
        }
        load_z(tile + 1 < t1 ? tile + 1 : tile, zin);
        // ---- layer 1 (bf16 x 3): 2 M blocks
        const f32x4 h1a = mma6r(rw1[0], xb, rb1[0]);
        const f32x4 h1b = mma6r(rw1[1], xb, rb1[1]);
        f16x8g xh, xl;
        {
            const float x[8] = {sigmoidf_(h1a[0]), sigmoidf_(h1a[1]), sigmoidf_(h1a[2]), sigmoidf_(h1a[3]),
                                sigmoidf_(h1b[0]), sigmoidf_(h1b[1]), sigmoidf_(h1b[2]), sigmoidf_(h1b[3])};
            split2_f16(x, 1.0f, xh, xl);
        }
        // ---- layer 2 (fp16 pairs): 4 M blocks
        f32x4 h2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) h2[mb] = mma3r(rw2[mb], xh, xl, rb2[mb]);
        f32x4 o0 = rb3[0], o1 = rb3[1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float x[8] = {sig2(h2[2 * ks][0]), sig2(h2[2 * ks][1]), sig2(h2[2 * ks][2]), sig2(h2[2 * ks][3]),
                                sig2(h2[2 * ks + 1][0]), sig2(h2[2 * ks + 1][1]), sig2(h2[2 * ks + 1][2]), sig2(h2[2 * ks + 1][3])};
            split2_f16(x, 1.0f, xh, xl);
            o0 = mma3r(rw3[0 * 2 + ks], xh, xl, o0);
            o1 = mma3r(rw3[1 * 2 + ks], xh, xl, o1);
        }
        // ---- epilogue: row fn, features 4 q + r and 16 + 4 q + r.  (The BatchNorm behind this kernel divides columns that vary
        //      by ~1 % of their value: every ulp here is ~100 ulp there -- hence the careful sigmoid.  LeakyReLU of a sigmoid: identity.)
        float4 r0, r1;
        r0.x = sigmoid_out_f32(o0[0] * inv3); r0.y = sigmoid_out_f32(o0[1] * inv3); r0.z = sigmoid_out_f32(o0[2] * inv3); r0.w = sigmoid_out_f32(o0[3] * inv3);
        r1.x = sigmoid_out_f32(o1[0] * inv3); r1.y = sigmoid_out_f32(o1[1] * inv3); r1.z = sigmoid_out_f32(o1[2] * inv3); r1.w = sigmoid_out_f32(o1[3] * inv3);
        const int64_t row = tile * 16 + fn;
        if (row < n) {
            *reinterpret_cast<float4 *>(out + row * 32 + 4 * fq) = r0;
            *reinterpret_cast<float4 *>(out + row * 32 + 16 + 4 * fq) = r1;
            const float v[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] += (double)v[e];
                cq[e] += (double)v[e] * (double)v[e];
            }
        }
    }
    if (bn_partial) {
        // lanes of one 16-lane row hold the same 8 features: fold the 16 batch rows (fixed butterfly), then the waves
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int d = 1; d <= 8; d <<= 1) {
                cs[e] += __shfl_xor(cs[e], d, 64);
                cq[e] += __shfl_xor(cq[e], d, 64);
            }
        if (fn == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int feat = (e < 4 ? 0 : 16) + 4 * fq + (e & 3);
                red[wave * 64 + feat] = cs[e];
                red[wave * 64 + 32 + feat] = cq[e];
            }
        }
        __syncthreads();
        if (tid < 64) {
            double tot = 0.0;
            for (int w = 0; w < kMlp16Waves; ++w) tot += red[w * 64 + tid];
            // (written through when another block of THIS launch reads the row: see gin32_mlp_kernel)
            if (fin.counter) st_partial_sc1(bn_partial, (int64_t)blockIdx.x * 64 + tid, tot);
            else bn_partial[(int64_t)blockIdx.x * 64 + tid] = tot;
        }
        if (fin.counter) {
            // the last block to get here writes the BatchNorm's record (tgnn_common.h: bn_fold_finish -- bn_finalize_kernel's tree)
            __shared__ __attribute__((aligned(16))) unsigned char fold_scratch[bn_fold_scratch_bytes(32)];
            GinFin f = fin;
            f.job.partials = bn_partial;
            f.job.n_partials = (int)gridDim.x;
            bn_fold_finish<32>(f, bn_partial, reinterpret_cast<double *>(fold_scratch));
        }
    }
}

// ------------------------------------------------------------------------------------------
// The two kernels above as ONE (packed 128-byte rows): z never travels through HBM (25.6 MB written and read again per layer at
// 100 000 nodes: a fifth of the forward's excess traffic) and the layer costs one launch instead of two.
// A block = 16 waves in 4 TEAMS, one per SIMD: an MLP wave and three GATHER waves.  A gather wave sums the neighbourhoods of its
// tiles (every third tile of the team's share) -- whole 128-byte rows, 8 source rows of 2 x 8 destination rows per step,
// gin_tile_gather -- and hands z over through a ring of LDS tiles; the MLP wave takes the tiles from the ring in order, in the
// matrix layout, and runs gin_tile_mlp (forward_persist.h: the 32 -> 32 -> 64 -> 32 sigmoid MLP on bf16 x 3 fragments of an LDS
// image, the arithmetic of gin32_mlp_kernel, same bits).  The memory-bound and the matrix-bound half of the layer overlap instead
// of following each other, and twelve gather waves per CU keep its L2 -> L1 fill path (the bound of this op) busy (tried first:
// every wave gathers, then multiplies -- the sum of both, 63 us per layer inside the forward against 31 + 29 for the two kernels;
// then one streaming gather wave per MLP wave: 65 us, a wave's chain of dependent round trips per tile is too long).  BatchNorm partial row and the folded finalize (GinFin) as in gin32_mlp_kernel.
// ------------------------------------------------------------------------------------------
constexpr int kFusedWaves = 16, kFusedThreads = kFusedWaves * 64, kFusedGatherPerPair = 3, kFusedRing = 2 * kFusedGatherPerPair;
constexpr int kFusedLdsFloats = kSpGinFrags * 4 + kSpGinW + 128 + 4 * kMidTileFloats + 4 * kFusedRing * kMidTileFloats + 64;
constexpr size_t kFusedLdsBytes = (size_t)kFusedLdsFloats * 4 + (size_t)(4 * 64 + 16 * 64 + 64) * 8;

__global__ __launch_bounds__(kFusedThreads) void gin32_fused_kernel(
    const float *__restrict__ a, const float *__restrict__ in_stat, const int *__restrict__ rowptr, const int *__restrict__ col_src,
    const float *__restrict__ eps_p, const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3, int64_t n, float *__restrict__ out,
    double *__restrict__ bn_partial, GinFin fin) {
    extern __shared__ __attribute__((aligned(16))) float fl[];
    float *gw = fl;                                               // [W1 | W2 | W3] images, small_pack_kernel's layout
    float *spl = gw + kSpGinFrags * 4;                            // parameter vectors: 1 + eps, the three biases
    float *st2 = spl + kSpGinW;                                   // the producer's BatchNorm record
    float *tiles = st2 + 128;                                     // the MLP waves' own tiles
    float *rings = tiles + 4 * kMidTileFloats;                    // [team][slot] hand-over tiles
    volatile int *flags = reinterpret_cast<volatile int *>(rings + 4 * kFusedRing * kMidTileFloats);   // [team][slot]: tiles handed over / taken
    double *red = reinterpret_cast<double *>(fl + kFusedLdsFloats);   // [4][64]
    double *fred = red + 4 * 64, *ftot = fred + 16 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16x8 *W1s = reinterpret_cast<bf16x8 *>(gw), *W2s = W1s + 3 * 2 * 64, *W3s = W2s + 3 * 4 * 64;
    for (int i = tid; i < 2 * 64; i += kFusedThreads) {     // item = (M block, i, q): 8 weights
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w1[(16 * mb + ii) * 32 + 8 * q + e];
        gin_split3(x, W1s[(0 * 2 + mb) * 64 + ii * 4 + q], W1s[(1 * 2 + mb) * 64 + ii * 4 + q], W1s[(2 * 2 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += kFusedThreads) {
        const int mb = i >> 6, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w2[(16 * mb + ii) * 32 + gin_kf(q, e)];
        gin_split3(x, W2s[(0 * 4 + mb) * 64 + ii * 4 + q], W2s[(1 * 4 + mb) * 64 + ii * 4 + q], W2s[(2 * 4 + mb) * 64 + ii * 4 + q]);
    }
    for (int i = tid; i < 4 * 64; i += kFusedThreads) {     // item = (M block, K step, i, q)
        const int mb = i >> 7, ks = (i >> 6) & 1, ii = (i >> 2) & 15, q = i & 3;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w3[(16 * mb + ii) * 64 + 32 * ks + gin_kf(q, e)];
        const int o = (mb * 2 + ks) * 64 + ii * 4 + q;
        gin_split3(x, W3s[0 * 256 + o], W3s[1 * 256 + o], W3s[2 * 256 + o]);
    }
    if (tid < 32) spl[kSpGinB + tid] = b1[tid];
    else if (tid < 96) spl[kSpGinB + tid] = b2[tid - 32];
    else if (tid < 128) spl[kSpGinB + tid] = b3[tid - 96];
    if (tid == 128) spl[kSpEps] = 1.0f + eps_p[0];
    if (tid >= 256 && tid < 384) st2[tid - 256] = in_stat ? in_stat[tid - 256] : 0.f;
    if (tid >= 384 && tid < 384 + 8 * kFusedRing) flags[tid - 384] = 0;
    for (int i = tid; i < 4 * kMidTileFloats; i += kFusedThreads) tiles[i] = 0.f;
    __syncthreads();

    // the tile share of a team (= of a SIMD): as gin32_mlp_kernel shares the tiles out over the SIMD slots of the grid
    const int64_t n_tiles = (n + 15) / 16;
    const int nblk = gridDim.x, team = wave & 3, role = wave >> 2;   // role 0: the MLP wave, 1 .. 3: gather waves
    int64_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (int64_t)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int64_t slot = blk * 4 + team, n_slots = (int64_t)nblk * 4;
    const int64_t t0 = n_tiles * slot / n_slots, t1 = n_tiles * (slot + 1) / n_slots;
    const GinGraph G{rowptr, col_src, n};
    float *ring = rings + team * kFusedRing * kMidTileFloats;
    // ring slot s: ready[s] = 1 + the number of the last tile handed over there, taken[s] = 1 + the number of the last tile read
    volatile int *ready = flags + 2 * kFusedRing * team, *taken = ready + kFusedRing;
    double bn = 0.0;
    if (role > 0) {
        for (int64_t i = role - 1; t0 + i < t1; i += kFusedGatherPerPair) {
            const int s = (int)(i % kFusedRing);
            while (i >= kFusedRing && taken[s] < (int)(i - kFusedRing) + 1) __builtin_amdgcn_s_sleep(1);   // the slot's last tile has been read
            gin_tile_gather<0>(G, t0 + i, in_stat != nullptr, a, spl, st2, ring + s * kMidTileFloats, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) ready[s] = (int)i + 1;
        }
    } else {
        float *tbuf = tiles + team * kMidTileFloats;
        for (int64_t i = 0; t0 + i < t1; ++i) {
            const int s = (int)(i % kFusedRing);
            while (ready[s] < (int)i + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            gin_tile_mlp<0>(G, t0 + i, out, gw, spl, ring + s * kMidTileFloats, tbuf, lane, bn, taken + s, (int)i + 1);
        }
    }

    if (bn_partial) {
        if (role == 0) red[team * 64 + lane] = bn;           // lane = (channel, sum | sum of squares): the partial row's order
        __syncthreads();
        using u32x2_ = __attribute__((ext_vector_type(2))) unsigned int;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(bn_partial, 0, (int)0x80000000u, 0x00020000);
        if (tid < 64) {
            double tot = 0.0;
            for (int w = 0; w < 4; ++w) tot += red[w * 64 + tid];
            if (fin.counter)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, tot), prs, ((uint32_t)blockIdx.x * 64u + (uint32_t)tid) * 8u, 0, kCpSc1);
            else
                bn_partial[(int64_t)blockIdx.x * 64 + tid] = tot;
        }
        if (fin.counter) {                                    // the last block writes the BatchNorm's record (see gin32_mlp_kernel)
            __shared__ unsigned ticket;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (ticket == gridDim.x - 1) {                    // (uniform)
                const int j = tid & 63, h = tid >> 6;
                if (tid < 512) {                              // (the 8 waves of bn_finalize_kernel's tree)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int g = h + 8 * half;
                        double acc = 0.0;
                        const int np = (int)gridDim.x;
                        for (int p = g; p < np; p += 8 * 16) {
                            u32x2_ v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int pp = p + u * 16;
                                v[u] = __builtin_amdgcn_raw_buffer_load_b64(prs, pp < np ? ((uint32_t)pp * 64u + (uint32_t)j) * 8u : 0x80000000u, 0, kCpSc1);
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (p + u * 16 < np) acc += __builtin_bit_cast(double, v[u]);
                        }
                        fred[g * 64 + j] = acc;
                    }
                }
                __syncthreads();
                if (tid < 64) {
                    double t = 0.0;
                    for (int gg = 0; gg < 16; ++gg) t += fred[gg * 64 + tid];
                    ftot[tid] = t;
                }
                __syncthreads();
                bn_record_from_sums(fin.job, ftot, 32, fin.n_total, fin.eps, fin.momentum);
                if (tid == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Generic fallback (any C <= 256): one wave per row, everything through LDS / L2.
__global__ __launch_bounds__(256) void gin_generic_kernel(
    const float *__restrict__ a, int64_t lda, const float *__restrict__ in_stat, const int *__restrict__ rowptr,
    const int *__restrict__ col_src, const float *__restrict__ eps_p, const float *__restrict__ w1,
    const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
    const float *__restrict__ w3, const float *__restrict__ b3, int64_t n, int c, int act, float *__restrict__ out,
    double *__restrict__ bn_partial) {
    __shared__ float zs[4][256];
    __shared__ float h1s[4][32];
    __shared__ float h2s[4][64];
    __shared__ double red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float one_eps = 1.0f + eps_p[0];
    for (int k = lane; k < 2 * c; k += 64) red[wave][k] = 0.0;
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < n; v += (int64_t)gridDim.x * 4) {
        const int beg = rowptr[v], end = rowptr[v + 1];
        for (int k = lane; k < c; k += 64) {
            float mh = 0.f, ml = 0.f, gg = 1.f, bb = 0.f;
            if (in_stat) { mh = in_stat[k]; ml = in_stat[c + k]; gg = in_stat[2 * c + k]; bb = in_stat[3 * c + k]; }
            float acc = 0.f;
            for (int e = beg; e < end; ++e) acc += (a[(int64_t)col_src[e] * lda + k] - mh) - ml;
            const float self = (a[v * lda + k] - mh) - ml;
            zs[wave][k] = fmaf(gg, fmaf(one_eps, self, acc), (one_eps + (float)(end - beg)) * bb);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) {
            float acc = b1[lane];
            for (int i = 0; i < c; ++i) acc = fmaf(zs[wave][i], w1[lane * c + i], acc);
            h1s[wave][lane] = sigmoidf_(acc);
        }
        __builtin_amdgcn_wave_barrier();
        {
            float acc = b2[lane];
            for (int i = 0; i < 32; ++i) acc = fmaf(h1s[wave][i], w2[lane * 32 + i], acc);
            h2s[wave][lane] = sigmoidf_(acc);
        }
        __builtin_amdgcn_wave_barrier();
        for (int o = lane; o < c; o += 64) {
            float acc = b3[o];
            for (int i = 0; i < 64; ++i) acc = fmaf(h2s[wave][i], w3[o * 64 + i], acc);
            acc = sigmoidf_(acc);
            if (act == TGNN_ACT_LEAKY_RELU) acc = leakyf_(acc);
            out[v * c + o] = acc;
            red[wave][o] += (double)acc;
            red[wave][c + o] += (double)acc * (double)acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (bn_partial)
        for (int k = threadIdx.x; k < 2 * c; k += 256)
            bn_partial[(int64_t)blockIdx.x * 2 * c + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
}

}  // namespace tgnn

using namespace tgnn;
namespace tgnn { std::atomic<int> g_debug_block_cap[2]; static std::atomic<int> g_gin_fused{1}; std::atomic<int> g_gin_mlp16{0}; }
extern "C" int32_t tgnn_set_gin_mlp_f16(int32_t on) { return tgnn::g_gin_mlp16.exchange(on ? 1 : 0); }
extern "C" int32_t tgnn_set_gin_fused(int32_t mode) { return tgnn::g_gin_fused.exchange(mode < 0 ? 0 : (mode > 2 ? 2 : mode)); }
#ifdef TGNN_DEBUG
extern "C" void tgnn_debug_set_block_caps(int32_t nnconv_blocks, int32_t gin_mlp_blocks) {
    g_debug_block_cap[0].store(nnconv_blocks);
    g_debug_block_cap[1].store(gin_mlp_blocks);
}
#endif

namespace tgnn {
int gin32_fwd_folded(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr, const int32_t *col_src, const float *eps,
                     const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                     int64_t n_nodes, int32_t act, float *out, float *z_scratch, double *bn_partial, int32_t *n_partials_host,
                     const GinFin &fin, hipStream_t s, bool need_z) {
    if (!(n_nodes >= 1 && lda % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)out % 16) == 0 && z_scratch &&
          ((uintptr_t)z_scratch % 16) == 0 && bn_partial))
        return TGNN_ERR_UNSUPPORTED;
    // Measured, cached-layout forward, two kernels / fused: 20 000 nodes 0.880 / 0.916 ms, 100 000: 1.78 / 1.85, 300 000: 4.53 / 4.39;
    // the op alone (events, single stream) 43.6 / 38.1 us at 100 000 nodes, 94.0 / 83.3 at 300 000.  Inside the two-chain forward both
    // forms cost ~62 us of wall clock per layer at 100 000 nodes (the chains take the CUs in turns); the fused form wins where the
    // layer is long enough for its z traffic (25.6 MB per 100 000 nodes) to matter: from kGinFusedMinNodes on
    constexpr int64_t kGinFusedMinNodes = 200000;
    const int fused_mode = g_gin_fused.load(std::memory_order_relaxed);   // 0 never, 1 by size, 2 always
    if (!need_z && lda == 32 && act == TGNN_ACT_LEAKY_RELU && (fused_mode == 2 || (fused_mode == 1 && n_nodes >= kGinFusedMinNodes)) &&
        n_nodes * 128 < (int64_t(1) << 31)) {
        // one launch, no z round trip (gin32_fused_kernel)
        int blocks = producer_blocks(n_nodes, 16 * kMlpWaves);
        constexpr int reserve = 32;
        int cap = cus_minus(reserve);
        if (const int dbg = g_debug_block_cap[1].load(); dbg > 0) cap = dbg < device_cus() ? dbg : device_cus();
        if (blocks > cap) blocks = cap;
        if (blocks >= 8) blocks &= ~7;
        GinFin f = fin;
        f.job.partials = bn_partial;
        f.job.n_partials = blocks;
        static LdsOptIn site;
        TGNN_CHECK_HIP(opt_in_dynamic_lds(gin32_fused_kernel, (int)kFusedLdsBytes, site));
        gin32_fused_kernel<<<blocks, kFusedThreads, kFusedLdsBytes, s>>>(a, in_stat, rowptr, col_src, eps, w1, b1, w2, b2, w3, b3, n_nodes, out, bn_partial, f);
        if (n_partials_host) *n_partials_host = blocks;
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }
    const int64_t rows_per_xcd = (n_nodes + 7) / 8;
    const unsigned agg_blocks = (unsigned)(8 * ((rows_per_xcd + 31) / 32));
    gin32_aggregate_kernel<<<agg_blocks, 256, 0, s>>>(a, lda, in_stat, rowptr, col_src, eps, n_nodes, z_scratch);
    return launch_gin32_mlp(z_scratch, w1, b1, w2, b2, w3, b3, n_nodes, act, out, bn_partial, n_partials_host, s, &fin, !need_z);
}
}  // namespace tgnn

extern "C" int tgnn_gin_fwd(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr,
                            const int32_t *col_src, const float *eps, const float *w1, const float *b1,
                            const float *w2, const float *b2, const float *w3, const float *b3, int64_t n_nodes,
                            int32_t c, int32_t act, float *out, float *z_scratch, double *bn_partial,
                            int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0 && c >= 1 && c <= 256, "shape");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    if (n_nodes == 0) {
        if (n_partials_host) *n_partials_host = 0;
        return TGNN_OK;
    }
    TGNN_CHECK_ARG(a && rowptr && eps && w1 && b1 && w2 && b2 && w3 && b3 && out, "null pointer");
    TGNN_CHECK_ARG(lda >= c, "lda");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks;
    if (c == 32 && lda % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)out % 16) == 0 && z_scratch &&
        ((uintptr_t)z_scratch % 16) == 0) {
        const int64_t rows_per_xcd = (n_nodes + 7) / 8;
        const unsigned agg_blocks = (unsigned)(8 * ((rows_per_xcd + 31) / 32));
        gin32_aggregate_kernel<<<agg_blocks, 256, 0, s>>>(a, lda, in_stat, rowptr, col_src, eps, n_nodes, z_scratch);
        // persistent 8-wave blocks; the waves of one SIMD split a contiguous share of 32-row tiles
        blocks = producer_blocks(n_nodes, 16 * kMlpWaves);
        // one block per CU (the weight prologue is paid once per block), minus a few CUs left to the other chain's small
        // kernels (see launch_cols_t in nnconv_cols.hip)
        constexpr int reserve = 32;
        if (blocks > cus_minus(reserve)) blocks = cus_minus(reserve);
        if (blocks >= 8) blocks &= ~7;
        gin32_mlp_kernel<<<blocks, kMlpThreads, 0, s>>>(z_scratch, w1, b1, w2, b2, w3, b3, n_nodes, act, out,
                                                        bn_partial, GinFin{});
    } else {
        blocks = producer_blocks(n_nodes, 4);
        gin_generic_kernel<<<blocks, 256, 0, s>>>(a, lda, in_stat, rowptr, col_src, eps, w1, b1, w2, b2, w3, b3,
                                                  n_nodes, c, act, out, bn_partial);
    }
    if (n_partials_host) *n_partials_host = blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

namespace tgnn {
// the MLP half of tgnn_gin_fwd (width 32) on an aggregate tgnn_gin_aggregate wrote; forward.hip's sharded schedule runs the two
// halves on different streams
int launch_gin32_mlp(const float *z, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                     const float *b3, int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                     hipStream_t s, const GinFin *fin, bool inference) {
    if (inference && act == TGNN_ACT_LEAKY_RELU && g_gin_mlp16.load(std::memory_order_relaxed)) {
        // the inference forward's kernel (gin32_mlp16_kernel): fp16 pairs in layers 2 / 3, 16 waves per block
        int blocks = producer_blocks(n_nodes, 16 * kMlp16Waves);
        constexpr int reserve = 32;
        int cap = cus_minus(reserve);
        if (const int dbg = g_debug_block_cap[1].load(); dbg > 0) cap = dbg < device_cus() ? dbg : device_cus();
        if (blocks > cap) blocks = cap;
        if (blocks >= 8) blocks &= ~7;
        GinFin f{};
        if (fin && bn_partial) f = *fin;
        gin32_mlp16_kernel<<<blocks, kMlp16Threads, 0, s>>>(z, w1, b1, w2, b2, w3, b3, n_nodes, out, bn_partial, f);
        if (n_partials_host) *n_partials_host = blocks;
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }
    int blocks = producer_blocks(n_nodes, 16 * kMlpWaves);
    constexpr int reserve = 32;
    int cap = cus_minus(reserve);
    if (const int dbg = g_debug_block_cap[1].load(); dbg > 0) cap = dbg < device_cus() ? dbg : device_cus();
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks &= ~7;
    GinFin f{};
    if (fin && bn_partial) {
        f = *fin;
        f.job.partials = bn_partial;
        f.job.n_partials = blocks;
    }
    gin32_mlp_kernel<<<blocks, kMlpThreads, 0, s>>>(z, w1, b1, w2, b2, w3, b3, n_nodes, act, out, bn_partial, f);
    if (n_partials_host) *n_partials_host = blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
}  // namespace tgnn

/* z = (1 + eps) * BN_in(a) + sum over the row's CSR slots of BN_in(a)[src]  (the input of GINConv's MLP, PyG
 * gin_conv.py; width 32).  Stand-alone because the backward needs it twice per layer: to re-derive the MLP's input, and --
 * on the TRANSPOSED collision graph -- as the adjoint of the aggregation itself. */
extern "C" int tgnn_gin_aggregate(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr,
                                  const int32_t *col_src, const float *eps, int64_t n_nodes, int32_t c, float *z,
                                  tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0, "shape");
    if (c != 32) {
        set_error("tgnn_gin_aggregate: width 32 only");
        return TGNN_ERR_UNSUPPORTED;
    }
    if (n_nodes == 0) return TGNN_OK;
    TGNN_CHECK_ARG(a && rowptr && eps && z, "null pointer");
    TGNN_CHECK_ARG(lda >= c && lda % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)z % 16) == 0, "alignment");
    const int64_t rows_per_xcd = (n_nodes + 7) / 8;
    const unsigned agg_blocks = (unsigned)(8 * ((rows_per_xcd + 31) / 32));
    gin32_aggregate_kernel<<<agg_blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(a, lda, in_stat, rowptr, col_src, eps,
                                                                                    n_nodes, z);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
