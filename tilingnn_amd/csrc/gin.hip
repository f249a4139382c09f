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
#include "tgnn_common.h"

namespace tgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------------------------
// K6: neighbourhood sum (HBM-bound gather).  8 lanes x float4 per destination row, rows of one
// XCD contiguous so that its private L2 holds the activations its gathers touch; no LDS, few
// registers -> full occupancy hides the rowptr -> col_src -> a[src] dependent-load chain.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gin32_aggregate_kernel(
    const float *__restrict__ a, int64_t lda, const float *__restrict__ in_stat, const int *__restrict__ rowptr,
    const int *__restrict__ col_src, const float *__restrict__ eps_p, int64_t n, float *__restrict__ z) {
    constexpr int C = 32;
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
// K7: the GIN MLP (32 -> 32 -> 64 -> 32, sigmoid after every Linear) on matrix cores.
// Block = 4 waves, tile = 128 rows; wave w owns rows [32w, 32w+32) through all three layers, so
// the hidden activations only ever travel wave-privately through LDS (no block barrier in the
// loop).  v_mfma_f32_32x32x2_f32: exact fp32.  LDS: weights 21 KB (rows padded: conflict-free
// fragment reads) + 12.5 KB per wave.
// ------------------------------------------------------------------------------------------
constexpr int kMlpThreads = 256;
constexpr int kW1Ld = 33, kW2Ld = 33, kW3Ld = 65;        // padded k-strides
constexpr int kBufALd = 65, kBufBLd = 33;                // per-wave buffers: A = Z then H2, B = H1

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kMlpThreads) void gin32_mlp_kernel(
    const float *__restrict__ z, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial) {
    __shared__ float W1s[32 * kW1Ld];
    __shared__ float W2s[64 * kW2Ld];
    __shared__ float W3s[32 * kW3Ld];
    __shared__ float bufA[4][32 * kBufALd];
    __shared__ float bufB[4][32 * kBufBLd];
    __shared__ double red[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    for (int i = tid; i < 32 * 32; i += kMlpThreads) W1s[(i >> 5) * kW1Ld + (i & 31)] = w1[i];
    for (int i = tid; i < 64 * 32; i += kMlpThreads) W2s[(i >> 5) * kW2Ld + (i & 31)] = w2[i];
    for (int i = tid; i < 32 * 64; i += kMlpThreads) W3s[(i >> 6) * kW3Ld + (i & 63)] = w3[i];
    const float bias1 = b1[fr], bias2a = b2[fr], bias2b = b2[32 + fr], bias3 = b3[fr];
    __syncthreads();

    float *A = bufA[wave], *B = bufB[wave];
    double csum = 0.0, csq = 0.0;
    const int64_t n_tiles = (n + 127) / 128;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * 128 + wave * 32;
        // ---- Z rows of this wave -> bufA (row stride 65): 4 x (64 lanes x float4) coalesced loads
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 64 + lane, r = idx >> 3, qq = idx & 7;
            int64_t zr = row0 + r;                       // clamped, unconditional: rows >= n are masked at the store
            zr = zr < n ? zr : n - 1;
            const float4 v4 = *reinterpret_cast<const float4 *>(z + zr * 32 + 4 * qq);
            float *d = A + r * kBufALd + 4 * qq;
            d[0] = v4.x; d[1] = v4.y; d[2] = v4.z; d[3] = v4.w;
        }
        wave_lds_fence();
        // ---- layer 1: H1 = sigmoid(Z W1^T + b1)          [32 x 32], K = 32
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[fr * kBufALd + kk + fk], W1s[fr * kW1Ld + kk + fk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            B[((r & 3) + 8 * (r >> 2) + 4 * fk) * kBufBLd + fr] = sigmoidf_(acc[r] + bias1);
        wave_lds_fence();
        // ---- layer 2: H2 = sigmoid(H1 W2^T + b2)         [32 x 64], K = 32
        f32x16 acc2a, acc2b;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2a[r] = acc2b[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float av = B[fr * kBufBLd + kk + fk];
            acc2a = __builtin_amdgcn_mfma_f32_32x32x2f32(av, W2s[fr * kW2Ld + kk + fk], acc2a, 0, 0, 0);
            acc2b = __builtin_amdgcn_mfma_f32_32x32x2f32(av, W2s[(32 + fr) * kW2Ld + kk + fk], acc2b, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fk;
            A[row * kBufALd + fr] = sigmoidf_(acc2a[r] + bias2a);       // Z is dead: reuse bufA
            A[row * kBufALd + 32 + fr] = sigmoidf_(acc2b[r] + bias2b);
        }
        wave_lds_fence();
        // ---- layer 3: OUT = act(sigmoid(H2 W3^T + b3))    [32 x 32], K = 64
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[fr * kBufALd + kk + fk], W3s[fr * kW3Ld + kk + fk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
            float v = sigmoidf_(acc[r] + bias3);
            if (act == TGNN_ACT_LEAKY_RELU) v = leakyf_(v);
            if (row < n) {
                out[row * 32 + fr] = v;
                csum += (double)v;
                csq += (double)v * (double)v;
            }
        }
        wave_lds_fence();   // bufA is rewritten by the next tile's Z
    }
    if (bn_partial) {
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (lane < 32) {
            red[wave * 64 + lane] = csum;
            red[wave * 64 + 32 + lane] = csq;
        }
        __syncthreads();
        if (tid < 64)
            bn_partial[(int64_t)blockIdx.x * 64 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
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

extern "C" int tgnn_gin_fwd(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr,
                            const int32_t *col_src, const float *eps, const float *w1, const float *b1,
                            const float *w2, const float *b2, const float *w3, const float *b3, int64_t n_nodes,
                            int32_t c, int32_t act, float *out, float *z_scratch, double *bn_partial,
                            int32_t *n_partials_host, tgnn_stream_t stream) {
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
        blocks = producer_blocks(n_nodes, 128);
        gin32_mlp_kernel<<<blocks, kMlpThreads, 0, s>>>(z_scratch, w1, b1, w2, b2, w3, b3, n_nodes, act, out,
                                                        bn_partial);
    } else {
        blocks = producer_blocks(n_nodes, 4);
        gin_generic_kernel<<<blocks, 256, 0, s>>>(a, lda, in_stat, rowptr, col_src, eps, w1, b1, w2, b2, w3, b3,
                                                  n_nodes, c, act, out, bn_partial);
    }
    if (n_partials_host) *n_partials_host = blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
