// [r6] The init MLP of TilinGNN as three launches that RECOMPUTE instead of storing.
//
// Reference: init_node_feature_trans = MLP(Fx -> 32 -> 32, LeakyReLU, BatchNorm)  (/root/reference/graph_networks/networks/
// TilinGNN.py:31-33, called at :54; Linear_trans.forward, layers/util.py:31-37: Linear -> activation -> train-mode BatchNorm).
//
// Launch-per-op this is dense_in8 -> bn_finalize -> dense_mfma -> bn_finalize -> bn_apply: five kernels, 58 us on the critical
// chain in front of the first NNConv at 100 000 nodes for a [N, 3] input (profiles/r05_trace_100000.txt), three round trips of
// [N, 32] activations through HBM.  A row's whole MLP is ~1 200 fused multiply-adds on 3 .. 8 inputs, so every pass recomputes
// it from x (1.2 MB at 100 000 nodes) and nothing but the two BatchNorms' column sums and the result crosses HBM:
//   init_stats_kernel<0>   x -> LeakyReLU(W0 x + b0)                               -> fp64 column sums (BatchNorm 0)
//   init_stats_kernel<1>   ... -> BN0 -> LeakyReLU(W1 . + b1)                       -> fp64 column sums (BatchNorm 1)
//   init_apply_kernel      ... -> BN1 -> middle[0]  (+ its largest magnitude for the fp16-pair NNConv)
// A consumer derives the BatchNorm record from the producer's partial rows itself (bn32_fold_rows: bn_finalize_kernel's tree,
// the same bits) -- no finalize launch; block 0 writes the record and the running statistics.  Both Linears run on the exact-fp32
// matrix instruction, a 32-row tile per wave, with layer 0 computed TRANSPOSED so that its result registers are layer 1's operand
// (see below): nothing goes through LDS but the BatchNorm records.  (A first version with one thread per row on the vector pipe --
// ~1 200 FMAs per row, weights broadcast from LDS -- took 38 + 42 us: hipcc packs the chains into v_pk_fma_f32 behind 1.4 moves each.)
#include "tgnn_common.h"

namespace tgnn {

constexpr int kInitThreads = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct InitParams {
    const float *w0, *b0, *w1, *b1;     // W0 [32][fx], W1 [32][32]
    int fx;
};

struct InitLds {
    float st0[4 * 32];     // BatchNorm 0 record
    float st1[4 * 32];     // BatchNorm 1 record
    double red[16 * 64];
    double tot[64];
};

// The record [4][32] of a width-32 BatchNorm from its producer's partial rows (bn_finalize_kernel's tree: 16 row groups p = g, g + 16,
// .., each summed in ascending p, then the groups in ascending g -- the same bits), by all 256 threads of the block; `writer`:
// this block also stores the record and updates the running statistics.  Ends with a barrier.  At most 256 partial rows.
__device__ __forceinline__ void bn32_fold_rows(const BnJob &jb, int64_t n_total, float eps, float momentum, double *red, double *tot,
                                               float *st, bool writer) {
    constexpr int c = 32, two_f = 64, groups = 16;
    const int tid = threadIdx.x;
    float pre_gamma = 1.f, pre_beta = 0.f, pre_rm = 0.f, pre_rv = 1.f;
    if (tid < c) {
        pre_gamma = jb.gamma[tid];
        pre_beta = jb.beta[tid];
        if (writer && jb.running_mean) {
            pre_rm = jb.running_mean[tid];
            pre_rv = jb.running_var[tid];
        }
    }
    {
        // ONE round trip: a thread asks for all its rows at once (4 row groups x 16 rows of column j: up to 256 partial rows; a row
        // past the end re-reads row 0 and is left out of the sum), then adds them in bn_finalize_kernel's order
        const int j = tid & 63, g0 = (tid >> 6) * 4;          // this thread: row groups g0 .. g0 + 3 of column j
        const double *src = jb.partials + j;
        double v[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int p = g0 + u + q * groups;
                v[u][q] = src[(int64_t)(p < jb.n_partials ? p : 0) * two_f];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (g0 + u + q * groups < jb.n_partials) acc += v[u][q];
            red[(g0 + u) * two_f + j] = acc;
        }
    }
    __syncthreads();
    if (tid < two_f) {
        double t = 0.0;
        for (int gg = 0; gg < groups; ++gg) t += red[gg * two_f + tid];
        tot[tid] = t;
    }
    __syncthreads();
    if (tid < c) {
        const double inv_n = 1.0 / (double)n_total;
        const double mean = tot[tid] * inv_n;
        double var = tot[c + tid] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mh = (float)mean;
        st[tid] = mh;
        st[c + tid] = (float)(mean - (double)mh);
        st[2 * c + tid] = (float)((double)pre_gamma / sqrt(var + (double)eps));
        st[3 * c + tid] = pre_beta;
        if (writer) {
            if (jb.stat) {
                jb.stat[tid] = st[tid]; jb.stat[c + tid] = st[c + tid];
                jb.stat[2 * c + tid] = st[2 * c + tid]; jb.stat[3 * c + tid] = st[3 * c + tid];
            }
            if (jb.running_mean) {
                const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
                jb.running_mean[tid] = (float)((1.0 - (double)momentum) * (double)pre_rm + (double)momentum * mean);
                jb.running_var[tid] = (float)((1.0 - (double)momentum) * (double)pre_rv + (double)momentum * unbiased);
            }
            if (tid == 0 && jb.num_batches_tracked) *jb.num_batches_tracked += 1;
        }
    }
    __syncthreads();
}

// Both Linears on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), one 32-row tile per wave and step, nothing through LDS:
//   layouts: A operand lane (i = lane & 31, g = lane >> 5) = A[i][k = g]; B operand lane (j, g) = B[k = g][j];
//            C lane (j = lane & 31, fg = lane >> 5), register q = C[row (q & 3) + 8 (q >> 2) + 4 fg][j].
//   layer 0, plain:      T = X . W0^T   A = x rows, B = W0 -> C lane = CHANNEL, registers = rows: the column sums of init_stats_kernel<0>
//   layer 0, transposed: T^T = W0 . X^T (the same operand registers swapped; the same products, the same sums) -> C lane = ROW,
//                        registers = channels c(q, fg) -- which IS the A operand of layer 1 when step q of its K loop takes
//                        k = c(q, 0) from the lanes g = 0 and k = c(q, 1) from the lanes g = 1 (or, as below, one of the two and a zero)
//   layer 1:             V = BN0(T) . W1^T -> C lane = channel, registers = rows: column sums / BatchNorm 1 / 128-byte row stores
__device__ __forceinline__ int init_chan(int q, int fg) { return (q & 3) + 8 * (q >> 2) + 4 * fg; }

// The sums run in the order of the launch-per-op kernels they replace (dense_in8_kernel: bias first, then k ascending;
// dense_mfma_kernel: k ascending from zero, bias last), so that middle[0] comes out with THE SAME BITS (tests/test_hip_parity.py
// compares the two; the sharded forward, which keeps the launches, stays comparable with the single-device one at 1e-6): layer 0
// starts from the bias; layer 1 takes ONE k per matrix instruction -- the lanes of the other half feed a zero -- in ascending k,
// 32 instructions instead of 16 (the kernels are bound by latency, not by the matrix pipe).
struct InitRegs {
    float w0[4];       // lane (c, g): W0[c][2 s + g]
    float b0q[16];     // lane (., fg): b0[c(q, fg)]         (transposed layer 0)
    float w1[32];      // lane (c, .): W1[c][k]
    float b0c, b1c;    // lane (c, .): b0[c], b1[c]
};
__device__ __forceinline__ void init_load_regs(const InitParams &P, InitRegs &R, bool need_l1) {
    const int lane = threadIdx.x & 63, c = lane & 31, g = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) R.w0[s] = (2 * s + g) < P.fx ? P.w0[c * P.fx + 2 * s + g] : 0.f;
    R.b0c = P.b0[c];
    R.b1c = P.b1[c];
    if (need_l1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) R.b0q[q] = P.b0[init_chan(q, g)];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float4 w = *reinterpret_cast<const float4 *>(P.w1 + c * 32 + 4 * k4);
            R.w1[4 * k4] = w.x; R.w1[4 * k4 + 1] = w.y; R.w1[4 * k4 + 2] = w.z; R.w1[4 * k4 + 3] = w.w;
        }
    }
}
// x of the tile's rows as the MFMA operand: lane (r, g): x[row r][2 s + g]
__device__ __forceinline__ void init_load_x(const float *__restrict__ x, int64_t ldx, int fx, int64_t row0, int64_t n, float (&xr)[4]) {
    const int lane = threadIdx.x & 63, g = lane >> 5;
    int64_t row = row0 + (lane & 31);
    row = row < n ? row : n - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) xr[s] = (2 * s + g) < fx ? x[row * ldx + 2 * s + g] : 0.f;
}
// layer 1's pre-BatchNorm output of a tile (lane = channel, registers = rows) from x; st0 = BatchNorm 0's record (LDS)
__device__ __forceinline__ f32x16 init_tile_layer1(const float (&xr)[4], const InitRegs &R, const float *st0) {
    const int g = (threadIdx.x & 63) >> 5;
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 16; ++q) t[q] = R.b0q[q];
#pragma unroll
    for (int s = 0; s < 4; ++s) t = __builtin_amdgcn_mfma_f32_32x32x2f32(R.w0[s], xr[s], t, 0, 0, 0);   // T^T: lane = row
    float u[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = init_chan(q, g);
        u[q] = bn_apply1(leakyf_(t[q]), st0[c], st0[32 + c], st0[64 + c], st0[96 + c]);
    }
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        // channel k sits in register q of the lanes of half gk: k = (q & 3) + 8 (q >> 2) + 4 gk
        const int gk = (k >> 2) & 1, q = (k & 3) + 4 * (k >> 3);
        v = __builtin_amdgcn_mfma_f32_32x32x2f32(g == gk ? u[q] : 0.f, R.w1[k], v, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = leakyf_(v[q] + R.b1c);
    return v;
}
// adds a tile's column sums (lane = channel, registers = rows; rows >= n left out) to (s, q)
__device__ __forceinline__ void init_tile_sums(const f32x16 &v, int64_t row0, int64_t n, double &s, double &q) {
    const int fg = (threadIdx.x & 63) >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fg;
        if (row < n) {
            s += (double)v[r];
            q += (double)v[r] * (double)v[r];
        }
    }
}
// the block's partial row [sum 32 | sum of squares 32] from the lanes' (s, q): the two row halves of a wave, then the four waves
__device__ __forceinline__ void init_block_partial(double s, double q, double *red, double *part_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    if (lane < 32) {
        red[wave * 64 + lane] = s;
        red[wave * 64 + 32 + lane] = q;
    }
    __syncthreads();
    if (tid < 64) part_out[(int64_t)blockIdx.x * 64 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
}

// LAYER 0: column sums of layer 0's output; LAYER 1: of layer 1's (BatchNorm 0 from job0's partial rows)
template <int LAYER>
__global__ __launch_bounds__(kInitThreads) void init_stats_kernel(const float *__restrict__ x, int64_t ldx, InitParams P, BnJob job0,
                                                                   int64_t n, float eps, float momentum, double *__restrict__ part_out) {
    __shared__ InitLds L;
    const int tid = threadIdx.x, wave = tid >> 6;
    InitRegs R;
    init_load_regs(P, R, LAYER == 1);
    // a wave's tiles in chunks of four whose x rows are requested together: one memory round trip per chunk instead of one per
    // tile (a tile is ~1 us of work; at 100 000 rows a wave has 3.05 tiles) -- and the first chunk's while the partial rows are folded
    const int64_t tiles = (n + 31) / 32, stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    float xq[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) init_load_x(x, ldx, P.fx, (tile + i * stride < tiles ? tile + i * stride : 0) * 32, n, xq[i]);
    if (LAYER == 1) bn32_fold_rows(job0, n, eps, momentum, L.red, L.tot, L.st0, blockIdx.x == 0);
    double s = 0.0, q = 0.0;
    for (; tile < tiles; tile += 4 * stride) {
        float xn[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t t2 = tile + (4 + i) * stride;
            init_load_x(x, ldx, P.fx, (t2 < tiles ? t2 : 0) * 32, n, xn[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ti = tile + i * stride;
            if (ti < tiles) {                                 // (uniform per wave)
                f32x16 v;
                if (LAYER == 1) {
                    v = init_tile_layer1(xq[i], R, L.st0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = R.b0c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v = __builtin_amdgcn_mfma_f32_32x32x2f32(xq[i][k], R.w0[k], v, 0, 0, 0);   // T: lane = channel
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = leakyf_(v[r]);
                }
                init_tile_sums(v, ti * 32, n, s, q);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) xq[i][k] = xn[i][k];
    }
    __syncthreads();                                          // (L.red: the fold's reads are over)
    init_block_partial(s, q, L.red, part_out);
}

__global__ __launch_bounds__(kInitThreads) void init_apply_kernel(const float *__restrict__ x, int64_t ldx, InitParams P, BnJob job0,
                                                                   BnJob job1, int64_t n, float eps, float momentum,
                                                                   float *__restrict__ out, unsigned *__restrict__ absmax_out) {
    __shared__ InitLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, fg = lane >> 5;
    InitRegs R;
    init_load_regs(P, R, true);
    const int64_t tiles = (n + 31) / 32, stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    float xq[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) init_load_x(x, ldx, P.fx, (tile + i * stride < tiles ? tile + i * stride : 0) * 32, n, xq[i]);
    if (tid < 128) L.st0[tid] = job0.stat[tid];              // (the record init_stats_kernel<1>'s block 0 wrote)
    bn32_fold_rows(job1, n, eps, momentum, L.red, L.tot, L.st1, blockIdx.x == 0);
    const float mh = L.st1[c], ml = L.st1[32 + c], gi = L.st1[64 + c], be = L.st1[96 + c];
    float am = 0.f;
    for (; tile < tiles; tile += 4 * stride) {
        float xn[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t t2 = tile + (4 + i) * stride;
            init_load_x(x, ldx, P.fx, (t2 < tiles ? t2 : 0) * 32, n, xn[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ti = tile + i * stride;
            if (ti < tiles) {                                 // (uniform per wave)
                const f32x16 v = init_tile_layer1(xq[i], R, L.st0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
                    if (row < n) {
                        const float o = bn_apply1(v[r], mh, ml, gi, be);
                        out[row * 32 + c] = o;                // a half wave = the 128 bytes of one row
                        am = fmaxf(am, fabsf(o));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) xq[i][k] = xn[i][k];
    }
    absmax_flush(am, absmax_out);
}

// blocks of the three launches (= partial rows of the two BatchNorms, which every consumer block folds): one per CU at most
int init_mlp_fused_blocks(int64_t n) {
    const int64_t block_tiles = (n + 127) / 128;
    const int64_t cap = device_cus() < 256 ? device_cus() : 256;   // (bn32_fold_rows: 256 rows)
    return (int)(block_tiles < 1 ? 1 : (block_tiles > cap ? cap : block_tiles));
}

// x [n][fx] (fx <= 8) -> out [n][32] = middle[0]; job0 / job1: the two BatchNorms (partials = scratch of >= blocks x 64 doubles each,
// n_partials is filled in here); absmax_out: a zeroed word or NULL
int launch_init_mlp_fused(const float *x, int64_t ldx, int fx, const float *w0, const float *b0, const float *w1, const float *b1,
                          BnJob job0, BnJob job1, int64_t n, float eps, float momentum, float *out, unsigned *absmax_out,
                          hipStream_t s) {
    if (fx < 1 || fx > 8 || n < 1 || !job0.partials || !job1.partials || job0.partials == job1.partials) return TGNN_ERR_UNSUPPORTED;
    const int nb = init_mlp_fused_blocks(n);
    job0.n_partials = nb;
    job1.n_partials = nb;
    const InitParams P{w0, b0, w1, b1, fx};
    init_stats_kernel<0><<<nb, kInitThreads, 0, s>>>(x, ldx, P, job0, n, eps, momentum, const_cast<double *>(job0.partials));
    init_stats_kernel<1><<<nb, kInitThreads, 0, s>>>(x, ldx, P, job0, n, eps, momentum, const_cast<double *>(job1.partials));
    init_apply_kernel<<<nb, kInitThreads, 0, s>>>(x, ldx, P, job0, job1, n, eps, momentum, out, absmax_out);
    return TGNN_OK;
}

}  // namespace tgnn
