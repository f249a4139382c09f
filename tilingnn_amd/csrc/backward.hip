// Backward kernels of the training step.
//
// Reference: Trainer.train, /root/reference/solver/ml_solver/trainer.py:68-84 -- forward in train mode, the
// unsupervised loss (solver/ml_solver/losses.py:48-116), loss.backward(), optimizer.step().  torch.autograd derives the
// adjoints there; here they are written out per forward kernel:
//
//   Linear_trans (layers/util.py:31-37)     dX = dZ . W (tgnn_dense_act_fwd on W^T, tgnn_transpose), dW = dZ^T . X
//                                           (tgnn_wgrad, fp32 matrix cores), db = column sums (tgnn_colsum)
//   BatchNorm1d, train mode                 dz = act'(a) gamma invstd (dy - mean(dy) - xhat mean(dy xhat)):
//                                           one pass for the three column sums (dy, dy c, c^2 with c = a - mean; the
//                                           variance is re-derived here, the forward's record only carries gamma invstd),
//                                           a finalize, one pass to apply
//   branch merge (TilinGNN.py:64-71)        dy1 = dh y2, dy2 = dh y1 + carry, residual += dh; fused with the two
//                                           BatchNorm reductions that follow it (tgnn_merge_bwd_reduce)
//   NNConv mean (edge_conv.py:25)           per node the sums of gathered g = dz / deg rows per edge type over the transposed
//                                           graph (tgnn_nnconv_type_sum) turn both the weight gradient and the input
//                                           gradient into plain dense products over [N, (T+1) 32] (tilingnn_amd/train.py)
//   GIN MLP sigmoids (coll_conv.py:14-18)   d . t (1 - t)  (tgnn_sigmoid_bwd)
//   loss (losses.py:48-116)                 d loss / d probs: per-edge terms scattered with fp64 atomics, the area term and
//                                           the product rule in a second pass (tgnn_unsupervised_loss_bwd)
//
// All column sums run in fp64 over fixed trees (per-block partials, then one pass over the partial rows), like the
// forward's BatchNorm statistics; the only atomics are the loss scatter's (fp64: the order shows up at 1e-16).
#include "tgnn_common.h"

namespace tgnn {

constexpr int kRedThreads = 256;           // 64 columns x 4 row lanes
constexpr int kMaxRedPartials = 512;

static inline int red_partials(int64_t n) {
    int64_t p = (n + 127) / 128;
    return (int)(p < 1 ? 1 : (p > kMaxRedPartials ? kMaxRedPartials : p));
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ void transpose_kernel(const float *__restrict__ w, int rows, int cols, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)rows * cols) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        out[(int64_t)c * rows + r] = w[i];
    }
}

// up to three transposes and one zero-fill in ONE launch (grid.y = job; the last job is the fill): the weight matrices of a
// 3-layer MLP for the dx products of its backward
struct TransposeJobs {
    const float *src[3];
    float *dst[3];
    int rows[3], cols[3];
    int n;
    float *zero;
    int n_zero;
};
__global__ void transpose_multi_kernel(TransposeJobs jobs) {
    const int j = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == jobs.n) {
        for (int64_t i = t0; i < jobs.n_zero; i += stride) jobs.zero[i] = 0.f;
        return;
    }
    const int rows = jobs.rows[j], cols = jobs.cols[j];
    const float *src = jobs.src[j];
    float *dst = jobs.dst[j];
    for (int64_t i = t0; i < (int64_t)rows * cols; i += stride) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        dst[(int64_t)c * rows + r] = src[i];
    }
}

// out[b][a][c] = in[a][b][c], the rows of `out` out_da (>= da) entries long
__global__ void swap_leading_kernel(const float *__restrict__ in, int da, int db, int dc, float *__restrict__ out,
                                    int out_da) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)da * db * dc) {
        const int c = (int)(i % dc);
        const int b = (int)((i / dc) % db);
        const int a = (int)(i / ((int64_t)dc * db));
        out[((int64_t)b * out_da + a) * dc + c] = in[i];
    }
}

__global__ void sigmoid_bwd_kernel(const float *__restrict__ d, int64_t ld_d, const float *__restrict__ t, int64_t ld_t,
                                   int64_t n, int c, float *__restrict__ out, int64_t ld_o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * c) {
        const int64_t r = i / c;
        const int k = (int)(i % c);
        const float tv = t[r * ld_t + k];
        out[r * ld_o + k] = d[r * ld_d + k] * tv * (1.0f - tv);
    }
}

__global__ void add_into_kernel(const float *__restrict__ src, int64_t ld_s, int64_t n, int c, float *__restrict__ dst,
                                int64_t ld_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * c) {
        const int64_t r = i / c;
        const int k = (int)(i % c);
        dst[r * ld_d + k] += src[r * ld_s + k];
    }
}

// ------------------------------------------------------------------------------------------------ column sums
// grid (P, ceil(c / 64)); partial [P][c]
__global__ __launch_bounds__(kRedThreads) void colsum_partial_kernel(const float *__restrict__ x, int64_t ld, int64_t n,
                                                                     int c, int64_t rows_per_block,
                                                                     double *__restrict__ partial) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    double s = 0.0;
    if (col < c)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += (double)x[r * ld + col];
    __shared__ double sh[4][64];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < c) partial[(int64_t)blockIdx.x * c + col] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

// Sum of the partial rows of one column over a fixed tree: 16 lanes take rows l, l + 16, ..; then a butterfly.
// (A first version let one thread walk the <= 512 rows: 73 us per call, a quarter of the whole backward.)
constexpr int kFinLanes = 16, kFinCols = 16;            // block = 256 threads = 16 columns x 16 lanes
__device__ __forceinline__ double partial_column_sum(const double *__restrict__ base, int64_t row_stride, int n_partials,
                                                     int lane) {
    double s = 0.0;
    for (int p = lane; p < n_partials; p += kFinLanes) s += base[(int64_t)p * row_stride];
#pragma unroll
    for (int d = kFinLanes / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    return s;
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const double *__restrict__ partial, int n_partials, int c,
                                                           float *__restrict__ out) {
    const int lane = threadIdx.x & (kFinLanes - 1);
    const int col = blockIdx.x * kFinCols + (threadIdx.x >> 4);
    const int colc = col < c ? col : c - 1;              // keep the whole wave in the butterfly
    const double s = partial_column_sum(partial + colc, c, n_partials, lane);
    if (lane == 0 && col < c) out[col] = (float)s;
}

// ------------------------------------------------------------------------------------------------ BatchNorm backward
// partial [P][3][F]: sum dy, sum dy c, sum c^2 with c = (a - mean_hi) - mean_lo of the forward's record
__global__ __launch_bounds__(kRedThreads) void bn_bwd_reduce_kernel(const float *__restrict__ dy, int64_t ld_dy,
                                                                    const float *__restrict__ a, int64_t ld_a,
                                                                    const float *__restrict__ stat, int64_t n, int f,
                                                                    int64_t rows_per_block, double *__restrict__ partial) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (col < f) {
        const float mhi = stat[col], mlo = stat[f + col];
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const float g = dy[r * ld_dy + col];
            const float c = (a[r * ld_a + col] - mhi) - mlo;
            s0 += (double)g;
            s1 += (double)g * (double)c;
            s2 += (double)c * (double)c;
        }
    }
    __shared__ double sh[3][4][64];
    sh[0][rl][cl] = s0; sh[1][rl][cl] = s1; sh[2][rl][cl] = s2;
    __syncthreads();
    if (rl < 3 && col < f)
        partial[((int64_t)blockIdx.x * 3 + rl) * f + col] = (sh[rl][0][cl] + sh[rl][1][cl]) + (sh[rl][2][cl] + sh[rl][3][cl]);
}

// partial rows of `row_doubles` doubles; this set's three sums start at `offset`.  coef [2][F] = mean(dy),
// invstd^2 mean(dy c); dgamma = invstd sum(dy c); dbeta = sum(dy).
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double *__restrict__ partial, int64_t row_doubles, int64_t offset,
                                       int n_partials, int f, int64_t n_rows, float eps, float *__restrict__ coef,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int lane = threadIdx.x & (kFinLanes - 1);
    const int col_raw = blockIdx.x * kFinCols + (threadIdx.x >> 4);
    const int col = col_raw < f ? col_raw : f - 1;
    const double *base = partial + offset + col;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;               // one sweep, three independent chains
    for (int p = lane; p < n_partials; p += kFinLanes) {
        const double *row = base + (int64_t)p * row_doubles;
        s0 += row[0];
        s1 += row[f];
        s2 += row[2 * f];
    }
#pragma unroll
    for (int d = kFinLanes / 2; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d);
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    if (lane != 0 || col_raw >= f) return;
    const double inv_n = 1.0 / (double)n_rows;
    const double invstd = 1.0 / sqrt(s2 * inv_n + (double)eps);
    coef[col] = (float)(s0 * inv_n);
    coef[f + col] = (float)(invstd * invstd * s1 * inv_n);
    if (dgamma) dgamma[col] = (float)(invstd * s1);
    if (dbeta) dbeta[col] = (float)s0;
}

// dz = act'(a) ginv (dy - k0 - c k1);  act: TGNN_ACT_NONE or TGNN_ACT_LEAKY_RELU (a = leaky(z): a > 0 <=> z > 0).
// scaled (may be NULL) = dz * row_scale[r].
__global__ void bn_bwd_apply_kernel(const float *__restrict__ dy, int64_t ld_dy, const float *__restrict__ a, int64_t ld_a,
                                    const float *__restrict__ stat, const float *__restrict__ coef, int64_t n, int f,
                                    int act, float *__restrict__ dz, int64_t ld_dz, const float *__restrict__ row_scale,
                                    float *__restrict__ scaled, int64_t ld_s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * f) return;
    const int64_t r = i / f;
    const int col = (int)(i % f);
    const float av = a[r * ld_a + col];
    const float c = (av - stat[col]) - stat[f + col];
    float v = stat[2 * f + col] * (dy[r * ld_dy + col] - coef[col] - c * coef[f + col]);
    if (act == TGNN_ACT_LEAKY_RELU && !(av > 0.f)) v *= kLeakySlope;
    dz[r * ld_dz + col] = v;
    if (scaled) scaled[r * ld_s + col] = v * row_scale[r];
}

// ------------------------------------------------------------------------------------------------ merge backward
// h_out = BN1(a1) * BN2(a2) (+ resid).  Given dh (row stride ld_dh):  dy1 = dh y2, dy2 = dh y1 (+ carry),
// resid_grad (may be NULL, row stride ld_r) += dh, and the six column sums the two BatchNorm backward passes need:
// partial [P][2][3][32].  Width 32: 8 threads x float4 per row.
__global__ __launch_bounds__(256) void merge_bwd_reduce_kernel(
    const float *__restrict__ dh, int64_t ld_dh, const float *__restrict__ a1, const float *__restrict__ stat1,
    const float *__restrict__ a2, const float *__restrict__ stat2, const float *__restrict__ carry, int64_t n,
    int64_t rows_per_block, float *__restrict__ dy1, float *__restrict__ dy2, float *__restrict__ resid_grad,
    int64_t ld_r, double *__restrict__ partial) {
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;          // channels 4 cg .. 4 cg + 3, row lane 0..31
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float m1h[4], m1l[4], g1[4], b1[4], m2h[4], m2l[4], g2[4], b2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ch = 4 * cg + k;
        m1h[k] = stat1[ch]; m1l[k] = stat1[32 + ch]; g1[k] = stat1[64 + ch]; b1[k] = stat1[96 + ch];
        m2h[k] = stat2[ch]; m2l[k] = stat2[32 + ch]; g2[k] = stat2[64 + ch]; b2[k] = stat2[96 + ch];
    }
    double s[6][4];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[q][k] = 0.0;
    for (int64_t r = r0 + rl; r < r1; r += 32) {
        const float4 d4 = *reinterpret_cast<const float4 *>(dh + r * ld_dh + 4 * cg);
        const float4 x1 = *reinterpret_cast<const float4 *>(a1 + r * 32 + 4 * cg);
        const float4 x2 = *reinterpret_cast<const float4 *>(a2 + r * 32 + 4 * cg);
        float4 cr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (carry) cr = *reinterpret_cast<const float4 *>(carry + r * 32 + 4 * cg);
        const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, v1[4] = {x1.x, x1.y, x1.z, x1.w}, v2[4] = {x2.x, x2.y, x2.z, x2.w};
        const float cv[4] = {cr.x, cr.y, cr.z, cr.w};
        float o1[4], o2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c1 = (v1[k] - m1h[k]) - m1l[k], c2 = (v2[k] - m2h[k]) - m2l[k];
            const float y1 = c1 * g1[k] + b1[k], y2 = c2 * g2[k] + b2[k];
            o1[k] = dv[k] * y2;
            o2[k] = dv[k] * y1 + cv[k];
            s[0][k] += (double)o1[k]; s[1][k] += (double)o1[k] * (double)c1; s[2][k] += (double)c1 * (double)c1;
            s[3][k] += (double)o2[k]; s[4][k] += (double)o2[k] * (double)c2; s[5][k] += (double)c2 * (double)c2;
        }
        *reinterpret_cast<float4 *>(dy1 + r * 32 + 4 * cg) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<float4 *>(dy2 + r * 32 + 4 * cg) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        if (resid_grad) {
            float4 *rg = reinterpret_cast<float4 *>(resid_grad + r * ld_r + 4 * cg);
            float4 old = *rg;
            old.x += d4.x; old.y += d4.y; old.z += d4.z; old.w += d4.w;
            *rg = old;
        }
    }
    // fixed tree: over the 8 row lanes of a wave (lane bits 3..5), then over the 4 waves through LDS
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = s[q][k];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            s[q][k] = v;
        }
    __shared__ double sh[4][6][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) sh[wave][q][4 * lane + k] = s[q][k];
    }
    __syncthreads();
    if (threadIdx.x < 192) {
        const int q = threadIdx.x / 32, ch = threadIdx.x % 32;
        partial[((int64_t)blockIdx.x * 6 + q) * 32 + ch] = (sh[0][q][ch] + sh[1][q][ch]) + (sh[2][q][ch] + sh[3][q][ch]);
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// out[co][ci] = sum_r dz[r][co] * x[r][ci]  on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation per
// wave over its rows, fp64 across waves' partial tiles in the final pass).  Block = 4 waves over one (32 TM) x (32 TN)
// output tile and one row range; wave w takes rows r0 + 2w + 8i (+ lane >> 5).
// x element (r, k): x[(k / 32) * x_kblock_stride + r * ld_x + k % 32] when x_kblock_stride != 0 (the slot-major
// skip buffer), else x[r * ld_x + k].
template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_kernel(const float *__restrict__ dz, int64_t ld_dz,
                                                    const float *__restrict__ x, int64_t ld_x, int64_t x_kblock_stride,
                                                    int64_t n, int cout, int cin, int64_t rows_per_block,
                                                    float *__restrict__ partial, int64_t partial_stride, int with_bias,
                                                    float *__restrict__ direct_bias) {
    // direct_bias != NULL: there is ONE row range; `partial` is the result itself and the bias gradient goes to direct_bias
    typedef float f16v __attribute__((ext_vector_type(16)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int co0 = blockIdx.y * 32 * TM, ci0 = blockIdx.z * 32 * TN;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    f16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    int64_t xoff[TN];
    bool xok[TN], zok[TM];
    float bsum[TM];                     // column sums of dz (the bias gradient), taken by the blocks of the first x tile
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + 32 * j + (lane & 31);
        xok[j] = ci < cin;
        xoff[j] = x_kblock_stride ? (int64_t)(ci >> 5) * x_kblock_stride + (ci & 31) : (int64_t)ci;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) zok[i] = co0 + 32 * i + (lane & 31) < cout;
    // two row pairs per trip, all loads of a trip issued before its first MFMA (the loop is bound by load latency: the
    // compiler keeps one trip in flight, and a block walks its rows alone)
    for (int64_t rb = r0 + 2 * wave; rb < r1; rb += 16) {                  // uniform trip count: MFMA needs the whole wave
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t r = rb + 8 * h + (lane >> 5);
            const bool ok = r < r1;
#pragma unroll
            for (int i = 0; i < TM; ++i) av[h][i] = ok && zok[i] ? dz[r * ld_dz + co0 + 32 * i + (lane & 31)] : 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[h][j] = ok && xok[j] ? x[r * ld_x + xoff[j]] : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < TM; ++i) bsum[i] += av[h][i];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[h][i], bv[h][j], acc[i][j], 0, 0, 0);
        }
    }
    // cross-wave sum through LDS, one output tile at a time
    __shared__ float sh[4][32 * 32];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);          // m = co within the tile
                sh[wave][row * 32 + (lane & 31)] = acc[i][j][q];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 1024; e += 256) {
                const int co = co0 + 32 * i + (e >> 5), ci = ci0 + 32 * j + (e & 31);
                if (co < cout && ci < cin)
                    partial[(int64_t)blockIdx.x * partial_stride + (int64_t)co * cin + ci] =
                        (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
            }
        }
    if (with_bias && blockIdx.z == 0) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float v = bsum[i] + __shfl_xor(bsum[i], 32);
            if (lane < 32) sh[wave][32 * i + lane] = v;
        }
        __syncthreads();
        if (threadIdx.x < 32 * TM) {
            const int co = co0 + threadIdx.x;
            if (co < cout) {
                const float v = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
                if (direct_bias) direct_bias[co] = v;
                else partial[(int64_t)blockIdx.x * partial_stride + (int64_t)cout * cin + co] = v;
            }
        }
    }
}

// 64 consecutive elements x 4 lanes over the partial tiles (coalesced 256-byte rows), fixed tree through LDS
// elements [0, w_elems) go to `out`, [w_elems, elems) to `dbias`.  16 consecutive elements x 16 lanes over the partial
// tiles, butterfly over the lanes (fixed tree).  (4 lanes per element left every thread 128 dependent-latency loads at 512
// partial tiles: 21 us per call, the largest single item of the training step.)
__global__ __launch_bounds__(256) void wgrad_final_kernel(const float *__restrict__ partial, int n_partials, int64_t elems,
                                                          int64_t w_elems, float *__restrict__ out,
                                                          float *__restrict__ dbias) {
    const int pl = threadIdx.x & 15, el = threadIdx.x >> 4;
    const int64_t i_raw = (int64_t)blockIdx.x * 16 + el;
    const int64_t i = i_raw < elems ? i_raw : elems - 1;
    double s = 0.0;
    for (int p = pl; p < n_partials; p += 16) s += (double)partial[(int64_t)p * elems + i];
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (pl == 0 && i_raw < elems) {
        if (i < w_elems) out[i] = (float)s;
        else dbias[i - w_elems] = (float)s;
    }
}

static inline void wgrad_plan(int64_t n, int cout, int cin, int &tm, int &tn, int &parts, int64_t &rows_per_block) {
    tm = cout > 32 ? 2 : 1;
    tn = cin > 32 ? 2 : 1;
    const int tiles = ((cout + 32 * tm - 1) / (32 * tm)) * ((cin + 32 * tn - 1) / (32 * tn));
    int64_t want = (2048 + tiles - 1) / tiles;
    const int64_t max_by_rows = (n + 63) / 64;
    if (want > max_by_rows) want = max_by_rows;
    if (want > 512) want = 512;
    if (want < 1 || n <= 1024) want = 1;       // few rows: one row range, the result written directly (no second launch)
    rows_per_block = ((n + want - 1) / want + 15) / 16 * 16;
    if (rows_per_block < 16) rows_per_block = 16;
    parts = (int)((n + rows_per_block - 1) / rows_per_block);
    if (parts < 1) parts = 1;
}

// ------------------------------------------------------------------------------------------------ NNConv type sums
// out[j][t][:] = sum over the CSR slots e of row j with type[e] == t of rows[src[e]][:]   (t < T)
// out[j][T][:] = own[j][:] * root_scale[j]                                               (root_scale NULL: 1)
// Width 32: a half wave per destination row, its T + 1 accumulator rows in LDS (lane = channel: no two lanes ever
// touch the same word, so no barrier), four gathers in flight.
constexpr int kTsRows = 8;                  // rows (half waves) per block
__global__ __launch_bounds__(256) void nnconv_type_sum_kernel(const float *__restrict__ rows, int64_t ld_rows,
                                                              const float *__restrict__ own, int64_t ld_own,
                                                              const float *__restrict__ root_scale,
                                                              const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ src,
                                                              const int32_t *__restrict__ type, int64_t n, int n_types,
                                                              float *__restrict__ out) {
    extern __shared__ float acc_all[];
    const int hw = threadIdx.x >> 5, lane = threadIdx.x & 31;
    float *acc = acc_all + (int64_t)hw * (n_types + 1) * 32;
    const int64_t ld_out = (int64_t)(n_types + 1) * 32;
    for (int64_t j = (int64_t)blockIdx.x * kTsRows + hw; j < n; j += (int64_t)gridDim.x * kTsRows) {
        for (int t = 0; t <= n_types; ++t) acc[t * 32 + lane] = 0.f;
        const int e0 = rowptr[j], e1 = rowptr[j + 1];
        int e = e0;
        for (; e + 4 <= e1; e += 4) {
            const int s0 = src[e], s1 = src[e + 1], s2 = src[e + 2], s3 = src[e + 3];
            const int t0 = type[e], t1 = type[e + 1], t2 = type[e + 2], t3 = type[e + 3];
            const float v0 = rows[(int64_t)s0 * ld_rows + lane], v1 = rows[(int64_t)s1 * ld_rows + lane];
            const float v2 = rows[(int64_t)s2 * ld_rows + lane], v3 = rows[(int64_t)s3 * ld_rows + lane];
            acc[t0 * 32 + lane] += v0;
            acc[t1 * 32 + lane] += v1;
            acc[t2 * 32 + lane] += v2;
            acc[t3 * 32 + lane] += v3;
        }
        for (; e < e1; ++e) acc[type[e] * 32 + lane] += rows[(int64_t)src[e] * ld_rows + lane];
        const float rs = root_scale ? root_scale[j] : 1.0f;
        acc[n_types * 32 + lane] = own[j * ld_own + lane] * rs;
        float *o = out + j * ld_out;
        for (int t = 0; t <= n_types; ++t) o[t * 32 + lane] = acc[t * 32 + lane];
    }
}

// deg[j] = max(rowptr[j + 1] - rowptr[j], 1) and its reciprocal
__global__ void degree_kernel(const int32_t *__restrict__ rowptr, int64_t n, float *__restrict__ deg,
                              float *__restrict__ inv_deg) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int d = rowptr[j + 1] - rowptr[j];
    const float f = (float)(d > 1 ? d : 1);
    deg[j] = f;
    inv_deg[j] = 1.0f / f;
}

// ------------------------------------------------------------------------------------------------ loss backward
constexpr float kLossEpsB = 1e-7f;          // losses.py:10
struct LossCoef {
    double a, b, c;                        // the three factors of the product (losses.py:104-106)
};
__device__ __forceinline__ LossCoef loss_factors(const double *terms, float wc, float wl, float wa) {
    LossCoef k;
    k.a = 1.0 - (double)wa * terms[0];
    k.b = 1.0 - (double)wc * terms[1];
    k.c = 1.0 - (double)wl * terms[2];
    return k;
}

__global__ __launch_bounds__(256) void loss_bwd_edges_kernel(const float *__restrict__ p, int64_t ldp,
                                                             const int64_t *__restrict__ col, int64_t ec,
                                                             const int64_t *__restrict__ adj, int64_t ea,
                                                             const float *__restrict__ len, int64_t ldl,
                                                             const double *__restrict__ terms, float wc, float wl,
                                                             float wa, double *__restrict__ acc, int64_t n) {
    const LossCoef k = loss_factors(terms, wc, wl, wa);
    const double coef_c = ec > 0 ? -(double)wc * k.a * k.c / (double)ec : 0.0;                   // d loss / d (sum log(1 - pp))
    const double coef_l = ea > 0 ? -(double)wl * k.a * k.b / ((double)ea * 2.302585092994046) : 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = t0; e < ec; e += stride) {
        const int64_t i = col[e], j = col[ec + e];
        if (i < 0 || i >= n || j < 0 || j >= n) continue;       // (the forward reports these: NaN loss -> IndexError)
        const float pi = p[i * ldp], pj = p[j * ldp];
        const float pp = pi * pj;
        if (pp >= kLossEpsB && pp <= 1.0f - kLossEpsB) {         // torch.clamp passes the gradient inside [min, max]
            const double g = -coef_c / (double)(1.0f - pp);
            atomicAdd(acc + i, g * (double)pj);
            atomicAdd(acc + j, g * (double)pi);
        }
    }
    for (int64_t e = t0; e < ea; e += stride) {
        const int64_t i = adj[e], j = adj[ea + e];
        if (i < 0 || i >= n || j < 0 || j >= n) continue;
        const float pi = p[i * ldp], pj = p[j * ldp];
        const float pp = pi * pj * len[e * ldl];
        if (pp >= kLossEpsB) {                                  // d log(pi pj len) / d pi = 1 / pi
            atomicAdd(acc + i, coef_l / (double)pi);
            atomicAdd(acc + j, coef_l / (double)pj);
        }
    }
}

__global__ void loss_bwd_final_kernel(const double *__restrict__ acc, const float *__restrict__ area, int64_t lda,
                                      int64_t n, const double *__restrict__ terms, float wc, float wl, float wa,
                                      const float *__restrict__ grad_out, float *__restrict__ dp, int64_t ld_dp) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const LossCoef k = loss_factors(terms, wc, wl, wa);
    // l_area = log(max(m, eps)), m = mean(area p): d l_area / d p_v = area_v / (n m) while m >= eps
    const double m = exp(terms[0]);
    double g = acc[v];
    if (m > (double)kLossEpsB * (1.0 + 1e-9)) g += -(double)wa * k.b * k.c * (double)area[v * lda] / ((double)n * m);
    dp[v * ld_dp] = (float)(g * (double)(grad_out ? grad_out[0] : 1.0f));
}

}  // namespace tgnn

using namespace tgnn;

// =================================================================================================== C ABI
extern "C" {

int tgnn_transpose(const float *w, int32_t rows, int32_t cols, float *out, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(rows >= 0 && cols >= 0, "shape");
    if ((int64_t)rows * cols == 0) return TGNN_OK;
    TGNN_CHECK_ARG(w && out, "null pointer");
    const int64_t total = (int64_t)rows * cols;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, rows,
                       cols, out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int tgnn_swap_leading(const float *in, int32_t da, int32_t db, int32_t dc, float *out, int32_t out_da,
                      tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(da >= 0 && db >= 0 && dc >= 0 && out_da >= da, "shape");
    const int64_t total = (int64_t)da * db * dc;
    if (total == 0) return TGNN_OK;
    TGNN_CHECK_ARG(in && out, "null pointer");
    hipLaunchKernelGGL(swap_leading_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, da,
                       db, dc, out, out_da);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int tgnn_sigmoid_bwd(const float *d, int64_t ld_d, const float *t, int64_t ld_t, int64_t n_rows, int32_t c, float *out,
                     int64_t ld_o, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 0 && c >= 1, "shape");
    if (n_rows == 0) return TGNN_OK;
    TGNN_CHECK_ARG(d && t && out, "null pointer");
    const int64_t total = n_rows * c;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, ld_d,
                       t, ld_t, n_rows, c, out, ld_o);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int tgnn_add_into(const float *src, int64_t ld_s, int64_t n_rows, int32_t c, float *dst, int64_t ld_d,
                  tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 0 && c >= 1, "shape");
    if (n_rows == 0) return TGNN_OK;
    TGNN_CHECK_ARG(src && dst, "null pointer");
    const int64_t total = n_rows * c;
    hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_s,
                       n_rows, c, dst, ld_d);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

size_t tgnn_reduce_workspace_bytes(int32_t width) {
    return align_up((size_t)kMaxRedPartials * 6 * (size_t)(width > 32 ? width : 32) * sizeof(double), 256);
}

int tgnn_colsum(const float *x, int64_t ld, int64_t n_rows, int32_t c, float *out, void *ws, size_t ws_bytes,
                tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 0 && c >= 1 && ld >= c, "shape");
    TGNN_CHECK_ARG(out && (x || n_rows == 0), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) {
        TGNN_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * c, s));
        return TGNN_OK;
    }
    const int parts = red_partials(n_rows);
    TGNN_CHECK_ARG(ws && ws_bytes >= (size_t)parts * c * sizeof(double), "workspace");
    const int64_t rpb = (n_rows + parts - 1) / parts;
    double *partial = static_cast<double *>(ws);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(parts, (c + 63) / 64), dim3(kRedThreads), 0, s, x, ld, n_rows, c, rpb,
                       partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((c + kFinCols - 1) / kFinCols), dim3(256), 0, s, partial, parts, c, out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* BatchNorm1d backward (train mode), fused with the derivative of the activation in front of it.
 * coef: device float [2][F] scratch that tgnn_bn_bwd_apply reads. */
int tgnn_bn_bwd_reduce(const float *dy, int64_t ld_dy, const float *a, int64_t ld_a, const float *stat, int64_t n_rows,
                       int32_t f, float eps, float *coef, float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                       tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 1 && f >= 1, "shape");
    TGNN_CHECK_ARG(dy && a && stat && coef, "null pointer");
    const int parts = red_partials(n_rows);
    TGNN_CHECK_ARG(ws && ws_bytes >= (size_t)parts * 3 * f * sizeof(double), "workspace");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rpb = (n_rows + parts - 1) / parts;
    double *partial = static_cast<double *>(ws);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(parts, (f + 63) / 64), dim3(kRedThreads), 0, s, dy, ld_dy, a, ld_a, stat,
                       n_rows, f, rpb, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((f + kFinCols - 1) / kFinCols), dim3(256), 0, s, partial, (int64_t)3 * f,
                       (int64_t)0, parts, f, n_rows, eps, coef, dgamma, dbeta);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int tgnn_bn_bwd_apply(const float *dy, int64_t ld_dy, const float *a, int64_t ld_a, const float *stat, const float *coef,
                      int64_t n_rows, int32_t f, int32_t act, float *dz, int64_t ld_dz, const float *row_scale,
                      float *scaled, int64_t ld_scaled, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 0 && f >= 1, "shape");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    if (n_rows == 0) return TGNN_OK;
    TGNN_CHECK_ARG(dy && a && stat && coef && dz, "null pointer");
    TGNN_CHECK_ARG(!scaled || row_scale, "row_scale");
    const int64_t total = n_rows * f;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       ld_dy, a, ld_a, stat, coef, n_rows, f, act, dz, ld_dz, row_scale, scaled, ld_scaled);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* Backward of the branch merge + the reductions of both BatchNorms behind it (width 32).
 * coef1 / coef2: float [2][32] each; dgamma / dbeta per branch. */
int tgnn_merge_bwd_reduce(const float *dh, int64_t ld_dh, const float *a1, const float *stat1, const float *a2,
                          const float *stat2, const float *carry, int64_t n_rows, int32_t c, float eps1, float eps2,
                          float *dy1, float *dy2, float *resid_grad, int64_t ld_resid, float *coef1, float *dgamma1,
                          float *dbeta1, float *coef2, float *dgamma2, float *dbeta2, void *ws, size_t ws_bytes,
                          tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 1, "shape");
    if (c != 32) {
        set_error("tgnn_merge_bwd_reduce: width 32 only");
        return TGNN_ERR_UNSUPPORTED;
    }
    TGNN_CHECK_ARG(dh && a1 && a2 && stat1 && stat2 && dy1 && dy2 && coef1 && coef2, "null pointer");
    TGNN_CHECK_ARG(ld_dh % 4 == 0 && (!resid_grad || ld_resid % 4 == 0), "row strides must be multiples of 4");
    const int parts = red_partials(n_rows);
    TGNN_CHECK_ARG(ws && ws_bytes >= (size_t)parts * 6 * 32 * sizeof(double), "workspace");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rpb = ((n_rows + parts - 1) / parts + 31) / 32 * 32;
    const int used = (int)((n_rows + rpb - 1) / rpb);
    double *partial = static_cast<double *>(ws);
    hipLaunchKernelGGL(merge_bwd_reduce_kernel, dim3(used), dim3(256), 0, s, dh, ld_dh, a1, stat1, a2, stat2, carry, n_rows,
                       rpb, dy1, dy2, resid_grad, ld_resid, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(32 / kFinCols), dim3(256), 0, s, partial, (int64_t)6 * 32, (int64_t)0, used,
                       32, n_rows, eps1, coef1, dgamma1, dbeta1);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(32 / kFinCols), dim3(256), 0, s, partial, (int64_t)6 * 32, (int64_t)3 * 32,
                       used, 32, n_rows, eps2, coef2, dgamma2, dbeta2);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

size_t tgnn_wgrad_workspace_bytes(int64_t n_rows, int32_t cout, int32_t cin) {
    int tm, tn, parts;
    int64_t rpb;
    wgrad_plan(n_rows > 0 ? n_rows : 1, cout, cin, tm, tn, parts, rpb);
    return align_up((size_t)parts * ((size_t)cout * cin + cout) * sizeof(float), 256);
}

/* out [cout, cin] (row-major: torch's Linear.weight layout) = dz^T . x over n_rows rows;
 * dbias [cout] (may be NULL) = column sums of dz, from the same pass. */
int tgnn_wgrad(const float *dz, int64_t ld_dz, const float *x, int64_t ld_x, int64_t x_kblock_stride, int64_t n_rows,
               int32_t cout, int32_t cin, float *out, float *dbias, void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 0 && cout >= 1 && cin >= 1, "shape");
    TGNN_CHECK_ARG(out, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) {
        TGNN_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * cout * cin, s));
        if (dbias) TGNN_CHECK_HIP(hipMemsetAsync(dbias, 0, sizeof(float) * cout, s));
        return TGNN_OK;
    }
    TGNN_CHECK_ARG(dz && x, "null pointer");
    int tm, tn, parts;
    int64_t rpb;
    wgrad_plan(n_rows, cout, cin, tm, tn, parts, rpb);
    const int64_t w_elems = (int64_t)cout * cin, elems = w_elems + (dbias ? cout : 0);
    const bool direct = parts == 1;            // one row range (small layouts): no partial tiles, no second launch
    TGNN_CHECK_ARG(direct || (ws && ws_bytes >= (size_t)parts * elems * sizeof(float)), "workspace");
    float *partial = direct ? out : static_cast<float *>(ws);
    const dim3 grid(parts, (cout + 32 * tm - 1) / (32 * tm), (cin + 32 * tn - 1) / (32 * tn));
#define TGNN_WGRAD(TM_, TN_)                                                                                         \
    hipLaunchKernelGGL((wgrad_kernel<TM_, TN_>), grid, dim3(256), 0, s, dz, ld_dz, x, ld_x, x_kblock_stride, n_rows, cout, \
                       cin, rpb, partial, elems, dbias ? 1 : 0, direct ? dbias : nullptr)
    if (tm == 2 && tn == 2) TGNN_WGRAD(2, 2);
    else if (tm == 2) TGNN_WGRAD(2, 1);
    else if (tn == 2) TGNN_WGRAD(1, 2);
    else TGNN_WGRAD(1, 1);
#undef TGNN_WGRAD
    if (!direct)
        hipLaunchKernelGGL(wgrad_final_kernel, dim3((unsigned)((elems + 15) / 16)), dim3(256), 0, s, partial, parts, elems,
                           w_elems, out, dbias);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* Backward of a 3-layer sigmoid MLP without BatchNorm (GraphConv's edge MLP edge_conv.py:17-18, GINConv's MLP
 * coll_conv.py:14-18: Linear -> Sigmoid three times) in ONE call: re-derives the hidden activations t1, t2 from the input,
 * then per layer  dpre = d * t (1 - t),  dW = dpre^T . in (+ bias gradient),  d_in = dpre . W.  The same kernels the
 * per-op schedule launches, sequenced here instead of from Python (17 library calls -> 1).
 * x [n, d0] dense; w_k [d_k, d_{k-1}]; t3 = the MLP's output [n, d3] dense; d_out: gradient at t3; dx may be NULL. */
size_t tgnn_sigmoid_mlp_bwd_workspace_bytes(int64_t n_rows, int32_t d0, int32_t d1, int32_t d2, int32_t d3) {
    const int64_t n = n_rows > 0 ? n_rows : 1;
    const int dmax = d1 > d2 ? (d1 > d3 ? d1 : d3) : (d2 > d3 ? d2 : d3);
    const size_t wt = align_up((size_t)d3 * d2 * 4, 256) + align_up((size_t)d2 * d1 * 4, 256) + align_up((size_t)d1 * d0 * 4, 256);
    size_t wg = tgnn_wgrad_workspace_bytes(n, d3, d2);
    if (tgnn_wgrad_workspace_bytes(n, d2, d1) > wg) wg = tgnn_wgrad_workspace_bytes(n, d2, d1);
    if (tgnn_wgrad_workspace_bytes(n, d1, d0) > wg) wg = tgnn_wgrad_workspace_bytes(n, d1, d0);
    const int zmax = d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2);
    return align_up((size_t)n * d1 * 4, 256) + align_up((size_t)n * d2 * 4, 256) + 2 * align_up((size_t)n * dmax * 4, 256) +
           wt + align_up((size_t)zmax * 4, 256) + wg + 256;
}

int tgnn_sigmoid_mlp_bwd(const float *x, int64_t n_rows, int32_t d0, int32_t d1, int32_t d2, int32_t d3, const float *w1,
                         const float *b1, const float *w2, const float *b2, const float *w3, const float *t3,
                         const float *d_out, int64_t ld_dout, float *dw1, float *db1, float *dw2, float *db2, float *dw3,
                         float *db3, float *dx, void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 0 && d0 >= 1 && d1 >= 1 && d2 >= 1 && d3 >= 1, "shape");
    TGNN_CHECK_ARG(w1 && b1 && w2 && b2 && w3 && dw1 && db1 && dw2 && db2 && dw3 && db3, "null pointer");
    TGNN_CHECK_ARG(n_rows == 0 || (x && t3 && d_out), "null pointer");
    TGNN_CHECK_ARG(ws && ws_bytes >= tgnn_sigmoid_mlp_bwd_workspace_bytes(n_rows, d0, d1, d2, d3), "workspace");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = n_rows;
    const int dmax = d1 > d2 ? (d1 > d3 ? d1 : d3) : (d2 > d3 ? d2 : d3);
    const int zmax = d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2);
    Carver cv(ws, ws_bytes);
    const int64_t nn = n > 0 ? n : 1;
    float *t1 = cv.take<float>((size_t)nn * d1), *t2 = cv.take<float>((size_t)nn * d2);
    float *buf_a = cv.take<float>((size_t)nn * dmax), *buf_b = cv.take<float>((size_t)nn * dmax);
    float *wt3 = cv.take<float>((size_t)d3 * d2), *wt2 = cv.take<float>((size_t)d2 * d1), *wt1 = cv.take<float>((size_t)d1 * d0);
    float *zero = cv.take<float>((size_t)zmax);
    void *wg = cv.take<unsigned char>(1);
    const size_t wg_bytes = ws_bytes - (size_t)((unsigned char *)wg - (unsigned char *)ws);
    {   // W3^T, W2^T (and W1^T when dx is wanted) + the zero bias of the dx products: one launch
        TransposeJobs tj{};
        tj.src[0] = w3; tj.dst[0] = wt3; tj.rows[0] = d3; tj.cols[0] = d2;
        tj.src[1] = w2; tj.dst[1] = wt2; tj.rows[1] = d2; tj.cols[1] = d1;
        tj.src[2] = w1; tj.dst[2] = wt1; tj.rows[2] = d1; tj.cols[2] = d0;
        tj.n = dx ? 3 : 2;
        tj.zero = zero;
        tj.n_zero = zmax;
        size_t biggest = (size_t)d3 * d2;
        if ((size_t)d2 * d1 > biggest) biggest = (size_t)d2 * d1;
        if (dx && (size_t)d1 * d0 > biggest) biggest = (size_t)d1 * d0;
        unsigned gx = (unsigned)((biggest + 255) / 256);
        if (gx > 64) gx = 64;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(transpose_multi_kernel, dim3(gx, tj.n + 1), dim3(256), 0, s, tj);
    }
#define TGNN_TRY_(expr)            \
    do {                           \
        const int rc__ = (expr);   \
        if (rc__ != TGNN_OK) return rc__; \
    } while (0)
    if (n > 0) {
        TGNN_TRY_(tgnn_dense_act_fwd(x, d0, 32, nullptr, w1, b1, n, d0, d1, TGNN_ACT_SIGMOID, t1, d1, nullptr, nullptr, stream));
        TGNN_TRY_(tgnn_dense_act_fwd(t1, d1, 32, nullptr, w2, b2, n, d1, d2, TGNN_ACT_SIGMOID, t2, d2, nullptr, nullptr, stream));
    }
    // layer 3
    TGNN_TRY_(tgnn_sigmoid_bwd(d_out, ld_dout, t3, d3, n, d3, buf_a, d3, stream));
    TGNN_TRY_(tgnn_wgrad(buf_a, d3, t2, d2, 0, n, d3, d2, dw3, db3, wg, wg_bytes, stream));
    if (n > 0) {
        TGNN_TRY_(tgnn_dense_act_fwd(buf_a, d3, 32, nullptr, wt3, zero, n, d3, d2, TGNN_ACT_NONE, buf_b, d2, nullptr, nullptr, stream));
    }
    // layer 2
    TGNN_TRY_(tgnn_sigmoid_bwd(buf_b, d2, t2, d2, n, d2, buf_a, d2, stream));
    TGNN_TRY_(tgnn_wgrad(buf_a, d2, t1, d1, 0, n, d2, d1, dw2, db2, wg, wg_bytes, stream));
    if (n > 0) {
        TGNN_TRY_(tgnn_dense_act_fwd(buf_a, d2, 32, nullptr, wt2, zero, n, d2, d1, TGNN_ACT_NONE, buf_b, d1, nullptr, nullptr, stream));
    }
    // layer 1
    TGNN_TRY_(tgnn_sigmoid_bwd(buf_b, d1, t1, d1, n, d1, buf_a, d1, stream));
    TGNN_TRY_(tgnn_wgrad(buf_a, d1, x, d0, 0, n, d1, d0, dw1, db1, wg, wg_bytes, stream));
    if (dx && n > 0) {
        TGNN_TRY_(tgnn_dense_act_fwd(buf_a, d1, 32, nullptr, wt1, zero, n, d1, d0, TGNN_ACT_NONE, dx, d0, nullptr, nullptr, stream));
    }
#undef TGNN_TRY_
    return TGNN_OK;
}

/* Per-type sums of gathered rows + the (scaled) own row: out [n_nodes][(n_types + 1) * 32].  rowptr / src / type:
 * a CSR of tgnn_csr_build with the type of every slot. */
int tgnn_nnconv_type_sum(const float *rows, int64_t ld_rows, const float *own, int64_t ld_own, const float *root_scale,
                         const int32_t *rowptr, const int32_t *src, const int32_t *type, int64_t n_nodes, int32_t n_types,
                         int32_t c, float *out, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0 && n_types >= 0, "shape");
    if (c != 32 || n_types > 63) {
        set_error("tgnn_nnconv_type_sum: width 32 and at most 63 edge types");
        return TGNN_ERR_UNSUPPORTED;
    }
    if (n_nodes == 0) return TGNN_OK;
    TGNN_CHECK_ARG(rows && own && rowptr && out && (src || n_types == 0), "null pointer");
    const size_t lds = (size_t)kTsRows * (n_types + 1) * 32 * sizeof(float);
    int64_t blocks = (n_nodes + kTsRows - 1) / kTsRows;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nnconv_type_sum_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, rows, ld_rows, own,
                       ld_own, root_scale, rowptr, src, type, n_nodes, n_types, out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int tgnn_csr_degree(const int32_t *rowptr, int64_t n_nodes, float *deg, float *inv_deg, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0, "shape");
    if (n_nodes == 0) return TGNN_OK;
    TGNN_CHECK_ARG(rowptr && deg && inv_deg, "null pointer");
    hipLaunchKernelGGL(degree_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr,
                       n_nodes, deg, inv_deg);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* d loss / d probs[:, map] of tgnn_unsupervised_loss for ONE probability map (the arg-min map: the reference
 * back-propagates through torch.min, losses.py:108).  probs points at that map's column; terms: the three doubles
 * tgnn_unsupervised_loss wrote for it; grad_out (device float, may be NULL = 1): d objective / d loss.
 * ws: n_nodes doubles. */
int tgnn_unsupervised_loss_bwd(const float *probs, int64_t ld_probs, const float *area_ratio, int64_t ld_area,
                               int64_t n_nodes, const int64_t *col_edge_index, int64_t n_col_edges,
                               const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_len, int64_t ld_len,
                               float collision_weight, float align_length_weight, float avg_area_weight,
                               const double *terms, const float *grad_out, float *dprobs, int64_t ld_dprobs, void *ws,
                               size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_col_edges >= 0 && n_adj_edges >= 0, "shape");
    TGNN_CHECK_ARG(probs && area_ratio && terms && dprobs, "null pointer");
    TGNN_CHECK_ARG((n_col_edges == 0 || col_edge_index) && (n_adj_edges == 0 || (adj_edge_index && adj_edge_len)),
                   "null edge pointer");
    TGNN_CHECK_ARG(ws && ws_bytes >= (size_t)n_nodes * sizeof(double), "workspace");
    hipStream_t s = (hipStream_t)stream;
    double *acc = static_cast<double *>(ws);
    TGNN_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * n_nodes, s));
    const int64_t work = n_col_edges > n_adj_edges ? n_col_edges : n_adj_edges;
    if (work > 0) {
        int64_t blocks = (work + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(loss_bwd_edges_kernel, dim3((unsigned)blocks), dim3(256), 0, s, probs, ld_probs, col_edge_index,
                           n_col_edges, adj_edge_index, n_adj_edges, adj_edge_len, ld_len, terms, collision_weight,
                           align_length_weight, avg_area_weight, acc, n_nodes);
    }
    hipLaunchKernelGGL(loss_bwd_final_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, s, acc, area_ratio,
                       ld_area, n_nodes, terms, collision_weight, align_length_weight, avg_area_weight, grad_out, dprobs,
                       ld_dprobs);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

}  // extern "C"
