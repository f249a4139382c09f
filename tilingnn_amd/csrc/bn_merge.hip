// Train-mode BatchNorm1d statistics, BN apply, branch merge and row gather/scatter on gfx950.
//
// Reference: nn.BatchNorm1d instances inside Linear_trans (layers/util.py:28,35-36), GraphConv
// (edge_conv.py:22,28-29) and CollConv (coll_conv.py:22,28-29) -- 46 per forward, ALL in train
// mode because ML_Solver.load_saved_network ends with network.train() (ml_solver.py:129-131):
// batch statistics over all N rows, running stats updated with momentum 0.1 / unbiased variance.
// Branch merge: TilinGNN.forward, graph_networks/networks/TilinGNN.py:64-71.
//
// The producing kernels emit per-block fp64 column sums; tgnn_bn_finalize turns them into the
// 4-row stat record consumed while loading (mean split hi/lo so that v - mean stays exact for the
// near-constant columns the collision branch produces).  Deterministic: fixed reduction tree.
#include "tgnn_common.h"

namespace tgnn {

// mode 0: partials->stat, 1: partials->sums, 2: sums->stat, 3: running stats->stat (eval mode)
// One block per (job, SLICE of 32 features): the block reduces the 64 column entries of its slice (32 sums, 32 sums of squares)
// over all partial rows -- 16 row groups of the rows in order, every thread one batch of independent loads, then the fixed 16-way
// fold.  For F = 32 that is the whole job (and the tree merge_bn1 / the shard kernels / the GIN kernel's last block repeat:
// the same bits); a wide BatchNorm (the final MLP's 256 / 128 / 64 features, 512 partial rows of 4 KB) is spread over F / 32
// blocks instead of one CU reading 2 MB by itself (14 -> 5 us per launch, six launches on the forward's serial tail).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(BnJobs jobs, int mode, int f, int slices, int64_t n_total, float eps,
                                                           float momentum) {
    __shared__ double red[1024];
    __shared__ double tot[64];
    const BnJob jb = jobs.job[blockIdx.x / slices];
    const int slice = blockIdx.x % slices, f0 = slice * 32, fw = f - f0 < 32 ? f - f0 : 32;
    const int tid = threadIdx.x, two_f = 2 * f;
    // the affine parameters and running buffers are fetched up front: their round trip then overlaps the partial
    // rows' instead of following it (this kernel is pure latency)
    float pre_gamma = 1.f, pre_beta = 0.f, pre_rm = 0.f, pre_rv = 1.f;
    if (tid < fw && mode != 1) {
        pre_gamma = jb.gamma[f0 + tid];
        pre_beta = jb.beta[f0 + tid];
        if (jb.running_mean) {
            pre_rm = jb.running_mean[f0 + tid];
            pre_rv = jb.running_var[f0 + tid];
        }
    }
    if (mode == 3) {
        if (tid < fw) {
            const double mean = (double)pre_rm;
            const double var = (double)pre_rv;
            const float mh = (float)mean;
            jb.stat[f0 + tid] = mh;
            jb.stat[f + f0 + tid] = (float)(mean - (double)mh);
            jb.stat[2 * f + f0 + tid] = (float)((double)pre_gamma / sqrt(var + (double)eps));
            jb.stat[3 * f + f0 + tid] = pre_beta;
        }
        return;
    }
    // column entry of thread j < 64 of the slice: j < 32 the sum of feature f0 + j, else the sum of squares of feature f0 + j - 32
    const int j = tid & 63, g = tid >> 6;
    const bool live = (j & 31) < fw;
    const int col = (j >> 5) * f + f0 + (j & 31);
    if (mode == 2) {
        if (tid < 64) tot[tid] = live ? jb.sums[col] : 0.0;
    } else {
        constexpr int groups = 16;
        double acc = 0.0;
        if (live) {
            // all of a thread's partial rows are fetched before the first add: with <= 512 partial rows and 16 row groups
            // that is one batch of <= 32 independent loads, i.e. ONE L2 round trip instead of one per row
            const double *src = jb.partials + col;
            int p = g;
            for (; p + 31 * groups < jb.n_partials; p += 32 * groups) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = src[(int64_t)(p + u * groups) * two_f];
#pragma unroll
                for (int u = 0; u < 32; ++u) acc += v[u];
            }
            for (; p + 7 * groups < jb.n_partials; p += 8 * groups) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(p + u * groups) * two_f];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; p < jb.n_partials; p += groups) acc += src[(int64_t)p * two_f];
        }
        red[tid] = acc;
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
            for (int gg = 0; gg < groups; ++gg) t += red[gg * 64 + tid];
            tot[tid] = t;
            if (mode == 1 && live) jb.sums[col] = t;
        }
        if (mode == 1) return;
    }
    __syncthreads();
    if (tid < fw) {
        const double inv_n = 1.0 / (double)n_total;
        const double mean = tot[tid] * inv_n;
        double var = tot[32 + tid] * inv_n - mean * mean;    // biased; fp64 sums of fp32 data
        if (var < 0.0) var = 0.0;
        const float mh = (float)mean;
        jb.stat[f0 + tid] = mh;
        jb.stat[f + f0 + tid] = (float)(mean - (double)mh);
        jb.stat[2 * f + f0 + tid] = (float)((double)pre_gamma / sqrt(var + (double)eps));
        jb.stat[3 * f + f0 + tid] = pre_beta;
        if (jb.running_mean) {
            const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
            jb.running_mean[f0 + tid] = (float)((1.0 - (double)momentum) * (double)pre_rm + (double)momentum * mean);
            jb.running_var[f0 + tid] = (float)((1.0 - (double)momentum) * (double)pre_rv + (double)momentum * unbiased);
        }
    }
    if (tid == 0 && slice == 0 && jb.num_batches_tracked) *jb.num_batches_tracked += 1;
}

// absmax_out (optional, here and in the merge kernels): the largest |out| as float bits, by atomicMax into a word the caller
// zeroed -- the bound the fp16-pair kernels scale their operand with (tgnn_common.h: split2_f16)
__global__ void bn_apply_kernel(const float *__restrict__ v, int64_t ldv, const float *__restrict__ stat, int64_t n,
                                int f, float *__restrict__ out, int64_t ldo, unsigned *__restrict__ absmax_out) {
    const int64_t total = n * f;
    float am = 0.f;
    if (f == 32 && ldv == 32 && ldo == 32 && (((uintptr_t)v | (uintptr_t)out | (uintptr_t)stat) & 15) == 0) {
        // packed rows of 32 (the init MLP's output -> middle[0]): float4 per thread, the record's four float4 in registers
        const int q = threadIdx.x & 7;                        // (the grid stride is a multiple of 8 float4)
        const float4 mh = reinterpret_cast<const float4 *>(stat)[q], ml = reinterpret_cast<const float4 *>(stat + 32)[q];
        const float4 gi = reinterpret_cast<const float4 *>(stat + 64)[q], be = reinterpret_cast<const float4 *>(stat + 96)[q];
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total / 4; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 x = reinterpret_cast<const float4 *>(v)[i];
            float4 o;
            o.x = bn_apply1(x.x, mh.x, ml.x, gi.x, be.x);
            o.y = bn_apply1(x.y, mh.y, ml.y, gi.y, be.y);
            o.z = bn_apply1(x.z, mh.z, ml.z, gi.z, be.z);
            o.w = bn_apply1(x.w, mh.w, ml.w, gi.w, be.w);
            reinterpret_cast<float4 *>(out)[i] = o;
            am = absmax4(am, o);
        }
        absmax_flush(am, absmax_out);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / f;
        const int c = (int)(i - r * f);
        const float o = bn_apply1(v[r * ldv + c], stat[c], stat[f + c], stat[2 * f + c], stat[3 * f + c]);
        out[r * ldo + c] = o;
        am = fmaxf(am, fabsf(o));
    }
    absmax_flush(am, absmax_out);
}

// out = BN1(a1) * BN2(a2) (+ resid); h2_out = BN2(a2) (optional).  float4 per thread, C % 4 == 0.
__global__ __launch_bounds__(256) void merge_kernel(const float *__restrict__ a1, const float *__restrict__ st1,
                                                    const float *__restrict__ a2, const float *__restrict__ st2,
                                                    const float *__restrict__ resid, int64_t n4, int c,
                                                    float *__restrict__ out, float *__restrict__ h2_out,
                                                    unsigned *__restrict__ absmax_out) {
    float am = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)((i * 4) % c);
        const float4 x1 = reinterpret_cast<const float4 *>(a1)[i];
        const float4 x2 = reinterpret_cast<const float4 *>(a2)[i];
        const float4 m1h = *reinterpret_cast<const float4 *>(st1 + col), m1l = *reinterpret_cast<const float4 *>(st1 + c + col);
        const float4 g1 = *reinterpret_cast<const float4 *>(st1 + 2 * c + col), b1 = *reinterpret_cast<const float4 *>(st1 + 3 * c + col);
        const float4 m2h = *reinterpret_cast<const float4 *>(st2 + col), m2l = *reinterpret_cast<const float4 *>(st2 + c + col);
        const float4 g2 = *reinterpret_cast<const float4 *>(st2 + 2 * c + col), b2 = *reinterpret_cast<const float4 *>(st2 + 3 * c + col);
        float4 y2, o;
        y2.x = bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x);
        y2.y = bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y);
        y2.z = bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z);
        y2.w = bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w);
        o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * y2.x;
        o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * y2.y;
        o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * y2.z;
        o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * y2.w;
        if (resid) {
            const float4 r = reinterpret_cast<const float4 *>(resid)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4 *>(out)[i] = o;
        am = absmax4(am, o);
        if (h2_out) reinterpret_cast<float4 *>(h2_out)[i] = y2;
    }
    absmax_flush(am, absmax_out);
}

// merge with the FIRST BatchNorm's statistics taken straight from the producer's partial rows: every block repeats
// the (small) fixed-tree reduction of bn_finalize_kernel -- same grouping, same order, hence the same bits -- instead of
// waiting for a separate 1-block finalize launch in the middle of the critical NNConv -> finalize -> merge chain.
// Block 0 also updates the running buffers.  C = 32 only (2 F = 64 columns, 16 row groups as bn_finalize_kernel).
__global__ __launch_bounds__(256) void merge_bn1_kernel(const float *__restrict__ a1, BnJob j1, int64_t n_total, float eps,
                                                        float momentum, const float *__restrict__ a2,
                                                        const float *__restrict__ st2, const float *__restrict__ resid,
                                                        int64_t n4, float *__restrict__ out, unsigned *__restrict__ absmax_out) {
    constexpr int c = 32, two_f = 64, groups = 16;
    __shared__ double red[groups * two_f];
    __shared__ double tot[two_f];
    __shared__ __attribute__((aligned(16))) float st1[4 * c];
    const int tid = threadIdx.x;
    float pre_gamma = 1.f, pre_beta = 0.f, pre_rm = 0.f, pre_rv = 1.f;
    if (tid < c) {
        pre_gamma = j1.gamma[tid];
        pre_beta = j1.beta[tid];
        if (blockIdx.x == 0 && j1.running_mean) {
            pre_rm = j1.running_mean[tid];
            pre_rv = j1.running_var[tid];
        }
    }
    {
        const int j = tid & 63, g0 = (tid >> 6) * 4;          // this thread: row groups g0 .. g0 + 3 of column j
        const double *src = j1.partials + j;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u;
            double acc = 0.0;
            int p = g;
            for (; p + 31 * groups < j1.n_partials; p += 32 * groups) {
                double v[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) v[q] = src[(int64_t)(p + q * groups) * two_f];
#pragma unroll
                for (int q = 0; q < 32; ++q) acc += v[q];
            }
            for (; p + 7 * groups < j1.n_partials; p += 8 * groups) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(p + q * groups) * two_f];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q];
            }
            for (; p < j1.n_partials; p += groups) acc += src[(int64_t)p * two_f];
            red[g * two_f + j] = acc;
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
        st1[tid] = mh;
        st1[c + tid] = (float)(mean - (double)mh);
        st1[2 * c + tid] = (float)((double)pre_gamma / sqrt(var + (double)eps));
        st1[3 * c + tid] = pre_beta;
        if (blockIdx.x == 0) {
            if (j1.stat) {
                j1.stat[tid] = st1[tid]; j1.stat[c + tid] = st1[c + tid];
                j1.stat[2 * c + tid] = st1[2 * c + tid]; j1.stat[3 * c + tid] = st1[3 * c + tid];
            }
            if (j1.running_mean) {
                const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
                j1.running_mean[tid] = (float)((1.0 - (double)momentum) * (double)pre_rm + (double)momentum * mean);
                j1.running_var[tid] = (float)((1.0 - (double)momentum) * (double)pre_rv + (double)momentum * unbiased);
            }
            if (tid == 0 && j1.num_batches_tracked) *j1.num_batches_tracked += 1;
        }
    }
    __syncthreads();
    float am = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)((i * 4) % c);
        const float4 x1 = reinterpret_cast<const float4 *>(a1)[i];
        const float4 x2 = reinterpret_cast<const float4 *>(a2)[i];
        const float4 m1h = *reinterpret_cast<const float4 *>(st1 + col), m1l = *reinterpret_cast<const float4 *>(st1 + c + col);
        const float4 g1 = *reinterpret_cast<const float4 *>(st1 + 2 * c + col), b1 = *reinterpret_cast<const float4 *>(st1 + 3 * c + col);
        const float4 m2h = *reinterpret_cast<const float4 *>(st2 + col), m2l = *reinterpret_cast<const float4 *>(st2 + c + col);
        const float4 g2 = *reinterpret_cast<const float4 *>(st2 + 2 * c + col), b2 = *reinterpret_cast<const float4 *>(st2 + 3 * c + col);
        float4 o;
        o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x);
        o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y);
        o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z);
        o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w);
        if (resid) {
            const float4 r = reinterpret_cast<const float4 *>(resid)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4 *>(out)[i] = o;
        am = absmax4(am, o);
    }
    absmax_flush(am, absmax_out);
}

// The same for layouts whose producer emits many partial rows (one per CU: 224 at 100k nodes): 1024 threads per block
// reduce them exactly as bn_finalize_kernel does (16 row groups x 64 columns, every thread ONE batch of independent loads,
// then the fixed 16-way fold: the same tree, the same bits), so that NNConv -> merge needs no launch in between.  256 blocks
// x 115 KB of partial rows come out of L2; the element-wise part then walks ~3 float4 per thread.
__global__ __launch_bounds__(1024) void merge_bn1_wide_kernel(const float *__restrict__ a1, BnJob j1, int64_t n_total, float eps,
                                                              float momentum, const float *__restrict__ a2,
                                                              const float *__restrict__ st2, const float *__restrict__ resid,
                                                              int64_t n4, float *__restrict__ out, unsigned *__restrict__ absmax_out) {
    constexpr int c = 32, two_f = 64, groups = 16;
    __shared__ double red[groups * two_f];
    __shared__ double tot[two_f];
    __shared__ __attribute__((aligned(16))) float st1[4 * c];
    __shared__ __attribute__((aligned(16))) float st2s[4 * c];
    const int tid = threadIdx.x;
    float pre_gamma = 1.f, pre_beta = 0.f, pre_rm = 0.f, pre_rv = 1.f;
    if (tid < c) {
        pre_gamma = j1.gamma[tid];
        pre_beta = j1.beta[tid];
        if (blockIdx.x == 0 && j1.running_mean) {
            pre_rm = j1.running_mean[tid];
            pre_rv = j1.running_var[tid];
        }
    }
    if (tid >= 1024 - 4 * c) st2s[tid - (1024 - 4 * c)] = st2[tid - (1024 - 4 * c)];
    // [r5] ONE round trip for the partial rows -- every thread asks for its <= 32 rows (g, g + 16, ..) at once through a buffer
    // descriptor that ends with the rows (a row past the end returns 0.0 without touching memory and adds nothing: the same
    // sum, term by term, as bn_finalize_kernel's batches) -- and the first kPf items' operands go out BEHIND them: the wait
    // for the rows leaves those HBM loads in flight, so the reduction (~4 us in front of every block's stream before) now
    // runs under them.  (At 100 000 nodes a thread has 3.05 items in all.)
    constexpr int kPf = 3;
    const int64_t i_first = (int64_t)blockIdx.x * blockDim.x + tid, i_step = (int64_t)gridDim.x * blockDim.x;
    float4 px1[kPf], px2[kPf], pr[kPf];
    {
        using u32x2_ = __attribute__((ext_vector_type(2))) unsigned int;
        const int j = tid % two_f, g = tid / two_f;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(j1.partials), 0, (int)((uint32_t)j1.n_partials * (uint32_t)two_f * 8u), 0x00020000);
        static_assert(TGNN_BN_MAX_PARTIALS <= 32 * groups, "32 rows per thread cover every partial row");
        u32x2_ v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q)
            v[q] = __builtin_amdgcn_raw_buffer_load_b64(prs, ((uint32_t)(g + q * groups) * (uint32_t)two_f + (uint32_t)j) * 8u, 0, 0);
#pragma unroll
        for (int k = 0; k < kPf; ++k) {
            const int64_t i = i_first + k * i_step;
            const int64_t ii = i < n4 ? i : n4 - 1;          // (clamped: nothing is loaded inside a branch)
            px1[k] = reinterpret_cast<const float4 *>(a1)[ii];
            px2[k] = reinterpret_cast<const float4 *>(a2)[ii];
            pr[k] = resid ? reinterpret_cast<const float4 *>(resid)[ii] : float4{0.f, 0.f, 0.f, 0.f};
        }
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) acc += __builtin_bit_cast(double, v[q]);
        red[tid] = acc;
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
        st1[tid] = mh;
        st1[c + tid] = (float)(mean - (double)mh);
        st1[2 * c + tid] = (float)((double)pre_gamma / sqrt(var + (double)eps));
        st1[3 * c + tid] = pre_beta;
        if (blockIdx.x == 0) {
            if (j1.stat) {
                j1.stat[tid] = st1[tid]; j1.stat[c + tid] = st1[c + tid];
                j1.stat[2 * c + tid] = st1[2 * c + tid]; j1.stat[3 * c + tid] = st1[3 * c + tid];
            }
            if (j1.running_mean) {
                const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
                j1.running_mean[tid] = (float)((1.0 - (double)momentum) * (double)pre_rm + (double)momentum * mean);
                j1.running_var[tid] = (float)((1.0 - (double)momentum) * (double)pre_rv + (double)momentum * unbiased);
            }
            if (tid == 0 && j1.num_batches_tracked) *j1.num_batches_tracked += 1;
        }
    }
    __syncthreads();
    const int col = (tid * 4) % c;                            // (the grid stride is a multiple of 32 floats)
    const float4 m1h = *reinterpret_cast<const float4 *>(st1 + col), m1l = *reinterpret_cast<const float4 *>(st1 + c + col);
    const float4 g1 = *reinterpret_cast<const float4 *>(st1 + 2 * c + col), b1 = *reinterpret_cast<const float4 *>(st1 + 3 * c + col);
    const float4 m2h = *reinterpret_cast<const float4 *>(st2s + col), m2l = *reinterpret_cast<const float4 *>(st2s + c + col);
    const float4 g2 = *reinterpret_cast<const float4 *>(st2s + 2 * c + col), b2 = *reinterpret_cast<const float4 *>(st2s + 3 * c + col);
    float am = 0.f;
    auto item = [&](const float4 &x1, const float4 &x2, const float4 &r, int64_t i) {
        // (no contraction of the product with the residual add: with r in a register the compiler would fuse them into one
        //  fma -- one rounding less than merge_kernel / the sharded step's merge, whose add sits behind a branch.  Same bits.)
#pragma clang fp contract(off)
        float4 o;
        o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x);
        o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y);
        o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z);
        o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w);
        if (resid) { o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        reinterpret_cast<float4 *>(out)[i] = o;
        am = absmax4(am, o);
    };
#pragma unroll
    for (int k = 0; k < kPf; ++k) {
        const int64_t i = i_first + k * i_step;
        if (i < n4) item(px1[k], px2[k], pr[k], i);
    }
    for (int64_t i = i_first + kPf * i_step; i < n4; i += i_step) {
        const float4 x1 = reinterpret_cast<const float4 *>(a1)[i];
        const float4 x2 = reinterpret_cast<const float4 *>(a2)[i];
        const float4 r = resid ? reinterpret_cast<const float4 *>(resid)[i] : float4{0.f, 0.f, 0.f, 0.f};
        item(x1, x2, r, i);
    }
    absmax_flush(am, absmax_out);
}

__global__ void rows_gather_kernel(const float *__restrict__ src, int64_t ld, const int *__restrict__ idx, int64_t n_idx,
                                   int c, float *__restrict__ out, int64_t ld_out) {
    const int64_t total = n_idx * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int k = (int)(i - r * c);
        out[r * ld_out + k] = src[(int64_t)idx[r] * ld + k];
    }
}

__global__ void rows_scatter_kernel(const float *__restrict__ in, const int *__restrict__ idx, int64_t n_idx, int c,
                                    float *__restrict__ dst, int64_t ld) {
    const int64_t total = n_idx * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int k = (int)(i - r * c);
        dst[(int64_t)idx[r] * ld + k] = in[i];
    }
}

static inline unsigned ew_grid(int64_t n, int cap = 256 * 8) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------
// Sharded forward: ONE all-to-all per layer carries, to every peer, the raw (pre-BatchNorm) rows of both branches
// that the peer keeps as halo AND this shard's local BatchNorm sums (4 extra rows of 64 floats = 128 doubles:
// [bn1: sum, sumsq][bn2: sum, sumsq] x 32).  Every shard then adds the sums of all shards in rank order (the same
// order everywhere -> identical statistics everywhere, no all-reduce) and merges its own AND its halo rows itself.
// idx >= 0: a row to pack / a halo slot to fill; idx < 0: sum row number -1 - idx.
// ------------------------------------------------------------------------------------------
__global__ void shard_pack_kernel(const float *__restrict__ a1, const float *__restrict__ a2, const int *__restrict__ idx,
                                  int64_t n_rows, const float *__restrict__ sums_as_float, float *__restrict__ out) {
    const int64_t total = n_rows * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 6;
        const int k = (int)(i & 63), id = idx[r];
        float v;
        if (id >= 0) v = k < 32 ? a1[(int64_t)id * 32 + k] : a2[(int64_t)id * 32 + (k - 32)];
        else v = sums_as_float[(-1 - id) * 64 + k];
        out[i] = v;
    }
}
// idx >= 0: halo slot j -> a1[(n_own + j)], a2[(n_own + j)];  idx < 0: code = -1 - idx = 4 * peer + k -> peer_sums
__global__ void shard_unpack_kernel(const float *__restrict__ in, const int *__restrict__ idx, int64_t n_rows,
                                    int64_t n_own, float *__restrict__ a1, float *__restrict__ a2,
                                    float *__restrict__ peer_sums_as_float) {
    const int64_t total = n_rows * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 6;
        const int k = (int)(i & 63), id = idx[r];
        const float v = in[i];
        if (id >= 0) {
            if (k < 32) a1[(n_own + id) * 32 + k] = v;
            else a2[(n_own + id) * 32 + (k - 32)] = v;
        } else {
            peer_sums_as_float[(int64_t)(-1 - id) * 64 + k] = v;
        }
    }
}
// total[j] = sum over the shards, in rank order, of their 128 sums (this shard's own at position `rank`)
__global__ void shard_sum_peers_kernel(const double *__restrict__ peer_sums, const double *__restrict__ own, int world,
                                       int rank, double *__restrict__ total) {
    const int j = threadIdx.x;                             // 128 threads
    double t = 0.0;
    for (int p = 0; p < world; ++p) t += p == rank ? own[j] : peer_sums[(int64_t)p * 128 + j];
    total[j] = t;
}

// ---- sharded forward, one all-to-all per layer: the statistics work of a layer folded into the two kernels that move the
// halo rows anyway.  Unfused the chain behind the convolutions was  finalize(partials -> sums) -> pack -> all-to-all ->
// unpack -> sum over peers -> finalize(sums -> records) -> merge: five 1-block-class launches of ~5 us each on a
// launch-latency-bound chain, 20 times per forward.  Block 0 of the pack kernel now reduces the partial rows itself and
// block 0 of the unpack kernel adds the shards' sums (rank order) and writes the records; the other blocks move rows.
__device__ __forceinline__ void bn_sums_from_partials_1024(const BnJob &jb, int two_f, double *red, double *tot) {
    const int tid = threadIdx.x;
    const int groups = 1024 / two_f;
    const int j = tid % two_f, g = tid / two_f;
    double acc = 0.0;
    if (g < groups) {                                     // same tree as bn_finalize_kernel: the same bits
        const double *src = jb.partials + j;
        int p = g;
        for (; p + 31 * groups < jb.n_partials; p += 32 * groups) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = src[(int64_t)(p + u * groups) * two_f];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
        }
        for (; p + 7 * groups < jb.n_partials; p += 8 * groups) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(p + u * groups) * two_f];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; p < jb.n_partials; p += groups) acc += src[(int64_t)p * two_f];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < two_f) {
        double t = 0.0;
        for (int gg = 0; gg < groups; ++gg) t += red[gg * two_f + tid];
        tot[tid] = t;
    }
    __syncthreads();
}

// jobs.job[0/1].sums: this shard's 2 x 64 sums (128 contiguous doubles), also copied bit for bit into the message's sums
// rows (idx < 0: floats (-1 - idx) * 64 .. of those 128 doubles).  Width 32.
__global__ __launch_bounds__(1024) void shard_pack_sums_kernel(const float *__restrict__ a1, const float *__restrict__ a2,
                                                               const int *__restrict__ idx, int64_t n_rows, BnJobs jobs,
                                                               float *__restrict__ out) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        __shared__ double red[1024];
        __shared__ double tot[128];
        bn_sums_from_partials_1024(jobs.job[0], 64, red, tot);
        bn_sums_from_partials_1024(jobs.job[1], 64, red, tot + 64);
        if (tid < 128) jobs.job[tid >> 6].sums[tid & 63] = tot[tid];
        const float *tf = reinterpret_cast<const float *>(tot);
        for (int64_t r = tid >> 6; r < n_rows; r += 16) {
            const int id = idx[r];
            if (id < 0) out[r * 64 + (tid & 63)] = tf[(-1 - id) * 64 + (tid & 63)];
        }
        return;
    }
    const int64_t total = n_rows * 64;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + tid; i < total; i += (int64_t)(gridDim.x - 1) * 1024) {
        const int64_t r = i >> 6;
        const int k = (int)(i & 63), id = idx[r];
        if (id >= 0) out[i] = k < 32 ? a1[(int64_t)id * 32 + k] : a2[(int64_t)id * 32 + (k - 32)];
    }
}

// idx >= 0: halo slot j -> a1[n_own + j], a2[n_own + j];  idx < 0: code = -1 - idx = 4 * peer + k: row k of that peer's
// sums.  Block 0: total = own + the peers' sums in rank order, then the two records (+ running statistics).
__global__ __launch_bounds__(1024) void shard_unpack_finalize_kernel(const float *__restrict__ in, const int *__restrict__ idx,
                                                                     int64_t n_rows, int64_t n_own, float *__restrict__ a1,
                                                                     float *__restrict__ a2, BnJobs jobs, int world, int rank,
                                                                     int64_t n_total, float eps, float momentum) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        __shared__ int pos[4 * 64];
        __shared__ double tot[128];
        for (int64_t r = tid; r < n_rows; r += 1024) {
            const int id = idx[r];
            if (id < 0 && -1 - id < 4 * 64) pos[-1 - id] = (int)r;
        }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 5, jj = tid & 31;
            const double *own = jobs.job[0].sums;
            double t = 0.0;
            for (int p = 0; p < world; ++p)
                t += p == rank ? own[tid] : reinterpret_cast<const double *>(in + (int64_t)pos[4 * p + k] * 64)[jj];
            tot[tid] = t;
        }
        __syncthreads();
        bn_record_from_sums(jobs.job[0], tot, 32, n_total, eps, momentum);
        bn_record_from_sums(jobs.job[1], tot + 64, 32, n_total, eps, momentum);
        return;
    }
    const int64_t total = n_rows * 64;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + tid; i < total; i += (int64_t)(gridDim.x - 1) * 1024) {
        const int64_t r = i >> 6;
        const int k = (int)(i & 63), id = idx[r];
        if (id >= 0) {
            const float v = in[i];
            if (k < 32) a1[(n_own + id) * 32 + k] = v;
            else a2[(n_own + id) * 32 + (k - 32)] = v;
        }
    }
}

// The same for ONE branch (split exchange: the collision branch's rows and sums travel on the side stream as soon as its GIN
// is through, the adjacency branch's behind the NNConv -- forward.hip): rows of 32 floats, 4 sums rows = 64 doubles per peer.
__global__ __launch_bounds__(1024) void shard_pack1_kernel(const float *__restrict__ a, const int *__restrict__ idx,
                                                           int64_t n_rows, BnJob job, float *__restrict__ out) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        __shared__ double red[1024];
        __shared__ double tot[64];
        bn_sums_from_partials_1024(job, 64, red, tot);
        if (tid < 64) job.sums[tid] = tot[tid];
        const float *tf = reinterpret_cast<const float *>(tot);
        for (int64_t r = tid >> 5; r < n_rows; r += 32) {
            const int id = idx[r];
            if (id < 0) out[r * 32 + (tid & 31)] = tf[(-1 - id) * 32 + (tid & 31)];
        }
        return;
    }
    const int64_t total = n_rows * 32;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + tid; i < total; i += (int64_t)(gridDim.x - 1) * 1024) {
        const int id = idx[i >> 5];
        if (id >= 0) out[i] = a[(int64_t)id * 32 + (i & 31)];
    }
}
__global__ __launch_bounds__(1024) void shard_unpack1_kernel(const float *__restrict__ in, const int *__restrict__ idx,
                                                             int64_t n_rows, int64_t n_own, float *__restrict__ a, BnJob job,
                                                             int world, int rank, int64_t n_total, float eps, float momentum) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        __shared__ int pos[4 * 64];
        __shared__ double tot[64];
        for (int64_t r = tid; r < n_rows; r += 1024) {
            const int id = idx[r];
            if (id < 0 && -1 - id < 4 * 64) pos[-1 - id] = (int)r;
        }
        __syncthreads();
        if (tid < 64) {
            const int k = tid >> 4, jj = tid & 15;            // sums row k of a peer holds doubles 16 k .. 16 k + 15
            double t = 0.0;
            for (int p = 0; p < world; ++p)
                t += p == rank ? job.sums[tid] : reinterpret_cast<const double *>(in + (int64_t)pos[4 * p + k] * 32)[jj];
            tot[tid] = t;
        }
        __syncthreads();
        bn_record_from_sums(job, tot, 32, n_total, eps, momentum);
        return;
    }
    const int64_t total = n_rows * 32;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + tid; i < total; i += (int64_t)(gridDim.x - 1) * 1024) {
        const int id = idx[i >> 5];
        if (id >= 0) a[(n_own + id) * 32 + (i & 31)] = in[i];
    }
}

// [r6] unpack1 + merge of the adjacency branch in ONE launch (the sharded step's chain NNConv -> pack -> all-to-all -> unpack ->
// merge is serial on the main stream; every launch on it costs its duration plus a 4 - 12 us gap: profiles/r06_sharded_trace.txt).
// Every block derives the first BatchNorm's record from the shards' sums itself (own sums + the peers' sums rows of the message,
// added in rank order exactly as shard_unpack1_kernel's block 0 does: the same bits; block 0 writes the record and the running
// statistics) and reads the HALO rows of a1 straight out of the message instead of from a scattered copy: halo row h of peer p is
// message row h + 4 p (a peer's rows are followed by its 4 sums rows), p found from the sums rows' positions.
__global__ __launch_bounds__(256) void shard_unpack1_merge_kernel(const float *__restrict__ in, const int *__restrict__ idx,
                                                                  int64_t n_in, int64_t n_own, const float *__restrict__ a1, BnJob job,
                                                                  int world, int rank, int64_t n_total, float eps, float momentum,
                                                                  const float *__restrict__ a2, const float *__restrict__ st2,
                                                                  const float *__restrict__ resid, int64_t n_rows,
                                                                  float *__restrict__ out, unsigned *__restrict__ absmax_out) {
    constexpr int c = 32;
    __shared__ int pos[4 * 64];
    __shared__ int hend[64];                                  // halo rows of peers 0 .. p end here
    __shared__ double tot[64];
    __shared__ __attribute__((aligned(16))) float st1[4 * c];
    const int tid = threadIdx.x;
    for (int64_t r = tid; r < n_in; r += 256) {
        const int id = idx[r];
        if (id < 0 && -1 - id < 4 * 64) pos[-1 - id] = (int)r;
    }
    float pre_gamma = 1.f, pre_beta = 0.f, pre_rm = 0.f, pre_rv = 1.f;
    if (tid < c) {
        pre_gamma = job.gamma[tid];
        pre_beta = job.beta[tid];
        if (blockIdx.x == 0 && job.running_mean) {
            pre_rm = job.running_mean[tid];
            pre_rv = job.running_var[tid];
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int k = tid >> 4, jj = tid & 15;                // sums row k of a peer holds doubles 16 k .. 16 k + 15
        double t = 0.0;
        for (int p = 0; p < world; ++p)
            t += p == rank ? job.sums[tid] : reinterpret_cast<const double *>(in + (int64_t)pos[4 * p + k] * 32)[jj];
        tot[tid] = t;
        if (tid < world) hend[tid] = pos[4 * tid] - 4 * tid;  // (the first sums row of peer p sits behind its halo rows and 4 p sums rows)
    }
    __syncthreads();
    if (tid < c) {                                            // bn_record_from_sums' arithmetic
        const double inv_n = 1.0 / (double)n_total;
        const double mean = tot[tid] * inv_n;
        double var = tot[c + tid] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mh = (float)mean;
        st1[tid] = mh;
        st1[c + tid] = (float)(mean - (double)mh);
        st1[2 * c + tid] = (float)((double)pre_gamma / sqrt(var + (double)eps));
        st1[3 * c + tid] = pre_beta;
        if (blockIdx.x == 0) {
            job.stat[tid] = st1[tid]; job.stat[c + tid] = st1[c + tid];
            job.stat[2 * c + tid] = st1[2 * c + tid]; job.stat[3 * c + tid] = st1[3 * c + tid];
            if (job.running_mean) {
                const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
                job.running_mean[tid] = (float)((1.0 - (double)momentum) * (double)pre_rm + (double)momentum * mean);
                job.running_var[tid] = (float)((1.0 - (double)momentum) * (double)pre_rv + (double)momentum * unbiased);
            }
            if (tid == 0 && job.num_batches_tracked) *job.num_batches_tracked += 1;
        }
    }
    __syncthreads();
    float am = 0.f;
    const int64_t n4 = n_rows * c / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)((i * 4) % c);
        const int64_t row = i >> 3;
        float4 x1;
        if (row < n_own) {
            x1 = reinterpret_cast<const float4 *>(a1)[i];
        } else {
            const int h = (int)(row - n_own);
            int p = 0;
            while (p + 1 < world && h >= hend[p]) ++p;
            x1 = reinterpret_cast<const float4 *>(in)[((int64_t)h + 4 * p) * 8 + (i & 7)];
        }
        const float4 x2 = reinterpret_cast<const float4 *>(a2)[i];
        const float4 m1h = *reinterpret_cast<const float4 *>(st1 + col), m1l = *reinterpret_cast<const float4 *>(st1 + c + col);
        const float4 g1 = *reinterpret_cast<const float4 *>(st1 + 2 * c + col), b1 = *reinterpret_cast<const float4 *>(st1 + 3 * c + col);
        const float4 m2h = *reinterpret_cast<const float4 *>(st2 + col), m2l = *reinterpret_cast<const float4 *>(st2 + c + col);
        const float4 g2 = *reinterpret_cast<const float4 *>(st2 + 2 * c + col), b2 = *reinterpret_cast<const float4 *>(st2 + 3 * c + col);
        float4 y2, o;
        y2.x = bn_apply1(x2.x, m2h.x, m2l.x, g2.x, b2.x);
        y2.y = bn_apply1(x2.y, m2h.y, m2l.y, g2.y, b2.y);
        y2.z = bn_apply1(x2.z, m2h.z, m2l.z, g2.z, b2.z);
        y2.w = bn_apply1(x2.w, m2h.w, m2l.w, g2.w, b2.w);
        o.x = bn_apply1(x1.x, m1h.x, m1l.x, g1.x, b1.x) * y2.x;
        o.y = bn_apply1(x1.y, m1h.y, m1l.y, g1.y, b1.y) * y2.y;
        o.z = bn_apply1(x1.z, m1h.z, m1l.z, g1.z, b1.z) * y2.z;
        o.w = bn_apply1(x1.w, m1h.w, m1l.w, g1.w, b1.w) * y2.w;
        if (resid) {
            const float4 r = reinterpret_cast<const float4 *>(resid)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4 *>(out)[i] = o;
        am = absmax4(am, o);
    }
    absmax_flush(am, absmax_out);
}

static inline unsigned shard_copy_blocks(int64_t n_rows) {
    int64_t b = (n_rows * 64 + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (unsigned)b;
}
void launch_shard_pack_sums(const float *a1, const float *a2, const int *idx, int64_t n_rows, const BnJobs &jobs, float *out,
                            hipStream_t s) {
    shard_pack_sums_kernel<<<1 + shard_copy_blocks(n_rows), 1024, 0, s>>>(a1, a2, idx, n_rows, jobs, out);
}
void launch_shard_unpack_finalize(const float *in, const int *idx, int64_t n_rows, int64_t n_own, float *a1, float *a2,
                                  const BnJobs &jobs, int world, int rank, int64_t n_total, float eps, float momentum,
                                  hipStream_t s) {
    shard_unpack_finalize_kernel<<<1 + shard_copy_blocks(n_rows), 1024, 0, s>>>(in, idx, n_rows, n_own, a1, a2, jobs, world,
                                                                                rank, n_total, eps, momentum);
}

void launch_shard_pack1(const float *a, const int *idx, int64_t n_rows, const BnJob &job, float *out, hipStream_t s) {
    shard_pack1_kernel<<<1 + shard_copy_blocks((n_rows + 1) / 2), 1024, 0, s>>>(a, idx, n_rows, job, out);
}
void launch_shard_unpack1(const float *in, const int *idx, int64_t n_rows, int64_t n_own, float *a, const BnJob &job, int world,
                          int rank, int64_t n_total, float eps, float momentum, hipStream_t s) {
    shard_unpack1_kernel<<<1 + shard_copy_blocks((n_rows + 1) / 2), 1024, 0, s>>>(in, idx, n_rows, n_own, a, job, world, rank,
                                                                                 n_total, eps, momentum);
}

void launch_shard_unpack1_merge(const float *in, const int *idx, int64_t n_in, int64_t n_own, const float *a1, const BnJob &job,
                                int world, int rank, int64_t n_total, float eps, float momentum, const float *a2, const float *stat2,
                                const float *resid, int64_t n_rows, float *out, unsigned *absmax_out, hipStream_t s) {
    const int64_t n4 = n_rows * 32 / 4;
    shard_unpack1_merge_kernel<<<ew_grid(n4, 512), 256, 0, s>>>(in, idx, n_in, n_own, a1, job, world, rank, n_total, eps, momentum, a2,
                                                               stat2, resid, n_rows, out, absmax_out);
}

void launch_shard_pack(const float *a1, const float *a2, const int *idx, int64_t n_rows, const double *sums, float *out,
                       hipStream_t s) {
    if (n_rows > 0)
        shard_pack_kernel<<<ew_grid(n_rows * 64), 256, 0, s>>>(a1, a2, idx, n_rows, reinterpret_cast<const float *>(sums), out);
}
void launch_shard_unpack(const float *in, const int *idx, int64_t n_rows, int64_t n_own, float *a1, float *a2,
                         double *peer_sums, hipStream_t s) {
    if (n_rows > 0)
        shard_unpack_kernel<<<ew_grid(n_rows * 64), 256, 0, s>>>(in, idx, n_rows, n_own, a1, a2,
                                                                 reinterpret_cast<float *>(peer_sums));
}
void launch_shard_sum_peers(const double *peer_sums, const double *own, int world, int rank, double *total, hipStream_t s) {
    shard_sum_peers_kernel<<<1, 128, 0, s>>>(peer_sums, own, world, rank, total);
}

void launch_merge_bn1(const float *a1, const BnJob &j1, int64_t n_total, float eps, float momentum, const float *a2,
                      const float *stat2, const float *resid, int64_t n_nodes, float *out, hipStream_t s, unsigned *absmax_out) {
    const int64_t n4 = n_nodes * 32 / 4;
    if (j1.n_partials > 128) {
        int64_t blocks = (n4 + 1023) / 1024;
        if (blocks > 256) blocks = 256;
        merge_bn1_wide_kernel<<<(unsigned)blocks, 1024, 0, s>>>(a1, j1, n_total, eps, momentum, a2, stat2, resid, n4, out, absmax_out);
        return;
    }
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256) blocks = 256;          // every block repeats the statistics reduction
    merge_bn1_kernel<<<(unsigned)blocks, 256, 0, s>>>(a1, j1, n_total, eps, momentum, a2, stat2, resid, n4, out, absmax_out);
}

// largest |h[0 .. n_floats)| (n_floats % 4 == 0) as float bits, atomicMax into *max_bits (a zeroed word or a running bound)
__global__ void absmax_kernel(const float *__restrict__ h, int64_t n4, unsigned *__restrict__ out_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(h)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    absmax_flush(m, out_bits);
}
void launch_absmax(const float *h, int64_t n_floats, unsigned *max_bits, hipStream_t s) {
    const int64_t n4 = n_floats / 4;
    if (n4 <= 0) return;
    int64_t g = (n4 + 255) / 256;
    if (g > 512) g = 512;
    absmax_kernel<<<(unsigned)g, 256, 0, s>>>(h, n4, max_bits);
}

void launch_merge(const float *a1, const float *stat1, const float *a2, const float *stat2, const float *resid, int64_t n_nodes,
                  int c, float *out, float *h2_out, unsigned *absmax_out, hipStream_t s) {
    const int64_t n4 = n_nodes * c / 4;
    merge_kernel<<<ew_grid(n4, absmax_out ? 512 : 256 * 8), 256, 0, s>>>(a1, stat1, a2, stat2, resid, n4, c, out, h2_out, absmax_out);
}
void launch_bn_apply(const float *v, int64_t ldv, const float *stat, int64_t n_rows, int f, float *out, int64_t ldo,
                     unsigned *absmax_out, hipStream_t s) {
    bn_apply_kernel<<<ew_grid(n_rows * f, absmax_out ? 512 : 256 * 8), 256, 0, s>>>(v, ldv, stat, n_rows, f, out, ldo, absmax_out);
}

void launch_bn_finalize(const BnJobs &jobs, int n_jobs, int mode, int f, int64_t n_total, float eps, float momentum,
                        hipStream_t s) {
    const int slices = (f + 31) / 32;
    bn_finalize_kernel<<<n_jobs * slices, 1024, 0, s>>>(jobs, mode, f, slices, n_total, eps, momentum);
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int tgnn_bn_finalize(int32_t mode, const double *partials, int32_t n_partials, double *sums, int32_t f,
                                int64_t n_rows_total, const float *gamma, const float *beta, float eps, float momentum,
                                float *running_mean, float *running_var, int64_t *num_batches_tracked, float *stat,
                                tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(mode >= 0 && mode <= 3, "mode");
    TGNN_CHECK_ARG(f >= 1 && f <= 256, "feature count must be in [1,256]");
    TGNN_CHECK_ARG(mode == 3 || n_rows_total >= 1, "n_rows_total");
    TGNN_CHECK_ARG(mode == 1 || (gamma && beta && stat), "null pointer");
    TGNN_CHECK_ARG(!(mode == 0 || mode == 1) || (partials && n_partials >= 1 && n_partials <= TGNN_BN_MAX_PARTIALS), "partials");
    TGNN_CHECK_ARG(!(mode == 1 || mode == 2) || sums, "sums");
    TGNN_CHECK_ARG(mode != 3 || (running_mean && running_var), "running stats");
    TGNN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running stats come in pairs");
    BnJobs jobs{};
    jobs.job[0] = BnJob{partials, n_partials, sums, gamma, beta, running_mean, running_var, num_batches_tracked, stat};
    const int slices = (f + 31) / 32;
    bn_finalize_kernel<<<slices, 1024, 0, static_cast<hipStream_t>(stream)>>>(jobs, mode, f, slices, n_rows_total, eps, momentum);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_bn_apply(const float *v, int64_t ldv, const float *stat, int64_t n_rows, int32_t f, float *out,
                             int64_t ldo, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (n_rows <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(v && stat && out && f >= 1 && ldv >= f && ldo >= f, "arguments");
    launch_bn_apply(v, ldv, stat, n_rows, f, out, ldo, nullptr, static_cast<hipStream_t>(stream));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_merge_fwd(const float *a1, const float *stat1, const float *a2, const float *stat2,
                              const float *resid, int64_t n_nodes, int32_t c, float *out, float *h2_out,
                              tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (n_nodes <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(a1 && stat1 && a2 && stat2 && out, "null pointer");
    TGNN_CHECK_ARG(c >= 4 && c % 4 == 0, "width must be a multiple of 4");
    TGNN_CHECK_ARG(((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)h2_out |
                    (uintptr_t)stat1 | (uintptr_t)stat2) % 16 == 0, "pointers must be 16-byte aligned");
    launch_merge(a1, stat1, a2, stat2, resid, n_nodes, c, out, h2_out, nullptr, static_cast<hipStream_t>(stream));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_rows_gather(const float *src, int64_t ld_src, const int32_t *idx, int64_t n_idx, int32_t c,
                                float *out, int64_t ld_out, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (n_idx <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(src && idx && out && c >= 1 && ld_src >= c && ld_out >= c, "arguments");
    rows_gather_kernel<<<ew_grid(n_idx * c), 256, 0, static_cast<hipStream_t>(stream)>>>(src, ld_src, idx, n_idx, c, out,
                                                                                         ld_out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_rows_scatter(const float *in, const int32_t *idx, int64_t n_idx, int32_t c, float *dst,
                                 int64_t ld_dst, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (n_idx <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(in && idx && dst && c >= 1 && ld_dst >= c, "arguments");
    rows_scatter_kernel<<<ew_grid(n_idx * c), 256, 0, static_cast<hipStream_t>(stream)>>>(in, idx, n_idx, c, dst, ld_dst);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
