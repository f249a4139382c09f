// Shared helpers for the gfx950 kernels of libtgnn.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/tgnn.h"

namespace tgnn {

constexpr int kWave = 64;                 // CDNA wavefront
constexpr float kLeakySlope = 0.01f;      // torch.nn.LeakyReLU() default (TilinGNN.py:31)

void set_error(const char *fmt, ...);     // thread-local message, api.hip

#define TGNN_CHECK_ARG(cond, msg)                                                     \
    do {                                                                              \
        if (!(cond)) {                                                                \
            ::tgnn::set_error("%s: invalid argument: %s (%s)", __func__, msg, #cond); \
            return TGNN_ERR_INVALID_ARG;                                              \
        }                                                                             \
    } while (0)

#define TGNN_CHECK_LAUNCH()                                                                 \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) {                                                            \
            ::tgnn::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));   \
            return TGNN_ERR_LAUNCH;                                                         \
        }                                                                                   \
    } while (0)

#define TGNN_CHECK_HIP(expr)                                                                \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            ::tgnn::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e__)); \
            return TGNN_ERR_LAUNCH;                                                         \
        }                                                                                   \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Every entry point launches on the device its stream belongs to, whatever the calling thread's current device is
// (a model on cuda:1 called while cuda:0 is current): the guard switches for the duration of the call.  The NULL
// stream belongs to the current device by definition.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(tgnn_stream_t stream) {
        if (!stream) return;
        int cur = -1;
        hipDevice_t dev = -1;
        if (hipGetDevice(&cur) != hipSuccess || hipStreamGetDevice(static_cast<hipStream_t>(stream), &dev) != hipSuccess) return;
        if (dev != cur && hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// The opt-in to more than 64 KB of dynamic LDS is a per-device attribute of a kernel: applied once per device and
// call site (`done` = that site's static flag array; idempotent, racing setters write the same value).
struct LdsOptIn {
    std::atomic<bool> done[64];
};
template <class Kern>
static inline hipError_t opt_in_dynamic_lds(Kern kern, int bytes, LdsOptIn &site) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && site.done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && tracked) site.done[dev].store(true, std::memory_order_release);
    return e;
}

// Compute units of the current device (of the partition, in CPX / NPS modes), cached per device; 256 when the query fails.
// The producers size their persistent grids as "one block per CU minus a few left to the other chain" with it.
static inline int device_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev].store(cus, std::memory_order_relaxed);
    return cus;
}
// blocks of a persistent producer: all CUs but `reserve` (never fewer than a quarter of them)
static inline int cus_minus(int reserve) {
    const int cus = device_cus();
    const int left = cus - reserve;
    return left > cus / 4 ? left : (cus / 4 > 0 ? cus / 4 : 1);
}

// Carves aligned sub-buffers out of a caller-provided workspace.
struct Carver {
    char *base;
    size_t off = 0, cap;
    Carver(void *p, size_t bytes) : base(static_cast<char *>(p)), cap(bytes) {}
    template <typename T>
    T *take(size_t count) {
        off = align_up(off, 256);
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
    bool ok() const { return off <= cap; }
};

// sigmoid on the hardware transcendentals: v_exp_f32 (via exp2(x * log2 e)) and v_rcp_f32, both
// ~1 ulp; worst-case relative error ~6e-7 at |x| = 10 (argument rounding), two orders below the
// 1e-5 parity bar.  The IEEE expf + division sequence cost as many VALU cycles as the GIN MLP's
// MFMAs (64 sigmoids per lane per 32-row tile).
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// 1 / (1 + exp(-v)) at the accuracy of libm's expf + an IEEE division (both ~3 ulp at worst, 0.4 on average: the form restated on
// the host in tests/test_sigmoid_form.py; the hardware transcendentals alone, sigmoidf_, lose ~30 ulp at |v| ~ 20) in a third of their instructions -- the output
// sigmoid of the collision branch's MLP, 40 % of that kernel's vector stream with libm (profiles/r05_pmc_gin_mlp.txt):
// t = -v log2 e as th + tl (the rounding of v log2 e is what costs accuracy at |v| ~ 10), e = 2^th (1 + tl ln 2), r = 1 / (1 + e)
// from the hardware reciprocal refined by one Newton step.  EVERY schedule's collision MLP ends in this function (gin.hip,
// forward_persist.h, forward_small.hip): the schedules are compared bit for bit.
__device__ __forceinline__ float sigmoid_out_f32(float v) {
    constexpr float kL2eH = 1.44269502162933349609375f, kL2eL = 1.925963033500011e-8f, kLn2 = 0.693147182464599609375f;
    const float nv = -fminf(fmaxf(v, -87.0f), 87.0f);         // (e stays finite and non-zero: beyond +-87 the result is 0 / 1 to fp32 either way)
    const float th = nv * kL2eH;
    const float tl = fmaf(nv, kL2eH, -th) + nv * kL2eL;
    const float eh = __builtin_amdgcn_exp2f(th);
    const float e = fmaf(eh, tl * kLn2, eh);
    const float d = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return v != v ? v : r;                                    // (fmaxf / fminf drop a NaN: a diverged branch must stay visible, as in torch)
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == TGNN_ACT_LEAKY_RELU) return v >= 0.f ? v : v * kLeakySlope;
    if (act == TGNN_ACT_SIGMOID) return sigmoidf_(v);
    return v;
}
__device__ __forceinline__ float leakyf_(float v) { return v >= 0.f ? v : v * kLeakySlope; }

// BatchNorm apply from a stat record [4][F] (mean_hi, mean_lo, ginv, beta) -- see tgnn.h
__device__ __forceinline__ float bn_apply1(float v, float mhi, float mlo, float g, float b) {
    return ((v - mhi) - mlo) * g + b;
}

// ------------------------------------------------------------------------------------------
// Exact split of 8 fp32 values into three bf16 pieces (hi + mid + lo == x), packed for the bf16 MFMAs.
// By TRUNCATION on the bit pattern: hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = x - hi - mid.  The 24
// significant bits of x fall 8 + 8 + 8 into the three pieces, every subtraction is exact, lo needs no rounding.
// 4.5 vector instructions per element (2 AND, 2 SUB as packed pairs, 3 byte-permutes per pair) against ~7 for
// the convert-based sequence (convert, re-expand, subtract, twice).
// ------------------------------------------------------------------------------------------
using tgnn_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using tgnn_u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ void split3_trunc(const float (&x)[8], tgnn_bf16x8 &hi, tgnn_bf16x8 &mid, tgnn_bf16x8 &lo) {
    tgnn_u32x4 ph, pm, pl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned int a0 = __float_as_uint(x[2 * p]), a1 = __float_as_uint(x[2 * p + 1]);
        const float r0 = x[2 * p] - __uint_as_float(a0 & 0xffff0000u);
        const float r1 = x[2 * p + 1] - __uint_as_float(a1 & 0xffff0000u);
        const unsigned int b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
        const float s0 = r0 - __uint_as_float(b0 & 0xffff0000u);
        const float s1 = r1 - __uint_as_float(b1 & 0xffff0000u);
        // element 2p in the low half, 2p+1 in the high half: bytes {a0[2], a0[3], a1[2], a1[3]}
        ph[p] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        pm[p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        pl[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    hi = __builtin_bit_cast(tgnn_bf16x8, ph);
    mid = __builtin_bit_cast(tgnn_bf16x8, pm);
    lo = __builtin_bit_cast(tgnn_bf16x8, pl);
}

// ------------------------------------------------------------------------------------------
// fp16 x 2 split precision (the 3-term alternative to bf16 x 3 where a bound on the operand's magnitude is at hand): with a
// power-of-two scale s such that |s x| < 2^15,   hi = RN16(s x),  lo = RN16(s x - hi)   hold s x to 2^-22 relative (fp16
// subnormals: 2^-25 absolute, i.e. 2^-39 of the scaled maximum), and  hi.hi + hi.lo + lo.hi  is the product to the same
// 2^-22 -- three matrix instructions instead of six, 24 vector instructions per 8 elements instead of 44.
// The bound travels as the BIT PATTERN of max |x| (atomicMax on unsigned: the order of non-negative floats), written by the
// kernel that produced x.
// ------------------------------------------------------------------------------------------
using tgnn_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ void split2_f16(const float (&x)[8], float s, tgnn_f16x8 &hi, tgnn_f16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = x[j] * s;
        // the product as ONE fp32 value: left to itself hipcc folds the multiply into the conversions (v_fma_mix) in one place and
        // not in the other, and where s is not a power of two (the NNConv root column: deg * 2^k) the two roundings of "hi"
        // then differ by an fp16 ulp now and then -- lo belongs to another hi, 2^-11 off (measured: 5e-5 instead of 2e-7)
        asm volatile("" : "+v"(a));
        hi[j] = (_Float16)a;
        lo[j] = (_Float16)(a - (float)hi[j]);                 // the difference is exact
    }
}
// the power of two s with  s * bound * 2^extra_log2 < 2^15,  bound given as float bits (0, inf, nan -> 1)
__device__ __host__ __forceinline__ float pow2_scale_for(unsigned bound_bits, int extra_log2) {
    const int e = (int)((bound_bits >> 23) & 0xffu);         // bound < 2^(e - 126)
    if (e == 0 || e == 255) return 1.0f;
    int se = 127 + 15 - (e - 126) - extra_log2;              // biased exponent of s
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    const unsigned b = (unsigned)se << 23;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
// max of |x| over the BLOCK into a device word holding float bits (nothing when out == NULL; every thread of the block calls).
// Same-address atomics cost ~5 ns each at the L2 and the waves of an element-wise kernel all finish together: one atomic per
// wavefront made a 256-block merge 25 us slower, so the block folds its waves through LDS first and thread 0 sends one
// (fire and forget: reading the word first to skip the atomic put a second L2 round trip on the tail of every block).
__device__ __forceinline__ void absmax_flush(float m, unsigned *out) {
    if (!out) return;                                         // (uniform)
    __shared__ float wave_max[16];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    if (lane == 0) wave_max[wave] = m;
    __syncthreads();
    if (wave == 0) {
        m = lane < n_waves ? wave_max[lane] : 0.f;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        if (lane == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}
__device__ __forceinline__ float absmax4(float m, const float4 &v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// Number of persistent blocks for a row-parallel producer: also the number of BN partial rows.
static inline int producer_blocks(int64_t n_rows, int rows_per_block) {
    int64_t nb = (n_rows + rows_per_block - 1) / rows_per_block;
    if (nb < 1) nb = 1;
    if (nb > TGNN_BN_MAX_PARTIALS) nb = TGNN_BN_MAX_PARTIALS;
    return static_cast<int>(nb);
}

// ---- parameter table of tgnn_forward / tgnn_backward: indices into params_host (order = tgnn_param_name) ---------------
constexpr int kInitStride = 7, kLayerStride = 25, kFinalStride = 7;
struct BnPtrs {
    const float *gamma, *beta;
    float *rm, *rv;
    int64_t *nbt;
};
struct Params {
    const void *const *p;
    int depth;
    const float *f(int i) const { return static_cast<const float *>(p[i]); }
    BnPtrs bn(int i) const {
        return BnPtrs{f(i), f(i + 1), const_cast<float *>(f(i + 2)), const_cast<float *>(f(i + 3)),
                      const_cast<int64_t *>(static_cast<const int64_t *>(p[i + 4]))};
    }
    int init(int l) const { return l * kInitStride; }                // w, b, bn x5
    int layer(int i) const { return 2 * kInitStride + i * kLayerStride; }
    // layer block: 0-5 edge mlp (w1 b1 w2 b2 w3 b3), 6 root, 7 bias, 8-12 bn1, 13 eps, 14-19 gin mlp, 20-24 bn2
    int fin(int l) const { return 2 * kInitStride + depth * kLayerStride + l * kFinalStride; }
    int last() const { return fin(4); }
};

// ---- internal launchers shared between translation units -------------------------------
struct BnJob {
    const double *partials;
    int n_partials;
    double *sums;  // mode 1 output / mode 2 input, [2][F]
    const float *gamma, *beta;
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    float *stat;  // [4][F]
};
struct BnJobs {
    BnJob job[2];
};
// threads 0 .. f-1: record + running statistics from the column sums tot[0 .. 2f) over n_total rows (bn_finalize mode 2)
__device__ __forceinline__ void bn_record_from_sums(const BnJob &jb, const double *tot, int f, int64_t n_total, float eps,
                                                    float momentum) {
    const int tid = threadIdx.x;
    if (tid < f) {
        const float gamma = jb.gamma[tid], beta = jb.beta[tid];
        const double inv_n = 1.0 / (double)n_total;
        const double mean = tot[tid] * inv_n;
        double var = tot[f + tid] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mh = (float)mean;
        jb.stat[tid] = mh;
        jb.stat[f + tid] = (float)(mean - (double)mh);
        jb.stat[2 * f + tid] = (float)((double)gamma / sqrt(var + (double)eps));
        jb.stat[3 * f + tid] = beta;
        if (jb.running_mean) {
            const double unbiased = n_total > 1 ? var * ((double)n_total / (double)(n_total - 1)) : var;
            jb.running_mean[tid] = (float)((1.0 - (double)momentum) * (double)jb.running_mean[tid] + (double)momentum * mean);
            jb.running_var[tid] = (float)((1.0 - (double)momentum) * (double)jb.running_var[tid] + (double)momentum * unbiased);
        }
    }
    if (tid == 0 && jb.num_batches_tracked) *jb.num_batches_tracked += 1;
}

// The collision branch's BatchNorm record written by the LAST block of the GIN MLP kernel to finish (gin.hip): a ticket counter
// (zero before the launch, reset by that block) replaces the 1-block finalize launch behind the kernel; the reduction of the
// partial rows follows bn_finalize_kernel's tree (16 row groups, fixed fold), hence the same bits.
struct GinFin {
    unsigned *counter;           // NULL: no folded finalize
    BnJob job;                   // (n_partials is filled in by the launcher)
    int64_t n_total;
    float eps, momentum;
};
// The same for any producer whose blocks leave one partial row of 2 F doubles each (F <= blockDim.x): called by ALL threads of
// every block after the block's row has been stored with st_partial_sc1 (write-through: another block reads it).  scratch =
// (16 + 1) * 2 F doubles of LDS + one word behind them.  bn_finalize_kernel's tree, the same bits.
__device__ __forceinline__ void st_partial_sc1(double *bn_partial, int64_t idx, double v) {
    using u32x2_ = __attribute__((ext_vector_type(2))) unsigned int;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(bn_partial, 0, (int)0x80000000u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, v), prs, (uint32_t)idx * 8u, 0, 16);
}
constexpr size_t bn_fold_scratch_bytes(int f) { return (size_t)17 * 2 * f * sizeof(double) + 16; }
template <int F>
__device__ __forceinline__ void bn_fold_finish(const GinFin &fin, double *bn_partial, double *scratch) {
    using u32x2_ = __attribute__((ext_vector_type(2))) unsigned int;
    constexpr int kRow = 2 * F;
    double *fred = scratch, *ftot = scratch + 16 * kRow;
    unsigned *ticket = reinterpret_cast<unsigned *>(ftot + kRow);
    const int tid = threadIdx.x, nthr = blockDim.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this block's row has been written through before its ticket
    __syncthreads();
    if (tid == 0) *ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*ticket != gridDim.x - 1) return;                    // (uniform)
#ifdef TGNN_ABL_NOFOLDWORK
    // (timing ablation, VERDICT r5 item 2: the last block resets the ticket and leaves -- no fold, a stale record: what moving the
    //  fold into the consumers could give the forward at most)
    if (tid == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
#endif
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(bn_partial, 0, (int)0x80000000u, 0x00020000);
    const int np = (int)gridDim.x;
    for (int item = tid; item < 16 * kRow; item += nthr) {
        const int j = item % kRow, g = item / kRow;
        double acc = 0.0;
        for (int p = g; p < np; p += 8 * 16) {               // eight coherent loads in flight (rows past the end fetch nothing)
            u32x2_ v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pp = p + u * 16;
                v[u] = __builtin_amdgcn_raw_buffer_load_b64(prs, pp < np ? ((uint32_t)pp * (uint32_t)kRow + (uint32_t)j) * 8u : 0x80000000u, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p + u * 16 < np) acc += __builtin_bit_cast(double, v[u]);
        }
        fred[g * kRow + j] = acc;
    }
    __syncthreads();
    for (int j = tid; j < kRow; j += nthr) {
        double t = 0.0;
        for (int gg = 0; gg < 16; ++gg) t += fred[gg * kRow + j];
        ftot[j] = t;
    }
    __syncthreads();
    bn_record_from_sums(fin.job, ftot, F, fin.n_total, fin.eps, fin.momentum);
    if (tid == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// [r6] The same in TWO levels, for any width F <= blockDim.x: called by ALL threads of every block after the block's partial row
// (2 F doubles at bn_partial + blockIdx.x * 2 F) has been stored with st_partial_sc1.  The last block of each of the 16 row
// GROUPS g = blockIdx.x & 15 (ticket fin.counter[1 + g]) folds its group's rows in ascending order into group_rows[g]; the last
// group-finisher (ticket fin.counter[0]) adds the group rows in ascending g and writes the record -- bn_finalize_kernel's tree,
// the same bits -- so that only 16 rows are folded behind the kernel's last block instead of all of them.  fin.counter: 17
// zeroed words, left zeroed; group_rows: 16 x 2 F doubles; ftot: 2 F doubles of LDS, ticket: one LDS word (both free to reuse
// dead tiles).  fin.job.n_partials is not read (gridDim.x rows).
template <int F>
__device__ __forceinline__ void bn_fold_two_level(const GinFin &fin, double *bn_partial, double *group_rows, double *ftot,
                                                  unsigned *ticket) {
    using u32x2_ = __attribute__((ext_vector_type(2))) unsigned int;
    constexpr int kRow = 2 * F, kSc1 = 16;
    const int tid = threadIdx.x, nthr = blockDim.x, np = (int)gridDim.x, g = (int)(blockIdx.x & 15u);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(bn_partial, 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(group_rows, 0, (int)0x80000000u, 0x00020000);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this block's row has been written through before its ticket
    __syncthreads();
    if (tid == 0) *ticket = __hip_atomic_fetch_add(fin.counter + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*ticket != (unsigned)((np - g + 15) / 16) - 1u) return;   // (uniform) not the last of its group
    for (int j = tid; j < kRow; j += nthr) {
        double acc = 0.0;
        for (int u0 = 0; g + u0 * 16 < np; u0 += 16) {        // sixteen rows at a time: one batch up to 256 partial rows
            u32x2_ v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int pp = g + (u0 + u) * 16;
                v[u] = __builtin_amdgcn_raw_buffer_load_b64(prs, pp < np ? ((uint32_t)pp * (uint32_t)kRow + (uint32_t)j) * 8u : 0x80000000u, 0, kSc1);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (g + (u0 + u) * 16 < np) acc += __builtin_bit_cast(double, v[u]);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, acc), grs, ((uint32_t)g * (uint32_t)kRow + (uint32_t)j) * 8u, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(fin.counter + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int n_groups = np < 16 ? np : 16;
    if (*ticket != (unsigned)n_groups - 1u) return;          // (uniform) not the last group-finisher
    for (int j = tid; j < kRow; j += nthr) {
        u32x2_ v[16];
#pragma unroll
        for (int gg = 0; gg < 16; ++gg)
            v[gg] = __builtin_amdgcn_raw_buffer_load_b64(grs, gg < n_groups ? ((uint32_t)gg * (uint32_t)kRow + (uint32_t)j) * 8u : 0x80000000u, 0, kSc1);
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg)
            if (gg < n_groups) t += __builtin_bit_cast(double, v[gg]);
        ftot[j] = t;
    }
    __syncthreads();
    bn_record_from_sums(fin.job, ftot, F, fin.n_total, fin.eps, fin.momentum);
    if (tid == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one block per job (1 or 2 BatchNorms of the same width finalised by one launch); bn_merge.hip
void launch_bn_finalize(const BnJobs &jobs, int n_jobs, int mode, int f, int64_t n_total, float eps, float momentum,
                        hipStream_t s);
// merge (width 32) that derives the first BatchNorm's record from its partial rows itself (mode-0 semantics of
// bn_finalize incl. running-stat update, bit-identical statistics); bn_merge.hip
void launch_merge_bn1(const float *a1, const BnJob &j1, int64_t n_total, float eps, float momentum, const float *a2,
                      const float *stat2, const float *resid, int64_t n_nodes, float *out, hipStream_t s,
                      unsigned *absmax_out = nullptr);
// tgnn_merge_fwd / tgnn_bn_apply with the optional bound output (largest |out| as float bits, atomicMax into a zeroed word)
void launch_merge(const float *a1, const float *stat1, const float *a2, const float *stat2, const float *resid, int64_t n_nodes,
                  int c, float *out, float *h2_out, unsigned *absmax_out, hipStream_t s);
void launch_bn_apply(const float *v, int64_t ldv, const float *stat, int64_t n_rows, int f, float *out, int64_t ldo,
                     unsigned *absmax_out, hipStream_t s);
// sharded forward, one all-to-all per layer (width 32): pack halo rows of both branches + local BN sums, unpack, add
// the shards' sums in rank order; bn_merge.hip
void launch_shard_pack(const float *a1, const float *a2, const int *idx, int64_t n_rows, const double *sums, float *out,
                       hipStream_t s);
void launch_shard_unpack(const float *in, const int *idx, int64_t n_rows, int64_t n_own, float *a1, float *a2,
                         double *peer_sums, hipStream_t s);
void launch_shard_sum_peers(const double *peer_sums, const double *own, int world, int rank, double *total, hipStream_t s);
// the same with the statistics folded in: block 0 of the pack kernel reduces the two BatchNorms' partial rows to this
// shard's sums (jobs.job[k].sums, 2 x 64 contiguous doubles) and writes them into the message; block 0 of the unpack kernel
// adds the shards' sums in rank order and writes both records (jobs.job[k].stat, running statistics).  world <= 64.
void launch_shard_pack_sums(const float *a1, const float *a2, const int *idx, int64_t n_rows, const BnJobs &jobs, float *out,
                            hipStream_t s);
// collectives over RCCL on stream s (rccl_comm.hip); comm from tgnn_rccl_comm_create
int rccl_alltoall_rows(void *comm, const float *send, float *recv, const int64_t *send_counts, const int64_t *recv_counts,
                       int world, int32_t row_floats, int32_t extra_rows, hipStream_t s);
int rccl_allreduce_f64(void *comm, double *buf, int64_t count, hipStream_t s);
// one branch at a time (rows of 32 floats; 4 sums rows = 64 doubles per peer): the split exchange of forward.hip
void launch_shard_pack1(const float *a, const int *idx, int64_t n_rows, const BnJob &job, float *out, hipStream_t s);
void launch_shard_unpack1(const float *in, const int *idx, int64_t n_rows, int64_t n_own, float *a, const BnJob &job, int world,
                          int rank, int64_t n_total, float eps, float momentum, hipStream_t s);
// [r6] unpack1 of the adjacency branch + merge over own AND halo rows in one launch (halo rows of a1 read from the message)
void launch_shard_unpack1_merge(const float *in, const int *idx, int64_t n_in, int64_t n_own, const float *a1, const BnJob &job,
                                int world, int rank, int64_t n_total, float eps, float momentum, const float *a2, const float *stat2,
                                const float *resid, int64_t n_rows, float *out, unsigned *absmax_out, hipStream_t s);
void launch_shard_unpack_finalize(const float *in, const int *idx, int64_t n_rows, int64_t n_own, float *a1, float *a2,
                                  const BnJobs &jobs, int world, int rank, int64_t n_total, float eps, float momentum,
                                  hipStream_t s);

// GraphConv edge-MLP parameters of up to 64 layers, passed by value to one batched launch; nnconv.hip
struct EdgeMlpLayer {
    const float *w1, *b1, *w2, *b2, *w3, *b3;
};
constexpr int kMaxDepth = 64;
struct EdgeMlpLayers {
    EdgeMlpLayer l[kMaxDepth];
};
// wtab [depth][T][C*C]
// roots + wimg_all given (width 32): the same launch also writes the NNConv operand images [(T+1)][kWtType] of all layers
unsigned edge_weight_table_blocks(int n_types, int fe, int depth, int c, bool image);   // what done_ctr of the launch below reaches
void launch_edge_weight_table_batched(const float *edge_attr, const int *type_rep_edge, int n_types, int fe,
                                      const EdgeMlpLayers &layers, int depth, int c, float *wtab, const float *const *roots,
                                      float *wimg_all, hipStream_t s, unsigned *done_ctr = nullptr,
                                      const unsigned *root_max = nullptr, float img_scale = 1.0f, const int *n_types_dev = nullptr,
                                      int max_types_dev = 0);
// [r6] n_types_dev (device): the kernel reads the type count itself (more than max_types_dev: it writes nothing) -- only where the
// launch's grid does not depend on the count:
bool edge_weight_table_device_count_ok(int fe, int c);
// img_scale (a power of two): on top of nnconv_weight_scale in the fp16-pair images -- kEgImageScale for the edge-group kernel
// root_max (device, [depth] words = max |root_i| as float bits, forward_scales below): the images are fp16-pair images
// [(T+1)][kWtTypeF16] instead.
// [r6] the init MLP (fx <= 8 -> 32 -> 32, LeakyReLU, train-mode BatchNorm) as three launches that recompute from x: init_mlp.hip.
// job0 / job1: the two BatchNorms, partials = distinct scratch areas of init_mlp_fused_blocks(n) x 64 doubles; out [n][32].
int init_mlp_fused_blocks(int64_t n);
int launch_init_mlp_fused(const float *x, int64_t ldx, int fx, const float *w0, const float *b0, const float *w1, const float *b1,
                          BnJob job0, BnJob job1, int64_t n, float eps, float momentum, float *out, unsigned *absmax_out,
                          hipStream_t s);
// Bounds of a forward's fp16-pair operands in one launch behind a memset: words [0, n_zero) = 0 (the slots' maxima, filled by the
// producers), root_max[i] = max |roots[i]|, *dense_max = max |dense_w[0 .. dense_n)|
// (dense_w == NULL [r6]: ONE kernel and no memset -- every root maximum is one block's plain store, block 0 clears the other
//  words; root_max must lie inside words[0, n_zero))
void launch_forward_scales(unsigned *words, int n_zero, const float *const *roots, int depth, unsigned *root_max,
                           const float *dense_w, int64_t dense_n, unsigned *dense_max, hipStream_t s);
// NNConv B-operand weight images [(T+1)][1152] for `depth` layers (roots[i] = layer i's root matrix); nnconv.hip
// (root_max given -- [depth] words, max |root_i| as float bits: fp16-pair images [(T+1)][kWtTypeF16] instead)
void launch_nnconv_weight_image(const float *wtab_all, const float *const *roots, int n_types, int depth,
                                float *wimg_all, hipStream_t s, const unsigned *root_max = nullptr, float img_scale = 1.0f);
constexpr float kEgImageScale = 1.0f / 32768.0f;   // nnconv_eg.hip: every weight below 1, so that 32 products of < 2^10 stay below 2^15
// h_max + root_max given (device words, float bits: max |h|, max |root|) with max_in_degree >= 1: wimg is an fp16-pair image and
// the kernel runs the 3-term fp16 split (operands scaled by powers of two from the bounds); else the bf16 x 3 image / split
int launch_nnconv_cols(const float *h, int64_t ldh, const int32_t *tile_col_ptr, const int32_t *col_meta,
                       const int32_t *col_src, const float *wimg, int32_t n_types, const float *bias,
                       int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                       hipStream_t s, const unsigned *h_max = nullptr, const unsigned *root_max = nullptr,
                       int max_in_degree = 0, unsigned long long *stamp = nullptr);
// the same NNConv over the edge-group structure (graph_prep.hip: nnconv_eg_kernel; nnconv_eg.hip): fp16-pair images and bounds as
// above, no in-degree needed (nothing is summed before the split)
// pack (counter != NULL; LeakyReLU, bn_partial given) [r6]: the sharded step's pack inside the kernel -- the epilogue also stores a row
// into the message slots that carry it, the last block writes the shard's BatchNorm sums (sums + the message's sums rows)
struct EgShardPack {
    const int *send_row_ptr, *send_row_slot;   // device, [n + 1], [n_send]
    float *msg;                                // device: the message, 32 floats per row
    const int *msg_idx;                        // device, [n_msg]: >= 0 a row, -1 - k the sums row k
    int64_t n_msg;
    double *sums;                              // device, [64]
    unsigned *counter;                         // device, 17 zeroed words (left zeroed)
    double *group_rows;                        // device, 16 x 64 doubles
};
int launch_nnconv_eg(const float *h, const int32_t *tile_grp_ptr, const int32_t *grp, const float *wimg, int32_t n_types, const float *bias, int64_t n_nodes, int32_t act, float *out,
                     double *bn_partial, int32_t *n_partials_host, hipStream_t s, const unsigned *h_max, const unsigned *root_max,
                     unsigned long long *stamp = nullptr, const EgShardPack *pack = nullptr);
// largest |h[0 .. n_floats)| (n_floats % 4 == 0) as float bits, atomicMax into *max_bits; bn_merge.hip
void launch_absmax(const float *h, int64_t n_floats, unsigned *max_bits, hipStream_t s);
// max |W_k| and a bound of |BN(v)| from the BatchNorm's parameters alone (nnconv.hip: dense_bounds_kernel), atomicMax into
// zeroed words: what the final MLP's inner layers need to run the fp16-pair kernel on a BatchNorm-on-load input
// (gamma[k] == NULL: the weights' bound only; the words are zeroed by the caller: several blocks per job fold into them)
void launch_dense_bounds(int n_jobs, const float *const *w, const int64_t *w_n, const float *const *gamma, const float *const *beta,
                         const int *f, unsigned *const *w_max, unsigned *const *a_max, int64_t n_total, hipStream_t s);
// tgnn_dense_act_fwd with bounds of both operands (a_max: of the input AFTER in_stat's BatchNorm, if any): dense.hip
// [r6] fold (counter != NULL) + fold_group_rows (16 x 2 out_dim doubles): the kernels that can -- the rows and the resident kernels --
// write the BatchNorm's record themselves (bn_fold_two_level) and set *folded; otherwise the caller launches its finalize
int dense_act_bounded(const float *a, int64_t lda, int64_t a_kblock_stride, const float *in_stat, const float *w, const float *b,
                      int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act, float *out, int64_t ldo, double *bn_partial,
                      int32_t *n_partials_host, const unsigned *a_max, int n_a_max, const unsigned *w_max, hipStream_t s,
                      const void *wimg = nullptr, const GinFin *fold = nullptr, double *fold_group_rows = nullptr, bool *folded = nullptr);
// the fp16-pair operand image of a Linear's W [out_dim][in_dim] (out_dim 64 / 128 / 256, in_dim % 32 == 0) that routes
// dense_act_bounded / dense_act_slots_bounded to the rows-per-wave kernel (dense.hip: dense_f16_rows_kernel)
constexpr int64_t kDenseRowsKernelMin = 49152;       // rows from which that kernel is taken (below: the block-tile kernels win)
size_t dense_f16_image_size(int in_dim, int out_dim);
int dense_f16_image_build(const float *w, int in_dim, int out_dim, const unsigned *w_max, void *wimg, hipStream_t s);
// [r6] up to four images in ONE launch (the final MLP's Linears; same bits as dense_f16_image_build each)
int dense_f16_images_build(int n_jobs, const float *const *w, const int *in_dim, const int *out_dim, const unsigned *const *w_max,
                           void *const *wimg, hipStream_t s);
// tgnn_dense_act_slots_fwd (no input BatchNorm) with bounds of both operands: a_max[0 .. n_a_max) / w_max = max |a| per slot /
// max |w| as float bits (device) -> the fp16-pair kernel (dense.hip: dense_split_kernel<.., F16>); dense.hip
int dense_act_slots_bounded(const float *a, int32_t slot_width, int64_t slot_stride, const float *w, const float *b,
                            int64_t n_rows, int32_t in_dim, int32_t out_dim, int32_t act, float *out, int64_t ldo,
                            double *bn_partial, int32_t *n_partials_host, const unsigned *a_max, int n_a_max,
                            const unsigned *w_max, hipStream_t s, const void *wimg = nullptr, const GinFin *fold = nullptr,
                            double *fold_group_rows = nullptr, bool *folded = nullptr);
// the MLP half of tgnn_gin_fwd (width 32) behind tgnn_gin_aggregate; gin.hip
// experiment knob (tgnn_debug_set_block_caps): upper bounds of the whole-CU kernels' grids, 0 = the built-in policy
extern std::atomic<int> g_debug_block_cap[2];   // [0] column NNConv, [1] GIN MLP
// inference: the forward's fp16-pair kernel (gin32_mlp16_kernel) instead of the bf16 x 3 one the training forward keeps
int launch_gin32_mlp(const float *z, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                     const float *b3, int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                     hipStream_t s, const GinFin *fin = nullptr, bool inference = false);
// tgnn_gin_fwd (width 32 fast path only: returns TGNN_ERR_UNSUPPORTED otherwise) with the BatchNorm finalize folded into the MLP
// kernel's last block
int gin32_fwd_folded(const float *a, int64_t lda, const float *in_stat, const int32_t *rowptr, const int32_t *col_src, const float *eps,
                     const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                     int64_t n_nodes, int32_t act, float *out, float *z_scratch, double *bn_partial, int32_t *n_partials_host,
                     const GinFin &fin, hipStream_t s, bool need_z = true);
// need_z = false (nobody reads the aggregate z afterwards; packed rows, LeakyReLU): aggregate and MLP as ONE kernel, z_scratch
// is not written (gin.hip: gin32_fused_kernel; tgnn_set_gin_fused(0) keeps the two-kernel form)
// Kernels that synchronise their blocks with spin barriers (the persistent small-layout forward, the one-launch small-layout
// preparation) need ALL their blocks resident; two of them started side by side from different streams or threads could each
// hold part of the CUs and wait for the rest for ever.  Every such launch goes through this per-device gate (forward_small.hip):
// it runs beside the spin kernels still in flight when all of them together fit the device, else behind them (events,
// whichever streams they are on).  `launch` queues the kernel on `s`; cus_needed = CUs its blocks occupy.
// (Stream capture: the shared events make these launches uncapturable into a HIP graph.)
int spin_kernel_chain(hipStream_t s, void (*launch)(void *ctx, hipStream_t s), void *ctx, int cus_needed);
// Small layouts: the whole forward behind a pre-pass as one persistent kernel (forward_small.hip).  small_layout_teams:
// 0 = not eligible (too large, too many edge types for LDS, ...); the packs / images are built per forward.
int small_layout_teams(const tgnn_model_dims *d, int64_t n_nodes, int n_types, int max_in_degree);
size_t small_pack_floats(int depth);
// zero / zero_bytes (a multiple of 16): memory the same launch clears (the mid-size kernel's tagged partial rows)
// fin0_f16_max: the final MLP's first Linear as an fp16-pair image scaled by pow2_scale_for(*fin0_f16_max) (forward_tail.hip)
void launch_small_pack(const Params &P, int depth, float *pack, unsigned *barrier_ctr, hipStream_t s, bool dense_images = true,
                       void *zero = nullptr, size_t zero_bytes = 0, const unsigned *fin0_f16_max = nullptr,
                       hipStream_t final_images_stream = nullptr);
int launch_forward_small(const tgnn_model_dims *d, const Params &P, const float *x, float *probs, float *mid, float *a2_0,
                         float *a2_1, const float *wimg, float *pack, const tgnn_graph *graph, double *part, double *part_wide,
                         double *runstat, unsigned *ctr, const unsigned *weights_done, unsigned weights_target, int64_t n,
                         int update_running, float eps, float momentum, hipStream_t s);
// Mid-size layouts (up to 65 536 nodes): the 20 layers as ONE persistent kernel between the general schedule's init and final
// MLP (forward_mid.hip).  mid_layout_tiles_per_block: 0 = not eligible; the pack (GIN images + parameter vectors) is
// launch_small_pack's, the NNConv images are the fp16-pair ones of launch_edge_weight_table_batched.
int mid_layout_tiles_per_block(const tgnn_model_dims *d, const tgnn_graph *g, int64_t n_nodes, int *blocks_out);
size_t mid_part_doubles();
int launch_forward_mid(const tgnn_model_dims *d, const Params &P, float *mid, float *a1, float *a2_0, float *a2_1, const float *wimg,
                       const float *pack, const tgnn_graph *graph, double *part, double *runstat, unsigned *ctr, unsigned *bounds,
                       int64_t n, int tiles_per_block, int blocks, int update_running, float eps, float momentum, hipStream_t s,
                       const unsigned *weights_done = nullptr, unsigned weights_target = 0, double *const *tail_zero = nullptr,
                       size_t tail_zero_doubles = 0, const float *x_init = nullptr);
// x_init: the node features -- the init MLP then runs in the kernel's prologue (node_features_dim <= 8; the pack must carry the
// dense images) instead of as launches in front of it
// ... and the final MLP behind it as one persistent kernel too (forward_tail.hip): 0 = not eligible / switched off
// (tgnn_set_mid_tail); needs a pack built with dense_images; part / gpart: mid_tail_part_doubles() doubles each, ZEROED by
// the layer loop's kernel in front of it (launch_forward_mid: tail_zero)
int mid_tail_tiles_per_block(const tgnn_model_dims *d, int64_t n_nodes, int *blocks_out);
bool mid_init_in_kernel();     // tgnn_set_mid_tail bit 1
size_t mid_tail_part_doubles();
const float *small_dense_image(const float *pack, int depth, int k);
int launch_forward_tail(const tgnn_model_dims *d, const Params &P, const float *mid, const float *pack, float *probs, double *part,
                        double *gpart, const unsigned *slot_max, const unsigned *w0_max, int64_t n, int tiles_per_block, int blocks,
                        int update_running, float eps, float momentum, hipStream_t s, const int *verdict = nullptr);
// MFMA weight image of the column NNConv, per type: [plane 3 (hi, mid, lo)][M block 2][g 4][i 16] x 8 bf16 --
// the A fragment of lane 16 g + i for one (plane, M block) is one 16-byte read, a wavefront reads 1 KB in lane order
// (conflict-free: SQ_LDS_BANK_CONFLICT 2.3e6 -> 2.3e5 per launch against the [i][g] order of round 1); 6144 B per type
constexpr int kWtPlane = 2 * 64 * 4;       // floats (16-byte fragments x 4) per plane
constexpr int kWtType = 3 * kWtPlane;      // floats per type
// fp16-pair image (layouts whose largest in-degree is known, single device): [plane 2 (hi, lo)] of the same fragment order,
// every weight multiplied by the layer's power of two nnconv_weight_scale(max |root|) first; 4096 B per type
constexpr int kWtTypeF16 = 2 * kWtPlane;
__device__ __host__ __forceinline__ float nnconv_weight_scale(unsigned root_max_bits) {
    // the edge-MLP weights are sigmoids (< 1), the root matrix is a free parameter: one scale for both (they share accumulators)
    float m;
    __builtin_memcpy(&m, &root_max_bits, 4);
    m = m > 1.0f ? m : 1.0f;
    unsigned b;
    __builtin_memcpy(&b, &m, 4);
    return pow2_scale_for(b, 0);
}
}  // namespace tgnn
