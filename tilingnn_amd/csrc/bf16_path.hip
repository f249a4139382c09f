// BASELINE config 3: network_width 64 with bf16 STORAGE of the activations that travel through HBM between kernels
// (the skip buffer `mid`, the pre-BatchNorm branch outputs a1 / a2, the GIN aggregate z) -- fp32 accumulation in every
// product and sum, fp64 BatchNorm statistics, fp32 stat records, fp32 final-MLP activations.
// (/root/reference/inputs/config.py:17 "30-60-90 + equilateral" -> tile_count 4; :37-38 depth / width; SURVEY 8d #3.)
//
// What changes against the fp32 path (csrc/nnconv_cols.hip, gin.hip, bn_merge.hip, dense.hip):
//   * a gathered row IS a matrix-core operand: 64 bf16 = 128 bytes, lane (row fj, k-group fq) loads the 8 channels
//     32 kc + 8 fq .. of K chunk kc with one 16-byte load -- no split into bf16 pieces, no pre-add in registers: every
//     column of the type-column structure goes straight into 8 MFMAs (4 output blocks x 2 K chunks) against the
//     bf16 image of W_type, fp32 accumulators;
//   * the per-type NNConv matrices, the GIN MLP weights and the first final Linear are rounded to bf16 once per
//     forward (one plane instead of three): products are exact, sums fp32;
//   * every producer rounds its output to bf16 (RNE) when storing and takes the BatchNorm sums of the ROUNDED values, so
//     that the statistics are those of the data the consumer normalises.
// Parity (tests/test_bf16_path.py): per op, teacher forced on bf16-rounded inputs against the fp64 oracle; the stated
// tolerance is 2^-7 = 7.8e-3 of the output's max-norm (one bf16 rounding of the weights, one of the output).
#include <atomic>
#include <type_traits>

#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using bf16x4s = __attribute__((ext_vector_type(4))) short;   // (operand type of the K = 16 bf16 matrix builtin)

constexpr int kC = 64;                      // network_width of this path
constexpr int kW64Frag = 8 * 64;            // 16-byte fragments per type image: [M block 4][K chunk 2][lane 64]
constexpr int kW64Floats = kW64Frag * 4;    // the same in 4-byte units (8 KB per type)

__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ void unpack8(const bf16x8 &p, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (float)p[k];
}

// ------------------------------------------------------------------------------------------ small element-wise kernels
// out (bf16) = BN(v) for a [N, F] fp32 activation: middle[0] = brch_1 = brch_2 (TilinGNN.py:55,58)
__global__ void bn_apply_bf16_kernel(const float *__restrict__ v, const float *__restrict__ stat, int64_t n, int f,
                                     __bf16 *__restrict__ out) {
    const int64_t total = n * f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % f);
        out[i] = (__bf16)bn_apply1(v[i], stat[c], stat[f + c], stat[2 * f + c], stat[3 * f + c]);
    }
}

// the same for 64 columns, 8 per thread (a thread keeps its columns: the grid stride is a multiple of 64), the record through
// registers: BatchNorm of the collision branch applied to the fp32 rows its MLP left (gin64_bf16_mlp_kernel<3>)
__global__ __launch_bounds__(256) void bn_apply64_bf16_kernel(const float *__restrict__ v, const float *__restrict__ stat, int64_t n8,
                                                              __bf16 *__restrict__ out) {
    // [r5] the record as 8 16-byte loads per thread instead of 32 4-byte ones, and enough rows per thread to pay for them
    // (launch_bn_apply64: ~6 items per thread, two in flight): 20 -> ~10 us per layer at 100 000 nodes, on config 3's long chain
    const int c0 = (int)(threadIdx.x & 7) * 8;               // ((block * 256 + thread) % 8: 256 is a multiple of 8)
    float mh[8], ml[8], g[8], b[8];
    {
        const float4 *s4 = reinterpret_cast<const float4 *>(stat + c0);
        const float4 a0 = s4[0], a1 = s4[1], b0 = s4[kC / 4], b1 = s4[kC / 4 + 1], c0v = s4[2 * kC / 4], c1v = s4[2 * kC / 4 + 1],
                     d0 = s4[3 * kC / 4], d1 = s4[3 * kC / 4 + 1];
        const float t0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, t1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w},
                    t2[8] = {c0v.x, c0v.y, c0v.z, c0v.w, c1v.x, c1v.y, c1v.z, c1v.w}, t3[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) { mh[k] = t0[k]; ml[k] = t1[k]; g[k] = t2[k]; b[k] = t3[k]; }
    }
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto item = [&](const float4 &x0, const float4 &x1, int64_t i) {
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)bn_apply1(x[k], mh[k], ml[k], g[k], b[k]);
        reinterpret_cast<bf16x8 *>(out)[i] = o;
    };
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n8; i += 2 * stride) {               // two items' loads in flight
        const float4 *p = reinterpret_cast<const float4 *>(v) + 2 * i, *q = reinterpret_cast<const float4 *>(v) + 2 * (i + stride);
        const float4 x0 = p[0], x1 = p[1], y0 = q[0], y1 = q[1];
        item(x0, x1, i);
        item(y0, y1, i + stride);
    }
    if (i < n8) {
        const float4 *p = reinterpret_cast<const float4 *>(v) + 2 * i;
        const float4 x0 = p[0], x1 = p[1];
        item(x0, x1, i);
    }
}

__global__ void f32_to_bf16_kernel(const float *__restrict__ v, int64_t total, __bf16 *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (__bf16)v[i];
}

// middle[i+1] = BN1(a1) * BN2(a2) (+ middle[i-2]), everything bf16 in HBM, 8 channels per thread (TilinGNN.py:64-71);
// st2 == NULL: a2 holds the collision branch's BatchNorm OUTPUT already (the forward stores it normalised)
__global__ __launch_bounds__(256) void merge_bf16_kernel(const __bf16 *__restrict__ a1, const float *__restrict__ st1,
                                                         const __bf16 *__restrict__ a2, const float *__restrict__ st2,
                                                         const __bf16 *__restrict__ resid, int64_t n8, int c,
                                                         __bf16 *__restrict__ out) {
    // c divides the grid stride (c <= 2048 = 8 x 256): a thread keeps its 8 columns; the two records go through LDS once
    // per block (64 scalar loads per thread in front of a ~2-item loop cost more than the loop: 20.6 us vs 8.6 for fp32)
    extern __shared__ __attribute__((aligned(16))) float rec[];          // [2][4][c]
    for (int k = threadIdx.x; k < 4 * c; k += blockDim.x) {
        rec[k] = st1[k];
        rec[4 * c + k] = st2 ? st2[k] : (k >= 2 * c && k < 3 * c ? 1.f : 0.f);
    }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int col = (int)((i0 * 8) % c);
    float m1h[8], m1l[8], g1[8], b1[8], m2h[8], m2l[8], g2[8], b2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cc = col + k;
        m1h[k] = rec[cc]; m1l[k] = rec[c + cc]; g1[k] = rec[2 * c + cc]; b1[k] = rec[3 * c + cc];
        m2h[k] = rec[4 * c + cc]; m2l[k] = rec[5 * c + cc]; g2[k] = rec[6 * c + cc]; b2[k] = rec[7 * c + cc];
    }
    for (int64_t i = i0; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x1[8], x2[8], r[8];
        unpack8(reinterpret_cast<const bf16x8 *>(a1)[i], x1);
        unpack8(reinterpret_cast<const bf16x8 *>(a2)[i], x2);
        if (resid) unpack8(reinterpret_cast<const bf16x8 *>(resid)[i], r);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float y2 = st2 ? bn_apply1(x2[k], m2h[k], m2l[k], g2[k], b2[k]) : x2[k];
            float v = bn_apply1(x1[k], m1h[k], m1l[k], g1[k], b1[k]) * y2;
            if (resid) v += r[k];
            o[k] = (__bf16)v;
        }
        reinterpret_cast<bf16x8 *>(out)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------ NNConv, width 64
// A-operand images of the per-type matrices: type t of layer l -> [M block mb 4][K chunk kc 2][lane (i, g)] x 8 bf16 with
// element e = W[k = 32 kc + 8 g + e][o = 16 mb + i] (wtab flat index k * 64 + o, NNConv's .view(-1, C_in, C_out)).
struct RootPtrs64 {
    const float *p[kMaxDepth];
};
__global__ __launch_bounds__(256) void nnconv64_image_kernel(const float *__restrict__ wtab_all, RootPtrs64 roots, int n_types,
                                                             __bf16 *__restrict__ wimg_all) {
    const int t = blockIdx.x, layer = blockIdx.y;
    const float *src = t < n_types ? wtab_all + ((int64_t)layer * n_types + t) * (kC * kC) : roots.p[layer];
    __bf16 *dst = wimg_all + ((int64_t)layer * (n_types + 1) + t) * (kW64Frag * 8);
    for (int r = threadIdx.x; r < kC * kC; r += 256) {
        const int k = r >> 6, o = r & 63;
        const int mb = o >> 4, i = o & 15, kc = k >> 5, g = (k >> 3) & 3, e = k & 7;
        dst[(((mb * 2 + kc) * 64) + g * 16 + i) * 8 + e] = (__bf16)src[r];
    }
}

constexpr int kMetaFirst = 1 << 8, kMetaEnd = 1 << 10, kMetaSkip = 1 << 11;    // (bit 9: last of a run, unused here)
constexpr int kStage64 = 16 * 20;

// Same column structure and the same walk as nnconv32_cols_kernel (one wave = a run of 16-row tiles, columns streamed
// in groups of four through a register pipeline); per column two 16-byte buffer loads per lane and 8 MFMAs.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 4) void nnconv64_bf16_cols_kernel(
    const __bf16 *__restrict__ h, uint32_t h_bytes, const int *__restrict__ tile_col_ptr, const int *__restrict__ col_meta,
    const int *__restrict__ col_src, const __bf16 *__restrict__ wimg, int n_types, const float *__restrict__ bias,
    int64_t n, int act, __bf16 *__restrict__ out, double *__restrict__ bn_partial, GinFin fin) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;                                        // [(T+1)][8 fragments][64 lanes] x 16 B
    float *bias_s = lds + (n_types + 1) * kW64Floats;       // [64]
    float *stage = bias_s + kC;                             // [WAVES][16][20]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    constexpr int kThreads = WAVES * 64;
    {
        const int n4 = (n_types + 1) * kW64Floats / 4;
        const float4 *src = reinterpret_cast<const float4 *>(wimg);
        for (int i = tid; i < n4; i += 4 * kThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[i + u * kThreads < n4 ? i + u * kThreads : n4 - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * kThreads < n4) reinterpret_cast<float4 *>(wl)[i + u * kThreads] = v[u];
        }
        if (tid < kC) bias_s[tid] = bias[tid];
    }
    float *stg = stage + wave * kStage64;

    static_assert(WAVES % 4 == 0, "whole SIMD quads");
    const int64_t n_tiles = (n + 15) / 16;
    const int nblk = gridDim.x;
    int64_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (int64_t)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int64_t slot = blk * 4 + (wave & 3), n_slots = (int64_t)nblk * 4;
    const int64_t q0 = n_tiles * slot / n_slots, q1 = n_tiles * (slot + 1) / n_slots;
    constexpr int kSubs = WAVES / 4;
    const int sub = wave >> 2;
    const int64_t t0 = q0 + (q1 - q0) * sub / kSubs, t1 = q0 + (q1 - q0) * (sub + 1) / kSubs;
    const int cbeg = __builtin_amdgcn_readfirstlane(tile_col_ptr[t0]);
    const int cend = __builtin_amdgcn_readfirstlane(tile_col_ptr[t1]);

    double bs[4] = {0, 0, 0, 0}, bq[4] = {0, 0, 0, 0};       // BN sums of channel 16 mb + fj over rows 4 fq .. 4 fq + 3
    __syncthreads();

    const __amdgpu_buffer_rsrc_t h_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(h), 0, (int)h_bytes, 0x00020000);
    auto load_group = [&](int p, int &s4, int &m4) {        // columns p .. p+3 (reads past cend stay inside the slack)
        const int pc = p < cend ? p : cbeg;
        s4 = col_src[(int64_t)pc * 16 + lane];
        m4 = col_meta[pc + (lane & 3)];
    };
    auto unpack = [&](int p, int u, int s4, int m4, int &s, int &m) {
        const bool ok = p + u < cend;                        // wave-uniform
        const int sv = __shfl(s4, u * 16 + fj, 64);
        const int mv = __builtin_amdgcn_readlane(m4, u);
        s = ok ? sv : -1;
        m = ok ? mv : kMetaSkip;
    };
    int64_t gtile = t0;
    auto issue_gather = [&](int s, int mu, u32x4 (&x)[2]) {
        const bool root = (mu & 0xff) == n_types && !(mu & kMetaSkip);
        const uint32_t row = root ? (uint32_t)(gtile * 16 + fj) : (uint32_t)s;
        const uint32_t off = s >= 0 ? row * 128u + (uint32_t)fq * 16u : 0x80000000u;   // s < 0: empty slot / row >= n
        x[0] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off, 0, 0);               // K chunk 0: channels 8 fq ..
        x[1] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off + 64u, 0, 0);         // K chunk 1: channels 32 + 8 fq ..
        if (root) ++gtile;
    };

    f32x4 d[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[8];                                            // W_type fragments of the current run: [mb][kc]
    int64_t ctile = t0;
    auto consume = [&](int s, int mu, const u32x4 (&x)[2]) {
        if (mu & kMetaSkip) return;
        const int t = mu & 0xff;
        const bool root = t == n_types;
        if (mu & kMetaFirst) {
            if (root) {
                // the edge part is complete: turn the sum into the mean before the root product joins the accumulators
                const float inv = s >= 0 ? 1.0f / __int_as_float(s) : 0.f;   // root column: max(deg, 1) in the source slot
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) d[mb] *= inv;
            }
            const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wl + t * kW64Floats) + lane;
#pragma unroll
            for (int f = 0; f < 8; ++f) wf[f] = wp[f * 64];
        }
        const bf16x8 x0 = __builtin_bit_cast(bf16x8, x[0]), x1 = __builtin_bit_cast(bf16x8, x[1]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            d[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * mb], x0, d[mb], 0, 0, 0);
            d[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * mb + 1], x1, d[mb], 0, 0, 0);
        }
        if (!(mu & kMetaEnd)) return;
        // ---- tile complete: lane (fj, fq) holds channels 16 mb + 4 fq + r of row fj
        const bool valid = s >= 0;
        const int64_t v = ctile * 16 + fj;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias_s + 16 * mb + 4 * fq);
            float o[4] = {d[mb][0] + b4.x, d[mb][1] + b4.y, d[mb][2] + b4.z, d[mb][3] + b4.w};
            bf16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (act == TGNN_ACT_LEAKY_RELU) o[r] = leakyf_(o[r]);
                ob[r] = (__bf16)o[r];
                o[r] = valid ? (float)ob[r] : 0.f;            // statistics of what is stored
            }
            if (valid) *reinterpret_cast<bf16x4 *>(out + v * kC + 16 * mb + 4 * fq) = ob;
            if (bn_partial) {
                *reinterpret_cast<float4 *>(stg + fj * 20 + 4 * fq) = make_float4(o[0], o[1], o[2], o[3]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double sum = 0, sq = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double val = (double)stg[(4 * fq + r) * 20 + fj];
                    sum += val;
                    sq += val * val;
                }
                bs[mb] += sum;                                // (mb is a compile-time constant after unrolling)
                bq[mb] += sq;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ++ctile;
    };

    int s4n, m4n;
    int xs[4], xm[4];
    u32x4 x[4][2];
    {
        int s4, m4;
        load_group(cbeg, s4, m4);
        load_group(cbeg + 4, s4n, m4n);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unpack(cbeg, u, s4, m4, xs[u], xm[u]);
            issue_gather(xs[u], xm[u], x[u]);
        }
    }
    for (int base = cbeg; base < cend; base += 4) {
        int s4c, m4c;
        load_group(base + 8, s4c, m4c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            consume(xs[u], xm[u], x[u]);
            unpack(base + 4, u, s4n, m4n, xs[u], xm[u]);
            issue_gather(xs[u], xm[u], x[u]);
        }
        s4n = s4c;
        m4n = m4c;
    }

    if (bn_partial) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds);       // [WAVES][64 lanes][8]
        double *mine = red + ((int64_t)wave * 64 + lane) * 8;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { mine[mb] = bs[mb]; mine[4 + mb] = bq[mb]; }
        __syncthreads();
        if (tid < 128) {                                     // tid = which * 64 + channel
            const int which = tid >> 6, ch = tid & 63, mb = ch >> 4, j = ch & 15;
            double acc = 0;
            for (int w = 0; w < WAVES; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += red[((int64_t)w * 64 + q * 16 + j) * 8 + which * 4 + mb];
            if (fin.counter) st_partial_sc1(bn_partial, (int64_t)blockIdx.x * 128 + tid, acc);
            else bn_partial[(int64_t)blockIdx.x * 128 + tid] = acc;
        }
        // [r4] the BatchNorm's record by the last block to finish instead of a 1-block launch behind the kernel (6.7 us on the chain)
        if (fin.counter) bn_fold_finish<kC>(fin, bn_partial, red + (size_t)WAVES * 64 * 8);
    }
}

static size_t nnconv64_lds_bytes(int n_types, int waves) {
    const size_t a = ((size_t)(n_types + 1) * kW64Floats + kC + (size_t)waves * kStage64) * sizeof(float);
    const size_t b = (size_t)waves * 64 * 8 * sizeof(double) + bn_fold_scratch_bytes(kC);
    return a > b ? a : b;
}
constexpr size_t kMaxLds64 = 160 * 1024 - 256;

static int launch_nnconv64(const __bf16 *h, int64_t n_src_rows, const int32_t *tile_col_ptr, const int32_t *col_meta,
                           const int32_t *col_src, const __bf16 *wimg, int32_t n_types, const float *bias, int64_t n_nodes,
                           int32_t act, __bf16 *out, double *bn_partial, int32_t *n_partials_host, hipStream_t s,
                           const GinFin *fin = nullptr) {
    constexpr int WAVES = 16;
    auto kern = nnconv64_bf16_cols_kernel<WAVES>;
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(kern, (int)kMaxLds64, site));
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t blocks = (n_tiles + 3) / 4;
    const int64_t cap = cus_minus(32);                       // (CUs left to the collision chain, as in the fp32 path)
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks &= ~(int64_t)7;
    if (blocks < 1) blocks = 1;
    GinFin f{};
    if (fin && bn_partial) {
        f = *fin;
        f.job.partials = bn_partial;
        f.job.n_partials = (int)blocks;
    }
    kern<<<(unsigned)blocks, WAVES * 64, nnconv64_lds_bytes(n_types, WAVES), s>>>(
        h, (uint32_t)(n_src_rows * 128), tile_col_ptr, col_meta, col_src, wimg, n_types, bias, n_nodes, act, out, bn_partial, f);
    if (n_partials_host) *n_partials_host = (int32_t)blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// ------------------------------------------------------------------------------------------ GIN, width 64
// z[v] = (1 + eps) f(a[v]) + sum_{src} f(a[src]),  f = the BatchNorm of the previous layer folded in (affine: commutes with
// the sum; in_stat == NULL: identity).  8 lanes x 16 bytes = one whole 128-byte row per 8 consecutive lanes (the cheapest
// gather shape on this chip: scratch/ubench/vmem3.hip), 32 rows per block, rows of an XCD contiguous.
__global__ __launch_bounds__(256) void gin64_bf16_aggregate_kernel(
    const __bf16 *__restrict__ a, const float *__restrict__ in_stat, const int *__restrict__ rowptr,
    const int *__restrict__ col_src, const float *__restrict__ eps_p, int64_t n, __bf16 *__restrict__ z) {
    const int tid = threadIdx.x, g = tid >> 3, q = tid & 7;
    const int xcd = blockIdx.x & 7, chunk = blockIdx.x >> 3;
    const int64_t r_beg = n * xcd / 8, r_end = n * (xcd + 1) / 8;
    const int64_t v = r_beg + (int64_t)chunk * 32 + g;
    if (v >= r_end) return;
    const float one_eps = 1.0f + eps_p[0];
    float mh[8], ml[8], gv[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = 8 * q + k;
        mh[k] = in_stat ? in_stat[c] : 0.f;
        ml[k] = in_stat ? in_stat[kC + c] : 0.f;
        gv[k] = in_stat ? in_stat[2 * kC + c] : 1.f;
        bv[k] = in_stat ? in_stat[3 * kC + c] : 0.f;
    }
    const int beg = rowptr[v], end = rowptr[v + 1];
    const bf16x8 *rows = reinterpret_cast<const bf16x8 *>(a);
    float self[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unpack8(rows[v * 8 + q], self);
    int e = beg;
    for (; e + 4 <= end; e += 4) {                           // 4 independent gathers in flight, summed in edge order
        const int s0 = col_src[e], s1 = col_src[e + 1], s2 = col_src[e + 2], s3 = col_src[e + 3];
        const bf16x8 p0 = rows[(int64_t)s0 * 8 + q], p1 = rows[(int64_t)s1 * 8 + q], p2 = rows[(int64_t)s2 * 8 + q],
                     p3 = rows[(int64_t)s3 * 8 + q];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc[k] += ((float)p0[k] - mh[k]) - ml[k];
            acc[k] += ((float)p1[k] - mh[k]) - ml[k];
            acc[k] += ((float)p2[k] - mh[k]) - ml[k];
            acc[k] += ((float)p3[k] - mh[k]) - ml[k];
        }
    }
    for (; e < end; ++e) {
        const bf16x8 p0 = rows[(int64_t)col_src[e] * 8 + q];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += ((float)p0[k] - mh[k]) - ml[k];
    }
    const float kb = one_eps + (float)(end - beg);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        o[k] = (__bf16)fmaf(gv[k], fmaf(one_eps, (self[k] - mh[k]) - ml[k], acc[k]), kb * bv[k]);
    reinterpret_cast<bf16x8 *>(z)[v * 8 + q] = o;
}

// GIN MLP 64 -> 32 -> 64 -> 64, sigmoid after every Linear, on bf16 MFMAs with the activations resident in registers: the
// transposed product H^T = W . Z^T leaves, in lane (n, q), features 16 mb + 4 q + r of row n; two M blocks are the 8
// values per lane of one K chunk of the next layer if that layer's K order is DEFINED as kf(q, e) (gin.hip).
//
// Precision: the collision branch is where 16-bit arithmetic would hurt -- its outputs are sigmoids whose columns vary by
// ~5e-3 around 0.5 over the nodes (measured on the labyrinth graph), and the BatchNorm behind them divides by that spread:
// a 2^-9 error before the BatchNorm is an O(1) error behind it (scratch/bf16_bn_probe.py: 1.2e-1 with one-plane weights and
// pre-BatchNorm bf16 storage).  So (1) weights and hidden activations are split into TWO bf16 pieces (hi + lo, 16+ bits;
// three cross terms, fp32 accumulation; the input z is bf16 already: two terms), and (2) what is STORED is the
// BatchNorm's OUTPUT (unit variance: bf16's relative precision is harmless there), which takes two passes over the MLP
// because the statistics are global: MODE 1 = statistics of the fp32 outputs only, MODE 2 = recompute, normalise with the
// finished record, round, store; MODE 3 [r4] = statistics AND the fp32 outputs kept in a scratch array, so that a light
// element-wise pass (bn_apply64_bf16_kernel: same formula on the same fp32 values, the same bits) replaces the second MLP pass
// (41 us per layer on what had become the forward's critical chain; the scratch costs 25.6 MB written + read per 100 000 nodes).  MODE 0 = the GINConv seam by itself (pre-BatchNorm output rounded to bf16 + its sums).
constexpr int kMlp64Waves = 8, kMlp64Threads = kMlp64Waves * 64;
__device__ __forceinline__ int kf64(int q, int e) { return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4); }
__device__ __forceinline__ void split2(const float (&x)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        hi[k] = (__bf16)x[k];
        lo[k] = (__bf16)(x[k] - (float)hi[k]);
    }
}

template <int MODE>
__global__ __launch_bounds__(kMlp64Threads, 2) void gin64_bf16_mlp_kernel(
    const __bf16 *__restrict__ z, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
    const float *__restrict__ b3, int64_t n, int act, const float *__restrict__ out_stat, __bf16 *__restrict__ out,
    double *__restrict__ bn_partial, float *__restrict__ out32 = nullptr, GinFin fin = GinFin{}) {
    __shared__ bf16x8 W1s[2][2 * 2 * 64];       // [plane hi / lo][M block 2][K chunk 2][lane]   K natural (Z comes from memory)
    __shared__ bf16x8 W2s[2][4 * 64];           // [plane][M block 4][lane]                      K = 32 in kf order
    __shared__ bf16x8 W3s[2][4 * 2 * 64];       // [plane][M block 4][K chunk 2][lane]           K = 64 in kf order per chunk
    __shared__ __attribute__((aligned(16))) float Bs[160];   // b1 (32) | b2 (64) | b3 (64)
    __shared__ __attribute__((aligned(16))) float St[4 * kC];
    __shared__ double red[kMlp64Waves * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fn = lane & 15, fq = lane >> 4;

    const int64_t n_tiles = (n + 15) / 16;
    const int nblk = gridDim.x;
    int64_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (int64_t)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int64_t slot = blk * 4 + (wave & 3), n_slots = (int64_t)nblk * 4;
    const int64_t q0 = n_tiles * slot / n_slots, q1 = n_tiles * (slot + 1) / n_slots;
    constexpr int kSubs = kMlp64Waves / 4;
    const int sub = wave >> 2;
    const int64_t t0 = q0 + (q1 - q0) * sub / kSubs, t1 = q0 + (q1 - q0) * (sub + 1) / kSubs;

    auto load_z = [&](int64_t tile, bf16x8 (&zin)[2]) {     // lane (n, q): channels 8 q .. and 32 + 8 q ..
        int64_t zr = tile * 16 + fn;
        zr = zr < n ? zr : n - 1;
        const bf16x8 *pz = reinterpret_cast<const bf16x8 *>(z + zr * kC);
        zin[0] = pz[fq];
        zin[1] = pz[4 + fq];
    };
    bf16x8 zin[2];
    load_z(t0 < t1 ? t0 : 0, zin);

    for (int i = tid; i < 2 * 2 * 64; i += kMlp64Threads) {  // item = (mb, kc, q, ii): lane index q * 16 + ii
        const int mb = i >> 7, kc = (i >> 6) & 1, q = (i >> 4) & 3, ii = i & 15;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w1[(16 * mb + ii) * 64 + 32 * kc + 8 * q + e];
        split2(x, W1s[0][i], W1s[1][i]);
    }
    for (int i = tid; i < 4 * 64; i += kMlp64Threads) {
        const int mb = i >> 6, q = (i >> 4) & 3, ii = i & 15;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w2[(16 * mb + ii) * 32 + kf64(q, e)];
        split2(x, W2s[0][i], W2s[1][i]);
    }
    for (int i = tid; i < 4 * 2 * 64; i += kMlp64Threads) {
        const int mb = i >> 7, kc = (i >> 6) & 1, q = (i >> 4) & 3, ii = i & 15;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w3[(16 * mb + ii) * 64 + 32 * kc + kf64(q, e)];
        split2(x, W3s[0][i], W3s[1][i]);
    }
    if (tid < 32) Bs[tid] = b1[tid];
    else if (tid < 96) Bs[tid] = b2[tid - 32];
    else if (tid < 160) Bs[tid] = b3[tid - 96];
    if (MODE == 2 && tid < 4 * kC) St[tid] = out_stat[tid];
    __syncthreads();

    auto bias4 = [&](int base, int mb) {
        const float4 t = *reinterpret_cast<const float4 *>(Bs + base + 16 * mb + 4 * fq);
        return f32x4{t.x, t.y, t.z, t.w};
    };
    // acc += (Wh + Wl) . (xh + xl), smallest terms first; xl == nullptr: x is exact in bf16
    auto mma = [&](const bf16x8 *wh, const bf16x8 *wl, int idx, const bf16x8 &xh, const bf16x8 *xl, f32x4 acc) {
        const bf16x8 a_h = wh[idx], a_l = wl[idx];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, xh, acc, 0, 0, 0);
        if (xl) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, *xl, acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, xh, acc, 0, 0, 0);
    };
    auto sig8 = [&](const f32x4 &a, const f32x4 &b, bf16x8 &hi, bf16x8 &lo) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = sigmoidf_(a[e]); x[4 + e] = sigmoidf_(b[e]); }
        split2(x, hi, lo);
    };
    double cs[16], cq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = cq[e] = 0.0;

    for (int64_t tile = t0; tile < t1; ++tile) {
        const bf16x8 za = zin[0], zb = zin[1];
        load_z(tile + 1 < t1 ? tile + 1 : tile, zin);
        // ---- layer 1: 2 M blocks x 2 K chunks, z exact
        // [r5] (its 8 weight fragments are re-read from LDS every tile: hoisted out of the loop with those of layers 2 and 3 they
        //  made 290 live registers, 34 of them in scratch memory and reloaded -- 13 scratch loads -- in every tile)
        const bf16x8 *w1h = W1s[0], *w1l = W1s[1];
        asm volatile("" : "+v"(w1h), "+v"(w1l));
        f32x4 h1[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            h1[mb] = mma(w1h, w1l, (mb * 2 + 0) * 64 + lane, za, nullptr, bias4(0, mb));
            h1[mb] = mma(w1h, w1l, (mb * 2 + 1) * 64 + lane, zb, nullptr, h1[mb]);
        }
        bf16x8 x1h, x1l;
        sig8(h1[0], h1[1], x1h, x1l);
        // ---- layer 2: 4 M blocks, K = 32
        f32x4 h2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) h2[mb] = mma(W2s[0], W2s[1], mb * 64 + lane, x1h, &x1l, bias4(32, mb));
        bf16x8 xah, xal, xbh, xbl;
        sig8(h2[0], h2[1], xah, xal);
        sig8(h2[2], h2[3], xbh, xbl);
        // ---- layer 3: 4 M blocks x 2 K chunks
        const int64_t row = tile * 16 + fn;
        const bool valid = row < n;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x4 o = mma(W3s[0], W3s[1], (mb * 2 + 0) * 64 + lane, xah, &xal, bias4(96, mb));
            o = mma(W3s[0], W3s[1], (mb * 2 + 1) * 64 + lane, xbh, &xbl, o);
            bf16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sigmoidf_(o[r]);
                if (act == TGNN_ACT_LEAKY_RELU) v = leakyf_(v);
                if (MODE == 2) {
                    const int c = 16 * mb + 4 * fq + r;
                    v = bn_apply1(v, St[c], St[kC + c], St[2 * kC + c], St[3 * kC + c]);
                }
                ob[r] = (__bf16)v;
                if (MODE == 3) o[r] = v;
                if (MODE != 2 && valid) {
                    const double dv = MODE == 0 ? (double)(float)ob[r] : (double)v;
                    cs[4 * mb + r] += dv;
                    cq[4 * mb + r] += dv * dv;
                }
            }
            if (MODE == 3) {
                if (valid) *reinterpret_cast<f32x4 *>(out32 + row * kC + 16 * mb + 4 * fq) = o;   // the fp32 rows the BatchNorm is applied to
            } else if (MODE != 1 && valid) {
                *reinterpret_cast<bf16x4 *>(out + row * kC + 16 * mb + 4 * fq) = ob;
            }
        }
    }
    if (MODE != 2 && bn_partial) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int dd = 1; dd <= 8; dd <<= 1) {
                cs[e] += __shfl_xor(cs[e], dd, 64);
                cq[e] += __shfl_xor(cq[e], dd, 64);
            }
        if (fn == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int feat = 16 * (e >> 2) + 4 * fq + (e & 3);
                red[wave * 128 + feat] = cs[e];
                red[wave * 128 + 64 + feat] = cq[e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            double tot = 0.0;
            for (int w = 0; w < kMlp64Waves; ++w) tot += red[w * 128 + tid];
            if (MODE == 3 && fin.counter) st_partial_sc1(bn_partial, (int64_t)blockIdx.x * 128 + tid, tot);
            else bn_partial[(int64_t)blockIdx.x * 128 + tid] = tot;
        }
        if (MODE == 3 && fin.counter) {                       // the record by the last block to finish (tgnn_common.h)
            __shared__ __attribute__((aligned(16))) unsigned char fold[bn_fold_scratch_bytes(kC)];
            bn_fold_finish<kC>(fin, bn_partial, reinterpret_cast<double *>(fold));
        }
    }
}

// ------------------------------------------------------------------------------------------ first final Linear
// out [N, M] (fp32) = act(cat . W^T + b), cat = the slot-major bf16 skip buffer [S][N][64] read in place (TilinGNN.py:74-76),
// W rounded to bf16 once (wb [M][K], K = 64 S).  Block = 8 waves, 128 rows x 256 outputs (every output column: the
// 269 MB of activations are read ONCE), one slot (64 k) per step staged through LDS with the next step's pieces already
// in registers; wave (wr, wc) owns 64 x 64 as 2 x 2 v_mfma_f32_32x32x16_bf16 tiles.  Persistent over row tiles: one
// BatchNorm partial row per block.
constexpr int kDbM = 128, kDbN = 256, kDbK = 64, kDbLd = kDbK + 8;   // LDS row pitch in bf16 (pad: 16 B)
constexpr int kDbThreads = 512;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(kDbThreads) void dense_bf16_slots_kernel(const __bf16 *__restrict__ a, int64_t slot_stride,
                                                                      int n_slots, const __bf16 *__restrict__ wb,
                                                                      const float *__restrict__ bias, int64_t n, int out_dim,
                                                                      int act, float *__restrict__ out,
                                                                      double *__restrict__ bn_partial) {
    __shared__ __attribute__((aligned(16))) __bf16 As[kDbM * kDbLd];
    __shared__ __attribute__((aligned(16))) __bf16 Bsm[kDbN * kDbLd];
    __shared__ double red[2][2][kDbN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, lg = lane >> 5;
    const int K = n_slots * kDbK;
    const int64_t row_tiles = (n + kDbM - 1) / kDbM;
    double csum[2] = {0.0, 0.0}, csq[2] = {0.0, 0.0};        // this lane's columns wc * 64 + j * 32 + li
    // pieces of 16 bytes this thread moves per step: A 128 x 8 = 2 per thread, B 256 x 8 = 4 per thread
    bf16x8 pa[2], pb[4];
    auto fetch = [&](int64_t rt, int sl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kDbThreads, r = i >> 3, p = i & 7;
            int64_t row = rt * kDbM + r;
            row = row < n ? row : n - 1;
            pa[u] = *reinterpret_cast<const bf16x8 *>(a + (int64_t)sl * slot_stride + row * kDbK + 8 * p);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * kDbThreads, r = i >> 3, p = i & 7;
            const int o = r < out_dim ? r : out_dim - 1;
            pb[u] = *reinterpret_cast<const bf16x8 *>(wb + (int64_t)o * K + sl * kDbK + 8 * p);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * kDbThreads, r = i >> 3, p = i & 7;
            *reinterpret_cast<bf16x8 *>(As + r * kDbLd + 8 * p) = pa[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * kDbThreads, r = i >> 3, p = i & 7;
            *reinterpret_cast<bf16x8 *>(Bsm + r * kDbLd + 8 * p) = pb[u];
        }
    };
    for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        fetch(rt, 0);
        for (int sl = 0; sl < n_slots; ++sl) {
            __syncthreads();                                  // everybody is done reading the previous step's tiles
            stash();
            __syncthreads();
            if (sl + 1 < n_slots) fetch(rt, sl + 1);          // in flight during the products
#pragma unroll
            for (int ks = 0; ks < kDbK / 16; ++ks) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i] = *reinterpret_cast<const bf16x8 *>(As + (wr * 64 + i * 32 + li) * kDbLd + ks * 16 + lg * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8 *>(Bsm + (wc * 64 + j * 32 + li) * kDbLd + ks * 16 + lg * 8);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        // epilogue: acc[i][j][e]: row = wr*64 + i*32 + (e&3) + 8*(e>>2) + 4*lg, col = wc*64 + j*32 + li
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wc * 64 + j * 32 + li;
            const float bcol = col < out_dim ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = rt * kDbM + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
                    const float v = act_apply(acc[i][j][e] + bcol, act);
                    if (row < n && col < out_dim) {
                        out[row * out_dim + col] = v;
                        csum[j] += (double)v;
                        csq[j] += (double)v * (double)v;
                    }
                }
        }
    }
    if (bn_partial) {
        // fold the two row halves (lg) of a wave, then the two row-waves (wr) of a column
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            csum[j] += __shfl_xor(csum[j], 32, 64);
            csq[j] += __shfl_xor(csq[j], 32, 64);
        }
        __syncthreads();
        if (lg == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                red[wr][0][wc * 64 + j * 32 + li] = csum[j];
                red[wr][1][wc * 64 + j * 32 + li] = csq[j];
            }
        }
        __syncthreads();
        if (tid < kDbN && tid < out_dim) {
            bn_partial[(int64_t)blockIdx.x * 2 * out_dim + tid] = red[0][0][tid] + red[1][0][tid];
            bn_partial[(int64_t)blockIdx.x * 2 * out_dim + out_dim + tid] = red[0][1][tid] + red[1][1][tid];
        }
    }
}

// ------------------------------------------------------------------------------------------ first final Linear, many rows
// [r4] The rows-per-wave form of dense.hip's dense_f16_rows_kernel for the bf16 skip buffer (no split: one matrix term): 4 waves x
// 32 rows x all 256 output columns; A fragments straight from memory (lane (i, g): 8 consecutive bf16 of row i), one slot (64 k)
// ahead; W as a bf16 operand image in fragment order ([slot][kb 4][tn 8][lane] x 16 B, dense_bf16_image_kernel) copied by DMA
// into a double-buffered 32 KB LDS tile, one barrier per slot.  k ascending per output as in dense_bf16_slots_kernel: same bits.
using u32x4_ = __attribute__((ext_vector_type(4))) unsigned int;
typedef __attribute__((address_space(3))) void lds_void64_t;
constexpr int kBrThreads = 256, kBrTileVec = 4 * 8 * 64, kBrRB = kBrTileVec / kBrThreads;
constexpr int64_t kBf16RowsKernelMin = 49152;

__global__ __launch_bounds__(256) void dense_bf16_image_kernel(const float *__restrict__ w, int n_slots, u32x4_ *__restrict__ wimg) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= n_slots * kBrTileVec) return;
    const int lane = item & 63, tn = (item >> 6) & 7, kb = (item >> 9) & 3, sl = item >> 11;
    const int col = tn * 32 + (lane & 31), k0 = sl * kDbK + kb * 16 + 8 * (lane >> 5);
    const float4 *p = reinterpret_cast<const float4 *>(w + (int64_t)col * n_slots * kDbK + k0);
    const float4 x0 = p[0], x1 = p[1];
    bf16x8 o;
    o[0] = (__bf16)x0.x; o[1] = (__bf16)x0.y; o[2] = (__bf16)x0.z; o[3] = (__bf16)x0.w;
    o[4] = (__bf16)x1.x; o[5] = (__bf16)x1.y; o[6] = (__bf16)x1.z; o[7] = (__bf16)x1.w;
    wimg[item] = __builtin_bit_cast(u32x4_, o);
}

__global__ __launch_bounds__(kBrThreads, 2) void dense_bf16_rows_kernel(const __bf16 *__restrict__ a, int64_t slot_stride, int n_slots,
                                                                       const u32x4_ *__restrict__ wimg, const float *__restrict__ bias,
                                                                       int64_t n, int act, float *__restrict__ out,
                                                                       double *__restrict__ bn_partial) {
    constexpr int N = 256, TN = 8, TNH = 4;
    __shared__ __attribute__((aligned(1024))) u32x4_ Bs0[kBrTileVec];
    __shared__ __attribute__((aligned(1024))) u32x4_ Bs1[kBrTileVec];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 31, fg = lane >> 5;
    const int64_t row_tiles = (n + 127) / 128;
    double bsum[2] = {0.0, 0.0};                              // entries tid and tid + 256 of the block's row [sum 256 | sum of squares 256]
    double *red = reinterpret_cast<double *>((n_slots & 1) ? Bs1 : Bs0);   // [wave][2][N]: the tile the last step does not read
    const bool leaky = act == TGNN_ACT_LEAKY_RELU;
    for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
        const int64_t m0 = rt * 128 + wave * 32;
        int64_t row = m0 + fi;
        row = row < n ? row : n - 1;
        const __bf16 *arow = a + row * kDbK + 8 * fg;
        auto load_a = [&](int sl, bf16x8 (&r)[4]) {
            const __bf16 *p = arow + (int64_t)sl * slot_stride;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) r[kb] = *reinterpret_cast<const bf16x8 *>(p + 16 * kb);
        };
        auto dma_b = [&](int sl, u32x4_ *dst) {
            const u32x4_ *p = wimg + (int64_t)sl * kBrTileVec + tid;
#pragma unroll
            for (int j = 0; j < kBrRB; ++j)
                __builtin_amdgcn_global_load_lds(p + kBrThreads * j, (lds_void64_t *)(dst + kBrThreads * j + wave * 64), 16, 0, 0);
        };
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
        bf16x8 abuf0[4], abuf1[4];
        load_a(0, abuf0);
        __syncthreads();                                      // the previous row tile's last reads of the tiles are over
        dma_b(0, Bs0);
        auto step = [&](int sl, const u32x4_ *bcur, u32x4_ *bnxt, const bf16x8 (&acur)[4], bf16x8 (&anxt)[4]) {
            __syncthreads();                                  // tile sl is complete (the barrier's fence waits for the copy), the other free
            if (sl + 1 < n_slots) {
                dma_b(sl + 1, bnxt);
                load_a(sl + 1, anxt);
            }
            const u32x4_ *bt = bcur + lane;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[kb], __builtin_bit_cast(bf16x8, bt[(kb * TN + tn) * 64]), acc[tn], 0, 0, 0);
        };
        for (int sl = 0; sl < n_slots; sl += 2) {
            step(sl, Bs0, Bs1, abuf0, abuf1);
            if (sl + 1 < n_slots) step(sl + 1, Bs1, Bs0, abuf1, abuf0);
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
                    const float u = acc[tn][r] + b;
                    const float v = leaky ? (u >= 0.f ? u : u * kLeakySlope) : act_apply(u, act);
                    out[orow * N + col] = v;
                    s_ += (double)v;
                    q_ += (double)v * (double)v;
                }
            }
            s_ += __shfl_xor(s_, 32, 64);
            q_ += __shfl_xor(q_, 32, 64);
            if (bn_partial && (tn / TNH) == fg) {
                red[(wave * 2 + 0) * N + col] = s_;
                red[(wave * 2 + 1) * N + col] = q_;
            }
        }
        if (bn_partial) {
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = tid + kBrThreads * h, which = i / N, cl = i % N;
                double tot = 0.0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) tot += red[(wv * 2 + which) * N + cl];
                bsum[h] += tot;
            }
        }
    }
    if (bn_partial) {
#pragma unroll
        for (int h = 0; h < 2; ++h) bn_partial[(int64_t)blockIdx.x * 2 * N + tid + kBrThreads * h] = bsum[h];
    }
}

// ------------------------------------------------------------------------------------------ workspace of the forward
struct Ws64 {
    __bf16 *mid, *a1, *a2[2], *z, *wimg, *wfin;
    float *t0, *ainit, *f1, *f2, *f3, *f4, *wtab, *pre32;
    unsigned *ctr;                                            // [0], [16]: the folded finalizes' tickets (bn_fold_finish)
    double *part1, *part2, *partf;
    float *stat1, *stat2[2], *stat_i[2], *stat_f[4];
    unsigned char *dimg[3];                                   // [r6] fp16-pair operand images of the final MLP's Linears 1 .. 3 (dense.hip)
    size_t bytes;
};
static Ws64 carve64(const tgnn_model_dims &d, int64_t n, int32_t n_types, void *ws, size_t ws_bytes) {
    Carver cv(ws, ws_bytes);
    const int D = d.network_depth;
    Ws64 w{};
    w.mid = cv.take<__bf16>((size_t)(D + 1) * n * kC);
    w.a1 = cv.take<__bf16>((size_t)n * kC);
    w.a2[0] = cv.take<__bf16>((size_t)n * kC);
    w.a2[1] = cv.take<__bf16>((size_t)n * kC);
    w.z = cv.take<__bf16>((size_t)n * kC);
    // [r6] the two type-dependent pieces sized for at least 16 types (what the kernels take): the layout then does not depend on the
    // type count, and tgnn_forward_bf16_begin can fill middle[0] before the layout is prepared
    const int tc = n_types < 16 ? 16 : n_types;
    w.wimg = cv.take<__bf16>((size_t)D * (tc + 1) * kC * kC);
    w.wfin = cv.take<__bf16>((size_t)256 * kC * (D + 1));
    w.pre32 = cv.take<float>((size_t)n * kC);
    w.ctr = cv.take<unsigned>(64);                           // ([32 .. 37]: the final MLP's bounds, dense_bounds_kernel)
    {
        const int fd[4] = {256, 128, 64, kC};
        for (int l = 0; l < 3; ++l) w.dimg[l] = cv.take<unsigned char>(dense_f16_image_size(fd[l], fd[l + 1]));
    }
    w.t0 = cv.take<float>((size_t)n * kC);
    w.ainit = cv.take<float>((size_t)n * kC);
    w.f1 = cv.take<float>((size_t)n * 256);
    w.f2 = cv.take<float>((size_t)n * 128);
    w.f3 = cv.take<float>((size_t)n * 64);
    w.f4 = cv.take<float>((size_t)n * kC);
    w.wtab = cv.take<float>((size_t)D * tc * kC * kC);
    w.part1 = cv.take<double>((size_t)TGNN_BN_MAX_PARTIALS * 2 * kC);
    w.part2 = cv.take<double>((size_t)TGNN_BN_MAX_PARTIALS * 2 * kC);
    w.partf = cv.take<double>((size_t)TGNN_BN_MAX_PARTIALS * 2 * 256);
    w.stat1 = cv.take<float>(4 * kC);
    w.stat2[0] = cv.take<float>(4 * kC);
    w.stat2[1] = cv.take<float>(4 * kC);
    w.stat_i[0] = cv.take<float>(4 * kC);
    w.stat_i[1] = cv.take<float>(4 * kC);
    for (int l = 0; l < 4; ++l) w.stat_f[l] = cv.take<float>(4 * 256);
    w.bytes = cv.off + 256;
    return w;
}

static inline unsigned ew_grid64(int64_t n, int cap = 256 * 8) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace tgnn

using namespace tgnn;

#define TGNN_TRY64(expr)                  \
    do {                                  \
        int rc__ = (expr);                \
        if (rc__ != TGNN_OK) return rc__; \
    } while (0)

// ------------------------------------------------------------------------------------------ C ABI: per-op entries (tests)
extern "C" int tgnn_f32_to_bf16(const float *src, int64_t count, void *dst, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (count <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(src && dst, "null pointer");
    f32_to_bf16_kernel<<<ew_grid64(count), 256, 0, static_cast<hipStream_t>(stream)>>>(src, count, static_cast<__bf16 *>(dst));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" size_t tgnn_nnconv64_image_elems(int32_t n_types) { return (size_t)(n_types + 1) * kC * kC; }

extern "C" int tgnn_nnconv64_bf16_fwd(const void *h_bf16, int64_t n_src_rows, const int32_t *tile_col_ptr,
                                      const int32_t *col_meta, const int32_t *col_src, const float *wtab, int32_t n_types,
                                      const float *root, const float *bias, int64_t n_nodes, int32_t act, void *out_bf16,
                                      void *wimg_scratch, double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_src_rows >= n_nodes && n_src_rows * 128 < (int64_t(1) << 31), "rows");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h_bf16 && tile_col_ptr && col_meta && col_src && root && bias && out_bf16 && wimg_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    if (nnconv64_lds_bytes(n_types, 16) > kMaxLds64) {
        set_error("tgnn_nnconv64_bf16_fwd: %d edge types do not fit the LDS weight image", n_types);
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    RootPtrs64 rp{};
    rp.p[0] = root;
    nnconv64_image_kernel<<<dim3(n_types + 1, 1), 256, 0, s>>>(wtab, rp, n_types, static_cast<__bf16 *>(wimg_scratch));
    return launch_nnconv64(static_cast<const __bf16 *>(h_bf16), n_src_rows, tile_col_ptr, col_meta, col_src,
                           static_cast<const __bf16 *>(wimg_scratch), n_types, bias, n_nodes, act,
                           static_cast<__bf16 *>(out_bf16), bn_partial, n_partials_host, s);
}

// ------------------------------------------------------------------------------------------ NNConv, width 64, on EDGE GROUPS
// The same op over the edge-group structure of nnconv_eg.hip (graph_prep.hip: nnconv_eg_kernel) instead of the type columns:
// a gather instruction fetches the sources of 16 in-edges of ONE type of a 16-row tile (15.6 groups per tile 68 % full at the
// benchmark layout against 34.8 columns 29 % full), the gathered rows ARE the A operand (bf16 storage: no split) of
//     M [16 edges x 64] = G [16 x 64] . W_t            8 x v_mfma_f32_16x16x32_bf16 (4 output blocks x 2 K chunks)
// and the messages, rounded to bf16 once (the operands and the result of this path are bf16: one more rounding of 2^-9, inside
// the stated 2^-7), are folded into their destination rows by
//     out [16 rows x 64] += S [16 rows x 16 edges] . M   S = the group's 0 / 1 selection matrix (4 x v_mfma_f32_16x16x16_bf16)
// The accumulator layout of the first product is the B-operand layout of the second.  The root group (last of a tile): the edge
// sum is first turned into the mean (rows' 1 / max(deg, 1)), then the root product joins with S = I.
struct Msg64 {
    u32x2 m[4];                                             // M[edges 4 fq + r][channel 16 mb + fj] as 4 bf16, mb = 0 .. 3
    bf16x4 sel;                                             // S[row fj][edges 4 fq .. 4 fq + 3]
};
constexpr int kEg64Root = 1 << 8;

template <int WAVES, int ACT>
__global__ __launch_bounds__(WAVES * 64, 4) void nnconv64_bf16_eg_kernel(
    const __bf16 *__restrict__ h, uint32_t h_bytes, const int *__restrict__ tile_grp_ptr, const int2 *__restrict__ grp,
    const __bf16 *__restrict__ wimg, int n_types, const float *__restrict__ bias, int64_t n, __bf16 *__restrict__ out,
    double *__restrict__ bn_partial, GinFin fin) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;                                        // [(T+1)][8 fragments][64 lanes] x 16 B
    float *bias_s = lds + (n_types + 1) * kW64Floats;       // [64]
    bf16x4 *lut = reinterpret_cast<bf16x4 *>(bias_s + kC);  // [16]: 4 selection bits -> 4 bf16 of 0 / 1
    float *stage = bias_s + kC + 32;                        // [WAVES][16][20]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;
    constexpr int kThreads = WAVES * 64;
    float *stg = stage + wave * kStage64;

    static_assert(WAVES % 4 == 0, "whole SIMD quads");
    const uint32_t n_tiles = (uint32_t)((n + 15) / 16);
    const uint32_t nblk = gridDim.x;
    uint32_t blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const uint32_t slot = blk * 4 + (wave & 3), n_slots = nblk * 4;
    const uint32_t tq = n_tiles / n_slots, tr = n_tiles % n_slots;
    const uint32_t q0 = tq * slot + tr * slot / n_slots, q1 = tq * (slot + 1) + tr * (slot + 1) / n_slots;
    constexpr uint32_t kSubs = WAVES / 4;
    const uint32_t sub = wave >> 2;
    const int64_t t0 = q0 + (q1 - q0) * sub / kSubs, t1 = q0 + (q1 - q0) * (sub + 1) / kSubs;
    const int cbeg = __builtin_amdgcn_readfirstlane(tile_grp_ptr[t0]);
    const int cend = __builtin_amdgcn_readfirstlane(tile_grp_ptr[t1]);

    double bs[4] = {0, 0, 0, 0}, bq[4] = {0, 0, 0, 0};       // BN sums of channel 16 mb + fj over rows 4 fq .. 4 fq + 3

    const __amdgpu_buffer_rsrc_t h_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(h), 0, (int)h_bytes, 0x00020000);
    const uint32_t fq_bytes = (uint32_t)fq * 16u;
    auto load_four = [&](int p, int &s4, int &m4) {         // groups p .. p+3: lane (fj, fq) <- group p + fq, word fj
        const int pc = p < cend ? p : cbeg;
        const int2 v = grp[(int64_t)pc * 16 + lane];
        s4 = v.x;
        m4 = v.y;
    };
    auto unpack = [&](auto steady, int p, int u, int s4, int m4, int &s, int &m, int &meta) {
        s = __shfl(s4, u * 16 + fj, 64);
        m = __shfl(m4, u * 16 + fj, 64);
        meta = __builtin_amdgcn_readlane(m4, u * 16) >> 16;
        if constexpr (!decltype(steady)::value)
            if (p + u >= cend) {                             // wave-uniform: past the share = an empty group of type 0
                s = -1;
                m = 0;
                meta = 0;
            }
    };
    int64_t gtile = t0;
    auto own_off_of = [&](int64_t tile) -> uint32_t {
        const int64_t r = tile * 16 + fj;
        return r < n ? (uint32_t)r * 128u + fq_bytes : 0x80000000u;
    };
    uint32_t own_off = own_off_of(gtile);
    auto issue_gather = [&](int s, int meta, u32x4 (&x)[2]) {
        const bool root = (meta & kEg64Root) != 0;           // wave-uniform
        const uint32_t off = root ? own_off : ((uint32_t)s << 7) + fq_bytes;   // s = -1: beyond the rows, loads zeros
        x[0] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off, 0, 0);       // K chunk 0: channels 8 fq ..
        x[1] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off + 64u, 0, 0); // K chunk 1: channels 32 + 8 fq ..
        if (root) {
            ++gtile;
            own_off = own_off_of(gtile);
        }
    };

    f32x4 d[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int64_t ctile = t0;
    auto stage1 = [&](int m, int meta, const u32x4 (&x)[2]) -> Msg64 {
        const int t = meta & 0xff;
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(wl + t * kW64Floats) + lane;
        Msg64 r;
        r.sel = lut[(m >> (4 * fq)) & 15];
        const bf16x8 x0 = __builtin_bit_cast(bf16x8, x[0]), x1 = __builtin_bit_cast(bf16x8, x[1]);
        // two output blocks at a time: all eight fragments in flight at once are 32 registers the gather pipeline needs
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            bf16x8 w[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) w[f] = wp[(4 * hb + f) * 64];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int mb = 2 * hb + k;
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w[2 * k], zero4, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w[2 * k + 1], acc, 0, 0, 0);
                const bf16x4 mv = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
                r.m[mb] = __builtin_bit_cast(u32x2, mv);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return r;
    };
    auto fold16 = [&](const Msg64 &g) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            d[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4s, g.sel), __builtin_bit_cast(bf16x4s, g.m[mb]), d[mb], 0, 0, 0);
    };
    // the edge sum of rows 4 fq + r -> their mean (s: float bits of max(deg, 1) of row fj, -1 = row >= n)
    auto to_mean = [&](int s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dg = __shfl(s, 4 * fq + r, 64);
            const float inv = dg >= 0 ? 1.0f / __int_as_float(dg) : 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) d[mb][r] *= inv;
        }
    };
    auto finish_tile = [&](int s) {
        // lane (fj, fq): channel 16 mb + fj of rows 4 fq + r -> through the wave's LDS tile -> row fj, channels 16 mb + 4 fq .. + 3
        bool rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = __shfl(s, 4 * fq + r, 64) >= 0;
        const bool valid = s >= 0;
        const int64_t v = ctile * 16 + fj;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float b = bias_s[16 * mb + fj];
            double sum = 0, sq = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = d[mb][r] + b;
                if constexpr (ACT == TGNN_ACT_LEAKY_RELU) o = leakyf_(o);
                o = bf16_round(o);                             // statistics of what is stored
                stg[(4 * fq + r) * 20 + fj] = o;
                if (rv[r]) {
                    sum += (double)o;
                    sq += (double)o * (double)o;
                }
            }
            bs[mb] += sum;
            bq[mb] += sq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float4 o4 = *reinterpret_cast<const float4 *>(stg + fj * 20 + 4 * fq);
            const bf16x4 ob = {(__bf16)o4.x, (__bf16)o4.y, (__bf16)o4.z, (__bf16)o4.w};   // (exact: rounded above)
            if (valid) *reinterpret_cast<bf16x4 *>(out + v * kC + 16 * mb + 4 * fq) = ob;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            d[mb] = zero4;
        }
        ++ctile;
    };
    // one group at a time (two groups per K = 32 fold, as the fp32 kernel does, keep two groups' messages and 16 operand
    // fragments alive: 128 registers do not hold them beside the gather pipeline)
    auto consume = [&](int s, int m, int meta, const u32x4 (&x)[2]) {
        const Msg64 g = stage1(m, meta, x);
        const bool root = (meta & kEg64Root) != 0;           // wave-uniform
        if (root) to_mean(s);
        fold16(g);
        if (root) finish_tile(s);
    };

    int s4n, m4n;
    int xs[4], xm[4], xt[4];
    u32x4 x[4][2];
    {
        int s4, m4;
        load_four(cbeg, s4, m4);
        load_four(cbeg + 4, s4n, m4n);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unpack(std::false_type{}, cbeg, u, s4, m4, xs[u], xm[u], xt[u]);
            issue_gather(xs[u], xt[u], x[u]);
        }
    }
    {   // (the first gathers are in flight: the block's weight image lands behind them)
        const int n4 = (n_types + 1) * kW64Floats / 4;
        const float4 *src = reinterpret_cast<const float4 *>(wimg);
        for (int i = tid; i < n4; i += 4 * kThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[i + u * kThreads < n4 ? i + u * kThreads : n4 - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * kThreads < n4) reinterpret_cast<float4 *>(wl)[i + u * kThreads] = v[u];
        }
        if (tid < kC) bias_s[tid] = bias[tid];
        if (tid < 16) {
            bf16x4 e;
#pragma unroll
            for (int b = 0; b < 4; ++b) e[b] = (tid >> b & 1) ? (__bf16)1.0f : (__bf16)0.0f;
            lut[tid] = e;
        }
    }
    __syncthreads();
    int base = cbeg;
    for (; base + 8 <= cend; base += 4) {                    // steady state: the four gathered in this round lies before cend
        int s4c, m4c;
        load_four(base + 8, s4c, m4c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            consume(xs[u], xm[u], xt[u], x[u]);
            unpack(std::true_type{}, base + 4, u, s4n, m4n, xs[u], xm[u], xt[u]);
            issue_gather(xs[u], xt[u], x[u]);
        }
        s4n = s4c;
        m4n = m4c;
    }
    for (; base < cend; base += 4) {
        int s4c, m4c;
        load_four(base + 8, s4c, m4c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            consume(xs[u], xm[u], xt[u], x[u]);
            unpack(std::false_type{}, base + 4, u, s4n, m4n, xs[u], xm[u], xt[u]);
            issue_gather(xs[u], xt[u], x[u]);
        }
        s4n = s4c;
        m4n = m4c;
    }

    if (bn_partial) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds);       // [WAVES][64 lanes][8]
        double *mine = red + ((int64_t)wave * 64 + lane) * 8;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { mine[mb] = bs[mb]; mine[4 + mb] = bq[mb]; }
        __syncthreads();
        if (tid < 128) {                                     // tid = which * 64 + channel
            const int which = tid >> 6, ch = tid & 63, mb = ch >> 4, j = ch & 15;
            double acc = 0;
            for (int w = 0; w < WAVES; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += red[((int64_t)w * 64 + q * 16 + j) * 8 + which * 4 + mb];
            if (fin.counter) st_partial_sc1(bn_partial, (int64_t)blockIdx.x * 128 + tid, acc);
            else bn_partial[(int64_t)blockIdx.x * 128 + tid] = acc;
        }
        if (fin.counter) bn_fold_finish<kC>(fin, bn_partial, red + (size_t)WAVES * 64 * 8);
    }
}

static size_t nnconv64_eg_lds_bytes(int n_types, int waves) {
    const size_t a = ((size_t)(n_types + 1) * kW64Floats + kC + 32 + (size_t)waves * kStage64) * sizeof(float);
    const size_t b = (size_t)waves * 64 * 8 * sizeof(double) + bn_fold_scratch_bytes(kC);
    return a > b ? a : b;
}

static int launch_nnconv64_eg(const __bf16 *h, int64_t n_src_rows, const int32_t *tile_grp_ptr, const int32_t *grp, const __bf16 *wimg,
                              int32_t n_types, const float *bias, int64_t n_nodes, int32_t act, __bf16 *out, double *bn_partial,
                              int32_t *n_partials_host, hipStream_t s, const GinFin *fin = nullptr) {
    constexpr int WAVES = 16;
    const bool leaky = act == TGNN_ACT_LEAKY_RELU;
    auto kern = leaky ? nnconv64_bf16_eg_kernel<WAVES, TGNN_ACT_LEAKY_RELU> : nnconv64_bf16_eg_kernel<WAVES, TGNN_ACT_NONE>;
    static LdsOptIn site[2];
    TGNN_CHECK_HIP(opt_in_dynamic_lds(kern, (int)kMaxLds64, site[leaky]));
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t blocks = (n_tiles + 3) / 4;
    const int64_t cap = cus_minus(32);                       // (CUs left to the collision chain, as in the fp32 path)
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks &= ~(int64_t)7;
    if (blocks < 1) blocks = 1;
    GinFin f{};
    if (fin && bn_partial) {
        f = *fin;
        f.job.partials = bn_partial;
        f.job.n_partials = (int)blocks;
    }
    kern<<<(unsigned)blocks, WAVES * 64, nnconv64_eg_lds_bytes(n_types, WAVES), s>>>(
        h, (uint32_t)(n_src_rows * 128), tile_grp_ptr, reinterpret_cast<const int2 *>(grp), wimg, n_types, bias, n_nodes, out, bn_partial, f);
    if (n_partials_host) *n_partials_host = (int32_t)blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_nnconv64_bf16_eg_fwd(const void *h_bf16, int64_t n_src_rows, const int32_t *tile_grp_ptr, const int32_t *grp,
                                         const float *wtab, int32_t n_types, const float *root, const float *bias, int64_t n_nodes,
                                         int32_t act, void *out_bf16, void *wimg_scratch, double *bn_partial,
                                         int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_src_rows >= n_nodes && n_src_rows * 128 < (int64_t(1) << 31), "rows");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h_bf16 && tile_grp_ptr && grp && root && bias && out_bf16 && wimg_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    if (nnconv64_eg_lds_bytes(n_types, 16) > kMaxLds64) {
        set_error("tgnn_nnconv64_bf16_eg_fwd: %d edge types do not fit the LDS weight image", n_types);
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    RootPtrs64 rp{};
    rp.p[0] = root;
    nnconv64_image_kernel<<<dim3(n_types + 1, 1), 256, 0, s>>>(wtab, rp, n_types, static_cast<__bf16 *>(wimg_scratch));
    return launch_nnconv64_eg(static_cast<const __bf16 *>(h_bf16), n_src_rows, tile_grp_ptr, grp, static_cast<const __bf16 *>(wimg_scratch),
                              n_types, bias, n_nodes, act, static_cast<__bf16 *>(out_bf16), bn_partial, n_partials_host, s);
}

static unsigned gin64_agg_blocks(int64_t n) {
    const int64_t rows_per_xcd = (n + 7) / 8;
    return (unsigned)(8 * ((rows_per_xcd + 31) / 32));
}
static int gin64_mlp_blocks(int64_t n) {
    int blocks = producer_blocks(n, 16 * kMlp64Waves);
    if (blocks > cus_minus(32)) blocks = cus_minus(32);
    if (blocks >= 8) blocks &= ~7;
    return blocks;
}
// the GINConv seam: aggregate + MLP, pre-BatchNorm output rounded to bf16, BatchNorm sums of the rounded values
static int gin64_launch(const __bf16 *a, const float *in_stat, const int32_t *rowptr, const int32_t *col_src, const float *eps,
                        const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                        int64_t n, int32_t act, __bf16 *out, __bf16 *z, double *bn_partial, int32_t *n_partials_host,
                        hipStream_t s) {
    gin64_bf16_aggregate_kernel<<<gin64_agg_blocks(n), 256, 0, s>>>(a, in_stat, rowptr, col_src, eps, n, z);
    const int blocks = gin64_mlp_blocks(n);
    gin64_bf16_mlp_kernel<0><<<blocks, kMlp64Threads, 0, s>>>(z, w1, b1, w2, b2, w3, b3, n, act, nullptr, out, bn_partial);
    if (n_partials_host) *n_partials_host = blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
// the CollConv seam (coll_conv.py:24-30): GINConv -> LeakyReLU -> BatchNorm1d (batch statistics), its OUTPUT stored as bf16
static int collconv64_launch(const __bf16 *h2_in, const int32_t *rowptr, const int32_t *col_src, const float *eps,
                             const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                             const float *b3, const BnJob &bn, int64_t n, int64_t n_total, float bn_eps, float momentum,
                             __bf16 *out, __bf16 *z, float *pre32, double *bn_partial, hipStream_t s, unsigned *fold_counter = nullptr) {
    gin64_bf16_aggregate_kernel<<<gin64_agg_blocks(n), 256, 0, s>>>(h2_in, nullptr, rowptr, col_src, eps, n, z);
    const int blocks = gin64_mlp_blocks(n);
    // one pass over the MLP: statistics + the fp32 rows (pre32); the BatchNorm is applied to those by an element-wise pass.
    // fold_counter (a zeroed device word): the record is written by the MLP kernel's last block, no finalize launch
    GinFin fin{};
    fin.counter = fold_counter;
    fin.job = bn;
    fin.job.partials = bn_partial;
    fin.job.n_partials = blocks;
    fin.n_total = n_total;
    fin.eps = bn_eps;
    fin.momentum = momentum;
    gin64_bf16_mlp_kernel<3><<<blocks, kMlp64Threads, 0, s>>>(z, w1, b1, w2, b2, w3, b3, n, TGNN_ACT_LEAKY_RELU, nullptr, nullptr,
                                                             bn_partial, pre32, fin);
    if (!fold_counter) {
        BnJobs jobs{};
        jobs.job[0] = fin.job;
        launch_bn_finalize(jobs, 1, 0, kC, n_total, bn_eps, momentum, s);
    }
    bn_apply64_bf16_kernel<<<ew_grid64(n * kC / 8, 512), 256, 0, s>>>(pre32, bn.stat, n * kC / 8, out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_gin64_bf16_fwd(const void *a_bf16, const float *in_stat, const int32_t *rowptr, const int32_t *col_src,
                                   const float *eps, const float *w1, const float *b1, const float *w2, const float *b2,
                                   const float *w3, const float *b3, int64_t n_nodes, int32_t act, void *out_bf16,
                                   void *z_scratch_bf16, double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1, "rows");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(a_bf16 && rowptr && eps && w1 && b1 && w2 && b2 && w3 && b3 && out_bf16 && z_scratch_bf16, "null pointer");
    return gin64_launch(static_cast<const __bf16 *>(a_bf16), in_stat, rowptr, col_src, eps, w1, b1, w2, b2, w3, b3, n_nodes, act,
                        static_cast<__bf16 *>(out_bf16), static_cast<__bf16 *>(z_scratch_bf16), bn_partial, n_partials_host,
                        static_cast<hipStream_t>(stream));
}

/* CollConv.forward (coll_conv.py:24-30) with its BatchNorm in train mode: out_bf16 = BN(LeakyReLU(GINConv(h2_in))) rounded
 * to bf16; stat_scratch: 4 x 64 floats; running statistics updated when given. */
extern "C" int tgnn_collconv64_bf16_fwd(const void *h2_in_bf16, const int32_t *rowptr, const int32_t *col_src, const float *eps,
                                        const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                                        const float *b3, const float *gamma, const float *beta, float *running_mean,
                                        float *running_var, int64_t *num_batches_tracked, int64_t n_nodes, void *out_bf16,
                                        void *z_scratch_bf16, float *pre_scratch_f32, float *stat_scratch, double *bn_partial,
                                        tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 2, "train-mode BatchNorm needs more than one row");
    TGNN_CHECK_ARG(h2_in_bf16 && rowptr && eps && w1 && b1 && w2 && b2 && w3 && b3 && gamma && beta && out_bf16 && z_scratch_bf16 &&
                       pre_scratch_f32 && stat_scratch && bn_partial, "null pointer");
    TGNN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running stats come in pairs");
    const BnJob bn{nullptr, 0, nullptr, gamma, beta, running_mean, running_var, num_batches_tracked, stat_scratch};
    return collconv64_launch(static_cast<const __bf16 *>(h2_in_bf16), rowptr, col_src, eps, w1, b1, w2, b2, w3, b3, bn, n_nodes,
                             n_nodes, 1e-5f, 0.1f, static_cast<__bf16 *>(out_bf16), static_cast<__bf16 *>(z_scratch_bf16), pre_scratch_f32,
                             bn_partial, static_cast<hipStream_t>(stream));
}

extern "C" int tgnn_merge_bf16_fwd(const void *a1, const float *stat1, const void *a2, const float *stat2, const void *resid,
                                   int64_t n_nodes, int32_t c, void *out, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0 && c >= 8 && c <= 2048 && (2048 % c) == 0, "width must divide 2048 (8 columns per thread)");
    if (n_nodes == 0) return TGNN_OK;
    TGNN_CHECK_ARG(a1 && stat1 && a2 && out, "null pointer");       /* stat2 == NULL: a2 is normalised already */
    const int64_t n8 = n_nodes * c / 8;
    merge_bf16_kernel<<<ew_grid64(n8), 256, (size_t)8 * c * sizeof(float), static_cast<hipStream_t>(stream)>>>(
        static_cast<const __bf16 *>(a1), stat1, static_cast<const __bf16 *>(a2), stat2, static_cast<const __bf16 *>(resid), n8, c,
        static_cast<__bf16 *>(out));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

static int dense_bf16_slots_launch(const __bf16 *a, int64_t slot_stride, int n_slots, const __bf16 *wb, const float *bias,
                                   int64_t n, int out_dim, int act, float *out, double *bn_partial, int32_t *n_partials_host,
                                   hipStream_t s) {
    int64_t blocks = (n + kDbM - 1) / kDbM;
    if (blocks > TGNN_BN_MAX_PARTIALS) blocks = TGNN_BN_MAX_PARTIALS;
    if (blocks < 1) blocks = 1;
    dense_bf16_slots_kernel<<<(unsigned)blocks, kDbThreads, 0, s>>>(a, slot_stride, n_slots, wb, bias, n, out_dim, act, out,
                                                                    bn_partial);
    if (n_partials_host) *n_partials_host = (int32_t)blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// fp32 W -> bf16 (row-major for the block-tile kernel, fragment order for the rows kernel) into wb_scratch, then the product.
// Measured at 100 000 rows x 21 slots -> 256: block-tile kernel 150 us, rows kernel 125 us (inside the config-3 forward; its A rows are one slot ahead only -- the barrier drains them: latency, not issue, bounds it).
static int dense_bf16_slots_from_f32(const __bf16 *a, int64_t slot_stride, int n_slots, const float *w, const float *bias, int64_t n,
                                     int out_dim, int act, float *out, void *wb_scratch, double *bn_partial, int32_t *n_partials_host,
                                     hipStream_t s) {
    if (out_dim == 256 && n >= kBf16RowsKernelMin && ((uintptr_t)w % 16) == 0 && ((uintptr_t)wb_scratch % 16) == 0) {
        const int items = n_slots * kBrTileVec;
        dense_bf16_image_kernel<<<(items + 255) / 256, 256, 0, s>>>(w, n_slots, static_cast<u32x4_ *>(wb_scratch));
        int64_t blocks = (n + 127) / 128;
        const int64_t cap = 2 * (int64_t)device_cus();
        if (blocks > cap) blocks = cap;
        if (blocks > TGNN_BN_MAX_PARTIALS) blocks = TGNN_BN_MAX_PARTIALS;
        dense_bf16_rows_kernel<<<(unsigned)blocks, kBrThreads, 0, s>>>(a, slot_stride, n_slots, static_cast<const u32x4_ *>(wb_scratch), bias,
                                                                      n, act, out, bn_partial);
        if (n_partials_host) *n_partials_host = (int32_t)blocks;
        TGNN_CHECK_LAUNCH();
        return TGNN_OK;
    }
    const int64_t wn = (int64_t)out_dim * kDbK * n_slots;
    f32_to_bf16_kernel<<<ew_grid64(wn), 256, 0, s>>>(w, wn, static_cast<__bf16 *>(wb_scratch));
    return dense_bf16_slots_launch(a, slot_stride, n_slots, static_cast<const __bf16 *>(wb_scratch), bias, n, out_dim, act, out, bn_partial,
                                   n_partials_host, s);
}

/* act(cat . w^T + b) over the slot-major bf16 skip buffer [n_slots][n_rows][64]; w fp32 [out_dim][64 n_slots] is rounded to
 * bf16 into wb_scratch (out_dim * 64 * n_slots bf16) first. */
extern "C" int tgnn_dense_bf16_slots_fwd(const void *a_bf16, int64_t slot_stride, int32_t n_slots, const float *w,
                                         const float *b, int64_t n_rows, int32_t out_dim, int32_t act, float *out,
                                         void *wb_scratch, double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_rows >= 1 && n_slots >= 1 && out_dim >= 1 && out_dim <= 256, "shape");
    TGNN_CHECK_ARG(a_bf16 && w && b && out && wb_scratch && slot_stride >= n_rows * kC, "null pointer / stride");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dense_bf16_slots_from_f32(static_cast<const __bf16 *>(a_bf16), slot_stride, n_slots, w, b, n_rows, out_dim, act, out, wb_scratch,
                                     bn_partial, n_partials_host, s);
}

// ------------------------------------------------------------------------------------------ the whole forward
extern "C" size_t tgnn_forward_bf16_workspace_bytes(const tgnn_model_dims *dims, int64_t n_nodes, int32_t n_types) {
    if (!dims || dims->network_width != kC || n_nodes < 0) return 0;
    return carve64(*dims, n_nodes, n_types, nullptr, 0).bytes;
}

// ---- init MLP (TilinGNN.py:54; fp32 products on the existing dense kernels), middle[0] stored as bf16; everything on `stream`
static int init_mlp64(const tgnn_model_dims *dims, const Params &P, const float *x, const Ws64 &w, int64_t n, int update_running,
                      tgnn_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int fx = dims->node_features_dim;
    int32_t np1 = 0;
    auto finalize1 = [&](const BnPtrs &b, float *stat) {
        BnJobs jobs{};
        jobs.job[0] = BnJob{w.partf, np1, nullptr, b.gamma, b.beta, update_running ? b.rm : nullptr, update_running ? b.rv : nullptr,
                            update_running ? b.nbt : nullptr, stat};
        launch_bn_finalize(jobs, 1, 0, kC, n, 1e-5f, 0.1f, s);
    };
    TGNN_TRY64(tgnn_dense_act_fwd(x, fx, 32, nullptr, P.f(P.init(0)), P.f(P.init(0) + 1), n, fx, kC, TGNN_ACT_LEAKY_RELU, w.t0, kC,
                                  w.partf, &np1, stream));
    finalize1(P.bn(P.init(0) + 2), w.stat_i[0]);
    TGNN_TRY64(tgnn_dense_act_fwd(w.t0, kC, 32, w.stat_i[0], P.f(P.init(1)), P.f(P.init(1) + 1), n, kC, kC, TGNN_ACT_LEAKY_RELU,
                                  w.ainit, kC, w.partf, &np1, stream));
    finalize1(P.bn(P.init(1) + 2), w.stat_i[1]);
    bn_apply_bf16_kernel<<<ew_grid64(n * kC), 256, 0, s>>>(w.ainit, w.stat_i[1], n, kC, w.mid);
    return TGNN_OK;
}

// [r6] the init MLP of a NEW layout's forward on stream2, BEFORE / beside the layout's preparation (it needs nothing of the graph;
// the workspace's layout does not depend on the type count): the next tgnn_forward_bf16 of this thread with the same workspace and
// node count waits for the event and skips its own (its running-statistics update has been applied here)
struct Head64 {
    hipEvent_t ev = nullptr;
    const void *ws = nullptr;
    int64_t n = 0;
};
static thread_local Head64 g_head64[64];

extern "C" int tgnn_forward_bf16_begin(const tgnn_model_dims *dims, const void *const *params_host, const float *x, int64_t n_nodes,
                                       int32_t update_running, void *ws, size_t ws_bytes, tgnn_stream_t stream2) {
    DeviceGuard guard__(stream2);
    TGNN_CHECK_ARG(dims && dims->network_width == kC && dims->network_depth >= 1 && dims->network_depth <= kMaxDepth && params_host && x &&
                   n_nodes >= 2 && stream2, "arguments");
    Ws64 w = carve64(*dims, n_nodes, 0, ws, ws_bytes);
    if (!ws || w.bytes > ws_bytes) {
        set_error("tgnn_forward_bf16_begin: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return TGNN_ERR_WORKSPACE;
    }
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
    Head64 &he = g_head64[dev];
    if (!he.ev) TGNN_CHECK_HIP(hipEventCreateWithFlags(&he.ev, hipEventDisableTiming));
    const Params P{params_host, dims->network_depth};
    TGNN_TRY64(init_mlp64(dims, P, x, w, n_nodes, update_running, stream2));
    TGNN_CHECK_HIP(hipEventRecord(he.ev, static_cast<hipStream_t>(stream2)));
    he.ws = ws;
    he.n = n_nodes;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* TilinGNN.forward (TilinGNN.py:51-78) at network_width 64 with bf16 activation storage; same parameter table, graph
 * structure and BatchNorm semantics as tgnn_forward.  Train mode only (the mode the reference runs inference in). */
extern "C" int tgnn_forward_bf16(const tgnn_model_dims *dims, const void *const *params_host, const float *x,
                                 const float *adj_edge_attr, const tgnn_graph *graph, int32_t update_running, float *probs,
                                 void *ws, size_t ws_bytes, tgnn_stream_t stream, tgnn_stream_t stream2) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(dims && dims->network_width == kC && dims->network_depth >= 1 && dims->network_depth <= kMaxDepth,
                   "the bf16-storage path is built for network_width 64");
    TGNN_CHECK_ARG(params_host && graph && probs && x, "null pointer");
    const int64_t n = graph->n_nodes;
    TGNN_CHECK_ARG(n >= 2, "train-mode BatchNorm needs more than one row");
    const bool have_cols = graph->nn_tile_col_ptr && graph->nn_col_meta && graph->nn_col_src;
    const bool eg = graph->nn_tile_grp_ptr && graph->nn_grp && tgnn_set_nnconv_eg(-1) != 0;   // (edge groups: nnconv64_bf16_eg_kernel)
    TGNN_CHECK_ARG(graph->adj_rowptr && graph->col_rowptr && (have_cols || eg),
                   "graph pointers (the NNConv column structure or the edge groups are required)");
    TGNN_CHECK_ARG(n * 128 < (int64_t(1) << 31), "rows must lie within 2 GB");
    const int T = graph->n_types, D = dims->network_depth, fe = dims->adj_edge_features_dim;
    TGNN_CHECK_ARG(T == 0 || (adj_edge_attr && graph->type_rep_edge), "adjacency pointers");
    if ((eg ? nnconv64_eg_lds_bytes(T, 16) : nnconv64_lds_bytes(T, 16)) > kMaxLds64) {
        set_error("tgnn_forward_bf16: %d edge types do not fit the LDS weight image", T);
        return TGNN_ERR_UNSUPPORTED;
    }
    Ws64 w = carve64(*dims, n, T, ws, ws_bytes);
    if (!ws || w.bytes > ws_bytes) {
        set_error("tgnn_forward_bf16: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // two-chain schedule as in tgnn_forward: the collision branch is a chain of its own (CollConv_i reads only CollConv_{i-1},
    // TilinGNN.py:63) and runs on stream2 beside the adjacency branch; the chains meet in merge only
    hipStream_t s2 = static_cast<hipStream_t>(stream2);
    if (s2 == s) s2 = nullptr;
    constexpr int kEv = 3 + 2 * kMaxDepth;        // [0] middle[0] done, [1 + i] CollConv_i done, [1 + kMaxDepth + i] merge_i done, then fork / edge weights
    constexpr int kEvFork64 = 1 + 2 * kMaxDepth, kEvWeights64 = 2 + 2 * kMaxDepth;
    static thread_local hipEvent_t ev_cache[64][kEv] = {};
    hipEvent_t *ev = nullptr;
    if (s2) {
        int dev = 0;
        TGNN_CHECK_HIP(hipGetDevice(&dev));
        TGNN_CHECK_ARG(dev >= 0 && dev < 64, "device index");
        if (!ev_cache[dev][0])
            for (int k = 0; k < kEv; ++k) TGNN_CHECK_HIP(hipEventCreateWithFlags(&ev_cache[dev][k], hipEventDisableTiming));
        ev = ev_cache[dev];
    }
    const Params P{params_host, D};
    const float eps = 1e-5f, momentum = 0.1f;
    int32_t np1 = 0;
    auto bn_job = [&](double *part, int nparts, const BnPtrs &bp, float *stat) {
        return BnJob{part, nparts, nullptr, bp.gamma, bp.beta, update_running ? bp.rm : nullptr, update_running ? bp.rv : nullptr,
                     update_running ? bp.nbt : nullptr, stat};
    };
    auto finalize1 = [&](double *part, int nparts, int f, const BnPtrs &b, float *stat) {
        BnJobs jobs{};
        jobs.job[0] = bn_job(part, nparts, b, stat);
        launch_bn_finalize(jobs, 1, 0, f, n, eps, momentum, s);
    };
    // ---- init MLP: tgnn_forward_bf16_begin's (this thread, this workspace and node count), or queued below
    bool head_picked = false;
    {
        int devh = 0;
        TGNN_CHECK_HIP(hipGetDevice(&devh));
        Head64 *he = devh >= 0 && devh < 64 ? &g_head64[devh] : nullptr;
        if (he && he->ws && he->ws == ws && he->n == n) {
            head_picked = true;
            TGNN_CHECK_HIP(hipStreamWaitEvent(s, he->ev, 0));
        }
        if (he) he->ws = nullptr;
    }
    // ---- per-type NNConv matrices of all layers (fp32 table, then the bf16 operand images)
    // [r6] on the side stream, beside the init MLP (the first NNConv waits for them; ~40 us of the head were serial) -- or, with the
    // init MLP done by tgnn_forward_bf16_begin, on `stream` itself, straight behind the layout's preparation and BEHIND the release
    // of the collision chain: no cross-queue hand-over in front of them or in front of the first NNConv (as tgnn_forward_resume)
    hipStream_t sw = s;
    auto queue_weights = [&]() {
        if (T > 0) {
            EdgeMlpLayers layers{};
            for (int i = 0; i < D; ++i) {
                const int b = P.layer(i);
                layers.l[i] = EdgeMlpLayer{P.f(b), P.f(b + 1), P.f(b + 2), P.f(b + 3), P.f(b + 4), P.f(b + 5)};
            }
            launch_edge_weight_table_batched(adj_edge_attr, graph->type_rep_edge, T, fe, layers, D, kC, w.wtab, nullptr, nullptr, sw);
        }
        RootPtrs64 rp{};
        for (int i = 0; i < D; ++i) rp.p[i] = P.f(P.layer(i) + 6);
        nnconv64_image_kernel<<<dim3(T + 1, D), 256, 0, sw>>>(w.wtab, rp, T, w.wimg);
    };
    const bool weights_on_main = head_picked && s2 != nullptr;
    if (!weights_on_main) {
#ifndef TGNN_ABL_C3SERIALHEAD
        if (s2)
#else
        if (false)
#endif
        {
            TGNN_CHECK_HIP(hipEventRecord(ev[kEvFork64], s));    // what the caller queued on `stream` so far (x, the parameters, the layout)
            TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[kEvFork64], 0));
            sw = s2;
        }
        queue_weights();
        if (sw != s) TGNN_CHECK_HIP(hipEventRecord(ev[kEvWeights64], s2));
        if (!head_picked) TGNN_TRY64(init_mlp64(dims, P, x, w, n, update_running, stream));
    }
    // ---- main loop.  a2[i & 1] holds h2_i = the collision branch's BatchNorm OUTPUT (stored normalised, see the MLP kernel)
    TGNN_CHECK_HIP(hipMemsetAsync(w.ctr, 0, 64 * sizeof(unsigned), s));
    if (s2) {
        TGNN_CHECK_HIP(hipEventRecord(ev[0], s));
        TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[0], 0));
    }
    if (weights_on_main) queue_weights();                    // (sw == s)
    hipStream_t sc = s2 ? s2 : s;
    // [r6] the final MLP's Linears 1 .. 3 (256 -> 128 -> 64 -> 64, BatchNorm on load) on the fp32 path's fp16-pair kernels with W
    // resident in LDS (dense.hip: dense_f16_resident_kernel; bf16 x 3 block-tile kernels before: 84 + 31 + 31 us at 100 000 rows):
    // the weights' bounds, the inputs' bounds from the producers' BatchNorm parameters and the operand images, in front of the
    // collision chain (every merge joins it, so the final MLP finds them done)
    const int fdim[5] = {0, 256, 128, 64, kC};
    bool tail_f16 = n >= kDenseRowsKernelMin && tgnn_set_split_precision(-1) != 0;
    if (tail_f16) {
        const float *bw[3], *bg[3], *bb[3];
        int64_t bwn[3];
        int bf[3], iin[3], iout[3];
        unsigned *bwm[3], *bam[3];
        const unsigned *iwm[3];
        void *iimg[3];
        for (int l = 1; l <= 3; ++l) {
            const BnPtrs bp = P.bn(P.fin(l - 1) + 2);
            bw[l - 1] = P.f(P.fin(l));
            bwn[l - 1] = (int64_t)fdim[l] * fdim[l + 1];
            bg[l - 1] = bp.gamma; bb[l - 1] = bp.beta; bf[l - 1] = fdim[l];
            bwm[l - 1] = w.ctr + 32 + 2 * (l - 1);
            bam[l - 1] = w.ctr + 33 + 2 * (l - 1);
            iin[l - 1] = fdim[l]; iout[l - 1] = fdim[l + 1]; iwm[l - 1] = bwm[l - 1]; iimg[l - 1] = w.dimg[l - 1];
        }
        launch_dense_bounds(3, bw, bwn, bg, bb, bf, bwm, bam, n, sc);
        tail_f16 = dense_f16_images_build(3, bw, iin, iout, iwm, iimg, sc) == TGNN_OK;
    }
    if (sw != s) TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[kEvWeights64], 0));
    for (int i = 0; i < D; ++i) {
        const int b = P.layer(i);
        const __bf16 *h1 = w.mid + (size_t)i * n * kC;
        const __bf16 *h2_in = i == 0 ? w.mid : w.a2[(i - 1) & 1];
        if (s2 && i >= 2) TGNN_CHECK_HIP(hipStreamWaitEvent(s2, ev[1 + kMaxDepth + i - 2], 0));   // a2[i & 1] was read by merge_{i-2}
        TGNN_TRY64(collconv64_launch(h2_in, graph->col_rowptr, graph->col_src, P.f(b + 13), P.f(b + 14), P.f(b + 15), P.f(b + 16),
                                     P.f(b + 17), P.f(b + 18), P.f(b + 19), bn_job(nullptr, 0, P.bn(b + 20), w.stat2[i & 1]), n, n,
                                     eps, momentum, w.a2[i & 1], w.z, w.pre32, w.part2, sc, w.ctr));
        if (s2) TGNN_CHECK_HIP(hipEventRecord(ev[1 + i], s2));
        GinFin fin1{};
        fin1.counter = w.ctr + 16;
        fin1.job = bn_job(nullptr, 0, P.bn(b + 8), w.stat1);
        fin1.n_total = n;
        fin1.eps = eps;
        fin1.momentum = momentum;
        if (eg)
            TGNN_TRY64(launch_nnconv64_eg(h1, n, graph->nn_tile_grp_ptr, graph->nn_grp, w.wimg + (size_t)i * (T + 1) * kC * kC, T,
                                          P.f(b + 7), n, TGNN_ACT_LEAKY_RELU, w.a1, w.part1, &np1, s, &fin1));
        else
            TGNN_TRY64(launch_nnconv64(h1, n, graph->nn_tile_col_ptr, graph->nn_col_meta, graph->nn_col_src,
                                       w.wimg + (size_t)i * (T + 1) * kC * kC, T, P.f(b + 7), n, TGNN_ACT_LEAKY_RELU, w.a1, w.part1,
                                       &np1, s, &fin1));
        if (s2) TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev[1 + i], 0));
        const __bf16 *resid = i >= 2 ? w.mid + (size_t)(i - 2) * n * kC : nullptr;
        merge_bf16_kernel<<<ew_grid64(n * kC / 8), 256, 8 * kC * sizeof(float), s>>>(w.a1, w.stat1, w.a2[i & 1], nullptr, resid, n * kC / 8, kC,
                                                                w.mid + (size_t)(i + 1) * n * kC);
        if (s2) TGNN_CHECK_HIP(hipEventRecord(ev[1 + kMaxDepth + i], s));
    }
    // ---- final MLP: the first Linear reads the bf16 skip buffer in place, the rest runs on the fp32 dense kernels
    {
        const int pi = P.fin(0);
        TGNN_TRY64(dense_bf16_slots_from_f32(w.mid, (int64_t)n * kC, D + 1, P.f(pi), P.f(pi + 1), n, 256, TGNN_ACT_LEAKY_RELU, w.f1, w.wfin,
                                             w.partf, &np1, s));
        finalize1(w.partf, np1, 256, P.bn(pi + 2), w.stat_f[0]);
    }
    float *fbuf[4] = {w.f1, w.f2, w.f3, w.f4};
    for (int l = 1; l < 4; ++l) {
        const int pi = P.fin(l);
        if (tail_f16)
            TGNN_TRY64(dense_act_bounded(fbuf[l - 1], fdim[l], 32, w.stat_f[l - 1], P.f(pi), P.f(pi + 1), n, fdim[l], fdim[l + 1],
                                         TGNN_ACT_LEAKY_RELU, fbuf[l], fdim[l + 1], w.partf, &np1, w.ctr + 33 + 2 * (l - 1), 1,
                                         w.ctr + 32 + 2 * (l - 1), s, w.dimg[l - 1], nullptr, nullptr, nullptr));
        else
            TGNN_TRY64(tgnn_dense_act_fwd(fbuf[l - 1], fdim[l], 32, w.stat_f[l - 1], P.f(pi), P.f(pi + 1), n, fdim[l], fdim[l + 1],
                                          TGNN_ACT_LEAKY_RELU, fbuf[l], fdim[l + 1], w.partf, &np1, stream));
        finalize1(w.partf, np1, fdim[l + 1], P.bn(pi + 2), w.stat_f[l]);
    }
    TGNN_TRY64(tgnn_dense_act_fwd(fbuf[3], kC, 32, w.stat_f[3], P.f(P.last()), P.f(P.last() + 1), n, kC, dims->output_dim,
                                  TGNN_ACT_SIGMOID, probs, dims->output_dim, nullptr, nullptr, stream));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}
