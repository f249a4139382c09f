// NNConv(aggr="mean") for TilinGNN's adjacency branch on gfx950 (K1-K4 of SURVEY.md section 2b).
//
// Reference semantics (GraphConv.forward, /root/reference/graph_networks/layers/edge_conv.py:24-27,
// calling PyG 1.3.2 NNConv): every edge e carries a [C,C] matrix W_e = edge_mlp(edge_attr_e);
//   out[v] = mean_{e: dst_e = v} h[src_e] . W_e  +  h[v] . root + bias          (+ LeakyReLU)
// The reference materialises all E matrices (4 KB each, ~83 % of its forward time).  Edge
// attributes come from a small codebook, so W_e = wtab[type_e] with T types (13 on real data):
//   * tgnn_edge_weight_table evaluates the edge MLP on the T representative rows only;
//   * tgnn_nnconv_mean_fwd keeps the T matrices (+ root as type T) resident in LDS and walks a
//     destination-sorted CSR: one wavefront per destination row, no atomics, sums in the original
//     edge order.
//
// nnconv_lds_kernel<32> lane mapping (wave64 = 4 DPP rows of 16 lanes):
//     lane = 32*p + 16*s + c      p: which of two in-flight items (edges) of the row
//                                 s: which half of the input channels, i in [16s, 16s+16)
//                                 c: output channels c and c+16
//   each lane loads ONE float of h[src] (h[src][16s+c]: the two rows of a pair read one 128-B line)
//   and gets the 16 inputs of its half by DPP row broadcast (no LDS traffic for activations);
//   the weights come from LDS as 8 x ds_read_b128 per item pair.  LDS image per type:
//   [s][o][20] floats (16 used): bank(20*o) hits 16 distinct 4-bank slots, so every ds_read_b128
//   lane group is conflict free.
#include "tgnn_common.h"

namespace tgnn {

// ------------------------------------------------------------------------------------------
// K1 on the T representative rows
// ------------------------------------------------------------------------------------------
constexpr int kEH1 = 32, kEH2 = 64;  // edge-MLP hidden sizes (edge_conv.py:9)

// grid = (T, depth): blockIdx.y selects the layer
// (root matrices of all layers, by value: the weight image treats the root as pseudo-type T)
struct RootPtrs {
    const float *p[kMaxDepth];
};

// One weight -> its slot of the matrix-core kernel's operand image (nnconv_cols.hip): element (k, o) of a type's [32][32]
// matrix, flat index r = k * 32 + o (NNConv's .view(-1, C_in, C_out)), split exactly into three bf16 pieces (hi + mid + lo)
//   -> plane p, [M block o >> 4][g = k >> 3][i = o & 15][k & 7]
__device__ __forceinline__ void weight_image_put(__bf16 *dst, int r, float x) {
    const int k = r >> 5, o = r & 31;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;                       // exact
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;                      // exact
    const int at = (((o >> 4) * 4 + (k >> 3)) * 16 + (o & 15)) * 8 + (k & 7);
    dst[0 * kWtPlane * 2 + at] = h;                      // (kWtPlane floats = 2 kWtPlane bf16)
    dst[1 * kWtPlane * 2 + at] = m;
    dst[2 * kWtPlane * 2 + at] = (__bf16)r2;
}

// the fp16-pair image's slot: the same fragment order, two planes (hi, lo) of the weight times the layer's power of two
__device__ __forceinline__ void weight_image_put_f16(_Float16 *dst, int r, float xs) {
    const int k = r >> 5, o = r & 31;
    const _Float16 h = (_Float16)xs;
    const int at = (((o >> 4) * 4 + (k >> 3)) * 16 + (o & 15)) * 8 + (k & 7);
    dst[at] = h;
    dst[kWtPlane * 2 + at] = (_Float16)(xs - (float)h);
}

__global__ __launch_bounds__(256) void edge_weight_table_kernel(
    const float *__restrict__ edge_attr, const int *__restrict__ type_rep_edge, int fe, EdgeMlpLayers layers, int cc,
    float *__restrict__ wtab_all, int n_types, RootPtrs roots, float *__restrict__ wimg_all, unsigned *__restrict__ done_ctr,
    const unsigned *__restrict__ root_max, float img_scale) {
    // wimg_all != NULL (width 32): the block also writes its type's slice of the matrix-core operand image, and one more
    // block per layer (blockIdx.x == n_types) the root matrix's -- no second launch on the way to the first NNConv
    // root_max != NULL: fp16-pair images (two planes, scaled), else bf16 x 3
    const bool f16 = root_max != nullptr;
    const float wscale = f16 ? nnconv_weight_scale(root_max[blockIdx.y]) * img_scale : 1.0f;   // (img_scale: a power of two)
    const int64_t img_at = ((int64_t)blockIdx.y * (n_types + 1) + blockIdx.x) * (f16 ? kWtTypeF16 : kWtType);
    __bf16 *img = wimg_all ? reinterpret_cast<__bf16 *>(wimg_all + img_at) : nullptr;
    _Float16 *img16 = reinterpret_cast<_Float16 *>(img);
    if ((int)blockIdx.x == n_types) {
        const float *src = roots.p[blockIdx.y];
        if (f16) for (int r = threadIdx.x; r < 1024; r += 256) weight_image_put_f16(img16, r, src[r] * wscale);
        else for (int r = threadIdx.x; r < 1024; r += 256) weight_image_put(img, r, src[r]);
        if (done_ctr) {                                     // (a consumer on another stream counts the finished blocks)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's stores are in L2 before thread 0 writes L2 back
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const EdgeMlpLayer L = layers.l[blockIdx.y];
    const float *__restrict__ w1 = L.w1, *__restrict__ b1 = L.b1, *__restrict__ w2 = L.w2, *__restrict__ b2 = L.b2,
                *__restrict__ w3 = L.w3, *__restrict__ b3 = L.b3;
    float *__restrict__ wtab = wtab_all + (int64_t)blockIdx.y * n_types * cc;
    __shared__ float e_s[1024];
    __shared__ float h1_s[kEH1];
    __shared__ float h2_s[kEH2];
    const int t = blockIdx.x;
    const int64_t row = type_rep_edge[t];
    for (int k = threadIdx.x; k < fe; k += blockDim.x) e_s[k] = edge_attr[row * fe + k];
    __syncthreads();
    if (threadIdx.x < kEH1) {
        float acc = b1[threadIdx.x];
        for (int k = 0; k < fe; ++k) acc = fmaf(e_s[k], w1[threadIdx.x * fe + k], acc);
        h1_s[threadIdx.x] = sigmoidf_(acc);
    }
    __syncthreads();
    if (threadIdx.x < kEH2) {
        float acc = b2[threadIdx.x];
#pragma unroll
        for (int k = 0; k < kEH1; ++k) acc = fmaf(h1_s[k], w2[threadIdx.x * kEH1 + k], acc);
        h2_s[threadIdx.x] = sigmoidf_(acc);
    }
    __syncthreads();
    // L3: [cc, 64] row-major.  One wave per output row chunk: lane = k, shuffle-reduce would cost
    // more than it saves at T ~ 13; each thread owns outputs j, j+256, ... and reads its row as
    // 16 x float4 (rows are 256-B aligned).
    for (int j = threadIdx.x; j < cc; j += blockDim.x) {
        const float4 *wr = reinterpret_cast<const float4 *>(w3 + (int64_t)j * kEH2);
        float acc = b3[j];
#pragma unroll
        for (int q = 0; q < kEH2 / 4; ++q) {
            const float4 w = wr[q];
            acc = fmaf(h2_s[4 * q + 0], w.x, acc);
            acc = fmaf(h2_s[4 * q + 1], w.y, acc);
            acc = fmaf(h2_s[4 * q + 2], w.z, acc);
            acc = fmaf(h2_s[4 * q + 3], w.w, acc);
        }
        const float v = sigmoidf_(acc);
        wtab[(int64_t)t * cc + j] = v;
        if (img) {
            if (f16) weight_image_put_f16(img16, j, v * wscale);
            else weight_image_put(img, j, v);
        }
    }
    if (done_ctr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------
// NNConv mean, C = 32, weight table in LDS
// ------------------------------------------------------------------------------------------
constexpr int kNNThreads = 1024;           // 16 waves = 4 per SIMD; one block per CU, one LDS image per CU
constexpr int kNNWaves = kNNThreads / 64;
constexpr int kTypeStride = 2 * 32 * 20;   // floats per type in LDS

template <int K>
__device__ __forceinline__ float row_bcast(float x) {  // lane K of every 16-lane DPP row -> whole row
#ifdef TGNN_ABLATE_NO_DPP
    return x;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xf, 0xf, true));
#endif
}

#ifdef TGNN_ABLATE_NO_LDSW
#define TGNN_LOADW(Q) const float4 wa = make_float4(bias0, bias1, bias0, bias1), wb = make_float4(bias1, bias0, bias1, bias0);
#else
#define TGNN_LOADW(Q) const float4 wa = wlo[Q], wb = whi[Q];
#endif
#define TGNN_FMA4(Q, XA, XB, XC, XD)                          \
    {                                                         \
        TGNN_LOADW(Q)                                         \
        m0 = fmaf(XA, wa.x, m0); m1 = fmaf(XA, wb.x, m1);     \
        m0 = fmaf(XB, wa.y, m0); m1 = fmaf(XB, wb.y, m1);     \
        m0 = fmaf(XC, wa.z, m0); m1 = fmaf(XC, wb.z, m1);     \
        m0 = fmaf(XD, wa.w, m0); m1 = fmaf(XD, wb.w, m1);     \
    }

// One item = one in-edge (or the root pseudo-edge) of the current row, for this lane's stream p.
#ifdef TGNN_ABLATE_NO_FMA
#define TGNN_ITEM(X, T, SCALE) { acc0 = fmaf((SCALE), (X), acc0); acc1 += (float)(T); }
#else
#define TGNN_ITEM(X, T, SCALE)                                                                        \
    {                                                                                                 \
        const float4 *wlo = reinterpret_cast<const float4 *>(lds + (T) * kTypeStride + lane_w_off);   \
        const float4 *whi = wlo + 80; /* +16 outputs * 20 floats */                                   \
        float m0 = 0.f, m1 = 0.f;                                                                     \
        const float x_ = (X);                                                                         \
        TGNN_FMA4(0, row_bcast<0>(x_), row_bcast<1>(x_), row_bcast<2>(x_), row_bcast<3>(x_))          \
        TGNN_FMA4(1, row_bcast<4>(x_), row_bcast<5>(x_), row_bcast<6>(x_), row_bcast<7>(x_))          \
        TGNN_FMA4(2, row_bcast<8>(x_), row_bcast<9>(x_), row_bcast<10>(x_), row_bcast<11>(x_))        \
        TGNN_FMA4(3, row_bcast<12>(x_), row_bcast<13>(x_), row_bcast<14>(x_), row_bcast<15>(x_))      \
        acc0 = fmaf((SCALE), m0, acc0);                                                               \
        acc1 = fmaf((SCALE), m1, acc1);                                                               \
    }
#endif

constexpr int kSlots = 8;  // items per stream held in registers: rows with in-degree <= 15 are fully pipelined

__global__ __launch_bounds__(kNNThreads) void nnconv32_lds_kernel(
    const float *__restrict__ h, int64_t ldh, const int *__restrict__ rowptr, const int *__restrict__ col_src,
    const int *__restrict__ col_type, const float *__restrict__ wtab, int n_types, const float *__restrict__ root,
    const float *__restrict__ bias, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- weight table (+ root as pseudo-type T) into LDS, transposed to [t][s][o][20]
    // kNNThreads == 1024: thread r owns element r = i*32+o of every type; types are independent
    // loads, issued 4 at a time.
    {
        const int r = threadIdx.x & 1023, i = r >> 5, o = r & 31;
        const int dst = (i >> 4) * 640 + o * 20 + (i & 15);
        int t = 0;
        for (; t + 4 <= n_types; t += 4) {
            const float v0 = wtab[(t + 0) * 1024 + r], v1 = wtab[(t + 1) * 1024 + r];
            const float v2 = wtab[(t + 2) * 1024 + r], v3 = wtab[(t + 3) * 1024 + r];
            lds[(t + 0) * kTypeStride + dst] = v0;
            lds[(t + 1) * kTypeStride + dst] = v1;
            lds[(t + 2) * kTypeStride + dst] = v2;
            lds[(t + 3) * kTypeStride + dst] = v3;
        }
        for (; t < n_types; ++t) lds[t * kTypeStride + dst] = wtab[t * 1024 + r];
        lds[n_types * kTypeStride + dst] = root[r];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane >> 5, s = (lane >> 4) & 1, c = lane & 15;
    const float bias0 = bias[c], bias1 = bias[c + 16];
    const int lane_w_off = s * 640 + c * 20;  // + t*kTypeStride; second accumulator: + 16*20
    const int lane_x_off = 16 * s + c;

    // XCD-aware row ownership: block b runs on XCD b % 8 (observed placement, speed only): give
    // every XCD one contiguous node range so that its private L2 holds that range's activations.
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, bl = blockIdx.x >> 3;
    int64_t r_beg = 0, r_end = n;
    int wl = blockIdx.x * kNNWaves + wave, wcount = nblk * kNNWaves;
    if (nblk >= 8) {
        r_beg = n * xcd / 8;
        r_end = n * (xcd + 1) / 8;
        wl = bl * kNNWaves + wave;
        wcount = (nblk >> 3) * kNNWaves;
    }

    double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;  // BN partials of this lane's two columns

    // Software pipeline over this wave's rows (each stage one dependent-load level ahead):
    //   A: rowptr of row v+3   B: col_src/col_type of row v+2   C: h[src] of row v+1   D: FMAs of row v
    int beg_a = 0, deg_a = -1;                         // stage A result (row v+3 -> consumed by B)
    int src_b[kSlots], typ_b[kSlots], deg_b = -1, beg_b = 0;      // stage B result (row v+2 -> consumed by C)
    float x_c[kSlots]; int typ_c[kSlots], deg_c = -1, beg_c = 0;  // stage C result (row v+1 -> consumed by D)
#pragma unroll
    for (int k = 0; k < kSlots; ++k) { src_b[k] = -1; typ_b[k] = n_types; x_c[k] = 0.f; typ_c[k] = n_types; }

    const int64_t v0 = r_beg + wl;
    for (int64_t v = v0 - 3 * (int64_t)wcount; v < r_end; v += wcount) {
        // ---------------- stage D operands are the registers filled by stage C one iteration ago
        float x_d[kSlots]; int typ_d[kSlots];
        const int deg_d = deg_c, beg_d = beg_c;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) { x_d[k] = x_c[k]; typ_d[k] = typ_c[k]; }
        // ---------------- stage C: gathers of row v+1 (indices loaded by stage B one iteration ago).
        // Loads are unconditional (invalid slots read row 0 and are zeroed by a select): no branches,
        // so all of a stage's loads issue back to back and stay in flight across the FMA block.
        deg_c = deg_b; beg_c = beg_b;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            typ_c[k] = typ_b[k];
            const int sidx = src_b[k] >= 0 ? src_b[k] : 0;
            const float xv = h[(int64_t)sidx * ldh + lane_x_off];
            x_c[k] = src_b[k] >= 0 ? xv : 0.f;
        }
        // ---------------- stage B: edge indices of row v+2 (rowptr loaded by stage A one iteration ago)
        const int64_t vb = v + 2 * (int64_t)wcount;
        deg_b = deg_a; beg_b = beg_a;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const int j = p + 2 * k;
            const bool is_edge = j < deg_b;
            const int eidx = is_edge ? beg_b + j : 0;          // slot 0 of the CSR arrays always exists
            const int sv = col_src[eidx], tv = col_type[eidx];
            src_b[k] = is_edge ? sv : (j == deg_b ? (int)vb : -1);   // j == deg: root pseudo-edge reads h[v]
            typ_b[k] = is_edge ? tv : n_types;
        }
        // ---------------- stage A: rowptr of row v+3
        const int64_t va = v + 3 * (int64_t)wcount;
        {
            const bool ok = va >= v0 && va < r_end;
            const int64_t vv = ok ? va : 0;
            const int b0 = rowptr[vv], b1 = rowptr[vv + 1];
            beg_a = b0;
            deg_a = ok ? b1 - b0 : -1;
        }
        // ---------------- stage D: compute row v
        if (deg_d < 0) continue;                        // pipeline fill / drain (wave-uniform)
        const float inv_deg = 1.0f / (float)(deg_d > 0 ? deg_d : 1);
        float acc0 = 0.f, acc1 = 0.f;
        const int n_items = deg_d + 1;
        const int n_iter = (n_items + 1) >> 1;          // iterations of the widest stream
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            if (k < n_iter) {                           // wave-uniform
                const int j = p + 2 * k;
                const float scale = j < deg_d ? inv_deg : (j == deg_d ? 1.0f : 0.0f);
                TGNN_ITEM(x_d[k], typ_d[k], scale)
            }
        }
        // rows with more than 2*kSlots-1 in-edges: remaining items, not pipelined (rare on tile graphs)
        for (int j = p + 2 * kSlots; j <= deg_d; j += 2) {
            const bool is_edge = j < deg_d;
            const int src = is_edge ? col_src[beg_d + j] : (int)v;
            const int t = is_edge ? col_type[beg_d + j] : n_types;
            const float scale = is_edge ? inv_deg : 1.0f;
            TGNN_ITEM(h[(int64_t)src * ldh + lane_x_off], t, scale)
        }
        // combine the two input halves (lanes ^16) and the two item streams (lanes ^32)
        acc0 += __shfl_xor(acc0, 16, 64);
        acc1 += __shfl_xor(acc1, 16, 64);
        acc0 += __shfl_xor(acc0, 32, 64);
        acc1 += __shfl_xor(acc1, 32, 64);
        float o0 = acc0 + bias0, o1 = acc1 + bias1;
        if (act == TGNN_ACT_LEAKY_RELU) {
            o0 = leakyf_(o0);
            o1 = leakyf_(o1);
        }
        if (lane < 16) {
            out[v * 32 + c] = o0;
            out[v * 32 + 16 + c] = o1;
            s0 += (double)o0; q0 += (double)o0 * (double)o0;
            s1 += (double)o1; q1 += (double)o1 * (double)o1;
        }
    }

    if (bn_partial) {
        __syncthreads();  // every wave is done with the weight image: reuse the front of LDS
        double *red = reinterpret_cast<double *>(lds);
        if (lane < 16) {
            double *r = red + (wave * 16 + c) * 4;
            r[0] = s0; r[1] = s1; r[2] = q0; r[3] = q1;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int q = threadIdx.x >> 4, cc = threadIdx.x & 15;
            double tot = 0.0;
            for (int w = 0; w < kNNWaves; ++w) tot += red[(w * 16 + cc) * 4 + q];
            // partial row layout [2][32]: sum then sumsq; q = 0: sum col c, 1: sum col c+16, 2/3: sumsq
            bn_partial[(int64_t)blockIdx.x * 64 + (q >> 1) * 32 + (q & 1) * 16 + cc] = tot;
        }
    }
}
#undef TGNN_ITEM
#undef TGNN_FMA4

// ------------------------------------------------------------------------------------------
// Weight image of the matrix-core kernel (nnconv_cols.hip), built once per forward for all layers.
// ------------------------------------------------------------------------------------------
// wtab [T][32][32] (+ root [32][32] as pseudo-type T) -> MFMA operand image, every weight split exactly into three
// bf16 pieces (hi + mid + lo):  element (k, o) of type t -> plane p, M block o >> 4, row o & 15, group k >> 3,
// element k & 7.  grid = (T+1, layers); done once per forward so that every NNConv block fills its LDS with a
// straight coalesced 16-byte copy.
__global__ __launch_bounds__(256) void nnconv_weight_image_kernel(const float *__restrict__ wtab_all, RootPtrs roots,
                                                                  int n_types, float *__restrict__ wimg_all,
                                                                  const unsigned *__restrict__ root_max, float img_scale) {
    const int t = blockIdx.x, layer = blockIdx.y;
    const float *src = t < n_types ? wtab_all + ((int64_t)layer * n_types + t) * 1024 : roots.p[layer];
    if (root_max) {                                           // fp16-pair image
        const float wscale = nnconv_weight_scale(root_max[layer]) * img_scale;
        _Float16 *dst = reinterpret_cast<_Float16 *>(wimg_all + ((int64_t)layer * (n_types + 1) + t) * kWtTypeF16);
        for (int r = threadIdx.x; r < 1024; r += 256) weight_image_put_f16(dst, r, src[r] * wscale);
        return;
    }
    __bf16 *dst = reinterpret_cast<__bf16 *>(wimg_all + ((int64_t)layer * (n_types + 1) + t) * kWtType);
    for (int r = threadIdx.x; r < 1024; r += 256) weight_image_put(dst, r, src[r]);
}

// ------------------------------------------------------------------------------------------
// Generic fallback (any C, any T): one thread per (row, output); weights read through L2.
// Used when the table does not fit LDS or C != 32.  Correct, not fast.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nnconv_generic_kernel(
    const float *__restrict__ h, int64_t ldh, const int *__restrict__ rowptr, const int *__restrict__ col_src,
    const int *__restrict__ col_type, const float *__restrict__ wtab, const float *__restrict__ root,
    const float *__restrict__ bias, int64_t n, int c, int act, float *__restrict__ out,
    double *__restrict__ bn_partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [rows_per_block][c] staging of h rows
    double *red = reinterpret_cast<double *>(lds);
    const int rows_per_block = blockDim.x / c;
    const int r_local = threadIdx.x / c, o = threadIdx.x % c;
    const bool active_thread = r_local < rows_per_block;
    double s = 0.0, q = 0.0;
    for (int64_t base = (int64_t)blockIdx.x * rows_per_block; base < n; base += (int64_t)gridDim.x * rows_per_block) {
        const int64_t v = base + r_local;
        if (active_thread && v < n) {
            const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
            float acc = 0.f;
            for (int j = 0; j < deg; ++j) {
                const float *x = h + (int64_t)col_src[beg + j] * ldh;
                const float *w = wtab + (int64_t)col_type[beg + j] * c * c + o;
                float m = 0.f;
                for (int i = 0; i < c; ++i) m = fmaf(x[i], w[(int64_t)i * c], m);
                acc += m;
            }
            acc *= 1.0f / (float)(deg > 0 ? deg : 1);
            const float *x = h + v * ldh;
            float m = 0.f;
            for (int i = 0; i < c; ++i) m = fmaf(x[i], root[(int64_t)i * c + o], m);
            float r = acc + m + bias[o];
            if (act == TGNN_ACT_LEAKY_RELU) r = leakyf_(r);
            out[v * c + o] = r;
            s += (double)r;
            q += (double)r * (double)r;
        }
    }
    if (bn_partial) {
        __syncthreads();
        if (active_thread) {
            red[threadIdx.x * 2] = s;
            red[threadIdx.x * 2 + 1] = q;
        }
        __syncthreads();
        if ((int)threadIdx.x < c) {
            double ts = 0.0, tq = 0.0;
            for (int r = 0; r < rows_per_block; ++r) {
                ts += red[(r * c + threadIdx.x) * 2];
                tq += red[(r * c + threadIdx.x) * 2 + 1];
            }
            bn_partial[(int64_t)blockIdx.x * 2 * c + threadIdx.x] = ts;
            bn_partial[(int64_t)blockIdx.x * 2 * c + c + threadIdx.x] = tq;
        }
    }
}

// [r5] The same table with the blocks laid over (layer, 256 OUTPUTS) instead of (layer, type): a thread owns one of the cc outputs of
// the last Linear -- its row of W3 (64 floats) sits in registers ONCE for all types -- and the two small hidden layers of every type
// are recomputed by each of the layer's cc / 256 blocks (T x 2 528 multiply-adds: nothing).  With a block per (layer, type) every
// block streamed the whole 256 KB W3 out of L2 for one type: 280 blocks x 256 KB and ~29 us on the side stream at 13 types x 20
// layers -- during which the persistent layer loops (forward_small.hip, forward_mid.hip), started beside it, found no free CUs for
// half of their blocks.  Same multiply-add order per output as edge_weight_table_kernel: the same bits.
constexpr int kEwTypes = 16, kEwMaxFe = 64;   // types per pass; attribute columns (beyond: the kernel above)
__global__ __launch_bounds__(256) void edge_weight_table_chunks_kernel(
    const float *__restrict__ edge_attr, const int *__restrict__ type_rep_edge, int fe, EdgeMlpLayers layers, int cc,
    float *__restrict__ wtab_all, int n_types, RootPtrs roots, float *__restrict__ wimg_all, unsigned *__restrict__ done_ctr,
    const unsigned *__restrict__ root_max, float img_scale, const int *__restrict__ n_types_dev, int max_types_dev) {
    // [r6] n_types_dev: the type count read on the device (tgnn_forward_begin_weights: queued before the host knows it); more types
    // than the workspace was carved for: nothing is written (the caller finds that out from the count and runs the general call)
    unsigned own_root_max = 0u;
    if (n_types_dev) {
        n_types = *n_types_dev;
        if (n_types > max_types_dev || n_types < 0) return;
        // (... and the root's bound taken here -- max |root| is exact, the same word forward_scales leaves in root_max -- so that the
        //  launch depends on nothing but the preparation it is queued behind)
        if (root_max) {
            __shared__ float wm[4];
            float m = 0.f;
            m = absmax4(m, reinterpret_cast<const float4 *>(roots.p[blockIdx.y])[threadIdx.x]);   // 1024 floats
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
            if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
            __syncthreads();
            own_root_max = __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
        }
    }
    const bool f16 = root_max != nullptr;
    const float wscale = f16 ? nnconv_weight_scale(n_types_dev ? own_root_max : root_max[blockIdx.y]) * img_scale : 1.0f;
    const int chunks = cc / 256, tid = threadIdx.x;
    auto image_of = [&](int t) { return wimg_all + ((int64_t)blockIdx.y * (n_types + 1) + t) * (f16 ? kWtTypeF16 : kWtType); };
    if ((int)blockIdx.x == chunks) {                             // the root matrix's image (width 32 only)
        const float *src = roots.p[blockIdx.y];
        __bf16 *img = reinterpret_cast<__bf16 *>(image_of(n_types));
        if (f16) for (int r = tid; r < 1024; r += 256) weight_image_put_f16(reinterpret_cast<_Float16 *>(img), r, src[r] * wscale);
        else for (int r = tid; r < 1024; r += 256) weight_image_put(img, r, src[r]);
    } else {
        const EdgeMlpLayer L = layers.l[blockIdx.y];
        const float *__restrict__ w1 = L.w1, *__restrict__ b1 = L.b1, *__restrict__ w2 = L.w2, *__restrict__ b2 = L.b2,
                    *__restrict__ w3 = L.w3, *__restrict__ b3 = L.b3;
        float *__restrict__ wtab = wtab_all + (int64_t)blockIdx.y * n_types * cc;
        __shared__ float e_s[kEwTypes * kEwMaxFe];
        __shared__ float h1_s[kEwTypes * kEH1];
        __shared__ __attribute__((aligned(16))) float h2_s[kEwTypes * kEH2];
        const int j = blockIdx.x * 256 + tid;                     // this thread's output
        float4 wr[kEH2 / 4];
#pragma unroll
        for (int q = 0; q < kEH2 / 4; ++q) wr[q] = reinterpret_cast<const float4 *>(w3 + (int64_t)j * kEH2)[q];
        const float bj = b3[j];
        for (int t0 = 0; t0 < n_types; t0 += kEwTypes) {
            const int nt = n_types - t0 < kEwTypes ? n_types - t0 : kEwTypes;
            __syncthreads();                                      // (the previous pass's h2 has been read)
            for (int i = tid; i < nt * fe; i += 256) {
                const int t = i / fe, k = i - t * fe;
                e_s[t * kEwMaxFe + k] = edge_attr[(int64_t)type_rep_edge[t0 + t] * fe + k];
            }
            __syncthreads();
            for (int i = tid; i < nt * kEH1; i += 256) {
                const int t = i / kEH1, u = i % kEH1;
                float acc = b1[u];
                for (int k = 0; k < fe; ++k) acc = fmaf(e_s[t * kEwMaxFe + k], w1[u * fe + k], acc);
                h1_s[t * kEH1 + u] = sigmoidf_(acc);
            }
            __syncthreads();
            for (int i = tid; i < nt * kEH2; i += 256) {
                const int t = i / kEH2, v = i % kEH2;
                float acc = b2[v];
#pragma unroll
                for (int k = 0; k < kEH1; ++k) acc = fmaf(h1_s[t * kEH1 + k], w2[v * kEH1 + k], acc);
                h2_s[t * kEH2 + v] = sigmoidf_(acc);
            }
            __syncthreads();
            for (int t = 0; t < nt; ++t) {
                float acc = bj;
#pragma unroll
                for (int q = 0; q < kEH2 / 4; ++q) {
                    const float4 h = *reinterpret_cast<const float4 *>(h2_s + t * kEH2 + 4 * q);   // (one address per wave: a broadcast)
                    acc = fmaf(h.x, wr[q].x, acc);
                    acc = fmaf(h.y, wr[q].y, acc);
                    acc = fmaf(h.z, wr[q].z, acc);
                    acc = fmaf(h.w, wr[q].w, acc);
                }
                const float v = sigmoidf_(acc);
                wtab[(int64_t)(t0 + t) * cc + j] = v;
                if (wimg_all) {
                    __bf16 *img = reinterpret_cast<__bf16 *>(image_of(t0 + t));
                    if (f16) weight_image_put_f16(reinterpret_cast<_Float16 *>(img), j, v * wscale);
                    else weight_image_put(img, j, v);
                }
            }
        }
    }
    if (done_ctr) {                                               // (a consumer on another stream counts the finished blocks)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave's stores are in L2 before thread 0 writes L2 back
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr size_t kMaxDynLds = 160 * 1024 - 256;

static bool edge_table_by_chunks(int fe, int c) { return fe <= kEwMaxFe && (c * c) % 256 == 0; }
bool edge_weight_table_device_count_ok(int fe, int c) { return edge_table_by_chunks(fe, c); }   // (the grid does not depend on the type count)
// blocks of the launch below = what its done counter reaches
unsigned edge_weight_table_blocks(int n_types, int fe, int depth, int c, bool image) {
    image = image && c == 32;
    return (unsigned)depth * (unsigned)((edge_table_by_chunks(fe, c) ? c * c / 256 : n_types) + (image ? 1 : 0));
}

void launch_edge_weight_table_batched(const float *edge_attr, const int *type_rep_edge, int n_types, int fe,
                                      const EdgeMlpLayers &layers, int depth, int c, float *wtab, const float *const *roots,
                                      float *wimg_all, hipStream_t s, unsigned *done_ctr, const unsigned *root_max, float img_scale,
                                      const int *n_types_dev, int max_types_dev) {
    RootPtrs rp{};
    const bool image = wimg_all && roots && c == 32;
    if (image)
        for (int i = 0; i < depth; ++i) rp.p[i] = roots[i];
    if (edge_table_by_chunks(fe, c)) {
        edge_weight_table_chunks_kernel<<<dim3(c * c / 256 + (image ? 1 : 0), depth), 256, 0, s>>>(
            edge_attr, type_rep_edge, fe, layers, c * c, wtab, n_types, rp, image ? wimg_all : nullptr, done_ctr, image ? root_max : nullptr,
            img_scale, n_types_dev, max_types_dev);
        return;
    }
    if (n_types_dev) return;                                  // (callers check edge_weight_table_device_count_ok first)
    edge_weight_table_kernel<<<dim3(n_types + (image ? 1 : 0), depth), 256, 0, s>>>(edge_attr, type_rep_edge, fe, layers, c * c, wtab,
                                                                                  n_types, rp, image ? wimg_all : nullptr, done_ctr,
                                                                                  image ? root_max : nullptr, img_scale);
}

// words [0, n_zero) = 0; then block b < depth: max |roots[b]| -> root_max[b]; the other blocks: their share of dense_w -> *dense_max
__global__ __launch_bounds__(256) void forward_scales_kernel(RootPtrs roots, int depth, unsigned *__restrict__ root_max,
                                                             const float *__restrict__ dense_w, int64_t dense_n4,
                                                             unsigned *__restrict__ dense_max) {
    float m = 0.f;
    unsigned *dst;
    if ((int)blockIdx.x < depth) {
        const float4 v = reinterpret_cast<const float4 *>(roots.p[blockIdx.x])[threadIdx.x];   // 1024 floats
        m = absmax4(m, v);
        dst = root_max + blockIdx.x;
    } else {
        const int nb = gridDim.x - depth, b = blockIdx.x - depth;
        for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < dense_n4; i += (int64_t)nb * 256)
            m = absmax4(m, reinterpret_cast<const float4 *>(dense_w)[i]);
        dst = dense_max;
    }
    absmax_flush(m, dst);
}
// Bounds of the final MLP's inner layers (their input is a train-mode BatchNorm's output): block l -> max |W_l| and
//   max_c ( |gamma_c| sqrt(n) + |beta_c| )  >=  max |BN(v)|   -- Samuelson's inequality: no sample lies further than
// sqrt(n - 1) standard deviations from the mean of the n samples the BatchNorm normalises with, so the bound needs the
// PARAMETERS only (loose by a factor ~sqrt(n) / 5: a scaled fp16 pair loses nothing to a loose bound, only to an overflow).
struct DenseBoundJob {
    const float *w;
    int64_t w_n4;
    const float *gamma, *beta;
    int f;
    unsigned *w_max, *a_max;
};
struct DenseBoundJobs {
    DenseBoundJob job[4];
};
// [r6] grid (jobs, slices): a job's weights are walked by `slices` blocks (one block per job took 17-20 us for 32 dependent rounds
// of loads); the words are zeroed by the caller, the blocks fold into them with atomicMax
__global__ __launch_bounds__(256) void dense_bounds_kernel(DenseBoundJobs jobs, float sqrt_n) {
    const DenseBoundJob j = jobs.job[blockIdx.x];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < j.w_n4; i += (int64_t)gridDim.y * 256)
        m = absmax4(m, reinterpret_cast<const float4 *>(j.w)[i]);
    absmax_flush(m, j.w_max);
    if (blockIdx.y != 0 || !j.gamma) return;                  // (uniform)
    float a = 0.f;
    for (int c = threadIdx.x; c < j.f; c += 256) a = fmaxf(a, fmaf(fabsf(j.gamma[c]), sqrt_n, fabsf(j.beta[c])));
    absmax_flush(a, j.a_max);
}
void launch_dense_bounds(int n_jobs, const float *const *w, const int64_t *w_n, const float *const *gamma, const float *const *beta,
                         const int *f, unsigned *const *w_max, unsigned *const *a_max, int64_t n_total, hipStream_t s) {
    DenseBoundJobs jobs{};
    for (int k = 0; k < n_jobs && k < 4; ++k) jobs.job[k] = DenseBoundJob{w[k], w_n[k] / 4, gamma[k], beta[k], f[k], w_max[k], a_max[k]};
    dense_bounds_kernel<<<dim3(n_jobs, 16), 256, 0, s>>>(jobs, sqrtf((float)n_total) * 1.0001f);
}

// [r6] launch_forward_scales without a memset and without atomics: block b = max |roots[b]| as a plain store; block 0 also clears
// every word of [0, n_zero) outside [root_off, root_off + depth)
__global__ __launch_bounds__(256) void forward_scales_store_kernel(RootPtrs roots, int depth, unsigned *__restrict__ words, int n_zero,
                                                                   int root_off) {
    __shared__ float wave_max[4];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256)
            if (i < root_off || i >= root_off + depth) words[i] = 0u;
    float m = 0.f;
    m = absmax4(m, reinterpret_cast<const float4 *>(roots.p[blockIdx.x])[threadIdx.x]);   // 1024 floats
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) words[root_off + blockIdx.x] = __float_as_uint(fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3])));
}

void launch_forward_scales(unsigned *words, int n_words, const float *const *roots, int depth, unsigned *root_max,
                           const float *dense_w, int64_t dense_n, unsigned *dense_max, hipStream_t s) {
    RootPtrs rp{};
    for (int i = 0; i < depth; ++i) rp.p[i] = roots[i];
    if (!dense_w && depth > 0 && root_max >= words && root_max + depth <= words + n_words) {
        forward_scales_store_kernel<<<depth, 256, 0, s>>>(rp, depth, words, n_words, (int)(root_max - words));
        return;
    }
    (void)hipMemsetAsync(words, 0, (size_t)n_words * sizeof(unsigned), s);
    const int dense_blocks = dense_w ? 32 : 0;
    if (depth + dense_blocks > 0)
        forward_scales_kernel<<<depth + dense_blocks, 256, 0, s>>>(rp, depth, root_max, dense_w, dense_n / 4, dense_max);
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int tgnn_edge_weight_table(const float *edge_attr, const int32_t *type_rep_edge, int32_t n_types,
                                      int32_t fe, const float *w1, const float *b1, const float *w2,
                                      const float *b2, const float *w3, const float *b3, int32_t c, float *wtab,
                                      tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (n_types <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(edge_attr && type_rep_edge && w1 && b1 && w2 && b2 && w3 && b3 && wtab, "null pointer");
    TGNN_CHECK_ARG(fe >= 1 && fe <= 1024, "edge feature dim must be in [1,1024]");
    TGNN_CHECK_ARG(c >= 1, "width");
    TGNN_CHECK_ARG((uintptr_t)w3 % 16 == 0, "w3 must be 16-byte aligned");
    EdgeMlpLayers layers{};
    layers.l[0] = EdgeMlpLayer{w1, b1, w2, b2, w3, b3};
    launch_edge_weight_table_batched(edge_attr, type_rep_edge, n_types, fe, layers, 1, c, wtab, nullptr, nullptr,
                                     static_cast<hipStream_t>(stream));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_nnconv_mean_fwd(const float *h, int64_t ldh, const int32_t *rowptr, const int32_t *col_src,
                                    const int32_t *col_type, const float *wtab, int32_t n_types,
                                    const float *root, const float *bias, int64_t n_nodes, int32_t c, int32_t act,
                                    float *out, double *bn_partial, int32_t *n_partials_host,
                                    tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0 && c >= 1, "shape");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    if (n_nodes == 0) {
        if (n_partials_host) *n_partials_host = 0;
        return TGNN_OK;
    }
    TGNN_CHECK_ARG(h && rowptr && root && bias && out, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || (col_src && col_type && wtab), "null graph pointer");
    TGNN_CHECK_ARG(ldh >= c, "ldh");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds_bytes = (size_t)(n_types + 1) * kTypeStride * sizeof(float);
    if (c == 32 && lds_bytes <= kMaxDynLds) {
        static LdsOptIn site;
        TGNN_CHECK_HIP(opt_in_dynamic_lds(nnconv32_lds_kernel, (int)kMaxDynLds, site));
        // one wave per row; at least one row per wave; a multiple of 8 blocks for the XCD split;
        // at most one block per CU (the LDS image is per block)
        int blocks = producer_blocks(n_nodes, kNNWaves);
        if (blocks > 256) blocks = 256;
        if (blocks >= 8) blocks &= ~7;
        nnconv32_lds_kernel<<<blocks, kNNThreads, lds_bytes, s>>>(h, ldh, rowptr, col_src, col_type, wtab, n_types,
                                                                 root, bias, n_nodes, act, out, bn_partial);
        if (n_partials_host) *n_partials_host = blocks;
    } else {
        TGNN_CHECK_ARG(c <= 256, "width must be <= 256");
        const int rows_per_block = 256 / c;
        const int blocks = producer_blocks(n_nodes, rows_per_block);
        nnconv_generic_kernel<<<blocks, 256, 256 * 2 * sizeof(double), s>>>(h, ldh, rowptr, col_src, col_type, wtab,
                                                                            root, bias, n_nodes, c, act, out,
                                                                            bn_partial);
        if (n_partials_host) *n_partials_host = blocks;
    }
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

namespace tgnn {
void launch_nnconv_weight_image(const float *wtab_all, const float *const *roots, int n_types, int depth,
                                float *wimg_all, hipStream_t s, const unsigned *root_max, float img_scale) {
    RootPtrs rp{};
    for (int i = 0; i < depth; ++i) rp.p[i] = roots[i];
    nnconv_weight_image_kernel<<<dim3(n_types + 1, depth), 256, 0, s>>>(wtab_all, rp, n_types, wimg_all, root_max, img_scale);
}
}  // namespace tgnn

extern "C" size_t tgnn_nnconv_weight_image_floats(int32_t n_types) { return (size_t)(n_types + 1) * kWtType; }
