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
__global__ __launch_bounds__(256) void edge_weight_table_kernel(
    const float *__restrict__ edge_attr, const int *__restrict__ type_rep_edge, int fe, EdgeMlpLayers layers, int cc,
    float *__restrict__ wtab_all) {
    const EdgeMlpLayer L = layers.l[blockIdx.y];
    const float *__restrict__ w1 = L.w1, *__restrict__ b1 = L.b1, *__restrict__ w2 = L.w2, *__restrict__ b2 = L.b2,
                *__restrict__ w3 = L.w3, *__restrict__ b3 = L.b3;
    float *__restrict__ wtab = wtab_all + (int64_t)blockIdx.y * gridDim.x * cc;
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
        wtab[(int64_t)t * cc + j] = sigmoidf_(acc);
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
// NNConv mean on matrix cores (C = 32, T + 1 <= 20): the production kernel.
//
// A tile = 64 destination rows.  Its in-edges arrive grouped by edge type in 16-slot chunks
// (tgnn_nnconv_tiles_build): one chunk = 16 gathered source rows x one [32,32] type matrix
//     M[16 x 32] = X[src(16) x 32] . W_t            = 2 column tiles x 8 x v_mfma_f32_16x16x4_f32
// (exact fp32).  The block's 8 waves split the tile's chunks into contiguous ranges (type changes
// ~3x per wave), keep the next two chunks' gathers in flight (8 VGPRs per chunk), and scatter-add
// each M row into a WAVE-PRIVATE [64 x 32] LDS accumulator with ds_add_f32 -- wave-private means the
// order of the adds is program order, so results are bit-reproducible.  The root term h[v].root
// rides along as 4 extra chunks of pseudo-type T whose rows are pre-multiplied by max(deg,1), so one
// common 1/deg scale applies at the end.  After a barrier 512 threads fold the 8 accumulators in
// fixed order: (sum)/deg + bias -> LeakyReLU -> store, fp64 BN column sums on the side.
//
// LDS: (T+1) x 4.5 KB weight image (B-operand order, padded) + 8 x 8 KB accumulators = 128 KB @T=13.
// ------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMfThreads = 512, kMfWaves = 8;
constexpr int kWtNt = 16 * 36;             // floats per (type, column tile): [j=16][q=4][ks=8], row stride 36
constexpr int kWtType = 2 * kWtNt;         // floats per type
constexpr int kAccFloats = 65 * 32;        // per-wave accumulator: 64 tile rows + 1 scratch row
constexpr int kHalf = 4;                   // chunks per gather stage
constexpr int kCap = 2 * kHalf;            // chunks per pipeline unit (indices of one unit live in registers)

// wtab [T][32][32] (+ root [32][32] as pseudo-type T) -> B-operand image [(T+1)][2][16][36]:
// element (i, o) of type t -> wimg[t][o >> 4][o & 15][(i >> 3) * 8 + (i & 7)]  (row padded 32 -> 36 floats).
// grid = (T+1, layers); done once per layer so that every NNConv block fills its LDS with a straight
// coalesced float4 copy instead of a 28-trip scatter.
struct RootPtrs {
    const float *p[kMaxDepth];
};
__global__ __launch_bounds__(256) void nnconv_weight_image_kernel(const float *__restrict__ wtab_all, RootPtrs roots,
                                                                  int n_types, float *__restrict__ wimg_all) {
    const int t = blockIdx.x, layer = blockIdx.y;
    const float *src = t < n_types ? wtab_all + ((int64_t)layer * n_types + t) * 1024 : roots.p[layer];
    float *dst = wimg_all + ((int64_t)layer * (n_types + 1) + t) * kWtType;
    for (int r = threadIdx.x; r < 1024; r += 256) {
        const int i = r >> 5, o = r & 31;
        dst[(o >> 4) * kWtNt + (o & 15) * 36 + (i >> 3) * 8 + (i & 7)] = src[r];
    }
    for (int k = threadIdx.x; k < 2 * 16 * 4; k += 256)      // zero the 4 padding floats of every row
        dst[(k >> 6) * kWtNt + ((k >> 2) & 15) * 36 + 32 + (k & 3)] = 0.f;
}

#ifdef TGNN_TIMING
__device__ unsigned long long g_nn_timing[256 * 8 * 8];
#define TGNN_T(slot) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[slot] += now_ - tlast; tlast = now_; }
#else
#define TGNN_T(slot)
#endif

struct UnitIdx {                           // per-lane view of one pipeline unit (<= kCap chunks of one tile)
    int type[kCap];                        // edge type of the chunk (wave-uniform; T = root chunk)
    int src[kCap];                         // source row of slot (lane & 15); -1 = padding / no chunk
    int rows4[kCap];                       // destination rows of slots 4q..4q+3 (q = lane >> 4), one byte each
};

// Vector-memory instructions are a scarce resource of this kernel: on gfx950 every wave-level load costs the
// CU's texture-address / L1 path ~40 cycles (dword) to ~70 cycles (the 16-rows x 4 x 16-B gather) no matter
// how few bytes it moves (scratch/ubench/vmem.hip), i.e. ~16 B/clk/CU.  The two gather instructions of a
// chunk (2 KB) are compulsory; the per-chunk index data is kept to three dword loads (type, 4 row bytes, src).
__global__ __launch_bounds__(kMfThreads) void nnconv32_mfma_kernel(
    const float *__restrict__ h, int64_t ldh, const int *__restrict__ rowptr, const int *__restrict__ tile_chunk_ptr,
    const int *__restrict__ chunk_meta, const int *__restrict__ slot_src, const float *__restrict__ wimg, int n_types,
    const float *__restrict__ bias, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;                                        // [(T+1)][2][16][36]
    float *accs = lds + (n_types + 1) * kWtType;            // [8 waves + 1 root][65][32]
    float *root_acc = accs + kMfWaves * kAccFloats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fj = lane & 15, fq = lane >> 4;

    {   // weight image: straight copy, all loads of a thread issued before the first LDS store
        const int n4 = (n_types + 1) * kWtType / 4;
        for (int i = tid; i < n4; i += 4 * kMfThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + u * kMfThreads < n4 ? i + u * kMfThreads : n4 - 1;
                v[u] = reinterpret_cast<const float4 *>(wimg)[ii];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * kMfThreads < n4) reinterpret_cast<float4 *>(wl)[i + u * kMfThreads] = v[u];
        }
    }
    float *acc_w = accs + wave * kAccFloats;
    for (int i = tid; i < (kMfWaves + 1) * kAccFloats / 4; i += kMfThreads)
        reinterpret_cast<float4 *>(accs)[i] = make_float4(0, 0, 0, 0);
    __syncthreads();

    const int64_t n_tiles = (n + 63) / 64;
    // final-phase mapping: thread -> (row fr, columns 4*fc .. 4*fc+3)
    const int fr = tid >> 3, fc = tid & 7;
    const float4 bias4 = reinterpret_cast<const float4 *>(bias)[fc];
    double cs[4] = {0, 0, 0, 0}, cq[4] = {0, 0, 0, 0};

    // XCD-contiguous tile ranges (block b runs on XCD b % 8; speed only)
    const int nblk = gridDim.x;
    int64_t t_beg = blockIdx.x, t_end = n_tiles, t_step = nblk;
    if (nblk >= 8 && (nblk & 7) == 0) {
        const int xcd = blockIdx.x & 7;
        t_beg = n_tiles * xcd / 8 + (blockIdx.x >> 3);
        t_end = n_tiles * (xcd + 1) / 8;
        t_step = nblk >> 3;
    }

    // ---- helpers (all loads unconditional + selects: a predicated load becomes a branch, and hipcc drains
    //      vmcnt(0) at the join, which would serialise the pipeline)
    auto tile_range = [&](int64_t tile, int &c_lo, int &c_hi) {       // chunk range of a tile, empty past the end
        const int64_t tt = tile < t_end ? tile : 0;
        const int lo = tile_chunk_ptr[tt], hi = tile_chunk_ptr[tt + 1];
        c_lo = lo;
        c_hi = tile < t_end ? hi : lo;
    };
    auto wave_share = [&](int c_lo, int c_hi, int &c0, int &cnt) {    // contiguous share of this wave
        const int n_ch = c_hi - c_lo;
        c0 = c_lo + n_ch * wave / kMfWaves;
        cnt = c_lo + n_ch * (wave + 1) / kMfWaves - c0;
    };
    auto load_unit = [&](int c0, int cnt, UnitIdx &ui) {
        // NB: fetching the wave-uniform record through the scalar cache (s_load_dwordx8) was tried and is
        // SLOWER (76.7 vs 69.5 us): SMEM shares lgkmcnt with the LDS traffic of the scatter and returns out
        // of order, so every use forces lgkmcnt(0).  Three dword vector loads per chunk it is.
#pragma unroll
        for (int u = 0; u < kCap; ++u) {
            const bool ok = u < cnt;
            const int cc = ok ? c0 + u : 0;
            const int ty = chunk_meta[cc * 8], r4 = chunk_meta[cc * 8 + 4 + fq], sr = slot_src[cc * 16 + fj];
            ui.type[u] = ty;
            ui.src[u] = ok ? sr : -1;
            ui.rows4[u] = r4;
        }
    };
    auto gather = [&](int src, float4 (&x)[2]) {          // 8 k-values (32 B) of one source row
        const int sidx = src >= 0 ? src : 0;
        const float4 *px = reinterpret_cast<const float4 *>(h + (int64_t)sidx * ldh + fq * 8);
        x[0] = px[0];
        x[1] = px[1];
    };

    int cur_type = -1;
    float bw0[8], bw1[8];
    // ---- scatter of one chunk's D tiles into an accumulator that only this wave writes.
    // D row (4q + r) belongs to destination row byte r of rows4; col = fj (+16).  LDS float atomics run
    // ~2 cycles PER LANE on gfx950 (ds_add_f32: 148 cycles / instruction measured), so equal-row slots --
    // adjacent, the group is in CSR order -- are first summed in registers (segmented scan: in-lane over r,
    // then a 3-step carry across the four 16-lane groups); the last slot of every run then owns its
    // accumulator row: plain read-add-write.
    auto scatter = [&](float *accb, const f32x4 &d0, const f32x4 &d1, int rows4) {
        const int r0 = rows4 & 0xff, r1 = (rows4 >> 8) & 0xff, r2 = (rows4 >> 16) & 0xff, r3 = (rows4 >> 24) & 0xff;
        const bool e1 = r1 == r0, e2 = r2 == r1, e3 = r3 == r2;
        float s0[4], s1[4];
        s0[0] = d0[0]; s1[0] = d1[0];
        s0[1] = e1 ? s0[0] + d0[1] : d0[1]; s1[1] = e1 ? s1[0] + d1[1] : d1[1];
        s0[2] = e2 ? s0[1] + d0[2] : d0[2]; s1[2] = e2 ? s1[1] + d1[2] : d1[2];
        s0[3] = e3 ? s0[2] + d0[3] : d0[3]; s1[3] = e3 ? s1[2] + d1[3] : d1[3];
        const bool lead[4] = {true, e1, e1 && e2, e1 && e2 && e3};     // slots still in the lane's first run
        const int prev_r3 = __shfl_up(r3, 16, 64), next_r0 = __shfl_down(r0, 16, 64);
        const bool joins_prev = fq > 0 && prev_r3 == r0;
        if (__any(joins_prev)) {                          // wave-uniform: some run crosses a 16-lane group
#pragma unroll
            for (int step = 1; step <= 3; ++step) {
                const float t0 = __shfl_up(s0[3], 16, 64), t1 = __shfl_up(s1[3], 16, 64);
                if (fq == step && joins_prev) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (lead[r]) { s0[r] += t0; s1[r] += t1; }
                }
            }
        }
        const bool last[4] = {!e1, !e2, !e3, fq == 3 || next_r0 != r3};
        const int rr[4] = {r0, r1, r2, r3};
        float *dst[4];
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                    // non-final slots are parked on the scratch row 64
            dst[r] = accb + (last[r] ? rr[r] : 64) * 32 + fj;
            o0[r] = dst[r][0];
            o1[r] = dst[r][16];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dst[r][0] = o0[r] + s0[r];
            dst[r][16] = o1[r] + s1[r];
        }
    };
    // The scatter of chunk k is issued AFTER the MFMAs of chunk k+1 (its D tiles wait in 8 VGPRs), so that its
    // shuffle / LDS round trips run in the shadow of the matrix pipe instead of between two MFMA bursts.
    f32x4 pend0 = {0.f, 0.f, 0.f, 0.f}, pend1 = {0.f, 0.f, 0.f, 0.f};
    int pend_rows = 0;
    float *pend_acc = acc_w;
    bool pending = false;
    auto flush = [&]() {
        if (pending) scatter(pend_acc, pend0, pend1, pend_rows);
        pending = false;
    };
    auto compute = [&](int type_v, int src, int rows4, const float4 (&x)[2]) {
        // B fragments only when the type changes (chunks are type-sorted).  The type came with the prefetched
        // indices: a load issued here would queue behind the prefetches (vmcnt is in-order).
        const int t = __builtin_amdgcn_readfirstlane(type_v);
        if (t != cur_type) {                             // wave-uniform
            cur_type = t;
            const float *wp = wl + t * kWtType + fj * 36 + fq * 8;
            const float4 p0 = *reinterpret_cast<const float4 *>(wp), p1 = *reinterpret_cast<const float4 *>(wp + 4);
            const float4 p2 = *reinterpret_cast<const float4 *>(wp + kWtNt), p3 = *reinterpret_cast<const float4 *>(wp + kWtNt + 4);
            bw0[0] = p0.x; bw0[1] = p0.y; bw0[2] = p0.z; bw0[3] = p0.w; bw0[4] = p1.x; bw0[5] = p1.y; bw0[6] = p1.z; bw0[7] = p1.w;
            bw1[0] = p2.x; bw1[1] = p2.y; bw1[2] = p2.z; bw1[3] = p2.w; bw1[4] = p3.x; bw1[5] = p3.y; bw1[6] = p3.z; bw1[7] = p3.w;
        }
        const bool valid = src >= 0;                     // padding slots multiply zeros
        const float xv[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
        float a8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] = valid ? xv[k] : 0.f;
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a8[k], bw0[k], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a8[k], bw1[k], d1, 0, 0, 0);
        }
        if (pending) scatter(pend_acc, pend0, pend1, pend_rows);   // previous chunk, in this chunk's MFMA shadow
        // root chunks (type T) carry h[v] . root, which is NOT divided by the in-degree: they go to their own
        // accumulator (each root chunk owns 16 distinct rows, so the waves holding them never collide)
        pend0 = d0; pend1 = d1; pend_rows = rows4; pend_acc = t == n_types ? root_acc : acc_w; pending = true;
    };

    // ---- pipeline over UNITS (<= 8 chunks of one tile; a tile is 1+ units per wave):
    //   unit u+1's indices load while unit u computes; its first-half gathers are issued before unit u's second
    //   half runs, its second-half gathers before its own first half.  Tile ranges are fetched one tile ahead.
    //   Every dependent-load level (range -> indices -> gathers) therefore has >= 4 chunks of MFMAs to land,
    //   also ACROSS the tile boundary (barrier + fold), where a per-tile pipeline would restart cold.
    int64_t tile = t_beg;
    int lo, hi, c0, rem;                                 // current tile: range, next chunk of this wave, chunks left
    tile_range(tile, lo, hi);
    wave_share(lo, hi, c0, rem);
    int nlo, nhi;                                        // range of the next tile (prefetched)
    tile_range(tile + t_step, nlo, nhi);
    UnitIdx ua;
    load_unit(c0, rem, ua);
    float4 xa[kHalf][2], xb[kHalf][2];
#pragma unroll
    for (int u = 0; u < kHalf; ++u) gather(ua.src[u], xa[u]);

#ifdef TGNN_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    while (tile < t_end) {
        const bool last_unit = rem <= kCap;
        TGNN_T(7)
        // ---- next unit
        // (no load may sit inside a branch: hipcc drains vmcnt(0) at the join -- everything here is
        //  loaded unconditionally and selected)
        const int64_t vtile = last_unit ? tile + t_step : tile;
        int sc0, srem;
        wave_share(nlo, nhi, sc0, srem);
        const int vc0 = last_unit ? sc0 : c0 + kCap, vrem = last_unit ? srem : rem - kCap;
        int qlo, qhi;
        tile_range(tile + 2 * t_step, qlo, qhi);         // range after the next tile (used once we cross over)
        const int plo = last_unit ? qlo : nlo, phi = last_unit ? qhi : nhi;
        // in-degree of this thread's fold row, fetched a whole unit before the fold needs it
        const int64_t frow = tile * 64 + fr;
        const int64_t frc = frow < n ? frow : n - 1;
        const int fdeg = rowptr[frc + 1] - rowptr[frc];
        UnitIdx ub;
        load_unit(vc0, vrem, ub);
#pragma unroll
        for (int u = 0; u < kHalf; ++u) gather(ua.src[kHalf + u], xb[u]);
        TGNN_T(0)
#pragma unroll
        for (int u = 0; u < kHalf; ++u)
            if (u < rem) compute(ua.type[u], ua.src[u], ua.rows4[u], xa[u]);
        TGNN_T(1)
#pragma unroll
        for (int u = 0; u < kHalf; ++u) gather(ub.src[u], xa[u]);
#pragma unroll
        for (int u = 0; u < kHalf; ++u)
            if (kHalf + u < rem) compute(ua.type[kHalf + u], ua.src[kHalf + u], ua.rows4[kHalf + u], xb[u]);
        TGNN_T(2)

#ifdef TGNN_ABLATE_MF_NOFOLD
        if (last_unit && tile + t_step >= t_end) {
#else
        if (last_unit) {
#endif
            flush();
            TGNN_T(3)
            __syncthreads();
            TGNN_T(4)
            // ---- fold the 8 accumulators (fixed order), finish the row, re-zero for the next tile
            const int64_t v = tile * 64 + fr;
            float4 sum = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < kMfWaves; ++w) {
                float4 *pa = reinterpret_cast<float4 *>(accs + w * kAccFloats + fr * 32 + fc * 4);
                const float4 a4 = *pa;
                sum.x += a4.x; sum.y += a4.y; sum.z += a4.z; sum.w += a4.w;
                *pa = make_float4(0, 0, 0, 0);
            }
            float4 *pr = reinterpret_cast<float4 *>(root_acc + fr * 32 + fc * 4);
            const float4 rt = *pr;
            *pr = make_float4(0, 0, 0, 0);
            if (v < n) {
                const float inv = 1.0f / (float)(fdeg > 0 ? fdeg : 1);
                float4 o;
                o.x = fmaf(sum.x, inv, rt.x) + bias4.x; o.y = fmaf(sum.y, inv, rt.y) + bias4.y;
                o.z = fmaf(sum.z, inv, rt.z) + bias4.z; o.w = fmaf(sum.w, inv, rt.w) + bias4.w;
                if (act == TGNN_ACT_LEAKY_RELU) { o.x = leakyf_(o.x); o.y = leakyf_(o.y); o.z = leakyf_(o.z); o.w = leakyf_(o.w); }
                *reinterpret_cast<float4 *>(out + v * 32 + fc * 4) = o;
                cs[0] += (double)o.x; cq[0] += (double)o.x * (double)o.x;
                cs[1] += (double)o.y; cq[1] += (double)o.y * (double)o.y;
                cs[2] += (double)o.z; cq[2] += (double)o.z * (double)o.z;
                cs[3] += (double)o.w; cq[3] += (double)o.w * (double)o.w;
            }
            TGNN_T(5)
            __syncthreads();
            TGNN_T(6)
        }
        // ---- rotate
        tile = vtile; c0 = vc0; rem = vrem; nlo = plo; nhi = phi;
        ua = ub;
    }

#ifdef TGNN_TIMING
    if (lane == 0)
        for (int k = 0; k < 8; ++k) g_nn_timing[(blockIdx.x * 8 + wave) * 8 + k] = tacc[k];
#endif
    if (bn_partial) {
        // 64 row-threads per column group: fold in fixed order through LDS ([64][64] doubles = 32 KB, aliases accs)
        __syncthreads();
        double *red = reinterpret_cast<double *>(accs);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[fr * 64 + fc * 4 + k] = cs[k];
            red[fr * 64 + 32 + fc * 4 + k] = cq[k];
        }
        __syncthreads();
        if (tid < 64) {
            double tot = 0.0;
            for (int r = 0; r < 64; ++r) tot += red[r * 64 + tid];
            bn_partial[(int64_t)blockIdx.x * 64 + tid] = tot;   // [2][32]: sums then sums of squares
        }
    }
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

constexpr size_t kMaxDynLds = 160 * 1024 - 256;

void launch_edge_weight_table_batched(const float *edge_attr, const int *type_rep_edge, int n_types, int fe,
                                      const EdgeMlpLayers &layers, int depth, int c, float *wtab, hipStream_t s) {
    edge_weight_table_kernel<<<dim3(n_types, depth), 256, 0, s>>>(edge_attr, type_rep_edge, fe, layers, c * c, wtab);
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int tgnn_edge_weight_table(const float *edge_attr, const int32_t *type_rep_edge, int32_t n_types,
                                      int32_t fe, const float *w1, const float *b1, const float *w2,
                                      const float *b2, const float *w3, const float *b3, int32_t c, float *wtab,
                                      tgnn_stream_t stream) {
    if (n_types <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(edge_attr && type_rep_edge && w1 && b1 && w2 && b2 && w3 && b3 && wtab, "null pointer");
    TGNN_CHECK_ARG(fe >= 1 && fe <= 1024, "edge feature dim must be in [1,1024]");
    TGNN_CHECK_ARG(c >= 1, "width");
    TGNN_CHECK_ARG((uintptr_t)w3 % 16 == 0, "w3 must be 16-byte aligned");
    EdgeMlpLayers layers{};
    layers.l[0] = EdgeMlpLayer{w1, b1, w2, b2, w3, b3};
    launch_edge_weight_table_batched(edge_attr, type_rep_edge, n_types, fe, layers, 1, c, wtab,
                                     static_cast<hipStream_t>(stream));
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_nnconv_mean_fwd(const float *h, int64_t ldh, const int32_t *rowptr, const int32_t *col_src,
                                    const int32_t *col_type, const float *wtab, int32_t n_types,
                                    const float *root, const float *bias, int64_t n_nodes, int32_t c, int32_t act,
                                    float *out, double *bn_partial, int32_t *n_partials_host,
                                    tgnn_stream_t stream) {
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
        static bool attr_set = false;  // idempotent, racing setters write the same value
        if (!attr_set) {
            TGNN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nnconv32_lds_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
            attr_set = true;
        }
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
                                float *wimg_all, hipStream_t s) {
    RootPtrs rp{};
    for (int i = 0; i < depth; ++i) rp.p[i] = roots[i];
    nnconv_weight_image_kernel<<<dim3(n_types + 1, depth), 256, 0, s>>>(wtab_all, rp, n_types, wimg_all);
}

static size_t tiled_lds_bytes(int n_types) {
    return ((size_t)(n_types + 1) * kWtType + (size_t)(kMfWaves + 1) * kAccFloats) * sizeof(float);
}

int launch_nnconv_tiled(const float *h, int64_t ldh, const int32_t *rowptr, const int32_t *tile_chunk_ptr,
                        const int32_t *chunk_meta, const int32_t *slot_src, const float *wimg, int32_t n_types,
                        const float *bias, int64_t n_nodes, int32_t act, float *out, double *bn_partial,
                        int32_t *n_partials_host, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        TGNN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nnconv32_mfma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
        attr_set = true;
    }
    int blocks = producer_blocks(n_nodes, 64);
    if (blocks > 256) blocks = 256;            // one block per CU (LDS), persistent over tiles
    if (blocks >= 8) blocks &= ~7;
    nnconv32_mfma_kernel<<<blocks, kMfThreads, tiled_lds_bytes(n_types), s>>>(
        h, ldh, rowptr, tile_chunk_ptr, chunk_meta, slot_src, wimg, n_types, bias, n_nodes, act, out, bn_partial);
    if (n_partials_host) *n_partials_host = blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

}  // namespace tgnn
using namespace tgnn;

#ifdef TGNN_TIMING
extern "C" int tgnn_debug_nn_timing(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nn_timing), sizeof(unsigned long long) * 256 * 8 * 8);
}
#endif

extern "C" size_t tgnn_nnconv_weight_image_floats(int32_t n_types) { return (size_t)(n_types + 1) * kWtType; }

extern "C" int tgnn_nnconv_mean_tiled_fwd(const float *h, int64_t ldh, const int32_t *rowptr,
                                          const int32_t *tile_chunk_ptr, const int32_t *chunk_meta,
                                          const int32_t *slot_src, const float *wtab, int32_t n_types, const float *root, const float *bias,
                                          int64_t n_nodes, int32_t c, int32_t act, float *out, float *wimg_scratch,
                                          double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream) {
    TGNN_CHECK_ARG(n_nodes >= 1 && c == 32, "tiled NNConv is built for network_width 32");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && rowptr && tile_chunk_ptr && chunk_meta && slot_src && root && bias && out && wimg_scratch,
                   "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(ldh >= 32 && ldh % 4 == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                       ((uintptr_t)bias % 16) == 0 && ((uintptr_t)wimg_scratch % 16) == 0, "alignment");
    if (tiled_lds_bytes(n_types) > kMaxDynLds) {
        set_error("tgnn_nnconv_mean_tiled_fwd: %d edge types do not fit the LDS weight image (max %d)", n_types,
                  tgnn_nnconv_tiled_max_types());
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_nnconv_weight_image(wtab, &root, n_types, 1, wimg_scratch, s);
    return launch_nnconv_tiled(h, ldh, rowptr, tile_chunk_ptr, chunk_meta, slot_src, wimg_scratch, n_types, bias,
                               n_nodes, act, out, bn_partial, n_partials_host, s);
}

extern "C" int32_t tgnn_nnconv_tiled_max_types(void) {
    return (int32_t)((kMaxDynLds / sizeof(float) - (kMfWaves + 1) * kAccFloats) / kWtType) - 1;
}
