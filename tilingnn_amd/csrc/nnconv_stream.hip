// NNConv(aggr="mean"), network_width 32: the throughput kernel -- gathered rows as ONE stream through an LDS ring.
//
// Reference semantics: GraphConv.forward (/root/reference/graph_networks/layers/edge_conv.py:24-27) over PyG 1.3.2 NNConv:
//     out[v] = mean_{e: dst_e = v} h[src_e] . W_{type_e}  +  h[v] . root + bias  (+ LeakyReLU)
// Same algebra as nnconv_cols.hip (the sum over a row's edges of one type moves inside the product), another machine.
// What bounds this op on gfx950 is not HBM, LDS or the matrix pipe but INSTRUCTION ISSUE: a SIMD starts about one
// instruction per 4 cycles, whatever its kind (rocprof: instructions executed = SIMD-quad-cycles of the launch, for this
// kernel and for nnconv_cols.hip alike), so the design goal is the smallest instruction count per 16-row tile:
//
//   * the source rows of a block's tiles are ONE stream of 128-byte rows ("entries": per 16-row tile its edge types in
//     order -- the tile's own rows last = the root run --, inside a type the destination rows ascending, a row's edges in
//     CSR order), fetched by two loader waves with LDS-DMA (buffer_load_dwordx4 ... lds: 8 whole rows per instruction,
//     8 consecutive lanes per row: the cheap shape for the CU's address path) into a 96 KB ring: no register holds a row
//     in flight, the loaders run two to three tiles ahead of the arithmetic, and a loader spends ~8 instructions per 8 rows
//     (the entry word IS the buffer offset, the piece permutation one XOR with a lane constant);
//   * 8 multiplying waves own the edge types round-robin (type t -> wave t & 7): their weight fragments (three bf16 planes,
//     48 VGPRs for two types) stay in REGISTERS for the whole kernel.  Per tile and type a wave gets two 16-bit planes of
//     the rows' edge counts (0..3) and the first entry of the type: a lane's first entry = base + prefix count (4
//     instructions), its further entries sit at +128 B (immediate offsets, executed under the hardware mask built from the
//     planes on the scalar unit: no per-column address arithmetic, absent rows read a zero slot).  Sum in registers (CSR
//     order: the bits of nnconv_cols.hip's pre-add), split, 12 MFMAs, one 16 x 32 partial product per wave;
//   * 2 epilogue waves add the eight partials in wave order, scale by 1 / max(deg, 1), add the root product (kept apart:
//     no pre-multiplication by the degree), bias, LeakyReLU, store, and keep the BatchNorm sums in fp64.
//   One block barrier per tile; the three roles work on tiles s+1.., s and s-1 of the block's contiguous tile range.
//
// Ring image: entry p of the stream (absolute position) sits in slot p mod 768; the eight 16-byte pieces of the row of
// destination row j (0..15 inside its tile) are stored at q ^ X(j), X = 2 ((j >> 1) & 3) (the loader permutes the SOURCE
// piece per lane -- LDS-DMA writes lane-linear --, the permutation travels in the entry word), and kgroup g of the matrix
// operand holds channels 4g..4g+3 and 16+4g..16+4g+3 (pieces g and g+4): a lane's LDS address is then
// 128 slot + a lane constant, and the sixteen rows of every ds_read_b128 lane group fall on sixteen different bank
// quads whatever rows are present (rows 2m and 2m+1 share X and sit in slots of opposite parity).
#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kStRingChunks = 96;                        // 1 KB chunks = 8 row slots each
constexpr int kStRingSlots = kStRingChunks * 8;          // 768
constexpr int kStMult = 8, kStDma = 2, kStEpi = 2, kStWaves = kStMult + kStDma + kStEpi;
constexpr int kStInfoQuads = 2 * kStMult;                // per tile: 2 x uint4 for every multiplying wave
constexpr int kStMaxTileChunks = (kStRingChunks - 8) / 2;      // two consecutive tiles + one group of slack fit the ring
constexpr int kStMaxTypes = 2 * kStMult - 1;                   // T + 1 runs over 8 waves, two per wave
constexpr int kStMaxMult = 15;                                 // same-type in-edges of one row (four count planes)
constexpr unsigned kStPadWord = 0xffffff80u;                   // entry word beyond any buffer: the DMA fetches nothing

// LDS map (bytes)
constexpr int kStOffDump = kStRingChunks * 1024;               // 1 KB: chunks past the block's range land here
constexpr int kStOffZero = kStOffDump + 1024;                  // 128 B of zeros: what an absent row reads
constexpr int kStOffIdx = kStOffZero + 256;                    // [2 loader waves][4][64] dwords: entry words of a group
constexpr int kStOffPart = (kStOffIdx + 2 * 4 * 256 + 1023) / 1024 * 1024;   // [2][9][2][64] x 16 B
constexpr int kStPartBuf = 9 * 2 * 1024;
constexpr int kStLdsBytes = kStOffPart + 2 * kStPartBuf;

template <int N>
__device__ __forceinline__ void st_wait_vmcnt() {
#ifdef TGNN_ST_WAIT0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// all of this wave's loader groups up to one that has `m` younger groups behind it are complete (a group = 8 row DMAs +
// the entry-word DMA of the group three ahead: 9 m + 1 younger instructions may stay in flight)
__device__ __forceinline__ void st_wait_groups_behind(int m) {
    switch (m) {
        case 0: st_wait_vmcnt<1>(); break;
        case 1: st_wait_vmcnt<10>(); break;
        case 2: st_wait_vmcnt<19>(); break;
        case 3: st_wait_vmcnt<28>(); break;
        case 4: st_wait_vmcnt<37>(); break;
        case 5: st_wait_vmcnt<46>(); break;
        case 6: st_wait_vmcnt<55>(); break;
        default: break;                                   // 64+ cannot be outstanding
    }
}
#ifdef TGNN_ST_TIMING
__device__ unsigned long long g_st_time[512 * 12 * 4];   // [block][wave]: work cycles, barrier-wait cycles, steps, total
#define ST_BARRIER()                                                            \
    {                                                                           \
        const unsigned long long a_ = __builtin_readcyclecounter();            \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
        const unsigned long long b_ = __builtin_readcyclecounter();            \
        t_work += a_ - t_last; t_wait += b_ - a_; t_last = b_; ++t_steps;       \
    }
#else
#define ST_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#ifdef TGNN_ST_DEBUG
__device__ int g_st_dbg[512 * 64];        // [block][tile of the block]: ring slots of the tile that differ from h[src]
#endif

__device__ __forceinline__ unsigned long long st_rep64(unsigned m16) {   // row mask -> the four kgroups of a wave
    const unsigned r = m16 * 0x00010001u;
    return ((unsigned long long)r << 32) | r;
}

__global__ __launch_bounds__(kStWaves * 64) void nnconv32_stream_kernel(
    const float *__restrict__ h, unsigned h_bytes, const int *__restrict__ tile_ent_ptr, const unsigned *__restrict__ ent,
    const uint4 *__restrict__ info, const float *__restrict__ inv_deg, const float *__restrict__ wimg, int n_types,
    const float *__restrict__ bias, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;

    // ---- this block's contiguous run of tiles; blocks of one XCD (b % 8) take neighbouring runs (shared L2 lines)
    const int n_tiles = (int)((n + 15) >> 4);
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int t0 = (int)((int64_t)n_tiles * blk / nblk), t1 = (int)((int64_t)n_tiles * (blk + 1) / nblk);

    if (tid < 32) reinterpret_cast<float *>(lds + kStOffZero)[tid] = 0.f;
#ifdef TGNN_ST_TIMING
    unsigned long long t_work = 0, t_wait = 0, t_steps = 0, t_last = __builtin_readcyclecounter();
    const unsigned long long t_begin = t_last;
#endif

    if (wave < kStMult) {
        // =================================================================== multiplying waves
        const int w = wave, r0 = w, r1 = w + kStMult, n_runs = n_types + 1;
        // weight fragments of this wave's runs, K order of the ring image: element j of kgroup g = channel 4g + j (j < 4),
        // 16 + 4g + j - 4 (j >= 4): two 8-byte halves of two fragments of the natural-order image (tgnn_common.h: kWtType)
        bf16x8 wa[6], wb[6];
#pragma unroll
        for (int pm = 0; pm < 6; ++pm) {
            const int p = pm >> 1, m = pm & 1;
            const int f_lo = p * 128 + m * 64 + 16 * (fq >> 1) + fj, f_hi = f_lo + 32;
            bf16x4 a_lo = {0, 0, 0, 0}, a_hi = a_lo, b_lo = a_lo, b_hi = a_lo;
            if (r0 < n_runs) {
                const bf16x4 *img = reinterpret_cast<const bf16x4 *>(wimg + (size_t)r0 * kWtType);
                a_lo = img[2 * f_lo + (fq & 1)];
                a_hi = img[2 * f_hi + (fq & 1)];
            }
            if (r1 < n_runs) {
                const bf16x4 *img = reinterpret_cast<const bf16x4 *>(wimg + (size_t)r1 * kWtType);
                b_lo = img[2 * f_lo + (fq & 1)];
                b_hi = img[2 * f_hi + (fq & 1)];
            }
            wa[pm] = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
            wb[pm] = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        const bool root0 = r0 == n_types, root1 = r1 == n_types;
        const unsigned lt16 = (1u << fj) - 1u, lt2 = lt16 | (lt16 << 16);
        const unsigned xsw = 2u * ((fj >> 1) & 3);           // this destination row's piece permutation
        const unsigned c1 = ((unsigned)fq ^ xsw) << 4;       // piece fq; piece fq + 4 sits 64 bytes from it
        const unsigned z0 = kStOffZero + fq * 16;

        // One type run of the tile -> the 8 channels of this lane's row, summed in CSR order.  p01 / p23: bit j (16 + j) =
        // bit 0 / 2 (1 / 3) of row j's edge count; b: ring slot of the run's first entry.
        auto run_sum = [&](unsigned p01, unsigned p23, unsigned b, unsigned len, float (&af)[8]) {
            const unsigned l01 = p01 & lt2;
            unsigned start = __builtin_popcount(l01 & 0xffffu) + b;
            start += 2u * __builtin_popcount(l01 >> 16);
            unsigned ge1 = (p01 | (p01 >> 16)) & 0xffffu, ge2 = p01 >> 16, ge3 = p01 & (p01 >> 16) & 0xffffu;
            if (p23) {                                       // (a row with 4+ edges of one type: ~one run in eight)
                const unsigned l23 = p23 & lt2, any23 = (p23 | (p23 >> 16)) & 0xffffu;
                start += 4u * __builtin_popcount(l23 & 0xffffu);
                start += 8u * __builtin_popcount(l23 >> 16);
                ge1 |= any23; ge2 |= any23; ge3 |= any23;
            }
            if (b + len <= (unsigned)kStRingSlots) {
                // (the run does not straddle the ring's end: a row's further entries are 128 bytes on)
                const unsigned a0 = (start << 7) + c1;
                const bool pr0 = __builtin_amdgcn_inverse_ballot_w64(st_rep64(ge1));
                const unsigned a0z = pr0 ? a0 : z0;
                const float4 x0 = *reinterpret_cast<const float4 *>(lds + a0z);
                const float4 x1 = *reinterpret_cast<const float4 *>(lds + (a0z ^ 64u));
                af[0] = x0.x; af[1] = x0.y; af[2] = x0.z; af[3] = x0.w;
                af[4] = x1.x; af[5] = x1.y; af[6] = x1.z; af[7] = x1.w;
                if (ge2) {                                   // some row has a second edge of this type
                    if (__builtin_amdgcn_inverse_ballot_w64(st_rep64(ge2))) {
                        const float4 y0 = *reinterpret_cast<const float4 *>(lds + a0 + 128);
                        const float4 y1 = *reinterpret_cast<const float4 *>(lds + (a0 ^ 64u) + 128);
                        af[0] += y0.x; af[1] += y0.y; af[2] += y0.z; af[3] += y0.w;
                        af[4] += y1.x; af[5] += y1.y; af[6] += y1.z; af[7] += y1.w;
                    }
                    if (ge3) {
                        if (__builtin_amdgcn_inverse_ballot_w64(st_rep64(ge3))) {
                            const float4 y0 = *reinterpret_cast<const float4 *>(lds + a0 + 256);
                            const float4 y1 = *reinterpret_cast<const float4 *>(lds + (a0 ^ 64u) + 256);
                            af[0] += y0.x; af[1] += y0.y; af[2] += y0.z; af[3] += y0.w;
                            af[4] += y1.x; af[5] += y1.y; af[6] += y1.z; af[7] += y1.w;
                        }
                        if (p23) {
                            const unsigned cnt = ((p01 >> fj) & 1u) + 2u * ((p01 >> (16 + fj)) & 1u) + 4u * ((p23 >> fj) & 1u) +
                                                 8u * ((p23 >> (16 + fj)) & 1u);
                            const unsigned top = (p23 >> 16) ? 15u : 7u;
                            for (unsigned j = 3; j < top; ++j) {
                                if (cnt > j) {
                                    const float4 y0 = *reinterpret_cast<const float4 *>(lds + a0 + j * 128);
                                    const float4 y1 = *reinterpret_cast<const float4 *>(lds + (a0 ^ 64u) + j * 128);
                                    af[0] += y0.x; af[1] += y0.y; af[2] += y0.z; af[3] += y0.w;
                                    af[4] += y1.x; af[5] += y1.y; af[6] += y1.z; af[7] += y1.w;
                                }
                            }
                        }
                    }
                }
            } else {
                // the run wraps: every entry's slot on its own (one run in ~4 tiles)
                const unsigned cnt = ((p01 >> fj) & 1u) + 2u * ((p01 >> (16 + fj)) & 1u) + 4u * ((p23 >> fj) & 1u) +
                                     8u * ((p23 >> (16 + fj)) & 1u);
                const unsigned top = p23 ? ((p23 >> 16) ? 15u : 7u) : 3u;
#pragma unroll
                for (int k = 0; k < 8; ++k) af[k] = 0.f;
                for (unsigned j = 0; j < top; ++j) {
                    if (cnt > j) {
                        unsigned pos = start + j;
                        pos = pos >= (unsigned)kStRingSlots ? pos - kStRingSlots : pos;
                        const unsigned a = (pos << 7) + c1;
                        const float4 y0 = *reinterpret_cast<const float4 *>(lds + a);
                        const float4 y1 = *reinterpret_cast<const float4 *>(lds + (a ^ 64u));
                        af[0] += y0.x; af[1] += y0.y; af[2] += y0.z; af[3] += y0.w;
                        af[4] += y1.x; af[5] += y1.y; af[6] += y1.z; af[7] += y1.w;
                    }
                }
            }
        };
        auto run_mfma = [&](const float (&af)[8], const bf16x8 (&wf)[6], f32x4 &d0, f32x4 &d1) {
            bf16x8 xh, xm, xl;
            split3_trunc(af, xh, xm, xl);
            // wf[2 p + m]: plane p (hi, mid, lo), M block m; six cross terms, smallest first, fp32 accumulation
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[4], xh, d0, 0, 0, 0);   // lo . hi
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[5], xh, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], xl, d0, 0, 0, 0);   // hi . lo
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], xl, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2], xm, d0, 0, 0, 0);   // mid . mid
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3], xm, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2], xh, d0, 0, 0, 0);   // mid . hi
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3], xh, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], xm, d0, 0, 0, 0);   // hi . mid
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], xm, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], xh, d0, 0, 0, 0);   // hi . hi
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], xh, d1, 0, 0, 0);
        };

        // this wave's 8 info words of a tile (scalar load, fetched one tile ahead): count planes 0|1, 2|3 of run A, of run B,
        // then base A | len A << 9 | base B << 16 | len B << 25, ring slot of the tile's first entry
        const uint4 *my_info = info + 2 * w;
        uint4 iw = t0 < t1 ? my_info[(int64_t)t0 * kStInfoQuads] : uint4{0, 0, 0, 0};
        uint4 iv = t0 < t1 ? my_info[(int64_t)t0 * kStInfoQuads + 1] : uint4{0, 0, 0, 0};
        ST_BARRIER();
        for (int s = t0; s <= t1; ++s) {
            if (s < t1) {
                const uint4 cur = iw, cuv = iv;
                if (s + 1 < t1) {
                    iw = my_info[(int64_t)(s + 1) * kStInfoQuads];
                    iv = my_info[(int64_t)(s + 1) * kStInfoQuads + 1];
                }
#ifdef TGNN_ST_DEBUG
                if (w == 0) {
                    const int e_tile = tile_ent_ptr[s], e_next = tile_ent_ptr[s + 1];
                    int bad = 0;
                    for (int p = lane; p < e_next - e_tile; p += 64) {
                        const unsigned word = ent[e_tile + p];
                        if (word == kStPadWord) continue;
                        const unsigned pos = (unsigned)(e_tile + p) % kStRingSlots, xs = (word >> 4) & 7u;
                        for (unsigned q = 0; q < 8; ++q) {
                            const float4 got = *reinterpret_cast<const float4 *>(lds + (pos << 7) + ((q ^ xs) << 4));
                            const float4 want = *reinterpret_cast<const float4 *>(h + (size_t)(word >> 7) * 32 + 4 * q);
                            if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) ++bad;
                        }
                    }
                    for (int dl = 1; dl < 64; dl <<= 1) bad += __shfl_xor(bad, dl, 64);
                    if (lane == 0 && blockIdx.x < 512 && s - t0 < 64) g_st_dbg[blockIdx.x * 64 + (s - t0)] = bad;
                }
#endif
                f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, e0 = d0, e1 = d0;
                float af[8];
                const unsigned slot0 = cuv.y & 0xffffu;
                const unsigned base_a = cuv.x & 0x1ffu, len_a = (cuv.x >> 9) & 0x7fu, base_b = (cuv.x >> 16) & 0x1ffu, len_b = cuv.x >> 25;
                if (len_a) {
                    unsigned b = slot0 + base_a;
                    b = b >= (unsigned)kStRingSlots ? b - kStRingSlots : b;
                    run_sum(cur.x, cur.y, b, len_a, af);
                    if (root0) run_mfma(af, wa, e0, e1);
                    else run_mfma(af, wa, d0, d1);
                }
                if (len_b) {
                    unsigned b = slot0 + base_b;
                    b = b >= (unsigned)kStRingSlots ? b - kStRingSlots : b;
                    run_sum(cur.z, cur.w, b, len_b, af);
                    if (root1) run_mfma(af, wb, e0, e1);
                    else run_mfma(af, wb, d0, d1);
                }
                unsigned char *pb = lds + kStOffPart + ((s - t0) & 1) * kStPartBuf;
                *reinterpret_cast<f32x4 *>(pb + (w * 2 + 0) * 1024 + lane * 16) = d0;
                *reinterpret_cast<f32x4 *>(pb + (w * 2 + 1) * 1024 + lane * 16) = d1;
                if (root0 || root1) {
                    *reinterpret_cast<f32x4 *>(pb + (8 * 2 + 0) * 1024 + lane * 16) = e0;
                    *reinterpret_cast<f32x4 *>(pb + (8 * 2 + 1) * 1024 + lane * 16) = e1;
                }
            }
            ST_BARRIER();
        }
    } else if (wave < kStMult + kStDma) {
        // =================================================================== loader waves
        const int d = wave - kStMult;
        // chunks (8 entries = 1 KB of rows) are numbered over the whole stream: chunk c sits in ring chunk c % 96
        const int c_first = tile_ent_ptr[t0] >> 3, c_end = tile_ent_ptr[t1] >> 3;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(h), 0, (int)h_bytes, 0x00020000);
        const unsigned lane_piece = (unsigned)(lane & 7) << 4;   // ^ the entry word's permutation bits = source piece
        // local group k <-> the 8 chunks from c_first + 8 (2 k + d) on (64 entries)
        const int n_groups = (c_end - c_first + 7) >> 3;
        const int k_count = n_groups > d ? (n_groups - d + 1) >> 1 : 0;
        // entry words of a group: by LDS-DMA too, into this wave's 4-slot ring -- no register is the destination of a load
        // hipcc does not know of (it copies loop-carried registers at will: a copy taken before the data has landed is
        // stale), and every VMEM instruction of the wave is of one kind, so vmcnt counts them in order
        unsigned char *idx_ring = lds + kStOffIdx + d * 4 * 256;
        auto idx_load = [&](int k) {
            int g = 2 * k + d;
            g = g < n_groups ? g : (n_groups > 0 ? n_groups - 1 : 0);
            const unsigned *p = ent + ((int64_t)c_first + (int64_t)g * 8) * 8 + lane;
            __builtin_amdgcn_global_load_lds(p, (lds_void_t *)(idx_ring + (k & 3) * 256), 4, 0, 0);
        };
        const unsigned idx_addr0 = (unsigned)(kStOffIdx + d * 4 * 256 + ((lane >> 3) << 2));
        auto issue_group = [&](int k) {
            const int c0 = c_first + (2 * k + d) * 8;
            unsigned sr[8];
            const unsigned ia = idx_addr0 + (unsigned)((k & 3) * 256);
            // (asm: an ordinary LDS read behind a pending LDS-DMA makes hipcc drain vmcnt(0))
            asm volatile(
                "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:32\n\tds_read_b32 %2, %8 offset:64\n\t"
                "ds_read_b32 %3, %8 offset:96\n\tds_read_b32 %4, %8 offset:128\n\tds_read_b32 %5, %8 offset:160\n\t"
                "ds_read_b32 %6, %8 offset:192\n\tds_read_b32 %7, %8 offset:224\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(sr[0]), "=&v"(sr[1]), "=&v"(sr[2]), "=&v"(sr[3]), "=&v"(sr[4]), "=&v"(sr[5]), "=&v"(sr[6]), "=&v"(sr[7])
                : "v"(ia)
                : "memory");
            int ring = c0 % kStRingChunks;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned voff = sr[i] ^ lane_piece;    // row offset | (piece position ^ the row's permutation) << 4
                const int dst = c0 + i < c_end ? ring : kStRingChunks;   // past the range: the dump chunk
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(lds + dst * 1024), 16, voff, 0, 0, 0);
                ring = ring + 1 == kStRingChunks ? 0 : ring + 1;
            }
        };
        idx_load(0);
        idx_load(1);
        idx_load(2);
        st_wait_vmcnt<0>();
        int k = 0;
        // issues groups while the ring has room (nothing at or behind chunk `live` may be overwritten), then waits until
        // every chunk below `need` has landed
        auto ensure = [&](int live, int need) {
            for (;;) {
                if (k >= k_count) break;
                const int c_last = c_first + (2 * k + d) * 8 + 7;
                if (c_last - live >= kStRingChunks) break;
                if (k >= 3) st_wait_vmcnt<18>();             // this group's entry words (loaded behind group k - 3)
                issue_group(k);
                idx_load(k + 3);
                ++k;
            }
            if (need > c_first && k > 0) {
                const int g_need = (need - 1 - c_first) >> 3;                    // group of chunk need - 1
                const int k_need = g_need >= d ? (g_need - d) >> 1 : -1;        // last local group at or below it
                if (k_need >= 0) st_wait_groups_behind(k - 1 - (k_need < k ? k_need : k - 1));
            }
        };
        ensure(c_first, tile_ent_ptr[t0 + (t0 < t1)] >> 3);
        ST_BARRIER();
        for (int s = t0; s <= t1; ++s) {
            if (s + 1 < t1) ensure(tile_ent_ptr[s] >> 3, tile_ent_ptr[s + 2] >> 3);
            ST_BARRIER();
        }
        st_wait_vmcnt<0>();
    } else {
        // =================================================================== epilogue waves
        const int e = wave - kStMult - kStDma;               // M block: channels 16 e + 4 fq + r of row fj
        const float4 bias4 = *reinterpret_cast<const float4 *>(bias + 16 * e + 4 * fq);
        double bs[4] = {0, 0, 0, 0}, bq[4] = {0, 0, 0, 0};
        float invd_next = t0 < t1 ? inv_deg[(int64_t)t0 * 16 + fj] : 0.f;
        ST_BARRIER();
        for (int s = t0; s <= t1; ++s) {
            const float invd = invd_next;
            if (s < t1) invd_next = inv_deg[(int64_t)s * 16 + fj];
            float4 o = {0.f, 0.f, 0.f, 0.f};
            const int64_t v = (int64_t)(s - 1) * 16 + fj;
            const bool valid = s > t0 && v < n;
            if (s > t0) {
                const unsigned char *pb = lds + kStOffPart + ((s - 1 - t0) & 1) * kStPartBuf + e * 1024 + lane * 16;
                f32x4 acc = *reinterpret_cast<const f32x4 *>(pb);
#pragma unroll
                for (int w = 1; w < kStMult; ++w) acc += *reinterpret_cast<const f32x4 *>(pb + w * 2048);
                const f32x4 rt = *reinterpret_cast<const f32x4 *>(pb + 8 * 2048);
                o.x = fmaf(acc[0], invd, rt[0]) + bias4.x;
                o.y = fmaf(acc[1], invd, rt[1]) + bias4.y;
                o.z = fmaf(acc[2], invd, rt[2]) + bias4.z;
                o.w = fmaf(acc[3], invd, rt[3]) + bias4.w;
                if (act == TGNN_ACT_LEAKY_RELU) { o.x = leakyf_(o.x); o.y = leakyf_(o.y); o.z = leakyf_(o.z); o.w = leakyf_(o.w); }
                if (valid) {
                    bs[0] += (double)o.x; bq[0] += (double)o.x * (double)o.x;
                    bs[1] += (double)o.y; bq[1] += (double)o.y * (double)o.y;
                    bs[2] += (double)o.z; bq[2] += (double)o.z * (double)o.z;
                    bs[3] += (double)o.w; bq[3] += (double)o.w * (double)o.w;
                    *reinterpret_cast<float4 *>(out + v * 32 + 16 * e + 4 * fq) = o;
                }
            }
            ST_BARRIER();
        }
        if (bn_partial) {
            // lanes (fj, fq): fold the 16 row positions fj (xor butterfly inside a 16-lane row: fixed order)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int dl = 1; dl < 16; dl <<= 1) {
                    bs[r] += __shfl_xor(bs[r], dl, 64);
                    bq[r] += __shfl_xor(bq[r], dl, 64);
                }
            }
            if (fj == 0) {
                double *row = bn_partial + (int64_t)blockIdx.x * 64;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    row[16 * e + 4 * fq + r] = bs[r];
                    row[32 + 16 * e + 4 * fq + r] = bq[r];
                }
            }
        }
    }
#ifdef TGNN_ST_TIMING
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *o = g_st_time + ((size_t)blockIdx.x * 12 + wave) * 4;
        o[0] = t_work; o[1] = t_wait; o[2] = t_steps; o[3] = __builtin_readcyclecounter() - t_begin;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The stream structure, from the adjacency CSR + the edge types in CSR order (once per layout):
//   tile_ent_ptr [tiles + 1]   entries before every tile (multiples of 8: a tile starts a 1 KB chunk)
//   ent          [..]          entry words: source row << 7 | X << 4, X = the destination row's piece permutation (header);
//                              per tile: runs 0..T (run T = the tile's own rows), inside a run the destination rows
//                              ascending, a row's edges of that type in CSR order; padding = kStPadWord
//   info         [tiles][8]    2 x uint4 per multiplying wave w (runs A = w, B = w + 8): count planes 0|1, 2|3 of A, of B (bit j of
//                              plane k = bit k of row j's edge count; planes k, k+1 in one word); base A | len A << 9 |
//                              base B << 16 | len B << 25 (first entry of the run inside the tile, entries of the run), ring
//                              slot of the tile's first entry | padded entry count << 16, 0, 0
//   inv_deg      [16 tiles]    1 / max(in-degree, 1)
// result: [0] largest padded entry count of a tile, [1] largest number of same-type in-edges of a row, [2] 1 = built
// ------------------------------------------------------------------------------------------
__global__ void stream_count_kernel(const int *__restrict__ rowptr, int64_t n, int n_tiles, int *__restrict__ cnt,
                                    const int *__restrict__ gate) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_tiles) return;
    int c = 0;
    if (k < n_tiles && !(gate && !*gate)) {
        const int64_t v0 = (int64_t)k * 16, v1 = v0 + 16 < n ? v0 + 16 : n;
        c = (rowptr[v1] - rowptr[v0]) + (int)(v1 - v0);
        c = (c + 7) & ~7;
    }
    cnt[k] = c;
}

// one block = 64 rows = 4 tiles, one thread per row (the 16 threads of a tile are lanes of one wavefront)
__global__ __launch_bounds__(64) void stream_fill_kernel(const int *__restrict__ rowptr, const int *__restrict__ col_src,
                                                         const int *__restrict__ col_type, int64_t n, int n_tiles,
                                                         int n_types_host, const int *__restrict__ n_types_dev,
                                                         const int *__restrict__ tile_ent_ptr, unsigned *__restrict__ ent,
                                                         uint4 *__restrict__ info, float *__restrict__ inv_deg,
                                                         int *__restrict__ result, const int *__restrict__ gate) {
    if (gate && !*gate) return;
    const int n_types = n_types_dev ? *n_types_dev : n_types_host;
    if (n_types > kStMaxTypes) return;                       // (result[2] stays 0)
    __shared__ unsigned short cnt[64][17];                   // edges of row x type (odd stride); later: the write cursor
    __shared__ unsigned short pos[64][17];                   // first entry of (row, type) inside the tile
    __shared__ unsigned run_base[4][16], run_len[4][16], run_p01[4][16], run_p23[4][16];
    const int tid = threadIdx.x, k4 = tid >> 4, i = tid & 15;
    const int tile = blockIdx.x * 4 + k4;
    const int64_t row = (int64_t)tile * 16 + i;
    const int n_runs = n_types + 1;
    for (int t = 0; t < 16; ++t) cnt[tid][t] = 0;
    int e0 = 0, e1 = 0;
    if (row < n) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
    int maxmult = 0;
    for (int e = e0; e < e1; ++e) {
        const int t = col_type[e];
        if ((unsigned)t < (unsigned)n_types) {
            const int c = ++cnt[tid][t];
            maxmult = c > maxmult ? c : maxmult;
        }
    }
    cnt[tid][n_types] = row < n ? 1 : 0;                     // root run: the row itself
    unsigned running = 0;
    for (int t = 0; t < 16; ++t) {
        const unsigned c = t < n_runs ? cnt[tid][t] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int dl = 1; dl < 16; dl <<= 1) {
            const unsigned up = __shfl_up(incl, dl, 16);
            if (i >= dl) incl += up;
        }
        const unsigned total = __shfl(incl, 15, 16);
        pos[tid][t] = (unsigned short)(running + incl - c);
        const unsigned long long b0 = __ballot((c & 1u) != 0), b1 = __ballot((c & 2u) != 0), b2 = __ballot((c & 4u) != 0),
                                 b3 = __ballot((c & 8u) != 0);
        if (i == 0) {
            run_base[k4][t] = running;
            run_len[k4][t] = total;
            run_p01[k4][t] = ((unsigned)(b0 >> (16 * k4)) & 0xffffu) | (((unsigned)(b1 >> (16 * k4)) & 0xffffu) << 16);
            run_p23[k4][t] = ((unsigned)(b2 >> (16 * k4)) & 0xffffu) | (((unsigned)(b3 >> (16 * k4)) & 0xffffu) << 16);
        }
        running += total;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tile < n_tiles) {
        const int t_e0 = tile_ent_ptr[tile], t_e1 = tile_ent_ptr[tile + 1];
        const unsigned xs = (2u * ((i >> 1) & 3)) << 4;
        for (int t = 0; t < 16; ++t) cnt[tid][t] = 0;
        for (int e = e0; e < e1; ++e) {
            const int t = col_type[e];
            if ((unsigned)t < (unsigned)n_types) {
                const int c = cnt[tid][t]++;
                ent[t_e0 + pos[tid][t] + c] = ((unsigned)col_src[e] << 7) | xs;
            }
        }
        if (row < n) ent[t_e0 + pos[tid][n_types]] = ((unsigned)row << 7) | xs;
        for (int p = (int)running + i; p < t_e1 - t_e0; p += 16) ent[t_e0 + p] = kStPadWord;
        if (i < kStMult) {
            const int ra = i, rb = i + kStMult;
            uint4 w, v;
            w.x = ra < n_runs ? run_p01[k4][ra] : 0u;
            w.y = ra < n_runs ? run_p23[k4][ra] : 0u;
            w.z = rb < n_runs ? run_p01[k4][rb] : 0u;
            w.w = rb < n_runs ? run_p23[k4][rb] : 0u;
            const unsigned ba = ra < n_runs ? run_base[k4][ra] : 0u, la = ra < n_runs ? run_len[k4][ra] : 0u;
            const unsigned bb = rb < n_runs ? run_base[k4][rb] : 0u, lb = rb < n_runs ? run_len[k4][rb] : 0u;
            v.x = (ba & 0x1ffu) | ((la & 0x7fu) << 9) | ((bb & 0x1ffu) << 16) | ((lb & 0x7fu) << 25);
            v.y = ((unsigned)t_e0 % (unsigned)kStRingSlots) | ((unsigned)(t_e1 - t_e0) << 16);
            v.z = 0; v.w = 0;
            info[(int64_t)tile * kStInfoQuads + 2 * i] = w;
            info[(int64_t)tile * kStInfoQuads + 2 * i + 1] = v;
        }
        float r = 1.0f;
        if (row < n) r = 1.0f / (float)(e1 - e0 > 0 ? e1 - e0 : 1);
        inv_deg[row] = r;
        atomicMax(&result[1], maxmult);
        if (i == 0) {
            atomicMax(&result[0], t_e1 - t_e0);
            if (tile == 0) result[2] = 1;
        }
    }
}

int launch_nnconv_stream(const float *h, int64_t n_src_rows, const int32_t *tile_ent_ptr, const uint32_t *ent_src,
                         const uint32_t *info, const float *inv_deg, const float *wimg, int32_t n_types, const float *bias,
                         int64_t n_nodes, int32_t act, float *out, double *bn_partial, int32_t *n_partials_host,
                         int reserve_cus, hipStream_t s) {
    auto kern = nnconv32_stream_kernel;
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(kern, kStLdsBytes, site));
    int dev = 0, cus = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    TGNN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t blocks = cus - reserve_cus;
    if (blocks > TGNN_BN_MAX_PARTIALS) blocks = TGNN_BN_MAX_PARTIALS;
    if (blocks > n_tiles) blocks = n_tiles;
    if (blocks >= 8) blocks &= ~7;
    if (blocks < 1) blocks = 1;
    kern<<<(unsigned)blocks, kStWaves * 64, kStLdsBytes, s>>>(h, (unsigned)(n_src_rows * 128), tile_ent_ptr, ent_src,
                                                              reinterpret_cast<const uint4 *>(info), inv_deg, wimg, n_types,
                                                              bias, n_nodes, act, out, bn_partial);
    if (n_partials_host) *n_partials_host = (int32_t)blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// exclusive scan of graph_prep.hip
void exclusive_scan_i32_shared(const int *in, int *out, int64_t n, int *ws, hipStream_t s);
size_t scan_ws_ints_shared(int64_t n);

int nnconv_stream_build_gated(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type, int64_t n_nodes,
                              int32_t n_types, const int32_t *n_types_dev, const int32_t *gate, int32_t *tile_ent_ptr,
                              uint32_t *ent_src, uint32_t *info, float *inv_deg, int32_t *result, void *ws, hipStream_t s) {
    const int64_t n_tiles = (n_nodes + 15) / 16;
    TGNN_CHECK_HIP(hipMemsetAsync(result, 0, 4 * sizeof(int32_t), s));
    stream_count_kernel<<<(unsigned)((n_tiles + 1 + 255) / 256), 256, 0, s>>>(rowptr, n_nodes, (int)n_tiles, tile_ent_ptr, gate);
    exclusive_scan_i32_shared(tile_ent_ptr, tile_ent_ptr, n_tiles + 1, static_cast<int *>(ws), s);
    stream_fill_kernel<<<(unsigned)((n_tiles + 3) / 4), 64, 0, s>>>(rowptr, col_src, col_type, n_nodes, (int)n_tiles, n_types,
                                                                    n_types_dev, tile_ent_ptr, ent_src,
                                                                    reinterpret_cast<uint4 *>(info), inv_deg, result, gate);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

}  // namespace tgnn

using namespace tgnn;

#ifdef TGNN_ST_TIMING
extern "C" int tgnn_debug_stream_time(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tgnn::g_st_time), sizeof(unsigned long long) * 512 * 12 * 4);
}
#endif
#ifdef TGNN_ST_DEBUG
extern "C" int tgnn_debug_stream_ring(int *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tgnn::g_st_dbg), sizeof(int) * 512 * 64);
}
#endif

extern "C" void tgnn_nnconv_stream_limits(int32_t *max_types, int32_t *max_tile_entries, int32_t *max_multiplicity) {
    if (max_types) *max_types = kStMaxTypes;
    if (max_tile_entries) *max_tile_entries = kStMaxTileChunks * 8;
    if (max_multiplicity) *max_multiplicity = kStMaxMult;
}

extern "C" int64_t tgnn_nnconv_stream_max_entries(int64_t n_nodes, int64_t n_edges) {
    const int64_t n_tiles = (n_nodes + 15) / 16;
    return n_edges + n_nodes + 7 * n_tiles + 256;            // padding to 8 per tile + the loaders' read-ahead
}

extern "C" size_t tgnn_nnconv_stream_scan_ws_bytes(int64_t n_nodes) {
    return scan_ws_ints_shared((n_nodes + 15) / 16 + 1) * 4 + 256;
}

extern "C" int tgnn_nnconv_stream_build(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type,
                                        int64_t n_nodes, int32_t n_types, const int32_t *n_types_dev, int32_t *tile_ent_ptr,
                                        uint32_t *ent_src, uint32_t *info, float *inv_deg, int32_t *result, void *ws,
                                        size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(rowptr && tile_ent_ptr && ent_src && info && inv_deg && result, "null pointer");
    TGNN_CHECK_ARG(n_nodes >= 1 && n_nodes < (int64_t(1) << 24), "node count");
    TGNN_CHECK_ARG(((uintptr_t)info % 16) == 0, "alignment of info");
    TGNN_CHECK_ARG(ws && ws_bytes >= tgnn_nnconv_stream_scan_ws_bytes(n_nodes), "workspace");
    return nnconv_stream_build_gated(rowptr, col_src, col_type, n_nodes, n_types, n_types_dev, nullptr, tile_ent_ptr, ent_src,
                                     info, inv_deg, result, ws, static_cast<hipStream_t>(stream));
}

extern "C" int tgnn_nnconv_mean_stream_fwd(const float *h, int64_t n_src_rows, const int32_t *tile_ent_ptr,
                                           const uint32_t *ent_src, const uint32_t *info, const float *inv_deg,
                                           const float *wtab, int32_t n_types, const float *root, const float *bias,
                                           int64_t n_nodes, int32_t c, int32_t act, float *out, float *wimg_scratch,
                                           double *bn_partial, int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && c == 32, "the stream NNConv kernel is built for network_width 32");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && tile_ent_ptr && ent_src && info && inv_deg && root && bias && out && wimg_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(n_types <= kStMaxTypes, "too many edge types for the stream kernel");
    TGNN_CHECK_ARG(n_src_rows >= n_nodes && n_src_rows < (int64_t(1) << 24), "source rows (dense [rows][32] within 2 GB)");
    TGNN_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                       ((uintptr_t)wimg_scratch % 16) == 0 && ((uintptr_t)info % 16) == 0, "alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_nnconv_weight_image(wtab, &root, n_types, 1, wimg_scratch, s);
    return launch_nnconv_stream(h, n_src_rows, tile_ent_ptr, ent_src, info, inv_deg, wimg_scratch, n_types, bias, n_nodes, act,
                                out, bn_partial, n_partials_host, 0, s);
}
