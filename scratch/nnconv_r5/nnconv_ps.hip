// NNConv(aggr="mean"), network_width 32: the type-column formulation of nnconv_cols.hip on PRE-SPLIT source rows streamed
// through LDS -- no arithmetic on the gathered data outside the matrix pipe.
//
// Reference semantics: GraphConv.forward (/root/reference/graph_networks/layers/edge_conv.py:24-27) over PyG 1.3.2 NNConv:
//     out[v] = mean_{e: dst_e = v} h[src_e] . W_{type_e}  +  h[v] . root + bias  (+ LeakyReLU)
//
// What bounded the column kernel of rounds 1-4 (profiles/r04_pmc_nnconv_raw.txt, profiles/r05_gather_ceiling.txt):
//   * its gathers put ONE ROW PER LANE (the matrix operand's lane map): 16 cache lines per quarter wave, and two
//     instructions per column however few of its 16 slots are filled (29 % on the benchmark's layouts): ~25 us of the CU's
//     vector-memory path per launch, against 14 us for the same rows fetched whole (8 lanes x 16 B per row);
//   * per type run 8 adds per column and a 28-instruction fp32 -> fp16-pair split in front of 6 matrix instructions: 1 190
//     vector instructions per 16-row tile against 84 matrix instructions.
// Here:
//   * the producer of h (the merge / BatchNorm-apply kernels) leaves every row ALSO as an fp16 pair, scaled by a power of two
//     (row = hi[32] | lo[32], 128 bytes: tgnn_rows_split16): a gathered row IS a matrix operand;
//   * per 16-row tile the source rows of all its columns form a dense ENTRY stream (column after column, a column's filled
//     slots in row order; the tile's own rows = the root column last), fetched 8 whole rows per instruction by LDS-DMA
//     (global_load_lds_dwordx4) into a ring of 1 KB chunks per wave, a few chunks ahead of their use;
//   * a column is a 16-bit occupancy mask: lane (row n, k-group kg) finds its entry by a population count, reads its two
//     16-byte operand pieces from the ring (empty slot: a row of zeros) and EVERY column goes to the matrix pipe on its own
//     (3 terms x 2 output halves = 6 x v_mfma_f32_16x16x32_f16): the sum over a row's edges of one type, the sum over the
//     types and the mean's numerator are all the accumulator.  210 matrix instructions per tile instead of 84, ~420 vector
//     instructions instead of 1 190, 24 vector-memory instructions instead of 100.
// Ring image: entry q (position in the wave's stream mod ring size) occupies 128 bytes; its source piece c (16 bytes; pieces
// 0-3 = hi of k-group 0-3, 4-7 = lo) sits at slot (c + q) & 7, so that the 16 lanes of one ds_read_b128 lane group -- 8
// k-group-kg lanes and 8 k-group-(kg+1) lanes on consecutive entries -- fall on 16 different bank quads.  LDS-DMA writes
// lane-linear, so the loader permutes the SOURCE piece: lane (r = l / 8, j = l % 8) fetches piece (j - r) & 7 of entry r.
#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = tgnn_f16x8;
using u32x4 = tgnn_u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;

void exclusive_scan_i32_shared(const int *in, int *out, int64_t n, int *ws, hipStream_t s);   // graph_prep.hip
size_t scan_ws_ints_shared(int64_t n);

// col_word: bits 0-15 occupancy mask | 16-19 type (the root column: T) | 20-23 type of the run BEHIND this column's run (what
// the kernel fetches ahead; behind a tile's root run: the next tile's first run) | 24-25 chunks of the entry stream this column
// needs beyond the column before it | 26-31 ring position of its first entry (its absolute entry number mod kPsRingEnt)
#ifndef TGNN_PS_WAVES
#define TGNN_PS_WAVES 12
#endif
#ifndef TGNN_PS_RING
#define TGNN_PS_RING 6
#endif
constexpr int kPsWaves = TGNN_PS_WAVES;
constexpr int kPsRing = TGNN_PS_RING;      // chunks of a wave's ring: chunk k of the stream lives in ring chunk k % kPsRing
constexpr int kPsRingEnt = kPsRing * 8;    // (<= 64: the ring position has 6 bits in a column word)
static_assert(kPsRingEnt <= 64 && kPsRing >= 4 && kPsRing <= 8, "ring size");

// ------------------------------------------------------------------------------------------
// structure: from the type columns of tgnn_nnconv_cols_build (graph_prep.hip)
// ------------------------------------------------------------------------------------------
__global__ void ps_count_kernel(const int *__restrict__ rowptr, int64_t n, int64_t n_tiles, int *__restrict__ tile_cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { tile_cnt[t] = 0; return; }
    const int64_t r0 = t * 16, r1 = r0 + 16 < n ? r0 + 16 : n;
    const int cnt = rowptr[r1] - rowptr[r0] + (int)(r1 - r0);
    tile_cnt[t] = (cnt + 7) & ~7;
}

// one block = 4 tiles, 16 threads per tile (thread i = slot i of every column)
__global__ __launch_bounds__(64) void ps_fill_kernel(const int *__restrict__ tile_col_ptr, const int *__restrict__ col_meta,
                                                     const int *__restrict__ col_src, const int *__restrict__ tile_ent_ptr,
                                                     int64_t n, int64_t n_tiles, unsigned *__restrict__ col_word,
                                                     int *__restrict__ ent_src) {
    const int tid = threadIdx.x, k = tid >> 4, i = tid & 15;
    const int64_t tile = (int64_t)blockIdx.x * 4 + k;
    const bool live = tile < n_tiles;
    const int c0 = live ? tile_col_ptr[tile] : 0, c1 = live ? tile_col_ptr[tile + 1] : 0;
    const int e0 = live ? tile_ent_ptr[tile] : 0, e1 = live ? tile_ent_ptr[tile + 1] : 0;
    const int64_t row = tile * 16 + i;
    int pos = 0;
    const int n_cols = c1 - c0;
    int max_cols = n_cols;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) max_cols = max(max_cols, __shfl_xor(max_cols, d, 64));
    for (int cc = 0; cc < max_cols; ++cc) {                  // (all 64 lanes walk together: the ballot below is a whole-wave one)
        const bool in = cc < n_cols;
        const int c = c0 + cc;
        const int meta = in ? col_meta[c] : 0;
        const bool root = (meta & (1 << 10)) != 0;
        const int s = in ? col_src[(int64_t)c * 16 + i] : -1;
        const bool valid = in && (root ? row < n : s >= 0);
        const unsigned long long m64 = __ballot(valid);
        const unsigned m = (unsigned)(m64 >> (16 * k)) & 0xffffu;
        const int cnt = __popc(m);
        if (valid) ent_src[e0 + pos + __popc(m & ((1u << i) - 1u))] = root ? (int)row : s;
        if (in && i == 0) {
            unsigned w = m | ((unsigned)(meta & 0xf) << 16);
            const int need = (pos + cnt - 1) >> 3, before = (pos - 1) >> 3;       // (arithmetic shifts: -1 in front of the tile's first chunk)
            w |= (unsigned)(need - before) << 24;
            w |= (unsigned)((e0 + pos) % kPsRingEnt) << 26;
            int c2 = c + 1;
            while (c2 < c1 && !(col_meta[c2] & (1 << 8))) ++c2;
            // (behind the root run: the first run of the next tile -- the wave goes on there, or stops and never uses it)
            const int nt = (c2 < c1 || tile + 1 < n_tiles) ? (col_meta[c2] & 0xf) : 0;
            w |= (unsigned)nt << 20;
            col_word[c] = w;
        }
        pos += cnt;
    }
    // padding up to the tile's chunk boundary: any row that exists (it is fetched and never read)
    if (live)
        for (int p = e0 + pos + i; p < e1; p += 16) ent_src[p] = (int)(tile * 16);
    // the loader runs a few chunks past the end of a wave's stream: behind the last tile that is this slack (row 0)
    if (blockIdx.x == 0) {
        const int total = tile_ent_ptr[n_tiles];
        for (int p = tid; p < 512; p += 64) ent_src[total + p] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// rows as fp16 pairs: out row = hi[32] | lo[32] of s * h (s: a power of two with s * max |h| < 2^15)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_store4(const float4 &o, float s, void *__restrict__ hs, int64_t i4) {
    // i4 = index of the float4 in the packed [rows][32] array: row = i4 / 8, channels 4 (i4 % 8) ..
    const float a[4] = {o.x * s, o.y * s, o.z * s, o.w * s};
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (_Float16)a[j];
        lo[j] = (_Float16)(a[j] - (float)hi[j]);
    }
    using h4 = __attribute__((ext_vector_type(4))) _Float16;
    const h4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
    unsigned char *rowp = static_cast<unsigned char *>(hs) + (i4 >> 3) * 128 + (i4 & 7) * 8;
    *reinterpret_cast<h4 *>(rowp) = vh;
    *reinterpret_cast<h4 *>(rowp + 64) = vl;
}

__global__ void rows_split16_kernel(const float *__restrict__ h, int64_t n4, const unsigned *__restrict__ h_max,
                                    void *__restrict__ hs, float *__restrict__ scale_out) {
    const float s = pow2_scale_for(*h_max, 0);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        split_store4(reinterpret_cast<const float4 *>(h)[i], s, hs, i);
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void ps_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <class T>
__device__ __forceinline__ T ps_lds_ld(unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T *>((uintptr_t)addr);
}
// LDS-DMA by hand (hipcc drains vmcnt(0) in front of every LDS read behind a DMA it knows of): 64 lanes x 16 (4) bytes from
// base + this lane's offset to lds_dst + 16 (4) * lane; counted on vmcnt, waited for with ps_wait_vmcnt
__device__ __forceinline__ void ps_dma16(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void ps_dma4(const void *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

constexpr int kPsStage = 16 * 20;           // floats of the per-wave BatchNorm staging tile (nnconv_cols.hip)

struct PsLds {
    unsigned zero, idx, cw, ring, stage, total;
};
__host__ __device__ inline PsLds ps_lds_map(int n_types, int waves) {
    PsLds m;
    m.zero = (unsigned)(n_types + 1) * kWtTypeF16 * 4u;      // 128 bytes of zeros behind the weight image
    m.idx = m.zero + 256u;                                   // [waves][2][64] entry words
    m.cw = m.idx + (unsigned)waves * 512u;                   // [waves][2][64] column words
    m.ring = (m.cw + (unsigned)waves * 512u + 1023u) & ~1023u;
    m.stage = m.ring + (unsigned)waves * kPsRing * 1024u;
    m.total = m.stage + (unsigned)waves * kPsStage * 4u;
    const unsigned red = (unsigned)waves * 64u * 4u * 8u;    // the block's BatchNorm fold (aliases the image at the end)
    if (m.total < red) m.total = red;
    return m;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void nnconv32_ps_kernel(
    const void *__restrict__ hs, const int *__restrict__ tile_col_ptr, const unsigned *__restrict__ col_word,
    const int *__restrict__ tile_ent_ptr, const int *__restrict__ ent_src, const float *__restrict__ wimg, int n_types,
    const float *__restrict__ bias, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial,
    const float *__restrict__ h_scale, const unsigned *__restrict__ root_max, unsigned long long *__restrict__ stamp) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    // LOOK chunks are in flight beyond the last chunk of the column whose operands are being read (a column spans <= 3 chunks).
    // LOOK <= 5: the wait of the column before a batch's first chunk then also covers that batch's entry words (issued >= 8
    // chunks earlier, at most 3 + LOOK of them younger than what that wait covered)
    constexpr int LOOK = kPsRing - 3;
    static_assert(LOOK >= 1 && LOOK <= 5 && WAVES % 4 == 0, "shape");
    if (stamp && threadIdx.x == 0) atomicMin(stamp, wall_clock64());
    const PsLds L = ps_lds_map(n_types, WAVES);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t *)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;
    constexpr int kThreads = WAVES * 64;
    {   // weight image: straight copy (all loads of a thread issued before the first LDS store); the row of zeros
        float *wl = reinterpret_cast<float *>(lds_raw);
        const int n4 = (n_types + 1) * kWtTypeF16 / 4;
        for (int i = tid; i < n4; i += 4 * kThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + u * kThreads < n4 ? i + u * kThreads : n4 - 1;
                v[u] = reinterpret_cast<const float4 *>(wimg)[ii];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * kThreads < n4) reinterpret_cast<float4 *>(wl)[i + u * kThreads] = v[u];
        }
        if (tid < 64) reinterpret_cast<float *>(lds_raw + L.zero)[tid] = 0.f;
    }
    const float sx = *h_scale;
    const float unscale = 1.0f / (sx * nnconv_weight_scale(*root_max));   // (powers of two: exact)
    float *stg = reinterpret_cast<float *>(lds_raw + L.stage) + wave * kPsStage;

    // ---- this wave's run of 16-row tiles: as nnconv_cols.hip (shares follow the XCD, then the block, then the SIMD)
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
    const int k0 = __builtin_amdgcn_readfirstlane(tile_ent_ptr[t0]) >> 3;      // first chunk of the wave's entry stream

    const float4 bias0 = *reinterpret_cast<const float4 *>(bias + 4 * fq);
    const float4 bias1 = *reinterpret_cast<const float4 *>(bias + 16 + 4 * fq);
    double bs[2] = {0, 0}, bq[2] = {0, 0};
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing of the compiler's is in flight when the hand-counted DMAs start

    // ---- the loader half: chunks of 8 entries (chunk k -> ring chunk k % kPsRing), their source rows named by entry words
    // that arrive 64 at a time (two slots).  Every column asks for the chunks it needs beyond its predecessor's, so exactly
    // LOOK chunk DMAs are younger than a column's last chunk when its operands are read: ONE constant wait.  (The loader runs
    // up to LOOK chunks past the end of the wave's stream: rows of the next wave's tiles or the slack's row 0, never read.)
    const unsigned ring0 = lds0 + L.ring + (unsigned)wave * kPsRing * 1024u;
    const unsigned ring_end = ring0 + kPsRing * 1024u;
    const unsigned idx0 = lds0 + L.idx + (unsigned)wave * 512u;
    const unsigned cw0 = lds0 + L.cw + (unsigned)wave * 512u;
#ifdef TGNN_ABL_PSNOROT
    const unsigned dma_piece = (unsigned)(lane & 7) << 4;
#else
    const unsigned dma_piece = (unsigned)(((lane & 7) - (lane >> 3)) & 7) << 4;
#endif
    unsigned kpos = 0;                                       // chunks asked for so far (the next chunk's number in the wave's stream)
    unsigned ring_dst = ring0 + (unsigned)(k0 % kPsRing) * 1024u;   // ... its place in the ring
    const unsigned idx_lane = idx0 + ((unsigned)(lane >> 3) << 2);  // entry words: one per 8 lanes, two slots of 64 = 16 chunks
    unsigned src_next = 0;                                   // the next chunk's entry word (read one chunk ahead)
    auto idx_dma = [&](unsigned b) {                         // entry words [8 k0 + 64 b, + 64)
        ps_dma4(ent_src, (unsigned)(((k0 << 3) + (int)(b << 6) + lane) << 2), idx0 + (b & 1u) * 256u);
    };
    auto cw_dma = [&](unsigned b) {                          // column words [cbeg + 64 b, + 64)
        ps_dma4(col_word, (unsigned)((cbeg + (int)(b << 6) + lane) << 2), cw0 + (b & 1u) * 256u);
    };
    auto issue_chunk = [&]() {
#ifndef TGNN_ABL_PSNODMA
        ps_dma16(hs, (src_next << 7) + dma_piece, ring_dst);
#endif
        ring_dst += 1024u;
        ring_dst = ring_dst == ring_end ? ring0 : ring_dst;
        ++kpos;
        // the next chunk opens a batch (landed: see LOOK above): the slot of the batch just finished takes the one behind it
        if ((kpos & 7u) == 0u) idx_dma((kpos >> 3) + 1u);
        src_next = ps_lds_ld<unsigned>(idx_lane + ((kpos & 15u) << 5));
    };

    const unsigned lt_mask = (1u << fj) - 1u;
    const unsigned zero_addr = lds0 + L.zero;
    const unsigned w_lane = lds0 + (unsigned)lane * 16u;
    const unsigned piece0 = (unsigned)fq << 4;

    // ---- the column stream.  Outer loop: tiles; inner loop: a tile's edge columns, one per turn, the accumulators touched in ONE
    //      place (hipcc keeps them where they are).  Column words come 64 per DMA into two slots and are read one column ahead
    //      (every lane the same address); a column's six operands -- this lane's two pieces of its entry, the four fragments of
    //      its type -- are read from LDS side by side.
    if (cbeg < cend) {
        const unsigned t_bits = (unsigned)n_types << 16;
        idx_dma(0);
        idx_dma(1);
        cw_dma(0);
        cw_dma(1);
        ps_wait_vmcnt<0>();
        src_next = ps_lds_ld<unsigned>(idx_lane);
#pragma unroll
        for (int i = 0; i < LOOK; ++i) issue_chunk();
        unsigned cpos = 0;                                   // columns taken so far
        unsigned w_next_v = ps_lds_ld<unsigned>(cw0);
        const unsigned n_cols = (unsigned)(cend - cbeg);
        int64_t ctile = t0;
#pragma unroll 1
        while (cpos < n_cols) {
            // D^T tiles (channels 4 fq + r and 16 + 4 fq + r of row fj): the edge sum's hi.hi terms / its two small terms
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, e0 = d0, e1 = d0;
            unsigned degcnt = 0, valid;
            f16x8 wh0, wh1, wl0, wl1, xh, xl;
#pragma unroll 1
            for (;;) {
                const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)w_next_v);
                ++cpos;
                // (two words into a batch the slot of the batch before it is free and takes the batch behind this one; the word
                //  read now was asked for >= 62 columns earlier: landed long ago)
                if ((cpos & 63u) == 2u && cpos > 64u) cw_dma((cpos >> 6) + 1u);
                w_next_v = ps_lds_ld<unsigned>(cw0 + ((cpos & 127u) << 2));
                // the column's chunks
                const unsigned n_new = (w >> 24) & 3u;
                if (n_new > 0u) {
                    issue_chunk();
                    if (n_new > 1u) {
                        issue_chunk();
                        if (n_new > 2u) issue_chunk();
                    }
                }
                // this lane's entry: ring position of the column's first entry + the filled slots below this lane's
                valid = (w >> fj) & 1u;
                unsigned q = (unsigned)__builtin_popcount(w & lt_mask) + (w >> 26);
                if constexpr ((kPsRingEnt & (kPsRingEnt - 1)) == 0) q &= (unsigned)(kPsRingEnt - 1);
                else q = min(q, q - (unsigned)kPsRingEnt);   // (q < 2 ring sizes)
                unsigned addr = ring0 + (q << 7) + (((q << 4) + piece0) & 0x70u);
                addr = valid ? addr : zero_addr;
                const unsigned wa = w_lane + ((w >> 4) & 0xf000u);          // type * 4096 bytes
#ifndef TGNN_ABL_PSNOWAIT
                ps_wait_vmcnt<LOOK>();
#endif
#ifdef TGNN_ABL_PSNOLDS
                xh = __builtin_bit_cast(f16x8, u32x4{addr, addr, addr, addr}); xl = xh;
#else
                xh = __builtin_bit_cast(f16x8, ps_lds_ld<u32x4>(addr));
                xl = __builtin_bit_cast(f16x8, ps_lds_ld<u32x4>(addr ^ 64u));
#endif
                wh0 = __builtin_bit_cast(f16x8, ps_lds_ld<u32x4>(wa));
                wh1 = __builtin_bit_cast(f16x8, ps_lds_ld<u32x4>(wa + 1024u));
                wl0 = __builtin_bit_cast(f16x8, ps_lds_ld<u32x4>(wa + 2048u));
                wl1 = __builtin_bit_cast(f16x8, ps_lds_ld<u32x4>(wa + 3072u));
                if ((w & 0xf0000u) == t_bits) break;         // (scalar) the root column: behind the loop
                degcnt += valid;
#ifdef TGNN_ABL_PSNOMUL
                e0[0] += (float)xh[0] * (float)wl0[0] + (float)xl[1] * (float)wh1[0] + (float)wh0[0] * (float)wl1[2];
#else
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, xh, e0, 0, 0, 0);   // lo . hi
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, xh, e1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xh, d0, 0, 0, 0);   // hi . hi
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xh, d1, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xl, e0, 0, 0, 0);   // hi . lo
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xl, e1, 0, 0, 0);
#endif
            }
            // ---- the root column ends the tile
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, xh, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, xh, r1, 0, 0, 0);
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xl, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xl, r1, 0, 0, 0);
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xh, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xh, r1, 0, 0, 0);
            const float inv = unscale / (float)(degcnt > 0u ? degcnt : 1u);
            const int64_t v = ctile * 16 + fj;
            float4 o0, o1;
            o0.x = fmaf(d0[0] + e0[0], inv, fmaf(r0[0], unscale, bias0.x)); o0.y = fmaf(d0[1] + e0[1], inv, fmaf(r0[1], unscale, bias0.y));
            o0.z = fmaf(d0[2] + e0[2], inv, fmaf(r0[2], unscale, bias0.z)); o0.w = fmaf(d0[3] + e0[3], inv, fmaf(r0[3], unscale, bias0.w));
            o1.x = fmaf(d1[0] + e1[0], inv, fmaf(r1[0], unscale, bias1.x)); o1.y = fmaf(d1[1] + e1[1], inv, fmaf(r1[1], unscale, bias1.y));
            o1.z = fmaf(d1[2] + e1[2], inv, fmaf(r1[2], unscale, bias1.z)); o1.w = fmaf(d1[3] + e1[3], inv, fmaf(r1[3], unscale, bias1.w));
            if (act == TGNN_ACT_LEAKY_RELU) {
                o0.x = leakyf_(o0.x); o0.y = leakyf_(o0.y); o0.z = leakyf_(o0.z); o0.w = leakyf_(o0.w);
                o1.x = leakyf_(o1.x); o1.y = leakyf_(o1.y); o1.z = leakyf_(o1.z); o1.w = leakyf_(o1.w);
            }
            if (valid) {
                *reinterpret_cast<float4 *>(out + v * 32 + 4 * fq) = o0;
                *reinterpret_cast<float4 *>(out + v * 32 + 16 + 4 * fq) = o1;
            }
            if (bn_partial) {
                // column sums in fp64: transpose through the wave's own LDS tile, one 16-channel half at a time (nnconv_cols.hip)
                auto half_sums = [&](float4 o, double &sum, double &sq) {
                    o.x = valid ? o.x : 0.f; o.y = valid ? o.y : 0.f;
                    o.z = valid ? o.z : 0.f; o.w = valid ? o.w : 0.f;
                    *reinterpret_cast<float4 *>(stg + fj * 20 + 4 * fq) = o;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double val = (double)stg[(4 * fq + r) * 20 + fj];
                        sum += val;
                        sq += val * val;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                half_sums(o0, bs[0], bq[0]);
                half_sums(o1, bs[1], bq[1]);
            }
            ++ctile;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (nothing of this wave's is still on its way into the LDS that is reused below)

    // ---- BN partials of the block: lanes (fj, fq) -> channel 16 m + fj; fold fq, then the waves, in fixed order
    if (bn_partial) {
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds_raw);   // [WAVES][64 lanes][4]
        double *mine = red + ((int64_t)wave * 64 + lane) * 4;
        mine[0] = bs[0]; mine[1] = bs[1]; mine[2] = bq[0]; mine[3] = bq[1];
        __syncthreads();
        if (tid < 64) {                                      // tid = which * 32 + channel
            const int which = tid >> 5, ch = tid & 31, m2 = ch >> 4, j = ch & 15;
            double acc = 0;
            for (int w = 0; w < WAVES; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += red[((int64_t)w * 64 + q * 16 + j) * 4 + which * 2 + m2];
            bn_partial[(int64_t)blockIdx.x * 64 + tid] = acc;
        }
    }
    if (stamp && tid == 0) atomicMax(stamp + 1, wall_clock64());
}

int32_t nnconv_ps_max_types() {
    // the largest T whose LDS map fits
    int t = 0;
    while (t < 15 && ps_lds_map(t + 1, kPsWaves).total <= 160u * 1024u - 256u) ++t;   // (a column word has 4 bits for the type, the root's included)
    return t;
}

int launch_nnconv_ps(const void *hs, const int32_t *tile_col_ptr, const uint32_t *col_word, const int32_t *tile_ent_ptr,
                     const int32_t *ent_src, const float *wimg, int32_t n_types, const float *bias, int64_t n_nodes, int32_t act,
                     float *out, double *bn_partial, int32_t *n_partials_host, const float *h_scale, const unsigned *root_max,
                     hipStream_t s, unsigned long long *stamp) {
    constexpr int WAVES = kPsWaves;
    auto kern = nnconv32_ps_kernel<WAVES>;
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(kern, 160 * 1024, site));
    const PsLds L = ps_lds_map(n_types, WAVES);
    if (L.total > 160u * 1024u) {
        set_error("launch_nnconv_ps: %d edge types do not fit the LDS map", n_types);
        return TGNN_ERR_UNSUPPORTED;
    }
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t blocks = (n_tiles + 3) / 4;                      // one tile per SIMD before a second wave of a SIMD gets one
    constexpr int reserve = 32;                              // CUs left to the other chain of the two-stream forward (nnconv_cols.hip)
    int64_t cap = cus_minus(reserve);
    if (const int dbg = g_debug_block_cap[0].load(); dbg > 0) cap = dbg < device_cus() ? dbg : device_cus();
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks &= ~7;
    if (blocks < 1) blocks = 1;
    kern<<<(unsigned)blocks, WAVES * 64, L.total, s>>>(hs, tile_col_ptr, col_word, tile_ent_ptr, ent_src, wimg, n_types, bias,
                                                       n_nodes, act, out, bn_partial, h_scale, root_max, stamp);
    if (n_partials_host) *n_partials_host = (int32_t)blocks;
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int launch_nnconv_ps_build(const int32_t *rowptr, const int32_t *tile_col_ptr, const int32_t *col_meta, const int32_t *col_src,
                           int64_t n_nodes, int32_t *tile_ent_ptr, uint32_t *col_word, int32_t *ent_src, int *scan_ws,
                           hipStream_t s) {
    const int64_t n_tiles = (n_nodes + 15) / 16;
    ps_count_kernel<<<(unsigned)((n_tiles + 1 + 255) / 256), 256, 0, s>>>(rowptr, n_nodes, n_tiles, tile_ent_ptr);
    exclusive_scan_i32_shared(tile_ent_ptr, tile_ent_ptr, n_tiles + 1, scan_ws, s);
    ps_fill_kernel<<<(unsigned)((n_tiles + 3) / 4), 64, 0, s>>>(tile_col_ptr, col_meta, col_src, tile_ent_ptr, n_nodes, n_tiles,
                                                                col_word, ent_src);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

void launch_rows_split16(const float *h, int64_t n_rows, const unsigned *h_max, void *hs, float *scale_out, hipStream_t s) {
    const int64_t n4 = n_rows * 8;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    rows_split16_kernel<<<(unsigned)blocks, 256, 0, s>>>(h, n4, h_max, hs, scale_out);
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int32_t tgnn_nnconv_ps_max_types(void) { return nnconv_ps_max_types(); }
extern "C" int64_t tgnn_nnconv_ps_max_entries(int64_t n_nodes, int64_t n_edges) {
    // every edge + the tile's own rows + padding to whole chunks, + the loader's look-ahead slack
    return n_edges + ((n_nodes + 15) / 16) * (16 + 7) + 512;
}
extern "C" size_t tgnn_nnconv_ps_workspace_bytes(int64_t n_nodes) {
    return scan_ws_ints_shared((n_nodes + 15) / 16 + 1) * 4 + 256;
}

extern "C" int tgnn_nnconv_ps_build(const int32_t *rowptr, const int32_t *tile_col_ptr, const int32_t *col_meta,
                                    const int32_t *col_src, int64_t n_nodes, int32_t *tile_ent_ptr, uint32_t *col_word,
                                    int32_t *ent_src, void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && rowptr && tile_col_ptr && col_meta && col_src && tile_ent_ptr && col_word && ent_src, "arguments");
    if (!ws || ws_bytes < tgnn_nnconv_ps_workspace_bytes(n_nodes)) {
        set_error("tgnn_nnconv_ps_build: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    return launch_nnconv_ps_build(rowptr, tile_col_ptr, col_meta, col_src, n_nodes, tile_ent_ptr, col_word, ent_src,
                                  static_cast<int *>(ws), static_cast<hipStream_t>(stream));
}

extern "C" int tgnn_rows_split16(const float *h, int64_t n_rows, uint32_t *max_bits_scratch, void *hs, float *scale_out,
                                 tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(h && hs && max_bits_scratch && scale_out && n_rows >= 1, "arguments");
    TGNN_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)hs % 128) == 0, "alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TGNN_CHECK_HIP(hipMemsetAsync(max_bits_scratch, 0, 4, s));
    launch_absmax(h, n_rows * 32, max_bits_scratch, s);
    launch_rows_split16(h, n_rows, max_bits_scratch, hs, scale_out, s);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_nnconv_mean_ps_fwd(const float *h, int64_t n_src_rows, const int32_t *tile_col_ptr, const uint32_t *col_word,
                                       const int32_t *tile_ent_ptr, const int32_t *ent_src, const float *wtab, int32_t n_types,
                                       const float *root, const float *bias, int64_t n_nodes, int32_t act, float *out,
                                       float *wimg_scratch, void *hs_scratch, uint32_t *bounds_scratch, double *bn_partial,
                                       int32_t *n_partials_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_src_rows >= n_nodes && n_src_rows * 128 < (int64_t(1) << 31), "shape");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && tile_col_ptr && col_word && tile_ent_ptr && ent_src && root && bias && out && wimg_scratch && hs_scratch &&
                       bounds_scratch, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                       ((uintptr_t)wimg_scratch % 16) == 0 && ((uintptr_t)root % 16) == 0 && ((uintptr_t)hs_scratch % 128) == 0, "alignment");
    if (n_types > nnconv_ps_max_types()) {
        set_error("tgnn_nnconv_mean_ps_fwd: %d edge types do not fit the LDS map (max %d)", n_types, nnconv_ps_max_types());
        return TGNN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // bounds: [0] = max |h| over every row that can be gathered, [1] = max |root|, [2] = the rows' scale (a float)
    launch_forward_scales(bounds_scratch, 2, &root, 1, bounds_scratch + 1, nullptr, 0, nullptr, s);
    launch_absmax(h, n_src_rows * 32, bounds_scratch, s);
    launch_rows_split16(h, n_src_rows, bounds_scratch, hs_scratch, reinterpret_cast<float *>(bounds_scratch + 2), s);
    launch_nnconv_weight_image(wtab, &root, n_types, 1, wimg_scratch, s, bounds_scratch + 1);
    return launch_nnconv_ps(hs_scratch, tile_col_ptr, col_word, tile_ent_ptr, ent_src, wimg_scratch, n_types, bias, n_nodes, act, out,
                            bn_partial, n_partials_host, reinterpret_cast<const float *>(bounds_scratch + 2), bounds_scratch + 1, s,
                            nullptr);
}
