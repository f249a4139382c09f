// NNConv(aggr="mean"), network_width 32: the throughput kernel -- gathered rows as ONE stream through an LDS ring,
// one dense matrix product per edge type and 16 EDGES, the sum over a row's edges in the epilogue.
//
// Reference semantics: GraphConv.forward (/root/reference/graph_networks/layers/edge_conv.py:24-27) over PyG 1.3.2 NNConv:
//     out[v] = mean_{e: dst_e = v} h[src_e] . W_{type_e}  +  h[v] . root + bias  (+ LeakyReLU)
// i.e. message, scatter-mean, update -- and that is the order here too (nnconv_cols.hip sums first and multiplies later).
//
// What bounds this op on gfx950 is neither HBM nor LDS but the SIMDs' time: matrix-pipe cycles and every other issued
// instruction (~4.5 cycles each, scalar ones included) ADD UP per SIMD (rocprof: instructions = SIMD-quad-cycles of the
// launch; a version whose two waves per SIMD alternated matrix and address phases ran no faster).  So the design goal is
// the smallest (16 x matrix instructions + 4.5 x all others) per 16-row tile:
//   * rows arrive PRE-SPLIT into fp16 pairs (hi + lo, scaled by a power of two: split16_kernel; 128 bytes per row like
//     fp32): a gathered row IS a matrix operand, no conversion, no arithmetic on the data outside the matrix pipe;
//   * the source rows of a block's tiles are ONE stream of 128-byte rows ("entries": per 16-row tile its edge types in
//     order -- the tile's own rows last = the root run --, inside a type the destination rows ascending, a row's edges in
//     CSR order), fetched by two loader waves with LDS-DMA (buffer_load_dwordx4 ... lds: 8 whole rows per instruction,
//     8 consecutive lanes per row: the cheap shape for the CU's address path) into an 80 KB ring; the entry word IS the
//     buffer offset, the piece permutation one XOR with a lane constant; the loaders run two tiles ahead;
//   * 8 multiplying waves own the edge types round-robin (type t -> wave t & 7), their weight fragments (fp16 pairs, 32
//     VGPRs for two types) stay in registers.  16 consecutive entries of one type are one B operand as they lie in the
//     ring (a lane's address = first slot * 128 + a lane constant: 2 reads, 6 products, 2 stores per 16 edges): DENSE
//     blocks -- a type has ~12 edges per tile -- where the formulation by destination row (a column = every row's k-th edge
//     of the type) multiplies 3.6 x as many zero-padded columns; columns of a block beyond the type's edges hold whatever
//     the ring held, and nobody reads them.  The 16 x 32 messages go to a second ring ("Y", fp32, 64 KB);
//   * 2 epilogue waves (8 destination rows each, 8 lanes per row) add a row's messages in CSR order from a per-tile list of
//     Y slots, scale by 1 / max(deg, 1), add the root message, bias, LeakyReLU, store 128-byte rows, BatchNorm sums in fp64.
//   One block barrier per tile; the three roles work on tiles s+1.., s and s-1 of the block's contiguous tile range.
//
// Ring image: entry p of the stream (absolute position) sits in slot p mod 640; the eight 16-byte pieces of the k-th entry of
// a type run are stored at q ^ (k & 6) (the loader permutes the SOURCE piece per lane -- LDS-DMA writes lane-linear --, the
// permutation travels in the entry word); pieces g and g + 4 are kgroup g's hi and lo fragment: sixteen consecutive entries
// read by one ds_read_b128 lane group fall on sixteen different bank quads.
#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kStRingChunks = 80;                        // 1 KB chunks = 8 row slots each (a multiple of 8)
constexpr int kStRingSlots = kStRingChunks * 8;          // 640
constexpr int kStYSlots = 512;                           // message ring: 512 x 128 B (a power of two)
constexpr int kStMult = 8, kStDma = 2, kStEpi = 2, kStWaves = kStMult + kStDma + kStEpi;
constexpr int kStInfoWords = 64;                         // per tile: 4 words per multiplying wave, 9 words of row starts
constexpr int kStMaxPair = 512;                          // entries of two consecutive tiles (both rings hold two tiles)
constexpr int kStMaxTypes = 2 * kStMult - 1;             // T + 1 runs over 8 waves, two per wave
constexpr unsigned kStPadWord = 0xffffff80u;             // entry word beyond any buffer: the DMA fetches nothing

// LDS map (bytes)
constexpr int kStOffDump = kStRingChunks * 1024;         // 1 KB: chunks past the block's range land here
constexpr int kStOffInfo = kStOffDump + 1024;            // [8][64] dwords: info blocks of the tiles s - 1 .. s + 3 (ring of 8)
constexpr int kStOffIdx = kStOffInfo + 8 * 256;          // [2 loader waves][4][64] dwords: entry words of a group
constexpr int kStOffList = kStOffIdx + 2 * 4 * 256;      // [3][512] u16: entry -> message slot of the tiles s .. s + 2 (ring of 3)
constexpr int kStOffY = (kStOffList + 3 * 1024 + 1023) / 1024 * 1024;   // [512][32] fp32 messages
constexpr int kStLdsBytes = kStOffY + (kStYSlots + 1) * 128;   // (+ one row of zeros: slot 512)

template <int N>
__device__ __forceinline__ void st_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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
// LDS access by absolute byte address (`lds + offset` makes hipcc add the array's base -- a link-time zero -- with a vector
// instruction in front of every access; the callers fold the base into their lane constants once)
template <class T>
__device__ __forceinline__ T st_lds_ld(unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T *>((uintptr_t)addr);
}
template <class T>
__device__ __forceinline__ void st_lds_st(unsigned addr, const T &v) {
    *reinterpret_cast<__attribute__((address_space(3))) T *>((uintptr_t)addr) = v;
}

// LDS-DMA by hand (hipcc drains vmcnt(0) in front of every LDS read that follows a DMA it knows of): 64 lanes x 16 / 4 bytes
// from each lane's address to lds_dst + 16 / 4 * lane; counted on vmcnt, waited for by hand
__device__ __forceinline__ void st_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void st_dma4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#ifdef TGNN_ST_TIMING
__device__ unsigned long long g_st_time[512 * 12 * 4];   // [block][wave]: work cycles, barrier-wait cycles, steps, total
__device__ unsigned long long g_st_seg[512 * 12 * 4];    // [block][wave]: cycles of up to four segments of a step
#define ST_SEG(k) { const unsigned long long n_ = __builtin_readcyclecounter(); t_seg[k] += n_ - t_mark; t_mark = n_; }
#define ST_MARK() t_mark = __builtin_readcyclecounter();
#define ST_BARRIER()                                                            \
    {                                                                           \
        const unsigned long long a_ = __builtin_readcyclecounter();            \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
        const unsigned long long b_ = __builtin_readcyclecounter();            \
        t_work += a_ - t_last; t_wait += b_ - a_; t_last = b_; ++t_steps;       \
    }
#else
#define ST_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define ST_SEG(k)
#define ST_MARK()
#endif

__global__ __launch_bounds__(kStWaves * 64) void nnconv32_stream_kernel(
    const void *__restrict__ hs, unsigned h_bytes, const float *__restrict__ hs_scale, const int *__restrict__ tile_ent_ptr,
    const unsigned *__restrict__ ent, const unsigned *__restrict__ rowlist, const unsigned *__restrict__ info,
    const float *__restrict__ inv_deg, const float *__restrict__ wtab, const float *__restrict__ root, int n_types,
    const float *__restrict__ bias, int64_t n, int act, float *__restrict__ out, double *__restrict__ bn_partial) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;          // (0 unless something puts static LDS in front)

    // ---- this block's contiguous run of tiles; blocks of one XCD (b % 8) take neighbouring runs (shared L2 lines)
    const int n_tiles = (int)((n + 15) >> 4);
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    if (nblk >= 8 && (nblk & 7) == 0) blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int t0 = (int)((int64_t)n_tiles * blk / nblk), t1 = (int)((int64_t)n_tiles * (blk + 1) / nblk);
#ifdef TGNN_ST_TIMING
    unsigned long long t_work = 0, t_wait = 0, t_steps = 0, t_last = __builtin_readcyclecounter();
    const unsigned long long t_begin = t_last;
    unsigned long long t_seg[4] = {0, 0, 0, 0}, t_mark = t_last;
#endif

    if (wave < kStMult) {
        // =================================================================== multiplying waves
        const int w = wave, r0 = w, r1 = w + kStMult, n_runs = n_types + 1;
        const int en = lane & 15, g = lane >> 4;             // entry of the block, kgroup / group of 4 output channels
        // weight fragments of this wave's runs as fp16 pairs (hi = RN16(w), lo = RN16(w - hi): 22+ significant bits), straight
        // from the fp32 tables: A operand of D^T = W^T . X^T -- lane (i, g), M block m: W[k = 8 g + j][16 m + i], j = 0..7
        f16x8 wa[4], wb[4];                                  // [2 piece (hi, lo) + m]
        {
            const float *ta = r0 < n_types ? wtab + (size_t)r0 * 1024 : root;
            const float *tb = r1 < n_types ? wtab + (size_t)r1 * 1024 : root;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f16x8 ah, al, bh, bl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int at = (8 * g + j) * 32 + 16 * m + en;
                    const float va = r0 < n_runs ? ta[at] : 0.f, vb = r1 < n_runs ? tb[at] : 0.f;
                    ah[j] = (_Float16)va; al[j] = (_Float16)(va - (float)ah[j]);
                    bh[j] = (_Float16)vb; bl[j] = (_Float16)(vb - (float)bh[j]);
                }
                wa[m] = ah; wa[2 + m] = al; wb[m] = bh; wb[2 + m] = bl;
            }
        }
        const unsigned sw16 = ((unsigned)g ^ (unsigned)(en & 6)) << 4;       // kgroup g's hi piece of the entry at position en
        const unsigned rd_c = (unsigned)en * 128u + sw16 + lds0;             // ... its lo piece sits 64 bytes from it
        const unsigned map_c = 2u * (unsigned)en + lds0, y_base = kStOffY + lds0;
        // 16 entries of a run -> 16 x 32 messages: pos = ring slot of the first, ys = Y slot of the first
        struct Blk { f16x8 xh, xl; unsigned ymv; };
        auto blk_read = [&](unsigned pos, unsigned ym) -> Blk {
            Blk r;
            // where this entry's message goes (staged by the epilogue waves): (slot of the message ring) << 3 | destination row & 7
            r.ymv = st_lds_ld<unsigned short>(ym + map_c);
            unsigned a;
            if (pos + 16 <= (unsigned)kStRingSlots) {
                a = (pos << 7) + rd_c;
            } else {                                         // (the block wraps at the ring's end)
                unsigned p = pos + en;
                p = p >= (unsigned)kStRingSlots ? p - kStRingSlots : p;
                a = (p << 7) + sw16 + lds0;
            }
            r.xh = st_lds_ld<f16x8>(a);
            r.xl = st_lds_ld<f16x8>(a ^ 64u);
            return r;
        };
        auto blk_mul = [&](const Blk &x, unsigned left, const f16x8 (&wf)[4]) {
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[2], x.xh, d0, 0, 0, 0);   // lo . hi
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[3], x.xh, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], x.xl, d0, 0, 0, 0);   // hi . lo
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], x.xl, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], x.xh, d0, 0, 0, 0);   // hi . hi
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], x.xh, d1, 0, 0, 0);
            // lane (en, g): channels 4 g .. 4 g + 3 (d0) and 16 + the same (d1) of entry en.  A message row's eight 16-byte
            // pieces are permuted by its destination row (piece c at c ^ (row & 7)): the 16 entries of a block go to 16
            // scattered rows of the message ring at the SAME piece -- unpermuted, one bank group.  The columns past the run's
            // last entry were computed from whatever the ring held there: not stored.
            if ((unsigned)en < left) {
                const unsigned ya = ((x.ymv ^ (unsigned)g) << 4) + y_base;
                st_lds_st<f32x4>(ya, d0);
                st_lds_st<f32x4>(ya ^ 64u, d1);
            }
        };
        auto ring_next = [&](unsigned pos) -> unsigned {
            pos += 16;
            return pos >= (unsigned)kStRingSlots ? pos - kStRingSlots : pos;
        };
        // the two runs of a tile (info word of a run: ring slot of its first entry | entries << 10 | first entry inside the tile
        // << 19): the first two blocks of both are read before the first product (a type has ~12 edges per tile: one block,
        // sometimes two), further blocks one after the other
        auto tile_runs = [&](unsigned qa_, unsigned qb_, unsigned map0) {
            const unsigned la = (qa_ >> 10) & 0x1ffu, lb = (qb_ >> 10) & 0x1ffu;
            const unsigned pa = qa_ & 0x3ffu, pb = qb_ & 0x3ffu, ma = map0 + 2 * (qa_ >> 19), mb = map0 + 2 * (qb_ >> 19);
            Blk a0, a1, b0, b1;
            if (la) a0 = blk_read(pa, ma);
            if (la > 16) a1 = blk_read(ring_next(pa), ma + 32);
            if (lb) b0 = blk_read(pb, mb);
            if (lb > 16) b1 = blk_read(ring_next(pb), mb + 32);
            if (la) blk_mul(a0, la, wa);
            if (la > 16) blk_mul(a1, la - 16, wa);
            if (lb) blk_mul(b0, lb, wb);
            if (lb > 16) blk_mul(b1, lb - 16, wb);
            if (la > 32) {
                unsigned pos = ring_next(ring_next(pa)), ym = ma + 64;
                for (unsigned b = 32; b < la; b += 16) { blk_mul(blk_read(pos, ym), la - b, wa); pos = ring_next(pos); ym += 32; }
            }
            if (lb > 32) {
                unsigned pos = ring_next(ring_next(pb)), ym = mb + 64;
                for (unsigned b = 32; b < lb; b += 16) { blk_mul(blk_read(pos, ym), lb - b, wb); pos = ring_next(pos); ym += 32; }
            }
        };
        // this wave's info words of a tile: run A, run B, the tile's first entry (absolute); brought into LDS by the epilogue
        // waves two tiles ahead (a scalar load here would sit on lgkmcnt, which every LDS wait of the wave then has to drain)
        const unsigned char *info_l = lds + kStOffInfo + w * 16;
        unsigned qa = 0, qb = 0, qt = 0;
        auto info_read = [&](int tile) {
            const uint4 v = *reinterpret_cast<const uint4 *>(info_l + ((tile - t0) & 7) * 256);
            qa = (unsigned)__builtin_amdgcn_readfirstlane(v.x);
            qb = (unsigned)__builtin_amdgcn_readfirstlane(v.y);
            qt = (unsigned)__builtin_amdgcn_readfirstlane(v.z);
        };
        ST_BARRIER();
        if (t0 < t1) info_read(t0);
        for (int s = t0; s <= t1; ++s) {
            if (s < t1) {
                const unsigned ca = qa, cb = qb, map0 = kStOffList + (unsigned)((s - t0) % 3) * 1024;
                ST_MARK()
                if (s + 1 < t1) info_read(s + 1);
                ST_SEG(0)
                tile_runs(ca, cb, map0);
                ST_SEG(1)
            }
            ST_BARRIER();
        }
    } else if (wave < kStMult + kStDma) {
        // =================================================================== loader waves
        const int d = wave - kStMult;
        // chunks (8 entries = 1 KB of rows) are numbered over the whole stream: chunk c sits in ring chunk c % 80
        const int c_first = tile_ent_ptr[t0] >> 3, c_end = tile_ent_ptr[t1] >> 3;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(hs), 0, (int)h_bytes, 0x00020000);
        const unsigned lane_piece = (unsigned)(lane & 7) << 4;   // ^ the entry word's permutation bits = source piece
        // local group k <-> the 8 chunks from c_first + 8 (2 k + d) on (64 entries)
        const int n_groups = (c_end - c_first + 7) >> 3;
        const int k_count = n_groups > d ? (n_groups - d + 1) >> 1 : 0;
        // entry words of a group: by LDS-DMA too, into this wave's 4-slot ring -- no register is the destination of a load
        // hipcc does not know of (it copies loop-carried registers at will: a copy taken before the data has landed is
        // stale), and every VMEM instruction of the wave is of one kind, so vmcnt counts them in order
        unsigned char *idx_ring = lds + kStOffIdx + d * 4 * 256;
        auto idx_load = [&](int k) {
            int gr = 2 * k + d;
            gr = gr < n_groups ? gr : (n_groups > 0 ? n_groups - 1 : 0);
            const unsigned *p = ent + ((int64_t)c_first + (int64_t)gr * 8) * 8 + lane;
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
            if (c0 + 8 <= c_end && ring + 8 <= kStRingChunks) {
                // (the common case: all 8 chunks inside the block's range and before the ring's end)
                unsigned char *dst = lds + ring * 1024;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(dst + i * 1024), 16, sr[i] ^ lane_piece, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int dstc = c0 + i < c_end ? ring : kStRingChunks;   // past the range: the dump chunk
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(lds + dstc * 1024), 16, sr[i] ^ lane_piece, 0, 0, 0);
                    ring = ring + 1 == kStRingChunks ? 0 : ring + 1;
                }
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
        // the tiles' first chunks: a window of 64 of them in a register (lane l = tile win + l), read with v_readlane
        int win = t0;
        int tp = tile_ent_ptr[(win + lane < n_tiles ? win + lane : n_tiles)] >> 3;
        ensure(c_first, __builtin_amdgcn_readlane(tp, t0 < t1 ? 1 : 0));
        ST_BARRIER();
        for (int s = t0; s <= t1; ++s) {
            if (s + 1 < t1) {
                if (s + 2 - win > 63) {
                    win = s;
                    tp = tile_ent_ptr[(win + lane < n_tiles ? win + lane : n_tiles)] >> 3;
                }
                ensure(__builtin_amdgcn_readlane(tp, s - win), __builtin_amdgcn_readlane(tp, s + 2 - win));
            }
            ST_BARRIER();
        }
        st_wait_vmcnt<0>();
    } else {
        // =================================================================== epilogue waves
        const int e = wave - kStMult - kStDma;               // rows 8 e .. 8 e + 7 of a tile; lane: row 8 e + (lane >> 3),
        const int r = 8 * e + (lane >> 3), cg = lane & 7;    // channels 4 cg .. 4 cg + 3
        const float4 bias4 = *reinterpret_cast<const float4 *>(bias + 4 * cg);
        double bs[4] = {0, 0, 0, 0}, bq[4] = {0, 0, 0, 0};
        const float unscale = hs_scale[1];                   // the rows arrive multiplied by a power of two (split16_kernel)
        const unsigned y_c = kStOffY + (((unsigned)cg ^ ((unsigned)r & 7u)) << 4) + lds0;   // piece cg of this row's messages
        if (e == 0 && lane < 32) reinterpret_cast<float *>(lds + kStOffY + kStYSlots * 128)[lane] = 0.f;   // the row of zeros
        // What the other waves need in LDS, by LDS-DMA (no register, no wait in front of this wave's own LDS reads): the info
        // block of a tile (wave e: the tiles of its parity; read by the multiplying waves from the step before the tile's, and
        // here -- row starts, 1 / deg -- in the step after it) two steps ahead, the tile's map entry -> message slot (wave 0; read
        // by the multiplying waves in the tile's step) one step ahead.
        int win = t0;
        int tp = tile_ent_ptr[(win + lane < n_tiles ? win + lane : n_tiles)];   // lane l = first entry of tile win + l
        auto map_dma = [&](int tile) {                       // (a tile past the block's range: the same bytes to the dump chunk)
            const int tc = tile < t1 ? tile : (t1 > t0 ? t1 - 1 : t0);
            if (tc - win > 63) {
                win = tc;
                tp = tile_ent_ptr[(win + lane < n_tiles ? win + lane : n_tiles)];
            }
            const unsigned char *src = reinterpret_cast<const unsigned char *>(rowlist) +
                                       2 * (size_t)(unsigned)__builtin_amdgcn_readlane(tp, tc - win) + 16 * lane;
            st_dma16(src, tile < t1 ? (unsigned)(kStOffList + ((tile - t0) % 3) * 1024) : (unsigned)kStOffDump);
        };
        auto info_dma = [&](int tile) {
            const int tc = tile < t1 ? tile : (t1 > t0 ? t1 - 1 : t0);
            st_dma4(info + (int64_t)tc * kStInfoWords + lane, tile < t1 ? (unsigned)(kStOffInfo + ((tile - t0) & 7) * 256) : (unsigned)kStOffDump);
        };
        // wave 0 brings the info blocks (tile s + 3 in step s), wave 1 the maps (tile s + 2 in step s): ONE DMA and one row store
        // per wave and step, so that "the DMA of the step before has landed" is the constant wait vmcnt(3)
        if (e == 0) { info_dma(t0); info_dma(t0 + 1); info_dma(t0 + 2); }
        else { map_dma(t0); map_dma(t0 + 1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ST_BARRIER();
        for (int s = t0; s <= t1; ++s) {
            // the request first (it lands while the rows of this and the next step are summed), then the rows of tile s - 1
            ST_MARK()
            if (e == 0) info_dma(s + 3); else map_dma(s + 2);
            ST_SEG(0)
            if (s > t0) {
                const int tile = s - 1;
                const int64_t v = (int64_t)tile * 16 + r;
                // the tile's info block: word 2 = its first absolute entry, words 32 .. 40 = 17 u16 row starts, words 48 .. 63 =
                // 1 / max(deg, 1) of its rows.  A row's messages sit in consecutive slots of the message ring: its edges in
                // CSR order, then its own.
                const unsigned char *ib = lds + kStOffInfo + ((tile - t0) & 7) * 256;
                const unsigned short *rsv = reinterpret_cast<const unsigned short *>(ib + 128);
                const unsigned st = rsv[r], cnt = rsv[r + 1] - st;
                const unsigned y0 = *reinterpret_cast<const unsigned *>(ib + 8) + st;
                const float invd = *reinterpret_cast<const float *>(ib + 192 + 4 * r) * unscale;
                const unsigned ne = cnt > 0 ? cnt - 1 : 0;   // edges
                const unsigned zrow = y_c + (unsigned)kStYSlots * 128u;
                auto slot_addr = [&](unsigned k, unsigned lim) -> unsigned {
                    return k < lim ? y_c + (((y0 + k) & (unsigned)(kStYSlots - 1)) << 7) : zrow;
                };
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ST_SEG(1)
                const f32x4 rt = st_lds_ld<f32x4>(slot_addr(ne, cnt));
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (unsigned k0 = 0; k0 < ne; k0 += 12) {   // twelve reads in flight; rows of the wave with fewer edges add zeros
                    f32x4 m[12];
#pragma unroll
                    for (int q = 0; q < 12; ++q) m[q] = st_lds_ld<f32x4>(slot_addr(k0 + q, ne));
#pragma unroll
                    for (int q = 0; q < 12; ++q) acc += m[q];
                }
                ST_SEG(2)
                float4 o;
                o.x = fmaf(acc[0], invd, fmaf(rt[0], unscale, bias4.x));
                o.y = fmaf(acc[1], invd, fmaf(rt[1], unscale, bias4.y));
                o.z = fmaf(acc[2], invd, fmaf(rt[2], unscale, bias4.z));
                o.w = fmaf(acc[3], invd, fmaf(rt[3], unscale, bias4.w));
                if (act == TGNN_ACT_LEAKY_RELU) { o.x = leakyf_(o.x); o.y = leakyf_(o.y); o.z = leakyf_(o.z); o.w = leakyf_(o.w); }
                if (v < n) {
                    bs[0] += (double)o.x; bq[0] += (double)o.x * (double)o.x;
                    bs[1] += (double)o.y; bq[1] += (double)o.y * (double)o.y;
                    bs[2] += (double)o.z; bq[2] += (double)o.z * (double)o.z;
                    bs[3] += (double)o.w; bq[3] += (double)o.w * (double)o.w;
                    *reinterpret_cast<float4 *>(out + v * 32 + 4 * cg) = o;
                }
            }
            // the DMA of the step BEFORE has landed before anybody reads it (behind it: that step's row store, this step's DMA
            // and row store; the first step has no row store: drained)
            if (s > t0 + 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ST_SEG(3)
            ST_BARRIER();
        }
        if (bn_partial) {
            // lanes (row, cg): fold the 8 rows of the wave (xor butterfly over lane bits 3..5: fixed order); one partial row
            // per epilogue wave
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int dl = 8; dl < 64; dl <<= 1) {
                    bs[q] += __shfl_xor(bs[q], dl, 64);
                    bq[q] += __shfl_xor(bq[q], dl, 64);
                }
            }
            if (lane < 8) {
                double *row = bn_partial + ((int64_t)blockIdx.x * 2 + e) * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    row[4 * cg + q] = bs[q];
                    row[32 + 4 * cg + q] = bq[q];
                }
            }
        }
    }
#ifdef TGNN_ST_TIMING
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *o = g_st_time + ((size_t)blockIdx.x * 12 + wave) * 4;
        o[0] = t_work; o[1] = t_wait; o[2] = t_steps; o[3] = __builtin_readcyclecounter() - t_begin;
        unsigned long long *gs = g_st_seg + ((size_t)blockIdx.x * 12 + wave) * 4;
        gs[0] = t_seg[0]; gs[1] = t_seg[1]; gs[2] = t_seg[2]; gs[3] = t_seg[3];
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The stream structure, from the adjacency CSR + the edge types in CSR order (once per layout):
//   tile_ent_ptr [tiles + 1]   entries before every tile (multiples of 8: a tile starts a 1 KB chunk)
//   ent          [..]          entry words: source row << 7 | X << 4, X = (position inside its type run) & 6 (header); per tile:
//                              runs 0..T (run T = the tile's own rows), inside a run the destination rows ascending, a row's
//                              edges of that type in CSR order; padding = kStPadWord
//   rowlist      [..] u16      parallel to ent: the slot of the message ring the entry's message goes to = (the tile's first
//                              absolute entry + the entry's position in ROW-major order: rows ascending, a row's edges in CSR
//                              order, then its own entry) & 511 -- a row's messages are consecutive slots --, << 3 | row & 7
//   info         [tiles][64]   words 4 w .. 4 w + 2 (multiplying wave w, runs A = w, B = w + 8): A, B = ring slot of the run's
//                              first entry ((absolute entry) % 640) | entries << 10 | first entry inside the tile << 19, then
//                              the tile's first absolute entry; words 32 .. 40: 17 u16 = first row-major position of every row
//                              (row 16 = the tile's entry count); words 48 .. 63: 1 / max(in-degree, 1) of the 16 rows
//   inv_deg      [16 tiles]    1 / max(in-degree, 1)
// result: [0] most entries of two consecutive tiles (padded), [1] most entries of one type run, [2] 1 = built
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
                                                         unsigned short *__restrict__ rowlist, unsigned *__restrict__ info,
                                                         float *__restrict__ inv_deg, int *__restrict__ result,
                                                         const int *__restrict__ gate) {
    if (gate && !*gate) return;
    const int n_types = n_types_dev ? *n_types_dev : n_types_host;
    if (n_types > kStMaxTypes) return;                       // (result[2] stays 0)
    __shared__ unsigned short cnt[64][17];                   // edges of row x type (odd stride); later: the write cursor
    __shared__ unsigned short pos[64][17];                   // first entry of (row, type) inside the tile
    __shared__ unsigned run_base[4][16], run_len[4][16];
    __shared__ unsigned short rstart[4][18];
    const int tid = threadIdx.x, k4 = tid >> 4, i = tid & 15;
    const int tile = blockIdx.x * 4 + k4;
    const int64_t row = (int64_t)tile * 16 + i;
    const int n_runs = n_types + 1;
    for (int t = 0; t < 16; ++t) cnt[tid][t] = 0;
    int e0 = 0, e1 = 0;
    if (row < n) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
    for (int e = e0; e < e1; ++e) {
        const int t = col_type[e];
        if ((unsigned)t < (unsigned)n_types) ++cnt[tid][t];
    }
    cnt[tid][n_types] = row < n ? 1 : 0;                     // root run: the row itself
    unsigned running = 0, maxrun = 0;
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
        if (i == 0) {
            run_base[k4][t] = running;
            run_len[k4][t] = total;
        }
        maxrun = total > maxrun ? total : maxrun;
        running += total;
    }
    // list position of every row: its edges + itself, rows ascending
    {
        const unsigned mine = row < n ? (unsigned)(e1 - e0) + 1u : 0u;
        unsigned incl = mine;
#pragma unroll
        for (int dl = 1; dl < 16; dl <<= 1) {
            const unsigned up = __shfl_up(incl, dl, 16);
            if (i >= dl) incl += up;
        }
        rstart[k4][i] = (unsigned short)(incl - mine);
        if (i == 15) { rstart[k4][16] = (unsigned short)incl; rstart[k4][17] = 0; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tile < n_tiles) {
        const int t_e0 = tile_ent_ptr[tile], t_e1 = tile_ent_ptr[tile + 1];
        for (int t = 0; t < 16; ++t) cnt[tid][t] = 0;
        for (int e = e0; e < e1; ++e) {
            const int t = col_type[e];
            if ((unsigned)t < (unsigned)n_types) {
                const unsigned c = cnt[tid][t]++;
                const unsigned p = pos[tid][t] + c;          // entry inside the tile
                ent[t_e0 + p] = ((unsigned)col_src[e] << 7) | (((p - run_base[k4][t]) & 6u) << 4);
                rowlist[t_e0 + p] = (unsigned short)((((unsigned)(t_e0 + rstart[k4][i] + (e - e0)) & (unsigned)(kStYSlots - 1)) << 3) | (unsigned)(i & 7));
            }
        }
        if (row < n) {
            const unsigned p = pos[tid][n_types];
            ent[t_e0 + p] = ((unsigned)row << 7) | (((p - run_base[k4][n_types]) & 6u) << 4);
            rowlist[t_e0 + p] = (unsigned short)((((unsigned)(t_e0 + rstart[k4][i] + (e1 - e0)) & (unsigned)(kStYSlots - 1)) << 3) | (unsigned)(i & 7));
        }
        for (int p = (int)running + i; p < t_e1 - t_e0; p += 16) {
            ent[t_e0 + p] = kStPadWord;
            rowlist[t_e0 + p] = 0;
        }
        unsigned *iw = info + (int64_t)tile * kStInfoWords;
        if (i < kStMult) {
            const int ra = i, rb = i + kStMult;
            unsigned qa = 0, qb = 0;
            if (ra < n_runs && run_len[k4][ra] > 0)
                qa = (((unsigned)t_e0 + run_base[k4][ra]) % (unsigned)kStRingSlots) | (run_len[k4][ra] << 10) | (run_base[k4][ra] << 19);
            if (rb < n_runs && run_len[k4][rb] > 0)
                qb = (((unsigned)t_e0 + run_base[k4][rb]) % (unsigned)kStRingSlots) | (run_len[k4][rb] << 10) | (run_base[k4][rb] << 19);
            iw[4 * i + 0] = qa; iw[4 * i + 1] = qb; iw[4 * i + 2] = (unsigned)t_e0; iw[4 * i + 3] = 0;
        }
        if (i < 9) iw[32 + i] = (unsigned)rstart[k4][2 * i] | ((unsigned)rstart[k4][2 * i + 1] << 16);
        else iw[32 + i] = 0;                                 // (words 41 .. 47)
        float r = 1.0f;
        if (row < n) r = 1.0f / (float)(e1 - e0 > 0 ? e1 - e0 : 1);
        inv_deg[row] = r;
        iw[48 + i] = __float_as_uint(r);
        if (i == 0) {
            const int t2 = tile + 2 <= n_tiles ? tile + 2 : n_tiles;
            atomicMax(&result[0], tile_ent_ptr[t2] - t_e0);
            atomicMax(&result[1], (int)maxrun);
            if (tile == 0) result[2] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Rows as the matrix pipe reads them: every fp32 row h[r][0..32) becomes 128 bytes of fp16 pairs, hi = RN16(s h),
// lo = RN16(s h - hi) (22+ significant bits; s = a power of two that brings the largest |h| just below 2^15: exact, and
// undone in the NNConv epilogue), laid out [hi ch 0-7][hi 8-15][hi 16-23][hi 24-31][lo 0-7]..[lo 24-31] -- piece g / 4 + g
// is kgroup g's B-operand fragment.  absmax: the largest |h| as float bits (non-negative floats order like integers).
// ------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const float *__restrict__ h, int64_t n4, unsigned *__restrict__ out_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(h)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    absmax_flush(m, out_bits);
}
__device__ __forceinline__ float st_scale_for(unsigned max_bits) {
    const int e = (int)(max_bits >> 23) & 0xff;              // biased exponent of the largest |h|
    if (e == 0 || e == 255) return 1.0f;                     // zero / subnormal / inf / nan: leave as it is
    int k = 14 - (e - 127);                                   // max * 2^k in [2^14, 2^15)
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return __uint_as_float((unsigned)(k + 127) << 23);
}
// one thread = 8 channels (kgroup g) of one row
__global__ void split16_kernel(const float *__restrict__ h, int64_t n_rows, const unsigned *__restrict__ max_bits,
                               void *__restrict__ hs, float *__restrict__ scale_out) {
    const float s = st_scale_for(*max_bits);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = s; scale_out[1] = 1.0f / s; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * 4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 2;
        const int g = (int)(i & 3);
        const float4 v0 = reinterpret_cast<const float4 *>(h)[r * 8 + 2 * g], v1 = reinterpret_cast<const float4 *>(h)[r * 8 + 2 * g + 1];
        const float x[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (_Float16)x[j];
            lo[j] = (_Float16)(x[j] - (float)hi[j]);
        }
        f16x8 *dst = reinterpret_cast<f16x8 *>(static_cast<unsigned char *>(hs) + r * 128);
        dst[g] = hi;
        dst[4 + g] = lo;
    }
}

void launch_absmax(const float *h, int64_t n_floats, unsigned *max_bits, hipStream_t s) {
    const int64_t n4 = n_floats / 4;
    if (n4 <= 0) return;
    int64_t g = (n4 + 255) / 256;
    if (g > 512) g = 512;
    absmax_kernel<<<(unsigned)g, 256, 0, s>>>(h, n4, max_bits);
}

int launch_nnconv_split16(const float *h, int64_t n_rows, void *hs, float *scale2, unsigned *max_bits, hipStream_t s) {
    TGNN_CHECK_HIP(hipMemsetAsync(max_bits, 0, 4, s));
    const int64_t n4 = n_rows * 8;
    int64_t g = (n4 + 255) / 256;
    if (g > 1024) g = 1024;
    absmax_kernel<<<(unsigned)g, 256, 0, s>>>(h, n4, max_bits);
    int64_t g2 = (n_rows * 4 + 255) / 256;
    if (g2 > 2048) g2 = 2048;
    split16_kernel<<<(unsigned)g2, 256, 0, s>>>(h, n_rows, max_bits, hs, scale2);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

int launch_nnconv_stream(const void *hs, const float *hs_scale, int64_t n_src_rows, const int32_t *tile_ent_ptr,
                         const uint32_t *ent_src, const uint32_t *rowlist, const uint32_t *info, const float *inv_deg,
                         const float *wtab, const float *root, int32_t n_types, const float *bias, int64_t n_nodes, int32_t act,
                         float *out, double *bn_partial, int32_t *n_partials_host, int reserve_cus, hipStream_t s) {
    auto kern = nnconv32_stream_kernel;
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(kern, kStLdsBytes, site));
    const int64_t n_tiles = (n_nodes + 15) / 16;
    int64_t blocks = cus_minus(reserve_cus);
    if (blocks > TGNN_BN_MAX_PARTIALS / 2) blocks = TGNN_BN_MAX_PARTIALS / 2;   // two partial rows per block
    if (blocks > n_tiles) blocks = n_tiles;
    if (blocks >= 8) blocks &= ~7;
    if (blocks < 1) blocks = 1;
    kern<<<(unsigned)blocks, kStWaves * 64, kStLdsBytes, s>>>(hs, (unsigned)(n_src_rows * 128), hs_scale, tile_ent_ptr, ent_src,
                                                              rowlist, info, inv_deg, wtab, root, n_types, bias, n_nodes, act, out,
                                                              bn_partial);
    if (n_partials_host) *n_partials_host = (int32_t)(2 * blocks);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// exclusive scan of graph_prep.hip
void exclusive_scan_i32_shared(const int *in, int *out, int64_t n, int *ws, hipStream_t s);
size_t scan_ws_ints_shared(int64_t n);

int nnconv_stream_build_gated(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type, int64_t n_nodes,
                              int32_t n_types, const int32_t *n_types_dev, const int32_t *gate, int32_t *tile_ent_ptr,
                              uint32_t *ent_src, uint32_t *rowlist, uint32_t *info, float *inv_deg, int32_t *result, void *ws,
                              hipStream_t s) {
    const int64_t n_tiles = (n_nodes + 15) / 16;
    TGNN_CHECK_HIP(hipMemsetAsync(result, 0, 4 * sizeof(int32_t), s));
    stream_count_kernel<<<(unsigned)((n_tiles + 1 + 255) / 256), 256, 0, s>>>(rowptr, n_nodes, (int)n_tiles, tile_ent_ptr, gate);
    exclusive_scan_i32_shared(tile_ent_ptr, tile_ent_ptr, n_tiles + 1, static_cast<int *>(ws), s);
    stream_fill_kernel<<<(unsigned)((n_tiles + 3) / 4), 64, 0, s>>>(rowptr, col_src, col_type, n_nodes, (int)n_tiles, n_types,
                                                                    n_types_dev, tile_ent_ptr, ent_src,
                                                                    reinterpret_cast<unsigned short *>(rowlist), info, inv_deg,
                                                                    result, gate);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

}  // namespace tgnn

using namespace tgnn;

#ifdef TGNN_ST_TIMING
extern "C" int tgnn_debug_stream_time(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tgnn::g_st_time), sizeof(unsigned long long) * 512 * 12 * 4);
}
extern "C" int tgnn_debug_stream_seg(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tgnn::g_st_seg), sizeof(unsigned long long) * 512 * 12 * 4);
}
#endif

extern "C" void tgnn_nnconv_stream_limits(int32_t *max_types, int32_t *max_pair_entries, int32_t *max_run_entries) {
    if (max_types) *max_types = kStMaxTypes;
    if (max_pair_entries) *max_pair_entries = kStMaxPair;
    if (max_run_entries) *max_run_entries = 511;
}

extern "C" int64_t tgnn_nnconv_stream_max_entries(int64_t n_nodes, int64_t n_edges) {
    const int64_t n_tiles = (n_nodes + 15) / 16;
    return n_edges + n_nodes + 7 * n_tiles + 1024;           // padding to 8 per tile + the loaders' / the epilogue's read-ahead
}

extern "C" size_t tgnn_nnconv_stream_scan_ws_bytes(int64_t n_nodes) {
    return scan_ws_ints_shared((n_nodes + 15) / 16 + 1) * 4 + 256;
}

extern "C" int tgnn_nnconv_stream_build(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type,
                                        int64_t n_nodes, int32_t n_types, const int32_t *n_types_dev, int32_t *tile_ent_ptr,
                                        uint32_t *ent_src, uint32_t *rowlist, uint32_t *info, float *inv_deg, int32_t *result,
                                        void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(rowptr && tile_ent_ptr && ent_src && rowlist && info && inv_deg && result, "null pointer");
    TGNN_CHECK_ARG(n_nodes >= 1 && n_nodes < (int64_t(1) << 24), "node count");
    TGNN_CHECK_ARG(ws && ws_bytes >= tgnn_nnconv_stream_scan_ws_bytes(n_nodes), "workspace");
    return nnconv_stream_build_gated(rowptr, col_src, col_type, n_nodes, n_types, n_types_dev, nullptr, tile_ent_ptr, ent_src,
                                     rowlist, info, inv_deg, result, ws, static_cast<hipStream_t>(stream));
}

extern "C" size_t tgnn_nnconv_stream_split_bytes(int64_t n_src_rows) { return (size_t)n_src_rows * 128 + 256; }

extern "C" int tgnn_nnconv_mean_stream_fwd(const float *h, int64_t n_src_rows, const int32_t *tile_ent_ptr,
                                           const uint32_t *ent_src, const uint32_t *rowlist, const uint32_t *info,
                                           const float *inv_deg, const float *wtab, int32_t n_types, const float *root,
                                           const float *bias, int64_t n_nodes, int32_t c, int32_t act, float *out,
                                           void *split_scratch, double *bn_partial, int32_t *n_partials_host,
                                           tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && c == 32, "the stream NNConv kernel is built for network_width 32");
    TGNN_CHECK_ARG(act == TGNN_ACT_NONE || act == TGNN_ACT_LEAKY_RELU, "activation");
    TGNN_CHECK_ARG(h && tile_ent_ptr && ent_src && rowlist && info && inv_deg && root && bias && out && split_scratch,
                   "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || wtab, "null weight table");
    TGNN_CHECK_ARG(n_types <= kStMaxTypes, "too many edge types for the stream kernel");
    TGNN_CHECK_ARG(n_src_rows >= n_nodes && n_src_rows < (int64_t(1) << 24), "source rows (dense [rows][32] within 2 GB)");
    TGNN_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                       ((uintptr_t)split_scratch % 256) == 0, "alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // split_scratch: [n_src_rows][128 B] fp16 pairs, then 2 floats (scale, 1 / scale) and the max word
    unsigned char *sc = static_cast<unsigned char *>(split_scratch);
    float *scale2 = reinterpret_cast<float *>(sc + (size_t)n_src_rows * 128);
    unsigned *max_bits = reinterpret_cast<unsigned *>(scale2 + 2);
    const int rc = launch_nnconv_split16(h, n_src_rows, sc, scale2, max_bits, s);
    if (rc != TGNN_OK) return rc;
    return launch_nnconv_stream(sc, scale2, n_src_rows, tile_ent_ptr, ent_src, rowlist, info, inv_deg, wtab, root, n_types, bias,
                                n_nodes, act, out, bn_partial, n_partials_host, 0, s);
}
