// Graph preparation for the TilinGNN forward on gfx950: COO(int64) -> CSR-by-destination(int32),
// exact edge-attribute de-duplication, small index utilities.
//
// The reference hands the network an unsorted int64 edge_index [2,E] per layout
// (/root/reference/util/data_util.py:110-117; pairs (u,v),(v,u) consecutive,
// tiling/tile_graph.py:206-207) and PyG's MessagePassing.propagate scatters messages to
// edge_index[1].  Here the scatter becomes a gather: rows of a destination-sorted CSR, built once
// per layout and reused by all 20 layers of both branches.
#include <atomic>
#include <chrono>
#include <mutex>

#include "tgnn_common.h"

namespace tgnn {

// ------------------------------------------------------------------------------------------
// exclusive scan of int32 (three-phase, recursive on the block sums)
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;                         // per thread
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048 per block

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// out[i] = exclusive prefix inside the tile; block_sums[b] = tile total
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(const int *in, int *out,  // may alias
                                                                  int *__restrict__ block_sums, int64_t n) {
    __shared__ int wave_tot[kScanThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems];
    int tsum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        tsum += v[k];
    }
    const int incl = wave_inclusive_scan(tsum);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_tot[w];
    int run = woff + incl - tsum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kScanThreads - 1) block_sums[blockIdx.x] = run;
}

__global__ __launch_bounds__(kScanThreads) void scan_add_offsets_kernel(int *__restrict__ out,
                                                                        const int *__restrict__ block_offsets,
                                                                        int64_t n) {
    const int off = block_offsets[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) out[base + k] += off;
}

// [r6] scan_tiles_kernel whose LAST block to finish (a ticket) also scans the tile totals: the three launches of a long scan
// become one when the readers add their tile's offset themselves (bk_scanned below) -- out[i] stays the prefix inside its tile,
// tile_off[i / kScanTile] the prefix of the tiles before.  At most kScanFusedTiles tiles (2 M entries); *ticket == 0 on entry
// and on exit.
constexpr int kScanFusedTiles = 1024;
__global__ __launch_bounds__(kScanThreads) void scan_tiles_fused_kernel(const int *in, int *out, int *__restrict__ block_sums,
                                                                        int *__restrict__ tile_off, int64_t n, unsigned *ticket) {
    __shared__ int wave_tot[kScanThreads / 64];
    __shared__ unsigned my_ticket;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems];
    int tsum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        tsum += v[k];
    }
    const int incl = wave_inclusive_scan(tsum);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_tot[w];
    int run = woff + incl - tsum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kScanThreads - 1) {
        __hip_atomic_store(block_sums + blockIdx.x, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by another block: past L1 / the XCD's L2)
        __threadfence();
        my_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (my_ticket != gridDim.x - 1) return;                  // (uniform)
    __threadfence();
    const int nb = (int)gridDim.x, per = (nb + kScanThreads - 1) / kScanThreads;     // <= 4 consecutive totals per thread
    int t[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x * per + k;
        t[k] = (k < per && i < nb) ? __hip_atomic_load(block_sums + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        sum += t[k];
    }
    __syncthreads();                                         // (wave_tot is reused)
    const int incl2 = wave_inclusive_scan(sum);
    if (lane == 63) wave_tot[wave] = incl2;
    __syncthreads();
    int run2 = incl2 - sum;
    for (int w = 0; w < wave; ++w) run2 += wave_tot[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x * per + k;
        if (k < per && i < nb) tile_off[i] = run2;
        run2 += t[k];
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t scan_ws_ints(int64_t n) {
    size_t total = 0;
    while (n > kScanTile) {
        int64_t nb = (n + kScanTile - 1) / kScanTile;
        total += align_up((size_t)nb, 64) * 2;  // sums + their scan
        n = nb;
    }
    return total + 128 + 64;                                 // (+ the fused scan's ticket word)
}

// short arrays (<= 32 768 entries): one block, one launch instead of three (each ~5 us of mostly launch latency)
constexpr int kScanOneThreads = 1024, kScanOneMax = kScanOneThreads * 32;
__global__ __launch_bounds__(kScanOneThreads) void scan_one_block_kernel(const int *in, int *out, int n) {   // may alias
    __shared__ int wave_tot[kScanOneThreads / 64];
    const int tid = threadIdx.x;
    const int per = (n + kScanOneThreads - 1) / kScanOneThreads;            // <= 32 consecutive entries per thread
    const int b = tid * per, e = min(b + per, n);
    int v[32];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = (k < per && b + k < e) ? in[b + k] : 0;
        sum += v[k];
    }
    const int incl = wave_inclusive_scan(sum);
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) run += wave_tot[w];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k < per && b + k < e) out[b + k] = run;
        run += v[k];
    }
}

// exclusive scan; `in` and `out` may alias.  ws holds scan_ws_ints(n) ints.
static void exclusive_scan_i32(const int *in, int *out, int64_t n, int *ws, hipStream_t s) {
    if (n <= 0) return;
    if (n <= kScanOneMax) {
        scan_one_block_kernel<<<1, kScanOneThreads, 0, s>>>(in, out, (int)n);
        return;
    }
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    int *sums = ws;
    int *sums_scan = ws + align_up((size_t)nb, 64);
    scan_tiles_kernel<<<(unsigned)nb, kScanThreads, 0, s>>>(in, out, sums, n);
    if (nb > 1) {
        exclusive_scan_i32(sums, sums_scan, nb, ws + 2 * align_up((size_t)nb, 64), s);
        scan_add_offsets_kernel<<<(unsigned)nb, kScanThreads, 0, s>>>(out, sums_scan, n);
    }
}

// (shared with nnconv_stream.hip)
void exclusive_scan_i32_shared(const int *in, int *out, int64_t n, int *ws, hipStream_t s) { exclusive_scan_i32(in, out, n, ws, s); }
size_t scan_ws_ints_shared(int64_t n) { return scan_ws_ints(n); }

// ------------------------------------------------------------------------------------------
// CSR by destination
// ------------------------------------------------------------------------------------------
// Pass 1: in-degree counts; the value the atomic returns is the edge's (arbitrary but unique) rank inside its row,
// kept so that the fill pass is a plain scatter without a second round of returning atomics (device-scope atomics
// are served by the memory side of the fabric on this chip, not by an XCD's L2: ~50 us per 1.25 M of them).
__global__ void csr_count_kernel(const int64_t *__restrict__ ei, int64_t e, int64_t n, int64_t n_src, int drop_self,
                                 int *__restrict__ cnt, int *__restrict__ rank, int *__restrict__ err_flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = ei[i], d = ei[e + i];
        int r = -1;
        if (s < 0 || s >= n_src || d < 0 || d >= n) {
            if (err_flag) *err_flag = 1;
        } else if (!(drop_self && s == d)) {
            r = atomicAdd(&cnt[d], 1);
        }
        rank[i] = r;
    }
}

__global__ void csr_fill_kernel(const int64_t *__restrict__ ei, int64_t e, const int *__restrict__ rowptr,
                                const int *__restrict__ rank, int *__restrict__ col_src, int *__restrict__ col_eid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = rank[i];
        if (r < 0) continue;
        const int pos = rowptr[ei[e + i]] + r;
        col_src[pos] = (int)ei[i];
        col_eid[pos] = (int)i;
    }
}

// Restores the original edge order inside every row (the ranks above are in arrival order of the atomics): the
// order the reference's scatter sees.  One block = 256 consecutive rows = one contiguous CSR segment, staged in
// LDS with coalesced loads, every thread insertion-sorts its own (short) row there, coalesced write-back.
// A segment that does not fit (very high in-degrees) is sorted in place in global memory.
constexpr int kSortRows = 256, kSortCap = 6144;   // 48 KB of LDS
__global__ __launch_bounds__(kSortRows) void csr_sort_rows_kernel(const int *__restrict__ rowptr, int64_t n,
                                                                  int *__restrict__ col_src, int *__restrict__ col_eid) {
    __shared__ int key_s[kSortCap], val_s[kSortCap];
    const int64_t v0 = (int64_t)blockIdx.x * kSortRows;
    const int64_t v1 = v0 + kSortRows < n ? v0 + kSortRows : n;
    const int seg_b = rowptr[v0], seg_e = rowptr[v1], seg_n = seg_e - seg_b;
    const int64_t v = v0 + threadIdx.x;
    if (seg_n > kSortCap) {                                  // uniform per block
        if (v < v1) {
            const int b = rowptr[v], e = rowptr[v + 1];
            for (int i = b + 1; i < e; ++i) {
                const int key = col_eid[i], val = col_src[i];
                int j = i - 1;
                while (j >= b && col_eid[j] > key) {
                    col_eid[j + 1] = col_eid[j];
                    col_src[j + 1] = col_src[j];
                    --j;
                }
                col_eid[j + 1] = key;
                col_src[j + 1] = val;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < seg_n; i += kSortRows) {
        key_s[i] = col_eid[seg_b + i];
        val_s[i] = col_src[seg_b + i];
    }
    __syncthreads();
    if (v < v1) {
        const int b = rowptr[v] - seg_b, e = rowptr[v + 1] - seg_b;
        for (int i = b + 1; i < e; ++i) {
            const int key = key_s[i], val = val_s[i];
            int j = i - 1;
            while (j >= b && key_s[j] > key) {
                key_s[j + 1] = key_s[j];
                val_s[j + 1] = val_s[j];
                --j;
            }
            key_s[j + 1] = key;
            val_s[j + 1] = val;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < seg_n; i += kSortRows) {
        col_eid[seg_b + i] = key_s[i];
        col_src[seg_b + i] = val_s[i];
    }
}

// ------------------------------------------------------------------------------------------
// edge-attribute row de-duplication
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t canon_bits(float f) {
    uint32_t u = __float_as_uint(f);
    return u == 0x80000000u ? 0u : u;  // -0.0 == +0.0
}

__device__ __forceinline__ uint64_t mix64(uint64_t h, uint32_t v) {
    h ^= v;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h;
}

__device__ __forceinline__ bool rows_equal(const float *__restrict__ attr, int fe, int64_t a, int64_t b) {
    for (int k = 0; k < fe; ++k)
        if (canon_bits(attr[a * fe + k]) != canon_bits(attr[b * fe + k])) return false;
    return true;
}

// table[slot] = smallest edge number carrying the row content that hashed/probed to the slot.
// Two levels: a block first de-duplicates its own 256 rows in LDS (rows staged with coalesced loads, an LDS
// hash table keyed by content, representative = smallest thread index); only the block representatives --
// a dozen per block on real layouts -- then probe the global table.  A flat version in which all 1e6 edges
// probed the global table spent 220-280 us hammering 13 hot slots.
constexpr int kDedupRows = 256;
constexpr int kDedupListMax = 4096;                      // distinct rows the list-based numbering of tgnn_graph_prep takes
constexpr int kDedupLocal = 512;                       // LDS table slots (>= 2 x rows: never full)
__global__ __launch_bounds__(kDedupRows) void dedup_insert_kernel(const float *__restrict__ attr, int64_t e, int fe,
                                                                  int *__restrict__ table, uint32_t mask,
                                                                  int *__restrict__ slot_of_edge,
                                                                  int *__restrict__ rep_cnt = nullptr,
                                                                  int *__restrict__ rep_slots = nullptr) {
    // rep_cnt / rep_slots (tgnn_graph_prep): every slot is put on a list by the thread that claims it -- one entry per distinct
    // row content; *rep_cnt starts at -1 (it lies behind the table, in the same 0xFF fill)
    extern __shared__ uint32_t rows_s[];                 // [256][ld], ld = fe | 1 (odd: conflict-free per-thread rows)
    __shared__ int ltab[kDedupLocal];
    __shared__ int lslot[kDedupRows];                    // global slot found by each block representative
    const int ld = fe | 1, tid = threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * kDedupRows; base < e; base += (int64_t)gridDim.x * kDedupRows) {
        const int64_t n_here = e - base < kDedupRows ? e - base : kDedupRows;
        const int64_t total = n_here * fe;
        __syncthreads();
        {   // (row, column) of element k = tid + 256 j advance by (256 / fe, 256 % fe): no division per element
            const int step_r = kDedupRows / fe, step_c = kDedupRows % fe;
            int r = tid / fe, c = tid % fe;
            const float *src = attr + base * fe;
            for (int k = tid; k < (int)total; k += kDedupRows) {
                rows_s[r * ld + c] = canon_bits(src[k]);
                r += step_r;
                c += step_c;
                if (c >= fe) { c -= fe; ++r; }
            }
        }
        ltab[tid] = -1;
        ltab[tid + kDedupRows] = -1;
        __syncthreads();
        const int64_t i = base + tid;
        const bool ok = i < e;
        const uint32_t *mine = rows_s + tid * ld;
        uint64_t h = 0xCBF29CE484222325ull;
        if (ok)
            for (int k = 0; k < fe; ++k) h = mix64(h, mine[k]);
        const uint32_t h32 = (uint32_t)(h ^ (h >> 32));
        // ---- level 1: representative inside the block
        int my_l = 0;
        if (ok) {
            uint32_t sl = h32 & (kDedupLocal - 1);
            while (true) {
                int cur = ltab[sl];
                if (cur < 0) {
                    const int prev = atomicCAS(&ltab[sl], -1, tid);
                    if (prev < 0) break;
                    cur = prev;
                }
                bool same = cur == tid;
                if (!same) {                                         // (no early exit: the reads go out back to back)
                    const uint32_t *other = rows_s + cur * ld;
                    uint32_t diff = 0;
                    for (int k = 0; k < fe; ++k) diff |= other[k] ^ mine[k];
                    same = diff == 0;
                }
                if (same) {
                    if (tid < cur) atomicMin(&ltab[sl], tid);
                    break;
                }
                sl = (sl + 1) & (kDedupLocal - 1);
            }
            my_l = (int)sl;
        }
        __syncthreads();
        // ---- level 2: block representatives probe the global table
        if (ok && ltab[my_l] == tid) {
            uint32_t slot = h32 & mask;
            while (true) {
                int cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur < 0) {
                    const int prev = atomicCAS(&table[slot], -1, (int)i);
                    if (prev < 0) {       // claimed the slot
                        if (rep_cnt) {
                            const int k = atomicAdd(rep_cnt, 1) + 1;
                            if (k < kDedupListMax) rep_slots[k] = (int)slot;
                        }
                        break;
                    }
                    cur = prev;
                }
                bool same = cur == (int)i;
                if (!same) {
                    // a handful of hot rows: L1 / L2 hits -- all fe loads in flight at once (with an early exit they were a chain of
                    // fe round trips per probing thread, and this level is the latency of every block)
                    const float *other = attr + (int64_t)cur * fe;
                    uint32_t diff = 0;
#pragma unroll 8
                    for (int k = 0; k < fe; ++k) diff |= canon_bits(other[k]) ^ mine[k];
                    same = diff == 0;
                }
                if (same) {
                    if ((int)i < cur) atomicMin(&table[slot], (int)i);     // the representative only ever decreases
                    break;
                }
                slot = (slot + 1) & mask;  // table is at most half full: terminates
            }
            lslot[tid] = (int)slot;
        }
        __syncthreads();
        if (ok) slot_of_edge[i] = lslot[ltab[my_l]];
    }
}

// Wide rows (Fe > 60: the staged tile would not fit 64 KB of LDS): every thread reads its row from HBM.
__global__ void dedup_insert_direct_kernel(const float *__restrict__ attr, int64_t e, int fe, int *__restrict__ table,
                                           uint32_t mask, int *__restrict__ slot_of_edge, int *__restrict__ rep_cnt = nullptr,
                                           int *__restrict__ rep_slots = nullptr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = 0xCBF29CE484222325ull;
        for (int k = 0; k < fe; ++k) h = mix64(h, canon_bits(attr[i * fe + k]));
        uint32_t slot = (uint32_t)(h ^ (h >> 32)) & mask;
        while (true) {
            int cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                const int prev = atomicCAS(&table[slot], -1, (int)i);
                if (prev < 0) {
                    if (rep_cnt) {
                        const int k = atomicAdd(rep_cnt, 1) + 1;
                        if (k < kDedupListMax) rep_slots[k] = (int)slot;
                    }
                    break;
                }
                cur = prev;
            }
            if (cur == (int)i || rows_equal(attr, fe, cur, i)) {
                if ((int)i < cur) atomicMin(&table[slot], (int)i);
                break;
            }
            slot = (slot + 1) & mask;
        }
        slot_of_edge[i] = (int)slot;
    }
}

__global__ void dedup_mark_first_kernel(const int *__restrict__ table, const int *__restrict__ slot_of_edge,
                                        int64_t e, int *__restrict__ is_first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x)
        is_first[i] = table[slot_of_edge[i]] == (int)i ? 1 : 0;
}

// first_rank = exclusive scan of is_first
__global__ void dedup_assign_kernel(const int *__restrict__ table, const int *__restrict__ slot_of_edge,
                                    const int *__restrict__ first_rank, int64_t e, int *__restrict__ edge_type,
                                    int *__restrict__ type_rep_edge, int *__restrict__ n_types,
                                    const int *__restrict__ skip_flag = nullptr) {
    // n_types == NULL: the count was written by dedup_rank_kernel; *skip_flag set: too many distinct rows for that kernel's
    // list, first_rank holds nothing (the caller falls back)
    if (skip_flag && *skip_flag) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        const int rep = table[slot_of_edge[i]];
        const int t = first_rank[rep];
        edge_type[i] = t;
        if (rep == (int)i) type_rep_edge[t] = (int)i;
        if (n_types && i == e - 1) *n_types = first_rank[i] + (rep == (int)i ? 1 : 0);
    }
}

// The list-based numbering (tgnn_graph_prep): type id of a distinct row = rank of its first edge among the first edges of all
// distinct rows -- what the mark / scan / assign sequence computes with a scan over all E edges, here from the <= 4 096 list
// entries in one block.  More entries than the list holds: *fallback = 1 (nothing else is written).
__global__ __launch_bounds__(1024) void dedup_rank_kernel(const int *__restrict__ table, const int *__restrict__ rep_cnt,
                                                          const int *__restrict__ rep_slots, int *__restrict__ first_rank,
                                                          int *__restrict__ type_rep_edge, int *__restrict__ n_types,
                                                          int *__restrict__ fallback) {
    __shared__ int reps[kDedupListMax];
    const int cnt = *rep_cnt + 1;
    if (cnt > kDedupListMax) {
        if (threadIdx.x == 0) { *fallback = 1; *n_types = cnt; }
        return;
    }
    for (int k = threadIdx.x; k < cnt; k += 1024) reps[k] = table[rep_slots[k]];
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += 1024) {
        const int mine = reps[k];
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += reps[j] < mine ? 1 : 0;
        first_rank[mine] = rank;
        type_rep_edge[rank] = mine;
    }
    if (threadIdx.x == 0) *n_types = cnt;
}

__global__ void gather_i32_kernel(const int *__restrict__ src, int64_t n_src, const int *__restrict__ idx,
                                  int64_t n, int *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = idx[i];                       // CSR slots past rowptr[N] (skipped edges) hold garbage
        out[i] = (k >= 0 && k < n_src) ? src[k] : 0;
    }
}

// ------------------------------------------------------------------------------------------
// NNConv type columns (nnconv_cols.hip): per 16 destination rows a list of columns sorted by edge type;
// column (t, r) holds the source of every row's r-th in-edge of type t (CSR = original order) or -1.
// The last column of a tile is the root column (type T): max(deg,1) as float bits, -1 for rows >= n.
// col_meta = type | first-of-type << 8 | last-of-type << 9 | end-of-tile << 10.
// One block = 64 rows = 4 tiles, one thread per row.
// ------------------------------------------------------------------------------------------
constexpr int kColTileRows = 16;
constexpr int kMaxColTypes = 40;

template <bool FILL>
__global__ __launch_bounds__(64) void nnconv_col_kernel(const int *__restrict__ rowptr, const int *__restrict__ col_src,
                                                        const int *__restrict__ col_type, int64_t n, int n_types_host,
                                                        int *__restrict__ tile_cols,            // !FILL: out, columns per tile
                                                        const int *__restrict__ tile_col_ptr,   // FILL
                                                        int *__restrict__ col_meta, int *__restrict__ col_slot_src,
                                                        const int *__restrict__ n_types_dev = nullptr, int max_types = kMaxColTypes,
                                                        int *__restrict__ built_flag = nullptr,
                                                        const int *__restrict__ edge_type = nullptr,
                                                        const int *__restrict__ col_eid = nullptr,
                                                        int *__restrict__ col_type_out = nullptr) {
    // n_types_dev: the count is still on the device (tgnn_graph_prep queues the whole preparation without a host round trip);
    // more types than the structure / the matrix-core kernel take: nothing is built, *built_flag stays 0
    const int n_types = n_types_dev ? *n_types_dev : n_types_host;
    if (!FILL && col_type_out) {
        // tgnn_graph_prep: the types in CSR order (col_type_out = what col_type points to) are gathered here, every row by its
        // own thread, instead of by a launch of their own in front of this one
        const int64_t row0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
        if (row0 < n)
            for (int e = rowptr[row0]; e < rowptr[row0 + 1]; ++e) col_type_out[e] = edge_type[col_eid[e]];
    }
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) tile_cols[(n + kColTileRows - 1) / kColTileRows] = 0;   // (the scan's last entry)
    if (n_types > max_types || n_types > kMaxColTypes) {
        if (!FILL) {                                          // (the scan behind this launch reads every entry)
            const int64_t tile0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4);
            if ((threadIdx.x & 15) == 0 && tile0 < (n + kColTileRows - 1) / kColTileRows) tile_cols[tile0] = 0;
        }
        return;
    }
    if (built_flag && blockIdx.x == 0 && threadIdx.x == 0) *built_flag = 1;
    __shared__ int cnt[64][kMaxColTypes + 1];    // +1: odd stride, the per-row walks hit distinct banks
    __shared__ int maxm[4][kMaxColTypes];
    __shared__ int base[4][kMaxColTypes + 1];
    const int tid = threadIdx.x, k = tid >> 4, i = tid & 15;
    const int64_t row = (int64_t)blockIdx.x * 64 + tid;
    const int64_t tile = (int64_t)blockIdx.x * 4 + k;
    const int64_t n_tiles = (n + kColTileRows - 1) / kColTileRows;
    for (int t = 0; t < n_types; ++t) cnt[tid][t] = 0;
    int e0 = 0, e1 = 0;
    if (row < n) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
    if (!FILL && col_type_out) {
        for (int e = e0; e < e1; ++e) cnt[tid][col_type_out[e]]++;          // (this thread's own stores above)
    } else {
        for (int e = e0; e < e1; ++e) cnt[tid][col_type[e]]++;
    }
    __syncthreads();
    for (int t = i; t < n_types; t += 16) {
        int m = 0;
        for (int r = 0; r < 16; ++r) m = max(m, cnt[k * 16 + r][t]);
        maxm[k][t] = m;
    }
    __syncthreads();
    if (i == 0) {
        int acc = 0;
        for (int t = 0; t < n_types; ++t) { base[k][t] = acc; acc += maxm[k][t]; }
        base[k][n_types] = acc;
        if (!FILL && tile < n_tiles) tile_cols[tile] = acc + 1;
    }
    if (!FILL) return;
    __syncthreads();
    if (tile >= n_tiles) return;                 // whole 16-thread group: no barrier below
    const int64_t c0 = tile_col_ptr[tile];
    const int n_edge_cols = base[k][n_types];
    for (int c = 0; c < n_edge_cols; ++c) col_slot_src[(c0 + c) * 16 + i] = -1;
    for (int t = i; t < n_types; t += 16) {
        const int m = maxm[k][t];
        for (int r = 0; r < m; ++r)
            col_meta[c0 + base[k][t] + r] = t | (r == 0 ? 1 << 8 : 0) | (r == m - 1 ? 1 << 9 : 0);
    }
    // root column
    const int deg = e1 - e0;
    col_slot_src[(c0 + n_edge_cols) * 16 + i] = row < n ? __float_as_int((float)(deg > 0 ? deg : 1)) : -1;
    if (i == 0) col_meta[c0 + n_edge_cols] = n_types | (1 << 8) | (1 << 9) | (1 << 10);
    // the 16 threads of a tile are lanes of one wavefront: their -1 stores above are ordered before these
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < n_types; ++t) cnt[tid][t] = 0;
    for (int e = e0; e < e1; ++e) {
        const int t = col_type[e];
        const int r = cnt[tid][t]++;
        col_slot_src[(c0 + base[k][t] + r) * 16 + i] = col_src[e];
    }
}

// ------------------------------------------------------------------------------------------
// NNConv edge groups (nnconv_eg.hip): per 16 destination rows its in-edges sorted by (type, row, CSR order) and cut into
// GROUPS of up to 16 edges of ONE type -- a gather instruction of the kernel fetches 16 source rows whatever rows of the
// tile they go to (the type columns above fill 29 % of their slots at 10 edges over 13 types; groups 70 %).
//   grp [16 * n_groups] of (src, sm) pairs:
//     src of slot k: source row, -1 = none; in the root group (the last of a tile) the float bits of max(in-degree, 1) of
//                    row k, -1 for rows >= N
//     sm of word j:  (mask of the slots that go to row j) | (type | root << 8) << 16
// One block = 64 rows = 4 tiles, one thread per row.
// ------------------------------------------------------------------------------------------
constexpr int kEgRoot = 1 << 8;
constexpr int kPrepWordsMagic = 0x600D0001;                 // word 31 of the pinned result words: "the other 31 have landed"

template <bool FILL>
__global__ __launch_bounds__(64) void nnconv_eg_kernel(const int *__restrict__ rowptr, const int *__restrict__ col_src,
                                                       const int *__restrict__ col_type, int64_t n, int n_types_host,
                                                       int *__restrict__ tile_grps,            // !FILL: out, groups per tile
                                                       const int *__restrict__ tile_grp_ptr,   // FILL
                                                       int2 *__restrict__ grp,
                                                       const int *__restrict__ n_types_dev = nullptr, int max_types = kMaxColTypes,
                                                       int *__restrict__ built_flag = nullptr,
                                                       const int *__restrict__ edge_type = nullptr,
                                                       const int *__restrict__ col_eid = nullptr,
                                                       int *__restrict__ col_type_out = nullptr,
                                                       const int *__restrict__ result_words = nullptr,
                                                       int *__restrict__ host_words = nullptr) {
    // [r6] tgnn_graph_prep's result words straight into the caller's pinned host buffer by the first wave of the first launch behind
    // the join of its two chains -- no copy launch, no hand-over to another queue, the host polls word 31 (tgnn_graph_prep_wait)
    if (!FILL && host_words && blockIdx.x == 0 && threadIdx.x < 32) {
        const int v = result_words[threadIdx.x];
        if (threadIdx.x < 31) __hip_atomic_store(host_words + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();                              // (one wave: its 31 stores are acknowledged before word 31 goes out)
        if (threadIdx.x == 31) __hip_atomic_store(host_words + 31, kPrepWordsMagic, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int n_types = n_types_dev ? *n_types_dev : n_types_host;
    const int64_t n_tiles = (n + kColTileRows - 1) / kColTileRows;
    if (!FILL && col_type_out) {   // tgnn_graph_prep without the column structure: the types in CSR order are gathered here
        const int64_t row0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
        if (row0 < n)
            for (int e = rowptr[row0]; e < rowptr[row0 + 1]; ++e) col_type_out[e] = edge_type[col_eid[e]];
    }
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) tile_grps[n_tiles] = 0;   // (the scan's last entry)
    const int tid = threadIdx.x, k = tid >> 4, i = tid & 15;
    const int64_t row = (int64_t)blockIdx.x * 64 + tid;
    const int64_t tile = (int64_t)blockIdx.x * 4 + k;
    if (n_types > max_types || n_types > kMaxColTypes) {
        if (!FILL && i == 0 && tile < n_tiles) tile_grps[tile] = 0;
        return;
    }
    if (built_flag && blockIdx.x == 0 && threadIdx.x == 0) *built_flag = 1;
    // 16-bit counters: 2 x 5 KB of LDS per 64-row block (in-degrees beyond 2048 are not served by the kernel that reads this
    // structure -- tgnn_forward checks nn_max_in_degree --; both passes wrap alike, every store stays inside the tile's range)
    __shared__ unsigned short cnt[64][kMaxColTypes + 1];    // edges of (row, type); +1: odd stride
    __shared__ unsigned short pre[FILL ? 64 : 1][kMaxColTypes + 1];    // edges of the type in the tile's rows above this one
    __shared__ int base[4][kMaxColTypes + 1];    // first group of (tile, type)
    for (int t = 0; t < n_types; ++t) cnt[tid][t] = 0;
    int e0 = 0, e1 = 0;
    if (row < n) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
    if (!FILL && col_type_out) {
        for (int e = e0; e < e1; ++e) cnt[tid][col_type_out[e]]++;          // (this thread's own stores above)
    } else {
        for (int e = e0; e < e1; ++e) cnt[tid][col_type[e]]++;
    }
    __syncthreads();
    for (int t = i; t < n_types; t += 16) {
        int acc = 0;
        for (int r = 0; r < 16; ++r) {
            if constexpr (FILL) pre[k * 16 + r][t] = (unsigned short)acc;
            acc += cnt[k * 16 + r][t];
        }
        base[k][t] = (acc + 15) >> 4;            // (groups of the type: turned into offsets below)
    }
    __syncthreads();
    if (i == 0) {
        int acc = 0;
        for (int t = 0; t < n_types; ++t) { const int g = base[k][t]; base[k][t] = acc; acc += g; }
        base[k][n_types] = acc;
        if (!FILL && tile < n_tiles) tile_grps[tile] = acc + 1;
    }
    if (!FILL) return;
    __syncthreads();
    if (tile >= n_tiles) return;                 // whole 16-thread group: no barrier below
    const int64_t g0 = tile_grp_ptr[tile];
    const int n_edge_grps = base[k][n_types];
    // thread i: slot i of every group's sources, and row i's word of every group
    for (int t = 0; t < n_types; ++t)
        for (int g = base[k][t]; g < base[k][t + 1]; ++g) grp[(g0 + g) * 16 + i] = make_int2(-1, t << 16);
    const int deg = e1 - e0;
    grp[(g0 + n_edge_grps) * 16 + i] =
        make_int2(row < n ? __float_as_int((float)(deg > 0 ? deg : 1)) : -1, (n_types | kEgRoot) << 16 | 1 << i);
    // the 16 threads of a tile are lanes of one wavefront: their -1 stores above are ordered before the sources below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < n_types; ++t) {
        const int c = cnt[tid][t];
        if (c == 0) continue;
        const int p0 = pre[tid][t], p1 = p0 + c;                  // this row's slots of the type's sorted list
        for (int g = p0 >> 4; g <= (p1 - 1) >> 4; ++g) {
            const int lo = max(p0 - 16 * g, 0), hi = min(p1 - 16 * g, 16);
            grp[(g0 + base[k][t] + g) * 16 + i].y = t << 16 | (((1 << hi) - 1) & ~((1 << lo) - 1));
        }
        cnt[tid][t] = 0;
    }
    for (int e = e0; e < e1; ++e) {
        const int t = col_type[e];
        const int p = (unsigned short)(pre[tid][t] + cnt[tid][t]++);
        grp[(g0 + base[k][t] + (p >> 4)) * 16 + (p & 15)].x = col_src[e];
    }
}

// ------------------------------------------------------------------------------------------
// Small layouts (what the greedy loop scores every round: ~1 000 nodes, ~20 000 edges): the whole preparation -- both CSRs,
// the exact edge-type de-duplication, types in CSR order, the NNConv column structure -- in ONE launch instead of ~30 tiny
// ones (0.19 ms per layout of which the kernels themselves were a fifth): up to 16 resident blocks that pass through the
// phases together (grid barrier: counter + agent-scope release / acquire by one thread per block, ~1 us at 16 blocks).
// A single block was tried first: 150-200 us, every phase a chain of global round trips with one CU to hide them behind.
// Every output is bit-identical to what tgnn_csr_build / tgnn_edge_type_dedup / tgnn_gather_i32 / tgnn_nnconv_cols_build
// produce (tests/test_graph_prep_small.py).
// ------------------------------------------------------------------------------------------
constexpr int kSmallPrepThreads = 1024, kSmallPrepMaxNodes = 4096, kSmallPrepMaxBlocks = 16;
constexpr int kSmallPrepLocal = 8192;                 // slots of a block's LDS de-dup table (its <= 8192 edges, <= 4096 distinct)
constexpr int kSmallPrepGlobal = 8192;                // slots of the global table of block representatives (<= 1024 types)
constexpr int kSmallPrepMaxTypes = 1024;

struct SmallPrepArgs {
    const int64_t *adj_ei, *col_ei;                   // [2][E]: sources, then destinations
    const float *attr;                                // [Ea][fe]
    int64_t n, ea, ec;
    int fe, max_col_types;
    int *adj_rowptr, *adj_src, *adj_eid, *adj_type;   // CSR of the adjacency set, types in CSR order
    int *edge_type, *type_rep;                        // types in edge order; first edge of every type
    int *col_rowptr, *col_src, *col_eid;              // CSR of the collision set (self loops dropped)
    int *tile_col_ptr, *col_meta, *col_slot_src;      // NNConv column structure
    int *result_host;                                 // NULL, or the caller's pinned words (device address): written behind `result`, word 31 last
    int *tmp;                                         // tgnn_graph_prep_small_tmp_ints()
    int *result;                                      // [8]: n_types, adj_err, col_err, n_col_edges, max_in_degree, cols_built, fallback
    unsigned *ctr;                                    // [2] barrier counter, exit counter: zero before the first use, zero on return
};

// block-wide exclusive scan of a[0 .. m) in LDS, m <= 5 * 1024; returns the total
__device__ int small_block_scan(int *a, int m, int *wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int v[5], tsum = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        v[k] = tid * 5 + k < m ? a[tid * 5 + k] : 0;
        tsum += v[k];
    }
    const int incl = wave_inclusive_scan(tsum);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < kSmallPrepThreads / 64; ++w) {
        if (w < wave) woff += wave_tot[w];
        total += wave_tot[w];
    }
    int run = woff + incl - tsum;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (tid * 5 + k < m) a[tid * 5 + k] = run;
        run += v[k];
    }
    __syncthreads();
    return total;
}

// all blocks resident (<= 16); ordinary loads / stores on both sides: one thread per block releases / acquires at agent scope
__device__ __forceinline__ void small_prep_barrier(unsigned *ctr, unsigned &target, unsigned nblk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave's stores are in L2 before thread 0 writes L2 back
    __syncthreads();
    target += nblk;
    if (nblk > 1 && threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(kSmallPrepThreads) void graph_prep_small_kernel(SmallPrepArgs A) {
    extern __shared__ int sm[];
    int *rpA = sm;                                    // [n + 1] row starts of the adjacency CSR (every block scans its own copy)
    int *rpC = rpA + kSmallPrepMaxNodes + 8;
    int *ltab = rpC + kSmallPrepMaxNodes + 8;         // [kSmallPrepLocal] this block's de-dup table; later: type of a global slot
    int *gslot = ltab + kSmallPrepLocal;              // [kSmallPrepLocal] global slot of a local slot; later: the column counts
    int *misc = gslot + kSmallPrepLocal;              // [64]
    int *wave_tot = misc;
    const int tid = threadIdx.x, NT = kSmallPrepThreads, G = gridDim.x, blk = blockIdx.x;
    const int gtid = blk * NT + tid, GT = G * NT;
    const int n = (int)A.n, ea = (int)A.ea, ec = (int)A.ec;
    const int emax = ea > ec ? ea : ec, nt16 = (n + kColTileRows - 1) / kColTileRows;
    // global scratch
    int *g_cntA = A.tmp, *g_cntC = g_cntA + n + 1;
    int *g_rankA = g_cntC + n + 1, *g_rankC = g_rankA + ea;
    int *t_eidA = g_rankC + ec, *t_eidC = t_eidA + ea;
    int *slot_of_edge = t_eidC + ec;
    int *g_table = slot_of_edge + ea;
    int *g_tile_cols = g_table + kSmallPrepGlobal;
    int *g_flags = g_tile_cols + nt16 + 1;            // 0 adj_err, 1 col_err, 2 fallback, 3 claimed global slots
    (void)emax;
    unsigned target = 0;
#ifdef TGNN_PREP_TIMING
    unsigned long long tl = wall_clock64();
    int tslot = 8;
#define TGNN_PT { const unsigned long long now_ = wall_clock64(); if (gtid == 0) A.result[tslot] = (int)(now_ - tl); ++tslot; tl = now_; }
#else
#define TGNN_PT
#endif

    // ---- phase 0: zero the counters
    for (int i = gtid; i <= n; i += GT) g_cntA[i] = g_cntC[i] = 0;
    for (int i = gtid; i < kSmallPrepGlobal; i += GT) g_table[i] = -1;
    if (gtid < 16) g_flags[gtid] = 0;
    small_prep_barrier(A.ctr, target, G);

    TGNN_PT
    // ---- phase 1: in-degree counts; the value the atomic returns is the edge's arrival rank inside its row (csr_count_kernel)
    for (int i = gtid; i < ea; i += GT) {
        const int64_t s = A.adj_ei[i], d = A.adj_ei[(int64_t)ea + i];
        int r = -1;
        if (s < 0 || s >= n || d < 0 || d >= n) g_flags[0] = 1;
        else r = atomicAdd(&g_cntA[d], 1);
        g_rankA[i] = r;
    }
    for (int i = gtid; i < ec; i += GT) {
        const int64_t s = A.col_ei[i], d = A.col_ei[(int64_t)ec + i];
        int r = -1;
        if (s < 0 || s >= n || d < 0 || d >= n) g_flags[1] = 1;
        else if (s != d) r = atomicAdd(&g_cntC[d], 1);
        g_rankC[i] = r;
    }
    small_prep_barrier(A.ctr, target, G);

    TGNN_PT
    // ---- phase 2: row starts (every block its own copy in LDS); block 0 writes them out
    int maxdeg = 0;
    for (int i = tid; i <= n; i += NT) {
        rpA[i] = g_cntA[i];
        rpC[i] = g_cntC[i];
        if (i < n) maxdeg = max(maxdeg, rpA[i]);
    }
    if (tid < 64) misc[32 + (tid & 15)] = 0;
    __syncthreads();
    atomicMax(&misc[32], maxdeg);
    const int ea_valid = small_block_scan(rpA, n + 1, wave_tot);
    const int ec_valid = small_block_scan(rpC, n + 1, wave_tot);
    maxdeg = misc[32];
    if (blk == 0)
        for (int i = tid; i <= n; i += NT) {
            A.adj_rowptr[i] = rpA[i];
            A.col_rowptr[i] = rpC[i];
        }

    TGNN_PT
    // ---- phase 3: edge numbers in arrival order; phase 4: every edge finds its rank inside its row by edge number -- the
    //      original edge order, which is what the reference's scatter sees (csr_fill_kernel + csr_sort_rows_kernel)
    for (int i = gtid; i < ea; i += GT) {
        const int r = g_rankA[i];
        if (r >= 0) t_eidA[rpA[A.adj_ei[(int64_t)ea + i]] + r] = i;
    }
    for (int i = gtid; i < ec; i += GT) {
        const int r = g_rankC[i];
        if (r >= 0) t_eidC[rpC[A.col_ei[(int64_t)ec + i]] + r] = i;
    }
    small_prep_barrier(A.ctr, target, G);
    TGNN_PT
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t *ei = pass ? A.col_ei : A.adj_ei;
        const int e = pass ? ec : ea;
        const int *rank_in = pass ? g_rankC : g_rankA, *t_eid = pass ? t_eidC : t_eidA, *rp = pass ? rpC : rpA;
        int *o_src = pass ? A.col_src : A.adj_src, *o_eid = pass ? A.col_eid : A.adj_eid;
        for (int i = gtid; i < e; i += GT) {
            if (rank_in[i] < 0) continue;
            const int d = (int)ei[(int64_t)e + i], b = rp[d], en = rp[d + 1];
            int rank = 0;
            for (int q = b; q < en; q += 8) {          // 8 keys of the row in flight
                int kq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kq[u] = q + u < en ? t_eid[q + u] : 0x7fffffff;
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += kq[u] < i;
            }
            o_src[b + rank] = (int)ei[i];
            o_eid[b + rank] = i;
        }
    }

    TGNN_PT
    // ---- phase 5: exact de-duplication of the attribute rows (tgnn_edge_type_dedup).  Level 1: the block's contiguous share
    //      of the edges in its LDS table (slot = smallest of its edges with that content); level 2: the block's
    //      representatives in the global table (slot = smallest edge overall)
    auto same_row = [&](int a, int b) {                 // all loads of both rows before the first compare (no early exit)
        const float *ra = A.attr + (int64_t)a * A.fe, *rb = A.attr + (int64_t)b * A.fe;
        uint32_t diff = 0;
#pragma unroll 8
        for (int k = 0; k < A.fe; ++k) diff |= canon_bits(ra[k]) ^ canon_bits(rb[k]);
        return diff == 0;
    };
    for (int i = tid; i < kSmallPrepLocal; i += NT) ltab[i] = -1;
    if (tid == 0) misc[40] = 0;                        // distinct rows of this block
    __syncthreads();
    const int e_lo = (int)((int64_t)ea * blk / G), e_hi = (int)((int64_t)ea * (blk + 1) / G);
    int my_l[8];                                        // local slots of this thread's (at most 8) edges
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = e_lo + u * NT + tid;
        my_l[u] = -1;
        if (i < e_hi) {
            uint64_t h = 0xCBF29CE484222325ull;
#pragma unroll 8
            for (int k = 0; k < A.fe; ++k) h = mix64(h, canon_bits(A.attr[(int64_t)i * A.fe + k]));
            uint32_t sl = (uint32_t)(h ^ (h >> 32)) & (kSmallPrepLocal - 1);
            int probes = 0;
            while (true) {
                int cur = ltab[sl];
                if (cur < 0) {
                    const int prev = atomicCAS(&ltab[sl], -1, i);
                    if (prev < 0) {
                        if (atomicAdd(&misc[40], 1) >= kSmallPrepLocal / 2) g_flags[2] = 1;
                        break;
                    }
                    cur = prev;
                }
                if (cur == i || same_row(cur, i)) {
                    if (i < cur) atomicMin(&ltab[sl], i);
                    break;
                }
                sl = (sl + 1) & (kSmallPrepLocal - 1);
                if (++probes >= kSmallPrepLocal) {
                    g_flags[2] = 1;
                    break;
                }
            }
            my_l[u] = (int)sl;
        }
    }
    __syncthreads();
    for (int sl = tid; sl < kSmallPrepLocal; sl += NT) {
        const int i = ltab[sl];
        if (i < 0) continue;
        uint64_t h = 0xCBF29CE484222325ull;
#pragma unroll 8
        for (int k = 0; k < A.fe; ++k) h = mix64(h, canon_bits(A.attr[(int64_t)i * A.fe + k]));
        uint32_t slot = (uint32_t)(h ^ (h >> 32)) & (kSmallPrepGlobal - 1);
        int probes = 0;
        while (true) {
            int cur = __hip_atomic_load(&g_table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                const int prev = atomicCAS(&g_table[slot], -1, i);
                if (prev < 0) {
                    if (atomicAdd(&g_flags[3], 1) >= kSmallPrepMaxTypes) g_flags[2] = 1;
                    break;
                }
                cur = prev;
            }
            if (cur == i || same_row(cur, i)) {
                if (i < cur) atomicMin(&g_table[slot], i);
                break;
            }
            slot = (slot + 1) & (kSmallPrepGlobal - 1);
            if (++probes >= kSmallPrepGlobal) {
                g_flags[2] = 1;
                break;
            }
        }
        gslot[sl] = (int)slot;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = e_lo + u * NT + tid;
        if (i < e_hi) slot_of_edge[i] = my_l[u] >= 0 ? gslot[my_l[u]] : 0;
    }
    small_prep_barrier(A.ctr, target, G);

    TGNN_PT
    // ---- phase 6: types are numbered in the order of their first edges: rank of a representative among the representatives
    //      (every block for itself: type of a global slot in LDS)
    const bool fallback = __hip_atomic_load(&g_flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    int *type_of = ltab, *reps = gslot, *rep_slot = gslot + kSmallPrepMaxTypes, *occ = gslot + 2 * kSmallPrepMaxTypes;
    int n_types = 0;
    if (!fallback) {
        constexpr int per = kSmallPrepGlobal / kSmallPrepThreads;
        int c = 0;
        for (int k = 0; k < per; ++k) {
            type_of[tid * per + k] = g_table[tid * per + k];
            c += type_of[tid * per + k] >= 0;
        }
        occ[tid] = c;
        __syncthreads();
        n_types = small_block_scan(occ, NT, wave_tot);       // (<= kSmallPrepMaxTypes: more would have raised the flag)
        int at = occ[tid];
        for (int k = 0; k < per; ++k) {
            const int sl = tid * per + k;
            if (type_of[sl] >= 0) {
                reps[at] = type_of[sl];
                rep_slot[at] = sl;
                ++at;
            }
        }
        __syncthreads();
        if (tid < n_types) {
            const int mine = reps[tid];
            int rank = 0;
            for (int j = 0; j < n_types; ++j) rank += reps[j] < mine;
            if (blk == 0) A.type_rep[rank] = mine;
            type_of[rep_slot[tid]] = rank;
        }
        __syncthreads();
        // ---- phase 7: types in edge order and in CSR order
        for (int i = gtid; i < ea; i += GT) A.edge_type[i] = type_of[slot_of_edge[i]];
        for (int p = gtid; p < ea; p += GT) A.adj_type[p] = p < ea_valid ? type_of[slot_of_edge[A.adj_eid[p]]] : 0;
    }
    small_prep_barrier(A.ctr, target, G);

    TGNN_PT
    // ---- phase 8: NNConv column structure (nnconv_col_kernel, both passes): groups of tiles round the blocks, one thread per row
    const bool cols = !fallback && n_types <= A.max_col_types && n_types <= kMaxColTypes;
    if (cols) {
        // counts [rows of the group][ld], ld = (types + 1) | 1 (odd: the per-row walks hit distinct banks)
        const int ld = (n_types + 1) | 1;
        int tiles_pp = (kSmallPrepLocal - 2 * 64 * (kMaxColTypes + 1)) / (16 * ld);
        tiles_pp = tiles_pp > 64 ? 64 : tiles_pp;
        const int rows_pp = tiles_pp * 16;
        int *cnt = gslot;                              // (the de-dup lists are done with)
        int *maxm = cnt + rows_pp * ld;                // [tiles_pp][kMaxColTypes]
        int *base = maxm + 64 * kMaxColTypes;          // [tiles_pp][kMaxColTypes + 1]
        int *tile_ptr = rpC;                           // [n_tiles + 1] (the collision row starts are done with)
        const int k = tid >> 4, i = tid & 15;          // tile of the group, row of the tile
        const bool act = tid < rows_pp;
        for (int fill = 0; fill < 2; ++fill) {
            for (int r0 = blk * rows_pp; r0 < n; r0 += G * rows_pp) {
                const int row = r0 + tid, tile = r0 / kColTileRows + k;
                int e0 = 0, e1 = 0;
                if (act) {
                    for (int t = 0; t < n_types; ++t) cnt[tid * ld + t] = 0;
                    if (row < n) {
                        e0 = rpA[row];
                        e1 = rpA[row + 1];
                    }
                    for (int e = e0; e < e1; e += 8) {
                        int ty[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) ty[u] = e + u < e1 ? A.adj_type[e + u] : -1;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (ty[u] >= 0) cnt[tid * ld + ty[u]]++;
                    }
                }
                __syncthreads();
                if (act)
                    for (int t = i; t < n_types; t += 16) {
                        int m = 0;
                        for (int r = 0; r < 16; ++r) m = max(m, cnt[(k * 16 + r) * ld + t]);
                        maxm[k * kMaxColTypes + t] = m;
                    }
                __syncthreads();
                if (act && i == 0) {
                    int acc = 0;
                    for (int t = 0; t < n_types; ++t) {
                        base[k * (kMaxColTypes + 1) + t] = acc;
                        acc += maxm[k * kMaxColTypes + t];
                    }
                    base[k * (kMaxColTypes + 1) + n_types] = acc;
                    if (!fill && tile < nt16) g_tile_cols[tile] = acc + 1;
                }
                __syncthreads();
                if (fill && act && tile < nt16) {
                    const int64_t c0 = tile_ptr[tile];
                    const int n_edge_cols = base[k * (kMaxColTypes + 1) + n_types];
                    for (int c = 0; c < n_edge_cols; ++c) A.col_slot_src[(c0 + c) * 16 + i] = -1;
                    for (int t = i; t < n_types; t += 16) {
                        const int m = maxm[k * kMaxColTypes + t];
                        for (int r = 0; r < m; ++r)
                            A.col_meta[c0 + base[k * (kMaxColTypes + 1) + t] + r] = t | (r == 0 ? 1 << 8 : 0) | (r == m - 1 ? 1 << 9 : 0);
                    }
                    const int deg = e1 - e0;
                    A.col_slot_src[(c0 + n_edge_cols) * 16 + i] = row < n ? __float_as_int((float)(deg > 0 ? deg : 1)) : -1;
                    if (i == 0) A.col_meta[c0 + n_edge_cols] = n_types | (1 << 8) | (1 << 9) | (1 << 10);
                    // the 16 threads of a tile are lanes of one wavefront: their -1 stores above are ordered before these
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int t = 0; t < n_types; ++t) cnt[tid * ld + t] = 0;
                    for (int e = e0; e < e1; e += 8) {
                        int ty[8], sr[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            ty[u] = e + u < e1 ? A.adj_type[e + u] : -1;
                            sr[u] = e + u < e1 ? A.adj_src[e + u] : 0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (ty[u] >= 0) {
                                const int r = cnt[tid * ld + ty[u]]++;
                                A.col_slot_src[(c0 + base[k * (kMaxColTypes + 1) + ty[u]] + r) * 16 + i] = sr[u];
                            }
                    }
                }
                __syncthreads();
            }
            if (!fill) {
                small_prep_barrier(A.ctr, target, G);
                for (int t = tid; t <= nt16; t += NT) tile_ptr[t] = t < nt16 ? g_tile_cols[t] : 0;
                __syncthreads();
                small_block_scan(tile_ptr, nt16 + 1, wave_tot);
                if (blk == 0)
                    for (int t = tid; t <= nt16; t += NT) A.tile_col_ptr[t] = tile_ptr[t];
                __syncthreads();
            }
        }
    } else {
        small_prep_barrier(A.ctr, target, G);          // (every block passes the same barriers)
    }
    TGNN_PT
    if (blk == 0 && tid == 0) {
        A.result[0] = n_types;
        A.result[1] = g_flags[0];
        A.result[2] = g_flags[1];
        A.result[3] = ec_valid;
        A.result[4] = maxdeg;
        A.result[5] = cols ? 1 : 0;
        A.result[6] = fallback ? 1 : 0;
        if (A.result_host) {                               // [r6] no copy, no stream synchronise: the host polls word 31 (tgnn_graph_prep_wait)
            const int w[7] = {n_types, g_flags[0], g_flags[1], ec_valid, maxdeg, cols ? 1 : 0, fallback ? 1 : 0};
#pragma unroll
            for (int k = 0; k < 7; ++k) __hip_atomic_store(A.result_host + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int k = 7; k < 31; ++k) __hip_atomic_store(A.result_host + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(A.result_host + 31, kPrepWordsMagic, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // the last block out re-arms the counters for the next call
    __syncthreads();
    if (tid == 0 && atomicAdd(&A.ctr[1], 1u) == (unsigned)G - 1) {
        __hip_atomic_store(&A.ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&A.ctr[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static inline unsigned grid_for(int64_t n, int threads = 256, int cap = 256 * 16) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

static inline uint32_t dedup_table_size(int64_t e) {
    uint64_t want = (uint64_t)(e > 512 ? e : 512) * 2;
    uint32_t cap = 1024;
    while (cap < want) cap <<= 1;
    return cap;
}


// ------------------------------------------------------------------------------------------
// Both CSRs of a layout in four kernels and no device-scope returning atomic (tgnn_graph_prep).
//
// tgnn_csr_build ranks every edge inside its row with a returning atomicAdd on a global counter; those are served by the
// memory side of the fabric on this chip (~50 us per 1.25 M), and the pass is followed by a scan, a scatter and a per-row sort:
// seven launches and ~120 us per edge set at 100k nodes.  Here the edges go through BUCKETS of 512 destination rows:
//   1. bk_hist:    every block counts its contiguous share of the edges per bucket in LDS          -> hist[set][bucket][block]
//   2. one exclusive scan over all of hist (both sets): where each (bucket, block) cell's edges go
//   3. bk_scatter: the same blocks hand their edges out to the cells (LDS counters) as records (edge number, source, row in
//                  bucket); order inside a cell: whatever the LDS atomics give
//   4. bk_rows:    one block per bucket counts its rows, scans them (-> rowptr), places the records by row in LDS and sorts
//                  every row by edge number -- the original order, the one the reference's scatter sees and tgnn_csr_build
//                  restores -- then writes col_src / col_eid of the bucket in one coalesced sweep.
// Both edge sets ride in the same launches (blockIdx.y).  Bit-identical to tgnn_csr_build (tests/test_graph_prep_small.py).
// A bucket with more edges than the LDS arrays hold (very skewed in-degrees) takes a slow in-place path.
// ------------------------------------------------------------------------------------------
// (256 rows and 60 KB of LDS per bucket -- two blocks per CU -- where the average in-degree allows it; else 512 rows)
constexpr int kBkLogMax = 9, kBkRows = 1 << kBkLogMax, kBkMaxBuckets = 4096, kBkThreads = 1024;
constexpr int kBkCap = 15360;                        // records of a bucket sorted in LDS: 10 B each = 150 KB (512 rows)
constexpr int kBkCapHalf = 7168;                     // ... 70 KB (256 rows)
struct BkSet {
    const int64_t *ei;                               // [2][e]
    int64_t e;
    int drop_self;
    int *rowptr, *col_src, *col_eid, *err_flag;
    int *rec_eid, *rec_src;
    unsigned short *rec_row;
    int *total_out, *max_deg_out;                    // (optional) edges kept; largest in-degree (atomicMax into a zeroed word)
};
struct BkArgs {
    BkSet set[2];
    int64_t n, n_src;
    int nb, nblk;                                    // buckets, edge blocks
    int lg;                                          // log2 of the rows per bucket: 8 or 9
    int cap;                                         // <= cap_lds (tests lower it to reach the slow path)
    int cap_lds;                                     // records the LDS arrays of bk_rows hold: kBkCap / kBkCapHalf
    int *hist;                                       // [2][nb][nblk] + 1
    const int *tile_off;                             // [r6] NULL: hist is the finished scan; else + tile_off[i / kScanTile] (scan_tiles_fused_kernel)
    unsigned *scan_ticket;                           // ... and its ticket word, cleared by bk_hist_kernel
};
__device__ __forceinline__ int bk_scanned(const BkArgs &A, int64_t i) {
    const int v = A.hist[i];
    return A.tile_off ? v + A.tile_off[i / kScanTile] : v;
}

__global__ __launch_bounds__(kBkThreads) void bk_hist_kernel(BkArgs A) {
    extern __shared__ int bk_h[];
    const BkSet S = A.set[blockIdx.y];
    const int tid = threadIdx.x, blk = blockIdx.x;
    for (int b = tid; b < A.nb; b += kBkThreads) bk_h[b] = 0;
    __syncthreads();
    const int64_t i0 = S.e * blk / A.nblk, i1 = S.e * (blk + 1) / A.nblk;
    for (int64_t i = i0 + tid; i < i1; i += kBkThreads) {
        const int64_t s = S.ei[i], d = S.ei[S.e + i];
        if (s < 0 || s >= A.n_src || d < 0 || d >= A.n) {
            if (S.err_flag) *S.err_flag = 1;
        } else if (!(S.drop_self && s == d)) {
            atomicAdd(&bk_h[d >> A.lg], 1);
        }
    }
    __syncthreads();
    for (int b = tid; b < A.nb; b += kBkThreads) A.hist[((int64_t)blockIdx.y * A.nb + b) * A.nblk + blk] = bk_h[b];
    if (blk == 0 && blockIdx.y == 1 && tid == 0) {
        A.hist[(int64_t)2 * A.nb * A.nblk] = 0;
        if (A.scan_ticket) *A.scan_ticket = 0u;
    }
}

// The records of a (bucket, block) cell leave the block as ONE contiguous run: the block first sorts its edges by bucket in LDS
// (8 192 at a time: local ranks from LDS counters, a scan over the buckets), then every thread writes consecutive records.
// (Handing every edge straight to its global slot cost 48 us at 100k nodes: 6.7 M scattered 2- and 4-byte stores.)
constexpr int kBkChunk = kBkThreads * 8;
__global__ __launch_bounds__(kBkThreads) void bk_scatter_kernel(BkArgs A) {
    extern __shared__ int bk_h[];
    int *cur = bk_h, *lcnt = cur + A.nb, *lstart = lcnt + A.nb, *wtot = lstart + A.nb;   // [nb] each, [16]
    int *l_eid = wtot + 16, *l_src = l_eid + kBkChunk;
    unsigned short *l_row = reinterpret_cast<unsigned short *>(l_src + kBkChunk), *l_bkt = l_row + kBkChunk;
    const BkSet S = A.set[blockIdx.y];
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int base = bk_scanned(A, (int64_t)blockIdx.y * A.nb * A.nblk);      // where this set's records start in the scan
    for (int b = tid; b < A.nb; b += kBkThreads) cur[b] = bk_scanned(A, ((int64_t)blockIdx.y * A.nb + b) * A.nblk + blk) - base;
    const int64_t i0 = S.e * blk / A.nblk, i1 = S.e * (blk + 1) / A.nblk;
    const int per = (A.nb + kBkThreads - 1) / kBkThreads;                     // buckets per thread in the scan (<= 4)
    for (int64_t c0 = i0; c0 < i1; c0 += kBkChunk) {
        for (int b = tid; b < A.nb; b += kBkThreads) lcnt[b] = 0;
        __syncthreads();
        int e_b[8], e_r[8], e_s[8], e_row[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = c0 + tid + (int64_t)u * kBkThreads;
            e_b[u] = -1;
            if (i < i1) {
                const int64_t sv = S.ei[i], d = S.ei[S.e + i];
                if (!(sv < 0 || sv >= A.n_src || d < 0 || d >= A.n || (S.drop_self && sv == d))) {
                    e_b[u] = (int)(d >> A.lg);
                    e_row[u] = (int)(d & ((1 << A.lg) - 1));
                    e_s[u] = (int)sv;
                    e_r[u] = atomicAdd(&lcnt[e_b[u]], 1);
                }
            }
        }
        __syncthreads();
        {   // exclusive scan of lcnt over the buckets: thread t owns buckets [t per, (t + 1) per)
            int mine = 0;
            for (int q = 0; q < per; ++q) {
                const int b = tid * per + q;
                mine += b < A.nb ? lcnt[b] : 0;
            }
            const int incl = wave_inclusive_scan(mine);
            if ((tid & 63) == 63) wtot[tid >> 6] = incl;
            __syncthreads();
            int run = incl - mine;
            for (int w = 0; w < (tid >> 6); ++w) run += wtot[w];
            for (int q = 0; q < per; ++q) {
                const int b = tid * per + q;
                if (b < A.nb) {
                    lstart[b] = run;
                    run += lcnt[b];
                }
            }
        }
        __syncthreads();
        int total = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e_b[u] >= 0) {
                const int at = lstart[e_b[u]] + e_r[u];
                l_eid[at] = (int)(c0 + tid + (int64_t)u * kBkThreads);
                l_src[at] = e_s[u];
                l_row[at] = (unsigned short)e_row[u];
                l_bkt[at] = (unsigned short)e_b[u];
            }
        total = lstart[A.nb - 1] + lcnt[A.nb - 1];
        __syncthreads();
        for (int j = tid; j < total; j += kBkThreads) {
            const int b = l_bkt[j];
            const int g = cur[b] + (j - lstart[b]);
            S.rec_eid[g] = l_eid[j];
            S.rec_src[g] = l_src[j];
            S.rec_row[g] = l_row[j];
        }
        __syncthreads();
        for (int b = tid; b < A.nb; b += kBkThreads) cur[b] += lcnt[b];
        // (the next round's first barrier orders these updates before its reads)
    }
}

__global__ __launch_bounds__(kBkThreads) void bk_rows_kernel(BkArgs A) {
    extern __shared__ int bk_l[];
    const int rows = 1 << A.lg;
    int *cnt = bk_l, *off = bk_l + rows, *wtot = off + rows + 1, *key_s = wtot + 16, *val_s = key_s + A.cap_lds;
    unsigned short *row_s = reinterpret_cast<unsigned short *>(val_s + A.cap_lds);
    const BkSet S = A.set[blockIdx.y];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int base = bk_scanned(A, (int64_t)blockIdx.y * A.nb * A.nblk);
    const int lo = bk_scanned(A, ((int64_t)blockIdx.y * A.nb + b) * A.nblk) - base;
    const int hi = bk_scanned(A, ((int64_t)blockIdx.y * A.nb + b + 1) * A.nblk) - base;   // (the next set's start / the total behind the last)
    const int m = hi - lo;
    const bool in_lds = m <= A.cap;                                           // uniform
    if (tid < rows) cnt[tid] = 0;
    __syncthreads();
    // rows' counts
    if (in_lds) {
        for (int j = tid; j < m; j += kBkThreads) atomicAdd(&cnt[S.rec_row[lo + j]], 1);
    } else if (tid < rows) {
        int c = 0;
        for (int j = 0; j < m; ++j) c += S.rec_row[lo + j] == tid ? 1 : 0;
        cnt[tid] = c;
    }
    __syncthreads();
    // exclusive scan of the 512 counts (8 waves)
    int mine = 0, incl = 0;
    if (tid < rows) {
        mine = cnt[tid];
        incl = wave_inclusive_scan(mine);
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < rows) {
        int woff = 0;
        for (int w = 0; w < (tid >> 6); ++w) woff += wtot[w];
        off[tid] = woff + incl - mine;
        if (tid == rows - 1) off[rows] = woff + incl;
        const int64_t row = (int64_t)b * rows + tid;
        if (row < A.n) S.rowptr[row] = lo + off[tid];
        cnt[tid] = 0;
    }
    if (b == A.nb - 1 && tid == 0) {
        S.rowptr[A.n] = hi;
        if (S.total_out) *S.total_out = hi;
    }
    if (S.max_deg_out && tid < rows) {                                        // (whole waves: rows is a multiple of 64)
        int md = mine;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) md = max(md, __shfl_xor(md, d, 64));
        if ((tid & 63) == 0) wtot[8 + (tid >> 6)] = md;
    }
    __syncthreads();
    if (S.max_deg_out && tid == 0) {
        int md = 0;
        for (int w = 0; w < (rows >> 6); ++w) md = max(md, wtot[8 + w]);
        if (md > 0) atomicMax(S.max_deg_out, md);
    }
    if (in_lds) {
        for (int j = tid; j < m; j += kBkThreads) {
            const int r = S.rec_row[lo + j];
            const int at = off[r] + atomicAdd(&cnt[r], 1);
            key_s[at] = S.rec_eid[lo + j];
            val_s[at] = S.rec_src[lo + j];
            row_s[at] = (unsigned short)r;
        }
        __syncthreads();
        // every row back into edge order: a record's place in its row = how many of the row's records carry a smaller edge
        // number -- one thread per RECORD, independent LDS reads (an insertion sort by one thread per row was a chain of
        // dependent LDS round trips as long as the longest row of the wave: 16 of this kernel's 30 us)
        for (int j = tid; j < m; j += kBkThreads) {
            const int r = row_s[j], rb = off[r], re = off[r + 1], key = key_s[j];
            int rank = 0;
            for (int q = rb; q < re; ++q) rank += key_s[q] < key ? 1 : 0;
            S.col_eid[lo + rb + rank] = key;
            S.col_src[lo + rb + rank] = val_s[j];
        }
    } else if (tid < rows) {
        // too many edges for the LDS arrays: every row's thread walks the bucket's records itself (they lie in the order of the
        // scatter's atomics; the walk writes them by increasing edge number only after a sort in place)
        int *ke = S.col_eid + lo + off[tid], *ks = S.col_src + lo + off[tid];
        int c = 0;
        for (int j = 0; j < m; ++j)
            if (S.rec_row[lo + j] == tid) {
                ke[c] = S.rec_eid[lo + j];
                ks[c] = S.rec_src[lo + j];
                ++c;
            }
        for (int i = 1; i < c; ++i) {
            const int key = ke[i], val = ks[i];
            int j = i - 1;
            while (j >= 0 && ke[j] > key) {
                ke[j + 1] = ke[j];
                ks[j + 1] = ks[j];
                --j;
            }
            ke[j + 1] = key;
            ks[j + 1] = val;
        }
    }
}

static std::atomic<int> g_bk_cap{kBkCap};
// the preparation's side stream and its fork / join events: per device the stream, per calling thread and device the events
// (two threads preparing layouts on one device share the stream -- work of both is ordered on it, never wrong, at worst serial)
static int prep_side_stream(hipStream_t *st, hipEvent_t *fork, hipEvent_t *join) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    static thread_local hipEvent_t events[64][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TGNN_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> lock(mu);
        // ([r6] a LOWEST-priority stream was tried here -- the de-duplication's 1 M-thread insert kernel has the CSR chain's small scans
        //  queue behind it -- and withdrawn: with it the full GPU suite lost the mid-size persistent kernel's edge-weight wait (reason
        //  bits 4) once per run, in isolation never; another priority level is another pool of hardware queues, and which streams
        //  then share a queue with the persistent kernel's is no longer what _lib.concurrent_streams measured)
        if (!streams[dev] && hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) return TGNN_ERR_INVALID_ARG;
    }
    if (!events[dev][0])
        for (int k = 0; k < 2; ++k)
            if (hipEventCreateWithFlags(&events[dev][k], hipEventDisableTiming) != hipSuccess) return TGNN_ERR_INVALID_ARG;
    *st = streams[dev];
    *fork = events[dev][0];
    *join = events[dev][1];
    return TGNN_OK;
}

// the event behind tgnn_graph_prep's early copy of the result words: one per host thread and device
static thread_local hipEvent_t g_prep_words_ev[64] = {};
static hipError_t prep_words_event(hipEvent_t *ev) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!g_prep_words_ev[dev]) {
        e = hipEventCreateWithFlags(&g_prep_words_ev[dev], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    *ev = g_prep_words_ev[dev];
    return hipSuccess;
}

// [r6] ... or, when the words are written by a kernel of the preparation itself (nnconv_eg_kernel<false>), the pinned buffer whose
// word 31 the host polls: one per host thread and device, NULL = the event above
static thread_local volatile int32_t *g_prep_poll[64] = {};
static std::atomic<int> g_prep_poll_on{1};                   // tgnn_set_prep_words_poll

extern "C" int32_t tgnn_set_prep_words_poll(int32_t on) {
    if (on < 0) return g_prep_poll_on.load();
    return g_prep_poll_on.exchange(on ? 1 : 0);
}

extern "C" int tgnn_graph_prep_wait(tgnn_stream_t stream) {
    DeviceGuard guard__(stream);                             // (the event slot of the device tgnn_graph_prep recorded on: the stream's)
    int dev = 0;
    TGNN_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && g_prep_poll[dev]) {
        volatile int32_t *h = g_prep_poll[dev];
        g_prep_poll[dev] = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(const_cast<const int32_t *>(h) + 31, __ATOMIC_ACQUIRE) == kPrepWordsMagic) return TGNN_OK;
            __builtin_ia32_pause();
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        }
        // (not seen for two seconds: wait for the stream itself -- the kernel that writes the words is on it -- and look again)
        TGNN_CHECK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
        if (__atomic_load_n(const_cast<const int32_t *>(h) + 31, __ATOMIC_ACQUIRE) == kPrepWordsMagic) return TGNN_OK;
        set_error("tgnn_graph_prep_wait: the result words never reached the host buffer");
        return TGNN_ERR_INVALID_ARG;
    }
    hipEvent_t ev = nullptr;
    TGNN_CHECK_HIP(prep_words_event(&ev));
    TGNN_CHECK_HIP(hipEventSynchronize(ev));
    return TGNN_OK;
}

static int bk_blocks(int64_t e_max) {
    int64_t nblk = (e_max + 8191) / 8192;
    return (int)(nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk));
}
static int bk_log_rows(int64_t n, int64_t e_max) {   // 256-row buckets when they stay within the table sizes and ~half their LDS
    return (n <= (int64_t)256 * kBkMaxBuckets && e_max * 256 <= (int64_t)3584 * (n > 0 ? n : 1)) ? 8 : 9;
}
static size_t bk_workspace_bytes(int64_t n, int64_t ea, int64_t ec) {
    const int64_t rows = (int64_t)1 << bk_log_rows(n, ea > ec ? ea : ec);
    const int64_t nb = (n + rows - 1) / rows, nblk = bk_blocks(ea > ec ? ea : ec), cells = 2 * nb * nblk + 1;
    return align_up((size_t)cells * 4, 256) + scan_ws_ints(cells) * 4 + align_up((size_t)(ea > 0 ? ea : 1) * 10, 256) +
           align_up((size_t)(ec > 0 ? ec : 1) * 10, 256) + 2048;
}
static bool bk_fits(int64_t n) { return (n + kBkRows - 1) / kBkRows <= kBkMaxBuckets; }

// adjacency set -> (rowptr / col_src / col_eid / err) a, collision set (self loops dropped) -> c
static int csr_build_pair_bucketed(const int64_t *adj_ei, int64_t ea, const int64_t *col_ei, int64_t ec, int64_t n, int64_t n_src,
                                   int *a_rowptr, int *a_src, int *a_eid, int *a_err, int *c_rowptr, int *c_src, int *c_eid,
                                   int *c_err, int *a_max_deg, int *c_total, void *ws, size_t ws_bytes, hipStream_t s) {
    BkArgs A{};
    A.n = n; A.n_src = n_src;
    A.lg = bk_log_rows(n, ea > ec ? ea : ec);
    A.nb = (int)((n + ((int64_t)1 << A.lg) - 1) >> A.lg);
    A.nblk = bk_blocks(ea > ec ? ea : ec);
    A.cap_lds = A.lg == 8 ? kBkCapHalf : kBkCap;
    A.cap = g_bk_cap.load() < A.cap_lds ? g_bk_cap.load() : A.cap_lds;
    const int64_t cells = (int64_t)2 * A.nb * A.nblk + 1;
    Carver cv(ws, ws_bytes);
    A.hist = cv.take<int>(cells);
    int *scan_ws = cv.take<int>(scan_ws_ints(cells));
    const int64_t es[2] = {ea, ec};
    for (int k = 0; k < 2; ++k) {
        const int64_t e1 = es[k] > 0 ? es[k] : 1;
        unsigned char *rec = cv.take<unsigned char>((size_t)e1 * 10);
        A.set[k].rec_eid = reinterpret_cast<int *>(rec);
        A.set[k].rec_src = reinterpret_cast<int *>(rec + (size_t)e1 * 4);
        A.set[k].rec_row = reinterpret_cast<unsigned short *>(rec + (size_t)e1 * 8);
    }
    A.set[0].ei = adj_ei; A.set[0].e = ea; A.set[0].drop_self = 0;
    A.set[0].rowptr = a_rowptr; A.set[0].col_src = a_src; A.set[0].col_eid = a_eid; A.set[0].err_flag = a_err;
    A.set[1].ei = col_ei; A.set[1].e = ec; A.set[1].drop_self = 1;
    A.set[1].rowptr = c_rowptr; A.set[1].col_src = c_src; A.set[1].col_eid = c_eid; A.set[1].err_flag = c_err;
    A.set[0].max_deg_out = a_max_deg;
    A.set[1].total_out = c_total;
    const size_t lds_h = (size_t)A.nb * sizeof(int);
    // [r6] one launch for the scan of the cells where it is long enough to need three (the readers add the tile offsets themselves)
    const int64_t scan_tiles = (cells + kScanTile - 1) / kScanTile;
    const bool fused_scan = cells > kScanOneMax && scan_tiles <= kScanFusedTiles;
    if (fused_scan) {
        A.tile_off = scan_ws + align_up((size_t)scan_tiles, 64);
        A.scan_ticket = reinterpret_cast<unsigned *>(scan_ws + scan_ws_ints(cells) - 32);
    }
    bk_hist_kernel<<<dim3(A.nblk, 2), kBkThreads, lds_h, s>>>(A);
    if (fused_scan)
        scan_tiles_fused_kernel<<<(unsigned)scan_tiles, kScanThreads, 0, s>>>(A.hist, A.hist, scan_ws, const_cast<int *>(A.tile_off), cells,
                                                                             A.scan_ticket);
    else
        exclusive_scan_i32(A.hist, A.hist, cells, scan_ws, s);
    const size_t lds_s = (size_t)(3 * A.nb + 16 + 2 * kBkChunk) * sizeof(int) + (size_t)2 * kBkChunk * sizeof(unsigned short);
    static LdsOptIn site_s;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(bk_scatter_kernel, (int)(160 * 1024 - 256), site_s));
    bk_scatter_kernel<<<dim3(A.nblk, 2), kBkThreads, lds_s, s>>>(A);
    const size_t lds_r = (size_t)(2 * (1 << A.lg) + 1 + 16 + 2 * A.cap_lds) * sizeof(int) + (size_t)A.cap_lds * sizeof(unsigned short);
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(bk_rows_kernel, (int)(160 * 1024 - 256), site));
    bk_rows_kernel<<<dim3(A.nb, 2), kBkThreads, lds_r, s>>>(A);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

}  // namespace tgnn

using namespace tgnn;

#ifdef TGNN_DEBUG
extern "C" int32_t tgnn_debug_set_csr_bucket_cap(int32_t cap) {
    const int prev = g_bk_cap.load();
    if (cap >= 0) g_bk_cap.store(cap > kBkCap ? kBkCap : cap);
    return prev;
}
#endif

extern "C" size_t tgnn_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    return align_up((size_t)(n_nodes + 1) * 4, 256) + align_up((size_t)(n_edges > 0 ? n_edges : 1) * 4, 256) +
           scan_ws_ints(n_nodes + 1) * 4 + 1024;
}

extern "C" int tgnn_csr_build(const int64_t *edge_index, int64_t n_edges, int64_t n_nodes, int64_t n_src_nodes,
                              int drop_self_loops, int32_t *rowptr, int32_t *col_src, int32_t *col_eid, int32_t *err_flag, void *ws,
                              size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 0 && n_nodes < (1ll << 31) - 1, "n_nodes must fit int32");
    TGNN_CHECK_ARG(n_src_nodes >= n_nodes && n_src_nodes < (1ll << 31) - 1, "n_src_nodes must be >= n_nodes and fit int32");
    TGNN_CHECK_ARG(n_edges >= 0 && n_edges < (1ll << 31) - 1, "n_edges must fit int32");
    TGNN_CHECK_ARG(rowptr && (n_edges == 0 || (edge_index && col_src && col_eid)), "null pointer");
    if (ws_bytes < tgnn_csr_workspace_bytes(n_nodes, n_edges) || !ws) {
        set_error("tgnn_csr_build: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver cv(ws, ws_bytes);
    int *cnt = cv.take<int>(n_nodes + 1);
    int *rank = cv.take<int>(n_edges > 0 ? n_edges : 1);
    int *scan_ws = cv.take<int>(scan_ws_ints(n_nodes + 1));
    TGNN_CHECK_HIP(hipMemsetAsync(cnt, 0, (size_t)(n_nodes + 1) * 4, s));
    if (n_edges > 0)
        csr_count_kernel<<<grid_for(n_edges), 256, 0, s>>>(edge_index, n_edges, n_nodes, n_src_nodes, drop_self_loops, cnt,
                                                           rank, err_flag);
    exclusive_scan_i32(cnt, rowptr, n_nodes + 1, scan_ws, s);
    if (n_edges > 0) {
        csr_fill_kernel<<<grid_for(n_edges), 256, 0, s>>>(edge_index, n_edges, rowptr, rank, col_src, col_eid);
        csr_sort_rows_kernel<<<(unsigned)((n_nodes + kSortRows - 1) / kSortRows), kSortRows, 0, s>>>(rowptr, n_nodes, col_src,
                                                                                                  col_eid);
    }
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" size_t tgnn_edge_dedup_workspace_bytes(int64_t n_edges, int32_t fe) {
    (void)fe;
    return (size_t)dedup_table_size(n_edges) * 4 + align_up((size_t)n_edges * 4, 256) * 2 +
           scan_ws_ints(n_edges) * 4 + 2048 + (size_t)(kDedupListMax + 64) * 4 + 512;
}

// tgnn_edge_type_dedup for tgnn_graph_prep: four launches instead of eight (fill, insert, rank, assign); *fallback = 1 (and
// nothing usable in edge_type) when the layout has more than kDedupListMax distinct attribute rows
static int edge_type_dedup_listed(const float *edge_attr, int64_t n_edges, int32_t fe, int32_t *edge_type, int32_t *type_rep_edge,
                                  int32_t *n_types, int32_t *fallback, void *ws, size_t ws_bytes, hipStream_t s) {
    (void)ws_bytes;
    Carver cv(ws, ws_bytes);
    const uint32_t cap = dedup_table_size(n_edges);
    int *table = cv.take<int>(cap + 64);                 // + the list counter (table[cap]): one fill for both
    int *slot_of_edge = cv.take<int>(n_edges);
    int *first_rank = cv.take<int>(n_edges);
    int *rep_slots = cv.take<int>(kDedupListMax);
    TGNN_CHECK_HIP(hipMemsetAsync(table, 0xFF, (size_t)(cap + 64) * 4, s));
    if (fe <= 60)
        dedup_insert_kernel<<<grid_for(n_edges, kDedupRows), kDedupRows, (size_t)kDedupRows * (fe | 1) * 4, s>>>(
            edge_attr, n_edges, fe, table, cap - 1, slot_of_edge, table + cap, rep_slots);
    else
        dedup_insert_direct_kernel<<<grid_for(n_edges), 256, 0, s>>>(edge_attr, n_edges, fe, table, cap - 1, slot_of_edge, table + cap,
                                                                     rep_slots);
    dedup_rank_kernel<<<1, 1024, 0, s>>>(table, table + cap, rep_slots, first_rank, type_rep_edge, n_types, fallback);
    dedup_assign_kernel<<<grid_for(n_edges), 256, 0, s>>>(table, slot_of_edge, first_rank, n_edges, edge_type, type_rep_edge,
                                                          nullptr, fallback);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_edge_type_dedup(const float *edge_attr, int64_t n_edges, int32_t fe, int32_t *edge_type,
                                    int32_t *type_rep_edge, int32_t *n_types, void *ws, size_t ws_bytes,
                                    tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_edges >= 0 && n_edges < (1ll << 30), "n_edges out of range");
    TGNN_CHECK_ARG(fe >= 1, "fe must be >= 1");
    TGNN_CHECK_ARG(n_types, "null n_types");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_edges == 0) {
        TGNN_CHECK_HIP(hipMemsetAsync(n_types, 0, 4, s));
        return TGNN_OK;
    }
    TGNN_CHECK_ARG(edge_attr && edge_type && type_rep_edge, "null pointer");
    if (ws_bytes < tgnn_edge_dedup_workspace_bytes(n_edges, fe) || !ws) {
        set_error("tgnn_edge_type_dedup: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    Carver cv(ws, ws_bytes);
    const uint32_t cap = dedup_table_size(n_edges);
    int *table = cv.take<int>(cap);
    int *slot_of_edge = cv.take<int>(n_edges);
    int *is_first = cv.take<int>(n_edges);
    int *scan_ws = cv.take<int>(scan_ws_ints(n_edges));
    TGNN_CHECK_HIP(hipMemsetAsync(table, 0xFF, (size_t)cap * 4, s));
    if (fe <= 60)
        dedup_insert_kernel<<<grid_for(n_edges, kDedupRows), kDedupRows, (size_t)kDedupRows * (fe | 1) * 4, s>>>(
            edge_attr, n_edges, fe, table, cap - 1, slot_of_edge);
    else
        dedup_insert_direct_kernel<<<grid_for(n_edges), 256, 0, s>>>(edge_attr, n_edges, fe, table, cap - 1, slot_of_edge);
    dedup_mark_first_kernel<<<grid_for(n_edges), 256, 0, s>>>(table, slot_of_edge, n_edges, is_first);
    exclusive_scan_i32(is_first, is_first, n_edges, scan_ws, s);
    dedup_assign_kernel<<<grid_for(n_edges), 256, 0, s>>>(table, slot_of_edge, is_first, n_edges, edge_type,
                                                          type_rep_edge, n_types);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int tgnn_gather_i32(const int32_t *src, int64_t n_src, const int32_t *idx, int64_t n, int32_t *out,
                               tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    if (n <= 0) return TGNN_OK;
    TGNN_CHECK_ARG(src && idx && out, "null pointer");
    gather_i32_kernel<<<grid_for(n), 256, 0, static_cast<hipStream_t>(stream)>>>(src, n_src, idx, n, out);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int64_t tgnn_nnconv_cols_max_columns(int64_t n_nodes, int64_t n_edges) {
    // every edge column holds at least one edge; one root column per tile; slack: the kernel fetches index words in
    // groups of 4 columns and up to 3 groups ahead
    return n_edges + (n_nodes + kColTileRows - 1) / kColTileRows + 32;
}

extern "C" size_t tgnn_nnconv_cols_workspace_bytes(int64_t n_nodes) {
    const int64_t ntiles = (n_nodes + kColTileRows - 1) / kColTileRows;
    return align_up((size_t)(ntiles + 1) * 4, 256) + scan_ws_ints(ntiles + 1) * 4 + 1024;
}

extern "C" int tgnn_nnconv_cols_build(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type,
                                      int64_t n_nodes, int32_t n_types, int32_t *tile_col_ptr, int32_t *col_meta,
                                      int32_t *col_slot_src, void *ws, size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1, "n_nodes");
    TGNN_CHECK_ARG(n_types >= 0 && n_types <= kMaxColTypes, "the column NNConv structure supports at most 40 edge types");
    TGNN_CHECK_ARG(rowptr && tile_col_ptr && col_meta && col_slot_src, "null pointer");
    TGNN_CHECK_ARG(n_types == 0 || (col_src && col_type), "null CSR pointer");
    if (!ws || ws_bytes < tgnn_nnconv_cols_workspace_bytes(n_nodes)) {
        set_error("tgnn_nnconv_cols_build: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nt16 = (n_nodes + kColTileRows - 1) / kColTileRows;
    Carver cv(ws, ws_bytes);
    int *tile_cols = cv.take<int>(nt16 + 1);
    int *scan_ws = cv.take<int>(scan_ws_ints(nt16 + 1));
    TGNN_CHECK_HIP(hipMemsetAsync(tile_cols + nt16, 0, 4, s));
    const unsigned blocks = (unsigned)((n_nodes + 63) / 64);
    nnconv_col_kernel<false><<<blocks, 64, 0, s>>>(rowptr, col_src, col_type, n_nodes, n_types, tile_cols, nullptr,
                                                   nullptr, nullptr);
    exclusive_scan_i32(tile_cols, tile_col_ptr, nt16 + 1, scan_ws, s);
    nnconv_col_kernel<true><<<blocks, 64, 0, s>>>(rowptr, col_src, col_type, n_nodes, n_types, nullptr, tile_col_ptr,
                                                  col_meta, col_slot_src);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

extern "C" int64_t tgnn_nnconv_eg_max_groups(int64_t n_nodes, int64_t n_edges, int32_t n_types) {
    // per (tile, type) at most one group that is not full; one root group per tile; slack: the kernel fetches index words
    // in fours and up to 3 fours ahead
    const int64_t ntiles = (n_nodes + kColTileRows - 1) / kColTileRows;
    const int64_t a = n_edges / 16 + ntiles * (int64_t)n_types, b = n_edges;
    return (a < b ? a : b) + ntiles + 32;
}

// edge_type + col_eid given: the count pass also gathers the types into CSR order (col_type is then its OUTPUT)
static void launch_nnconv_eg_build(const int32_t *rowptr, const int32_t *col_src, int32_t *col_type, int64_t n_nodes,
                                   int32_t n_types, const int *n_types_dev, int max_types, int32_t *tile_grps,
                                   int32_t *tile_grp_ptr, int32_t *grp, int *scan_ws, int *built_flag, hipStream_t s,
                                   const int32_t *edge_type = nullptr, const int32_t *col_eid = nullptr,
                                   const int32_t *result_words = nullptr, int32_t *host_words = nullptr) {
    const int64_t nt16 = (n_nodes + kColTileRows - 1) / kColTileRows;
    const unsigned blocks = (unsigned)((n_nodes + 63) / 64);
    nnconv_eg_kernel<false><<<blocks, 64, 0, s>>>(rowptr, col_src, col_type, n_nodes, n_types, tile_grps, nullptr, nullptr,
                                                  n_types_dev, max_types, built_flag, edge_type, col_eid,
                                                  edge_type ? col_type : nullptr, result_words, host_words);
    exclusive_scan_i32(tile_grps, tile_grp_ptr, nt16 + 1, scan_ws, s);
    nnconv_eg_kernel<true><<<blocks, 64, 0, s>>>(rowptr, col_src, col_type, n_nodes, n_types, nullptr, tile_grp_ptr,
                                                 reinterpret_cast<int2 *>(grp), n_types_dev, max_types, nullptr);
}

extern "C" int tgnn_nnconv_eg_build(const int32_t *rowptr, const int32_t *col_src, const int32_t *col_type,
                                    int64_t n_nodes, int32_t n_types, int32_t *tile_grp_ptr, int32_t *grp, void *ws,
                                    size_t ws_bytes, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1, "n_nodes");
    TGNN_CHECK_ARG(n_types >= 0 && n_types <= kMaxColTypes, "the NNConv edge-group structure supports at most 40 edge types");
    TGNN_CHECK_ARG(rowptr && tile_grp_ptr && grp && ((uintptr_t)grp % 8) == 0, "null / misaligned pointer");
    TGNN_CHECK_ARG(n_types == 0 || (col_src && col_type), "null CSR pointer");
    if (!ws || ws_bytes < tgnn_nnconv_cols_workspace_bytes(n_nodes)) {
        set_error("tgnn_nnconv_eg_build: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nt16 = (n_nodes + kColTileRows - 1) / kColTileRows;
    Carver cv(ws, ws_bytes);
    int *tile_grps = cv.take<int>(nt16 + 1);
    int *scan_ws = cv.take<int>(scan_ws_ints(nt16 + 1));
    launch_nnconv_eg_build(rowptr, col_src, const_cast<int32_t *>(col_type), n_nodes, n_types, nullptr, kMaxColTypes, tile_grps,
                           tile_grp_ptr, grp, scan_ws, nullptr, s);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

// ------------------------------------------------------------------------------------------
// Sub-layout of the still unlabelled nodes (greedy assembly loop).
//
// Reference: BrickLayout.compute_sub_layout, /root/reference/tiling/brick_layout.py:248-286 -- the unlabelled nodes
// in ascending order become nodes 0..N'-1, an edge survives iff both ends are unlabelled, surviving edges keep their
// order and carry re-indexed ends and their attribute rows.  The reference runs four Python comprehensions with
// dict look-ups over all edges per round; here: flags -> exclusive scans -> scatters (stream compaction), everything
// stays on the device, only the three counts travel to the host.
// ------------------------------------------------------------------------------------------
__global__ void sub_node_flag_kernel(const int *__restrict__ alive, int64_t n, int *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x)
        flag[i] = i < n ? (alive[i] != 0) : 0;
}
__global__ void sub_edge_flag_kernel(const int64_t *__restrict__ ei, int64_t e, const int *__restrict__ alive, int64_t n,
                                     int *__restrict__ flag, int *__restrict__ err_flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= e; i += (int64_t)gridDim.x * blockDim.x) {
        int f = 0;
        if (i < e) {
            const int64_t a = ei[i], b = ei[e + i];
            if (a < 0 || a >= n || b < 0 || b >= n) {
                if (err_flag) *err_flag = 1;
            } else {
                f = alive[a] != 0 && alive[b] != 0;
            }
        }
        flag[i] = f;
    }
}
// pos = exclusive scan of the node flags (pos[n] = N')
__global__ void sub_node_scatter_kernel(const int *__restrict__ alive, const int *__restrict__ pos, int64_t n,
                                        const float *__restrict__ x, int fx, float *__restrict__ x_out,
                                        int64_t *__restrict__ inverse, int64_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (alive[i] == 0) continue;
        const int64_t j = pos[i];
        inverse[j] = i;
        for (int k = 0; k < fx; ++k) x_out[j * fx + k] = x[i * fx + k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = pos[n];
}
// epos = exclusive scan of the edge flags (epos[e] = E'); rows of the output index are E' apart
__global__ void sub_edge_scatter_kernel(const int64_t *__restrict__ ei, int64_t e, const int *__restrict__ eflag_pos,
                                        const int *__restrict__ alive, const int *__restrict__ npos,
                                        const float *__restrict__ attr, int fe, int64_t *__restrict__ ei_out,
                                        float *__restrict__ attr_out, int64_t *__restrict__ count_out) {
    const int64_t e_out = eflag_pos[e];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = ei[i], b = ei[e + i];
        if (alive[a] == 0 || alive[b] == 0) continue;
        const int64_t j = eflag_pos[i];
        ei_out[j] = npos[a];
        ei_out[e_out + j] = npos[b];
        if (attr)
            for (int k = 0; k < fe; ++k) attr_out[j * fe + k] = attr[i * fe + k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = e_out;
}

extern "C" size_t tgnn_sublayout_workspace_bytes(int64_t n_nodes, int64_t n_adj_edges, int64_t n_col_edges) {
    const int64_t m = n_adj_edges > n_col_edges ? n_adj_edges : n_col_edges;
    return align_up((size_t)(n_nodes + 1) * 4, 256) + align_up((size_t)(m + 1) * 4, 256) +
           scan_ws_ints((m > n_nodes ? m : n_nodes) + 1) * 4 + 1024;
}

extern "C" int tgnn_sublayout_compact(const int32_t *alive, int64_t n_nodes, const float *x, int32_t fx,
                                      const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_attr,
                                      int32_t fe, const int64_t *col_edge_index, int64_t n_col_edges, float *x_out,
                                      int64_t *inverse_out, int64_t *adj_out, float *adj_attr_out, int64_t *col_out,
                                      int64_t *counts_out, int32_t *err_flag, void *ws, size_t ws_bytes,
                                      tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_nodes < (1ll << 31) - 1 && fx >= 1, "node shape");
    TGNN_CHECK_ARG(n_adj_edges >= 0 && n_adj_edges < (1ll << 31) - 1 && n_col_edges >= 0 && n_col_edges < (1ll << 31) - 1,
                   "edge counts must fit int32");
    TGNN_CHECK_ARG(alive && x && x_out && inverse_out && counts_out, "null pointer");
    TGNN_CHECK_ARG(n_adj_edges == 0 || (adj_edge_index && adj_out && adj_edge_attr && adj_attr_out && fe >= 1), "adjacency arrays");
    TGNN_CHECK_ARG(n_col_edges == 0 || (col_edge_index && col_out), "collision arrays");
    if (!ws || ws_bytes < tgnn_sublayout_workspace_bytes(n_nodes, n_adj_edges, n_col_edges)) {
        set_error("tgnn_sublayout_compact: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t m = n_adj_edges > n_col_edges ? n_adj_edges : n_col_edges;
    Carver cv(ws, ws_bytes);
    int *npos = cv.take<int>(n_nodes + 1);
    int *epos = cv.take<int>(m + 1);
    int *scan_ws = cv.take<int>(scan_ws_ints((m > n_nodes ? m : n_nodes) + 1));
    sub_node_flag_kernel<<<grid_for(n_nodes + 1), 256, 0, s>>>(alive, n_nodes, npos);
    exclusive_scan_i32(npos, npos, n_nodes + 1, scan_ws, s);
    sub_node_scatter_kernel<<<grid_for(n_nodes), 256, 0, s>>>(alive, npos, n_nodes, x, fx, x_out, inverse_out, counts_out);
    for (int set = 0; set < 2; ++set) {
        const int64_t e = set == 0 ? n_adj_edges : n_col_edges;
        const int64_t *ei = set == 0 ? adj_edge_index : col_edge_index;
        if (e == 0) {
            TGNN_CHECK_HIP(hipMemsetAsync(counts_out + 1 + set, 0, sizeof(int64_t), s));
            continue;
        }
        sub_edge_flag_kernel<<<grid_for(e + 1), 256, 0, s>>>(ei, e, alive, n_nodes, epos, err_flag);
        exclusive_scan_i32(epos, epos, e + 1, scan_ws, s);
        sub_edge_scatter_kernel<<<grid_for(e), 256, 0, s>>>(ei, e, epos, alive, npos, set == 0 ? adj_edge_attr : nullptr, fe,
                                                            set == 0 ? adj_out : col_out, set == 0 ? adj_attr_out : nullptr,
                                                            counts_out + 1 + set);
    }
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* ---- a SHARD's rows of the next greedy round (tilingnn_amd.dist.compact_shard_device): the owned rows that are still
 *      unlabelled, and of the halo rows those that are unlabelled AND still the source of an edge whose two ends are -- the halo
 *      list of the round before shrinks with the layout.  alive_global over the current global numbering, gid [n_rows] = global
 *      number of every local row (owned rows first), edges in local numbering (destinations are owned rows). */
__global__ void shard_alive_own_kernel(const int *__restrict__ alive_global, const int64_t *__restrict__ gid, int64_t n_own,
                                       int64_t n_rows, int *__restrict__ alive_local) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x)
        alive_local[i] = i < n_own ? (alive_global[gid[i]] != 0) : 0;
}
__global__ void shard_alive_halo_kernel(const int *__restrict__ alive_global, const int64_t *__restrict__ gid, int64_t n_own,
                                        int64_t n_rows, const int64_t *__restrict__ ei, int64_t e, int *__restrict__ alive_local,
                                        int *__restrict__ err_flag) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < e; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = ei[k], dst = ei[e + k];
        if (src < 0 || src >= n_rows || dst < 0 || dst >= n_rows) { *err_flag = 1; continue; }
        if (src >= n_own && alive_global[gid[src]] != 0 && alive_global[gid[dst]] != 0) alive_local[src] = 1;   // (every writer stores 1)
    }
}
extern "C" int tgnn_shard_alive_rows(const int32_t *alive_global, const int64_t *gid, int64_t n_own, int64_t n_rows,
                                     const int64_t *adj_edge_index, int64_t n_adj_edges, const int64_t *col_edge_index,
                                     int64_t n_col_edges, int32_t *alive_local, int32_t *err_flag, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_own >= 0 && n_rows >= n_own && n_rows >= 1 && n_adj_edges >= 0 && n_col_edges >= 0, "shape");
    TGNN_CHECK_ARG(alive_global && gid && alive_local && err_flag, "null pointer");
    TGNN_CHECK_ARG((n_adj_edges == 0 || adj_edge_index) && (n_col_edges == 0 || col_edge_index), "null edge index");
    hipStream_t s = static_cast<hipStream_t>(stream);
    shard_alive_own_kernel<<<grid_for(n_rows), 256, 0, s>>>(alive_global, gid, n_own, n_rows, alive_local);
    if (n_adj_edges > 0)
        shard_alive_halo_kernel<<<grid_for(n_adj_edges), 256, 0, s>>>(alive_global, gid, n_own, n_rows, adj_edge_index, n_adj_edges,
                                                                     alive_local, err_flag);
    if (n_col_edges > 0)
        shard_alive_halo_kernel<<<grid_for(n_col_edges), 256, 0, s>>>(alive_global, gid, n_own, n_rows, col_edge_index, n_col_edges,
                                                                     alive_local, err_flag);
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* ---- small layouts: everything tgnn_forward needs of a layout in one launch -------------------------------------------- */
extern "C" int64_t tgnn_graph_prep_small_max_nodes(void) { return kSmallPrepMaxNodes; }
extern "C" int64_t tgnn_graph_prep_small_max_edges(void) { return (int64_t)kSmallPrepMaxBlocks * kSmallPrepLocal; }
extern "C" size_t tgnn_graph_prep_small_tmp_ints(int64_t n_nodes, int64_t n_adj_edges, int64_t n_col_edges) {
    return (size_t)(2 * (n_nodes + 1) + 3 * n_adj_edges + 2 * n_col_edges + kSmallPrepGlobal + (n_nodes + 15) / 16 + 1 + 16 + 64);
}

extern "C" int tgnn_graph_prep_small(const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_attr, int32_t fe,
                                     const int64_t *col_edge_index, int64_t n_col_edges, int64_t n_nodes, int32_t *adj_rowptr,
                                     int32_t *adj_src, int32_t *adj_eid, int32_t *adj_type, int32_t *edge_type,
                                     int32_t *type_rep_edge, int32_t *col_rowptr, int32_t *col_src, int32_t *col_eid,
                                     int32_t *tile_col_ptr, int32_t *col_meta, int32_t *col_slot_src, int32_t *tmp,
                                     int32_t *result, uint32_t *counters, int32_t *result_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && n_nodes <= kSmallPrepMaxNodes, "n_nodes");
    const int64_t emax_allowed = (int64_t)kSmallPrepMaxBlocks * kSmallPrepLocal;
    TGNN_CHECK_ARG(n_adj_edges >= 0 && n_adj_edges <= emax_allowed && n_col_edges >= 0 && n_col_edges <= emax_allowed, "edge counts");
    TGNN_CHECK_ARG(fe >= 1, "fe");
    TGNN_CHECK_ARG(adj_rowptr && col_rowptr && tile_col_ptr && col_meta && col_slot_src && tmp && result && counters, "null pointer");
    TGNN_CHECK_ARG(n_adj_edges == 0 || (adj_edge_index && adj_edge_attr && adj_src && adj_eid && adj_type && edge_type && type_rep_edge),
                   "null adjacency pointer");
    TGNN_CHECK_ARG(n_col_edges == 0 || (col_edge_index && col_src && col_eid), "null collision pointer");
    SmallPrepArgs A{};
    A.adj_ei = adj_edge_index; A.col_ei = col_edge_index; A.attr = adj_edge_attr;
    A.n = n_nodes; A.ea = n_adj_edges; A.ec = n_col_edges; A.fe = fe;
    A.max_col_types = tgnn_nnconv_cols_max_types();
    A.adj_rowptr = adj_rowptr; A.adj_src = adj_src; A.adj_eid = adj_eid; A.adj_type = adj_type;
    A.edge_type = edge_type; A.type_rep = type_rep_edge;
    A.col_rowptr = col_rowptr; A.col_src = col_src; A.col_eid = col_eid;
    A.tile_col_ptr = tile_col_ptr; A.col_meta = col_meta; A.col_slot_src = col_slot_src;
    A.tmp = tmp; A.result = result; A.ctr = counters;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int poll_dev = -1;
    if (result_host && g_prep_poll_on.load(std::memory_order_relaxed)) {   // (as in tgnn_graph_prep: the words stored by the kernel, polled by the host)
        int dev = 0;
        void *dptr = nullptr;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && hipHostGetDevicePointer(&dptr, result_host, 0) == hipSuccess && dptr) {
            if (g_prep_poll[dev]) TGNN_CHECK_HIP(hipDeviceSynchronize());   // (an earlier preparation nobody waited for)
            A.result_host = static_cast<int *>(dptr);
            __atomic_store_n(result_host + 31, 0, __ATOMIC_RELEASE);
            g_prep_poll[dev] = result_host;
            poll_dev = dev;
        } else {
            (void)hipGetLastError();                           // (not addressable by this device: the copy + event below instead)
        }
    }
    // blocks: ~2048 edges each, and enough of them that a block's share of the adjacency edges fits its LDS table
    const int64_t emax = n_adj_edges > n_col_edges ? n_adj_edges : n_col_edges;
    int64_t blocks = (emax + 2047) / 2048;
    const int64_t need = (n_adj_edges + kSmallPrepLocal - 1) / kSmallPrepLocal;
    if (blocks < need) blocks = need;
    if (blocks < 1) blocks = 1;
    if (blocks > kSmallPrepMaxBlocks) blocks = kSmallPrepMaxBlocks;
    const size_t lds = (size_t)(2 * (kSmallPrepMaxNodes + 8) + 2 * kSmallPrepLocal + 64) * sizeof(int);
    static LdsOptIn site;
    TGNN_CHECK_HIP(opt_in_dynamic_lds(graph_prep_small_kernel, (int)lds, site));
    // the counters are zero on entry whatever a previous call left behind (an aborted launch would otherwise make every
    // later preparation on this pair hang or pass its barriers early); on the per-device chain of spin-barrier kernels: beside
    // a persistent forward of another stream / thread neither could get all its blocks resident
    TGNN_CHECK_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(uint32_t), s));
    struct Ctx { SmallPrepArgs *A; unsigned blocks; size_t lds; } ctx{&A, (unsigned)blocks, lds};
    const int rc = spin_kernel_chain(s, [](void *c, hipStream_t st) {
        Ctx *x = static_cast<Ctx *>(c);
        graph_prep_small_kernel<<<x->blocks, kSmallPrepThreads, x->lds, st>>>(*x->A);
    }, &ctx, (int)blocks);
    if (rc != TGNN_OK) {
        if (poll_dev >= 0) g_prep_poll[poll_dev] = nullptr;   // (nothing was launched: nothing to wait for)
        return rc;
    }
    if (result_host && !A.result_host) {                     // the words by copy, tgnn_graph_prep_wait by event (as tgnn_graph_prep without polling)
        hipEvent_t ev_words = nullptr;
        TGNN_CHECK_HIP(prep_words_event(&ev_words));
        TGNN_CHECK_HIP(hipMemcpyAsync(result_host, result, 32 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        TGNN_CHECK_HIP(hipEventRecord(ev_words, s));
    }
    TGNN_CHECK_LAUNCH();
    return TGNN_OK;
}

/* ---- any size: the same preparation as one call that queues every launch itself (no host round trip in the middle: the
 *      column structure reads the type count from the device) ------------------------------------------------------------- */
// result[3] = collision edges kept; result[4] = the largest adjacency in-degree (the small-layout kernel's limit, the bound of
// the fp16-pair NNConv operands)
__global__ __launch_bounds__(256) void prep_result_kernel(const int *__restrict__ col_rowptr, const int *__restrict__ adj_rowptr,
                                                          int64_t n, int *__restrict__ result) {
    if (blockIdx.x == 0 && threadIdx.x == 0) result[3] = col_rowptr[n];
    int md = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        md = max(md, adj_rowptr[r + 1] - adj_rowptr[r]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) md = max(md, __shfl_xor(md, d, 64));
    __shared__ int wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = md;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(result + 4, max(max(wm[0], wm[1]), max(wm[2], wm[3])));
}

extern "C" size_t tgnn_graph_prep_workspace_bytes(int64_t n_nodes, int64_t n_adj_edges, int64_t n_col_edges, int32_t fe) {
    const int64_t emax = n_adj_edges > n_col_edges ? n_adj_edges : n_col_edges;
    return align_up(tgnn_csr_workspace_bytes(n_nodes, emax), 256) + align_up(bk_workspace_bytes(n_nodes, n_adj_edges, n_col_edges), 256) +
           align_up(tgnn_edge_dedup_workspace_bytes(n_adj_edges, fe), 256) +
           align_up(tgnn_nnconv_cols_workspace_bytes(n_nodes), 256) + 1024;
}

extern "C" int tgnn_graph_prep(const int64_t *adj_edge_index, int64_t n_adj_edges, const float *adj_edge_attr, int32_t fe,
                               const int64_t *col_edge_index, int64_t n_col_edges, int64_t n_nodes, int64_t n_src_nodes,
                               int32_t *adj_rowptr, int32_t *adj_src, int32_t *adj_eid, int32_t *adj_type, int32_t *edge_type,
                               int32_t *type_rep_edge,
                               int32_t *col_rowptr, int32_t *col_src, int32_t *col_eid, int32_t *tile_col_ptr, int32_t *col_meta,
                               int32_t *col_slot_src, int32_t *mid_tile_nb, uint32_t *mid_ent, int32_t *tile_grp_ptr, int32_t *grp,
                               void *ws, size_t ws_bytes, int32_t *result, int32_t *result_host, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG(n_nodes >= 1 && fe >= 1 && n_adj_edges >= 0 && n_col_edges >= 0, "shape");
    TGNN_CHECK_ARG(n_src_nodes >= n_nodes && n_src_nodes < (1ll << 31) - 1, "n_src_nodes must be >= n_nodes and fit int32");
    const bool want_cols = tile_col_ptr != nullptr, want_eg = tile_grp_ptr != nullptr;
    TGNN_CHECK_ARG(adj_rowptr && col_rowptr && result, "null pointer");
    TGNN_CHECK_ARG(want_cols || want_eg, "no NNConv structure asked for (type columns and / or edge groups)");
    TGNN_CHECK_ARG(!want_cols || (col_meta && col_slot_src), "type columns: tile_col_ptr, col_meta, col_slot_src go together");
    TGNN_CHECK_ARG(!want_eg || (grp && ((uintptr_t)grp % 8) == 0), "edge groups: tile_grp_ptr and an 8-byte aligned grp go together");
    TGNN_CHECK_ARG(want_cols || !(mid_tile_nb && mid_ent), "the mid-size batches are built from the type columns");
    if (!ws || ws_bytes < tgnn_graph_prep_workspace_bytes(n_nodes, n_adj_edges, n_col_edges, fe)) {
        set_error("tgnn_graph_prep: workspace too small");
        return TGNN_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t emax = n_adj_edges > n_col_edges ? n_adj_edges : n_col_edges;
    Carver cv(ws, ws_bytes);
    const size_t csr_b = tgnn_csr_workspace_bytes(n_nodes, emax), dd_b = tgnn_edge_dedup_workspace_bytes(n_adj_edges, fe),
                 col_b = tgnn_nnconv_cols_workspace_bytes(n_nodes);
    void *ws_csr = cv.take<unsigned char>(csr_b), *ws_dd = cv.take<unsigned char>(dd_b), *ws_col = cv.take<unsigned char>(col_b);
    TGNN_CHECK_HIP(hipMemsetAsync(result, 0, 32 * sizeof(int32_t), s));
    // The edge-type de-duplication reads only the attribute rows, the CSRs only the index arrays: two independent chains of
    // ~85 and ~125 us at 100k nodes, neither of which fills the chip (few-block scans, one block per 8 192 edges) -- the first
    // one runs on a side stream of the library's own (one per device, created once) between a fork and a join event
    hipStream_t s_side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (n_adj_edges > 0 && prep_side_stream(&s_side, &ev_fork, &ev_join) == TGNN_OK) {
        TGNN_CHECK_HIP(hipEventRecord(ev_fork, s));
        TGNN_CHECK_HIP(hipStreamWaitEvent(s_side, ev_fork, 0));
    } else {
        s_side = nullptr;
    }
    int rc;
    bool bucketed = false;
    // an error return between the fork and the join must not leave side-stream work in flight on the caller's buffers (the
    // Python side frees the scratch as soon as the call raises): the caller's stream waits for whatever was queued there
    auto bail = [&](int code) {
        if (s_side && hipEventRecord(ev_join, s_side) == hipSuccess) (void)hipStreamWaitEvent(s, ev_join, 0);
        return code;
    };
    // [r6] the host needs ~4 us per launch, so the order of the two chains' launches matters: at benchmark sizes the CSR chain is the longer
    // one (90 against 57 us at 100 000 nodes) and goes FIRST; on smaller layouts the chains are level (29 against 23 us at 10 000 nodes) and
    // the de-duplication, which has one launch more in front of its first kernel, goes first as it always did (measured the other way round:
    // +20 us at 10 000 nodes, profiles/r06_prep_trace_10000.txt)
    const bool csr_first = n_adj_edges + n_col_edges >= 1000000;
    auto queue_dedup = [&]() -> int {
        if (!s_side) return TGNN_OK;
        rc = edge_type_dedup_listed(adj_edge_attr, n_adj_edges, fe, edge_type, type_rep_edge, result + 0, result + 6, ws_dd, dd_b, s_side);
        if (rc != TGNN_OK) return bail(rc);
        if (hipEventRecord(ev_join, s_side) != hipSuccess) {
            (void)hipStreamSynchronize(s_side);
            set_error("tgnn_graph_prep: hipEventRecord failed");
            return TGNN_ERR_LAUNCH;
        }
        return TGNN_OK;
    };
    if (!csr_first) {
        const int rcd = queue_dedup();
        if (rcd != TGNN_OK) return rcd;
    }
    if (bk_fits(n_nodes) && n_adj_edges < (int64_t(1) << 31) - 1 && n_col_edges < (int64_t(1) << 31) - 1) {
        // both CSRs through 512-row buckets: four kernels + one scan, no device-scope returning atomic
        const size_t bk_b = bk_workspace_bytes(n_nodes, n_adj_edges, n_col_edges);
        void *ws_bk = cv.take<unsigned char>(bk_b);
        rc = csr_build_pair_bucketed(adj_edge_index, n_adj_edges, col_edge_index, n_col_edges, n_nodes, n_src_nodes, adj_rowptr, adj_src,
                                     adj_eid, result + 1, col_rowptr, col_src, col_eid, result + 2, result + 4, result + 3, ws_bk, bk_b, s);
        if (rc != TGNN_OK) return bail(rc);
        bucketed = true;
    } else {
        rc = tgnn_csr_build(adj_edge_index, n_adj_edges, n_nodes, n_src_nodes, 0, adj_rowptr, adj_src, adj_eid, result + 1, ws_csr, csr_b, stream);
        if (rc != TGNN_OK) return bail(rc);
        rc = tgnn_csr_build(col_edge_index, n_col_edges, n_nodes, n_src_nodes, 1, col_rowptr, col_src, col_eid, result + 2, ws_csr, csr_b, stream);
        if (rc != TGNN_OK) return bail(rc);
    }
    if (csr_first) {
        const int rcd = queue_dedup();
        if (rcd != TGNN_OK) return rcd;
    }
    if (s_side) {
        TGNN_CHECK_HIP(hipStreamWaitEvent(s, ev_join, 0));
    } else {
        rc = tgnn_edge_type_dedup(adj_edge_attr, n_adj_edges, fe, edge_type, type_rep_edge, result + 0, ws_dd, dd_b, stream);
        if (rc != TGNN_OK) return rc;
    }
    if (!bucketed) {                                          // (the bucketed builder leaves result[3], result[4] itself)
        int64_t rb = (n_nodes + 1023) / 1024;
        prep_result_kernel<<<(unsigned)(rb > 256 ? 256 : rb), 256, 0, s>>>(col_rowptr, adj_rowptr, n_nodes, result);
    }
    // [r5] result_host (pinned): the words the caller reads -- type count, index errors, collision slots, largest in-degree, the
    // fall-back flag: all of them final here -- travel to the host BEFORE the NNConv structure is built; the caller waits for
    // the copy alone (tgnn_graph_prep_wait) and queues its forward while the structure's three launches still run
    int32_t *host_words_dev = nullptr;                       // (the device's address of result_host when a kernel writes the words)
    if (result_host && want_eg && !want_cols && !(mid_tile_nb && mid_ent) && g_prep_poll_on.load(std::memory_order_relaxed)) {
        int dev = 0;
        void *dptr = nullptr;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && hipHostGetDevicePointer(&dptr, result_host, 0) == hipSuccess && dptr) {
            // (a preparation of this thread whose words were never waited for -- a caller that gave up in between -- may still have
            //  its store in flight: it must not land behind the reset below)
            if (g_prep_poll[dev]) TGNN_CHECK_HIP(hipDeviceSynchronize());   // (rare; whatever stream it was on)
            host_words_dev = static_cast<int32_t *>(dptr);
            __atomic_store_n(result_host + 31, 0, __ATOMIC_RELEASE);   // (the last preparation's words were waited for: nothing in flight)
            g_prep_poll[dev] = result_host;
        } else {
            (void)hipGetLastError();
        }
    }
    if (result_host && !host_words_dev) {
        hipEvent_t ev_words = nullptr;
        TGNN_CHECK_HIP(prep_words_event(&ev_words));
        if (s_side) {
            // [r6] on the side stream (idle by now): the copy's ~10 us of launch and gap leave the chain the NNConv structure hangs on
            TGNN_CHECK_HIP(hipEventRecord(ev_fork, s));
            TGNN_CHECK_HIP(hipStreamWaitEvent(s_side, ev_fork, 0));
            TGNN_CHECK_HIP(hipMemcpyAsync(result_host, result, 32 * sizeof(int32_t), hipMemcpyDeviceToHost, s_side));
            TGNN_CHECK_HIP(hipEventRecord(ev_words, s_side));
        } else {
            TGNN_CHECK_HIP(hipMemcpyAsync(result_host, result, 32 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            TGNN_CHECK_HIP(hipEventRecord(ev_words, s));
        }
    }
    // column structure with the type count read on the device
    const int64_t nt16 = (n_nodes + kColTileRows - 1) / kColTileRows;
    Carver cc(ws_col, col_b);
    int *tile_cols = cc.take<int>(nt16 + 1);
    int *scan_ws = cc.take<int>(scan_ws_ints(nt16 + 1));
    const unsigned blocks = (unsigned)((n_nodes + 63) / 64);
    const int max_types = tgnn_nnconv_cols_max_types();
    // (the first pass also gathers the types into CSR order -- adj_type -- and writes every entry the scan reads)
    if (want_cols) {
        nnconv_col_kernel<false><<<blocks, 64, 0, s>>>(adj_rowptr, adj_src, adj_type, n_nodes, 0, tile_cols, nullptr, nullptr, nullptr,
                                                       result + 0, max_types, nullptr, n_adj_edges > 0 ? edge_type : nullptr, adj_eid,
                                                       n_adj_edges > 0 ? adj_type : nullptr);
        exclusive_scan_i32(tile_cols, tile_col_ptr, nt16 + 1, scan_ws, s);
        nnconv_col_kernel<true><<<blocks, 64, 0, s>>>(adj_rowptr, adj_src, adj_type, n_nodes, 0, nullptr, tile_col_ptr, col_meta,
                                                      col_slot_src, result + 0, max_types, result + 5);
    }
    if (want_eg) {   // (behind the columns on the same stream: the same scratch; without them its first pass gathers the types)
        const bool gather_types = !want_cols && n_adj_edges > 0;
        launch_nnconv_eg_build(adj_rowptr, adj_src, adj_type, n_nodes, 0, result + 0, max_types, tile_cols, tile_grp_ptr, grp, scan_ws,
                               result + 10, s, gather_types ? edge_type : nullptr, gather_types ? adj_eid : nullptr,
                               host_words_dev ? result : nullptr, host_words_dev);
    }
    TGNN_CHECK_LAUNCH();
    if (mid_tile_nb && mid_ent) {
        // the batches of the mid-size persistent layer loop (forward_mid.hip), straight from the columns; result[8..9]
        rc = tgnn_mid_entries_build(tile_col_ptr, col_meta, col_slot_src, n_nodes, result + 5, mid_tile_nb, mid_ent, result + 8, stream);
        if (rc != TGNN_OK) return rc;
    }
    return TGNN_OK;
}
