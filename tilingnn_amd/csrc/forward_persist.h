// Shared pieces of the two persistent layer-loop kernels (forward_small.hip: layouts up to 4 096 nodes, one tile per block;
// forward_mid.hip: up to 65 536 nodes, several tiles per block): the per-layer parameter pack, the agent-scope (sc1) buffer
// accesses cross-block data travels through, and the BOUNDED spin every cross-block wait of these kernels goes through.
#pragma once
#include "tgnn_common.h"

namespace tgnn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

// ---- per-layer parameter pack (floats): small vectors, then the GIN MLP's MFMA weight image -----------------------------
constexpr int kSpBias = 0, kSpG1 = 32, kSpB1 = 64, kSpG2 = 96, kSpB2 = 128, kSpEps = 160, kSpGinB = 192, kSpGinW = 320;
constexpr int kSpGinFrags = 3 * 2 * 64 + 3 * 4 * 64 + 3 * 2 * 2 * 64;      // 1920 fragments of 16 bytes
constexpr int kSpStride = kSpGinW + kSpGinFrags * 4;                       // 8000 floats per layer

// running-statistics buffers of the two BatchNorms of every layer (updated once, after the last layer)
struct SmallRun {
    float *rm1, *rv1;
    int64_t *nbt1;
    float *rm2, *rv2;
    int64_t *nbt2;
};
struct SmallRunTab {
    SmallRun l[kMaxDepth];
};

constexpr int kCpSc1 = 16;       // cache-policy bit of the raw buffer builtins: sc1 = agent scope (coherent across the XCDs' L2s)
constexpr uint32_t kOob = 0x80000000u;   // offset outside the 2 GB window of every descriptor here: the load returns 0

__device__ __forceinline__ float4 ld_sc1_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kCpSc1));
}
__device__ __forceinline__ void st_sc1_f4(__amdgpu_buffer_rsrc_t r, uint32_t off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, kCpSc1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)0x80000000u, 0x00020000);
}

__device__ __forceinline__ f32x4 small_mma6(const bf16x8 *wpl, int plane_stride, const bf16x8 (&x)[3], f32x4 acc) {
    const bf16x8 w0 = wpl[0], w1 = wpl[plane_stride], w2 = wpl[2 * plane_stride];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x[0], acc, 0, 0, 0);   // lo . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[2], acc, 0, 0, 0);   // hi . lo
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[1], acc, 0, 0, 0);   // mid . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[0], acc, 0, 0, 0);   // mid . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[1], acc, 0, 0, 0);   // hi . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[0], acc, 0, 0, 0);   // hi . hi
    return acc;
}

// ---- bounded spins ------------------------------------------------------------------------------------------------------
// A persistent kernel's blocks wait for each other (grid barriers, tagged partial rows, a flag of the pre-pass kernel).  That is
// only sound while ALL of its blocks are resident, which the launch gate (spin_kernel_chain) guarantees inside ONE process --
// a second process on the same GPU, or any tenant holding CUs, is invisible to it.  So no wait spins without bound: after
// kSpinBudgetTicks of the device's constant 100 MHz clock (wall_clock64) a waiter gives up, ORs a reason code into the
// device's error word and lets its block run on WITHOUT waiting (so that the kernel, and every block still waiting for this
// one, terminates); blocks also leave their waits as soon as they see the word set by anybody.  The forward's results are
// then garbage: the host reads the word (tgnn_spin_error_poll), disables the persistent schedules and runs the general
// launch schedule instead (tilingnn_amd.TilinGNN does this for ML_Solver.predict).
constexpr unsigned long long kSpinBudgetTicksDefault = 25ull * 1000 * 1000;   // 0.25 s: ~500 x the longest forward these kernels run
constexpr unsigned kSpinErrBarrier = 1u, kSpinErrRows = 2u, kSpinErrWeights = 4u;

struct SpinCtx {
    unsigned *err;                       // device error word (never NULL)
    unsigned long long budget;           // ticks
    bool gave_up;                        // this thread has given up (it then stops waiting for good)
};
// true: go on waiting; false: stop (budget used up, or somebody reported an error)
__device__ __forceinline__ bool spin_continue(SpinCtx &sp, unsigned long long t0, unsigned iter, unsigned code) {
    if (sp.gave_up) return false;
    if ((iter & 63u) != 63u) return true;
    if (__hip_atomic_load(sp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        sp.gave_up = true;
        return false;
    }
    if (wall_clock64() - t0 > sp.budget) {
        __hip_atomic_fetch_or(sp.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sp.gave_up = true;
        return false;
    }
    return true;
}
// one thread: wait until *ctr >= target
__device__ __forceinline__ void spin_until_ge(const unsigned *ctr, unsigned target, SpinCtx &sp, unsigned code) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++it) {
        if (!spin_continue(sp, t0, it, code)) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// the per-device error word + budget of the spin kernels (forward_small.hip); NULL when the allocation failed
unsigned *spin_error_word();
unsigned long long spin_budget_ticks();
// test hook (tgnn_debug_spin_fault): 1 = the launch being queued runs with its last block absent (it returns at once), which
// is what a block that never becomes resident looks like to the others
int spin_take_fault();

}  // namespace tgnn
