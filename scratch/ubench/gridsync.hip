// Cost of a grid-wide barrier inside one persistent kernel on gfx950 (cooperative groups vs a hand-written
// monotonic-counter barrier), with a cross-block data exchange checked every phase.
//   hipcc --offload-arch=gfx950 -O3 -o gridsync gridsync.hip && ./gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
namespace cg = cooperative_groups;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>   // 0: cg grid.sync, 1: counter barrier, thread 0 release/acquire, 2: counter barrier, fences by every thread
__device__ __forceinline__ void gbar(unsigned *ctr, unsigned &target, unsigned nblk) {
    if (MODE == 0) {
        cg::this_grid().sync();
        return;
    }
    if (MODE == 5 || MODE == 6 || MODE == 7) {                            // no cache maintenance: the exchanged data itself is sc1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores are acknowledged
        __syncthreads();
        if (MODE == 5 || MODE == 7) {
            target += nblk;
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
        } else {
            ++target;
            unsigned *flags = ctr + 64;
            if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < nblk)
                while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        if (MODE == 7 && threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1 only
        __syncthreads();
        return;
    }
    if (MODE == 3) {                                         // one flag per block, everybody polls everybody's
        __syncthreads();
        ++target;
        unsigned *flags = ctr + 64;
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < nblk)
            while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        return;
    }
    if (MODE == 4) {                                         // two levels: 8 group counters (block % 8), then one
        __syncthreads();
        ++target;
        if (threadIdx.x == 0) {
            const unsigned g = blockIdx.x & 7, gsize = (nblk - g + 7) / 8;
            unsigned *gc = ctr + 16 * (1 + g);
            const unsigned old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == target * gsize) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned ngroups = nblk < 8 ? nblk : 8;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target * ngroups) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        return;
    }
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    target += nblk;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int MODE, bool CHECK>
__global__ __launch_bounds__(1024) void k(unsigned *ctr, float *buf, int phases, unsigned *errs) {
    const unsigned nblk = gridDim.x;
    unsigned target = 0, bad = 0;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int p = 1; p <= phases; ++p) {
        if (CHECK) {
            if (MODE >= 5) __hip_atomic_store(buf + (size_t)b * 1024 + tid, (float)(p * (b + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else buf[(size_t)b * 1024 + tid] = (float)(p * (b + 1));
        }
        gbar<MODE>(ctr, target, nblk);
        if (CHECK) {
            const int o = (b + 17) % nblk;
            const float v = (MODE == 5 || MODE == 6) ? __hip_atomic_load(buf + (size_t)o * 1024 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : buf[(size_t)o * 1024 + tid];
            bad += v != (float)(p * (o + 1));
            gbar<MODE>(ctr, target, nblk);
        }
    }
    if (CHECK && bad) atomicAdd(errs, bad);
}

template <int MODE, bool CHECK>
static int run(int nblk, int phases) {
    unsigned *ctr, *errs;
    float *buf;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&errs, 4)); CK(hipMalloc(&buf, (size_t)nblk * 1024 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    unsigned herr = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(errs, 0, 4));
        void *args[] = {&ctr, &buf, &phases, &errs};
        CK(hipEventRecord(a, 0));
        CK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k<MODE, CHECK>), dim3(nblk), dim3(1024), args, 0, 0));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        unsigned e; CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); herr += e;
    }
    const int nbar = CHECK ? 2 * phases : phases;
    printf("mode %d check %d blocks %3d: %.2f us per barrier (%d barriers, %.3f ms), errors %u\n", MODE, (int)CHECK, nblk,
           best * 1e3f / nbar, nbar, best, herr);
    CK(hipFree(ctr)); CK(hipFree(errs)); CK(hipFree(buf));
    return 0;
}

int main() {
    for (int nblk : {16, 80, 256}) {
        run<1, false>(nblk, 400); run<5, false>(nblk, 400); run<7, false>(nblk, 400);
        run<1, true>(nblk, 200); run<5, true>(nblk, 200); run<7, true>(nblk, 200);
    }
    return 0;
}
