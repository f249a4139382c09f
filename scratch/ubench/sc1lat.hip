// Latency of agent-scope (sc1) and plain loads of data another XCD wrote (a) long ago (previous kernel), (b) just before a
// grid barrier inside the same kernel; 80 blocks, one wave measures a chain of dependent loads.
//   hipcc --offload-arch=gfx950 -O3 -o sc1lat sc1lat.hip && ./sc1lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ void gbar(unsigned *ctr, unsigned &target, unsigned nblk) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    target += nblk;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// buf: [nblk][1024] uints; each block writes its region (value = index of the next element to visit in the NEIGHBOUR block's region)
template <int AUX, bool FRESH>
__global__ __launch_bounds__(256) void k(unsigned *ctr, unsigned *buf, unsigned long long *out, int chain) {
    const unsigned nblk = gridDim.x;
    unsigned target = 0;
    const int b = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)0x80000000u, 0x00020000);
    if (FRESH) {
        for (int i = tid; i < 1024; i += 256)
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)((i * 37 + 11) & 1023), rs, (unsigned)(b * 1024 + i) * 4u, 0, 16);
    }
    gbar(ctr, target, nblk);
    if (tid < 64) {
        const int o = (b + 3) % nblk;                         // block b + 3: another XCD (blocks go round the 8 XCDs)
        unsigned idx = tid * 16;
        const unsigned long long t0 = wall_clock64();
        for (int c = 0; c < chain; ++c) idx = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(o * 1024 + idx) * 4u, 0, AUX);
        const unsigned long long t1 = wall_clock64();
        if (tid == 0) out[b] = t1 - t0;
        if (idx == 0xffffffffu) out[b] = 0;
    }
}

__global__ void fill(unsigned *buf, int nblk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nblk * 1024) buf[i] = (unsigned)(((i & 1023) * 37 + 11) & 1023);
}

template <int AUX, bool FRESH>
static int run(const char *name, int nblk) {
    unsigned *ctr, *buf; unsigned long long *out;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, (size_t)nblk * 4096)); CK(hipMalloc(&out, nblk * 8));
    const int chain = 16;
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(ctr, 0, 4));
        fill<<<nblk * 4, 256>>>(buf, nblk);
        CK(hipDeviceSynchronize());
        k<AUX, FRESH><<<nblk, 256>>>(ctr, buf, out, chain);
        CK(hipDeviceSynchronize());
        unsigned long long h[256];
        CK(hipMemcpy(h, out, nblk * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (int i = 0; i < nblk; ++i) mean += (double)h[i];
        mean = mean / nblk / chain * 10.0;                    // ns per dependent load (100 MHz clock)
        if (mean < best) best = mean;
    }
    printf("%-44s %7.0f ns per dependent load\n", name, best);
    CK(hipFree(ctr)); CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}

int main() {
    run<0, false>("plain load, data of a previous kernel", 80);
    run<16, false>("sc1 load, data of a previous kernel", 80);
    run<16, true>("sc1 load, data sc1-stored before the barrier", 80);
    run<17, true>("sc0 sc1 load, data sc1-stored before the barrier", 80);
    run<0, true>("plain load (stale-prone), fresh data", 80);
    return 0;
}
