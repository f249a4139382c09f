// Where does an LDS-DMA land when the destination is above 64 KB?  (M0 width on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const float* src, int off, float* out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    for (int i = threadIdx.x; i < 160 * 256; i += 64) reinterpret_cast<float*>(lds)[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1024, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + off), 16, threadIdx.x * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report the first LDS dword index that is not -1
    int found = -1;
    for (int i = 0; i < 160 * 256; ++i) if (reinterpret_cast<float*>(lds)[i] != -1.f) { found = i; break; }
    if (threadIdx.x == 0) { out[0] = (float)found; out[1] = reinterpret_cast<float*>(lds)[off / 4]; }
}
int main() {
    float *src, *out; hipMalloc(&src, 1024); hipMalloc(&out, 64);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
    hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int off : {0, 61440, 66560, 98304, 132096, 162816}) {
        k<<<1, 64, 160 * 1024>>>(src, off, out);
        float r[2]; hipMemcpy(r, out, 8, hipMemcpyDeviceToHost);
        printf("dest byte %6d: first changed dword at byte %8d, value at dest %.0f  (%s)\n", off, (int)r[0] * 4, r[1], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
