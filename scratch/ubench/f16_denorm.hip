// Do the fp16 matrix instructions of gfx950 keep subnormal inputs?  A = 2^-20 (fp16 subnormal), B = 2^10, K = 32:
// every output is 32 * 2^-10 = 2^-5 when they do, 0 when inputs are flushed.  Also the convert: (half)(2^-20f).
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(float tiny, float big, float *out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)tiny; b[i] = (_Float16)big; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)a[0]; }
}
int main() {
    float *d; hipMalloc(&d, 16);
    k<<<1, 64>>>(9.5367431640625e-07f, 1024.f, d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("16x16x32: %g (kept: %g)  32x32x16: %g (kept: %g)  convert of 2^-20: %g\n", h[0], 32 * 9.5367431640625e-07 * 1024, h[1],
           16 * 9.5367431640625e-07 * 1024, h[2]);
    return 0;
}
