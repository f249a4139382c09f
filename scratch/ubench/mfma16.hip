// Issue cost of the 16-bit matrix instructions of gfx950 per SIMD: independent accumulators, one wave per SIMD and two.
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int MODE> __global__ __launch_bounds__(512) void k(float *out, int iters) {
    f16x8 a, b; bf16x8 ab, bb;
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    f16x4 a4, b4; for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)(threadIdx.x * 0.001f); b4[i] = (_Float16)1.0f; }
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)1.0f; ab[i] = (__bf16)(threadIdx.x * 0.001f); bb[i] = (__bf16)1.0f; }
    f32x16 c0, c1, c2, c3; f32x4 d0 = {0,0,0,0}, d1 = d0, d2 = d0, d3 = d0;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                         c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0); }
        if (MODE == 1) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
                         d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0); }
        if (MODE == 2) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
                         c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0); }
        if (MODE == 3) { d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d1, 0, 0, 0);
                         d2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d3, 0, 0, 0); }
        if (MODE == 4) { d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0);
                         d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); }
        if (MODE == 5) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
                         d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    s += d0[0] + d1[0] + d2[0] + d3[0];
    if (s == 123.456f) out[0] = s;
}
template <int MODE> void run(const char *name, int threads, double flop) {
    float *d; hipMalloc(&d, 4); const int iters = 20000, blocks = 256;
    k<MODE><<<blocks, threads>>>(d, 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64 / 4.0, n = (double)iters * 4 * waves_per_simd;   // MFMAs per SIMD
    printf("%-22s %d waves/SIMD: %.1f ns per instruction per SIMD (%.1f cycles at 2.4 GHz), %.0f TFLOP/s\n", name, (int)waves_per_simd, ms * 1e6 / n,
           ms * 1e6 / n * 2.4, n * 1024 * flop / ms / 1e9);
}
int main() {
    run<0>("32x32x16 f16", 256, 32768); run<0>("32x32x16 f16", 512, 32768);
    run<1>("16x16x32 f16", 256, 16384); run<1>("16x16x32 f16", 512, 16384);
    run<2>("32x32x16 bf16", 256, 32768); run<2>("32x32x16 bf16", 512, 32768);
    run<3>("16x16x16 f16", 256, 8192); run<3>("16x16x16 f16", 512, 8192);
    run<4>("16x16x16 f16 dependent", 256, 8192); run<4>("16x16x16 f16 dependent", 1024, 8192);
    run<5>("16x16x32 f16 dependent", 256, 16384); run<5>("16x16x32 f16 dependent", 1024, 16384);
    return 0;
}
