// Issue-rate facts the round-2 NNConv kernel rests on (gfx950):
//   A  plain v_add_f32 vs v_pk_add_f32 throughput per SIMD at 1 / 2 / 4 waves per SIMD
//   B  one wave: 12 dependent-pair bf16 16x16x32 MFMAs with n independent VALU adds interleaved -- do they hide?
//   C  the same with the VALU work in a DIFFERENT wave of the same SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void ka(float *out, const float *in, int iters) {
    float a[16];
    f32x2 p[16];
    const float x = in[threadIdx.x & 63];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = x + i; p[i] = f32x2{x + i, x - i}; }
    const f32x2 px = {x, x * 2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) a[i] += x;
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px));
                if (MODE == 2) a[i] = fmaf(a[i], x, x);
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(px));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// B: NV adds per MFMA, same wave
template <int NV>
__global__ void kb(float *out, const float *in, int iters) {
    const float x = in[threadIdx.x & 63];
    bf16x8 w;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (__bf16)(x + i);
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0;
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, w, d0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[(u * NV + i) & 15] += x;
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, w, d1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) a[(u * NV + i + 8) & 15] += x;
        }
    }
    float s = d0[0] + d1[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// C: waves 0-3 MFMA (12 per iter), waves 4-7 VALU (NV*12 adds per iter) -- one of each per SIMD
template <int NV>
__global__ __launch_bounds__(512) void kc(float *out, const float *in, int iters, int mode) {
    const int role = (threadIdx.x >> 8) & 1;
    const float x = in[threadIdx.x & 63];
    float s = 0;
    if (role == 0 && (mode & 1)) {
        bf16x8 w;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = (__bf16)(x + i);
        f32x4 d0 = {0, 0, 0, 0}, d1 = d0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, w, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, w, d1, 0, 0, 0);
            }
        }
        s = d0[0] + d1[1];
    } else if (role == 1 && (mode & 2)) {
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 12 * NV; ++u) a[u & 15] += x;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float timeit(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(20000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / 20000.f;      // cycles per iteration @ 2.4 GHz
}

int main() {
    float *out, *in;
    hipMalloc(&out, 1024 * 1024 * 4);
    hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    const char *na[] = {"v_add_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_fma_f32"};
    for (int wps = 1; wps <= 4; wps *= 2) {
        float c[4];
        c[0] = timeit([&](int it) { ka<0><<<256, 256 * wps>>>(out, in, it); });
        c[1] = timeit([&](int it) { ka<1><<<256, 256 * wps>>>(out, in, it); });
        c[2] = timeit([&](int it) { ka<2><<<256, 256 * wps>>>(out, in, it); });
        c[3] = timeit([&](int it) { ka<3><<<256, 256 * wps>>>(out, in, it); });
        for (int m = 0; m < 4; ++m)
            printf("A %-13s waves/SIMD %d: %7.1f cycles / 64 instr / wave -> %.2f cycles per instr per SIMD\n", na[m], wps, c[m],
                   c[m] / 64.f / wps);
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        float c0 = timeit([&](int it) { kb<0><<<256, 256 * wps>>>(out, in, it); });
        float c2 = timeit([&](int it) { kb<2><<<256, 256 * wps>>>(out, in, it); });
        float c4 = timeit([&](int it) { kb<4><<<256, 256 * wps>>>(out, in, it); });
        float c8 = timeit([&](int it) { kb<8><<<256, 256 * wps>>>(out, in, it); });
        printf("B waves/SIMD %d: 12 MFMA + {0, 24, 48, 96} adds in the SAME wave: %.0f %.0f %.0f %.0f cycles per iteration per wave\n",
               wps, c0, c2, c4, c8);
    }
    {
        float m = timeit([&](int it) { kc<4><<<256, 512>>>(out, in, it, 1); });
        float v = timeit([&](int it) { kc<4><<<256, 512>>>(out, in, it, 2); });
        float b = timeit([&](int it) { kc<4><<<256, 512>>>(out, in, it, 3); });
        printf("C 12 MFMA wave alone %.0f | 48-add wave alone %.0f | both on one SIMD %.0f cycles per iteration\n", m, v, b);
        float v8 = timeit([&](int it) { kc<8><<<256, 512>>>(out, in, it, 2); });
        float b8 = timeit([&](int it) { kc<8><<<256, 512>>>(out, in, it, 3); });
        printf("C 96-add wave alone %.0f | with the MFMA wave %.0f\n", v8, b8);
    }
    return 0;
}
