// Do VALU instructions of one wave issue in the shadow of another wave's MFMAs on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode bit0: even waves run MFMA; bit1: odd waves run VALU.  Block = 512 threads = 8 waves = 2 per SIMD (waves w and w+4 share a SIMD)
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode_a, int mode_b) {
  const int wave = threadIdx.x >> 6;
  const int role = (wave >> 2) & 1;                 // waves 0-3: role 0, waves 4-7: role 1 (one of each per SIMD)
  const int what = role == 0 ? mode_a : mode_b;     // 0 idle, 1 MFMA, 2 VALU, 3 LDS-ish none
  float r = threadIdx.x * 1e-3f;
  if (what == 1) {
    f32x4 d0 = {0,0,0,0}, d1 = d0, d2 = d0, d3 = d0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, d3, 0, 0, 0);
      }
    }
    r = d0[0] + d1[1] + d2[2] + d3[3];
  } else if (what == 3) {     // bf16 16x16x32: 8 passes? measure
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 a; for (int i = 0; i < 8; ++i) a[i] = (__bf16)r;
    f32x4 d0 = {0,0,0,0}, d1 = d0, d2 = d0, d3 = d0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, d3, 0, 0, 0);
      }
    }
    r = d0[0] + d1[1] + d2[2] + d3[3];
  } else if (what == 2) {
    float a0 = r, a1 = r + 1, a2 = r + 2, a3 = r + 3, a4 = r + 4, a5 = r + 5, a6 = r + 6, a7 = r + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {   // 128 independent-ish FMAs per iteration = 512 issue cycles, same as 16 MFMAs x 32
        a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
        a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
      }
    }
    r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
float run(int a, int b, int iters) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256, 512>>>(out, 10, a, b);
  hipEventRecord(e0); k<<<256, 512>>>(out, iters, a, b); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out); return ms;
}
int main() {
  const int iters = 20000;
  const char* names[] = {"idle", "MFMA", "VALU", "BF16"};
  int combos[][2] = {{1,0},{2,0},{1,1},{2,2},{1,2},{3,0},{3,3},{3,2}};
  for (auto& c : combos) {
    float ms = run(c[0], c[1], iters);
    printf("wave A: %-5s wave B: %-5s  %.3f ms  (%.0f cycles per iteration @2.4GHz; 512 = one role at full rate)\n", names[c[0]], names[c[1]], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
