// VMEM issue-rate microbenchmark: cost per wave-level load instruction on one CU (L2/L1-resident data).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(const float* __restrict__ in, float* out, int iters, int nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f; float4 acc4 = make_float4(0,0,0,0);
  unsigned r = (blockIdx.x * 977u + wave * 131u) % nrows;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      r = (r * 1664525u + 1013904223u) % nrows;        // wave-uniform pseudo-random row
      if (MODE == 0) acc += in[(size_t)r * 32];                                  // broadcast dword
      if (MODE == 1) acc += in[(size_t)r * 32 + (lane & 15)];                     // 16 distinct dwords, one line
      if (MODE == 2) acc += in[(size_t)((r + (lane & 15) * 37) % nrows) * 32];    // 16 distinct lines, dword
      if (MODE == 3) { const float4 v = *(const float4*)(in + (size_t)((r + (lane & 15) * 37) % nrows) * 32 + (lane >> 4) * 8); acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w; } // nnconv gather shape
      if (MODE == 4) { const float4 v = *(const float4*)(in + (size_t)r * 32 * 8 + lane * 4); acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w; } // coalesced 1 KB
      if (MODE == 5) { const float4 v = *(const float4*)(in + (size_t)((r + (lane >> 3) * 37) % nrows) * 32 + (lane & 7) * 4); acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w; } // 8 rows x 128 B
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc4.x + acc4.y + acc4.z + acc4.w;
}
template <int MODE> void run(const char* name, int waves_per_cu) {
  const int nrows = 65536; // 8 MB: L2/MALL resident
  float *in, *out; hipMalloc(&in, (size_t)nrows * 32 * 4 * 8); hipMemset(in, 0, (size_t)nrows * 32 * 4 * 8); hipMalloc(&out, 256 * 1024 * 4);
  int iters = 2000; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, waves_per_cu * 64>>>(in, out, 10, nrows);
  hipEventRecord(a); k<MODE><<<256, waves_per_cu * 64>>>(in, out, iters, nrows); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_cu = (double)iters * 8 * waves_per_cu;
  printf("%-44s waves/CU=%2d  %.3f ms  %.1f ns per instr per CU (%.0f cycles @2.1GHz)\n", name, waves_per_cu, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
  hipFree(in); hipFree(out);
}
int main() {
  for (int w : {8, 16}) {
    if (w == 8) { run<0>("dword broadcast", 8); run<1>("dword 16 distinct, one line", 8); run<2>("dword 16 distinct lines", 8); run<3>("dwordx4 16 rows x 4 x 16B (gather shape)", 8); run<4>("dwordx4 coalesced 1 KB", 8); run<5>("dwordx4 8 rows x 128B", 8); }
    else { run<0>("dword broadcast", 16); run<3>("dwordx4 16 rows x 4 x 16B (gather shape)", 16); run<5>("dwordx4 8 rows x 128B", 16); }
  }
  return 0;
}
