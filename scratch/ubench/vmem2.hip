// VMEM cost model, part 2: what does a gather instruction cost as a function of its valid lanes / lines?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
__device__ v4f bufload(rsrc_t rsrc, uint32_t off) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0)); }
template <int MODE>
__global__ void k(const float* __restrict__ in, float* out, int iters, int nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fj = lane & 15, fq = lane >> 4;
  float4 acc4 = make_float4(0,0,0,0); float acc = 0;
  unsigned r = (blockIdx.x * 977u + wave * 131u) % nrows;
  const rsrc_t rsrc = make_rsrc(in, (uint32_t)nrows * 128u);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      r = (r * 1664525u + 1013904223u) % nrows;
      const unsigned row = (r + fj * 37) % nrows;
      if (MODE == 0) { const float4 v = *(const float4*)(in + (size_t)row * 32 + fq * 8); acc4.x += v.x; acc4.y += v.w; }                // 16 rows valid
      if (MODE == 1) { const unsigned rr = fj < 2 ? row : 0; const float4 v = *(const float4*)(in + (size_t)rr * 32 + fq * 8); acc4.x += v.x; acc4.y += v.w; }  // 2 valid, 14 dummy row
      if (MODE == 2) { const uint32_t off = fj < 2 ? row * 128u + fq * 32 : 0xffffffffu; const v4f v = bufload(rsrc, off); acc4.x += v.x; acc4.y += v.w; }    // 2 valid, 14 OOB
      if (MODE == 3) { const uint32_t off = row * 128u + fq * 32; const v4f v = bufload(rsrc, off); acc4.x += v.x; acc4.y += v.w; }       // 16 valid, buffer
      if (MODE == 4) { const unsigned rr = (r + fj) % nrows; const float4 v = *(const float4*)(in + (size_t)rr * 32 + fq * 8); acc4.x += v.x; acc4.y += v.w; }   // 16 adjacent rows (2 KB contiguous, half lines)
      if (MODE == 5) { acc += in[(size_t)r * 32 + lane]; }                                                       // dword, 256 B contiguous
      if (MODE == 6) { const uint32_t off = fj < 8 ? row * 128u + fq * 32 : 0xffffffffu; const v4f v = bufload(rsrc, off); acc4.x += v.x; acc4.y += v.w; }    // 8 valid
      if (MODE == 7) { const uint32_t off = 0xffffffffu; const v4f v = bufload(rsrc, off + (r & 0)); acc4.x += v.x; acc4.y += v.w; }       // all OOB
      if (MODE == 8) { const float4 v = *(const float4*)(in + (size_t)((r + (lane >> 3) * 37) % nrows) * 32 + (lane & 7) * 4); acc4.x += v.x; acc4.y += v.w; } // 8 rows x 128 B
      if (MODE == 9) { const float2 v = *(const float2*)(in + (size_t)r * 32 + lane * 2); acc4.x += v.x; acc4.y += v.y; }               // dwordx2 512B contiguous
      if (MODE == 10) { const float4 v = *(const float4*)(in + (size_t)(r & ~7u) * 32 + lane * 4); acc4.x += v.x; acc4.y += v.w; }        // dwordx4 1KB contiguous
      if (MODE == 11) { const unsigned rr = (r + (lane >> 2) * 37) % nrows; const float4 v = *(const float4*)(in + (size_t)rr * 32 + (lane & 3) * 4 + (u & 1) * 16); acc4.x += v.x; acc4.y += v.w; } // 16 rows x 64B contiguous per row
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc4.x + acc4.y + acc4.z + acc4.w;
}
template <int MODE> void run(const char* name, int waves_per_cu, int nrows) {
  float *in, *out; hipMalloc(&in, (size_t)nrows * 128 + 4096); hipMemset(in, 0, (size_t)nrows * 128 + 4096); hipMalloc(&out, 256 * 1024 * 4);
  int iters = 2000; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, waves_per_cu * 64>>>(in, out, 10, nrows);
  hipEventRecord(a); k<MODE><<<256, waves_per_cu * 64>>>(in, out, iters, nrows); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_cu = (double)iters * 8 * waves_per_cu;
  printf("%-52s rows=%6d w/CU=%2d %.3f ms %6.1f ns/instr/CU (%4.0f cyc @2.4GHz)\n", name, nrows, waves_per_cu, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
  hipFree(in); hipFree(out);
}
int main() {
  for (int nrows : {8192, 100000}) {
    run<0>("gather 16 rows x 4 x 16B, all valid", 16, nrows);
    run<3>("  same, buffer_load", 16, nrows);
    run<1>("gather 2 valid + 14 dummy-row lanes", 16, nrows);
    run<2>("gather 2 valid + 14 OOB (buffer)", 16, nrows);
    run<6>("gather 8 valid + 8 OOB (buffer)", 16, nrows);
    run<7>("gather all OOB (buffer)", 16, nrows);
    run<4>("gather 16 adjacent rows", 16, nrows);
    run<8>("dwordx4 8 rows x 128B", 16, nrows);
    run<11>("dwordx4 16 rows x 64B (4 lanes/row)", 16, nrows);
    run<5>("dword 256B contiguous", 16, nrows);
    run<9>("dwordx2 512B contiguous", 16, nrows);
    run<10>("dwordx4 1KB contiguous", 16, nrows);
  }
  return 0;
}
