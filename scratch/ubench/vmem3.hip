// VMEM cost model, part 3: does the per-instruction floor of a sparse gather go away when the empty lanes are masked
// by EXEC instead of carrying out-of-range offsets?  (16 waves per CU, L2-resident table, buffer_load_dwordx4)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
__device__ v4f bufload(rsrc_t rsrc, uint32_t off) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0)); }
template <int MODE, int NV>
__global__ void k(const float* __restrict__ in, float* out, int iters, int nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fj = lane & 15, fq = lane >> 4;
  v4f acc = {0, 0, 0, 0};
  unsigned r = (blockIdx.x * 977u + wave * 131u) % nrows;
  const rsrc_t rsrc = make_rsrc(in, (uint32_t)nrows * 128u);
  const rsrc_t nullr = make_rsrc(in, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      r = (r * 1664525u + 1013904223u) % nrows;
      const unsigned row = (r + fj * 37) % nrows;
      if (MODE == 0) { const uint32_t off = fj < NV ? row * 128u + fq * 32 : 0x80000000u; acc += bufload(rsrc, off); }            // OOB offsets
      if (MODE == 1) { if (fj < NV) acc += bufload(rsrc, row * 128u + fq * 32); }                                                   // EXEC-masked, scattered lanes
      if (MODE == 2) { const unsigned rw = (r + (lane >> 2) * 37) % nrows; if (lane < 4 * NV) acc += bufload(rsrc, rw * 128u + (lane & 3) * 32); }   // EXEC-masked, packed lanes
      if (MODE == 3) { acc += bufload(nullr, row * 128u + fq * 32); }                                                               // null descriptor
      if (MODE == 4) { const unsigned rw = (r + (lane >> 3) * 37) % nrows; if (lane < 8 * NV) acc += bufload(rsrc, rw * 128u + (lane & 7) * 16); }   // packed, 8 lanes x 16 B per row (whole 128-B rows)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE, int NV> void run(const char* name, int nrows) {
  const int waves_per_cu = 16;
  float *in, *out; hipMalloc(&in, (size_t)nrows * 128 + 4096); hipMemset(in, 0, (size_t)nrows * 128 + 4096); hipMalloc(&out, 256 * 1024 * 4);
  int iters = 2000; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, NV><<<256, waves_per_cu * 64>>>(in, out, 10, nrows);
  hipEventRecord(a); k<MODE, NV><<<256, waves_per_cu * 64>>>(in, out, iters, nrows); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_cu = (double)iters * 8 * waves_per_cu;
  printf("%-58s valid rows %2d: %6.1f ns/instr/CU\n", name, NV, ms * 1e6 / instr_per_cu);
  hipFree(in); hipFree(out);
}
int main() {
  const int nrows = 100000;
  run<0, 16>("16 rows x 4 lanes x 16 B, empty lanes = OOB offsets", nrows);
  run<0, 8>("16 rows x 4 lanes x 16 B, empty lanes = OOB offsets", nrows);
  run<0, 3>("16 rows x 4 lanes x 16 B, empty lanes = OOB offsets", nrows);
  run<0, 1>("16 rows x 4 lanes x 16 B, empty lanes = OOB offsets", nrows);
  run<0, 0>("16 rows x 4 lanes x 16 B, empty lanes = OOB offsets", nrows);
  run<1, 8>("  empty lanes masked by EXEC (scattered)", nrows);
  run<1, 3>("  empty lanes masked by EXEC (scattered)", nrows);
  run<1, 1>("  empty lanes masked by EXEC (scattered)", nrows);
  run<2, 16>("  valid rows packed into the low lanes, rest masked", nrows);
  run<2, 8>("  valid rows packed into the low lanes, rest masked", nrows);
  run<2, 3>("  valid rows packed into the low lanes, rest masked", nrows);
  run<2, 1>("  valid rows packed into the low lanes, rest masked", nrows);
  run<3, 16>("null descriptor (every lane out of range)", nrows);
  run<4, 8>("8 lanes x 16 B per row (whole rows), packed, rest masked", nrows);
  run<4, 4>("8 lanes x 16 B per row (whole rows), packed, rest masked", nrows);
  run<4, 2>("8 lanes x 16 B per row (whole rows), packed, rest masked", nrows);
  run<4, 1>("8 lanes x 16 B per row (whole rows), packed, rest masked", nrows);
  return 0;
}
