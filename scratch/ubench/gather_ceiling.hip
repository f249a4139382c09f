// Row-gather ceiling of one CU's vector-memory path: best sustained rate of gathering 128-byte rows of an
// L2 / Infinity-Cache resident 12.8 MB array (100 000 rows), band-local random rows (+- 2 048 rows round a slowly moving base: the
// index distance of the benchmark's layouts), by access SHAPE:
//   A  production column NNConv (r1-r4): lane (row l % 16, quarter l / 16) loads 16 B at quarter * 32 (+ 16 in the second
//      instruction): 16 rows per instruction, every instruction touches all 16 lines
//   B  the same lane map, 16 B at quarter * 16 (+ 64): 16 rows x one contiguous 64-byte half-line per instruction
//   C  whole rows into registers: lane l loads chunk l % 8 of row l / 8: 8 rows x 128 B per instruction
//   D  whole rows by LDS-DMA (global_load_lds_dwordx4): same lane map as C, data lands in LDS without a register
//   E  D through a buffer descriptor (raw_ptr_buffer_load_lds)
// Prints ns per wave-instruction per CU, bytes / clock / CU at the nominal 2.4 GHz and chip-wide GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int MODE>
__global__ void k(const float *__restrict__ in, float *out, int iters, int nrows) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    v4f acc = {0, 0, 0, 0};
    const rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)((uint32_t)nrows * 128u), 0x00020000);
    // the row of this lane's group: A/B: group = l % 16; C/D/E: group = l / 8
    const int grp = (MODE <= 1) ? (lane & 15) : (lane >> 3);
    unsigned r = (blockIdx.x * 977u + wave * 131u + grp * 7919u) * 2654435761u + 12345u;
    unsigned base = ((blockIdx.x * n_waves + wave) * 389u) % (unsigned)(nrows - 4096 - 16 * 9);
    float *my = lds + wave * 2048;   // 8 KB ring per wave (D/E)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r = r * 1664525u + 1013904223u;
            const unsigned row = base + ((r >> 10) & 4095u);
            if (MODE == 0) {
                const uint32_t off = row * 128u + (uint32_t)(lane >> 4) * 32u;
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0));
            }
            if (MODE == 1) {
                const uint32_t off = row * 128u + (uint32_t)(lane >> 4) * 16u;
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 64u, 0, 0));
            }
            if (MODE == 2) {
                const uint32_t off = row * 128u + (uint32_t)(lane & 7) * 16u;
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            }
            if (MODE == 3) {
                const float *p = in + (size_t)row * 32 + (lane & 7) * 4;
                __builtin_amdgcn_global_load_lds(p, (lds_void_t *)(my + (u & 7) * 256), 16, 0, 0);
            }
            if (MODE == 4) {
                const uint32_t off = row * 128u + (uint32_t)(lane & 7) * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(my + (u & 7) * 256), 16, off, 0, 0, 0);
            }
        }
        base += 16;
        if (base >= (unsigned)(nrows - 4096 - 16)) base = 0;
    }
    if (MODE >= 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc[0] += my[lane];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(const char *name, int waves_per_cu, int nrows, const float *in, float *out) {
    const int iters = 1500;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const size_t lds_bytes = (size_t)waves_per_cu * 8192;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<MODE><<<256, waves_per_cu * 64, lds_bytes>>>(in, out, 10, nrows);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        k<MODE><<<256, waves_per_cu * 64, lds_bytes>>>(in, out, iters, nrows);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const int instr_per_col = MODE <= 1 ? 2 : 1;
    const double rows_per_iter8 = MODE <= 1 ? 16.0 * 8 : 8.0 * 8;          // rows of 128 B moved per wave per 8 steps
    const double instr_per_cu = (double)iters * 8 * instr_per_col * waves_per_cu;
    const double bytes_per_cu = (double)iters * rows_per_iter8 * 128.0 * waves_per_cu;
    const double ns = best * 1e6;
    printf("%-64s %2d waves/CU: %6.1f ns/instr/CU  %6.2f ns/row/CU  %5.1f B/clk/CU @2.4GHz  %7.0f GB/s chip (%s)\n", name, waves_per_cu,
           ns / instr_per_cu, ns / (bytes_per_cu / 128.0), bytes_per_cu / (ns * 2.4), bytes_per_cu * 256 / ns,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    const int nrows = 100000;
    float *in, *out;
    hipMalloc(&in, (size_t)nrows * 128 + 4096);
    hipMemset(in, 0, (size_t)nrows * 128 + 4096);
    hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {4, 8, 16}) {
        run<0>("A production: 16 rows x (4 x 16 B at stride 32), 2 instr/column", w, nrows, in, out);
        run<1>("B half-lines: 16 rows x 64 B contiguous, 2 instr/column", w, nrows, in, out);
        run<2>("C whole rows -> registers: 8 rows x 128 B", w, nrows, in, out);
        run<3>("D whole rows -> LDS (global_load_lds b128): 8 rows x 128 B", w, nrows, in, out);
        run<4>("E whole rows -> LDS (buffer_load .. lds b128): 8 rows x 128 B", w, nrows, in, out);
    }
    return 0;
}
