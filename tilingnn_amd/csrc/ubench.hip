// Measurement helper of bench.py (roofline.gather_bound): the best sustained rate at which one launch can GATHER 128-byte rows
// of an L2 / Infinity-Cache resident table (band-local random rows: +- 2 048 rows round a slowly moving base, the index distance
// of the benchmark's layouts) -- the CU's vector-memory path, not HBM, is what the column NNConv and the GIN neighbourhood sum
// run against (profiles/r05_nnconv_study.txt; the stand-alone form with more shapes: scratch/ubench/gather_ceiling.hip).
//   shape 0  the column NNConv's: lane (row l % 16, quarter l / 16) loads 2 x 16 B of its row: 16 rows per pair of instructions
//   shape 1  whole rows: lane l loads piece l % 8 of row l / 8: 8 rows per instruction (the GIN aggregate's shape)
#include "tgnn_common.h"

namespace tgnn {

using v4f = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ void row_gather_kernel(const float *__restrict__ table, float *__restrict__ sink, int iters, int nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    v4f acc = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, (int)((uint32_t)nrows * 128u), 0x00020000);
    const int grp = SHAPE == 0 ? (lane & 15) : (lane >> 3);
    unsigned r = (blockIdx.x * 977u + wave * 131u + grp * 7919u) * 2654435761u + 12345u;
    const unsigned span = (unsigned)(nrows - 4096 - 16 * 9);
    unsigned base = ((blockIdx.x * n_waves + wave) * 389u) % span;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r = r * 1664525u + 1013904223u;
            const unsigned row = base + ((r >> 10) & 4095u);
            if (SHAPE == 0) {
                const uint32_t off = row * 128u + (uint32_t)(lane >> 4) * 32u;
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0));
            } else {
                const uint32_t off = row * 128u + (uint32_t)(lane & 7) * 16u;
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            }
        }
        base += 16;
        if (base >= span) base = 0;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

}  // namespace tgnn

using namespace tgnn;

/* rows of 128 bytes gathered per second by one launch over every CU (16 waves each), best of `reps` launches: *rows_per_s_out.
 * table: n_rows x 128 bytes (n_rows >= 8192), sink: 256 x 1024 floats of scratch.  Synchronises the stream. */
extern "C" int tgnn_ubench_row_gather(int32_t shape, const float *table, int64_t n_rows, float *sink, int32_t iters, int32_t reps,
                                      double *rows_per_s_out, tgnn_stream_t stream) {
    DeviceGuard guard__(stream);
    TGNN_CHECK_ARG((shape == 0 || shape == 1) && table && sink && rows_per_s_out && n_rows >= 8192 && n_rows < (1 << 24) &&
                       iters >= 1 && reps >= 1, "arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = device_cus() > 256 ? 256 : device_cus();
    hipEvent_t a, b;
    TGNN_CHECK_HIP(hipEventCreate(&a));
    TGNN_CHECK_HIP(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep <= reps; ++rep) {                  // (rep 0: warm-up)
        (void)hipEventRecord(a, s);
        if (shape == 0) row_gather_kernel<0><<<blocks, 1024, 0, s>>>(table, sink, rep == 0 ? 10 : iters, (int)n_rows);
        else row_gather_kernel<1><<<blocks, 1024, 0, s>>>(table, sink, rep == 0 ? 10 : iters, (int)n_rows);
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    TGNN_CHECK_LAUNCH();
    const double rows = (double)iters * 8.0 * (shape == 0 ? 16.0 : 8.0) * 16.0 * blocks;   // rows per launch
    *rows_per_s_out = rows / ((double)best * 1e-3);
    return TGNN_OK;
}
