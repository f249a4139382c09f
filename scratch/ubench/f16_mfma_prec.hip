// Accuracy of v_mfma_f32_16x16x32_f16 against the exact dot products of its fp16 inputs, by operand magnitude: random fp16
// values (11-bit mantissas) scaled by 2^ka / 2^kb; error relative to sum |a b| of each output.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const _Float16 *a, const _Float16 *b, float *out) {   // a [16][32] (row i, k), b [16][32] (col j, k)
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    f16x8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = a[i * 32 + 8 * g + e]; bv[e] = b[i * 32 + 8 * g + e]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + i] = c[r];          // row 4 g + r (of a), column i (of b)
}
int main() {
    _Float16 ha[512], hb[512], *da, *db; float *dout, ho[256];
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 1024);
    srand(1);
    for (int ka : {0, 7, 14}) for (int kb : {0, 7, 14}) for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 512; ++i) {
            float x = (float)rand() / RAND_MAX * 2 - 1, y = (float)rand() / RAND_MAX * 2 - 1;
            if (mode == 1) { x *= std::pow(2.f, -(rand() % 12)); y = std::fabs(y); }      // wide-range a, positive b
            ha[i] = (_Float16)(x * std::pow(2.f, ka)); hb[i] = (_Float16)(y * std::pow(2.f, kb));
        }
        hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, dout);
        hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            double s = 0, sa = 0;
            for (int q = 0; q < 32; ++q) { const double p = (double)(float)ha[r * 32 + q] * (double)(float)hb[c * 32 + q]; s += p; sa += std::fabs(p); }
            worst = std::fmax(worst, std::fabs(ho[r * 16 + c] - s) / sa);
        }
        printf("a * 2^%-2d b * 2^%-2d %s: worst |mfma - exact| / sum|ab| = %.2e\n", ka, kb, mode ? "wide-range a" : "uniform     ", worst);
    }
    return 0;
}
