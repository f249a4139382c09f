#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void stream_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  float *d; hipMalloc(&d, 1 << 20);
  float4 *a, *b; size_t n4 = (size_t)(12.8e6 / 16 * 4); hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 256, 2048}) {
    for (int i = 0; i < 10; ++i) tiny<<<grid, 256>>>(d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 1000; ++i) tiny<<<grid, 256>>>(d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("tiny kernel grid=%4d: %.2f us per dependent launch\n", grid, ms);
  }
  // 51 MB copy kernel back-to-back (merge-like traffic)
  for (int i = 0; i < 5; ++i) stream_k<<<2048, 256>>>(a, b, n4);
  hipDeviceSynchronize(); hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) stream_k<<<2048, 256>>>(a, b, n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("copy %.1f MB r+w: %.2f us per launch -> %.2f TB/s\n", n4 * 32 / 1e6, ms / 200 * 1e3, n4 * 32 / (ms / 200 * 1e-3) / 1e12);
  return 0;
}
