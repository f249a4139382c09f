#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* out, const float* in, int iters, float sa) {
  float a[8], x[8], y[8];
#pragma unroll
  for (int i=0;i<8;++i){ a[i]=0.f; x[i]=in[threadIdx.x+i*64]; y[i]=in[threadIdx.x+i*64+512]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i=0;i<8;++i) {
        if (MODE==0) a[i] = fmaf(x[i], y[i], a[i]);            // 3 vgpr
        if (MODE==1) a[i] = fmaf(x[i], sa, a[i]);              // vgpr * sgpr + vgpr
        if (MODE==2) a[i] = fmaf(x[i], y[(i+u)&7], a[i]);      // 3 vgpr, varying pairs
        if (MODE==3) a[i] = x[i] * y[(i+u)&7] + a[(i+1)&7];    // dst != src2
      }
    }
  }
  float s=0; for (int i=0;i<8;++i) s+=a[i];
  out[blockIdx.x*blockDim.x+threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int blocks) {
  float *d, *in; hipMalloc(&d, blocks*256*4); hipMalloc(&in, 4096*4); hipMemset(in, 0, 4096*4);
  int iters = 20000;
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks,256>>>(d, in, 100, 1.0001f);
  hipEventRecord(a); k<MODE><<<blocks,256>>>(d, in, iters, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double winstr = (double)iters*64; double wps = (double)blocks*256/64/1024.0;
  printf("%-34s waves/SIMD=%.0f  %.3f ms -> %.2f ns/instr/SIMD\n", name, wps, ms, ms*1e6/(winstr*wps));
}
int main(){
  run<0>("fmac a+=x*y (3 vgpr)",256); run<0>("fmac a+=x*y (3 vgpr)",1024);
  run<1>("fmac a+=x*s (sgpr)",256); run<1>("fmac a+=x*s (sgpr)",1024);
  run<2>("fmac a+=x*y[(i+u)&7]",256); run<2>("fmac a+=x*y[(i+u)&7]",1024);
  run<3>("fma a=x*y+a' (dst!=src2)",256); run<3>("fma a=x*y+a'",1024);
  return 0;
}
