#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  if (MODE == 0) { // 32x32x2, 4 independent accumulators
    f32x16 c0={0},c1={0},c2={0},c3={0};
    for (int i=0;i<iters;++i){
      c0=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c0,0,0,0); c1=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c1,0,0,0);
      c2=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c2,0,0,0); c3=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c3,0,0,0);
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=c0[0]+c1[1]+c2[2]+c3[3];
  } else if (MODE == 1) { // 16x16x4, 2 accumulators (like nnconv)
    f32x4 c0={0},c1={0};
    for (int i=0;i<iters;++i){
      c0=__builtin_amdgcn_mfma_f32_16x16x4f32(a,b,c0,0,0,0); c1=__builtin_amdgcn_mfma_f32_16x16x4f32(a,b,c1,0,0,0);
      c0=__builtin_amdgcn_mfma_f32_16x16x4f32(a,b,c0,0,0,0); c1=__builtin_amdgcn_mfma_f32_16x16x4f32(a,b,c1,0,0,0);
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=c0[0]+c1[1];
  } else { // 32x32x2, 1 accumulator chain (like gin layer 1)
    f32x16 c0={0};
    for (int i=0;i<iters;++i){
      c0=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c0,0,0,0); c0=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c0,0,0,0);
      c0=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c0,0,0,0); c0=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c0,0,0,0);
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=c0[0];
  }
}
template <int MODE> void run(const char* name, int blocks, double flop_per_mfma) {
  float* d; hipMalloc(&d, blocks*256*4);
  int iters = 20000; hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks,256>>>(d,100,1.f,1.f);
  hipEventRecord(a); k<MODE><<<blocks,256>>>(d,iters,1.0001f,0.5f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double n_mfma = (double)iters*4*blocks*4; // per wave 4 per iter; waves = blocks*4
  printf("%-36s blocks=%5d  %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, blocks, ms, n_mfma*flop_per_mfma/ms/1e9, ms*1e6/((double)iters*4*(blocks*4/1024.0)));
}
int main(){
  run<0>("32x32x2 4 acc, 1 wave/SIMD",256, 2.0*32*32*2); run<0>("32x32x2 4 acc, 2 waves/SIMD",512, 2.0*32*32*2);
  run<1>("16x16x4 2 acc, 1 wave/SIMD",256, 2.0*16*16*4); run<1>("16x16x4 2 acc, 2 waves/SIMD",512, 2.0*16*16*4);
  run<2>("32x32x2 1 acc chain, 1 wave/SIMD",256, 2.0*32*32*2); run<2>("32x32x2 1 acc chain, 2 waves/SIMD",512, 2.0*32*32*2);
  return 0;
}
