// VALU issue-rate microbenchmark: one wave per SIMD (256 threads/block, 256 blocks) and 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int K> __device__ __forceinline__ float bc(float x){ return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150+K, 0xf, 0xf, true)); }

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0+1, x2 = x0+2, x3 = x0+3, x4=x0+4, x5=x0+5, x6=x0+6, x7=x0+7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent v_fma chains, 8 instrs per trip x 8 unroll
#pragma unroll
      for (int u = 0; u < 8; ++u) { x0=fmaf(x0,a,b); x1=fmaf(x1,a,b); x2=fmaf(x2,a,b); x3=fmaf(x3,a,b); x4=fmaf(x4,a,b); x5=fmaf(x5,a,b); x6=fmaf(x6,a,b); x7=fmaf(x7,a,b); }
    } else if (MODE == 1) { // dpp mov + 2 fma
#pragma unroll
      for (int u = 0; u < 8; ++u) { float t0=bc<1>(x0), t1=bc<2>(x1), t2=bc<3>(x2), t3=bc<4>(x3); x4=fmaf(t0,a,x4); x5=fmaf(t1,a,x5); x6=fmaf(t2,a,x6); x7=fmaf(t3,a,x7); x0+=b; x1+=b; x2+=b; x3+=b; }
    } else if (MODE == 2) { // v_fmac with vgpr operands (acc += x*y)
#pragma unroll
      for (int u = 0; u < 8; ++u) { x0=fmaf(x4,x5,x0); x1=fmaf(x5,x6,x1); x2=fmaf(x6,x7,x2); x3=fmaf(x7,x4,x3); x4=fmaf(x0,x1,x4); x5=fmaf(x1,x2,x5); x6=fmaf(x2,x3,x6); x7=fmaf(x3,x0,x7);}
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x] = x0+x1+x2+x3+x4+x5+x6+x7;
}
template <int MODE> void run(const char* name, int blocks, int threads, int instr_per_trip) {
  float* d; hipMalloc(&d, blocks*threads*4);
  int iters = 20000;
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks,threads>>>(d, 100, 1.0001f, 0.5f);
  hipEventRecord(a); k<MODE><<<blocks,threads>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double winstr = (double)iters*instr_per_trip; // per wave
  double waves_per_simd = (double)blocks*threads/64/1024.0;
  printf("%-28s blocks=%d thr=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, blocks, threads, ms, ms*1e6/(winstr*waves_per_simd), ms*1e6/(winstr*waves_per_simd)*2.4);
  hipFree(d);
}
int main(){
  run<0>("v_fma (imm/sgpr operands)", 256, 256, 64); run<0>("v_fma 4 waves/SIMD", 1024, 256, 64);
  run<2>("v_fma vgpr operands", 256, 256, 64); run<2>("v_fma vgpr 4w/SIMD", 1024, 256, 64);
  run<1>("dpp mov+fma+add (12/trip)", 256, 256, 96); run<1>("dpp 4w/SIMD", 1024, 256, 96);
  return 0;
}
