// Instruction-rate microbenchmark for gfx950: decides the field-arithmetic design
// (quarter-rate u32 MAD vs f64 FMA vs u24 MAD).  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

#define ITER 512
// 8 independent chains per lane, each ITER long
__global__ void k_mad64(u64* out, u32 a, u32 b){
  u64 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j;
  u32 x=a+threadIdx.x, y=b;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(c[j]) : "v"(x), "v"(y) : "s10","s11");
  }
  u64 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_mullo(u64* out, u32 a, u32 b){
  u32 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  u32 y=b|1;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[j]) : "v"(y));
  }
  u32 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_mulhi(u64* out, u32 a, u32 b){
  u32 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  u32 y=b|1;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c[j]) : "v"(y));
  }
  u32 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_mad24(u64* out, u32 a, u32 b){
  u32 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  u32 y=b|1;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c[j]) : "v"(y));
  }
  u32 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_add32(u64* out, u32 a, u32 b){
  u32 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  u32 y=b|1;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[j]) : "v"(y));
  }
  u32 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_add64(u64* out, u32 a, u32 b){
  u64 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  u64 y=b|1;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[j]) : "v"(y));
  }
  u64 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_shr64(u64* out, u32 a, u32 b){
  u64 c[8]; for(int j=0;j<8;j++) c[j]=~(u64)(threadIdx.x+j+a);
  u32 y=(b&1);
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(c[j]) : "v"(y));
  }
  u64 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_fma64(u64* out, u32 a, u32 b){
  double c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  double x=1.0000001, y=b*1e-9;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c[j]) : "v"(x), "v"(y));
  }
  double s=0; for(int j=0;j<8;j++) s+=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u64)s;
}
__global__ void k_fma32(u64* out, u32 a, u32 b){
  float c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  float x=1.0000001f, y=b*1e-9f;
  for(int i=0;i<ITER;i++){
    #pragma unroll
    for(int j=0;j<8;j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[j]) : "v"(x), "v"(y));
  }
  float s=0; for(int j=0;j<8;j++) s+=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=(u64)s;
}
// carry chain: add_co + addc back to back (hazard cost)
__global__ void k_carry(u64* out, u32 a, u32 b){
  u32 c[8]; for(int j=0;j<8;j++) c[j]=threadIdx.x+j+a;
  u32 y=b|1;
  for(int i=0;i<ITER;i++){
    asm volatile(
      "v_add_co_u32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %8, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %8, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %8, vcc\n\t"
      : "+v"(c[0]),"+v"(c[1]),"+v"(c[2]),"+v"(c[3]),"+v"(c[4]),"+v"(c[5]),"+v"(c[6]),"+v"(c[7]) : "v"(y) : "vcc");
  }
  u32 s=0; for(int j=0;j<8;j++) s^=c[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

// ---- 29-bit-limb Montgomery multiplication (plain C) ----
struct Fe { u32 l[9]; };
#define MASK 0x1fffffffu
__device__ __forceinline__ u64 mad(u32 a, u32 b, u64 c){ return (u64)a*b + c; }
__device__ __forceinline__ Fe mont_mul(const Fe& a, const Fe& b){
  constexpr u32 PL[9] = {0x187cfd47,0x10460b6,0x1c72a34f,0x2d522d0,0x1585d978,0x2db40c0,0xa6e141,0xe5c2634,0x30644e};
  constexpr u32 PINV = 0x1f5ba9b9u;
  u64 acc=0; u32 m[9]; Fe r;
  #pragma unroll
  for(int k=0;k<9;k++){
    #pragma unroll
    for(int i=0;i<=k;i++) acc = mad(a.l[i], b.l[k-i], acc);
    #pragma unroll
    for(int i=0;i<k;i++) acc = mad(m[i], PL[k-i], acc);
    m[k] = ((u32)acc * PINV) & MASK;
    acc = mad(m[k], PL[0], acc);
    acc >>= 29;
  }
  #pragma unroll
  for(int k=9;k<17;k++){
    #pragma unroll
    for(int i=k-8;i<9;i++) acc = mad(a.l[i], b.l[k-i], acc);
    #pragma unroll
    for(int i=k-8;i<9;i++) acc = mad(m[i], PL[k-i], acc);
    r.l[k-9] = (u32)acc & MASK; acc >>= 29;
  }
  r.l[8]=(u32)acc;
  return r;
}
template<int NCH>
__global__ void k_mont(u64* out, u32 a, u32 b, int iters){
  Fe x[NCH], y;
  for(int c=0;c<NCH;c++) for(int j=0;j<9;j++) x[c].l[j]=(threadIdx.x*2654435761u+j*40503u+a+c)&MASK;
  for(int j=0;j<9;j++) y.l[j]=(threadIdx.x*40503u+j*2654435761u+b)&MASK;
  for(int t=0;t<iters;t++){
    #pragma unroll
    for(int c=0;c<NCH;c++) x[c]=mont_mul(x[c],y);
  }
  u32 s=0; for(int c=0;c<NCH;c++) for(int j=0;j<9;j++) s^=x[c].l[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

template<class F> double timeit(F f){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for(int r=0;r<5;r++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/5.0;
}
int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  int CU=prop.multiProcessorCount;
  u64* out; CK(hipMalloc(&out, sizeof(u64)*CU*16*256));
  int wavesPerSimd[] = {1,2,4,8};
  #define RUN(name, kern, opsPerLaneIter) for(int w: wavesPerSimd){ int blocks=CU*w; /* 256 thr = 4 waves = 1 per SIMD */ \
      double ms=timeit([&]{ hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u, 678u); }); \
      double ops=(double)blocks*256*ITER*opsPerLaneIter; \
      double cyc = ms*1e-3*2.4e9 * (CU*4.0) / (ops/64.0); \
      printf("%-10s waves/SIMD=%d  %.3f ms  %.2f Gop/s  ~%.2f cyc/wave-instr @2.4GHz\n", name, w, ms, ops/ms*1e-6, cyc); }
  RUN("mad_u64", k_mad64, 8)
  RUN("mul_lo", k_mullo, 8)
  RUN("mul_hi", k_mulhi, 8)
  RUN("mad_u24", k_mad24, 8)
  RUN("add_u32", k_add32, 8)
  RUN("add_u64", k_add64, 8)
  RUN("shr_b64", k_shr64, 8)
  RUN("fma_f64", k_fma64, 8)
  RUN("fma_f32", k_fma32, 8)
  RUN("carry8", k_carry, 8)
  for(int w: wavesPerSimd){ int blocks=CU*w; int iters=256;
    double ms=timeit([&]{ hipLaunchKernelGGL(k_mont<1>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 678u, iters); });
    double muls=(double)blocks*256*iters; double cyc = ms*1e-3*2.4e9*(CU*4.0)/(muls/64.0);
    printf("montmul x1 waves/SIMD=%d %.3f ms  %.2f Gmul/s  ~%.0f cyc/wave-mul\n", w, ms, muls/ms*1e-6, cyc);
    ms=timeit([&]{ hipLaunchKernelGGL(k_mont<2>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 678u, iters); });
    muls=(double)blocks*256*iters*2; cyc = ms*1e-3*2.4e9*(CU*4.0)/(muls/64.0);
    printf("montmul x2 waves/SIMD=%d %.3f ms  %.2f Gmul/s  ~%.0f cyc/wave-mul\n", w, ms, muls/ms*1e-6, cyc);
  }
  return 0;
}
