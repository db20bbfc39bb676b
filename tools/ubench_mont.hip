// Variants of the 29-bit-limb Montgomery multiplication measured at the occupancy of the level-1 kernel
// (3 waves per SIMD): how much of a multiplication is NOT v_mad_u64_u32, and what removing it buys.
//   A  plain C (what fp29.cuh compiles to: hipcc gives every column its own accumulator and then ripples
//      the carries with one 64-bit shift + one 64-bit add per column)
//   B  the carry folded into the first multiply-add of the next column (asm MADs, 64-bit shift)
//   C  as B, the 64-bit shift replaced by v_alignbit_b32 + v_lshrrev_b32 (off the 64-bit pipe)
// Build: hipcc --offload-arch=gfx950 -O3 ubench_mont.hip -o ubench_mont
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <fenv.h>
#include <vector>
#include "fp64_mont.h"
typedef uint32_t u32; typedef uint64_t u64;
#define MASK 0x1fffffffu
struct Fe { u32 l[9]; };
__device__ constexpr u32 PL[9] = {0x187cfd47,0x10460b6,0x1c72a34f,0x2d522d0,0x1585d978,0x2db40c0,0xa6e141,0xe5c2634,0x30644e};
constexpr u32 PINV = 0x1f5ba9b9u;

__device__ __forceinline__ u64 mad(u32 a, u32 b, u64 c){ return (u64)a*b + c; }
__device__ __forceinline__ Fe mul_A(const Fe& a, const Fe& b){
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
// asm multiply-add: acc += a*b, scalar carry-out discarded into vcc
__device__ __forceinline__ void amad(u64& acc, u32 a, u32 b){
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void amads(u64& acc, u32 a, u32 s){   // s: wave-uniform constant in an SGPR
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(s) : "vcc");
}
template<bool ALIGNBIT>
__device__ __forceinline__ u64 shr29(u64 acc){
  if constexpr (!ALIGNBIT) return acc >> 29;
  u32 lo=(u32)acc, hi=(u32)(acc>>32), nlo, nhi;
  asm("v_alignbit_b32 %0, %1, %2, 29" : "=v"(nlo) : "v"(hi), "v"(lo));
  nhi = hi >> 29;
  return ((u64)nhi<<32)|nlo;
}
template<bool ALIGNBIT>
__device__ __forceinline__ Fe mul_BC(const Fe& a, const Fe& b){
  u64 acc=0; u32 m[9]; Fe r;
  #pragma unroll
  for(int k=0;k<9;k++){
    #pragma unroll
    for(int i=0;i<=k;i++) amad(acc, a.l[i], b.l[k-i]);
    #pragma unroll
    for(int i=0;i<k;i++) amads(acc, m[i], PL[k-i]);
    m[k] = ((u32)acc * PINV) & MASK;
    amads(acc, m[k], PL[0]);
    acc = shr29<ALIGNBIT>(acc);
  }
  #pragma unroll
  for(int k=9;k<17;k++){
    #pragma unroll
    for(int i=k-8;i<9;i++) amad(acc, a.l[i], b.l[k-i]);
    #pragma unroll
    for(int i=k-8;i<9;i++) amads(acc, m[i], PL[k-i]);
    r.l[k-9] = (u32)acc & MASK; acc = shr29<ALIGNBIT>(acc);
  }
  r.l[8]=(u32)acc;
  return r;
}
//   E  one asm statement per column chain (a.b terms, then m.P terms): the carry of column k-1 is the addend of the
//      first multiply-add of column k, no 64-bit adds, and only two hazard nops per column instead of one per MAD
__device__ __forceinline__ Fe mul_E(const Fe& a, const Fe& b){
  u64 acc=0; u32 m[9]; Fe r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[0]), "v"(b.l[0]) : "vcc");
  m[0] = ((u32)acc * PINV) & MASK; acc = mad(m[0], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[0]), "s"(PL[1]) : "vcc");
  m[1] = ((u32)acc * PINV) & MASK; acc = mad(m[1], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[2] = ((u32)acc * PINV) & MASK; acc = mad(m[2], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[3] = ((u32)acc * PINV) & MASK; acc = mad(m[3], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[4] = ((u32)acc * PINV) & MASK; acc = mad(m[4], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[5] = ((u32)acc * PINV) & MASK; acc = mad(m[5], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[6] = ((u32)acc * PINV) & MASK; acc = mad(m[6], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[7] = ((u32)acc * PINV) & MASK; acc = mad(m[7], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %10, %0\n\tv_mad_u64_u32 %0, vcc, %2, %11, %0\n\tv_mad_u64_u32 %0, vcc, %3, %12, %0\n\tv_mad_u64_u32 %0, vcc, %4, %13, %0\n\tv_mad_u64_u32 %0, vcc, %5, %14, %0\n\tv_mad_u64_u32 %0, vcc, %6, %15, %0\n\tv_mad_u64_u32 %0, vcc, %7, %16, %0\n\tv_mad_u64_u32 %0, vcc, %8, %17, %0\n\tv_mad_u64_u32 %0, vcc, %9, %18, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[8] = ((u32)acc * PINV) & MASK; acc = mad(m[8], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  r.l[0] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]) : "vcc");
  r.l[1] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]) : "vcc");
  r.l[2] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]) : "vcc");
  r.l[3] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]) : "vcc");
  r.l[4] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]) : "vcc");
  r.l[5] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]) : "vcc");
  r.l[6] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[8]), "v"(b.l[8]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[8]), "s"(PL[8]) : "vcc");
  r.l[7] = (u32)acc & MASK; acc >>= 29;
  r.l[8]=(u32)acc;
  return r;
}
// F / G / H: variant E with the 64-bit shift between columns replaced by v_alignbit_b32 + v_lshrrev_b32 (F, G) and
// the m digit (low 29 bits of acc * PINV) computed with 24-bit multiplies instead of v_mul_lo_u32 (G, H): both move
// work off the multiplier's pipe onto full-rate instructions.
__device__ __forceinline__ void shr29(u64& acc){
  u32 lo=(u32)acc, hi=(u32)(acc>>32), nlo, nhi;
  asm("v_alignbit_b32 %0, %1, %2, 29" : "=v"(nlo) : "v"(hi), "v"(lo));
  asm("v_lshrrev_b32 %0, 29, %1" : "=v"(nhi) : "v"(hi));
  acc = ((u64)nhi<<32)|nlo;
}
__device__ __forceinline__ u32 mdig24(u32 lo){
  constexpr u32 P24 = PINV & 0xffffffu, PH = PINV >> 24;
  const u32 lohi = (lo >> 24) & 0x1fu;
  const u32 t1 = __umul24(lo, P24);
  const u32 t2 = __umul24(lo, PH) + __umul24(lohi, P24);
  return (t1 + (t2 << 24)) & MASK;
}
__device__ __forceinline__ Fe mul_F(const Fe& a, const Fe& b){
  u64 acc=0; u32 m[9]; Fe r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[0]), "v"(b.l[0]) : "vcc");
  m[0] = ((u32)acc * PINV) & MASK; acc = mad(m[0], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[0]), "s"(PL[1]) : "vcc");
  m[1] = ((u32)acc * PINV) & MASK; acc = mad(m[1], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[2] = ((u32)acc * PINV) & MASK; acc = mad(m[2], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[3] = ((u32)acc * PINV) & MASK; acc = mad(m[3], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[4] = ((u32)acc * PINV) & MASK; acc = mad(m[4], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[5] = ((u32)acc * PINV) & MASK; acc = mad(m[5], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[6] = ((u32)acc * PINV) & MASK; acc = mad(m[6], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[7] = ((u32)acc * PINV) & MASK; acc = mad(m[7], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %10, %0\n\tv_mad_u64_u32 %0, vcc, %2, %11, %0\n\tv_mad_u64_u32 %0, vcc, %3, %12, %0\n\tv_mad_u64_u32 %0, vcc, %4, %13, %0\n\tv_mad_u64_u32 %0, vcc, %5, %14, %0\n\tv_mad_u64_u32 %0, vcc, %6, %15, %0\n\tv_mad_u64_u32 %0, vcc, %7, %16, %0\n\tv_mad_u64_u32 %0, vcc, %8, %17, %0\n\tv_mad_u64_u32 %0, vcc, %9, %18, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[8] = ((u32)acc * PINV) & MASK; acc = mad(m[8], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  r.l[0] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]) : "vcc");
  r.l[1] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]) : "vcc");
  r.l[2] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]) : "vcc");
  r.l[3] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]) : "vcc");
  r.l[4] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]) : "vcc");
  r.l[5] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]) : "vcc");
  r.l[6] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[8]), "v"(b.l[8]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[8]), "s"(PL[8]) : "vcc");
  r.l[7] = (u32)acc & MASK; shr29(acc);
  r.l[8]=(u32)acc;
  return r;
}
__device__ __forceinline__ Fe mul_G(const Fe& a, const Fe& b){
  u64 acc=0; u32 m[9]; Fe r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[0]), "v"(b.l[0]) : "vcc");
  m[0] = mdig24((u32)acc); acc = mad(m[0], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[0]), "s"(PL[1]) : "vcc");
  m[1] = mdig24((u32)acc); acc = mad(m[1], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[2] = mdig24((u32)acc); acc = mad(m[2], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[3] = mdig24((u32)acc); acc = mad(m[3], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[4] = mdig24((u32)acc); acc = mad(m[4], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[5] = mdig24((u32)acc); acc = mad(m[5], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[6] = mdig24((u32)acc); acc = mad(m[6], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[7] = mdig24((u32)acc); acc = mad(m[7], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %10, %0\n\tv_mad_u64_u32 %0, vcc, %2, %11, %0\n\tv_mad_u64_u32 %0, vcc, %3, %12, %0\n\tv_mad_u64_u32 %0, vcc, %4, %13, %0\n\tv_mad_u64_u32 %0, vcc, %5, %14, %0\n\tv_mad_u64_u32 %0, vcc, %6, %15, %0\n\tv_mad_u64_u32 %0, vcc, %7, %16, %0\n\tv_mad_u64_u32 %0, vcc, %8, %17, %0\n\tv_mad_u64_u32 %0, vcc, %9, %18, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[8] = mdig24((u32)acc); acc = mad(m[8], PL[0], acc); shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  r.l[0] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]) : "vcc");
  r.l[1] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]) : "vcc");
  r.l[2] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]) : "vcc");
  r.l[3] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]) : "vcc");
  r.l[4] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]) : "vcc");
  r.l[5] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]) : "vcc");
  r.l[6] = (u32)acc & MASK; shr29(acc);
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[8]), "v"(b.l[8]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[8]), "s"(PL[8]) : "vcc");
  r.l[7] = (u32)acc & MASK; shr29(acc);
  r.l[8]=(u32)acc;
  return r;
}
__device__ __forceinline__ Fe mul_H(const Fe& a, const Fe& b){
  u64 acc=0; u32 m[9]; Fe r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[0]), "v"(b.l[0]) : "vcc");
  m[0] = mdig24((u32)acc); acc = mad(m[0], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[0]), "s"(PL[1]) : "vcc");
  m[1] = mdig24((u32)acc); acc = mad(m[1], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[2] = mdig24((u32)acc); acc = mad(m[2], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[3] = mdig24((u32)acc); acc = mad(m[3], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[4] = mdig24((u32)acc); acc = mad(m[4], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[5] = mdig24((u32)acc); acc = mad(m[5], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[6] = mdig24((u32)acc); acc = mad(m[6], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[7] = mdig24((u32)acc); acc = mad(m[7], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %10, %0\n\tv_mad_u64_u32 %0, vcc, %2, %11, %0\n\tv_mad_u64_u32 %0, vcc, %3, %12, %0\n\tv_mad_u64_u32 %0, vcc, %4, %13, %0\n\tv_mad_u64_u32 %0, vcc, %5, %14, %0\n\tv_mad_u64_u32 %0, vcc, %6, %15, %0\n\tv_mad_u64_u32 %0, vcc, %7, %16, %0\n\tv_mad_u64_u32 %0, vcc, %8, %17, %0\n\tv_mad_u64_u32 %0, vcc, %9, %18, %0" : "+v"(acc) : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]), "v"(b.l[0]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  m[8] = mdig24((u32)acc); acc = mad(m[8], PL[0], acc); acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]), "v"(b.l[1]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\tv_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\tv_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\tv_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0" : "+v"(acc) : "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]), "s"(PL[1]) : "vcc");
  r.l[0] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]), "v"(b.l[2]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %8, %0\n\tv_mad_u64_u32 %0, vcc, %2, %9, %0\n\tv_mad_u64_u32 %0, vcc, %3, %10, %0\n\tv_mad_u64_u32 %0, vcc, %4, %11, %0\n\tv_mad_u64_u32 %0, vcc, %5, %12, %0\n\tv_mad_u64_u32 %0, vcc, %6, %13, %0\n\tv_mad_u64_u32 %0, vcc, %7, %14, %0" : "+v"(acc) : "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]), "s"(PL[2]) : "vcc");
  r.l[1] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]), "v"(b.l[3]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %7, %0\n\tv_mad_u64_u32 %0, vcc, %2, %8, %0\n\tv_mad_u64_u32 %0, vcc, %3, %9, %0\n\tv_mad_u64_u32 %0, vcc, %4, %10, %0\n\tv_mad_u64_u32 %0, vcc, %5, %11, %0\n\tv_mad_u64_u32 %0, vcc, %6, %12, %0" : "+v"(acc) : "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]), "s"(PL[3]) : "vcc");
  r.l[2] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]), "v"(b.l[4]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %6, %0\n\tv_mad_u64_u32 %0, vcc, %2, %7, %0\n\tv_mad_u64_u32 %0, vcc, %3, %8, %0\n\tv_mad_u64_u32 %0, vcc, %4, %9, %0\n\tv_mad_u64_u32 %0, vcc, %5, %10, %0" : "+v"(acc) : "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]), "s"(PL[4]) : "vcc");
  r.l[3] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]), "v"(b.l[5]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_mad_u64_u32 %0, vcc, %4, %8, %0" : "+v"(acc) : "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]), "s"(PL[5]) : "vcc");
  r.l[4] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]), "v"(b.l[6]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %4, %0\n\tv_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_mad_u64_u32 %0, vcc, %3, %6, %0" : "+v"(acc) : "v"(m[6]), "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]), "s"(PL[6]) : "vcc");
  r.l[5] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(a.l[7]), "v"(a.l[8]), "v"(b.l[8]), "v"(b.l[7]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %4, %0" : "+v"(acc) : "v"(m[7]), "v"(m[8]), "s"(PL[8]), "s"(PL[7]) : "vcc");
  r.l[6] = (u32)acc & MASK; acc >>= 29;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a.l[8]), "v"(b.l[8]) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[8]), "s"(PL[8]) : "vcc");
  r.l[7] = (u32)acc & MASK; acc >>= 29;
  r.l[8]=(u32)acc;
  return r;
}
template<int V> __device__ __forceinline__ Fe mulv(const Fe& a, const Fe& b){
  if constexpr (V==0) return mul_A(a,b);
  else if constexpr (V==1) return mul_BC<false>(a,b);
  else if constexpr (V==2) return mul_BC<true>(a,b);
  else if constexpr (V==3) return mul_E(a,b);
  else if constexpr (V==4) return mul_F(a,b);
  else if constexpr (V==5) return mul_G(a,b);
  else return mul_H(a,b);
}
// NCH independent chains per lane; dynamic LDS request limits the occupancy to WPS waves per SIMD
template<int V, int NCH>
__global__ void __launch_bounds__(256) k_mont(u64* out, u32 a, u32 b, int iters){
  Fe x[NCH], y;
  for(int c=0;c<NCH;c++) for(int j=0;j<9;j++) x[c].l[j]=(threadIdx.x*2654435761u+j*40503u+a+c)&MASK;
  for(int j=0;j<9;j++) y.l[j]=(threadIdx.x*40503u+j*2654435761u+b)&MASK;
  for(int t=0;t<iters;t++){
    #pragma unroll
    for(int c=0;c<NCH;c++) x[c]=mulv<V>(x[c],y);
  }
  u32 s=0; for(int c=0;c<NCH;c++) for(int j=0;j<9;j++) s^=x[c].l[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<class F> double timeit(F f){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for(int r=0;r<5;r++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/5.0;
}
template<int V, int NCH> void run(const char* name, u64* out, int CU){
  // correctness vs variant A on one block
  for(int w : {1,2,3,4,8}){
    int blocks=CU*w, iters=256;
    double ms=timeit([&]{ hipLaunchKernelGGL((k_mont<V,NCH>), dim3(blocks), dim3(256), 0, 0, out, 12345u, 678u, iters); });
    double muls=(double)blocks*256*iters*NCH; double cyc = ms*1e-3*2.4e9*(CU*4.0)/(muls/64.0);
    printf("%-28s x%d waves/SIMD=%d %.3f ms  %.2f Gmul/s  ~%.0f cyc/wave-mul\n", name, NCH, w, ms, muls/ms*1e-6, cyc);
  }
}

// ---- round 4 ---------------------------------------------------------------------------------------------------
// J: the FP64-FMA multiplication of tools/fp64_mont.h (52-bit limbs, hi / lo FMA pairs in round-toward-zero mode)
template<int NCH>
__global__ void __launch_bounds__(256) k_mont_fp64(u64* out, u32 a, u32 b, int iters){
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");   // FP_ROUND of f64 / f16: toward zero
  using namespace fp64mont;
  Consts k;
  for(int i=0;i<5;i++) k.p[i]=to_d(P52[i]);
  k.pinv=to_d(PINV52);
  Fe5 x[NCH], y;
  for(int c=0;c<NCH;c++) for(int j=0;j<5;j++) x[c].d[j]=to_d(((u64)(threadIdx.x*2654435761u+j*40503u+a+c)<<17 ^ (u64)(j*77u+c)) & (j<4? M52 : ((1ull<<46)-1)));
  for(int j=0;j<5;j++) y.d[j]=to_d(((u64)(threadIdx.x*40503u+j*2654435761u+b)<<19 ^ (u64)(j*13u)) & (j<4? M52 : ((1ull<<46)-1)));
  for(int t=0;t<iters;t++){
    #pragma unroll
    for(int c=0;c<NCH;c++) x[c]=mul(x[c],y,k);
  }
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0");
  u64 s=0; for(int c=0;c<NCH;c++) for(int j=0;j<5;j++) s^=(u64)x[c].d[j] << (j*3);
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// the same chain on the host (fesetround) for the parity line
static u64 fp64_host_lane(u32 tid, u32 a, u32 b, int iters){
  using namespace fp64mont;
  Consts k; for(int i=0;i<5;i++) k.p[i]=to_d(P52[i]); k.pinv=to_d(PINV52);
  Fe5 x,y;
  for(int j=0;j<5;j++) x.d[j]=to_d(((u64)(tid*2654435761u+j*40503u+a+0)<<17 ^ (u64)(j*77u+0)) & (j<4? M52 : ((1ull<<46)-1)));
  for(int j=0;j<5;j++) y.d[j]=to_d(((u64)(tid*40503u+j*2654435761u+b)<<19 ^ (u64)(j*13u)) & (j<4? M52 : ((1ull<<46)-1)));
  for(int t=0;t<iters;t++) x=mul(x,y,k);
  u64 s=0; for(int j=0;j<5;j++) s^=(u64)x.d[j] << (j*3);
  return s;
}
// EL: variant E with a loop body as long as the level-1 loop (ten multiplications over five values, straight-line,
// ~2300 instructions = ~19 KB of code): does the instruction stream of a BIG loop issue as well as the 232-instruction
// loop of E?  (Level 1 measures 5.4 cycles per instruction at 3 waves per SIMD, E 4.0.)
__global__ void __launch_bounds__(256) k_mont_long(u64* out, u32 a, u32 b, int iters, unsigned long long* clk){
  Fe v[5];
  for(int c=0;c<5;c++) for(int j=0;j<9;j++) v[c].l[j]=(threadIdx.x*2654435761u+j*40503u+a+c*977u)&MASK;
  const unsigned long long w0=wall_clock64(), c0=clock64();
  for(int t=0;t<iters;t++){
    v[0]=mul_E(v[0],v[1]); v[1]=mul_E(v[1],v[2]); v[2]=mul_E(v[2],v[3]); v[3]=mul_E(v[3],v[4]); v[4]=mul_E(v[4],v[0]);
    v[0]=mul_E(v[0],v[2]); v[1]=mul_E(v[1],v[3]); v[2]=mul_E(v[2],v[4]); v[3]=mul_E(v[3],v[0]); v[4]=mul_E(v[4],v[1]);
  }
  const unsigned long long w1=wall_clock64(), c1=clock64();
  if(clk && threadIdx.x==0){ clk[2*blockIdx.x]=w1-w0; clk[2*blockIdx.x+1]=c1-c0; }
  u32 s=0; for(int c=0;c<5;c++) for(int j=0;j<9;j++) s^=v[c].l[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// E with clock stamps: clock64() (s_memtime) against wall_clock64() (s_memrealtime, 100 MHz) over the kernel's loop
template<int NCH>
__global__ void __launch_bounds__(256) k_mont_clk(u64* out, u32 a, u32 b, int iters, unsigned long long* clk){
  Fe x[NCH], y;
  for(int c=0;c<NCH;c++) for(int j=0;j<9;j++) x[c].l[j]=(threadIdx.x*2654435761u+j*40503u+a+c)&MASK;
  for(int j=0;j<9;j++) y.l[j]=(threadIdx.x*40503u+j*2654435761u+b)&MASK;
  const unsigned long long w0=wall_clock64(), c0=clock64();
  for(int t=0;t<iters;t++){
    #pragma unroll
    for(int c=0;c<NCH;c++) x[c]=mul_E(x[c],y);
  }
  const unsigned long long w1=wall_clock64(), c1=clock64();
  if(clk && threadIdx.x==0){ clk[2*blockIdx.x]=w1-w0; clk[2*blockIdx.x+1]=c1-c0; }
  u32 s=0; for(int c=0;c<NCH;c++) for(int j=0;j<9;j++) s^=x[c].l[j]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
static void report_clk(const char* what, unsigned long long* d_clk, int blocks){
  std::vector<unsigned long long> h(2*(size_t)blocks);
  hipMemcpy(h.data(), d_clk, h.size()*8, hipMemcpyDeviceToHost);
  double rmin=1e30,rmax=0,rsum=0; int n=0;
  for(int i=0;i<blocks;i++){ if(!h[2*i]) continue; double r=(double)h[2*i+1]/(double)h[2*i]; rmin=r<rmin?r:rmin; rmax=r>rmax?r:rmax; rsum+=r; n++; }
  printf("    %s: clock64 ticks per wall_clock64 tick (100 MHz): mean %.3f min %.3f max %.3f over %d workgroups -> clock64 runs at %.1f MHz\n", what, rsum/n, rmin, rmax, n, rsum/n*100.0);
}

int main(){
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop,0);
  int CU=prop.multiProcessorCount;
  u64 *o0,*o1; hipMalloc(&o0, sizeof(u64)*CU*16*256); hipMalloc(&o1, sizeof(u64)*CU*16*256);
  // parity of the variants (same inputs -> same checksum words)
  u64 h[7][256];
  for(int v=0;v<7;v++){
    if(v==0) hipLaunchKernelGGL((k_mont<0,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    if(v==1) hipLaunchKernelGGL((k_mont<1,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    if(v==2) hipLaunchKernelGGL((k_mont<2,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    if(v==3) hipLaunchKernelGGL((k_mont<3,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    if(v==4) hipLaunchKernelGGL((k_mont<4,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    if(v==5) hipLaunchKernelGGL((k_mont<5,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    if(v==6) hipLaunchKernelGGL((k_mont<6,1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    hipMemcpy(h[v], o0, sizeof(u64)*256, hipMemcpyDeviceToHost);
  }
  int bad=0; for(int i=0;i<256;i++) { for(int v=1;v<7;v++) if(h[0][i]!=h[v][i]) { bad++; break; } }
  printf("variant parity: %s\n", bad? "MISMATCH":"ok");
  run<0,1>("A plain C", o0, CU);
  run<1,1>("B chained carry", o0, CU);
  run<2,1>("C chained + alignbit", o0, CU);
  run<3,1>("E per-column asm chains", o0, CU);
  run<4,1>("F = E, 32-bit shifts", o0, CU);
  run<5,1>("G = F, 24-bit m digit", o0, CU);
  run<6,1>("H = E, 24-bit m digit", o0, CU);
  run<0,2>("A plain C", o0, CU);
  run<1,2>("B chained carry", o0, CU);
  run<2,2>("C chained + alignbit", o0, CU);
  run<3,2>("E per-column asm chains", o0, CU);

  // ---- round 4
  {
    // parity of J: device == host (round toward zero on both) for every lane of one workgroup
    fesetround(FE_TOWARDZERO);
    hipLaunchKernelGGL((k_mont_fp64<1>), dim3(1), dim3(256), 0, 0, o0, 12345u, 678u, 7);
    u64 hj[256]; hipMemcpy(hj, o0, sizeof(hj), hipMemcpyDeviceToHost);
    int badj=0; for(u32 i=0;i<256;i++) if(hj[i]!=fp64_host_lane(i,12345u,678u,7)) badj++;
    fesetround(FE_TONEAREST);
    printf("variant J parity (device FP64-FMA chain == host chain, itself checked against exact integers by tools/fp64_mont_hostcheck.cc): %s\n", badj? "MISMATCH":"ok");
    for(int nch : {1,2}) for(int w : {1,2,3,4,8}){
      int blocks=CU*w, iters=256;
      double ms = nch==1 ? timeit([&]{ hipLaunchKernelGGL((k_mont_fp64<1>), dim3(blocks), dim3(256), 0, 0, o0, 12345u, 678u, iters); })
                         : timeit([&]{ hipLaunchKernelGGL((k_mont_fp64<2>), dim3(blocks), dim3(256), 0, 0, o0, 12345u, 678u, iters); });
      double muls=(double)blocks*256*iters*nch; double cyc = ms*1e-3*2.4e9*(CU*4.0)/(muls/64.0);
      printf("%-28s x%d waves/SIMD=%d %.3f ms  %.2f Gmul/s  ~%.0f cyc/wave-mul\n", "J FP64 FMA, 52-bit limbs", nch, w, ms, muls/ms*1e-6, cyc);
    }
    unsigned long long* d_clk; hipMalloc(&d_clk, sizeof(unsigned long long)*2*CU*16);
    // E in a loop body of ten multiplications (code size of the level-1 loop), 3 waves per SIMD
    for(int w : {1,2,3,4}){
      int blocks=CU*w, iters=26;
      hipMemset(d_clk,0,sizeof(unsigned long long)*2*blocks);
      double ms=timeit([&]{ hipLaunchKernelGGL(k_mont_long, dim3(blocks), dim3(256), 0, 0, o0, 12345u, 678u, iters, d_clk); });
      double muls=(double)blocks*256*iters*10; double cyc = ms*1e-3*2.4e9*(CU*4.0)/(muls/64.0);
      printf("%-28s x1 waves/SIMD=%d %.3f ms  %.2f Gmul/s  ~%.0f cyc/wave-mul\n", "EL = E, 10 muls per loop body", w, ms, muls/ms*1e-6, cyc);
      if(w==3) report_clk("EL", d_clk, blocks);
    }
    // E sustained: the same kernel for 0.3 ms, 1.2 ms (one level-1 launch), 5 ms and 40 ms — does the rate hold?
    for(int iters : {256, 1024, 4096, 32768}){
      int blocks=CU*3;
      hipMemset(d_clk,0,sizeof(unsigned long long)*2*blocks);
      double ms=timeit([&]{ hipLaunchKernelGGL((k_mont_clk<1>), dim3(blocks), dim3(256), 0, 0, o0, 12345u, 678u, iters, d_clk); });
      double muls=(double)blocks*256*iters; double cyc = ms*1e-3*2.4e9*(CU*4.0)/(muls/64.0);
      printf("%-28s x1 waves/SIMD=3 iters=%d %.3f ms  %.2f Gmul/s  ~%.0f cyc/wave-mul\n", "E sustained", iters, ms, muls/ms*1e-6, cyc);
      report_clk("E sustained", d_clk, blocks);
    }
    hipFree(d_clk);
  }
  return 0;
}
