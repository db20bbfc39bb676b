// The G2 level-1 formula alone: RunAccLds<G2Cfg>::accumulate_q (the XYZZ accumulator in LDS, the staged mixed addition of
// k_segreduce<G2Cfg,true,false>) in a loop over a register-resident q with per-lane signs, at the kernel's occupancy (two
// workgroups of 256 per CU: 72 KiB of LDS each) — against the kernel's 3.89 ms per 2^20 MSM (16.78 M additions).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOZK_WITH_G2 tools/ubench_madd_g2.hip -o tools/ubench_madd_g2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../octopuszk_amd/csrc/msm_var.cuh"
#include "../octopuszk_amd/csrc/fq2.cuh"
using namespace ozk;

__global__ void __launch_bounds__(256, 2) k_madd_g2(u32* out, u32 seed, int iters) {
  extern __shared__ u32 lds[];
  using CV = G2Cfg;
  using EA = typename CV::EA;
  using ET = ElemTraits<EA>;
  u32 w[16];
  for (int j = 0; j < 16; j++) w[j] = (threadIdx.x * 2654435761u + j * 40503u + seed) & 0x0fffffffu;
  Aff<EA> q, q2;
  q.x = ET::from_wire(w);
  for (int j = 0; j < 16; j++) w[j] = (w[j] * 1664525u + 1013904223u) & 0x0fffffffu;
  q.y = ET::from_wire(w);
  for (int j = 0; j < 16; j++) w[j] = (w[j] * 1664525u + 1013904223u) & 0x0fffffffu;
  q2.x = ET::from_wire(w);
  q2.y = q.y;
  RunAccLds<CV> acc;
  acc.init(lds);
  acc.start_q(q2);
  const EA ny = EA(reduce_to<17>(neg(q.y)));
  for (int t = 0; t < iters; t++) {
    Aff<EA> qs = q;
    qs.y = select_el((((threadIdx.x * 2654435761u) >> (t & 31)) & 1) != 0, ny, q.y);
    acc.accumulate_q(qs);
  }
  const Xyzz<CV> a = acc.get();
  u32 o[ET::RAW_WORDS];
  ElemTraits<typename CV::XX>::store_raw(a.X, o);
  u32 s = 0;
  for (int j = 0; j < ET::RAW_WORDS; j++) s ^= o[j];
  ElemTraits<typename CV::XZZZ>::store_raw(a.ZZZ, o);
  for (int j = 0; j < ET::RAW_WORDS; j++) s ^= o[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int CU = prop.multiProcessorCount;
  u32* out;
  (void)hipMalloc(&out, sizeof(u32) * CU * 4 * 256);
  const size_t lds = (size_t)RunAccLds<G2Cfg>::LDS_WORDS * 256 * 4;
  (void)hipFuncSetAttribute((const void*)k_madd_g2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; rep++) {
    const int blocks = CU * 2, iters = 64;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_madd_g2, dim3(blocks), dim3(256), lds, 0, out, 7u, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_madd_g2, dim3(blocks), dim3(256), lds, 0, out, 7u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5.0f;
    const double adds = (double)blocks * 256 * iters;
    printf("G2 staged mixed addition (LDS accumulator), registers-only q, 2 workgroups per CU: %.3f ms for %.2f M additions -> %.3f ms per 16.78 M (a 2^20 G2 MSM's level 1); LDS %zu B per workgroup, error %s\n",
           ms, adds * 1e-6, ms * 16.777216e6 / adds, lds, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
