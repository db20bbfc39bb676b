// What does the level-1 FORMULA cost when nothing else happens?  A register-only loop of the library's own mixed
// addition (ec.cuh xyzz_madd_lazy over G1Cfg, the function k_segreduce<G1Cfg,true,true> calls) at the kernel's
// occupancy — no gather, no run logic, no stores — against the same count of bare multiplications (variant E of
// tools/ubench_mont.hip): the difference between this and the kernel (1.18-1.22 ms for 16.7 M additions alone) is what
// the gather / run-boundary / store machinery costs; the difference between this and 10 multiplications at E's rate
// is what the additions, carries and selects inside the formula cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOZK_WITH_G2 tools/ubench_madd.hip -o tools/ubench_madd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../octopuszk_amd/csrc/curve.cuh"
using namespace ozk;

template <int MODE>   // 0: xyzz_madd_lazy, alternating sign; 1: the carried xyzz_madd, no sign; 2: lazy, no sign; 3: carried, alternating sign
__global__ void __launch_bounds__(256) k_madd(u32* out, u32 seed, int iters) {
  extern __shared__ u32 dummy[];   // (occupancy cap only)
  using EA = typename G1Cfg::EA;
  u32 w[8];
  for (int j = 0; j < 8; j++) w[j] = (threadIdx.x * 2654435761u + j * 40503u + seed) & 0x0fffffffu;
  Aff<EA> q, q2;
  q.x = EA(to_mont<FqParams>(w));
  for (int j = 0; j < 8; j++) w[j] = (w[j] * 1664525u + 1013904223u) & 0x0fffffffu;
  q.y = EA(to_mont<FqParams>(w));
  for (int j = 0; j < 8; j++) w[j] = (w[j] * 1664525u + 1013904223u) & 0x0fffffffu;
  q2.x = EA(to_mont<FqParams>(w));
  q2.y = q.y;
  Xyzz<G1Cfg> acc = xyzz_from_affine<G1Cfg>(q2);
  for (int t = 0; t < iters; t++) {
    if constexpr (MODE == 0) acc = xyzz_madd_lazy(acc, q, (t & 1) != 0);
    else if constexpr (MODE == 1) acc = xyzz_madd(acc, q);
    else if constexpr (MODE == 2) acc = xyzz_madd_lazy(acc, q, false);
    else {   // the carried form with the sign the way RunAcc::decode applies it
      Aff<EA> qs = q;
      qs.y = select_el((t & 1) != 0, EA(reduce_to<17>(neg(q.y))), q.y);
      acc = xyzz_madd(acc, qs);
    }
  }
  u32 s = 0;
  for (int j = 0; j < 9; j++) s ^= acc.X.l[j] ^ acc.Y.l[j] ^ acc.ZZ.l[j] ^ acc.ZZZ.l[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + dummy[0] * 0;
}

template <class F> double timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; r++) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int CU = prop.multiProcessorCount;
  u32* out;
  (void)hipMalloc(&out, sizeof(u32) * CU * 8 * 256);
  const char* names[4] = {"xyzz_madd_lazy, alternating sign", "xyzz_madd (carried), no sign", "xyzz_madd_lazy, no sign",
                          "xyzz_madd (carried), alternating sign"};
  for (int mode = 0; mode < 4; mode++)
    for (int w : {2, 3, 4}) {
      const int blocks = CU * w, iters = 86;   // one round of the chip, 86 additions per lane: the lone level-1 launch
      const size_t lds = w == 3 ? 41216 : (w == 2 ? 65536 : 0);
      double ms = mode == 0   ? timeit([&] { hipLaunchKernelGGL(k_madd<0>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 1 ? timeit([&] { hipLaunchKernelGGL(k_madd<1>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 2 ? timeit([&] { hipLaunchKernelGGL(k_madd<2>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                              : timeit([&] { hipLaunchKernelGGL(k_madd<3>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); });
      const double adds = (double)blocks * 256 * iters;
      printf("%-38s waves/SIMD=%d  %.3f ms for %.2f M additions -> %.3f ms per 16.78 M (a 2^20 MSM's level 1); %.1f G mulmod-equivalents/s (10 per addition)\n",
             names[mode], w, ms, adds * 1e-6,
             ms * 16.777216e6 / adds, adds * 10 / ms * 1e-6);
    }
  return 0;
}
