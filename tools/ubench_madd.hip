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

// xyzz_madd_lazy with the sign applied AFTER the product y ZZZ instead of to y before it:
//   R = +-(y ZZZ) - Y = negate ? (K2 p - S) + (K p - Y) : S + (K p - Y), one carry pass (R is squared)
template <class CV>
__device__ __forceinline__ Xyzz<CV> xyzz_madd_lazy2(const Xyzz<CV>& p, const Aff<typename CV::EA>& q, bool negate) {
  if (is_inf(q)) return p;
  if (is_inf(p)) {
    Aff<typename CV::EA> qs = q;
    qs.y = select_el(negate, typename CV::EA(reduce_to<17>(neg(q.y))), q.y);
    return xyzz_from_affine<CV>(qs);
  }
  const auto U2 = mul(q.x, p.ZZ);
  const auto S = mul(q.y, p.ZZZ);
  const auto P = sub(U2, p.X);
  const auto nY = neg_nc(p.Y);           // K p - Y, loose
  const auto nS = neg_nc(S);             // K2 p - S, loose
  // (both branches as loose sums, then one carry pass)
  FeL<FqParams, 96, 8> Rl;   // value < (64 + 32) p / 16: K p - Y with K = 4 (Y < 52/16 p), K2 p - S with K2 = 2
#pragma unroll
  for (int i = 0; i < 9; i++) Rl.l[i] = nY.l[i] + (negate ? nS.l[i] : S.l[i]);
  const auto R = normalise(Rl);
  const auto PP = sqr(P);
  if (is_zero(PP)) {
    if (is_zero(R)) {
      Aff<typename CV::EA> qs = q;
      qs.y = select_el(negate, typename CV::EA(reduce_to<17>(neg(q.y))), q.y);
      return xyzz_dbl_affine<CV>(qs);
    }
    Xyzz<CV> z = p;
    z.ZZ = typename CV::XZZ(el_zero(q.x));
    z.ZZZ = typename CV::XZZZ(el_zero(q.x));
    return z;
  }
  const auto PPP = mul(P, PP);
  const auto Q = mul(p.X, PP);
  const auto X3 = sub_sub2(sqr(R), PPP, Q);
  const auto Y3 = mul2(sub_nc(Q, X3), R, nY, PPP);
  Xyzz<CV> out;
  out.X = typename CV::XX(X3);
  out.Y = typename CV::XY(Y3);
  out.ZZ = typename CV::XZZ(mul(p.ZZ, PP));
  out.ZZZ = typename CV::XZZZ(mul(p.ZZZ, PPP));
  return out;
}

struct G1CfgB : G1Cfg {   // the fixed point of lazy2's schedule: X3 < (20 + 32 + 64) p / 16
  using XX = Fe<FqParams, 116>;
};

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
  using CVX = std::conditional_t<(MODE >= 5), G1CfgB, G1Cfg>;
  Xyzz<CVX> acc = xyzz_from_affine<CVX>(q2);
  for (int t = 0; t < iters; t++) {
    if constexpr (MODE == 0) acc = xyzz_madd_lazy(acc, q, (t & 1) != 0);
    else if constexpr (MODE == 1) acc = xyzz_madd(acc, q);
    else if constexpr (MODE == 2) acc = xyzz_madd_lazy(acc, q, false);
    else if constexpr (MODE == 4) acc = xyzz_madd_lazy(acc, q, ((threadIdx.x * 2654435761u) >> (t & 31)) & 1);   // per-lane signs
    else if constexpr (MODE == 5) acc = xyzz_madd_lazy2(acc, q, ((threadIdx.x * 2654435761u) >> (t & 31)) & 1);
    else if constexpr (MODE == 6) acc = xyzz_madd_lazy2(acc, q, (t & 1) != 0);
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
  const char* names[7] = {"xyzz_madd_lazy, alternating sign", "xyzz_madd (carried), no sign", "xyzz_madd_lazy, no sign",
                          "xyzz_madd (carried), alternating sign", "xyzz_madd_lazy, per-lane sign", "lazy2 (sign after y ZZZ), per-lane",
                          "lazy2 (sign after y ZZZ), alternating"};
  for (int mode = 0; mode < 7; mode++)
    for (int w : {3, 4}) {
      const int blocks = CU * w, iters = 86;   // one round of the chip, 86 additions per lane: the lone level-1 launch
      const size_t lds = w == 3 ? 41216 : (w == 2 ? 65536 : 0);
      double ms = mode == 0   ? timeit([&] { hipLaunchKernelGGL(k_madd<0>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 1 ? timeit([&] { hipLaunchKernelGGL(k_madd<1>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 2 ? timeit([&] { hipLaunchKernelGGL(k_madd<2>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 3 ? timeit([&] { hipLaunchKernelGGL(k_madd<3>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 4 ? timeit([&] { hipLaunchKernelGGL(k_madd<4>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                  : mode == 5 ? timeit([&] { hipLaunchKernelGGL(k_madd<5>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); })
                              : timeit([&] { hipLaunchKernelGGL(k_madd<6>, dim3(blocks), dim3(256), lds, 0, out, 7u, iters); });
      const double adds = (double)blocks * 256 * iters;
      printf("%-38s waves/SIMD=%d  %.3f ms for %.2f M additions -> %.3f ms per 16.78 M (a 2^20 MSM's level 1); %.1f G mulmod-equivalents/s (10 per addition)\n",
             names[mode], w, ms, adds * 1e-6,
             ms * 16.777216e6 / adds, adds * 10 / ms * 1e-6);
    }
  return 0;
}
