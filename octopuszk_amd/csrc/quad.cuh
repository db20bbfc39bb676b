// G1 group law on a lane QUAD, for the serial chains (the Horner evaluation over the windows at the end of an
// MSM: 112 dependent doublings; the doubling chain of the fixed-base table: 128).
//
// One lane runs a Jacobian doubling as seven dependent Fq multiplications (~0.6 us each on a lone wave,
// profiles/r01_ubench.txt), although the formula is only three multiplications DEEP:
//     level 1   X^2   Y^2   Y Z                 level 2   (3 X^2)^2   (Y^2)^2   (X + Y^2)^2
//     level 3   E (D - X3)
// and an addition (add-2007-bl, sixteen) five.  Here the four lanes of a quad hold the SAME point; at every
// level lane k multiplies the k-th operand pair, and the products come back to all four lanes through DPP
// quad-permute moves (one VALU instruction per limb, no LDS) — a doubling costs three multiplication times
// instead of seven, an addition five instead of sixteen.  The linear parts are computed redundantly by all
// four lanes, so the quad never diverges and every lane always holds the whole result.  Formulas, bounds
// and special cases are those of jac_dbl / jac_add (ec.cuh), whose bound-carrying types check them.
//
// G2 keeps its lane PAIR (Fe2L, fq2.cuh), which splits every Fq2 product instead.
#pragma once
#include "ec.cuh"

namespace ozk {

struct G1CfgQ {   // same coordinate types as G1Cfg; a distinct tag selects the overloads below
  using EX = G1Cfg::EX;
  using EY = G1Cfg::EY;
  using EZ = G1Cfg::EZ;
  using EA = G1Cfg::EA;
};

#if defined(__HIPCC__)
__device__ __forceinline__ Jac<G1CfgQ> to_pair(const Jac<G1Cfg>& p) {
  Jac<G1CfgQ> r;
  r.X = p.X;
  r.Y = p.Y;
  r.Z = p.Z;
  return r;
}
__device__ __forceinline__ Jac<G1Cfg> from_pair(const Jac<G1CfgQ>& p) {
  Jac<G1Cfg> r;
  r.X = p.X;
  r.Y = p.Y;
  r.Z = p.Z;
  return r;
}

// value of lane K of this lane's quad
template <int K, class P, int B>
__device__ __forceinline__ Fe<P, B> quad_bcast(const Fe<P, B>& v) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)v.l[i], K * 0x55, 0xf, 0xf, true);
  return r;
}

// out[k] = a[k] * b[k], k < 4, in ONE multiplication time: lane (k mod 4) computes product k
template <class P, int BA, int BB>
__device__ __forceinline__ void quad_mul4(const Fe<P, BA> (&a)[4], const Fe<P, BB> (&b)[4],
                                          decltype(mul(Fe<P, BA>(), Fe<P, BB>())) (&out)[4]) {
  const int role = threadIdx.x & 3;
  Fe<P, BA> x = a[0];
  Fe<P, BB> y = b[0];
#pragma unroll
  for (int k = 1; k < 4; k++) {
    x = select_el(role == k, a[k], x);
    y = select_el(role == k, b[k], y);
  }
  const auto r = mul(x, y);
  out[0] = quad_bcast<0>(r);
  out[1] = quad_bcast<1>(r);
  out[2] = quad_bcast<2>(r);
  out[3] = quad_bcast<3>(r);
}

// dbl-2009-l on a quad (jac_dbl of ec.cuh, BNG1.java:133-161): three multiplication times.
// The bounds are left to grow where the multiplier's input slack takes them (B1 B2 <= 169 x 256: 96 x 96, 116 x 116
// and 60 x 234 are far inside), so that only the two loop-carried coordinates X3, Y3 are brought back — four
// conditional subtractions per doubling instead of the fifteen of the first version (which reduced every operand
// to < 2p on entry): k_fb_chain (128 doublings) 455 -> 369 us, k_finalize 560 -> 530 us.
__device__ __forceinline__ Jac<G1CfgQ> jac_dbl(const Jac<G1CfgQ>& p) {
  using FI = Fe<FqParams, 96>;   // the coordinates as they come: X < 94, Y < 73, Z < 78 (sixteenths of p)
  const FI X1 = FI(p.X), Y1 = FI(p.Y), Z1 = FI(p.Z);
  decltype(mul(FI(), FI())) l1[4];                                // < 20
  {
    const FI a[4] = {X1, Y1, Y1, Y1}, b[4] = {X1, Y1, Z1, Z1};
    quad_mul4(a, b, l1);
  }
  const auto A = l1[0], B = l1[1], YZ = l1[2];
  const auto E = add(dbl(A), A);                                  // 3 X^2        (< 60)
  using F2 = Fe<FqParams, 116>;
  static_assert(std::is_convertible<decltype(add(X1, B)), F2>::value && std::is_convertible<decltype(E), F2>::value,
                "level-2 operand bound");
  decltype(mul(F2(), F2())) l2[4];                                // < 21
  {
    const F2 a[4] = {F2(E), F2(B), F2(add(X1, B)), F2(B)};
    quad_mul4(a, a, l2);
  }
  const auto F = l2[0], CC = l2[1], T = l2[2];
  const auto D = dbl(sub(T, add(A, CC)));                         // 2 ((X + B)^2 - A - C)
  const auto X3 = reduce_to<80>(sub(F, dbl(D)));
  const auto C8 = dbl(dbl(dbl(CC)));
  const auto Y3 = sub(mul(E, sub(D, X3)), C8);                    // (the same product on all four lanes)
  Jac<G1CfgQ> r;
  r.X = G1CfgQ::EX(X3);
  r.Y = G1CfgQ::EY(reduce_to<64>(Y3));
  r.Z = G1CfgQ::EZ(dbl(YZ));
  return r;
}

// add-2007-bl on a quad (jac_add of ec.cuh, BNG1.java:38-97): five multiplication times
__device__ __forceinline__ Jac<G1CfgQ> jac_add(const Jac<G1CfgQ>& p, const Jac<G1CfgQ>& q) {
  if (is_inf(p)) return q;   // (uniform over the quad: all four lanes hold the same points)
  if (is_inf(q)) return p;
  using FI = Fe<FqParams, 96>;   // the coordinates as they come (see jac_dbl above)
  const FI X1 = FI(p.X), Y1 = FI(p.Y), Z1 = FI(p.Z);
  const FI X2 = FI(q.X), Y2 = FI(q.Y), Z2 = FI(q.Z);
  using FL1 = decltype(mul(FI(), FI()));                          // < 20
  FL1 l1[4];
  {
    const FI a[4] = {Z1, Z2, Y1, Y2}, b[4] = {Z1, Z2, Z2, Z1};
    quad_mul4(a, b, l1);
  }
  const auto Z1Z1 = l1[0], Z2Z2 = l1[1], Y1Z2 = l1[2], Y2Z1 = l1[3];
  decltype(mul(FI(), FL1())) l2[4];
  {
    const FI a[4] = {X1, X2, FI(Y1Z2), FI(Y2Z1)};
    const FL1 b[4] = {Z2Z2, Z1Z1, Z2Z2, Z1Z1};
    quad_mul4(a, b, l2);
  }
  const auto U1 = l2[0], U2 = l2[1], S1 = l2[2], S2 = l2[3];
  const auto H = sub(U2, U1);
  const auto rh = sub(S2, S1);
  if (is_zero(H)) {
    if (is_zero(rh)) return jac_dbl(p);
    return jac_infinity<G1CfgQ>();
  }
  const auto H2 = dbl(H);
  const auto r = dbl(rh);
  const auto ZS = add(Z1, Z2);
  using F3 = Fe<FqParams, 192>;
  static_assert(std::is_convertible<decltype(H2), F3>::value && std::is_convertible<decltype(r), F3>::value &&
                    std::is_convertible<decltype(ZS), F3>::value, "level-3 operand bound");
  decltype(mul(F3(), F3())) l3[4];
  {
    const F3 a[4] = {F3(H2), F3(r), F3(ZS), F3(ZS)};
    quad_mul4(a, a, l3);
  }
  const auto I = l3[0], RR = l3[1], ZZ = l3[2];
  const auto Zd = sub(ZZ, add(Z1Z1, Z2Z2));
  using F4A = Fe<FqParams, 80>;
  using F4B = Fe<FqParams, 49>;
  static_assert(std::is_convertible<decltype(Zd), F4A>::value && std::is_convertible<decltype(H), F4B>::value &&
                    std::is_convertible<decltype(I), F4B>::value, "level-4 operand bound");
  decltype(mul(F4A(), F4B())) l4[4];
  {
    const F4A a[4] = {F4A(H), F4A(U1), F4A(Zd), F4A(Zd)};
    const F4B b[4] = {F4B(I), F4B(I), F4B(H), F4B(H)};
    quad_mul4(a, b, l4);
  }
  const auto J = l4[0], V = l4[1], Z3 = l4[2];
  const auto X3 = sub(RR, add(J, dbl(V)));
  const auto Y3 = mulsub(r, sub(V, X3), dbl(S1), J);            // (one dual product, on all four lanes)
  Jac<G1CfgQ> out;
  out.X = G1CfgQ::EX(X3);
  out.Y = G1CfgQ::EY(Y3);
  out.Z = G1CfgQ::EZ(Z3);
  return out;
}
#endif  // __HIPCC__

}  // namespace ozk
