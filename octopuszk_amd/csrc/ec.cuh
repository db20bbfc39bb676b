// Short-Weierstrass (a = 0) group law in Jacobian coordinates, generic over the
// coordinate field (Fq for G1, Fq2 for G2), one point per lane.
//
// Formulas restate the reference's Java/CUDA group law with lazy reductions:
//   add   = add-2007-bl   (algebra/curves/barreto_naehrig/BNG1.java:38-97,
//                          algebra_msm_VariableBaseMSM.cu:396-547)
//   twice = dbl-2009-l    (BNG1.java:133-161, VariableBaseMSM.cu:290-394)
//   madd  = madd-2007-bl  (the "Potentially use mixed addition here" of
//                          algebra/msm/VariableBaseMSM.java:167; Z2 = 1)
// Infinity is Z == 0 (mod p) as in BNG1.java:103-105.  Results are equal to the
// reference's as group elements; byte-equality is defined on the affine-normalised
// point (SURVEY.md §7 "Hard parts").
//
// CV is a curve configuration giving the loop-carried coordinate types EX / EY / EZ
// (e.g. Fe<FqParams, 94> / <73> / <78>: the fixed point of the madd schedule below with
// NO conditional subtraction inside the loop) and the affine coordinate type EA.
// Every function converts its results to those types, and that conversion
// static_asserts the bound, so the lazy-reduction schedule is proven at compile time.
#pragma once
#include "fp29.cuh"

namespace ozk {

template <class EA>
struct Aff {  // affine, Montgomery form, coordinates < p.  (0, 0) encodes infinity.
  EA x, y;
};

template <class CV>
struct Jac {
  typename CV::EX X;
  typename CV::EY Y;
  typename CV::EZ Z;
};

template <class CV>
OZK_HD bool is_inf(const Jac<CV>& p) { return is_zero(p.Z); }

template <class EA>
OZK_HD bool is_inf(const Aff<EA>& q) { return is_zero(q.x) && is_zero(q.y); }

template <class CV>
OZK_HD Jac<CV> from_affine(const Aff<typename CV::EA>& q) {
  Jac<CV> r;
  r.X = typename CV::EX(q.x);
  r.Y = typename CV::EY(q.y);
  r.Z = is_inf(q) ? typename CV::EZ(el_zero(q.x)) : typename CV::EZ(el_one(q.x));
  return r;
}

template <class CV>
OZK_HD Jac<CV> jac_infinity() {
  Jac<CV> r;
  r.X = typename CV::EX(el_zero(r.X));
  r.Y = typename CV::EY(el_one(r.X));
  r.Z = typename CV::EZ(el_zero(r.X));
  return r;
}

// dbl-2009-l, a = 0.  2M + 5S.  (Not on the bucket hot path: used by the window
// combine, the fixed-base table build and the P == Q case of add.)
template <class CV>
OZK_HD Jac<CV> jac_dbl(const Jac<CV>& p) {
  const auto X1 = reduce_to<32>(p.X), Y1 = reduce_to<32>(p.Y), Z1 = reduce_to<32>(p.Z);
  const auto A = sqr(X1);
  const auto B = sqr(Y1);
  const auto CC = sqr(B);
  const auto t = reduce_to<32>(sub(sqr(add(X1, B)), add(A, CC)));
  const auto D = dbl(t);                                  // 2*((X1+B)^2 - A - C)
  const auto E = add(dbl(A), A);                          // 3*A
  const auto F = sqr(E);
  const auto X3 = reduce_to<32>(sub(F, dbl(D)));          // F - 2*D
  const auto C8 = dbl(dbl(dbl(CC)));
  const auto Y3 = sub(mul(E, reduce_to<48>(sub(D, X3))), C8);
  const auto Z3 = dbl(mul(Y1, Z1));
  Jac<CV> r;
  r.X = typename CV::EX(X3);
  r.Y = typename CV::EY(reduce_to<64>(Y3));
  r.Z = typename CV::EZ(Z3);
  return r;
}

// doubling of an affine point (Z1 = 1): mdbl-2007-bl.  1M + 5S.
template <class CV>
OZK_HD Jac<CV> aff_dbl(const Aff<typename CV::EA>& q) {
  const auto XX = sqr(q.x);
  const auto YY = sqr(q.y);
  const auto YYYY = sqr(YY);
  const auto S = dbl(reduce_to<32>(sub(sqr(add(q.x, YY)), add(XX, YYYY))));
  const auto M = add(dbl(XX), XX);
  const auto T = reduce_to<32>(sub(sqr(M), dbl(S)));
  const auto Y3 = sub(mul(M, reduce_to<48>(sub(S, T))), dbl(dbl(dbl(YYYY))));
  Jac<CV> r;
  r.X = typename CV::EX(T);
  r.Y = typename CV::EY(reduce_to<64>(Y3));
  r.Z = typename CV::EZ(dbl(q.y));
  return r;
}

// madd-2007-bl: Jacobian + affine.  7M + 4S.  THE hot operation of bucket accumulation.
template <class CV>
OZK_HD Jac<CV> jac_madd(const Jac<CV>& p, const Aff<typename CV::EA>& q) {
  if (is_inf(q)) return p;
  if (is_inf(p)) return from_affine<CV>(q);
  const auto Z1Z1 = sqr(p.Z);
  const auto U2 = mul(q.x, Z1Z1);
  const auto S2 = mul(mul(q.y, p.Z), Z1Z1);
  const auto H = sub(U2, p.X);
  const auto rh = sub(S2, p.Y);
  if (is_zero(H)) {
    if (is_zero(rh)) return aff_dbl<CV>(q);  // P == Q  (BNG1.java:76-79)
    return jac_infinity<CV>();               // P == -Q
  }
  const auto HH = sqr(H);
  const auto I = dbl(dbl(HH));
  const auto J = mul(H, I);
  const auto r = dbl(rh);
  const auto V = mul(p.X, I);
  const auto X3 = sub(sqr(r), add(J, dbl(V)));
  const auto Y3 = sub(mul(r, sub(V, X3)), dbl(mul(p.Y, J)));
  const auto Z3 = sub(sqr(add(p.Z, H)), add(Z1Z1, HH));
  Jac<CV> out;
  out.X = typename CV::EX(X3);
  out.Y = typename CV::EY(Y3);
  out.Z = typename CV::EZ(Z3);
  return out;
}

// add-2007-bl: Jacobian + Jacobian.  11M + 5S.
template <class CV>
OZK_HD Jac<CV> jac_add(const Jac<CV>& p, const Jac<CV>& q) {
  if (is_inf(p)) return q;
  if (is_inf(q)) return p;
  const auto X1 = reduce_to<48>(p.X), Y1 = reduce_to<48>(p.Y), Z1 = reduce_to<48>(p.Z);
  const auto X2 = reduce_to<48>(q.X), Y2 = reduce_to<48>(q.Y), Z2 = reduce_to<48>(q.Z);
  const auto Z1Z1 = sqr(Z1);
  const auto Z2Z2 = sqr(Z2);
  const auto U1 = mul(X1, Z2Z2);
  const auto U2 = mul(X2, Z1Z1);
  const auto S1 = mul(mul(Y1, Z2), Z2Z2);
  const auto S2 = mul(mul(Y2, Z1), Z1Z1);
  const auto H = sub(U2, U1);
  const auto rh = sub(S2, S1);
  if (is_zero(H)) {
    if (is_zero(rh)) return jac_dbl(p);
    return jac_infinity<CV>();
  }
  const auto I = sqr(dbl(H));
  const auto J = mul(H, I);
  const auto r = dbl(rh);
  const auto V = mul(U1, I);
  const auto X3 = sub(sqr(r), add(J, dbl(V)));
  const auto Y3 = sub(mul(r, sub(V, X3)), dbl(mul(S1, J)));
  const auto Z3 = mul(sub(sqr(add(Z1, Z2)), add(Z1Z1, Z2Z2)), H);
  Jac<CV> out;
  out.X = typename CV::EX(X3);
  out.Y = typename CV::EY(Y3);
  out.Z = typename CV::EZ(Z3);
  return out;
}

template <class CV>
OZK_HD Jac<CV> jac_neg(const Jac<CV>& p) {
  Jac<CV> r = p;
  r.Y = typename CV::EY(reduce_to<32>(neg(reduce_to<32>(p.Y))));
  return r;
}

// G1 over Fq: loop-carried bounds = fixed point of jac_madd (tools/bounds_fixpoint.py)
struct G1Cfg {
  using EX = Fe<FqParams, 94>;
  using EY = Fe<FqParams, 73>;
  using EZ = Fe<FqParams, 78>;
  using EA = Fe<FqParams, 17>;
};

}  // namespace ozk
