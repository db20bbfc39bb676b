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
// Over the base field (CV::WIDE_INPUTS) the coordinates go into the products as they are — the multiplier's input
// slack (B1 B2 <= 169 x 256 sixteenths of p squared) takes 96 x 96 and 156 x 156 — and only X3, Y3 are brought back
// under the loop-carried bounds: 4 conditional subtractions per doubling instead of 15, none instead of 6 per
// addition.  Over Fq2 the squaring wants its input below 2p, so the entry reductions stay.
template <int TB, class CV, class E>
OZK_HD auto entry_reduce(const E& a) {
  if constexpr (CV::WIDE_INPUTS) return a;
  else return reduce_to<TB>(a);
}
template <class CV>
OZK_HD Jac<CV> jac_dbl(const Jac<CV>& p) {
  if constexpr (CV::WIDE_INPUTS) {
    const auto A = sqr(p.X);
    const auto B = sqr(p.Y);
    const auto CC = sqr(B);
    const auto D = dbl(sub(sqr(add(p.X, B)), add(A, CC)));  // 2*((X1+B)^2 - A - C)
    const auto E = add(dbl(A), A);                          // 3*A
    const auto X3 = reduce_to<80>(sub(sqr(E), dbl(D)));     // F - 2*D
    const auto C8 = dbl(dbl(dbl(CC)));
    const auto Y3 = sub(mul(E, sub(D, X3)), C8);
    Jac<CV> r;
    r.X = typename CV::EX(X3);
    r.Y = typename CV::EY(reduce_to<64>(Y3));
    r.Z = typename CV::EZ(dbl(mul(p.Y, p.Z)));
    return r;
  }
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
  const auto Y3 = mulsub(r, sub(V, X3), dbl(p.Y), J);
  const auto Z3 = sub(sqr(add(p.Z, H)), add(Z1Z1, HH));
  Jac<CV> out;
  out.X = typename CV::EX(X3);
  out.Y = typename CV::EY(Y3);
  out.Z = typename CV::EZ(Z3);
  return out;
}

// ---------------------------------------------------------------------------------------
// XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) for the bucket-accumulation hot loop:
// the mixed addition madd-2008-s is 8M + 2S = 1638 MADs against 7M + 4S = 1737 for
// madd-2007-bl, with 7 instead of 13 limb-wise add/sub passes, and its loop-carried bounds
// (X < 83/16 p, Y < 52/16 p, ZZ, ZZZ < 17/16 p) again need no conditional subtraction.
// A finished run converts to Jacobian (X*ZZ, Y*ZZZ, ZZ) with two multiplications.
template <class CV>
struct Xyzz {
  typename CV::XX X;
  typename CV::XY Y;
  typename CV::XZZ ZZ;
  typename CV::XZZZ ZZZ;
};

template <class CV>
OZK_HD bool is_inf(const Xyzz<CV>& p) { return is_zero(p.ZZ); }

template <class CV>
OZK_HD Xyzz<CV> xyzz_from_affine(const Aff<typename CV::EA>& q) {
  Xyzz<CV> r;
  r.X = typename CV::XX(q.x);
  r.Y = typename CV::XY(q.y);
  if (is_inf(q)) {
    r.ZZ = typename CV::XZZ(el_zero(q.x));
    r.ZZZ = typename CV::XZZZ(el_zero(q.x));
  } else {
    r.ZZ = typename CV::XZZ(el_one(q.x));
    r.ZZZ = typename CV::XZZZ(el_one(q.x));
  }
  return r;
}

// doubling of an affine point into XYZZ: mdbl-2008-s-1.  2M + 4S... (U = 2y, V = U^2, W = U V,
// S = x V, M = 3 x^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W y, ZZ3 = V, ZZZ3 = W)
template <class CV>
OZK_HD Xyzz<CV> xyzz_dbl_affine(const Aff<typename CV::EA>& q) {
  const auto U = dbl(q.y);
  const auto V = sqr(U);
  const auto W = mul(U, V);
  const auto S = mul(q.x, V);
  const auto xx = sqr(q.x);
  const auto M = add(dbl(xx), xx);
  const auto X3 = sub(sqr(M), dbl(S));
  const auto Y3 = mulsub(M, sub(S, X3), W, q.y);
  Xyzz<CV> r;
  r.X = typename CV::XX(X3);
  r.Y = typename CV::XY(Y3);
  r.ZZ = typename CV::XZZ(V);
  r.ZZZ = typename CV::XZZZ(W);
  return r;
}

// madd-2008-s: XYZZ + affine.  8M + 2S.
template <class CV>
OZK_HD Xyzz<CV> xyzz_madd(const Xyzz<CV>& p, const Aff<typename CV::EA>& q) {
  if (is_inf(q)) return p;
  if (is_inf(p)) return xyzz_from_affine<CV>(q);
  const auto U2 = mul(q.x, p.ZZ);
  const auto S2 = mul(q.y, p.ZZZ);
  const auto P = sub(U2, p.X);
  const auto R = sub(S2, p.Y);
  if (is_zero(P)) {
    if (is_zero(R)) return xyzz_dbl_affine<CV>(q);  // P == Q
    Xyzz<CV> z = p;                                  // P == -Q: infinity
    z.ZZ = typename CV::XZZ(el_zero(q.x));
    z.ZZZ = typename CV::XZZZ(el_zero(q.x));
    return z;
  }
  const auto PP = sqr(P);
  const auto PPP = mul(P, PP);
  const auto Q = mul(p.X, PP);
  const auto X3 = sub(sqr(R), add(PPP, dbl(Q)));
  const auto Y3 = mulsub(R, sub(Q, X3), p.Y, PPP);
  Xyzz<CV> out;
  out.X = typename CV::XX(X3);
  out.Y = typename CV::XY(Y3);
  out.ZZ = typename CV::XZZ(mul(p.ZZ, PP));
  out.ZZZ = typename CV::XZZZ(mul(p.ZZZ, PPP));
  return out;
}

// The same madd-2008-s with the point SUBTRACTED when `negate` is set, carries left out wherever the consumer is a
// multiplication (fp29.cuh FeL), base field only: three carry passes (P, R: they are squared; X3: it is stored)
// instead of seven, and the negated y enters its one product as K p - y without reduction or carry.  The level-1
// loop of the bucket accumulation is this function; loop-carried bounds CV::LX / LY (the fixed point of THIS
// schedule: X3 collects 2 p + 4 p of bias instead of 4 p).
template <class CV>
OZK_HD Xyzz<CV> xyzz_madd_lazy(const Xyzz<CV>& p, const Aff<typename CV::EA>& q, bool negate) {
  if (is_inf(q)) return p;
  const auto yq = select_el(negate, neg_nc(q.y), q.y);  // +-y, loose
  if (is_inf(p)) {
    Aff<typename CV::EA> qs = q;
    qs.y = typename CV::EA(reduce_to<17>(normalise(yq)));
    return xyzz_from_affine<CV>(qs);
  }
  const auto U2 = mul(q.x, p.ZZ);
  const auto S2 = mul(yq, p.ZZZ);
  const auto P = sub(U2, p.X);
  const auto R = sub(S2, p.Y);
  // P == 0 is tested on P^2 (< 2p: two candidates to compare with, not the ten of P < 9.1 p)
  const auto PP = sqr(P);
  if (is_zero(PP)) {
    if (is_zero(R)) {  // P == +-Q
      Aff<typename CV::EA> qs = q;
      qs.y = typename CV::EA(reduce_to<17>(normalise(yq)));
      return xyzz_dbl_affine<CV>(qs);
    }
    Xyzz<CV> z = p;  // P == -(+-Q): infinity
    z.ZZ = typename CV::XZZ(el_zero(q.x));
    z.ZZZ = typename CV::XZZZ(el_zero(q.x));
    return z;
  }
  const auto PPP = mul(P, PP);
  const auto Q = mul(p.X, PP);
  const auto X3 = sub_sub2(sqr(R), PPP, Q);
  const auto Y3 = mul2(sub_nc(Q, X3), R, neg_nc(p.Y), PPP);
  Xyzz<CV> out;
  out.X = typename CV::XX(X3);
  out.Y = typename CV::XY(Y3);
  out.ZZ = typename CV::XZZ(mul(p.ZZ, PP));
  out.ZZZ = typename CV::XZZZ(mul(p.ZZZ, PPP));
  return out;
}

// dbl-2008-s-1: XYZZ doubling (a = 0).  6M + 4S... used only when a general addition meets P == Q.
template <class CV>
OZK_HD Xyzz<CV> xyzz_dbl(const Xyzz<CV>& p) {
  if (is_inf(p)) return p;
  const auto U = dbl(p.Y);
  const auto V = sqr(U);
  const auto W = mul(U, V);
  const auto S = mul(p.X, V);
  const auto xx = sqr(p.X);
  const auto M = add(dbl(xx), xx);
  const auto X3 = sub(sqr(M), dbl(S));
  const auto Y3 = mulsub(M, sub(S, X3), W, p.Y);
  Xyzz<CV> r;
  r.X = typename CV::XX(X3);
  r.Y = typename CV::XY(Y3);
  r.ZZ = typename CV::XZZ(mul(V, p.ZZ));
  r.ZZZ = typename CV::XZZZ(mul(W, p.ZZZ));
  return r;
}

// add-2008-s: XYZZ + XYZZ.  12M + 2S (the Jacobian add-2007-bl is 11M + 5S, plus 2M per XYZZ input
// converted first), complete: handles infinity on either side, P == Q and P == -Q.
template <class CV>
OZK_HD Xyzz<CV> xyzz_add(const Xyzz<CV>& p, const Xyzz<CV>& q) {
  if (is_inf(q)) return p;
  if (is_inf(p)) return q;
  const auto U1 = mul(p.X, q.ZZ);
  const auto U2 = mul(q.X, p.ZZ);
  const auto S1 = mul(p.Y, q.ZZZ);
  const auto S2 = mul(q.Y, p.ZZZ);
  const auto P = sub(U2, U1);
  const auto R = sub(S2, S1);
  if (is_zero(P)) {
    if (is_zero(R)) return xyzz_dbl(p);
    Xyzz<CV> z = p;
    z.ZZ = typename CV::XZZ(el_zero(p.ZZ));
    z.ZZZ = typename CV::XZZZ(el_zero(p.ZZ));
    return z;
  }
  const auto PP = sqr(P);
  const auto PPP = mul(P, PP);
  const auto Q = mul(U1, PP);
  const auto X3 = sub(sqr(R), add(PPP, dbl(Q)));
  const auto Y3 = mulsub(R, sub(Q, X3), S1, PPP);
  Xyzz<CV> out;
  out.X = typename CV::XX(X3);
  out.Y = typename CV::XY(Y3);
  out.ZZ = typename CV::XZZ(mul(mul(p.ZZ, q.ZZ), PP));
  out.ZZZ = typename CV::XZZZ(mul(mul(p.ZZZ, q.ZZZ), PPP));
  return out;
}

// XYZZ -> Jacobian with Z = ZZ:  (X*ZZ, Y*ZZZ, ZZ)   [Y*ZZ^3/ZZZ = Y*ZZZ since ZZ^3 = ZZZ^2]
template <class CV>
OZK_HD Jac<CV> xyzz_to_jac(const Xyzz<CV>& p) {
  Jac<CV> r;
  r.X = typename CV::EX(mul(p.X, p.ZZ));
  r.Y = typename CV::EY(mul(p.Y, p.ZZZ));
  r.Z = typename CV::EZ(p.ZZ);
  return r;
}

// add-2007-bl: Jacobian + Jacobian.  11M + 5S.
template <class CV>
OZK_HD Jac<CV> jac_add(const Jac<CV>& p, const Jac<CV>& q) {
  if (is_inf(p)) return q;
  if (is_inf(q)) return p;
  // X and Y only ever meet a factor below 2p (Z^2, Z), which the lazy products of both fields take unreduced; Z is
  // squared and added: the base field takes that too, Fq2's squaring wants it below 2p
  const auto& X1 = p.X;
  const auto& Y1 = p.Y;
  const auto& X2 = q.X;
  const auto& Y2 = q.Y;
  const auto Z1 = entry_reduce<32, CV>(p.Z);
  const auto Z2 = entry_reduce<32, CV>(q.Z);
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
  const auto Y3 = mulsub(r, sub(V, X3), dbl(S1), J);
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
struct G1CfgQ;
struct G1Cfg {
  using Pair = G1CfgQ;                    // serial chains (Horner, fixed-base doubling chain) run on a lane QUAD
  static constexpr int PAIR_LANES = 4;    // (quad.cuh)
  static constexpr bool WIDE_INPUTS = true;   // jac_add / jac_dbl skip their entry reductions (base field)
  static constexpr bool LDS_ACC = false;  // level-1 accumulator in registers (137 VGPRs with the prefetched base, 3 waves per SIMD)
  static constexpr bool LAZY_MADD = true; // level 1 adds with xyzz_madd_lazy (carries only where a value is squared or stored)
  using EX = Fe<FqParams, 94>;
  using EY = Fe<FqParams, 73>;
  using EZ = Fe<FqParams, 78>;
  using EA = Fe<FqParams, 17>;
  // XYZZ accumulator of the hot loop: fixed point of xyzz_madd_lazy (X3 = R^2 + 2p - PPP + 2 (2p - Q) < 115/16 p; the
  // carried schedule of xyzz_madd / xyzz_add stays below 83/16 p from these inputs)
  using XX = Fe<FqParams, 115>;
  using XY = Fe<FqParams, 52>;
  using XZZ = Fe<FqParams, 17>;
  using XZZZ = Fe<FqParams, 17>;
};

}  // namespace ozk
