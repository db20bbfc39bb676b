// TEST INFRASTRUCTURE: compiles the device arithmetic headers (fp29.cuh, ec.cuh) for the
// HOST with g++ so the field / group-law code (and its compile-time bound proofs) can be
// unit-tested against the oracle without a GPU.  Not linked into the product library.
#include "../../octopuszk_amd/csrc/ec.cuh"
#include <string.h>
using namespace ozk;


template <class P>
static void fe_binop(int op, const u32* a, const u32* b, u32* out) {
  u32 wa[8], wb[8], wo[8];
  memcpy(wa, a, 32); memcpy(wb, b, 32);
  auto x = to_mont<P>(wa); auto y = to_mont<P>(wb);
  switch (op) {
    case 0: from_mont(mul(x, y), wo); break;
    case 1: from_mont(sqr(x), wo); break;
    case 2: from_mont(add(x, y), wo); break;
    case 3: from_mont(sub(x, y), wo); break;
    case 4: from_mont(inv(x), wo); break;
    case 9: from_mont(inv_fermat(x), wo); break;
    case 10: { auto big = sub(dbl(dbl(dbl(x))), y); from_mont(inv(big), wo); break; }  // non-canonical, bound > 128
    case 5: from_mont(neg(x), wo); break;
    case 6: from_mont(dbl(dbl(dbl(x))), wo); break;
    case 7: { auto t = sub(sub(sub(x, y), y), y); from_mont(reduce_to<32>(t), wo); break; }
    case 8: { u32 t[8]; pack(canonical(x), t); auto z = unpack<P, 16>(t); from_mont(z, wo); break; }
    // ---- loose (un-normalised) elements, fp29.cuh FeL: every result must equal the carried computation
    case 20: from_mont(mul(sub_nc(x, y), y), wo); break;                                   // (x - y) y
    case 21: from_mont(mul(select_el(true, neg_nc(x), x), y), wo); break;                  // (-x) y
    case 22: from_mont(sub_sub2(sqr(x), mul(x, y), sqr(y)), wo); break;                    // x^2 - x y - 2 y^2
    case 23: from_mont(mul_ll(add_nc(loose(x), loose(y)), sub_nc(x, y)), wo); break;       // x^2 - y^2
    case 24: from_mont(mul2_ll(loose(x), loose(y), neg_nc(y), loose(x)), wo); break;       // x y - y x = 0
    case 25: from_mont(mul2(sub_nc(x, y), x, neg_nc(y), y), wo); break;                    // (x - y) x - y^2
    case 26: from_mont(normalise(add_nc(dbl_nc(sub_nc(x, y)), loose(y))), wo); break;      // 2 (x - y) + y
    case 27: {                                                                             // four loose products
      const auto r = mul4_ll(loose(x), loose(y), neg_nc(x), loose(y), neg_nc(y), loose(x), loose(y), loose(x));
      from_mont(r, wo);                                                                    // x y - x y - y x + y x = 0
      break;
    }
    default: memset(wo, 0, 32);
  }
  memcpy(out, wo, 32);
}

// reduce_q on k * a (k = 1, 2, 4, 6, 7: value bounds 85 .. 595), raw words out: must be < 17 p / 16 and = k a (mod p)
template <class P>
static void fe_reduce_q(int k, const u32* a, u32* out) {
  u32 wa[8], wo[8];
  memcpy(wa, a, 32);
  const auto x = unpack<P, 85>(wa);
  const auto x2 = add(x, x);
  const auto x4 = add(x2, x2);
  switch (k) {
    case 1: pack(reduce_q(x), wo); break;
    case 2: pack(reduce_q(x2), wo); break;
    case 4: pack(reduce_q(x4), wo); break;
    case 6: pack(reduce_q(add(x4, x2)), wo); break;
    case 7: pack(reduce_q(add(add(x4, x2), x)), wo); break;
    default: memset(wo, 0, 32);
  }
  memcpy(out, wo, 32);
}
extern "C" void hc_fq_reduce_q(int k, const u32* a, u32* out) { fe_reduce_q<FqParams>(k, a, out); }
extern "C" void hc_fr_reduce_q(int k, const u32* a, u32* out) { fe_reduce_q<FrParams>(k, a, out); }

extern "C" void hc_fq_op(int op, const u32* a, const u32* b, u32* out) { fe_binop<FqParams>(op, a, b, out); }
extern "C" void hc_fr_op(int op, const u32* a, const u32* b, u32* out) { fe_binop<FrParams>(op, a, b, out); }
extern "C" int hc_fq_is_zero_after_sub(const u32* a, const u32* b) {
  u32 wa[8], wb[8]; memcpy(wa, a, 32); memcpy(wb, b, 32);
  return eq(to_mont<FqParams>(wa), to_mont<FqParams>(wb)) ? 1 : 0;
}

typedef G1Cfg J1;
typedef G1Cfg::EA A1;

static Jac<J1> load_jac(const u32* w) {  // wire X|Y|Z, 8 words each, canonical
  u32 t[8]; Jac<J1> p;
  memcpy(t, w, 32); p.X = to_mont<FqParams>(t);
  memcpy(t, w + 8, 32); p.Y = to_mont<FqParams>(t);
  memcpy(t, w + 16, 32); p.Z = to_mont<FqParams>(t);
  return p;
}
static void store_jac(const Jac<J1>& p, u32* w) {
  u32 t[8];
  from_mont(p.X, t); memcpy(w, t, 32);
  from_mont(p.Y, t); memcpy(w + 8, t, 32);
  from_mont(p.Z, t); memcpy(w + 16, t, 32);
}
// op 0: add(P,Q) 1: dbl(P) 2: madd(P, affine(Q.x,Q.y))  3: chain: P + k*Q via repeated madd
extern "C" void hc_g1_op(int op, const u32* pw, const u32* qw, int k, u32* out) {
  Jac<J1> p = load_jac(pw), q = load_jac(qw), r;
  Aff<A1> qa; qa.x = A1(reduce_to<17>(q.X)); qa.y = A1(reduce_to<17>(q.Y));
  if (is_zero(q.Z)) { qa.x = A1(el_zero(qa.x)); qa.y = A1(el_zero(qa.x)); }  // (0, 0) encodes infinity
  switch (op) {
    case 0: r = jac_add(p, q); break;
    case 1: r = jac_dbl(p); break;
    case 2: r = jac_madd(p, qa); break;
    case 3: r = p; for (int i = 0; i < k; i++) r = jac_madd(r, qa); break;
    case 4: r = p; for (int i = 0; i < k; i++) r = jac_add(r, q); break;
    case 5: r = p; for (int i = 0; i < k; i++) r = jac_dbl(r); break;
    case 7: {  // XYZZ accumulator: affine(P) then k mixed additions of affine(Q), back to Jacobian
      Aff<A1> pa; pa.x = A1(reduce_to<17>(p.X)); pa.y = A1(reduce_to<17>(p.Y));
      if (is_zero(p.Z)) { pa.x = A1(el_zero(pa.x)); pa.y = A1(el_zero(pa.x)); }
      Xyzz<J1> a = xyzz_from_affine<J1>(pa);
      for (int i = 0; i < k; i++) a = xyzz_madd(a, qa);
      r = xyzz_to_jac(a);
      break;
    }
    case 8: case 9: case 10: case 11: {  // general XYZZ addition: A = P + kQ, B = Q + kP (both with ZZ != 1)
      Aff<A1> pa; pa.x = A1(reduce_to<17>(p.X)); pa.y = A1(reduce_to<17>(p.Y));
      Xyzz<J1> a = xyzz_from_affine<J1>(pa), b = xyzz_from_affine<J1>(qa);
      for (int i = 0; i < k; i++) { a = xyzz_madd(a, qa); b = xyzz_madd(b, pa); }
      Xyzz<J1> z = a; z.ZZ = J1::XZZ(el_zero(pa.x)); z.ZZZ = J1::XZZZ(el_zero(pa.x));
      Xyzz<J1> na = a; na.Y = J1::XY(reduce_to<32>(neg(reduce_to<32>(a.Y))));
      if (op == 8) r = xyzz_to_jac(xyzz_add(a, b));                 // (k+1)(P+Q)
      else if (op == 9) r = xyzz_to_jac(xyzz_add(a, a));            // doubling branch: 2(P+kQ)
      else if (op == 10) r = xyzz_to_jac(xyzz_add(a, na));          // infinity
      else r = xyzz_to_jac(xyzz_add(xyzz_add(z, a), z));            // infinity on either side: P+kQ
      break;
    }
    case 12: case 13: case 14: {  // the level-1 form: lazily carried mixed additions with the digit's sign
      Aff<A1> pa; pa.x = A1(reduce_to<17>(p.X)); pa.y = A1(reduce_to<17>(p.Y));
      if (is_zero(p.Z)) { pa.x = A1(el_zero(pa.x)); pa.y = A1(el_zero(pa.x)); }
      Xyzz<J1> a = xyzz_from_affine<J1>(pa);   // op 12: P + kQ, 13: P - kQ, 14: P + Q - Q + Q ...
      for (int i = 0; i < k; i++) a = xyzz_madd_lazy(a, qa, op == 13 || (op == 14 && (i & 1)));
      r = xyzz_to_jac(a);
      break;
    }
    default: r = p;
  }
  store_jac(r, out);
}

// ---------------------------------------------------------------- Fq2 / G2
#include "../../octopuszk_amd/csrc/fq2.cuh"
extern "C" void hc_fq2_op(int op, const u32* a, const u32* b, u32* out) {  // 16 words each: c0|c1
  using ET = ElemTraits<Fe2<17>>;
  Fe2<17> x = ET::from_wire(a), y = ET::from_wire(b);
  switch (op) {
    case 0: ElemTraits<Fe2<32>>::to_wire(mul(x, y), out); break;
    case 1: ElemTraits<Fe2<32>>::to_wire(sqr(x), out); break;
    case 2: ElemTraits<Fe2<34>>::to_wire(add(x, y), out); break;
    case 3: ElemTraits<Fe2<49>>::to_wire(sub(x, y), out); break;
    case 4: ElemTraits<Fe2<32>>::to_wire(inv(x), out); break;
    case 5: { auto big = sub(dbl(dbl(x)), y); ElemTraits<Fe2<32>>::to_wire(mul(big, big), out); break; }
    case 6: {  // a b - c d as two four-term product sums, with unreduced inputs
      auto big = sub(dbl(dbl(x)), y);
      auto r = mulsub(x, big, add(x, y), y);
      ElemTraits<Fe2<32>>::to_wire(Fe2<32>(reduce_to<32>(r)), out);
      break;
    }
    // ---- the lazily carried forms (what the G2 level-1 addition uses)
    case 7: ElemTraits<Fe2<32>>::to_wire(mul_lz(x, y), out); break;
    case 8: ElemTraits<Fe2<32>>::to_wire(sqr_lz(x), out); break;
    case 9: {  // a b - c d with unreduced (< 2p, < 3p) inputs, as accumulate_q feeds it
      auto b2 = reduce_to<32>(sub(x, y));
      auto r = mulsub_lz(reduce_to<32>(add(x, y)), b2, Fe2<80>(sub(dbl(x), y)), y);
      ElemTraits<Fe2<32>>::to_wire(Fe2<32>(reduce_to<32>(r)), out);
      break;
    }
    case 10: ElemTraits<Fe2<32>>::to_wire(Fe2<32>(reduce_to<32>(sub_sub2(sqr(x), mul(x, y), sqr(y)))), out); break;
    default: break;
  }
}
typedef G2Cfg J2;
extern "C" void hc_g2_op(int op, const u32* pw, const u32* qw, int k, u32* out) {  // 48 words per point
  using IO = CurveIO<J2>;
  Jac<J2> p = IO::jac_from_wire(pw), q = IO::jac_from_wire(qw), r;
  Aff<J2::EA> qa; qa.x = J2::EA(reduce_to<17>(q.X)); qa.y = J2::EA(reduce_to<17>(q.Y));
  if (is_zero(q.Z)) { qa.x = J2::EA(el_zero(qa.x)); qa.y = J2::EA(el_zero(qa.x)); }
  switch (op) {
    case 0: r = jac_add(p, q); break;
    case 1: r = jac_dbl(p); break;
    case 2: r = jac_madd(p, qa); break;
    case 3: r = p; for (int i = 0; i < k; i++) r = jac_madd(r, qa); break;
    case 4: r = p; for (int i = 0; i < k; i++) r = jac_add(r, q); break;
    case 5: r = p; for (int i = 0; i < k; i++) r = jac_dbl(r); break;
    case 7: {
      Aff<J2::EA> pa; pa.x = J2::EA(reduce_to<17>(p.X)); pa.y = J2::EA(reduce_to<17>(p.Y));
      if (is_zero(p.Z)) { pa.x = J2::EA(el_zero(pa.x)); pa.y = J2::EA(el_zero(pa.x)); }
      Xyzz<J2> a = xyzz_from_affine<J2>(pa);
      for (int i = 0; i < k; i++) a = xyzz_madd(a, qa);
      r = xyzz_to_jac(a);
      break;
    }
    case 8: case 9: case 10: case 11: {  // general XYZZ addition: A = P + kQ, B = Q + kP (both with ZZ != 1)
      Aff<J2::EA> pa; pa.x = J2::EA(reduce_to<17>(p.X)); pa.y = J2::EA(reduce_to<17>(p.Y));
      Xyzz<J2> a = xyzz_from_affine<J2>(pa), b = xyzz_from_affine<J2>(qa);
      for (int i = 0; i < k; i++) { a = xyzz_madd(a, qa); b = xyzz_madd(b, pa); }
      Xyzz<J2> z = a; z.ZZ = J2::XZZ(el_zero(pa.x)); z.ZZZ = J2::XZZZ(el_zero(pa.x));
      Xyzz<J2> na = a; na.Y = J2::XY(reduce_to<32>(neg(reduce_to<32>(a.Y))));
      if (op == 8) r = xyzz_to_jac(xyzz_add(a, b));                 // (k+1)(P+Q)
      else if (op == 9) r = xyzz_to_jac(xyzz_add(a, a));            // doubling branch: 2(P+kQ)
      else if (op == 10) r = xyzz_to_jac(xyzz_add(a, na));          // infinity
      else r = xyzz_to_jac(xyzz_add(xyzz_add(z, a), z));            // infinity on either side: P+kQ
      break;
    }
    case 12: {  // the schedule of RunAccLds::accumulate_q (msm_var.cuh) over the lazily carried Fq2 products, k times
      Aff<J2::EA> pa; pa.x = J2::EA(reduce_to<17>(p.X)); pa.y = J2::EA(reduce_to<17>(p.Y));
      if (is_zero(p.Z)) { pa.x = J2::EA(el_zero(pa.x)); pa.y = J2::EA(el_zero(pa.x)); }
      Xyzz<J2> a = xyzz_from_affine<J2>(pa);
      for (int i = 0; i < k; i++) {
        if (is_inf(qa)) continue;
        if (is_zero(a.ZZ)) { a = xyzz_from_affine<J2>(qa); continue; }
        const auto U2 = mul_lz(qa.x, a.ZZ);
        const auto P = sub(U2, a.X);
        const auto R = sub(mul_lz(qa.y, a.ZZZ), a.Y);
        if (is_zero(P)) {
          if (is_zero(R)) { a = xyzz_dbl_affine<J2>(qa); }
          else { a.ZZ = J2::XZZ(el_zero(qa.x)); a.ZZZ = J2::XZZZ(el_zero(qa.x)); }
          continue;
        }
        const auto Pr = reduce_to<32>(P);
        const auto PP = sqr_lz(Pr);
        const auto PPP = mul_lz(Pr, PP);
        const auto ZZn = mul_lz(a.ZZ, PP), ZZZn = mul_lz(a.ZZZ, PPP);
        const auto Q = mul_lz(reduce_to<32>(a.X), PP);
        const auto X3 = sub_sub2(sqr_lz(reduce_to<32>(R)), PPP, Q);
        const auto Y3 = mulsub_lz(reduce_to<32>(R), reduce_to<32>(sub(Q, X3)), a.Y, PPP);
        a.X = J2::XX(X3); a.Y = J2::XY(Y3); a.ZZ = J2::XZZ(ZZn); a.ZZZ = J2::XZZZ(ZZZn);
      }
      r = xyzz_to_jac(a);
      break;
    }
    default: r = p;
  }
  ElemTraits<J2::EX>::to_wire(r.X, out);
  ElemTraits<J2::EY>::to_wire(r.Y, out + 16);
  ElemTraits<J2::EZ>::to_wire(r.Z, out + 32);
}

// ---- GLV decomposition (glv.cuh) ----
#include "../../octopuszk_amd/csrc/glv.cuh"
// out[0..4) = |k1|, out[4..8) = |k2|, out[8] = neg1, out[9] = neg2
extern "C" void hc_glv(const u32* k, u32* out) {
  u32 kk[8], k1[4], k2[4];
  bool n1, n2;
  memcpy(kk, k, 32);
  glv_decompose(kk, k1, n1, k2, n2);
  memcpy(out, k1, 16);
  memcpy(out + 4, k2, 16);
  out[8] = n1;
  out[9] = n2;
}
// beta constants out of Montgomery form: out[0..8) = beta_G1, out[8..16) = beta_G2
extern "C" void hc_glv_beta(u32* out) {
  u32 w[8];
  from_mont(fe_const<FqParams, 16>(GlvConsts::BETA_G1), w);
  memcpy(out, w, 32);
  from_mont(fe_const<FqParams, 16>(GlvConsts::BETA_G2), w);
  memcpy(out + 8, w, 32);
}

// ---- signed digit recoding (msm_var.cuh): codes of one 128-bit half scalar
#include "../../octopuszk_amd/csrc/msm_var.cuh"
extern "C" void hc_signed_digits(const u32* k4, int c, int W, int neg, uint16_t* codes) {
  u32 e[8] = {k4[0], k4[1], k4[2], k4[3], 0, 0, 0, 0};
  signed_digit_codes(e, c, W, neg != 0, [&](int w, uint16_t code) { codes[w] = code; });
}
