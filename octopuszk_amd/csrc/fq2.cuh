// Fq2 = Fq[u]/(u^2 + 1) for BN254 G2, on top of fp29.cuh.  One Fq2 element per lane.
//
// Restates algebra/fields/Fp2.java:44-107 (Karatsuba mul :59-72, complex squaring
// :94-103, inverse :105-115) with non-residue -1 applied as a subtraction — the
// reference's CUDA multiplies by q-1 with a full mul+rem (VariableBaseMSM.cu:229-232).
//
// Bounds: Fe2<B> has both components < B*p/16.  mul / sqr reduce their inputs to < 2p
// when the Karatsuba sums would break the Montgomery precondition and always return
// components < 2p, so the generic group law of ec.cuh type-checks unchanged.
#pragma once
#include "curve.cuh"

namespace ozk {

template <int B>
struct Fe2 {
  Fe<FqParams, B> c0, c1;
  OZK_HD Fe2() {}
  template <int B2, class = std::enable_if_t<(B2 <= B)>>
  OZK_HD Fe2(const Fe2<B2>& o) : c0(o.c0), c1(o.c1) {}
};

template <int B>
OZK_HD Fe2<1> el_zero(const Fe2<B>&) {
  Fe2<1> r;
  r.c0 = fe_zero<FqParams>();
  r.c1 = fe_zero<FqParams>();
  return r;
}
template <int B>
OZK_HD Fe2<16> el_one(const Fe2<B>&) {
  Fe2<16> r;
  r.c0 = fe_one<FqParams>();
  r.c1 = Fe<FqParams, 16>(fe_zero<FqParams>());
  return r;
}

template <int B1, int B2>
OZK_HD auto add(const Fe2<B1>& a, const Fe2<B2>& b) {
  Fe2<B1 + B2> r;
  r.c0 = add(a.c0, b.c0);
  r.c1 = add(a.c1, b.c1);
  return r;
}
template <int B1>
OZK_HD auto dbl(const Fe2<B1>& a) {
  Fe2<2 * B1> r;
  r.c0 = dbl(a.c0);
  r.c1 = dbl(a.c1);
  return r;
}
template <int B1, int B2>
OZK_HD auto sub(const Fe2<B1>& a, const Fe2<B2>& b) {
  Fe2<B1 + 16 * (B2 / 16 + 1)> r;
  r.c0 = sub(a.c0, b.c0);
  r.c1 = sub(a.c1, b.c1);
  return r;
}
template <int B2>
OZK_HD auto neg(const Fe2<B2>& b) {
  Fe2<16 * (B2 / 16 + 1)> r;
  r.c0 = neg(b.c0);
  r.c1 = neg(b.c1);
  return r;
}
template <int TB, int B>
OZK_HD auto reduce_to(const Fe2<B>& a) {
  if constexpr (B <= TB) {
    return a;
  } else {
    auto x = reduce_to<TB>(a.c0);
    auto y = reduce_to<TB>(a.c1);
    using T = decltype(x);
    Fe2<TB> r;
    r.c0 = Fe<FqParams, TB>(x);
    r.c1 = Fe<FqParams, TB>(y);
    (void)sizeof(T);
    return r;
  }
}
template <int B>
OZK_HD Fe2<16> canonical(const Fe2<B>& a) {
  Fe2<16> r;
  r.c0 = canonical(a.c0);
  r.c1 = canonical(a.c1);
  return r;
}
template <int B>
OZK_HD bool is_zero(const Fe2<B>& a) {  // Fp2.java:78-80
  return is_zero(a.c0) && is_zero(a.c1);
}

// bounds for which a0 b0 + (-a1) b1 (the larger of the two dual products) meets the Montgomery precondition
constexpr bool lazy_ok(int B1, int B2) {
  return (long long)B1 * B2 + 16LL * (B1 / 16 + 1) * B2 <= (long long)MONT_SLACK * 256;
}

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u.  Same values as Fp2.java:59-72 (which
// uses Karatsuba); here each component is ONE lazily reduced dual product (fp29.cuh mul2), which is
// fewer instructions than three single products plus Karatsuba's additions on this multiplier.
template <int B1, int B2>
OZK_HD Fe2<32> mul(const Fe2<B1>& a_in, const Fe2<B2>& b_in) {
  if constexpr (!lazy_ok(B1, B2)) {
    return mul(reduce_to<(B1 > 32 ? 32 : B1)>(a_in), reduce_to<(B2 > 32 ? 32 : B2)>(b_in));
  } else {
    Fe2<32> r;
    r.c0 = Fe<FqParams, 32>(mul2(a_in.c0, b_in.c0, neg(a_in.c1), b_in.c1));
    r.c1 = Fe<FqParams, 32>(mul2(a_in.c0, b_in.c1, a_in.c1, b_in.c0));
    return r;
  }
}
// a b - c d over Fq2: each component is one four-term product sum
//   c0 = a0 b0 + (-a1) b1 + (-c0) d0 + c1 d1,   c1 = a0 b1 + a1 b0 + (-c0) d1 + (-c1) d0
template <int B1, int B2, int B3, int B4>
OZK_HD auto mulsub(const Fe2<B1>& a, const Fe2<B2>& b, const Fe2<B3>& c, const Fe2<B4>& d) {
  constexpr long long N1 = 16LL * (B1 / 16 + 1), N3 = 16LL * (B3 / 16 + 1);
  constexpr long long BB0 = (long long)B1 * B2 + N1 * B2 + N3 * B4 + (long long)B3 * B4;
  constexpr long long BB1 = 2LL * B1 * B2 + 2 * N3 * B4;
  constexpr long long BB = BB0 > BB1 ? BB0 : BB1;
  if constexpr (BB <= (long long)MONT_SLACK * 256) {
    constexpr int BO = 16 + ceil_div(BB, 16 * MONT_SLACK);
    const auto na1 = neg(a.c1);
    const auto nc0 = neg(c.c0);
    const auto nc1 = neg(c.c1);
    Fe2<BO> r;
    r.c0 = Fe<FqParams, BO>(mul4(a.c0, b.c0, na1, b.c1, nc0, d.c0, c.c1, d.c1));
    r.c1 = Fe<FqParams, BO>(mul4(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0));
    return r;
  } else {
    return sub(mul(a, b), mul(c, d));
  }
}
// ---- the same products with the carry passes left out wherever the consumer is the multiplier (fp29.cuh FeL):
// -a1 and -c enter their products as K p - x without a carry, the sums and differences inside the squaring go
// straight into its one product, a - b - 2c takes one carry instead of three.  The level-1 mixed addition of G2
// (msm_var.cuh RunAccLds::accumulate_q) spends 27 carry passes per addition in the carried forms above, 8 in these.
template <int B1, int B2>
OZK_HD Fe2<32> mul_lz(const Fe2<B1>& a, const Fe2<B2>& b) {
  static_assert(lazy_ok(B1, B2), "reduce the factors first");
  Fe2<32> r;
  r.c0 = Fe<FqParams, 32>(mul2_ll(loose(a.c0), loose(b.c0), neg_nc(a.c1), loose(b.c1)));
  r.c1 = Fe<FqParams, 32>(mul2(a.c0, b.c1, a.c1, b.c0));
  return r;
}
template <int B1>
OZK_HD Fe2<32> sqr_lz(const Fe2<B1>& a) {
  static_assert(B1 <= 32, "reduce the argument first");
  const auto t = mul_ll(dbl_nc(loose(a.c0)), loose(a.c1));
  const auto d = mul_ll(add_nc(loose(a.c0), loose(a.c1)), sub_nc(a.c0, a.c1));
  Fe2<32> r;
  r.c0 = Fe<FqParams, 32>(reduce_to<32>(d));
  r.c1 = Fe<FqParams, 32>(reduce_to<32>(t));
  return r;
}
template <int B1, int B2, int B3, int B4>
OZK_HD auto mulsub_lz(const Fe2<B1>& a, const Fe2<B2>& b, const Fe2<B3>& c, const Fe2<B4>& d) {
  constexpr long long N1 = 16LL * (B1 / 16 + 1), N3 = 16LL * (B3 / 16 + 1);
  constexpr long long BB0 = (long long)B1 * B2 + N1 * B2 + N3 * B4 + (long long)B3 * B4;
  constexpr long long BB1 = 2LL * B1 * B2 + 2 * N3 * B4;
  constexpr long long BB = BB0 > BB1 ? BB0 : BB1;
  static_assert(BB <= (long long)MONT_SLACK * 256, "Montgomery input bounds too large");
  constexpr int BO = 16 + ceil_div(BB, 16 * MONT_SLACK);
  const auto na1 = neg_nc(a.c1);
  const auto nc0 = neg_nc(c.c0);
  const auto nc1 = neg_nc(c.c1);
  Fe2<BO> r;
  r.c0 = Fe<FqParams, BO>(mul4_ll(loose(a.c0), loose(b.c0), na1, loose(b.c1), nc0, loose(d.c0), loose(c.c1), loose(d.c1)));
  r.c1 = Fe<FqParams, BO>(mul4_ll(loose(a.c0), loose(b.c1), loose(a.c1), loose(b.c0), nc0, loose(d.c1), nc1, loose(d.c0)));
  return r;
}
template <int B1, int B2, int B3>
OZK_HD auto sub_sub2(const Fe2<B1>& a, const Fe2<B2>& b, const Fe2<B3>& c) {
  Fe2<B1 + 16 * (B2 / 16 + 1) + 32 * (B3 / 16 + 1)> r;
  r.c0 = sub_sub2(a.c0, b.c0, c.c0);
  r.c1 = sub_sub2(a.c1, b.c1, c.c1);
  return r;
}

template <int B>
OZK_HD Fe2<B> select_el(bool c, const Fe2<B>& a, const Fe2<B>& b) {
  Fe2<B> r;
  r.c0 = select_el(c, a.c0, b.c0);
  r.c1 = select_el(c, a.c1, b.c1);
  return r;
}
template <int B1>
OZK_HD Fe2<32> scale(const Fe2<B1>& a, const Fe<FqParams, 16>& k) {
  Fe2<32> r;
  r.c0 = Fe<FqParams, 32>(mul(a.c0, k));
  r.c1 = Fe<FqParams, 32>(mul(a.c1, k));
  return r;
}
// (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u        (Fp2.java:94-103 with nonresidue = -1)
template <int B1>
OZK_HD Fe2<32> sqr(const Fe2<B1>& a_in) {
  if constexpr (B1 > 32) {
    return sqr(reduce_to<32>(a_in));
  } else {
    const auto t = mul(dbl(a_in.c0), a_in.c1);  // doubling an input keeps the product < 2p: no conditional subtraction
    const auto d = mul(add(a_in.c0, a_in.c1), sub(a_in.c0, a_in.c1));
    Fe2<32> r;
    r.c0 = Fe<FqParams, 32>(reduce_to<32>(d));
    r.c1 = Fe<FqParams, 32>(reduce_to<32>(t));
    return r;
  }
}
// Fp2.java:105-115 (Algorithm 8): (a0 - a1 u) / (a0^2 + a1^2)
template <int B>
OZK_HD Fe2<32> inv(const Fe2<B>& a_in) {
  const auto a = reduce_to<32>(a_in);
  const auto t = add(sqr(a.c0), sqr(a.c1));
  const auto ti = inv(t);
  Fe2<32> r;
  r.c0 = Fe<FqParams, 32>(mul(a.c0, ti));
  r.c1 = Fe<FqParams, 32>(reduce_to<32>(neg(mul(a.c1, ti))));
  return r;
}

template <int B>
struct ElemTraits<Fe2<B>> {
  using E1 = ElemTraits<Fe<FqParams, B>>;
  static constexpr int WORDS = 16;
  static OZK_HD Fe2<B> load(const u32* p) {
    Fe2<B> r;
    r.c0 = E1::load(p);
    r.c1 = E1::load(p + 8);
    return r;
  }
  static OZK_HD void store(const Fe2<B>& e, u32* p) {
    E1::store(e.c0, p);
    E1::store(e.c1, p + 8);
  }
  static constexpr int RAW_WORDS = 18;
  static OZK_HD Fe2<B> load_raw(const u32* p) {
    Fe2<B> r;
    r.c0 = E1::load_raw(p);
    r.c1 = E1::load_raw(p + 9);
    return r;
  }
  static OZK_HD void store_raw(const Fe2<B>& e, u32* p) {
    E1::store_raw(e.c0, p);
    E1::store_raw(e.c1, p + 9);
  }
  static OZK_HD Fe2<B> from_wire(const u32* p) {
    Fe2<B> r;
    r.c0 = E1::from_wire(p);
    r.c1 = E1::from_wire(p + 8);
    return r;
  }
  static OZK_HD void to_wire(const Fe2<B>& e, u32* p) {
    E1::to_wire(e.c0, p);
    E1::to_wire(e.c1, p + 8);
  }
  // c0 (64 B) | c1 (64 B), VariableBaseMSM.cu:1781-1784 order Xa|Xb
  static OZK_HD void to_wire_out(const Fe2<B>& e, u32* p) {
    E1::to_wire_out(e.c0, p);
    E1::to_wire_out(e.c1, p + 16);
  }
  static OZK_HD Fe2<B> from_wire_out(const u32* p) {
    Fe2<B> r;
    r.c0 = E1::from_wire(p);
    r.c1 = E1::from_wire(p + 16);
    return r;
  }
  static OZK_HD bool wire_is_zero(const u32* p) { return E1::wire_is_zero(p) && E1::wire_is_zero(p + 8); }
  static OZK_HD bool wire_is_one(const u32* p) { return E1::wire_is_one(p) && E1::wire_is_zero(p + 8); }
};

#if defined(__HIPCC__)
template <int B>
__device__ __forceinline__ Fe2<B> shfl_down_el(const Fe2<B>& v, int o) {
  Fe2<B> r;
  r.c0 = shfl_down_el(v.c0, o);
  r.c1 = shfl_down_el(v.c1, o);
  return r;
}
#endif

// G2 over Fq2.  Loop-carried bounds: mul / sqr return < 2p, so the madd outputs are
// X3 = sqr - (J + 2V) < 2p + 7p, Y3 < 2p + 5p, Z3 < 2p + 5p.
struct G2CfgO;
struct G2Cfg {
  using Pair = G2CfgO;                    // serial chains (Horner, fixed-base doubling chain) on a lane OCTET
  static constexpr int PAIR_LANES = 8;    // (below; a lane PAIR until round 2)
  static constexpr bool WIDE_INPUTS = false;
  static constexpr bool LDS_ACC = true;   // level-1 accumulator in LDS (msm_var.cuh RunAccLds)
  static constexpr bool LAZY_MADD = false;
  static constexpr bool LAZY_FQ2 = true;  // RunAccLds::accumulate_q uses mul_lz / sqr_lz / mulsub_lz / sub_sub2
  using EX = Fe2<144>;
  using EY = Fe2<112>;
  using EZ = Fe2<112>;
  using EA = Fe2<17>;
  using XX = Fe2<176>;   // sqr - PPP - 2Q with its biases: < 2p + 3p + 6p (the carried form: 2p + 7p)
  using XY = Fe2<80>;    // mul - mul < 2p + 3p
  using XZZ = Fe2<32>;
  using XZZZ = Fe2<32>;
};

#if defined(__HIPCC__)
// ---- Fq2 on a LANE PAIR, for the serial chains -------------------------------------------------
// A lone lane runs a G2 doubling (5 Fq2 squarings + 2 Fq2 multiplications = 16 Fq multiplications) in
// ~20 us, and both the Horner kernel (120 of them) and the fixed-base doubling chain are exactly that.
// Fe2L is the same element held IDENTICALLY by two adjacent lanes (an even / odd pair): the two Fq
// products of a squaring, and two of the three of a Karatsuba multiplication, run side by side, one per
// lane, and are exchanged with one __shfl_xor per limb; everything else is computed redundantly, so the
// pair never diverges.
template <int B>
struct Fe2L {
  Fe<FqParams, B> c0, c1;
  __device__ __forceinline__ Fe2L() {}
  template <int B2, class = std::enable_if_t<(B2 <= B)>>
  __device__ __forceinline__ Fe2L(const Fe2L<B2>& o) : c0(o.c0), c1(o.c1) {}
};
template <int B>
__device__ __forceinline__ Fe2L<B> to_pair(const Fe2<B>& a) {
  Fe2L<B> r;
  r.c0 = a.c0;
  r.c1 = a.c1;
  return r;
}
template <int B>
__device__ __forceinline__ Fe2<B> from_pair(const Fe2L<B>& a) {
  Fe2<B> r;
  r.c0 = a.c0;
  r.c1 = a.c1;
  return r;
}
template <int B>
__device__ __forceinline__ Fe2L<1> el_zero(const Fe2L<B>&) {
  Fe2L<1> r;
  r.c0 = fe_zero<FqParams>();
  r.c1 = fe_zero<FqParams>();
  return r;
}
template <int B>
__device__ __forceinline__ Fe2L<16> el_one(const Fe2L<B>&) {
  Fe2L<16> r;
  r.c0 = fe_one<FqParams>();
  r.c1 = Fe<FqParams, 16>(fe_zero<FqParams>());
  return r;
}
template <int B1, int B2>
__device__ __forceinline__ auto add(const Fe2L<B1>& a, const Fe2L<B2>& b) {
  Fe2L<B1 + B2> r;
  r.c0 = add(a.c0, b.c0);
  r.c1 = add(a.c1, b.c1);
  return r;
}
template <int B1>
__device__ __forceinline__ auto dbl(const Fe2L<B1>& a) {
  Fe2L<2 * B1> r;
  r.c0 = dbl(a.c0);
  r.c1 = dbl(a.c1);
  return r;
}
template <int B1, int B2>
__device__ __forceinline__ auto sub(const Fe2L<B1>& a, const Fe2L<B2>& b) {
  Fe2L<B1 + 16 * (B2 / 16 + 1)> r;
  r.c0 = sub(a.c0, b.c0);
  r.c1 = sub(a.c1, b.c1);
  return r;
}
template <int B2>
__device__ __forceinline__ auto neg(const Fe2L<B2>& b) {
  Fe2L<16 * (B2 / 16 + 1)> r;
  r.c0 = neg(b.c0);
  r.c1 = neg(b.c1);
  return r;
}
template <int TB, int B>
__device__ __forceinline__ auto reduce_to(const Fe2L<B>& a) {
  if constexpr (B <= TB) {
    return a;
  } else {
    Fe2L<TB> r;
    r.c0 = Fe<FqParams, TB>(reduce_to<TB>(a.c0));
    r.c1 = Fe<FqParams, TB>(reduce_to<TB>(a.c1));
    return r;
  }
}
template <int B>
__device__ __forceinline__ bool is_zero(const Fe2L<B>& a) {
  return is_zero(a.c0) && is_zero(a.c1);
}
// lane 0 of the pair: (a0 + a1)(a0 - a1); lane 1: a0 (2 a1)
template <int B1>
__device__ __forceinline__ Fe2L<32> sqr(const Fe2L<B1>& a_in) {
  if constexpr (B1 > 32) {
    return sqr(reduce_to<32>(a_in));
  } else {
    const bool odd = (threadIdx.x & 1) != 0;
    const auto s = add(a_in.c0, a_in.c1);
    const auto d = sub(a_in.c0, a_in.c1);
    using TS = std::decay_t<decltype(s)>;
    using TD = std::decay_t<decltype(d)>;
    const TS x = select_el(odd, TS(a_in.c0), s);
    const TD y = select_el(odd, TD(dbl(a_in.c1)), d);
    const auto p = mul(x, y);
    const auto q = shfl_xor_el(p, 1);
    const auto t0 = select_el(odd, q, p);  // (a0 + a1)(a0 - a1)
    const auto t1 = select_el(odd, p, q);  // 2 a0 a1
    Fe2L<32> r;
    r.c0 = Fe<FqParams, 32>(reduce_to<32>(t0));
    r.c1 = Fe<FqParams, 32>(reduce_to<32>(t1));
    return r;
  }
}
// lane 0 of the pair: a0 b0 + (-a1) b1; lane 1: a0 b1 + a1 b0 (one dual product each, then exchange)
template <int B1, int B2>
__device__ __forceinline__ Fe2L<32> mul(const Fe2L<B1>& a_in, const Fe2L<B2>& b_in) {
  if constexpr (!lazy_ok(B1, B2)) {
    return mul(reduce_to<(B1 > 32 ? 32 : B1)>(a_in), reduce_to<(B2 > 32 ? 32 : B2)>(b_in));
  } else {
    const bool odd = (threadIdx.x & 1) != 0;
    const auto na1 = neg(a_in.c1);
    using TN = std::decay_t<decltype(na1)>;
    const auto p = mul2(a_in.c0, select_el(odd, b_in.c1, b_in.c0),
                        select_el(odd, TN(a_in.c1), na1), select_el(odd, b_in.c0, b_in.c1));
    const auto q = shfl_xor_el(p, 1);
    Fe2L<32> r;
    r.c0 = Fe<FqParams, 32>(select_el(odd, q, p));
    r.c1 = Fe<FqParams, 32>(select_el(odd, p, q));
    return r;
  }
}
template <int B1, int B2, int B3, int B4>
__device__ __forceinline__ auto mulsub(const Fe2L<B1>& a, const Fe2L<B2>& b, const Fe2L<B3>& c, const Fe2L<B4>& d) {
  return sub(mul(a, b), mul(c, d));
}
// ---------------------------------------------------------------- G2 on a lane OCTET
// The serial chains of G2 (Horner over the windows, the fixed-base doubling chain) on FOUR lane pairs: as on G1's
// lane quad (quad.cuh) all eight lanes hold the same point, at every level of the formula pair k multiplies the k-th
// operand pair (each product itself split over the two lanes of the pair, as above), and the four products come
// back to all lanes through one cross-lane read per limb.  A doubling is three Fq2 multiplication times deep instead
// of seven in a row, an addition five instead of sixteen: k_finalize<G2> 1.75 -> 1.31 ms, k_fb_chain<G2> 1.37 -> 0.90 ms
// (a level costs an Fq2 product, ~3.6 k cycles on a pair, plus as much again in selects, 72 cross-lane reads and the
// conditional subtractions that bring operands back under the squaring's input bound).
struct G2CfgO {
  using EX = Fe2L<144>;
  using EY = Fe2L<112>;
  using EZ = Fe2L<112>;
};
__device__ __forceinline__ Jac<G2CfgO> to_pair(const Jac<G2Cfg>& p) {
  Jac<G2CfgO> r;
  r.X = to_pair(p.X);
  r.Y = to_pair(p.Y);
  r.Z = to_pair(p.Z);
  return r;
}
__device__ __forceinline__ Jac<G2Cfg> from_pair(const Jac<G2CfgO>& p) {
  Jac<G2Cfg> r;
  r.X = from_pair(p.X);
  r.Y = from_pair(p.Y);
  r.Z = from_pair(p.Z);
  return r;
}
template <int B>
__device__ __forceinline__ Fe2L<B> select_el(bool c, const Fe2L<B>& a, const Fe2L<B>& b) {
  Fe2L<B> r;
  r.c0 = select_el(c, a.c0, b.c0);
  r.c1 = select_el(c, a.c1, b.c1);
  return r;
}
// the value held by pair K of this lane's octet (same parity of lane: both lanes of a pair hold the same element)
template <int K, int B>
__device__ __forceinline__ Fe2L<B> octet_bcast(const Fe2L<B>& v) {
  const int src = ((int)threadIdx.x & ~7) | (2 * K) | ((int)threadIdx.x & 1);
  Fe2L<B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.c0.l[i] = __shfl(v.c0.l[i], src);
    r.c1.l[i] = __shfl(v.c1.l[i], src);
  }
  return r;
}
// out[k] = a[k] * b[k], k < 4, in ONE Fq2 multiplication time: pair k computes product k
template <int BA, int BB>
__device__ __forceinline__ void octet_mul4(const Fe2L<BA> (&a)[4], const Fe2L<BB> (&b)[4], Fe2L<32> (&out)[4]) {
  const int role = ((int)threadIdx.x >> 1) & 3;
  Fe2L<BA> x = a[0];
  Fe2L<BB> y = b[0];
#pragma unroll
  for (int k = 1; k < 4; k++) {
    x = select_el(role == k, a[k], x);
    y = select_el(role == k, b[k], y);
  }
  const Fe2L<32> r = mul(x, y);
  out[0] = octet_bcast<0>(r);
  out[1] = octet_bcast<1>(r);
  out[2] = octet_bcast<2>(r);
  out[3] = octet_bcast<3>(r);
}
template <int BA>
__device__ __forceinline__ void octet_sqr4(const Fe2L<BA> (&a)[4], Fe2L<32> (&out)[4]) {
  const int role = ((int)threadIdx.x >> 1) & 3;
  Fe2L<BA> x = a[0];
#pragma unroll
  for (int k = 1; k < 4; k++) x = select_el(role == k, a[k], x);
  const Fe2L<32> r = sqr(x);
  out[0] = octet_bcast<0>(r);
  out[1] = octet_bcast<1>(r);
  out[2] = octet_bcast<2>(r);
  out[3] = octet_bcast<3>(r);
}

// dbl-2009-l (jac_dbl of ec.cuh) on an octet: three levels
__device__ __forceinline__ Jac<G2CfgO> jac_dbl(const Jac<G2CfgO>& p) {
  using F32 = Fe2L<32>;
  using FI = Fe2L<112>;          // Y and Z as they come (< 112 sixteenths of p: inside the lazy product's slack);
  static_assert(lazy_ok(112, 112), "level-1 operand bound");
  const F32 X1 = reduce_to<32>(p.X);   // X (< 144) is not, and X + Y^2 feeds a squaring (input < 2p)
  const FI Y1 = FI(p.Y), Z1 = FI(p.Z);
  F32 l1[4];
  {
    const FI a[4] = {FI(X1), Y1, Y1, Y1}, b[4] = {FI(X1), Y1, Z1, Z1};
    octet_mul4(a, b, l1);
  }
  const F32 A = l1[0], B = l1[1], YZ = l1[2];
  const auto E = add(dbl(A), A);                                  // 3 X^2        (96)
  F32 l2[4];
  {
    using F96 = Fe2L<96>;
    const F96 a[4] = {F96(E), F96(B), F96(add(X1, B)), F96(B)};
    octet_sqr4(a, l2);
  }
  const F32 F = l2[0], CC = l2[1], T = l2[2];
  const auto t = reduce_to<32>(sub(T, add(A, CC)));
  const auto D = dbl(t);
  const auto X3 = reduce_to<32>(sub(F, dbl(D)));
  const auto C8 = dbl(dbl(dbl(CC)));
  const auto Y3 = sub(mul(E, reduce_to<48>(sub(D, X3))), C8);    // (the same product on all four pairs)
  Jac<G2CfgO> r;
  r.X = G2CfgO::EX(X3);
  r.Y = G2CfgO::EY(reduce_to<64>(Y3));
  r.Z = G2CfgO::EZ(dbl(YZ));
  return r;
}

// add-2007-bl (jac_add of ec.cuh) on an octet: five levels
__device__ __forceinline__ Jac<G2CfgO> jac_add(const Jac<G2CfgO>& p, const Jac<G2CfgO>& q) {
  if (is_inf(p)) return q;   // (uniform over the octet: all eight lanes hold the same points)
  if (is_inf(q)) return p;
  using F32 = Fe2L<32>;
  using FY = Fe2L<112>;
  using FX = Fe2L<144>;
  static_assert(lazy_ok(112, 32) && lazy_ok(144, 32), "operand bounds of the first two levels");
  // X and Y enter their products as they come (the other factor is < 2p); only Z, which is squared and added, is reduced
  const FX X1 = FX(p.X), X2 = FX(q.X);
  const FY Y1 = FY(p.Y), Y2 = FY(q.Y);
  const F32 Z1 = reduce_to<32>(p.Z), Z2 = reduce_to<32>(q.Z);
  F32 l1[4];
  {
    const FY a[4] = {FY(Z1), FY(Z2), Y1, Y2};
    const F32 b[4] = {Z1, Z2, Z2, Z1};
    octet_mul4(a, b, l1);
  }
  const F32 Z1Z1 = l1[0], Z2Z2 = l1[1], Y1Z2 = l1[2], Y2Z1 = l1[3];
  F32 l2[4];
  {
    const FX a[4] = {X1, X2, FX(Y1Z2), FX(Y2Z1)};
    const F32 b[4] = {Z2Z2, Z1Z1, Z2Z2, Z1Z1};
    octet_mul4(a, b, l2);
  }
  const F32 U1 = l2[0], U2 = l2[1], S1 = l2[2], S2 = l2[3];
  const auto H = sub(U2, U1);
  const auto rh = sub(S2, S1);
  if (is_zero(H)) {
    if (is_zero(rh)) return jac_dbl(p);
    return jac_infinity<G2CfgO>();
  }
  const auto H2 = dbl(H);
  const auto r = dbl(rh);
  const auto ZS = add(Z1, Z2);
  using F3 = Fe2L<160>;
  static_assert(std::is_convertible<decltype(H2), F3>::value && std::is_convertible<decltype(r), F3>::value &&
                    std::is_convertible<decltype(ZS), F3>::value, "level-3 operand bound");
  F32 l3[4];
  {
    const F3 a[4] = {F3(H2), F3(r), F3(ZS), F3(ZS)};
    octet_sqr4(a, l3);
  }
  const F32 I = l3[0], RR = l3[1], ZZ = l3[2];
  const auto Zd = reduce_to<80>(sub(ZZ, add(Z1Z1, Z2Z2)));
  using F4 = Fe2L<80>;
  static_assert(std::is_convertible<decltype(Zd), F4>::value && std::is_convertible<decltype(H), F4>::value,
                "level-4 operand bound");
  F32 l4[4];
  {
    const F4 a[4] = {F4(H), F4(U1), F4(Zd), F4(Zd)};
    const F4 b[4] = {F4(I), F4(I), F4(H), F4(H)};
    octet_mul4(a, b, l4);
  }
  const F32 J = l4[0], V = l4[1], Z3 = l4[2];
  const auto X3 = reduce_to<64>(sub(RR, add(J, dbl(V))));
  F32 l5[4];                                                     // Y3 = r (V - X3) - 2 S1 J: both products at once
  {
    using F5A = Fe2L<160>;
    using F5B = Fe2L<112>;
    const auto VX = sub(V, X3);
    static_assert(std::is_convertible<decltype(VX), F5B>::value && std::is_convertible<decltype(r), F5A>::value &&
                      lazy_ok(160, 112), "level-5 operand bound");
    const F5A a[4] = {F5A(r), F5A(dbl(S1)), F5A(r), F5A(r)};
    const F5B b[4] = {F5B(VX), F5B(J), F5B(VX), F5B(VX)};
    octet_mul4(a, b, l5);
  }
  Jac<G2CfgO> out;
  out.X = G2CfgO::EX(X3);
  out.Y = G2CfgO::EY(sub(l5[0], l5[1]));
  out.Z = G2CfgO::EZ(Z3);
  return out;
}

template <class CV>
__device__ __forceinline__ const Jac<CV>& to_pair(const Jac<CV>& p) { return p; }   // one-lane curves: identity
template <class CV>
__device__ __forceinline__ const Jac<CV>& from_pair(const Jac<CV>& p) { return p; }
#endif

}  // namespace ozk
