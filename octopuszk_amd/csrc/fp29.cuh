// 254-bit prime-field arithmetic for gfx950 in 9 x 29-bit unsaturated limbs,
// Montgomery radix R = 2^261, one field element per lane (registers only).
//
// Why this shape (measured on MI355X, profiles/r01_ubench.txt): v_mad_u64_u32 issues
// at ~5 cycles/wave but has no carry-in, and VALU carry chains through VCC/SGPRs cost
// ~4.4 cycles per link on gfx950.  With 29-bit limbs a whole product column
// (<= 9 a*b + 9 m*p terms) fits a 64-bit accumulator, so a Montgomery multiplication
// is 171 plain `(u64)a*b + acc` MADs with no carry handling (~915 cycles/wave,
// 172 G mulmod/s chip-wide) and add/sub never touch carry flags.
//
// Replaces the role of CGBN 512-bit `mul` + `rem` in the reference
// (algebra_msm_VariableBaseMSM.cu:316-317 pattern) and
// of java.math.BigInteger in algebra/fields/Fp.java:38-92.
//
// Value bounds are tracked at COMPILE TIME: Fe<P,B> holds an integer < B*p/16 whose
// limbs 0..7 are < 2^29 ("normalised"); every operation static_asserts its
// precondition and returns the tightest bound it can prove, so a formula whose
// lazy reductions are insufficient does not compile.
#pragma once
#include "consts_gen.h"
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OZK_HD __host__ __device__ __forceinline__
#else
#define OZK_HD inline
#endif

namespace ozk {

constexpr int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
// 2^261 / p > 169 for both BN254 primes (p < 2^253.6): out/p < 1 + (A/p)(B/p)/169.
constexpr int MONT_SLACK = 169;
constexpr int FE_BMAX = 16 * FE_MAXK;

template <class P, int B>
struct Fe {
  static_assert(B >= 1 && B <= FE_BMAX, "bound out of range");
  u32 l[9];
  OZK_HD Fe() {}
  // widening (never narrowing) conversion between bounds
  template <int B2, class = std::enable_if_t<(B2 <= B)>>
  OZK_HD Fe(const Fe<P, B2>& o) {
#pragma unroll
    for (int i = 0; i < 9; i++) l[i] = o.l[i];
  }
};

template <class P, int B>
OZK_HD Fe<P, B> fe_const(const u32 (&c)[9]) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c[i];
  return r;
}
template <class P>
OZK_HD Fe<P, 16> fe_one() { return fe_const<P, 16>(P::ONE); }
template <class P>
OZK_HD Fe<P, 1> fe_zero() {
  Fe<P, 1> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = 0;
  return r;
}

// zero / one of the same field as the argument (lets ec.cuh stay generic over Fq / Fq2)
template <class P, int B>
OZK_HD Fe<P, 1> el_zero(const Fe<P, B>&) { return fe_zero<P>(); }
template <class P, int B>
OZK_HD Fe<P, 16> el_one(const Fe<P, B>&) { return fe_one<P>(); }

OZK_HD u64 mad64(u32 a, u32 b, u64 c) { return (u64)a * b + c; }

// carry-propagate limbs 0..7 down to < 2^29 (limb 8 keeps the rest)
template <class P, int B>
OZK_HD void fe_carry(Fe<P, B>& a) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a.l[i + 1] += a.l[i] >> FE_W;
    a.l[i] &= FE_MASK;
  }
}

// ---------------------------------------------------------------- device Montgomery kernel
// On the GPU the product columns are built from chains of v_mad_u64_u32 held in ONE asm statement each
// (mad_chain_gen.h): the carry of column k - 1 is the accumulator the chain of column k starts from, so there is
// no 64-bit add per column, and hipcc cannot split the column sums over separate accumulators and ripple the
// carries afterwards (its default schedule: 171 multiply-adds + 17 64-bit shifts + 17 64-bit adds, all on the
// multiplier's pipe, and the ripple a dependent tail at the end of every multiplication).  Measured at 3 waves per
// SIMD with one dependent stream per wave: 1211 -> 921 cycles per multiplication; 179 G mulmod/s chip-wide at 8
// waves against 172 (profiles/r02_ubench_mont.txt).  NP products a_t * b_t share the accumulator (mul2 / mul4).
#if defined(__HIPCC__)
}  // namespace ozk
#include "mad_chain_gen.h"
namespace ozk {
template <class P, int K, int NP>
__device__ __forceinline__ void mont_col(u64& acc, u32 (&m)[9], const u32 (&a)[NP][9], const u32 (&b)[NP][9], u32 (&r)[9]) {
  constexpr int i0 = K < 9 ? 0 : K - 8, i1 = K < 9 ? K : 8, N = i1 - i0 + 1;
#pragma unroll
  for (int t = 0; t < NP; t++) {
    u32 xs[9], ys[9];
#pragma unroll
    for (int j = 0; j < N; j++) {
      xs[j] = a[t][i0 + j];
      ys[j] = b[t][K - i0 - j];
    }
    mad_chain_vv<N>(acc, xs, ys);
  }
  {
    constexpr int N2 = K < 9 ? K : 17 - K, m0 = K < 9 ? 0 : K - 8;
    u32 xs[9], ys[9];
#pragma unroll
    for (int j = 0; j < N2; j++) {
      xs[j] = m[m0 + j];
      ys[j] = P::P[K - m0 - j];
    }
    mad_chain_vs<N2>(acc, xs, ys);
  }
  if constexpr (K < 9) {
    m[K] = ((u32)acc * P::PINV) & FE_MASK;
    acc = mad64(m[K], P::P[0], acc);
  } else {
    r[K - 9] = (u32)acc & FE_MASK;
  }
  acc >>= FE_W;
  if constexpr (K < 16) mont_col<P, K + 1, NP>(acc, m, a, b, r);
}
template <class P, int NP>
__device__ __forceinline__ void mont_device(const u32 (&a)[NP][9], const u32 (&b)[NP][9], u32 (&r)[9]) {
  u64 acc = 0;
  u32 m[9];
  mont_col<P, 0, NP>(acc, m, a, b, r);
  r[8] = (u32)acc;
}
// squaring: column K = sum_{2i < K} (2 a_i) a_{K-i} + [K even] a_{K/2}^2
template <class P, int K>
__device__ __forceinline__ void mont_sqr_col(u64& acc, u32 (&m)[9], const u32 (&a)[9], const u32 (&a2)[9], u32 (&r)[9]) {
  constexpr int lo = K < 9 ? 0 : K - 8, hi = K == 0 ? -1 : (K - 1) / 2, NX = hi >= lo ? hi - lo + 1 : 0;
  constexpr int N = NX + (K % 2 == 0 ? 1 : 0);
  {
    u32 xs[9], ys[9];
#pragma unroll
    for (int j = 0; j < NX; j++) {
      xs[j] = a2[lo + j];
      ys[j] = a[K - lo - j];
    }
    if constexpr (K % 2 == 0) {
      xs[NX] = a[K / 2];
      ys[NX] = a[K / 2];
    }
    mad_chain_vv<N>(acc, xs, ys);
  }
  {
    constexpr int N2 = K < 9 ? K : 17 - K, m0 = K < 9 ? 0 : K - 8;
    u32 xs[9], ys[9];
#pragma unroll
    for (int j = 0; j < N2; j++) {
      xs[j] = m[m0 + j];
      ys[j] = P::P[K - m0 - j];
    }
    mad_chain_vs<N2>(acc, xs, ys);
  }
  if constexpr (K < 9) {
    m[K] = ((u32)acc * P::PINV) & FE_MASK;
    acc = mad64(m[K], P::P[0], acc);
  } else {
    r[K - 9] = (u32)acc & FE_MASK;
  }
  acc >>= FE_W;
  if constexpr (K < 16) mont_sqr_col<P, K + 1>(acc, m, a, a2, r);
}
#endif

// ---------------------------------------------------------------- mul / sqr
template <class P, int B1, int B2>
OZK_HD auto mul(const Fe<P, B1>& a, const Fe<P, B2>& b) {
  static_assert((long long)B1 * B2 <= (long long)MONT_SLACK * 256, "Montgomery input bounds too large");
  constexpr int BO = 16 + ceil_div((long long)B1 * B2, 16 * MONT_SLACK);
  Fe<P, BO> r;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    u32 aa[1][9], bb[1][9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      aa[0][i] = a.l[i];
      bb[0][i] = b.l[i];
    }
    mont_device<P, 1>(aa, bb, r.l);
    return r;
  }
#endif
  u32 m[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; i++) acc = mad64(m[i], P::P[k - i], acc);
    m[k] = ((u32)acc * P::PINV) & FE_MASK;
    acc = mad64(m[k], P::P[0], acc);
    acc >>= FE_W;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::P[k - i], acc);
    r.l[k - 9] = (u32)acc & FE_MASK;
    acc >>= FE_W;
  }
  r.l[8] = (u32)acc;
  return r;
}

// a*b + c*d with ONE Montgomery reduction (the two 9 x 9 limb products share the column accumulator:
// 18 + 9 products of < 2^58 plus the carry stay below 2^63).  This is what makes a lazily reduced Fq2
// product cheap: 2 x (162 + 90) multiply-adds and no carry-normalising additions, against 3 x 171
// plus five additions / subtractions and two conditional subtractions for Karatsuba.
template <class P, int B1, int B2, int B3, int B4>
OZK_HD auto mul2(const Fe<P, B1>& a, const Fe<P, B2>& b, const Fe<P, B3>& c, const Fe<P, B4>& d) {
  constexpr long long BB = (long long)B1 * B2 + (long long)B3 * B4;
  static_assert(BB <= (long long)MONT_SLACK * 256, "Montgomery input bounds too large");
  constexpr int BO = 16 + ceil_div(BB, 16 * MONT_SLACK);
  Fe<P, BO> r;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    u32 aa[2][9], bb[2][9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      aa[0][i] = a.l[i];
      bb[0][i] = b.l[i];
      aa[1][i] = c.l[i];
      bb[1][i] = d.l[i];
    }
    mont_device<P, 2>(aa, bb, r.l);
    return r;
  }
#endif
  u32 m[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = 0; i <= k; i++) acc = mad64(c.l[i], d.l[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; i++) acc = mad64(m[i], P::P[k - i], acc);
    m[k] = ((u32)acc * P::PINV) & FE_MASK;
    acc = mad64(m[k], P::P[0], acc);
    acc >>= FE_W;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(c.l[i], d.l[k - i], acc);
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::P[k - i], acc);
    r.l[k - 9] = (u32)acc & FE_MASK;
    acc >>= FE_W;
  }
  r.l[8] = (u32)acc;
  return r;
}

// a*b + c*d + e*f + g*h with one reduction (36 + 9 products of < 2^58 plus the carry: < 2^63.6).
// Two of these make an Fq2 "a b - c d" (fq2.cuh mulsub).
template <class P, int B1, int B2, int B3, int B4, int B5, int B6, int B7, int B8>
OZK_HD auto mul4(const Fe<P, B1>& a, const Fe<P, B2>& b, const Fe<P, B3>& c, const Fe<P, B4>& d,
                 const Fe<P, B5>& e, const Fe<P, B6>& f, const Fe<P, B7>& g, const Fe<P, B8>& h) {
  constexpr long long BB = (long long)B1 * B2 + (long long)B3 * B4 + (long long)B5 * B6 + (long long)B7 * B8;
  static_assert(BB <= (long long)MONT_SLACK * 256, "Montgomery input bounds too large");
  constexpr int BO = 16 + ceil_div(BB, 16 * MONT_SLACK);
  Fe<P, BO> r;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    u32 aa[4][9], bb[4][9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      aa[0][i] = a.l[i];
      bb[0][i] = b.l[i];
      aa[1][i] = c.l[i];
      bb[1][i] = d.l[i];
      aa[2][i] = e.l[i];
      bb[2][i] = f.l[i];
      aa[3][i] = g.l[i];
      bb[3][i] = h.l[i];
    }
    mont_device<P, 4>(aa, bb, r.l);
    return r;
  }
#endif
  u32 m[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      acc = mad64(a.l[i], b.l[k - i], acc);
      acc = mad64(c.l[i], d.l[k - i], acc);
      acc = mad64(e.l[i], f.l[k - i], acc);
      acc = mad64(g.l[i], h.l[k - i], acc);
    }
#pragma unroll
    for (int i = 0; i < k; i++) acc = mad64(m[i], P::P[k - i], acc);
    m[k] = ((u32)acc * P::PINV) & FE_MASK;
    acc = mad64(m[k], P::P[0], acc);
    acc >>= FE_W;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc = mad64(a.l[i], b.l[k - i], acc);
      acc = mad64(c.l[i], d.l[k - i], acc);
      acc = mad64(e.l[i], f.l[k - i], acc);
      acc = mad64(g.l[i], h.l[k - i], acc);
    }
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::P[k - i], acc);
    r.l[k - 9] = (u32)acc & FE_MASK;
    acc >>= FE_W;
  }
  r.l[8] = (u32)acc;
  return r;
}

// c ? a : b, limb-wise (v_cndmask)
template <class P, int B>
OZK_HD Fe<P, B> select_el(bool c, const Fe<P, B>& a, const Fe<P, B>& b) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// multiplication by a base-field constant; overloaded component-wise for Fq2 (fq2.cuh)
template <class P, int B1>
OZK_HD auto scale(const Fe<P, B1>& a, const Fe<P, 16>& k) { return mul(a, k); }

template <class P, int B1>
OZK_HD auto sqr(const Fe<P, B1>& a) {
  static_assert((long long)B1 * B1 <= (long long)MONT_SLACK * 256, "Montgomery input bounds too large");
  constexpr int BO = 16 + ceil_div((long long)B1 * B1, 16 * MONT_SLACK);
  Fe<P, BO> r;
  u32 m[9], a2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
  u64 acc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  mont_sqr_col<P, 0>(acc, m, a.l, a2, r.l);
  r.l[8] = (u32)acc;
  return r;
#endif
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc = mad64(a2[i], a.l[k - i], acc);
    if (k % 2 == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
    for (int i = 0; i < k; i++) acc = mad64(m[i], P::P[k - i], acc);
    m[k] = ((u32)acc * P::PINV) & FE_MASK;
    acc = mad64(m[k], P::P[0], acc);
    acc >>= FE_W;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; i++) acc = mad64(a2[i], a.l[k - i], acc);
    if (k % 2 == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc = mad64(m[i], P::P[k - i], acc);
    r.l[k - 9] = (u32)acc & FE_MASK;
    acc >>= FE_W;
  }
  r.l[8] = (u32)acc;
  return r;
}

// ---------------------------------------------------------------- add / sub
template <class P, int B1, int B2>
OZK_HD auto add(const Fe<P, B1>& a, const Fe<P, B2>& b) {
  Fe<P, B1 + B2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  fe_carry(r);
  return r;
}

template <class P, int B1>
OZK_HD auto dbl(const Fe<P, B1>& a) {
  Fe<P, 2 * B1> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1;
  fe_carry(r);
  return r;
}

// a - b + K*p with the smallest K*p that dominates b limb-wise.
template <class P, int B1, int B2>
OZK_HD auto sub(const Fe<P, B1>& a, const Fe<P, B2>& b) {
  constexpr int K = B2 / 16 + 1;
  static_assert(K <= FE_MAXK, "sub bias table too small");
  Fe<P, B1 + 16 * K> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (P::BIAS[K][i] - b.l[i]);
  fe_carry(r);
  return r;
}

template <class P, int B2>
OZK_HD auto neg(const Fe<P, B2>& b) {
  constexpr int K = B2 / 16 + 1;
  static_assert(K <= FE_MAXK, "sub bias table too small");
  Fe<P, 16 * K> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = P::BIAS[K][i] - b.l[i];
  fe_carry(r);
  return r;
}

// a*b - c*d: one dual product on (a, b, -c, d) when the bounds allow, i.e. one reduction and no
// subtraction of the two reduced products (the Y3 of every addition / doubling formula in ec.cuh)
template <class P, int B1, int B2, int B3, int B4>
OZK_HD auto mulsub(const Fe<P, B1>& a, const Fe<P, B2>& b, const Fe<P, B3>& c, const Fe<P, B4>& d) {
  if constexpr ((long long)B1 * B2 + 16LL * (B3 / 16 + 1) * B4 <= (long long)MONT_SLACK * 256) {
    return mul2(a, b, neg(c), d);
  } else {
    return sub(mul(a, b), mul(c, d));
  }
}

// ---------------------------------------------------------------- un-normalised ("loose") elements
// A sum or difference whose carry pass (24 instructions: shift, add, mask per limb) has been LEFT OUT.  The value
// bound B means what it means for Fe; limbs 0..7 are no longer below 2^29 but below LU * 2^28 (a normalised
// element has LU = 2, a BIAS limb is < 2^30: LU = 4).  Legal uses, each checked at compile time:
//   * ONE factor of a multiplication whose other factor is normalised:  9 (LU 2^28) 2^29 + 9 2^58 + carry < 2^64
//     needs LU <= 12; the two loose factors of a dual product a b + c d share that budget (LU_a + LU_c <= 12);
//   * further un-normalised sums, as long as every limb stays below 2^32 (LU <= 15);
//   * normalise(): the carry pass, back to an Fe.
// In the bucket accumulation's mixed addition this removes 4 of 7 carry passes and the conditional subtraction of
// the negated y (ec.cuh xyzz_madd_lazy): ~170 of ~2290 instructions per addition.
template <class P, int B, int LU>
struct FeL {
  static_assert(B >= 1 && B <= FE_BMAX, "bound out of range");
  static_assert(LU >= 2 && LU <= 15, "limb bound out of range");
  u32 l[9];
  OZK_HD FeL() {}
  OZK_HD FeL(const Fe<P, B>& o) {  // a normalised element is a loose one
#pragma unroll
    for (int i = 0; i < 9; i++) l[i] = o.l[i];
  }
};
template <class P, int B, int LU>
OZK_HD Fe<P, B> normalise(const FeL<P, B, LU>& a) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i];
  fe_carry(r);
  return r;
}
// K p - b, no carry (limbs < 2^30)
template <class P, int B2>
OZK_HD auto neg_nc(const Fe<P, B2>& b) {
  constexpr int K = B2 / 16 + 1;
  static_assert(K <= FE_MAXK, "sub bias table too small");
  FeL<P, 16 * K, 4> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = P::BIAS[K][i] - b.l[i];
  return r;
}
// a - b + K p, no carry; the subtrahend must be normalised (a BIAS limb dominates a normalised limb only)
template <class P, int B1, int B2>
OZK_HD auto sub_nc(const Fe<P, B1>& a, const Fe<P, B2>& b) {
  constexpr int K = B2 / 16 + 1;
  static_assert(K <= FE_MAXK, "sub bias table too small");
  FeL<P, B1 + 16 * K, 6> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (P::BIAS[K][i] - b.l[i]);
  return r;
}
// a - b - 2 c with ONE carry pass at the end (the X3 = R^2 - PPP - 2 Q of the addition formulas): limbs stay below
// 2^29 + 2^30 + 2 2^30 < 2^32 on the way
template <class P, int B1, int B2, int B3>
OZK_HD auto sub_sub2(const Fe<P, B1>& a, const Fe<P, B2>& b, const Fe<P, B3>& c) {
  constexpr int K2 = B2 / 16 + 1, K3 = B3 / 16 + 1;
  static_assert(K2 <= FE_MAXK && K3 <= FE_MAXK, "sub bias table too small");
  Fe<P, B1 + 16 * K2 + 32 * K3> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (P::BIAS[K2][i] - b.l[i]) + ((P::BIAS[K3][i] - c.l[i]) << 1);
  fe_carry(r);
  return r;
}
// c ? a : b where either may be loose
template <class P, int B1, int LU1, int B2>
OZK_HD auto select_el(bool c, const FeL<P, B1, LU1>& a, const Fe<P, B2>& b) {
  FeL<P, (B1 > B2 ? B1 : B2), LU1> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
// loose x normalised
template <class P, int B1, int LU, int B2>
OZK_HD auto mul(const FeL<P, B1, LU>& a, const Fe<P, B2>& b) {
  static_assert(LU <= 12, "loose factor too wide for the 64-bit column accumulator");
  Fe<P, B1> an;  // same limbs, the type only carries the value bound into the product
#pragma unroll
  for (int i = 0; i < 9; i++) an.l[i] = a.l[i];
  return mul(an, b);
}
// a b + c d, a and c loose, b and d normalised
template <class P, int B1, int LU1, int B2, int B3, int LU3, int B4>
OZK_HD auto mul2(const FeL<P, B1, LU1>& a, const Fe<P, B2>& b, const FeL<P, B3, LU3>& c, const Fe<P, B4>& d) {
  static_assert(LU1 + LU3 <= 12, "loose factors too wide for the 64-bit column accumulator");
  Fe<P, B1> an;
  Fe<P, B3> cn;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    an.l[i] = a.l[i];
    cn.l[i] = c.l[i];
  }
  return mul2(an, b, cn, d);
}

// ---- general product sums over loose factors (the lazily carried Fq2 arithmetic, fq2.cuh).  Every factor is an
// FeL (a normalised element is FeL<., ., 2>: loose()); the 64-bit column accumulator holds
//     sum over the products of 9 (LUx 2^28)(LUy 2^28) + 9 2^58 (the m p terms) + carry < 2^64
// as long as the LU products add up to at most 24 (a product of two normalised factors counts 4).
template <class P, int B>
OZK_HD FeL<P, B, 2> loose(const Fe<P, B>& a) { return FeL<P, B, 2>(a); }
template <class P, int B1, int L1, int B2, int L2>
OZK_HD auto add_nc(const FeL<P, B1, L1>& a, const FeL<P, B2, L2>& b) {
  FeL<P, B1 + B2, L1 + L2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <class P, int B1, int L1>
OZK_HD auto dbl_nc(const FeL<P, B1, L1>& a) {
  FeL<P, 2 * B1, 2 * L1> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1;
  return r;
}
template <class P, int B, int LU>
OZK_HD Fe<P, B> as_fe(const FeL<P, B, LU>& a) {  // the limbs as they are, for the multiplier only
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i];
  return r;
}
template <class P, int B1, int L1, int B2, int L2>
OZK_HD auto mul_ll(const FeL<P, B1, L1>& a, const FeL<P, B2, L2>& b) {
  static_assert(L1 * L2 <= 24, "loose factors too wide for the 64-bit column accumulator");
  return mul(as_fe(a), as_fe(b));
}
template <class P, int B1, int L1, int B2, int L2, int B3, int L3, int B4, int L4>
OZK_HD auto mul2_ll(const FeL<P, B1, L1>& a, const FeL<P, B2, L2>& b, const FeL<P, B3, L3>& c, const FeL<P, B4, L4>& d) {
  static_assert(L1 * L2 + L3 * L4 <= 24, "loose factors too wide for the 64-bit column accumulator");
  return mul2(as_fe(a), as_fe(b), as_fe(c), as_fe(d));
}
template <class P, int B1, int L1, int B2, int L2, int B3, int L3, int B4, int L4, int B5, int L5, int B6, int L6, int B7,
          int L7, int B8, int L8>
OZK_HD auto mul4_ll(const FeL<P, B1, L1>& a, const FeL<P, B2, L2>& b, const FeL<P, B3, L3>& c, const FeL<P, B4, L4>& d,
                    const FeL<P, B5, L5>& e, const FeL<P, B6, L6>& f, const FeL<P, B7, L7>& g, const FeL<P, B8, L8>& h) {
  static_assert(L1 * L2 + L3 * L4 + L5 * L6 + L7 * L8 <= 24, "loose factors too wide for the 64-bit column accumulator");
  return mul4(as_fe(a), as_fe(b), as_fe(c), as_fe(d), as_fe(e), as_fe(f), as_fe(g), as_fe(h));
}

// if (a >= K*p) a -= K*p          (branch-free, signed borrow propagation)
template <int K, class P, int B>
OZK_HD auto csub(const Fe<P, B>& a) {
  static_assert(K >= 1 && K <= FE_MAXK, "csub constant out of range");
  constexpr int BO = (B - 16 * K > 16 * K) ? (B - 16 * K) : 16 * K;
  Fe<P, BO> r;
  int32_t c = 0;
  u32 d[9];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int32_t t = (int32_t)a.l[i] - (int32_t)P::KP[K][i] + c;
    d[i] = (u32)t & FE_MASK;
    c = t >> FE_W;
  }
  int32_t top = (int32_t)a.l[8] - (int32_t)P::KP[K][8] + c;
  d[8] = (u32)top;
  const bool keep = top < 0;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = keep ? a.l[i] : d[i];
  return r;
}

// reduce to < TB*p/16 (TB >= 16) with the fewest conditional subtractions
template <int TB, class P, int B>
OZK_HD auto reduce_to(const Fe<P, B>& a) {
  if constexpr (B <= TB) {
    return a;
  } else {
    constexpr int KT = TB / 16;  // largest multiple of p not above the target
    // one csub<KT> lands at max(B-16KT, 16KT) <= TB when B-16KT <= TB; otherwise halve first
    constexpr int K1 = (B - 16 * KT <= TB) ? KT : ((B + 31) / 32);
    static_assert(K1 >= 1 && K1 <= FE_MAXK, "reduce_to: K out of range");
    return reduce_to<TB>(csub<K1>(a));
  }
}

// Reduction by an estimated quotient: a < B*p/16 (B up to 16*FE_MAXK) -> a - q*p < 17*p/16 in ONE subtraction of a
// table row, instead of reduce_to's chain of conditional subtractions (three or four of them, ~37 instructions each,
// from the bounds the FFT passes end with).  With ptop = p >> 232 (the top limb of p) and top = a >> 232:
//   q = floor(top / (ptop + 1))   =>   q*p <= q*(ptop+1)*2^232 <= top*2^232 <= a            (never negative)
//   a - q*p < (top+1)*2^232 - q*p <= (q+1)(ptop+1)*2^232 - q*ptop*2^232 = (ptop + q + 1)*2^232 < p*(1 + 2^-16)
// The division is exact as floor(top*M / 2^53), M = ceil(2^53 / (ptop+1)): the product overshoots top/(ptop+1) by
// less than top/2^53 < 2^-21 while the fractional part of top/(ptop+1) is at most 1 - 1/(ptop+1) = 1 - 2^-21.6.
template <class P, int B>
OZK_HD Fe<P, 17> reduce_q(const Fe<P, B>& a) {
  static_assert(B <= 16 * FE_MAXK, "reduce_q: no table row for the largest quotient");
  constexpr u64 D = (u64)P::P[8] + 1;
  static_assert(D > (1u << 21) && D < (1u << 22), "reduce_q: the modulus must have 254 bits");
  constexpr u64 M = ((1ull << 53) + D - 1) / D;
  static_assert(M < (1ull << 32), "reduce_q: reciprocal does not fit a word");
  static_assert((u64)B * D < (16ull << 31), "reduce_q: the top limb must stay below 2^31");
  const u32 q = (u32)(((u64)a.l[8] * (u32)M) >> 53);   // v_mul_hi_u32 + shift
  Fe<P, 17> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int32_t t = (int32_t)a.l[i] - (int32_t)P::KP[q][i] + c;
    r.l[i] = (u32)t & FE_MASK;
    c = t >> FE_W;
  }
  r.l[8] = (u32)((int32_t)a.l[8] - (int32_t)P::KP[q][8] + c);
  return r;
}

// canonical representative in [0, p)
template <class P, int B>
OZK_HD Fe<P, 16> canonical(const Fe<P, B>& a) {
  auto u = csub<1>(Fe<P, 32>(reduce_to<32>(a)));  // < 2p, then one more conditional -p
  Fe<P, 16> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = u.l[i];
  return r;
}

// the same through reduce_q: two subtractions whatever the bound (canonical() needs five from B = 352)
template <class P, int B>
OZK_HD Fe<P, 16> canonical_q(const Fe<P, B>& a) {
  auto u = csub<1>(reduce_q(a));
  Fe<P, 16> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = u.l[i];
  return r;
}

// a == 0 (mod p) ?   a < B*p/16, so a is one of 0, p, 2p, ...
template <class P, int B>
OZK_HD bool is_zero(const Fe<P, B>& a) {
  constexpr int KM = (B + 15) / 16;  // multiples 0..KM-1 are possible
  bool z = false;
#pragma unroll
  for (int k = 0; k < KM; k++) {
    if (a.l[0] == P::KP[k][0]) {
      bool e = true;
#pragma unroll
      for (int i = 1; i < 9; i++) e = e && (a.l[i] == P::KP[k][i]);
      z = z || e;
    }
  }
  return z;
}

template <class P, int B1, int B2>
OZK_HD bool eq(const Fe<P, B1>& a, const Fe<P, B2>& b) {
  return is_zero(sub(a, b));
}

// ---------------------------------------------------------------- pack / unpack
// 8 x u32 little-endian words (value < 2^256)  <->  9 x 29-bit limbs
template <class P, int B = 85>
OZK_HD Fe<P, B> unpack(const u32 (&w)[8]) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = FE_W * i, wi = bit / 32, sh = bit % 32;
    u32 v = w[wi] >> sh;
    if (sh + FE_W > 32 && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
    r.l[i] = (i < 8) ? (v & FE_MASK) : v;
  }
  return r;
}

template <class P, int B>
OZK_HD void pack(const Fe<P, B>& a, u32 (&w)[8]) {
  static_assert(B <= 84, "value may not fit 256 bits");
#pragma unroll
  for (int j = 0; j < 8; j++) {
    // word j = bits [32j, 32j+32)
    const int lo = (32 * j) / FE_W, sh = (32 * j) % FE_W;
    u32 v = a.l[lo] >> sh;
    if (lo + 1 < 9) v |= a.l[lo + 1] << (FE_W - sh);
    if (FE_W - sh + FE_W < 32 && lo + 2 < 9) v |= a.l[lo + 2] << (2 * FE_W - sh);
    w[j] = v;
  }
}

// wire (canonical, non-Montgomery; any 256-bit value accepted, reduced mod p) -> Montgomery
template <class P>
OZK_HD Fe<P, 17> to_mont(const u32 (&w)[8]) {
  return Fe<P, 17>(mul(unpack<P, 85>(w), fe_const<P, 16>(P::R2)));
}

// Montgomery -> canonical integer words in [0, p)
template <class P, int B>
OZK_HD void from_mont(const Fe<P, B>& a, u32 (&w)[8]) {
  Fe<P, 1> one = fe_zero<P>();
  one.l[0] = 1;
  auto t = mul(a, one);
  pack(canonical(t), w);
}

// ---------------------------------------------------------------- inversion
// a^-1 (Montgomery form in, Montgomery form out); a == 0 -> 0.  Replaces BigInteger.modInverse
// (Fp.java:90-92).
//
// Bernstein-Yang "safegcd" divsteps in the fixed-iteration form of libsecp256k1's modinv32
// (https://gcd.cr.yp.to/safegcd-20190413.pdf; 20 rounds of 30 divsteps bound any 256-bit modulus):
// every round derives a 2x2 transition matrix with 30-bit entries from the low words of (f, g) with
// 30 branch-free steps of 32-bit ALU work, then applies it to the 9 x 30-bit signed limbs of (f, g)
// and, modulo p, of (d, e) with 64-bit multiply-adds.  No data-dependent branch (all lanes of a wave
// run in lockstep), ~100 multiply-adds and ~400 ALU ops per round: ~10x cheaper than the Fermat
// exponentiation (254 squarings + 127 multiplications = 60 k multiply-adds) that it replaces in the
// Horner kernels, the fixed-base normalisation, the Jacobian-input conversion and the QAP constants.
// inv_fermat is kept as the independent check (tests/test_host_arith.py).
template <class P, int B>
OZK_HD Fe<P, 32> inv_fermat(const Fe<P, B>& a_in) {
  const Fe<P, 32> a = reduce_to<32>(a_in);
  Fe<P, 32> r = fe_one<P>();
  for (int i = 253; i >= 0; i--) {
    r = Fe<P, 32>(sqr(r));
    if ((P::PM2[i >> 5] >> (i & 31)) & 1) r = Fe<P, 32>(mul(r, a));
  }
  return r;
}

namespace safegcd {
constexpr int32_t M30 = (int32_t)(0xffffffffu >> 2);
struct Trans {
  int32_t u, v, q, r;
};
// 30 divsteps on the low words; zeta = -(delta + 1/2)
OZK_HD int32_t divsteps_30(int32_t zeta, u32 f0, u32 g0, Trans& t) {
  u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 6
  for (int i = 0; i < 30; i++) {
    u32 m1 = (u32)(zeta >> 31);
    const u32 m2 = 0u - (g & 1u);
    const u32 x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;
    g += x & m2;
    q += y & m2;
    r += z & m2;
    m1 &= m2;
    zeta = (zeta ^ (int32_t)m1) - 1;
    f += g & m1;
    u += q & m1;
    v += r & m1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return zeta;
}
// (f, g) <- t (f, g) / 2^30   (exact)
OZK_HD void update_fg(int32_t (&f)[9], int32_t (&g)[9], const Trans& t) {
  int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
  int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
    cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
    f[i - 1] = (int32_t)cf & M30;
    cf >>= 30;
    g[i - 1] = (int32_t)cg & M30;
    cg >>= 30;
  }
  f[8] = (int32_t)cf;
  g[8] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod p, kept in (-2p, p)
template <class P>
OZK_HD void update_de(int32_t (&d)[9], int32_t (&e)[9], const Trans& t) {
  const int32_t sd = d[8] >> 31, se = e[8] >> 31;
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
  int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
  md -= (int32_t)((P::PINV30 * (u32)cd + (u32)md) & (u32)M30);
  me -= (int32_t)((P::PINV30 * (u32)ce + (u32)me) & (u32)M30);
  cd += (int64_t)(int32_t)P::P30[0] * md;
  ce += (int64_t)(int32_t)P::P30[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)(int32_t)P::P30[i] * md;
    ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)(int32_t)P::P30[i] * me;
    d[i - 1] = (int32_t)cd & M30;
    cd >>= 30;
    e[i - 1] = (int32_t)ce & M30;
    ce >>= 30;
  }
  d[8] = (int32_t)cd;
  e[8] = (int32_t)ce;
}
// r in (-2p, p), sign of f -> r * sign(f) in [0, p)
template <class P>
OZK_HD void normalize(int32_t (&r)[9], int32_t sign) {
  int32_t ca = r[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] += (int32_t)P::P30[i] & ca;
  const int32_t cn = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = (r[i] ^ cn) - cn;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[i + 1] += r[i] >> 30;
    r[i] &= M30;
  }
  ca = r[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] += (int32_t)P::P30[i] & ca;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[i + 1] += r[i] >> 30;
    r[i] &= M30;
  }
}
// x in [0, p) as 8 x u32 words -> x^-1 mod p (0 -> 0), 8 x u32 words
template <class P>
OZK_HD void modinv_words(const u32 (&x)[8], u32 (&out)[8]) {
  int32_t f[9], g[9], d[9], e[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
    u32 v = x[wi] >> sh;
    if (sh + 30 > 32 && wi + 1 < 8) v |= x[wi + 1] << (32 - sh);
    g[i] = (int32_t)(v & (u32)M30);
    f[i] = (int32_t)P::P30[i];
    d[i] = 0;
    e[i] = (i == 0);
  }
  int32_t zeta = -1;
  for (int it = 0; it < 20; it++) {
    Trans t;
    zeta = divsteps_30(zeta, (u32)f[0], (u32)g[0], t);
    update_de<P>(d, e, t);
    update_fg(f, g, t);
  }
  normalize<P>(d, f[8]);
#pragma unroll
  for (int w = 0; w < 8; w++) {
    // word w = bits [32w, 32w + 32) of sum d[i] 2^(30 i)
    const int bit = 32 * w, li = bit / 30, sh = bit % 30;
    u32 v = (u32)d[li] >> sh;
    if (li + 1 < 9) v |= (u32)d[li + 1] << (30 - sh);
    if (30 - sh + 30 < 32 && li + 2 < 9) v |= (u32)d[li + 2] << (60 - sh);
    out[w] = v;
  }
}
}  // namespace safegcd

template <class P, int B>
OZK_HD Fe<P, 32> inv(const Fe<P, B>& a_in) {
  u32 w[8], r[8];
  pack(canonical(a_in), w);              // the integer a R mod p
  safegcd::modinv_words<P>(w, r);        // a^-1 R^-1
  return Fe<P, 32>(mul(unpack<P, 16>(r), fe_const<P, 16>(P::R3)));  // (a^-1 R^-1) R^3 / R = a^-1 R
}

}  // namespace ozk
