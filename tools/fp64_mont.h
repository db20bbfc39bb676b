// Variant "J" of tools/ubench_mont.hip (VERDICT r3 "next" 2 calls it variant F; F, G and H were already taken in that
// file): a Montgomery multiplication mod the BN254 base-field prime on the FP64 FMA pipe.
//   * 5 limbs of 52 bits, R = 2^260; a limb travels as an exact integer-valued double in [0, 2^52).
//   * a 52 x 52-bit limb product is split EXACTLY by two fused multiply-adds in round-toward-zero mode
//     (Emmart / Zheng / Weems, "Faster modular exponentiation using double precision floating point arithmetic on
//     the GPU", ARITH 2018):   h = fma_rz(a, b, 2^104)            = 2^104 + H,  H = a b rounded down to a multiple of 2^52
//                              l = fma_rz(a, b, (2^104 + 2^52) - h) = 2^52 + L,   L = a b - H   in [0, 2^52)
//     so the 52-bit mantissa FIELDS of h and l are H / 2^52 and L; column sums are 64-bit INTEGER additions of the
//     raw bit patterns, the exponent fields cancel against per-column compile-time constants.
//   * word-serial Montgomery reduction with 52-bit digits; the digit q = t p' mod 2^52 is the L of one more product.
// Instruction count per multiplication: 55 limb products x (2 v_fma_f64 + 1 v_add_f64) = 165 FP64 operations + 110
// 64-bit integer additions + ~60 for carries, masks and the integer -> double conversions of the result, against
// 162 v_mad_u64_u32 + ~70 for the shipped 9 x 29-bit form (variant E).  Both kinds of instruction issue at ~5 cycles
// per wave on this chip (profiles/r01_ubench.txt: mad_u64 5.4, fma_f64 5.1, add_u64 4.8), so the count decides.
//
// The same source compiles for the host (tools/fp64_mont_hostcheck.cc: fesetround(FE_TOWARDZERO) + std::fma) where
// it is checked against an exact big-integer a b R^-1 mod p.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FP64_HD __host__ __device__ __forceinline__
#else
#define FP64_HD inline
#endif

namespace fp64mont {

constexpr uint64_t M52 = (1ull << 52) - 1;
constexpr uint64_t BITS_2P52 = 0x4330000000000000ull;   // bit pattern of 2^52
constexpr uint64_t BITS_2P104 = 0x4670000000000000ull;  // bit pattern of 2^104

// p = 21888242871839275222246405745257275088696311157297823662689037894645226208583 in 52-bit limbs, -p^-1 mod 2^52
constexpr uint64_t P52[5] = {0x8c16d87cfd47ull, 0x916871ca8d3c2ull, 0x181585d97816aull, 0xa029b85045b68ull, 0x30644e72e131ull};
constexpr uint64_t PINV52 = 0x20782e4866389ull;

struct Fe5 {
  double d[5];  // exact integers in [0, 2^52)
};

FP64_HD uint64_t dbits(double x) {
  uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = (uint64_t)__double_as_longlong(x);
#else
  memcpy(&u, &x, 8);
#endif
  return u;
}
FP64_HD double bitsd(uint64_t u) {
  double x;
#if defined(__HIP_DEVICE_COMPILE__)
  x = __longlong_as_double((long long)u);
#else
  memcpy(&x, &u, 8);
#endif
  return x;
}
// exact integer v < 2^52 -> double
FP64_HD double to_d(uint64_t v) { return bitsd(BITS_2P52 | v) - 4503599627370496.0; }

// round-toward-zero fused multiply-add: the device kernel sets MODE.FP_ROUND(f64) = toward zero once; the host check
// sets fesetround(FE_TOWARDZERO)
FP64_HD double fma_rz(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return __builtin_fma(a, b, c);
#endif
}

struct Consts {
  double p[5];   // limbs of p
  double pinv;   // -p^-1 mod 2^52
};

// hi / lo bit patterns of one limb product
FP64_HD void prod(double a, double b, uint64_t& hbits, uint64_t& lbits) {
  const double C1 = 20282409603651670423947251286016.0;                  // 2^104
  const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;  // 2^104 + 2^52 (exact: ulp is 2^52)
  const double h = fma_rz(a, b, C1);
  const double s = C2 - h;   // exact
  const double l = fma_rz(a, b, s);
  hbits = dbits(h);
  lbits = dbits(l);
}

// r = a b R^-1 mod p (R = 2^260), r < 2p for a, b < 2p... in fact < p (1 + 2^-4) — limbs in [0, 2^52), top limb may
// carry the excess.  Everything fully unrolled: the per-column exponent constants fold at compile time.
FP64_HD Fe5 mul(const Fe5& a, const Fe5& b, const Consts& k) {
  uint64_t col[11];
#pragma unroll
  for (int i = 0; i < 11; i++) col[i] = 0;
  // n_lo[c] / n_hi[c]: how many lo / hi patterns column c has received (compile-time after unrolling)
  int n_lo[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, n_hi[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 5; i++) {
#pragma unroll
    for (int j = 0; j < 5; j++) {
      uint64_t h, l;
      prod(a.d[i], b.d[j], h, l);
      col[i + j] += l;
      n_lo[i + j]++;
      col[i + j + 1] += h;
      n_hi[i + j + 1]++;
    }
  }
  uint64_t carry = 0;
#pragma unroll
  for (int r = 0; r < 5; r++) {
    // value of column r so far (exponent fields removed), plus the carry of the column below
    const uint64_t t = col[r] - ((uint64_t)n_lo[r] * BITS_2P52 + (uint64_t)n_hi[r] * BITS_2P104) + carry;
    // q = t p' mod 2^52: the L of the product (t mod 2^52) x p'
    const double td = to_d(t & M52);
    uint64_t hq, lq;
    prod(td, k.pinv, hq, lq);
    const double qd = bitsd(lq) - 4503599627370496.0;   // L as a double (l = 2^52 + L): exact
    uint64_t h0, l0;
    prod(qd, k.p[0], h0, l0);
    const uint64_t done = t + (l0 - BITS_2P52);   // low 52 bits are zero now
    carry = done >> 52;
    col[r + 1] += h0;
    n_hi[r + 1]++;
#pragma unroll
    for (int j = 1; j < 5; j++) {
      uint64_t h, l;
      prod(qd, k.p[j], h, l);
      col[r + j] += l;
      n_lo[r + j]++;
      col[r + j + 1] += h;
      n_hi[r + j + 1]++;
    }
  }
  Fe5 out;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int c = 5 + i;
    const uint64_t t = col[c] - ((uint64_t)n_lo[c] * BITS_2P52 + (uint64_t)n_hi[c] * BITS_2P104) + carry;
    // the top limb keeps whatever is left (the result is < 2^255: nothing beyond limb 4)
    const uint64_t limb = i < 4 ? (t & M52) : t;
    carry = t >> 52;
    out.d[i] = to_d(limb);
  }
  return out;
}

}  // namespace fp64mont
