// GLV endomorphism for BN254 (j = 0): phi(x, y) = (beta x, y) = lambda * (x, y) on the order-r
// subgroups of G1 (beta = BETA_G1) and of the twist G2 (beta = BETA_G2 = BETA_G1^2, same lambda).
// A scalar k (reduced mod r) splits as k = k1 + k2 * lambda (mod r) with |k1|, |k2| < 2^127, so
//     sum k_i P_i = sum |k1_i| (+-P_i) + sum |k2_i| (+-phi(P_i)):
// twice the points, half the scalar length — the same number of bucket additions, but HALF the
// windows: half the bucket-to-window-sum work and a Horner chain of 112 instead of 240 dependent
// doublings (the serial tail that dominates a single MSM's latency).
// Decomposition (lattice basis from the extended Euclid on (r, lambda); constants generated and
// cross-checked by tools/gen_glv.py; model and bound test in tests/test_glv.py):
//     c1 = floor(k * G1C / 2^256),  c2 = floor(k * G2C / 2^256)      (G1C ~ 2^256 b2 / r, G2C ~ 2^256 |b1| / r)
//     k1 = k - c1 * A1 - c2 * A2,   k2 = c1 * B1ABS - c2 * B2
// The reference has no counterpart (its windows cover all 254 bits, VariableBaseMSM.java:137-143);
// the group element computed is the same.
#pragma once
#include "fp29.cuh"

namespace ozk {

struct GlvConsts {
  static constexpr u32 R32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};      // r
  static constexpr u32 G1C[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
  static constexpr u32 G2C[5] = {0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
  static constexpr u32 A1[2] = {0x94d213e3u, 0x89d32568u};
  static constexpr u32 A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
  static constexpr u32 B1ABS[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
  static constexpr u32 B2[2] = {0x94d213e3u, 0x89d32568u};
  // beta in Montgomery form (R = 2^261), 9 x 29-bit limbs
  static constexpr u32 BETA_G1[9] = {0xa337995u, 0x158d1d23u, 0x189c9b98u, 0x12fa4e45u, 0x185faadcu, 0x176f16du, 0xeed93bau, 0x14291140u, 0xc0afeu};
  static constexpr u32 BETA_G2[9] = {0x18ccb791u, 0x175b1c3au, 0xb83d6e2u, 0xe8ed071u, 0x1282bee2u, 0x4220e84u, 0x1fe4017fu, 0x15084d4au, 0x169119u};
};

// r[0..NR) = low NR words of a[0..NA) * b[0..NB)
template <int NA, int NB, int NR>
OZK_HD void mp_mul_lo(const u32* a, const u32* b, u32* r) {
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < NR; k++) {
    u64 lo = carry & 0xffffffffull;
    u64 hi = carry >> 32;
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int j = k - i;
      if (j >= 0 && j < NB) {
        const u64 p = (u64)a[i] * b[j];
        lo += p & 0xffffffffull;
        hi += p >> 32;
      }
    }
    r[k] = (u32)lo;
    carry = hi + (lo >> 32);
  }
}
// full product, NA + NB words
template <int NA, int NB>
OZK_HD void mp_mul(const u32* a, const u32* b, u32* r) { mp_mul_lo<NA, NB, NA + NB>(a, b, r); }

template <int N>
OZK_HD void mp_sub(u32* a, const u32* b) {  // a -= b (mod 2^(32N))
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u64 t = (u64)a[i] - b[i] - br;
    a[i] = (u32)t;
    br = (t >> 32) & 1;
  }
}
template <int N>
OZK_HD bool mp_geq(const u32* a, const u32* b) {
  for (int i = N - 1; i >= 0; i--) {
    if (a[i] > b[i]) return true;
    if (a[i] < b[i]) return false;
  }
  return true;
}
template <int N>
OZK_HD void mp_neg(u32* a) {  // two's complement negate
  u64 c = 1;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (u64)(~a[i]);
    a[i] = (u32)c;
    c >>= 32;
  }
}

// k: any 256-bit value (reduced mod r first).  Outputs |k1|, |k2| as 4 words each (< 2^128) + signs.
OZK_HD void glv_decompose(const u32 (&k_in)[8], u32 (&k1)[4], bool& neg1, u32 (&k2)[4], bool& neg2) {
  u32 k[8];
#pragma unroll
  for (int i = 0; i < 8; i++) k[i] = k_in[i];
  for (int it = 0; it < 6 && mp_geq<8>(k, GlvConsts::R32); it++) mp_sub<8>(k, GlvConsts::R32);
  u32 p1[11], p2[13];
  mp_mul<8, 3>(k, GlvConsts::G1C, p1);
  mp_mul<8, 5>(k, GlvConsts::G2C, p2);
  const u32* c1 = p1 + 8;  // 3 words
  const u32* c2 = p2 + 8;  // 5 words
  // everything modulo 2^160, two's complement: the true values are below 2^128 in magnitude
  u32 t1[5], t2[5], s1[5], s2[5];
  mp_mul_lo<3, 2, 5>(c1, GlvConsts::A1, t1);
  mp_mul_lo<5, 4, 5>(c2, GlvConsts::A2, t2);
#pragma unroll
  for (int i = 0; i < 5; i++) s1[i] = k[i];
  mp_sub<5>(s1, t1);
  mp_sub<5>(s1, t2);                                  // k1 = k - c1 a1 - c2 a2
  mp_mul_lo<3, 4, 5>(c1, GlvConsts::B1ABS, s2);
  mp_mul_lo<5, 2, 5>(c2, GlvConsts::B2, t2);
  mp_sub<5>(s2, t2);                                  // k2 = c1 |b1| - c2 b2
  neg1 = (s1[4] >> 31) != 0;
  neg2 = (s2[4] >> 31) != 0;
  if (neg1) mp_neg<5>(s1);
  if (neg2) mp_neg<5>(s2);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    k1[i] = s1[i];
    k2[i] = s2[i];
  }
}

}  // namespace ozk
