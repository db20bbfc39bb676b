// Host check of tools/fp64_mont.h: the FP64-FMA Montgomery multiplication against exact integers.
//   g++ -O2 -std=c++17 -mfma -frounding-math tools/fp64_mont_hostcheck.cc -o /tmp/fp64check && /tmp/fp64check
#include <cfenv>
#include <cstdio>
#include <cstdlib>

#include "fp64_mont.h"

typedef unsigned __int128 u128;
using namespace fp64mont;

// 320-bit little-endian words
struct Big { uint64_t w[5]; };
static Big from_limbs52(const uint64_t* l) {   // l[i] may exceed 52 bits slightly (top limb)
  Big r = {{0, 0, 0, 0, 0}};
  for (int i = 0; i < 5; i++) {
    u128 v = (u128)l[i];
    int bit = 52 * i, wi = bit >> 6, sh = bit & 63;
    u128 x = v << sh;
    u128 c = (u128)r.w[wi] + (uint64_t)x;
    r.w[wi] = (uint64_t)c;
    u128 hi = (x >> 64) + (c >> 64);
    for (int k = wi + 1; k < 5 && hi; k++) {
      u128 t = (u128)r.w[k] + (uint64_t)hi;
      r.w[k] = (uint64_t)t;
      hi = (hi >> 64) + (t >> 64);
    }
  }
  return r;
}
static int cmp(const Big& a, const Big& b) {
  for (int i = 4; i >= 0; i--)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}
static Big add(const Big& a, const Big& b) {
  Big r;
  u128 c = 0;
  for (int i = 0; i < 5; i++) {
    c += (u128)a.w[i] + b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return r;
}
static Big sub(const Big& a, const Big& b) {
  Big r;
  __int128 c = 0;
  for (int i = 0; i < 5; i++) {
    c += (__int128)a.w[i] - b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return r;
}
static Big half_mod(const Big& a, const Big& P) {   // a / 2 mod P, a < P
  Big t = a;
  if (t.w[0] & 1) t = add(t, P);
  for (int i = 0; i < 5; i++) t.w[i] = (t.w[i] >> 1) | (i < 4 ? t.w[i + 1] << 63 : 0);
  return t;
}
static Big mulmod(const Big& a, const Big& b, const Big& P) {   // a, b < P: double-and-add
  Big r = {{0, 0, 0, 0, 0}};
  for (int bit = 319; bit >= 0; bit--) {
    r = add(r, r);
    if (cmp(r, P) >= 0) r = sub(r, P);
    if ((b.w[bit >> 6] >> (bit & 63)) & 1) {
      r = add(r, a);
      if (cmp(r, P) >= 0) r = sub(r, P);
    }
  }
  return r;
}
static Big reduce(Big a, const Big& P) {
  while (cmp(a, P) >= 0) a = sub(a, P);
  return a;
}

int main() {
  fesetround(FE_TOWARDZERO);
  Consts k;
  for (int i = 0; i < 5; i++) k.p[i] = to_d(P52[i]);
  k.pinv = to_d(PINV52);
  const Big P = from_limbs52(P52);
  uint64_t s = 0x243F6A8885A308D3ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  int bad = 0;
  for (int it = 0; it < 20000; it++) {
    uint64_t al[5], bl[5];
    for (int i = 0; i < 5; i++) { al[i] = rnd() & M52; bl[i] = rnd() & M52; }
    al[4] &= (1ull << 46) - 1;   // < 2^254: below 2p
    bl[4] &= (1ull << 46) - 1;
    if (it == 0) for (int i = 0; i < 5; i++) { al[i] = P52[i]; bl[i] = P52[i]; }          // p x p
    if (it == 1) for (int i = 0; i < 5; i++) { al[i] = M52; bl[i] = M52; al[4] = bl[4] = (1ull << 47) - 1; }   // 2^255 - 1
    if (it == 2) for (int i = 0; i < 5; i++) { al[i] = 0; bl[i] = M52; }
    Fe5 a, b;
    for (int i = 0; i < 5; i++) { a.d[i] = to_d(al[i]); b.d[i] = to_d(bl[i]); }
    // a chain of three multiplications: the (slightly unreduced) output must be a valid input
    Fe5 r = mul(a, b, k);
    Fe5 r2 = mul(r, b, k);
    uint64_t rl[5], r2l[5];
    for (int i = 0; i < 5; i++) { rl[i] = (uint64_t)r.d[i]; r2l[i] = (uint64_t)r2.d[i]; if (r.d[i] != (double)rl[i]) bad++; }
    const Big A = reduce(from_limbs52(al), P), B = reduce(from_limbs52(bl), P);
    Big want = mulmod(A, B, P);
    for (int h = 0; h < 260; h++) want = half_mod(want, P);
    Big got = reduce(from_limbs52(rl), P);
    if (cmp(got, want) != 0) { bad++; if (bad < 5) printf("mismatch at %d\n", it); }
    Big want2 = mulmod(want, B, P);
    for (int h = 0; h < 260; h++) want2 = half_mod(want2, P);
    Big got2 = reduce(from_limbs52(r2l), P);
    if (cmp(got2, want2) != 0) { bad++; if (bad < 5) printf("chain mismatch at %d\n", it); }
    // bound: result < 2^255
    if (rl[4] >> 47) { bad++; if (bad < 5) printf("result too large at %d\n", it); }
  }
  printf(bad ? "fp64 montgomery: %d FAILURES\n" : "fp64 montgomery: 20000 products + chains equal a b R^-1 mod p (R = 2^260)\n", bad);
  return bad != 0;
}
