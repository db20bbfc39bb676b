/* ozk_oracle.c — plain-C CPU restatement of the reference's serial hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): linked/executed only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never by the product.
 *
 * Restates, single-threaded:
 *   Fp add/sub/mul/square/inverse      algebra/fields/Fp.java:38-92   (mul-then-mod; here
 *                                       Montgomery internally — exact, so same values)
 *   BNG1.add / twice / toAffine        algebra/curves/barreto_naehrig/BNG1.java:38-172
 *   VariableBaseMSM.pippengerMSM       algebra/msm/VariableBaseMSM.java:134-188
 *                                       (window c = L - L/3, L = max(1, log2 n); digit 0
 *                                       skipped; running sum; c doublings per window)
 *   FFTAuxiliary.serialRadix2FFT       algebra/fft/FFTAuxiliary.java:100-123
 *   FixedBaseMSM.serialMSM semantics   algebra/msm/FixedBaseMSM.java:141-167
 * Pinned against oracle/bn254.py (exact Python ints == BigInteger) and the reference's
 * KATs in tests/test_oracle.py.  "port" CPU baseline: this is a C stand-in for the Java
 * BigInteger path and is considerably FASTER than it (64-bit Montgomery vs BigInteger).
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libozk_oracle.so oracle/ozk_oracle.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;   /* Montgomery form, < p */

typedef struct {
  uint64_t p[4];
  uint64_t pinv;      /* -p^-1 mod 2^64 */
  uint64_t r2[4];     /* 2^512 mod p */
  uint64_t one[4];    /* 2^256 mod p */
} field_t;

/* BN254aFqParameters.java:33 / BN254aFrParameters.java:33 */
static const field_t FQ = {
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0x87d20782e4866389ULL,
  {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
  {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
static const field_t FR = {
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0xc2e1f593efffffffULL,
  {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
  {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};

static int ge(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
static void sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - (uint64_t)br; r[i] = (uint64_t)t; br = (t >> 64) & 1; }
}
static fe f_add(const field_t* F, fe a, fe b) {
  fe r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
  if (c || ge(r.v, F->p)) sub_n(r.v, r.v, F->p);
  return r;
}
static fe f_sub(const field_t* F, fe a, fe b) {
  fe r; u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a.v[i] - b.v[i] - (uint64_t)br; r.v[i] = (uint64_t)t; br = (t >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.v[i] + F->p[i]; r.v[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
static fe f_mul(const field_t* F, fe a, fe b) {   /* CIOS Montgomery */
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->pinv;
    c = (u128)m * F->p[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe r; memcpy(r.v, t, 32);
  if (t[4] || ge(r.v, F->p)) sub_n(r.v, r.v, F->p);
  return r;
}
static fe f_sqr(const field_t* F, fe a) { return f_mul(F, a, a); }
static int f_is_zero(fe a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static int f_eq(fe a, fe b) { return memcmp(a.v, b.v, 32) == 0; }
static fe f_from_bytes(const field_t* F, const uint8_t* b) {  /* 32 B LE canonical -> Montgomery */
  fe a, r2; memcpy(a.v, b, 32); memcpy(r2.v, F->r2, 32);
  return f_mul(F, a, r2);
}
static void f_to_bytes(const field_t* F, fe a, uint8_t* b) {
  fe one = {{1, 0, 0, 0}};
  fe r = f_mul(F, a, one);
  memcpy(b, r.v, 32);
}
static fe f_one(const field_t* F) { fe r; memcpy(r.v, F->one, 32); return r; }
static fe f_zero(void) { fe r = {{0, 0, 0, 0}}; return r; }
static fe f_pow(const field_t* F, fe a, const uint64_t* e) {
  fe r = f_one(F);
  for (int i = 255; i >= 0; i--) {
    r = f_sqr(F, r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = f_mul(F, r, a);
  }
  return r;
}
static fe f_inv(const field_t* F, fe a) {  /* Fp.java:90-92 modInverse; a^(p-2) */
  uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0};
  sub_n(e, F->p, two);
  return f_pow(F, a, e);
}

/* ------------------------------------------------------------------ G1 (BNG1.java) */
typedef struct { fe X, Y, Z; } g1;
static g1 g1_zero(void) { g1 r; r.X = f_zero(); r.Y = f_one(&FQ); r.Z = f_zero(); return r; }  /* BN254aG1Parameters.java:52-55 */
static int g1_is_zero(const g1* p) { return f_is_zero(p->Z); }                                  /* BNG1.java:103-105 */

static g1 g1_twice(const g1* p) {  /* BNG1.java:133-161 */
  if (g1_is_zero(p)) return *p;
  const field_t* F = &FQ;
  fe A = f_sqr(F, p->X), B = f_sqr(F, p->Y), C = f_sqr(F, B);
  fe D = f_sub(F, f_sub(F, f_sqr(F, f_add(F, p->X, B)), A), C);
  D = f_add(F, D, D);
  fe E = f_add(F, f_add(F, A, A), A);
  fe Fv = f_sqr(F, E);
  g1 r;
  r.X = f_sub(F, Fv, f_add(F, D, D));
  fe eightC = f_add(F, C, C); eightC = f_add(F, eightC, eightC); eightC = f_add(F, eightC, eightC);
  r.Y = f_sub(F, f_mul(F, E, f_sub(F, D, r.X)), eightC);
  fe Y1Z1 = f_mul(F, p->Y, p->Z);
  r.Z = f_add(F, Y1Z1, Y1Z1);
  return r;
}
static g1 g1_add(const g1* p, const g1* q) {  /* BNG1.java:38-97 */
  if (g1_is_zero(p)) return *q;
  if (g1_is_zero(q)) return *p;
  const field_t* F = &FQ;
  fe Z1Z1 = f_sqr(F, p->Z), Z2Z2 = f_sqr(F, q->Z);
  fe U1 = f_mul(F, p->X, Z2Z2), U2 = f_mul(F, q->X, Z1Z1);
  fe Z1c = f_mul(F, p->Z, Z1Z1), Z2c = f_mul(F, q->Z, Z2Z2);
  fe S1 = f_mul(F, p->Y, Z2c), S2 = f_mul(F, q->Y, Z1c);
  if (f_eq(U1, U2) && f_eq(S1, S2)) return g1_twice(p);
  fe H = f_sub(F, U2, U1), S2mS1 = f_sub(F, S2, S1);
  fe I = f_sqr(F, f_add(F, H, H));
  fe J = f_mul(F, H, I);
  fe r = f_add(F, S2mS1, S2mS1);
  fe V = f_mul(F, U1, I);
  g1 o;
  o.X = f_sub(F, f_sub(F, f_sqr(F, r), J), f_add(F, V, V));
  fe S1J = f_mul(F, S1, J);
  o.Y = f_sub(F, f_mul(F, r, f_sub(F, V, o.X)), f_add(F, S1J, S1J));
  o.Z = f_mul(F, f_sub(F, f_sub(F, f_sqr(F, f_add(F, p->Z, q->Z)), Z1Z1), Z2Z2), H);
  return o;
}
static g1 g1_to_affine(const g1* p) {  /* BNG1.java:163-172 */
  if (g1_is_zero(p)) return g1_zero();
  const field_t* F = &FQ;
  fe zi = f_inv(F, p->Z), z2 = f_sqr(F, zi), z3 = f_mul(F, z2, zi);
  g1 r; r.X = f_mul(F, p->X, z2); r.Y = f_mul(F, p->Y, z3); r.Z = f_one(F);
  return r;
}
static g1 g1_from_wire(const uint8_t* b) {
  g1 r; r.X = f_from_bytes(&FQ, b); r.Y = f_from_bytes(&FQ, b + 32); r.Z = f_from_bytes(&FQ, b + 64);
  return r;
}
static void g1_to_out_le(const g1* p, uint8_t* out) {  /* 3 x 64 B LE, upper half zero */
  memset(out, 0, 192);
  f_to_bytes(&FQ, p->X, out); f_to_bytes(&FQ, p->Y, out + 64); f_to_bytes(&FQ, p->Z, out + 128);
}

static int test_bit(const uint8_t* s, int bit) { return bit < 256 ? (s[bit >> 3] >> (bit & 7)) & 1 : 0; }

static int java_log2(int x) { return (int)(log((double)x) / log(2.0)); }  /* common/MathUtils.java:8-10 */

int oracle_pippenger_window(int n) {  /* VariableBaseMSM.java:137-139 */
  int l = java_log2(n); if (l < 1) l = 1;
  return l - l / 3;
}

/* VariableBaseMSM.pippengerMSM (VariableBaseMSM.java:134-188), result affine-normalised.
 * bases: n x 96 B wire, scalars: n x 32 B LE, out: 192 B. */
int oracle_pippenger_g1(const uint8_t* bases, const uint8_t* scalars, int n, int num_bits, uint8_t* out) {
  const int c = oracle_pippenger_window(n);
  const int num_buckets = 1 << c, num_groups = (num_bits + c - 1) / c;
  g1* pts = (g1*)malloc(sizeof(g1) * (size_t)n);
  g1* buckets = (g1*)malloc(sizeof(g1) * (size_t)num_buckets);
  if (!pts || !buckets) { free(pts); free(buckets); return -1; }
  for (int i = 0; i < n; i++) pts[i] = g1_from_wire(bases + (size_t)i * 96);
  g1 result = g1_zero();
  for (int k = num_groups - 1; k >= 0; k--) {
    for (int b = 0; b < num_buckets; b++) buckets[b] = g1_zero();
    for (int i = 0; i < n; i++) {
      int id = 0;
      for (int j = 0; j < c; j++) if (test_bit(scalars + (size_t)i * 32, k * c + j)) id |= 1 << j;  /* :157-160 */
      if (id == 0) continue;                                                                        /* :163 */
      buckets[id] = g1_add(&buckets[id], &pts[i]);                                                  /* :168 */
    }
    g1 running = g1_zero();
    for (int i = num_buckets - 1; i > 0; i--) {                                                     /* :171-177 */
      running = g1_add(&running, &buckets[i]);
      result = g1_add(&result, &running);
    }
    if (k > 0) for (int i = 0; i < c; i++) result = g1_twice(&result);                              /* :180-184 */
  }
  g1 a = g1_to_affine(&result);
  g1_to_out_le(&a, out);
  free(pts); free(buckets);
  return 0;
}

/* NaiveMSM.variableBaseMSM (NaiveMSM.java:33-46) with AbstractGroup.mul (AbstractGroup.java:29-51) */
int oracle_naive_g1(const uint8_t* bases, const uint8_t* scalars, int n, uint8_t* out) {
  g1 result = g1_zero();
  for (int i = 0; i < n; i++) {
    g1 base = g1_from_wire(bases + (size_t)i * 96);
    g1 r = g1_zero();
    int found = 0;
    for (int b = 255; b >= 0; b--) {
      if (found) r = g1_twice(&r);
      if (test_bit(scalars + (size_t)i * 32, b)) { found = 1; r = g1_add(&r, &base); }
    }
    result = g1_add(&result, &r);
  }
  g1 a = g1_to_affine(&result);
  g1_to_out_le(&a, out);
  return 0;
}

/* out[i] = s_i * base, affine-normalised, 192 B BE per point: the value FixedBaseMSM.serialMSM
 * (FixedBaseMSM.java:141-167) produces for the first outerc windows of the scalar. */
int oracle_fixed_base_g1(const uint8_t* base, const uint8_t* scalars, int n, int outerc, int window, uint8_t* out) {
  g1 B = g1_from_wire(base);
  const int bits = outerc * window;
  for (int i = 0; i < n; i++) {
    g1 r = g1_zero();
    int found = 0;
    for (int b = (bits < 256 ? bits : 256) - 1; b >= 0; b--) {
      if (found) r = g1_twice(&r);
      if (test_bit(scalars + (size_t)i * 32, b)) { found = 1; r = g1_add(&r, &B); }
    }
    g1 a = g1_to_affine(&r);
    uint8_t le[192];
    g1_to_out_le(&a, le);
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 64; j++) out[(size_t)i * 192 + k * 64 + j] = le[k * 64 + 63 - j];
  }
  return 0;
}

/* ------------------------------------------------------------------ FFT over Fr */
static int bitreverse(int n, int bits) {  /* common/MathUtils.java:43-53 */
  int count = bits - 1, reverse = n;
  n >>= 1;
  while (n > 0) { reverse = (reverse << 1) | (n & 1); n >>= 1; count--; }
  return (int)(((unsigned)reverse << count) & ((1u << bits) - 1));
}
/* FFTAuxiliary.serialRadix2FFT (FFTAuxiliary.java:100-123); in: n x 32 B LE, out: n x 64 B LE */
int oracle_fft_fr(const uint8_t* in, int n, const uint8_t* omega32, uint8_t* out) {
  const field_t* F = &FR;
  int logn = java_log2(n);
  if ((1 << logn) != n) return -1;
  fe* a = (fe*)malloc(sizeof(fe) * (size_t)n);
  if (!a) return -1;
  for (int i = 0; i < n; i++) a[i] = f_from_bytes(F, in + (size_t)i * 32);
  fe omega = f_from_bytes(F, omega32);
  if (n > 1) {
    for (int k = 0; k < n; k++) { int rk = bitreverse(k, logn); if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; } }
    int m = 1;
    for (int s = 1; s <= logn; s++) {
      uint64_t e[4] = {(uint64_t)(n / (2 * m)), 0, 0, 0};
      fe w_m = f_pow(F, omega, e);
      for (int k = 0; k < n; k += 2 * m) {
        fe w = f_one(F);
        for (int j = 0; j < m; j++) {
          fe t = f_mul(F, w, a[k + j + m]);
          a[k + j + m] = f_sub(F, a[k + j], t);
          a[k + j] = f_add(F, a[k + j], t);
          w = f_mul(F, w, w_m);
        }
      }
      m *= 2;
    }
  }
  memset(out, 0, (size_t)n * 64);
  for (int i = 0; i < n; i++) f_to_bytes(F, a[i], out + (size_t)i * 64);
  free(a);
  return 0;
}

/* x_i * b mod r; in: (n+1) x 32 B LE (last = b); out: n x 64 B BE (FixedBaseMSM.cu:1241-1266) */
int oracle_field_batch_mul(const uint8_t* in, int n, uint8_t* out) {
  const field_t* F = &FR;
  fe b = f_from_bytes(F, in + (size_t)n * 32);
  for (int i = 0; i < n; i++) {
    uint8_t le[32];
    f_to_bytes(F, f_mul(F, f_from_bytes(F, in + (size_t)i * 32), b), le);
    memset(out + (size_t)i * 64, 0, 32);
    for (int j = 0; j < 32; j++) out[(size_t)i * 64 + 32 + j] = le[31 - j];
  }
  return 0;
}

/* raw field op for pinning the C arithmetic against Python ints: op 0 mul 1 add 2 sub 3 inv */
int oracle_field_op(int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  const field_t* F = field ? &FR : &FQ;
  fe x = f_from_bytes(F, a), y = f_from_bytes(F, b), r;
  switch (op) { case 0: r = f_mul(F, x, y); break; case 1: r = f_add(F, x, y); break;
                case 2: r = f_sub(F, x, y); break; default: r = f_inv(F, x); }
  f_to_bytes(F, r, out);
  return 0;
}
