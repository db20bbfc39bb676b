// Packed HBM layouts and wire codecs for curve points, generic over G1 / G2.
//
// HBM layouts (all Montgomery form unless "wire"):
//   coordinate  : CW = 8 (Fq) or 16 (Fq2: c0|c1) little-endian u32 words, value < 2^256
//   affine base : x|y            = 2*CW words (64 B G1 / 128 B G2), canonical (< p);
//                 x = y = 0 encodes the point at infinity
//   Jacobian    : X|Y|Z as UNPACKED 29-bit limbs, 9 (Fq) / 18 (Fq2) words per coordinate, record
//                 padded to 28 / 56 words (112 B / 224 B): intermediate points (buckets, partial
//                 slots, window-sum elements, fixed-base tables) are written and read a few times
//                 by our own kernels only, so they skip the conditional subtractions + bit
//                 packing a 256-bit record would need at every run end of the hot loop and keep
//                 their lazy value bounds (the coordinate TYPES of the curve config)
// wire in  (reference JNI input, VariableBaseMSM.java:221-228): 3*CW words, canonical,
//          non-Montgomery, little-endian.
// wire out (reference JNI output, VariableBaseMSM.cu:1655-1659): per Fq value 64 B LE,
//          upper 32 B zero.
#pragma once
#include "ec.cuh"
#include "quad.cuh"

namespace ozk {

// ---- element <-> packed words (overloaded on the element type; Fq2 overloads in fq2.cuh)
template <class E>
struct ElemTraits;

template <class P, int B>
struct ElemTraits<Fe<P, B>> {
  static constexpr int WORDS = 8;
  // value stored is < 2^256 and known to respect bound B (caller's contract)
  static OZK_HD Fe<P, B> load(const u32* p) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = p[i];
    return unpack<P, B>(w);
  }
  static OZK_HD void store(const Fe<P, B>& e, u32* p) {
    u32 w[8];
    pack(reduce_to<64>(e), w);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = w[i];
  }
  static constexpr int RAW_WORDS = 9;
  static OZK_HD Fe<P, B> load_raw(const u32* p) {
    Fe<P, B> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
  }
  static OZK_HD void store_raw(const Fe<P, B>& e, u32* p) {
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = e.l[i];
  }
  static OZK_HD Fe<P, B> from_wire(const u32* p) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = p[i];
    return Fe<P, B>(to_mont<P>(w));
  }
  static OZK_HD void to_wire(const Fe<P, B>& e, u32* p) {  // canonical, non-Montgomery
    u32 w[8];
    from_mont(e, w);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = w[i];
  }
  // reference return layout: 64 B little-endian per Fq value, upper 32 B zero
  static OZK_HD void to_wire_out(const Fe<P, B>& e, u32* p) {
    to_wire(e, p);
#pragma unroll
    for (int i = 8; i < 16; i++) p[i] = 0;
  }
  static OZK_HD Fe<P, B> from_wire_out(const u32* p) { return from_wire(p); }
  static OZK_HD bool wire_is_zero(const u32* p) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= p[i];
    return o == 0;
  }
  static OZK_HD bool wire_is_one(const u32* p) {
    u32 o = p[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; i++) o |= p[i];
    return o == 0;
  }
};

template <class CV>
struct CurveIO {
  using EX = typename CV::EX;
  using EY = typename CV::EY;
  using EZ = typename CV::EZ;
  using EA = typename CV::EA;
  static constexpr int CW = ElemTraits<EA>::WORDS;
  static constexpr int AFF_WORDS = 2 * CW;
  static constexpr int WIRE_JAC_WORDS = 3 * CW;             // JNI wire input point
  static constexpr int RW = ElemTraits<EA>::RAW_WORDS;      // unpacked words per coordinate
  static constexpr int JAC_WORDS = (3 * RW + 3) / 4 * 4;    // stored Jacobian record (16-B multiple)

  static OZK_HD Aff<EA> load_aff(const u32* p) {
    Aff<EA> q;
    q.x = ElemTraits<EA>::load(p);
    q.y = ElemTraits<EA>::load(p + CW);
    return q;
  }
  static OZK_HD void store_aff(const Aff<EA>& q, u32* p) {
    ElemTraits<EA>::store(q.x, p);
    ElemTraits<EA>::store(q.y, p + CW);
  }
  // stored coordinates keep the bounds of EX / EY / EZ (every kernel of one pipeline uses the
  // same curve config); an all-zero record is the point at infinity (Z = 0)
  static OZK_HD Jac<CV> load_jac(const u32* p) {
    Jac<CV> r;
    r.X = ElemTraits<EX>::load_raw(p);
    r.Y = ElemTraits<EY>::load_raw(p + RW);
    r.Z = ElemTraits<EZ>::load_raw(p + 2 * RW);
    return r;
  }
  static OZK_HD void store_jac(const Jac<CV>& r, u32* p) {
    ElemTraits<EX>::store_raw(r.X, p);
    ElemTraits<EY>::store_raw(r.Y, p + RW);
    ElemTraits<EZ>::store_raw(r.Z, p + 2 * RW);
  }
  // Tagged point record for buckets and partial slots (REC_WORDS words): level 1 dumps its XYZZ
  // accumulator as it is (a run end is a divergent branch of the hot loop: no arithmetic there),
  // later kernels store Jacobian points; readers convert XYZZ with two multiplications.
  static constexpr int REC_TAG = 4 * RW;                        // word index of the format tag
  static constexpr int REC_WORDS = (4 * RW + 1 + 3) / 4 * 4;    // 40 (G1) / 76 (G2) words
  static constexpr u32 TAG_XYZZ = 0, TAG_JAC = 1;
  static OZK_HD void store_rec_xyzz(const Xyzz<CV>& a, u32* p) {
    ElemTraits<typename CV::XX>::store_raw(a.X, p);
    ElemTraits<typename CV::XY>::store_raw(a.Y, p + RW);
    ElemTraits<typename CV::XZZ>::store_raw(a.ZZ, p + 2 * RW);
    ElemTraits<typename CV::XZZZ>::store_raw(a.ZZZ, p + 3 * RW);
    p[REC_TAG] = TAG_XYZZ;
  }
  // a record known to carry TAG_XYZZ (the head / tail pieces the level-1 kernel writes)
  static OZK_HD Xyzz<CV> load_xyzz(const u32* p) {
    Xyzz<CV> a;
    a.X = ElemTraits<typename CV::XX>::load_raw(p);
    a.Y = ElemTraits<typename CV::XY>::load_raw(p + RW);
    a.ZZ = ElemTraits<typename CV::XZZ>::load_raw(p + 2 * RW);
    a.ZZZ = ElemTraits<typename CV::XZZZ>::load_raw(p + 3 * RW);
    return a;
  }
  static OZK_HD void store_rec_jac(const Jac<CV>& r, u32* p) {
    store_jac(r, p);
    p[REC_TAG] = TAG_JAC;
  }
  static OZK_HD Jac<CV> load_rec(const u32* p) {
    if (p[REC_TAG] == TAG_JAC) return load_jac(p);
    Xyzz<CV> a;
    a.X = ElemTraits<typename CV::XX>::load_raw(p);
    a.Y = ElemTraits<typename CV::XY>::load_raw(p + RW);
    a.ZZ = ElemTraits<typename CV::XZZ>::load_raw(p + 2 * RW);
    a.ZZZ = ElemTraits<typename CV::XZZZ>::load_raw(p + 3 * RW);
    return xyzz_to_jac(a);
  }
  static OZK_HD Jac<CV> jac_from_wire(const u32* p) {
    Jac<CV> r;
    r.X = ElemTraits<EX>::from_wire(p);
    r.Y = ElemTraits<EY>::from_wire(p + CW);
    r.Z = ElemTraits<EZ>::from_wire(p + 2 * CW);
    return r;
  }
};

#if defined(__HIPCC__)
// Workgroup barrier with an EXPLICIT wait for this wave's outstanding LDS and global operations.
// __syncthreads() is supposed to imply it, but hipcc (ROCm 7.2, gfx950) leaves the s_waitcnt out when
// the pending operation is a non-returning LDS atomic issued in a loop whose back edge jumps over the
// barrier block: k_sortbig_count read its counters before the last ds_add_u32 of another wave had
// landed and lost exactly one wave-instruction's worth of counts (64, or all 24 of a short last chunk)
// — found with tools/diag_sortbig.py, visible in the ISA as `ds_add_u32 ... s_branch ... s_barrier`
// with no s_waitcnt in between.  Every barrier in this library goes through here.
__device__ __forceinline__ void block_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// hipcc scalarises a provably wave-uniform computation onto the SALU, where the 64-bit
// MAD chains run ~4x slower (measured: 11.6 us per doubling); an opaque zero in a VGPR
// keeps serial single-lane tails on the vector ALU.
__device__ __forceinline__ u32 opaque_zero() {
  u32 z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}
template <class P, int B>
__device__ __forceinline__ Fe<P, B> shfl_xor_el(const Fe<P, B>& v, int m) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = __shfl_xor(v.l[i], m);
  return r;
}
template <class P, int B>
__device__ __forceinline__ Fe<P, B> shfl_down_el(const Fe<P, B>& v, int o) {
  Fe<P, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = __shfl_down(v.l[i], o);
  return r;
}
#endif

}  // namespace ozk
