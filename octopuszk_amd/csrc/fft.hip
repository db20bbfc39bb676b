// Radix-2 FFT over BN254 Fr on gfx950: kernels, host driver, C ABI (include/ozk.h).
//
// Replaces cuda_fft_first_step / cuda_fft_second_step / best_fft and the JNI native of
// algebra.fft.FFTAuxiliary (algebra_fft_FFTAuxiliary.cu:70-260) and computes exactly what
// FFTAuxiliary.serialRadix2FFT does (FFTAuxiliary.java:100-123): bit-reversal permutation,
// then log2(n) decimation-in-time stages with w_m = omega^(n/2m); out[i] = sum_j in[j] *
// omega^(i*j) in natural order.  omega is an argument (SerialFFT passes omega or omega^-1,
// SerialFFT.java:75-95), so inverse and coset transforms are the same entry point.
//
// Design (HBM-lean, integer-ALU-bound):
//  * data stays in PLAIN (non-Montgomery) representation; only the twiddles are in
//    Montgomery form, so mont_mul(twiddle, y) is the plain product — no conversions.
//  * the reference recomputes two modular exponentiations per butterfly
//    (FFT.cu:127,138); here omega^t, t < n/2, is built once per call by a two-level
//    table (2 x <=2048 square-and-multiply lanes, then one multiply per entry).
//  * log2(n) stages run in ceil(log2(n)/8) passes; a pass keeps a tile of 2^K x T
//    elements in LDS (limb-major, conflict-free) for K stages with NO modular reduction
//    between stages (value bounds are tracked at compile time), reads/writes HBM once,
//    in >= 128-byte contiguous segments.  The bit reversal is fused into the first pass.
#include <algorithm>
#include <new>
#include <vector>

#include "curve.cuh"
#include "host_ctx.h"
#include "pin_cache.h"

namespace ozk {

using FrP = FrParams;
// Elements per workgroup tile: 1024 (36 KiB of LDS at 9 words each; 256 threads, four elements per thread and stage
// pair), at most 8 stages per pass: 2^22 is three passes.  A 2048-element tile (two passes of 11 stages) and a
// 512-element one were carried as options through round 3 and measured slower (0.62-0.67 / 0.57 ms against 0.545).
constexpr int FFT_TILE_SMALL = 1024;
constexpr int FFT_THREADS = 256;
constexpr int TW_LO = 2048;

// ---- twiddle table -------------------------------------------------------
// small[0..lo) = omega^i, small[lo..lo+hi) = omega^(i*lo), Montgomery form, packed 8 words
__global__ void __launch_bounds__(256) k_tw_small(const u32* __restrict__ omega_wire, int lo, int hi,
                                                  u32* __restrict__ small) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lo + hi) return;
  u32 w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = omega_wire[i];
  const Fe<FrP, 32> base = Fe<FrP, 32>(to_mont<FrP>(w));
  const unsigned e = (t < lo) ? (unsigned)t : (unsigned)(t - lo) * (unsigned)lo;
  Fe<FrP, 32> r = fe_one<FrP>();
  for (int b = 31; b >= 0; b--) {
    r = Fe<FrP, 32>(sqr(r));
    if ((e >> b) & 1) r = Fe<FrP, 32>(mul(r, base));
  }
  u32 o[8];
  pack(canonical(r), o);
#pragma unroll
  for (int i = 0; i < 8; i++) small[(size_t)t * 8 + i] = o[i];
}
__global__ void __launch_bounds__(256) k_tw_full(const u32* __restrict__ small, int lo, int half,
                                                 u32* __restrict__ tw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= half) return;
  using ET = ElemTraits<Fe<FrP, 16>>;
  const auto a = ET::load(small + (size_t)(t % lo) * 8);
  const auto b = ET::load(small + (size_t)(lo + t / lo) * 8);
  u32 o[8];
  pack(canonical(mul(a, b)), o);
  uint4* dst = reinterpret_cast<uint4*>(tw + (size_t)t * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// The twiddle PYRAMID (round 4): level s holds omega^(j 2^s), j < n / 2^(s+1), CONTIGUOUS, at entry n - (n >> s) of
// the table (level 0 = the plain table omega^t, t < n / 2, at entry 0; n - 1 entries in all, twice the plain table).
// A stage whose butterflies step through the plain table with stride 2^s reads level s instead: the last pass of a
// 2^22 transform touched every 128-byte line of the 64 MiB table three times (strides 1, 2, 4: 64 MiB fetched each
// for 64, 32 and 16 MiB of twiddles) and half of it once more — 246 of the pass's 630 MiB
// (profiles/r03_pmc_hbm_fft_summary.csv); through the levels the same stages fetch 64 + 32 + 16 + 8 + 4 + 2 MiB.
__global__ void __launch_bounds__(256) k_tw_pyramid(u32* __restrict__ tw, int n, int logn) {
  const long long o = (long long)n / 2 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o > (long long)n - 2) return;
  const unsigned r = (unsigned)((long long)n - o);            // 2 .. n / 2: distance to the end of the table
  const int k = 32 - __clz(r - 1u);                           // ceil(log2 r): the level's size is 2^k / 2 ... entries
  const int s = logn - k;
  const long long base = (long long)n - ((long long)n >> s);
  const long long j = o - base;
  const uint4* src = reinterpret_cast<const uint4*>(tw + (size_t)(j << s) * 8);
  uint4* dst = reinterpret_cast<uint4*>(tw + (size_t)o * 8);
  dst[0] = src[0];
  dst[1] = src[1];
}

// ---- one pass of K stages over an LDS tile ---------------------------------
struct PassArgs {
  const u32* in;    // FIRST: wire input (n x 8 words, plain canonical); else packed workspace
  u32* out;         // LAST: wire out (n x out_stride words LE); else packed workspace (n x 8 words)
  int out_stride;   // 16: the JNI's 64-byte elements (upper half zero); 8: compact 32-byte elements
  const u32* tw;    // the twiddle pyramid (k_tw_pyramid): omega^t, t < n/2, then its stride-2^s subsamples; Montgomery, packed
  int n, logn;
  int sbits;        // stages already done = log2 of the butterfly distance entering this pass
  int K;            // stages in this pass
  int pyr;          // 1: twiddles from the pyramid's levels (default), 0: from level 0 with a stride
  const u32* scale; // LAST pass only, or null: out[i] is multiplied by scale[i] (n x 8 words, Montgomery form) on its
                    // way out — the coset / 1-over-m scalings of the witness map ride on the transform before them
};

// omega^(j 2^s): entry j of level s of the pyramid
__device__ __forceinline__ const u32* tw_at(const PassArgs& a, u32 j, int s) {
  if (!a.pyr) return a.tw + ((size_t)j << s) * 8;   // (OZK_FFT_TW_PYRAMID=0: the plain table with a stride, rounds 1-3)
  return a.tw + ((size_t)a.n - ((size_t)a.n >> s) + (size_t)j) * 8;
}

template <int B, int TILE>
__device__ __forceinline__ Fe<FrP, B> lds_load(const u32* lds, int e) {
  Fe<FrP, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = lds[i * TILE + e];
  return r;
}
template <int TILE, int B>
__device__ __forceinline__ void lds_store(u32* lds, int e, const Fe<FrP, B>& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) lds[i * TILE + e] = v.l[i];
}

// Index of tile element (mid, ul) of workgroup `blk` in the n-element array.
//   later passes (sbits > 0): u = blk T + ul enumerates the (hi, lo) pairs around the K bits this pass works
//     on: i = hi << (sbits + K) | mid << sbits | lo — runs of T consecutive elements (T x 32 B segments);
//   FIRST pass (sbits = 0): ul sits in the TOP log2(T) bits, i = ul << (logn - log2 T) | blk << K | mid, so that
//     the bit-reversed SOURCES of a tile come in runs of T consecutive elements too (the reversal turns the top
//     bits into the lowest ones) instead of 1024 isolated 32-byte reads, and the tile's outputs are T
//     contiguous runs of 2^K elements.
template <bool FIRST>
__device__ __forceinline__ u32 tile_index(const PassArgs& a, u32 blk, int T, int logT, u32 mid, u32 ul) {
  if (FIRST) return (ul << (a.logn - logT)) | (blk << a.K) | mid;
  const u32 u = blk * (u32)T + ul;
  const u32 hi = u >> a.sbits, lo = u & ((1u << a.sbits) - 1u);
  return (hi << (a.sbits + a.K)) | (mid << a.sbits) | lo;
}
// the `lo` part of the twiddle exponent for tile element ul (0 in the first pass: sbits = 0)
template <bool FIRST>
__device__ __forceinline__ u32 tile_lo(const PassArgs& a, u32 blk, int T, u32 ul) {
  if (FIRST) return 0u;
  return (blk * (u32)T + ul) & ((1u << a.sbits) - 1u);
}

// ONE stage (Q, 1-based inside the pass) of radix-2 butterflies: TILE / 2 butterflies over TILE / 4 threads
template <bool FIRST, int Q, int BIN, int TILE>
__device__ __forceinline__ void fft_stage1(u32* lds, const PassArgs& a, int T, int logT) {
  for (u32 b = threadIdx.x; b < TILE / 2; b += TILE / 4) {
    const u32 ul = b & (u32)(T - 1);
    const u32 r = b >> logT;
    const u32 low = r & ((1u << (Q - 1)) - 1u);
    const u32 mid0 = ((r >> (Q - 1)) << Q) | low;
    const u32 mid1 = mid0 | (1u << (Q - 1));
    const u32 j = (low << a.sbits) + tile_lo<FIRST>(a, blockIdx.x, T, ul);
    const auto w = ElemTraits<Fe<FrP, 16>>::load(tw_at(a, j, a.logn - a.sbits - Q));
    const u32 e0 = mid0 * T + ul, e1 = mid1 * T + ul;
    const auto x = lds_load<BIN, TILE>(lds, e0);
    const auto y = lds_load<BIN, TILE>(lds, e1);
    const auto t = mul(w, y);                       // plain product (w is Montgomery)
    lds_store<TILE>(lds, e0, Fe<FrP, BIN + 32>(add(x, t)));
    lds_store<TILE>(lds, e1, Fe<FrP, BIN + 32>(sub(x, t)));
  }
  block_sync();
}

// one tile element straight from / to global memory (the first / last stages of a pass skip LDS)
template <bool FIRST, int TILE>
__device__ __forceinline__ void gload(const PassArgs& a, int T, int logT, u32 mid, u32 ul, uint4& v0, uint4& v1) {
  const u32 i = tile_index<FIRST>(a, blockIdx.x, T, logT, mid, ul);
  u32 src = i;
  if (FIRST) src = __brev(i) >> (32 - a.logn);
  const uint4* sp = reinterpret_cast<const uint4*>(a.in + (size_t)src * 8);
  v0 = sp[0];
  v1 = sp[1];
}
__device__ __forceinline__ Fe<FrP, 96> gunpack(const uint4& v0, const uint4& v1) {
  const u32 w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  return Fe<FrP, 96>(unpack<FrP, 85>(w));   // FIRST: arbitrary 256-bit wire value (85); later passes: stored < 4p
}
// LAST: 0 = a pass that is not the last (packed < 17 p / 16 to the workspace), 1 = the last pass (canonical, wire
// format), 2 = the last pass with the output scaling.  A template parameter, not a flag: the kernels of the middle
// passes do not carry the registers of the scaling product (the register count of a kernel is that of its worst path).
template <bool FIRST, int TILE, int LAST, int B>
__device__ __forceinline__ void gstore(const PassArgs& a, int T, int logT, u32 mid, u32 ul, const Fe<FrP, B>& v) {
  const u32 i = tile_index<FIRST>(a, blockIdx.x, T, logT, mid, ul);
  u32 o[8];
  if constexpr (LAST != 0) {
    if constexpr (LAST == 2) pack(canonical(mul(v, ElemTraits<Fe<FrP, 16>>::load(a.scale + (size_t)i * 8))), o);
    else pack(canonical_q(v), o);
    uint4* dst = reinterpret_cast<uint4*>(a.out + (size_t)i * a.out_stride);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    if (a.out_stride == 16) {
      dst[2] = make_uint4(0, 0, 0, 0);
      dst[3] = make_uint4(0, 0, 0, 0);
    }
  } else {
    pack(reduce_q(v), o);
    uint4* dst = reinterpret_cast<uint4*>(a.out + (size_t)i * 8);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
  }
}

// a + b and a - b WITHOUT the carry pass (limbs may exceed 29 bits; BIAS limbs are < 2^30, a product's < 2^29).
// Where a stage pair uses them, and why no limb overflows 32 bits and no product column 64:
//   tile elements in LDS (the y outputs of a pair)   x < 1.5 x 2^30   (carried sum / difference +- a product)
//   p, q = w1 * x01, w1 * x11                        one factor < 1.5 x 2^30, the twiddle normalised
//   a0, a1 = x00 +- p                                CARRIED (the two values that are never multiplied in this pair)
//   b0 = x10 + q < 2^31, b1 = x10 + BIAS - q < 2.5 x 2^30     uncarried; each is one factor of u, v = w2 * b:
//                                                    9 (2.5 x 2^30 x 2^29 + 2^58) + carry < 1.7 x 2^63
//   y = a +- u (to LDS)                              uncarried: < 2^29 + 2^30 = 1.5 x 2^30; carried when it leaves
//                                                    the tile for global memory
// The subtrahend is always a product (normalised).  Two carry passes per pair instead of eight: 0.59 -> 0.55 ms at 2^22.
template <int B1, int B2>
__device__ __forceinline__ Fe<FrP, B1 + B2> add_nc(const Fe<FrP, B1>& a, const Fe<FrP, B2>& b) {
  Fe<FrP, B1 + B2> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <int B1, int B2>
__device__ __forceinline__ auto sub_nc(const Fe<FrP, B1>& a, const Fe<FrP, B2>& b) {
  constexpr int K = B2 / 16 + 1;
  static_assert(K <= FE_MAXK, "sub bias table too small");
  Fe<FrP, B1 + 16 * K> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (FrP::BIAS[K][i] - b.l[i]);
  return r;
}

// TWO stages (Q, Q + 1) in registers: every thread owns the four elements that differ in bits Q - 1 and Q of
// `mid` — one LDS round trip, one barrier and one index computation for two stages (round 1 ran every stage
// through LDS; without the multiplications that skeleton alone took 0.29 of the 0.75 ms at 2^22,
// profiles/r02_fft_experiments.txt).  Twiddles: stage Q pairs (x00, x01) and (x10, x11) with the same w1;
// stage Q + 1 pairs (., x10') with w2a and (., x11') with w2b (their `low` differs in bit Q - 1).
// SRC_G: the four inputs come straight from global memory (the first pair of a pass, Q = 1);
// DST_G: the four outputs go straight to global memory (the last pair) — two LDS round trips fewer per pass.
template <bool FIRST, int Q, int BIN, int TILE, bool SRC_G, bool DST_G, int LAST>
__device__ __forceinline__ void fft_stage2(u32* lds, const PassArgs& a, int T, int logT) {
  using TW = ElemTraits<Fe<FrP, 16>>;
  const u32 g = threadIdx.x;                         // TILE / 4 groups, one per thread
  // lane -> (ul, r): ul fastest, so that a wave's global accesses come in runs of T consecutive elements; the
  // first pass writes consecutive `mid` to consecutive addresses instead (tile_index), so there r is fastest
  const int logR = 31 - __clz(TILE / 4 / T);
  const u32 ul = (FIRST && DST_G) ? (g >> logR) : (g & (u32)(T - 1));
  const u32 r = (FIRST && DST_G) ? (g & ((1u << logR) - 1u)) : (g >> logT);
  const u32 low = r & ((1u << (Q - 1)) - 1u);
  const u32 m00 = ((r >> (Q - 1)) << (Q + 1)) | low;
  const u32 m01 = m00 | (1u << (Q - 1)), m10 = m00 | (1u << Q), m11 = m01 | (1u << Q);
  const u32 lo = tile_lo<FIRST>(a, blockIdx.x, T, ul);
  const int sh1 = a.logn - a.sbits - Q;   // stage Q steps through omega^t with stride 2^sh1, stage Q + 1 with half of it
  const u32 j1 = (low << a.sbits) + lo;
  const u32 j2b = ((low | (1u << (Q - 1))) << a.sbits) + lo;
  const u32 e00 = m00 * T + ul, e01 = m01 * T + ul, e10 = m10 * T + ul, e11 = m11 * T + ul;
  Fe<FrP, BIN> x00, x01, x10, x11;
  if constexpr (SRC_G) {
    static_assert(BIN >= 96, "bound of a freshly loaded element");
    uint4 v0[4], v1[4];
    gload<FIRST, TILE>(a, T, logT, m00, ul, v0[0], v1[0]);
    gload<FIRST, TILE>(a, T, logT, m01, ul, v0[1], v1[1]);
    gload<FIRST, TILE>(a, T, logT, m10, ul, v0[2], v1[2]);
    gload<FIRST, TILE>(a, T, logT, m11, ul, v0[3], v1[3]);
    x00 = gunpack(v0[0], v1[0]);
    x01 = gunpack(v0[1], v1[1]);
    x10 = gunpack(v0[2], v1[2]);
    x11 = gunpack(v0[3], v1[3]);
  } else {
    x00 = lds_load<BIN, TILE>(lds, e00);
    x01 = lds_load<BIN, TILE>(lds, e01);
    x10 = lds_load<BIN, TILE>(lds, e10);
    x11 = lds_load<BIN, TILE>(lds, e11);
  }
  const auto w2b = TW::load(tw_at(a, j2b, sh1 - 1));
  Fe<FrP, 24> p, q, u;   // products / reduce_q results: < 19 p / 16
  Fe<FrP, BIN + 32> b1;
  if constexpr (FIRST && Q == 1) {
    // the first two stages of the transform: three of the four twiddles are omega^0 (low = lo = 0; w2b =
    // omega^(n/4)).  A product by one is the operand itself, reduced: a table-row subtraction (reduce_q, ~30
    // instructions) instead of a Montgomery multiplication (~330) — 3 of the 16 products of an 8-stage first pass
    p = reduce_q(x01);
    q = reduce_q(x11);
    u = reduce_q(add(x10, q));
    b1 = sub_nc(x10, q);
  } else {
    const auto w1 = TW::load(tw_at(a, j1, sh1));
    const auto w2a = TW::load(tw_at(a, j1, sh1 - 1));
    p = mul(w1, x01);
    q = mul(w1, x11);
    const Fe<FrP, BIN + 32> b0 = add_nc(x10, q);
    b1 = sub_nc(x10, q);
    u = mul(w2a, b0);
  }
  const Fe<FrP, BIN + 32> a0 = add(x00, p), a1 = sub(x00, p);
  const auto v = mul(w2b, b1);
  if constexpr (DST_G) {
    const Fe<FrP, BIN + 64> y00 = add(a0, u), y10 = sub(a0, u), y01 = add(a1, v), y11 = sub(a1, v);
    gstore<FIRST, TILE, LAST>(a, T, logT, m00, ul, y00);
    gstore<FIRST, TILE, LAST>(a, T, logT, m01, ul, y01);
    gstore<FIRST, TILE, LAST>(a, T, logT, m10, ul, y10);
    gstore<FIRST, TILE, LAST>(a, T, logT, m11, ul, y11);
  } else {
    const Fe<FrP, BIN + 64> y00 = add_nc(a0, u), y10 = sub_nc(a0, u), y01 = add_nc(a1, v), y11 = sub_nc(a1, v);
    lds_store<TILE>(lds, e00, y00);
    lds_store<TILE>(lds, e10, y10);
    lds_store<TILE>(lds, e01, y01);
    lds_store<TILE>(lds, e11, y11);
    block_sync();
  }
}

// stage 1 alone, inputs from global memory, outputs to LDS (passes with an odd number of stages open with it):
// the thread's four elements mid = 4r .. 4r + 3 are two butterflies with the same twiddle
template <bool FIRST, int TILE>
__device__ __forceinline__ void fft_stage1_from_global(u32* lds, const PassArgs& a, int T, int logT) {
  const u32 g = threadIdx.x;
  const u32 ul = g & (u32)(T - 1), r = g >> logT;
  const u32 m0 = r << 2;
  uint4 v0[4], v1[4];
#pragma unroll
  for (int k = 0; k < 4; k++) gload<FIRST, TILE>(a, T, logT, m0 + k, ul, v0[k], v1[k]);
  const u32 lo = tile_lo<FIRST>(a, blockIdx.x, T, ul);
  const auto w = ElemTraits<Fe<FrP, 16>>::load(tw_at(a, lo, a.logn - a.sbits - 1));
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const auto x = gunpack(v0[k], v1[k]), y = gunpack(v0[k + 1], v1[k + 1]);
    Fe<FrP, 24> t;
    if constexpr (FIRST) t = reduce_q(y);   // stage 1 of the transform: every twiddle is omega^0
    else t = mul(w, y);
    lds_store<TILE>(lds, (m0 + k) * T + ul, Fe<FrP, 128>(add(x, t)));
    lds_store<TILE>(lds, (m0 + k + 1) * T + ul, Fe<FrP, 128>(sub(x, t)));
  }
  block_sync();
}

// stages Q .. K of a pass whose first stage(s) already ran from global memory: pairs, the last one to global
template <bool FIRST, int Q, int BIN, int TILE, int LAST>
__device__ __forceinline__ void fft_pairs_to_global(u32* lds, const PassArgs& a, int T, int logT) {
  if constexpr ((2 << Q) <= TILE) {
    if (Q + 1 == a.K) {
      fft_stage2<FIRST, Q, BIN, TILE, false, true, LAST>(lds, a, T, logT);
      return;
    }
    if (Q + 1 < a.K) {
      fft_stage2<FIRST, Q, BIN, TILE, false, false, LAST>(lds, a, T, logT);
      fft_pairs_to_global<FIRST, Q + 2, BIN + 64, TILE, LAST>(lds, a, T, logT);
    }
  }
}

// the K stages of a pass through LDS: pairs while two are left, then a single one (K odd).  Used for short
// passes (K < 3); longer ones go through fft_pairs_to_global
template <bool FIRST, int Q, int BIN, int TILE>
__device__ __forceinline__ void fft_stages(u32* lds, const PassArgs& a, int T, int logT) {
  if (Q > a.K) return;
  if constexpr (Q <= 2) {
    if (Q + 1 <= a.K) {
      fft_stage2<FIRST, Q, BIN, TILE, false, false, 0>(lds, a, T, logT);
      fft_stages<FIRST, Q + 2, BIN + 64, TILE>(lds, a, T, logT);
      return;
    }
    fft_stage1<FIRST, Q, BIN, TILE>(lds, a, T, logT);
  }
}

template <bool FIRST, int BEND, int TILE, int LAST>
__device__ __forceinline__ void fft_store_tile(u32* lds, const PassArgs& a, int T, int logT) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const u32 e = threadIdx.x + (u32)k * (TILE / 4);
    // consecutive lanes -> consecutive ul (contiguous addresses within a T-run) in the later passes; in the
    // first pass consecutive mid (contiguous outputs) — see tile_index
    const u32 ul = FIRST ? (e >> (31 - __clz(TILE / T))) : (e & (u32)(T - 1));
    const u32 mid = FIRST ? (e & (u32)(TILE / T - 1)) : (e >> logT);
    auto v = lds_load<BEND, TILE>(lds, mid * T + ul);
    fe_carry(v);   // (a pair leaves its outputs uncarried in LDS)
    gstore<FIRST, TILE, LAST>(a, T, logT, mid, ul, v);
  }
}

// KODD: the pass has an odd number of stages (it opens with one stage from global memory instead of a pair).  A
// template parameter because the two paths differ by 60 registers: compiled into one kernel, the odd path's 155 set
// the occupancy of both (3 waves per SIMD instead of 5).
template <bool FIRST, int TILE, int LAST, bool KODD>
__global__ void __launch_bounds__(TILE / 4) k_fft_pass(PassArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];  // 9 * TILE words
  const int M = 1 << a.K;
  const int T = TILE / M;
  const int logT = 31 - __clz(T);
  constexpr int B0 = 96;  // >= 85 (any 256-bit input) and >= 64 (inter-pass storage)
  constexpr int LOGT = TILE == 2048 ? 11 : (TILE == 1024 ? 10 : 9);
  if (a.K >= 3) {
    // the first stage(s) read the tile from global memory, the last pair writes it back: LDS is only the
    // exchange between the stage pairs in between (three round trips instead of five for 8 stages)
    if constexpr (KODD) {
      fft_stage1_from_global<FIRST, TILE>(lds, a, T, logT);
      fft_pairs_to_global<FIRST, 2, B0 + 32, TILE, LAST>(lds, a, T, logT);
    } else {
      fft_stage2<FIRST, 1, B0, TILE, true, false, LAST>(lds, a, T, logT);
      fft_pairs_to_global<FIRST, 3, B0 + 64, TILE, LAST>(lds, a, T, logT);
    }
    return;
  }
  if constexpr (KODD) return;   // (short passes are launched with KODD = false)
  // short passes (tiny transforms): tile in, stages through LDS, tile out
  {
    uint4 v0[4], v1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 e = threadIdx.x + (u32)k * (TILE / 4);
      gload<FIRST, TILE>(a, T, logT, e >> logT, e & (u32)(T - 1), v0[k], v1[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 e = threadIdx.x + (u32)k * (TILE / 4);
      lds_store<TILE>(lds, (e >> logT) * T + (e & (u32)(T - 1)), gunpack(v0[k], v1[k]));
    }
  }
  block_sync();
  fft_stages<FIRST, 1, B0, TILE>(lds, a, T, logT);
  fft_store_tile<FIRST, B0 + 32 * LOGT, TILE, LAST>(lds, a, T, logT);
}

// n == 1 or tiny n (< FFT_TILE): one workgroup, direct global-memory version of the same
// algorithm (bit reversal + stages), one butterfly per lane per stage.
__global__ void __launch_bounds__(FFT_THREADS) k_fft_small(const u32* __restrict__ in, u32* __restrict__ out,
                                                           const u32* __restrict__ tw, int n, int logn,
                                                           u32* __restrict__ scratch, int out_stride) {
  using ET = ElemTraits<Fe<FrP, 32>>;
  for (int i = threadIdx.x; i < n; i += FFT_THREADS) {
    const int src = logn ? (int)(__brev((unsigned)i) >> (32 - logn)) : 0;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = in[(size_t)src * 8 + k];
    u32 o[8];
    pack(canonical(unpack<FrP, 85>(w)), o);
#pragma unroll
    for (int k = 0; k < 8; k++) scratch[(size_t)i * 8 + k] = o[k];
  }
  block_sync();
  for (int s = 1; s <= logn; s++) {
    const int m = 1 << (s - 1);
    for (int b = threadIdx.x; b < n / 2; b += FFT_THREADS) {
      const int j = b & (m - 1);
      const int k0 = ((b >> (s - 1)) << s) | j;
      const auto w = ElemTraits<Fe<FrP, 16>>::load(tw + (size_t)((long long)j << (logn - s)) * 8);
      const auto x = ET::load(scratch + (size_t)k0 * 8);
      const auto y = ET::load(scratch + (size_t)(k0 + m) * 8);
      const auto t = mul(w, y);
      u32 o[8];
      pack(Fe<FrP, 32>(reduce_to<32>(add(x, t))), o);
#pragma unroll
      for (int k = 0; k < 8; k++) scratch[(size_t)k0 * 8 + k] = o[k];
      pack(Fe<FrP, 32>(reduce_to<32>(sub(x, t))), o);
#pragma unroll
      for (int k = 0; k < 8; k++) scratch[(size_t)(k0 + m) * 8 + k] = o[k];
    }
    block_sync();
  }
  for (int i = threadIdx.x; i < n; i += FFT_THREADS) {
    u32 o[8];
    pack(canonical(ET::load(scratch + (size_t)i * 8)), o);
#pragma unroll
    for (int k = 0; k < 8; k++) out[(size_t)i * out_stride + k] = o[k];
    for (int k = 8; k < out_stride; k++) out[(size_t)i * out_stride + k] = 0;
  }
}

struct FftLayout {
  u32 *omega, *small, *tw, *buf[2];
  size_t bytes;
  int lo, hi;
};
static FftLayout fft_layout(int n, void* wsp, size_t wsb) {
  FftLayout L;
  Bump b(wsp, wsb);
  const int half = n / 2 > 0 ? n / 2 : 1;
  L.lo = half < TW_LO ? half : TW_LO;
  L.hi = (half + L.lo - 1) / L.lo;
  L.omega = b.take<u32>(8);
  L.small = b.take<u32>((size_t)(L.lo + L.hi) * 8);
  L.tw = b.take<u32>((size_t)(n > 1 ? n : 1) * 8);   // the pyramid: n - 1 entries
  L.buf[0] = b.take<u32>((size_t)n * 8);
  L.buf[1] = b.take<u32>((size_t)n * 8);
  b.take<u32>(64);
  L.bytes = b.off;
  return L;
}

// omega (device, wire form) -> the twiddle pyramid: tw[t] = omega^t, t < n/2, and its subsampled levels behind it
// (n entries of 8 words; scratch: small, (lo + hi) x 8 words)
static void fft_build_twiddles(const u32* d_omega, int n, u32* small, u32* tw, hipStream_t st) {
  const int half = n / 2 > 0 ? n / 2 : 1;
  const int lo = half < TW_LO ? half : TW_LO;
  const int hi = (half + lo - 1) / lo;
  hipLaunchKernelGGL(k_tw_small, dim3((lo + hi + 255) / 256), dim3(256), 0, st, d_omega, lo, hi, small);
  hipLaunchKernelGGL(k_tw_full, dim3((half + 255) / 256), dim3(256), 0, st, small, lo, half, tw);
  if (n >= 4) {
    int logn = 0;
    while ((1 << logn) < n) logn++;
    hipLaunchKernelGGL(k_tw_pyramid, dim3((half + 255) / 256), dim3(256), 0, st, tw, n, logn);
  }
}

// the transform proper: d_in (n x 8 words) -> d_out (n x out_stride words), in place allowed only
// through the two ping-pong buffers (d_out may be one of them only if it is not read by the last pass)
static int fft_core(const u32* d_in, int n, const u32* tw, u32* d_out, int out_stride, u32* buf0, u32* buf1,
                    hipStream_t st, const u32* scale = nullptr) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  const int logn = ilog2((uint32_t)n);
  if (n < FFT_TILE_SMALL) {
    if (scale) return fail(OZK_E_INTERNAL, "output scaling needs the tiled transform");
    hipLaunchKernelGGL(k_fft_small, dim3(1), dim3(FFT_THREADS), 0, st, d_in, d_out, tw, n, logn, buf0, out_stride);
    OZK_HIP(hipGetLastError());
    return OZK_OK;
  }
  // Passes of at most 8 stages.  Every pass AFTER the first gets an EVEN number of stages (its kernel then opens with
  // a stage pair: 93 registers, 4 waves per SIMD with the tile's 36 KiB of LDS; the odd form needs 150 and runs 3), so
  // the first pass — whose kernel is lean either way (79) — takes the parity of log2 n; within that, the larger shares
  // first (a later pass with K stages touches HBM in runs of TILE / 2^K elements).  2^22: 8 + 8 + 6; 2^21: 7 + 8 + 6.
  // OZK_FFT_PLAN=0: the even split of rounds 1-2 (2^22: 8 + 7 + 7).
  constexpr int tile = FFT_TILE_SMALL;
  int maxk = env_int("OZK_FFT_MAXK", 8);
  if (maxk < 3) maxk = 3;
  if (maxk > 10) maxk = 10;
  int npass = (logn + maxk - 1) / maxk;
  int plan[16], planned = 0;
  if (env_int("OZK_FFT_PLAN", 1) && npass > 1) {
    const int evenmax = maxk & ~1;
    for (int np = npass; np <= npass + 1 && !planned; np++)
      for (int k0 = (logn < maxk ? logn : maxk); k0 >= 3 && !planned; k0--) {
        const int rest = logn - k0, parts = np - 1;
        if (rest <= 0 || (rest & 1) || rest < 4 * parts || rest > evenmax * parts) continue;
        plan[0] = k0;
        const int units = rest / 2;   // pairs of stages, spread over the later passes, larger shares first
        for (int i = 0; i < parts; i++) plan[1 + i] = 2 * (units / parts + (i < units % parts ? 1 : 0));
        planned = np;
      }
    if (planned) npass = planned;
  }
  if (const char* ks = getenv("OZK_FFT_KS")) {   // experiment: explicit stage counts, e.g. "8,6,8" (must sum to log2 n)
    int k[16], c = 0, sum = 0;
    for (const char* q = ks; *q && c < 16;) {
      k[c] = atoi(q);
      sum += k[c++];
      while (*q && *q != ',') q++;
      if (*q == ',') q++;
    }
    bool ok = sum == logn && c >= 1;
    for (int i = 0; i < c; i++) ok = ok && k[i] >= 3 && k[i] <= 10;
    if (ok) {
      for (int i = 0; i < c; i++) plan[i] = k[i];
      planned = npass = c;
    }
  }
  int sbits = 0, cur = 0;
  u32* bufs[2] = {buf0, buf1};
  const u32* src = d_in;
  const size_t lds_bytes = (size_t)9 * tile * 4;
  for (int pass = 0; pass < npass; pass++) {
    const int left = logn - sbits, passes_left = npass - pass;
    const int K = planned ? plan[pass] : (left + passes_left - 1) / passes_left;
    const bool last = pass == npass - 1;
    PassArgs a;
    a.in = src;
    a.out = last ? d_out : bufs[cur];
    a.out_stride = out_stride;
    a.tw = tw;
    a.n = n;
    a.logn = logn;
    a.sbits = sbits;
    a.K = K;
    a.pyr = env_int("OZK_FFT_TW_PYRAMID", 1) != 0;
    a.scale = last ? scale : nullptr;
    const int tiles = n / tile;
    const int mode = !last ? 0 : (a.scale ? 2 : 1);
#define OZK_FFT_LAUNCH(F, MODE)                                                                                          \
  do {                                                                                                                   \
    if (K >= 3 && (K & 1))                                                                                               \
      hipLaunchKernelGGL((k_fft_pass<F, FFT_TILE_SMALL, MODE, true>), dim3(tiles), dim3(FFT_TILE_SMALL / 4), lds_bytes,  \
                         st, a);                                                                                         \
    else                                                                                                                 \
      hipLaunchKernelGGL((k_fft_pass<F, FFT_TILE_SMALL, MODE, false>), dim3(tiles), dim3(FFT_TILE_SMALL / 4), lds_bytes, \
                         st, a);                                                                                         \
  } while (0)
    if (pass == 0) {
      if (mode == 0) OZK_FFT_LAUNCH(true, 0);
      else if (mode == 1) OZK_FFT_LAUNCH(true, 1);
      else OZK_FFT_LAUNCH(true, 2);
    } else {
      if (mode == 0) OZK_FFT_LAUNCH(false, 0);
      else if (mode == 1) OZK_FFT_LAUNCH(false, 1);
      else OZK_FFT_LAUNCH(false, 2);
    }
#undef OZK_FFT_LAUNCH
    src = a.out;
    cur ^= 1;
    sbits += K;
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// ---- plan cache -------------------------------------------------------------------------------
// Everything that depends only on the domain — the twiddle table omega^t (n/2 x 32 B: 64 MiB at 2^22), and for
// the witness map also omega^-1's table, the two coset power tables and the four constants — is built on the
// first call with a given (device, n, omega[, g]) and kept in library-owned HBM: a prover transforms over ONE
// domain, seven times per proof (R1CStoQAP.java:163-230), and rebuilding the table was 55 us of a 0.67 ms
// transform at 2^22 and ~0.2 ms of the 2.4 ms witness map at 2^21.  (The reference recomputes two modular
// exponentiations per butterfly, algebra_fft_FFTAuxiliary.cu:127,138.)  Four plans PER DEVICE, least recently
// used out.  A plan is PINNED (refcount) from plan_get until the caller has enqueued its last kernel that reads it
// (PlanPin); only unpinned plans are evicted, and an evicted plan's memory is released with hipFree, which waits
// for the kernels already enqueued.  (Round 2 kept four slots for the whole process and handed out raw pointers
// into them: eight GPUs, or five concurrent domains, evicted each other's tables between plan_get and the launches.)
// When every plan of a device is pinned the cache grows past four and shrinks again on release.
// OZK_FFT_PLAN_CACHE=0 builds the tables per call in the caller's workspace, as round 1 did.
struct QapConsts;
struct FftPlan : PinCacheItem {
  int n = 0;
  bool qap = false;
  uint8_t omega[32] = {0}, g[32] = {0};
  uint8_t* mem = nullptr;
  u32 *tw_f = nullptr, *tw_i = nullptr, *pw_g = nullptr, *pw_gi = nullptr, *small = nullptr;
  u32 *sc_g = nullptr, *sc_gi = nullptr;
  QapConsts* consts = nullptr;
  hipEvent_t ready = nullptr;
  bool same_key(const FftPlan& o) const {
    return n == o.n && qap == o.qap && memcmp(omega, o.omega, 32) == 0 && (!qap || memcmp(g, o.g, 32) == 0);
  }
};
// Round 4: the bookkeeping is pin_cache.h's — a plan is allocated, built and freed with NO lock held (round 3 held
// one process-wide mutex across the hipMalloc, the build enqueue and an evicted plan's device-synchronising hipFree).
constexpr int FFT_PLANS = 4;  // per device
static PinCache<FftPlan> g_plans;
static PinCacheLimits plan_limits() {
  long mb = env_int("OZK_FFT_PLAN_CACHE_MB", 4096);   // (a 2^26 witness-map plan is 4 GiB; larger domains build per call)
  if (mb < 0) mb = 0;
  return PinCacheLimits{FFT_PLANS, (size_t)mb << 20};
}

// (no lock held) releases the plans' device memory; the caller's current device is restored
static void plans_free(std::vector<FftPlan*>& dead) {
  if (dead.empty()) return;
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (FftPlan* p : dead) {
    if (p->mem) {
      (void)hipSetDevice(p->device);
      (void)hipFree(p->mem);
    }
    if (p->ready) (void)hipEventDestroy(p->ready);
    delete p;
  }
  dead.clear();
  if (have_cur) (void)hipSetDevice(cur);
}

// returns the cached plan for (current device, n, omega[, g]), PINNED, with its build enqueued on `st` if it is
// new; the caller releases it (plan_release / PlanPin) after enqueueing the last kernel that reads the tables.
// *out stays null when the plan does not fit the cache's byte budget: the caller builds its tables per call.
static int plan_get(int n, const uint8_t* omega, const uint8_t* g, hipStream_t st, FftPlan** out);
static void plan_release(FftPlan* p) {
  std::vector<FftPlan*> dead;
  g_plans.release(p, plan_limits(), &dead);
  plans_free(dead);
}
struct PlanPin {
  FftPlan* p = nullptr;
  ~PlanPin() {
    if (p) plan_release(p);
  }
};

static int fft_dev(const void* d_in, int n, const uint8_t* omega_host, void* d_out, void* wsp, size_t wsb,
                   hipStream_t st, int out_stride = 16) {
  const FftLayout L = fft_layout(n, wsp, wsb);
  if (L.bytes > wsb) return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
  if (n >= 2 && env_int("OZK_FFT_PLAN_CACHE", 1)) {
    PlanPin pin;
    int rc = plan_get(n, omega_host, nullptr, st, &pin.p);
    if (rc) return rc;
    if (pin.p) return fft_core((const u32*)d_in, n, pin.p->tw_f, (u32*)d_out, out_stride, L.buf[0], L.buf[1], st);
    // (the plan does not fit the cache's byte budget: tables in the caller's workspace, below)
  }
  OZK_HIP(hipMemcpyAsync(L.omega, omega_host, 32, hipMemcpyHostToDevice, st));
  fft_build_twiddles(L.omega, n, L.small, L.tw, st);
  return fft_core((const u32*)d_in, n, L.tw, (u32*)d_out, out_stride, L.buf[0], L.buf[1], st);
}

// ---------------------------------------------------------------------------------------------
// QAP witness map: the caller of the FFT path.
// R1CStoQAP.R1CStoQAPWitness (reductions/r1cs_to_qap/R1CStoQAP.java:163-230) from the evaluations
// of A, B, C on the domain S (m = |S| a power of two) to the m + 1 coefficients of H:
//     a = iFFT_S(A), b = iFFT_S(B), c = iFFT_S(C)                 SerialFFT.radix2InverseFFT  (:86-95)
//     A' = FFT_S(a_i g^i), B', C' likewise                        radix2CosetFFT (:100-105), multiplyByCoset
//                                                                 (FFTAuxiliary.java:224-232)
//     H_T = (A' o B' - C') / Z(g),  Z(g) = g^m - 1                divideByZOnCoset (SerialFFT.java:158-163)
//     h = iFFT_S(H_T) o g^-i ; h_m = 0                            radix2CosetInverseFFT (:111-115), :225
// Seven transforms over two twiddle tables (omega, omega^-1) built once, and four streaming pointwise
// kernels that fold the 1/m of the inverse transforms into the coset powers; data never leaves HBM and
// stays in plain (non-Montgomery) form — constants and power tables are in Montgomery form, so every
// pointwise product is one multiplication.
struct QapConsts {  // device-resident, packed 8 words each
  u32 omega[8], omega_inv[8], g[8], g_inv[8];  // wire form (plain canonical)
  u32 m_inv_mont[8];                           // (1/m) R
  u32 zinv_mont[8];                            // (1 / (g^m - 1)) R
};

// omega^-1 = omega^(m-1), g^-1, 1/m, 1/Z(g): one block (one lane) each, side by side — every one is
// a ~0.25 ms serial exponentiation / Fermat inversion
__global__ void k_qap_consts(QapConsts* __restrict__ c, int m) {
  if (threadIdx.x != 0) return;
  u32 w[8], o[8];
  if (blockIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = c->omega[i];
    const Fe<FrP, 32> om = Fe<FrP, 32>(to_mont<FrP>(w));
    Fe<FrP, 32> oi = fe_one<FrP>();
    const unsigned e1 = (unsigned)(m - 1);
    for (int b = 31; b >= 0; b--) {
      oi = Fe<FrP, 32>(sqr(oi));
      if ((e1 >> b) & 1) oi = Fe<FrP, 32>(mul(oi, om));
    }
    from_mont(oi, o);
#pragma unroll
    for (int i = 0; i < 8; i++) c->omega_inv[i] = o[i];
  } else if (blockIdx.x == 1) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = c->g[i];
    from_mont(inv(to_mont<FrP>(w)), o);
#pragma unroll
    for (int i = 0; i < 8; i++) c->g_inv[i] = o[i];
  } else if (blockIdx.x == 2) {
    u32 mw[8] = {(u32)m, 0, 0, 0, 0, 0, 0, 0};
    pack(canonical(inv(to_mont<FrP>(mw))), o);
#pragma unroll
    for (int i = 0; i < 8; i++) c->m_inv_mont[i] = o[i];
  } else if (blockIdx.x == 3) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = c->g[i];
    const Fe<FrP, 32> g = Fe<FrP, 32>(to_mont<FrP>(w));
    Fe<FrP, 32> gm = fe_one<FrP>();
    const unsigned e2 = (unsigned)m;
    for (int b = 31; b >= 0; b--) {
      gm = Fe<FrP, 32>(sqr(gm));
      if ((e2 >> b) & 1) gm = Fe<FrP, 32>(mul(gm, g));
    }
    pack(canonical(inv(sub(gm, fe_one<FrP>()))), o);  // Z(g) != 0: g generates Fr*, m < r - 1
#pragma unroll
    for (int i = 0; i < 8; i++) c->zinv_mont[i] = o[i];
  }
}

// data[i] <- data[i] * base^i * k   (pw: two-level power table of base as built by k_tw_small with
// lo = TW_LO: pw[j] = base^j, pw[lo + j] = base^(j lo); k in Montgomery form).  base^0 = 1 leaves
// element 0 multiplied by k only, as multiplyByCoset does (FFTAuxiliary.java:227-231).
__global__ void __launch_bounds__(256) k_coset_scale(u32* __restrict__ data, int n, int stride,
                                                     const u32* __restrict__ pw, int lo,
                                                     const u32* __restrict__ k_mont) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  using ET = ElemTraits<Fe<FrP, 16>>;
  const uint4* sp = reinterpret_cast<const uint4*>(data + (size_t)i * stride);
  const uint4 v0 = sp[0], v1 = sp[1];
  const u32 w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  const auto x = unpack<FrP, 85>(w);
  const auto p = mul(ET::load(pw + (size_t)(i % lo) * 8), ET::load(pw + (size_t)(lo + i / lo) * 8));  // base^i R
  // x p / R = x base^i (plain); then times k: mul(., kR) keeps it plain
  u32 o[8];
  pack(canonical(mul(mul(x, p), ET::load(k_mont))), o);
  uint4* dst = reinterpret_cast<uint4*>(data + (size_t)i * stride);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// h[i] <- (a[i] b[i] - c[i]) zinv   (all plain; zinv, R^2 in Montgomery / raw form)
__global__ void __launch_bounds__(256) k_qap_pointwise(const u32* a, const u32* __restrict__ b,
                                                       const u32* __restrict__ c, int n,
                                                       const u32* __restrict__ zinv_mont, u32* h) {  // h may be a
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  using ET = ElemTraits<Fe<FrP, 16>>;
  const auto x = ET::load(a + (size_t)i * 8), y = ET::load(b + (size_t)i * 8), z = ET::load(c + (size_t)i * 8);
  const auto r2 = fe_const<FrP, 16>(FrP::R2);
  const auto xy = mul(mul(x, y), r2);          // x y (plain)
  u32 o[8];
  pack(canonical(mul(sub(xy, z), ET::load(zinv_mont))), o);
  uint4* dst = reinterpret_cast<uint4*>(h + (size_t)i * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// full[i] = base^i * k for i < n, Montgomery form (pw: two-level power table of base, k_mont = k R): the table the
// last pass of a transform multiplies its outputs by (PassArgs::scale)
__global__ void __launch_bounds__(256) k_scale_table(const u32* __restrict__ pw, int lo, int n,
                                                     const u32* __restrict__ k_mont, u32* __restrict__ full) {
  using ET = ElemTraits<Fe<FrP, 16>>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto p = mul(ET::load(pw + (size_t)(i % lo) * 8), ET::load(pw + (size_t)(lo + i / lo) * 8));
  u32 o[8];
  pack(canonical(mul(p, ET::load(k_mont))), o);
  uint4* dst = reinterpret_cast<uint4*>(full + (size_t)i * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

struct QapLayout {
  QapConsts* consts;
  u32 *small, *tw_f, *tw_i, *pw_g, *pw_gi, *sc_g, *sc_gi, *buf[2], *va, *vb, *vc;
  size_t bytes;
  int lo;
};
static QapLayout qap_layout(int m, void* wsp, size_t wsb) {
  QapLayout L;
  Bump b(wsp, wsb);
  const int half = m / 2 > 0 ? m / 2 : 1;
  L.lo = TW_LO;
  const int hi = (m + TW_LO - 1) / TW_LO + 1;
  L.consts = b.take<QapConsts>(1);
  L.small = b.take<u32>((size_t)(TW_LO + hi) * 8);
  L.tw_f = b.take<u32>((size_t)m * 8);   // twiddle pyramids: m - 1 entries each
  L.tw_i = b.take<u32>((size_t)m * 8);
  L.pw_g = b.take<u32>((size_t)(TW_LO + hi) * 8);
  L.pw_gi = b.take<u32>((size_t)(TW_LO + hi) * 8);
  L.sc_g = b.take<u32>((size_t)m * 8);    // g^i / m and g^-i / m for every i (only used without the plan cache)
  L.sc_gi = b.take<u32>((size_t)m * 8);
  L.buf[0] = b.take<u32>((size_t)m * 8);
  L.buf[1] = b.take<u32>((size_t)m * 8);
  L.va = b.take<u32>((size_t)m * 8);
  L.vb = b.take<u32>((size_t)m * 8);
  L.vc = b.take<u32>((size_t)m * 8);
  b.take<u32>(64);
  L.bytes = b.off;
  return L;
}

// the domain-dependent part of the witness map: constants, both twiddle tables, both coset power tables
static int qap_build_tables(QapConsts* consts, u32* small, u32* tw_f, u32* tw_i, u32* pw_g, u32* pw_gi, u32* sc_g,
                            u32* sc_gi, int m, const uint8_t* omega_host, const uint8_t* g_host, hipStream_t st) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  OZK_HIP(hipMemcpyAsync(consts->omega, omega_host, 32, hipMemcpyHostToDevice, st));
  OZK_HIP(hipMemcpyAsync(consts->g, g_host, 32, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_qap_consts, dim3(4), dim3(64), 0, st, consts, m);
  fft_build_twiddles(consts->omega, m, small, tw_f, st);
  fft_build_twiddles(consts->omega_inv, m, small, tw_i, st);
  const int hi = (m + TW_LO - 1) / TW_LO + 1;
  hipLaunchKernelGGL(k_tw_small, dim3((TW_LO + hi + 255) / 256), dim3(256), 0, st, consts->g, TW_LO, hi, pw_g);
  hipLaunchKernelGGL(k_tw_small, dim3((TW_LO + hi + 255) / 256), dim3(256), 0, st, consts->g_inv, TW_LO, hi, pw_gi);
  hipLaunchKernelGGL(k_scale_table, dim3((m + 255) / 256), dim3(256), 0, st, pw_g, TW_LO, m, consts->m_inv_mont, sc_g);
  hipLaunchKernelGGL(k_scale_table, dim3((m + 255) / 256), dim3(256), 0, st, pw_gi, TW_LO, m, consts->m_inv_mont, sc_gi);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

static int plan_get(int n, const uint8_t* omega, const uint8_t* g, hipStream_t st, FftPlan** out) {
  *out = nullptr;
  FftPlan key;
  OZK_HIP(hipGetDevice(&key.device));
  key.n = n;
  key.qap = g != nullptr;
  memcpy(key.omega, omega, 32);
  if (g) memcpy(key.g, g, 32);
  const int half = n / 2 > 0 ? n / 2 : 1;
  const int hi = (n + TW_LO - 1) / TW_LO + 1;
  auto carve = [&](uint8_t* base, FftPlan* dst) {   // same order with and without memory: sizes, then pointers
    Bump b(base, ~(size_t)0);
    QapConsts* c = b.take<QapConsts>(1);
    u32* sm = b.take<u32>((size_t)(TW_LO + hi) * 8);
    u32* twf = b.take<u32>((size_t)(n > 1 ? n : 1) * 8);   // twiddle pyramids: n - 1 entries each
    u32* twi = g ? b.take<u32>((size_t)n * 8) : nullptr;
    u32* pg = g ? b.take<u32>((size_t)(TW_LO + hi) * 8) : nullptr;
    u32* pgi = g ? b.take<u32>((size_t)(TW_LO + hi) * 8) : nullptr;
    u32* sg = g ? b.take<u32>((size_t)n * 8) : nullptr;
    u32* sgi = g ? b.take<u32>((size_t)n * 8) : nullptr;
    b.take<u32>(64);
    if (dst) {
      dst->sc_g = sg;
      dst->sc_gi = sgi;
      dst->consts = c;
      dst->small = sm;
      dst->tw_f = twf;
      dst->tw_i = twi;
      dst->pw_g = pg;
      dst->pw_gi = pgi;
    }
    return b.off;
  };
  const size_t bytes = carve(nullptr, nullptr);
  std::vector<FftPlan*> dead;
  FftPlan* p = nullptr;
  const auto res = g_plans.acquire(
      key, bytes, plan_limits(), false,
      [&]() -> FftPlan* {
        FftPlan* np = new (std::nothrow) FftPlan();
        if (np) {
          np->device = key.device;
          np->n = key.n;
          np->qap = key.qap;
          memcpy(np->omega, key.omega, 32);
          memcpy(np->g, key.g, 32);
        }
        return np;
      },
      &p, &dead);
  plans_free(dead);   // evicted plans: hipFree outside the cache's lock
  if (res == PinCache<FftPlan>::PER_CALL) return OZK_OK;
  if (res == PinCache<FftPlan>::BUILD_FAILED) return fail(OZK_E_NOMEM, "FFT plan could not be built (host memory, or a concurrent build of the same plan failed)");
  if (res == PinCache<FftPlan>::BUILD) {
    int rc = OZK_OK;
    hipError_t e = hipMalloc((void**)&p->mem, bytes);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ready, hipEventDisableTiming);
    if (e != hipSuccess) rc = fail(OZK_E_NOMEM, "FFT plan allocation (%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (!rc) {
      carve(p->mem, p);
      if (g) {
        rc = qap_build_tables(p->consts, p->small, p->tw_f, p->tw_i, p->pw_g, p->pw_gi, p->sc_g, p->sc_gi, n, omega, g, st);
      } else {
        // (from the plan's own copy of omega: host memory that outlives the asynchronous copy)
        hipError_t e2 = hipMemcpyAsync(p->consts->omega, p->omega, 32, hipMemcpyHostToDevice, st);
        if (e2 != hipSuccess) rc = fail(OZK_E_NO_DEVICE, "hipMemcpyAsync failed: %s", hipGetErrorString(e2));
        else fft_build_twiddles(p->consts->omega, n, p->small, p->tw_f, st);
      }
    }
    if (!rc && hipEventRecord(p->ready, st) != hipSuccess) rc = fail(OZK_E_NO_DEVICE, "hipEventRecord failed");
    g_plans.publish(p, rc == OZK_OK, &dead);
    plans_free(dead);
    if (rc) return rc;
  }
  const hipError_t we = hipStreamWaitEvent(st, p->ready, 0);   // a no-op on the stream that built it
  if (we != hipSuccess) {
    plan_release(p);
    return fail(OZK_E_NO_DEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(we));
  }
  *out = p;
  return OZK_OK;
}

static int qap_witness_dev(const void* d_A, const void* d_B, const void* d_C, int m, const uint8_t* omega_host,
                           const uint8_t* g_host, void* d_H, void* wsp, size_t wsb, hipStream_t st) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  QapLayout L = qap_layout(m, wsp, wsb);
  if (L.bytes > wsb) return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
  PlanPin pin;  // held until the last launch below is enqueued
  if (env_int("OZK_FFT_PLAN_CACHE", 1)) {
    int prc = plan_get(m, omega_host, g_host, st, &pin.p);
    if (prc) return prc;
  }
  if (pin.p) {   // (else: cache off, or the plan does not fit its byte budget — tables in the workspace)
    FftPlan* pl = pin.p;
    L.consts = pl->consts;
    L.tw_f = pl->tw_f;
    L.tw_i = pl->tw_i;
    L.pw_g = pl->pw_g;
    L.pw_gi = pl->pw_gi;
    L.sc_g = pl->sc_g;
    L.sc_gi = pl->sc_gi;
  } else {
    int brc = qap_build_tables(L.consts, L.small, L.tw_f, L.tw_i, L.pw_g, L.pw_gi, L.sc_g, L.sc_gi, m, omega_host, g_host, st);
    if (brc) return brc;
  }
  const int TB = 256, nb = (m + TB - 1) / TB;
  const u32* in[3] = {(const u32*)d_A, (const u32*)d_B, (const u32*)d_C};
  u32* v[3] = {L.va, L.vb, L.vc};
  int rc;
  // The scalings ride on the inverse transforms before them: their last pass multiplies every output by g^i / m
  // (resp. g^-i / m) from a full table of the plan instead of four extra passes over the data (44 k_coset_scale
  // launches = 0.22 of the 2.07 ms of round 2's map at 2^21).  Transforms too small for the tiled kernel keep the
  // separate kernel.
  const bool fold = m >= FFT_TILE_SMALL && env_int("OZK_QAP_FOLD_SCALE", 1) != 0;
  for (int k = 0; k < 3; k++) {
    // coefficients (times m), then a_i g^i / m, then the evaluations on the coset
    if ((rc = fft_core(in[k], m, L.tw_i, v[k], 8, L.buf[0], L.buf[1], st, fold ? L.sc_g : nullptr))) return rc;
    if (!fold)
      hipLaunchKernelGGL(k_coset_scale, dim3(nb), dim3(TB), 0, st, v[k], m, 8, L.pw_g, TW_LO, L.consts->m_inv_mont);
    if ((rc = fft_core(v[k], m, L.tw_f, v[k], 8, L.buf[0], L.buf[1], st))) return rc;
  }
  hipLaunchKernelGGL(k_qap_pointwise, dim3(nb), dim3(TB), 0, st, L.va, L.vb, L.vc, m, L.consts->zinv_mont, L.va);
  if ((rc = fft_core(L.va, m, L.tw_i, (u32*)d_H, 8, L.buf[0], L.buf[1], st, fold ? L.sc_gi : nullptr))) return rc;
  if (!fold)
    hipLaunchKernelGGL(k_coset_scale, dim3(nb), dim3(TB), 0, st, (u32*)d_H, m, 8, L.pw_gi, TW_LO, L.consts->m_inv_mont);
  OZK_HIP(hipMemsetAsync((u32*)d_H + (size_t)m * 8, 0, 32, st));  // coefficientsH.add(zero), R1CStoQAP.java:225
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// ---------------------------------------------------------------------------------------------
// Constraint evaluation: the step of R1CStoQAP.R1CStoQAPWitness BEFORE its transforms
// (reductions/r1cs_to_qap/R1CStoQAP.java:143-160,195-199): out[i] = <row i of a sparse matrix, assignment>, with
// LinearCombination.evaluate's rule that a term with index 0 contributes `one` whatever its coefficient
// (relations/objects/LinearCombination.java:39-50).  The Java loops over List<Fp> on the CPU ("this evaluation
// part is costy", R1CStoQAP.java:157); here the R1CS sits in HBM as three CSR matrices and the witness arrives
// as one 32-byte-per-element buffer, so the whole witness side of a proof stays on the device.
// One lane per row; rows longer than R1CS_LONG terms (the closing constraint of the reference's synthetic
// circuits sums every variable, R1CSConstruction.java:87-96) are left to one workgroup each.
constexpr int R1CS_LONG = 64;
using FrAcc = Fe<FrP, 32>;

// value of one term (plain, < 2r): z[j], times its coefficient when there is a coefficient array
// ONE0: LinearCombination.evaluate's rule for the variable "one" (constraint evaluation); without it the kernels
// are a plain sparse matrix x vector product over Fr (the QAP instance of the setup, ozk_qap_instance_dev)
template <bool ONE0>
__device__ __forceinline__ Fe<FrP, 32> r1cs_term(const u32* __restrict__ idx, const u32* __restrict__ coeff,
                                                 const u32* __restrict__ z, u32 t) {
  using ET = ElemTraits<Fe<FrP, 17>>;
  const u32 j = idx[t];
  if (ONE0 && j == 0) {
    Fe<FrP, 32> one = Fe<FrP, 32>(fe_zero<FrP>());
    one.l[0] = 1;
    return one;
  }
  const uint4* zp = reinterpret_cast<const uint4*>(z + (size_t)j * 8);
  const uint4 a = zp[0], b = zp[1];
  const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const auto x = unpack<FrP, 85>(w);
  if (coeff == nullptr) return Fe<FrP, 32>(reduce_to<32>(x));
  const auto cm = ET::from_wire(coeff + (size_t)t * 8);   // coefficient * R
  return Fe<FrP, 32>(mul(x, cm));                         // plain product
}

template <bool ONE0>
__global__ void __launch_bounds__(256) k_r1cs_eval(const u32* __restrict__ ptr, const u32* __restrict__ idx,
                                                   const u32* __restrict__ coeff, const u32* __restrict__ z, int rows,
                                                   u32* __restrict__ out) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const u32 b = ptr[row], e = ptr[row + 1];
  if (e - b > (u32)R1CS_LONG) return;
  FrAcc acc = FrAcc(fe_zero<FrP>());
  for (u32 t = b; t < e; t++) acc = FrAcc(reduce_to<32>(add(acc, r1cs_term<ONE0>(idx, coeff, z, t))));
  u32 o[8];
  pack(canonical(acc), o);
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)row * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// a long row is cut into R1CS_SPLIT slices, one workgroup each: strided partial sums per thread, a tree over the
// 256 partials in LDS, one 32-byte partial per slice; k_r1cs_eval_long2 then adds the slices of every long row
constexpr int R1CS_SPLIT = 64;
__device__ __forceinline__ FrAcc r1cs_block_sum(FrAcc acc, u32* part) {
#pragma unroll
  for (int i = 0; i < 9; i++) part[i * 256 + threadIdx.x] = acc.l[i];
  block_sync();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      FrAcc x, y;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        x.l[i] = part[i * 256 + threadIdx.x];
        y.l[i] = part[i * 256 + threadIdx.x + o];
      }
      const FrAcc s = FrAcc(reduce_to<32>(add(x, y)));
#pragma unroll
      for (int i = 0; i < 9; i++) part[i * 256 + threadIdx.x] = s.l[i];
    }
    block_sync();
  }
  FrAcc r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = part[i * 256];
  return r;
}
template <bool ONE0>
__global__ void __launch_bounds__(256) k_r1cs_eval_long1(const u32* __restrict__ ptr, const u32* __restrict__ idx,
                                                         const u32* __restrict__ coeff, const u32* __restrict__ z,
                                                         const u32* __restrict__ long_rows, u32* __restrict__ partial) {
  __shared__ u32 part[9 * 256];
  const u32 lr = blockIdx.x / R1CS_SPLIT, sl = blockIdx.x % R1CS_SPLIT;
  const u32 row = long_rows[lr];
  const u32 b = ptr[row], e = ptr[row + 1];
  const u32 per = (e - b + R1CS_SPLIT - 1) / R1CS_SPLIT;
  const u32 lo = b + sl * per;
  const u32 hi = (lo + per < e) ? lo + per : e;
  FrAcc acc = FrAcc(fe_zero<FrP>());
  for (u32 t = lo + threadIdx.x; t < hi; t += 256) acc = FrAcc(reduce_to<32>(add(acc, r1cs_term<ONE0>(idx, coeff, z, t))));
  const FrAcc r = r1cs_block_sum(acc, part);
  if (threadIdx.x == 0) {
    u32 o[8];
    pack(canonical(r), o);
#pragma unroll
    for (int i = 0; i < 8; i++) partial[(size_t)blockIdx.x * 8 + i] = o[i];
  }
}
__global__ void __launch_bounds__(256) k_r1cs_eval_long2(const u32* __restrict__ long_rows, const u32* __restrict__ partial,
                                                         u32* __restrict__ out) {
  __shared__ u32 part[9 * 256];
  const u32 row = long_rows[blockIdx.x];
  FrAcc acc = FrAcc(fe_zero<FrP>());
  if (threadIdx.x < R1CS_SPLIT) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = partial[((size_t)blockIdx.x * R1CS_SPLIT + threadIdx.x) * 8 + i];
    acc = FrAcc(reduce_to<32>(unpack<FrP, 16>(w)));
  }
  const FrAcc r = r1cs_block_sum(acc, part);
  if (threadIdx.x == 0) {
    u32 o[8];
    pack(canonical(r), o);
#pragma unroll
    for (int i = 0; i < 8; i++) out[(size_t)row * 8 + i] = o[i];
  }
}

// ---------------------------------------------------------------------------------------------
// The QAP instance of the setup (R1CStoQAP.R1CStoQAPRelation, reductions/r1cs_to_qap/
// R1CStoQAP.java:37-98): Lagrange coefficients of the domain at the secret point t
// (FFTAuxiliary.serialRadix2LagrangeCoefficients, FFTAuxiliary.java:250-302), the powers t^i, and
// At / Bt / Ct as sparse products over the TRANSPOSED constraint matrices (k_r1cs_eval<false> above).
// Round 2 did all of it in Python integers: 4.3 s of a 5.7 s setup at 2^20 constraints.
//
// L_i(t) = (Z / m) omega^i / (t - omega^i), Z = t^m - 1 (t not in the domain: the caller checks t^m != 1).
// A lane takes LAG_BATCH indices t, t + lanes, ... (interleaved: consecutive lanes write consecutive records) and
// shares ONE inversion among their denominators (Montgomery's trick: the Java inverts m times, :295).
// pw: two-level power table of omega (k_tw_small with lo = TW_LO, Montgomery form).
constexpr int LAG_BATCH = 8;
struct LagConsts {  // packed 8 words each, filled by k_lag_consts
  u32 t_wire[8];    // t (plain canonical), uploaded by the host
  u32 t_mont[8];    // t R
  u32 c_mont[8];    // (t^m - 1) / m * R
  u32 z_wire[8];    // t^m - 1 (plain): Zt of the QAP instance
};
__global__ void k_lag_consts(LagConsts* __restrict__ c, int m) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  u32 w[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = c->t_wire[i];
  const Fe<FrP, 32> t = Fe<FrP, 32>(to_mont<FrP>(w));
  Fe<FrP, 32> tm = fe_one<FrP>();
  const unsigned e = (unsigned)m;
  for (int b = 31; b >= 0; b--) {
    tm = Fe<FrP, 32>(sqr(tm));
    if ((e >> b) & 1) tm = Fe<FrP, 32>(mul(tm, t));
  }
  const auto Z = sub(tm, fe_one<FrP>());
  u32 mw[8] = {(u32)m, 0, 0, 0, 0, 0, 0, 0};
  const auto mi = inv(to_mont<FrP>(mw));
  pack(canonical(t), o);
#pragma unroll
  for (int i = 0; i < 8; i++) c->t_mont[i] = o[i];
  pack(canonical(mul(Z, mi)), o);
#pragma unroll
  for (int i = 0; i < 8; i++) c->c_mont[i] = o[i];
  from_mont(Z, o);
#pragma unroll
  for (int i = 0; i < 8; i++) c->z_wire[i] = o[i];
}
__global__ void __launch_bounds__(256) k_lagrange(const LagConsts* __restrict__ c, const u32* __restrict__ pw, int lo,
                                                  int m, u32* __restrict__ out) {
  using ET = ElemTraits<Fe<FrP, 16>>;
  using E32 = Fe<FrP, 32>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lanes = (m + LAG_BATCH - 1) / LAG_BATCH;
  if (t >= lanes) return;
  const auto tm = ET::load(c->t_mont);
  const auto cm = ET::load(c->c_mont);
  auto at = [&](int k) { return k * lanes + t; };
  E32 prefix[LAG_BATCH], wpow[LAG_BATCH];
  E32 run = E32(fe_one<FrP>());
#pragma unroll
  for (int k = 0; k < LAG_BATCH; k++) {
    const int i = at(k);
    if (i < m) {
      wpow[k] = E32(mul(ET::load(pw + (size_t)(i % lo) * 8), ET::load(pw + (size_t)(lo + i / lo) * 8)));  // omega^i R
      const auto d = reduce_to<32>(sub(tm, wpow[k]));                                                      // (t - omega^i) R
      run = E32(mul(run, d));
    }
    prefix[k] = run;
  }
  E32 invrun = E32(inv(run));
#pragma unroll
  for (int k = LAG_BATCH - 1; k >= 0; k--) {
    const int i = at(k);
    if (i < m) {
      E32 di = invrun;                                  // 1 / (t - omega^i), Montgomery form
      if (k > 0) di = E32(mul(invrun, prefix[k - 1]));
      invrun = E32(mul(invrun, reduce_to<32>(sub(tm, wpow[k]))));
      u32 o[8];
      from_mont(mul(mul(cm, wpow[k]), di), o);          // plain, canonical
      uint4* dst = reinterpret_cast<uint4*>(out + (size_t)i * 8);
      dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
      dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
  }
}
// out[i] = base^i * k for i < n, plain canonical (pw: two-level Montgomery power table of `base`; k_mont = k R).
// The Ht = t^i of the QAP instance (R1CStoQAP.java:90-96) with k = 1, and the H query's scalars t^i Z / delta
// (SerialSetup.java:146-151) with k = Z / delta, in one pass.
__global__ void __launch_bounds__(256) k_powers_scaled(const u32* __restrict__ pw, int lo, int n,
                                                       const u32* __restrict__ k_mont, u32* __restrict__ out) {
  using ET = ElemTraits<Fe<FrP, 16>>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto p = mul(ET::load(pw + (size_t)(i % lo) * 8), ET::load(pw + (size_t)(lo + i / lo) * 8));  // base^i R
  Fe<FrP, 1> one = fe_zero<FrP>();
  one.l[0] = 1;
  u32 o[8];
  pack(canonical(mul(mul(p, ET::load(k_mont)), one)), o);   // (base^i R)(k R)/R = base^i k R; /R -> plain
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}
// out[i] = (ka a[i] + kb b[i] + c[i]) kk, all plain canonical; ka, kb, kk: three Montgomery-form constants at k3
// (the deltaABC / gammaABC scalars of the setup, SerialSetup.java:61-74)
__global__ void __launch_bounds__(256) k_lincomb3(const u32* __restrict__ a, const u32* __restrict__ b,
                                                  const u32* __restrict__ c, int n, const u32* __restrict__ k3,
                                                  u32* __restrict__ out) {
  using ET = ElemTraits<Fe<FrP, 16>>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto x = ET::load(a + (size_t)i * 8), y = ET::load(b + (size_t)i * 8), z = ET::load(c + (size_t)i * 8);
  const auto s = add(add(mul(x, ET::load(k3)), mul(y, ET::load(k3 + 8))), z);   // plain
  u32 o[8];
  pack(canonical(mul(reduce_to<32>(s), ET::load(k3 + 16))), o);
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}
// wire (plain canonical) constants -> Montgomery form, in place: one lane per 8-word element
__global__ void k_to_mont_inplace(u32* __restrict__ v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8], o[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = v[(size_t)i * 8 + k];
  pack(canonical(to_mont<FrP>(w)), o);
#pragma unroll
  for (int k = 0; k < 8; k++) v[(size_t)i * 8 + k] = o[k];
}

}  // namespace ozk

using namespace ozk;

namespace ozk {
void fft_plan_cache_release() {
  std::vector<FftPlan*> dead;
  g_plans.drain(&dead);   // pinned plans (a call in flight on another thread) stay
  plans_free(dead);
}
}  // namespace ozk

extern "C" {

size_t ozk_fft_workspace_bytes(int32_t n) {
  if (n <= 0 || (n & (n - 1))) return 0;
  return fft_layout(n, nullptr, 0).bytes;
}

int ozk_fft_dev(const void* d_in, int32_t n, const uint8_t* omega_host32, void* d_out, void* d_workspace,
                size_t workspace_bytes, void* stream) {
  if (!d_in || !d_out || !omega_host32 || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || (n & (n - 1)) || n > (1 << 28))
    return fail(OZK_E_INVALID, "FFT size %d is not a power of two in [1, 2^28]", n);
  return fft_dev(d_in, n, omega_host32, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
}

int ozk_fft_compact_dev(const void* d_in, int32_t n, const uint8_t* omega_host32, void* d_out, void* d_workspace,
                        size_t workspace_bytes, void* stream) {
  if (!d_in || !d_out || !omega_host32 || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || (n & (n - 1)) || n > (1 << 28))
    return fail(OZK_E_INVALID, "FFT size %d is not a power of two in [1, 2^28]", n);
  return fft_dev(d_in, n, omega_host32, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, 8);
}

size_t ozk_qap_witness_workspace_bytes(int32_t m) {
  if (m <= 1 || (m & (m - 1))) return 0;
  return qap_layout(m, nullptr, 0).bytes;
}

int ozk_qap_witness_dev(const void* d_A, const void* d_B, const void* d_C, int32_t m, const uint8_t* omega_host32,
                        const uint8_t* g_host32, void* d_H, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
  if (!d_A || !d_B || !d_C || !omega_host32 || !g_host32 || !d_H || !d_workspace)
    return fail(OZK_E_INVALID, "null pointer argument");
  if (m <= 1 || (m & (m - 1)) || m > (1 << 28))
    return fail(OZK_E_INVALID, "domain size %d is not a power of two in [2, 2^28]", m);
  return qap_witness_dev(d_A, d_B, d_C, m, omega_host32, g_host32, d_H, d_workspace, workspace_bytes,
                         (hipStream_t)stream);
}

size_t ozk_r1cs_evaluate_workspace_bytes(int32_t n_long) {
  return n_long > 0 ? (size_t)n_long * R1CS_SPLIT * 32 + 256 : 256;
}

int ozk_r1cs_evaluate_dev(const void* d_row_ptr, const void* d_index, const void* d_coeff, const void* d_assignment,
                          int32_t rows, const void* d_long_rows, int32_t n_long, void* d_out, void* d_workspace,
                          size_t workspace_bytes, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!d_row_ptr || !d_index || !d_assignment || !d_out || (n_long > 0 && (!d_long_rows || !d_workspace)))
    return fail(OZK_E_INVALID, "null pointer argument");
  if (rows <= 0 || n_long < 0) return fail(OZK_E_INVALID, "bad row count");
  if (n_long > 0 && workspace_bytes < ozk_r1cs_evaluate_workspace_bytes(n_long))
    return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", ozk_r1cs_evaluate_workspace_bytes(n_long),
                workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_r1cs_eval<true>, dim3((rows + 255) / 256), dim3(256), 0, st, (const u32*)d_row_ptr, (const u32*)d_index,
                     (const u32*)d_coeff, (const u32*)d_assignment, rows, (u32*)d_out);
  if (n_long > 0) {
    hipLaunchKernelGGL(k_r1cs_eval_long1<true>, dim3(n_long * R1CS_SPLIT), dim3(256), 0, st, (const u32*)d_row_ptr,
                       (const u32*)d_index, (const u32*)d_coeff, (const u32*)d_assignment, (const u32*)d_long_rows,
                       (u32*)d_workspace);
    hipLaunchKernelGGL(k_r1cs_eval_long2, dim3(n_long), dim3(256), 0, st, (const u32*)d_long_rows,
                       (const u32*)d_workspace, (u32*)d_out);
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// out = M x v over Fr for a CSR matrix resident in HBM (the same kernels as ozk_r1cs_evaluate_dev without the rule
// for variable 0): the QAP instance uses it on the transposed constraint matrices with v = the Lagrange coefficients
int ozk_sparse_mat_vec_dev(const void* d_row_ptr, const void* d_index, const void* d_coeff, const void* d_vec,
                           int32_t rows, const void* d_long_rows, int32_t n_long, void* d_out, void* d_workspace,
                           size_t workspace_bytes, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!d_row_ptr || !d_index || !d_vec || !d_out || (n_long > 0 && (!d_long_rows || !d_workspace)))
    return fail(OZK_E_INVALID, "null pointer argument");
  if (rows <= 0 || n_long < 0) return fail(OZK_E_INVALID, "bad row count");
  if (n_long > 0 && workspace_bytes < ozk_r1cs_evaluate_workspace_bytes(n_long))
    return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", ozk_r1cs_evaluate_workspace_bytes(n_long),
                workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_r1cs_eval<false>, dim3((rows + 255) / 256), dim3(256), 0, st, (const u32*)d_row_ptr,
                     (const u32*)d_index, (const u32*)d_coeff, (const u32*)d_vec, rows, (u32*)d_out);
  if (n_long > 0) {
    hipLaunchKernelGGL(k_r1cs_eval_long1<false>, dim3(n_long * R1CS_SPLIT), dim3(256), 0, st, (const u32*)d_row_ptr,
                       (const u32*)d_index, (const u32*)d_coeff, (const u32*)d_vec, (const u32*)d_long_rows,
                       (u32*)d_workspace);
    hipLaunchKernelGGL(k_r1cs_eval_long2, dim3(n_long), dim3(256), 0, st, (const u32*)d_long_rows,
                       (const u32*)d_workspace, (u32*)d_out);
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// Lagrange coefficients of the radix-2 domain of size m at t (FFTAuxiliary.java:250-302) and Z(t) = t^m - 1.
// d_out: m x 32 B plain LE; d_zt: 32 B; workspace: ozk_qap_lagrange_workspace_bytes(m).  t must not be a domain
// element: the entry point computes t^m on the host and returns OZK_E_INVALID when it is 1 (the caller then takes the
// reference's indicator branch itself: it never happens for a random t).  omega: the domain's root of unity.
size_t ozk_qap_lagrange_workspace_bytes(int32_t m) {
  if (m <= 1 || (m & (m - 1))) return 0;
  const int hi = (m + TW_LO - 1) / TW_LO + 1;
  return pad256(sizeof(LagConsts)) + pad256((size_t)(TW_LO + hi) * 32) + 512;
}
int ozk_qap_lagrange_dev(const uint8_t* t_host32, const uint8_t* omega_host32, int32_t m, void* d_out, void* d_zt,
                         void* d_workspace, size_t workspace_bytes, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!t_host32 || !omega_host32 || !d_out || !d_zt || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  if (m <= 1 || (m & (m - 1)) || m > (1 << 28)) return fail(OZK_E_INVALID, "domain size %d is not a power of two in [2, 2^28]", m);
  if (workspace_bytes < ozk_qap_lagrange_workspace_bytes(m)) return fail(OZK_E_INVALID, "workspace too small");
  {
    // t in the domain (t^m == 1) makes one denominator t - omega^i zero, and the shared inversion of its lane would
    // silently zero all LAG_BATCH coefficients of that lane: refused here, on the host (log2 m squarings), so that a
    // C-ABI / JNI caller that forgot the test gets an error instead of a wrong CRS.  The reference takes its
    // indicator branch in that case (FFTAuxiliary.java:262-276); a caller does the same on its side.
    u32 tw[8];
    memcpy(tw, t_host32, 32);
    Fe<FrP, 32> tm = Fe<FrP, 32>(to_mont<FrP>(tw));
    for (int k = m; k > 1; k >>= 1) tm = Fe<FrP, 32>(sqr(tm));
    u32 o[8], one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    from_mont(tm, o);
    if (memcmp(o, one, 32) == 0)
      return fail(OZK_E_INVALID, "t is an element of the size-%d domain (t^m = 1): the Lagrange coefficients are an indicator vector, "
                                 "take that branch on the caller's side", m);
  }
  hipStream_t st = (hipStream_t)stream;
  uint8_t* w = (uint8_t*)d_workspace;
  LagConsts* c = (LagConsts*)w;
  u32* pw = (u32*)(w + pad256(sizeof(LagConsts)));
  u32* d_omega = (u32*)(w + pad256(sizeof(LagConsts)) + pad256((size_t)(TW_LO + (m + TW_LO - 1) / TW_LO + 1) * 32));
  OZK_HIP(hipMemcpyAsync(c->t_wire, t_host32, 32, hipMemcpyHostToDevice, st));
  OZK_HIP(hipMemcpyAsync(d_omega, omega_host32, 32, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_lag_consts, dim3(1), dim3(64), 0, st, c, m);
  const int hi = (m + TW_LO - 1) / TW_LO + 1;
  hipLaunchKernelGGL(k_tw_small, dim3((TW_LO + hi + 255) / 256), dim3(256), 0, st, d_omega, TW_LO, hi, pw);
  const int lanes = (m + LAG_BATCH - 1) / LAG_BATCH;
  hipLaunchKernelGGL(k_lagrange, dim3((lanes + 255) / 256), dim3(256), 0, st, c, pw, TW_LO, m, (u32*)d_out);
  OZK_HIP(hipMemcpyAsync(d_zt, c->z_wire, 32, hipMemcpyDeviceToDevice, st));
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// d_out[i] = base^i * k mod r for i < n (32-byte plain LE); base, k: 32-byte LE host values.
size_t ozk_fr_powers_workspace_bytes(int32_t n) {
  if (n <= 0) return 0;
  const int hi = (n + TW_LO - 1) / TW_LO + 1;
  return pad256((size_t)(TW_LO + hi) * 32) + 512;
}
int ozk_fr_powers_dev(const uint8_t* base_host32, const uint8_t* k_host32, int32_t n, void* d_out, void* d_workspace,
                      size_t workspace_bytes, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!base_host32 || !k_host32 || !d_out || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 28) + 1) return fail(OZK_E_INVALID, "count %d out of range", n);
  if (workspace_bytes < ozk_fr_powers_workspace_bytes(n)) return fail(OZK_E_INVALID, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int hi = (n + TW_LO - 1) / TW_LO + 1;
  u32* pw = (u32*)d_workspace;
  u32* cst = (u32*)((uint8_t*)d_workspace + pad256((size_t)(TW_LO + hi) * 32));  // [0..8) base, [8..16) k
  OZK_HIP(hipMemcpyAsync(cst, base_host32, 32, hipMemcpyHostToDevice, st));
  OZK_HIP(hipMemcpyAsync(cst + 8, k_host32, 32, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_to_mont_inplace, dim3(1), dim3(64), 0, st, cst + 8, 1);
  hipLaunchKernelGGL(k_tw_small, dim3((TW_LO + hi + 255) / 256), dim3(256), 0, st, cst, TW_LO, hi, pw);
  hipLaunchKernelGGL(k_powers_scaled, dim3((n + 255) / 256), dim3(256), 0, st, pw, TW_LO, n, cst + 8, (u32*)d_out);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// d_out[i] = (ka a_i + kb b_i + c_i) * kk mod r; a, b, c, out: n x 32 B plain LE in HBM (out may alias an input);
// ka, kb, kk: 32-byte LE host values; d_scratch: 96 bytes of device memory for the three constants.
int ozk_fr_lincomb3_dev(const void* d_a, const void* d_b, const void* d_c, int32_t n, const uint8_t* ka_host32,
                        const uint8_t* kb_host32, const uint8_t* kk_host32, void* d_out, void* d_scratch96, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!d_a || !d_b || !d_c || !ka_host32 || !kb_host32 || !kk_host32 || !d_out || !d_scratch96)
    return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0) return fail(OZK_E_INVALID, "count %d out of range", n);
  hipStream_t st = (hipStream_t)stream;
  u32* k3 = (u32*)d_scratch96;
  OZK_HIP(hipMemcpyAsync(k3, ka_host32, 32, hipMemcpyHostToDevice, st));
  OZK_HIP(hipMemcpyAsync(k3 + 8, kb_host32, 32, hipMemcpyHostToDevice, st));
  OZK_HIP(hipMemcpyAsync(k3 + 16, kk_host32, 32, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_to_mont_inplace, dim3(1), dim3(64), 0, st, k3, 3);
  hipLaunchKernelGGL(k_lincomb3, dim3((n + 255) / 256), dim3(256), 0, st, (const u32*)d_a, (const u32*)d_b,
                     (const u32*)d_c, n, k3, (u32*)d_out);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

int ozk_qap_witness_host(const uint8_t* A, const uint8_t* B, const uint8_t* C, int32_t m, const uint8_t* omega,
                         const uint8_t* g, int32_t task_id, uint8_t* H) {
  if (!A || !B || !C || !omega || !g || !H) return fail(OZK_E_INVALID, "null pointer argument");
  if (m <= 1 || (m & (m - 1)) || m > (1 << 28))
    return fail(OZK_E_INVALID, "domain size %d is not a power of two in [2, 2^28]", m);
  const size_t vb = (size_t)m * 32, hb = ((size_t)m + 1) * 32;
  const size_t vpad = pad256(vb), hpad = pad256(hb);
  const size_t wsb = ozk_qap_witness_workspace_bytes(m);
  CtxGuard guard;
  int rc = ctx_acquire(task_id, &guard.c);
  if (rc) return rc;
  HostCtx* c = guard.c;
  if ((rc = ctx_reserve(c, 3 * vpad + hpad + wsb + 1024))) return rc;
  uint8_t* d = c->arena;
  uint8_t *dA = d, *dB = d + vpad, *dC = d + 2 * vpad, *dH = d + 3 * vpad, *dW = d + 3 * vpad + hpad;
  hipStream_t st = c->st[0];
  if ((rc = staged_h2d(c, dA, A, vb, st))) return rc;
  if ((rc = staged_h2d(c, dB, B, vb, st))) return rc;
  if ((rc = staged_h2d(c, dC, C, vb, st))) return rc;
  if ((rc = qap_witness_dev(dA, dB, dC, m, omega, g, dH, dW, wsb, st))) return rc;
  return staged_d2h(c, H, dH, hb, st);
}

static int fft_host(const uint8_t* in, int32_t n, const uint8_t* omega, int32_t task_id, uint8_t* out, int out_stride) {
  if (!in || !omega || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || (n & (n - 1)) || n > (1 << 28))
    return fail(OZK_E_INVALID, "FFT size %d is not a power of two in [1, 2^28]", n);
  const size_t in_bytes = (size_t)n * 32, out_bytes = (size_t)n * 4 * out_stride;
  const size_t wsb = ozk_fft_workspace_bytes(n);
  CtxGuard g;
  int rc = ctx_acquire(task_id, &g.c);
  if (rc) return rc;
  HostCtx* c = g.c;
  if ((rc = ctx_reserve(c, pad256(in_bytes) + pad256(out_bytes) + wsb + 1024))) return rc;
  uint8_t* d = c->arena;
  uint8_t* d_out = d + pad256(in_bytes);
  uint8_t* d_ws = d_out + pad256(out_bytes);
  hipStream_t st = c->st[0];
  if ((rc = staged_h2d(c, d, in, in_bytes, st))) return rc;
  if ((rc = fft_dev(d, n, omega, d_out, d_ws, wsb, st, out_stride))) return rc;
  return staged_d2h(c, out, d_out, out_bytes, st);
}

int ozk_fft_host(const uint8_t* in, int32_t n, const uint8_t* omega, int32_t task_id, uint8_t* out) {
  return fft_host(in, n, omega, task_id, out, 16);
}
int ozk_fft_compact_host(const uint8_t* in, int32_t n, const uint8_t* omega, int32_t task_id, uint8_t* out) {
  return fft_host(in, n, omega, task_id, out, 8);
}

}  // extern "C"
