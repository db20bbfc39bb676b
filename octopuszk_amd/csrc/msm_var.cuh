// Variable-base MSM (Pippenger bucket method) for gfx950 — kernels.
//
// Replaces the per-window CUDA pipeline of the reference
// (algebra_msm_VariableBaseMSM.cu:736-1241 kernels, :1246-1428 driver; SURVEY.md §2a,
// §8a rows V5-V9) and computes the same group element as the serial Java
// VariableBaseMSM.pippengerMSM (VariableBaseMSM.java:134-188): sum_i s_i * P_i.
// The window size, bucket order and addition order differ from the reference (any
// order gives the same group element); the result is affine-normalised so its bytes
// are unique.
//
// Pipeline (one launch each, all windows at once, no host round trips):
//   digits    scalars -> GLV split into two signed half scalars (glv.cuh), signed c-bit digit codes per
//             window with the half scalar's sign folded in (plain unsigned 256-bit windows above 2^23)
//   convert   bases wire (Jacobian, canonical) -> affine Montgomery 64 B records, (x, y) and (beta x, y)
//   sort      two-level counting sort of the (window, bucket) keys: hi part across blocks with
//             per-block LDS counts + one global scan, lo part inside one block per coarse bin
//             (register tile + LDS staging, coalesced write-out); LDS atomics only; bins of skewed
//             inputs are split over many blocks (k_sortbig_*)
//   segreduce level 1: every lane takes L consecutive sorted entries and accumulates runs of equal
//             bucket id with XYZZ mixed additions (8M + 2S), negating y where the digit is negative;
//             complete runs go to the bucket array, runs cut by a chunk boundary become "partials"
//             (2 slots per lane)
//   runmerge  sums the <= 16 pieces of every cut bucket (general XYZZ additions)
//   levels    the same segmented scheme over surviving slots with Jacobian adds, until one lane
//             remains — load-balanced for ANY digit distribution (a bucket holding half the input
//             is summed by N/2L lanes, not by one)
//   wsum      per-window  sum_b (b + 1) * B_b: a fused first level (serial running sums per lane, then
//             an in-register wave combine), wave-cooperative upper levels
//   finalize  the last few elements per window, Horner over the windows, affine normalisation
//             (safegcd inversion), wire-out bytes
// Every workgroup barrier is block_sync() (curve.cuh): hipcc dropped the wait for a pending LDS atomic
// before one s_barrier, which silently lost counts on skewed inputs.
#pragma once
#include "curve.cuh"
#include "glv.cuh"

namespace ozk {

constexpr u32 BID_NONE = 0xffffffffu;

struct MsmPlan {
  int n_in;     // (scalar, base) pairs handed in
  int glv;      // 1: every pair is split into two half-length pairs by the GLV endomorphism (glv.cuh)
  int n;        // points the pipeline sorts and accumulates: n_in, or 2 * n_in with GLV
  int sd;       // 1: signed digits in (-2^(c-1), 2^(c-1)] (GLV mode only): half the buckets per window
  int c;        // window bits (<= 16)
  int cb;       // bucket-index bits per window: c, or c - 1 with signed digits
  int W;        // windows = ceil(256 / c), or ceil(128 / c) with GLV
  int L1;       // sorted entries per lane at level 1
  int LK;       // slots per lane at levels >= 2
  int S;        // buckets per lane per wsum level (power of two)
};

OZK_HD u32 scalar_digit(const u32 (&s)[8], int w, int c) {
  const int bit = w * c;
  const int wi = bit >> 5, sh = bit & 31;
  u32 v = s[wi] >> sh;
  if (sh + c > 32 && wi + 1 < 8) v |= s[wi + 1] << (32 - sh);
  return v & ((1u << c) - 1u);
}

// Digit codes (u16, 0 = zero digit, skipped).  Unsigned windows: code = digit d, bucket index b = d,
// weight b.  Signed windows: d in (-2^(c-1), 2^(c-1)], code = (((|d| - 1) << 1) | (d < 0)) + 1, bucket
// index b = |d| - 1 < 2^(c-1), weight b + 1; the point is subtracted when d < 0.
OZK_HD bool digit_decode(u32 code, int sd, u32& b, u32& neg) {
  if (sd) {
    const u32 t = code - 1u;
    b = t >> 1;
    neg = t & 1u;
  } else {
    b = code;
    neg = 0;
  }
  return code != 0;
}
// Signed recoding of one half scalar |k| < 2^126.97 (glv.cuh) whose sign `neg` is folded into every
// digit: sum_w sign_w (b_w + 1) 2^(c w) = (neg ? -|k| : |k|).  Carry-propagated; the top digit stays
// below 2^(c-1), so no carry leaves the top window.  The u16 code has room for 2^16 - 1 non-zero
// values, one short of +-[1, 2^15] at c = 16: digits of a positive half scalar are recoded to
// (-2^15, 2^15], those of a negative one to [-2^15, 2^15), so that after the fold -2^15 never occurs.
template <class Emit>
OZK_HD void signed_digit_codes(const u32 (&e)[8], int c, int W, bool neg, Emit emit) {
  const u32 half = 1u << (c - 1);
  const u32 thr = (c == 16 && neg) ? half - 1u : half;
  u32 cy = 0;
  for (int w = 0; w < W; w++) {
    const u32 d = scalar_digit(e, w, c) + cy;
    cy = d > thr;
    const u32 m = cy ? (1u << c) - d : d;
    emit(w, (uint16_t)(m ? (((m - 1u) << 1) | (cy ^ (u32)neg)) + 1u : 0u));
  }
}

// Packed coarse word: index << 8 | lo8, lo8 = low bucket bits (8 unsigned / 7 signed) | neg << 7 (signed).
// The sorted ENTRIES are 8-byte pairs (base index, bucket id), the sign of the digit in bit 31 of the index (round 4:
// one array of pairs instead of an index array and a bucket-id array — every lane of the level-1 kernel walks its own
// chunk, so each of its streams costs one scattered VMEM instruction per entry: one 8-byte load instead of two 4-byte
// ones measured 3-4 % on the loop, profiles/r04_ubench_l1loop.txt variant I).
constexpr u32 SIDX_NEG = 0x80000000u;

#if defined(__HIPCC__)

// ------------------------------------------------------------------ convert
// One lane per base.  Z == 1 (the prover's normal case: keys produced by this library
// are affine) costs 2 Montgomery conversions; Z == 0 -> infinity marker; any other Z is
// normalised with a per-lane Fermat inversion (correct for reference-produced
// Jacobian keys, slower).
// beta of the GLV endomorphism for this curve's base field (G2: beta^2, acting on both Fq2 components)
template <class CV>
OZK_HD Fe<FqParams, 16> glv_beta() {
  if constexpr (CurveIO<CV>::CW == 16) return fe_const<FqParams, 16>(GlvConsts::BETA_G2);
  else return fe_const<FqParams, 16>(GlvConsts::BETA_G1);
}

// With GLV: record i = (x, y), record n + i = (beta x, y).  Signed-digit plans fold the signs of the
// two half scalars into the digit signs (k_digits_glv), so the records depend on the bases alone —
// which is what makes them reusable across MSMs (ozk_var_msm_prepare_dev).  Unsigned GLV plans
// (neg_flags != nullptr; k_digits_glv has run) write (x, +-y), (beta x, +-y) instead.
template <class CV>
__global__ void __launch_bounds__(256) k_convert_bases(const u32* __restrict__ wire,
                                                       u32* __restrict__ aff, int n, int glv,
                                                       const uint8_t* __restrict__ neg_flags) {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
  using ET = ElemTraits<EA>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32* p = wire + (size_t)i * IO::WIRE_JAC_WORDS;
  u32* o = aff + (size_t)i * IO::AFF_WORDS;
  Aff<EA> q;
  if (ET::wire_is_zero(p + 2 * IO::CW)) {
    q.x = EA(el_zero(q.x));
    q.y = EA(el_zero(q.x));
  } else if (ET::wire_is_one(p + 2 * IO::CW)) {
    q.x = ET::from_wire(p);
    q.y = ET::from_wire(p + IO::CW);
  } else {
    const EA X = ET::from_wire(p), Y = ET::from_wire(p + IO::CW), Z = ET::from_wire(p + 2 * IO::CW);
    const auto zi = inv(Z);
    if (is_zero(Z)) {  // Z == p etc.: still infinity
      q.x = EA(el_zero(q.x));
      q.y = EA(el_zero(q.x));
    } else {
      const auto zi2 = sqr(zi);
      q.x = EA(canonical(mul(X, zi2)));
      q.y = EA(canonical(mul(Y, mul(zi2, zi))));
    }
  }
  // canonical (< p) so that equal points have equal records
  q.x = EA(canonical(q.x));
  q.y = EA(canonical(q.y));
  if (!glv) {
    IO::store_aff(q, o);
    return;
  }
  Aff<EA> q2;
  q2.x = EA(canonical(scale(q.x, glv_beta<CV>())));
  q2.y = q.y;
  if (neg_flags != nullptr) {
    const EA ny = EA(canonical(neg(q.y)));   // infinity marker (0, 0) stays (0, 0)
    if (neg_flags[i]) q.y = ny;
    if (neg_flags[n + i]) q2.y = ny;
  }
  IO::store_aff(q, o);
  IO::store_aff(q2, aff + (size_t)(n + i) * IO::AFF_WORDS);
}

// ------------------------------------------------------------------ digits
// digits[w*n + i] (u16): the c-bit digit of scalar i in window w.  One pass over the scalars.
static __global__ void __launch_bounds__(256) k_digits(const u32* __restrict__ scalars, int n, int c, int W,
                                                uint16_t* __restrict__ digits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[8];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = sp[0], b = sp[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
  s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  for (int w = 0; w < W; w++) digits[(size_t)w * n + i] = (uint16_t)scalar_digit(s, w, c);
}

// GLV form: scalar i (reduced mod r) -> |k1|, |k2| < 2^127 (glv.cuh); virtual scalar i = |k1| and
// n + i = |k2| of a 2n-point MSM with W = ceil(128 / c) windows.  Signed-digit plans XOR the sign of
// the half scalar into every digit's sign; unsigned plans leave the signs in neg_flags for
// k_convert_bases.
static __global__ void __launch_bounds__(256) k_digits_glv(const u32* __restrict__ scalars, int n, int c, int W, int sd,
                                                    uint16_t* __restrict__ digits, uint8_t* __restrict__ neg_flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[8];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = sp[0], b = sp[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
  s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  u32 k1[4], k2[4];
  bool n1, n2;
  glv_decompose(s, k1, n1, k2, n2);
  u32 e1[8] = {k1[0], k1[1], k1[2], k1[3], 0, 0, 0, 0};
  u32 e2[8] = {k2[0], k2[1], k2[2], k2[3], 0, 0, 0, 0};
  const size_t ne = 2 * (size_t)n;
  if (sd) {
    signed_digit_codes(e1, c, W, n1, [&](int w, uint16_t code) { digits[(size_t)w * ne + i] = code; });
    signed_digit_codes(e2, c, W, n2, [&](int w, uint16_t code) { digits[(size_t)w * ne + n + i] = code; });
    return;
  } else {
    for (int w = 0; w < W; w++) {
      digits[(size_t)w * ne + i] = (uint16_t)scalar_digit(e1, w, c);
      digits[(size_t)w * ne + n + i] = (uint16_t)scalar_digit(e2, w, c);
    }
  }
  neg_flags[i] = n1;
  neg_flags[n + i] = n2;
}

// ------------------------------------------------------------------ two-level counting sort
// Sorting the (window, digit) keys with global returning atomics + a fully random scatter cost
// 1.2 ms at 2^20 (profiles/r01_kernel_stats_v2_serial.csv).  Instead: split the digit into
// hi (c - 8 bits) and lo (8 bits).
//   sort1 count / scatter : a block takes SORT_CHUNK consecutive scalars of one window, counts
//       / ranks them per hi value with LDS atomics; entries go to "coarse" bins (window, hi) as
//       packed (index << 8 | lo) words — runs of ~16 consecutive words per bin and block.
//   sort2 : one block per coarse bin: LDS histogram of lo, block scan -> bucket offsets and
//       counts (written for EVERY bucket, so no memset), LDS-ranked scatter inside the bin's
//       own few-KB window of the sorted arrays.
// All atomics are LDS atomics; a wave whose live lanes share one key issues one.
constexpr int SORT_CHUNK = 4096;
constexpr int SORT_BLOCK = 256;

// LDS atomic add of 1 with wave aggregation when every live lane has the same key
template <bool AGGREGATE = true>
__device__ __forceinline__ u32 lds_rank(u32* cnt, u32 key, bool live) {
  if constexpr (!AGGREGATE) return live ? atomicAdd(&cnt[key], 1u) : 0u;
  const unsigned long long m = __ballot(live);
  u32 r = 0;
  if (m != 0ull) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    const u32 kl = __shfl(key, leader);
    if (__all(!live || key == kl)) {
      u32 base = 0;
      if (lane == leader) base = atomicAdd(&cnt[kl], (u32)__popcll(m));
      base = __shfl(base, leader);
      r = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
    } else if (live) {
      r = atomicAdd(&cnt[key], 1u);
    }
  }
  return r;
}

// C1[(w*NH + h)*nblk + blk] = number of non-zero digits with hi value h in this block's chunk
static __global__ void __launch_bounds__(SORT_BLOCK) k_sort1_count(const uint16_t* __restrict__ digits, int n, int lo_bits,
                                                            int sd, int NH, int nblk, u32* __restrict__ C1) {
  extern __shared__ u32 cnt[];
  const int w = blockIdx.y, blk = blockIdx.x;
  for (int h = threadIdx.x; h < NH; h += SORT_BLOCK) cnt[h] = 0;
  block_sync();
  const int i0 = blk * SORT_CHUNK;
  constexpr int PER = SORT_CHUNK / SORT_BLOCK;
  u32 dreg[PER];  // all loads first (independent, pipelined), then the LDS atomics
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = i0 + k * SORT_BLOCK + threadIdx.x;
    dreg[k] = (i < n) ? digits[(size_t)w * n + i] : 0u;
  }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    u32 b, neg;
    const bool live = digit_decode(dreg[k], sd, b, neg);
    lds_rank(cnt, b >> lo_bits, live);
  }
  block_sync();
  for (int h = threadIdx.x; h < NH; h += SORT_BLOCK) C1[((size_t)w * NH + h) * nblk + blk] = cnt[h];
}

// coarse[P1[(w*NH+h)*nblk + blk] + rank] = (i << 8) | lo8
// The chunk is first sorted by hi inside LDS and then written out in bin order: consecutive lanes
// write consecutive words of a bin (runs of ~16 words per bin and block = one 64-byte request
// instead of 16 four-byte ones; the scattered form was bound by the L2 request rate).
static __global__ void __launch_bounds__(SORT_BLOCK) k_sort1_scatter(const uint16_t* __restrict__ digits, int n, int lo_bits,
                                                              int sd, int NH, int nblk, const u32* __restrict__ P1,
                                                              const u32* __restrict__ total, size_t nC1,
                                                              u32* __restrict__ coarse) {
  extern __shared__ u32 sm[];
  u32* gdelta = sm;            // [NH] global start of (bin, block) minus its local start
  u32* lcur = sm + NH;         // [NH] local cursor
  u32* stage = sm + 2 * NH;    // [SORT_CHUNK] packed entries in bin order
  u32* dest = stage + SORT_CHUNK;  // [SORT_CHUNK] their positions in coarse[]
  __shared__ u32 wsum[SORT_BLOCK / 64];
  __shared__ u32 n_local;
  const int w = blockIdx.y, blk = blockIdx.x;
  const int t = threadIdx.x;
  // NH <= 256 = SORT_BLOCK (c <= 16): one bin per thread
  u32 g0 = 0, ct = 0;
  if (t < NH) {
    const size_t idx = ((size_t)w * NH + t) * nblk + blk;
    g0 = P1[idx];
    ct = ((idx + 1 < nC1) ? P1[idx + 1] : *total) - g0;
  }
  u32 incl = ct;
  const int lane = t & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 x = __shfl_up(incl, o);
    if (lane >= o) incl += x;
  }
  if (lane == 63) wsum[t >> 6] = incl;
  block_sync();
  u32 woff = 0;
  for (int k = 0; k < (t >> 6); k++) woff += wsum[k];
  const u32 excl = woff + incl - ct;
  if (t < NH) {
    lcur[t] = excl;
    gdelta[t] = g0 - excl;
  }
  if (t == SORT_BLOCK - 1) n_local = woff + incl;
  block_sync();
  const int i0 = blk * SORT_CHUNK;
  const u32 lo_mask = (1u << lo_bits) - 1u;
  constexpr int PER = SORT_CHUNK / SORT_BLOCK;
  u32 dreg[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = i0 + k * SORT_BLOCK + threadIdx.x;
    dreg[k] = (i < n) ? digits[(size_t)w * n + i] : 0u;
  }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = i0 + k * SORT_BLOCK + threadIdx.x;
    u32 b, neg;
    const bool live = digit_decode(dreg[k], sd, b, neg);
    const u32 hi = b >> lo_bits;
    const u32 r = lds_rank(lcur, hi, live);
    if (live) {
      stage[r] = ((u32)i << 8) | (neg << 7) | (b & lo_mask);
      dest[r] = r + gdelta[hi];
    }
  }
  block_sync();
  const u32 m = n_local;
  for (u32 j = t; j < m; j += SORT_BLOCK) coarse[dest[j]] = stage[j];
}

// Coarse bins above max(SORT_BIG, n/64) entries (skewed digits: the reference's own profiler inputs put half
// of a window into one bucket) are not left to one block: k_sortbig_list enumerates them and cuts
// them into SORTBIG_CHUNK-entry work items, k_sortbig_count / _scan / _scatter run the same
// count - scan - ranked-scatter scheme over those items with a fixed grid.  With no big bin the
// three kernels return at once.
#ifndef OZK_SORTBIG_AGG
#define OZK_SORTBIG_AGG true
#endif
constexpr u32 SORT_BIG = 1u << 16;
constexpr u32 SORTBIG_CHUNK = 1u << 13;
constexpr int SORTBIG_MAXBINS = 1024;   // >= number of bins that can exceed SORT_BIG: cap / SORT_BIG
struct BigBins {                        // device-resident work list
  u32 n_big, n_items;
  u32 bin[SORTBIG_MAXBINS], b0[SORTBIG_MAXBINS], size[SORTBIG_MAXBINS], first_item[SORTBIG_MAXBINS];
};

static __global__ void __launch_bounds__(256) k_sortbig_list(const u32* __restrict__ P1, const u32* __restrict__ total,
                                                      int nblk, int nbins, u32 big_thresh,
                                                      BigBins* __restrict__ bb) {
  __shared__ u32 n_big, n_items;
  if (threadIdx.x == 0) n_big = n_items = 0;
  block_sync();
  for (int bin = threadIdx.x; bin < nbins; bin += blockDim.x) {
    const u32 b0 = P1[(size_t)bin * nblk];
    const u32 b1 = (bin + 1 < nbins) ? P1[(size_t)(bin + 1) * nblk] : *total;
    if (b1 - b0 > big_thresh) {
      const u32 k = atomicAdd(&n_big, 1u);
      const u32 items = (b1 - b0 + SORTBIG_CHUNK - 1) / SORTBIG_CHUNK;
      const u32 first = atomicAdd(&n_items, items);
      if (k < (u32)SORTBIG_MAXBINS) {
        bb->bin[k] = (u32)bin;
        bb->b0[k] = b0;
        bb->size[k] = b1 - b0;
        bb->first_item[k] = first;
      }
    }
  }
  block_sync();
  if (threadIdx.x == 0) {
    // big_thresh >= n/64 bounds the number of big bins by 64 * W <= SORTBIG_MAXBINS
    bb->n_big = n_big < (u32)SORTBIG_MAXBINS ? n_big : (u32)SORTBIG_MAXBINS;
    bb->n_items = n_items;
  }
}

// item -> (index r into the big-bin list, chunk k inside that bin)
__device__ __forceinline__ bool sortbig_item(const BigBins* bb, u32 item, u32& r, u32& k) {
  const u32 nb = bb->n_big;
  for (u32 q = 0; q < nb; q++) {
    const u32 f = bb->first_item[q];
    const u32 items = (bb->size[q] + SORTBIG_CHUNK - 1) / SORTBIG_CHUNK;
    if (item >= f && item < f + items) {
      r = q;
      k = item - f;
      return true;
    }
  }
  return false;
}

// T[item][lo] = count of lo in the item's chunk
static __global__ void __launch_bounds__(SORT_BLOCK) k_sortbig_count(const u32* __restrict__ coarse,
                                                              const BigBins* __restrict__ bb, u32 lo_mask,
                                                              u32* __restrict__ T) {
  __shared__ u32 cnt[256];
  const u32 n_items = bb->n_items;
  for (u32 item = blockIdx.x; item < n_items; item += gridDim.x) {
    u32 r, k;
    if (!sortbig_item(bb, item, r, k)) continue;
    cnt[threadIdx.x] = 0;
    block_sync();
    const u32 lo0 = bb->b0[r] + k * SORTBIG_CHUNK;
    const u32 end = bb->b0[r] + bb->size[r];
    const u32 hi0 = (lo0 + SORTBIG_CHUNK < end) ? lo0 + SORTBIG_CHUNK : end;
    for (u32 e = lo0 + threadIdx.x; e - threadIdx.x < hi0; e += SORT_BLOCK) {
      const bool live = e < hi0;
      const u32 v = live ? coarse[e] : 0u;
      lds_rank<OZK_SORTBIG_AGG>(cnt, v & lo_mask, live);
    }
    block_sync();
    T[(size_t)item * 256 + threadIdx.x] = cnt[threadIdx.x];
    block_sync();
  }
}

// one block per big bin, lane = lo: bucket totals, bucket bases, per-item bases (in place in T)
static __global__ void __launch_bounds__(256) k_sortbig_scan(const BigBins* __restrict__ bb, u32* __restrict__ T, int c,
                                                      int lo_bits, int NH, u32* __restrict__ hist) {
  __shared__ u32 wsum[4];
  const u32 r = blockIdx.x;
  if (r >= bb->n_big || r >= (u32)SORTBIG_MAXBINS) return;
  const u32 items = (bb->size[r] + SORTBIG_CHUNK - 1) / SORTBIG_CHUNK;
  const u32 f = bb->first_item[r];
  const int t = threadIdx.x;
  u32 tot = 0;
  for (u32 k = 0; k < items; k++) tot += T[(size_t)(f + k) * 256 + t];
  // exclusive scan of tot over lo
  u32 incl = tot;
  const int lane = t & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 x = __shfl_up(incl, o);
    if (lane >= o) incl += x;
  }
  if (lane == 63) wsum[t >> 6] = incl;
  block_sync();
  u32 woff = 0;
  for (int q = 0; q < (t >> 6); q++) woff += wsum[q];
  u32 run = bb->b0[r] + woff + incl - tot;
  const int bin = (int)bb->bin[r];
  const int w = bin / NH, h = bin - w * NH;
  if (t < (1 << lo_bits)) hist[(((u32)w << c) | ((u32)h << lo_bits)) + t] = tot;
  for (u32 k = 0; k < items; k++) {
    const u32 cnt = T[(size_t)(f + k) * 256 + t];
    T[(size_t)(f + k) * 256 + t] = run;
    run += cnt;
  }
}

static __global__ void __launch_bounds__(SORT_BLOCK) k_sortbig_scatter(const u32* __restrict__ coarse,
                                                                const BigBins* __restrict__ bb,
                                                                const u32* __restrict__ T, int c, int lo_bits,
                                                                u32 sign_bit, int NH, uint2* __restrict__ sent) {
  __shared__ u32 cur[256];
  const u32 lo_mask = (1u << lo_bits) - 1u;
  const u32 n_items = bb->n_items;
  for (u32 item = blockIdx.x; item < n_items; item += gridDim.x) {
    u32 r, k;
    if (!sortbig_item(bb, item, r, k)) continue;
    cur[threadIdx.x] = T[(size_t)item * 256 + threadIdx.x];
    block_sync();
    const int bin = (int)bb->bin[r];
    const int w = bin / NH, h = bin - w * NH;
    const u32 bucket0 = ((u32)w << c) | ((u32)h << lo_bits);
    const u32 lo0 = bb->b0[r] + k * SORTBIG_CHUNK;
    const u32 end = bb->b0[r] + bb->size[r];
    const u32 hi0 = (lo0 + SORTBIG_CHUNK < end) ? lo0 + SORTBIG_CHUNK : end;
    for (u32 e = lo0 + threadIdx.x; e - threadIdx.x < hi0; e += SORT_BLOCK) {
      const bool live = e < hi0;
      const u32 v = live ? coarse[e] : 0u;
      const u32 lo = v & lo_mask;
      const u32 pos = lds_rank<OZK_SORTBIG_AGG>(cur, lo, live);
      if (live) {
        sent[pos] = make_uint2((v >> 8) | ((v & sign_bit) << 24), bucket0 + lo);
      }
    }
    block_sync();
  }
}

// one block per coarse bin (w, h): finishes the sort inside the bin.
// Shaped to run BESIDE the bucket accumulation of the previous MSM (three-stage schedule): that kernel
// leaves 128 VGPRs per SIMD and ~38 KiB of LDS per CU free, so this one is 512 threads (two waves per SIMD) of at most
// 64 VGPRs and 38 KiB: the register tile is 17 entries per thread (round 2: 256 threads x 36 = 181 VGPRs, 42 KiB —
// it could not be resident next to three accumulation blocks and the overlapped sort took 1.5 ms instead of 0.33).
// The bin goes through the register tile + LDS staging array tile by tile (ONE tile for every bin of a uniform input
// up to 2^20 pairs): per tile a local count, a local scan, a local rank into the staging array, then every bucket's
// piece of the tile is appended to its run in the output — consecutive lanes write consecutive words.  Bins of
// several tiles (n >= 2^21: 16384 and more entries per bin) need the bin-wide bucket offsets first: one extra
// streaming count over the bin.  (The plain two-sweep form with scattered 4-byte writes took 0.59 ms at 2^21 and
// 2.9 ms at 2^23.)  Loads are unpredicated (index clamped to the bin's last entry): 17 predicated loads cost 34
// SGPRs of masks and a branch each.
constexpr int SORT2_BLOCK = 512;
constexpr int S2_PER = 17;
constexpr u32 S2_TILE = S2_PER * SORT2_BLOCK;  // 8704: a coarse bin of 2^21 points (GLV at 2^20) holds 8192 +- 90
static __global__ void __launch_bounds__(SORT2_BLOCK, 8) k_sort2(const u32* __restrict__ coarse, const u32* __restrict__ P1,
                                                      const u32* __restrict__ total, int c, int lo_bits, int NH,
                                                      u32 sign_bit, int nblk, int nbins, u32 big_thresh,
                                                      u32* __restrict__ hist, uint2* __restrict__ sent) {
  __shared__ u32 cnt[256];   // per-tile bucket counts (first: the bin-wide counts of a multi-tile bin)
  __shared__ u32 cur[256];   // next output position of every bucket
  __shared__ u32 lcur[256], gdelta[256];
  __shared__ u32 wsum[4];
  __shared__ u32 stage[S2_TILE];
  const int bin = blockIdx.x;
  const int NLO = 1 << lo_bits;
  const u32 lo_mask = (1u << lo_bits) - 1u;
  const u32 b0 = P1[(size_t)bin * nblk];
  const u32 b1 = (bin + 1 < nbins) ? P1[(size_t)(bin + 1) * nblk] : *total;
  const int w = bin / NH, h = bin - w * NH;
  const u32 bucket0 = ((u32)w << c) | ((u32)h << lo_bits);
  if (b1 - b0 > big_thresh) return;  // split over many blocks by the k_sortbig_* kernels
  const int t = threadIdx.x;
  const int lane = t & 63;
  if (b1 == b0) {  // empty bin: every bucket's count is still written (no memset of hist)
    if (t < NLO) hist[bucket0 + t] = 0;
    return;
  }
  const u32 last = b1 - 1u;
  const bool small = (b1 - b0) <= S2_TILE;
  // exclusive scan of the 256 counters in cnt[] by the first four waves; returns this thread's count and offset
  auto scan_cnt = [&](u32& ct, u32& excl) {
    u32 incl = 0;
    ct = 0;
    if (t < 256) {
      ct = cnt[t];
      incl = ct;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const u32 x = __shfl_up(incl, o);
        if (lane >= o) incl += x;
      }
      if (lane == 63) wsum[t >> 6] = incl;
    }
    block_sync();
    u32 woff = 0;
    if (t < 256)
      for (int k = 0; k < (t >> 6); k++) woff += wsum[k];
    excl = woff + incl - ct;
  };
  if (t < 256) cnt[t] = 0;
  block_sync();
  if (!small) {
    for (u32 e = b0 + t; e - t < b1; e += SORT2_BLOCK) {  // uniform trip count
      const bool live = e < b1;
      const u32 v = coarse[live ? e : last];
      lds_rank(cnt, v & lo_mask, live);
    }
    block_sync();
    u32 ct, excl;
    scan_cnt(ct, excl);
    if (t < NLO) {
      hist[bucket0 + t] = ct;
      cur[t] = b0 + excl;
    }
    block_sync();
    if (t < 256) cnt[t] = 0;
    block_sync();
  }
  u32 vreg[S2_PER];
  for (u32 t0 = b0; t0 < b1; t0 += S2_TILE) {  // uniform trip count
    const u32 t1 = (t0 + S2_TILE < b1) ? t0 + S2_TILE : b1;
#pragma unroll
    for (int k = 0; k < S2_PER; k++) {
      const u32 e = t0 + k * SORT2_BLOCK + t;
      vreg[k] = coarse[e < last ? e : last];
    }
#pragma unroll
    for (int k = 0; k < S2_PER; k++) {
      const u32 e = t0 + k * SORT2_BLOCK + t;
      lds_rank(cnt, vreg[k] & lo_mask, e < t1);
    }
    block_sync();
    u32 tc, lstart;  // this bucket's count in the tile, start of its piece inside the staged tile
    scan_cnt(tc, lstart);
    if (t < 256) {
      u32 cur_t;
      if (small) {
        cur_t = b0 + lstart;
        if (t < NLO) hist[bucket0 + t] = tc;
      } else {
        cur_t = cur[t];
        cur[t] = cur_t + tc;
      }
      lcur[t] = lstart;
      gdelta[t] = cur_t - lstart;  // output position = staged position + gdelta
      cnt[t] = 0;                  // (for the next tile; everybody has read its count)
    }
    block_sync();
#pragma unroll
    for (int k = 0; k < S2_PER; k++) {
      const u32 e = t0 + k * SORT2_BLOCK + t;
      const bool live = e < t1;
      const u32 r = lds_rank(lcur, vreg[k] & lo_mask, live);
      if (live) stage[r] = vreg[k];
    }
    block_sync();
    for (u32 j = t; j < t1 - t0; j += SORT2_BLOCK) {
      const u32 v = stage[j];
      const u32 lo = v & lo_mask;
      const u32 pos = j + gdelta[lo];
      sent[pos] = make_uint2((v >> 8) | ((v & sign_bit) << 24), bucket0 + lo);
    }
    block_sync();
  }
}

// ------------------------------------------------------------------ the whole sort of a SMALL MSM in one launch
// Up to 8192 sorted points (GLV: 4096 pairs) and 1024 buckets per window: one workgroup per window counts, ranks and
// scatters its entries out of registers (LDS atomics), and reserves its piece of the sorted arrays with one global
// atomic — the windows' pieces may land in any order: the accumulation only needs every bucket's entries adjacent.
// Replaces k_sort1_count, the three scan kernels, k_sort1_scatter, k_sort2 and the four k_sortbig_* launches, which
// for such an input are ten dependent dispatches of 1-6 us each with ~10 us between them: 2^10 pairs spent 130 of
// their 940 us there (profiles/r04_small_n_probe.txt).  *total (zeroed by the caller) ends as the number of entries.
constexpr int SORTS_BLOCK = 1024;
constexpr int SORTS_PER = 8;
constexpr int SORTS_MAX_N = SORTS_BLOCK * SORTS_PER;
constexpr int SORTS_MAX_CB = 10;
static __global__ void __launch_bounds__(SORTS_BLOCK) k_sort_small(const uint16_t* __restrict__ digits, int n, int cb, int sd,
                                                                   u32* __restrict__ total, u32* __restrict__ hist,
                                                                   uint2* __restrict__ sent) {
  __shared__ u32 cnt[1 << SORTS_MAX_CB];
  __shared__ u32 off[1 << SORTS_MAX_CB];
  __shared__ u32 wsum[SORTS_BLOCK / 64];
  __shared__ u32 base_s;
  const int w = blockIdx.x, t = threadIdx.x;
  const int NB = 1 << cb;
  if (t < NB) cnt[t] = 0;
  block_sync();
  u32 bk[SORTS_PER], rk[SORTS_PER];
#pragma unroll
  for (int k = 0; k < SORTS_PER; k++) {
    const int i = k * SORTS_BLOCK + t;
    const u32 code = (i < n) ? digits[(size_t)w * n + i] : 0u;
    u32 b, neg;
    const bool live = digit_decode(code, sd, b, neg);
    rk[k] = live ? atomicAdd(&cnt[b], 1u) : 0u;
    bk[k] = live ? (b | (neg << 31)) : BID_NONE;   // (b < 2^10: bit 31 is free for the sign; all ones = no entry)
  }
  block_sync();
  const u32 c = (t < NB) ? cnt[t] : 0u;
  u32 incl = c;
  const int lane = t & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 x = __shfl_up(incl, o);
    if (lane >= o) incl += x;
  }
  if (lane == 63) wsum[t >> 6] = incl;
  block_sync();
  u32 woff = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < SORTS_BLOCK / 64; q++) {
    if (q < (t >> 6)) woff += wsum[q];
    tot += wsum[q];
  }
  if (t < NB) {
    off[t] = woff + incl - c;
    hist[((u32)w << cb) + (u32)t] = c;   // every bucket's count is written (no memset of hist)
  }
  if (t == 0) base_s = atomicAdd(total, tot);
  block_sync();
  const u32 base = base_s;
#pragma unroll
  for (int k = 0; k < SORTS_PER; k++) {
    if (bk[k] == BID_NONE) continue;
    const u32 b = bk[k] & 0x7fffffffu;
    const u32 pos = base + off[b] + rk[k];
    sent[pos] = make_uint2((u32)(k * SORTS_BLOCK + t) | (bk[k] & SIDX_NEG), ((u32)w << cb) | b);
  }
}

// ------------------------------------------------------------------ exclusive scan (3 kernels)
constexpr int SCAN_ITEMS = 16;   // per lane
constexpr int SCAN_BLOCK = 256;  // lanes -> 4096 items per block
static __global__ void __launch_bounds__(SCAN_BLOCK) k_scan_blocksum(const u32* __restrict__ in, int n,
                                                              u32* __restrict__ blocksum) {
  __shared__ u32 sh[SCAN_BLOCK / 64];
  const size_t base = (size_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (size_t)threadIdx.x * SCAN_ITEMS;
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += (base + k < (size_t)n) ? in[base + k] : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  block_sync();
  if (threadIdx.x == 0) {
    u32 t = 0;
    for (int k = 0; k < SCAN_BLOCK / 64; k++) t += sh[k];
    blocksum[blockIdx.x] = t;
  }
}
// single block: exclusive scan of nb block sums in place; total -> *total
static __global__ void __launch_bounds__(1024) k_scan_top(u32* __restrict__ blocksum, int nb, u32* __restrict__ total) {
  __shared__ u32 sh[1024];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  block_sync();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const u32 v = (i < nb) ? blocksum[i] : 0u;
    sh[threadIdx.x] = v;
    block_sync();
    for (int o = 1; o < 1024; o <<= 1) {
      const u32 t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0u;
      block_sync();
      sh[threadIdx.x] += t;
      block_sync();
    }
    const u32 incl = sh[threadIdx.x];
    if (i < nb) blocksum[i] = carry + incl - v;
    block_sync();
    if (threadIdx.x == 1023) carry += incl;
    block_sync();
  }
  if (threadIdx.x == 0) *total = carry;
}
static __global__ void __launch_bounds__(SCAN_BLOCK) k_scan_final(const u32* __restrict__ in, int n,
                                                           const u32* __restrict__ blocksum,
                                                           u32* __restrict__ out) {
  __shared__ u32 sh[SCAN_BLOCK / 64];
  const size_t base = (size_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (size_t)threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    v[k] = (base + k < (size_t)n) ? in[base + k] : 0u;
    s += v[k];
  }
  // inclusive scan of lane sums across the wave
  u32 incl = s;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) sh[threadIdx.x >> 6] = incl;
  block_sync();
  u32 wave_off = 0;
  for (int k = 0; k < (int)(threadIdx.x >> 6); k++) wave_off += sh[k];
  u32 run = blocksum[blockIdx.x] + wave_off + incl - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < (size_t)n) out[base + k] = run;
    run += v[k];
  }
}

// ------------------------------------------------------------------ segmented reduction
// Lane t owns input positions [t*L, min((t+1)*L, n_in)).  A maximal run of equal bucket id
// that lies entirely inside the chunk is complete -> buckets[bid].  A run cut by the chunk
// start and/or end is a partial: the cut-at-start run goes to slot 2t, the cut-at-end run
// to slot 2t+1 (a run cut on both sides fills 2t and puts an infinity with the same id in
// 2t+1), so pieces of one bucket stay adjacent in slot order and the next level can apply
// the same rule.  Unused slots carry BID_NONE.  Each bucket is written exactly once, at
// the level where its last pieces meet.
// accumulator of one run: XYZZ + mixed additions over affine bases at level 1, Jacobian +
// Jacobian additions over partial slots at the higher levels
template <class CV, bool FIRST>
struct RunAcc;
template <class CV>
struct RunAcc<CV, true> {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
  Xyzz<CV> a;
  // sorted index entry: base index, bit 31 = subtract (negative signed digit): y -> -y, branch-free.
  // -0 comes out as p, which is_zero() accepts, so the (0, 0) infinity marker survives.
  static __device__ __forceinline__ Aff<EA> load_signed(const u32* pts, const u32* idx, long long p) {
    const u32 v = idx[p];
    Aff<EA> q = IO::load_aff(pts + (size_t)(v & ~SIDX_NEG) * IO::AFF_WORDS);
    const EA ny = EA(reduce_to<17>(neg(q.y)));
    q.y = select_el((v & SIDX_NEG) != 0, ny, q.y);
    return q;
  }
  // The same in two halves, for the software-pipelined gather of the level-1 loop: the raw words of the base
  // record are requested one entry ahead and only turned into field elements (unpack, conditional negation) when
  // the entry is consumed — any arithmetic on them at request time would make the wave wait for the data first.
  struct Raw {
    uint4 w[IO::AFF_WORDS / 4];
  };
  static __device__ __forceinline__ Raw load_raw(const u32* pts, u32 v) {
    Raw r;
    const uint4* p = reinterpret_cast<const uint4*>(pts + (size_t)(v & ~SIDX_NEG) * IO::AFF_WORDS);
#pragma unroll
    for (int k = 0; k < IO::AFF_WORDS / 4; k++) r.w[k] = p[k];
    return r;
  }
  // The same through a BUFFER RESOURCE (round 4): AFF_WORDS / 4 buffer_load_dwordx4 off ONE 32-bit offset register with
  // immediate offsets, instead of what hipcc makes of the pointer form — for the 64-byte G1 record five global loads of
  // 8 / 16 / 16 / 12 / 16 bytes (it scalarises the uint4 loads and re-vectorises them) behind 64-bit address
  // arithmetic.  On the rebuilt loop: 1.10 -> 1.05 ms per 16.78 M additions (profiles/r04_ubench_l1loop.txt,
  // variant K).  The table is at most 2^24 records of 64 / 128 bytes: offsets fit 32 bits.
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const u32* pts) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)pts, 0, (int)0xffffffffu, 0x00020000);   // raw, no swizzle, no range clip
  }
  static __device__ __forceinline__ Raw load_raw_buf(__amdgpu_buffer_rsrc_t rs, u32 v) {
    Raw r;
    const int off = (int)((v & ~SIDX_NEG) * (u32)(IO::AFF_WORDS * 4));
#pragma unroll
    for (int k = 0; k < IO::AFF_WORDS / 4; k++) {
      const v4i_ w4 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16 * k, 0, 0);
      r.w[k] = make_uint4((u32)w4.x, (u32)w4.y, (u32)w4.z, (u32)w4.w);
    }
    return r;
  }
  static __device__ __forceinline__ Aff<EA> decode(const Raw& r, u32 v) {
    u32 w[IO::AFF_WORDS];
#pragma unroll
    for (int k = 0; k < IO::AFF_WORDS / 4; k++) {
      w[4 * k] = r.w[k].x;
      w[4 * k + 1] = r.w[k].y;
      w[4 * k + 2] = r.w[k].z;
      w[4 * k + 3] = r.w[k].w;
    }
    Aff<EA> q = IO::load_aff(w);
    const EA ny = EA(reduce_to<17>(neg(q.y)));
    q.y = select_el((v & SIDX_NEG) != 0, ny, q.y);
    return q;
  }
  __device__ __forceinline__ void start(const u32* pts, const u32* idx, long long p) {
    a = xyzz_from_affine<CV>(load_signed(pts, idx, p));
  }
  __device__ __forceinline__ void accumulate(const u32* pts, const u32* idx, long long p) {
    a = xyzz_madd(a, load_signed(pts, idx, p));
  }
  __device__ __forceinline__ void start_q(const Aff<EA>& q) { a = xyzz_from_affine<CV>(q); }
  __device__ __forceinline__ void accumulate_q(const Aff<EA>& q) { a = xyzz_madd(a, q); }
  // the level-1 loop's form: the record as stored plus the sign of the digit; the addition negates y inside its
  // one product (ec.cuh xyzz_madd_lazy) instead of normalising -y first
  static __device__ __forceinline__ Aff<EA> decode_unsigned(const Raw& r) {
    u32 w[IO::AFF_WORDS];
#pragma unroll
    for (int k = 0; k < IO::AFF_WORDS / 4; k++) {
      w[4 * k] = r.w[k].x;
      w[4 * k + 1] = r.w[k].y;
      w[4 * k + 2] = r.w[k].z;
      w[4 * k + 3] = r.w[k].w;
    }
    return IO::load_aff(w);
  }
  __device__ __forceinline__ void start_signed(const Aff<EA>& q, bool negate) {
    Aff<EA> qs = q;
    qs.y = select_el(negate, EA(reduce_to<17>(neg(q.y))), q.y);
    a = xyzz_from_affine<CV>(qs);
  }
  __device__ __forceinline__ void accumulate_signed(const Aff<EA>& q, bool negate) { a = xyzz_madd_lazy(a, q, negate); }
  __device__ __forceinline__ void store(u32* dst) const { IO::store_rec_xyzz(a, dst); }
  __device__ __forceinline__ void negate() { a.Y = typename CV::XY(reduce_to<32>(neg(reduce_to<32>(a.Y)))); }
  // neutral element in the XYZZ record format (chunk cut on both sides)
  __device__ __forceinline__ void store_infinity(u32* dst) const {
    Xyzz<CV> z = a;
    z.ZZ = typename CV::XZZ(el_zero(z.ZZ));
    z.ZZZ = typename CV::XZZZ(el_zero(z.ZZ));
    IO::store_rec_xyzz(z, dst);
  }
};

// The same accumulator kept in LDS (one column of 4 x RAW_WORDS words per lane, word-major so that a
// wave's accesses are conflict-free), for curves whose XYZZ point does not fit the register budget of
// two waves per SIMD: with the G2 accumulator in registers the level-1 kernel takes 256 VGPRs + 159
// AGPRs = ONE wave per SIMD.  The mixed addition is staged so that each coordinate is read right before
// its use and written back right after (LDS traffic is ~150 words per addition against ~28 Fq
// multiplications).
template <class CV>
struct RunAccLds {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
  static constexpr int RW = IO::RW;
  static constexpr int LDS_WORDS = 4 * RW;  // per lane
  u32* col;                                 // this lane's column: word k at col[k * blockDim.x]
  template <class E>
  __device__ __forceinline__ E ld(int c) const {
    u32 w[RW];
#pragma unroll
    for (int k = 0; k < RW; k++) w[k] = col[(size_t)(c * RW + k) * 256];
    return ElemTraits<E>::load_raw(w);
  }
  template <class E>
  __device__ __forceinline__ void st(int c, const E& e) const {
    u32 w[RW];
    ElemTraits<E>::store_raw(e, w);
#pragma unroll
    for (int k = 0; k < RW; k++) col[(size_t)(c * RW + k) * 256] = w[k];
  }
  __device__ __forceinline__ void init(u32* lds) { col = lds + threadIdx.x; }
  __device__ __forceinline__ void put(const Xyzz<CV>& a) const {
    st(0, a.X);
    st(1, a.Y);
    st(2, a.ZZ);
    st(3, a.ZZZ);
  }
  static __device__ __forceinline__ Aff<EA> load_signed(const u32* pts, const u32* idx, long long p) {
    return RunAcc<CV, true>::load_signed(pts, idx, p);
  }
  using Raw = typename RunAcc<CV, true>::Raw;
  static __device__ __forceinline__ Raw load_raw(const u32* pts, u32 v) { return RunAcc<CV, true>::load_raw(pts, v); }
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const u32* pts) { return RunAcc<CV, true>::table_rsrc(pts); }
  static __device__ __forceinline__ Raw load_raw_buf(__amdgpu_buffer_rsrc_t rs, u32 v) { return RunAcc<CV, true>::load_raw_buf(rs, v); }
  static __device__ __forceinline__ Aff<EA> decode(const Raw& r, u32 v) { return RunAcc<CV, true>::decode(r, v); }
  __device__ __forceinline__ void start(const u32* pts, const u32* idx, long long p) {
    start_q(load_signed(pts, idx, p));
  }
  __device__ __forceinline__ void accumulate(const u32* pts, const u32* idx, long long p) {
    accumulate_q(load_signed(pts, idx, p));
  }
  __device__ __forceinline__ void start_q(const Aff<EA>& q) { put(xyzz_from_affine<CV>(q)); }
  // madd-2008-s, staged over the LDS-resident accumulator
  __device__ __forceinline__ void accumulate_q(const Aff<EA>& q) {
    if (is_inf(q)) return;
    const auto ZZ = ld<typename CV::XZZ>(2);
    if (is_zero(ZZ)) {  // accumulator at infinity
      put(xyzz_from_affine<CV>(q));
      return;
    }
    if constexpr (CV::LAZY_FQ2) {
      // the same schedule over the lazily carried Fq2 products (fq2.cuh): 8 carry passes instead of 27
      const auto U2 = mul_lz(q.x, ZZ);
      const auto P = sub(U2, ld<typename CV::XX>(0));
      const auto ZZZ = ld<typename CV::XZZZ>(3);
      const auto R = sub(mul_lz(q.y, ZZZ), ld<typename CV::XY>(1));
      if (is_zero(P)) {
        if (is_zero(R)) {
          put(xyzz_dbl_affine<CV>(q));  // P == Q
        } else {                        // P == -Q
          st(2, typename CV::XZZ(el_zero(q.x)));
          st(3, typename CV::XZZZ(el_zero(q.x)));
        }
        return;
      }
      const auto Pr = reduce_to<32>(P);
      const auto PP = sqr_lz(Pr);
      const auto PPP = mul_lz(Pr, PP);
      st(2, typename CV::XZZ(mul_lz(ZZ, PP)));
      st(3, typename CV::XZZZ(mul_lz(ZZZ, PPP)));
      asm volatile("" ::: "memory");  // keep the reloads below where they are written
      const auto Q = mul_lz(reduce_to<32>(ld<typename CV::XX>(0)), PP);
      const auto X3 = sub_sub2(sqr_lz(reduce_to<32>(R)), PPP, Q);
      st(0, typename CV::XX(X3));
      asm volatile("" ::: "memory");
      const auto Y3 = mulsub_lz(reduce_to<32>(R), reduce_to<32>(sub(Q, X3)), ld<typename CV::XY>(1), PPP);
      st(1, typename CV::XY(Y3));
      return;
    }
    const auto U2 = mul(q.x, ZZ);
    const auto P = sub(U2, ld<typename CV::XX>(0));
    const auto ZZZ = ld<typename CV::XZZZ>(3);
    const auto R = sub(mul(q.y, ZZZ), ld<typename CV::XY>(1));
    if (is_zero(P)) {
      if (is_zero(R)) {
        put(xyzz_dbl_affine<CV>(q));  // P == Q
      } else {                        // P == -Q
        st(2, typename CV::XZZ(el_zero(q.x)));
        st(3, typename CV::XZZZ(el_zero(q.x)));
      }
      return;
    }
    const auto PP = sqr(P);
    const auto PPP = mul(P, PP);
    st(2, typename CV::XZZ(mul(ZZ, PP)));
    st(3, typename CV::XZZZ(mul(ZZZ, PPP)));
    asm volatile("" ::: "memory");  // keep the reloads below where they are written
    const auto Q = mul(ld<typename CV::XX>(0), PP);
    const auto X3 = sub(sqr(R), add(PPP, dbl(Q)));
    st(0, typename CV::XX(X3));
    asm volatile("" ::: "memory");
    const auto Y3 = mulsub(R, sub(Q, X3), ld<typename CV::XY>(1), PPP);
    st(1, typename CV::XY(Y3));
  }
  __device__ __forceinline__ Xyzz<CV> get() const {
    Xyzz<CV> a;
    a.X = ld<typename CV::XX>(0);
    a.Y = ld<typename CV::XY>(1);
    a.ZZ = ld<typename CV::XZZ>(2);
    a.ZZZ = ld<typename CV::XZZZ>(3);
    return a;
  }
  __device__ __forceinline__ void store(u32* dst) const { IO::store_rec_xyzz(get(), dst); }
  __device__ __forceinline__ void negate() const {
    st(1, typename CV::XY(reduce_to<32>(neg(reduce_to<32>(ld<typename CV::XY>(1))))));
  }
  __device__ __forceinline__ void store_infinity(u32* dst) const {
    Xyzz<CV> z = get();
    z.ZZ = typename CV::XZZ(el_zero(z.ZZ));
    z.ZZZ = typename CV::XZZZ(el_zero(z.ZZ));
    IO::store_rec_xyzz(z, dst);
  }
};
template <class CV>
struct RunAcc<CV, false> {
  using IO = CurveIO<CV>;
  Jac<CV> a;
  __device__ __forceinline__ void start(const u32* pts, const u32*, long long p) {
    a = IO::load_rec(pts + (size_t)p * IO::REC_WORDS);
  }
  __device__ __forceinline__ void accumulate(const u32* pts, const u32*, long long p) {
    a = jac_add(a, IO::load_rec(pts + (size_t)p * IO::REC_WORDS));
  }
  __device__ __forceinline__ void store(u32* dst) const { IO::store_rec_jac(a, dst); }
};

template <class CV, bool FIRST, bool LAZY = false>
__device__ __forceinline__ void
segreduce_lane(const int t, const u32* __restrict__ bid_in, const u32* __restrict__ idx_in,
               const u32* __restrict__ pts_in,  // FIRST: affine bases; else Jacobian slots
               const u32* __restrict__ d_count, int n_in_static, int L,
               u32* __restrict__ buckets, u32* __restrict__ bid_out, u32* __restrict__ pts_out) {
  using IO = CurveIO<CV>;
  // levels >= 2: d_count is the number of partial slots still alive after the run merge;
  // 0 means every bucket is already complete and the level has nothing to do
  if (!FIRST && *d_count == 0) return;
  const long long n_in = FIRST ? (long long)(*d_count) : (long long)n_in_static;
  // level 1 reads (index, bucket id) pairs (handed over through idx_in); the generic levels a plain id array
  const uint2* ent = reinterpret_cast<const uint2*>(idx_in);
  auto bid_at = [&](long long p) -> u32 {
    if constexpr (FIRST) return ent[p].y;
    else return bid_in[p];
  };
  const long long s = (long long)t * L;
  u32 head_bid = BID_NONE, tail_bid = BID_NONE;
  bool live = s < n_in;
  if (!FIRST && live) {  // chunk of empty slots: nothing to reduce (independent loads, pipelined)
    const long long e0 = (s + L < n_in) ? (s + L) : n_in;
    bool any = false;
    for (long long p = s; p < e0; p++) any = any || (bid_in[p] != BID_NONE);
    live = any;
  }
  if (live) {
    const long long e = (s + L < n_in) ? (s + L) : n_in;
    const u32 first_bid = bid_at(s);
    const bool cb = (s > 0) && (bid_at(s - 1) == first_bid) && (first_bid != BID_NONE);
    const u32 last_bid = bid_at(e - 1);
    const bool cf = (e < n_in) && (bid_at(e) == last_bid) && (last_bid != BID_NONE);
    u32 cur = BID_NONE;
    bool cur_cb = false;
    using Acc = std::conditional_t<(FIRST && CV::LDS_ACC), RunAccLds<CV>, RunAcc<CV, FIRST>>;
    Acc acc;
    if constexpr (FIRST && CV::LDS_ACC) {
      extern __shared__ u32 ozk_acc_lds[];
      acc.init(ozk_acc_lds);
    }
    if constexpr (FIRST) {
      // level 1, software-pipelined gather: while entry p is added, the base record of entry p + 1 (its index
      // word is already here) and the index word of entry p + 2 are in flight; nothing is computed on them
      // until they are consumed, so the dependent idx -> base chain never stalls the additions
      // (every entry of the sorted array is a live base index at this level)
      // The look-ahead loads are UNCONDITIONAL (positions clamped to the chunk's last entry, which is simply read
      // again): a load under `if (p + 1 < e)` makes every look-ahead register a phi of "loaded" and "kept", and the
      // compiler then carries the whole 16-word record and the index words through ~45 v_mov per addition.
      const u32 n_e = (u32)(e - s);         // entries of this chunk (<= L)
      const uint2* ent_c = ent + s;
      const uint2 e_first = ent_c[0];
      uint2 e_next = ent_c[n_e > 1 ? 1 : 0];
      u32 v_cur = e_first.x;
      const __amdgpu_buffer_rsrc_t table = Acc::table_rsrc(pts_in);
      typename Acc::Raw r = Acc::load_raw_buf(table, v_cur);
      u32 b_cur = first_bid;
      for (u32 k = 0; k < n_e; k++) {
        const u32 b = b_cur;
        const bool negate = (v_cur & SIDX_NEG) != 0;
        const auto q = [&] {
          if constexpr (LAZY) return Acc::decode_unsigned(r);
          else return Acc::decode(r, v_cur);
        }();
        // request entry k + 1's record (its pair arrived one iteration ago) and the pair of entry k + 2; both are
        // consumed one iteration later
        const u32 k2 = (k + 2 < n_e) ? k + 2 : n_e - 1;
        const uint2 en = e_next;
        r = Acc::load_raw_buf(table, en.x);
        b_cur = en.y;
        v_cur = en.x;
        e_next = ent_c[k2];
        if (b != cur) {
          if (cur != BID_NONE) {
            if (cur_cb) {
              head_bid = cur;
              acc.store(pts_out + (size_t)(2 * (size_t)t) * IO::REC_WORDS);
            } else {
              acc.store(buckets + (size_t)cur * IO::REC_WORDS);
            }
          }
          cur = b;
          cur_cb = (k == 0) && cb;
          if constexpr (LAZY) acc.start_signed(q, negate);
          else acc.start_q(q);
        } else {
          if constexpr (LAZY) acc.accumulate_signed(q, negate);
          else acc.accumulate_q(q);
        }
      }
    } else {
    for (long long p = s; p < e; p++) {
      const u32 b = bid_in[p];
      if (b != cur) {
        if (cur != BID_NONE) {  // run [.., p) ended inside the chunk
          if (cur_cb) {
            head_bid = cur;
            acc.store(pts_out + (size_t)(2 * (size_t)t) * IO::REC_WORDS);
          } else {
            acc.store(buckets + (size_t)cur * IO::REC_WORDS);
          }
        }
        cur = b;
        cur_cb = (p == s) && cb;
        if (b != BID_NONE) acc.start(pts_in, idx_in, p);
      } else if (b != BID_NONE) {
        acc.accumulate(pts_in, idx_in, p);
      }
    }
    }
    if (cur != BID_NONE) {  // last run of the chunk
      if (cur_cb) {
        head_bid = cur;
        acc.store(pts_out + (size_t)(2 * (size_t)t) * IO::REC_WORDS);
        if (cf) {  // cut on both sides: neutral element keeps the pieces adjacent
          tail_bid = cur;
          if constexpr (FIRST) {
            acc.store_infinity(pts_out + (size_t)(2 * (size_t)t + 1) * IO::REC_WORDS);
          } else {
            IO::store_rec_jac(jac_infinity<CV>(), pts_out + (size_t)(2 * (size_t)t + 1) * IO::REC_WORDS);
          }
        }
      } else if (cf) {
        tail_bid = cur;
        acc.store(pts_out + (size_t)(2 * (size_t)t + 1) * IO::REC_WORDS);
      } else {
        acc.store(buckets + (size_t)cur * IO::REC_WORDS);
      }
    }
  }
  bid_out[2 * (size_t)t] = head_bid;
  bid_out[2 * (size_t)t + 1] = tail_bid;
}

// `clk` (level 1 only; may be null): four words of this launch, zero before it.  Every workgroup leaves
// max(~start) in clk[0] and max(end) in clk[1] (wall_clock64: the constant-rate counter, hipDeviceAttributeWallClockRate),
// so the launch's duration is clk[1] - ~clk[0] without any help from the runtime: HIP events around (or on) the
// dispatch cost the three-stage schedule 4-13 % of its throughput, see ozk_prof_enable.  Round 4: clk[2] += the
// SHADER clock ticks (clock64 = s_memtime: measured at 2.35-2.38 GHz under load against the 100 MHz of
// wall_clock64, profiles/r04_ubench_mont.txt) and clk[3] += the constant-rate ticks that the first wave of every
// workgroup spent in the kernel: clk[2] / clk[3] x the constant rate = the clock the chip actually ran this launch
// at — what makes two boxes' (or two rounds') kernel times comparable (the sysfs sclk read after a run was noise).
template <class CV, bool FIRST, bool LAZY = false>
__global__ void __launch_bounds__(256, (FIRST && CV::LDS_ACC) ? 2 : 1)
k_segreduce(const u32* __restrict__ bid_in, const u32* __restrict__ idx_in, const u32* __restrict__ pts_in,
            const u32* __restrict__ d_count, int n_in_static, int L,
            u32* __restrict__ buckets, u32* __restrict__ bid_out, u32* __restrict__ pts_out,
            int n_lanes, unsigned long long* __restrict__ clk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long w0 = 0, c0 = 0;
  if constexpr (FIRST) {
    if (clk != nullptr && threadIdx.x == 0) {
      w0 = (unsigned long long)wall_clock64();
      c0 = (unsigned long long)clock64();
      atomicMax(&clk[0], ~w0);
    }
  }
  if (t < n_lanes)
    segreduce_lane<CV, FIRST, LAZY>(t, bid_in, idx_in, pts_in, d_count, n_in_static, L, buckets, bid_out, pts_out);
  if constexpr (FIRST) {
    if (clk != nullptr && (threadIdx.x & 63) == 0) {
      const unsigned long long w1 = (unsigned long long)wall_clock64();
      atomicMax(&clk[1], w1);  // (every wave: the last one to finish counts)
      if (threadIdx.x == 0) {
        atomicAdd(&clk[2], (unsigned long long)clock64() - c0);
        atomicAdd(&clk[3], w1 - w0);
      }
    }
  }
}

// the device's constant-rate counter, for calibrating it against the host clock (ozk_prof_enable)
static __global__ void k_read_clock(unsigned long long* out) { out[0] = (unsigned long long)wall_clock64(); }

// The last generic levels (a few hundred lanes and fewer) in ONE single-block launch instead of one
// 6-us launch per level: level after level with a block barrier in between, ping-ponging between the
// two slot arrays, until one lane has seen everything.  Ends with the result in the buckets.
template <class CV>
__global__ void __launch_bounds__(256)
k_segreduce_small(const u32* __restrict__ d_count, int n_in, int L, u32* __restrict__ buckets,
                  u32* __restrict__ bid_a, u32* __restrict__ pts_a, u32* __restrict__ bid_b, u32* __restrict__ pts_b) {
  if (*d_count == 0) return;
  u32 *bi = bid_a, *pi = pts_a, *bo = bid_b, *po = pts_b;
  while (true) {
    const int lanes = (n_in + L - 1) / L;
    for (int t = threadIdx.x; t < lanes; t += blockDim.x)
      segreduce_lane<CV, false>(t, bi, nullptr, pi, d_count, n_in, L, buckets, bo, po);
    if (lanes == 1) break;
    block_sync();  // (also orders the global writes of this level before the next level's reads)
    n_in = 2 * lanes;
    u32* tb = bi; bi = bo; bo = tb;
    u32* tp = pi; pi = po; po = tp;
  }
}

// Run merge between level 1 and the generic levels.  After level 1 a bucket cut by lane
// boundaries is a run of adjacent slots with the same id (two for almost every bucket; a few
// more for the top window, whose digits have fewer significant bits).  Every run of at most
// RUN_MAX slots is summed by the lane that owns its first slot and written to its bucket; all
// its slots die.  Longer runs (skewed digit distributions) are left to the generic levels and
// counted in *remaining.  Each lane decides from bid_in alone, so there are no races.
constexpr int RUN_MAX = 16;
template <class CV>
__global__ void __launch_bounds__(256)
k_runmerge(const u32* __restrict__ bid_in, const u32* __restrict__ pts, int n_slots,
           u32* __restrict__ buckets, u32* __restrict__ bid_out, u32* __restrict__ remaining) {
  using IO = CurveIO<CV>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  u32 alive = 0;
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    // lane t owns slots 2t+1 and 2t+2; lane 0 also owns slot 0 (always empty: no run starts there)
    const int x = 2 * t + 1 + k;
    if (x >= n_slots) break;
    const u32 bx = bid_in[x];
    u32 out = BID_NONE;
    if (bx != BID_NONE) {
      int s = x, e = x;
      while (s > 0 && x - s < RUN_MAX && bid_in[s - 1] == bx) s--;
      while (e + 1 < n_slots && e - x < RUN_MAX && bid_in[e + 1] == bx) e++;
      const bool is_short = (x - s < RUN_MAX) && (e - x < RUN_MAX) && (e - s + 1 <= RUN_MAX);
      if (!is_short) {
        out = bx;
      } else if (s == x) {  // owner of the run: sum it
        // level-1 pieces are all XYZZ records (the neutral placeholder of a chunk cut on both sides
        // included): 12M + 2S general additions, no conversion
        // A run is [tail piece of lane t | head piece of lane t + 1 | (placeholder of t + 1 | head piece of t + 2)* ]:
        // it starts at an odd slot (a tail piece), every other odd slot inside it is the infinity placeholder of a
        // chunk cut on both sides, every even slot a head piece — so only the even slots are added (round 4: a run of
        // four slots costs two additions instead of three, one of them with infinity; almost every wave holds one)
        Xyzz<CV> acc = IO::load_xyzz(pts + (size_t)s * IO::REC_WORDS);
        for (int q = s + 1; q <= e; q += 2) acc = xyzz_add(acc, IO::load_xyzz(pts + (size_t)q * IO::REC_WORDS));
        IO::store_rec_xyzz(acc, buckets + (size_t)bx * IO::REC_WORDS);
      }
    }
    bid_out[x] = out;
    alive += (out != BID_NONE);
  }
  if (t == 0 && n_slots > 0) {
    bid_out[0] = bid_in[0];
    alive += (bid_in[0] != BID_NONE);
  }
  for (int o = 32; o > 0; o >>= 1) alive += __shfl_xor(alive, o);
  if ((threadIdx.x & 63) == 0 && alive) atomicAdd(remaining, alive);
}

// ------------------------------------------------------------------ window sums
// Per window the elements e = 0..m-1 carry (A_e, R_e) with
//     S_w = sum_e A_e + 2^g * sum_e e * R_e.
// Level 0: elements are the buckets (A = infinity, R = B_d, g = 0).  One level with
// segments of S = 2^sg elements turns segment j into
//     A'_j = sum A_e + 2^g * sum (e - jS) R_e,   R'_j = sum R_e,   g' = g + sg.
// FIRSTLEVEL reads R from the bucket array (skipping buckets whose histogram count is 0:
// they were never written, so the bucket array needs no clearing) and has no A.
template <class CV, bool FIRSTLEVEL>
__global__ void __launch_bounds__(256)
k_wsum(const u32* __restrict__ A_in, const u32* __restrict__ R_in, const u32* __restrict__ hist,
       int m_in, int S, int g, u32* __restrict__ A_out, u32* __restrict__ R_out, int m_out, int W, int prio) {
  using IO = CurveIO<CV>;
  if (prio) __builtin_amdgcn_s_setprio(3);  // tail phase: see k_wsum_wave
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m_out * W) return;
  const int w = t / m_out, j = t - w * m_out;
  const size_t base = (size_t)w * m_in;
  const int lo = j * S;
  int hi = lo + S;
  if (hi > m_in) hi = m_in;
  Jac<CV> run = jac_infinity<CV>(), ws = jac_infinity<CV>(), asum = jac_infinity<CV>();
  for (int e = hi - 1; e >= lo; e--) {
    if constexpr (FIRSTLEVEL) {
      // buckets no entry was sorted into were never written: treat as infinity
      if (hist[base + e] != 0) run = jac_add(run, IO::load_rec(R_in + (base + e) * IO::REC_WORDS));
    } else {
      run = jac_add(run, IO::load_jac(R_in + (base + e) * IO::JAC_WORDS));
    }
    if (e > lo) ws = jac_add(ws, run);  // after the loop: sum (e - lo) * R_e
    if constexpr (!FIRSTLEVEL) asum = jac_add(asum, IO::load_jac(A_in + (base + e) * IO::JAC_WORDS));
  }
  for (int k = 0; k < g; k++) ws = jac_dbl(ws);
  if constexpr (!FIRSTLEVEL) ws = jac_add(ws, asum);
  IO::store_jac(ws, A_out + ((size_t)w * m_out + j) * IO::JAC_WORDS);
  IO::store_jac(run, R_out + ((size_t)w * m_out + j) * IO::JAC_WORDS);
}

// Wave-cooperative window-sum level: one wave per 64 consecutive elements of one window.
//   T_l  = sum_{l' >= l} R_l'          suffix scan by shuffles           (log2 steps)
//   V_l  = A_l + 2^g * (l >= 1 ? T_l : 0)                                (g doublings, in parallel)
//   A'   = sum_l V_l                   tree reduction by shuffles        (log2 steps)
//   R'   = T_0
// since sum_{l>=1} T_l = sum_l l * R_l.  13 dependent additions per 64x reduction instead of
// the 24 per 8x of the serial form: the upper levels are latency-bound, not throughput-bound.
template <class CV>
__device__ __forceinline__ Jac<CV> shfl_down_jac(const Jac<CV>& p, int o) {
  Jac<CV> r;
  r.X = shfl_down_el(p.X, o);
  r.Y = shfl_down_el(p.Y, o);
  r.Z = shfl_down_el(p.Z, o);
  return r;
}

// lanes l < valid hold element l = (A, T := R); returns (A', R') of the 64x coarser element in lane 0
template <class CV>
__device__ __forceinline__ void wave_combine(Jac<CV>& A, Jac<CV>& T, int l, int valid, int g) {
  // suffix scan of R
  for (int o = 1; o < valid; o <<= 1) {
    const Jac<CV> t = shfl_down_jac(T, o);
    const Jac<CV> s = jac_add(T, t);
    if (l + o < 64) T = s;
  }
  // V = A + 2^g * (l >= 1 ? T : inf)
  Jac<CV> U = T;
  for (int k = 0; k < g; k++) U = jac_dbl(U);
  const Jac<CV> AV = jac_add(A, U);
  Jac<CV> V = (l >= 1) ? AV : A;
  // tree reduction of V
  int top = 1;
  while (top < valid) top <<= 1;
  for (int o = top >> 1; o > 0; o >>= 1) {
    const Jac<CV> v = shfl_down_jac(V, o);
    const Jac<CV> s = jac_add(V, v);
    if (l < o) V = s;
  }
  A = V;
}

template <class CV>
__global__ void __launch_bounds__(64)
k_wsum_wave(const u32* __restrict__ A_in, const u32* __restrict__ R_in, int m_in, int g,
            u32* __restrict__ A_out, u32* __restrict__ R_out, int m_out, int W) {
  using IO = CurveIO<CV>;
  // latency-bound tail: when it overlaps another MSM's throughput kernels on the same SIMDs,
  // issue priority keeps its dependent chains from being starved (measured: pipelined step
  // 3.09 ms without, vs max(head, tail) = 2.45 ms)
  __builtin_amdgcn_s_setprio(3);
  const int wave = blockIdx.x;  // one 64-lane block per (window, group of 64 elements)
  if (wave >= m_out * W) return;
  const int w = wave / m_out, j = wave - w * m_out;
  const int l = threadIdx.x & 63;
  const int e = j * 64 + l;
  int valid = m_in - j * 64;
  if (valid > 64) valid = 64;
  Jac<CV> A = jac_infinity<CV>(), T = jac_infinity<CV>();
  if (e < m_in) {
    A = IO::load_jac(A_in + ((size_t)w * m_in + e) * IO::JAC_WORDS);
    T = IO::load_jac(R_in + ((size_t)w * m_in + e) * IO::JAC_WORDS);
  }
  wave_combine<CV>(A, T, l, valid, g);
  if (l == 0) {
    IO::store_jac(A, A_out + ((size_t)w * m_out + j) * IO::JAC_WORDS);
    IO::store_jac(T, R_out + ((size_t)w * m_out + j) * IO::JAC_WORDS);
  }
}

// Fused first level: every lane sums S consecutive buckets serially (as k_wsum<FIRSTLEVEL>), then the
// wave combines its 64 lane results while they are still in registers: 64 S buckets -> one element,
// no intermediate round trip and one launch less on the critical path of the window sums
// (2 S + 13 dependent additions for a 64 S-fold reduction; the unfused pair needed 2 S + 13 for S x 64
// too, but as two latency-bound kernels with a 2^c / S-element hand-off).
template <class CV>
__global__ void __launch_bounds__(64)
k_wsum_fused(const u32* __restrict__ buckets, const u32* __restrict__ hist, int m_in, int S, int sg,
             u32* __restrict__ A_out, u32* __restrict__ R_out, int m_out, int W, int prio) {
  using IO = CurveIO<CV>;
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int wave = blockIdx.x;
  if (wave >= m_out * W) return;
  const int w = wave / m_out, j = wave - w * m_out;
  const int l = threadIdx.x & 63;
  const int nseg = (m_in + S - 1) / S;  // segments of this window
  int valid = nseg - j * 64;
  if (valid > 64) valid = 64;
  const size_t base = (size_t)w * m_in;
  const int lo = (j * 64 + l) * S;
  int hi = lo + S;
  if (hi > m_in) hi = m_in;
  Jac<CV> run = jac_infinity<CV>(), ws = jac_infinity<CV>();
  for (int e = hi - 1; e >= lo; e--) {  // (empty for lanes past the window's end)
    if (hist[base + e] != 0) run = jac_add(run, IO::load_rec(buckets + (base + e) * IO::REC_WORDS));
    if (e > lo) ws = jac_add(ws, run);  // after the loop: sum (e - lo) * B_e
  }
  wave_combine<CV>(ws, run, l, valid, sg);
  if (l == 0) {
    IO::store_jac(ws, A_out + ((size_t)w * m_out + j) * IO::JAC_WORDS);
    IO::store_jac(run, R_out + ((size_t)w * m_out + j) * IO::JAC_WORDS);
  }
}

// ------------------------------------------------------------------ bucket arrays of several slices
// The host entry point accumulates index-range slices of one MSM into separate bucket arrays (each slice's
// kernels run while the next slice is still on its way over PCIe); this adds arrays 1 .. K-1 into array 0,
// one lane per bucket, so that ONE latency-bound tail serves the whole call.
struct SliceBuckets {
  const u32* buckets[16];
  const u32* hist[16];
};
template <class CV>
__global__ void __launch_bounds__(256) k_bucket_combine(SliceBuckets t, int K, u32* __restrict__ buckets0,
                                                        u32* __restrict__ hist0, size_t NB) {
  using IO = CurveIO<CV>;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= NB) return;
  bool have = hist0[e] != 0, changed = false;
  Jac<CV> acc = jac_infinity<CV>();
  if (have) acc = IO::load_rec(buckets0 + e * IO::REC_WORDS);
  for (int k = 1; k < K; k++) {
    if (t.hist[k][e] == 0) continue;  // never written
    acc = jac_add(acc, IO::load_rec(t.buckets[k] + e * IO::REC_WORDS));
    changed = true;
  }
  if (changed) {
    IO::store_rec_jac(acc, buckets0 + e * IO::REC_WORDS);
    hist0[e] = 1;
  }
}

// ------------------------------------------------------------------ finalize
// affine-normalise and emit the reference's return layout: per Fq value 64 B LE, upper
// 32 B zero (VariableBaseMSM.cu:1655-1659); infinity -> (0, 1, 0) (BNG1.java:163-172).
template <class CV>
__device__ void write_normalised(const Jac<CV>& r, u32* out) {
  using EA = typename CV::EA;
  using ET = ElemTraits<EA>;
  constexpr int OW = 2 * ET::WORDS;  // wire-out words per coordinate
  if (is_inf(r)) {
    ET::to_wire_out(EA(el_zero(r.X)), out);
    ET::to_wire_out(EA(el_one(r.X)), out + OW);
    ET::to_wire_out(EA(el_zero(r.X)), out + 2 * OW);
  } else {
    const auto zi = inv(r.Z);
    const auto zi2 = sqr(zi);
    ET::to_wire_out(EA(reduce_to<17>(mul(r.X, zi2))), out);
    ET::to_wire_out(EA(reduce_to<17>(mul(r.Y, mul(zi2, zi)))), out + OW);
    ET::to_wire_out(EA(el_one(r.X)), out + 2 * OW);
  }
}

// A_w holds S_w (one element per window after the last wsum level, with 2^g * 0 * R = 0).
//
// The Horner chain over the windows is serial: one wave (one lane, or a lane pair for G2), issue
// priority raised.  (Round 1 also carried a variant that declared a whole CU's register files so that no
// accumulation wave shared its SIMDs.  A process abort seen with it was the sort's missing barrier wait,
// see block_sync() in curve.cuh, not this kernel: with that fixed the variant passed the whole GPU suite,
// and measured 519.0 against 516.1 Mscalar-mul/s pipelined — inside the run-to-run noise — so it was
// removed; profiles/r02_finalize_exclusive_ab.txt.)
template <class CV>
__global__ void __launch_bounds__(64) k_finalize(const u32* __restrict__ A_w, const u32* __restrict__ R_w, int m,
                                                 int g, int W, int c, int sd, u32* __restrict__ out) {
  using IO = CurveIO<CV>;
  __shared__ u32 sw[64 * IO::JAC_WORDS];  // S_w of up to 64 windows at a time
  if (blockIdx.x != 0) return;
  __builtin_amdgcn_s_setprio(3);
  // The last window-sum level left m (<= a few) elements (A_j, R_j) per window:
  //     S_w = sum_j A_j + 2^g * sum_j j * R_j     (+ sum_j R_j with signed digits: bucket b weighs b + 1)
  // — window w is finished by lane w (G2: by the lane pair 2w, 2w + 1, see Fe2L in fq2.cuh), all windows
  // in parallel; then lane 0 (the pair 0, 1) runs Horner over the S_w.
  using CP = typename CV::Pair;
  constexpr int LP = CV::PAIR_LANES;
  constexpr int WB = 64 / LP;  // windows per batch
  const int l = threadIdx.x;
  const int wl = l / LP;
  Jac<CP> r = jac_infinity<CP>();
  bool started = false;
  for (int w0 = ((W - 1) / WB) * WB; w0 >= 0; w0 -= WB) {  // (one batch unless the window size is forced tiny)
    const int w = w0 + wl;
    if (w < W) {
      Jac<CP> run = jac_infinity<CP>(), acc = jac_infinity<CP>(), asum = jac_infinity<CP>();
      for (int j = m - 1; j >= 0; j--) {
        asum = jac_add(asum, to_pair(IO::load_jac(A_w + ((size_t)w * m + j) * IO::JAC_WORDS)));
        if (j >= 1) {
          run = jac_add(run, to_pair(IO::load_jac(R_w + ((size_t)w * m + j) * IO::JAC_WORDS)));
          acc = jac_add(acc, run);
        }
      }
      if (m > 1) {
        for (int k = 0; k < g; k++) acc = jac_dbl(acc);
        asum = jac_add(asum, acc);
      }
      if (sd) {
        run = jac_add(run, to_pair(IO::load_jac(R_w + (size_t)w * m * IO::JAC_WORDS)));  // + R_0: all buckets
        asum = jac_add(asum, run);
      }
      if (l % LP == 0) IO::store_jac(from_pair(asum), sw + (size_t)wl * IO::JAC_WORDS);
    }
    __builtin_amdgcn_wave_barrier();
    if (l < LP) {
      const u32* sp = sw + opaque_zero();
      const int top = (W - w0 < WB) ? (W - w0) : WB;
      for (int i = top - 1; i >= 0; i--) {
        const Jac<CP> si = to_pair(IO::load_jac(sp + (size_t)i * IO::JAC_WORDS));
        if (!started) {   // the top window: nothing to double yet (c doublings of infinity were 45 us for G1, 130 for G2)
          r = si;
          started = true;
          continue;
        }
        for (int k = 0; k < c; k++) r = jac_dbl(r);
        r = jac_add(r, si);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (l == 0) write_normalised<CV>(from_pair(r), out);
}

// sum of k points given in wire-out format (affine or (0,1,0)); used for the multi-GPU reduce
// (records `stride_words` apart: the sharded double MSM gathers G1 || G2 records of 576 bytes)
template <class CV>
__global__ void __launch_bounds__(64) k_points_sum(const u32* __restrict__ pts, int k, int stride_words, u32* __restrict__ out) {
  using EA = typename CV::EA;
  using ET = ElemTraits<EA>;
  constexpr int OW = 2 * ET::WORDS;
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  __builtin_amdgcn_s_setprio(3);  // one lane of one wave, usually beside another MSM's accumulation kernel
  pts += opaque_zero();
  Jac<CV> r = jac_infinity<CV>();
  for (int i = 0; i < k; i++) {
    const u32* p = pts + (size_t)i * stride_words;
    Jac<CV> q;
    q.X = ET::from_wire_out(p);
    q.Y = ET::from_wire_out(p + OW);
    q.Z = ET::from_wire_out(p + 2 * OW);
    r = jac_add(r, q);
  }
  write_normalised<CV>(r, out);
}

// ------------------------------------------------------------------ synthetic bases
// Bench/test utility (BASELINE.md config inputs): P_i = k_i * G with k_i = splitmix64(seed + i),
// written in the JNI wire-in format (affine, Z = 1).  Known discrete logs make the full-size
// MSM checkable on the CPU: sum s_i P_i = (sum s_i k_i mod r) * G.
OZK_HD u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <class CV>
__global__ void __launch_bounds__(256) k_gen_bases(u64 seed, int n, const u32* __restrict__ gen_wire,
                                                   u32* __restrict__ out_wire) {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
  using ET = ElemTraits<EA>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 k = splitmix64(seed + (u64)i);
  if (k == 0) k = 1;
  Aff<EA> g;
  g.x = ET::from_wire(gen_wire);
  g.y = ET::from_wire(gen_wire + IO::CW);
  Jac<CV> r = jac_infinity<CV>();
  for (int b = 63; b >= 0; b--) {
    r = jac_dbl(r);
    if ((k >> b) & 1) r = jac_madd(r, g);
  }
  const auto zi = inv(r.Z);
  const auto zi2 = sqr(zi);
  u32* o = out_wire + (size_t)i * IO::WIRE_JAC_WORDS;
  ET::to_wire(EA(reduce_to<17>(mul(r.X, zi2))), o);
  ET::to_wire(EA(reduce_to<17>(mul(r.Y, mul(zi2, zi)))), o + IO::CW);
  ET::to_wire(EA(el_one(r.X)), o + 2 * IO::CW);
}

#endif  // __HIPCC__

}  // namespace ozk
