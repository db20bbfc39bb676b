// Fixed-base batch MSM (G1 / G2) and the Fr vector-times-scalar product: kernels, host
// driver and C ABI (include/ozk.h).
//
// Replaces fixed_batch_MSMG1 / G2, fixed_double_batch_MSM, field_MSM and their kernels
// (algebra_msm_FixedBaseMSM.cu:750-1266) and the three JNI natives of
// algebra.msm.FixedBaseMSM (.cu:1276-1558).  Value computed per scalar, as
// FixedBaseMSM.serialMSM (FixedBaseMSM.java:141-167) with the table of
// getWindowTable (FixedBaseMSM.java:71-99):
//     result_i = sum_{w < outerc} digit_w(s_i) * 2^(w*windowSize) * B
// Outputs are affine-normalised (so keys produced here take the Z == 1 fast path of the
// variable-base MSM) and written in the reference's layout: 64-byte BIG-endian
// coordinates (FixedBaseMSM.cu:783-787).
//
// Device pipeline:
//   k_fb_chain    D[j] = 2^j * B, j < outerc*windowSize           (one serial doubling chain)
//   k_fb_level k  table[w][2^k + j] = table[w][j] + D[w*ws + k]   (windowSize launches, all
//                 windows and all j in parallel; the reference re-doubles per window and adds
//                 popcount(i) points per entry, FixedBaseMSM.cu:851-992)
//   k_fb_main     per scalar: gather-add one table entry per window (Jacobian).  Default form
//                 (k_fb_main_glv): the scalar is split by the GLV endomorphism (glv.cuh) into two
//                 127-bit halves that share ONE table of ceil(128 / windowSize) windows — s B =
//                 +-(sum_w T[w][d1_w]) + phi(+-(sum_w T[w][d2_w])), phi(X, Y, Z) = (beta X, Y, Z) —
//                 which halves the serial doubling chain (the longest single item: 254 doublings on
//                 one lane, 1.2 ms for G1 and 5 ms for G2) and the table
//   k_fb_norm     per lane a batch of results: one shared inversion (Montgomery's trick),
//                 big-endian stores
#include <algorithm>
#include <new>
#include <vector>

#include "fq2.cuh"
#include "glv.cuh"
#include "msm_var.cuh"  // RunAccLds: the LDS-resident XYZZ accumulator (G2)
#include "host_ctx.h"
#include "pin_cache.h"

namespace ozk {

template <class CV>
__device__ __forceinline__ Fe<FqParams, 16> glv_beta_fixed() {
  if constexpr (CurveIO<CV>::CW == 16) return fe_const<FqParams, 16>(GlvConsts::BETA_G2);
  else return fe_const<FqParams, 16>(GlvConsts::BETA_G1);
}

template <class CV>
__global__ void __launch_bounds__(64) k_fb_chain(const u32* __restrict__ base_wire, int total, u32* __restrict__ D) {
  using IO = CurveIO<CV>;
  using CP = typename CV::Pair;  // G2: the chain runs on a lane pair (Fe2L, fq2.cuh)
  if (blockIdx.x != 0 || threadIdx.x >= CV::PAIR_LANES) return;
  base_wire += opaque_zero();
  Jac<CP> p = to_pair(IO::jac_from_wire(base_wire));
  for (int j = 0; j < total; j++) {
    if (threadIdx.x == 0) IO::store_jac(from_pair(p), D + (size_t)j * IO::JAC_WORDS);
    p = jac_dbl(p);
  }
}

// level k: entries [2^k, 2^(k+1)) of every window.  table[w][0] = infinity (zeroed).
template <class CV>
__global__ void __launch_bounds__(256) k_fb_level(u32* __restrict__ table, const u32* __restrict__ D, int outerc,
                                                  int ws, int k) {
  using IO = CurveIO<CV>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = 1 << k;
  if (t >= outerc * half) return;
  const int w = t >> k, j = t & (half - 1);
  const size_t row = (size_t)w << ws;
  const Jac<CV> add = IO::load_jac(D + (size_t)(w * ws + k) * IO::JAC_WORDS);
  Jac<CV> r;
  if (j == 0) {
    r = add;
  } else {
    r = jac_add(IO::load_jac(table + (row + j) * IO::JAC_WORDS), add);
  }
  IO::store_jac(r, table + (row + half + j) * IO::JAC_WORDS);
}

template <class CV>
__global__ void __launch_bounds__(256) k_fb_main(const u32* __restrict__ scalars, const u32* __restrict__ table,
                                                 int n, int outerc, int ws, u32* __restrict__ jac_out) {
  using IO = CurveIO<CV>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[8];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = sp[0], b = sp[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
  s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  Jac<CV> acc = jac_infinity<CV>();
  for (int w = 0; w < outerc; w++) {
    // digit = bits [w*ws, w*ws + ws) (testBit loop, FixedBaseMSM.java:153-159); ws may exceed 16
    u32 d = 0;
    const int bit = w * ws;
    if (bit < 256) {
      const int wi = bit >> 5, sh = bit & 31;
      unsigned long long v = s[wi];
      if (wi + 1 < 8) v |= (unsigned long long)s[wi + 1] << 32;
      d = (u32)(v >> sh) & ((1u << ws) - 1u);
      // ws <= 22 and sh <= 31: two words always suffice
    }
    if (d != 0) acc = jac_add(acc, IO::load_jac(table + (((size_t)w << ws) + d) * IO::JAC_WORDS));
  }
  IO::store_jac(acc, jac_out + (size_t)i * IO::JAC_WORDS);
}

// GLV form of the gather-add: one table of `oc` windows covering 128 bits, two digit strings.
template <class CV>
__global__ void __launch_bounds__(256) k_fb_main_glv(const u32* __restrict__ scalars, const u32* __restrict__ table,
                                                     int n, int oc, int ws, u32* __restrict__ jac_out) {
  using IO = CurveIO<CV>;
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
  Jac<CV> part[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const u32* k = h ? k2 : k1;
    Jac<CV> acc = jac_infinity<CV>();
    for (int w = 0; w < oc; w++) {
      const int bit = w * ws;
      u32 d = 0;
      if (bit < 128) {
        const int wi = bit >> 5, sh = bit & 31;
        unsigned long long v = k[wi];
        if (wi + 1 < 4) v |= (unsigned long long)k[wi + 1] << 32;
        d = (u32)(v >> sh) & ((1u << ws) - 1u);
      }
      if (d != 0) acc = jac_add(acc, IO::load_jac(table + (((size_t)w << ws) + d) * IO::JAC_WORDS));
    }
    if (h ? n2 : n1) acc = jac_neg(acc);
    part[h] = acc;
  }
  // phi(X, Y, Z) = (beta X, Y, Z)
  part[1].X = typename CV::EX(reduce_to<32>(scale(part[1].X, glv_beta_fixed<CV>())));
  IO::store_jac(jac_add(part[0], part[1]), jac_out + (size_t)i * IO::JAC_WORDS);
}

// 64-byte big-endian store of one Fq value: 8 zero words, then the value's words reversed
// and byte-swapped (FixedBaseMSM.cu:740-748 swap_helper layout)
template <class P, int B>
__device__ __forceinline__ void store_be64(const Fe<P, B>& e, u32* p) {
  u32 w[8];
  from_mont(e, w);
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) p[8 + i] = __builtin_bswap32(w[7 - i]);
}
template <int B>
__device__ __forceinline__ void store_be_coord(const Fe<FqParams, B>& e, u32* p) { store_be64(e, p); }
template <int B>
__device__ __forceinline__ void store_be_coord(const Fe2<B>& e, u32* p) {
  store_be64(e.c0, p);
  store_be64(e.c1, p + 16);
}

constexpr int FB_BATCH = 8;
// lane t normalises FB_BATCH results: prefix products of Z, one inversion, back-substitution.
// out element i is at out + i*out_stride_words (+ coordinate offsets).
// compact = 1: X|Y|Z as 32-byte little-endian values (the wire-IN format of the variable-base natives,
// VariableBaseMSM.java:221-228), so that keys go from the setup to the prover without being reformatted
// (SURVEY.md §8f N4); compact = 0: the reference's 64-byte big-endian coordinates.
template <class CV>
__global__ void __launch_bounds__(256) k_fb_norm(const u32* __restrict__ jac, int n, u32* __restrict__ out,
                                                 int out_stride_words, int compact) {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
  using ET = ElemTraits<EA>;
  using EZ32 = decltype(reduce_to<32>(typename CV::EZ()));
  constexpr int OW = 2 * ET::WORDS;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lanes = (n + FB_BATCH - 1) / FB_BATCH;
  if (t >= lanes) return;
  // the lane's batch is INTERLEAVED — elements t, t + lanes, t + 2 lanes, ... — so that consecutive lanes touch
  // consecutive records (any grouping serves Montgomery's trick; consecutive elements per lane made every access
  // a 100-B record at a 864-B stride).  k_fb_norm 216 -> 203 us at 2^20: the kernel is bound by its 2^17 safegcd
  // inversions (45 % of its instructions), not by these accesses.
  auto at = [&](int k) { return (size_t)k * (size_t)lanes + (size_t)t; };
  // prefix[k] = product of the non-zero Z_0..Z_k
  EZ32 prefix[FB_BATCH];
  EZ32 run = EZ32(el_one(prefix[0]));
#pragma unroll
  for (int k = 0; k < FB_BATCH; k++) {
    if (at(k) < (size_t)n) {
      const auto Z = reduce_to<32>(ElemTraits<typename CV::EZ>::load_raw(jac + at(k) * IO::JAC_WORDS + 2 * IO::RW));
      if (!is_zero(Z)) run = EZ32(mul(run, Z));
    }
    prefix[k] = run;
  }
  EZ32 invrun = EZ32(inv(run));
#pragma unroll
  for (int k = FB_BATCH - 1; k >= 0; k--) {
    if (at(k) < (size_t)n) {
      const Jac<CV> p = IO::load_jac(jac + at(k) * IO::JAC_WORDS);
      u32* o = out + at(k) * out_stride_words;
      const auto Z = reduce_to<32>(p.Z);
      if (is_zero(Z)) {  // (0, 1, 0), BNG1.java:163-166
        if (compact) {
          ET::to_wire(EA(el_zero(p.X)), o);
          ET::to_wire(EA(el_one(p.X)), o + ET::WORDS);
          ET::to_wire(EA(el_zero(p.X)), o + 2 * ET::WORDS);
        } else {
          store_be_coord(EA(el_zero(p.X)), o);
          store_be_coord(EA(el_one(p.X)), o + OW);
          store_be_coord(EA(el_zero(p.X)), o + 2 * OW);
        }
      } else {
        // 1/Z_k = invrun * prefix[k-1];  invrun <- invrun * Z_k
        EZ32 zi = invrun;
        if (k > 0) zi = EZ32(mul(invrun, prefix[k - 1]));
        invrun = EZ32(mul(invrun, Z));
        const auto zi2 = sqr(zi);
        const EA ax = EA(reduce_to<17>(mul(p.X, zi2))), ay = EA(reduce_to<17>(mul(p.Y, mul(zi2, zi))));
        if (compact) {
          ET::to_wire(ax, o);
          ET::to_wire(ay, o + ET::WORDS);
          ET::to_wire(EA(el_one(p.X)), o + 2 * ET::WORDS);
        } else {
          store_be_coord(ax, o);
          store_be_coord(ay, o + OW);
          store_be_coord(EA(el_one(p.X)), o + 2 * OW);
        }
      }
    }
  }
}

// The GLV gather-add over an AFFINE table: lane t normalises FB_BATCH consecutive table entries with one
// shared inversion (k_fb_table_affine), after which every gather is a mixed XYZZ addition (madd-2008-s,
// 8M + 2S) instead of a full Jacobian one (11M + 5S); 16 gathers per scalar against ~16 multiplications
// per table entry for the normalisation.  Infinity entries (entry 0 of every window; every entry when the
// base is infinity) become the (0, 0) marker.
template <class CV>
__global__ void __launch_bounds__(256) k_fb_table_affine(const u32* __restrict__ jac, int n, u32* __restrict__ aff,
                                                         u32* __restrict__ aff_phi) {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
  using EZ32 = decltype(reduce_to<32>(typename CV::EZ()));
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lanes = (n + FB_BATCH - 1) / FB_BATCH;
  if (t >= lanes) return;
  // the lane's batch is INTERLEAVED — elements t, t + lanes, t + 2 lanes, ... — so that consecutive lanes touch
  // consecutive records (any grouping serves Montgomery's trick; consecutive elements per lane made every access
  // a 100-B record at a 864-B stride).  k_fb_norm 216 -> 203 us at 2^20: the kernel is bound by its 2^17 safegcd
  // inversions (45 % of its instructions), not by these accesses.
  auto at = [&](int k) { return (size_t)k * (size_t)lanes + (size_t)t; };
  EZ32 prefix[FB_BATCH];
  EZ32 run = EZ32(el_one(prefix[0]));
#pragma unroll
  for (int k = 0; k < FB_BATCH; k++) {
    if (at(k) < (size_t)n) {
      const auto Z = reduce_to<32>(ElemTraits<typename CV::EZ>::load_raw(jac + at(k) * IO::JAC_WORDS + 2 * IO::RW));
      if (!is_zero(Z)) run = EZ32(mul(run, Z));
    }
    prefix[k] = run;
  }
  EZ32 invrun = EZ32(inv(run));
#pragma unroll
  for (int k = FB_BATCH - 1; k >= 0; k--) {
    if (at(k) < (size_t)n) {
      const Jac<CV> p = IO::load_jac(jac + at(k) * IO::JAC_WORDS);
      const auto Z = reduce_to<32>(p.Z);
      Aff<EA> q;
      if (is_zero(Z)) {
        q.x = EA(el_zero(p.X));
        q.y = EA(el_zero(p.X));
      } else {
        EZ32 zi = invrun;
        if (k > 0) zi = EZ32(mul(invrun, prefix[k - 1]));
        invrun = EZ32(mul(invrun, Z));
        const auto zi2 = sqr(zi);
        q.x = EA(reduce_to<17>(mul(p.X, zi2)));
        q.y = EA(reduce_to<17>(mul(p.Y, mul(zi2, zi))));
      }
      IO::store_aff(q, aff + at(k) * IO::AFF_WORDS);
      // the same entry under the endomorphism, phi(x, y) = (beta x, y): the second half scalar gathers from this
      // copy, so both halves add into ONE accumulator (k_fb_main_glv_affine); (0, 0) stays (0, 0)
      q.x = EA(reduce_to<17>(scale(q.x, glv_beta_fixed<CV>())));
      IO::store_aff(q, aff_phi + at(k) * IO::AFF_WORDS);
    }
  }
}

// s B = s2 (s1 s2 (sum_w T[w][d1_w]) + sum_w T_phi[w][d2_w]) with s1, s2 = +-1 the signs of the two half scalars: ONE
// accumulator takes the 2 ceil(128 / ws) gathers of both halves, negated between the halves when the signs differ
// and at the end when the second is negative.  (Round 2 kept one accumulator per half and joined them with a
// general XYZZ addition: for G2 that is two 72-word points live at once — 256 VGPRs, 40 of them spilled, and
// 720 bytes of scratch per lane; and 17 multiplications per scalar the single accumulator does not need.)
template <class CV>
__global__ void __launch_bounds__(256, CV::LDS_ACC ? 2 : 1) k_fb_main_glv_affine(const u32* __restrict__ scalars, const u32* __restrict__ aff,
                                                            const u32* __restrict__ aff_phi, int n, int oc, int ws,
                                                            u32* __restrict__ jac_out) {
  using IO = CurveIO<CV>;
  using EA = typename CV::EA;
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
  Aff<EA> inf;
  inf.x = EA(el_zero(inf.x));
  inf.y = EA(el_zero(inf.x));
  // G2: the accumulator lives in LDS (as in the variable-base level 1) — two waves per SIMD instead of one
  using Acc = std::conditional_t<CV::LDS_ACC, RunAccLds<CV>, RunAcc<CV, true>>;
  Acc acc;
  if constexpr (CV::LDS_ACC) {
    extern __shared__ u32 ozk_acc_lds[];
    acc.init(ozk_acc_lds);
  }
  acc.start_q(inf);
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    const u32 kk0 = h ? k2[0] : k1[0], kk1 = h ? k2[1] : k1[1], kk2 = h ? k2[2] : k1[2], kk3 = h ? k2[3] : k1[3];
    const u32* tab = h ? aff_phi : aff;
    for (int w = 0; w < oc; w++) {
      const int bit = w * ws;
      u32 d = 0;
      if (bit < 128) {
        const int wi = bit >> 5, sh = bit & 31;
        const u32 lo = wi == 0 ? kk0 : wi == 1 ? kk1 : wi == 2 ? kk2 : kk3;
        const u32 hi = wi == 0 ? kk1 : wi == 1 ? kk2 : wi == 2 ? kk3 : 0u;
        const unsigned long long v = (unsigned long long)lo | ((unsigned long long)hi << 32);
        d = (u32)(v >> sh) & ((1u << ws) - 1u);
      }
      if (d != 0) acc.accumulate_q(IO::load_aff(tab + (((size_t)w << ws) + d) * IO::AFF_WORDS));
    }
    if (h == 0 ? (n1 != n2) : n2) acc.negate();
  }
  Xyzz<CV> r;
  if constexpr (CV::LDS_ACC) r = acc.get();
  else r = acc.a;
  IO::store_jac(xyzz_to_jac(r), jac_out + (size_t)i * IO::JAC_WORDS);
}

// x_i * b mod r, 64-byte big-endian out (field_MSM, FixedBaseMSM.cu:1241-1266)
__global__ void __launch_bounds__(256) k_field_mul(const u32* __restrict__ in, int n, u32* __restrict__ out) {
  using ET = ElemTraits<Fe<FrParams, 17>>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto b = ET::from_wire(in + (size_t)n * 8);
  const auto x = ET::from_wire(in + (size_t)i * 8);
  store_be64(mul(x, b), out + (size_t)i * 16);
}

struct FbLayout {
  u32 *D, *table, *aff, *aff_phi, *jac, *base;   // base: 256 B for a copy of the base point
  size_t bytes;
};
template <class CV>
static FbLayout fb_layout(int outerc, int ws, int n, void* wsp, size_t wsb) {
  using IO = CurveIO<CV>;
  FbLayout L;
  Bump b(wsp, wsb);
  L.D = b.take<u32>((size_t)outerc * ws * IO::JAC_WORDS);
  L.table = b.take<u32>(((size_t)outerc << ws) * IO::JAC_WORDS);
  L.aff = b.take<u32>(((size_t)outerc << ws) * IO::AFF_WORDS);  // the table again, affine (GLV form) ...
  L.aff_phi = b.take<u32>(((size_t)outerc << ws) * IO::AFF_WORDS);  // ... and its image under the endomorphism
  L.jac = b.take<u32>((size_t)n * IO::JAC_WORDS);
  L.base = b.take<u32>(64);
  L.bytes = b.off;
  return L;
}

static int fb_check_args(int outerc, int ws, int n) {
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (ws < 1 || ws > 22) return fail(OZK_E_INVALID, "windowSize %d out of range [1, 22]", ws);
  if (outerc < 1 || (long long)outerc * ws > 512) return fail(OZK_E_INVALID, "outerc %d out of range", outerc);
  return OZK_OK;
}

// The table of a call (fb_table: doubling chain, level launches, affine copy) and the per-scalar part over any
// sub-range of the scalars (fb_apply: gather-add + batched normalisation) are separate steps, so that the host
// entry point can download the results of one range while the next is computed.
struct FbPlan {
  bool glv, affine;
  int wt, oc;   // table window size / windows actually used
  const u32 *aff = nullptr, *aff_phi = nullptr;   // affine table and its image under the endomorphism; null: the
                                                  // copies in the call's workspace (FbLayout)
};
// form of the table for a call of n scalars (no device work)
static FbPlan fb_choose(int outerc, int ws, int n) {
  // GLV form when the caller's windows cover a whole Fr scalar (they always do in the reference:
  // outerc = ceil(scalarSize / windowSize), FixedBaseMSM.java:71-99); the plain form otherwise
  const bool glv = env_int("OZK_MSM_GLV", 1) != 0 && (long long)outerc * ws >= 254;
  // In the GLV form the table's window size is the library's choice (s B does not depend on it): the
  // caller's (17 bits at 2^20) balances ONE 254-bit digit string against the table; with two 127-bit
  // strings over one table a narrower window is cheaper.  Cost in multiplications: 2 ceil(128/w) mixed
  // additions of ~10 per scalar + ~32 per table entry (level addition + normalisation).  Never wider
  // than the caller's, so the workspace the caller sized always suffices.
  int wt = ws;
  if (glv) {
    double best = 0;
    for (int w = 1; w <= ws; w++) {
      const double o = (128 + w - 1) / w;
      const double cost = 2.0 * o * 10.0 * (double)n + o * 32.0 * (double)(1u << w);  // (the doubling chain is o w ~ 128 long either way)
      if (w == 1 || cost < best) { best = cost; wt = w; }
    }
    wt = env_int("OZK_FB_WS", wt);
    if (wt < 1 || wt > ws) wt = ws;
  }
  FbPlan fp;
  fp.glv = glv;
  fp.affine = glv && env_int("OZK_FB_AFFINE", 1) != 0;
  fp.wt = wt;
  fp.oc = glv ? (128 + wt - 1) / wt : outerc;
  return fp;
}
// builds the table of `fp` from the base point at d_base; the Jacobian table and the doubling chain live in the call's
// workspace, the affine copies go to (aff, aff_phi) when given (a cached table), else into the workspace as well
template <class CV>
static int fb_build(const FbPlan& fp, int outerc, int ws, int n, const void* d_base, void* wsp, size_t wsb, hipStream_t st,
                    u32* aff = nullptr, u32* aff_phi = nullptr) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  using IO = CurveIO<CV>;
  const FbLayout L = fb_layout<CV>(outerc, ws, n, wsp, wsb);
  if (L.bytes > wsb) return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
  const int TB = 256;
  const int wt = fp.wt, oc = fp.oc;
  hipLaunchKernelGGL((k_fb_chain<CV>), dim3(1), dim3(64), 0, st, (const u32*)d_base, oc * wt, L.D);
  // entry 0 of every window is infinity: clear those records (all-zero Jacobian has Z = 0)
  OZK_HIP(hipMemset2DAsync(L.table, ((size_t)1 << wt) * IO::JAC_WORDS * 4, 0, IO::JAC_WORDS * 4, oc, st));
  for (int k = 0; k < wt; k++) {
    const int tot = oc << k;
    hipLaunchKernelGGL((k_fb_level<CV>), dim3((tot + TB - 1) / TB), dim3(TB), 0, st, L.table, L.D, oc, wt, k);
  }
  if (fp.affine) {
    const int entries = oc << wt;
    const int tl = (entries + FB_BATCH - 1) / FB_BATCH;
    hipLaunchKernelGGL((k_fb_table_affine<CV>), dim3((tl + TB - 1) / TB), dim3(TB), 0, st, L.table, entries,
                       aff ? aff : L.aff, aff_phi ? aff_phi : L.aff_phi);
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}
template <class CV>
static int fb_table(int outerc, int ws, int n, const void* d_base, void* wsp, size_t wsb, hipStream_t st, FbPlan* plan) {
  *plan = fb_choose(outerc, ws, n);
  return fb_build<CV>(*plan, outerc, ws, n, d_base, wsp, wsb, st);
}

// scalars [lo, lo + cnt) of the call's n: d_scalars / d_out point at element 0
template <class CV>
static int fb_apply(const FbPlan& fp, int outerc, int ws, int n, int lo, int cnt, const void* d_scalars, void* d_out,
                    int out_stride_words, void* wsp, size_t wsb, hipStream_t st, int compact) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  using IO = CurveIO<CV>;
  const FbLayout L = fb_layout<CV>(outerc, ws, n, wsp, wsb);
  const int TB = 256;
  const u32* sc = (const u32*)d_scalars + (size_t)lo * 8;
  u32* jac = L.jac + (size_t)lo * IO::JAC_WORDS;
  if (fp.affine) {
    const size_t acc_lds = CV::LDS_ACC ? (size_t)RunAccLds<CV>::LDS_WORDS * TB * sizeof(u32) : 0;  // 72 KiB for G2
    if (acc_lds > 65536)
      OZK_HIP(hipFuncSetAttribute((const void*)(k_fb_main_glv_affine<CV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)acc_lds));
    hipLaunchKernelGGL((k_fb_main_glv_affine<CV>), dim3((cnt + TB - 1) / TB), dim3(TB), acc_lds, st, sc,
                       fp.aff ? fp.aff : L.aff, fp.aff_phi ? fp.aff_phi : L.aff_phi, cnt, fp.oc, fp.wt, jac);
  } else if (fp.glv)
    hipLaunchKernelGGL((k_fb_main_glv<CV>), dim3((cnt + TB - 1) / TB), dim3(TB), 0, st, sc, L.table, cnt, fp.oc, fp.wt,
                       jac);
  else
    hipLaunchKernelGGL((k_fb_main<CV>), dim3((cnt + TB - 1) / TB), dim3(TB), 0, st, sc, L.table, cnt, outerc, ws, jac);
  const int lanes = (cnt + FB_BATCH - 1) / FB_BATCH;
  hipLaunchKernelGGL((k_fb_norm<CV>), dim3((lanes + TB - 1) / TB), dim3(TB), 0, st, jac, cnt,
                     (u32*)d_out + (size_t)lo * out_stride_words, out_stride_words, compact);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

template <class CV>
static int fixed_batch_dev(int outerc, int ws, int n, const void* d_base, const void* d_scalars, void* d_out,
                           int out_stride_words, void* wsp, size_t wsb, hipStream_t st, int compact = 0) {
  FbPlan fp;
  int rc = fb_table<CV>(outerc, ws, n, d_base, wsp, wsb, st, &fp);
  if (rc) return rc;
  return fb_apply<CV>(fp, outerc, ws, n, 0, n, d_scalars, d_out, out_stride_words, wsp, wsb, st, compact);
}

// ---- table cache ------------------------------------------------------------------------------
// The window table depends only on (base, windowSize, outerc) and the size class of the call; the reference's Java
// side computes it once per key element (getWindowTable, FixedBaseMSM.java:71-99) and its native side rebuilds it
// inside every call (FixedBaseMSM.cu:851-992) — as this library did through round 3: doubling chain + levels +
// batched normalisation = 0.65 ms (G1) / 2.1 ms (G2) of a 2.0 / 5.5 ms call at 2^20, the serial chain alone half of
// it.  A setup issues many calls over the same generator (one per 2^20-scalar chunk and per key vector), so the
// affine table of the default (GLV) form is kept in library-owned HBM, keyed by (device, curve, outerc, windowSize,
// chosen table window, base bytes): four per device, least recently used out, pinned while a caller is between
// tab_get and its last launch — the FFT plan cache's rules (fft.hip).  OZK_FB_TABLE_CACHE=0: per-call tables.
struct FbTab : PinCacheItem {
  int type = 0, outerc = 0, ws = 0, wt = 0, oc = 0;
  uint8_t base[192] = {0};
  uint8_t* mem = nullptr;
  u32 *aff = nullptr, *aff_phi = nullptr, *d_base = nullptr;
  hipEvent_t ready = nullptr;
  bool same_key(const FbTab& o) const {
    return type == o.type && outerc == o.outerc && ws == o.ws && wt == o.wt && oc == o.oc &&
           memcmp(base, o.base, type == OZK_G1 ? 96 : 192) == 0;
  }
};
// Round 4 (pin_cache.h): the table is allocated, built and freed with NO lock held (round 3 held one process-wide mutex
// across a ~13 ms hipMalloc, the build enqueue and the evicted table's device-synchronising hipFree); a table is
// cached from the SECOND call with its key on (the first leaves a marker and builds in the caller's workspace, as
// before the cache existed), so a setup whose batches all differ in size pays nothing for tables nobody reuses;
// and the tables of a device stay inside a byte budget (OZK_FB_TABLE_CACHE_MB, default 1024; four tables at most).
constexpr int FB_TABS = 4;  // per device
static PinCache<FbTab> g_tabs;
static PinCacheLimits tab_limits() {
  long mb = env_int("OZK_FB_TABLE_CACHE_MB", 1024);
  if (mb < 0) mb = 0;
  return PinCacheLimits{FB_TABS, (size_t)mb << 20};
}
// (no lock held) releases the tables' device memory; the caller's current device is restored
static void tabs_free(std::vector<FbTab*>& dead) {
  if (dead.empty()) return;
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (FbTab* t : dead) {
    if (t->mem) {
      (void)hipSetDevice(t->device);
      (void)hipFree(t->mem);   // waits for the kernels already enqueued
    }
    if (t->ready) (void)hipEventDestroy(t->ready);
    delete t;
  }
  dead.clear();
  if (have_cur) (void)hipSetDevice(cur);
}
static void tab_release(FbTab* t) {
  std::vector<FbTab*> dead;
  g_tabs.release(t, tab_limits(), &dead);
  tabs_free(dead);
}
struct TabPin {
  FbTab* t = nullptr;
  ~TabPin() {
    if (t) tab_release(t);
  }
};
void fb_table_cache_release() {
  std::vector<FbTab*> dead;
  g_tabs.drain(&dead);
  tabs_free(dead);
}

// the cached affine table for (current device, curve, outerc, ws, fp.wt, base), PINNED, its build enqueued on `st`
// (in the caller's workspace) if it is new; fp.aff / fp.aff_phi are set.  Only for fp.affine.  *out stays null when
// nothing is cached for this call (first sight of the key, or no room): the caller builds a per-call table.
template <class CV>
static int tab_get(FbPlan* fp, int outerc, int ws, int n, const uint8_t* base_host, void* wsp, size_t wsb, hipStream_t st,
                   FbTab** out) {
  using IO = CurveIO<CV>;
  constexpr int type = std::is_same<CV, G1Cfg>::value ? OZK_G1 : OZK_G2;
  constexpr size_t base_bytes = type == OZK_G1 ? 96 : 192;
  *out = nullptr;
  FbTab key;
  OZK_HIP(hipGetDevice(&key.device));
  key.type = type;
  key.outerc = outerc;
  key.ws = ws;
  key.wt = fp->wt;
  key.oc = fp->oc;
  memcpy(key.base, base_host, base_bytes);
  const size_t entries = (size_t)fp->oc << fp->wt;
  const size_t half = pad256(entries * IO::AFF_WORDS * 4);
  const size_t bytes = 2 * half + 256;
  std::vector<FbTab*> dead;
  FbTab* t = nullptr;
  const auto res = g_tabs.acquire(
      key, bytes, tab_limits(), env_int("OZK_FB_TABLE_CACHE_SECOND_USE", 1) != 0,
      [&]() -> FbTab* {
        FbTab* nt = new (std::nothrow) FbTab();
        if (nt) {
          nt->device = key.device;
          nt->type = key.type;
          nt->outerc = key.outerc;
          nt->ws = key.ws;
          nt->wt = key.wt;
          nt->oc = key.oc;
          memcpy(nt->base, key.base, sizeof(key.base));
        }
        return nt;
      },
      &t, &dead);
  tabs_free(dead);   // evicted tables: hipFree outside the cache's lock
  if (res == PinCache<FbTab>::PER_CALL) return OZK_OK;
  if (res == PinCache<FbTab>::BUILD_FAILED) return OZK_OK;   // (a concurrent build of this key failed: per-call table)
  if (res == PinCache<FbTab>::BUILD) {
    int rc = OZK_OK;
    hipError_t e = hipMalloc((void**)&t->mem, bytes);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&t->ready, hipEventDisableTiming);
    if (e != hipSuccess) rc = fail(OZK_E_NOMEM, "fixed-base table allocation (%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (!rc) {
      t->aff = (u32*)t->mem;
      t->aff_phi = (u32*)(t->mem + half);
      t->d_base = (u32*)(t->mem + 2 * half);
      // (from the table's own copy of the base: host memory that outlives the asynchronous copy)
      e = hipMemcpyAsync(t->d_base, t->base, base_bytes, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) rc = fail(OZK_E_NO_DEVICE, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    }
    if (!rc) rc = fb_build<CV>(*fp, outerc, ws, n, t->d_base, wsp, wsb, st, t->aff, t->aff_phi);
    if (!rc && hipEventRecord(t->ready, st) != hipSuccess) rc = fail(OZK_E_NO_DEVICE, "hipEventRecord failed");
    g_tabs.publish(t, rc == OZK_OK, &dead);
    tabs_free(dead);
    if (rc) return rc;
  }
  const hipError_t we = hipStreamWaitEvent(st, t->ready, 0);   // a no-op on the stream that built it
  if (we != hipSuccess) {
    tab_release(t);
    return fail(OZK_E_NO_DEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(we));
  }
  fp->aff = t->aff;
  fp->aff_phi = t->aff_phi;
  *out = t;
  return OZK_OK;
}

// table for a call whose base is known to the HOST (the `*_host` entry points, ozk_fixed_batch_msm_base_dev): the
// cached one when the form allows, else built in the workspace from a copy of the base at `d_base_scratch`
template <class CV>
static int fb_table_host_base(int outerc, int ws, int n, const uint8_t* base_host, u32* d_base_scratch, void* wsp, size_t wsb,
                              hipStream_t st, FbPlan* fp, TabPin* pin) {
  constexpr size_t base_bytes = std::is_same<CV, G1Cfg>::value ? 96 : 192;
  *fp = fb_choose(outerc, ws, n);
  if (fp->affine && env_int("OZK_FB_TABLE_CACHE", 1)) {
    const int rc = tab_get<CV>(fp, outerc, ws, n, base_host, wsp, wsb, st, &pin->t);
    if (rc || pin->t) return rc;   // (no cached table for this call: build one in the workspace, below)
  }
  OZK_HIP(hipMemcpyAsync(d_base_scratch, base_host, base_bytes, hipMemcpyHostToDevice, st));
  OZK_HIP(hipStreamSynchronize(st));   // `base_host` is the caller's (pageable) memory
  return fb_build<CV>(*fp, outerc, ws, n, d_base_scratch, wsp, wsb, st);
}

}  // namespace ozk

using namespace ozk;

extern "C" {

size_t ozk_fixed_batch_msm_workspace_bytes(int32_t outerc, int32_t ws, int32_t n, int32_t bn_type) {
  if (fb_check_args(outerc, ws, n)) return 0;
  return bn_type == OZK_G1 ? fb_layout<G1Cfg>(outerc, ws, n, nullptr, 0).bytes
                           : fb_layout<G2Cfg>(outerc, ws, n, nullptr, 0).bytes;
}

int ozk_fixed_batch_msm_dev(int32_t outerc, int32_t ws, int32_t n, const void* d_base, const void* d_scalars,
                            int32_t bn_type, void* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!d_base || !d_scalars || !d_out || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  int rc = fb_check_args(outerc, ws, n);
  if (rc) return rc;
  if (bn_type == OZK_G1)
    return fixed_batch_dev<G1Cfg>(outerc, ws, n, d_base, d_scalars, d_out, 48, d_workspace, workspace_bytes,
                                  (hipStream_t)stream);
  return fixed_batch_dev<G2Cfg>(outerc, ws, n, d_base, d_scalars, d_out, 96, d_workspace, workspace_bytes,
                                (hipStream_t)stream);
}

int ozk_fixed_batch_msm_compact_dev(int32_t outerc, int32_t ws, int32_t n, const void* d_base, const void* d_scalars,
                                    int32_t bn_type, void* d_out, void* d_workspace, size_t workspace_bytes,
                                    void* stream) {
  if (!d_base || !d_scalars || !d_out || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  int rc = fb_check_args(outerc, ws, n);
  if (rc) return rc;
  if (bn_type == OZK_G1)
    return fixed_batch_dev<G1Cfg>(outerc, ws, n, d_base, d_scalars, d_out, 24, d_workspace, workspace_bytes,
                                  (hipStream_t)stream, 1);
  return fixed_batch_dev<G2Cfg>(outerc, ws, n, d_base, d_scalars, d_out, 48, d_workspace, workspace_bytes,
                                (hipStream_t)stream, 1);
}

// the same with the base point given as HOST bytes (wire format, 96 / 192 B; scalars and results stay in device
// memory), so that the window table can come from the cache: what a caller that issues many batches over one
// generator should use.  compact != 0: 32-byte coordinates (ozk_fixed_batch_msm_compact_dev's layout).
int ozk_fixed_batch_msm_base_dev(int32_t outerc, int32_t ws, int32_t n, const uint8_t* base_host, const void* d_scalars,
                                 int32_t bn_type, void* d_out, int32_t compact, void* d_workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!base_host || !d_scalars || !d_out || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  int rc = fb_check_args(outerc, ws, n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  FbPlan fp;
  TabPin pin;  // held until the launches below are enqueued
  if (bn_type == OZK_G1) {
    const FbLayout L = fb_layout<G1Cfg>(outerc, ws, n, d_workspace, workspace_bytes);
    if (L.bytes > workspace_bytes) return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    if ((rc = fb_table_host_base<G1Cfg>(outerc, ws, n, base_host, L.base, d_workspace, workspace_bytes, st, &fp, &pin))) return rc;
    return fb_apply<G1Cfg>(fp, outerc, ws, n, 0, n, d_scalars, d_out, compact ? 24 : 48, d_workspace, workspace_bytes, st, compact ? 1 : 0);
  }
  const FbLayout L = fb_layout<G2Cfg>(outerc, ws, n, d_workspace, workspace_bytes);
  if (L.bytes > workspace_bytes) return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
  if ((rc = fb_table_host_base<G2Cfg>(outerc, ws, n, base_host, L.base, d_workspace, workspace_bytes, st, &fp, &pin))) return rc;
  return fb_apply<G2Cfg>(fp, outerc, ws, n, 0, n, d_scalars, d_out, compact ? 48 : 96, d_workspace, workspace_bytes, st, compact ? 1 : 0);
}

static int fixed_batch_host(int32_t outerc, int32_t ws, int32_t n, const uint8_t* base, const uint8_t* scalars,
                            int32_t bn_type, int32_t task_id, uint8_t* out, int compact) {
  if (!base || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  int rc = fb_check_args(outerc, ws, n);
  if (rc) return rc;
  const bool g1 = bn_type == OZK_G1;
  const size_t sc_bytes = (size_t)n * 32;
  const size_t out_bytes = (size_t)n * (g1 ? 192 : 384) / (compact ? 2 : 1);
  const size_t wsb = ozk_fixed_batch_msm_workspace_bytes(outerc, ws, n, bn_type);
  CtxGuard g;
  if ((rc = ctx_acquire(task_id, &g.c))) return rc;
  HostCtx* c = g.c;
  const size_t a0 = 256, a1 = a0 + pad256(sc_bytes), a2 = a1 + pad256(out_bytes);
  if ((rc = ctx_reserve(c, a2 + wsb + 256))) return rc;
  uint8_t* d = c->arena;
  hipStream_t st = c->st[0], cp = c->st[2];
  // The table (~1 ms at window 17, mostly the serial doubling chain) is built while the scalars are on their way;
  // the per-scalar part then runs in ranges, each range's results going back to the host (on the copy stream,
  // staged through the pinned ring) while the next ranges are computed: the output is 6 (compact: 3) times the
  // input, so the call is bound by the download and nothing else should add to it.
  const int osw = (g1 ? 48 : 96) / (compact ? 2 : 1);
  FbPlan fp;
  TabPin pin;  // the cached table stays pinned until every range below is enqueued
  rc = g1 ? fb_table_host_base<G1Cfg>(outerc, ws, n, base, (u32*)d, d + a2, wsb, st, &fp, &pin)
          : fb_table_host_base<G2Cfg>(outerc, ws, n, base, (u32*)d, d + a2, wsb, st, &fp, &pin);
  if (rc) return rc;
  if ((rc = staged_h2d(c, d + a0, scalars, sc_bytes, cp))) return rc;
  OZK_HIP(hipEventRecord(c->ev[0], cp));
  OZK_HIP(hipStreamWaitEvent(st, c->ev[0], 0));
  // (4 ranges: 8 measured 0.2 ms faster at 2^20, but with occasional 10-30 ms calls that 4 never showed)
  int K = env_int("OZK_FB_HOST_RANGES", 4);
  if (K > MAX_SLICES) K = MAX_SLICES;
  if (K > n / 4096) K = n / 4096;
  if (K < 1) K = 1;
  const int per = (((n + K - 1) / K) + 255) & ~255;
  int ranges = 0;
  for (int lo = 0; lo < n; lo += per, ranges++) {
    const int cnt = n - lo < per ? n - lo : per;
    rc = g1 ? fb_apply<G1Cfg>(fp, outerc, ws, n, lo, cnt, d + a0, d + a1, osw, d + a2, wsb, st, compact)
            : fb_apply<G2Cfg>(fp, outerc, ws, n, lo, cnt, d + a0, d + a1, osw, d + a2, wsb, st, compact);
    if (rc) return rc;
    OZK_HIP(hipEventRecord(c->slice_ev[ranges], st));
  }
  // one continuous staged download; a piece is queued once the ranges it covers have been ordered before it
  struct Gate {
    HostCtx* c;
    hipStream_t cp;
    size_t range_bytes;
    int ranges, waited;
  } gate = {c, cp, (size_t)per * osw * 4, ranges, 0};
  auto gate_fn = [](void* a, size_t end) -> int {
    Gate* g = (Gate*)a;
    int need = (int)((end + g->range_bytes - 1) / g->range_bytes);
    if (need > g->ranges) need = g->ranges;
    for (; g->waited < need; g->waited++) OZK_HIP(hipStreamWaitEvent(g->cp, g->c->slice_ev[g->waited], 0));
    return OZK_OK;
  };
  return staged_d2h_gated(c, out, d + a1, out_bytes, cp, gate_fn, &gate);
}

int ozk_fixed_batch_msm_host(int32_t outerc, int32_t ws, int32_t out_len, int32_t inner_len, int32_t n,
                             int32_t scalar_size, const uint8_t* base, const uint8_t* scalars, int32_t bn_type,
                             int32_t task_id, uint8_t* out) {
  (void)out_len; (void)inner_len; (void)scalar_size;  // table shape is implied by outerc / windowSize
  return fixed_batch_host(outerc, ws, n, base, scalars, bn_type, task_id, out, 0);
}

int ozk_fixed_batch_msm_compact_host(int32_t outerc, int32_t ws, int32_t n, const uint8_t* base,
                                     const uint8_t* scalars, int32_t bn_type, int32_t task_id, uint8_t* out) {
  return fixed_batch_host(outerc, ws, n, base, scalars, bn_type, task_id, out, 1);
}

int ozk_fixed_double_batch_msm_host(int32_t outerc1, int32_t ws1, int32_t outerc2, int32_t ws2, int32_t out_len1,
                                    int32_t inner_len1, int32_t out_len2, int32_t inner_len2, int32_t n,
                                    const uint8_t* base_g1, const uint8_t* base_g2, const uint8_t* scalars,
                                    int32_t task_id, uint8_t* out) {
  (void)out_len1; (void)inner_len1; (void)out_len2; (void)inner_len2;
  if (!base_g1 || !base_g2 || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  int rc = fb_check_args(outerc1, ws1, n);
  if (rc) return rc;
  if ((rc = fb_check_args(outerc2, ws2, n))) return rc;
  // per element G1 (3 x 64 BE) || G2 (6 x 64 BE) = 576 B  (FixedBaseMSM.cu:1479-1482)
  const size_t sc_bytes = (size_t)n * 32, out_bytes = (size_t)n * 576;
  const size_t w1 = fb_layout<G1Cfg>(outerc1, ws1, n, nullptr, 0).bytes;
  const size_t w2 = fb_layout<G2Cfg>(outerc2, ws2, n, nullptr, 0).bytes;
  const size_t wsb = w1 > w2 ? w1 : w2;
  CtxGuard g;
  if ((rc = ctx_acquire(task_id, &g.c))) return rc;
  HostCtx* c = g.c;
  const size_t a0 = 512, a1 = a0 + pad256(sc_bytes), a2 = a1 + pad256(out_bytes);
  if ((rc = ctx_reserve(c, a2 + wsb + 256))) return rc;
  uint8_t* d = c->arena;
  hipStream_t st = c->st[0];
  // (tables from the cache when the form allows; the G1 and G2 parts share the workspace, so a G2 table that has to
  // be built is built after the G1 part's kernels, in stream order)
  if ((rc = staged_h2d(c, d + a0, scalars, sc_bytes, st))) return rc;
  FbPlan fp1, fp2;
  TabPin pin1, pin2;
  if ((rc = fb_table_host_base<G1Cfg>(outerc1, ws1, n, base_g1, (u32*)d, d + a2, wsb, st, &fp1, &pin1))) return rc;
  if ((rc = fb_apply<G1Cfg>(fp1, outerc1, ws1, n, 0, n, d + a0, d + a1, 144, d + a2, wsb, st, 0))) return rc;
  if ((rc = fb_table_host_base<G2Cfg>(outerc2, ws2, n, base_g2, (u32*)(d + 256), d + a2, wsb, st, &fp2, &pin2))) return rc;
  rc = fb_apply<G2Cfg>(fp2, outerc2, ws2, n, 0, n, d + a0, d + a1 + 192, 144, d + a2, wsb, st, 0);
  if (rc) return rc;
  return staged_d2h(c, out, d + a1, out_bytes, st);
}

int ozk_field_batch_mul_dev(const void* d_in, int32_t n, void* d_out, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!d_in || !d_out || n <= 0) return fail(OZK_E_INVALID, "bad argument");
  hipLaunchKernelGGL(k_field_mul, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u32*)d_in, n,
                     (u32*)d_out);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

int ozk_field_batch_mul_host(const uint8_t* in, int32_t n, int32_t task_id, uint8_t* out) {
  if (!in || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range", n);
  const size_t in_bytes = (size_t)(n + 1) * 32, out_bytes = (size_t)n * 64;
  CtxGuard g;
  int rc = ctx_acquire(task_id, &g.c);
  if (rc) return rc;
  HostCtx* c = g.c;
  const size_t a1 = pad256(in_bytes);
  if ((rc = ctx_reserve(c, a1 + out_bytes + 256))) return rc;
  uint8_t* d = c->arena;
  hipStream_t st = c->st[0];
  if ((rc = staged_h2d(c, d, in, in_bytes, st))) return rc;
  if ((rc = ozk_field_batch_mul_dev(d, n, d + a1, st))) return rc;
  return staged_d2h(c, out, d + a1, out_bytes, st);
}

}  // extern "C"
