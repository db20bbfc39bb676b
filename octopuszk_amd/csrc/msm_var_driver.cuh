// Variable-base MSM: host driver templates (plan, workspace layout, stage launchers, host-buffer and
// prepared-bases paths), shared by the two translation units that instantiate them — msm_var.hip
// (G1 + the C ABI) and msm_var_g2.hip (G2) — so that the two heavy instantiations compile in parallel.
#pragma once
#include <hip/hip_ext.h>
#include <pthread.h>
#include <stdlib.h>

#include <atomic>
#include <vector>

#ifndef OZK_WITH_G2
#define OZK_WITH_G2 1
#endif
#include "msm_var.cuh"
#include "host_ctx.h"
#include "fq2.cuh"

namespace ozk {

// Optional per-launch timing of the dominant kernel (level-1 segmented reduce); bench.py reads the statistics after
// its timed region (roofline.achieved).  Two sources:
//   PROF_CLOCK  the kernel's own waves stamp the device's constant-rate clock (k_segreduce, `clk`): no runtime
//               involvement, so the schedule being measured is the schedule that runs;
//   PROF_EVENTS HIP events on the dispatch (hipExtLaunchKernelGGL start / stop events).  Exact too, but an event-
//               carrying dispatch costs the three-stage schedule throughput (684 -> 597 Mscalar-mul/s with every
//               launch timed, 660 with one in ten; separate hipEventRecord calls around the launch: 589), so
//               bench.py uses it in a second pass as the cross-check, not inside the timed region.
enum { PROF_OFF = 0, PROF_EVENTS = 1, PROF_CLOCK = 2 };
// Process-wide, and since round 4 safe under concurrent callers: the launch counter, the event pool and the clock
// buffer are claimed under `mu` (one uncontended lock per level-1 launch, only while profiling is on), and a launch
// is timed only when it runs on the device the events / the clock buffer belong to — an MSM that task_id % count or
// a shard routes to another device is simply not recorded (round 3 handed device A's buffer to a kernel on device
// B: a memory fault waiting for its first multi-GPU run).
struct ProfState {
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  std::atomic<int> mode{PROF_OFF};
  int src = PROF_EVENTS;    // where the launches recorded since the last enable were timed
  int every = 1, seen = 0;  // PROF_EVENTS: time one launch in `every`
  int count = 0;
  std::vector<hipEvent_t> e0, e1;  // grown in blocks of 512 as launches are recorded (no cap)
  bool created = false;
  int ev_device = -1;                   // PROF_EVENTS: the device the pool's events were created on
  unsigned long long* d_clk = nullptr;  // PROF_CLOCK: 4 words per launch (k_segreduce), zeroed by ozk_prof_enable
  int clk_cap = 0, clk_device = -1;
  double clk_khz = 0.0;  // measured against the host's steady clock
  // event pair for launch number `count`, or false when the pool cannot grow  (mu held)
  bool slot(hipEvent_t* a, hipEvent_t* b) {
    if ((size_t)count >= e0.size()) {
      const size_t want = e0.size() + 512;
      try {
        e0.reserve(want);
        e1.reserve(want);
      } catch (const std::exception&) {
        return false;
      }
      while (e0.size() < want) {
        hipEvent_t x = nullptr, y = nullptr;
        if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return false;
        e0.push_back(x);
        e1.push_back(y);
      }
    }
    *a = e0[count];
    *b = e1[count];
    return true;
  }
  // What the level-1 launch on the calling thread's current device carries: start / stop events (PROF_EVENTS) or
  // two words of the clock buffer (PROF_CLOCK), or nothing.  Claims the launch's index.
  void claim(hipEvent_t* a, hipEvent_t* b, unsigned long long** clk) {
    *a = *b = nullptr;
    *clk = nullptr;
    if (mode.load(std::memory_order_relaxed) == PROF_OFF) return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return;
    pthread_mutex_lock(&mu);
    const int m = mode.load(std::memory_order_relaxed);
    if (m == PROF_EVENTS && created && dev == ev_device) {
      if ((seen++ % every) == 0 && slot(a, b)) count++;
      else *a = *b = nullptr;
    } else if (m == PROF_CLOCK && d_clk && dev == clk_device && count < clk_cap) {
      *clk = d_clk + 4 * (size_t)count++;
    }
    pthread_mutex_unlock(&mu);
  }
};
inline ProfState g_prof;

// GLV needs 2n <= 2^24 sortable points (24-bit index in the packed coarse words).
constexpr int GLV_MAX_N = 1 << 23;

static MsmPlan make_plan(int n) {
  MsmPlan p;
  p.n_in = n;
  p.glv = env_int("OZK_MSM_GLV", 1) != 0 && n <= GLV_MAX_N;
  p.n = p.glv ? 2 * n : n;
  // Window size by cost model: bucket additions (points x windows) plus ~3.4 addition-equivalents
  // per bucket for the window sums (two Jacobian additions of 16-18 multiplications against 10 for a
  // mixed XYZZ addition).  A window count that leaves the top window nearly empty (c = 15 over 128
  // bits: 9 windows, the ninth 7 bits wide) also piles thousands of entries into a few buckets and
  // wakes the generic reduction levels — measured on G2 at 2^18: 8.1 ms with c = 15, the model's
  // c = 16 avoids it.
  const int sd_ok = p.glv && env_int("OZK_MSM_SIGNED", 1) != 0;
  const int bits = p.glv ? 128 : 256;
  int c = 4;
  double best = 1e300;
  for (int k = 4; k <= 16; k++) {
    const int Wk = (bits + k - 1) / k;
    const double W = (double)Wk;
    double cost = (double)p.n * W + 3.4 * W * (double)(1u << (k - sd_ok));
    if (p.glv) {
      // half scalars have ~127 bits: the top window's digits span top_bits, so its buckets hold 2^(k - top_bits)
      // times the average; once such a bucket is cut into more pieces than the run merge takes (RUN_MAX) the
      // generic levels wake up: +0.3 ms, about 4 M bucket additions' worth (2^17 with c = 13: 1.72 ms, the top
      // window's buckets in 13 pieces; 1.5 ms with c = 16)
      const int top_bits = 127 - (Wk - 1) * k;
      const long long per_top = top_bits >= 1 ? ((long long)p.n >> (top_bits - sd_ok > 0 ? top_bits - sd_ok : 0)) : (long long)p.n;
      long long l1 = ((long long)p.n >> (k - sd_ok)) * 5 / 16;
      if (l1 < 40) l1 = 40;
      if (per_top / l1 > 10) cost += 4.0e6;
    }
    if (cost < best) {
      best = cost;
      c = k;
    }
  }
  c = env_int("OZK_MSM_C", c);
  if (c < 1) c = 1;
  if (c > 16) c = 16;
  p.c = c;
  p.sd = p.glv && c >= 2 && env_int("OZK_MSM_SIGNED", 1) != 0;
  p.cb = c - p.sd;
  p.W = ((p.glv ? 128 : 256) + c - 1) / c;
  // entries per level-1 lane: 56 at 2^20 (64 entries per bucket: a bucket is cut into ~2 pieces, which
  // the run merge sums).  Buckets grow with n; at a fixed 40 a 2^23 MSM cut every bucket into 26 pieces
  // (more than RUN_MAX: everything fell through to the generic levels, +1.4 ms) and a 2^22 one into 13
  // (run merge 1.3 ms of 11.4).  Keep ~3-4 pieces per bucket.
  {
    const long long per_bucket = (long long)p.n >> p.cb;  // entries per bucket and window
    long long l1 = per_bucket * 5 / 16;
    // at least 40 entries per lane; 56 from 2^20 points on (fewer pieces for the run merge: pipelined 2^20 G1 +1.5 %,
    // G2 2^20 -0.15 ms, a 2^20-constraint proof -0.3 ms; smaller MSMs lose 30-100 us with it)
    // (round 4, with the 8-byte entry stream: 52 measures 0.8 % above 56 and 48 on the 100-step bench, twice on one box —
    // 48 / 52 / 56 / 60 / 64 / 72 -> 702 / 709 / 703 / 694 / 670 / 637 Mscalar-mul/s, profiles/r04_l1_chunk_sweep.txt)
    long long l1_min = env_int("OZK_MSM_L1_MIN", p.n >= (1 << 20) ? 52 : 40);
    // SMALL MSMs (round 4): below ~2^17 pairs the launch does not fill the chip — 818 lanes at 2^10 — and the level is
    // a chain of L1 dependent additions on lone waves (~5 us each): 200 us of a 0.94 ms MSM at n = 2^10
    // (profiles/r04_small_n_probe.txt).  Shorter chunks, down to 8 entries, as long as (a) the lanes still fit one
    // wave per SIMD (65536 lanes) and (b) the buckets of the TOP window — whose digits have only top_bits significant
    // bits, so that they hold 2^(c - top_bits) times the average — are cut into at most ~10 pieces: beyond RUN_MAX
    // pieces the generic levels wake up (+0.2-0.3 ms: 2^12 pairs with 8-entry chunks took 1.19 ms against 0.88 with 16,
    // same box, profiles/r04_small_n_ab.txt).  2^10: 0.97 -> 0.75 ms, 2^14: 1.06 -> 0.94, 2^15: 1.11 -> 0.97.
    if (p.glv) {
      const long long cap = (long long)p.n * p.W;
      long long fit = cap / 65536;
      const int top_bits = 127 - (p.W - 1) * p.c;
      const int tb = top_bits - p.sd;
      const long long per_top = (top_bits >= 1 && top_bits < p.c) ? ((long long)p.n >> (tb > 0 ? tb : 0)) : per_bucket;
      const long long pieces = (per_top + 9) / 10;
      if (fit < pieces) fit = pieces;
      if (fit < 8) fit = 8;
      if (fit < l1_min) l1_min = fit;
    }
    if (l1 < l1_min) l1 = l1_min;
    if (l1 > 1024) l1 = 1024;
    p.L1 = env_int("OZK_MSM_L1", (int)l1);
  }
  p.LK = env_int("OZK_MSM_LK", 16);
  if (p.L1 < 2) p.L1 = 2;
  if (p.LK < 4) p.LK = 4;
  int S = env_int("OZK_MSM_S", 4);
  int sg = ilog2((uint32_t)(S < 2 ? 2 : S));
  p.S = 1 << sg;
  return p;
}

// The plan for a curve.  G2's level-1 kernel keeps its accumulators in LDS (72 KiB per workgroup): two workgroups per
// CU, 512 on the chip, all doing the same amount of work — so the number of workgroups should be just under a multiple
// of 512.  With the default chunk of 56 entries a 2^20 MSM is 1171 workgroups = 2.29 rounds, the last of which runs
// on a third of the chip; 64 entries make it 1024 (two rounds: 7.34-7.39 -> 7.0-7.14 ms on two boxes), 128 make it 512
// (one round: the same); 2^19: 5.35 -> 4.86 ms, 2^21: 11.99 -> 11.13.
// The chunk is raised (never lowered) to the next such size.  G1 (three workgroups of 124 registers per CU, waves
// that speed up when their SIMD empties) measured best at the default and keeps it.  OZK_MSM_L1 overrides both.
// A LONE G1 MSM (the single-call entry points: nothing runs beside its level-1 kernel) gains from the same rule with
// its three workgroups per CU: 2^20 = 1171 workgroups of 56 entries = 1.52 rounds of 768; 86 entries make it 763 (one
// round): single MSM 2.52 -> 2.44 ms — and 85 entries (771 workgroups, three too many) 2.75 ms.  In the three-stage
// schedule a one-round launch leaves the tails and the next sort no workgroup slot until it ends (86: 628 against
// 690-709 Mscalar-mul/s), so the staged entry points keep the default chunk.  The single-call entry points mark their
// thread (LonePlan) for the duration of the call.
static thread_local int g_lone_plan = 0;
struct LonePlan {
  int prev;
  LonePlan() : prev(g_lone_plan) { g_lone_plan = 1; }
  ~LonePlan() { g_lone_plan = prev; }
};
template <class CV>
static MsmPlan plan_for(int n) {
  MsmPlan p = make_plan(n);
  if ((CV::LDS_ACC || g_lone_plan) && env_int("OZK_MSM_L1", 0) == 0 && env_int("OZK_MSM_L1_ROUNDS", 1) != 0) {
    static int cu_count = 0;   // (benign race: every thread computes the same value)
    if (cu_count == 0) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
          cus > 0)
        cu_count = cus;
      else
        cu_count = 256;
    }
    const int slots = (CV::LDS_ACC ? 2 : 3) * cu_count;   // workgroups resident at once
    const long long cap = (long long)p.n * p.W;           // sorted entries (upper bound: zero digits are skipped)
    const long long blocks = ((cap + p.L1 - 1) / p.L1 + 255) / 256;
    long long rounds = blocks / slots > 0 ? blocks / slots : 1;
    // at most two rounds (2^21: 128 entries per lane, 11.30 -> 11.13 ms against the four rounds of 64; one round of
    // 256 measures the same); OZK_MSM_L1_ROUNDS=0 switches the rule off, 1 asks for a single round
    const long long max_rounds = env_int("OZK_MSM_L1_ROUNDS", 2);
    if (rounds > max_rounds) rounds = max_rounds;
    long long l1 = (cap + 256LL * slots * rounds - 1) / (256LL * slots * rounds);
    if (l1 > p.L1 && l1 <= 1024) p.L1 = (int)l1;
  }
  return p;
}

struct MsmLayout {
  // all device pointers into the workspace
  u32* aff;
  u32 *hist, *C1, *P1, *blocksum, *total;  // total[0] = sorted entries, total[1] = live partial slots
  uint16_t* digits;
  uint8_t* neg_flags;  // GLV: sign of each half scalar
  u32* coarse;
  uint2* sent;   // sorted entries: (base index | sign << 31, bucket id), bucket-major
  BigBins* bigbins;
  u32* bigT;
  size_t big_items_max;
  int lo_bits, NH, nblk;
  size_t nC1;
  u32* buckets;   // tail region: bucket records, read by the first window-sum level
  u32* hist_t;    // tail region: copy of the bucket counts (the sorted set is reused by the next head)
  u32 *slot_bid[2], *slot_pts[2], *slot_bid2;
  u32 *wA[2], *wR[2];
  size_t bytes;
  size_t cap;   // n * W sorted entries at most
  size_t NB;    // W << c buckets
  size_t slots0, slots1;
  size_t m1;    // wsum elements per window after the first level
};

// Three regions, so that a caller can pipeline three stages of consecutive MSMs:
//   sorted set   : affine bases, bucket counts, sorted (index, bucket id) arrays, counters —
//                  the hand-off from the SORT stage to the ACCUMULATE stage (double-buffer it)
//   sort scratch : digits, per-block counts and their scan, coarse bins, big-bin work list
//   accum scratch: bucket records and partial slots
struct RegionBytes {
  size_t sorted, sort_ws, accum_ws;
};
template <class CV>
MsmLayout make_layout3(const MsmPlan& p, void* sorted, void* sort_ws, void* accum_ws, RegionBytes* rb,
                              const void* prepared = nullptr) {
  using IO = CurveIO<CV>;
  MsmLayout L;
  L.cap = (size_t)p.n * p.W;
  L.NB = (size_t)p.W << p.cb;
  const int lo_max = p.sd ? 7 : 8;  // signed: bit 7 of the packed coarse word carries the sign
  L.lo_bits = p.cb < lo_max ? p.cb : lo_max;
  L.NH = 1 << (p.cb - L.lo_bits);
  L.nblk = (p.n + SORT_CHUNK - 1) / SORT_CHUNK;
  L.nC1 = (size_t)p.W * L.NH * L.nblk;
  Bump a(sorted, ~(size_t)0);
  L.aff = a.take<u32>((size_t)p.n * IO::AFF_WORDS);
  if (prepared) L.aff = (u32*)prepared;  // affine records kept across MSMs (ozk_var_msm_prepare_dev)
  L.hist = a.take<u32>(L.NB);
  L.total = a.take<u32>(4);
  L.sent = a.take<uint2>(L.cap + 1);
  a.take<u32>(64);
  Bump b(sort_ws, ~(size_t)0);
  L.C1 = b.take<u32>(L.nC1);
  L.P1 = b.take<u32>(L.nC1);
  L.blocksum = b.take<u32>(L.nC1 / (SCAN_BLOCK * SCAN_ITEMS) + 2);
  L.digits = b.take<uint16_t>(L.cap);
  L.neg_flags = b.take<uint8_t>((size_t)p.n);
  L.coarse = b.take<u32>(L.cap);
  L.bigbins = b.take<BigBins>(1);
  L.big_items_max = L.cap / SORTBIG_CHUNK + SORTBIG_MAXBINS + 1;
  L.bigT = b.take<u32>(L.big_items_max * 256);
  b.take<u32>(64);
  Bump c(accum_ws, ~(size_t)0);
  const size_t T1 = (L.cap + p.L1 - 1) / p.L1;
  L.slots0 = 2 * T1;
  const size_t T2 = (L.slots0 + p.LK - 1) / p.LK;
  L.slots1 = 2 * T2;
  L.slot_bid[0] = c.take<u32>(L.slots0);
  L.slot_pts[0] = c.take<u32>(L.slots0 * IO::REC_WORDS);
  L.slot_bid2 = c.take<u32>(L.slots0);
  L.slot_bid[1] = c.take<u32>(L.slots1);
  L.slot_pts[1] = c.take<u32>(L.slots1 * IO::REC_WORDS);
  c.take<u32>(64);
  L.m1 = (((size_t)1 << p.cb) + p.S - 1) / p.S;
  if (rb) {
    rb->sorted = (a.off + 255) & ~(size_t)255;
    rb->sort_ws = (b.off + 255) & ~(size_t)255;
    rb->accum_ws = (c.off + 255) & ~(size_t)255;
  }
  L.bytes = a.off + b.off + c.off;
  return L;
}
template <class CV>
RegionBytes region_bytes(int n) {
  RegionBytes rb;
  make_layout3<CV>(plan_for<CV>(n), nullptr, nullptr, nullptr, &rb);
  return rb;
}

// The "tail" buffers (bucket records + counts, window-sum elements): the only state the
// latency-bound tail phase reads.
// They live outside the main workspace so that a caller can keep several MSMs in flight: the
// head phase of the next MSM may reuse the whole main workspace while this MSM's tail still
// runs on another stream.
template <class CV>
size_t tail_layout(const MsmPlan& p, MsmLayout& L, void* tail, size_t tail_bytes) {
  using IO = CurveIO<CV>;
  Bump b(tail, tail_bytes);
  L.NB = (size_t)p.W << p.cb;
  L.m1 = (((size_t)1 << p.cb) + p.S - 1) / p.S;
  L.buckets = b.take<u32>(L.NB * IO::REC_WORDS);
  L.hist_t = b.take<u32>(L.NB);
  for (int k = 0; k < 2; k++) {
    L.wA[k] = b.take<u32>((size_t)p.W * L.m1 * IO::JAC_WORDS);
    L.wR[k] = b.take<u32>((size_t)p.W * L.m1 * IO::JAC_WORDS);
  }
  b.take<u32>(64);
  return b.off;
}

// First window-sum level.  Fused form (default): a lane sums S = 4 buckets, its wave combines the 64
// lane results in registers -> W * 2^cb / 256 elements.  Everything after the bucket accumulation is
// multiplier-issue work spread over few waves, so WHERE it runs matters more than how deep it is
// (measured at 2^20 with unsigned digits, two MSMs in flight, Mscalar-mul/s / single-MSM ms):
//     unfused S=16, closing the head phase                      444-447 / 3.40-3.44
//     fused S=4, opening the tail phase, no issue priority      461-465 / 3.48-3.64
//     fused S=16 / unfused S=8 in the tail, with or without priority: 372-420 (their 512-1024 waves
//     sit on the same SIMDs as the next MSM's bucket accumulation and stretch it 1.35 -> 1.5-1.9 ms)
// Two shapes of the window sums, chosen per call (OZK_MSM_TAIL_MODE=0/1 forces one):
//   LATENCY (a lone MSM: ozk_var_msm_dev and the host-buffer entries): fused first level + wave-cooperative levels —
//     the fewest dependent additions (8 + 13, then 13 per 64x), most of them on mostly idle lanes;
//   THROUGHPUT (the staged entry points, whose callers keep several MSMs in flight): S buckets per lane serially
//     (2 additions per bucket), then serial levels of S elements per lane (3 additions per element) down to 64
//     elements per window, then one wave level — 3.0 additions per bucket instead of 5.25.  With the three-stage
//     schedule the chip is bound by the SUM of everybody's vector-ALU work (the accumulation kernel stretches by
//     whatever runs beside it), so the cheaper form wins although it is 4 launches and ~40 dependent additions
//     longer: 607 -> 664 Mscalar-mul/s at 2^20 (profiles/r03_schedule_experiments.txt).
enum { TAIL_LATENCY = 0, TAIL_THROUGHPUT = 1 };
struct TailShape {
  bool fused;
  int serial_above;
};
static TailShape tail_shape(int mode) {
  const int forced = env_int("OZK_MSM_TAIL_MODE", -1);
  if (forced == 0 || forced == 1) mode = forced;
  TailShape t;
  t.fused = env_int("OZK_MSM_WSUM_FUSED", mode == TAIL_THROUGHPUT ? 0 : 1) != 0;
  t.serial_above = env_int("OZK_MSM_TAIL_SERIAL_ABOVE", mode == TAIL_THROUGHPUT ? 64 : (1 << 30));
  return t;
}
// elements per window the first level leaves, and the g (log2 of buckets per element) they carry
static void first_level_shape(const MsmPlan& p, bool fused, int* m_out, int* g_out) {
  const int m_in = 1 << p.cb;
  const int nseg = (m_in + p.S - 1) / p.S;
  const int sg = ilog2((uint32_t)p.S);
  if (fused) {
    *m_out = (nseg + 63) / 64;
    *g_out = sg + 6;
  } else {
    *m_out = nseg;
    *g_out = sg;
  }
}
// First window-sum level, always the first kernel of the TAIL (it reads the bucket records of the tail region).
// Fused form: a lane sums S = 4 buckets, its wave combines the 64 lane results in registers -> W * 2^cb / 256
// elements.
template <class CV>
void launch_wsum0(const MsmPlan& p, const MsmLayout& L, hipStream_t st, int prio, bool fused) {
  const int TB = 256;
  const int m_in = 1 << p.cb;
  int m_out, g;
  first_level_shape(p, fused, &m_out, &g);
  const int tot = m_out * p.W;
  if (fused)
    hipLaunchKernelGGL((k_wsum_fused<CV>), dim3(tot), dim3(64), 0, st, L.buckets, L.hist_t, m_in, p.S,
                       ilog2((uint32_t)p.S), L.wA[0], L.wR[0], m_out, p.W, prio);
  else
    hipLaunchKernelGGL((k_wsum<CV, true>), dim3((tot + TB - 1) / TB), dim3(TB), 0, st, (const u32*)nullptr,
                       L.buckets, L.hist_t, m_in, p.S, 0, L.wA[0], L.wR[0], m_out, p.W, prio);
}

// SORT stage: bases -> affine Montgomery, digits, two-level counting sort.  Memory / LDS-bound;
// leaves the "sorted set".
template <class CV>
int var_msm_sort(const void* d_bases, const void* d_scalars, int n, void* sorted, size_t sorted_bytes,
                        void* sort_ws, size_t sort_ws_bytes, hipStream_t st, hipEvent_t order_ev = nullptr,
                        const void* prepared = nullptr) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  const MsmPlan p = plan_for<CV>(n);
  if (prepared && p.glv && !p.sd) return fail(OZK_E_INVALID, "prepared bases need the signed-digit plan");
  RegionBytes rb;
  const MsmLayout L = make_layout3<CV>(p, sorted, sort_ws, nullptr, &rb, prepared);
  if (rb.sorted > sorted_bytes || rb.sort_ws > sort_ws_bytes)
    return fail(OZK_E_INVALID, "sort buffers too small: need %zu + %zu bytes, got %zu + %zu", rb.sorted, rb.sort_ws,
                sorted_bytes, sort_ws_bytes);
  const int TB = 256;
  const u32* bases = (const u32*)d_bases;
  const u32* scalars = (const u32*)d_scalars;
  const int n_in = p.n_in;
  n = p.n;  // from here on: the points the pipeline sorts (2 * n_in with GLV)
  OZK_HIP(hipMemsetAsync(L.total, 0, 4 * sizeof(u32), st));
  if (p.glv) {
    hipLaunchKernelGGL(k_digits_glv, dim3((n_in + TB - 1) / TB), dim3(TB), 0, st, scalars, n_in, p.c, p.W, p.sd,
                       L.digits, L.neg_flags);
    if (!prepared)
      hipLaunchKernelGGL((k_convert_bases<CV>), dim3((n_in + TB - 1) / TB), dim3(TB), 0, st, bases, L.aff, n_in, 1,
                         p.sd ? (const uint8_t*)nullptr : (const uint8_t*)L.neg_flags);
  } else {
    if (!prepared)
      hipLaunchKernelGGL((k_convert_bases<CV>), dim3((n_in + TB - 1) / TB), dim3(TB), 0, st, bases, L.aff, n_in, 0,
                         (const uint8_t*)nullptr);
    hipLaunchKernelGGL(k_digits, dim3((n_in + TB - 1) / TB), dim3(TB), 0, st, scalars, n_in, p.c, p.W, L.digits);
  }
  if (n <= SORTS_MAX_N && p.cb <= SORTS_MAX_CB && env_int("OZK_MSM_SMALL_SORT", 1)) {
    // a small MSM: the whole sort in one launch (msm_var.cuh k_sort_small); L.total[0] was zeroed above
    hipLaunchKernelGGL(k_sort_small, dim3(p.W), dim3(SORTS_BLOCK), 0, st, L.digits, n, p.cb, p.sd, L.total, L.hist, L.sent);
    if (order_ev && env_int("OZK_MSM_ORDER", 1)) OZK_HIP(hipStreamWaitEvent(st, order_ev, 0));
    OZK_HIP(hipGetLastError());
    return OZK_OK;
  }
  // two-level counting sort by (window, digit): per-block LDS counts of the hi part, one global
  // exclusive scan, coarse scatter, then one block per coarse bin finishes by the lo part
  const size_t lds1 = (size_t)L.NH * sizeof(u32);
  hipLaunchKernelGGL(k_sort1_count, dim3(L.nblk, p.W), dim3(SORT_BLOCK), lds1, st, L.digits, n, L.lo_bits, p.sd,
                     L.NH, L.nblk, L.C1);
  const int items = SCAN_BLOCK * SCAN_ITEMS;
  const int nb = (int)((L.nC1 + items - 1) / items);
  hipLaunchKernelGGL(k_scan_blocksum, dim3(nb), dim3(SCAN_BLOCK), 0, st, L.C1, (int)L.nC1, L.blocksum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, L.blocksum, nb, L.total);
  hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_BLOCK), 0, st, L.C1, (int)L.nC1, L.blocksum, L.P1);
  const size_t lds_sc = ((size_t)2 * L.NH + 2 * SORT_CHUNK) * sizeof(u32);
  hipLaunchKernelGGL(k_sort1_scatter, dim3(L.nblk, p.W), dim3(SORT_BLOCK), lds_sc, st, L.digits, n, L.lo_bits,
                     p.sd, L.NH, L.nblk, L.P1, L.total, L.nC1, L.coarse);
  const int nbins = p.W * L.NH;
  const u32 sign_bit = p.sd ? 0x80u : 0u;
  const u32 big_thresh = (u32)(n / 64) > SORT_BIG ? (u32)(n / 64) : SORT_BIG;
  // at most cap / (big_thresh + 1) bins can exceed the threshold; the device work list holds SORTBIG_MAXBINS
  // (always enough for the library's own plans: 64 W <= 1024 for W <= 16; a forced tiny window at a large n is not)
  if (L.cap / ((size_t)big_thresh + 1) > (size_t)SORTBIG_MAXBINS)
    return fail(OZK_E_INVALID, "window plan c=%d (W=%d) at n=%d can produce more than %d oversized sort bins",
                p.c, p.W, p.n_in, SORTBIG_MAXBINS);
  hipLaunchKernelGGL(k_sort2, dim3(nbins), dim3(SORT2_BLOCK), 0, st, L.coarse, L.P1, L.total, p.cb, L.lo_bits, L.NH,
                     sign_bit, L.nblk, nbins, big_thresh, L.hist, L.sent);
  // Ordering hint for pipelined MSMs (see ozk_var_msm_tail_ordered_dev): everything up to here may
  // overlap the previous MSM's window-sum levels; the bucket accumulation that follows fills every
  // SIMD's register file, so the previous MSM's single-wave Horner kernel has to be resident first.
  if (order_ev && env_int("OZK_MSM_ORDER", 1)) OZK_HIP(hipStreamWaitEvent(st, order_ev, 0));
  // bins above the threshold (skewed digits), split over a fixed grid; no-ops otherwise
  hipLaunchKernelGGL(k_sortbig_list, dim3(1), dim3(256), 0, st, L.P1, L.total, L.nblk, nbins, big_thresh, L.bigbins);
  hipLaunchKernelGGL(k_sortbig_count, dim3(1024), dim3(SORT_BLOCK), 0, st, L.coarse, L.bigbins,
                     (1u << L.lo_bits) - 1u, L.bigT);
  hipLaunchKernelGGL(k_sortbig_scan, dim3(SORTBIG_MAXBINS), dim3(256), 0, st, L.bigbins, L.bigT, p.cb, L.lo_bits, L.NH,
                     L.hist);
  hipLaunchKernelGGL(k_sortbig_scatter, dim3(1024), dim3(SORT_BLOCK), 0, st, L.coarse, L.bigbins, L.bigT, p.cb,
                     L.lo_bits, sign_bit, L.NH, L.sent);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// ACCUMULATE stage: bucket accumulation, run merge, generic levels, first window-sum level.
// Vector-ALU-bound.  Reads the sorted set, leaves W * 2^c / S window-sum elements in the tail buffers.
// ACCUMULATE in two parts for callers that run them on different streams (ozk_var_msm_accum_part_dev): level 1 is
// the vector-ALU-bound kernel; the REST (run merge, the short generic levels, the copy of the bucket counts) is 0.1 ms
// of low-occupancy work that reads what level 1 wrote in the accumulate scratch and the sorted set.
enum { ACCUM_ALL = 0, ACCUM_LEVEL1 = 1, ACCUM_REST = 2 };
template <class CV>
int var_msm_accum(int n, void* sorted, size_t sorted_bytes, void* accum_ws, size_t accum_ws_bytes, void* tail,
                         size_t tail_bytes, hipStream_t st, const void* prepared = nullptr, int part = ACCUM_ALL) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  using CT = CV;  // (an out-of-line-multiplication variant for the tails measured 40 % slower)
  const MsmPlan p = plan_for<CV>(n);
  RegionBytes rb;
  MsmLayout L = make_layout3<CV>(p, sorted, nullptr, accum_ws, &rb, prepared);
  if (rb.sorted > sorted_bytes || rb.accum_ws > accum_ws_bytes)
    return fail(OZK_E_INVALID, "accumulate buffers too small: need %zu + %zu bytes, got %zu + %zu", rb.sorted,
                rb.accum_ws, sorted_bytes, accum_ws_bytes);
  const size_t tneed = tail_layout<CV>(p, L, tail, tail_bytes);
  if (tneed > tail_bytes) return fail(OZK_E_INVALID, "tail buffer too small: need %zu bytes, got %zu", tneed, tail_bytes);
  const int TB = 256;
  // level 1 over the sorted entries
  size_t lanes = (L.cap + p.L1 - 1) / p.L1;
  if (part != ACCUM_REST) {
  // optional timing of this launch (ProfState::claim): events that ride on the kernel's own dispatch packet
  // (hipExtLaunchKernelGGL start / stop events — separate hipEventRecord calls put two barrier packets around the
  // launch, which cost the three-stage schedule 8 % of its throughput: 637 -> 589 Mscalar-mul/s,
  // profiles/r03_schedule_experiments.txt), or two words the kernel's own waves stamp the device clock into
  hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
  unsigned long long* clk = nullptr;
  g_prof.claim(&prof_e0, &prof_e1, &clk);
  const bool prof = prof_e0 != nullptr;
  size_t acc_lds = CV::LDS_ACC ? (size_t)RunAccLds<CV>::LDS_WORDS * TB * sizeof(u32) : 0;  // 72 KiB for G2
  // G1: the kernel needs no LDS, but 4 blocks of 4 x 128 VGPRs fill a CU's register files completely, and a
  // kernel of another stream (a concurrent MSM's tail) dispatched later finds no wave slot until a block
  // retires (~0.7 ms).  Asking for just over a quarter of the CU's 160 KiB of LDS caps it at 3 blocks per CU —
  // one wave slot and 128 VGPRs per SIMD stay free.  Measured: the kernel itself is as fast with 3 waves per
  // SIMD as with 4 (1.404 vs 1.409 ms); two free-running MSM streams 416 -> 459 Mscalar-mul/s; the pipelined
  // bench without the launch-order hint 417 -> 476 (with it: unchanged, 480).
  // 40.25 KiB, not the 48 KiB of rounds 1-2: three blocks then leave 39.25 KiB to whatever runs beside them, which
  // is what the next MSM's sort kernels (k_sort1_scatter 34 KiB, k_sort2 38 KiB) need to be resident at all.
  if (!CV::LDS_ACC) acc_lds = (size_t)env_int("OZK_L1_LDS", 41216);
  auto launch_l1 = [&](auto kern) {
    if (acc_lds > 65536)
      OZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds));
    hipExtLaunchKernelGGL(kern, dim3((unsigned)((lanes + TB - 1) / TB)), dim3(TB), (uint32_t)acc_lds, st,
                          prof ? prof_e0 : (hipEvent_t) nullptr, prof ? prof_e1 : (hipEvent_t) nullptr, 0u,
                          (const u32*)nullptr, (const u32*)L.sent, (const u32*)L.aff, (const u32*)L.total, 0, p.L1,
                          L.buckets, L.slot_bid[0], L.slot_pts[0], (int)lanes, clk);
    return OZK_OK;
  };
  // the lazily carried mixed addition (ec.cuh xyzz_madd_lazy) where the curve has it; OZK_L1_LAZY=0: the carried form
  int rc_l1;
  if constexpr (CV::LAZY_MADD) {
    if (env_int("OZK_L1_LAZY", 1)) rc_l1 = launch_l1(k_segreduce<CV, true, true>);
    else rc_l1 = launch_l1(k_segreduce<CV, true, false>);
  } else {
    rc_l1 = launch_l1(k_segreduce<CV, true, false>);
  }
  if (rc_l1) return rc_l1;
  }
  if (part == ACCUM_LEVEL1) {
    OZK_HIP(hipGetLastError());
    return OZK_OK;
  }
  // run merge: completes every bucket cut into at most RUN_MAX pieces; counts the surviving slots
  size_t n_in = 2 * lanes;
  hipLaunchKernelGGL((k_runmerge<CT>), dim3((unsigned)((lanes + TB - 1) / TB)), dim3(TB), 0, st, L.slot_bid[0],
                     L.slot_pts[0], (int)n_in, L.buckets, L.slot_bid2, L.total + 1);
  // levels >= 2 over the surviving partial slots, ping-pong, until a single lane has seen everything
  // (they return at once when nothing survived)
  int cur = 0;
  bool first_generic = true;
  const int small_lanes = env_int("OZK_MSM_SMALL_LEVEL_LANES", 1024);
  while (true) {
    lanes = (n_in + p.LK - 1) / p.LK;
    if (!first_generic && (int)lanes <= small_lanes) {  // the remaining levels in one single-block launch
      hipLaunchKernelGGL((k_segreduce_small<CT>), dim3(1), dim3(TB), 0, st, L.total + 1, (int)n_in, p.LK, L.buckets,
                         L.slot_bid[cur], L.slot_pts[cur], L.slot_bid[cur ^ 1], L.slot_pts[cur ^ 1]);
      break;
    }
    hipLaunchKernelGGL((k_segreduce<CT, false>), dim3((unsigned)((lanes + TB - 1) / TB)), dim3(TB), 0, st,
                       first_generic ? L.slot_bid2 : L.slot_bid[cur], (const u32*)nullptr, L.slot_pts[cur],
                       L.total + 1, (int)n_in, p.LK, L.buckets, L.slot_bid[cur ^ 1], L.slot_pts[cur ^ 1], (int)lanes,
                       (unsigned long long*)nullptr);
    first_generic = false;
    if (lanes == 1) break;
    n_in = 2 * lanes;
    cur ^= 1;
  }
  // the tail needs the bucket counts after the next head has reused the sorted set
  OZK_HIP(hipMemcpyAsync(L.hist_t, L.hist, L.NB * sizeof(u32), hipMemcpyDeviceToDevice, st));
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// Head = SORT + ACCUMULATE on one stream, the three regions carved from one workspace.
template <class CV>
int var_msm_head(const void* d_bases, const void* d_scalars, int n, void* ws, size_t ws_bytes, void* tail,
                        size_t tail_bytes, hipStream_t st, hipEvent_t order_ev = nullptr,
                        const void* prepared = nullptr) {
  const RegionBytes rb = region_bytes<CV>(n);
  if (rb.sorted + rb.sort_ws + rb.accum_ws > ws_bytes)
    return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", rb.sorted + rb.sort_ws + rb.accum_ws,
                ws_bytes);
  uint8_t* w = (uint8_t*)ws;
  int rc = var_msm_sort<CV>(d_bases, d_scalars, n, w, rb.sorted, w + rb.sorted, rb.sort_ws, st, order_ev, prepared);
  if (rc) return rc;
  return var_msm_accum<CV>(n, w, rb.sorted, w + rb.sorted + rb.sort_ws, rb.accum_ws, tail, tail_bytes, st, prepared);
}

// Tail phase: the latency-bound remainder (wave-cooperative window-sum levels, Horner over the
// windows, affine normalisation).  Reads only the tail buffers; writes the wire-out result.
template <class CV>
int var_msm_tail(int n, void* tail, size_t tail_bytes, void* d_out, hipStream_t st,
                        hipEvent_t order_ev = nullptr, int mode = TAIL_LATENCY) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  using CT = CV;
  MsmPlan p = plan_for<CV>(n);
  MsmLayout L;
  L.m1 = (((size_t)1 << p.cb) + p.S - 1) / p.S;
  const size_t tneed = tail_layout<CV>(p, L, tail, tail_bytes);
  if (tneed > tail_bytes) return fail(OZK_E_INVALID, "tail buffer too small: need %zu bytes, got %zu", tneed, tail_bytes);
  const TailShape shape = tail_shape(mode);
  if (shape.fused && mode == TAIL_LATENCY) {
    // the fused first level of a LONE tail: 8 buckets per lane for G1 (2^20: single MSM 2.63-2.71 -> 2.51-2.53 ms on
    // one box; 16: 2.60), the plan's 4 for G2 (8: 7.43 against 7.38 ms).  The buffers are sized for the plan's S,
    // which is never larger, so the fewer elements of a wider first level always fit.
    // (small MSMs — fewer than 2^13 buckets per window — keep 4: the serial part of the level is what they wait for,
    // 2^10: 0.84 -> 0.80 ms)
    // (The fused level's output is 64 S times smaller than its input, so it fits the buffers — sized for the plan's
    // S — for ANY S; the plain form needs S >= the plan's.)
    int s_dflt = p.S;
    if (std::is_same<CV, G1Cfg>::value && p.cb >= 13) s_dflt = 8;   // (below 2^13 buckets per window the plan's 4 measures
                                                                      // as well or better: profiles/r04_small_n_ab.txt)
    const int s_lat = env_int("OZK_MSM_S_LAT", s_dflt);
    if (s_lat >= 2 && s_lat <= 64) p.S = 1 << ilog2((uint32_t)s_lat);
  }
  launch_wsum0<CV>(p, L, st, env_int("OZK_MSM_WSUM0_PRIO", 0), shape.fused);
  int m_in, g, k = 0;
  first_level_shape(p, shape.fused, &m_in, &g);
  const int TB = 256;
  const int sg = ilog2((uint32_t)p.S);
  // Serial S-per-lane levels first in THROUGHPUT mode (3 additions per element instead of the wave form's 13, but
  // 12 dependent additions deep per level); none in LATENCY mode (tail_shape above).
  const int serial_above = shape.serial_above < 1 ? 1 : shape.serial_above;
  while (m_in > serial_above) {
    const int m_out = (m_in + p.S - 1) / p.S;
    const int tot = m_out * p.W;
    hipLaunchKernelGGL((k_wsum<CT, false>), dim3((tot + TB - 1) / TB), dim3(TB), 0, st, L.wA[k], L.wR[k],
                       (const u32*)nullptr, m_in, p.S, g, L.wA[k ^ 1], L.wR[k ^ 1], m_out, p.W, 1);
    m_in = m_out;
    g += sg;
    k ^= 1;
  }
  // wave-cooperative levels until a handful of elements per window is left; k_finalize finishes those
  int fin_max = env_int("OZK_MSM_FIN_MAX", 4);
  if (fin_max < 1) fin_max = 1;
  if (fin_max > 16) fin_max = 16;
  // Launch-order hint for pipelined MSMs: the event fires here, after the FIRST window-sum level, so that the next
  // accumulation starts as soon as its sort is done and the wave level behind this point is already resident.
  // Round 1 recorded it after the wave level (OZK_MSM_ORDER_EARLY=0), when the accumulation kernel filled every
  // register file; since that kernel is capped at 3 blocks per CU the late event only delays the next
  // accumulation by ~0.18 ms per MSM: measured 538 (late) against 576 Mscalar-mul/s (early, or no hint at all).
  // Also measured and not shipped (profiles/r02_schedule_experiments.txt): the merge of the level-1 pieces moved
  // from the head to the tail stream (524-550), a three-stage schedule sort | accumulate | tail (562-599).
  const bool order_early = env_int("OZK_MSM_ORDER_EARLY", 1) != 0;
  if (order_ev && order_early) OZK_HIP(hipEventRecord(order_ev, st));
  while (m_in > fin_max) {
    const int m_out = (m_in + 63) / 64;
    const int tot = m_out * p.W;
    hipLaunchKernelGGL((k_wsum_wave<CT>), dim3(tot), dim3(64), 0, st, L.wA[k], L.wR[k], m_in, g, L.wA[k ^ 1],
                       L.wR[k ^ 1], m_out, p.W);
    m_in = m_out;
    g += 6;
    k ^= 1;
  }
  if (order_ev && !order_early) OZK_HIP(hipEventRecord(order_ev, st));  // the multi-wave levels are done
  // one wave: Horner over the windows is serial
  hipLaunchKernelGGL((k_finalize<CT>), dim3(1), dim3(64), 0, st, L.wA[k], L.wR[k], m_in, g, p.W, p.c, p.sd,
                     (u32*)d_out);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

template <class CV>
size_t var_msm_tail_bytes(int n) {
  const MsmPlan p = plan_for<CV>(n);
  MsmLayout L;
  L.m1 = (((size_t)1 << p.cb) + p.S - 1) / p.S;
  return tail_layout<CV>(p, L, nullptr, 0);
}

// head + tail on one stream, the tail buffers carved from the end of the workspace
template <class CV>
int var_msm_dev(const void* d_bases, const void* d_scalars, int n, void* d_out, void* ws,
                       size_t ws_bytes, hipStream_t st, const void* prepared = nullptr) {
  LonePlan lone;   // (before any size is computed: the lone plan's regions are never larger than the staged plan's)
  const RegionBytes rb0 = region_bytes<CV>(n);
  const size_t main_bytes = rb0.sorted + rb0.sort_ws + rb0.accum_ws;
  const size_t tb = var_msm_tail_bytes<CV>(n);
  if (main_bytes + tb > ws_bytes)
    return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", main_bytes + tb, ws_bytes);
  uint8_t* tail = (uint8_t*)ws + main_bytes;
  int rc = var_msm_head<CV>(d_bases, d_scalars, n, ws, main_bytes, tail, tb, st, nullptr, prepared);
  if (rc) return rc;
  return var_msm_tail<CV>(n, tail, tb, d_out, st);
}

template <class CV>
size_t var_msm_head_ws_bytes(int n) {
  const RegionBytes rb = region_bytes<CV>(n);
  return rb.sorted + rb.sort_ws + rb.accum_ws;
}
template <class CV>
size_t var_msm_ws_bytes(int n) {
  return var_msm_head_ws_bytes<CV>(n) + var_msm_tail_bytes<CV>(n);
}

// host-buffer variant (what the JNI native calls): staged upload, run, download, all on a cached context
// (host_ctx.h) — no allocation, stream creation or pageable copy per call.  Re-entrant: concurrent callers
// (Spark task threads in the reference, SURVEY.md §8b "Threading") each take their own context.
//
// A large call is cut into index-range SLICES that share one set of windows and ONE tail: slice s is uploaded on
// the context's copy stream and its sort + bucket accumulation (into its own bucket array) are queued behind that
// upload before the host thread starts staging slice s + 1, so all but the last slice's throughput-bound work
// runs under the PCIe transfer (128 MiB at 2^20 G1: ~2.9 ms of upload against ~1.9 ms of such work); then the
// bucket arrays are added (k_bucket_combine) and the latency-bound tail runs once.  (Complete MSMs per slice,
// summed at the end, measured SLOWER than no slicing beyond two slices: every slice pays the ~1.1 ms tail, and
// they queue — profiles/r02_host_path.txt.)  All slices use the plan of the slice size; the last one is padded
// with zero scalars / infinity bases.  Within a slice the scalars go up first.
// Slice count: slices of at least 2^18 pairs (below that the head of an MSM stops shrinking with its size: a
// 2^17 MSM takes as long as a 2^18 one, tools/host_slices_probe.py), at most OZK_HOST_SLICES (8).  Measured
// ozk_var_msm_host at 2^20 G1: 6.0 ms unsliced, 4.8 ms in 4 slices; 2^22: 19.7 -> 12.8 ms in 8; G2 2^20: 13.6 -> 11.5.
inline int host_slices(int n) {
  int kmax = env_int("OZK_HOST_SLICES", 8);
  if (kmax > MAX_SLICES) kmax = MAX_SLICES;
  int min_log = env_int("OZK_HOST_SLICE_MIN_LOG", 18);  // (tests lower it to slice small inputs)
  if (min_log < 4) min_log = 4;
  if (min_log > 30) min_log = 30;
  int k = n >> min_log;
  if (k > kmax) k = kmax;
  return k < 1 ? 1 : k;
}

// Size of the first K - 1 slices.  The LAST slice is the one whose sort + accumulation cannot hide under an upload,
// so it gets half an average slice (n / 2K pairs) and the others share the rest (2^20: G1 unchanged at 4.8 ms, G2
// 11.2 -> 11.0 ms, the double MSM 13.95 -> 13.5 ms).
// (OZK_HOST_SLICE_TAPER=0: K equal slices.)
inline int host_slice_per(int n, int K) {
  if (K < 2) return n;
  if (env_int("OZK_HOST_SLICE_TAPER", 1) == 0) return (n + K - 1) / K;
  const long long rest = (long long)n - (long long)n / (2 * K);
  return (int)((rest + K - 2) / (K - 1));
}
static bool plans_agree(int n1, int n2) {
  const MsmPlan a = make_plan(n1), b = make_plan(n2);
  return a.c == b.c && a.cb == b.cb && a.W == b.W && a.sd == b.sd && a.glv == b.glv && a.S == b.S;
}

// The sliced part of a host MSM, on a context the caller holds: K - 1 slices of `per` pairs and a last one with the
// rest (run with its own size when its plan has the same windows as the others', else padded to `per`).
// d_bases / d_sc hold K * per records; `scalars` == nullptr: the caller has already queued the upload of ALL scalars
// (and the zero padding) on `up`, so only the bases go up here.  Slice events ev[0 .. K).  Leaves the result
// (wire-out) in d_out on `st`.
template <class CV>
size_t host_sliced_ws_bytes(int K, int per) {
  return pad256(var_msm_head_ws_bytes<CV>(per)) + (size_t)K * pad256(var_msm_tail_bytes<CV>(per));
}
template <class CV>
int host_sliced_msm(HostCtx* c, const uint8_t* bases, const uint8_t* scalars, int n, int K, int per, uint8_t* d_bases,
                    uint8_t* d_sc, uint8_t* d_ws, uint8_t* d_out, hipEvent_t* ev, hipStream_t st, hipStream_t up) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  using IO = CurveIO<CV>;
  const size_t base_rec = (size_t)IO::WIRE_JAC_WORDS * 4;
  const size_t padded = (size_t)K * per;
  const size_t main_bytes = pad256(var_msm_head_ws_bytes<CV>(per));
  const size_t tb = pad256(var_msm_tail_bytes<CV>(per));
  uint8_t* d_tails = d_ws + main_bytes;
  int rc;
  const long long last_ns = (long long)n - (long long)(K - 1) * per;
  if (last_ns < 1) return fail(OZK_E_INTERNAL, "slice plan: %d slices of %d pairs exceed n = %d", K, per, n);
  const bool exact_last = last_ns < per && plans_agree(per, (int)last_ns);
  if (padded > (size_t)n && !exact_last) {
    if (scalars) OZK_HIP(hipMemsetAsync(d_sc + (size_t)n * 32, 0, (padded - n) * 32, up));
    OZK_HIP(hipMemsetAsync(d_bases + (size_t)n * base_rec, 0, (padded - n) * base_rec, up));  // Z = 0: infinity
  }
  const MsmPlan p = plan_for<CV>(per);
  SliceBuckets sb = {};
  for (int s = 0; s < K; s++) {
    const size_t lo = (size_t)s * per;
    const size_t ns = ((size_t)n - lo < (size_t)per) ? ((size_t)n - lo) : (size_t)per;
    if (scalars && (rc = staged_h2d(c, d_sc + lo * 32, scalars + lo * 32, ns * 32, up))) return rc;
    if ((rc = staged_h2d(c, d_bases + lo * base_rec, bases + lo * base_rec, ns * base_rec, up))) return rc;
    OZK_HIP(hipEventRecord(ev[s], up));
    OZK_HIP(hipStreamWaitEvent(st, ev[s], 0));
    uint8_t* tail = d_tails + (size_t)s * tb;
    const int n_head = (s == K - 1 && exact_last) ? (int)last_ns : per;
    if ((rc = var_msm_head<CV>(d_bases + lo * base_rec, d_sc + lo * 32, n_head, d_ws, main_bytes, tail, tb, st))) return rc;
    MsmLayout L;
    tail_layout<CV>(p, L, tail, tb);
    sb.buckets[s] = L.buckets;
    sb.hist[s] = L.hist_t;
  }
  const size_t NB = (size_t)p.W << p.cb;
  hipLaunchKernelGGL((k_bucket_combine<CV>), dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, st, sb, K,
                     (u32*)sb.buckets[0], (u32*)sb.hist[0], NB);
  OZK_HIP(hipGetLastError());
  return var_msm_tail<CV>(per, d_tails, tb, d_out, st);
}

// where a host entry point's result goes: the caller's host memory (through the context's pinned result buffer), or —
// `d_result` != nullptr, the sharded entry's RCCL form — a device buffer on the same device (the partial then never
// visits the host: it is all-gathered from there)
inline int host_result(HostCtx* c, uint8_t* out, uint8_t* d_result, const uint8_t* d_out, size_t bytes, hipStream_t st) {
  if (!d_result) return staged_d2h(c, out, d_out, bytes, st);
  OZK_HIP(hipMemcpyAsync(d_result, d_out, bytes, hipMemcpyDeviceToDevice, st));
  StatTimer tm(host_call_stats().sync_ms);
  OZK_HIP(hipStreamSynchronize(st));
  return OZK_OK;
}

template <class CV>
int var_msm_host(const uint8_t* bases, const uint8_t* scalars, int n, int task_id, uint8_t* out,
                 uint8_t* d_result = nullptr) {
  using IO = CurveIO<CV>;
  CtxGuard g;
  int rc = ctx_acquire(task_id, &g.c);
  if (rc) return rc;
  HostCtx* c = g.c;
  const size_t base_rec = (size_t)IO::WIRE_JAC_WORDS * 4;
  const size_t out_bytes = (size_t)IO::WIRE_JAC_WORDS * 8;
  const int K = host_slices(n);
  if (K == 1) {
    const size_t base_bytes = (size_t)n * base_rec, sc_bytes = (size_t)n * 32;
    const size_t ws_bytes = var_msm_ws_bytes<CV>(n);
    if ((rc = ctx_reserve(c, pad256(base_bytes) + pad256(sc_bytes) + 1024 + ws_bytes + 1024))) return rc;
    uint8_t* d_bases = c->arena;
    uint8_t* d_sc = d_bases + pad256(base_bytes);
    uint8_t* d_out = d_sc + pad256(sc_bytes);
    uint8_t* d_ws = d_out + 1024;
    hipStream_t st = c->st[0];
    if ((rc = staged_h2d(c, d_sc, scalars, sc_bytes, st))) return rc;
    if ((rc = staged_h2d(c, d_bases, bases, base_bytes, st))) return rc;
    if ((rc = var_msm_dev<CV>(d_bases, d_sc, n, d_out, d_ws, ws_bytes, st))) return rc;
    return host_result(c, out, d_result, d_out, out_bytes, st);
  }
  const int per = host_slice_per(n, K);          // workspace and tails are laid out for `per` pairs
  const size_t padded = (size_t)K * per;
  if ((rc = ctx_reserve(c, pad256(padded * base_rec) + pad256(padded * 32) + 1024 + host_sliced_ws_bytes<CV>(K, per) + 1024)))
    return rc;
  uint8_t* d_bases = c->arena;
  uint8_t* d_sc = d_bases + pad256(padded * base_rec);
  uint8_t* d_out = d_sc + pad256(padded * 32);
  uint8_t* d_ws = d_out + 1024;
  // uploads on the context's copy stream: they must not wait behind kernels
  if ((rc = host_sliced_msm<CV>(c, bases, scalars, n, K, per, d_bases, d_sc, d_ws, d_out, c->slice_ev, c->st[0], c->st[2])))
    return rc;
  return host_result(c, out, d_result, d_out, out_bytes, c->st[0]);
}

// ---- prepared bases: the affine Montgomery records (GLV: both halves) of a base
// array, kept in HBM across MSMs.  A Groth16 proving key is fixed; the reference re-marshals and
// re-uploads it for every proof (VariableBaseMSM.java:224-227).
template <class CV>
size_t prepared_bytes(int n) {
  const MsmPlan p = plan_for<CV>(n);
  return (((size_t)p.n * CurveIO<CV>::AFF_WORDS * sizeof(u32)) + 255) & ~(size_t)255;
}
template <class CV>
int var_msm_prepare(const void* d_bases, int n, void* d_prepared, size_t bytes, hipStream_t st) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  const MsmPlan p = make_plan(n);
  if (p.glv && !p.sd) return fail(OZK_E_INVALID, "prepared bases need the signed-digit plan");
  if (bytes < prepared_bytes<CV>(n)) return fail(OZK_E_INVALID, "prepared buffer too small");
  hipLaunchKernelGGL((k_convert_bases<CV>), dim3((n + 255) / 256), dim3(256), 0, st, (const u32*)d_bases,
                     (u32*)d_prepared, n, p.glv, (const uint8_t*)nullptr);
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

// Host-side handle: prepared bases + everything an MSM over them needs, allocated once.
constexpr uint32_t BASES_MAGIC = 0x4f5a4b42u;  // "OZKB": set by create, cleared by destroy
struct BasesHandle {
  uint32_t magic;
  int refs;  // callers holding the handle (msm_var.hip, handle table)
  int device, n, type;
  hipStream_t st;
  uint8_t *d_prepared, *d_scalars, *d_out, *d_ws;
  uint8_t* h_out;  // pinned host, 1 KiB: the result lands here (host_ctx.h, small_d2h_begin)
  size_t ws_bytes;
  pthread_mutex_t mu;
};

template <class CV>
int bases_create(const uint8_t* bases, int n, int type, int task_id, BasesHandle** out) {
  using IO = CurveIO<CV>;
  int rc = select_device(task_id);
  if (rc) return rc;
  BasesHandle* h = (BasesHandle*)calloc(1, sizeof(BasesHandle));
  if (!h) return fail(OZK_E_NOMEM, "out of host memory");
  hipGetDevice(&h->device);
  h->n = n;
  h->type = type == OZK_G1 ? OZK_G1 : OZK_G2;
  pthread_mutex_init(&h->mu, nullptr);
  const size_t wire = (size_t)n * IO::WIRE_JAC_WORDS * 4, pb = prepared_bytes<CV>(n);
  h->ws_bytes = var_msm_ws_bytes<CV>(n);
  hipError_t e = hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking);
  uint8_t* d_wire = nullptr;
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_prepared, pb);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_scalars, (size_t)n * 32 + 256);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_out, 1024);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_out, 1024, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_ws, h->ws_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&d_wire, wire);
  rc = OZK_OK;
  if (e == hipSuccess) {
    CtxGuard g;  // only for its pinned staging ring
    rc = ctx_acquire(task_id, &g.c);
    if (!rc) rc = staged_h2d(g.c, d_wire, bases, wire, h->st);
    if (!rc) e = hipStreamSynchronize(h->st);
  }
  if (e == hipSuccess && !rc) rc = var_msm_prepare<CV>(d_wire, n, h->d_prepared, pb, h->st);
  if (e == hipSuccess && !rc) e = hipStreamSynchronize(h->st);
  if (d_wire) hipFree(d_wire);
  if (e != hipSuccess || rc) {
    if (h->d_prepared) hipFree(h->d_prepared);
    if (h->d_scalars) hipFree(h->d_scalars);
    if (h->d_out) hipFree(h->d_out);
    if (h->h_out) hipHostFree(h->h_out);
    if (h->d_ws) hipFree(h->d_ws);
    if (h->st) hipStreamDestroy(h->st);
    free(h);
    if (rc) return rc;
    return fail(OZK_E_NOMEM, "HIP failure while preparing bases: %s", hipGetErrorString(e));
  }
  h->magic = BASES_MAGIC;
  *out = h;
  return OZK_OK;
}

template <class CV>
int bases_msm(BasesHandle* h, const uint8_t* scalars, uint8_t* out) {
  const size_t out_bytes = (size_t)CurveIO<CV>::WIRE_JAC_WORDS * 8;
  pthread_mutex_lock(&h->mu);
  if (h->magic != BASES_MAGIC) {  // released while this call waited for the lock
    pthread_mutex_unlock(&h->mu);
    return fail(OZK_E_INVALID, "bases handle was released");
  }
  int rc = OZK_OK;
  hipError_t e = hipSetDevice(h->device);
  do {
    if (e != hipSuccess) break;
    {
      CtxGuard g;  // pinned staging ring for the scalars
      rc = ctx_acquire(h->device, &g.c);
      if (!rc) rc = staged_h2d(g.c, h->d_scalars, scalars, (size_t)h->n * 32, h->st);
      if (!rc) e = hipStreamSynchronize(h->st);   // the ring goes back to the pool with nothing in flight
    }
    if (rc || e != hipSuccess) break;
    rc = var_msm_dev<CV>(nullptr, h->d_scalars, h->n, h->d_out, h->d_ws, h->ws_bytes, h->st, h->d_prepared);
    if (rc) break;
    if ((e = hipMemcpyAsync(h->h_out, h->d_out, out_bytes, hipMemcpyDeviceToHost, h->st)) != hipSuccess) break;
    e = hipStreamSynchronize(h->st);
    if (e == hipSuccess) memcpy(out, h->h_out, out_bytes);
  } while (0);
  pthread_mutex_unlock(&h->mu);
  if (rc) return rc;
  if (e != hipSuccess) return fail(OZK_E_NO_DEVICE, "HIP failure in MSM over prepared bases: %s", hipGetErrorString(e));
  return OZK_OK;
}

}  // namespace ozk

// the G2 instantiations live in msm_var_g2.hip
#define OZK_G2_DRIVER_INSTANCES(PREFIX)                                                                              \
  PREFIX template int ozk::var_msm_sort<ozk::G2Cfg>(const void*, const void*, int, void*, size_t, void*, size_t,    \
                                                    hipStream_t, hipEvent_t, const void*);                          \
  PREFIX template int ozk::var_msm_accum<ozk::G2Cfg>(int, void*, size_t, void*, size_t, void*, size_t, hipStream_t,  \
                                                     const void*, int);                                              \
  PREFIX template int ozk::var_msm_head<ozk::G2Cfg>(const void*, const void*, int, void*, size_t, void*, size_t,     \
                                                    hipStream_t, hipEvent_t, const void*);                          \
  PREFIX template int ozk::var_msm_tail<ozk::G2Cfg>(int, void*, size_t, void*, hipStream_t, hipEvent_t, int);        \
  PREFIX template int ozk::var_msm_dev<ozk::G2Cfg>(const void*, const void*, int, void*, void*, size_t, hipStream_t,  \
                                                   const void*);                                                     \
  PREFIX template int ozk::var_msm_host<ozk::G2Cfg>(const uint8_t*, const uint8_t*, int, int, uint8_t*, uint8_t*);             \
  PREFIX template size_t ozk::host_sliced_ws_bytes<ozk::G2Cfg>(int, int);                                            \
  PREFIX template int ozk::host_sliced_msm<ozk::G2Cfg>(ozk::HostCtx*, const uint8_t*, const uint8_t*, int, int, int, \
                                                       uint8_t*, uint8_t*, uint8_t*, uint8_t*, hipEvent_t*,         \
                                                       hipStream_t, hipStream_t);                                   \
  PREFIX template int ozk::var_msm_prepare<ozk::G2Cfg>(const void*, int, void*, size_t, hipStream_t);                \
  PREFIX template int ozk::bases_create<ozk::G2Cfg>(const uint8_t*, int, int, int, ozk::BasesHandle**);              \
  PREFIX template int ozk::bases_msm<ozk::G2Cfg>(ozk::BasesHandle*, const uint8_t*, uint8_t*);
