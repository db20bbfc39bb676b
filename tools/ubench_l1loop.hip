// Bisecting the level-1 loop (msm_var.cuh segreduce_lane<G1Cfg, true, true>): the formula alone runs at 0.95 ms per
// 16.78 M additions (tools/ubench_madd.hip), the kernel at 1.18-1.25 ms with only ~6 % more instructions.  This file
// rebuilds the loop piece by piece on synthetic sorted arrays (one round of the chip, 86 entries per lane):
//   A  formula on a register-resident q                         (= ubench_madd mode 4)
//   B  + the gather: index word -> 64-byte record, requested one entry ahead, decoded when consumed; indices
//        random over a 128 MiB table
//   C  B with sequential indices (cache-friendly)
//   G  B with the prefetched record in LDS instead of registers (gfx950 LDS-direct 16-byte loads)
//   D  B + the run logic (bucket id per entry, run boundaries every ~64 entries, finished runs stored as 160-byte records)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOZK_WITH_G2 tools/ubench_l1loop.hip -o tools/ubench_l1loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../octopuszk_amd/csrc/msm_var.cuh"
using namespace ozk;

template <int MODE>
__global__ void __launch_bounds__(256) k_loop(const u32* __restrict__ idx, const u32* __restrict__ bid, const u32* __restrict__ pts,
                                              u32* __restrict__ buckets, u32* __restrict__ out, int L,
                                              const uint2* __restrict__ ent = nullptr) {
  extern __shared__ u32 dummy[];
  using CV = G1Cfg;
  using IO = CurveIO<CV>;
  using Acc = RunAcc<CV, true>;
  using EA = typename CV::EA;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32* idx_c = idx + (size_t)t * L;
  const u32* bid_c = bid + (size_t)t * L;
  Acc acc;
  const u32 n_e = (u32)L;
  u32 v_cur = idx_c[0];
  u32 v_next = idx_c[1];
  typename Acc::Raw r = Acc::load_raw(pts, v_cur);
  // MODE 6: the prefetched record goes straight into LDS (global_load_lds_dwordx4: gfx950's 16-byte LDS-direct load,
  // no VGPRs in between) — two 4 KiB buffers per wave, lane l's q-th 16 bytes at [q][l] — and is read back with
  // ds_read_b128 when the entry is consumed
  u32* wbuf = dummy + (threadIdx.x >> 6) * 2048;
  const u32 lane = threadIdx.x & 63;
  auto lds_issue = [&](u32 v, u32 b) {
    const u32* rec = pts + (size_t)(v & ~SIDX_NEG) * 16;
#pragma unroll
    for (int q = 0; q < 4; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + 4 * q),
                                       (__attribute__((address_space(3))) void*)(wbuf + b * 1024 + q * 256), 16, 0, 0);
  };
  auto lds_take = [&](u32 b) {
    typename Acc::Raw x;
#pragma unroll
    for (int q = 0; q < 4; q++) x.w[q] = *reinterpret_cast<const uint4*>(wbuf + b * 1024 + q * 256 + lane * 4);
    return x;
  };
  if constexpr (MODE == 6) lds_issue(v_cur, 0);
  // MODE 7: the table holds UNPACKED records — nine 29-bit limbs per coordinate, 80 bytes, five aligned 16-byte loads —
  // so an entry needs no unpacking (40 instructions per addition) for 25 % more gather bytes
  struct Raw80 { uint4 w[5]; };
  auto load80 = [&](u32 v) {
    Raw80 x;
    const uint4* p = reinterpret_cast<const uint4*>(pts + (size_t)(v & ~SIDX_NEG) * 20);
#pragma unroll
    for (int q = 0; q < 5; q++) x.w[q] = p[q];
    return x;
  };
  auto dec80 = [&](const Raw80& x) {
    const u32 w[20] = {x.w[0].x, x.w[0].y, x.w[0].z, x.w[0].w, x.w[1].x, x.w[1].y, x.w[1].z, x.w[1].w, x.w[2].x, x.w[2].y,
                       x.w[2].z, x.w[2].w, x.w[3].x, x.w[3].y, x.w[3].z, x.w[3].w, x.w[4].x, x.w[4].y, x.w[4].z, x.w[4].w};
    Aff<EA> a;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      a.x.l[j] = w[j];
      a.y.l[j] = w[9 + j];
    }
    return a;
  };
  Raw80 r80;
  if constexpr (MODE == 7) r80 = load80(v_cur);
  u32 b_cur = bid_c[0];
  // MODE 8: entries (index, bucket id) interleaved: ONE 8-byte load two entries ahead instead of two 4-byte loads
  const size_t n_lanes_all = (size_t)gridDim.x * blockDim.x;
  const uint2* ent_c = (MODE == 9) ? ent + t : ent + (size_t)t * L;
  const size_t estride = (MODE == 9) ? n_lanes_all : 1;
  uint2 e_next = make_uint2(0, 0);
  if constexpr (MODE == 8 || MODE == 9 || MODE == 10) {
    const uint2 e0 = ent_c[0];
    e_next = ent_c[estride];
    v_cur = e0.x;
    b_cur = e0.y;
    v_next = e_next.x;
    r = Acc::load_raw(pts, v_cur);
  }
  // MODE 10: the record through a buffer resource: four buffer_load_dwordx4 off ONE 32-bit offset register (immediate
  // offsets 0 / 16 / 32 / 48) instead of the five odd-sized global loads and the 64-bit address arithmetic hipcc makes
  typedef int v4i __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pts, 0, (int)0x7fffffff, 0x00020000);
  auto buf_load = [&](u32 v) {
    typename Acc::Raw x;
    const int off = (int)((v & ~SIDX_NEG) << 6);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const v4i w4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16 * q, 0, 0);
      x.w[q] = make_uint4((u32)w4.x, (u32)w4.y, (u32)w4.z, (u32)w4.w);
    }
    return x;
  };
  if constexpr (MODE == 10) r = buf_load(v_cur);
  u32 cur = BID_NONE;
  Aff<EA> qfix = Acc::decode_unsigned(r);
  u32 sink = 0;
  if constexpr (MODE == 4 || MODE == 5) {   // start away from +-qfix so that no doubling / cancellation ever happens
    typename Acc::Raw r2 = Acc::load_raw(pts, v_next ^ 1u);
    acc.start_signed(Acc::decode_unsigned(r2), false);
    cur = b_cur;
  }
  for (u32 k = 0; k < n_e; k++) {
    const u32 b = b_cur;
    const bool negate = (v_cur & SIDX_NEG) != 0;
    Aff<EA> q;
    if constexpr (MODE == 0 || MODE == 4) q = qfix;
    else if constexpr (MODE == 5) {   // a different q every iteration without any load: limbs of qfix rotated by k
      q = qfix;
#pragma unroll
      for (int j = 0; j < 8; j++) q.x.l[j] = (qfix.x.l[j] ^ (k << j)) & 0x1fffffffu;
    } else q = Acc::decode_unsigned(r);
    if constexpr (MODE == 4) sink ^= r.w[0].x ^ r.w[1].y ^ r.w[2].z ^ r.w[3].w;
    const u32 k1 = (k + 1 < n_e) ? k + 1 : n_e - 1;
    const u32 k2 = (k + 2 < n_e) ? k + 2 : n_e - 1;
    if constexpr (MODE == 8 || MODE == 9 || MODE == 10) {
      const uint2 en = e_next;            // entry k + 1: requested one iteration ago
      if constexpr (MODE == 10) r = buf_load(en.x);
      else r = Acc::load_raw(pts, en.x);
      b_cur = en.y;
      v_next = en.x;
      e_next = ent_c[(size_t)k2 * estride];   // entry k + 2: ONE 8-byte load
    } else if constexpr (MODE == 7) r80 = load80(v_next);
    else if constexpr (MODE == 6) lds_issue(v_next, (k + 1u) & 1u);
    else if constexpr (MODE != 0 && MODE != 5 && MODE != 8 && MODE != 9 && MODE != 10) r = Acc::load_raw(pts, v_next);
    if constexpr (MODE == 3) b_cur = bid_c[k1];
    v_cur = v_next;
    if constexpr (MODE == 8 || MODE == 9 || MODE == 10) {
    } else if constexpr (MODE != 0 && MODE != 5) v_next = idx_c[k2];
    else v_next = v_next * 1664525u + 1013904223u;
    if ((MODE == 3 || MODE == 8 || MODE == 9 || MODE == 10) ? (b != cur) : ((MODE == 4 || MODE == 5) ? false : (k == 0))) {
      if ((MODE == 3 || MODE == 8 || MODE == 9 || MODE == 10) && cur != BID_NONE) acc.store(buckets + (size_t)cur * IO::REC_WORDS);
      cur = b;
      acc.start_signed(q, negate);
    } else {
      acc.accumulate_signed(q, negate);
    }
  }
  u32 s = 0;
  for (int j = 0; j < 9; j++) s ^= acc.a.X.l[j] ^ acc.a.Y.l[j] ^ acc.a.ZZ.l[j] ^ acc.a.ZZZ.l[j];
  out[t] = s + sink + dummy[0] * 0;
}

template <class F> double timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; r++) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int CU = prop.multiProcessorCount;
  const int L = 86, blocks = CU * 3, lanes = blocks * 256;
  const size_t NP = (size_t)1 << 21;   // records of the table (2^20 bases, GLV: twice)
  const size_t NE = (size_t)lanes * L;
  std::vector<u32> h_idx(NE), h_bid(NE), h_seq(NE);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  u32 bucket = 0, left = 0;
  for (size_t i = 0; i < NE; i++) {
    if (left == 0) { bucket = (bucket + 1) & 0x3ffff; left = 32 + (u32)(rnd() % 64); }   // runs of 32..95 entries
    left--;
    h_bid[i] = bucket;
    h_idx[i] = (u32)(rnd() % NP) | ((rnd() & 1) ? SIDX_NEG : 0u);
    h_seq[i] = (u32)(i % NP) | ((rnd() & 1) ? SIDX_NEG : 0u);
  }
  u32 *d_idx, *d_seq, *d_bid, *d_pts, *d_buckets, *d_out;
  (void)hipMalloc(&d_idx, NE * 4);
  (void)hipMalloc(&d_seq, NE * 4);
  (void)hipMalloc(&d_bid, NE * 4);
  (void)hipMalloc(&d_pts, NP * 64);
  (void)hipMalloc(&d_buckets, ((size_t)1 << 18) * 160);
  (void)hipMalloc(&d_out, (size_t)lanes * 4);
  (void)hipMemcpy(d_idx, h_idx.data(), NE * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_seq, h_seq.data(), NE * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_bid, h_bid.data(), NE * 4, hipMemcpyHostToDevice);
  {   // table: pseudo-random 29-bit-limb-compatible words (values below 2^254)
    std::vector<u32> h_pts(NP * 16);
    for (size_t i = 0; i < h_pts.size(); i++) h_pts[i] = (u32)rnd() & ((i % 8) == 7 ? 0x0fffffffu : 0xffffffffu);
    (void)hipMemcpy(d_pts, h_pts.data(), NP * 64, hipMemcpyHostToDevice);
  }
  uint2* d_ent;
  (void)hipMalloc(&d_ent, NE * 8);
  {
    std::vector<uint2> h(NE);
    for (size_t i = 0; i < NE; i++) h[i] = make_uint2(h_idx[i], h_bid[i]);
    (void)hipMemcpy(d_ent, h.data(), NE * 8, hipMemcpyHostToDevice);
  }
  uint2* d_ent_t;
  (void)hipMalloc(&d_ent_t, NE * 8);
  {
    std::vector<uint2> h(NE);
    for (size_t tt = 0; tt < (size_t)lanes; tt++)
      for (int k = 0; k < L; k++) h[(size_t)k * lanes + tt] = make_uint2(h_idx[tt * L + k], h_bid[tt * L + k]);
    (void)hipMemcpy(d_ent_t, h.data(), NE * 8, hipMemcpyHostToDevice);
  }
  u32* d_pts80;
  (void)hipMalloc(&d_pts80, NP * 80);
  {
    std::vector<u32> h(NP * 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = ((i % 20) >= 18) ? 0u : ((u32)rnd() & (((i % 20) % 9) == 8 ? 0x003fffffu : 0x1fffffffu));
    (void)hipMemcpy(d_pts80, h.data(), NP * 80, hipMemcpyHostToDevice);
  }
  const size_t lds = 41216;
  const char* names[11] = {"A formula, q in registers (same q: +-q cancels, INVALID)", "B + gather (random, one entry ahead)", "C + gather (sequential indices)",
                          "D + run logic and run-end stores (random gather)", "E gather issued, record unused; formula on a register q",
                          "F no loads, q changed by register ops every entry",
                          "G = B with the record prefetched into LDS (global_load_lds_dwordx4)",
                          "H = B over UNPACKED 80-byte records (no unpacking, 25 % more bytes)",
                          "I = D with (index, bucket id) interleaved: one 8-byte load per entry instead of two 4-byte ones",
                          "J = I with the entry stream transposed (entry k of lane t at [k][t]): coalesced entry reads",
                          "K = I with the record by four buffer_load_dwordx4 off one 32-bit offset"};
  for (int mode = 0; mode < 11; mode++) {
    double ms = mode == 0   ? timeit([&] { hipLaunchKernelGGL(k_loop<0>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L); })
                : mode == 1 ? timeit([&] { hipLaunchKernelGGL(k_loop<1>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L); })
                : mode == 2 ? timeit([&] { hipLaunchKernelGGL(k_loop<1>, dim3(blocks), dim3(256), lds, 0, d_seq, d_bid, d_pts, d_buckets, d_out, L); })
                : mode == 4 ? timeit([&] { hipLaunchKernelGGL(k_loop<4>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L); })
                : mode == 5 ? timeit([&] { hipLaunchKernelGGL(k_loop<5>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L); })
                : mode == 10 ? timeit([&] { hipLaunchKernelGGL(k_loop<10>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L, d_ent); })
                : mode == 9 ? timeit([&] { hipLaunchKernelGGL(k_loop<9>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L, d_ent_t); })
                : mode == 8 ? timeit([&] { hipLaunchKernelGGL(k_loop<8>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L, d_ent); })
                : mode == 7 ? timeit([&] { hipLaunchKernelGGL(k_loop<7>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts80, d_buckets, d_out, L); })
                : mode == 6 ? timeit([&] { hipLaunchKernelGGL(k_loop<6>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L); })
                            : timeit([&] { hipLaunchKernelGGL(k_loop<3>, dim3(blocks), dim3(256), lds, 0, d_idx, d_bid, d_pts, d_buckets, d_out, L); });
    const double adds = (double)NE;
    printf("%-52s %.3f ms for %.2f M additions -> %.3f ms per 16.78 M\n", names[mode], ms, adds * 1e-6, ms * 16.777216e6 / adds);
  }
  return 0;
}
