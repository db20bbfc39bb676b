// VERDICT r3 "missing" 5, second half: BATCH-AFFINE additions measured, not estimated.
//
// Level 1 of the variable-base MSM adds 16.78 M affine points into XYZZ accumulators (8 M + 2 S each, 1.08-1.14 ms
// alone for a 2^20 MSM; the formula alone 0.95-0.97 ms: profiles/r04_ubench_madd.txt).  The alternative every CPU
// Pippenger of the last years uses is the AFFINE chord addition with the inversions batched (Montgomery's trick):
//     d_j = xB_j - xA_j;  pref_j = pref_(j-1) d_j;  ONE inversion of pref_(b-1);  walking back:
//     inv_j = inv * pref_(j-1), inv *= d_j;  lambda = (yB - yA) inv_j;  x3 = lambda^2 - xA - xB;  y3 = lambda (xA - x3) - yA
// = 5 M + 1 S per addition + (one inversion) / b.  On a GPU the batch has to live PER LANE (a serial inversion on one
// lane costs the wave as much as 64 of them), so a lane needs b prefix products somewhere: 36 B each.  This program
// is the best case of that design — no sorting, no tree bookkeeping, no equal-x handling, pairs laid out so that every
// index read, prefix write and result write is coalesced:
//   * MODE 0: operands GATHERED at random from a 128 MiB table of 64-byte affine records (the first tree level:
//     half of all additions); MODE 1: operands contiguous (the later levels read what the level before wrote);
//   * prefix products in global memory, limb-major ([j][limb][thread]) — per-lane batches of 16-64 do not fit the LDS
//     (256 lanes x 16 x 36 B = 147 KB);
//   * "+noinv": the inversion left out (the limit b -> infinity).
// Results are checked on the host for MODE 0 against the chord formula over exact integers
// (tools/ubench_batch_affine_check.py reads the CHK lines).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOZK_WITH_G2 tools/ubench_batch_affine.hip -o tools/ubench_batch_affine
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../octopuszk_amd/csrc/curve.cuh"
using namespace ozk;

using IO = CurveIO<G1Cfg>;
using EA = typename G1Cfg::EA;
constexpr int TBL_LOG = 21;   // 2^21 records x 64 B = 128 MiB: the converted base table of a 2^20 MSM with GLV

__device__ __forceinline__ u32 mix(u32 x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void k_fill(u32* table, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32* p = table + (size_t)i * 16;
  for (int j = 0; j < 16; j++) p[j] = mix(i * 16 + j + 0x9e3779b9u);
  p[7] &= 0x0fffffffu;    // < 2^252 < p: a valid residue
  p[15] &= 0x0fffffffu;
}

// pair k: two distinct records
__device__ __forceinline__ uint2 pair_of(u32 k, bool contiguous) {
  constexpr u32 M = (1u << TBL_LOG) - 1;
  if (contiguous) return make_uint2((2 * k) & M, (2 * k + 1) & M);
  const u32 a = mix(2 * k + 1) & M;
  const u32 b = (a + 1 + mix(2 * k + 2) % M) & M;   // never a
  return make_uint2(a, b);
}

template <int B, bool CONTIG, bool NOINV>
__global__ void __launch_bounds__(256) k_batch(const u32* __restrict__ table, u32* __restrict__ scratch,
                                               u32* __restrict__ out) {
  const u32 T = gridDim.x * 256, tid = blockIdx.x * 256 + threadIdx.x;
  Fe<FqParams, 32> pref = Fe<FqParams, 32>(fe_one<FqParams>());
#pragma unroll 1
  for (int j = 0; j < B; j++) {
    const uint2 pr = pair_of(j * T + tid, CONTIG);
    const EA xa = ElemTraits<EA>::load(table + (size_t)pr.x * 16), xb = ElemTraits<EA>::load(table + (size_t)pr.y * 16);
    const auto d = sub(xb, xa);
    pref = Fe<FqParams, 32>(reduce_to<32>(mul(pref, d)));
#pragma unroll
    for (int l = 0; l < 9; l++) scratch[((size_t)j * 9 + l) * T + tid] = pref.l[l];
  }
  Fe<FqParams, 32> inv_run;
  if constexpr (NOINV) inv_run = pref;
  else inv_run = inv(pref);
#pragma unroll 1
  for (int j = B - 1; j >= 0; j--) {
    const u32 k = j * T + tid;
    const uint2 pr = pair_of(k, CONTIG);
    const Aff<EA> a = IO::load_aff(table + (size_t)pr.x * 16), b = IO::load_aff(table + (size_t)pr.y * 16);
    Fe<FqParams, 32> pj = Fe<FqParams, 32>(fe_one<FqParams>());
    if (j > 0) {
#pragma unroll
      for (int l = 0; l < 9; l++) pj.l[l] = scratch[((size_t)(j - 1) * 9 + l) * T + tid];
    }
    const auto d = sub(b.x, a.x);
    const auto invj = mul(inv_run, pj);
    inv_run = Fe<FqParams, 32>(reduce_to<32>(mul(inv_run, d)));
    const auto lam = reduce_to<32>(mul(sub(b.y, a.y), invj));
    const auto x3 = reduce_to<32>(sub(sub(sqr(lam), a.x), b.x));
    const auto y3 = sub(mul(lam, sub(a.x, x3)), a.y);
    Aff<EA> r;
    r.x = EA(reduce_to<17>(x3));
    r.y = EA(reduce_to<17>(y3));
    IO::store_aff(r, out + (size_t)k * 16);
  }
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

template <int B, bool CONTIG, bool NOINV>
void run(const u32* table, u32* scratch, u32* out, size_t n_adds, bool check) {
  const int blocks = (int)(n_adds / B / 256);
  const double ms = timeit([&] { hipLaunchKernelGGL((k_batch<B, CONTIG, NOINV>), dim3(blocks), dim3(256), 0, 0, table, scratch, out); });
  hipError_t e = hipDeviceSynchronize();
  const double adds = (double)blocks * 256 * B;
  // bytes the design moves per addition: 2 x 64 (pass 1 touches both records for their x) + 36 + 2 x 64 + 36 + 64
  printf("b=%-3d %-10s %-6s %4d workgroups  %.3f ms for %.2f M additions -> %.3f ms per 16.78 M; %.2f G add/s; %.2f TB/s at 392 B per addition%s\n",
         B, CONTIG ? "contiguous" : "gathered", NOINV ? "noinv" : "", blocks, ms, adds * 1e-6, ms * 16.777216e6 / adds, adds / ms * 1e-6,
         adds * 392 / ms * 1e-9, e == hipSuccess ? "" : "  [HIP ERROR]");
  if (check && !NOINV) {
    // four additions with their operands, as stored (Montgomery residues, R = 2^261), for the host-side check
    const u32 T = blocks * 256;
    u32 h[16], ha[16], hb[16];
    for (u32 k : {0u, 12345u, T + 77u, (u32)(B - 1) * T + 5u}) {
      const u32 M = (1u << TBL_LOG) - 1;
      auto hmix = [](u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
      const u32 ia = hmix(2 * k + 1) & M, ib = (ia + 1 + hmix(2 * k + 2) % M) & M;
      (void)hipMemcpy(h, out + (size_t)k * 16, 64, hipMemcpyDeviceToHost);
      (void)hipMemcpy(ha, table + (size_t)ia * 16, 64, hipMemcpyDeviceToHost);
      (void)hipMemcpy(hb, table + (size_t)ib * 16, 64, hipMemcpyDeviceToHost);
      printf("CHK");
      for (const u32* p : {ha, hb, h})
        for (int c = 0; c < 2; c++) {
          printf(" ");
          for (int w = 7; w >= 0; w--) printf("%08x", p[c * 8 + w]);
        }
      printf("\n");
    }
  }
}

int main(int argc, char** argv) {
  const size_t n_adds = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 24);
  u32 *table, *scratch, *out;
  (void)hipMalloc(&table, ((size_t)64) << TBL_LOG);
  (void)hipMalloc(&scratch, n_adds * 36);
  (void)hipMalloc(&out, n_adds * 64);
  hipLaunchKernelGGL(k_fill, dim3((1 << TBL_LOG) / 256), dim3(256), 0, 0, table, 1 << TBL_LOG);
  (void)hipDeviceSynchronize();
  printf("# batch-affine additions, per-lane batches of b (tools/ubench_batch_affine.hip); compare: k_segreduce<G1Cfg,true,true> 1.08-1.14 ms per 16.78 M alone,\n"
         "# its formula alone 0.95-0.97 ms (profiles/r04_ubench_madd.txt)\n");
  run<8, false, false>(table, scratch, out, n_adds, true);
  run<16, false, false>(table, scratch, out, n_adds, true);
  run<32, false, false>(table, scratch, out, n_adds, true);
  run<64, false, false>(table, scratch, out, n_adds, true);
  run<128, false, false>(table, scratch, out, n_adds, false);
  run<16, true, false>(table, scratch, out, n_adds, false);
  run<32, true, false>(table, scratch, out, n_adds, false);
  run<64, true, false>(table, scratch, out, n_adds, false);
  run<32, false, true>(table, scratch, out, n_adds, false);
  run<32, true, true>(table, scratch, out, n_adds, false);
  run<64, true, true>(table, scratch, out, n_adds, false);
  return 0;
}
