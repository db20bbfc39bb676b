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
#include "curve.cuh"
#include "ozk_common.h"

namespace ozk {

using FrP = FrParams;
constexpr int FFT_TILE = 1024;   // elements per workgroup tile
constexpr int FFT_THREADS = 256;
constexpr int FFT_MAXK = 8;      // stages per pass
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

// ---- one pass of K stages over an LDS tile ---------------------------------
struct PassArgs {
  const u32* in;    // FIRST: wire input (n x 8 words, plain canonical); else packed workspace
  u32* out;         // LAST: wire out (n x 16 words LE); else packed workspace (n x 8 words)
  const u32* tw;    // omega^t, t < n/2, Montgomery, packed
  int n, logn;
  int sbits;        // stages already done = log2 of the butterfly distance entering this pass
  int K;            // stages in this pass
};

template <int B>
__device__ __forceinline__ Fe<FrP, B> lds_load(const u32* lds, int e) {
  Fe<FrP, B> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = lds[i * FFT_TILE + e];
  return r;
}
template <int B>
__device__ __forceinline__ void lds_store(u32* lds, int e, const Fe<FrP, B>& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) lds[i * FFT_TILE + e] = v.l[i];
}

// stage q (1-based inside the pass) with element bound BIN; recursion unrolls the K stages
template <int Q, int BIN>
__device__ __forceinline__ void fft_stages(u32* lds, const PassArgs& a, int T, int logT, long long u0) {
  if (Q > a.K) return;
  const int M2T = FFT_TILE / 2;  // butterflies per stage in the tile
  for (int b = threadIdx.x; b < M2T; b += FFT_THREADS) {
    const int ul = b & (T - 1);
    const int r = b >> logT;
    const int low = r & ((1 << (Q - 1)) - 1);
    const int high = r >> (Q - 1);
    const int mid0 = (high << Q) | low;
    const int mid1 = mid0 | (1 << (Q - 1));
    const long long u = u0 + ul;
    const long long lo = u & ((1ll << a.sbits) - 1);
    const long long j = ((long long)low << a.sbits) + lo;
    const long long ti = j << (a.logn - a.sbits - Q);
    const auto w = ElemTraits<Fe<FrP, 16>>::load(a.tw + (size_t)ti * 8);
    const int e0 = mid0 * T + ul, e1 = mid1 * T + ul;
    const auto x = lds_load<BIN>(lds, e0);
    const auto y = lds_load<BIN>(lds, e1);
    const auto t = mul(w, y);                       // plain product (w is Montgomery)
    lds_store(lds, e0, Fe<FrP, BIN + 32>(add(x, t)));
    lds_store(lds, e1, Fe<FrP, BIN + 32>(sub(x, t)));
  }
  __syncthreads();
  if constexpr (Q < FFT_MAXK) fft_stages<Q + 1, BIN + 32>(lds, a, T, logT, u0);
}

template <int BEND>
__device__ __forceinline__ void fft_store_tile(u32* lds, const PassArgs& a, int T, int logT, long long u0,
                                               bool last) {
  const int M = FFT_TILE >> logT;
  (void)M;
  for (int e = threadIdx.x; e < FFT_TILE; e += FFT_THREADS) {
    // consecutive lanes -> consecutive ul (contiguous addresses within a T-run)
    const int ul = e & (T - 1), mid = e >> logT;
    const long long u = u0 + ul;
    const long long hi = u >> a.sbits, lo = u & ((1ll << a.sbits) - 1);
    const long long i = (hi << (a.sbits + a.K)) | ((long long)mid << a.sbits) | lo;
    const auto v = lds_load<BEND>(lds, mid * T + ul);
    u32 o[8];
    if (last) {
      pack(canonical(v), o);
      uint4* dst = reinterpret_cast<uint4*>(a.out + (size_t)i * 16);
      dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
      dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
      dst[2] = make_uint4(0, 0, 0, 0);
      dst[3] = make_uint4(0, 0, 0, 0);
    } else {
      pack(reduce_to<64>(v), o);
      uint4* dst = reinterpret_cast<uint4*>(a.out + (size_t)i * 8);
      dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
      dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
  }
}

template <bool FIRST>
__global__ void __launch_bounds__(FFT_THREADS) k_fft_pass(PassArgs a, int last) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];  // 9 * FFT_TILE words
  const int M = 1 << a.K;
  const int T = FFT_TILE / M;
  const int logT = 31 - __clz(T);
  const long long u0 = (long long)blockIdx.x * T;
  // load the tile: element (mid, ul) <- global index i (FIRST: bit-reversed source)
  for (int e = threadIdx.x; e < FFT_TILE; e += FFT_THREADS) {
    const int ul = e & (T - 1), mid = e >> logT;
    const long long u = u0 + ul;
    const long long hi = u >> a.sbits, lo = u & ((1ll << a.sbits) - 1);
    const long long i = (hi << (a.sbits + a.K)) | ((long long)mid << a.sbits) | lo;
    long long src = i;
    if (FIRST) src = (long long)(__brev((unsigned)i) >> (32 - a.logn));
    const uint4* sp = reinterpret_cast<const uint4*>(a.in + (size_t)src * 8);
    const uint4 v0 = sp[0], v1 = sp[1];
    const u32 w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    // FIRST: arbitrary 256-bit wire value (bound 85); later passes: stored < 4p
    lds_store(lds, mid * T + ul, unpack<FrP, 85>(w));
  }
  __syncthreads();
  constexpr int B0 = 96;  // >= 85 (any 256-bit input) and >= 64 (inter-pass storage)
  fft_stages<1, B0>(lds, a, T, logT, u0);
  // after K stages the bound is B0 + 32*K <= B0 + 32*FFT_MAXK
  fft_store_tile<B0 + 32 * FFT_MAXK>(lds, a, T, logT, u0, last != 0);
}

// n == 1 or tiny n (< FFT_TILE): one workgroup, direct global-memory version of the same
// algorithm (bit reversal + stages), one butterfly per lane per stage.
__global__ void __launch_bounds__(FFT_THREADS) k_fft_small(const u32* __restrict__ in, u32* __restrict__ out,
                                                           const u32* __restrict__ tw, int n, int logn,
                                                           u32* __restrict__ scratch) {
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
  __syncthreads();
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
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += FFT_THREADS) {
    u32 o[8];
    pack(canonical(ET::load(scratch + (size_t)i * 8)), o);
#pragma unroll
    for (int k = 0; k < 8; k++) out[(size_t)i * 16 + k] = o[k];
#pragma unroll
    for (int k = 8; k < 16; k++) out[(size_t)i * 16 + k] = 0;
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
  L.tw = b.take<u32>((size_t)half * 8);
  L.buf[0] = b.take<u32>((size_t)n * 8);
  L.buf[1] = b.take<u32>((size_t)n * 8);
  b.take<u32>(64);
  L.bytes = b.off;
  return L;
}

static int fft_dev(const void* d_in, int n, const uint8_t* omega_host, void* d_out, void* wsp, size_t wsb,
                   hipStream_t st) {
  const int logn = ilog2((uint32_t)n);
  const FftLayout L = fft_layout(n, wsp, wsb);
  if (L.bytes > wsb) return fail(OZK_E_INVALID, "workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
  OZK_HIP(hipMemcpyAsync(L.omega, omega_host, 32, hipMemcpyHostToDevice, st));
  const int half = n / 2 > 0 ? n / 2 : 1;
  hipLaunchKernelGGL(k_tw_small, dim3((L.lo + L.hi + 255) / 256), dim3(256), 0, st, L.omega, L.lo, L.hi, L.small);
  hipLaunchKernelGGL(k_tw_full, dim3((half + 255) / 256), dim3(256), 0, st, L.small, L.lo, half, L.tw);
  if (n < FFT_TILE) {
    hipLaunchKernelGGL(k_fft_small, dim3(1), dim3(FFT_THREADS), 0, st, (const u32*)d_in, (u32*)d_out, L.tw, n, logn,
                       L.buf[0]);
    OZK_HIP(hipGetLastError());
    return OZK_OK;
  }
  // passes of up to FFT_MAXK stages; the tile needs 2^K <= FFT_TILE
  int sbits = 0, cur = 0;
  const u32* src = (const u32*)d_in;
  bool first = true;
  while (sbits < logn) {
    int K = logn - sbits;
    if (K > FFT_MAXK) K = FFT_MAXK;
    const bool last = sbits + K == logn;
    PassArgs a;
    a.in = src;
    a.out = last ? (u32*)d_out : L.buf[cur];
    a.tw = L.tw;
    a.n = n;
    a.logn = logn;
    a.sbits = sbits;
    a.K = K;
    const int tiles = n / FFT_TILE;
    const size_t lds_bytes = (size_t)9 * FFT_TILE * 4;
    if (first)
      hipLaunchKernelGGL((k_fft_pass<true>), dim3(tiles), dim3(FFT_THREADS), lds_bytes, st, a, last ? 1 : 0);
    else
      hipLaunchKernelGGL((k_fft_pass<false>), dim3(tiles), dim3(FFT_THREADS), lds_bytes, st, a, last ? 1 : 0);
    src = a.out;
    cur ^= 1;
    sbits += K;
    first = false;
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

}  // namespace ozk

using namespace ozk;

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

int ozk_fft_host(const uint8_t* in, int32_t n, const uint8_t* omega, int32_t task_id, uint8_t* out) {
  if (!in || !omega || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || (n & (n - 1)) || n > (1 << 28))
    return fail(OZK_E_INVALID, "FFT size %d is not a power of two in [1, 2^28]", n);
  int rc = select_device(task_id);
  if (rc) return rc;
  const size_t in_bytes = (size_t)n * 32, out_bytes = (size_t)n * 64;
  const size_t wsb = ozk_fft_workspace_bytes(n);
  uint8_t* d = nullptr;
  hipStream_t st = nullptr;
  OZK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipError_t e = hipMalloc((void**)&d, in_bytes + out_bytes + wsb + 1024);
  if (e != hipSuccess) {
    hipStreamDestroy(st);
    return fail(OZK_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  uint8_t* d_out = d + ((in_bytes + 255) & ~(size_t)255);
  uint8_t* d_ws = d_out + ((out_bytes + 255) & ~(size_t)255);
  rc = OZK_OK;
  do {
    if ((e = hipMemcpyAsync(d, in, in_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) break;
    rc = fft_dev(d, n, omega, d_out, d_ws, wsb, st);
    if (rc) break;
    if ((e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    e = hipStreamSynchronize(st);
  } while (0);
  hipFree(d);
  hipStreamDestroy(st);
  if (rc) return rc;
  if (e != hipSuccess) return fail(OZK_E_NO_DEVICE, "HIP failure in fft_host: %s", hipGetErrorString(e));
  return OZK_OK;
}

}  // extern "C"
