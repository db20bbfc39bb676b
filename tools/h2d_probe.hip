// Host-to-device strategies for the JNI-shaped entry points (128 MiB = one 2^20 G1 call):
// pageable hipMemcpyAsync, pinned, hipHostRegister + copy + unregister, staged through pinned chunks
// (1 and 4 copy threads).  Build: hipcc --offload-arch=gfx950 -O2 h2d_probe.hip -o h2d_probe -lpthread
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
  const size_t N = 128u << 20;
  uint8_t* d; CK(hipMalloc((void**)&d, N));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint8_t* pageable = (uint8_t*)malloc(N); memset(pageable, 1, N);
  uint8_t* pinned; CK(hipHostMalloc((void**)&pinned, N, hipHostMallocDefault)); memset(pinned, 2, N);
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now(); CK(hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t1 = now();
    printf("pageable hipMemcpyAsync      %.2f ms  %.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
  }
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now(); CK(hipMemcpyAsync(d, pinned, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t1 = now();
    printf("pinned hipMemcpyAsync        %.2f ms  %.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
  }
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now(); CK(hipHostRegister(pageable, N, hipHostRegisterDefault)); double t1 = now();
    CK(hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double t2 = now();
    CK(hipHostUnregister(pageable)); double t3 = now();
    printf("register %.2f + copy %.2f + unregister %.2f = %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
  }
  // staged: K pinned chunks, host memcpy by T threads, DMA of chunk k overlaps the memcpy of chunk k+1
  for (int T : {1, 2, 4, 8}) for (size_t CH : {(size_t)4 << 20, (size_t)16 << 20}) {
    const int NB = 3;
    uint8_t* stage[NB]; hipEvent_t done[NB];
    for (int i = 0; i < NB; i++) { CK(hipHostMalloc((void**)&stage[i], CH, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      size_t off = 0; int k = 0;
      while (off < N) {
        const size_t len = (N - off < CH) ? (N - off) : CH;
        const int b = k % NB;
        if (k >= NB) CK(hipEventSynchronize(done[b]));
        std::vector<std::thread> th;
        const size_t per = (len + T - 1) / T;
        for (int t = 1; t < T; t++) {
          const size_t o = t * per; if (o >= len) break;
          const size_t l = (len - o < per) ? (len - o) : per;
          th.emplace_back([=] { memcpy(stage[b] + o, pageable + off + o, l); });
        }
        memcpy(stage[b], pageable + off, per < len ? per : len);
        for (auto& x : th) x.join();
        CK(hipMemcpyAsync(d + off, stage[b], len, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(done[b], st));
        off += len; k++;
      }
      CK(hipStreamSynchronize(st));
      double t1 = now();
      if (rep) printf("staged %2zu MiB chunks, %d copy thread(s)  %.2f ms  %.1f GB/s\n", CH >> 20, T, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
    }
    for (int i = 0; i < NB; i++) { hipHostFree(stage[i]); hipEventDestroy(done[i]); }
  }
  // device -> host, 192 MiB (fixed-base G1 output at 2^20)
  {
    const size_t M = 128u << 20;
    for (int rep = 0; rep < 2; rep++) { double t0 = now(); CK(hipMemcpyAsync(pageable, d, M, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t1 = now();
      printf("D2H pageable                 %.2f ms  %.1f GB/s\n", (t1 - t0) * 1e3, M / (t1 - t0) / 1e9); }
    for (int rep = 0; rep < 2; rep++) { double t0 = now(); CK(hipMemcpyAsync(pinned, d, M, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t1 = now();
      printf("D2H pinned                   %.2f ms  %.1f GB/s\n", (t1 - t0) * 1e3, M / (t1 - t0) / 1e9); }
  }
  double t0 = now(); uint8_t* x; CK(hipMalloc((void**)&x, (size_t)1 << 30)); double t1 = now(); CK(hipFree(x)); double t2 = now();
  printf("hipMalloc(1 GiB) %.2f ms, hipFree %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  t0 = now(); hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); t1 = now(); CK(hipStreamDestroy(s2)); t2 = now();
  printf("hipStreamCreate %.3f ms, destroy %.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  return 0;
}
