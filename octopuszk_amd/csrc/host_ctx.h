// Host-side resources of the JNI-shaped (`*_host`) entry points, cached across calls.
//
// The reference allocates, copies synchronously from pageable memory and frees inside every native call
// (algebra_msm_VariableBaseMSM.cu:1260-1265,1292-1303,1405-1409); round 1 of this library did the same with
// a stream created and destroyed per call.  Measured on the MI355X box (profiles/r02_h2d_probe.txt):
// creating a stream costs up to 20 ms, a cold pageable hipMemcpyAsync of 128 MiB 11 ms against 2.3 ms
// once the runtime has the pages pinned — so the per-call cost was dominated by set-up, and varied 4x
// from box to box.  Here:
//   * a POOL of contexts per device (a caller takes one for the duration of its call, so concurrent Spark
//     task threads — SURVEY.md §8b "Threading" — never share a stream or a buffer, and never serialise on
//     a device-wide hipFree): two streams, a few events, a grow-only device arena;
//   * uploads / downloads staged through a ring of pinned 16 MiB buffers filled by a small pool of copy
//     threads (48 GB/s from memory the runtime has never seen, against 57 GB/s for pinned memory), the DMA of
//     chunk k overlapping the memcpy of chunk k + 1, and the kernels that only need the first buffers
//     (digit extraction, base conversion) overlapping the rest of the upload.
// Nothing here computes; ozk_host_cache_release() gives everything back.
#pragma once
#include <chrono>

#include "ozk_common.h"

namespace ozk {

constexpr size_t STAGE_BYTES = (size_t)16 << 20;
constexpr int STAGE_RING = 3;
constexpr int MAX_SLICES = 16;
constexpr size_t RESULT_BYTES = (size_t)64 << 10;

struct HostCtx {
  int device = -1;
  hipStream_t st[3] = {nullptr, nullptr, nullptr};  // [0] compute (+ its copies), [1] second engine (G2 / odd slices),
                                                    // [2] uploads that must not queue behind kernels
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t slice_ev[MAX_SLICES] = {};  // upload of slice s complete
  uint8_t* arena = nullptr;  // device, grow-only
  size_t arena_cap = 0;
  uint8_t* stage[STAGE_RING] = {nullptr, nullptr, nullptr};  // pinned host
  hipEvent_t stage_free[STAGE_RING] = {nullptr, nullptr, nullptr};
  bool stage_busy[STAGE_RING] = {false, false, false};
  uint8_t* result = nullptr;  // pinned host, RESULT_BYTES: where small results land before the memcpy into the caller's memory
  int stage_next = 0;
  HostCtx* next = nullptr;
};

// Takes a context for `task_id % device_count` (creating one when the pool is empty) and makes that device
// current for the calling thread.  Returns OZK_OK or an error code (message in ozk_last_error()).
int ctx_acquire(int task_id, HostCtx** out);
void ctx_release(HostCtx* c);
// device arena of at least `bytes` (contents are NOT preserved when it grows)
int ctx_reserve(HostCtx* c, size_t bytes);
// pageable host -> device on `st`, staged through the pinned ring; returns once every byte has been handed to
// the DMA engine (the copies themselves complete in stream order)
int staged_h2d(HostCtx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t st);
// device -> pageable host, staged; synchronises `st` (the data is in h_dst on return)
int staged_d2h(HostCtx* c, void* h_dst, const void* d_src, size_t bytes, hipStream_t st);
// the same for a buffer that is still being produced: before the copy of bytes [.., end) is queued, gate(arg, end)
// must make `st` wait for their producer (hipStreamWaitEvent) — the download of the first ranges of a result
// overlaps the computation of the later ones
int staged_d2h_gated(HostCtx* c, void* h_dst, const void* d_src, size_t bytes, hipStream_t st,
                     int (*gate)(void*, size_t), void* gate_arg);

// Where the wall time of the calling thread's current / last `*_host` call went (milliseconds; reset by
// ctx_acquire).  A call that takes 5x its median has to show up in exactly one of these waits — how the 10-38 ms
// calls of round 2 were attributed; read with ozk_host_call_stats().
struct HostCallStats {
  double acquire_ms = 0, reserve_ms = 0, stage_wait_ms = 0, memcpy_in_ms = 0, memcpy_out_ms = 0, enqueue_ms = 0,
         sync_ms = 0;
  int stage_waits = 0, memcpys = 0;
  double total_ms = 0;  // from the start of ctx_acquire to the release of the context, as the library sees it
  std::chrono::steady_clock::time_point t_begin;
};
HostCallStats& host_call_stats();
struct StatTimer {   // adds the scope's wall time to a field of the calling thread's stats
  double& field;
  std::chrono::steady_clock::time_point t0;
  explicit StatTimer(double& f) : field(f), t0(std::chrono::steady_clock::now()) {}
  ~StatTimer() { field += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// A small result (a point, a few KiB) from device memory to the caller's PAGEABLE memory, through the context's
// pinned result buffer: the only caller memory an entry point hands to the HIP runtime is a constant of <= 192 bytes
// (a root of unity, a base point: copied through the runtime's own staging, never pinned).  (The runtime may pin
// — register with the kernel driver — pageable ranges it is given and keep the registration; when such a range is
// later unmapped or recycled by the caller's allocator, the driver's MMU notifier evicts ALL GPU queues of the
// process for 10-30 ms.  That is what the alternating 6 / 25 ms calls of tools/host_path.py were: the harness's own
// torch.from_numpy(large temporary).cuda() uploads, DESIGN.md section 6.  Large transfers already went through the
// staging ring; this closes the last pageable destinations.)
// small_d2h_begin queues the copy into the result buffer at `slot_off`; small_d2h_end, after the stream has been
// waited for, copies it out.
int small_d2h_begin(HostCtx* c, size_t slot_off, const void* d_src, size_t bytes, hipStream_t st);
void small_d2h_end(HostCtx* c, size_t slot_off, void* h_dst, size_t bytes);

struct CtxGuard {  // release on scope exit
  HostCtx* c = nullptr;
  ~CtxGuard() {
    if (!c) return;
    // an entry point that fails half way may leave work queued that still uses the arena or the staging ring:
    // nothing of a context is handed to the next caller before its streams have drained (idle streams: ~us)
    {
      StatTimer tm(host_call_stats().sync_ms);
      for (auto& s : c->st)
        if (s) hipStreamSynchronize(s);
    }
    ctx_release(c);
    HostCallStats& hs = host_call_stats();
    hs.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs.t_begin).count();
  }
};

inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace ozk
