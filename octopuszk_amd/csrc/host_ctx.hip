// Context pool, pinned staging ring and copy threads of the `*_host` entry points (host_ctx.h).
#include "host_ctx.h"

#include <atomic>

namespace ozk {

// ---------------------------------------------------------------- copy threads
// A handful of helper threads that do nothing but memcpy between caller memory and the pinned ring: one
// core moves ~12 GB/s, the PCIe link 57 GB/s.  Started on first use; they sleep on a condition variable.
namespace {
constexpr int COPY_HELPERS = 3;  // + the calling thread (default; OZK_COPY_HELPERS up to COPY_HELPERS_MAX)
constexpr int COPY_HELPERS_MAX = 11;
static int copy_helpers() {
  static const int h = env_int("OZK_COPY_HELPERS", COPY_HELPERS);   // 0: the calling thread copies alone
  return h < 0 ? 0 : (h > COPY_HELPERS_MAX ? COPY_HELPERS_MAX : h);
}
constexpr int COPY_QUEUE = 64;
struct CopyTask {
  void* dst;
  const void* src;
  size_t len;
  std::atomic<int>* pending;
};
struct CopyPool {
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
  CopyTask q[COPY_QUEUE];
  int head = 0, count = 0;
  bool started = false, failed = false, stop = false;
  int running = 0;
  pthread_t th[COPY_HELPERS_MAX];
};
CopyPool g_copy;

// next queued piece; with `wait` a helper sleeps here until there is one, or until the pool is shut down (false)
bool copy_pop(CopyTask* t, bool wait) {
  pthread_mutex_lock(&g_copy.mu);
  while (g_copy.count == 0) {
    if (!wait || g_copy.stop) {
      pthread_mutex_unlock(&g_copy.mu);
      return false;
    }
    pthread_cond_wait(&g_copy.cv, &g_copy.mu);
  }
  *t = g_copy.q[g_copy.head];
  g_copy.head = (g_copy.head + 1) % COPY_QUEUE;
  g_copy.count--;
  pthread_mutex_unlock(&g_copy.mu);
  return true;
}
void* copy_worker(void*) {
  CopyTask t;
  while (copy_pop(&t, true)) {
    memcpy(t.dst, t.src, t.len);
    t.pending->fetch_sub(1, std::memory_order_release);
  }
  return nullptr;
}
void copy_start_locked() {
  if (g_copy.started || g_copy.failed) return;
  g_copy.stop = false;
  g_copy.running = 0;
  for (int i = 0; i < copy_helpers(); i++) {
    if (pthread_create(&g_copy.th[i], nullptr, copy_worker, nullptr) != 0) {
      g_copy.failed = true;  // (already running helpers keep working; the caller copies the rest itself)
      break;
    }
    g_copy.running++;
  }
  g_copy.started = g_copy.running > 0;
}
// Stops and JOINS the helpers (ozk_host_cache_release, and the library's destructor: a JVM that unloads the
// native library, or a dlclose, must not leave threads sleeping in unmapped text).  Callers in flight finish their
// own pieces (parallel_memcpy drains the queue itself); the next call starts new helpers.
void copy_shutdown() {
  // One shutdown at a time, and the thread ids leave the pool before anybody joins them: two concurrent
  // ozk_host_cache_release calls (or a release racing the library destructor) used to join the same pthread_t
  // twice — undefined behaviour.
  static pthread_mutex_t shutdown_mu = PTHREAD_MUTEX_INITIALIZER;
  pthread_mutex_lock(&shutdown_mu);
  pthread_t th[COPY_HELPERS_MAX];
  pthread_mutex_lock(&g_copy.mu);
  const int n = g_copy.running;
  for (int i = 0; i < n; i++) th[i] = g_copy.th[i];
  g_copy.running = 0;
  g_copy.stop = true;
  pthread_cond_broadcast(&g_copy.cv);
  pthread_mutex_unlock(&g_copy.mu);
  for (int i = 0; i < n; i++) pthread_join(th[i], nullptr);
  pthread_mutex_lock(&g_copy.mu);
  g_copy.started = false;
  g_copy.failed = false;
  pthread_mutex_unlock(&g_copy.mu);
  pthread_mutex_unlock(&shutdown_mu);
}
// A forked child has none of the helper threads (only the forking thread survives): its pool starts empty, so the
// destructor's join is a no-op there instead of a wait for threads that do not exist.
void copy_atfork_child() {
  pthread_mutex_init(&g_copy.mu, nullptr);
  pthread_cond_init(&g_copy.cv, nullptr);
  g_copy.head = g_copy.count = 0;
  g_copy.running = 0;
  g_copy.started = g_copy.failed = g_copy.stop = false;
}
__attribute__((constructor)) void copy_register_atfork() { pthread_atfork(nullptr, nullptr, copy_atfork_child); }
void parallel_memcpy(void* dst, const void* src, size_t len) {
  const int helpers = copy_helpers();
  if (len < ((size_t)1 << 20) || helpers <= 0) {
    memcpy(dst, src, len);
    return;
  }
  const int parts = helpers + 1;
  const size_t per = ((len + parts - 1) / parts + 63) & ~(size_t)63;
  std::atomic<int> pending(0);
  pthread_mutex_lock(&g_copy.mu);
  copy_start_locked();
  int queued = 0;
  if (g_copy.started) {
    for (int p = 1; p < parts; p++) {
      const size_t o = (size_t)p * per;
      if (o >= len || g_copy.count == COPY_QUEUE) break;
      const size_t l = (len - o < per) ? (len - o) : per;
      pending.fetch_add(1, std::memory_order_relaxed);
      g_copy.q[(g_copy.head + g_copy.count) % COPY_QUEUE] = CopyTask{(uint8_t*)dst + o, (const uint8_t*)src + o, l, &pending};
      g_copy.count++;
      queued++;
    }
    if (queued) pthread_cond_broadcast(&g_copy.cv);
  }
  pthread_mutex_unlock(&g_copy.mu);
  const size_t mine = queued ? per : len;   // the queued parts are 1 .. queued; the rest is this thread's
  memcpy(dst, src, mine < len ? mine : len);
  const size_t done_upto = (size_t)(queued + 1) * per;
  if (queued && done_upto < len) memcpy((uint8_t*)dst + done_upto, (const uint8_t*)src + done_upto, len - done_upto);
  // help with whatever is still queued (ours or another caller's), then wait for ours
  CopyTask t;
  while (pending.load(std::memory_order_acquire) != 0) {
    if (copy_pop(&t, false)) {
      memcpy(t.dst, t.src, t.len);
      t.pending->fetch_sub(1, std::memory_order_release);
    } else {
      sched_yield();
    }
  }
}
}  // namespace

// ---------------------------------------------------------------- context pool
namespace {
constexpr int MAX_DEVICES = 64;
pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
HostCtx* g_free[MAX_DEVICES] = {nullptr};

void ctx_destroy(HostCtx* c) {
  hipSetDevice(c->device);
  for (auto& s : c->st)
    if (s) {
      hipStreamSynchronize(s);
      hipStreamDestroy(s);
    }
  for (auto& e : c->ev)
    if (e) hipEventDestroy(e);
  for (auto& e : c->slice_ev)
    if (e) hipEventDestroy(e);
  for (int i = 0; i < STAGE_RING; i++) {
    if (c->stage[i]) hipHostFree(c->stage[i]);
    if (c->stage_free[i]) hipEventDestroy(c->stage_free[i]);
  }
  if (c->result) hipHostFree(c->result);
  if (c->arena) hipFree(c->arena);
  delete c;
}
}  // namespace

static __global__ void k_ctx_warm() {}

HostCallStats& host_call_stats() {
  static thread_local HostCallStats s;
  return s;
}

int ctx_acquire(int task_id, HostCtx** out) {
  host_call_stats() = HostCallStats();
  host_call_stats().t_begin = std::chrono::steady_clock::now();
  StatTimer tm(host_call_stats().acquire_ms);
  int rc = select_device(task_id);
  if (rc) return rc;
  int dev = 0;
  OZK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(OZK_E_INTERNAL, "device index %d out of range", dev);
  pthread_mutex_lock(&g_pool_mu);
  HostCtx* c = g_free[dev];
  if (c) g_free[dev] = c->next;
  pthread_mutex_unlock(&g_pool_mu);
  if (!c) {
    c = new HostCtx();
    c->device = dev;
    hipError_t e = hipSuccess;
    for (auto& s : c->st)
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& v : c->ev)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&v, hipEventDisableTiming);
    for (auto& v : c->slice_ev)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&v, hipEventDisableTiming);
    for (int i = 0; i < STAGE_RING; i++) {
      if (e == hipSuccess) e = hipHostMalloc((void**)&c->stage[i], STAGE_BYTES, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->stage_free[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->result, RESULT_BYTES, hipHostMallocDefault);
    // The runtime creates a stream's hardware queue at the stream's FIRST operation, not at hipStreamCreate: 7 ms
    // inside whichever hipMemcpyAsync happened to be first on it (the "enqueue 7.09 ms" call of
    // profiles/r03_host_path.txt: the double MSM is the only user of the second stream, so the cost surfaced in one
    // of its calls, long after the context's own first call).  Paid here, once, where a context is created.
    for (auto& s : c->st) {
      if (e != hipSuccess) break;
      hipLaunchKernelGGL(k_ctx_warm, dim3(1), dim3(1), 0, s);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) {
      ctx_destroy(c);
      return fail(e == hipErrorOutOfMemory ? OZK_E_NOMEM : OZK_E_NO_DEVICE, "creating a host context failed: %s",
                  hipGetErrorString(e));
    }
  }
  c->next = nullptr;
  *out = c;
  return OZK_OK;
}

void ctx_release(HostCtx* c) {
  if (!c) return;
  pthread_mutex_lock(&g_pool_mu);
  c->next = g_free[c->device];
  g_free[c->device] = c;
  pthread_mutex_unlock(&g_pool_mu);
}

int ctx_reserve(HostCtx* c, size_t bytes) {
  if (bytes <= c->arena_cap) return OZK_OK;
  StatTimer tm(host_call_stats().reserve_ms);
  // everything queued on this context's streams may still use the old arena
  for (auto& s : c->st) OZK_HIP(hipStreamSynchronize(s));
  if (c->arena) OZK_HIP(hipFree(c->arena));
  c->arena = nullptr;
  c->arena_cap = 0;
  const size_t want = pad256(bytes + bytes / 8);  // some slack: sizes creep with n
  hipError_t e = hipMalloc((void**)&c->arena, want);
  if (e != hipSuccess) return fail(OZK_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  c->arena_cap = want;
  return OZK_OK;
}

static int stage_wait(HostCtx* c, int b) {
  if (c->stage_busy[b]) {
    StatTimer tm(host_call_stats().stage_wait_ms);
    host_call_stats().stage_waits++;
    const auto t0 = std::chrono::steady_clock::now();
    OZK_HIP(hipEventSynchronize(c->stage_free[b]));
    if (env_int("OZK_HOST_TRACE", 0)) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms > 1.0) fprintf(stderr, "[ozk] stage_wait #%d buffer %d: %.2f ms\n", host_call_stats().stage_waits, b, ms);
    }
    c->stage_busy[b] = false;
  }
  return OZK_OK;
}

int small_d2h_begin(HostCtx* c, size_t slot_off, const void* d_src, size_t bytes, hipStream_t st) {
  if (slot_off + bytes > RESULT_BYTES) return fail(OZK_E_INTERNAL, "result of %zu bytes does not fit the pinned result buffer", bytes);
  StatTimer tq(host_call_stats().enqueue_ms);
  OZK_HIP(hipMemcpyAsync(c->result + slot_off, d_src, bytes, hipMemcpyDeviceToHost, st));
  return OZK_OK;
}
void small_d2h_end(HostCtx* c, size_t slot_off, void* h_dst, size_t bytes) { memcpy(h_dst, c->result + slot_off, bytes); }

int staged_h2d(HostCtx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t st) {
  size_t off = 0;
  while (off < bytes) {
    const size_t len = (bytes - off < STAGE_BYTES) ? (bytes - off) : STAGE_BYTES;
    const int b = c->stage_next;
    c->stage_next = (b + 1) % STAGE_RING;
    int rc = stage_wait(c, b);
    if (rc) return rc;
    {
      StatTimer tm(host_call_stats().memcpy_in_ms);
      host_call_stats().memcpys++;
      parallel_memcpy(c->stage[b], (const uint8_t*)h_src + off, len);
    }
    StatTimer tq(host_call_stats().enqueue_ms);
    const auto te0 = std::chrono::steady_clock::now();
    OZK_HIP(hipMemcpyAsync((uint8_t*)d_dst + off, c->stage[b], len, hipMemcpyHostToDevice, st));
    OZK_HIP(hipEventRecord(c->stage_free[b], st));
    if (env_int("OZK_HOST_TRACE", 0) >= 2)   // absolute CLOCK_MONOTONIC ns: comparable with rocprofv3's timestamps
      fprintf(stderr, "[ozk] h2d chunk %zu B buffer %d stream %p: enqueue began %lld ns, returned %lld ns\n", len, b, (void*)st,
              (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(te0.time_since_epoch()).count(),
              (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
    c->stage_busy[b] = true;
    off += len;
  }
  return OZK_OK;
}

int staged_d2h(HostCtx* c, void* h_dst, const void* d_src, size_t bytes, hipStream_t st) {
  return staged_d2h_gated(c, h_dst, d_src, bytes, st, nullptr, nullptr);
}

int staged_d2h_gated(HostCtx* c, void* h_dst, const void* d_src, size_t bytes, hipStream_t st,
                     int (*gate)(void*, size_t), void* gate_arg) {
  if (bytes <= ((size_t)1 << 16)) {  // a result point: one small synchronous copy
    if (gate) {
      int rc = gate(gate_arg, bytes);
      if (rc) return rc;
    }
    StatTimer tm(host_call_stats().sync_ms);
    int rc = small_d2h_begin(c, 0, d_src, bytes, st);
    if (rc) return rc;
    OZK_HIP(hipStreamSynchronize(st));
    small_d2h_end(c, 0, h_dst, bytes);
    return OZK_OK;
  }
  const size_t nchunks = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
  int buf_of[STAGE_RING];
  auto issue = [&](size_t k) -> int {
    const int b = c->stage_next;
    c->stage_next = (b + 1) % STAGE_RING;
    int rc = stage_wait(c, b);
    if (rc) return rc;
    const size_t off = k * STAGE_BYTES;
    const size_t len = (bytes - off < STAGE_BYTES) ? (bytes - off) : STAGE_BYTES;
    if (gate) {  // the producer of bytes [0, off + len) must have been ordered before this copy
      rc = gate(gate_arg, off + len);
      if (rc) return rc;
    }
    {
      StatTimer tq(host_call_stats().enqueue_ms);
      OZK_HIP(hipMemcpyAsync(c->stage[b], (const uint8_t*)d_src + off, len, hipMemcpyDeviceToHost, st));
      OZK_HIP(hipEventRecord(c->stage_free[b], st));
    }
    c->stage_busy[b] = true;
    buf_of[k % STAGE_RING] = b;
    return OZK_OK;
  };
  size_t issued = 0;
  for (; issued < nchunks && issued < (size_t)STAGE_RING; issued++) {
    int rc = issue(issued);
    if (rc) return rc;
  }
  for (size_t k = 0; k < nchunks; k++) {
    const int b = buf_of[k % STAGE_RING];
    int rc = stage_wait(c, b);
    if (rc) return rc;
    const size_t off = k * STAGE_BYTES;
    const size_t len = (bytes - off < STAGE_BYTES) ? (bytes - off) : STAGE_BYTES;
    {
      StatTimer tm(host_call_stats().memcpy_out_ms);
      host_call_stats().memcpys++;
      parallel_memcpy((uint8_t*)h_dst + off, c->stage[b], len);
    }
    if (issued < nchunks) {
      rc = issue(issued++);
      if (rc) return rc;
    }
  }
  return OZK_OK;
}

}  // namespace ozk

namespace ozk {
void fft_plan_cache_release();   // fft.hip
void fb_table_cache_release();   // msm_fixed.hip
}

namespace {
__attribute__((destructor)) void ozk_host_ctx_unload() { ozk::copy_shutdown(); }
}  // namespace

// stats10 = {acquire, reserve, stage_wait, memcpy_in, memcpy_out, enqueue, sync (ms), stage waits, memcpys, total ms
// inside the library} of the calling thread's last *_host call
extern "C" int ozk_host_call_stats(double* stats9) {
  if (!stats9) return ozk::fail(OZK_E_INVALID, "null pointer argument");
  stats9[9] = ozk::host_call_stats().total_ms;
  const ozk::HostCallStats& s = ozk::host_call_stats();
  stats9[0] = s.acquire_ms;
  stats9[1] = s.reserve_ms;
  stats9[2] = s.stage_wait_ms;
  stats9[3] = s.memcpy_in_ms;
  stats9[4] = s.memcpy_out_ms;
  stats9[5] = s.enqueue_ms;
  stats9[6] = s.sync_ms;
  stats9[7] = s.stage_waits;
  stats9[8] = s.memcpys;
  return OZK_OK;
}

// Streams confined to a range of compute units (round 4): the three-stage schedule's latency-bound tail kernels slow
// the bucket accumulation by sitting on its SIMDs; a caller may give the tails `n_cus` units of their own and the
// accumulation the rest (device.VarMsmPipeline3(tail_cus=...)).  The runtime creates such a stream as a BLOCKING stream:
// it synchronises with the device's null stream, so the schedule that uses it must not run on the null stream.
extern "C" int ozk_device_cu_count(void) {
  using namespace ozk;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return prop.multiProcessorCount;
}
extern "C" int ozk_stream_create_cu_range(int32_t first_cu, int32_t n_cus, void** stream) {
  using namespace ozk;
  hip_clear_stale();
  if (!stream) return fail(OZK_E_INVALID, "null pointer argument");
  const int total = ozk_device_cu_count();
  if (total <= 0) return fail(OZK_E_NO_DEVICE, "no device properties");
  if (first_cu < 0 || n_cus <= 0 || first_cu + n_cus > total)
    return fail(OZK_E_INVALID, "compute-unit range [%d, %d) outside the device's %d", first_cu, first_cu + n_cus, total);
  uint32_t mask[32] = {0};
  const int words = (total + 31) / 32;
  if (words > 32) return fail(OZK_E_INVALID, "%d compute units: mask too long", total);
  for (int i = first_cu; i < first_cu + n_cus; i++) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  OZK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *stream = (void*)s;
  return OZK_OK;
}
extern "C" int ozk_stream_destroy(void* stream) {
  using namespace ozk;
  if (!stream) return OZK_OK;
  OZK_HIP(hipStreamDestroy((hipStream_t)stream));
  return OZK_OK;
}

extern "C" int ozk_host_cache_release(void) {
  using namespace ozk;
  copy_shutdown();
  fft_plan_cache_release();
  fb_table_cache_release();
  ozk_shard_comms_release();   // msm_var.hip: the sharded entry's RCCL communicators
  pthread_mutex_lock(&g_pool_mu);
  for (int d = 0; d < MAX_DEVICES; d++) {
    HostCtx* c = g_free[d];
    g_free[d] = nullptr;
    while (c) {
      HostCtx* n = c->next;
      ctx_destroy(c);
      c = n;
    }
  }
  pthread_mutex_unlock(&g_pool_mu);
  return OZK_OK;
}
