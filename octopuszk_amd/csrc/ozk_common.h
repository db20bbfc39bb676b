// Host-side plumbing shared by the C-ABI translation units: thread-local error string,
// HIP error checks, device selection by taskID, a small bump allocator over a caller- or
// library-owned workspace.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <pthread.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ozk.h"

namespace ozk {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define OZK_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return ::ozk::fail(e_ == hipErrorOutOfMemory ? OZK_E_NOMEM : OZK_E_NO_DEVICE,         \
                         "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                         __LINE__);                                                         \
  } while (0)

// select the device the reference would: taskID % num_gpus
// (algebra_msm_VariableBaseMSM.cu:1249-1257)
inline int select_device(int task_id) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(OZK_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  int dev = ((task_id % n) + n) % n;
  OZK_HIP(hipSetDevice(dev));
  return OZK_OK;
}

struct Bump {
  uint8_t* base;
  size_t size, off;
  Bump(void* p, size_t n) : base((uint8_t*)p), size(n), off(0) {}
  template <class T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* r = (T*)(base ? base + off : nullptr);
    off += count * sizeof(T);
    return r;
  }
  bool ok() const { return off <= size; }
};

inline int ilog2(uint32_t v) {
  int r = 0;
  while (v >>= 1) r++;
  return r;
}
// Tuning switches (OZK_* environment variables) are read ONCE and cached: an MSM asks for its plan several
// times (workspace-size query, head, tail), and a variable changing in between would desynchronise the
// layouts.  ozk_tuning_reload() (tests, tuning scripts) drops the cache.
// The cache is PER THREAD (round 4): a host entry point reads dozens of knobs, and through round 3 every read
// took one process-wide mutex — eight Spark task threads on eight devices serialised on it.  A thread's table
// is valid for one generation of the process-wide counter that env_reload() bumps; a read costs a pointer
// compare (the names are string literals) or a short strcmp, no lock and no shared write.
inline std::atomic<unsigned>& env_generation() {
  static std::atomic<unsigned> g(1);
  return g;
}
struct EnvCache {
  unsigned gen = 0;
  int n = 0;
  const char* ptr[96];
  char names[96][40];
  int vals[96];
  bool has[96];
};
inline int env_int(const char* name, int dflt) {
  static thread_local EnvCache c;
  const unsigned g = env_generation().load(std::memory_order_acquire);
  if (c.gen != g) {
    c.gen = g;
    c.n = 0;
  }
  int k = 0;
  for (; k < c.n; k++)
    if (c.ptr[k] == name || strcmp(c.names[k], name) == 0) break;
  if (k == c.n) {
    const char* s = getenv(name);
    const bool has = s && *s;
    const int val = has ? atoi(s) : 0;
    if (c.n < 96 && strlen(name) < 40) {
      c.ptr[k] = name;
      strcpy(c.names[k], name);
      c.has[k] = has;
      c.vals[k] = val;
      c.n++;
    }
    return has ? val : dflt;
  }
  return c.has[k] ? c.vals[k] : dflt;
}
inline void env_reload() { env_generation().fetch_add(1, std::memory_order_acq_rel); }

// The calling thread's HIP error state may hold an error left by somebody else's call (another native library in
// the same process, torch's probes; the runtime keeps the last NON-success code until it is read —
// hip_runtime_api.h, hipGetLastError).  Every function here that checks its kernel launches with
// hipGetLastError() first drops whatever was there, so that the check reports this library's launches only.
inline void hip_clear_stale() { (void)hipGetLastError(); }

}  // namespace ozk
