// Host-side plumbing shared by the C-ABI translation units: thread-local error string,
// HIP error checks, device selection by taskID, a small bump allocator over a caller- or
// library-owned workspace.
#pragma once
#include <hip/hip_runtime.h>
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
// Tuning switches (OZK_* environment variables) are read ONCE per process and cached: an MSM asks for
// its plan several times (workspace-size query, head, tail), and a variable changing in between would
// desynchronise the layouts.  ozk_tuning_reload() (tests, tuning scripts) drops the cache.
struct EnvCache {
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  int n = 0;
  char names[64][40];
  int vals[64];
  bool has[64];
};
inline EnvCache& env_cache() {
  static EnvCache c;
  return c;
}
inline int env_int(const char* name, int dflt) {
  EnvCache& c = env_cache();
  pthread_mutex_lock(&c.mu);
  int k = 0;
  for (; k < c.n; k++)
    if (strcmp(c.names[k], name) == 0) break;
  if (k == c.n && c.n < 64 && strlen(name) < 40) {
    const char* s = getenv(name);
    strcpy(c.names[k], name);
    c.has[k] = s && *s;
    c.vals[k] = c.has[k] ? atoi(s) : 0;
    c.n++;
  }
  int r = dflt;
  if (k < c.n) {
    if (c.has[k]) r = c.vals[k];
  } else {  // table full: uncached
    const char* s = getenv(name);
    if (s && *s) r = atoi(s);
  }
  pthread_mutex_unlock(&c.mu);
  return r;
}
inline void env_reload() {
  EnvCache& c = env_cache();
  pthread_mutex_lock(&c.mu);
  c.n = 0;
  pthread_mutex_unlock(&c.mu);
}

}  // namespace ozk
