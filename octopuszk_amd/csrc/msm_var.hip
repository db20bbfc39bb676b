// Variable-base MSM: C ABI (include/ozk.h) and the G1 instantiation of the driver templates
// (msm_var_driver.cuh); the G2 instantiation is msm_var_g2.hip.
// Replaces pippengerMSMG1 / pippengerMSMG2 (algebra_msm_VariableBaseMSM.cu:1246-1604) and
// the two JNI natives of algebra.msm.VariableBaseMSM (.cu:1614-1788).
#include "msm_var_driver.cuh"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <string>
#include <thread>
#include <vector>

OZK_G2_DRIVER_INSTANCES(extern)

using namespace ozk;

extern "C" {

size_t ozk_var_msm_workspace_bytes(int32_t n, int32_t type) {
  if (n <= 0) return 0;
  if (type == OZK_G1) return var_msm_ws_bytes<G1Cfg>(n);
#if defined(OZK_WITH_G2)
  return var_msm_ws_bytes<G2Cfg>(n);
#else
  return 0;
#endif
}

int ozk_var_msm_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type, void* d_out,
                    void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!d_bases || !d_scalars || !d_out || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_dev<G1Cfg>(d_bases, d_scalars, n, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
#if defined(OZK_WITH_G2)
  return var_msm_dev<G2Cfg>(d_bases, d_scalars, n, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
#else
  return fail(OZK_E_INVALID, "G2 not built");
#endif
}

size_t ozk_var_msm_head_workspace_bytes(int32_t n, int32_t type) {
  if (n <= 0) return 0;
  return type == OZK_G1 ? var_msm_head_ws_bytes<G1Cfg>(n) : var_msm_head_ws_bytes<G2Cfg>(n);
}
size_t ozk_var_msm_tail_bytes(int32_t n, int32_t type) {
  if (n <= 0) return 0;
  return type == OZK_G1 ? var_msm_tail_bytes<G1Cfg>(n) : var_msm_tail_bytes<G2Cfg>(n);
}
int ozk_var_msm_head_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type, void* d_workspace,
                         size_t workspace_bytes, void* d_tail, size_t tail_bytes, void* stream) {
  if (!d_bases || !d_scalars || !d_workspace || !d_tail) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_head<G1Cfg>(d_bases, d_scalars, n, d_workspace, workspace_bytes, d_tail, tail_bytes,
                               (hipStream_t)stream);
  return var_msm_head<G2Cfg>(d_bases, d_scalars, n, d_workspace, workspace_bytes, d_tail, tail_bytes,
                             (hipStream_t)stream);
}
size_t ozk_var_msm_prepared_bytes(int32_t n, int32_t type) {
  if (n <= 0 || n > (1 << 24)) return 0;
  return type == OZK_G1 ? prepared_bytes<G1Cfg>(n) : prepared_bytes<G2Cfg>(n);
}
int ozk_var_msm_prepare_dev(const void* d_bases, int32_t n, int32_t type, void* d_prepared, size_t prepared_size,
                            void* stream) {
  if (!d_bases || !d_prepared) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1) return var_msm_prepare<G1Cfg>(d_bases, n, d_prepared, prepared_size, (hipStream_t)stream);
  return var_msm_prepare<G2Cfg>(d_bases, n, d_prepared, prepared_size, (hipStream_t)stream);
}
int ozk_var_msm_head_prepared_dev(const void* d_prepared, const void* d_scalars, int32_t n, int32_t type,
                                  void* d_workspace, size_t workspace_bytes, void* d_tail, size_t tail_bytes,
                                  void* stream, void* previous_levels_done) {
  if (!d_prepared || !d_scalars || !d_workspace || !d_tail) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_head<G1Cfg>(nullptr, d_scalars, n, d_workspace, workspace_bytes, d_tail, tail_bytes,
                               (hipStream_t)stream, (hipEvent_t)previous_levels_done, d_prepared);
  return var_msm_head<G2Cfg>(nullptr, d_scalars, n, d_workspace, workspace_bytes, d_tail, tail_bytes,
                             (hipStream_t)stream, (hipEvent_t)previous_levels_done, d_prepared);
}
int ozk_var_msm_prepared_dev(const void* d_prepared, const void* d_scalars, int32_t n, int32_t type, void* d_out,
                             void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!d_prepared || !d_scalars || !d_out || !d_workspace) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_dev<G1Cfg>(nullptr, d_scalars, n, d_out, d_workspace, workspace_bytes, (hipStream_t)stream,
                              d_prepared);
  return var_msm_dev<G2Cfg>(nullptr, d_scalars, n, d_out, d_workspace, workspace_bytes, (hipStream_t)stream,
                            d_prepared);
}
// ---- handles of prepared bases: generation-checked table.  The value handed to the caller is a TOKEN
// (generation << 32 | slot + 1), never a pointer: a stale or forged token resolves to nothing instead of to freed —
// or recycled — memory, and a released handle's memory really is released (round 2 kept every dead handle's
// header allocated for the life of the process so that late callers could be told apart).  A handle in use is
// pinned by a reference count; the last user of a destroyed handle frees it.
namespace {
struct HandleSlot {
  BasesHandle* h = nullptr;
  uint32_t gen = 1;
};
pthread_mutex_t g_handle_mu = PTHREAD_MUTEX_INITIALIZER;
std::vector<HandleSlot> g_handles;

void bases_free(BasesHandle* h) {  // nobody else can reach h any more
  hipSetDevice(h->device);
  if (h->st) hipStreamSynchronize(h->st);
  hipFree(h->d_prepared);
  hipFree(h->d_scalars);
  hipFree(h->d_out);
  if (h->h_out) hipHostFree(h->h_out);
  hipFree(h->d_ws);
  if (h->st) hipStreamDestroy(h->st);
  pthread_mutex_destroy(&h->mu);
  free(h);
}
void* handle_publish(BasesHandle* h) {
  pthread_mutex_lock(&g_handle_mu);
  size_t i = 0;
  for (; i < g_handles.size(); i++)
    if (!g_handles[i].h) break;
  if (i == g_handles.size()) g_handles.push_back(HandleSlot());
  g_handles[i].h = h;
  const uint64_t token = ((uint64_t)g_handles[i].gen << 32) | (uint64_t)(i + 1);
  pthread_mutex_unlock(&g_handle_mu);
  return (void*)(uintptr_t)token;
}
// token -> pinned handle (nullptr: stale, released or never issued); `unpublish` also retires the slot
BasesHandle* handle_pin(void* token, bool unpublish = false) {
  const uint64_t t = (uint64_t)(uintptr_t)token;
  const uint64_t idx = (t & 0xffffffffull);
  const uint32_t gen = (uint32_t)(t >> 32);
  BasesHandle* h = nullptr;
  pthread_mutex_lock(&g_handle_mu);
  if (idx >= 1 && idx <= g_handles.size() && g_handles[idx - 1].h && g_handles[idx - 1].gen == gen) {
    h = g_handles[idx - 1].h;
    h->refs++;
    if (unpublish) {
      h->magic = 0;  // dead: freed by whoever drops the last reference
      g_handles[idx - 1].h = nullptr;
      if (++g_handles[idx - 1].gen == 0) g_handles[idx - 1].gen = 1;
    }
  }
  pthread_mutex_unlock(&g_handle_mu);
  return h;
}
void handle_unpin(BasesHandle* h) {
  pthread_mutex_lock(&g_handle_mu);
  const bool last = --h->refs == 0 && h->magic != BASES_MAGIC;
  pthread_mutex_unlock(&g_handle_mu);
  if (last) bases_free(h);
}
}  // namespace

int ozk_bases_create_host(const uint8_t* bases, int32_t n, int32_t type, int32_t task_id, void** handle) {
  if (!bases || !handle) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  BasesHandle* h = nullptr;
  const int rc = type == OZK_G1 ? bases_create<G1Cfg>(bases, n, type, task_id, &h)
                                : bases_create<G2Cfg>(bases, n, type, task_id, &h);
  if (rc) return rc;
  try {
    *handle = handle_publish(h);
  } catch (const std::exception&) {
    bases_free(h);
    return fail(OZK_E_NOMEM, "out of host memory");
  }
  return OZK_OK;
}
int ozk_var_msm_bases_host(void* handle, const uint8_t* scalars, int32_t n, uint8_t* out) {
  if (!handle || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  BasesHandle* h = handle_pin(handle);
  if (!h) return fail(OZK_E_INVALID, "not a live bases handle (stale or already released)");
  int rc;
  if (n != h->n) rc = fail(OZK_E_INVALID, "batch_size %d does not match the prepared bases (%d)", n, h->n);
  else rc = h->type == OZK_G1 ? bases_msm<G1Cfg>(h, scalars, out) : bases_msm<G2Cfg>(h, scalars, out);
  handle_unpin(h);
  return rc;
}
int ozk_bases_type(void* handle) {
  BasesHandle* h = handle ? handle_pin(handle) : nullptr;
  if (!h) return 0;
  const int t = h->type;
  handle_unpin(h);
  return t;
}
int ozk_bases_destroy(void* handle) {
  if (!handle) return OZK_OK;
  BasesHandle* h = handle_pin(handle, true);
  if (!h) return fail(OZK_E_INVALID, "not a live bases handle (stale or already released)");
  handle_unpin(h);  // frees now, or when the MSM in flight on it returns
  return OZK_OK;
}

int ozk_tuning_reload(void) {
  env_reload();
  return OZK_OK;
}

int ozk_order_event_create(void** ev) {
  if (!ev) return fail(OZK_E_INVALID, "null pointer argument");
  hipEvent_t e = nullptr;
  OZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *ev = (void*)e;
  return OZK_OK;
}
int ozk_order_event_destroy(void* ev) {
  if (ev) OZK_HIP(hipEventDestroy((hipEvent_t)ev));
  return OZK_OK;
}
int ozk_var_msm_head_ordered_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type,
                                 void* d_workspace, size_t workspace_bytes, void* d_tail, size_t tail_bytes,
                                 void* stream, void* previous_levels_done) {
  if (!d_bases || !d_scalars || !d_workspace || !d_tail) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_head<G1Cfg>(d_bases, d_scalars, n, d_workspace, workspace_bytes, d_tail, tail_bytes,
                               (hipStream_t)stream, (hipEvent_t)previous_levels_done);
  return var_msm_head<G2Cfg>(d_bases, d_scalars, n, d_workspace, workspace_bytes, d_tail, tail_bytes,
                             (hipStream_t)stream, (hipEvent_t)previous_levels_done);
}
int ozk_var_msm_tail_ordered_dev(int32_t n, int32_t type, void* d_tail, size_t tail_bytes, void* d_out, void* stream,
                                 void* levels_done) {
  if (!d_tail || !d_out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_tail<G1Cfg>(n, d_tail, tail_bytes, d_out, (hipStream_t)stream, (hipEvent_t)levels_done);
  return var_msm_tail<G2Cfg>(n, d_tail, tail_bytes, d_out, (hipStream_t)stream, (hipEvent_t)levels_done);
}
// the tail with the shape of its window sums chosen by the caller: mode 0 = latency (a lone MSM), 1 = throughput
// (the caller keeps the chip busy with other work: a prover with five MSMs and a witness map in flight)
int ozk_var_msm_tail_mode_dev(int32_t n, int32_t type, void* d_tail, size_t tail_bytes, void* d_out, void* stream,
                              void* levels_done, int32_t mode) {
  if (!d_tail || !d_out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  const int m = mode ? TAIL_THROUGHPUT : TAIL_LATENCY;
  if (type == OZK_G1)
    return var_msm_tail<G1Cfg>(n, d_tail, tail_bytes, d_out, (hipStream_t)stream, (hipEvent_t)levels_done, m);
  return var_msm_tail<G2Cfg>(n, d_tail, tail_bytes, d_out, (hipStream_t)stream, (hipEvent_t)levels_done, m);
}
int ozk_var_msm_stage_bytes(int32_t n, int32_t type, size_t* sorted_bytes, size_t* sort_ws_bytes,
                            size_t* accum_ws_bytes) {
  if (n <= 0 || n > (1 << 24) || !sorted_bytes || !sort_ws_bytes || !accum_ws_bytes)
    return fail(OZK_E_INVALID, "bad argument");
  const RegionBytes rb = type == OZK_G1 ? region_bytes<G1Cfg>(n) : region_bytes<G2Cfg>(n);
  *sorted_bytes = rb.sorted;
  *sort_ws_bytes = rb.sort_ws;
  *accum_ws_bytes = rb.accum_ws;
  return OZK_OK;
}
int ozk_var_msm_sort_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type, void* d_sorted,
                         size_t sorted_bytes, void* d_sort_ws, size_t sort_ws_bytes, void* stream) {
  if (!d_bases || !d_scalars || !d_sorted || !d_sort_ws) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_sort<G1Cfg>(d_bases, d_scalars, n, d_sorted, sorted_bytes, d_sort_ws, sort_ws_bytes,
                               (hipStream_t)stream);
  return var_msm_sort<G2Cfg>(d_bases, d_scalars, n, d_sorted, sorted_bytes, d_sort_ws, sort_ws_bytes,
                             (hipStream_t)stream);
}
int ozk_var_msm_accum_dev(int32_t n, int32_t type, void* d_sorted, size_t sorted_bytes, void* d_accum_ws,
                          size_t accum_ws_bytes, void* d_tail, size_t tail_bytes, void* stream) {
  if (!d_sorted || !d_accum_ws || !d_tail) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_accum<G1Cfg>(n, d_sorted, sorted_bytes, d_accum_ws, accum_ws_bytes, d_tail, tail_bytes,
                                (hipStream_t)stream);
  return var_msm_accum<G2Cfg>(n, d_sorted, sorted_bytes, d_accum_ws, accum_ws_bytes, d_tail, tail_bytes,
                              (hipStream_t)stream);
}
int ozk_var_msm_accum_part_dev(const void* d_prepared, int32_t n, int32_t type, void* d_sorted, size_t sorted_bytes,
                               void* d_accum_ws, size_t accum_ws_bytes, void* d_tail, size_t tail_bytes, void* stream,
                               int32_t part) {
  if (!d_sorted || !d_accum_ws || !d_tail) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (part < ACCUM_ALL || part > ACCUM_REST) return fail(OZK_E_INVALID, "part %d is not 0 (all), 1 (level 1) or 2 (rest)", part);
  if (type == OZK_G1)
    return var_msm_accum<G1Cfg>(n, d_sorted, sorted_bytes, d_accum_ws, accum_ws_bytes, d_tail, tail_bytes,
                                (hipStream_t)stream, d_prepared, part);
  return var_msm_accum<G2Cfg>(n, d_sorted, sorted_bytes, d_accum_ws, accum_ws_bytes, d_tail, tail_bytes,
                              (hipStream_t)stream, d_prepared, part);
}
int ozk_var_msm_sort_prepared_dev(const void* d_prepared, const void* d_scalars, int32_t n, int32_t type, void* d_sorted,
                                  size_t sorted_bytes, void* d_sort_ws, size_t sort_ws_bytes, void* stream) {
  if (!d_prepared || !d_scalars || !d_sorted || !d_sort_ws) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_sort<G1Cfg>(nullptr, d_scalars, n, d_sorted, sorted_bytes, d_sort_ws, sort_ws_bytes,
                               (hipStream_t)stream, nullptr, d_prepared);
  return var_msm_sort<G2Cfg>(nullptr, d_scalars, n, d_sorted, sorted_bytes, d_sort_ws, sort_ws_bytes,
                             (hipStream_t)stream, nullptr, d_prepared);
}
int ozk_var_msm_accum_prepared_dev(const void* d_prepared, int32_t n, int32_t type, void* d_sorted, size_t sorted_bytes,
                                   void* d_accum_ws, size_t accum_ws_bytes, void* d_tail, size_t tail_bytes,
                                   void* stream) {
  if (!d_prepared || !d_sorted || !d_accum_ws || !d_tail) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1)
    return var_msm_accum<G1Cfg>(n, d_sorted, sorted_bytes, d_accum_ws, accum_ws_bytes, d_tail, tail_bytes,
                                (hipStream_t)stream, d_prepared);
  return var_msm_accum<G2Cfg>(n, d_sorted, sorted_bytes, d_accum_ws, accum_ws_bytes, d_tail, tail_bytes,
                              (hipStream_t)stream, d_prepared);
}
int ozk_var_msm_tail_dev(int32_t n, int32_t type, void* d_tail, size_t tail_bytes, void* d_out, void* stream) {
  if (!d_tail || !d_out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1) return var_msm_tail<G1Cfg>(n, d_tail, tail_bytes, d_out, (hipStream_t)stream, nullptr, TAIL_THROUGHPUT);
  return var_msm_tail<G2Cfg>(n, d_tail, tail_bytes, d_out, (hipStream_t)stream, nullptr, TAIL_THROUGHPUT);
}

int ozk_var_msm_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type, int32_t task_id,
                     uint8_t* out) {
  if (!bases || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  if (type == OZK_G1) return var_msm_host<G1Cfg>(bases, scalars, n, task_id, out);
#if defined(OZK_WITH_G2)
  return var_msm_host<G2Cfg>(bases, scalars, n, task_id, out);
#else
  return fail(OZK_E_INVALID, "G2 not built");
#endif
}

// d_result != nullptr (the sharded entry's RCCL form): the 576 bytes stay on the device, G1 (192) || G2 (384)
static int var_double_msm_host_impl(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars,
                                    int32_t n, int32_t task_id, uint8_t* out, uint8_t* d_result) {
  // G1 and G2 over the same scalars (VariableBaseMSM.cu:1772-1773); out = G1 (192) || G2 (384).
  // The reference runs them back to back, each with its own uploads.  Here the scalars go up once, and
  // the 192 n bytes of G2 bases are uploaded (the host thread staging pageable memory) while the G1 MSM
  // already runs on its own stream; the G1 tail then overlaps the G2 head.
  if (!bases_g1 || !bases_g2 || !scalars || (!out && !d_result)) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  CtxGuard g;
  int rc = ctx_acquire(task_id, &g.c);
  // both results out: to the host through the pinned result buffer, or device to device
  auto finish = [&](HostCtx* c, uint8_t* d_out, hipStream_t s1, hipStream_t s2) -> int {
    int r;
    if (d_result) {
      OZK_HIP(hipMemcpyAsync(d_result, d_out, 192, hipMemcpyDeviceToDevice, s1));
      OZK_HIP(hipMemcpyAsync(d_result + 192, d_out + 256, 384, hipMemcpyDeviceToDevice, s2));
    } else {
      if ((r = small_d2h_begin(c, 0, d_out, 192, s1))) return r;
      if ((r = small_d2h_begin(c, 1024, d_out + 256, 384, s2))) return r;
    }
    OZK_HIP(hipStreamSynchronize(s1));
    OZK_HIP(hipStreamSynchronize(s2));
    if (!d_result) {
      small_d2h_end(c, 0, out, 192);
      small_d2h_end(c, 1024, out + 192, 384);
    }
    return OZK_OK;
  };
  if (rc) return rc;
  HostCtx* c = g.c;
  const int K = host_slices(n);
  if (K > 1 && 2 * K <= MAX_SLICES) {
    // Large call: the scalars go up once, then the G1 bases and the G2 bases slice by slice; each slice's sort +
    // bucket accumulation is queued behind its upload (G1 on the first stream, G2 on the second), one tail per
    // curve (var_msm_host's form) — the 288 n bytes of bases hide most of both accumulations.
    const int per = host_slice_per(n, K);
    const size_t padded = (size_t)K * per;
    const size_t w1 = host_sliced_ws_bytes<G1Cfg>(K, per), w2 = host_sliced_ws_bytes<G2Cfg>(K, per);
    if ((rc = ctx_reserve(c, pad256(padded * 32) + pad256(padded * 96) + pad256(padded * 192) + 1024 + w1 + w2 + 1024)))
      return rc;
    uint8_t* d_sc = c->arena;
    uint8_t* d_b1 = d_sc + pad256(padded * 32);
    uint8_t* d_b2 = d_b1 + pad256(padded * 96);
    uint8_t* d_out = d_b2 + pad256(padded * 192);
    uint8_t* d_w1 = d_out + 1024;
    uint8_t* d_w2 = d_w1 + w1;
    hipStream_t up = c->st[2];
    if (padded > (size_t)n) OZK_HIP(hipMemsetAsync(d_sc + (size_t)n * 32, 0, (padded - n) * 32, up));
    if ((rc = staged_h2d(c, d_sc, scalars, (size_t)n * 32, up))) return rc;
    if ((rc = host_sliced_msm<G1Cfg>(c, bases_g1, nullptr, n, K, per, d_b1, d_sc, d_w1, d_out, c->slice_ev, c->st[0], up)))
      return rc;
    if ((rc = host_sliced_msm<G2Cfg>(c, bases_g2, nullptr, n, K, per, d_b2, d_sc, d_w2, d_out + 256, c->slice_ev + K,
                                     c->st[1], up)))
      return rc;
    return finish(c, d_out, c->st[0], c->st[1]);
  }
  const size_t b1 = (size_t)n * 96, b2 = (size_t)n * 192, sc = (size_t)n * 32;
  const size_t w1 = var_msm_ws_bytes<G1Cfg>(n), w2 = var_msm_ws_bytes<G2Cfg>(n);
  if ((rc = ctx_reserve(c, pad256(b1) + pad256(b2) + pad256(sc) + 1024 + pad256(w1) + pad256(w2) + 1024))) return rc;
  uint8_t* d_b1 = c->arena;
  uint8_t* d_b2 = d_b1 + pad256(b1);
  uint8_t* d_sc = d_b2 + pad256(b2);
  uint8_t* d_out = d_sc + pad256(sc);
  uint8_t* d_w1 = d_out + 1024;
  uint8_t* d_w2 = d_w1 + pad256(w1);
  hipStream_t s1 = c->st[0], s2 = c->st[1];
  // scalars and G1 bases on s1, G1 MSM on s1; the G2 bases follow on s2 (the host thread staging them while
  // the G1 MSM runs) and the G2 MSM waits only for the scalars
  if ((rc = staged_h2d(c, d_sc, scalars, sc, s1))) return rc;
  OZK_HIP(hipEventRecord(c->ev[0], s1));
  if ((rc = staged_h2d(c, d_b1, bases_g1, b1, s1))) return rc;
  if ((rc = var_msm_dev<G1Cfg>(d_b1, d_sc, n, d_out, d_w1, w1, s1))) return rc;
  OZK_HIP(hipStreamWaitEvent(s2, c->ev[0], 0));
  if ((rc = staged_h2d(c, d_b2, bases_g2, b2, s2))) return rc;
  if ((rc = var_msm_dev<G2Cfg>(d_b2, d_sc, n, d_out + 256, d_w2, w2, s2))) return rc;
  return finish(c, d_out, s1, s2);
}
int ozk_var_double_msm_host(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars,
                            int32_t n, int32_t task_id, uint8_t* out) {
  if (!out) return fail(OZK_E_INVALID, "null pointer argument");
  return var_double_msm_host_impl(bases_g1, bases_g2, scalars, n, task_id, out, nullptr);
}

// In-process multi-GPU MSM for ONE caller (a serial Java prover calls the native once, with taskID 0: the
// reference then uses one GPU, algebra_msm_VariableBaseMSM.cu:1249-1257; its multi-GPU form needs Spark
// partitions, VariableBaseMSM.java:775-786 / :805-818 for the double MSM).  The (scalar, base) index range is cut
// into `shards` contiguous slices (shards <= 0: one per visible device); slice i runs the whole single-GPU pipeline
// on device i % device_count from its own host thread and context.  Then the exchange north_star names — an
// "all-reduce" of the partial accumulators, which for a group law RCCL does not know is an ALL-GATHER of the
// affine partials plus a local point sum (the reduce(GroupT::add) of VariableBaseMSM.java:783):
//   RCCL form (default when librccl loads and the communicators come up): every slice leaves its 192 / 384 / 576-byte
//     partial in a send buffer on ITS device; one ncclAllGather per device (a group call over communicators made
//     once per process by ncclCommInitAll) moves them over xGMI; k_points_sum runs on device 0 over the gathered
//     records.  The partials never visit the host.
//   host form (OZK_SHARD_RCCL=0, or RCCL unavailable / failing to initialise — e.g. a JVM host without librccl):
//     partials back through the host, sum on device 0.  Round 2-3's only form.
// Both give the same bytes (tests/test_sharded_gpu.py runs both on one device: ncclCommInitAll(ndev = 1)).  The
// exchange is <= 64 x 576 B: latency-bound either way; nothing of it has been timed on more than one device.
namespace {
// ---- RCCL, bound at run time: libozk_hip.so must load in a JVM whose host has no librccl
struct RcclApi {
  void* lib = nullptr;
  bool tried = false, ok = false;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
constexpr size_t SHARD_REC = 576;        // room for G1 (192) || G2 (384); single-curve calls use the first 192 / 384
constexpr int SHARD_MAX = 64;
struct ShardComm {   // the communicators over devices 0 .. D-1 and their buffers, made once per process and D
  int D = 0;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> st;
  std::vector<uint8_t*> d_send, d_recv;   // per device: SHARD_MAX records out, D * SHARD_MAX records in
  uint8_t* d_sum = nullptr;               // device 0: 1 KiB for the summed point(s)
  uint8_t* h_pin = nullptr;               // pinned host: infinity records in, result out
};
pthread_mutex_t g_shard_mu = PTHREAD_MUTEX_INITIALIZER;   // one sharded call at a time owns the communicators
RcclApi g_rccl;
std::vector<ShardComm*> g_shard_comms;
thread_local int g_last_exchange = -1;

bool rccl_load() {   // (g_shard_mu held)
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) return false;
  auto sym = [&](const char* n) { return dlsym(g_rccl.lib, n); };
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd &&
              g_rccl.GetErrorString;
  return g_rccl.ok;
}
void shard_comm_free(ShardComm* sc) {
  for (int r = 0; r < (int)sc->comm.size(); r++)
    if (sc->comm[r]) g_rccl.CommDestroy(sc->comm[r]);
  for (int r = 0; r < sc->D; r++) {
    if (hipSetDevice(r) != hipSuccess) continue;
    if (r < (int)sc->st.size() && sc->st[r]) (void)hipStreamDestroy(sc->st[r]);
    if (r < (int)sc->d_send.size() && sc->d_send[r]) (void)hipFree(sc->d_send[r]);
    if (r < (int)sc->d_recv.size() && sc->d_recv[r]) (void)hipFree(sc->d_recv[r]);
  }
  if (sc->d_sum && hipSetDevice(0) == hipSuccess) (void)hipFree(sc->d_sum);
  if (sc->h_pin) (void)hipHostFree(sc->h_pin);
  delete sc;
}
// the communicator set over D devices, or nullptr (-> host form).  (g_shard_mu held)
ShardComm* shard_comm_get(int D) {
  for (ShardComm* sc : g_shard_comms)
    if (sc->D == D) return sc;
  if (!rccl_load()) return nullptr;
  ShardComm* sc = new (std::nothrow) ShardComm();
  if (!sc) return nullptr;
  bool ok = true;
  try {
    sc->D = D;
    sc->comm.assign(D, nullptr);
    sc->st.assign(D, nullptr);
    sc->d_send.assign(D, nullptr);
    sc->d_recv.assign(D, nullptr);
    std::vector<int> devs(D);
    for (int r = 0; r < D; r++) devs[r] = r;
    for (int r = 0; r < D && ok; r++) {
      ok = hipSetDevice(r) == hipSuccess && hipStreamCreateWithFlags(&sc->st[r], hipStreamNonBlocking) == hipSuccess &&
           hipMalloc((void**)&sc->d_send[r], SHARD_MAX * SHARD_REC) == hipSuccess &&
           hipMalloc((void**)&sc->d_recv[r], (size_t)D * SHARD_MAX * SHARD_REC) == hipSuccess;
    }
    ok = ok && hipSetDevice(0) == hipSuccess && hipMalloc((void**)&sc->d_sum, 1024) == hipSuccess &&
         hipHostMalloc((void**)&sc->h_pin, SHARD_MAX * SHARD_REC + 1024, hipHostMallocPortable) == hipSuccess;   // (read by every device)
    if (ok) {
      const ncclResult_t r = g_rccl.CommInitAll(sc->comm.data(), D, devs.data());
      if (r != ncclSuccess) {
        if (env_int("OZK_HOST_TRACE", 0)) fprintf(stderr, "[ozk] ncclCommInitAll(%d) failed: %s\n", D, g_rccl.GetErrorString(r));
        ok = false;
      }
    }
    if (ok) g_shard_comms.push_back(sc);
  } catch (const std::exception&) {
    ok = false;
  }
  if (!ok) {
    shard_comm_free(sc);
    return nullptr;
  }
  return sc;
}
// wire-out infinity (0, 1, 0) records for the slots no slice fills: kind 1 = G1, 2 = G2, 3 = G1 || G2
void fill_infinity(uint8_t* rec, int kind) {
  memset(rec, 0, SHARD_REC);
  if (kind == 1) rec[64] = 1;
  else if (kind == 2) rec[128] = 1;
  else {
    rec[64] = 1;
    rec[192 + 128] = 1;
  }
}
}  // namespace

static int points_sum_strided(const void* d_points, int k, int type, size_t stride_bytes, void* d_out, hipStream_t st);

// kind: 1 = G1 MSM, 2 = G2 MSM, 3 = double MSM (G1 || G2 over the same scalars)
static int var_msm_shard(const uint8_t* b1, const uint8_t* b2, const uint8_t* scalars, int n, int kind, int task_id,
                         uint8_t* out, uint8_t* d_result) {
  if (kind == 1) return var_msm_host<G1Cfg>(b1, scalars, n, task_id, out, d_result);
  if (kind == 2) return var_msm_host<G2Cfg>(b2, scalars, n, task_id, out, d_result);
  return var_double_msm_host_impl(b1, b2, scalars, n, task_id, out, d_result);
}

static int var_msm_sharded(const uint8_t* b1, const uint8_t* b2, const uint8_t* scalars, int32_t n, int kind, int32_t shards,
                           uint8_t* out) {
  if (n <= 0) return fail(OZK_E_INVALID, "batch_size %d out of range", n);
  const int ndev = ozk_device_count();
  if (ndev <= 0) return fail(OZK_E_NO_DEVICE, "no HIP device available; this library has no CPU path");
  int k = shards > 0 ? shards : ndev;
  if (k > n) k = n;
  if (k > SHARD_MAX) k = SHARD_MAX;
  const size_t ob = kind == 1 ? 192 : kind == 2 ? 384 : 576;
  g_last_exchange = -1;
  if (k == 1) return var_msm_shard(b1, b2, scalars, n, kind, 0, out, nullptr);
  const int D = k < ndev ? k : ndev;             // devices in use: slice i on device i % ndev
  const int slots = (k + D - 1) / D;             // slices per device
  // the RCCL form needs the process's communicator set; a second sharded call that arrives while one is in flight
  // takes the host form rather than wait (trylock)
  ShardComm* sc = nullptr;
  bool have_mu = false;
  if (env_int("OZK_SHARD_RCCL", 1) && pthread_mutex_trylock(&g_shard_mu) == 0) {
    have_mu = true;
    sc = shard_comm_get(D);
  }
  struct Unlock {
    bool on;
    ~Unlock() {
      if (on) pthread_mutex_unlock(&g_shard_mu);
    }
  } unlock{have_mu};
  if (sc) {   // every slot an infinity record; the slices overwrite theirs
    for (int j = 0; j < slots; j++) fill_infinity(sc->h_pin + (size_t)j * SHARD_REC, kind);
    for (int r = 0; r < D; r++) {
      OZK_HIP(hipSetDevice(r));
      OZK_HIP(hipMemcpyAsync(sc->d_send[r], sc->h_pin, (size_t)slots * SHARD_REC, hipMemcpyHostToDevice, sc->st[r]));
      OZK_HIP(hipStreamSynchronize(sc->st[r]));
    }
  }
  // (nothing may throw through the extern "C" boundary into a JVM: allocation and thread-start failures become codes)
  std::vector<uint8_t> partial;
  std::vector<int> rcs;
  std::vector<std::string> msgs;
  std::vector<std::thread> th;
  const int base_n = n / k, rem = n % k;
  int started = 0;
  bool spawn_failed = false;
  try {
    partial.resize((size_t)k * ob);
    rcs.assign(k, OZK_OK);
    msgs.resize(k);
    th.reserve(k);
    for (int i = 0; i < k; i++) {
      const size_t lo = (size_t)i * base_n + (size_t)(i < rem ? i : rem);
      const int cnt = base_n + (i < rem ? 1 : 0);
      uint8_t* d_res = sc ? sc->d_send[i % ndev] + (size_t)(i / ndev) * SHARD_REC : nullptr;
      th.emplace_back([&, i, lo, cnt, d_res] {
        rcs[i] = var_msm_shard(b1 ? b1 + lo * 96 : nullptr, b2 ? b2 + lo * 192 : nullptr, scalars + lo * 32, cnt, kind, i,
                               partial.data() + (size_t)i * ob, d_res);
        if (rcs[i]) msgs[i] = err_buf();   // the message lives in that thread's buffer
      });
      started++;
    }
  } catch (const std::exception&) {  // std::system_error from the thread constructor, std::bad_alloc
    spawn_failed = true;
  }
  for (auto& t : th)
    if (t.joinable()) t.join();
  if (spawn_failed) return fail(OZK_E_NOMEM, "sharded MSM: could not start shard %d of %d (threads / host memory)", started, k);
  for (int i = 0; i < k; i++)
    if (rcs[i]) return fail(rcs[i], "shard %d of %d: %s", i, k, msgs[i].c_str());
  if (sc) {
    // all-gather of every device's `slots` records (one group call from this thread: the communicators of
    // ncclCommInitAll), then the point sum on device 0
    const size_t send = (size_t)slots * SHARD_REC;
    ncclResult_t nr = g_rccl.GroupStart();
    for (int r = 0; r < D && nr == ncclSuccess; r++)
      nr = g_rccl.AllGather(sc->d_send[r], sc->d_recv[r], send, ncclUint8, sc->comm[r], sc->st[r]);
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return fail(OZK_E_NO_DEVICE, "RCCL all-gather of the partial results failed: %s", g_rccl.GetErrorString(nr));
    OZK_HIP(hipSetDevice(0));
    for (int r = 1; r < D; r++) {   // (only device 0's copy is read; the other ranks' streams just drain)
      OZK_HIP(hipSetDevice(r));
      OZK_HIP(hipStreamSynchronize(sc->st[r]));
    }
    OZK_HIP(hipSetDevice(0));
    int rc;
    const int recs = D * slots;
    if (kind == 3) {
      if ((rc = points_sum_strided(sc->d_recv[0], recs, OZK_G1, SHARD_REC, sc->d_sum, sc->st[0]))) return rc;
      if ((rc = points_sum_strided(sc->d_recv[0] + 192, recs, OZK_G2, SHARD_REC, sc->d_sum + 192, sc->st[0]))) return rc;
    } else {
      if ((rc = points_sum_strided(sc->d_recv[0], recs, kind == 1 ? OZK_G1 : OZK_G2, SHARD_REC, sc->d_sum, sc->st[0]))) return rc;
    }
    OZK_HIP(hipMemcpyAsync(sc->h_pin, sc->d_sum, ob, hipMemcpyDeviceToHost, sc->st[0]));
    OZK_HIP(hipStreamSynchronize(sc->st[0]));
    memcpy(out, sc->h_pin, ob);
    g_last_exchange = 1;
    return OZK_OK;
  }
  // host form: sum of the partials on device 0
  CtxGuard g;
  int rc = ctx_acquire(0, &g.c);
  if (rc) return rc;
  HostCtx* c = g.c;
  if ((rc = ctx_reserve(c, pad256(partial.size()) + 2048))) return rc;
  // (through the pinned result buffer both ways: host_ctx.h, small_d2h_begin)
  if (partial.size() + 1024 > RESULT_BYTES) return fail(OZK_E_INTERNAL, "partials do not fit the pinned result buffer");
  memcpy(c->result + 1024, partial.data(), partial.size());
  OZK_HIP(hipMemcpyAsync(c->arena, c->result + 1024, partial.size(), hipMemcpyHostToDevice, c->st[0]));
  uint8_t* d_out = c->arena + pad256(partial.size());
  if (kind == 3) {
    if ((rc = points_sum_strided(c->arena, k, OZK_G1, 576, d_out, c->st[0]))) return rc;
    if ((rc = points_sum_strided(c->arena + 192, k, OZK_G2, 576, d_out + 192, c->st[0]))) return rc;
  } else {
    if ((rc = points_sum_strided(c->arena, k, kind == 1 ? OZK_G1 : OZK_G2, ob, d_out, c->st[0]))) return rc;
  }
  if ((rc = small_d2h_begin(c, 0, d_out, ob, c->st[0]))) return rc;
  OZK_HIP(hipStreamSynchronize(c->st[0]));
  small_d2h_end(c, 0, out, ob);
  g_last_exchange = 0;
  return OZK_OK;
}

int ozk_var_msm_sharded_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type, int32_t shards,
                             uint8_t* out) {
  if (!bases || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (type == OZK_G1) return var_msm_sharded(bases, nullptr, scalars, n, 1, shards, out);
  return var_msm_sharded(nullptr, bases, scalars, n, 2, shards, out);
}
// the double MSM the same way (VariableBaseMSM.distributedDoubleMSM, VariableBaseMSM.java:805-818: per-partition
// doubleMSM + reduce): out = 576 B, G1 (192) || G2 (384)
int ozk_var_double_msm_sharded_host(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars, int32_t n,
                                    int32_t shards, uint8_t* out) {
  if (!bases_g1 || !bases_g2 || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  return var_msm_sharded(bases_g1, bases_g2, scalars, n, 3, shards, out);
}
// how the calling thread's last sharded call exchanged its partials: 1 = RCCL all-gather, 0 = through the host,
// -1 = no exchange (one shard, or no sharded call yet)
int ozk_shard_last_exchange(void) { return g_last_exchange; }
// drops the process's RCCL communicators and their buffers (ozk_host_cache_release calls it)
void ozk_shard_comms_release(void) {
  pthread_mutex_lock(&g_shard_mu);
  for (ShardComm* sc : g_shard_comms) shard_comm_free(sc);
  g_shard_comms.clear();
  pthread_mutex_unlock(&g_shard_mu);
}

// What the JNI native calls: ONE GPU, taskID % count, as the reference (algebra_msm_VariableBaseMSM.cu:1249-1257) —
// Spark runs one task per partition concurrently, so T task threads already cover the devices, and spreading every
// call over all of them would multiply contexts and arenas by the device count.  A serial Java prover (one caller,
// taskID 0) opts in to spreading its large calls with OZK_SHARD=1 (calls of at least OZK_SHARD_MIN_N pairs,
// default 2^21, over OZK_SHARD_COUNT devices, default all).  The multi-device form has only ever run on a one-GPU
// box (all slices on device 0): unverified on multi-GPU hardware.
int ozk_var_msm_auto_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type, int32_t task_id,
                          uint8_t* out) {
  if (env_int("OZK_SHARD", 0) && n >= env_int("OZK_SHARD_MIN_N", 1 << 21) && ozk_device_count() > 1)
    return ozk_var_msm_sharded_host(bases, scalars, n, type, env_int("OZK_SHARD_COUNT", 0), out);
  return ozk_var_msm_host(bases, scalars, n, type, task_id, out);
}
int ozk_var_double_msm_auto_host(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars, int32_t n,
                                 int32_t task_id, uint8_t* out) {
  if (env_int("OZK_SHARD", 0) && n >= env_int("OZK_SHARD_MIN_N", 1 << 21) && ozk_device_count() > 1)
    return ozk_var_double_msm_sharded_host(bases_g1, bases_g2, scalars, n, env_int("OZK_SHARD_COUNT", 0), out);
  return ozk_var_double_msm_host(bases_g1, bases_g2, scalars, n, task_id, out);
}

static int points_sum_strided(const void* d_points, int k, int type, size_t stride_bytes, void* d_out, hipStream_t st) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!d_points || !d_out || k <= 0 || (stride_bytes & 3)) return fail(OZK_E_INVALID, "bad argument");
  if (type == OZK_G1) {
    hipLaunchKernelGGL((k_points_sum<G1Cfg>), dim3(1), dim3(64), 0, st, (const u32*)d_points, k, (int)(stride_bytes / 4),
                       (u32*)d_out);
  } else {
#if defined(OZK_WITH_G2)
    hipLaunchKernelGGL((k_points_sum<G2Cfg>), dim3(1), dim3(64), 0, st, (const u32*)d_points, k, (int)(stride_bytes / 4),
                       (u32*)d_out);
#else
    return fail(OZK_E_INVALID, "G2 not built");
#endif
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}
int ozk_points_sum_dev(const void* d_points, int32_t k, int32_t type, void* d_out, void* stream) {
  return points_sum_strided(d_points, k, type, type == OZK_G1 ? 192 : 384, d_out, (hipStream_t)stream);
}

int ozk_gen_bases_dev(uint64_t seed, int32_t n, int32_t type, void* d_out_wire, void* stream) {
  hip_clear_stale();   // (ozk_common.h: a stale error of the calling thread is not this call's)
  if (!d_out_wire || n <= 0) return fail(OZK_E_INVALID, "bad argument");
  if (type != OZK_G1) return fail(OZK_E_INVALID, "only G1 synthetic bases are generated");
  // generator (1, 2) (BN254aG1Parameters.java:23-24), wire format
  u32 gen[16];
  memset(gen, 0, sizeof(gen));
  gen[0] = 1;
  gen[8] = 2;
  u32* d_gen = nullptr;
  OZK_HIP(hipMalloc((void**)&d_gen, sizeof(gen)));
  OZK_HIP(hipMemcpyAsync(d_gen, gen, sizeof(gen), hipMemcpyHostToDevice, (hipStream_t)stream));
  hipLaunchKernelGGL((k_gen_bases<G1Cfg>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, n,
                     d_gen, (u32*)d_out_wire);
  OZK_HIP(hipGetLastError());
  OZK_HIP(hipStreamSynchronize((hipStream_t)stream));
  OZK_HIP(hipFree(d_gen));
  return OZK_OK;
}

// Enable / disable the timing of level-1 launches.  Single-device by construction: events and the clock buffer
// belong to the device that is current here, and only launches on that device are recorded (ProfState::claim).
// Serialised against concurrent launches by g_prof.mu; two threads enabling at once serialise too.
int ozk_prof_enable(int on) {
  struct Lock {
    Lock() { pthread_mutex_lock(&g_prof.mu); }
    ~Lock() { pthread_mutex_unlock(&g_prof.mu); }
  } lock;
  int dev = 0;
  OZK_HIP(hipGetDevice(&dev));
  if (on == PROF_CLOCK) {
    constexpr int CAP = 1 << 16;  // launches per enable
    g_prof.mode = PROF_OFF;       // (nothing claims a slot while the buffer is replaced / cleared)
    if (g_prof.d_clk && g_prof.clk_device != dev) {
      const int prev = g_prof.clk_device;
      if (hipSetDevice(prev) == hipSuccess) (void)hipFree(g_prof.d_clk);   // a buffer is freed on its own device
      OZK_HIP(hipSetDevice(dev));
      g_prof.d_clk = nullptr;
      g_prof.clk_khz = 0.0;   // the calibration is per device
    }
    if (!g_prof.d_clk) {
      OZK_HIP(hipMalloc((void**)&g_prof.d_clk, (size_t)CAP * 4 * sizeof(unsigned long long)));
      g_prof.clk_cap = CAP;
      g_prof.clk_device = dev;
    }
    if (g_prof.clk_khz <= 0.0) {
      // calibrate: two reads of the counter ~25 ms apart against std::chrono::steady_clock (each read is a
      // synchronous one-lane launch; its jitter of a few microseconds is < 0.1 % of the interval).  Measured on
      // MI355X: 100 011.8 kHz against the nominal 100 000 of hipDeviceAttributeWallClockRate.
      unsigned long long h[2] = {0, 0};
      std::chrono::steady_clock::time_point tp[2];
      for (int k = 0; k < 2; k++) {
        if (k) std::this_thread::sleep_for(std::chrono::milliseconds(25));
        OZK_HIP(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_read_clock, dim3(1), dim3(1), 0, nullptr, g_prof.d_clk);
        OZK_HIP(hipDeviceSynchronize());
        tp[k] = std::chrono::steady_clock::now();
        OZK_HIP(hipMemcpy(&h[k], g_prof.d_clk, sizeof(h[k]), hipMemcpyDeviceToHost));
      }
      const double ms = std::chrono::duration<double, std::milli>(tp[1] - tp[0]).count();
      if (ms <= 0.0 || h[1] <= h[0]) return fail(OZK_E_INTERNAL, "device clock calibration failed");
      g_prof.clk_khz = (double)(h[1] - h[0]) / ms;
    }
    OZK_HIP(hipMemset(g_prof.d_clk, 0, (size_t)CAP * 4 * sizeof(unsigned long long)));
    g_prof.src = PROF_CLOCK;
    g_prof.count = 0;
    g_prof.mode = PROF_CLOCK;
    return OZK_OK;
  }
  if (on) {
    g_prof.mode = PROF_OFF;
    if (g_prof.created && g_prof.ev_device != dev) {   // events are per device: a pool of another device is dropped
      for (auto e : g_prof.e0) (void)hipEventDestroy(e);
      for (auto e : g_prof.e1) (void)hipEventDestroy(e);
      g_prof.e0.clear();
      g_prof.e1.clear();
      g_prof.created = false;
    }
    g_prof.count = 0;
    if (!g_prof.created) {
      hipEvent_t a, b;
      if (!g_prof.slot(&a, &b)) return fail(OZK_E_NOMEM, "cannot create profiling events");
      g_prof.created = true;
      g_prof.ev_device = dev;
    }
    g_prof.src = PROF_EVENTS;
    g_prof.every = on >= 16 ? on - 16 + 1 : 1;  // on = 16 + k: time every (k + 1)-th launch only
    g_prof.seen = 0;
    g_prof.mode = PROF_EVENTS;
  } else {
    g_prof.mode = PROF_OFF;  // (the recorded launches stay readable until the next enable)
  }
  return OZK_OK;
}

// stats[0..3] = mean, median, min, max duration (ms) of the level-1 launches recorded since the last ozk_prof_enable
int ozk_prof_dominant_kernel_stats(double* stats4, int* launches) {
  if (!stats4 || !launches) return fail(OZK_E_INVALID, "null pointer argument");
  std::vector<double> d;
  double tot = 0;
  struct Lock {   // (no launch claims a slot while the records are read)
    Lock() { pthread_mutex_lock(&g_prof.mu); }
    ~Lock() { pthread_mutex_unlock(&g_prof.mu); }
  } lock;
  if (g_prof.d_clk && g_prof.src == PROF_CLOCK) {
    int dev = -1;
    OZK_HIP(hipGetDevice(&dev));
    if (dev != g_prof.clk_device) return fail(OZK_E_INVALID, "profiling was enabled on device %d, the caller is on device %d", g_prof.clk_device, dev);
    const double khz = g_prof.clk_khz;
    if (khz <= 0.0) return fail(OZK_E_INTERNAL, "device clock not calibrated");
    OZK_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)g_prof.count * 4);
    if (g_prof.count)
      OZK_HIP(hipMemcpy(h.data(), g_prof.d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < g_prof.count; i++) {
      const unsigned long long t0 = ~h[4 * i], t1 = h[4 * i + 1];
      if (h[4 * i] == 0 || t1 < t0) continue;  // launch never ran
      const double ms = (double)(t1 - t0) / khz;
      d.push_back(ms);
      tot += ms;
    }
  } else {
    for (int i = 0; i < g_prof.count; i++) {
      float ms = 0;
      OZK_HIP(hipEventSynchronize(g_prof.e1[i]));
      OZK_HIP(hipEventElapsedTime(&ms, g_prof.e0[i], g_prof.e1[i]));
      d.push_back(ms);
      tot += ms;
    }
  }
  *launches = (int)d.size();
  stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0.0;
  if (d.empty()) return OZK_OK;
  std::sort(d.begin(), d.end());
  stats4[0] = tot / (double)d.size();
  stats4[1] = d[d.size() / 2];
  stats4[2] = d.front();
  stats4[3] = d.back();
  return OZK_OK;
}

int ozk_prof_dominant_kernel_ms(double* avg_ms, int* launches) {
  if (!avg_ms || !launches) return fail(OZK_E_INVALID, "null pointer argument");
  double st[4];
  const int rc = ozk_prof_dominant_kernel_stats(st, launches);
  if (rc) return rc;
  *avg_ms = st[0];
  return OZK_OK;
}

// stats[0..3] = mean, median, min, max over the launches recorded since ozk_prof_enable(2) of the SHADER clock (MHz)
// each launch ran at: shader-clock ticks / constant-rate ticks spent inside the kernel by the first wave of every
// workgroup (k_segreduce), times the calibrated constant rate
int ozk_prof_dominant_kernel_clock_mhz(double* stats4, int* launches) {
  if (!stats4 || !launches) return fail(OZK_E_INVALID, "null pointer argument");
  struct Lock {
    Lock() { pthread_mutex_lock(&g_prof.mu); }
    ~Lock() { pthread_mutex_unlock(&g_prof.mu); }
  } lock;
  *launches = 0;
  stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0.0;
  if (!g_prof.d_clk || g_prof.src != PROF_CLOCK || g_prof.clk_khz <= 0.0) return OZK_OK;
  int dev = -1;
  OZK_HIP(hipGetDevice(&dev));
  if (dev != g_prof.clk_device) return fail(OZK_E_INVALID, "profiling was enabled on device %d, the caller is on device %d", g_prof.clk_device, dev);
  OZK_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)g_prof.count * 4);
  if (g_prof.count) OZK_HIP(hipMemcpy(h.data(), g_prof.d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> m;
  double tot = 0;
  for (int i = 0; i < g_prof.count; i++) {
    if (h[4 * i + 3] == 0) continue;
    const double mhz = (double)h[4 * i + 2] / (double)h[4 * i + 3] * g_prof.clk_khz * 1e-3;
    m.push_back(mhz);
    tot += mhz;
  }
  *launches = (int)m.size();
  if (m.empty()) return OZK_OK;
  std::sort(m.begin(), m.end());
  stats4[0] = tot / (double)m.size();
  stats4[1] = m[m.size() / 2];
  stats4[2] = m.front();
  stats4[3] = m.back();
  return OZK_OK;
}

// ticks per millisecond of the device clock as calibrated by ozk_prof_enable(2) (0 before the first calibration)
double ozk_prof_clock_khz(void) { return g_prof.clk_khz; }

int ozk_var_msm_plan(int32_t n, int32_t* window_bits, int32_t* windows) {
  if (n <= 0 || !window_bits || !windows) return fail(OZK_E_INVALID, "bad argument");
  const MsmPlan p = make_plan(n);
  *window_bits = p.c;
  *windows = p.W;
  return OZK_OK;
}

int ozk_var_msm_glv(int32_t n) { return n > 0 ? make_plan(n).glv : 0; }

const char* ozk_last_error(void) { return err_buf(); }
int ozk_version(void) { return 1; }
int ozk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
