// Variable-base MSM: C ABI (include/ozk.h) and the G1 instantiation of the driver templates
// (msm_var_driver.cuh); the G2 instantiation is msm_var_g2.hip.
// Replaces pippengerMSMG1 / pippengerMSMG2 (algebra_msm_VariableBaseMSM.cu:1246-1604) and
// the two JNI natives of algebra.msm.VariableBaseMSM (.cu:1614-1788).
#include "msm_var_driver.cuh"

#include <algorithm>
#include <chrono>
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

int ozk_var_double_msm_host(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars,
                            int32_t n, int32_t task_id, uint8_t* out) {
  // G1 and G2 over the same scalars (VariableBaseMSM.cu:1772-1773); out = G1 (192) || G2 (384).
  // The reference runs them back to back, each with its own uploads.  Here the scalars go up once, and
  // the 192 n bytes of G2 bases are uploaded (the host thread staging pageable memory) while the G1 MSM
  // already runs on its own stream; the G1 tail then overlaps the G2 head.
  if (!bases_g1 || !bases_g2 || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0 || n > (1 << 24)) return fail(OZK_E_INVALID, "batch_size %d out of range [1, 2^24]", n);
  CtxGuard g;
  int rc = ctx_acquire(task_id, &g.c);
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
    if ((rc = small_d2h_begin(c, 0, d_out, 192, c->st[0]))) return rc;
    if ((rc = small_d2h_begin(c, 1024, d_out + 256, 384, c->st[1]))) return rc;
    OZK_HIP(hipStreamSynchronize(c->st[0]));
    OZK_HIP(hipStreamSynchronize(c->st[1]));
    small_d2h_end(c, 0, out, 192);
    small_d2h_end(c, 1024, out + 192, 384);
    return OZK_OK;
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
  if ((rc = small_d2h_begin(c, 0, d_out, 192, s1))) return rc;
  if ((rc = small_d2h_begin(c, 1024, d_out + 256, 384, s2))) return rc;
  OZK_HIP(hipStreamSynchronize(s1));
  OZK_HIP(hipStreamSynchronize(s2));
  small_d2h_end(c, 0, out, 192);
  small_d2h_end(c, 1024, out + 192, 384);
  return OZK_OK;
}

// In-process multi-GPU MSM for ONE caller (a serial Java prover calls the native once, with taskID 0: the
// reference then uses one GPU, algebra_msm_VariableBaseMSM.cu:1249-1257; its multi-GPU form needs Spark
// partitions, VariableBaseMSM.java:775-786).  The (scalar, base) index range is cut into `shards` contiguous
// slices (shards <= 0: one per visible device); slice i runs the whole single-GPU pipeline on device
// i % device_count from its own host thread and context; the 192 / 384-byte partials come back to the host
// and are added on device 0 (the reduce(GroupT::add) of VariableBaseMSM.java:783) — the exchange is
// shards x 192 B, so there is nothing for a collective library to do here (the one-process-per-GPU form,
// octopuszk_amd/distributed.py, all-gathers the same partials over RCCL).
static int var_msm_shard(const uint8_t* bases, const uint8_t* scalars, int n, int type, int task_id, uint8_t* out) {
  return type == OZK_G1 ? var_msm_host<G1Cfg>(bases, scalars, n, task_id, out)
                        : var_msm_host<G2Cfg>(bases, scalars, n, task_id, out);
}

int ozk_var_msm_sharded_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type, int32_t shards,
                             uint8_t* out) {
  if (!bases || !scalars || !out) return fail(OZK_E_INVALID, "null pointer argument");
  if (n <= 0) return fail(OZK_E_INVALID, "batch_size %d out of range", n);
  const int ndev = ozk_device_count();
  if (ndev <= 0) return fail(OZK_E_NO_DEVICE, "no HIP device available; this library has no CPU path");
  int k = shards > 0 ? shards : ndev;
  if (k > n) k = n;
  if (k > 64) k = 64;
  const size_t pt = type == OZK_G1 ? 96 : 192, ob = type == OZK_G1 ? 192 : 384;
  if (k == 1) return var_msm_shard(bases, scalars, n, type, 0, out);
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
      th.emplace_back([&, i, lo, cnt] {
        rcs[i] = var_msm_shard(bases + lo * pt, scalars + lo * 32, cnt, type, i, partial.data() + (size_t)i * ob);
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
  // sum of the partials on device 0
  CtxGuard g;
  int rc = ctx_acquire(0, &g.c);
  if (rc) return rc;
  HostCtx* c = g.c;
  if ((rc = ctx_reserve(c, pad256(partial.size()) + 1024))) return rc;
  // (through the pinned result buffer both ways: host_ctx.h, small_d2h_begin)
  if (partial.size() + 1024 > RESULT_BYTES) return fail(OZK_E_INTERNAL, "partials do not fit the pinned result buffer");
  memcpy(c->result + 1024, partial.data(), partial.size());
  OZK_HIP(hipMemcpyAsync(c->arena, c->result + 1024, partial.size(), hipMemcpyHostToDevice, c->st[0]));
  uint8_t* d_out = c->arena + pad256(partial.size());
  if ((rc = ozk_points_sum_dev(c->arena, k, type, d_out, c->st[0]))) return rc;
  if ((rc = small_d2h_begin(c, 0, d_out, ob, c->st[0]))) return rc;
  OZK_HIP(hipStreamSynchronize(c->st[0]));
  small_d2h_end(c, 0, out, ob);
  return OZK_OK;
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

int ozk_points_sum_dev(const void* d_points, int32_t k, int32_t type, void* d_out, void* stream) {
  if (!d_points || !d_out || k <= 0) return fail(OZK_E_INVALID, "bad argument");
  if (type == OZK_G1) {
    hipLaunchKernelGGL((k_points_sum<G1Cfg>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const u32*)d_points, k,
                       (u32*)d_out);
  } else {
#if defined(OZK_WITH_G2)
    hipLaunchKernelGGL((k_points_sum<G2Cfg>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const u32*)d_points, k,
                       (u32*)d_out);
#else
    return fail(OZK_E_INVALID, "G2 not built");
#endif
  }
  OZK_HIP(hipGetLastError());
  return OZK_OK;
}

int ozk_gen_bases_dev(uint64_t seed, int32_t n, int32_t type, void* d_out_wire, void* stream) {
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

int ozk_prof_enable(int on) {
  if (on == PROF_CLOCK) {
    constexpr int CAP = 1 << 16;  // launches per enable
    int dev = 0;
    OZK_HIP(hipGetDevice(&dev));
    if (g_prof.d_clk && g_prof.clk_device != dev) {
      hipFree(g_prof.d_clk);
      g_prof.d_clk = nullptr;
    }
    if (!g_prof.d_clk) {
      OZK_HIP(hipMalloc((void**)&g_prof.d_clk, (size_t)CAP * 2 * sizeof(unsigned long long)));
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
    OZK_HIP(hipMemset(g_prof.d_clk, 0, (size_t)CAP * 2 * sizeof(unsigned long long)));
    g_prof.mode = g_prof.src = PROF_CLOCK;
    g_prof.count = 0;
    return OZK_OK;
  }
  if (on && !g_prof.created) {
    hipEvent_t a, b;
    g_prof.count = 0;
    if (!g_prof.slot(&a, &b)) return fail(OZK_E_NOMEM, "cannot create profiling events");
    g_prof.created = true;
  }
  if (on) {
    g_prof.mode = g_prof.src = PROF_EVENTS;
    g_prof.every = on >= 16 ? on - 16 + 1 : 1;  // on = 16 + k: time every (k + 1)-th launch only
    g_prof.seen = 0;
    g_prof.count = 0;
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
  if (g_prof.d_clk && g_prof.src == PROF_CLOCK) {
    const double khz = g_prof.clk_khz;
    if (khz <= 0.0) return fail(OZK_E_INTERNAL, "device clock not calibrated");
    OZK_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)g_prof.count * 2);
    if (g_prof.count)
      OZK_HIP(hipMemcpy(h.data(), g_prof.d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < g_prof.count; i++) {
      const unsigned long long t0 = ~h[2 * i], t1 = h[2 * i + 1];
      if (h[2 * i] == 0 || t1 < t0) continue;  // launch never ran
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
