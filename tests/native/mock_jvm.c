/* TEST INFRASTRUCTURE: a mock JNIEnv (function table with the slots the shims use) plus
 * ctypes-callable drivers that dlopen a shim library and call its Java_* natives the way
 * a JVM would.  Validates the shim logic, the slot indices of include/ozk_jni.h and the
 * byte plumbing without a JDK (jni.h is absent from the image). */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ozk_jni.h"

typedef struct { int kind; jsize len; jbyte* data; int pinned; } MArray;   /* kind 1 */
typedef struct { int kind; int n; MArray** items; } MList;                 /* kind 2 */

static char g_exc[1024];
static int g_exc_pending, g_live_pins, g_local_refs_deleted;
static int g_cls_rte = 11, g_cls_list = 22, g_mid_size = 33, g_mid_get = 44;

static jclass m_FindClass(JNIEnv* e, const char* name) {
  (void)e;
  if (!strcmp(name, "java/lang/RuntimeException")) return &g_cls_rte;
  if (!strcmp(name, "java/util/List") || !strcmp(name, "java/util/ArrayList")) return &g_cls_list;
  return NULL;
}
static jint m_ThrowNew(JNIEnv* e, jclass c, const char* msg) {
  (void)e; (void)c;
  snprintf(g_exc, sizeof(g_exc), "%s", msg);
  g_exc_pending = 1;
  return 0;
}
static void m_ExceptionClear(JNIEnv* e) { (void)e; g_exc_pending = 0; }
static jboolean m_ExceptionCheck(JNIEnv* e) { (void)e; return (jboolean)g_exc_pending; }
static void m_DeleteLocalRef(JNIEnv* e, jobject o) { (void)e; (void)o; g_local_refs_deleted++; }
static jmethodID m_GetMethodID(JNIEnv* e, jclass c, const char* name, const char* sig) {
  (void)e; (void)sig;
  if (c != &g_cls_list) return NULL;
  if (!strcmp(name, "size")) return &g_mid_size;
  if (!strcmp(name, "get")) return &g_mid_get;
  return NULL;
}
static jint m_CallIntMethod(JNIEnv* e, jobject o, jmethodID m, ...) {
  (void)e;
  if (m == &g_mid_size) return ((MList*)o)->n;
  return -1;
}
#include <stdarg.h>
static jobject m_CallObjectMethod(JNIEnv* e, jobject o, jmethodID m, ...) {
  (void)e;
  if (m != &g_mid_get) return NULL;
  va_list ap; va_start(ap, m);
  jint i = va_arg(ap, jint);
  va_end(ap);
  MList* l = (MList*)o;
  return (i >= 0 && i < l->n) ? l->items[i] : NULL;
}
static jsize m_GetArrayLength(JNIEnv* e, jarray a) { (void)e; return ((MArray*)a)->len; }
static jbyteArray m_NewByteArray(JNIEnv* e, jsize n) {
  (void)e;
  MArray* a = (MArray*)calloc(1, sizeof(MArray));
  a->kind = 1; a->len = n; a->data = (jbyte*)calloc((size_t)(n > 0 ? n : 1), 1);
  return a;
}
static jbyte* m_GetByteArrayElements(JNIEnv* e, jbyteArray a, jboolean* isCopy) {
  (void)e; if (isCopy) *isCopy = 0;
  ((MArray*)a)->pinned++; g_live_pins++;
  return ((MArray*)a)->data;
}
static void m_ReleaseByteArrayElements(JNIEnv* e, jbyteArray a, jbyte* p, jint mode) {
  (void)e; (void)p; (void)mode;
  ((MArray*)a)->pinned--; g_live_pins--;
}
static void m_GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, jbyte* buf) {
  (void)e; memcpy(buf, ((MArray*)a)->data + s, (size_t)l);
}
static void m_SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, const jbyte* buf) {
  (void)e; memcpy(((MArray*)a)->data + s, buf, (size_t)l);
}

static struct OzkJNINativeInterface g_table;
static const struct OzkJNINativeInterface* g_env_ptr = &g_table;

static JNIEnv* env(void) {
  memset(&g_table, 0, sizeof(g_table));
  g_table.FindClass = m_FindClass;
  g_table.ThrowNew = m_ThrowNew;
  g_table.ExceptionClear = m_ExceptionClear;
  g_table.ExceptionCheck = m_ExceptionCheck;
  g_table.DeleteLocalRef = m_DeleteLocalRef;
  g_table.GetMethodID = m_GetMethodID;
  g_table.CallIntMethod = m_CallIntMethod;
  g_table.CallObjectMethod = m_CallObjectMethod;
  g_table.GetArrayLength = m_GetArrayLength;
  g_table.NewByteArray = m_NewByteArray;
  g_table.GetByteArrayElements = m_GetByteArrayElements;
  g_table.ReleaseByteArrayElements = m_ReleaseByteArrayElements;
  g_table.GetByteArrayRegion = m_GetByteArrayRegion;
  g_table.SetByteArrayRegion = m_SetByteArrayRegion;
  g_exc_pending = 0; g_exc[0] = 0; g_live_pins = 0; g_local_refs_deleted = 0;
  return (JNIEnv*)&g_env_ptr;
}
static MArray* mk(const void* p, long n) {
  MArray* a = (MArray*)calloc(1, sizeof(MArray));
  a->kind = 1; a->len = (jsize)n; a->data = (jbyte*)malloc((size_t)(n > 0 ? n : 1));
  if (n > 0) memcpy(a->data, p, (size_t)n);
  return a;
}
static void fr(MArray* a) { if (a) { free(a->data); free(a); } }

/* copy result out; return length, or -1 with the exception text in err */
static long finish(jbyteArray r, unsigned char* out, long cap, char* err) {
  if (g_live_pins != 0) { snprintf(err, 512, "shim leaked %d pinned arrays", g_live_pins); return -2; }
  if (!r) { snprintf(err, 512, "%s", g_exc_pending ? g_exc : "NULL without exception"); return -1; }
  MArray* a = (MArray*)r;
  long n = a->len;
  if (n > cap) { snprintf(err, 512, "result too large"); fr(a); return -3; }
  memcpy(out, a->data, (size_t)n);
  fr(a);
  return n;
}
static void* sym(const char* lib, const char* name, char* err) {
  void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { snprintf(err, 512, "dlopen: %s", dlerror()); return NULL; }
  void* f = dlsym(h, name);
  if (!f) snprintf(err, 512, "dlsym %s: %s", name, dlerror());
  return f;
}

typedef jbyteArray (*fn_var)(JNIEnv*, jclass, jbyteArray, jbyteArray, jint, jint, jint);
long mock_var_msm(const char* lib, const void* bases, long bl, const void* sc, long sl, int n, int type, int task,
                  unsigned char* out, long cap, char* err) {
  fn_var f = (fn_var)sym(lib, "Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *a = mk(bases, bl), *b = mk(sc, sl);
  jbyteArray r = f(e, NULL, a, b, n, type, task);
  long rc = finish(r, out, cap, err);
  fr(a); fr(b);
  return rc;
}
typedef jbyteArray (*fn_dvar)(JNIEnv*, jclass, jbyteArray, jbyteArray, jbyteArray, jint, jint);
long mock_var_double_msm(const char* lib, const void* b1, long l1, const void* b2, long l2, const void* sc, long sl,
                         int n, int task, unsigned char* out, long cap, char* err) {
  fn_dvar f = (fn_dvar)sym(lib, "Java_algebra_msm_VariableBaseMSM_variableBaseDoubleMSMNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *a = mk(b1, l1), *b = mk(b2, l2), *c = mk(sc, sl);
  long rc = finish(f(e, NULL, a, b, c, n, task), out, cap, err);
  fr(a); fr(b); fr(c);
  return rc;
}
typedef jbyteArray (*fn_fb)(JNIEnv*, jclass, jint, jint, jint, jint, jint, jint, jbyteArray, jbyteArray, jint, jint);
long mock_fixed_batch(const char* lib, int outerc, int ws, int out_len, int inner_len, int n, int scalar_size,
                      const void* base, long bl, const void* sc, long sl, int bn, int task, unsigned char* out,
                      long cap, char* err) {
  fn_fb f = (fn_fb)sym(lib, "Java_algebra_msm_FixedBaseMSM_batchMSMNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *a = mk(base, bl), *b = mk(sc, sl);
  long rc = finish(f(e, NULL, outerc, ws, out_len, inner_len, n, scalar_size, a, b, bn, task), out, cap, err);
  fr(a); fr(b);
  return rc;
}
typedef jbyteArray (*fn_dfb)(JNIEnv*, jclass, jint, jint, jint, jint, jint, jint, jint, jint, jint, jbyteArray,
                             jbyteArray, jbyteArray, jint);
long mock_fixed_double_batch(const char* lib, int oc1, int ws1, int oc2, int ws2, int n, const void* b1, long l1,
                             const void* b2, long l2, const void* sc, long sl, int task, unsigned char* out, long cap,
                             char* err) {
  fn_dfb f = (fn_dfb)sym(lib, "Java_algebra_msm_FixedBaseMSM_doubleBatchMSMNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *a = mk(b1, l1), *b = mk(b2, l2), *c = mk(sc, sl);
  long rc = finish(f(e, NULL, oc1, ws1, oc2, ws2, oc1, 1 << ws1, oc2, 1 << ws2, n, a, b, c, task), out, cap, err);
  fr(a); fr(b); fr(c);
  return rc;
}
typedef jbyteArray (*fn_fm)(JNIEnv*, jclass, jbyteArray, jint, jint);
long mock_field_mul(const char* lib, const void* in, long il, int n, int task, unsigned char* out, long cap, char* err) {
  fn_fm f = (fn_fm)sym(lib, "Java_algebra_msm_FixedBaseMSM_fieldBatchMSMNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray* a = mk(in, il);
  long rc = finish(f(e, NULL, a, n, task), out, cap, err);
  fr(a);
  return rc;
}
typedef jbyteArray (*fn_fft)(JNIEnv*, jclass, jobject, jbyteArray, jint);
/* elements: n items, item i has lens[i] bytes at data + offs[i] */
long mock_fft(const char* lib, const unsigned char* data, const long* offs, const int* lens, int n, const void* omega,
              long ol, int task, unsigned char* out, long cap, char* err, int* refs_deleted) {
  fn_fft f = (fn_fft)sym(lib, "Java_algebra_fft_FFTAuxiliary_serialRadix2FFTNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MList l; l.kind = 2; l.n = n; l.items = (MArray**)calloc((size_t)(n > 0 ? n : 1), sizeof(MArray*));
  for (int i = 0; i < n; i++) l.items[i] = mk(data + offs[i], lens[i]);
  MArray* om = mk(omega, ol);
  long rc = finish(f(e, NULL, &l, om, task), out, cap, err);
  if (refs_deleted) *refs_deleted = g_local_refs_deleted;
  for (int i = 0; i < n; i++) fr(l.items[i]);
  free(l.items); fr(om);
  return rc;
}

typedef jbyteArray (*fn_qap)(JNIEnv*, jclass, jbyteArray, jbyteArray, jbyteArray, jint, jbyteArray, jbyteArray, jint);
long mock_qap_witness(const char* lib, const void* a, const void* b, const void* c, long vl, int m, const void* omega,
                      long ol, const void* g, long gl, int task, unsigned char* out, long cap, char* err) {
  fn_qap f = (fn_qap)sym(lib, "Java_algebra_fft_FFTAuxiliary_qapWitnessNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *ma = mk(a, vl), *mb = mk(b, vl), *mc = mk(c, vl), *mo = mk(omega, ol), *mg = mk(g, gl);
  long rc = finish(f(e, NULL, ma, mb, mc, m, mo, mg, task), out, cap, err);
  fr(ma); fr(mb); fr(mc); fr(mo); fr(mg);
  return rc;
}

typedef jlong (*fn_prep)(JNIEnv*, jclass, jbyteArray, jint, jint, jint);
typedef jbyteArray (*fn_prep_msm)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint);
typedef void (*fn_rel)(JNIEnv*, jclass, jlong);
/* prepare once, run `reps` MSMs with the same scalars (outputs must agree), release */
long mock_prepared_msm(const char* lib, const void* bases, long bl, const void* sc, long sl, int n, int type, int task,
                       int reps, unsigned char* out, long cap, char* err) {
  fn_prep fp = (fn_prep)sym(lib, "Java_algebra_msm_VariableBaseMSM_prepareBasesNativeHelper", err);
  fn_prep_msm fm = (fn_prep_msm)sym(lib, "Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMPreparedNativeHelper", err);
  fn_rel fr_ = (fn_rel)sym(lib, "Java_algebra_msm_VariableBaseMSM_releaseBasesNativeHelper", err);
  if (!fp || !fm || !fr_) return -9;
  JNIEnv* e = env();
  MArray *a = mk(bases, bl), *b = mk(sc, sl);
  jlong h = fp(e, NULL, a, n, type, task);
  long rc = -1;
  if (g_exc_pending || !h) {
    snprintf(err, 512, "%s", g_exc_pending ? g_exc : "null handle without exception");
  } else if (g_live_pins != 0) {
    snprintf(err, 512, "shim leaked %d pinned arrays", g_live_pins);
    rc = -2;
  } else {
    for (int k = 0; k < reps; k++) {
      rc = finish(fm(e, NULL, h, b, n, type), out, cap, err);
      if (rc < 0) break;
    }
    fr_(e, NULL, h);
  }
  fr(a); fr(b);
  return rc;
}

/* ---- optional compact natives */
typedef jbyteArray (*fn_fbc)(JNIEnv*, jclass, jint, jint, jint, jbyteArray, jbyteArray, jint, jint);
long mock_fixed_batch_compact(const char* lib, int outerc, int ws, int n, const void* base, long bl, const void* sc,
                              long sl, int bn, int task, unsigned char* out, long cap, char* err) {
  fn_fbc f = (fn_fbc)sym(lib, "Java_algebra_msm_FixedBaseMSM_batchMSMCompactNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *a = mk(base, bl), *b = mk(sc, sl);
  long rc = finish(f(e, NULL, outerc, ws, n, a, b, bn, task), out, cap, err);
  fr(a); fr(b);
  return rc;
}
typedef jbyteArray (*fn_fftflat)(JNIEnv*, jclass, jbyteArray, jint, jbyteArray, jint);
long mock_fft_flat(const char* lib, const void* in, long il, int n, const void* omega, long ol, int task,
                   unsigned char* out, long cap, char* err) {
  fn_fftflat f = (fn_fftflat)sym(lib, "Java_algebra_fft_FFTAuxiliary_serialRadix2FFTFlatNativeHelper", err);
  if (!f) return -9;
  JNIEnv* e = env();
  MArray *a = mk(in, il), *om = mk(omega, ol);
  long rc = finish(f(e, NULL, a, n, om, task), out, cap, err);
  fr(a); fr(om);
  return rc;
}
