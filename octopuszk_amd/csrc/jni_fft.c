/* libAlgebraFFTAuxiliary.so — JNI native of algebra.fft.FFTAuxiliary
 * (replaces algebra_fft_FFTAuxiliary.cu:219-260).  The List<byte[]> is walked with
 * List.size()/get(i) like the reference does (FFT.cu:224-235), each element copied with
 * GetByteArrayRegion into a flat n x 32 B buffer and its local reference deleted (the
 * reference leaks one local ref and one pinned array per element). */
#include "jni_common.h"

static int ozk_small_le(JNIEnv* env, jbyteArray arr, uint8_t out[32], const char* name);

JNIEXPORT jbyteArray JNICALL Java_algebra_fft_FFTAuxiliary_serialRadix2FFTNativeHelper(
    JNIEnv* env, jclass cls, jobject inputs, jbyteArray omega, jint taskID) {
  (void)cls;
  if (!inputs || !omega) return ozk_throw(env, "null argument");
  jclass list = (*env)->FindClass(env, "java/util/List");
  if (!list) return NULL;
  jmethodID m_size = (*env)->GetMethodID(env, list, "size", "()I");
  jmethodID m_get = (*env)->GetMethodID(env, list, "get", "(I)Ljava/lang/Object;");
  if (!m_size || !m_get) return NULL;
  const jint n = (*env)->CallIntMethod(env, inputs, m_size);
  if ((*env)->ExceptionCheck(env)) return NULL;
  if (n <= 0 || (n & (n - 1))) return ozk_throw(env, "FFT input size must be a power of two");
  uint8_t* flat = (uint8_t*)calloc((size_t)n, 32);
  uint8_t* out = (uint8_t*)malloc((size_t)n * 64);
  if (!flat || !out) { free(flat); free(out); return ozk_throw(env, "out of host memory"); }
  jbyteArray result = NULL;
  int ok = 1;
  for (jint i = 0; i < n && ok; i++) {
    jbyteArray el = (jbyteArray)(*env)->CallObjectMethod(env, inputs, m_get, i);
    if ((*env)->ExceptionCheck(env) || !el) { ok = 0; break; }
    const jsize len = (*env)->GetArrayLength(env, el);
    if (len > 32) { ozk_throw(env, "FFT element longer than 32 bytes"); ok = 0; }
    else (*env)->GetByteArrayRegion(env, el, 0, len, (jbyte*)(flat + (size_t)i * 32));
    (*env)->DeleteLocalRef(env, el);
  }
  uint8_t om[32];
  memset(om, 0, sizeof(om));
  if (ok) {
    const jsize olen = (*env)->GetArrayLength(env, omega);
    if (olen > 32) { ozk_throw(env, "omega longer than 32 bytes"); ok = 0; }
    else (*env)->GetByteArrayRegion(env, omega, 0, olen, (jbyte*)om);
  }
  if (ok) {
    const int rc = ozk_fft_host(flat, n, om, taskID, out);
    if (rc) ozk_throw_last(env, "serialRadix2FFTNativeHelper", rc);
    else result = ozk_result(env, out, 64LL * n);
  }
  free(flat);
  free(out);
  return result;
}

/* OPTIONAL native (no counterpart in the reference): the whole witness map on the GPU. */
static int ozk_small_le(JNIEnv* env, jbyteArray arr, uint8_t out[32], const char* name) {
  memset(out, 0, 32);
  if (!arr) { ozk_throw(env, "null byte[] argument"); return 0; }
  const jsize len = (*env)->GetArrayLength(env, arr);
  if (len > 32) {
    char buf[96];
    snprintf(buf, sizeof(buf), "%s longer than 32 bytes", name);
    ozk_throw(env, buf);
    return 0;
  }
  (*env)->GetByteArrayRegion(env, arr, 0, len, (jbyte*)out);
  return 1;
}

JNIEXPORT jbyteArray JNICALL Java_algebra_fft_FFTAuxiliary_qapWitnessNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray a, jbyteArray b, jbyteArray c, jint m, jbyteArray omega, jbyteArray g,
    jint taskID) {
  (void)cls;
  if (m < 2 || (m & (m - 1))) return ozk_throw(env, "QAP domain size must be a power of two >= 2");
  uint8_t om[32], gg[32];
  if (!ozk_small_le(env, omega, om, "omega") || !ozk_small_le(env, g, gg, "g")) return NULL;
  const long long need = 32LL * m;
  jbyte* pa = ozk_borrow(env, a, need, "A");
  jbyte* pb = pa ? ozk_borrow(env, b, need, "B") : NULL;
  jbyte* pc = pb ? ozk_borrow(env, c, need, "C") : NULL;
  jbyteArray result = NULL;
  if (pc) {
    uint8_t* out = (uint8_t*)malloc((size_t)(need + 32));
    if (!out) ozk_throw(env, "out of host memory");
    else {
      const int rc = ozk_qap_witness_host((const uint8_t*)pa, (const uint8_t*)pb, (const uint8_t*)pc, m, om, gg,
                                          taskID, out);
      if (rc) ozk_throw_last(env, "qapWitnessNativeHelper", rc);
      else result = ozk_result(env, out, need + 32);
      free(out);
    }
  }
  if (pc) ozk_release(env, c, pc);
  if (pb) ozk_release(env, b, pb);
  if (pa) ozk_release(env, a, pa);
  return result;
}

/* ---- OPTIONAL native: the transform over ONE flat byte[] — n x 32 B
 * little-endian in, n x 32 B out — instead of a java.util.List<byte[]> walked with one JNI call per element
 * and 64-byte results (algebra_fft_FFTAuxiliary.cu:228-255, FFTAuxiliary.java:41-51). */
JNIEXPORT jbyteArray JNICALL Java_algebra_fft_FFTAuxiliary_serialRadix2FFTFlatNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray in, jint n, jbyteArray omega, jint taskID) {
  (void)cls;
  if (n <= 0 || (n & (n - 1))) return ozk_throw(env, "FFT input size must be a power of two");
  uint8_t om[32];
  if (!ozk_small_le(env, omega, om, "omega")) return NULL;
  jbyte* p = ozk_borrow(env, in, 32LL * n, "input");
  if (!p) return NULL;
  uint8_t* out = (uint8_t*)malloc((size_t)n * 32);
  int rc = out ? ozk_fft_compact_host((const uint8_t*)p, n, om, taskID, out) : OZK_E_NOMEM;
  ozk_release(env, in, p);
  jbyteArray r = NULL;
  if (!out) r = ozk_throw(env, "out of host memory");
  else if (rc) r = ozk_throw_last(env, "serialRadix2FFTFlatNativeHelper", rc);
  else r = ozk_result(env, out, 32LL * n);
  free(out);
  return r;
}
