/* libAlgebraMSMVariableBaseMSM.so — JNI natives of algebra.msm.VariableBaseMSM
 * (replaces algebra_msm_VariableBaseMSM.cu:1614-1788). */
#include "jni_common.h"

JNIEXPORT jbyteArray JNICALL Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray bases, jbyteArray scalars, jint batch_size, jint type, jint taskID) {
  (void)cls;
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  const long long pt = type == OZK_G1 ? 96 : 192;
  jbyte* b = ozk_borrow(env, bases, pt * batch_size, "bases");
  if (!b) return NULL;
  jbyte* s = ozk_borrow(env, scalars, 32LL * batch_size, "scalars");
  if (!s) { ozk_release(env, bases, b); return NULL; }
  uint8_t out[384];
  /* one GPU (taskID % count), as the reference; OZK_SHARD=1 (opt-in) spreads calls of >= 2^21 pairs over the visible
   * GPUs (include/ozk.h, ozk_var_msm_auto_host) */
  const int rc = ozk_var_msm_auto_host((const uint8_t*)b, (const uint8_t*)s, batch_size, type, taskID, out);
  ozk_release(env, scalars, s);
  ozk_release(env, bases, b);
  if (rc) return ozk_throw_last(env, "variableBaseSerialMSMNativeHelper", rc);
  return ozk_result(env, out, type == OZK_G1 ? 192 : 384);
}

JNIEXPORT jbyteArray JNICALL Java_algebra_msm_VariableBaseMSM_variableBaseDoubleMSMNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray bases_g1, jbyteArray bases_g2, jbyteArray scalars, jint batch_size,
    jint taskID) {
  (void)cls;
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  jbyte* b1 = ozk_borrow(env, bases_g1, 96LL * batch_size, "G1 bases");
  if (!b1) return NULL;
  jbyte* b2 = ozk_borrow(env, bases_g2, 192LL * batch_size, "G2 bases");
  if (!b2) { ozk_release(env, bases_g1, b1); return NULL; }
  jbyte* s = ozk_borrow(env, scalars, 32LL * batch_size, "scalars");
  if (!s) { ozk_release(env, bases_g2, b2); ozk_release(env, bases_g1, b1); return NULL; }
  uint8_t out[576];
  const int rc = ozk_var_double_msm_auto_host((const uint8_t*)b1, (const uint8_t*)b2, (const uint8_t*)s, batch_size,
                                              taskID, out);
  ozk_release(env, scalars, s);
  ozk_release(env, bases_g2, b2);
  ozk_release(env, bases_g1, b1);
  if (rc) return ozk_throw_last(env, "variableBaseDoubleMSMNativeHelper", rc);
  return ozk_result(env, out, 576);
}

/* ---- OPTIONAL natives (not declared by the reference's Java): bases kept on the GPU
 * across MSMs.  The handle travels through Java as a long. */
JNIEXPORT jlong JNICALL Java_algebra_msm_VariableBaseMSM_prepareBasesNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray bases, jint batch_size, jint type, jint taskID) {
  (void)cls;
  if (batch_size <= 0) { ozk_throw(env, "batch_size must be positive"); return 0; }
  const long long pt = type == OZK_G1 ? 96 : 192;
  jbyte* b = ozk_borrow(env, bases, pt * batch_size, "bases");
  if (!b) return 0;
  void* h = NULL;
  const int rc = ozk_bases_create_host((const uint8_t*)b, batch_size, type, taskID, &h);
  ozk_release(env, bases, b);
  if (rc) { ozk_throw_last(env, "prepareBasesNativeHelper", rc); return 0; }
  return (jlong)(intptr_t)h;
}

JNIEXPORT jbyteArray JNICALL Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMPreparedNativeHelper(
    JNIEnv* env, jclass cls, jlong handle, jbyteArray scalars, jint batch_size, jint type) {
  (void)cls;
  if (!handle) return ozk_throw(env, "null bases handle");
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  /* the result's size follows the HANDLE's group; a mismatching `type` is a caller bug, not a truncation */
  const int htype = ozk_bases_type((void*)(intptr_t)handle);
  if (htype == 0) return ozk_throw(env, "stale or released bases handle");
  if ((type == OZK_G1) != (htype == OZK_G1)) return ozk_throw(env, "type does not match the prepared bases");
  jbyte* s = ozk_borrow(env, scalars, 32LL * batch_size, "scalars");
  if (!s) return NULL;
  uint8_t out[384] = {0};
  const int rc = ozk_var_msm_bases_host((void*)(intptr_t)handle, (const uint8_t*)s, batch_size, out);
  ozk_release(env, scalars, s);
  if (rc) return ozk_throw_last(env, "variableBaseSerialMSMPreparedNativeHelper", rc);
  return ozk_result(env, out, htype == OZK_G1 ? 192 : 384);
}

JNIEXPORT void JNICALL Java_algebra_msm_VariableBaseMSM_releaseBasesNativeHelper(JNIEnv* env, jclass cls,
                                                                                 jlong handle) {
  (void)env;
  (void)cls;
  ozk_bases_destroy((void*)(intptr_t)handle);
}
