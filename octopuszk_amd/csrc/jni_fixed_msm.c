/* libAlgebraMSMFixedBaseMSM.so — JNI natives of algebra.msm.FixedBaseMSM
 * (replaces algebra_msm_FixedBaseMSM.cu:1276-1558). */
#include "jni_common.h"

JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_batchMSMNativeHelper(
    JNIEnv* env, jclass cls, jint outerc, jint windowSize, jint out_len, jint inner_len, jint batch_size,
    jint scalarSize, jbyteArray base, jbyteArray scalars, jint BNType, jint taskID) {
  (void)cls;
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  const long long per = BNType == OZK_G1 ? 192 : 384;
  jbyte* b = ozk_borrow(env, base, BNType == OZK_G1 ? 96 : 192, "base");
  if (!b) return NULL;
  jbyte* s = ozk_borrow(env, scalars, 32LL * batch_size, "scalars");
  if (!s) { ozk_release(env, base, b); return NULL; }
  uint8_t* out = (uint8_t*)malloc((size_t)(per * batch_size));
  int rc = out ? ozk_fixed_batch_msm_host(outerc, windowSize, out_len, inner_len, batch_size, scalarSize,
                                          (const uint8_t*)b, (const uint8_t*)s, BNType, taskID, out)
               : OZK_E_NOMEM;
  ozk_release(env, scalars, s);
  ozk_release(env, base, b);
  jbyteArray r = NULL;
  if (!out) r = ozk_throw(env, "out of host memory");
  else if (rc) r = ozk_throw_last(env, "batchMSMNativeHelper", rc);
  else r = ozk_result(env, out, per * batch_size);
  free(out);
  return r;
}

JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_doubleBatchMSMNativeHelper(
    JNIEnv* env, jclass cls, jint outerc1, jint windowSize1, jint outerc2, jint windowSize2, jint out_len1,
    jint inner_len1, jint out_len2, jint inner_len2, jint batch_size, jbyteArray base_g1, jbyteArray base_g2,
    jbyteArray scalars, jint taskID) {
  (void)cls;
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  jbyte* b1 = ozk_borrow(env, base_g1, 96, "G1 base");
  if (!b1) return NULL;
  jbyte* b2 = ozk_borrow(env, base_g2, 192, "G2 base");
  if (!b2) { ozk_release(env, base_g1, b1); return NULL; }
  jbyte* s = ozk_borrow(env, scalars, 32LL * batch_size, "scalars");
  if (!s) { ozk_release(env, base_g2, b2); ozk_release(env, base_g1, b1); return NULL; }
  uint8_t* out = (uint8_t*)malloc((size_t)(576LL * batch_size));
  int rc = out ? ozk_fixed_double_batch_msm_host(outerc1, windowSize1, outerc2, windowSize2, out_len1, inner_len1,
                                                 out_len2, inner_len2, batch_size, (const uint8_t*)b1,
                                                 (const uint8_t*)b2, (const uint8_t*)s, taskID, out)
               : OZK_E_NOMEM;
  ozk_release(env, scalars, s);
  ozk_release(env, base_g2, b2);
  ozk_release(env, base_g1, b1);
  jbyteArray r = NULL;
  if (!out) r = ozk_throw(env, "out of host memory");
  else if (rc) r = ozk_throw_last(env, "doubleBatchMSMNativeHelper", rc);
  else r = ozk_result(env, out, 576LL * batch_size);
  free(out);
  return r;
}

JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_fieldBatchMSMNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray in, jint batch_size, jint taskID) {
  (void)cls;
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  jbyte* p = ozk_borrow(env, in, 32LL * (batch_size + 1LL), "scalars||base");
  if (!p) return NULL;
  uint8_t* out = (uint8_t*)malloc((size_t)(64LL * batch_size));
  int rc = out ? ozk_field_batch_mul_host((const uint8_t*)p, batch_size, taskID, out) : OZK_E_NOMEM;
  ozk_release(env, in, p);
  jbyteArray r = NULL;
  if (!out) r = ozk_throw(env, "out of host memory");
  else if (rc) r = ozk_throw_last(env, "fieldBatchMSMNativeHelper", rc);
  else r = ozk_result(env, out, 64LL * batch_size);
  free(out);
  return r;
}

/* ---- OPTIONAL native (not declared by the reference's Java): the same
 * batch with COMPACT output — X|Y|Z as 32-byte little-endian values, n x 96 B (G1) / n x 192 B (G2), i.e. half
 * the bytes of batchMSMNativeHelper (64-byte big-endian coordinates, algebra_msm_FixedBaseMSM.cu:783-787) and
 * exactly what the variable-base natives take as `bases`, so a key element is never re-marshalled. */
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_batchMSMCompactNativeHelper(
    JNIEnv* env, jclass cls, jint outerc, jint windowSize, jint batch_size, jbyteArray base, jbyteArray scalars,
    jint BNType, jint taskID) {
  (void)cls;
  if (batch_size <= 0) return ozk_throw(env, "batch_size must be positive");
  const long long per = BNType == OZK_G1 ? 96 : 192;
  jbyte* b = ozk_borrow(env, base, BNType == OZK_G1 ? 96 : 192, "base");
  if (!b) return NULL;
  jbyte* s = ozk_borrow(env, scalars, 32LL * batch_size, "scalars");
  if (!s) { ozk_release(env, base, b); return NULL; }
  uint8_t* out = (uint8_t*)malloc((size_t)(per * batch_size));
  int rc = out ? ozk_fixed_batch_msm_compact_host(outerc, windowSize, batch_size, (const uint8_t*)b, (const uint8_t*)s,
                                                  BNType, taskID, out)
               : OZK_E_NOMEM;
  ozk_release(env, scalars, s);
  ozk_release(env, base, b);
  jbyteArray r = NULL;
  if (!out) r = ozk_throw(env, "out of host memory");
  else if (rc) r = ozk_throw_last(env, "batchMSMCompactNativeHelper", rc);
  else r = ozk_result(env, out, per * batch_size);
  free(out);
  return r;
}
