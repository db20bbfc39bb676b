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
  const int rc = ozk_var_msm_host((const uint8_t*)b, (const uint8_t*)s, batch_size, type, taskID, out);
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
  const int rc = ozk_var_double_msm_host((const uint8_t*)b1, (const uint8_t*)b2, (const uint8_t*)s, batch_size,
                                         taskID, out);
  ozk_release(env, scalars, s);
  ozk_release(env, bases_g2, b2);
  ozk_release(env, bases_g1, b1);
  if (rc) return ozk_throw_last(env, "variableBaseDoubleMSMNativeHelper", rc);
  return ozk_result(env, out, 576);
}
