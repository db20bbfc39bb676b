/* Helpers shared by the three JNI shim libraries (C, JNI "C view": (*env)->Fn(env, ...)).
 * The shims only move bytes between JVM arrays and the C ABI of libozk_hip.so
 * (include/ozk.h); inputs are borrowed with GetByteArrayElements and ALWAYS released
 * (JNI_ABORT: no copy-back) — the reference never releases them
 * (algebra_msm_VariableBaseMSM.cu:1624,1634). */
#ifndef OZK_JNI_COMMON_H
#define OZK_JNI_COMMON_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ozk.h"
#include "ozk_jni.h"

static jbyteArray ozk_throw(JNIEnv* env, const char* msg) {
  jclass rte = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (rte) (*env)->ThrowNew(env, rte, msg);
  return NULL;
}

static jbyteArray ozk_throw_last(JNIEnv* env, const char* what, int rc) {
  char buf[640];
  snprintf(buf, sizeof(buf), "%s failed (%d): %s", what, rc, ozk_last_error());
  return ozk_throw(env, buf);
}

/* borrow a byte[] of at least `need` bytes; NULL (exception pending) if too short */
static jbyte* ozk_borrow(JNIEnv* env, jbyteArray arr, long long need, const char* name) {
  if (!arr) { ozk_throw(env, "null byte[] argument"); return NULL; }
  if ((long long)(*env)->GetArrayLength(env, arr) < need) {
    char buf[160];
    snprintf(buf, sizeof(buf), "byte[] %s is shorter than batch_size requires (%lld bytes)", name, need);
    ozk_throw(env, buf);
    return NULL;
  }
  jbyte* p = (*env)->GetByteArrayElements(env, arr, NULL);
  if (!p) ozk_throw(env, "GetByteArrayElements returned NULL");
  return p;
}

static void ozk_release(JNIEnv* env, jbyteArray arr, jbyte* p) {
  if (p) (*env)->ReleaseByteArrayElements(env, arr, p, JNI_ABORT);
}

static jbyteArray ozk_result(JNIEnv* env, const uint8_t* data, long long n) {
  if (n > 0x7fffffffLL) return ozk_throw(env, "result exceeds the 2 GiB Java array limit");
  jbyteArray r = (*env)->NewByteArray(env, (jsize)n);
  if (!r) return NULL; /* OutOfMemoryError pending */
  (*env)->SetByteArrayRegion(env, r, 0, (jsize)n, (const jbyte*)data);
  return r;
}
#endif
