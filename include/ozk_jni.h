/* ozk_jni.h — the JNI surface of the drop-in and a minimal, self-contained
 * declaration of the JNI types it needs.
 *
 * There is no JDK (no jni.h) in the build image, so the JNIEnv function table is declared
 * here slot by slot in the order fixed by the Java Native Interface Specification
 * ("JNI Functions", interface function table; identical in JDK 8 .. 21).  Only the slots
 * the shims call are typed; the rest are void* placeholders that keep the indices right
 * (checked by static_asserts below: FindClass 6, ThrowNew 14, GetMethodID 33,
 * CallObjectMethod 34, CallIntMethod 49, GetArrayLength 171, NewByteArray 176,
 * Get/ReleaseByteArrayElements 184/192, Get/SetByteArrayRegion 200/208, ExceptionCheck 228).
 * When a JDK is available, compile the shims with -DOZK_USE_SYSTEM_JNI to use <jni.h> instead.
 *
 * The six exported symbols are exactly the natives the reference's Java declares:
 *   algebra.msm.VariableBaseMSM  (VariableBaseMSM.java:193-197, 473-478; algebra_msm_VariableBaseMSM.h:13-24)
 *   algebra.msm.FixedBaseMSM     (FixedBaseMSM.java:102-109, 473-485, 747-749; algebra_msm_FixedBaseMSM.h:13-32)
 *   algebra.fft.FFTAuxiliary     (FFTAuxiliary.java:53-55; algebra_fft_FFTAuxiliary.h:13-16)
 * built into libAlgebraMSMVariableBaseMSM.so, libAlgebraMSMFixedBaseMSM.so and
 * libAlgebraFFTAuxiliary.so — the names System.loadLibrary asks for
 * (VariableBaseMSM.java:31-34, FixedBaseMSM.java:44-47, SerialFFT.java:20-23).
 * Unlike the reference (prints and exit(-1), VariableBaseMSM.cu:1417-1422) every failure
 * becomes a java.lang.RuntimeException and a NULL return.
 */
#ifndef OZK_JNI_H
#define OZK_JNI_H

#ifdef OZK_USE_SYSTEM_JNI
#include <jni.h>
#else
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
typedef void* jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef void* jmethodID;

#define JNI_ABORT 2
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

#ifdef __cplusplus
#define OZK_SA(c, m) static_assert(c, m)
#else
#define OZK_SA(c, m) _Static_assert(c, m)
#endif

struct OzkJNINativeInterface;
typedef const struct OzkJNINativeInterface* JNIEnv; /* C view: JNIEnv is a pointer to the table pointer */

struct OzkJNINativeInterface {
  void* reserved0; /* 0 */
  void* reserved1; /* 1 */
  void* reserved2; /* 2 */
  void* reserved3; /* 3 */
  void* GetVersion; /* 4 */
  void* DefineClass; /* 5 */
  jclass (*FindClass)(JNIEnv*, const char*); /* 6 */
  void* FromReflectedMethod; /* 7 */
  void* FromReflectedField; /* 8 */
  void* ToReflectedMethod; /* 9 */
  void* GetSuperclass; /* 10 */
  void* IsAssignableFrom; /* 11 */
  void* ToReflectedField; /* 12 */
  void* Throw; /* 13 */
  jint (*ThrowNew)(JNIEnv*, jclass, const char*); /* 14 */
  void* ExceptionOccurred; /* 15 */
  void* ExceptionDescribe; /* 16 */
  void (*ExceptionClear)(JNIEnv*); /* 17 */
  void* FatalError; /* 18 */
  void* PushLocalFrame; /* 19 */
  void* PopLocalFrame; /* 20 */
  void* NewGlobalRef; /* 21 */
  void* DeleteGlobalRef; /* 22 */
  void (*DeleteLocalRef)(JNIEnv*, jobject); /* 23 */
  void* IsSameObject; /* 24 */
  void* NewLocalRef; /* 25 */
  void* EnsureLocalCapacity; /* 26 */
  void* AllocObject; /* 27 */
  void* NewObject; /* 28 */
  void* NewObjectV; /* 29 */
  void* NewObjectA; /* 30 */
  jclass (*GetObjectClass)(JNIEnv*, jobject); /* 31 */
  void* IsInstanceOf; /* 32 */
  jmethodID (*GetMethodID)(JNIEnv*, jclass, const char*, const char*); /* 33 */
  jobject (*CallObjectMethod)(JNIEnv*, jobject, jmethodID, ...); /* 34 */
  void* CallObjectMethodV; /* 35 */
  void* CallObjectMethodA; /* 36 */
  void* CallBooleanMethod; /* 37 */
  void* CallBooleanMethodV; /* 38 */
  void* CallBooleanMethodA; /* 39 */
  void* CallByteMethod; /* 40 */
  void* CallByteMethodV; /* 41 */
  void* CallByteMethodA; /* 42 */
  void* CallCharMethod; /* 43 */
  void* CallCharMethodV; /* 44 */
  void* CallCharMethodA; /* 45 */
  void* CallShortMethod; /* 46 */
  void* CallShortMethodV; /* 47 */
  void* CallShortMethodA; /* 48 */
  jint (*CallIntMethod)(JNIEnv*, jobject, jmethodID, ...); /* 49 */
  void* CallIntMethodV; /* 50 */
  void* CallIntMethodA; /* 51 */
  void* CallLongMethod; /* 52 */
  void* CallLongMethodV; /* 53 */
  void* CallLongMethodA; /* 54 */
  void* CallFloatMethod; /* 55 */
  void* CallFloatMethodV; /* 56 */
  void* CallFloatMethodA; /* 57 */
  void* CallDoubleMethod; /* 58 */
  void* CallDoubleMethodV; /* 59 */
  void* CallDoubleMethodA; /* 60 */
  void* CallVoidMethod; /* 61 */
  void* CallVoidMethodV; /* 62 */
  void* CallVoidMethodA; /* 63 */
  void* CallNonvirtualObjectMethod; /* 64 */
  void* CallNonvirtualObjectMethodV; /* 65 */
  void* CallNonvirtualObjectMethodA; /* 66 */
  void* CallNonvirtualBooleanMethod; /* 67 */
  void* CallNonvirtualBooleanMethodV; /* 68 */
  void* CallNonvirtualBooleanMethodA; /* 69 */
  void* CallNonvirtualByteMethod; /* 70 */
  void* CallNonvirtualByteMethodV; /* 71 */
  void* CallNonvirtualByteMethodA; /* 72 */
  void* CallNonvirtualCharMethod; /* 73 */
  void* CallNonvirtualCharMethodV; /* 74 */
  void* CallNonvirtualCharMethodA; /* 75 */
  void* CallNonvirtualShortMethod; /* 76 */
  void* CallNonvirtualShortMethodV; /* 77 */
  void* CallNonvirtualShortMethodA; /* 78 */
  void* CallNonvirtualIntMethod; /* 79 */
  void* CallNonvirtualIntMethodV; /* 80 */
  void* CallNonvirtualIntMethodA; /* 81 */
  void* CallNonvirtualLongMethod; /* 82 */
  void* CallNonvirtualLongMethodV; /* 83 */
  void* CallNonvirtualLongMethodA; /* 84 */
  void* CallNonvirtualFloatMethod; /* 85 */
  void* CallNonvirtualFloatMethodV; /* 86 */
  void* CallNonvirtualFloatMethodA; /* 87 */
  void* CallNonvirtualDoubleMethod; /* 88 */
  void* CallNonvirtualDoubleMethodV; /* 89 */
  void* CallNonvirtualDoubleMethodA; /* 90 */
  void* CallNonvirtualVoidMethod; /* 91 */
  void* CallNonvirtualVoidMethodV; /* 92 */
  void* CallNonvirtualVoidMethodA; /* 93 */
  void* GetFieldID; /* 94 */
  void* GetObjectField; /* 95 */
  void* GetBooleanField; /* 96 */
  void* GetByteField; /* 97 */
  void* GetCharField; /* 98 */
  void* GetShortField; /* 99 */
  void* GetIntField; /* 100 */
  void* GetLongField; /* 101 */
  void* GetFloatField; /* 102 */
  void* GetDoubleField; /* 103 */
  void* SetObjectField; /* 104 */
  void* SetBooleanField; /* 105 */
  void* SetByteField; /* 106 */
  void* SetCharField; /* 107 */
  void* SetShortField; /* 108 */
  void* SetIntField; /* 109 */
  void* SetLongField; /* 110 */
  void* SetFloatField; /* 111 */
  void* SetDoubleField; /* 112 */
  void* GetStaticMethodID; /* 113 */
  void* CallStaticObjectMethod; /* 114 */
  void* CallStaticObjectMethodV; /* 115 */
  void* CallStaticObjectMethodA; /* 116 */
  void* CallStaticBooleanMethod; /* 117 */
  void* CallStaticBooleanMethodV; /* 118 */
  void* CallStaticBooleanMethodA; /* 119 */
  void* CallStaticByteMethod; /* 120 */
  void* CallStaticByteMethodV; /* 121 */
  void* CallStaticByteMethodA; /* 122 */
  void* CallStaticCharMethod; /* 123 */
  void* CallStaticCharMethodV; /* 124 */
  void* CallStaticCharMethodA; /* 125 */
  void* CallStaticShortMethod; /* 126 */
  void* CallStaticShortMethodV; /* 127 */
  void* CallStaticShortMethodA; /* 128 */
  void* CallStaticIntMethod; /* 129 */
  void* CallStaticIntMethodV; /* 130 */
  void* CallStaticIntMethodA; /* 131 */
  void* CallStaticLongMethod; /* 132 */
  void* CallStaticLongMethodV; /* 133 */
  void* CallStaticLongMethodA; /* 134 */
  void* CallStaticFloatMethod; /* 135 */
  void* CallStaticFloatMethodV; /* 136 */
  void* CallStaticFloatMethodA; /* 137 */
  void* CallStaticDoubleMethod; /* 138 */
  void* CallStaticDoubleMethodV; /* 139 */
  void* CallStaticDoubleMethodA; /* 140 */
  void* CallStaticVoidMethod; /* 141 */
  void* CallStaticVoidMethodV; /* 142 */
  void* CallStaticVoidMethodA; /* 143 */
  void* GetStaticFieldID; /* 144 */
  void* GetStaticObjectField; /* 145 */
  void* GetStaticBooleanField; /* 146 */
  void* GetStaticByteField; /* 147 */
  void* GetStaticCharField; /* 148 */
  void* GetStaticShortField; /* 149 */
  void* GetStaticIntField; /* 150 */
  void* GetStaticLongField; /* 151 */
  void* GetStaticFloatField; /* 152 */
  void* GetStaticDoubleField; /* 153 */
  void* SetStaticObjectField; /* 154 */
  void* SetStaticBooleanField; /* 155 */
  void* SetStaticByteField; /* 156 */
  void* SetStaticCharField; /* 157 */
  void* SetStaticShortField; /* 158 */
  void* SetStaticIntField; /* 159 */
  void* SetStaticLongField; /* 160 */
  void* SetStaticFloatField; /* 161 */
  void* SetStaticDoubleField; /* 162 */
  void* NewString; /* 163 */
  void* GetStringLength; /* 164 */
  void* GetStringChars; /* 165 */
  void* ReleaseStringChars; /* 166 */
  void* NewStringUTF; /* 167 */
  void* GetStringUTFLength; /* 168 */
  void* GetStringUTFChars; /* 169 */
  void* ReleaseStringUTFChars; /* 170 */
  jsize (*GetArrayLength)(JNIEnv*, jarray); /* 171 */
  void* NewObjectArray; /* 172 */
  void* GetObjectArrayElement; /* 173 */
  void* SetObjectArrayElement; /* 174 */
  void* NewBooleanArray; /* 175 */
  jbyteArray (*NewByteArray)(JNIEnv*, jsize); /* 176 */
  void* NewCharArray; /* 177 */
  void* NewShortArray; /* 178 */
  void* NewIntArray; /* 179 */
  void* NewLongArray; /* 180 */
  void* NewFloatArray; /* 181 */
  void* NewDoubleArray; /* 182 */
  void* GetBooleanArrayElements; /* 183 */
  jbyte* (*GetByteArrayElements)(JNIEnv*, jbyteArray, jboolean*); /* 184 */
  void* GetCharArrayElements; /* 185 */
  void* GetShortArrayElements; /* 186 */
  void* GetIntArrayElements; /* 187 */
  void* GetLongArrayElements; /* 188 */
  void* GetFloatArrayElements; /* 189 */
  void* GetDoubleArrayElements; /* 190 */
  void* ReleaseBooleanArrayElements; /* 191 */
  void (*ReleaseByteArrayElements)(JNIEnv*, jbyteArray, jbyte*, jint); /* 192 */
  void* ReleaseCharArrayElements; /* 193 */
  void* ReleaseShortArrayElements; /* 194 */
  void* ReleaseIntArrayElements; /* 195 */
  void* ReleaseLongArrayElements; /* 196 */
  void* ReleaseFloatArrayElements; /* 197 */
  void* ReleaseDoubleArrayElements; /* 198 */
  void* GetBooleanArrayRegion; /* 199 */
  void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*); /* 200 */
  void* GetCharArrayRegion; /* 201 */
  void* GetShortArrayRegion; /* 202 */
  void* GetIntArrayRegion; /* 203 */
  void* GetLongArrayRegion; /* 204 */
  void* GetFloatArrayRegion; /* 205 */
  void* GetDoubleArrayRegion; /* 206 */
  void* SetBooleanArrayRegion; /* 207 */
  void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*); /* 208 */
  void* SetCharArrayRegion; /* 209 */
  void* SetShortArrayRegion; /* 210 */
  void* SetIntArrayRegion; /* 211 */
  void* SetLongArrayRegion; /* 212 */
  void* SetFloatArrayRegion; /* 213 */
  void* SetDoubleArrayRegion; /* 214 */
  void* RegisterNatives; /* 215 */
  void* UnregisterNatives; /* 216 */
  void* MonitorEnter; /* 217 */
  void* MonitorExit; /* 218 */
  void* GetJavaVM; /* 219 */
  void* GetStringRegion; /* 220 */
  void* GetStringUTFRegion; /* 221 */
  void* (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*); /* 222 */
  void (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint); /* 223 */
  void* GetStringCritical; /* 224 */
  void* ReleaseStringCritical; /* 225 */
  void* NewWeakGlobalRef; /* 226 */
  void* DeleteWeakGlobalRef; /* 227 */
  jboolean (*ExceptionCheck)(JNIEnv*); /* 228 */
  void* NewDirectByteBuffer; /* 229 */
  void* GetDirectBufferAddress; /* 230 */
  void* GetDirectBufferCapacity; /* 231 */
  void* GetObjectRefType; /* 232 */
  void* GetModule; /* 233 */
};

OZK_SA(offsetof(struct OzkJNINativeInterface, FindClass) == 6 * sizeof(void*), "JNI slot FindClass");
OZK_SA(offsetof(struct OzkJNINativeInterface, ThrowNew) == 14 * sizeof(void*), "JNI slot ThrowNew");
OZK_SA(offsetof(struct OzkJNINativeInterface, NewGlobalRef) == 21 * sizeof(void*), "JNI slot NewGlobalRef");
OZK_SA(offsetof(struct OzkJNINativeInterface, DeleteLocalRef) == 23 * sizeof(void*), "JNI slot DeleteLocalRef");
OZK_SA(offsetof(struct OzkJNINativeInterface, GetObjectClass) == 31 * sizeof(void*), "JNI slot GetObjectClass");
OZK_SA(offsetof(struct OzkJNINativeInterface, GetMethodID) == 33 * sizeof(void*), "JNI slot GetMethodID");
OZK_SA(offsetof(struct OzkJNINativeInterface, CallObjectMethod) == 34 * sizeof(void*), "JNI slot CallObjectMethod");
OZK_SA(offsetof(struct OzkJNINativeInterface, CallIntMethod) == 49 * sizeof(void*), "JNI slot CallIntMethod");
OZK_SA(offsetof(struct OzkJNINativeInterface, GetArrayLength) == 171 * sizeof(void*), "JNI slot GetArrayLength");
OZK_SA(offsetof(struct OzkJNINativeInterface, NewByteArray) == 176 * sizeof(void*), "JNI slot NewByteArray");
OZK_SA(offsetof(struct OzkJNINativeInterface, GetByteArrayElements) == 184 * sizeof(void*), "JNI slot GetByteArrayElements");
OZK_SA(offsetof(struct OzkJNINativeInterface, ReleaseByteArrayElements) == 192 * sizeof(void*), "JNI slot ReleaseByteArrayElements");
OZK_SA(offsetof(struct OzkJNINativeInterface, GetByteArrayRegion) == 200 * sizeof(void*), "JNI slot GetByteArrayRegion");
OZK_SA(offsetof(struct OzkJNINativeInterface, SetByteArrayRegion) == 208 * sizeof(void*), "JNI slot SetByteArrayRegion");
OZK_SA(offsetof(struct OzkJNINativeInterface, GetPrimitiveArrayCritical) == 222 * sizeof(void*), "JNI slot GetPrimitiveArrayCritical");
OZK_SA(offsetof(struct OzkJNINativeInterface, ReleasePrimitiveArrayCritical) == 223 * sizeof(void*), "JNI slot ReleasePrimitiveArrayCritical");
OZK_SA(offsetof(struct OzkJNINativeInterface, ExceptionCheck) == 228 * sizeof(void*), "JNI slot ExceptionCheck");

#ifdef __cplusplus
}
#endif
#endif /* OZK_USE_SYSTEM_JNI */

#ifdef __cplusplus
extern "C" {
#endif

/* ([B[BIII)[B — bases (n x 96|192 B), scalars (n x 32 B), batch_size, type (1 = G1, else G2), taskID
 * -> 192 | 384 B.  Forwards to ozk_var_msm_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray bases, jbyteArray scalars, jint batch_size, jint type, jint taskID);
/* ([B[B[BII)[B -> 576 B.  Forwards to ozk_var_double_msm_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_VariableBaseMSM_variableBaseDoubleMSMNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray bases_g1, jbyteArray bases_g2, jbyteArray scalars, jint batch_size,
    jint taskID);
/* (IIIIII[B[BII)[B -> n x 192 | 384 B (64-B big-endian coordinates).  ozk_fixed_batch_msm_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_batchMSMNativeHelper(
    JNIEnv* env, jclass cls, jint outerc, jint windowSize, jint out_len, jint inner_len, jint batch_size,
    jint scalarSize, jbyteArray base, jbyteArray scalars, jint BNType, jint taskID);
/* (IIIIIIIII[B[B[BI)[B -> n x 576 B.  ozk_fixed_double_batch_msm_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_doubleBatchMSMNativeHelper(
    JNIEnv* env, jclass cls, jint outerc1, jint windowSize1, jint outerc2, jint windowSize2, jint out_len1,
    jint inner_len1, jint out_len2, jint inner_len2, jint batch_size, jbyteArray base_g1, jbyteArray base_g2,
    jbyteArray scalars, jint taskID);
/* ([BII)[B — (n+1) x 32 B in, n x 64 B big-endian out.  ozk_field_batch_mul_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_FixedBaseMSM_fieldBatchMSMNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray in, jint batch_size, jint taskID);
/* (Ljava/util/List;[BI)[B — List<byte[]> (LE, <= 32 B each), omega bytes, taskID -> n x 64 B LE.
 * ozk_fft_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_fft_FFTAuxiliary_serialRadix2FFTNativeHelper(
    JNIEnv* env, jclass cls, jobject inputs, jbyteArray omega, jint taskID);


/* ---- OPTIONAL natives (not declared by the reference's Java; INTEGRATION.md §5 shows the two-line
 * Java change that uses them) -------------------------------------------------------------------- */
/* ([B[B[BI[B[BI)[B — the QAP witness map of R1CStoQAP.R1CStoQAPWitness (R1CStoQAP.java:163-230):
 * evaluations A, B, C (m x 32 B LE each, flat), m, omega, g (<= 32 B LE), taskID -> (m + 1) x 32 B LE
 * coefficients of H.  ozk_qap_witness_host. */
JNIEXPORT jbyteArray JNICALL Java_algebra_fft_FFTAuxiliary_qapWitnessNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray a, jbyteArray b, jbyteArray c, jint m, jbyteArray omega, jbyteArray g,
    jint taskID);
/* ([BIII)J / (J[BII)[B / (J)V — prepared bases: upload and convert a base array
 * once, run any number of MSMs over it (only the scalars travel), release it.  ozk_bases_create_host,
 * ozk_var_msm_bases_host, ozk_bases_destroy.  The bytes returned equal those of
 * variableBaseSerialMSMNativeHelper on the same inputs. */
JNIEXPORT jlong JNICALL Java_algebra_msm_VariableBaseMSM_prepareBasesNativeHelper(
    JNIEnv* env, jclass cls, jbyteArray bases, jint batch_size, jint type, jint taskID);
JNIEXPORT jbyteArray JNICALL Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMPreparedNativeHelper(
    JNIEnv* env, jclass cls, jlong handle, jbyteArray scalars, jint batch_size, jint type);
JNIEXPORT void JNICALL Java_algebra_msm_VariableBaseMSM_releaseBasesNativeHelper(JNIEnv* env, jclass cls,
                                                                                 jlong handle);

#ifdef __cplusplus
}
#endif
#endif /* OZK_JNI_H */
