/*
 * TEST-ONLY prototype header: the handful of JNI declarations jvm/src/main/c/cssm_jni.c uses, written from the public JNI
 * specification (Java Native Interface Specification, chapter 3 "JNI Types and Data Structures" and chapter 4 "JNI Functions")
 * so that `gcc -fsyntax-only` can check the glue's arity and argument types where no JDK exists
 * (tests/test_jvm_binding.py::test_jni_glue_passes_a_syntax_only_compile).  It is hygiene, not a binding: the function table
 * below holds ONLY the members the glue calls, in no particular order, so nothing compiled against it may ever be linked or
 * run -- a real build uses $JAVA_HOME/include/jni.h (the gpu-marked test of the same file does, wherever a JDK exists).
 */
#ifndef CSSM_TEST_JNI_STUB_H
#define CSSM_TEST_JNI_STUB_H

#include <stdint.h>

/* primitive types (spec table 3-1) */
typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

/* reference types (spec section 3.2: in C all of them are the same opaque pointer) */
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbooleanArray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_COMMIT 1
#define JNI_ABORT 2

#define JNIEXPORT __attribute__((visibility("default")))
#define JNIIMPORT
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

/* the members cssm_jni.c calls, with the signatures of the specification's chapter 4 */
struct JNINativeInterface_ {
  jclass (JNICALL *FindClass)(JNIEnv* env, const char* name);
  jint (JNICALL *ThrowNew)(JNIEnv* env, jclass clazz, const char* message);
  jsize (JNICALL *GetArrayLength)(JNIEnv* env, jarray array);
  jbyte* (JNICALL *GetByteArrayElements)(JNIEnv* env, jbyteArray array, jboolean* isCopy);
  jint* (JNICALL *GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
  jdouble* (JNICALL *GetDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jboolean* isCopy);
  void (JNICALL *ReleaseByteArrayElements)(JNIEnv* env, jbyteArray array, jbyte* elems, jint mode);
  void (JNICALL *ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  void (JNICALL *ReleaseDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jdouble* elems, jint mode);
  void (JNICALL *SetDoubleArrayRegion)(JNIEnv* env, jdoubleArray array, jsize start, jsize len, const jdouble* buf);
};

#endif
