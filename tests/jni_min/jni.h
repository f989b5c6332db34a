/*
 * jni.h -- a MINIMAL stand-in for the JDK's header, written from the Java Native Interface Specification (chapter 3, "JNI
 * Types and Data Structures", and chapter 4, "JNI Functions"): the primitive and reference types, JNIEXPORT / JNICALL,
 * JNI_ABORT, and a JNINativeInterface_ function table that holds ONLY the entries ahocorasick_amd/java/jni/acgpu_jni.c calls,
 * with the signatures the specification gives them.
 *
 * TEST INFRASTRUCTURE.  The build image has no JDK (no jni.h, no javac, no JVM).  This header exists so that the JNI glue goes
 * through a compiler (-Wall -Wextra -Werror, ASan + UBSan) and runs against a mock JNIEnv (tests/jni_min/mock_env.c) -- it is
 * NOT binary compatible with a JVM: the real table has some 230 entries in a fixed order.  A maintainer builds the glue
 * against $JAVA_HOME/include/jni.h; nothing under ahocorasick_amd/ includes this file.
 */
#ifndef ACGPU_TESTS_JNI_MIN_H
#define ACGPU_TESTS_JNI_MIN_H

#include <stdarg.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* primitive types (specification, table 3-1) */
typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_COMMIT 1
#define JNI_ABORT 2

/* reference types: opaque pointers, as in C builds of the real header */
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbooleanArray;
typedef jarray jcharArray;
typedef jarray jintArray;
typedef jarray jobjectArray;

struct _jmethodID;
typedef struct _jmethodID *jmethodID;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNIIMPORT
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

/* the entries the glue uses (names and signatures: specification chapter 4) */
struct JNINativeInterface_ {
    void *mock; /* (the mock environment's own state; the real table begins with reserved slots as well) */

    jclass (JNICALL *FindClass)(JNIEnv *env, const char *name);
    jint (JNICALL *Throw)(JNIEnv *env, jthrowable obj);
    jint (JNICALL *ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jboolean (JNICALL *ExceptionCheck)(JNIEnv *env);
    void (JNICALL *DeleteLocalRef)(JNIEnv *env, jobject obj);

    jobject (JNICALL *NewObject)(JNIEnv *env, jclass clazz, jmethodID methodID, ...);
    jmethodID (JNICALL *GetMethodID)(JNIEnv *env, jclass clazz, const char *name, const char *sig);
    jobject (JNICALL *CallObjectMethod)(JNIEnv *env, jobject obj, jmethodID methodID, ...);

    jstring (JNICALL *NewStringUTF)(JNIEnv *env, const char *utf);
    jsize (JNICALL *GetStringLength)(JNIEnv *env, jstring str);
    void (JNICALL *GetStringRegion)(JNIEnv *env, jstring str, jsize start, jsize len, jchar *buf);

    jsize (JNICALL *GetArrayLength)(JNIEnv *env, jarray array);
    jobject (JNICALL *GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);

    jintArray (JNICALL *NewIntArray)(JNIEnv *env, jsize len);
    void (JNICALL *GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
    void (JNICALL *SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (JNICALL *GetCharArrayRegion)(JNIEnv *env, jcharArray array, jsize start, jsize len, jchar *buf);

    jboolean *(JNICALL *GetBooleanArrayElements)(JNIEnv *env, jbooleanArray array, jboolean *isCopy);
    void (JNICALL *ReleaseBooleanArrayElements)(JNIEnv *env, jbooleanArray array, jboolean *elems, jint mode);
    jchar *(JNICALL *GetCharArrayElements)(JNIEnv *env, jcharArray array, jboolean *isCopy);
    void (JNICALL *ReleaseCharArrayElements)(JNIEnv *env, jcharArray array, jchar *elems, jint mode);
};

#ifdef __cplusplus
}
#endif
#endif
