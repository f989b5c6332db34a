/*
 * acgpu_jni.c -- JNI glue between com.roklenarcic.util.strings.gpu.NativeAutomaton and the C ABI (include/acgpu.h).
 * Build (on a machine with a JDK; not possible in the build image):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../../include \
 *       -o libacgpu_jni.so acgpu_jni.c -L../../lib -lacgpu -Wl,-rpath,'$ORIGIN'
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "acgpu.h"

static void throw_new(JNIEnv *env, const char *cls, const char *msg) {
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, msg);
}

JNIEXPORT jlong JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_build(JNIEnv *env, jclass cls, jint mode,
                                                                                      jobjectArray keywords, jboolean cs,
                                                                                      jcharArray lower, jbooleanArray wordChars) {
    (void)cls;
    jsize n = (*env)->GetArrayLength(env, keywords);
    uint64_t *off = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    uint64_t total = 0;
    for (jsize i = 0; i < n; i++) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, keywords, i);
        total += s ? (uint64_t)(*env)->GetStringLength(env, s) : 0; /* null keyword == empty range: skipped */
        off[i + 1] = total;
        if (s) (*env)->DeleteLocalRef(env, s);
    }
    uint16_t *units = (uint16_t *)malloc((size_t)(total ? total : 1) * sizeof(uint16_t));
    for (jsize i = 0; i < n; i++) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, keywords, i);
        if (s) {
            (*env)->GetStringRegion(env, s, 0, (jsize)(off[i + 1] - off[i]), (jchar *)(units + off[i]));
            (*env)->DeleteLocalRef(env, s);
        }
    }
    jchar *lo = lower ? (*env)->GetCharArrayElements(env, lower, NULL) : NULL;
    uint8_t *wc = NULL;
    if (wordChars) {
        jboolean *b = (*env)->GetBooleanArrayElements(env, wordChars, NULL);
        wc = (uint8_t *)malloc(65536);
        for (int i = 0; i < 65536; i++) wc[i] = b[i] ? 1 : 0;
        (*env)->ReleaseBooleanArrayElements(env, wordChars, b, JNI_ABORT);
    }
    acgpu_automaton *a = NULL;
    int64_t bad = -1;
    int rc = acgpu_build(mode, units, off, (uint32_t)n, cs ? 1 : 0, (const uint16_t *)lo, wc, &a, &bad);
    if (lo) (*env)->ReleaseCharArrayElements(env, lower, lo, JNI_ABORT);
    free(wc);
    if (rc == ACGPU_E_NONWORD) {
        /* the reference's message: keyword + " contains non-word characters." (S/WholeWordMatchMap.java:265) */
        char msg[256] = "keyword contains non-word characters.";
        throw_new(env, "java/lang/IllegalArgumentException", msg);
    } else if (rc != ACGPU_OK) {
        throw_new(env, "java/lang/IllegalStateException", acgpu_strerror(rc));
    }
    free(units);
    free(off);
    return (jlong)(intptr_t)a;
}

JNIEXPORT jintArray JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_match(JNIEnv *env, jclass cls, jlong handle,
                                                                                          jstring haystack, jboolean withIds) {
    (void)cls;
    if (!haystack) {
        throw_new(env, "java/lang/NullPointerException", "haystack"); /* reference: haystack.length() on null */
        return NULL;
    }
    const acgpu_automaton *a = (const acgpu_automaton *)(intptr_t)handle;
    jsize n = (*env)->GetStringLength(env, haystack);
    const jchar *units = (*env)->GetStringCritical(env, haystack, NULL); /* no copy on most JVMs */
    const int kind = withIds ? ACGPU_REC_MAP : ACGPU_REC_SET;
    uint64_t cap = (uint64_t)n / 64 + 4096, n_out = 0;
    void *buf = malloc(cap * (size_t)kind);
    int rc = acgpu_match_u16(a, (const uint16_t *)units, (uint64_t)n, kind, buf, cap, &n_out);
    if (rc == ACGPU_E_OVERFLOW) { /* retry once with the exact capacity */
        cap = n_out;
        buf = realloc(buf, cap * (size_t)kind);
        rc = acgpu_match_u16(a, (const uint16_t *)units, (uint64_t)n, kind, buf, cap, &n_out);
    }
    (*env)->ReleaseStringCritical(env, haystack, units);
    jintArray out = NULL;
    if (rc == ACGPU_OK) {
        out = (*env)->NewIntArray(env, (jsize)(n_out * (uint64_t)(kind / 4)));
        if (out) (*env)->SetIntArrayRegion(env, out, 0, (jsize)(n_out * (uint64_t)(kind / 4)), (const jint *)buf);
    } else {
        throw_new(env, "java/lang/IllegalStateException", acgpu_strerror(rc));
    }
    free(buf);
    return out;
}

JNIEXPORT void JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_free(JNIEnv *env, jclass cls, jlong handle) {
    (void)env;
    (void)cls;
    acgpu_free((acgpu_automaton *)(intptr_t)handle);
}

/* ---- match(Readable, ReadableMatchListener<T>): acgpu_stream_* ---- */
JNIEXPORT jlong JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamOpen(JNIEnv *env, jclass cls, jlong handle) {
    (void)cls;
    acgpu_stream *s = NULL;
    int rc = acgpu_stream_open((const acgpu_automaton *)(intptr_t)handle, &s);
    if (rc != ACGPU_OK) throw_new(env, rc == ACGPU_E_UNSUPPORTED ? "java/lang/UnsupportedOperationException" : "java/lang/IllegalStateException",
                                  acgpu_strerror(rc));
    return (jlong)(intptr_t)s;
}

JNIEXPORT jintArray JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamFeed(JNIEnv *env, jclass cls, jlong stream,
                                                                                               jcharArray chunk, jint length,
                                                                                               jboolean last) {
    (void)cls;
    acgpu_stream *s = (acgpu_stream *)(intptr_t)stream;
    jchar *units = (*env)->GetCharArrayElements(env, chunk, NULL);
    uint64_t cap = (uint64_t)length / 64 + 4096, n_out = 0;
    int64_t base = 0;
    int32_t *buf = (int32_t *)malloc(cap * ACGPU_REC_MAP);
    int rc = acgpu_stream_feed(s, (const uint16_t *)units, (uint64_t)length, last ? 1 : 0, ACGPU_REC_MAP, buf, cap, &n_out, &base);
    if (rc == ACGPU_E_OVERFLOW) { /* nothing was consumed: same feed, exact capacity */
        cap = n_out;
        buf = (int32_t *)realloc(buf, cap * ACGPU_REC_MAP);
        rc = acgpu_stream_feed(s, (const uint16_t *)units, (uint64_t)length, last ? 1 : 0, ACGPU_REC_MAP, buf, cap, &n_out, &base);
    }
    (*env)->ReleaseCharArrayElements(env, chunk, units, JNI_ABORT);
    jintArray out = NULL;
    if (rc == ACGPU_OK) {
        for (uint64_t i = 0; i < n_out; i++) buf[i] = buf[3 * i + 2]; /* the Readable listener only sees the value */
        out = (*env)->NewIntArray(env, (jsize)n_out);
        if (out) (*env)->SetIntArrayRegion(env, out, 0, (jsize)n_out, (const jint *)buf);
    } else {
        throw_new(env, "java/lang/IllegalStateException", acgpu_strerror(rc));
    }
    free(buf);
    return out;
}

JNIEXPORT void JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamClose(JNIEnv *env, jclass cls, jlong stream) {
    (void)env;
    (void)cls;
    acgpu_stream_close((acgpu_stream *)(intptr_t)stream);
}
