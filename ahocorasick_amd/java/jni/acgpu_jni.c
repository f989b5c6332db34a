/*
 * acgpu_jni.c -- JNI glue between com.roklenarcic.util.strings.gpu.NativeAutomaton and the C ABI (include/acgpu.h).
 * Build (on a machine with a JDK; not possible in the build image):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../../include \
 *       -o libacgpu_jni.so acgpu_jni.c -L../../lib -lacgpu -Wl,-rpath,'$ORIGIN'
 *
 * Rules this file keeps:
 *  - no JNI critical region is ever open across a call into libacgpu (those calls take a mutex, allocate device memory,
 *    copy over PCIe and wait for a stream; a critical region would lock the garbage collector out for all of that): the
 *    haystack is COPIED out of the String with GetStringRegion, in bounded slices, into a native buffer;
 *  - every allocation is checked (java.lang.OutOfMemoryError), every JNI call that can leave an exception pending is
 *    followed by a check;
 *  - a result that does not fit a Java int[] is an error, not a truncated array.
 */
#include <jni.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "acgpu.h"

#define REGION_SLICE (32 * 1024 * 1024) /* chars per GetStringRegion call (64 MiB) */

static void throw_new(JNIEnv *env, const char *cls, const char *msg) {
    if ((*env)->ExceptionCheck(env)) return; /* keep the first one */
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, msg);
}

static void throw_oom(JNIEnv *env, const char *what) { throw_new(env, "java/lang/OutOfMemoryError", what); }

static void throw_rc(JNIEnv *env, int rc) {
    if (rc == ACGPU_E_NOMEM) throw_oom(env, acgpu_strerror(rc));
    else if (rc == ACGPU_E_UNSUPPORTED) throw_new(env, "java/lang/UnsupportedOperationException", acgpu_strerror(rc));
    else throw_new(env, "java/lang/IllegalStateException", acgpu_strerror(rc));
}

/* IllegalArgumentException(keyword + " contains non-word characters.") -- the reference's message,
 * S/WholeWordMatchMap.java:265 -- built from the String itself (any characters, any length) */
static void throw_nonword(JNIEnv *env, jobjectArray keywords, int64_t bad) {
    jstring kw = (bad >= 0 && bad < (int64_t)(*env)->GetArrayLength(env, keywords))
                     ? (jstring)(*env)->GetObjectArrayElement(env, keywords, (jsize)bad) : NULL;
    jclass scls = (*env)->FindClass(env, "java/lang/String");
    jclass ecls = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (!kw || !scls || !ecls) {
        throw_new(env, "java/lang/IllegalArgumentException", "keyword contains non-word characters.");
        return;
    }
    jmethodID concat = (*env)->GetMethodID(env, scls, "concat", "(Ljava/lang/String;)Ljava/lang/String;");
    jmethodID ctor = (*env)->GetMethodID(env, ecls, "<init>", "(Ljava/lang/String;)V");
    jstring tail = (*env)->NewStringUTF(env, " contains non-word characters.");
    if (!concat || !ctor || !tail) return; /* (an exception is pending) */
    jstring msg = (jstring)(*env)->CallObjectMethod(env, kw, concat, tail);
    if ((*env)->ExceptionCheck(env) || !msg) return;
    jthrowable ex = (jthrowable)(*env)->NewObject(env, ecls, ctor, msg);
    if (ex) (*env)->Throw(env, ex);
}

JNIEXPORT jlong JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_build(JNIEnv *env, jclass cls, jint mode,
                                                                                      jobjectArray keywords, jboolean cs,
                                                                                      jcharArray lower, jbooleanArray wordChars) {
    (void)cls;
    const jsize n = (*env)->GetArrayLength(env, keywords);
    uint64_t *off = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    uint16_t *units = NULL;
    uint8_t *wc = NULL;
    jchar *lo = NULL;
    acgpu_automaton *a = NULL;
    if (!off) { throw_oom(env, "keyword offsets"); return 0; }
    uint64_t total = 0;
    for (jsize i = 0; i < n; i++) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, keywords, i);
        total += s ? (uint64_t)(*env)->GetStringLength(env, s) : 0; /* null keyword == empty range: skipped */
        off[i + 1] = total;
        if (s) (*env)->DeleteLocalRef(env, s);
    }
    units = (uint16_t *)malloc((size_t)(total ? total : 1) * sizeof(uint16_t));
    if (!units) { throw_oom(env, "keyword units"); goto done; }
    for (jsize i = 0; i < n; i++) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, keywords, i);
        if (s) {
            (*env)->GetStringRegion(env, s, 0, (jsize)(off[i + 1] - off[i]), (jchar *)(units + off[i]));
            (*env)->DeleteLocalRef(env, s);
            if ((*env)->ExceptionCheck(env)) goto done;
        }
    }
    if (wordChars) {
        if ((*env)->GetArrayLength(env, wordChars) < 65536) {
            throw_new(env, "java/lang/IllegalArgumentException", "wordChars must have 65536 entries");
            goto done;
        }
        jboolean *b = (*env)->GetBooleanArrayElements(env, wordChars, NULL);
        if (!b) goto done; /* OutOfMemoryError pending */
        wc = (uint8_t *)malloc(65536);
        if (wc) for (int i = 0; i < 65536; i++) wc[i] = b[i] ? 1 : 0;
        (*env)->ReleaseBooleanArrayElements(env, wordChars, b, JNI_ABORT);
        if (!wc) { throw_oom(env, "word-character table"); goto done; }
    }
    if (lower) {
        if ((*env)->GetArrayLength(env, lower) < 65536) {
            throw_new(env, "java/lang/IllegalArgumentException", "lower must have 65536 entries");
            goto done;
        }
        lo = (*env)->GetCharArrayElements(env, lower, NULL); /* (not a critical region: acgpu_build may take a while) */
        if (!lo) goto done;
    }
    {
        int64_t bad = -1;
        const int rc = acgpu_build(mode, units, off, (uint32_t)n, cs ? 1 : 0, (const uint16_t *)lo, wc, &a, &bad);
        if (rc == ACGPU_E_NONWORD) throw_nonword(env, keywords, bad);
        else if (rc != ACGPU_OK) throw_rc(env, rc);
    }
done:
    if (lo) (*env)->ReleaseCharArrayElements(env, lower, lo, JNI_ABORT);
    free(wc);
    free(units);
    free(off);
    return (jlong)(intptr_t)a;
}

/* records -> int[]; a result beyond Integer.MAX_VALUE ints cannot be represented */
static jintArray to_int_array(JNIEnv *env, const void *buf, uint64_t n_ints) {
    if (n_ints > (uint64_t)INT_MAX - 8) {
        throw_new(env, "java/lang/IllegalStateException", "more match records than a Java int[] can hold; scan the haystack in parts");
        return NULL;
    }
    jintArray out = (*env)->NewIntArray(env, (jsize)n_ints);
    if (out && n_ints) (*env)->SetIntArrayRegion(env, out, 0, (jsize)n_ints, (const jint *)buf);
    return out; /* NULL: OutOfMemoryError pending */
}

/* devices: null = the current HIP device alone (acgpu_match_u16); else the device list of -Dacgpu.devices -- the haystack is
 * cut into one contiguous share per entry and all of them are scanned at once (acgpu_match_u16_multi, ONE call, one process:
 * the shape of S/StringSet.java:3-5). */
JNIEXPORT jintArray JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_match(JNIEnv *env, jclass cls, jlong handle,
                                                                                          jstring haystack, jboolean withIds,
                                                                                          jintArray devices) {
    (void)cls;
    jint devs[64];
    jsize n_devs = 0;
    if (devices) {
        n_devs = (*env)->GetArrayLength(env, devices);
        if (n_devs < 1 || n_devs > 64) { throw_new(env, "java/lang/IllegalArgumentException", "acgpu.devices: 1..64 device ordinals"); return NULL; }
        (*env)->GetIntArrayRegion(env, devices, 0, n_devs, devs);
        if ((*env)->ExceptionCheck(env)) return NULL;
    }
    if (!haystack) {
        throw_new(env, "java/lang/NullPointerException", "haystack"); /* reference: haystack.length() on null */
        return NULL;
    }
    const acgpu_automaton *a = (const acgpu_automaton *)(intptr_t)handle;
    const jsize n = (*env)->GetStringLength(env, haystack);
    const int kind = withIds ? ACGPU_REC_MAP : ACGPU_REC_SET;
    /* the haystack is copied out of the String (a compact-strings JVM inflates Latin-1 here); no critical region */
    jchar *units = (jchar *)malloc((size_t)(n ? n : 1) * sizeof(jchar));
    if (!units) { throw_oom(env, "haystack copy"); return NULL; }
    for (jsize at = 0; at < n; at += REGION_SLICE) {
        const jsize len = n - at < REGION_SLICE ? n - at : REGION_SLICE;
        (*env)->GetStringRegion(env, haystack, at, len, units + at);
        if ((*env)->ExceptionCheck(env)) { free(units); return NULL; }
    }
    uint64_t cap = (uint64_t)n / 64 + 4096, n_out = 0;
    void *buf = malloc(cap * (size_t)kind);
    jintArray out = NULL;
    if (!buf) { throw_oom(env, "match records"); free(units); return NULL; }
    int rc = n_devs ? acgpu_match_u16_multi(a, (const uint16_t *)units, (uint64_t)n, (const int *)devs, (int)n_devs, kind, buf, cap, &n_out)
                    : acgpu_match_u16(a, (const uint16_t *)units, (uint64_t)n, kind, buf, cap, &n_out);
    if (rc == ACGPU_E_OVERFLOW) { /* retry once with the exact capacity */
        cap = n_out;
        void *bigger = realloc(buf, cap * (size_t)kind);
        if (!bigger) { throw_oom(env, "match records"); free(buf); free(units); return NULL; }
        buf = bigger;
        rc = n_devs ? acgpu_match_u16_multi(a, (const uint16_t *)units, (uint64_t)n, (const int *)devs, (int)n_devs, kind, buf, cap, &n_out)
                    : acgpu_match_u16(a, (const uint16_t *)units, (uint64_t)n, kind, buf, cap, &n_out);
    }
    free(units);
    if (rc == ACGPU_OK) out = to_int_array(env, buf, n_out * (uint64_t)(kind / 4));
    else throw_rc(env, rc);
    free(buf);
    return out;
}

/* many short haystacks in one device call: acgpu_match_batch_u16 */
JNIEXPORT jintArray JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_matchBatch(JNIEnv *env, jclass cls, jlong handle,
                                                                                               jobjectArray haystacks, jboolean withIds) {
    (void)cls;
    if (!haystacks) {
        throw_new(env, "java/lang/NullPointerException", "haystacks");
        return NULL;
    }
    const acgpu_automaton *a = (const acgpu_automaton *)(intptr_t)handle;
    const jsize n = (*env)->GetArrayLength(env, haystacks);
    const int kind = withIds ? ACGPU_REC_MAP : ACGPU_REC_SET;
    uint64_t *off = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    jchar *units = NULL;
    void *buf = NULL;
    jintArray out = NULL;
    if (!off) { throw_oom(env, "haystack offsets"); return NULL; }
    uint64_t total = 0;
    for (jsize i = 0; i < n; i++) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, haystacks, i);
        if (!s) { throw_new(env, "java/lang/NullPointerException", "haystack"); goto done; } /* reference: haystack.length() on null */
        total += (uint64_t)(*env)->GetStringLength(env, s);
        off[i + 1] = total;
        (*env)->DeleteLocalRef(env, s);
    }
    units = (jchar *)malloc((size_t)(total ? total : 1) * sizeof(jchar));
    if (!units) { throw_oom(env, "haystack copies"); goto done; }
    for (jsize i = 0; i < n; i++) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, haystacks, i);
        (*env)->GetStringRegion(env, s, 0, (jsize)(off[i + 1] - off[i]), units + off[i]);
        (*env)->DeleteLocalRef(env, s);
        if ((*env)->ExceptionCheck(env)) goto done;
    }
    {
        uint64_t cap = total / 16 + 4096, n_out = 0;
        buf = malloc(cap * (size_t)(kind + 4));
        if (!buf) { throw_oom(env, "match records"); goto done; }
        int rc = acgpu_match_batch_u16(a, (const uint16_t *)units, off, (uint32_t)n, kind, buf, cap, &n_out);
        if (rc == ACGPU_E_OVERFLOW) { /* retry once with the exact capacity */
            cap = n_out;
            void *bigger = realloc(buf, cap * (size_t)(kind + 4));
            if (!bigger) { throw_oom(env, "match records"); goto done; }
            buf = bigger;
            rc = acgpu_match_batch_u16(a, (const uint16_t *)units, off, (uint32_t)n, kind, buf, cap, &n_out);
        }
        if (rc == ACGPU_OK) out = to_int_array(env, buf, n_out * (uint64_t)(kind / 4 + 1));
        else throw_rc(env, rc);
    }
done:
    free(buf);
    free(units);
    free(off);
    return out;
}

JNIEXPORT void JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_free(JNIEnv *env, jclass cls, jlong handle) {
    (void)env;
    (void)cls;
    acgpu_free((acgpu_automaton *)(intptr_t)handle);
}

/* ---- match(Readable, ReadableMatchListener<T>): acgpu_stream_* ---- */
/* pipelined (-Dacgpu.stream.pipelined=true): acgpu_stream_set_pipelined -- a feed returns the PREVIOUS chunk's records, the last
 * one both.  The chunk is copied out of the Java array into a buffer of this call (GetCharArrayRegion) and the library's pool of
 * copy threads moves it into its pinned staging memory under the previous chunk's scan: 33 GB/s.  (Reading the Java array
 * straight into the staging memory -- acgpu_stream_reserve -- makes that ONE single-threaded copy which no scan overlaps: 16 GB/s,
 * profiles/r04/v6_stream_rate.txt; the entry stays in the ABI for callers that fill the memory with several threads.) */
JNIEXPORT jlong JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamOpen(JNIEnv *env, jclass cls, jlong handle,
                                                                                           jboolean pipelined) {
    (void)cls;
    acgpu_stream *s = NULL;
    int rc = acgpu_stream_open((const acgpu_automaton *)(intptr_t)handle, &s);
    if (rc == ACGPU_OK && pipelined) {
        rc = acgpu_stream_set_pipelined(s, 1);
        if (rc != ACGPU_OK) { acgpu_stream_close(s); s = NULL; }
    }
    if (rc != ACGPU_OK) throw_rc(env, rc);
    return (jlong)(intptr_t)s;
}

JNIEXPORT jintArray JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamFeed(JNIEnv *env, jclass cls, jlong stream,
                                                                                               jcharArray chunk, jint length,
                                                                                               jboolean last, jboolean pipelined) {
    (void)cls;
    acgpu_stream *s = (acgpu_stream *)(intptr_t)stream;
    if (length < 0 || length > (*env)->GetArrayLength(env, chunk)) {
        throw_new(env, "java/lang/ArrayIndexOutOfBoundsException", "length");
        return NULL;
    }
    /* a copy (GetCharArrayRegion), for the same reason as in match(): the feed blocks on the GPU */
    (void)pipelined;
    jchar *units = NULL, *owned = NULL;
    units = owned = (jchar *)malloc((size_t)(length ? length : 1) * sizeof(jchar));
    if (!units) { throw_oom(env, "chunk copy"); return NULL; }
    (*env)->GetCharArrayRegion(env, chunk, 0, length, units);
    uint64_t cap = (uint64_t)length / 64 + 4096, n_out = 0;
    int64_t base = 0;
    int32_t *buf = (int32_t *)malloc(cap * ACGPU_REC_MAP);
    jintArray out = NULL;
    if (!buf) { throw_oom(env, "match records"); free(owned); return NULL; }
    int rc = acgpu_stream_feed(s, (const uint16_t *)units, (uint64_t)length, last ? 1 : 0, ACGPU_REC_MAP, buf, cap, &n_out, &base);
    if (rc == ACGPU_E_OVERFLOW) { /* nothing was consumed: same feed, exact capacity */
        cap = n_out;
        int32_t *bigger = (int32_t *)realloc(buf, cap * ACGPU_REC_MAP);
        if (!bigger) { throw_oom(env, "match records"); free(buf); free(owned); return NULL; }
        buf = bigger;
        rc = acgpu_stream_feed(s, (const uint16_t *)units, (uint64_t)length, last ? 1 : 0, ACGPU_REC_MAP, buf, cap, &n_out, &base);
    }
    free(owned);
    if (rc == ACGPU_OK) {
        for (uint64_t i = 0; i < n_out; i++) buf[i] = buf[3 * i + 2]; /* the Readable listener only sees the value */
        out = to_int_array(env, buf, n_out);
    } else {
        throw_rc(env, rc);
    }
    free(buf);
    return out;
}

JNIEXPORT void JNICALL Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamClose(JNIEnv *env, jclass cls, jlong stream) {
    (void)env;
    (void)cls;
    acgpu_stream_close((acgpu_stream *)(intptr_t)stream);
}
