/*
 * mock_env.c -- a mock JNIEnv and a C harness around ahocorasick_amd/java/jni/acgpu_jni.c (TEST INFRASTRUCTURE).
 *
 * The glue is #included below, so that its static helpers are reachable and it is compiled with this file's flags
 * (tests/test_jni_glue.py: gcc -Wall -Wextra -Werror, with and without -fsanitize=address,undefined).  The mock implements the
 * JNI functions the glue calls with the behaviour the JNI specification gives them -- pending exceptions, bounds checks of
 * the region functions, copies handed out by Get<Type>ArrayElements that must be released -- and counts what a JVM would
 * not forgive: a JNI call (other than the few the specification allows) made while an exception is pending, array elements
 * that were never released.  The jh_* entry points take plain C arrays (the tests call them through ctypes), build the mock
 * Strings and arrays, call the glue's Java_..._NativeAutomaton_* functions and hand back the int[] or the pending exception.
 *
 * Linked against libacgpu.so (GPU box: the results must equal the ctypes binding's) or against tests/jni_min/stub_acgpu.c (the
 * CPU suite, under the sanitizers).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jni.h"

#include "../../ahocorasick_amd/java/jni/acgpu_jni.c"

/* ---- mock objects ------------------------------------------------------------------------------------------------------- */
enum { K_STRING = 1, K_CLASS, K_OBJARRAY, K_INTARRAY, K_CHARARRAY, K_BOOLARRAY, K_THROWABLE };

struct _jobject {
    int kind;
    jsize len;           /* strings, arrays */
    jchar *chars;        /* K_STRING, K_CHARARRAY */
    jint *ints;          /* K_INTARRAY */
    jboolean *bools;     /* K_BOOLARRAY */
    jobject *elems;      /* K_OBJARRAY (borrowed references) */
    char name[96];       /* K_CLASS: binary name; K_THROWABLE: its class */
    jobject message;     /* K_THROWABLE: a K_STRING or NULL */
    struct _jobject *next;
};
struct _jmethodID {
    char name[32];
};

static struct _jobject *g_objs; /* everything created during one harness call; freed when it returns */
static jobject g_pending;       /* the pending exception */
static long long g_violations, g_outstanding, g_local_refs;
static long long g_int_array_limit = -1; /* NewIntArray fails (OutOfMemoryError) above this length; -1: never */
static char g_exc_class[96];
static jchar *g_exc_msg;
static jsize g_exc_msg_len;
static struct _jmethodID g_mid_concat = {"concat"}, g_mid_init = {"<init>"};

static jobject new_obj(int kind) {
    struct _jobject *o = (struct _jobject *)calloc(1, sizeof(*o));
    if (!o) abort();
    o->kind = kind;
    o->next = g_objs;
    g_objs = o;
    return o;
}
static jobject new_string(const jchar *chars, jsize len) {
    jobject s = new_obj(K_STRING);
    s->len = len;
    s->chars = (jchar *)malloc((size_t)(len ? len : 1) * sizeof(jchar));
    if (!s->chars) abort();
    if (len) memcpy(s->chars, chars, (size_t)len * sizeof(jchar));
    return s;
}
static void free_objs(void) {
    while (g_objs) {
        struct _jobject *o = g_objs;
        g_objs = o->next;
        free(o->chars);
        free(o->ints);
        free(o->bools);
        free(o->elems);
        free(o);
    }
}
static void set_pending(const char *cls, jobject message) {
    jobject t = new_obj(K_THROWABLE);
    snprintf(t->name, sizeof(t->name), "%s", cls);
    t->message = message;
    g_pending = t;
}
static jobject ascii_string(const char *s) {
    const size_t n = strlen(s);
    jchar *tmp = (jchar *)malloc((n ? n : 1) * sizeof(jchar));
    if (!tmp) abort();
    for (size_t i = 0; i < n; i++) tmp[i] = (jchar)(unsigned char)s[i];
    jobject r = new_string(tmp, (jsize)n);
    free(tmp);
    return r;
}
/* "JNI functions other than ExceptionOccurred/Describe/Clear/Check, ReleaseXxx, DeleteXxxRef ... must not be called while an
 * exception is pending" (specification, chapter 2, Exception Handling) */
static void no_pending(const char *fn) {
    if (g_pending) {
        g_violations++;
        fprintf(stderr, "[mock JNIEnv] %s called with an exception pending (%s)\n", fn, g_pending->name);
    }
}

/* ---- the function table --------------------------------------------------------------------------------------------------- */
static jclass JNICALL m_FindClass(JNIEnv *env, const char *name) {
    (void)env;
    no_pending("FindClass");
    jobject c = new_obj(K_CLASS);
    snprintf(c->name, sizeof(c->name), "%s", name);
    return c;
}
static jint JNICALL m_Throw(JNIEnv *env, jthrowable obj) {
    (void)env;
    no_pending("Throw");
    g_pending = obj;
    return 0;
}
static jint JNICALL m_ThrowNew(JNIEnv *env, jclass clazz, const char *msg) {
    (void)env;
    no_pending("ThrowNew");
    set_pending(clazz->name, msg ? ascii_string(msg) : NULL);
    return 0;
}
static jboolean JNICALL m_ExceptionCheck(JNIEnv *env) {
    (void)env;
    return g_pending ? JNI_TRUE : JNI_FALSE;
}
static void JNICALL m_DeleteLocalRef(JNIEnv *env, jobject obj) {
    (void)env;
    if (obj) g_local_refs--;
}
static jobject JNICALL m_NewObject(JNIEnv *env, jclass clazz, jmethodID methodID, ...) {
    (void)env;
    no_pending("NewObject");
    if (methodID != &g_mid_init) {
        g_violations++;
        return NULL;
    }
    va_list ap;
    va_start(ap, methodID);
    jobject msg = va_arg(ap, jobject); /* (the one constructor the glue calls: Throwable(String)) */
    va_end(ap);
    jobject t = new_obj(K_THROWABLE);
    snprintf(t->name, sizeof(t->name), "%s", clazz->name);
    t->message = msg;
    return t;
}
static jmethodID JNICALL m_GetMethodID(JNIEnv *env, jclass clazz, const char *name, const char *sig) {
    (void)env;
    (void)sig;
    no_pending("GetMethodID");
    if (!strcmp(name, "concat") && !strcmp(clazz->name, "java/lang/String")) return &g_mid_concat;
    if (!strcmp(name, "<init>")) return &g_mid_init;
    set_pending("java/lang/NoSuchMethodError", ascii_string(name));
    return NULL;
}
static jobject JNICALL m_CallObjectMethod(JNIEnv *env, jobject obj, jmethodID methodID, ...) {
    (void)env;
    no_pending("CallObjectMethod");
    if (methodID != &g_mid_concat || !obj || obj->kind != K_STRING) {
        g_violations++;
        return NULL;
    }
    va_list ap;
    va_start(ap, methodID);
    jobject tail = va_arg(ap, jobject);
    va_end(ap);
    jobject r = new_obj(K_STRING); /* String.concat */
    r->len = obj->len + tail->len;
    r->chars = (jchar *)malloc((size_t)(r->len ? r->len : 1) * sizeof(jchar));
    if (!r->chars) abort();
    memcpy(r->chars, obj->chars, (size_t)obj->len * sizeof(jchar));
    memcpy(r->chars + obj->len, tail->chars, (size_t)tail->len * sizeof(jchar));
    g_local_refs++;
    return r;
}
static jstring JNICALL m_NewStringUTF(JNIEnv *env, const char *utf) {
    (void)env;
    no_pending("NewStringUTF");
    g_local_refs++;
    return ascii_string(utf);
}
static jsize JNICALL m_GetStringLength(JNIEnv *env, jstring str) {
    (void)env;
    no_pending("GetStringLength");
    return str->len;
}
static void JNICALL m_GetStringRegion(JNIEnv *env, jstring str, jsize start, jsize len, jchar *buf) {
    (void)env;
    no_pending("GetStringRegion");
    if (start < 0 || len < 0 || (long long)start + len > str->len) { /* "THROWS StringIndexOutOfBoundsException: on index overflow" */
        set_pending("java/lang/StringIndexOutOfBoundsException", NULL);
        return;
    }
    if (len) memcpy(buf, str->chars + start, (size_t)len * sizeof(jchar));
}
static jsize JNICALL m_GetArrayLength(JNIEnv *env, jarray array) {
    (void)env;
    no_pending("GetArrayLength");
    return array->len;
}
static jobject JNICALL m_GetObjectArrayElement(JNIEnv *env, jobjectArray array, jsize index) {
    (void)env;
    no_pending("GetObjectArrayElement");
    if (index < 0 || index >= array->len) {
        set_pending("java/lang/ArrayIndexOutOfBoundsException", NULL);
        return NULL;
    }
    if (array->elems[index]) g_local_refs++;
    return array->elems[index];
}
static jintArray JNICALL m_NewIntArray(JNIEnv *env, jsize len) {
    (void)env;
    no_pending("NewIntArray");
    if (len < 0 || (g_int_array_limit >= 0 && len > g_int_array_limit)) {
        set_pending("java/lang/OutOfMemoryError", ascii_string("Java heap space"));
        return NULL;
    }
    jobject a = new_obj(K_INTARRAY);
    a->len = len;
    a->ints = (jint *)calloc((size_t)(len ? len : 1), sizeof(jint));
    if (!a->ints) abort();
    return a;
}
static void JNICALL m_GetIntArrayRegion(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf) {
    (void)env;
    no_pending("GetIntArrayRegion");
    if (start < 0 || len < 0 || (long long)start + len > array->len) {
        set_pending("java/lang/ArrayIndexOutOfBoundsException", NULL);
        return;
    }
    if (len) memcpy(buf, array->ints + start, (size_t)len * sizeof(jint));
}
static void JNICALL m_SetIntArrayRegion(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf) {
    (void)env;
    no_pending("SetIntArrayRegion");
    if (start < 0 || len < 0 || (long long)start + len > array->len) {
        set_pending("java/lang/ArrayIndexOutOfBoundsException", NULL);
        return;
    }
    if (len) memcpy(array->ints + start, buf, (size_t)len * sizeof(jint));
}
static void JNICALL m_GetCharArrayRegion(JNIEnv *env, jcharArray array, jsize start, jsize len, jchar *buf) {
    (void)env;
    no_pending("GetCharArrayRegion");
    if (start < 0 || len < 0 || (long long)start + len > array->len) {
        set_pending("java/lang/ArrayIndexOutOfBoundsException", NULL);
        return;
    }
    if (len) memcpy(buf, array->chars + start, (size_t)len * sizeof(jchar));
}
/* Get<PrimitiveType>ArrayElements: a COPY (isCopy = JNI_TRUE is always allowed), so that a glue that writes through it, or
 * never releases it, is caught (the copy is freed by the release, whatever the mode) */
static jboolean *JNICALL m_GetBooleanArrayElements(JNIEnv *env, jbooleanArray array, jboolean *isCopy) {
    (void)env;
    no_pending("GetBooleanArrayElements");
    jboolean *c = (jboolean *)malloc((size_t)(array->len ? array->len : 1));
    if (!c) abort();
    memcpy(c, array->bools, (size_t)array->len);
    if (isCopy) *isCopy = JNI_TRUE;
    g_outstanding++;
    return c;
}
static void JNICALL m_ReleaseBooleanArrayElements(JNIEnv *env, jbooleanArray array, jboolean *elems, jint mode) {
    (void)env;
    if (mode != JNI_ABORT) memcpy(array->bools, elems, (size_t)array->len);
    if (mode != JNI_COMMIT) {
        free(elems);
        g_outstanding--;
    }
}
static jchar *JNICALL m_GetCharArrayElements(JNIEnv *env, jcharArray array, jboolean *isCopy) {
    (void)env;
    no_pending("GetCharArrayElements");
    jchar *c = (jchar *)malloc((size_t)(array->len ? array->len : 1) * sizeof(jchar));
    if (!c) abort();
    memcpy(c, array->chars, (size_t)array->len * sizeof(jchar));
    if (isCopy) *isCopy = JNI_TRUE;
    g_outstanding++;
    return c;
}
static void JNICALL m_ReleaseCharArrayElements(JNIEnv *env, jcharArray array, jchar *elems, jint mode) {
    (void)env;
    if (mode != JNI_ABORT) memcpy(array->chars, elems, (size_t)array->len * sizeof(jchar));
    if (mode != JNI_COMMIT) {
        free(elems);
        g_outstanding--;
    }
}

static const struct JNINativeInterface_ g_table = {
    NULL,
    m_FindClass, m_Throw, m_ThrowNew, m_ExceptionCheck, m_DeleteLocalRef,
    m_NewObject, m_GetMethodID, m_CallObjectMethod,
    m_NewStringUTF, m_GetStringLength, m_GetStringRegion,
    m_GetArrayLength, m_GetObjectArrayElement,
    m_NewIntArray, m_GetIntArrayRegion, m_SetIntArrayRegion, m_GetCharArrayRegion,
    m_GetBooleanArrayElements, m_ReleaseBooleanArrayElements, m_GetCharArrayElements, m_ReleaseCharArrayElements,
};
static JNIEnv g_env = &g_table;

/* ---- the harness: plain C in, plain C out ---------------------------------------------------------------------------------- */
#define JH __attribute__((visibility("default")))

static void begin_call(void) {
    g_pending = NULL;
    g_exc_class[0] = 0;
    free(g_exc_msg);
    g_exc_msg = NULL;
    g_exc_msg_len = -1;
}
/* what the native method left behind: the pending exception is copied out, every mock object of the call is freed */
static void end_call(void) {
    if (g_pending) {
        snprintf(g_exc_class, sizeof(g_exc_class), "%s", g_pending->name);
        if (g_pending->message) {
            g_exc_msg_len = g_pending->message->len;
            g_exc_msg = (jchar *)malloc((size_t)(g_exc_msg_len ? g_exc_msg_len : 1) * sizeof(jchar));
            if (!g_exc_msg) abort();
            memcpy(g_exc_msg, g_pending->message->chars, (size_t)g_exc_msg_len * sizeof(jchar));
        }
    }
    g_pending = NULL;
    free_objs();
}
/* the returned int[] as a malloc'd copy (released with jh_release); -1: the method returned null */
static long long take_int_array(jintArray a, int32_t **out) {
    *out = NULL;
    if (!a) return -1;
    *out = (int32_t *)malloc((size_t)(a->len ? a->len : 1) * sizeof(int32_t));
    if (!*out) abort();
    memcpy(*out, a->ints, (size_t)a->len * sizeof(int32_t));
    return a->len;
}
static jobject strings_array(const uint16_t *units, const uint64_t *off, const uint8_t *is_null, int n) {
    jobject arr = new_obj(K_OBJARRAY);
    arr->len = n;
    arr->elems = (jobject *)calloc((size_t)(n ? n : 1), sizeof(jobject));
    if (!arr->elems) abort();
    for (int i = 0; i < n; i++)
        arr->elems[i] = (is_null && is_null[i]) ? NULL : new_string(units + off[i], (jsize)(off[i + 1] - off[i]));
    return arr;
}

/* class name of the exception the last call left pending ("" = none); its message (UTF-16) through jh_exception_message */
JH const char *jh_exception_class(void) { return g_exc_class; }
JH long long jh_exception_message(uint16_t *buf, long long cap) { /* -1: no message */
    if (g_exc_msg_len < 0) return -1;
    for (long long i = 0; i < g_exc_msg_len && i < cap; i++) buf[i] = g_exc_msg[i];
    return g_exc_msg_len;
}
JH long long jh_violations(void) { return g_violations; }
JH long long jh_outstanding_elements(void) { return g_outstanding; }
JH void jh_set_int_array_limit(long long n) { g_int_array_limit = n; }
JH void jh_release(int32_t *p) { free(p); }

/* table_len: entries of the lower / wordChars arrays handed to the glue (65536; less: the glue must refuse them) */
JH long long jh_build(int mode, const uint16_t *units, const uint64_t *off, const uint8_t *is_null, int n_kw, int cs, const uint16_t *lower,
                      const uint8_t *wordchars, int table_len) {
    begin_call();
    jobject kws = strings_array(units, off, is_null, n_kw);
    jobject lo = NULL, wc = NULL;
    if (lower) {
        lo = new_obj(K_CHARARRAY);
        lo->len = table_len;
        lo->chars = (jchar *)malloc((size_t)table_len * sizeof(jchar));
        if (!lo->chars) abort();
        memcpy(lo->chars, lower, (size_t)table_len * sizeof(jchar));
    }
    if (wordchars) {
        wc = new_obj(K_BOOLARRAY);
        wc->len = table_len;
        wc->bools = (jboolean *)malloc((size_t)table_len);
        if (!wc->bools) abort();
        memcpy(wc->bools, wordchars, (size_t)table_len);
    }
    const jlong h = Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_build(&g_env, NULL, mode, kws, (jboolean)(cs ? 1 : 0), lo, wc);
    end_call();
    return (long long)h;
}
JH void jh_free(long long handle) {
    begin_call();
    Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_free(&g_env, NULL, (jlong)handle);
    end_call();
}
/* n < 0: a null haystack; devices == NULL: no device list */
JH long long jh_match(long long handle, const uint16_t *hay, long long n, int with_ids, const int32_t *devices, int n_devices, int32_t **out) {
    begin_call();
    jobject s = n < 0 ? NULL : new_string(hay, (jsize)n);
    jobject d = NULL;
    if (devices) {
        d = new_obj(K_INTARRAY);
        d->len = n_devices;
        d->ints = (jint *)malloc((size_t)(n_devices ? n_devices : 1) * sizeof(jint));
        if (!d->ints) abort();
        memcpy(d->ints, devices, (size_t)n_devices * sizeof(jint));
    }
    jintArray r = Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_match(&g_env, NULL, (jlong)handle, s, (jboolean)(with_ids ? 1 : 0), d);
    const long long k = take_int_array(r, out);
    end_call();
    return k;
}
JH long long jh_match_batch(long long handle, const uint16_t *units, const uint64_t *off, const uint8_t *is_null, int n_hay, int with_ids, int32_t **out) {
    begin_call();
    jobject arr = n_hay < 0 ? NULL : strings_array(units, off, is_null, n_hay);
    jintArray r = Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_matchBatch(&g_env, NULL, (jlong)handle, arr, (jboolean)(with_ids ? 1 : 0));
    const long long k = take_int_array(r, out);
    end_call();
    return k;
}
JH long long jh_stream_open(long long handle, int pipelined) {
    begin_call();
    const jlong s = Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamOpen(&g_env, NULL, (jlong)handle, (jboolean)(pipelined ? 1 : 0));
    end_call();
    return (long long)s;
}
/* array_len: length of the char[] handed over (>= length, as a Reader's buffer is); length: the chars of it that count */
JH long long jh_stream_feed(long long stream, const uint16_t *chunk, int array_len, int length, int last, int pipelined, int32_t **out) {
    begin_call();
    jobject a = new_obj(K_CHARARRAY);
    a->len = array_len;
    a->chars = (jchar *)malloc((size_t)(array_len ? array_len : 1) * sizeof(jchar));
    if (!a->chars) abort();
    if (array_len) memcpy(a->chars, chunk, (size_t)array_len * sizeof(jchar));
    jintArray r = Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamFeed(&g_env, NULL, (jlong)stream, a, length, (jboolean)(last ? 1 : 0),
                                                                                    (jboolean)(pipelined ? 1 : 0));
    const long long k = take_int_array(r, out);
    end_call();
    return k;
}
JH void jh_stream_close(long long stream) {
    begin_call();
    Java_com_roklenarcic_util_strings_gpu_NativeAutomaton_streamClose(&g_env, NULL, (jlong)stream);
    end_call();
}
/* the glue's records -> int[] step on its own: a record list beyond what an int[] holds must be refused BEFORE any allocation
 * (nothing is read from the buffer when it is: NULL is passed) */
JH long long jh_to_int_array(unsigned long long n_ints) {
    begin_call();
    static const jint one[4] = {1, 2, 3, 4};
    jintArray r = to_int_array(&g_env, n_ints <= 4 ? (const void *)one : NULL, (uint64_t)n_ints);
    const long long k = r ? r->len : -1;
    end_call();
    return k;
}
