/*
 * cpu_driver.c -- scenarios for the JNI glue on the CPU (TEST INFRASTRUCTURE): mock JNIEnv (mock_env.c) + the stub of the C ABI
 * (stub_acgpu.c), built with -fsanitize=address,undefined and run with leak detection on by tests/test_jni_glue.py.  Exit code 0:
 * every scenario held, nothing leaked, no JNI call was made with an exception pending, no array elements stayed pinned.
 */
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

const char *jh_exception_class(void);
long long jh_exception_message(uint16_t *buf, long long cap);
long long jh_violations(void);
long long jh_outstanding_elements(void);
void jh_set_int_array_limit(long long n);
void jh_release(int32_t *p);
long long jh_build(int mode, const uint16_t *units, const uint64_t *off, const uint8_t *is_null, int n_kw, int cs, const uint16_t *lower,
                   const uint8_t *wordchars, int table_len);
void jh_free(long long handle);
long long jh_match(long long handle, const uint16_t *hay, long long n, int with_ids, const int32_t *devices, int n_devices, int32_t **out);
long long jh_match_batch(long long handle, const uint16_t *units, const uint64_t *off, const uint8_t *is_null, int n_hay, int with_ids, int32_t **out);
long long jh_stream_open(long long handle, int pipelined);
long long jh_stream_feed(long long stream, const uint16_t *chunk, int array_len, int length, int last, int pipelined, int32_t **out);
void jh_stream_close(long long stream);
long long jh_to_int_array(unsigned long long n_ints);
int stub_last_devices(int *out);
unsigned long long stub_overflows(const void *a);

static int g_failed;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_failed++;                                                      \
        }                                                                    \
    } while (0)
static int exc_is(const char *cls) { return !strcmp(jh_exception_class(), cls); }
static int msg_is(const char *ascii) {
    uint16_t buf[512];
    const long long n = jh_exception_message(buf, 512);
    if (n != (long long)strlen(ascii)) return 0;
    for (long long i = 0; i < n; i++)
        if (buf[i] != (uint16_t)(unsigned char)ascii[i]) return 0;
    return 1;
}
static uint16_t *units_of(const char *s, uint64_t *n) {
    *n = strlen(s);
    uint16_t *u = (uint16_t *)malloc((*n ? *n : 1) * 2);
    for (uint64_t i = 0; i < *n; i++) u[i] = (uint16_t)(unsigned char)s[i];
    return u;
}

int main(void) {
    uint16_t *lower = (uint16_t *)malloc(65536 * 2);
    uint8_t *word = (uint8_t *)malloc(65536);
    for (int i = 0; i < 65536; i++) {
        lower[i] = (uint16_t)((i >= 'A' && i <= 'Z') ? i + 32 : i);
        word[i] = (uint8_t)((i >= 'a' && i <= 'z') || (i >= 'A' && i <= 'Z') || (i >= '0' && i <= '9'));
    }
    /* ---- build: four keywords, one of them null (skipped by the reference: an empty range here) ---- */
    uint64_t nk;
    uint16_t *kw = units_of("heshehers", &nk);
    const uint64_t off[5] = {0, 2, 5, 5, 9};
    const uint8_t nulls[4] = {0, 0, 1, 0};
    long long h = jh_build(0, kw, off, nulls, 4, 1, NULL, NULL, 65536);
    CHECK(h != 0 && exc_is(""));
    /* case-insensitive word matcher with both tables */
    long long hw = jh_build(2, kw, off, nulls, 4, 0, lower, word, 65536);
    CHECK(hw != 0 && exc_is(""));
    /* tables of the wrong length are refused before they are read */
    CHECK(jh_build(2, kw, off, nulls, 4, 0, lower, word, 100) == 0 && exc_is("java/lang/IllegalArgumentException") && msg_is("wordChars must have 65536 entries"));
    CHECK(jh_build(0, kw, off, nulls, 4, 0, lower, NULL, 100) == 0 && exc_is("java/lang/IllegalArgumentException") && msg_is("lower must have 65536 entries"));
    /* a keyword with a non-word character: IllegalArgumentException(keyword + " contains non-word characters.") */
    {
        uint64_t n2;
        uint16_t *k2 = units_of("okno!no", &n2);
        const uint64_t o2[3] = {0, 2, 7};
        CHECK(jh_build(2, k2, o2, NULL, 2, 1, NULL, word, 65536) == 0);
        CHECK(exc_is("java/lang/IllegalArgumentException") && msg_is("no!no contains non-word characters."));
        free(k2);
    }
    /* the library's other refusals: IllegalStateException with its text */
    CHECK(jh_build(9, kw, off, nulls, 4, 1, NULL, NULL, 65536) == 0 && exc_is("java/lang/IllegalStateException") && msg_is("invalid argument"));

    /* ---- match ---- */
    int32_t *out = NULL;
    {
        uint64_t n;
        uint16_t *hay = units_of("axbxxc", &n);
        long long k = jh_match(h, hay, (long long)n, 0, NULL, 0, &out);
        CHECK(k == 6 && out[0] == 1 && out[1] == 2 && out[2] == 3 && out[3] == 4 && out[4] == 4 && out[5] == 5);
        jh_release(out);
        k = jh_match(h, hay, (long long)n, 1, NULL, 0, &out);
        CHECK(k == 9 && out[2] == 1 && out[5] == 3 && out[8] == 4);
        jh_release(out);
        k = jh_match(h, hay, 0, 1, NULL, 0, &out); /* the empty String */
        CHECK(k == 0 && exc_is(""));
        jh_release(out);
        /* a device list goes through as it is; 0 or more than 64 ordinals are refused */
        const int32_t devs[3] = {2, 0, 1};
        int seen[64];
        k = jh_match(h, hay, (long long)n, 0, devs, 3, &out);
        CHECK(k == 6 && stub_last_devices(seen) == 3 && seen[0] == 2 && seen[1] == 0 && seen[2] == 1);
        jh_release(out);
        int32_t many[65] = {0};
        CHECK(jh_match(h, hay, (long long)n, 0, many, 65, &out) == -1 && exc_is("java/lang/IllegalArgumentException"));
        CHECK(jh_match(h, hay, (long long)n, 0, many, 0, &out) == -1 && exc_is("java/lang/IllegalArgumentException"));
        const int32_t bad[2] = {0, 9};
        CHECK(jh_match(h, hay, (long long)n, 0, bad, 2, &out) == -1 && exc_is("java/lang/IllegalStateException") && msg_is("no HIP device"));
        /* null haystack: the reference throws NullPointerException (haystack.length()) */
        CHECK(jh_match(h, NULL, -1, 0, NULL, 0, &out) == -1 && exc_is("java/lang/NullPointerException"));
        free(hay);
    }
    { /* error codes -> exception classes */
        uint64_t n;
        uint16_t *e = units_of("E3xx", &n);
        CHECK(jh_match(h, e, (long long)n, 0, NULL, 0, &out) == -1 && exc_is("java/lang/OutOfMemoryError"));
        e[1] = '7';
        CHECK(jh_match(h, e, (long long)n, 0, NULL, 0, &out) == -1 && exc_is("java/lang/UnsupportedOperationException"));
        e[1] = '5';
        CHECK(jh_match(h, e, (long long)n, 0, NULL, 0, &out) == -1 && exc_is("java/lang/IllegalStateException") && msg_is("HIP runtime error"));
        free(e);
    }
    { /* more records than the first capacity (n / 64 + 4096): ACGPU_E_OVERFLOW, one retry with the exact capacity */
        const long long n = 300000;
        uint16_t *hay = (uint16_t *)malloc((size_t)n * 2);
        for (long long i = 0; i < n; i++) hay[i] = 'x';
        const unsigned long long before = stub_overflows((const void *)(intptr_t)h);
        long long k = jh_match(h, hay, n, 1, NULL, 0, &out);
        CHECK(k == 3 * n && out[3 * (n - 1)] == (int32_t)(n - 1) && out[3 * (n - 1) + 2] == (int32_t)((n - 1) % 1000));
        CHECK(stub_overflows((const void *)(intptr_t)h) == before + 1);
        jh_release(out);
        /* the int[] cannot be allocated: OutOfMemoryError stays pending, null comes back, nothing else is called */
        jh_set_int_array_limit(1000);
        CHECK(jh_match(h, hay, n, 1, NULL, 0, &out) == -1 && exc_is("java/lang/OutOfMemoryError"));
        jh_set_int_array_limit(-1);
        free(hay);
    }
    { /* a haystack beyond one GetStringRegion slice (32 Mi chars): the slices must meet exactly */
        const long long n = 32ll * 1024 * 1024 + 12345;
        uint16_t *hay = (uint16_t *)malloc((size_t)n * 2);
        for (long long i = 0; i < n; i++) hay[i] = 'a';
        const long long at[5] = {0, 32ll * 1024 * 1024 - 1, 32ll * 1024 * 1024, 32ll * 1024 * 1024 + 1, n - 1};
        for (int i = 0; i < 5; i++) hay[at[i]] = 'x';
        long long k = jh_match(h, hay, n, 0, NULL, 0, &out);
        CHECK(k == 10);
        for (int i = 0; i < 5 && k == 10; i++) CHECK(out[2 * i] == (int32_t)at[i] && out[2 * i + 1] == (int32_t)at[i] + 1);
        jh_release(out);
        free(hay);
    }
    /* ---- records beyond what an int[] holds: refused before anything is allocated ---- */
    CHECK(jh_to_int_array(4) == 4);
    CHECK(jh_to_int_array((unsigned long long)INT_MAX - 7) == -1 && exc_is("java/lang/IllegalStateException"));
    CHECK(jh_to_int_array(3ull << 31) == -1 && exc_is("java/lang/IllegalStateException"));

    /* ---- matchBatch ---- */
    {
        uint64_t n;
        uint16_t *u = units_of("xaxxbbx", &n);
        const uint64_t o[5] = {0, 2, 2, 4, 7}; /* "xa", "", "xx", "bbx" */
        long long k = jh_match_batch(h, u, o, NULL, 4, 0, &out);
        CHECK(k == 12 && out[0] == 0 && out[1] == 0 && out[3] == 2 && out[4] == 0 && out[6] == 2 && out[7] == 1 && out[9] == 3 && out[10] == 2);
        jh_release(out);
        k = jh_match_batch(h, u, o, NULL, 4, 1, &out);
        CHECK(k == 16 && out[3] == 0 && out[15] == 2);
        jh_release(out);
        const uint8_t nn[4] = {0, 0, 1, 0};
        CHECK(jh_match_batch(h, u, o, nn, 4, 0, &out) == -1 && exc_is("java/lang/NullPointerException"));
        CHECK(jh_match_batch(h, u, o, NULL, -1, 0, &out) == -1 && exc_is("java/lang/NullPointerException"));
        k = jh_match_batch(h, u, o, NULL, 0, 0, &out);
        CHECK(k == 0 && exc_is(""));
        jh_release(out);
        free(u);
        const int nh = 3;
        const long long each = 40000; /* total / 16 + 4096 records do not hold 120000 */
        uint16_t *big = (uint16_t *)malloc((size_t)(nh * each) * 2);
        for (long long i = 0; i < nh * each; i++) big[i] = 'x';
        const uint64_t ob[4] = {0, (uint64_t)each, (uint64_t)(2 * each), (uint64_t)(3 * each)};
        k = jh_match_batch(h, big, ob, NULL, nh, 0, &out);
        CHECK(k == 3 * nh * each && out[3 * (nh * each - 1)] == nh - 1 && out[3 * (nh * each - 1) + 1] == (int32_t)(each - 1));
        jh_release(out);
        free(big);
    }
    /* ---- streams: match(Readable, ...) ---- */
    for (int pipelined = 0; pipelined < 2; pipelined++) {
        long long s = jh_stream_open(h, pipelined);
        CHECK(s != 0 && exc_is(""));
        uint64_t n;
        uint16_t *c = units_of("axbx????", &n); /* a Reader's buffer: 8 chars of which 4 count */
        long long k = jh_stream_feed(s, c, 8, 4, 0, pipelined, &out);
        CHECK(k == 2 && out[0] == 1 && out[1] == 3); /* (the Readable listener only sees the value: id = position % 1000) */
        jh_release(out);
        k = jh_stream_feed(s, c, 8, 4, 1, pipelined, &out);
        CHECK(k == 2 && out[0] == 1 && out[1] == 3);
        jh_release(out);
        CHECK(jh_stream_feed(s, c, 8, 9, 0, pipelined, &out) == -1 && exc_is("java/lang/ArrayIndexOutOfBoundsException"));
        CHECK(jh_stream_feed(s, c, 8, -1, 0, pipelined, &out) == -1 && exc_is("java/lang/ArrayIndexOutOfBoundsException"));
        CHECK(jh_stream_feed(s, c, 8, 4, 0, pipelined, &out) == -1 && exc_is("java/lang/IllegalStateException")); /* after the last feed */
        jh_stream_close(s);
        free(c);
        s = jh_stream_open(h, pipelined);
        const int len = 500000; /* more records than length / 64 + 4096: the same feed again with the exact capacity */
        uint16_t *xs = (uint16_t *)malloc((size_t)len * 2);
        for (int i = 0; i < len; i++) xs[i] = 'x';
        k = jh_stream_feed(s, xs, len, len, 1, pipelined, &out);
        CHECK(k == len && out[len - 1] == (len - 1) % 1000);
        jh_release(out);
        jh_stream_close(s);
        free(xs);
    }
    jh_free(h);
    jh_free(hw);
    jh_free(0);
    free(kw);
    free(lower);
    free(word);
    CHECK(jh_violations() == 0);
    CHECK(jh_outstanding_elements() == 0);
    if (g_failed) {
        fprintf(stderr, "%d check(s) failed\n", g_failed);
        return 1;
    }
    printf("jni glue: all scenarios ok\n");
    return 0;
}
