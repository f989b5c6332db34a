/*
 * stub_acgpu.c -- a CPU stand-in for the entry points of include/acgpu.h that the JNI glue calls (TEST INFRASTRUCTURE; the
 * CPU suite links tests/jni_min/mock_env.c against it so that the glue runs under ASan + UBSan without a GPU).  It is NOT a
 * matcher: "keywords" are ignored and a "match" is every unit equal to 'x' (U+0078), id = position modulo 1000 -- enough to
 * drive every path of the glue: capacity protocol (ACGPU_E_OVERFLOW + retry), device lists, batches, streams, error codes.
 *   mode ACGPU_MODE_WHOLEWORD: a keyword that holds '!' is refused with ACGPU_E_NONWORD and its index;
 *   a haystack that begins with "E1" .. "E7" makes acgpu_match_u16 return the error code -1 .. -7.
 */
#include <stdlib.h>
#include <string.h>

#include "acgpu.h"

struct acgpu_automaton {
    int mode, cs, have_lower, have_word;
    uint32_t n_kw;
    uint64_t n_units;
    unsigned long long calls, overflows;
};
struct acgpu_stream {
    const acgpu_automaton *a;
    uint64_t pos;
    int pipelined, finished;
};

static int g_last_devices[64], g_last_n_devices;

const char *acgpu_strerror(int code) {
    switch (code) {
    case ACGPU_OK: return "ok";
    case ACGPU_E_INVALID: return "invalid argument";
    case ACGPU_E_NONWORD: return "keyword contains non-word characters";
    case ACGPU_E_NOMEM: return "out of memory";
    case ACGPU_E_OVERFLOW: return "output capacity too small";
    case ACGPU_E_HIP: return "HIP runtime error";
    case ACGPU_E_NODEVICE: return "no HIP device";
    case ACGPU_E_UNSUPPORTED: return "unsupported";
    default: return "unknown error";
    }
}

int acgpu_build(int mode, const uint16_t *kw_units, const uint64_t *kw_off, uint32_t n_kw, int case_sensitive, const uint16_t *lower_tbl,
                const uint8_t *wordchar_tbl, acgpu_automaton **out, int64_t *bad_keyword) {
    if (!out) return ACGPU_E_INVALID;
    *out = NULL;
    if (mode < 0 || mode > 4) return ACGPU_E_INVALID;
    if (!case_sensitive && !lower_tbl) return ACGPU_E_INVALID;
    if ((mode == ACGPU_MODE_WHOLEWORD || mode == ACGPU_MODE_WWLONGEST) && !wordchar_tbl) return ACGPU_E_INVALID;
    uint64_t total = 0;
    for (uint32_t k = 0; k < n_kw; k++) {
        for (uint64_t i = kw_off[k]; i < kw_off[k + 1]; i++) {
            total += kw_units[i]; /* (every unit is read: the sanitizers see a short buffer) */
            if (mode == ACGPU_MODE_WHOLEWORD && kw_units[i] == '!') {
                if (bad_keyword) *bad_keyword = (int64_t)k;
                return ACGPU_E_NONWORD;
            }
        }
    }
    if (lower_tbl)
        for (int i = 0; i < 65536; i++) total += lower_tbl[i];
    if (wordchar_tbl)
        for (int i = 0; i < 65536; i++) total += wordchar_tbl[i];
    acgpu_automaton *a = (acgpu_automaton *)calloc(1, sizeof(*a));
    if (!a) return ACGPU_E_NOMEM;
    a->mode = mode;
    a->cs = case_sensitive;
    a->have_lower = lower_tbl != NULL;
    a->have_word = wordchar_tbl != NULL;
    a->n_kw = n_kw;
    a->n_units = total;
    *out = a;
    return ACGPU_OK;
}

void acgpu_free(acgpu_automaton *a) { free(a); }

static int scan(const uint16_t *h, uint64_t n, int record_kind, int32_t *out, uint64_t cap, uint64_t *n_out, int32_t tag, int64_t shift) {
    const int w = record_kind / 4 + (tag >= 0 ? 1 : 0);
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (h[i] != 'x') continue;
        if (k < cap) {
            int32_t *o = out + k * (uint64_t)w;
            if (tag >= 0) *o++ = tag;
            o[0] = (int32_t)((int64_t)i + shift);
            o[1] = (int32_t)((int64_t)i + shift + 1);
            if (record_kind == ACGPU_REC_MAP) o[2] = (int32_t)(i % 1000);
        }
        k++;
    }
    *n_out = k;
    return k > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

int acgpu_match_u16(const acgpu_automaton *a, const uint16_t *haystack, uint64_t n_units, int record_kind, void *out, uint64_t cap,
                    uint64_t *n_out) {
    if (!a || !n_out || (n_units && !haystack) || (cap && !out)) return ACGPU_E_INVALID;
    ((acgpu_automaton *)a)->calls++;
    if (n_units >= 2 && haystack[0] == 'E' && haystack[1] >= '1' && haystack[1] <= '7') return -(int)(haystack[1] - '0');
    const int rc = scan(haystack, n_units, record_kind, (int32_t *)out, cap, n_out, -1, 0);
    if (rc == ACGPU_E_OVERFLOW) ((acgpu_automaton *)a)->overflows++;
    return rc;
}

int acgpu_match_u16_multi(const acgpu_automaton *a, const uint16_t *haystack, uint64_t n_units, const int *devices, int n_devices,
                          int record_kind, void *out, uint64_t cap, uint64_t *n_out) {
    if (!devices || n_devices < 1 || n_devices > 64) return ACGPU_E_INVALID;
    g_last_n_devices = n_devices;
    memcpy(g_last_devices, devices, (size_t)n_devices * sizeof(int));
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] > 7) return ACGPU_E_NODEVICE;
    return acgpu_match_u16(a, haystack, n_units, record_kind, out, cap, n_out);
}

int acgpu_match_batch_u16(const acgpu_automaton *a, const uint16_t *units, const uint64_t *offsets, uint32_t n_haystacks, int record_kind,
                          void *out, uint64_t cap, uint64_t *n_out) {
    if (!a || !n_out || !offsets) return ACGPU_E_INVALID;
    const int w = record_kind / 4 + 1;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_haystacks; i++) {
        uint64_t k = 0;
        const uint64_t room = total < cap ? cap - total : 0;
        scan(units + offsets[i], offsets[i + 1] - offsets[i], record_kind, (int32_t *)out + (total < cap ? total : cap) * (uint64_t)w, room, &k,
             (int32_t)i, 0);
        total += k;
    }
    *n_out = total;
    return total > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

int acgpu_stream_open(const acgpu_automaton *a, acgpu_stream **out) {
    if (!a || !out) return ACGPU_E_INVALID;
    *out = (acgpu_stream *)calloc(1, sizeof(acgpu_stream));
    if (!*out) return ACGPU_E_NOMEM;
    (*out)->a = a;
    return ACGPU_OK;
}
int acgpu_stream_set_pipelined(acgpu_stream *s, int on) {
    if (!s || s->pos) return ACGPU_E_INVALID;
    s->pipelined = on;
    return ACGPU_OK;
}
int acgpu_stream_reserve(acgpu_stream *s, uint64_t n_units, uint16_t **buf) {
    (void)s;
    (void)n_units;
    (void)buf;
    return ACGPU_E_UNSUPPORTED; /* (the glue does not call it) */
}
/* records relative to *base = the position of the feed's first unit; ACGPU_E_OVERFLOW consumes nothing */
int acgpu_stream_feed(acgpu_stream *s, const uint16_t *units, uint64_t n_units, int final, int record_kind, void *out, uint64_t cap,
                      uint64_t *n_out, int64_t *base) {
    if (!s || !n_out || !base || s->finished || (n_units && !units)) return ACGPU_E_INVALID;
    *base = 0;
    const int rc = scan(units, n_units, record_kind, (int32_t *)out, cap, n_out, -1, (int64_t)s->pos);
    if (rc != ACGPU_OK) return rc;
    s->pos += n_units;
    s->finished = final;
    return ACGPU_OK;
}
void acgpu_stream_close(acgpu_stream *s) { free(s); }

/* what the stub saw (the tests look at it) */
__attribute__((visibility("default"))) int stub_last_devices(int *out) {
    memcpy(out, g_last_devices, sizeof(int) * (size_t)g_last_n_devices);
    return g_last_n_devices;
}
__attribute__((visibility("default"))) unsigned long long stub_overflows(const acgpu_automaton *a) { return a->overflows; }
