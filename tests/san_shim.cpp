// Test infrastructure: the host builder (ahocorasick_amd/csrc/acgpu_build.cpp) compiled WITHOUT the HIP runtime and WITH
// AddressSanitizer / UBSan (tests/test_sanitizers.py builds libacgpu_build_san.so from that file and this shim with g++).
// The shim mirrors acgpu_build / acgpu_get_info / acgpu_debug_tables of the C ABI so that the tables the sanitized builder
// makes can be compared with the product library's, array by array.
#include <cstring>
#include <new>

#include "../ahocorasick_amd/csrc/acgpu_internal.h"

using namespace acgpu;

extern "C" {

int san_build(int mode, const uint16_t *kw_units, const uint64_t *kw_off, uint32_t n_kw, int cs, const uint16_t *lower,
              const uint8_t *wordchar, void **out, int64_t *bad) {
    HostTables *t = new (std::nothrow) HostTables();
    if (!t) return ACGPU_E_NOMEM;
    int rc;
    try {
        rc = build_tables(mode, kw_units, kw_off, n_kw, cs, lower, wordchar, *t, bad);
    } catch (...) {
        rc = ACGPU_E_NOMEM;
    }
    if (rc != ACGPU_OK) {
        delete t;
        return rc;
    }
    *out = t;
    return ACGPU_OK;
}

void san_free(void *h) { delete static_cast<HostTables *>(h); }

// n_states, n_cls, dense, n_kw, min_len, max_len, first_out, fold_consistent, fold_clean, filt_k, root_b, sep_unit
void san_info(const void *h, int64_t *o) {
    const HostTables &t = *static_cast<const HostTables *>(h);
    o[0] = t.n_states; o[1] = t.n_cls; o[2] = t.dense; o[3] = t.n_kw; o[4] = t.min_len; o[5] = t.max_len; o[6] = t.first_out;
    o[7] = t.fold_consistent; o[8] = t.fold_clean; o[9] = t.filt_k; o[10] = t.root_b; o[11] = t.sep_unit;
}

int san_tables(const void *h, uint16_t *cls_lut, uint32_t *dfa, uint32_t *out_len, uint32_t *out_link, uint32_t *out_id,
               uint32_t *depth) {
    const HostTables &t = *static_cast<const HostTables *>(h);
    if (cls_lut) std::memcpy(cls_lut, t.cls_lut.data(), 65536 * sizeof(uint16_t));
    if (dfa) {
        if (!t.dense) return ACGPU_E_UNSUPPORTED;
        std::memcpy(dfa, t.dfa.data(), t.dfa.size() * sizeof(uint32_t));
    }
    if (out_len) std::memcpy(out_len, t.out_len.data(), t.n_states * sizeof(uint32_t));
    if (out_link) std::memcpy(out_link, t.out_link.data(), t.n_states * sizeof(uint32_t));
    if (out_id) std::memcpy(out_id, t.out_id.data(), t.n_states * sizeof(uint32_t));
    if (depth) std::memcpy(depth, t.depth.data(), t.n_states * sizeof(uint32_t));
    return ACGPU_OK;
}

} // extern "C"
