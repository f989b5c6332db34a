// acgpu_api.hip -- the C ABI of include/acgpu.h: automaton lifetime, device residency of the tables,
// per-device scratch pool, and the match pipelines (scan -> prefix sum of chunk counts -> permutation).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <cctype>
#include <pthread.h>
#include <sched.h>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "acgpu_host.h"
#include "acgpu_internal.h"
#include "acgpu_kernels.h"
#include "acgpu_small.h"

using namespace acgpu;

namespace acgpu {
thread_local int g_last_hip_error = 0;
}

namespace {

template <typename T>
int upload(DeviceState &d, const std::vector<T> &v, const T **out) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_TRY(hipMalloc(&p, bytes));
    d.table_allocs.push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(p);
    return ACGPU_OK;
}

} // namespace

namespace {

uint32_t lds_states_for(const HostTables &t) {
    if (!t.dense) return 0;
    int64_t budget = tunables().lds_table_bytes;
    const int64_t max_budget = 160 * 1024 - (int64_t)scan_queue_bytes(scan_block_threads()) - 1024;
    budget = std::max<int64_t>(0, std::min(budget, max_budget));
    // table classes: k_ac_dfa keeps the class pages behind the rows when they are small next to them (at most a quarter of the
    // budget: 3000 CJK units are 3.5 KB of pages)
    if (!t.range_cls && !t.dfa_pages.empty() && (int64_t)t.dfa_pages.size() * 2 + 16 <= budget / 4)
        budget -= (int64_t)t.dfa_pages.size() * 2 + 16;
    uint64_t row = (uint64_t)t.n_cls * t.entry_bytes;
    uint64_t s = row ? (uint64_t)budget / row : 0;
    return (uint32_t)std::min<uint64_t>(s, t.n_states);
}

// caller holds a->mu
int ensure_device(acgpu_automaton *a, DeviceState **out, int lane) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    auto it = a->dev.find({dev, lane});
    if (it != a->dev.end()) {
        *out = it->second.get();
        return ACGPU_OK;
    }
    std::unique_ptr<DeviceState> d(new (std::nothrow) DeviceState());
    if (!d) return ACGPU_E_NOMEM;
    d->device = dev;
    d->lane = lane;
    if (lane > 0) HIP_TRY(hipStreamCreateWithFlags(&d->call_stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    d->n_cu = d->n_cu_phys = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const HostTables &t = a->t;
    DevTables &T = d->T;
    int rc;
    if ((rc = upload(*d, t.cls_lut, &T.cls_lut))) return rc;
    if ((rc = upload(*d, t.lower, &T.lower))) return rc;
    if ((rc = upload(*d, t.wflags, &T.wflags))) return rc;
    if ((rc = upload(*d, t.wbits, &T.wbits))) return rc;
    if (!t.fold_consistent) {
        if ((rc = upload(*d, t.wflags_f, &d->wflags_f))) return rc;
        if ((rc = upload(*d, t.wbits_f, &d->wbits_f))) return rc;
    }
    if ((rc = upload(*d, t.out_len, &T.out_len))) return rc;
    if ((rc = upload(*d, t.out_link, &T.out_link))) return rc;
    if ((rc = upload(*d, t.out_id, &T.out_id))) return rc;
    if ((rc = upload(*d, t.fail, &T.fail))) return rc;
    if ((rc = upload(*d, t.depth, &T.depth))) return rc;
    if ((rc = upload(*d, t.term_id, &T.term_id))) return rc;
    if ((rc = upload(*d, t.hkeys, &T.hkeys))) return rc;
    if ((rc = upload(*d, t.hvals, &T.hvals))) return rc;
    T.dfa = nullptr;
    if (t.dense) {
        if (t.entry_bytes == 2) {
            std::vector<uint16_t> narrow(t.dfa.size());
            for (size_t i = 0; i < t.dfa.size(); i++) narrow[i] = (uint16_t)t.dfa[i];
            const uint16_t *p;
            if ((rc = upload(*d, narrow, &p))) return rc;
            T.dfa = p;
        } else {
            const uint32_t *p;
            if ((rc = upload(*d, t.dfa, &p))) return rc;
            T.dfa = p;
        }
    }
    T.root_tab = nullptr; T.root_b = t.root_b; T.root_rk = t.root_rk;
    if (t.root_b && (rc = upload(*d, t.root_tab, &T.root_tab))) return rc;
    T.bits_tab = nullptr; T.bits_rk = t.bits_rk;
    if (t.bits_rk && (rc = upload(*d, t.bits_tab, &T.bits_tab))) return rc;
    T.bits_idkeys = nullptr; T.bits_idmask = t.bits_idmask;
    if (t.bits_rk && !t.bits_idkeys.empty() && (rc = upload(*d, t.bits_idkeys, &T.bits_idkeys))) return rc;
    T.hy_dense = T.hy_nodes = T.hy_mask = T.hy_out = T.hy_ids = nullptr;
    T.hy_n_dense = t.hy_n_dense; T.hy_n_states = t.hy_n_states;
    if (t.hy_n_states) {
        { // one allocation, rows first (padded to 16 bytes), nodes behind them: k_ac_states reads either with ONE 16-byte gather
            std::vector<uint32_t> all(t.hy_dense);
            all.resize((all.size() + 3) & ~(size_t)3, 0u);
            const size_t node_at = all.size();
            all.insert(all.end(), t.hy_nodes.begin(), t.hy_nodes.end());
            all.resize(all.size() + 4, 0u);
            if ((rc = upload(*d, all, &T.hy_dense))) return rc;
            T.hy_nodes = T.hy_dense + node_at;
        }
        if ((rc = upload(*d, t.hy_mask, &T.hy_mask))) return rc;
        if ((rc = upload(*d, t.hy_out, &T.hy_out))) return rc;
        if ((rc = upload(*d, t.hy_ids, &T.hy_ids))) return rc;
    }
    if ((rc = upload(*d, t.filt_bits, &T.filt_bits))) return rc;
    if ((rc = upload(*d, t.kgram_node, &T.kgram_node))) return rc;
    T.fold_range = t.fold_range; T.fr_base = t.fr_base; T.fr_span = t.fr_span; T.fr_base2 = t.fr_base2; T.fr_himask = t.fr_himask;
    T.fr_base3 = t.fr_base3; T.fr_base4 = t.fr_base4; T.fr_nr = t.fr_nr;
    T.l2_bloom = nullptr; T.l2_big = nullptr; T.l2_depth = 0;
    if (t.l2_depth) {
        if ((rc = upload(*d, t.l2_bloom, &T.l2_bloom))) return rc;
        if (!t.l2_big.empty() && (rc = upload(*d, t.l2_big, &T.l2_big))) return rc;
        T.l2_depth = t.l2_depth;
    }
    if ((rc = upload(*d, t.rterm, &T.rterm))) return rc;
    if ((rc = upload(*d, t.rtab, &T.rtab))) return rc;
    T.kshort = nullptr; T.ks_keys = nullptr; T.ks_vals = nullptr; T.ks_mask = t.ks_mask; T.has_short = t.has_short ? 1u : 0u;
    if (t.has_short && !t.kshort.empty() && (rc = upload(*d, t.kshort, &T.kshort))) return rc;
    if (t.has_short && !t.ks_keys.empty()) {
        if ((rc = upload(*d, t.ks_keys, &T.ks_keys))) return rc;
        if ((rc = upload(*d, t.ks_vals, &T.ks_vals))) return rc;
    }
    T.rdense = t.rdense;
    if ((rc = upload(*d, t.rhkeys, &T.rhkeys))) return rc;
    if ((rc = upload(*d, t.rhvals, &T.rhvals))) return rc;
    if ((rc = upload(*d, t.tile_lut, &T.tile_lut))) return rc;
    T.cls_pages = nullptr; T.cls_pages_bytes = (uint32_t)t.cls_pages.size();
    if (!t.cls_pages.empty() && (rc = upload(*d, t.cls_pages, &T.cls_pages))) return rc;
    T.dfa_pages = nullptr; T.dfa_pages_bytes = (uint32_t)t.dfa_pages.size() * 2;
    if (!t.dfa_pages.empty() && (rc = upload(*d, t.dfa_pages, &T.dfa_pages))) return rc;
    if ((rc = upload(*d, t.kg_keys, &T.kg_keys))) return rc;
    if ((rc = upload(*d, t.kg_vals, &T.kg_vals))) return rc;
    T.kg_mask = t.kg_mask; T.hashk = t.hashk;
    if ((rc = upload(*d, t.ww_fat, &T.ww_fat))) return rc;
    if ((rc = upload(*d, t.ww_recs, &T.ww_recs))) return rc;
    T.ww_bp_idx = nullptr; T.ww_bp_pages = nullptr; T.ww_bp_delta = nullptr; T.ww_bp_n = t.ww_bp_n; T.ww_bp_wbits = T.wbits;
    if (t.ww_bp_n && ((rc = upload(*d, t.ww_bp_idx, &T.ww_bp_idx)) || (rc = upload(*d, t.ww_bp_pages, &T.ww_bp_pages)) ||
                      (rc = upload(*d, t.ww_bp_delta, &T.ww_bp_delta)))) return rc;
    T.ww_ph = nullptr; T.ww_ph_disp = nullptr; T.ww_ph_n = t.ww_ph_n; T.ww_ph_buckets = t.ww_ph_buckets;
    if (!t.ww_ph.empty() && ((rc = upload(*d, t.ww_ph, &T.ww_ph)) || (rc = upload(*d, t.ww_ph_disp, &T.ww_ph_disp)))) return rc;
    if ((rc = upload(*d, t.fold_pgidx, &T.fold_pgidx))) return rc;
    if ((rc = upload(*d, t.fold_pages, &T.fold_pages))) return rc;
    if ((rc = upload(*d, t.ww_bloom, &T.ww_bloom))) return rc;
    T.ww_fat_mask = t.ww_fat_mask; T.ww_seed = t.ww_seed; T.fold_n_pages = t.fold_n_pages; T.fold_direct_n = t.fold_direct_n; T.ww_bloom_mask = t.ww_bloom_mask;
    T.rhmask = t.rhmask; T.filt_k = t.filt_k; T.filt_n = t.filt_n; T.filt_other = t.filt_other;
    T.filt_words = (uint32_t)t.filt_bits.size(); T.filt_row_bytes = t.filt_row_bytes;
    T.hmask = t.hmask;
    T.n_states = t.n_states; T.n_cls = t.n_cls; T.first_out = t.first_out; T.max_len = t.max_len; T.min_len = t.min_len;
    T.cls_base = t.cls_base; T.cls_span = t.cls_span; T.range_cls = t.range_cls; T.cs = t.cs; T.dense = t.dense;
    T.entry_bytes = (int32_t)t.entry_bytes;
    T.lds_entries = lds_states_for(t) * t.n_cls;
    HIP_TRY(hipHostMalloc((void **)&d->h_counter, 64, hipHostMallocDefault));
    for (auto &e : d->ev) HIP_TRY(hipEventCreate(&e));
    for (auto &tk : d->tickets) {
        for (auto &e : tk.ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreateWithFlags(&tk.done, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void **)&tk.h_count, 64, hipHostMallocDefault));
        tk.owner = d.get();
    }
    *out = d.get();
    a->dev[{dev, lane}] = std::move(d);
    return ACGPU_OK;
}

uint32_t round_up8(uint64_t v) { return (uint32_t)((v + 7) & ~7ull); }

// Region size of the tile kernels for a long shard: every wave scans r regions of R units one after the other, so the scan
// lasts as long as r * R units of one wave -- with a fixed R the step from "r regions fill the waves exactly" to one region
// more costs a whole region per wave (2^29 units in 32768-unit regions on 4096 waves: r = 4; one unit more, or four CUs
// fewer (tunable reserve_cus): r = 5, a quarter slower).  Chosen here: whole tile groups, between r_lo and r_hi regions per
// wave, the smallest r * R that covers the shard; among equals the R nearest `prefer`.
uint64_t balanced_region_units(uint64_t len, uint64_t waves, uint64_t g, uint64_t r_min, uint64_t r_hi, uint64_t prefer) {
    uint64_t best_R = 0, best_cost = ~0ull, best_dist = ~0ull;
    for (uint64_t r = 1; r <= r_hi; ++r) {
        const uint64_t m = (len + waves * r * g - 1) / (waves * r * g);
        const uint64_t R = std::max<uint64_t>(m, 1) * g;
        if (R < r_min || R > 65536) continue;
        const uint64_t cost = r * R, dist = R > prefer ? R - prefer : prefer - R;
        if (cost < best_cost || (cost == best_cost && dist < best_dist)) {
            best_cost = cost;
            best_dist = dist;
            best_R = R;
        }
    }
    return best_R;
}

// Which ALL-mode kernel serves this dictionary: the position-parallel K-gram tile kernel when the suffix filter
// exists and is selective, otherwise the general DFA chunk scan (any alphabet, any keyword lengths).
// force_kernel: 0 = automatic, 1 = DFA chunk scan, 2 = fused tile kernel, 3 = split tile kernels (filter + verification)
// The tile kernel serves every dictionary that has a suffix filter: measured on 0.5 GiB it beats the DFA chunk scan
// 3x even when the filter passes every position (10 k keywords: 0.22 ms; 100 k keywords, density 0.20: 0.72 against
// 1.78 ms; 300 k, density 0.48: 1.6 against 4.4 ms; 200 two-to-four-unit keywords over {a,b,c,d}, density 1.0: 25
// against 76 ms, both bound by emitting 647 M records).
bool use_tile_kernel(const HostTables &t) {
    if (t.filt_k == 0) return false;
    const int64_t f = tunables().force_kernel;
    if (f == 1) return false;
    if (f == 2 || f == 3) return true;
    // The packed forms (range classes, folded or merged ranges) always win: with K = 4 even where the filter passes everything --
    // the 235 886-word list of the reference's README (52 letters in two ranges, the single letters among the keywords: every
    // position ends a keyword) 23.5 against 62.9 ms per 2^28 units, both bound by 412 M records (tools/readme_shapes.py).
    if (t.range_cls || t.fold_range) return true;
    // The class-table forms (bucketed classes: more than 63 distinct units; or up to 63 classes that no range arithmetic
    // gives) run the scalar filter -- three LDS reads per unit and, bucketed, a hash probe of its K units per candidate.
    // Against them (tools/wide_alphabets.py, 2^28 units, round 4): a DFA table that stays in the L2 cache makes k_ac_dfa the
    // faster kernel as soon as the filter passes more than a tenth of the positions (300 CJK units, 2 k keywords of 2-4
    // units, a 3.6 MB table, density 0.17: 0.65 against 1.06 ms; config 2's phrases case-insensitive, density 0.24: 0.60
    // against 1.04 ms; density 0.08: 0.39 ms for the tile kernel), while a table far beyond the cache (3000 CJK units) loses
    // to the tile kernel whatever the filter passes (20 k keywords of 2-8 units, density 0.54: 2.97 against 5.19 ms; of 1-4
    // units, 1.03: 10.6 against 25.8 ms, 218 M records; 20 k of 2 units, K = 2: 2.63 against 3.92 ms; only 100 k keywords of
    // 2-3 units went the other way, 6.8 against 5.35 ms), as does the sparse form (4.7 ms and more).
    const uint64_t table_bytes = t.dense ? (uint64_t)t.n_states * t.n_cls * (uint64_t)t.entry_bytes : ~0ull;
    if (table_bytes <= (6ull << 20)) return t.filt_density <= 0.1;
    return true;
}

// LONGEST takes the all-matches pipeline only when matches are expected to be sparse
bool filter_is_selective(const HostTables &t) { return t.filt_k != 0 && t.filt_density <= 0.08; }

// the split form needs the filter rows to fit the smaller static LDS array of the filter-only kernel
// (measured at config 2: filter 0.27 ms + verification 0.33 ms against 0.39 ms fused -- the fused kernel verifies a
// candidate while its text is still in the L2 of the XCD that streamed it; the split form gathers it from HBM again --
// so the split form is only taken on request)
bool use_split_form(const DevTables &T) { return tunables().force_kernel == 3 && tile_split_supported(T); }

// The device tables as a scan that folds in EVERY lookup sees them (word-character tables that are not fold-consistent):
// w'[c] = word[lower[c]] in place of both word-character tables.
DevTables folded_tables(const DeviceState &d) {
    DevTables T = d.T;
    if (d.wflags_f) {
        T.wflags = d.wflags_f;
        T.wbits = d.wbits_f;
    }
    return T;
}

// ALL-mode pipeline on one shard, the form for texts with dense matches: see match_all.
constexpr double kStatesFormDensity = 0.05; // records per unit of the pool's last call from which k_ac_states is taken
int match_all_states(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap, uint64_t *n_out,
                     hipStream_t stream, acgpu_profile *prof, Ticket *tk, uint32_t hot_rows) {
    const HostTables &t = a->t;
    hipEvent_t *ev = tk ? tk->ev : d.ev;
    const bool timed = tk ? tk->profiled : prof != nullptr;
    const uint64_t own_len = sh->own_end - sh->own_begin;
    int rc;
    AcStatesLaunch S{};
    S.d_hay = sh->d_hay;
    S.n_units = (uint32_t)sh->n_units;
    S.own_begin = (uint32_t)sh->own_begin;
    S.own_end = (uint32_t)sh->own_end;
    S.g0 = S.own_begin & ~3u;
    S.halo = t.max_len - 1;
    S.hot_rows = hot_rows;
    const uint64_t span = sh->own_end - S.g0;
    // a lane's chunk: 1024 units, shorter (down to 256) when the text would leave lanes of the chip without one
    S.chunk_log2 = 10;
    while (S.chunk_log2 > 8 && (span >> S.chunk_log2) < (uint64_t)ac_states_lanes_per_cu() * d.n_cu) --S.chunk_log2;
    const uint64_t chunks = (span + (1ull << S.chunk_log2) - 1) >> S.chunk_log2;
    S.n_waves = (uint32_t)((chunks + 63) / 64);
    S.n_chunks = (uint32_t)chunks;
    if ((rc = d.counter.ensure(64))) return rc;
    if (tunables().tile_debug & (1ll << 40)) return ACGPU_E_NOMEM; // (tests: the allocation "fails", the caller falls back)
    if ((rc = d.statebuf.ensure((((size_t)S.n_waves * 64) << S.chunk_log2) * 4 + 64))) return rc;
    if ((rc = d.chunk_counts.ensure((size_t)S.n_chunks * 4))) return rc;
    if ((rc = d.offsets.ensure((size_t)S.n_chunks * 8))) return rc;
    if ((rc = d.scan_tmp.ensure(((size_t)S.n_chunks / 2048 + 2) * 8))) return rc;
    S.d_state = (uint32_t *)d.statebuf.p;
    S.d_counts = (uint32_t *)d.chunk_counts.p;
    S.d_offsets = (const uint64_t *)d.offsets.p;
    S.d_out = d_out;
    S.cap = cap;
    S.grid = (int)std::min<uint64_t>((uint64_t)d.n_cu * (ac_states_lanes_per_cu() / 1024u), (S.n_waves + 15) / 16);
    HIP_TRY(hipMemsetAsync(d.counter.p, 0, 64, stream)); // (word 1: the "redo" flag of the result -- never raised here)
    d.cclean[0] = false; // (match_all's first set of slot counters lives here)
    if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
    HIP_TRY(launch_ac_states(d.T, S, t.range_cls, stream));
    if (timed) HIP_TRY(hipEventRecord(ev[1], stream));
    HIP_TRY(launch_exclusive_scan(S.d_counts, S.n_chunks, (uint64_t *)d.offsets.p, (uint64_t *)d.scan_tmp.p, stream));
    HIP_TRY(launch_ac_states_out(d.T, S, record_kind == ACGPU_REC_MAP, stream));
    if (timed) HIP_TRY(hipEventRecord(ev[2], stream));
    unsigned long long *h_slot = tk ? tk->h_count : d.h_counter, *d_slot = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&d_slot, h_slot, 0));
    HIP_TRY(launch_publish_result((const unsigned long long *)d.scan_tmp.p + scan_tiles_for(S.n_chunks), (const unsigned long long *)d.counter.p, d_slot,
                                  reinterpret_cast<acgpu_device_result *>(sh->d_result), stream));
    if (tk) {
        tk->shard = *sh;
        tk->record_kind = record_kind;
        tk->d_out = d_out;
        tk->stream = stream;
        tk->done_is_ev2 = false;
        HIP_TRY(hipEventRecord(tk->done, stream));
        tk->scanned = own_len;
        std::snprintf(tk->kname, sizeof(tk->kname), "k_ac_states");
        return ACGPU_OK;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = *d.h_counter;
    d.all_density = (double)*n_out / (double)own_len;
    if (prof) {
        HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
        HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, d.ev[1], d.ev[2]));
        HIP_TRY(hipEventElapsedTime(&prof->total_ms, d.ev[0], d.ev[2]));
        prof->scan_units = own_len;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "k_ac_states");
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// ALL-mode pipeline on one shard.
// With a ticket the call returns after enqueueing (no host synchronisation); acgpu_match_device_end collects it.
int match_all(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
              uint64_t *n_out, hipStream_t stream, acgpu_profile *prof, Ticket *tk = nullptr, bool fused_only = false,
              const DevTables *Tov = nullptr) {
    // (Tov: WHOLEWORD -- the tables of a folding scan, see folded_tables)
    // (fused_only: the redo after an overflow of the split form's candidate slices or of a scratch slice -- the fused
    // kernel, one scratch slice)
    const HostTables &t = a->t;
    hipEvent_t *ev = tk ? tk->ev : d.ev;
    const bool timed = tk ? tk->profiled : prof != nullptr;
    const uint64_t own_len = sh->own_end - sh->own_begin;
    // WHOLEWORD (fold-consistent tables) is the same pipeline around another scan kernel: run starts instead of K-gram
    // candidates, ranks by match start, halos of 1 unit on the left and max_len + 1 on the right
    const bool ww = t.mode == ACGPU_MODE_WHOLEWORD;
    const uint32_t halo = ww ? 1u : (t.max_len > 0 ? t.max_len - 1 : 0);
    if (!sh->text_begin && sh->own_begin < halo) return ACGPU_E_INVALID; // left halo too short
    if (ww && !sh->text_end && sh->n_units - sh->own_end < (uint64_t)t.max_len + 1) return ACGPU_E_INVALID; // right halo
    if (prof) {
        std::memset(prof, 0, sizeof(*prof));
    }
    if (own_len == 0 || t.n_states <= 1) {
        if (sh->d_result) HIP_TRY(hipMemsetAsync(sh->d_result, 0, sizeof(acgpu_device_result), stream));
        if (tk) {
            tk->h_count[0] = tk->h_count[1] = 0; // (no kernel will write the slot)
            tk->profiled = false;
            tk->stream = stream;
            HIP_TRY(hipEventRecord(tk->done, stream));
            return ACGPU_OK;
        }
        *n_out = 0;
        return ACGPU_OK;
    }
    // Texts in which this dictionary matches densely (natural words in natural text: every filter passes, every verification walk
    // is long): the automaton's state behind every unit (k_ac_states over the compact automaton of acgpu_build.cpp 6d), then the
    // records from the states (acgpu_states.hip).  Its cost does not depend on the text (~ one gather per unit), the tile kernel's
    // does: what this pool's last call found decides (records per unit; a pool's first call looks at the beginning of a long text,
    // and takes the tile kernel for a short one or when it may not wait).
    // Tunable all_form, bits: 1 = never, 2 = whatever the last call found, 4 = also for short texts.
    {
        const int64_t aform = tunables().all_form;
        const size_t st_pages = (!t.range_cls && !t.dfa_pages.empty()) ? t.dfa_pages.size() * 2 : 0;
        const uint32_t st_hot = (!ww && !Tov && t.hy_n_states && (t.range_cls || st_pages > 0))
                                    ? ac_states_hot_rows(t.n_cls, t.hy_n_dense, (uint32_t)st_pages) : 0;
        const bool usable = st_hot > 0 && !(aform & 1) && tunables().force_kernel == 0 && !fused_only && (own_len >= (1ull << 20) || (aform & 4));
        // a pool that knows nothing yet and a long text (a call that may wait): the first 2^20 units of the shard are counted
        // first (this form, no records written: 60 us) -- the whole text then takes the form its beginning suggests
        if (usable && !tk && d.all_density < 0.0 && !(aform & 2) && own_len >= (1ull << 23)) {
            acgpu_shard head = *sh;
            head.own_end = head.own_begin + (1ull << 20);
            head.d_result = nullptr;
            uint64_t n_head = 0;
            const int prc = match_all_states(a, d, &head, record_kind, d_out, 0, &n_head, stream, nullptr, nullptr, st_hot);
            // (no room for the probe's state words: like the call itself below, the tile kernel it is -- the pool stays without a
            // density, so a later call asks again)
            if (prc != ACGPU_OK && prc != ACGPU_E_OVERFLOW && prc != ACGPU_E_NOMEM) return prc;
        }
        if (usable && ((aform & 2) || d.all_density >= kStatesFormDensity)) {
            const int src = match_all_states(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, tk, st_hot);
            if (src != ACGPU_E_NOMEM) return src; // (no room for 4 bytes of state per unit -- before anything was launched: the tile kernel it is)
        }
    }
    int rc;
    const size_t counter_bytes = (size_t)kMaxSlices * kCounterStride * 8; // one set; layout: [set 0][set 1][overflow word]
    if ((rc = d.counter.ensure(2 * counter_bytes + 64))) return rc;
    if (d.counter.p != d.counter_seen) {
        d.counter_seen = d.counter.p;
        d.cclean[0] = d.cclean[1] = false;
        HIP_TRY(hipMemsetAsync((char *)d.counter.p + 2 * counter_bytes, 0, 64, stream));
    }
    const int cs = d.cset;
    unsigned long long *counters = (unsigned long long *)((char *)d.counter.p + (size_t)cs * counter_bytes);
    unsigned long long *counters_next = (unsigned long long *)((char *)d.counter.p + (size_t)(1 - cs) * counter_bytes);
    uint32_t *overflow_word = (uint32_t *)((char *)d.counter.p + 2 * counter_bytes);
    // the tile kernel reserves scratch slots 256 at a time per wave: head-room for the unused tails; a quarter more than
    // the caller's capacity so that the scratch slices (one per workgroup) tolerate unevenly spread matches
    uint64_t scratch_cap = std::min<uint64_t>(
        std::max<uint64_t>(cap, 1) + cap / 4 +
            (uint64_t)d.n_cu * (ww ? ww_blocks_per_cu() : 1) * (tile_block_threads() / 64) * tile_reserve_slots(),
        0xffffffe0ull);
    if ((rc = d.scratch.ensure(scratch_cap * sizeof(ScratchRec)))) return rc;
    if (!d.cclean[cs]) HIP_TRY(hipMemsetAsync(counters, 0, counter_bytes, stream)); // (normally zeroed by the previous call's permute pass)
    // from here on this set is in use; the other set only counts as clean once the permute pass that zeroes it has been
    // launched (below) -- an early error return leaves both marked dirty and the next call clears its set itself
    d.cclean[0] = d.cclean[1] = false;
    uint32_t n_slices = 1;
    uint64_t slice_slots = scratch_cap;
    const char *kname = "";
    uint64_t scanned = 0;
    uint32_t n_chunks = 0, chunk_units = 0, perm_base = (uint32_t)sh->own_begin;
    const uint32_t *id_map = nullptr;
    bool split = false, fused_finalize = false, ww_direct = false, ext_timed = false, fused_tail = false;
    uint32_t regions_per_wg = 0, ww_region_cap = 0;
    int by_start = 0;
    if (ww) {
        TileLaunch L{};
        L.block = tile_block_threads();
        const int waves_per_block = L.block / 64;
        // regions as large as still gives every wave one (fewer forced drains: 65536 against 16384 units -2 % at config 5's share)
        const uint64_t ww_waves = (uint64_t)d.n_cu * ww_blocks_per_cu() * waves_per_block;
        uint64_t R = tunables().region_units > 0 ? (uint64_t)tunables().region_units
                     : own_len >= 65536 * ww_waves ? 65536 : own_len >= 32768 * ww_waves ? 32768 : 16384;
        if (tunables().region_units <= 0 && own_len >= 32768 * ww_waves) { // long shards: regions that fill the waves evenly
            const uint64_t Rb = balanced_region_units(sh->own_end - (sh->own_begin & ~7ull), ww_waves, tile_group_units(), 16384, 16, 65536);
            if (Rb) R = Rb;
        }
        { const uint64_t g = tile_group_units(); R = std::max<uint64_t>(g, (R + g - 1) / g * g); }
        L.region_units = (uint32_t)R;
        const uint64_t base8 = sh->own_begin & ~7ull;
        L.n_regions = (uint32_t)((sh->own_end - base8 + R - 1) / R);
        L.regions_per_wave = (uint32_t)((L.n_regions + ww_waves - 1) / ww_waves);
        const uint64_t waves_used = ((uint64_t)L.n_regions + L.regions_per_wave - 1) / L.regions_per_wave;
        L.grid = (int)((waves_used + waves_per_block - 1) / waves_per_block);
        perm_base = (uint32_t)base8;
        by_start = 1;
        L.d_hay = sh->d_hay;
        L.n_units = (uint32_t)sh->n_units;
        L.own_begin = (uint32_t)sh->own_begin;
        L.own_end = (uint32_t)sh->own_end;
        L.cap = scratch_cap;
        L.lds_bytes = ww_lds_bytes(L.block, Tov ? *Tov : d.T);
        L.debug = (uint32_t)tunables().tile_debug | (tunables().force_kernel == 1 ? 256u : 0u); // 256: trie-walk verification
        L.d_overflow = overflow_word;
        // one scratch slice and slot counter per workgroup (config 5 emits 15 M records per shard: 60 k reservations that one
        // counter would serve at under 100 per microsecond); a slice that fills up -> redo with one slice
        if (!fused_only && L.grid > 1 && !(L.debug & 16384u)) {
            n_slices = (uint32_t)std::min<int>(L.grid, kMaxSlices);
            slice_slots = scratch_cap / n_slices;
        }
        L.n_slices = n_slices;
        L.slice_slots = (uint32_t)slice_slots;
        if ((rc = d.chunk_counts.ensure((size_t)L.n_regions * 4))) return rc;
        if ((rc = d.offsets.ensure((size_t)L.n_regions * 8))) return rc;
        if ((rc = d.scan_tmp.ensure(((size_t)L.n_regions / 2048 + 2) * 8))) return rc;
        L.d_scratch = (ScratchRec *)d.scratch.p;
        L.d_counter = counters;
        L.d_region_counts = (uint32_t *)d.chunk_counts.p;
        // region-local record slots (a region of R units holds at most R/2 + 1 words): no slot reservations in the scan, and a
        // coalesced copy instead of the permutation (tunable tile_debug bit 134217728: the scratch slices + k_permute, for A/B)
        L.d_region_recs = nullptr;
        L.region_cap = (uint32_t)(R / 2 + 1);
        const uint64_t ww_rec_bytes = (uint64_t)L.n_regions * L.region_cap * 12;
        if (!(tunables().tile_debug & 134217728) && ww_rec_bytes <= (24ull << 30)) {
            // (about 6 bytes per haystack unit: on a device that cannot spare them the call falls back to the scratch slices +
            // k_permute instead of failing; tunable tile_debug bit 2^40: the allocation "fails", for the test of that path)
            rc = (tunables().tile_debug & (1ll << 40)) ? ACGPU_E_NOMEM : d.ww_recs.ensure(ww_rec_bytes + 64);
            if (rc == ACGPU_OK) {
                L.d_region_recs = (int32_t *)d.ww_recs.p;
                ww_direct = true;
            } else if (rc != ACGPU_E_NOMEM) {
                return rc;
            }
        }
        // The fused tail of k_ww_pp (TileLaunch::fused_tail, ft_total16): no counts, prefix sums or copy pass behind the scan -- a
        // wave's records go to its own area and, when the workgroups with lower numbers are done, from there to their final
        // place.  (Tunable ww_ramp_pm: spans that grow with the workgroup's number, so that copies would run while later
        // workgroups still scan -- measured slower at every slope, 0 by default: EXPERIMENTS.md, round 6.)
        // Tunable tile_form bit 2: never (the region-local slots + k_ww_compact: A/B, tests).
        if (ww_direct && !fused_only && !(tunables().tile_form & 2) && ww_pp_serves(Tov ? *Tov : d.T, L)) {
            // (tunable ww_block: workgroups of fewer waves, two to a CU when their LDS allows -- A/B)
            const int64_t wb = tunables().ww_block;
            const int block_ft = (wb >= 64 && wb <= 1024 && wb % 64 == 0) ? (int)wb : L.block;
            const uint64_t wpb = (uint64_t)block_ft / 64;
            // (two workgroups: when each needs at most half the LDS, and for the 16-unit form only -- the 32-unit form's registers allow four waves per SIMD)
            const uint64_t per_cu = wb > 0 && t.max_len <= 16 && ww_pp_lds_total(Tov ? *Tov : d.T, L, block_ft) <= 80 * 1024 ? 2 : 1;
            const uint64_t tiles = (sh->own_end - base8 + 511) / 512, total16 = (tiles + wpb - 1) / wpb;
            const uint64_t G = std::min<uint64_t>((uint64_t)d.n_cu * per_cu, total16);
            const uint64_t area_recs = total16 * wpb * 512 / 2 + G * wpb + 8;
            if (G >= 1 && G <= (uint64_t)kMaxSlices && area_recs < (1ull << 32) && (rc = d.ww_recs.ensure(area_recs * 12 + 64)) == ACGPU_OK) {
                fused_tail = true;
                L.d_region_recs = (int32_t *)d.ww_recs.p;
                L.fused_tail = 1;
                L.grid = (int)G;
                L.block = block_ft;
                L.ft_total16 = (uint32_t)total16;
                const int64_t ramp = tunables().ww_ramp_pm;
                L.ft_ramp_pm = (uint32_t)(ramp < 0 ? 0 : std::min<int64_t>(ramp, 1000));
                L.d_out = d_out;
                L.out_cap = cap;
                L.out_map = record_kind == ACGPU_REC_MAP ? 1 : 0;
                L.d_id_map = nullptr;
                unsigned long long *h_slot_t = tk ? tk->h_count : d.h_counter, *d_slot_t = nullptr;
                HIP_TRY(hipHostGetDevicePointer((void **)&d_slot_t, h_slot_t, 0));
                L.tail_result = d_slot_t;
                L.tail_d_result = reinterpret_cast<acgpu_device_result *>(sh->d_result);
                L.tail_zero_counters = counters_next;
            } else if (rc != ACGPU_OK && rc != ACGPU_E_NOMEM) {
                return rc;
            }
        }
        if (!fused_tail) HIP_TRY(hipMemsetAsync(d.chunk_counts.p, 0, (size_t)L.n_regions * 4, stream));
#ifdef ACGPU_TIMING
        static DevBuf ww_timing;
        if ((rc = ww_timing.ensure((size_t)L.grid * 16 * 8 * 8))) return rc;
        HIP_TRY(hipMemsetAsync(ww_timing.p, 0, (size_t)L.grid * 16 * 8 * 8, stream));
        L.d_timing = (unsigned long long *)ww_timing.p;
#endif
        if (timed) { // (the kernel's own dispatch timestamps: no marker packets around it)
            L.ev_start = ev[0];
            L.ev_stop = ev[1];
            ext_timed = true;
        }
        if (fused_tail) L.ev_stop = (timed || tk) ? ev[2] : nullptr; // the scan is the call's only kernel: its end is the call's
        HIP_TRY(launch_ww_tile(Tov ? *Tov : d.T, L, stream, &kname));
#ifdef ACGPU_TIMING
        if (!tk) { // where a wave's time goes (s_memtime ticks, 100 MHz), averaged over the waves
            HIP_TRY(hipStreamSynchronize(stream));
            std::vector<unsigned long long> h((size_t)L.grid * 16 * 8);
            HIP_TRY(hipMemcpy(h.data(), ww_timing.p, h.size() * 8, hipMemcpyDeviceToHost));
            double sum[8] = {0}; size_t nw = 0;
            for (size_t w = 0; w < h.size() / 8; ++w) {
                if (!h[w * 8]) continue;
                nw++;
                for (int i = 0; i < 8; ++i) sum[i] += (double)h[w * 8 + i];
            }
            if (fused_tail) { // the workgroups in the order of their numbers: scan end, counts there, copy done (s_memtime ticks from the first scan end)
                unsigned long long t0 = ~0ull;
                for (size_t w = 0; w < h.size() / 8; ++w) if (h[w * 8]) t0 = std::min(t0, h[w * 8]);
                const size_t G = (size_t)L.grid;
                for (size_t b0 = 0; b0 < G; b0 += std::max<size_t>(G / 16, 1)) {
                    double se = 0, be = 0, ce = 0; size_t k = 0;
                    for (size_t b = b0; b < std::min(G, b0 + std::max<size_t>(G / 16, 1)); ++b)
                        for (size_t w = b * 16; w < b * 16 + 16; ++w) if (h[w * 8]) { se = std::max(se, (double)(h[w * 8] - t0)); be = std::max(be, (double)(h[w * 8 + 1] - t0)); ce = std::max(ce, (double)(h[w * 8 + 2] - t0)); k++; }
                    fprintf(stderr, "[ww fused tail] workgroups %3zu..: last scan end %8.0f | counts below there %8.0f | last copy done %8.0f\n", b0, se, be, ce);
                }
            } else
            if (nw) fprintf(stderr, "[ww timing] waves %zu total %.0f | windows %.0f | chunk1 %.0f | chunk2 %.0f | hash+bloom %.0f | probes %.0f | emission %.0f | calls %.1f\n",
                            nw, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw, sum[5] / nw, sum[6] / nw, sum[7] / nw);
        }
#endif
        n_chunks = L.n_regions;
        chunk_units = L.region_units;
        scanned = own_len;
        ww_region_cap = L.region_cap;
    } else if (use_tile_kernel(t)) {
        TileLaunch L{};
        L.block = tile_block_threads();
        const int waves_per_block = L.block / 64;
        // regions of 16384 units, or 32768 when that still leaves every wave two of them (fewer forced drains: -1.1 % at
        // config 2 in interleaved A/B; 65536 was no better)
        uint64_t R = tunables().region_units > 0 ? (uint64_t)tunables().region_units
                     : own_len >= 2ull * 32768 * d.n_cu * waves_per_block ? 32768 : 16384;
        if (tunables().region_units <= 0 && own_len >= 2ull * 16384 * d.n_cu * waves_per_block) { // long shards: regions that fill the waves evenly
            const uint64_t Rb = balanced_region_units(sh->own_end - (sh->own_begin & ~7ull), (uint64_t)d.n_cu * waves_per_block, tile_group_units(),
                                                      12288, 16, 32768);
            if (Rb) R = Rb;
        }
        { const uint64_t g = tile_group_units(); R = std::max<uint64_t>(g, (R + g - 1) / g * g); }
        L.region_units = (uint32_t)R;
        const uint64_t base8 = sh->own_begin & ~7ull; // regions are laid out from the 16-byte aligned start
        L.n_regions = (uint32_t)((sh->own_end - base8 + R - 1) / R);
        const uint64_t waves_max = (uint64_t)d.n_cu * waves_per_block;
        L.regions_per_wave = (uint32_t)((L.n_regions + waves_max - 1) / waves_max);
        const uint64_t waves_used = ((uint64_t)L.n_regions + L.regions_per_wave - 1) / L.regions_per_wave;
        L.grid = (int)((waves_used + waves_per_block - 1) / waves_per_block);
        perm_base = (uint32_t)base8;
        id_map = d.T.rterm;
        L.d_hay = sh->d_hay;
        L.n_units = (uint32_t)sh->n_units;
        L.own_begin = (uint32_t)sh->own_begin;
        L.own_end = (uint32_t)sh->own_end;
        L.cap = scratch_cap;
        L.lds_bytes = tile_lds_bytes(d.T, L.block);
        L.debug = (uint32_t)tunables().tile_debug;
        L.d_overflow = overflow_word;
        // one scratch slice and slot counter per workgroup (the redo after an overflow takes one slice)
        if (!fused_only && L.grid > 1 && !(L.debug & 16384u)) { // 16384: A/B, one counter
            n_slices = (uint32_t)std::min<int>(L.grid, kMaxSlices);
            slice_slots = scratch_cap / n_slices;
        }
        L.n_slices = n_slices;
        L.slice_slots = (uint32_t)slice_slots;
        L.wg_sums = 0;
        if ((rc = d.chunk_counts.ensure((size_t)L.n_regions * 4))) return rc;
        if ((rc = d.offsets.ensure((size_t)L.n_regions * 8))) return rc;
        if ((rc = d.scan_tmp.ensure(((size_t)L.n_regions / 2048 + 2) * 8))) return rc;
        L.d_scratch = (ScratchRec *)d.scratch.p;
        L.d_counter = counters;
        L.d_region_counts = (uint32_t *)d.chunk_counts.p;
        // (every region's count is written by the wave that owns the region: no memset)
        split = !fused_only && use_split_form(d.T);
        if (split) {
            L.n_slices = n_slices = 1; // (the verification kernel's grid is not the filter's)
            L.slice_slots = (uint32_t)(slice_slots = scratch_cap);
            // a wave's slice holds one candidate per 8 units of its span (the filter passes ~2 % on selective
            // dictionaries); a haystack that needs more is redone with the fused kernel
            const uint64_t per_wave = (uint64_t)L.regions_per_wave * R / (uint64_t)std::max<int64_t>(1, tunables().split_cand_div) + 2 * 1024;
            if (per_wave * waves_used >= (1ull << 32)) split = false;
            else {
                L.cands_per_wave = (uint32_t)per_wave;
                if ((rc = d.cands.ensure(per_wave * waves_used * 4 + 64))) return rc;
                if ((rc = d.region_cands.ensure((size_t)L.n_regions * 8))) return rc;
                L.d_cands = (uint32_t *)d.cands.p;
                L.d_region_cands = (uint2 *)d.region_cands.p;
                HIP_TRY(hipMemsetAsync(d.region_cands.p, 0, (size_t)L.n_regions * 8, stream)); // unwritten = no candidates
                L.verify_grid = (int)std::min<uint64_t>(((uint64_t)L.n_regions + 3) / 4, (uint64_t)d.n_cu * 8);
                // every verification wave may hold one partly used reservation of scratch slots
                const uint64_t need = std::min<uint64_t>(std::max<uint64_t>(cap, 1) + (uint64_t)L.verify_grid * 4 * tile_reserve_slots(),
                                                         0xffffffe0ull);
                if (need > L.cap) {
                    if ((rc = d.scratch.ensure(need * sizeof(ScratchRec)))) return rc;
                    L.cap = scratch_cap = need;
                    L.slice_slots = (uint32_t)(slice_slots = scratch_cap);
                    L.d_scratch = (ScratchRec *)d.scratch.p;
                }
            }
        }
        // one finalize launch instead of three (prefix-sum kernels + permute) when a scratch slice is a workgroup: the scan
        // kernel leaves every workgroup's record count next to its slot counter and the permute pass (k_permute_wg) derives
        // its offsets from those and the region counts of its own workgroup.  (tunable tile_debug bit 262144: the old way)
        fused_finalize = !split && n_slices == (uint32_t)L.grid && L.grid <= kMaxSlices &&
                         (uint64_t)waves_per_block * L.regions_per_wave <= kPermuteWgRegions && !(L.debug & 262144u);
        L.wg_sums = fused_finalize ? 1u : 0u;
        regions_per_wg = (uint32_t)waves_per_block * L.regions_per_wave;
        // The fused tail (TileLaunch::fused_tail): no finalize launch at all -- the scan's workgroups put their own slices in order
        // when their spans are scanned, each behind the counts of the workgroups that started before it, and the last one
        // reports the call's result.  Same conditions as the one-launch finalize.  Tunable tile_form bit 1: never (A/B, tests).
        if (fused_finalize && !(tunables().tile_form & 1)) {
            fused_tail = true;
            L.fused_tail = 1;
            L.wg_sums = 0; // (the workgroups' sums go through their own LDS)
            L.d_out = d_out;
            L.out_cap = cap;
            L.out_map = record_kind == ACGPU_REC_MAP ? 1 : 0;
            L.d_id_map = id_map;
            unsigned long long *h_slot_t = tk ? tk->h_count : d.h_counter, *d_slot_t = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void **)&d_slot_t, h_slot_t, 0));
            L.tail_result = d_slot_t;
            L.tail_d_result = reinterpret_cast<acgpu_device_result *>(sh->d_result);
            L.tail_zero_counters = counters_next;
        }
#ifdef ACGPU_TIMING
        static DevBuf timing;
        if ((rc = timing.ensure((size_t)L.grid * 16 * 8 * 8))) return rc;
        HIP_TRY(hipMemsetAsync(timing.p, 0, (size_t)L.grid * 16 * 8 * 8, stream));
        L.d_timing = (unsigned long long *)timing.p;
#endif
        if (split) {
            if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
            HIP_TRY(launch_ac_filter(d.T, L, stream, &kname));
            if (timed) HIP_TRY(hipEventRecord(ev[1], stream)); // the verification is accounted with the ordering
            HIP_TRY(launch_ac_verify(d.T, L, stream));
        } else {
            if (timed) { // (the kernel's own dispatch timestamps: no marker packets around it)
                L.ev_start = ev[0];
                L.ev_stop = ev[1];
                ext_timed = true;
            }
            if (fused_tail) L.ev_stop = (timed || tk) ? ev[2] : nullptr; // the scan is the call's only kernel: its end is the call's
            HIP_TRY(launch_ac_tile(d.T, L, stream, &kname));
        }
#ifdef ACGPU_TIMING
        if (!tk && !split) {
            HIP_TRY(hipStreamSynchronize(stream));
            std::vector<unsigned long long> h((size_t)L.grid * 16 * 8);
            HIP_TRY(hipMemcpy(h.data(), timing.p, h.size() * 8, hipMemcpyDeviceToHost));
            double sum[8] = {0}, mx0 = 0, vt[4] = {0, 0, 0, 0}; size_t nw = 0;
            for (size_t w = 0; w < h.size() / 8; ++w) {
                if (!h[w * 8]) continue;
                nw++;
                for (int i = 0; i < 6; ++i) sum[i] += (double)h[w * 8 + i];
                vt[0] += (double)(h[w * 8 + 6] & 0xffffffffu); vt[1] += (double)(h[w * 8 + 6] >> 32);
                vt[2] += (double)(h[w * 8 + 7] & 0xffffffffu); vt[3] += (double)(h[w * 8 + 7] >> 32);
                mx0 = std::max(mx0, (double)h[w * 8]);
            }
            { // spread of the waves' durations: per XCD (workgroup modulo 8) and per workgroup
                double xs[8] = {0}, xm[8] = {0}; size_t xn[8] = {0}; double bmin = 1e30, bmax = 0, wmin = 1e30;
                for (size_t b = 0; b < (size_t)L.grid; ++b) {
                    double bs = 0; size_t bn = 0;
                    for (size_t w = b * 16; w < b * 16 + 16; ++w) if (h[w * 8]) { bs += (double)h[w * 8]; bn++; xm[b % 8] = std::max(xm[b % 8], (double)h[w * 8]); wmin = std::min(wmin, (double)h[w * 8]); }
                    if (!bn) continue;
                    xs[b % 8] += bs; xn[b % 8] += bn;
                    bmin = std::min(bmin, bs / bn); bmax = std::max(bmax, bs / bn);
                }
                fprintf(stderr, "[timing] wave min %.0f; workgroup averages %.0f .. %.0f; per XCD avg/max:", wmin, bmin, bmax);
                for (int x = 0; x < 8; ++x) if (xn[x]) fprintf(stderr, " %.0f/%.0f", xs[x] / xn[x], xm[x]);
                fprintf(stderr, "\n[timing] by wave slot in the workgroup:");
                for (size_t sl = 0; sl < 16; ++sl) {
                    double a = 0; size_t n2 = 0;
                    for (size_t b = 0; b < (size_t)L.grid; ++b) if (h[(b * 16 + sl) * 8]) { a += (double)h[(b * 16 + sl) * 8]; n2++; }
                    fprintf(stderr, " %.0f", n2 ? a / n2 : 0.0);
                }
                fprintf(stderr, "\n");
            }
            if (nw) fprintf(stderr, "[timing] verification: windows %.0f | K-gram nodes %.0f | walks %.0f | emission %.0f\n", vt[0] / nw, vt[1] / nw, vt[2] / nw, vt[3] / nw);
            if (nw) fprintf(stderr, "[timing] waves %zu  total avg %.0f max %.0f | stream wait %.0f | drain %.0f (%.1f calls) | filter+L2 %.0f | passes %.1f  (s_memtime ticks, 100 MHz)\n",
                            nw, sum[0] / nw, mx0, sum[1] / nw, sum[2] / nw, sum[5] / nw, sum[3] / nw, sum[4] / nw);
        }
#endif
        n_chunks = L.n_regions;
        chunk_units = L.region_units;
        scanned = own_len;
    } else {
        ScanLaunch L{};
        L.block = scan_block_threads();
        L.grid = d.n_cu * (int)std::max<int64_t>(1, tunables().blocks_per_cu);
        // tile_debug bit 2^43: the one-chain kernel of rounds 1-3 (A/B); bit 2^45 (ablation build): no lookups in global memory
        L.debug = (uint32_t)((tunables().tile_debug >> 43) & 5);
        if (sh->n_units < 64) L.debug |= 1u; // (k_ac_dfa takes the buffer's last vector whole: the old kernel reads unit by unit)
        const uint64_t lanes = (uint64_t)L.grid * L.block * (uint64_t)((L.debug & 1u) ? 1 : std::max(1, scan_chains(d.T)));
        uint64_t C = tunables().chunk_units > 0 ? (uint64_t)tunables().chunk_units
                                                : std::max<uint64_t>({(own_len + lanes - 1) / lanes, 256, 16ull * halo});
        C = std::max<uint32_t>(8, round_up8(C));
        L.chunk_units = (uint32_t)C;
        L.n_chunks = (uint32_t)((own_len + C - 1) / C);
        // do not launch more workgroups than there are chunks
        L.grid = (int)std::min<uint64_t>((uint64_t)L.grid, ((uint64_t)L.n_chunks + L.block - 1) / L.block);
        L.d_hay = sh->d_hay;
        L.n_units = (uint32_t)sh->n_units;
        L.own_begin = (uint32_t)sh->own_begin;
        L.own_end = (uint32_t)sh->own_end;
        L.cap = scratch_cap; // every slot below min(counter, scratch_cap) must be written: the permute pass reads them all
        L.lds_bytes = scan_queue_bytes(L.block) + (size_t)d.T.lds_entries * t.entry_bytes + 16;
        if ((rc = d.chunk_counts.ensure((size_t)L.n_chunks * 4))) return rc;
        if ((rc = d.offsets.ensure((size_t)L.n_chunks * 8))) return rc;
        if ((rc = d.scan_tmp.ensure(((size_t)L.n_chunks / 2048 + 2) * 8))) return rc;
        L.d_scratch = (ScratchRec *)d.scratch.p;
        L.d_counter = counters;
        L.d_chunk_counts = (uint32_t *)d.chunk_counts.p;
        if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
        HIP_TRY(launch_ac_scan(d.T, L, stream, &kname));
        if (timed) HIP_TRY(hipEventRecord(ev[1], stream));
        n_chunks = L.n_chunks;
        chunk_units = L.chunk_units;
        scanned = own_len + (uint64_t)L.n_chunks * halo;
    }
    if (fused_tail) { // nothing behind the scan kernel: it has ordered its records, reports the count and has zeroed the other counter set
        d.cclean[1 - cs] = true;
        d.cset = 1 - cs;
        if (tk) {
            tk->shard = *sh;
            tk->record_kind = record_kind;
            tk->d_out = d_out;
            tk->stream = stream;
            tk->done_is_ev2 = true;
            tk->one_kernel = true;
            tk->scanned = scanned;
            std::snprintf(tk->kname, sizeof(tk->kname), "%s", kname);
            return ACGPU_OK;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if ((uint32_t)d.h_counter[1] != 0) // a scratch slice overflowed: fused kernel, one scratch slice
            return match_all(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, nullptr, true, Tov);
        *n_out = *d.h_counter;
        if (!ww) d.all_density = (double)*n_out / (double)own_len;
        if (prof) {
            HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[2]));
            prof->finalize_ms = 0.0f;
            prof->total_ms = prof->scan_ms;
            prof->scan_units = scanned;
            prof->n_matches = *n_out;
            std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "%s", kname);
        }
        return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
    }
    if (!fused_finalize)
        HIP_TRY(launch_exclusive_scan((const uint32_t *)d.chunk_counts.p, n_chunks, (uint64_t *)d.offsets.p,
                                      (uint64_t *)d.scan_tmp.p, stream));
    // the permute pass reports {record count, overflow word} into the pinned host slot of this call, clears the word and
    // zeroes the other set of slot counters for the next call: no copy or memset operations on the stream
    unsigned long long *h_slot = tk ? tk->h_count : d.h_counter, *d_slot = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&d_slot, h_slot, 0));
    const PermuteTail tail{d_slot, (const uint64_t *)d.scan_tmp.p + scan_tiles_for(n_chunks), overflow_word, counters_next,
                           reinterpret_cast<acgpu_device_result *>(sh->d_result)};
    // (profiled: the last kernel of the finalize delivers its own end timestamp where it is ONE kernel that ends the call)
    // (a ticket's completion is that timestamp too, profiled or not: no marker packet behind the call)
    const bool ext_stop = ((timed && ext_timed) || (tk != nullptr && !split)) && (ww_direct || fused_finalize);
    if (ww_direct)
        HIP_TRY(launch_ww_compact((const int32_t *)d.ww_recs.p, ww_region_cap, (const uint32_t *)d.chunk_counts.p, (const uint64_t *)d.offsets.p,
                                  n_chunks, record_kind, d_out, cap, stream, &tail, ext_stop ? ev[2] : nullptr));
    else if (fused_finalize)
        HIP_TRY(launch_permute_wg((const ScratchRec *)d.scratch.p, counters, n_slices, slice_slots, (const uint32_t *)d.chunk_counts.p,
                                  n_chunks, regions_per_wg, perm_base, chunk_units, record_kind, d_out, cap, id_map, stream, &tail,
                                  ext_stop ? ev[2] : nullptr));
    else
        HIP_TRY(launch_permute((const ScratchRec *)d.scratch.p, counters, n_slices, slice_slots,
                               (const uint64_t *)d.offsets.p, perm_base, chunk_units, by_start, record_kind, d_out, cap, id_map,
                               stream, &tail));
    d.cclean[1 - cs] = true; // zeroed by the pass just launched
    d.cset = 1 - cs;
    if (timed && !ext_stop) HIP_TRY(hipEventRecord(ev[2], stream));
    // exact record count = grand total of the per-chunk counts (the slot counter also counts reservation holes)
    if (tk) {
        tk->shard = *sh;
        tk->record_kind = record_kind;
        tk->d_out = d_out;
        tk->stream = stream;
        // (the call's last kernel delivered its end to ev[2]: that IS the call's completion -- one marker packet less per step)
        tk->done_is_ev2 = ext_stop;
        if (!ext_stop) HIP_TRY(hipEventRecord(tk->done, stream));
        tk->scanned = scanned;
        std::snprintf(tk->kname, sizeof(tk->kname), "%s", kname);
        return ACGPU_OK;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if ((uint32_t)d.h_counter[1] != 0) // a candidate slice / scratch slice overflowed: fused kernel, one scratch slice
        return match_all(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, nullptr, true, Tov);
    *n_out = *d.h_counter;
    if (!ww) d.all_density = (double)*n_out / (double)own_len;
    if (prof) {
        HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
        HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, d.ev[1], d.ev[2]));
        HIP_TRY(hipEventElapsedTime(&prof->total_ms, d.ev[0], d.ev[2]));
        prof->scan_units = scanned;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "%s", kname);
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// Marks the chain k0, nxt[k0], nxt[nxt[k0]], ... (nxt[k] in (k, M], nxt[M] = M; d_mark[k0] = 1 on entry, every other mark 0):
// afterwards d_mark[k] = 1 exactly for the chain's elements.  One pass when the jumps are short: with nxt[k] - k as the
// "length" this is the greedy chain of LongestMatchSet, and the Longest chain kernels mark it (tiles of indices with
// synchronisation points, one lane per tile, a bit per visited index; acgpu_longest.hip, acgpu_wwlongest.hip: k_wwl_jumps);
// pointer doubling -- ceil(log2 M) rounds over all M elements -- otherwise (tiny inputs, jumps beyond 16 bits, tunable
// tile_debug bit 2097152).  The doubling squares the jump table: *d_nxt_kept is where the successors survive
// (d_nxt itself, or d_nxt_copy -- M + 1 words, may be null if the caller does not need them).
// jump_bound: what the caller knows nxt[k] - k cannot exceed (0: unknown -- the largest jump is measured).  The one pass has a
// fixed cost (a read-back, four small launches) that 21 doubling rounds over a million elements do not reach: it is taken
// from 4 M elements on.
int mark_chain(DeviceState &d, uint32_t *d_nxt, uint32_t *d_tmp, uint32_t *d_mark, uint32_t M, hipStream_t stream,
               uint32_t *d_nxt_copy, const uint32_t **d_nxt_kept, uint32_t jump_bound) {
    int rc;
    if (d_nxt_kept) *d_nxt_kept = d_nxt;
    // (tunable tile_debug: bit 2097152 = always the doubling, bit 4194304 = the one pass from 64 elements on -- tests)
    const uint32_t one_pass_from = (tunables().tile_debug & 4194304) ? 64u : (1u << 22);
    bool one_pass = M >= one_pass_from && jump_bound <= 60000 && !(tunables().tile_debug & 2097152);
    uint64_t head = ~0ull, max_jump = 0;
    if (one_pass) {
        // counter words used here: [2] chain head, [3] largest jump (bytes 16..32).  They lie inside match_all's first slot
        // counter line (word 0 = slot counter, word 1 = workgroup sum, the rest of the 128-byte line is padding), which every
        // caller marks dirty (cclean[0] = false) so that the next AhoCorasick call clears it
        static_assert(kCounterStride >= 4, "mark_chain keeps its head and largest jump in words 2 and 3 of the first counter line");
        if ((rc = d.counter.ensure(64))) return rc;
        if ((rc = d.lenbuf.ensure((size_t)M * 2 + 128))) return rc;
        if ((rc = d.blockmax.ensure(((size_t)M / 64 + 2) * 4))) return rc;
        HIP_TRY(hipMemsetAsync((char *)d.counter.p + 16, 0xff, 8, stream)); // the chain head's index (none: all ones)
        HIP_TRY(hipMemsetAsync((char *)d.counter.p + 24, 0, 8, stream));    // the largest jump
        HIP_TRY(launch_wwl_jumps(d_nxt, d_mark, M, (uint16_t *)d.lenbuf.p, (uint32_t *)d.blockmax.p,
                                 (unsigned long long *)d.counter.p + 2, jump_bound == 0, stream));
        HIP_TRY(hipMemcpyAsync(d.h_counter + 3, (const char *)d.counter.p + 16, 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        head = d.h_counter[3];
        max_jump = jump_bound ? jump_bound : d.h_counter[4];
        if (head >= M) return ACGPU_OK; // no head: nothing is marked, nothing to mark
        if (max_jump > 60000) one_pass = false;
    }
    if (!one_pass) {
        if (d_nxt_copy) {
            HIP_TRY(hipMemcpyAsync(d_nxt_copy, d_nxt, ((size_t)M + 1) * 4, hipMemcpyDeviceToDevice, stream));
            if (d_nxt_kept) *d_nxt_kept = d_nxt_copy;
        }
        HIP_TRY(launch_chain_mark(d_nxt, d_tmp, d_mark, M, stream));
        return ACGPU_OK;
    }
    // (jumps are ~1, so a lane makes about one step per index: short tiles, i.e. many lanes)
    const uint32_t tile_units = 512;
    const size_t bit_bytes = ((size_t)M / 128 + 2) * 16;
    if ((rc = d.chainbits.ensure(bit_bytes))) return rc;
    HIP_TRY(hipMemsetAsync(d.chainbits.p, 0, bit_bytes, stream));
    LongestChainLaunch Cn{};
    Cn.d_len = d.lenbuf.p;
    Cn.len_bytes = 2;
    Cn.own_begin = 0;
    Cn.own_end = M;
    Cn.d_blockmax = (const uint32_t *)d.blockmax.p;
    Cn.entry = (uint32_t)head;
    Cn.tile_units = tile_units;
    Cn.n_tiles = (uint32_t)(((uint64_t)M - head + tile_units - 1) / tile_units);
    Cn.max_len = (uint32_t)std::max<uint64_t>(max_jump, 1);
    if ((rc = d.chunk_counts.ensure((size_t)Cn.n_tiles * 4))) return rc;
    Cn.d_counts = (uint32_t *)d.chunk_counts.p; // (per-tile counts nobody reads)
    Cn.d_exit = (unsigned long long *)d.counter.p + 3;
    Cn.len_units = M;
    Cn.d_bits = (uint32_t *)d.chainbits.p;
    Cn.record_kind = ACGPU_REC_SET;
    if ((rc = d.chain.ensure((size_t)Cn.n_tiles * 4 + 64))) return rc;
    HIP_TRY(launch_longest_sync(Cn, (uint32_t *)d.chain.p, stream));
    HIP_TRY(launch_longest_chain_lds(Cn, (const uint32_t *)d.chain.p, /*write_pass=*/false, stream));
    HIP_TRY(launch_wwl_bits_to_mark((const uint32_t *)d.chainbits.p, M, d_mark, stream));
    return ACGPU_OK;
}

// LONGEST over a dictionary whose suffix filter is selective: matches are sparse, so leftmost-longest is a selection
// over the all-matches list (the AhoCorasick tile pipeline into an internal buffer + k_long_next + chain marking)
// instead of a trie walk from every position.  Returns ACGPU_E_UNSUPPORTED when the haystack turns out to be dense in
// matches (the caller then takes the walk).
int match_longest_sparse(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                         uint64_t *n_out, hipStream_t stream, acgpu_profile *prof, uint64_t entry) {
    const HostTables &t = a->t;
    const uint64_t halo = t.max_len > 0 ? t.max_len - 1 : 0;
    const uint64_t own_len = sh->own_end - sh->own_begin;
    acgpu_shard all = *sh; // every occurrence that ENDS in the owned range or its right halo
    all.own_end = std::min<uint64_t>(sh->n_units, sh->own_end + halo);
    all.text_begin = 1; // occurrences that begin before the buffer begin before own_begin: not ours anyway
    int rc;
    uint64_t m = 0;
    const uint64_t dense_limit = own_len / 4 + 4096;
    uint64_t acap = std::max<uint64_t>(d.short_recs.bytes > 16 ? (d.short_recs.bytes - 16) / ACGPU_REC_MAP : 0, own_len / 32 + (1 << 16));
    acgpu_profile all_prof;
    for (;;) {
        if ((rc = d.short_recs.ensure(acap * ACGPU_REC_MAP + 16))) return rc;
        rc = match_all(a, d, &all, ACGPU_REC_MAP, d.short_recs.p, acap, &m, stream, prof ? &all_prof : nullptr);
        if (rc == ACGPU_E_OVERFLOW) {
            if (m > dense_limit) return ACGPU_E_UNSUPPORTED;
            acap = m;
            continue;
        }
        if (rc != ACGPU_OK) return rc;
        break;
    }
    if (m > dense_limit) return ACGPU_E_UNSUPPORTED;
    if (prof) *prof = all_prof;
    *n_out = 0;
    sh->chain_exit = (int64_t)std::max<uint64_t>(entry, sh->own_end);
    if (m == 0) return ACGPU_OK;
    const uint32_t M = (uint32_t)m;
    if ((rc = d.short_nxt.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_tmp.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_mark.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.offsets.ensure((size_t)M * 8))) return rc;
    if ((rc = d.scan_tmp.ensure(((size_t)M / 2048 + 2) * 8))) return rc;
    if ((rc = d.counter.ensure(64))) return rc;
    if (prof) HIP_TRY(hipEventRecord(d.ev[0], stream));
    HIP_TRY(launch_longest_select((const int32_t *)d.short_recs.p, M, (int64_t)entry, (int64_t)sh->own_end, t.max_len,
                                  (uint32_t *)d.short_nxt.p, (uint32_t *)d.short_tmp.p, (uint32_t *)d.short_mark.p, stream));
    if ((rc = mark_chain(d, (uint32_t *)d.short_nxt.p, (uint32_t *)d.short_tmp.p, (uint32_t *)d.short_mark.p, M, stream, nullptr,
                         nullptr, 0)))
        return rc;
    HIP_TRY(launch_exclusive_scan((const uint32_t *)d.short_mark.p, M, (uint64_t *)d.offsets.p, (uint64_t *)d.scan_tmp.p, stream));
    const uint64_t *d_total = (const uint64_t *)d.scan_tmp.p + scan_tiles_for(M);
    HIP_TRY(launch_shortest_emit((const int32_t *)d.short_recs.p, M, (const uint32_t *)d.short_mark.p,
                                 (const uint64_t *)d.offsets.p, d_total, record_kind, d_out, cap, (int64_t)entry,
                                 (unsigned long long *)d.counter.p, stream));
    d.cclean[0] = false; // (the exit position went where match_all's first set of slot counters lives)
    if (prof) HIP_TRY(hipEventRecord(d.ev[1], stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter, d_total, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter + 1, d.counter.p, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = d.h_counter[0];
    // the chain leaves the owned range at the end of its last match, or walks out of it one unit at a time
    if (*n_out) sh->chain_exit = (int64_t)std::max<uint64_t>(d.h_counter[1], sh->own_end);
    if (prof) {
        float sel_ms = 0;
        HIP_TRY(hipEventElapsedTime(&sel_ms, d.ev[0], d.ev[1]));
        prof->finalize_ms += sel_ms;
        prof->total_ms += sel_ms;
        prof->scan_units = own_len;
        prof->n_matches = *n_out;
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// LONGEST-mode pipeline on one shard: reverse scan -> chain count -> prefix sum -> chain write.
// With a ticket (the walk pipeline only: want_async_longest) the call returns after enqueueing; acgpu_match_device_end collects it.
int match_longest(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                  uint64_t *n_out, hipStream_t stream, acgpu_profile *prof, Ticket *tk = nullptr, int bits_level = 0) {
    const HostTables &t = a->t;
    hipEvent_t *ev = tk ? tk->ev : d.ev;
    const bool timed = tk ? tk->profiled : prof != nullptr;
    const uint32_t halo = t.max_len > 0 ? t.max_len - 1 : 0;
    if (!sh->text_end && sh->n_units - sh->own_end < halo) return ACGPU_E_INVALID; // right halo too short
    if (sh->chain_entry < (int64_t)sh->own_begin) return ACGPU_E_INVALID;
    if (prof) std::memset(prof, 0, sizeof(*prof));
    const uint64_t entry = (uint64_t)sh->chain_entry;
    sh->chain_exit = (int64_t)std::max<uint64_t>(entry, sh->own_end);
    if (entry >= sh->own_end || t.n_states <= 1) {
        *n_out = 0;
        if (entry < sh->own_end) sh->chain_exit = (int64_t)sh->own_end;
        if (tk) { // (no kernel will write the slot)
            tk->h_count[0] = tk->h_count[1] = 0;
            tk->h_count[2] = (unsigned long long)sh->chain_exit;
            tk->profiled = false;
            tk->stream = stream;
            if (sh->d_result) HIP_TRY(hipMemsetAsync(sh->d_result, 0, sizeof(acgpu_device_result), stream));
            HIP_TRY(hipEventRecord(tk->done, stream));
        }
        return ACGPU_OK;
    }
    const uint64_t own_len = sh->own_end - sh->own_begin;
    if (!tk && filter_is_selective(t) && tunables().force_kernel != 1) { // selective suffix filter: selection over all matches
        const int src = match_longest_sparse(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, entry);
        if (src != ACGPU_E_UNSUPPORTED) return src;
        sh->chain_exit = (int64_t)std::max<uint64_t>(entry, sh->own_end); // dense in matches after all: the walk
    }
    // A two-letter alphabet in which every letter is a keyword: the text as one bit per unit, the chain's own positions only
    // (k_longest_bits, acgpu_longest_bits.hip) -- no length array, no synchronisation pass; Map records look their keyword ids up
    // by the matched text's own bits when they are written (keywords of up to 32 units; the rare longer ones by a walk).  The kernel checks
    // its own result (every segment's exit against the next one's entry) and raises the bail flag -- also for a unit outside
    // the alphabet --: the call is then redone right here, or in acgpu_match_device_end: once more with a run-up of a whole
    // segment (bits_level 1), then by the walk pipeline below (bits_level 2).
    // Tunable longest_form, bits: 1 = never, 4 = also for short texts (tests).
    const int64_t lform = tunables().longest_form;
    const bool bits_form = bits_level < 2 && (record_kind == ACGPU_REC_SET || d.T.bits_idkeys != nullptr) && d.T.bits_rk != 0 && !(lform & 1) &&
                           (own_len >= (1ull << 21) || (lform & 4)) && tunables().force_kernel == 0;
    if (bits_form) {
        int rc;
        LongestBitsLaunch Bl{};
        Bl.d_hay = sh->d_hay;
        Bl.n_units = (uint32_t)sh->n_units;
        Bl.own_end = (uint32_t)sh->own_end;
        Bl.entry = (uint32_t)entry;
        Bl.g0 = (uint32_t)entry & ~127u; // (bitmap words in groups of four: 16-byte stores)
        const uint32_t region_units = longest_bits_region_units();
        Bl.n_regions = (uint32_t)((sh->own_end - Bl.g0 + region_units - 1) / region_units);
        Bl.runup = bits_level == 0 ? longest_bits_seg_units() / 2 : longest_bits_seg_units();
        Bl.max_len = t.max_len;
        Bl.d_out = d_out;
        Bl.cap = cap;
        const size_t n_blk = ((size_t)Bl.n_regions + 63) / 64, state_words = 8 + (size_t)Bl.n_regions + n_blk + 2;
        // exit / flag / count, a word per region, a word per block of 64 regions, the region counter: zero at the start of a call --
        // the call's last kernel leaves them so; a memset only for a fresh (or larger) buffer and after a call that failed half way
        const size_t state_had = d.bits_state.bytes;
        if ((rc = d.bits_state.ensure(state_words * 8))) return rc;
        if (d.bits_state.p != d.bits_state_seen || d.bits_state.bytes != state_had) {
            HIP_TRY(hipMemsetAsync(d.bits_state.p, 0, d.bits_state.bytes, stream));
            d.bits_state_seen = d.bits_state.p;
        }
        if ((rc = d.blockmax.ensure((size_t)Bl.n_regions * 8 + 64))) return rc;
        if ((rc = d.chainbits.ensure((size_t)Bl.n_regions * longest_bits_region_scratch_bytes() + 64))) return rc;
        Bl.d_exit = (unsigned long long *)d.bits_state.p;
        Bl.d_agg = Bl.d_exit + 8;
        Bl.d_blk = Bl.d_agg + Bl.n_regions;
        Bl.d_next = (uint32_t *)(Bl.d_blk + n_blk);
        Bl.d_marks = (uint32_t *)d.chainbits.p;
        Bl.d_xout = Bl.d_marks + (size_t)Bl.n_regions * (region_units / 32);
        Bl.d_text = nullptr;
        if (record_kind == ACGPU_REC_MAP) { // the regions' text bits, parked for the keyword ids (acgpu_longest_bits.hip)
            if ((rc = d.lenbuf.ensure((size_t)Bl.n_regions * longest_bits_region_text_bytes() + 64))) return rc;
            Bl.d_text = (uint32_t *)d.lenbuf.p;
        }
        Bl.d_pred = (uint32_t *)d.blockmax.p;
        Bl.d_true = Bl.d_pred + Bl.n_regions;
        Bl.grid = (int)std::min<uint64_t>((uint64_t)d.n_cu, (Bl.n_regions + 15) / 16);
        Bl.debug = (uint32_t)(tunables().tile_debug >> 32);
        unsigned long long *h_slot = tk ? tk->h_count : d.h_counter, *d_slot = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&d_slot, h_slot, 0));
        void *const seen = d.bits_state_seen;
        d.bits_state_seen = nullptr; // (until both kernels are enqueued: an error below leaves the words in an unknown state)
        // the whole pipeline in two launches: text in, records out; then the seams, the result and the state for the next call.
        // (Profiled calls take the dispatches' own start / stop timestamps: marker packets between the steps cost more than the
        // second kernel does.)
        HIP_TRY(launch_longest_bits(d.T, Bl, d_slot, tk ? reinterpret_cast<acgpu_device_result *>(sh->d_result) : nullptr,
                                    (unsigned long long *)d.bits_state.p, (uint32_t)state_words, stream, timed ? ev[0] : nullptr,
                                    timed ? ev[1] : nullptr, timed ? ev[2] : nullptr));
        d.bits_state_seen = seen;
        if (tk) {
            tk->stream = stream;
            tk->shard = *sh;
            tk->record_kind = record_kind;
            tk->d_out = d_out;
            tk->bits_level = bits_level;
            tk->done_is_ev2 = timed; // (the finish kernel's own end)
            if (!timed) HIP_TRY(hipEventRecord(tk->done, stream));
            tk->scanned = own_len;
            std::snprintf(tk->kname, sizeof(tk->kname), "k_longest_bits");
            return ACGPU_OK;
        }
        HIP_TRY(hipStreamSynchronize(stream));
#ifdef ACGPU_ABLATION
        if (Bl.debug) {
            unsigned long long mism = 0;
            (void)hipMemcpy(&mism, (const char *)d.bits_state.p + 24, 8, hipMemcpyDeviceToHost); // (the finish kernel has zeroed it: kept for builds that skip it)
            fprintf(stderr, "[k_longest_bits debug %u] segments whose assumed entry was not the exit before them: %llu\n", Bl.debug, mism);
        }
#endif
        if (d.h_counter[1] != 0) // (1: a chain that did not merge inside the run-up; 2: a unit outside the alphabet)
            return match_longest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, nullptr, d.h_counter[1] == 1 ? bits_level + 1 : 2);
        *n_out = d.h_counter[0];
        sh->chain_exit = (int64_t)d.h_counter[2];
        if (prof) {
            HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
            HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, d.ev[1], d.ev[2]));
            HIP_TRY(hipEventElapsedTime(&prof->total_ms, d.ev[0], d.ev[2]));
            prof->scan_units = own_len;
            prof->n_matches = *n_out;
            std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "k_longest_bits");
        }
        return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
    }
    // Any other dense dictionary with range classes or small class pages, long texts, Set and Map records: the walks of the chain's own
    // positions only (k_longest_follow, acgpu_longest_follow.hip) in place of the length array, the synchronisation points and the chain
    // pass; the bitmaps, counts and first positions it leaves are what the prefix sum and k_longest_emit_ends below read.  It checks its
    // own result like k_longest_bits and is redone the same way.  Tunable longest_form, bits: 2 = never, 4 = also for short texts.
    const size_t fol_pages = (!t.range_cls && !t.dfa_pages.empty()) ? t.dfa_pages.size() * 2 : 0;
    const bool fol_classes = t.dense && ((t.range_cls && t.n_cls == t.cls_span + 1) || fol_pages > 0);
    const uint32_t fol_hot = fol_classes ? longest_follow_hot_rows(t.n_cls, t.n_states, (uint32_t)fol_pages) : 0;
    if (bits_level < d.fol_level && !bits_form) bits_level = d.fol_level; // (what earlier calls on this pool have learnt about its texts)
    // (Not where the walk pipeline has its root table -- dictionaries over up to four letters whose first 14 or 7 units one lookup
    // decides: config 4's dictionary with Map records 3.71 against 5.79 ms per 2^29 units, tools/longest_shapes.py.  Tunable
    // longest_form bit 8: there too, for A/B.)
    const bool follow_form = bits_level < 2 && fol_hot > 0 && !(lform & 2) && (own_len >= (1ull << 20) || (lform & 4)) && tunables().force_kernel == 0 &&
                             (t.root_b == 0 || (lform & 8));
    if (follow_form) {
        int rc;
        LongestFollowLaunch F{};
        F.d_hay = sh->d_hay;
        F.n_units = (uint32_t)sh->n_units;
        F.own_end = (uint32_t)sh->own_end;
        F.entry = (uint32_t)entry;
        F.g0 = (uint32_t)entry & ~31u;
        // a lane walks a segment of 1024 positions behind a run-up.  (Segments of 512 for texts that leave half the chip's lanes
        // without one were measured: 4.95 against 3.30 ms per 2^28 units of the README word list -- the kernel is bound by the
        // number of gathers, and a run-up of a whole segment is a third more of them.)  Tunable region_units (64 .. 1024): the first
        // try's run-up, for A/B.
        const uint32_t seg_units = longest_follow_seg_units();
        F.seg_log2 = 10;
        const uint32_t region_units = 64 * seg_units;
        F.n_regions = (uint32_t)((sh->own_end - F.g0 + region_units - 1) / region_units);
        // The first try's run-up is 128 positions: on a text with separators every chain lands on each of them (no keyword goes
        // across), so chains merge within a word, and the run-up is a fifth of the gathers at 512 (measured on the README word
        // list: 3.27 ms per 2^28 units at 512, 2.85 at 256, 2.65 at 128).  A text on which that fails -- the kernel notices --
        // is redone with a whole segment, and this pool remembers it (d.fol_level): the next call starts there, or, if chains
        // do not merge within 1024 positions either, with the walk pipeline.
        const int64_t ru = tunables().region_units;
        F.runup = bits_level == 0 ? (ru >= 64 && ru <= 1024 ? (uint32_t)ru : 128u) : seg_units;
        F.tile_log2 = 2; // (emit tiles of 4096 positions)
        F.hot_rows = fol_hot;
        const uint32_t n_tiles = (F.n_regions * (region_units / seg_units)) >> F.tile_log2;
        if ((rc = d.counter.ensure(64))) return rc;
        if ((rc = d.chunk_counts.ensure((size_t)n_tiles * 4))) return rc;
        if ((rc = d.offsets.ensure((size_t)n_tiles * 8))) return rc;
        if ((rc = d.scan_tmp.ensure(((size_t)n_tiles / 2048 + 2) * 8))) return rc;
        if ((rc = d.chain.ensure((size_t)n_tiles * 4 + 64))) return rc;
        if ((rc = d.blockmax.ensure((size_t)F.n_regions * 8 + 64))) return rc;
        const size_t bit_bytes = ((size_t)sh->n_units / 128 + 2) * 16;
        if ((rc = d.chainbits.ensure(bit_bytes * 2))) return rc;
        F.d_bits = (uint32_t *)d.chainbits.p;
        F.d_ebits = F.d_bits + bit_bytes / 4;
        F.d_state = nullptr;
        if (record_kind == ACGPU_REC_MAP) {
            if ((rc = d.statebuf.ensure((size_t)sh->n_units * 4 + 64))) return rc;
            F.d_state = (uint32_t *)d.statebuf.p;
        }
        F.d_sync = (uint32_t *)d.chain.p;
        F.d_counts = (uint32_t *)d.chunk_counts.p;
        F.d_exit = (unsigned long long *)d.counter.p;
        F.d_pred = (uint32_t *)d.blockmax.p;
        F.d_true = F.d_pred + F.n_regions;
        F.grid = (int)std::min<uint64_t>(2ull * d.n_cu, (F.n_regions + 15) / 16);
        HIP_TRY(hipMemsetAsync(d.counter.p, 0, 64, stream));
        d.cclean[0] = false; // (match_all's first set of slot counters lives here)
        { // the end bits are merged with atomicOr: zeros from the first word the chain can touch to where its last match can end
            const size_t first = F.g0 >> 5, last = std::min<size_t>(bit_bytes / 4, (((size_t)sh->own_end + t.max_len) >> 5) + 2);
            if (last > first) HIP_TRY(hipMemsetAsync(F.d_ebits + first, 0, (last - first) * 4, stream));
        }
        if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
        HIP_TRY(launch_longest_follow(d.T, F, t.range_cls, record_kind == ACGPU_REC_MAP, stream));
        if (timed) HIP_TRY(hipEventRecord(ev[1], stream));
        LongestChainLaunch Cn{};
        Cn.d_state = F.d_state;
        Cn.d_out_id = d.T.term_id;
        Cn.len_bytes = 1;
        Cn.own_begin = (uint32_t)sh->own_begin;
        Cn.own_end = (uint32_t)sh->own_end;
        Cn.entry = (uint32_t)entry;
        Cn.tile_units = seg_units << F.tile_log2;
        Cn.n_tiles = n_tiles;
        Cn.max_len = t.max_len;
        Cn.d_counts = F.d_counts;
        Cn.d_offsets = (const uint64_t *)d.offsets.p;
        Cn.d_out = d_out;
        Cn.cap = cap;
        Cn.record_kind = record_kind;
        Cn.d_exit = F.d_exit;
        Cn.len_units = (uint32_t)sh->n_units;
        Cn.d_bits = F.d_bits;
        Cn.d_ebits = F.d_ebits;
        HIP_TRY(launch_exclusive_scan(Cn.d_counts, Cn.n_tiles, (uint64_t *)d.offsets.p, (uint64_t *)d.scan_tmp.p, stream));
        HIP_TRY(launch_longest_emit(Cn, F.d_sync, stream));
        if (timed) HIP_TRY(hipEventRecord(ev[2], stream));
        unsigned long long *h_slot = tk ? tk->h_count : d.h_counter, *d_slot = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&d_slot, h_slot, 0));
        HIP_TRY(launch_publish_result((const unsigned long long *)d.scan_tmp.p + scan_tiles_for(Cn.n_tiles), (const unsigned long long *)d.counter.p,
                                      d_slot, tk ? reinterpret_cast<acgpu_device_result *>(sh->d_result) : nullptr, stream));
        if (tk) {
            tk->stream = stream;
            tk->shard = *sh;
            tk->record_kind = record_kind;
            tk->d_out = d_out;
            tk->bits_level = bits_level;
            HIP_TRY(hipEventRecord(tk->done, stream));
            tk->scanned = own_len;
            std::snprintf(tk->kname, sizeof(tk->kname), "k_longest_follow");
            return ACGPU_OK;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (d.h_counter[1] != 0) { // (a chain that did not merge inside the run-up)
            d.fol_level = std::max(d.fol_level, bits_level + 1);
            return match_longest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, nullptr, bits_level + 1);
        }
        *n_out = d.h_counter[0];
        sh->chain_exit = (int64_t)d.h_counter[2];
        if (prof) {
            HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
            HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, d.ev[1], d.ev[2]));
            HIP_TRY(hipEventElapsedTime(&prof->total_ms, d.ev[0], d.ev[2]));
            prof->scan_units = own_len;
            prof->n_matches = *n_out;
            std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "k_longest_follow");
        }
        return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
    }
    LongestScanLaunch S{};
    S.block = 1024;
    // two workgroups per CU share the LDS (hot trie rows: at most 72 KB each); a short haystack gets fewer (every
    // workgroup stages the rows before it starts)
    S.grid = (int)std::min<uint64_t>(2ull * d.n_cu, (own_len + S.block - 1) / S.block);
    S.chunk_units = 0;
    S.n_chunks = 0;
    S.d_hay = sh->d_hay;
    S.n_units = (uint32_t)sh->n_units;
    S.own_begin = (uint32_t)sh->own_begin;
    S.own_end = (uint32_t)sh->own_end;
    S.len_bytes = t.max_len < 65536 ? 2 : 4;
    uint32_t lds_rows = 0;
    // (table classes: the general walk keeps the class pages behind its rows -- acgpu_build.cpp 7c -- when they are small)
    // (measured, tools/longest_shapes.py: the README word list case-insensitive, 27 classes: 8.13 -> 6.08 ms per 2^28 units; 3000 CJK
    // units, where a row is 12 KB and the walk waits for the table in global memory anyway: 4.74 -> 5.05 -- so: small alphabets only)
    const size_t walk_pages = (t.dense && !t.range_cls && !t.dfa_pages.empty() && t.dfa_pages.size() * 2 <= 16 * 1024 && t.n_cls <= 512) ? t.dfa_pages.size() * 2 + 16 : 0;
    if (t.dense && t.n_cls) lds_rows = (uint32_t)std::min<uint64_t>(t.n_states, (72 * 1024 - walk_pages) / ((uint64_t)t.n_cls * 4));
    // range classes (case sensitive, keyword units within a span of 63) take the lean walk; tunable force_kernel=1
    // keeps the general one
    S.pairs = t.dense && t.range_cls && t.n_cls == t.cls_span + 1 && tunables().force_kernel != 1 &&
              (uint64_t)t.n_states * t.n_cls * 4 < (1ull << 31);
    if (S.pairs) lds_rows = (uint32_t)std::min<uint64_t>(t.n_states, (72 * 1024) / ((uint64_t)t.n_cls * 4) - 2);
    S.lds_rows = lds_rows;
    S.lds_bytes = std::max<size_t>((size_t)(lds_rows + (S.pairs ? 2 : 0)) * t.n_cls * 4, 16) + (S.pairs ? 0 : walk_pages);
    S.pages_bytes = S.pairs ? 0u : (uint32_t)walk_pages; // (0: classes from the table in global memory)
    // the work-list form of the range-class walk (tunable force_kernel=4 keeps the lock-step form): 16-bit lengths, LDS rows
    // below 64 KiB (the row offset is the low word of an entry), two workgroups per CU
    if (S.pairs && S.len_bytes == 2 && t.max_len < 64000 && tunables().force_kernel != 4) {
        S.pairs = 2;
        S.lds_rows = lds_rows = std::min<uint32_t>(t.n_states, longest_list_max_rows(t.n_cls, record_kind == ACGPU_REC_MAP));
        S.lds_bytes = longest_list_lds_bytes(record_kind == ACGPU_REC_MAP);
        // two workgroups per CU (Set records: 60 KiB of rows + 17 KiB of lists each; Map: 52 + 26)
        S.grid = (int)std::min<uint64_t>(2ull * d.n_cu, (own_len + 16 * 1024 - 1) / (16 * 1024));
    }
    int rc;
    // the work-list form stores ONE byte per length, 255 = "255 or more: see the 16-bit side array" (tunable tile_debug bit
    // 8388608: 16-bit lengths, for A/B) -- half the bytes written by the walk and read back by the chain passes
    if (S.pairs == 2 && !(tunables().tile_debug & 8388608)) {
        S.len_bytes = 1;
        if ((rc = d.lenbig.ensure((size_t)sh->n_units * 2 + 64))) return rc;
        S.d_len_big = (uint16_t *)d.lenbig.p;
    }
    if ((rc = d.lenbuf.ensure((size_t)sh->n_units * S.len_bytes + 64))) return rc;
    S.d_len = d.lenbuf.p;
    if ((rc = d.blockmax.ensure((own_len / 64 + 2) * 4))) return rc;
    S.d_blockmax = (uint32_t *)d.blockmax.p;
    S.d_state = nullptr;
    if (record_kind == ACGPU_REC_MAP) {
        if ((rc = d.statebuf.ensure((size_t)sh->n_units * 4 + 64))) return rc;
        S.d_state = (uint32_t *)d.statebuf.p;
    }
    // Set records over a small alphabet: k_longest_block (first round through the root table, the live walks through its own
    // work list), then the general kernel for the chunks it flagged (tunable tile_debug bit 16777216: the general kernel for
    // everything).  A wave's span must fit the 16-bit lane positions of its queue.
    const uint64_t blk_waves = (uint64_t)S.grid * (S.block / 64), blk_chunks = (own_len + 1023) / 1024;
    const uint64_t blk_span = (blk_chunks + blk_waves - 1) / std::max<uint64_t>(blk_waves, 1);
    const bool root_form = S.pairs == 2 && S.len_bytes == 1 && record_kind == ACGPU_REC_SET && d.T.root_b != 0 &&
                           !(tunables().tile_debug & 16777216) && blk_span * 1024 <= (1u << 19) && sh->n_units >= 4096 &&
                           (sh->own_begin & 7) == 0;
    if (root_form) {
        S.span_chunks = (uint32_t)blk_span;
        if ((rc = d.todo.ensure(blk_chunks + 64))) return rc;
        S.d_todo_w = (uint8_t *)d.todo.p;
    }
    LongestChainLaunch Cn{};
    // positions per chain lane: the synchronisation scan skips 64-position blocks that cannot reach the tile, so tiles
    // can stay small (more lanes, shorter dependent chains) even when keywords are long
    // (measured at config 4: 6144 positions per lane are best with 256-position chunks of one-byte lengths -- 4096: +15 %,
    // 8192: +3 %, 12288: +22 % for the chain passes; small inputs get more, shorter lanes)
    const uint64_t T_units = tunables().region_units > 0 ? (uint64_t)tunables().region_units : (own_len >= (1ull << 24) ? 6144 : 1024);
    Cn.tile_units = (uint32_t)T_units;
    Cn.n_tiles = (uint32_t)((sh->own_end - entry + T_units - 1) / T_units);
    if ((rc = d.counter.ensure(64))) return rc;
    if ((rc = d.chunk_counts.ensure((size_t)Cn.n_tiles * 4))) return rc;
    if ((rc = d.offsets.ensure((size_t)Cn.n_tiles * 8))) return rc;
    if ((rc = d.scan_tmp.ensure(((size_t)Cn.n_tiles / 2048 + 2) * 8))) return rc;
    Cn.d_len = d.lenbuf.p;
    Cn.d_len_big = S.d_len_big;
    Cn.d_state = S.d_state;
    Cn.d_out_id = d.T.term_id; // state[] holds the trie node of the longest keyword starting at a position
    Cn.len_bytes = S.len_bytes;
    Cn.own_begin = (uint32_t)sh->own_begin;
    Cn.own_end = (uint32_t)sh->own_end;
    Cn.d_blockmax = S.d_blockmax;
    Cn.entry = (uint32_t)entry;
    Cn.max_len = t.max_len;
    Cn.d_counts = (uint32_t *)d.chunk_counts.p;
    Cn.d_offsets = (const uint64_t *)d.offsets.p;
    Cn.d_out = d_out;
    Cn.cap = cap;
    Cn.record_kind = record_kind;
    Cn.d_exit = (unsigned long long *)d.counter.p;

    HIP_TRY(hipMemsetAsync(d.counter.p, 0, 64, stream));
    d.cclean[0] = false; // (match_all's first set of slot counters lives here)
    if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
    const char *kname = "";
    if (root_form) {
        LongestScanLaunch Sb = S;
        Sb.debug = (uint32_t)(tunables().tile_debug >> 32);
        Sb.lds_rows = std::min<uint32_t>(t.n_states, longest_block_max_rows(t.n_cls));
        HIP_TRY(launch_longest_block(d.T, Sb, stream, &kname));
        S.d_todo = S.d_todo_w;
        HIP_TRY(launch_longest_scan(d.T, S, stream, nullptr));
    } else {
        HIP_TRY(launch_longest_scan(d.T, S, stream, &kname));
    }
    if (timed) HIP_TRY(hipEventRecord(ev[1], stream));
    if ((rc = d.chain.ensure((size_t)Cn.n_tiles * 4 + 64))) return rc;
    uint32_t *d_sync = (uint32_t *)d.chain.p;
    // Chain: synchronisation points, a count pass that also marks the chain's matches in a bitmap, prefix sum, and the
    // records written position-parallel from the bitmap (k_longest_emit).  16-bit lengths take the count pass that reads the
    // lengths through LDS in chunks (k_longest_chain_lds).  Tunable tile_debug, for A/B: bit 65536 = the serial write pass
    // instead of bitmap + emit, bit 131072 = the count/write passes that read the lengths from global memory.
    Cn.d_bits = nullptr;
    Cn.d_ebits = nullptr;
    Cn.len_units = (uint32_t)sh->n_units;
    const bool serial_write = (tunables().tile_debug & 65536) != 0;
    const bool chain_lds = Cn.len_bytes <= 2 && !(tunables().tile_debug & 131072);
    if (!serial_write) {
        const size_t bit_bytes = ((size_t)sh->n_units / 128 + 2) * 16; // whole groups of four words (16-byte stores)
        // (a second bitmap of match ends for the emit pass; tile_debug bit 524288: without it, the emit pass looks lengths up)
        const bool end_bits = chain_lds && !(tunables().tile_debug & 524288);
        if ((rc = d.chainbits.ensure(bit_bytes * (end_bits ? 2 : 1)))) return rc;
        Cn.d_bits = (uint32_t *)d.chainbits.p;
        if (end_bits) Cn.d_ebits = Cn.d_bits + bit_bytes / 4;
        HIP_TRY(hipMemsetAsync(d.chainbits.p, 0, bit_bytes * (end_bits ? 2 : 1), stream));
    }
    HIP_TRY(launch_longest_sync(Cn, d_sync, stream));
    if (chain_lds) HIP_TRY(launch_longest_chain_lds(Cn, d_sync, /*write_pass=*/false, stream));
    else HIP_TRY(launch_longest_chain(Cn, d_sync, /*write_pass=*/false, stream));
    HIP_TRY(launch_exclusive_scan(Cn.d_counts, Cn.n_tiles, (uint64_t *)d.offsets.p, (uint64_t *)d.scan_tmp.p, stream));
    if (!serial_write) HIP_TRY(launch_longest_emit(Cn, d_sync, stream));
    else if (chain_lds) HIP_TRY(launch_longest_chain_lds(Cn, d_sync, /*write_pass=*/true, stream));
    else HIP_TRY(launch_longest_chain(Cn, d_sync, /*write_pass=*/true, stream));
    if (timed) HIP_TRY(hipEventRecord(ev[2], stream));
    // {count, 0, exit} into the call's pinned host slot (and the device result) by the pipeline's last kernel
    unsigned long long *h_slot = tk ? tk->h_count : d.h_counter, *d_slot = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&d_slot, h_slot, 0));
    HIP_TRY(launch_publish_result((const unsigned long long *)d.scan_tmp.p + scan_tiles_for(Cn.n_tiles), (const unsigned long long *)d.counter.p,
                                  d_slot, tk ? reinterpret_cast<acgpu_device_result *>(sh->d_result) : nullptr, stream));
    if (tk) {
        tk->stream = stream;
        HIP_TRY(hipEventRecord(tk->done, stream));
        tk->scanned = own_len;
        std::snprintf(tk->kname, sizeof(tk->kname), "%s", kname);
        return ACGPU_OK;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = d.h_counter[0];
    sh->chain_exit = (int64_t)d.h_counter[2];
    if (prof) {
        HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
        HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, d.ev[1], d.ev[2]));
        HIP_TRY(hipEventElapsedTime(&prof->total_ms, d.ev[0], d.ev[2]));
        prof->scan_units = own_len;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "%s", kname);
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// WHOLEWORD with a word-character table that is not fold-consistent: the reference's mixed folded/raw lookups make
// token boundaries history dependent -- whole text, one lane (k_ww_sequential).  (Fold-consistent tables: match_all.)
int match_wholeword_sequential(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                               uint64_t *n_out, hipStream_t stream, acgpu_profile *prof) {
    const uint64_t own_len = sh->own_end - sh->own_begin;
    if (prof) std::memset(prof, 0, sizeof(*prof));
    if (own_len == 0) {
        *n_out = 0;
        return ACGPU_OK;
    }
    int rc;
    if ((rc = d.counter.ensure(64))) return rc;
    HIP_TRY(hipMemsetAsync(d.counter.p, 0, 64, stream));
    d.cclean[0] = false; // (match_all's first set of slot counters lives here)
    if (!sh->text_begin || !sh->text_end || sh->own_begin != 0 || sh->own_end != sh->n_units) return ACGPU_E_UNSUPPORTED;
    if (prof) HIP_TRY(hipEventRecord(d.ev[0], stream));
    HIP_TRY(launch_ww_sequential(d.T, sh->d_hay, (uint32_t)sh->n_units, d_out, cap, record_kind,
                                 (unsigned long long *)d.counter.p, stream));
    if (prof) HIP_TRY(hipEventRecord(d.ev[1], stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter, d.counter.p, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = *d.h_counter;
    if (prof) {
        HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
        prof->total_ms = prof->scan_ms;
        prof->scan_units = own_len;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "k_ww_sequential");
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// SHORTEST-mode pipeline on one shard: the AhoCorasick pipeline into an internal buffer, then the greedy selection.
int match_shortest(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                   uint64_t *n_out, hipStream_t stream, acgpu_profile *prof) {
    const int64_t entry = sh->chain_entry > 0 ? sh->chain_entry : 0;
    sh->chain_exit = entry;
    int rc;
    uint64_t m = 0;
    // what the buffer already holds (its size includes 16 spare bytes), or a first guess
    uint64_t acap = std::max<uint64_t>(d.short_recs.bytes > 16 ? (d.short_recs.bytes - 16) / ACGPU_REC_MAP : 0,
                                       (sh->own_end - sh->own_begin) / 32 + (1 << 16));
    acgpu_profile all_prof;
    for (;;) { // all matches (end ascending, longest first) with keyword ids; retried once with the exact capacity
        if ((rc = d.short_recs.ensure(acap * ACGPU_REC_MAP + 16))) return rc;
        rc = match_all(a, d, sh, ACGPU_REC_MAP, d.short_recs.p, acap, &m, stream, prof ? &all_prof : nullptr);
        if (rc == ACGPU_E_OVERFLOW) {
            acap = m;
            continue;
        }
        if (rc != ACGPU_OK) return rc;
        break;
    }
    if (prof) *prof = all_prof;
    *n_out = 0;
    if (m == 0) return ACGPU_OK;
    if (m >= 0xfffffff0ull) return ACGPU_E_UNSUPPORTED;
    const uint32_t M = (uint32_t)m;
    if ((rc = d.short_nxt.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_tmp.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_mark.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.offsets.ensure((size_t)M * 8))) return rc;
    if ((rc = d.scan_tmp.ensure(((size_t)M / 2048 + 2) * 8))) return rc;
    if ((rc = d.counter.ensure(64))) return rc;
    if (prof) HIP_TRY(hipEventRecord(d.ev[0], stream));
    HIP_TRY(launch_shortest_select((const int32_t *)d.short_recs.p, M, entry, (uint32_t *)d.short_nxt.p,
                                   (uint32_t *)d.short_tmp.p, (uint32_t *)d.short_mark.p, stream));
    if ((rc = mark_chain(d, (uint32_t *)d.short_nxt.p, (uint32_t *)d.short_tmp.p, (uint32_t *)d.short_mark.p, M, stream, nullptr,
                         nullptr, 0)))
        return rc;
    HIP_TRY(launch_exclusive_scan((const uint32_t *)d.short_mark.p, M, (uint64_t *)d.offsets.p, (uint64_t *)d.scan_tmp.p, stream));
    const uint64_t *d_total = (const uint64_t *)d.scan_tmp.p + scan_tiles_for(M);
    HIP_TRY(launch_shortest_emit((const int32_t *)d.short_recs.p, M, (const uint32_t *)d.short_mark.p,
                                 (const uint64_t *)d.offsets.p, d_total, record_kind, d_out, cap, entry,
                                 (unsigned long long *)d.counter.p, stream));
    d.cclean[0] = false; // (the exit position went where match_all's first set of slot counters lives)
    if (prof) HIP_TRY(hipEventRecord(d.ev[1], stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter, d_total, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter + 1, d.counter.p, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = d.h_counter[0];
    sh->chain_exit = (int64_t)d.h_counter[1];
    if (prof) {
        float sel_ms = 0;
        HIP_TRY(hipEventElapsedTime(&sel_ms, d.ev[0], d.ev[1]));
        prof->finalize_ms += sel_ms; // ordering of the all-matches list + the selection
        prof->total_ms += sel_ms;
        prof->n_matches = *n_out;
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// WholeWordLongestMatchSet.match(String) with a word-character table that is not fold-consistent: the reference mixes folded
// and raw lookups (S/WholeWordLongestMatchSet.java:126 against :151,:156), which makes token boundaries history dependent --
// whole text, one lane (k_wwl_sequential).
int match_wwlongest_sequential(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                               uint64_t *n_out, hipStream_t stream, acgpu_profile *prof) {
    const HostTables &t = a->t;
    if (prof) std::memset(prof, 0, sizeof(*prof));
    *n_out = 0;
    const uint32_t n = (uint32_t)sh->n_units;
    sh->chain_exit = (int64_t)std::max<int64_t>(sh->chain_entry, (int64_t)sh->own_begin);
    int rc;
    if (!sh->text_begin || !sh->text_end || sh->own_begin != 0 || sh->own_end != sh->n_units) return ACGPU_E_UNSUPPORTED;
    if (n == 0 || t.n_states <= 1) return ACGPU_OK;
    if ((rc = d.counter.ensure(64))) return rc;
    d.cclean[0] = false; // (match_all's first set of slot counters lives here)
    if (prof) HIP_TRY(hipEventRecord(d.ev[0], stream));
    HIP_TRY(launch_wwl_sequential(d.T, sh->d_hay, n, d_out, cap, record_kind, (unsigned long long *)d.counter.p, stream));
    if (prof) HIP_TRY(hipEventRecord(d.ev[1], stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter, d.counter.p, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = *d.h_counter;
    sh->chain_exit = (int64_t)n;
    if (prof) {
        HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
        prof->total_ms = prof->scan_ms;
        prof->scan_units = n;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "k_wwl_sequential");
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

// WWLONGEST-mode pipeline on one shard.  A walk belongs to the shard that owns its first unit; the scan visits the first walk
// start at or after chain_entry and leaves chain_exit = the position behind the stop of its last visited walk.
// T: the device tables the scan sees (d.T, or folded_tables(d) for the loops that fold in every lookup).
// plain_words: the walk reports only a whole path that is a keyword and ends at a word boundary -- no carried fail match --
// and does without the first-word table: WholeWordMatchMap's loop (S/WholeWordMatchMap.java:55-153), which is this walk
// without fail matches, over a WHOLEWORD automaton whose folded keywords hold non-word units (HostTables::fold_clean).
int match_wwlongest(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                    uint64_t *n_out, hipStream_t stream, acgpu_profile *prof, const DevTables &T, bool plain_words) {
    const HostTables &t = a->t;
    if (prof) std::memset(prof, 0, sizeof(*prof));
    *n_out = 0;
    const uint32_t n = (uint32_t)sh->n_units;
    const uint64_t entry = (uint64_t)std::max<int64_t>(sh->chain_entry, (int64_t)sh->own_begin);
    sh->chain_exit = (int64_t)entry;
    int rc;
    if (!sh->text_begin && sh->own_begin < 1) return ACGPU_E_INVALID;                              // left context: 1 unit
    if (!sh->text_end && sh->n_units - sh->own_end < (uint64_t)t.max_len + 1) return ACGPU_E_INVALID; // right halo
    if (n == 0 || t.n_states <= 1 || sh->own_end == sh->own_begin || entry >= sh->own_end) return ACGPU_OK;
    const uint32_t n_tiles = wwl_tiles(n);
    if ((rc = d.chunk_counts.ensure((size_t)n_tiles * 4))) return rc;
    if ((rc = d.offsets.ensure((size_t)n_tiles * 8))) return rc;
    if ((rc = d.scan_tmp.ensure(((size_t)n_tiles / 2048 + 2) * 8))) return rc;
    if ((rc = d.counter.ensure(64))) return rc;
    d.cclean[0] = false; // (match_all's first set of slot counters lives here)
    if (prof) HIP_TRY(hipEventRecord(d.ev[0], stream));
    HIP_TRY(launch_wwl_starts(T, sh->d_hay, n, d.n_cu, false, (uint32_t *)d.chunk_counts.p, nullptr, nullptr, sh->text_begin, d.start_behind, stream));
    HIP_TRY(launch_exclusive_scan((const uint32_t *)d.chunk_counts.p, n_tiles, (uint64_t *)d.offsets.p, (uint64_t *)d.scan_tmp.p,
                                  stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter, (const uint64_t *)d.scan_tmp.p + scan_tiles_for(n_tiles), 8, hipMemcpyDeviceToHost,
                           stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const uint32_t M = (uint32_t)*d.h_counter; // walk starts of the buffer (halos included)
    if (M == 0) return ACGPU_OK;
    if ((rc = d.wwl_rs.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.wwl_mend.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.wwl_mid.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.wwl_sel.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.wwl_stop.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_nxt.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_tmp.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = d.short_mark.ensure(((size_t)M + 1) * 4))) return rc;
    { // the exit position, preset to the entry (no visited walk start: nothing changes hands)
        d.h_counter[2] = entry;
        HIP_TRY(hipMemcpyAsync(d.counter.p, d.h_counter + 2, 8, hipMemcpyHostToDevice, stream));
    }
    HIP_TRY(launch_wwl_starts(T, sh->d_hay, n, d.n_cu, true, nullptr, (const uint64_t *)d.offsets.p, (uint32_t *)d.wwl_rs.p,
                              sh->text_begin, d.start_behind, stream));
    HIP_TRY(launch_wwl_walk(T, plain_words, sh->d_hay, n, (const uint32_t *)d.wwl_rs.p, M, (uint32_t *)d.short_nxt.p,
                            (uint32_t *)d.short_mark.p, (int32_t *)d.wwl_mend.p, (int32_t *)d.wwl_mid.p, (uint32_t *)d.wwl_stop.p,
                            (uint32_t)entry, d.n_cu, stream));
    if (prof) HIP_TRY(hipEventRecord(d.ev[1], stream));
    // Which starts does the scan visit?  The chain k0, NXT[k0], ... over the start indices: mark_chain (one pass -- a walk
    // runs over few later starts; pointer doubling took 2.5 ms of 8.4 on config 5's text).  The select pass needs the
    // successors afterwards.
    const uint32_t *nxt_for_select = nullptr;
    if ((rc = d.wwl_nxt0.ensure(((size_t)M + 1) * 4))) return rc;
    if ((rc = mark_chain(d, (uint32_t *)d.short_nxt.p, (uint32_t *)d.short_tmp.p, (uint32_t *)d.short_mark.p, M, stream,
                         (uint32_t *)d.wwl_nxt0.p, &nxt_for_select, t.max_len / 2 + 2)))
        return rc;
    HIP_TRY(launch_wwl_select((const uint32_t *)d.short_mark.p, (const int32_t *)d.wwl_mend.p, (const uint32_t *)d.wwl_rs.p,
                              nxt_for_select, (const uint32_t *)d.wwl_stop.p, (uint32_t *)d.wwl_sel.p, M,
                              (uint32_t)sh->own_begin, (uint32_t)sh->own_end, (unsigned long long *)d.counter.p, stream));
    if (scan_tile_elems() != 2048) return ACGPU_E_INVALID; // (k_wwl_emit ranks one prefix-sum tile per workgroup)
    if ((rc = d.scan_tmp.ensure(((size_t)M / 2048 + 2) * 8))) return rc;
    HIP_TRY(launch_scan_tile_offsets((const uint32_t *)d.wwl_sel.p, M, (uint64_t *)d.scan_tmp.p, stream));
    HIP_TRY(launch_wwl_emit((const uint32_t *)d.wwl_rs.p, (const uint32_t *)d.wwl_sel.p, (const int32_t *)d.wwl_mend.p,
                            (const int32_t *)d.wwl_mid.p, (const uint64_t *)d.scan_tmp.p, M, record_kind, d_out, cap, stream));
    if (prof) HIP_TRY(hipEventRecord(d.ev[2], stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter, (const uint64_t *)d.scan_tmp.p + scan_tiles_for(M), 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(d.h_counter + 1, d.counter.p, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *n_out = *d.h_counter;
    sh->chain_exit = (int64_t)d.h_counter[1];
    if (prof) {
        HIP_TRY(hipEventElapsedTime(&prof->scan_ms, d.ev[0], d.ev[1]));
        HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, d.ev[1], d.ev[2]));
        HIP_TRY(hipEventElapsedTime(&prof->total_ms, d.ev[0], d.ev[2]));
        prof->scan_units = n;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "k_wwl_walk");
    }
    return *n_out > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

} // namespace

namespace acgpu {

// what the tunables decide per call: the LDS residency of the DFA rows, and the CUs the scan kernels size their grids for --
// "reserve_cus" leaves that many CUs without a scan workgroup (the scans hold a whole CU's LDS per workgroup, so a grid of
// n_cu - k workgroups keeps k CUs free: room for the workgroups of a collective that runs under the scan)
void refresh_call_state(acgpu_automaton *a, DeviceState &d) {
    d.T.lds_entries = lds_states_for(a->t) * a->t.n_cls;
    const int64_t k = std::max<int64_t>(0, tunables().reserve_cus);
    d.n_cu = (int)std::max<int64_t>(8, (int64_t)d.n_cu_phys - k);
}

int device_for_call(acgpu_automaton *a, DeviceState **d, int lane) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return ACGPU_E_NODEVICE;
    }
    std::lock_guard<std::mutex> lock(a->mu); // (the map; a new pool uploads its tables under it)
    return ensure_device(a, d, lane);
}

// validates a shard and runs the pipeline of the automaton's family; caller holds d.mu.
// readable: the call stands for match(Readable, ...) (acgpu_stream_feed) -- the word matchers' Readable loops fold in every
// lookup where their String loops mix folded and raw ones, which only matters for tables that are not fold-consistent.
int match_shard(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                uint64_t *n_out, hipStream_t stream, acgpu_profile *prof, bool readable) {
    refresh_call_state(a, d);
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    if (sh->n_units >= (1ull << 31)) return ACGPU_E_INVALID;
    if (sh->own_begin > sh->own_end || sh->own_end > sh->n_units) return ACGPU_E_INVALID;
    if (sh->n_units && (!sh->d_hay || ((uintptr_t)sh->d_hay & 15))) return ACGPU_E_INVALID;
    if (cap && (!d_out || ((uintptr_t)d_out & 3))) return ACGPU_E_INVALID;
    if (sh->d_result && ((uintptr_t)sh->d_result & 15)) return ACGPU_E_INVALID;
    if (d.inflight > 0 && stream != d.inflight_stream) return ACGPU_E_INVALID; // stream rule (include/acgpu.h)
    *n_out = 0;
    const HostTables &t = a->t;
    if (t.mode == ACGPU_MODE_ALL || (t.mode == ACGPU_MODE_WHOLEWORD && t.fold_consistent))
        return match_all(a, d, sh, record_kind, d_out, cap, n_out, stream, prof);
    if (t.mode == ACGPU_MODE_WHOLEWORD && readable && t.fold_clean) { // the Readable loop: an ordinary scan over w' = word o lower
        const DevTables Tf = folded_tables(d);
        return match_all(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, nullptr, false, &Tf);
    }
    // the other families end with their count on the host (and some run the ALL pipeline inside): the device copy of the
    // result is written behind the pipeline
    acgpu_device_result *d_res = reinterpret_cast<acgpu_device_result *>(sh->d_result);
    sh->d_result = nullptr;
    int rc;
    switch (t.mode) {
    case ACGPU_MODE_LONGEST: rc = match_longest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof); break;
    case ACGPU_MODE_WHOLEWORD:
        if (readable) { // folded keywords with non-word units: the WholeWordLongest walk without fail matches, unit by unit
            DevTables Tf = folded_tables(d);
            Tf.ww_fat = nullptr;
            rc = match_wwlongest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, Tf, true);
        } else {
            rc = match_wholeword_sequential(a, d, sh, record_kind, d_out, cap, n_out, stream, prof);
        }
        break;
    case ACGPU_MODE_SHORTEST: rc = match_shortest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof); break;
    case ACGPU_MODE_WWLONGEST:
        // not fold-consistent: the Map class's String loop and both Readable loops fold in every lookup
        // (S/WholeWordLongestMatchMap.java:252-288, :404) -- position parallel over w'; the Set class's String loop mixes
        // raw and folded lookups (S/WholeWordLongestMatchSet.java:126,151,156) -- sequential
        if (t.fold_consistent) rc = match_wwlongest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, d.T, false);
        else if (readable || record_kind == ACGPU_REC_MAP)
            rc = match_wwlongest(a, d, sh, record_kind, d_out, cap, n_out, stream, prof, folded_tables(d), false);
        else rc = match_wwlongest_sequential(a, d, sh, record_kind, d_out, cap, n_out, stream, prof);
        break;
    default: rc = ACGPU_E_UNSUPPORTED;
    }
    sh->d_result = d_res;
    if (d_res && (rc == ACGPU_OK || rc == ACGPU_E_OVERFLOW)) HIP_TRY(launch_write_result(d_res, *n_out, stream));
    return rc;
}

// acgpu_match_device_begin on a given scratch pool (the caller holds dd.mu and has made dd's device current)
int begin_shard(acgpu_automaton *a, DeviceState &dd, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap, hipStream_t stream,
                int want_profile, acgpu_ticket **ticket) {
    *ticket = nullptr;
    const HostTables &t = a->t;
    DeviceState *d = &dd;
    refresh_call_state(a, dd);
    int rc;
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    if (sh->n_units >= (1ull << 31) || sh->own_begin > sh->own_end || sh->own_end > sh->n_units) return ACGPU_E_INVALID;
    if (sh->n_units && (!sh->d_hay || ((uintptr_t)sh->d_hay & 15))) return ACGPU_E_INVALID;
    if (cap && (!d_out || ((uintptr_t)d_out & 3))) return ACGPU_E_INVALID;
    if (sh->d_result && ((uintptr_t)sh->d_result & 15)) return ACGPU_E_INVALID;
    if (d->inflight > 0 && stream != d->inflight_stream) return ACGPU_E_INVALID; // stream rule (include/acgpu.h)
    Ticket *tk = nullptr;
    for (auto &cand : d->tickets)
        if (!cand.busy) { tk = &cand; break; }
    if (!tk) return ACGPU_E_INVALID; // too many calls in flight: collect one first
    tk->profiled = want_profile != 0;
    tk->done_is_ev2 = false;
    tk->one_kernel = false;
    tk->cap = cap;
    tk->user_shard = sh;
    tk->kname[0] = 0;
    uint64_t dummy = 0;
    // enqueued without waiting: the AhoCorasick / WholeWord pipeline (one scan + ordering pass) and the LongestMatch walk
    // pipeline (lengths, synchronisation points, chain count, prefix sum, emit: nothing of it needs the host).  The other
    // families -- and LongestMatch over a dictionary with a selective suffix filter, whose sparse form falls back to the walk
    // after looking at the match count -- run their call inside _begin: the ticket is complete when _begin returns.
    if (t.mode == ACGPU_MODE_ALL || (t.mode == ACGPU_MODE_WHOLEWORD && t.fold_consistent)) {
        tk->kind = 0;
        rc = match_all(a, *d, sh, record_kind, d_out, cap, &dummy, stream, nullptr, tk);
    } else if (t.mode == ACGPU_MODE_LONGEST && !(filter_is_selective(t) && tunables().force_kernel != 1)) {
        tk->kind = 1;
        rc = match_longest(a, *d, sh, record_kind, d_out, cap, &dummy, stream, nullptr, tk);
    } else {
        tk->kind = 2;
        tk->sync_n = 0;
        std::memset(&tk->sync_prof, 0, sizeof(tk->sync_prof));
        tk->sync_rc = match_shard(a, *d, sh, record_kind, d_out, cap, &tk->sync_n, stream, want_profile ? &tk->sync_prof : nullptr);
        if (tk->sync_rc != ACGPU_OK && tk->sync_rc != ACGPU_E_OVERFLOW) return tk->sync_rc;
        tk->busy = true; // (complete: nothing in flight on the device, the stream rule does not apply to it)
        *ticket = reinterpret_cast<acgpu_ticket *>(tk);
        return ACGPU_OK;
    }
    if (rc != ACGPU_OK) return rc;
    tk->busy = true;
    d->inflight++;
    d->inflight_stream = stream;
    *ticket = reinterpret_cast<acgpu_ticket *>(tk);
    return ACGPU_OK;
}


} // namespace acgpu

extern "C" {

uint32_t acgpu_abi_version(void) { return ACGPU_ABI_VERSION; }

int acgpu_last_hip_error(void) { return g_last_hip_error; }

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

int64_t acgpu_set_tunable(const char *name, int64_t value) {
    if (!name) return -1;
    if (!std::strcmp(name, "ablation_build")) { // (read only) 1: built with -DACGPU_ABLATION, the ablation bits of tile_debug work
#ifdef ACGPU_ABLATION
        return 1;
#else
        return 0;
#endif
    }
    Tunables &t = tunables();
    std::atomic<int64_t> *slot = nullptr;
    if (!std::strcmp(name, "chunk_units")) slot = &t.chunk_units;
    else if (!std::strcmp(name, "blocks_per_cu")) slot = &t.blocks_per_cu;
    else if (!std::strcmp(name, "lds_table_bytes")) slot = &t.lds_table_bytes;
    else if (!std::strcmp(name, "force_sparse")) slot = &t.force_sparse;
    else if (!std::strcmp(name, "dense_budget_bytes")) slot = &t.dense_budget_bytes;
    else if (!std::strcmp(name, "force_kernel")) slot = &t.force_kernel;
    else if (!std::strcmp(name, "region_units")) slot = &t.region_units;
    else if (!std::strcmp(name, "tile_form")) slot = &t.tile_form;
    else if (!std::strcmp(name, "tile_debug")) slot = &t.tile_debug;
    else if (!std::strcmp(name, "ww_first_seed")) slot = &t.ww_first_seed;
    else if (!std::strcmp(name, "ww_no_bloom")) slot = &t.ww_no_bloom;
    else if (!std::strcmp(name, "ww_no_ph")) slot = &t.ww_no_ph;
    else if (!std::strcmp(name, "ww_no_byte_pages")) slot = &t.ww_no_byte_pages;
    else if (!std::strcmp(name, "ww_ramp_pm")) slot = &t.ww_ramp_pm;
    else if (!std::strcmp(name, "ww_block")) slot = &t.ww_block;
    else if (!std::strcmp(name, "ww_ph_lambda")) slot = &t.ww_ph_lambda;
    else if (!std::strcmp(name, "rdense_budget_bytes")) slot = &t.rdense_budget_bytes;
    else if (!std::strcmp(name, "filter_max_bytes")) slot = &t.filter_max_bytes;
    else if (!std::strcmp(name, "no_merged_ranges")) slot = &t.no_merged_ranges;
    else if (!std::strcmp(name, "no_short_keywords")) slot = &t.no_short_keywords;
    else if (!std::strcmp(name, "reserve_cus")) slot = &t.reserve_cus;
    else if (!std::strcmp(name, "no_bits_trie")) slot = &t.no_bits_trie;
    else if (!std::strcmp(name, "all_form")) slot = &t.all_form;
    else if (!std::strcmp(name, "no_state_form")) slot = &t.no_state_form;
    else if (!std::strcmp(name, "longest_form")) slot = &t.longest_form;
    else if (!std::strcmp(name, "multi_min_share")) slot = &t.multi_min_share;
    else if (!std::strcmp(name, "no_big_l2")) slot = &t.no_big_l2;
    else if (!std::strcmp(name, "no_class_pages")) slot = &t.no_class_pages;
    else if (!std::strcmp(name, "split_cand_div")) slot = &t.split_cand_div;
    if (!slot) return -1;
    return slot->exchange(value, std::memory_order_relaxed);
}

int acgpu_build(int mode, const uint16_t *kw_units, const uint64_t *kw_off, uint32_t n_kw, int case_sensitive,
                const uint16_t *lower_tbl, const uint8_t *wordchar_tbl, acgpu_automaton **out, int64_t *bad_keyword) {
    if (!out) return ACGPU_E_INVALID;
    *out = nullptr;
    acgpu_automaton *a = new (std::nothrow) acgpu_automaton();
    if (!a) return ACGPU_E_NOMEM;
    int rc;
    try {
        rc = build_tables(mode, kw_units, kw_off, n_kw, case_sensitive, lower_tbl, wordchar_tbl, a->t, bad_keyword);
    } catch (const std::bad_alloc &) {
        rc = ACGPU_E_NOMEM;
    } catch (...) {
        rc = ACGPU_E_INVALID;
    }
    if (rc != ACGPU_OK) {
        delete a;
        return rc;
    }
    *out = a;
    return ACGPU_OK;
}

void acgpu_stream_detach(acgpu_stream *s); // acgpu_stream.hip

void acgpu_free(acgpu_automaton *a) {
    if (!a) return;
    { // streams still open on it (include/acgpu.h asks for them to be closed first): detached -- their feeds return
      // ACGPU_E_INVALID from now on and acgpu_stream_close touches nothing of the automaton
        std::lock_guard<std::mutex> l(a->mu);
        for (acgpu_stream *s : a->open_streams) acgpu_stream_detach(s);
        a->open_streams.clear();
    }
    int cur = -1;
    bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto &kv : a->dev) {
        if (have) (void)hipSetDevice(kv.first.first);
        kv.second.reset();
    }
    for (auto &b : a->stream_cache) {
        if (have) (void)hipSetDevice(b.device);
        b.release();
    }
    if (have) (void)hipSetDevice(cur);
    delete a;
}

int acgpu_get_info(const acgpu_automaton *a, acgpu_info *info) {
    if (!a || !info) return ACGPU_E_INVALID;
    const HostTables &t = a->t;
    std::memset(info, 0, sizeof(*info));
    info->abi_version = ACGPU_ABI_VERSION;
    info->mode = (uint32_t)t.mode;
    info->case_sensitive = t.cs;
    info->n_states = t.n_states;
    info->n_classes = t.n_cls;
    info->n_keywords = t.n_kw;
    info->min_keyword_len = t.min_len;
    info->max_keyword_len = t.max_len;
    info->dense = t.dense;
    info->entry_bytes = t.entry_bytes;
    info->table_bytes = t.dense ? (uint64_t)t.dfa.size() * t.entry_bytes : (uint64_t)t.hkeys.size() * 12 + (uint64_t)t.n_states * 4;
    info->lds_states = lds_states_for(t);
    info->fold_consistent = t.fold_consistent;
    info->filter_k = t.filt_k;
    info->filter_bits = (uint32_t)(t.filt_bits.size() * 32);
    info->tile_kernel = t.mode == ACGPU_MODE_LONGEST ? (filter_is_selective(t) && tunables().force_kernel != 1) : use_tile_kernel(t);
    info->filter_density = (float)t.filt_density;
    info->fold_clean = t.fold_clean;
    return ACGPU_OK;
}

int acgpu_debug_tables(const acgpu_automaton *a, uint16_t *cls_lut, uint32_t *dfa, uint32_t *out_len, uint32_t *out_link,
                       uint32_t *out_id, uint32_t *depth, uint32_t *first_out_state) {
    if (!a) return ACGPU_E_INVALID;
    const HostTables &t = a->t;
    if (cls_lut) std::memcpy(cls_lut, t.cls_lut.data(), 65536 * sizeof(uint16_t));
    if (dfa) {
        if (!t.dense) return ACGPU_E_UNSUPPORTED;
        std::memcpy(dfa, t.dfa.data(), t.dfa.size() * sizeof(uint32_t));
    }
    if (out_len) std::memcpy(out_len, t.out_len.data(), t.n_states * sizeof(uint32_t));
    if (out_link) std::memcpy(out_link, t.out_link.data(), t.n_states * sizeof(uint32_t));
    if (out_id) std::memcpy(out_id, t.out_id.data(), t.n_states * sizeof(uint32_t));
    if (depth) std::memcpy(depth, t.depth.data(), t.n_states * sizeof(uint32_t));
    if (first_out_state) *first_out_state = t.first_out;
    return ACGPU_OK;
}

int acgpu_debug_states(const acgpu_automaton *a, uint64_t sizes[6], uint32_t *rows, uint32_t *nodes, uint32_t *mask, uint32_t *out,
                       uint32_t *ids) {
    if (!a || !sizes) return ACGPU_E_INVALID;
    const HostTables &t = a->t;
    const bool have = t.hy_n_states != 0 && (t.mode == ACGPU_MODE_ALL || t.mode == ACGPU_MODE_SHORTEST);
    sizes[0] = have ? t.hy_n_states : 0;
    sizes[1] = have ? t.hy_n_dense : 0;
    sizes[2] = t.n_cls;
    sizes[3] = have ? t.hy_dense.size() : 0;
    sizes[4] = have ? t.hy_nodes.size() : 0;
    sizes[5] = have ? t.hy_ids.size() : 0;
    if (!have) return ACGPU_OK;
    if (rows) std::memcpy(rows, t.hy_dense.data(), t.hy_dense.size() * sizeof(uint32_t));
    if (nodes) std::memcpy(nodes, t.hy_nodes.data(), t.hy_nodes.size() * sizeof(uint32_t));
    if (mask) std::memcpy(mask, t.hy_mask.data(), t.hy_mask.size() * sizeof(uint32_t));
    if (out) std::memcpy(out, t.hy_out.data(), t.hy_out.size() * sizeof(uint32_t));
    if (ids) std::memcpy(ids, t.hy_ids.data(), t.hy_ids.size() * sizeof(uint32_t));
    return ACGPU_OK;
}

int acgpu_debug_wordhash_perfect(const acgpu_automaton *a, uint32_t sizes[3], uint32_t *slots, uint16_t *disp, uint8_t *bp_idx,
                                 uint8_t *bp_pages, uint16_t *bp_delta) {
    if (!a || !sizes) return ACGPU_E_INVALID;
    const HostTables &t = a->t;
    if (t.mode != ACGPU_MODE_WHOLEWORD) return ACGPU_E_UNSUPPORTED;
    sizes[0] = t.ww_ph_n;
    sizes[1] = t.ww_ph_buckets;
    sizes[2] = t.ww_bp_n;
    if (slots && !t.ww_ph.empty()) std::memcpy(slots, t.ww_ph.data(), t.ww_ph.size() * sizeof(uint32_t));
    if (disp && !t.ww_ph_disp.empty()) std::memcpy(disp, t.ww_ph_disp.data(), (size_t)t.ww_ph_buckets * sizeof(uint16_t));
    if (bp_idx && t.ww_bp_n) std::memcpy(bp_idx, t.ww_bp_idx.data(), 256);
    if (bp_pages && t.ww_bp_n) std::memcpy(bp_pages, t.ww_bp_pages.data(), t.ww_bp_pages.size());
    if (bp_delta && t.ww_bp_n) std::memcpy(bp_delta, t.ww_bp_delta.data(), 128 * sizeof(uint16_t));
    return ACGPU_OK;
}

int acgpu_debug_wordhash(const acgpu_automaton *a, uint32_t *n_slots, uint32_t *slots, uint64_t *n_rec_words, uint32_t *recs,
                         uint8_t *fold_pgidx, uint32_t *n_pages, uint16_t *fold_pages, uint32_t *seed) {
    if (!a) return ACGPU_E_INVALID;
    const HostTables &t = a->t;
    if (t.mode != ACGPU_MODE_WHOLEWORD) return ACGPU_E_UNSUPPORTED;
    if (n_slots) *n_slots = (uint32_t)(t.ww_fat.size() / 8);
    if (slots) std::memcpy(slots, t.ww_fat.data(), t.ww_fat.size() * sizeof(uint32_t));
    if (n_rec_words) *n_rec_words = t.ww_recs.size();
    if (recs) std::memcpy(recs, t.ww_recs.data(), t.ww_recs.size() * sizeof(uint32_t));
    if (fold_pgidx) std::memcpy(fold_pgidx, t.fold_pgidx.data(), 256);
    if (n_pages) *n_pages = t.fold_n_pages;
    if (fold_pages) std::memcpy(fold_pages, t.fold_pages.data(), t.fold_pages.size() * sizeof(uint16_t));
    if (seed) *seed = t.ww_seed;
    return ACGPU_OK;
}

int acgpu_match_device(const acgpu_automaton *ca, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                       uint64_t *n_out, void *stream_, acgpu_profile *prof) {
    if (!ca || !sh || !n_out) return ACGPU_E_INVALID;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    DeviceState *d = nullptr;
    int rc = device_for_call(a, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(d->mu);
    return match_shard(a, *d, sh, record_kind, d_out, cap, n_out, reinterpret_cast<hipStream_t>(stream_), prof);
}

int acgpu_match_device_begin(const acgpu_automaton *ca, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap,
                             void *stream_, int want_profile, acgpu_ticket **ticket) {
    if (!ca || !sh || !ticket) return ACGPU_E_INVALID;
    *ticket = nullptr;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    DeviceState *d = nullptr;
    int rc = device_for_call(a, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(d->mu);
    return begin_shard(a, *d, sh, record_kind, d_out, cap, reinterpret_cast<hipStream_t>(stream_), want_profile, ticket);
}

int acgpu_match_device_abandon(const acgpu_automaton *ca, acgpu_ticket *ticket) {
    if (!ca || !ticket) return ACGPU_E_INVALID;
    Ticket *tk = reinterpret_cast<Ticket *>(ticket);
    DeviceState *own = reinterpret_cast<DeviceState *>(tk->owner); // (set when the pool was created)
    hipEvent_t done = nullptr;
    {
        std::lock_guard<std::mutex> lock(own->mu);
        if (!tk->busy) return ACGPU_E_INVALID;
        if (tk->kind == 2) {
            tk->busy = false;
            return ACGPU_OK;
        }
        done = tk->done_is_ev2 ? tk->ev[2] : tk->done;
    }
    HIP_TRY(hipEventSynchronize(done)); // (its kernels still write the caller's buffers until then)
    std::lock_guard<std::mutex> lock(own->mu);
    if (!tk->busy) return ACGPU_E_INVALID;
    tk->busy = false;
    reinterpret_cast<DeviceState *>(tk->owner)->inflight--;
    return ACGPU_OK;
}

int acgpu_match_device_end(const acgpu_automaton *ca, acgpu_ticket *ticket, uint64_t *n_out, acgpu_profile *prof) {
    return end_ticket(ca, ticket, n_out, prof, nullptr);
}

} // extern "C"

namespace acgpu {

int end_ticket(const acgpu_automaton *ca, acgpu_ticket *ticket, uint64_t *n_out, acgpu_profile *prof, bool *redone) {
    if (redone) *redone = false;
    if (!ca || !ticket || !n_out) return ACGPU_E_INVALID;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    Ticket *tk = reinterpret_cast<Ticket *>(ticket);
    DeviceState *own = reinterpret_cast<DeviceState *>(tk->owner); // (set when the pool was created)
    hipEvent_t done = nullptr;
    {
        std::lock_guard<std::mutex> lock(own->mu);
        if (!tk->busy) return ACGPU_E_INVALID;
        if (tk->kind == 2) { // ran inside _begin
            tk->busy = false;
            *n_out = tk->sync_n;
            if (prof) *prof = tk->sync_prof;
            return tk->sync_rc;
        }
        done = tk->done_is_ev2 ? tk->ev[2] : tk->done;
    }
    HIP_TRY(hipEventSynchronize(done)); // outside the lock: other calls may be enqueued meanwhile
    std::lock_guard<std::mutex> lock(own->mu);
    if (!tk->busy) return ACGPU_E_INVALID; // (collected by another thread meanwhile)
    DeviceState *d = reinterpret_cast<DeviceState *>(tk->owner);
    if (tk->kind == 0 && (uint32_t)tk->h_count[1] != 0) { // a candidate slice / scratch slice was too small: redo with the fused kernel, one slice
        // (the redo shares the scratch with the tickets still in flight: same stream, so stream order keeps them apart)
        const int rc = match_all(a, *d, &tk->shard, tk->record_kind, tk->d_out, tk->cap, n_out, tk->stream, prof, nullptr, true);
        tk->busy = false;
        d->inflight--;
        if (redone) *redone = true;
        return rc;
    }
    if (tk->kind == 1 && tk->h_count[1] != 0) { // k_longest_bits / k_longest_follow bailed out (a unit outside the alphabet, a chain that did not merge)
        if (!std::strcmp(tk->kname, "k_longest_follow")) d->fol_level = std::max(d->fol_level, tk->bits_level + 1);
        const int rc = match_longest(a, *d, &tk->shard, tk->record_kind, tk->d_out, tk->cap, n_out, tk->stream, prof, nullptr,
                                     tk->h_count[1] == 1 ? tk->bits_level + 1 : 2);
        if (tk->user_shard) tk->user_shard->chain_exit = tk->shard.chain_exit;
        tk->busy = false;
        d->inflight--;
        if (redone) *redone = true;
        return rc;
    }
    *n_out = *tk->h_count;
    if (tk->kind == 1 && tk->user_shard) tk->user_shard->chain_exit = (int64_t)tk->h_count[2];
    if (tk->kind == 0 && a->t.mode != ACGPU_MODE_WHOLEWORD && tk->shard.own_end > tk->shard.own_begin)
        d->all_density = (double)*n_out / (double)(tk->shard.own_end - tk->shard.own_begin);
    tk->busy = false; // (whatever happens below, the ticket is collected)
    d->inflight--;
    if (prof) {
        std::memset(prof, 0, sizeof(*prof));
        if (tk->profiled && tk->one_kernel) {
            HIP_TRY(hipEventElapsedTime(&prof->scan_ms, tk->ev[0], tk->ev[2]));
            prof->total_ms = prof->scan_ms;
        } else if (tk->profiled) {
            HIP_TRY(hipEventElapsedTime(&prof->scan_ms, tk->ev[0], tk->ev[1]));
            HIP_TRY(hipEventElapsedTime(&prof->finalize_ms, tk->ev[1], tk->ev[2]));
            HIP_TRY(hipEventElapsedTime(&prof->total_ms, tk->ev[0], tk->ev[2]));
        }
        prof->scan_units = tk->scanned;
        prof->n_matches = *n_out;
        std::snprintf(prof->scan_kernel, sizeof(prof->scan_kernel), "%s", tk->kname);
    }
    return *n_out > tk->cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

} // namespace acgpu

extern "C" {

} // extern "C"

namespace acgpu {

// acgpu_match_u16 on a long haystack, pipelined: the text goes to the device in chunks -- worker threads copy the caller's
// (pageable) memory into a ring of pinned staging buffers and enqueue the DMA on a copy stream -- while the chunks that have
// arrived are scanned as shards of the device buffer (own range = the chunk, halos = its neighbours already / also there;
// the chain families hand their entry on from shard to shard).  The scans are a fraction of the transfer time, so the call
// runs at the rate of the slower of the host copy and the link instead of copy + scan + copy back in sequence.
// The general form serves one device's share of a multi-device call as well: the device buffer holds the units [lo, hi) of
// the text (the share plus its halos), of which [own_lo, own_hi) is owned; positions in the records and in *chain are
// relative to the buffer.
constexpr uint64_t kHostChunkUnits = 1ull << 24; // 32 MiB per chunk

// The CPUs of the NUMA node a device hangs on (/sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist):
// the threads that copy a share's text into pinned memory run there, so that on a two-socket host eight devices are fed by both
// sockets' memory controllers, each from its own side.  false: unknown (node -1, a container without /sys, ...): no affinity is set.
static bool device_numa_cpus(int dev, cpu_set_t *set) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) != hipSuccess) return false;
    for (char *p = bdf; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
    char path[160];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = std::fscanf(f, "%d", &node);
    std::fclose(f);
    if (got != 1 || node < 0) return false;
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = std::fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool have = std::fgets(list, (int)sizeof(list), f) != nullptr;
    std::fclose(f);
    if (!have) return false;
    CPU_ZERO(set);
    int n_set = 0;
    for (char *p = list; *p;) { // "0-63,128-191"
        char *end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            CPU_SET((int)c, set);
            ++n_set;
        }
        if (*p == ',') ++p;
    }
    return n_set > 0;
}

int scan_host_range(acgpu_automaton *a, DeviceState &d, const uint16_t *haystack, uint64_t n_units, uint64_t lo, uint64_t hi,
                    uint64_t own_lo, uint64_t own_hi, int record_kind, uint64_t cap, uint64_t *n_out, int64_t *chain_io) {
    const HostTables &t = a->t;
    const uint64_t C = kHostChunkUnits;
    const uint64_t nb = hi - lo; // units in the device buffer
    const uint32_t n_chunks = (uint32_t)((nb + C - 1) / C);
    const uint64_t ob = own_lo - lo, oe = own_hi - lo; // the owned range in the buffer
    hipStream_t stream = d.call_stream;
    *n_out = 0;
    int rc;
    if ((rc = d.stage_hay.ensure(nb * 2 + 16))) return rc;
    if ((rc = d.stage_out.ensure(cap * (uint64_t)record_kind + 16))) return rc;
    if (!d.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&d.copy_stream, hipStreamNonBlocking));
    // the ring: as many slots as this buffer has chunks (at most kPinSlots), each as large as its largest chunk -- a share of a
    // few megabytes of a multi-device call, or a 40 MiB haystack, does not pin 8 x 32 MiB
    const size_t slot_bytes = (((size_t)std::min<uint64_t>(nb, C) * 2 + (1u << 20) - 1) >> 20) << 20;
    const int slots_needed = (int)std::min<uint32_t>(n_chunks, (uint32_t)DeviceState::kPinSlots);
    if (d.pin_bytes < slot_bytes) {
        for (auto &q : d.pin) {
            if (q) (void)hipHostFree(q);
            q = nullptr;
        }
        d.pin_n = 0;
        d.pin_bytes = slot_bytes;
    }
    while (d.pin_n < slots_needed) {
        HIP_TRY(hipHostMalloc(&d.pin[d.pin_n], d.pin_bytes, hipHostMallocDefault));
        d.pin_n++;
    }
    while (d.chunk_ev.size() < n_chunks) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d.chunk_ev.push_back(e);
    }
    // producers: chunk k -> pinned slot k % kPinSlots (free once chunk k - kPinSlots has been copied to the device) -> DMA
    const int n_workers = (int)std::min<uint32_t>({(uint32_t)DeviceState::kPinSlots - 2, n_chunks, std::max(2u, std::thread::hardware_concurrency() / 2)});
    std::atomic<uint32_t> next_chunk{0};
    std::atomic<int> worker_rc{ACGPU_OK};
    std::vector<std::atomic<int>> ready(n_chunks); // 1: the chunk's DMA and event are enqueued
    for (auto &r : ready) r.store(0, std::memory_order_relaxed);
    const int dev = d.device;
    cpu_set_t node_cpus;
    const bool have_node = device_numa_cpus(dev, &node_cpus);
    auto worker = [&]() {
        if (have_node) (void)pthread_setaffinity_np(pthread_self(), sizeof(node_cpus), &node_cpus); // (best effort)
        if (hipSetDevice(dev) != hipSuccess) {
            worker_rc.store(ACGPU_E_HIP);
            return;
        }
        for (;;) {
            const uint32_t k = next_chunk.fetch_add(1);
            if (k >= n_chunks || worker_rc.load() != ACGPU_OK) return;
            const uint64_t b0 = (uint64_t)k * C, len = std::min<uint64_t>(C, nb - b0);
            const int slot = (int)(k % DeviceState::kPinSlots);
            if (k >= (uint32_t)DeviceState::kPinSlots) { // the slot's previous chunk must have left it
                while (!ready[k - DeviceState::kPinSlots].load(std::memory_order_acquire)) {
                    if (worker_rc.load() != ACGPU_OK) return;
                    std::this_thread::yield();
                }
                if (hipEventSynchronize(d.chunk_ev[k - DeviceState::kPinSlots]) != hipSuccess) {
                    worker_rc.store(ACGPU_E_HIP);
                    return;
                }
            }
            std::memcpy(d.pin[slot], haystack + lo + b0, len * 2);
            if (hipMemcpyAsync((char *)d.stage_hay.p + b0 * 2, d.pin[slot], len * 2, hipMemcpyHostToDevice, d.copy_stream) != hipSuccess ||
                hipEventRecord(d.chunk_ev[k], d.copy_stream) != hipSuccess) {
                worker_rc.store(ACGPU_E_HIP);
                return;
            }
            ready[k].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int i = 0; i < n_workers; ++i) pool.emplace_back(worker);
    } catch (...) {
        worker_rc.store(ACGPU_E_NOMEM);
    }
    auto join_all = [&]() {
        for (auto &th : pool) if (th.joinable()) th.join();
    };
    // consumer: shard k once the chunks its right halo reaches into have arrived (the chunks before a shard always have)
    const uint64_t right = (t.mode == ACGPU_MODE_WHOLEWORD || t.mode == ACGPU_MODE_WWLONGEST) ? (uint64_t)t.max_len + 1
                           : (t.mode == ACGPU_MODE_LONGEST ? (t.max_len ? t.max_len - 1 : 0) : 0);
    uint64_t total = 0;
    int64_t chain = chain_io ? *chain_io : 0;
    uint32_t waited = 0; // chunks whose arrival the compute stream already waits for
    int result = ACGPU_OK;
    for (uint32_t k = 0; k < n_chunks && result == ACGPU_OK; ++k) {
        const uint64_t c0 = std::max<uint64_t>((uint64_t)k * C, ob), c1 = std::min<uint64_t>({nb, (uint64_t)(k + 1) * C, oe});
        if (c0 >= c1) continue; // a chunk of halo units only
        const uint32_t need = (uint32_t)std::min<uint64_t>(n_chunks, (std::min<uint64_t>(nb, c1 + right) + C - 1) / C); // chunks [0, need)
        for (; waited < need && result == ACGPU_OK; ++waited) {
            while (!ready[waited].load(std::memory_order_acquire)) {
                if (worker_rc.load() != ACGPU_OK) { result = worker_rc.load(); break; }
                std::this_thread::yield();
            }
            if (result == ACGPU_OK && hipStreamWaitEvent(stream, d.chunk_ev[waited], 0) != hipSuccess) result = ACGPU_E_HIP;
        }
        if (result != ACGPU_OK) break;
        acgpu_shard sh{};
        sh.d_hay = (const uint16_t *)d.stage_hay.p;
        sh.n_units = std::min<uint64_t>(nb, (uint64_t)need * C); // what has arrived
        sh.own_begin = c0;
        sh.own_end = c1;
        sh.text_begin = lo == 0 ? 1 : 0;
        sh.text_end = (sh.n_units == nb && hi == n_units) ? 1 : 0;
        sh.chain_entry = t.mode == ACGPU_MODE_SHORTEST ? chain : std::max<int64_t>(chain, (int64_t)c0);
        uint64_t n_k = 0;
        const uint64_t room = total < cap ? cap - total : 0;
        rc = match_shard(a, d, &sh, record_kind, (char *)d.stage_out.p + std::min(total, cap) * (uint64_t)record_kind, room, &n_k, stream, nullptr);
        if (rc != ACGPU_OK && rc != ACGPU_E_OVERFLOW) {
            result = rc;
            break;
        }
        total += n_k; // (beyond cap: the remaining shards only count)
        if (t.mode == ACGPU_MODE_SHORTEST) chain = n_k ? sh.chain_exit : chain;
        else chain = sh.chain_exit;
    }
    if (result != ACGPU_OK) worker_rc.store(result); // (stops the producers)
    join_all();
    if (result == ACGPU_OK && worker_rc.load() != ACGPU_OK) result = worker_rc.load();
    (void)hipStreamSynchronize(d.copy_stream); // nothing of this call stays in flight
    if (result != ACGPU_OK) {
        if (result == ACGPU_E_HIP) g_last_hip_error = (int)hipGetLastError();
        return result;
    }
    *n_out = total;
    if (chain_io) *chain_io = chain;
    return total > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

} // namespace acgpu

namespace {

int match_u16_pipelined(acgpu_automaton *a, DeviceState &d, const uint16_t *haystack, uint64_t n_units, int record_kind, void *out,
                        uint64_t cap, uint64_t *n_out) {
    int64_t chain = 0;
    const int rc = scan_host_range(a, d, haystack, n_units, 0, n_units, 0, n_units, record_kind, cap, n_out, &chain);
    if (rc != ACGPU_OK) return rc;
    if (*n_out) {
        HIP_TRY(hipMemcpyAsync(out, d.stage_out.p, *n_out * (uint64_t)record_kind, hipMemcpyDeviceToHost, d.call_stream));
        HIP_TRY(hipStreamSynchronize(d.call_stream));
    }
    return ACGPU_OK;
}

} // namespace

namespace {

// acgpu_match_u16 on a short haystack: ONE launch of one workgroup (acgpu_small.hip) that reads the haystack from, and writes
// the records to, host-mapped pinned memory; the host waits on a flag in that memory.  *handled = false: the kernel could not
// hold the call (too many occurrences) -- the general path takes it.
int match_small(acgpu_automaton *a, DeviceState &d, const uint16_t *haystack, uint64_t n_units, int record_kind, void *out,
                uint64_t cap, uint64_t *n_out, bool *handled) {
    *handled = false;
    constexpr size_t kHayBytes = (size_t)kSmallMaxUnits * 2 + 64, kOutBytes = (size_t)kSmallMaxRecs * ACGPU_REC_MAP + 64;
    if (!d.small_pin) {
        // fine-grained (coherent) host memory: the device's writes are visible to the host while the kernel is still running
        HIP_TRY(hipHostMalloc(&d.small_pin, 64 + kHayBytes + kOutBytes, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(&d.small_pin_dev, d.small_pin, 0));
        HIP_TRY(hipStreamCreateWithFlags(&d.small_stream, hipStreamNonBlocking));
    }
    volatile unsigned long long *status = reinterpret_cast<volatile unsigned long long *>(d.small_pin);
    char *h_hay = (char *)d.small_pin + 64, *h_out = h_hay + kHayBytes;
    char *dev = (char *)d.small_pin_dev;
    std::memcpy(h_hay, haystack, n_units * 2);
    std::memset(h_hay + n_units * 2, 0, 8); // (the kernel reads whole 8-byte groups)
    status[0] = 0;
    status[1] = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    SmallCall c{};
    c.hay = reinterpret_cast<const uint16_t *>(dev + 64);
    c.n_units = (uint32_t)n_units;
    c.record_kind = record_kind;
    c.out = dev + 64 + kHayBytes;
    c.cap = (uint32_t)std::min<uint64_t>(cap, kSmallMaxRecs);
    c.status = reinterpret_cast<unsigned long long *>(dev);
    HIP_TRY(launch_small(d.T, a->t.mode, c, d.small_stream));
    // the flag; every so often the stream itself, so that a launch that failed behind the call cannot hang the host
    for (uint64_t spins = 1;; ++spins) {
        if (status[0] != 0) break;
        if ((spins & 0xfffffu) == 0) {
            const hipError_t q = hipStreamQuery(d.small_stream);
            if (q != hipErrorNotReady && status[0] == 0) {
                if (q == hipSuccess) continue; // (finished: the flag is on its way)
                g_last_hip_error = (int)q;
                (void)hipGetLastError();
                return ACGPU_E_HIP;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (status[0] != 1) return ACGPU_OK; // not handled
    *handled = true;
    *n_out = status[1];
    if (*n_out > cap) return ACGPU_E_OVERFLOW;
    if (*n_out) std::memcpy(out, h_out, *n_out * (uint64_t)record_kind);
    return ACGPU_OK;
}

} // namespace

extern "C" {

int acgpu_match_u16(const acgpu_automaton *ca, const uint16_t *haystack, uint64_t n_units, int record_kind, void *out,
                    uint64_t cap, uint64_t *n_out) {
    if (!ca || !n_out || (n_units && !haystack) || (cap && !out)) return ACGPU_E_INVALID;
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    if (n_units >= (1ull << 31)) return ACGPU_E_INVALID;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    DeviceState *d = nullptr;
    int rc = device_for_call(a, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(d->mu); // staging buffers are part of the per-device scratch pool
    // long haystacks of the families whose shards chain: the pipelined form (the loops that only exist as a sequential kernel
    // over the whole text -- WholeWord / WholeWordLongestSet with a fold-inconsistent table -- take the plain one)
    const HostTables &t = a->t;
    // short haystacks: one launch, no copies (tunable tile_debug bit 2^41, or a kernel form forced by "force_kernel": the
    // general path -- for A/B, and for the tests that run the scan kernels on the short edge-case inputs)
    if (n_units > 0 && n_units <= kSmallMaxUnits && d->inflight == 0 && small_call_supported(t) && !(tunables().tile_debug & (1ll << 41)) &&
        tunables().force_kernel == 0) {
        bool handled = false;
        rc = match_small(a, *d, haystack, n_units, record_kind, out, cap, n_out, &handled);
        if (rc != ACGPU_OK || handled) return rc;
    }
    const bool sequential_only = (t.mode == ACGPU_MODE_WHOLEWORD && !t.fold_consistent) ||
                                 (t.mode == ACGPU_MODE_WWLONGEST && !t.fold_consistent && record_kind == ACGPU_REC_SET);
    if (n_units >= 2 * kHostChunkUnits && !sequential_only && d->inflight == 0 && !(tunables().tile_debug & 33554432) &&
        (uint64_t)t.max_len + 2 < kHostChunkUnits)
        return match_u16_pipelined(a, *d, haystack, n_units, record_kind, out, cap, n_out);
    if ((rc = d->stage_hay.ensure(n_units * 2 + 16))) return rc;
    if ((rc = d->stage_out.ensure(cap * (uint64_t)record_kind + 16))) return rc;
    if (n_units) HIP_TRY(hipMemcpy(d->stage_hay.p, haystack, n_units * 2, hipMemcpyHostToDevice));
    acgpu_shard sh{};
    sh.d_hay = (const uint16_t *)d->stage_hay.p;
    sh.n_units = n_units;
    sh.own_begin = 0;
    sh.own_end = n_units;
    sh.text_begin = 1;
    sh.text_end = 1;
    sh.chain_entry = 0;
    rc = match_shard(a, *d, &sh, record_kind, d->stage_out.p, cap, n_out, nullptr, nullptr);
    if (rc != ACGPU_OK) return rc;
    if (*n_out) HIP_TRY(hipMemcpy(out, d->stage_out.p, *n_out * (uint64_t)record_kind, hipMemcpyDeviceToHost));
    return ACGPU_OK;
}

int acgpu_match_batch_u16(const acgpu_automaton *ca, const uint16_t *units, const uint64_t *offsets, uint32_t n_haystacks,
                          int record_kind, void *out, uint64_t cap, uint64_t *n_out) {
    if (!ca || !n_out || !offsets || (cap && !out)) return ACGPU_E_INVALID;
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    *n_out = 0;
    if (n_haystacks == 0) return ACGPU_OK;
    for (uint32_t i = 0; i < n_haystacks; i++)
        if (offsets[i] > offsets[i + 1]) return ACGPU_E_INVALID;
    const uint64_t total = offsets[n_haystacks] - offsets[0];
    if (total && !units) return ACGPU_E_INVALID;
    const uint64_t cat = total + n_haystacks; // one separator behind every haystack
    if (cat >= (1ull << 31)) return ACGPU_E_INVALID;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    const HostTables &t = a->t;
    const size_t out_rec = (size_t)record_kind + 4;
    // no unit can stand between two haystacks (every one of the 65536 is in use), or a word matcher over a table that is not
    // fold-consistent: one call per haystack.  (Such a table makes some loops sequential kernels over one whole text; and in
    // the folding scans a keyword's FOLDED first unit need not be a word character, so a walk that begins at position 0 of a
    // text -- where the scan starts whatever stands there -- is not a walk that begins behind a separator.)
    const bool per_haystack = (t.mode == ACGPU_MODE_WHOLEWORD || t.mode == ACGPU_MODE_WWLONGEST) && !t.fold_consistent;
    if (t.sep_unit < 0 || per_haystack) {
        std::vector<int32_t> tmp;
        uint64_t n = 0;
        for (uint32_t i = 0; i < n_haystacks; i++) {
            const uint64_t len = offsets[i + 1] - offsets[i];
            uint64_t got = 0, room = cap > n ? cap - n : 0;
            try {
                tmp.resize(std::max<size_t>(room * (record_kind / 4), 4));
            } catch (...) {
                return ACGPU_E_NOMEM;
            }
            const int rc = acgpu_match_u16(ca, units + offsets[i], len, record_kind, tmp.data(), room, &got);
            if (rc != ACGPU_OK && rc != ACGPU_E_OVERFLOW) return rc;
            if (rc == ACGPU_OK) {
                const int W = record_kind / 4;
                for (uint64_t r = 0; r < got; r++) {
                    int32_t *o = (int32_t *)((char *)out + (n + r) * out_rec);
                    o[0] = (int32_t)i;
                    for (int w = 0; w < W; w++) o[1 + w] = tmp[r * W + w];
                }
            }
            n += got;
        }
        *n_out = n;
        return n > cap ? ACGPU_E_OVERFLOW : ACGPU_OK;
    }
    DeviceState *d = nullptr;
    int rc = device_for_call(a, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(d->mu); // staging buffers are part of the per-device scratch pool
    if (d->inflight > 0) return ACGPU_E_INVALID; // (the NULL stream: see the stream rule)
    const size_t off_bytes = ((size_t)n_haystacks + 1) * 4, pin_need = cat * 2 + 64 + off_bytes;
    if (d->batch_pin_bytes < pin_need) {
        if (d->batch_pin) (void)hipHostFree(d->batch_pin);
        d->batch_pin = nullptr;
        d->batch_pin_bytes = 0;
        HIP_TRY(hipHostMalloc(&d->batch_pin, pin_need + pin_need / 4, hipHostMallocDefault));
        d->batch_pin_bytes = pin_need + pin_need / 4;
    }
    uint16_t *h_cat = (uint16_t *)d->batch_pin;
    uint32_t *h_off = (uint32_t *)((char *)d->batch_pin + ((cat * 2 + 63) & ~(size_t)63));
    {
        uint64_t at = 0;
        for (uint32_t i = 0; i < n_haystacks; i++) {
            const uint64_t len = offsets[i + 1] - offsets[i];
            h_off[i] = (uint32_t)at;
            if (len) std::memcpy(h_cat + at, units + offsets[i], len * 2);
            at += len;
            h_cat[at++] = (uint16_t)t.sep_unit;
        }
        h_off[n_haystacks] = (uint32_t)at;
    }
    if ((rc = d->stage_hay.ensure(cat * 2 + 16))) return rc;
    if ((rc = d->stage_out.ensure(cap * (uint64_t)record_kind + 16))) return rc;
    if ((rc = d->batch_off.ensure(off_bytes + 16))) return rc;
    if ((rc = d->batch_out.ensure(cap * out_rec + 16))) return rc;
    HIP_TRY(hipMemcpyAsync(d->stage_hay.p, h_cat, cat * 2, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(d->batch_off.p, h_off, off_bytes, hipMemcpyHostToDevice, nullptr));
    acgpu_shard sh{};
    sh.d_hay = (const uint16_t *)d->stage_hay.p;
    sh.n_units = cat;
    sh.own_begin = 0;
    sh.own_end = cat;
    sh.text_begin = 1;
    sh.text_end = 1;
    sh.chain_entry = 0;
    d->start_behind = t.sep_unit; // (WholeWordLongest: every haystack's first unit is a walk start, as position 0 of a text is)
    rc = match_shard(a, *d, &sh, record_kind, d->stage_out.p, cap, n_out, nullptr, nullptr);
    d->start_behind = -1;
    if (rc != ACGPU_OK) return rc; // ACGPU_E_OVERFLOW: *n_out is the capacity to retry with
    if (*n_out) {
        HIP_TRY(launch_batch_tag(d->stage_out.p, *n_out, record_kind, (const uint32_t *)d->batch_off.p, n_haystacks, d->batch_out.p, nullptr));
        HIP_TRY(hipMemcpy(out, d->batch_out.p, *n_out * out_rec, hipMemcpyDeviceToHost));
    }
    return ACGPU_OK;
}

int acgpu_stream_probe(const void *d_buf, uint64_t n_bytes, void *stream_, int repeats, int pattern, float *ms_median) {
    if (pattern != 0 && pattern != 1) return ACGPU_E_INVALID;
    if (!d_buf || !ms_median || ((uintptr_t)d_buf & 15) || n_bytes < (1ull << 20) || repeats < 1 || repeats > 64) return ACGPU_E_INVALID;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return ACGPU_E_HIP; }
    unsigned *d_sink = nullptr;
    int rc = ACGPU_OK;
    std::vector<float> ms;
    int dev = 0, n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n_cu = prop.multiProcessorCount;
    if (hipMalloc((void **)&d_sink, 64) != hipSuccess) rc = ACGPU_E_NOMEM;
    for (int r = 0; rc == ACGPU_OK && r <= repeats; ++r) { // (the first run is a warm-up)
        float t = 0;
        if (hipEventRecord(e0, stream) != hipSuccess || launch_stream_probe(d_buf, n_bytes, n_cu, d_sink, pattern, stream) != hipSuccess ||
            hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&t, e0, e1) != hipSuccess) {
            g_last_hip_error = (int)hipGetLastError();
            rc = ACGPU_E_HIP;
        } else if (r) ms.push_back(t);
    }
    if (d_sink) (void)hipFree(d_sink);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != ACGPU_OK) return rc;
    std::sort(ms.begin(), ms.end());
    *ms_median = ms[ms.size() / 2];
    return ACGPU_OK;
}

int acgpu_synth_fill(uint16_t *d_dst, uint64_t n_units, uint64_t start_index, uint64_t seed, const uint16_t *table,
                     uint32_t table_len, void *stream) {
    if ((n_units && !d_dst) || !table || table_len == 0 || table_len > 64) return ACGPU_E_INVALID;
    HIP_TRY(launch_synth_fill(d_dst, n_units, start_index, seed, table, table_len, reinterpret_cast<hipStream_t>(stream)));
    return ACGPU_OK;
}

int acgpu_synth_tokens(uint16_t *d_dst, uint64_t n_units, uint64_t seed, const uint16_t *kw_units, const uint64_t *kw_off,
                       uint32_t n_kw, const uint16_t *swapcase_tbl, void *stream_) {
    if ((n_units && !d_dst) || (n_kw && (!kw_units || !kw_off))) return ACGPU_E_INVALID;
    if (n_units == 0) return ACGPU_OK;
    // a token is a word -- of the dictionary, or a random one of 2..12 units -- plus 1..3 separators: its shortest form bounds
    // the number of tokens that cover the haystack (a dictionary with a 1-unit or empty word makes 2- and 1-unit tokens)
    uint64_t min_word = 2;
    for (uint32_t i = 0; i < n_kw; ++i) min_word = std::min<uint64_t>(min_word, kw_off[i + 1] - kw_off[i]);
    const uint64_t n_tokens64 = n_units / (min_word + 1) + 2;
    if (n_tokens64 >= (1ull << 32)) return ACGPU_E_INVALID;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    // the benchmark's scripts and separators (ahocorasick_amd/synth.py: _SCRIPTS, _SEPARATORS)
    static const uint16_t ranges[][2] = {{0x41, 0x5A}, {0x61, 0x7A}, {0xC0, 0xD6}, {0xD8, 0xF6}, {0xF8, 0xFF}, // latin
                                         {0x0391, 0x03A1}, {0x03A3, 0x03A9}, {0x03B1, 0x03C9},                // greek
                                         {0x0410, 0x044F},                                                    // cyrillic
                                         {0x4E00, 0x9FA5},                                                    // cjk
                                         {0xAC00, 0xD7A3},                                                    // hangul
                                         {0x0621, 0x063A}, {0x0641, 0x064A}};                                 // arabic
    static const int script_first[7] = {0, 5, 8, 9, 10, 11, 13};
    static const uint16_t seps[6] = {0x20, ',', '.', 0x0A, 0x3002, 0x2014};
    std::vector<uint16_t> scripts;
    uint32_t script_off[7];
    std::vector<uint32_t> off32;
    try {
        for (int sc = 0; sc < 6; ++sc) {
            script_off[sc] = (uint32_t)scripts.size();
            for (int r = script_first[sc]; r < script_first[sc + 1]; ++r)
                for (uint32_t u = ranges[r][0]; u <= ranges[r][1]; ++u) scripts.push_back((uint16_t)u);
        }
        script_off[6] = (uint32_t)scripts.size();
        off32.resize((size_t)n_kw + 1);
        for (uint32_t i = 0; i <= n_kw; ++i) {
            const uint64_t o = n_kw ? kw_off[i] - kw_off[0] : 0;
            if (o >= (1ull << 32)) return ACGPU_E_INVALID;
            off32[i] = (uint32_t)o;
        }
    } catch (...) {
        return ACGPU_E_NOMEM;
    }
    const uint32_t n_tokens = (uint32_t)n_tokens64; // these cover the haystack
    const size_t kw_bytes = n_kw ? (size_t)off32[n_kw] * 2 : 0;
    DevBuf b_kw, b_off, b_sw, b_sc, b_len, b_start, b_tmp;
    auto release = [&]() { b_kw.release(); b_off.release(); b_sw.release(); b_sc.release(); b_len.release(); b_start.release(); b_tmp.release(); };
    int rc = ACGPU_OK;
    if ((rc = b_kw.ensure(kw_bytes + 16)) || (rc = b_off.ensure(off32.size() * 4 + 16)) || (rc = b_sc.ensure(scripts.size() * 2 + 16)) ||
        (rc = b_len.ensure((size_t)n_tokens * 4 + 16)) || (rc = b_start.ensure((size_t)n_tokens * 8 + 16)) ||
        (rc = b_tmp.ensure(((size_t)n_tokens / 2048 + 2) * 8 + 16)) || (swapcase_tbl && (rc = b_sw.ensure(65536 * 2)))) {
        release();
        return rc;
    }
    hipError_t e = hipSuccess;
    if (kw_bytes) e = hipMemcpy(b_kw.p, kw_units + kw_off[0], kw_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b_off.p, off32.data(), off32.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b_sc.p, scripts.data(), scripts.size() * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess && swapcase_tbl) e = hipMemcpy(b_sw.p, swapcase_tbl, 65536 * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launch_token_stream(d_dst, n_units, seed, (const uint16_t *)b_kw.p, (const uint32_t *)b_off.p, n_kw,
                                swapcase_tbl ? (const uint16_t *)b_sw.p : nullptr, (const uint16_t *)b_sc.p, script_off, seps, n_tokens,
                                (uint32_t *)b_len.p, (uint64_t *)b_start.p, (uint64_t *)b_tmp.p, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); // (the scratch is released below)
    release();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? ACGPU_E_NOMEM : ACGPU_E_HIP;
    }
    return ACGPU_OK;
}

} // extern "C"
