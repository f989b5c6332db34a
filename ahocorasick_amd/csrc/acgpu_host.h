// acgpu_host.h -- host-side state shared by the C ABI glue (acgpu_api.hip) and the multi-device driver (acgpu_multi.hip):
// grow-only device buffers, tickets, the per-automaton per-device scratch pool, the automaton handle.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "acgpu_internal.h"
#include "acgpu_kernels.h"

namespace acgpu {

extern thread_local int g_last_hip_error;

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            g_last_hip_error = (int)_e;                \
            (void)hipGetLastError();                   \
            return _e == hipErrorOutOfMemory ? ACGPU_E_NOMEM : (_e == hipErrorNoDevice ? ACGPU_E_NODEVICE : ACGPU_E_HIP); \
        }                                              \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return ACGPU_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4 + 256;
        HIP_TRY(hipMalloc(&p, want));
        bytes = want;
        return ACGPU_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// One asynchronous call in flight (acgpu_match_device_begin/_end): its own events and pinned count slot.
struct Ticket {
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t done = nullptr;
    unsigned long long *h_count = nullptr; // pinned, 64 bytes
    bool busy = false, profiled = false;
    bool done_is_ev2 = false; // the completion to wait for is ev[2] (the finalize kernel's own end), not `done`
    bool one_kernel = false;  // the call was one kernel (the tile kernel with the fused tail): ev[0] .. ev[2] is its dispatch, ev[1] is not recorded
    // how _end collects it: 0 = the AhoCorasick / WholeWord pipeline (count and overflow word in h_count), 1 = a chain pipeline
    // that was enqueued (LONGEST walk: count in h_count[0], chain exit in h_count[2]), 2 = the call ran synchronously inside
    // _begin (the other families): everything is in the fields below
    int kind = 0;
    int sync_rc = 0;
    uint64_t sync_n = 0;
    acgpu_profile sync_prof{};
    acgpu_shard *user_shard = nullptr; // receives chain_exit in _end
    uint64_t cap = 0, scanned = 0;
    char kname[64] = {0};
    void *owner = nullptr; // the DeviceState it belongs to
    // what _end needs to redo the call when the split form's candidate slices were too small
    acgpu_shard shard{};
    int record_kind = 0;
    void *d_out = nullptr;
    hipStream_t stream = nullptr;
    int bits_level = 0; // LONGEST, k_longest_bits / k_longest_follow: the run-up level the call was enqueued with (0: short, 1: a whole segment)
};

struct DeviceState {
    std::mutex mu;   // one call at a time on this scratch pool (the automaton's own mutex only guards its map of these)
    int device = -1;
    int lane = 0;    // 0: what the single-device entries use; > 0: further shards of a multi-device call on the SAME device
    hipStream_t call_stream = nullptr; // the stream the host-haystack entries work on: the NULL stream for lane 0, its own otherwise
    int n_cu = 256;      // CUs the scan kernels size their grids for: n_cu_phys minus the tunable reserve_cus
    int n_cu_phys = 256; // multiProcessorCount of the device
    DevTables T{};
    const uint8_t *wflags_f = nullptr; // word-character tables of the loops that fold in every lookup (HostTables::wflags_f)
    const uint32_t *wbits_f = nullptr;
    int start_behind = -1; // set for the duration of a batch call: the separator unit (k_wwl_starts: a haystack's first unit is a walk start)
    std::vector<void *> table_allocs;
    // scratch pool (one in-flight match per automaton and device)
    DevBuf counter, chunk_counts, offsets, scan_tmp, scratch, chain, lenbuf, statebuf;
    DevBuf stage_hay, stage_out; // acgpu_match_u16 staging
    // the one-launch form for short haystacks (acgpu_small.hip): host-mapped pinned block [status | haystack | records], its stream
    void *small_pin = nullptr, *small_pin_dev = nullptr;
    hipStream_t small_stream = nullptr;
    DevBuf multi_win, multi_tail; // multi-device calls, chain families: records of a repair window, the kept tail of the speculation
    // batch entry: pinned concatenation + offsets, device offsets, tagged records
    void *batch_pin = nullptr;
    size_t batch_pin_bytes = 0;
    DevBuf batch_off, batch_out;
    // pipelined host entry: pinned staging ring (one slot per chunk in flight), its copy stream, one event per chunk
    static constexpr int kPinSlots = 8;
    void *pin[kPinSlots] = {nullptr};
    size_t pin_bytes = 0; // bytes per slot
    int pin_n = 0;        // slots allocated
    hipStream_t multi_stream = nullptr; // acgpu_match_u16_multi: this pool's stream for the duration of such a call (kept between calls)
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> chunk_ev;
    DevBuf short_recs, short_nxt, short_tmp, short_mark; // SHORTEST: all-matches list + selection scratch
    DevBuf ww_recs;                                      // WHOLEWORD: region-local record slots (TileLaunch::d_region_recs)
    DevBuf blockmax;                                     // LONGEST: farthest landing per 64 positions
    DevBuf lenbig, todo;                                 // LONGEST: escaped lengths; root-table form: flagged chunks
    DevBuf chainbits;                                    // LONGEST: one bit per position, set where the chain reports a match
    DevBuf bits_state;                                   // k_longest_bits: exit / flag / count, look-back words, region counter -- zero between calls
    double all_density = -1.0;                           // ALL: records per unit of this pool's last call (-1: none yet): k_ac_states or the tile kernel
    int fol_level = 0;                                   // k_longest_follow: 0 = run-up of 128 positions, 1 = of a whole segment (a call's chains had not merged), 2 = not for this pool's texts
    void *bits_state_seen = nullptr;                     // (a re-allocated buffer, or a call that failed half way, is zeroed by a memset)
    DevBuf cands, region_cands;                          // ALL, split form: candidate positions, {first, count} per region
    DevBuf wwl_rs, wwl_mend, wwl_mid, wwl_sel, wwl_stop, wwl_nxt0; // WWLONGEST: walk starts, what each would report, where it stops
    unsigned long long *h_counter = nullptr; // pinned
    // match_all: two sets of slot counters alternate; the permute pass of a call zeroes the set the next call uses
    int cset = 0;
    bool cclean[2] = {false, false};
    void *counter_seen = nullptr; // (a re-allocated counter buffer is not clean)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    Ticket tickets[4];
    // stream rule (include/acgpu.h): while tickets are in flight every call on this automaton and device uses their stream
    int inflight = 0;
    hipStream_t inflight_stream = nullptr;
    ~DeviceState() {
        for (void *p : table_allocs) (void)hipFree(p);
        counter.release(); chunk_counts.release(); offsets.release(); scan_tmp.release(); scratch.release();
        chain.release(); lenbuf.release(); statebuf.release(); stage_hay.release(); stage_out.release();
        multi_win.release(); multi_tail.release();
        if (call_stream) (void)hipStreamDestroy(call_stream);
        if (multi_stream) (void)hipStreamDestroy(multi_stream);
        if (small_stream) (void)hipStreamDestroy(small_stream);
        if (small_pin) (void)hipHostFree(small_pin);
        short_recs.release(); short_nxt.release(); short_tmp.release(); short_mark.release();
        wwl_rs.release(); wwl_mend.release(); wwl_mid.release(); wwl_sel.release(); wwl_stop.release(); wwl_nxt0.release(); ww_recs.release(); blockmax.release(); lenbig.release(); todo.release(); chainbits.release(); bits_state.release(); cands.release(); region_cands.release();
        if (h_counter) (void)hipHostFree(h_counter);
        for (auto &q : pin) if (q) (void)hipHostFree(q);
        if (batch_pin) (void)hipHostFree(batch_pin);
        batch_off.release(); batch_out.release();
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto &e : chunk_ev) if (e) (void)hipEventDestroy(e);
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        for (auto &tk : tickets) {
            for (auto &e : tk.ev) if (e) (void)hipEventDestroy(e);
            if (tk.done) (void)hipEventDestroy(tk.done);
            if (tk.h_count) (void)hipHostFree(tk.h_count);
        }
    }
};


} // namespace acgpu

namespace acgpu {
// what a pipelined stream (acgpu_stream.hip) works in: two chunks in pinned host memory and on the device, one scan's records on
// the device and in host-mapped pinned memory.  Pinning 30 MB takes longer than streaming 100 MB, and a stream is single-use:
// a closed stream leaves its buffers to the automaton, the next one on the same device takes them over.
struct StreamBufs {
    int device = -1;
    void *pin[2] = {nullptr, nullptr};
    size_t pin_bytes[2] = {0, 0};
    DevBuf dev[2];
    DevBuf out_dev;
    void *out_pin = nullptr, *out_pin_dev = nullptr;
    size_t out_pin_bytes = 0;
    hipStream_t copy_stream = nullptr;          // the chunk transfers' stream (creating one per stream object cost milliseconds)
    hipEvent_t arrived[2] = {nullptr, nullptr}; // chunk i has arrived on the device
    void release() { // (the caller has made `device` current)
        if (copy_stream) {
            (void)hipStreamSynchronize(copy_stream);
            (void)hipStreamDestroy(copy_stream);
            copy_stream = nullptr;
        }
        for (auto &e : arrived) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            pin[i] = nullptr;
            pin_bytes[i] = 0;
            dev[i].release();
        }
        out_dev.release();
        if (out_pin) (void)hipHostFree(out_pin);
        out_pin = out_pin_dev = nullptr;
        out_pin_bytes = 0;
    }
};
} // namespace acgpu

struct acgpu_automaton {
    acgpu::HostTables t;
    std::vector<acgpu::StreamBufs> stream_cache;                                 // guarded by mu
    std::set<struct acgpu_stream *> open_streams;                                // guarded by mu: acgpu_free detaches them (acgpu_stream::a = nullptr)
    std::mutex mu;                                                               // guards `dev`, `stream_cache` and `open_streams`
    std::map<std::pair<int, int>, std::unique_ptr<acgpu::DeviceState>> dev;      // (HIP device, lane) -> scratch pool + tables
};

namespace acgpu {

// The scratch pool of `a` on the CURRENT HIP device (created and the tables uploaded on first use).  lane > 0: a further,
// independent pool on the same device (its own stream), for multi-device calls that list a device more than once.
int device_for_call(acgpu_automaton *a, DeviceState **d, int lane = 0);

// Validates a shard and runs the pipeline of the automaton's family on `stream`; the caller holds d.mu.
// readable: the call stands for match(Readable, ...) (acgpu_stream_feed).
int match_shard(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap, uint64_t *n_out,
                hipStream_t stream, acgpu_profile *prof, bool readable = false);

// acgpu_match_device_begin on a given scratch pool: the caller holds d.mu and has made d's device current.  The ticket is
// collected with acgpu_match_device_end / _abandon (which take d.mu themselves).
int begin_shard(acgpu_automaton *a, DeviceState &d, acgpu_shard *sh, int record_kind, void *d_out, uint64_t cap, hipStream_t stream,
                int want_profile, acgpu_ticket **ticket);

// acgpu_match_device_end; *redone: the call had to be redone inside (a scratch slice had filled up) -- the records and the device
// result were written a second time, AFTER whatever the caller had enqueued behind the first attempt.
int end_ticket(const acgpu_automaton *a, acgpu_ticket *ticket, uint64_t *n_out, acgpu_profile *prof, bool *redone);

// acgpu_match_u16 on a long haystack, pipelined over chunks (acgpu_api.hip): the units [lo, hi) of `haystack` become the
// device buffer d.stage_hay (buffer unit 0 = unit lo), the owned range [own_lo, own_hi) is scanned as shards of it on
// d.call_stream, the records (buffer relative) land in d.stage_out.  chain: in = entry, out = exit (buffer relative).
int scan_host_range(acgpu_automaton *a, DeviceState &d, const uint16_t *haystack, uint64_t n_units, uint64_t lo, uint64_t hi,
                    uint64_t own_lo, uint64_t own_hi, int record_kind, uint64_t cap, uint64_t *n_out, int64_t *chain);

} // namespace acgpu
