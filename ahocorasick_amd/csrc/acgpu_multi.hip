// acgpu_multi.hip -- ONE host process, several devices: the multi-GPU split of a long haystack behind the C ABI
// (include/acgpu.h: acgpu_match_u16_multi, acgpu_comm_*, acgpu_match_device_allgather).
//
// SURVEY.md 8e, one row per matcher family -- the same partitioning ahocorasick_amd/dist.py does with one process per GPU:
//  * AhoCorasick: contiguous shares, (max_len-1) units of left halo, a match belongs to the share that owns its LAST unit;
//  * WholeWord: a word belongs to the share that owns its FIRST unit; 1 unit of left context, max_len+1 units of right halo;
//  * Longest / Shortest / WholeWordLongest: what a share reports depends on ONE number from the share before it (where the
//    greedy chain enters / where matching last restarted / from where the scan looks for its next word start).  Every share
//    is scanned speculatively, all devices at once ("nothing comes in"); then, share by share, a share whose true entry
//    differs from the assumption re-runs a short WINDOW at its head twice -- from the assumed and from the true entry -- until
//    both scans leave the window in the same state: from there on they are the same scan, and the share's records are the
//    true window's followed by the speculation's behind the window.  (Chains merge within a few keyword lengths: a window of
//    4096 units as a rule, x4 until it works, at worst the whole share -- then its exit changes and the next share sees that.)
// Share-local order is the reference's listener-call order, so the shares' record lists, concatenated by share, are the
// reference's call order for the whole haystack.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "acgpu_host.h"

using namespace acgpu;

namespace {

// ---- small device helpers -------------------------------------------------------------------------------------------------
// records {start, end[, id]}: start and end shifted by `delta` (view-relative -> global positions)
__global__ void k_shift_records(int32_t *recs, uint64_t n, int cols, int32_t delta) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    recs[i * cols] += delta;
    recs[i * cols + 1] += delta;
}

// first record whose field (0: start, 1: end) is >= value (upper: > value); records ascend in that field.  One lane.
__global__ void k_record_bound(const int32_t *recs, uint64_t n, int cols, int field, int32_t value, int upper,
                               unsigned long long *out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const int32_t v = recs[mid * cols + field];
        if (upper ? v <= value : v < value) lo = mid + 1;
        else hi = mid;
    }
    *out = lo;
}

struct Halo {
    uint64_t left, right;
};

Halo halos_for(const HostTables &t) {
    const uint64_t m = t.max_len;
    switch (t.mode) {
    case ACGPU_MODE_WHOLEWORD:
    case ACGPU_MODE_WWLONGEST: return {1, m + 1};
    case ACGPU_MODE_LONGEST: return {0, m ? m - 1 : 0};
    default: return {m ? m - 1 : 0, 0}; // ALL, SHORTEST
    }
}

bool chain_family(const HostTables &t) {
    return t.mode == ACGPU_MODE_LONGEST || t.mode == ACGPU_MODE_SHORTEST || t.mode == ACGPU_MODE_WWLONGEST;
}

uint64_t round_up8(uint64_t v) { return (v + 7) & ~7ull; }

// One share of the haystack on one device.  All positions are relative to the share's device buffer.
struct Share {
    int device = 0, lane = 0;
    DeviceState *d = nullptr;
    hipStream_t stream = nullptr;
    const uint16_t *d_hay = nullptr;
    uint64_t n_units = 0, own_begin = 0, own_end = 0;
    int text_begin = 0, text_end = 0;
    // the speculative scan: its records, their number, the state in which it left the share (chain_exit of the call)
    void *spec = nullptr;
    uint64_t spec_cap = 0, n_spec = 0;
    int64_t spec_exit = 0;
    // the share's final record list = `win` (records of the true repair window, in d->multi_win) + the speculation from `keep_from`
    uint64_t n_win = 0, keep_from = 0;
    bool repaired = false;
    // what the next share needs: the state in which the TRUE scan leaves this share, relative to own_end
    int64_t exit_rel = 0;
    bool exit_valid = true; // SHORTEST: false = this share (and nothing before it) reported no match: "nothing comes in"
    int rc = ACGPU_OK;
    int hip_error = 0; // g_last_hip_error of the host thread that worked on the share (thread_local: the caller's thread republishes it)
    uint64_t n_final() const { return repaired ? n_win + (n_spec - keep_from) : n_spec; }
};

int set_device(int dev) {
    HIP_TRY(hipSetDevice(dev));
    return ACGPU_OK;
}

// one synchronous native call on the sub-range [own_begin, w_end) of a share, into a grow-only private buffer
int scan_window(acgpu_automaton *a, Share &s, uint64_t w_end, int64_t entry, DevBuf &dst, int record_kind, uint64_t *n, int64_t *exit) {
    uint64_t cap = (dst.bytes > 16 ? dst.bytes - 16 : 0) / (uint64_t)record_kind; // (what it holds already: no growth per call)
    if (cap < 1024) cap = 1024;
    for (;;) {
        int rc = dst.ensure(cap * (uint64_t)record_kind + 16);
        if (rc) return rc;
        acgpu_shard sh{};
        sh.d_hay = s.d_hay;
        sh.n_units = s.n_units;
        sh.own_begin = s.own_begin;
        sh.own_end = w_end;
        sh.text_begin = s.text_begin;
        sh.text_end = s.text_end;
        sh.chain_entry = entry;
        rc = match_shard(a, *s.d, &sh, record_kind, dst.p, cap, n, s.stream, nullptr);
        if (rc == ACGPU_E_OVERFLOW) {
            cap = *n + *n / 4 + 16;
            continue;
        }
        if (rc) return rc;
        *exit = sh.chain_exit;
        return ACGPU_OK;
    }
}

int record_bound(Share &s, const void *recs, uint64_t n, int record_kind, int field, int64_t value, bool upper, uint64_t *idx) {
    if (n == 0 || value > 0x7fffffffll) {
        *idx = value > 0x7fffffffll ? n : 0;
        return ACGPU_OK;
    }
    unsigned long long *d_slot = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&d_slot, s.d->h_counter + 5, 0));
    hipLaunchKernelGGL(k_record_bound, dim3(1), dim3(1), 0, s.stream, (const int32_t *)recs, n, record_kind / 4, field,
                       (int32_t)std::max<int64_t>(value, -0x7fffffffll), upper ? 1 : 0, d_slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s.stream));
    *idx = s.d->h_counter[5];
    return ACGPU_OK;
}

// The chain families, after every share has been scanned speculatively: share by share, the true entry from the share before.
// Speculation per family (what the scan phase assumed): LONGEST / WWLONGEST the scan enters at the share's first unit;
// SHORTEST no restart position restricts anything.
int repair_chains(acgpu_automaton *a, std::vector<Share> &sh, int record_kind) {
    const HostTables &t = a->t;
    const int64_t halo = t.max_len ? (int64_t)t.max_len - 1 : 0;
    // share 0's exit is the speculation's (its entry is the text's)
    for (size_t i = 0; i < sh.size(); ++i) {
        Share &s = sh[i];
        int rc = set_device(s.device);
        if (rc) return rc;
        s.repaired = false;
        s.keep_from = 0;
        s.n_win = 0;
        if (t.mode == ACGPU_MODE_SHORTEST) {
            // the state: the end of the last match reported so far (nothing reported: what came in)
            // (share 0 scanned from the CALLER's entry, which its chain_exit hands on when it reports nothing: always the truth)
            s.exit_valid = s.n_spec > 0 || i == 0;
            s.exit_rel = s.spec_exit - (int64_t)s.own_end;
        } else {
            s.exit_valid = true;
            s.exit_rel = s.spec_exit - (int64_t)s.own_end;
        }
        if (i == 0) continue;
        const Share &p = sh[i - 1];
        // the true entry, in this share's coordinates (the previous share's own_end is this share's own_begin)
        int64_t entry;
        bool differs;
        if (t.mode == ACGPU_MODE_SHORTEST) {
            if (!p.exit_valid) continue; // nothing has been reported yet: the speculation is the truth
            entry = (int64_t)s.own_begin + p.exit_rel;
            differs = entry > (int64_t)s.own_begin - halo; // a restart left of every match that can end in this share forbids nothing
            if (!differs) {
                if (s.n_spec == 0) { // what came in goes on
                    s.exit_valid = true;
                    s.exit_rel = entry - (int64_t)s.own_end;
                }
                continue;
            }
            if (entry < 0) entry = 0; // (left of the buffer: restricts nothing more than the buffer's start does)
        } else {
            entry = (int64_t)s.own_begin + p.exit_rel;
            differs = entry > (int64_t)s.own_begin;
            if (!differs) continue;
        }
        const int64_t spec_entry = t.mode == ACGPU_MODE_SHORTEST ? 0 : (int64_t)s.own_begin;
        uint64_t w = 4096;
        for (;;) {
            // the window reaches beyond the true entry (a match of the previous share may cover this share's head)
            uint64_t w_end = s.own_begin + w;
            if ((int64_t)w_end <= entry) w_end = (uint64_t)entry + w;
            w_end = std::min<uint64_t>(round_up8(w_end), s.own_end);
            uint64_t n_t = 0, n_s = 0;
            int64_t ex_t = 0, ex_s = 0;
            if ((rc = scan_window(a, s, w_end, entry, s.d->multi_win, record_kind, &n_t, &ex_t))) return rc;
            if (w_end == s.own_end) { // the window is the whole share: nothing of the speculation is kept
                s.repaired = true;
                s.n_win = n_t;
                s.keep_from = s.n_spec;
                if (t.mode == ACGPU_MODE_SHORTEST) {
                    s.exit_valid = true;
                    s.exit_rel = (n_t ? ex_t : entry) - (int64_t)s.own_end;
                } else {
                    s.exit_rel = ex_t - (int64_t)s.own_end;
                }
                break;
            }
            if ((rc = scan_window(a, s, w_end, spec_entry, s.d->multi_tail, record_kind, &n_s, &ex_s))) return rc;
            bool same;
            if (t.mode == ACGPU_MODE_SHORTEST) {
                // a restart at or left of w_end - halo restricts nothing that ends behind the window
                const int64_t floor = (int64_t)w_end - halo;
                const int64_t st_t = std::max<int64_t>(n_t ? ex_t : entry, floor), st_s = std::max<int64_t>(n_s ? ex_s : -1, floor);
                same = st_t == st_s;
            } else {
                same = ex_t == ex_s;
            }
            if (same) {
                // the speculation's records behind the window: LONGEST / WWLONGEST own by their first unit (start >= w_end),
                // SHORTEST by their last (end > w_end)
                uint64_t idx = 0;
                if (t.mode == ACGPU_MODE_SHORTEST) rc = record_bound(s, s.spec, s.n_spec, record_kind, 1, (int64_t)w_end, true, &idx);
                else rc = record_bound(s, s.spec, s.n_spec, record_kind, 0, (int64_t)w_end, false, &idx);
                if (rc) return rc;
                s.repaired = true;
                s.n_win = n_t;
                s.keep_from = idx;
                if (t.mode == ACGPU_MODE_SHORTEST && idx == s.n_spec) { // no speculative record is left: the window's last one, or what came in
                    s.exit_valid = true;
                    s.exit_rel = (n_t ? ex_t : entry) - (int64_t)s.own_end;
                }
                break;
            }
            w *= 4;
        }
    }
    return ACGPU_OK;
}

// ---- RCCL, bound at run time (libacgpu.so does not link it: a process that never gathers never loads it) --------------------
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        // a process that has PyTorch-ROCm loaded already holds its bundled copy under this name; a JVM gets /opt/rocm's
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (x.lib) {
            x.CommInitAll = (int (*)(void **, int, const int *))dlsym(x.lib, "ncclCommInitAll");
            x.CommDestroy = (int (*)(void *))dlsym(x.lib, "ncclCommDestroy");
            x.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(x.lib, "ncclAllGather");
            x.GroupStart = (int (*)())dlsym(x.lib, "ncclGroupStart");
            x.GroupEnd = (int (*)())dlsym(x.lib, "ncclGroupEnd");
            x.GetErrorString = (const char *(*)(int))dlsym(x.lib, "ncclGetErrorString");
        }
        return x;
    }();
    return r;
}

thread_local int g_last_rccl_error = 0;

} // namespace

struct acgpu_comm {
    std::vector<int> devices, lanes;
    std::vector<hipStream_t> streams;
    std::vector<void *> comms; // ncclComm_t per device (RCCL transport)
    int transport = ACGPU_TRANSPORT_PEER;
    ~acgpu_comm() {
        int cur = -1;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        for (void *c : comms)
            if (c && rccl().CommDestroy) (void)rccl().CommDestroy(c);
        for (size_t i = 0; i < streams.size(); ++i)
            if (streams[i]) {
                (void)hipSetDevice(devices[i]);
                (void)hipStreamDestroy(streams[i]);
            }
        if (have) (void)hipSetDevice(cur);
    }
};

namespace {

// lane of entry i = how many earlier entries name the same device
std::vector<int> lanes_of(const int *devices, int n) {
    std::vector<int> lanes(n, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) lanes[i]++;
    return lanes;
}

int check_devices(const int *devices, int n_devices) {
    if (!devices || n_devices < 1 || n_devices > 64) return ACGPU_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return ACGPU_E_NODEVICE;
    }
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return ACGPU_E_INVALID;
    return ACGPU_OK;
}

struct DeviceRestore {
    int cur = -1;
    bool have = false;
    DeviceRestore() { have = hipGetDevice(&cur) == hipSuccess; }
    ~DeviceRestore() {
        if (have) (void)hipSetDevice(cur);
    }
};

// runs fn(i) for every share on its own host thread (device made current first), waits for all; the first error wins
template <typename F>
int on_all_shares(std::vector<Share> &sh, F fn) {
    std::vector<std::thread> pool;
    auto body = [&](size_t i) {
        Share &s = sh[i];
        if (hipSetDevice(s.device) != hipSuccess) {
            s.hip_error = (int)hipGetLastError();
            s.rc = ACGPU_E_HIP;
            return;
        }
        s.rc = fn(i);
        if (s.rc == ACGPU_E_HIP) s.hip_error = g_last_hip_error;
    };
    try {
        for (size_t i = 1; i < sh.size(); ++i) pool.emplace_back(body, i);
    } catch (...) {
        for (auto &th : pool) th.join();
        return ACGPU_E_NOMEM;
    }
    body(0);
    for (auto &th : pool) th.join();
    for (auto &s : sh)
        if (s.rc != ACGPU_OK) {
            if (s.rc == ACGPU_E_HIP && s.hip_error) g_last_hip_error = s.hip_error; // (acgpu_last_hip_error() reads the CALLER's thread)
            return s.rc;
        }
    return ACGPU_OK;
}

} // namespace

extern "C" {

int acgpu_last_rccl_error(void) { return g_last_rccl_error; }

uint64_t acgpu_gather_slot_bytes(uint64_t gcap, int record_kind) { return (16 + gcap * (uint64_t)record_kind + 15) & ~(uint64_t)15; }

int acgpu_match_u16_multi(const acgpu_automaton *ca, const uint16_t *haystack, uint64_t n_units, const int *devices, int n_devices,
                          int record_kind, void *out, uint64_t cap, uint64_t *n_out) {
    if (!ca || !n_out || (n_units && !haystack) || (cap && !out)) return ACGPU_E_INVALID;
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    if (n_units >= (1ull << 31)) return ACGPU_E_INVALID;
    int rc = check_devices(devices, n_devices);
    if (rc) return rc;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    const HostTables &t = a->t;
    *n_out = 0;
    DeviceRestore restore;
    // the loops that only exist as a sequential kernel over the whole text (word matchers over a table that is not
    // fold-consistent), and texts too short to be worth cutting: one device, the single-device entry
    const bool sequential_only = (t.mode == ACGPU_MODE_WHOLEWORD && !t.fold_consistent) ||
                                 (t.mode == ACGPU_MODE_WWLONGEST && !t.fold_consistent && record_kind == ACGPU_REC_SET);
    // A share has to be worth its fixed costs -- a host thread, a staging ring, two extra synchronisations --: 2^22 units (8 MiB)
    // at least, below that fewer devices or the single-device entry (a 100 KB string under -Dacgpu.devices=0..7 is ONE call on
    // one device).  Tunable multi_min_share: the tests cut texts of a few thousand units.
    const uint64_t min_share = (uint64_t)std::max<int64_t>(8, tunables().multi_min_share);
    const int K = sequential_only ? 1 : (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_devices, n_units / min_share));
    if (K == 1) {
        if ((rc = set_device(devices[0]))) return rc;
        return acgpu_match_u16(ca, haystack, n_units, record_kind, out, cap, n_out);
    }
    const Halo h = halos_for(t);
    const std::vector<int> lanes = lanes_of(devices, n_devices);
    std::vector<Share> sh((size_t)K);
    std::vector<uint64_t> v0((size_t)K), v1((size_t)K), lo((size_t)K), hi((size_t)K);
    for (int i = 0; i < K; ++i) {
        lo[i] = i == 0 ? 0 : ((uint64_t)i * n_units / (uint64_t)K) & ~7ull;
        hi[i] = i == K - 1 ? n_units : ((uint64_t)(i + 1) * n_units / (uint64_t)K) & ~7ull;
        const uint64_t lpad = round_up8(h.left);
        v0[i] = lo[i] > lpad ? lo[i] - lpad : 0; // (lo and the pad are multiples of 8: the view and the owned range start 16-byte aligned)
        v1[i] = std::min<uint64_t>(n_units, hi[i] + h.right);
        sh[i].device = devices[i];
        sh[i].lane = lanes[i];
    }
    std::vector<std::unique_lock<std::mutex>> locks((size_t)K);
    std::vector<hipStream_t> saved((size_t)K, nullptr), own_stream((size_t)K, nullptr);
    // the pools, their locks (taken in (device, lane) order: two multi-device calls on one automaton cannot deadlock) and a
    // stream per share
    std::vector<int> order((size_t)K);
    for (int i = 0; i < K; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return std::make_pair(sh[x].device, sh[x].lane) < std::make_pair(sh[y].device, sh[y].lane); });
    for (int oi = 0; oi < K && rc == ACGPU_OK; ++oi) {
        const int i = order[oi];
        if ((rc = set_device(sh[i].device))) break;
        if ((rc = device_for_call(a, &sh[i].d, sh[i].lane))) break;
        locks[i] = std::unique_lock<std::mutex>(sh[i].d->mu);
        if (sh[i].d->inflight > 0) rc = ACGPU_E_INVALID; // (stream rule: tickets of the asynchronous entry are in flight)
        else if (!sh[i].d->multi_stream && hipStreamCreateWithFlags(&sh[i].d->multi_stream, hipStreamNonBlocking) != hipSuccess) rc = ACGPU_E_HIP;
        else {
            own_stream[i] = sh[i].d->multi_stream; // (the pool's own, created once)
            saved[i] = sh[i].d->call_stream;
            sh[i].d->call_stream = own_stream[i];
            sh[i].stream = own_stream[i];
        }
    }
    auto cleanup = [&]() {
        for (int i = 0; i < K; ++i) {
            if (own_stream[i]) {
                (void)hipSetDevice(sh[i].device);
                (void)hipStreamSynchronize(own_stream[i]);
                sh[i].d->call_stream = saved[i];
            }
        }
    };
    if (rc) {
        cleanup();
        return rc;
    }
    // ---- every share scanned at once, each by its own host thread: pipelined copy of its view + scans of the chunks ----
    rc = on_all_shares(sh, [&](size_t i) -> int {
        Share &s = sh[i];
        uint64_t cap_i = std::max<uint64_t>(cap, 1024);
        for (;;) {
            int64_t chain = 0; // LONGEST / WWLONGEST: the scan enters at the share's first unit; SHORTEST: nothing restricts
            uint64_t n = 0;
            const int r = scan_host_range(a, *s.d, haystack, n_units, v0[i], v1[i], lo[i], hi[i], record_kind, cap_i, &n, &chain);
            if (r == ACGPU_E_OVERFLOW) { // (the caller's capacity only has to hold in the end: a share keeps all its records)
                cap_i = n + n / 8 + 16;
                continue;
            }
            if (r) return r;
            s.d_hay = (const uint16_t *)s.d->stage_hay.p;
            s.n_units = v1[i] - v0[i];
            s.own_begin = lo[i] - v0[i];
            s.own_end = hi[i] - v0[i];
            s.text_begin = v0[i] == 0;
            s.text_end = v1[i] == n_units;
            s.spec = s.d->stage_out.p;
            s.spec_cap = cap_i;
            s.n_spec = n;
            s.spec_exit = chain;
            return ACGPU_OK;
        }
    });
    if (rc == ACGPU_OK && chain_family(t)) rc = repair_chains(a, sh, record_kind);
    uint64_t total = 0;
    if (rc == ACGPU_OK) {
        for (auto &s : sh) total += s.n_final();
        *n_out = total;
        if (total > cap) rc = ACGPU_E_OVERFLOW;
    }
    if (rc == ACGPU_OK && total) {
        // global positions on the device, then every share's pieces straight to their place in the caller's buffer
        std::vector<uint64_t> off((size_t)K, 0);
        for (int i = 1; i < K; ++i) off[i] = off[i - 1] + sh[i - 1].n_final();
        const int cols = record_kind / 4;
        rc = on_all_shares(sh, [&](size_t i) -> int {
            Share &s = sh[i];
            char *dst = (char *)out + off[i] * (uint64_t)record_kind;
            const void *piece[2] = {s.repaired ? s.d->multi_win.p : nullptr, (const char *)s.spec + s.keep_from * (uint64_t)record_kind};
            const uint64_t cnt[2] = {s.repaired ? s.n_win : 0, s.n_spec - s.keep_from};
            for (int k = 0; k < 2; ++k) {
                if (!cnt[k]) continue;
                if (v0[i]) {
                    hipLaunchKernelGGL(k_shift_records, dim3((unsigned)((cnt[k] + 255) / 256)), dim3(256), 0, s.stream,
                                       (int32_t *)const_cast<void *>(piece[k]), cnt[k], cols, (int32_t)v0[i]);
                    HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipMemcpyAsync(dst, piece[k], cnt[k] * (uint64_t)record_kind, hipMemcpyDeviceToHost, s.stream));
                dst += cnt[k] * (uint64_t)record_kind;
            }
            HIP_TRY(hipStreamSynchronize(s.stream));
            return ACGPU_OK;
        });
    }
    cleanup();
    return rc;
}

int acgpu_comm_open(const int *devices, int n_devices, int transport, acgpu_comm **out) {
    if (!out) return ACGPU_E_INVALID;
    *out = nullptr;
    if (transport != ACGPU_TRANSPORT_AUTO && transport != ACGPU_TRANSPORT_RCCL && transport != ACGPU_TRANSPORT_PEER) return ACGPU_E_INVALID;
    int rc = check_devices(devices, n_devices);
    if (rc) return rc;
    DeviceRestore restore;
    acgpu_comm *c = new (std::nothrow) acgpu_comm();
    if (!c) return ACGPU_E_NOMEM;
    c->devices.assign(devices, devices + n_devices);
    c->lanes = lanes_of(devices, n_devices);
    c->streams.assign((size_t)n_devices, nullptr);
    bool distinct = true;
    for (int l : c->lanes) distinct = distinct && l == 0;
    for (int i = 0; i < n_devices; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking) != hipSuccess) {
            g_last_hip_error = (int)hipGetLastError();
            delete c;
            return ACGPU_E_HIP;
        }
    }
    // RCCL needs one rank per device; a device listed twice (tests on a one-GPU box) leaves the peer copies
    if (transport == ACGPU_TRANSPORT_RCCL && !distinct) {
        delete c;
        return ACGPU_E_INVALID;
    }
    if (transport != ACGPU_TRANSPORT_PEER && distinct) {
        Rccl &r = rccl();
        if (r.ok()) {
            c->comms.assign((size_t)n_devices, nullptr);
            const int e = r.CommInitAll(c->comms.data(), n_devices, devices);
            if (e == 0) c->transport = ACGPU_TRANSPORT_RCCL;
            else {
                g_last_rccl_error = e;
                c->comms.clear();
            }
        }
        if (c->transport != ACGPU_TRANSPORT_RCCL && transport == ACGPU_TRANSPORT_RCCL) {
            delete c;
            return ACGPU_E_UNSUPPORTED; // librccl missing or ncclCommInitAll failed (acgpu_last_rccl_error)
        }
    }
    if (c->transport == ACGPU_TRANSPORT_PEER) {
        for (int i = 0; i < n_devices; ++i) { // direct access between the devices of the list, where the platform allows it
            (void)hipSetDevice(devices[i]);
            for (int j = 0; j < n_devices; ++j) {
                int can = 0;
                if (devices[j] != devices[i] && hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(devices[j], 0);
            }
        }
        (void)hipGetLastError(); // (already enabled: not an error)
    }
    *out = c;
    return ACGPU_OK;
}

void acgpu_comm_close(acgpu_comm *c) { delete c; }

int acgpu_comm_transport(const acgpu_comm *c) { return c ? c->transport : ACGPU_E_INVALID; }

void *acgpu_comm_stream(const acgpu_comm *c, int i) { return (c && i >= 0 && (size_t)i < c->streams.size()) ? (void *)c->streams[(size_t)i] : nullptr; }

int acgpu_match_device_allgather(const acgpu_automaton *ca, acgpu_comm *c, acgpu_shard *shards, int record_kind, void *const *d_gather,
                                 uint64_t gcap, uint64_t *counts, acgpu_profile *profs) {
    if (!ca || !c || !shards || !d_gather || !counts) return ACGPU_E_INVALID;
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    acgpu_automaton *a = const_cast<acgpu_automaton *>(ca);
    const HostTables &t = a->t;
    const int K = (int)c->devices.size();
    const uint64_t slot_bytes = acgpu_gather_slot_bytes(gcap, record_kind);
    for (int i = 0; i < K; ++i) {
        if (!d_gather[i] || ((uintptr_t)d_gather[i] & 15)) return ACGPU_E_INVALID;
        counts[i] = 0;
    }
    if (profs) std::memset(profs, 0, sizeof(acgpu_profile) * (size_t)K);
    DeviceRestore restore;
    int rc = ACGPU_OK;
    auto slot_of = [&](int i) { return (char *)d_gather[i] + (uint64_t)i * slot_bytes; };
    bool over = false;
    const bool async_family = t.mode == ACGPU_MODE_ALL || (t.mode == ACGPU_MODE_WHOLEWORD && t.fold_consistent);
    std::vector<DeviceState *> pools((size_t)K, nullptr);
    for (int i = 0; i < K; ++i) {
        if ((rc = set_device(c->devices[i]))) return rc;
        if ((rc = device_for_call(a, &pools[i], c->lanes[i]))) return rc;
    }
    std::vector<acgpu_ticket *> tickets((size_t)K, nullptr);
    std::vector<acgpu_shard> local(shards, shards + K);
    if (async_family) {
        // ---- AhoCorasick / WholeWord: scan -> [header | records] of the device's own slot, enqueued on every device without a
        // host round trip; the header is written by the scan's last kernel in stream order, the gather follows on the same stream
        for (int i = 0; i < K && rc == ACGPU_OK; ++i) {
            if ((rc = set_device(c->devices[i]))) break;
            std::lock_guard<std::mutex> lock(pools[i]->mu);
            local[i].d_result = slot_of(i);
            rc = begin_shard(a, *pools[i], &local[i], record_kind, slot_of(i) + 16, gcap, c->streams[i], profs ? 1 : 0, &tickets[i]);
        }
    } else {
        // ---- the other families end with their count on the host: one host thread per device for the speculative scans,
        // then the repairs share by share, then header + records into the slot
        std::vector<Share> sh((size_t)K);
        std::vector<std::unique_lock<std::mutex>> locks((size_t)K);
        std::vector<int> order((size_t)K);
        for (int i = 0; i < K; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return std::make_pair(c->devices[x], c->lanes[x]) < std::make_pair(c->devices[y], c->lanes[y]); });
        for (int oi = 0; oi < K; ++oi) locks[order[oi]] = std::unique_lock<std::mutex>(pools[order[oi]]->mu);
        for (int i = 0; i < K; ++i) {
            if (pools[i]->inflight > 0) return ACGPU_E_INVALID;
            Share &s = sh[i];
            s.device = c->devices[i];
            s.lane = c->lanes[i];
            s.d = pools[i];
            s.stream = c->streams[i];
            s.d_hay = shards[i].d_hay;
            s.n_units = shards[i].n_units;
            s.own_begin = shards[i].own_begin;
            s.own_end = shards[i].own_end;
            s.text_begin = shards[i].text_begin;
            s.text_end = shards[i].text_end;
        }
        const bool chains = chain_family(t) && K > 1;
        rc = on_all_shares(sh, [&](size_t i) -> int {
            Share &s = sh[i];
            acgpu_shard one = shards[i];
            one.d_result = nullptr;
            // speculation (see repair_chains); share 0 takes the caller's entry
            if (chain_family(t)) one.chain_entry = i == 0 ? shards[0].chain_entry : (t.mode == ACGPU_MODE_SHORTEST ? 0 : (int64_t)s.own_begin);
            uint64_t n = 0;
            int r = match_shard(a, *s.d, &one, record_kind, slot_of((int)i) + 16, gcap, &n, s.stream, profs ? &profs[i] : nullptr);
            s.spec = slot_of((int)i) + 16;
            s.spec_cap = gcap;
            if (r == ACGPU_E_OVERFLOW && chains) { // the repairs need every record: once more, into a private buffer
                if ((r = s.d->stage_out.ensure(n * (uint64_t)record_kind + 16))) return r;
                s.spec = s.d->stage_out.p;
                s.spec_cap = n;
                r = match_shard(a, *s.d, &one, record_kind, s.spec, n, &n, s.stream, nullptr);
            }
            if (r != ACGPU_OK && r != ACGPU_E_OVERFLOW) return r;
            s.n_spec = n;
            s.spec_exit = one.chain_exit;
            s.exit_rel = s.spec_exit - (int64_t)s.own_end;
            return ACGPU_OK;
        });
        if (rc == ACGPU_OK && chains) rc = repair_chains(a, sh, record_kind);
        for (int i = 0; i < K && rc == ACGPU_OK; ++i) {
            Share &s = sh[i];
            counts[i] = s.n_final();
            if (counts[i] > gcap) over = true;
            shards[i].chain_exit = (int64_t)s.own_end + s.exit_rel;
        }
        // (from here to the synchronisation below the slots are filled by copies out of the pools' scratch -- multi_win,
        // multi_tail, stage_out --: the pools stay locked until those copies have RUN, not just been enqueued, or a concurrent
        // call on the same automaton and device could overwrite their sources)
        auto fill_slot = [&](int i) -> int {
            Share &s = sh[i];
            int rc = set_device(s.device);
            if (rc) return rc;
            char *recs = slot_of(i) + 16;
            if (s.repaired) { // window ++ kept tail of the speculation (through a private copy: the ranges overlap)
                const uint64_t n_tail = s.n_spec - s.keep_from, rk = (uint64_t)record_kind;
                if (n_tail) {
                    if ((rc = s.d->multi_tail.ensure(n_tail * rk + 16))) return rc;
                    HIP_TRY(hipMemcpyAsync(s.d->multi_tail.p, (const char *)s.spec + s.keep_from * rk, n_tail * rk, hipMemcpyDeviceToDevice, s.stream));
                }
                if (s.n_win) HIP_TRY(hipMemcpyAsync(recs, s.d->multi_win.p, s.n_win * rk, hipMemcpyDeviceToDevice, s.stream));
                if (n_tail) HIP_TRY(hipMemcpyAsync(recs + s.n_win * rk, s.d->multi_tail.p, n_tail * rk, hipMemcpyDeviceToDevice, s.stream));
            } else if (s.spec != (void *)recs && s.n_spec) {
                HIP_TRY(hipMemcpyAsync(recs, s.spec, s.n_spec * (uint64_t)record_kind, hipMemcpyDeviceToDevice, s.stream));
            }
            HIP_TRY(launch_write_result(reinterpret_cast<acgpu_device_result *>(slot_of(i)), counts[i], s.stream));
            return ACGPU_OK;
        };
        for (int i = 0; i < K && rc == ACGPU_OK && !over; ++i) rc = fill_slot(i);
        for (int i = 0; i < K; ++i) { // (on the error paths too: nothing of this call is left in flight when the locks go)
            (void)hipSetDevice(c->devices[i]);
            if (hipStreamSynchronize(c->streams[i]) != hipSuccess && rc == ACGPU_OK) {
                g_last_hip_error = (int)hipGetLastError();
                rc = ACGPU_E_HIP;
            }
        }
        if (rc) return rc;
    }
    // ---- the gather: every device's slot to every other device, in place ----
    auto gather = [&]() -> int {
        if (c->transport == ACGPU_TRANSPORT_RCCL) {
            Rccl &r = rccl();
            int e = r.GroupStart();
            for (int i = 0; i < K && e == 0; ++i)
                e = r.AllGather(slot_of(i), d_gather[i], (size_t)slot_bytes, /*ncclChar*/ 0, c->comms[(size_t)i], c->streams[(size_t)i]);
            const int e2 = r.GroupEnd();
            if (e == 0) e = e2;
            if (e != 0) {
                g_last_rccl_error = e;
                return ACGPU_E_HIP;
            }
            return ACGPU_OK;
        }
        // peer copies: slot i goes from device i to every other device, on device i's stream
        for (int i = 0; i < K; ++i) {
            const int r = set_device(c->devices[i]);
            if (r) return r;
            for (int j = 0; j < K; ++j) {
                if (j == i) continue;
                HIP_TRY(hipMemcpyPeerAsync((char *)d_gather[j] + (uint64_t)i * slot_bytes, c->devices[j], slot_of(i), c->devices[i], slot_bytes,
                                           c->streams[i]));
            }
        }
        return ACGPU_OK;
    };
    if (rc == ACGPU_OK && !over) rc = gather();
    // ---- collect ----
    int first_err = rc;
    bool any_redone = false;
    for (int i = 0; i < K; ++i) {
        if (tickets[i]) {
            (void)hipSetDevice(c->devices[i]);
            uint64_t n = 0;
            bool redone = false;
            const int r = end_ticket(ca, tickets[i], &n, profs ? &profs[i] : nullptr, &redone);
            counts[i] = n;
            any_redone = any_redone || redone;
            if (r == ACGPU_E_OVERFLOW) over = true;
            else if (r != ACGPU_OK && first_err == ACGPU_OK) first_err = r;
        }
    }
    auto sync_all = [&]() {
        for (int i = 0; i < K; ++i) {
            (void)hipSetDevice(c->devices[i]);
            if (hipStreamSynchronize(c->streams[i]) != hipSuccess && first_err == ACGPU_OK) {
                g_last_hip_error = (int)hipGetLastError();
                first_err = ACGPU_E_HIP;
            }
        }
    };
    sync_all();
    // a scan that had to be redone (a scratch slice had filled up) wrote its slot again BEHIND the gather: once more
    if (first_err == ACGPU_OK && !over && any_redone && K > 1) {
        first_err = gather();
        sync_all();
    }
    if (first_err != ACGPU_OK) return first_err;
    return over ? ACGPU_E_OVERFLOW : ACGPU_OK;
}

} // extern "C"
