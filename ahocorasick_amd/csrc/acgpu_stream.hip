// acgpu_stream.hip -- match(Readable, ReadableMatchListener<T>) (S/StringMap.java:6; S/AhoCorasickMap.java:208-275,
// S/LongestMatchMap.java:203-286, S/WholeWordMatchMap.java:55-153, S/ShortestMatchMap.java:199-291,
// S/WholeWordLongestMatchMap.java:54-181): the haystack arrives in chunks, the stream keeps what a later chunk can still change.
//
// Two forms of acgpu_stream_feed:
//  * synchronous (default): copy in, scan, copy out -- a feed returns the records ITS chunk made decidable;
//  * pipelined (acgpu_stream_set_pipelined): the chunk is copied into a pinned staging buffer by several host threads while the
//    calling thread scans the PREVIOUS chunk (whose transfer the previous feed enqueued) and returns ITS records; when the
//    copy is done the calling thread enqueues ONE transfer of the whole chunk (transfers enqueued piece by piece from
//    several threads made the launches of the scan slow): copy, transfer and scan of neighbouring chunks overlap, a feed
//    costs the slowest of them instead of their sum.  The last feed (final != 0) returns the records
//    of both the previous and its own chunk.  The concatenation over all feeds is the same in both forms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <thread>
#include <vector>

#include "acgpu_host.h"

using namespace acgpu;

namespace {

// records device -> host-mapped pinned memory, coalesced 16-byte pieces (a hipMemcpy of the same 100 KB queues behind the
// chunk transfers that are in flight and takes as long as the scan)
__global__ void k_copy_out(const uint4 *src, uint4 *dst, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// what one feed scans, from lengths alone: the buffer is [carry | chunk] = `total` units, of which [own_begin, own_end) become
// decidable now, and the units from keep_from on are carried to the next feed
struct FeedPlan {
    uint64_t total, own_begin, own_end, keep_from;
};

int scan_mode(const HostTables &t) {
    // what the scan carries between feeds: a WHOLEWORD automaton whose folded keywords hold non-word units is scanned by the
    // WholeWordLongest walk (match_shard), which hands the position of its next word start on
    return (t.mode == ACGPU_MODE_WHOLEWORD && !t.fold_consistent && !t.fold_clean) ? ACGPU_MODE_WWLONGEST : t.mode;
}

FeedPlan plan_feed(const HostTables &t, uint64_t n_carry, uint64_t n_units, uint64_t own_from, uint64_t carry_pos, bool final) {
    const int mode = scan_mode(t);
    FeedPlan p{};
    p.total = n_carry + n_units;
    p.own_begin = own_from - carry_pos;
    p.own_end = p.total;
    if (mode == ACGPU_MODE_ALL || mode == ACGPU_MODE_SHORTEST) {
        const uint64_t halo = t.max_len > 0 ? t.max_len - 1 : 0;
        p.keep_from = p.total > halo ? p.total - halo : 0;
    } else if (mode == ACGPU_MODE_WHOLEWORD || mode == ACGPU_MODE_WWLONGEST) {
        const uint64_t hold = (uint64_t)t.max_len + 1; // a word / walk that starts here may still grow
        if (!final) p.own_end = std::max<uint64_t>(p.own_begin, p.total > hold ? p.total - hold : 0);
        p.keep_from = p.own_end > 0 ? p.own_end - 1 : 0; // one unit of left context
    } else {
        const uint64_t halo = t.max_len > 0 ? t.max_len - 1 : 0;
        if (!final) p.own_end = std::max<uint64_t>(p.own_begin, p.total > halo ? p.total - halo : 0);
        p.keep_from = p.own_end;
    }
    return p;
}

// a few persistent host threads that copy pieces of a chunk (thread start-up costs as much as copying a megabyte)
class CopyPool {
  public:
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    // runs job(i) for i in [0, n) on the workers and the calling thread; returns when all are done
    void run(int n, const std::function<void(int)> &job) {
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = &job;
            next_ = 0;
            n_ = n;
            left_ = n;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return left_ == 0; });
        job_ = nullptr;
    }
    // the same, but the calling thread does something else meanwhile: start(), ..., wait()
    void start(int n, const std::function<void(int)> &job) {
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = &job;
            next_ = 0;
            n_ = n;
            left_ = n;
        }
        cv_.notify_all();
    }
    void wait() {
        work(); // (whatever is left)
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return left_ == 0; });
        job_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            int i;
            const std::function<void(int)> *job;
            {
                std::lock_guard<std::mutex> l(mu_);
                if (!job_ || next_ >= n_) return;
                i = next_++;
                job = job_;
            }
            (*job)(i);
            {
                std::lock_guard<std::mutex> l(mu_);
                if (--left_ == 0) done_.notify_all();
            }
        }
    }
    void loop() {
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || (job_ && next_ < n_); });
                if (stop_) return;
            }
            work();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)> *job_ = nullptr;
    int next_ = 0, n_ = 0, left_ = 0;
    bool stop_ = false;
};

// one chunk on its way: its units in pinned host memory and (soon) in device memory, and what its scan will be
struct Slot {
    bool pending = false; // transferred (or on its way), not yet scanned
    FeedPlan plan{};
    uint64_t carry_pos = 0; // global position of buffer unit 0
    bool final = false;
};

} // namespace

struct acgpu_stream {
    acgpu_automaton *a = nullptr;
    std::vector<uint16_t> carry; // units a later chunk can still change the answer for (context + held back)
    uint64_t carry_pos = 0;      // global position of carry[0]
    uint64_t own_from = 0;       // global position of the first unit no earlier feed has owned
    uint64_t chain_entry = 0;    // LONGEST: global position at which the greedy chain continues
    bool finished = false;
    std::vector<uint16_t> buf;
    // ---- pipelined form ----
    bool pipelined = false, started = false;
    int device = -1;
    Slot slot[2];
    int cur = 0; // the slot the NEXT feed fills
    CopyPool *pool = nullptr;
    // the chunks' staging memory, and the records of one scan: a device buffer, and pinned host memory they are copied back to (a
    // hipMemcpy straight into the caller's pageable memory cost more than the scan of a chunk; kernels writing records into
    // host-mapped memory: 50x more).  Taken from / left to the automaton's cache (StreamBufs, acgpu_host.h).
    StreamBufs b;
    std::vector<char> undelivered;   // records a feed could not hand over (capacity too small): the same feed, called again, gets them
    uint64_t undelivered_n = 0;
    int64_t undelivered_base = 0;
    bool redeliver = false;
    ~acgpu_stream() {
        delete pool;
        if (a) { // (acgpu_free on an automaton with open streams detaches them: a == nullptr, nothing of it is touched here)
            std::lock_guard<std::mutex> l(a->mu);
            a->open_streams.erase(this);
        }
        if (device >= 0) {
            int cur_dev = -1;
            const bool have = hipGetDevice(&cur_dev) == hipSuccess;
            (void)hipSetDevice(device);
            if (b.copy_stream) (void)hipStreamSynchronize(b.copy_stream); // (nothing of this stream stays in flight)
            if (b.device >= 0 && a) { // (the buffers outlive the stream: the next one on this device takes them over)
                std::lock_guard<std::mutex> l(a->mu);
                try {
                    a->stream_cache.push_back(std::move(b));
                    b = StreamBufs();
                } catch (...) {
                }
            }
            if (b.device >= 0) b.release();
            if (have) (void)hipSetDevice(cur_dev);
        }
    }
};

namespace {

// scans the chunk in slot `sl` (its transfer has been enqueued) on the NULL stream and appends its records, relative to
// `base`, to out[n_done ..); *n_out = records so far (beyond cap: a count only)
int scan_slot(acgpu_stream *s, Slot &sl, int record_kind, void *out, uint64_t cap, uint64_t n_done, int64_t base, uint64_t *n_out) {
    acgpu_automaton *a = s->a;
    const HostTables &t = a->t;
    const int mode = scan_mode(t);
    const FeedPlan &p = sl.plan;
    const int si = (int)(&sl - s->slot);
    *n_out = n_done;
    uint64_t chain_exit = mode == ACGPU_MODE_SHORTEST ? s->chain_entry : std::max<uint64_t>(s->chain_entry, sl.carry_pos + p.own_end);
    if (p.own_end > p.own_begin) {
        DeviceState *d = nullptr;
        int rc = device_for_call(a, &d);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(d->mu);
        const bool trace = (tunables().tile_debug & (1ll << 42)) != 0;
        const auto t0 = std::chrono::steady_clock::now();
        auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
        if (trace) {
            HIP_TRY(hipEventSynchronize(s->b.arrived[si]));
            fprintf(stderr, "[scan] chunk arrived after %.0f us", since());
        }
        HIP_TRY(hipStreamWaitEvent(nullptr, s->b.arrived[si], 0));
        // (what the buffer holds already -- NOT a record more, or the buffer would grow by a quarter with every feed)
        uint64_t n = 0, scap = std::max<uint64_t>((s->b.out_dev.bytes > 16 ? s->b.out_dev.bytes - 16 : 0) / (uint64_t)record_kind, 1 << 16);
        acgpu_shard sh{};
        for (;;) { // (the scan keeps ALL its records: the caller's capacity only decides what this feed can hand over)
            if ((rc = s->b.out_dev.ensure(scap * (uint64_t)record_kind + 16))) return rc;
            sh = acgpu_shard{};
            sh.d_hay = (const uint16_t *)s->b.dev[si].p;
            sh.n_units = p.total;
            sh.own_begin = p.own_begin;
            sh.own_end = p.own_end;
            sh.text_begin = sl.carry_pos == 0 ? 1 : 0;
            sh.text_end = sl.final ? 1 : 0;
            sh.chain_entry = (int64_t)(s->chain_entry > sl.carry_pos ? s->chain_entry - sl.carry_pos : 0);
            if (mode != ACGPU_MODE_SHORTEST) sh.chain_entry = std::max<int64_t>(sh.chain_entry, (int64_t)p.own_begin);
            rc = match_shard(a, *d, &sh, record_kind, s->b.out_dev.p, scap, &n, nullptr, nullptr, /*readable=*/true);
            if (rc == ACGPU_E_OVERFLOW) {
                scap = n + n / 8 + 16;
                continue;
            }
            if (rc) return rc;
            break;
        }
        if (trace) fprintf(stderr, ", scanned after %.0f us (%llu records)\n", since(), (unsigned long long)n);
        if (mode == ACGPU_MODE_LONGEST || mode == ACGPU_MODE_WWLONGEST) chain_exit = sl.carry_pos + (uint64_t)sh.chain_exit;
        if (mode == ACGPU_MODE_SHORTEST && n) chain_exit = sl.carry_pos + (uint64_t)sh.chain_exit;
        if (n) {
            // positions relative to `base` (the final feed hands over two chunks' records under one base)
            const int64_t delta = (int64_t)sl.carry_pos - base;
            const size_t bytes = n * (size_t)record_kind;
            char *dst;
            if (n_done + n <= cap) {
                dst = (char *)out + n_done * (size_t)record_kind;
            } else { // does not fit: kept for the call that comes back with the capacity
                try {
                    s->undelivered.resize((n_done + n) * (size_t)record_kind);
                } catch (...) {
                    return ACGPU_E_NOMEM;
                }
                if (n_done && n_done <= cap) std::memcpy(s->undelivered.data(), out, n_done * (size_t)record_kind);
                dst = s->undelivered.data() + n_done * (size_t)record_kind;
            }
            if (s->b.out_pin_bytes < bytes) {
                if (s->b.out_pin) (void)hipHostFree(s->b.out_pin);
                s->b.out_pin = nullptr;
                s->b.out_pin_bytes = 0;
                HIP_TRY(hipHostMalloc(&s->b.out_pin, bytes + bytes / 4 + 4096, hipHostMallocMapped));
                HIP_TRY(hipHostGetDevicePointer(&s->b.out_pin_dev, s->b.out_pin, 0));
                s->b.out_pin_bytes = bytes + bytes / 4 + 4096;
            }
            {
                const uint64_t n16 = (bytes + 15) / 16;
                hipLaunchKernelGGL(k_copy_out, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, nullptr, (const uint4 *)s->b.out_dev.p,
                                   (uint4 *)s->b.out_pin_dev, n16);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(nullptr));
            }
            if (!delta) {
                std::memcpy(dst, s->b.out_pin, bytes);
            } else {
                const int32_t *src = reinterpret_cast<const int32_t *>(s->b.out_pin);
                int32_t *r = reinterpret_cast<int32_t *>(dst);
                const int cols = record_kind / 4;
                for (uint64_t i = 0; i < n; ++i) {
                    r[i * cols] = src[i * cols] + (int32_t)delta;
                    r[i * cols + 1] = src[i * cols + 1] + (int32_t)delta;
                    if (cols == 3) r[i * cols + 2] = src[i * cols + 2];
                }
            }
            *n_out = n_done + n;
        }
    }
    s->chain_entry = chain_exit;
    sl.pending = false;
    return ACGPU_OK;
}

// The staging buffers, the copy stream and the events belong to the device of the FIRST feed (include/acgpu.h: every feed comes
// from a thread whose current device is that one).  A feed from another device -- a worker thread that another call left on a
// different device -- would scan device-A pointers with device-B tables: an error code instead.
int check_feed_device(const acgpu_stream *s) {
    if (s->device < 0) return ACGPU_OK;
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    return cur == s->device ? ACGPU_OK : ACGPU_E_INVALID;
}

// the stream's buffers: those a closed stream on this device left to the automaton, or none yet (they grow on demand)
int adopt_bufs(acgpu_stream *s) {
    if (s->b.device >= 0) return ACGPU_OK;
    if (s->device < 0) HIP_TRY(hipGetDevice(&s->device));
    std::lock_guard<std::mutex> l(s->a->mu);
    for (size_t i = 0; i < s->a->stream_cache.size(); ++i) {
        if (s->a->stream_cache[i].device != s->device) continue;
        s->b = std::move(s->a->stream_cache[i]);
        s->a->stream_cache.erase(s->a->stream_cache.begin() + (ptrdiff_t)i);
        return ACGPU_OK;
    }
    s->b.device = s->device;
    return ACGPU_OK;
}

int feed_pipelined(acgpu_stream *s, const uint16_t *units, uint64_t n_units, int final, int record_kind, void *out, uint64_t cap,
                   uint64_t *n_out, int64_t *base) {
    acgpu_automaton *a = s->a;
    if (!a) return ACGPU_E_INVALID; // (its automaton has been freed)
    const HostTables &t = a->t;
    *n_out = 0;
    { const int drc = check_feed_device(s); if (drc) return drc; }
    if (s->redeliver) { // the same feed again, with the capacity the first call reported: its chunk was consumed then
        *base = s->undelivered_base;
        *n_out = s->undelivered_n;
        if (s->undelivered_n > cap) return ACGPU_E_OVERFLOW;
        if (s->undelivered_n) std::memcpy(out, s->undelivered.data(), s->undelivered_n * (size_t)record_kind);
        s->redeliver = false;
        s->undelivered.clear();
        s->undelivered_n = 0;
        s->finished = final != 0;
        return ACGPU_OK;
    }
    if (!s->started) {
        if (s->device < 0) HIP_TRY(hipGetDevice(&s->device));
        { const int arc = adopt_bufs(s); if (arc) return arc; }
        if (!s->b.copy_stream) { // the transfers at the LOWEST priority: where they run as copy kernels they must not hold the scan's CUs
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_TRY(hipStreamCreateWithPriority(&s->b.copy_stream, hipStreamNonBlocking, lo));
        }
        for (auto &e : s->b.arrived)
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const int workers = (int)std::min<unsigned>(5, std::max(1u, std::thread::hardware_concurrency() / 2));
        s->pool = new (std::nothrow) CopyPool(workers);
        if (!s->pool) return ACGPU_E_NOMEM;
        s->started = true;
    }
    const uint64_t n_carry = s->carry.size();
    // (half the synchronous form's limit: the final feed hands two chunks' records over under one base)
    if (n_carry + n_units >= (1ull << 30)) return ACGPU_E_INVALID;
    Slot &sl = s->slot[s->cur];
    Slot &prev = s->slot[1 - s->cur];
    const int si = s->cur;
    const FeedPlan p = plan_feed(t, n_carry, n_units, s->own_from, s->carry_pos, final != 0);
    int rc;
    if (s->b.pin_bytes[si] < p.total * 2 + 64) {
        if (s->b.pin[si]) (void)hipHostFree(s->b.pin[si]);
        s->b.pin[si] = nullptr;
        s->b.pin_bytes[si] = 0;
        const size_t want = p.total * 2 + p.total / 2 + 4096;
        HIP_TRY(hipHostMalloc(&s->b.pin[si], want, hipHostMallocNumaUser)); // (pages where the copying threads run, not on the device's node: 3x the copy rate)
        s->b.pin_bytes[si] = want;
    }
    if ((rc = s->b.dev[si].ensure(p.total * 2 + 64))) return rc;
    uint16_t *h = reinterpret_cast<uint16_t *>(s->b.pin[si]);
    if (n_carry) std::memcpy(h, s->carry.data(), n_carry * 2);
    // the chunk: pieces of 1 MiB copied by the pool's threads (units == our own staging memory: acgpu_stream_reserve --
    // nothing to copy).  The threads only copy; the ONE transfer of the whole buffer is enqueued by the calling thread behind
    // the previous chunk's scan (transfers enqueued from several threads while the scan's launches go out from this one made
    // every launch wait for the runtime: the scan took 310 us instead of 75)
    const bool in_place = units == h + n_carry;
    const uint64_t piece = 1ull << 19; // units
    const int n_pieces = in_place ? 0 : (int)((n_units + piece - 1) / piece);
    std::atomic<int> copy_rc{ACGPU_OK};
    std::function<void(int)> job = [&](int i) {
        const uint64_t o = (uint64_t)i * piece, len = std::min<uint64_t>(piece, n_units - o);
        std::memcpy(h + n_carry + o, units + o, len * 2);
    };
    const bool trace = (tunables().tile_debug & (1ll << 42)) != 0; // development: where a feed's time goes (stderr)
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
    if (n_pieces) s->pool->start(n_pieces, job);
    const double t_started = since();
    // meanwhile: the previous chunk's scan, whose records this feed returns
    *base = prev.pending ? (int64_t)prev.carry_pos : (int64_t)s->carry_pos;
    int scan_rc = ACGPU_OK;
    uint64_t n_done = 0;
    if (prev.pending) scan_rc = scan_slot(s, prev, record_kind, out, cap, 0, *base, &n_done);
    const double t_scanned = since();
    if (n_pieces) s->pool->wait();
    if (trace) fprintf(stderr, "[feed] %llu units: copies started %.0f us, previous chunk scanned %.0f us, copies done %.0f us\n",
                       (unsigned long long)n_units, t_started, t_scanned, since());
    if (scan_rc == ACGPU_OK && copy_rc.load() != ACGPU_OK) scan_rc = copy_rc.load();
    if (scan_rc != ACGPU_OK) {
        (void)hipStreamSynchronize(s->b.copy_stream);
        s->finished = true; // (a stream that failed half way cannot go on)
        return scan_rc;
    }
    if (p.total) HIP_TRY(hipMemcpyAsync(s->b.dev[si].p, s->b.pin[si], p.total * 2, hipMemcpyHostToDevice, s->b.copy_stream));
    HIP_TRY(hipEventRecord(s->b.arrived[si], s->b.copy_stream));
    // commit this chunk
    sl.plan = p;
    sl.carry_pos = s->carry_pos;
    sl.final = final != 0;
    sl.pending = true;
    s->own_from = s->carry_pos + p.own_end;
    try {
        s->carry.assign(h + p.keep_from, h + p.total);
    } catch (...) {
        return ACGPU_E_NOMEM;
    }
    s->carry_pos += p.keep_from;
    s->cur = 1 - s->cur;
    if (final) { // nothing comes after it: its own records as well
        rc = scan_slot(s, sl, record_kind, out, cap, n_done, *base, &n_done);
        if (rc) {
            s->finished = true;
            return rc;
        }
    }
    *n_out = n_done;
    if (n_done > cap) { // the chunk HAS been consumed: the same feed, called again with this capacity, receives the records
        s->undelivered_n = n_done;
        s->undelivered_base = *base;
        s->redeliver = true;
        return ACGPU_E_OVERFLOW;
    }
    s->finished = final != 0;
    return ACGPU_OK;
}

} // namespace

extern "C" {

int acgpu_stream_open(const acgpu_automaton *a, acgpu_stream **out) {
    if (!a || !out) return ACGPU_E_INVALID;
    *out = nullptr;
    // (word-character tables that are not fold-consistent: the reference's Readable loops fold in EVERY lookup,
    // S/WholeWordMatchMap.java:112,117,328, S/WholeWordLongestMatchMap.java:404 -- ordinary scans over word o lower,
    // match_shard(..., readable))
    acgpu_stream *s = new (std::nothrow) acgpu_stream();
    if (!s) return ACGPU_E_NOMEM;
    s->a = const_cast<acgpu_automaton *>(a);
    try {
        std::lock_guard<std::mutex> l(s->a->mu);
        s->a->open_streams.insert(s);
    } catch (...) {
        s->a = nullptr;
        delete s;
        return ACGPU_E_NOMEM;
    }
    *out = s;
    return ACGPU_OK;
}

void acgpu_stream_close(acgpu_stream *s) { delete s; }

// acgpu_free with this stream still open (the caller holds the automaton's mutex): nothing of the stream refers to it any more;
// the staging buffers stay the stream's own and go when it is closed
void acgpu_stream_detach(acgpu_stream *s) { s->a = nullptr; }

int acgpu_stream_set_pipelined(acgpu_stream *s, int on) {
    if (!s || s->started || s->carry_pos != 0 || !s->carry.empty() || s->own_from != 0) return ACGPU_E_INVALID; // before the first feed
    s->pipelined = on != 0;
    return ACGPU_OK;
}

int acgpu_stream_reserve(acgpu_stream *s, uint64_t n_units, uint16_t **buf) {
    if (!s || !buf || !s->pipelined || s->finished || s->redeliver) return ACGPU_E_INVALID;
    *buf = nullptr;
    const uint64_t n_carry = s->carry.size();
    if (n_carry + n_units >= (1ull << 30) || !s->a) return ACGPU_E_INVALID;
    { const int drc = check_feed_device(s); if (drc) return drc; }
    const int si = s->cur;
    { const int arc = adopt_bufs(s); if (arc) return arc; }
    if (s->b.pin_bytes[si] < (n_carry + n_units) * 2 + 64) {
        if (s->b.pin[si]) (void)hipHostFree(s->b.pin[si]);
        s->b.pin[si] = nullptr;
        s->b.pin_bytes[si] = 0;
        const size_t want = (n_carry + n_units) * 2 + (n_carry + n_units) / 2 + 4096;
        HIP_TRY(hipHostMalloc(&s->b.pin[si], want, hipHostMallocNumaUser)); // (pages where the copying threads run, not on the device's node: 3x the copy rate)
        s->b.pin_bytes[si] = want;
    }
    *buf = reinterpret_cast<uint16_t *>(s->b.pin[si]) + n_carry;
    return ACGPU_OK;
}

int acgpu_stream_feed(acgpu_stream *s, const uint16_t *units, uint64_t n_units, int final, int record_kind, void *out,
                      uint64_t cap, uint64_t *n_out, int64_t *base) {
    if (!s || !n_out || !base || (n_units && !units) || (cap && !out) || s->finished) return ACGPU_E_INVALID;
    if (record_kind != ACGPU_REC_SET && record_kind != ACGPU_REC_MAP) return ACGPU_E_INVALID;
    if (!s->a) return ACGPU_E_INVALID; // (its automaton has been freed)
    if (s->pipelined) return feed_pipelined(s, units, n_units, final, record_kind, out, cap, n_out, base);
    acgpu_automaton *a = s->a;
    const HostTables &t = a->t;
    const int mode = scan_mode(t);
    const FeedPlan p = plan_feed(t, s->carry.size(), n_units, s->own_from, s->carry_pos, final != 0);
    const uint64_t total = p.total, own_begin = p.own_begin, own_end = p.own_end, keep_from = p.keep_from;
    if (total >= (1ull << 31)) return ACGPU_E_INVALID;
    *n_out = 0;
    *base = (int64_t)s->carry_pos;
    try {
        s->buf.resize(total);
    } catch (...) {
        return ACGPU_E_NOMEM;
    }
    if (!s->carry.empty()) std::memcpy(s->buf.data(), s->carry.data(), s->carry.size() * 2);
    if (n_units) std::memcpy(s->buf.data() + s->carry.size(), units, n_units * 2);
    uint64_t chain_exit = mode == ACGPU_MODE_SHORTEST ? s->chain_entry : std::max<uint64_t>(s->chain_entry, s->carry_pos + own_end);
    if (own_end > own_begin) {
        DeviceState *d = nullptr;
        int rc = device_for_call(a, &d);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(d->mu); // staging buffers are part of the per-device scratch pool
        if ((rc = d->stage_hay.ensure(total * 2 + 16))) return rc;
        if ((rc = d->stage_out.ensure(cap * (uint64_t)record_kind + 16))) return rc;
        HIP_TRY(hipMemcpy(d->stage_hay.p, s->buf.data(), total * 2, hipMemcpyHostToDevice));
        acgpu_shard sh{};
        sh.d_hay = (const uint16_t *)d->stage_hay.p;
        sh.n_units = total;
        sh.own_begin = own_begin;
        sh.own_end = own_end;
        sh.text_begin = s->carry_pos == 0 ? 1 : 0;
        sh.text_end = final ? 1 : 0;
        sh.chain_entry = (int64_t)(s->chain_entry > s->carry_pos ? s->chain_entry - s->carry_pos : 0);
        if (mode != ACGPU_MODE_SHORTEST) sh.chain_entry = std::max<int64_t>(sh.chain_entry, (int64_t)own_begin);
        rc = match_shard(a, *d, &sh, record_kind, d->stage_out.p, cap, n_out, nullptr, nullptr, /*readable=*/true);
        if (rc != ACGPU_OK) return rc; // ACGPU_E_OVERFLOW: nothing consumed, *n_out = capacity to retry with
        if (*n_out) HIP_TRY(hipMemcpy(out, d->stage_out.p, *n_out * (uint64_t)record_kind, hipMemcpyDeviceToHost));
        if (mode == ACGPU_MODE_LONGEST || mode == ACGPU_MODE_WWLONGEST) chain_exit = s->carry_pos + (uint64_t)sh.chain_exit;
        // SHORTEST: the last restart; an exit equal to the relative entry means "no match in this feed"
        if (mode == ACGPU_MODE_SHORTEST && *n_out) chain_exit = s->carry_pos + (uint64_t)sh.chain_exit;
    }
    // commit
    s->own_from = s->carry_pos + own_end;
    s->chain_entry = chain_exit;
    s->carry.assign(s->buf.begin() + (ptrdiff_t)keep_from, s->buf.end());
    s->carry_pos += keep_from;
    s->finished = final != 0;
    return ACGPU_OK;
}

} // extern "C"
