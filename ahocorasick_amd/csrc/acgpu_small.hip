// acgpu_small.hip -- ONE launch for a short haystack (acgpu_match_u16 on up to kSmallMaxUnits units).
//
// The reference's only published workload is one match() call on one paragraph (R/README.md:126-152: 3.6 us per call for
// AhoCorasickSet against 235 886 words).  The general entry pays two copies, two or more launches and a synchronisation for
// such a call (about 140 us); here the haystack is read from, and the records are written to, host-mapped pinned memory
// by ONE workgroup in ONE launch, and the host learns of the end from a flag in that memory -- no hipMemcpy, no second
// launch, no stream synchronisation.
//
// One workgroup, every position its own walk: from position p the keyword trie is followed through the hashed goto edges
// (built for every automaton: acgpu_build.cpp 5) for as long as an edge exists -- every terminal node on the way is an
// occurrence that STARTS at p.  What the families make of the occurrences:
//   ALL       every one of them, ordered by (end ascending, start ascending) = the reference's call order
//             (S/AhoCorasickSet.java:522-535): a counting sort by end in LDS, ranks inside an end by start;
//   SHORTEST  the greedy selection over that list (an occurrence is reported iff it starts at or behind the end of the
//             previously reported one, S/ShortestMatchSet.java:193-262; closed form of DESIGN.md 4.5), by one lane;
//   LONGEST   L[p] = the deepest terminal of p's walk; the greedy chain p -> p + max(L[p], 1) from 0 is marked by pointer
//             doubling in LDS (S/LongestMatchSet.java:192-265 = T/LongestMatchTest.java:30-42);
//   WHOLEWORD (fold-consistent tables) a walk from every run start; it reports the run iff it ends exactly where the run
//             does, on a terminal node (S/WholeWordMatchMap.java:155-240).
// Anything this form cannot hold -- more than kSmallMaxRecs occurrences, keywords beyond kSmallMaxLen units, the families
// and tables without a kernel here -- is reported back (status 2) and takes the general path.
#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_host.h"
#include "acgpu_small.h"

namespace acgpu {

namespace {

constexpr int kSmallBlock = 1024;
constexpr int kWalks = kSmallMaxUnits / kSmallBlock; // walks per lane

struct SmallRec {
    int32_t start, end, id;
};

__device__ __forceinline__ void small_store(void *out, int record_kind, uint32_t at, const SmallRec &r) {
    if (record_kind == ACGPU_REC_SET) {
        reinterpret_cast<int2 *>(out)[at] = make_int2(r.start, r.end);
    } else {
        int32_t *o = reinterpret_cast<int32_t *>(out) + (size_t)at * 3;
        o[0] = r.start;
        o[1] = r.end;
        o[2] = r.id;
    }
}

// exclusive prefix sum over v[0..n) in place (n <= kSmallMaxUnits + 1), by the whole workgroup; returns the total
__device__ uint32_t block_exclusive_scan(uint32_t *v, uint32_t n, uint32_t *wave_sums) {
    constexpr int kPer = (kSmallMaxUnits + kSmallBlock) / kSmallBlock; // 5 elements per lane cover 4097
    const uint32_t t = threadIdx.x, base = t * kPer;
    uint32_t x[kPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        x[i] = base + i < n ? v[base + i] : 0u;
        sum += x[i];
    }
    const uint32_t incl = wave_inclusive_scan(sum);
    if ((t & 63u) == 63u) wave_sums[t >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < kSmallBlock / 64; ++w) {
        const uint32_t s = wave_sums[w];
        if (w < (t >> 6)) before += s;
        total += s;
    }
    uint32_t run = before + incl - sum;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        if (base + i < n) v[base + i] = run;
        run += x[i];
    }
    __syncthreads();
    return total;
}

// MODE: the automaton's family (ACGPU_MODE_ALL, _SHORTEST, _LONGEST, _WHOLEWORD)
template <int MODE>
__global__ __launch_bounds__(kSmallBlock) void k_small(DevTables T, SmallCall C) {
    __shared__ uint16_t units[kSmallMaxUnits + 8]; // folded units
    __shared__ uint8_t wordf[kSmallMaxUnits + 8];   // WHOLEWORD: word-character flag of every unit (one more behind the end: 0)
    __shared__ uint32_t hist[kSmallMaxUnits + 8];   // ALL: records per end -> offsets; LONGEST: jump table; flags for the compaction
    __shared__ uint32_t aux[kSmallMaxUnits + 8];    // ALL: cursors per end; LONGEST: marks; WHOLEWORD / LONGEST: keyword ids
    __shared__ SmallRec recs[kSmallMaxRecs];        // ALL: occurrences as found; LONGEST / WHOLEWORD: per position {len, id}
    __shared__ SmallRec sorted[MODE == ACGPU_MODE_ALL || MODE == ACGPU_MODE_SHORTEST ? kSmallMaxRecs : 1];
    __shared__ uint32_t wave_sums[kSmallBlock / 64];
    __shared__ uint32_t n_found;
    const uint32_t n = C.n_units, t = threadIdx.x;
    volatile unsigned long long *status = C.status;

    // ---- the haystack: host-mapped memory -> folded units in LDS (one 8-byte load per lane) ----
    if (t == 0) n_found = 0;
    for (uint32_t i = t; i < kSmallMaxUnits + 8; i += kSmallBlock) {
        hist[i] = 0;
        aux[i] = 0;
    }
    if (t < 8) { // (behind the last unit: no unit, no word character)
        units[kSmallMaxUnits + t] = 0;
        wordf[kSmallMaxUnits + t] = 0;
    }
    {
        const uint32_t p0 = t * 4;
        uint32_t a = 0, b = 0;
        if (p0 < n) {
            const uint2 w = reinterpret_cast<const uint2 *>(C.hay)[t]; // (the staging buffer is padded: whole 8-byte groups exist)
            a = w.x;
            b = w.y;
        }
        uint32_t u[4] = {a & 0xffffu, a >> 16, b & 0xffffu, b >> 16};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t f = u[i], wf = 0;
            if (p0 + i < n) {
                if (!T.cs) f = T.lower[u[i]];
                if (MODE == ACGPU_MODE_WHOLEWORD) wf = T.wflags[u[i]] & 1u;
            } else {
                f = 0;
            }
            units[p0 + i] = (uint16_t)f;
            wordf[p0 + i] = (uint8_t)wf;
        }
    }
    __syncthreads();

    // ---- the walks: position p = t + k * kSmallBlock, kWalks of them per lane, their probes in flight together ----
    uint32_t node[kWalks], pos[kWalks], slot[kWalks], best[kWalks], best_id[kWalks];
    bool alive[kWalks];
#pragma unroll
    for (int k = 0; k < kWalks; ++k) {
        const uint32_t p = t + k * kSmallBlock;
        node[k] = 0;
        pos[k] = p;
        best[k] = 0;
        best_id[k] = ~0u;
        alive[k] = p < n;
        if (MODE == ACGPU_MODE_WHOLEWORD) alive[k] = alive[k] && wordf[p] && !(p > 0 && wordf[p - 1]);
        slot[k] = alive[k] ? edge_hash(edge_key(0, units[p])) & T.hmask : 0u;
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < kWalks; ++k) any = any || alive[k];
        if (!any) break;
        uint64_t key[kWalks];
        uint32_t val[kWalks];
#pragma unroll
        for (int k = 0; k < kWalks; ++k) {
            key[k] = alive[k] ? T.hkeys[slot[k]] : kEmptyKey;
            val[k] = alive[k] ? T.hvals[slot[k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kWalks; ++k) {
            if (!alive[k]) continue;
            const uint32_t p = t + k * kSmallBlock;
            const uint64_t want = edge_key(node[k], units[pos[k]]);
            if (key[k] == want) { // the edge exists: one unit further
                node[k] = val[k];
                ++pos[k];
                const uint32_t depth = pos[k] - p;
                if (node[k] >= T.first_out) { // has an output (own or inherited): is it its own?
                    const uint32_t id = T.term_id[node[k]];
                    if (id != ~0u) {
                        if (MODE == ACGPU_MODE_ALL || MODE == ACGPU_MODE_SHORTEST) {
                            const uint32_t at = atomicAdd(&n_found, 1u);
                            if (at < kSmallMaxRecs) recs[at] = SmallRec{(int32_t)p, (int32_t)pos[k], (int32_t)id};
                            atomicAdd(&hist[pos[k]], 1u);
                        } else {
                            best[k] = depth;
                            best_id[k] = id;
                        }
                    }
                }
                if (pos[k] >= n) alive[k] = false;
                else slot[k] = edge_hash(edge_key(node[k], units[pos[k]])) & T.hmask;
            } else if (key[k] == kEmptyKey) { // no such edge: the walk is over
                alive[k] = false;
            } else {
                slot[k] = (slot[k] + 1) & T.hmask; // linear probing
            }
        }
    }
    if (MODE == ACGPU_MODE_WHOLEWORD) { // the run must end where the walk stopped, on the terminal the walk last saw
#pragma unroll
        for (int k = 0; k < kWalks; ++k) {
            // (branch free: every lane reads the flag behind its walk -- position n and beyond hold zeros)
            const uint32_t p = t + k * kSmallBlock;
            const uint32_t behind = wordf[min(pos[k], kSmallMaxUnits)];
            best[k] = (best[k] == pos[k] - p && behind == 0u) ? best[k] : 0u;
        }
    }
    __syncthreads();

    uint32_t n_out = 0;
    if (MODE == ACGPU_MODE_ALL || MODE == ACGPU_MODE_SHORTEST) {
        const uint32_t m = n_found;
        if (m > kSmallMaxRecs) {
            if (t == 0) {
                status[1] = m;
                __threadfence_system();
                status[0] = 2; // not this kernel's: the general path
            }
            return;
        }
        // counting sort by end; inside an end the smaller start (the longer keyword) first
        block_exclusive_scan(hist, n + 1, wave_sums);
        for (uint32_t i = t; i < m; i += kSmallBlock) {
            const SmallRec r = recs[i];
            sorted[hist[r.end] + atomicAdd(&aux[r.end], 1u)] = r;
        }
        __syncthreads();
        for (uint32_t i = t; i < m; i += kSmallBlock) { // (recs is free again: the ordered list goes there)
            const SmallRec r = sorted[i];
            const uint32_t b0 = hist[r.end], b1 = b0 + aux[r.end];
            uint32_t rank = 0;
            for (uint32_t q = b0; q < b1; ++q) rank += sorted[q].start < r.start ? 1u : 0u;
            recs[b0 + rank] = r;
        }
        __syncthreads();
        if (MODE == ACGPU_MODE_ALL) {
            n_out = m;
            if (m <= C.cap)
                for (uint32_t i = t; i < m; i += kSmallBlock) small_store(C.out, C.record_kind, i, recs[i]);
        } else {
            if (t == 0) { // the greedy selection is a chain through the list: one lane
                uint32_t cnt = 0;
                int32_t last_end = 0;
                for (uint32_t i = 0; i < m; ++i) {
                    const SmallRec r = recs[i];
                    if (r.start >= last_end) {
                        if (cnt < C.cap) small_store(C.out, C.record_kind, cnt, r);
                        ++cnt;
                        last_end = r.end;
                    }
                }
                n_found = cnt;
            }
            __syncthreads();
            n_out = n_found;
        }
    } else {
        // per position: the record it would report (LONGEST: if the chain visits it)
        uint32_t *jump = hist, *mark = aux;
#pragma unroll
        for (int k = 0; k < kWalks; ++k) {
            const uint32_t p = t + k * kSmallBlock;
            if (p < n) recs[p] = SmallRec{(int32_t)p, (int32_t)(p + best[k]), (int32_t)best_id[k]};
        }
        if (MODE == ACGPU_MODE_LONGEST) {
#pragma unroll
            for (int k = 0; k <= kWalks; ++k) {
                const uint32_t p = t + k * kSmallBlock;
                if (p > n) continue;
                const uint32_t L = k < kWalks ? best[k < kWalks ? k : 0] : 0u;
                jump[p] = p < n ? min(p + max(L, 1u), n) : n;
                mark[p] = p == 0 ? 1u : 0u;
            }
            __syncthreads();
            // pointer doubling: round r marks the 2^r-th successors of everything marked so far, then squares the jump table
            for (uint32_t span = 1; span < n; span <<= 1) {
                uint32_t nj[kWalks + 1], tgt[kWalks + 1];
                bool mk[kWalks + 1];
#pragma unroll
                for (int k = 0; k <= kWalks; ++k) {
                    const uint32_t p = t + k * kSmallBlock;
                    mk[k] = p <= n && mark[p] != 0;
                    tgt[k] = p <= n ? jump[p] : n;
                    nj[k] = p <= n ? jump[tgt[k]] : n;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k <= kWalks; ++k) {
                    const uint32_t p = t + k * kSmallBlock;
                    if (p <= n) {
                        if (mk[k]) mark[tgt[k]] = 1u;
                        jump[p] = nj[k];
                    }
                }
                __syncthreads();
            }
        }
        // compaction in position order
        uint32_t *flag = MODE == ACGPU_MODE_LONGEST ? hist : aux; // (the jump table is no longer needed)
        uint32_t keep[kWalks];
#pragma unroll
        for (int k = 0; k < kWalks; ++k) {
            const uint32_t p = t + k * kSmallBlock;
            keep[k] = p < n && best[k] != 0 && (MODE != ACGPU_MODE_LONGEST || mark[p] != 0) ? 1u : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kWalks; ++k) {
            const uint32_t p = t + k * kSmallBlock;
            if (p <= n) flag[p] = p < n ? keep[k] : 0u;
        }
        if (t == 0) flag[n] = 0;
        __syncthreads();
        n_out = block_exclusive_scan(flag, n + 1, wave_sums);
        if (n_out <= C.cap) {
#pragma unroll
            for (int k = 0; k < kWalks; ++k) {
                const uint32_t p = t + k * kSmallBlock;
                if (keep[k]) small_store(C.out, C.record_kind, flag[p], recs[p]);
            }
        }
    }
    // ---- the end: count, then -- behind a system-scope fence -- the flag the host polls ----
    __threadfence_system();
    __syncthreads();
    if (t == 0) {
        status[1] = n_out;
        __threadfence_system();
        status[0] = 1;
    }
}

} // namespace

bool small_call_supported(const HostTables &t) {
    if (t.max_len > kSmallMaxLen || t.n_states <= 1) return false;
    switch (t.mode) {
    case ACGPU_MODE_ALL:
    case ACGPU_MODE_SHORTEST:
    case ACGPU_MODE_LONGEST: return true;
    case ACGPU_MODE_WHOLEWORD: return t.fold_consistent && t.fold_clean;
    default: return false;
    }
}

hipError_t launch_small(const DevTables &T, int mode, const SmallCall &c, hipStream_t stream) {
    switch (mode) {
    case ACGPU_MODE_ALL: hipLaunchKernelGGL(k_small<ACGPU_MODE_ALL>, dim3(1), dim3(kSmallBlock), 0, stream, T, c); break;
    case ACGPU_MODE_SHORTEST: hipLaunchKernelGGL(k_small<ACGPU_MODE_SHORTEST>, dim3(1), dim3(kSmallBlock), 0, stream, T, c); break;
    case ACGPU_MODE_LONGEST: hipLaunchKernelGGL(k_small<ACGPU_MODE_LONGEST>, dim3(1), dim3(kSmallBlock), 0, stream, T, c); break;
    case ACGPU_MODE_WHOLEWORD: hipLaunchKernelGGL(k_small<ACGPU_MODE_WHOLEWORD>, dim3(1), dim3(kSmallBlock), 0, stream, T, c); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace acgpu
