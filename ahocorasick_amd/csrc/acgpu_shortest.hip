// acgpu_shortest.hip -- ShortestMatchSet/Map on gfx950.
//
// The reference (S/ShortestMatchSet.java:193-262) reports a match as soon as ANY keyword ends and restarts at the root
// with the next unit, so an occurrence is reported iff it starts at or after the end of the previously reported one,
// taking occurrences by increasing end and, at equal end, the longest first (oracle/brute.py `shortest`, fuzz-checked
// against the literal restatement).  That is a greedy selection over the all-matches list the AhoCorasick pipeline
// already delivers in exactly that order (acgpu_tile.hip / acgpu_kernels.hip):
//
//   k_short_next : nxt[k] = first record after k that starts at or after record k's end.  Records are grouped by end
//                  with starts ascending inside a group, so the first candidate of a group is found by looking at the
//                  group's last record and a binary search; groups are located by binary searches on the end values.
//                  Entry M+... the virtual record "-1" (end = the shard's chain entry) gives the first selected record.
//   k_short_round: the selected records are the chain k0, nxt[k0], nxt[nxt[k0]], ...: marked by pointer doubling --
//                  round t marks the 2^t-th successors of everything marked so far, then squares the jump table.
//   k_short_emit : prefix sum over the marks (acgpu_kernels.hip), ordered scatter of the selected records.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "acgpu_kernels.h"

namespace acgpu {

namespace {

// first index in [lo, hi) whose END is greater than e (records: 3 int32 each, ends ascending)
__device__ __forceinline__ uint32_t upper_end(const int32_t *recs, uint32_t lo, uint32_t hi, int32_t e) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (recs[3 * (uint64_t)mid + 1] <= e) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// first record at index >= from that starts at or after `e`, or M
__device__ __forceinline__ uint32_t first_starting_at(const int32_t *recs, uint32_t M, uint32_t from, int32_t e) {
    uint32_t j = from;
    while (j < M) {
        const int32_t ge_end = recs[3 * (uint64_t)j + 1];
        const uint32_t gl = upper_end(recs, j, M, ge_end) - 1; // last record of j's end group (the shortest keyword)
        if (recs[3 * (uint64_t)gl] >= e) {
            uint32_t lo = j, hi = gl; // starts ascend inside the group: first one >= e
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (recs[3 * (uint64_t)mid] >= e) hi = mid;
                else lo = mid + 1;
            }
            return lo;
        }
        j = gl + 1;
    }
    return M;
}

__global__ __launch_bounds__(256) void k_short_next(const int32_t *recs, uint32_t M, int32_t entry, uint32_t *nxt,
                                                    uint32_t *mark) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    if (k == M) { // sentinel + the chain's first record
        nxt[M] = M;
        mark[M] = 0;
        const uint32_t k0 = first_starting_at(recs, M, upper_end(recs, 0, M, entry), entry); // end <= entry cannot start there
        if (k0 < M) atomicExch(&mark[k0], 1u);
        return;
    }
    const int32_t e = recs[3 * (uint64_t)k + 1];
    nxt[k] = first_starting_at(recs, M, upper_end(recs, k + 1, M, e), e);
}

// Longest (leftmost-longest, non-overlapping) over the same end-ordered list: the record the greedy chain takes at
// position `pos` is the one with the smallest start >= pos (and < limit), the longest among those -- found by scanning
// forwards from the first record that ends after pos until no later record can start that early (a record starts at most
// max_len units before its end).
__device__ __forceinline__ uint32_t leftmost_longest_from(const int32_t *recs, uint32_t M, uint32_t from, int32_t pos,
                                                          int32_t limit, int32_t max_len) {
    uint32_t cur = M;
    int32_t cur_s = 0x7fffffff;
    for (uint32_t j = from; j < M; ++j) {
        const int32_t s = recs[3 * (uint64_t)j], e = recs[3 * (uint64_t)j + 1];
        if (cur != M && e > cur_s + max_len) break;
        if (s >= pos && s < limit && s <= cur_s) { // same start, later in the list = longer
            cur = j;
            cur_s = s;
        }
    }
    return cur;
}

__global__ __launch_bounds__(256) void k_long_next(const int32_t *recs, uint32_t M, int32_t entry, int32_t limit,
                                                   int32_t max_len, uint32_t *nxt, uint32_t *mark) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    if (k == M) {
        nxt[M] = M;
        mark[M] = 0;
        const uint32_t k0 = leftmost_longest_from(recs, M, upper_end(recs, 0, M, entry), entry, limit, max_len);
        if (k0 < M) atomicExch(&mark[k0], 1u);
        return;
    }
    const int32_t e = recs[3 * (uint64_t)k + 1];
    nxt[k] = leftmost_longest_from(recs, M, upper_end(recs, k + 1, M, e), e, limit, max_len);
}

__global__ __launch_bounds__(256) void k_short_clear(uint32_t *mark, uint32_t M) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < M) mark[k] = 0;
}

// One launch = kRoundLog doubling rounds: with the jump table G = nxt^(8^t), a marked element marks G(x), G^2(x), ..., G^7(x) and the
// table becomes G^8 -- if everything at chain distance < 8^t from the head was marked, everything < 8^(t+1) is afterwards.  (The
// rounds are launch-bound: 21 launches of a few microseconds each for 1.3 M records; 7 now.)
constexpr int kRoundLog = 3;
__global__ __launch_bounds__(256) void k_short_round(const uint32_t *jump_in, uint32_t *jump_out, uint32_t *mark, uint32_t M) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    // a record marked earlier in this same launch is on the chain too, so are its successors: the race is benign
    const bool on_chain = k < M && mark[k];
    uint32_t p = k;
#pragma unroll
    for (int i = 1; i < (1 << kRoundLog); ++i) {
        p = jump_in[p]; // (jump_in[M] = M: a chain that has left the list stays there)
        if (on_chain && p < M) mark[p] = 1u;
    }
    jump_out[k] = jump_in[p];
}

template <int REC>
__global__ __launch_bounds__(256) void k_short_emit(const int32_t *recs, uint32_t M, const uint32_t *mark,
                                                    const uint64_t *offsets, const uint64_t *total, void *out, uint64_t cap,
                                                    int64_t entry, unsigned long long *exit_pos) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0 && *total == 0) *exit_pos = (unsigned long long)entry;
    if (k >= M || !mark[k]) return;
    const uint64_t dst = offsets[k];
    const int32_t s = recs[3 * (uint64_t)k], e = recs[3 * (uint64_t)k + 1], id = recs[3 * (uint64_t)k + 2];
    if (dst + 1 == *total) *exit_pos = (unsigned long long)e; // matching restarts at the end of the last reported match
    if (dst >= cap) return;
    if (REC == ACGPU_REC_SET) {
        reinterpret_cast<int2 *>(out)[dst] = make_int2(s, e);
    } else {
        int32_t *o = reinterpret_cast<int32_t *>(out) + dst * 3;
        o[0] = s; o[1] = e; o[2] = id;
    }
}

} // namespace

// d_nxt, d_tmp: M+1 uint32 each; d_mark: M+1 uint32.  After the call d_nxt holds every record's successor and d_mark[k] = 1
// for the head of the chain; marking the chain is the caller's next step (acgpu_api.hip: mark_chain).
hipError_t launch_shortest_select(const int32_t *d_recs, uint32_t M, int64_t entry, uint32_t *d_nxt, uint32_t *d_tmp,
                                  uint32_t *d_mark, hipStream_t stream) {
    if (M == 0) return hipSuccess;
    const dim3 block(256), grid((M + 1 + 255) / 256);
    const int32_t e32 = (int32_t)std::min<int64_t>(std::max<int64_t>(entry, 0), 0x7fffffff);
    hipLaunchKernelGGL(k_short_clear, grid, block, 0, stream, d_mark, M);
    hipLaunchKernelGGL(k_short_next, grid, block, 0, stream, d_recs, M, e32, d_nxt, d_mark);
    (void)d_tmp; // (the chain is marked by the caller: acgpu_api.hip mark_chain)
    return hipGetLastError();
}

// Longest over the all-matches list: see k_long_next.  limit = own_end (matches must start before it).
hipError_t launch_longest_select(const int32_t *d_recs, uint32_t M, int64_t entry, int64_t limit, uint32_t max_len,
                                 uint32_t *d_nxt, uint32_t *d_tmp, uint32_t *d_mark, hipStream_t stream) {
    if (M == 0) return hipSuccess;
    const dim3 block(256), grid((M + 1 + 255) / 256);
    hipLaunchKernelGGL(k_short_clear, grid, block, 0, stream, d_mark, M);
    hipLaunchKernelGGL(k_long_next, grid, block, 0, stream, d_recs, M, (int32_t)std::min<int64_t>(entry, 0x7fffffff),
                       (int32_t)std::min<int64_t>(limit, 0x7fffffff), (int32_t)max_len, d_nxt, d_mark);
    (void)d_tmp; // (the chain is marked by the caller: acgpu_api.hip mark_chain)
    return hipGetLastError();
}

// Marks every element of the chain that starts at the elements already marked: nxt[k] in (k, M], nxt[M] = M.
// d_nxt and d_tmp (M+1 entries each) are both clobbered.
hipError_t launch_chain_mark(uint32_t *d_nxt, uint32_t *d_tmp, uint32_t *d_mark, uint32_t M, hipStream_t stream) {
    const dim3 block(256), grid((M + 1 + 255) / 256);
    uint32_t *in = d_nxt, *out = d_tmp;
    for (uint64_t reach = 1; reach <= M; reach <<= kRoundLog) { // after the launch with 8^t-step jumps, 8^(t+1) chain elements are marked
        hipLaunchKernelGGL(k_short_round, grid, block, 0, stream, in, out, d_mark, M);
        std::swap(in, out);
    }
    return hipGetLastError();
}

hipError_t launch_shortest_emit(const int32_t *d_recs, uint32_t M, const uint32_t *d_mark, const uint64_t *d_offsets,
                                const uint64_t *d_total, int record_kind, void *d_out, uint64_t cap, int64_t entry,
                                unsigned long long *d_exit, hipStream_t stream) {
    const dim3 block(256), grid((std::max<uint32_t>(M, 1) + 255) / 256);
    if (record_kind == ACGPU_REC_SET)
        hipLaunchKernelGGL(k_short_emit<ACGPU_REC_SET>, grid, block, 0, stream, d_recs, M, d_mark, d_offsets, d_total, d_out, cap,
                           entry, d_exit);
    else
        hipLaunchKernelGGL(k_short_emit<ACGPU_REC_MAP>, grid, block, 0, stream, d_recs, M, d_mark, d_offsets, d_total, d_out, cap,
                           entry, d_exit);
    return hipGetLastError();
}

} // namespace acgpu
