// acgpu_longest.hip -- LongestMatchSet/Map on gfx950.
//
// The reference (S/LongestMatchSet.java:192-265 with S/SetMatchQueue.java:45-95) delivers the leftmost-longest
// non-overlapping matches, i.e. the greedy chain  pos -> pos + max(L[pos], 1)  started at 0, where L[pos] is the
// length of the longest keyword starting at pos (T/LongestMatchTest.java:30-42 is the same statement).
//
//  k_longest_walk  : L[pos] for every unit: a forward walk of the keyword trie from every position (position
//                    parallel, hot trie rows in LDS).  (A right-to-left scan with the automaton of the reversed
//                    keywords would bound the work per unit, but reversing a prefix-closed dictionary -- config 4 --
//                    blows 50k trie nodes up to 24M states.)
//  k_longest_sync  : one synchronisation point per tile -- a position every greedy chain that can enter the tile must
//                    pass (found by following all candidate chains until they have merged).
//  k_longest_chain : one lane per tile follows the chain from its synchronisation point to the next tile's, so lanes
//                    are independent and their records concatenate in position order.  Two passes: count, (prefix
//                    sum), write.
#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

constexpr int kLScanBlock = 1024;

struct __attribute__((packed, aligned(2))) Units8 { // 8 UTF-16 units at any unit address (one global_load_dwordx4)
    uint32_t d[4];
};

// L[pos] for every owned position: walk the keyword trie forward from pos, remember the deepest node that ends a
// keyword.  Position parallel (lane i of a wave = position base+i: text loads and the len[] stores are coalesced).
// DENSE: the hot (shallow, BFS-first) rows of the class-indexed goto table sit in LDS, re-encoded while they are staged
// as {bit 31: the child ends a keyword, low bits: BYTE offset of the child's row}, so a step is one add, one ds_read and
// a few selects; rows beyond the LDS budget are read from the table in global memory (rare: deep nodes).
template <typename LenT, bool DENSE>
__global__ __launch_bounds__(kLScanBlock) void k_longest_walk(DevTables T, LongestScanLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *rows = reinterpret_cast<uint32_t *>(smem);
    const uint32_t *glob = reinterpret_cast<const uint32_t *>(T.dfa);
    const uint32_t row_bytes = T.n_cls * 4u;
    const uint32_t lds_entries = DENSE ? L.lds_rows * T.n_cls : 0;
    const uint32_t lds_bytes = lds_entries * 4u;
    const uint32_t lds_last = lds_bytes ? lds_bytes - 4u : 0u;
    for (uint32_t i = threadIdx.x; i < lds_entries; i += blockDim.x) {
        const uint32_t e = glob[i];
        rows[i] = e ? ((e & 0x80000000u) | ((e & 0x7fffffffu) * row_bytes)) : 0u;
    }
    __syncthreads();
    LenT *out_len = reinterpret_cast<LenT *>(L.d_len);
    const uint16_t *hay = L.d_hay;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n = L.n_units;
    // every lane of a wave runs the same number of iterations (64 consecutive positions per wave and iteration), so
    // that the wave can publish the farthest landing position of its 64 positions
    for (uint32_t p0 = L.own_begin + blockIdx.x * blockDim.x; p0 < L.own_end; p0 += stride) {
        const uint32_t p = p0 + threadIdx.x;
        uint32_t reach = 0;
        if (p >= L.own_end) {
            // past the end of the owned range: contributes nothing
        } else if (DENSE) {
            uint32_t off = 0, depth = 0, best = 0, best_off = 0, i = p;
            bool alive = true;
            while (alive) {
                // eight units per load; the walk usually ends inside the first window
                const uint32_t nvalid = min(n - min(i, n), 8u);
                uint32_t w[4] = {0, 0, 0, 0};
                if (nvalid == 8) {
                    const Units8 u = *reinterpret_cast<const Units8 *>(hay + i);
                    w[0] = u.d[0]; w[1] = u.d[1]; w[2] = u.d[2]; w[3] = u.d[3];
                } else {
                    for (uint32_t j = 0; j < nvalid; ++j) w[j >> 1] |= (uint32_t)hay[i + j] << (16 * (j & 1));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t unit = (w[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    uint32_t cls4;
                    if (T.range_cls) {
                        const uint32_t dlt = unit - T.cls_base;
                        cls4 = dlt < T.cls_span ? dlt * 4u + 4u : 0u;
                    } else {
                        cls4 = (uint32_t)T.cls_lut[unit] * 4u;
                    }
                    const uint32_t at = off + cls4;
                    uint32_t e = rows[min(at, lds_last) >> 2]; // always an LDS read (a dead lane's is ignored)
                    if (at >= lds_bytes) {                            // deep row: global table, child ids
                        const uint32_t g = alive ? glob[at >> 2] : 0u;
                        e = g ? ((g & 0x80000000u) | ((g & 0x7fffffffu) * row_bytes)) : 0u;
                    }
                    alive = alive && (uint32_t)j < nvalid && e != 0;
                    if (alive) {
                        off = e & 0x7fffffffu;
                        ++depth;
                        if (e >> 31) {
                            best = depth;
                            best_off = off;
                        }
                    }
                }
                i += 8;
            }
            out_len[p] = (LenT)best;
            if (L.d_state) L.d_state[p] = best_off / row_bytes;
            reach = p + (best ? best : 1u);
        } else {
            uint32_t node = 0, best = 0, best_node = 0, i = p;
            bool alive = true;
            while (alive && i < n) {
                const uint32_t unit = hay[i];
                const uint32_t f = T.cs ? unit : (uint32_t)T.lower[unit];
                const uint32_t c = hashed_goto(T.hkeys, T.hvals, T.hmask, node, f);
                if (c == ~0u) {
                    alive = false;
                } else {
                    node = c;
                    ++i;
                    if (T.term_id[c] != ~0u) {
                        best = i - p;
                        best_node = node;
                    }
                }
            }
            out_len[p] = (LenT)best;
            if (L.d_state) L.d_state[p] = best_node;
            reach = p + (best ? best : 1u);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) reach = max(reach, (uint32_t)__shfl_xor((int)reach, d));
        const uint32_t wave_p0 = p0 + (threadIdx.x & ~63u);
        if ((threadIdx.x & 63u) == 0 && wave_p0 < L.own_end) L.d_blockmax[(wave_p0 - L.own_begin) >> 6] = reach;
    }
}

// The lean form for range classes (case sensitive, keyword units within a span of 63: config 4, DNA, a-z ...), the one
// that matters for throughput: the kernel is VALU bound (2.8 G wave-instructions at config 4), so a step is cut down to
// unit extract, sub + min (column), lshl_add (address), ds_read, and (next row), cmp + cndmask (longest keyword so far):
//  * the LDS rows are re-encoded while they are staged: column = min(unit - base, span) (the last column = "any other
//    unit"), entry = {bit 31: the child ends a keyword, low bits: BYTE offset of the child's row};
//  * a missing transition leads to a DEAD row that loops to itself, so dead lanes need no predicate, and the depth of a
//    step is the same for every lane of the wave (all start together), i.e. a constant of the unrolled step;
//  * children beyond the LDS rows lead to a DEEP row (also a self loop); a lane that ends there (never on config 4:
//    nodes deeper than ~120 units) redoes its walk through the table in global memory.
template <typename LenT, bool STATE>
__global__ __launch_bounds__(kLScanBlock) void k_longest_walk_range(DevTables T, LongestScanLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *rows = reinterpret_cast<uint32_t *>(smem);
    const uint32_t *glob = reinterpret_cast<const uint32_t *>(T.dfa);
    const uint32_t n = T.n_cls, span = T.cls_span, base = T.cls_base; // n == span + 1
    const uint32_t row_bytes = n * 4u;
    const uint32_t real_bytes = L.lds_rows * row_bytes, dead_off = real_bytes, deep_off = real_bytes + row_bytes;
    for (uint32_t i = threadIdx.x; i < (L.lds_rows + 2) * n; i += blockDim.x) {
        const uint32_t r = i / n, col = i - r * n;
        uint32_t e;
        if (r >= L.lds_rows) {
            e = r == L.lds_rows ? dead_off : deep_off;
        } else {
            const uint32_t g = col < span ? glob[r * n + col + 1] : 0u; // class of column j is j+1; the last column has none
            if (!g) e = dead_off;
            else if ((g & 0x7fffffffu) >= L.lds_rows) e = deep_off;
            else e = (g & 0x80000000u) | ((g & 0x7fffffffu) * row_bytes);
        }
        rows[i] = e;
    }
    __syncthreads();
    LenT *out_len = reinterpret_cast<LenT *>(L.d_len);
    const uint16_t *hay = L.d_hay;
    const unsigned char *rows8 = reinterpret_cast<const unsigned char *>(rows);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t nu = L.n_units;
    for (uint32_t p0 = L.own_begin + blockIdx.x * blockDim.x; p0 < L.own_end; p0 += stride) {
        const uint32_t p = p0 + threadIdx.x;
        uint32_t reach = 0;
        if (p < L.own_end) {
            uint32_t off = 0, best = 0, best_off = 0, i = p, k0 = 0;
            for (;;) {
                const uint32_t nvalid = min(nu - min(i, nu), 8u);
                uint32_t w[4] = {0, 0, 0, 0};
                if (nvalid == 8) {
                    const Units8 u = *reinterpret_cast<const Units8 *>(hay + i);
                    w[0] = u.d[0]; w[1] = u.d[1]; w[2] = u.d[2]; w[3] = u.d[3];
                } else {
                    for (uint32_t j = 0; j < nvalid; ++j) w[j >> 1] |= (uint32_t)hay[i + j] << (16 * (j & 1));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t unit = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu);
                    uint32_t col = min(unit - base, span);
                    if (nvalid != 8 && (uint32_t)j >= nvalid) col = span; // past the end of the buffer: no transition
                    const uint32_t e = *reinterpret_cast<const uint32_t *>(rows8 + off + col * 4u);
                    off = e & 0x7fffffffu;
                    if ((int32_t)e < 0) {
                        best = k0 + (uint32_t)j + 1u;
                        if (STATE) best_off = off;
                    }
                }
                if (off >= real_bytes) break; // dead or deep
                i += 8;
                k0 += 8;
            }
            uint32_t best_node = best_off / row_bytes;
            if (off == deep_off) { // rare: redo the walk through the table in global memory
                uint32_t node = 0, j = p;
                best = 0;
                best_node = 0;
                while (j < nu) {
                    const uint32_t dlt = hay[j] - base;
                    const uint32_t e = glob[node * n + (dlt < span ? dlt + 1u : 0u)];
                    if (!e) break;
                    node = e & 0x7fffffffu;
                    ++j;
                    if (e >> 31) {
                        best = j - p;
                        best_node = node;
                    }
                }
            }
            out_len[p] = (LenT)best;
            if (STATE) L.d_state[p] = best_node;
            reach = p + (best ? best : 1u);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) reach = max(reach, (uint32_t)__shfl_xor((int)reach, d));
        const uint32_t wave_p0 = p0 + (threadIdx.x & ~63u);
        if ((threadIdx.x & 63u) == 0 && wave_p0 < L.own_end) L.d_blockmax[(wave_p0 - L.own_begin) >> 6] = reach;
    }
}

hipError_t launch_longest_scan(const DevTables &t, const LongestScanLaunch &l, hipStream_t stream, const char **kernel_name) {
#define ACGPU_LAUNCH(KERNEL, NAME)                                                                                      \
    do {                                                                                                                \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)l.lds_bytes);                                                           \
        if (e != hipSuccess) return e;                                                                                  \
        hipLaunchKernelGGL(KERNEL, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, t, l);                             \
        if (kernel_name) *kernel_name = NAME;                                                                           \
    } while (0)
    if (l.pairs) { // (field name kept: the lean range-class form)
        if (l.d_state) {
            if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_walk_range<uint16_t, true>), "k_longest_walk_range<unsigned short, true>");
            else ACGPU_LAUNCH((k_longest_walk_range<uint32_t, true>), "k_longest_walk_range<unsigned int, true>");
        } else {
            if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_walk_range<uint16_t, false>), "k_longest_walk_range<unsigned short, false>");
            else ACGPU_LAUNCH((k_longest_walk_range<uint32_t, false>), "k_longest_walk_range<unsigned int, false>");
        }
    } else if (t.dense) {
        if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_walk<uint16_t, true>), "k_longest_walk<unsigned short, true>");
        else ACGPU_LAUNCH((k_longest_walk<uint32_t, true>), "k_longest_walk<unsigned int, true>");
    } else {
        if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_walk<uint16_t, false>), "k_longest_walk<unsigned short, false>");
        else ACGPU_LAUNCH((k_longest_walk<uint32_t, false>), "k_longest_walk<unsigned int, false>");
    }
#undef ACGPU_LAUNCH
    return hipGetLastError();
}

// ---- chain -------------------------------------------------------------------------------------------------
// Synchronisation point of tile t (t >= 1): follow, in lock step by position, the greedy chains of EVERY position at
// which a chain can enter the tile -- the landings q + max(L[q],1) >= tb of the max_len positions before the tile
// start tb (the true chain's last position before tb is one of those q).  Chains that land on the same position are
// one chain from there on; when a single one is left, its position lies on every possible chain, hence on the true
// one.  Stored in S[t] if it falls inside the tile, else ~0u (the tile then belongs to an earlier lane's segment).
#ifndef ACGPU_SYNC_SET
#define ACGPU_SYNC_SET 16
#endif
constexpr int kSyncSet = ACGPU_SYNC_SET;

template <typename LenT>
__global__ __launch_bounds__(256) void k_longest_sync(LongestChainLaunch L, uint32_t *S) {
    __shared__ uint32_t set_all[256][kSyncSet + 1];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_tiles) return;
    const LenT *len = reinterpret_cast<const LenT *>(L.d_len);
    uint32_t *set = set_all[threadIdx.x];
    if (t == 0) {
        S[0] = L.entry;
        return;
    }
    const uint64_t tb64 = (uint64_t)L.entry + (uint64_t)t * L.tile_units;
    if (tb64 >= L.own_end) {
        S[t] = ~0u;
        return;
    }
    const uint32_t tb = (uint32_t)tb64;
    const uint32_t te = (uint32_t)min((uint64_t)L.own_end, tb64 + L.tile_units);
    uint32_t cnt = 0;
    bool ok = true;
    auto insert = [&](uint32_t v) {
        for (uint32_t i = 0; i < cnt; ++i)
            if (set[i] == v) return; // two chains merged
        if (cnt == kSyncSet) {
            ok = false;
            return;
        }
        set[cnt++] = v;
    };
    const uint32_t w = L.max_len > 0 ? L.max_len : 1;
    uint32_t q = tb - L.entry > w ? tb - w : L.entry;
    while (q < tb && ok) { // 64 positions at a time; blocks none of whose positions lands at or after tb are skipped
        const uint32_t b = (q - L.own_begin) >> 6;
        const uint32_t bend = min(tb, L.own_begin + ((b + 1) << 6));
        if (L.d_blockmax[b] >= tb) {
            for (; q < bend && ok; ++q) {
                const uint32_t l = (uint32_t)len[q];
                const uint32_t land = q + (l > 0 ? l : 1u);
                if (land >= tb) insert(land);
            }
        }
        q = bend;
    }
    while (ok && cnt > 1) {
        uint32_t mi = 0;
        for (uint32_t i = 1; i < cnt; ++i)
            if (set[i] < set[mi]) mi = i;
        const uint32_t p = set[mi];
        if (p >= te) { // no merge inside the tile
            ok = false;
            break;
        }
        set[mi] = set[--cnt];
        const uint32_t l = (uint32_t)len[p];
        insert(p + (l > 0 ? l : 1u));
    }
    S[t] = (ok && cnt == 1 && set[0] < te) ? set[0] : ~0u;
}

// One lane per tile with a synchronisation point: follows the chain from S[t] to the next tile's synchronisation point
// (or out of the owned range), counting or writing the matches met.  The write pass stages a lane's records in LDS and
// stores them as whole aligned groups (8 Set records = 64 bytes, 4 Map records = 48 bytes, as consecutive 16-byte
// stores by the same lane), so that the L2 sees full sectors instead of one 8-byte store per line and instruction.
constexpr int kChainBlock = 256;

template <typename LenT, bool WRITE>
__global__ __launch_bounds__(kChainBlock) void k_longest_chain(LongestChainLaunch L, const uint32_t *S) {
    __shared__ int2 ring_se[WRITE ? 8 : 1][kChainBlock]; // [slot][lane]: conflict-free for a lane's own slots
    __shared__ int ring_id[WRITE ? 4 : 1][kChainBlock];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_tiles) return;
    const LenT *len = reinterpret_cast<const LenT *>(L.d_len);
    const uint32_t start = S[t];
    if (start == ~0u || start >= L.own_end) {
        if (!WRITE) L.d_counts[t] = 0;
        if (t == 0 && !WRITE) *L.d_exit = L.entry; // entry at/after the end of the owned range
        return;
    }
    uint32_t target = ~0u; // the next synchronisation point on the chain
    for (uint32_t t2 = t + 1; t2 < L.n_tiles; ++t2) {
        const uint32_t v = S[t2];
        if (v != ~0u) {
            target = v;
            break;
        }
    }
    const bool set_kind = L.record_kind == ACGPU_REC_SET;
    const uint32_t gmask = set_kind ? 7u : 3u; // records per aligned group - 1
    uint32_t pos = start, count = 0;
    uint64_t dst = WRITE ? L.d_offsets[t] : 0;
    uint32_t gfirst = (uint32_t)dst & gmask; // first valid slot of the group being filled
    const uint32_t lane = threadIdx.x;
    // stores slots [from, to) of the current group one record at a time (group not complete, or beyond cap)
    auto flush_scalar = [&](uint64_t gbase, uint32_t from, uint32_t to) {
        for (uint32_t k = from; k < to; ++k) {
            const uint64_t d = gbase + k;
            if (d >= L.cap) break;
            const int2 se = ring_se[k][lane];
            if (set_kind) {
                reinterpret_cast<int2 *>(L.d_out)[d] = se;
            } else {
                int32_t *o = reinterpret_cast<int32_t *>(L.d_out) + d * 3;
                o[0] = se.x; o[1] = se.y; o[2] = ring_id[k][lane];
            }
        }
    };
    // 16-entry window of len[] in registers (two 16-byte loads): a dense chain (config 4: 6.4 positions per step) then
    // waits for memory once per 16 positions instead of once per step (a 32-entry window was slower: 3.7 ms against
    // 1.5 ms for the chain kernels at config 4)
    uint32_t win[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t wbase = ~0u - 64u; // no window yet
    auto len_at = [&](uint32_t q) -> uint32_t {
        if (sizeof(LenT) != 2) return (uint32_t)len[q];
        if (q - wbase >= 16u) { // (also true for q < wbase: the chain only moves forwards)
            const uint16_t *src = reinterpret_cast<const uint16_t *>(len) + q;
            const Units8 a = *reinterpret_cast<const Units8 *>(src), b = *reinterpret_cast<const Units8 *>(src + 8);
            win[0] = a.d[0]; win[1] = a.d[1]; win[2] = a.d[2]; win[3] = a.d[3];
            win[4] = b.d[0]; win[5] = b.d[1]; win[6] = b.d[2]; win[7] = b.d[3];
            wbase = q;
        }
        const uint32_t k = q - wbase;
        uint32_t x0 = (k & 8u) ? win[4] : win[0], x1 = (k & 8u) ? win[5] : win[1];
        uint32_t x2 = (k & 8u) ? win[6] : win[2], x3 = (k & 8u) ? win[7] : win[3];
        x0 = (k & 4u) ? x2 : x0;
        x1 = (k & 4u) ? x3 : x1;
        x0 = (k & 2u) ? x1 : x0;
        return (k & 1u) ? (x0 >> 16) : (x0 & 0xffffu);
    };
    while (pos < target && pos < L.own_end) {
        const uint32_t l = len_at(pos);
        if (l > 0) {
            if (WRITE) {
                const uint32_t k = (uint32_t)dst & gmask;
                ring_se[k][lane] = make_int2((int)pos, (int)(pos + l));
                if (!set_kind) ring_id[k][lane] = (int)L.d_out_id[L.d_state[pos]];
                if (k == gmask) { // the group is full up to its last slot
                    const uint64_t gbase = dst - gmask;
                    if (gfirst == 0 && dst < L.cap) {
                        if (set_kind) {
                            uint4 *o = reinterpret_cast<uint4 *>(reinterpret_cast<int2 *>(L.d_out) + gbase);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int2 r0 = ring_se[2 * q][lane], r1 = ring_se[2 * q + 1][lane];
                                o[q] = make_uint4((uint32_t)r0.x, (uint32_t)r0.y, (uint32_t)r1.x, (uint32_t)r1.y);
                            }
                        } else {
                            uint4 *o = reinterpret_cast<uint4 *>(reinterpret_cast<int32_t *>(L.d_out) + gbase * 3);
                            const int2 r0 = ring_se[0][lane], r1 = ring_se[1][lane], r2 = ring_se[2][lane], r3 = ring_se[3][lane];
                            o[0] = make_uint4((uint32_t)r0.x, (uint32_t)r0.y, (uint32_t)ring_id[0][lane], (uint32_t)r1.x);
                            o[1] = make_uint4((uint32_t)r1.y, (uint32_t)ring_id[1][lane], (uint32_t)r2.x, (uint32_t)r2.y);
                            o[2] = make_uint4((uint32_t)ring_id[2][lane], (uint32_t)r3.x, (uint32_t)r3.y, (uint32_t)ring_id[3][lane]);
                        }
                    } else {
                        flush_scalar(gbase, gfirst, gmask + 1);
                    }
                    gfirst = 0;
                }
            }
            ++dst;
            ++count;
            pos += l;
        } else {
            // no keyword starts here: skip the run of such positions eight at a time (sparse dictionaries: most of
            // the haystack), reading 16 bytes of lengths per step
            ++pos;
            if (sizeof(LenT) == 2) {
                const uint32_t limit = min(target, L.own_end);
                while (pos + 8 <= limit) {
                    const Units8 z = *reinterpret_cast<const Units8 *>(reinterpret_cast<const uint16_t *>(len) + pos);
                    const uint32_t any = z.d[0] | z.d[1] | z.d[2] | z.d[3];
                    if (any) {
                        // first non-zero 16-bit entry
                        uint32_t k = 0;
                        if (z.d[0]) k = (z.d[0] & 0xffffu) ? 0 : 1;
                        else if (z.d[1]) k = (z.d[1] & 0xffffu) ? 2 : 3;
                        else if (z.d[2]) k = (z.d[2] & 0xffffu) ? 4 : 5;
                        else k = (z.d[3] & 0xffffu) ? 6 : 7;
                        pos += k;
                        break;
                    }
                    pos += 8;
                }
            }
        }
    }
    if (WRITE) {
        const uint32_t k = (uint32_t)dst & gmask;
        if (k > gfirst) flush_scalar(dst - k, gfirst, k);
    }
    if (!WRITE) {
        L.d_counts[t] = count;
        if (pos >= L.own_end) *L.d_exit = pos; // exactly one lane's segment crosses the end of the owned range
    }
}

hipError_t launch_longest_sync(const LongestChainLaunch &l, uint32_t *d_sync, hipStream_t stream) {
    if (l.n_tiles == 0) return hipSuccess;
    const dim3 grid((l.n_tiles + 255) / 256), block(256);
    if (l.len_bytes == 2) hipLaunchKernelGGL((k_longest_sync<uint16_t>), grid, block, 0, stream, l, d_sync);
    else hipLaunchKernelGGL((k_longest_sync<uint32_t>), grid, block, 0, stream, l, d_sync);
    return hipGetLastError();
}

hipError_t launch_longest_chain(const LongestChainLaunch &l, const uint32_t *d_sync, bool write_pass, hipStream_t stream) {
    if (l.n_tiles == 0) return hipSuccess;
    const dim3 grid((l.n_tiles + kChainBlock - 1) / kChainBlock), block(kChainBlock);
    if (l.len_bytes == 2) {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain<uint16_t, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain<uint16_t, false>), grid, block, 0, stream, l, d_sync);
    } else {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain<uint32_t, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain<uint32_t, false>), grid, block, 0, stream, l, d_sync);
    }
    return hipGetLastError();
}

} // namespace acgpu
