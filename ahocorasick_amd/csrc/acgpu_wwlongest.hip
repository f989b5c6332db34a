// acgpu_wwlongest.hip -- WholeWordLongestMatchSet/Map on gfx950.
//
// The reference (S/WholeWordLongestMatchSet.java:47-178) walks a plain keyword trie from a word start through word AND
// non-word units (keywords may contain spaces) for as long as a transition exists; where the walk stops (unit i) it
// reports the whole path if it is a keyword and unit i is not a word character, else the last keyword on the path that
// was followed by a non-word unit (the node's carried "fail match", :226-244); then it skips to the first word start
// after i -- words covered by the walk are not rescanned (:85-99).  In parallel form:
//
//   k_wwl_starts (count / fill): the walk starts = position 0 and every word character whose left neighbour is not one,
//                  compacted in text order into RS[] (65536-bit table in LDS, one tile of 2048 units per block step).
//   k_wwl_walk   : one lane per walk start: the trie walk (hashed edges, L2 resident), the record it would report, and
//                  NXT[k] = the first walk start after the stop position (binary search in RS).
//   chain        : the walk starts the scan really visits are 0, NXT[0], NXT[NXT[0]], ...: marked by pointer doubling
//                  (launch_chain_mark, shared with acgpu_shortest.hip); the marked starts that report something are
//                  prefix-summed and scattered in order.
// Shards: a walk belongs to the shard that owns its first unit (left context 1 unit, right halo max_keyword_len + 1
// units); which walk starts the scan visits depends on where the previous shard's last walk stopped, handed on as
// chain_entry / chain_exit (the scan visits the first walk start at or after that position) -- the same hop as the Longest
// chain.  Word-character tables that are not fold-consistent (custom tables in case-insensitive mode: the reference
// mixes folded and raw lookups, S/WholeWordLongestMatchSet.java:127-157) take k_wwl_sequential, a literal single-lane
// restatement of the reference loop over the whole haystack.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

namespace {

constexpr int kStartsBlock = 256;
constexpr uint32_t kStartsTile = kStartsBlock * 8; // units per block step

__device__ __forceinline__ uint32_t wbit(const uint32_t *wbits, uint32_t unit) {
    return __builtin_amdgcn_ubfe(wbits[unit >> 5], unit, 1);
}

// FILL == false: counts[tile] = walk starts in the tile.  FILL == true: RS[offsets[tile] + rank] = position.
template <bool FILL>
__global__ __launch_bounds__(kStartsBlock) void k_wwl_starts(DevTables T, const uint16_t *hay, uint32_t n, uint32_t n_tiles,
                                                            uint32_t *counts, const uint64_t *offsets, uint32_t *rs, int text_begin,
                                                            int start_behind) {
    // start_behind >= 0 (acgpu_match_batch_u16: the separator unit): the unit behind every such unit is a walk start too --
    // the first unit of a haystack, where the reference's scan starts whatever stands there (a keyword without word
    // characters is kept as it is, R/WordCharacters.java:41-62, so the root can have a transition on a non-word unit)
    __shared__ uint32_t wbits[2048];
    __shared__ uint32_t wave_tot[kStartsBlock / kWave];
    for (uint32_t w = threadIdx.x; w < 2048; w += blockDim.x) wbits[w] = T.wbits[w];
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x / kWave;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t v = tile * kStartsTile + threadIdx.x * 8;
        uint32_t wm = 0, bm = 0; // bm: bit j + 1 = unit v + j is start_behind
        if (v + 8 <= n) {
            const uint4 q = *reinterpret_cast<const uint4 *>(hay + v);
            const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t u = (ww[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                wm |= wbit(wbits, u) << j;
                bm |= ((int)u == start_behind ? 2u : 0u) << j;
            }
        } else {
            for (uint32_t j = 0; j < 8 && v + j < n; ++j) {
                wm |= wbit(wbits, hay[v + j]) << j;
                bm |= ((int)hay[v + j] == start_behind ? 2u : 0u) << j;
            }
        }
        const uint32_t pu = (v > 0 && v <= n) ? (uint32_t)hay[v - 1] : 0x10000u;
        const uint32_t prev = pu < 0x10000u ? wbit(wbits, pu) : 0u;
        uint32_t sm = wm & ~((wm << 1) | prev) & 0xffu;
        if (start_behind >= 0) {
            bm |= (int)pu == start_behind ? 1u : 0u;
            const uint32_t in = v < n ? min(n - v, 8u) : 0u;
            sm |= bm & ((1u << in) - 1u);
        }
        if (v == 0 && n > 0 && text_begin) sm |= 1u; // the scan starts at position 0 of the TEXT whatever stands there
        const uint32_t cnt = __popc(sm);
        const uint32_t incl = wave_inclusive_scan(cnt);
        if (lane == kWave - 1) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kStartsBlock / kWave; ++w) {
            if ((uint32_t)w < wave) base += wave_tot[w];
            total += wave_tot[w];
        }
        if (!FILL) {
            if (threadIdx.x == 0) counts[tile] = total;
        } else {
            uint64_t dst = offsets[tile] + base + incl - cnt;
            while (sm) {
                rs[dst++] = v + (uint32_t)__builtin_ctz(sm);
                sm &= sm - 1;
            }
        }
        __syncthreads();
    }
}

// One lane per walk start.  Every step of a walk is a chain of dependent memory accesses, so the kernel is as fast as that
// chain is short: the text comes 8 units per 16-byte load, the fold table sits in LDS as shared pages (as in k_ww_tile;
// persistent workgroups, so it is staged once per workgroup), and a probe of the hashed trie edges requests key and value
// together -- one memory round trip per unit where the first version made four (unit, fold table, key, value).
constexpr uint32_t kWwlFoldPagesMax = 24; // 12 KB of LDS per workgroup (Unicode 13 simple lower-casing needs 18 pages)
struct __attribute__((packed, aligned(2))) WwlUnits8 {
    uint32_t d[4];
};

constexpr int kWwlWalkBlock = 512;
template <bool PAGED>
__global__ __launch_bounds__(kWwlWalkBlock) void k_wwl_walk(DevTables T, const uint16_t *hay, uint32_t n, const uint32_t *rs, uint32_t M,
                                                  uint32_t *nxt, uint32_t *mark, int32_t *mend, int32_t *mid, uint32_t *stop,
                                                  uint32_t entry, uint32_t plain_words) {
    // plain_words: no carried fail matches -- the walk of WholeWordMatchMap's loops (S/WholeWordMatchMap.java:55-153) over a
    // WHOLEWORD automaton (whose out_len holds the keyword's own length, not a fail match)
    __shared__ __attribute__((aligned(16))) unsigned char pgidx[PAGED ? 256 : 16];
    extern __shared__ __attribute__((aligned(16))) uint16_t pages[]; // PAGED: fold_n_pages * 256 deltas; then the Bloom words
    __shared__ __attribute__((aligned(16))) uint32_t wbits[2048]; // 65536 word-character bits
    // Bloom filter over the hashes of the first-word table (the builder's, as in k_ww_tile): a first word that is certainly
    // not in the table -- half of a text's words -- does not go to memory for it
    uint32_t *bloom = reinterpret_cast<uint32_t *>(pages + (PAGED ? T.fold_n_pages * 256u : 0u));
    const uint32_t bloom_words = T.ww_fat ? (T.ww_bloom_mask + 1u) / 32u : 0u;
    for (uint32_t i = threadIdx.x; i < bloom_words / 4; i += blockDim.x)
        reinterpret_cast<uint4 *>(bloom)[i] = reinterpret_cast<const uint4 *>(T.ww_bloom)[i];
    for (uint32_t i = threadIdx.x; i < 2048 / 4; i += blockDim.x)
        reinterpret_cast<uint4 *>(wbits)[i] = reinterpret_cast<const uint4 *>(T.wbits)[i];
    if (!PAGED) __syncthreads();
    if (PAGED) {
        for (uint32_t i = threadIdx.x; i < 256 / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(pgidx)[i] = reinterpret_cast<const uint4 *>(T.fold_pgidx)[i];
        for (uint32_t i = threadIdx.x; i < T.fold_n_pages * 32u; i += blockDim.x)
            reinterpret_cast<uint4 *>(pages)[i] = reinterpret_cast<const uint4 *>(T.fold_pages)[i];
        __syncthreads();
    }
    auto fold = [&](uint32_t u) -> uint32_t {
        if (T.cs) return u;
        if (PAGED) return (u + pages[(uint32_t)pgidx[u >> 8] * 256u + (u & 255u)]) & 0xffffu;
        return (uint32_t)T.lower[u];
    };
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k <= M; k += gridDim.x * blockDim.x) {
        if (k == M) {
            nxt[M] = M;
            mark[M] = 0;
            continue;
        }
        const uint32_t ws = rs[k];
        // the scan of this shard starts at the first walk start at or after the entry position
        mark[k] = (ws >= entry && (k == 0 || rs[k - 1] < entry)) ? 1u : 0u;
        uint32_t node = 0, i = ws, stop_unit = 0;
        bool walking = true, resolved = false; // resolved: the first-word table has settled the walk (nothing else to read)
        int32_t word_id = -1;
        // The first word of the walk, looked up WHOLE (the builder's table of the trie nodes a walk can stand on when its
        // first word ends: word-character paths that are a keyword or go on with a non-word unit; the hashing and the
        // two-choice table of k_ww_tile): one probe instead of one per unit.  A first word that is not in the table
        // reports nothing -- the walk dies inside it or at its end on a node without a keyword, and no keyword "followed by
        // a non-word unit" has been passed yet -- and the scan goes on behind it either way.  Runs of more than 12 units
        // and a walk that starts on a non-word unit (position 0) take the unit-by-unit walk from the start.
        if (T.ww_fat && ws + 16 <= n) {
            const WwlUnits8 a = *reinterpret_cast<const WwlUnits8 *>(hay + ws), b = *reinterpret_cast<const WwlUnits8 *>(hay + ws + 8);
            const uint32_t wd[8] = {a.d[0], a.d[1], a.d[2], a.d[3], b.d[0], b.d[1], b.d[2], b.d[3]};
            uint32_t wm = 0, f[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t u = (wd[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                wm |= ((wbits[u >> 5] >> (u & 31u)) & 1u) << j;
                f[j] = fold(u);
            }
            const uint32_t r = (uint32_t)__builtin_ctz(~wm | 0x10000u); // word characters from the start on (0..16)
            if (r >= 1 && r <= kWwInlineUnits) {
                uint32_t h = T.ww_seed, g = T.ww_seed, fw[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t lo = (uint32_t)(2 * q) < r ? f[2 * q] : 0u, hi = (uint32_t)(2 * q + 1) < r ? f[2 * q + 1] : 0u;
                    fw[q] = lo | (hi << 16);
                    h = ww_hash_step(h, fw[q]);
                    g = ww_hash2_step(g, fw[q]);
                }
                h = ww_hash_final(h);
                const uint32_t tag = ww_tag(h, r);
                const uint4 *fat = reinterpret_cast<const uint4 *>(T.ww_fat);
                const uint32_t s1 = ww_slot1(h, T.ww_fat_mask), s2 = ww_slot2(h, g, T.ww_fat_mask);
                const uint32_t b1 = ww_bloom_bit1(h, T.ww_bloom_mask), b2 = ww_bloom_bit2(h, T.ww_bloom_mask);
                const bool maybe = ((bloom[b1 >> 5] >> (b1 & 31u)) & (bloom[b2 >> 5] >> (b2 & 31u)) & 1u) != 0;
                uint4 ea0 = make_uint4(0u, 0u, 0u, 0u), ea1 = ea0, eb0 = ea0, eb1 = ea0;
                if (maybe) {
                    ea0 = fat[2 * s1]; ea1 = fat[2 * s1 + 1]; eb0 = fat[2 * s2]; eb1 = fat[2 * s2 + 1];
                }
                const bool in_a = ea0.x == tag && ea0.z == fw[0] && ea0.w == fw[1] && ea1.x == fw[2] && ea1.y == fw[3] &&
                                  ea1.z == fw[4] && ea1.w == fw[5];
                const bool in_b = eb0.x == tag && eb0.z == fw[0] && eb0.w == fw[1] && eb1.x == fw[2] && eb1.y == fw[3] &&
                                  eb1.z == fw[4] && eb1.w == fw[5];
                i = ws + r; // (the unit behind the run: a non-word unit, inside the buffer)
                walking = false;
                if (in_a || in_b) {
                    const uint32_t payload = in_a ? ea0.y : eb0.y;
                    if (payload >> 31) { // the path goes on with a non-word unit: unit by unit from its node
                        node = payload & 0x7fffffffu;
                        walking = true;
                    } else { // the walk ends here, on a keyword followed by a non-word unit: the payload is its id
                        word_id = (int32_t)payload;
                    }
                }
                resolved = !walking;
            }
        }
        while (walking && i < n) {
            // 8 units per load (unit by unit at the end of the buffer)
            WwlUnits8 w{{0u, 0u, 0u, 0u}};
            const uint32_t have = min(n - i, 8u);
            if (have == 8) {
                w = *reinterpret_cast<const WwlUnits8 *>(hay + i);
            } else {
                for (uint32_t j = 0; j < have; ++j) w.d[j >> 1] |= (uint32_t)hay[i + j] << (16 * (j & 1));
            }
            for (uint32_t j = 0; j < have; ++j) {
                const uint32_t u = (w.d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                // hashed goto edge (node, folded unit) -> child: key and value of a slot are requested together
                const uint64_t key = edge_key(node, fold(u));
                uint32_t slot = edge_hash(key) & T.hmask, child = ~0u;
                for (;;) {
                    const uint64_t kk = T.hkeys[slot];
                    const uint32_t vv = T.hvals[slot];
                    if (kk == key) {
                        child = vv;
                        break;
                    }
                    if (kk == kEmptyKey) break;
                    slot = (slot + 1) & T.hmask;
                }
                if (child == ~0u) {
                    stop_unit = u;
                    walking = false;
                    break;
                }
                node = child;
                ++i;
            }
        }
        // what the reference reports where the walk stops
        int32_t end = 0, id = -1;
        const bool at_end = i >= n;
        const bool stop_is_word = !resolved && !at_end && (T.wflags[stop_unit] & 1u);
        if (resolved) {
            if (word_id >= 0) {
                end = (int32_t)i;
                id = word_id;
            }
        } else if (!stop_is_word && node != 0 && T.term_id[node] != ~0u) {
            end = (int32_t)i; // the whole path is a keyword and ends at a word boundary
            id = (int32_t)T.term_id[node];
        } else if (!plain_words && T.out_len[node] != 0) { // the carried fail match: ends out_link[node] units before the stop
            end = (int32_t)(i - T.out_link[node]);
            id = (int32_t)T.out_id[node];
        }
        mend[k] = end;
        mid[k] = id;
        stop[k] = i;
        // the scan resumes at the first walk start after the stop position: almost always the very next start, otherwise a
        // few starts on (a walk runs over few words) -- a galloping search from k + 1, not a binary search over all M starts
        uint32_t lo = k + 1;
        if (lo < M && rs[lo] <= i) {
            uint32_t step = 1;
            while (lo + step < M && rs[lo + step] <= i) {
                lo += step;
                step <<= 1;
            }
            uint32_t hi = min(M, lo + step); // rs[lo] <= i, and rs[hi] > i or hi == M
            ++lo;
            while (lo < hi) {
                const uint32_t mid_k = lo + ((hi - lo) >> 1);
                if (rs[mid_k] <= i) lo = mid_k + 1;
                else hi = mid_k;
            }
        }
        nxt[k] = lo;
    }
}

// visited walk starts that report something and lie in the owned range; the LAST visited start of the owned range leaves
// the position behind its stop as the shard's chain exit (d_exit is preset to the entry: no visited start, no change)
__global__ __launch_bounds__(256) void k_wwl_select(const uint32_t *mark, const int32_t *mend, const uint32_t *rs, const uint32_t *nxt,
                                                    const uint32_t *stop, uint32_t *sel, uint32_t M, uint32_t own_begin,
                                                    uint32_t own_end, unsigned long long *d_exit) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const bool mine = mark[k] && rs[k] >= own_begin && rs[k] < own_end;
    sel[k] = (mine && mend[k] != 0) ? 1u : 0u;
    if (mine) {
        const uint32_t nk = nxt[k];
        if (nk >= M || rs[nk] >= own_end) *d_exit = (unsigned long long)stop[k] + 1ull;
    }
}

// ---- chain marking in one pass (instead of pointer doubling) ---------------------------------------------------------------
// The visited walk starts are the chain k0, NXT[k0], NXT[NXT[k0]], ... over the start INDICES: the same problem as the
// greedy chain of LongestMatchSet with "length" NXT[k] - k (>= 1, bounded by the starts a walk can run over), so the
// Longest chain kernels mark it: synchronisation points per tile of indices, one lane per tile following its segment and
// setting a bit per visited index (acgpu_longest.hip).  These two kernels translate to and from that form.
// jumps: len16[k] = NXT[k] - k, the farthest landing per 64 indices, the chain head k0 (the one element marked so far, d_head[0])
// and the largest jump (d_head[1])
template <bool MEASURE>
__global__ __launch_bounds__(256) void k_wwl_jumps(const uint32_t *nxt, const uint32_t *mark, uint32_t M, uint16_t *len16,
                                                   uint32_t *blockmax, unsigned long long *d_head) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t land = 0;
    if (k < M) {
        land = nxt[k];
        len16[k] = (uint16_t)min(land - k, 65535u);
        if (mark[k]) d_head[0] = k;
    }
    if (MEASURE) { // the largest jump (d_head[1]): bounds how far back a synchronisation scan has to look
        uint32_t j = k < M ? land - k : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) j = max(j, (uint32_t)__shfl_xor((int)j, d));
        // (one address for every wave: the atomic only when the value read is smaller -- half a million atomics on one
        // address were 5 ms)
        if ((threadIdx.x & 63u) == 0 && (unsigned long long)j > *reinterpret_cast<volatile unsigned long long *>(&d_head[1]))
            atomicMax(&d_head[1], (unsigned long long)j);
    }
    // (64 consecutive indices are one wave)
    uint32_t m = land;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63u) == 0 && k < M) blockmax[k >> 6] = m;
}

__global__ __launch_bounds__(256) void k_wwl_bits_to_mark(const uint32_t *bits, uint32_t M, uint32_t *mark) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < M) mark[k] = (bits[k >> 5] >> (k & 31u)) & 1u;
}

// Literal restatement of S/WholeWordLongestMatchSet.java:47-178 by ONE lane (word-character tables that are not
// fold-consistent; the whole haystack is one shard).  wflags bit 0 = wordChars[raw unit], bit 1 = wordChars[folded unit].
__global__ void k_wwl_sequential(DevTables T, const uint16_t *hay, uint32_t len, void *out, uint64_t cap, int record_kind,
                                 unsigned long long *counter) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long n = 0;
    auto emit = [&](uint32_t start, uint32_t end, uint32_t id) {
        if (n < cap) {
            if (record_kind == ACGPU_REC_SET) {
                reinterpret_cast<int2 *>(out)[n] = make_int2((int)start, (int)end);
            } else {
                int32_t *o = reinterpret_cast<int32_t *>(out) + n * 3;
                o[0] = (int)start; o[1] = (int)end; o[2] = (int)id;
            }
        }
        ++n;
    };
    // matchLength != 0 <=> the node is a keyword (its depth); failMatchLength/Offset/value = out_len/out_link/out_id
    auto report = [&](uint32_t node, uint32_t idx, bool own_match_allowed) {
        if (own_match_allowed && node != 0 && T.term_id[node] != ~0u) emit(idx - T.depth[node], idx, T.term_id[node]);
        else if (T.out_len[node] != 0) {
            const uint32_t fe = idx - T.out_link[node];
            emit(fe - T.out_len[node], fe, T.out_id[node]);
        }
    };
    uint32_t node = 0, idx = 0;
    while (idx < len) {
        const uint32_t raw = hay[idx];
        const uint32_t c = T.cs ? raw : (uint32_t)T.lower[raw];
        const uint32_t next = hashed_goto(T.hkeys, T.hvals, T.hmask, node, c);
        if (next == ~0u) {
            if (!(T.wflags[raw] & 2u)) { // !wordChars[c], c = folded unit (:73 / :128)
                report(node, idx, true);
            } else {
                report(node, idx, false); // only the fail match (:86-93)
                while (++idx < len && (T.wflags[hay[idx]] & 1u)) { // raw units (:95 / :150)
                }
            }
            while (++idx < len && !(T.wflags[hay[idx]] & 1u)) {
            }
            node = 0;
        } else {
            ++idx;
            node = next;
        }
    }
    report(node, idx, true);
    *counter = n;
}

// One workgroup per prefix-sum tile (2048 walk starts, 8 passes of 256): the selected starts are ranked inside the tile
// here -- ballots, the waves' totals through LDS -- on top of the tile's offset, so no 8-byte offset per start is written
// and read back (k_scan_apply: 0.24 ms for config 5's 30 M starts).
template <int REC>
__global__ __launch_bounds__(256) void k_wwl_emit(const uint32_t *rs, const uint32_t *sel, const int32_t *mend,
                                                  const int32_t *mid, const uint64_t *tile_offsets, uint32_t M, void *out,
                                                  uint64_t cap) {
    // element j * 256 + thread of the tile in pass j: every access coalesced; the rank of a selected start = the tile's offset +
    // the selected ones of the passes before + those of this pass in the waves and lanes before (ballots)
    __shared__ uint32_t tot[8][4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * 2048u;
    uint32_t pre[8];
    bool f[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t k = base + j * 256u + threadIdx.x;
        f[j] = k < M && sel[k] != 0;
        const uint64_t bal = __ballot(f[j]);
        pre[j] = (uint32_t)__popcll(bal & lanemask_lt());
        if (lane == 0) tot[j][wave] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    uint64_t running = tile_offsets[blockIdx.x];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t t = tot[j][w];
            before += w < wave ? t : 0u;
            all += t;
        }
        if (f[j]) {
            const uint32_t k = base + j * 256u + threadIdx.x;
            const uint64_t dst = running + before + pre[j];
            if (dst < cap) {
                if (REC == ACGPU_REC_SET) {
                    reinterpret_cast<int2 *>(out)[dst] = make_int2((int)rs[k], mend[k]);
                } else {
                    int32_t *o = reinterpret_cast<int32_t *>(out) + dst * 3;
                    o[0] = (int)rs[k]; o[1] = mend[k]; o[2] = mid[k];
                }
            }
        }
        running += all;
    }
}

} // namespace

hipError_t launch_wwl_jumps(const uint32_t *d_nxt, const uint32_t *d_mark, uint32_t M, uint16_t *d_len16, uint32_t *d_blockmax,
                            unsigned long long *d_head, bool measure_max_jump, hipStream_t stream) {
    if (measure_max_jump)
        hipLaunchKernelGGL(k_wwl_jumps<true>, dim3((M + 255) / 256), dim3(256), 0, stream, d_nxt, d_mark, M, d_len16, d_blockmax, d_head);
    else
        hipLaunchKernelGGL(k_wwl_jumps<false>, dim3((M + 255) / 256), dim3(256), 0, stream, d_nxt, d_mark, M, d_len16, d_blockmax, d_head);
    return hipGetLastError();
}

hipError_t launch_wwl_bits_to_mark(const uint32_t *d_bits, uint32_t M, uint32_t *d_mark, hipStream_t stream) {
    hipLaunchKernelGGL(k_wwl_bits_to_mark, dim3((M + 255) / 256), dim3(256), 0, stream, d_bits, M, d_mark);
    return hipGetLastError();
}


uint32_t wwl_tiles(uint32_t n_units) { return (n_units + kStartsTile - 1) / kStartsTile; }

hipError_t launch_wwl_starts(const DevTables &t, const uint16_t *d_hay, uint32_t n, int n_cu, bool fill, uint32_t *d_counts,
                             const uint64_t *d_offsets, uint32_t *d_rs, int text_begin, int start_behind,
                             hipStream_t stream) {
    const uint32_t n_tiles = wwl_tiles(n);
    if (n_tiles == 0) return hipSuccess;
    const dim3 block(kStartsBlock), grid(std::min<uint32_t>(n_tiles, (uint32_t)n_cu * 8));
    if (fill) hipLaunchKernelGGL(k_wwl_starts<true>, grid, block, 0, stream, t, d_hay, n, n_tiles, d_counts, d_offsets, d_rs, text_begin, start_behind);
    else hipLaunchKernelGGL(k_wwl_starts<false>, grid, block, 0, stream, t, d_hay, n, n_tiles, d_counts, d_offsets, d_rs, text_begin, start_behind);
    return hipGetLastError();
}

hipError_t launch_wwl_walk(const DevTables &t, bool plain_words, const uint16_t *d_hay, uint32_t n, const uint32_t *d_rs, uint32_t M,
                           uint32_t *d_nxt, uint32_t *d_mark, int32_t *d_mend, int32_t *d_mid, uint32_t *d_stop, uint32_t entry,
                           int n_cu, hipStream_t stream) {
    // persistent workgroups of 512 lanes (the LDS tables are staged once each): three per CU -- 8 KB of word bits, the fold
    // pages (Unicode: 9 KB) and up to 32 KB of Bloom words each -- fewer when there is less to do
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)std::max(n_cu, 1) * 3, ((uint64_t)M + kWwlWalkBlock) / kWwlWalkBlock);
    const bool paged = !t.cs && t.fold_n_pages >= 1 && t.fold_n_pages <= kWwlFoldPagesMax;
    const size_t lds = (paged ? (size_t)t.fold_n_pages * 512 : 0) + (t.ww_fat ? ((size_t)t.ww_bloom_mask + 1) / 8 : 0);
    if (paged) hipLaunchKernelGGL(k_wwl_walk<true>, dim3(grid), dim3(kWwlWalkBlock), lds, stream, t, d_hay, n, d_rs, M, d_nxt, d_mark, d_mend, d_mid, d_stop, entry, plain_words ? 1u : 0u);
    else hipLaunchKernelGGL(k_wwl_walk<false>, dim3(grid), dim3(kWwlWalkBlock), lds, stream, t, d_hay, n, d_rs, M, d_nxt, d_mark, d_mend, d_mid, d_stop, entry, plain_words ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_wwl_select(const uint32_t *d_mark, const int32_t *d_mend, const uint32_t *d_rs, const uint32_t *d_nxt,
                             const uint32_t *d_stop, uint32_t *d_sel, uint32_t M, uint32_t own_begin, uint32_t own_end,
                             unsigned long long *d_exit, hipStream_t stream) {
    hipLaunchKernelGGL(k_wwl_select, dim3((M + 255) / 256), dim3(256), 0, stream, d_mark, d_mend, d_rs, d_nxt, d_stop, d_sel, M,
                       own_begin, own_end, d_exit);
    return hipGetLastError();
}

hipError_t launch_wwl_sequential(const DevTables &t, const uint16_t *d_hay, uint32_t len, void *d_out, uint64_t cap,
                                 int record_kind, unsigned long long *d_counter, hipStream_t stream) {
    hipLaunchKernelGGL(k_wwl_sequential, dim3(1), dim3(64), 0, stream, t, d_hay, len, d_out, cap, record_kind, d_counter);
    return hipGetLastError();
}

hipError_t launch_wwl_emit(const uint32_t *d_rs, const uint32_t *d_sel, const int32_t *d_mend, const int32_t *d_mid,
                           const uint64_t *d_offsets, uint32_t M, int record_kind, void *d_out, uint64_t cap, hipStream_t stream) {
    const dim3 grid((M + 2047) / 2048), block(256); // (one workgroup per prefix-sum tile: scan_tile_elems() == 2048)
    if (record_kind == ACGPU_REC_SET)
        hipLaunchKernelGGL(k_wwl_emit<ACGPU_REC_SET>, grid, block, 0, stream, d_rs, d_sel, d_mend, d_mid, d_offsets, M, d_out, cap);
    else
        hipLaunchKernelGGL(k_wwl_emit<ACGPU_REC_MAP>, grid, block, 0, stream, d_rs, d_sel, d_mend, d_mid, d_offsets, M, d_out, cap);
    return hipGetLastError();
}

} // namespace acgpu
