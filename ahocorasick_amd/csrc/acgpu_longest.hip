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

#include <cstdio>
#include <type_traits>

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
    // table classes: the class pages (acgpu_build.cpp 7c) behind the rows -- two LDS reads per unit instead of a gather from
    // the 128 KB class table, in front of every step's row lookup
    const uint32_t pg_off = (max(lds_bytes, 16u) + 15u) & ~15u;
    const bool cls_lds = DENSE && L.pages_bytes != 0 && T.dfa_pages != nullptr;
    if (cls_lds)
        for (uint32_t i = threadIdx.x; i < T.dfa_pages_bytes / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(smem + pg_off)[i] = reinterpret_cast<const uint4 *>(T.dfa_pages)[i];
    const unsigned char *pg8 = smem + pg_off;
    const uint16_t *pg16 = reinterpret_cast<const uint16_t *>(smem + pg_off + 256);
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
                    } else if (cls_lds) { // (uniform)
                        cls4 = (uint32_t)pg16[((uint32_t)pg8[unit >> 8] << 8) + (unit & 255u)] * 4u;
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

// ---- the work-list form of the range-class walk ------------------------------------------------------------------------
// k_longest_walk_range keeps a wave in lock step until its LONGEST walk ends: at config 4 a wave runs ~24 trie steps for
// walks that average ~9 (SQ counters: 2.8 G VALU wave-instructions), i.e. 60 % of its lanes are dead weight.  Here a wave
// owns a chunk of 1024 consecutive positions and walks in rounds of 8 steps:
//   first round : 64 consecutive positions per lane group, all lanes alive at the start (text loads and len[] stores
//                 coalesced), two independent walks per lane (p and p+64) to cover the LDS latency of the dependent reads;
//   survivors   : walks still alive after a round are appended, compacted, to the wave's work list in LDS
//                 {row offset, longest keyword so far, position, depth}; whenever the list holds 64 entries a full batch
//                 takes its next round, and the list is run dry at the end of the chunk -- lanes are only idle in the few
//                 last batches of a chunk.
// A step is 3.5 VALU instructions instead of 7: the columns of two units come out of three packed 16-bit operations
// (v_pk_sub_u16, v_pk_min_u16, v_pk_lshlrev_b16), the LDS rows sit below 64 KiB so that the row offset is the low WORD of
// an entry (the address is one v_add_u32 with SDWA word selects, no masking), and "this node ends a keyword" (bit 31 of the
// entry) is shifted into a history word by one v_alignbit_b32 -- the longest keyword of a round is read off that word once
// per round.  The farthest landing of every 64 positions (d_blockmax) is collected in LDS: the first round's finished lanes
// by a wave reduction, later finishers by ds_max_u32.
// wave64 maximum in lane 63 with DPP row shifts and row broadcasts (VALU only: __shfl_xor is a ds_bpermute each, and the
// LDS is this kernel's busiest unit)
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t x) {
    // (written as v_max_u32_dpp: the compiler's own choice is a v_mov_b32_dpp and a v_max_u32 per step; lanes without a
    // source keep their value: bound_ctrl off, the old value is the destination itself)
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

constexpr int kWlChunk = 1024;  // positions per wave and chunk
constexpr int kWlCap = 128;     // work-list entries per wave (a round is run as soon as there are 64)

__device__ __forceinline__ uint32_t wl_pk_col4(uint32_t units2, uint32_t base2, uint32_t span2) {
    uint32_t t, u, r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t) : "v"(units2), "s"(base2));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(u) : "v"(t), "s"(span2));
    // (the shift count is packed too: an inline constant would only reach the low half)
    asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(r) : "s"(0x00020002u), "v"(u));
    return r;
}

// one round: 8 trie steps from row offset `e` (low word) through the 8 units in w[]; returns the last entry read.
// tm collects the keyword-end flags (bit 7-j = step j); hist[j] = entry after step j (STATE: which node the longest keyword is)
template <bool STATE, bool CHECK>
__device__ __forceinline__ uint32_t wl_round(const unsigned char *rows8, const uint32_t (&w)[4], uint32_t nvalid, uint32_t base2,
                                             uint32_t span2, uint32_t span4, uint32_t e, uint32_t &tm, uint32_t (&hist)[8]) {
    tm = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t cc = wl_pk_col4(w[d], base2, span2);
        if (CHECK) { // units past the end of the buffer: no transition
            if ((uint32_t)(2 * d) >= nvalid) cc = (cc & 0xffff0000u) | span4;
            if ((uint32_t)(2 * d + 1) >= nvalid) cc = (cc & 0xffffu) | (span4 << 16);
        }
        // address = row offset (low word of the entry) + column offset (low / high word of cc): one SDWA add each
        uint32_t a0, a1;
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(a0) : "v"(e), "v"(cc));
        e = *reinterpret_cast<const uint32_t *>(rows8 + a0);
        tm = __builtin_amdgcn_alignbit(tm, e, 31);
        if (STATE) hist[2 * d] = e;
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(a1) : "v"(e), "v"(cc));
        e = *reinterpret_cast<const uint32_t *>(rows8 + a1);
        tm = __builtin_amdgcn_alignbit(tm, e, 31);
        if (STATE) hist[2 * d + 1] = e;
    }
    return e;
}

// trie rows in STATIC LDS at offset 0 (a row offset is the ds_read address as it stands): 60 KiB, or 52 KiB next to the
// larger work lists of the Map flavour, so that two workgroups share a CU either way
constexpr int kWlRowWordsSet = 13312, kWlRowWordsMap = 13312; // 52 KiB of rows

// One byte per length with an escape: 255 = "255 or more, the value is in d_len_big[p]" (a full-size 16-bit array that is only
// written -- and read -- for such positions: config 4's keywords go up to 1000 units, its walks die below 40).
constexpr uint32_t kLenEscape = 255u;
template <typename LenT>
__device__ __forceinline__ void store_len(const LongestScanLaunch &L, uint32_t p, uint32_t best) {
    if (sizeof(LenT) == 1) {
        reinterpret_cast<uint8_t *>(L.d_len)[p] = (uint8_t)min(best, kLenEscape);
        if (best >= kLenEscape) L.d_len_big[p] = (uint16_t)best;
    } else {
        reinterpret_cast<LenT *>(L.d_len)[p] = (LenT)best;
    }
}

// ---- the first round through a ROOT TABLE: k_longest_block (Set records, small alphabets) ----------------------------------
// Every walk starts at the root, so what its first RK units do is a function of those units alone: the host builder's root
// table (acgpu_build.cpp 6b), one byte per RK-gram {bit 7: the walk is still alive | longest keyword among the RK steps},
// copied to LDS (16 KiB).  The index is a BIT FIELD: a unit's code (unit - base) takes B bits (B = 1: alphabets of up to two
// letters, RK = 14; B = 2: up to four letters, RK = 7), so the table has 2^14 entries instead of (letters + 1)^RK -- config
// 4: 1.9 % of the walks are alive after 14 units where 25 % survive 8.
//  * A wave owns a contiguous span of 1024-position chunks and streams it in steps of 512 positions; a lane owns 8
//    consecutive positions and loads ONLY their 16 bytes.  It packs the codes of its 8 units into one register -- even units
//    in the low half, odd units in the high half (one v_pk_sub_u16 and one v_lshl_or_b32 per two units) -- and takes the
//    packed codes of the units behind them from the next lanes (v_mov_b32_dpp wave_shl:1; the last lanes from the NEXT step's
//    first lanes: the stream runs two steps ahead, which also keeps a wave's loads in flight across its steps).
//  * The index of a position is two bit-field extracts -- the units at even offsets of its window from one half, those at odd
//    offsets from the other; the table is laid out for that de-interleaved index, and the index IS the LDS byte address.  One
//    LDS read per position, the 8 lengths leave as ONE store, the block maxima are preset to a bound.
//  * A lane some of whose 8 walks are still alive queues {first position, which of the 8} in LDS; when 64 lanes have queued
//    up they are expanded into the wave's work list and the listed walks run, from the root, in rounds of 16 steps through
//    the LDS rows (both 16-byte pieces of a walk's text requested together) -- every round but the wave's last few with all
//    64 lanes, the text still in the cache, and the one-byte stores of the final lengths into lines this wave has just written.
//    (Inside k_longest_walk_list the survivors of a 1024-position chunk, ~29, were drained chunk by chunk: a third of the
//    lanes busy, every round a memory latency the wave spent alone.  As two kernels -- survivor slices in global memory, a
//    kernel for the listed walks -- the byte stores went to lines long evicted: 0.12 ms of partial writes at config 4.)
//  * A chunk the kernel cannot do -- a unit outside the alphabet, the end of the buffer -- is FLAGGED (d_todo) and left to the
//    general kernel (k_longest_walk_list over the flagged chunks only).
constexpr uint32_t kBlockMaxSlack = 40; // block maxima are preset to last position + this: only a longer match needs the atomic
constexpr int kBlkRowWords = 7168;      // 28 KiB of trie rows next to the root table and the queues: two workgroups per CU
constexpr int kBlkQueueWords = 64 + 128 + 3 * kWlCap; // per wave: 128 u16 lane positions, 128 alive masks, the work list

__device__ __forceinline__ uint32_t dpp_wave_shl1(uint32_t x) { // lane i <- lane i + 1 (lane 63: 0)
    uint32_t r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1" : "=v"(r) : "v"(x));
    return r;
}

template <uint32_t B>
__global__ __launch_bounds__(kLScanBlock, 8) void k_longest_block(DevTables T, LongestScanLaunch L) {
    constexpr uint32_t RK = 14u / B, HE = (RK + 1) / 2, HO = RK / 2;
    constexpr uint32_t kEntries = 1u << 14;
    static_assert(B * RK == 14, "the root table has 2^14 entries");
    __shared__ __attribute__((aligned(16))) uint8_t rt[kEntries];
    __shared__ __attribute__((aligned(16))) uint32_t rows[kBlkRowWords];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[]; // per wave: kBlkQueueWords words
    const uint32_t *glob = reinterpret_cast<const uint32_t *>(T.dfa);
    const uint32_t n = T.n_cls, span = T.cls_span, base = T.cls_base; // n == span + 1
    const uint32_t row_bytes = n * 4u;
    const uint32_t real_bytes = L.lds_rows * row_bytes, dead_off = real_bytes, deep_off = real_bytes + row_bytes;
    for (uint32_t i = threadIdx.x; i < kEntries / 16; i += blockDim.x)
        reinterpret_cast<uint4 *>(rt)[i] = reinterpret_cast<const uint4 *>(T.root_tab)[i];
    for (uint32_t i = threadIdx.x; i < (L.lds_rows + 2) * n; i += blockDim.x) { // (as k_longest_walk_list stages them)
        const uint32_t r = i / n, col = i - r * n;
        uint32_t e;
        if (r >= L.lds_rows) {
            e = r == L.lds_rows ? dead_off : deep_off;
        } else {
            const uint32_t g = col < span ? glob[r * n + col + 1] : 0u;
            if (!g) e = dead_off;
            else if ((g & 0x7fffffffu) >= L.lds_rows) e = deep_off;
            else e = (g & 0x80000000u) | ((g & 0x7fffffffu) * row_bytes);
        }
        rows[i] = e;
    }
    __syncthreads();
    constexpr int kWaves = kLScanBlock / kWave;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t lane = lane_id();
    const uint32_t wave_global = blockIdx.x * kWaves + wave_in_block;
    uint32_t *wq = reinterpret_cast<uint32_t *>(smem) + wave_in_block * kBlkQueueWords;
    uint16_t *pbq = reinterpret_cast<uint16_t *>(wq); // queued lanes: (first position - span start) / 8
    uint32_t *mq = wq + 64;                           //               which of the 8 walks are alive
    uint32_t *wl0 = wq + 192, *wl1 = wl0 + kWlCap, *wl2 = wl0 + 2 * kWlCap; // work list: {row offset | best << 16}, position, depth
    const uint16_t *hay = L.d_hay;
    uint8_t *out_len = reinterpret_cast<uint8_t *>(L.d_len);
    const unsigned char *rows8 = reinterpret_cast<const unsigned char *>(rows);
    const uint32_t nu = L.n_units;
    const uint32_t base2 = base * 0x10001u, span2 = span * 0x10001u, span4 = span * 4u;
    const uint64_t own_len = (uint64_t)L.own_end - L.own_begin;
    const uint32_t n_chunks = (uint32_t)((own_len + kWlChunk - 1) / kWlChunk);
    const uint32_t c0 = (uint32_t)min((uint64_t)n_chunks, (uint64_t)wave_global * L.span_chunks);
    const uint32_t c1 = (uint32_t)min((uint64_t)n_chunks, (uint64_t)c0 + L.span_chunks);
    if (c0 >= c1) return;
    const uint32_t sp0 = L.own_begin + c0 * kWlChunk; // first position of the span
    uint32_t eq_n = 0, wl_n = 0;                      // wave-uniform: queued lanes, listed walks

    auto append = [&](bool alive, uint32_t off16, uint32_t best, uint32_t p, uint32_t depth) {
        const uint64_t bal = __ballot(alive);
        if (bal) {
            if (alive) {
                const uint32_t at = wl_n + (uint32_t)__popcll(bal & lanemask_lt());
                wl0[at] = off16 | (best << 16);
                wl1[at] = p;
                wl2[at] = depth;
            }
            wl_n += (uint32_t)__popcll(bal);
        }
    };
    // one more round of 16 steps for the top min(wl_n, 64) walks of the list (a listed walk is known to go on for RK units);
    // the DEAD row loops to itself, so the steps behind the end of a walk change nothing
    auto list_round = [&]() {
        const uint32_t nb = min(wl_n, (uint32_t)kWave);
        const uint32_t first = wl_n - nb;
        const bool act = lane < nb;
        uint32_t e = dead_off, best = 0, p = sp0, depth = 0;
        if (act) {
            const uint32_t a = wl0[first + lane];
            e = a & 0xffffu; best = a >> 16;
            p = wl1[first + lane];
            depth = wl2[first + lane];
        }
        __builtin_amdgcn_wave_barrier();
        wl_n = first;
        // (no bounds check: the chunk of a listed walk ends max_len + 24 units before the buffer does)
        uint32_t w0[4], w1[4], tm0, tm1, hist[8];
        {
            const uint16_t *src = hay + (act ? p + depth : sp0);
            const Units8 u = *reinterpret_cast<const Units8 *>(src), v = *reinterpret_cast<const Units8 *>(src + 8);
            w0[0] = u.d[0]; w0[1] = u.d[1]; w0[2] = u.d[2]; w0[3] = u.d[3];
            w1[0] = v.d[0]; w1[1] = v.d[1]; w1[2] = v.d[2]; w1[3] = v.d[3];
        }
        e = wl_round<false, false>(rows8, w0, 8u, base2, span2, span4, e, tm0, hist);
        e = wl_round<false, false>(rows8, w1, 8u, base2, span2, span4, e, tm1, hist);
        if (tm1 & 0xffu) best = depth + 16u - (uint32_t)__builtin_ctz(tm1 & 0xffu); // the last flagged step of the round
        else if (tm0 & 0xffu) best = depth + 8u - (uint32_t)__builtin_ctz(tm0 & 0xffu);
        const uint32_t off16 = e & 0xffffu;
        const bool alive = act && off16 < real_bytes && depth < 65000u;
        if (act && !alive) {
            if (off16 == deep_off) { // rare: redo the walk through the table in global memory
                uint32_t node = 0, j = p;
                best = 0;
                while (j < nu) {
                    const uint32_t dlt = hay[j] - base;
                    const uint32_t g = glob[node * n + (dlt < span ? dlt + 1u : 0u)];
                    if (!g) break;
                    node = g & 0x7fffffffu;
                    ++j;
                    if (g >> 31) best = j - p;
                }
            }
#ifdef ACGPU_ABLATION
            if (!(L.debug & 1u))
#endif
            store_len<uint8_t>(L, p, best);
            // (the block maximum was preset to its last position + kBlockMaxSlack: only a longer match raises it)
            const uint32_t blk = (p - L.own_begin) >> 6;
            if (p + best > L.own_begin + (blk << 6) + 63u + kBlockMaxSlack) atomicMax(&L.d_blockmax[blk], p + best);
        }
        append(alive, off16, best, p, depth + 16u);
        __builtin_amdgcn_wave_barrier();
    };
    // the top min(eq_n, 64) queued lanes -> their live walks into the work list (from the root), rounds as the list fills
    auto expand = [&]() {
        const uint32_t nb = min(eq_n, (uint32_t)kWave);
        const uint32_t first = eq_n - nb;
        uint32_t pb = 0, m = 0;
        if (lane < nb) {
            pb = sp0 + (uint32_t)pbq[first + lane] * 8u;
            m = mq[first + lane];
        }
        __builtin_amdgcn_wave_barrier();
        eq_n = first;
#ifdef ACGPU_ABLATION
        if (L.debug & 16u) m = 0; // queued lanes are dropped: no walks
#endif
        for (int i = 0; i < 8; ++i) { // alive bits: 7, 15, 23, 31 (positions 0-3) and 3, 11, 19, 27 (positions 4-7)
            append((m >> (i < 4 ? 8 * i + 7 : 8 * (i - 4) + 3)) & 1u, 0u, 0u, pb + (uint32_t)i, 0u);
            __builtin_amdgcn_wave_barrier();
            while (wl_n >= (uint32_t)kWave) list_round(); // fewer than 64 walks are left: the next 64 fit
        }
    };

    // ---- the stream: steps of 512 positions; own-unit codes of step s+1 and the text of step s+2 are ahead of step s ----
    const uint32_t n_steps = (c1 - c0) * 2u;
    const uint32_t last_load = (nu - 8u) & ~7u; // (the launch makes sure the buffer holds more than a chunk)
    auto load_own = [&](uint32_t st, uint32_t (&w)[4]) {
        const uint32_t pb = min(sp0 + st * 512u + lane * 8u, last_load); // (steps behind the span / the buffer: values nobody uses)
        const uint4 u = *reinterpret_cast<const uint4 *>(hay + pb);      // (16-byte aligned: the span starts on a chunk of the aligned own range)
        w[0] = u.x; w[1] = u.y; w[2] = u.z; w[3] = u.w;
    };
    // a lane's 8 units as packed codes: even units in bits 0.., odd units in bits 16..; O: the raw differences or-ed (units
    // outside the alphabet leave bits above a code's)
    auto pack_own = [&](const uint32_t (&w)[4], uint32_t &E, uint32_t &O) {
        uint32_t t1, t2, t3;
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(E) : "v"(w[0]), "s"(base2));
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t1) : "v"(w[1]), "s"(base2));
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t2) : "v"(w[2]), "s"(base2));
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t3) : "v"(w[3]), "s"(base2));
        O = E | t1 | t2 | t3;
        E = (t1 << B) | E;
        E = (t2 << (2 * B)) | E;
        E = (t3 << (3 * B)) | E;
    };
    constexpr uint32_t kCodeMask = ((1u << B) - 1u) * 0x10001u;
    uint32_t wa[4], wb[4], Ecur, Ocur, Enext, Onext;
    load_own(0, wa);
    load_own(1, wb);
    pack_own(wa, Ecur, Ocur);
    load_own(2, wa);
    bool chunk_bad = false;
    for (uint32_t st = 0; st < n_steps; ++st) {
        // (wa: text of step st + 2, in flight; wb: text of step st + 1)
        pack_own(wb, Enext, Onext);
#pragma unroll
        for (int k = 0; k < 4; ++k) wb[k] = wa[k];
        load_own(st + 3, wa);
        const uint32_t s0 = sp0 + st * 512u, pb = s0 + lane * 8u;
        // A chunk begins: its 16 block maxima are preset BEFORE any of its walks is queued -- a walk of this chunk can run to its
        // end inside expand() / list_round() during this very step (64 lanes queued, the list at 64 or more: dense live walks),
        // and a plain store at the END of the chunk would overwrite what its atomicMax had raised.  The store is issued ahead of
        // every text load of those rounds, and a round consumes its load (s_waitcnt vmcnt(0): stores pending) before the atomic.
        // A chunk that turns out to be flagged gets all 16 maxima from the general kernel, which runs behind this one.
        if (!(st & 1u) && lane < kWlChunk / 64 && (uint64_t)s0 + lane * 64u < (uint64_t)L.own_end)
            L.d_blockmax[((s0 - L.own_begin) >> 6) + lane] = s0 + lane * 64u + 63u + kBlockMaxSlack;
        // the codes of the units behind the lane's own: the next lanes' (the last lanes: the next step's first lanes')
        uint32_t E1 = dpp_wave_shl1(Ecur);
        E1 = lane == 63 ? (uint32_t)__builtin_amdgcn_readlane((int)Enext, 0) : E1;
        uint32_t E = (E1 << (4 * B)) | Ecur;
        uint32_t Oall = Ocur | (lane < 2 ? Onext : 0u);
        if (B == 1) {
            uint32_t E2 = dpp_wave_shl1(E1);
            E2 = lane == 63 ? (uint32_t)__builtin_amdgcn_readlane((int)Enext, 1) : E2;
            E = (E2 << 8) | E;
        }
        const uint32_t chunk_end = (uint32_t)min((uint64_t)L.own_end, (uint64_t)(s0 - ((st & 1u) ? 512u : 0u)) + kWlChunk);
        // not this kernel's: a unit outside the alphabet (also among the units of the next step that this step's last windows
        // cover), a chunk near the end of the buffer (every text load of its walks stays inside the buffer: walks are at most
        // max_len deep, a round reads 16 units from a walk's current depth), a last, partial chunk
        const bool bad = __any((Oall & ~kCodeMask) != 0u) || (uint64_t)chunk_end + max(T.max_len, 16u) + 24u > (uint64_t)nu ||
                         (uint64_t)(s0 - ((st & 1u) ? 512u : 0u)) + kWlChunk > (uint64_t)L.own_end;
        chunk_bad = ((st & 1u) ? chunk_bad : false) || bad;
        if (!bad) {
            uint32_t e[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
                const uint32_t k = i >> 1;
                const uint32_t idx = (i & 1u) ? (__builtin_amdgcn_ubfe(E, 16 + B * k, B * HE) | (__builtin_amdgcn_ubfe(E, B * (k + 1), B * HO) << (B * HE)))
                                              : (__builtin_amdgcn_ubfe(E, B * k, B * HE) | (__builtin_amdgcn_ubfe(E, 16 + B * k, B * HO) << (B * HE)));
                e[i] = rt[idx];
            }
            const uint32_t p0 = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24), p1 = e[4] | (e[5] << 8) | (e[6] << 16) | (e[7] << 24);
            // (a length of the first round is at most RK: no escape)
#ifdef ACGPU_ABLATION
            if (!(L.debug & 8u))
#endif
            *reinterpret_cast<uint2 *>(out_len + pb) = make_uint2(p0 & 0x0f0f0f0fu, p1 & 0x0f0f0f0fu);
            uint32_t m = (p0 & 0x80808080u) | ((p1 & 0x80808080u) >> 4);
#ifdef ACGPU_ABLATION
            if (L.debug & 4u) m = 0; // no live walks
#endif
            const uint64_t bal = __ballot(m != 0u);
            if (bal) {
                if (m != 0u) {
                    const uint32_t at = eq_n + (uint32_t)__popcll(bal & lanemask_lt());
                    pbq[at] = (uint16_t)((pb - sp0) >> 3);
                    mq[at] = m;
                }
                eq_n += (uint32_t)__popcll(bal);
                __builtin_amdgcn_wave_barrier();
                if (eq_n >= (uint32_t)kWave) expand(); // fewer than 64 lanes are left: the next step's fit
            }
        }
        if (st & 1u) { // the chunk is complete
            const uint32_t ck = c0 + (st >> 1);
            if (lane == 0) L.d_todo_w[ck] = chunk_bad ? 1 : 0;
        }
        Ecur = Enext;
        Ocur = Onext;
    }
    while (eq_n) expand();
    while (wl_n) list_round();
}

#ifdef ACGPU_TIMING
__device__ unsigned long long g_wl_rounds;
__device__ unsigned long long g_wl_timing[8]; // total, text wait, index + root lookups + store, appends + their rounds, tail rounds, waves
#define WL_MARK(i) { const unsigned long long t_ = clock64(); wlt[i] += t_ - wlt0; wlt0 = t_; }
#else
#define WL_MARK(i)
#endif
template <typename LenT, bool STATE>
__global__ __launch_bounds__(kLScanBlock, 8) void k_longest_walk_list(DevTables T, LongestScanLaunch L) { // (8 waves per SIMD: two workgroups per CU)
    __shared__ __attribute__((aligned(16))) uint32_t rows[STATE ? kWlRowWordsMap : kWlRowWordsSet];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[]; // the work lists
    const uint32_t *glob = reinterpret_cast<const uint32_t *>(T.dfa);
    const uint32_t n = T.n_cls, span = T.cls_span, base = T.cls_base; // n == span + 1
    const uint32_t row_bytes = n * 4u;
    const uint32_t real_bytes = L.lds_rows * row_bytes, dead_off = real_bytes, deep_off = real_bytes + row_bytes;
    if (L.d_todo != nullptr) { // only the chunks k_longest_block left: a workgroup none of whose chunks is flagged leaves at once
        const uint64_t nw = (uint64_t)gridDim.x * (kLScanBlock / kWave);
        const uint64_t nck = ((uint64_t)L.own_end - L.own_begin + kWlChunk - 1) / kWlChunk;
        bool has = false;
        for (uint64_t ck = (uint64_t)blockIdx.x * (kLScanBlock / kWave) + (threadIdx.x >> 6) + (uint64_t)(threadIdx.x & 63u) * nw; ck < nck;
             ck += 64ull * nw)
            has |= L.d_todo[ck] != 0;
        if (!__syncthreads_or(has)) return;
    }
    for (uint32_t i = threadIdx.x; i < (L.lds_rows + 2) * n; i += blockDim.x) { // (as k_longest_walk_range stages them)
        const uint32_t r = i / n, col = i - r * n;
        uint32_t e;
        if (r >= L.lds_rows) {
            e = r == L.lds_rows ? dead_off : deep_off;
        } else {
            const uint32_t g = col < span ? glob[r * n + col + 1] : 0u;
            if (!g) e = dead_off;
            else if ((g & 0x7fffffffu) >= L.lds_rows) e = deep_off;
            else e = (g & 0x80000000u) | ((g & 0x7fffffffu) * row_bytes);
        }
        rows[i] = e;
    }
    constexpr int kWaves = kLScanBlock / kWave;
    constexpr int kEntryWords = STATE ? 3 : 2;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t lane = lane_id();
    // behind the rows: per wave the work list (kEntryWords arrays of kWlCap words) and 16 block maxima
    uint32_t *wl = reinterpret_cast<uint32_t *>(smem) + wave_in_block * (kEntryWords * kWlCap + kWlChunk / 64);
    uint32_t *wl0 = wl, *wl1 = wl + kWlCap, *wl2 = wl + 2 * kWlCap; // {row offset | best << 16}, {position in chunk | depth << 16}, [best node's entry]
    uint32_t *bm = wl + kEntryWords * kWlCap;
    __syncthreads();
    const uint16_t *hay = L.d_hay;
    const unsigned char *rows8 = reinterpret_cast<const unsigned char *>(rows);
    const uint32_t nu = L.n_units;
    const uint32_t base2 = base * 0x10001u, span2 = span * 0x10001u, span4 = span * 4u;
    const uint32_t n_waves = gridDim.x * kWaves;
    const uint32_t wave_global = blockIdx.x * kWaves + wave_in_block;
    const uint64_t own_len = (uint64_t)L.own_end - L.own_begin;
    const uint32_t n_chunks = (uint32_t)((own_len + kWlChunk - 1) / kWlChunk);

#ifdef ACGPU_TIMING
    unsigned long long wlt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wlt0 = clock64();
    const unsigned long long wl_start = wlt0;
#endif
    // a finished walk: len[] (and the node), its landing into the block maximum; a walk that ran into the DEEP row is redone
    // through the table in global memory first (rare: nodes beyond the LDS rows)
    auto finish = [&](uint32_t p, uint32_t prel, uint32_t off16, uint32_t best, uint32_t best_e, bool by_atomic) {
        uint32_t best_node = STATE ? (best_e & 0xffffu) / row_bytes : 0u;
        if (off16 == deep_off) {
            uint32_t node = 0, j = p;
            best = 0;
            best_node = 0;
            // (large dictionaries: a few hundred of a million rows fit LDS, nearly EVERY walk comes through here.)  The first 16
            // units of text up front -- two loads in flight together -- so that a step waits for the table alone, not for the
            // text and then the table
            if (p + 16u <= nu) {
                const Units8 u0 = *reinterpret_cast<const Units8 *>(hay + p), u1 = *reinterpret_cast<const Units8 *>(hay + p + 8);
                const uint32_t w[8] = {u0.d[0], u0.d[1], u0.d[2], u0.d[3], u1.d[0], u1.d[1], u1.d[2], u1.d[3]};
                bool going = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (going) {
                        const uint32_t dlt = ((w[k >> 1] >> (16 * (k & 1))) & 0xffffu) - base;
                        const uint32_t g = glob[node * n + (dlt < span ? dlt + 1u : 0u)];
                        if (!g) {
                            going = false;
                        } else {
                            node = g & 0x7fffffffu;
                            if (g >> 31) {
                                best = (uint32_t)k + 1u;
                                best_node = node;
                            }
                        }
                    }
                }
                j = going ? p + 16u : nu; // (a walk that has ended skips the loop below)
            }
            while (j < nu) {
                const uint32_t dlt = hay[j] - base;
                const uint32_t g = glob[node * n + (dlt < span ? dlt + 1u : 0u)];
                if (!g) break;
                node = g & 0x7fffffffu;
                ++j;
                if (g >> 31) {
                    best = j - p;
                    best_node = node;
                }
            }
        }
        store_len<LenT>(L, p, best);
        if (STATE) L.d_state[p] = best_node;
        const uint32_t reach = p + (best ? best : 1u);
        if (by_atomic) atomicMax(&bm[prel >> 6], reach);
        return reach;
    };
    // the longest keyword of a round: the LAST flagged step (lowest set bit of the history), depth0 steps done before it
    auto round_best = [&](uint32_t tm, uint32_t depth0, const uint32_t (&hist)[8], uint32_t &best, uint32_t &best_e) {
        const uint32_t t8 = tm & 0xffu;
        if (t8) {
            const uint32_t b = (uint32_t)__builtin_ctz(t8); // step 7 - b
            best = depth0 + 8u - b;
            if (STATE) {
                uint32_t x0 = (b & 4u) ? hist[3] : hist[7], x1 = (b & 4u) ? hist[2] : hist[6];
                uint32_t x2 = (b & 4u) ? hist[1] : hist[5], x3 = (b & 4u) ? hist[0] : hist[4];
                x0 = (b & 2u) ? x2 : x0;
                x1 = (b & 2u) ? x3 : x1;
                best_e = (b & 1u) ? x1 : x0;
            }
        }
    };
    auto load8 = [&](uint32_t i, uint32_t (&w)[4], uint32_t &nvalid, bool check) {
        nvalid = 8;
        if (!check) {
            const Units8 u = *reinterpret_cast<const Units8 *>(hay + i);
            w[0] = u.d[0]; w[1] = u.d[1]; w[2] = u.d[2]; w[3] = u.d[3];
            return;
        }
        nvalid = min(nu - min(i, nu), 8u);
        w[0] = w[1] = w[2] = w[3] = 0;
        if (nvalid == 8) {
            const Units8 u = *reinterpret_cast<const Units8 *>(hay + i);
            w[0] = u.d[0]; w[1] = u.d[1]; w[2] = u.d[2]; w[3] = u.d[3];
        } else {
            for (uint32_t j = 0; j < nvalid; ++j) w[j >> 1] |= (uint32_t)hay[i + j] << (16 * (j & 1));
        }
    };
    uint32_t wl_n = 0; // wave-uniform: entries in the work list
    // survivors of a round -> work list (compacted by ballot); alive lanes only
    auto append = [&](bool alive, uint32_t e, uint32_t best, uint32_t best_e, uint32_t prel, uint32_t depth) {
        const uint64_t bal = __ballot(alive);
        if (alive) {
            const uint32_t at = wl_n + (uint32_t)__popcll(bal & lanemask_lt());
            wl0[at] = (e & 0xffffu) | (best << 16);
            wl1[at] = prel | (depth << 16);
            if (STATE) wl2[at] = best_e;
        }
        wl_n += (uint32_t)__popcll(bal);
    };
    // one more round for the top min(wl_n, 64) entries of the list
    auto list_round = [&](uint32_t chunk0, bool check) {
        const uint32_t nb = min(wl_n, (uint32_t)kWave);
        const uint32_t first = wl_n - nb;
        const bool act = lane < nb;
        uint32_t e = dead_off, best = 0, best_e = 0, prel = 0, depth = 0;
        if (act) {
            const uint32_t a = wl0[first + lane], b = wl1[first + lane];
            e = a & 0xffffu; best = a >> 16;
            prel = b & 0xffffu; depth = b >> 16;
            if (STATE) best_e = wl2[first + lane];
        }
        __builtin_amdgcn_wave_barrier();
        wl_n = first;
        const uint32_t p = chunk0 + prel;
        uint32_t w[4], nvalid, tm, hist[8];
#ifdef ACGPU_TIMING
        const unsigned long long lr0 = clock64();
#endif
        load8(act ? p + depth : chunk0, w, nvalid, check);
#ifdef ACGPU_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long lr1 = clock64();
#endif
        if (check) e = wl_round<STATE, true>(rows8, w, nvalid, base2, span2, span4, e, tm, hist);
        else e = wl_round<STATE, false>(rows8, w, nvalid, base2, span2, span4, e, tm, hist);
        round_best(tm, depth, hist, best, best_e);
#ifdef ACGPU_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long lr2 = clock64();
        wlt[5] += lr1 - lr0; wlt[6] += lr2 - lr1; wlt[7] += 1;
#endif
        const uint32_t off16 = e & 0xffffu;
        const bool alive = act && off16 < real_bytes && depth < 65000u; // (no keyword is that long: see the launch condition)
        if (act && !alive) (void)finish(p, prel, off16, best, best_e, true);
        append(alive, e, best, best_e, prel, depth + 8u);
        __builtin_amdgcn_wave_barrier();
    };

    // the general first round of positions [sa, sb) of a chunk: two walks per lane (p and p + 64) through the rows
    auto first_round = [&](uint32_t chunk0, uint32_t chunk_end, uint32_t sa, uint32_t sb, bool check) {
        for (uint32_t s0 = sa; s0 < sb; s0 += 2 * kWave) { // positions s0+lane and s0+64+lane
            uint32_t pp[2] = {s0 + lane, s0 + kWave + lane};
            uint32_t ee[2], tmm[2], hh[2][8], ww[2][4], nv[2];
            bool in[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                in[h] = pp[h] < chunk_end;
                load8(in[h] ? pp[h] : chunk0, ww[h], nv[h], check);
            }
            // (the two walks are written out side by side: their dependent LDS reads interleave)
            if (check) {
                ee[0] = wl_round<STATE, true>(rows8, ww[0], nv[0], base2, span2, span4, 0u, tmm[0], hh[0]);
                ee[1] = wl_round<STATE, true>(rows8, ww[1], nv[1], base2, span2, span4, 0u, tmm[1], hh[1]);
            } else {
                ee[0] = wl_round<STATE, false>(rows8, ww[0], nv[0], base2, span2, span4, 0u, tmm[0], hh[0]);
                ee[1] = wl_round<STATE, false>(rows8, ww[1], nv[1], base2, span2, span4, 0u, tmm[1], hh[1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t best = 0, best_e = 0;
                round_best(tmm[h], 0u, hh[h], best, best_e);
                const uint32_t off16 = ee[h] & 0xffffu;
                const bool alive = in[h] && off16 < real_bytes;
                const uint32_t prel = pp[h] - chunk0;
                uint32_t reach = 0;
                if (in[h] && !alive) reach = finish(pp[h], prel, off16, best, best_e, false);
                reach = wave_max_dpp(reach); // (no survivor has touched this block's maximum yet: a plain store)
                if (lane == 0 && s0 + h * kWave < chunk_end) bm[(s0 + h * kWave - chunk0) >> 6] = reach;
                append(alive, ee[h], best, best_e, prel, 8u);
                __builtin_amdgcn_wave_barrier();
                while (wl_n >= (uint32_t)kWave) list_round(chunk0, check); // fewer than 64 entries are left: the next append fits
            }
        }
    };
    // chunks wave_global, wave_global + n_waves, ...; with L.d_todo only the flagged ones (what k_longest_block left): the flags
    // of 64 of the wave's chunks are fetched at once
    for (uint64_t it0 = 0; (uint64_t)wave_global + it0 * n_waves < n_chunks; it0 += kWave) {
        const uint64_t ck_lane = (uint64_t)wave_global + (it0 + lane) * n_waves;
        const bool flagged = ck_lane < n_chunks && (L.d_todo == nullptr || L.d_todo[ck_lane] != 0);
        uint64_t todo_mask = __ballot(flagged);
      while (todo_mask) {
        const uint32_t ck = (uint32_t)((uint64_t)wave_global + (it0 + (uint32_t)__builtin_ctzll(todo_mask)) * n_waves);
        todo_mask &= todo_mask - 1;
        const uint32_t chunk0 = L.own_begin + ck * kWlChunk;
        const uint32_t chunk_end = (uint32_t)min((uint64_t)L.own_end, (uint64_t)chunk0 + kWlChunk);
        // every 16-byte text load of this chunk stays inside the buffer (walks are at most max_len deep)
        const bool check = (uint64_t)chunk_end + T.max_len + 8u > (uint64_t)nu;
        if (lane < kWlChunk / 64) bm[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        first_round(chunk0, chunk_end, chunk0, chunk_end, check);
        WL_MARK(3)
        while (wl_n) list_round(chunk0, check);
        __builtin_amdgcn_wave_barrier();
        const uint32_t nblk = (chunk_end - chunk0 + 63u) >> 6;
        if (lane < nblk) L.d_blockmax[((chunk0 - L.own_begin) >> 6) + lane] = bm[lane];
        __builtin_amdgcn_wave_barrier();
        WL_MARK(4)
      }
    }
#ifdef ACGPU_TIMING
    if (lane == 0) {
        atomicAdd(&g_wl_timing[0], clock64() - wl_start);
        for (int i = 1; i < 5; ++i) atomicAdd(&g_wl_timing[i], wlt[i]);
        atomicAdd(&g_wl_timing[5], 1ull);
        atomicAdd(&g_wl_timing[6], wlt[5]);
        atomicAdd(&g_wl_timing[7], wlt[6]);
        atomicAdd(&g_wl_rounds, wlt[7]);
    }
#endif
}

// dynamic LDS of k_longest_walk_list (the work lists); its trie rows are static: at most longest_list_max_rows(n_cls) rows
size_t longest_list_lds_bytes(bool state) { return (size_t)(kLScanBlock / kWave) * ((state ? 3 : 2) * kWlCap + kWlChunk / 64) * 4; }
uint32_t longest_list_max_rows(uint32_t n_cls, bool state) {
    return n_cls ? (uint32_t)(state ? kWlRowWordsMap : kWlRowWordsSet) / n_cls - 2u : 0u;
}

size_t longest_block_lds_bytes() { return (size_t)(kLScanBlock / kWave) * kBlkQueueWords * 4; }
uint32_t longest_block_max_rows(uint32_t n_cls) { return n_cls ? (uint32_t)kBlkRowWords / n_cls - 2u : 0u; }

// k_longest_block (one-byte lengths; l.d_todo_w, l.span_chunks, l.lds_rows <= longest_block_max_rows set); the chunks flagged in
// d_todo_w are left for launch_longest_scan with l.d_todo
hipError_t launch_longest_block(const DevTables &t, const LongestScanLaunch &l, hipStream_t stream, const char **kernel_name) {
    const size_t lds = longest_block_lds_bytes();
    hipError_t e;
    if (t.root_b == 1) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_longest_block<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_longest_block<1>), dim3(l.grid), dim3(l.block), lds, stream, t, l);
        if (kernel_name) *kernel_name = "k_longest_block<1u>";
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_longest_block<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_longest_block<2>), dim3(l.grid), dim3(l.block), lds, stream, t, l);
        if (kernel_name) *kernel_name = "k_longest_block<2u>";
    }
    return hipGetLastError();
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
    if (l.pairs == 2) { // the work-list form (8- or 16-bit lengths, LDS rows below 64 KiB)
        if (l.len_bytes == 1) {
            if (l.d_state) ACGPU_LAUNCH((k_longest_walk_list<uint8_t, true>), "k_longest_walk_list<unsigned char, true>");
            else ACGPU_LAUNCH((k_longest_walk_list<uint8_t, false>), "k_longest_walk_list<unsigned char, false>");
        } else if (l.d_state) ACGPU_LAUNCH((k_longest_walk_list<uint16_t, true>), "k_longest_walk_list<unsigned short, true>");
        else ACGPU_LAUNCH((k_longest_walk_list<uint16_t, false>), "k_longest_walk_list<unsigned short, false>");
    } else if (l.pairs) { // (field name kept: the lean range-class form)
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
#ifdef ACGPU_TIMING
    {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[8] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wl_timing), sizeof(h));
        unsigned long long nr = 0;
        (void)hipMemcpyFromSymbol(&nr, HIP_SYMBOL(g_wl_rounds), sizeof(nr));
        if (h[5]) fprintf(stderr, "[walk timing] waves %llu: total %.0f | text wait %.0f | index+lookups+store %.0f | appends+rounds %.0f | tail rounds %.0f | list rounds %.1f: text %.0f, steps %.0f (cycles per wave)\n",
                          h[5], (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5],
                          (double)nr / h[5], (double)h[6] / h[5], (double)h[7] / h[5]);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wl_rounds), z, sizeof(nr));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wl_timing), z, sizeof(z));
    }
#endif
    return hipGetLastError();
}

// a length as the chain kernels read it: one-byte lengths carry the escape kLenEscape (the value is in d_len_big)
template <typename LenT>
__device__ __forceinline__ uint32_t len_full(const LongestChainLaunch &L, uint32_t pos, uint32_t l) {
    if (sizeof(LenT) == 1 && l == kLenEscape) return (uint32_t)L.d_len_big[pos];
    return l;
}
struct __attribute__((packed, aligned(1))) Bytes16 { // 16 one-byte lengths at any address
    uint32_t d[4];
};

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
            if (sizeof(LenT) <= 2 && bend >= L.own_begin + 64u) {
                // the 64 lengths below bend in eight (one-byte lengths: four) independent 16-byte loads (one memory latency
                // for the block, not 64); which positions land at or beyond the tile start is decided in registers, only
                // those are inserted
                const uint32_t b0 = bend - 64u;
                uint64_t hit = 0;
                if (sizeof(LenT) == 2) {
                    Units8 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const Units8 *>(reinterpret_cast<const uint16_t *>(len) + b0 + 8 * j);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const uint32_t pos = b0 + 8u * j + i;
                            const uint32_t l = (v[j].d[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                            if (pos >= q && pos + (l > 0 ? l : 1u) >= tb) hit |= 1ull << (8 * j + i);
                        }
                } else {
                    Bytes16 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const Bytes16 *>(reinterpret_cast<const uint8_t *>(len) + b0 + 16 * j);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const uint32_t pos = b0 + 16u * j + i;
                            const uint32_t l = (v[j].d[i >> 2] >> (8 * (i & 3))) & 0xffu; // (an escaped length lands at pos + 255 or beyond)
                            if (pos >= q && (l == kLenEscape || pos + (l > 0 ? l : 1u) >= tb)) hit |= 1ull << (16 * j + i);
                        }
                }
                while (hit && ok) {
                    const uint32_t pos = b0 + (uint32_t)__builtin_ctzll(hit);
                    hit &= hit - 1;
                    const uint32_t l = len_full<LenT>(L, pos, (uint32_t)len[pos]);
                    if (pos + (l > 0 ? l : 1u) >= tb) insert(pos + (l > 0 ? l : 1u));
                }
            } else {
                for (; q < bend && ok; ++q) {
                    const uint32_t l = len_full<LenT>(L, q, (uint32_t)len[q]);
                    const uint32_t land = q + (l > 0 ? l : 1u);
                    if (land >= tb) insert(land);
                }
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
        const uint32_t l = len_full<LenT>(L, p, (uint32_t)len[p]);
        insert(p + (l > 0 ? l : 1u));
    }
    S[t] = (ok && cnt == 1 && set[0] < te) ? set[0] : ~0u;
}

// One lane per tile with a synchronisation point: follows the chain from S[t] to the next tile's synchronisation point
// (or out of the owned range), counting or writing the matches met.  The write pass stages a lane's records in LDS and
// stores them as whole aligned groups (8 Set records = 64 bytes, 4 Map records = 48 bytes, as consecutive 16-byte
// stores by the same lane), so that the L2 sees full sectors instead of one 8-byte store per line and instruction.
constexpr int kChainBlock = 256;

// BITS (count pass): the lane also marks every position at which it reports a match in the bitmap L.d_bits -- with it
// the records are written by k_longest_emit, in parallel over all positions, instead of a second serial pass over the
// chain (WRITE).  A segment's first and last bitmap word may be shared with its neighbours (atomicOr); the words between
// are its own (plain stores into the zeroed bitmap).
template <typename LenT, bool WRITE, bool BITS = false>
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
        if (sizeof(LenT) != 2) return len_full<LenT>(L, q, (uint32_t)len[q]);
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
    // BITS: the aligned group of four bitmap words (128 positions) being filled.  Stores sit in the lane's in-order vmcnt
    // stream in front of its next window load (gfx950 counts stores there), so they are few and wide: one 16-byte store per
    // group; only the first and the last group of a segment can be shared with a neighbour (atomicOr per word).
    uint32_t bg = ~0u;
    unsigned long long blo = 0, bhi = 0;
    bool bfirst = true;
    auto flush_bits = [&](bool shared) {
        if (!(blo | bhi)) return;
        uint32_t *dst = L.d_bits + (size_t)bg * 4u;
        if (shared) {
            if ((uint32_t)blo) atomicOr(dst, (uint32_t)blo);
            if ((uint32_t)(blo >> 32)) atomicOr(dst + 1, (uint32_t)(blo >> 32));
            if ((uint32_t)bhi) atomicOr(dst + 2, (uint32_t)bhi);
            if ((uint32_t)(bhi >> 32)) atomicOr(dst + 3, (uint32_t)(bhi >> 32));
        } else {
            *reinterpret_cast<uint4 *>(dst) = make_uint4((uint32_t)blo, (uint32_t)(blo >> 32), (uint32_t)bhi, (uint32_t)(bhi >> 32));
        }
    };
    while (pos < target && pos < L.own_end) {
        const uint32_t l = len_at(pos);
        if (l > 0) {
            if (BITS) {
                const uint32_t g = pos >> 7;
                if (g != bg) {
                    flush_bits(bfirst);
                    bfirst = bg == ~0u; // (still nothing flushed: the next flush is the segment's first)
                    bg = g;
                    blo = bhi = 0;
                }
                const unsigned long long bit = 1ull << (pos & 63u);
                if (pos & 64u) bhi |= bit;
                else blo |= bit;
            }
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
    if (BITS) flush_bits(true);
    if (!WRITE) {
        L.d_counts[t] = count;
        if (pos >= L.own_end) *L.d_exit = pos; // exactly one lane's segment crosses the end of the owned range
    }
}

// The records of segment t -- the matches of the chain between the synchronisation points S[t] and the next one -- from
// the bitmap the count pass left: one wave per segment, 64 bitmap words per step (lane i: word i), a wave prefix sum of
// the popcounts gives every record its place behind d_offsets[t]; the records are staged in LDS in order and leave as
// coalesced stores.  Position parallel: this pass runs at memory speed where the serial write pass waited for the chain.
constexpr int kEmitBlock = 256;
constexpr int kEmitCap = 1024; // records staged per wave and step (a denser step goes out in several rounds)

template <typename LenT, int REC>
__global__ __launch_bounds__(kEmitBlock) void k_longest_emit(LongestChainLaunch L, const uint32_t *S) {
    __shared__ int2 st_se[kEmitBlock / kWave][kEmitCap];
    __shared__ int st_id[REC == ACGPU_REC_MAP ? kEmitBlock / kWave : 1][REC == ACGPU_REC_MAP ? kEmitCap : 1];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t t = blockIdx.x * (kEmitBlock / kWave) + wave;
    if (t >= L.n_tiles) return;
    const uint32_t lane = lane_id();
    const uint32_t start = __builtin_amdgcn_readfirstlane(S[t]);
    if (start == ~0u || start >= L.own_end) return;
    uint32_t target = L.own_end; // the next synchronisation point on the chain (the segment ends before it)
    for (uint32_t t2 = t + 1; t2 < L.n_tiles; ++t2) {
        const uint32_t v = __builtin_amdgcn_readfirstlane(S[t2]);
        if (v != ~0u) {
            target = min(v, L.own_end);
            break;
        }
    }
    const LenT *len = reinterpret_cast<const LenT *>(L.d_len);
    uint64_t base = L.d_offsets[t];
    const uint32_t w_first = start >> 5, w_last = (target - 1u) >> 5;
    int2 *se = st_se[wave];
    for (uint32_t w0 = w_first; w0 <= w_last; w0 += kWave) {
        const uint32_t wi = w0 + lane;
        uint32_t bits = wi <= w_last ? L.d_bits[wi] : 0u;
        // positions of this segment only: [start, target)
        if (wi == w_first) bits &= ~0u << (start & 31u);
        if (wi == w_last && (target & 31u)) bits &= ~(~0u << (target & 31u));
        const uint32_t cnt = __popc(bits);
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if ((int)lane >= d) incl += o;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        const uint32_t first = incl - cnt;
        for (uint32_t done = 0; done < total; done += kEmitCap) { // (one round unless the chain is dense)
            uint32_t b = bits, k = first;
            while (b) {
                const uint32_t p = wi * 32u + (uint32_t)__builtin_ctz(b);
                b &= b - 1u;
                if (k >= done && k < done + kEmitCap) {
                    se[k - done] = make_int2((int)p, (int)(p + len_full<LenT>(L, p, (uint32_t)len[p])));
                    if (REC == ACGPU_REC_MAP) st_id[wave][k - done] = (int)L.d_out_id[L.d_state[p]];
                }
                ++k;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t m = min(total - done, (uint32_t)kEmitCap);
            for (uint32_t j = lane; j < m; j += kWave) {
                const uint64_t dst = base + done + j;
                if (dst >= L.cap) break;
                const int2 r = se[j];
                if (REC == ACGPU_REC_SET) {
                    reinterpret_cast<int2 *>(L.d_out)[dst] = r;
                } else {
                    int32_t *o = reinterpret_cast<int32_t *>(L.d_out) + dst * 3;
                    o[0] = r.x; o[1] = r.y; o[2] = st_id[wave][j];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        base += total;
    }
}

// The same with the second bitmap (L.d_ebits: bit end-1 of every match): matches do not overlap, so inside a segment the
// k-th start and the k-th end are one record and no length is looked up (the lookups were, in effect, a second read of
// the whole len[] array).  A step covers 1024 positions: lanes 0-31 take the words of the starts, lanes 32-63 the words of
// the ends; at most one match is open across a step boundary (its start is carried in slot 0).  The last segment of the
// owned range looks for its last end up to max_len positions beyond the range.
template <int REC>
__global__ __launch_bounds__(kEmitBlock, 8) void k_longest_emit_ends(LongestChainLaunch L, const uint32_t *S) {
    // staged positions are relative to the step's first position (16 bits: 4 KB per wave, so that 32 waves share a CU and
    // hide each other's start-up loads); the start carried over a step boundary lives in a scalar register
    constexpr int kCap = 1024 + 8;
    __shared__ uint16_t st_x[kEmitBlock / kWave][kCap], st_y[kEmitBlock / kWave][kCap];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t t = blockIdx.x * (kEmitBlock / kWave) + wave;
    if (t >= L.n_tiles) return;
    const uint32_t lane = lane_id();
    // (the three start-up loads are independent: one memory latency, not three)
    const uint32_t s_here = S[t], s_next = t + 1 < L.n_tiles ? S[t + 1] : ~0u;
    uint64_t base = L.d_offsets[t];
    const uint32_t start = __builtin_amdgcn_readfirstlane(s_here);
    if (start == ~0u || start >= L.own_end) return;
    uint32_t target = L.own_end; // the next synchronisation point on the chain (the segment ends before it)
    {
        uint32_t v = __builtin_amdgcn_readfirstlane(s_next);
        for (uint32_t t2 = t + 2; v == ~0u && t2 < L.n_tiles; ++t2) v = __builtin_amdgcn_readfirstlane(S[t2]);
        if (v != ~0u) target = min(v, L.own_end);
    }
    const uint32_t e_end = target >= L.own_end ? (uint32_t)min((uint64_t)L.len_units, (uint64_t)L.own_end + L.max_len) : target;
    const uint32_t half = lane >> 5; // 0: starts, 1: ends
    const uint32_t *bitmap = half ? L.d_ebits : L.d_bits;
    const uint32_t my_end = half ? e_end : target; // this half looks at positions [start, my_end)
    const uint32_t w_first = start >> 5, w_last_all = (e_end - 1u) >> 5, w_last = (my_end - 1u) >> 5;
    uint16_t *sx = st_x[wave], *sy = st_y[wave];
    uint16_t *mine = half ? sy : sx; // (one store for both halves: no divergent branch in the loop)
    uint32_t carry = 0, carry_start = 0, carry_id = 0; // wave-uniform: a start whose end has not been seen yet
    auto load_bits = [&](uint32_t w0) {
        const uint32_t wi = w0 + (lane & 31u);
        uint32_t bits = wi <= w_last ? bitmap[wi] : 0u;
        if (wi == w_first) bits &= ~0u << (start & 31u);
        if (wi == w_last && (my_end & 31u)) bits &= ~(~0u << (my_end & 31u));
        return bits;
    };
    uint32_t next_bits = load_bits(w_first);
    for (uint32_t w0 = w_first; w0 <= w_last_all; w0 += 32) {
        const uint32_t step_base = w0 * 32u;
        const uint32_t bits = next_bits;
        if (w0 + 32 <= w_last_all) next_bits = load_bits(w0 + 32); // (in flight under this step's staging and stores)
        const uint32_t cnt = __popc(bits);
        const uint32_t incl = wave_inclusive_scan<uint32_t>(cnt);
        const uint32_t total_m = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
        const uint32_t total_e = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1) - total_m;
        uint32_t k = incl - cnt + (half ? 0u - total_m : carry);
        uint32_t b = bits;
        const uint32_t r0 = (lane & 31u) * 32u + half; // starts: p, ends: p + 1 (relative to the step)
        while (b) {
            const uint32_t r = r0 + (uint32_t)__builtin_ctz(b);
            b &= b - 1u;
            if (k < (uint32_t)kCap) { // (always, with consistent bitmaps)
                mine[k] = (uint16_t)r; // (Map records: the keyword's id is fetched when the record is written -- 64 independent gathers, not a chain per lane here)
            }
            ++k;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t m = min(total_e, carry + total_m); // records complete after this step
        for (uint32_t j = lane; j < m; j += kWave) {
            const uint64_t dst = base + j;
            if (dst >= L.cap) break;
            const bool carried = carry && j == 0;
            const int x = (int)(carried ? carry_start : step_base + sx[j]), y = (int)(step_base + sy[j]);
            if (REC == ACGPU_REC_SET) {
                reinterpret_cast<int2 *>(L.d_out)[dst] = make_int2(x, y);
            } else {
                int32_t *o = reinterpret_cast<int32_t *>(L.d_out) + dst * 3;
                o[0] = x; o[1] = y; o[2] = carried ? (int)carry_id : (int)L.d_out_id[L.d_state[x]];
            }
        }
        const uint32_t open = carry + total_m - m; // 0 or 1
        if (open && !(carry && m == 0)) {           // the open start is this step's last one (slot m)
            carry_start = __builtin_amdgcn_readfirstlane(step_base + sx[m]);
            if (REC == ACGPU_REC_MAP) carry_id = __builtin_amdgcn_readfirstlane(L.d_out_id[L.d_state[carry_start]]);
        }
        __builtin_amdgcn_wave_barrier();
        carry = open ? 1u : 0u;
        base += m;
    }
}

// ---- chain passes through LDS --------------------------------------------------------------------------------------------
// k_longest_chain waits for global memory once per 16 positions of its (serial) chain: ~250 dependent round trips per lane,
// 0.46 ms per pass at config 4 with every wave of the grid resident.  Here a lane still follows its own segment, but the
// lengths come through LDS in chunks of 256 positions: a lane requests its whole chunk at once (32 independent 16-byte
// loads: one memory latency per 256 positions instead of one per 16) and then walks it with LDS reads.  The walk is cheap
// enough to run twice -- count, (prefix sum), write -- without the bitmap and the separate emit pass.
// (measured at config 4 with one-byte lengths, whole chain pipeline: 32 pieces / 1 wave per SIMD / 4096-position tiles 0.795 ms;
// 16 pieces / 2 waves per SIMD: 0.75 at 4096-position tiles, 0.65 at 6144, 0.67 at 8192; 8 pieces / 4 waves: 0.70 / 0.68 / 0.73)
#ifndef ACGPU_C2PIECES
#define ACGPU_C2PIECES 16
#endif
constexpr int kC2Pieces = ACGPU_C2PIECES; // 16-byte pieces per lane and chunk: 512 bytes, loaded by one half wave of one LDS-DMA instruction
                              // (global_load_lds_dwordx4: lane i's 16 bytes land at base + 16 i) -- 256 positions of 16-bit
                              // lengths, 512 positions of one-byte lengths

// BITS (count pass): the matches are also marked in the bitmap L.d_bits for k_longest_emit -- collected per chunk in LDS
// (chunks start on a bitmap word) and merged into the zeroed bitmap with one atomicOr per non-zero word; these stores are
// issued when a chunk is done and complete under the next chunk's load.
#ifndef ACGPU_C2WAVES
#define ACGPU_C2WAVES 2
#endif
#ifdef ACGPU_TIMING
__device__ unsigned long long g_c2_timing[8]; // total, zero+issue, load wait, walk, bit stores, chunks (per-wave sums)
#define C2_MARK(i) { const unsigned long long t_ = clock64(); c2t[i] += t_ - c2t0; c2t0 = t_; }
#else
#define C2_MARK(i)
#endif
template <typename LenT, bool WRITE, bool BITS = false>
__global__ __launch_bounds__(kWave, ACGPU_C2WAVES) void k_longest_chain_lds(LongestChainLaunch L, const uint32_t *S) {
    constexpr uint32_t kPP = 16 / sizeof(LenT);            // lengths per piece: 8 or 16
    constexpr uint32_t kPPLog = sizeof(LenT) == 2 ? 3 : 4;
    constexpr int kC2Chunk = kC2Pieces * (int)kPP;         // positions per lane and chunk
    __shared__ __attribute__((aligned(16))) unsigned char buf[kC2Pieces * kWave * 16]; // [piece][lane][16 bytes of lengths]
    __shared__ uint32_t lbits[BITS ? kC2Chunk / 32 : 1][kWave];
    __shared__ uint32_t lebits[BITS ? kC2Chunk / 32 + 1 : 1][kWave]; // the ends (bit end-1), when L.d_ebits is there (+ a dummy word)
    __shared__ int2 ring_se[WRITE ? 8 : 1][kWave];
    __shared__ int ring_id[WRITE ? 4 : 1][kWave];
    const uint32_t lane = threadIdx.x;
    const uint32_t t = blockIdx.x * kWave + lane;
    const LenT *len = reinterpret_cast<const LenT *>(L.d_len);
    uint32_t start = t < L.n_tiles ? S[t] : ~0u;
    bool active = start != ~0u && start < L.own_end;
    if (t < L.n_tiles && !active && !WRITE) {
        L.d_counts[t] = 0;
        if (t == 0) *L.d_exit = L.entry; // entry at/after the end of the owned range
    }
    uint32_t target = ~0u; // the next synchronisation point on the chain
    if (active)
        for (uint32_t t2 = t + 1; t2 < L.n_tiles; ++t2) {
            const uint32_t v = S[t2];
            if (v != ~0u) {
                target = v;
                break;
            }
        }
    const uint32_t limit = min(target, L.own_end);
    const bool set_kind = L.record_kind == ACGPU_REC_SET;
    const uint32_t gmask = set_kind ? 7u : 3u; // records per aligned group - 1
    uint32_t pos = active ? start : 0u, count = 0;
    uint64_t dst = (WRITE && active) ? L.d_offsets[t] : 0;
    uint32_t gfirst = (uint32_t)dst & gmask;
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
    // LDS layout: a lane's chunk is contiguous (kC2Chunk * 2 bytes at lane * kC2Chunk * 2), its 16-byte pieces permuted by
    // piece ^ lane -- the lanes walk their chunks at about the same pace, and without the permutation they would all sit in
    // the same banks.  Length x of the lane's chunk: piece x >> 3, entry x & 7.
    constexpr uint32_t kPieces = kC2Pieces, kChunkBytes = kC2Pieces * 16;
    static_assert(kPieces == 32 || kPieces == 16 || kPieces == 8, "a half (quarter, eighth) wave loads one lane's chunk: 32 (16, 8) pieces of 16 bytes");
    constexpr uint32_t kOwnersPerLoad = kWave / kPieces; // lanes whose chunks one load instruction brings
    const unsigned char *mine = buf + lane * kChunkBytes;
    // (slot = piece ^ lane, i.e. byte offset ((x ^ lane's piece bits) * sizeof(LenT)): two instructions from x to the address)
    const uint32_t swz = (lane & (kPieces - 1u)) << kPPLog;
    auto at = [&](uint32_t x) -> const unsigned char * { return mine + ((x ^ swz) * (uint32_t)sizeof(LenT)); };
    // a chunk starts on a 16-byte boundary of len[]; pieces past its end are read from the last whole piece instead (never
    // consulted: the walk stops at limit <= the end of the owned range; the allocation of len[] has 64 bytes of slack)
    const uint32_t last_piece = L.len_units & ~(kPP - 1u);
    const bool ebits = BITS && L.d_ebits != nullptr;
#ifdef ACGPU_TIMING
    unsigned long long c2t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c2t0 = clock64();
    const unsigned long long c2start = c2t0;
#endif
    uint32_t pend = ~0u; // an end beyond the chunk in which its match started: it lies in the first word of the lane's next chunk
    while (__any(active)) {
        const uint32_t cb = pos & (BITS ? ~31u : ~(kPP - 1u));
        if (BITS) {
#pragma unroll
            for (int w = 0; w < kC2Chunk / 32; ++w) lbits[w][lane] = 0;
            if (ebits) {
#pragma unroll
                for (int w = 0; w < kC2Chunk / 32; ++w) lebits[w][lane] = 0;
            }
        }
        if (ebits && active && pend != ~0u) { // (an end beyond its chunk lies in the first word of the lane's next chunk)
            lebits[0][lane] = 1u << (pend & 31u);
            pend = ~0u;
        }
        {
            // 32 loads straight into LDS, all in flight together.  Load j brings the whole chunks of lanes 2j and 2j+1: one
            // half wave each, 32 consecutive pieces = 512 contiguous bytes per half (every cache line is requested once and
            // used whole; a lane asking for its own 16 bytes per load requested each line four times, from 64 places per
            // instruction: 0.56 ms for the loads alone at config 4)
            const uint32_t my_piece = lane & (kPieces - 1u), sub = lane / kPieces;
#pragma unroll
            for (int j = 0; j < (int)kPieces; ++j) {
                uint32_t cbo = (uint32_t)__builtin_amdgcn_readlane((int)cb, kOwnersPerLoad * j);
#pragma unroll
                for (int o = 1; o < (int)kOwnersPerLoad; ++o) {
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cb, kOwnersPerLoad * j + o);
                    cbo = sub == (uint32_t)o ? c : cbo;
                }
                const uint32_t owner = kOwnersPerLoad * j + sub;
                const uint32_t src = cbo + ((my_piece ^ owner) & (kPieces - 1u)) * kPP;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(len + min(src, last_piece)),
                                                 (__attribute__((address_space(3))) void *)(buf + j * (kWave * 16)), 16, 0, 0);
            }
        }
        C2_MARK(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C2_MARK(2)
        __builtin_amdgcn_wave_barrier();
        if (active) {
            // The chain inside the chunk, on the chunk-relative position `rel` (the loop-carried chain is rel -> address (two
            // instructions) -> length (one LDS read) -> rel + length).  BITS: chunks start on a bitmap word, so bit numbers are
            // rel's low five bits; the ors go to LDS without return; only the LAST match of a chunk can end beyond it -- its
            // bit goes to a dummy word and its position into `pend`.
            const uint32_t rel_end = limit > cb ? min(limit - cb, (uint32_t)kC2Chunk) : 0u;
            uint32_t rel = pos - cb, pend_rel = ~0u;
            while (rel < rel_end) {
                uint32_t l = *reinterpret_cast<const LenT *>(at(rel));
                if (sizeof(LenT) == 1 && l == kLenEscape) l = (uint32_t)L.d_len_big[cb + rel]; // (rare: 255 units and more)
                if (l > 0) {
                    if (BITS) atomicOr(&lbits[rel >> 5][lane], 1u << (rel & 31u));
                    if (ebits) {
                        const uint32_t er = rel + l - 1u;
                        const bool in = er < (uint32_t)kC2Chunk;
                        atomicOr(&lebits[in ? er >> 5 : (uint32_t)(kC2Chunk / 32)][lane], 1u << (er & 31u));
                        pend_rel = in ? pend_rel : er;
                    }
                    if constexpr (WRITE) {
                        pos = cb + rel;
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
                    rel += l;
                } else {
                    // no keyword starts here; from a piece boundary on, whole pieces of eight such positions are skipped at a stroke
                    ++rel;
                    while ((rel & (kPP - 1u)) == 0 && rel + kPP <= rel_end) {
                        const uint4 z = *reinterpret_cast<const uint4 *>(at(rel));
                        if (z.x | z.y | z.z | z.w) break;
                        rel += kPP;
                    }
                }
            }
            pos = cb + rel;
            if (ebits && pend_rel != ~0u) pend = cb + pend_rel;
            C2_MARK(3)
            if (BITS) { // words wholly inside the segment are this lane's own (plain stores); the two at its ends may be shared
                if (ebits && pend != ~0u && ((pend + 1u) & 31u) == 0) { // a word between this lane's chunks: nobody stores it
                    atomicOr(&L.d_ebits[pend >> 5], 0x80000000u);
                    pend = ~0u;
                }
                auto store_bits = [&](const uint32_t (*lb)[kWave], uint32_t *bitmap) {
                    uint32_t bw[kC2Chunk / 32];
#pragma unroll
                    for (int w = 0; w < kC2Chunk / 32; ++w) bw[w] = lb[w][lane];
                    uint32_t *dstw = bitmap + (cb >> 5);
                    if (cb >= start && cb + kC2Chunk <= limit) {
                        uint4 *d4 = reinterpret_cast<uint4 *>(dstw); // (cb is a multiple of 32 positions, not of 128: 4-byte aligned only)
                        if (kC2Chunk >= 128 && (cb & 127u) == 0) {
#pragma unroll
                            for (int g = 0; g < kC2Chunk / 128; ++g) d4[g] = make_uint4(bw[4 * g], bw[4 * g + 1], bw[4 * g + 2], bw[4 * g + 3]);
                        } else {
#pragma unroll
                            for (int w = 0; w < kC2Chunk / 32; ++w) dstw[w] = bw[w];
                        }
                    } else {
#pragma unroll
                        for (int w = 0; w < kC2Chunk / 32; ++w) {
                            const uint32_t wp = cb + 32u * w;
                            if (wp >= start && wp + 32u <= limit) dstw[w] = bw[w];
                            else if (bw[w]) atomicOr(&dstw[w], bw[w]);
                        }
                    }
                };
                store_bits(lbits, L.d_bits);
                if (ebits) store_bits(lebits, L.d_ebits);
            }
            if (pos >= limit) active = false;
        }
        __builtin_amdgcn_wave_barrier();
        C2_MARK(4)
#ifdef ACGPU_TIMING
        c2t[5] += 1;
#endif
    }
#ifdef ACGPU_TIMING
    if (lane == 0) {
        atomicAdd(&g_c2_timing[0], clock64() - c2start);
        for (int i = 1; i < 6; ++i) atomicAdd(&g_c2_timing[i], c2t[i]);
        atomicAdd(&g_c2_timing[6], 1ull);
    }
#endif
    if (t < L.n_tiles && start != ~0u && start < L.own_end) {
        if (WRITE) {
            const uint32_t k = (uint32_t)dst & gmask;
            if (k > gfirst) flush_scalar(dst - k, gfirst, k);
        } else {
            L.d_counts[t] = count;
            if (pos >= L.own_end) *L.d_exit = pos; // exactly one lane's segment crosses the end of the owned range
            if (ebits && pend != ~0u) atomicOr(&L.d_ebits[pend >> 5], 1u << (pend & 31u)); // (the segment ended before that chunk)
        }
    }
}

hipError_t launch_longest_chain_lds(const LongestChainLaunch &l, const uint32_t *d_sync, bool write_pass, hipStream_t stream) {
    if (l.n_tiles == 0) return hipSuccess;
    const dim3 grid((l.n_tiles + kWave - 1) / kWave), block(kWave);
    if (l.len_bytes == 1) {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain_lds<uint8_t, true>), grid, block, 0, stream, l, d_sync);
        else if (l.d_bits) hipLaunchKernelGGL((k_longest_chain_lds<uint8_t, false, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain_lds<uint8_t, false>), grid, block, 0, stream, l, d_sync);
    } else {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain_lds<uint16_t, true>), grid, block, 0, stream, l, d_sync);
        else if (l.d_bits) hipLaunchKernelGGL((k_longest_chain_lds<uint16_t, false, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain_lds<uint16_t, false>), grid, block, 0, stream, l, d_sync);
    }
#ifdef ACGPU_TIMING
    if (!write_pass) {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[8] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c2_timing), sizeof(h));
        if (h[6]) fprintf(stderr, "[c2 timing] waves %llu: total %.0f | zero+issue %.0f | load wait %.0f | walk %.0f | bit stores %.0f | chunks %.1f (cycles per wave)\n",
                          h[6], (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[3] / h[6], (double)h[4] / h[6], (double)h[5] / h[6]);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c2_timing), z, sizeof(z));
    }
#endif
    return hipGetLastError();
}

hipError_t launch_longest_sync(const LongestChainLaunch &l, uint32_t *d_sync, hipStream_t stream) {
    if (l.n_tiles == 0) return hipSuccess;
    const dim3 grid((l.n_tiles + 255) / 256), block(256);
    if (l.len_bytes == 1) hipLaunchKernelGGL((k_longest_sync<uint8_t>), grid, block, 0, stream, l, d_sync);
    else if (l.len_bytes == 2) hipLaunchKernelGGL((k_longest_sync<uint16_t>), grid, block, 0, stream, l, d_sync);
    else hipLaunchKernelGGL((k_longest_sync<uint32_t>), grid, block, 0, stream, l, d_sync);
    return hipGetLastError();
}

hipError_t launch_longest_chain(const LongestChainLaunch &l, const uint32_t *d_sync, bool write_pass, hipStream_t stream) {
    if (l.n_tiles == 0) return hipSuccess;
    const dim3 grid((l.n_tiles + kChainBlock - 1) / kChainBlock), block(kChainBlock);
    if (l.len_bytes == 1) {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain<uint8_t, true>), grid, block, 0, stream, l, d_sync);
        else if (l.d_bits) hipLaunchKernelGGL((k_longest_chain<uint8_t, false, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain<uint8_t, false>), grid, block, 0, stream, l, d_sync);
    } else if (l.len_bytes == 2) {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain<uint16_t, true>), grid, block, 0, stream, l, d_sync);
        else if (l.d_bits) hipLaunchKernelGGL((k_longest_chain<uint16_t, false, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain<uint16_t, false>), grid, block, 0, stream, l, d_sync);
    } else {
        if (write_pass) hipLaunchKernelGGL((k_longest_chain<uint32_t, true>), grid, block, 0, stream, l, d_sync);
        else if (l.d_bits) hipLaunchKernelGGL((k_longest_chain<uint32_t, false, true>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_chain<uint32_t, false>), grid, block, 0, stream, l, d_sync);
    }
    return hipGetLastError();
}

hipError_t launch_longest_emit(const LongestChainLaunch &l, const uint32_t *d_sync, hipStream_t stream) {
    if (l.n_tiles == 0) return hipSuccess;
    const dim3 grid((l.n_tiles + kEmitBlock / kWave - 1) / (kEmitBlock / kWave)), block(kEmitBlock);
    const bool set_kind = l.record_kind == ACGPU_REC_SET;
    if (l.d_ebits) {
        if (set_kind) hipLaunchKernelGGL((k_longest_emit_ends<ACGPU_REC_SET>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_emit_ends<ACGPU_REC_MAP>), grid, block, 0, stream, l, d_sync);
    } else if (l.len_bytes == 1) {
        if (set_kind) hipLaunchKernelGGL((k_longest_emit<uint8_t, ACGPU_REC_SET>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_emit<uint8_t, ACGPU_REC_MAP>), grid, block, 0, stream, l, d_sync);
    } else if (l.len_bytes == 2) {
        if (set_kind) hipLaunchKernelGGL((k_longest_emit<uint16_t, ACGPU_REC_SET>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_emit<uint16_t, ACGPU_REC_MAP>), grid, block, 0, stream, l, d_sync);
    } else {
        if (set_kind) hipLaunchKernelGGL((k_longest_emit<uint32_t, ACGPU_REC_SET>), grid, block, 0, stream, l, d_sync);
        else hipLaunchKernelGGL((k_longest_emit<uint32_t, ACGPU_REC_MAP>), grid, block, 0, stream, l, d_sync);
    }
    return hipGetLastError();
}

} // namespace acgpu
