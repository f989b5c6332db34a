// acgpu_tile.hip -- position-parallel AhoCorasick (all matches) scan for gfx950.
//
// Same results as the reference loop S/AhoCorasickSet.java:204-226 + output walk :522-535, different shape:
// instead of carrying an automaton state along the text (a serial dependent-load chain per lane), every text
// position e is tested independently for "can a keyword END here?":
//
//   filter : the K units before e, mapped to character classes, index a bitmap of all K-suffixes of the
//            dictionary (K <= shortest keyword).  The bitmap lives in LDS (27 classes, K=4: 66 KB); one
//            ds_read_b32 per position, no dependence on any other position.
//   verify : the few surviving positions (about 2 % on the 10k-keyword benchmark dictionary) are compacted,
//            in text order, into a per-wave LDS queue; 64 at a time, one per lane, they walk the trie of REVERSED
//            keywords leftwards from the depth-K node (dense class-indexed rows, or hashed edges for huge
//            alphabets; L2-resident), collecting every keyword that ends at e -- exactly the keywords on the
//            reference's output chain of the state reached at e.
//
//   second level (the L2 form, small range-class dictionaries): between the two, still in LDS, a Bloom filter over
//            the last K+2 units, probed from an LDS copy of the tile's classes, so that almost only true matches reach
//            the verification -- and they arrive with their K-gram index, so that no text window is read for them.
//
// Work distribution: a wave owns a contiguous span of REGIONS of the haystack and streams it as 1024-unit tiles (L2 form:
// 2048-unit), lane l holding 16 (32) consecutive units of a tile, the next tile's loads in flight while one is filtered.
// Because a wave meets its candidates in text order, a record's rank inside its region is a running wave-uniform
// count plus a wave prefix sum; the finalize pass (prefix sum over regions + permutation) then yields the
// reference's emission order (end ascending, longest first) without any sort.  Records go straight to HBM into
// slots the wave reserves 256 at a time with one atomic.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "acgpu_tile_common.h"

namespace acgpu {

int tile_block_threads() { return kTileBlock; }
uint32_t tile_reserve_slots() { return kReserve; }
#ifndef ACGPU_VEC
#define ACGPU_VEC 2
#endif
// AhoCorasick tile geometry: a lane holds kAcVec 16-byte vectors = 8*kAcVec consecutive units of a tile, so the per-tile
// fixed work (prefix sum, cross-lane carry, queue append) is shared by 16 positions per lane instead of 8
constexpr int kAcVec = ACGPU_VEC;
constexpr int kAcLaneUnits = 8 * kAcVec;
constexpr int kAcTileUnits = kWave * kAcLaneUnits;
constexpr int kAcCandCap = kAcTileUnits + kVerifyBatches * kWave;
uint32_t tile_group_units() { return kPrefetch * kTileUnits; }

#ifndef ACGPU_FILTER_WORDS
#define ACGPU_FILTER_WORDS 22016
#endif
constexpr int kFilterWordsMax = ACGPU_FILTER_WORDS; // 88064 bytes of static LDS for the filter rows (tunable filter_max_bytes <= 88000)
constexpr int kFilterWordsSplit = 20224; // the filter-only kernel: 79 KiB, so that two workgroups fit one CU's 160 KiB
bool tile_split_supported(const DevTables &t) {
    return t.filt_k >= 1 && t.filt_words <= (uint32_t)kFilterWordsSplit && !(t.hashk && t.fold_range); // (merged ranges: fused only)
}

// L2 form (second-level filter in LDS, see l2_gram in acgpu_internal.h): smaller static array for the rows, and per wave a
// queue of SURVIVORS (kL2Cap), a copy of the current tile as packed classes behind an 8-unit halo (kTbBytes) and the list
// of the tile's first-level candidates (kL2Fresh tile-relative positions); the Bloom words follow
constexpr int kFilterWordsL2 = 19712;  // 78848 bytes: 27 classes, K = 4
constexpr int kL2Cap = 256;            // a drain leaves fewer than 128; a tile adds at most 128 through the second level
static_assert(kVerifyBatches * kWave + 128 <= kL2Cap, "the queue holds what a drain leaves plus a tile's survivors (-DACGPU_NB=4 hangs)");
constexpr int kL2Fresh = 128;
constexpr int kBigFresh = 704;          // BIG: candidates of one tile a wave lists (the LDS of the Bloom words: 1440 bytes per wave)
static_assert(kBigFresh * 2 * (kTileBlock / kWave) <= kL2Words * 4, "the lists of the large second level live where the Bloom words did");
constexpr int kL2Vec = 4;              // the L2 form takes 32 units per lane: every per-tile cost is shared by 2048 positions
constexpr int kL2TileUnits = kWave * 8 * kL2Vec;
constexpr int kTbBytes = 16 + kL2TileUnits; // one BYTE per class: [8 spare][8 classes before the tile][the tile]
constexpr size_t kL2WaveBytes = kL2Cap * 4 + kL2Cap * 2 + kTbBytes + kL2Fresh * 2; // queue (info + pos16), tile copy, list

// dynamic LDS only: the candidate queues
size_t tile_lds_bytes(const DevTables &t, int block_threads) {
    (void)t;
    return (size_t)(block_threads / kWave) * kAcCandCap * sizeof(uint32_t);
}
size_t tile_l2_lds_bytes(int block_threads) { return (size_t)(block_threads / kWave) * kL2WaveBytes + kL2Words * 4; }

struct __attribute__((packed, aligned(2))) Units8 { // 8 UTF-16 units at any unit address (one global_load_dwordx4)
    uint32_t d[4];
};

// the haystack stream: read once.  `unit` < 2^31, so the byte offset fits 32 bits: uniform base + 32-bit lane offset
// (global_load ... s[base:base+1]) instead of a 64-bit address per lane.  A vector travels as two 64-bit halves so that
// taking over a prefetched group is 8 v_mov_b64 per tile, not 16 v_mov_b32.
struct Vec16 {
    unsigned long long lo, hi;
};
__device__ __forceinline__ Vec16 stream_load(const uint16_t *hay, uint32_t unit) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(hay) + (size_t)(unit * 2u);
    typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
#ifdef ACGPU_STREAM_NT
    const v2ul v = __builtin_nontemporal_load(reinterpret_cast<const v2ul *>(p));
#else
    const v2ul v = *reinterpret_cast<const v2ul *>(p);
#endif
    return Vec16{v.x, v.y};
}

// RANGE: min(unit - base, span), outside [base, base+span) -> span ("other").  Merged stretches (hashk with fold_range): the
// smallest of up to four range classes.  Otherwise the class table.
template <bool RANGE>
__device__ __forceinline__ uint32_t tile_class_t(const DevTables &T, uint32_t unit) {
    if (RANGE) return min(unit - T.cls_base, T.cls_span);
    if (T.hashk && T.fold_range) { // merged stretches: the smallest of the range classes; units beyond the low zone by the table
        if (unit & T.fr_himask) return T.tile_lut[unit];
        return min(min(min(unit - T.fr_base, unit - T.fr_base2), min(unit - T.fr_base3, unit - T.fr_base4)), T.fr_span);
    }
    return T.tile_lut[unit];
}

// flagged ref of the child of reverse-trie node `id` whose edge is `unit` (raw) / class `cls`, 0 = none
__device__ __forceinline__ uint32_t rchild(const DevTables &T, uint32_t id, uint32_t unit, uint32_t cls) {
    if (T.rdense) return T.rtab[id * T.filt_n + cls];
    const uint32_t u = T.cs ? unit : (uint32_t)T.lower[unit];
    const uint32_t r = hashed_goto(T.rhkeys, T.rhvals, T.rhmask, id, u);
    return r == ~0u ? 0u : r;
}

// One step of the leftward walk from a flagged node ref: returns the child's ref or 0.  The only-child hint lets a
// mismatching unit end the walk without a memory access.
template <bool RANGE>
__device__ __forceinline__ uint32_t walk_step(const DevTables &T, uint32_t ref, uint32_t unit) {
    const uint32_t cls = tile_class_t<RANGE>(T, unit);
    const uint32_t hint = (ref >> kRefHintShift) & kRefHintMask;
    if (hint != 0 && hint - 1 != cls) return 0;
    return rchild(T, ref & kRefIdMask, unit, cls);
}

// Verification of up to kVerifyBatches*64 queued candidates: kVerifyBatches per lane, advanced in lock step so that
// their dependent loads (text window -> K-gram node -> rare deeper steps) are in flight together.  Every lane of the
// wave calls this.  Records carry the reversed-trie NODE id; the permute pass translates it to the keyword id.
// HASHK: bucketed tile classes (dictionaries with more than 63 distinct units): the K units themselves, folded and packed
// in text order, are looked up in kg_keys/kg_vals, and every step of the walk goes through the unit-keyed hashed edges.
// QI: queue entries carry the K-gram index and the left neighbour's class (TileCtx::pos16): no text window is read for them.
// SHORTS = false: the dictionary has no keyword shorter than K (compiled out: the config-2 form is register- and SGPR-tight)
template <int K, bool RANGE, bool HASHK = false, bool QI = false, bool SHORTS = true>
__device__ __forceinline__ void verify_multi(TileCtx &c, uint32_t head, uint32_t n_cand) {
    constexpr int NB = kVerifyBatches;
    const DevTables &T = *c.Tp;
    const TileLaunch &L = *c.Lp;
    const uint16_t *hay = L.d_hay;
    const uint32_t lane = lane_id();
    bool act[NB];
    uint32_t e[NB], ref[NB], ref0[NB], child0[NB], left_unit[NB], d[NB], m[NB], one_len[NB], one_node[NB], info[NB], lcls[NB];
    Units8 win[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t q = b * kWave + lane;
        act[b] = q < n_cand;
        if (QI) {
            e[b] = act[b] ? c.pos_base + (uint32_t)c.pos16[head + q] + 1 : 8;
            info[b] = act[b] ? c.cand[head + q] : 0u;
        } else {
            e[b] = act[b] ? c.cand[head + q] + 1 : 8; // exclusive end
            info[b] = 0;
        }
        m[b] = 0; one_len[b] = 0; one_node[b] = 0; d[b] = K; left_unit[b] = 0;
    }
    uint32_t kidx[NB]; // the K-gram index (short keywords are looked up by its last K-1 classes)
    bool need_win = true; // wave-uniform: some candidate comes without its K-gram index
    if (QI) {
        bool unknown = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) unknown |= act[b] && !(info[b] & kQiKnown);
        need_win = __any(unknown);
    }
#ifdef ACGPU_TIMING
    unsigned long long vt0 = clock64();
#define VT_MARK(i) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); c.vt[i] += t_ - vt0; vt0 = t_; }
#else
#define VT_MARK(i)
#endif
    // text windows: units e-8 .. e-1 in one unaligned 16-byte load (zeros before the buffer start)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (!need_win) {
            win[b] = Units8{{0, 0, 0, 0}};
        } else if (e[b] >= 8) {
            if (ACGPU_DBG(L, 8u)) win[b] = Units8{{e[b], e[b] * 3u, e[b] * 5u, e[b] * 7u}}; // ablation: no window load
            else win[b] = *reinterpret_cast<const Units8 *>(hay + e[b] - 8);
        } else {
            win[b] = Units8{{0, 0, 0, 0}}; // within 8 units of the buffer start (rare): unit by unit
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (e[b] + j >= 8) win[b].d[j >> 1] |= (uint32_t)hay[e[b] + j - 8] << (16 * (j & 1));
            }
        }
    }
    VT_MARK(0)
    // K-gram index (last unit least significant) -> flagged ref of the depth-K node of the reversed trie
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint32_t idx = 0;
#pragma unroll
        for (int j = 8 - K; j < 8; ++j) idx = __umul24(idx, T.filt_n) + tile_class_t<RANGE>(T, (win[b].d[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        if (K < 8) left_unit[b] = (win[b].d[(7 - K) >> 1] >> (16 * ((7 - K) & 1))) & 0xffffu;
        lcls[b] = 0;
        if (QI && (info[b] & kQiKnown)) {
            idx = info[b] & kQiIdxMask;
            lcls[b] = (info[b] >> kQiLeftShift) & 31u;
        } else if (K < 8 && need_win) {
            lcls[b] = tile_class_t<RANGE>(T, left_unit[b]);
        }
        kidx[b] = idx;
        uint2 ent = make_uint2(idx & 1u, 0u); // 16: ablation, no K-gram node load
        if (HASHK) {
            ent = make_uint2(0u, 0u);
            if (act[b]) {
                uint64_t key = 0;
#pragma unroll
                for (int j = 8 - K; j < 8; ++j) {
                    const uint32_t u = (win[b].d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    key |= (uint64_t)(T.cs ? u : (uint32_t)T.lower[u]) << (16 * (j - (8 - K)));
                }
                uint32_t slot = edge_hash(key) & T.kg_mask;
                for (;;) {
                    const uint64_t kk = T.kg_keys[slot];
                    if (kk == key) {
                        ent.x = T.kg_vals[slot];
                        break;
                    }
                    if (kk == kEmptyKey) break; // the buckets matched, the units do not
                    slot = (slot + 1) & T.kg_mask;
                }
            }
        } else if (!ACGPU_DBG(L, 16u)) {
            ent = act[b] ? reinterpret_cast<const uint2 *>(T.kgram_node)[idx] : make_uint2(0u, 0u);
        }
        if (!act[b]) ent = make_uint2(0u, 0u);
        ref[b] = ent.x;
        ref0[b] = ent.x;
        child0[b] = ent.y; // the depth-K node's only child, travelling with it
    }
    VT_MARK(1)
    // leftward walks in lock step; every terminal node met is a keyword ending at e (increasing length)
    for (;;) {
        bool go[NB];
        bool any_go = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (ref[b] & kRefTerminal) {
                ++m[b];
                one_len[b] = d[b];
                one_node[b] = ref[b] & kRefIdMask;
            }
            go[b] = (ref[b] & kRefHasChildren) && e[b] > d[b]; // not a leaf, and the buffer does not start here
            any_go |= go[b];
        }
        if (!__any(any_go) || ACGPU_DBG(L, 64u)) break; // 64: ablation, no walk beyond the K-gram node
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t next = 0;
            if (go[b]) {
                const bool first = d[b] == K && K < 8; // the unit in front of the K-gram: known from the window / the queue
                uint32_t unit = 0, cls;
                if (first && QI && (info[b] & kQiKnown)) {
                    cls = lcls[b];
                    if (!T.rdense) unit = (uint32_t)hay[e[b] - 1 - d[b]]; // (hashed edges are keyed by the unit)
                } else if (first) {
                    unit = left_unit[b];
                    cls = lcls[b];
                } else {
                    unit = (uint32_t)hay[e[b] - 1 - d[b]];
                    cls = tile_class_t<RANGE>(T, unit);
                }
                const uint32_t hint = (ref[b] >> kRefHintShift) & kRefHintMask;
                if (!HASHK && d[b] == K && hint != 0) // first step from a one-child node: no memory access at all
                    next = (hint - 1 == cls) ? child0[b] : 0u;
                else if (hint != 0 && hint - 1 != cls) next = 0u; // (walk_step, with the class already in hand)
                else next = rchild(T, ref[b] & kRefIdMask, unit, cls);
            }
            ref[b] = next;
            ++d[b];
        }
    }
    VT_MARK(2)
    // keywords of fewer than K units that end at a candidate (DevTables::kshort; the second-level stage marks the candidates
    // they can end at, an entry without that knowledge is looked up): node + 1 per length 1..3, in the order of the lengths
    uint32_t ms[NB];
    uint4 sh[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        ms[b] = 0;
        sh[b] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (SHORTS && !HASHK && T.kshort != nullptr) { // wave-uniform
        uint32_t grams = 1;
#pragma unroll
        for (int j = 0; j < K - 1; ++j) grams *= T.filt_n;
        bool want[NB], any_want = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            want[b] = act[b] && (!(QI && (info[b] & kQiChecked)) || (info[b] & kQiShort));
            any_want |= want[b];
        }
        if (__any(any_want)) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (want[b]) sh[b] = reinterpret_cast<const uint4 *>(T.kshort)[kidx[b] % grams];
                if (e[b] < 1u) sh[b].x = 0; // (a keyword does not begin before the buffer does)
                if (e[b] < 2u) sh[b].y = 0;
                if (e[b] < 3u) sh[b].z = 0;
                ms[b] = (sh[b].x != 0) + (sh[b].y != 0) + (sh[b].z != 0);
            }
        }
    }
    if (SHORTS && HASHK && T.ks_keys != nullptr) { // bucketed / merged classes: the short keywords by their units (the window is loaded)
        bool want[NB], any_want = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            want[b] = act[b] && (!(QI && (info[b] & kQiChecked)) || (info[b] & kQiShort));
            any_want |= want[b];
        }
        if (__any(any_want)) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint32_t found[3] = {0u, 0u, 0u};
                if (want[b]) {
#pragma unroll
                    for (uint32_t len = 1; len <= 3; ++len) {
                        if (len >= (uint32_t)K || e[b] < len) continue;
                        uint64_t key = (uint64_t)len << 48;
                        for (uint32_t t = 0; t < len; ++t) {
                            const uint32_t j = 8 - len + t; // unit e - len + t of the window
                            const uint32_t u = (win[b].d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                            key |= (uint64_t)(T.cs ? u : (uint32_t)T.lower[u]) << (16 * t);
                        }
                        uint32_t slot = edge_hash(key) & T.ks_mask;
                        for (;;) {
                            const uint64_t kk = T.ks_keys[slot];
                            if (kk == key) {
                                found[len - 1] = T.ks_vals[slot];
                                break;
                            }
                            if (kk == kEmptyKey) break;
                            slot = (slot + 1) & T.ks_mask;
                        }
                    }
                }
                sh[b] = make_uint4(found[0], found[1], found[2], 0u);
                ms[b] = (found[0] != 0) + (found[1] != 0) + (found[2] != 0);
            }
        }
    }
    // record slots and ranks in text order: batch 0's lanes, then batch 1's, ...
    uint32_t prefix[NB], total = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t incl = wave_inclusive_scan_dpp(m[b] + ms[b]);
        prefix[b] = total + incl - (m[b] + ms[b]);
        total += __builtin_amdgcn_readlane(incl, kWave - 1);
    }
    if (total == 0 || ACGPU_DBG(L, 32u)) return; // 32: ablation, no record emission
    const SlotRange sr = reserve_slots(c, total);
    auto slot_of = [&](uint32_t k) -> uint32_t { return sr.slot(k); };
    bool multi = false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (m[b] == 1) store_rec(L, slot_of(prefix[b]), e[b] - one_len[b], e[b], one_node[b], c.rank_base + prefix[b]);
        multi |= m[b] >= 2;
    }
    if (__any(multi)) {
        // several keywords end at one position: the reference reports the longest first (S/AhoCorasickSet.java:526-532),
        // the walk meets them shortest first -> second walk, giving the j-th one met the (m-1-j)-th place
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (m[b] >= 2) {
                uint32_t r = ref0[b], dd = K, j = 0;
                for (;;) {
                    if (r & kRefTerminal) {
                        const uint32_t k = prefix[b] + (m[b] - 1 - j);
                        store_rec(L, slot_of(k), e[b] - dd, e[b], r & kRefIdMask, c.rank_base + k);
                        if (++j == m[b]) break;
                    }
                    if (!(r & kRefHasChildren) || e[b] <= dd) break;
                    r = walk_step<RANGE>(T, r, hay[e[b] - 1 - dd]);
                    if (r == 0) break;
                    ++dd;
                }
            }
        }
    }
    // the short keywords come last (longest first): of them, too, the longer before the shorter
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (ms[b] == 0) continue;
        const uint32_t ids[3] = {sh[b].x, sh[b].y, sh[b].z};
        uint32_t r = 0;
#pragma unroll
        for (uint32_t len = 1; len <= 3; ++len) {
            if (ids[len - 1] == 0) continue;
            const uint32_t k = prefix[b] + m[b] + (ms[b] - 1 - r);
            store_rec(L, slot_of(k), e[b] - len, e[b], ids[len - 1] - 1, c.rank_base + k);
            ++r;
        }
    }
    c.rank_base += total;
    VT_MARK(3)
}

// drain the candidate queue down to fewer than `keep_below` entries
template <int K, bool RANGE, bool HASHK, bool QI = false, bool SHORTS = true>
__device__ __forceinline__ void drain(TileCtx &c, uint32_t keep_below) {
    uint32_t head = 0;
    while (c.cand_n > head && c.cand_n - head >= keep_below) {
        const uint32_t nb = min(c.cand_n - head, (uint32_t)(kVerifyBatches * kWave));
        if (!ACGPU_DBG(*c.Lp, 1u)) verify_multi<K, RANGE, HASHK, QI, SHORTS>(c, head, nb);
        head += nb;
    }
    if (head) { // move the leftovers (fewer than kVerifyBatches*64) to the front
        const uint32_t left = c.cand_n - head;
        uint32_t tmp[kVerifyBatches], tmp16[kVerifyBatches];
#pragma unroll
        for (int b = 0; b < kVerifyBatches; ++b) {
            const uint32_t q = b * kWave + lane_id();
            tmp[b] = q < left ? c.cand[head + q] : 0u;
            tmp16[b] = (QI && q < left) ? (uint32_t)c.pos16[head + q] : 0u;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < kVerifyBatches; ++b) {
            const uint32_t q = b * kWave + lane_id();
            if (q < left) c.cand[q] = tmp[b];
            if (QI && q < left) c.pos16[q] = (uint16_t)tmp16[b];
        }
        __builtin_amdgcn_wave_barrier();
        c.cand_n = left;
    }
}

// hs*n + ROWB*cl_new - cl_old*nK1s in three full-rate VALU ops (every factor < 2^24, exact modulo 2^32)
template <uint32_t ROWB>
__device__ __forceinline__ uint32_t roll_row(uint32_t hs, uint32_t n, uint32_t cl_old, int neg_nK1s, uint32_t cl_new) {
    uint32_t t, u, r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(t) : "v"(hs), "s"(n));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(u) : "v"(cl_old), "s"(neg_nK1s), "v"(t));
    if (ROWB == 4) asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(cl_new), "v"(u));
    else asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(cl_new), "v"(u));
    return r;
}


// Packed 16-bit filter arithmetic (PK form of the filter): two units per register all the way.
//   classes  : min(unit - base, span) for both halves            v_pk_sub_u16 + v_pk_min_u16   (2 ops per 2 units)
//   row index: Horner over the (K-1)-gram, both positions at once  v_pk_mad_u16                 (K-2 ops per 2 positions)
// (usable when the row index n^(K-1) fits 16 bits; every intermediate is exact modulo 2^16)
#ifdef ACGPU_PK_C
typedef unsigned short pk_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_u16x2 pk_v(uint32_t x) { return __builtin_bit_cast(pk_u16x2, x); }
__device__ __forceinline__ uint32_t pk_u(pk_u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_class(uint32_t units2, uint32_t base2, uint32_t span2) {
    return pk_u(__builtin_elementwise_min(pk_v(units2) - pk_v(base2), pk_v(span2)));
}
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t n2, uint32_t c) { return pk_u(pk_v(a) * pk_v(n2) + pk_v(c)); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_min(pk_v(a), pk_v(b))); }
#else
__device__ __forceinline__ uint32_t pk_class(uint32_t units2, uint32_t base2, uint32_t span2) {
    uint32_t t, r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t) : "v"(units2), "s"(base2));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(t), "s"(span2));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t n2, uint32_t c) {
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(n2), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif

// The fused tail (TileLaunch::fused_tail): what k_permute_wg did in a launch of its own, by the scan's workgroups themselves.
// Every wave of the workgroup calls it when its span is scanned.  A record's rank counts through its wave's whole span, so
// its final index is (records of the workgroups before this one) + (records of the workgroup's waves before its wave) + rank:
// nothing but the sixteen wave totals and the slice's fill mark is handed over, through LDS -- the tail is a chain of memory
// round trips behind the slowest workgroup's scan, and every word that stays in LDS is one of them less.
__device__ __forceinline__ void tile_fused_tail(const TileLaunch &L, uint32_t wg, uint32_t *wg_words, uint32_t wave_total, uint32_t res_cur,
                                                uint32_t res_left, uint32_t slot_limit) {
    const uint32_t lane = lane_id(), wave = threadIdx.x / kWave;
    constexpr uint32_t kWaves = kTileBlock / kWave;
    const ScratchRec *slice = L.d_scratch + (size_t)wg * L.slice_slots;
    // the wave's unused reservation: holes the loop below skips; the slice is filled up to the end of the last reservation
    for (uint32_t i = lane; i < res_left; i += kWave)
        if (res_cur + i < slot_limit) store_rec(L, res_cur + i, 0, 0, 0, ~0u);
    if (lane == 0) {
        wg_words[2 + wave] = wave_total;
        if (slot_limit != 0) atomicMax(&wg_words[1], min(res_cur + res_left, slot_limit) - wg * L.slice_slots); // (0: the wave never reserved)
    }
    // every record of this workgroup has been written when all of its waves are here (the stores of a wave are acknowledged by
    // the cache all of its CU's waves read through)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t m = wg_words[1];
    const uint32_t wv = lane < kWaves ? wg_words[2 + lane] : 0u;
    const uint32_t wv_incl = wave_inclusive_scan_dpp(wv);
    const uint32_t wv_excl = wv_incl - wv;                                  // lane w: records of the waves before wave w
    const uint32_t mine = __builtin_amdgcn_readlane(wv_incl, kWaves - 1);  // the workgroup's records
    if (threadIdx.x == 0) ft_publish(L, wg, mine);
    // Where this workgroup's records begin: the records of every workgroup that started before it (ft_below), asked for only
    // when the first records of the slice are in registers: a workgroup that is done before the ones it waits for has read
    // its slice by the time their counts arrive.
    uint32_t below = 0;
    bool have_below = false;
    auto wait_below = [&]() {
        below = ft_below(L, wg, wg_words);
        have_below = true;
    };
    // the slice, every slot below its fill mark: (wave, rank) -> final index
    const uint32_t base8 = L.own_begin & ~7u;
    const uint32_t span_units = L.region_units * L.regions_per_wave, w0g = wg * kWaves;
    constexpr uint32_t kPer = 8; // records per thread and step: their loads (and id lookups) go out together
    for (uint32_t i0 = 0; i0 < m; i0 += kPer * kTileBlock) {
        uint4 raw[kPer];
        uint32_t id[kPer], dst[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint32_t i = i0 + q * kTileBlock + threadIdx.x;
            raw[q] = make_uint4(0u, 8u, 0u, ~0u);
            if (i < m) raw[q] = *reinterpret_cast<const uint4 *>(&slice[i]);
        }
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const bool hole = raw[q].w == ~0u; // beyond the mark, or left by a slot reservation
            id[q] = raw[q].z;
            if (L.out_map && L.d_id_map && !hole) id[q] = L.d_id_map[raw[q].z];
            const uint32_t w = hole ? 0u : (raw[q].y - 1u - base8) / span_units - w0g; // the wave that owns the match's last unit
            dst[q] = (uint32_t)__shfl((int)wv_excl, (int)(w & (kWaves - 1))) + raw[q].w;
            if (hole) dst[q] = ~0u;
        }
        if (!have_below) wait_below(); // (m is the same for every wave: all of them come here, or none)
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint32_t at = below + dst[q];
            if (dst[q] == ~0u || at >= L.out_cap) continue;
            if (L.out_map) {
                typedef int32_t v3i __attribute__((ext_vector_type(3)));
                const v3i rec = {(int32_t)raw[q].x, (int32_t)raw[q].y, (int32_t)id[q]};
                *reinterpret_cast<v3i *>(reinterpret_cast<int32_t *>(L.d_out) + (size_t)at * 3) = rec;
            } else {
                reinterpret_cast<int2 *>(L.d_out)[at] = make_int2((int)raw[q].x, (int)raw[q].y);
            }
        }
    }
    if (!have_below) wait_below();
    ft_report(L, wg, (unsigned long long)below + mine);
}

// A wave owns a contiguous SPAN of regions.  Region boundaries sit at base8 + r * region_units (base8 = own_begin
// rounded down to 8 units, region_units a multiple of the 2048-unit tile group), so a tile group never straddles two
// regions and the tile stream -- with its double-buffered register groups and the cross-lane carry -- runs through the
// whole span.
//
// SPLIT: the filter-only form.  Nothing is verified here: the candidates of the wave's span go, in text order, to the
// wave's slice of L.d_cands (the "queue" is that slice and is never drained), with a {first index, count} pair per
// region for k_ac_verify.  No LDS besides the filter rows.
// NR4: merged stretches with three or four ranges (DevTables::fr_nr > 2; with PK only)
// BIG (with L2): large dictionaries -- the second level is the SAME Bloom structure at 2 MB in global memory (DevTables::l2_big,
// resident in the L2 cache) instead of 22.5 KB in LDS, which 100 k keywords saturate; the LDS that held the Bloom words holds a
// list of up to kBigFresh first-level candidates per wave and tile, tested four batches at a time (their cached reads in flight
// together); only the survivors -- about one candidate in fifty instead of nearly all -- reach the gather-bound verification
template <int K, bool RANGE, bool WIDE, bool SPLIT, bool HASHK = false, bool PK = false, bool L2 = false, bool NR4 = false, bool SHORTS = true,
          bool BIG = false>
__global__ __launch_bounds__(kTileBlock) void k_ac_tile(DevTables T, TileLaunch L) {
    static_assert(!BIG || L2, "the large second level is a form of the L2 kernel");
    const bool has_short = SHORTS && T.has_short != 0; // (SHORTS = false: the launcher knows there is none)
    // the filter rows are STATIC LDS (offset 0, so a scaled row index is the ds_read address with nothing to add);
    // the candidate queues are the dynamic part behind it
    __shared__ __attribute__((aligned(16))) uint32_t rows32[SPLIT ? kFilterWordsSplit : (L2 ? kFilterWordsL2 : kFilterWordsMax)];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t wg_words[kFtWords]; // fused tail: the workgroup's number, its slice's fill mark, its waves' records
    const bool FT = !SPLIT && L.fused_tail != 0; // kernel-uniform
    if (FT && threadIdx.x == 0) ft_take_number(L, wg_words); // (the answer is there when the tables are: the barrier below)
    const unsigned char *rows8 = reinterpret_cast<const unsigned char *>(rows32);
    uint32_t *cand_all = reinterpret_cast<uint32_t *>(smem);
    { // the filter rows, 16 bytes per thread and step (the kernel does not stream before this is done)
        const uint32_t n4 = T.filt_words / 4;
        const uint4 *src4 = reinterpret_cast<const uint4 *>(T.filt_bits);
        uint4 *dst4 = reinterpret_cast<uint4 *>(rows32);
        for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) dst4[i] = src4[i];
        for (uint32_t i = n4 * 4 + threadIdx.x; i < T.filt_words; i += blockDim.x) rows32[i] = T.filt_bits[i];
    }
    // LUT forms with the scalar filter: the class table as pages behind the rows, when it fits (acgpu_build.cpp 7b)
    const uint32_t pg_off = (T.filt_words * 4u + 15u) & ~15u;
    const bool cls_lds = !RANGE && !PK && T.cls_pages != nullptr && (uint64_t)pg_off + T.cls_pages_bytes <= sizeof(rows32) && !ACGPU_DBG(L, 1u << 15); // (ablation build, tile_debug bit 32768: the class table in global memory on the same tables)
    if (cls_lds)
        for (uint32_t i = threadIdx.x; i < T.cls_pages_bytes / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(rows32)[pg_off / 16 + i] = reinterpret_cast<const uint4 *>(T.cls_pages)[i];
    const unsigned char *pg8 = rows8 + pg_off;
    constexpr int D2 = K + 2 < 6 ? K + 2 : 6; // depth of the second-level filter
    // tile geometry of this variant (the names hide the namespace-scope defaults)
    constexpr int kAcVec = L2 ? kL2Vec : acgpu::kAcVec;
    constexpr int kAcLaneUnits = 8 * kAcVec;
    constexpr int kAcTileUnits = kWave * kAcLaneUnits;
    constexpr int kAcTiles = kPrefetch / kAcVec;
    static_assert(kAcTiles >= 1 && kAcLaneUnits <= 32, "tile geometry");
    constexpr int kAcCandCap = kAcTileUnits + kVerifyBatches * kWave;
    constexpr uint32_t kQueueCap = L2 ? kL2Cap : kAcCandCap;
    // (readfirstlane: the wave index and everything derived from it -- spans, tile positions, queue counts -- is then
    // scalar for the compiler too: bookkeeping on the SALU, scalar branches instead of exec masking; -1.6 % at config 2)
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // L2: [queues: info][queues: pos16][tile buffers][fresh lists][Bloom words]
    constexpr int kWavesPerBlock = kTileBlock / kWave;
    uint16_t *pos16_all = reinterpret_cast<uint16_t *>(smem + kWavesPerBlock * kL2Cap * 4);
    unsigned char *tb = smem + kWavesPerBlock * kL2Cap * 6 + wave_in_block * kTbBytes;
    uint16_t *fresh = reinterpret_cast<uint16_t *>(smem + kWavesPerBlock * (kL2Cap * 6 + kTbBytes)) + wave_in_block * kL2Fresh;
    uint32_t *bloom = reinterpret_cast<uint32_t *>(smem + (kTileBlock / kWave) * kL2WaveBytes);
    if (L2 && !BIG) for (uint32_t i = threadIdx.x; i < kL2Words / 4; i += blockDim.x)
        reinterpret_cast<uint4 *>(bloom)[i] = reinterpret_cast<const uint4 *>(T.l2_bloom)[i];
    uint16_t *bigl = reinterpret_cast<uint16_t *>(bloom) + wave_in_block * kBigFresh; // BIG: the tile's candidates, text order
    static_assert(kL2Words % 4 == 0, "Bloom words are copied 16 bytes at a time");
    __syncthreads();

    const uint32_t lane = lane_id();
    // (the fused tail orders the workgroups by their start, not by blockIdx: the number names span, slice and counters)
    const uint32_t wg = FT ? __builtin_amdgcn_readfirstlane(wg_words[0]) : blockIdx.x;
    const uint32_t wave_global = wg * (kTileBlock / kWave) + wave_in_block;
    TileCtx c{&T, &L, SPLIT ? L.d_cands + (size_t)wave_global * L.cands_per_wave : cand_all + wave_in_block * kQueueCap, 0, 0, 0u, 0};
    c.wg = wg;
    if (L2) c.pos16 = pos16_all + wave_in_block * kL2Cap;
    uint32_t lane0 = 0; // L2: lanes below it have been enqueued already (a dense tile taken in pieces); wave-uniform
    const uint32_t slice_base = SPLIT ? wave_global * L.cands_per_wave : 0u; // (< 2^32: the host sizes the slices)
    uint32_t region_first = 0; // SPLIT: index (inside the slice) of the current region's first candidate

    constexpr uint32_t ROWB = WIDE ? 8 : 4;
    const uint32_t n = T.filt_n;
    uint32_t nK1s = ROWB; // ROWB * n^(K-1): weight of the unit that leaves the (K-1)-gram window
#pragma unroll
    for (int i = 0; i < K - 1; ++i) nK1s *= n;
    const int neg_nK1s = -(int)nK1s;
    // dwords of the previous 8 units that hold the K-1 (L2: D2-1) units before the lane's first one
    constexpr int NP = L2 ? D2 / 2 : K / 2;

    const uint32_t first_region = wave_global * L.regions_per_wave;
    const bool has_work = first_region < L.n_regions; // wave-uniform
    if (!has_work && !FT) return; // (fused tail: a wave without a region passes through the loop and joins the workgroup's tail)
    const uint32_t last_region = min(first_region + L.regions_per_wave, L.n_regions);
    const uint32_t base8 = L.own_begin & ~7u;
    const uint32_t R = L.region_units;
    const uint32_t span_begin = max(L.own_begin, base8 + first_region * R);
    uint32_t span_end = base8 + last_region * R;
    if (span_end > L.own_end || last_region == L.n_regions) span_end = L.own_end;
    // full 16-byte vectors end at nfull; the (at most 7) units behind it are handled after the tile stream, so the
    // stream's loads are unconditional (clamped address)
    const uint32_t nfull = L.n_units & ~7u;
    const uint32_t hi = min(span_end, nfull);
    const uint32_t last_vec = nfull >= 8 ? nfull - 8 : 0;
    const uint16_t *hay = L.d_hay;

    uint32_t wave_total = 0; // wave-uniform: records of the regions this wave has finished
    uint32_t region = first_region;
    uint32_t boundary = base8 + (region + 1) * R; // first tile of the next region
    uint32_t rb = span_begin;
    uint32_t re = min(span_end, boundary);
    c.pos_base = boundary - R; // L2: queue positions are relative to the start of the current region
    uint32_t tile = base8 + first_region * R;

    bool vec_todo = has_work && tile < hi;          // tile groups left in the vector part of the span
    bool tail_todo = has_work && span_end > nfull;  // the units behind the last full vector of the buffer (fewer than 8)
    uint32_t d0 = 0;                    // tile of the current group to resume at (after a mid-group drain)
    uint32_t prio_turn = 0;             // passes so far (issue priority rotation)
    bool mid = false;                   // the current group is being resumed (its registers are live, its loads are out)
    uint32_t carry[4] = {0, 0, 0, 0};
    Vec16 nxt[kAcTiles][kAcVec], grp[kAcTiles][kAcVec];
#pragma unroll
    for (int d = 0; d < kAcTiles; ++d)
#pragma unroll
        for (int u = 0; u < kAcVec; ++u) nxt[d][u] = grp[d][u] = Vec16{0ull, 0ull};
    if (vec_todo) {
        if (K > 1 && tile >= 8) {
            const uint4 p = *reinterpret_cast<const uint4 *>(hay + tile - 8);
            carry[0] = p.x; carry[1] = p.y; carry[2] = p.z; carry[3] = p.w;
        }
        // double-buffered tile groups: while the tiles of the current group are filtered out of registers, the loads of
        // the next group are in flight (8 KiB per wave); they are awaited together at the next group's start
#pragma unroll
        for (int d = 0; d < kAcTiles; ++d)
#pragma unroll
            for (int u = 0; u < kAcVec; ++u)
                nxt[d][u] = stream_load(hay, min(tile + d * kAcTileUnits + lane * kAcLaneUnits + u * 8, last_vec));
    }

#ifdef ACGPU_TIMING
    // instrumented build (tools/build_variant.sh timing -DACGPU_TIMING): where a wave's time goes, in s_memtime ticks
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // 0 total, 1 wait for the stream, 2 drains, 3 filter + second level, 4 passes, 5 drain calls
    const unsigned long long tm_start = clock64();
#define TM_BEGIN const unsigned long long tm_t0 = clock64()
#define TM_END(i) tm[i] += clock64() - tm_t0
#else
#define TM_BEGIN
#define TM_END(i)
#endif
    // One loop, ONE verification site: every pass first drains the candidate queue as far as its state requires
    // (completely at a region seam / before the tail / at the end; down to < 256 otherwise), then does one unit of
    // streaming work.  Keeping the (large) verification code in a single place keeps the kernel small.
    for (;;) {
        const bool seam = vec_todo ? (!mid && tile >= boundary) : true; // wave-uniform
        const uint32_t keep = SPLIT ? ~0u : (seam ? 1u : (uint32_t)(kVerifyBatches * kWave));
        if (vec_todo && !mid) {
#ifndef ACGPU_NO_SETPRIO
            // The issue arbiter favours the oldest wave of a SIMD: with equal spans the four waves of a SIMD (slots s, s+4,
            // s+8, s+12 of the workgroup) finish 513 k, 531 k, 557 k and 603 k ticks after the start, and the kernel ends
            // with the youngest.  Every wave therefore rotates its own issue priority from tile to tile (s_setprio), so
            // that each holds every rank equally often: 535 / 553 / 559 / 580 k, kernel -1.5 % (no-verify build -4.6 %).
            // (A feedback form -- priority = number of SIMD siblings ahead, progress published in LDS -- evened the slots
            // out completely and made every one of them 29 % slower.)
            switch ((wave_in_block / 4u + prio_turn++) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
#endif
#ifdef ACGPU_TIMING
            { TM_BEGIN; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TM_END(1); tm[4]++; }
#endif
            // take over the prefetched group FIRST: this wait also covers the record stores of the previous pass's
            // verification (gfx950 counts stores in vmcnt), which have had a whole tile group of time to finish
#pragma unroll
            for (int d = 0; d < kAcTiles; ++d)
#pragma unroll
                for (int u = 0; u < kAcVec; ++u) grp[d][u] = nxt[d][u];
        }
        if (!SPLIT && c.cand_n >= keep && c.cand_n != 0) {
            TM_BEGIN;
            drain<K, RANGE, HASHK, L2, SHORTS>(c, keep);
            TM_END(2);
#ifdef ACGPU_TIMING
            tm[5]++;
#endif
        }

        if (vec_todo) {
            if (!mid) {
                if (tile >= boundary) { // the stream enters the next region (regions hold whole tile groups)
                    if (SPLIT) {
                        if (lane == 0) L.d_region_cands[region] = make_uint2(slice_base + region_first, c.cand_n - region_first);
                        region_first = c.cand_n;
                    } else if (lane == 0 && !FT) {
                        L.d_region_counts[region] = c.rank_base;
                    }
                    if (!FT) { // (fused tail: a record's rank counts through the wave's whole span -- no region counts)
                        wave_total += c.rank_base;
                        c.rank_base = 0;
                    }
                    ++region;
                    rb = boundary;
                    c.pos_base = boundary;
                    boundary += R;
                    re = min(span_end, boundary);
                }
                // the next group's loads go out AFTER the verification, so that its gathers never wait for them
#pragma unroll
                for (int d = 0; d < kAcTiles; ++d)
#pragma unroll
                    for (int u = 0; u < kAcVec; ++u)
                        nxt[d][u] = stream_load(hay, min(tile + (kAcTiles + d) * kAcTileUnits + lane * kAcLaneUnits + u * 8, last_vec));
            }
#ifdef ACGPU_TIMING
            const unsigned long long tm_f0 = clock64();
#endif
            // positions a lane may report: inside the region, in the vector part of the buffer, with K units to their
            // left in the buffer.  Only groups at the edges of a region need the per-lane mask.
            // (with short keywords every position counts: the units before the buffer are zeros in the carry, and the wild
            // cards of the short keywords cover whatever class those get)
            const uint32_t lo = has_short ? rb : max(rb, (uint32_t)(K - 1));
            const uint32_t top = min(re, hi);
            const bool edge = tile < lo || tile + kAcTiles * kAcTileUnits > top; // wave-uniform
            bool resume = false;
#pragma unroll
            for (int d = 0; d < kAcTiles; ++d) {
                if ((uint32_t)d < d0) continue; // wave-uniform
                const uint32_t cur = tile + d * kAcTileUnits;
                if (cur >= hi) break; // wave-uniform
                if (SPLIT) {
                    if (c.cand_n + kAcTileUnits > L.cands_per_wave) { // the slice is too small for this haystack
                        if (lane == 0) atomicOr(L.d_overflow, 1u);
                        return; // wave-uniform; the host redoes the call with the fused kernel
                    }
                } else if (!L2 && c.cand_n > kAcCandCap - kAcTileUnits) { // rare: the queue cannot take a worst-case tile -> drain first
                    d0 = d;
                    resume = true;
                    break;
                }
                uint32_t carry_in[4]; // L2 decides after the filter whether the tile fits the queue: a redo starts from here
#pragma unroll
                for (int q = 0; q < 4; ++q) carry_in[q] = carry[q];
                const uint32_t v = cur + lane * kAcLaneUnits;
                uint32_t ww[4 * kAcVec];
#pragma unroll
                for (int u = 0; u < kAcVec; ++u) {
                    ww[4 * u + 0] = (uint32_t)grp[d][u].lo; ww[4 * u + 1] = (uint32_t)(grp[d][u].lo >> 32);
                    ww[4 * u + 2] = (uint32_t)grp[d][u].hi; ww[4 * u + 3] = (uint32_t)(grp[d][u].hi >> 32);
                }
                // the previous lane's last dwords give the K-1 units before v; lane 0 takes the previous tile's lane 63
                uint32_t pp[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 4 - NP; q < 4; ++q) {
                    const uint32_t mine = ww[4 * (kAcVec - 1) + q];
                    pp[q] = from_prev_lane(mine, carry[q]);
                    carry[q] = __builtin_amdgcn_readlane(mine, 63);
                }
                uint32_t mask = 0;
                if (ACGPU_DBG(L, 4u)) { // ablation: stream only
                    uint32_t x = 0;
#pragma unroll
                    for (int q = 0; q < 4 * kAcVec; ++q) x ^= ww[q];
                    mask = (x == 0x12345678u) ? 1u : 0u;
                } else if (PK) {
                    // dword D of the 8 previous units (D < 4) and of the lane's own units (D >= 4) as packed classes;
                    // MM[D] = the pair of units (2D-1, 2D), i.e. the misaligned neighbour of CC[D]
                    constexpr int ND = 4 + 4 * kAcVec;
                    // RANGE: class = min(unit - base, span).  Otherwise folded range classes (DevTables::fold_range): the
                    // smaller of that and the same thing for the partner range; a tile in which some unit has a bit of
                    // fr_himask (the few other units that fold into the range all do) takes the class table instead
                    const uint32_t base2 = (RANGE ? T.cls_base : T.fr_base) * 0x10001u, span2 = (RANGE ? T.cls_span : T.fr_span) * 0x10001u;
                    const uint32_t baseB2 = T.fr_base2 * 0x10001u, n2 = n * 0x10001u;
                    const uint32_t baseC2 = T.fr_base3 * 0x10001u, baseD2 = T.fr_base4 * 0x10001u;
                    bool by_table = false;
                    if (!RANGE && T.fr_himask != 0) {
                        uint32_t any_bits = 0;
#pragma unroll
                        for (int D = 4 - NP; D < 4; ++D) any_bits |= pp[D];
#pragma unroll
                        for (int q = 0; q < 4 * kAcVec; ++q) any_bits |= ww[q];
                        by_table = __any((any_bits & (T.fr_himask * 0x10001u)) != 0);
                    }
                    auto classes_of = [&](uint32_t units2) -> uint32_t {
                        if (RANGE) return pk_class(units2, base2, span2);
                        if (by_table) return (uint32_t)T.tile_lut[units2 & 0xffffu] | ((uint32_t)T.tile_lut[units2 >> 16] << 16);
                        const uint32_t c1 = pk_class(units2, base2, span2), c2 = pk_class(units2, baseB2, span2);
                        if (!NR4) return pk_min(c1, c2);
                        const uint32_t c3 = pk_class(units2, baseC2, span2), c4 = pk_class(units2, baseD2, span2);
                        return pk_min(pk_min(c1, c2), pk_min(c3, c4));
                    };
                    uint32_t CC[ND], MM[ND], B8[ND / 2];
#pragma unroll
                    for (int D = 4 - NP; D < 4; ++D) CC[D] = classes_of(pp[D]);
#pragma unroll
                    for (int D = 4 - NP + 1; D < 4; ++D) MM[D] = __builtin_amdgcn_alignbit(CC[D], CC[D - 1], 16);
                    if (L2) { // classes of the 8 units before the tile, one byte each (see below)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            B8[i] = (2 * i + 1 >= 4 - NP) ? __builtin_amdgcn_perm(CC[2 * i + 1], 2 * i >= 4 - NP ? CC[2 * i] : 0u, 0x06040200u) : 0u;
                        if (lane == 0) *reinterpret_cast<uint2 *>(tb + 8) = make_uint2(B8[0], B8[1]);
                    }
                    uint32_t acc = 0;
                    // one 16-byte vector (8 positions) at a time, start to finish, so that only a few packed classes are
                    // live at any point (the L2 form sits at the 128-register limit)
#pragma unroll
                    for (int u = 0; u < kAcVec; ++u) {
#pragma unroll
                        for (int D = 4 + 4 * u; D < 8 + 4 * u; ++D) {
                            CC[D] = classes_of(ww[D - 4]);
                            MM[D] = __builtin_amdgcn_alignbit(CC[D], CC[D - 1], 16);
                        }
#pragma unroll
                        for (int D = 4 + 4 * u; D < 8 + 4 * u; ++D) {
                            // positions 2D (low half) and 2D+1 (high half): row = the K-1 units before, bit = own class
                            const int S0 = 2 * D - (K - 1); // first unit of the low position's (K-1)-gram
                            uint32_t H = (S0 & 1) ? MM[(S0 + 1) / 2] : CC[S0 / 2];
#pragma unroll
                            for (int t = 1; t < K - 1; ++t) {
                                const int S = S0 + t;
                                H = pk_mad(H, n2, (S & 1) ? MM[(S + 1) / 2] : CC[S / 2]);
                            }
                            // (8192: ablation, timing only -- every lane reads a word of its own bank, 32 rows around the right one:
                            // what the rows' bank conflicts cost)
                            const uint32_t row_lo = rows32[ACGPU_DBG(L, 8192u) ? ((H & 0xffe0u) | (lane & 31u)) : (H & 0xffffu)];
                            const uint32_t row_hi = rows32[ACGPU_DBG(L, 8192u) ? (((H >> 16) & 0xffe0u) | (lane & 31u)) : (H >> 16)];
                            acc = __builtin_amdgcn_alignbit(row_lo >> (CC[D] & 31u), acc, 1);
                            acc = __builtin_amdgcn_alignbit(row_hi >> ((CC[D] >> 16) & 31u), acc, 1);
                        }
                        if (L2) { // the tile as one byte per class (v_perm_b32 takes bytes 0 and 2 of two registers)
                            B8[2 + 2 * u] = __builtin_amdgcn_perm(CC[5 + 4 * u], CC[4 + 4 * u], 0x06040200u);
                            B8[3 + 2 * u] = __builtin_amdgcn_perm(CC[7 + 4 * u], CC[6 + 4 * u], 0x06040200u);
                            if (u & 1)
                                reinterpret_cast<uint4 *>(tb + 16 + lane * kAcLaneUnits)[u / 2] =
                                    make_uint4(B8[2 * u], B8[2 * u + 1], B8[2 * u + 2], B8[2 * u + 3]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    mask = kAcLaneUnits == 32 ? acc : acc >> ((32 - kAcLaneUnits) & 31);
                    if (edge) { // wave-uniform, rare (the empty asm keeps this a branch: if-converted it is 16 VALU in every tile)
                        asm volatile("" ::: "memory");
                        const uint32_t first = lo > v ? min(lo - v, (uint32_t)kAcLaneUnits) : 0u;
                        const uint32_t last = top > v ? min(top - v, (uint32_t)kAcLaneUnits) : 0u;
                        mask &= (uint32_t)((1ull << last) - 1ull) & ~(uint32_t)((1ull << first) - 1ull);
                    }
                } else {
                    // classes of units v-(K-1) .. v+kAcLaneUnits-1
                    uint32_t a[kAcLaneUnits + K - 1];
                    if (!RANGE && cls_lds) { // (wave-uniform) two LDS reads per unit: page index, class byte
                        auto cls_of = [&](uint32_t unit) -> uint32_t { return pg8[256u + ((uint32_t)pg8[unit >> 8] << 8) + (unit & 255u)]; };
#pragma unroll
                        for (int j = 0; j < K - 1; ++j) {
                            const int u = 8 - (K - 1) + j;
                            a[j] = cls_of((pp[u >> 1] >> (16 * (u & 1))) & 0xffffu);
                        }
#pragma unroll
                        for (int j = 0; j < kAcLaneUnits; ++j) a[K - 1 + j] = cls_of((ww[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                    } else {
#pragma unroll
                    for (int j = 0; j < K - 1; ++j) {
                        const int u = 8 - (K - 1) + j; // unit index inside the previous 8
                        a[j] = tile_class_t<RANGE>(T, (pp[u >> 1] >> (16 * (u & 1))) & 0xffffu);
                    }
#pragma unroll
                    for (int j = 0; j < kAcLaneUnits; ++j)
                        a[K - 1 + j] = tile_class_t<RANGE>(T, (ww[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                    }
                    // byte offset of the filter row of position v+j = ROWB * index of its (K-1)-gram a[j .. j+K-2],
                    // rolling; every factor is < 2^24, so the full-rate 24-bit multiplies are exact modulo 2^32
                    uint32_t hs = 0;
#pragma unroll
                    for (int j = 0; j < K - 1; ++j) hs = __umul24(hs, n) + a[j] * ROWB;
#pragma unroll
                    for (int j = 0; j < kAcLaneUnits; ++j) {
                        if (j > 0 && K > 1) hs = roll_row<ROWB>(hs, n, a[j - 1], neg_nK1s, a[j + K - 2]);
                        uint32_t bit;
                        if (WIDE) {
                            const uint64_t row = *reinterpret_cast<const uint64_t *>(rows8 + hs);
                            bit = (uint32_t)(row >> a[K - 1 + j]) & 1u;
                        } else {
                            const uint32_t row = *reinterpret_cast<const uint32_t *>(rows8 + hs);
                            bit = __builtin_amdgcn_ubfe(row, a[K - 1 + j], 1); // v_bfe_u32 takes the offset from a[4:0]
                        }
                        mask |= bit << j;
                    }
                    if (edge) {
                        const uint32_t first = lo > v ? min(lo - v, (uint32_t)kAcLaneUnits) : 0u;
                        const uint32_t last = top > v ? min(top - v, (uint32_t)kAcLaneUnits) : 0u;
                        mask &= ((1u << last) - 1u) & ~((1u << first) - 1u);
                    }
                }
                if (ACGPU_DBG(L, 512u)) mask &= 0x0101u << (lane & 7u); // ablation: one candidate in eight survives (timing only)
                if (!L2) {
                    enqueue(c, mask, v);
                    continue;
                }
                // ---- second-level filter: only candidates whose last D2 units still look like a keyword are queued ----
                if (lane < lane0) mask = 0;
                uint32_t cnt = __popc(mask);
                uint32_t incl = wave_inclusive_scan_dpp(cnt);
                uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
                if (total == 0) {
                    lane0 = 0;
                    continue;
                }
                struct __attribute__((packed, aligned(1), may_alias)) Win8b { uint32_t lo, hi; }; // classes of units p-7 .. p
                if (BIG && total <= (uint32_t)kBigFresh) {
                    { // the tile's candidates, tile relative, text order
                        uint32_t slot = incl - cnt, m = mask;
                        while (__any(m != 0)) {
                            if (m != 0) {
                                bigl[slot++] = (uint16_t)(lane * kAcLaneUnits + (uint32_t)__builtin_ctz(m));
                                m &= m - 1;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const unsigned char *tb8 = tb + 16;
                    const uint32_t *bigtab = T.l2_big;
                    uint32_t survivors = 0;
                    for (uint32_t b = 0; b < total; b += 4u * kWave) { // four batches: their (cached) reads are in flight together
                        uint32_t word[4], hh[4], pp[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t k = b + (uint32_t)q * kWave + lane;
                            pp[q] = k < total ? (uint32_t)bigl[k] : 0u;
                            const Win8b w = *reinterpret_cast<const Win8b *>(tb8 + (int)pp[q] - 7);
                            const uint32_t cl[4] = {w.hi >> 24, (w.hi >> 16) & 0xffu, (w.hi >> 8) & 0xffu, w.hi & 0xffu};
                            hh[q] = l2_hash(K == 4 ? w.hi : l2_gram(cl, K));
                            word[q] = bigtab[l2_word_big(hh[q])]; // (a lane beyond the list reads the word of position 0: cached)
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t k = b + (uint32_t)q * kWave + lane;
                            const bool act = k < total;
                            const Win8b w = *reinterpret_cast<const Win8b *>(tb8 + (int)pp[q] - 7);
                            const uint32_t cls[6] = {w.hi >> 24, (w.hi >> 16) & 0xffu, (w.hi >> 8) & 0xffu, w.hi & 0xffu, w.lo >> 24, (w.lo >> 16) & 0xffu};
                            const uint32_t pat = l2_pattern(hh[q]);
                            bool pass = false;
#pragma unroll
                            for (int len = K; len <= D2; ++len) {
                                const uint32_t bits = __builtin_amdgcn_alignbit(pat, pat, l2_rot(cls, len, K));
                                pass |= (word[q] & bits) == bits;
                            }
                            if (has_short) { // a keyword of fewer than K units ends here (see below)
                                uint32_t hrow = T.filt_other;
#pragma unroll
                                for (int j = K - 2; j >= 1; --j) hrow = hrow * n + cls[j];
                                pass = pass || ((rows32[hrow] >> cls[0]) & 1u);
                            }
                            pass = pass && act;
                            if (act) bigl[k] = (uint16_t)(pp[q] | (pass ? 0x8000u : 0u));
                            survivors += (uint32_t)__popcll(__ballot(pass));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (survivors <= 128u && c.cand_n + survivors <= (uint32_t)kL2Cap) {
                        for (uint32_t b = 0; b < total; b += kWave) {
                            const uint32_t k = b + lane;
                            const uint32_t e = k < total ? (uint32_t)bigl[k] : 0u;
                            const bool pass = (e & 0x8000u) != 0;
                            const uint32_t p = e & 0x7fffu;
                            const uint64_t bal = __ballot(pass);
                            if (bal == 0) continue; // wave-uniform
                            if (pass) {
                                const Win8b w = *reinterpret_cast<const Win8b *>(tb8 + (int)p - 7);
                                const uint32_t cls[6] = {w.hi >> 24, (w.hi >> 16) & 0xffu, (w.hi >> 8) & 0xffu, w.hi & 0xffu, w.lo >> 24, (w.lo >> 16) & 0xffu};
                                uint32_t shortbit = 0;
                                if (has_short) {
                                    uint32_t hrow = T.filt_other;
#pragma unroll
                                    for (int j = K - 2; j >= 1; --j) hrow = hrow * n + cls[j];
                                    shortbit = (rows32[hrow] >> cls[0]) & 1u;
                                }
                                const uint32_t at = c.cand_n + (uint32_t)__popcll(bal & lanemask_lt());
                                uint32_t idx = cls[K - 1];
#pragma unroll
                                for (int j = K - 2; j >= 0; --j) idx = __umul24(idx, n) + cls[j];
                                c.pos16[at] = (uint16_t)(cur + p - c.pos_base);
                                c.cand[at] = HASHK ? (kQiChecked | (shortbit ? kQiShort : 0u)) : kQiKnown | kQiChecked | (shortbit ? kQiShort : 0u) | (cls[K] << kQiLeftShift) | idx;
                            }
                            c.cand_n += (uint32_t)__popcll(bal);
                        }
                        __builtin_amdgcn_wave_barrier();
                        lane0 = 0;
                        continue;
                    }
                    // The survivors do not fit the queue.  What has been learnt is kept: every lane's mask shrinks to its survivors,
                    // and the ways below -- a drain first and the tile once more, or as many whole lanes as fit -- see only those
                    // (a tile with more than 128 TRUE candidates used to hand all of its first-level candidates to the verification)
                    {
                        uint32_t keep = 0;
                        for (uint32_t j = 0; j < cnt; ++j) {
                            const uint32_t e = bigl[incl - cnt + j];
                            if (e & 0x8000u) keep |= 1u << ((e & 0x7fffu) - lane * kAcLaneUnits);
                        }
                        __builtin_amdgcn_wave_barrier();
                        mask = keep;
                        cnt = __popc(mask);
                        incl = wave_inclusive_scan_dpp(cnt);
                        total = __builtin_amdgcn_readlane(incl, kWave - 1);
                    }
                }
                if (!BIG && total <= (uint32_t)kL2Fresh && c.cand_n + total <= (uint32_t)kL2Cap) {
                    uint32_t slot = incl - cnt, m = mask;
                    while (__any(m != 0)) { // tile-relative positions, text order
                        if (m != 0) {
                            fresh[slot++] = (uint16_t)(lane * kAcLaneUnits + (uint32_t)__builtin_ctz(m));
                            m &= m - 1;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    struct __attribute__((packed, aligned(1), may_alias)) Win8 { uint32_t lo, hi; }; // classes of units p-7 .. p
                    const unsigned char *tb8 = tb + 16; // tb8[p] = class of the tile's unit p
                    for (uint32_t b = 0; b < total; b += kWave) {
                        const uint32_t k = b + lane;
                        const bool act = k < total;
                        const uint32_t p = act ? (uint32_t)fresh[k] : 0u;
                        const Win8 w = *reinterpret_cast<const Win8 *>(tb8 + (int)p - 7);
                        const uint32_t cls[6] = {w.hi >> 24, (w.hi >> 16) & 0xffu, (w.hi >> 8) & 0xffu, w.hi & 0xffu, w.lo >> 24, (w.lo >> 16) & 0xffu};
                        // l2_gram(cls, K): one byte per class, text[e-1] highest -- for K = 4 that is the window's high word as it stands
                        const uint32_t h = l2_hash(K == 4 ? w.hi : l2_gram(cls, K));
                        const uint32_t word = bloom[l2_word(h)];
                        const uint32_t pat = l2_pattern(h);
                        bool pass = false;
#pragma unroll
                        for (int len = K; len <= D2; ++len) {
                            const uint32_t bits = __builtin_amdgcn_alignbit(pat, pat, l2_rot(cls, len, K));
                            pass |= (word & bits) == bits;
                        }
                        // a keyword of fewer than K units: only their wild cards set a bit in a row whose leading class is "other"
                        uint32_t shortbit = 0;
                        if (has_short) { // wave-uniform
                            uint32_t hrow = T.filt_other;
#pragma unroll
                            for (int j = K - 2; j >= 1; --j) hrow = hrow * n + cls[j];
                            shortbit = (rows32[hrow] >> cls[0]) & 1u;
                        }
                        pass = (pass || shortbit || ACGPU_DBG(L, 4096u)) && act; // 4096: ablation, the second level passes everything
                        const uint64_t bal = __ballot(pass);
                        if (pass) { // the entry: position in the region, K-gram index (oldest unit most significant), left class
                            const uint32_t at = c.cand_n + (uint32_t)__popcll(bal & lanemask_lt());
                            uint32_t idx = cls[K - 1]; // (v_mad_u32_u24: the compiler's own choice here is the quarter-rate 64-bit mad)
#pragma unroll
#ifdef ACGPU_PK_C
                            for (int j = K - 2; j >= 0; --j) idx = __umul24(idx, n) + cls[j];
#else
                            for (int j = K - 2; j >= 0; --j) asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(idx) : "v"(idx), "s"(n), "v"(cls[j]));
#endif
                            c.pos16[at] = (uint16_t)(cur + p - c.pos_base);
                            // (HASHK: a class K-gram does not name a K-gram of units -- the verification reads the window)
                            c.cand[at] = HASHK ? (kQiChecked | (shortbit ? kQiShort : 0u)) : kQiKnown | kQiChecked | (shortbit ? kQiShort : 0u) | (cls[K] << kQiLeftShift) | idx;
                        }
                        c.cand_n += (uint32_t)__popcll(bal);
                    }
                    __builtin_amdgcn_wave_barrier();
                    lane0 = 0;
                    continue;
                }
                if (c.cand_n >= (uint32_t)(kVerifyBatches * kWave)) { // make room first (drain at the loop top), then redo the tile
#pragma unroll
                    for (int q = 0; q < 4; ++q) carry[q] = carry_in[q];
                    d0 = d;
                    resume = true;
                    break;
                }
                { // dense tile: no second level; as many whole lanes as fit go straight to the queue, the rest after a drain
                    const uint32_t room = (uint32_t)kL2Cap - c.cand_n; // > 16: a lane holds at most 16 candidates
                    const uint32_t nl = (uint32_t)__popcll(__ballot(incl <= room)); // incl is monotone: lanes [0, nl) fit
                    uint32_t slot = c.cand_n + incl - cnt, m = lane < nl ? mask : 0u;
                    while (__any(m != 0)) {
                        if (m != 0) {
                            c.pos16[slot] = (uint16_t)(v + (uint32_t)__builtin_ctz(m) - c.pos_base);
                            c.cand[slot++] = 0u; // (no K-gram index: the verification reads the text window)
                            m &= m - 1;
                        }
                    }
                    c.cand_n += __builtin_amdgcn_readlane(incl, nl - 1);
                    __builtin_amdgcn_wave_barrier();
                    if (nl < (uint32_t)kWave) {
                        lane0 = nl;
#pragma unroll
                        for (int q = 0; q < 4; ++q) carry[q] = carry_in[q];
                        d0 = d;
                        resume = true;
                        break;
                    }
                    lane0 = 0;
                }
            }
#ifdef ACGPU_TIMING
            tm[3] += clock64() - tm_f0;
#endif
            mid = resume;
            if (!resume) {
                d0 = 0;
                tile += kAcTiles * kAcTileUnits;
                vec_todo = tile < hi;
            }
            continue;
        }
        if (tail_todo) { // the queue is empty here (seam drain above)
            tail_todo = false;
            const uint32_t t0 = max(nfull, span_begin);
            if (t0 >= boundary) { // the tail opens a new region
                if (SPLIT) {
                    if (lane == 0) L.d_region_cands[region] = make_uint2(slice_base + region_first, c.cand_n - region_first);
                    region_first = c.cand_n;
                } else if (lane == 0 && !FT) {
                    L.d_region_counts[region] = c.rank_base;
                }
                if (!FT) {
                    wave_total += c.rank_base;
                    c.rank_base = 0;
                }
                ++region;
                c.pos_base = boundary;
            }
            if (SPLIT && c.cand_n + kWave > L.cands_per_wave) {
                if (lane == 0) atomicOr(L.d_overflow, 1u);
                return;
            }
            const uint32_t pos = t0 + lane;
            uint32_t mask = 0;
            if (pos < span_end && (pos + 1 >= (uint32_t)K || has_short) && !ACGPU_DBG(L, 4u)) {
                uint32_t hrow = 0;
                for (int j = K - 1; j >= 1; --j) hrow = hrow * n + (pos >= (uint32_t)j ? tile_class_t<RANGE>(T, hay[pos - j]) : T.filt_other);
                const uint32_t last = tile_class_t<RANGE>(T, hay[pos]);
                const uint32_t word = rows32[hrow * (ROWB / 4) + (last >> 5)];
                mask = (word >> (last & 31)) & 1u;
            }
            if (L2) { // (queue entries without a K-gram index: the verification reads their text window)
                const uint64_t bal = __ballot(mask != 0);
                if (mask != 0) {
                    const uint32_t at = c.cand_n + (uint32_t)__popcll(bal & lanemask_lt());
                    c.pos16[at] = (uint16_t)(pos - c.pos_base);
                    c.cand[at] = 0u;
                }
                c.cand_n += (uint32_t)__popcll(bal);
                __builtin_amdgcn_wave_barrier();
            } else {
                enqueue(c, mask, pos); // one position per lane: lane order is text order
            }
            continue;
        }
        break;
    }
    if (SPLIT) {
        if (lane == 0) L.d_region_cands[region] = make_uint2(slice_base + region_first, c.cand_n - region_first);
        return;
    }
#ifdef ACGPU_TIMING
    tm[0] = clock64() - tm_start;
    if (lane == 0 && L.d_timing)
    {
        for (int i = 0; i < 8; ++i) L.d_timing[(size_t)wave_global * 8 + i] = tm[i];
        L.d_timing[(size_t)wave_global * 8 + 6] = c.vt[0] | (c.vt[1] << 32);
        L.d_timing[(size_t)wave_global * 8 + 7] = c.vt[2] | (c.vt[3] << 32);
    }
#endif
    if (lane == 0 && has_work && !FT) L.d_region_counts[region] = c.rank_base;
    // the workgroup's record count, next to its slot counter: the fused permute pass (k_permute_wg) turns the per-workgroup
    // sums and the region counts into offsets itself, so no prefix-sum kernels run between the scan and the permutation
    wave_total += c.rank_base;
    if (lane == 0 && L.wg_sums && wave_total) atomicAdd(L.d_counter + (size_t)wg * kCounterStride + 1, (unsigned long long)wave_total);
    if (FT) {
        tile_fused_tail(L, wg, wg_words, wave_total, c.res_cur, c.res_left, c.slot_limit);
        return;
    }
    // hand back the unused tail of the last reservation as holes the permute pass skips
    for (uint32_t i = lane; i < c.res_left; i += kWave)
        if (c.res_cur + i < c.slot_limit) store_rec(L, c.res_cur + i, 0, 0, 0, ~0u);
}

// The verification half of the split form: one region per wave (grid-stride), its candidates read from L.d_cands in
// batches of kVerifyBatches*64, records ranked inside the region exactly as the fused kernel ranks them.
template <int K, bool RANGE, bool HASHK = false>
__global__ __launch_bounds__(256) void k_ac_verify(DevTables T, TileLaunch L) {
    const uint32_t lane = lane_id();
    const uint32_t waves = gridDim.x * (blockDim.x / kWave);
    // a candidate slice overflowed: the region table is incomplete and the host redoes the call with the fused kernel
    if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile const uint32_t *>(L.d_overflow)) != 0) return;
    TileCtx c{&T, &L, nullptr, 0, 0, 0u, 0};
    for (uint32_t region = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; region < L.n_regions; region += waves) {
        const uint2 rc = L.d_region_cands[region];
        c.cand = L.d_cands + rc.x;
        c.cand_n = rc.y;
        c.rank_base = 0;
        for (uint32_t head = 0; head < rc.y; head += kVerifyBatches * kWave)
            verify_multi<K, RANGE, HASHK>(c, head, min(rc.y - head, (uint32_t)(kVerifyBatches * kWave)));
        if (lane == 0) L.d_region_counts[region] = c.rank_base;
    }
    for (uint32_t i = lane; i < c.res_left; i += kWave)
        if (c.res_cur + i < c.slot_limit) store_rec(L, c.res_cur + i, 0, 0, 0, ~0u); // (not into the next slice)
}

template <int K, bool RANGE, bool WIDE>
static hipError_t launch_tile_variant(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<K, RANGE, WIDE, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
    if (e != hipSuccess) return e;
    ACGPU_LAUNCH_EV((k_ac_tile<K, RANGE, WIDE, false>), dim3(l.grid), dim3(l.block), l.lds_bytes, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}

template <int K, bool RANGE, bool WIDE>
static hipError_t launch_filter_variant(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    ACGPU_LAUNCH_EV((k_ac_tile<K, RANGE, WIDE, true>), dim3(l.grid), dim3(l.block), 0, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}

template <int K>
static hipError_t launch_filter_k(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    const bool wide = t.filt_row_bytes == 8;
    if (t.range_cls) return wide ? launch_filter_variant<K, true, true>(t, l, stream) : launch_filter_variant<K, true, false>(t, l, stream);
    return wide ? launch_filter_variant<K, false, true>(t, l, stream) : launch_filter_variant<K, false, false>(t, l, stream);
}

template <int K>
static hipError_t launch_verify_k(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    if (t.range_cls) hipLaunchKernelGGL((k_ac_verify<K, true>), dim3(l.verify_grid), dim3(256), 0, stream, t, l);
    else hipLaunchKernelGGL((k_ac_verify<K, false>), dim3(l.verify_grid), dim3(256), 0, stream, t, l);
    return hipGetLastError();
}

// bucketed classes: LUT classes, 8-byte rows, K <= 3
template <int K, bool SPLIT>
static hipError_t launch_tile_hashk(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    if (!SPLIT) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<K, false, true, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
        if (e != hipSuccess) return e;
        ACGPU_LAUNCH_EV((k_ac_tile<K, false, true, false, true>), dim3(l.grid), dim3(l.block), l.lds_bytes, stream, l.ev_start, l.ev_stop, t, l);
    } else {
        ACGPU_LAUNCH_EV((k_ac_tile<K, false, true, true, true>), dim3(l.grid), dim3(l.block), 0, stream, l.ev_start, l.ev_stop, t, l);
    }
    return hipGetLastError();
}

// the packed 16-bit filter: range classes, 4-byte rows, row index below 2^16 (l.debug & 1024 keeps the scalar filter: A/B)
static bool tile_pk_usable(const DevTables &t, const TileLaunch &l) {
    if (!(t.range_cls || t.fold_range) || t.filt_row_bytes != 4 || t.filt_k < 2 || (t.hashk && !t.fold_range) || (l.debug & 1024u)) return false;
    uint64_t rows = 1;
    for (uint32_t i = 0; i + 1 < t.filt_k; ++i) rows *= t.filt_n;
    if (rows > 65536) return false;
    if (t.fold_range) return t.fr_base + t.fr_span <= 65536 && t.fr_base2 + t.fr_span <= 65536 &&
                             (t.fr_nr <= 2 || (t.fr_base3 + t.fr_span <= 65536 && t.fr_base4 + t.fr_span <= 65536));
    return t.cls_base + t.cls_span <= 65536;
}

// two merged ranges (DevTables::hashk with fold_range): the packed two-range filter, the verification by units; K <= 4
// the large second level (K = 4, DevTables::l2_big): tile_debug bit 2^30 keeps the LDS form for A/B
static bool tile_big_usable(const DevTables &t, const TileLaunch &l) { return t.l2_big != nullptr && t.filt_k == 4 && !(l.debug & (1u << 30)); }

template <bool NR4>
static hipError_t launch_tile_merged_big(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    const size_t lds = tile_l2_lds_bytes(l.block);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<4, false, false, false, true, true, true, NR4, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    ACGPU_LAUNCH_EV((k_ac_tile<4, false, false, false, true, true, true, NR4, true, true>), dim3(l.grid), dim3(l.block), lds, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}

template <int K, bool L2, bool NR4>
static hipError_t launch_tile_merged(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    const size_t lds = L2 ? tile_l2_lds_bytes(l.block) : l.lds_bytes;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<K, false, false, false, true, true, L2, NR4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    ACGPU_LAUNCH_EV((k_ac_tile<K, false, false, false, true, true, L2, NR4>), dim3(l.grid), dim3(l.block), lds, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}
template <int K>
static hipError_t launch_tile_merged_k(const DevTables &t, const TileLaunch &l, bool l2, hipStream_t stream) {
    if (K == 4 && l2 && tile_big_usable(t, l)) return t.fr_nr > 2 ? launch_tile_merged_big<true>(t, l, stream) : launch_tile_merged_big<false>(t, l, stream);
    if (t.fr_nr > 2) return l2 ? launch_tile_merged<K, true, true>(t, l, stream) : launch_tile_merged<K, false, true>(t, l, stream);
    return l2 ? launch_tile_merged<K, true, false>(t, l, stream) : launch_tile_merged<K, false, false>(t, l, stream);
}

template <int K, bool RANGE>
static hipError_t launch_tile_pk(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<K, RANGE, false, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
    if (e != hipSuccess) return e;
    ACGPU_LAUNCH_EV((k_ac_tile<K, RANGE, false, false, false, true>), dim3(l.grid), dim3(l.block), l.lds_bytes, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}

// the PK form with the second-level filter (l.debug & 2048 keeps the one-level form: A/B)
static bool tile_l2_usable(const DevTables &t, const TileLaunch &l) {
    uint64_t grams = 1; // queue entries hold the K-gram index in 20 bits and the position in its region in 16
    for (uint32_t i = 0; i < t.filt_k; ++i) grams *= t.filt_n;
    return tile_pk_usable(t, l) && t.l2_bloom != nullptr && t.l2_depth != 0 && t.filt_k <= 5 && grams <= (1u << 20) &&
           l.region_units <= 65536u && t.filt_words <= (uint32_t)kFilterWordsL2 && !(l.debug & 2048u);
}

template <int K, bool RANGE, bool SHORTS>
static hipError_t launch_tile_l2s(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    const size_t lds = tile_l2_lds_bytes(l.block);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<K, RANGE, false, false, false, true, true, false, SHORTS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    ACGPU_LAUNCH_EV((k_ac_tile<K, RANGE, false, false, false, true, true, false, SHORTS>), dim3(l.grid), dim3(l.block), lds, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}
template <bool RANGE, bool SHORTS>
static hipError_t launch_tile_l2_big(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    const size_t lds = tile_l2_lds_bytes(l.block);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<4, RANGE, false, false, false, true, true, false, SHORTS, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    ACGPU_LAUNCH_EV((k_ac_tile<4, RANGE, false, false, false, true, true, false, SHORTS, true>), dim3(l.grid), dim3(l.block), lds, stream, l.ev_start, l.ev_stop, t, l);
    return hipGetLastError();
}
template <int K, bool RANGE>
static hipError_t launch_tile_l2(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    if (K == 4 && tile_big_usable(t, l)) return t.has_short ? launch_tile_l2_big<RANGE, true>(t, l, stream) : launch_tile_l2_big<RANGE, false>(t, l, stream);
    // (short keywords imply K <= 4: the K = 5 form needs no SHORTS instantiation)
    if (K <= 4 && t.has_short) return launch_tile_l2s<K, RANGE, (K <= 4)>(t, l, stream);
    return launch_tile_l2s<K, RANGE, false>(t, l, stream);
}

template <int K>
static hipError_t launch_tile_k(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    const bool wide = t.filt_row_bytes == 8;
    if (t.range_cls) return wide ? launch_tile_variant<K, true, true>(t, l, stream) : launch_tile_variant<K, true, false>(t, l, stream);
    return wide ? launch_tile_variant<K, false, true>(t, l, stream) : launch_tile_variant<K, false, false>(t, l, stream);
}

hipError_t launch_ac_tile(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name) {
    static thread_local char name[64];
    if (t.filt_k < 1 || t.filt_k > 8) return hipErrorInvalidValue;
    std::snprintf(name, sizeof(name), "k_ac_tile<%u, %s, %s, false>", t.filt_k, t.range_cls ? "true" : "false",
                  t.filt_row_bytes == 8 ? "true" : "false");
    if (kernel_name) *kernel_name = name;
    if (t.hashk && t.fold_range) {
        if (!tile_pk_usable(t, l)) return hipErrorInvalidValue; // (the builder chooses this form only where it is)
        const bool l2 = tile_l2_usable(t, l);
        std::snprintf(name, sizeof(name), "k_ac_tile<%u, false, false, false, true, true, %s, %s>", t.filt_k, l2 ? "true" : "false",
                      t.fr_nr > 2 ? "true" : "false");
        if (l2 && tile_big_usable(t, l))
            std::snprintf(name, sizeof(name), "k_ac_tile<4, false, false, false, true, true, true, %s, true, true>", t.fr_nr > 2 ? "true" : "false");
        switch (t.filt_k) {
        case 2: return launch_tile_merged_k<2>(t, l, l2, stream);
        case 3: return launch_tile_merged_k<3>(t, l, l2, stream);
        case 4: return launch_tile_merged_k<4>(t, l, l2, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (t.hashk) {
        std::snprintf(name, sizeof(name), "k_ac_tile<%u, false, true, false, true>", t.filt_k);
        switch (t.filt_k) {
        case 1: return launch_tile_hashk<1, false>(t, l, stream);
        case 2: return launch_tile_hashk<2, false>(t, l, stream);
        case 3: return launch_tile_hashk<3, false>(t, l, stream);
        default: return hipErrorInvalidValue;
        }
    }
    const char *rg = t.range_cls ? "true" : "false";
    if (tile_l2_usable(t, l)) {
        std::snprintf(name, sizeof(name), "k_ac_tile<%u, %s, false, false, false, true, true>", t.filt_k, rg);
        if (tile_big_usable(t, l))
            std::snprintf(name, sizeof(name), "k_ac_tile<4, %s, false, false, false, true, true, false, %s, true>", rg, t.has_short ? "true" : "false");
        switch (t.filt_k) {
        case 2: return t.range_cls ? launch_tile_l2<2, true>(t, l, stream) : launch_tile_l2<2, false>(t, l, stream);
        case 3: return t.range_cls ? launch_tile_l2<3, true>(t, l, stream) : launch_tile_l2<3, false>(t, l, stream);
        case 4: return t.range_cls ? launch_tile_l2<4, true>(t, l, stream) : launch_tile_l2<4, false>(t, l, stream);
        case 5: return t.range_cls ? launch_tile_l2<5, true>(t, l, stream) : launch_tile_l2<5, false>(t, l, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (tile_pk_usable(t, l)) {
        std::snprintf(name, sizeof(name), "k_ac_tile<%u, %s, false, false, false, true>", t.filt_k, rg);
        switch (t.filt_k) {
        case 2: return t.range_cls ? launch_tile_pk<2, true>(t, l, stream) : launch_tile_pk<2, false>(t, l, stream);
        case 3: return t.range_cls ? launch_tile_pk<3, true>(t, l, stream) : launch_tile_pk<3, false>(t, l, stream);
        case 4: return t.range_cls ? launch_tile_pk<4, true>(t, l, stream) : launch_tile_pk<4, false>(t, l, stream);
        case 5: return t.range_cls ? launch_tile_pk<5, true>(t, l, stream) : launch_tile_pk<5, false>(t, l, stream);
        case 6: return t.range_cls ? launch_tile_pk<6, true>(t, l, stream) : launch_tile_pk<6, false>(t, l, stream);
        case 7: return t.range_cls ? launch_tile_pk<7, true>(t, l, stream) : launch_tile_pk<7, false>(t, l, stream);
        case 8: return t.range_cls ? launch_tile_pk<8, true>(t, l, stream) : launch_tile_pk<8, false>(t, l, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (t.filt_k) {
    case 1: return launch_tile_k<1>(t, l, stream);
    case 2: return launch_tile_k<2>(t, l, stream);
    case 3: return launch_tile_k<3>(t, l, stream);
    case 4: return launch_tile_k<4>(t, l, stream);
    case 5: return launch_tile_k<5>(t, l, stream);
    case 6: return launch_tile_k<6>(t, l, stream);
    case 7: return launch_tile_k<7>(t, l, stream);
    case 8: return launch_tile_k<8>(t, l, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_ac_filter(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name) {
    static thread_local char name[64];
    if (!tile_split_supported(t) || t.filt_k > 8) return hipErrorInvalidValue;
    std::snprintf(name, sizeof(name), "k_ac_tile<%u, %s, %s, true>", t.filt_k, t.range_cls ? "true" : "false",
                  t.filt_row_bytes == 8 ? "true" : "false");
    if (kernel_name) *kernel_name = name;
    if (t.hashk) {
        std::snprintf(name, sizeof(name), "k_ac_tile<%u, false, true, true, true>", t.filt_k);
        switch (t.filt_k) {
        case 1: return launch_tile_hashk<1, true>(t, l, stream);
        case 2: return launch_tile_hashk<2, true>(t, l, stream);
        case 3: return launch_tile_hashk<3, true>(t, l, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (t.filt_k) {
    case 1: return launch_filter_k<1>(t, l, stream);
    case 2: return launch_filter_k<2>(t, l, stream);
    case 3: return launch_filter_k<3>(t, l, stream);
    case 4: return launch_filter_k<4>(t, l, stream);
    case 5: return launch_filter_k<5>(t, l, stream);
    case 6: return launch_filter_k<6>(t, l, stream);
    case 7: return launch_filter_k<7>(t, l, stream);
    case 8: return launch_filter_k<8>(t, l, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_ac_verify(const DevTables &t, const TileLaunch &l, hipStream_t stream) {
    if (t.hashk) {
        switch (t.filt_k) {
        case 1: hipLaunchKernelGGL((k_ac_verify<1, false, true>), dim3(l.verify_grid), dim3(256), 0, stream, t, l); break;
        case 2: hipLaunchKernelGGL((k_ac_verify<2, false, true>), dim3(l.verify_grid), dim3(256), 0, stream, t, l); break;
        case 3: hipLaunchKernelGGL((k_ac_verify<3, false, true>), dim3(l.verify_grid), dim3(256), 0, stream, t, l); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (t.filt_k) {
    case 1: return launch_verify_k<1>(t, l, stream);
    case 2: return launch_verify_k<2>(t, l, stream);
    case 3: return launch_verify_k<3>(t, l, stream);
    case 4: return launch_verify_k<4>(t, l, stream);
    case 5: return launch_verify_k<5>(t, l, stream);
    case 6: return launch_verify_k<6>(t, l, stream);
    case 7: return launch_verify_k<7>(t, l, stream);
    case 8: return launch_verify_k<8>(t, l, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace acgpu
