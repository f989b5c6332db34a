// acgpu_wholeword.hip -- WholeWordMatchSet/Map on gfx950.
//
// The reference (S/WholeWordMatchMap.java:155-240) tokenises the haystack into maximal runs of word characters and
// reports a run iff its (case-folded) text is a keyword.  With a fold-consistent word-character table
// (wordchar[c] == wordchar[lower[c]] for every c: every case-sensitive use and the default table) that is a
// position-parallel problem:
//
//   filter : a position starts a run iff it is a word character and its left neighbour is not.  The 65536-bit
//            word-character table lives in LDS (8 KB); one ds_read_b32 per unit.
//   verify : run starts are compacted in text order into the per-wave LDS queue; kWwBatches*64 at a time each
//            lane reads its run 8 units per load, folds it (paged delta table in LDS), hashes it (h*33 + packed units, murmur finaliser) up to the
//            first non-word unit, looks the hash up in the table of whole keywords and compares the folded run with
//            the keyword record unit for unit -- three dependent memory accesses per word instead of a trie edge
//            per unit.  At most one record per run start, ranks by wave prefix sums (as in acgpu_tile.hip).
//            (debug bit 256 selects the older verification that walks the keyword trie through its hashed edges.)
//
// Inconsistent tables (possible only with a custom table in case-insensitive mode, where the reference mixes folded
// and raw lookups, S/WholeWordMatchMap.java:204,209 vs :221,:226) take k_ww_sequential: a literal single-lane
// restatement of the reference loop -- slow, but exact.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "acgpu_tile_common.h"

namespace acgpu {

__device__ __forceinline__ uint32_t word_bit(const uint32_t *wbits, uint32_t unit) {
    return __builtin_amdgcn_ubfe(wbits[unit >> 5], unit, 1);
}

// word-character bit of unit j (0..7) of a packed 8-unit window, or-ed into bit j of wm.  Spelled out because it runs for
// every unit of the haystack and of every run: word address = (unit >> 5) * 4 straight from the packed register (two
// instructions), the bit offset is the packed register itself for the low unit (v_bfe_u32 looks at five bits of it), and
// the bit lands in the mask by one v_lshl_or_b32.
template <int J>
__device__ __forceinline__ uint32_t word_bit_into(const uint32_t *wbits, uint32_t packed, uint32_t wm) {
    const uint32_t off = ((J & 1) ? packed >> 19 : packed >> 3) & 0x1ffcu;
    const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(wbits) + off);
    uint32_t bit, r;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(bit) : "v"(word), "v"((J & 1) ? packed >> 16 : packed));
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(bit), "n"(J), "v"(wm));
    return r;
}

// (the first N units only, N a compile-time constant: no branches between the lookups, all of them in flight together)
template <int N, int J = 0>
__device__ __forceinline__ uint32_t word_bits8(const uint32_t *wbits, const uint32_t (&d)[4], uint32_t wm = 0) {
    if constexpr (J < N) return word_bits8<N, J + 1>(wbits, d, word_bit_into<J>(wbits, d[J >> 1], wm));
    else return wm;
}

#ifndef ACGPU_WW_NB
#define ACGPU_WW_NB 3
#endif
constexpr int kWwBatches = ACGPU_WW_NB;   // run starts verified per lane and call (independent lookup chains in flight)
#ifndef ACGPU_WW_PREFETCH
#define ACGPU_WW_PREFETCH 2
#endif
#ifndef ACGPU_WW_BLOCKS
#define ACGPU_WW_BLOCKS 1
#endif
constexpr int kWwPrefetch = ACGPU_WW_PREFETCH; // tiles per load group (the verification needs the registers)
constexpr int kWwBlocksPerCu = ACGPU_WW_BLOCKS; // resident blocks per CU the register budget is set for
// run starts per tile <= 256 (a start needs a non-word unit before it); the queue holds one verification call's worth
// (kept until the next call) plus one tile
constexpr int kWwCandCap = kWwBatches * 64 + 256 + 64;

int ww_blocks_per_cu() { return kWwBlocksPerCu; }
#ifndef ACGPU_FOLD_PAGES_MAX
#define ACGPU_FOLD_PAGES_MAX 64
#endif
constexpr uint32_t kFoldPagesMax = ACGPU_FOLD_PAGES_MAX; // 32 KB of LDS; Unicode 13 simple lower-casing needs 18 pages
constexpr uint32_t kBytePagesMax = 64;                    // k_ww_pp, FOLD 3: 16 KB (acgpu_build.cpp caps HostTables::ww_bp_n at this)

uint32_t ww_fold_pages_in_lds(const DevTables &t) { return (!t.cs && t.fold_n_pages <= kFoldPagesMax) ? t.fold_n_pages : 0u; }

static size_t ww_bloom_bytes(const DevTables &t) { return ((size_t)t.ww_bloom_mask + 1) / 8; }

// dynamic LDS of k_ww_tile: [Bloom words | candidate queues]; the word-character bits (8 KB), the page index (256 B) and
// the pages of the fold table (32 KB) are STATIC LDS, at addresses the compiler knows: a lookup is then address arithmetic on the unit alone
// (with everything behind a dynamic base the compiler added the base -- a literal 0 -- to every one of them)
size_t ww_lds_bytes(int block_threads, const DevTables &t) {
    const uint32_t fold_pages = ww_fold_pages_in_lds(t);
    (void)fold_pages; // (the pages are static LDS as well: kFoldPagesMax of them)
    return ww_bloom_bytes(t) + (size_t)(block_threads / kWave) * kWwCandCap * sizeof(uint32_t);
}

struct __attribute__((packed, aligned(2))) WwUnits8 { // 8 UTF-16 units at any unit address (one global_load_dwordx4)
    uint32_t d[4];
};

struct FoldLds {
    const uint32_t *bloom;  // Bloom filter over the keyword hashes (not part of folding; travels with the LDS tables)
    uint32_t bloom_mask;
    const uint8_t *pgidx;   // 256 page numbers
    const uint16_t *pages;  // pages of 256 deltas (page 0: all zero)
};

// FOLD: 0 = case sensitive, 1 = paged delta table in LDS, 2 = the 65536-entry table in global memory
template <int FOLD>
__device__ __forceinline__ uint32_t ww_fold(const DevTables &T, const FoldLds &F, uint32_t u) {
    if (FOLD == 0) return u;
    if (FOLD == 1) {
        // branch free: page 0 (nothing on the page folds) is stored as 256 zero deltas, so every unit takes the same two
        // dependent LDS reads and the eight units of a chunk have theirs in flight together.  (A direct table for the low,
        // cased pages saved one read for those units but made the fold a divergent branch with a wait per unit:
        // 20 instructions and up to two LDS round trips per unit, one unit after the other.)
        const uint32_t pg = F.pgidx[u >> 8];
        return (u + F.pages[pg * 256u + (u & 255u)]) & 0xffffu;
    }
    return T.lower[u];
}

// 8 units starting at unit index p (p < n): one unaligned 16-byte load, or unit by unit at the end of the buffer
__device__ __forceinline__ WwUnits8 ww_window(const uint16_t *hay, uint32_t p, uint32_t n) {
    if (p + 8 <= n) return *reinterpret_cast<const WwUnits8 *>(hay + p);
    WwUnits8 w{{0, 0, 0, 0}};
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (p + j < n) w.d[j >> 1] |= (uint32_t)hay[p + j] << (16 * (j & 1));
    return w;
}

// One 8-unit chunk of a run: word bits, run length inside the chunk (0..8; `valid` = units that exist in the buffer),
// folded units packed two per word and zeroed beyond the run.
// `look` (wave-uniform, 1..8): only the first `look` units count -- a run that is still going after them is longer
// than every keyword anyway (the second chunk of a run needs max_len + 1 - 8 units, not 8).  N >= look units are looked up,
// N a compile-time constant: with a run-time bound every unit sat behind its own scalar branch, and its two dependent LDS
// reads were waited for one unit after the other (instrumented build: the 5-unit second chunk took 3.4x the 8-unit first).
template <int FOLD, int N = 8>
__device__ __forceinline__ uint32_t ww_chunk(const DevTables &T, const FoldLds &F, const uint32_t *wbits, const WwUnits8 &w,
                                             uint32_t valid, uint32_t out[4], uint32_t look = 8) {
    uint32_t wm = word_bits8<N>(wbits, w.d), f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f[j] = 0;
        if (j >= N) continue;
        const uint32_t u = (w.d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        f[j] = ww_fold<FOLD>(T, F, u);
    }
    wm &= (1u << min(valid, look)) - 1u;
    const uint32_t rl = (uint32_t)__builtin_ctz(~wm | 0x100u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t d = f[2 * i] | (f[2 * i + 1] << 16);
        const int m = (int)rl - 2 * i;
        out[i] = m >= 2 ? d : (m == 1 ? (d & 0xffffu) : 0u);
    }
    return rl;
}

// Verification of up to kWwBatches*64 run starts by hashing the whole run.
template <int FOLD>
__device__ __forceinline__ void ww_verify_hash(TileCtx &c, const uint32_t *wbits, const FoldLds F, uint32_t head, uint32_t n_cand) {
    constexpr int NB = kWwBatches;
    const DevTables &T = *c.Tp;
    const TileLaunch &L = *c.Lp;
    const uint16_t *hay = L.d_hay;
    const uint32_t lane = lane_id();
    const uint32_t n = L.n_units;
    uint32_t s[NB], r[NB], h[NB], fw[NB][8];
    bool act[NB], run[NB]; // run: every unit so far was a word character
#ifdef ACGPU_TIMING
    unsigned long long vt0 = clock64();
    c.vt[6] += 1;
#define WT_MARK(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); c.vt[i] += t_ - vt0; vt0 = t_; }
#else
#define WT_MARK(i)
#endif
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t q = b * kWave + lane;
        act[b] = q < n_cand;
        s[b] = act[b] ? c.cand[head + q] : 0u;
        r[b] = 0;
        run[b] = act[b];
#pragma unroll
        for (int k = 0; k < 8; ++k) fw[b][k] = 0;
    }
    // units 0..15 of every run: the folded units are kept (packed) for the exact comparison.  Both 16-byte windows of every
    // run start are requested up front: one memory round trip per call instead of one per chunk (a third of the runs go on
    // into the second window, so the wave would wait for it anyway).
    WwUnits8 win2[2][NB];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            win2[k][b] = WwUnits8{{0, 0, 0, 0}};
            if (act[b] && s[b] + 8 * k < n) win2[k][b] = ww_window(hay, s[b] + 8 * k, n);
        }
    WT_MARK(0)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        WwUnits8 win[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            win[b] = win2[k][b];
            if (!run[b]) win[b] = WwUnits8{{0, 0, 0, 0}};
        }
        // units 8..: a run of more than max_len units matches nothing, so only max_len + 1 - 8 of them matter
        const uint32_t look = k == 0 ? 8u : min(8u, T.max_len > 7u ? T.max_len - 7u : 1u);
        // (no branch inside this loop: the batches' LDS lookups interleave, and a branch per batch -- a shorter variant of the
        // second chunk was one -- puts every batch's two dependent LDS round trips end to end)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t valid = run[b] ? min(n - min(s[b] + 8 * k, n), 8u) : 0u;
            const uint32_t rl = ww_chunk<FOLD, 8>(T, F, wbits, win[b], valid, &fw[b][4 * k], look);
            r[b] += rl;
            run[b] = rl == 8;
        }
        bool any_run = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) any_run |= run[b];
        if (k == 0) { WT_MARK(1) } else { WT_MARK(2) }
        if (!__any(any_run)) break;
    }
    uint32_t g[NB]; // the second hash (names the keyword's other slot in the table)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        h[b] = g[b] = T.ww_seed;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[b] = ww_hash_step(h[b], fw[b][i]);
            g[b] = ww_hash2_step(g[b], fw[b][i]);
        }
    }
    // longer runs: hash only (the comparison re-reads the text beyond unit 16)
    for (uint32_t k = 2;; ++k) {
        bool any_run = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (run[b] && 8 * k > T.max_len) { // longer than every keyword
                run[b] = false;
                act[b] = false;
            }
            any_run |= run[b];
        }
        if (!__any(any_run)) break;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (!run[b]) continue;
            const uint32_t p = s[b] + 8 * k;
            if (p >= n) {
                run[b] = false;
                continue;
            }
            uint32_t out[4];
            const uint32_t rl = ww_chunk<FOLD>(T, F, wbits, ww_window(hay, p, n), min(n - p, 8u), out);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (rl > 2u * i) {
                    h[b] = ww_hash_step(h[b], out[i]);
                    g[b] = ww_hash2_step(g[b], out[i]);
                }
            r[b] += rl;
            run[b] = rl == 8;
        }
    }
    // table lookup: a keyword sits in one of TWO 32-byte slots -- {tag, id, its first 12 folded units} -- that the hash names;
    // both are gathered at once, so a word of up to 12 units is decided by one round of memory accesses, the same for
    // every lane (the tag carries the length); longer keywords also compare their record, whose offset takes the id's place
    const uint4 *fat = reinterpret_cast<const uint4 *>(T.ww_fat);
    const uint4 *recs = reinterpret_cast<const uint4 *>(T.ww_recs);
    uint32_t id[NB];
    bool probing[NB];
    uint4 ea0[NB], ea1[NB], eb0[NB], eb1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        id[b] = ~0u;
        h[b] = ww_hash_final(h[b]);
        probing[b] = act[b] && r[b] != 0 && r[b] <= T.max_len && !ACGPU_DBG(L, 2u); // 2: ablation, no table lookup
        if (!ACGPU_DBG(L, 4u)) { // 4: ablation, no Bloom filter in front of the table
            const uint32_t b1 = ww_bloom_bit1(h[b], F.bloom_mask), b2 = ww_bloom_bit2(h[b], F.bloom_mask);
            probing[b] = probing[b] && ((F.bloom[b1 >> 5] >> (b1 & 31)) & (F.bloom[b2 >> 5] >> (b2 & 31)) & 1u);
        }
        ea0[b] = ea1[b] = eb0[b] = eb1[b] = make_uint4(0u, 0u, 0u, 0u);
        if (probing[b]) {
            const uint32_t s1 = ww_slot1(h[b], T.ww_fat_mask), s2 = ww_slot2(h[b], g[b], T.ww_fat_mask);
            ea0[b] = fat[2 * s1];
            ea1[b] = fat[2 * s1 + 1];
            eb0[b] = fat[2 * s2];
            eb1[b] = fat[2 * s2 + 1];
        }
    }
    WT_MARK(3)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t tag = ww_tag(h[b], r[b]);
        const bool in_a = ea0[b].x == tag && ea0[b].z == fw[b][0] && ea0[b].w == fw[b][1] && ea1[b].x == fw[b][2] &&
                          ea1[b].y == fw[b][3] && ea1[b].z == fw[b][4] && ea1[b].w == fw[b][5];
        const bool in_b = eb0[b].x == tag && eb0[b].z == fw[b][0] && eb0[b].w == fw[b][1] && eb1[b].x == fw[b][2] &&
                          eb1[b].y == fw[b][3] && eb1[b].z == fw[b][4] && eb1[b].w == fw[b][5];
        if (probing[b] && (in_a || in_b)) id[b] = in_a ? ea0[b].y : eb0[b].y;
    }
    {
        // keywords of more than 12 units (rare): tag and first 12 units agree, the rest from the record.  Two keywords
        // can share tag and 12 units; then both slots say yes and both records are looked at.
        bool any_long = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) any_long |= id[b] != ~0u && r[b] > kWwInlineUnits;
        if (__any(any_long)) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (id[b] == ~0u || r[b] <= kWwInlineUnits) continue;
                const uint32_t tag = ww_tag(h[b], r[b]);
                uint32_t found = ~0u;
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const uint4 &e0 = which ? eb0[b] : ea0[b], &e1 = which ? eb1[b] : ea1[b];
                    const bool head = e0.x == tag && e0.z == fw[b][0] && e0.w == fw[b][1] && e1.x == fw[b][2] &&
                                      e1.y == fw[b][3] && e1.z == fw[b][4] && e1.w == fw[b][5];
                    if (!head || found != ~0u) continue;
                    const uint4 *rec = recs + e0.y;
                    const uint4 a = rec[0], q = rec[2];
                    bool same = a.y == r[b] && q.x == fw[b][6] && q.y == fw[b][7];
                    if (same && r[b] > 16) {
                        const uint16_t *ru = reinterpret_cast<const uint16_t *>(rec) + 4; // the record's units
                        for (uint32_t i = 16; i < r[b] && same; ++i) same = ww_fold<FOLD>(T, F, hay[s[b] + i]) == ru[i];
                    }
                    if (same) found = a.x;
                }
                id[b] = found;
            }
        }
    }
    WT_MARK(4)
    uint32_t m[NB], prefix[NB], total = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        m[b] = id[b] != ~0u ? 1u : 0u;
        const uint32_t incl = wave_inclusive_scan_dpp(m[b]);
        prefix[b] = total + incl - m[b];
        total += __builtin_amdgcn_readlane(incl, kWave - 1);
    }
    if (total == 0) return;
    if (L.d_region_recs) { // region-local records: the slot is the rank (see TileLaunch::d_region_recs)
        int32_t *base = L.d_region_recs + ((size_t)c.region * L.region_cap + c.rank_base) * 3;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (m[b]) {
                typedef int32_t v3i __attribute__((ext_vector_type(3)));
                const v3i rec = {(int32_t)s[b], (int32_t)(s[b] + r[b]), (int32_t)id[b]};
                __builtin_nontemporal_store(rec, reinterpret_cast<v3i *>(base + (size_t)prefix[b] * 3));
            }
    } else {
        const SlotRange sr = reserve_slots(c, total);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (m[b]) store_rec(L, sr.slot(prefix[b]), s[b], s[b] + r[b], id[b], c.rank_base + prefix[b]);
    }
    c.rank_base += total;
    WT_MARK(5)
}

// Verification of up to kWwBatches*64 run starts, kWwBatches per lane, advanced in lock step.
__device__ __forceinline__ void ww_verify(TileCtx &c, const uint32_t *wbits, uint32_t head, uint32_t n_cand) {
    constexpr int NB = kWwBatches;
    const DevTables &T = *c.Tp;
    const TileLaunch &L = *c.Lp;
    const uint16_t *hay = L.d_hay;
    const uint32_t lane = lane_id();
    uint32_t s[NB], i[NB], node[NB];
    bool go[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t q = b * kWave + lane;
        go[b] = q < n_cand;
        s[b] = go[b] ? c.cand[head + q] : 0u;
        i[b] = s[b];
        node[b] = go[b] ? 0u : ~0u; // ~0u: no match
    }
    for (;;) { // one unit of every live run per round
        bool any_go = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) any_go |= go[b];
        if (!__any(any_go)) break;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (go[b]) {
                if (i[b] >= L.n_units) {
                    go[b] = false; // the run ends with the buffer
                } else {
                    const uint32_t u = hay[i[b]];
                    if (!word_bit(wbits, u)) {
                        go[b] = false; // the run ends here: node[b] decides
                    } else {
                        const uint32_t f = T.cs ? u : (uint32_t)T.lower[u];
                        const uint32_t child = hashed_goto(T.hkeys, T.hvals, T.hmask, node[b], f);
                        if (child == ~0u) { // a word character with no continuation: the run is not a keyword
                            node[b] = ~0u;
                            go[b] = false;
                        } else {
                            node[b] = child;
                            ++i[b];
                        }
                    }
                }
            }
        }
    }
    uint32_t id[NB], m[NB], prefix[NB], total = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        id[b] = node[b] != ~0u && node[b] != 0u ? T.term_id[node[b]] : ~0u;
        m[b] = id[b] != ~0u ? 1u : 0u;
        const uint32_t incl = wave_inclusive_scan_dpp(m[b]);
        prefix[b] = total + incl - m[b];
        total += __builtin_amdgcn_readlane(incl, kWave - 1);
    }
    if (total == 0) return;
    if (L.d_region_recs) {
        int32_t *base = L.d_region_recs + ((size_t)c.region * L.region_cap + c.rank_base) * 3;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (m[b]) {
                int32_t *o = base + (size_t)prefix[b] * 3;
                o[0] = (int32_t)s[b]; o[1] = (int32_t)i[b]; o[2] = (int32_t)id[b];
            }
    } else {
        const SlotRange sr = reserve_slots(c, total);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (m[b]) store_rec(L, sr.slot(prefix[b]), s[b], i[b], id[b], c.rank_base + prefix[b]);
    }
    c.rank_base += total;
}

template <int FOLD>
__device__ __forceinline__ void ww_drain(TileCtx &c, const uint32_t *wbits, const FoldLds F, uint32_t keep_below) {
    uint32_t head = 0;
    while (c.cand_n > head && c.cand_n - head >= keep_below) {
        const uint32_t nb = min(c.cand_n - head, (uint32_t)(kWwBatches * kWave));
        if (c.Lp->debug & 256u) ww_verify(c, wbits, head, nb);
        else if (!ACGPU_DBG(*c.Lp, 1u)) ww_verify_hash<FOLD>(c, wbits, F, head, nb); // 1: ablation, run starts are dropped
        head += nb;
    }
    if (head) {
        const uint32_t left = c.cand_n - head;
        uint32_t tmp[kWwBatches];
#pragma unroll
        for (int b = 0; b < kWwBatches; ++b) {
            const uint32_t q = b * kWave + lane_id();
            tmp[b] = q < left ? c.cand[head + q] : 0u;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < kWwBatches; ++b) {
            const uint32_t q = b * kWave + lane_id();
            if (q < left) c.cand[q] = tmp[b];
        }
        __builtin_amdgcn_wave_barrier();
        c.cand_n = left;
    }
}

// Same span/region/tile-group structure as k_ac_tile (acgpu_tile.hip); only the filter and the verification differ.
template <int FOLD>
__global__ __launch_bounds__(kTileBlock, kWwBlocksPerCu * (kTileBlock / 256)) void k_ww_tile(DevTables T, TileLaunch L) {
    __shared__ __attribute__((aligned(16))) uint32_t wbits[2048];       // 65536 word-character bits
    __shared__ __attribute__((aligned(16))) unsigned char fold_base[FOLD == 1 ? 256 : 16];        // page index of the fold table
    __shared__ __attribute__((aligned(16))) uint16_t pages[FOLD == 1 ? kFoldPagesMax * 256 : 8]; // its pages of 256 deltas
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t bloom_bytes = (T.ww_bloom_mask + 1u) / 8u;
    uint32_t *bloom = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cand_all = reinterpret_cast<uint32_t *>(smem + bloom_bytes);
    // the LDS tables, 16 bytes per load (every workgroup copies them while nothing else runs: the packed word bits come
    // ready-made from the builder -- packing them here took 64 byte loads per thread)
    for (uint32_t w = threadIdx.x; w < bloom_bytes / 16; w += blockDim.x)
        reinterpret_cast<uint4 *>(bloom)[w] = reinterpret_cast<const uint4 *>(T.ww_bloom)[w];
    for (uint32_t w = threadIdx.x; w < 2048 / 4; w += blockDim.x)
        reinterpret_cast<uint4 *>(wbits)[w] = reinterpret_cast<const uint4 *>(T.wbits)[w];
    FoldLds F{bloom, T.ww_bloom_mask, fold_base, pages};
    if (FOLD == 1) {
        for (uint32_t i = threadIdx.x; i < 256 / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(fold_base)[i] = reinterpret_cast<const uint4 *>(T.fold_pgidx)[i];
        for (uint32_t i = threadIdx.x; i < T.fold_n_pages * 32u; i += blockDim.x) // (a page = 512 bytes)
            reinterpret_cast<uint4 *>(pages)[i] = reinterpret_cast<const uint4 *>(T.fold_pages)[i];
    }
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave); // (scalar bookkeeping: see k_ac_tile)
    const uint32_t wave_global = blockIdx.x * (kTileBlock / kWave) + wave_in_block;
    TileCtx c{&T, &L, cand_all + wave_in_block * kWwCandCap, 0, 0, 0u, 0};
    c.wg = blockIdx.x;

    const uint32_t first_region = wave_global * L.regions_per_wave;
    if (first_region >= L.n_regions) return;
#ifdef ACGPU_TIMING
    const unsigned long long tm_start = clock64();
#endif
    const uint32_t last_region = min(first_region + L.regions_per_wave, L.n_regions);
    const uint32_t base8 = L.own_begin & ~7u;
    const uint32_t R = L.region_units;
    const uint32_t span_begin = max(L.own_begin, base8 + first_region * R);
    uint32_t span_end = base8 + last_region * R;
    if (span_end > L.own_end || last_region == L.n_regions) span_end = L.own_end;
    const uint32_t nfull = L.n_units & ~7u;
    const uint32_t hi = min(span_end, nfull);
    const uint32_t last_vec = nfull >= 8 ? nfull - 8 : 0;
    const uint16_t *hay = L.d_hay;

    uint32_t region = first_region;
    c.region = region;
    uint32_t boundary = base8 + (region + 1) * R;
    uint32_t rb = span_begin;
    uint32_t re = min(span_end, boundary);
    uint32_t tile = base8 + first_region * R;

    bool vec_todo = tile < hi;
    bool tail_todo = span_end > nfull;
    uint32_t d0 = 0;
    uint32_t carry = 0; // word-character bit of the unit just before the current tile
    uint4 nxt[kWwPrefetch], grp[kWwPrefetch];
#pragma unroll
    for (int d = 0; d < kWwPrefetch; ++d) nxt[d] = grp[d] = make_uint4(0, 0, 0, 0);
    if (vec_todo) {
        if (tile >= 1) carry = word_bit(wbits, hay[tile - 1]);
#pragma unroll
        for (int d = 0; d < kWwPrefetch; ++d)
            nxt[d] = *reinterpret_cast<const uint4 *>(hay + min(tile + d * kTileUnits + lane * 8, last_vec));
    }

    uint32_t prio_turn = 0;
    for (;;) {
#ifndef ACGPU_NO_SETPRIO
        // the waves of a SIMD take turns at every issue priority (see k_ac_tile: the arbiter's oldest-first order lets the
        // youngest waves finish last)
        switch ((wave_in_block / 4u + prio_turn++) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        const bool seam = vec_todo ? (d0 == 0 && tile >= boundary) : true;
        const uint32_t keep = seam ? 1u : (uint32_t)(kWwBatches * kWave);
        if (c.cand_n >= keep && c.cand_n != 0) ww_drain<FOLD>(c, wbits, F, keep);

        if (vec_todo) {
            if (d0 == 0) {
                if (tile >= boundary) {
                    if (lane == 0) L.d_region_counts[region] = c.rank_base;
                    c.rank_base = 0;
                    ++region;
                    c.region = region;
                    rb = boundary;
                    boundary += R;
                    re = min(span_end, boundary);
                }
#pragma unroll
                for (int d = 0; d < kWwPrefetch; ++d) grp[d] = nxt[d];
#pragma unroll
                for (int d = 0; d < kWwPrefetch; ++d)
                    nxt[d] = *reinterpret_cast<const uint4 *>(
                        hay + min(tile + (kWwPrefetch + d) * kTileUnits + lane * 8, last_vec));
            }
            const uint32_t top = min(re, hi);
            const bool edge = tile < rb || tile + kWwPrefetch * kTileUnits > top;
            bool resume = false;
#pragma unroll
            for (int d = 0; d < kWwPrefetch; ++d) {
                if ((uint32_t)d < d0) continue;
                const uint32_t cur = tile + d * kTileUnits;
                if (cur >= hi) break;
                if (c.cand_n > kWwCandCap - 256) {
                    d0 = d;
                    resume = true;
                    break;
                }
                const uint4 w = grp[d];
                const uint32_t v = cur + lane * 8;
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
                const uint32_t wm = word_bits8<8>(wbits, ww); // word-character bits of the lane's 8 units
                const uint32_t prev = from_prev_lane(wm >> 7, carry); // bit of the unit left of v
                carry = __builtin_amdgcn_readlane(wm, 63) >> 7;
                uint32_t mask = wm & ~((wm << 1) | (prev & 1u)) & 0xffu; // run starts
                if (edge) {
                    const uint32_t first = rb > v ? min(rb - v, 8u) : 0u;
                    const uint32_t last = top > v ? min(top - v, 8u) : 0u;
                    mask &= ((1u << last) - 1u) & ~((1u << first) - 1u);
                }
                enqueue(c, mask, v);
            }
            if (!resume) {
                d0 = 0;
                tile += kWwPrefetch * kTileUnits;
                vec_todo = tile < hi;
            }
            continue;
        }
        if (tail_todo) {
            tail_todo = false;
            const uint32_t t0 = max(nfull, span_begin);
            if (t0 >= boundary) {
                if (lane == 0) L.d_region_counts[region] = c.rank_base;
                c.rank_base = 0;
                ++region;
                c.region = region;
            }
            const uint32_t pos = t0 + lane;
            uint32_t mask = 0;
            if (pos < span_end) {
                const uint32_t here = word_bit(wbits, hay[pos]);
                const uint32_t left = pos > 0 ? word_bit(wbits, hay[pos - 1]) : 0u;
                mask = here & ~left & 1u;
            }
            enqueue(c, mask, pos);
            continue;
        }
        break;
    }
    if (lane == 0) L.d_region_counts[region] = c.rank_base;
#ifdef ACGPU_TIMING
    if (lane == 0 && L.d_timing) {
        L.d_timing[(size_t)wave_global * 8 + 0] = clock64() - tm_start;
        for (int i = 0; i < 7; ++i) L.d_timing[(size_t)wave_global * 8 + 1 + i] = c.vt[i];
    }
#endif
    for (uint32_t i = lane; i < c.res_left; i += kWave)
        if (c.res_cur + i < c.slot_limit) store_rec(L, c.res_cur + i, 0, 0, 0, ~0u); // (not into the next slice)
}

// ---- k_ww_pp: the position-parallel form (keywords of at most 16 units, fold table in LDS or none) ----------------------
// k_ww_tile verifies a run start by reading the run back from memory: two text windows, then word bits, fold and hash of 16
// units per run, then the table round trip -- a chain of dependent stages, 70 % of a wave's time.  Here every unit is folded
// ONCE, at stream time, by the lane that holds it: the folded units of the current and the next tile and their
// word-character bits sit in a small per-wave LDS ring, a run start reads its 16 folded units and its run length from there
// (two LDS reads, no text window), and the table probes of one batch of run starts are in flight while the next tile is
// folded and hashed -- they are compared, in text order, just before the next batch's probes go out.
//   per wave: ring   = 2 tile slots of 512 folded units + a copy of slot 0's first 32 units behind slot 1 (a run of the
//                      tile in slot 1 goes on in slot 0)
//             bits   = the same for the word-character bits, one byte per lane and tile
//             list   = the run starts of the tile being verified, tile relative, in text order
constexpr int kPpRingUnits = 2 * kTileUnits + 32;
constexpr int kPpBitBytes = 2 * (kTileUnits / 8) + 8;
constexpr int kPpListCap = kTileUnits / 2; // a run start needs a unit that is no word character before it
constexpr int kPpWaveBytes = (kPpRingUnits * 2 + kPpBitBytes + kPpListCap * 2 + 15) & ~15;
constexpr uint32_t kPpMaxLen = 32; // longer keywords: k_ww_tile (up to 16 units: the LONG = false form, one 32-byte ring read per run)

// the perfect hash's displacements take the Bloom filter's place in LDS (tile_debug bit 2^29: the two-choice table behind the filter, A/B)
static bool ww_pp_perfect(const DevTables &t, const TileLaunch &l) { return t.ww_ph != nullptr && !(l.debug & (1u << 29)); }
static size_t ww_pp_front_bytes(const DevTables &t, bool ph) { return ph ? ((size_t)t.ww_ph_buckets + 7) / 8 * 16 : ww_bloom_bytes(t); }
static size_t ww_pp_lds_bytes(int block_threads, const DevTables &t, bool ph) {
    return ww_pp_front_bytes(t, ph) + (size_t)(block_threads / kWave) * kPpWaveBytes;
}

// One batch of run starts, hashed and ready to probe: everything but the probed slots (registers; PpBatch travels from the
// pass that computes it to the next pass, which issues its probes, and compares them at its end -- the LOADED registers
// never cross the loop's back edge, where a register copy would wait for the loads right behind their issue)
template <int NW> // folded units kept per run, two per word: 8 (keywords of up to 16 units) or 16 (up to 32)
struct PpFlight {
    uint32_t n = 0; // wave-uniform: entries of the batch, 0 = nothing to do
    bool probing = false;
    uint32_t s = 0, r = 0, tag = 0, s1 = 0, s2 = 0, fw[NW] = {};
};
struct PpProbe {
    uint4 ea0, ea1, eb0, eb1;
};

// the four probe loads of a batch, unconditional (a lane that does not probe reads slot 0: one cached line): the number of
// memory operations of a pass is then the same on every path, and the compiler's waits can count
// PH: the perfect hash -- ONE slot per run (HostTables::ww_ph)
template <int NW, bool PH>
__device__ __forceinline__ PpProbe pp_issue(const DevTables &T, const PpFlight<NW> &fl) {
    const uint4 *fat = reinterpret_cast<const uint4 *>(PH ? T.ww_ph : T.ww_fat);
    PpProbe pr;
    pr.ea0 = fat[2 * fl.s1];
    pr.ea1 = fat[2 * fl.s1 + 1];
    if (!PH) {
        pr.eb0 = fat[2 * fl.s2];
        pr.eb1 = fat[2 * fl.s2 + 1];
    } else {
        pr.eb0 = pr.eb1 = make_uint4(0u, 0u, 0u, 0u);
    }
    return pr;
}

// units 16.. of a run against its keyword's record (record words: payload, length, then the folded units two per word; rec4[2]
// = words 6..9 of them is in q).  Only words the keyword has are looked at: the groups behind a record belong to the next one.
template <int NW>
__device__ __forceinline__ bool pp_tail_same(const PpFlight<NW> &fl, const uint4 *rec, const uint4 &q) {
    if (NW <= 8 || fl.r <= 16) return true;
    bool same = q.z == fl.fw[8] && (fl.r <= 18 || q.w == fl.fw[9]);
    if (fl.r > 20) {
        const uint4 t = rec[3];
        same = same && t.x == fl.fw[10] && (fl.r <= 22 || t.y == fl.fw[11]) && (fl.r <= 24 || t.z == fl.fw[12]) && (fl.r <= 26 || t.w == fl.fw[13]);
    }
    if (fl.r > 28) {
        const uint4 t = rec[4];
        same = same && t.x == fl.fw[14] && (fl.r <= 30 || t.y == fl.fw[15]);
    }
    return same;
}

// compare the probed slots with the runs of the batch and emit its records (text order: lane order)
template <int NW, bool PH>
__device__ __forceinline__ void pp_consume(TileCtx &c, const PpFlight<NW> &fl, const PpProbe &pr) {
    if (fl.n == 0) return; // wave-uniform
    const DevTables &T = *c.Tp;
    const TileLaunch &L = *c.Lp;
    const uint4 a0 = pr.ea0, a1 = pr.ea1, b0 = pr.eb0, b1 = pr.eb1;
    const uint32_t ida = a0.y, idb = b0.y;
    const bool in_a = a0.x == fl.tag && a0.z == fl.fw[0] && a0.w == fl.fw[1] && a1.x == fl.fw[2] && a1.y == fl.fw[3] &&
                      a1.z == fl.fw[4] && a1.w == fl.fw[5];
    const bool in_b = !PH && b0.x == fl.tag && b0.z == fl.fw[0] && b0.w == fl.fw[1] && b1.x == fl.fw[2] && b1.y == fl.fw[3] &&
                      b1.z == fl.fw[4] && b1.w == fl.fw[5];
    uint32_t id = ~0u;
    if (fl.probing && (in_a || in_b)) id = in_a ? ida : idb;
    if (__any(id != ~0u && fl.r > kWwInlineUnits)) { // keywords of more than 12 units: the rest from the record (see k_ww_tile)
        if (id != ~0u && fl.r > kWwInlineUnits) {
            const uint4 *recs = reinterpret_cast<const uint4 *>(T.ww_recs);
            uint32_t found = ~0u;
            if (in_a) {
                const uint4 *rec = recs + ida;
                const uint4 a = rec[0], q = rec[2];
                if (a.y == fl.r && q.x == fl.fw[6] && q.y == fl.fw[7] && pp_tail_same<NW>(fl, rec, q)) found = a.x;
            }
            if (in_b && found == ~0u) {
                const uint4 *rec = recs + idb;
                const uint4 a = rec[0], q = rec[2];
                if (a.y == fl.r && q.x == fl.fw[6] && q.y == fl.fw[7] && pp_tail_same<NW>(fl, rec, q)) found = a.x;
            }
            id = found;
        }
    }
    const uint32_t m = id != ~0u ? 1u : 0u;
    const uint32_t incl = wave_inclusive_scan_dpp(m);
    const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
    if (total == 0) return;
    const uint32_t prefix = incl - m;
    if (L.d_region_recs) {
        int32_t *base = L.d_region_recs + ((size_t)c.area + c.rank_base) * 3;
        if (m) {
            typedef int32_t v3i __attribute__((ext_vector_type(3)));
            const v3i rec = {(int32_t)fl.s, (int32_t)(fl.s + fl.r), (int32_t)id};
#ifdef ACGPU_WW_PLAIN_RECS // (A/B build: plain stores and loads, so that a cache might keep the records until the tail reads them -- 0.436 against 0.442 ms, within the noise)
            *reinterpret_cast<v3i *>(base + (size_t)prefix * 3) = rec;
#else
            __builtin_nontemporal_store(rec, reinterpret_cast<v3i *>(base + (size_t)prefix * 3));
#endif
        }
    } else {
        const SlotRange sr = reserve_slots(c, total);
        if (m) store_rec(L, sr.slot(prefix), fl.s, fl.s + fl.r, id, c.rank_base + prefix);
    }
    c.rank_base += total;
}

template <int FOLD, bool LONG, bool PH>
__global__ __launch_bounds__(kTileBlock, kWwBlocksPerCu * (kTileBlock / 256)) void k_ww_pp(DevTables T, TileLaunch L) {
    static_assert(FOLD == 0 || FOLD == 1 || FOLD == 3, "the fold table sits in LDS");
    // FOLD 3: word-character bit and fold delta of a unit in one byte of a page (HostTables::ww_bp_*): no word bits, no delta pages
    constexpr int NW = LONG ? 16 : 8;           // words of folded units per run
    constexpr uint32_t kRunCap = 2 * NW + 1;    // run lengths are counted up to here: longer than every keyword
    typedef PpFlight<NW> Flight;
    __shared__ __attribute__((aligned(16))) uint32_t wbits[FOLD == 3 ? 4 : 2048];
    __shared__ __attribute__((aligned(16))) unsigned char fold_base[FOLD != 0 ? 256 : 16];
    __shared__ __attribute__((aligned(16))) uint16_t pages[FOLD == 1 ? kFoldPagesMax * 256 : 8];
    __shared__ __attribute__((aligned(16))) unsigned char bp_pages[FOLD == 3 ? kBytePagesMax * 256 : 16];
    __shared__ __attribute__((aligned(16))) uint16_t bp_delta[FOLD == 3 ? 128 : 8];
    __shared__ uint32_t wg_words[kFtWords]; // fused tail (acgpu_tile_common.h)
    const bool FT = L.fused_tail != 0;      // kernel-uniform
    if (FT && threadIdx.x == 0) ft_take_number(L, wg_words); // (the answer is there when the tables are: the barrier below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // in front of the waves' rings: the Bloom filter over the keyword hashes, or (PH) the displacements of the perfect hash
    const uint32_t bloom_bytes = PH ? (T.ww_ph_buckets + 7u) / 8u * 16u : (T.ww_bloom_mask + 1u) / 8u;
    uint32_t *bloom = reinterpret_cast<uint32_t *>(smem);
    const uint16_t *disp = reinterpret_cast<const uint16_t *>(smem);
    for (uint32_t w = threadIdx.x; w < bloom_bytes / 16; w += blockDim.x)
        reinterpret_cast<uint4 *>(bloom)[w] = reinterpret_cast<const uint4 *>(PH ? reinterpret_cast<const uint32_t *>(T.ww_ph_disp) : T.ww_bloom)[w];
    if (FOLD != 3)
        for (uint32_t w = threadIdx.x; w < 2048 / 4; w += blockDim.x)
            reinterpret_cast<uint4 *>(wbits)[w] = reinterpret_cast<const uint4 *>(T.wbits)[w];
    FoldLds F{bloom, T.ww_bloom_mask, fold_base, pages};
    if (FOLD == 1) {
        for (uint32_t i = threadIdx.x; i < 256 / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(fold_base)[i] = reinterpret_cast<const uint4 *>(T.fold_pgidx)[i];
        for (uint32_t i = threadIdx.x; i < T.fold_n_pages * 32u; i += blockDim.x)
            reinterpret_cast<uint4 *>(pages)[i] = reinterpret_cast<const uint4 *>(T.fold_pages)[i];
    }
    if (FOLD == 3) {
        for (uint32_t i = threadIdx.x; i < 256 / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(fold_base)[i] = reinterpret_cast<const uint4 *>(T.ww_bp_idx)[i];
        for (uint32_t i = threadIdx.x; i < T.ww_bp_n * 16u; i += blockDim.x) // (a page = 256 bytes)
            reinterpret_cast<uint4 *>(bp_pages)[i] = reinterpret_cast<const uint4 *>(T.ww_bp_pages)[i];
        for (uint32_t i = threadIdx.x; i < 256 / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(bp_delta)[i] = reinterpret_cast<const uint4 *>(T.ww_bp_delta)[i];
    }
    __syncthreads();
    // word-character bit of one unit (the tile loop has its own, eight units at a time)
    auto is_word = [&](uint32_t u) -> uint32_t {
        if (FOLD == 3) return bp_pages[((uint32_t)fold_base[u >> 8] << 8) | (u & 255u)] & 1u;
        return word_bit(wbits, u);
    };

    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t wg = FT ? __builtin_amdgcn_readfirstlane(wg_words[0]) : blockIdx.x; // (fused tail: workgroups are numbered by their start)
    const uint32_t n_waves_wg = blockDim.x / kWave; // (16; the fused tail's launch may say otherwise: tunable ww_block)
    const uint32_t wave_global = wg * n_waves_wg + wave_in_block;
    unsigned char *mine = smem + bloom_bytes + wave_in_block * kPpWaveBytes;
    uint16_t *ring = reinterpret_cast<uint16_t *>(mine);
    unsigned char *bits = mine + kPpRingUnits * 2;
    uint16_t *list = reinterpret_cast<uint16_t *>(mine + kPpRingUnits * 2 + kPpBitBytes);
    TileCtx c{&T, &L, nullptr, 0, 0, 0u, 0};
    c.wg = wg;

    const uint32_t base8 = L.own_begin & ~7u;
    const uint32_t R = L.region_units;
    const uint32_t n = L.n_units;
    const uint16_t *hay = L.d_hay;
    // the wave's span: regions_per_wave regions, or (fused tail) its share of the workgroup's tiles (TileLaunch::ft_total16)
    uint32_t first_region = wave_global * L.regions_per_wave, tile0, span_begin, span_end, boundary;
    bool has_work;
    if (FT) {
        const uint32_t tb = ww_ft_tiles_before(wg, gridDim.x, L.ft_total16, L.ft_ramp_pm, n_waves_wg);
        const uint32_t q = (ww_ft_tiles_before(wg + 1u, gridDim.x, L.ft_total16, L.ft_ramp_pm, n_waves_wg) - tb) / n_waves_wg;
        tile0 = base8 + (tb + wave_in_block * q) * kTileUnits;
        has_work = q != 0u && tile0 < L.own_end;
        span_begin = max(L.own_begin, tile0);
        span_end = min(L.own_end, tile0 + q * kTileUnits);
        boundary = ~0u; // (no regions: ranks count through the span, records go to the wave's own area)
        first_region = 0;
        c.area = (tile0 - base8) / 2u + wave_global;
    } else {
        has_work = first_region < L.n_regions;
        if (!has_work) return;
        const uint32_t last_region = min(first_region + L.regions_per_wave, L.n_regions);
        span_begin = max(L.own_begin, base8 + first_region * R);
        span_end = base8 + last_region * R;
        if (span_end > L.own_end || last_region == L.n_regions) span_end = L.own_end;
        tile0 = base8 + first_region * R;
        boundary = tile0 + R;
        c.area = first_region * L.region_cap;
    }

    uint32_t region = first_region;
    c.region = region;
    uint32_t carry = tile0 >= 1 && tile0 - 1 < n ? is_word(hay[tile0 - 1]) : 0u; // bit of the unit before the tile

    // the lane's 8 units of the tile at `cur` (zeros beyond the buffer)
    auto load_tile = [&](uint32_t cur) -> uint4 {
        const uint32_t v = cur + lane * 8;
        if (cur + kTileUnits <= n) return *reinterpret_cast<const uint4 *>(hay + v); // wave-uniform: the whole tile exists
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (v + 8 <= n) {
            w = *reinterpret_cast<const uint4 *>(hay + v);
        } else if (v < n) { // the last, partial vector of the buffer: unit by unit
            unsigned long long lo = 0, hi = 0;
            for (uint32_t j = 0; j < 8 && v + j < n; ++j) {
                const unsigned long long u = (unsigned long long)hay[v + j] << (16 * (j & 3u));
                if (j < 4) lo |= u;
                else hi |= u;
            }
            w = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        }
        return w;
    };
    // Stage a tile: word bits and folded units of the lane's 8 units into the ring slot of tile j; returns the lane's run starts
    auto stage = [&](uint32_t j, const uint4 w) -> uint32_t {
        const uint32_t cur = tile0 + j * kTileUnits, v = cur + lane * 8, slot = j & 1u;
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        uint32_t wm, fdw[4]; // word-character bits of the lane's 8 units; (FOLD 3) their folded units, two per word
        if (FOLD == 3) {
            // one byte per unit {delta index << 1 | word character}: page index, page byte, delta -- three dependent LDS reads per
            // unit, eight units in flight; the pieces of the addresses come straight out of the packed words
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t d = ww[k >> 1];
                const uint32_t hi8 = (k & 1) ? d >> 24 : (d >> 8) & 0xffu, lo8 = (k & 1) ? (d >> 16) & 0xffu : d & 0xffu;
                e[k] = bp_pages[((uint32_t)fold_base[hi8] << 8) | lo8];
            }
            wm = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) wm |= (e[k] & 1u) << k;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t d2 = (uint32_t)bp_delta[e[2 * k] >> 1] | ((uint32_t)bp_delta[e[2 * k + 1] >> 1] << 16);
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                fdw[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, ww[k]) + __builtin_bit_cast(u16x2, d2));
            }
        } else {
            wm = ACGPU_DBG(L, 8u) ? ((ww[0] ^ ww[2]) & 0xbfu) | 1u : word_bits8<8>(wbits, ww); // 8: ablation, no word-bit lookups (timing only)
        }
        if (cur + kTileUnits > n) wm &= (1u << (v < n ? min(n - v, 8u) : 0u)) - 1u; // (wave-uniform) nothing beyond the buffer is a word
        const uint32_t prev = from_prev_lane(wm >> 7, carry);
        carry = __builtin_amdgcn_readlane(wm, 63) >> 7;
        uint32_t sm = wm & ~((wm << 1) | (prev & 1u)) & 0xffu;
        if (cur < span_begin || cur + kTileUnits > span_end) { // wave-uniform: a tile at the edges of the span
            const uint32_t first = span_begin > v ? min(span_begin - v, 8u) : 0u;
            const uint32_t last = span_end > v ? min(span_end - v, 8u) : 0u;
            sm &= ((1u << last) - 1u) & ~((1u << first) - 1u);
        }
        uint4 fd;
        if (FOLD == 3) {
            fd = make_uint4(fdw[0], fdw[1], fdw[2], fdw[3]);
        } else {
            uint32_t f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = ww_fold<FOLD == 3 ? 0 : FOLD>(T, F, (ww[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            fd = make_uint4(f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16));
        }
        *reinterpret_cast<uint4 *>(ring + slot * kTileUnits + lane * 8) = fd;
        bits[slot * (kTileUnits / 8) + lane] = (unsigned char)wm;
        if (slot == 0 && lane < 8) { // the copy behind slot 1
            if (lane < 4) *reinterpret_cast<uint4 *>(ring + 2 * kTileUnits + lane * 8) = fd;
            bits[2 * (kTileUnits / 8) + lane] = (unsigned char)wm;
        }
        return sm;
    };

    // the run starts of a staged tile (its lanes' start masks), tile relative and in text order, into the list; returns their number
    auto build_list = [&](uint32_t sm) -> uint32_t {
        const uint32_t k = __popc(sm);
        const uint32_t incl = wave_inclusive_scan_dpp(k);
        uint32_t at = incl - k, m = sm;
        while (__any(m != 0)) {
            if (m != 0) {
                list[at++] = (uint16_t)(lane * 8 + (uint32_t)__builtin_ctz(m));
                m &= m - 1;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t cnt = __builtin_amdgcn_readlane(incl, kWave - 1);
        return ACGPU_DBG(L, 1u) ? 0u : cnt; // 1: ablation, run starts are dropped
    };
    // One batch of the listed run starts of tile j (entries b0 .. b0+63): run length and folded units from the ring, both
    // hashes, Bloom test, slots.  Tile j + 1 has been staged.
    auto batch = [&](uint32_t j, uint32_t cnt, uint32_t b0) -> Flight {
        const uint32_t slot = j & 1u, q = b0 + lane;
        const bool act = q < cnt;
        const uint32_t p = act ? (uint32_t)list[q] : 0u;
        // run length: the word-character bits from p on (17 or 33 of them decide: keywords have at most 16 / 32 units)
        uint32_t r;
        if (LONG) {
            struct __attribute__((packed, aligned(1), may_alias)) Bits8 { unsigned long long v; };
            const unsigned long long wb = reinterpret_cast<const Bits8 *>(bits + slot * (kTileUnits / 8) + (p >> 3))->v >> (p & 7u);
            r = (uint32_t)__builtin_ctzll(~wb | (1ull << kRunCap));
        } else {
            struct __attribute__((packed, aligned(1), may_alias)) Bits4 { uint32_t v; };
            const uint32_t wb = reinterpret_cast<const Bits4 *>(bits + slot * (kTileUnits / 8) + (p >> 3))->v >> (p & 7u);
            r = (uint32_t)__builtin_ctz(~wb | (1u << kRunCap));
        }
        struct __attribute__((packed, aligned(2), may_alias)) Run16 { uint32_t d[8]; };
        const Run16 run = *reinterpret_cast<const Run16 *>(ring + slot * kTileUnits + p);
        Flight fl;
        uint32_t h = T.ww_seed, g = T.ww_seed;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // units 2i and 2i+1 of the run, zero from unit r on.  (Plain C on purpose: a packed-arithmetic version of this in
            // inline assembly, and a two-instruction h*33+w, made the kernel slower -- see DESIGN.md 4.4.)
            // (The first 16 units are hashed as 8 words whatever the length.  Skipping the words that are zero for every run that
            // can be a keyword -- config 5: words 6 and 7 -- behind wave-uniform branches, one multiplication and one rotation
            // in their place, made the kernel 20 % SLOWER on the same box, 0.455 against 0.374 ms: EXPERIMENTS.md, round 6.)
            const int m = (int)r - 2 * i;
            fl.fw[i] = m >= 2 ? run.d[i] : (m == 1 ? (run.d[i] & 0xffffu) : 0u);
            h = ww_hash_step(h, fl.fw[i]);
            g = ww_hash2_step(g, fl.fw[i]);
        }
        if (LONG) {
            // units 16..31: only when some run of the batch goes that far (wave-uniform; rare in a text of words) -- further
            // words are hashed only as far as the run goes (k_ww_tile's chunk loop and the builder do the same)
#pragma unroll
            for (int i = 8; i < NW; ++i) fl.fw[i] = 0u;
            if (__any(act && r > 16u)) {
                const Run16 more = *reinterpret_cast<const Run16 *>(ring + slot * kTileUnits + p + 16);
#pragma unroll
                for (int i = 8; i < NW; ++i) {
                    const int m = (int)r - 2 * i;
                    fl.fw[i] = m >= 2 ? more.d[i - 8] : (m == 1 ? (more.d[i - 8] & 0xffffu) : 0u);
                    if (m > 0) {
                        h = ww_hash_step(h, fl.fw[i]);
                        g = ww_hash2_step(g, fl.fw[i]);
                    }
                }
            }
        }
        h = ww_hash_final(h);
        bool probing = act && r <= T.max_len && !ACGPU_DBG(L, 2u); // 2: ablation, no table lookup
        if (PH) { // the bucket's displacement (LDS) names the one slot a keyword with these hashes would sit in
            const uint32_t d = disp[ww_ph_bucket(h, T.ww_ph_buckets)];
            fl.s1 = probing ? ww_ph_slot(g, h, d, T.ww_ph_n) : 0u;
            fl.s2 = 0u;
        } else {
        if (!ACGPU_DBG(L, 4u)) { // 4: ablation, no Bloom filter in front of the table
            const uint32_t b1 = ww_bloom_bit1(h, F.bloom_mask), b2 = ww_bloom_bit2(h, F.bloom_mask);
            probing = probing && ((F.bloom[b1 >> 5] >> (b1 & 31)) & (F.bloom[b2 >> 5] >> (b2 & 31)) & 1u);
        }
        fl.s1 = probing ? ww_slot1(h, T.ww_fat_mask) : 0u;
        fl.s2 = probing ? (ACGPU_DBG(L, 16u) ? fl.s1 ^ 1u : ww_slot2(h, g, T.ww_fat_mask)) : 0u; // 16: ablation (timing only), both slots in one line
        }
        fl.n = cnt > b0 ? min(cnt - b0, (uint32_t)kWave) : 0u;
        fl.probing = probing;
        fl.s = tile0 + j * kTileUnits + p;
        fl.r = r;
        fl.tag = ww_tag(h, r);
        return fl;
    };

    // more than 64 run starts in a tile are rare: all but the tile's last batch are probed and compared at once
    auto early_batches = [&](uint32_t j, uint32_t cnt, uint32_t b_last) {
        for (uint32_t b0 = 0; b0 < b_last; b0 += kWave) {
            const Flight now = batch(j, cnt, b0);
            pp_consume<NW, PH>(c, now, pp_issue<NW, PH>(T, now));
        }
    };
    // Pass j: the probes of tile j's last batch go out first; then tile j+2 is staged, tile j+1's run starts are listed and its
    // last batch is hashed -- LDS and VALU work, under which the probes arrive -- and then tile j's batch is compared and
    // emitted.  The stream runs one tile ahead of the staging in ONE set of registers (the load goes out right after the
    // registers are staged and has a whole pass to arrive; rotating register sets would copy registers whose loads are in
    // flight, and such a copy waits for everything issued before it).
    if (has_work) {
    Flight fl;
    uint32_t sm_next;
    {
        const uint32_t sm0 = stage(0, load_tile(tile0));
        sm_next = stage(1, load_tile(tile0 + kTileUnits));
        const uint32_t cnt = build_list(sm0);
        const uint32_t b_last = cnt ? ((cnt - 1) & ~(uint32_t)(kWave - 1)) : 0u;
        early_batches(0, cnt, b_last);
        fl = batch(0, cnt, b_last);
    }
    uint4 nx = load_tile(tile0 + 2 * kTileUnits);
    uint32_t prio_turn = 0;
    for (uint32_t j = 0;; ++j) {
        const uint32_t cur = tile0 + j * kTileUnits; // tile j: `fl` is its last batch
        if (cur >= span_end) break; // wave-uniform
#ifndef ACGPU_NO_SETPRIO
        switch ((wave_in_block / 4u + prio_turn++) & 3u) { // (see k_ac_tile: the waves of a SIMD take turns at every issue priority)
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        const PpProbe pr = pp_issue<NW, PH>(T, fl);
        const uint32_t sm2 = stage(j + 2, nx);
        nx = load_tile(cur + 3 * kTileUnits);
        const uint32_t cnt = build_list(sm_next); // tile j + 1 (none beyond the span: its start mask is empty)
        const uint32_t b_last = cnt ? ((cnt - 1) & ~(uint32_t)(kWave - 1)) : 0u;
        const Flight nf = batch(j + 1, cnt, b_last);
        pp_consume<NW, PH>(c, fl, pr);
        if (cur + kTileUnits >= boundary && cur + kTileUnits < span_end) { // tile j + 1 opens the next region
            if (lane == 0) L.d_region_counts[region] = c.rank_base;
            c.rank_base = 0;
            ++region;
            c.region = region;
            c.area = region * L.region_cap;
            boundary += R;
        }
        early_batches(j + 1, cnt, b_last);
        fl = nf;
        sm_next = sm2;
    }
    } // has_work
    if (FT) {
        // The fused tail (see acgpu_tile_common.h and k_ac_tile): the wave's records lie in its own area, in the reference's
        // order; they go behind those of the workgroups with lower numbers and of the workgroup's waves before this one.
        // A wave copies what it wrote itself: nothing of another wave's is read but sixteen counts in LDS.
        const uint32_t kWaves = n_waves_wg;
        const uint32_t cnt = c.rank_base;
#ifdef ACGPU_TIMING
        const unsigned long long ft_t_scan = __builtin_amdgcn_s_memtime(); // (one clock for the whole chip)
#endif
        if (lane == 0) wg_words[2 + wave_in_block] = cnt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the wave's own record stores)
        __syncthreads();
        const uint32_t wv = lane < kWaves ? wg_words[2 + lane] : 0u;
        const uint32_t wv_incl = wave_inclusive_scan_dpp(wv);
        const uint32_t total_wg = __builtin_amdgcn_readlane(wv_incl, kWave - 1); // (lanes beyond the workgroup's waves hold 0)
        const uint32_t before_me = __builtin_amdgcn_readlane(wv_incl - wv, wave_in_block);
        if (threadIdx.x == 0) ft_publish(L, wg, total_wg);
        // the first records of the area are asked for before the wait for the lower numbers: they are there when it ends.  Two
        // sets of registers in turn, the next set's loads in front of this set's stores: a wait for loads is a wait for every
        // older store's acknowledgement as well (one counter), so a loop of load-then-store pays both latencies per step
        typedef int32_t v3i __attribute__((ext_vector_type(3)));
        const int32_t *src = L.d_region_recs + (size_t)c.area * 3; // (12-byte records: a v3i is 16 bytes wide as an array element)
        constexpr uint32_t kPer = 8, kStep = kPer * kWave;
        v3i ra[kPer], rb[kPer];
        auto load_set = [&](v3i (&r)[kPer], uint32_t k0) {
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q) {
                const uint32_t k = k0 + q * kWave + lane;
                r[q] = v3i{0, 0, 0};
#ifdef ACGPU_WW_PLAIN_RECS
                if (k < cnt) r[q] = *reinterpret_cast<const v3i *>(src + (size_t)k * 3);
#else
                if (k < cnt) r[q] = __builtin_nontemporal_load(reinterpret_cast<const v3i *>(src + (size_t)k * 3));
#endif
            }
        };
        load_set(ra, 0);
        const uint32_t below = ft_below(L, wg, wg_words);
#ifdef ACGPU_TIMING
        const unsigned long long ft_t_below = __builtin_amdgcn_s_memtime();
#endif
        const unsigned long long at0 = (unsigned long long)below + before_me;
        auto store_set = [&](const v3i (&r)[kPer], uint32_t k0) {
#pragma unroll
            for (uint32_t q = 0; q < kPer; ++q) {
                const uint32_t k = k0 + q * kWave + lane;
                const unsigned long long at = at0 + k;
                if (k >= cnt || at >= L.out_cap) continue;
                if (L.out_map) *reinterpret_cast<v3i *>(reinterpret_cast<int32_t *>(L.d_out) + at * 3) = r[q];
                else reinterpret_cast<int2 *>(L.d_out)[at] = make_int2(r[q].x, r[q].y);
            }
        };
        for (uint32_t k0 = 0; k0 < cnt; k0 += 2 * kStep) {
            if (k0 + kStep < cnt) load_set(rb, k0 + kStep);
            store_set(ra, k0);
            if (k0 + 2 * kStep < cnt) load_set(ra, k0 + 2 * kStep);
            if (k0 + kStep < cnt) store_set(rb, k0 + kStep);
        }
#ifdef ACGPU_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && L.d_timing) { // per wave: when its scan ended, when the counts before it were there, when its copy was done
            L.d_timing[(size_t)wave_global * 8 + 0] = ft_t_scan;
            L.d_timing[(size_t)wave_global * 8 + 1] = ft_t_below;
            L.d_timing[(size_t)wave_global * 8 + 2] = __builtin_amdgcn_s_memtime();
            L.d_timing[(size_t)wave_global * 8 + 3] = cnt;
        }
#endif
        ft_report(L, wg, (unsigned long long)below + total_wg);
        return;
    }
    if (lane == 0) L.d_region_counts[region] = c.rank_base;
    for (uint32_t i = lane; i < c.res_left; i += kWave)
        if (c.res_cur + i < c.slot_limit) store_rec(L, c.res_cur + i, 0, 0, 0, ~0u); // (not into the next slice)
}

// Literal restatement of S/WholeWordMatchMap.java:155-240 by ONE lane, for word-character tables that are not
// fold-consistent.  The whole haystack is one shard.  Records come out in order; count in *counter.
__global__ void k_ww_sequential(DevTables T, const uint16_t *hay, uint32_t len, void *out, uint64_t cap, int record_kind,
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
    uint32_t node = 0, idx = 0;
    while (idx < len) {
        const uint32_t raw = hay[idx];
        const uint32_t c = T.cs ? raw : (uint32_t)T.lower[raw];
        const uint32_t next = hashed_goto(T.hkeys, T.hvals, T.hmask, node, c);
        if (next == ~0u) {
            if (!(T.wflags[raw] & 2u)) { // !wordChars[c], c = folded unit
                if (node != 0 && T.term_id[node] != ~0u) emit(idx - T.depth[node], idx, T.term_id[node]);
            } else {
                while (++idx < len && (T.wflags[hay[idx]] & 1u)) {
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
    if (node != 0 && T.term_id[node] != ~0u) emit(idx - T.depth[node], idx, T.term_id[node]);
    *counter = n;
}

// the position-parallel form serves keywords of at most 16 units whose fold table (if any) fits LDS, when its LDS fits next
// to the Bloom filter (tile_debug bit 268435456 keeps k_ww_tile: A/B; bit 256, the trie-walk verification, exists only there)
// the byte pages serve the scan they were built for (case-insensitive, the automaton's own word bits)
static bool ww_pp_byte_pages(const DevTables &t) { return !t.cs && t.ww_bp_n != 0 && t.ww_bp_n <= kBytePagesMax && t.wbits == t.ww_bp_wbits; }
static size_t ww_pp_fixed_lds(const DevTables &t) { // the kernel's static LDS
    const int fold = t.cs ? 0 : (ww_pp_byte_pages(t) ? 3 : ww_fold_pages_in_lds(t) ? 1 : 2);
    return (fold == 3 ? 16 + 256 + 16 + kBytePagesMax * 256 + 256 : fold == 1 ? 8192 + 256 + kFoldPagesMax * 512 + 32 : 8192 + 64) + kFtWords * 4;
}
size_t ww_pp_lds_total(const DevTables &t, const TileLaunch &l, int block_threads) {
    return ww_pp_lds_bytes(block_threads, t, ww_pp_perfect(t, l)) + ww_pp_fixed_lds(t);
}
static bool ww_pp_usable(const DevTables &t, const TileLaunch &l) {
    const int fold = t.cs ? 0 : (ww_pp_byte_pages(t) ? 3 : ww_fold_pages_in_lds(t) ? 1 : 2);
    return fold != 2 && t.max_len <= kPpMaxLen && !(l.debug & (256u | 268435456u)) && ww_pp_lds_total(t, l, l.block) <= 160 * 1024;
}

bool ww_pp_serves(const DevTables &t, const TileLaunch &l) { return ww_pp_usable(t, l); }

hipError_t launch_ww_tile(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name) {
    int fold = t.cs ? 0 : (ww_fold_pages_in_lds(t) ? 1 : 2);
    if (ww_pp_usable(t, l)) {
        if (ww_pp_byte_pages(t)) fold = 3;
        const bool ph = ww_pp_perfect(t, l);
        const size_t lds = ww_pp_lds_bytes(l.block, t, ph);
        const bool lng = t.max_len > 16;
        static thread_local char name[48];
        std::snprintf(name, sizeof(name), "k_ww_pp<%d, %s, %s>", fold, lng ? "true" : "false", ph ? "true" : "false");
        if (kernel_name) *kernel_name = name;
#define ACGPU_WW_PP(F, LG, PHV)                                                                                              \
    do {                                                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ww_pp<F, LG, PHV>),                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return e;                                                                                       \
        ACGPU_LAUNCH_EV((k_ww_pp<F, LG, PHV>), dim3(l.grid), dim3(l.block), lds, stream, l.ev_start, l.ev_stop, t, l);       \
        return hipGetLastError();                                                                                            \
    } while (0)
        if (fold == 0) {
            if (lng) { if (ph) ACGPU_WW_PP(0, true, true); else ACGPU_WW_PP(0, true, false); }
            else { if (ph) ACGPU_WW_PP(0, false, true); else ACGPU_WW_PP(0, false, false); }
        } else if (fold == 3) {
            if (lng) { if (ph) ACGPU_WW_PP(3, true, true); else ACGPU_WW_PP(3, true, false); }
            else { if (ph) ACGPU_WW_PP(3, false, true); else ACGPU_WW_PP(3, false, false); }
        } else {
            if (lng) { if (ph) ACGPU_WW_PP(1, true, true); else ACGPU_WW_PP(1, true, false); }
            else { if (ph) ACGPU_WW_PP(1, false, true); else ACGPU_WW_PP(1, false, false); }
        }
#undef ACGPU_WW_PP
        fold = t.cs ? 0 : (ww_fold_pages_in_lds(t) ? 1 : 2); // (not reached: every branch above returns)
    }
    const void *fn = fold == 0 ? reinterpret_cast<const void *>(&k_ww_tile<0>)
                   : fold == 1 ? reinterpret_cast<const void *>(&k_ww_tile<1>) : reinterpret_cast<const void *>(&k_ww_tile<2>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
    if (e != hipSuccess) return e;
    if (fold == 0) ACGPU_LAUNCH_EV(k_ww_tile<0>, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, l.ev_start, l.ev_stop, t, l);
    else if (fold == 1) ACGPU_LAUNCH_EV(k_ww_tile<1>, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, l.ev_start, l.ev_stop, t, l);
    else ACGPU_LAUNCH_EV(k_ww_tile<2>, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, l.ev_start, l.ev_stop, t, l);
    if (kernel_name) *kernel_name = fold == 0 ? "k_ww_tile<0>" : fold == 1 ? "k_ww_tile<1>" : "k_ww_tile<2>";
    return hipGetLastError();
}

hipError_t launch_ww_sequential(const DevTables &t, const uint16_t *d_hay, uint32_t len, void *d_out, uint64_t cap,
                                int record_kind, unsigned long long *d_counter, hipStream_t stream) {
    hipLaunchKernelGGL(k_ww_sequential, dim3(1), dim3(64), 0, stream, t, d_hay, len, d_out, cap, record_kind, d_counter);
    return hipGetLastError();
}

} // namespace acgpu
