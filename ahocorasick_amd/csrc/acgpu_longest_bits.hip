// acgpu_longest_bits.hip -- LongestMatchSet over a two-letter alphabet: the text as ONE BIT per unit (gfx950).
//
// Replaces, for such dictionaries, the per-unit loop of LongestMatchSet.match (S/LongestMatchSet.java:192-265: the
// automaton step and the pending-match queue S/SetMatchQueue.java:45-95).  What that loop reports is the greedy chain
// pos -> pos + max(L[pos], 1), L[pos] = the longest keyword that starts at pos.  The walk pipeline of
// acgpu_longest.hip computes L for EVERY position and writes it to memory (0.54 GB at config 4), then finds a
// synchronisation point per tile and follows the chain through those lengths.  Here nothing but the chain's own positions
// is ever looked up:
//  * a wave owns a region of 64 segments of 1024 positions; it streams the region's text once and keeps it in LDS as one
//    bit per unit (32 words per segment + copies of the next segment's first words, stride 35: lanes that walk their
//    segments at the same pace sit in different banks);
//  * L[p] comes from the path-compressed keyword trie (acgpu_build.cpp 6c, 17.5 KiB of LDS): the text's next 9 units
//    index an entry {label, terminal bits, meta, next}; the label -- the one-child path below that node, 31 units -- is
//    compared with the text by one xor and a count of trailing zeros; longer paths and branches take further entries (rare);
//  * pass 1: lane j follows a chain through the last `runup` positions of segment j-1 (from wherever that stretch begins).
//    Chains that start at different positions merge within a few matches, so where it leaves the segment is (with
//    overwhelming probability) where the true chain does: lane j's entry.  Measured at config 4 (2^29 units, 524 288 segments):
//    a run-up of 128 positions misses 2509 entries, 256 positions 18, 512 positions none (0.766 per step: 5e-10 per segment);
//    the call starts with 512 and, should the check below fail, is redone once with the whole segment (1024);
//  * pass 2: lane j follows the chain of its own segment from that entry, counts the matches and marks their starts -- in
//    the place of the text words it has left behind.  Then every lane's exit is compared with its neighbour's entry (the
//    first region's first entry is the call's chain entry, so equal everywhere means exact everywhere); a difference
//    anywhere -- or a unit outside the alphabet -- raises the call's bail flag and the call is redone: once with a run-up of a
//    whole segment, then by the walk pipeline;
//  * every position of a text over the alphabet starts a match (the builder checks that every letter is a keyword), so the
//    chain's positions ARE the match starts and the next position of the chain is a match's end.  A region publishes its
//    count, parks its marks in memory and, one region later, writes its records straight to their final place: ranks from
//    the counts of the regions before it (two-level look-back), starts staged in LDS, 16-byte coalesced stores.
// HBM-bound: 2 B per unit read once, 8 B per match written once (+ 2 x 1 bit per unit for the parked marks).  No length
// array, no synchronisation pass, no separate prefix-sum or emit kernel.
#include <cstdio>
#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

constexpr int kBitsWaves = 16;                  // one workgroup of 16 waves per CU
constexpr int kBitsBlock = kBitsWaves * kWave;
constexpr uint32_t kBitsSegWords = 32;          // a lane's segment: 32 words = 1024 positions
constexpr uint32_t kBitsSegUnits = 32 * kBitsSegWords;
constexpr uint32_t kBitsLook = 3;               // copies of the next segment's first words behind a segment's own
constexpr uint32_t kBitsStride = kBitsSegWords + kBitsLook; // 35 words: odd, so equal word numbers of 32 segments are 32 banks
constexpr uint32_t kBitsSegs = 64;
constexpr uint32_t kBitsRegionUnits = kBitsSegs * kBitsSegUnits;            // 65536 positions per wave and round
constexpr uint32_t kBitsWaveWords = (kBitsSegs + 1) * kBitsStride + 1;      // segments -1 .. 63, and "segment 64, word 0"
constexpr uint32_t kBitsRegionWords = kBitsRegionUnits / 32u;                // a region's marks: 2048 words
constexpr uint32_t kBitsTextWords = kBitsRegionWords + 8u;                   // MAP: a region's parked text, the words behind it, padding to 32 bytes
constexpr unsigned long long kBitsPub = 1ull << 62, kBitsCount = (1ull << 40) - 1ull; // LongestBitsLaunch::d_agg / d_blk
constexpr uint32_t kBitsTileUnits = 2048;                                   // 64 lanes x 32 units: one word per lane
constexpr uint32_t kBitsTextMax = 33u * 32u;    // a 32-bit window may begin below this segment-relative position
static_assert(kBitsWaves * kBitsWaveWords * 4 + kBitsTabEntries * 16 + 4 <= 160 * 1024, "LDS");
static_assert(kBitsRK + 32 <= 64, "a first-level lookup reads three words");

struct BitsCtx {
    const uint16_t *hay;
    const uint32_t *dfa;
    uint32_t n_units, n_cls, base, span;
};

__device__ __forceinline__ uint32_t bits_pk_sub(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
__device__ __forceinline__ uint32_t bits_pk_max(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t bits_ffbl(uint32_t x) { // trailing zeros; all ones for x == 0
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t bits_ffbh(uint32_t x) { // leading zeros; all ones for x == 0
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// 32 units (16 registers of two) -> 32 bits, unit i at bit i; dmax collects the largest code met (packed)
__device__ __forceinline__ uint32_t bits_pack(const uint32_t (&w)[16], uint32_t base2, uint32_t &dmax) {
    uint32_t d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = bits_pk_sub(w[i], base2);
#pragma unroll
    for (int i = 0; i < 16; ++i) dmax = bits_pk_max(dmax, d[i]);
    uint32_t a = d[0], b = d[8];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        a = (d[i] << (2 * i)) | a; // even units at bits 0, 2, .. 14; odd units at bits 16, 18, .. 30
        b = (d[8 + i] << (2 * i)) | b;
    }
    const uint32_t pa = (a | (a >> 15)) & 0xffffu, pb = b | (b >> 15);
    return (pb << 16) | pa;
}

// the longest keyword at absolute position p, through the trie in global memory (the rare ways out of the fast path)
__device__ __noinline__ uint32_t bits_slow(const BitsCtx &c, uint32_t p) {
    uint32_t node = 0, best = 0, j = p;
    while (j < c.n_units) {
        const uint32_t dlt = (uint32_t)c.hay[j] - c.base;
        const uint32_t g = c.dfa[(uint64_t)node * c.n_cls + (dlt < c.span ? dlt + 1u : 0u)];
        if (!g) break;
        node = g & 0x7fffffffu;
        ++j;
        if (g >> 31) best = j - p;
    }
    return best;
}

// entry e's label against the 32 text bits tw behind the walk's first `depth` units; raises best to the longest keyword
// that ends inside the matched part; true: the whole label matched AND the trie goes on behind it (meta bits 16-21: the
// label's length for such entries, 63 -- what no count of matched units equals -- for the others)
__device__ __forceinline__ bool bits_label(const uint4 &e, uint32_t tw, uint32_t depth, uint32_t &best) {
    const uint32_t len = e.z & 63u;                        // at most 31
    const uint32_t m = min(bits_ffbl(tw ^ e.x), len);      // matched units of the label
    const uint32_t tm = e.y & ((1u << m) - 1u);            // keywords that end inside them
    const uint32_t h = bits_ffbh(tm);                      // (all ones: none)
    best = tm ? depth + 32u - h : best;
    return m == ((e.z >> 16) & 63u);
}

// what follows a label that matched whole: further entries (the path is longer than 31 units, or branches), and the walk
// through global memory where the table or the lane's text in LDS ends
__device__ __noinline__ uint32_t bits_more(const BitsCtx &c, const uint4 *tab, const uint32_t *seg, uint32_t q, uint32_t p, uint4 e, uint32_t best) {
    uint32_t depth = kBitsRK;
    for (;;) {
        depth += e.z & 63u;
        const uint32_t kind = (e.z >> 6) & 3u;
        if (kind == kBitsLeaf) return best;
        uint32_t t = q + depth;
        if (kind == kBitsDeep || t + 1u >= kBitsTextMax) return bits_slow(c, p);
        if (kind == kBitsJunction) {
            const uint32_t code = (seg[t >> 5] >> (t & 31u)) & 1u;
            e = tab[e.w + code];
            if (!(e.z & kBitsAlive)) return best;
            ++depth;
            ++t;
            if ((e.z >> 8) & 0xffu) best = depth;
        } else {
            e = tab[e.w];
        }
        const uint32_t *v = seg + (t >> 5);
        if (!bits_label(e, __builtin_amdgcn_alignbit(v[1], v[0], t & 31u), depth, best)) return best;
    }
}

// A chain through one segment: from segment-relative position q to the first chain position at or behind qend (returned).
// Every step is the longest keyword that starts at the current position -- what LongestMatchSet's queue would report from
// there (at least 1: every letter is a keyword).  seg: the lane's segment in the wave's LDS image (the words behind it: the
// next segment's first ones); seg_pos: its absolute position; qsafe: matches that end behind this relative position may
// reach beyond the buffer, whose missing units were packed as the first letter.
//  * The text comes through a window of registers -- the words k, k+1, k+2 of the segment, word k+3 on its way -- that moves
//    on when the chain enters the next word: the step's dependent chain holds ONE LDS round trip (the table entry), not two.
//  * MARK (pass 2): the step's position is counted and marked; the marks of a word are written in the place of its text when
//    the chain has left it, the words the chain does not touch become zeros.
template <bool MARK>
__device__ __forceinline__ uint32_t bits_walk(const BitsCtx &c, const uint4 *tab, uint32_t *seg, uint32_t q, uint32_t qend, uint32_t seg_pos,
                                              uint32_t qsafe, uint32_t &cnt) {
    constexpr uint32_t RK = kBitsRK;
    uint32_t cur = 0, cw = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (q < qend) {
        cur = q >> 5;
        if (MARK)
            for (uint32_t z = 0; z < cur; ++z) seg[z] = 0u;
        const uint32_t *p = seg + cur;
        t0 = p[0]; t1 = p[1]; t2 = p[2]; t3 = p[3];
    }
    while (q < qend) {
        const uint32_t k = q >> 5;
        if (k != cur) { // the chain has left word cur
            if (MARK) {
                seg[cur] = cw;
                cw = 0u;
            }
            if (__builtin_expect(k != cur + 1u, 0)) { // (a match of 32 units or more)
                if (MARK)
                    for (uint32_t z = cur + 1u; z < k; ++z) seg[z] = 0u;
                const uint32_t *p = seg + k;
                t0 = p[0]; t1 = p[1]; t2 = p[2]; t3 = p[3];
            } else {
                t0 = t1; t1 = t2; t2 = t3;
                t3 = seg[k + 3u];
            }
            cur = k;
        }
        const uint32_t sh = q & 31u;
        const uint32_t lo = __builtin_amdgcn_alignbit(t1, t0, sh), mid = __builtin_amdgcn_alignbit(t2, t1, sh);
        const uint4 e = tab[lo & ((1u << RK) - 1u)];
        uint32_t best = (e.z >> 8) & 0xffu; // the longest keyword among the first RK units
        const bool more = bits_label(e, __builtin_amdgcn_alignbit(mid, lo, RK), RK, best);
        if (__builtin_expect(more || q + best > qsafe, 0)) { // rare: the trie goes on behind the label; the end of the buffer
            if (more) best = bits_more(c, tab, seg, q, seg_pos + q, e, best);
            if (q + best > qsafe) best = bits_slow(c, seg_pos + q);
            best = max(best, 1u); // (0: a unit outside the alphabet -- the call bails out; the chain moves on whatever it reads)
        }
        if (MARK) {
            cw |= 1u << sh;
            ++cnt;
        }
        q += best;
    }
    if (MARK) {
        seg[cur] = cw;
        for (uint32_t z = cur + 1u; z < kBitsSegWords; ++z) seg[z] = 0u;
    }
    return q;
}

// wave64 inclusive prefix sum (DPP row shifts and row broadcasts), and sums over the wave
__device__ __forceinline__ uint32_t bits_wave_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t bits_wave_sum(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)bits_wave_scan(x), 63); }
__device__ __forceinline__ unsigned long long bits_wave_sum64(unsigned long long x) { // (values below 2^62; a wave's sum below 2^63)
    for (int d = 32; d > 0; d >>= 1) x += (unsigned long long)__shfl_xor((long long)x, d);
    return x;
}

#ifdef ACGPU_TIMING
__device__ unsigned long long g_bits_timing[8];
__device__ unsigned long long g_bits_tail[2]; // the last region of a wave: waiting for the counts before it, writing its records // per-wave sums: gate wait, text, pass 1, pass 2, look-back, records, total; waves
#define BITS_MARK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); bt[i] += t_ - bt0; bt0 = t_; }
#else
#define BITS_MARK(i)
#endif
// MAP: {start, end, keyword id} records.  The region's text bits are parked in memory beside its marks (they are overwritten by
// the marks in LDS); when the records are written, one region later, a record's keyword is its own bits in that text: one
// lookup of (length, bits) in HostTables::bits_idkeys per record (keywords beyond 32 units: a walk through the table in memory).
template <bool MAP>
__global__ __launch_bounds__(kBitsBlock) void k_longest_bits(DevTables T, LongestBitsLaunch L) {
    __shared__ __attribute__((aligned(16))) uint4 tab[kBitsTabEntries];
    __shared__ uint32_t img_all[kBitsWaves][kBitsWaveWords];
    __shared__ uint32_t gate; // waves of the first half that have their first region's text
    for (uint32_t i = threadIdx.x; i < kBitsTabEntries; i += blockDim.x) tab[i] = reinterpret_cast<const uint4 *>(T.bits_tab)[i];
    if (threadIdx.x == 0) gate = 0u;
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t lane = lane_id();
    uint32_t *img = img_all[wave];
    BitsCtx c;
    c.hay = L.d_hay;
    c.dfa = reinterpret_cast<const uint32_t *>(T.dfa);
    c.n_units = L.n_units;
    c.n_cls = T.n_cls;
    c.base = T.cls_base;
    c.span = T.cls_span;
    const uint32_t base2 = c.base * 0x10001u;
    const uint16_t *hay = L.d_hay;
    const uint32_t nu = L.n_units;
    // A wave alternates between streaming text (memory) and following chains (instruction issue, LDS).  The second half of
    // the workgroup starts when the first half has its text: from then on one half streams while the other walks.
    bool first = true;
#ifdef ACGPU_TIMING
    unsigned long long bt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt0 = __builtin_amdgcn_s_memtime();
    const unsigned long long bstart = bt0;
#endif
    if (wave >= kBitsWaves / 2) {
#ifdef ACGPU_ABLATION
        if (!(L.debug & 16u))
#endif
        while (__hip_atomic_load(&gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)(kBitsWaves / 2)) __builtin_amdgcn_s_sleep(64);
    }

    BITS_MARK(0)
    // The records of a finished region, straight to their final place.  They begin behind the matches of all regions before
    // it: the regions before it in its block of 64 (one load of up to 63 words {published, count}) and the blocks before its
    // own (one load per 64 blocks {regions published, their matches}, a block complete when all 64 have published); nothing
    // but these words is handed over, so relaxed device-scope accesses suffice.  Every position of the chain is the start of a
    // match and the NEXT position of the chain is its end.  Two segments per step (a word of marks per lane, text order =
    // lane order): ranks from one wave prefix sum, the starts -- 16 bits, relative to the step -- into a staging list in LDS
    // (the image is free: the region's marks come back from memory into registers), then the records {start k, start k + 1}
    // as whole 16-byte stores, two per lane, 1 KB contiguous per instruction.  (A lane writing the records of its own word
    // straight to memory -- 64 lanes, 64 places, 8 bytes each -- cost 250 k cycles per region and held up every other
    // wave's memory operations: such stores saturate the store PATH.)  Behind a step's last start comes where the chain left
    // the segment (the exit of the lane that walked it).
#ifdef ACGPU_TIMING
    unsigned long long t_lookback_done = 0;
#endif
    auto emit_region = [&](uint32_t r) {
        const BitsCtx &bc = c; // (a step's count is called c below)
        unsigned long long before = 0;
        {
            const uint32_t blk = r >> 6, in_blk = r & 63u;
#ifdef ACGPU_ABLATION
            if (!(L.debug & 32u)) // 32: no look-back (every region writes from record 0: timing only)
#endif
            {
                if (in_blk) {
                    unsigned long long v;
                    do {
                        v = lane < in_blk ? __hip_atomic_load(&L.d_agg[(blk << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kBitsPub;
                        if (__all((v & kBitsPub) != 0ull)) break;
                        __builtin_amdgcn_s_sleep(8);
                    } while (true);
                    before += bits_wave_sum64(v & kBitsCount);
                }
                for (uint32_t b0 = 0; b0 < blk; b0 += 64u) { // (every block before it has 64 regions)
                    unsigned long long v;
                    do {
                        v = b0 + lane < blk ? __hip_atomic_load(&L.d_blk[b0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (64ull << 40);
                        if (__all((v >> 40) == 64ull)) break;
                        __builtin_amdgcn_s_sleep(8);
                    } while (true);
                    before += bits_wave_sum64(v & kBitsCount);
                }
            }
        }
#ifdef ACGPU_ABLATION
        if (L.debug & 4u) return; // 4: no record stores
#endif
#ifdef ACGPU_TIMING
        t_lookback_done = __builtin_amdgcn_s_memtime();
#endif
        struct __attribute__((packed, aligned(4))) Rec2 { uint32_t s0, e0, s1, e1; };
        int2 *outp = reinterpret_cast<int2 *>(L.d_out);
        uint32_t *stg = img; // the starts of a step (at most 2048) and, twice, what comes behind them
        static_assert(2u * kBitsSegUnits + 2u <= kBitsWaveWords, "a step's starts fit the image");
        const uint32_t *gm = L.d_marks + (size_t)r * kBitsRegionWords;
        const uint32_t xo = L.d_xout[(size_t)r * kBitsSegs + lane];
        const uint32_t R0 = L.g0 + r * kBitsRegionUnits;
        const uint32_t rel = R0 + (lane >> 5) * kBitsSegUnits + (lane & 31u) * 32u; // the lane's word in step 0
        unsigned long long done = before;
        // the marks of kCh steps at a time (MAP: and the text words under them and one word on, for the keywords' bits)
        constexpr uint32_t kCh = MAP ? 8u : 16u;
        uint32_t mk[kCh], txa[MAP ? kCh : 1u], txb[MAP ? kCh : 1u];
        const uint32_t *tx = MAP ? L.d_text + (size_t)r * kBitsTextWords : nullptr;
#pragma unroll
        for (uint32_t half = 0; half < 32u / kCh; ++half) {
#pragma unroll
            for (uint32_t t = 0; t < kCh; ++t) mk[t] = gm[(half * kCh + t) * 64u + lane];
            if (MAP) {
#pragma unroll
                for (uint32_t t = 0; t < kCh; ++t) {
                    txa[t] = tx[(half * kCh + t) * 64u + lane];
                    txb[t] = tx[(half * kCh + t) * 64u + lane + 1u];
                }
            }
#pragma unroll
            for (uint32_t t = 0; t < kCh; ++t) {
                const uint32_t i = half * kCh + t; // segments 2i and 2i + 1
                uint32_t w = mk[t];
                const uint32_t c = (uint32_t)__popc(w);
                const uint32_t incl = bits_wave_scan(c);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (tot == 0u) continue; // wave-uniform
                const uint32_t in_second = tot - (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
                const uint32_t edge = in_second ? (uint32_t)__builtin_amdgcn_readlane((int)xo, (int)(2u * i + 1u))
                                                : (uint32_t)__builtin_amdgcn_readlane((int)xo, (int)(2u * i));
                const uint32_t pb = rel + i * 2u * kBitsSegUnits;
                uint32_t *at = stg + (incl - c);
                while (w) {
                    *at++ = pb + (uint32_t)__builtin_ctz(w);
                    w &= w - 1u;
                }
                if (lane == 0) {
                    stg[tot] = edge;
                    stg[tot + 1u] = edge;
                }
                __builtin_amdgcn_wave_barrier();
                if (MAP) {
                    // A record per lane and round.  Its keyword's id: the keyword's own bits -- out of the step's text words, which
                    // the lanes hold (a shuffle, no memory access) -- looked up in HostTables::bits_idkeys (one 16-byte gather; the
                    // text's few hundred frequent keywords stay in the L1).  The round's 64 records are then written as ONE stream
                    // of dwords through LDS: whole aligned 16-byte pieces, 16 per lane -- lanes storing their own 12 bytes each
                    // cost five times the Set records' pairs (measured: 0.26 against 0.055 ms for the 84 M records of config 4).
                    int32_t *out32 = reinterpret_cast<int32_t *>(L.d_out);
                    uint32_t *spare = img + 2u * kBitsSegUnits + 4u; // (behind the starts: 192 words)
                    static_assert(2u * kBitsSegUnits + 4u + 3u * kWave <= kBitsWaveWords, "a round's records fit the image behind the starts");
                    typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
                    const v2ul *slots = reinterpret_cast<const v2ul *>(T.bits_idkeys);
                    constexpr uint32_t kRounds = 1; // rounds whose lookups are in flight together (2: measured slower, 1.40 against 1.13 ms -- the kernel sits at its register limit)
                    for (uint32_t k0 = 0; k0 < tot; k0 += kRounds * kWave) {
                        uint32_t s0[kRounds], e0[kRounds], id[kRounds], slot[kRounds];
                        unsigned long long key[kRounds];
                        v2ul kk[kRounds];
#pragma unroll
                        for (uint32_t q = 0; q < kRounds; ++q) {
                            const uint32_t k = min(k0 + q * kWave + lane, tot - 1u); // (a lane beyond the list repeats the last record)
                            s0[q] = stg[k];
                            e0[q] = stg[k + 1u];
                            const uint32_t len = e0[q] - s0[q], rl = s0[q] - R0;
                            const uint32_t j = (rl >> 5) - i * 64u; // the lane that holds the start's text word
                            const uint32_t t0 = (uint32_t)__shfl((int)txa[t], (int)j), t1 = (uint32_t)__shfl((int)txb[t], (int)j);
                            const uint32_t bits = __builtin_amdgcn_alignbit(t1, t0, rl & 31u) & (len < 32u ? (1u << (len & 31u)) - 1u : ~0u);
                            key[q] = ((unsigned long long)len << 32) | bits;
                            slot[q] = bits_id_hash(key[q]) & T.bits_idmask;
                            kk[q] = slots[slot[q]];
                        }
#pragma unroll
                        for (uint32_t q = 0; q < kRounds; ++q) {
                            const uint32_t len = e0[q] - s0[q];
                            id[q] = ~0u;
#ifdef ACGPU_ABLATION
                            if (L.debug & 64u) continue; // 64: no id lookups (timing only)
#endif
                            if (__builtin_expect(len <= 32u, 1)) {
                                for (uint32_t probe = 0; probe < 64u; ++probe) { // (a call that bails out -- units outside the alphabet -- may ask for what is no keyword)
                                    if (kk[q].x == key[q]) {
                                        id[q] = (uint32_t)kk[q].y;
                                        break;
                                    }
                                    if (kk[q].x == kEmptyKey) break;
                                    slot[q] = (slot[q] + 1u) & T.bits_idmask;
                                    kk[q] = slots[slot[q]];
                                }
                            } else { // a keyword of more than 32 units: its node by the walk
                                uint32_t node = 0;
                                for (uint32_t u = s0[q]; u < e0[q] && u < nu; ++u) {
                                    const uint32_t dlt = (uint32_t)hay[u] - bc.base;
                                    const uint32_t g = bc.dfa[(uint64_t)node * bc.n_cls + (dlt < bc.span ? dlt + 1u : 0u)];
                                    if (!g) break;
                                    node = g & 0x7fffffffu;
                                }
                                id[q] = T.term_id[node];
                            }
                        }
#pragma unroll
                        for (uint32_t q = 0; q < kRounds; ++q) {
                            const uint32_t kq = k0 + q * kWave;
                            if (kq >= tot) break; // wave-uniform
                            const uint32_t n = min(tot - kq, (uint32_t)kWave);
                            if (lane < n) {
                                spare[3u * lane] = s0[q];
                                spare[3u * lane + 1u] = e0[q];
                                spare[3u * lane + 2u] = id[q];
                            }
                            __builtin_amdgcn_wave_barrier();
                            const unsigned long long first = done + kq; // the round's first record
                            if (__builtin_expect(first + n <= L.cap, 1)) { // wave-uniform: all of them fit
                                const unsigned long long d0 = first * 3ull;
                                // (dwords in front of the first 16-byte boundary of the output: whatever alignment the caller's buffer has)
                                const uint32_t nd = 3u * n, head = min(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(out32 + d0) & 15u)) & 15u) / 4u, nd), body = (nd - head) & ~3u;
                                if (lane < head) out32[d0 + lane] = (int32_t)spare[lane];
                                if (4u * lane < body) {
                                    const uint32_t *sp = spare + head + 4u * lane;
                                    *reinterpret_cast<uint4 *>(out32 + d0 + head + 4u * lane) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
                                }
                                if (lane < nd - head - body) out32[d0 + head + body + lane] = (int32_t)spare[head + body + lane];
                            } else if (lane < n && first + lane < L.cap) {
                                out32[(first + lane) * 3ull] = (int32_t)s0[q];
                                out32[(first + lane) * 3ull + 1ull] = (int32_t)e0[q];
                                out32[(first + lane) * 3ull + 2ull] = (int32_t)id[q];
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                } else
                if (__builtin_expect(done + tot <= L.cap, 1)) { // wave-uniform: all of them fit
                    // pairs of records on 16-byte boundaries of the output; a single one in front and / or behind them
                    const uint32_t odd = (uint32_t)done & 1u;
                    const uint32_t pairs_end = odd + ((tot - odd) & ~1u);
                    for (uint32_t k = odd + 2u * lane; k < pairs_end; k += 2u * kWave) {
                        const uint32_t s0 = stg[k], s1 = stg[k + 1u], s2 = stg[k + 2u];
                        const Rec2 two = {s0, s1, s1, s2};
                        *reinterpret_cast<Rec2 *>(outp + (done + k)) = two;
                    }
                    if (lane == 0 && odd) outp[done] = make_int2((int)stg[0], (int)stg[1]);
                    if (lane == 1 && pairs_end < tot) outp[done + pairs_end] = make_int2((int)stg[pairs_end], (int)stg[pairs_end + 1u]);
                } else {
                    for (uint32_t k = lane; k < tot; k += kWave)
                        if (done + k < L.cap) outp[done + k] = make_int2((int)stg[k], (int)stg[k + 1u]);
                }
                __builtin_amdgcn_wave_barrier();
                done += tot;
            }
        }
        if (r + 1u == L.n_regions && lane == 0) L.d_exit[2] = done; // the call's record count
    };
    bool have_prev = false;
    uint32_t prev_r = 0;
    // Regions are handed out in the order in which waves ask for them (one atomic per region): a region's records go behind
    // those of every region before it, and every region before it belongs to a wave that is already running -- the look-back
    // can wait for them whatever share of the grid is resident.
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(L.d_next, 1u);
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= L.n_regions) break;
        const uint32_t R0 = L.g0 + r * kBitsRegionUnits; // first position of segment 0
        // ---- the region's text, one bit per unit: segments -1 .. 63 and the first words of segment 64 -----------------------
        uint32_t dmax = 0;
        // lane's 32 units from position u0 on; units behind the end of the buffer read as the first letter
        auto load_careful = [&](uint32_t (&w)[16], uint32_t u0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t pos = u0 + 8u * v;
                if ((uint64_t)pos + 8u <= nu) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(hay + pos);
                    w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t a = pos + 2u * k < nu ? (uint32_t)hay[pos + 2u * k] : c.base;
                        const uint32_t b = pos + 2u * k + 1u < nu ? (uint32_t)hay[pos + 2u * k + 1u] : c.base;
                        w[4 * v + k] = a | (b << 16);
                    }
                }
            }
        };
        auto load_fast = [&](uint32_t (&w)[16], uint32_t u0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(hay + u0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint4 x = src[v];
                w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
            }
        };
        // word W of the region (W = -32 .. 2050) -> its place(s) in the image
        auto put = [&](int32_t W, uint32_t word) {
            const int32_t sg = W >> 5;
            const uint32_t k = (uint32_t)W & 31u;
            if (sg < (int32_t)kBitsSegs) img[(uint32_t)(sg + 1) * kBitsStride + k] = word;
            if (k < kBitsLook && sg >= 0) img[(uint32_t)sg * kBitsStride + kBitsSegWords + k] = word; // (behind segment sg - 1)
        };
        // the words around the region in ONE more tile: lanes 32-63 the 1024 positions before it (segment -1), lanes 0-2 the
        // first words of the next region; the other lanes read their words of tile 0 once more and drop them
        const bool aux_lane = (r > 0 && lane >= 32) || lane < kBitsLook;
        const int32_t aux_W = lane >= 32 ? (int32_t)lane - 64 : (int32_t)(kBitsRegionUnits / 32u + lane);
        const uint32_t aux_pos = aux_lane ? R0 + (uint32_t)(aux_W * 32) : R0 + lane * 32u;
        if ((uint64_t)R0 + kBitsRegionUnits + kBitsLook * 32u <= nu) { // every load inside the buffer: the stream, two tiles ahead
            // 33 tiles (the one above, then the region's 32) through three register sets: 8 KB per wave in flight, so that the
            // half of the workgroup that streams keeps the memory busy on its own
            uint32_t wa[16], wb[16], wc[16];
            auto tile_pos = [&](int32_t t) { return t < 0 ? aux_pos : R0 + (uint32_t)t * kBitsTileUnits + lane * 32u; };
            auto tile_put = [&](int32_t t, uint32_t word) {
                if (t < 0) {
                    if (aux_lane) put(aux_W, word);
                } else put((int32_t)((uint32_t)t * 64u + lane), word);
            };
            constexpr int32_t kTiles = (int32_t)(kBitsRegionUnits / kBitsTileUnits);
            static_assert((kTiles + 1) % 3 == 0, "three register sets in turn");
            load_fast(wa, tile_pos(-1));
            load_fast(wb, tile_pos(0));
#pragma unroll 1
            for (int32_t t0 = -1; t0 < kTiles; t0 += 3) {
                load_fast(wc, tile_pos(t0 + 2));
                tile_put(t0, bits_pack(wa, base2, dmax));
                load_fast(wa, tile_pos(min(t0 + 3, kTiles - 1))); // (unconditional: the wait counts stay exact; the last tile's
                tile_put(t0 + 1, bits_pack(wb, base2, dmax));     // words are read twice more and dropped)
                load_fast(wb, tile_pos(min(t0 + 4, kTiles - 1)));
                tile_put(t0 + 2, bits_pack(wc, base2, dmax));
            }
        } else {
            uint32_t w[16];
            if (aux_lane) {
                load_careful(w, aux_pos);
                put(aux_W, bits_pack(w, base2, dmax));
            }
#pragma unroll 1
            for (uint32_t tl = 0; tl < kBitsRegionUnits / kBitsTileUnits; ++tl) {
                load_careful(w, R0 + tl * kBitsTileUnits + lane * 32u);
                put((int32_t)(tl * 64u + lane), bits_pack(w, base2, dmax));
            }
        }
        const bool foreign = (dmax & 0xffffu) >= c.span || (dmax >> 16) >= c.span;
        if (__any(foreign) && lane == 0) L.d_exit[1] = 2ull; // a unit outside the alphabet: not this kernel's text
        __builtin_amdgcn_wave_barrier();
#ifdef ACGPU_ABLATION
        if (!(L.debug & 128u)) // 128: the text is not parked (timing only)
#endif
        if (MAP) { // the region's text bits (and the three words behind it) to memory: pass 2 writes its marks in their place
            uint32_t *tx = L.d_text + (size_t)r * kBitsTextWords;
#pragma unroll 4
            for (uint32_t i = 0; i < kBitsRegionWords / 256u; ++i) {
                const uint32_t g = i * 64u + lane, sg = g >> 3, k = (g & 7u) * 4u;
                const uint32_t *sp = img + (sg + 1u) * kBitsStride + k;
                *reinterpret_cast<uint4 *>(tx + 4u * g) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
            }
            if (lane < kBitsLook) tx[kBitsRegionWords + lane] = img[kBitsSegs * kBitsStride + kBitsSegWords + lane];
            if (lane == kBitsLook) tx[kBitsRegionWords + lane] = 0u;
        }
        if (first && wave < kBitsWaves / 2 && lane == 0) atomicAdd(&gate, 1u);
        first = false;
        BITS_MARK(1)

        // ---- pass 1: where the chain enters the lane's segment --------------------------------------------------------------
        const uint32_t start = R0 + lane * kBitsSegUnits;
        uint32_t e_in;
        if (r == 0 && lane == 0) {
            e_in = L.entry;
        } else {
            const uint32_t *seg = img + lane * kBitsStride; // segment lane - 1
            const uint32_t ps = start - kBitsSegUnits;
            uint32_t p0 = max(ps, L.entry);
            const uint32_t lim = min(start, L.own_end);
            if (lim > L.runup) p0 = max(p0, lim - L.runup); // (the run-up: the last L.runup positions of the segment)
            uint32_t q = p0 - ps;
            const uint32_t qlim = lim > ps ? lim - ps : 0u;
            const uint32_t qsafe = nu - ps; // (ps < nu: the segment before one that begins inside the owned range)
#ifdef ACGPU_ABLATION
            if (L.debug & 1u) q = max(q, qlim); // no pass 1 (wrong entries: timing only)
#endif
            uint32_t none = 0;
            e_in = ps + bits_walk<false>(c, tab, const_cast<uint32_t *>(seg), q, qlim, ps, qsafe, none);
        }
        __builtin_amdgcn_wave_barrier();

        BITS_MARK(2)
        // ---- pass 2: the lane's own segment from there: count, and mark the starts in place of the text left behind ---------
        uint32_t *seg = img + (lane + 1u) * kBitsStride;
        const uint32_t bound = min(start + kBitsSegUnits, L.own_end);
        const uint32_t qend = bound > start ? bound - start : 0u;
        const uint32_t qsafe = nu > start ? nu - start : 0u;
        uint32_t q = e_in - start; // (an entry beyond the segment, or a segment beyond the owned range: no steps)
        uint32_t cnt = 0;
#ifdef ACGPU_ABLATION
        if (L.debug & 2u) q = max(q, qend); // no pass 2
#endif
        q = bits_walk<true>(c, tab, seg, q, qend, start, qsafe, cnt);
        const uint32_t x_out = start + q; // the chain's first position at or behind min(segment end, own_end)
        // every exit is the next lane's entry (boundaries inside the owned range)
        const uint32_t e_next = __shfl_down(e_in, 1);
        const bool differs = lane < 63u && (uint64_t)start + kBitsSegUnits < L.own_end && x_out != e_next;
#ifdef ACGPU_ABLATION
        if (L.debug) {
            const uint64_t bd = __ballot(differs);
            if (bd && lane == 0) atomicAdd(&L.d_exit[3], (unsigned long long)__popcll(bd));
        } else
#endif
        if (__any(differs) && lane == 0) atomicMax(&L.d_exit[1], 1ull);
        if (lane == 0) L.d_pred[r] = e_in;
        if (lane == 63) L.d_true[r] = x_out;
        if (start < L.own_end && bound == L.own_end) L.d_exit[0] = (unsigned long long)x_out; // the segment with the last owned position: its exit is the call's
        __builtin_amdgcn_wave_barrier();

        BITS_MARK(3)
        // ---- out ------------------------------------------------------------------------------------------------------------
        // The region's count is published at once; its marks and exits go to memory (8.3 KB), and its RECORDS are written one
        // region later (emit_region below): where they begin is the number of matches of ALL regions before it, and the 4096
        // waves of a round finish their walks at about the same time -- a wave that waited for the slowest of them here stood
        // still for a third of its time (measured), one that comes back after its next region finds every count there.
        const uint32_t region_total = bits_wave_sum(cnt);
        if (lane == 0) {
            __hip_atomic_store(&L.d_agg[r], kBitsPub | region_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&L.d_blk[r >> 6], (1ull << 40) | region_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        {
            uint32_t *gm = L.d_marks + (size_t)r * kBitsRegionWords;
#pragma unroll 4
            for (uint32_t i = 0; i < kBitsRegionWords / 256u; ++i) { // four words per lane: 16-byte stores
                const uint32_t g = i * 64u + lane, sg = g >> 3, k = (g & 7u) * 4u;
                const uint32_t *sp = img + (sg + 1u) * kBitsStride + k;
                *reinterpret_cast<uint4 *>(gm + 4u * g) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
            }
            L.d_xout[(size_t)r * kBitsSegs + lane] = x_out;
        }
        __builtin_amdgcn_wave_barrier();
        BITS_MARK(4)
        if (have_prev) emit_region(prev_r);
        prev_r = r;
        have_prev = true;
        __builtin_amdgcn_wave_barrier();
#ifdef ACGPU_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the stores are this phase's)
#endif
        BITS_MARK(5)
    }
#ifdef ACGPU_TIMING
    const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif
    if (have_prev) emit_region(prev_r); // (the wave's last region: here it does wait for the regions before it)
#ifdef ACGPU_TIMING
    if (have_prev && lane == 0) {
        atomicAdd(&g_bits_tail[0], t_lookback_done - t_loop_end);
        atomicAdd(&g_bits_tail[1], __builtin_amdgcn_s_memtime() - t_lookback_done);
    }
#endif
#ifdef ACGPU_TIMING
    if (lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_bits_timing[i], bt[i]);
        atomicAdd(&g_bits_timing[6], __builtin_amdgcn_s_memtime() - bstart);
        atomicAdd(&g_bits_timing[7], 1ull);
    }
#endif
    if (first && wave < kBitsWaves / 2 && lane == 0) atomicAdd(&gate, 1u); // (a wave without a region)
}

// The call's last kernel, one workgroup: every region's entry must be the exit of the region before it (else the bail flag);
// then {count, bail flag, exit} to the call's pinned host slot and to acgpu_shard::d_result, in stream order; and the call's
// state words -- look-back words, region counter, exit / flag / count -- are zeroed for the NEXT call (no memset operations
// of its own on the stream: each is a launch gap).
__global__ __launch_bounds__(1024) void k_longest_bits_finish(LongestBitsLaunch L, unsigned long long *h_slot, acgpu_device_result *res,
                                                              unsigned long long *state, uint32_t state_words) {
    int seam = 0;
    for (uint32_t r = 1u + threadIdx.x; r < L.n_regions; r += blockDim.x) seam |= L.d_pred[r] != L.d_true[r - 1u] ? 1 : 0;
    const int any_seam = __syncthreads_or(seam);
    if (threadIdx.x == 0) {
        unsigned long long flag = L.d_exit[1];
        if (any_seam && flag == 0ull) flag = 1ull;
        const unsigned long long n = L.d_exit[2], ex = L.d_exit[0];
        h_slot[1] = flag;
        h_slot[2] = ex;
        h_slot[0] = n;
        if (res) {
            res->n_records = n;
            res->redone = flag != 0ull; // (the records are not there yet: acgpu_match_device_end redoes the call)
            res->reserved = 0;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < state_words; i += blockDim.x) state[i] = 0ull;
}

uint32_t longest_bits_region_units() { return kBitsRegionUnits; }
uint32_t longest_bits_seg_units() { return kBitsSegUnits; }
size_t longest_bits_region_scratch_bytes() { return (size_t)kBitsRegionWords * 4 + kBitsSegs * 4; }
size_t longest_bits_region_text_bytes() { return (size_t)kBitsTextWords * 4; }

hipError_t launch_longest_bits(const DevTables &t, const LongestBitsLaunch &l, unsigned long long *h_slot_dev, acgpu_device_result *d_result,
                               unsigned long long *d_state, uint32_t state_words, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_mid,
                               hipEvent_t ev_stop) {
    if (l.d_text) ACGPU_LAUNCH_EV(k_longest_bits<true>, dim3(l.grid), dim3(kBitsBlock), 0, stream, ev_start, ev_mid, t, l);
    else ACGPU_LAUNCH_EV(k_longest_bits<false>, dim3(l.grid), dim3(kBitsBlock), 0, stream, ev_start, ev_mid, t, l);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
#ifdef ACGPU_TIMING
    {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[8] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bits_timing), sizeof(h));
        if (h[7]) fprintf(stderr, "[bits timing] waves %llu: total %.0f | gate %.0f | text %.0f | pass 1 %.0f | pass 2 %.0f | publish + marks %.0f | records of the region before %.0f (s_memtime ticks per wave)\n",
                          h[7], (double)h[6] / h[7], (double)h[0] / h[7], (double)h[1] / h[7], (double)h[2] / h[7], (double)h[3] / h[7], (double)h[4] / h[7], (double)h[5] / h[7]);
        unsigned long long tl[2] = {0, 0};
        (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_bits_tail), sizeof(tl));
        if (h[7]) fprintf(stderr, "[bits timing] last region of a wave: waits %.0f for the counts before it, writes its records in %.0f\n", (double)tl[0] / h[7], (double)tl[1] / h[7]);
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bits_timing), z, sizeof(z));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bits_tail), z, sizeof(tl));
    }
#endif
    const hipEvent_t ev_none = nullptr;
    ACGPU_LAUNCH_EV(k_longest_bits_finish, dim3(1), dim3(1024), 0, stream, ev_none, ev_stop, l, h_slot_dev, d_result, d_state, state_words);
    return hipGetLastError();
}

} // namespace acgpu
