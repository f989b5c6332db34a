// acgpu_longest_follow.hip -- LongestMatchSet/Map over any dense dictionary: the walks of the chain's own positions only (gfx950).
//
// Replaces, for long texts, the first three stages of the walk pipeline of acgpu_longest.hip (S/LongestMatchSet.java:192-265,
// S/SetMatchQueue.java:45-95): that pipeline walks the keyword trie from EVERY position (k_longest_walk_list: the length array),
// finds a synchronisation point per tile (k_longest_sync) and then follows the greedy chain pos -> pos + max(L[pos], 1) through
// the lengths (k_longest_chain_lds).  The chain visits one position in four or five of a natural text, and over dictionaries of
// natural words (235 886 words: 1.46 M trie nodes, a 345 MB table) every walk is a handful of dependent gathers into a table far
// beyond the caches: the walks ARE the cost.  Here only the chain's positions are walked:
//  * a lane owns a segment of 1024 positions; pass 1 follows a chain through the last `runup` positions of the segment before
//    it -- chains that start at different positions merge (at the latest where no keyword goes on: a space), so where it leaves
//    that segment is the true chain's entry into the lane's own --, pass 2 follows the lane's own segment from there, counts
//    the matches and marks their starts and ends (bit end - 1) in the two bitmaps k_longest_emit_ends reads; then every lane's
//    exit is compared with its neighbour's entry (the first entry is the call's: equal everywhere = exact everywhere), the
//    seams between regions by k_longest_follow_check; a difference raises the bail flag and the walk pipeline redoes the call;
//  * the walks do not run in lock step: every iteration of a wave is ONE trie transition of every lane, wherever its chain is --
//    a lane whose walk ends starts the next one in the same iteration (a lock-step walk lasts as long as the deepest of 64);
//  * the text comes through a ring of four 8-unit blocks per lane in LDS, one block ahead of the walk; the first rows of the
//    table (root, the nodes under it: half of all transitions) sit in LDS, class pages behind them for table classes.
// 64 registers, two workgroups of 16 waves per CU: 2048 independent chains per CU hide the gathers' latency.
// Bound by the rate of random gathers into the table, not by the text stream: 2 B per unit are read once (by 16-byte pieces).
#include <algorithm>

#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

constexpr int kFolBlock = 1024;                 // 16 waves; two workgroups per CU
constexpr uint32_t kFolSegUnits = 1024;         // a lane's segment (LongestFollowLaunch::seg_log2 = 10; 9 for texts that would not fill the chip)
constexpr uint32_t kFolRingWords = 4 * 64 * 4;  // per wave: [4 blocks][64 lanes] of 16 bytes
constexpr uint32_t kFolFlushEvery = 8;          // iterations between two flushes of the finished bitmap words
constexpr uint32_t kFolRowBytesMax = 76 * 1024 - (kFolBlock / kWave) * kFolRingWords * 4; // LDS left for rows and pages: 12 KiB

struct __attribute__((packed, aligned(2))) FolUnits8 { // 8 UTF-16 units at a unit address that is a multiple of 8 (16-byte aligned buffers)
    uint32_t d[4];
};

// one chain through [p, plim): returns the chain's first position at or behind plim.  MARK: matches are counted and marked.
template <bool RANGE, bool STATE, bool MARK>
__device__ __forceinline__ uint32_t fol_walk(const DevTables &T, const LongestFollowLaunch &L, const uint32_t *rows, const uint16_t *pages,
                                             uint4 *ring, uint32_t lane, uint32_t p, uint32_t plim, uint32_t seg_first_word, uint32_t seg_end_word,
                                             uint32_t &cnt) {
    const uint32_t *dfa = reinterpret_cast<const uint32_t *>(T.dfa);
    const uint16_t *hay = L.d_hay;
    const uint32_t nu = L.n_units, n_cls = T.n_cls, hot = L.hot_rows;
    uint32_t d = 0, node = 0, best = 0, bnode = 0;
    uint32_t have_end = p >> 3;              // blocks [have_end - 4, have_end) are in the ring (none yet)
    uint32_t sw = 0, swi = seg_first_word;   // MARK: the word of start bits being collected, its index; words below it are written
    uint32_t ew = 0, ewi = seg_first_word;   // the same for the end bits (merged with atomicOr: an end may lie in another lane's word)
    // MARK: a finished bitmap word waits here for the next flush -- all lanes store together every kFolFlushEvery iterations: a
    // store in the body would be waited for by the next gather (loads and stores share one counter, and a store's
    // acknowledgement takes longer than a gather), and with 64 chains at 64 different places some lane stores in every iteration
    uint32_t pend_sw = 0, pend_swi = ~0u, pend_ew = 0, pend_ewi = ~0u, it = 0;
    uint32_t sp0 = 0, sb0 = 0, sp1 = 0, sb1 = 0, sp2 = 0, sb2 = 0, sp3 = 0, sb3 = 0, n_sp = 0; // STATE: {position, node} of the matches since the last flush
    bool active = p < plim;
    while (__any(active)) {
        const uint32_t x = p + d, xb = x >> 3;
        // the block after the one the walk is in goes out now; it lands behind this iteration's gather
        const bool want_load = active && have_end <= xb + 1u && have_end * 8u < nu;
        const uint32_t load_b = have_end;
        uint4 blk = make_uint4(0u, 0u, 0u, 0u);
        if (want_load) {
            const uint32_t b0 = have_end * 8u;
            if (b0 + 8u <= nu) {
                const FolUnits8 v = *reinterpret_cast<const FolUnits8 *>(hay + b0);
                blk = make_uint4(v.d[0], v.d[1], v.d[2], v.d[3]);
            } else { // the buffer's last, partial block
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (uint32_t k = 0; k < 8u && b0 + k < nu; ++k) w[k >> 1] |= (uint32_t)hay[b0 + k] << (16u * (k & 1u));
                blk = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        const bool can_step = active && (xb < have_end || x >= nu);
        bool keep = true; // the block that is on its way goes into the ring
        uint32_t e = 0;
        if (can_step && x < nu) {
            const uint32_t u = reinterpret_cast<const uint16_t *>(ring + (xb & 3u) * 64u + lane)[x & 7u];
            uint32_t cls;
            if (RANGE) {
                const uint32_t dlt = u - T.cls_base;
                cls = dlt < T.cls_span ? dlt + 1u : 0u;
            } else {
                cls = pages[128u + ((uint32_t)reinterpret_cast<const unsigned char *>(pages)[u >> 8] << 8) + (u & 255u)];
            }
            const uint32_t idx = node * n_cls + cls;
            e = node < hot ? rows[idx] : dfa[idx]; // (class 0: no edge in any row)
        }
        if (can_step) {
            if (e == 0u) { // the walk ends: the chain's next position
                if (MARK) {
                    if (best) {
                        ++cnt;
                        const uint32_t wi = p >> 5;
                        if (wi != swi) {
                            if (pend_swi != ~0u) L.d_bits[pend_swi] = pend_sw; // (rare: two words within one flush interval)
                            pend_sw = sw;
                            pend_swi = swi;
                            for (uint32_t z = swi + 1u; z < wi; ++z) L.d_bits[z] = 0u;
                            sw = 0u;
                            swi = wi;
                        }
                        sw |= 1u << (p & 31u);
                        const uint32_t q = p + best - 1u, qi = q >> 5;
                        if (qi != ewi) {
                            if (ew) {
                                if (pend_ewi != ~0u) atomicOr(&L.d_ebits[pend_ewi], pend_ew);
                                pend_ew = ew;
                                pend_ewi = ewi;
                            }
                            ew = 0u;
                            ewi = qi;
                        }
                        ew |= 1u << (q & 31u);
                        if (STATE) { // (Map records: the matched node, stored with the next flush like the bitmap words -- four waiting at most)
                            if (n_sp == 4u) { // (cannot happen between two flushes of 8 iterations: a match takes two at least)
                                L.d_state[sp0] = sb0;
                                sp0 = sp1; sb0 = sb1; sp1 = sp2; sb1 = sb2; sp2 = sp3; sb2 = sb3;
                                n_sp = 3u;
                            }
                            sp0 = n_sp == 0u ? p : sp0; sb0 = n_sp == 0u ? bnode : sb0;
                            sp1 = n_sp == 1u ? p : sp1; sb1 = n_sp == 1u ? bnode : sb1;
                            sp2 = n_sp == 2u ? p : sp2; sb2 = n_sp == 2u ? bnode : sb2;
                            sp3 = n_sp == 3u ? p : sp3; sb3 = n_sp == 3u ? bnode : sb3;
                            ++n_sp;
                        }
                    }
                }
                p += best ? best : 1u;
                d = 0u;
                node = 0u;
                best = 0u;
                bnode = 0u;
                // the ring holds blocks [have_end - 4, have_end): a walk of more than 24 units has pushed the chain's next
                // position out of it, a match of many units has jumped beyond it -- the blocks are asked for again from there
                if ((p >> 3) + 4u < have_end || (p >> 3) > have_end) have_end = p >> 3;
                else if ((p >> 3) + 4u == have_end) keep = false; // (it would take the slot of the block the chain goes on in)
                active = p < plim;
            } else {
                node = e & 0x7fffffffu;
                ++d;
                if (e >> 31) {
                    best = d;
                    bnode = node;
                }
            }
        }
        if (MARK && (++it & (kFolFlushEvery - 1u)) == 0u) {
            if (pend_swi != ~0u) L.d_bits[pend_swi] = pend_sw;
            if (pend_ewi != ~0u) atomicOr(&L.d_ebits[pend_ewi], pend_ew);
            pend_swi = ~0u;
            pend_ewi = ~0u;
            if (STATE) {
                if (n_sp > 0u) L.d_state[sp0] = sb0;
                if (n_sp > 1u) L.d_state[sp1] = sb1;
                if (n_sp > 2u) L.d_state[sp2] = sb2;
                if (n_sp > 3u) L.d_state[sp3] = sb3;
                n_sp = 0u;
            }
        }
        if (want_load && keep && have_end == load_b) { // (behind the gather's wait: the block has arrived with it; not if the ring has just started over)
            ring[(load_b & 3u) * 64u + lane] = blk;
            have_end = load_b + 1u;
        }
    }
    if (MARK) {
        if (STATE) {
            if (n_sp > 0u) L.d_state[sp0] = sb0;
            if (n_sp > 1u) L.d_state[sp1] = sb1;
            if (n_sp > 2u) L.d_state[sp2] = sb2;
            if (n_sp > 3u) L.d_state[sp3] = sb3;
        }
        if (pend_swi != ~0u) L.d_bits[pend_swi] = pend_sw;
        if (pend_ewi != ~0u) atomicOr(&L.d_ebits[pend_ewi], pend_ew);
        if (swi < seg_end_word) {
            L.d_bits[swi] = sw;
            for (uint32_t z = swi + 1u; z < seg_end_word; ++z) L.d_bits[z] = 0u;
        }
        if (ew) atomicOr(&L.d_ebits[ewi], ew);
    }
    return p;
}

// (Map records keep four {position, node} pairs waiting for the next flush: that form takes the registers of one workgroup per CU)
template <bool RANGE, bool STATE>
__global__ __launch_bounds__(kFolBlock, STATE ? 4 : 8) void k_longest_follow(DevTables T, LongestFollowLaunch L) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // [hot rows][class pages (table classes)][rings]
    const uint32_t row_words = L.hot_rows * T.n_cls, page_words = RANGE ? 0u : (T.dfa_pages_bytes + 3u) / 4u;
    for (uint32_t i = threadIdx.x; i < row_words; i += blockDim.x) smem[i] = reinterpret_cast<const uint32_t *>(T.dfa)[i];
    for (uint32_t i = threadIdx.x; i < page_words; i += blockDim.x) smem[row_words + i] = reinterpret_cast<const uint32_t *>(T.dfa_pages)[i];
    __syncthreads();
    const uint32_t *rows = smem;
    const uint16_t *pages = reinterpret_cast<const uint16_t *>(smem + row_words);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = lane_id();
    uint4 *ring = reinterpret_cast<uint4 *>(smem + ((row_words + page_words + 3u) & ~3u)) + wave * (kFolRingWords / 4u);
    const uint32_t waves_total = gridDim.x * (kFolBlock / kWave);
    for (uint32_t r = blockIdx.x * (kFolBlock / kWave) + wave; r < L.n_regions; r += waves_total) {
        const uint32_t seg_units = 1u << L.seg_log2;
        const uint32_t start = L.g0 + (r * 64u + lane) * seg_units;
        // ---- pass 1: where the chain enters the lane's segment ----
        uint32_t e_in, none = 0;
        if (r == 0 && lane == 0) {
            e_in = L.entry;
        } else {
            const uint32_t ps = start - seg_units;
            uint32_t p0 = max(ps, L.entry);
            const uint32_t lim = min(start, L.own_end);
            if (lim > L.runup) p0 = max(p0, lim - L.runup);
            e_in = fol_walk<RANGE, false, false>(T, L, rows, pages, ring, lane, p0, lim, 0u, 0u, none);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- pass 2: the lane's own segment from there ----
        const uint32_t bound = min(start + seg_units, L.own_end);
        uint32_t cnt = 0;
        // the lane's words of the start bitmap: [first, end) -- all of them are written (zeros where the chain marks nothing)
        const uint32_t w_first = start >> 5, w_end = start < L.own_end ? ((bound - 1u) >> 5) + 1u : w_first;
        // (an entry at or beyond the bound -- the chain jumps over the segment, or the segment lies behind the owned range: no step)
        const uint32_t exit_pos = fol_walk<RANGE, STATE, true>(T, L, rows, pages, ring, lane, e_in, bound, w_first, w_end, cnt);
        const uint32_t e_next = __shfl_down(e_in, 1);
        const bool differs = lane < 63u && (uint64_t)start + seg_units < L.own_end && exit_pos != e_next;
        if (__any(differs) && lane == 0) L.d_exit[1] = 1ull;
        if (lane == 0) L.d_pred[r] = e_in;
        if (lane == 63) L.d_true[r] = exit_pos;
        if (start < L.own_end && bound == L.own_end) L.d_exit[0] = (unsigned long long)exit_pos;
        {
            const uint32_t ts = 1u << L.tile_log2; // segments per tile of the emit pass
            uint32_t csum = cnt;
            for (uint32_t dd = 1; dd < ts; dd <<= 1) csum += __shfl_down(csum, dd);
            if ((lane & (ts - 1u)) == 0) {
                const uint32_t tile = (r * 64u + lane) >> L.tile_log2;
                const uint32_t tend = (uint32_t)min((uint64_t)L.own_end, (uint64_t)start + (uint64_t)ts * seg_units);
                L.d_sync[tile] = e_in < tend ? e_in : ~0u;
                L.d_counts[tile] = csum;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// every region's entry is the exit of the region before it
__global__ __launch_bounds__(256) void k_longest_follow_check(LongestFollowLaunch L) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0 || r >= L.n_regions) return;
    if (L.d_pred[r] != L.d_true[r - 1]) L.d_exit[1] = 1ull;
}

uint32_t longest_follow_seg_units() { return kFolSegUnits; }
uint32_t longest_follow_lanes_per_cu() { return 2u * kFolBlock; }
// rows of the table the kernel can keep in LDS next to `page_bytes` of class pages (0: range classes)
uint32_t longest_follow_hot_rows(uint32_t n_cls, uint32_t n_states, uint32_t page_bytes) {
    if (!n_cls || page_bytes + 16u > kFolRowBytesMax) return 0u;
    return (uint32_t)std::min<uint64_t>(n_states, (kFolRowBytesMax - page_bytes - 16u) / ((uint64_t)n_cls * 4u));
}

hipError_t launch_longest_follow(const DevTables &t, const LongestFollowLaunch &l, bool range, bool state, hipStream_t stream) {
    const size_t lds = (((size_t)l.hot_rows * t.n_cls + (range ? 0 : (t.dfa_pages_bytes + 3) / 4) + 3) & ~(size_t)3) * 4 + (size_t)(kFolBlock / kWave) * kFolRingWords * 4;
#define ACGPU_FOL(R, S)                                                                                                                   \
    do {                                                                                                                                  \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_longest_follow<R, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                                    \
        hipLaunchKernelGGL((k_longest_follow<R, S>), dim3(l.grid), dim3(kFolBlock), lds, stream, t, l);                                   \
    } while (0)
    if (range) {
        if (state) ACGPU_FOL(true, true);
        else ACGPU_FOL(true, false);
    } else {
        if (state) ACGPU_FOL(false, true);
        else ACGPU_FOL(false, false);
    }
#undef ACGPU_FOL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (l.n_regions > 1) hipLaunchKernelGGL(k_longest_follow_check, dim3((l.n_regions + 255) / 256), dim3(256), 0, stream, l);
    return hipGetLastError();
}

} // namespace acgpu
