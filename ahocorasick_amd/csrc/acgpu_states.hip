// acgpu_states.hip -- AhoCorasickSet/Map over dictionaries that match nearly everywhere: the automaton's state at every unit,
// then the records from the states (gfx950).
//
// The reference's loop (S/AhoCorasickSet.java:204-226: one transition per unit, fail links until a node has the edge; output walk
// :522-535: the node's own keyword, then its suffixes, longest first) is what runs here, chunk by chunk:
//  * k_ac_states: a lane owns a chunk of 1024 units (fewer for texts that would not fill the chip), starts at the root max_len - 1
//    units before it (from there on its state is the sequential automaton's) and leaves the state behind every unit -- 4 bytes
//    per unit, the h-id of acgpu_build.cpp 6d with bit 23 = "reports matches".  A state of the DENSE group (the root, depth 1
//    and 2, more than three children) has a row of resolved transitions: the first rows in LDS, the others in memory; a COMPACT
//    state is one 16-byte node {fail | counts, three edges}: a miss moves to the fail state WITHOUT taking the unit (one more
//    iteration) -- the automaton of the reference's README dictionary is 29 MB this way (345 MB as a resolved table, a cache miss
//    per unit).  Every iteration is one step of every lane's own chain (no lock step: a lane that follows a fail link falls one
//    iteration behind), written as selects; ONE 16-byte gather serves a row in memory and a node alike.  Text: a ring of 8-unit
//    blocks per lane in LDS, topped up by all lanes together; states: staged in LDS and stored ROW by row -- group g of all 64
//    lanes is a kilobyte of consecutive memory (st_index) -- once every walking lane has passed the group.  The number of keywords
//    a state reports rides in the transitions: the kernel leaves its chunks' record counts.
//  * (a prefix sum over the chunks' counts: launch_exclusive_scan)
//  * k_ac_states_out<MAP>: a wave per chunk; states -> hy_mask (Map: {mask, id list} in one gather) -> every lane writes the records
//    of its four positions into an LDS window at their place among the step's records (the masks' bits from the top: position,
//    then longest first) -> the window is copied out, 64 consecutive records per store; Map records fetch their ids then.
// Bound by the rate of L2 requests -- one gather per lane and step in k_ac_states, one per position (Map: and per record) in
// k_ac_states_out -- not by the text stream and not by latency (two workgroups per CU are slower: EXPERIMENTS.md, round 5).
#include <algorithm>
#include <cstdio>

#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

// Two shapes of the walk kernel (ACGPU_ST_WGS workgroups of 16 waves per CU): one with deep rings and staging, two with half of each
#ifndef ACGPU_ST_WGS
#define ACGPU_ST_WGS 1
#endif
constexpr int kStBlock = 1024;                  // 16 waves
constexpr int kStWgsPerCu = ACGPU_ST_WGS;
constexpr uint32_t kStChunkLog2 = 10;           // a lane's chunk: 1024 units (less for texts that would not fill the chip)
constexpr uint32_t kStRingBlocks = kStWgsPerCu == 1 ? 4 : 2;   // 8-unit blocks of text per lane in LDS
constexpr uint32_t kStRingWords = kStRingBlocks * 64 * 4;      // per wave: [block][64 lanes] of 16 bytes of text
constexpr uint32_t kStRefillEvery = kStWgsPerCu == 1 ? 16 : 4; // iterations between two refills of the rings (a lane takes at most one unit per iteration)
constexpr uint32_t kStStageSlots = kStWgsPerCu == 1 ? 16 : 8;  // staged states per lane
constexpr uint32_t kStStageWords = kStStageSlots * 64;         // per wave: [slot][64 lanes] of states on their way to memory
constexpr uint32_t kStWindow = 1024;            // records a wave of the record pass stages in LDS at a time
constexpr uint32_t kStFlushEvery = kStWgsPerCu == 1 ? 8 : 4;   // iterations between two flushes of the staged states
constexpr uint32_t kStRowBytesMax = (kStWgsPerCu == 1 ? 159 : 79) * 1024 - (kStBlock / kWave) * (kStRingWords + kStStageWords) * 4; // LDS left for rows and pages: 31 / 15 KiB

// Where the state behind the o-th unit of the chunk of lane l of wave w lies: [wave][group of four units][lane][4] -- lanes that
// walk at the same pace store a kilobyte of consecutive memory with one instruction (64 x 16 bytes), where chunk-major order
// would touch 64 lines; the record pass reads a chunk's groups a kilobyte apart (the 64 chunks of a wave on one XCD: its L2 holds the rows).
__device__ __forceinline__ size_t st_index(uint32_t chunk_log2, uint32_t w, uint32_t l, uint32_t o) {
    return ((((size_t)w << (chunk_log2 - 2u)) + (o >> 2)) * 64u + l) * 4u + (o & 3u);
}

// wave64 inclusive prefix sum with DPP row shifts and row broadcasts (as in the tile kernels)
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t st_sel(bool c, uint32_t a, uint32_t b) { // c ? a : b without control flow
    const uint32_t m = 0u - (uint32_t)c;
    return (a & m) | (b & ~m);
}

struct __attribute__((packed, aligned(2))) StUnits8 {
    uint32_t d[4];
};

#ifdef ACGPU_TIMING
__device__ unsigned long long g_st_timing[8]; // s_memtime ticks (core clocks here) summed over the waves: refills, flushes, steps up to the transition, of them the gather, total, iterations, waves
#define ST_T0() const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define ST_ACC(i) tm[i] += __builtin_amdgcn_s_memtime() - t0_
#else
#define ST_T0()
#define ST_ACC(i)
#endif

template <bool RANGE>
__global__ __launch_bounds__(kStBlock, 4 * kStWgsPerCu) void k_ac_states(DevTables T, AcStatesLaunch L) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // [hot rows][class pages (table classes)][text rings][staged states]
    const uint32_t n_cls = T.n_cls;
    const uint32_t row_words = L.hot_rows * n_cls, page_words = RANGE ? 0u : (T.dfa_pages_bytes + 3u) / 4u;
    for (uint32_t i = threadIdx.x; i < row_words; i += blockDim.x) smem[i] = T.hy_dense[i];
    for (uint32_t i = threadIdx.x; i < page_words; i += blockDim.x) smem[row_words + i] = reinterpret_cast<const uint32_t *>(T.dfa_pages)[i];
    __syncthreads();
    const uint32_t *rows = smem;
    const uint16_t *pages = reinterpret_cast<const uint16_t *>(smem + row_words);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = lane_id();
    uint32_t *wave_mem = smem + ((row_words + page_words + 3u) & ~3u) + wave * (kStRingWords + kStStageWords);
    uint4 *ring = reinterpret_cast<uint4 *>(wave_mem);
    uint32_t *stage = wave_mem + kStRingWords;
    const uint32_t waves_total = gridDim.x * (kStBlock / kWave);
    const uint16_t *hay = L.d_hay;
    const uint32_t nu = L.n_units, hot_last = row_words ? row_words - 1u : 0u, n_dense = T.hy_n_dense;
    const uint32_t dense_last = n_dense * n_cls - 1u, node_last = T.hy_n_states - n_dense ? T.hy_n_states - n_dense - 1u : 0u;
    const uint32_t node_quad0 = (uint32_t)((T.hy_nodes - T.hy_dense) >> 2);
    for (uint32_t w = blockIdx.x * (kStBlock / kWave) + wave; w < L.n_waves; w += waves_total) {
        const uint64_t wb64 = (uint64_t)L.g0 + ((uint64_t)(w * 64u + lane) << L.chunk_log2);
        const bool mine = wb64 < L.own_end;
        const uint32_t wb = mine ? (uint32_t)wb64 : L.own_end;                   // the first position whose state the lane writes
        const uint32_t we = (uint32_t)std::min<uint64_t>(wb64 + (1u << L.chunk_log2), L.own_end); // one past the last
        uint32_t pos = wb > L.halo ? wb - L.halo : 0u;                           // the root stands here
        uint32_t s = 0;
        uint32_t have_end = pos >> 3;   // blocks [have_end - kStRingBlocks, have_end) are in the ring (none yet)
        uint32_t cnt = 0;               // records of the chunk
        const uint32_t count_from = std::max(wb, L.own_begin);
        bool active = mine && pos < we;
        // The states go to memory through the staging slots, ROW by row: group g of four units of all 64 lanes is one kilobyte of
        // consecutive memory (st_index), written by one instruction once every lane that is still walking has passed it --
        // whole lines, and only every kStFlushEvery iterations: a store in the loop's body would be waited for (stores and loads
        // count in one counter here, a wait for a gather's data is a wait for every store issued before it, and the
        // acknowledgement of a store -- of a partly written line above all -- takes longer than a gather).  A lane that runs ahead
        // of the slowest by the staging's depth waits for it (the wave lasts as long as its slowest lane anyway).
        uint32_t rows_out = 0; // wave-uniform: groups [0, rows_out) of every lane are in memory
        auto flush = [&](bool last) {
            const uint32_t o_done = pos > wb ? pos - wb : 0u;
            uint32_t lim = wave_min_u32((active && !last) ? (o_done >> 2) : ~0u);
            const uint32_t top = wave_max_u32(mine ? ((we - wb) >> 2) : 0u); // (complete groups of the longest chunk)
            lim = std::min(lim, top);
            for (uint32_t g = rows_out; g < lim; ++g) {
                if (mine && (g + 1u) * 4u <= we - wb) {
                    const uint32_t o = g * 4u;
                    const uint4 v = make_uint4(stage[((o + 0u) & (kStStageSlots - 1u)) * 64u + lane], stage[((o + 1u) & (kStStageSlots - 1u)) * 64u + lane],
                                               stage[((o + 2u) & (kStStageSlots - 1u)) * 64u + lane], stage[((o + 3u) & (kStStageSlots - 1u)) * 64u + lane]);
                    *reinterpret_cast<uint4 *>(L.d_state + st_index(L.chunk_log2, w, lane, o)) = v;
                }
            }
            rows_out = std::max(rows_out, lim);
        };
        // The text comes through the ring, ALL lanes topping theirs up together every kStRefillEvery iterations (to four blocks
        // from the one the lane stands in: 25 units and more, a lane takes at most one per iteration).  A lane that fetched its
        // next block whenever it needed one made every iteration wait for memory: 64 lanes at 64 phases, so in every iteration
        // some lane's block is the first of a 128-byte line -- a miss of 2 us that the whole wave waits for.
        auto load_block = [&](uint32_t b) -> uint4 {
            const uint32_t b0 = b * 8u;
            if (b0 + 8u <= nu) {
                const StUnits8 v = *reinterpret_cast<const StUnits8 *>(hay + b0);
                return make_uint4(v.d[0], v.d[1], v.d[2], v.d[3]);
            }
            uint32_t t4[4] = {0u, 0u, 0u, 0u}; // the buffer's last, partial block
            for (uint32_t k = 0; k < 8u && b0 + k < nu; ++k) t4[k >> 1] |= (uint32_t)hay[b0 + k] << (16u * (k & 1u));
            return make_uint4(t4[0], t4[1], t4[2], t4[3]);
        };
        auto refill = [&]() {
            const uint32_t want_end = (pos >> 3) + kStRingBlocks;
            uint4 blk[kStRingBlocks];
            bool take[kStRingBlocks];
#pragma unroll
            for (uint32_t k = 0; k < kStRingBlocks; ++k) {
                take[k] = active && have_end + k < want_end && (have_end + k) * 8u < nu;
                blk[k] = take[k] ? load_block(have_end + k) : make_uint4(0u, 0u, 0u, 0u);
            }
            uint32_t got = 0;
#pragma unroll
            for (uint32_t k = 0; k < kStRingBlocks; ++k)
                if (take[k]) {
                    ring[((have_end + k) & (kStRingBlocks - 1u)) * 64u + lane] = blk[k];
                    ++got;
                }
            have_end += got;
        };
        auto class_at = [&](uint32_t p) -> uint32_t {
            const uint32_t u = reinterpret_cast<const uint16_t *>(ring + ((p >> 3) & (kStRingBlocks - 1u)) * 64u + lane)[p & 7u];
            if (RANGE) {
                const uint32_t dlt = u - T.cls_base;
                return dlt < T.cls_span ? dlt + 1u : 0u;
            }
            return pages[128u + ((uint32_t)reinterpret_cast<const unsigned char *>(pages)[u >> 8] << 8) + (u & 255u)];
        };
        uint32_t cls_next = 0;
        bool cls_ok = false;
        uint32_t it = 0;
#ifdef ACGPU_TIMING
        unsigned long long tm[4] = {0, 0, 0, 0};
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        while (__any(active)) {
            if ((it & (kStRefillEvery - 1u)) == 0u) {
                ST_T0();
                refill();
                ST_ACC(0);
            }
            ST_T0();
            const uint32_t xb = pos >> 3;
            const bool can_step = active && xb < have_end && (pos < wb || pos - wb < rows_out * 4u + kStStageSlots); // (text there, a staging slot free)
            if (can_step) {
                // the unit's class: looked up during the step before (under its gather) if the lane took a unit then
                const uint32_t cls = cls_ok ? cls_next : class_at(pos);
                // the three places a transition can come from, all asked (a lane that does not need one reads entry 0 of it)
                const bool in_dense = s < n_dense;
                const uint32_t idx = in_dense ? s * n_cls + cls : 0u;
                const uint32_t e_lds = rows[std::min(idx, hot_last)];
                // (rows and nodes lie in one allocation, the nodes node_quad0 16-byte groups behind the first row: one gather serves
                // either.  Everything below is selects on purpose: as if / else chains the compiler built twenty branches per step.)
                const bool in_lds = idx < row_words;
                const uint32_t quad = st_sel(in_dense, st_sel(in_lds, 0u, std::min(idx, dense_last) >> 2), node_quad0 + std::min(s - n_dense, node_last));
#ifdef ACGPU_TIMING
                asm volatile("s_nop 0" :: "v"(quad), "v"(e_lds));
                const unsigned long long tg_ = __builtin_amdgcn_s_memtime();
#endif
                const uint4 nd = reinterpret_cast<const uint4 *>(T.hy_dense)[quad];
                // (while the gather is on its way: the class of the NEXT unit -- one to three LDS round trips off the chain of the next step)
                const bool have1 = ((pos + 1u) >> 3) < have_end;
                const uint32_t cls1 = have1 ? class_at(pos + 1u) : 0u;
#ifdef ACGPU_TIMING
                asm volatile("s_nop 0" :: "v"(nd.x));
                tm[3] += __builtin_amdgcn_s_memtime() - tg_;
#endif
                const uint32_t iw = idx & 3u;
                const uint32_t e_glb = st_sel(iw == 0u, nd.x, st_sel(iw == 1u, nd.y, st_sel(iw == 2u, nd.z, nd.w)));
                const uint32_t e_d = st_sel(in_lds, e_lds, e_glb);
                const bool m1 = (nd.y >> 24) == cls, m2 = (nd.z >> 24) == cls, m3 = (nd.w >> 24) == cls;
                const bool hit = cls != 0u && (m1 || m2 || m3);
                const uint32_t edge = st_sel(m1, nd.y, st_sel(m2, nd.z, nd.w));
                const uint32_t shift = st_sel(m1, kHyNodeCountShift, st_sel(m2, kHyNodeCountShift + 3u, kHyNodeCountShift + 6u));
                // a node: the edge's child; a unit of no keyword: the root, whatever the state; else the fail state, which looks at this unit again
                const uint32_t ns_c = st_sel(hit, edge & 0xffffffu, st_sel(cls == 0u, 0u, nd.x & kHyIdMask));
                const uint32_t ns = st_sel(in_dense, e_d & 0xffffffu, ns_c); // the state behind the unit (| kHyOut)
                uint32_t n_rep = st_sel(in_dense, e_d >> kHyDenseCountShift, st_sel(hit, (nd.x >> shift) & 7u, 0u)); // how many keywords it reports
                const bool took = in_dense || hit || cls == 0u;
                s = ns & kHyIdMask;
                cls_next = st_sel(took, cls1, cls); // (a fail state looks at the same unit again)
                cls_ok = took ? have1 : true;
#ifdef ACGPU_TIMING
                asm volatile("s_nop 0" :: "v"(s));
                tm[2] += __builtin_amdgcn_s_memtime() - t0_;
#endif
                if (took) {
                    if (!in_dense && n_rep == kHyNodeCountMany) n_rep = (uint32_t)__popc(T.hy_mask[s]); // (rare)
                    if (pos >= count_from) cnt += n_rep;
                    if (pos >= wb) stage[((pos - wb) & (kStStageSlots - 1u)) * 64u + lane] = ns;
                    ++pos;
                    active = pos < we;
                }
            }
            if ((++it & (kStFlushEvery - 1u)) == 0u) {
                ST_T0();
                flush(false);
                ST_ACC(1);
            }
        }
        flush(true);
#ifdef ACGPU_TIMING
        if (lane == 0) {
            atomicAdd(&g_st_timing[0], tm[0]);
            atomicAdd(&g_st_timing[1], tm[1]);
            atomicAdd(&g_st_timing[2], tm[2]);
            atomicAdd(&g_st_timing[3], tm[3]);
            atomicAdd(&g_st_timing[4], __builtin_amdgcn_s_memtime() - tw0);
            atomicAdd(&g_st_timing[5], (unsigned long long)it);
            atomicAdd(&g_st_timing[6], 1ull);
        }
#endif
        if (mine && ((we - wb) & 3u)) // the chunk's last, partial group (only where the owned range ends)
            for (uint32_t q = wb + ((we - wb) & ~3u); q < we; ++q) L.d_state[st_index(L.chunk_log2, w, lane, q - wb)] = stage[((q - wb) & (kStStageSlots - 1u)) * 64u + lane];
        if (mine) L.d_counts[w * 64u + lane] = cnt;
        __builtin_amdgcn_wave_barrier();
    }
}

// the records of the chunks, behind the prefix sums of their counts: a wave per chunk, 256 positions per step (four per lane);
// the states and masks of all (up to four) steps are asked for before the first is used
template <bool MAP>
__global__ __launch_bounds__(256) void k_ac_states_out(DevTables T, AcStatesLaunch L) {
    constexpr int kSteps = (1 << kStChunkLog2) / 256;
    const uint32_t lane = lane_id();
    // the 16 workgroups that take the 64 chunks of one wave of k_ac_states sit on ONE XCD (workgroups go round the eight XCDs)
    const uint32_t bq = blockIdx.x / 8u;
    const uint32_t kw = (bq / 16u) * 8u + blockIdx.x % 8u, kl = (bq % 16u) * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t chunk = kw * 64u + kl;
    if (chunk >= L.n_chunks) return;
    const uint64_t cb = (uint64_t)L.g0 + ((uint64_t)chunk << L.chunk_log2);
    const uint2 *outs = reinterpret_cast<const uint2 *>(T.hy_out);
    unsigned long long base = L.d_offsets[chunk];
    const uint32_t steps = (1u << L.chunk_log2) / 256u;
    uint32_t sv[kSteps][4], m[kSteps][4];
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
        const uint64_t p64 = cb + (uint32_t)st * 256u + lane * 4u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)st < steps) {
            const size_t at = st_index(L.chunk_log2, kw, kl, (uint32_t)st * 256u + lane * 4u);
            if (p64 + 4u <= L.own_end) {
                v = *reinterpret_cast<const uint4 *>(L.d_state + at);
            } else if (p64 < L.own_end) { // the owned range's last, partial group
                v.x = L.d_state[at];
                if (p64 + 1u < L.own_end) v.y = L.d_state[at + 1u];
                if (p64 + 2u < L.own_end) v.z = L.d_state[at + 2u];
            }
            if (p64 < L.own_begin) { // (the first group reaches up to three positions in front of the owned range)
                if (p64 + 0u < L.own_begin) v.x = 0u;
                if (p64 + 1u < L.own_begin) v.y = 0u;
                if (p64 + 2u < L.own_begin) v.z = 0u;
                if (p64 + 3u < L.own_begin) v.w = 0u;
            }
        }
        sv[st][0] = v.x; sv[st][1] = v.y; sv[st][2] = v.z; sv[st][3] = v.w;
    }
#pragma unroll
    for (int st = 0; st < kSteps; ++st)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t at = (sv[st][k] & kHyOut) ? (sv[st][k] & kHyIdMask) : 0u; // (the root reports nothing)
            if (MAP) { // (the mask and where the state's keyword ids begin: one gather)
                const uint2 o = outs[at];
                m[st][k] = o.x;
                sv[st][k] = o.y;
            } else {
                m[st][k] = T.hy_mask[at];
            }
        }
    // A step's records go through LDS: every lane writes those of its four positions into the wave's staging window at their
    // place among the step's records (one per iteration: the masks of positions 0/1 and 2/3 as two 64-bit words whose bits, taken
    // from the top, come in the records' order -- position, then longest first), then the wave copies the window out, 64
    // consecutive records per store.  (Lanes storing their own records straight to memory: scattered 8-byte stores, 2.2 ms on
    // the README word list; rounds of 64 records with a binary search for each record's lane: 80 instructions per round, 1.5 ms.)
    __shared__ int2 win_all[256 / kWave][kStWindow];
    __shared__ int32_t wid_all[MAP ? 256 / kWave : 1][MAP ? kStWindow : 1];
    int2 *win = win_all[threadIdx.x / kWave];
    int32_t *wid = wid_all[MAP ? threadIdx.x / kWave : 0];
    int32_t *out = reinterpret_cast<int32_t *>(L.d_out);
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
        const uint32_t c = (uint32_t)(__popc(m[st][0]) + __popc(m[st][1]) + __popc(m[st][2]) + __popc(m[st][3]));
        const uint32_t incl = wave_inclusive_scan_dpp(c);
        const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
        if (total == 0) continue; // wave-uniform
        const uint32_t p0 = (uint32_t)(cb + (uint32_t)st * 256u) + lane * 4u;
        const uint32_t ex = incl - c;
        for (uint32_t w0 = 0; w0 < total; w0 += kStWindow) { // (one window unless 256 positions hold more than kStWindow records)
            uint32_t at = ex;
            unsigned long long cur = ((unsigned long long)m[st][0] << 32) | m[st][1], nxt = ((unsigned long long)m[st][2] << 32) | m[st][3];
            if (ex + c <= w0 || ex >= w0 + kStWindow) cur = nxt = 0ull; // none of the lane's records in this window
            uint32_t pair = 0;          // 0: positions 0 and 1, 2: positions 2 and 3
            uint32_t idx = 0, ipos = ~0u; // Map records: where the next record's keyword id stands in hy_ids, the position it belongs to
            while (__any((cur | nxt) != 0ull)) {
                if (cur == 0ull && nxt != 0ull) {
                    cur = nxt;
                    nxt = 0ull;
                    pair = 2u;
                }
                if (cur != 0ull) {
                    const uint32_t b = 63u - (uint32_t)__clzll((long long)cur);
                    cur &= ~(1ull << b);
                    const uint32_t k = pair + (b < 32u ? 1u : 0u), len = (b & 31u) + 1u;
                    const uint32_t end = p0 + k + 1u;
                    if (MAP && ipos != k) { // the first (longest) keyword of this position
                        idx = sv[st][0];
                        idx = k == 1u ? sv[st][1] : idx;
                        idx = k == 2u ? sv[st][2] : idx;
                        idx = k == 3u ? sv[st][3] : idx;
                        ipos = k;
                    }
                    if (at >= w0 && at < w0 + kStWindow) {
                        win[at - w0] = make_int2((int)(end - len), (int)end);
                        if (MAP) wid[at - w0] = (int32_t)idx; // (the id itself is fetched when the window is copied out: 64 independent gathers)
                    }
                    ++idx;
                    ++at;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_win = min(total - w0, kStWindow);
            for (uint32_t j = lane; j < n_win; j += kWave) {
                const unsigned long long dst = base + w0 + j;
                if (dst < L.cap) {
                    const int2 r = win[j];
                    if (MAP) {
                        out[dst * 3] = r.x;
                        out[dst * 3 + 1] = r.y;
                        const uint32_t iw = (uint32_t)wid[j]; // (bit 31: the id itself -- the only keyword of its state)
                        out[dst * 3 + 2] = (iw & 0x80000000u) ? (int32_t)(iw & 0x7fffffffu) : (int32_t)T.hy_ids[iw];
                    } else {
                        reinterpret_cast<int2 *>(out)[dst] = r;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        base += total;
    }
}

uint32_t ac_states_chunk_units() { return 1u << kStChunkLog2; }
uint32_t ac_states_lanes_per_cu() { return (uint32_t)kStWgsPerCu * kStBlock; }
// rows of the dense group the kernel can keep in LDS next to `page_bytes` of class pages (0: range classes)
uint32_t ac_states_hot_rows(uint32_t n_cls, uint32_t n_dense, uint32_t page_bytes) {
    if (!n_cls || page_bytes + 16u > kStRowBytesMax) return 0u;
    return (uint32_t)std::min<uint64_t>(n_dense, (kStRowBytesMax - page_bytes - 16u) / ((uint64_t)n_cls * 4u));
}

hipError_t launch_ac_states(const DevTables &t, const AcStatesLaunch &l, bool range, hipStream_t stream) {
    const size_t lds = (((size_t)l.hot_rows * t.n_cls + (range ? 0 : (t.dfa_pages_bytes + 3) / 4) + 3) & ~(size_t)3) * 4 + (size_t)(kStBlock / kWave) * (kStRingWords + kStStageWords) * 4;
    const void *fn = range ? reinterpret_cast<const void *>(&k_ac_states<true>) : reinterpret_cast<const void *>(&k_ac_states<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (range) hipLaunchKernelGGL((k_ac_states<true>), dim3(l.grid), dim3(kStBlock), lds, stream, t, l);
    else hipLaunchKernelGGL((k_ac_states<false>), dim3(l.grid), dim3(kStBlock), lds, stream, t, l);
#ifdef ACGPU_TIMING
    {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[8] = {0}, z[8] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_st_timing), sizeof(h));
        if (h[6]) fprintf(stderr, "[states timing] waves %llu, %.0f iterations each: total %.0f | refills %.0f | flushes %.0f | step to the transition %.0f, of it the gather %.0f (s_memtime ticks per wave)\n",
                          h[6], (double)h[5] / h[6], (double)h[4] / h[6], (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[3] / h[6]);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_timing), z, sizeof(z));
    }
#endif
    return hipGetLastError();
}

hipError_t launch_ac_states_out(const DevTables &t, const AcStatesLaunch &l, bool map, hipStream_t stream) {
    const dim3 grid(((l.n_waves + 7) / 8) * 8 * 16), block(256); // (16 workgroups per wave of k_ac_states, waves in groups of eight: see the kernel)
    if (map) hipLaunchKernelGGL((k_ac_states_out<true>), grid, block, 0, stream, t, l);
    else hipLaunchKernelGGL((k_ac_states_out<false>), grid, block, 0, stream, t, l);
    return hipGetLastError();
}

} // namespace acgpu
