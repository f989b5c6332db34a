// acgpu_tile_common.h -- pieces shared by the position-parallel tile kernels (acgpu_tile.hip: AhoCorasick all-matches,
// acgpu_wholeword.hip: WholeWord): per-wave candidate queue in LDS, text-order compaction, DPP scans, record slots.
#pragma once
#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

constexpr int kTileBlock = 1024;               // 16 waves share one LDS copy of the filter rows
constexpr int kTileUnits = 512;                // units per wave tile (64 lanes x 8 units)
#ifndef ACGPU_NB
#define ACGPU_NB 2
#endif
#ifndef ACGPU_PREFETCH
#define ACGPU_PREFETCH 4
#endif
constexpr int kVerifyBatches = ACGPU_NB;              // candidates verified per lane and call (independent load chains in flight)
constexpr int kCandCap = 1024;                 // candidate queue entries per wave; a tile adds at most 512
constexpr int kPrefetch = ACGPU_PREFETCH;                   // tiles per group; one group of loads is in flight per wave
#ifndef ACGPU_RESERVE
#define ACGPU_RESERVE 256
#endif
constexpr uint32_t kReserve = ACGPU_RESERVE;             // scratch slots a wave reserves per atomic

// The ablation switches of TileLaunch::debug exist only in builds with -DACGPU_ABLATION (tools/build_variant.sh abl
// -DACGPU_ABLATION; tools/kbench.py and tools/collect_profiles.sh select that library for their ablation variants): in
// the product build every test is a constant 0 and costs neither an SGPR nor a branch in the hot loops.
#ifdef ACGPU_ABLATION
#define ACGPU_DBG(L, bits) ((L).debug & (bits))
#else
#define ACGPU_DBG(L, bits) 0u
#endif

struct TileCtx {
    const DevTables *Tp;
    const TileLaunch *Lp;
    uint32_t *cand;     // this wave's candidate queue in LDS: end positions (last unit index), in text order
    uint32_t cand_n;    // wave-uniform
    uint32_t rank_base; // wave-uniform: records emitted so far in the current region
    uint32_t res_cur;   // wave-uniform: next free reserved scratch slot (scratch capacity < 2^32)
    uint32_t res_left;          // wave-uniform: reserved slots left
    uint32_t slot_limit = 0;    // wave-uniform: end of this workgroup's scratch slice (set by the first reservation)
    // L2 form of the AhoCorasick kernel: a queue entry is a position RELATIVE to pos_base (the start of the current region:
    // the queue is drained at every region seam) in pos16[], and in cand[] what the second-level stage already knew about
    // the candidate: kQiKnown | class of the unit in front of the K-gram << 20 | K-gram index -- with it the verification
    // goes straight to the K-gram node (0: not known, the verification reads the text window first)
    uint16_t *pos16 = nullptr;
    uint32_t pos_base = 0;
    uint32_t region = 0; // WholeWord, region-local records: the region the queued run starts belong to (wave-uniform)
    uint32_t area = 0;   // k_ww_pp: record index in TileLaunch::d_region_recs of the current region's (fused tail: the wave's) first record
    uint32_t wg = 0;     // the workgroup's number: its scratch slice and counters (blockIdx.x, or the start ticket of the fused tail)
#ifdef ACGPU_TIMING
    unsigned long long vt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // verification phases (k_ac_tile: windows, K-gram nodes, walks, emission;
                                                          // k_ww_tile: windows, chunk 1, chunk 2, hash + Bloom, probes, emission, calls)
#endif
};

__device__ __forceinline__ void store_rec(const TileLaunch &L, uint32_t slot, uint32_t start, uint32_t end, uint32_t id,
                                          uint32_t rank) {
    if ((uint64_t)slot < L.cap && !ACGPU_DBG(L, 128u)) { // 128: ablation, records are not stored
        // non-temporal: the records are read once, by the permute pass, and must not push the just-streamed text (which
        // the verification gathers from) out of the L2; about 1 % at config 2, same-box A/B against -DACGPU_REC_PLAIN
#ifndef ACGPU_REC_PLAIN
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u v = {start, end, id, rank};
        __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(&L.d_scratch[slot]));
#else
        const uint4 v = make_uint4(start, end, id, rank);
        *reinterpret_cast<uint4 *>(&L.d_scratch[slot]) = v;
#endif
    }
}

constexpr uint32_t kQiKnown = 0x80000000u, kQiIdxMask = 0xfffffu, kQiLeftShift = 20;
constexpr uint32_t kQiShort = 0x40000000u; // a keyword of fewer than K units may end here (DevTables::kshort / ks_keys)
constexpr uint32_t kQiChecked = 0x20000000u; // the second level has looked: no kQiShort = no short keyword (entries without it: unknown)

// wave64 inclusive prefix sum with DPP row shifts and row broadcasts (6 dependent v_add_u32_dpp)
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
    return x;
}

// value of x in lane-1; lane 0 receives `carry` (v_mov_b32_dpp wave_shr:1)
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t x, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)x, 0x138, 0xf, 0xf, false);
}

// append this lane's candidates (bit j of mask: position v+j) to the wave queue in text order
__device__ __forceinline__ void enqueue(TileCtx &c, uint32_t mask, uint32_t v) {
    const uint32_t cnt = __popc(mask);
    const uint32_t incl = wave_inclusive_scan_dpp(cnt);
    const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
    if (total) {
        uint32_t slot = c.cand_n + incl - cnt;
        while (__any(mask != 0)) { // as many rounds as the busiest lane has candidates (2-3 at 2 % density)
            if (mask != 0) {
                c.cand[slot++] = v + (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1;
            }
        }
        c.cand_n += total;
        __builtin_amdgcn_wave_barrier();
    }
}


// Reserve `total` record slots for this wave (wave-uniform); returns a functor-like pair through references:
// slot of the k-th record = k < old_left ? old_cur + k : fresh + (k - old_left).
struct SlotRange {
    uint32_t old_cur, old_left, fresh, limit;
    // slots at or beyond the slice's end do not exist (~0u: store_rec drops the record; the host redoes the call with one
    // slice or reports ACGPU_E_OVERFLOW) -- but every slot below it is written, which the permute pass relies on
    __device__ __forceinline__ uint32_t slot(uint32_t k) const {
        const uint32_t s = k < old_left ? old_cur + k : fresh + (k - old_left);
        return s < limit ? s : ~0u;
    }
};

__device__ __forceinline__ SlotRange reserve_slots(TileCtx &c, uint32_t total) {
    SlotRange r{c.res_cur, c.res_left, 0, c.slot_limit};
    if (total > r.old_left) {
        const uint32_t need = total - r.old_left;
        const uint32_t take = need > kReserve ? need : kReserve;
        // this workgroup's slice of the scratch and its counter (see TileLaunch::n_slices)
        const uint32_t slice = c.Lp->n_slices > 1 ? c.wg % c.Lp->n_slices : 0u;
        const uint32_t S = c.Lp->slice_slots, base = slice * S; // (n_slices * S <= scratch capacity < 2^32)
        uint32_t fresh = 0;
        if (lane_id() == 0) {
            const unsigned long long got = atomicAdd(c.Lp->d_counter + (size_t)slice * kCounterStride, (unsigned long long)take);
            fresh = got < (unsigned long long)S ? base + (uint32_t)got : base + S;
            // the slice is full: the host redoes the call with one slice (n_slices == 1: the scratch is full and the host
            // reports ACGPU_E_OVERFLOW from the exact counts)
            if (got + take > (unsigned long long)S && c.Lp->n_slices > 1) atomicOr(c.Lp->d_overflow, 2u);
        }
        c.slot_limit = r.limit = base + S;
        r.fresh = __builtin_amdgcn_readfirstlane(fresh);
        c.res_cur = r.fresh + need;
        c.res_left = take - need;
    } else {
        c.res_cur = r.old_cur + total;
        c.res_left = r.old_left - total;
    }
    return r;
}

// ---- fused tail (TileLaunch::fused_tail): what the tile kernels' workgroups hand each other ------------------------------------
constexpr unsigned long long kFtDone = 1ull << 62; // d_counter[number * kCounterStride + 2]: {done, the workgroup's records}
constexpr uint32_t kFtWords = 2 + kTileBlock / kWave; // LDS: [0] the workgroup's number (later: the records below it), [1] the slice's fill mark, [2 + w] wave w's records
__device__ __forceinline__ uint32_t ft_wave_sum(uint32_t x) { return __builtin_amdgcn_readlane(wave_inclusive_scan_dpp(x), kWave - 1); }

// The workgroup's number: workgroups are numbered in the order in which they start, so that every lower number is running or
// done whatever share of the grid is resident.  Thread 0, before the kernel's first barrier; read wg_words[0] behind it.
__device__ __forceinline__ void ft_take_number(const TileLaunch &L, uint32_t *wg_words) {
    wg_words[0] = (uint32_t)atomicAdd(L.d_counter + 3, 1ull);
    wg_words[1] = 0u;
}

// this workgroup is done scanning and has `mine` records (one thread)
__device__ __forceinline__ void ft_publish(const TileLaunch &L, uint32_t wg, uint32_t mine) {
    __hip_atomic_store(L.d_counter + (size_t)wg * kCounterStride + 2, kFtDone | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The records of all workgroups with a lower number; waits for them.  Every wave of the workgroup calls it (a barrier inside):
// ONE wave asks -- a grid of waiting workgroups that all poll takes the memory system away from the workgroups still scanning --
// and the sum reaches the others through the LDS word that held the number.
__device__ __forceinline__ uint32_t ft_below(const TileLaunch &L, uint32_t wg, uint32_t *wg_words) {
    const uint32_t lane = lane_id();
    if (threadIdx.x / kWave == 0) {
        uint32_t below = 0;
        for (uint32_t w0 = 0; w0 < wg; w0 += kWave) {
            unsigned long long v;
            do {
                v = w0 + lane < wg ? __hip_atomic_load(L.d_counter + (size_t)(w0 + lane) * kCounterStride + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFtDone;
                if (__all((v & kFtDone) != 0ull)) break;
                __builtin_amdgcn_s_sleep(16);
            } while (true);
            below += ft_wave_sum((uint32_t)v); // (record counts fit 32 bits: the scratch holds fewer than 2^32 records)
        }
        if (lane == 0) wg_words[0] = below;
    }
    __syncthreads();
    return wg_words[0];
}

// The workgroup with the last number has waited for all the others' counts: the call's count, the overflow word (final: a
// workgroup publishes its count when its scan is over), and the NEXT call's counter set zeroed -- no copy, no memset and no
// further launch on the stream.  Every thread of every workgroup calls it.
__device__ __forceinline__ void ft_report(const TileLaunch &L, uint32_t wg, unsigned long long total) {
    if (wg + 1u != gridDim.x) return;
    if (threadIdx.x == 0) {
        const uint32_t flag = __hip_atomic_load(L.d_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (L.tail_d_result) {
            L.tail_d_result->n_records = total;
            L.tail_d_result->redone = flag;
            L.tail_d_result->reserved = 0;
        }
        if (L.tail_result) {
            L.tail_result[0] = total;
            L.tail_result[1] = flag;
        }
        __hip_atomic_store(L.d_overflow, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
    }
    if (L.tail_zero_counters)
        for (uint32_t i = threadIdx.x; i < (uint32_t)kMaxSlices; i += blockDim.x) {
            unsigned long long *z = L.tail_zero_counters + (size_t)i * kCounterStride;
            z[0] = 0; z[1] = 0; z[2] = 0; z[3] = 0;
        }
}

} // namespace acgpu
