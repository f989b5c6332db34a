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
//            keywords leftwards from the depth-K node (hashed edges in HBM/L2), collecting every keyword that ends
//            at e -- exactly the keywords on the reference's output chain of the state reached at e.
//
// Work distribution: a wave owns a contiguous REGION of the haystack and streams it as 512-unit tiles, lane l
// holding units [8l, 8l+8) of the tile (one coalesced 1 KiB load per wave and tile).  Because a wave meets its
// candidates in text order, a record's rank inside its region is a running wave-uniform count plus a wave prefix
// sum; the finalize pass (prefix sum over regions + permutation) then yields the reference's emission order
// (end ascending, longest first) without any sort.
#include <hip/hip_runtime.h>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

constexpr int kTileBlock = 1024;               // 16 waves share one LDS copy of the bitmap
constexpr int kTileUnits = 512;                // units per wave tile (64 lanes x 8 units)
constexpr int kCandCap = 64 + kTileUnits;      // a tile adds at most 512 candidates to fewer than 64 pending ones

int tile_block_threads() { return kTileBlock; }

size_t tile_lds_bytes(const DevTables &t, int block_threads) {
    return (size_t)t.filt_words * 4 + (size_t)(block_threads / kWave) * kCandCap * sizeof(uint2);
}

struct TileCtx {
    const DevTables &T;
    const TileLaunch &L;
    uint2 *cand;        // this wave's candidate queue in LDS: (pos, kgram index)
    uint32_t cand_n;    // wave-uniform
    uint32_t rank_base; // wave-uniform: records emitted so far in the current region
};

// Verification of up to 64 queued candidates, one per lane.  Every lane of the wave calls this.
__device__ __forceinline__ void verify_batch(TileCtx &c, uint32_t head, uint32_t nb) {
    const DevTables &T = c.T;
    const uint32_t lane = lane_id();
    const bool act = lane < nb;
    uint32_t e = 0, node = 0, m = 0, one_len = 0, one_id = 0;
    if (act) {
        const uint2 ent = c.cand[head + lane];
        e = ent.x + 1; // exclusive end
        node = T.kgram_node[ent.y];
        uint32_t d = T.filt_k;
        // walk the reversed trie leftwards; every terminal node met is a keyword ending at e (increasing length)
        for (;;) {
            const uint32_t info = T.rinfo[node];
            if (info & 0x7fffffffu) {
                ++m;
                one_len = d;
                one_id = (info & 0x7fffffffu) - 1;
            }
            if (!(info >> 31) || e <= d) break; // leaf, or the buffer starts here
            uint32_t u = c.L.d_hay[e - 1 - d];
            if (!T.cs) u = T.lower[u];
            const uint32_t child = hashed_goto(T.rhkeys, T.rhvals, T.rhmask, node, u);
            if (child == ~0u) break;
            node = child;
            ++d;
        }
    }
    const uint32_t incl = wave_inclusive_scan(m);
    const uint32_t total = __shfl(incl, kWave - 1);
    if (total == 0) return;
    const uint32_t prefix = incl - m;
    unsigned long long gbase = 0;
    if (lane == 0) gbase = atomicAdd(c.L.d_counter, (unsigned long long)total);
    gbase = __shfl(gbase, 0);
    if (m == 1) {
        const unsigned long long slot = gbase + prefix;
        if (slot < c.L.cap) {
            ScratchRec r{(int32_t)(e - one_len), (int32_t)e, (int32_t)one_id, c.rank_base + prefix};
            *reinterpret_cast<uint4 *>(&c.L.d_scratch[slot]) = *reinterpret_cast<const uint4 *>(&r);
        }
    }
    if (__any(m >= 2)) {
        // several keywords end here: the reference reports the longest first (S/AhoCorasickSet.java:526-532), the walk
        // meets them shortest first -> second walk, writing the j-th one met to slot (m-1-j)
        if (m >= 2) {
            uint32_t node2 = T.kgram_node[c.cand[head + lane].y];
            uint32_t d = T.filt_k, j = 0;
            for (;;) {
                const uint32_t info = T.rinfo[node2];
                if (info & 0x7fffffffu) {
                    const uint32_t k = prefix + (m - 1 - j);
                    const unsigned long long slot = gbase + k;
                    if (slot < c.L.cap) {
                        ScratchRec r{(int32_t)(e - d), (int32_t)e, (int32_t)((info & 0x7fffffffu) - 1), c.rank_base + k};
                        *reinterpret_cast<uint4 *>(&c.L.d_scratch[slot]) = *reinterpret_cast<const uint4 *>(&r);
                    }
                    if (++j == m) break;
                }
                if (!(info >> 31) || e <= d) break;
                uint32_t u = c.L.d_hay[e - 1 - d];
                if (!T.cs) u = T.lower[u];
                const uint32_t child = hashed_goto(T.rhkeys, T.rhvals, T.rhmask, node2, u);
                if (child == ~0u) break;
                node2 = child;
                ++d;
            }
        }
    }
    c.rank_base += total;
}

// drain the candidate queue down to fewer than `keep_below` entries (64 inside a region, 1 at its end)
__device__ __forceinline__ void drain(TileCtx &c, uint32_t keep_below) {
    uint32_t head = 0;
    while (c.cand_n - head >= keep_below && c.cand_n > head) {
        const uint32_t nb = min(c.cand_n - head, (uint32_t)kWave);
        verify_batch(c, head, nb);
        head += nb;
    }
    if (head) { // move the leftovers (fewer than 64) to the front
        const uint32_t left = c.cand_n - head;
        uint2 tmp = make_uint2(0, 0);
        if (lane_id() < left) tmp = c.cand[head + lane_id()];
        __builtin_amdgcn_wave_barrier();
        if (lane_id() < left) c.cand[lane_id()] = tmp;
        __builtin_amdgcn_wave_barrier();
        c.cand_n = left;
    }
}

template <int K>
__global__ __launch_bounds__(kTileBlock) void k_ac_tile(DevTables T, TileLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem);
    uint2 *cand_all = reinterpret_cast<uint2 *>(smem + (size_t)T.filt_words * 4);
    for (uint32_t i = threadIdx.x; i < T.filt_words; i += blockDim.x) bits[i] = T.filt_bits[i];
    __syncthreads();

    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = threadIdx.x / kWave;
    const uint32_t waves_total = gridDim.x * (kTileBlock / kWave);
    TileCtx c{T, L, cand_all + wave_in_block * kCandCap, 0, 0};

    const uint32_t n = T.filt_n;
    uint32_t nK = 1;
#pragma unroll
    for (int i = 0; i < K; ++i) nK *= n;
    const bool range_cls = T.range_cls != 0;
    const uint32_t cls_base = T.cls_base, cls_span = T.cls_span;
    const uint16_t *cls_lut = T.cls_lut;
    auto tcls = [&](uint32_t unit) -> uint32_t {
        if (range_cls) return min(unit - cls_base, cls_span); // units outside [base, base+span) -> span ("other")
        return cls_lut[unit];
    };

    for (uint32_t region = blockIdx.x * (kTileBlock / kWave) + wave_in_block; region < L.n_regions; region += waves_total) {
        const uint32_t rb = L.own_begin + region * L.region_units;
        uint32_t re = rb + L.region_units;
        if (re > L.own_end || re < rb) re = L.own_end;
        c.rank_base = 0;
        c.cand_n = 0;
        // tiles start 16-byte aligned; units before rb belong to the previous region (or the halo) and are masked out
        for (uint32_t tile = rb & ~7u; tile < re; tile += kTileUnits) {
            const uint32_t v = tile + lane * 8;
            uint32_t cur[8];
            uint32_t prv[8]; // the 8 units before v (only the last K-1 are used)
            {
                uint4 w = make_uint4(0, 0, 0, 0), p = make_uint4(0, 0, 0, 0);
                if (v < re) {
                    if (v + 8 <= L.n_units) {
                        w = *reinterpret_cast<const uint4 *>(L.d_hay + v);
                    } else {
                        uint32_t tmp[4] = {0, 0, 0, 0};
                        for (uint32_t j = 0; j < 8 && v + j < L.n_units; ++j) tmp[j >> 1] |= (uint32_t)L.d_hay[v + j] << (16 * (j & 1));
                        w = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
                    }
                    if (K > 1 && v >= 8) p = *reinterpret_cast<const uint4 *>(L.d_hay + v - 8);
                }
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w}, pp[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    cur[j] = (ww[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    prv[j] = (pp[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                }
            }
            // classes of units v-(K-1) .. v+7
            uint32_t a[8 + K - 1];
#pragma unroll
            for (int j = 0; j < K - 1; ++j) a[j] = tcls(prv[8 - (K - 1) + j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[K - 1 + j] = tcls(cur[j]);
            // K-gram index of position v+j (last unit least significant), rolling
            uint32_t h = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) h = h * n + a[j];
            uint32_t mask = 0;
            uint32_t idx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j > 0) h = h * n + a[K - 1 + j] - a[j - 1] * nK;
                idx[j] = h;
                const uint32_t pos = v + j;
                const uint32_t word = bits[h >> 5];
                const bool ok = ((word >> (h & 31)) & 1u) && pos >= rb && pos < re && pos + 1 >= (uint32_t)K;
                mask |= (ok ? 1u : 0u) << j;
            }
            // compaction in text order: lane-major, then position within the lane
            const uint32_t cnt = __popc(mask);
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t total = __shfl(incl, kWave - 1);
            if (total) {
                uint32_t slot = c.cand_n + incl - cnt;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (mask & (1u << j)) c.cand[slot++] = make_uint2(v + j, idx[j]);
                }
                c.cand_n += total;
                __builtin_amdgcn_wave_barrier();
                if (c.cand_n >= kWave) drain(c, kWave);
            }
        }
        drain(c, 1);
        if (lane == 0) L.d_region_counts[region] = c.rank_base;
    }
}

hipError_t launch_ac_tile(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name) {
#define ACGPU_TILE_CASE(KK)                                                                                             \
    case KK: {                                                                                                          \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_tile<KK>),                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);               \
        if (e != hipSuccess) return e;                                                                                  \
        hipLaunchKernelGGL(k_ac_tile<KK>, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, t, l);                      \
        if (kernel_name) *kernel_name = "k_ac_tile<" #KK ">";                                                           \
        break;                                                                                                          \
    }
    switch (t.filt_k) {
        ACGPU_TILE_CASE(1)
        ACGPU_TILE_CASE(2)
        ACGPU_TILE_CASE(3)
        ACGPU_TILE_CASE(4)
        ACGPU_TILE_CASE(5)
        ACGPU_TILE_CASE(6)
        ACGPU_TILE_CASE(7)
        ACGPU_TILE_CASE(8)
    default: return hipErrorInvalidValue;
    }
#undef ACGPU_TILE_CASE
    return hipGetLastError();
}

} // namespace acgpu
