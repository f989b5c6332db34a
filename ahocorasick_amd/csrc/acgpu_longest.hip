// acgpu_longest.hip -- LongestMatchSet/Map on gfx950.
//
// The reference (S/LongestMatchSet.java:192-265 with S/SetMatchQueue.java:45-95) delivers the leftmost-longest
// non-overlapping matches, i.e. the greedy chain  pos -> pos + max(L[pos], 1)  started at 0, where L[pos] is the
// length of the longest keyword starting at pos (T/LongestMatchTest.java:30-42 is the same statement).
//
//  k_longest_scan  : L[pos] for every unit.  The automaton of the REVERSED keywords is run right-to-left over the
//                    haystack; after consuming text[pos..] backwards, the longest keyword on the state's output chain
//                    is the longest keyword starting at pos.  One chunk of start positions per lane, warmed up on
//                    the (max_keyword_len-1) units to its right; hot rows and their lengths in LDS.
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

template <typename E>
struct RevDenseStep {
    const E *lds, *glob;
    uint32_t lds_entries, n_cls;
    const uint16_t *cls_lut;
    uint32_t cls_base, cls_span;
    bool range_cls;
    __device__ __forceinline__ uint32_t operator()(uint32_t s, uint32_t unit) const {
        uint32_t cls;
        if (range_cls) {
            const uint32_t d = unit - cls_base;
            cls = d < cls_span ? d + 1 : 0;
        } else {
            cls = cls_lut[unit];
        }
        const uint32_t idx = s * n_cls + cls;
        return idx < lds_entries ? (uint32_t)lds[idx] : (uint32_t)glob[idx];
    }
};

struct RevSparseStep {
    const uint64_t *hkeys;
    const uint32_t *hvals, *fail;
    const uint16_t *lower;
    uint32_t hmask;
    bool cs;
    __device__ __forceinline__ uint32_t operator()(uint32_t s, uint32_t unit) const {
        const uint32_t u = cs ? unit : (uint32_t)lower[unit];
        for (;;) {
            const uint32_t n = hashed_goto(hkeys, hvals, hmask, s, u);
            if (n != ~0u) return n;
            if (s == 0) return 0;
            s = fail[s];
        }
    }
};

template <typename Step, typename LenT>
__device__ __forceinline__ void longest_scan_body(const DevTables &T, const LongestScanLaunch &L, const Step &step,
                                                  const uint32_t *lds_len, uint32_t lds_states) {
    const uint32_t halo = T.max_len > 0 ? T.max_len - 1 : 0;
    const uint32_t lanes_total = gridDim.x * blockDim.x;
    LenT *out_len = reinterpret_cast<LenT *>(L.d_len);
    const uint32_t rounds = (L.n_chunks + lanes_total - 1) / lanes_total;
    const uint32_t n_line = (L.chunk_units + halo + 126) / 64 + 1; // wave-uniform trip count, in 128-byte lines
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t chunk = round * lanes_total + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = chunk < L.n_chunks;
        const uint32_t cb = valid ? L.own_begin + chunk * L.chunk_units : L.own_end; // first owned start position
        uint32_t ce = cb + L.chunk_units;
        if (ce > L.own_end || ce < cb) ce = L.own_end;
        uint32_t top = ce + halo; // scan units [cb, top) right to left
        if (top > L.n_units || top < ce) top = L.n_units;
        const uint32_t l_first = top > 0 ? ((top - 1) & ~63u) : 0; // highest 128-byte line touched
        uint32_t s = 0;
        for (uint32_t it = 0; it < n_line; ++it) {
            const bool line_ok = valid && it * 64 <= l_first;
            const uint32_t vb = l_first - (line_ok ? it * 64 : 0);
            // a lane requests the 8 vectors of a line back to back: the line crosses the fabric once
            uint4 line[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t v = vb + k * 8;
                uint4 w = make_uint4(0, 0, 0, 0);
                if (line_ok && v + 8 > cb && v < top) {
                    if (v + 8 <= L.n_units) {
                        w = *reinterpret_cast<const uint4 *>(L.d_hay + v);
                    } else {
                        uint32_t tmp[4] = {0, 0, 0, 0};
                        for (uint32_t j = 0; j < 8 && v + j < L.n_units; ++j) tmp[j >> 1] |= (uint32_t)L.d_hay[v + j] << (16 * (j & 1));
                        w = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
                    }
                }
                line[k] = w;
            }
#pragma unroll
            for (int k = 7; k >= 0; --k) {
                const uint32_t v = vb + k * 8;
                const bool act = line_ok && v + 8 > cb && v < top; // vector intersects [cb, top)
                const uint32_t words[4] = {line[k].x, line[k].y, line[k].z, line[k].w};
                uint32_t lens[8], states[8];
#pragma unroll
                for (int j = 7; j >= 0; --j) {
                    const uint32_t unit = (words[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    const uint32_t pos = v + j;
                    if (act && pos < top) s = step(s, unit); // units at/after `top` are outside the warm-up window or the buffer
                    states[j] = s;
                    lens[j] = s < T.first_out ? 0u : (s < lds_states ? lds_len[s] : T.out_len[s]);
                }
                if (act && v < ce) {
                    if (v >= cb && v + 8 <= ce) {
                        if (sizeof(LenT) == 2) {
                            const uint4 o = make_uint4(lens[0] | lens[1] << 16, lens[2] | lens[3] << 16, lens[4] | lens[5] << 16,
                                                       lens[6] | lens[7] << 16);
                            *reinterpret_cast<uint4 *>(out_len + v) = o;
                        } else {
                            *reinterpret_cast<uint4 *>(out_len + v) = make_uint4(lens[0], lens[1], lens[2], lens[3]);
                            *reinterpret_cast<uint4 *>(out_len + v + 4) = make_uint4(lens[4], lens[5], lens[6], lens[7]);
                        }
                        if (L.d_state) {
                            *reinterpret_cast<uint4 *>(L.d_state + v) = make_uint4(states[0], states[1], states[2], states[3]);
                            *reinterpret_cast<uint4 *>(L.d_state + v + 4) = make_uint4(states[4], states[5], states[6], states[7]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t pos = v + j;
                            if (pos >= cb && pos < ce) {
                                out_len[pos] = (LenT)lens[j];
                                if (L.d_state) L.d_state[pos] = states[j];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <typename E, typename LenT>
__global__ __launch_bounds__(kLScanBlock) void k_longest_scan_dense(DevTables T, LongestScanLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds_states = T.n_cls ? T.lds_entries / T.n_cls : 0;
    uint32_t *lds_len = reinterpret_cast<uint32_t *>(smem);
    E *tab = reinterpret_cast<E *>(smem + (size_t)lds_states * 4);
    const E *glob = reinterpret_cast<const E *>(T.dfa);
    for (uint32_t i = threadIdx.x; i < lds_states; i += blockDim.x) lds_len[i] = T.out_len[i];
    for (uint32_t i = threadIdx.x; i < T.lds_entries; i += blockDim.x) tab[i] = glob[i];
    __syncthreads();
    RevDenseStep<E> step{tab, glob, T.lds_entries, T.n_cls, T.cls_lut, T.cls_base, T.cls_span, T.range_cls != 0};
    longest_scan_body<RevDenseStep<E>, LenT>(T, L, step, lds_len, lds_states);
}

template <typename LenT>
__global__ __launch_bounds__(kLScanBlock) void k_longest_scan_sparse(DevTables T, LongestScanLaunch L) {
    RevSparseStep step{T.hkeys, T.hvals, T.fail, T.lower, T.hmask, T.cs != 0};
    longest_scan_body<RevSparseStep, LenT>(T, L, step, nullptr, 0);
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
    if (t.dense) {
        if (t.entry_bytes == 2) {
            if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_scan_dense<uint16_t, uint16_t>), "k_longest_scan_dense<unsigned short, unsigned short>");
            else ACGPU_LAUNCH((k_longest_scan_dense<uint16_t, uint32_t>), "k_longest_scan_dense<unsigned short, unsigned int>");
        } else {
            if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_scan_dense<uint32_t, uint16_t>), "k_longest_scan_dense<unsigned int, unsigned short>");
            else ACGPU_LAUNCH((k_longest_scan_dense<uint32_t, uint32_t>), "k_longest_scan_dense<unsigned int, unsigned int>");
        }
    } else {
        if (l.len_bytes == 2) ACGPU_LAUNCH((k_longest_scan_sparse<uint16_t>), "k_longest_scan_sparse<unsigned short>");
        else ACGPU_LAUNCH((k_longest_scan_sparse<uint32_t>), "k_longest_scan_sparse<unsigned int>");
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
constexpr int kSyncSet = 16;

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
    for (uint32_t q = tb - L.entry > w ? tb - w : L.entry; q < tb && ok; ++q) {
        const uint32_t l = (uint32_t)len[q];
        const uint32_t land = q + (l > 0 ? l : 1u);
        if (land >= tb) insert(land);
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
// (or out of the owned range), counting or writing the matches met.
template <typename LenT, bool WRITE>
__global__ __launch_bounds__(256) void k_longest_chain(LongestChainLaunch L, const uint32_t *S) {
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
    uint32_t pos = start, count = 0;
    uint64_t dst = WRITE ? L.d_offsets[t] : 0;
    while (pos < target && pos < L.own_end) {
        const uint32_t l = (uint32_t)len[pos];
        if (l > 0) {
            if (WRITE && dst < L.cap) {
                if (L.record_kind == ACGPU_REC_SET) {
                    reinterpret_cast<int2 *>(L.d_out)[dst] = make_int2((int)pos, (int)(pos + l));
                } else {
                    int32_t *o = reinterpret_cast<int32_t *>(L.d_out) + dst * 3;
                    o[0] = (int)pos;
                    o[1] = (int)(pos + l);
                    o[2] = (int)L.d_out_id[L.d_state[pos]];
                }
            }
            ++dst;
            ++count;
        }
        pos += l > 0 ? l : 1u;
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
    const dim3 grid((l.n_tiles + 255) / 256), block(256);
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
