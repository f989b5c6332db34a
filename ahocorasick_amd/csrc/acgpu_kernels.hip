// acgpu_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the matching hot path.
//
// k_ac_scan   : the reference's per-unit automaton loop (S/AhoCorasickSet.java:204-226) + output
//               walk (:522-535), re-cast as: one contiguous chunk of the haystack per lane, started at the
//               root (max_keyword_len-1) units before the chunk so that every match ending inside the
//               chunk is seen; transition rows of the shallow states staged in LDS; match records
//               compacted with a wavefront ballot into a per-wave LDS queue and appended to HBM in
//               batches with one atomic per batch.  Dense tables: k_ac_dfa (round 4: two chunks per lane,
//               branch-free steps); hashed edges + fail links: k_ac_scan_sparse; k_ac_scan_dense is the
//               one-chain form of rounds 1-3 (haystacks below 64 units, A/B).
// k_scan_*    : exclusive prefix sum of the per-chunk match counts.
// k_permute   : scatters the unordered records to their final, reference-ordered slots.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "acgpu_device.h"
#include "acgpu_kernels.h"

namespace acgpu {

#ifndef ACGPU_SCAN_MINBLOCKS
#define ACGPU_SCAN_MINBLOCKS 1
#endif
#ifndef ACGPU_DFA_CHAINS
#define ACGPU_DFA_CHAINS 2 // chunks per lane in k_ac_dfa
#endif
constexpr int kScanBlock = 1024;                 // 16 waves: one workgroup per CU shares one LDS copy of the hot rows
constexpr int kQueueCap = 128;                   // records per wave queue
constexpr int kQueueFlush = kQueueCap - kWave;   // flush when fewer than 64 free slots remain

int scan_block_threads() { return kScanBlock; }
size_t scan_queue_bytes(int block_threads) { return (size_t)(block_threads / kWave) * kQueueCap * sizeof(ScratchRec); }

// ---- per-wave record queue in LDS ---------------------------------------------------------------------
struct WaveQueue {
    ScratchRec *q;  // this wave's kQueueCap slots in LDS
    uint32_t n;     // wave-uniform fill count
};

__device__ __forceinline__ void queue_flush(WaveQueue &wq, ScratchRec *scratch, uint64_t cap,
                                            unsigned long long *counter) {
    if (wq.n == 0) return;
    __builtin_amdgcn_wave_barrier();
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(counter, (unsigned long long)wq.n);
    base = __shfl(base, 0);
    for (uint32_t i = lane_id(); i < wq.n; i += kWave) {
        if (base + i < cap) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&wq.q[i]);
            *reinterpret_cast<uint4 *>(&scratch[base + i]) = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    wq.n = 0;
}

// Every lane of the wave must call this (wave-uniform control flow); `t` is the lane's pending output
// state (0 = nothing to emit).  Walks the compressed output chain: own/inherited longest match first,
// then each shorter suffix match (S/AhoCorasickSet.java:526-532).
__device__ __forceinline__ void emit_chain(const DevTables &T, uint32_t t, uint32_t end, uint32_t &rank, WaveQueue &wq,
                                           ScratchRec *scratch, uint64_t cap, unsigned long long *counter) {
    for (;;) {
        const uint64_t m = __ballot(t != 0);
        if (m == 0) break;
        if (t != 0) {
            ScratchRec r;
            r.end = (int32_t)end;
            r.start = (int32_t)(end - T.out_len[t]);
            r.id = (int32_t)T.out_id[t];
            r.rank = rank++;
            const uint32_t slot = wq.n + (uint32_t)__popcll(m & lanemask_lt());
            *reinterpret_cast<uint4 *>(&wq.q[slot]) = *reinterpret_cast<const uint4 *>(&r);
            t = T.out_link[t];
        }
        wq.n += (uint32_t)__popcll(m);
        if (wq.n > kQueueFlush) queue_flush(wq, scratch, cap, counter);
    }
}

// ---- transition functions -------------------------------------------------------------------------------
template <typename E>
struct DenseStep {
    const E *lds;      // first lds_entries entries of the table
    const E *glob;
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

struct SparseStep {
    const uint64_t *hkeys;
    const uint32_t *hvals;
    const uint32_t *fail;
    const uint16_t *lower;
    uint32_t hmask;
    bool cs;
    __device__ __forceinline__ uint32_t goto_edge(uint32_t s, uint32_t u) const { // ~0u when absent
        return hashed_goto(hkeys, hvals, hmask, s, u);
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t s, uint32_t unit) const {
        const uint32_t u = cs ? unit : (uint32_t)lower[unit];
        for (;;) { // S/AhoCorasickSet.java:207-221: follow fail links until a node has a transition (root always does)
            const uint32_t n = goto_edge(s, u);
            if (n != ~0u) return n;
            if (s == 0) return 0;
            s = fail[s];
        }
    }
};

// ---- the scan kernel ------------------------------------------------------------------------------------
template <typename Step>
__device__ __forceinline__ void ac_scan_body(const DevTables &T, const ScanLaunch &L, const Step &step, WaveQueue &wq) {
    const uint32_t halo = T.max_len > 0 ? T.max_len - 1 : 0;
    const uint32_t lanes_total = gridDim.x * blockDim.x;
    // all waves run the same number of chunk rounds so that control flow stays wave-uniform
    const uint32_t rounds = (L.n_chunks + lanes_total - 1) / lanes_total;
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t chunk = round * lanes_total + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = chunk < L.n_chunks;
        const uint32_t cb = valid ? L.own_begin + chunk * L.chunk_units : L.own_end; // first owned unit
        uint32_t ce = cb + L.chunk_units;                                            // one past the last owned unit
        if (ce > L.own_end || ce < cb) ce = L.own_end;
        uint32_t rs = cb > halo ? cb - halo : 0;
        rs &= ~63u; // whole 128-byte lines: a lane requests the 8 vectors of a line back to back, so the line crosses
                    // the fabric once instead of once per vector; the extra warm-up is harmless
        // wave-uniform trip count: chunk + halo + alignment slack, in 64-unit lines
        const uint32_t n_line = (L.chunk_units + halo + 63 + 63) / 64;
        uint32_t s = 0;
        uint32_t rank = 0;
        for (uint32_t it = 0; it < n_line; ++it) {
            const uint32_t vb = rs + it * 64;
            uint4 line[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t v = vb + k * 8;
                uint4 w = make_uint4(0, 0, 0, 0);
                if (valid && v < ce) {
                    if (v + 8 <= L.n_units) {
                        w = *reinterpret_cast<const uint4 *>(L.d_hay + v);
                    } else { // tail of the buffer: never read past n_units
                        uint32_t tmp[4] = {0, 0, 0, 0};
                        for (uint32_t j = 0; j < 8 && v + j < L.n_units; ++j) tmp[j >> 1] |= (uint32_t)L.d_hay[v + j] << (16 * (j & 1));
                        w = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
                    }
                }
                line[k] = w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t v = vb + k * 8;
                const bool act = valid && v < ce;
                const uint32_t words[4] = {line[k].x, line[k].y, line[k].z, line[k].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t unit = (words[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    const uint32_t pos = v + j;
                    s = step(s, unit);
                    const bool has = act && s >= T.first_out && pos >= cb && pos < ce;
                    if (__any(has)) emit_chain(T, has ? s : 0u, pos + 1, rank, wq, L.d_scratch, L.cap, L.d_counter);
                }
            }
        }
        if (valid) L.d_chunk_counts[chunk] = rank;
    }
    queue_flush(wq, L.d_scratch, L.cap, L.d_counter);
}

template <typename E>
__global__ __launch_bounds__(kScanBlock, ACGPU_SCAN_MINBLOCKS) void k_ac_scan_dense(DevTables T, ScanLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ScratchRec *queues = reinterpret_cast<ScratchRec *>(smem);
    E *tab = reinterpret_cast<E *>(smem + (size_t)(kScanBlock / kWave) * kQueueCap * sizeof(ScratchRec));
    const E *glob = reinterpret_cast<const E *>(T.dfa);
    for (uint32_t i = threadIdx.x; i < T.lds_entries; i += blockDim.x) tab[i] = glob[i];
    __syncthreads();
    WaveQueue wq{queues + (threadIdx.x / kWave) * kQueueCap, 0};
    DenseStep<E> step{tab, glob, T.lds_entries, T.n_cls, T.cls_lut, T.cls_base, T.cls_span, T.range_cls != 0};
    ac_scan_body(T, L, step, wq);
}

// ---- k_ac_dfa: the dense chunk scan as straight-line code, NCH chunks per lane -----------------------------
// What bounds a chunk scan is the chain state -> table entry -> state: one LDS read or one cached gather per unit, each
// waiting for the one before.  k_ac_scan_dense ran ONE such chain per lane behind ~30 instructions per unit (class branch,
// LDS-or-global branch, range tests and a ballot per unit); here
//  * a lane owns NCH chunks and steps them alternately -- NCH independent chains per lane, their lookups in flight together;
//  * a step is branch-free: class by subtract / compare / select (range classes; table classes: the 8 class loads of a
//    vector are issued together, ahead of its steps), index by one 24-bit multiply-add, the LDS read with a clamped index AND
//    the global read with a masked index (a lane whose row is in LDS reads entry 0: one cached line for the whole wave) both
//    issued unconditionally, one select;
//  * outputs are looked for once per 8 units (the maximum of the 8 states against first_out, one ballot); the rare vector
//    with an output re-examines its 8 saved states, in order, through the same emit_chain / wave queue as before.
// Records, chunk counts and ordering are k_ac_scan_dense's: chunk = the owned range's chunk_units-slice the match ends in.
template <typename E, bool RANGE, bool GLOB, int NCH>
__global__ __launch_bounds__(kScanBlock) void k_ac_dfa(DevTables T, ScanLaunch L) {
    // (static LDS: addresses are immediates; the host never stages more than kDfaLdsBytes of rows)
    __shared__ __attribute__((aligned(16))) ScratchRec queues[(kScanBlock / kWave) * kQueueCap];
    __shared__ __attribute__((aligned(16))) unsigned char tab8[kDfaLdsBytes];
    E *tab = reinterpret_cast<E *>(tab8);
    const E *glob = reinterpret_cast<const E *>(T.dfa);
    for (uint32_t i = threadIdx.x; i < T.lds_entries; i += blockDim.x) tab[i] = glob[i];
    const uint32_t row_bytes = T.n_cls * (uint32_t)sizeof(E), fo = T.first_out;
    const uint32_t lds_bytes = T.lds_entries * (uint32_t)sizeof(E);
    // table classes: the class table as pages behind the rows (acgpu_build.cpp 7b; the host leaves the room) -- two LDS reads
    // per unit instead of a gather from the 128 KB table in global memory
    const uint32_t pg_off = (lds_bytes + 15u) & ~15u;
    const bool cls_lds = !RANGE && T.dfa_pages != nullptr && (uint64_t)pg_off + T.dfa_pages_bytes <= (uint64_t)kDfaLdsBytes;
    if (cls_lds)
        for (uint32_t i = threadIdx.x; i < T.dfa_pages_bytes / 16; i += blockDim.x)
            reinterpret_cast<uint4 *>(tab8 + pg_off)[i] = reinterpret_cast<const uint4 *>(T.dfa_pages)[i];
    const unsigned char *pg8 = tab8 + pg_off;                                  // the page index: one byte per 256 units
    const uint16_t *pg16 = reinterpret_cast<const uint16_t *>(tab8 + pg_off + 256); // the pages
    __syncthreads();
    WaveQueue wq{queues + (threadIdx.x / kWave) * kQueueCap, 0};
    const uint32_t lds_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_bytes ? lds_bytes - (uint32_t)sizeof(E) : 0u));
    const uint32_t base = T.cls_base, span = T.cls_span;
    const uint16_t *lut = T.cls_lut;
    const uint32_t halo = T.max_len > 0 ? T.max_len - 1 : 0;
    const uint32_t lanes_total = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t rounds = (uint32_t)(((uint64_t)L.n_chunks + (uint64_t)lanes_total * NCH - 1) / ((uint64_t)lanes_total * NCH));
    const uint32_t n_line = (L.chunk_units + halo + 63 + 63) / 64; // chunk + halo + alignment slack, in 128-byte lines
    for (uint32_t round = 0; round < rounds; ++round) {
        uint32_t chunk[NCH], cb[NCH], ce[NCH], rs[NCH], s[NCH], rank[NCH];
        bool valid[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint64_t ck = ((uint64_t)round * NCH + c) * lanes_total + gtid;
            valid[c] = ck < L.n_chunks;
            chunk[c] = (uint32_t)ck;
            cb[c] = valid[c] ? L.own_begin + chunk[c] * L.chunk_units : L.own_end;
            ce[c] = cb[c] + L.chunk_units;
            if (ce[c] > L.own_end || ce[c] < cb[c]) ce[c] = L.own_end;
            rs[c] = (cb[c] > halo ? cb[c] - halo : 0u) & ~63u; // whole lines (see ac_scan_body)
            s[c] = 0;
            rank[c] = 0;
        }
        for (uint32_t it = 0; it < n_line; ++it) {
            uint4 line[NCH][8];
            bool tail = false; // some line of the wave reaches beyond the buffer (its last chunks only)
#pragma unroll
            for (int c = 0; c < NCH; ++c) tail = tail || (uint64_t)rs[c] + (uint64_t)it * 64 + 64 > (uint64_t)L.n_units;
            if (!__any(tail)) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int k = 0; k < 8; ++k) line[c][k] = *reinterpret_cast<const uint4 *>(L.d_hay + rs[c] + it * 64 + k * 8);
            } else { // the vector that holds the buffer's last 8 units, shifted down: units behind the buffer read as zero
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint64_t v = (uint64_t)rs[c] + (uint64_t)it * 64 + k * 8;
                        const uint64_t over = v + 8 > (uint64_t)L.n_units ? v + 8 - L.n_units : 0; // units of the vector behind the buffer
                        const uint4 w = *reinterpret_cast<const uint4 *>(L.d_hay + (over ? (uint64_t)L.n_units - 8 : v));
                        uint64_t lo = (uint64_t)w.x | ((uint64_t)w.y << 32), hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
                        const uint32_t sh = (uint32_t)min(over, (uint64_t)8);
                        if (sh >= 8) { lo = 0; hi = 0; }
                        else if (sh >= 4) { lo = hi >> (16 * (sh - 4)); hi = 0; }
                        else if (sh) { lo = (lo >> (16 * sh)) | (hi << (64 - 16 * sh)); hi >>= 16 * sh; }
                        line[c][k] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t cls[NCH][8], st[NCH][8];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint32_t words[4] = {line[c][k].x, line[c][k].y, line[c][k].z, line[c][k].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t unit = (j & 1) ? words[j >> 1] >> 16 : words[j >> 1] & 0xffffu;
                        if (RANGE) { // (classes as byte offsets into a row)
                            const uint32_t d = unit - base;
                            cls[c][j] = d < span ? (d + 1u) * (uint32_t)sizeof(E) : 0u;
                        } else if (cls_lds) { // (wave-uniform)
                            cls[c][j] = (uint32_t)pg16[((uint32_t)pg8[unit >> 8] << 8) + (unit & 255u)] * (uint32_t)sizeof(E);
                        } else {
                            cls[c][j] = (uint32_t)lut[unit] * (uint32_t)sizeof(E);
                        }
                    }
                }
                uint32_t mx = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const uint32_t off = __umul24(s[c], row_bytes) + cls[c][j]; // byte offset of the entry
                        uint32_t nx = *reinterpret_cast<const E *>(reinterpret_cast<const unsigned char *>(tab) + min(off, lds_last));
                        if (GLOB) {
#ifdef ACGPU_ABLATION // (timing only, bit 4: no lookups in global memory -- such a state goes to the root's entry 0 instead)
                            const uint32_t g = *reinterpret_cast<const E *>(reinterpret_cast<const unsigned char *>(glob) + (off >= lds_bytes && !(L.debug & 4u) ? off : 0u));
#else
                            const uint32_t g = *reinterpret_cast<const E *>(reinterpret_cast<const unsigned char *>(glob) + (off >= lds_bytes ? off : 0u));
#endif
                            nx = off >= lds_bytes ? g : nx;
                        }
                        s[c] = nx;
                        st[c][j] = nx;
                        mx = max(mx, nx);
                    }
                }
                if (__any(mx >= fo)) { // rare: some state of these 8 steps has an output
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const uint32_t v = rs[c] + it * 64 + k * 8;
                        uint32_t m8 = 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            m8 |= (valid[c] && st[c][j] >= fo && v + j >= cb[c] && v + j < ce[c]) ? 1u << j : 0u;
                        while (__any(m8 != 0u)) { // every lane its own lowest step first: a chunk's records in position order
                            const uint32_t j = m8 ? (uint32_t)__builtin_ctz(m8) : 0u;
                            uint32_t t = st[c][0];
#pragma unroll
                            for (int q = 1; q < 8; ++q) t = j == (uint32_t)q ? st[c][q] : t;
                            emit_chain(T, m8 ? t : 0u, v + j + 1, rank[c], wq, L.d_scratch, L.cap, L.d_counter);
                            m8 &= m8 - 1u;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (valid[c]) L.d_chunk_counts[chunk[c]] = rank[c];
    }
    queue_flush(wq, L.d_scratch, L.cap, L.d_counter);
}

int scan_chains(const DevTables &t) { // chunks per lane of the dense chunk scan (the host sizes the chunks for it); 0: k_ac_dfa cannot run
    return t.dense && t.n_states < (1u << 24) && t.n_cls < (1u << 22) && (uint64_t)t.n_states * t.n_cls < (1ull << 30) ? ACGPU_DFA_CHAINS : 0;
}

template <typename E, bool RANGE, bool GLOB>
static hipError_t launch_ac_dfa(const DevTables &t, const ScanLaunch &l, hipStream_t stream) {
    auto *k = &k_ac_dfa<E, RANGE, GLOB, ACGPU_DFA_CHAINS>;
    if ((uint64_t)t.lds_entries * sizeof(E) > (uint64_t)kDfaLdsBytes) return hipErrorInvalidValue; // (lds_states_for keeps below it)
    hipLaunchKernelGGL(k, dim3(l.grid), dim3(l.block), 0, stream, t, l);
    return hipSuccess;
}

__global__ __launch_bounds__(kScanBlock) void k_ac_scan_sparse(DevTables T, ScanLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ScratchRec *queues = reinterpret_cast<ScratchRec *>(smem);
    WaveQueue wq{queues + (threadIdx.x / kWave) * kQueueCap, 0};
    SparseStep step{T.hkeys, T.hvals, T.fail, T.lower, T.hmask, T.cs != 0};
    ac_scan_body(T, L, step, wq);
}

hipError_t launch_ac_scan(const DevTables &t, const ScanLaunch &l, hipStream_t stream, const char **kernel_name) {
    hipError_t e;
    if (t.dense && scan_chains(t) > 0 && !(l.debug & 1u)) { // (debug bit 1: the one-chain kernel of rounds 1-3, for A/B)
        const bool glob = (uint64_t)t.lds_entries < (uint64_t)t.n_states * t.n_cls;
        const bool u16 = t.entry_bytes == 2;
#define ACGPU_DFA_CASE(E, R, G, NAME)                                                    \
    if (u16 == (sizeof(E) == 2) && (t.range_cls != 0) == R && glob == G) {               \
        e = launch_ac_dfa<E, R, G>(t, l, stream);                                          \
        if (e != hipSuccess) return e;                                                    \
        if (kernel_name) *kernel_name = NAME;                                             \
        return hipGetLastError();                                                         \
    }
        ACGPU_DFA_CASE(uint16_t, true, true, "k_ac_dfa<unsigned short, true, true>")
        ACGPU_DFA_CASE(uint16_t, true, false, "k_ac_dfa<unsigned short, true, false>")
        ACGPU_DFA_CASE(uint16_t, false, true, "k_ac_dfa<unsigned short, false, true>")
        ACGPU_DFA_CASE(uint16_t, false, false, "k_ac_dfa<unsigned short, false, false>")
        ACGPU_DFA_CASE(uint32_t, true, true, "k_ac_dfa<unsigned int, true, true>")
        ACGPU_DFA_CASE(uint32_t, true, false, "k_ac_dfa<unsigned int, true, false>")
        ACGPU_DFA_CASE(uint32_t, false, true, "k_ac_dfa<unsigned int, false, true>")
        ACGPU_DFA_CASE(uint32_t, false, false, "k_ac_dfa<unsigned int, false, false>")
#undef ACGPU_DFA_CASE
    }
    if (t.dense) {
        if (t.entry_bytes == 2) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_scan_dense<uint16_t>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_ac_scan_dense<uint16_t>, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, t, l);
            if (kernel_name) *kernel_name = "k_ac_scan_dense<unsigned short>";
        } else {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ac_scan_dense<uint32_t>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_ac_scan_dense<uint32_t>, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, t, l);
            if (kernel_name) *kernel_name = "k_ac_scan_dense<unsigned int>";
        }
    } else {
        hipLaunchKernelGGL(k_ac_scan_sparse, dim3(l.grid), dim3(l.block), l.lds_bytes, stream, t, l);
        if (kernel_name) *kernel_name = "k_ac_scan_sparse";
    }
    return hipGetLastError();
}

// ---- exclusive prefix sum of per-chunk counts ------------------------------------------------------------
constexpr int kScanTile = 2048; // elements per 256-thread block

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *total
__device__ __forceinline__ uint64_t block_exclusive_scan_256(uint64_t v, uint64_t *total) {
    __shared__ uint64_t wsum[4];
    const uint64_t inc = wave_inclusive_scan(v);
    const int w = threadIdx.x / kWave;
    if (lane_id() == kWave - 1) wsum[w] = inc;
    __syncthreads();
    uint64_t base = 0;
    for (int i = 0; i < w; ++i) base += wsum[i];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t *counts, uint32_t n, uint64_t *tile_sums) {
    const uint32_t base = blockIdx.x * kScanTile;
    uint64_t v = 0;
    for (int k = 0; k < kScanTile / 256; ++k) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < n) v += counts[i];
    }
    uint64_t total;
    block_exclusive_scan_256(v, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_tile_offsets(uint64_t *tile_sums, uint32_t n_tiles) {
    // single block: exclusive scan in place, 256 tiles per step with a running carry
    __shared__ uint64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan_256(v, &total);
        const uint64_t carry = carry_s;
        if (i < n_tiles) tile_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry_s; // grand total
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *counts, uint32_t n, const uint64_t *tile_offsets,
                                                   uint64_t *offsets) {
    const uint32_t base = blockIdx.x * kScanTile;
    constexpr int per = kScanTile / 256; // 8 consecutive elements per thread
    uint32_t c[per];
    uint64_t v = 0;
    for (int k = 0; k < per; ++k) {
        const uint32_t i = base + threadIdx.x * per + k;
        c[k] = i < n ? counts[i] : 0;
        v += c[k];
    }
    uint64_t ex = block_exclusive_scan_256(v, nullptr) + tile_offsets[blockIdx.x];
    for (int k = 0; k < per; ++k) {
        const uint32_t i = base + threadIdx.x * per + k;
        if (i < n) offsets[i] = ex;
        ex += c[k];
    }
}

// k_scan_apply for at most 256 tiles: every block derives its tile's offset from the raw tile sums itself (and block 0
// leaves the grand total behind them), which saves the single-block launch in between
__global__ __launch_bounds__(256) void k_scan_apply_few(const uint32_t *counts, uint32_t n, uint64_t *tile_sums, uint32_t n_tiles,
                                                       uint64_t *offsets) {
    __shared__ uint64_t tile_base;
    {
        const uint64_t s = threadIdx.x < n_tiles ? tile_sums[threadIdx.x] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan_256(s, &total);
        if (threadIdx.x == blockIdx.x) tile_base = ex;
        if (blockIdx.x == 0 && threadIdx.x == 0) tile_sums[n_tiles] = total; // grand total
        __syncthreads();
    }
    const uint32_t base = blockIdx.x * kScanTile;
    constexpr int per = kScanTile / 256;
    uint32_t c[per];
    uint64_t v = 0;
    for (int k = 0; k < per; ++k) {
        const uint32_t i = base + threadIdx.x * per + k;
        c[k] = i < n ? counts[i] : 0;
        v += c[k];
    }
    uint64_t ex = block_exclusive_scan_256(v, nullptr) + tile_base;
    for (int k = 0; k < per; ++k) {
        const uint32_t i = base + threadIdx.x * per + k;
        if (i < n) offsets[i] = ex;
        ex += c[k];
    }
}

hipError_t launch_exclusive_scan(const uint32_t *d_counts, uint32_t n, uint64_t *d_offsets, uint64_t *d_tmp,
                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, stream, d_counts, n, d_tmp);
    if (n_tiles <= 256) {
        hipLaunchKernelGGL(k_scan_apply_few, dim3(n_tiles), dim3(256), 0, stream, d_counts, n, d_tmp, n_tiles, d_offsets);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(256), 0, stream, d_tmp, n_tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_tiles), dim3(256), 0, stream, d_counts, n, (const uint64_t *)d_tmp, d_offsets);
    return hipGetLastError();
}

// the tile level of the prefix sum alone: d_tmp[t] = number of flagged elements in front of tile t (kScanTile elements per tile),
// d_tmp[scan_tiles_for(n)] = their total -- for a consumer that ranks inside a tile itself (k_wwl_emit) and needs no offset per element
hipError_t launch_scan_tile_offsets(const uint32_t *d_counts, uint32_t n, uint64_t *d_tmp, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(256), 0, stream, d_counts, n, d_tmp);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(256), 0, stream, d_tmp, n_tiles);
    return hipGetLastError();
}
uint32_t scan_tile_elems() { return kScanTile; }

// ---- permutation to reference order ----------------------------------------------------------------------
template <int REC>
__global__ __launch_bounds__(256) void k_permute(const ScratchRec *scratch, const unsigned long long *counter,
                                                uint64_t slice_slots, const uint64_t *offsets, uint32_t own_begin,
                                                uint32_t chunk_units, int by_start, void *out, uint64_t cap,
                                                const uint32_t *id_map, PermuteTail tail) {
    if (blockIdx.x == 0 && blockIdx.y == 0) { // (see PermuteTail)
        if (tail.zero_counters)
            for (uint32_t i = threadIdx.x; i < (uint32_t)kMaxSlices; i += blockDim.x) {
                tail.zero_counters[(size_t)i * kCounterStride] = 0;
                tail.zero_counters[(size_t)i * kCounterStride + 1] = 0; // (the workgroup sums of the fused finalize)
                tail.zero_counters[(size_t)i * kCounterStride + 2] = 0; // (the fused tail's words: a workgroup's {done, records}, the start tickets)
                tail.zero_counters[(size_t)i * kCounterStride + 3] = 0;
            }
        if (tail.result && threadIdx.x == 0) {
            const unsigned long long total = *tail.total;
            const uint32_t flag = tail.flag ? *tail.flag : 0u;
            tail.result[0] = total;
            tail.result[1] = flag;
            if (tail.d_result) {
                tail.d_result->n_records = total;
                tail.d_result->redone = flag;
                tail.d_result->reserved = 0;
            }
            if (tail.flag) *tail.flag = 0;
            __threadfence_system();
        }
    }
    // blockIdx.y = slice of the scratch (its own counter, slice_slots slots)
    unsigned long long m = counter[(size_t)blockIdx.y * kCounterStride];
    if (m > slice_slots) m = slice_slots; // overflow: the host redoes the call / reports ACGPU_E_OVERFLOW
    scratch += (size_t)blockIdx.y * slice_slots;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(&scratch[i]);
        const int32_t start = (int32_t)raw.x, end = (int32_t)raw.y, id = (int32_t)raw.z;
        const uint32_t rank = raw.w;
        if (rank == ~0u) continue; // hole left by a slot reservation
        // ordering unit: the chunk owning the match's last unit (ALL) or first unit (LONGEST/WHOLEWORD)
        const uint32_t key = by_start ? (uint32_t)start : (uint32_t)(end - 1);
        const uint32_t chunk = (key - own_begin) / chunk_units;
        const uint64_t dst = offsets[chunk] + rank;
        if (dst >= cap) continue;
        if (REC == ACGPU_REC_SET) {
            reinterpret_cast<int2 *>(out)[dst] = make_int2(start, end);
        } else {
            int32_t *o = reinterpret_cast<int32_t *>(out) + dst * 3;
            o[0] = start;
            o[1] = end;
            o[2] = id_map ? (int32_t)id_map[id] : id; // tile-kernel records name the reversed-trie node: map to keyword id
        }
    }
}

// WholeWord, region-local records (TileLaunch::d_region_recs): region r's records lie, in order, in its own slots; this pass
// copies them -- whole regions, coalesced dwords -- to out[offsets[r] ...].  One workgroup per region (grid-stride).
template <int REC>
__global__ __launch_bounds__(256) void k_ww_compact(const int32_t *recs, uint32_t region_cap, const uint32_t *counts, const uint64_t *offsets,
                                                   uint32_t n_regions, int32_t *out, uint64_t cap, PermuteTail tail) {
    if (blockIdx.x == 0) { // (see PermuteTail)
        if (tail.zero_counters)
            for (uint32_t i = threadIdx.x; i < (uint32_t)kMaxSlices; i += blockDim.x) {
                tail.zero_counters[(size_t)i * kCounterStride] = 0;
                tail.zero_counters[(size_t)i * kCounterStride + 1] = 0;
                tail.zero_counters[(size_t)i * kCounterStride + 2] = 0; // (the fused tail's words: a workgroup's {done, records}, the start tickets)
                tail.zero_counters[(size_t)i * kCounterStride + 3] = 0;
            }
        if (tail.result && threadIdx.x == 0) {
            const unsigned long long total = *tail.total;
            const uint32_t flag = tail.flag ? *tail.flag : 0u;
            tail.result[0] = total;
            tail.result[1] = flag;
            if (tail.d_result) {
                tail.d_result->n_records = total;
                tail.d_result->redone = flag;
                tail.d_result->reserved = 0;
            }
            if (tail.flag) *tail.flag = 0;
            __threadfence_system();
        }
    }
    constexpr uint32_t W = REC / 4; // dwords per output record
    for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const uint64_t off = offsets[r];
        if (off >= cap) continue;
        const uint64_t n = min((uint64_t)counts[r], cap - off);
        const int32_t *src = recs + (size_t)r * region_cap * 3;
        // a record per lane and step: one 12-byte load, one 12- or 8-byte store
        typedef int32_t v3i __attribute__((ext_vector_type(3)));
        for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const v3i rec = *reinterpret_cast<const v3i *>(src + i * 3);
            if (W == 3) *reinterpret_cast<v3i *>(out + (off + i) * 3) = rec;
            else reinterpret_cast<int2 *>(out)[off + i] = make_int2(rec.x, rec.y);
        }
    }
}

hipError_t launch_ww_compact(const int32_t *d_region_recs, uint32_t region_cap, const uint32_t *d_region_counts, const uint64_t *d_offsets,
                             uint32_t n_regions, int record_kind, void *d_out, uint64_t out_cap, hipStream_t stream, const PermuteTail *tail,
                             hipEvent_t ev_stop) {
    const PermuteTail tl = tail ? *tail : PermuteTail{nullptr, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid(std::max<uint32_t>(std::min<uint32_t>(n_regions, 8192u), 1u)), block(256);
    const hipEvent_t ev_none = nullptr;
    if (record_kind == ACGPU_REC_SET)
        ACGPU_LAUNCH_EV(k_ww_compact<ACGPU_REC_SET>, grid, block, 0, stream, ev_none, ev_stop, d_region_recs, region_cap, d_region_counts, d_offsets,
                        n_regions, (int32_t *)d_out, out_cap, tl);
    else
        ACGPU_LAUNCH_EV(k_ww_compact<ACGPU_REC_MAP>, grid, block, 0, stream, ev_none, ev_stop, d_region_recs, region_cap, d_region_counts, d_offsets,
                        n_regions, (int32_t *)d_out, out_cap, tl);
    return hipGetLastError();
}

#ifndef ACGPU_PERMUTE_BLOCKS
#define ACGPU_PERMUTE_BLOCKS 2048
#endif
constexpr uint32_t kPermuteBlocks = ACGPU_PERMUTE_BLOCKS;

hipError_t launch_permute(const ScratchRec *d_scratch, const unsigned long long *d_counter, uint32_t n_slices, uint64_t slice_slots,
                          const uint64_t *d_offsets, uint32_t own_begin, uint32_t chunk_units, int by_start,
                          int record_kind, void *d_out, uint64_t out_cap, const uint32_t *d_id_map, hipStream_t stream,
                          const PermuteTail *tail) {
    if (n_slices < 1) n_slices = 1;
    const PermuteTail tl = tail ? *tail : PermuteTail{nullptr, nullptr, nullptr, nullptr, nullptr};
    // (8192 or 32768 workgroups instead of 2048 were not faster: the pass is bound by its scattered 12-byte stores)
    const dim3 grid(std::max<uint32_t>(kPermuteBlocks / n_slices, 8u), n_slices);
    if (record_kind == ACGPU_REC_SET)
        hipLaunchKernelGGL(k_permute<ACGPU_REC_SET>, grid, dim3(256), 0, stream, d_scratch, d_counter, slice_slots,
                           d_offsets, own_begin, chunk_units, by_start, d_out, out_cap, d_id_map, tl);
    else
        hipLaunchKernelGGL(k_permute<ACGPU_REC_MAP>, grid, dim3(256), 0, stream, d_scratch, d_counter, slice_slots,
                           d_offsets, own_begin, chunk_units, by_start, d_out, out_cap, d_id_map, tl);
    return hipGetLastError();
}

// ---- fused finalize of the tile kernel: offsets + permutation in one launch -------------------------------------------------
// wave64 inclusive prefix sum (DPP row shifts + row broadcasts), as in the tile kernels
__device__ __forceinline__ uint32_t pw_wave_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

template <int REC>
__global__ __launch_bounds__(256) void k_permute_wg(const ScratchRec *scratch, const unsigned long long *counter, uint32_t n_wg,
                                                   uint64_t slice_slots, const uint32_t *region_counts, uint32_t n_regions,
                                                   uint32_t regions_per_wg, uint32_t own_begin, uint32_t chunk_units, void *out,
                                                   uint64_t cap, const uint32_t *id_map, PermuteTail tail) {
    __shared__ uint32_t local[kPermuteWgRegions]; // exclusive offsets of this workgroup's regions
    __shared__ uint32_t part[3][4];               // per wave: records below this workgroup, all records, region counts
    const uint32_t wg = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    // The slice's fill mark and this thread's first record are asked for HERE, with the counts below: none of these loads
    // depends on another, and the pass is a chain of memory round trips (20 us for 1.3 M records at config 2), not bandwidth.
    unsigned long long m = counter[(size_t)wg * kCounterStride];
    const uint64_t i_first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const ScratchRec *slice = scratch + (size_t)wg * slice_slots;
    const uint4 raw_first = *reinterpret_cast<const uint4 *>(&slice[i_first < slice_slots ? i_first : 0]); // (inside the slice whatever it holds)
    // (record counts fit 32 bits: the scratch holds fewer than 2^32 records)
    uint32_t below = 0, all = 0;
    for (uint32_t w = threadIdx.x; w < n_wg; w += blockDim.x) {
        const uint32_t v = (uint32_t)counter[(size_t)w * kCounterStride + 1];
        if (w < wg) below += v;
        all += v;
    }
    // the workgroup's region counts: 4 consecutive ones per thread, wave prefix sums, the waves' totals through LDS
    const uint32_t r0 = wg * regions_per_wg;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = threadIdx.x * 4 + k;
        c[k] = (i < regions_per_wg && r0 + i < n_regions) ? region_counts[r0 + i] : 0u;
        sum += c[k];
    }
    const uint32_t incl = pw_wave_scan(sum);
    below = pw_wave_scan(below);
    all = pw_wave_scan(all);
    if (lane == 63) {
        part[0][wave] = below;
        part[1][wave] = all;
        part[2][wave] = incl;
    }
    __syncthreads();
    const uint32_t base32 = part[0][0] + part[0][1] + part[0][2] + part[0][3];
    const uint32_t total = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    uint32_t ex = incl - sum;
    for (uint32_t w = 0; w < wave; ++w) ex += part[2][w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = threadIdx.x * 4 + k;
        if (i < kPermuteWgRegions) local[i] = ex;
        ex += c[k];
    }
    __syncthreads();
    if (first) { // (see PermuteTail)
        if (tail.zero_counters)
            for (uint32_t i = threadIdx.x; i < (uint32_t)kMaxSlices; i += blockDim.x) {
                tail.zero_counters[(size_t)i * kCounterStride] = 0;
                tail.zero_counters[(size_t)i * kCounterStride + 1] = 0;
                tail.zero_counters[(size_t)i * kCounterStride + 2] = 0; // (the fused tail's words: a workgroup's {done, records}, the start tickets)
                tail.zero_counters[(size_t)i * kCounterStride + 3] = 0;
            }
        if (tail.result && threadIdx.x == 0) {
            const uint32_t flag = tail.flag ? *tail.flag : 0u;
            tail.result[0] = total;
            tail.result[1] = flag;
            if (tail.d_result) {
                tail.d_result->n_records = total;
                tail.d_result->redone = flag;
                tail.d_result->reserved = 0;
            }
            if (tail.flag) *tail.flag = 0;
            __threadfence_system();
        }
    }
    const uint64_t base = base32;
    if (m > slice_slots) m = slice_slots; // overflow: the host redoes the call / reports ACGPU_E_OVERFLOW
    scratch = slice;
    for (uint64_t i = i_first; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 raw = i == i_first ? raw_first : *reinterpret_cast<const uint4 *>(&scratch[i]);
        const int32_t start = (int32_t)raw.x, end = (int32_t)raw.y, id = (int32_t)raw.z;
        const uint32_t rank = raw.w;
        if (rank == ~0u) continue; // hole left by a slot reservation
        const uint32_t chunk = ((uint32_t)(end - 1) - own_begin) / chunk_units; // the region owning the match's last unit
        const uint64_t dst = base + local[chunk - r0] + rank;
        if (dst >= cap) continue;
        if (REC == ACGPU_REC_SET) {
            reinterpret_cast<int2 *>(out)[dst] = make_int2(start, end);
        } else {
            int32_t *o = reinterpret_cast<int32_t *>(out) + dst * 3;
            o[0] = start;
            o[1] = end;
            o[2] = id_map ? (int32_t)id_map[id] : id;
        }
    }
}

hipError_t launch_permute_wg(const ScratchRec *d_scratch, const unsigned long long *d_counter, uint32_t n_wg, uint64_t slice_slots,
                             const uint32_t *d_region_counts, uint32_t n_regions, uint32_t regions_per_wg, uint32_t own_begin,
                             uint32_t chunk_units, int record_kind, void *d_out, uint64_t out_cap, const uint32_t *d_id_map,
                             hipStream_t stream, const PermuteTail *tail, hipEvent_t ev_stop) {
    const PermuteTail tl = tail ? *tail : PermuteTail{nullptr, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid(std::max<uint32_t>(kPermuteBlocks / std::max<uint32_t>(n_wg, 1u), 8u), n_wg);
    const hipEvent_t ev_none = nullptr;
    if (record_kind == ACGPU_REC_SET)
        ACGPU_LAUNCH_EV(k_permute_wg<ACGPU_REC_SET>, grid, dim3(256), 0, stream, ev_none, ev_stop, d_scratch, d_counter, n_wg, slice_slots,
                        d_region_counts, n_regions, regions_per_wg, own_begin, chunk_units, d_out, out_cap, d_id_map, tl);
    else
        ACGPU_LAUNCH_EV(k_permute_wg<ACGPU_REC_MAP>, grid, dim3(256), 0, stream, ev_none, ev_stop, d_scratch, d_counter, n_wg, slice_slots,
                        d_region_counts, n_regions, regions_per_wg, own_begin, chunk_units, d_out, out_cap, d_id_map, tl);
    return hipGetLastError();
}

uint32_t scan_tiles_for(uint32_t n) { return (n + kScanTile - 1) / kScanTile; }

__global__ void k_write_result(acgpu_device_result *r, unsigned long long n) {
    r->n_records = n;
    r->redone = 0;
    r->reserved = 0;
}

// the end of a chain pipeline (LONGEST): record count and chain exit from device memory into the call's pinned host slot
// {count, 0, exit} and, if wanted, the device result -- in stream order, no copy operations
__global__ void k_publish_result(const unsigned long long *d_total, const unsigned long long *d_exit, unsigned long long *h_slot,
                                 acgpu_device_result *r) {
    const unsigned long long n = *d_total;
    h_slot[1] = d_exit[1]; // (k_longest_bits: the bail flag; 0 otherwise -- the caller zeroes the block)
    h_slot[2] = *d_exit;
    h_slot[0] = n;
    if (r) {
        r->n_records = n;
        r->redone = d_exit[1] != 0; // (the records are not there yet: acgpu_match_device_end redoes the call)
        r->reserved = 0;
    }
}

hipError_t launch_publish_result(const unsigned long long *d_total, const unsigned long long *d_exit, unsigned long long *h_slot_dev,
                                 acgpu_device_result *d_result, hipStream_t stream) {
    hipLaunchKernelGGL(k_publish_result, dim3(1), dim3(1), 0, stream, d_total, d_exit, h_slot_dev, d_result);
    return hipGetLastError();
}

// acgpu_match_batch_u16: records of the scan over the concatenation -> records tagged with their haystack.  cat_off[i] = first
// unit of haystack i in the concatenation (one separator unit behind every haystack), cat_off[n_hay] = its length.
template <int REC>
__global__ __launch_bounds__(256) void k_batch_tag(const int32_t *recs, uint64_t n, const uint32_t *cat_off, uint32_t n_hay, int32_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = REC / 4;
    const int32_t start = recs[i * W], end = recs[i * W + 1];
    uint32_t lo = 0, hi = n_hay; // the last haystack that begins at or before start
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (cat_off[mid] <= (uint32_t)start) lo = mid;
        else hi = mid;
    }
    const int32_t base = (int32_t)cat_off[lo];
    int32_t *o = out + i * (W + 1);
    o[0] = (int32_t)lo;
    o[1] = start - base;
    o[2] = end - base;
    if (W == 3) o[3] = recs[i * W + 2];
}

hipError_t launch_batch_tag(const void *d_recs, uint64_t n, int record_kind, const uint32_t *d_cat_off, uint32_t n_hay, void *d_out,
                            hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (record_kind == ACGPU_REC_SET)
        hipLaunchKernelGGL(k_batch_tag<ACGPU_REC_SET>, grid, block, 0, stream, (const int32_t *)d_recs, n, d_cat_off, n_hay, (int32_t *)d_out);
    else
        hipLaunchKernelGGL(k_batch_tag<ACGPU_REC_MAP>, grid, block, 0, stream, (const int32_t *)d_recs, n, d_cat_off, n_hay, (int32_t *)d_out);
    return hipGetLastError();
}

hipError_t launch_write_result(acgpu_device_result *d_result, uint64_t n_records, hipStream_t stream) {
    hipLaunchKernelGGL(k_write_result, dim3(1), dim3(1), 0, stream, d_result, (unsigned long long)n_records);
    return hipGetLastError();
}

// ---- synthetic haystack generator (ahocorasick_amd/synth.py is its numpy twin) ---------------
struct SynthTable {
    uint16_t t[64];
    uint32_t len;
};

__global__ __launch_bounds__(256) void k_synth_fill(uint16_t *dst, uint64_t n, uint64_t start, uint64_t seed, SynthTable tab) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = seed + (start + i + 1) * 0x9E3779B97F4A7C15ull;
        uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const uint32_t k = (uint32_t)(((z >> 32) * (uint64_t)tab.len) >> 32);
        dst[i] = tab.t[k];
    }
}

hipError_t launch_synth_fill(uint16_t *d_dst, uint64_t n, uint64_t start, uint64_t seed, const uint16_t *table,
                             uint32_t table_len, hipStream_t stream) {
    if (table_len == 0 || table_len > 64) return hipErrorInvalidValue;
    SynthTable tab;
    for (uint32_t i = 0; i < 64; ++i) tab.t[i] = i < table_len ? table[i] : 0;
    tab.len = table_len;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_fill, dim3(4096), dim3(256), 0, stream, d_dst, n, start, seed, tab);
    return hipGetLastError();
}


// ---- token-stream generator (config 5's haystack; ahocorasick_amd/synth.py: token_stream_haystack is its numpy twin) --------
// Token t owns the draws 32 t .. 32 t + 31 of the SplitMix64 stream, so a token is a function of (seed, t) alone: draw 0: a
// dictionary word (0) or a random word; dictionary word: draw 1 its index, draws 2.. a case flip per unit (the first 24);
// random word: draw 1 the script, draw 2 the length 2..12, draws 3.. the units; draw 28: 1..3 separators, draws 29.. which.
// The haystack is the tokens one after the other, cut at n_units: lengths, a prefix sum, a fill.
struct TokenTables {
    const uint16_t *kw_units;
    const uint32_t *kw_off; // n_kw + 1
    uint32_t n_kw;
    const uint16_t *swapcase; // 65536 entries or nullptr
    const uint16_t *script_units; // the six script tables one after the other
    uint32_t script_off[7];
    uint16_t seps[6];
};
__device__ __forceinline__ uint64_t synth_draw(uint64_t seed, uint64_t idx) {
    uint64_t x = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t synth_bounded(uint64_t z, uint32_t k) { return (uint32_t)(((z >> 32) * (uint64_t)k) >> 32); }

__global__ __launch_bounds__(256) void k_token_lengths(TokenTables T, uint64_t seed, uint32_t n_tokens, uint32_t *len) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tokens) return;
    const uint64_t d0 = (uint64_t)t * 32;
    uint32_t wl;
    if (synth_bounded(synth_draw(seed, d0), 2) == 0 && T.n_kw) {
        const uint32_t wi = synth_bounded(synth_draw(seed, d0 + 1), T.n_kw);
        wl = T.kw_off[wi + 1] - T.kw_off[wi];
    } else {
        wl = 2 + synth_bounded(synth_draw(seed, d0 + 2), 11);
    }
    len[t] = wl + 1 + synth_bounded(synth_draw(seed, d0 + 28), 3);
}

__global__ __launch_bounds__(256) void k_token_fill(TokenTables T, uint64_t seed, uint32_t n_tokens, const uint64_t *start, uint16_t *dst,
                                                    uint64_t n_units) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tokens) return;
    uint64_t at = start[t];
    if (at >= n_units) return;
    const uint64_t d0 = (uint64_t)t * 32;
    auto put = [&](uint32_t u) {
        if (at < n_units) dst[at] = (uint16_t)u;
        ++at;
    };
    if (synth_bounded(synth_draw(seed, d0), 2) == 0 && T.n_kw) {
        const uint32_t wi = synth_bounded(synth_draw(seed, d0 + 1), T.n_kw);
        const uint32_t a = T.kw_off[wi], wl = T.kw_off[wi + 1] - a;
        for (uint32_t i = 0; i < wl; ++i) {
            uint32_t u = T.kw_units[a + i];
            if (T.swapcase && i < 24 && synth_bounded(synth_draw(seed, d0 + 2 + i), 2)) u = T.swapcase[u];
            put(u);
        }
    } else {
        const uint32_t sc = synth_bounded(synth_draw(seed, d0 + 1), 6);
        const uint32_t wl = 2 + synth_bounded(synth_draw(seed, d0 + 2), 11);
        const uint32_t a = T.script_off[sc], tl = T.script_off[sc + 1] - a;
        for (uint32_t i = 0; i < wl; ++i) put(T.script_units[a + synth_bounded(synth_draw(seed, d0 + 3 + i), tl)]);
    }
    const uint32_t ns = 1 + synth_bounded(synth_draw(seed, d0 + 28), 3);
    for (uint32_t i = 0; i < ns; ++i) put(T.seps[synth_bounded(synth_draw(seed, d0 + 29 + i), 6)]);
}

hipError_t launch_token_stream(uint16_t *d_dst, uint64_t n_units, uint64_t seed, const uint16_t *d_kw_units, const uint32_t *d_kw_off,
                               uint32_t n_kw, const uint16_t *d_swapcase, const uint16_t *d_script_units, const uint32_t *script_off,
                               const uint16_t *seps, uint32_t n_tokens, uint32_t *d_len, uint64_t *d_start, uint64_t *d_tmp,
                               hipStream_t stream) {
    TokenTables T;
    T.kw_units = d_kw_units; T.kw_off = d_kw_off; T.n_kw = n_kw; T.swapcase = d_swapcase; T.script_units = d_script_units;
    for (int i = 0; i < 7; ++i) T.script_off[i] = script_off[i];
    for (int i = 0; i < 6; ++i) T.seps[i] = seps[i];
    const dim3 grid((n_tokens + 255) / 256), block(256);
    hipLaunchKernelGGL(k_token_lengths, grid, block, 0, stream, T, seed, n_tokens, d_len);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = launch_exclusive_scan(d_len, n_tokens, d_start, d_tmp, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_token_fill, grid, block, 0, stream, T, seed, n_tokens, d_start, d_dst, n_units);
    return hipGetLastError();
}

// ---- attainable-bandwidth probe ---------------------------------------------------------------------------
// The read side of k_ac_tile and nothing else: one 1024-thread workgroup per CU, every wave owns a contiguous span and
// streams it as 4 KiB tiles, lane l holding the 64 consecutive bytes at l*64 (four 16-byte loads), the next tile's loads
// in flight while the current one is reduced.  What this kernel reaches is the ceiling of the tile kernels' access
// pattern on the box at hand (bench.py reports it as roofline.attainable next to the 8 TB/s spec peak).
__global__ __launch_bounds__(1024) void k_stream_probe(const uint4 *__restrict__ p, uint64_t n_tiles, unsigned *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x / 64);
    const uint64_t per_wave = (n_tiles + n_waves - 1) / n_waves;
    const uint64_t t0 = wave * per_wave, t1 = t0 + per_wave < n_tiles ? t0 + per_wave : n_tiles;
    if (t0 >= t1) return;
    uint32_t x = 0;
    uint4 cur[4], nxt[4];
    const uint4 *b = p + t0 * 256 + lane * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) nxt[u] = b[u];
    for (uint64_t t = t0; t < t1; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        const uint4 *nb = p + (t + 1 < t1 ? t + 1 : t) * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = nb[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) x ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
    }
    if (x == 0x12345678u) sink[0] = x;
}

// The fastest pure read found on this chip (tools/micro/stream_patterns.hip, pattern B2): groups of four 2 KiB tiles dealt
// round robin to the waves of 2048 small workgroups, 32 bytes per lane and tile, four tiles in flight -- 6.2-6.3 TB/s where the
// tile kernels' own pattern (k_stream_probe: one contiguous span per wave, 64 bytes per lane) reads at 5.4.  The ceiling
// bench.py reports as roofline.attainable.
__global__ __launch_bounds__(256) void k_stream_probe_best(const uint4 *__restrict__ p, uint64_t n_groups, unsigned *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x / 64);
    uint32_t x = 0;
    for (uint64_t g = wave; g < n_groups; g += n_waves) {
        uint4 buf[4][2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint4 *b = p + (g * 4 + d) * 128;
            buf[d][0] = b[lane * 2];
            buf[d][1] = b[lane * 2 + 1];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) x ^= buf[d][0].x ^ buf[d][0].y ^ buf[d][0].z ^ buf[d][0].w ^ buf[d][1].x ^ buf[d][1].y ^ buf[d][1].z ^ buf[d][1].w;
    }
    if (x == 0x12345678u) sink[0] = x;
}

hipError_t launch_stream_probe(const void *d_buf, uint64_t n_bytes, int n_cu, unsigned *d_sink, int pattern, hipStream_t stream) {
    if (pattern == 1) {
        const uint64_t n_groups = n_bytes / 8192;
        if (n_groups == 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_stream_probe_best, dim3(n_cu * 8), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(d_buf), n_groups, d_sink);
        return hipGetLastError();
    }
    const uint64_t n_tiles = n_bytes / 4096;
    if (n_tiles == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_stream_probe, dim3(n_cu), dim3(1024), 0, stream, reinterpret_cast<const uint4 *>(d_buf), n_tiles, d_sink);
    return hipGetLastError();
}

} // namespace acgpu
