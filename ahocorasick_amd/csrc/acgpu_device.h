// acgpu_device.h -- small wave64 device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include "acgpu_internal.h"

namespace acgpu {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t l = lane_id();
    return l == 0 ? 0ull : (~0ull >> (64 - l));
}

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const T o = __shfl_up(v, d);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

// hashed goto edge (state, folded unit) -> child, ~0u when absent; linear probing
__device__ __forceinline__ uint32_t hashed_goto(const uint64_t *hkeys, const uint32_t *hvals, uint32_t hmask, uint32_t s,
                                                uint32_t u) {
    const uint64_t key = edge_key(s, u);
    uint32_t slot = edge_hash(key) & hmask;
    for (;;) {
        const uint64_t k = hkeys[slot];
        if (k == key) return hvals[slot];
        if (k == kEmptyKey) return ~0u;
        slot = (slot + 1) & hmask;
    }
}

} // namespace acgpu
