// acgpu_small.h -- the one-launch form of acgpu_match_u16 for short haystacks (acgpu_small.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "acgpu_internal.h"

namespace acgpu {

constexpr uint32_t kSmallMaxUnits = 4096; // haystack units the one workgroup takes
constexpr uint32_t kSmallMaxRecs = 4096;  // occurrences it can order in LDS (more: the general path)
constexpr uint32_t kSmallMaxLen = 256;    // keywords beyond this make walks too long for one latency-bound workgroup

// one call: everything the kernel touches is host-mapped pinned memory (device pointers)
struct SmallCall {
    const uint16_t *hay;        // n_units units, padded to whole 8-byte groups
    uint32_t n_units;
    int record_kind;
    void *out;                  // min(cap, kSmallMaxRecs) records
    uint32_t cap;
    unsigned long long *status; // [0]: 0 -> 1 done / 2 not handled (the general path); [1]: record count
};

bool small_call_supported(const HostTables &t);
hipError_t launch_small(const DevTables &T, int mode, const SmallCall &c, hipStream_t stream);

} // namespace acgpu
