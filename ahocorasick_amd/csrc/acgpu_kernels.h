// acgpu_kernels.h -- launch wrappers of the gfx950 kernels (implemented in acgpu_kernels.hip).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include "acgpu_internal.h"

namespace acgpu {

// A match as the scan kernels emit it: 16 bytes, unordered in the scratch buffer.  `rank` is the
// record's ordinal among the records of its ordering unit ("chunk": a contiguous range of owned
// positions scanned by one lane), in reference emission order; the finalize pass turns
// (chunk, rank) into the record's final index through a prefix sum of per-chunk counts.
struct ScratchRec {
    int32_t start, end, id;
    uint32_t rank;
};

struct ScanLaunch {
    const uint16_t *d_hay;
    uint32_t n_units, own_begin, own_end;
    uint32_t chunk_units; // owned units per chunk (multiple of 8)
    uint32_t n_chunks;
    ScratchRec *d_scratch;
    uint64_t cap;                  // scratch capacity in records
    unsigned long long *d_counter; // total records emitted (may exceed cap: then nothing past cap is stored)
    uint32_t *d_chunk_counts;      // n_chunks entries
    int grid, block;
    size_t lds_bytes;
    uint32_t debug; // 1: the one-chain kernel (k_ac_scan_dense) where k_ac_dfa would run
};

// AC-all scan: dense (state x class table, hot rows in LDS) or sparse (hashed goto + fail links).
hipError_t launch_ac_scan(const DevTables &t, const ScanLaunch &l, hipStream_t stream, const char **kernel_name);

// Position-parallel AC-all scan (suffix K-gram filter in LDS + reversed-trie verification).  Ordering unit
// ("chunk" for the permute pass) = one wave region of region_units owned units.
// launch with the dispatch's own start / stop timestamps delivered to two events (either may be nullptr; both: a plain launch)
#define ACGPU_LAUNCH_EV(kernel, grid, block, lds, stream, ev0, ev1, ...)                                        \
    do {                                                                                                        \
        if ((ev0) != nullptr || (ev1) != nullptr)                                                               \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (ev0), (ev1), 0, __VA_ARGS__);              \
        else                                                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                  \
    } while (0)

struct TileLaunch {
    const uint16_t *d_hay;
    uint32_t n_units, own_begin, own_end;
    uint32_t region_units; // multiple of 2048 (the unrolled tile group); regions start at (own_begin & ~7) + r * region_units
    uint32_t n_regions;
    uint32_t regions_per_wave; // a wave scans this many consecutive regions
    ScratchRec *d_scratch;
    uint64_t cap;
    unsigned long long *d_counter;   // slot counters, one per slice, kCounterStride apart
    // Record slots come from n_slices equal slices of the scratch (slice = workgroup modulo n_slices), each with its own
    // counter: the waves of the grid reach their region seams -- and with them their slot reservations -- in lock step,
    // and thousands of atomics on ONE address are served at under 100 per microsecond.  A slice that fills up sets bit 1
    // of *d_overflow; the host then redoes the call with one slice.  n_slices = 1: the whole scratch, one counter.
    uint32_t n_slices, slice_slots;
    uint32_t wg_sums; // 1: every workgroup also adds its record count to d_counter[blockIdx.x * kCounterStride + 1] (the fused
                      // permute pass needs n_slices == grid: slice == workgroup)
    uint32_t *d_region_counts;
    int grid, block;
    size_t lds_bytes;
    // split form (filter kernel + verification kernel): the filter writes the candidate end positions of its span, in
    // text order, to its own slice of d_cands and a {first index, count} pair per region; the verification kernel
    // takes one region per wave.  *d_overflow becomes non-zero if a slice was too small (the call is then redone with
    // the fused kernel).
    uint32_t *d_cands;
    uint32_t cands_per_wave;
    uint2 *d_region_cands;
    uint32_t *d_overflow;
    int verify_grid;
    // WholeWord, region-local records: a word belongs to the region of its first unit and a region of R units holds at most
    // R/2 + 1 words, so every region owns region_cap 12-byte slots {start, end, id} of d_region_recs and a record goes
    // straight to slot region * region_cap + rank -- no slot reservations, no (region, rank) tags; k_ww_compact then copies
    // every region's records, whole and coalesced, to its final place (offsets = prefix sum of the region counts)
    // Profiled calls: the scan kernel's own dispatch timestamps instead of two hipEventRecord markers around it (host side only;
    // nullptr = a plain launch).  A marker is a packet of its own with a barrier: three of them cost 13-20 us per step at config 2.
    hipEvent_t ev_start, ev_stop;
    // Fused tail (k_ac_tile, the fused forms; `fused_tail` != 0): no finalize launch.  Workgroups take a number in the order in
    // which they start (d_counter[3] of the call's counter set; the number names the workgroup's span, slice and counters), so
    // every workgroup with a lower number is running or done.  When its waves have scanned their regions a workgroup publishes
    // {done, its records} (d_counter[number * kCounterStride + 2]), waits for the words of the workgroups before it -- their
    // sum is where its records begin -- and puts the records of its own scratch slice in their final place in d_out
    // (region offsets through LDS, which is free by then).  The workgroup with the last number knows the call's count: it
    // reports {count, overflow word} (tail_result / tail_d_result), clears the word and zeroes the other counter set.
    uint32_t fused_tail;
    void *d_out;
    uint64_t out_cap;
    int out_map;              // 1: {start, end, id} records (12 bytes), 0: {start, end}
    const uint32_t *d_id_map; // reversed-trie node -> keyword id (or nullptr)
    unsigned long long *tail_result;       // device-visible pinned host memory: [0] = count, [1] = overflow word
    acgpu_device_result *tail_d_result;    // acgpu_shard::d_result or nullptr
    unsigned long long *tail_zero_counters; // the counter set of the NEXT call (kMaxSlices x words 0..3), or nullptr
    // k_ww_pp with the fused tail: no regions -- wave w of workgroup b scans the tiles ww_ft_tiles_before(b) + w * q .. + q
    // (ft_total16 = tiles of the shard / 16, rounded up; ft_ramp_pm: see there), its records go to its own area of
    // d_region_recs (from record (first unit of the span - base8) / 2 + wave number on: a span of U units holds at most
    // U / 2 + 1 words) and, when the workgroups before it are done, from there to their final place
    uint32_t ft_total16, ft_ramp_pm;
    int32_t *d_region_recs; // nullptr: the scratch slices + k_permute
    uint32_t region_cap;
    unsigned long long *d_timing; // -DACGPU_TIMING builds only: 8 cycle counters per wave (tools/build_variant.sh timing)
    uint32_t debug; // ablation switches (tunable "tile_debug"): 1 = drop candidates unverified, 4 = no filter arithmetic
                    // at all (stream + reduce only), 8 = verification without the text-window load, 16 = without the
                    // K-gram node load, 32 = no record emission, 64 = no walk beyond the K-gram node, 128 = records
                    // not stored, 512 = one candidate in eight is kept, 4096 = the second-level filter passes everything.
                    // Kernel selection (results stay right): 1024 = scalar filter instead of the packed one, 2048 = no
                    // second-level filter, 16384 = one slot counter.  0 in production.
};
hipError_t launch_ac_tile(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name);
// split form: launch_ac_filter, then launch_ac_verify on the same stream (same TileLaunch)
hipError_t launch_ac_filter(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name);
hipError_t launch_ac_verify(const DevTables &t, const TileLaunch &l, hipStream_t stream);
bool tile_split_supported(const DevTables &t);
size_t tile_lds_bytes(const DevTables &t, int block_threads);
int tile_block_threads();

// exclusive prefix sum of d_counts[0..n) into d_offsets (uint64); d_tmp holds >= ceil(n/2048)+1 uint64
hipError_t launch_exclusive_scan(const uint32_t *d_counts, uint32_t n, uint64_t *d_offsets, uint64_t *d_tmp,
                                 hipStream_t stream);

constexpr uint32_t kCounterStride = 16; // in counters (128 bytes): one cache line per slice counter
constexpr int kMaxSlices = 512;

// What the permute pass -- the last kernel of a call -- does on the side, so that a call needs no copy and no memset
// operations of its own on the stream (each costs a launch gap): it reports {record count, overflow word} straight into
// pinned host memory, clears the overflow word, and zeroes the slot counters the NEXT call will use (the two sets of slot
// counters alternate).
struct PermuteTail {
    unsigned long long *result;       // device-visible pinned host memory: [0] = *total, [1] = *flag
    const uint64_t *total;
    uint32_t *flag;                   // read, reported, cleared
    unsigned long long *zero_counters; // kMaxSlices counters, kCounterStride apart, or nullptr
    acgpu_device_result *d_result;     // acgpu_shard::d_result (device memory) or nullptr: the same report, in stream order
};

// {n_records, redone = 0} into an acgpu_shard::d_result, in stream order (the families whose pipeline ends with a count
// on the host)
hipError_t launch_write_result(acgpu_device_result *d_result, uint64_t n_records, hipStream_t stream);
hipError_t launch_token_stream(uint16_t *d_dst, uint64_t n_units, uint64_t seed, const uint16_t *d_kw_units, const uint32_t *d_kw_off,
                               uint32_t n_kw, const uint16_t *d_swapcase, const uint16_t *d_script_units, const uint32_t *script_off,
                               const uint16_t *seps, uint32_t n_tokens, uint32_t *d_len, uint64_t *d_start, uint64_t *d_tmp,
                               hipStream_t stream);
hipError_t launch_batch_tag(const void *d_recs, uint64_t n, int record_kind, const uint32_t *d_cat_off, uint32_t n_hay, void *d_out,
                            hipStream_t stream);
hipError_t launch_publish_result(const unsigned long long *d_total, const unsigned long long *d_exit, unsigned long long *h_slot_dev,
                                 acgpu_device_result *d_result, hipStream_t stream);

// scratch (unordered) -> final records in reference order
// (slots whose rank is ~0u are holes left by slot reservations and are skipped)
hipError_t launch_ww_compact(const int32_t *d_region_recs, uint32_t region_cap, const uint32_t *d_region_counts, const uint64_t *d_offsets,
                             uint32_t n_regions, int record_kind, void *d_out, uint64_t out_cap, hipStream_t stream, const PermuteTail *tail,
                             hipEvent_t ev_stop = nullptr);
hipError_t launch_permute(const ScratchRec *d_scratch, const unsigned long long *d_counter, uint32_t n_slices, uint64_t slice_slots,
                          const uint64_t *d_offsets, uint32_t own_begin, uint32_t chunk_units, int by_start,
                          int record_kind, void *d_out, uint64_t out_cap, const uint32_t *d_id_map, hipStream_t stream,
                          const PermuteTail *tail = nullptr);
uint32_t scan_tiles_for(uint32_t n); // number of prefix-sum tiles; the grand total is d_tmp[scan_tiles_for(n)]
hipError_t launch_scan_tile_offsets(const uint32_t *d_counts, uint32_t n, uint64_t *d_tmp, hipStream_t stream); // tile level only
uint32_t scan_tile_elems();          // elements per prefix-sum tile

// Fused finalize of the tile kernel (one launch instead of prefix-sum kernels + permute): slice y of the scratch holds the
// records of workgroup y, whose regions are [y * regions_per_wg, ...); the block derives its offsets from the workgroup sums
// the scan kernel left next to the slot counters (d_counter[w * kCounterStride + 1]) and the region counts of its own
// workgroup.  Needs n_slices == number of workgroups and regions_per_wg <= kPermuteWgRegions.
constexpr uint32_t kPermuteWgRegions = 1024;
hipError_t launch_permute_wg(const ScratchRec *d_scratch, const unsigned long long *d_counter, uint32_t n_wg, uint64_t slice_slots,
                             const uint32_t *d_region_counts, uint32_t n_regions, uint32_t regions_per_wg, uint32_t own_begin,
                             uint32_t chunk_units, int record_kind, void *d_out, uint64_t out_cap, const uint32_t *d_id_map,
                             hipStream_t stream, const PermuteTail *tail,
                             hipEvent_t ev_stop = nullptr);
uint32_t tile_reserve_slots();
uint32_t tile_group_units(); // regions must hold whole tile groups

// pure read of n_bytes in the tile kernels' access pattern (see acgpu_stream_probe)
hipError_t launch_stream_probe(const void *d_buf, uint64_t n_bytes, int n_cu, unsigned *d_sink, int pattern, hipStream_t stream);

hipError_t launch_synth_fill(uint16_t *d_dst, uint64_t n, uint64_t start, uint64_t seed, const uint16_t *table,
                             uint32_t table_len, hipStream_t stream);

int scan_block_threads();
int scan_chains(const DevTables &t); // chunks per lane of the dense chunk scan: the host sizes the chunks for lanes x this
size_t scan_queue_bytes(int block_threads);

} // namespace acgpu

namespace acgpu {
// ---- LONGEST (leftmost-longest, non-overlapping) pipeline -------------------------------------------------
struct LongestScanLaunch {
    const uint16_t *d_hay;
    uint32_t n_units, own_begin, own_end;
    uint32_t chunk_units, n_chunks; // owned START positions per lane chunk (multiple of 8)
    void *d_len;                    // per unit of the buffer: length of the longest keyword starting there (u16 or u32)
    uint32_t *d_state;              // optional: automaton state per unit (for the keyword id), or nullptr
    uint32_t *d_blockmax;           // per 64 owned positions: max(p + max(L[p],1)) -- lets the chain kernels skip
    int len_bytes;                  // 2 or 4
    uint32_t lds_rows;              // trie rows staged in LDS
    int pairs;                      // 1: the lean range-class walk (k_longest_walk_range), 2: its work-list form
                                    // (k_longest_walk_list: one workgroup per CU, grid-stride over 1024-position chunks)
    int grid, block;
    size_t lds_bytes;
    uint16_t *d_len_big;            // len_bytes == 1: the lengths of 255 units and more (escape 255 in d_len), sparse
    // root-table form (k_longest_block, then k_longest_walk_list over the flagged chunks)
    const uint8_t *d_todo;          // k_longest_walk_list: only the chunks (1024 positions) flagged here; nullptr = all
    uint8_t *d_todo_w;              // k_longest_block: one flag per chunk
    uint32_t span_chunks;           // k_longest_block: chunks per wave (contiguous)
    uint32_t pages_bytes;           // k_longest_walk: DevTables::dfa_pages copied behind the LDS rows (0: classes from cls_lut in global memory)
    uint32_t debug;                 // ACGPU_ABLATION builds: timing experiments (results are wrong)
};
hipError_t launch_longest_block(const DevTables &t, const LongestScanLaunch &l, hipStream_t stream, const char **kernel_name);
size_t longest_block_lds_bytes();
uint32_t longest_block_max_rows(uint32_t n_cls);
hipError_t launch_longest_scan(const DevTables &t, const LongestScanLaunch &l, hipStream_t stream, const char **kernel_name);
size_t longest_list_lds_bytes(bool state);       // dynamic LDS of k_longest_walk_list (work lists)
uint32_t longest_list_max_rows(uint32_t n_cls, bool state); // trie rows its static LDS holds

struct LongestChainLaunch {
    const void *d_len;
    const uint16_t *d_len_big; // len_bytes == 1: see LongestScanLaunch
    const uint32_t *d_state; // or nullptr (Set records)
    const uint32_t *d_out_id;
    int len_bytes;
    uint32_t own_begin, own_end;
    const uint32_t *d_blockmax; // see LongestScanLaunch
    uint32_t entry;       // first greedy-chain position of this shard
    uint32_t tile_units;  // positions per lane
    uint32_t n_tiles;
    uint32_t max_len;
    uint32_t *d_counts;         // per tile
    const uint64_t *d_offsets;  // per tile (write pass)
    void *d_out;
    uint64_t cap;
    int record_kind;
    unsigned long long *d_exit; // first chain position >= own_end
    uint32_t len_units;         // entries of d_len that hold lengths (the chain passes through LDS read whole chunks)
    uint32_t *d_bits;           // one bit per buffer position: set by the count pass where a match is reported (zeroed by the
                                // caller), read by k_longest_emit; nullptr: the serial write pass is used instead
    uint32_t *d_ebits;          // optional second bitmap (zeroed by the caller): bit end-1 of every reported match.  Matches do
                                // not overlap, so the k-th set bit of d_bits and the k-th of d_ebits are one record and the
                                // emit pass needs no length lookups (16-bit lengths through k_longest_chain_lds only)
};
// k_longest_bits (acgpu_longest_bits.hip): Set records, two-letter alphabets in which every letter is a keyword
struct LongestBitsLaunch {
    const uint16_t *d_hay;
    uint32_t n_units, own_end, entry;
    uint32_t g0;          // entry & ~127: first position of segment 0 of region 0
    uint32_t n_regions;   // regions of longest_bits_region_units() positions from g0 on, up to own_end
    uint32_t runup;       // pass 1 follows a chain through this many positions in front of a segment (at most a segment)
    uint32_t max_len;
    void *d_out;          // acgpu_set_match records, in text order
    uint64_t cap;         // records d_out holds (further ones are counted, not stored)
    unsigned long long *d_exit;  // [0]: the chain's first position at or behind own_end, [1]: bail flag, [2]: the record count (zero at the start)
    unsigned long long *d_agg;   // per region: {published, matches}; d_blk, per 64 regions: {regions published, their matches} (zeroed by the caller)
    unsigned long long *d_blk;
    uint32_t *d_next;            // the next region to hand out (zeroed by the caller)
    uint32_t *d_marks, *d_xout;  // per region: its marks (2048 words) and its 64 segment exits, between its walk and its records
    uint32_t *d_text;            // Map records: per region its text bits (longest_bits_region_text_bytes()); nullptr: Set records
    uint32_t *d_pred, *d_true;   // per region: the entry it assumed, the exit it found
    int grid;
    uint32_t debug;              // ACGPU_ABLATION builds: timing experiments (results are wrong)
};
uint32_t longest_bits_region_units();
uint32_t longest_bits_seg_units();
size_t longest_bits_region_scratch_bytes(); // d_marks + d_xout, per region
size_t longest_bits_region_text_bytes();    // d_text, per region
// k_longest_bits, then k_longest_bits_finish: the seams between the regions checked, {count, bail flag, exit} to the pinned host
// slot and to d_result (may be nullptr), the call's state words (d_exit, d_agg, d_blk, d_next: one allocation) zeroed for the next call
hipError_t launch_longest_bits(const DevTables &t, const LongestBitsLaunch &l, unsigned long long *h_slot_dev, acgpu_device_result *d_result,
                               unsigned long long *d_state, uint32_t state_words, hipStream_t stream, hipEvent_t ev_start = nullptr,
                               hipEvent_t ev_mid = nullptr, hipEvent_t ev_stop = nullptr); // (profiled calls: the dispatches' own timestamps)
// k_longest_follow (acgpu_longest_follow.hip): any dense dictionary with range classes or small class pages; Set and Map records;
// fills what k_longest_emit_ends reads (both bitmaps, per tile the first chain position and the count, the node per match start)
struct LongestFollowLaunch {
    const uint16_t *d_hay;
    uint32_t n_units, own_end, entry;
    uint32_t g0;          // entry & ~31: first position of segment 0 of region 0
    uint32_t seg_log2;    // a lane's segment: 2^seg_log2 positions (10; 9 for texts that would leave half the chip's lanes without one)
    uint32_t n_regions;   // regions of 64 segments from g0 on, up to own_end
    uint32_t runup;       // pass 1 follows a chain through this many positions in front of a segment (at most a segment)
    uint32_t tile_log2;   // segments per tile of d_sync / d_counts (log2)
    uint32_t hot_rows;    // leading rows of the table kept in LDS
    uint32_t *d_bits, *d_ebits;  // bit p: a match starts at p / ends at p + 1; d_ebits zeroed by the caller (ends are merged with atomicOr)
    uint32_t *d_state;           // Map records: the trie node of the match that starts at p, or nullptr
    uint32_t *d_sync, *d_counts; // per tile: the chain's first position in it (~0u: none), its matches
    unsigned long long *d_exit;  // [0]: the chain's first position at or behind own_end, [1]: bail flag (zeroed by the caller)
    uint32_t *d_pred, *d_true;   // per region: the entry it assumed, the exit it found
    int grid;
};
uint32_t longest_follow_seg_units();   // the default segment (1024)
uint32_t longest_follow_lanes_per_cu();
uint32_t longest_follow_hot_rows(uint32_t n_cls, uint32_t n_states, uint32_t page_bytes); // 0: does not fit
hipError_t launch_longest_follow(const DevTables &t, const LongestFollowLaunch &l, bool range, bool state, hipStream_t stream);
hipError_t launch_longest_sync(const LongestChainLaunch &l, uint32_t *d_sync, hipStream_t stream);
hipError_t launch_longest_chain(const LongestChainLaunch &l, const uint32_t *d_sync, bool write_pass, hipStream_t stream);
// count / write pass with the lengths staged through LDS in chunks (16-bit lengths)
hipError_t launch_longest_chain_lds(const LongestChainLaunch &l, const uint32_t *d_sync, bool write_pass, hipStream_t stream);
// the records from the bitmap of the count pass (d_bits), one wave per chain segment
hipError_t launch_longest_emit(const LongestChainLaunch &l, const uint32_t *d_sync, hipStream_t stream);
} // namespace acgpu

namespace acgpu {
// ---- AhoCorasick over dictionaries that match nearly everywhere: states per unit, then the records (acgpu_states.hip) ----
struct AcStatesLaunch {
    const uint16_t *d_hay;
    uint32_t n_units, own_begin, own_end;
    uint32_t g0;          // own_begin & ~3: chunk k of k_ac_states covers [g0 + (k << chunk_log2), g0 + ((k + 1) << chunk_log2)), the record pass takes a chunk per wave
    uint32_t chunk_log2;  // k_ac_states: a lane's chunk (log2 units)
    uint32_t n_waves;     // k_ac_states: waves of 64 chunks from g0 on, up to own_end
    uint32_t halo;        // max_len - 1: a chunk's walk starts at the root this many units before it
    uint32_t hot_rows;    // leading rows of hy_dense kept in LDS
    uint32_t *d_state;    // per position: h-id of the automaton's state behind the unit | kHyOut, in the order [wave][group of 4 units][lane][4] (st_index)
    uint32_t n_chunks;
    uint32_t *d_counts;            // per chunk: its records (k_ac_states)
    const uint64_t *d_offsets;     // their exclusive prefix sums
    void *d_out;
    uint64_t cap;
    int grid;
};
uint32_t ac_states_chunk_units();
uint32_t ac_states_lanes_per_cu();
uint32_t ac_states_hot_rows(uint32_t n_cls, uint32_t n_dense, uint32_t page_bytes); // 0: does not fit
hipError_t launch_ac_states(const DevTables &t, const AcStatesLaunch &l, bool range, hipStream_t stream);
hipError_t launch_ac_states_out(const DevTables &t, const AcStatesLaunch &l, bool map, hipStream_t stream);
} // namespace acgpu

namespace acgpu {
// ---- WHOLEWORD pipeline (acgpu_wholeword.hip) -----------------------------------------------------------------
// ShortestMatch: greedy selection over the ordered all-matches list (acgpu_shortest.hip)
hipError_t launch_shortest_select(const int32_t *d_recs, uint32_t M, int64_t entry, uint32_t *d_nxt, uint32_t *d_tmp,
                                  uint32_t *d_mark, hipStream_t stream);
hipError_t launch_longest_select(const int32_t *d_recs, uint32_t M, int64_t entry, int64_t limit, uint32_t max_len,
                                 uint32_t *d_nxt, uint32_t *d_tmp, uint32_t *d_mark, hipStream_t stream);
hipError_t launch_chain_mark(uint32_t *d_nxt, uint32_t *d_tmp, uint32_t *d_mark, uint32_t M, hipStream_t stream);
hipError_t launch_shortest_emit(const int32_t *d_recs, uint32_t M, const uint32_t *d_mark, const uint64_t *d_offsets,
                                const uint64_t *d_total, int record_kind, void *d_out, uint64_t cap, int64_t entry,
                                unsigned long long *d_exit, hipStream_t stream);

// WholeWordLongest (acgpu_wwlongest.hip)
uint32_t wwl_tiles(uint32_t n_units);
hipError_t launch_wwl_starts(const DevTables &t, const uint16_t *d_hay, uint32_t n, int n_cu, bool fill, uint32_t *d_counts,
                             const uint64_t *d_offsets, uint32_t *d_rs, int text_begin, int start_behind,
                             hipStream_t stream);
hipError_t launch_wwl_walk(const DevTables &t, bool plain_words, const uint16_t *d_hay, uint32_t n, const uint32_t *d_rs, uint32_t M,
                           uint32_t *d_nxt, uint32_t *d_mark, int32_t *d_mend, int32_t *d_mid, uint32_t *d_stop, uint32_t entry, int n_cu,
                           hipStream_t stream);
hipError_t launch_wwl_select(const uint32_t *d_mark, const int32_t *d_mend, const uint32_t *d_rs, const uint32_t *d_nxt,
                             const uint32_t *d_stop, uint32_t *d_sel, uint32_t M, uint32_t own_begin, uint32_t own_end,
                             unsigned long long *d_exit, hipStream_t stream);
// chain marking in one pass: index jumps for the Longest chain kernels, and their bitmap back into the mark array
hipError_t launch_wwl_jumps(const uint32_t *d_nxt, const uint32_t *d_mark, uint32_t M, uint16_t *d_len16, uint32_t *d_blockmax,
                            unsigned long long *d_head, bool measure_max_jump, hipStream_t stream);
hipError_t launch_wwl_bits_to_mark(const uint32_t *d_bits, uint32_t M, uint32_t *d_mark, hipStream_t stream);
hipError_t launch_wwl_sequential(const DevTables &t, const uint16_t *d_hay, uint32_t len, void *d_out, uint64_t cap,
                                 int record_kind, unsigned long long *d_counter, hipStream_t stream);
// (d_tile_offsets: launch_scan_tile_offsets over d_sel)
hipError_t launch_wwl_emit(const uint32_t *d_rs, const uint32_t *d_sel, const int32_t *d_mend, const int32_t *d_mid,
                           const uint64_t *d_offsets, uint32_t M, int record_kind, void *d_out, uint64_t cap, hipStream_t stream);

uint32_t ww_fold_pages_in_lds(const DevTables &t); // 0: the fold table is not staged (case sensitive / too many pages)
size_t ww_lds_bytes(int block_threads, const DevTables &t);
int ww_blocks_per_cu();
hipError_t launch_ww_tile(const DevTables &t, const TileLaunch &l, hipStream_t stream, const char **kernel_name);
size_t ww_pp_lds_total(const DevTables &t, const TileLaunch &l, int block_threads); // k_ww_pp's LDS, static + dynamic, for a workgroup of that size
bool ww_pp_serves(const DevTables &t, const TileLaunch &l); // launch_ww_tile would take k_ww_pp (the kernel that has the fused tail)
hipError_t launch_ww_sequential(const DevTables &t, const uint16_t *d_hay, uint32_t len, void *d_out, uint64_t cap,
                                int record_kind, unsigned long long *d_counter, hipStream_t stream);
} // namespace acgpu
