// acgpu_internal.h -- shared between the host builder, the C ABI glue and the HIP kernels.
#pragma once
#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/acgpu.h"

namespace acgpu {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kRefHasChildren = 0x80000000u, kRefTerminal = 0x40000000u, kRefIdMask = 0x00ffffffu;
constexpr uint32_t kRefHintShift = 24, kRefHintMask = 0x3fu; // class+1 of an only child, 0 = no hint

// 64-bit finaliser (splitmix) used for the hashed goto edges; identical on host and device.
#if defined(__HIPCC__)
#define ACGPU_HD __host__ __device__
#else
#define ACGPU_HD
#endif
ACGPU_HD inline uint32_t edge_hash(uint64_t key) {
    key ^= key >> 30;
    key *= 0xBF58476D1CE4E5B9ull;
    key ^= key >> 27;
    key *= 0x94D049BB133111EBull;
    key ^= key >> 31;
    return (uint32_t)key;
}
ACGPU_HD inline uint64_t edge_key(uint32_t state, uint32_t unit) { return ((uint64_t)state << 16) | (uint64_t)unit; }

// WholeWord: hash of a whole folded word, identical on host and device.  The folded units are packed two per 32-bit
// word (zero padded); the hash runs over max(8, ceil(len/2)) of those words, then a murmur3 finaliser.  Slots are probed
// in aligned groups of 4 starting at the group of (hash & mask).
constexpr uint32_t kWwHashSeed = 0x811C9DC5u, kWwEmpty = 0xffffffffu;
ACGPU_HD inline uint32_t ww_hash_step(uint32_t h, uint32_t packed_units) { return h * 33u + packed_units; }
ACGPU_HD inline uint32_t ww_hash_final(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
ACGPU_HD inline uint32_t ww_tag(uint32_t h, uint32_t length) { return (h & 0xffffff00u) | (length < 255u ? length : 255u); }
constexpr uint32_t kWwInlineUnits = 12;
// second hash of a keyword (rotate-xor over the same packed words: linear over GF(2), where ww_hash_step is linear modulo
// 2^32 -- dictionaries over dense alphabets do hold triples of keywords with one ww_hash, hardly with both)
ACGPU_HD inline uint32_t ww_hash2_step(uint32_t g, uint32_t packed_units) { return ((g << 5) | (g >> 27)) ^ packed_units; }
ACGPU_HD inline uint32_t ww_slot1(uint32_t h, uint32_t mask) { return h & mask; }
ACGPU_HD inline uint32_t ww_slot2(uint32_t h, uint32_t g, uint32_t mask) {
    const uint32_t x = g ^ (h >> 16) ^ (g >> 13);
    const uint32_t s = ((x * 0x2C1B3C6Du) >> 11) & mask;
    return s != (h & mask) ? s : s ^ 1u;
}
// The perfect hash of the whole keywords ("hash and displace"): bucket = ww_ph_bucket(hash), slot = ww_ph_slot(hash2, hash,
// displacement of the bucket) -- every keyword in a slot of its own, found with ONE probe
ACGPU_HD inline uint32_t ww_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
ACGPU_HD inline uint32_t ww_ph_bucket(uint32_t h, uint32_t n_buckets) { return ww_mulhi(h, n_buckets); }
ACGPU_HD inline uint32_t ww_ph_slot(uint32_t g, uint32_t h, uint32_t d, uint32_t n_slots) {
    uint32_t t = (g ^ (h << 7)) + d * 0x9E3779B9u;
    t ^= t >> 15;
    t *= 0x2C1B3C6Du;
    t ^= t >> 13;
    return ww_mulhi(t, n_slots);
}
// Fused tail of k_ww_pp: the workgroups' spans, optionally growing with their number (so that a workgroup would find the ones
// before it done when its own scan ends and copy its records while the ones behind it still scan: ramp_pm, 0 in the
// product -- the copies are bound by memory bandwidth and fastest when all workgroups copy together).  Tiles (512 units) in front of
// workgroup b of G: 16 * floor(total16 * F(b / G)), F(x) = (1 - r/2) x + (r/2) x^2, r = ramp_pm / 1000 (the spans of the
// first and the last workgroup differ by r of the average); every wave of workgroup b takes (before(b + 1) - before(b)) / 16.
ACGPU_HD inline uint32_t ww_ft_tiles_before(uint32_t b, uint32_t G, uint32_t total16, uint32_t ramp_pm, uint32_t waves = 16u) {
    // (total16: the shard's tiles / waves of a workgroup, rounded up -- 16 waves unless the launch says otherwise)
    const uint64_t num = (uint64_t)b * ((uint64_t)(2000u - ramp_pm) * G + (uint64_t)ramp_pm * b); // <= 2000 G^2
    const uint64_t den = 2000ull * G * G;
    return b >= G ? waves * total16 : waves * (uint32_t)(((uint64_t)total16 * num) / den);
}
// two bit positions of the Bloom filter in front of the table (the filter sits in LDS)
ACGPU_HD inline uint32_t ww_bloom_bit1(uint32_t h, uint32_t mask) { return (h >> 7) & mask; }
ACGPU_HD inline uint32_t ww_bloom_bit2(uint32_t h, uint32_t mask) { return ((h >> 19) | (h << 13)) & mask; }

// Second-level filter of the tile kernel (L2): a blocked Bloom filter, in LDS, over the class sequence of the last
// D = min(K+2, 6) units of every keyword (keywords shorter than D: their whole class sequence, under their length), so
// that a position that passed the K-gram filter is only handed to the gather-bound verification when its last D units
// still look like a keyword.  c[j] = class (below 32) of text[e-1-j].  The WORD is chosen by the K-gram alone (one LDS
// read per candidate); a key of length L in [K, D] sets / tests a two-bit pattern of that word, rotated by an amount
// that depends on L and on the L-K classes in front of the K-gram.  Full-rate 24-bit multiplies only.  Identical on
// host and device.
constexpr uint32_t kL2Words = 5760; // 22.5 KiB of LDS (what is left next to the rows, the queues and the tile copies)
ACGPU_HD inline uint32_t l2_mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); } // v_mul_u32_u24
ACGPU_HD inline uint32_t l2_gram(const uint32_t *c, uint32_t K) { // the K-gram, one byte per class, text[e-1] highest
    uint32_t g = 0;
    for (uint32_t j = 0; j < K && j < 4; ++j) g |= c[j] << (8u * (3u - j));
    if (K == 5) g |= ((c[4] & 7u) << 5) | ((c[4] >> 3) << 13); // (the classes leave bits 5-7 of every byte free)
    return g;
}
ACGPU_HD inline uint32_t l2_hash(uint32_t gram) { return l2_mul24(gram ^ (gram >> 13), 0x9E3779u); }
ACGPU_HD inline uint32_t l2_word(uint32_t h) { return (((h >> 11) & 0x1fffu) * kL2Words) >> 13; }
// the LARGE form of the same filter (dictionaries that saturate 22.5 KB): 2 MB in global memory, resident in the L2 cache
constexpr uint32_t kDfaLdsBytes = 127 * 1024; // k_ac_dfa: static LDS for the hot rows (next to 32 KB of record queues)
constexpr uint32_t kL2BigWords = 1u << 19;
ACGPU_HD inline uint32_t l2_word_big(uint32_t h) { return (l2_mul24(h ^ (h >> 11), 0x85EBCBu) >> 4) & (kL2BigWords - 1u); }
ACGPU_HD inline uint32_t l2_pattern(uint32_t h) {
    const uint32_t x = l2_mul24(h >> 8, 0xC2B2AFu);
#if defined(ACGPU_L2_TWO_BITS)
    return (1u << (x >> 27)) | (1u << ((x >> 22) & 31u));
#elif defined(ACGPU_L2_FOUR_BITS)
    return (1u << (x >> 27)) | (1u << ((x >> 22) & 31u)) | (1u << ((x >> 17) & 31u)) | (1u << ((x >> 12) & 31u));
#else
    return (1u << (x >> 27)) | (1u << ((x >> 22) & 31u)) | (1u << ((x >> 17) & 31u)); // three bits: fewer false survivors
#endif
}
ACGPU_HD inline uint32_t l2_rot(const uint32_t *c, uint32_t L, uint32_t K) { // rotation of the pattern for a key of length L
    if (L == K) return 0;
    if (L == K + 1) return (c[K] * 5u + 11u) & 31u;
    return (c[K] * 5u + c[K + 1] * 9u + 23u) & 31u;
}
ACGPU_HD inline uint32_t l2_rotr(uint32_t x, uint32_t r) { return (x >> (r & 31u)) | (x << ((32u - r) & 31u)); }

// k_longest_bits (acgpu_longest_bits.hip) and its table (acgpu_build.cpp 6c).  An entry is four words {label, terminal bits,
// meta, next}; meta = label length (bits 0-5) | kind (bits 6-7) | best (bits 8-15: a first-level entry's longest keyword
// among the RK units, a junction child's "the child ends a keyword") | bits 16-21: the label's length if the trie goes on
// behind the label, else 63 | kBitsAlive.
constexpr uint32_t kBitsRK = 9;            // first level: 2^9 entries indexed by the text's next 9 units
constexpr uint32_t kBitsTabEntries = 1120; // 17.5 KiB of LDS: what 16 waves' text images leave
constexpr uint32_t kHyOut = 1u << 23;   // HostTables::hy_dense / hy_nodes: the target state reports matches
constexpr uint32_t kHyIdMask = kHyOut - 1u;
constexpr uint32_t kHyDenseCountShift = 24;   // hy_dense: bits 24..29 = how many keywords the target state reports
constexpr uint32_t kHyNodeCountShift = 23;    // hy_nodes word 0 (the fail state's h-id in bits 0..22): 3 bits per edge, the same number ...
constexpr uint32_t kHyNodeCountMany = 7u;     // ... or this: seven and more (hy_mask says how many)
constexpr uint32_t kBitsAlive = 1u << 24;
ACGPU_HD inline uint32_t bits_id_hash(uint64_t key) { // (key = length << 32 | bits: see HostTables::bits_idkeys)
    key ^= key >> 29;
    key *= 0xBF58476D1CE4E5B9ull;
    key ^= key >> 32;
    return (uint32_t)key;
}
constexpr uint32_t kBitsLeaf = 0, kBitsJunction = 1, kBitsCont = 2, kBitsDeep = 3;

// Host-side automaton tables.  State numbering: root = 0; states WITHOUT any output (own or inherited
// keyword) come first in BFS order, states WITH output after them in BFS order, so that
// "state >= first_out" is the has-output test and a prefix of the numbering is the shallow, hot part.
struct HostTables {
    int mode = 0;
    bool cs = true;
    uint32_t n_states = 1, n_cls = 1, n_kw = 0, min_len = 0, max_len = 0;
    uint32_t first_out = 1;
    int32_t sep_unit = -1; // a unit no match, word or walk reaches across (batches of haystacks); -1: every unit is in use
    // character classes: class 0 = unit occurs in no keyword.  range_cls (case-sensitive only):
    // class = unit - cls_base + 1 for unit in [cls_base, cls_base + cls_span), computed arithmetically.
    bool range_cls = false;
    uint32_t cls_base = 0, cls_span = 0;
    std::vector<uint16_t> cls_lut; // 65536: raw unit -> class of its folded unit (dense tables only; a dictionary
                                   // using > 65535 distinct units is always built sparse and needs no classes)
    std::vector<uint16_t> lower;   // 65536 fold table (identity when case sensitive)
    std::vector<uint8_t> wflags;   // WHOLEWORD: bit0 = word[raw], bit1 = word[lower[raw]]
    std::vector<uint32_t> wbits;   // bit0 of wflags packed 32 units per word (2048 words: what the kernels keep in LDS)
    bool fold_consistent = true;
    // Word-character tables that are NOT fold-consistent (custom tables, case-insensitive): the reference loops that fold
    // in EVERY lookup -- match(Readable) of the word matchers (S/WholeWordMatchMap.java:112,117,328,
    // S/WholeWordLongestMatchMap.java:404) and WholeWordLongestMatchMap.match(String) (:252,258,283,288) -- are ordinary
    // scans over w'[c] = word[lower[c]]: the same kernels with these tables in place of wflags / wbits.
    std::vector<uint8_t> wflags_f;  // bit0 = bit1 = word[lower[raw]]
    std::vector<uint32_t> wbits_f;  // bit0 of wflags_f packed
    bool fold_clean = true;         // WHOLEWORD: every FOLDED keyword unit is a word character (keywords are validated on
                                    // their raw units, S/WholeWordMatchMap.java:263-267); if not, a keyword holds units that
                                    // are no word characters to a folding scan, which then has to walk through them
    // per state
    std::vector<uint32_t> depth, fail, out_len, out_link, out_id, term_id; // term_id: own keyword id or ~0u
    // dense delta (AC semantics, fail transitions resolved): n_states * n_cls entries
    bool dense = false;
    uint32_t entry_bytes = 4;
    std::vector<uint32_t> dfa;
    // LONGEST, small alphabets: root table of the walk's first round (acgpu_build.cpp 6b); root_b = 0: none
    std::vector<uint8_t> root_tab;
    uint32_t root_b = 0, root_rk = 0;
    // LONGEST over a two-letter alphabet in which every letter is a keyword: the path-compressed trie of k_longest_bits
    // (acgpu_build.cpp 6c): kBitsTabEntries entries of four words; bits_rk = 0: none
    std::vector<uint32_t> bits_tab;
    uint32_t bits_rk = 0;
    // ... and, for Map records, the keyword id of every keyword of at most 32 units by its text: key = length << 32 | the
    // keyword as bits (unit i at bit i), open addressing (bits_id_hash), at most half full.  Longer keywords: the kernel walks
    // the table in global memory for their node.
    std::vector<uint64_t> bits_idkeys; // two words per slot: {key, keyword id} -- one 16-byte gather answers a probe
    uint32_t bits_idmask = 0;
    // ALL / SHORTEST: the automaton in the form k_ac_states walks (acgpu_build.cpp 6d; hy_n_states = 0: none).  States are numbered
    // anew ("h-ids", 23 bits): the DENSE group first -- the root, depth 1 and 2, every state with more than three children -- in
    // BFS order, each with a row of n_cls resolved transitions (fail links followed at build time); then the COMPACT group, one
    // 16-byte node each: {fail state | counts, three edges (class << 24 | child)} -- a miss goes to the fail state and looks at the same
    // unit again.  Bit 23 (kHyOut) of a transition = the state it leads to reports matches.
    std::vector<uint32_t> hy_dense;  // hy_n_dense * n_cls
    std::vector<uint32_t> hy_nodes;  // 4 words per compact state
    std::vector<uint32_t> hy_mask;   // per h-id: bit L-1 = a keyword of L units ends in this state (its own or a suffix's)
    std::vector<uint32_t> hy_out;    // per h-id, two words: {hy_mask again, where the state's keyword ids begin in hy_ids -- or, bit 31 set, the id of its only keyword} (Map records: one gather for both)
    std::vector<uint32_t> hy_ids;    // the keyword ids of what a state reports, longest first, state after state
    uint32_t hy_n_dense = 0, hy_n_states = 0;
    // hashed goto edges keyed by (state, folded unit): open addressing, linear probing
    std::vector<uint64_t> hkeys;
    std::vector<uint32_t> hvals;
    uint32_t hmask = 0;
    uint64_t n_edges = 0;
    // ---- suffix k-gram filter + reversed trie (ALL mode, position-parallel kernel) ----
    // A match can end at text position e only if the K units before e form the K-suffix of some keyword
    // (K <= min keyword length).  filt_bits has one bit per K-gram of tile classes (direct index, last unit least
    // significant); kgram_node maps a set K-gram to the depth-K node of the trie of REVERSED keywords, from which the
    // verification walk continues leftwards through the hashed reverse edges.
    uint32_t filt_k = 0;      // 0 = filter not available for this dictionary
    uint32_t filt_n = 0;      // tile classes (radix of the K-gram index)
    uint32_t filt_other = 0;  // tile class of a unit that occurs in no keyword
    double filt_density = 0;  // set bits / (filt_n-1)^K
    std::vector<uint32_t> filt_bits;   // filt_n^(K-1) rows of filt_row_bytes
    // Reverse-trie node references carry flags so that the common verification walk needs no per-node load:
    // bit31 = node has children, bit30 = node is terminal (a keyword starts here), bits 29..24 = only-child class hint,
    // low 24 bits = node id.
    uint32_t filt_row_bytes = 4;
    std::vector<uint32_t> kgram_node;  // filt_n^K pairs {flagged ref of the depth-K node (0 = none), flagged ref of its
                                       // only child (0 = none or several)}: the common one-step walk needs no second load
    std::vector<uint32_t> rterm;       // per reverse node: keyword id (valid when the terminal flag is set)
    bool has_short = false;            // keywords of fewer than filt_k units exist (acgpu_build.cpp 7, "Short keywords")
    std::vector<uint32_t> kshort;      // filt_n^(filt_k-1) entries of 4 words: node + 1 of the keyword of 1, 2, 3 units that ends here
    std::vector<uint64_t> ks_keys;     // bucketed / merged classes: the same by units -- (length << 48 | folded units) -> node + 1
    std::vector<uint32_t> ks_vals;
    uint32_t ks_mask = 0;
    bool rdense = false;
    std::vector<uint32_t> rtab;        // dense: n_rstates * filt_n flagged child refs indexed by tile class (0 = none)
    std::vector<uint64_t> rhkeys;      // hashed: (node, folded unit) -> flagged child ref
    std::vector<uint32_t> rhvals;
    uint32_t rhmask = 0;
    uint32_t n_rstates = 0;
    // Dictionaries with more than 63 distinct (folded) units: the tile classes are 63 BUCKETS of units (tile_lut), so the
    // filter is a superset test and a class K-gram no longer names one K-gram of units.  The verification then looks the
    // K units themselves up (kg_keys/kg_vals: packed folded units -> flagged ref of the depth-K reverse node, open
    // addressing) and walks the reversed trie through its hashed, unit-keyed edges only.
    // Folded range classes (tile kernel, case-insensitive dictionaries whose folded keyword units span at most 31 code
    // points): tile class = lower[unit] - fr_base inside the range, fr_span outside ("other").  The raw units that fold
    // into the range are the range itself, a partner range of the same length (fr_base2: the other case, every unit of
    // it folds to its opposite number) and a few exceptions (U+0130 -> i, U+212A -> k ...), all of which have a bit of
    // fr_himask set -- so the packed filter computes classes from the two ranges and takes the class table (tile_lut)
    // for a tile only when some unit of it has such a bit.
    // fold_range together with hashk: MERGED stretches (acgpu_build.cpp 7; keywords in mixed case, phrases with spaces and
    // digits, case-sensitive or not): the same filter arithmetic over up to four ranges (fr_base .. fr_base4, fr_nr of them in
    // use), units of different stretches sharing classes, the verification by the units themselves as for bucketed classes.
    bool fold_range = false;
    uint32_t fr_base = 0, fr_span = 0, fr_base2 = 0, fr_himask = 0;
    uint32_t fr_base3 = 0, fr_base4 = 0, fr_nr = 2; // merged stretches: up to four ranges (unused ones repeat fr_base)
    // second-level (Bloom) filter of the tile kernel: see l2_gram; l2_depth = D, 0 = not built
    uint32_t l2_depth = 0;
    std::vector<uint32_t> l2_bloom; // kL2Words
    double l2_density = 0;          // fraction of set bits
    std::vector<uint32_t> l2_big;   // kL2BigWords: the same keys in the large form; empty unless the small one is saturated
    bool hashk = false;
    std::vector<uint16_t> tile_lut;    // 65536: raw unit -> tile class (== cls_lut when the classes are injective)
    std::vector<uint8_t> cls_pages;    // the same table as [256-byte page index][distinct 256-byte pages] (acgpu_build.cpp 7b); empty: none
    std::vector<uint16_t> dfa_pages;   // cls_lut as [256-byte page index][distinct pages of 256 uint16] (7c: k_ac_dfa); empty: none
    std::vector<uint64_t> kg_keys;
    std::vector<uint32_t> kg_vals;
    uint32_t kg_mask = 0;
    // ---- WholeWord: hash table of whole (folded) keywords + paged fold table ----
    // A maximal run of word characters matches iff its folded text IS a keyword, so the run is hashed once and looked up.
    // ww_recs = records {u32 keyword id, u32 length, folded units packed 2 per u32, zero padded}, each 16-byte aligned
    // (what a keyword of more than 12 units is compared with, unit for unit).
    std::vector<uint32_t> ww_recs;
    // What the kernel probes: a two-choice (cuckoo) table with the keyword INLINE, 32 bytes per slot -- {tag, id | record
    // offset, folded units 0..11 packed 2 per u32}; tag = hash with its low byte replaced by min(length, 255) (0 = free
    // slot), so a tag match fixes the length.  A keyword sits in slot ww_slot1(hash) or ww_slot2(hash, hash2): a lookup gathers
    // both at once and a word of up to 12 units is decided by ONE round of memory accesses, for every lane alike (linear
    // probing made a wave wait for its unluckiest lane's probe sequence).  Longer keywords: the record, whose offset
    // takes the id's place, is compared as well.
    std::vector<uint32_t> ww_fat;   // 8 u32 per slot
    uint32_t ww_fat_mask = 0;
    // The same slots behind a PERFECT hash (WHOLEWORD, keywords of at most 32 units: k_ww_pp): ww_ph_n slots for the keywords at
    // a load of 0.97 -- 3.3 MB for 100 k keywords where the two-choice table has 8.4 MB, so it stays in an XCD's L2 cache --
    // and one 16-bit displacement per bucket of about four keywords (50 KB: the kernel keeps them in LDS where the Bloom filter
    // was).  A run of word characters reads ONE slot; a run that is no keyword reads some keyword's slot and fails the compare.
    // Empty: not built (too many keywords for the displacements' LDS, or two keywords with both hashes equal).
    std::vector<uint32_t> ww_ph;        // 8 u32 per slot
    std::vector<uint16_t> ww_ph_disp;   // per bucket; padded to a multiple of 8 entries
    uint32_t ww_ph_n = 0, ww_ph_buckets = 0;
    uint32_t ww_seed = kWwHashSeed; // start value of both hashes (another one if the two-choice table cannot be built)
    // fold table as pages of 256 deltas (lower[u] - u mod 2^16), identical pages shared: fits LDS for real tables
    std::vector<uint8_t> fold_pgidx;   // 256
    std::vector<uint16_t> fold_pages;  // fold_n_pages * 256
    uint32_t fold_n_pages = 0;
    uint32_t fold_direct_n = 0;        // units below this fold through a direct table (the cased scripts of the low pages)
    // Word-character bit and fold of a unit in ONE page entry (k_ww_pp, case-insensitive, fold-consistent tables): pages of 256
    // bytes {index of the unit's fold delta << 1 | word character}, identical pages shared, and the table of the (at most 128)
    // distinct deltas.  Unicode's simple lower-casing with Character.isLetterOrDigit: 79 deltas, 54 pages = 13.5 KB where the
    // word bits (8 KB) and the delta pages (9 KB) took two lookups with their own address arithmetic.  Empty: more deltas or
    // pages than that (the two tables serve).
    std::vector<uint8_t> ww_bp_idx;    // 256: page of the units u >> 8
    std::vector<uint8_t> ww_bp_pages;  // ww_bp_n * 256
    std::vector<uint16_t> ww_bp_delta; // 128
    uint32_t ww_bp_n = 0;
    std::vector<uint32_t> ww_bloom;    // 2-hash Bloom filter over the keyword hashes, a power of two of bits (<= 64 KB)
    uint32_t ww_bloom_mask = 0;        // bits - 1
};

int build_tables(int mode, const uint16_t *kw_units, const uint64_t *kw_off, uint32_t n_kw, int case_sensitive,
                 const uint16_t *lower_tbl, const uint8_t *wordchar_tbl, HostTables &t, int64_t *bad_keyword);

// Device view handed to kernels by value.
struct DevTables {
    const uint16_t *cls_lut;
    const uint16_t *lower;
    const uint8_t *wflags;
    const uint32_t *wbits;
    const void *dfa; // uint16_t or uint32_t entries
    const uint32_t *out_len, *out_link, *out_id, *fail, *depth, *term_id;
    const uint64_t *hkeys;
    const uint32_t *hvals;
    uint32_t hmask;
    uint32_t n_states, n_cls, first_out, max_len, min_len;
    uint32_t cls_base, cls_span;
    int32_t range_cls, cs, dense, entry_bytes;
    uint32_t lds_entries; // leading dfa entries staged in LDS by the scan kernel
    // k-gram filter / reversed trie
    const uint32_t *filt_bits, *kgram_node, *rterm, *rtab, *kshort; // kshort: nullptr = no keyword is shorter than filt_k (or hashk)
    const uint64_t *ks_keys;                                        // hashk: short keywords by units (nullptr: none)
    const uint32_t *ks_vals;
    uint32_t ks_mask, has_short;
    const uint64_t *rhkeys;
    const uint32_t *rhvals;
    uint32_t rhmask, filt_k, filt_n, filt_other, filt_words, filt_row_bytes;
    int32_t rdense;
    int32_t hashk;             // 1: bucketed tile classes, K-gram looked up by its units (see HostTables)
    const uint32_t *l2_bloom;  // second-level filter (kL2Words words) or nullptr
    const uint32_t *l2_big;    // its large form in global memory (kL2BigWords words) or nullptr
    uint32_t l2_depth;
    uint32_t fold_range, fr_base, fr_span, fr_base2, fr_himask; // see HostTables::fold_range
    uint32_t fr_base3, fr_base4, fr_nr;
    const uint16_t *tile_lut;  // tile classes of the LUT mode (cls_lut, or the bucket table)
    const uint8_t *cls_pages;  // HostTables::cls_pages (nullptr: none), cls_pages_bytes a multiple of 256
    uint32_t cls_pages_bytes;
    const uint16_t *dfa_pages; // HostTables::dfa_pages (nullptr: none)
    uint32_t dfa_pages_bytes;
    const uint64_t *kg_keys;
    const uint32_t *kg_vals;
    uint32_t kg_mask;
    // WholeWord word hash
    const uint32_t *ww_fat;   // 8 u32 per slot (see HostTables::ww_fat)
    uint32_t ww_fat_mask, ww_seed;
    const uint32_t *ww_ph;    // HostTables::ww_ph (nullptr: none)
    const uint16_t *ww_ph_disp;
    uint32_t ww_ph_n, ww_ph_buckets;
    const uint32_t *ww_recs;  // 16-byte aligned records
    const uint8_t *fold_pgidx;
    const uint16_t *fold_pages;
    uint32_t fold_n_pages, fold_direct_n;
    const uint8_t *ww_bp_idx, *ww_bp_pages; // HostTables::ww_bp_* (nullptr: none)
    const uint16_t *ww_bp_delta;
    uint32_t ww_bp_n;
    const uint32_t *ww_bp_wbits;            // the word bits the byte pages hold (a scan over other ones -- folded_tables -- does not take them)
    const uint32_t *ww_bloom;
    uint32_t ww_bloom_mask;
    const uint8_t *root_tab; // LONGEST: see HostTables::root_tab
    uint32_t root_b, root_rk;
    const uint32_t *bits_tab; // LONGEST: see HostTables::bits_tab (nullptr: none)
    uint32_t bits_rk;
    const uint64_t *bits_idkeys; // HostTables::bits_idkeys (nullptr: none)
    uint32_t bits_idmask;
    const uint32_t *hy_dense, *hy_nodes, *hy_mask, *hy_out, *hy_ids; // see HostTables::hy_dense (hy_n_states = 0: none)
    uint32_t hy_n_dense, hy_n_states;
};

// development/test knobs (acgpu_set_tunable): relaxed atomics, read when a call is enqueued
struct Tunables {
    std::atomic<int64_t> chunk_units{0};      // 0 = auto
    std::atomic<int64_t> blocks_per_cu{1};
    std::atomic<int64_t> lds_table_bytes{127 * 1024};
    std::atomic<int64_t> force_sparse{0};
    std::atomic<int64_t> dense_budget_bytes{1ll << 30};
    std::atomic<int64_t> force_kernel{0};     // 0 auto, 1 = DFA chunk scan, 2 = k-gram tile scan (when the filter exists)
    std::atomic<int64_t> region_units{0};     // tile kernel: owned units per wave region (0 = auto)
    std::atomic<int64_t> rdense_budget_bytes{256ll << 20};
    std::atomic<int64_t> tile_debug{0};       // ablation switches of the tile kernel (see TileLaunch::debug); 0 in production
#ifndef ACGPU_FILTER_MAX_BYTES
#define ACGPU_FILTER_MAX_BYTES 88000
#endif
    std::atomic<int64_t> filter_max_bytes{ACGPU_FILTER_MAX_BYTES};  // the filter rows must fit LDS next to the candidate queues
    std::atomic<int64_t> no_short_keywords{0}; // builder: 1 = the filter's K stays at most the shortest keyword (A/B)
    std::atomic<int64_t> no_merged_ranges{0}; // builder: 1 = mixed-case dictionaries keep the 8-byte-row scalar filter (A/B)
    std::atomic<int64_t> ww_block{0};         // k_ww_pp, fused tail: threads per workgroup (0 = 1024; a multiple of 64 -- two workgroups share a CU when their LDS allows: A/B)
    std::atomic<int64_t> ww_ramp_pm{-1};      // k_ww_pp, fused tail: the spans' ramp in thousandths of the average span (-1 = the default)
    std::atomic<int64_t> ww_no_byte_pages{0}; // WHOLEWORD builder: 1 = no combined word-bit / fold pages (A/B, tests)
    std::atomic<int64_t> ww_no_ph{0};         // WHOLEWORD builder: 1 = no perfect hash (the two-choice table behind the Bloom filter: A/B, tests)
    std::atomic<int64_t> ww_ph_lambda{0};     // WHOLEWORD builder: keywords per bucket of the perfect hash (0 = 4; tests: large buckets, many displacements tried)
    std::atomic<int64_t> ww_no_bloom{0};      // WHOLEWORD builder: 1 = no Bloom filter in front of the keyword table (every run of keyword length probes it)
    std::atomic<int64_t> ww_first_seed{0};    // WHOLEWORD builder: first hash seed tried (tests: the fallback seeds end to end)
    std::atomic<int64_t> split_cand_div{8};   // split form: a wave's candidate slice holds one candidate per this many units of its span
    std::atomic<int64_t> no_class_pages{0};   // builder: 1 = the tile kernel's LUT forms look classes up in global memory (A/B)
    std::atomic<int64_t> no_big_l2{0};        // builder: 1 = large dictionaries keep the (saturated) second level in LDS (A/B)
    std::atomic<int64_t> multi_min_share{1ll << 22}; // acgpu_match_u16_multi: a share is at least this many units (tests: 1024)
    std::atomic<int64_t> longest_form{0};     // LONGEST, bits: 1 = never k_longest_bits, 2 = never k_longest_follow, 4 = both also for short texts (tests, A/B)
    std::atomic<int64_t> all_form{0};         // ALL, bits: 1 = never k_ac_states, 2 = always (where its tables exist), 4 = also for short texts (tests, A/B)
    std::atomic<int64_t> no_state_form{0};    // builder: 1 = no compact automaton for k_ac_states (A/B)
    std::atomic<int64_t> no_bits_trie{0};     // builder: 1 = no path-compressed trie for k_longest_bits (the walk pipeline instead: A/B)
    std::atomic<int64_t> reserve_cus{0};      // CUs left without a scan workgroup (room for a collective's kernels under the scan)
    std::atomic<int64_t> tile_form{0};        // ALL, tile kernel, bits: 1 = never the fused tail (a finalize launch behind the scan: tests, A/B)
};
Tunables &tunables();

} // namespace acgpu
