/*
 * acgpu.h -- C ABI of the MI355X-native multi-pattern matcher (libacgpu.so).
 *
 * This is the drop-in boundary for the hot path of RokLenarcic/AhoCorasick:
 * the reference has no FFI seam of its own -- its seam is the Java interface
 * pair StringSet / StringMap (S/StringSet.java:3-5, S/StringMap.java:5-9) and
 * the listeners (S/SetMatchListener.java:6, S/MapMatchListener.java:6).  A Java
 * facade implementing those interfaces binds exactly the entry points below
 * through JNI (INTEGRATION.md shows the stub); tests and bench.py bind them
 * through ctypes.  S/ = src/main/java/com/roklenarcic/util/strings/.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ or torch types cross this boundary;
 *  - every function returns ACGPU_OK (0) or a negative ACGPU_E_* code; nothing
 *    throws across the ABI;
 *  - keywords and haystacks are UTF-16 code units exactly as a Java String holds
 *    them; match positions are code-unit indices, `end` exclusive
 *    (R/README.md:69);
 *  - an automaton is immutable after acgpu_build and may be shared by threads;
 *    concurrent matches on one automaton are safe (they serialise on its
 *    per-device scratch pool);
 *  - there is NO CPU matching backend: every acgpu_match_* call needs a HIP
 *    device and fails with ACGPU_E_NODEVICE / ACGPU_E_HIP without one;
 *  - device memory: the tables of an automaton (built once per device; the
 *    reference's 235 886-word README dictionary: 0.5 GB) and a grow-only scratch
 *    pool per automaton and device -- up to 16 bytes per record of the largest
 *    call, and for the matchers that keep a value per haystack unit
 *    (LongestMatch: 1-4 bytes; AhoCorasick on texts with dense matches,
 *    csrc/acgpu_states.hip: 4 bytes) that much per unit of the largest shard.
 *    A pool that cannot grow makes the call take a form that needs less, or
 *    fail with ACGPU_E_NOMEM; acgpu_free releases everything.
 *    Worst case per shard of N units and a caller capacity of C records, by
 *    the form a call takes (what a rank of a sharded text must be able to spare
 *    beside its text and its output; 8 GiB over 8 ranks: N = 2^29):
 *      AhoCorasick, tile kernel (the default)   20 C + 17 MB  (16-byte slots for
 *        1.25 C records + the waves' reservations)       C = N/128: 0.1 GB
 *      AhoCorasick, states form (texts with dense matches: 0.05 records per
 *        unit or more in the pool's last call)  4 N + N/128      N = 2^29: 2.0 GB
 *        -> without room: the tile kernel (nothing is launched before the
 *           buffers are there), also for the probe of a pool's first call
 *      AhoCorasick, DFA chunk scan              20 C
 *      LongestMatch, k_longest_bits             N/8 + N/1024      (marks, exits)
 *      LongestMatch, k_longest_follow           N/4 (+ 4 N for Map records)
 *      LongestMatch, walk pipeline              1.25 N .. 4.25 N  (lengths,
 *        block maxima, two bitmaps; states for Map records)
 *        -> a pool that cannot hold a form's buffers fails the call with
 *           ACGPU_E_NOMEM (the forms are not tried in turn)
 *      WholeWord                                6 N + 0.2 MB      (a 12-byte slot
 *        per two units)
 *        -> without room: 20 C + 17 MB (scratch slices + ordering pass)
 *      Shortest / WholeWordLongest              the AhoCorasick / WholeWord form
 *        + 16 bytes per record of the all-matches list / per walk start
 */
#ifndef ACGPU_H
#define ACGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACGPU_ABI_VERSION 5

/* error codes */
#define ACGPU_OK 0
#define ACGPU_E_INVALID (-1)     /* bad argument (NULL, misaligned device pointer, bad range ...)   */
#define ACGPU_E_NONWORD (-2)     /* java.lang.IllegalArgumentException of the WholeWord ctors:
                                    "<keyword> contains non-word characters."
                                    (S/WholeWordMatchMap.java:263-267, S/WholeWordMatchSet.java init) */
#define ACGPU_E_NOMEM (-3)       /* host or device allocation failed                                */
#define ACGPU_E_OVERFLOW (-4)    /* output capacity too small; *n_out holds the required record count */
#define ACGPU_E_HIP (-5)         /* HIP runtime error; see acgpu_last_hip_error()                    */
#define ACGPU_E_NODEVICE (-6)    /* no HIP device visible                                            */
#define ACGPU_E_UNSUPPORTED (-7) /* combination not implemented by this build                        */

/* matcher families (which reference class an automaton replaces) */
#define ACGPU_MODE_ALL 0       /* AhoCorasickSet / AhoCorasickMap      S/AhoCorasickSet.java:193-252, S/AhoCorasickMap.java:277-336 */
#define ACGPU_MODE_LONGEST 1   /* LongestMatchSet / LongestMatchMap    S/LongestMatchSet.java:192-265, S/LongestMatchMap.java:288-360 */
#define ACGPU_MODE_WHOLEWORD 2 /* WholeWordMatchSet / WholeWordMatchMap S/WholeWordMatchSet.java:47-132, S/WholeWordMatchMap.java:155-240 */
#define ACGPU_MODE_SHORTEST 3  /* ShortestMatchSet / ShortestMatchMap   S/ShortestMatchSet.java:193-262, S/ShortestMatchMap.java:294-372;
                                  keyword_id = FIRST duplicate (S/ShortestMatchMap.java:47-49) */
#define ACGPU_MODE_WWLONGEST 4 /* WholeWordLongestMatchSet / Map        S/WholeWordLongestMatchSet.java:47-178, S/WholeWordLongestMatchMap.java:54-305;
                                  keywords are trimmed but may contain non-word characters */

/*
 * Word-character tables that are NOT fold-consistent (acgpu_info.fold_consistent == 0: a custom table, case-insensitive,
 * with wordChars[c] != wordChars[toLowerCase(c)] for some c).  The reference's loops differ in which lookups fold, and so
 * do the calls that replace them:
 *  - WholeWordMatchSet/Map.match(String) (S/WholeWordMatchMap.java:204,209 folded; :221,:226 raw) and
 *    WholeWordLongestMatchSet.match(String) (S/WholeWordLongestMatchSet.java:126 folded; :151,:156 raw) MIX folded and raw
 *    lookups, which makes token boundaries history dependent: acgpu_match_u16 / acgpu_match_device serve them with a
 *    sequential kernel over the whole haystack as ONE shard (own range = buffer, text_begin = text_end = 1;
 *    ACGPU_E_UNSUPPORTED for other shards).  For WWLONGEST that is the call with ACGPU_REC_SET.
 *  - WholeWordLongestMatchMap.match(String) (S/WholeWordLongestMatchMap.java:252,258,283,288: every lookup folded) -- the
 *    WWLONGEST call with ACGPU_REC_MAP -- and every match(Readable) loop (S/WholeWordMatchMap.java:112,117,328,
 *    S/WholeWordLongestMatchMap.java:404) -- acgpu_stream_feed -- fold in EVERY lookup: ordinary position-parallel scans
 *    over wordChars o toLowerCase; shards and streams as with a fold-consistent table.
 */

/* output record layouts */
#define ACGPU_REC_SET 8  /* acgpu_set_match: what SetMatchListener.match(haystack, start, end) receives */
#define ACGPU_REC_MAP 12 /* acgpu_map_match: MapMatchListener.match(haystack, start, end, value);
                            keyword_id indexes the caller's value array                             */

typedef struct acgpu_set_match {
    int32_t start, end;
} acgpu_set_match;

typedef struct acgpu_map_match {
    int32_t start, end;
    int32_t keyword_id; /* index (in the acgpu_build input) of the LAST keyword equal to the matched
                           (folded) string: reproduces "last value wins", S/AhoCorasickMap.java:49-50 */
} acgpu_map_match;

typedef struct acgpu_automaton acgpu_automaton;

/*
 * Replaces the constructors: AhoCorasickSet(Iterable<String>, boolean[, Thresholder])
 * S/AhoCorasickSet.java:16-191, AhoCorasickMap(...) S/AhoCorasickMap.java:20-206,
 * LongestMatchSet(...) S/LongestMatchSet.java:15-190, WholeWordMatchMap/Set(...) + init
 * S/WholeWordMatchMap.java:21-53,246-323.
 *
 *  kw_units/kw_off : keyword i is kw_units[kw_off[i] .. kw_off[i+1]).  A null or empty Java
 *                    keyword is an empty range (both are skipped, S/AhoCorasickSet.java:27).
 *  case_sensitive  : 0 => every keyword unit and haystack unit is mapped through lower_tbl
 *                    (Character.toLowerCase(char), S/AhoCorasickSet.java:33,229).
 *  lower_tbl       : 65536 entries, required when case_sensitive == 0 (the caller's JVM fills
 *                    it with its own Character.toLowerCase so that parity holds for that JVM).
 *  wordchar_tbl    : 65536 flags, required for ACGPU_MODE_WHOLEWORD and ACGPU_MODE_WWLONGEST
 *                    (WordCharacters.generateWordCharsFlags*, S/WordCharacters.java:6-39).
 *  bad_keyword     : on ACGPU_E_NONWORD receives the index of the offending keyword (may be NULL).
 * The Thresholder argument of the reference constructors is a results-neutral memory/speed
 * knob of its node representation and has no counterpart here.
 */
int acgpu_build(int mode, const uint16_t *kw_units, const uint64_t *kw_off, uint32_t n_kw, int case_sensitive,
                const uint16_t *lower_tbl, const uint8_t *wordchar_tbl, acgpu_automaton **out, int64_t *bad_keyword);

void acgpu_free(acgpu_automaton *a);

typedef struct acgpu_info {
    uint32_t abi_version;
    uint32_t mode;
    uint32_t case_sensitive;
    uint32_t n_states;       /* trie nodes incl. root                                   */
    uint32_t n_classes;      /* character classes incl. class 0 = "in no keyword"       */
    uint32_t n_keywords;     /* distinct non-empty (folded, trimmed) keywords           */
    uint32_t min_keyword_len;
    uint32_t max_keyword_len;
    uint32_t dense;          /* 1: dense state x class table, 0: hashed edges + fail links */
    uint32_t entry_bytes;    /* 2 or 4                                                   */
    uint64_t table_bytes;    /* bytes of the transition structure resident in HBM        */
    uint32_t lds_states;     /* states whose rows are staged in LDS by the scan kernel   */
    uint32_t fold_consistent;/* WHOLEWORD / WWLONGEST: wordchar[c] == wordchar[lower[c]] for all c (see above) */
    uint32_t filter_k;       /* ALL: length of the suffix K-gram filter, 0 = none          */
    uint32_t filter_bits;    /* ALL: size of the K-gram bitmap in bits                      */
    uint32_t tile_kernel;    /* ALL/SHORTEST: 1 if the position-parallel K-gram kernel will be used (whenever the filter
                                exists); LONGEST: 1 if the filter is selective (all-matches pipeline + selection) */
    float filter_density;    /* ALL: fraction of K-grams (over keyword units) that pass     */
    uint32_t fold_clean;     /* WHOLEWORD: every FOLDED keyword unit is a word character (always 1 with a fold-consistent
                                table: keywords are validated on their raw units, S/WholeWordMatchMap.java:263-267); 0: a
                                folding scan walks the keyword trie through units it does not take for word characters */
} acgpu_info;

int acgpu_get_info(const acgpu_automaton *a, acgpu_info *info);

/*
 * Replaces StringSet.match(String, SetMatchListener) / StringMap.match(String, MapMatchListener)
 * for a haystack in HOST memory: copies it to the current HIP device, scans, and returns every
 * record the reference would have passed to its listener, in the reference's call order
 * (ALL: end ascending then start ascending, S/AhoCorasickSet.java:522-535; LONGEST and
 * WHOLEWORD: position order).  The facade then runs the listener loop itself, stopping at the
 * first `false` -- observationally identical to the reference's early stop
 * (S/AhoCorasickSet.java:223-225).
 *  record_kind : ACGPU_REC_SET or ACGPU_REC_MAP (layout of `out`).
 *  cap         : capacity of `out` in records.  On ACGPU_E_OVERFLOW nothing useful is in `out`
 *                and *n_out is the capacity to retry with.
 *  n_units     : < 2^31 (Java String limit).
 */
int acgpu_match_u16(const acgpu_automaton *a, const uint16_t *haystack, uint64_t n_units, int record_kind, void *out,
                    uint64_t cap, uint64_t *n_out);

/*
 * Many short haystacks in ONE call (the reference publishes one workload: a paragraph against a 235 k-word dictionary at
 * 3.6 us per match() call, R/README.md:130-148; a call here has tens of microseconds of fixed cost, so short inputs are
 * batched): haystack i is units[offsets[i] .. offsets[i+1]).  The haystacks are scanned as one text with a separator unit
 * between them that no keyword contains (word matchers: and that is no word character), so every haystack's matches are
 * exactly what match(String) reports for it alone -- any family.  Records carry the haystack index in front:
 *   ACGPU_REC_SET -> acgpu_batch_set_match {haystack, start, end}, ACGPU_REC_MAP -> acgpu_batch_map_match {haystack, start,
 *   end, keyword_id}; positions are relative to their haystack; order: haystack ascending, inside a haystack the
 *   reference's listener-call order.  The facade calls the listener per record and, if it returns false, skips the rest of
 *   THAT haystack's records.
 * WholeWordLongest: the reference's scan starts a walk at position 0 whatever stands there (and a keyword without word
 * characters gives the root a transition on a non-word unit), so the unit behind every separator is a walk start too.
 * One call per haystack inside the library instead, same results: a dictionary that uses all 65536 units, and the word
 * matchers over a table that is not fold-consistent (acgpu_info.fold_consistent == 0).
 * offsets[n_haystacks] - offsets[0] + n_haystacks must stay below 2^31.  On ACGPU_E_OVERFLOW *n_out is the capacity to retry with.
 */
typedef struct acgpu_batch_set_match {
    int32_t haystack, start, end;
} acgpu_batch_set_match;
typedef struct acgpu_batch_map_match {
    int32_t haystack, start, end, keyword_id;
} acgpu_batch_map_match;
int acgpu_match_batch_u16(const acgpu_automaton *a, const uint16_t *units, const uint64_t *offsets, uint32_t n_haystacks,
                          int record_kind, void *out, uint64_t cap, uint64_t *n_out);

/*
 * Device-resident form of the same call, and the unit of multi-GPU sharding.
 * The buffer holds an owned range plus halos; positions in the records are relative to the
 * buffer start.  Ownership: ALL -> a match belongs to the shard that owns its LAST unit (left
 * halo >= max_keyword_len-1 units needed); WHOLEWORD -> to the shard that owns the first unit of
 * the word (left halo 1 unit, right halo up to the end of the word or max_keyword_len+1 units);
 * LONGEST -> to the shard that owns its first unit, given the greedy chain's entry position
 * (right halo >= max_keyword_len-1 units); WWLONGEST -> to the shard that owns its first unit (left halo 1 unit, right
 * halo max_keyword_len+1 units), given the position from which the scan looks for its next word start (chain_entry;
 * chain_exit: a position behind the last walk this shard's scan made from which the search for the next word start finds
 * what the reference's scan finds -- behind the walk's stop, or behind the word in which the walk died); SHORTEST -> to the shard that owns its LAST unit (left halo as ALL), given
 * the position at which matching last restarted (chain_entry: no match may start before it; chain_exit: the end of
 * the last match reported, or chain_entry if there was none).
 */
typedef struct acgpu_shard {
    const uint16_t *d_hay; /* device pointer, 16-byte aligned                                   */
    uint64_t n_units;      /* units in the buffer, < 2^31                                        */
    uint64_t own_begin;    /* owned range [own_begin, own_end) inside the buffer                 */
    uint64_t own_end;
    int32_t text_begin;    /* 1: buffer unit 0 is the first unit of the whole haystack           */
    int32_t text_end;      /* 1: buffer unit n_units-1 is the last unit of the whole haystack    */
    int64_t chain_entry;   /* LONGEST in : first greedy-chain position >= own_begin (own_begin on the first shard);
                              WWLONGEST in: the scan visits the first word start at or after this position;
                              SHORTEST in: position of the last restart (0 on the first shard)      */
    int64_t chain_exit;    /* LONGEST out: first greedy-chain position >= own_end; WWLONGEST / SHORTEST out: see above */
    void *d_result;        /* optional device pointer (16-byte aligned) to an acgpu_device_result that the call's last
                              kernel (or a copy enqueued behind it) fills in STREAM ORDER: a multi-GPU driver points it
                              into the buffer it all-gathers, so the record count travels with the records and no host
                              round trip sits between the scan and the collective.  NULL: not wanted.            */
} acgpu_shard;

/* what acgpu_shard.d_result receives */
typedef struct acgpu_device_result {
    uint64_t n_records; /* records the call produced (may exceed cap: then d_out is incomplete)                     */
    uint32_t redone;    /* non-zero: the library has to redo this call (acgpu_match_device_end does) -- the records
                           behind this header are not valid yet; a driver that gathered them gathers again          */
    uint32_t reserved;
} acgpu_device_result;

/* optional per-call timing of the device work, measured with HIP events on `stream` */
typedef struct acgpu_profile {
    float scan_ms;     /* the dominant kernel: one pass over the haystack units                  */
    float finalize_ms; /* ordering of records (prefix sum of per-chunk counts + permutation)     */
    float total_ms;    /* first launch to last launch of the call                                */
    uint64_t scan_units;   /* haystack units the scan kernel processed (owned + halo)            */
    uint64_t n_matches;
    char scan_kernel[64];  /* name of the dominant kernel as it appears in rocprofv3             */
} acgpu_profile;

/*
 *  d_out  : device pointer to cap records of record_kind (16-byte aligned).
 *  stream : hipStream_t (NULL = the default stream).  The call enqueues its kernels on `stream`
 *           and synchronises that stream once, to read back the record count.
 *  prof   : NULL, or receives HIP-event timings of this call.
 */
int acgpu_match_device(const acgpu_automaton *a, acgpu_shard *shard, int record_kind, void *d_out, uint64_t cap,
                       uint64_t *n_out, void *stream, acgpu_profile *prof);

/*
 * Asynchronous form: _begin enqueues the whole pipeline on `stream` and returns without waiting; _end waits for that call only
 * (an event, not the stream) and returns its count / timings.  Up to 4 calls may be in flight per automaton and device.
 * Lets a host keep the GPU busy across calls: the next scan is queued while the previous count travels back.
 * want_profile != 0 records the HIP events that _end turns into acgpu_profile.
 * The acgpu_shard passed to _begin must stay alive until _end (or _abandon) has returned: _end writes chain_exit into it.
 *  - Enqueued without any host round trip: ACGPU_MODE_ALL, ACGPU_MODE_WHOLEWORD (fold-consistent tables) and
 *    ACGPU_MODE_LONGEST through its walk pipeline (shard->chain_entry is read at _begin; chain_exit is written to the
 *    SAME acgpu_shard object when _end returns, so it must outlive the ticket).
 *  - The other families (and LONGEST over a dictionary with a selective suffix filter, whose sparse form decides on the
 *    host) run inside _begin: the ticket is complete when _begin returns and _end only hands the result over.
 *
 *  - ACGPU_MODE_ALL chooses between the tile kernel and the states form by what the pool's LAST call found (records per
 *    unit).  A pool's first call asks the text itself -- the first 2^20 units are counted first -- only when it is
 *    synchronous and the text has 2^23 units or more: a first call through _begin, or on a shorter text, takes the tile
 *    kernel.  The records are the same either way; on a word list in natural text the first enqueued call is the slow one.
 *
 * STREAM RULE.  All calls on one automaton and device share that automaton's scratch pool; stream order is what keeps
 * them apart.  While tickets are in flight, EVERY call on that automaton and device -- another _begin, a synchronous
 * acgpu_match_device, acgpu_match_u16 or acgpu_stream_feed (both use the NULL stream) -- must use the stream of the
 * tickets; a call on a different stream is refused with ACGPU_E_INVALID instead of racing on the scratch.  With no
 * ticket in flight any stream may be used (a synchronous call has left the scratch idle when it returns).
 */
typedef struct acgpu_ticket acgpu_ticket;
int acgpu_match_device_begin(const acgpu_automaton *a, acgpu_shard *shard, int record_kind, void *d_out, uint64_t cap,
                             void *stream, int want_profile, acgpu_ticket **ticket);
int acgpu_match_device_end(const acgpu_automaton *a, acgpu_ticket *ticket, uint64_t *n_out, acgpu_profile *prof);
/* Gives a ticket up without collecting it: waits until its kernels have finished (they write the caller's buffers until
 * then), never redoes the call.  For a driver that already knows -- from the device result it gathered -- that the step has
 * to be done again with larger buffers. */
int acgpu_match_device_abandon(const acgpu_automaton *a, acgpu_ticket *ticket);

/*
 * ---- several devices, ONE host process -------------------------------------------------------------------------------------
 * BASELINE north_star: "long haystacks shard naturally across the 8 GPUs of one node with an RCCL all-gather over xGMI of
 * per-shard match buffers plus a (longest-pattern - 1) halo".  The reference's call is one call in one process
 * (S/StringSet.java:3-5, S/StringMap.java:5-9: match(String haystack, listener)); so is this one.
 *
 * acgpu_match_u16_multi: acgpu_match_u16 with the haystack cut into n_devices contiguous shares, share i on HIP device
 * devices[i] -- each with its own pinned staging ring, copy stream, scan stream and copy of the automaton's tables, each fed
 * and scanned by its own host thread, all at once.  A share carries the halo its family needs (ALL / SHORTEST: max_len-1
 * units on the left; WHOLEWORD / WWLONGEST: 1 unit on the left, max_len+1 on the right; LONGEST: max_len-1 on the right).
 * The chain families (LONGEST, SHORTEST, WWLONGEST) scan every share speculatively and repair the head of a share whose
 * true entry -- the previous share's exit -- differs (two short window scans, as a rule).  Records arrive in `out` in the
 * reference's listener-call order for the WHOLE haystack with global positions: exactly what acgpu_match_u16 returns.
 *  devices   : n_devices HIP device ordinals.  A device may be named more than once (further shares on the same device get
 *              their own scratch pool and stream): how a one-GPU box tests the split.
 *  A share has fixed costs (a host thread, a pinned staging ring, two synchronisations), so a haystack is cut into at most
 *  n_units / 2^22 shares (8 MiB of text each; development tunable "multi_min_share"); shorter ones, and the loops that only
 *  exist as one sequential scan (word matchers over a table that is not fold-consistent), run on devices[0] alone through
 *  acgpu_match_u16.
 *  Scaling: every share is fed from the CALLER's memory by host threads copying pageable -> pinned memory (about 45 GB/s per
 *  share on one socket), so this entry is bound by the host's copy bandwidth beyond about two devices per socket; the feeder
 *  threads of a share are pinned to the NUMA node of its device where the platform reports one.  The form that scales with the
 *  device count is acgpu_match_device_allgather below: shards resident in device memory, nothing of the text on the host.
 * The calling thread's current device is restored.  Errors and ACGPU_E_OVERFLOW as acgpu_match_u16.
 */
int acgpu_match_u16_multi(const acgpu_automaton *a, const uint16_t *haystack, uint64_t n_units, const int *devices, int n_devices,
                          int record_kind, void *out, uint64_t cap, uint64_t *n_out);

/*
 * Device-resident form with the gather the north_star names: every device scans ITS shard (already in its memory, halos in
 * place: acgpu_shard as for acgpu_match_device) into its own slot of a gather buffer, and one all-gather leaves every
 * device with every shard's records -- config 3 of BASELINE.json.
 *
 * acgpu_comm: the devices of a job, one stream per device, and the transport of the gather:
 *   ACGPU_TRANSPORT_RCCL : single-process RCCL (ncclCommInitAll over the device list, one ncclAllGather per device inside
 *                          ncclGroupStart/End, in place).  librccl.so.1 is bound at run time (dlopen): libacgpu.so does not
 *                          link it, and ACGPU_E_UNSUPPORTED is returned where it is missing.  Needs distinct devices.
 *   ACGPU_TRANSPORT_PEER : hipMemcpyPeerAsync of every slot to every other device over the direct xGMI links (also what a
 *                          device list that names a device twice gets).
 *   ACGPU_TRANSPORT_AUTO : RCCL when the devices are distinct and the library loads, else peer copies.
 *
 * Gather buffer of device i: n_devices slots of acgpu_gather_slot_bytes(gcap, record_kind) bytes; slot j =
 * [the 16 bytes of an acgpu_device_result | gcap records] of shard j, positions relative to shard j's buffer.  d_gather[i] is a
 * device pointer on devices[i], 16-byte aligned.
 *  shards  : n_devices shards, shard i resident on devices[i].  Chain families: shards[0].chain_entry is the chain's entry
 *            into the whole text; every shards[i].chain_exit is set to the true scan's exit from shard i.
 *  counts  : receives the n_devices record counts (also in the gathered headers).
 *  profs   : NULL, or n_devices acgpu_profile structs: HIP-event timings of each device's scan.
 * Returns ACGPU_E_OVERFLOW when some count exceeds gcap: counts[] are exact, the buffers hold nothing useful, nothing was
 * gathered; call again with gcap >= the largest count.
 * AhoCorasick and WholeWord (fold-consistent tables) are enqueued on every device without a host round trip (scan, header
 * and gather in stream order; one host wait at the end); the other families run their shards on one host thread per
 * device, then the repairs, then the gather.
 */
#define ACGPU_TRANSPORT_AUTO 0
#define ACGPU_TRANSPORT_RCCL 1
#define ACGPU_TRANSPORT_PEER 2
typedef struct acgpu_comm acgpu_comm;
int acgpu_comm_open(const int *devices, int n_devices, int transport, acgpu_comm **out);
void acgpu_comm_close(acgpu_comm *c);
int acgpu_comm_transport(const acgpu_comm *c);       /* the transport in use: ACGPU_TRANSPORT_RCCL or _PEER              */
void *acgpu_comm_stream(const acgpu_comm *c, int i); /* hipStream_t of device i: work a caller enqueues there (filling the
                                                        shard, reading the gathered records) is ordered with the call's   */
uint64_t acgpu_gather_slot_bytes(uint64_t gcap, int record_kind); /* (16 + gcap * record_kind), rounded up to 16 */
int acgpu_match_device_allgather(const acgpu_automaton *a, acgpu_comm *c, acgpu_shard *shards, int record_kind,
                                 void *const *d_gather, uint64_t gcap, uint64_t *counts, acgpu_profile *profs);
int acgpu_last_rccl_error(void); /* ncclResult_t of the last RCCL failure on this thread */

/*
 * Replaces StringMap.match(Readable, ReadableMatchListener<T>) (S/StringMap.java:6; S/AhoCorasickMap.java:208-275,
 * S/LongestMatchMap.java:203-286, S/WholeWordMatchMap.java:55-153): the haystack arrives in chunks of any size (the
 * reference reads it through a CharBuffer) and may be longer than a Java String.  Every feed returns the records that
 * have become decidable, in the reference's call order; the concatenation over all feeds is what match(String) would
 * deliver for the whole text (T/MapTest.java:178-188).  The facade passes record.keyword_id -> value to the listener
 * (the Readable listener receives only the value) and may stop feeding when the listener returns false.
 *  feed   : units = the next n_units of the haystack in HOST memory (may be 0); final != 0 marks the end of the
 *           haystack, after which only close is valid.  Internally the stream keeps the few units a later chunk can
 *           still change the answer for (ALL: max_keyword_len-1 units of left context; WHOLEWORD: one unit of context
 *           plus the last max_keyword_len+1 units, whose words are decided by the next feed; LONGEST: the last
 *           max_keyword_len-1 units plus the position at which the greedy chain continues; WWLONGEST: as WHOLEWORD plus
 *           the position from which the scan looks for its next word start).
 *  out    : cap records of record_kind; start/end are int32 relative to *base (the global position, in units since
 *           the first feed, that record coordinate 0 stands for).  On ACGPU_E_OVERFLOW nothing was consumed: *n_out
 *           is the capacity to call the SAME feed again with.
 *  carried units + n_units must stay below 2^31.
 *  Word-character tables that are not fold-consistent: the Readable loops fold in every lookup (see above), and so do
 *  the feeds.
 *  Lifetime and device: close a stream BEFORE acgpu_free of its automaton (one that is still open then is detached: its
 *  feeds return ACGPU_E_INVALID, acgpu_stream_close stays safe).  Every feed and acgpu_stream_reserve must come from a
 *  thread whose current HIP device is the one of the stream's first feed: another device returns ACGPU_E_INVALID.
 */
typedef struct acgpu_stream acgpu_stream;
int acgpu_stream_open(const acgpu_automaton *a, acgpu_stream **out);
int acgpu_stream_feed(acgpu_stream *s, const uint16_t *units, uint64_t n_units, int final, int record_kind, void *out,
                      uint64_t cap, uint64_t *n_out, int64_t *base);
void acgpu_stream_close(acgpu_stream *s);
/*
 * The pipelined form of the feeds (call before the first feed; on != 0).  A feed then copies its chunk into pinned staging
 * memory with several host threads -- every piece goes on its way to the device as soon as it has been copied -- while the
 * calling thread scans the PREVIOUS chunk and returns ITS records: host copy, transfer and scan of neighbouring chunks
 * overlap.  Differences to the default form, all of them about WHEN records arrive, none about which:
 *  - a feed returns the records the previous feed's chunk made decidable (the first feed returns none); the feed with
 *    final != 0 returns the previous chunk's and its own; *base is the position record coordinate 0 stands for, as before;
 *  - ACGPU_E_OVERFLOW: the chunk HAS been consumed; call the same feed again with the capacity in *n_out and it hands the
 *    records over (its units are not looked at again);
 *  - carried units + n_units must stay below 2^30; all feeds of a stream come from threads whose current HIP device is the
 *    one the first feed ran on.
 * acgpu_stream_reserve (pipelined streams): the address at which the next chunk of up to n_units units may be WRITTEN by the
 * caller -- the staging memory itself; a feed whose `units` is that address copies nothing (a JNI glue reads the Java char[]
 * straight into it).
 */
int acgpu_stream_set_pipelined(acgpu_stream *s, int on);
int acgpu_stream_reserve(acgpu_stream *s, uint64_t n_units, uint16_t **buf);

/*
 * Synthetic haystack generator of the benchmark: unit i of the stream is
 * table[((z_i >> 32) * table_len) >> 32] with z_i = SplitMix64 output for counter
 * start_index + i of `seed` (see ahocorasick_amd/synth.py).  d_dst: device pointer.
 */
int acgpu_synth_fill(uint16_t *d_dst, uint64_t n_units, uint64_t start_index, uint64_t seed, const uint16_t *table,
                     uint32_t table_len, void *stream);

/*
 * Config 5's haystack (tokens -- a dictionary word with random per-unit case flips, or a random single-script
 * word -- separated by 1-3 separator units), generated in place on the device.  Token t owns draws 32 t .. 32 t + 31 of the
 * SplitMix64 stream `seed`, so the text is a function of (seed, dictionary) alone (ahocorasick_amd/synth.py:
 * token_stream_haystack is the numpy twin).  kw_units / kw_off / swapcase_tbl (65536 entries, may be NULL: no case flips) are
 * HOST pointers; d_dst is a device pointer.  Synchronises `stream`.
 */
int acgpu_synth_tokens(uint16_t *d_dst, uint64_t n_units, uint64_t seed, const uint16_t *kw_units, const uint64_t *kw_off,
                       uint32_t n_kw, const uint16_t *swapcase_tbl, void *stream);

/*
 * Measurement helper ("also report vs. a measured streaming-read kernel on the same box"): a pure read
 * of n_bytes (>= 1 MiB, 16-byte aligned device pointer), `repeats` timed launches after a warm-up (HIP events on `stream`,
 * synchronous), the median in *ms_median.
 *  pattern 1: the fastest pure read found on this chip (groups of 2 KiB tiles dealt round robin to small workgroups, 32 bytes
 *             per lane): the attainable ceiling bench.py reports next to the 8 TB/s spec peak;
 *  pattern 0: the access pattern of the tile kernels themselves -- every wave a contiguous span, 64 bytes per lane and tile,
 *             the next tile's loads in flight -- i.e. what their stream alone costs.
 */
int acgpu_stream_probe(const void *d_buf, uint64_t n_bytes, void *stream, int repeats, int pattern, float *ms_median);

/* tuning knobs: DEVELOPMENT AND TEST HOOK, not part of the product surface a JVM binds.  Process-wide, read when a call
 * is enqueued; set them only while no match call is running (each knob is a relaxed atomic, so a concurrent reader sees
 * the old or the new value, never a torn one, but a call may then mix settings).  name: "chunk_units",
 * "blocks_per_cu", "lds_table_bytes" (rows of the state x class table the DFA chunk scan keeps in LDS: default and maximum
 * 127 KB), "force_sparse", "dense_budget_bytes", "force_kernel" (0 auto, 1 DFA chunk
 * scan, 2 K-gram tile scan), "region_units", "filter_max_bytes", "ww_first_seed" (WHOLEWORD builder: index of the first
 * hash seed tried), and the builder's A/B switches "no_merged_ranges" (dictionaries over several ranges keep the class-table
 * filter), "no_short_keywords" (the filter's K stays at most the shortest keyword), "no_class_pages" (the class-table forms of
 * the tile kernel look classes up in global memory instead of LDS pages); "reserve_cus" (the scan kernels size their
 * grids for that many CUs fewer: a scan workgroup holds a whole CU's LDS, so k CUs stay free for the kernels of a collective
 * that runs under the scan -- RCCL's all-gather in a multi-GPU job); "tile_form" (bit 0: the K-gram tile scan leaves the
 * ordering of its records to a second kernel instead of its own tail, bit 1: the same for the WHOLEWORD kernel -- the
 * separate launches bench.py times beside the one-kernel call); the WHOLEWORD builder's and kernel's A/B switches
 * "ww_no_ph" (two-choice hash table instead of the perfect one), "ww_ph_lambda" (keys per bucket of the perfect hash,
 * default 4), "ww_no_byte_pages" (units staged as class codes, not as one byte each), "ww_block" (threads of a workgroup,
 * a multiple of 64) and "ww_ramp_pm" (per mille by which the spans of the last workgroups shrink); "longest_form" (bits:
 * 1 no k_longest_bits, 2 no k_longest_follow, 4 both for short texts too, 8 k_longest_follow over alphabets of up to four
 * letters).  Returns the previous value, -1 for an unknown name. */
int64_t acgpu_set_tunable(const char *name, int64_t value);

const char *acgpu_strerror(int code);
int acgpu_last_hip_error(void);  /* hipError_t of the last ACGPU_E_HIP on this thread */
uint32_t acgpu_abi_version(void);

/* Test hook: copies the host-side tables of an automaton (NULL pointers are skipped).  Lets
 * CPU-only tests check the builder without a device.  dfa receives n_states*n_classes uint32
 * entries when info.dense, per-state arrays receive n_states entries. */
int acgpu_debug_tables(const acgpu_automaton *a, uint16_t *cls_lut /*65536*/, uint32_t *dfa, uint32_t *out_len,
                       uint32_t *out_link, uint32_t *out_id, uint32_t *depth, uint32_t *first_out_state);

/* Test hook (ACGPU_MODE_ALL / ACGPU_MODE_SHORTEST): the compact automaton k_ac_states walks (csrc/acgpu_build.cpp 6d), if the
 * dictionary has one (at most 255 classes, keywords of at most 32 units, fewer than 2^23 states).  sizes[6] is always written:
 * {states, states of the dense group, classes, words of rows, words of nodes, words of ids}; 0 states = none.  Arrays are copied
 * when the pointer is non-NULL: rows = dense-group rows of `classes` resolved transitions (target state | bit 23 "reports
 * matches" | its number of keywords << 24); nodes = 4 words per other state {fail state | 3 bits per edge: the child's number of
 * keywords (7 = more), then three edges class << 24 | bit 23 | child}; mask[state] = bit L - 1 per keyword length L ending there;
 * out = 2 words per state {mask, index of its keyword ids in ids -- or bit 31 | the id of its only keyword}; ids. */
int acgpu_debug_states(const acgpu_automaton *a, uint64_t sizes[6], uint32_t *rows, uint32_t *nodes, uint32_t *mask, uint32_t *out,
                       uint32_t *ids);

/* Test hook (ACGPU_MODE_WHOLEWORD): the whole-keyword hash table the device kernel probes.  Sizes and the seed are always
 * written; arrays are copied when the pointer is non-NULL.
 * Hashes of a folded keyword over its max(8, ceil(length/2)) packed words w (two units per word, zero beyond the keyword):
 *   h = seed; h = h*33 + w            ... then the murmur3 32-bit finaliser        (hash)
 *   g = seed; g = rotl(g, 5) ^ w                                                    (second hash)
 * slots: a two-choice table of n_slots (a power of two) slots of 8 uint32: {tag, id, units 0..11 packed two per word};
 * tag = (h & 0xffffff00) | min(length, 255), 0 = free slot; a keyword sits in slot (h & (n_slots-1)) or in slot
 * s2 = (((g ^ (h >> 16) ^ (g >> 13)) * 0x2C1B3C6D) >> 11) & (n_slots-1)  (s2 ^ 1 if that equals the first).  Keywords of
 * more than 12 units hold their record's offset (in 16-byte units) in place of the id.
 * recs: uint32 words, record = {keyword id, length, folded units packed two per word, zero padded to 16 bytes}.
 * fold_pgidx[256] / fold_pages[n_pages*256]: lower[u] = (u + fold_pages[fold_pgidx[u>>8]*256 + (u&255)]) & 0xffff;
 * page 0 is all zero. */
int acgpu_debug_wordhash(const acgpu_automaton *a, uint32_t *n_slots, uint32_t *slots, uint64_t *n_rec_words, uint32_t *recs,
                         uint8_t *fold_pgidx, uint32_t *n_pages, uint16_t *fold_pages, uint32_t *seed);

/* Test hook (ACGPU_MODE_WHOLEWORD): what the position-parallel word kernel (k_ww_pp) probes in place of the two-choice table and
 * looks units up in.  sizes = {slots of the perfect hash, its buckets, byte pages}; 0 slots / 0 pages = not built.  Arrays are
 * copied when the pointer is non-NULL.
 * Perfect hash ("hash and displace") over the same hashes h, g as acgpu_debug_wordhash:
 *   bucket = (h * n_buckets) >> 32;  d = disp[bucket];
 *   t = (g ^ (h << 7)) + d * 0x9E3779B9;  t ^= t >> 15;  t *= 0x2C1B3C6D;  t ^= t >> 13;  slot = (t * n_slots) >> 32
 * every keyword sits in the slot its hashes name (slots: 8 uint32 each, the two-choice table's format); a run of word characters
 * that is no keyword reads some slot and fails the comparison of tag and units.
 * Byte pages (case-insensitive automata whose word-character table is fold-consistent): e = pages[idx[u >> 8] * 256 + (u & 255)];
 * word character = e & 1, lower[u] = (u + delta[e >> 1]) & 0xffff (delta: 128 entries). */
int acgpu_debug_wordhash_perfect(const acgpu_automaton *a, uint32_t sizes[3], uint32_t *slots, uint16_t *disp, uint8_t *bp_idx,
                                 uint8_t *bp_pages, uint16_t *bp_delta);

#ifdef __cplusplus
}
#endif
#endif /* ACGPU_H */
