// acgpu_build.cpp -- host builder: dictionary -> automaton tables.
//
// Replaces the constructors of the reference matchers (trie insertion, BFS fail links and
// compressed output links, S/AhoCorasickSet.java:20-191; Map value rules S/AhoCorasickMap.java:32-50,
// 129-133; WholeWord trimming/validation S/WholeWordMatchMap.java:246-323).  The reference's node
// representation (HashmapNode/RangeNode, Thresholder, gap fill) is a results-neutral speed knob of a
// pointer-chasing CPU loop and is not reproduced: the device works on one flat table.
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <deque>
#include <unordered_map>

#include "acgpu_internal.h"

namespace acgpu {

Tunables &tunables() {
    static Tunables t;
    return t;
}

namespace {

struct TrieNode {
    uint32_t parent;
    uint16_t unit;
    uint32_t depth;
    uint32_t kw; // own keyword id (last wins) or ~0u
};

// WordCharacters.trim, S/WordCharacters.java:41-62
void trim_keyword(const uint8_t *word, const uint16_t *w, uint64_t len, uint64_t &ws, uint64_t &we) {
    ws = 0;
    we = len;
    for (uint64_t i = 0; i < len; i++)
        if (word[w[i]]) { ws = i; break; }
    for (uint64_t i = len; i-- > 0;)
        if (word[w[i]]) { we = i + 1; break; }
}

} // namespace

int build_tables(int mode, const uint16_t *kw_units, const uint64_t *kw_off, uint32_t n_kw, int case_sensitive,
                 const uint16_t *lower_tbl, const uint8_t *wordchar_tbl, HostTables &t, int64_t *bad_keyword) {
    if (mode != ACGPU_MODE_ALL && mode != ACGPU_MODE_LONGEST && mode != ACGPU_MODE_WHOLEWORD && mode != ACGPU_MODE_SHORTEST &&
        mode != ACGPU_MODE_WWLONGEST)
        return ACGPU_E_INVALID;
    const bool word_mode = mode == ACGPU_MODE_WHOLEWORD || mode == ACGPU_MODE_WWLONGEST;
    if (n_kw && (!kw_units || !kw_off)) return ACGPU_E_INVALID;
    if (!case_sensitive && !lower_tbl) return ACGPU_E_INVALID;
    if (word_mode && !wordchar_tbl) return ACGPU_E_INVALID;

    t.mode = mode;
    t.cs = case_sensitive != 0;
    t.lower.resize(65536);
    for (uint32_t c = 0; c < 65536; c++) t.lower[c] = t.cs ? (uint16_t)c : lower_tbl[c];
    if (word_mode) {
        t.wflags.resize(65536);
        t.fold_consistent = true;
        for (uint32_t c = 0; c < 65536; c++) {
            uint8_t raw = wordchar_tbl[c] ? 1 : 0, fold = wordchar_tbl[t.lower[c]] ? 1 : 0;
            t.wflags[c] = (uint8_t)(raw | (fold << 1));
            if (raw != fold) t.fold_consistent = false;
        }
        t.wbits.assign(2048, 0u);
        for (uint32_t c = 0; c < 65536; c++) t.wbits[c >> 5] |= (uint32_t)(t.wflags[c] & 1u) << (c & 31);
        if (!t.fold_consistent) { // the tables of the loops that fold in every lookup (HostTables::wflags_f)
            t.wflags_f.resize(65536);
            t.wbits_f.assign(2048, 0u);
            for (uint32_t c = 0; c < 65536; c++) {
                const uint32_t f = (t.wflags[c] >> 1) & 1u;
                t.wflags_f[c] = (uint8_t)(f | (f << 1));
                t.wbits_f[c >> 5] |= f << (c & 31);
            }
        }
    }

    // ---- 1. trie insertion (keyword order matters: the LAST duplicate's index wins) ----
    std::vector<TrieNode> nodes;
    nodes.push_back({0, 0, 0, ~0u});
    std::unordered_map<uint64_t, uint32_t> edge; // (node<<16 | unit) -> child
    edge.reserve(1024);
    t.min_len = 0;
    t.max_len = 0;
    uint32_t n_terminal = 0;
    for (uint32_t k = 0; k < n_kw; k++) {
        const uint16_t *w = kw_units + kw_off[k];
        uint64_t len = kw_off[k + 1] - kw_off[k], ws = 0, we = len;
        if (mode == ACGPU_MODE_WWLONGEST) trim_keyword(wordchar_tbl, w, len, ws, we); // inner non-word units are allowed
        if (mode == ACGPU_MODE_WHOLEWORD) {
            trim_keyword(wordchar_tbl, w, len, ws, we);
            for (uint64_t i = ws; i < we; i++) {
                if (!wordchar_tbl[w[i]]) { // validated on the un-folded units, S/WholeWordMatchMap.java:263-267
                    if (bad_keyword) *bad_keyword = (int64_t)k;
                    return ACGPU_E_NONWORD;
                }
            }
        }
        if (we <= ws) continue; // null / empty keywords are skipped, S/AhoCorasickSet.java:27
        uint32_t cur = 0;
        for (uint64_t i = ws; i < we; i++) {
            uint16_t u = t.lower[w[i]];
            uint64_t key = ((uint64_t)cur << 16) | u;
            auto it = edge.find(key);
            if (it == edge.end()) {
                uint32_t id = (uint32_t)nodes.size();
                if (id == 0x7fffffffu) return ACGPU_E_UNSUPPORTED;
                nodes.push_back({cur, u, nodes[cur].depth + 1, ~0u});
                edge.emplace(key, id);
                cur = id;
            } else {
                cur = it->second;
            }
        }
        const bool dup = nodes[cur].kw != ~0u;
        if (!dup) n_terminal++;
        // AhoCorasick/Longest/WholeWord: the LAST duplicate's value wins; Shortest: the first keeps the node
        // (S/ShortestMatchMap.java:47-49)
        if (!dup || mode != ACGPU_MODE_SHORTEST) nodes[cur].kw = k;
        uint32_t L = (uint32_t)(we - ws);
        if (t.max_len == 0 || L > t.max_len) t.max_len = L;
        if (t.min_len == 0 || L < t.min_len) t.min_len = L;
    }
    const uint32_t N = (uint32_t)nodes.size();
    t.fold_clean = true;
    if (mode == ACGPU_MODE_WHOLEWORD)
        for (uint32_t i = 1; i < N; i++)
            if (!wordchar_tbl[nodes[i].unit]) t.fold_clean = false;
    t.n_states = N;
    t.n_kw = n_terminal;
    t.n_edges = N - 1;
    // a unit that can stand between two haystacks of a batch (acgpu_match_batch_u16): its folded form occurs in no keyword and
    // -- word matchers -- neither it nor its folded form is a word character, so no match, word or walk reaches across it
    {
        std::vector<uint8_t> in_kw(65536, 0);
        for (uint32_t i = 1; i < N; i++) in_kw[nodes[i].unit] = 1;
        t.sep_unit = -1;
        for (int32_t u = 0xffff; u >= 0; --u) { // (from the top: U+FFFF is a noncharacter)
            const uint16_t f = t.lower[u];
            if (in_kw[f] || in_kw[u]) continue;
            if (word_mode && (wordchar_tbl[u] || wordchar_tbl[f])) continue;
            t.sep_unit = u;
            break;
        }
    }

    // ---- 2. children lists (CSR by insertion id), sorted by unit for a deterministic BFS ----
    std::vector<uint32_t> child_begin(N + 1, 0);
    for (uint32_t i = 1; i < N; i++) child_begin[nodes[i].parent + 1]++;
    for (uint32_t i = 0; i < N; i++) child_begin[i + 1] += child_begin[i];
    std::vector<uint32_t> child_ids(N ? N - 1 : 0);
    {
        std::vector<uint32_t> fill(child_begin.begin(), child_begin.end() - 1);
        for (uint32_t i = 1; i < N; i++) child_ids[fill[nodes[i].parent]++] = i;
        for (uint32_t i = 0; i < N; i++)
            std::sort(child_ids.begin() + child_begin[i], child_ids.begin() + child_begin[i + 1],
                      [&](uint32_t a, uint32_t b) { return nodes[a].unit < nodes[b].unit; });
    }
    auto find_child = [&](uint32_t s, uint16_t u) -> uint32_t { // ~0u if absent
        uint32_t lo = child_begin[s], hi = child_begin[s + 1];
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            uint16_t mu = nodes[child_ids[mid]].unit;
            if (mu < u) lo = mid + 1; else hi = mid;
        }
        return (lo < child_begin[s + 1] && nodes[child_ids[lo]].unit == u) ? child_ids[lo] : ~0u;
    };

    // ---- 3. BFS: fail links and compressed outputs (AC semantics; S/AhoCorasickSet.java:58-121) ----
    std::vector<uint32_t> bfs;
    bfs.reserve(N);
    bfs.push_back(0);
    std::vector<uint32_t> fail(N, 0), olen(N, 0), olink(N, 0), oid(N, ~0u);
    const bool want_fail = !word_mode; // WholeWord / WholeWordLongest are plain tries (S/WholeWordMatchMap.java:303-321)
    for (size_t qi = 0; qi < bfs.size(); qi++) {
        uint32_t s = bfs[qi];
        for (uint32_t ci = child_begin[s]; ci < child_begin[s + 1]; ci++) {
            uint32_t c = child_ids[ci];
            bfs.push_back(c);
            uint16_t u = nodes[c].unit;
            uint32_t f = 0;
            if (want_fail && s != 0) {
                uint32_t pf = fail[s];
                for (;;) {
                    uint32_t m = find_child(pf, u);
                    if (m != ~0u) { f = m; break; }
                    if (pf == 0) { f = 0; break; }
                    pf = fail[pf];
                }
            }
            fail[c] = f;
            bool own = nodes[c].kw != ~0u;
            // the nearest fail ancestor carrying a match is f itself whenever any exists, because every
            // shallower node already inherited its nearest match (S/AhoCorasickSet.java:110-121)
            bool g = want_fail && f != 0 && olen[f] > 0;
            if (own) {
                olen[c] = nodes[c].depth;
                oid[c] = nodes[c].kw;
                olink[c] = g ? f : 0;
            } else if (g) {
                olen[c] = olen[f];
                oid[c] = oid[f];
                olink[c] = olink[f];
            }
        }
    }

    // ---- 4. renumber: no-output states in BFS order, then output states in BFS order ----
    std::vector<uint32_t> newid(N);
    uint32_t n_plain = 0;
    for (uint32_t s : bfs) if (olen[s] == 0) newid[s] = n_plain++;
    t.first_out = n_plain;
    {
        uint32_t next = n_plain;
        for (uint32_t s : bfs) if (olen[s] != 0) newid[s] = next++;
    }
    t.depth.assign(N, 0); t.fail.assign(N, 0); t.out_len.assign(N, 0); t.out_link.assign(N, 0);
    t.out_id.assign(N, ~0u); t.term_id.assign(N, ~0u);
    for (uint32_t s = 0; s < N; s++) {
        uint32_t n = newid[s];
        t.depth[n] = nodes[s].depth;
        t.fail[n] = newid[fail[s]];
        t.out_len[n] = olen[s];
        t.out_link[n] = olink[s] ? newid[olink[s]] : 0;
        t.out_id[n] = oid[s];
        t.term_id[n] = nodes[s].kw;
    }

    if (mode == ACGPU_MODE_WWLONGEST) {
        // the last keyword on the path that was followed by a non-word unit, carried down the trie
        // (S/WholeWordLongestMatchSet.java:226-244): out_len = its length, out_link = its distance from the node, out_id
        std::vector<uint32_t> flen(N, 0), foff(N, 0), fid(N, ~0u);
        for (uint32_t s : bfs) {
            for (uint32_t ci = child_begin[s]; ci < child_begin[s + 1]; ci++) {
                const uint32_t c = child_ids[ci];
                if (nodes[s].kw != ~0u && !wordchar_tbl[nodes[c].unit]) {
                    flen[c] = nodes[s].depth;
                    foff[c] = 1;
                    fid[c] = nodes[s].kw;
                } else {
                    flen[c] = flen[s];
                    foff[c] = foff[s] + 1;
                    fid[c] = fid[s];
                }
            }
        }
        for (uint32_t s = 0; s < N; s++) {
            t.out_len[newid[s]] = flen[s];
            t.out_link[newid[s]] = foff[s];
            t.out_id[newid[s]] = fid[s];
        }
    }

    // ---- 5. hashed goto edges keyed by (state, folded unit) ----
    {
        uint64_t cap = 16;
        while (cap < 2 * (uint64_t)(N ? N - 1 : 0) + 2) cap <<= 1;
        t.hkeys.assign(cap, kEmptyKey);
        t.hvals.assign(cap, 0);
        t.hmask = (uint32_t)(cap - 1);
        for (uint32_t i = 1; i < N; i++) {
            uint64_t key = edge_key(newid[nodes[i].parent], nodes[i].unit);
            uint32_t slot = edge_hash(key) & t.hmask;
            while (t.hkeys[slot] != kEmptyKey) slot = (slot + 1) & t.hmask;
            t.hkeys[slot] = key;
            t.hvals[slot] = newid[i];
        }
    }

    // ---- 5b. WholeWord: hash table of whole folded keywords + paged fold table ----
    // WHOLEWORD: the table holds the keywords (payload: keyword id).  WWLONGEST: it holds the trie nodes a walk can stand on
    // when its first word ends -- paths of word characters only that are a keyword or go on with a non-word unit (payload:
    // bit 31 and the node number if the path goes on, else the keyword id); any other first word reports nothing and the walk is over (acgpu_wwlongest.hip).
    if (mode == ACGPU_MODE_WHOLEWORD || mode == ACGPU_MODE_WWLONGEST) {
        // records {payload, length, folded units}, 16-byte aligned (what keywords of more than 12 units are compared with)
        struct WwKey { uint32_t h, g, off16; };
        std::vector<WwKey> keys;
        std::vector<uint16_t> word;
        std::vector<uint8_t> allword, goes_on;
        if (mode == ACGPU_MODE_WWLONGEST) {
            allword.assign(N, 0);
            goes_on.assign(N, 0);
            allword[0] = 1;
            for (uint32_t s : bfs) // (parents before children)
                for (uint32_t ci = child_begin[s]; ci < child_begin[s + 1]; ci++) {
                    const uint32_t c = child_ids[ci];
                    const bool wc = wordchar_tbl[nodes[c].unit] != 0;
                    allword[c] = allword[s] && wc;
                    if (!wc) goes_on[s] = 1;
                }
        }
        for (uint32_t s = 1; s < N; s++) {
            uint32_t payload = nodes[s].kw;
            if (mode == ACGPU_MODE_WWLONGEST) {
                if (!allword[s] || nodes[s].depth > kWwInlineUnits || (nodes[s].kw == ~0u && !goes_on[s])) continue;
                // (a node the walk cannot leave needs no number: what it reports is its keyword)
                payload = goes_on[s] ? (newid[s] | 0x80000000u) : nodes[s].kw;
            } else if (nodes[s].kw == ~0u) {
                continue;
            }
            const uint32_t len = nodes[s].depth;
            word.resize(len);
            for (uint32_t n = s, i = len; n != 0; n = nodes[n].parent) word[--i] = nodes[n].unit;
            const uint64_t off16 = t.ww_recs.size() / 4;
            if (off16 >= kWwEmpty) return ACGPU_E_UNSUPPORTED;
            const size_t words = 2 + (len + 1) / 2;
            t.ww_recs.resize(t.ww_recs.size() + (words + 3) / 4 * 4, 0u);
            uint32_t *rec = &t.ww_recs[off16 * 4];
            rec[0] = payload;
            rec[1] = len;
            for (uint32_t i = 0; i < len; i++) rec[2 + (i >> 1)] |= (uint32_t)word[i] << (16 * (i & 1));
            keys.push_back(WwKey{0u, 0u, (uint32_t)off16});
        }
        t.ww_recs.resize(t.ww_recs.size() + 8, 0u); // the compare may read one 16-byte group past a short record
        // Bloom filter: ~16 bits per keyword, at least 1 Kbit, at most 512 Kbit (64 KB of LDS)
        uint64_t bits = 1024;
        // (WWLONGEST: 256 Kbit, so that three workgroups of its walk kernel share a CU)
        const uint64_t max_bits = mode == ACGPU_MODE_WWLONGEST ? (256u << 10) : (512u << 10);
        while (bits < 16 * (uint64_t)keys.size() && bits < max_bits && !tunables().ww_no_bloom) bits <<= 1;
        t.ww_bloom_mask = (uint32_t)(bits - 1);
        uint64_t cap = 16;
        while (cap < 2 * (uint64_t)keys.size() + 2) cap <<= 1;
        // The two-choice table the kernel probes (HostTables::ww_fat): cuckoo insertion (load factor below 1/2: a random
        // walk of evictions finds room), the table doubled if a keyword cannot be placed.  Three keywords that agree in both
        // hashes can never be placed: the hashes are then taken from the next seed.
        bool placed = false;
        const uint32_t first_attempt = (uint32_t)std::min<int64_t>(std::max<int64_t>(tunables().ww_first_seed, 0), 7);
        for (uint32_t attempt = first_attempt; attempt < 8 && !placed; attempt++) {
            t.ww_seed = kWwHashSeed + attempt * 0x9E3779B9u;
            for (WwKey &k : keys) {
                const uint32_t *rec = &t.ww_recs[(size_t)k.off16 * 4];
                const uint32_t packed = (rec[1] + 1) / 2;
                uint32_t h = t.ww_seed, g = t.ww_seed;
                for (uint32_t i = 0; i < std::max(8u, packed); i++) {
                    h = ww_hash_step(h, i < packed ? rec[2 + i] : 0u);
                    g = ww_hash2_step(g, i < packed ? rec[2 + i] : 0u);
                }
                k.h = ww_hash_final(h);
                k.g = g;
            }
            for (uint64_t fcap = cap; fcap <= 8 * cap && fcap <= (1ull << 30) && !placed; fcap <<= 1) {
                const uint32_t fmask = (uint32_t)(fcap - 1);
                std::vector<uint32_t> owner(fcap, kWwEmpty); // slot -> index of the keyword that sits there
                bool ok = true;
                uint64_t rng = 0x9E3779B97F4A7C15ull;
                for (uint32_t ki = 0; ki < keys.size() && ok; ki++) {
                    uint32_t cur = ki;
                    for (int kick = 0;; kick++) {
                        const uint32_t s1 = ww_slot1(keys[cur].h, fmask), s2 = ww_slot2(keys[cur].h, keys[cur].g, fmask);
                        if (owner[s1] == kWwEmpty) { owner[s1] = cur; break; }
                        if (owner[s2] == kWwEmpty) { owner[s2] = cur; break; }
                        if (kick >= 2048) { ok = false; break; }
                        // both taken: evict one occupant (pseudo-random choice) and place IT again
                        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                        std::swap(cur, owner[(rng >> 40) & 1 ? s1 : s2]);
                    }
                }
                if (!ok) continue;
                t.ww_fat.assign((size_t)fcap * 8, 0u);
                t.ww_fat_mask = fmask;
                for (uint64_t slot = 0; slot < fcap; slot++) {
                    if (owner[slot] == kWwEmpty) continue;
                    const WwKey &k = keys[owner[slot]];
                    const uint32_t *rec = &t.ww_recs[(size_t)k.off16 * 4];
                    uint32_t *fat = &t.ww_fat[slot * 8];
                    const uint32_t len = rec[1];
                    fat[0] = ww_tag(k.h, len);
                    fat[1] = len <= kWwInlineUnits ? rec[0] : k.off16;
                    for (uint32_t i = 0; i < kWwInlineUnits / 2; i++) fat[2 + i] = i < (len + 1) / 2 ? rec[2 + i] : 0u;
                }
                placed = true;
            }
        }
        if (!placed) return ACGPU_E_UNSUPPORTED;
        // The perfect hash (HostTables::ww_ph; WHOLEWORD, what k_ww_pp probes).  Hash and displace: buckets of about four
        // keywords, the fullest first; a bucket takes the first displacement (of 65536) that sends all of its keywords to free
        // slots.  The last, single keywords find one of the 3 % free slots within a few hundred tries.
        if (mode == ACGPU_MODE_WHOLEWORD && !tunables().ww_no_ph && !keys.empty() && t.max_len <= 32) {
            const uint64_t n = keys.size();
            const uint64_t lambda = tunables().ww_ph_lambda > 0 ? (uint64_t)tunables().ww_ph_lambda : 4;
            const uint64_t nb = (n + lambda - 1) / lambda, m = std::max<uint64_t>(n + n / 32 + 1, 8);
            if (nb * 2 <= (60u << 10) && m < (1ull << 31)) { // the displacements share LDS with the word bits, the fold pages and the waves' rings
                std::vector<uint32_t> head(nb, kWwEmpty), next(n, kWwEmpty), size(nb, 0u);
                for (uint32_t ki = 0; ki < n; ki++) {
                    const uint32_t b = ww_ph_bucket(keys[ki].h, (uint32_t)nb);
                    next[ki] = head[b];
                    head[b] = ki;
                    size[b]++;
                }
                std::vector<uint32_t> order(nb);
                for (uint32_t b = 0; b < nb; b++) order[b] = b;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return size[x] > size[y]; });
                std::vector<uint32_t> owner(m, kWwEmpty);
                std::vector<uint16_t> disp((nb + 7) / 8 * 8, 0);
                std::vector<uint32_t> pos;
                bool ok = true;
                for (uint32_t oi = 0; oi < nb && ok; oi++) {
                    const uint32_t b = order[oi];
                    if (!size[b]) break;
                    uint32_t d = 0;
                    for (; d < 65536; d++) {
                        pos.clear();
                        bool fits = true;
                        for (uint32_t ki = head[b]; ki != kWwEmpty && fits; ki = next[ki]) {
                            const uint32_t s = ww_ph_slot(keys[ki].g, keys[ki].h, d, (uint32_t)m);
                            fits = owner[s] == kWwEmpty && std::find(pos.begin(), pos.end(), s) == pos.end();
                            pos.push_back(s);
                        }
                        if (fits) break;
                    }
                    if (d == 65536) { ok = false; break; } // (two keywords of a bucket with both hashes equal, or no luck: no perfect hash)
                    disp[b] = (uint16_t)d;
                    uint32_t j = 0;
                    for (uint32_t ki = head[b]; ki != kWwEmpty; ki = next[ki]) owner[pos[j++]] = ki;
                }
                if (ok) {
                    t.ww_ph.assign((size_t)m * 8, 0u);
                    for (uint64_t slot = 0; slot < m; slot++) {
                        if (owner[slot] == kWwEmpty) continue;
                        const WwKey &k = keys[owner[slot]];
                        const uint32_t *rec = &t.ww_recs[(size_t)k.off16 * 4];
                        uint32_t *fat = &t.ww_ph[slot * 8];
                        const uint32_t len = rec[1];
                        fat[0] = ww_tag(k.h, len);
                        fat[1] = len <= kWwInlineUnits ? rec[0] : k.off16;
                        for (uint32_t i = 0; i < kWwInlineUnits / 2; i++) fat[2 + i] = i < (len + 1) / 2 ? rec[2 + i] : 0u;
                    }
                    t.ww_ph_disp = std::move(disp);
                    t.ww_ph_n = (uint32_t)m;
                    t.ww_ph_buckets = (uint32_t)nb;
                }
            }
        }
        t.ww_bloom.assign(bits / 32, 0u);
        for (const WwKey &k : keys) {
            const uint32_t b1 = ww_bloom_bit1(k.h, t.ww_bloom_mask), b2 = ww_bloom_bit2(k.h, t.ww_bloom_mask);
            t.ww_bloom[b1 >> 5] |= 1u << (b1 & 31);
            t.ww_bloom[b2 >> 5] |= 1u << (b2 & 31);
        }
    }
    // the fold table as shared pages of 256 deltas (what the word kernels keep in LDS)
    if (mode == ACGPU_MODE_WHOLEWORD || mode == ACGPU_MODE_WWLONGEST) {
        t.fold_pgidx.assign(256, 0);
        t.fold_pages.assign(256, 0); // page 0 = identity (no unit of the page folds): the kernel skips its lookup
        t.fold_n_pages = 1;
        for (uint32_t pg = 0; pg < 256; pg++) {
            uint16_t delta[256];
            for (uint32_t i = 0; i < 256; i++) delta[i] = (uint16_t)(t.lower[pg * 256 + i] - (pg * 256 + i));
            uint32_t found = t.fold_n_pages;
            for (uint32_t q = 0; q < t.fold_n_pages; q++)
                if (!std::memcmp(&t.fold_pages[q * 256], delta, sizeof(delta))) { found = q; break; }
            if (found == t.fold_n_pages) {
                t.fold_pages.insert(t.fold_pages.end(), delta, delta + 256);
                t.fold_n_pages++;
            }
            t.fold_pgidx[pg] = (uint8_t)found; // at most 256 distinct pages
        }
        t.fold_direct_n = 0;
        for (uint32_t pg = 0; pg < 8; pg++) if (t.fold_pgidx[pg] != 0) t.fold_direct_n = (pg + 1) * 256;
        // word-character bit and fold delta in one byte per unit (HostTables::ww_bp_*)
        if (mode == ACGPU_MODE_WHOLEWORD && !case_sensitive && t.fold_consistent && !tunables().ww_no_byte_pages) {
            std::vector<uint16_t> deltas;
            std::vector<uint8_t> pages, idx(256, 0);
            bool ok = true;
            for (uint32_t pg = 0; pg < 256 && ok; pg++) {
                uint8_t page[256];
                for (uint32_t i = 0; i < 256 && ok; i++) {
                    const uint32_t u = pg * 256 + i;
                    const uint16_t d = (uint16_t)(t.lower[u] - u);
                    size_t k = std::find(deltas.begin(), deltas.end(), d) - deltas.begin();
                    if (k == deltas.size()) {
                        if (deltas.size() == 128) { ok = false; break; }
                        deltas.push_back(d);
                    }
                    page[i] = (uint8_t)((k << 1) | (t.wflags[u] & 1u));
                }
                if (!ok) break;
                size_t found = pages.size() / 256;
                for (size_t q = 0; q < pages.size() / 256; q++)
                    if (!std::memcmp(&pages[q * 256], page, 256)) { found = q; break; }
                if (found == pages.size() / 256) {
                    if (found == 64) { ok = false; break; } // (what the kernel's LDS array holds)
                    pages.insert(pages.end(), page, page + 256);
                }
                idx[pg] = (uint8_t)found;
            }
            if (ok) {
                deltas.resize(128, 0);
                t.ww_bp_idx = std::move(idx);
                t.ww_bp_pages = std::move(pages);
                t.ww_bp_delta = std::move(deltas);
                t.ww_bp_n = (uint32_t)(t.ww_bp_pages.size() / 256);
            }
        }
    }

    // ---- 6. character classes + dense delta table (AC/LONGEST only) ----
    t.cls_lut.assign(65536, 0);
    t.n_cls = 1;
    t.dense = false;
    t.range_cls = false;
    std::vector<uint32_t> cls_of(65536, 0); // class of a FOLDED unit (0 = occurs in no keyword)
    if (!word_mode) {
        std::vector<uint8_t> used(65536, 0);
        uint32_t n_used = 0, minu = 65535, maxu = 0;
        for (uint32_t i = 1; i < N; i++) {
            uint16_t u = nodes[i].unit;
            if (!used[u]) { used[u] = 1; n_used++; }
            if (u < minu) minu = u;
            if (u > maxu) maxu = u;
        }
        if (n_used > 0 && n_used <= 65535) {
            if (t.cs && (maxu - minu + 1) <= 63) {
                t.range_cls = true;
                t.cls_base = minu;
                t.cls_span = maxu - minu + 1;
                t.n_cls = t.cls_span + 1;
                for (uint32_t u = minu; u <= maxu; u++) cls_of[u] = u - minu + 1;
            } else {
                uint32_t c = 0;
                for (uint32_t u = 0; u < 65536; u++) if (used[u]) cls_of[u] = ++c;
                t.n_cls = c + 1;
            }
            for (uint32_t raw = 0; raw < 65536; raw++) t.cls_lut[raw] = (uint16_t)cls_of[t.lower[raw]];
            uint64_t entries = (uint64_t)N * t.n_cls;
            uint32_t eb = (N <= 65536 && mode != ACGPU_MODE_LONGEST) ? 2 : 4;
            if (!tunables().force_sparse && entries * eb <= (uint64_t)tunables().dense_budget_bytes &&
                entries < (1ull << 32) && (mode != ACGPU_MODE_LONGEST || entries * 4 < (1ull << 31))) {
                t.dense = true;
                t.entry_bytes = eb;
                t.dfa.assign(entries, 0);
                // rows in BFS order: copy the fail row, then overwrite with own goto edges.  Root row: edges or 0.
                for (uint32_t s : bfs) {
                    uint32_t n = newid[s];
                    uint32_t *row = &t.dfa[(uint64_t)n * t.n_cls];
                    if (mode == ACGPU_MODE_LONGEST) {
                        // LONGEST walks the plain keyword trie forward from every position: goto edges only, bit 31 of an
                        // entry = "the child is the end of a keyword" (0 = no edge; the root is never a child)
                        for (uint32_t ci = child_begin[s]; ci < child_begin[s + 1]; ci++) {
                            uint32_t c = child_ids[ci];
                            row[cls_of[nodes[c].unit]] = newid[c] | (nodes[c].kw != ~0u ? 0x80000000u : 0u);
                        }
                        continue;
                    }
                    if (s != 0) {
                        const uint32_t *frow = &t.dfa[(uint64_t)newid[fail[s]] * t.n_cls];
                        std::memcpy(row, frow, sizeof(uint32_t) * t.n_cls);
                    }
                    for (uint32_t ci = child_begin[s]; ci < child_begin[s + 1]; ci++) {
                        uint32_t c = child_ids[ci];
                        row[cls_of[nodes[c].unit]] = newid[c];
                    }
                }
            }
        }
    }

    // ---- 6d. ALL / SHORTEST: dense rows for the shallow and the branching states, 16-byte nodes for the rest (k_ac_states) ----
    // Dictionaries of natural words match nearly everywhere in natural text: no filter helps, and the resolved table (a 128-byte
    // row per state, nearly all of it copied from the fail state's row) misses every cache.  Here a state below depth 3 or
    // with more than three children keeps its row; every other state is {fail, three edges}: 16 bytes, the whole automaton of the
    // reference's README dictionary in 23 MB.
    t.hy_n_states = 0;
    if ((mode == ACGPU_MODE_ALL || mode == ACGPU_MODE_SHORTEST) && t.n_cls >= 2 && t.n_cls <= 255 && N > 1 && N < kHyOut && t.max_len <= 32 &&
        !tunables().no_state_form) {
        std::vector<uint16_t> cls_unit(t.n_cls, 0);
        for (uint32_t u = 0; u < 65536; u++)
            if (cls_of[u]) cls_unit[cls_of[u]] = (uint16_t)u;
        auto is_dense = [&](uint32_t s) { return nodes[s].depth <= 2 || child_begin[s + 1] - child_begin[s] > 3; };
        // numbering: the dense group by the size of the subtree below the state (the first rows are the ones kept in LDS); the compact group in the order
        // of a depth-first walk that takes the child with the most keywords below it first -- a walk down a word's tail then
        // reads consecutive nodes (two per 32-byte sector, eight per line) instead of one line per unit
        std::vector<uint32_t> hid(N);
        uint32_t nd = 0;
        {
            std::vector<uint32_t> below(N, 1u); // nodes of the subtree: what a text of the dictionary's words visits most has most below it
            for (size_t i = bfs.size(); i-- > 1;) below[nodes[bfs[i]].parent] += below[bfs[i]];
            std::vector<uint32_t> dn;
            for (uint32_t s : bfs) if (is_dense(s)) dn.push_back(s);
            std::stable_sort(dn.begin(), dn.end(), [&](uint32_t a, uint32_t b) { return below[a] > below[b]; }); // (the root first)
            for (uint32_t s : dn) hid[s] = nd++;
            t.hy_n_dense = nd;
            std::vector<uint32_t> stack, kids;
            stack.push_back(0);
            while (!stack.empty()) {
                const uint32_t s = stack.back();
                stack.pop_back();
                if (!is_dense(s)) hid[s] = nd++;
                kids.assign(child_ids.begin() + child_begin[s], child_ids.begin() + child_begin[s + 1]);
                std::sort(kids.begin(), kids.end(), [&](uint32_t a, uint32_t b) { return below[a] != below[b] ? below[a] < below[b] : a > b; });
                for (uint32_t c : kids) stack.push_back(c); // (the heaviest child is popped first)
            }
        }
        t.hy_n_states = N;
        // what a state reports: the lengths of the keywords that end in it (bit L - 1), their number
        std::vector<uint32_t> omask(N, 0u);
        for (uint32_t s : bfs)
            if (olen[s]) omask[s] = (1u << (olen[s] - 1)) | (olink[s] ? omask[olink[s]] : 0u); // (a suffix state comes earlier in BFS order)
        auto n_out_of = [&](uint32_t target) { return (uint32_t)__builtin_popcount(omask[target]); };
        auto to = [&](uint32_t target) { return hid[target] | (olen[target] ? kHyOut : 0u); };
        t.hy_dense.assign((size_t)t.hy_n_dense * t.n_cls, 0u);
        t.hy_nodes.assign((size_t)(N - t.hy_n_dense) * 4, 0u);
        t.hy_mask.assign(N, 0u);
        t.hy_out.assign((size_t)N * 2, 0u);
        t.hy_ids.clear();
        for (uint32_t s : bfs) {
            const uint32_t h = hid[s];
            if (olen[s]) {
                t.hy_mask[h] = omask[s];
                t.hy_out[(size_t)h * 2] = omask[s];
                if ((omask[s] & (omask[s] - 1u)) == 0u && oid[s] < 0x80000000u) { // one keyword: its id itself
                    t.hy_out[(size_t)h * 2 + 1] = oid[s] | 0x80000000u;
                } else {
                    t.hy_out[(size_t)h * 2 + 1] = (uint32_t)t.hy_ids.size();
                    for (uint32_t x = s; x != 0 && olen[x]; x = olink[x]) t.hy_ids.push_back(oid[x]); // (longest first: the order of the mask's bits from the top)
                }
            }
            if (is_dense(s)) {
                uint32_t *row = &t.hy_dense[(size_t)h * t.n_cls];
                for (uint32_t c = 1; c < t.n_cls; c++) {
                    uint32_t f = s, target = 0;
                    for (;;) {
                        const uint32_t m = find_child(f, cls_unit[c]);
                        if (m != ~0u) { target = m; break; }
                        if (f == 0) break;
                        f = fail[f];
                    }
                    row[c] = target ? to(target) | (n_out_of(target) << kHyDenseCountShift) : 0u; // (at most 32 keywords end in a state)
                }
            } else {
                uint32_t *node = &t.hy_nodes[(size_t)(h - t.hy_n_dense) * 4];
                node[0] = hid[fail[s]];
                uint32_t k = 1;
                for (uint32_t ci = child_begin[s]; ci < child_begin[s + 1]; ci++, k++) {
                    const uint32_t c = child_ids[ci];
                    node[k] = (cls_of[nodes[c].unit] << 24) | to(c);
                    node[0] |= std::min(n_out_of(c), kHyNodeCountMany) << (kHyNodeCountShift + 3u * (k - 1u)); // (3 bits per edge)
                }
            }
        }
    }

    // ---- 6b. LONGEST: root table of the walk's first round (small alphabets) ----
    // Every walk starts at the root, so what the first RK units of a walk do is a function of those units alone: one byte
    // per RK-gram {bit 7: the walk is still alive after RK units | low bits: longest keyword among the RK steps}, indexed by
    // a BIT FIELD of unit codes (unit - cls_base, B bits each: B = 1 for alphabets of up to two letters, RK = 14; B = 2
    // for up to four, RK = 7: 16384 entries either way), de-interleaved: the units at even offsets of the window in the
    // low field, those at odd offsets above it -- the order in which k_longest_block's packed arithmetic produces them
    // (acgpu_longest.hip).
    t.root_b = 0;
    t.root_rk = 0;
    if (mode == ACGPU_MODE_LONGEST && t.dense && t.range_cls && t.n_cls == t.cls_span + 1 && t.cls_span <= 4) {
        const uint32_t B = t.cls_span <= 2 ? 1u : 2u, RK = 14u / B, HE = (RK + 1) / 2;
        const uint32_t total = 1u << (B * RK), cmask = (1u << B) - 1u;
        t.root_tab.resize(total);
        for (uint32_t i = 0; i < total; i++) {
            uint32_t node = 0, best = 0;
            bool alive = true;
            for (uint32_t j = 0; j < RK && alive; j++) {
                const uint32_t code = (i >> (B * ((j & 1u) ? HE + (j >> 1) : (j >> 1)))) & cmask;
                const uint32_t e = code < t.cls_span ? t.dfa[(uint64_t)node * t.n_cls + code + 1] : 0u;
                if (!e) { alive = false; break; }
                node = e & 0x7fffffffu;
                if (e >> 31) best = j + 1;
            }
            t.root_tab[i] = (uint8_t)(best | (alive ? 0x80u : 0u));
        }
        t.root_b = B;
        t.root_rk = RK;
    }

    // ---- 6c. LONGEST over a two-letter alphabet: the keyword trie, path compressed, for k_longest_bits ----
    // The text is held as ONE BIT per unit, so a stretch of the trie without branching is a bit string that a walk compares
    // with the text 31 units at a time (xor, count trailing zeros).  Entries of four words {label bits, terminal bits, meta,
    // next}: the first 2^RK entries are indexed by the RK-gram of the text itself (what round one of the root table above
    // is), an entry's label is the one-child path below its node (at most 31 units), and behind the label the walk either
    // ends (leaf), goes on with one more entry (the path is longer) or branches (next + code of the following unit: the
    // child's entry).  What does not fit kBitsTabEntries is marked "deep": the kernel walks those from the root through the
    // table in global memory.  Only for dictionaries in which every letter of the alphabet is itself a keyword: then every
    // position of a text over the alphabet starts a match, the chain's positions are the match starts, and one bitmap serves.
    t.bits_tab.clear();
    t.bits_rk = 0;
    if (mode == ACGPU_MODE_LONGEST && t.dense && t.range_cls && t.n_cls == t.cls_span + 1 && t.cls_span == 2 && !tunables().no_bits_trie) {
        bool all_letters = true;
        for (uint32_t c = 0; c < t.cls_span; c++) all_letters = all_letters && (t.dfa[c + 1] >> 31);
        if (all_letters) {
            constexpr uint32_t RK = kBitsRK, NF = 1u << RK, UL = 31, kContMax = 3;
            auto trie_edge = [&](uint32_t node, uint32_t code) -> uint32_t { return code < t.cls_span ? t.dfa[(uint64_t)node * t.n_cls + code + 1] : 0u; };
            std::vector<uint32_t> &E = t.bits_tab;
            E.assign((size_t)NF * 4, 0);
            struct Todo { uint32_t entry, node, cont; }; // the label of `entry` is the path below `node`
            std::deque<Todo> todo;
            for (uint32_t i = 0; i < NF; i++) {
                uint32_t node = 0, best = 0;
                bool alive = true;
                for (uint32_t j = 0; j < RK; j++) {
                    const uint32_t e = trie_edge(node, (i >> j) & 1u);
                    if (!e) { alive = false; break; }
                    node = e & 0x7fffffffu;
                    if (e >> 31) best = j + 1;
                }
                E[4 * i + 2] = (best << 8) | (alive ? kBitsAlive : 0u) | (alive ? 0u : 63u << 16);
                if (alive) todo.push_back({i, node, 0});
            }
            while (!todo.empty()) { // breadth first: what is near the root gets its entries first
                const Todo td = todo.front();
                todo.pop_front();
                uint32_t node = td.node, len = 0, label = 0, term = 0;
                auto children = [&](uint32_t v) { return (trie_edge(v, 0) ? 1u : 0u) + (trie_edge(v, 1) ? 1u : 0u); };
                while (len < UL && children(node) == 1) {
                    const uint32_t code = trie_edge(node, 0) ? 0u : 1u, e = trie_edge(node, code);
                    label |= code << len;
                    term |= (e >> 31) << len;
                    node = e & 0x7fffffffu;
                    len++;
                }
                uint32_t kind = kBitsLeaf, next = 0;
                const uint32_t nch = children(node);
                if (nch == 1) { // the path goes on
                    if (td.cont < kContMax && E.size() / 4 + 1 <= kBitsTabEntries) {
                        kind = kBitsCont;
                        next = (uint32_t)(E.size() / 4);
                        E.resize(E.size() + 4, 0);
                        E[4 * next + 2] = kBitsAlive;
                        todo.push_back({next, node, td.cont + 1});
                    } else kind = kBitsDeep;
                } else if (nch == 2) {
                    if (E.size() / 4 + 2 <= kBitsTabEntries) {
                        kind = kBitsJunction;
                        next = (uint32_t)(E.size() / 4);
                        E.resize(E.size() + 8, 0);
                        for (uint32_t c = 0; c < 2; c++) {
                            const uint32_t e = trie_edge(node, c);
                            E[4 * (next + c) + 2] = kBitsAlive | ((e >> 31) << 8);
                            todo.push_back({next + c, e & 0x7fffffffu, 0});
                        }
                    } else kind = kBitsDeep;
                }
                E[4 * td.entry + 0] = label;
                E[4 * td.entry + 1] = term;
                E[4 * td.entry + 2] |= len | (kind << 6) | ((kind != kBitsLeaf ? len : 63u) << 16); // (bits 16-21: bits_label)
                E[4 * td.entry + 3] = next;
            }
            E.resize((size_t)kBitsTabEntries * 4, 0);
            t.bits_rk = RK;
            // Map records: keyword ids by the keyword's own bits (HostTables::bits_idkeys), keywords of up to 32 units
            {
                struct At { uint32_t node, len, bits; };
                std::vector<At> stack{{0u, 0u, 0u}};
                std::vector<std::pair<uint64_t, uint32_t>> found;
                while (!stack.empty()) {
                    const At a = stack.back();
                    stack.pop_back();
                    if (a.len == 32) continue;
                    for (uint32_t c = 0; c < 2; c++) {
                        const uint32_t e = trie_edge(a.node, c);
                        if (!e) continue;
                        const At b{e & 0x7fffffffu, a.len + 1, a.bits | (c << a.len)};
                        if (e >> 31) found.push_back({((uint64_t)b.len << 32) | b.bits, t.term_id[b.node]});
                        stack.push_back(b);
                    }
                }
                uint64_t capk = 16;
                while (capk < 2 * (uint64_t)found.size() + 2) capk <<= 1;
                t.bits_idkeys.assign(2 * capk, kEmptyKey);
                t.bits_idmask = (uint32_t)(capk - 1);
                for (const auto &kv : found) {
                    uint32_t slot = bits_id_hash(kv.first) & t.bits_idmask;
                    while (t.bits_idkeys[2 * (size_t)slot] != kEmptyKey) slot = (slot + 1) & t.bits_idmask;
                    t.bits_idkeys[2 * (size_t)slot] = kv.first;
                    t.bits_idkeys[2 * (size_t)slot + 1] = kv.second;
                }
            }
        }
    }

    // ---- 7. suffix K-gram filter + reversed trie (ALL mode) ----
    // Filter layout: one ROW per (K-1)-gram of tile classes (the K-1 units before the last one), one BIT per class of
    // the last unit; rows are 4 bytes (n <= 32 classes) or 8 bytes (n <= 64).  A position survives iff its K-gram is
    // the K-suffix of some keyword (K <= shortest keyword, so every keyword has one).
    t.filt_k = 0;
    // (LONGEST: for dictionaries with a selective filter the matches are sparse, and leftmost-longest is a selection
    // over the all-matches list instead of a trie walk from every position)
    t.hashk = false;
    t.tile_lut = t.cls_lut;
    if ((mode == ACGPU_MODE_ALL || mode == ACGPU_MODE_SHORTEST || mode == ACGPU_MODE_LONGEST) && t.n_cls > 1 && t.min_len >= 1) {
        // tile classes: range mode -> min(unit - base, span) (other = span); LUT mode -> tile_lut (other = 0)
        // folded range classes (see HostTables::fold_range)
        t.fold_range = false;
        if (!t.cs && !t.hashk && !t.range_cls && !tunables().force_sparse) {
            uint32_t minu = 65535, maxu = 0;
            for (uint32_t i = 1; i < N; i++) {
                minu = std::min<uint32_t>(minu, nodes[i].unit);
                maxu = std::max<uint32_t>(maxu, nodes[i].unit);
            }
            const uint32_t span = maxu - minu + 1;
            bool ok = N > 1 && span <= 31;
            for (uint32_t u = minu; ok && u <= maxu; u++) ok = t.lower[u] == u; // the range folds onto itself
            if (ok) {
                std::vector<uint32_t> pre; // raw units outside the range that fold into it
                std::vector<uint32_t> votes(65536, 0);
                uint32_t delta = 0, best = 0;
                for (uint32_t raw = 0; raw < 65536; raw++) {
                    const uint32_t f = t.lower[raw];
                    if ((raw < minu || raw > maxu) && f >= minu && f <= maxu) {
                        pre.push_back(raw);
                        const uint32_t dlt = (f - raw) & 0xffffu;
                        if (++votes[dlt] > best) { best = votes[dlt]; delta = dlt; }
                    }
                }
                // partner range: unit u of it folds to u + delta (none: the range stands for itself a second time)
                uint32_t base2 = minu;
                if (!pre.empty()) {
                    const int64_t b2 = (int64_t)minu - (int64_t)(int16_t)delta;
                    ok = b2 >= 0 && b2 + span <= 65536 && (b2 + span <= minu || b2 > maxu);
                    base2 = (uint32_t)b2;
                    for (uint32_t i = 0; ok && i < span; i++) ok = t.lower[base2 + i] == minu + i;
                }
                // everything else that folds into the range must lie beyond the low zone that holds both ranges
                uint32_t bits = 0;
                while (bits < 16 && (1u << bits) <= std::max(maxu, base2 + span - 1)) bits++;
                const uint32_t himask = (0xffffu << bits) & 0xffffu;
                for (size_t i = 0; ok && i < pre.size(); i++)
                    if (pre[i] < base2 || pre[i] >= base2 + span) ok = (pre[i] & himask) != 0;
                if (ok) {
                    t.fold_range = true;
                    t.fr_base = minu; t.fr_span = span; t.fr_base2 = base2; t.fr_himask = himask;
                    t.tile_lut.assign(65536, (uint16_t)span);
                    for (uint32_t raw = 0; raw < 65536; raw++) {
                        const uint32_t f = t.lower[raw];
                        if (f >= minu && f <= maxu) t.tile_lut[raw] = (uint16_t)(f - minu);
                    }
                }
            }
        }
        // Merged stretches (round 3).  The (folded) keyword units are cut greedily into stretches of at most 31 code points; with at
        // most four ranges of raw units -- the stretches themselves and, case-insensitive, their partner ranges of the other case --
        // unit u of a range gets the class u - (start of its range): classes computed by packed arithmetic (the smallest of the
        // range classes), 4-byte rows, K up to 4 and the second level, for dictionaries that used to take the class table, 8-byte
        // rows and K = 3 (keywords in mixed case: A-Z / a-z; phrases: space and digits / letters; both).  Units of different
        // stretches share classes, so the filter is a superset test and the verification is the bucketed form's: the K units
        // themselves looked up, the walk through the unit-keyed hashed edges.  Exactness is needed for the raw units that fold to
        // a keyword unit only (any other unit may get any class: a false positive the verification drops), and it is CHECKED here
        // for all 65536 raw units; raw units beyond the low zone that holds the ranges (U+0130 -> i, U+212A -> k) go through the
        // class table, as for the folded range classes.
        // (keywords shorter than the filter's K sit beside it, see "Short keywords" below: then the LONGEST keyword bounds K)
        const bool shorts_ok = !tunables().no_short_keywords && !tunables().force_sparse;
        bool merged = false;
        std::vector<uint32_t> mg_starts; // first unit of every stretch (ascending)
        uint32_t mg_span = 0;
        auto mg_class = [&](uint32_t f) -> uint32_t { // class of a folded keyword unit: its offset in its stretch
            size_t i = std::upper_bound(mg_starts.begin(), mg_starts.end(), f) - mg_starts.begin();
            return f - mg_starts[i - 1];
        };
        if (!t.hashk && !t.fold_range && (t.range_cls ? t.n_cls > 32 : true) && (shorts_ok ? t.max_len : t.min_len) >= 2 && !tunables().force_sparse &&
            !tunables().no_merged_ranges) {
            std::vector<uint8_t> is_kw(65536, 0);
            for (uint32_t i = 1; i < N; i++) is_kw[nodes[i].unit] = 1;
            for (uint32_t u = 0; u < 65536; u++) {
                if (!is_kw[u]) continue;
                if (mg_starts.empty() || u - mg_starts.back() > 30) mg_starts.push_back(u);
                mg_span = std::max(mg_span, u - mg_starts.back() + 1);
            }
            std::vector<uint32_t> bases(mg_starts);
            merged = !bases.empty() && bases.size() <= 4;
            if (merged && !t.cs) { // partner ranges: the base that gives a raw unit the class of the unit it folds to, by votes
                std::vector<uint32_t> votes(65536, 0);
                for (uint32_t r = 0; r < 65536; r++) {
                    const uint32_t f = t.lower[r];
                    if (f == r || !is_kw[f]) continue;
                    const uint32_t c = mg_class(f);
                    if (r >= c) votes[r - c]++;
                }
                while (bases.size() < 4) {
                    uint32_t best = 0, at = 0;
                    for (uint32_t b = 0; b < 65536; b++)
                        if (votes[b] > best && std::find(bases.begin(), bases.end(), b) == bases.end()) { best = votes[b]; at = b; }
                    if (best < 4) break; // (single units that fold into a stretch from far away are left to the class table)
                    bases.push_back(at);
                }
            }
            uint32_t himask = 0;
            if (merged) {
                uint32_t top = 0, bits = 0;
                for (uint32_t b : bases) top = std::max(top, b + mg_span - 1);
                merged = top < 65535;
                while (bits < 16 && (1u << bits) <= top) bits++;
                himask = (0xffffu << bits) & 0xffffu;
                if (t.cs) himask = 0; // (nothing folds: the arithmetic is the class function for every unit)
            }
            if (merged) {
                while (bases.size() < 4) bases.push_back(bases[0]);
                t.tile_lut.assign(65536, (uint16_t)mg_span);
                for (uint32_t r = 0; r < 65536 && merged; r++) {
                    const uint32_t f = t.lower[r];
                    if (r & himask) {
                        if (is_kw[f]) t.tile_lut[r] = (uint16_t)mg_class(f);
                        continue;
                    }
                    uint32_t a = mg_span;
                    for (uint32_t b : bases) a = std::min(a, (r - b) & 0xffffu);
                    if (is_kw[f] && a != mg_class(f)) merged = false; // the ranges overlap where it matters
                    t.tile_lut[r] = (uint16_t)a;
                }
            }
            if (merged) {
                uint32_t nr = 4;
                while (nr > 1 && bases[nr - 1] == bases[0]) nr--;
                t.hashk = true; // (K-gram and edges by the units themselves)
                t.fold_range = true;
                t.fr_base = bases[0]; t.fr_base2 = bases[1]; t.fr_base3 = bases[2]; t.fr_base4 = bases[3];
                t.fr_nr = nr;
                t.fr_span = mg_span; t.fr_himask = himask;
            } else {
                t.tile_lut = t.cls_lut;
            }
        }
        std::vector<uint32_t> bucket_of; // hashk: folded unit -> bucket 1..63
        if (!merged && t.n_cls > 64) {
            // more than 63 distinct units: 63 buckets, filled round robin in order of decreasing frequency so that the
            // buckets (and with them the class K-grams) are used evenly
            t.hashk = true;
            std::vector<uint32_t> freq(65536, 0);
            for (uint32_t i = 1; i < N; i++) freq[nodes[i].unit]++;
            std::vector<uint32_t> order;
            for (uint32_t u = 0; u < 65536; u++) if (freq[u]) order.push_back(u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return freq[a] > freq[b]; });
            bucket_of.assign(65536, 0);
            for (size_t i = 0; i < order.size(); i++) bucket_of[order[i]] = 1 + (uint32_t)(i % 63);
            t.tile_lut.assign(65536, 0);
            for (uint32_t raw = 0; raw < 65536; raw++) t.tile_lut[raw] = (uint16_t)bucket_of[t.lower[raw]];
        }
        const uint32_t n = t.fold_range ? t.fr_span + 1 : t.hashk ? 64 : t.n_cls;
        t.filt_n = n;
        t.filt_other = t.fold_range ? t.fr_span : t.hashk ? 0 : t.range_cls ? t.cls_span : 0;
        t.filt_row_bytes = n <= 32 ? 4 : 8;
        auto tcls = [&](uint16_t folded_unit) -> uint32_t {
            if (merged) return mg_class(folded_unit);
            if (t.hashk) return bucket_of[folded_unit];
            if (t.fold_range) return (uint32_t)folded_unit - t.fr_base; // only called on keyword units: inside the range
            if (t.range_cls) return (uint32_t)folded_unit - t.cls_base;
            return cls_of[folded_unit];
        };
        // Short keywords (round 3).  K used to be at most the SHORTEST keyword, so one two-letter entry took a 10 k-word dictionary
        // from K = 4 to K = 2 and its scan from 0.26 to 6 ms.  Now keywords of fewer than K units (K <= 4 then: lengths 1..3) sit
        // beside the K-gram filter: a short keyword sets the filter bit of EVERY K-gram that ends with it (its missing left
        // context as a wild card), so a position passes where a long keyword's K-suffix or a whole short keyword ends; the
        // second level lets such a position through by testing the row with "other" as the leading class (only wild cards set
        // it); and the verification takes the short keywords that end at a candidate from a table indexed by the last K-1
        // classes (kshort) -- bucketed / merged classes, whose verification goes by units: from a hash table keyed by the length
        // and the units themselves (ks_keys / ks_vals).
        const uint32_t k_cap = (shorts_ok && t.min_len < 4) ? std::min<uint32_t>(t.max_len, 4u) : t.min_len;
        uint32_t K = 1;
        uint64_t rows = 1; // n^(K-1)
        while (K < k_cap && K < (merged ? 4u : t.hashk ? 3u : 8u) && rows * n * t.filt_row_bytes <= (uint64_t)tunables().filter_max_bytes &&
               rows * n * n <= (1ull << 24)) {
            rows *= n;
            K++;
        }
        t.has_short = K > t.min_len;
        t.kshort.clear();
        t.ks_keys.clear();
        t.ks_vals.clear();
        {
            // reversed trie: walking a terminal node's parent chain in the forward trie spells the reversed keyword
            struct RNode { uint32_t parent; uint16_t unit; uint32_t depth; uint32_t kw; uint32_t n_child; uint32_t only_child; };
            std::vector<RNode> rn;
            rn.push_back({0, 0, 0, ~0u, 0, 0});
            std::unordered_map<uint64_t, uint32_t> redge;
            redge.reserve(N * 2);
            for (uint32_t s = 1; s < N; s++) {
                if (nodes[s].kw == ~0u) continue;
                uint32_t cur = 0;
                for (uint32_t p = s; p != 0; p = nodes[p].parent) {
                    uint64_t key = ((uint64_t)cur << 16) | nodes[p].unit;
                    auto it = redge.find(key);
                    if (it == redge.end()) {
                        uint32_t id = (uint32_t)rn.size();
                        rn.push_back({cur, nodes[p].unit, rn[cur].depth + 1, ~0u, 0, 0});
                        rn[cur].n_child++;
                        rn[cur].only_child = id;
                        redge.emplace(key, id);
                        cur = id;
                    } else {
                        cur = it->second;
                    }
                }
                rn[cur].kw = nodes[s].kw;
            }
            const uint32_t RN = (uint32_t)rn.size();
            if (RN <= kRefIdMask) {
                t.n_rstates = RN;
                t.rterm.assign(RN, ~0u);
                // flagged reference: children / terminal flags, and for a node with exactly one child the class of that
                // child's unit (+1) as a hint: a walk whose next unit has another class stops without touching memory
                auto ref = [&](uint32_t i) -> uint32_t {
                    uint32_t r = i | (rn[i].n_child ? kRefHasChildren : 0u) | (rn[i].kw != ~0u ? kRefTerminal : 0u);
                    if (rn[i].n_child == 1) {
                        uint32_t c = tcls(rn[rn[i].only_child].unit);
                        if (c + 1 < 64) r |= (c + 1) << kRefHintShift;
                    }
                    return r;
                };
                for (uint32_t i = 0; i < RN; i++) t.rterm[i] = rn[i].kw;
                // (bucketed classes do not name a unit: only the hashed, unit-keyed edges are exact)
                t.rdense = !t.hashk && (uint64_t)RN * n * 4 <= (uint64_t)tunables().rdense_budget_bytes && !tunables().force_sparse;
                if (t.rdense) {
                    t.rtab.assign((size_t)RN * n, 0);
                    for (uint32_t i = 1; i < RN; i++) t.rtab[(size_t)rn[i].parent * n + tcls(rn[i].unit)] = ref(i);
                    t.rhkeys.assign(16, kEmptyKey);
                    t.rhvals.assign(16, 0);
                    t.rhmask = 15;
                } else {
                    uint64_t cap = 16;
                    while (cap < 2 * (uint64_t)RN + 2) cap <<= 1;
                    t.rhkeys.assign(cap, kEmptyKey);
                    t.rhvals.assign(cap, 0);
                    t.rhmask = (uint32_t)(cap - 1);
                    for (uint32_t i = 1; i < RN; i++) {
                        uint64_t key = edge_key(rn[i].parent, rn[i].unit);
                        uint32_t slot = edge_hash(key) & t.rhmask;
                        while (t.rhkeys[slot] != kEmptyKey) slot = (slot + 1) & t.rhmask;
                        t.rhkeys[slot] = key;
                        t.rhvals[slot] = ref(i);
                    }
                }
                // rows of the filter and K-gram -> depth-K reverse node (index: last unit least significant)
                t.filt_bits.assign(rows * (t.filt_row_bytes / 4), 0);
                if (!t.hashk) t.kgram_node.assign(rows * n * 2, 0); // pairs: {ref of the depth-K node, ref of its only child or 0}
                std::vector<std::pair<uint64_t, uint32_t>> kg; // hashk: packed K units (text order) -> ref
                uint64_t n_set = 0;
                std::vector<uint32_t> path;
                for (uint32_t i = 1; i < RN; i++) {
                    if (rn[i].depth != K) continue;
                    // root -> i spells text[e-1], text[e-2], ..., text[e-K]; walking up from i meets them deepest first
                    path.clear();
                    for (uint32_t p = i; p != 0; p = rn[p].parent) path.push_back(tcls(rn[p].unit)); // [text[e-K] .. text[e-1]]
                    uint64_t hi = 0;
                    for (size_t j = 0; j + 1 < path.size(); j++) hi = hi * n + path[j];
                    const uint32_t last = path.back();
                    uint32_t *word = t.filt_row_bytes == 4 ? &t.filt_bits[hi] : &t.filt_bits[hi * 2 + (last >> 5)];
                    const uint32_t bit = 1u << (last & 31);
                    if (!(*word & bit)) n_set++; // (several K-grams of units can share one K-gram of buckets)
                    *word |= bit;
                    if (t.hashk) {
                        uint64_t key = 0;
                        // walking up from i meets text[e-K] first: text order, leftmost unit in the lowest 16 bits (K <= 3, or
                        // K = 4 over units below 0xffff: a key never equals the all-ones empty marker)
                        for (uint32_t p = i, sh = 0; p != 0; p = rn[p].parent, sh += 16) key |= (uint64_t)rn[p].unit << sh;
                        kg.emplace_back(key, ref(i));
                    } else {
                        t.kgram_node[(hi * n + last) * 2] = ref(i);
                        t.kgram_node[(hi * n + last) * 2 + 1] = rn[i].n_child == 1 ? ref(rn[i].only_child) : 0u;
                    }
                }
                if (t.has_short) {
                    // kshort[g], g = index of the last K-1 classes (oldest most significant): {node + 1 of the keyword that is the
                    // last unit, the last two, the last three, 0}; and the wild-card bits of the filter
                    if (!t.hashk) t.kshort.assign(rows * 4, 0u);
                    std::vector<std::pair<uint64_t, uint32_t>> ks; // hashk: (length << 48 | folded units, leftmost lowest) -> node + 1
                    std::vector<uint32_t> sc; // a short keyword's classes in text order
                    for (uint32_t i = 1; i < RN; i++) {
                        const uint32_t Ls = rn[i].depth;
                        if (Ls >= K || rn[i].kw == ~0u) continue;
                        sc.assign(Ls, 0u);
                        for (uint32_t p = i, j = 0; p != 0; p = rn[p].parent, j++) sc[j] = tcls(rn[p].unit); // walking up: text order
                        uint64_t body = 0, tail = 0; // the keyword without its last class / whole, as digits
                        for (uint32_t j = 0; j + 1 < Ls; j++) body = body * n + sc[j];
                        for (uint32_t j = 0; j < Ls; j++) tail = tail * n + sc[j];
                        uint64_t pw_body = 1, pw_tail = 1, free_rows = 1, free_grams = 1;
                        for (uint32_t j = 0; j + 1 < Ls; j++) pw_body *= n;
                        for (uint32_t j = 0; j < Ls; j++) pw_tail *= n;
                        for (uint32_t j = 0; j < K - Ls; j++) free_rows *= n;      // rows: K-1 classes, the last Ls-1 fixed
                        for (uint32_t j = 0; j + 1 < K - Ls; j++) free_grams *= n; // kshort: K-1 classes, the last Ls fixed
                        const uint32_t last = sc[Ls - 1];
                        for (uint64_t f = 0; f < free_rows; f++) {
                            const uint64_t hi = f * pw_body + body;
                            uint32_t *word = t.filt_row_bytes == 4 ? &t.filt_bits[hi] : &t.filt_bits[hi * 2 + (last >> 5)];
                            const uint32_t bit = 1u << (last & 31);
                            if (!(*word & bit)) n_set++;
                            *word |= bit;
                        }
                        if (!t.hashk) {
                            for (uint64_t f = 0; f < free_grams; f++) t.kshort[(f * pw_tail + tail) * 4 + (Ls - 1)] = i + 1;
                        } else {
                            uint64_t key = (uint64_t)Ls << 48;
                            for (uint32_t p = i, shift = 0; p != 0; p = rn[p].parent, shift += 16) key |= (uint64_t)rn[p].unit << shift;
                            ks.emplace_back(key, i + 1);
                        }
                    }
                    if (t.hashk) {
                        uint64_t cap = 16;
                        while (cap < 2 * (uint64_t)ks.size() + 2) cap <<= 1;
                        t.ks_keys.assign(cap, kEmptyKey);
                        t.ks_vals.assign(cap, 0);
                        t.ks_mask = (uint32_t)(cap - 1);
                        for (auto &kv : ks) {
                            uint32_t slot = edge_hash(kv.first) & t.ks_mask;
                            while (t.ks_keys[slot] != kEmptyKey) slot = (slot + 1) & t.ks_mask;
                            t.ks_keys[slot] = kv.first;
                            t.ks_vals[slot] = kv.second;
                        }
                    }
                }
                // second-level filter (range classes below 32, K <= 5): every reverse node of depth D, and every terminal
                // node of a depth in [K, D), is a key
                t.l2_depth = 0;
                t.l2_bloom.clear();
                t.l2_big.clear();
                if ((!t.hashk || merged) && (t.range_cls || t.fold_range) && n <= 32 && K >= 2 && K <= 5) {
                    const uint32_t D = std::min<uint32_t>(K + 2, 6);
                    t.l2_bloom.assign(kL2Words, 0);
                    for (uint32_t i = 1; i < RN; i++) {
                        const uint32_t L = rn[i].depth;
                        if (L < K || L > D || (L < D && rn[i].kw == ~0u)) continue;
                        uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // c[j] = class of text[e-1-j]
                        for (uint32_t p = i; p != 0; p = rn[p].parent) c[rn[p].depth - 1] = tcls(rn[p].unit);
                        const uint32_t h = l2_hash(l2_gram(c, K));
                        t.l2_bloom[l2_word(h)] |= l2_rotr(l2_pattern(h), l2_rot(c, L, K));
                    }
                    uint64_t set = 0;
                    for (uint32_t w : t.l2_bloom) set += (uint64_t)__builtin_popcount(w);
                    t.l2_density = (double)set / (32.0 * kL2Words);
                    t.l2_depth = D;
                    // a saturated filter rejects nothing (100 k keywords set 60 % of its bits): the same keys in 2 MB
                    if (t.l2_density > 0.25 && K == 4 && !tunables().no_big_l2) {
                        t.l2_big.assign(kL2BigWords, 0);
                        for (uint32_t i = 1; i < RN; i++) {
                            const uint32_t L = rn[i].depth;
                            if (L < K || L > D || (L < D && rn[i].kw == ~0u)) continue;
                            uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                            for (uint32_t p = i; p != 0; p = rn[p].parent) c[rn[p].depth - 1] = tcls(rn[p].unit);
                            const uint32_t h = l2_hash(l2_gram(c, K));
                            t.l2_big[l2_word_big(h)] |= l2_rotr(l2_pattern(h), l2_rot(c, L, K));
                        }
                    }
                }
                if (t.hashk) {
                    uint64_t cap = 16;
                    while (cap < 2 * (uint64_t)kg.size() + 2) cap <<= 1;
                    t.kg_keys.assign(cap, kEmptyKey);
                    t.kg_vals.assign(cap, 0);
                    t.kg_mask = (uint32_t)(cap - 1);
                    for (auto &kv : kg) {
                        uint32_t slot = edge_hash(kv.first) & t.kg_mask;
                        while (t.kg_keys[slot] != kEmptyKey) slot = (slot + 1) & t.kg_mask;
                        t.kg_keys[slot] = kv.first;
                        t.kg_vals[slot] = kv.second;
                    }
                }
                double denom = 1;
                for (uint32_t j = 0; j < K; j++) denom *= (double)(n > 1 ? n - 1 : 1);
                t.filt_density = (double)n_set / denom;
                t.filt_k = K;
            }
        }
    }
    // ---- 7b. the class table of the tile kernel's LUT forms as PAGES for LDS ----
    // tile_lut is 128 KB: a class looked up there is a 2-byte gather from a table eight times a CU's L1 -- a 128-byte line
    // through the L1 fill path for every unit of the text, which is what bounded the class-table forms.  A
    // dictionary's units sit in a few 256-unit pages: [256-byte page index][distinct pages, one byte per unit], copied to
    // LDS behind the filter rows when it fits there (k_ac_tile).
    t.cls_pages.clear();
    if (t.filt_k >= 1 && !t.range_cls && !tunables().no_class_pages) {
        bool bytes_ok = true;
        for (uint32_t raw = 0; raw < 65536 && bytes_ok; raw++) bytes_ok = t.tile_lut[raw] < 256;
        if (bytes_ok) {
            t.cls_pages.assign(256, 0);
            uint32_t n_pages = 0;
            for (uint32_t pg = 0; pg < 256; pg++) {
                uint32_t same = n_pages;
                for (uint32_t q = 0; q < n_pages && same == n_pages; q++) {
                    bool eq = true;
                    for (uint32_t i = 0; i < 256 && eq; i++) eq = t.cls_pages[256 + q * 256 + i] == (uint8_t)t.tile_lut[pg * 256 + i];
                    if (eq) same = q;
                }
                if (same == n_pages) {
                    for (uint32_t i = 0; i < 256; i++) t.cls_pages.push_back((uint8_t)t.tile_lut[pg * 256 + i]);
                    n_pages++;
                }
                t.cls_pages[pg] = (uint8_t)same;
            }
        }
    }
    // ---- 7c. the same for the DFA chunk scan: cls_lut (any number of classes: 16-bit entries) as pages for k_ac_dfa ----
    t.dfa_pages.clear();
    if ((t.dense || t.hy_n_states) && !t.range_cls && !tunables().no_class_pages) {
        t.dfa_pages.assign(128, 0); // (uint16 words: the 256-byte index, one byte per page, first)
        uint8_t *index = reinterpret_cast<uint8_t *>(t.dfa_pages.data());
        uint32_t n_pages = 0;
        for (uint32_t pg = 0; pg < 256; pg++) {
            uint32_t same = n_pages;
            for (uint32_t q = 0; q < n_pages && same == n_pages; q++)
                if (std::memcmp(&t.dfa_pages[128 + (size_t)q * 256], &t.cls_lut[pg * 256], 512) == 0) same = q;
            if (same == n_pages) {
                t.dfa_pages.insert(t.dfa_pages.end(), t.cls_lut.begin() + pg * 256, t.cls_lut.begin() + pg * 256 + 256);
                index = reinterpret_cast<uint8_t *>(t.dfa_pages.data());
                n_pages++;
            }
            index[pg] = (uint8_t)same;
        }
        if (t.dfa_pages.size() * 2 > 32 * 1024) t.dfa_pages.clear(); // (units all over the plane: the table in global memory it is)
    }
    return ACGPU_OK;
}

} // namespace acgpu
