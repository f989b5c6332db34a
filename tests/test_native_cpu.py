"""CPU-side checks of the native library: it loads, exports every symbol include/acgpu.h declares, the host
builder's tables reproduce the oracle when *simulated in this test* (test code -- the product has no CPU
matcher), and match calls fail loudly without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

from ahocorasick_amd import _native as N
from ahocorasick_amd.strings import Automaton, IllegalArgumentException, utf16
from oracle.oracle import FAM_AC, Oracle
from tests.helpers import LOWER, WORD, fixture_inputs, rand_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "acgpu.h")).read()
    declared = set(re.findall(r"\b(acgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    L = ctypes.CDLL(N.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert N.lib().acgpu_abi_version() == N.ABI_VERSION == int(re.search(r"#define ACGPU_ABI_VERSION (\d+)", hdr).group(1))


def _tables(a):
    info = a.info()
    ns, nc = info["n_states"], info["n_classes"]
    cls = np.zeros(65536, np.uint16)
    dfa = np.zeros(ns * nc, np.uint32) if info["dense"] else None
    out_len, out_link, out_id, depth = (np.zeros(ns, np.uint32) for _ in range(4))
    first = ctypes.c_uint32(0)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p) if x is not None else None
    N.check(N.lib().acgpu_debug_tables(a.handle, vp(cls), vp(dfa), vp(out_len), vp(out_link), vp(out_id), vp(depth),
                                       ctypes.byref(first)), "debug_tables")
    return info, cls, (dfa.reshape(ns, nc) if dfa is not None else None), out_len, out_link, out_id, depth, first.value


def _simulate_all(a, hay):
    """Test-only DFA walk over the builder's tables."""
    info, cls, dfa, out_len, out_link, out_id, depth, first = _tables(a)
    s = 0
    out = []
    for i, u in enumerate(hay.tolist()):
        s = int(dfa[s, cls[u]])
        if s >= first:
            t = s
            while t:
                out.append([i + 1 - int(out_len[t]), i + 1, int(out_id[t])])
                t = int(out_link[t])
    return out


def test_builder_tables_reproduce_oracle_on_fixtures(fixtures):
    for fx in fixtures:
        if "keywords_gen" in fx:
            continue  # 65536 classes: built sparse, covered on the GPU
        hay, kws = fixture_inputs(fx)
        a = Automaton(N.MODE_ALL, kws, True)
        assert _simulate_all(a, hay) == fx["AC"], fx["name"]


@pytest.mark.parametrize("seed", range(4))
def test_builder_tables_fuzz(seed):
    rng = np.random.default_rng(seed)
    alpha = [ord(c) for c in "abcAB"] + [0x00E9, 0x00C9]
    for _ in range(20):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 15)), 6, int(rng.integers(0, 300)))
        for cs in (True, False):
            a = Automaton(N.MODE_ALL, kws, cs)
            want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            assert _simulate_all(a, hay) == want


def test_builder_numbering_and_info():
    a = Automaton(N.MODE_ALL, ["he", "she", "his", "hers"], True)
    info, cls, dfa, out_len, out_link, out_id, depth, first = _tables(a)
    assert info["n_states"] == 10 and info["n_keywords"] == 4
    assert info["min_keyword_len"] == 2 and info["max_keyword_len"] == 4
    assert info["dense"] == 1 and info["entry_bytes"] == 2
    # no-output states first, output states after: the has-output test is "state >= first"
    assert (out_len[:first] == 0).all() and (out_len[first:] > 0).all()
    assert depth[0] == 0 and (np.diff(depth[:first]) >= 0).all()


def test_wholeword_ctor_rejects_nonword_keywords():
    with pytest.raises(IllegalArgumentException):
        Automaton(N.MODE_WHOLEWORD, ["A B"], True, word_chars=WORD)
    a = Automaton(N.MODE_WHOLEWORD, [" abc,", "", None, "de"], True, word_chars=WORD)
    assert a.info()["n_keywords"] == 2 and a.info()["fold_consistent"] == 1


def test_full_alphabet_dictionary_builds_sparse():
    kws = [np.array([i], dtype=np.uint16) for i in range(65536)]
    a = Automaton(N.MODE_ALL, kws, True)
    info = a.info()
    assert info["n_states"] == 65537 and info["dense"] == 0


def test_match_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    a = Automaton(N.MODE_ALL, ["ab"], True)
    with pytest.raises(N.AcgpuError):
        a.match_host(utf16("zabz"), with_ids=True)


def test_abi_argument_checking_without_device():
    L = N.lib()
    h = ctypes.c_void_p()
    bad = ctypes.c_int64(-1)
    units = np.array([97, 98], np.uint16)
    off = np.array([0, 2], np.uint64)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    # unknown mode, missing fold table for case-insensitive, missing word-char table for WHOLEWORD
    assert L.acgpu_build(7, vp(units), vp(off), 1, 1, None, None, ctypes.byref(h), ctypes.byref(bad)) == N.E_INVALID
    assert L.acgpu_build(N.MODE_ALL, vp(units), vp(off), 1, 0, None, None, ctypes.byref(h), ctypes.byref(bad)) == N.E_INVALID
    assert L.acgpu_build(N.MODE_WHOLEWORD, vp(units), vp(off), 1, 1, None, None, ctypes.byref(h), ctypes.byref(bad)) == N.E_INVALID
    assert L.acgpu_build(N.MODE_ALL, vp(units), vp(off), 1, 1, None, None, None, None) == N.E_INVALID
    # a good build; info, strerror, unknown tunable
    assert L.acgpu_build(N.MODE_ALL, vp(units), vp(off), 1, 1, None, None, ctypes.byref(h), ctypes.byref(bad)) == N.OK
    info = N.Info()
    assert L.acgpu_get_info(h, ctypes.byref(info)) == N.OK and info.n_states == 3 and info.max_keyword_len == 2
    assert L.acgpu_get_info(None, ctypes.byref(info)) == N.E_INVALID
    assert L.acgpu_strerror(N.E_OVERFLOW) == b"output capacity too small"
    assert L.acgpu_set_tunable(b"no_such_knob", 1) == -1
    n_out = ctypes.c_uint64(0)
    assert L.acgpu_match_u16(h, None, 5, N.REC_MAP, None, 0, ctypes.byref(n_out)) == N.E_INVALID
    assert L.acgpu_match_u16(h, vp(units), 2, 5, None, 0, ctypes.byref(n_out)) == N.E_INVALID  # bad record kind
    L.acgpu_free(h)
    L.acgpu_free(None)


def test_longest_and_wholeword_automata_build_on_cpu():
    from ahocorasick_amd import synth
    a = Automaton(N.MODE_LONGEST, synth.prefix_closed_keywords(1004, 3000, word_len=200), True).info()
    assert a["n_states"] == 3001 and a["dense"] == 1 and a["entry_bytes"] == 4  # the forward trie, not a reversed automaton
    words = synth.mixed_script_words(1005, 2000)
    w = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD).info()
    assert w["n_keywords"] <= 2000 and w["fold_consistent"] == 1 and w["fold_clean"] == 1
    wc = np.zeros(65536, np.uint8)
    wc[ord("A")] = 1
    i = Automaton(N.MODE_WHOLEWORD, ["A"], False, word_chars=wc).info()  # 'A' is a word character, its folded form 'a' is not
    assert i["fold_consistent"] == 0 and i["fold_clean"] == 0
    wc[ord("a")], wc[ord("B")] = 1, 1  # now only 'b' is missing: the keyword "A" folds to word characters
    i = Automaton(N.MODE_WHOLEWORD, ["A"], False, word_chars=wc).info()
    assert i["fold_consistent"] == 0 and i["fold_clean"] == 1


# ---- WholeWord: the whole-keyword hash table + paged fold table (what k_ww_tile probes) ----------------------------

def _wordhash_tables(a):
    n_slots, n_pages, n_words, seed = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint32(0)
    f = N.lib().acgpu_debug_wordhash
    N.check(f(a.handle, ctypes.byref(n_slots), None, ctypes.byref(n_words), None, None, ctypes.byref(n_pages), None,
              ctypes.byref(seed)), "sizes")
    slots = np.zeros(8 * n_slots.value, np.uint32)
    recs = np.zeros(n_words.value, np.uint32)
    pgidx = np.zeros(256, np.uint8)
    pages = np.zeros(max(n_pages.value, 1) * 256, np.uint16)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    N.check(f(a.handle, None, vp(slots), None, vp(recs), vp(pgidx), None, vp(pages), None), "tables")
    return slots.reshape(-1, 8), recs, pgidx, pages, seed.value


def _simulate_wholeword(a, hay, word, cs):
    """Test-only restatement of the kernel's verification: maximal word runs, folded through the paged table, hashed twice
    over the packed units (h*33 + w with the murmur finaliser; rotl(g,5) ^ w), looked up in the two slots the hashes name --
    tag (hash | length), the first 12 units inline, longer keywords through their record."""
    slots, recs, pgidx, pages, seed = _wordhash_tables(a)
    mask = len(slots) - 1
    fold = (lambda u: u) if cs else (lambda u: (u + int(pages[int(pgidx[u >> 8]) * 256 + (u & 255)])) & 0xffff)
    out, i, n = [], 0, len(hay)
    h_list = hay.tolist()
    while i < n:
        if not word[h_list[i]]:
            i += 1
            continue
        j = i
        while j < n and word[h_list[j]]:
            j += 1
        f = [fold(u) for u in h_list[i:j]]
        packed = [f[k] | ((f[k + 1] if k + 1 < len(f) else 0) << 16) for k in range(0, len(f), 2)]
        packed += [0] * (8 - len(packed))
        h = g = seed
        for d in packed:
            h = (h * 33 + d) & 0xffffffff
            g = (((g << 5) | (g >> 27)) & 0xffffffff) ^ d
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & 0xffffffff
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xffffffff
        h ^= h >> 16
        tag = (h & 0xffffff00) | min(len(f), 255)
        s1 = h & mask
        s2 = ((((g ^ (h >> 16) ^ (g >> 13)) * 0x2C1B3C6D) & 0xffffffff) >> 11) & mask
        if s2 == s1:
            s2 ^= 1
        for s in (s1, s2):
            e = [int(x) for x in slots[s]]
            if e[0] != tag or e[2:8] != (packed + [0] * 6)[:6]:
                continue
            if len(f) <= 12:
                out.append([i, j, e[1]])
                break
            off = e[1] * 4
            ln = int(recs[off + 1])
            units = [(int(recs[off + 2 + (k >> 1)]) >> (16 * (k & 1))) & 0xffff for k in range(ln)]
            if units == f:
                out.append([i, j, int(recs[off])])
                break
        i = j
    return out


def test_wordhash_fold_pages_equal_the_fold_table():
    a = Automaton(N.MODE_WHOLEWORD, ["a"], False, word_chars=WORD)
    _, _, pgidx, pages, _ = _wordhash_tables(a)
    u = np.arange(65536)
    got = (u + pages[pgidx[u >> 8].astype(np.int64) * 256 + (u & 255)]) & 0xffff
    assert (got == LOWER).all()
    assert len(pages) // 256 <= 64  # fits the kernel's LDS budget (Unicode 13 simple lower-casing: 18 pages)


def test_wordhash_tables_reproduce_oracle(fixtures):
    from oracle.oracle import FAM_WHOLEWORD
    for fx in fixtures:
        if fx["WW"] == "IllegalArgumentException":
            continue
        hay, kws = fixture_inputs(fx)
        a = Automaton(N.MODE_WHOLEWORD, kws, True, word_chars=WORD)
        assert _simulate_wholeword(a, hay, WORD, True) == fx["WW"], fx["name"]
    rng = np.random.default_rng(5)
    alpha = [ord(c) for c in "abAB -_."] + [0x00E9, 0x00C9, 0x0130, 0x3002]
    for it in range(40):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 40)), int(rng.choice([3, 6, 20])), int(rng.integers(0, 400)))
        kws = [k for k in kws if all(WORD[c] for c in k.tolist())] or [np.array([97], np.uint16)]
        for cs in (True, False):
            a = Automaton(N.MODE_WHOLEWORD, kws, cs, word_chars=WORD)
            want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay).tolist()
            assert _simulate_wholeword(a, hay, WORD, cs) == want, (it, cs)


def test_wordhash_two_choice_table_holds_config5_dictionary():
    """Config 5's 100 k mixed-script words: the primary hash has triples of keywords with one 32-bit value (h*33 + w is linear
    modulo 2^32 and the alphabets are dense), which a two-choice table can only place because the second slot comes from the
    second hash.  Every keyword must be found where the kernel will look, non-keywords must not."""
    from ahocorasick_amd import synth
    from oracle.oracle import FAM_WHOLEWORD
    words = synth.mixed_script_words(1005, 100000)
    a = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD)
    slots, _, _, _, seed = _wordhash_tables(a)
    used = int((slots[:, 0] != 0).sum())
    assert used == a.info()["n_keywords"] and len(slots) >= 2 * used and seed == 0x811C9DC5
    rng = np.random.default_rng(8)
    pick = rng.choice(len(words), 3000, replace=False)
    sep = np.array([32], np.uint16)
    toks = []
    for i in pick:
        toks += [np.asarray(words[int(i)], dtype=np.uint16), sep, np.asarray(words[int(i)], dtype=np.uint16)[::-1].copy(), sep]
    hay = np.concatenate(toks)
    got = _simulate_wholeword(a, hay, WORD, False)
    want = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay).tolist()
    assert len(want) >= 3000 and got == want


# ---- WholeWord: the perfect hash and the byte pages (what k_ww_pp probes and looks units up in) -----------------------------------

def _perfect_tables(a):
    sizes = (ctypes.c_uint32 * 3)()
    f = N.lib().acgpu_debug_wordhash_perfect
    N.check(f(a.handle, sizes, None, None, None, None, None), "sizes")
    n_slots, n_buckets, n_pages = int(sizes[0]), int(sizes[1]), int(sizes[2])
    slots = np.zeros(8 * max(n_slots, 1), np.uint32)
    disp = np.zeros(max(n_buckets, 1), np.uint16)
    idx, pages, delta = np.zeros(256, np.uint8), np.zeros(max(n_pages, 1) * 256, np.uint8), np.zeros(128, np.uint16)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    N.check(f(a.handle, sizes, vp(slots), vp(disp), vp(idx), vp(pages), vp(delta)), "tables")
    return n_slots, n_buckets, n_pages, slots.reshape(-1, 8), disp, idx, pages, delta


def _simulate_wholeword_perfect(a, hay, cs):
    """Test-only restatement of k_ww_pp's lookups: word character and fold of a unit from the byte pages (case-insensitive)
    or the word table (case-sensitive), the run's two hashes, ONE slot of the perfect hash (acgpu.h: acgpu_debug_wordhash_perfect)."""
    n_slots, n_buckets, n_pages, slots, disp, idx, pages, delta = _perfect_tables(a)
    _, recs, _, _, seed = _wordhash_tables(a)
    assert n_slots > 0
    if cs:
        word, fold = (lambda u: bool(WORD[u])), (lambda u: u)
    else:
        assert n_pages > 0
        ent = lambda u: int(pages[int(idx[u >> 8]) * 256 + (u & 255)])
        word, fold = (lambda u: bool(ent(u) & 1)), (lambda u: (u + int(delta[ent(u) >> 1])) & 0xffff)
    out, i, n = [], 0, len(hay)
    h_list = hay.tolist()
    M = 0xffffffff
    while i < n:
        if not word(h_list[i]):
            i += 1
            continue
        j = i
        while j < n and word(h_list[j]):
            j += 1
        f = [fold(u) for u in h_list[i:j]]
        packed = [f[k] | ((f[k + 1] if k + 1 < len(f) else 0) << 16) for k in range(0, len(f), 2)]
        packed += [0] * (8 - len(packed))
        h = g = seed
        for d in packed:
            h = (h * 33 + d) & M
            g = (((g << 5) | (g >> 27)) & M) ^ d
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & M
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & M
        h ^= h >> 16
        tag = (h & 0xffffff00) | min(len(f), 255)
        d = int(disp[(h * n_buckets) >> 32])
        t = ((g ^ ((h << 7) & M)) + d * 0x9E3779B9) & M
        t ^= t >> 15
        t = (t * 0x2C1B3C6D) & M
        t ^= t >> 13
        e = [int(x) for x in slots[(t * n_slots) >> 32]]
        if e[0] == tag and e[2:8] == (packed + [0] * 6)[:6]:
            if len(f) <= 12:
                out.append([i, j, e[1]])
            else:
                off = e[1] * 4
                ln = int(recs[off + 1])
                units = [(int(recs[off + 2 + (k >> 1)]) >> (16 * (k & 1))) & 0xffff for k in range(ln)]
                if units == f:
                    out.append([i, j, int(recs[off])])
        i = j
    return out


def test_byte_pages_equal_the_word_table_and_the_fold_table():
    a = Automaton(N.MODE_WHOLEWORD, ["a"], False, word_chars=WORD)
    _, _, n_pages, _, _, idx, pages, delta = _perfect_tables(a)
    u = np.arange(65536)
    e = pages[idx[u >> 8].astype(np.int64) * 256 + (u & 255)].astype(np.int64)
    assert ((e & 1) == (np.asarray(WORD) != 0)).all()
    assert (((u + delta[e >> 1]) & 0xffff) == LOWER).all()
    assert 0 < n_pages <= 64  # (Unicode's simple lower-casing with isLetterOrDigit: 54 pages, 79 deltas)
    # a case-sensitive automaton, or a table that is not fold-consistent, has none
    assert _perfect_tables(Automaton(N.MODE_WHOLEWORD, ["a"], True, word_chars=WORD))[2] == 0
    wc = np.asarray(WORD).copy()
    wc[ord("A")] = 0
    assert _perfect_tables(Automaton(N.MODE_WHOLEWORD, ["b"], False, word_chars=wc))[2] == 0


def test_perfect_hash_tables_reproduce_oracle(fixtures):
    from oracle.oracle import FAM_WHOLEWORD
    for fx in fixtures:
        if fx["WW"] == "IllegalArgumentException":
            continue
        hay, kws = fixture_inputs(fx)
        if max(len(k) for k in kws) > 32:
            continue  # (the perfect hash is k_ww_pp's: keywords of at most 32 units)
        a = Automaton(N.MODE_WHOLEWORD, kws, True, word_chars=WORD)
        assert _simulate_wholeword_perfect(a, hay, True) == fx["WW"], fx["name"]
    rng = np.random.default_rng(6)
    alpha = [ord(c) for c in "abAB -_."] + [0x00E9, 0x00C9, 0x0130, 0x3002]
    for it in range(40):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 60)), int(rng.choice([3, 6, 20, 32])), int(rng.integers(0, 400)))
        kws = [k for k in kws if all(WORD[c] for c in k.tolist())] or [np.array([97], np.uint16)]
        for cs in (True, False):
            a = Automaton(N.MODE_WHOLEWORD, kws, cs, word_chars=WORD)
            want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay).tolist()
            assert _simulate_wholeword_perfect(a, hay, cs) == want, (it, cs)


def test_perfect_hash_holds_config5_dictionary_in_a_table_that_fits_the_l2_cache():
    """Config 5's 100 k mixed-script words: 103 126 slots of 32 bytes (3.3 MB: an XCD's L2 holds it, where the two-choice table
    has 8.4 MB), 25 000 displacements (50 KB of LDS).  Every keyword sits where its hashes say, every slot holds at most one."""
    from ahocorasick_amd import synth
    from oracle.oracle import FAM_WHOLEWORD
    words = synth.mixed_script_words(1005, 100000)
    a = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD)
    n_slots, n_buckets, _, slots, _, _, _, _ = _perfect_tables(a)
    n_kw = a.info()["n_keywords"]
    assert int((slots[:, 0] != 0).sum()) == n_kw and n_kw < n_slots <= n_kw + n_kw // 32 + 8 and n_slots * 32 < (4 << 20)
    assert n_buckets * 2 <= 60 * 1024 and (n_kw + 3) // 4 == n_buckets
    rng = np.random.default_rng(9)
    pick = rng.choice(len(words), 3000, replace=False)
    sep = np.array([32], np.uint16)
    toks = []
    for i in pick:
        toks += [np.asarray(words[int(i)], dtype=np.uint16), sep, np.asarray(words[int(i)], dtype=np.uint16)[::-1].copy(), sep]
    hay = np.concatenate(toks)
    got = _simulate_wholeword_perfect(a, hay, False)
    want = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay).tolist()
    assert len(want) >= 3000 and got == want


def test_perfect_hash_is_left_out_where_it_cannot_be_built():
    # more keywords than 16-bit displacements fit LDS for (4 per bucket: beyond 122 880); buckets too large to place; keywords beyond 32 units
    N.set_tunable("ww_ph_lambda", 4096)
    try:
        a = Automaton(N.MODE_WHOLEWORD, ["w%d" % i for i in range(9000)], True, word_chars=WORD)
    finally:
        N.set_tunable("ww_ph_lambda", 0)
    assert _perfect_tables(a)[0] == 0
    assert _perfect_tables(Automaton(N.MODE_WHOLEWORD, ["a" * 33, "b"], True, word_chars=WORD))[0] == 0
    assert _perfect_tables(Automaton(N.MODE_WHOLEWORD, ["a" * 32, "b"], True, word_chars=WORD))[0] > 0


def test_wordhash_fallback_seed_tables_reproduce_oracle():
    """Tables built from a later hash seed (test hook ww_first_seed; the builder goes there by itself when a dictionary
    cannot be placed) still find exactly the oracle's matches."""
    from ahocorasick_amd import synth
    from oracle.oracle import FAM_WHOLEWORD
    words = synth.mixed_script_words(1005, 1500)
    hay = synth.mixed_script_haystack(2052, 20000, words, swapcase_tbl=synth.swapcase_table())
    want = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay).tolist()
    try:
        N.set_tunable("ww_first_seed", 5)
        a = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD)
    finally:
        N.set_tunable("ww_first_seed", 0)
    assert _wordhash_tables(a)[4] != 0x811C9DC5
    assert len(want) > 500 and _simulate_wholeword(a, hay, WORD, False) == want


def test_stream_argument_checks_without_a_device():
    a = Automaton(N.MODE_ALL, ["ab"], True)
    L = N.lib()
    h = ctypes.c_void_p()
    assert L.acgpu_stream_open(None, ctypes.byref(h)) == N.E_INVALID
    assert L.acgpu_stream_open(a.handle, ctypes.byref(h)) == N.OK and h
    n_out, base = ctypes.c_uint64(0), ctypes.c_int64(0)
    buf = np.zeros(4, np.uint16)
    out = np.zeros((4, 3), np.int32)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    # record kind, missing pointers
    assert L.acgpu_stream_feed(h, vp(buf), 4, 0, 7, vp(out), 4, ctypes.byref(n_out), ctypes.byref(base)) == N.E_INVALID
    assert L.acgpu_stream_feed(h, None, 4, 0, N.REC_MAP, vp(out), 4, ctypes.byref(n_out), ctypes.byref(base)) == N.E_INVALID
    # an empty, non-final feed needs no device
    assert L.acgpu_stream_feed(h, None, 0, 0, N.REC_MAP, vp(out), 4, ctypes.byref(n_out), ctypes.byref(base)) == N.OK
    assert n_out.value == 0
    # a real feed fails loudly without a GPU (no CPU fallback)
    import torch
    if not torch.cuda.is_available():
        rc = L.acgpu_stream_feed(h, vp(buf), 4, 1, N.REC_MAP, vp(out), 4, ctypes.byref(n_out), ctypes.byref(base))
        assert rc in (N.E_NODEVICE, N.E_HIP)
    L.acgpu_stream_close(h)


def _states_tables(a):
    sizes = (ctypes.c_uint64 * 6)()
    N.check(N.lib().acgpu_debug_states(a.handle, sizes, None, None, None, None, None), "debug_states")
    n, n_dense, n_cls, w_rows, w_nodes, w_ids = (int(x) for x in sizes)
    if n == 0:
        return None
    rows, nodes, mask, out, ids = (np.zeros(max(k, 1), np.uint32) for k in (w_rows, w_nodes, n, 2 * n, w_ids))
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    N.check(N.lib().acgpu_debug_states(a.handle, sizes, vp(rows), vp(nodes), vp(mask), vp(out), vp(ids)), "debug_states")
    return n, n_dense, n_cls, rows.reshape(n_dense, n_cls), nodes[:w_nodes].reshape(-1, 4), mask, out.reshape(n, 2), ids


def _simulate_states(a, hay):
    """Test-only restatement of what k_ac_states / k_ac_states_out do with the compact automaton (csrc/acgpu_states.hip): one step per
    unit, a miss in a node goes to the fail state and looks at the unit again; records from the mask's bits, longest first; the
    counts that ride in the transitions must be the masks' popcounts."""
    n, n_dense, n_cls, rows, nodes, mask, out, ids = _states_tables(a)
    cls = _tables(a)[1]
    s, recs = 0, []
    for i, u in enumerate(hay.tolist()):
        c = int(cls[u])
        while True:
            if s < n_dense:
                e = int(rows[s, c])
                ns, flag, rep = e & 0x7FFFFF, (e >> 23) & 1, e >> 24
                break
            nd = [int(x) for x in nodes[s - n_dense]]
            if c == 0:
                ns, flag, rep = 0, 0, 0
                break
            hit = [k for k in (1, 2, 3) if nd[k] >> 24 == c]
            if hit:
                k = hit[0]
                ns, flag, rep = nd[k] & 0x7FFFFF, (nd[k] >> 23) & 1, (nd[0] >> (23 + 3 * (k - 1))) & 7
                break
            s = nd[0] & 0x7FFFFF
        s = ns
        m = int(mask[s])
        assert (m != 0) == bool(flag) and (rep == bin(m).count("1") or (rep == 7 and bin(m).count("1") >= 7)), (i, s, m, rep)
        assert int(out[s, 0]) == m
        at, k = int(out[s, 1]), 0
        for L in range(32, 0, -1):
            if m >> (L - 1) & 1:
                kid = (at & 0x7FFFFFFF) if at >> 31 else int(ids[at + k])
                recs.append([i + 1 - L, i + 1, kid])
                k += 1
    return recs


def test_compact_automaton_of_the_state_form_reproduces_the_oracle(fixtures):
    """acgpu_build.cpp 6d on the CPU: fixtures, a fuzz over small alphabets (fail hops, nested keywords, duplicates), both case modes."""
    done = 0
    for fx in fixtures:
        if "keywords_gen" in fx:
            continue
        hay, kws = fixture_inputs(fx)
        a = Automaton(N.MODE_ALL, kws, True)
        if _states_tables(a) is None:
            continue  # (a keyword of more than 32 units)
        assert _simulate_states(a, hay) == fx["AC"], fx["name"]
        done += 1
    assert done >= 3
    rng = np.random.default_rng(11)
    alpha = [ord(c) for c in "abcAB"] + [0x00E9, 0x00C9]
    for _ in range(40):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 25)), 7, int(rng.integers(0, 400)))
        for cs in (True, False):
            a = Automaton(N.MODE_ALL, kws, cs)
            want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            assert _simulate_states(a, hay) == want
    long_kw = [np.full(33, ord("a"), np.uint16)]
    assert _states_tables(Automaton(N.MODE_ALL, long_kw, True)) is None  # no mask bit for 33 units: no compact automaton


def test_source_hash_ignores_comments_and_nothing_else():
    """profiles/latest_traffic.json is keyed by a hash of the kernel sources without their comments (ahocorasick_amd/_native.py):
    a comment edit must not ask for a new counter collection, a code edit must -- also around digit separators and raw strings."""
    f = N._code_only
    assert f("a /* x */ b // y\n c") == "a b c"
    assert f("int n = 1'000'000; // c'd\nchar q = '\\''; x") == "int n = 1'000'000; char q = '\\''; x"
    assert f('const char *s = R"ab(// not a comment )" /* nor this */)ab"; /* gone */ y') == 'const char *s = R"ab(// not a comment )" /* nor this */)ab"; y'
    assert f('s = "a // b"; // c') == 's = "a // b";'
    assert f("x = 1; // one") == f("x = 1; /* uno */") != f("x = 2;")
