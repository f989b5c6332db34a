"""GPU parity tests: the HIP path, called through the C ABI (include/acgpu.h), against the CPU oracle
(oracle/ac_oracle.c) -- bit-exact records in the reference's listener-call order."""
import ctypes

import numpy as np
import pytest

from ahocorasick_amd import (AhoCorasickMap, AhoCorasickSet, IllegalArgumentException, LongestMatchMap, LongestMatchSet,
                             WholeWordMatchMap, WholeWordMatchSet)
from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton, utf16
from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, Oracle
from tests.helpers import oracle_parallel, LOWER, WORD, fixture_inputs, rand_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_tunables():
    yield
    for k, v in [("chunk_units", 0), ("blocks_per_cu", 1), ("lds_table_bytes", 127 * 1024), ("force_sparse", 0),
                 ("force_kernel", 0), ("region_units", 0), ("ww_first_seed", 0), ("tile_debug", 0), ("all_form", 0), ("tile_form", 0)]:
        N.set_tunable(k, v)


@pytest.fixture(params=["dfa_chunk_scan", "kgram_tile_scan", "kgram_split_scan"])
def ac_kernel(request):
    """Runs a test once per ALL-mode kernel: the DFA chunk scan, the fused K-gram tile kernel and its split form (filter
    kernel + verification kernel).  The tile kernels only exist for dictionaries with a K-gram filter; the others take
    the DFA scan either way, which the first parameter already covers."""
    N.set_tunable("force_kernel", {"dfa_chunk_scan": 1, "kgram_tile_scan": 2, "kgram_split_scan": 3}[request.param])
    return request.param


@pytest.fixture(params=["one_launch", "general_path"])
def host_path(request):
    """Tests on short haystacks run twice: through the one-launch form acgpu_match_u16 takes for up to 4096 units
    (csrc/acgpu_small.hip) and through the general path (copy in, scan kernels of the family, copy out) that every longer
    haystack takes -- the short inputs are where the edge cases live, both paths must see them."""
    if request.param == "general_path":
        N.set_tunable("tile_debug", 1 << 41)
    return request.param


def _ids(n):
    return list(range(n))


# ---- the reference's own deterministic scenarios -----------------------------------------------------------

def test_fixtures_ahocorasick(fixtures, ac_kernel):
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        m = AhoCorasickMap(kws, _ids(len(kws)), True)
        assert m.find_all(hay).tolist() == fx["AC"], fx["name"]
        s = AhoCorasickSet(kws, True)
        assert s.find_all(hay).tolist() == [r[:2] for r in fx["AC"]], fx["name"]


def test_listener_contract_and_early_stop(host_path):
    # SetTest-style counting listener + membership assertion (T/SetTest.java:156-165), and early stop (R/README.md:70)
    kws = ["a", "aa", "aaa", "aaaa"]
    hay = "aaaa"
    s = AhoCorasickSet(kws, True)
    seen = []

    def listener(h, start, end):
        assert h[start:end] in kws
        seen.append((start, end))
        return True

    s.match(hay, listener)
    assert len(seen) == 10
    full = Oracle(FAM_AC, kws).match(hay)
    for k in range(1, 11):
        got = []
        s.match(hay, lambda h, a, b: (got.append((a, b)) or len(got) < k))
        assert got == [tuple(r[:2]) for r in Oracle(FAM_AC, kws).match(hay, stop_after=k).tolist()]
    assert [tuple(r[:2]) for r in full.tolist()] == seen
    # Map: values are delivered, last duplicate wins (S/AhoCorasickMap.java:49-50)
    m = AhoCorasickMap(["ab", "x", "ab"], ["first", "x", "last"], True)
    vals = []
    m.match("zabz", lambda h, a, b, v: vals.append((a, b, v)) or True)
    assert vals == [(1, 3, "last")]
    with pytest.raises(TypeError):
        s.match(None, listener)


# ---- seeded fuzz against the oracle --------------------------------------------------------------------------

@pytest.mark.parametrize("seed", range(6))
def test_fuzz_ahocorasick_small_alphabets(seed, ac_kernel):
    rng = np.random.default_rng(seed)
    alpha = [ord(c) for c in "ab"] if seed % 2 == 0 else [ord(c) for c in "abcAB"] + [0x00E9, 0x00C9, 0x0130]
    for it in range(12):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 20)), int(rng.integers(1, 9)), int(rng.integers(0, 5000)))
        for cs in (True, False):
            want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            got = AhoCorasickMap(kws, _ids(len(kws)), cs).find_all(hay).tolist()
            assert got == want, (seed, it, cs)


@pytest.mark.parametrize("chunk_units,lds_bytes,sparse", [(8, 96 * 1024, 0), (64, 0, 0), (256, 1024, 0), (64, 0, 1), (0, 0, 1)])
def test_chunking_lds_and_sparse_variants_agree(chunk_units, lds_bytes, sparse):
    # every lane starts (max_len-1) units before its chunk: tiny chunks stress the halo logic; lds_bytes=0 forces
    # every row through HBM/L2; force_sparse exercises the hashed goto + fail-link path
    N.set_tunable("chunk_units", chunk_units)
    N.set_tunable("lds_table_bytes", lds_bytes)
    N.set_tunable("force_sparse", sparse)
    rng = np.random.default_rng(42)
    for alpha in ([ord(c) for c in "ab"], list(range(ord("a"), ord("z") + 1)), list(range(0x4E00, 0x4E40))):
        hay, kws = rand_case(rng, alpha, 40, 7, 20000)
        want = Oracle(FAM_AC, kws).match(hay).tolist()
        got = AhoCorasickMap(kws, _ids(len(kws)), True).find_all(hay).tolist()
        assert got == want


def test_dfa_chunk_scan_two_chunks_per_lane_buffer_tails_and_table_forms():
    # k_ac_dfa (round 4): every lane steps TWO chunks alternately, looks for outputs once per 8 units and takes the buffer's
    # last vector whole -- so: haystack lengths around multiples of 8 and 64 (the tail vector, shifted), chunks that end at the
    # buffer's end, fewer chunks than lanes, 16- and 32-bit tables (more than 65536 states), range classes and table classes
    # (case-insensitive), rows in LDS / partly / not at all; against the oracle and the one-chain kernel of rounds 1-3.
    N.set_tunable("force_kernel", 1)
    rng = np.random.default_rng(404)
    lower26 = list(range(ord("a"), ord("z") + 1))
    cases = []
    for alpha, n_kw, max_len in (([ord(c) for c in "ab"], 30, 9), (lower26, 300, 6), ([ord(c) for c in "abcAB"] + [0x00E9, 0x00C9, 0x0130], 40, 5)):
        for n_units in (64, 65, 71, 72, 127, 1000, 1001, 4097, 20005, 65543):
            hay, kws = rand_case(rng, alpha, n_kw, max_len, n_units)
            cases.append((hay, kws))
    big_kws = [rng.choice(lower26, size=int(rng.integers(4, 10))).astype(np.uint16) for _ in range(22000)]  # > 65536 states: 32-bit entries
    big_hay = rng.choice(lower26, size=300001).astype(np.uint16)
    for k in range(0, 300001 - 10, 977):
        kw = big_kws[k % len(big_kws)]
        big_hay[k:k + len(kw)] = kw
    cases.append((big_hay, big_kws))
    seen = set()
    for hay, kws in cases:
        for cs in (True, False):
            want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            m = AhoCorasickMap(kws, _ids(len(kws)), cs)
            for knobs in ({}, {"lds_table_bytes": 1024}, {"lds_table_bytes": 0, "chunk_units": 64}, {"chunk_units": 8}, {"tile_debug": 1 << 43}):
                if len(hay) > 100000 and knobs.get("chunk_units") == 8:
                    continue
                for k, v in {"lds_table_bytes": 127 * 1024, "chunk_units": 0, "tile_debug": 0, **knobs}.items():
                    N.set_tunable(k, v)
                assert m.find_all(hay).tolist() == want, (len(hay), len(kws), cs, knobs)
            import torch
            N.set_tunable("tile_debug", 0)
            d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
            _, prof = _dev_match(m._auto, d_hay, len(hay), True, len(want) + 16, profile=True)
            seen.add(prof["scan_kernel"])
    assert any(k.startswith("k_ac_dfa<unsigned int") for k in seen) and any(k.startswith("k_ac_dfa<unsigned short, true") for k in seen) \
        and any(k.startswith("k_ac_dfa<unsigned short, false") for k in seen), seen


@pytest.mark.parametrize("region_units,min_len", [(512, 1), (512, 2), (1024, 3), (512, 4), (4096, 6), (0, 8)])
def test_tile_kernel_regions_and_filter_lengths(region_units, min_len):
    # the K-gram filter length: at most the shortest keyword when that has 4 units or more (or with the builder knob
    # no_short_keywords), else up to 4 with the shorter keywords beside the filter; small regions stress the region seams
    N.set_tunable("force_kernel", 2)
    N.set_tunable("region_units", region_units)
    rng = np.random.default_rng(1000 + min_len)
    for alpha in ([ord(c) for c in "ab"], list(range(ord("a"), ord("h") + 1)), [ord(c) for c in "abAB"] + [0x00E9, 0x00C9]):
        hay, kws = rand_case(rng, alpha, 30, min_len + 5, 30011, min_len=min_len)
        for cs in (True, False):
            want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            for no_short in (0, 1):
                N.set_tunable("no_short_keywords", no_short)
                try:
                    m = AhoCorasickMap(kws, _ids(len(kws)), cs)
                finally:
                    N.set_tunable("no_short_keywords", 0)
                info = m.automaton.info()
                k_max = info["min_keyword_len"] if (no_short or info["min_keyword_len"] >= 4) else min(4, info["max_keyword_len"])
                assert 1 <= info["filter_k"] <= min(k_max, 8) and info["tile_kernel"] == 1
                assert m.find_all(hay).tolist() == want


def test_kernel_selection_defaults():
    c2 = Automaton(N.MODE_ALL, synth.config_keywords("C2"), True).info()
    assert c2["filter_k"] == 4 and c2["tile_kernel"] == 1 and c2["filter_density"] < 0.05
    dense_dict = Automaton(N.MODE_ALL, ["a", "b", "ab"], True).info()
    assert dense_dict["tile_kernel"] == 1  # even a filter that passes every position beats the DFA chunk scan (measured)
    assert Automaton(N.MODE_LONGEST, ["a", "b", "ab"], True).info()["tile_kernel"] == 0  # Longest: dense -> the walk
    wide = Automaton(N.MODE_ALL, [chr(0x4E00 + i) + chr(0x4E01 + i) for i in range(200)], True).info()
    assert wide["filter_k"] == 2 and wide["tile_kernel"] == 1  # more than 63 distinct units: bucketed classes, exact K-gram lookup


def test_full_alphabet_dictionary_sparse():
    # T/SetTest.java:72-79 (65536 single-unit keywords) + an extended haystack
    kws = [np.array([i], dtype=np.uint16) for i in range(65536)]
    m = AhoCorasickMap(kws, _ids(65536), True)
    assert m.automaton.info()["dense"] == 0
    hay = np.array([0, 0xFFFF, 0xFFFE, 0xD800, 0x41], dtype=np.uint16)
    assert m.find_all(hay).tolist() == [[i, i + 1, int(u)] for i, u in enumerate(hay)]


def test_random_wide_alphabet_like_reference_full_random(ac_kernel):
    # T/SetTest.java:81-89 uses unseeded 2-3 unit keywords over the whole BMP; seeded here
    rng = np.random.default_rng(7)
    kws = []
    for _ in range(20000):
        ln = int(rng.integers(2, 4))
        u = np.where(rng.random(ln) < 0.5, rng.integers(0, 256, ln), rng.integers(0, 65536, ln)).astype(np.uint16)
        kws.append(u)
    hay = utf16("The quick red fox, jumps over the lazy brown dog.")
    hay = np.concatenate([hay, kws[5], hay, kws[77], kws[5][:1]])
    want = Oracle(FAM_AC, kws).match(hay).tolist()
    assert AhoCorasickMap(kws, _ids(len(kws)), True).find_all(hay).tolist() == want


def test_overflow_protocol(ac_kernel):
    a = Automaton(N.MODE_ALL, ["a", "aa"], True)
    hay = utf16("a" * 1000)
    out = np.empty((10, 3), np.int32)
    n_out = ctypes.c_uint64(0)
    rc = N.lib().acgpu_match_u16(a.handle, hay.ctypes.data_as(ctypes.c_void_p), hay.size, N.REC_MAP,
                                 out.ctypes.data_as(ctypes.c_void_p), 10, ctypes.byref(n_out))
    assert rc == N.E_OVERFLOW and n_out.value == 1999
    assert len(a.match_host(hay, True, cap=16)) == 1999


# ---- BASELINE.json config 1 (plumbing size) end to end -------------------------------------------------------

def test_config_c1_bit_exact(ac_kernel):
    c = synth.CONFIGS["C1"]
    kws = synth.config_keywords("C1")
    hay = synth.haystack(c["hay_seed"], c["n_units"])
    want = Oracle(FAM_AC, kws).match(hay)
    got = AhoCorasickSet(kws, True).find_all(hay)
    assert got.shape == want[:, :2].shape and (got == want[:, :2]).all()
    gotm = AhoCorasickMap(kws, _ids(len(kws)), True).find_all(hay)
    assert (gotm == want).all()


# ---- device-resident entry point, shards and halos -------------------------------------------------------------

def _dev_match(a, d_hay, n, with_ids, cap, **kw):
    import torch
    cols = 3 if with_ids else 2
    d_out = torch.empty((max(cap, 1), cols), dtype=torch.int32, device="cuda")
    n_out, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, with_ids, d_out.data_ptr(), cap,
                                        stream=torch.cuda.current_stream().cuda_stream, **kw)
    assert rc == N.OK, rc
    return d_out[:n_out].cpu().numpy(), prof


def test_synth_fill_matches_numpy_generator():
    import torch
    n = 100003
    for table, seed, start in ((synth.ALPHA_LOWER, 2002, 0), (synth.ALPHA_AB_75, 2004, 12345)):
        d = torch.empty(n, dtype=torch.int16, device="cuda")
        tab = np.ascontiguousarray(table, dtype=np.uint16)
        N.check(N.lib().acgpu_synth_fill(d.data_ptr(), n, start, seed, tab.ctypes.data_as(ctypes.c_void_p), len(tab), None),
                "synth")
        torch.cuda.synchronize()
        got = d.cpu().numpy().view(np.uint16)
        assert (got == synth.haystack(seed, n, table, start=start)).all()


def test_device_entry_and_shard_split_invariance(ac_kernel):
    import torch
    kws = synth.random_keywords(11, 300, 2, 9)
    hay = synth.haystack(77, 300000)
    a = Automaton(N.MODE_ALL, kws, True)
    want = Oracle(FAM_AC, kws).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    got, prof = _dev_match(a, d_hay, hay.size, True, len(want) + 10, profile=True)
    assert (got == want).all() and prof["n_matches"] == len(want) and prof["scan_ms"] > 0
    # split into 3 shards owning [0,a) [a,b) [b,n): each sees the whole buffer but owns a slice; concatenation == whole
    cuts = [0, 99991, 200003, hay.size]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p, _ = _dev_match(a, d_hay, hay.size, True, len(want) + 10, own=(lo, hi))
        parts.append(p)
    assert (np.concatenate(parts) == want).all()
    # a shard handed only its own slice + left halo (what a multi-GPU rank holds); positions are buffer-relative
    halo = a.info()["max_keyword_len"] - 1
    lo, hi = cuts[1], cuts[2]
    base = (lo - halo) // 8 * 8  # keep the device pointer 16-byte aligned
    sub = d_hay[base:hi].clone()
    p, _ = _dev_match(a, sub, hi - base, True, len(want) + 10, own=(lo - base, hi - base), text_begin=False, text_end=False)
    p[:, :2] += base
    assert (p == parts[1]).all()


# ---- BASELINE.json config 2 at FULL size: size-independent properties -------------------------------------------

def test_config_c2_full_size_every_record():
    import torch
    c = synth.CONFIGS["C2"]
    kws = synth.config_keywords("C2")
    n = c["n_units"]
    a = Automaton(N.MODE_ALL, kws, True)
    d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
    tab = np.ascontiguousarray(synth.ALPHA_LOWER)
    N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, c["hay_seed"], tab.ctypes.data_as(ctypes.c_void_p), len(tab),
                                     None), "synth")
    cap = 4_000_000
    got, prof = _dev_match(a, d_hay, n, True, cap, profile=True)
    m = len(got)
    assert 1_200_000 < m < 1_550_000  # ~2.5e-3 matches per unit for this dictionary
    # (1) reference order: end ascending, ties by start ascending
    end, start = got[:, 1].astype(np.int64), got[:, 0].astype(np.int64)
    key = end * (1 << 32) + start
    assert (np.diff(key) > 0).all()
    # (2) every record is a true occurrence of the keyword it names (sampled: 20000 records + first/last 1000)
    idx = np.unique(np.concatenate([np.arange(1000), np.arange(m - 1000, m),
                                    np.random.default_rng(0).integers(0, m, 20000)]))
    for i in idx.tolist():
        s, e, k = got[i].tolist()
        seg = synth.haystack(c["hay_seed"], e - s, start=s)
        assert (seg == kws[k]).all()
    # (3) EVERY record equals the oracle's on the whole 2^29-unit text, bit for bit and in order (the single-threaded
    #     oracle run on chunks in host threads and stitched: tests/helpers.py oracle_parallel; T/SetTest.java:145-192
    #     compares the count on the whole input)
    hay = d_hay.cpu().numpy().view(np.uint16)
    assert (hay[:1 << 16] == synth.haystack(c["hay_seed"], 1 << 16)).all()  # (the device generator is the numpy one)
    want = oracle_parallel(Oracle(FAM_AC, kws), hay, "ac", a.info()["max_keyword_len"], cap_per_unit=0.02)
    assert got.shape == want.shape and (got == want).all()
    del hay
    # (4) the independent kernels (fused K-gram tile scan / its split form / DFA chunk scan), other region and chunk
    #     sizes, and no LDS residency all give the identical record stream
    assert prof["scan_kernel"].startswith("k_ac_tile")
    for knobs in ({"force_kernel": 1}, {"force_kernel": 1, "chunk_units": 1000, "lds_table_bytes": 0},
                  {"force_kernel": 2, "region_units": 512 * 7}, {"force_kernel": 3}, {"force_kernel": 3, "region_units": 4096}):
        for k, v in knobs.items():
            N.set_tunable(k, v)
        got2, prof2 = _dev_match(a, d_hay, n, True, cap, profile=True)
        assert got2.shape == got.shape and (got2 == got).all(), (knobs, prof2["scan_kernel"])


# ---- LongestMatchSet / LongestMatchMap ------------------------------------------------------------------------------

def test_fixtures_longest(fixtures, host_path):
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        assert LongestMatchMap(kws, _ids(len(kws)), True).find_all(hay).tolist() == fx["L"], fx["name"]
        assert LongestMatchSet(kws, True).find_all(hay).tolist() == [r[:2] for r in fx["L"]], fx["name"]


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_longest_vs_oracle(seed, host_path):
    rng = np.random.default_rng(500 + seed)
    alpha = [ord(c) for c in "ab"] if seed % 2 == 0 else [ord(c) for c in "abcAB"] + [0x00E9, 0x00C9]
    for it in range(12):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 20)), int(rng.integers(1, 9)), int(rng.integers(0, 6000)))
        for cs in (True, False):
            want = Oracle(FAM_LONGEST, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            got = LongestMatchMap(kws, _ids(len(kws)), cs).find_all(hay).tolist()
            assert got == want, (seed, it, cs)


@pytest.mark.parametrize("chunk_units,lds_bytes,sparse", [(8, 96 * 1024, 0), (64, 0, 0), (0, 96 * 1024, 1)])
def test_longest_variants(chunk_units, lds_bytes, sparse):
    N.set_tunable("chunk_units", chunk_units)
    N.set_tunable("lds_table_bytes", lds_bytes)
    N.set_tunable("force_sparse", sparse)
    rng = np.random.default_rng(4242)
    for alpha in ([ord(c) for c in "ab"], list(range(ord("a"), ord("g") + 1)), list(range(0x4E00, 0x4E20))):
        hay, kws = rand_case(rng, alpha, 60, 9, 50000)
        want = Oracle(FAM_LONGEST, kws).match(hay).tolist()
        assert LongestMatchMap(kws, _ids(len(kws)), True).find_all(hay).tolist() == want


def test_longest_listener_and_early_stop(host_path):
    kws = ["a", "aa", "aaa", "aaaa"]
    hay = " aaaaaaa aaababababaabaa "
    s = LongestMatchSet(kws, True)
    full = Oracle(FAM_LONGEST, kws).match(hay)
    for k in (1, 3, len(full)):
        got = []
        s.match(hay, lambda h, a, b: (got.append((a, b)) or len(got) < k))
        assert got == [tuple(r[:2]) for r in Oracle(FAM_LONGEST, kws).match(hay, stop_after=k).tolist()]
    m = LongestMatchMap(["ab", "abc", "ab"], ["x", "y", "z"], True)
    vals = []
    m.match("ab abc", lambda h, a, b, v: vals.append((a, b, v)) or True)
    assert vals == [(0, 2, "z"), (3, 6, "y")]


def test_longest_prefix_closed_dictionary_like_config_c4():
    # BASELINE config 4 in miniature: every prefix of long {a,b} words (a, aa, aaa, ...), P(a)=0.75 haystack;
    # max keyword length 200 stresses the right-halo warm-up and the synchronisation-point logic
    kws = synth.prefix_closed_keywords(1004, 3000, word_len=200)
    hay = synth.haystack(2004, 1 << 18, table=synth.ALPHA_AB_75)
    want = Oracle(FAM_LONGEST, kws).match(hay)
    got = LongestMatchSet(kws, True).find_all(hay)
    assert got.shape == want[:, :2].shape and (got == want[:, :2]).all()
    # adversarial: one long run of 'a' (no synchronisation point inside the run)
    hay2 = np.concatenate([np.full(5000, ord("a"), np.uint16), utf16("bab"), np.full(777, ord("a"), np.uint16)])
    want2 = Oracle(FAM_LONGEST, kws).match(hay2)
    got2 = LongestMatchMap(kws, _ids(len(kws)), True).find_all(hay2)
    assert (got2 == want2).all()


@pytest.mark.parametrize("seed", [1, 2])
def test_longest_block_maxima_of_walks_that_finish_before_their_chunk_does(seed):
    """k_longest_block over text that keeps its work list full (long runs of 'a' under a^1..a^100: every walk is still alive
    behind the root table), so that walks run to their end while their 1024-position chunk is still being streamed -- the
    chunk's block maxima must already be preset then, or the maximum a long match has raised is overwritten and
    k_longest_sync skips the block.  Planted copies of a 100-unit word W whose inner positions only match single units
    make such a miss visible: the real chain jumps over W, a synchronisation point inside it reports its units one by one."""
    rng = np.random.default_rng(4200 + seed)
    a, b = ord("a"), ord("b")
    W = np.array([b, a, b, b] + [a if x else b for x in rng.integers(0, 2, 96).tolist()], dtype=np.uint16)
    kws = [np.full(k, a, np.uint16) for k in range(1, 101)] + [utf16("ab"), utf16("b")] + [W[:k] for k in (2, 3, 4, 100)]
    parts, n = [], 0
    while n < 600000:
        r = int(rng.integers(0, 10))
        if r < 3:
            parts.append(W)
        elif r < 8:  # runs of 'a' that outlive the root table's 14 units
            parts.append(np.concatenate([np.full(int(rng.integers(15, 130)), a, np.uint16), np.array([b], np.uint16)]))
        else:
            parts.append(np.where(rng.integers(0, 4, int(rng.integers(1, 40))) > 0, a, b).astype(np.uint16))
        n += len(parts[-1])
    hay = np.concatenate(parts)
    want = Oracle(FAM_LONGEST, kws).match(hay)
    m = LongestMatchSet(kws, True)
    for region in (0, 1024, 4096):  # chain tiles: the default and two fixed sizes (synchronisation points fall elsewhere)
        N.set_tunable("region_units", region)
        got = m.find_all(hay)
        assert got.shape == want[:, :2].shape and (got == want[:, :2]).all(), region
    assert int((want[:, 1] - want[:, 0]).max()) == 100


@pytest.mark.parametrize("letters,others", [("ab", ""), ("ab", " ,"), ("a", "b"), ("acgt", ""), ("acgt", "n"), ("abc", "xyz")])
def test_longest_walk_root_table_bit_fields_and_units_outside_the_alphabet(letters, others):
    """k_longest_walk_list's first round (Set records): the bit-field root table -- one bit per unit for alphabets of up to two
    letters (13 units deep), two bits for up to four (6 deep) -- and its fall-back to the general first round for a
    512-position step that holds a unit outside the alphabet; keywords longer than the table is deep go on through the work
    list, some beyond 255 units."""
    rng = np.random.default_rng(len(letters) * 100 + len(others))
    la = np.array([ord(c) for c in letters], dtype=np.uint16)
    n = 300000
    hay = la[rng.integers(0, len(la), n)]
    if others:  # a few stretches with other units, the rest clean (both paths run)
        oa = np.array([ord(c) for c in others], dtype=np.uint16)
        for at in rng.integers(0, n - 2000, 12).tolist():
            k = int(rng.integers(1, 900))
            idx = at + rng.integers(0, 2000, k)
            hay[idx] = oa[rng.integers(0, len(oa), k)]
    base = [hay[o:o + ln].copy() for o, ln in zip(rng.integers(0, n - 700, 40).tolist(), rng.integers(1, 600, 40).tolist())]
    kws = [b[:k] for b in base for k in sorted(set(rng.integers(1, len(b) + 1, 6).tolist()))]  # prefixes: long shared paths
    kws += [la[rng.integers(0, len(la), int(rng.integers(1, 16)))] for _ in range(200)]
    want = Oracle(FAM_LONGEST, kws).match(hay)
    m = LongestMatchSet(kws, True)
    got = m.find_all(hay)
    assert got.shape == want[:, :2].shape and (got == want[:, :2]).all()
    assert int((want[:, 1] - want[:, 0]).max()) > 255 or len(letters) == 1
    gm = LongestMatchMap(kws, _ids(len(kws)), True).find_all(hay)  # (the Map flavour walks without the root table)
    assert gm.shape == want.shape and (gm == want).all()


def test_longest_shards_with_chain_entry_and_exit():
    import torch
    kws = synth.random_keywords(21, 400, 1, 7, table=synth.ALPHA_LOWER[:5])
    hay = synth.haystack(78, 200000, table=synth.ALPHA_LOWER[:5])
    a = Automaton(N.MODE_LONGEST, kws, True)
    want = Oracle(FAM_LONGEST, kws).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    cap = len(want) + 10
    cuts = [0, 70001, 140003, hay.size]
    parts, entry = [], 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
        n_out, rc, _, chain_exit = a.match_device(d_hay.data_ptr(), hay.size, True, d_out.data_ptr(), cap, own=(lo, hi),
                                                  chain_entry=max(entry, lo))
        assert rc == N.OK
        parts.append(d_out[:n_out].cpu().numpy())
        entry = chain_exit
        assert entry >= hi
    assert (np.concatenate(parts) == want).all()


# ---- WholeWordMatchSet / WholeWordMatchMap ---------------------------------------------------------------------------

def test_fixtures_wholeword(fixtures, host_path):
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        if fx["WW"] == "IllegalArgumentException":
            # T/WholeWordMatchTest.java:31-57: keywords with inner non-word characters are rejected by the constructor
            with pytest.raises(IllegalArgumentException):
                WholeWordMatchSet(kws, True)
            with pytest.raises(IllegalArgumentException):
                WholeWordMatchMap(kws, _ids(len(kws)), True)
            continue
        assert WholeWordMatchMap(kws, _ids(len(kws)), True).find_all(hay).tolist() == fx["WW"], fx["name"]
        assert WholeWordMatchSet(kws, True).find_all(hay).tolist() == [r[:2] for r in fx["WW"]], fx["name"]


def test_wholeword_boundaries_like_reference_assertions(host_path):
    # T/WholeWordMatchTest.java:60-70: every reported needle is delimited by non-word characters or the string ends
    kws = ["The", "quick", "red", "fox", "jumps", "over", "the", "lazy", "brown", "dog", "re", "ox"]
    hay = "The quick red fox, jumps over the lazy brown dog. redfox ox-re re_d dog"
    s = WholeWordMatchSet(kws, True)
    wc = s.get_word_chars()
    seen = []

    def listener(h, a, b):
        assert h[a:b] in kws
        assert b == len(h) or not wc[ord(h[b])]
        assert a == 0 or not wc[ord(h[a - 1])]
        seen.append((a, b))
        return True

    s.match(hay, listener)
    assert seen == [tuple(r[:2]) for r in Oracle(FAM_WHOLEWORD, kws, word_chars=WORD).match(hay).tolist()] and len(seen) == 11


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_wholeword_vs_oracle(seed, host_path):
    rng = np.random.default_rng(900 + seed)
    alpha = [ord(c) for c in "abB -_.9"] + [0x00E9, 0x00C9, 0x4E2D, 0x3002, 0x0130]
    word_alpha = [c for c in alpha if WORD[c]]
    for it in range(10):
        hay, _ = rand_case(rng, alpha, 1, 1, int(rng.integers(0, 8000)))
        _, kws = rand_case(rng, word_alpha, int(rng.integers(1, 25)), 5, 1)
        kws = kws + [np.concatenate([utf16(" ,"), kws[0], utf16(". ")])]  # trimmed by the constructor
        for cs in (True, False):
            want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay).tolist()
            got = WholeWordMatchMap(kws, _ids(len(kws)), cs).find_all(hay).tolist()
            assert got == want, (seed, it, cs)


def test_wholeword_custom_tables_including_fold_inconsistent():
    rng = np.random.default_rng(77)
    alpha = [ord(c) for c in "aAbB xX."]
    hay, _ = rand_case(rng, alpha, 1, 1, 3000)
    # (1) explicit word-character list (S/WordCharacters.java:18-24): consistent under folding
    for wchars, cs in (("aAbB", True), ("aAbB", False), ("aAbBxX", False)):
        wc = np.zeros(65536, np.uint8)
        for ch in wchars:
            wc[ord(ch)] = 1
        kws = ["a", "ab", "B", "ba", "abba"]
        want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=wc).match(hay).tolist()
        m = WholeWordMatchMap(kws, _ids(len(kws)), cs, word_characters=list(wchars))
        assert m.automaton.info()["fold_consistent"] == 1
        assert m.find_all(hay).tolist() == want
    # (2) a table where 'A','B' are word characters but 'a','b' are not, case-insensitive: the reference mixes folded
    #     and raw lookups (S/WholeWordMatchMap.java:204,209 vs :221,:226); the sequential kernel restates it literally
    wc = np.zeros(65536, np.uint8)
    for ch in "ABx":
        wc[ord(ch)] = 1
    kws = ["A", "AB", "BA", "x"]
    want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=wc).match(hay).tolist()
    m = WholeWordMatchMap(kws, _ids(len(kws)), False, word_characters=list("ABx"))
    assert m.automaton.info()["fold_consistent"] == 0
    assert m.find_all(hay).tolist() == want
    # (3) default table with toggles (S/WordCharacters.java:26-39)
    from ahocorasick_amd.unicode_tables import word_chars_with_toggles
    wc = word_chars_with_toggles([".", "b"], [True, False])  # '.' becomes a word character, 'b' stops being one
    m = WholeWordMatchMap(["a.b", "ab"], [0, 1], True, word_characters=[".", "b"], toggle_flags=[True, False])
    want = Oracle(FAM_WHOLEWORD, ["a.b", "ab"], word_chars=wc).match("a.b ab a.b. b").tolist()
    assert want == [[0, 2, 0], [4, 5, 1], [7, 9, 0]]  # keywords are trimmed to "a." and "a" (S/WordCharacters.java:41-62)
    assert m.find_all("a.b ab a.b. b").tolist() == want
    with pytest.raises(IllegalArgumentException):
        WholeWordMatchSet(["a c"], True, word_characters=[".", "b"], toggle_flags=[True, False])


def test_wholeword_mixed_script_like_config_c5():
    # BASELINE config 5 in miniature: mixed-script dictionary, case-insensitive, default word characters
    words = synth.mixed_script_words(1005, 5000)
    hay = synth.mixed_script_haystack(2005, 300000, words, swapcase_tbl=synth.swapcase_table())
    want = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay)
    got = WholeWordMatchMap(words, _ids(len(words)), False).find_all(hay)
    assert len(want) > 10000 and got.shape == want.shape and (got == want).all()
    for region in (2048, 0):
        N.set_tunable("region_units", region)
        assert (WholeWordMatchSet(words, False).find_all(hay) == want[:, :2]).all()
    # a device that cannot spare the region-local record slots (6 bytes per unit): the scratch slices + k_permute, not an error
    for bits in (1 << 40, 134217728):  # the allocation "fails" / the slices on request
        N.set_tunable("tile_debug", bits)
        got2 = WholeWordMatchMap(words, _ids(len(words)), False).find_all(hay)
        assert got2.shape == want.shape and (got2 == want).all(), bits


@pytest.mark.parametrize("seed", range(8))
def test_wholeword_long_words_and_hash_table_paths(seed):
    """Words longer than the 16 units the kernel keeps in registers, many words sharing long prefixes, buffer ends
    inside a word, case folding on every unit.  Seeds 4 and 5: no keyword beyond 16 units -- the position-parallel kernel
    (k_ww_pp), whose keywords of 13..16 units compare their record, with runs of up to 17+ units in the text; seeds 6 and 7: up
    to 32 units, its LONG form (two ring reads per run, the tail compared with the record word by word)."""
    rng = np.random.default_rng(100 + seed)
    alpha = np.array([ord(c) for c in "abAB"] + [0x00E9, 0x00C9, 0x0391, 0x03B1], dtype=np.uint16)
    kws = []
    lens = ([1, 2, 5, 8, 9, 12, 13, 16, 17, 24, 33, 100] if seed < 4 else [1, 2, 5, 8, 9, 11, 12, 13, 14, 15, 16] if seed < 6 else
            [1, 2, 8, 12, 13, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32])  # 6, 7: k_ww_pp's 32-unit form
    for _ in range(300):
        ln = int(rng.choice(lens))
        kws.append(alpha[rng.integers(0, len(alpha), ln)])
    parts = []
    for _ in range(3000):
        k = kws[int(rng.integers(0, len(kws)))].copy()
        mode = int(rng.integers(0, 4))
        if mode == 1 and len(k) > 1:
            k[int(rng.integers(0, len(k)))] = alpha[int(rng.integers(0, len(alpha)))]  # near miss
        elif mode == 2:
            k = np.concatenate([k, alpha[rng.integers(0, len(alpha), 1)]])  # one unit longer
        parts.append(k)
        parts.append(np.array([32] * int(rng.integers(1, 3)), dtype=np.uint16))
    hay = np.concatenate(parts[:-1])  # ends inside a word
    for cs in (True, False):
        want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay)
        m = WholeWordMatchMap(kws, _ids(len(kws)), cs)
        got = m.find_all(hay)
        assert got.shape == want.shape and (got == want).all()
        import torch
        _, prof = _dev_match(m.automaton, torch.from_numpy(hay.view(np.int16)).cuda(), hay.size, True, len(want) + 8, profile=True)
        kn = prof["scan_kernel"]
        # k_ww_pp<fold, LONG, PH>: the 32-unit form for seeds 6 and 7, the perfect hash (csrc/acgpu_build.cpp 5b) for all of them
        assert kn.startswith("k_ww_tile") if seed < 4 else (kn.startswith("k_ww_pp") and kn.split(",")[1].strip() == ("false" if seed < 6 else "true") and kn.endswith(", true>")), kn
        N.set_tunable("force_kernel", 1)  # the trie-walk verification agrees
        got2 = WholeWordMatchMap(kws, _ids(len(kws)), cs).find_all(hay)
        N.set_tunable("force_kernel", 0)
        assert (got2 == want).all()


def test_wholeword_shards_own_their_word_starts():
    import torch
    words = synth.mixed_script_words(31, 2000)
    hay = synth.mixed_script_haystack(32, 120000, words)
    a = Automaton(N.MODE_WHOLEWORD, words, True, word_chars=WORD)
    want = Oracle(FAM_WHOLEWORD, words, word_chars=WORD).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    cap = len(want) + 10
    cuts = [0, 40001, 80003, hay.size]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p, _ = _dev_match(a, d_hay, hay.size, True, cap, own=(lo, hi))
        parts.append(p)
    assert (np.concatenate(parts) == want).all()
    # a rank holding only its slice: 1 unit of left context, max_keyword_len+1 units of right halo
    lo, hi = cuts[1], cuts[2]
    halo_r = a.info()["max_keyword_len"] + 1
    base = (lo - 1) // 8 * 8
    sub = d_hay[base:hi + halo_r].clone()
    p, _ = _dev_match(a, sub, hi + halo_r - base, True, cap, own=(lo - base, hi - base), text_begin=False, text_end=False)
    p[:, :2] += base
    assert (p == parts[1]).all()


# ---- BASELINE.json configs 4 and 5 at FULL size: size-independent properties -----------------------------------------

def test_config_c4_full_size_every_record():
    """LongestMatchSet, 50k prefix-closed keywords (max length 1000), 2^29 units with P(a)=0.75."""
    import torch
    c = synth.CONFIGS["C4"]
    kws = synth.config_keywords("C4")
    n = c["n_units"]
    a = Automaton(N.MODE_LONGEST, kws, True)
    d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
    tab = np.ascontiguousarray(synth.ALPHA_AB_75)
    N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, c["hay_seed"], tab.ctypes.data_as(ctypes.c_void_p), len(tab),
                                     None), "synth")
    cap = n // 2
    d_out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
    n_out, rc, prof, chain_exit = a.match_device(d_hay.data_ptr(), n, False, d_out.data_ptr(), cap, profile=True,
                                                 stream=torch.cuda.current_stream().cuda_stream)
    assert rc == N.OK and chain_exit >= n
    got = d_out[:n_out]
    print("C4 full size: %d matches, scan %.3f ms, chain %.3f ms" % (n_out, prof["scan_ms"], prof["finalize_ms"]))
    # (1) position order, non-overlapping: start[i+1] >= end[i]
    assert bool((got[1:, 0] >= got[:-1, 1]).all()) and bool((got[:, 1] > got[:, 0]).all())
    # (2) greedy chain: every unit between two matches starts no keyword -- here every 'a' starts one, so gaps are 'b's
    #     that start no keyword; and every match is maximal among the a-run family: checked against the oracle prefix
    #     EVERY record against the oracle on the whole 2^29-unit text (leftmost-longest is a chain: one oracle run, one
    #     thread, like the reference)
    hay = d_hay.cpu().numpy().view(np.uint16)
    want = Oracle(FAM_LONGEST, kws).match(hay, cap=n // 4)[:, :2]
    g = got.cpu().numpy()
    assert g.shape == want.shape and (g == want).all()
    del hay, want, g
    # (3) shard invariance at full size: two shards chained through chain_exit give the same stream (checksums)
    half = n // 2 + 12345
    d2 = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
    n1, rc1, _, ex1 = a.match_device(d_hay.data_ptr(), n, False, d2.data_ptr(), cap, own=(0, half))
    assert rc1 == N.OK
    first = d2[:n1].clone()
    n2, rc2, _, ex2 = a.match_device(d_hay.data_ptr(), n, False, d2.data_ptr(), cap, own=(half, n), chain_entry=ex1)
    assert rc2 == N.OK and n1 + n2 == n_out
    assert bool((first == got[:n1]).all()) and bool((d2[:n2] == got[n1:]).all())


def test_token_stream_generator_matches_numpy_twin():
    """acgpu_synth_tokens (config 5's haystack, generated in place on the device) against synth.token_stream_haystack: sizes
    that end inside a word, inside the separators and on a token boundary; with and without case flips; long words."""
    import torch
    words = synth.mixed_script_words(1005, 3000)
    sw = synth.swapcase_table()
    for n, seed, tbl in ((1, 2005, sw), (2, 7, sw), (1000, 2005, sw), (100003, 2006, sw), (65536, 2007, None)):
        d = torch.empty(n + 8, dtype=torch.int16, device="cuda")
        d.fill_(-1)
        synth.token_stream_on_device(d.data_ptr(), n, seed, words, tbl)
        got = d.cpu().numpy().view(np.uint16)
        assert (got[:n] == synth.token_stream_haystack(seed, n, words, tbl, chunk_tokens=4096)).all(), n
        assert (got[n:] == 0xffff).all()  # nothing written behind the end
    long_words = [np.full(40, 0x61, np.uint16), np.arange(0x4E00, 0x4E00 + 33, dtype=np.uint16)] + words[:50]
    d = torch.empty(50000, dtype=torch.int16, device="cuda")
    synth.token_stream_on_device(d.data_ptr(), 50000, 11, long_words, sw)
    assert (d.cpu().numpy().view(np.uint16) == synth.token_stream_haystack(11, 50000, long_words, sw)).all()
    # words of ONE unit make 2-unit tokens: the device must size its token count from the shortest token, not from 3 units
    short_words = [np.array([0x61], np.uint16), np.array([0x4E00], np.uint16), np.array([0x62, 0x63], np.uint16)]
    d = torch.empty(30000, dtype=torch.int16, device="cuda")
    d.fill_(-1)
    synth.token_stream_on_device(d.data_ptr(), 30000, 12, short_words, sw)
    assert (d.cpu().numpy().view(np.uint16) == synth.token_stream_haystack(12, 30000, short_words, sw)).all()


def test_config_c5_full_size_every_record():
    """WholeWordMatchMap, 100k mixed-script words, case-insensitive, 2^28 units (one GPU's share of config 5): the token stream
    of SURVEY.md 8d, aperiodic, generated on the device (acgpu_synth_tokens, seed 2005)."""
    import torch
    c = synth.CONFIGS["C5"]
    words = synth.config_keywords("C5")
    n = c["n_units"]
    a = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD)
    d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
    synth.token_stream_on_device(d_hay.data_ptr(), n, c["hay_seed"], words, synth.swapcase_table())
    hay = d_hay.cpu().numpy().view(np.uint16)
    assert (hay[:1 << 20] == synth.token_stream_haystack(c["hay_seed"], 1 << 20, words, synth.swapcase_table())).all()
    orc = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD)
    # (1) EVERY record equals the oracle's on the whole 2^28-unit text (chunks in host threads, stitched by word start)
    want = oracle_parallel(orc, hay, "wholeword", a.info()["max_keyword_len"], cap_per_unit=0.1)
    got, prof = _dev_match(a, d_hay, n, True, len(want) + 64, profile=True)
    print("C5 full size: %d matches, scan %.3f ms" % (len(got), prof["scan_ms"]))
    assert got.shape == want.shape and (got == want).all()
    del want
    # (2) position order; every record is delimited by non-word characters (T/WholeWordMatchTest.java:60-70)
    assert (np.diff(got[:, 0].astype(np.int64)) > 0).all()
    ends = got[:, 1].astype(np.int64)
    starts = got[:, 0].astype(np.int64)
    assert (WORD[hay[np.minimum(ends, n - 1)]][ends < n] == 0).all()
    assert (WORD[hay[np.maximum(starts - 1, 0)]][starts > 0] == 0).all()
    # (3) about half of the tokens are dictionary words (every one of them a match), the random words hardly ever
    n_tokens = int(np.count_nonzero((WORD[hay[1:]] != 0) & (WORD[hay[:-1]] == 0))) + 1
    assert 0.45 * n_tokens < len(got) < 0.56 * n_tokens


def test_config_c3_one_shard_full_size():
    """BASELINE config 3, shard g = 3 of 8: AhoCorasickSet (8-byte records), 2^29 units of stream 2003+3 behind the
    (max_keyword_len-1)-unit left halo a rank receives from rank g-1 (here: the tail of stream 2003+2), text_begin = 0,
    text_end = 0.  Order, membership, cross-kernel agreement and EVERY record against the oracle."""
    import torch
    c = synth.CONFIGS["C3"]
    kws = synth.config_keywords("C3")
    g, n = 3, c["n_units"]
    a = Automaton(N.MODE_ALL, kws, True)
    halo = a.info()["max_keyword_len"] - 1
    pad = (halo + 7) // 8 * 8
    buf = torch.zeros(pad + n, dtype=torch.int16, device="cuda")
    tab = np.ascontiguousarray(synth.ALPHA_LOWER)
    N.check(N.lib().acgpu_synth_fill(buf.data_ptr() + 2 * pad, n, 0, c["hay_seed"] + g, tab.ctypes.data_as(ctypes.c_void_p),
                                     len(tab), None), "synth")
    prev_tail = synth.haystack(c["hay_seed"] + g - 1, halo, start=n - halo)  # what rank g-1 sends
    buf[pad - halo:pad] = torch.from_numpy(prev_tail.view(np.int16)).cuda()
    # (one planted occurrence across the shard boundary -- 3 units in the halo, the rest owned -- so that the halo matters)
    k0 = max(kws, key=len)
    buf[pad - 3:pad - 3 + len(k0)] = torch.from_numpy(k0.view(np.int16)).cuda()
    cap = 4_000_000
    got, prof = _dev_match(a, buf, pad + n, False, cap, own=(pad, pad + n), text_begin=False, text_end=False, profile=True)
    m = len(got)
    assert got.shape[1] == 2 and 1_200_000 < m < 1_550_000 and prof["scan_kernel"].startswith("k_ac_tile")
    end, start = got[:, 1].astype(np.int64), got[:, 0].astype(np.int64)
    assert (np.diff(end * (1 << 32) + start) > 0).all()          # reference order
    assert end.min() > pad and end.max() <= pad + n              # owned: the LAST unit lies in the owned range
    assert (start < pad).any()                                    # (some match begins inside the halo: it is needed)
    # every record against the oracle: the view [halo | shard] scanned from its first unit, records that end in the shard
    view = buf[pad - halo:].cpu().numpy().view(np.uint16)
    want = oracle_parallel(Oracle(FAM_AC, kws), view, "ac", halo + 1, cap_per_unit=0.02)[:, :2]
    want = want[want[:, 1] - 1 >= halo]
    want[:, :2] += pad - halo
    assert got.shape == want.shape and (got == want).all()
    # membership: every record is an occurrence of a keyword (the Set listener contract, T/SetTest.java:156-165)
    kwset = {k.tobytes() for k in kws}
    for i in np.random.default_rng(1).integers(0, m, 5000).tolist():
        s0, e0 = got[i].tolist()
        assert view[s0 - (pad - halo):e0 - (pad - halo)].tobytes() in kwset
    # the DFA chunk scan and the split tile kernels deliver the identical stream
    for knobs in ({"force_kernel": 1}, {"force_kernel": 3}):
        for k, v in knobs.items():
            N.set_tunable(k, v)
        got2, _ = _dev_match(a, buf, pad + n, False, cap, own=(pad, pad + n), text_begin=False, text_end=False)
        assert got2.shape == got.shape and (got2 == got).all(), knobs


def test_maximum_size_haystack_just_under_2_31_units():
    """The ABI limit is the Java String limit (2^31-1 units).  A 4 GiB haystack, keyword planted in the last units."""
    import torch
    n = (1 << 31) - 3  # ragged: not a multiple of 8, exercises the scalar tail at the top of the 32-bit range
    kws = synth.config_keywords("C2")
    a = Automaton(N.MODE_ALL, kws, True)
    d_hay = torch.empty(n + 3, dtype=torch.int16, device="cuda")[:n]
    tab = np.ascontiguousarray(synth.ALPHA_LOWER)
    N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, 4242, tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
    k0 = kws[17]
    d_hay[n - len(k0):] = torch.from_numpy(k0.view(np.int16)).cuda()  # a match that ends exactly at the end of the text
    cap = 8_000_000
    got, prof = _dev_match(a, d_hay, n, True, cap, profile=True)
    assert 5_000_000 < len(got) < 5_700_000  # ~2.5e-3 per unit
    end = got[:, 1].astype(np.int64)
    assert (np.diff(end * (1 << 32) + got[:, 0]) > 0).all() and end[-1] == n and (got[:, 0] >= 0).all()
    assert got[-1].tolist() == [n - len(k0), n, 17] or got[-1, 1] == n
    assert any(r.tolist() == [n - len(k0), n, 17] for r in got[-4:])
    pre = 1 << 20
    want = Oracle(FAM_AC, kws).match(synth.haystack(4242, pre))
    assert (got[:len(want)] == want).all()
    # one unit more is refused, not mis-scanned
    d_out = torch.empty((16, 3), dtype=torch.int32, device="cuda")
    assert a.match_device(d_hay.data_ptr(), 1 << 31, True, d_out.data_ptr(), 16)[1] == N.E_INVALID


def test_async_begin_end_matches_synchronous_call():
    import torch
    kws = synth.random_keywords(11, 300, 2, 9)
    a = Automaton(N.MODE_ALL, kws, True)
    hays = [synth.haystack(90 + i, 200000 + 777 * i) for i in range(3)]
    d_hays = [torch.from_numpy(h.view(np.int16)).cuda() for h in hays]
    wants = [Oracle(FAM_AC, kws).match(h) for h in hays]
    cap = max(len(w) for w in wants) + 8
    outs = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in hays]
    st = torch.cuda.current_stream().cuda_stream
    tickets = []
    for d, o, h in zip(d_hays, outs, hays):  # three calls in flight on one stream
        tk, rc = a.match_device_begin(d.data_ptr(), h.size, True, o.data_ptr(), cap, stream=st, profile=True)
        assert rc == N.OK
        tickets.append(tk)
    for tk, o, w in zip(tickets, outs, wants):
        n, rc, prof = a.match_device_end(tk, profile=True)
        assert rc == N.OK and n == len(w) and prof["scan_ms"] > 0
        assert (o[:n].cpu().numpy() == w).all()
    # overflow is reported by _end with the exact count
    tk, rc = a.match_device_begin(d_hays[0].data_ptr(), hays[0].size, True, outs[0].data_ptr(), 5, stream=st)
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.E_OVERFLOW and n == len(wants[0])


def test_async_begin_end_for_the_chain_families():
    """acgpu_match_device_begin/_end beyond AhoCorasick: the LongestMatch walk pipeline is enqueued without a host round trip
    (count and chain exit arrive through the ticket: three calls in flight, shards chained through entry / exit, Set and Map
    records, overflow); the other families -- and LongestMatch over a selective dictionary -- run inside _begin."""
    import torch
    from ahocorasick_amd.strings import Automaton as A
    from oracle.oracle import FAM_SHORTEST, FAM_WWLONGEST
    st = torch.cuda.current_stream().cuda_stream
    kws = synth.random_keywords(32, 300, 2, 40, table=synth.ALPHA_LOWER[:2])
    lo = A(N.MODE_LONGEST, kws, True)
    assert lo.info()["tile_kernel"] == 0  # the walk pipeline
    orc = Oracle(FAM_LONGEST, kws)
    hays = [synth.haystack(140 + i, 300000 + 1111 * i, table=synth.ALPHA_LOWER[:2]) for i in range(3)]
    d_hays = [torch.from_numpy(h.view(np.int16)).cuda() for h in hays]
    wants = [orc.match(h) for h in hays]
    cap = max(len(w) for w in wants) + 8
    for with_ids in (False, True):
        cols = 3 if with_ids else 2
        outs = [torch.empty((cap, cols), dtype=torch.int32, device="cuda") for _ in hays]
        tickets = []
        for d, o, h in zip(d_hays, outs, hays):
            tk, rc = lo.match_device_begin(d.data_ptr(), h.size, with_ids, o.data_ptr(), cap, stream=st, profile=True)
            assert rc == N.OK
            tickets.append(tk)
        for tk, o, w, h in zip(tickets, outs, wants, hays):
            n, rc, prof = lo.match_device_end(tk, profile=True)
            assert rc == N.OK and n == len(w) and prof["scan_ms"] > 0 and prof["scan_kernel"].startswith("k_longest")
            assert (o[:n].cpu().numpy() == w[:, :cols]).all()
            assert tk.chain_exit >= h.size
    # shards of one buffer: every ticket's chain exit is the next shard's entry
    h, d, w = hays[0], d_hays[0], wants[0]
    out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    parts, entry = [], 0
    for lo_, hi_ in ((0, 100003), (100003, 100010), (100010, h.size)):
        tk, rc = lo.match_device_begin(d.data_ptr(), h.size, True, out.data_ptr(), cap, own=(lo_, hi_), stream=st, chain_entry=max(entry, lo_))
        assert rc == N.OK
        n, rc, _ = lo.match_device_end(tk)
        assert rc == N.OK
        parts.append(out[:n].cpu().numpy())
        entry = tk.chain_exit
        assert entry >= hi_
    assert (np.concatenate(parts) == w).all()
    tk, rc = lo.match_device_begin(d.data_ptr(), h.size, True, out.data_ptr(), 5, stream=st)
    n, rc, _ = lo.match_device_end(tk)
    assert rc == N.E_OVERFLOW and n == len(w)
    # a device result in stream order (what a multi-GPU driver gathers)
    buf = torch.zeros(4 + cap * 2, dtype=torch.int32, device="cuda")
    tk, rc = lo.match_device_begin(d.data_ptr(), h.size, False, buf.data_ptr() + 16, cap, stream=st, d_result=buf.data_ptr())
    assert rc == N.OK
    torch.cuda.current_stream().synchronize()
    assert int(buf[:2].cpu().numpy().view(np.int64)[0]) == len(w)
    assert lo.match_device_end(tk)[0] == len(w)
    # families whose call runs inside _begin: same records as the synchronous call
    kw2 = synth.random_keywords(34, 300, 2, 30, table=synth.ALPHA_LOWER[:3])
    hay2 = synth.haystack(144, 100000, table=synth.ALPHA_LOWER[:3])
    d2 = torch.from_numpy(hay2.view(np.int16)).cuda()
    sh = A(N.MODE_SHORTEST, kw2, True)
    want2 = Oracle(FAM_SHORTEST, kw2).match(hay2)
    out2 = torch.empty((len(want2) + 8, 3), dtype=torch.int32, device="cuda")
    tk, rc = sh.match_device_begin(d2.data_ptr(), hay2.size, True, out2.data_ptr(), len(want2) + 8, stream=st, profile=True)
    assert rc == N.OK
    n, rc, prof = sh.match_device_end(tk, profile=True)
    assert rc == N.OK and n == len(want2) and (out2[:n].cpu().numpy() == want2).all() and prof["scan_ms"] > 0
    sel = A(N.MODE_LONGEST, synth.random_keywords(11, 300, 4, 9), True)  # selective suffix filter: the sparse form, inside _begin
    assert sel.info()["tile_kernel"] == 1
    hay3 = synth.haystack(145, 200000)
    want3 = Oracle(FAM_LONGEST, synth.random_keywords(11, 300, 4, 9)).match(hay3)
    d3 = torch.from_numpy(hay3.view(np.int16)).cuda()
    out3 = torch.empty((len(want3) + 8, 3), dtype=torch.int32, device="cuda")
    tk, rc = sel.match_device_begin(d3.data_ptr(), hay3.size, True, out3.data_ptr(), len(want3) + 8, stream=st)
    n, rc, _ = sel.match_device_end(tk)
    assert rc == N.OK and n == len(want3) and (out3[:n].cpu().numpy() == want3).all()


@pytest.mark.parametrize("family", ["ac", "wholeword", "longest", "shortest", "wwlongest"])
def test_host_entry_pipelined_over_chunks_equals_the_oracle(family):
    """acgpu_match_u16 -- what StringSet/StringMap.match(String, ...) binds -- on a haystack of several 2^24-unit chunks: the
    text travels through the pinned staging ring while the chunks that have arrived are scanned as shards (halos, chain entry
    handed from shard to shard); records and the overflow protocol as for the plain call."""
    from oracle.oracle import FAM_SHORTEST, FAM_WWLONGEST
    n = (1 << 25) + (1 << 24) + 12345  # four chunks, the last one short
    if family == "ac":
        kws = synth.random_keywords(33, 2000, 3, 11, table=synth.ALPHA_LOWER[:12])
        hay = synth.haystack(401, n, table=synth.ALPHA_LOWER[:12])
        hay[(1 << 24) - 5:(1 << 24) + 6] = np.concatenate([kws[0], kws[1], kws[2]])[:11]  # matches across a chunk boundary
        auto, orc = Automaton(N.MODE_ALL, kws, True), Oracle(FAM_AC, kws)
        want = oracle_parallel(orc, hay, "ac", max(len(k) for k in kws), cap_per_unit=0.05)
    elif family == "wholeword":
        table = np.array([ord(c) for c in "abcdE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(31, 300, 1, 6, table=table[:5])
        hay = synth.haystack(402, n, table=table)
        auto = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=WORD)
        orc = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=WORD)
        want = oracle_parallel(orc, hay, "wholeword", max(len(k) for k in kws), cap_per_unit=0.2)
    elif family == "longest":
        kws = synth.random_keywords(32, 300, 2, 40, table=synth.ALPHA_LOWER[:2])
        hay = synth.haystack(403, n, table=synth.ALPHA_LOWER[:2])
        auto, orc = Automaton(N.MODE_LONGEST, kws, True), Oracle(FAM_LONGEST, kws)
        want = orc.match(hay, cap=n // 2)
    elif family == "shortest":
        kws = synth.random_keywords(34, 300, 2, 30, table=synth.ALPHA_LOWER[:3])
        hay = synth.haystack(404, n, table=synth.ALPHA_LOWER[:3])
        auto, orc = Automaton(N.MODE_SHORTEST, kws, True), Oracle(FAM_SHORTEST, kws)
        want = orc.match(hay, cap=n // 2)
    else:
        kws, hay = _wwl_case(405, n)
        auto = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD)
        want = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay, cap=n // 4)
    assert len(want) > 1000
    got = auto.match_host(hay, True, cap=len(want) + 16)
    assert got.shape == want.shape and (got == want).all()
    got2 = auto.match_host(hay, False, cap=1000)  # too small: the call reports the capacity to retry with, and the retry delivers
    assert got2.shape == (len(want), 2) and (got2 == want[:, :2]).all()
    N.set_tunable("tile_debug", 33554432)  # the plain form: one copy, one scan
    got3 = auto.match_host(hay, True, cap=len(want) + 16)
    assert (got3 == want).all()


@pytest.mark.parametrize("family", ["ac", "ac_ci", "wholeword", "longest", "shortest", "wwlongest"])
def test_batch_of_short_haystacks_equals_one_call_per_haystack(family):
    """acgpu_match_batch_u16: many short haystacks in one device call (separator units between them) report, haystack by
    haystack, exactly what the reference's match(String) reports for each of them alone -- empty haystacks, haystacks that end
    in the middle of a keyword / a word, keywords that would match across a boundary."""
    from oracle.oracle import FAM_SHORTEST, FAM_WWLONGEST
    rng = np.random.default_rng(77)
    if family in ("ac", "ac_ci"):
        cs = family == "ac"
        table = synth.ALPHA_LOWER[:6] if cs else np.array([ord(c) for c in "abcABC"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(51, 400, 1, 9, table=table)
        auto = Automaton(N.MODE_ALL, kws, cs)
        orc = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER)
    elif family == "wholeword":
        table = np.array([ord(c) for c in "abcdE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(52, 300, 1, 6, table=table[:5])
        auto = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=WORD)
        orc = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=WORD)
    elif family == "longest":
        table = synth.ALPHA_LOWER[:2]
        kws = synth.random_keywords(53, 300, 1, 30, table=table)
        auto, orc = Automaton(N.MODE_LONGEST, kws, True), Oracle(FAM_LONGEST, kws)
    elif family == "shortest":
        table = synth.ALPHA_LOWER[:3]
        kws = synth.random_keywords(54, 300, 2, 20, table=table)
        auto, orc = Automaton(N.MODE_SHORTEST, kws, True), Oracle(FAM_SHORTEST, kws)
    else:
        table = np.array([ord(c) for c in "abcE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws, _ = _wwl_case(55, 10)
        auto = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD)
        orc = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD)
    hays = [table[rng.integers(0, len(table), int(ln))] for ln in rng.integers(0, 300, 400)]
    hays[3] = np.zeros(0, np.uint16)
    hays[4] = np.zeros(0, np.uint16)
    hays[10] = np.asarray(kws[0])[: max(1, len(kws[0]) - 1)]  # ends inside a keyword; the next one begins with its tail
    hays[11] = np.asarray(kws[0])[max(1, len(kws[0]) - 1):]
    want = []
    for i, h in enumerate(hays):
        r = orc.match(h)
        want.append(np.concatenate([np.full((len(r), 1), i, np.int32), r], axis=1))
    want = np.concatenate(want)
    assert len(want) > 500
    got = auto.match_batch(hays, True, cap=len(want) + 8)
    assert got.shape == want.shape and (got == want).all()
    got2 = auto.match_batch(hays, False, cap=16)  # overflow: the call reports the capacity to retry with
    assert got2.shape == (len(want), 3) and (got2 == want[:, :3]).all()
    assert auto.match_batch([], True).shape == (0, 4) and auto.match_batch([hays[3]], True).shape == (0, 4)


def test_batch_wwlongest_keyword_without_word_characters_starts_a_walk_at_every_haystack():
    """A keyword without word characters is kept untrimmed (R/WordCharacters.java:41-62), so the root has a transition on a
    non-word unit; the reference's scan starts at position 0 whatever stands there, so in ' éa,' the walk that begins on the
    space swallows it and the word behind it is skipped to its end -- 'éa' is NOT reported, while '.éa,' reports it.  In a batch
    every haystack's first unit has to be such a walk start (found by the GPU fuzz, seed 2026)."""
    from oracle.oracle import FAM_WWLONGEST
    kws = [np.array([ord(c) for c in k], dtype=np.uint16) for k in (" ", "Éa", "b", ", ", "b a")]
    auto = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD)
    orc = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD)
    texts = ["b.", " éa, b", ".éa, b", "", " ", "  b a", ", b", "éa", " b"] * 40
    hays = [np.array([ord(c) for c in s], dtype=np.uint16) for s in texts]
    want = []
    for i, h in enumerate(hays):
        r = orc.match(h)
        want.append(np.concatenate([np.full((len(r), 1), i, np.int32), r.reshape(-1, 3)], axis=1))
    want = np.concatenate(want)
    assert [1, 1, 3, 1] not in want.tolist() and [2, 1, 3, 1] in want.tolist()
    got = auto.match_batch(hays, True, cap=len(want) + 8)
    assert got.shape == want.shape and (got == want).all()


def test_batch_facade_listener_and_a_dictionary_without_a_free_unit():
    m = AhoCorasickMap(["he", "she", "hers"], ["HE", "SHE", "HERS"], True)
    seen = []
    m.match_batch(["ushers", "", "she he"], lambda h, s, e, v: seen.append((h, s, e, v)) or not (h == "she he" and v == "SHE"))
    assert seen == [("ushers", 1, 4, "SHE"), ("ushers", 2, 4, "HE"), ("ushers", 2, 6, "HERS"), ("she he", 0, 3, "SHE")]
    # every one of the 65536 units is a keyword: no separator exists, the batch falls back to one call per haystack
    every = AhoCorasickSet([np.array([i], dtype=np.uint16) for i in range(65536)], True)
    got = every.automaton.match_batch([np.array([5, 6], np.uint16), np.array([7], np.uint16)], False).tolist()
    assert got == [[0, 0, 1], [0, 1, 2], [1, 0, 1]]


# ---- ShardedMatcher (ahocorasick_amd/dist.py) through the native scan: the ranks of one job emulated in one process ----

def _emulated_ranks(auto, whole, world, chain_window=4096):
    """What `world` ranks of ShardedMatcher.step() compute, minus the collectives: halos filled from the whole text,
    the chain families' all-gather of exits emulated rank by rank.  Returns the concatenation with global positions + repair count.
    (tests/test_dist_gpu.py runs the same thing in real separate processes.)"""
    import torch
    from ahocorasick_amd.dist import ShardedMatcher
    n = whole.size // world
    d_whole = torch.from_numpy(whole.view(np.int16)).cuda()
    box = [0] * world  # the exits (global) of the ranks done so far
    parts, repairs = [], 0
    for g in range(world):
        m = ShardedMatcher(auto, n, with_ids=True, cap=64)
        m.rank, m.world, m.collective = g, world, True
        m.chain_window = chain_window
        sb = m.sb
        sb.own.copy_(d_whole[g * n:(g + 1) * n])
        if g > 0 and sb.halo:
            sb.halo_view().copy_(d_whole[g * n - sb.halo:g * n])
        if g + 1 < world and sb.right:
            sb.right_view().copy_(d_whole[(g + 1) * n:(g + 1) * n + sb.right])
        m._gather_exits = lambda ex, g=g: (box.__setitem__(g, int(ex)), list(box))[1]  # (ranks before g: their final exits)
        cnt, _ = m._scan(False, m._new_step())
        r = m.out[:cnt].cpu().numpy().astype(np.int64)
        r[:, :2] += g * n - m.shift  # (records are relative to the rank's view of its buffer)
        parts.append(r)
        repairs += m.chain_repairs
    return np.concatenate(parts), repairs


@pytest.mark.parametrize("family", ["ac", "longest", "wholeword", "shortest"])
def test_sharded_matcher_native_scan_equals_whole_text(family):
    world, n = 4, 50000
    if family == "wholeword":
        table = np.array([ord(c) for c in "abcdE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(31, 300, 1, 6, table=table[:5])
        whole = synth.haystack(41, world * n, table=table)
        auto = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=WORD)
        want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(whole)
    elif family == "longest":
        kws = synth.random_keywords(32, 300, 2, 40, table=synth.ALPHA_LOWER[:2])
        whole = synth.haystack(42, world * n, table=synth.ALPHA_LOWER[:2])
        auto = Automaton(N.MODE_LONGEST, kws, True)
        want = Oracle(FAM_LONGEST, kws).match(whole)
    elif family == "shortest":
        from oracle.oracle import FAM_SHORTEST
        kws = synth.random_keywords(34, 300, 2, 30, table=synth.ALPHA_LOWER[:3])
        whole = synth.haystack(44, world * n, table=synth.ALPHA_LOWER[:3])
        auto = Automaton(N.MODE_SHORTEST, kws, True)
        want = Oracle(FAM_SHORTEST, kws).match(whole)
    else:
        kws = synth.random_keywords(33, 500, 2, 11, table=synth.ALPHA_LOWER[:8])
        whole = synth.haystack(43, world * n, table=synth.ALPHA_LOWER[:8])
        auto = Automaton(N.MODE_ALL, kws, True)
        want = Oracle(FAM_AC, kws).match(whole)
    got, repairs = _emulated_ranks(auto, whole, world, chain_window=32)
    assert got.shape == want.shape and (got == want.astype(np.int64)).all()
    if family in ("longest", "shortest"):
        assert repairs > 0  # some shard boundary fell inside a match: the window repair ran on the device


# ---- match(Readable, ReadableMatchListener): acgpu_stream_* ------------------------------------------------------------

def _stream_all(auto, hay, cuts, with_ids=True):
    """The records of all feeds, concatenated -- from the synchronous form; the pipelined form (a feed returns the previous
    chunk's records, copy / transfer / scan overlap) and the pipelined form fed through acgpu_stream_reserve (the chunk written
    straight into the staging memory) must deliver the very same list."""
    from ahocorasick_amd import Stream
    edges = [0] + list(cuts) + [hay.size]
    results = []
    for form in ("sync", "pipelined", "reserved"):
        st = Stream(auto, with_ids=with_ids, pipelined=form != "sync")
        parts = []
        for i, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
            chunk = hay[lo:hi]
            if form == "reserved" and hi > lo:
                view = st.reserve(hi - lo)
                view[:] = chunk
                chunk = view
            parts.append(st.feed(chunk, final=(i == len(edges) - 2), cap=8))  # cap=8: the overflow/retry protocol
        st.close()
        results.append(np.concatenate(parts))
        if form == "pipelined" and len(edges) > 2:
            assert len(parts[0]) == 0  # the first feed has no previous chunk to report
    assert results[1].shape == results[0].shape and (results[1] == results[0]).all(), "pipelined feeds differ"
    assert results[2].shape == results[0].shape and (results[2] == results[0]).all(), "reserved feeds differ"
    return results[0]


@pytest.mark.parametrize("family", ["ac", "longest", "wholeword"])
@pytest.mark.parametrize("seed", range(3))
def test_stream_feeds_equal_whole_text(family, seed):
    """T/MapTest.java:178-188: the Readable overload reports what the String overload reports -- here with positions,
    for chunkings that include empty chunks, one-unit chunks and chunks shorter than the keywords."""
    rng = np.random.default_rng(300 + seed)
    if family == "wholeword":
        table = np.array([ord(c) for c in "abcdE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(61 + seed, 200, 1, 9, table=table[:5])
        hay = synth.haystack(71 + seed, 60000, table=table)
        auto = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=WORD)
        want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay)
    elif family == "longest":
        kws = synth.random_keywords(62 + seed, 300, 2, 30, table=synth.ALPHA_LOWER[:2])
        hay = synth.haystack(72 + seed, 60000, table=synth.ALPHA_LOWER[:2])
        auto = Automaton(N.MODE_LONGEST, kws, True)
        want = Oracle(FAM_LONGEST, kws).match(hay)
    else:
        kws = synth.random_keywords(63 + seed, 400, 2, 11, table=synth.ALPHA_LOWER[:8])
        hay = synth.haystack(73 + seed, 60000, table=synth.ALPHA_LOWER[:8])
        auto = Automaton(N.MODE_ALL, kws, True)
        want = Oracle(FAM_AC, kws).match(hay)
    chunkings = [[], [30000], sorted(rng.integers(0, hay.size, 12).tolist()),
                 [1, 2, 3, 3, 4, 10, 11, 40, 5000, 5001, 59999], list(range(100, 400, 7))]
    for cuts in chunkings:
        got = _stream_all(auto, hay, cuts)
        assert got.shape == want.shape and (got == want.astype(np.int64)).all(), (family, cuts[:5])
    got2 = _stream_all(auto, hay, chunkings[2], with_ids=False)
    assert (got2 == want[:, :2].astype(np.int64)).all()


def test_readable_overload_values_and_early_stop():
    import io
    m = AhoCorasickMap(["he", "she", "hers"], ["HE", "SHE", "HERS"], True)
    seen = []
    m.match(io.StringIO("ushers and she"), lambda v: seen.append(v) or True)  # Readable: value-only listener
    assert seen == ["SHE", "HE", "HERS", "SHE", "HE"]
    seen = []
    m.match_readable(iter(["ush", "", "ers and", " she"]), lambda v: seen.append(v) or len(seen) < 2)
    assert seen == ["SHE", "HE"]
    w = WholeWordMatchMap(["Foo", "bar-baz"], [1, 2], False)
    seen = []
    w.match(io.StringIO("foo FOO, bar-baz! foobar"), lambda v: seen.append(v) or True)
    assert seen == [1, 1, 2]
    lo = LongestMatchMap(["a", "ab", "abc", "bcd"], [1, 2, 3, 4], True)
    seen = []
    lo.match_readable(iter(["ab", "cd", "a"]), lambda v: seen.append(v) or True)
    assert [v - 1 for v in seen] == Oracle(FAM_LONGEST, ["a", "ab", "abc", "bcd"]).match_readable("abcda").tolist() == [2, 0]
    # the oracle's Readable restatement agrees on a larger case
    kws = synth.random_keywords(5, 300, 2, 8, table=synth.ALPHA_LOWER[:6])
    hay = synth.haystack(6, 20000, table=synth.ALPHA_LOWER[:6])
    big = AhoCorasickMap(kws, _ids(len(kws)), True)
    seen = []
    big.match_readable(iter([hay[:7777], hay[7777:]]), lambda v: seen.append(v) or True)
    assert seen == Oracle(FAM_AC, kws).match_readable(hay).tolist()


def test_stream_fold_inconsistent_wholeword_folds_in_every_lookup():
    """WholeWordMatchMap.match(Readable) folds in EVERY word-character lookup (S/WholeWordMatchMap.java:112,117 and scroll()
    :328), so with a table that is not fold-consistent it is an ordinary scan over w' = wordChars o lower -- unlike the String
    loop, which mixes raw and folded lookups (:204,:209 vs :221,:226) and keeps the sequential kernel.  Two dictionary shapes:
    folded keywords made of word characters only (k_ww_tile over w') and folded keywords that hold units which are no word
    characters to the folding scan (the WholeWordLongest walk without fail matches)."""
    rng = np.random.default_rng(91)
    alpha = np.array([ord(c) for c in "abxyABXY ,."], dtype=np.uint16)
    for wchars, clean in (("ABXYxy", False), ("abxyAB", True)):
        # "ABXYxy": A, B are word characters, a, b are not -> a keyword "AB" folds to units the folding scan does not take for
        # word characters.  "abxyAB": X, Y are not word characters but fold to x, y which are -> every folded unit is one.
        wc = np.zeros(65536, np.uint8)
        for ch in wchars:
            wc[ord(ch)] = 1
        for it in range(12):
            pool = [ord(c) for c in wchars]
            kws = [np.array([pool[int(j)] for j in rng.integers(0, len(pool), int(rng.integers(1, 5)))], dtype=np.uint16)
                   for _ in range(10)]
            hay = alpha[rng.integers(0, len(alpha), int(rng.integers(1, 4000)))]
            a = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=wc)
            assert a.info()["fold_consistent"] == 0
            orc = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=wc)
            want = orc.match_readable(hay, 7, positions=True)
            assert (want == orc.match_readable(hay, 1024, positions=True)).all()
            for cuts in ([], [hay.size // 2], sorted(rng.integers(0, hay.size, 6).tolist()), list(range(1, min(hay.size, 40), 3))):
                got = _stream_all(a, hay, cuts)
                assert got.shape == want.shape and (got == want.astype(np.int64)).all(), (wchars, it, cuts[:4])
            # the String overload keeps the reference's mixed lookups: a different record list on some inputs
            want_s = orc.match(hay)
            got_s = a.match_host(hay, True)
            assert got_s.shape == want_s.shape and (got_s == want_s).all()
    w = WholeWordMatchMap(["AB", "x"], [1, 2], False, word_characters=list("ABXYxy"))
    seen = []
    w.match(iter(["AB x,", " ab X"]), lambda v: seen.append(v) or True)
    orc = Oracle(FAM_WHOLEWORD, ["AB", "x"], case_sensitive=False, lower=LOWER, word_chars=w.get_word_chars())
    assert [v - 1 for v in seen] == orc.match_readable("AB x, ab X").tolist()


# ---- ShortestMatchSet / ShortestMatchMap (ACGPU_MODE_SHORTEST) ----------------------------------------------------------

def test_fixtures_shortest(fixtures, ac_kernel):
    from ahocorasick_amd import ShortestMatchMap, ShortestMatchSet
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        if "keywords_gen" not in fx:
            kws = fx["S_keywords"]
        got = ShortestMatchMap(kws, _ids(len(kws)), True).find_all(hay)
        assert got.tolist() == fx["S"], fx["name"]
        assert len(got) == fx["S_count"]  # the count T/ShortestMatchTest.java:30-42 expects
        assert ShortestMatchSet(kws, True).find_all(hay).tolist() == [r[:2] for r in fx["S"]], fx["name"]


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_shortest_vs_oracle(seed, ac_kernel):
    from oracle.oracle import FAM_SHORTEST
    rng = np.random.default_rng(500 + seed)
    alphabets = [[97, 98], [97, 98, 99], [97, 98, 65, 66, 0x00E9, 0x00C9], list(range(97, 123))]
    for it in range(12):
        alpha = alphabets[it % len(alphabets)]
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 60)), int(rng.choice([1, 3, 8])), int(rng.choice([0, 1, 50, 3000, 70001])),
                             min_len=1)
        for cs in (True, False):
            want = Oracle(FAM_SHORTEST, kws, case_sensitive=cs, lower=LOWER).match(hay)
            got = Automaton(N.MODE_SHORTEST, kws, cs).match_host(hay, True, cap=16)
            assert got.shape == want.shape and (got == want).all(), (seed, it, cs)


def test_shortest_listener_first_duplicate_and_early_stop():
    from ahocorasick_amd import ShortestMatchMap, ShortestMatchSet
    m = ShortestMatchMap(["ab", "abcd", "ab", "d"], ["first", "long", "second", "D"], True)
    seen = []
    m.match("abcd ab", lambda h, s, e, v: seen.append((s, e, v)) or True)
    assert seen == [(0, 2, "first"), (3, 4, "D"), (5, 7, "first")]  # "abcd" never matches: its prefix ends first
    seen = []
    ShortestMatchSet(["a"], True).match("aaaa", lambda h, s, e: seen.append((s, e)) or len(seen) < 2)
    assert seen == [(0, 1), (1, 2)]
    assert ShortestMatchSet(["abcd", "bc", "d"], True).find_all("abcd").tolist() == [[1, 3], [3, 4]]  # earliest END wins


def test_shortest_shards_and_stream_carry_the_restart_position():
    import torch
    from oracle.oracle import FAM_SHORTEST
    kws = synth.random_keywords(44, 300, 1, 9, table=synth.ALPHA_LOWER[:4])
    hay = synth.haystack(45, 150000, table=synth.ALPHA_LOWER[:4])
    a = Automaton(N.MODE_SHORTEST, kws, True)
    want = Oracle(FAM_SHORTEST, kws).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    cap = len(want) + 8
    cuts = [0, 50001, 50002, 110003, hay.size]
    parts, entry = [], 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
        n_out, rc, _, ex = a.match_device(d_hay.data_ptr(), hay.size, True, d_out.data_ptr(), cap, own=(lo, hi), chain_entry=entry)
        assert rc == N.OK
        parts.append(d_out[:n_out].cpu().numpy())
        assert ex == (parts[-1][-1, 1] if n_out else entry)
        entry = ex
    assert (np.concatenate(parts) == want).all()
    for cuts in ([], [1, 2, 3, 40, 41, 90000], list(range(1000, 1100, 3))):
        got = _stream_all(a, hay, cuts)
        assert got.shape == want.shape and (got == want.astype(np.int64)).all()


def test_shortest_dense_overlaps_many_matches_per_position():
    """a^k dictionaries: up to 100 matches end at every position; the selection's group search must stay exact."""
    from oracle.oracle import FAM_SHORTEST
    kws = ["a" * k for k in range(2, 60)] + ["b", "ab" * 5]
    hay = synth.haystack(46, 20000, table=synth.ALPHA_AB_75)
    want = Oracle(FAM_SHORTEST, kws).match(hay)
    got = Automaton(N.MODE_SHORTEST, kws, True).match_host(hay, True)
    assert got.shape == want.shape and (got == want).all()


# ---- WholeWordLongestMatchSet / Map (ACGPU_MODE_WWLONGEST) --------------------------------------------------------------

def test_fixtures_wwlongest(fixtures):
    from ahocorasick_amd import WholeWordLongestMatchMap, WholeWordLongestMatchSet
    for fx in fixtures:
        if "keywords_gen" in fx:
            continue
        kws = fx["WWL_keywords"]
        got = WholeWordLongestMatchMap(kws, _ids(len(kws)), True).find_all(fx["haystack"])
        assert got.tolist() == fx["WWL"], fx["name"]
        assert len(got) == fx["WWL_count"]  # the count T/WholeWordLongestMatchTest.java:46-65 expects
        assert WholeWordLongestMatchSet(kws, True).find_all(fx["haystack"]).tolist() == [r[:2] for r in fx["WWL"]]


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_wwlongest_vs_oracle(seed):
    from oracle.oracle import FAM_WWLONGEST
    rng = np.random.default_rng(700 + seed)
    alpha = [ord(c) for c in "abAB  -."] + [0x00E9, 0x00C9]
    for it in range(10):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 40)), int(rng.choice([3, 6, 12])),
                             int(rng.choice([0, 1, 2, 60, 5000, 70001])))
        for cs in (True, False):
            want = Oracle(FAM_WWLONGEST, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay)
            got = Automaton(N.MODE_WWLONGEST, kws, cs, word_chars=WORD).match_host(hay, True, cap=16)
            assert got.shape == want.shape and (got == want).all(), (seed, it, cs)


def test_wwlongest_listener_examples_and_limits():
    from ahocorasick_amd import WholeWordLongestMatchMap, WholeWordLongestMatchSet
    m = WholeWordLongestMatchMap(["as", "if", "as if"], [1, 2, 3], True)
    seen = []
    m.match("as if, as in; ax if", lambda h, s, e, v: seen.append((s, e, v)) or True)
    assert seen == [(0, 5, 3), (7, 9, 1), (17, 19, 2)]
    seen = []
    WholeWordLongestMatchSet(["a"], True).match("a a a", lambda h, s, e: seen.append((s, e)) or len(seen) < 2)
    assert seen == [(0, 1), (2, 3)]
    # natural-language sized case: multi-word keywords over a token stream
    words = [w for w in synth.config_keywords("C5")[:3000]]
    sp = np.array([32], dtype=np.uint16)
    kws = words[:2000] + [np.concatenate([words[i], sp, words[i + 1]]) for i in range(0, 1000, 2)]
    block = synth.mixed_script_haystack(77, 1 << 18, words, swapcase_tbl=synth.swapcase_table())
    from oracle.oracle import FAM_WWLONGEST
    want = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(block)
    got = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD).match_host(block, True)
    assert len(want) > 1000 and got.shape == want.shape and (got == want).all()


def _wwl_case(seed, n, cs=False):
    """multi-word keywords over a token stream with punctuation: walks run across words and over shard boundaries"""
    table = np.array([ord(c) for c in "abcE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
    rng = np.random.default_rng(seed)
    words = synth.random_keywords(seed, 120, 1, 5, table=table[:4])
    sp = np.array([32], dtype=np.uint16)
    kws = list(words[:60]) + [np.concatenate([words[int(i)], sp, words[int(j)]]) for i, j in rng.integers(0, 120, (80, 2))] + \
          [np.concatenate([words[int(i)], sp, words[int(j)], np.array([44, 32], np.uint16), words[int(k)]])
           for i, j, k in rng.integers(0, 120, (30, 3))]
    hay = synth.haystack(seed + 1000, n, table=table)
    return kws, hay


def test_wwlongest_shards_chain_through_entry_and_exit():
    """acgpu_shard for WWLONGEST: a walk belongs to the shard that owns its first unit; chain_exit of one shard is the
    chain_entry of the next.  Shards of one buffer, and a shard handed only its slice + halos."""
    import torch
    from oracle.oracle import FAM_WWLONGEST
    for seed in range(3):
        kws, hay = _wwl_case(500 + seed, 120000)
        a = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD)
        want = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay)
        assert len(want) > 2000
        d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
        cap = len(want) + 16
        got, _ = _dev_match(a, d_hay, hay.size, True, cap)
        assert got.shape == want.shape and (got == want).all()
        cuts = [0, 39989, 40000, 40003, 90001, hay.size]  # (some shards are only a few units long)
        parts, entry = [], 0
        d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            n_out, rc, _, ex = a.match_device(d_hay.data_ptr(), hay.size, True, d_out.data_ptr(), cap, own=(lo, hi), chain_entry=entry)
            assert rc == N.OK and ex >= entry
            parts.append(d_out[:n_out].cpu().numpy())
            entry = ex
        assert (np.concatenate(parts) == want).all()
        # a rank's view: 1 unit of left context, max_keyword_len + 1 units of right halo, positions buffer-relative
        ml = a.info()["max_keyword_len"]
        lo, hi = cuts[1], cuts[4]
        base = (lo - 1) // 8 * 8
        sub = d_hay[base:hi + ml + 1].clone()
        entry = 0
        for l2, h2 in zip(cuts[:1], cuts[1:2]):  # the entry the previous shards leave
            _, _, _, entry = a.match_device(d_hay.data_ptr(), hay.size, True, d_out.data_ptr(), cap, own=(l2, h2), chain_entry=0)
        n_out, rc, _, ex = a.match_device(sub.data_ptr(), sub.numel(), True, d_out.data_ptr(), cap, own=(lo - base, hi - base),
                                          text_begin=False, text_end=False, chain_entry=entry - base)
        assert rc == N.OK
        p = d_out[:n_out].cpu().numpy()
        p[:, :2] += base
        assert (p == np.concatenate(parts[1:4])).all()
        # too short a right halo is refused
        assert a.match_device(sub.data_ptr(), sub.numel() - 2, True, d_out.data_ptr(), cap, own=(lo - base, hi - base),
                              text_begin=False, text_end=False, chain_entry=entry - base)[1] == N.E_INVALID


def test_wwlongest_fold_inconsistent_tables_set_and_map_flavours():
    """Custom word characters in case-insensitive mode where wordChars[c] != wordChars[lower(c)].  WholeWordLongestMatchSet's
    String loop tests the folded unit where a walk stops and RAW units in its skip loops (S/WholeWordLongestMatchSet.java:126,
    151,156): sequential kernel.  WholeWordLongestMatchMap's String loop folds there too (S/WholeWordLongestMatchMap.java:283,
    288), and so does its Readable loop (:404): position-parallel scans over w' = wordChars o lower -- whole text, shards and
    streams.  The two classes report different matches on the same input."""
    import torch
    from ahocorasick_amd import WholeWordLongestMatchMap, WholeWordLongestMatchSet
    from oracle.oracle import FAM_WWLONGEST
    from ahocorasick_amd.unicode_tables import word_chars_from_list
    wc = word_chars_from_list("abcdxyABCD")  # X, Y are not word characters although x, y are
    rng = np.random.default_rng(9)
    alpha = np.array([ord(c) for c in "abxyABXY ,"], dtype=np.uint16)
    differ = 0
    for it in range(30):
        kws = [alpha[rng.integers(0, 4, int(rng.integers(1, 5)))] for _ in range(12)]
        kws += [np.concatenate([kws[0], np.array([32], np.uint16), kws[1]])]
        hay = alpha[rng.integers(0, len(alpha), int(rng.integers(1, 3000)))]
        ms = WholeWordLongestMatchSet(kws, False, word_characters="abcdxyABCD")
        mm = WholeWordLongestMatchMap(kws, list(range(len(kws))), False, word_characters="abcdxyABCD")
        assert mm.automaton.info()["fold_consistent"] == 0
        o_set = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=wc)
        o_map = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=wc, map_flavour=True)
        want_s, want_m = o_set.match(hay), o_map.match(hay)
        got_s, got_m = ms.find_all(hay), mm.find_all(hay)
        assert got_s.shape == want_s[:, :2].shape and (got_s == want_s[:, :2]).all(), it
        assert got_m.shape == want_m.shape and (got_m == want_m).all(), it
        differ += int(want_s.shape != want_m.shape or not (want_s == want_m).all())
        # Map flavour: shards of one buffer chain through entry / exit
        a = mm.automaton
        d_hay = torch.from_numpy(np.concatenate([hay, np.zeros(8, np.uint16)]).view(np.int16)).cuda()
        cap = len(want_m) + 16
        d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
        cuts = sorted(set([0, hay.size] + rng.integers(0, hay.size + 1, 3).tolist()))
        parts, entry = [], 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            n_out, rc, _, ex = a.match_device(d_hay.data_ptr(), hay.size, True, d_out.data_ptr(), cap, own=(lo, hi), chain_entry=entry)
            assert rc == N.OK
            parts.append(d_out[:n_out].cpu().numpy())
            entry = ex
        assert (np.concatenate(parts) == want_m).all(), (it, cuts)
        # a shard of the Set flavour is refused (its loop is history dependent), the whole text is not
        if hay.size > 16:
            assert a.match_device(d_hay.data_ptr(), hay.size, False, d_out.data_ptr(), cap, own=(0, 8))[1] == N.E_UNSUPPORTED
        # streams: the Readable loop folds everywhere -- the Map flavour's records
        assert (o_map.match_readable(hay, 5, positions=True) == want_m).all()
        for cuts in ([], sorted(rng.integers(0, hay.size, 5).tolist())):
            got = _stream_all(a, hay, cuts)
            assert got.shape == want_m.shape and (got == want_m.astype(np.int64)).all(), (it, cuts)
    assert differ >= 3  # the inputs do tell the two classes apart


@pytest.mark.parametrize("seed", range(3))
def test_wwlongest_stream_feeds_equal_whole_text(seed):
    """WholeWordLongestMatchMap.match(Readable, ...) (S/WholeWordLongestMatchMap.java:54-181): the feeds' records are the
    String overload's; the values are what the oracle's literal Readable loop reports."""
    from oracle.oracle import FAM_WWLONGEST
    kws, hay = _wwl_case(700 + seed, 50000)
    auto = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD)
    orc = Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD)
    want = orc.match(hay)
    assert orc.match_readable(hay, 7).tolist() == want[:, 2].tolist()
    rng = np.random.default_rng(seed)
    for cuts in ([], [25000], sorted(rng.integers(0, hay.size, 12).tolist()), [1, 2, 3, 3, 4, 10, 11, 40, 5000, 5001, 49999],
                 list(range(100, 400, 7))):
        got = _stream_all(auto, hay, cuts)
        assert got.shape == want.shape and (got == want.astype(np.int64)).all(), cuts[:5]
    from ahocorasick_amd import WholeWordLongestMatchMap
    m = WholeWordLongestMatchMap(kws, list(range(len(kws))), False)
    seen = []
    text = "".join(chr(c) for c in hay[:5000].tolist())
    m.match(iter([text[:777], text[777:3000], "", text[3000:]]), lambda v: seen.append(v) or True)
    assert seen == orc.match(hay[:5000])[:, 2].tolist()


def test_split_form_falls_back_when_the_candidate_slices_overflow():
    """Every 4-gram over {a,b} is a keyword, so every position passes the filter: the split form's candidate slices
    overflow and the call is redone with the fused kernel -- synchronously and through begin/end."""
    import torch
    kws = ["".join(p) for p in __import__("itertools").product("ab", repeat=4)] + ["abababab"]
    a = Automaton(N.MODE_ALL, kws, True)
    assert a.info()["filter_k"] == 4
    hay = synth.haystack(9, 300000, table=synth.ALPHA_LOWER[:2])
    want = Oracle(FAM_AC, kws).match(hay)
    N.set_tunable("force_kernel", 3)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    cap = len(want) + 8
    got, prof = _dev_match(a, d_hay, hay.size, True, cap, profile=True)
    # the fused kernel delivered the result (4th template argument = SPLIT)
    assert prof["scan_kernel"].split("<")[1].rstrip(">").split(", ")[3] == "false"
    assert got.shape == want.shape and (got == want).all()
    out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, out.data_ptr(), cap, stream=torch.cuda.current_stream().cuda_stream)
    assert rc == N.OK
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.OK and n == len(want) and (out[:n].cpu().numpy() == want).all()


def test_longest_walk_beyond_the_lds_rows_takes_the_global_table():
    """Keywords deeper than the trie rows that fit LDS (about 6100 rows at 3 classes): those walks are redone through
    the table in global memory."""
    kws = ["a" * 7000, "a" * 6500, "ab", "b"]
    hay = np.concatenate([np.full(7200, ord("a")), np.array([ord("b")] * 3), np.full(6600, ord("a")), np.array([ord("b")]),
                          np.full(100, ord("a"))]).astype(np.uint16)
    want = Oracle(FAM_LONGEST, kws).match(hay)
    got = LongestMatchMap(kws, _ids(len(kws)), True).find_all(hay)
    assert got.shape == want.shape and (got == want).all()
    got_set = LongestMatchSet(kws, True).find_all(hay)
    assert (got_set == want[:, :2]).all()


def test_longest_over_selective_dictionaries_is_a_selection_of_all_matches():
    """Dictionaries with a selective suffix filter take the AhoCorasick tile pipeline + leftmost-longest selection
    instead of the trie walk: same records, shards, streams; a haystack dense in matches falls back to the walk."""
    import torch
    kws = synth.random_keywords(81, 3000, 4, 9)
    hay = synth.haystack(82, 400000)
    for i in range(0, 4000):  # plant overlapping and nested keywords
        k = kws[(i * 7) % len(kws)]
        p = (i * 97) % (hay.size - 16)
        hay[p:p + len(k)] = k
    a = Automaton(N.MODE_LONGEST, kws, True)
    assert a.info()["filter_k"] == 4 and a.info()["tile_kernel"] == 1
    want = Oracle(FAM_LONGEST, kws).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    d_out = torch.empty((len(want) + 64, 3), dtype=torch.int32, device="cuda")

    def dm(**kw):
        n_out, rc, prof, ex = a.match_device(d_hay.data_ptr(), hay.size, True, d_out.data_ptr(), d_out.shape[0], **kw)
        assert rc == N.OK
        return d_out[:n_out].cpu().numpy(), prof, ex

    got, prof, _ = dm(profile=True)
    assert prof["scan_kernel"].startswith("k_ac_tile")
    assert got.shape == want.shape and (got == want).all()
    N.set_tunable("force_kernel", 1)  # the walk gives the same stream
    got_w, prof_w, _ = dm(profile=True)
    assert prof_w["scan_kernel"].startswith("k_longest_walk") and (got_w == want).all()
    N.set_tunable("force_kernel", 0)
    # shards chained through chain_entry/exit, and streams
    cuts = [0, 100001, 100002, 250003, hay.size]
    parts, entry = [], 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p, _, ex = dm(own=(lo, hi), chain_entry=max(entry, lo))
        parts.append(p)
        assert ex >= hi
        entry = ex
    assert (np.concatenate(parts) == want).all()
    got_s = _stream_all(a, hay, [1, 5, 7, 20000, 20001, 300000])
    assert (got_s == want.astype(np.int64)).all()
    # dense in matches: every 4-gram of the haystack is a keyword -> more records than the sparse path accepts -> the
    # call falls back to the walk (the filter itself is still selective: 2 % of the 4-grams)
    dense = np.concatenate([kws[i % len(kws)] for i in range(2000)]).astype(np.uint16)[:8003]
    kws2 = list(kws) + [dense[i:i + 4] for i in range(0, 8000)]
    a2 = Automaton(N.MODE_LONGEST, kws2, True)
    assert a2.info()["tile_kernel"] == 1
    want2 = Oracle(FAM_LONGEST, kws2).match(dense)
    d2 = torch.from_numpy(dense.view(np.int16)).cuda()
    o2 = torch.empty((len(want2) + 8, 3), dtype=torch.int32, device="cuda")
    n2, rc2, prof2, _ = a2.match_device(d2.data_ptr(), dense.size, True, o2.data_ptr(), o2.shape[0], profile=True)
    assert rc2 == N.OK and prof2["scan_kernel"].startswith("k_longest_walk")
    assert n2 == len(want2) and (o2[:n2].cpu().numpy() == want2).all()


@pytest.mark.parametrize("seed", range(4))
def test_large_alphabets_bucketed_classes(seed, ac_kernel):
    """More than 63 distinct keyword units (CJK, mixed scripts): the tile classes are 63 buckets, the filter a superset
    test, the K-gram looked up by its units.  Units that share a bucket, case folding, near misses."""
    rng = np.random.default_rng(900 + seed)
    alpha = np.concatenate([np.arange(0x4E00, 0x4E00 + 400), np.arange(0x0391, 0x03AA), np.arange(0x03B1, 0x03CA),
                            np.arange(ord("a"), ord("z") + 1), np.arange(ord("A"), ord("Z") + 1)]).astype(np.uint16)
    kws = [alpha[rng.integers(0, len(alpha), int(rng.integers(1 + seed % 3, 9)))] for _ in range(1500)]
    parts = []
    for _ in range(20000):
        mode = int(rng.integers(0, 4))
        k = kws[int(rng.integers(0, len(kws)))].copy()
        if mode == 0:
            parts.append(alpha[rng.integers(0, len(alpha), int(rng.integers(1, 6)))])
        elif mode == 1 and len(k) > 1:
            k[int(rng.integers(0, len(k)))] = alpha[int(rng.integers(0, len(alpha)))]  # near miss (often the same bucket)
            parts.append(k)
        else:
            parts.append(k)
    hay = np.concatenate(parts).astype(np.uint16)
    for cs in (True, False):
        a = Automaton(N.MODE_ALL, kws, cs)
        assert a.info()["n_classes"] > 64 and a.info()["filter_k"] >= 1
        want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay)
        got = a.match_host(hay, True, cap=64)
        assert got.shape == want.shape and (got == want).all(), (seed, cs)
    from oracle.oracle import FAM_SHORTEST
    assert (Automaton(N.MODE_SHORTEST, kws, False).match_host(hay, True) ==
            Oracle(FAM_SHORTEST, kws, case_sensitive=False, lower=LOWER).match(hay)).all()
    # round 4: the class table comes from LDS pages (acgpu_build.cpp 7b) -- the same records with the table in global memory
    # (builder knob no_class_pages), and with a dictionary whose units lie in all 256 pages: 64 KB of pages do not fit behind
    # the filter rows, the kernel keeps the global table
    N.set_tunable("no_class_pages", 1)
    try:
        a = Automaton(N.MODE_ALL, kws, True)
    finally:
        N.set_tunable("no_class_pages", 0)
    assert (a.match_host(hay, True) == Oracle(FAM_AC, kws, case_sensitive=True, lower=LOWER).match(hay)).all()
    spread = (np.arange(256, dtype=np.uint32) * 256 + rng.integers(1, 255, 256)).astype(np.uint16)
    spread = spread[(spread < 0xD800) | (spread > 0xDFFF)]
    kws2 = [spread[rng.integers(0, len(spread), int(rng.integers(3, 7)))] for _ in range(800)]
    hay2 = np.concatenate([kws2[int(rng.integers(0, len(kws2)))] if rng.integers(0, 2) else spread[rng.integers(0, len(spread), 3)]
                           for _ in range(8000)]).astype(np.uint16)
    a2 = Automaton(N.MODE_ALL, kws2, True)
    assert a2.info()["n_classes"] > 64 and a2.info()["filter_k"] == 3
    want2 = Oracle(FAM_AC, kws2, case_sensitive=True, lower=LOWER).match(hay2)
    got2 = a2.match_host(hay2, True)
    assert got2.shape == want2.shape and (got2 == want2).all()


def test_scratch_slice_overflow_is_redone_with_one_slice():
    """The tile kernel takes record slots from one scratch slice per workgroup.  Here every match lies in the first
    workgroup's share of the haystack and the capacity is exact, so that slice fills up and the call is redone with one
    slice -- synchronously and through begin/end; results equal the oracle's and the one-counter form's."""
    import torch
    kws = ["".join(p) for k in (2, 3, 4) for p in __import__("itertools").product("ab", repeat=k)]
    a = Automaton(N.MODE_ALL, kws, True)
    n = 1 << 22
    hay = np.full(n, ord("z"), dtype=np.uint16)
    hay[: 1 << 18] = synth.haystack(5, 1 << 18, table=synth.ALPHA_LOWER[:2])
    want = Oracle(FAM_AC, kws).match(hay)
    assert len(want) > 700000
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    cap = len(want)
    got, _ = _dev_match(a, d_hay, n, True, cap)
    assert got.shape == want.shape and (got == want).all()
    out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), n, True, out.data_ptr(), cap, stream=torch.cuda.current_stream().cuda_stream)
    assert rc == N.OK
    m, rc, _ = a.match_device_end(tk)
    assert rc == N.OK and m == len(want) and (out[:m].cpu().numpy() == want).all()
    N.set_tunable("tile_debug", 16384)  # one slot counter for the whole grid
    got1, _ = _dev_match(a, d_hay, n, True, cap)
    assert (got1 == want).all()


def test_long_keywords_branching_suffixes_and_planted_matches():
    """Long keywords (up to 30 units), keywords that are suffixes of others, keywords that share long suffixes: the
    reversed-trie walk of the tile kernel in all its shapes.  Every keyword is planted a few times, the first one right at
    the start of the buffer."""
    rng = np.random.default_rng(77)
    alpha = list(range(ord("a"), ord("h") + 1))
    for min_len, max_len in ((4, 16), (5, 30), (3, 12)):
        hay, kws = rand_case(rng, alpha, 60, max_len, 60013, min_len=min_len)
        arr = lambda t: np.array([ord(ch) for ch in t], dtype=np.uint16)
        kws = list(kws) + [kws[0][-min_len:], kws[1][1:], arr("abcdefgh" * 3), arr("h" + "abcdefgh" * 2)]
        hay = hay.copy()
        pos = 0
        for k in kws:  # plant every keyword a few times, the first at offset 0
            u = np.asarray(k, dtype=np.uint16)
            for _ in range(3):
                if pos + u.size < hay.size:
                    hay[pos:pos + u.size] = u
                pos += u.size + int(rng.integers(0, 40))
        for cs in (True, False):
            m = AhoCorasickMap(kws, _ids(len(kws)), cs)
            want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay).tolist()
            assert len(want) > 3 * len(kws) - 10
            assert m.find_all(hay).tolist() == want


def test_wholeword_scratch_slice_overflow_is_redone_with_one_slice():
    """WholeWord takes record slots from one scratch slice per workgroup as well: all words in the first workgroup's share,
    exact capacity -> that slice fills up and the call is redone with one slice."""
    import torch
    words = ["ab", "abc", "b", "cab"]
    a = Automaton(N.MODE_WHOLEWORD, words, True, word_chars=WORD)
    n = 1 << 22
    rng = np.random.default_rng(3)
    hay = np.full(n, ord(" "), dtype=np.uint16)
    toks = rng.integers(0, len(words), 1 << 16)
    text = " ".join(words[i] for i in toks)
    head = np.array([ord(c) for c in text], dtype=np.uint16)
    hay[: head.size] = head
    from oracle.oracle import FAM_WHOLEWORD
    want = Oracle(FAM_WHOLEWORD, words, word_chars=WORD).match(hay)
    assert len(want) == 1 << 16
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    got, _ = _dev_match(a, d_hay, n, True, len(want))
    assert got.shape == want.shape and (got == want).all()


def test_wholeword_async_begin_end_pipelined_overflow_and_slice_redo():
    """acgpu_match_device_begin/_end for WHOLEWORD (fold-consistent tables): three calls in flight on one stream equal the
    synchronous results; too small a capacity is reported by _end with the exact count; a scratch slice that fills up is
    redone inside _end; a fold-inconsistent table has no asynchronous form."""
    import torch
    from oracle.oracle import FAM_WHOLEWORD
    words = synth.mixed_script_words(1005, 3000)
    a = Automaton(N.MODE_WHOLEWORD, words, False, lower=LOWER, word_chars=WORD)
    hays = [synth.mixed_script_haystack(2100 + i, 150000 + 999 * i, words, swapcase_tbl=synth.swapcase_table()) for i in range(3)]
    d_hays = [torch.from_numpy(h.view(np.int16)).cuda() for h in hays]
    wants = [Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD).match(h) for h in hays]
    cap = max(len(w) for w in wants) + 8
    outs = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in hays]
    st = torch.cuda.current_stream().cuda_stream
    tickets = []
    for d, o, h in zip(d_hays, outs, hays):
        tk, rc = a.match_device_begin(d.data_ptr(), h.size, True, o.data_ptr(), cap, stream=st, profile=True)
        assert rc == N.OK
        tickets.append(tk)
    for tk, o, w in zip(tickets, outs, wants):
        n, rc, prof = a.match_device_end(tk, profile=True)
        assert rc == N.OK and n == len(w) and len(w) > 3000 and prof["scan_kernel"].startswith("k_ww_")
        assert (o[:n].cpu().numpy() == w).all()
    tk, rc = a.match_device_begin(d_hays[0].data_ptr(), hays[0].size, True, outs[0].data_ptr(), 5, stream=st)
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.E_OVERFLOW and n == len(wants[0])
    # the device-side result header, and a shard with halos through the asynchronous form
    buf = torch.zeros(4 + 3 * cap, dtype=torch.int32, device="cuda")
    ml = a.info()["max_keyword_len"]
    lo, hi = 40000, 90001
    base = (lo - 1) // 8 * 8
    sub = d_hays[1][base:hi + ml + 1].clone()
    tk, rc = a.match_device_begin(sub.data_ptr(), sub.numel(), True, buf.data_ptr() + 16, cap, own=(lo - base, hi - base),
                                  text_begin=False, text_end=False, stream=st, d_result=buf.data_ptr())
    assert rc == N.OK
    n, rc, _ = a.match_device_end(tk)
    w = wants[1]
    w = w[(w[:, 0] >= lo) & (w[:, 0] < hi)].copy()
    w[:, :2] -= base
    host = buf.cpu().numpy()
    assert rc == N.OK and n == len(w) and int(host[0]) == n and int(host[2]) == 0
    assert (host[4:4 + 3 * n].reshape(n, 3) == w).all()
    # all words in the first workgroup's share, exact capacity: the slice overflow is redone with one slice inside _end
    small = ["ab", "abc", "b", "cab"]
    a2 = Automaton(N.MODE_WHOLEWORD, small, True, word_chars=WORD)
    n2 = 1 << 22
    rng = np.random.default_rng(3)
    hay = np.full(n2, ord(" "), dtype=np.uint16)
    head = np.array([ord(c) for c in " ".join(small[i] for i in rng.integers(0, len(small), 1 << 16))], dtype=np.uint16)
    hay[: head.size] = head
    want = Oracle(FAM_WHOLEWORD, small, word_chars=WORD).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    out = torch.empty((len(want), 3), dtype=torch.int32, device="cuda")
    tk, rc = a2.match_device_begin(d_hay.data_ptr(), n2, True, out.data_ptr(), len(want), stream=st)
    assert rc == N.OK
    m, rc, _ = a2.match_device_end(tk)
    assert rc == N.OK and m == len(want) and (out[:m].cpu().numpy() == want).all()
    # fold-inconsistent custom table: sequential kernel, synchronous call only
    wc = np.zeros(65536, np.uint8)
    for ch in "ABx":
        wc[ord(ch)] = 1
    a3 = Automaton(N.MODE_WHOLEWORD, ["A", "AB", "x"], False, lower=LOWER, word_chars=wc)
    assert a3.info()["fold_consistent"] == 0
    tk, rc = a3.match_device_begin(d_hay.data_ptr(), 4096, True, out.data_ptr(), 16, stream=st)  # (runs inside _begin)
    assert rc == N.OK and a3.match_device_end(tk)[1] in (N.OK, N.E_OVERFLOW)


def test_chain_marking_in_one_pass_equals_pointer_doubling():
    """Shortest, the sparse form of Longest and WholeWordLongest select a chain k0, nxt[k0], ... over their candidates.  From
    4 M candidates on the chain is marked in one pass by the Longest chain kernels, below that by pointer doubling; the test
    hook (tile_debug bit 4194304) takes the one pass from 64 candidates on, bit 2097152 forces the doubling: same records."""
    from oracle.oracle import FAM_SHORTEST, FAM_WWLONGEST
    kws = synth.random_keywords(21, 400, 2, 9, table=synth.ALPHA_LOWER[:6])
    hay = synth.haystack(77, 400000, table=synth.ALPHA_LOWER[:6])
    cases = [(N.MODE_SHORTEST, FAM_SHORTEST, kws, hay, {}), (N.MODE_LONGEST, FAM_LONGEST, kws, hay, {})]
    wkws, whay = _wwl_case(900, 300000)
    cases.append((N.MODE_WWLONGEST, FAM_WWLONGEST, wkws, whay, dict(case_sensitive=False, lower=LOWER, word_chars=WORD)))
    for mode, fam, k, h, okw in cases:
        want = Oracle(fam, k, **okw).match(h)
        assert len(want) > 1000
        for bits in (4194304, 2097152, 0):
            N.set_tunable("tile_debug", bits)
            a = Automaton(mode, k, okw.get("case_sensitive", True), word_chars=okw.get("word_chars"))
            got = a.match_host(h, True)
            assert got.shape == want.shape and (got == want).all(), (mode, bits)
        N.set_tunable("tile_debug", 0)


def test_wholeword_fallback_hash_seeds_end_to_end():
    """The builder takes another hash seed when the two-choice table cannot hold a dictionary (three keywords agreeing in
    both hashes).  That never happens by chance, so the test hook "ww_first_seed" starts the builder at a later seed: the
    kernel must hash with the seed the tables were built with (acgpu debug hook: seed != default), results unchanged."""
    from oracle.oracle import FAM_WHOLEWORD
    words = synth.mixed_script_words(1005, 4000)
    hay = synth.mixed_script_haystack(2051, 200000, words, swapcase_tbl=synth.swapcase_table())
    want = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay)
    for first in (0, 3, 7):
        N.set_tunable("ww_first_seed", first)
        a = Automaton(N.MODE_WHOLEWORD, words, False, lower=LOWER, word_chars=WORD)
        seed = ctypes.c_uint32(0)
        N.check(N.lib().acgpu_debug_wordhash(a.handle, None, None, None, None, None, None, None, ctypes.byref(seed)), "seed")
        assert (seed.value == 0x811C9DC5) == (first == 0)
        got = a.match_host(hay, True)
        assert len(want) > 5000 and got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("min_len", [2, 4, 6])
def test_case_insensitive_folded_range_classes(min_len):
    """Case-insensitive dictionaries over a short range of letters take the packed filter with FOLDED range classes: the
    range, its partner range of the other case, and the class table for tiles that hold units beyond the low zone -- where
    U+0130 (folds to i) and U+212A (folds to k) live.  Haystacks with and without such units, keywords given in mixed case."""
    import torch
    rng = np.random.default_rng(500 + min_len)
    low = list(range(ord("a"), ord("k") + 1))
    for hay_alpha in (low + [c - 32 for c in low], low + [c - 32 for c in low] + [0x0130, 0x212A, 0x00E9, 0x4E2D, ord(" ")]):
        hay, kws = rand_case(rng, low, 40, min_len + 6, 150001, min_len=min_len)
        hay = np.asarray(hay_alpha, dtype=np.uint16)[rng.integers(0, len(hay_alpha), hay.size)]
        kws = [np.where(rng.integers(0, 2, k.size) == 1, k - 32, k).astype(np.uint16) for k in kws]  # mixed-case keywords
        pos = 0
        for k in kws:  # plant them, some through the fold exceptions
            u = k.copy()
            if 0x0130 in hay_alpha:
                u[(u | 32) == ord("i")] = 0x0130 if pos % 2 else ord("I")
                u[(u | 32) == ord("k")] = 0x212A if pos % 3 else ord("k")
            if pos + u.size < hay.size:
                hay[pos:pos + u.size] = u
            pos += u.size + 50
        m = AhoCorasickMap(kws, _ids(len(kws)), False)
        want = Oracle(FAM_AC, kws, case_sensitive=False, lower=LOWER).match(hay)
        d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
        got, prof = _dev_match(m.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
        args = prof["scan_kernel"].split("<")[1].rstrip(">").split(", ")
        assert args[1] == "false" and len(args) >= 6 and args[5] == "true", prof["scan_kernel"]  # folded range + packed filter
        assert len(want) >= 40 and got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("min_len", [2, 3, 4, 6])
def test_case_sensitive_mixed_case_dictionary_merged_ranges(min_len):
    """Case-sensitive dictionaries whose units lie in two stretches of at most 31 code points (keywords in mixed case: A-Z and
    a-z, 33..64 classes) take the packed two-range filter with MERGED classes ('T' and 't' share one) and verify by the units
    themselves: a haystack full of case variants of the keywords -- which pass both filter levels -- must report exactly the
    exact-case occurrences, like the 8-byte-row scalar form (tunable no_merged_ranges) and the oracle."""
    import torch
    rng = np.random.default_rng(900 + min_len)
    low = list(range(ord("b"), ord("z") + 1))       # the two stretches are not aligned letter for letter: B..X and b..z
    up = list(range(ord("B"), ord("X") + 1))
    hay, kws = rand_case(rng, low[:12], 300, min_len + 7, 400001, min_len=min_len)
    kws = [np.where((rng.integers(0, 2, k.size) == 1) & (k - 32 <= up[-1]), k - 32, k).astype(np.uint16) for k in kws]
    kws.append(np.array([ord(c) for c in "zzXBb"[:max(min_len, 2)] + "zX"], dtype=np.uint16))  # both ends of both stretches
    hay_alpha = np.asarray(low[:12] + up[:12] + [ord("z"), ord("X"), ord(" "), 0x00E9, 0x4E2D, ord("A"), ord("a"), ord("Y"), 0xFFFF], dtype=np.uint16)
    hay = hay_alpha[rng.integers(0, len(hay_alpha), hay.size)]
    pos = 0
    for i, k in enumerate(kws * 3):  # plant keywords: exact, and with the case of one unit flipped (passes the filters, must fail)
        u = k.copy()
        if i % 2:
            j = int(rng.integers(0, u.size))
            u[j] = u[j] ^ 32
        if pos + u.size < hay.size:
            hay[pos:pos + u.size] = u
        pos += u.size + 37
    want = Oracle(FAM_AC, kws, case_sensitive=True).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    m = AhoCorasickMap(kws, _ids(len(kws)), True)
    assert m.automaton.info()["n_classes"] > 32
    got, prof = _dev_match(m.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
    args = prof["scan_kernel"].split("<")[1].rstrip(">").split(", ")
    assert args[1] == "false" and args[4] == "true" and args[5] == "true", prof["scan_kernel"]  # merged classes, by units, packed
    assert len(want) >= 300 and got.shape == want.shape and (got == want).all()
    # a tail of fewer than 8 units, and shards
    for n in (hay.size - 3, 4099):
        w2 = Oracle(FAM_AC, kws, case_sensitive=True).match(hay[:n])
        g2, _ = _dev_match(m.automaton, d_hay, n, True, len(w2) + 8, profile=True)
        assert g2.shape == w2.shape and (g2 == w2).all()
    N.lib().acgpu_set_tunable(b"no_merged_ranges", 1)
    try:
        m8 = AhoCorasickMap(kws, _ids(len(kws)), True)
        got8, prof8 = _dev_match(m8.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
    finally:
        N.lib().acgpu_set_tunable(b"no_merged_ranges", 0)
    assert prof8["scan_kernel"] != prof["scan_kernel"] and (got8 == want).all()


@pytest.mark.parametrize("shape", ["cs_phrases", "ci_phrases", "cs_mixed_phrases", "ci_mixed_phrases", "ci_greek_latin"])
def test_merged_stretches_phrases_and_case_insensitive(shape):
    """Dictionaries whose (folded) units fall into up to four ranges of at most 31 code points -- phrases with spaces, digits and
    hyphens, in lower or mixed case, case-sensitive or not; two scripts -- take the packed filter with merged classes (three or
    four ranges: the NR4 form) and verify by units.  The haystack holds case variants, the fold exceptions U+0130 / U+212A (beyond
    the low zone: class table), units between and around the ranges, and planted keywords."""
    import torch
    rng = np.random.default_rng(hash(shape) % 1000)
    cs = shape.startswith("cs")
    letters = list(range(ord("a"), ord("z") + 1))
    extra = [32, 45, 48, 49, 50, 57]
    if shape == "ci_greek_latin":
        # (Latin without i and k: U+0130 and U+212A fold to them from INSIDE / beyond a low zone that reaches up to the Greek
        # range, and a fold exception inside the low zone is something the arithmetic cannot express: such a dictionary keeps
        # the class-table form)
        # (and Greek without theta, to which U+03F4 folds from inside the low zone)
        letters = [ord(c) for c in "abcdefghjlmn"] + [c for c in range(0x03B1, 0x03B1 + 13) if c != 0x03B8]
        extra = []
    kws = []
    for _ in range(400):
        ln = int(rng.integers(3, 11))
        k = np.array(rng.choice(letters, ln), dtype=np.uint16)
        if extra and ln >= 5:
            k[int(rng.integers(1, ln - 1))] = int(rng.choice(extra))
        if "mixed" in shape or shape == "ci_greek_latin":
            m = (rng.integers(0, 2, ln) == 1) & np.isin(k, letters)
            k = np.where(m, k - 32, k).astype(np.uint16)
        kws.append(k)
    alpha = np.array(letters[:14] + [c - 32 for c in letters[:14]] + extra + [58, 64, 91, 96, 123, 0x0130, 0x212A, 0x00E9, 0x4E2D, 0xFFFF, 0],
                     dtype=np.uint16)
    hay = alpha[rng.integers(0, len(alpha), 300001)]
    pos = 0
    for i, k in enumerate(kws * 3):
        u = k.copy()
        if i % 3 == 1:  # flip the case of one letter: a match only when case-insensitive
            j = int(rng.integers(0, u.size))
            if u[j] in letters or (u[j] + 32) in letters:
                u[j] = u[j] ^ 32
        elif i % 3 == 2 and not cs:  # through the fold exceptions
            u[(u | 32) == ord("i")] = 0x0130
            u[(u | 32) == ord("k")] = 0x212A
        if pos + u.size < hay.size:
            hay[pos:pos + u.size] = u
        pos += u.size + 29
    want = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    m = AhoCorasickMap(kws, _ids(len(kws)), cs)
    got, prof = _dev_match(m.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
    args = prof["scan_kernel"].split("<")[1].rstrip(">").split(", ")
    assert args[4] == "true" and args[5] == "true", prof["scan_kernel"]  # verification by units, packed filter
    assert (args[7] == "true") == (shape != "cs_phrases"), prof["scan_kernel"]  # three or four ranges but for lower case + space/digits
    assert len(want) >= 400 and got.shape == want.shape and (got == want).all()
    w2 = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER).match(hay[:70003])
    g2, _ = _dev_match(m.automaton, d_hay, 70003, True, len(w2) + 8, profile=True)
    assert g2.shape == w2.shape and (g2 == w2).all()


@pytest.mark.parametrize("shape", ["range", "ci", "lut_wide", "only_short", "k3", "merged_cs", "merged_ci", "buckets"])
def test_short_keywords_beside_the_k_gram_filter(shape):
    """Keywords of fewer than K units no longer pull the filter's K down to their length: they set wild-card bits, pass the
    second level through the 'other'-led row and are reported from the table of the last K-1 classes -- after the longer
    keywords that end at the same position (longest first), at the very start of the text, in shards and with several short
    and long keywords ending together."""
    import torch
    rng = np.random.default_rng(hash(shape) % 997)
    cs = shape not in ("ci", "merged_ci")
    if shape == "lut_wide":  # more than 32 classes that are no range: the scalar filter with 8-byte rows
        alpha = [ord(c) for c in "abcdefghijklmnopqrstuvwxyzABCDEFGHIJ 0123"]
        N.set_tunable("no_merged_ranges", 1)
    elif shape in ("merged_cs", "merged_ci"):  # mixed case / phrases: merged stretches, verification by units (ks_keys)
        alpha = [ord(c) for c in "abcdefghABCDEFGH 019"]
    elif shape == "buckets":  # more than 63 distinct units: bucketed classes, K <= 3
        alpha = list(range(0x4E00, 0x4E00 + 300, 3))  # (100 units over 300 code points: more than four stretches of 31)
    else:
        alpha = list(range(ord("a"), ord("a") + (8 if shape != "k3" else 26)))
    kws = [np.array(rng.choice(alpha, int(rng.integers(4, 10))), dtype=np.uint16) for _ in range(300 if shape != "only_short" else 0)]
    shorts = [[alpha[0]], [alpha[1], alpha[0]], [alpha[0], alpha[1]], [alpha[2], alpha[0], alpha[1]], [alpha[3]] * 3, [alpha[3]] * 2,
              [alpha[4], alpha[5]]]
    if shape == "k3":
        shorts = [s for s in shorts if len(s) > 1] + [[alpha[9], alpha[7]]]
        kws = [k[:3 + int(rng.integers(0, 6))] for k in kws]  # longest keywords: K stays 4; some of 3
    kws += [np.array(s, dtype=np.uint16) for s in shorts]
    if kws and shape != "only_short":
        kws.append(np.concatenate([kws[0], np.array(shorts[3], dtype=np.uint16)]))  # a long keyword that ends with a short one
    if shape == "ci":
        kws = [np.where(rng.integers(0, 2, k.size) == 1, k - 32, k).astype(np.uint16) for k in kws]
    if shape == "buckets":
        kws = [k for k in kws if k.size != 3 or rng.integers(0, 2)]
    hay_alpha = np.array(alpha[:10] + ([c - 32 for c in alpha[:6]] if shape == "ci" else []) + [ord("!")], dtype=np.uint16)
    hay = hay_alpha[rng.integers(0, len(hay_alpha), 200003)]
    hay[:3] = np.array(shorts[3], dtype=np.uint16)  # short keywords at the very start of the text
    for i, k in enumerate(kws * 2):
        p0 = 50 + i * 41
        if p0 + k.size < hay.size:
            hay[p0:p0 + k.size] = k
    orc = Oracle(FAM_AC, kws, case_sensitive=cs, lower=LOWER)
    want = orc.match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    try:
        m = AhoCorasickMap(kws, _ids(len(kws)), cs)
    finally:
        N.set_tunable("no_merged_ranges", 0)
    info = m.automaton.info()
    assert info["filter_k"] > info["min_keyword_len"] and info["filter_k"] == (3 if shape in ("lut_wide", "buckets") else min(4, info["max_keyword_len"])), info
    got, prof = _dev_match(m.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
    assert len(want) > 1000 and got.shape == want.shape and (got == want).all(), prof["scan_kernel"]
    assert got[0, 1] <= 3 and (shape == "buckets" or 0 in got[:8, 0].tolist())  # matches at the very start of the text
    if shape in ("merged_cs", "merged_ci", "buckets"):
        assert prof["scan_kernel"].split(", ")[4].startswith("true"), prof["scan_kernel"]  # verification by units
    for n in (1, 2, 3, 5, hay.size - 5):  # texts shorter than K; a tail of fewer than 8 units
        w2 = orc.match(hay[:n])
        g2, _ = _dev_match(m.automaton, d_hay, n, True, len(w2) + 8, profile=True)
        assert g2.shape == w2.shape and (g2 == w2).all(), n
    for knobs in ({"force_kernel": 3}, {"tile_debug": 2048}, {"tile_debug": 1024}):  # split form, no second level, scalar filter
        if shape.startswith("merged") and knobs.get("tile_debug") == 1024:
            continue  # (merged classes exist in the packed filter only)
        for k, v in knobs.items():
            N.set_tunable(k, v)
        try:
            g3, p3 = _dev_match(m.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
        finally:
            for k in knobs:
                N.set_tunable(k, 0)
        assert g3.shape == want.shape and (g3 == want).all(), (knobs, p3["scan_kernel"])
    # shards: a rank owns the matches whose last unit it owns
    cuts = [0, 66671, 66673, hay.size]
    parts = [_dev_match(m.automaton, d_hay, hay.size, True, len(want) + 8, own=(cuts[r], cuts[r + 1]))[0] for r in range(3)]
    assert (np.concatenate(parts) == want).all()
    N.set_tunable("no_short_keywords", 1)
    try:
        m0 = AhoCorasickMap(kws, _ids(len(kws)), cs)
        assert m0.automaton.info()["filter_k"] == info["min_keyword_len"]
        g0, _ = _dev_match(m0.automaton, d_hay, hay.size, True, len(want) + 8, profile=True)
    finally:
        N.set_tunable("no_short_keywords", 0)
    assert (g0 == want).all()


# ---- the pipelined single-GPU driver (bench.py's N=1 path), device-side result header, stream rule ------------------

def test_pipelined_steps_survive_overflow_with_a_changing_haystack():
    """ShardedMatcher(overlap=True), world 1 (bench.py's N=1 driver): step k+1 is enqueued before step k is collected.
    A tiny record capacity makes steps overflow; phase 1 changes the haystack between steps (load() completes the step
    in flight first -- a redo needs its text), phase 2 keeps it and lets step k overflow while step k+1 is in flight in
    its own, equally small buffer.  Every step's published records must be the oracle's for ITS haystack."""
    from ahocorasick_amd.dist import ShardedMatcher
    kws = synth.random_keywords(11, 300, 2, 9)
    a = Automaton(N.MODE_ALL, kws, True)
    orc = Oracle(FAM_AC, kws)
    n = 200000
    hays = [synth.haystack(300 + i, n, table=synth.ALPHA_LOWER[:6 + 5 * (i % 3)]) for i in range(6)]  # very different densities
    wants = [orc.match(h) for h in hays]
    assert len({len(w) for w in wants}) > 3
    m = ShardedMatcher(a, n, with_ids=True, cap=8, overlap=True)
    results = []
    for h in hays:
        m.load(h)
        r = m.step()
        if r is not None:  # the PREVIOUS step's result: gathered/counts describe it now
            results.append((r, m.global_records().cpu().numpy()))
    results.append((m.finish(), m.global_records().cpu().numpy()))
    assert len(results) == len(hays) and m.redone_steps >= 1
    for (r, got), want in zip(results, wants):
        assert r["n_total"] == len(want) and got.shape == want.shape and (got == want).all()
    # phase 2: one haystack, pipelined for real; the first two steps both start with 8 record slots
    m = ShardedMatcher(a, n, with_ids=True, cap=8, overlap=True, adaptive=False)
    m.load(hays[1])
    outs = [m.step() for _ in range(4)] + [m.finish()]
    assert outs[0] is None and all(o["n_total"] == len(wants[1]) for o in outs[1:])
    assert m.redone_steps == 2 and (m.global_records().cpu().numpy() == wants[1]).all()


def test_device_result_header_and_stream_rule():
    import torch
    kws = synth.random_keywords(11, 300, 2, 9)
    hay = synth.haystack(91, 150000)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for mode, fam in ((N.MODE_ALL, FAM_AC), (N.MODE_LONGEST, 1), (N.MODE_SHORTEST, 3)):
        a = Automaton(mode, kws, True)
        want = Oracle(fam, kws).match(hay)
        buf = torch.zeros(4 + 3 * (len(want) + 8), dtype=torch.int32, device="cuda")
        n, rc, _, _ = a.match_device(d_hay.data_ptr(), hay.size, True, buf.data_ptr() + 16, len(want) + 8, stream=st,
                                     d_result=buf.data_ptr())
        torch.cuda.synchronize()
        hdr = buf[:4].cpu().numpy()
        assert rc == N.OK and n == len(want) and hdr.tolist() == [len(want), 0, 0, 0], (mode, hdr)
        assert (buf[4:4 + 3 * n].view(n, 3).cpu().numpy() == want).all()
        # too small a capacity: the header still carries the exact count
        n2, rc2, _, _ = a.match_device(d_hay.data_ptr(), hay.size, True, buf.data_ptr() + 16, 5, stream=st, d_result=buf.data_ptr())
        torch.cuda.synchronize()
        assert rc2 == N.E_OVERFLOW and n2 == len(want) and buf[:2].cpu().numpy().tolist() == [len(want), 0]
        assert a.match_device(d_hay.data_ptr(), hay.size, True, buf.data_ptr() + 16, 5, stream=st, d_result=buf.data_ptr() + 4)[1] == N.E_INVALID
    # asynchronous form: the header arrives in stream order; a call on ANOTHER stream while the ticket is in flight is refused
    a = Automaton(N.MODE_ALL, kws, True)
    want = Oracle(FAM_AC, kws).match(hay)
    buf = torch.zeros(4 + 3 * (len(want) + 8), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, buf.data_ptr() + 16, len(want) + 8, stream=st, d_result=buf.data_ptr())
    assert rc == N.OK
    other = torch.cuda.Stream()
    out2 = torch.empty((len(want) + 8, 3), dtype=torch.int32, device="cuda")
    assert a.match_device(d_hay.data_ptr(), hay.size, True, out2.data_ptr(), len(want) + 8, stream=other.cuda_stream)[1] == N.E_INVALID
    assert a.match_device_begin(d_hay.data_ptr(), hay.size, True, out2.data_ptr(), len(want) + 8, stream=other.cuda_stream)[1] == N.E_INVALID
    n3, rc3, _, _ = a.match_device(d_hay.data_ptr(), hay.size, True, out2.data_ptr(), len(want) + 8, stream=st)  # same stream: fine
    assert rc3 == N.OK and n3 == len(want)
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.OK and n == len(want) and buf[:4].cpu().numpy().tolist() == [len(want), 0, 0, 0]
    assert (buf[4:4 + 3 * n].view(n, 3).cpu().numpy() == want).all() and (out2[:n3].cpu().numpy() == want).all()
    # with nothing in flight any stream may be used again
    assert a.match_device(d_hay.data_ptr(), hay.size, True, out2.data_ptr(), len(want) + 8, stream=other.cuda_stream)[1] == N.OK
    other.synchronize()
    assert a.match_device_end(tk)[1] == N.E_INVALID  # a ticket is collected once


def test_stream_probe_reports_a_plausible_read_bandwidth():
    import torch
    d = torch.zeros(1 << 27, dtype=torch.int16, device="cuda")  # 256 MiB
    ms = ctypes.c_float(0)
    for pattern in (0, 1):  # the tile kernels' own pattern, the fastest pure read
        N.check(N.lib().acgpu_stream_probe(d.data_ptr(), d.numel() * 2, None, 5, pattern, ctypes.byref(ms)), "probe")
        gbps = d.numel() * 2 / (ms.value * 1e-3) / 1e9
        assert 500 < gbps < 8000, gbps  # below the 8 TB/s spec peak, far above anything a host path reaches
    assert N.lib().acgpu_stream_probe(d.data_ptr(), 1 << 21, None, 5, 2, ctypes.byref(ms)) == N.E_INVALID
    assert N.lib().acgpu_stream_probe(d.data_ptr() + 2, 1 << 21, None, 5, 0, ctypes.byref(ms)) == N.E_INVALID


# ---- the one-launch form of acgpu_match_u16 for short haystacks (csrc/acgpu_small.hip) ---------------------------------------

@pytest.mark.parametrize("family", ["ac", "ac_ci", "shortest", "longest", "wholeword", "wholeword_ci"])
def test_short_haystacks_one_launch_equals_oracle_and_general_path(family):
    """A haystack of up to 4096 units is matched by ONE launch of one workgroup that reads it from, and writes the records to,
    host-mapped pinned memory (no copies, no second launch): every family it serves against the oracle and against the general
    path (tunable tile_debug bit 2^41), Set and Map records, lengths 1 .. 4096, the capacity protocol, and the hand-over to
    the general path when the occurrences do not fit its LDS lists."""
    from oracle.oracle import FAM_SHORTEST
    rng = np.random.default_rng({"ac": 1, "ac_ci": 2, "shortest": 3, "longest": 4, "wholeword": 5, "wholeword_ci": 6}[family])
    alpha = np.array([ord(c) for c in "abcAB d,"] + [0x00E9, 0x00C9], dtype=np.uint16)
    for trial in range(12):
        n_kw = int(rng.integers(1, 400))
        letters = alpha[:5] if family.startswith("wholeword") else alpha[:int(rng.integers(2, 6))]
        kws = [letters[rng.integers(0, len(letters), int(rng.integers(1, 9)))] for _ in range(n_kw)]
        cs = not family.endswith("_ci")
        if family.startswith("ac"):
            auto, orc = Automaton(N.MODE_ALL, kws, cs), Oracle(FAM_AC, kws, case_sensitive=cs, lower=None if cs else LOWER)
        elif family == "shortest":
            auto, orc = Automaton(N.MODE_SHORTEST, kws, True), Oracle(FAM_SHORTEST, kws)
        elif family == "longest":
            auto, orc = Automaton(N.MODE_LONGEST, kws, True), Oracle(FAM_LONGEST, kws)
        else:
            auto = Automaton(N.MODE_WHOLEWORD, kws, cs, word_chars=WORD)
            orc = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=None if cs else LOWER, word_chars=WORD)
        for n in (1, 2, 7, 64, 333, 1024, 1025, 4095, 4096):
            hay = alpha[rng.integers(0, len(alpha), n)]
            want = orc.match(hay)
            for ids in (True, False):
                w = want if ids else want[:, :2]
                N.set_tunable("tile_debug", 0)
                got = auto.match_host(hay, ids, cap=max(len(w), 1))
                assert got.shape == w.shape and (got == w).all(), (family, trial, n, ids)
                N.set_tunable("tile_debug", 1 << 41)
                gen = auto.match_host(hay, ids, cap=max(len(w), 1))
                assert gen.shape == w.shape and (gen == w).all(), (family, trial, n, ids, "general path")
            N.set_tunable("tile_debug", 0)
            if len(want) > 1:  # capacity too small: ACGPU_E_OVERFLOW with the count, the retry delivers
                got = auto.match_host(hay, True, cap=len(want) - 1)
                assert got.shape == want.shape and (got == want).all()
    # more occurrences than the kernel's lists hold: a^1 .. a^8 on 4096 a's -> the general path takes the call
    if family == "ac":
        kws = [np.full(k, ord("a"), np.uint16) for k in range(1, 9)]
        hay = np.full(4096, ord("a"), np.uint16)
        want = Oracle(FAM_AC, kws).match(hay)
        assert len(want) > 30000
        got = Automaton(N.MODE_ALL, kws, True).match_host(hay, True)
        assert got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("shape", ["lower", "lower_with_short", "case_insensitive", "mixed_case"])
def test_large_dictionaries_second_level_in_global_memory(shape):
    """Dictionaries that saturate the 22.5 KB second-level filter in LDS (60 k keywords and more) take the same Bloom structure
    at 2 MB in global memory (k_ac_tile<..., BIG>): a wave lists a tile's first-level candidates (up to 704) and tests them four
    batches at a time; tiles with more candidates, and tiles whose survivors do not fit the queue, take the older ways.  Against
    the oracle, with planted keywords, and against the LDS form (builder tunable no_big_l2)."""
    import torch
    rng = np.random.default_rng({"lower": 1, "lower_with_short": 2, "case_insensitive": 3, "mixed_case": 4}[shape])
    kws = synth.random_keywords(70, 70000, 4, 12)
    cs = True
    if shape == "lower_with_short":
        kws = list(kws) + [utf16("qz"), utf16("xjq"), utf16("k")]
    if shape == "case_insensitive":
        cs = False
    if shape == "mixed_case":
        kws = [np.where(rng.integers(0, 3, len(k)) == 0, k - 32, k).astype(np.uint16) for k in kws]
    n = (1 << 22) + 12345
    hay = synth.haystack(71, n)
    if shape in ("case_insensitive", "mixed_case"):
        hay = np.where(rng.integers(0, 3, n) == 0, hay - 32, hay).astype(np.uint16)
    for _ in range(3000):  # whole keywords in the text: true candidates for the verification; some stretches dense in them
        k = kws[int(rng.integers(0, len(kws)))]
        p = int(rng.integers(0, n - 16))
        hay[p:p + len(k)] = k
    at = 100000
    for i in range(400):  # a tile with more than 128 true candidates: the survivors do not fit the queue
        k = kws[i]
        hay[at:at + len(k)] = k
        at += len(k)
    orc = Oracle(FAM_AC, kws, case_sensitive=cs, lower=None if cs else LOWER)
    want = oracle_parallel(orc, hay, "ac", 12, cap_per_unit=0.3)
    a = Automaton(N.MODE_ALL, kws, cs)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    N.set_tunable("all_form", 1)  # (the forms of the tile kernel are the subject: not k_ac_states, whatever a call finds -- "k" is a keyword)
    got, prof = _dev_match(a, d_hay, n, True, len(want) + 16, profile=True)
    assert prof["scan_kernel"].startswith("k_ac_tile<4, ") and len(prof["scan_kernel"]) == 63, prof["scan_kernel"]  # (the BIG form: ten arguments)
    assert got.shape == want.shape and (got == want).all()
    got_s, _ = _dev_match(a, d_hay, n, False, len(want) + 16, own=(4096, n - 77), text_begin=True, text_end=True)  # a shard, Set records
    w2 = want[(want[:, 1] - 1 >= 4096) & (want[:, 1] - 1 < n - 77)][:, :2]
    assert got_s.shape == w2.shape and (got_s == w2).all()
    N.set_tunable("tile_debug", 1 << 30)  # the LDS form of the second level on the same tables
    got_l, prof_l = _dev_match(a, d_hay, n, True, len(want) + 16, profile=True)
    assert prof_l["scan_kernel"] != prof["scan_kernel"] and (got_l == want).all()
    N.set_tunable("tile_debug", 0)
    N.set_tunable("no_big_l2", 1)
    try:
        b = Automaton(N.MODE_ALL, kws, cs)
    finally:
        N.set_tunable("no_big_l2", 0)
    got_b, prof_b = _dev_match(b, d_hay, n, True, len(want) + 16, profile=True)
    assert prof_b["scan_kernel"] == prof_l["scan_kernel"] and (got_b == want).all()


def test_balanced_regions_and_reserved_cus_on_long_shards():
    """Long shards: the region size is chosen so that (regions per wave) x (region units) covers the shard with the least
    slack, also when some CUs are kept free for a collective's kernels (tunable reserve_cus) -- odd shard lengths, a shard that
    starts off a 16-byte boundary, both tile kernels, against the oracle."""
    import torch
    n = (1 << 27) + 54321
    kws = synth.config_keywords("C2")
    hay = synth.haystack(2002, n)
    want = oracle_parallel(Oracle(FAM_AC, kws), hay, "ac", 12, cap_per_unit=0.02)
    a = Automaton(N.MODE_ALL, kws, True)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    for reserve, own in ((0, None), (24, None), (8, (1003, n - 77))):
        N.set_tunable("reserve_cus", reserve)
        try:
            kw = {} if own is None else dict(own=own)
            got, prof = _dev_match(a, d_hay, n, True, len(want) + 16, profile=True, **kw)
        finally:
            N.set_tunable("reserve_cus", 0)
        w = want if own is None else want[(want[:, 1] - 1 >= own[0]) & (want[:, 1] - 1 < own[1])]
        assert got.shape == w.shape and (got == w).all(), (reserve, own)
    words = synth.mixed_script_words(1005, 20000)
    d2 = torch.empty(n, dtype=torch.int16, device="cuda")
    synth.token_stream_on_device(d2.data_ptr(), n, 2005, words, synth.swapcase_table())  # (config 5's token stream, generated in place)
    hay2 = d2.cpu().numpy().view(np.uint16)
    want2 = oracle_parallel(Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD), hay2, "wholeword", 12, cap_per_unit=0.2)
    ww = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD)
    for reserve in (0, 24):
        N.set_tunable("reserve_cus", reserve)
        try:
            got2, _ = _dev_match(ww, d2, n, True, len(want2) + 16)
        finally:
            N.set_tunable("reserve_cus", 0)
        assert got2.shape == want2.shape and (got2 == want2).all(), reserve
