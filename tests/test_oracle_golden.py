"""Pins oracle/ac_oracle.c (the CPU restatement of the reference) against
 (1) the reference's deterministic test scenarios (tests/golden/reference_fixtures.json),
 (2) the exact push/flush sequences of T/MatchQueueTest.java:9-57,
 (3) the brute-force formulas of the reference tests on seeded random inputs
     (the reference's own random tests are unseeded: T/Generator.java:61-76)."""
import numpy as np
import pytest

from oracle import brute
from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, IllegalArgumentException, MatchQueue, Oracle, trim
from tests.helpers import LOWER, WORD, fixture_inputs, rand_case


def _as_list(a):
    return [list(map(int, r)) for r in a]


def test_fixtures_all_families(fixtures):
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        assert _as_list(Oracle(FAM_AC, kws).match(hay)) == fx["AC"], fx["name"]
        assert _as_list(Oracle(FAM_LONGEST, kws).match(hay)) == fx["L"], fx["name"]
        if fx["WW"] == "IllegalArgumentException":
            with pytest.raises(IllegalArgumentException):
                Oracle(FAM_WHOLEWORD, kws, word_chars=WORD)
        else:
            assert _as_list(Oracle(FAM_WHOLEWORD, kws, word_chars=WORD).match(hay)) == fx["WW"], fx["name"]


def test_appendix_b_hand_pins():
    # SURVEY.md Appendix B (independent brute-force script), typed by hand.
    ac = Oracle(FAM_AC, ["bc", "cc", "bcc", "ccddee", "ccddeee", "d"]).match("abbccddeef")[:, :2]
    assert _as_list(ac) == [[2, 4], [2, 5], [3, 5], [5, 6], [6, 7], [3, 9]]
    lm = Oracle(FAM_LONGEST, ["a", "aa", "aaa", "aaaa"]).match(" aaaaaaa aaababababaabaa ")[:, :2]
    assert _as_list(lm) == [[1, 5], [5, 8], [9, 12], [13, 14], [15, 16], [17, 18], [19, 21], [22, 24]]
    ww = Oracle(FAM_WHOLEWORD, ["la", "late", "eve", "evening"], word_chars=WORD).match("late evening")[:, :2]
    assert _as_list(ww) == [[0, 4], [5, 12]]
    assert len(Oracle(FAM_AC, ["a" * k for k in range(1, 101)]).match("a" * 100)) == 5050


def test_readme_worked_examples_hand_pins():
    # what R/README.md states in prose, typed by hand (the generated fixtures hold the same cases)
    from oracle.oracle import FAM_SHORTEST, FAM_WWLONGEST
    # "ShortestMatchSet/Map": a1b2c3d4 with 2, b2, 2c3d4 -> only b2; with b, 2, b2 -> both b and 2
    assert _as_list(Oracle(FAM_SHORTEST, ["2", "b2", "2c3d4"]).match("a1b2c3d4")[:, :2]) == [[2, 4]]
    assert _as_list(Oracle(FAM_SHORTEST, ["b", "2", "b2"]).match("a1b2c3d4")[:, :2]) == [[2, 3], [3, 4]]
    # "LongestMatchSet/Map": b, b2, 2c3d4 -> only b2
    assert _as_list(Oracle(FAM_LONGEST, ["b", "b2", "2c3d4"]).match("a1b2c3d4")[:, :2]) == [[2, 4]]
    # "WholeWordLongestMatchSet/Map": as if -> as if; ax if -> if; as of -> as
    wwl = Oracle(FAM_WWLONGEST, ["as if", "as", "if"], word_chars=WORD)
    assert _as_list(wwl.match("as if")[:, :2]) == [[0, 5]]
    assert _as_list(wwl.match("ax if")[:, :2]) == [[3, 5]]
    assert _as_list(wwl.match("as of")[:, :2]) == [[0, 2]]
    # "AhoCorasickSet/Map": aaaa with a, aa, aaa, aaaa -> a x4, aa x3, aaa x2, aaaa x1
    r = Oracle(FAM_AC, ["a", "aa", "aaa", "aaaa"]).match("aaaa")
    assert sorted(np.bincount(r[:, 2], minlength=4).tolist(), reverse=True) == [4, 3, 2, 1]


def test_match_queue_sequences():
    # T/MatchQueueTest.java:10-20  testMatchQueue
    q = MatchQueue()
    for ln, idx in [(3, 3), (3, 6), (3, 9), (9, 10)]:
        q.push(ln, idx)
    got = q.match_and_clear(10)
    q.push(7, 10)
    got += q.match_and_clear(10)
    assert [(e, e - s) for s, e in got] == [(3, 3), (6, 3), (9, 3), (10, 7)]
    # :23-31 testMatchQueueExtendingOverlap
    q = MatchQueue()
    q.push(3, 3); q.push(4, 4); q.push(2, 5)
    assert [(e, e - s) for s, e in q.match_and_clear(4)] == [(4, 4)]
    # :34-43 testMatchQueueSimple
    q = MatchQueue()
    q.push(3, 3); q.push(2, 3); q.push(2, 4); q.push(2, 5)
    assert [(e, e - s) for s, e in q.match_and_clear(5)] == [(3, 3), (5, 2)]
    # :46-57 testPartialClear
    q = MatchQueue()
    for ln, idx in [(3, 3), (3, 6), (3, 9), (9, 10)]:
        q.push(ln, idx)
    got = q.match_and_clear(4)
    q.push(7, 10)
    got += q.match_and_clear(10)
    assert [(e, e - s) for s, e in got] == [(3, 3), (10, 7)]


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_ac_and_longest_vs_bruteforce(seed):
    rng = np.random.default_rng(seed)
    alpha = [ord(c) for c in "ab"] if seed % 2 == 0 else [ord(c) for c in "abcA"]
    for _ in range(25):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 12)), int(rng.integers(1, 7)), int(rng.integers(0, 200)))
        for cs in (True, False):
            kw = dict(case_sensitive=cs, lower=LOWER)
            assert _as_list(Oracle(FAM_AC, kws, **kw).match(hay)) == [list(m) for m in brute.ac_all(hay, kws, cs, LOWER)]
            assert _as_list(Oracle(FAM_LONGEST, kws, **kw).match(hay)) == [list(m) for m in
                                                                          brute.longest(hay, kws, cs, LOWER)]


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_wholeword_vs_bruteforce(seed):
    rng = np.random.default_rng(100 + seed)
    alpha = [ord(c) for c in "abB -_.9"] + [0x00E9, 0x00C9, 0x4E2D, 0x3002]
    word_alpha = [c for c in alpha if WORD[c]]
    for _ in range(40):
        hay, _ = rand_case(rng, alpha, 1, 1, int(rng.integers(0, 120)))
        _, kws = rand_case(rng, word_alpha, int(rng.integers(1, 10)), 4, 1)
        for cs in (True, False):
            got = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay)
            assert _as_list(got) == [list(m) for m in brute.wholeword(hay, kws, WORD, cs, LOWER)]


def test_wholeword_trim_and_rejection():
    assert trim(" ,abc. ", WORD).tolist() == [ord(c) for c in "abc"]
    assert trim("...", WORD).tolist() == [ord(".")] * 3  # no word chars: left untouched (S/WordCharacters.java:41-62)
    with pytest.raises(IllegalArgumentException):
        Oracle(FAM_WHOLEWORD, ["..."], word_chars=WORD)
    # trimmed keywords match; empty-after-trim keywords are skipped
    o = Oracle(FAM_WHOLEWORD, [" abc,", "", None], word_chars=WORD)
    assert _as_list(o.match("abc abcd abc")) == [[0, 3, 0], [9, 12, 0]]


def test_early_stop_and_last_wins():
    o = Oracle(FAM_AC, ["a", "aa", "aaa", "aaaa"])
    full = o.match("aaaa")
    for k in range(1, len(full) + 1):
        assert _as_list(o.match("aaaa", stop_after=k)) == _as_list(full[:k])
    # duplicate keyword: last index wins (S/AhoCorasickMap.java:49-50)
    assert _as_list(Oracle(FAM_AC, ["ab", "x", "ab"]).match("zabz")) == [[1, 3, 2]]
    # case-insensitive duplicates collapse after folding
    assert _as_list(Oracle(FAM_AC, ["AB", "ab"], case_sensitive=False, lower=LOWER).match("aB")) == [[0, 2, 1]]
    lo = Oracle(FAM_LONGEST, ["a", "aa", "aaa", "aaaa"])
    fl = lo.match(" aaaaaaa aaababababaabaa ")
    for k in (1, 3, len(fl)):
        assert _as_list(lo.match(" aaaaaaa aaababababaabaa ", stop_after=k)) == _as_list(fl[:k])


def test_reference_shaped_nodes_used():
    # both node kinds must actually occur (HashmapNode and RangeNode), otherwise the timed baseline is not reference-shaped
    from ahocorasick_amd import synth
    kws = synth.random_keywords(7, 2000, 2, 6, table=np.concatenate([synth.ALPHA_LOWER, np.arange(0x4E00, 0x4F00, dtype=np.uint16)]))
    o = Oracle(FAM_AC, kws)
    assert o.num_nodes(1) > 0 and o.num_nodes(2) > 0


# ---- match(Readable, ReadableMatchListener): T/MapTest.java:178-188 pins "same number of matches as match(String)" ----

def test_readable_path_reports_the_string_paths_values(fixtures):
    from tests.helpers import LOWER, WORD, fixture_inputs
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        if "keywords_gen" in fx:
            continue
        for fam, key in ((FAM_AC, "AC"), (FAM_LONGEST, "L"), (FAM_WHOLEWORD, "WW")):
            if fx[key] == "IllegalArgumentException":
                continue
            o = Oracle(fam, kws, word_chars=WORD if fam == FAM_WHOLEWORD else None)
            want = [r[2] for r in fx[key]]
            for bufsize in (1, 3, 4096):
                assert o.match_readable(hay, bufsize).tolist() == want, (fx["name"], key, bufsize)


def test_readable_wholeword_scroll_and_refills_fuzz():
    """The Readable WholeWord loop (scroll(), buffer refills, folded lookups) gives the String loop's values whenever
    the word-character table is fold-consistent (always in case-sensitive mode and with the default table)."""
    from tests.helpers import LOWER, WORD, rand_case
    rng = np.random.default_rng(17)
    alpha = [ord(c) for c in "abAB -_.9"] + [0x00E9, 0x00C9, 0x3002]
    for it in range(150):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 30)), 6, int(rng.integers(0, 200)))
        kws = [k for k in kws if all(WORD[c] for c in k.tolist())] or [np.array([97], np.uint16)]
        for cs in (True, False):
            o = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD)
            want = o.match(hay)[:, 2].tolist()
            for bufsize in (1, 2, 5, 64):
                assert o.match_readable(hay, bufsize).tolist() == want, (it, cs, bufsize)
    # early stop: the listener's false ends the scan
    o = Oracle(FAM_WHOLEWORD, ["ab", "b"], word_chars=WORD)
    assert o.match_readable("ab b ab", 2, stop_after=2).tolist() == [0, 1]


def test_readable_wwlongest_reports_the_string_loops_values():
    """S/WholeWordLongestMatchMap.java:54-181 (Readable) against :180-305 (String): the fixtures and fuzzed inputs, buffer
    refills at every unit, fold-consistent tables."""
    import json, os
    from oracle.oracle import FAM_WWLONGEST
    from tests.helpers import LOWER, WORD, rand_case
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fx in json.load(open(os.path.join(root, "tests", "golden", "reference_fixtures.json"))):
        if "WWL" not in fx or "keywords_gen" in fx:
            continue
        o = Oracle(FAM_WWLONGEST, fx["WWL_keywords"], word_chars=WORD)
        for bufsize in (1, 3, 4096):
            assert o.match_readable(fx["haystack"], bufsize).tolist() == [r[2] for r in fx["WWL"]], (fx["name"], bufsize)
    rng = np.random.default_rng(23)
    alpha = [ord(c) for c in "abAB -_.9"] + [0x00E9, 0x00C9, 0x3002]
    for it in range(150):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 30)), 7, int(rng.integers(0, 200)))
        for cs in (True, False):
            o = Oracle(FAM_WWLONGEST, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD)
            want = o.match(hay)[:, 2].tolist()
            for bufsize in (1, 2, 5, 64):
                assert o.match_readable(hay, bufsize).tolist() == want, (it, cs, bufsize)


# ---- ShortestMatchSet / ShortestMatchMap (S/ShortestMatchSet.java) ------------------------------------------------------

def test_shortest_fixtures_and_reference_test_counts(fixtures):
    from oracle.oracle import FAM_SHORTEST
    for fx in fixtures:
        hay, kws = fixture_inputs(fx)
        if "keywords_gen" not in fx:
            kws = fx["S_keywords"]  # T/ShortestMatchTest.java:51-59 sorts the keywords by length
        got = _as_list(Oracle(FAM_SHORTEST, kws).match(hay))
        assert got == fx["S"], fx["name"]
        assert len(got) == fx["S_count"], fx["name"]  # the count T/ShortestMatchTest.java:30-42 expects


def test_shortest_literal_restatement_equals_closed_form_fuzz():
    """Insertion order (longer keyword before its prefix), duplicates (first value wins), inherited suffix matches,
    case folding: the literal restatement of the constructor and the lagging match loop against the closed form."""
    from oracle.oracle import FAM_SHORTEST
    rng = np.random.default_rng(23)
    for it in range(400):
        alpha = [[97, 98], [97, 98, 99], [97, 98, 65, 66, 0x00E9, 0x00C9]][it % 3]
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 25)), int(rng.choice([2, 4, 7])), int(rng.integers(0, 120)))
        for cs in (True, False):
            got = _as_list(Oracle(FAM_SHORTEST, kws, case_sensitive=cs, lower=LOWER).match(hay))
            want = [list(m) for m in brute.shortest(hay, kws, case_sensitive=cs, lower=LOWER)]
            assert got == want, (it, cs)
    # README-style example: the earliest END wins, not the leftmost start (S/ShortestMatchSet.java:8-9)
    assert _as_list(Oracle(FAM_SHORTEST, ["abcd", "bc", "d"]).match("abcd")) == [[1, 3, 1], [3, 4, 2]]
    # early stop
    assert len(Oracle(FAM_SHORTEST, ["a"]).match("aaaa", stop_after=2)) == 2


# ---- WholeWordLongestMatchSet / Map (S/WholeWordLongestMatchSet.java) ---------------------------------------------------

def test_wwlongest_fixtures_and_reference_test_counts(fixtures):
    from oracle.oracle import FAM_WWLONGEST
    for fx in fixtures:
        if "keywords_gen" in fx:
            continue
        got = _as_list(Oracle(FAM_WWLONGEST, fx["WWL_keywords"], word_chars=WORD).match(fx["haystack"]))
        assert got == fx["WWL"], fx["name"]
        assert len(got) == fx["WWL_count"], fx["name"]  # the count T/WholeWordLongestMatchTest.java:46-65 expects


def test_wwlongest_literal_restatement_equals_definition_fuzz():
    """Multi-word keywords, keywords that are prefixes of others across a word boundary, walks that stop inside a
    later word (the scan resumes after that word, S/WholeWordLongestMatchSet.java:85-99), case folding."""
    from oracle.oracle import FAM_WWLONGEST
    rng = np.random.default_rng(29)
    alpha = [ord(c) for c in "abAB  -."] + [0x00E9, 0x00C9]
    for it in range(500):
        hay, kws = rand_case(rng, alpha, int(rng.integers(1, 25)), int(rng.choice([3, 6, 10])), int(rng.integers(0, 150)))
        for cs in (True, False):
            got = _as_list(Oracle(FAM_WWLONGEST, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay))
            want = [list(m) for m in brute.wwlongest(hay, kws, WORD, case_sensitive=cs, lower=LOWER)]
            assert got == want, (it, cs)
    # README/T examples: the longer multi-word keyword wins only when it is a whole-word match
    o = Oracle(FAM_WWLONGEST, ["as", "if", "as if"], word_chars=WORD)
    assert _as_list(o.match("as if"))[0][:2] == [0, 5] and _as_list(o.match("as in"))[0][:2] == [0, 2]
    assert len(o.match("as if as if", stop_after=1)) == 1
