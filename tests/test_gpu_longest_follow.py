"""GPU parity tests of k_longest_follow (csrc/acgpu_longest_follow.hip): LongestMatchSet/Map over dense dictionaries with only the
greedy chain's own positions walked -- through the C ABI, against the CPU oracle's restatement of S/LongestMatchSet.java:192-265
and S/SetMatchQueue.java:45-95, bit for bit and in listener-call order; and against the walk pipeline it stands in for."""
import numpy as np
import pytest

from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton, utf16
from oracle.oracle import FAM_LONGEST, Oracle
from tests.helpers import LOWER

pytestmark = pytest.mark.gpu

FOLLOW_ALWAYS = 4  # longest_form: k_longest_follow (and k_longest_bits) for short texts too
FOLLOW_NEVER = 3    # longest_form: neither k_longest_follow nor k_longest_bits: the walk pipeline


@pytest.fixture(autouse=True)
def _reset_tunables():
    yield
    for k, v in [("force_kernel", 0), ("region_units", 0), ("tile_debug", 0), ("longest_form", 0)]:
        N.set_tunable(k, v)


def _run(a, hay, with_ids, own=None, entry=None, d_hay=None):
    import torch
    if d_hay is None:
        d_hay = torch.from_numpy(np.ascontiguousarray(hay).view(np.int16)).cuda()
    cap = hay.size + 8
    cols = 3 if with_ids else 2
    d_out = torch.empty((cap, cols), dtype=torch.int32, device="cuda")
    kw = {}
    if own is not None:
        kw["own"] = own
    if entry is not None:
        kw["chain_entry"] = entry
    n_out, rc, prof, chain_exit = a.match_device(d_hay.data_ptr(), hay.size, with_ids, d_out.data_ptr(), cap, profile=True,
                                                 stream=torch.cuda.current_stream().cuda_stream, **kw)
    assert rc == N.OK
    return d_out[:n_out].cpu().numpy(), prof["scan_kernel"], chain_exit


@pytest.fixture(scope="module")
def words():
    return synth.readme_dictionary(n=30000)


def test_follow_form_on_a_word_list_equals_the_oracle_at_every_size_set_and_map(words):
    a = Automaton(N.MODE_LONGEST, words, True)
    orc = Oracle(FAM_LONGEST, words)
    whole = synth.readme_text(4, (1 << 20) + 4099, words)
    assert a.info()["tile_kernel"] == 0  # (the single letters are words: no selective filter -- the walk family)
    N.set_tunable("longest_form", FOLLOW_ALWAYS)
    for n in (1, 2, 7, 8, 9, 31, 33, 1023, 1024, 1025, 4097, 65535, 65536, 65537, 200003, (1 << 20) + 4099):
        hay = whole[:n]
        want = orc.match(hay)
        for with_ids in (False, True):
            got, kname, ex = _run(a, hay, with_ids)
            assert kname == "k_longest_follow", (n, kname)
            w = want if with_ids else want[:, :2]
            assert got.shape == w.shape and (got == w).all(), (n, with_ids)
            assert ex >= n
    # long texts take it without the switch, short ones the walk pipeline -- same records
    N.set_tunable("longest_form", 0)
    got, kname, _ = _run(a, whole, True)
    assert kname == "k_longest_follow" and (got == orc.match(whole)).all()
    got, kname, _ = _run(a, whole[:70000], True)
    assert kname != "k_longest_follow" and (got == orc.match(whole[:70000])).all()
    N.set_tunable("longest_form", FOLLOW_NEVER)
    got, kname, _ = _run(a, whole, True)
    assert kname != "k_longest_follow" and (got == orc.match(whole)).all()


def test_follow_form_case_insensitive_dictionary_classes_from_lds_pages(words):
    a = Automaton(N.MODE_LONGEST, words, False)
    orc = Oracle(FAM_LONGEST, words, case_sensitive=False, lower=LOWER)
    hay = synth.readme_text(9, 300001, words).copy()
    rng = np.random.default_rng(3)
    flip = rng.random(hay.size) < 0.2  # upper-case letters all over the text
    hay[flip & (hay >= 97) & (hay <= 122)] -= 32
    N.set_tunable("longest_form", FOLLOW_ALWAYS)
    got, kname, _ = _run(a, hay, True)
    want = orc.match(hay)
    assert kname == "k_longest_follow" and got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("seed", range(3))
def test_follow_form_deep_walks_long_matches_and_texts_without_separators(seed):
    """Walks deeper than the lane's text ring (24 units), matches longer than the ring (a jump beyond it) and than a bitmap word,
    keywords that end on a path and not only at its end, a text of letters only (chains merge by chance, not at spaces), the end
    of the buffer inside a walk."""
    rng = np.random.default_rng(700 + seed)
    letters = np.array([ord(c) for c in "abcdefgh"], dtype=np.uint16)
    n = 500000 + int(rng.integers(0, 999))
    hay = letters[rng.integers(0, len(letters) - (seed % 2) * 4, n)]
    base = [hay[o:o + ln].copy() for o, ln in zip(rng.integers(0, n - 400, 60).tolist(), rng.integers(2, 300, 60).tolist())]
    kws = [utf16(c) for c in "abcd"]
    for b in base:
        kws += [b[:k] for k in sorted(set(rng.integers(1, len(b) + 1, 4).tolist()))]
    kws += [letters[rng.integers(0, 8, int(rng.integers(1, 12)))] for _ in range(2000)]
    tail = base[0][:40]
    hay[n - len(tail) + 3:] = tail[: len(tail) - 3]  # a long keyword cut off by the end of the buffer
    a = Automaton(N.MODE_LONGEST, kws, True)
    want = Oracle(FAM_LONGEST, kws).match(hay)
    N.set_tunable("longest_form", FOLLOW_ALWAYS)
    for with_ids in (False, True):
        got, kname, _ = _run(a, hay, with_ids)
        w = want if with_ids else want[:, :2]
        assert kname == "k_longest_follow" and got.shape == w.shape and (got == w).all()
    assert int((want[:, 1] - want[:, 0]).max()) > 64
    N.set_tunable("longest_form", FOLLOW_NEVER)
    old, kname, _ = _run(a, hay, True)
    assert kname != "k_longest_follow" and (old == want).all()


def test_follow_form_shards_chains_that_never_merge_and_tickets(words):
    import torch
    a = Automaton(N.MODE_LONGEST, words, True)
    hay = synth.readme_text(5, 400001, words)
    want = Oracle(FAM_LONGEST, words).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    N.set_tunable("longest_form", FOLLOW_ALWAYS)
    for cuts in ([0, 65536, 131072, hay.size], [0, 70001, 70002, 70040, 333333, hay.size], [0, 1, 2, 33, hay.size]):
        parts, entry = [], 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if entry >= hi:
                continue
            got, kname, ex = _run(a, hay, True, own=(lo, hi), entry=max(entry, lo), d_hay=d_hay)
            assert kname == "k_longest_follow" and ex >= hi
            parts.append(got)
            entry = ex
        got = np.concatenate(parts)
        assert got.shape == want.shape and (got == want).all(), cuts
    # enqueued calls: count and chain exit through the ticket, the device result in stream order
    st = torch.cuda.current_stream().cuda_stream
    cap = len(want) + 8
    out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, out.data_ptr(), cap, stream=st, profile=True)
    assert rc == N.OK
    n, rc, prof = a.match_device_end(tk, profile=True)
    assert rc == N.OK and n == len(want) and prof["scan_kernel"] == "k_longest_follow" and (out[:n].cpu().numpy() == want).all()
    tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, out.data_ptr(), 5, stream=st)
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.E_OVERFLOW and n == len(want)
    # a text on which chains that start at different positions never meet: {a, aaa} over a run of a's and nothing else -- the
    # kernel has to notice (every step is 3, a chain keeps its residue), and the walk pipeline answers
    kw2 = [utf16("c"), np.full(3, ord("c"), np.uint16), utf16("d")]
    a2 = Automaton(N.MODE_LONGEST, kw2, True)
    N.set_tunable("longest_form", FOLLOW_ALWAYS | 1 | 8)  # (1: every letter a keyword, k_longest_bits would take this one first; 8: k_longest_follow over an alphabet of up to four letters too)
    run = np.full(300000, ord("c"), np.uint16)
    got, kname, _ = _run(a2, run, False)
    want2 = Oracle(FAM_LONGEST, kw2).match(run)[:, :2]
    assert kname != "k_longest_follow" and got.shape == want2.shape and (got == want2).all()
    # ... and the pool remembers: its next call does not try again (a fresh automaton does)
    got, kname, _ = _run(a2, run[:200000], False)
    assert kname != "k_longest_follow" and (got == Oracle(FAM_LONGEST, kw2).match(run[:200000])[:, :2]).all()
    run[::1000] = ord("d")
    a3 = Automaton(N.MODE_LONGEST, kw2, True)
    got, kname, _ = _run(a3, run, False)
    want2 = Oracle(FAM_LONGEST, kw2).match(run)[:, :2]
    assert kname == "k_longest_follow" and got.shape == want2.shape and (got == want2).all()


def test_follow_form_two_letter_text_walks_that_run_three_blocks_ahead_of_their_start():
    """Over {a, b} with keywords of up to 40 units most walks are deep: one that ends with the block three ahead of its start on
    its way used to take the ring slot of the block the chain goes on in (the next walk read the wrong text)."""
    kws = synth.random_keywords(32, 300, 2, 40, table=synth.ALPHA_LOWER[:2])
    n = (1 << 21) + 12345
    hay = synth.haystack(403, n, table=synth.ALPHA_LOWER[:2])
    a = Automaton(N.MODE_LONGEST, kws, True)
    want = Oracle(FAM_LONGEST, kws).match(hay)
    got, kname, _ = _run(a, hay, True)  # (alphabets of up to four letters take the walk pipeline's root table by default)
    assert kname != "k_longest_follow" and got.shape == want.shape and (got == want).all()
    N.set_tunable("longest_form", 8)  # ... k_longest_follow there too
    got, kname, _ = _run(a, hay, True)
    assert kname == "k_longest_follow" and got.shape == want.shape and (got == want).all()
