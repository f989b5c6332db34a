"""GPU parity tests of the perfect hash behind k_ww_pp (csrc/acgpu_build.cpp 5b "hash and displace", csrc/acgpu_wholeword.hip
PH = true): every run of word characters reads ONE slot of a table that holds every keyword in a slot of its own.  Results are the
CPU oracle's (S/WholeWordMatchMap.java:155-240 restated in oracle/ac_oracle.c), record for record, through the C ABI."""
import numpy as np
import pytest

from ahocorasick_amd import WholeWordMatchMap, WholeWordMatchSet
from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton
from oracle.oracle import FAM_WHOLEWORD, Oracle
from tests.helpers import LOWER, WORD, oracle_parallel

pytestmark = pytest.mark.gpu

KNOBS = [("force_kernel", 0), ("tile_debug", 0), ("ww_no_ph", 0), ("ww_ph_lambda", 0), ("ww_first_seed", 0), ("region_units", 0), ("tile_form", 0),
         ("ww_ramp_pm", -1), ("ww_no_byte_pages", 0), ("ww_block", 0)]


@pytest.fixture(autouse=True)
def _reset_tunables():
    yield
    for k, v in KNOBS:
        N.set_tunable(k, v)


def _dev(a, hay, with_ids, cap):
    import torch
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    d_out = torch.empty((max(cap, 1), 3 if with_ids else 2), dtype=torch.int32, device="cuda")
    n_out, rc, prof, _ = a.match_device(d_hay.data_ptr(), hay.size, with_ids, d_out.data_ptr(), cap,
                                        stream=torch.cuda.current_stream().cuda_stream, profile=True)
    assert rc == N.OK, rc
    return d_out[:n_out].cpu().numpy(), prof


def _text(rng, kws, alpha, n_words, extra=()):
    """keywords, near misses (one unit changed / one unit more / one unit less), random runs of every length up to 40, and the
    `extra` runs, between one or two non-word units; the text ends inside a word"""
    parts = []
    pool = list(kws) + list(extra)
    for _ in range(n_words):
        k = np.array(pool[int(rng.integers(0, len(pool)))], dtype=np.uint16).copy()
        mode = int(rng.integers(0, 6))
        if mode == 1 and len(k) > 1:
            k[int(rng.integers(0, len(k)))] = alpha[int(rng.integers(0, len(alpha)))]
        elif mode == 2:
            k = np.concatenate([k, alpha[rng.integers(0, len(alpha), 1)]])
        elif mode == 3 and len(k) > 1:
            k = k[:-1]
        elif mode == 4:
            k = alpha[rng.integers(0, len(alpha), int(rng.integers(1, 41)))]
        parts.append(k)
        parts.append(np.array([0x20, 0x2C][: int(rng.integers(1, 3))], dtype=np.uint16))
    return np.concatenate(parts[:-1])


@pytest.mark.parametrize("lam", [0, 1, 6, 4096])
def test_perfect_hash_at_every_bucket_size(lam):
    """lambda = keywords per bucket: 4 is the product's; 1: every keyword a bucket of its own; 6: buckets of up to a dozen
    keywords that collide in the first level, whose displacement takes thousands of tries; 4096: so many keywords per bucket
    that no displacement places them all (no perfect hash: the two-choice table serves).  Keywords of 1 .. 32 units, case-insensitive, over Latin, Greek and accented
    letters whose folds meet."""
    rng = np.random.default_rng(900 + lam)
    alpha = np.array([ord(c) for c in "abcABC"] + [0x00E9, 0x00C9, 0x0391, 0x03B1, 0x0416, 0x0436, 0x4E2D], dtype=np.uint16)
    seen, kws = set(), []
    lens = list(range(1, 33))
    while len(kws) < 6000:
        k = alpha[rng.integers(0, len(alpha), int(rng.choice(lens)))]
        f = bytes(LOWER[k].astype(np.uint16).tobytes())
        if f not in seen:
            seen.add(f)
            kws.append(k)
    N.set_tunable("ww_ph_lambda", lam)
    hay = _text(rng, kws, alpha, 60000)
    for cs in (False, True):
        a = Automaton(N.MODE_WHOLEWORD, kws, cs, word_chars=WORD, lower=None if cs else LOWER)
        want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=cs, lower=LOWER, word_chars=WORD).match(hay, cap=1 << 20)
        got, prof = _dev(a, hay, True, len(want) + 8)
        assert got.shape == want.shape and (got == want).all(), (lam, cs)
        kn = prof["scan_kernel"]
        assert kn.startswith("k_ww_pp") and kn.endswith(", false>" if lam == 4096 else ", true>"), kn
        got2, _ = _dev(a, hay, False, len(want) + 8)
        assert (got2 == want[:, :2]).all()


def test_perfect_hash_and_two_choice_table_agree_on_config_5s_words():
    """20 k mixed-script words (BASELINE config 5's generator), a text of tokens drawn from them with their case swapped, split
    words and non-words: the perfect hash, the two-choice table behind the Bloom filter (tile_debug 2^29 on the same automaton;
    ww_no_ph at build time) and the oracle agree; every hash seed gives the same records."""
    words = synth.mixed_script_words(77, 20000)
    hay = synth.token_stream_haystack(78, 1 << 21, words, synth.swapcase_table())
    want = oracle_parallel(Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD), hay, "wholeword", 13)
    assert len(want) > 50000
    a = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD, lower=LOWER)
    got, prof = _dev(a, hay, True, len(want) + 8)
    assert prof["scan_kernel"].endswith(", true>") and got.shape == want.shape and (got == want).all()
    N.set_tunable("tile_debug", 1 << 29)
    got, prof = _dev(a, hay, True, len(want) + 8)
    assert prof["scan_kernel"].endswith(", false>") and (got == want).all()
    N.set_tunable("tile_debug", 0)
    N.set_tunable("ww_no_ph", 1)
    b = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD, lower=LOWER)
    got, prof = _dev(b, hay, True, len(want) + 8)
    assert prof["scan_kernel"].endswith(", false>") and (got == want).all()
    N.set_tunable("ww_no_ph", 0)
    for seed in (1, 5, 7):
        N.set_tunable("ww_first_seed", seed)
        c = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD, lower=LOWER)
        got, prof = _dev(c, hay, True, len(want) + 8)
        assert prof["scan_kernel"].endswith(", true>") and (got == want).all(), seed


def test_one_keyword_and_keywords_that_differ_in_their_last_unit_only():
    """the smallest tables (one slot; eight), keywords of 12, 13 and 16 units that agree in their first 12 (inline) units --
    the slot's tag and inline units pass, the record decides -- and runs that are a keyword plus one unit"""
    rng = np.random.default_rng(5)
    alpha = np.array([ord(c) for c in "xyz"], dtype=np.uint16)
    stem = np.array([ord(c) for c in "abcdefghijkl"], dtype=np.uint16)
    for kws in ([stem], [stem, np.append(stem, ord("m")), np.append(stem, ord("n")), np.concatenate([stem, stem[:4]]), np.concatenate([stem, stem[:3], [ord("q")]]),
                         np.array([ord("a")], dtype=np.uint16)]):
        hay = _text(rng, kws, alpha, 5000, extra=[np.append(stem, ord("o")), stem[:11], np.concatenate([stem, stem[:4], [ord("x")]])])
        a = Automaton(N.MODE_WHOLEWORD, kws, True, word_chars=WORD)
        want = Oracle(FAM_WHOLEWORD, kws, word_chars=WORD).match(hay, cap=1 << 20)
        assert len(want) > 500
        got, prof = _dev(a, hay, True, len(want) + 8)
        assert prof["scan_kernel"].endswith(", true>") and got.shape == want.shape and (got == want).all()


def test_listener_api_over_the_perfect_hash():
    kws = ["Zürich", "zurich", "ΑΘΗΝΑ", "x", "straße", "a1b2c3d4e5f6g7h8"]
    hay = "zürich ZURICH αθηνα X y STRASSE Straße a1b2c3d4e5f6g7h8 a1b2c3d4e5f6g7h8i x"
    m = WholeWordMatchMap(kws, list(range(len(kws))), False)
    want = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay).tolist()
    assert m.find_all(hay).tolist() == want and len(want) == 7
    seen = []
    WholeWordMatchSet(kws, False).match(hay, lambda h, s, e: (seen.append(h[s:e]) or len(seen) < 3))
    assert seen == ["zürich", "ZURICH", "αθηνα"]


def test_fused_tail_of_the_word_kernel_in_every_form():
    """k_ww_pp with the ordering fused into its tail (csrc/acgpu_wholeword.hip, TileLaunch::fused_tail): spans with and without a
    ramp, the copy pass of its own (tile_form 2), Set and Map records, a capacity smaller than the matches, shards with halos,
    short texts after long ones on one pool, tickets in flight, the delta pages instead of the byte pages -- all the oracle's."""
    import torch
    words = synth.mixed_script_words(91, 5000)
    hay = synth.token_stream_haystack(92, (1 << 21) + 77, words, synth.swapcase_table())
    orc = Oracle(FAM_WHOLEWORD, words, case_sensitive=False, lower=LOWER, word_chars=WORD)
    want = oracle_parallel(orc, hay, "wholeword", 13)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def run(a, n, with_ids, cap, **kw):
        d_out = torch.full((max(cap, 1), 3 if with_ids else 2), -7, dtype=torch.int32, device="cuda")
        n_out, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, with_ids, d_out.data_ptr(), cap, stream=st, profile=True, **kw)
        return d_out, n_out, rc, prof

    a = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD, lower=LOWER)
    for ramp in (-1, 0, 90, 1000):
        N.set_tunable("ww_ramp_pm", ramp)
        for with_ids in (True, False):
            d_out, n_out, rc, prof = run(a, hay.size, with_ids, len(want) + 50)
            assert rc == N.OK and n_out == len(want) and prof["finalize_ms"] == 0.0, (ramp, with_ids)
            assert (d_out[:n_out].cpu().numpy() == (want if with_ids else want[:, :2])).all(), (ramp, with_ids)
            assert (d_out[n_out:].cpu().numpy() == -7).all()
    N.set_tunable("ww_ramp_pm", -1)
    for block in (64, 640, 896):  # workgroups of 1, 10 and 14 waves (the spans are dealt by waves per workgroup)
        N.set_tunable("ww_block", block)
        d_out, n_out, rc, _ = run(a, hay.size, True, len(want) + 50)
        assert rc == N.OK and n_out == len(want) and (d_out[:n_out].cpu().numpy() == want).all(), block
    N.set_tunable("ww_block", 0)
    N.set_tunable("tile_form", 2)
    d_out, n_out, rc, prof = run(a, hay.size, True, len(want) + 50)
    assert rc == N.OK and n_out == len(want) and prof["finalize_ms"] > 0.0 and (d_out[:n_out].cpu().numpy() == want).all()
    N.set_tunable("tile_form", 0)
    cap = len(want) // 2
    d_out, n_out, rc, _ = run(a, hay.size, True, cap)
    assert rc == N.E_OVERFLOW and n_out == len(want) and (d_out[:cap].cpu().numpy() == want[:cap]).all()
    # shards of the buffer; a rank's buffer with one unit of left context and max_len + 1 units of right halo
    cuts = [0, 600_001, 600_002, 1_500_000, hay.size]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d_out, n_out, rc, _ = run(a, hay.size, True, len(want) + 50, own=(lo, hi))
        assert rc == N.OK
        parts.append(d_out[:n_out].cpu().numpy())
    assert (np.concatenate(parts) == want).all()
    # short texts after long ones (fewer workgroups, a text shorter than a tile): the counters were left clean
    for m in (3, 511, 513, 8200, 1 << 18):
        w = orc.match(hay[:m], cap=1 << 20)
        d_out, n_out, rc, _ = run(a, m, True, len(w) + 8)
        assert rc == N.OK and n_out == len(w) and (d_out[:n_out].cpu().numpy() == w).all(), m
    # three tickets one behind the other
    outs = [torch.empty((len(want) + 8, 3), dtype=torch.int32, device="cuda") for _ in range(3)]
    tks = []
    for o in outs:
        tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, o.data_ptr(), len(want) + 8, stream=st)
        assert rc == N.OK
        tks.append(tk)
    for tk, o in zip(tks, outs):
        n_out, rc, _ = a.match_device_end(tk)
        assert rc == N.OK and n_out == len(want) and (o[:n_out].cpu().numpy() == want).all()
    # the delta pages + word bits (FOLD 1) and the case-sensitive form under the same tail
    N.set_tunable("ww_no_byte_pages", 1)
    b = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD, lower=LOWER)
    N.set_tunable("ww_no_byte_pages", 0)
    d_out, n_out, rc, prof = run(b, hay.size, True, len(want) + 50)
    assert rc == N.OK and prof["scan_kernel"].startswith("k_ww_pp<1") and (d_out[:n_out].cpu().numpy() == want).all()
    cs = Automaton(N.MODE_WHOLEWORD, words, True, word_chars=WORD)
    want_cs = oracle_parallel(Oracle(FAM_WHOLEWORD, words, word_chars=WORD), hay, "wholeword", 13)
    d_out, n_out, rc, prof = run(cs, hay.size, True, len(want_cs) + 50)
    assert rc == N.OK and prof["scan_kernel"].startswith("k_ww_pp<0") and n_out == len(want_cs) and (d_out[:n_out].cpu().numpy() == want_cs).all()
