"""GPU parity tests of k_ac_states / k_ac_states_out (csrc/acgpu_states.hip): AhoCorasickSet/Map over dictionaries that match densely
-- the automaton's state behind every unit over the compact automaton of acgpu_build.cpp 6d, then the records from the states --
through the C ABI, against the CPU oracle's restatement of S/AhoCorasickSet.java:193-252 (output walk :522-535), bit for bit and in
listener-call order; and against the tile kernel it stands in for."""
import numpy as np
import pytest

from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton, utf16
from oracle.oracle import FAM_AC, FAM_SHORTEST, Oracle
from tests.helpers import LOWER, oracle_parallel

pytestmark = pytest.mark.gpu

STATES_ALWAYS = 6  # all_form: k_ac_states whatever the pool's last call found, for short texts too
STATES_NEVER = 1


@pytest.fixture(autouse=True)
def _reset_tunables():
    yield
    for k, v in [("force_kernel", 0), ("tile_debug", 0), ("all_form", 0), ("tile_form", 0)]:
        N.set_tunable(k, v)


def _run(a, hay, with_ids, own=None, d_hay=None, cap=None, text_begin=True):
    import torch
    if d_hay is None:
        d_hay = torch.from_numpy(np.ascontiguousarray(hay).view(np.int16)).cuda()
    cap = hay.size * 4 + 8 if cap is None else cap
    cols = 3 if with_ids else 2
    d_out = torch.empty((cap, cols), dtype=torch.int32, device="cuda")
    kw = {}
    if own is not None:
        kw["own"] = own
    n_out, rc, prof, _ = a.match_device(d_hay.data_ptr(), hay.size, with_ids, d_out.data_ptr(), cap, profile=True, text_begin=text_begin,
                                        stream=torch.cuda.current_stream().cuda_stream, **kw)
    assert rc == N.OK, rc
    return d_out[:n_out].cpu().numpy(), prof["scan_kernel"]


@pytest.fixture(scope="module")
def words():
    return synth.readme_dictionary(n=30000)


def test_states_form_on_a_word_list_equals_the_oracle_at_every_size_set_and_map(words):
    a = Automaton(N.MODE_ALL, words, True)
    orc = Oracle(FAM_AC, words)
    whole = synth.readme_text(4, (1 << 20) + 4099, words)
    N.set_tunable("all_form", STATES_ALWAYS)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 31, 33, 511, 512, 513, 1023, 1025, 4095, 4096, 4097, 32767, 32768, 32769, 200003, (1 << 20) + 4099):
        hay = whole[:n]
        want = orc.match(hay)
        for with_ids in (False, True):
            got, kname = _run(a, hay, with_ids)
            assert kname == "k_ac_states", (n, kname)
            w = want if with_ids else want[:, :2]
            assert got.shape == w.shape and (got == w).all(), (n, with_ids)
    N.set_tunable("all_form", STATES_NEVER)
    got, kname = _run(a, whole, True)
    assert kname != "k_ac_states" and (got == orc.match(whole)).all()


def test_states_form_is_taken_when_the_pools_last_call_found_dense_matches(words):
    a = Automaton(N.MODE_ALL, words, True)
    orc = Oracle(FAM_AC, words)
    hay = synth.readme_text(11, (1 << 20) + 77, words)
    want = orc.match(hay)
    got, kname = _run(a, hay, True)
    assert kname != "k_ac_states" and (got == want).all()  # (a pool's first call: the tile kernel)
    got, kname = _run(a, hay, True)
    assert kname == "k_ac_states" and got.shape == want.shape and (got == want).all()
    got, kname = _run(a, hay[:70000], True)  # short texts keep the tile kernel
    assert kname != "k_ac_states" and (got == orc.match(hay[:70000])).all()
    # a text in which nothing matches sends the pool back
    blank = np.full((1 << 20) + 5, ord("#"), np.uint16)
    got, kname = _run(a, blank, True)
    assert kname == "k_ac_states" and got.shape[0] == 0
    got, kname = _run(a, blank, True)
    assert kname != "k_ac_states" and got.shape[0] == 0


def test_states_form_a_pools_first_call_on_a_long_text_counts_its_beginning(words):
    a = Automaton(N.MODE_ALL, words, True)
    hay = synth.readme_text(12, (1 << 23) + 4321, words)
    got, kname = _run(a, hay, False, cap=hay.size * 2)
    want = Oracle(FAM_AC, words).match(hay, cap=hay.size * 2)[:, :2]
    assert kname == "k_ac_states" and got.shape == want.shape and (got == want).all()
    b = Automaton(N.MODE_ALL, words, True)  # ... and a text that begins without matches keeps the tile kernel
    hay2 = hay.copy()
    hay2[: (1 << 20) + 100] = ord("#")
    got, kname = _run(b, hay2, False, cap=hay.size * 2)
    want = Oracle(FAM_AC, words).match(hay2, cap=hay.size * 2)[:, :2]
    assert kname != "k_ac_states" and got.shape == want.shape and (got == want).all()


def test_states_form_case_insensitive_dictionary_classes_from_lds_pages(words):
    a = Automaton(N.MODE_ALL, words, False)
    orc = Oracle(FAM_AC, words, case_sensitive=False, lower=LOWER)
    hay = synth.readme_text(9, 300001, words).copy()
    rng = np.random.default_rng(3)
    flip = rng.random(hay.size) < 0.2
    hay[flip & (hay >= 97) & (hay <= 122)] -= 32
    N.set_tunable("all_form", STATES_ALWAYS)
    got, kname = _run(a, hay, True)
    want = orc.match(hay)
    assert kname == "k_ac_states" and got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("seed", range(4))
def test_states_form_fail_hops_over_small_alphabets_and_keywords_of_up_to_32_units(seed):
    """Texts of letters only over 2-8 letters: the walk leaves compact states through their fail links all the time (several hops
    per unit), keywords nest in each other (several records per position), the longest have 32 units (bit 31 of the mask)."""
    rng = np.random.default_rng(900 + seed)
    letters = np.array([ord(c) for c in "abcdefgh"[: [2, 3, 5, 8][seed]]], dtype=np.uint16)
    n = 400000 + int(rng.integers(0, 999))
    hay = letters[rng.integers(0, len(letters), n)]
    kws = [hay[o:o + ln].copy() for o, ln in zip(rng.integers(0, n - 40, 300).tolist(), rng.integers(1, 33, 300).tolist())]
    kws += [letters[rng.integers(0, len(letters), int(rng.integers(1, 10)))] for _ in range(500)]
    kws.append(hay[1000:1032].copy())
    a = Automaton(N.MODE_ALL, kws, True)
    want = Oracle(FAM_AC, kws).match(hay)
    N.set_tunable("all_form", STATES_ALWAYS)
    for with_ids in (False, True):
        got, kname = _run(a, hay, with_ids, cap=len(want) + 8)
        w = want if with_ids else want[:, :2]
        assert kname == "k_ac_states" and got.shape == w.shape and (got == w).all()
    assert int((want[:, 1] - want[:, 0]).max()) == 32
    # a keyword of 33 units: no mask bit for it, the dictionary keeps the other kernels
    a2 = Automaton(N.MODE_ALL, kws + [hay[2000:2033].copy()], True)
    got, kname = _run(a2, hay[:100000], True, cap=len(want) + 8)
    assert kname != "k_ac_states"


def test_states_form_shards_tickets_and_overflow(words):
    import torch
    a = Automaton(N.MODE_ALL, words, True)
    hay = synth.readme_text(5, 400001, words)
    want = Oracle(FAM_AC, words).match(hay)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    N.set_tunable("all_form", STATES_ALWAYS)
    for cuts in ([0, 65536, 131072, hay.size], [0, 70001, 70002, 70040, 333333, hay.size], [0, 1, 2, 33, hay.size]):
        parts = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            got, kname = _run(a, hay, True, own=(lo, hi), d_hay=d_hay)
            assert kname == "k_ac_states"
            parts.append(got)
        got = np.concatenate(parts)
        assert got.shape == want.shape and (got == want).all(), cuts
    st = torch.cuda.current_stream().cuda_stream
    cap = len(want) + 8
    out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, out.data_ptr(), cap, stream=st, profile=True)
    assert rc == N.OK
    n, rc, prof = a.match_device_end(tk, profile=True)
    assert rc == N.OK and n == len(want) and prof["scan_kernel"] == "k_ac_states" and (out[:n].cpu().numpy() == want).all()
    out.fill_(-7)
    tk, rc = a.match_device_begin(d_hay.data_ptr(), hay.size, True, out.data_ptr(), 5, stream=st)
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.E_OVERFLOW and n == len(want)
    assert (out[:5].cpu().numpy() == want[:5]).all() and (out[5:].cpu().numpy() == -7).all()


def test_shortest_over_the_states_form(words):
    a = Automaton(N.MODE_SHORTEST, words, True)
    hay = synth.readme_text(6, 300007, words)
    want = Oracle(FAM_SHORTEST, words).match(hay)
    N.set_tunable("all_form", STATES_ALWAYS)
    got, _ = _run(a, hay, True)
    assert got.shape == want.shape and (got == want).all()


def test_states_form_without_room_for_the_states_falls_back_to_the_tile_kernel(words):
    a = Automaton(N.MODE_ALL, words, True)
    hay = synth.readme_text(13, 300001, words)
    want = Oracle(FAM_AC, words).match(hay)
    N.set_tunable("all_form", STATES_ALWAYS)
    N.set_tunable("tile_debug", 1 << 40)  # the buffer of 4 bytes per unit "cannot be had"
    got, kname = _run(a, hay, True)
    assert kname != "k_ac_states" and got.shape == want.shape and (got == want).all()


def test_first_calls_density_probe_without_room_falls_back_too(words):
    """A pool's first call on a text of 2^23 units or more counts the records of its first 2^20 units through the states form
    (all_form 0).  If the probe's buffer of 4 bytes per unit cannot be had the call is the tile kernel's, not an error."""
    a = Automaton(N.MODE_ALL, words[:3000], True)
    hay = synth.readme_text(14, (1 << 23) + 5, words[:3000])
    want = oracle_parallel(Oracle(FAM_AC, words[:3000]), hay, "ac", a.info()["max_keyword_len"], cap_per_unit=2.0)
    N.set_tunable("tile_debug", 1 << 40)
    got, kname = _run(a, hay, True)
    assert kname != "k_ac_states" and got.shape == want.shape and (got == want).all()
    N.set_tunable("tile_debug", 0)
    got, kname = _run(a, hay, True)  # (the pool kept no density from the failed probe... but the call above left one: whichever form, the records are the same)
    assert got.shape == want.shape and (got == want).all()


def test_states_form_two_hundred_classes_from_pages_and_ids_of_nested_keywords():
    """200 distinct units spread over three blocks of the plane (classes from LDS pages, class numbers up to 200 in a node's edge byte), keywords of 1-4
    units that nest (several per position: the id lists of Map records), duplicates (the last one's index is the id)."""
    rng = np.random.default_rng(77)
    units = np.concatenate([rng.choice(np.arange(0x4E00, 0x5200), 120, replace=False), rng.choice(np.arange(0x0400, 0x0460), 50, replace=False),
                            rng.choice(np.arange(0x61, 0x7B), 20, replace=False), [0x3042, 0x3044, 0x3046, 0xFF21, 0xFF22, 0x1F00, 0x2000, 0x00E9, 0x0131, 0x20AC]]).astype(np.uint16)
    assert len(set(units.tolist())) == 200
    kws = [units[rng.integers(0, 200, int(rng.integers(1, 5)))] for _ in range(3000)]
    kws += [kws[i].copy() for i in range(0, 300, 7)]  # duplicates
    n = 300000 + 17
    hay = units[rng.integers(0, 200, n)]
    hay[::97] = 0x0020
    a = Automaton(N.MODE_ALL, kws, True)
    want = Oracle(FAM_AC, kws).match(hay)
    N.set_tunable("all_form", STATES_ALWAYS)
    for with_ids in (False, True):
        got, kname = _run(a, hay, with_ids, cap=len(want) + 8)
        w = want if with_ids else want[:, :2]
        assert kname == "k_ac_states" and got.shape == w.shape and (got == w).all()
    assert a.info()["n_classes"] == 201


def test_host_entry_and_stream_over_a_dense_text_change_to_the_states_form_on_the_way(words):
    """acgpu_match_u16 on a text of more than one 2^24-unit chunk, and the chunked match(Readable) entry: the first chunk runs the tile
    kernel, what it finds sends the later ones through k_ac_states; the records are the oracle's whatever form each chunk took."""
    from ahocorasick_amd.strings import Stream
    kws = [w for w in words if len(w) > 2]
    a = Automaton(N.MODE_ALL, kws, True)
    block = synth.readme_text(21, (1 << 22) + 12345, words)
    hay = np.tile(block, 5)[: (1 << 24) + (1 << 21) + 777]
    want = Oracle(FAM_AC, kws).match(hay, cap=hay.size)
    got = a.match_host(hay, True, cap=len(want) + 16)
    assert got.shape == want.shape and (got == want).all()
    st = Stream(a, with_ids=True)
    parts, step = [], (1 << 22) + 4321
    for lo in range(0, 1 << 24, step):
        hi = min(lo + step, 1 << 24)
        parts.append(st.feed(hay[lo:hi], final=(hi == 1 << 24), cap=1 << 22))
    st.close()
    got_s = np.concatenate(parts)
    w16 = want[want[:, 1] <= (1 << 24)]
    assert got_s.shape == w16.shape and (got_s == w16.astype(np.int64)).all()


def test_states_form_on_a_rank_buffer_that_begins_inside_the_text(words):
    """What one rank of the multi-GPU drivers holds: [max_len - 1 units of left halo | its own range], not the text's beginning --
    the walk of the first chunk starts at the buffer's first unit, a match belongs to the range that owns its last unit."""
    a = Automaton(N.MODE_ALL, words, True)
    hay = synth.readme_text(17, 700001, words)
    want = Oracle(FAM_AC, words).match(hay)
    halo = a.info()["max_keyword_len"] - 1
    N.set_tunable("all_form", STATES_ALWAYS)
    for lo, hi in ((123457, 400003), (halo, 5000), (halo + 1, halo + 2), (350000, 700001)):
        buf = hay[lo - halo:hi]
        got, kname = _run(a, buf, True, own=(halo, halo + hi - lo), text_begin=False)
        w = want[(want[:, 1] > lo) & (want[:, 1] <= hi)].copy()
        w[:, :2] -= lo - halo
        assert kname == "k_ac_states" and got.shape == w.shape and (got == w).all(), (lo, hi)
