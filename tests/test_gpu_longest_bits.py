"""GPU parity tests of k_longest_bits (csrc/acgpu_longest_bits.hip): LongestMatchSet over a two-letter alphabet with the text
held as one bit per unit -- through the C ABI, against the CPU oracle's restatement of S/LongestMatchSet.java:192-265, bit for
bit and in listener-call order; and against the walk pipeline it replaces."""
import numpy as np
import pytest

from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton, utf16
from oracle.oracle import FAM_LONGEST, Oracle

pytestmark = pytest.mark.gpu

BITS_ALWAYS = 4  # longest_form: k_longest_bits (and k_longest_follow) for short texts too
BITS_NEVER = 1   # longest_form: never k_longest_bits
A_, B_ = ord("a"), ord("b")


@pytest.fixture(autouse=True)
def _reset_tunables():
    yield
    for k, v in [("force_kernel", 0), ("region_units", 0), ("tile_debug", 0), ("no_bits_trie", 0), ("longest_form", 0)]:
        N.set_tunable(k, v)


def _run(a, hay, own=None, entry=None, d_hay=None, with_ids=False):
    """One synchronous device call with Set (or, with_ids, Map) records: (records, kernel name, chain exit)."""
    import torch
    if d_hay is None:
        d_hay = torch.from_numpy(np.ascontiguousarray(hay).view(np.int16)).cuda()
    cap = hay.size + 8
    d_out = torch.empty((cap, 3 if with_ids else 2), dtype=torch.int32, device="cuda")
    kw = {}
    if own is not None:
        kw["own"] = own
    if entry is not None:
        kw["chain_entry"] = entry
    n_out, rc, prof, chain_exit = a.match_device(d_hay.data_ptr(), hay.size, with_ids, d_out.data_ptr(), cap, profile=True,
                                                 stream=torch.cuda.current_stream().cuda_stream, **kw)
    assert rc == N.OK
    return d_out[:n_out].cpu().numpy(), prof["scan_kernel"], chain_exit


def _c4_like(n_kw=3000, word_len=200):
    kws = synth.prefix_closed_keywords(1004, n_kw, word_len=word_len)
    return kws + [utf16("b")]  # (every letter a keyword: what the bit form needs; 'a' is there already)


def test_bits_form_is_what_config_c4_shapes_take_and_equals_the_oracle_at_every_size():
    kws = _c4_like()
    a = Automaton(N.MODE_LONGEST, kws, True)
    orc = Oracle(FAM_LONGEST, kws)
    N.set_tunable("longest_form", BITS_ALWAYS)
    whole = synth.haystack(2004, (1 << 21) + 77, table=synth.ALPHA_AB_75)
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 65535, 65536, 65537, 65536 + 1024 + 33, 200001,
              3 * 65536, (1 << 21) + 77):
        hay = whole[:n]
        got, kname, ex = _run(a, hay)
        want = orc.match(hay)[:, :2]
        assert kname == "k_longest_bits", (n, kname)
        assert got.shape == want.shape and (got == want).all(), n
        assert ex >= n
    # without the switch: long texts take it by themselves, short ones the walk pipeline -- same records
    N.set_tunable("longest_form", 0)
    got, kname, _ = _run(a, whole)
    assert kname == "k_longest_bits" and (got == orc.match(whole)[:, :2]).all()
    got, kname, _ = _run(a, whole[:70000])
    assert kname != "k_longest_bits" and (got == orc.match(whole[:70000])[:, :2]).all()
    N.set_tunable("longest_form", BITS_NEVER)
    got, kname, _ = _run(a, whole)
    assert kname != "k_longest_bits" and (got == orc.match(whole)[:, :2]).all()


@pytest.mark.parametrize("seed", range(4))
def test_bits_form_general_dictionaries_terminals_branches_and_long_paths(seed):
    """Not prefix closed: keywords that end anywhere on a path (the terminal bits of a label), branches below the first level
    (junction entries), paths beyond 31 + 9 units (continuation entries) and beyond the table's reach (the walk through global
    memory), planted in a text whose letter frequencies change."""
    rng = np.random.default_rng(9100 + seed)
    ab = np.array([A_, B_], dtype=np.uint16)
    n = 400000 + int(rng.integers(0, 5000))
    p = rng.choice([0.5, 0.75, 0.9])
    hay = np.where(rng.random(n) < p, A_, B_).astype(np.uint16)
    base = [hay[o:o + ln].copy() for o, ln in zip(rng.integers(0, n - 700, 30).tolist(), rng.integers(20, 600, 30).tolist())]
    kws = [utf16("a"), utf16("b")]
    for b in base:  # a few prefixes of every planted word: terminals scattered over long shared paths
        kws += [b[:k] for k in sorted(set(rng.integers(1, len(b) + 1, 5).tolist()))]
    kws += [ab[rng.integers(0, 2, int(rng.integers(1, 40)))] for _ in range(300)]
    kws += [np.full(k, A_, np.uint16) for k in (3, 17, 40, 41, 72, 150, 400)]
    for at in rng.integers(0, n - 1000, 40).tolist():  # runs of 'a' of every length up to the longest keyword and beyond
        hay[at:at + int(rng.integers(1, 450))] = A_
    a = Automaton(N.MODE_LONGEST, kws, True)
    N.set_tunable("longest_form", BITS_ALWAYS)
    got, kname, _ = _run(a, hay)
    want = Oracle(FAM_LONGEST, kws).match(hay)[:, :2]
    assert kname == "k_longest_bits"
    assert got.shape == want.shape and (got == want).all()
    assert int((want[:, 1] - want[:, 0]).max()) > 100
    N.set_tunable("longest_form", BITS_NEVER)
    old, kname, _ = _run(a, hay)
    assert kname != "k_longest_bits" and (old == want).all()


def test_bits_form_shards_chain_through_entry_and_exit_at_any_position():
    import torch
    kws = _c4_like(2000, 300)
    hay = synth.haystack(77, 700001, table=synth.ALPHA_AB_75)
    a = Automaton(N.MODE_LONGEST, kws, True)
    want = Oracle(FAM_LONGEST, kws).match(hay)[:, :2]
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    N.set_tunable("longest_form", BITS_ALWAYS)
    for cuts in ([0, 65536, 131072, hay.size], [0, 70001, 70002, 70040, 333333, 600000 + 31, hay.size], [0, 1, 2, 33, hay.size]):
        parts, entry = [], 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if entry >= hi:  # the chain jumps over this shard
                continue
            got, kname, ex = _run(a, hay, own=(lo, hi), entry=max(entry, lo), d_hay=d_hay)
            assert kname == "k_longest_bits" and ex >= hi
            parts.append(got)
            entry = ex
        got = np.concatenate(parts)
        assert got.shape == want.shape and (got == want).all(), cuts


def test_bits_form_end_of_the_buffer_and_keywords_cut_off_by_it():
    kws = [utf16("a"), utf16("b"), np.full(50, A_, np.uint16), utf16("ab" * 30), utf16("ba" * 25 + "b")] + [np.full(k, A_, np.uint16) for k in range(2, 12)]
    a = Automaton(N.MODE_LONGEST, kws, True)
    orc = Oracle(FAM_LONGEST, kws)
    N.set_tunable("longest_form", BITS_ALWAYS)
    rng = np.random.default_rng(5)
    for n in (40, 49, 50, 51, 1024 + 49, 65536 + 45, 65536 * 2 - 3, 65536 * 2 + 50):
        for tail in ("a" * 49, "ab" * 29 + "a", "ba" * 25, "b"):
            hay = np.where(rng.random(n) < 0.7, A_, B_).astype(np.uint16)
            t = utf16(tail)[: n]
            hay[n - len(t):] = t
            got, kname, ex = _run(a, hay)
            want = orc.match(hay)[:, :2]
            assert kname == "k_longest_bits"
            assert got.shape == want.shape and (got == want).all(), (n, tail)


@pytest.mark.parametrize("others", [" ", "c\n"])
def test_bits_form_bails_out_on_units_outside_the_alphabet_and_the_walk_pipeline_answers(others):
    import torch
    kws = _c4_like(1500, 100)
    rng = np.random.default_rng(len(others))
    n = 300000
    hay = synth.haystack(11, n, table=synth.ALPHA_AB_75).copy()
    oa = np.array([ord(c) for c in others], dtype=np.uint16)
    for at in rng.integers(0, n, 7).tolist():
        hay[at] = oa[rng.integers(0, len(oa))]
    a = Automaton(N.MODE_LONGEST, kws, True)
    want = Oracle(FAM_LONGEST, kws).match(hay)[:, :2]
    N.set_tunable("longest_form", BITS_ALWAYS)
    got, kname, ex = _run(a, hay)
    assert kname != "k_longest_bits"  # (the profile names the kernel that produced the records)
    assert got.shape == want.shape and (got == want).all() and ex >= n
    # the enqueued form: the ticket learns of the bail-out in _end and redoes the call
    st = torch.cuda.current_stream().cuda_stream
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), n, False, out.data_ptr(), n, stream=st)
    assert rc == N.OK
    m, rc, _ = a.match_device_end(tk)
    assert rc == N.OK and m == len(want) and (out[:m].cpu().numpy() == want).all() and tk.chain_exit == ex
    # a clean text through the same automaton afterwards: the bit form again
    clean = synth.haystack(12, n, table=synth.ALPHA_AB_75)
    got, kname, _ = _run(a, clean)
    assert kname == "k_longest_bits" and (got == Oracle(FAM_LONGEST, kws).match(clean)[:, :2]).all()


def test_bits_form_enqueued_calls_tickets_and_the_device_result():
    import torch
    kws = _c4_like(2500, 150)
    a = Automaton(N.MODE_LONGEST, kws, True)
    orc = Oracle(FAM_LONGEST, kws)
    N.set_tunable("longest_form", BITS_ALWAYS)
    st = torch.cuda.current_stream().cuda_stream
    hays = [synth.haystack(300 + i, 250000 + 4097 * i, table=synth.ALPHA_AB_75) for i in range(3)]
    d_hays = [torch.from_numpy(h.view(np.int16)).cuda() for h in hays]
    wants = [orc.match(h)[:, :2] for h in hays]
    cap = max(len(w) for w in wants) + 8
    outs = [torch.empty((cap, 2), dtype=torch.int32, device="cuda") for _ in hays]
    tickets = []
    for d, o, h in zip(d_hays, outs, hays):
        tk, rc = a.match_device_begin(d.data_ptr(), h.size, False, o.data_ptr(), cap, stream=st, profile=True)
        assert rc == N.OK
        tickets.append(tk)
    for tk, o, w, h in zip(tickets, outs, wants, hays):
        n, rc, prof = a.match_device_end(tk, profile=True)
        assert rc == N.OK and n == len(w) and prof["scan_kernel"] == "k_longest_bits" and prof["scan_ms"] > 0
        assert (o[:n].cpu().numpy() == w).all() and tk.chain_exit >= h.size
    h, d, w = hays[0], d_hays[0], wants[0]
    tk, rc = a.match_device_begin(d.data_ptr(), h.size, False, outs[0].data_ptr(), 5, stream=st)
    n, rc, _ = a.match_device_end(tk)
    assert rc == N.E_OVERFLOW and n == len(w)
    buf = torch.zeros(4 + cap * 2, dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d.data_ptr(), h.size, False, buf.data_ptr() + 16, cap, stream=st, d_result=buf.data_ptr())
    assert rc == N.OK
    torch.cuda.current_stream().synchronize()
    assert int(buf[:2].cpu().numpy().view(np.int64)[0]) == len(w) and int(buf[2]) == 0
    assert a.match_device_end(tk)[0] == len(w)


def test_bits_form_chains_that_never_merge_and_one_letter_alphabets():
    """A text of one letter under {a, b, a^333}: every chain moves in steps of 333, chains that start at different positions
    never meet, so no segment's assumed entry is confirmed -- the kernel has to notice (bail flag) and the walk pipeline answers.
    A dictionary over ONE letter never takes the bit form."""
    kws = [utf16("a"), utf16("b"), np.full(333, A_, np.uint16)]
    a = Automaton(N.MODE_LONGEST, kws, True)
    N.set_tunable("longest_form", BITS_ALWAYS)
    hay = np.full(300000, A_, np.uint16)
    got, kname, _ = _run(a, hay)
    want = Oracle(FAM_LONGEST, kws).match(hay)[:, :2]
    assert kname != "k_longest_bits" and got.shape == want.shape and (got == want).all()
    hay[::1000] = B_  # now they do meet
    got, kname, _ = _run(a, hay)
    want = Oracle(FAM_LONGEST, kws).match(hay)[:, :2]
    assert kname == "k_longest_bits" and got.shape == want.shape and (got == want).all()
    kws1 = [np.full(k, A_, np.uint16) for k in (1, 2, 3, 5, 8, 13, 40, 41, 100, 333)]
    a1 = Automaton(N.MODE_LONGEST, kws1, True)
    hay1 = np.full(150000, A_, np.uint16)
    got, kname, _ = _run(a1, hay1)
    want = Oracle(FAM_LONGEST, kws1).match(hay1)[:, :2]
    assert kname != "k_longest_bits" and got.shape == want.shape and (got == want).all()


def test_bits_form_map_records_ids_from_the_keywords_own_bits():
    """LongestMatchMap through k_longest_bits (S/LongestMatchMap.java:288-360): the id of a record is looked up by the matched text's
    own bits when the records are written, one region later -- keywords of up to 32 units in the table, longer ones by a walk;
    duplicates (the LAST one's index is the id), every size across segment and region seams, shards, a capacity that is too
    small, an enqueued call."""
    import torch
    rng = np.random.default_rng(4242)
    kws = _c4_like(2500, 250)
    kws += [np.array([A_, B_][::-1] * k, dtype=np.uint16) for k in (1, 5, 16, 17, 30)]        # 'baba...' of 2 .. 60 units: around 32
    kws += [kws[i].copy() for i in rng.integers(0, len(kws), 200).tolist()]                  # duplicates: the last index is the id
    a = Automaton(N.MODE_LONGEST, kws, True)
    orc = Oracle(FAM_LONGEST, kws)
    whole = synth.haystack(2005, (1 << 21) + 4099, table=synth.ALPHA_AB_75).copy()
    for at in rng.integers(0, whole.size - 200, 300).tolist():                              # planted 'baba' runs and long runs of 'a'
        whole[at:at + 60] = np.array([B_, A_] * 30, dtype=np.uint16)
    for at in rng.integers(0, whole.size - 400, 100).tolist():
        whole[at:at + int(rng.integers(20, 300))] = A_
    N.set_tunable("longest_form", BITS_ALWAYS)
    for n in (1, 2, 33, 1024, 1025, 65535, 65536, 65537, 131072 + 31, 700001, whole.size):
        hay = whole[:n]
        got, kname, ex = _run(a, hay, with_ids=True)
        want = orc.match(hay)
        assert kname == "k_longest_bits", (n, kname)
        assert got.shape == want.shape and (got == want).all(), n
    want = orc.match(whole)
    assert int((want[:, 1] - want[:, 0]).max()) > 40 and len(set(want[:, 2].tolist())) > 100
    # shards: every shard's records carry the ids
    d_hay = torch.from_numpy(whole.view(np.int16)).cuda()
    parts, entry = [], 0
    for lo, hi in ((0, 333_333), (333_333, 333_400), (333_400, 1_500_001), (1_500_001, whole.size)):
        got, kname, ex = _run(a, whole, own=(lo, hi), entry=max(entry, lo), d_hay=d_hay, with_ids=True)
        assert kname == "k_longest_bits"
        parts.append(got)
        entry = ex
    assert (np.concatenate(parts) == want).all()
    # a capacity that is too small: the exact count, the first records; then through a ticket
    st = torch.cuda.current_stream().cuda_stream
    cap = len(want) // 2
    out = torch.full((cap, 3), -7, dtype=torch.int32, device="cuda")
    n_out, rc, _, _ = a.match_device(d_hay.data_ptr(), whole.size, True, out.data_ptr(), cap, stream=st)
    assert rc == N.E_OVERFLOW and n_out == len(want) and (out.cpu().numpy() == want[:cap]).all()
    out = torch.empty((len(want) + 8, 3), dtype=torch.int32, device="cuda")
    tk, rc = a.match_device_begin(d_hay.data_ptr(), whole.size, True, out.data_ptr(), len(want) + 8, stream=st, profile=True)
    assert rc == N.OK
    n_out, rc, prof = a.match_device_end(tk, profile=True)
    assert rc == N.OK and n_out == len(want) and prof["scan_kernel"] == "k_longest_bits" and (out[:n_out].cpu().numpy() == want).all()
    # and the walk pipeline gives the same records
    N.set_tunable("longest_form", 3)
    got, kname, _ = _run(a, whole, with_ids=True)
    assert kname != "k_longest_bits" and (got == want).all()
