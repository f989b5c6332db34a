"""GPU tests of the multi-device split behind the C ABI (include/acgpu.h: acgpu_match_u16_multi, acgpu_comm_*,
acgpu_match_device_allgather) -- ONE host process, the reference's one-call shape (S/StringSet.java:3-5).  A one-GPU box
names its device several times (every further share gets its own scratch pool, streams and table upload on that device),
so the sharding, the halos, the speculative scans and the window repairs of the chain families all run; RCCL itself runs as
a single-process communicator over the one device (ncclCommInitAll, ncclAllGather in a group)."""
import ctypes
import os

import numpy as np
import pytest

from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton, Comm
from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_SHORTEST, FAM_WHOLEWORD, FAM_WWLONGEST, Oracle
from tests.helpers import LOWER, WORD, oracle_parallel

pytestmark = pytest.mark.gpu

FAMILIES = ["ac", "ac_ci", "wholeword", "longest", "longest_map", "shortest", "wwlongest"]


@pytest.fixture(autouse=True)
def _reset_tunables():
    # (the product cuts a text only into shares of 2^22 units and more -- a share has fixed costs; these tests cut short texts)
    N.set_tunable("multi_min_share", 1024)
    yield
    for k, v in [("force_kernel", 0), ("region_units", 0), ("tile_debug", 0), ("multi_min_share", 1 << 22)]:
        N.set_tunable(k, v)


def _wwl_case(seed, n):
    table = np.array([ord(c) for c in "abcE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
    rng = np.random.default_rng(seed)
    words = synth.random_keywords(seed, 120, 1, 5, table=table[:4])
    sp = np.array([32], dtype=np.uint16)
    kws = list(words[:60]) + [np.concatenate([words[int(i)], sp, words[int(j)]]) for i, j in rng.integers(0, 120, (80, 2))] + \
          [np.concatenate([words[int(i)], sp, words[int(j)], np.array([44, 32], np.uint16), words[int(k)]])
           for i, j, k in rng.integers(0, 120, (30, 3))]
    return kws, synth.haystack(seed + 1000, n, table=table)


def family_case(family, n, seed=0):
    """(automaton, with_ids, haystack, the oracle's records for the whole haystack)"""
    if family in ("ac", "ac_ci"):
        kws = synth.random_keywords(33 + seed, 2000, 3, 11, table=synth.ALPHA_LOWER[:12])
        hay = synth.haystack(401 + seed, n, table=synth.ALPHA_LOWER[:12])
        cs = family == "ac"
        if not cs:  # upper-case stretches in the text: the folded range classes
            up = synth.haystack(77 + seed, n, table=np.array([0, 32], dtype=np.uint16))
            hay = (hay - up).astype(np.uint16)
        auto = Automaton(N.MODE_ALL, kws, cs)
        orc = Oracle(FAM_AC, kws, case_sensitive=cs, lower=None if cs else LOWER)
        want = oracle_parallel(orc, hay, "ac", max(len(k) for k in kws), cap_per_unit=0.05) if n > (1 << 22) else orc.match(hay)
        return auto, True, hay, want
    if family == "wholeword":
        table = np.array([ord(c) for c in "abcdE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(31 + seed, 300, 1, 6, table=table[:5])
        hay = synth.haystack(402 + seed, n, table=table)
        auto = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=WORD)
        orc = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=LOWER, word_chars=WORD)
        want = oracle_parallel(orc, hay, "wholeword", max(len(k) for k in kws), cap_per_unit=0.2) if n > (1 << 22) else orc.match(hay)
        return auto, True, hay, want
    if family in ("longest", "longest_map"):
        kws = synth.random_keywords(32 + seed, 300, 2, 40, table=synth.ALPHA_LOWER[:2])
        hay = synth.haystack(403 + seed, n, table=synth.ALPHA_LOWER[:2])
        want = Oracle(FAM_LONGEST, kws).match(hay, cap=n // 2 + 16)
        if family == "longest":  # Set records: the root-table stream kernel
            return Automaton(N.MODE_LONGEST, kws, True), False, hay, want[:, :2]
        return Automaton(N.MODE_LONGEST, kws, True), True, hay, want
    if family == "shortest":
        kws = synth.random_keywords(34 + seed, 300, 2, 30, table=synth.ALPHA_LOWER[:3])
        hay = synth.haystack(404 + seed, n, table=synth.ALPHA_LOWER[:3])
        return Automaton(N.MODE_SHORTEST, kws, True), True, hay, Oracle(FAM_SHORTEST, kws).match(hay, cap=n // 2 + 16)
    kws, hay = _wwl_case(405 + seed, n)
    auto = Automaton(N.MODE_WWLONGEST, kws, False, word_chars=WORD)
    return auto, True, hay, Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=LOWER, word_chars=WORD).match(hay, cap=n // 4 + 16)


@pytest.mark.parametrize("family", FAMILIES)
def test_match_u16_multi_shares_on_one_device_equal_the_oracle(family):
    """acgpu_match_u16_multi with the one GPU named three (and five, and one) times: contiguous shares with their halos,
    every share fed and scanned by its own host thread on its own streams, the chain families repaired share by share --
    the oracle's records for the whole text, in its order, with global positions."""
    n = 700001
    auto, ids, hay, want = family_case(family, n)
    assert len(want) > 1000
    for devices in ([0, 0, 0], [0], [0, 0, 0, 0, 0]):
        got = auto.match_host(hay, ids, cap=len(want) + 16, devices=devices)
        assert got.shape == want.shape and (got == want).all(), devices
    # the capacity protocol: too small -> ACGPU_E_OVERFLOW with the exact count (match_host retries with it)
    got = auto.match_host(hay, ids, cap=100, devices=[0, 0, 0])
    assert got.shape == want.shape and (got == want).all()
    # a text too short to cut runs on the first device alone; an empty one reports nothing
    assert (auto.match_host(hay[:1500], ids, devices=[0, 0, 0]) == auto.match_host(hay[:1500], ids)).all()
    assert len(auto.match_host(hay[:0], ids, devices=[0, 0])) == 0


def test_match_u16_multi_device_list_from_the_environment(monkeypatch):
    """The Python mirror of the facade: ACGPU_DEVICES (the Java side reads -Dacgpu.devices) makes match(String, ...) a
    multi-device call without touching the reference's API."""
    from ahocorasick_amd import AhoCorasickSet, LongestMatchSet
    kws = ["ab", "abc", "bca", "c"]
    hay = "abcabcab" * 2000
    want = AhoCorasickSet(kws, True).find_all(hay)
    wantl = LongestMatchSet(kws, True).find_all(hay)
    monkeypatch.setenv("ACGPU_DEVICES", "0,0,0,0")
    assert (AhoCorasickSet(kws, True).find_all(hay) == want).all()
    got = []
    LongestMatchSet(kws, True).match(hay, lambda h, s, e: got.append((s, e)) or True)
    assert got == [tuple(r) for r in wantl.tolist()]
    monkeypatch.setenv("ACGPU_DEVICES", "0,7")  # no such device on a one-GPU box
    import torch
    if torch.cuda.device_count() < 8:
        with pytest.raises(N.AcgpuError):
            AhoCorasickSet(kws, True).find_all(hay)


@pytest.mark.parametrize("family", ["ac", "longest", "wwlongest"])
def test_match_u16_multi_shares_of_several_chunks(family):
    """Shares longer than the 2^24-unit chunks of the host pipeline: every share streams its view through its own pinned
    ring, chunk by chunk, the chain handed from chunk to chunk inside a share and repaired between shares."""
    n = 3 * (1 << 24) + (1 << 23) + 4321  # two shares of 1.75 chunks
    auto, ids, hay, want = family_case(family, n, seed=3)
    got = auto.match_host(hay, ids, cap=len(want) + 16, devices=[0, 0])
    assert got.shape == want.shape and (got == want).all()


def test_multi_longest_match_covering_a_whole_share_and_dense_repairs():
    """Longest: a keyword longer than a share (the previous share's last match swallows this one: nothing is reported
    here and the exit passes through), and a dictionary whose chains never merge before the window has grown to the share."""
    a, b = ord("a"), ord("b")
    rng = np.random.default_rng(5)
    hay = np.where(rng.integers(0, 4, 12000) > 0, a, b).astype(np.uint16)
    big = hay[2500:8300].copy()  # 5800 units: covers share 1 of 4 (3000..6000) entirely
    kws = [big, np.array([a], np.uint16), np.array([a, b], np.uint16), np.array([b, a, a], np.uint16)]
    want = Oracle(FAM_LONGEST, kws).match(hay)
    auto = Automaton(N.MODE_LONGEST, kws, True)
    for devices in ([0, 0, 0, 0], [0, 0], [0, 0, 0]):
        got = auto.match_host(hay, True, devices=devices)
        assert got.shape == want.shape and (got == want).all(), devices
    assert int((want[:, 1] - want[:, 0]).max()) == 5800
    # chains that only merge at a rare unit: period-3 matches, the entry decides the phase up to the next 'b'
    hay2 = np.full(40000, a, np.uint16)
    hay2[[13001, 26002, 39000]] = b
    kws2 = [np.full(3, a, np.uint16)]
    want2 = Oracle(FAM_LONGEST, kws2).match(hay2)
    auto2 = Automaton(N.MODE_LONGEST, kws2, True)
    for devices in ([0, 0, 0], [0, 0, 0, 0, 0, 0, 0]):
        got2 = auto2.match_host(hay2, False, devices=devices)
        assert got2.shape == want2[:, :2].shape and (got2 == want2[:, :2]).all(), devices


def test_multi_shortest_state_passes_through_shares_without_matches():
    """Shortest: the restart position travels through shares that report nothing, and a match that ends just inside a
    share forbids the ones that overlap it."""
    a, b, c = ord("a"), ord("b"), ord("c")
    hay = np.full(9000, c, np.uint16)
    kws = [np.array([a, b, a], np.uint16), np.array([b, a, b], np.uint16)]
    for at in (2990, 2998, 5996, 5999, 6001):
        hay[at:at + 8] = [a, b, a, b, a, b, a, b]
    want = Oracle(FAM_SHORTEST, kws).match(hay)
    auto = Automaton(N.MODE_SHORTEST, kws, True)
    for devices in ([0, 0, 0], [0, 0, 0, 0, 0, 0]):
        got = auto.match_host(hay, True, devices=devices)
        assert got.shape == want.shape and (got == want).all(), devices


def _shards_of(hay, k, left, right):
    """Contiguous shares of a host haystack as device-resident shards (what a multi-GPU job holds): [pad | own | right halo],
    the owned range 16-byte aligned."""
    import torch
    n = int(hay.size)
    pad = (left + 7) // 8 * 8
    bufs, shards = [], []
    for i in range(k):
        lo = 0 if i == 0 else (i * n // k) & ~7
        hi = n if i == k - 1 else ((i + 1) * n // k) & ~7
        v0 = lo - pad if lo > pad else 0
        v1 = min(n, hi + right)
        t = torch.from_numpy(hay[v0:v1].view(np.int16).copy()).cuda()
        bufs.append(t)
        shards.append(dict(d_hay=t.data_ptr(), n_units=v1 - v0, own=(lo - v0, hi - v0), text_begin=v0 == 0, text_end=v1 == n,
                           base=v0))
    return bufs, shards


def _halos(auto):
    m = auto.info()["max_keyword_len"]
    if auto.mode in (N.MODE_WHOLEWORD, N.MODE_WWLONGEST):
        return 1, m + 1
    if auto.mode == N.MODE_LONGEST:
        return 0, max(m - 1, 0)
    return max(m - 1, 0), 0


def _gathered_records(gbufs, k, gcap, cols, shards):
    """Every device's gather buffer -> the record list with global positions; all devices must hold the same."""
    slot_words = N.gather_slot_bytes(gcap, cols * 4) // 4
    out = []
    for g in gbufs:
        h = g.cpu().numpy()
        parts = []
        for j in range(k):
            s = h[j * slot_words:(j + 1) * slot_words]
            cnt = int(s[:2].view(np.int64)[0])
            assert s[2] == 0  # (redone flag)
            r = s[4:4 + cnt * cols].reshape(cnt, cols).astype(np.int64)
            r[:, :2] += shards[j]["base"]
            parts.append(r)
        out.append(np.concatenate(parts))
    for o in out[1:]:
        assert o.shape == out[0].shape and (o == out[0]).all()
    return out[0]


@pytest.mark.parametrize("family", FAMILIES)
def test_device_resident_shards_allgather_by_peer_copies(family):
    """acgpu_match_device_allgather over a device list that names the one GPU three times: every share scans its resident
    shard into its slot of its own gather buffer, the slots travel to the other buffers (peer-copy transport), and every
    buffer then holds the oracle's records for the whole text.  Too small a gather capacity: ACGPU_E_OVERFLOW, exact counts."""
    import torch
    n = 500003
    auto, ids, hay, want = family_case(family, n, seed=1)
    cols = 3 if ids else 2
    k = 3
    left, right = _halos(auto)
    bufs, shards = _shards_of(hay, k, left, right)
    comm = Comm([0] * k, N.TRANSPORT_AUTO)
    assert comm.transport == N.TRANSPORT_PEER
    gcap = len(want)  # (every share's records fit)
    slot = N.gather_slot_bytes(gcap, cols * 4)
    gbufs = [torch.zeros(k * slot // 4, dtype=torch.int32, device="cuda") for _ in range(k)]
    torch.cuda.synchronize()
    rc, counts, exits, prof = comm.match_device_allgather(auto, shards, ids, [g.data_ptr() for g in gbufs], gcap, profile=True)
    assert rc == N.OK and sum(counts) == len(want)
    got = _gathered_records(gbufs, k, gcap, cols, shards)
    assert got.shape == want.shape and (got == want).all()
    assert all(p["scan_kernel"] for p in prof)
    if auto.mode in (N.MODE_LONGEST, N.MODE_WWLONGEST):  # the true scan's exits: at or behind each share's end
        assert all(e >= s["own"][1] for e, s in zip(exits, shards))
    # capacity protocol
    small = max(counts) - 1
    slot2 = N.gather_slot_bytes(small, cols * 4)
    g2 = [torch.zeros(k * slot2 // 4, dtype=torch.int32, device="cuda") for _ in range(k)]
    torch.cuda.synchronize()
    rc2, counts2, _, _ = comm.match_device_allgather(auto, shards, ids, [g.data_ptr() for g in g2], small)
    assert rc2 == N.E_OVERFLOW and counts2 == counts
    comm.close()


@pytest.mark.parametrize("family", ["ac", "wholeword", "longest"])
def test_device_resident_allgather_over_rccl_single_process_communicator(family):
    """The RCCL transport: ncclCommInitAll over the device list in THIS process, one ncclAllGather per device inside a group,
    in place, on the scan's stream.  A one-GPU box has a world of one -- the communicator, the group call and the kernel
    RCCL launches behind the scan are real; the boxes with more GPUs run the same code over all of them."""
    import torch
    ndev = torch.cuda.device_count()
    devices = list(range(ndev))
    n = 400000
    auto, ids, hay, want = family_case(family, n, seed=2)
    cols = 3 if ids else 2
    left, right = _halos(auto)
    k = ndev
    pad = (left + 7) // 8 * 8
    bufs, shards = [], []
    for i in range(k):
        lo = 0 if i == 0 else (i * n // k) & ~7
        hi = n if i == k - 1 else ((i + 1) * n // k) & ~7
        v0 = lo - pad if lo > pad else 0
        v1 = min(n, hi + right)
        t = torch.from_numpy(hay[v0:v1].view(np.int16).copy()).to("cuda:%d" % devices[i])
        bufs.append(t)
        shards.append(dict(d_hay=t.data_ptr(), n_units=v1 - v0, own=(lo - v0, hi - v0), text_begin=v0 == 0, text_end=v1 == n, base=v0))
    comm = Comm(devices, N.TRANSPORT_RCCL)
    assert comm.transport == N.TRANSPORT_RCCL
    gcap = len(want) + 8
    slot = N.gather_slot_bytes(gcap, cols * 4)
    gbufs = [torch.zeros(k * slot // 4, dtype=torch.int32, device="cuda:%d" % d) for d in devices]
    for d in devices:
        torch.cuda.synchronize(d)
    for _ in range(3):  # (several steps through one communicator)
        rc, counts, _, _ = comm.match_device_allgather(auto, shards, ids, [g.data_ptr() for g in gbufs], gcap)
        assert rc == N.OK and sum(counts) == len(want), (rc, N.lib().acgpu_last_rccl_error())
    got = _gathered_records(gbufs, k, gcap, cols, shards)
    assert got.shape == want.shape and (got == want).all()
    comm.close()
    with pytest.raises(N.AcgpuError):  # RCCL wants one rank per device
        Comm([0, 0], N.TRANSPORT_RCCL)


def test_allgather_redoes_the_gather_when_a_scan_had_to_be_redone():
    """A scratch slice that fills up makes acgpu_match_device_end redo the scan -- behind the gather that was already
    enqueued.  The driver sees that and gathers once more: all buffers hold the redone records."""
    import torch
    words = ["ab", "abc", "b", "cab"]
    auto = Automaton(N.MODE_WHOLEWORD, words, True, word_chars=WORD)
    n = 1 << 22
    rng = np.random.default_rng(3)
    hay = np.full(n, ord(" "), dtype=np.uint16)
    text = " ".join(words[i] for i in rng.integers(0, len(words), 1 << 16))
    head = np.array([ord(c) for c in text], dtype=np.uint16)
    hay[: head.size] = head  # all words in the first workgroup's share of share 0
    want = Oracle(FAM_WHOLEWORD, words, word_chars=WORD).match(hay)
    N.set_tunable("tile_debug", 134217728)  # (record slots from the scratch slices: the form that can overflow a slice)
    k = 2
    bufs, shards = _shards_of(hay, k, 1, auto.info()["max_keyword_len"] + 1)
    comm = Comm([0] * k, N.TRANSPORT_PEER)
    gcap = len(want)
    slot = N.gather_slot_bytes(gcap, 12)
    gbufs = [torch.zeros(k * slot // 4, dtype=torch.int32, device="cuda") for _ in range(k)]
    torch.cuda.synchronize()
    rc, counts, _, _ = comm.match_device_allgather(auto, shards, True, [g.data_ptr() for g in gbufs], gcap)
    assert rc == N.OK and counts == [len(want), 0]
    got = _gathered_records(gbufs, k, gcap, 3, shards)
    assert (got == want).all()
    comm.close()


def test_match_u16_multi_cuts_only_shares_worth_their_fixed_costs():
    """The product's rule (tunable multi_min_share at its default): a text below 2^23 units under a device list of two is ONE
    call on the first device -- no host thread, staging ring or extra synchronisation per share -- and gives the oracle's records;
    2^23 units and more are cut."""
    N.set_tunable("multi_min_share", 1 << 22)
    auto, with_ids, hay, want = family_case("ac", (1 << 22) + 4096)
    got = auto.match_host(hay, with_ids, devices=[0, 0])
    assert got.shape == want.shape and (got == want).all()
    auto, with_ids, hay, want = family_case("ac", (1 << 23) + 4096)
    got = auto.match_host(hay, with_ids, devices=[0, 0])
    assert got.shape == want.shape and (got == want).all()
