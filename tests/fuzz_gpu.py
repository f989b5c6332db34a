#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak test (development tool; the pytest suite holds the fixed regression cases).

Draws random dictionaries / haystacks / tunables / shard splits for all three matcher families and compares the
device records with oracle/ac_oracle.c bit for bit.  Usage: python tests/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root

import torch  # noqa: E402

from ahocorasick_amd import _native as N  # noqa: E402
from ahocorasick_amd.strings import Automaton  # noqa: E402
from ahocorasick_amd.unicode_tables import default_word_chars, java_lower_table  # noqa: E402
from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_SHORTEST, FAM_WHOLEWORD, FAM_WWLONGEST, Oracle  # noqa: E402

LOWER = java_lower_table()
WORD = default_word_chars()

ALPHABETS = [
    [ord(c) for c in "ab"],
    [ord(c) for c in "abc"],
    list(range(ord("a"), ord("z") + 1)),
    list(range(ord("a"), ord("z") + 1)) + list(range(ord("A"), ord("Z") + 1)),            # 52 classes: WIDE rows
    [ord(c) for c in "abcABC"] + [0x00E9, 0x00C9, 0x0130, 0x4E2D],                           # LUT classes
    list(range(0x4E00, 0x4E00 + 300)),                                                       # > 64 classes: DFA only
    [ord(c) for c in "ab -_.,9"] + [0x00E9, 0x00C9, 0x3002],                                 # separators (WholeWord)
    [ord(c) for c in "abcdefikABCDEFIK 0129-"] + [0x0130, 0x212A],                           # phrases: merged stretches, fold exceptions
]
DEFAULTS = {"chunk_units": 0, "lds_table_bytes": 127 * 1024, "force_sparse": 0, "force_kernel": 0, "region_units": 0,
            "rdense_budget_bytes": 256 << 20, "tile_debug": 0, "longest_form": 0, "all_form": 0, "tile_form": 0, "ww_block": 0, "ww_ramp_pm": -1}


def dev_match(a, d_hay, n, cap, with_ids=True, **kw):
    while True:
        d_out = torch.empty((max(cap, 1), 3 if with_ids else 2), dtype=torch.int32, device="cuda")
        n_out, rc, _, ex = a.match_device(d_hay.data_ptr(), n, with_ids, d_out.data_ptr(), cap, **kw)
        if rc == N.E_OVERFLOW:
            cap = n_out
            continue
        assert rc == N.OK, rc
        return d_out[:n_out].cpu().numpy(), ex


def one_case(rng, it):
    fam = int(rng.integers(0, 5))
    alpha = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))]
    cs = bool(rng.integers(0, 2))
    n_kw = int(rng.integers(1, 60))
    min_len = int(rng.integers(1, 6))
    max_len = min_len + int(rng.integers(0, 9))
    if fam == 2 and rng.integers(0, 4) == 0:
        max_len = min_len + int(rng.integers(9, 40))  # keywords beyond 16 units: k_ww_pp's 32-unit form, beyond 32: k_ww_tile
    if fam == 2:
        kw_alpha = [c for c in alpha if WORD[c]] or [ord("a")]
    else:
        kw_alpha = alpha
    kws = [np.array(rng.choice(kw_alpha, int(rng.integers(min_len, max_len + 1))), dtype=np.uint16) for _ in range(n_kw)]
    n = int(rng.choice([0, 1, 7, 63, 1000, 4097, 70001, 300007]))
    if fam == 1 and len(alpha) <= 3 and rng.integers(0, 2):
        # config 4's shape: all prefixes of a few long words (deep walks: root table, work lists, DEEP rows)
        kws = []
        for _ in range(int(rng.integers(1, 6))):
            w = np.array(rng.choice(alpha, int(rng.integers(5, 400))), dtype=np.uint16)
            kws += [w[:k].copy() for k in range(1, len(w) + 1)]
        n_kw, min_len, max_len = len(kws), 1, max(len(k) for k in kws)
    hay = np.array(rng.choice(alpha, n), dtype=np.uint16) if n else np.zeros(0, np.uint16)
    if n > 100 and rng.integers(0, 2):  # plant keywords so that deep matches occur
        for _ in range(int(rng.integers(1, 50))):
            k = kws[int(rng.integers(0, n_kw))]
            p = int(rng.integers(0, n - len(k)))
            hay[p:p + len(k)] = k
    knobs = dict(DEFAULTS)
    if rng.integers(0, 2):
        knobs["force_kernel"] = int(rng.integers(0, 4))
    if rng.integers(0, 3) == 0:
        knobs["chunk_units"] = int(rng.choice([8, 64, 1000]))
    if rng.integers(0, 3) == 0:
        knobs["lds_table_bytes"] = int(rng.choice([0, 1024, 96 * 1024, 127 * 1024]))
    if rng.integers(0, 4) == 0:
        knobs["force_sparse"] = 1
    if rng.integers(0, 3) == 0:
        knobs["region_units"] = int(rng.choice([4096, 8192]))
    if rng.integers(0, 4) == 0:
        knobs["rdense_budget_bytes"] = 0  # hashed reversed trie
    if rng.integers(0, 2):
        knobs["tile_debug"] = 4194304  # chain marking in one pass (Shortest, sparse Longest, WholeWordLongest) on small inputs too
    if rng.integers(0, 2):
        knobs["tile_debug"] |= 1 << 41  # short haystacks through the general path instead of the one-launch form
    if fam in (0, 3) and rng.integers(0, 2):
        knobs["all_form"] = int(rng.choice([6, 6, 4, 1]))  # k_ac_states for short texts too, whatever / only by what the pool's last call found / never
    if fam == 1 and rng.integers(0, 2):
        knobs["longest_form"] = int(rng.choice([4, 4, 5, 6, 12]))  # k_longest_bits / k_longest_follow for short texts too (5, 6: one of them never; 12: follow over small alphabets too)
    if rng.integers(0, 3) == 0:
        knobs["tile_form"] = int(rng.integers(1, 4))  # the ordering of the records in launches of their own (k_ac_tile / k_ww_pp / both) instead of the fused tails
    if fam == 2 and rng.integers(0, 3) == 0:
        knobs["ww_block"] = int(rng.choice([64, 320, 640, 896]))  # k_ww_pp's fused tail with workgroups of other sizes
    if fam == 2 and rng.integers(0, 3) == 0:
        knobs["ww_ramp_pm"] = int(rng.choice([0, 50, 400, 1000]))  # ... and spans that grow with the workgroup's number
    for k, v in knobs.items():
        N.set_tunable(k, v)
    mode = [N.MODE_ALL, N.MODE_LONGEST, N.MODE_WHOLEWORD, N.MODE_SHORTEST, N.MODE_WWLONGEST][fam]
    ofam = [FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, FAM_SHORTEST, FAM_WWLONGEST][fam]
    wc = WORD if fam in (2, 4) else None
    with_ids = bool(rng.integers(0, 3))  # one case in three runs the Set flavour (8-byte records)
    if fam in (2, 4) and rng.integers(0, 4) == 0:
        # a word-character table that is not fold-consistent: some lower-case letters of the alphabet stop being word
        # characters while their capitals stay (the reference's loops then differ in which lookups fold)
        wc = WORD.copy()
        for c in alpha:
            if LOWER[c] != c and rng.integers(0, 2):
                wc[LOWER[c]] = 0
        if fam == 2:  # WholeWord keywords are validated on their raw units
            ok = [c for c in alpha if wc[c]] or [ord("A")]
            kws = [np.array(rng.choice(ok, max(1, len(k))), dtype=np.uint16) for k in kws]
            wc[ord("A")] = 1
    no_pages = int(rng.integers(0, 4) == 0)  # builder knob: class tables in global memory instead of LDS pages
    N.set_tunable("no_class_pages", no_pages)
    # builder knobs of the word kernel: no perfect hash (the two-choice table behind the Bloom filter), other bucket sizes, no byte pages
    ww_build = {"ww_no_ph": int(rng.integers(0, 4) == 0), "ww_ph_lambda": int(rng.choice([0, 0, 1, 6, 300])), "ww_no_byte_pages": int(rng.integers(0, 3) == 0)}
    for k, v in ww_build.items():
        N.set_tunable(k, v)
    try:
        a = Automaton(mode, kws, cs, word_chars=wc)
    finally:
        N.set_tunable("no_class_pages", 0)
        for k in ww_build:
            N.set_tunable(k, 0)
    orc = Oracle(ofam, kws, case_sensitive=cs, lower=LOWER, word_chars=wc, map_flavour=(fam == 4 and with_ids))
    want = orc.match(hay)
    if fam in (2, 4) and n and rng.integers(0, 3) == 0:
        # match(Readable): the feeds' records are the Readable loop's (which folds in every lookup), under a random chunking
        from ahocorasick_amd.strings import Stream
        want_r = orc.match_readable(hay, int(rng.choice([1, 5, 1024])), positions=True)
        st = Stream(a, with_ids=True)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, int(rng.integers(0, 5)))]))
        parts = [st.feed(hay[lo:hi], final=(hi == n), cap=8) for lo, hi in zip(cuts[:-1], cuts[1:])]
        st.close()
        got_r = np.concatenate(parts) if parts else np.zeros((0, 3), np.int64)
        assert got_r.shape == want_r.shape and (got_r == want_r.astype(np.int64)).all(), ("stream", it, fam, cs, cuts)
    if n and rng.integers(0, 4) == 0:
        # the batch entry: the haystack cut into pieces, every piece its own haystack
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, int(rng.integers(1, 9)))]))
        pieces = [hay[lo:hi] for lo, hi in zip(cuts[:-1], cuts[1:])] + [np.zeros(0, np.uint16)]
        orc_b = orc if fam != 4 else Oracle(ofam, kws, case_sensitive=cs, lower=LOWER, word_chars=wc, map_flavour=True)  # (Map records)
        want_b = np.concatenate([np.concatenate([np.full((len(r), 1), i, np.int32), r], axis=1) for i, r in
                                 enumerate(orc_b.match(h) for h in pieces)])
        got_b = a.match_batch(pieces, True, cap=16)
        assert got_b.shape == want_b.shape and (got_b == want_b).all(), ("batch", it, fam, cs, cuts)
    if not with_ids:
        want = np.ascontiguousarray(want[:, :2])
    got = a.match_host(hay, with_ids, cap=64)
    desc = (it, fam, cs, with_ids, n_kw, min_len, max_len, n, len(alpha), knobs, a.info()["filter_k"], a.info()["tile_kernel"])
    assert got.shape == want.shape and (got == want).all(), ("host path", desc)
    if n >= 2048 and rng.integers(0, 3) == 0:
        # the multi-device entry: the one GPU named several times (shares, halos, speculative scans, window repairs)
        devs = [0] * int(rng.integers(2, 6))
        got_m = a.match_host(hay, with_ids, cap=64, devices=devs)
        assert got_m.shape == want.shape and (got_m == want).all(), ("multi-device", len(devs), desc)
    if n and rng.integers(0, 3) == 0 and (wc is None or a.info()["fold_consistent"]):
        # pipelined feeds (a feed returns the previous chunk's records): the String overload's records, under a random chunking
        from ahocorasick_amd.strings import Stream
        want_s = want if with_ids else orc.match(hay)
        st = Stream(a, with_ids=True, pipelined=True)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, int(rng.integers(0, 6)))]))
        parts = [st.feed(hay[lo:hi], final=(hi == n), cap=8) for lo, hi in zip(cuts[:-1], cuts[1:])]
        st.close()
        got_s = np.concatenate(parts) if parts else np.zeros((0, 3), np.int64)
        assert got_s.shape == want_s.shape and (got_s == want_s.astype(np.int64)).all(), ("pipelined stream", cuts, desc)
    fold_seq = wc is not None and not cs and a.info()["fold_consistent"] == 0 and (fam == 2 or not with_ids)
    if n >= 1000 and not fold_seq:  # (the loops that mix raw and folded lookups exist for the whole text only)
        # shards of the device-resident buffer
        d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(1, n, 2)]))
        parts, entry = [], 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            kw = dict(own=(lo, hi))
            if fam == 1:
                kw["chain_entry"] = max(entry, lo)
            if fam in (3, 4):
                kw["chain_entry"] = entry
            p, ex = dev_match(a, d_hay, n, 64, with_ids, **kw)
            entry = ex
            parts.append(p)
        cat = np.concatenate(parts) if parts else np.zeros((0, want.shape[1]), np.int32)
        assert cat.shape == want.shape and (cat == want).all(), ("shards", cuts, desc)
    return fam


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rng = np.random.default_rng(seed)
    t0 = time.time()
    counts = [0, 0, 0, 0, 0]
    it = 0
    t_said = t0
    while time.time() - t0 < budget:
        counts[one_case(rng, it)] += 1
        it += 1
        if time.time() - t_said > 30:  # a long soak keeps talking (a silent run is taken to be hung)
            t_said = time.time()
            print("  ... %d cases after %.0f s" % (it, t_said - t0), flush=True)
    print("fuzz ok: %d cases (AC %d, Longest %d, WholeWord %d, Shortest %d, WholeWordLongest %d) in %.0f s, seed %d" % (
        it, *counts, time.time() - t0, seed))


if __name__ == "__main__":
    main()
