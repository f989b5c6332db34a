"""GPU parity tests of the tile kernel's fused tail (csrc/acgpu_tile.hip, TileLaunch::fused_tail): the scan's workgroups, numbered in
the order in which they start, put their own scratch slices in order behind the counts of the workgroups before them and the last
one reports the call's result -- no finalize launch.  Every case goes through the C ABI and is compared with the CPU oracle
(S/AhoCorasickSet.java:193-252 restated in oracle/ac_oracle.c), record for record and in the reference's order."""
import numpy as np
import pytest

from ahocorasick_amd import _native as N
from ahocorasick_amd import synth
from ahocorasick_amd.strings import Automaton
from oracle.oracle import FAM_AC, Oracle
from tests.helpers import oracle_parallel

pytestmark = pytest.mark.gpu

KNOBS = [("force_kernel", 0), ("region_units", 0), ("tile_debug", 0), ("all_form", 0), ("tile_form", 0)]


@pytest.fixture(autouse=True)
def _reset_tunables():
    yield
    for k, v in KNOBS:
        N.set_tunable(k, v)


def _dev(a, d_hay, n, with_ids, cap, **kw):
    import torch
    d_out = torch.full((max(cap, 1), 3 if with_ids else 2), -7, dtype=torch.int32, device="cuda")
    n_out, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, with_ids, d_out.data_ptr(), cap,
                                        stream=torch.cuda.current_stream().cuda_stream, **kw)
    return d_out, n_out, rc, prof


@pytest.fixture(scope="module")
def case():
    """600 keywords of 4-9 letters over a-z (a selective 4-gram filter: the tile kernel), 3 M + 5 units of text"""
    import torch
    kws = synth.random_keywords(31, 600, 4, 9)
    # plant matches densely in a few places (a burst fills a ring) and thinly everywhere else
    n = 3 * (1 << 20) + 5
    hay = synth.haystack(4711, n).copy()
    rng = np.random.default_rng(5)
    for p in rng.integers(0, n - 16, 30000).tolist():
        k = kws[int(rng.integers(0, len(kws)))]
        hay[p:p + len(k)] = k
    burst = np.concatenate([kws[i % len(kws)] for i in range(4000)])
    hay[1_500_000:1_500_000 + burst.size] = burst
    a = Automaton(N.MODE_ALL, kws, True)
    want = oracle_parallel(Oracle(FAM_AC, kws), hay, "ac", a.info()["max_keyword_len"], cap_per_unit=0.3)
    d_hay = torch.from_numpy(hay.view(np.int16)).cuda()
    return dict(kws=kws, hay=hay, a=a, want=want, d_hay=d_hay, n=n)


def test_fused_tail_equals_the_oracle_at_every_region_size(case):
    a, want, n = case["a"], case["want"], case["n"]
    N.set_tunable("force_kernel", 2)
    for ru in (0, 2048, 6144, 65536):
        N.set_tunable("region_units", ru)
        for with_ids in (True, False):
            d_out, n_out, rc, prof = _dev(a, case["d_hay"], n, with_ids, len(want) + 100, profile=True)
            assert rc == N.OK and n_out == len(want), (ru, with_ids, n_out, len(want))
            got = d_out[:n_out].cpu().numpy()
            assert (got == (want if with_ids else want[:, :2])).all(), (ru, with_ids)
            assert (d_out[n_out:].cpu().numpy() == -7).all()  # nothing behind the last record was touched
            assert prof["scan_kernel"].startswith("k_ac_tile") and prof["finalize_ms"] == 0.0 and prof["scan_ms"] > 0


def test_fused_tail_and_the_finalize_launch_give_the_same_records(case):
    a, want, n = case["a"], case["want"], case["n"]
    N.set_tunable("force_kernel", 2)
    N.set_tunable("tile_form", 1)
    d_out, n_out, rc, prof = _dev(a, case["d_hay"], n, True, len(want) + 100, profile=True)
    assert rc == N.OK and n_out == len(want) and (d_out[:n_out].cpu().numpy() == want).all()
    assert prof["finalize_ms"] > 0.0  # (the permute pass ran)


def test_capacity_smaller_than_the_matches_reports_overflow_with_the_exact_count(case):
    a, want, n = case["a"], case["want"], case["n"]
    N.set_tunable("force_kernel", 2)
    cap = len(want) // 3
    d_out, n_out, rc, _ = _dev(a, case["d_hay"], n, True, cap)
    assert rc == N.E_OVERFLOW and n_out == len(want)
    assert (d_out[:cap].cpu().numpy() == want[:cap]).all()  # what fits is the beginning of the list, in order


def test_shards_with_halos_and_calls_of_changing_size_on_one_pool(case):
    a, want, n, hay = case["a"], case["want"], case["n"], case["hay"]
    N.set_tunable("force_kernel", 2)
    N.set_tunable("region_units", 4096)
    cuts = [0, 700_001, 700_003, 2_222_222, n]  # (a two-unit shard among them)
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d_out, n_out, rc, _ = _dev(a, case["d_hay"], n, True, len(want) + 10, own=(lo, hi))
        assert rc == N.OK
        parts.append(d_out[:n_out].cpu().numpy())
    assert (np.concatenate(parts) == want).all()
    # a rank's buffer: left halo only, 16-byte aligned start, positions relative to the buffer
    halo = a.info()["max_keyword_len"] - 1
    lo, hi = cuts[3], cuts[4]
    base = (lo - halo) // 8 * 8
    sub = case["d_hay"][base:hi].clone()
    d_out, n_out, rc, _ = _dev(a, sub, hi - base, True, len(want) + 10, own=(lo - base, hi - base), text_begin=False)
    p = d_out[:n_out].cpu().numpy()
    p[:, :2] += base
    assert (p == parts[3]).all()
    # short texts (fewer regions than waves, a text shorter than a vector) after long ones: the state words were left clean
    for m in (5, 8, 4099, 65536 + 3, 1 << 20):
        w = Oracle(FAM_AC, case["kws"]).match(hay[:m], cap=1 << 20)
        d_out, n_out, rc, _ = _dev(a, case["d_hay"], m, True, len(w) + 10)
        assert rc == N.OK and n_out == len(w) and (d_out[:n_out].cpu().numpy() == w).all(), m


def test_tickets_in_flight_one_behind_the_other(case):
    import torch
    a, want, n = case["a"], case["want"], case["n"]
    N.set_tunable("force_kernel", 2)
    st = torch.cuda.current_stream().cuda_stream
    outs = [torch.empty((len(want) + 10, 3), dtype=torch.int32, device="cuda") for _ in range(3)]
    d_res = torch.zeros(3 * 4, dtype=torch.int64, device="cuda")  # three acgpu_device_result (16 bytes each, 16-byte aligned)
    tks = []
    for i, o in enumerate(outs):
        tk, rc = a.match_device_begin(case["d_hay"].data_ptr(), n, True, o.data_ptr(), len(want) + 10, stream=st, profile=(i == 1),
                                      d_result=d_res.data_ptr() + 32 * i)
        assert rc == N.OK
        tks.append(tk)
    for i, (tk, o) in enumerate(zip(tks, outs)):
        n_out, rc, prof = a.match_device_end(tk, profile=(i == 1))
        assert rc == N.OK and n_out == len(want) and (o[:n_out].cpu().numpy() == want).all()
        if i == 1:
            assert prof["scan_ms"] > 0 and prof["finalize_ms"] == 0.0
        assert int(d_res[4 * i].item()) == len(want)  # n_records, written by the call's only kernel
