"""world_size-2 (3, 4) gloo tests of the sharding plumbing on CPU for the three matcher families: halo exchange (left
and right), the Longest chain hop with speculation + window repair, and the all-gather of match buffers.
The native scan needs a GPU, so the scan step is stood in for by the CPU oracle here (test infrastructure); the
GPU tests cover the same shard/halo contract through acgpu_match_device (test_device_entry_and_shard_split_invariance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_per_rank, tmpdir, overlap, family):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ahocorasick_amd import synth
        from ahocorasick_amd._native import MODE_ALL, MODE_LONGEST, MODE_SHORTEST, MODE_WHOLEWORD
        from ahocorasick_amd.dist import ShardedMatcher
        from ahocorasick_amd.unicode_tables import default_word_chars
        from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_SHORTEST, FAM_WHOLEWORD, Oracle

        if family == "ww":
            table = np.array([ord(c) for c in "abc ,"], dtype=np.uint16)
            kws = synth.random_keywords(5, 60, 1, 6, table=table[:3])
            whole = synth.haystack(99, n_per_rank * world, table=table)
            orc = Oracle(FAM_WHOLEWORD, kws, word_chars=default_word_chars())
        else:
            table = synth.ALPHA_LOWER[:6] if family == "ac" else synth.ALPHA_LOWER[:2 if family == "longest" else 3]
            kws = synth.random_keywords(5, 200, 2, 9, table=table)
            whole = synth.haystack(99, n_per_rank * world, table=table)
            orc = Oracle({"ac": FAM_AC, "longest": FAM_LONGEST, "shortest": FAM_SHORTEST}[family], kws)
        max_len = max(len(k) for k in kws)

        # stand-ins with the contract of acgpu_match_device on one shard (include/acgpu.h, acgpu_shard)
        def scan_ac(view, ob, oe, text_begin, text_end, entry):
            lo = 0 if text_begin else ob - (max_len - 1)  # scan from the halo, keep matches whose last unit is owned
            r = orc.match(view[lo:oe])
            r[:, :2] += lo
            return r[(r[:, 1] - 1 >= ob)], -1

        def scan_ww(view, ob, oe, text_begin, text_end, entry):
            lo = max(ob - 1, 0)  # one unit of left context; a word belongs to the shard that owns its first unit
            hi = len(view) if text_end else min(len(view), oe + max_len + 1)
            r = orc.match(view[lo:hi])
            r[:, :2] += lo
            return r[(r[:, 0] >= ob) & (r[:, 0] < oe)], -1

        def scan_longest(view, ob, oe, text_begin, text_end, entry):
            if entry >= oe:
                return np.zeros((0, 3), np.int32), entry
            hi = len(view) if text_end else min(len(view), oe + max_len - 1)
            r = orc.match(view[entry:hi])  # the greedy chain entered at `entry`
            r[:, :2] += entry
            r = r[r[:, 0] < oe]
            ex = max(oe, int(r[-1, 1])) if len(r) else oe
            return r, ex

        def scan_shortest(view, ob, oe, text_begin, text_end, entry):
            # occurrences by increasing end (longest first at equal end), reported iff they start at/after the last
            # restart; this shard reports those whose last unit it owns.  The oracle run from `lo` with the restart
            # position forced by cutting the text there.
            entry = max(entry, 0)
            lo = 0 if text_begin else max(ob - (max_len - 1), 0)
            lo = max(lo, entry)
            r = orc.match(view[lo:oe])  # restarting at lo == "no match may start before lo"
            r[:, :2] += lo
            if lo < entry or not len(r):
                pass
            r = r[r[:, 1] - 1 >= ob]
            return r, (int(r[-1, 1]) if len(r) else entry)

        mode, scan_fn, halo, right = {"ac": (MODE_ALL, scan_ac, max_len - 1, 0),
                                      "shortest": (MODE_SHORTEST, scan_shortest, max_len - 1, 0),
                                      "ww": (MODE_WHOLEWORD, scan_ww, 1, max_len + 1),
                                      "longest": (MODE_LONGEST, scan_longest, 0, max_len - 1)}[family]
        m = ShardedMatcher(None, n_per_rank, with_ids=True, cap=16, scan_fn=scan_fn, halo=halo, right_halo=right,
                           overlap=overlap, mode=mode)
        m.chain_window = 16  # Longest: small first repair window so that the convergence test is exercised
        m.sb.own.copy_(torch.from_numpy(whole[rank * n_per_rank:(rank + 1) * n_per_rank].view(np.int16)))
        for _ in range(3):  # several steps: exercises the double-buffered, overlapped all-gather
            r = m.step()
        m.finish()
        got = m.global_records().numpy()
        want = orc.match(whole).astype(np.int64)
        assert r["n_total"] == len(want)
        assert got.shape == want.shape and (got == want).all()
        # halos really came from the neighbours
        if rank > 0 and halo:
            assert (m.sb.halo_view().numpy().view(np.uint16) == whole[rank * n_per_rank - halo:rank * n_per_rank]).all()
        if rank + 1 < world and right:
            assert (m.sb.right_view().numpy().view(np.uint16) == whole[(rank + 1) * n_per_rank:(rank + 1) * n_per_rank + right]).all()
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok %d" % m.chain_repairs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap,family", [(2, False, "ac"), (3, False, "ac"), (2, True, "ac"), (2, False, "ww"),
                                                  (3, True, "ww"), (2, False, "longest"), (3, False, "longest"),
                                                  (4, True, "longest"), (2, False, "shortest"), (4, False, "shortest")])
def test_sharded_match_equals_whole_text(world, overlap, family, tmp_path):
    port = _free_port()
    n_per_rank = 4001 if family not in ("longest", "shortest") else 1003
    mp.spawn(_worker, args=(world, port, n_per_rank, str(tmp_path), overlap, family), nprocs=world, join=True)
    oks = [(tmp_path / ("ok%d" % r)) for r in range(world)]
    assert all(p.exists() for p in oks)
    if family in ("longest", "shortest"):  # at least one shard boundary falls inside a match, so the repair path ran
        assert sum(int(p.read_text().split()[1]) for p in oks) > 0


def test_bench_launcher_starts_ranks_without_touching_the_gpu():
    """bench.py --gpus 2 as a plain python call: the parent must spawn torch.distributed.run children (here, without a GPU,
    they fail loudly -- there is no CPU matching path) and hand their exit code on."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--units-log2", "16", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = p.stdout.decode(errors="replace")
    import torch
    if torch.cuda.device_count() == 0:
        assert p.returncode != 0 and "bench.py needs a GPU" in text, text[-2000:]
    else:
        assert p.returncode == 0, text[-2000:]
