#!/usr/bin/env python3
"""One rank of a sharded match with the NATIVE scan (test_dist_gpu.py starts 2-3 of these as fresh child processes, all
on cuda:0).  RCCL cannot put two ranks on one device, so the process group is gloo and ahocorasick_amd.dist stages its
collectives through host memory; everything else -- ShardedMatcher, acgpu_match_device[_begin/_end] with an acgpu_shard
(owned range, halos, chain entry/exit), the gather-buffer header written by the scan's last kernel -- is the code an
N-GPU job runs.  Test infrastructure: compares with the CPU oracle on the whole text.

usage: dist_gpu_worker.py FAMILY WORLD RANK PORT N_PER_RANK OUTDIR OVERLAP CAP [BACKEND [FORCE]]
(BACKEND nccl: one rank per GPU over RCCL -- only where the box has that many GPUs; FORCE 1: ShardedMatcher keeps its
collectives in a world of one -- a one-GPU box then drives the whole multi-rank step over a real RCCL communicator)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(family, n_total, variant=0):
    """(automaton, oracle, whole text) -- deterministic, so every rank builds the same."""
    from ahocorasick_amd import _native as N
    from ahocorasick_amd import synth
    from ahocorasick_amd.strings import Automaton
    from ahocorasick_amd.unicode_tables import default_word_chars, java_lower_table
    from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_SHORTEST, FAM_WHOLEWORD, Oracle
    if family == "wholeword":
        table = np.array([ord(c) for c in "abcdE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        kws = synth.random_keywords(31, 300, 1, 6, table=table[:5])
        whole = synth.haystack(41 + variant, n_total, table=table)
        return (Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=default_word_chars()),
                Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=java_lower_table(), word_chars=default_word_chars()), whole)
    if family == "wwlongest":
        from oracle.oracle import FAM_WWLONGEST
        table = np.array([ord(c) for c in "abcE -,"] + [0x00E9, 0x00C9], dtype=np.uint16)
        rng = np.random.default_rng(35)
        words = synth.random_keywords(35, 120, 1, 5, table=table[:4])
        sp = np.array([32], dtype=np.uint16)
        kws = list(words[:60]) + [np.concatenate([words[int(i)], sp, words[int(j)]]) for i, j in rng.integers(0, 120, (80, 2))]
        whole = synth.haystack(45 + variant, n_total, table=table)
        return (Automaton(N.MODE_WWLONGEST, kws, False, word_chars=default_word_chars()),
                Oracle(FAM_WWLONGEST, kws, case_sensitive=False, lower=java_lower_table(), word_chars=default_word_chars()), whole)
    if family == "longest":
        kws = synth.random_keywords(32, 300, 2, 40, table=synth.ALPHA_LOWER[:2])
        whole = synth.haystack(42 + variant, n_total, table=synth.ALPHA_LOWER[:2])
        return Automaton(N.MODE_LONGEST, kws, True), Oracle(FAM_LONGEST, kws), whole
    if family == "shortest":
        kws = synth.random_keywords(34, 300, 2, 30, table=synth.ALPHA_LOWER[:3])
        whole = synth.haystack(44 + variant, n_total, table=synth.ALPHA_LOWER[:3])
        return Automaton(N.MODE_SHORTEST, kws, True), Oracle(FAM_SHORTEST, kws), whole
    kws = synth.random_keywords(33, 500, 2, 11, table=synth.ALPHA_LOWER[:8])
    whole = synth.haystack(43 + variant, n_total, table=synth.ALPHA_LOWER[:8])
    return Automaton(N.MODE_ALL, kws, True), Oracle(FAM_AC, kws), whole


def main():
    family, world, rank, port, n, outdir, overlap, cap = sys.argv[1:9]
    backend = sys.argv[9] if len(sys.argv) > 9 else "gloo"
    force = len(sys.argv) > 10 and sys.argv[10] == "1"
    world, rank, n, overlap, cap = int(world), int(rank), int(n), int(overlap), int(cap)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ahocorasick_amd.dist import ShardedMatcher
        auto, orc, whole = case(family, n * world)
        m = ShardedMatcher(auto, n, with_ids=True, cap=cap, overlap=bool(overlap), force_collectives=force)
        assert m.scan_fn is None and m.device.type == "cuda"
        if force and backend == "nccl":
            # the halo exchange's transport, which a world of one never needs: byte views through batch_isend_irecv on RCCL
            # (rank 0 sends its tail to itself)
            src = torch.arange(4096, dtype=torch.int16, device="cuda")
            dst = torch.zeros(2 * 4096, dtype=torch.uint8, device="cuda")
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src.view(torch.uint8), rank), dist.P2POp(dist.irecv, dst, rank)]):
                w.wait()
            torch.cuda.synchronize()
            assert (dst.view(torch.int16) == src).all()
        m.chain_window = 32
        m.load(whole[rank * n:(rank + 1) * n])
        repairs = 0
        for _ in range(3):
            m.step()
            repairs += m.chain_repairs
        m.finish()
        want = orc.match(whole).astype(np.int64)
        got = m.global_records().cpu().numpy()
        assert got.shape == want.shape and (got == want).all(), (family, rank, got.shape, want.shape)
        redone = m.redone_steps
        # a second haystack through the same matcher: the halos are exchanged again, the gather buffers have adapted
        _, orc2, whole2 = case(family, n * world, variant=7)
        m.load(whole2[rank * n:(rank + 1) * n])
        for _ in range(2):
            m.step()
        m.finish()
        want2 = orc2.match(whole2).astype(np.int64)
        got2 = m.global_records().cpu().numpy()
        assert got2.shape == want2.shape and (got2 == want2).all(), (family, rank, "second haystack")
        with open(os.path.join(outdir, "ok%d" % rank), "w") as f:
            f.write("ok %d %d %d %d" % (repairs, redone, len(want), m.host_syncs))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
