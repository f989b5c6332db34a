"""world_size-2 (and 3) gloo tests of the sharding plumbing on CPU: halo exchange + all-gather of match buffers.
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


def _worker(rank, world, port, n_per_rank, tmpdir, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ahocorasick_amd import synth
        from ahocorasick_amd.dist import ShardedMatcher
        from oracle.oracle import FAM_AC, Oracle

        kws = synth.random_keywords(5, 200, 2, 9, table=synth.ALPHA_LOWER[:6])
        halo = max(len(k) for k in kws) - 1
        whole = synth.haystack(99, n_per_rank * world, table=synth.ALPHA_LOWER[:6])
        orc = Oracle(FAM_AC, kws)

        def scan_fn(buf, own_begin, own_end, text_begin):
            # what acgpu_match_device does on one shard: scan from the halo, keep matches whose last unit is owned
            lo = 0 if text_begin else own_begin - halo
            r = orc.match(buf[lo:own_end])
            r[:, :2] += lo
            return r[(r[:, 1] - 1 >= own_begin)]

        m = ShardedMatcher(None, n_per_rank, with_ids=True, cap=16, scan_fn=scan_fn, halo=halo, overlap=overlap)
        m.sb.own.copy_(torch.from_numpy(whole[rank * n_per_rank:(rank + 1) * n_per_rank].view(np.int16)))
        for _ in range(3):  # several steps: exercises the double-buffered, overlapped all-gather
            r = m.step()
        m.finish()
        got = m.global_records().numpy()
        want = orc.match(whole).astype(np.int64)
        assert r["n_total"] == len(want)
        assert got.shape == want.shape and (got == want).all()
        # halo really came from the left neighbour
        if rank > 0:
            assert (m.sb.halo_view().numpy().view(np.uint16) == whole[rank * n_per_rank - halo:rank * n_per_rank]).all()
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (2, True)])
def test_sharded_match_equals_whole_text(world, overlap, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 4001, str(tmp_path), overlap), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))
