"""The sharded driver (ahocorasick_amd/dist.py) with the NATIVE scan in real, separate processes: 2 and 3 fresh child
processes on cuda:0, gloo with host-staged collectives (RCCL cannot put two ranks on one device).  Same ShardedMatcher
code as an N-GPU job -- halo exchange, acgpu_match_device[_begin/_end] on an acgpu_shard, chain hop with speculation and
window repair, one all-gather of header + records -- compared with the oracle on the WHOLE text in every rank."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(family, world, overlap, cap, tmp_path, n=50000, backend="gloo", force=False):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, family, str(world), str(r), port, str(n), str(tmp_path), str(int(overlap)),
                               str(cap), backend, "1" if force else "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]  # fresh children: nothing that touched the GPU is re-executed
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    oks = [(tmp_path / ("ok%d" % r)).read_text().split() for r in range(world)]
    return [(int(o[1]), int(o[2]), int(o[3])) for o in oks]


@pytest.mark.parametrize("family,world,overlap,cap", [
    ("ac", 2, False, 1 << 16), ("ac", 3, True, 1 << 16), ("ac", 2, True, 64),   # cap 64: the redo path (gather buffers grow)
    ("wholeword", 2, False, 1 << 16), ("wholeword", 3, True, 64),
    ("longest", 2, False, 1 << 16), ("longest", 3, True, 64),
    ("shortest", 2, True, 1 << 16), ("wwlongest", 3, False, 1 << 16), ("wwlongest", 2, True, 64)])
def test_sharded_matcher_native_scan_in_separate_processes(family, world, overlap, cap, tmp_path):
    res = _run(family, world, overlap, cap, tmp_path)
    assert all(r[2] > 0 for r in res)  # there were matches to get right
    if family in ("longest", "shortest") and world >= 3:
        assert sum(r[0] for r in res) > 0  # some shard boundary fell inside a match: the window repair ran on the device
    if cap == 64:
        assert all(r[1] > 0 for r in res)  # every rank went through the collective redo


def _n_gpus():
    import torch
    return torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)


@pytest.mark.parametrize("family,world,overlap,cap", [
    ("ac", 2, True, 1 << 16), ("ac", 2, True, 64), ("wholeword", 2, True, 1 << 16), ("longest", 2, False, 64),
    ("shortest", 2, True, 1 << 16), ("wwlongest", 2, False, 1 << 16)])
def test_sharded_matcher_over_rccl_one_rank_per_gpu(family, world, overlap, cap, tmp_path):
    """The same workers with backend "nccl" (RCCL): device-to-device halo exchange (batch_isend_irecv of byte views), the
    asynchronous all-gather of header + records left in flight under the next scan, the chain hop on device tensors.
    Needs one GPU per rank: skipped on a one-GPU box."""
    if _n_gpus() < world:
        pytest.skip("needs %d GPUs" % world)
    res = _run(family, world, overlap, cap, tmp_path, backend="nccl")
    assert all(r[2] > 0 for r in res)
    if cap == 64:
        assert all(r[1] > 0 for r in res)


@pytest.mark.parametrize("family,overlap,cap", [("ac", True, 1 << 16), ("ac", True, 64), ("wholeword", True, 1 << 16),
                                                 ("longest", False, 1 << 16), ("shortest", False, 1 << 16), ("wwlongest", False, 64)])
def test_sharded_matcher_over_rccl_world_of_one_keeps_its_collectives(family, overlap, cap, tmp_path):
    """RCCL on the one-GPU box: a process group of ONE rank with backend "nccl", and a ShardedMatcher told to keep every
    collective it would run among N ranks (force_collectives) -- the gather buffers [header | records], the asynchronous
    all_gather_into_tensor left in flight under the next scan, the header read-back on the side stream, the chain families'
    all-gathers of exits, the collective redo with larger buffers (cap 64), and a batch_isend_irecv of byte views (the halo
    exchange's transport).  Everything an 8-GPU job executes on RCCL has then executed on RCCL once."""
    res = _run(family, 1, overlap, cap, tmp_path, n=150000, backend="nccl", force=True)
    assert res[0][2] > 0
    if cap == 64:
        assert res[0][1] > 0  # the collective redo ran


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` started plainly (no torch.distributed.run around it, the way the driver may start it):
    the parent spawns the ranks before it touches the GPU and relays rank 0's single JSON line."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--units-log2", "24", "--cpu-sample-log2", "20"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["verified"] is True
    assert out["config"]["parallelism"].startswith("shard2+halo") and out["config"]["parallelism"].endswith("allgather/gloo")
    assert [r["rank"] for r in out["roofline"]["per_rank"]] == [0, 1]
    assert all(r["kernel_ms"] > 0 and r["matches"] > 0 for r in out["roofline"]["per_rank"])
