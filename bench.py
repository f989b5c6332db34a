#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: haystack MB/s (+ matches/s) of AhoCorasick*.match().

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  N=1 : BASELINE.json config 2 -- AhoCorasickMap, 10k keywords (len 4..12, a-z), 1 GiB (2^29 UTF-16 units)
        haystack, records (start, end, keyword_id) delivered in reference order on the device.
  N>1 : config 3 -- AhoCorasickSet, same dictionary, 2^29 units PER GPU (weak scaling, shard g = stream 2003+g),
        each rank scans its shard with a (max_keyword_len-1) left halo received from rank g-1, then the per-shard
        match buffers are all-gathered over RCCL/xGMI (counts first, then padded record buffers).
  --config C4 | C5 : the sibling matchers (LongestMatchSet / WholeWordMatchMap case-insensitive) at BASELINE's sizes,
        same sharding driver; not the headline line.
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--units-log2", type=int, default=29, help="haystack units per GPU (default 2^29 = 1 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log2", type=int, default=28)
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE config (default: C2 at N=1, C3 at N>1 -- the configs the metric is quoted on; "
                         "C4 = LongestMatchSet, C5 = WholeWordMatchMap case-insensitive are the sibling matchers)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from ahocorasick_amd import _native as N
    from ahocorasick_amd import synth
    from ahocorasick_amd.dist import ShardedMatcher
    from ahocorasick_amd.strings import Automaton

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU matching path)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    multi = world > 1
    cfg_name = args.config or ("C3" if multi else "C2")
    if cfg_name == "C5" and args.units_log2 == 29:
        args.units_log2 = 28  # config 5 is 2^31 units over 8 GPUs
    n_units = 1 << args.units_log2
    # C2/C5 = *Map (12-byte records), C3/C4 = *Set (8-byte records)
    with_ids = cfg_name in ("C2", "C5")
    rec_bytes = 12 if with_ids else 8
    cfg = synth.CONFIGS[cfg_name]
    kws = synth.config_keywords(cfg_name)
    t0 = time.time()
    if cfg_name == "C4":
        auto = Automaton(N.MODE_LONGEST, kws, True)
    elif cfg_name == "C5":
        from ahocorasick_amd.unicode_tables import default_word_chars
        auto = Automaton(N.MODE_WHOLEWORD, kws, False, word_chars=default_word_chars())
    else:
        auto = Automaton(N.MODE_ALL, kws, True)
    build_s = time.time() - t0
    info = auto.info()
    halo = info["max_keyword_len"] - 1

    # synthetic shard, generated in place on the device
    seed = cfg["hay_seed"] + (rank if multi else 0)
    # N>1: the all-gather of step k overlaps the scan of step k+1 (double-buffered record buffers); every gather is
    # complete before the timed region ends (matcher.finish() + synchronize)
    cap = {"C4": n_units // 2, "C5": n_units // 8}.get(cfg_name, max(1 << 16, n_units // 128))
    matcher = ShardedMatcher(auto, n_units, with_ids=with_ids, cap=cap, overlap=True)
    if cfg_name == "C5":
        # the token stream is generated on the host: one 2^22-unit block of it, repeated (tests do the same)
        blk = min(n_units, 1 << 22)
        block = synth.mixed_script_haystack(seed, blk, kws, swapcase_tbl=synth.swapcase_table())
        matcher.sb.own.copy_(torch.from_numpy(block.view(np.int16)).cuda().repeat(n_units // blk))
    else:
        tab = np.ascontiguousarray(synth.ALPHA_AB_75 if cfg_name == "C4" else synth.ALPHA_LOWER)
        N.check(N.lib().acgpu_synth_fill(matcher.own_ptr(), n_units, 0, seed, tab.ctypes.data_as(ctypes.c_void_p), len(tab),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "synth_fill")
    torch.cuda.synchronize()

    def step(profile=False):
        return matcher.step(profile=profile)

    for _ in range(args.warmup):
        step()
    matcher.finish()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    # N=1 pipelines the calls (step k+1 is enqueued before the count of step k is read back): step() then returns the
    # PREVIOUS step's result and finish() the last one; all K steps are complete before the clock stops
    results = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        results.append(step(profile=True))
    results.append(matcher.finish())
    torch.cuda.synchronize()
    results = [r for r in results if r is not None]
    assert len(results) == args.steps, (len(results), args.steps)
    scan_ms = [r["scan_ms"] for r in results]
    fin_ms = [r["finalize_ms"] for r in results]
    n_matches_local, n_matches_total = results[-1]["n_local"], results[-1]["n_total"]
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_units = n_units * world
    ms_per_step = elapsed / args.steps * 1e3
    mb_per_s = total_units * 2 / (elapsed / args.steps) / 1e6
    kernel_ms = float(np.mean(scan_ms))
    alg_bytes = 2 * n_units + rec_bytes * n_matches_local  # per launch of the dominant kernel (one rank's shard)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    out = {
        "metric": "haystack MB/s (UTF-16 bytes scanned per second, records delivered in reference order)",
        "value": round(mb_per_s, 1),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16",
        "data": "synthetic",
        "matches_per_s": round(n_matches_total / (elapsed / args.steps), 1),
        "config": {
            "workload": {
                "C2": "BASELINE config 2: AhoCorasickMap, 10k keywords (len 4-12, a-z), 2^%d UTF-16 units (1 GiB at 29)",
                "C3": "BASELINE config 3: AhoCorasickSet, 10k keywords, 2^%d units per GPU, halo + all-gather of match buffers",
                "C4": "BASELINE config 4: LongestMatchSet, 50k prefix-closed keywords over {a,b} (max len 1000), 2^%d units P(a)=0.75",
                "C5": "BASELINE config 5: WholeWordMatchMap case-insensitive, 100k mixed-script words, 2^%d units per GPU",
            }[cfg_name] % args.units_log2,
            "keywords": len(kws), "states": info["n_states"], "classes": info["n_classes"],
            "table": ("dense u%d" % (8 * info["entry_bytes"])) if info["dense"] else "hashed",
            "lds_states": info["lds_states"], "units_per_gpu": n_units, "matches_per_gpu": n_matches_local,
            "matches_total": n_matches_total, "record_bytes": rec_bytes, "build_s": round(build_s, 3),
            "parallelism": "shard%d+halo(%d,%d)+allgather" % (world, matcher.sb.halo, matcher.sb.right) if multi else "single",
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
            "kernel": matcher.last_kernel, "kernel_ms": round(kernel_ms, 4), "finalize_ms": round(float(np.mean(fin_ms)), 4),
            "algorithmic_bytes": alg_bytes,
        },
    }

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (counters cannot be read from inside
    # this process); the committed measurement is attached when it is for this kernel and this size
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            tr = json.load(f)
        if tr["kernel"] == matcher.last_kernel and tr["units_per_gpu"] == n_units:
            out["roofline"]["traffic"] = tr["traffic_bytes"]
            out["roofline"]["traffic_source"] = tr["source"]
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0 and not multi and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg_name, kws, matcher, min(n_units, 1 << args.cpu_sample_log2))
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cfg_name, kws, matcher, sample_units):
    """The reference-shaped CPU restatement (oracle/ac_oracle.c, kind "port"), single thread like the reference,
    no-op listener (R/README.md:144), on a bounded prefix of the same haystack."""
    from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, Oracle
    if cfg_name == "C4":
        o = Oracle(FAM_LONGEST, kws)
    elif cfg_name == "C5":
        from ahocorasick_amd.unicode_tables import default_word_chars, java_lower_table
        o = Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=java_lower_table(), word_chars=default_word_chars())
    else:
        o = Oracle(FAM_AC, kws)
    hay = matcher.own_units_host(sample_units)
    o.count(hay[:1 << 20])  # warm-up
    t0 = time.perf_counter()
    n = o.count(hay)
    dt = time.perf_counter() - t0
    return {"value": round(hay.size * 2 / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first 2^%d units of the same haystack, %d matches, %.1f s, no-op listener" % (
                int(np.log2(hay.size)), n, dt),
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
