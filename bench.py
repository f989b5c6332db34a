#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: haystack MB/s (+ matches/s) of AhoCorasick*.match().

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  N=1 : BASELINE.json config 2 -- AhoCorasickMap, 10k keywords (len 4..12, a-z), 1 GiB (2^29 UTF-16 units)
        haystack, records (start, end, keyword_id) delivered in reference order on the device.
  N>1 : config 3 -- AhoCorasickSet, same dictionary, 2^29 units PER GPU (weak scaling, shard g = stream 2003+g),
        each rank scans its shard with a (max_keyword_len-1) left halo received from rank g-1 (once: the haystack does not
        change between steps), then the per-shard match buffers are all-gathered over RCCL/xGMI (one all-gather of
        [count header | records] per step, left in flight under the next step's scan).
  --config C3 --gpus 1 : one rank's share of config 3 (Set records, R = 8) without the collective.
  --config C4 | C5 : the sibling matchers (LongestMatchSet / WholeWordMatchMap case-insensitive) at BASELINE's sizes,
        same sharding driver; not the headline line.
  --backend gloo : the N>1 path with host-staged collectives (several ranks on one GPU; what tests/test_dist_gpu.py runs).
Prints ONE JSON line (rank 0).  Outside the timed region the line is checked and annotated: the records of the last
step are compared with the CPU oracle on a prefix of the haystack ("verified"), the CPU restatement is timed on that
prefix ("cpu_baseline": pinned to one core, median of 5 runs after a warm-up), the tile kernels' access pattern is timed
as a pure read ("roofline.attainable"), and the host-buffer entry point acgpu_match_u16 is timed ("end_to_end").
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ATTAINABLE_GBPS = 6200.0  # the best pure read of a shard measured on the chip (roofline.attainable of the default run): what a floor is priced at


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--units-log2", type=int, default=29, help="haystack units per GPU (default 2^29 = 1 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip everything that runs the CPU oracle")
    ap.add_argument("--cpu-sample-log2", type=int, default=27)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1: nccl = RCCL over xGMI, one rank per GPU; gloo = host-staged collectives, ranks may share a GPU")
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5", "README"],
                    help="BASELINE config (default: C2 at N=1, C3 at N>1 -- the configs the metric is quoted on; "
                         "C4 = LongestMatchSet, C5 = WholeWordMatchMap case-insensitive are the sibling matchers)")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE host process drives all N GPUs through the C ABI (acgpu_comm_open + acgpu_match_device_allgather: "
                         "single-process RCCL ncclCommInitAll / peer copies) instead of one torch.distributed rank per GPU -- the "
                         "shape a JVM calls (INTEGRATION.md)")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device list (default 0..N-1; a device may "
                                                    "be named twice on a box with fewer GPUs: peer-copy transport)")
    ap.add_argument("--no-separate-launches", action="store_true",
                    help="skip the roofline.separate_launches diagnostic (a few extra steps with the ordering in launches of their own): "
                         "what the rocprofv3 --stats evidence is collected with, so that the kernel's average is of the timed form alone")
    ap.add_argument("--tunable", action="append", default=[], metavar="NAME=VALUE",
                    help="development: an acgpu_set_tunable knob for A/B on one box (e.g. tile_form=1: a finalize launch behind the scan); "
                         "recorded in config.tunables")
    args = ap.parse_args()

    if args.config == "README":
        return main_readme(args)
    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # plain `python bench.py --gpus N`: this process never touches the GPU

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
    n_dev = torch.cuda.device_count()
    assert args.backend == "gloo" or local_rank < n_dev, \
        "--backend nccl (RCCL) needs one GPU per rank: %d GPUs visible, local rank %d (use --backend gloo to share a GPU)" % (n_dev, local_rank)
    torch.cuda.set_device(local_rank % n_dev if args.backend == "gloo" else local_rank)
    for kv in args.tunable:
        N.set_tunable(kv.split("=")[0], int(kv.split("=")[1]))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()  # (the communicator, and its banner, come up here at the latest)
            else:
                dist.init_process_group("gloo")

    multi = world > 1
    reserve = 0
    if multi and args.backend == "nccl":
        # the all-gather of step k runs under the scan of step k+1, and a scan workgroup holds a whole CU's LDS: a few CUs are
        # kept free for RCCL's kernels (measured cost of the reserve on one GPU: profiles/r04/*_reserve_cus.txt)
        reserve = int(os.environ.get("ACGPU_RESERVE_CUS", "8"))
        N.set_tunable("reserve_cus", reserve)
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

    # synthetic shard, generated in place on the device
    seed = cfg["hay_seed"] + (rank if multi else 0)
    # overlap: step k+1 is enqueued before step k is collected -- N>1: the all-gather of step k runs under the scan of step
    # k+1; N=1: the GPU never waits for the host between steps.  Every step is complete before the timed region ends
    # (matcher.finish() + synchronize)
    cap = {"C4": n_units // 2, "C5": n_units // 8}.get(cfg_name, max(1 << 16, n_units // 128))
    matcher = ShardedMatcher(auto, n_units, with_ids=with_ids, cap=cap, overlap=True)
    matcher.cfg_name = cfg_name
    if cfg_name == "C5":
        # the token stream of SURVEY 8d (seed 2005 + rank), aperiodic, generated in place on the device
        synth.token_stream_on_device(matcher.own_ptr(), n_units, seed, kws, synth.swapcase_table(),
                                     stream=torch.cuda.current_stream().cuda_stream)
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
    # step() returns the PREVIOUS step's result and finish() the last one; all K steps are complete before the clock stops.
    # (The interpreter's cyclic garbage collector is kept out of the timed region: a generation-2 collection over the 10 000
    # keyword arrays is 35-84 ms -- tools/latency.py -- i.e. hundreds of steps; nothing of the library's is skipped by that.)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
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
    gc.enable()
    gc.unfreeze()
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_units = n_units * world
    ms_per_step = elapsed / args.steps * 1e3
    mb_per_s = total_units * 2 / (elapsed / args.steps) / 1e6
    kernel_ms = float(np.mean(scan_ms))
    kernel_what = matcher.last_kernel
    if cfg_name == "C4":
        # LongestMatch: the walk kernel writes no record -- the records come out of the chain passes behind it -- so the
        # algorithmic bytes (text in, records out) are divided by the SUM of the family's kernels (HIP events around all of them)
        kernel_ms = float(np.mean(scan_ms)) + float(np.mean(fin_ms))
        if matcher.last_kernel == "k_longest_bits":  # (csrc/acgpu_longest_bits.hip: no length array, no synchronisation pass, no emit pass)
            kernel_what = "k_longest_bits + k_longest_bits_finish (whole pipeline: the walk kernel writes the records itself)"
        else:
            kernel_what = matcher.last_kernel + " + k_longest_sync + k_longest_chain_lds + k_scan_* + k_longest_emit_ends (whole pipeline)"
    if cfg_name == "C5":
        # WholeWord: the scan leaves region-local records; they are in the reference's order only when k_ww_compact has run, so the
        # fraction is taken over scan + ordering pass, as for config 4 (the scan kernel alone is reported beside it)
        kernel_ms = float(np.mean(scan_ms)) + float(np.mean(fin_ms))
        kernel_what = matcher.last_kernel + (" + k_scan_* + k_ww_compact (whole pipeline)" if float(np.mean(fin_ms)) > 0 else
                                             " (one kernel: the scan puts its records in order itself -- the fused tail, csrc/acgpu_wholeword.hip)")
    if cfg_name in ("C2", "C3") and float(np.mean(fin_ms)) == 0.0 and matcher.last_kernel.startswith("k_ac_tile"):
        kernel_what = matcher.last_kernel + " (one kernel: scan + ordering of the records -- the fused tail, csrc/acgpu_tile.hip)"
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
            **({"tunables": list(args.tunable)} if args.tunable else {}),
            "keywords": len(kws), "states": info["n_states"], "classes": info["n_classes"],
            "table": ("dense u%d" % (8 * info["entry_bytes"])) if info["dense"] else "hashed",
            "lds_states": info["lds_states"], "units_per_gpu": n_units, "matches_per_gpu": n_matches_local,
            "matches_total": n_matches_total, "record_bytes": rec_bytes, "build_s": round(build_s, 3),
            "parallelism": ("shard%d+halo(%d,%d)+allgather/%s" % (world, matcher.sb.halo, matcher.sb.right, args.backend))
            if multi else "single",
            "gather_records_per_rank": matcher.cap if multi else None, "reserve_cus": reserve,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
            "kernel": kernel_what, "kernel_ms": round(kernel_ms, 4), "scan_ms": round(float(np.mean(scan_ms)), 4),
            "finalize_ms": round(float(np.mean(fin_ms)), 4),
            "algorithmic_bytes": alg_bytes,
        },
    }

    if multi:  # every rank's own dominant-kernel figures (the headline roofline object above is rank 0's)
        mine = {"rank": rank, "device": torch.cuda.current_device(), "kernel": kernel_what, "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes": alg_bytes, "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "matches": n_matches_local, "host_syncs_per_step": matcher.host_syncs, "redone_steps": matcher.redone_steps}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        out["roofline"]["per_rank"] = per_rank

    # attainable ceiling: the fastest pure read of this shard found on the chip (pattern 1), timed in this run; and what the
    # tile kernels' own access pattern reads at (pattern 0: their stream alone)
    ms = ctypes.c_float(0)
    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if N.lib().acgpu_stream_probe(matcher.own_ptr(), n_units * 2, cur, 7, 1, ctypes.byref(ms)) == N.OK and ms.value > 0:
        att = n_units * 2 / (ms.value * 1e-3) / 1e9
        out["roofline"]["attainable"] = round(att, 1)
        out["roofline"]["attainable_ms"] = round(ms.value, 4)
        out["roofline"]["frac_of_attainable"] = round(achieved / att, 4)
        out["roofline"]["attainable_what"] = ("k_stream_probe_best: the fastest pure read of the shard found on this chip (2 KiB tiles dealt "
                                              "round robin to 2048 workgroups of 256 lanes, 32 B per lane, four tiles in flight)")
    if N.lib().acgpu_stream_probe(matcher.own_ptr(), n_units * 2, cur, 7, 0, ctypes.byref(ms)) == N.OK and ms.value > 0:
        out["roofline"]["kernel_pattern_read_gbps"] = round(n_units * 2 / (ms.value * 1e-3) / 1e9, 1)
        out["roofline"]["kernel_pattern_what"] = "k_stream_probe: the tile kernels' own pattern as a pure read (one span per wave, 64 B per lane and tile)"

    # The same steps with the ordering of the records in launches of their own behind the scan (tile_form 3: what rounds 2-5
    # timed), a few of them, outside the timed region: the scan kernel ALONE against the roofline, for comparison with those
    # rounds' figures -- `frac` above is of the whole call, which is one kernel now.
    if not multi and cfg_name in ("C2", "C3", "C5") and float(np.mean(fin_ms)) == 0.0 and not args.no_separate_launches and \
            not any(t.startswith("tile_form") for t in args.tunable):
        N.set_tunable("tile_form", 3)
        sep = []
        for _ in range(7):
            sep.append(step(profile=True))
        sep.append(matcher.finish())
        N.set_tunable("tile_form", 0)
        sep = [r for r in sep if r is not None][2:]
        s_ms, f_ms = float(np.mean([r["scan_ms"] for r in sep])), float(np.mean([r["finalize_ms"] for r in sep]))
        out["roofline"]["separate_launches"] = {
            "scan_ms": round(s_ms, 4), "finalize_ms": round(f_ms, 4), "frac_scan_alone": round(alg_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
            "frac_scan_and_finalize": round(alg_bytes / ((s_ms + f_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
            "what": "tunable tile_form = 3: the scan kernel, then the ordering of its records in launches of their own (the form of rounds 2-5)"}

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (counters cannot be read from inside
    # this process); the committed measurement is attached when it is for THIS configuration, this size and these kernel
    # sources (keyed by a hash of csrc/: a figure measured before a kernel edit is not carried over)
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            tr = json.load(f)
        src = N.source_hash()
        for ent in (tr if isinstance(tr, list) else [tr]):
            if ent.get("config") == cfg_name and ent["units_per_gpu"] == n_units:
                if ent.get("csrc_sha256") == src:
                    out["roofline"]["traffic"] = ent["traffic_bytes"]
                    out["roofline"]["traffic_source"] = ent["source"]
                    if "traffic_bytes_range" in ent:
                        out["roofline"]["traffic_range"] = ent["traffic_bytes_range"]
                else:
                    out["roofline"]["traffic_note"] = ("profiles/latest_traffic.json holds a figure for other kernel sources (%s, now %s): "
                                                       "not attached" % (ent.get("csrc_sha256"), src))
    except (OSError, KeyError, ValueError):
        pass

    # ---- what the timed steps produced, checked against the CPU oracle on a prefix of the shard ------------------------
    verified = None
    if not args.no_cpu_baseline:
        sample = min(n_units, 1 << (args.cpu_sample_log2 if not multi else min(args.cpu_sample_log2, 22)))
        verified, digest, n_checked = verify(cfg_name, kws, matcher, sample)
        if multi:
            t = torch.tensor([1 if verified else 0], dtype=torch.int64, device="cpu" if args.backend == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            verified = bool(t.item())
        out["verified"] = verified
        out["verified_what"] = ("records of the last timed step == oracle records on the first 2^%d units of %s shard "
                                "(%d records compared on rank 0)" % (int(np.log2(sample)), "every rank's" if multi else "the",
                                                                     n_checked))
        out["records_sha256"] = digest
        assert verified, "bench.py: the GPU records differ from the oracle on the sample prefix -- the line above is void"

    if rank == 0 and not multi and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg_name, kws, matcher, min(n_units, 1 << args.cpu_sample_log2))
        out["end_to_end"] = end_to_end(auto, matcher, with_ids, n_units, out.get("records_sha256"))
        out["end_to_end_stream"], out["single_call"] = end_to_end_stream(auto, matcher, with_ids, n_units)
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def main_readme(args):
    """`--config README`: the reference's OWN published workload (R/README.md:126-152) -- a 235 886-word dictionary (an
    English-shaped stand-in of that size: /usr/share/dict/words does not exist here), one paragraph of English text
    (T/SetTest.java:50-54) per match() call, for AhoCorasickSet / WholeWordMatchSet / LongestMatchSet -- three ways:
    (i) microseconds per single acgpu_match_u16 call (what StringSet.match(String, listener) costs through the C ABI),
    (ii) microseconds per haystack through the batch entry (10 000 paragraphs in one call), (iii) milliseconds per GiB on a
    device-resident text of such words; beside the reference's published 3.6 / 2.9 / 7.1 us per call (hardware unstated)."""
    import torch

    from ahocorasick_amd import _native as N
    from ahocorasick_amd import synth
    from ahocorasick_amd.strings import Automaton, utf16
    from ahocorasick_amd.unicode_tables import default_word_chars
    from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, Oracle

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU matching path)"
    words = synth.readme_dictionary()
    para = utf16(synth.README_PARAGRAPH)
    n_units = 1 << args.units_log2
    block = synth.readme_text(2006, min(n_units, 1 << 25), words)
    d_hay = torch.from_numpy(block.view(np.int16)).cuda().repeat(max(1, n_units // block.size))
    fams = [("AhoCorasickSet", N.MODE_ALL, FAM_AC, {}, 3.6), ("WholeWordMatchSet", N.MODE_WHOLEWORD, FAM_WHOLEWORD, {"word_chars": default_word_chars()}, 2.9),
            ("LongestMatchSet", N.MODE_LONGEST, FAM_LONGEST, {}, 7.1)]
    res = {}
    L = N.lib()
    stream = torch.cuda.current_stream().cuda_stream
    for name, mode, fam, kw, ref_us in fams:
        t0 = time.time()
        auto = Automaton(mode, words, True, **kw)
        build_s = time.time() - t0
        info = auto.info()
        r = {"build_s": round(build_s, 2), "states": info["n_states"], "classes": info["n_classes"], "reference_us_per_call": ref_us}
        # (i) one call: preallocated buffers, the bare ctypes call in a loop
        cap = 4096
        out = np.empty((cap, 2), dtype=np.int32)
        n_out = ctypes.c_uint64(0)
        hp, op = para.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)
        for _ in range(50):
            N.check(L.acgpu_match_u16(auto.handle, hp, para.size, N.REC_SET, op, cap, ctypes.byref(n_out)), "acgpu_match_u16")
        got = out[:n_out.value].copy()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(400):
                L.acgpu_match_u16(auto.handle, hp, para.size, N.REC_SET, op, cap, ctypes.byref(n_out))
            ts.append((time.perf_counter() - t0) / 400)
        r["us_per_call"] = round(float(np.median(ts)) * 1e6, 2)
        N.set_tunable("tile_debug", 1 << 41)  # the general path, for comparison
        for _ in range(5):
            L.acgpu_match_u16(auto.handle, hp, para.size, N.REC_SET, op, cap, ctypes.byref(n_out))
        t0 = time.perf_counter()
        for _ in range(100):
            L.acgpu_match_u16(auto.handle, hp, para.size, N.REC_SET, op, cap, ctypes.byref(n_out))
        r["us_per_call_general_path"] = round((time.perf_counter() - t0) / 100 * 1e6, 2)
        N.set_tunable("tile_debug", 0)
        r["matches_per_call"] = int(len(got))
        if not args.no_cpu_baseline:
            orc = Oracle(fam, words, **({"word_chars": default_word_chars()} if fam == FAM_WHOLEWORD else {}))
            want = orc.match(para)[:, :2]
            assert got.shape == want.shape and (got == want).all(), "bench.py --config README: %s differs from the oracle on the paragraph" % name
            orc.count(para)
            t0 = time.perf_counter()
            for _ in range(2000):
                orc.count(para)
            r["cpu_port_us_per_call"] = round((time.perf_counter() - t0) / 2000 * 1e6, 2)
        # (ii) 10 000 paragraphs through the batch entry
        hs = [para] * 10000
        b = auto.match_batch(hs, False, cap=len(got) * 10000 + 16)  # (the first call of a size allocates its staging buffers)
        tb = []
        for _ in range(3):
            t0 = time.perf_counter()
            b = auto.match_batch(hs, False, cap=len(got) * 10000 + 16)
            tb.append((time.perf_counter() - t0) / 10000)
        r["us_per_haystack_batched"] = round(float(np.median(tb)) * 1e6, 3)
        assert len(b) == len(got) * 10000
        # (iii) a device-resident text of dictionary words
        # (the word list holds the 52 single letters: AhoCorasickSet reports every letter of the text, as the reference does)
        capd = int(n_units * 1.75) if mode == N.MODE_ALL else n_units // 2
        d_out = torch.empty((capd, 2), dtype=torch.int32, device="cuda")
        ms, nm, kern = [], 0, ""
        for i in range(args.warmup + args.steps):
            nm, rc, prof, _ = auto.match_device(d_hay.data_ptr(), n_units, False, d_out.data_ptr(), capd, stream=stream, profile=True)
            if rc == N.E_OVERFLOW:
                capd = nm + 16
                d_out = torch.empty((capd, 2), dtype=torch.int32, device="cuda")
                nm, rc, prof, _ = auto.match_device(d_hay.data_ptr(), n_units, False, d_out.data_ptr(), capd, stream=stream, profile=True)
            N.check(rc, "acgpu_match_device")
            if i >= args.warmup:
                ms.append(prof["scan_ms"] + prof["finalize_ms"])
                kern = prof["scan_kernel"]
        gib = n_units * 2 / float(1 << 30)
        ab = 2 * n_units + 8 * nm
        # floor: the algorithmic bytes (2 per unit read + 8 per record written) at the best pure read rate measured on this chip
        r.update(ms_per_gib=round(float(np.median(ms)) / gib, 4), kernel=kern, matches_per_gib=int(nm / gib),
                 roofline_frac=round(ab / (float(np.median(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 floor_ms_per_gib=round(ab / gib / (ATTAINABLE_GBPS * 1e9) * 1e3, 3))
        if not args.no_cpu_baseline:  # the device records of a prefix against the oracle
            k = min(n_units, 1 << 22)
            want = orc.match(block[:k], cap=4 * k)[:, :2]
            recs = d_out[:min(nm, 4 * k)].cpu().numpy()
            if mode == N.MODE_ALL:
                ok = (recs[recs[:, 1] <= k] == want).all()
            else:
                lim = k - info["max_keyword_len"] - 2
                ok = (recs[recs[:, 1] < lim] == want[want[:, 1] < lim]).all()
            assert ok, "bench.py --config README: %s differs from the oracle on the text" % name
            r["verified"] = True
        res[name] = r
        del auto
    a = res["AhoCorasickSet"]
    out = {"metric": "haystack MB/s (UTF-16 bytes scanned per second, records delivered in reference order)",
           "value": round(float(1 << 30) / (a["ms_per_gib"] * 1e-3) / 1e6, 1), "unit": "MB/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": a["ms_per_gib"] * n_units * 2 / float(1 << 30), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u16", "data": "synthetic",
           "config": {"workload": "the reference's published workload (R/README.md:126-152): 235886-word dictionary (English-shaped stand-in), "
                                  "the paragraph of T/SetTest.java:50-54 per call (%d units), and 2^%d units of such words" % (para.size, args.units_log2),
                      "keywords": len(words)},
           "roofline": {"bound": "hbm", "achieved": round(a["roofline_frac"] * HBM_PEAK_GBPS, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": a["roofline_frac"], "traffic": None, "kernel": a["kernel"]},
           "readme": res}
    print(json.dumps(out))
    return 0


def main_single_process(args):
    """`--single-process`: the multi-GPU job of config 3 (or C2/C4/C5 with --config) behind the C ABI -- one process, one
    acgpu_comm over the device list, every step one acgpu_match_device_allgather call: each device scans its resident shard
    into its slot of its gather buffer, one all-gather (single-process RCCL, or peer copies) leaves every device with every
    shard's records.  Same JSON line as the one-rank-per-GPU job."""
    import torch

    from ahocorasick_amd import _native as N
    from ahocorasick_amd import synth
    from ahocorasick_amd.strings import Automaton, Comm

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU matching path)"
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    assert len(devices) == args.gpus, "--devices must name --gpus devices"
    k = len(devices)
    cfg_name = args.config or ("C3" if k > 1 else "C2")
    if cfg_name == "C5" and args.units_log2 == 29:
        args.units_log2 = 28
    n_units = 1 << args.units_log2
    with_ids = cfg_name in ("C2", "C5")
    rec_bytes = 12 if with_ids else 8
    cols = rec_bytes // 4
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
    m = info["max_keyword_len"]
    left, right = {"C4": (0, m - 1), "C5": (1, m + 1)}.get(cfg_name, (m - 1, 0))
    pad = (left + 7) // 8 * 8
    with stdout_to_stderr():
        comm = Comm(devices, N.TRANSPORT_AUTO)
    # shard g: [pad | 2^units_log2 own units (stream hay_seed + g) | right halo], resident on devices[g]
    bufs, shards = [], []
    for g, dev in enumerate(devices):
        torch.cuda.set_device(dev)
        first, last = g == 0, g == k - 1
        buf = torch.zeros(pad + n_units + (0 if last else right), dtype=torch.int16, device="cuda:%d" % dev)
        own = buf[pad:pad + n_units]
        if cfg_name == "C5":
            synth.token_stream_on_device(own.data_ptr(), n_units, cfg["hay_seed"] + g, kws, synth.swapcase_table(),
                                         stream=torch.cuda.current_stream().cuda_stream)
        else:
            tab = np.ascontiguousarray(synth.ALPHA_AB_75 if cfg_name == "C4" else synth.ALPHA_LOWER)
            N.check(N.lib().acgpu_synth_fill(own.data_ptr(), n_units, 0, cfg["hay_seed"] + g, tab.ctypes.data_as(ctypes.c_void_p), len(tab),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "synth_fill")
        torch.cuda.synchronize(dev)
        bufs.append(buf)
    for g in range(k):  # halos, once per haystack: device-to-device copies
        if g > 0 and left:
            bufs[g][pad - left:pad].copy_(bufs[g - 1][pad + n_units - left:pad + n_units])
        if g + 1 < k and right:
            bufs[g][pad + n_units:].copy_(bufs[g + 1][pad:pad + right])
    for g, dev in enumerate(devices):
        torch.cuda.synchronize(dev)
        v0 = pad if g == 0 else 0  # (the pad in front of the first shard is not text)
        shards.append(dict(d_hay=bufs[g].data_ptr() + 2 * v0, n_units=bufs[g].numel() - v0, own=(pad - v0, pad - v0 + n_units),
                           text_begin=g == 0, text_end=g == k - 1))
    gcap = {"C4": n_units // 2, "C5": n_units // 8}.get(cfg_name, max(1 << 16, n_units // 128))
    gbufs = None

    def alloc():
        slot = N.gather_slot_bytes(gcap, rec_bytes)
        return [torch.zeros(k * slot // 4, dtype=torch.int32, device="cuda:%d" % d) for d in devices]

    def step(profile=False):
        nonlocal gcap, gbufs
        while True:
            if gbufs is None:
                gbufs = alloc()
                for d in devices:
                    torch.cuda.synchronize(d)
            rc, counts, _, prof = comm.match_device_allgather(auto, shards, with_ids, [g.data_ptr() for g in gbufs], gcap, profile=profile)
            if rc == N.E_OVERFLOW:
                gcap, gbufs = int(max(counts) * 1.0625) + 1024, None
                continue
            N.check(rc, "acgpu_match_device_allgather")
            return counts, prof

    counts, _ = step()
    if max(counts) * 1.0625 + 1024 < 0.8 * gcap:  # the gather moves gcap records per device: follow the counts
        gcap, gbufs = int(max(counts) * 1.0625) + 1024, None
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    profs = []
    for _ in range(args.steps):
        counts, prof = step(profile=True)
        profs.append(prof)
    for d in devices:
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / args.steps * 1e3
    per_dev = []
    for g in range(k):
        sc = float(np.mean([p[g]["scan_ms"] for p in profs]))
        fin = float(np.mean([p[g]["finalize_ms"] for p in profs]))
        kms = sc + fin if cfg_name == "C4" else sc
        ab = 2 * n_units + rec_bytes * counts[g]
        per_dev.append({"share": g, "device": devices[g], "kernel": profs[-1][g]["scan_kernel"], "kernel_ms": round(kms, 4),
                        "scan_ms": round(sc, 4), "finalize_ms": round(fin, 4), "algorithmic_bytes": ab,
                        "achieved": round(ab / (kms * 1e-3) / 1e9, 2), "frac": round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                        "matches": counts[g]})
    out = {
        "metric": "haystack MB/s (UTF-16 bytes scanned per second, records delivered in reference order)",
        "value": round(n_units * k * 2 / (elapsed / args.steps) / 1e6, 1), "unit": "MB/s", "n_gpus": k, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16", "data": "synthetic", "matches_per_s": round(sum(counts) / (elapsed / args.steps), 1),
        "config": {
            "workload": "BASELINE config %s through ONE host process: %d x 2^%d UTF-16 units, acgpu_match_device_allgather" % (cfg_name[1], k, args.units_log2),
            "keywords": len(kws), "states": info["n_states"], "classes": info["n_classes"], "units_per_gpu": n_units,
            "matches_per_gpu": counts[0], "matches_total": sum(counts), "record_bytes": rec_bytes, "build_s": round(build_s, 3),
            "parallelism": "single-process shard%d+halo(%d,%d)+allgather/%s" % (k, left, right, "rccl" if comm.transport == N.TRANSPORT_RCCL else "peer-copies"),
            "devices": devices, "gather_records_per_device": gcap,
        },
        "roofline": dict(per_dev[0], bound="hbm", peak=HBM_PEAK_GBPS, unit="GB/s", traffic=None, per_device=per_dev),
    }
    if not args.no_cpu_baseline:  # share 0's slot of the LAST device's gather buffer against the oracle on a prefix of share 0
        sample = min(n_units, 1 << min(args.cpu_sample_log2, 24))
        o = _oracle(cfg_name, kws)
        hay = bufs[0][pad:pad + sample].cpu().numpy().view(np.uint16)
        want = o.match(hay, cap=max(1 << 16, sample // (2 if cfg_name == "C4" else 8)))[:, :cols]
        slot_words = N.gather_slot_bytes(gcap, rec_bytes) // 4
        recs = gbufs[-1][4:4 + counts[0] * cols].cpu().numpy().reshape(-1, cols)
        if cfg_name in ("C2", "C3"):
            got = recs[recs[:, 1] <= sample]
        else:
            want = want[want[:, 1] < sample - m - 1]
            got = recs[recs[:, 1] < sample - m - 1]
        hdr = [int(gbufs[-1][j * slot_words:j * slot_words + 2].cpu().numpy().view(np.int64)[0]) for j in range(k)]
        out["verified"] = bool(got.shape == want.shape and (got == want).all() and len(want) > 0 and hdr == counts)
        out["verified_what"] = ("share 0's records as gathered on the last device == oracle records on the first 2^%d units (%d "
                                "records), and every gathered header holds its share's count" % (int(np.log2(sample)), len(want)))
        assert out["verified"], "bench.py --single-process: the gathered records differ from the oracle"
    comm.close()
    print(json.dumps(out))
    return 0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N fresh ranks under torch.distributed.run (one per GPU,
    rendezvous on 127.0.0.1), relay what they print -- rank 0's single JSON line -- and return their exit code.  This parent
    has not imported torch or made any HIP call: nothing that touched the GPU is re-executed, the ranks are children."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def _oracle(cfg_name, kws):
    from oracle.oracle import FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, Oracle
    if cfg_name == "C4":
        return Oracle(FAM_LONGEST, kws)
    if cfg_name == "C5":
        from ahocorasick_amd.unicode_tables import default_word_chars, java_lower_table
        return Oracle(FAM_WHOLEWORD, kws, case_sensitive=False, lower=java_lower_table(), word_chars=default_word_chars())
    return Oracle(FAM_AC, kws)


def verify(cfg_name, kws, matcher, sample_units):
    """Outside the timed region: the records the last step left in this rank's buffer against the CPU oracle on the first
    `sample_units` units of this rank's view (halo included for rank > 0).  Returns (ok, sha256 of ALL local records, number
    of records compared)."""
    o = _oracle(cfg_name, kws)
    max_len = max(len(k) for k in kws)
    n = int(matcher.counts[matcher.rank])
    recs = matcher.gathered[matcher.rank, :n].cpu().numpy()
    digest = hashlib.sha256(np.ascontiguousarray(recs).tobytes()).hexdigest()
    sb = matcher.sb
    shift = matcher.shift
    v0 = sb.pad - shift  # view start in the buffer
    lo = shift - sb.halo if matcher.rank else 0  # first text unit of the view that is real text (the halo)
    if cfg_name == "C4" and matcher.rank > 0:
        # a later shard's greedy chain enters where the rank before it left: the oracle runs from that position (a chain
        # started there IS the rest of the whole text's chain)
        lo = shift + int(matcher.chain_entry_applied or 0)
    hay = sb.buf[v0 + lo:v0 + shift + sample_units].cpu().numpy().view(np.uint16)
    want = o.match(hay, cap=max(1 << 16, hay.size // (2 if cfg_name == "C4" else 8)))[:, :recs.shape[1]].copy()
    want[:, :2] += lo
    end = shift + sample_units
    if cfg_name in ("C2", "C3"):  # a record belongs to the prefix iff it ENDS there and its last unit is owned
        want = want[want[:, 1] - 1 >= shift]
        got = recs[recs[:, 1] <= end]
    else:  # position order; what the cut at the end of the prefix can change lies in its last max_len + 1 units
        want = want[(want[:, 0] >= shift) & (want[:, 1] < end - max_len - 1)]
        got = recs[recs[:, 1] < end - max_len - 1]
    ok = got.shape == want.shape and bool((got == want).all()) and len(want) > 0
    return ok, digest, len(want)


def cpu_baseline(cfg_name, kws, matcher, sample_units):
    """The reference-shaped CPU restatement (oracle/ac_oracle.c, kind "port"), single thread like the reference, pinned
    to one core, no-op listener (R/README.md:144), on a bounded prefix of the same haystack: 1 warm-up + 5 timed runs,
    median."""
    o = _oracle(cfg_name, kws)
    hay = matcher.own_units_host(sample_units)
    core, old = None, None
    try:
        old = os.sched_getaffinity(0)
        # the core this thread is running on (field 39 of /proc/self/stat): pinning it THERE keeps it -- and everything it
        # allocates afterwards -- on its NUMA node (pinning to the highest core moved the process to the far socket, and the
        # host-side copies of the measurements behind this one ran at a third of their rate)
        try:
            with open("/proc/thread-self/stat") as f:
                core = int(f.read().rsplit(")", 1)[1].split()[36])
            if core not in old:
                core = max(old)
        except (OSError, ValueError, IndexError):
            core = max(old)
        os.sched_setaffinity(0, {core})
    except (AttributeError, OSError):
        core = None
    try:
        o.count(hay)  # warm-up: the whole sample once (page faults, caches, branch predictors)
        times = []
        n = 0
        for _ in range(5):
            t0 = time.perf_counter()
            n = o.count(hay)
            times.append(time.perf_counter() - t0)
    finally:
        if old is not None and core is not None:
            os.sched_setaffinity(0, old)
    dt = float(np.median(times))
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(hay.size * 2 / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first 2^%d units of the same haystack, %d matches, no-op listener; 1 warm-up + 5 runs, median %.2f s "
                      "(min %.2f, max %.2f)" % (int(np.log2(hay.size)), n, dt, min(times), max(times)),
            "pinned_core": core, "host_cpu": model, "host_cpus": os.cpu_count()}


def end_to_end(auto, matcher, with_ids, sample_units, device_digest=None):
    """acgpu_match_u16 -- what the JNI facade calls -- on the whole shard in pageable host memory: chunks through the pinned
    staging ring to the device while the chunks that have arrived are scanned, records copied back.  PCIe-inclusive; reported
    beside `value`, never as it.  The records must be the device-resident run's (same SHA-256)."""
    hay = matcher.own_units_host(sample_units)
    cap = {"C4": sample_units // 2, "C5": sample_units // 8}.get(getattr(matcher, "cfg_name", ""), max(1 << 16, sample_units // 64))
    auto.match_host(hay[:1 << 20], with_ids, cap=cap)  # warm-up: staging buffers
    recs = auto.match_host(hay, with_ids, cap=cap)
    digest = hashlib.sha256(np.ascontiguousarray(recs).tobytes()).hexdigest()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        auto.match_host(hay, with_ids, cap=cap)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    r = {"value": round(hay.size * 2 / dt / 1e6, 1), "unit": "MB/s", "what": "acgpu_match_u16 on 2^%d units in pageable host "
         "memory: pinned staging ring + H2D under the shard scans + D2H of %d records, median of 3" % (int(np.log2(hay.size)), len(recs)),
         "ms": round(dt * 1e3, 2), "records_sha256": digest}
    if device_digest is not None:
        r["same_records_as_device_run"] = digest == device_digest
        assert digest == device_digest, "bench.py: acgpu_match_u16 delivered other records than the device-resident run"
    return r


def end_to_end_stream(auto, matcher, with_ids, sample_units):
    """acgpu_stream_feed -- match(Readable, ...) -- on a prefix of the shard in 4 Mi-unit chunks, pipelined form (a feed returns
    the previous chunk's records; host copy, transfer and scan overlap), and the fixed cost of ONE acgpu_match_u16 call on a
    paragraph-sized haystack (472 units: the reference's published workload is one such call, R/README.md:126-152)."""
    from ahocorasick_amd import _native as N
    from ahocorasick_amd.strings import Stream
    n = min(sample_units, 1 << 27)
    hay = matcher.own_units_host(n)
    chunk = 1 << 22
    best = None
    for pipelined in (False, True):
        dts = []
        for _ in range(4):  # (the first pass allocates the staging buffers; median of the other three: a pass now and then holds a
            st = Stream(auto, with_ids=with_ids, pipelined=pipelined)  # 20-40 ms stall of the host that is not the library's)
            t0 = time.perf_counter()
            total = 0
            for o in range(0, n, chunk):
                total += len(st.feed(hay[o:o + chunk], final=o + chunk >= n, cap=chunk // 8))
            dts.append(time.perf_counter() - t0)
            st.close()
        dt = float(np.median(dts[1:]))
        if pipelined:
            best = {"value": round(2.0 * n / dt / 1e6, 1), "unit": "MB/s", "what": "acgpu_stream_feed, pipelined form, 2^%d units in 2^22-unit "
                    "chunks from pageable host memory, %d records; median of 3 passes" % (int(np.log2(n)), total), "synchronous_form_mbps": sync_rate}
        else:
            sync_rate = round(2.0 * n / dt / 1e6, 1)
    para = np.ascontiguousarray(hay[:472])
    out = np.empty((4096, 3), dtype=np.int32)
    n_out = ctypes.c_uint64(0)
    L = N.lib()
    args = (auto.handle, para.ctypes.data_as(ctypes.c_void_p), para.size, N.REC_MAP if with_ids else N.REC_SET, out.ctypes.data_as(ctypes.c_void_p), 4096,
            ctypes.byref(n_out))
    for _ in range(50):
        L.acgpu_match_u16(*args)
    import gc
    gc.collect()
    gc.freeze()  # (a generation-2 collection of the interpreter over the dictionary's arrays is 35-84 ms: tools/latency.py -- not the library's time)
    ts = []
    for _ in range(1000):  # every call timed on its own: the median is what a call costs, p99 / max what a caller now and then waits
        t0 = time.perf_counter()
        L.acgpu_match_u16(*args)
        ts.append(time.perf_counter() - t0)
    gc.unfreeze()
    ts = np.array(ts) * 1e6
    return best, {"value": round(float(np.median(ts)), 2), "unit": "us", "p99": round(float(np.percentile(ts, 99)), 2), "max": round(float(ts.max()), 2),
                  "what": "one acgpu_match_u16 call on a 472-unit haystack (one launch, host-mapped buffers; the bare ctypes call, each of "
                          "1000 timed on its own: median, 99th percentile, slowest)"}


if __name__ == "__main__":
    main()
