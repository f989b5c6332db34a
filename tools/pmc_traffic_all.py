#!/usr/bin/env python3
"""HBM traffic of the configurations' kernels from the counter passes of tools/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE
and --pmc WRITE_SIZE, separate passes over tools/kbench.py --config C2|C4|C5): per kernel, the counters of the last dispatch of
each kernel, in bytes, and per configuration the sum over the family's kernels.
usage: pmc_traffic_all.py <dir with pmc_fetch_C4/ ... pmc_write_C5/> [--latest profiles/latest_traffic.json --tag profiles/r03/x.json]

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (16 B per
lane).  That factor is applied PER KERNEL, and only to kernels whose reads are such streams (STREAMING below: the text stream
of the scan kernels, 16-byte pieces of len[] through LDS-DMA, whole bitmap words); a kernel that gathers (synchronisation
points, table probes, ordering passes that read scattered slots) is taken as reported, and a kernel that does both gets a
RANGE [as reported, doubled] -- its figure is an estimate and is tagged so."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STREAMING = ("k_longest_chain_lds", "k_longest_emit_ends", "k_scan_", "k_stream_probe", "k_longest_bits")   # reads: 16 B/lane streams only
# (k_longest_bits: the text as 16-byte pieces, the trie sits in LDS; its parked marks come back as whole 16-byte words)
MIXED = ("k_ww_tile", "k_ww_pp", "k_ac_tile", "k_longest_block", "k_longest_walk_list", "k_permute",
         "k_ac_states", "k_longest_follow")       # a stream plus gathers
RECORD_BYTES = {"C2": 12, "C4": 8, "C5": 12}   # what collect_profiles.sh runs: Map records, config 4 with --set


GENERATORS = ("k_synth", "k_token")  # the benchmark's input generators (and everything dispatched before them: their prefix sums)


def last_per_kernel(path, counter):
    per = collections.OrderedDict()
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        rows = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "acgpu::" not in r["Kernel_Name"]:
                continue
            rows[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("acgpu::", "")
        gen = [d for d in names if names[d].startswith(GENERATORS)]
        first = max(gen) + 1 if gen else 0
        for d in sorted(rows):
            if d >= first:
                per[names[d]] = rows[d]  # the last dispatch of every kernel wins
    return per


def algorithmic_bytes(d, cfg, units):
    """2 B per unit + R per record, as bench.py counts them; the record count is what tools/kbench.py printed in the counter pass."""
    import re
    try:
        m = re.findall(r"n_out=(\d+) rc=0", open("%s/pmc_fetch_%s.log" % (d, cfg)).read())
    except OSError:
        m = []
    return 2 * units + RECORD_BYTES[cfg] * int(m[-1]) if m else None


def main():
    d = sys.argv[1]
    impossible = []
    from ahocorasick_amd import _native as N
    src = N.source_hash()
    out = {"csrc_sha256": src}
    latest = []
    for cfg, units in (("C2", 1 << 29), ("C4", 1 << 29), ("C5", 1 << 28)):
        fetch = last_per_kernel("%s/pmc_fetch_%s" % (d, cfg), "FETCH_SIZE")
        write = last_per_kernel("%s/pmc_write_%s" % (d, cfg), "WRITE_SIZE")
        if not fetch:
            continue
        ks = {}
        lo = hi = 0.0
        for k in fetch:
            f, w = fetch[k] * 1024, write.get(k, 0.0) * 1024
            if k.startswith(STREAMING):
                kind, a, b = "stream (FETCH_SIZE x2)", 2 * f + w, 2 * f + w
            elif k.startswith(MIXED):
                kind, a, b = "stream + gathers (range: FETCH_SIZE as reported .. x2)", f + w, 2 * f + w
            else:
                kind, a, b = "gathers (as reported)", f + w, f + w
            ks[k] = {"FETCH_SIZE_bytes_as_reported": f, "WRITE_SIZE_bytes": w, "reads": kind, "traffic_bytes_range": [a, b]}
            lo += a
            hi += b
        alg = algorithmic_bytes(d, cfg, units)
        out[cfg] = {"units": units, "kernels": ks, "pipeline_traffic_bytes_range": [lo, hi], "algorithmic_bytes": alg}
        if alg is not None and hi < 0.98 * alg:  # a pipeline cannot move less than the text it reads plus the records it writes
            impossible.append("%s: traffic %.3f GB < algorithmic %.3f GB (a kernel is missing from STREAMING / MIXED?)" % (cfg, hi / 1e9, alg / 1e9))
        latest.append({"config": cfg, "units_per_gpu": units, "csrc_sha256": src, "traffic_bytes": hi, "traffic_bytes_range": [lo, hi],
                       "source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/kbench.py, summed over the "
                                 "configuration's kernels; FETCH_SIZE x2 (gfx950 half-count of wide streaming reads, MI355X_MICROARCH.md) per "
                                 "kernel where its reads are streams, as reported where they are gathers; `traffic` is the upper end of "
                                 "`traffic_range`" % (sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else d)})
    if "--c2-calibrated" in sys.argv:
        # config 2's tile kernel has a better figure: tools/pmc_traffic.py calibrates the half-count on the kernel's own
        # stream-only build and applies it to the stream alone (gathers as reported)
        cal = json.load(open(sys.argv[sys.argv.index("--c2-calibrated") + 1]))
        for ent in latest:
            if ent["config"] == "C2" and "C2" in out:
                perm = sum(v["traffic_bytes_range"][1] for k, v in out["C2"]["kernels"].items() if not k.startswith("k_ac_tile"))
                ent["traffic_bytes"] = cal["traffic_bytes_corrected"]
                ent["traffic_bytes_all_kernels"] = cal["traffic_bytes_corrected"] + perm
                ent["source"] += "; config 2: `traffic` is the tile kernel's, from the calibrated passes of tools/pmc_traffic.py (" + cal["calibration"] + ")"
        out["C2"]["tile_kernel_calibrated"] = cal
    print(json.dumps(out, indent=1))
    if impossible:
        raise SystemExit("pmc_traffic_all: " + "; ".join(impossible))
    if "--latest" in sys.argv:
        json.dump(latest, open(sys.argv[sys.argv.index("--latest") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
