#!/usr/bin/env python3
"""Turns the two counter passes of tools/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over
tools/kbench.py with the variants tile / noverify / stream, in that order) into the HBM traffic figure bench.py attaches
as roofline.traffic.   usage: pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> <out.json> [--latest profiles/latest_traffic.json]

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of a wide coalesced streaming read.
The stream-only build of the kernel reads exactly the haystack (+ the LDS tables per workgroup), which calibrates that
factor on the spot; the gathers of the verification (full - stream) are another access shape and are taken as reported."""
import collections
import csv
import glob
import json
import sys


def per_variant(path, counter):
    rows = collections.OrderedDict()
    name = None
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ac_tile" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            rows[int(r["Dispatch_Id"])] = rows.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    vals = [rows[k] for k in sorted(rows)]
    assert len(vals) >= 3 and len(vals) % 3 == 0, (path, len(vals))
    last = vals[-3:]  # the last round: tile, noverify, stream
    return {"full": last[0], "no_verify": last[1], "stream_only": last[2]}, name


def main():
    d, out = sys.argv[1], sys.argv[2]
    fetch, kname = per_variant(d + "/pmc_fetch", "FETCH_SIZE")
    write, _ = per_variant(d + "/pmc_write", "WRITE_SIZE")
    units = 1 << 29
    stream_bytes = 2 * units
    factor = stream_bytes / (fetch["stream_only"] * 1024)
    traffic = (2 * fetch["stream_only"] + (fetch["full"] - fetch["stream_only"]) + write["full"]) * 1024
    kshort = kname.split("(")[0].replace("void acgpu::", "")
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/kbench.py "
                  "(full / no-verify / stream-only builds of the same kernel in one process; tools/collect_profiles.sh + "
                  "tools/pmc_traffic.py); counter units KB (x1024 B)",
        "kernel": kshort, "workload": "BASELINE config 2, 2^29 units",
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write["full"],
        "calibration": "the stream-only build reads exactly 1.0 GiB and reports %.4f GiB: measured factor %.3f, the gfx950 "
                       "FETCH_SIZE x2 correction (MI355X_MICROARCH.md, HBM) is applied to the stream; the verification's "
                       "gathers (full - stream) are taken as reported" % (fetch["stream_only"] * 1024 / 2 ** 30, factor),
        "traffic_bytes_corrected": traffic,
    }
    json.dump(res, open(out, "w"), indent=1)
    if "--latest" in sys.argv:
        latest = sys.argv[sys.argv.index("--latest") + 1]
        json.dump({"kernel": kshort, "units_per_gpu": units, "traffic_bytes": traffic,
                   "source": out + " (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 on the "
                             "calibrated 16 B/lane stream per MI355X_MICROARCH.md)"}, open(latest, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
