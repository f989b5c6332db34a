#!/usr/bin/env python3
"""Development tool: the status table of README.md from one collected set of bench lines (profiles/<round>/<prefix>_*.json), so
that the table the next reader sees is the collection's, not a mix of boxes.   usage: tools/readme_table.py profiles/r04 v2"""
import json
import os
import sys


def load(path):
    s = open(path).read()
    return json.loads(s[s.index('{"metric"'):])


def main():
    d, pre = sys.argv[1], sys.argv[2]
    rows = []
    for tag, name in (("bench_c2", "C2: AhoCorasickMap, 10 k keywords, 2^29 units"), ("bench_c3_1gpu", "C3 (one rank's share): AhoCorasickSet, 2^29 units"),
                      ("bench_c4", "C4: LongestMatchSet, 50 k prefix-closed keywords, 2^29 units"),
                      ("bench_c5", "C5 share: WholeWordMatchMap case-insensitive, 100 k words, 2^28 units"),
                      ("bench_c3_single_process_rccl", "C3 through ONE host process, 1 device, RCCL all-gather"),
                      ("bench_c3_single_process_peer2", "C3 through ONE host process, 2 shares of 2^28 units on one device, peer copies")):
        p = os.path.join(d, "%s_%s.json" % (pre, tag))
        if not os.path.exists(p):
            continue
        j = load(p)
        r = j["roofline"]
        rows.append("| %s | `%s` | %.4f ms | %.4f ms | %.1f %% | %s |" % (name, r["kernel"].split(" + ")[0][:60], j["ms_per_step"], r["kernel_ms"],
                                                                       100 * r["frac"], "verified" if j.get("verified") else "-"))
    print("| Configuration | Dominant kernel | Per step | Kernel(s) | of 8 TB/s | Records |")
    print("|---|---|---|---|---|---|")
    print("\n".join(rows))
    p = os.path.join(d, "%s_bench_c2.json" % pre)
    if os.path.exists(p):
        j = load(p)
        print("\nC2 extras: attainable %.0f GB/s, frac_of_attainable %.3f, traffic %s, cpu_baseline %.1f MB/s (1 core), end_to_end %.1f MB/s, "
              "end_to_end_stream %s MB/s (synchronous %s), single_call %s us" % (
                  j["roofline"].get("attainable", 0), j["roofline"].get("frac_of_attainable", 0), j["roofline"].get("traffic"), j["cpu_baseline"]["value"],
                  j["end_to_end"]["value"], j.get("end_to_end_stream", {}).get("value"), j.get("end_to_end_stream", {}).get("synchronous_form_mbps"),
                  j.get("single_call", {}).get("value")))
    p = os.path.join(d, "%s_bench_readme.json" % pre)
    if os.path.exists(p):
        j = load(p)
        print("\n| README workload (235 886 words) | us per call (reference) | general path | batched, per haystack | CPU port | ms per GiB | kernel |")
        print("|---|---|---|---|---|---|---|")
        for k, v in j["readme"].items():
            print("| %s | %.1f (%.1f) | %.0f | %.2f | %.1f | %.2f | `%s` |" % (k, v["us_per_call"], v["reference_us_per_call"], v["us_per_call_general_path"],
                                                                         v["us_per_haystack_batched"], v.get("cpu_port_us_per_call", 0), v["ms_per_gib"], v["kernel"][:48]))


if __name__ == "__main__":
    main()
