#!/usr/bin/env python3
"""N-thread CPU number for DESIGN.md ("additionally report an N-thread number ... labelled as not the
reference's behaviour"): the single-threaded reference-shaped restatement (oracle/ac_oracle.c) run on T chunks of the
config-2 haystack in T threads, each chunk extended by the (max_keyword_len-1)-unit halo a sharded CPU run would need.
Development tool; uses the oracle, so it is test infrastructure like bench.py's cpu_baseline leg."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import synth  # noqa: E402
from oracle.oracle import FAM_AC, Oracle  # noqa: E402


def main():
    log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    kws = synth.config_keywords("C2")
    halo = max(len(k) for k in kws) - 1
    hay = synth.haystack(synth.CONFIGS["C2"]["hay_seed"], 1 << log2)
    o = Oracle(FAM_AC, kws)
    o.count(hay[:1 << 20])
    for T in (1, 8, 32, 64, 128, 256):
        if T > (os.cpu_count() or 1):
            break
        edges = np.linspace(0, hay.size, T + 1).astype(np.int64)
        res = [0] * T

        def work(i):
            res[i] = o.count(hay[max(edges[i] - halo, 0):edges[i + 1]])

        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        print("threads %3d: %8.1f MB/s  (%d matches incl. halo repeats, %.2f s)" % (T, hay.size * 2 / dt / 1e6, sum(res), dt))


if __name__ == "__main__":
    main()
