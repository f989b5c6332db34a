#!/usr/bin/env python3
"""Fixed cost of one acgpu_match_u16 call on short haystacks (development tool).

The 35-84 ms call that rounds 4 and 5 saw here (AhoCorasick, n = 4096, "max 39 147 us") was the 252nd call of the process every
time -- whatever the size at that point (LATENCY_SIZES), however long the process had run (LATENCY_SLEEP), with or without a
stream synchronisation every 64 calls -- and it is gone with the interpreter's cyclic garbage collector off (LATENCY_NOGC=1: max
62 us): a generation-2 collection that walks the 10 000 keyword arrays of the dictionary.  It is the host program's, not the
library's or the HIP runtime's; the tool now freezes the objects that exist before the loops (gc.freeze) and reports every call
of more than a millisecond with its index, so that anything of the library's own would show."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import _native as N, synth  # noqa: E402
from ahocorasick_amd.strings import Automaton  # noqa: E402
from ahocorasick_amd.unicode_tables import default_word_chars  # noqa: E402

import gc  # noqa: E402
if os.environ.get("LATENCY_NOGC"):  # (is an outlier the interpreter's? no cyclic garbage collection during the loops)
    gc.disable()
T0 = time.perf_counter()
calls_before = 0
ORDER = [int(x) for x in os.environ.get("LATENCY_SIZES", "64,4096,65536,1048576").split(",")]  # (the order of the sizes: to bisect an outlier)
for name, mode, kw in (("AhoCorasick C2 dict", N.MODE_ALL, {}), ("Longest", N.MODE_LONGEST, {}), ("Shortest", N.MODE_SHORTEST, {}),
                       ("WholeWord", N.MODE_WHOLEWORD, {"word_chars": default_word_chars()})):
    a = Automaton(mode, synth.config_keywords("C2"), True, **kw)
    if not os.environ.get("LATENCY_GC_AS_IS"):  # the dictionary's 10 000 arrays out of the collector's sight (see above)
        gc.collect()
        gc.freeze()
    for n in ORDER:
        hay = synth.haystack(5, n)
        a.match_host(hay, True)
        if calls_before == 0 and os.environ.get("LATENCY_SLEEP"):  # (is an outlier a matter of calls or of time? sleep after the process's first call)
            time.sleep(float(os.environ["LATENCY_SLEEP"]))
        reps = 200 if n < (1 << 20) else 50
        ts, stamps = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = a.match_host(hay, True)
            ts.append(time.perf_counter() - t0)
            stamps.append(t0)
        ts = np.array(ts) * 1e6  # (median and p99; what the slowest call is: tools/latency_outlier.py)
        print("%-20s n=%8d  %8.1f us per call (median; p99 %.1f, mean %.1f, max %.1f)  (%d matches)" % (
            name, n, np.median(ts), np.percentile(ts, 99), ts.mean(), ts.max(), len(r)))
        # every call of more than a millisecond: its index in this loop and among all calls of the process
        for i in np.flatnonzero(ts > 1000.0).tolist():
            print("    slow call: %.1f us at index %d of this loop, call %d of the process, %.3f s after the start" % (
                ts[i], i, calls_before + 1 + i, stamps[i] - T0))
        calls_before += 1 + reps
