#!/usr/bin/env python3
"""Where do the rare multi-millisecond acgpu_match_u16 calls come from (development tool)?  Times every call of a long loop and
prints the calls beyond 1 ms with their index since the start of the process, for several orders of sizes: an outlier that
comes at a fixed number of calls (or seconds) into the process, whatever the call, is the runtime's; one that follows a size or a
path is the library's."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import _native as N, synth  # noqa: E402
from ahocorasick_amd.strings import Automaton  # noqa: E402

order = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,4096,65536").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
t_start = time.perf_counter()
a = Automaton(N.MODE_ALL, synth.config_keywords("C2"), True)
calls = 0
for n in order:
    hay = synth.haystack(5, n)
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        a.match_host(hay, True)
        dt = time.perf_counter() - t0
        calls += 1
        ts.append(dt)
        if dt > 1e-3 and i > 0:
            print("  n=%d call %d of this size, %d of the process, %.1f s after start: %.1f ms" % (n, i, calls, time.perf_counter() - t_start, dt * 1e3))
    ts = np.array(ts[1:]) * 1e6
    print("n=%8d  median %.1f us  p99 %.1f  max %.1f" % (n, np.median(ts), np.percentile(ts, 99), ts.max()))
