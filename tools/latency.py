#!/usr/bin/env python3
"""Fixed cost of one acgpu_match_u16 call on short haystacks (development tool)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import _native as N, synth  # noqa: E402
from ahocorasick_amd.strings import Automaton  # noqa: E402
from ahocorasick_amd.unicode_tables import default_word_chars  # noqa: E402

for name, mode, kw in (("AhoCorasick C2 dict", N.MODE_ALL, {}), ("Longest", N.MODE_LONGEST, {}), ("Shortest", N.MODE_SHORTEST, {}),
                       ("WholeWord", N.MODE_WHOLEWORD, {"word_chars": default_word_chars()})):
    a = Automaton(mode, synth.config_keywords("C2"), True, **kw)
    for n in (64, 4096, 1 << 16, 1 << 20):
        hay = synth.haystack(5, n)
        a.match_host(hay, True)
        reps = 200 if n < (1 << 20) else 50
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = a.match_host(hay, True)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6  # (median and p99; what the slowest call is: tools/latency_outlier.py)
        print("%-20s n=%8d  %8.1f us per call (median; p99 %.1f, mean %.1f, max %.1f)  (%d matches)" % (
            name, n, np.median(ts), np.percentile(ts, 99), ts.mean(), ts.max(), len(r)))
