#!/usr/bin/env python3
"""Development tool: amortised cost per haystack of acgpu_match_batch_u16 on the reference's published workload shape -- many
paragraph-sized inputs against a large dictionary (R/README.md:130-148: one paragraph, 235 k words, 3.6 us per match() call on
the reference's JVM) -- next to one acgpu_match_u16 call per haystack."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import _native as N, synth  # noqa: E402
from ahocorasick_amd.strings import Automaton  # noqa: E402

n_hay = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
kws = synth.random_keywords(1002, 100000, 4, 12)
a = Automaton(N.MODE_ALL, kws, True)
rng = np.random.default_rng(5)
text = synth.haystack(2002, n_hay * 600)
cuts = np.concatenate([[0], np.cumsum(rng.integers(400, 600, n_hay))])
hays = [text[cuts[i]:cuts[i + 1]] for i in range(n_hay)]
r = a.match_batch(hays, False)
times = []
for _ in range(5):
    t0 = time.perf_counter()
    r = a.match_batch(hays, False, cap=len(r) + 16)
    times.append(time.perf_counter() - t0)
dt = float(np.median(times))
print("acgpu_match_batch_u16: %d haystacks of ~500 units, 100 k keywords, %d records: %.3f ms per call = %.3f us per haystack" % (
    n_hay, len(r), dt * 1e3, dt * 1e6 / n_hay))
t0 = time.perf_counter()
for h in hays[:500]:
    a.match_host(h, False)
print("acgpu_match_u16 per haystack: %.1f us per call" % ((time.perf_counter() - t0) / 500 * 1e6))
