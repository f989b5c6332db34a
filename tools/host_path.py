#!/usr/bin/env python3
"""Development tool: end-to-end rate of the host-buffer entry point acgpu_match_u16 (pageable H2D + scan + D2H)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
kws = synth.config_keywords("C2")
a = Automaton(N.MODE_ALL, kws, True)
hay = synth.haystack(2002, 1 << 24)
hay = np.tile(hay, n >> 24)
for _ in range(3):
    t = time.perf_counter()
    r = a.match_host(hay, True, cap=n // 64)
    dt = time.perf_counter() - t
    print("acgpu_match_u16: %d units, %d matches, %.1f ms, %.1f GB/s of UTF-16 incl. H2D/D2H" % (n, len(r), dt * 1e3, 2 * n / dt / 1e9))
