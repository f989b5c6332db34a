import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton, Stream
kws = synth.config_keywords("C2")
auto = Automaton(N.MODE_ALL, kws, True)
n, chunk = 1 << 26, 1 << 22
hay = synth.haystack(2002, n)
for rep in range(3):
    N.set_tunable("tile_debug", (1 << 42) if rep == 2 else 0)
    st = Stream(auto, with_ids=True, pipelined=True)
    t0 = time.perf_counter(); ts = []
    for o in range(0, n, chunk):
        t1 = time.perf_counter()
        st.feed(hay[o:o + chunk], final=o + chunk >= n, cap=chunk // 8)
        ts.append((time.perf_counter() - t1) * 1e6)
    print("pipelined total %.1f ms (%.1f GB/s); per feed us:" % ((time.perf_counter() - t0) * 1e3, 2.0 * n / (time.perf_counter() - t0) / 1e9), ["%.0f" % t for t in ts], flush=True)
    st.close()
