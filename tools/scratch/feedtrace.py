import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton, Stream
n, chunk = 1 << 26, 1 << 22
a = Automaton(N.MODE_ALL, synth.config_keywords("C2"), True)
hay = synth.haystack(2002, n)
for rep in range(2):
    N.set_tunable("tile_debug", (1 << 42) if rep else 0)
    s = Stream(a, with_ids=True, pipelined=True)
    t0 = time.perf_counter(); ts = []
    for o in range(0, n, chunk):
        t1 = time.perf_counter()
        s.feed(hay[o:o + chunk], final=o + chunk >= n, cap=chunk // 8)
        ts.append((time.perf_counter() - t1) * 1e6)
    print("total %.1f ms; per feed us:" % ((time.perf_counter() - t0) * 1e3), ["%.0f" % t for t in ts])
    s.close()
# raw memcpy rate of this host, one thread
b = np.empty(chunk, np.uint16)
t0 = time.perf_counter()
for o in range(0, n, chunk):
    np.copyto(b, hay[o:o + chunk])
print("np.copyto pageable->pageable: %.1f GB/s" % (2.0 * n / (time.perf_counter() - t0) / 1e9))
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
