import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd import WholeWordMatchMap, LongestMatchSet
from ahocorasick_amd.strings import Automaton, utf16
kws = ["a", "aa", "aaa", "aaaa"]
m = WholeWordMatchMap(kws, [0, 1, 2, 3], True)
for hay in (" aaaaaaa ", " aaaa ", "aaaab", " aaaaaaa aaababababaabaa ", "aaaa", "aaaaa"):
    print(repr(hay), m.find_all(hay).tolist())
N.set_tunable("tile_debug", 1 << 41)
print("general", m.find_all(" aaaaaaa aaababababaabaa ").tolist())
N.set_tunable("tile_debug", 0)
a = Automaton(N.MODE_LONGEST, synth.config_keywords("C2"), True)
for n in (64, 65, 128, 1000, 4096):
    hay = synth.haystack(5, n)
    a.match_host(hay, True)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); a.match_host(hay, True); ts.append(time.perf_counter() - t0)
    print(n, ["%.0f" % (t * 1e6) for t in ts])
