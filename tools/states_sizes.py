"""Development tool: k_ac_states against the size of the automaton (subsets of the README word list on the same text): is the walk bound by misses?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
words = synth.readme_dictionary()
n = 1 << 28
block = synth.readme_text(2006, 1 << 25, words)
d_hay = torch.from_numpy(block.view(np.int16)).cuda().repeat(n // block.size)
st = torch.cuda.current_stream().cuda_stream
N.set_tunable("all_form", 2)
d_out = torch.empty((n * 2, 2), dtype=torch.int32, device="cuda")
rng = np.random.default_rng(1)
for k in (1000, 10000, 50000, len(words)):
    idx = rng.permutation(len(words))[:k]
    a = Automaton(N.MODE_ALL, [words[i] for i in idx], True)
    ms = []
    for i in range(3):
        nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, False, d_out.data_ptr(), n * 2, stream=st, profile=True)
        ms.append((prof["scan_ms"], prof["finalize_ms"]))
    print("%6d words, %8d states: states %.3f ms, records %.3f ms, %d records" % (k, a.info()["n_states"], min(ms)[0], min(ms)[1], nm), flush=True)
