"""Development tool: Set against Map records on the README word list (LongestMatch through k_longest_follow, AhoCorasick through k_ac_states)."""
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
for name, mode, cap in (("LongestMatch", N.MODE_LONGEST, n // 2), ("AhoCorasick", N.MODE_ALL, n * 2)):
    N.set_tunable("all_form", 2 if mode == N.MODE_ALL else 0)
    a = Automaton(mode, words, True)
    d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    for ids in (False, True):
        ms = []
        for i in range(3):
            nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, ids, d_out.data_ptr(), cap, stream=st, profile=True)
            ms.append((prof["scan_ms"], prof["finalize_ms"]))
        print("%-13s %s records: scan %.3f + finalize %.3f ms per 2^28 units, %d records, %s" % (name, "Map" if ids else "Set", min(ms)[0], min(ms)[1], nm, prof["scan_kernel"]), flush=True)
N.set_tunable("all_form", 0)
