"""Development tool: the AhoCorasick kernels on the README-sized dictionary (235 886 words) and on subsets of it -- which kernel
form each gets and what a 2^28-unit text costs ("The reference's dictionary scale")."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
words = synth.readme_dictionary()
n = 1 << 28
block = synth.readme_text(2006, 1 << 25, words)
d_hay = torch.from_numpy(block.view(np.int16)).cuda().repeat(n // block.size)
st = torch.cuda.current_stream().cuda_stream
for label, kws in (("all 235886", words), ("without the single letters", [w for w in words if len(w) > 1]), ("lengths >= 3", [w for w in words if len(w) > 2]),
                   ("lower-case, lengths >= 4", [w for w in words if len(w) > 3 and w[0] >= 97])):
    for fk in (0, 1, 9):  # 0: the tile kernel (all_form 1: never k_ac_states), 1: the DFA chunk scan, 9: k_ac_states (all_form 2)
        N.set_tunable("force_kernel", fk if fk < 9 else 0)
        N.set_tunable("all_form", 2 if fk == 9 else 1)
        a = Automaton(N.MODE_ALL, kws, True)
        info = a.info()
        cap = int(n * 1.75)
        d_out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
        ms = []
        for i in range(3):
            nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, False, d_out.data_ptr(), cap, stream=st, profile=True)
            assert rc == 0, rc
            ms.append(prof["scan_ms"] + prof["finalize_ms"])
        print("%-28s force_kernel=%d K=%d density=%.3f tile=%d: %8.3f ms per 2^28 units (scan %.3f) %d records %s" % (
            label, fk, info["filter_k"], info["filter_density"], info["tile_kernel"], min(ms), prof["scan_ms"], nm, prof["scan_kernel"][:40]), flush=True)
        del a, d_out
N.set_tunable("all_form", 0)
