"""Development tool: k_ac_states under its timing switches (tile_debug bits 1: no state stores, 2: conditional loads, 4/8: chunks of 1024/256)."""
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
a = Automaton(N.MODE_ALL, [w for w in words if len(w) > 2], True)
cap = n
d_out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
for dbg in (0,):
    N.set_tunable("tile_debug", dbg)
    ms = []
    for i in range(3):
        nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, False, d_out.data_ptr(), cap, stream=st, profile=True)
        ms.append((prof["scan_ms"], prof["finalize_ms"]))
    print("tile_debug %d: states %.3f ms, records %.3f ms, %d records" % (dbg, min(ms)[0], min(ms)[1], nm), flush=True)
N.set_tunable("tile_debug", 0)
