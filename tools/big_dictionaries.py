"""Development tool: AhoCorasick scan time of config 2's haystack under 30 k / 100 k / 300 k random keywords, with the second level
in global memory (the BIG form of k_ac_tile) and with the saturated one in LDS (tile_debug bit 2^30) -- DESIGN.md 7."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
n = 1 << 29
d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
tab = np.ascontiguousarray(synth.ALPHA_LOWER)
N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, 2002, tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
torch.cuda.synchronize()
cap = n // 16
d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
for name, kws in (("30k", synth.random_keywords(9, 30000, 4, 12)), ("100k", synth.random_keywords(10, 100000, 4, 12)), ("300k", synth.random_keywords(12, 300000, 4, 12))):
    a = Automaton(N.MODE_ALL, kws, True)
    for label, knobs in (("big L2 (global)", {}), ("LDS L2", {"tile_debug": 1 << 30})):
        for k, v in {"force_kernel": 0, "split_cand_div": 8, "tile_debug": 0}.items(): N.set_tunable(k, v)
        for k, v in knobs.items(): N.set_tunable(k, v)
        ts = []
        for r in range(4):
            nout, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, True, d_out.data_ptr(), cap, stream=torch.cuda.current_stream().cuda_stream, profile=True)
            if r: ts.append((prof["scan_ms"], prof["finalize_ms"]))
        print("%-5s %-12s scan %.3f ms  finalize(+verify) %.3f ms  n_out=%d rc=%d dens=%.3f %s" % (name, label, np.median([t[0] for t in ts]), np.median([t[1] for t in ts]), nout, rc, a.info()["filter_density"], prof["scan_kernel"]), flush=True)
