"""Development tool: where the scan of a LARGE dictionary spends its time -- config 2's haystack under 100 k random keywords (and the
README word list on its text) with the ablation switches of the -DACGPU_ABLATION build (tools/build_variant.sh abl -DACGPU_ABLATION;
ACGPU_LIB=ahocorasick_amd/lib_abl/libacgpu.so python tools/big_ablation.py)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
assert N.set_tunable("ablation_build", 0) == 1, "needs the -DACGPU_ABLATION build"
n = 1 << 28
st = torch.cuda.current_stream().cuda_stream
d_rand = torch.empty(n, dtype=torch.int16, device="cuda")
tab = np.ascontiguousarray(synth.ALPHA_LOWER)
N.check(N.lib().acgpu_synth_fill(d_rand.data_ptr(), n, 0, 2002, tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
words = synth.readme_dictionary()
block = synth.readme_text(2006, 1 << 25, words)
d_readme = torch.from_numpy(block.view(np.int16)).cuda().repeat(n // block.size)
torch.cuda.synchronize()
cap = n
d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
variants = (("full", 0), ("no record emission", 32), ("no walk behind the K-gram node", 64), ("no walk, no emission", 96), ("no K-gram node load", 16 | 64 | 32),
            ("no verification at all", 1), ("stream + first level only", 1 | 2), ("stream only", 5))
for name, kws, d_hay in (("100 k random keywords, random a-z text", synth.random_keywords(10, 100000, 4, 12), d_rand),
                         ("README words of 3 and more letters, README text", [w for w in words if len(w) > 2], d_readme)):
    a = Automaton(N.MODE_ALL, kws, True)
    for label, bits in variants:
        N.set_tunable("tile_debug", bits)
        ts = []
        for r in range(4):
            nout, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, True, d_out.data_ptr(), cap, stream=st, profile=True)
            if r: ts.append(prof["scan_ms"])
        N.set_tunable("tile_debug", 0)
        print("%-52s %-34s scan %.3f ms per 2^28 units  n_out=%d rc=%d %s" % (name, label, float(np.median(ts)), nout, rc, prof["scan_kernel"][:40]), flush=True)
