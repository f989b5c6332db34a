#!/usr/bin/env python3
"""Development tool: LongestMatchSet over dictionaries that do not get the range-class walk (case-insensitive, wide alphabets) --
which kernel each takes and what 2^28 units cost."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton

n = 1 << 28
rng = np.random.default_rng(5)
st = torch.cuda.current_stream().cuda_stream
words = synth.readme_dictionary()
block = synth.readme_text(2006, 1 << 25, words)
d_readme = torch.from_numpy(block.view(np.int16)).cuda().repeat(n // block.size)
cjk = np.arange(0x4E00, 0x4E00 + 3000, dtype=np.uint16)
g = torch.Generator(device="cuda"); g.manual_seed(9)
d_cjk = torch.from_numpy(cjk.view(np.int16)).cuda()[torch.randint(0, len(cjk), (n,), device="cuda", generator=g, dtype=torch.int32).long()].contiguous()
cjk_kws = [rng.choice(cjk, size=int(rng.integers(2, 6))).astype(np.uint16) for _ in range(20000)]
cases = [("README words, case-sensitive", words, True, d_readme), ("README words, case-insensitive", words, False, d_readme),
         ("README words of 4 and more letters, case-insensitive", [w for w in words if len(w) > 3], False, d_readme),
         ("3000 CJK units, 20 k keywords of 2-5 units", cjk_kws, True, d_cjk)]
cap = n
d_out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
for name, kws, cs, d_hay in cases:
    a = Automaton(N.MODE_LONGEST, kws, cs)
    for form, knob, ru in (("chain positions only", 0, 0), ("... run-up 256", 0, 256), ("... run-up 128", 0, 128), ("walk from every position", 2, 0)):  # longest_form 2: k_longest_follow never
        N.set_tunable("longest_form", knob)
        N.set_tunable("region_units", ru)
        ms = []
        for i in range(3):
            nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, False, d_out.data_ptr(), cap, stream=st, profile=True)
            assert rc == 0, rc
            ms.append((prof["scan_ms"] + prof["finalize_ms"], prof["scan_ms"]))
        print("%-58s %-26s %8.3f ms per 2^28 units (scan %.3f) %d records  %s" % (name, form, min(ms)[0], min(ms)[1], nm, prof["scan_kernel"]), flush=True)
    N.set_tunable("longest_form", 0)
    N.set_tunable("region_units", 0)
