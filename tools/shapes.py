#!/usr/bin/env python3
"""Development tool: scan time of config 2's haystack (2^29 units) under dictionaries of other shapes -- which form of the
tile kernel each one gets and what it costs."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton

n = 1 << 29
kws = synth.config_keywords("C2")
d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
tab = np.ascontiguousarray(synth.ALPHA_LOWER)
N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, synth.CONFIGS["C2"]["hay_seed"], tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
torch.cuda.synchronize()
cap = n // 16  # (33.5 M records: every shape below but the last, whose 931 M records no buffer here holds)
d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
rng = np.random.default_rng(1)
def upper_some(k):
    k = np.array(k, dtype=np.uint16).copy()
    m = rng.integers(0, 2, k.size).astype(bool)
    k[m] -= 32
    return k
mixed = [upper_some(k) for k in kws]
def phrase(k):
    k = np.array(k, dtype=np.uint16).copy()
    if k.size >= 6:
        k[int(rng.integers(2, k.size - 2))] = int(rng.choice([32, 45, 48, 49, 50, 57]))
    return k
phrases = [phrase(k) for k in kws]
def upper_letters(k):
    k = k.copy()
    m = rng.integers(0, 2, k.size).astype(bool) & (k >= 97) & (k <= 122)
    k[m] -= 32
    return k
shapes = {
    "C2 case-sensitive (range classes)": (kws, True),
    "C2 case-insensitive (LUT classes)": (kws, False),
    "C2 mixed-case keywords, case-sensitive (merged ranges)": (mixed, True),
    "C2 mixed-case keywords, case-sensitive (52 classes, wide rows)": (mixed, True, {"no_merged_ranges": 1}),
    "C2 keywords as phrases (space / digit / hyphen inside), case-sensitive": (phrases, True),
    "C2 keywords as phrases, case-insensitive": (phrases, False),
    "C2 mixed-case phrases, case-sensitive": ([upper_letters(k) for k in phrases], True),
    "C2 mixed-case phrases, case-insensitive": ([upper_letters(k) for k in phrases], False),
    "C2 keywords as phrases, case-insensitive, class table form": (phrases, False, {"no_merged_ranges": 1}),
    "C2 + one keyword of 3 units": (list(kws) + [np.array([113, 120, 122], dtype=np.uint16)], True),
    "C2 + one keyword of 2 units": (list(kws) + [np.array([113, 120], dtype=np.uint16)], True),
    "C2 + one keyword of 1 unit": (list(kws) + [np.array([113], dtype=np.uint16)], True),
    "C2 mixed-case + one keyword of 2 units": (mixed + [np.array([81, 120], dtype=np.uint16)], True),
    "C2 phrases case-insensitive + one keyword of 2 units": (phrases + [np.array([81, 120], dtype=np.uint16)], False),
    "1000 keywords len 3-8 (K=3)": (synth.random_keywords(7, 1000, 3, 8), True),
    "100 keywords len 2-6 (K=2)": (synth.random_keywords(8, 100, 2, 6), True),
    "30k keywords len 4-12": (synth.random_keywords(9, 30000, 4, 12), True),
    "100k keywords len 4-12": (synth.random_keywords(10, 100000, 4, 12), True),
    "235k keywords len 2-14": (synth.random_keywords(11, 235000, 2, 14), True),
}
for name, spec in shapes.items():
    k, cs = spec[0], spec[1]
    knobs = spec[2] if len(spec) > 2 else {}
    for kn, v in knobs.items(): N.lib().acgpu_set_tunable(kn.encode(), v)
    a = Automaton(N.MODE_ALL, k, cs)
    for kn in knobs: N.lib().acgpu_set_tunable(kn.encode(), 0)
    ts = []
    for r in range(6):
        nout, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, True, d_out.data_ptr(), cap, stream=torch.cuda.current_stream().cuda_stream, profile=True)
        if r: ts.append(prof["scan_ms"])
    info = a.info()
    print("%-62s %.3f ms  n_out=%d rc=%d K=%d dens=%.4f %s%s" % (name, float(np.median(ts)), nout, rc, info["filter_k"], info["filter_density"], prof["scan_kernel"],
                                                          "  (capacity too small: the time is of a call that counts all records and stores the first %d)" % cap if rc == N.E_OVERFLOW else ""), flush=True)
