#!/usr/bin/env python3
"""Development tool: dictionaries over wide alphabets (more than 63 distinct units: bucketed tile classes, or the DFA chunk scan when
the filter rejects nothing) and the class-table forms of the tile kernel -- scan time per 2^28 units with the class table as LDS pages
(acgpu_build.cpp 7b) and, for A/B, looked up in global memory (builder tunable no_class_pages), and through the DFA chunk scan."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton

n = 1 << 28
rng = np.random.default_rng(3)
cjk = np.arange(0x4E00, 0x4E00 + 3000, dtype=np.uint16)
mixed_alpha = np.concatenate([cjk, np.tile(np.arange(97, 123, dtype=np.uint16), 115)])  # half CJK, half a-z


def words(alpha, count, lo, hi):
    return [rng.choice(alpha, size=int(rng.integers(lo, hi + 1))).astype(np.uint16) for _ in range(count)]


def fill(alpha, seed):  # units uniform over `alpha`, drawn on the device (acgpu_synth_fill takes tables of at most 64 units)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    tab = torch.from_numpy(np.ascontiguousarray(alpha).view(np.int16)).cuda()
    d = tab[torch.randint(0, len(alpha), (n,), device="cuda", generator=g, dtype=torch.int32).long()]
    torch.cuda.synchronize()
    return d.contiguous()


c2 = synth.config_keywords("C2")
def phrase(k):
    k = np.array(k, dtype=np.uint16).copy()
    if k.size >= 6:
        k[int(rng.integers(2, k.size - 2))] = int(rng.choice([32, 45, 48, 49, 50, 57]))
    return k
cases = [
    ("3000 CJK units, 20 k keywords of 3-8 units", cjk, words(cjk, 20000, 3, 8), True, {}),
    ("3000 CJK units, 100 k keywords of 3-8 units", cjk, words(cjk, 100000, 3, 8), True, {}),
    ("3000 CJK units, 20 k keywords of 2-8 units", cjk, words(cjk, 20000, 2, 8), True, {}),
    ("3000 CJK units, 20 k keywords of 1-4 units", cjk, words(cjk, 20000, 1, 4), True, {}),
    ("3000 CJK units, 3 k keywords of 2 units", cjk, words(cjk, 3000, 2, 2), True, {}),
    ("3000 CJK units, 20 k keywords of 2 units", cjk, words(cjk, 20000, 2, 2), True, {}),
    ("3000 CJK units, 100 k keywords of 2-3 units", cjk, words(cjk, 100000, 2, 3), True, {}),
    ("300 CJK units, 2 k keywords of 2-4 units", cjk[:300], words(cjk[:300], 2000, 2, 4), True, {}),
    ("CJK + a-z text, 20 k CJK keywords of 3-8 units", mixed_alpha, words(cjk, 20000, 3, 8), True, {}),
    ("CJK + a-z text, 10 k CJK and 10 k a-z keywords, case-insensitive", mixed_alpha,
     words(cjk, 10000, 3, 8) + words(np.arange(97, 123, dtype=np.uint16), 10000, 4, 10), False, {}),
    ("a-z text, config 2's keywords as phrases, case-insensitive, class-table form", synth.ALPHA_LOWER, [phrase(k) for k in c2], False,
     {"no_merged_ranges": 1}),
]
cap = n // 16
d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
for name, alpha, kws, cs, knobs in cases:
    d_hay = fill(alpha, 77)
    row = []
    for label, build_knobs, run_knobs in (("pages", {}, {}), ("global table", {"no_class_pages": 1}, {}), ("DFA scan", {}, {"force_kernel": 1}), ("tile kernel", {}, {"force_kernel": 2})):
        for kn, v in {**knobs, **build_knobs}.items(): N.set_tunable(kn, v)
        a = Automaton(N.MODE_ALL, kws, cs)
        for kn in {**knobs, **build_knobs}: N.set_tunable(kn, 0)
        for kn, v in run_knobs.items(): N.set_tunable(kn, v)
        ts = []
        for r in range(5):
            nout, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, True, d_out.data_ptr(), cap, stream=torch.cuda.current_stream().cuda_stream, profile=True)
            if r: ts.append(prof["scan_ms"])
        for kn in run_knobs: N.set_tunable(kn, 0)
        info = a.info()
        row.append("%s %.3f ms (%s)" % (label, float(np.median(ts)), prof["scan_kernel"][:40]))
        rec = (nout, rc)
    print("%-70s K=%d dens=%.3f n_out=%d rc=%d | %s" % (name, info["filter_k"], info["filter_density"], rec[0], rec[1], " | ".join(row)), flush=True)
    del d_hay
