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
for name, kws, cs, d_hay in ([] if os.environ.get("LONGEST_SHAPES_C4_ONLY") else cases):
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


# ---- round 6: config 4's shape space (the review's item 4): the same prefix-closed dictionary with Map records, over four letters, and
# with stray units in the text -- which kernel each takes, the whole pipeline's time, and its share of the HBM peak by the
# contract's formula (2 bytes per unit + 8 / 12 per record)
import ctypes  # noqa: E402


def synth_text(table, seed, n_units):
    d = torch.empty(n_units, dtype=torch.int16, device="cuda")
    tab = np.ascontiguousarray(table, dtype=np.uint16)
    N.check(N.lib().acgpu_synth_fill(d.data_ptr(), n_units, 0, seed, tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
    return d


def prefix_closed(alphabet, seed, n_kw, word_len=1000):
    r = np.random.default_rng(seed)
    al = np.array([ord(c) for c in alphabet], dtype=np.uint16)
    seen, out, first = set(), [], True
    while len(out) < n_kw:
        w = np.full(word_len, al[0], dtype=np.uint16) if first else al[r.integers(0, len(al), word_len)]
        first = False
        for ln in range(1, word_len + 1):
            key = w[:ln].tobytes()
            if key not in seen:
                seen.add(key)
                out.append(w[:ln].copy())
                if len(out) >= n_kw:
                    break
    for c in al:  # (every letter a keyword, as in config 4)
        if bytes(np.array([c], dtype=np.uint16).tobytes()) not in seen:
            out.append(np.array([c], dtype=np.uint16))
    return out


n4 = 1 << 29
c4 = synth.config_keywords("C4")
d_ab = synth_text(synth.ALPHA_AB_75, synth.CONFIGS["C4"]["hay_seed"], n4)
d_stray = d_ab.clone()
g2 = torch.Generator(device="cuda"); g2.manual_seed(11)
d_stray[torch.randint(0, n4, (n4 // 1000,), device="cuda", generator=g2)] = ord("x")  # 0.1 % of the units outside the alphabet
acgt = prefix_closed("acgt", 21, 50000)
d_acgt = synth_text(np.array([ord(c) for c in "aaaaaaccgt"], dtype=np.uint16), 77, n4)  # P(a) = 0.6: long runs of the a, aa, aaa ... family
out4 = torch.empty((n4 // 2, 3), dtype=torch.int32, device="cuda")
for name, kws, d_hay, with_ids in (("config 4: {a,b} prefix-closed 50 k, Set records", c4, d_ab, False),
                                   ("config 4's dictionary, Map records", c4, d_ab, True),
                                   ("config 4's dictionary, 0.1 % of the text's units outside the alphabet, Set", c4, d_stray, False),
                                   ("{a,c,g,t} prefix-closed 50 k keywords, Set records", acgt, d_acgt, False),
                                   ("{a,c,g,t} prefix-closed 50 k keywords, Map records", acgt, d_acgt, True)):
    a = Automaton(N.MODE_LONGEST, kws, True)
    for form in ((0, 3) if os.environ.get("LONGEST_SHAPES_FORMS") else (0,)):  # 3: neither k_longest_bits nor k_longest_follow (the walk pipeline)
        N.set_tunable("longest_form", form)
        ms = []
        for i in range(4):
            nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n4, with_ids, out4.data_ptr(), n4 // 2, stream=st, profile=True)
            assert rc == 0, rc
            if i:
                ms.append(prof["scan_ms"] + prof["finalize_ms"])
        alg = 2.0 * n4 + (12 if with_ids else 8) * nm
        t = float(np.median(ms))
        print("%-82s %8.3f ms per 2^29 units  %9d records  %5.1f %% of 8 TB/s  %s%s" % (name, t, nm, 100.0 * alg / (t * 1e-3) / 8e12, prof["scan_kernel"],
                                                                                     "  (longest_form 3)" if form else ""), flush=True)
    N.set_tunable("longest_form", 0)
