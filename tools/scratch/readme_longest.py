import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
from ahocorasick_amd.unicode_tables import default_word_chars
words = synth.readme_dictionary()
n = 1 << 28
block = synth.readme_text(2006, 1 << 25, words)
d_hay = torch.from_numpy(block.view(np.int16)).cuda().repeat(n // block.size)
st = torch.cuda.current_stream().cuda_stream
cap = n // 2
d_out = torch.empty((cap, 2), dtype=torch.int32, device="cuda")
for label, knobs in (("dense", {}), ("sparse (hashed edges)", {"force_sparse": 1}), ("dense, general kernel", {"force_kernel": 1})):
    for k, v in {"force_sparse": 0, "force_kernel": 0}.items(): N.set_tunable(k, v)
    for k, v in knobs.items(): N.set_tunable(k, v)
    a = Automaton(N.MODE_LONGEST, words, True)
    info = a.info()
    ms = []
    for i in range(3):
        nm, rc, prof, _ = a.match_device(d_hay.data_ptr(), n, False, d_out.data_ptr(), cap, stream=st, profile=True)
        assert rc == 0, rc
        ms.append((prof["scan_ms"], prof["finalize_ms"]))
    print("Longest %-24s dense=%d table=%.0f MB: scan %.3f + chain %.3f ms per 2^28 units, %d records, %s" % (
        label, info["dense"], info["table_bytes"] / 1e6, min(m[0] for m in ms), min(m[1] for m in ms), nm, prof["scan_kernel"]), flush=True)
    del a
