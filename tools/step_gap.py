#!/usr/bin/env python3
"""Development tool: what the per-step HIP events cost -- config 2's pipelined steps with and without profiling."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
from ahocorasick_amd.dist import ShardedMatcher

n = 1 << 29
a = Automaton(N.MODE_ALL, synth.config_keywords("C2"), True)
m = ShardedMatcher(a, n, with_ids=True, cap=n // 128, overlap=True)
tab = np.ascontiguousarray(synth.ALPHA_LOWER)
N.check(N.lib().acgpu_synth_fill(m.own_ptr(), n, 0, synth.CONFIGS["C2"]["hay_seed"], tab.ctypes.data_as(ctypes.c_void_p), len(tab),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fill")
torch.cuda.synchronize()
for prof in (True, False, True, False):
    for _ in range(5):
        m.step()
    m.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        m.step(profile=prof)
    m.finish()
    torch.cuda.synchronize()
    print("profile=%s: %.4f ms per step" % (prof, (time.perf_counter() - t0) / 60 * 1e3), flush=True)
