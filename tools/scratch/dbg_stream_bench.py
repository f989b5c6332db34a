import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.dist import ShardedMatcher
from ahocorasick_amd.strings import Automaton, Stream
kws = synth.config_keywords("C2")
auto = Automaton(N.MODE_ALL, kws, True)
n = 1 << 27
m = ShardedMatcher(auto, n, with_ids=True, cap=n // 128, overlap=True)
tab = np.ascontiguousarray(synth.ALPHA_LOWER)
N.check(N.lib().acgpu_synth_fill(m.own_ptr(), n, 0, 2002, tab.ctypes.data_as(ctypes.c_void_p), len(tab), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fill")
torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "steps"
if which == "steps":
    for _ in range(5): m.step()
    m.finish()
print(bench.end_to_end_stream(auto, m, True, n))
# Longest n=64 per-call distribution
a = Automaton(N.MODE_LONGEST, kws, True)
for nn in (64, 65, 4096):
    hay = synth.haystack(5, nn)
    a.match_host(hay, True)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); a.match_host(hay, True); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts)
    print("Longest n=%d: median %.1f mean %.1f max %.1f us; > 100 us: %d calls" % (nn, np.median(ts), ts.mean(), ts.max(), (ts > 100).sum()))
