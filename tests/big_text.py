#!/usr/bin/env python3
"""Position arithmetic at the size limit (under tests/ because it checks against the oracle; also runnable on its own).  A haystack of just under 2^31 units (the ABI's limit) is scanned whole
and as four shards by the same automaton (records must concatenate to the same list), and the last 2^20 units are compared with
the oracle.  AhoCorasick (config 2's dictionary: the tile kernel, and k_ac_states with its 8 GB of states), WholeWord (config 5's dictionary on letters + spaces), Longest (config 4)."""
import ctypes, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton
from ahocorasick_amd.unicode_tables import default_word_chars, java_lower_table
from oracle.oracle import Oracle, FAM_AC, FAM_WHOLEWORD, FAM_LONGEST

n = (1 << 31) - 4096 - 24
stream = None
d_hay = None

def fill(tab, seed):
    tab = np.ascontiguousarray(np.asarray(tab, dtype=np.uint16))
    N.check(N.lib().acgpu_synth_fill(d_hay.data_ptr(), n, 0, seed, tab.ctypes.data_as(ctypes.c_void_p), len(tab), None), "synth")
    torch.cuda.synchronize()

def run(a, cap, own=None, with_ids=True, chain_entry=None, want_exit=False):
    d_out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
    kw = {} if own is None else {"own": own}
    if chain_entry is not None:
        kw["chain_entry"] = chain_entry
    n_out, rc, _, ex = a.match_device(d_hay.data_ptr(), n, with_ids, d_out.data_ptr(), cap, stream=stream, **kw)
    assert rc == N.OK, rc
    return (d_out[:n_out], ex) if want_exit else d_out[:n_out]

def main():
  global stream, d_hay
  stream = torch.cuda.current_stream().cuda_stream
  d_hay = torch.empty(n, dtype=torch.int16, device="cuda")
  cases = [
      ("AhoCorasick C2", N.MODE_ALL, synth.config_keywords("C2"), True, synth.ALPHA_LOWER, FAM_AC, n // 200),
      ("AhoCorasick C2 through k_ac_states", N.MODE_ALL, synth.config_keywords("C2"), True, synth.ALPHA_LOWER, FAM_AC, n // 200),
      ("WholeWord C5 words on letters+space", N.MODE_WHOLEWORD, synth.config_keywords("C5"), False,
       list(synth.ALPHA_LOWER[:8]) + [32, 32], FAM_WHOLEWORD, n // 4),
      ("Longest C4", N.MODE_LONGEST, synth.config_keywords("C4"), True, synth.ALPHA_AB_75, FAM_LONGEST, n // 4),
  ]
  for name, mode, kws, cs, tab, fam, cap in cases:
      fill(tab, 4242)
      wc = default_word_chars() if mode == N.MODE_WHOLEWORD else None
      a = Automaton(mode, kws, cs, word_chars=wc)
      N.set_tunable("all_form", 2 if "k_ac_states" in name else 0)  # (2: the state form whatever the text's density of matches)
      whole = run(a, cap)
      cuts = [0, n // 4 + 3, n // 2 + 1, 3 * (n // 4) + 5, n]
      if mode != N.MODE_LONGEST:
          parts = [run(a, cap // 2, own=(cuts[i], cuts[i + 1])) for i in range(4)]
      else:  # the greedy chain: every shard enters where the one before it left
          parts, entry = [], 0
          for i in range(4):
              r, entry = run(a, cap // 2, own=(cuts[i], cuts[i + 1]), chain_entry=max(entry, cuts[i]), want_exit=True)
              parts.append(r)
      cat = torch.cat(parts)
      assert cat.shape == whole.shape and bool((cat == whole).all()), name
      tail = 1 << 20
      host = d_hay[n - tail - 4096:].cpu().numpy().view(np.uint16)
      orc = Oracle(fam, kws, case_sensitive=cs, lower=java_lower_table(), word_chars=wc)
      want = orc.match(host)
      base = n - tail - 4096
      got = whole[whole[:, 0] >= base + 2048].cpu().numpy()  # (records that begin behind the oracle window's warm-up; filtered on the device)
      want = want[want[:, 0] >= 2048].copy()
      want[:, :2] += base
      if mode == N.MODE_LONGEST:
          # the greedy chain depends on the whole text, but a chain that enters at a position the text's chain VISITS is the rest
          # of that chain: the oracle starts at the start of a device record (a chain position) and must deliver the very same
          # records from there on
          p0 = int(got[0, 0])
          want = orc.match(host[p0 - base:], cap=tail)
          want[:, :2] += p0
          assert got.shape == want.shape and (got == want).all(), name
      else:
          assert got.shape == want.shape and (got == want).all(), name
      print("%-40s n=%d records=%d max_end=%d sha(last 2^20 records)=%s" % (name, n, len(whole), int(whole[:, 1].max()),
                                                                        hashlib.sha256(whole[-(1 << 20):].cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
      del whole, parts
      N.set_tunable("all_form", 0)
      torch.cuda.empty_cache()
  print("big text ok")
  d_hay = None
  torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
