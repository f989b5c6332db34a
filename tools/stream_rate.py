#!/usr/bin/env python3
"""Development tool: rate of the chunked entry (acgpu_stream_feed = match(Readable, ...)): config 2's haystack fed in 4 Mi-unit
chunks, as the Java facade does, for the AhoCorasick and the WholeWord family."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahocorasick_amd import _native as N, synth
from ahocorasick_amd.strings import Automaton, Stream
from ahocorasick_amd.unicode_tables import default_word_chars

n, chunk = 1 << 27, 1 << 22
for name, mode, kws, cs, hay in (
        ("AhoCorasick C2", N.MODE_ALL, synth.config_keywords("C2"), True, synth.haystack(2002, n)),
        ("WholeWord C5 words", N.MODE_WHOLEWORD, synth.config_keywords("C5"), False,
         np.asarray(list(synth.ALPHA_LOWER[:8]) + [32, 32], dtype=np.uint16)[np.random.default_rng(5).integers(0, 10, n)])):
    a = Automaton(mode, kws, cs, word_chars=default_word_chars() if mode == N.MODE_WHOLEWORD else None)
    for form in ("synchronous", "pipelined", "pipelined, chunks written into the reserved staging memory"):
        dts = []
        for rep in range(4):  # (first pass: the staging buffers are allocated; median of the other three)
            s = Stream(a, with_ids=True, pipelined=form != "synchronous")
            t0 = time.perf_counter()
            total = 0
            for o in range(0, n, chunk):
                c = hay[o:o + chunk]
                if form.endswith("memory"):  # (the producer's own write: a Reader would fill this buffer itself)
                    v = s.reserve(c.size)
                    np.copyto(v, c)
                    c = v
                total += len(s.feed(c, final=o + chunk >= n, cap=chunk // 8))
            dts.append(time.perf_counter() - t0)
            s.close()
        dt = float(np.median(dts[1:]))
        print("%-20s %-62s %d units in %d-unit chunks: %.1f ms, %.1f GB/s of UTF-16, %d records" % (name, form, n, chunk, dt * 1e3, 2.0 * n / dt / 1e9, total), flush=True)
