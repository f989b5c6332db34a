"""Shared helpers for the parity tests (test code only)."""
import numpy as np

from ahocorasick_amd.unicode_tables import default_word_chars, java_lower_table


def fixture_inputs(fx):
    """(haystack_units, keywords_as_list) for one golden fixture."""
    if "haystack_units" in fx:
        hay = np.array(fx["haystack_units"], dtype=np.uint16)
    else:
        hay = np.frombuffer(fx["haystack"].encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
    if fx.get("keywords_gen") == "all_single_units":
        kws = [np.array([i], dtype=np.uint16) for i in range(65536)]
    else:
        kws = fx["keywords"]
    return hay, kws


def rand_case(rng, alphabet, n_kw, max_len, hay_len, min_len=1):
    """Random dictionary + haystack over a tiny alphabet (forces overlaps, prefixes, fail chains)."""
    alphabet = np.asarray(alphabet, dtype=np.uint16)
    kws = []
    for _ in range(n_kw):
        ln = int(rng.integers(min_len, max_len + 1))
        kws.append(alphabet[rng.integers(0, len(alphabet), ln)])
    hay = alphabet[rng.integers(0, len(alphabet), hay_len)]
    return hay, kws


LOWER = java_lower_table()
WORD = default_word_chars()


def oracle_parallel(orc, hay, family, max_len, threads=None, cap_per_unit=0.25):
    """The single-threaded oracle on T chunks of one haystack in T threads (ctypes releases the GIL), stitched so that the
    result IS the oracle's record list for the whole text -- lets the full-size tests compare 100 % of the records:
      family "ac"        : a chunk is scanned from (max_len-1) units before it; a record belongs to the chunk that holds its
                           LAST unit (the sharding rule of SURVEY 8e; AhoCorasick restarted at the root max_len-1 units
                           earlier reports exactly the occurrences that end later);
      family "wholeword" : one unit of left context, max_len+1 units of right context; a record belongs to the chunk that
                           holds its FIRST unit (a run still going at the end of the right context is longer than every
                           keyword).
    Longest / Shortest are chains: run those whole, in one thread."""
    import os
    import threading
    n = int(hay.size)
    T = threads or min(32, os.cpu_count() or 1)
    T = max(1, min(T, n // (1 << 16) or 1))
    edges = np.linspace(0, n, T + 1).astype(np.int64)
    parts = [None] * T

    def work(i):
        lo, hi = int(edges[i]), int(edges[i + 1])
        if family == "ac":
            a = max(lo - (max_len - 1), 0)
            r = orc.match(hay[a:hi], cap=max(1024, int((hi - a) * cap_per_unit)))
            r[:, :2] += a
            parts[i] = r[r[:, 1] - 1 >= lo]
        else:
            a, b = max(lo - 1, 0), min(hi + max_len + 1, n)
            r = orc.match(hay[a:b], cap=max(1024, int((b - a) * cap_per_unit)))
            r[:, :2] += a
            parts[i] = r[(r[:, 0] >= lo) & (r[:, 0] < hi)]

    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return np.concatenate(parts)
