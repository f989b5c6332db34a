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
