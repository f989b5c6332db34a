#!/usr/bin/env python3
"""Runs inside a process that has libasan preloaded (tests/test_sanitizers.py): drives the sanitizer build of the host builder
over a set of dictionaries and writes what it built to OUTDIR/<case>.npz for the parent to compare with the product library.
usage: san_worker.py LIB OUTDIR"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases():
    """(name, mode, keywords, case_sensitive, word_chars) -- deterministic; the parent builds the same with the product library."""
    from ahocorasick_amd import synth
    from ahocorasick_amd.unicode_tables import default_word_chars
    W = default_word_chars()
    yield "ac_dense", 0, synth.random_keywords(11, 3000, 2, 9), True, None
    yield "ac_ci_wide", 0, synth.random_keywords(12, 800, 1, 7, table=np.arange(0x0391, 0x03C9, dtype=np.uint16)), False, None
    yield "ac_edge", 0, ["", None, "a", "a", "ab", "b" * 300], True, None
    yield "ac_sparse_all_units", 0, [np.array([i], dtype=np.uint16) for i in range(0, 65536, 1)], True, None
    yield "longest_prefix_closed", 1, synth.prefix_closed_keywords(1004, 3000, word_len=200), True, None
    yield "longest_dna", 1, synth.random_keywords(13, 500, 2, 30, table=np.array([ord(c) for c in "acgt"], dtype=np.uint16)), True, None
    yield "shortest", 3, synth.random_keywords(14, 500, 2, 20, table=synth.ALPHA_LOWER[:3]), True, None
    yield "wholeword_ci", 2, synth.mixed_script_words(1005, 3000), False, W
    wc = np.zeros(65536, np.uint8)
    for ch in "ABx":
        wc[ord(ch)] = 1
    yield "wholeword_fold_inconsistent", 2, ["A", "AB", "BA", "x"], False, wc
    sp = np.array([32], dtype=np.uint16)
    words = synth.mixed_script_words(1006, 400)
    yield "wwlongest", 4, list(words) + [np.concatenate([words[i], sp, words[i + 1]]) for i in range(0, 200, 2)], False, W
    yield "wholeword_nonword_error", 2, ["ok", "a b"], True, W


def pack(keywords):
    from ahocorasick_amd.strings import _pack
    return _pack(keywords)


def main():
    lib, outdir = sys.argv[1], sys.argv[2]
    from ahocorasick_amd.unicode_tables import java_lower_table
    L = ctypes.CDLL(lib)
    vp = ctypes.c_void_p
    L.san_build.restype = ctypes.c_int
    L.san_build.argtypes = [ctypes.c_int, vp, vp, ctypes.c_uint32, ctypes.c_int, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64)]
    L.san_free.argtypes = [vp]
    L.san_info.argtypes = [vp, vp]
    L.san_tables.restype = ctypes.c_int
    L.san_tables.argtypes = [vp] * 7
    lower = np.ascontiguousarray(java_lower_table(), dtype=np.uint16)
    ptr = lambda a: a.ctypes.data_as(vp) if a is not None else None  # noqa: E731
    for name, mode, kws, cs, wc in cases():
        units, off = pack(kws)
        h = vp()
        bad = ctypes.c_int64(-1)
        wcc = None if wc is None else np.ascontiguousarray(wc, dtype=np.uint8)
        rc = L.san_build(mode, ptr(units), ptr(off), len(off) - 1, 1 if cs else 0, None if cs else ptr(lower), ptr(wcc), ctypes.byref(h),
                         ctypes.byref(bad))
        if rc != 0:
            np.savez(os.path.join(outdir, name + ".npz"), rc=rc, bad=bad.value)
            continue
        info = np.zeros(12, np.int64)
        L.san_info(h, ptr(info))
        ns, nc, dense = int(info[0]), int(info[1]), int(info[2])
        cls = np.zeros(65536, np.uint16)
        dfa = np.zeros(ns * nc, np.uint32) if dense else None
        arrs = [np.zeros(ns, np.uint32) for _ in range(4)]
        assert L.san_tables(h, ptr(cls), ptr(dfa), *[ptr(a) for a in arrs]) == 0
        L.san_free(h)
        np.savez(os.path.join(outdir, name + ".npz"), rc=0, info=info, cls=cls, dfa=dfa if dfa is not None else np.zeros(0, np.uint32),
                 out_len=arrs[0], out_link=arrs[1], out_id=arrs[2], depth=arrs[3])
    print("san_worker: done")


if __name__ == "__main__":
    main()
