"""ctypes front-end for oracle/ac_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (ahocorasick_amd/) never does.

Parity status: pinned against the reference's deterministic test fixtures
(tests/golden/reference_fixtures.json); case-insensitive / non-ASCII word-char
behaviour is UNPINNED (no reference test covers it, no JDK here to run it).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")  # ORACLE_LIB: the sanitizer build

FAM_AC, FAM_LONGEST, FAM_WHOLEWORD, FAM_SHORTEST, FAM_WWLONGEST = 0, 1, 2, 3, 4
E_ILLEGAL_ARGUMENT = -2


class IllegalArgumentException(ValueError):
    """Mirror of the java.lang.IllegalArgumentException thrown by the WholeWord ctors
    (S/WholeWordMatchMap.java:263-267)."""


def build_lib(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "ac_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) and not os.environ.get("ORACLE_LIB"):
            build_lib()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
        L.oracle_build.restype = ctypes.c_int
        L.oracle_build.argtypes = [ctypes.c_int, vp, vp, u32, ctypes.c_int, vp, vp, ctypes.POINTER(vp),
                                   ctypes.POINTER(i64)]
        L.oracle_free.restype = None
        L.oracle_free.argtypes = [vp]
        L.oracle_match.restype = i64
        L.oracle_match.argtypes = [vp, vp, i32, vp, i64, i64]
        L.oracle_match_readable.restype = i64
        L.oracle_match_readable.argtypes = [vp, vp, i32, i32, vp, i64, i64]
        L.oracle_match_count.restype = i64
        L.oracle_match_count.argtypes = [vp, vp, i32]
        L.oracle_set_map_flavour.restype = None
        L.oracle_set_map_flavour.argtypes = [vp, ctypes.c_int]
        L.oracle_num_nodes.restype = i64
        L.oracle_num_nodes.argtypes = [vp, ctypes.c_int]
        L.oracle_queue_new.restype = vp
        L.oracle_queue_free.argtypes = [vp]
        L.oracle_queue_push.restype = ctypes.c_int
        L.oracle_queue_push.argtypes = [vp, i32, i32]
        L.oracle_queue_match_and_clear.restype = i64
        L.oracle_queue_match_and_clear.argtypes = [vp, i32, vp, i64]
        L.oracle_trim.argtypes = [vp, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        _lib = L
    return _lib


def utf16(s):
    """Python str (or array-like of code units) -> np.uint16 UTF-16 code units, as Java's String holds them."""
    if isinstance(s, str):
        return np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
    return np.ascontiguousarray(s, dtype=np.uint16)


def pack_keywords(keywords):
    """list of str / None / uint16 arrays -> (units, offsets[n+1]).  None and "" both become empty ranges
    (the reference skips both, S/AhoCorasickSet.java:27)."""
    parts = [utf16(k) if k is not None else np.zeros(0, np.uint16) for k in keywords]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    if parts:
        off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    units = np.concatenate(parts) if parts else np.zeros(0, np.uint16)
    if units.size == 0:
        units = np.zeros(1, np.uint16)
    return np.ascontiguousarray(units, dtype=np.uint16), off


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    """One reference matcher instance (family = FAM_AC / FAM_LONGEST / FAM_WHOLEWORD / FAM_SHORTEST)."""

    def __init__(self, family, keywords, case_sensitive=True, lower=None, word_chars=None, packed=None, map_flavour=False):
        """map_flavour: restate the *Map class where its String loop differs from the *Set class's -- only
        WholeWordLongestMatchMap's case-insensitive loop does (S/WholeWordLongestMatchMap.java:283,288 fold in the skip
        loops, S/WholeWordLongestMatchSet.java:151,156 do not); the match(Readable) loops exist for Maps only."""
        L = lib()
        units, off = packed if packed is not None else pack_keywords(keywords)
        self._keep = (units, off)
        lower = None if lower is None else np.ascontiguousarray(lower, dtype=np.uint16)
        word_chars = None if word_chars is None else np.ascontiguousarray(word_chars, dtype=np.uint8)
        h = ctypes.c_void_p()
        err_kw = ctypes.c_int64(-1)
        rc = L.oracle_build(family, _ptr(units), _ptr(off), len(off) - 1, 1 if case_sensitive else 0, _ptr(lower),
                            _ptr(word_chars), ctypes.byref(h), ctypes.byref(err_kw))
        if rc == E_ILLEGAL_ARGUMENT:
            raise IllegalArgumentException("keyword %d contains non-word characters." % err_kw.value)
        if rc != 0:
            raise MemoryError("oracle_build rc=%d" % rc)
        self._h = h
        self.family = family
        self.map_flavour = bool(map_flavour)
        if map_flavour:
            L.oracle_set_map_flavour(h, 1)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().oracle_free(h)
            self._h = None

    def match(self, haystack, stop_after=-1, cap=None):
        """Returns an (n,3) int32 array of (start, end, keyword_index) in the order the reference would call
        the listener.  stop_after=k: the listener returns false on its k-th call."""
        hay = utf16(haystack)
        n = int(hay.size)
        if n == 0:
            hay = np.zeros(1, np.uint16)
        L = lib()
        if cap is None:
            cap = max(1024, n // 4)
        while True:
            out = np.empty((cap, 3), dtype=np.int32)
            cnt = L.oracle_match(self._h, _ptr(hay), n, _ptr(out), cap, stop_after)
            if cnt <= cap:
                return out[:cnt].copy()
            cap = int(cnt)

    def match_readable(self, haystack, bufsize=1024, stop_after=-1, positions=False):
        """match(Readable, ReadableMatchListener): the keyword indices (values) in listener-call order.  positions=True:
        (n,3) records (start, end, value) -- the word matchers' Readable loops annotated with the positions the String loops'
        arithmetic gives (the other families ARE the String loops)."""
        hay = utf16(haystack)
        n = int(hay.size)
        if n == 0:
            hay = np.zeros(1, np.uint16)
        cap = max(1024, n // 4)
        while True:
            out = np.empty((cap, 3), dtype=np.int32)
            cnt = lib().oracle_match_readable(self._h, _ptr(hay), n, bufsize, _ptr(out), cap, stop_after)
            if cnt <= cap:
                return out[:cnt].copy() if positions else out[:cnt, 2].copy()
            cap = int(cnt)

    def count(self, haystack_units):
        """match() with the no-op listener; returns the number of listener calls (timed CPU baseline)."""
        hay = np.ascontiguousarray(haystack_units, dtype=np.uint16)
        return int(lib().oracle_match_count(self._h, _ptr(hay), int(hay.size)))

    def num_nodes(self, which=0):
        return int(lib().oracle_num_nodes(self._h, which))


class MatchQueue:
    """S/SetMatchQueue.java, for pinning T/MatchQueueTest.java."""

    def __init__(self):
        self._q = ctypes.c_void_p(lib().oracle_queue_new())

    def push(self, length, idx):
        return bool(lib().oracle_queue_push(self._q, length, idx))

    def match_and_clear(self, purge_to):
        out = np.empty((4096, 3), dtype=np.int32)
        n = lib().oracle_queue_match_and_clear(self._q, purge_to, _ptr(out), 4096)
        return [(int(s), int(e)) for s, e, _ in out[:n]]

    def __del__(self):
        if getattr(self, "_q", None):
            lib().oracle_queue_free(self._q)
            self._q = None


def trim(keyword, word_chars):
    """WordCharacters.trim (S/WordCharacters.java:41-62)."""
    u = utf16(keyword)
    ws, we = ctypes.c_int64(0), ctypes.c_int64(0)
    wc = np.ascontiguousarray(word_chars, dtype=np.uint8)
    lib().oracle_trim(_ptr(wc), _ptr(u if u.size else np.zeros(1, np.uint16)), int(u.size), ctypes.byref(ws),
                      ctypes.byref(we))
    return u[ws.value:we.value]
