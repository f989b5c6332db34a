"""Host-side mirror of the reference's public API (package com.roklenarcic.util.strings) over libacgpu.so.

Same class names, constructor arguments, listener contract and error behaviour as the reference, so the parity
tests read like the reference's own tests:

  StringSet.match(haystack, SetMatchListener)                  S/StringSet.java:3-5
  StringMap.match(haystack, MapMatchListener)                  S/StringMap.java:5-9
  SetMatchListener.match(haystack, start, end) -> bool         S/SetMatchListener.java:6
  MapMatchListener.match(haystack, start, end, value) -> bool  S/MapMatchListener.java:6
  AhoCorasickSet(keywords, caseSensitive)                      S/AhoCorasickSet.java:16
  AhoCorasickMap(keywords, values, caseSensitive)              S/AhoCorasickMap.java:20
  LongestMatchSet / LongestMatchMap                            S/LongestMatchSet.java:15, S/LongestMatchMap.java
  WholeWordMatchSet / WholeWordMatchMap (+ wordCharacters[, toggleFlags])   S/WholeWordMatchMap.java:21-53

The matching itself always runs on the GPU through the C ABI (include/acgpu.h); the listener loop runs here, and
stops at the first listener call that returns False -- which is observationally what the reference does
(S/AhoCorasickSet.java:223-225).  The Thresholder constructor argument of the reference is accepted and ignored
(results-neutral node-representation knob).  StringMap.match also takes a Readable (an object with read(n), or an
iterable of chunks) with a value-only ReadableMatchListener, S/StringMap.java:6-8: match_readable over acgpu_stream_*.
"""
import ctypes
import os

import numpy as np

from . import _native as N
from .unicode_tables import (default_word_chars, java_lower_table, word_chars_from_list, word_chars_with_toggles)


class IllegalArgumentException(ValueError):
    """java.lang.IllegalArgumentException of the WholeWord constructors (S/WholeWordMatchMap.java:263-267)."""


def utf16(s):
    """str -> UTF-16 code units exactly as a Java String holds them (surrogate pairs for supplementary chars)."""
    if isinstance(s, str):
        return np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
    return np.ascontiguousarray(s, dtype=np.uint16)


def _pack(keywords):
    parts = [utf16(k) if k is not None else np.zeros(0, np.uint16) for k in keywords]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    if parts:
        off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    units = np.concatenate(parts) if parts else np.zeros(0, np.uint16)
    if units.size == 0:
        units = np.zeros(1, np.uint16)
    return np.ascontiguousarray(units, dtype=np.uint16), off


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def configured_devices():
    """The device list of match(String, ...): the environment variable ACGPU_DEVICES ("0,1,2,3"), this mirror's counterpart of
    the Java facade's system property -Dacgpu.devices=....  None: the current device alone."""
    v = os.environ.get("ACGPU_DEVICES", "").strip()
    return [int(x) for x in v.split(",") if x.strip() != ""] if v else None


class Automaton:
    """Owns one acgpu_automaton handle."""

    def __init__(self, mode, keywords, case_sensitive, word_chars=None, lower=None):
        L = N.lib()
        units, off = _pack(keywords)
        self.keywords = keywords
        lower_t = None
        if not case_sensitive:
            lower_t = np.ascontiguousarray(java_lower_table() if lower is None else lower, dtype=np.uint16)
        wc = None if word_chars is None else np.ascontiguousarray(word_chars, dtype=np.uint8)
        h = ctypes.c_void_p()
        bad = ctypes.c_int64(-1)
        rc = L.acgpu_build(mode, _vp(units), _vp(off), len(off) - 1, 1 if case_sensitive else 0, _vp(lower_t), _vp(wc),
                           ctypes.byref(h), ctypes.byref(bad))
        if rc == N.E_NONWORD:
            kw = keywords[bad.value]
            raise IllegalArgumentException("%s contains non-word characters." % (kw if isinstance(kw, str) else bad.value))
        N.check(rc, "acgpu_build")
        self._h = h
        self.mode = mode
        self.word_chars = wc

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                N.lib().acgpu_free(h)
            except Exception:
                pass
            self._h = None

    @property
    def handle(self):
        return self._h

    def info(self):
        i = N.Info()
        N.check(N.lib().acgpu_get_info(self._h, ctypes.byref(i)), "acgpu_get_info")
        return {f: getattr(i, f) for f, _ in N.Info._fields_}

    def match_host(self, hay_units, with_ids, cap=None, devices=None):
        """acgpu_match_u16 -- or, with a device list (argument, or ACGPU_DEVICES), acgpu_match_u16_multi: haystack in host
        memory -> (n, 2|3) int32 array in reference call order."""
        hay = np.ascontiguousarray(hay_units, dtype=np.uint16)
        n = int(hay.size)
        kind = N.REC_MAP if with_ids else N.REC_SET
        cols = kind // 4
        if cap is None:
            cap = max(4096, n // 64)
        buf_in = hay if n else np.zeros(1, np.uint16)
        if devices is None:
            devices = configured_devices()
        devs = (ctypes.c_int * len(devices))(*devices) if devices else None
        while True:
            out = np.empty((cap, cols), dtype=np.int32)
            n_out = ctypes.c_uint64(0)
            if devs is not None:
                rc = N.lib().acgpu_match_u16_multi(self._h, _vp(buf_in), n, devs, len(devices), kind, _vp(out), cap, ctypes.byref(n_out))
            else:
                rc = N.lib().acgpu_match_u16(self._h, _vp(buf_in), n, kind, _vp(out), cap, ctypes.byref(n_out))
            if rc == N.E_OVERFLOW:
                cap = int(n_out.value)
                continue
            N.check(rc, "acgpu_match_u16_multi" if devs is not None else "acgpu_match_u16")
            return out[:n_out.value]

    def match_batch(self, haystacks, with_ids, cap=None):
        """acgpu_match_batch_u16: many short haystacks (str or uint16 arrays) in one call -> (n, 3|4) int32 array of
        (haystack index, start, end[, keyword index]) records, haystack by haystack in reference call order."""
        units, off = _pack(haystacks)
        kind = N.REC_MAP if with_ids else N.REC_SET
        cols = kind // 4 + 1
        if cap is None:
            cap = max(4096, int(off[-1]) // 16)
        while True:
            out = np.empty((cap, cols), dtype=np.int32)
            n_out = ctypes.c_uint64(0)
            rc = N.lib().acgpu_match_batch_u16(self._h, _vp(units), _vp(off), len(off) - 1, kind, _vp(out), cap, ctypes.byref(n_out))
            if rc == N.E_OVERFLOW:
                cap = int(n_out.value)
                continue
            N.check(rc, "acgpu_match_batch_u16")
            return out[:n_out.value]

    def match_device(self, d_hay_ptr, n_units, with_ids, d_out_ptr, cap, own=None, text_begin=True, text_end=True,
                     chain_entry=None, stream=0, profile=False, d_result=None):
        """acgpu_match_device on raw device pointers.  Returns (n_out, rc, profile_dict|None, chain_exit)."""
        sh = N.Shard()
        sh.d_result = d_result
        sh.d_hay = d_hay_ptr
        sh.n_units = n_units
        sh.own_begin, sh.own_end = (0, n_units) if own is None else own
        sh.text_begin = 1 if text_begin else 0
        sh.text_end = 1 if text_end else 0
        sh.chain_entry = sh.own_begin if chain_entry is None else chain_entry
        sh.chain_exit = -1
        prof = N.Profile() if profile else None
        n_out = ctypes.c_uint64(0)
        rc = N.lib().acgpu_match_device(self._h, ctypes.byref(sh), N.REC_MAP if with_ids else N.REC_SET, d_out_ptr, cap,
                                        ctypes.byref(n_out), ctypes.c_void_p(stream),
                                        ctypes.byref(prof) if profile else None)
        pd = None
        if profile:
            pd = dict(scan_ms=prof.scan_ms, finalize_ms=prof.finalize_ms, total_ms=prof.total_ms,
                      scan_units=prof.scan_units, n_matches=prof.n_matches, scan_kernel=prof.scan_kernel.decode())
        return int(n_out.value), rc, pd, int(sh.chain_exit)


    def match_device_begin(self, d_hay_ptr, n_units, with_ids, d_out_ptr, cap, own=None, text_begin=True, text_end=True,
                           stream=0, profile=False, d_result=None, chain_entry=None):
        """acgpu_match_device_begin: enqueue without waiting (AhoCorasick, WholeWord with fold-consistent tables, the LongestMatch
        walk pipeline; the other families run inside the call).  Returns (ticket, rc); the ticket keeps the acgpu_shard alive
        (the library writes chain_exit into it when the ticket is collected)."""
        sh = N.Shard()
        sh.d_result = d_result
        sh.d_hay = d_hay_ptr
        sh.n_units = n_units
        sh.own_begin, sh.own_end = (0, n_units) if own is None else own
        sh.text_begin = 1 if text_begin else 0
        sh.text_end = 1 if text_end else 0
        sh.chain_entry = sh.own_begin if chain_entry is None else chain_entry
        sh.chain_exit = -1
        tk = ctypes.c_void_p()
        rc = N.lib().acgpu_match_device_begin(self._h, ctypes.byref(sh), N.REC_MAP if with_ids else N.REC_SET, d_out_ptr, cap,
                                              ctypes.c_void_p(stream), 1 if profile else 0, ctypes.byref(tk))
        return Ticket(tk, sh), rc

    def match_device_end(self, ticket, profile=False):
        """acgpu_match_device_end: waits for that call only.  Returns (n_out, rc, profile_dict|None); ticket.chain_exit holds
        the shard's chain exit afterwards."""
        prof = N.Profile() if profile else None
        n_out = ctypes.c_uint64(0)
        rc = N.lib().acgpu_match_device_end(self._h, ticket.handle, ctypes.byref(n_out), ctypes.byref(prof) if profile else None)
        pd = None
        if profile:
            pd = dict(scan_ms=prof.scan_ms, finalize_ms=prof.finalize_ms, total_ms=prof.total_ms,
                      scan_units=prof.scan_units, n_matches=prof.n_matches, scan_kernel=prof.scan_kernel.decode())
        return int(n_out.value), rc, pd


    def match_device_abandon(self, ticket):
        """acgpu_match_device_abandon: give the ticket up (waits for its kernels, never redoes the call)."""
        return N.lib().acgpu_match_device_abandon(self._h, ticket.handle)


class Comm:
    """acgpu_comm: the devices of a single-process multi-GPU job, one stream per device, and the transport of the gather
    (RCCL ncclCommInitAll / peer copies).  match_device_allgather: every device scans its shard into its slot of its gather
    buffer, one all-gather leaves every device with every shard's records."""

    def __init__(self, devices, transport=N.TRANSPORT_AUTO):
        self.devices = list(devices)
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        N.check(N.lib().acgpu_comm_open(arr, len(self.devices), transport, ctypes.byref(h)), "acgpu_comm_open")
        self._h = h

    @property
    def transport(self):
        return N.lib().acgpu_comm_transport(self._h)

    def stream(self, i):
        return N.lib().acgpu_comm_stream(self._h, i) or 0

    def match_device_allgather(self, automaton, shards, with_ids, gather_ptrs, gcap, profile=False):
        """shards: list of dicts(d_hay, n_units, own=(b, e), text_begin, text_end[, chain_entry]) -- shard i on devices[i];
        gather_ptrs: device pointers, buffer i on devices[i].  Returns (rc, counts list, chain exits list, profiles | None)."""
        k = len(self.devices)
        arr = (N.Shard * k)()
        for i, sd in enumerate(shards):
            sh = arr[i]
            sh.d_hay = sd["d_hay"]
            sh.n_units = sd["n_units"]
            sh.own_begin, sh.own_end = sd.get("own", (0, sd["n_units"]))
            sh.text_begin = 1 if sd.get("text_begin", i == 0) else 0
            sh.text_end = 1 if sd.get("text_end", i == k - 1) else 0
            sh.chain_entry = sd.get("chain_entry", sh.own_begin)
            sh.chain_exit = -1
            sh.d_result = None
        ptrs = (ctypes.c_void_p * k)(*gather_ptrs)
        counts = (ctypes.c_uint64 * k)()
        profs = (N.Profile * k)() if profile else None
        rc = N.lib().acgpu_match_device_allgather(automaton.handle, self._h, arr, N.REC_MAP if with_ids else N.REC_SET, ptrs, gcap,
                                                  counts, profs)
        pd = None
        if profile:
            pd = [dict(scan_ms=p.scan_ms, finalize_ms=p.finalize_ms, total_ms=p.total_ms, scan_units=p.scan_units,
                       n_matches=p.n_matches, scan_kernel=p.scan_kernel.decode()) for p in profs]
        return rc, [int(c) for c in counts], [int(arr[i].chain_exit) for i in range(k)], pd

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().acgpu_comm_close(h)

    __del__ = close


class Ticket:
    """One acgpu_match_device_begin call in flight: the native ticket and the acgpu_shard it was begun with."""

    def __init__(self, handle, shard):
        self.handle = handle
        self.shard = shard

    @property
    def chain_exit(self):
        return int(self.shard.chain_exit)


class Stream:
    """acgpu_stream: the haystack arrives in chunks (match(Readable, ...)); feed() returns the records that have become
    decidable as an (n, 2|3) int64 array with GLOBAL positions (units since the first feed)."""

    def __init__(self, automaton, with_ids=True, pipelined=False):
        """pipelined: acgpu_stream_set_pipelined -- a feed returns the records of the PREVIOUS feed's chunk (the final feed both),
        host copy, transfer and scan of neighbouring chunks overlap."""
        self._auto = automaton  # keeps the handle alive
        self._kind = N.REC_MAP if with_ids else N.REC_SET
        h = ctypes.c_void_p()
        N.check(N.lib().acgpu_stream_open(automaton.handle, ctypes.byref(h)), "acgpu_stream_open")
        self._h = h
        if pipelined:
            N.check(N.lib().acgpu_stream_set_pipelined(h, 1), "acgpu_stream_set_pipelined")

    def reserve(self, n_units):
        """acgpu_stream_reserve: a uint16 numpy view of the staging memory the next chunk may be written into; feed that view."""
        p = ctypes.c_void_p()
        N.check(N.lib().acgpu_stream_reserve(self._h, int(n_units), ctypes.byref(p)), "acgpu_stream_reserve")
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint16)), shape=(int(n_units),))

    def feed(self, units, final=False, cap=None):
        u = np.ascontiguousarray(units, dtype=np.uint16)  # (a view of the reserved staging memory stays where it is)
        n = int(u.size)
        cols = self._kind // 4
        if cap is None:
            cap = max(4096, n // 64)
        src = u if n else np.zeros(1, np.uint16)
        while True:
            out = getattr(self, "_out", None)  # (one record buffer per stream: a fresh one per feed costs more than the feed)
            if out is None or out.shape != (cap, cols):
                out = self._out = np.empty((cap, cols), dtype=np.int32)
            n_out, base = ctypes.c_uint64(0), ctypes.c_int64(0)
            rc = N.lib().acgpu_stream_feed(self._h, _vp(src), n, 1 if final else 0, self._kind, _vp(out), cap,
                                           ctypes.byref(n_out), ctypes.byref(base))
            if rc == N.E_OVERFLOW:  # nothing was consumed: same feed, larger buffer
                cap = int(n_out.value)
                continue
            N.check(rc, "acgpu_stream_feed")
            r = out[:n_out.value].astype(np.int64)
            r[:, :2] += base.value
            return r

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().acgpu_stream_close(h)

    __del__ = close


class ReadableMatchListener:
    """S/ReadableMatchListener.java:3-9"""

    def match(self, value):
        raise NotImplementedError


def _chunks(readable, chunk_chars):
    """A Java Readable hands out characters until it returns -1; here: an object with read(n) -> str ('' at the end),
    or any iterable of str / uint16 arrays."""
    if hasattr(readable, "read"):
        while True:
            c = readable.read(chunk_chars)
            if not c:
                return
            yield c
    else:
        yield from readable


# ---- listener plumbing -----------------------------------------------------------------------------------

class SetMatchListener:
    """S/SetMatchListener.java:3-8"""

    def match(self, haystack, start_position, end_position):
        raise NotImplementedError


class MapMatchListener:
    """S/MapMatchListener.java:3-8"""

    def match(self, haystack, start_position, end_position, value):
        raise NotImplementedError


def _listener_fn(listener):
    return listener.match if hasattr(listener, "match") else listener


class StringSet:
    """S/StringSet.java:3-5"""
    _MODE = None

    def _init(self, keywords, case_sensitive, word_chars=None):
        self._keywords = list(keywords)
        self._auto = Automaton(self._MODE, self._keywords, bool(case_sensitive), word_chars=word_chars)

    def match(self, haystack, listener):
        if haystack is None:
            raise TypeError("haystack is None")  # the reference throws NullPointerException at haystack.length()
        recs = self._auto.match_host(utf16(haystack), with_ids=False)
        fn = _listener_fn(listener)
        for s, e in recs.tolist():
            if not fn(haystack, s, e):
                break

    def find_all(self, haystack):
        """Convenience (not in the reference): the (n,2) int32 array of (start, end) records."""
        return self._auto.match_host(utf16(haystack), with_ids=False)

    def match_batch(self, haystacks, listener):
        """Not in the reference: match(haystack, listener) for every haystack of a list in ONE device call (short inputs: a
        call has tens of microseconds of fixed cost).  A listener call that returns False ends THAT haystack's matches."""
        fn = _listener_fn(listener)
        skip = -1
        for h, s, e in self._auto.match_batch(haystacks, with_ids=False).tolist():
            if h != skip and not fn(haystacks[h], s, e):
                skip = h

    @property
    def automaton(self):
        return self._auto


class StringMap:
    """S/StringMap.java:5-9 (String overload only)"""
    _MODE = None

    def _init(self, keywords, values, case_sensitive, word_chars=None):
        # the reference consumes keywords and values pairwise and stops at the shorter one (S/AhoCorasickMap.java:32)
        pairs = list(zip(keywords, values))
        self._keywords = [k for k, _ in pairs]
        self._values = [v for _, v in pairs]
        self._auto = Automaton(self._MODE, self._keywords, bool(case_sensitive), word_chars=word_chars)

    def match(self, haystack, listener):
        """match(String, MapMatchListener<T>) -- or, when `haystack` has read() / is an iterable of chunks,
        match(Readable, ReadableMatchListener<T>) (S/StringMap.java:6-8)."""
        if haystack is None:
            raise TypeError("haystack is None")
        if not isinstance(haystack, (str, np.ndarray)):
            return self.match_readable(haystack, listener)
        recs = self._auto.match_host(utf16(haystack), with_ids=True)
        fn = _listener_fn(listener)
        vals = self._values
        for s, e, k in recs.tolist():
            if not fn(haystack, s, e, vals[k]):
                break

    def match_readable(self, readable, listener, chunk_chars=1 << 22):
        """S/AhoCorasickMap.java:208-275, S/LongestMatchMap.java:203-286, S/WholeWordMatchMap.java:55-153: the listener
        receives only the value; returning False stops the scan (and the reading)."""
        fn = _listener_fn(listener)
        vals = self._values
        st = Stream(self._auto, with_ids=True)
        try:
            for chunk in _chunks(readable, chunk_chars):
                for k in st.feed(utf16(chunk))[:, 2].tolist():
                    if not fn(vals[k]):
                        return
            for k in st.feed(np.zeros(0, np.uint16), final=True)[:, 2].tolist():
                if not fn(vals[k]):
                    return
        finally:
            st.close()

    def find_all(self, haystack):
        """Convenience (not in the reference): the (n,3) int32 array of (start, end, keyword_index) records."""
        return self._auto.match_host(utf16(haystack), with_ids=True)

    def match_batch(self, haystacks, listener):
        """Not in the reference: match(haystack, listener) for every haystack of a list in ONE device call (see
        StringSet.match_batch)."""
        fn = _listener_fn(listener)
        vals = self._values
        skip = -1
        for h, s, e, k in self._auto.match_batch(haystacks, with_ids=True).tolist():
            if h != skip and not fn(haystacks[h], s, e, vals[k]):
                skip = h

    @property
    def automaton(self):
        return self._auto


def _word_chars(word_characters, toggle_flags):
    if word_characters is None:
        return default_word_chars()
    if toggle_flags is None:
        return word_chars_from_list(word_characters)
    return word_chars_with_toggles(word_characters, toggle_flags)


class AhoCorasickSet(StringSet):
    """S/AhoCorasickSet.java:11-20: every occurrence of every keyword."""
    _MODE = N.MODE_ALL

    def __init__(self, keywords, case_sensitive, threshold_strategy=None):
        self._init(keywords, case_sensitive)


class AhoCorasickMap(StringMap):
    """S/AhoCorasickMap.java:14-24"""
    _MODE = N.MODE_ALL

    def __init__(self, keywords, values, case_sensitive, threshold_strategy=None):
        self._init(keywords, values, case_sensitive)


class LongestMatchSet(StringSet):
    """S/LongestMatchSet.java:11-19: leftmost-longest, non-overlapping."""
    _MODE = N.MODE_LONGEST

    def __init__(self, keywords, case_sensitive, threshold_strategy=None):
        self._init(keywords, case_sensitive)


class LongestMatchMap(StringMap):
    """S/LongestMatchMap.java"""
    _MODE = N.MODE_LONGEST

    def __init__(self, keywords, values, case_sensitive, threshold_strategy=None):
        self._init(keywords, values, case_sensitive)


class ShortestMatchSet(StringSet):
    """S/ShortestMatchSet.java:8-20: reports a match as soon as any keyword ends, then restarts after it (the
    reference's "leftmost shortest" matcher; non-overlapping)."""
    _MODE = N.MODE_SHORTEST

    def __init__(self, keywords, case_sensitive, threshold_strategy=None):
        self._init(keywords, case_sensitive)


class ShortestMatchMap(StringMap):
    """S/ShortestMatchMap.java:16-24; of equal keywords the FIRST one's value is kept (:47-49)."""
    _MODE = N.MODE_SHORTEST

    def __init__(self, keywords, values, case_sensitive, threshold_strategy=None):
        self._init(keywords, values, case_sensitive)


class WholeWordMatchSet(StringSet):
    """S/WholeWordMatchSet.java: keywords that span a whole maximal run of word characters."""
    _MODE = N.MODE_WHOLEWORD

    def __init__(self, keywords, case_sensitive, word_characters=None, toggle_flags=None, threshold_strategy=None):
        self._init(keywords, case_sensitive, word_chars=_word_chars(word_characters, toggle_flags))

    def get_word_chars(self):
        return self._auto.word_chars


class WholeWordMatchMap(StringMap):
    """S/WholeWordMatchMap.java:21-53"""
    _MODE = N.MODE_WHOLEWORD

    def __init__(self, keywords, values, case_sensitive, word_characters=None, toggle_flags=None,
                 threshold_strategy=None):
        self._init(keywords, values, case_sensitive, word_chars=_word_chars(word_characters, toggle_flags))

    def get_word_chars(self):
        return self._auto.word_chars


class WholeWordLongestMatchSet(StringSet):
    """S/WholeWordLongestMatchSet.java:7-45: whole-word matches of keywords that may contain non-word characters
    ("as if"); of the keywords starting at a word the longest whole-word one wins, and the scan continues after the
    text it looked at."""
    _MODE = N.MODE_WWLONGEST

    def __init__(self, keywords, case_sensitive, word_characters=None, toggle_flags=None, threshold_strategy=None):
        self._init(keywords, case_sensitive, word_chars=_word_chars(word_characters, toggle_flags))

    def get_word_chars(self):
        return self._auto.word_chars


class WholeWordLongestMatchMap(StringMap):
    """S/WholeWordLongestMatchMap.java:22-54"""
    _MODE = N.MODE_WWLONGEST

    def __init__(self, keywords, values, case_sensitive, word_characters=None, toggle_flags=None,
                 threshold_strategy=None):
        self._init(keywords, values, case_sensitive, word_chars=_word_chars(word_characters, toggle_flags))

    def get_word_chars(self):
        return self._auto.word_chars
