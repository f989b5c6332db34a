"""Closed-form, brute-force statements of what each reference matcher reports -- TEST INFRASTRUCTURE ONLY.

These follow the *test oracles* of the reference (not its matcher code), so they are an independent check on
oracle/ac_oracle.c:
  AC-all   : T/AhoCorasickTest.java:28-38   (every occurrence of every keyword)
  Longest  : T/LongestMatchTest.java:30-42  (greedy leftmost-longest, keywords sorted by length desc :50-58)
  WholeWord: T/WholeWordMatchTest.java:60-90 (maximal word-char runs equal to a keyword)
  Shortest : T/ShortestMatchTest.java:30-42 (count only: first keyword, shortest first, matching at each position) and,
             independently, the closed form "accept a match of the all-matches list iff it starts at or after the end of
             the last accepted one" (earliest end first, longest among equal ends)
plus the emission order documented in SURVEY.md Appendix A (end ascending, longest first).
Pure-Python loops: small cases only.
"""
import numpy as np


def _units(s):
    if isinstance(s, str):
        return tuple(np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).tolist())
    return tuple(int(x) for x in s)


def _fold(u, lower):
    return u if lower is None else tuple(int(lower[x]) for x in u)


def _dictionary(keywords, lower):
    """folded keyword -> index of the LAST input keyword producing it (S/AhoCorasickMap.java:49-50)."""
    d = {}
    for i, k in enumerate(keywords):
        if k is None:
            continue
        u = _fold(_units(k), lower)
        if len(u) > 0:
            d[u] = i
    return d


def ac_all(haystack, keywords, case_sensitive=True, lower=None):
    lo = None if case_sensitive else lower
    h = _fold(_units(haystack), lo)
    d = _dictionary(keywords, lo)
    lens = sorted({len(k) for k in d}, reverse=True)
    out = []
    for end in range(1, len(h) + 1):
        for L in lens:  # longest first == start ascending
            if L <= end:
                idx = d.get(h[end - L:end])
                if idx is not None:
                    out.append((end - L, end, idx))
    return out


def longest(haystack, keywords, case_sensitive=True, lower=None):
    lo = None if case_sensitive else lower
    h = _fold(_units(haystack), lo)
    d = _dictionary(keywords, lo)
    lens = sorted({len(k) for k in d}, reverse=True)
    out = []
    pos = 0
    n = len(h)
    while pos < n:
        hit = 0
        for L in lens:
            if pos + L <= n:
                idx = d.get(h[pos:pos + L])
                if idx is not None:
                    out.append((pos, pos + L, idx))
                    hit = L
                    break
        pos += hit if hit else 1
    return out


def trim(keyword_units, word_chars):
    """WordCharacters.trim semantics (S/WordCharacters.java:41-62), restated independently."""
    u = list(keyword_units)
    flags = [bool(word_chars[c]) for c in u]
    if not any(flags):
        return tuple(u)  # untouched when it has no word chars at all
    first = flags.index(True)
    last = len(flags) - 1 - flags[::-1].index(True)
    return tuple(u[first:last + 1])


class NonWordCharacters(ValueError):
    pass


def wholeword(haystack, keywords, word_chars, case_sensitive=True, lower=None):
    """Valid for fold-consistent tables (word_chars[c] == word_chars[lower[c]] for all c), which includes the
    default table and every case-sensitive use."""
    lo = None if case_sensitive else lower
    d = {}
    for i, k in enumerate(keywords):
        if k is None:
            continue
        u = trim(_units(k), word_chars)
        if any(not word_chars[c] for c in u):
            raise NonWordCharacters(i)
        if len(u) > 0:
            d[_fold(u, lo)] = i
    raw = _units(haystack)
    h = _fold(raw, lo)
    n = len(raw)
    out = []
    i = 0
    while i < n:
        if not word_chars[raw[i]]:
            i += 1
            continue
        j = i
        while j < n and word_chars[raw[j]]:
            j += 1
        idx = d.get(h[i:j])
        if idx is not None:
            out.append((i, j, idx))
        i = j
    return out


def shortest_test_count(haystack, keywords):
    """T/ShortestMatchTest.java:30-42 with prepareKeywords (:51-59: keywords sorted by length, stable)."""
    h = _units(haystack)
    needles = [_units(k) for k in sorted([k for k in keywords], key=len)]
    count = 0
    i = 0
    while i < len(h):
        for nd in needles:
            if i + len(nd) <= len(h) and h[i:i + len(nd)] == nd:
                count += 1
                i += len(nd) - 1
                break
        i += 1
    return count


def shortest(haystack, keywords, case_sensitive=True, lower=None):
    """ShortestMatchSet/Map in closed form: matching restarts at the end of every reported match, so a keyword
    occurrence is reported iff it starts at or after the end of the previously reported one, taking occurrences by
    increasing end and, at equal end, the longest first.  Value: the FIRST input keyword producing the folded string
    (a later duplicate finds the node already matched and is skipped, S/ShortestMatchMap.java:47-49)."""
    lo = None if case_sensitive else lower
    h = _fold(_units(haystack), lo)
    d = {}
    for i, k in enumerate(keywords):
        if k is None:
            continue
        u = _fold(_units(k), lo)
        if len(u) > 0 and u not in d:
            d[u] = i
    lens = sorted({len(k) for k in d}, reverse=True)
    out = []
    s = 0
    for end in range(1, len(h) + 1):
        for L in lens:
            if L <= end and end - L >= s:
                idx = d.get(h[end - L:end])
                if idx is not None:
                    out.append((end - L, end, idx))
                    s = end
                    break
    return out


def wwlongest_test_count(haystack, keywords, word_chars):
    """T/WholeWordLongestMatchTest.java:46-65 with prepareKeywords (:76-84: sorted by length, longest first, stable)
    and the trimming of instantiateSet (:68-73).  Character.isLetterOrDigit == word_chars minus '-' and '_'."""
    def lod(c):
        return bool(word_chars[c]) and c not in (0x2D, 0x5F)
    h = _units(haystack)
    needles = [trim(_units(k), word_chars) for k in sorted(keywords, key=len, reverse=True)]
    count = 0
    i = 0
    n = len(h)
    while i < n:
        for nd in needles:
            L = len(nd)
            if L > 0 and i + L <= n and h[i:i + L] == nd and (i + L == n or not lod(h[i + L])) and (i == 0 or not lod(h[i - 1])):
                count += 1
                i += L - 1
                i += 1
                while i < n and not word_chars[h[i]]:
                    i += 1
                i -= 1
                break
        i += 1
    return count


def wwlongest(haystack, keywords, word_chars, case_sensitive=True, lower=None):
    """WholeWordLongestMatchSet/Map from its definition (fold-consistent tables): from every word start that the scan
    reaches, follow the text as long as it stays a prefix of some keyword; where it stops (position i), report the
    whole path if it is a keyword and the next unit is not a word character, else the longest keyword prefix of the
    path that is followed by a non-word unit of the path; the scan continues at the first word start after i."""
    lo = None if case_sensitive else lower
    d = {}
    prefixes = set()
    for idx, k in enumerate(keywords):
        if k is None:
            continue
        u = _fold(trim(_units(k), word_chars), lo)
        if len(u) > 0:
            d[u] = idx
            for j in range(1, len(u) + 1):
                prefixes.add(u[:j])
    raw = _units(haystack)
    h = _fold(raw, lo)
    n = len(raw)
    out = []
    p = 0
    while p < n:
        i = p
        while i < n and h[p:i + 1] in prefixes:
            i += 1
        # fail match: longest keyword prefix h[p:q] (q < i... q <= i-1) whose next path unit h[q] is a non-word unit
        hit = None
        if h[p:i] in d and (i == n or not word_chars[h[i]]):
            hit = (p, i, d[h[p:i]])
        else:
            for q in range(i - 1, p, -1):
                if h[p:q] in d and not word_chars[h[q]]:
                    hit = (p, q, d[h[p:q]])
                    break
        if hit:
            out.append(hit)
        if i < n and word_chars[h[i]]:
            while i < n and word_chars[raw[i]]:
                i += 1
        i += 1 if i < n and not word_chars[raw[i]] else 0
        while i < n and not word_chars[raw[i]]:
            i += 1
        p = max(i, p + 1) if i <= p else i
    return out
