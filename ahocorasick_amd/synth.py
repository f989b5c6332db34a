"""Deterministic synthetic dictionaries and haystacks for the BASELINE.json configs.

PRNG = SplitMix64 used counter-style: the i-th draw (i = 0,1,...) of stream `seed` is
    x = seed + (i+1)*0x9E3779B97F4A7C15;  z = (x ^ x>>30)*0xBF58476D1CE4E5B9;
    z = (z ^ z>>27)*0x94D049BB133111EB;   z ^= z>>31
and a bounded draw in [0,k) is ((z>>32)*k)>>32.  Being counter-based, unit i of a haystack depends only on
(seed, i), so the device generator (csrc/acgpu_synth.hip: acgpu_synth_fill) and this numpy generator agree
bit-for-bit and a multi-GPU shard can be generated in place.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_at(seed, idx):
    """z values for draw indices idx (array of uint64) of stream `seed`."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * GOLDEN
        z = (x ^ (x >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def bounded(z, k):
    return ((z >> np.uint64(32)) * np.uint64(k)) >> np.uint64(32)


_MASK = (1 << 64) - 1


def _mix_int(x):
    z = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


class Stream:
    """Sequential view of one SplitMix64 stream (for dictionary generation); pure-int, same values as
    splitmix64_at(seed, i)."""

    def __init__(self, seed):
        self.seed = int(seed)
        self.i = 0

    def draw(self, k):
        self.i += 1
        z = _mix_int((self.seed + self.i * 0x9E3779B97F4A7C15) & _MASK)
        return ((z >> 32) * k) >> 32

    def draws(self, n, k):
        if n > 64:
            z = splitmix64_at(self.seed, np.arange(self.i, self.i + n, dtype=np.uint64))
            self.i += n
            return bounded(z, k).astype(np.int64)
        return np.array([self.draw(k) for _ in range(n)], dtype=np.int64)


# Haystack "alphabet programs": unit = table[bounded(z, len(table))]
ALPHA_LOWER = np.arange(ord("a"), ord("z") + 1, dtype=np.uint16)  # iid uniform a-z   (C1, C2, C3)
ALPHA_AB_75 = np.array([ord("a")] * 3 + [ord("b")], dtype=np.uint16)  # P(a)=0.75      (C4)


def haystack(seed, n_units, table=ALPHA_LOWER, start=0, chunk=1 << 24):
    """Units [start, start+n_units) of the haystack stream `seed` (numpy reference of acgpu_synth_fill)."""
    table = np.asarray(table, dtype=np.uint16)
    out = np.empty(n_units, dtype=np.uint16)
    for lo in range(0, n_units, chunk):
        hi = min(n_units, lo + chunk)
        z = splitmix64_at(seed, np.arange(start + lo, start + hi, dtype=np.uint64))
        out[lo:hi] = table[bounded(z, len(table))]
    return out


def random_keywords(seed, n, min_len, max_len, table=ALPHA_LOWER):
    """n DISTINCT keywords, length uniform in [min_len, max_len], units uniform over `table`.
    Returns a list of uint16 arrays (generation order)."""
    s = Stream(seed)
    seen = set()
    out = []
    table = np.asarray(table, dtype=np.uint16)
    while len(out) < n:
        ln = min_len + s.draw(max_len - min_len + 1)
        u = table[s.draws(ln, len(table))]
        key = u.tobytes()
        if key not in seen:
            seen.add(key)
            out.append(u)
    return out


def prefix_closed_keywords(seed, n, word_len=1000):
    """C4 dictionary: every prefix of base words over {a,b} of length word_len; the first base word is
    a^word_len (the literal a, aa, aaa, ... family), further base words are uniform random; stops at n distinct."""
    s = Stream(seed)
    seen = set()
    out = []
    first = True
    ab = np.array([ord("a"), ord("b")], dtype=np.uint16)
    while len(out) < n:
        if first:
            w = np.full(word_len, ord("a"), dtype=np.uint16)
            first = False
        else:
            w = ab[s.draws(word_len, 2)]
        for ln in range(1, word_len + 1):
            key = w[:ln].tobytes()
            if key not in seen:
                seen.add(key)
                out.append(w[:ln].copy())
                if len(out) >= n:
                    break
    return out


# ---- C5: mixed-script whole-word workload -------------------------------------------------------------

_SCRIPTS = [
    # (name, code-unit ranges)
    ("latin", [(0x41, 0x5A), (0x61, 0x7A), (0xC0, 0xD6), (0xD8, 0xF6), (0xF8, 0xFF)]),
    ("greek", [(0x0391, 0x03A1), (0x03A3, 0x03A9), (0x03B1, 0x03C9)]),
    ("cyrillic", [(0x0410, 0x044F)]),
    ("cjk", [(0x4E00, 0x9FA5)]),  # the Unicode 1.1 URO: letters in every JDK's tables
    ("hangul", [(0xAC00, 0xD7A3)]),
    ("arabic", [(0x0621, 0x063A), (0x0641, 0x064A)]),
]
_SEPARATORS = np.array([0x20, ord(","), ord("."), 0x0A, 0x3002, 0x2014], dtype=np.uint16)


def _script_tables():
    return [np.concatenate([np.arange(a, b + 1, dtype=np.uint16) for a, b in rs]) for _, rs in _SCRIPTS]


def mixed_script_words(seed, n, min_len=2, max_len=12):
    """n distinct single-script words (C5 dictionary)."""
    s = Stream(seed)
    tabs = _script_tables()
    seen = set()
    out = []
    while len(out) < n:
        t = tabs[s.draw(len(tabs))]
        ln = min_len + s.draw(max_len - min_len + 1)
        u = t[s.draws(ln, len(t))]
        key = u.tobytes()
        if key not in seen:
            seen.add(key)
            out.append(u)
    return out


def mixed_script_haystack(seed, n_units, words, swapcase_tbl=None):
    """C5 haystack: tokens (50% dictionary word with random per-unit case flips, 50% random word) separated by
    1-3 separator units; truncated to exactly n_units.  Sequential stream (host generation)."""
    s = Stream(seed)
    tabs = _script_tables()
    out = np.empty(n_units + 64, dtype=np.uint16)
    pos = 0
    nw = len(words)
    while pos < n_units:
        if s.draw(2) == 0 and nw:
            w = words[s.draw(nw)].copy()
            if swapcase_tbl is not None:
                flips = s.draws(len(w), 2).astype(bool)
                w[flips] = swapcase_tbl[w[flips]]
        else:
            t = tabs[s.draw(len(tabs))]
            ln = 2 + s.draw(11)
            w = t[s.draws(ln, len(t))]
        nsep = 1 + s.draw(3)
        sep = _SEPARATORS[s.draws(nsep, len(_SEPARATORS))]
        tok = np.concatenate([w, sep])
        take = min(len(tok), n_units + 64 - pos)
        out[pos:pos + take] = tok[:take]
        pos += take
    return out[:n_units].copy()


def token_stream_haystack(seed, n_units, words, swapcase_tbl=None, chunk_tokens=1 << 20):
    """C5 haystack, token-indexed (numpy twin of acgpu_synth_tokens): token t owns draws 32 t .. 32 t + 31 of stream `seed` --
    draw 0: dictionary word (0) or random word; dictionary word: draw 1 its index, draws 2.. a case flip per unit (the first
    24); random word: draw 1 the script, draw 2 the length 2..12, draws 3.. its units; draw 28: 1..3 separators, draws 29..
    which -- so the text is a function of (seed, dictionary) alone and every token can be generated independently.  The
    haystack is the tokens one after the other, cut at n_units."""
    tabs = _script_tables()
    nw = len(words)
    wl = np.array([len(w) for w in words], dtype=np.int64)
    woff = np.concatenate([[0], np.cumsum(wl)])
    wcat = np.concatenate([np.asarray(w, dtype=np.uint16) for w in words]) if nw else np.zeros(0, np.uint16)
    soff = np.concatenate([[0], np.cumsum([len(t) for t in tabs])])
    scat = np.concatenate(tabs)
    out = np.empty(n_units + (int(wl.max()) if nw else 0) + 16, dtype=np.uint16)  # (room for the token that crosses the end)
    pos, t0 = 0, 0
    while pos < n_units:
        t = np.arange(t0, t0 + chunk_tokens, dtype=np.uint64)
        d0 = t * np.uint64(32)

        def draw(j, k):
            return bounded(splitmix64_at(seed, d0 + np.uint64(j)), k).astype(np.int64)
        is_dict = (draw(0, 2) == 0) & (nw > 0)
        d1_dict = draw(1, max(nw, 1))
        d1_scr = draw(1, 6)
        length = np.where(is_dict, wl[d1_dict] if nw else 0, 2 + draw(2, 11))
        nsep = 1 + draw(28, 3)
        tok_len = length + nsep
        starts = pos + np.concatenate([[0], np.cumsum(tok_len)[:-1]])
        keep = starts < n_units
        for i in range(int(length.max())):  # unit i of every token's word
            m = keep & (i < length)
            md, mr = m & is_dict, m & ~is_dict
            if md.any():
                u = wcat[woff[d1_dict[md]] + i]
                if swapcase_tbl is not None and i < 24:
                    fl = draw(2 + i, 2)[md].astype(bool)
                    u = np.where(fl, swapcase_tbl[u], u)
                out[starts[md] + i] = u
            if mr.any():
                sc = d1_scr[mr]
                tl = (soff[sc + 1] - soff[sc])
                z = splitmix64_at(seed, d0[mr] + np.uint64(3 + i))
                out[starts[mr] + i] = scat[soff[sc] + (((z >> np.uint64(32)) * tl.astype(np.uint64)) >> np.uint64(32)).astype(np.int64)]
        for i in range(3):
            m = keep & (i < nsep)
            out[starts[m] + length[m] + i] = _SEPARATORS[draw(29 + i, 6)[m]]
        n_keep = int(keep.sum())
        pos = int(starts[n_keep - 1] + tok_len[n_keep - 1]) if n_keep else pos
        if n_keep < chunk_tokens:
            break
        t0 += chunk_tokens
    return out[:n_units].copy()


def token_stream_on_device(d_ptr, n_units, seed, words, swapcase_tbl=None, stream=0):
    """acgpu_synth_tokens: token_stream_haystack(seed, n_units, words, swapcase_tbl) written to the device buffer at d_ptr."""
    import ctypes
    from . import _native as N
    parts = [np.asarray(w, dtype=np.uint16) for w in words]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    if parts:
        off[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    units = np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(1, np.uint16), dtype=np.uint16)
    sw = None if swapcase_tbl is None else np.ascontiguousarray(swapcase_tbl, dtype=np.uint16)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None  # noqa: E731
    N.check(N.lib().acgpu_synth_tokens(ctypes.c_void_p(d_ptr), int(n_units), int(seed), vp(units), vp(off), len(parts), vp(sw),
                                       ctypes.c_void_p(stream)), "acgpu_synth_tokens")


def swapcase_table():
    """Per-unit case flip used by mixed_script_haystack (upper<->lower where a single-unit mapping exists)."""
    t = np.arange(65536, dtype=np.uint16)
    for c in range(65536):
        if 0xD800 <= c <= 0xDFFF:
            continue
        ch = chr(c)
        sw = ch.swapcase()
        if len(sw) == 1 and ord(sw) < 65536:
            t[c] = ord(sw)
    return t


CONFIGS = {
    # name: dictionary spec, haystack spec
    "C1": dict(matcher="AhoCorasickSet", dict_seed=1001, hay_seed=2001, n_kw=100, min_len=3, max_len=8,
               n_units=1 << 19),
    "C2": dict(matcher="AhoCorasickMap", dict_seed=1002, hay_seed=2002, n_kw=10000, min_len=4, max_len=12,
               n_units=1 << 29),
    "C3": dict(matcher="AhoCorasickSet", dict_seed=1002, hay_seed=2003, n_kw=10000, min_len=4, max_len=12,
               n_units=1 << 29),  # per GPU; shard g uses hay_seed + g
    "C4": dict(matcher="LongestMatchSet", dict_seed=1004, hay_seed=2004, n_kw=50000, word_len=1000,
               n_units=1 << 29),
    "C5": dict(matcher="WholeWordMatchMap", dict_seed=1005, hay_seed=2005, n_kw=100000, min_len=2, max_len=12,
               n_units=1 << 28),  # per GPU
}


def config_keywords(name):
    c = CONFIGS[name]
    if name in ("C1", "C2", "C3"):
        return random_keywords(c["dict_seed"], c["n_kw"], c["min_len"], c["max_len"])
    if name == "C4":
        return prefix_closed_keywords(c["dict_seed"], c["n_kw"], c["word_len"])
    if name == "C5":
        return mixed_script_words(c["dict_seed"], c["n_kw"], c["min_len"], c["max_len"])
    raise KeyError(name)


# ---- the reference's own published workload (R/README.md:126-152) --------------------------------------------------------------
# "Dictionary: OS X dictionary at /usr/share/dict/words, 235886 english words.  Input string: a paragraph of english text" -- the
# file does not exist in this image, so the dictionary is an English-SHAPED stand-in of the same size: words drawn letter by letter
# from the English letter frequencies, lengths 1..24 distributed like a word list's (mode 9), one word in twelve capitalised (the
# list's proper nouns), the 52 single letters and the words of the paragraph itself included (a real word list holds them too).
README_PARAGRAPH = (  # the paragraph of T/SetTest.java:50-54, typed in as data
    "Values specified as nondelimited strings are interpreted according their length. For a string 8 or 14 characters long, the year "
    "is assumed to be given by the first 4 characters. Otherwise, the year is assumed to be given by the first 2 characters. The string "
    "is interpreted from left to right to find year, month, day, hour, minute, and second values, for as many parts as are present in "
    "the string. This means you should not use strings that have fewer than 6 characters.")
README_WORDS = 235886
_LETTER_WEIGHTS = [("e", 127), ("t", 91), ("a", 82), ("o", 75), ("i", 70), ("n", 67), ("s", 63), ("h", 61), ("r", 60), ("d", 43), ("l", 40),
                   ("c", 28), ("u", 28), ("m", 24), ("w", 24), ("f", 22), ("g", 20), ("y", 20), ("p", 19), ("b", 15), ("v", 10), ("k", 8),
                   ("j", 2), ("x", 2), ("q", 1), ("z", 1)]
_LENGTH_WEIGHTS = [0, 0, 2, 10, 40, 90, 150, 210, 260, 280, 270, 230, 180, 130, 90, 60, 35, 20, 12, 7, 4, 2, 1, 1, 1]  # index = length


def readme_dictionary(seed=1006, n=README_WORDS):
    """n distinct words as uint16 arrays, deterministic (SplitMix64 stream `seed`)."""
    import re
    letters = np.concatenate([np.full(w, ord(c), np.uint16) for c, w in _LETTER_WEIGHTS])
    lens = np.concatenate([np.full(w, L, np.int64) for L, w in enumerate(_LENGTH_WEIGHTS)])
    seen = set()
    out = []

    def add(b):
        if b not in seen:
            seen.add(b)
            out.append(np.frombuffer(b, dtype=np.uint8).astype(np.uint16))

    for c in range(26):  # the single letters of the word list
        add(bytes([ord("a") + c]))
        add(bytes([ord("A") + c]))
    for w in re.findall(r"[A-Za-z]+", README_PARAGRAPH):
        add(w.encode())
        add(w.lower().encode())
    st = Stream(seed)
    while len(out) < n:
        m = 4096
        ls = lens[st.draws(m, len(lens))]
        caps = st.draws(m, 12) == 0
        us = letters[st.draws(int(ls.sum()), len(letters))]
        at = 0
        for L, cap in zip(ls.tolist(), caps.tolist()):
            w = us[at:at + L].astype(np.uint8)
            at += L
            if cap:
                w = w.copy()
                w[0] -= 32
            add(w.tobytes())
            if len(out) >= n:
                break
    return out[:n]


def readme_text(seed, n_units, words):
    """English-shaped text of n_units units: dictionary words (rank-skewed: low indices far more often) and, one token in five, a
    random letter string, separated by a space -- one in ten by ', ' or '. '.  Vectorised; deterministic."""
    wl = np.array([len(w) for w in words], dtype=np.int64)
    woff = np.concatenate([[0], np.cumsum(wl)])
    wcat = np.concatenate(words)
    letters = np.concatenate([np.full(w, ord(c), np.uint16) for c, w in _LETTER_WEIGHTS])
    n_tok = n_units // 4 + 16  # (tokens are at least 2 units long: more than enough)
    st = Stream(seed)
    u = st.draws(n_tok, 1 << 30).astype(np.float64) / float(1 << 30)
    idx = np.minimum((len(words) * u ** 3).astype(np.int64), len(words) - 1)  # cubic skew towards the head of the list
    rnd = st.draws(n_tok, 5) == 0
    rlen = 2 + st.draws(n_tok, 9)
    length = np.where(rnd, rlen, wl[idx])
    sepk = st.draws(n_tok, 20)
    nsep = np.where(sepk < 2, 2, 1)
    tok = length + nsep
    start = np.concatenate([[0], np.cumsum(tok)[:-1]])
    keep = int(np.searchsorted(start, n_units))
    length, nsep, tok, start, idx, rnd, sepk = (a[:keep] for a in (length, nsep, tok, start, idx, rnd, sepk))
    total = int(tok.sum())
    tid = np.repeat(np.arange(keep, dtype=np.int64), tok)
    pos = np.arange(total, dtype=np.int64) - np.repeat(start, tok)
    in_word = pos < length[tid]
    src = np.where(rnd[tid], 0, woff[idx[tid]]) + np.minimum(pos, np.maximum(length[tid] - 1, 0))
    out = np.where(in_word, wcat[np.minimum(src, len(wcat) - 1)], 32).astype(np.uint16)
    rmask = in_word & rnd[tid]
    out[rmask] = letters[st.draws(int(rmask.sum()), len(letters))]
    first_sep = (~in_word) & (pos == length[tid]) & (nsep[tid] == 2)
    out[first_sep] = np.where(sepk[tid[first_sep]] == 0, ord("."), ord(","))
    if total < n_units:
        out = np.concatenate([out, np.full(n_units - total, 32, np.uint16)])
    return np.ascontiguousarray(out[:n_units])
