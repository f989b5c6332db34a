"""65536-entry tables standing in for the two JDK functions the reference calls per UTF-16 code unit.

The reference folds case with ``Character.toLowerCase(char)`` (call sites S/AhoCorasickSet.java:33,229;
S/WholeWordMatchMap.java:204,277) and builds its default word-character table from
``Character.isLetterOrDigit(char)`` plus '-' and '_' (S/WordCharacters.java:6-16).  Both depend on the JVM's
Unicode version, so the native library takes them as *inputs* (include/acgpu.h: acgpu_build).  A Java facade
would fill them from its own JVM; this Python host fills them from ``unicodedata`` (Unicode 13.0 on this image
== JDK 15-18), per SURVEY.md 8c.
"""
import functools
import unicodedata

import numpy as np


@functools.lru_cache(maxsize=None)
def java_lower_table():
    """lower[c] == (int) Character.toLowerCase((char) c) for every UTF-16 code unit."""
    t = np.arange(65536, dtype=np.uint16)
    for c in range(65536):
        if 0xD800 <= c <= 0xDFFF:
            continue  # surrogate halves are never folded (per-code-unit API)
        lo = chr(c).lower()
        if len(lo) == 1 and ord(lo) < 65536:
            t[c] = ord(lo)
    # U+0130: Python applies SpecialCasing ("i" + U+0307); Java's char API uses the simple mapping 'i'.
    t[0x0130] = 0x0069
    t.setflags(write=False)
    return t


_LETTER_OR_DIGIT = {"Lu", "Ll", "Lt", "Lm", "Lo", "Nd"}


@functools.lru_cache(maxsize=None)
def default_word_chars():
    """WordCharacters.generateWordCharsFlags(): S/WordCharacters.java:6-16."""
    t = np.zeros(65536, dtype=np.uint8)
    for c in range(65536):
        if unicodedata.category(chr(c)) in _LETTER_OR_DIGIT:
            t[c] = 1
    t[ord("-")] = 1
    t[ord("_")] = 1
    t.setflags(write=False)
    return t


def word_chars_from_list(word_characters):
    """WordCharacters.generateWordCharsFlags(char[]): S/WordCharacters.java:18-24."""
    t = np.zeros(65536, dtype=np.uint8)
    for c in word_characters:
        t[ord(c) if isinstance(c, str) else int(c)] = 1
    return t


def word_chars_with_toggles(word_characters, toggle_flags):
    """WordCharacters.generateWordCharsFlags(char[], boolean[]): S/WordCharacters.java:26-39."""
    t = default_word_chars().copy()
    for c, f in zip(word_characters, toggle_flags):
        t[ord(c) if isinstance(c, str) else int(c)] = 1 if f else 0
    return t
