"""Generates tests/golden/reference_fixtures.json.

Inputs are the deterministic scenarios of the reference's own tests (haystack + keywords, typed in here as DATA:
T/SetTest.java:61-130 mirrored in T/MapTest.java:68-131, README worked examples R/README.md:88-124).  Expected
outputs are computed with oracle/brute.py, i.e. with the brute-force formulas the reference tests themselves
assert against (T/AhoCorasickTest.java:28-38, T/LongestMatchTest.java:30-42, T/WholeWordMatchTest.java:73-90,
T/ShortestMatchTest.java:30-42, T/WholeWordLongestMatchTest.java:46-65)
plus the documented emission order.  The reference (Java) cannot be executed in this image; see DESIGN.md.

Run:  python tests/golden/make_fixtures.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import brute  # noqa: E402
from ahocorasick_amd.unicode_tables import default_word_chars  # noqa: E402

SCENARIOS = [
    # (name, source, haystack, keywords)
    ("failureTransitions", "T/SetTest.java:69", "abbccddeef", ["bc", "cc", "bcc", "ccddee", "ccddeee", "d"]),
    ("literal", "T/SetTest.java:93", "The quick red fox, jumps over the lazy brown dog.",
     ["The", "quick", "red", "fox", "jumps", "over", "the", "lazy", "brown", "dog"]),
    ("longestMatch", "T/SetTest.java:98", "XXXYYZZ", ["XXX", "YY", "XXXYYZZZ"]),
    ("overlap1", "T/SetTest.java:112", "aaaa", ["a", "aa", "aaa", "aaaa"]),
    ("overlap2", "T/SetTest.java:113", " aaaaaaa aaababababaabaa ", ["a", "aa", "aaa", "aaaa"]),
    ("longKeywords", "T/SetTest.java:103-104", "a" * 100, ["a" * k for k in range(1, 101)]),
    ("shortest2", "T/SetTest.java:120", "abcyyyy", ["abcd", "bcxxxx", "cyyyy"]),
    ("wwl1", "T/SetTest.java:125", "as if", ["as", "if", "as if"]),
    ("wwl2", "T/SetTest.java:126", "ax if", ["as", "if", "as if"]),
    ("wwl3", "T/SetTest.java:127", "as in", ["as", "if", "as if"]),
    ("wwl4", "T/SetTest.java:128", "123 4x 1234 5x 1234 56 123 45 1x 345 12 34x 12 345x 123xb 1234 56s",
     ["123", "123 45", "1234 56", "12 345"]),
    ("wwl5", "T/SetTest.java:129", "abc 12", ["abc", "abc 123"]),
    ("readmeOverlap", "R/README.md:88-90", "aaaa", ["a", "aa", "aaa", "aaaa"]),
    ("readmeLongest", "R/README.md:94-96", "a1b2c3d4", ["b", "b2", "2c3d4"]),
    ("readmeWholeWord", "R/README.md:106-109", "late evening", ["la", "late", "eve", "evening"]),
    # README "ShortestMatchSet/Map": with 2, b2, 2c3d4 only b2 matches; with b, 2, b2 both b and 2 match
    ("readmeShortest1", "R/README.md:100", "a1b2c3d4", ["2", "b2", "2c3d4"]),
    ("readmeShortest2", "R/README.md:100", "a1b2c3d4", ["b", "2", "b2"]),
    # README "WholeWordLongestMatchSet/Map": `as if` -> as if; `ax if` -> if; `as of` -> as
    ("readmeWwlAsIf", "R/README.md:122-124", "as if", ["as if", "as", "if"]),
    ("readmeWwlAxIf", "R/README.md:122-124", "ax if", ["as if", "as", "if"]),
    ("readmeWwlAsOf", "R/README.md:122-124", "as of", ["as if", "as", "if"]),
    ("emptyHaystack", "T/SetTest.java:61-65", "", ["ab", "abc", "zz"]),
    ("nwcRejection", "T/WholeWordMatchTest.java:49-52", "A B", ["A B"]),
]


def main():
    wc = default_word_chars()
    out = []
    for name, src, hay, kws in SCENARIOS:
        e = {"name": name, "source": src, "haystack": hay, "keywords": kws}
        e["AC"] = [list(m) for m in brute.ac_all(hay, kws)]
        e["L"] = [list(m) for m in brute.longest(hay, kws)]
        # ShortestMatchTest extends SetTest and sorts the keywords by length before building (T/ShortestMatchTest.java:51-59)
        skws = sorted(kws, key=len)
        e["S_keywords"] = skws
        e["S_count"] = brute.shortest_test_count(hay, kws)
        e["S"] = [list(m) for m in brute.shortest(hay, skws)]
        # WholeWordLongestMatchTest extends SetTest, sorts the keywords longest first (T/WholeWordLongestMatchTest.java:76-84)
        lkws = sorted(kws, key=len, reverse=True)
        e["WWL_keywords"] = lkws
        e["WWL_count"] = brute.wwlongest_test_count(hay, kws, wc)
        e["WWL"] = [list(m) for m in brute.wwlongest(hay, lkws, wc)]
        try:
            e["WW"] = [list(m) for m in brute.wholeword(hay, kws, wc)]
        except brute.NonWordCharacters:
            e["WW"] = "IllegalArgumentException"
        out.append(e)
    # fullNode (T/SetTest.java:72-79): all 65536 single-unit keywords; haystack = units 0x0000 0xFFFF 0xFFFE
    out.append({"name": "fullNode", "source": "T/SetTest.java:72-79", "haystack_units": [0, 0xFFFF, 0xFFFE],
                "keywords_gen": "all_single_units",
                "AC": [[0, 1, 0], [1, 2, 0xFFFF], [2, 3, 0xFFFE]], "L": [[0, 1, 0], [1, 2, 0xFFFF], [2, 3, 0xFFFE]],
                "WW": "IllegalArgumentException", "S_count": 3, "S": [[0, 1, 0], [1, 2, 0xFFFF], [2, 3, 0xFFFE]]})  # WholeWordLongest: not run (65536 one-unit
    # keywords of which 16255 are non-word characters: nothing the reference's test asserts)
    with open(os.path.join(HERE, "reference_fixtures.json"), "w") as f:
        json.dump(out, f, indent=0, ensure_ascii=True)
    print("wrote", len(out), "fixtures")


if __name__ == "__main__":
    main()
