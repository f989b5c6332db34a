"""ahocorasick_amd -- MI355X-native drop-in for the match() hot path of RokLenarcic/AhoCorasick.

The product is ahocorasick_amd/lib/libacgpu.so (C ABI: include/acgpu.h; HIP kernels: ahocorasick_amd/csrc/).
This package is the host-side mirror of the reference's API on top of it.  No CPU fallback exists.
"""
from .strings import (AhoCorasickMap, AhoCorasickSet, Automaton, IllegalArgumentException, LongestMatchMap,
                      LongestMatchSet, MapMatchListener, ReadableMatchListener, SetMatchListener, ShortestMatchMap, ShortestMatchSet, Stream, StringMap,
                      StringSet, WholeWordLongestMatchMap, WholeWordLongestMatchSet, WholeWordMatchMap, WholeWordMatchSet,
                      utf16)

__all__ = ["AhoCorasickSet", "AhoCorasickMap", "LongestMatchSet", "LongestMatchMap", "WholeWordMatchSet",
           "WholeWordMatchMap", "ShortestMatchSet", "ShortestMatchMap", "WholeWordLongestMatchSet", "WholeWordLongestMatchMap", "StringSet", "StringMap", "SetMatchListener", "MapMatchListener", "ReadableMatchListener", "Stream", "Automaton",
           "IllegalArgumentException", "utf16"]
