package com.roklenarcic.util.strings.gpu;

import java.util.ArrayList;
import java.util.List;

import com.roklenarcic.util.strings.SetMatchListener;
import com.roklenarcic.util.strings.StringSet;
import com.roklenarcic.util.strings.threshold.Thresholder;

/** Drop-in for com.roklenarcic.util.strings.AhoCorasickSet (same constructor arguments, same listener contract). */
public class GpuAhoCorasickSet implements StringSet, AutoCloseable {
    final NativeAutomaton automaton;

    public GpuAhoCorasickSet(final Iterable<String> keywords, boolean caseSensitive) {
        this(NativeAutomaton.MODE_ALL, keywords, caseSensitive, null);
    }

    /**
     * Same signature as AhoCorasickSet(Iterable, boolean, Thresholder). The Thresholder only chooses between the
     * reference's two node representations (results-neutral); the GPU tables have no such choice, so it is ignored.
     */
    public GpuAhoCorasickSet(final Iterable<String> keywords, boolean caseSensitive, final Thresholder thresholdStrategy) {
        this(NativeAutomaton.MODE_ALL, keywords, caseSensitive, null);
    }

    GpuAhoCorasickSet(int mode, final Iterable<String> keywords, boolean caseSensitive, boolean[] wordChars) {
        List<String> l = new ArrayList<String>();
        for (String k : keywords) l.add(k); // null keywords are passed through and skipped natively
        automaton = new NativeAutomaton(mode, l.toArray(new String[l.size()]), caseSensitive, wordChars);
    }

    public void match(final String haystack, final SetMatchListener listener) {
        final int[] r = automaton.match(haystack, false); // NullPointerException for a null haystack, like the reference
        for (int i = 0; i < r.length; i += 2) {
            if (!listener.match(haystack, r[i], r[i + 1])) {
                return; // early stop exactly where the reference would stop; listener exceptions propagate
            }
        }
    }

    /**
     * Not in the reference: {@link #match(String, SetMatchListener)} for every haystack of an array in ONE device call (short
     * inputs: a call has tens of microseconds of fixed cost). A listener call that returns false ends THAT haystack's matches.
     */
    public void matchBatch(final String[] haystacks, final SetMatchListener listener) {
        final int[] r = automaton.matchBatch(haystacks, false);
        int skip = -1;
        for (int i = 0; i < r.length; i += 3) {
            if (r[i] != skip && !listener.match(haystacks[r[i]], r[i + 1], r[i + 2])) {
                skip = r[i];
            }
        }
    }

    public void close() {
        automaton.close();
    }
}
