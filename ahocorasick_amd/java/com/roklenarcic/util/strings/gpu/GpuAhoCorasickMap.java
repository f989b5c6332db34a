package com.roklenarcic.util.strings.gpu;

import java.io.IOException;
import java.util.ArrayList;
import java.util.Iterator;
import java.util.List;

import com.roklenarcic.util.strings.MapMatchListener;
import com.roklenarcic.util.strings.ReadableMatchListener;
import com.roklenarcic.util.strings.StringMap;

/** Drop-in for com.roklenarcic.util.strings.AhoCorasickMap&lt;T&gt; (String overload on the GPU). */
public class GpuAhoCorasickMap<T> implements StringMap<T>, AutoCloseable {
    final NativeAutomaton automaton;
    final List<T> values = new ArrayList<T>();

    public GpuAhoCorasickMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
        this(NativeAutomaton.MODE_ALL, keywords, values, caseSensitive, null);
    }

    GpuAhoCorasickMap(int mode, final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive,
            boolean[] wordChars) {
        List<String> l = new ArrayList<String>();
        Iterator<String> k = keywords.iterator();
        Iterator<? extends T> v = values.iterator();
        while (k.hasNext() && v.hasNext()) { // pairwise, stops at the shorter iterable like the reference
            l.add(k.next());
            this.values.add(v.next());
        }
        automaton = new NativeAutomaton(mode, l.toArray(new String[l.size()]), caseSensitive, wordChars);
    }

    public void match(final String haystack, final MapMatchListener<T> listener) {
        final int[] r = automaton.match(haystack, true);
        for (int i = 0; i < r.length; i += 3) {
            if (!listener.match(haystack, r[i], r[i + 1], values.get(r[i + 2]))) {
                return;
            }
        }
    }

    public void match(final Readable haystack, final ReadableMatchListener<T> listener) throws IOException {
        // streaming overload: out of scope of the GPU path (SURVEY.md 8f); delegate to the reference class
        throw new UnsupportedOperationException("use com.roklenarcic.util.strings.AhoCorasickMap for Readable input");
    }

    public void close() {
        automaton.close();
    }
}
