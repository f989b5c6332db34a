package com.roklenarcic.util.strings.gpu;

import java.io.IOException;
import java.util.ArrayList;
import java.util.Iterator;
import java.util.List;

import com.roklenarcic.util.strings.MapMatchListener;
import com.roklenarcic.util.strings.ReadableMatchListener;
import com.roklenarcic.util.strings.StringMap;
import com.roklenarcic.util.strings.threshold.Thresholder;

/** Drop-in for com.roklenarcic.util.strings.AhoCorasickMap&lt;T&gt; (both overloads on the GPU). */
public class GpuAhoCorasickMap<T> implements StringMap<T>, AutoCloseable {
    final NativeAutomaton automaton;
    final List<T> values = new ArrayList<T>();

    public GpuAhoCorasickMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
        this(NativeAutomaton.MODE_ALL, keywords, values, caseSensitive, null);
    }

    /** Same signature as AhoCorasickMap(Iterable, Iterable, boolean, Thresholder); the Thresholder is ignored (results-neutral). */
    public GpuAhoCorasickMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive,
            final Thresholder thresholdStrategy) {
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

    /**
     * Not in the reference: {@link #match(String, MapMatchListener)} for every haystack of an array in ONE device call (see
     * GpuAhoCorasickSet.matchBatch). A listener call that returns false ends THAT haystack's matches.
     */
    public void matchBatch(final String[] haystacks, final MapMatchListener<T> listener) {
        final int[] r = automaton.matchBatch(haystacks, true);
        int skip = -1;
        for (int i = 0; i < r.length; i += 4) {
            if (r[i] != skip && !listener.match(haystacks[r[i]], r[i + 1], r[i + 2], values.get(r[i + 3]))) {
                skip = r[i];
            }
        }
    }

    /** Chunked scan through acgpu_stream_*: the listener receives only the value; false stops scan and reading. */
    public void match(final Readable haystack, final ReadableMatchListener<T> listener) throws IOException {
        final java.nio.CharBuffer buf = java.nio.CharBuffer.allocate(1 << 22);
        final long stream = automaton.openStream();
        try {
            boolean more = true;
            while (more) {
                buf.clear();
                while (buf.hasRemaining() && (more = haystack.read(buf) != -1)) {
                    // fill the chunk: a Readable may hand out a few characters at a time
                }
                buf.flip();
                final int[] ids = NativeAutomaton.feed(stream, buf.array(), buf.limit(), !more);
                for (int i = 0; i < ids.length; i++) {
                    if (!listener.match(values.get(ids[i]))) {
                        return;
                    }
                }
            }
        } finally {
            NativeAutomaton.closeStream(stream);
        }
    }

    public void close() {
        automaton.close();
    }
}
