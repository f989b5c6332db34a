package com.roklenarcic.util.strings.gpu;

/** The remaining drop-ins: same shape as GpuAhoCorasickSet/Map with another native mode. */
public final class GpuMatchers {
    private GpuMatchers() {
    }

    /** Drop-in for LongestMatchSet. */
    public static class GpuLongestMatchSet extends GpuAhoCorasickSet {
        public GpuLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            super(NativeAutomaton.MODE_LONGEST, keywords, caseSensitive, null);
        }
    }

    /** Drop-in for LongestMatchMap. */
    public static class GpuLongestMatchMap<T> extends GpuAhoCorasickMap<T> {
        public GpuLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            super(NativeAutomaton.MODE_LONGEST, keywords, values, caseSensitive, null);
        }
    }

    /** Drop-in for ShortestMatchSet. */
    public static class GpuShortestMatchSet extends GpuAhoCorasickSet {
        public GpuShortestMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            super(NativeAutomaton.MODE_SHORTEST, keywords, caseSensitive, null);
        }
    }

    /** Drop-in for ShortestMatchMap (of equal keywords the first one's value is kept, as in the reference). */
    public static class GpuShortestMatchMap<T> extends GpuAhoCorasickMap<T> {
        public GpuShortestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            super(NativeAutomaton.MODE_SHORTEST, keywords, values, caseSensitive, null);
        }
    }

    /** WordCharacters.generateWordCharsFlags() with this JVM's Character.isLetterOrDigit. */
    static boolean[] defaultWordChars() {
        boolean[] f = new boolean[65536];
        f['-'] = true;
        f['_'] = true;
        for (int i = 0; i < 65536; i++) {
            if (Character.isLetterOrDigit((char) i)) f[i] = true;
        }
        return f;
    }

    static boolean[] wordChars(char[] wordCharacters, boolean[] toggleFlags) {
        if (wordCharacters == null) return defaultWordChars();
        boolean[] f = toggleFlags == null ? new boolean[65536] : defaultWordChars();
        for (int i = 0; i < wordCharacters.length; i++) f[wordCharacters[i]] = toggleFlags == null || toggleFlags[i];
        return f;
    }

    /** Drop-in for WholeWordMatchSet (all word-character overloads). */
    public static class GpuWholeWordMatchSet extends GpuAhoCorasickSet {
        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            this(keywords, caseSensitive, null, null);
        }

        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters, boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, wordChars(wordCharacters, toggleFlags));
        }
    }

    /** Drop-in for WholeWordMatchMap. */
    public static class GpuWholeWordMatchMap<T> extends GpuAhoCorasickMap<T> {
        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            this(keywords, values, caseSensitive, null, null);
        }

        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive,
                char[] wordCharacters, boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, wordChars(wordCharacters, toggleFlags));
        }
    }

    /** Drop-in for WholeWordLongestMatchSet (keywords may contain non-word characters). */
    public static class GpuWholeWordLongestMatchSet extends GpuAhoCorasickSet {
        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            this(keywords, caseSensitive, null, null);
        }

        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, wordChars(wordCharacters, toggleFlags));
        }
    }

    /** Drop-in for WholeWordLongestMatchMap (String overload; the Readable overload stays on the reference class). */
    public static class GpuWholeWordLongestMatchMap<T> extends GpuAhoCorasickMap<T> {
        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            this(keywords, values, caseSensitive, null, null);
        }

        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive,
                char[] wordCharacters, boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, wordChars(wordCharacters, toggleFlags));
        }
    }
}
