package com.roklenarcic.util.strings.gpu;

import com.roklenarcic.util.strings.threshold.Thresholder;

/**
 * The remaining drop-ins: same shape as GpuAhoCorasickSet/Map with another native mode, and every constructor overload
 * of the class they replace (S/LongestMatchSet.java:15-19, S/ShortestMatchSet.java:14-18, S/WholeWordMatchSet.java:16-43,
 * S/WholeWordMatchMap.java:21-53, S/WholeWordLongestMatchSet.java:15-43, S/WholeWordLongestMatchMap.java:20-50).
 */
public final class GpuMatchers {
    private GpuMatchers() {
    }

    /** WordCharacters.generateWordCharsFlags(): this JVM's Character.isLetterOrDigit plus '-' and '_'. */
    static boolean[] defaultWordChars() {
        boolean[] f = new boolean[65536];
        for (int i = 0; i < 65536; i++) {
            f[i] = Character.isLetterOrDigit((char) i);
        }
        f['-'] = true;
        f['_'] = true;
        return f;
    }

    /** WordCharacters.generateWordCharsFlags(char[]): the listed characters and nothing else. */
    static boolean[] wordCharsOnly(char[] wordCharacters) {
        boolean[] f = new boolean[65536];
        for (int i = 0; i < wordCharacters.length; i++) {
            f[wordCharacters[i]] = true;
        }
        return f;
    }

    /** WordCharacters.generateWordCharsFlags(char[], boolean[]): the default table, then the listed characters set to their flag. */
    static boolean[] wordCharsToggled(char[] wordCharacters, boolean[] toggleFlags) {
        boolean[] f = defaultWordChars();
        for (int i = 0; i < wordCharacters.length; i++) {
            f[wordCharacters[i]] = toggleFlags[i];
        }
        return f;
    }

    /** Drop-in for LongestMatchSet. */
    public static class GpuLongestMatchSet extends GpuAhoCorasickSet {
        public GpuLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            super(NativeAutomaton.MODE_LONGEST, keywords, caseSensitive, null);
        }

        /** the Thresholder is accepted and ignored (results-neutral) */
        public GpuLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_LONGEST, keywords, caseSensitive, null);
        }
    }

    /** Drop-in for LongestMatchMap. */
    public static class GpuLongestMatchMap<T> extends GpuAhoCorasickMap<T> {
        public GpuLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            super(NativeAutomaton.MODE_LONGEST, keywords, values, caseSensitive, null);
        }

        /** the Thresholder is accepted and ignored (results-neutral) */
        public GpuLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive,
                final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_LONGEST, keywords, values, caseSensitive, null);
        }
    }

    /** Drop-in for ShortestMatchSet. */
    public static class GpuShortestMatchSet extends GpuAhoCorasickSet {
        public GpuShortestMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            super(NativeAutomaton.MODE_SHORTEST, keywords, caseSensitive, null);
        }

        /** the Thresholder is accepted and ignored (results-neutral) */
        public GpuShortestMatchSet(final Iterable<String> keywords, boolean caseSensitive, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_SHORTEST, keywords, caseSensitive, null);
        }
    }

    /** Drop-in for ShortestMatchMap (of equal keywords the first one's value is kept, as in the reference). */
    public static class GpuShortestMatchMap<T> extends GpuAhoCorasickMap<T> {
        public GpuShortestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            super(NativeAutomaton.MODE_SHORTEST, keywords, values, caseSensitive, null);
        }

        /** the Thresholder is accepted and ignored (results-neutral) */
        public GpuShortestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive,
                final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_SHORTEST, keywords, values, caseSensitive, null);
        }
    }

    /** Drop-in for WholeWordMatchSet. */
    public static class GpuWholeWordMatchSet extends GpuAhoCorasickSet {
        /** digits, letters, '-' and '_' are word characters */
        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, defaultWordChars());
        }

        /** exactly the given characters are word characters */
        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, wordCharsOnly(wordCharacters));
        }

        /** the default table with the given characters switched on / off by their flags */
        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        /** (the Thresholder of the overloads below is accepted and ignored: results-neutral) */
        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, wordCharsOnly(wordCharacters));
        }

        public GpuWholeWordMatchSet(final Iterable<String> keywords, boolean caseSensitive, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, caseSensitive, defaultWordChars());
        }
    }

    /** Drop-in for WholeWordMatchMap. */
    public static class GpuWholeWordMatchMap<T> extends GpuAhoCorasickMap<T> {
        /** digits, letters, '-' and '_' are word characters */
        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, defaultWordChars());
        }

        /** exactly the given characters are word characters */
        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, wordCharsOnly(wordCharacters));
        }

        /** the default table with the given characters switched on / off by their flags */
        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        /** (the Thresholder of the overloads below is accepted and ignored: results-neutral) */
        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters,
                final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, wordCharsOnly(wordCharacters));
        }

        public GpuWholeWordMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WHOLEWORD, keywords, values, caseSensitive, defaultWordChars());
        }
    }

    /** Drop-in for WholeWordLongestMatchSet (keywords may contain non-word characters). */
    public static class GpuWholeWordLongestMatchSet extends GpuAhoCorasickSet {
        /** digits, letters, '-' and '_' are word characters */
        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, defaultWordChars());
        }

        /** exactly the given characters are word characters */
        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, wordCharsOnly(wordCharacters));
        }

        /** the default table with the given characters switched on / off by their flags */
        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        /** (the Thresholder of the overloads below is accepted and ignored: results-neutral) */
        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, char[] wordCharacters,
                final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, wordCharsOnly(wordCharacters));
        }

        public GpuWholeWordLongestMatchSet(final Iterable<String> keywords, boolean caseSensitive, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, caseSensitive, defaultWordChars());
        }
    }

    /** Drop-in for WholeWordLongestMatchMap. */
    public static class GpuWholeWordLongestMatchMap<T> extends GpuAhoCorasickMap<T> {
        /** digits, letters, '-' and '_' are word characters */
        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, defaultWordChars());
        }

        /** exactly the given characters are word characters */
        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, wordCharsOnly(wordCharacters));
        }

        /** the default table with the given characters switched on / off by their flags */
        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        /** (the Thresholder of the overloads below is accepted and ignored: results-neutral) */
        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters,
                boolean[] toggleFlags, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, wordCharsToggled(wordCharacters, toggleFlags));
        }

        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, char[] wordCharacters,
                final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, wordCharsOnly(wordCharacters));
        }

        public GpuWholeWordLongestMatchMap(final Iterable<String> keywords, final Iterable<? extends T> values, boolean caseSensitive, final Thresholder thresholdStrategy) {
            super(NativeAutomaton.MODE_WWLONGEST, keywords, values, caseSensitive, defaultWordChars());
        }
    }
}
