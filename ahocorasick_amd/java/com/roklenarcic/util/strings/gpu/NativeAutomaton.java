package com.roklenarcic.util.strings.gpu;

/**
 * Thin JNI binding of include/acgpu.h (libacgpu.so + libacgpu_jni.so). One instance owns one acgpu_automaton handle.
 * NOT compiled in the build image (no JDK there); see INTEGRATION.md for the build line.
 */
final class NativeAutomaton implements AutoCloseable {
    static final int MODE_ALL = 0, MODE_LONGEST = 1, MODE_WHOLEWORD = 2, MODE_SHORTEST = 3, MODE_WWLONGEST = 4;

    static {
        System.loadLibrary("acgpu_jni");
    }

    /**
     * The devices a match(String, ...) call is spread over: system property {@code acgpu.devices}, a comma-separated list of HIP
     * device ordinals ("0,1,2,3,4,5,6,7" on an 8-GPU node); unset = the current device alone. Read once per JVM.
     */
    private static final int[] DEVICES = parseDevices(System.getProperty("acgpu.devices"));

    static int[] parseDevices(String prop) {
        if (prop == null || prop.trim().isEmpty()) return null;
        String[] parts = prop.split(",");
        int[] d = new int[parts.length];
        for (int i = 0; i < parts.length; i++) d[i] = Integer.parseInt(parts[i].trim());
        return d;
    }

    /**
     * match(Readable, ...) through the pipelined feeds (system property {@code acgpu.stream.pipelined}): a feed returns the records of
     * the PREVIOUS chunk, so the listener hears of a match one chunk later and one more chunk has been read when it says stop.
     */
    private static final boolean PIPELINED = Boolean.getBoolean("acgpu.stream.pipelined");

    private long handle;

    NativeAutomaton(int mode, String[] keywords, boolean caseSensitive, boolean[] wordChars) {
        // the tables come from THIS JVM's Character methods, so parity holds for its Unicode version
        char[] lower = null;
        if (!caseSensitive) {
            lower = new char[65536];
            for (int i = 0; i < 65536; i++) lower[i] = Character.toLowerCase((char) i);
        }
        handle = build(mode, keywords, caseSensitive, lower, wordChars);
    }

    /** (start,end) pairs or (start,end,keywordIndex) triples, flattened, in the reference's listener-call order. */
    int[] match(String haystack, boolean withIds) {
        return match(handle, haystack, withIds, DEVICES);
    }

    /**
     * acgpu_match_batch_u16: many short haystacks in ONE device call (a call has tens of microseconds of fixed cost).
     * Returns (haystackIndex,start,end[,keywordIndex]) tuples, flattened, haystack by haystack, inside a haystack in the
     * reference's listener-call order; positions are relative to their haystack.
     */
    int[] matchBatch(String[] haystacks, boolean withIds) {
        return matchBatch(handle, haystacks, withIds);
    }

    /** acgpu_stream_*: the haystack arrives in chunks; returns a stream handle for {@link #feed}. */
    long openStream() {
        return streamOpen(handle, PIPELINED);
    }

    /** keyword indices of the records that became decidable with this chunk, in the reference's listener-call order */
    static int[] feed(long stream, char[] chunk, int length, boolean last) {
        return streamFeed(stream, chunk, length, last, PIPELINED);
    }

    static void closeStream(long stream) {
        streamClose(stream);
    }

    @Override
    public void close() {
        if (handle != 0) {
            free(handle);
            handle = 0;
        }
    }

    /** throws IllegalArgumentException("<keyword> contains non-word characters.") on ACGPU_E_NONWORD */
    private static native long build(int mode, String[] keywords, boolean caseSensitive, char[] lower, boolean[] wordChars);

    /** devices == null: acgpu_match_u16 on the current device; else acgpu_match_u16_multi over the list (one call, one process) */
    private static native int[] match(long handle, String haystack, boolean withIds, int[] devices);

    private static native int[] matchBatch(long handle, String[] haystacks, boolean withIds);

    private static native void free(long handle);

    private static native long streamOpen(long handle, boolean pipelined);

    private static native int[] streamFeed(long stream, char[] chunk, int length, boolean last, boolean pipelined);

    private static native void streamClose(long stream);
}
