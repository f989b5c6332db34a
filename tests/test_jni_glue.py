"""The JNI glue (ahocorasick_amd/java/jni/acgpu_jni.c) through a compiler and through its paces without a JDK (the
image has no JDK, jni.h or JVM): tests/jni_min/jni.h is a stand-in written from the JNI specification (the ~20 JNIEnv entries
the glue calls), tests/jni_min/mock_env.c a mock JNIEnv with a plain-C harness around the glue's native methods.

  * CPU (`-m "not gpu"`): glue + mock + a stub of the C ABI, `gcc -Wall -Wextra -Werror -fsanitize=address,undefined`, every
    scenario of tests/jni_min/cpu_driver.c with leak detection on -- exceptions and their messages, the capacity protocol, slices
    of a 32 Mi-char String, device lists, batches, feeds, the int[] limit, no JNI call with an exception pending.
  * GPU (`-m gpu`): the same glue + mock linked against the product library; what the native methods return must be what the
    ctypes binding returns for the same inputs (the binding is what every other GPU test checks against the oracle).

What this does NOT show: the glue inside a JVM (local-reference tables, GC, the real function table's layout)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JDIR = os.path.join(ROOT, "tests", "jni_min")
WARN = ["-std=c11", "-Wall", "-Wextra", "-Werror"]


def test_jni_glue_compiles_warning_free_and_survives_the_sanitizers(tmp_path):
    exe = str(tmp_path / "jni_cpu")
    cmd = ["gcc"] + WARN + ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1", "-I", JDIR, "-I", os.path.join(ROOT, "include"),
                            "-o", exe, os.path.join(JDIR, "cpu_driver.c"), os.path.join(JDIR, "mock_env.c"), os.path.join(JDIR, "stub_acgpu.c")]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if b.returncode != 0 and b"asan" in b.stdout.lower() and b"cannot find" in b.stdout.lower():
        pytest.skip("no libasan in this toolchain")
    assert b.returncode == 0, b.stdout.decode(errors="replace")[-4000:]
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "all scenarios ok" in out and "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]


def test_jni_glue_optimised_build_is_warning_free_too(tmp_path):
    # (-O2 sees more: uninitialised paths, format truncation; no sanitizer runtime)
    obj = str(tmp_path / "glue.o")
    b = subprocess.run(["gcc"] + WARN + ["-O2", "-fPIC", "-c", "-I", JDIR, "-I", os.path.join(ROOT, "include"), "-o", obj,
                                          os.path.join(JDIR, "mock_env.c")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b.returncode == 0, b.stdout.decode(errors="replace")[-4000:]


# ---- GPU: the glue over the product library == the ctypes binding ---------------------------------------------------------------

@pytest.fixture(scope="module")
def jh():
    from ahocorasick_amd import _native as N
    N.lib()  # (torch's HIP runtime first, then libacgpu.so: the harness binds to the copy that is already loaded)
    build = os.path.join(JDIR, "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libjh.so")
    libdir = os.path.dirname(N.LIB_PATH)
    subprocess.check_call(["gcc"] + WARN + ["-O1", "-g", "-fPIC", "-shared", "-I", JDIR, "-I", os.path.join(ROOT, "include"), "-o", so,
                                            os.path.join(JDIR, "mock_env.c"), "-L", libdir, "-l:" + os.path.basename(N.LIB_PATH),
                                            "-Wl,-rpath," + libdir])
    L = ctypes.CDLL(so)
    vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    L.jh_exception_class.restype = ctypes.c_char_p
    L.jh_exception_message.restype = ll
    L.jh_exception_message.argtypes = [vp, ll]
    L.jh_violations.restype = ll
    L.jh_outstanding_elements.restype = ll
    L.jh_release.argtypes = [vp]
    L.jh_build.restype = ll
    L.jh_build.argtypes = [ci, vp, vp, vp, ci, ci, vp, vp, ci]
    L.jh_free.argtypes = [ll]
    L.jh_match.restype = ll
    L.jh_match.argtypes = [ll, vp, ll, ci, vp, ci, ctypes.POINTER(vp)]
    L.jh_match_batch.restype = ll
    L.jh_match_batch.argtypes = [ll, vp, vp, vp, ci, ci, ctypes.POINTER(vp)]
    L.jh_stream_open.restype = ll
    L.jh_stream_open.argtypes = [ll, ci]
    L.jh_stream_feed.restype = ll
    L.jh_stream_feed.argtypes = [ll, vp, ci, ci, ci, ci, ctypes.POINTER(vp)]
    L.jh_stream_close.argtypes = [ll]
    yield L
    assert L.jh_violations() == 0 and L.jh_outstanding_elements() == 0


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _ints(L, k, p, cols):
    assert k >= 0, (L.jh_exception_class(), _message(L))
    a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int32)), shape=(int(k),)).copy() if k else np.zeros(0, np.int32)
    L.jh_release(p)
    return a.reshape(-1, cols)


def _message(L):
    buf = np.zeros(4096, np.uint16)
    n = L.jh_exception_message(_vp(buf), 4096)
    return None if n < 0 else buf[:n].tobytes().decode("utf-16-le", "surrogatepass")


def _jbuild(L, mode, keywords, cs, lower=None, word=None):
    from ahocorasick_amd.strings import _pack
    units, off = _pack([k if k is not None else "" for k in keywords])
    nulls = np.array([1 if k is None else 0 for k in keywords], dtype=np.uint8)
    lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.uint16)
    wc = None if word is None else np.ascontiguousarray(word, dtype=np.uint8)
    return L.jh_build(mode, _vp(units), _vp(off), _vp(nulls), len(keywords), 1 if cs else 0, _vp(lo), _vp(wc), 65536)


def _jmatch(L, h, hay, with_ids, devices=None):
    hay = np.ascontiguousarray(hay, dtype=np.uint16)
    out = ctypes.c_void_p()
    d = None if devices is None else np.array(devices, dtype=np.int32)
    k = L.jh_match(h, _vp(hay if hay.size else np.zeros(1, np.uint16)), hay.size, 1 if with_ids else 0, _vp(d), 0 if d is None else d.size,
                   ctypes.byref(out))
    return _ints(L, k, out, 3 if with_ids else 2)


@pytest.mark.gpu
def test_jni_native_methods_return_what_the_ctypes_binding_returns(jh):
    from ahocorasick_amd import _native as N, synth
    from ahocorasick_amd.strings import Automaton, utf16
    from tests.helpers import LOWER, WORD
    L = jh
    rng = np.random.default_rng(77)
    # AhoCorasickMap, case sensitive (a null keyword among them: skipped, as S/AhoCorasickSet.java:27 does)
    kws = synth.random_keywords(5, 800, 2, 9, table=synth.ALPHA_LOWER[:8])
    hay = synth.haystack(6, 300000, table=synth.ALPHA_LOWER[:8])
    a = Automaton(N.MODE_ALL, kws, True)
    h = _jbuild(L, N.MODE_ALL, list(kws[:400]) + [None] + list(kws[400:]), True)
    assert h != 0
    want = a.match_host(hay, True)
    got = _jmatch(L, h, hay, True)
    got[:, 2] -= (got[:, 2] > 400)  # (the null keyword took index 400 on the Java side)
    assert got.shape == want.shape and (got == want).all() and len(want) > 1000
    assert (_jmatch(L, h, hay, False) == want[:, :2]).all()
    for n in (0, 1, 7, 4096, 4097):  # the one-launch form and the general path behind the same native method
        assert (_jmatch(L, h, hay[:n], True)[:, :2] == a.match_host(hay[:n], True)[:, :2]).all()
    # the capacity protocol: more records than n / 64 + 4096
    dense = np.tile(kws[0], 40000)
    assert len(a.match_host(dense, False)) > dense.size // 64 + 4096
    assert (_jmatch(L, h, dense, False) == a.match_host(dense, False)).all()
    # -Dacgpu.devices: the one GPU named twice (shares of 2^22 units and more)
    big = synth.haystack(8, (1 << 23) + 4097, table=synth.ALPHA_LOWER[:8])
    wb = a.match_host(big, False)
    assert (_jmatch(L, h, big, False, devices=[0, 0]) == wb).all()
    L.jh_free(h)
    # beyond one GetStringRegion slice (32 Mi chars)
    n = (32 << 20) + 4321
    long_hay = synth.haystack(9, n, table=synth.ALPHA_LOWER[:8])
    h = _jbuild(L, N.MODE_ALL, kws, True)
    wl = a.match_host(long_hay, False)
    gl = _jmatch(L, h, long_hay, False)
    assert gl.shape == wl.shape and (gl == wl).all() and ((wl[:, 0] < (32 << 20)) & (wl[:, 1] > (32 << 20))).any()
    L.jh_free(h)

    # WholeWordMatchMap, case-insensitive, the JVM's tables handed over as arrays
    words = ["Alpha", "beta", "GAMMA", "été", "Αβ", "x1"]
    text = utf16(" alpha BETA, gamma-ÉTÉ x1 αΒ alphabet beta")
    text = np.tile(text, 2000)
    aw = Automaton(N.MODE_WHOLEWORD, words, False, word_chars=WORD, lower=LOWER)
    h = _jbuild(L, N.MODE_WHOLEWORD, words, False, lower=LOWER, word=WORD)
    assert h != 0
    want = aw.match_host(text, True)
    assert len(want) > 8000 and (_jmatch(L, h, text, True) == want).all()
    # matchBatch: (haystack, start, end, value) records
    hays = [text[int(o):int(o) + int(ln)] for o, ln in zip(rng.integers(0, text.size - 400, 300), rng.integers(0, 400, 300))]
    wantb = aw.match_batch(hays, True)
    from ahocorasick_amd.strings import _pack
    units, off = _pack(hays)
    out = ctypes.c_void_p()
    k = L.jh_match_batch(h, _vp(units), _vp(off), None, len(hays), 1, ctypes.byref(out))
    assert (_ints(L, k, out, 4) == wantb).all() and len(wantb) > 100
    # match(Readable, ...): the values, feed by feed, synchronous and pipelined
    from ahocorasick_amd.strings import Stream
    for pipelined in (0, 1):
        s = L.jh_stream_open(h, pipelined)
        assert s != 0
        ref = Stream(aw, True, pipelined=bool(pipelined))
        got_vals, want_vals = [], []
        cuts = [0, 5, 6, 4000, 4001, 30011, text.size]
        for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            last = i == len(cuts) - 2
            chunk = np.ascontiguousarray(np.concatenate([text[lo:hi], np.zeros(3, np.uint16)]))  # a buffer longer than what counts
            out = ctypes.c_void_p()
            k = L.jh_stream_feed(s, _vp(chunk), chunk.size, hi - lo, 1 if last else 0, pipelined, ctypes.byref(out))
            got_vals.append(_ints(L, k, out, 1)[:, 0])
            want_vals.append(ref.feed(text[lo:hi], final=last)[:, 2])
        assert (np.concatenate(got_vals) == np.concatenate(want_vals)).all() and (np.concatenate(got_vals) == want[:, 2]).all()
        L.jh_stream_close(s)
        ref.close()
    L.jh_free(h)
    # IllegalArgumentException(keyword + " contains non-word characters."): the String itself, whatever its characters
    h = _jbuild(L, N.MODE_WHOLEWORD, ["fine", "café crème\U0001F600"], True, word=WORD)
    assert h == 0 and L.jh_exception_class() == b"java/lang/IllegalArgumentException"
    assert _message(L) == "café crème\U0001F600 contains non-word characters."
    # LongestMatchSet over config 4's shape (the bit form behind the native method)
    kw4 = synth.prefix_closed_keywords(1004, 2000, word_len=120) + [utf16("b")]
    hay4 = synth.haystack(2004, (1 << 22) + 5, table=synth.ALPHA_AB_75)
    a4 = Automaton(N.MODE_LONGEST, kw4, True)
    h = _jbuild(L, N.MODE_LONGEST, kw4, True)
    assert (_jmatch(L, h, hay4, False) == a4.match_host(hay4, False, cap=hay4.size)).all()
    L.jh_free(h)
