"""CPU-side sanitizer runs (GPU AddressSanitizer is not available on this pool): the oracle's golden tests under
its ASan/UBSan build, and the host builder (ahocorasick_amd/csrc/acgpu_build.cpp) compiled with g++ -fsanitize=address,undefined
and driven over dictionaries of every family -- its tables must be the product library's, and the run must be clean."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()
    if not p or not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("no libasan in this toolchain")
    return p


def _env(asan, **extra):
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    env.update(extra)
    return env


def test_oracle_golden_suite_under_asan_ubsan():
    asan = _libasan()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=_env(asan, ORACLE_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]


def test_host_builder_under_asan_ubsan_builds_the_product_tables(tmp_path):
    asan = _libasan()
    lib = tmp_path / "libacgpu_build_san.so"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "include"), "-o", str(lib),
                           os.path.join(ROOT, "ahocorasick_amd", "csrc", "acgpu_build.cpp"), os.path.join(ROOT, "tests", "san_shim.cpp")])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_worker.py"), str(lib), str(tmp_path)], env=_env(asan),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "san_worker: done" in out and "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    # the same dictionaries through the product library (acgpu_build + the debug hooks of the C ABI)
    sys.path.insert(0, ROOT)
    from ahocorasick_amd import _native as N
    from ahocorasick_amd.strings import Automaton, IllegalArgumentException
    from tests.san_worker import cases
    from tests.test_native_cpu import _tables
    n_checked = 0
    for name, mode, kws, cs, wc in cases():
        got = np.load(tmp_path / (name + ".npz"))
        if int(got["rc"]) != 0:
            assert int(got["rc"]) == N.E_NONWORD and int(got["bad"]) == 1
            with pytest.raises(IllegalArgumentException):
                Automaton(mode, kws, cs, word_chars=wc)
            continue
        a = Automaton(mode, kws, cs, word_chars=wc)
        info, cls, dfa, out_len, out_link, out_id, depth, first = _tables(a)
        i = got["info"]
        assert (int(i[0]), int(i[1]), int(i[2]), int(i[3]), int(i[4]), int(i[5]), int(i[6])) == (
            info["n_states"], info["n_classes"], info["dense"], info["n_keywords"], info["min_keyword_len"], info["max_keyword_len"], first), name
        assert int(i[7]) == info["fold_consistent"] and int(i[8]) == info["fold_clean"] and int(i[9]) == info["filter_k"], name
        assert (got["cls"] == cls).all() and (got["out_len"] == out_len).all() and (got["out_link"] == out_link).all(), name
        assert (got["out_id"] == out_id).all() and (got["depth"] == depth).all(), name
        if dfa is not None:
            assert (got["dfa"].reshape(dfa.shape) == dfa).all(), name
        n_checked += 1
    assert n_checked >= 9
