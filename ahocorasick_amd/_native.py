"""ctypes binding of include/acgpu.h (libacgpu.so).  There is no Python/CPU fallback: if the HIP library is
missing or no device is usable, every match call fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACGPU_LIB") or os.path.join(_HERE, "lib", "libacgpu.so")  # ACGPU_LIB: A/B builds

OK, E_INVALID, E_NONWORD, E_NOMEM, E_OVERFLOW, E_HIP, E_NODEVICE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7
MODE_ALL, MODE_LONGEST, MODE_WHOLEWORD, MODE_SHORTEST, MODE_WWLONGEST = 0, 1, 2, 3, 4
REC_SET, REC_MAP = 8, 12
TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_PEER = 0, 1, 2


def gather_slot_bytes(gcap, record_kind):
    """acgpu_gather_slot_bytes (include/acgpu.h): [16-byte acgpu_device_result | gcap records], padded to 16 bytes."""
    return (16 + int(gcap) * int(record_kind) + 15) & ~15


class AcgpuError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = lib().acgpu_strerror(code).decode() if _lib is not None else str(code)
        if code == E_HIP:
            msg += " (hipError_t=%d)" % lib().acgpu_last_hip_error()
        super().__init__("%s: %s [%d]" % (where, msg, code))


class Info(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_uint32), ("mode", ctypes.c_uint32), ("case_sensitive", ctypes.c_uint32),
                ("n_states", ctypes.c_uint32), ("n_classes", ctypes.c_uint32), ("n_keywords", ctypes.c_uint32),
                ("min_keyword_len", ctypes.c_uint32), ("max_keyword_len", ctypes.c_uint32), ("dense", ctypes.c_uint32),
                ("entry_bytes", ctypes.c_uint32), ("table_bytes", ctypes.c_uint64), ("lds_states", ctypes.c_uint32),
                ("fold_consistent", ctypes.c_uint32), ("filter_k", ctypes.c_uint32), ("filter_bits", ctypes.c_uint32),
                ("tile_kernel", ctypes.c_uint32), ("filter_density", ctypes.c_float), ("fold_clean", ctypes.c_uint32)]


class Shard(ctypes.Structure):
    _fields_ = [("d_hay", ctypes.c_void_p), ("n_units", ctypes.c_uint64), ("own_begin", ctypes.c_uint64),
                ("own_end", ctypes.c_uint64), ("text_begin", ctypes.c_int32), ("text_end", ctypes.c_int32),
                ("chain_entry", ctypes.c_int64), ("chain_exit", ctypes.c_int64), ("d_result", ctypes.c_void_p)]


class DeviceResult(ctypes.Structure):  # acgpu_device_result: what Shard.d_result receives, in stream order
    _fields_ = [("n_records", ctypes.c_uint64), ("redone", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


ABI_VERSION = 5


class Profile(ctypes.Structure):
    _fields_ = [("scan_ms", ctypes.c_float), ("finalize_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("scan_units", ctypes.c_uint64), ("n_matches", ctypes.c_uint64), ("scan_kernel", ctypes.c_char * 64)]


# every symbol include/acgpu.h declares
SYMBOLS = ["acgpu_build", "acgpu_free", "acgpu_get_info", "acgpu_match_u16", "acgpu_match_batch_u16", "acgpu_match_device",
           "acgpu_match_device_begin", "acgpu_match_device_end", "acgpu_match_device_abandon", "acgpu_synth_fill", "acgpu_synth_tokens", "acgpu_stream_probe",
           "acgpu_set_tunable", "acgpu_strerror", "acgpu_last_hip_error", "acgpu_abi_version", "acgpu_debug_tables", "acgpu_debug_states",
           "acgpu_debug_wordhash", "acgpu_debug_wordhash_perfect", "acgpu_stream_open", "acgpu_stream_feed", "acgpu_stream_close",
           "acgpu_match_u16_multi", "acgpu_comm_open", "acgpu_comm_close", "acgpu_comm_transport", "acgpu_comm_stream",
           "acgpu_match_device_allgather", "acgpu_last_rccl_error", "acgpu_gather_slot_bytes",
           "acgpu_stream_set_pipelined", "acgpu_stream_reserve"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "ahocorasick_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C ahocorasick_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        try:
            # PyTorch-ROCm bundles its own libamdhip64.so (same SONAME, libamdhip64.so.7).  Loading torch first makes
            # libacgpu.so bind to that copy; the other order would put two HIP/HSA runtimes in one process and the
            # second one finds no GPU.  Without torch (JNI / plain C callers) libacgpu.so uses /opt/rocm's runtime.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, u64, i64, u32, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int
        L.acgpu_build.restype = ci
        L.acgpu_build.argtypes = [ci, vp, vp, u32, ci, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
        L.acgpu_free.restype = None
        L.acgpu_free.argtypes = [vp]
        L.acgpu_get_info.restype = ci
        L.acgpu_get_info.argtypes = [vp, ctypes.POINTER(Info)]
        L.acgpu_match_u16.restype = ci
        L.acgpu_match_u16.argtypes = [vp, vp, u64, ci, vp, u64, ctypes.POINTER(u64)]
        L.acgpu_match_batch_u16.restype = ci
        L.acgpu_match_batch_u16.argtypes = [vp, vp, vp, u32, ci, vp, u64, ctypes.POINTER(u64)]
        L.acgpu_match_device.restype = ci
        L.acgpu_match_device.argtypes = [vp, ctypes.POINTER(Shard), ci, vp, u64, ctypes.POINTER(u64), vp,
                                         ctypes.POINTER(Profile)]
        L.acgpu_match_device_begin.restype = ci
        L.acgpu_match_device_begin.argtypes = [vp, ctypes.POINTER(Shard), ci, vp, u64, vp, ci, ctypes.POINTER(vp)]
        L.acgpu_match_device_end.restype = ci
        L.acgpu_match_device_end.argtypes = [vp, vp, ctypes.POINTER(u64), ctypes.POINTER(Profile)]
        L.acgpu_match_device_abandon.restype = ci
        L.acgpu_match_device_abandon.argtypes = [vp, vp]
        L.acgpu_synth_fill.restype = ci
        L.acgpu_synth_fill.argtypes = [vp, u64, u64, u64, vp, u32, vp]
        L.acgpu_synth_tokens.restype = ci
        L.acgpu_synth_tokens.argtypes = [vp, u64, u64, vp, vp, u32, vp, vp]
        L.acgpu_stream_probe.restype = ci
        L.acgpu_stream_probe.argtypes = [vp, u64, vp, ci, ci, ctypes.POINTER(ctypes.c_float)]
        L.acgpu_set_tunable.restype = i64
        L.acgpu_set_tunable.argtypes = [ctypes.c_char_p, i64]
        L.acgpu_strerror.restype = ctypes.c_char_p
        L.acgpu_strerror.argtypes = [ci]
        L.acgpu_last_hip_error.restype = ci
        L.acgpu_abi_version.restype = u32
        L.acgpu_debug_states.restype = ci
        L.acgpu_debug_states.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.acgpu_debug_tables.restype = ci
        L.acgpu_debug_tables.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(u32)]
        L.acgpu_stream_open.restype = ci
        L.acgpu_stream_open.argtypes = [vp, ctypes.POINTER(vp)]
        L.acgpu_stream_feed.restype = ci
        L.acgpu_stream_feed.argtypes = [vp, vp, u64, ci, ci, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(i64)]
        L.acgpu_stream_close.restype = None
        L.acgpu_stream_close.argtypes = [vp]
        L.acgpu_stream_set_pipelined.restype = ci
        L.acgpu_stream_set_pipelined.argtypes = [vp, ci]
        L.acgpu_stream_reserve.restype = ci
        L.acgpu_stream_reserve.argtypes = [vp, u64, ctypes.POINTER(vp)]
        L.acgpu_debug_wordhash_perfect.restype = ci
        L.acgpu_debug_wordhash_perfect.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.acgpu_debug_wordhash.restype = ci
        L.acgpu_debug_wordhash.argtypes = [vp, ctypes.POINTER(u32), vp, ctypes.POINTER(u64), vp, vp, ctypes.POINTER(u32), vp,
                                           ctypes.POINTER(u32)]
        L.acgpu_match_u16_multi.restype = ci
        L.acgpu_match_u16_multi.argtypes = [vp, vp, u64, ctypes.POINTER(ci), ci, ci, vp, u64, ctypes.POINTER(u64)]
        L.acgpu_comm_open.restype = ci
        L.acgpu_comm_open.argtypes = [ctypes.POINTER(ci), ci, ci, ctypes.POINTER(vp)]
        L.acgpu_comm_close.restype = None
        L.acgpu_comm_close.argtypes = [vp]
        L.acgpu_comm_transport.restype = ci
        L.acgpu_comm_transport.argtypes = [vp]
        L.acgpu_comm_stream.restype = vp
        L.acgpu_comm_stream.argtypes = [vp, ci]
        L.acgpu_match_device_allgather.restype = ci
        L.acgpu_match_device_allgather.argtypes = [vp, vp, ctypes.POINTER(Shard), ci, ctypes.POINTER(vp), u64, ctypes.POINTER(u64),
                                                   ctypes.POINTER(Profile)]
        L.acgpu_last_rccl_error.restype = ci
        L.acgpu_gather_slot_bytes.restype = u64
        L.acgpu_gather_slot_bytes.argtypes = [u64, ci]
        if L.acgpu_abi_version() != ABI_VERSION:
            raise ImportError("ahocorasick_amd: %s has ABI version %d, this package binds version %d -- rebuild it"
                              % (LIB_PATH, L.acgpu_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def _code_only(text):
    """C / C++ / HIP source without comments and with runs of white space collapsed: what the compiler sees of it.  (String and
    character literals are kept as they are; a comment marker inside one is left alone.)"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' and i > 0 and text[i - 1] == "R" and not (i > 1 and (text[i - 2].isalnum() or text[i - 2] == "_")):
            # raw string literal R"delim( ... )delim": copied whole
            k = text.find("(", i)
            end = text.find(")" + text[i + 1:k] + '"', k) if k >= 0 else -1
            j = n - 1 if end < 0 else end + (k - i)
            out.append(text[i:j + 1])
            i = j + 1
        elif c == "'" and i > 0 and text[i - 1].isalnum():  # a digit separator (1'000'000), not a character literal
            out.append(c)
            i += 1
        elif c in "\"'":  # literal: copy to its closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def source_hash():
    """SHA-256 (first 16 hex digits) over the CODE of the kernel and builder sources -- comments and white space are not part of
    it: what measured evidence (profiles/latest_traffic.json) is keyed by, so that a counter figure is never attached to a kernel
    that has been edited since, while an edit of a comment does not ask for a new collection."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + [os.path.join(os.path.dirname(_HERE), "include", "acgpu.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "r", encoding="utf-8", errors="replace") as fh:
            h.update(_code_only(fh.read()).encode())
    return h.hexdigest()[:16]


def check(rc, where):
    if rc != OK:
        raise AcgpuError(rc, where)


def set_tunable(name, value):
    return lib().acgpu_set_tunable(name.encode(), int(value))
