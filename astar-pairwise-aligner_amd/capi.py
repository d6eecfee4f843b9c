"""ctypes binding of libastarpa_c_hip.so -- exactly the symbols include/*.h declare.

There is no CPU fallback: if the library is missing or no GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import _build

U64P = C.POINTER(C.c_uint64)

# every symbol include/astarpa.h and include/pa_bitpacking_hip.h declare
EXPORTED_SYMBOLS = [
    "astarpa2_simple", "astarpa2_full", "astarpa", "astarpa_gcsh", "astarpa_free_cigar",
    "pa_last_error", "pa_device_count", "pa_set_device",
    "pa_bp_profile_build", "pa_bp_compute", "pa_bp_fill", "pa_search", "pa_search_trace",
    "pa_batch_create", "pa_batch_run", "pa_batch_stats", "pa_batch_shape", "pa_batch_destroy",
    "pa_batch_create_banded", "pa_batch_create_trace", "pa_batch_align", "pa_batch_align_view", "pa_batch_trace_fallbacks", "pa_params_batch_align",
    "pa_pairs_read", "pa_pairs_count", "pa_pairs_get", "pa_pairs_free", "pa_write_results_csv", "pa_align_file",
    "pa_align", "pa_batch_align_multi", "pa_batch_create_trace_params",
    "pa_bp_ctx_create", "pa_bp_ctx_compute", "pa_bp_ctx_fill", "pa_bp_ctx_destroy",
    "pa_batch_create_params", "pa_batch_pair_stats", "pa_runtime_hints", "pa_batch_align_multi_params", "pa_release_pools", "pa_align_file_params", "pa_batch_params_supported", "pa_alloc_cache_stats", "pa_free_cigars",
    "pa_batch_full_info", "pa_batch_rdv_stats", "pa_combine_stats", "pa_params_nw", "pa_params_simple", "pa_params_full", "pa_debug_gcsh_probe", "pa_debug_gcsh_matches", "pa_batch_window_retries", "pa_batch_window_retry_bytes",
    "pa_batch_slice_info", "pa_set_reference_cost_only",
]

_lib = None


class PaError(RuntimeError):
    pass


def load(build_if_stale: bool = True) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if os.environ.get("PA_LIB_PATH"):  # diagnostics: an alternative build of the same sources (e.g. -DPA_SWEEP_PHASE_TIMERS)
        from pathlib import Path

        path, build_if_stale = Path(os.environ["PA_LIB_PATH"]).resolve(), False
    if build_if_stale and _build.is_stale():
        try:
            _build.build()
        except Exception as e:  # stale but present is still usable on a box without hipcc
            if not path.exists():
                raise PaError(f"cannot build {path.name}: {e}") from e
    if not path.exists():
        raise PaError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    L = C.CDLL(str(path))
    L.pa_runtime_hints.restype = C.c_int
    if hasattr(L, "pa_set_reference_cost_only"):  # (PA_LIB_PATH may point at an older build of the library: diagnostics)
        L.pa_set_reference_cost_only.argtypes = [C.c_int]
        L.pa_set_reference_cost_only.restype = None
    L.pa_release_pools.restype = None
    L.pa_alloc_cache_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
    L.pa_free_cigars.argtypes = [C.c_void_p, C.c_size_t]
    L.pa_free_cigars.restype = None
    L.pa_batch_params_supported.argtypes = [C.c_void_p]
    L.pa_batch_params_supported.restype = C.c_int
    L.pa_alloc_cache_stats.restype = None
    L.pa_runtime_hints()  # this package is the application here: more hardware queues, before the first HIP call
    vp, sz = C.c_void_p, C.c_size_t
    L.pa_last_error.restype = C.c_char_p
    L.pa_device_count.restype = C.c_int
    L.pa_set_device.argtypes = [C.c_int]
    L.pa_bp_profile_build.argtypes = [vp, sz, vp, sz, vp, vp]
    L.pa_bp_profile_build.restype = C.c_int
    L.pa_bp_compute.argtypes = [vp, sz, vp, sz, vp, vp, C.c_int]
    L.pa_bp_compute.restype = C.c_int32
    L.pa_bp_fill.argtypes = [vp, sz, vp, sz, vp, vp, vp]
    L.pa_bp_fill.restype = C.c_int32
    L.pa_search.argtypes = [vp, sz, vp, sz, C.c_float, vp]
    L.pa_search.restype = C.c_int
    L.pa_search_trace.argtypes = [vp, sz, vp, sz, C.c_float, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]
    L.pa_search_trace.restype = C.c_int
    L.pa_batch_create.argtypes = [vp, vp, vp, vp, sz]
    L.pa_batch_create.restype = vp
    L.pa_batch_run.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.pa_batch_run.restype = C.c_int
    L.pa_batch_stats.argtypes = [vp] + [C.POINTER(C.c_double)] * 4
    L.pa_batch_full_info.argtypes = [vp] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_double)]
    L.pa_batch_full_info.restype = None
    L.pa_batch_rdv_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.pa_combine_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pa_combine_stats.restype = None
    L.pa_batch_rdv_stats.restype = C.c_int
    L.pa_batch_shape.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.pa_batch_slice_info.argtypes = [vp] + [C.POINTER(C.c_double)] * 5
    L.pa_batch_slice_info.restype = C.c_int
    L.pa_batch_destroy.argtypes = [vp]
    L.pa_batch_create_banded.argtypes = [vp, vp, vp, vp, sz, C.c_float]
    L.pa_batch_create_banded.restype = vp
    L.pa_batch_create_trace.argtypes = [vp, vp, vp, vp, sz]
    L.pa_batch_create_trace.restype = vp
    L.pa_bp_ctx_create.argtypes = [vp, sz, vp, sz]
    L.pa_bp_ctx_create.restype = vp
    L.pa_bp_ctx_compute.argtypes = [vp, C.c_int32, C.c_int32, sz, sz, vp, C.c_int, C.POINTER(C.c_int32)]
    L.pa_bp_ctx_compute.restype = C.c_int
    L.pa_bp_ctx_fill.argtypes = [vp, C.c_int32, C.c_int32, sz, sz, vp, vp, vp]
    L.pa_bp_ctx_fill.restype = C.c_int
    L.pa_bp_ctx_destroy.argtypes = [vp]
    L.pa_batch_create_trace_params.argtypes = [vp, vp, vp, vp, sz, vp]
    L.pa_batch_create_trace_params.restype = vp
    L.pa_batch_create_params.argtypes = [vp, vp, vp, vp, sz, vp]
    L.pa_batch_create_params.restype = vp
    L.pa_batch_pair_stats.argtypes = [vp, vp]
    L.pa_batch_pair_stats.restype = C.c_int
    L.pa_batch_align.argtypes = [vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pa_batch_align.restype = C.c_int
    L.pa_batch_align_view.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pa_batch_align_view.restype = C.c_int
    L.pa_batch_trace_fallbacks.argtypes = [vp]
    L.pa_batch_trace_fallbacks.restype = C.c_size_t
    L.pa_batch_window_retries.argtypes = [vp]
    L.pa_batch_window_retries.restype = C.c_size_t
    L.pa_pairs_read.argtypes = [C.c_char_p]
    L.pa_pairs_read.restype = vp
    L.pa_pairs_count.argtypes = [vp]
    L.pa_pairs_count.restype = sz
    L.pa_pairs_get.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
    L.pa_pairs_get.restype = C.c_int
    L.pa_pairs_free.argtypes = [vp]
    L.pa_write_results_csv.argtypes = [C.c_char_p, vp, vp, sz]
    L.pa_write_results_csv.restype = C.c_int
    L.pa_batch_align_multi.argtypes = [vp, vp, vp, vp, sz, C.POINTER(C.c_int), C.c_int, vp, vp]
    L.pa_batch_align_multi.restype = C.c_int
    L.pa_batch_align_multi_params.argtypes = [vp, vp, vp, vp, sz, C.POINTER(C.c_int), C.c_int, vp, vp, vp, vp]
    L.pa_batch_align_multi_params.restype = C.c_int
    L.pa_align_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(sz)]
    L.pa_align_file.restype = C.c_int
    L.pa_align_file_params.argtypes = [C.c_char_p, C.c_char_p, vp, C.POINTER(sz)]
    L.pa_align_file_params.restype = C.c_int
    for name in ("astarpa2_simple", "astarpa2_full", "astarpa"):
        if hasattr(L, name):
            f = getattr(L, name)
            f.argtypes = [vp, sz, vp, sz, C.POINTER(vp), C.POINTER(sz)]
            f.restype = C.c_uint64
    if hasattr(L, "astarpa_gcsh"):
        L.astarpa_gcsh.argtypes = [vp, sz, vp, sz, sz, sz, C.c_bool, C.POINTER(vp), C.POINTER(sz)]
        L.astarpa_gcsh.restype = C.c_uint64
    if hasattr(L, "astarpa_free_cigar"):
        L.astarpa_free_cigar.argtypes = [vp]
    _lib = L
    return L


def last_error() -> str:
    return (load().pa_last_error() or b"").decode()


def _p(arr: np.ndarray):
    return arr.ctypes.data_as(C.c_void_p)


def _buf(b: bytes):
    return C.cast(C.c_char_p(b), C.c_void_p)


def _marshal_pairs(pairs):
    """(a pointers, a lengths, b pointers, b lengths) for the `const uint8_t* const*` / `const size_t*` arguments.  c_char_p arrays
    point straight into the bytes objects (the caller keeps `pairs` alive); building them is 40x cheaper than one ctypes cast per
    sequence, which used to cost more than the GPU work of a 10 000-pair batch."""
    n = len(pairs)
    if any(not isinstance(a, bytes) or not isinstance(b, bytes) for a, b in pairs):
        raise ValueError("pairs must be (bytes, bytes) tuples")
    ap = (C.c_char_p * max(n, 1))(*[a for a, _ in pairs])
    bp = (C.c_char_p * max(n, 1))(*[b for _, b in pairs])
    al = np.fromiter((len(a) for a, _ in pairs), dtype=np.uint64, count=n) if n else np.zeros(1, np.uint64)
    bl = np.fromiter((len(b) for _, b in pairs), dtype=np.uint64, count=n) if n else np.zeros(1, np.uint64)
    return ap, _p(al), bp, _p(bl)


def alloc_cache_stats() -> dict:
    """pa_alloc_cache_stats: the library's cache of large device buffers (pa_astarpa2.h)."""
    h, m, b = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    load().pa_alloc_cache_stats(C.byref(h), C.byref(m), C.byref(b))
    return {"hits": h.value, "misses": m.value, "cached_bytes": b.value}


def batch_params_supported(params) -> bool:
    """pa_batch_params_supported: does pa_batch_create_params take this AstarPa2Params (Domain::Astar over sparse 256-column blocks with a
    search: the `simple` and `full` presets and their relatives)?"""
    cp = params._to_c()
    return load().pa_batch_params_supported(C.byref(cp)) == 1


def gcsh_probe(a: bytes, b: bytes, k: int, p: int, queries) -> tuple[list[int], int]:
    """pa_debug_gcsh_probe (diagnostics): GCSH h at `queries` [(i, j), ..] as the GPU computes it -- contours derived and probed by one
    wavefront (csrc/apa2_full_kernel.hpp) -- and the number of contour layers."""
    L = load()
    L.pa_debug_gcsh_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
    L.pa_debug_gcsh_probe.restype = C.c_int
    q = np.ascontiguousarray(np.array(queries, np.int32).reshape(-1, 2))
    out = np.zeros(len(q) + 1, np.int32)
    rc = L.pa_debug_gcsh_probe(_buf(a), len(a), _buf(b), len(b), k, p, _p(q), len(q), _p(out))
    if rc != 0:
        raise PaError(f"pa_debug_gcsh_probe rc={rc}: {last_error()}")
    return out[:-1].tolist(), int(out[-1])


def gcsh_matches(a: bytes, b: bytes, k: int, p: int) -> list[tuple[int, int]]:
    """pa_debug_gcsh_matches (diagnostics): the matches GCSH keeps for this pair as the GPU finds them (csrc/gcsh_build_kernel.hpp), by start."""
    L = load()
    L.pa_debug_gcsh_matches.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]
    L.pa_debug_gcsh_matches.restype = C.c_long
    cap = len(a) // max(1, k) * 2 + 4096
    out = np.zeros((cap, 2), np.int32)
    n = L.pa_debug_gcsh_matches(_buf(a), len(a), _buf(b), len(b), k, p, _p(out), cap)
    if n < 0:
        raise PaError(f"pa_debug_gcsh_matches rc={n}: {last_error()}")
    return [tuple(x) for x in out[:n].tolist()]


def release_pools() -> None:
    """pa_release_pools: the calling thread's engine pools and the cache of large device buffers back to the driver."""
    load().pa_release_pools()


def require_gpu() -> None:
    if load().pa_device_count() <= 0:
        raise PaError("no MI355X visible: the HIP path is required (no CPU fallback)")


def profile_build(a: bytes, b: bytes):
    """BitProfile::build on the GPU -> (a2[n,2], b2[w,2]) uint64."""
    L = load()
    a2 = np.zeros((len(a), 2), np.uint64)
    b2 = np.zeros(((len(b) + 63) // 64, 2), np.uint64)
    rc = L.pa_bp_profile_build(_buf(a), len(a), _buf(b), len(b), _p(a2), _p(b2))
    if rc == -1:
        raise ValueError("sequence contains a character outside ACGT")
    if rc != 0:
        raise PaError(last_error())
    return a2, b2


def compute(a2, b2, h2, v2, exact_end: bool) -> int:
    r = load().pa_bp_compute(_p(a2), len(a2), _p(b2), len(b2), _p(h2), _p(v2), int(exact_end))
    if r == -(2 ** 31):
        raise PaError(last_error())
    return r


def fill(a2, b2, h2, v2):
    values = np.zeros((len(a2), len(b2), 2), np.uint64)
    r = load().pa_bp_fill(_p(a2), len(a2), _p(b2), len(b2), _p(h2), _p(v2), _p(values))
    if r == -(2 ** 31):
        raise PaError(last_error())
    return r, values


def search(pattern: bytes, text: bytes, unmatched_cost: float) -> list[int]:
    """pa_bitpacking::search(pattern, text, unmatched_cost).out on the GPU (search.rs:46-120)."""
    out = np.zeros(len(pattern) + len(text) + 1, np.int32)
    rc = load().pa_search(_buf(pattern), len(pattern), _buf(text), len(text), unmatched_cost, _p(out))
    if rc == -1:
        raise ValueError("unknown base")
    if rc != 0:
        raise PaError(f"pa_search rc={rc}: {last_error()}")
    return out.tolist()


def search_trace(pattern: bytes, text: bytes, unmatched_cost: float, idx: int):
    """SearchResult::trace(idx) (search.rs:125-228) -> (CIGAR string, [(text index, pattern index), ...])."""
    L = load()
    cig, path, npos = C.c_void_p(None), C.c_void_p(None), C.c_size_t(0)
    rc = L.pa_search_trace(_buf(pattern), len(pattern), _buf(text), len(text), unmatched_cost, idx, C.byref(cig), C.byref(path),
                           C.byref(npos))
    try:
        if rc == -1:
            raise ValueError("unknown base")
        if rc != 0:
            raise PaError(f"pa_search_trace rc={rc}: {last_error()}")
        text_cigar = C.string_at(cig.value).decode()
        arr = np.ctypeslib.as_array(C.cast(path.value, C.POINTER(C.c_int32)), shape=(2 * npos.value,)).copy() if npos.value else np.zeros(0, np.int32)
    finally:
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        if cig.value:
            libc.free(cig)
        if path.value:
            libc.free(path)
    return text_cigar, [(int(arr[2 * k]), int(arr[2 * k + 1])) for k in range(npos.value)]


def read_pairs(path: str) -> list[tuple[bytes, bytes]]:
    """Sequence pairs of a pa-bin input file or directory (.seq / .txt / .fna / .fa / .fasta; pa-bin/src/lib.rs:67-114)."""
    L = load()
    h = L.pa_pairs_read(str(path).encode())
    if not h:
        raise PaError(last_error())
    try:
        out = []
        for i in range(L.pa_pairs_count(h)):
            a, b, al, bl = C.c_void_p(None), C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
            L.pa_pairs_get(h, i, C.byref(a), C.byref(al), C.byref(b), C.byref(bl))
            out.append((C.string_at(a.value, al.value) if al.value else b"", C.string_at(b.value, bl.value) if bl.value else b""))
        return out
    finally:
        L.pa_pairs_free(h)


def align_file(input_path: str, output_path: str, params=None) -> int:
    """pa-bin's main loop on the GPU: every pair of the input gets one "{cost},{cigar}" line.  Returns the pair count.
    params: an AstarPa2Params (pa_align_file_params: batched A*PA2 for the `simple` family, a loop over pa_align otherwise)."""
    n = C.c_size_t(0)
    if params is not None:
        cp = params._to_c()
        rc = load().pa_align_file_params(str(input_path).encode(), str(output_path).encode(), C.byref(cp), C.byref(n))
    else:
        rc = load().pa_align_file(str(input_path).encode(), str(output_path).encode(), C.byref(n))
    if rc == -1:
        raise ValueError("sequence contains a character outside ACGT")
    if rc != 0:
        raise PaError(f"pa_align_file rc={rc}: {last_error()}")
    return int(n.value)


_str_from_c = C.pythonapi.PyUnicode_FromString  # C string -> str in one pass (string_at().decode() copies twice: 18 -> 9 ms per 10 000 CIGARs)
_str_from_c.restype = C.py_object
_str_from_c.argtypes = [C.c_void_p]


_str_from_c_n = C.pythonapi.PyUnicode_FromStringAndSize
_str_from_c_n.restype = C.py_object
_str_from_c_n.argtypes = [C.c_void_p, C.c_ssize_t]


def _c_strings(ptrs, n: int) -> list[str]:
    """The NUL-terminated strings a C call left in the pointer array `ptrs` (NULL -> "")."""
    return [_str_from_c(p) if p else "" for p in list(ptrs)[:n]]


class OperatorContext:
    """Device-resident operator handle (pa_bp_ctx_*): a, b, the profile and the stored h row stay on the GPU between calls."""

    H_NONE, H_INPUT, H_UPDATE, H_OUTPUT = 0, 1, 2, 3

    def __init__(self, a: bytes, b: bytes):
        L = load()
        self._keep = (a, b)
        self._h = L.pa_bp_ctx_create(C.cast(C.c_char_p(a), C.c_void_p), len(a), C.cast(C.c_char_p(b), C.c_void_p), len(b))
        if not self._h:
            raise PaError(last_error())

    @staticmethod
    def _check_v(v: np.ndarray, w0: int, w1: int) -> None:
        """The library reads and writes 16 * (w1 - w0) bytes at v: refuse anything that is not exactly that, in place."""
        if not isinstance(v, np.ndarray) or v.dtype != np.uint64 or v.shape != (w1 - w0, 2) or not v.flags["C_CONTIGUOUS"] or not v.flags["WRITEABLE"]:
            raise ValueError(f"v must be a writable C-contiguous uint64 array of shape ({w1 - w0}, 2)")

    def compute(self, i0: int, i1: int, w0: int, w1: int, v: np.ndarray, h_mode: int = 0) -> int:
        """v: uint64[w1 - w0, 2] (p, m), updated in place; returns the sum of the bottom-row deltas."""
        self._check_v(v, w0, w1)
        s = C.c_int32(0)
        rc = load().pa_bp_ctx_compute(self._h, i0, i1, w0, w1, _p(v), h_mode, C.byref(s))
        if rc != 0:
            raise PaError(f"pa_bp_ctx_compute rc={rc}: {last_error()}")
        return int(s.value)

    def fill(self, i0: int, i1: int, w0: int, w1: int, v: np.ndarray):
        """-> (values uint64[i1 - i0, w1 - w0, 2], bottom-row deltas int8[i1 - i0]); v updated in place."""
        self._check_v(v, w0, w1)
        values = np.zeros((i1 - i0, w1 - w0, 2), np.uint64)
        hb = np.zeros(max(i1 - i0, 1), np.int8)
        rc = load().pa_bp_ctx_fill(self._h, i0, i1, w0, w1, _p(v), _p(values), _p(hb))
        if rc != 0:
            raise PaError(f"pa_bp_ctx_fill rc={rc}: {last_error()}")
        return values, hb[: i1 - i0]

    def close(self):
        if self._h:
            load().pa_bp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_multi(pairs: list[tuple[bytes, bytes]], devices: list[int], trace: bool = True, params=None, stats: bool = False):
    """pa_batch_align_multi[_params]: a work queue of chunks of `pairs` served by one host thread per entry of `devices` (inside the
    library) -> (costs, CIGARs or None[, per-pair statistics]).  params: an AstarPa2Params of the `simple` family -> batched A*PA2."""
    L = load()
    n = len(pairs)
    if not devices or any(not isinstance(d, int) for d in devices):
        raise ValueError("devices must be a non-empty list of device indices")
    ap, al, bp, bl = _marshal_pairs(pairs)
    dev = (C.c_int * len(devices))(*devices)
    out = np.zeros(n, np.int32)
    cig = (C.c_void_p * n)() if trace else None
    st = None
    if params is not None:
        from .aligner import _StatsC

        cp = params._to_c()
        st = (_StatsC * max(n, 1))() if stats else None
        rc = L.pa_batch_align_multi_params(ap, al, bp, bl, n, dev, len(devices), C.byref(cp), _p(out), cig, st)
    else:
        rc = L.pa_batch_align_multi(ap, al, bp, bl, n, dev, len(devices), _p(out), cig)
    if rc == -1:
        raise ValueError("sequence contains a character outside ACGT")
    if rc != 0:
        raise PaError(f"pa_batch_align_multi rc={rc}: {last_error()}")
    cigars = None
    if trace:
        try:
            cigars = _c_strings(cig, n)
        finally:
            L.pa_free_cigars(cig, n)
    if st is not None:
        from .aligner import _StatsC

        return out, cigars, [{k: getattr(st[i], k) for k, _ in _StatsC._fields_} for i in range(n)]
    return out, cigars


class Batch:
    """Device-resident batch of independent pairs; run() = full-DP edit distance of every pair."""

    def __init__(self, pairs: list[tuple[bytes, bytes]], trace: bool = False, band: float | None = None, trace_params=None, params=None):
        """band: expected edit rate (e.g. 0.05) -> diagonal-band DP, re-run wider where it was too narrow (still exact).
        trace_params: an AstarPa2Params whose `front` (dt_trace, max_g, fr_drop) the batched traceback follows.
        params: an AstarPa2Params of the `simple` or `full` family -> batched A*PA2 (pa_batch_create_params): align() returns what a loop over
        AstarPa2(params).align(a, b) returns, pair_stats() the statistics."""
        L = load()
        self._keep = pairs
        self.trace = trace
        n = len(pairs)
        ap, al, bp, bl = _marshal_pairs(pairs)
        self.astar = params is not None
        if band is not None and trace:
            raise ValueError("banded batches are cost-only")
        if params is not None:
            cp = params._to_c()
            self.trace = True
            self._h = L.pa_batch_create_params(ap, al, bp, bl, n, C.byref(cp))
        elif band is not None:
            self._h = L.pa_batch_create_banded(ap, al, bp, bl, n, C.c_float(band))
        elif trace and trace_params is not None:
            cp = trace_params._to_c()
            self._h = L.pa_batch_create_trace_params(ap, al, bp, bl, n, C.byref(cp))
        else:
            self._h = (L.pa_batch_create_trace if trace else L.pa_batch_create)(ap, al, bp, bl, n)
        if not self._h:
            raise PaError(last_error())
        self.pairs = n

    def run(self):
        """-> (costs int32[pairs], strip-kernel milliseconds from HIP events)."""
        L = load()
        out = np.zeros(self.pairs, np.int32)
        ms = C.c_float(0)
        rc = L.pa_batch_run(self._h, _p(out), C.byref(ms))
        if rc == -1:
            raise ValueError("sequence contains a character outside ACGT")
        if rc != 0:
            raise PaError(f"pa_batch_run rc={rc}: {last_error()}")
        return out, float(ms.value)

    def align(self):
        """-> (costs int32[pairs], CIGAR strings, forward-kernel ms, traceback-kernel ms); needs trace=True.
        `self.last_c_abi_ms` = wall time of the C call itself (before Python turns the texts into str).  The call is pa_batch_align_view:
        the texts stay in the plan's host buffer and become str objects straight from there (pa_batch_align would malloc one C string per
        pair first, for Python to copy and free again)."""
        import time

        L = load()
        n = self.pairs
        out = np.zeros(n, np.int32)
        txt = (C.c_void_p * max(n, 1))()
        lens = np.zeros(max(n, 1), np.uint32)
        fms, tms = C.c_float(0), C.c_float(0)
        t0 = time.perf_counter()
        rc = L.pa_batch_align_view(self._h, _p(out), txt, _p(lens), C.byref(fms), C.byref(tms))
        self.last_c_abi_ms = (time.perf_counter() - t0) * 1e3
        if rc == -1:
            raise ValueError("sequence contains a character outside ACGT")
        if rc != 0:
            raise PaError(f"pa_batch_align_view rc={rc}: {last_error()}")
        cigars = [_str_from_c_n(p, l) if p else "" for p, l in zip(list(txt)[:n], lens[:n].tolist())]
        return out, cigars, float(fms.value), float(tms.value)

    def align_c_strings(self):
        """The same through pa_batch_align (one malloc'ed NUL-terminated string per pair, released with pa_free_cigars): the entry point a C
        caller of the reference's style uses.  -> (costs, CIGAR strings, forward-kernel ms, traceback-kernel ms)."""
        import time

        L = load()
        out = np.zeros(self.pairs, np.int32)
        cig = (C.c_void_p * max(self.pairs, 1))()
        fms, tms = C.c_float(0), C.c_float(0)
        t0 = time.perf_counter()
        rc = L.pa_batch_align(self._h, _p(out), cig, C.byref(fms), C.byref(tms))
        self.last_c_abi_ms = (time.perf_counter() - t0) * 1e3
        try:
            if rc == -1:
                raise ValueError("sequence contains a character outside ACGT")
            if rc != 0:
                raise PaError(f"pa_batch_align rc={rc}: {last_error()}")
            cigars = _c_strings(cig, self.pairs)
        finally:
            L.pa_free_cigars(cig, self.pairs)
        return out, cigars, float(fms.value), float(tms.value)

    def pair_stats(self) -> list[dict]:
        """AstarPa2Stats of every pair of the last align() (batches made with `params`)."""
        from .aligner import _StatsC

        arr = (_StatsC * max(self.pairs, 1))()
        rc = load().pa_batch_pair_stats(self._h, arr)
        if rc != 0:
            raise PaError(f"pa_batch_pair_stats rc={rc}: {last_error()}")
        return [{n: getattr(arr[i], n) for n, _ in _StatsC._fields_} for i in range(self.pairs)]

    def full_info(self) -> dict:
        """pa_batch_full_info: reporting for batches of the `full` family."""
        vals = [C.c_double(0) for _ in range(4)]
        ph = (C.c_double * 8)()
        load().pa_batch_full_info(self._h, *[C.byref(v) for v in vals], ph)
        d = dict(zip(("build_ms", "matches", "probes", "rounds"), (v.value for v in vals)))
        d["phase_wave_ms"] = dict(zip(("contours", "dp", "h", "index", "prune", "init", "total"), [x for x in ph][:7]))
        return d

    def rdv_stats(self) -> dict:
        """pa_batch_rdv_stats: how the half-wave blocks of the last forward pass met (batched A*PA2; diagnostics)."""
        out = (C.c_uint64 * 4)()
        rc = load().pa_batch_rdv_stats(self._h, out)
        if rc != 0:
            raise PaError(f"pa_batch_rdv_stats rc={rc}: {last_error()}")
        return dict(zip(("fused", "served", "alone", "withdrawn"), (int(x) for x in out)))

    def trace_fallbacks(self) -> int:
        return int(load().pa_batch_trace_fallbacks(self._h))

    def window_retry_bytes(self) -> float:
        """The largest full-height block-column store a second round (pairs whose band left their window) of this plan held at a time."""
        f = load().pa_batch_window_retry_bytes
        f.argtypes, f.restype = [C.c_void_p], C.c_double
        return float(f(self._h))

    def window_retries(self) -> int:
        """Pairs whose band left their window of the block-column store and were aligned again with full-height columns."""
        return int(load().pa_batch_window_retries(self._h))

    def stats(self) -> dict:
        vals = [C.c_double(0) for _ in range(4)]
        load().pa_batch_stats(self._h, *[C.byref(v) for v in vals])
        return dict(zip(("cells", "word_updates", "strips", "algo_bytes"), (v.value for v in vals)))

    def shape(self) -> dict:
        """How the batch was laid out: strip height k, chained strips or one wavefront per pair, VALU instructions per pass."""
        k, seq, vi = C.c_int(0), C.c_int(0), C.c_double(0)
        load().pa_batch_shape(self._h, C.byref(k), C.byref(seq), C.byref(vi))
        sl = [C.c_double(0) for _ in range(5)]
        rows = load().pa_batch_slice_info(self._h, *[C.byref(v) for v in sl])
        if rows:  # groups of 32 pairs, bit-sliced (csrc/slice_kernel.hpp)
            return {"k": 0, "sequential": False, "valu_instructions": vi.value, "kernel": f"pa::slice::slice_kernel<{rows}>", "sliced_rows_per_lane": rows,
                    "groups": int(sl[0].value), "jobs": int(sl[1].value), "computed_cells": sl[2].value, "device_bytes": sl[3].value,
                    "boundary_bytes": sl[4].value}
        return {"k": k.value, "sequential": bool(seq.value), "valu_instructions": vi.value,
                "kernel": f"pa::pair_kernel<{k.value}>" if seq.value else f"pa::strip_kernel<{k.value}, false, false>"}

    def close(self):
        if self._h:
            load().pa_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
