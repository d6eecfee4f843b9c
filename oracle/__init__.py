"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of the plain-C CPU restatement of the reference's hot path (oracle/pa_oracle.c,
oracle/strip_avx2.c, oracle/engine_cpu.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg import this package, and only as the checker / reported CPU baseline; the
product (astar-pairwise-aligner_amd) never does.

Parity status: pinned against the reference's own known answers (see pa_oracle.h header and
tests/test_oracle_kat.py); exact A*PA2 CIGAR strings are "parity unpinned" because the reference's
tests never compare them and the Rust reference cannot be built in this image.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB_PATH = _DIR / "_build" / "libpa_oracle.so"

V_DTYPE = np.dtype([("p", "<u8"), ("m", "<u8")])
H_DTYPE = np.dtype([("p", "<u8"), ("m", "<u8")])
BITS_DTYPE = np.dtype([("b0", "<u8"), ("b1", "<u8")])


_TARGETS = ("libpa_oracle.so", "libpa_engine_cpu.so", "libpa_sweep_emu.so", "libpa_apa2_emu.so", "libpa_apa2_full_emu.so", "libpa_rdv_emu.so", "libpa_combine_emu.so")


def _source_hash() -> str:
    """Content hash of everything the oracle libraries are built from (file times do not survive the copy to a GPU box, and make would
    rebuild the lot there -- a minute of g++ in front of the first GPU test)."""
    import hashlib

    h = hashlib.sha256()
    csrc = _DIR.parent / "astar-pairwise-aligner_amd" / "csrc"
    files = sorted(list(_DIR.glob("*.c")) + list(_DIR.glob("*.cpp")) + list(_DIR.glob("*.h")) + list(_DIR.glob("*.hpp")) + [_DIR / "Makefile"] +
                   [csrc / n for n in ("engine.hpp", "gcsh.hpp", "engine_capi.hpp", "sweep_logic.hpp", "sweep_wave.hpp", "sweep_host.hpp", "apa2_logic.hpp", "apa2_full_logic.hpp", "gcsh_dev.hpp", "rdv_logic.hpp", "combine_logic.hpp")] +
                   [_DIR.parent / "include" / "pa_astarpa2.h"])
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    # the libraries are built with -march=native: a tree copied to another machine (the GPU box) must not reuse binaries built for
    # this CPU, nor ones made by another compiler
    try:
        flags = next((ln for ln in Path("/proc/cpuinfo").read_text().splitlines() if ln.startswith("flags")), "")
        h.update(" ".join(sorted(flags.split(":", 1)[-1].split())).encode())
    except OSError:
        pass
    try:
        h.update(subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0].encode())
    except (OSError, IndexError):
        pass
    return h.hexdigest()


def build() -> Path:
    """Compile the oracle with gcc/g++ (oracle/Makefile).  Skipped when the libraries exist and were built from exactly these sources."""
    stamp = _DIR / "_build" / ".source_hash"
    cur = _source_hash()
    if all((_DIR / "_build" / t).exists() for t in _TARGETS) and stamp.exists() and stamp.read_text().strip() == cur:
        return _LIB_PATH
    subprocess.run(["make", "-C", str(_DIR), "-s"], check=True)
    stamp.write_text(cur + "\n")
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        vp = C.c_void_p
        sz = C.c_size_t
        L.pa_or_bitprofile_build.argtypes = [vp, sz, vp, sz, vp, vp]
        L.pa_or_bitprofile_build.restype = C.c_int
        for name in ("pa_or_scalar_row", "pa_or_scalar_col"):
            f = getattr(L, name)
            f.argtypes = [vp, sz, vp, sz, vp, vp]
            f.restype = C.c_int32
        L.pa_or_scalar_fill.argtypes = [vp, sz, vp, sz, vp, vp, vp]
        L.pa_or_scalar_fill.restype = C.c_int32
        L.pa_or_simd_fill.argtypes = [vp, sz, vp, sz, vp, vp, vp]
        L.pa_or_simd_fill.restype = C.c_int32
        L.pa_or_simd_compute.argtypes = [vp, sz, vp, sz, vp, vp, C.c_int, C.c_int]
        L.pa_or_simd_compute.restype = C.c_int32
        L.pa_or_simd_pad_rows.argtypes = [sz, sz, C.c_int, C.c_int]
        L.pa_or_simd_pad_rows.restype = sz
        L.pa_or_strip_compute_avx2.argtypes = [vp, sz, vp, sz, vp, vp, C.c_int]
        L.pa_or_strip_compute_avx2.restype = C.c_int32
        L.pa_or_nw_cost.argtypes = [vp, sz, vp, sz, C.c_int]
        L.pa_or_nw_cost.restype = C.c_int32
        L.pa_or_search.argtypes = [vp, sz, vp, sz, C.c_float, vp]
        L.pa_or_search.restype = C.c_int
        L.pa_or_search_trace.argtypes = [vp, sz, vp, sz, C.c_float, sz, vp, sz, vp, sz, C.POINTER(sz)]
        L.pa_or_search_trace.restype = C.c_int
        L.pa_or_levenshtein.argtypes = [vp, sz, vp, sz]
        L.pa_or_levenshtein.restype = C.c_int32
        L.pa_or_cigar_verify.argtypes = [C.c_char_p, vp, sz, vp, sz]
        L.pa_or_cigar_verify.restype = C.c_int32
        _lib = L
    return _lib


def _buf(b: bytes):
    return C.cast(C.c_char_p(b), C.c_void_p)


def _p(arr: np.ndarray):
    return arr.ctypes.data_as(C.c_void_p)


def bitprofile_build(a: bytes, b: bytes):
    """BitProfile::build -> (pa[n], pb[ceil(m/64)]) as structured arrays."""
    pa = np.zeros(max(len(a), 1), BITS_DTYPE)[: len(a)]
    pb = np.zeros(max((len(b) + 63) // 64, 1), BITS_DTYPE)[: (len(b) + 63) // 64]
    rc = lib().pa_or_bitprofile_build(_buf(a), len(a), _buf(b), len(b), _p(pa), _p(pb))
    if rc != 0:
        raise ValueError("sequence contains a character outside ACGT")
    return pa, pb


def ones_h(n: int) -> np.ndarray:
    h = np.zeros(n, H_DTYPE)
    h["p"] = 1
    return h


def ones_v(w: int) -> np.ndarray:
    v = np.zeros(w, V_DTYPE)
    v["p"] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return v


def scalar_row(pa, pb, h, v) -> int:
    return lib().pa_or_scalar_row(_p(pa), len(pa), _p(pb), len(pb), _p(h), _p(v))


def scalar_col(pa, pb, h, v) -> int:
    return lib().pa_or_scalar_col(_p(pa), len(pa), _p(pb), len(pb), _p(h), _p(v))


def scalar_fill(pa, pb, h, v):
    values = np.zeros((len(pa), len(pb)), V_DTYPE)
    r = lib().pa_or_scalar_fill(_p(pa), len(pa), _p(pb), len(pb), _p(h), _p(v), _p(values))
    return r, values


def simd_compute(pa, pb, h, v, exact_end: bool, ilp_n: int = 2) -> int:
    return lib().pa_or_simd_compute(_p(pa), len(pa), _p(pb), len(pb), _p(h), _p(v), int(exact_end), ilp_n)


def simd_pad_rows(n: int, w: int, exact_end: bool, ilp_n: int = 2) -> int:
    return lib().pa_or_simd_pad_rows(n, w, int(exact_end), ilp_n)


def strip_compute_avx2(pa, pb, h, v, exact_end: bool) -> int:
    return lib().pa_or_strip_compute_avx2(_p(pa), len(pa), _p(pb), len(pb), _p(h), _p(v), int(exact_end))


def nw_cost(a: bytes, b: bytes, use_avx2: bool = True) -> int:
    return lib().pa_or_nw_cost(_buf(a), len(a), _buf(b), len(b), int(use_avx2))


def search(pattern: bytes, text: bytes, unmatched_cost: float) -> list[int]:
    out = np.zeros(len(pattern) + len(text) + 1, np.int32)
    rc = lib().pa_or_search(_buf(pattern), len(pattern), _buf(text), len(text), unmatched_cost, _p(out))
    if rc != 0:
        raise ValueError(f"pa_or_search failed rc={rc}")
    return out.tolist()


def levenshtein(a: bytes, b: bytes) -> int:
    return lib().pa_or_levenshtein(_buf(a), len(a), _buf(b), len(b))


def search_trace(pattern: bytes, text: bytes, unmatched_cost: float, idx: int):
    """SearchResult::trace(idx) (search.rs:125-228) -> (cigar string, [(text index, pattern index), ...])."""
    cap = len(pattern) + len(text) + 8
    cig = C.create_string_buffer(2 * cap)
    path = np.zeros(2 * cap, np.int32)
    npos = C.c_size_t(0)
    rc = lib().pa_or_search_trace(_buf(pattern), len(pattern), _buf(text), len(text), unmatched_cost, idx, cig, 2 * cap,
                                  _p(path), cap, C.byref(npos))
    if rc != 0:
        raise ValueError(f"pa_or_search_trace failed rc={rc}")
    return cig.value.decode(), [(int(path[2 * k]), int(path[2 * k + 1])) for k in range(npos.value)]


def cigar_verify(cigar: str, a: bytes, b: bytes) -> int:
    """Cost of a valid unit-cost CIGAR for (a,b), or -1 if invalid."""
    return lib().pa_or_cigar_verify(cigar.encode(), _buf(a), len(a), _buf(b), len(b))


# ---- host block engine over the CPU kernels (oracle/engine_cpu.cpp) ---------------------------------
class BlockParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sparse", "simd", "no_ilp", "incremental_doubling", "dt_trace", "max_g", "fr_drop")]


class AstarPa2ParamsC(C.Structure):
    _fields_ = [("domain", C.c_int32), ("heuristic", C.c_int32), ("heuristic_k", C.c_int32), ("heuristic_p", C.c_int32), ("doubling", C.c_int32), ("doubling_start", C.c_int32),
                ("factor", C.c_float), ("delta", C.c_float), ("block_width", C.c_int32), ("front", BlockParamsC),
                ("sparse_h", C.c_int32), ("prune", C.c_int32)]


class AstarPa2StatsC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes",
                                           "dt_trace_tries", "dt_trace_success", "dt_trace_fallback", "fill_tries",
                                           "fill_success", "fill_fallback", "f_max_tries", "sanity_violations")] + \
               [(n, C.c_double) for n in ("t_compute", "t_dt", "t_fill", "t_precomp", "t_j_range", "t_fixed_j_range",
                                          "t_pruning", "t_contours_update")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


DOMAIN = {"full": 0, "gap_start": 1, "gap_gap": 2, "astar": 3}
HEURISTIC = {"none": 0, "gap": 1, "sh": 2, "gcsh": 3}
DOUBLING = {"none": 0, "band": 1, "linear": 2}
START = {"zero": 0, "gap": 1, "h0": 2}


def make_params(domain="astar", heuristic="gap", k=15, p=0, doubling="band", start="h0", factor=2.0, delta=1.0, block_width=256,
                sparse=True, simd=True, no_ilp=False, incremental_doubling=True, dt_trace=False, max_g=40, fr_drop=20,
                sparse_h=False, prune=False) -> AstarPa2ParamsC:
    """Defaults follow BlockParams::default() (blocks.rs:62-74)."""
    return AstarPa2ParamsC(DOMAIN[domain], HEURISTIC[heuristic], k, p, DOUBLING[doubling], START[start], factor, delta,
                           block_width, BlockParamsC(int(sparse), int(simd), int(no_ilp), int(incremental_doubling),
                                                     int(dt_trace), max_g, fr_drop), int(sparse_h), int(prune))


def params_nw():  # params.rs:46-68
    return make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=False,
                       incremental_doubling=False, dt_trace=False)


def params_simple():  # params.rs:70-96
    return make_params(domain="astar", heuristic="gap", doubling="band", start="h0", factor=2.0, block_width=256,
                       sparse=True, incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)


class EnginePanic(RuntimeError):
    pass


def params_full():  # params.rs:98-128
    return make_params(domain="astar", heuristic="gcsh", k=12, p=14, doubling="band", start="h0", factor=2.0, block_width=256,
                       sparse=True, incremental_doubling=True, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True, prune=True)


_elib = None


def engine_lib() -> C.CDLL:
    global _elib
    if _elib is None:
        build()
        L = C.CDLL(str(_DIR / "_build" / "libpa_engine_cpu.so"))
        L.pa_cpu_align.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(AstarPa2ParamsC), C.c_int,
                                   C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(AstarPa2StatsC)]
        L.pa_cpu_align.restype = C.c_int
        L.pa_cpu_free.argtypes = [C.c_void_p]
        _elib = L
    return _elib


def cpu_align(a: bytes, b: bytes, params: AstarPa2ParamsC, trace: bool = True, self_check: bool = False):
    """Host block engine over the CPU oracle kernels -> (cost, cigar or None, stats dict)."""
    cost = C.c_int32(0)
    cig = C.c_void_p(None)
    stats = AstarPa2StatsC()
    rc = engine_lib().pa_cpu_align(_buf(a), len(a), _buf(b), len(b), C.byref(params), int(trace), int(self_check),
                                   C.byref(cost), C.byref(cig), C.byref(stats))
    if rc == -1:
        raise ValueError("sequence contains a character outside ACGT")
    if rc == -5:
        raise EnginePanic("engine panic (see stderr)")
    if rc != 0:
        raise RuntimeError(f"pa_cpu_align rc={rc}")
    s = None
    if cig.value:
        s = C.string_at(cig.value).decode()
        engine_lib().pa_cpu_free(cig)
    return cost.value, s, stats.as_dict()


def cpu_many(pairs, params: AstarPa2ParamsC | None, nthreads: int) -> list[int]:
    """Costs of `pairs` computed on `nthreads` host threads inside the oracle library (one atomic work counter, no Python between
    the calls): params None = full DP cost-only (the AVX2 strip port), else the CPU-kernel engine with traceback."""
    L = engine_lib()
    L.pa_cpu_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pa_cpu_many.restype = C.c_int
    n = len(pairs)
    ap = (C.c_void_p * n)(*[C.cast(C.c_char_p(a), C.c_void_p) for a, _ in pairs])
    bp = (C.c_void_p * n)(*[C.cast(C.c_char_p(b), C.c_void_p) for _, b in pairs])
    al = (C.c_size_t * n)(*[len(a) for a, _ in pairs])
    bl = (C.c_size_t * n)(*[len(b) for _, b in pairs])
    out = np.zeros(n, np.int32)
    rc = L.pa_cpu_many(ap, al, bp, bl, n, C.byref(params) if params is not None else None, 0 if params is None else 1, nthreads, _p(out))
    if rc != 0:
        raise RuntimeError(f"pa_cpu_many rc={rc}")
    return out.tolist()


def sh_h(a: bytes, b: bytes, k: int) -> list[int]:
    """SH heuristic h(i) for i = 0..len(a) as the engine computes it (test hook)."""
    L = engine_lib()
    L.pa_cpu_sh_h.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    out = np.zeros(len(a) + 1, np.int32)
    L.pa_cpu_sh_h(_buf(a), len(a), _buf(b), len(b), k, _p(out))
    return out.tolist()


def gcsh_probe(a: bytes, b: bytes, k: int, p: int, queries):
    """GCSH h at `queries` [(i, j), ..] and the list of kept match starts (test hook)."""
    L = engine_lib()
    L.pa_cpu_gcsh_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    L.pa_cpu_gcsh_probe.restype = C.c_int
    qi = np.array([q[0] for q in queries], np.int32)
    qj = np.array([q[1] for q in queries], np.int32)
    out = np.zeros(len(queries), np.int32)
    cap = 1 << 20
    mo = np.zeros((cap, 2), np.int32)
    cnt = L.pa_cpu_gcsh_probe(_buf(a), len(a), _buf(b), len(b), k, p, _p(qi), _p(qj), len(queries), _p(out), _p(mo), cap)
    return out.tolist(), [tuple(x) for x in mo[:cnt].tolist()]


_slib = None


def sweep_emu_lib() -> C.CDLL:
    global _slib
    if _slib is None:
        build()
        L = C.CDLL(str(_DIR / "_build" / "libpa_sweep_emu.so"))
        L.pa_sweep_emu_align.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(AstarPa2ParamsC), C.c_int, C.c_int,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(AstarPa2StatsC), C.c_void_p]
        L.pa_sweep_emu_align.restype = C.c_int
        L.pa_sweep_jr_end.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p] + [C.c_int32] * 6
        L.pa_sweep_jr_end.restype = C.c_int32
        _slib = L
    return _slib


def sweep_emu_align(a: bytes, b: bytes, params: AstarPa2ParamsC, trace: bool = True, nwaves: int = 4):
    """The product's device-side sweep (sweep_wave.hpp) run on emulated wavefronts (host threads).
    -> (rc, cost, cigar, stats, info); rc 0 = ran, 1 = parameters not supported, 2 = the sweep asked for the host fallback."""
    cost = C.c_int32(0)
    cig = C.c_void_p(None)
    stats = AstarPa2StatsC()
    info = np.zeros(8, np.int32)
    rc = sweep_emu_lib().pa_sweep_emu_align(_buf(a), len(a), _buf(b), len(b), C.byref(params), int(trace), nwaves,
                                            C.byref(cost), C.byref(cig), C.byref(stats), _p(info))
    s = None
    if cig.value:
        s = C.string_at(cig.value).decode()
        engine_lib().pa_cpu_free(cig)
    return rc, cost.value, s, stats.as_dict(), info.tolist()


_alib = None


def apa2_emu_align(a: bytes, b: bytes, params: AstarPa2ParamsC):
    """The product's per-pair A*PA2 program (apa2_logic.hpp: what ONE wavefront runs per pair in the batched mode) over the CPU
    oracle kernels.  -> (rc, cost, cigar, stats, info); rc 0 = ran, 1 = not supported, 2 = handed back (info[0] = status);
    info[1] = fixed_j_range scans, info[2] = scans where the reference's jumping probes did not end on the first / last row."""
    global _alib
    if _alib is None:
        build()
        L = C.CDLL(str(_DIR / "_build" / "libpa_apa2_emu.so"))
        L.pa_apa2_emu_align.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(AstarPa2ParamsC), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_void_p), C.POINTER(AstarPa2StatsC), C.c_void_p]
        L.pa_apa2_emu_align.restype = C.c_int
        _alib = L
    cost = C.c_int32(0)
    cig = C.c_void_p(None)
    stats = AstarPa2StatsC()
    info = np.zeros(8, np.int32)
    rc = _alib.pa_apa2_emu_align(_buf(a), len(a), _buf(b), len(b), C.byref(params), C.byref(cost), C.byref(cig), C.byref(stats), _p(info))
    s = None
    if cig.value:
        s = C.string_at(cig.value).decode()
        engine_lib().pa_cpu_free(cig)
    return rc, cost.value, s, stats.as_dict(), info.tolist()


_flib = None


def apa2_full_emu_align(a: bytes, b: bytes, params: AstarPa2ParamsC):
    """The flat per-pair program of the whole A*PA2 family (csrc/apa2_full_logic.hpp: any heuristic, incremental doubling, pruning --
    groundwork for a batched `full`, not yet run by the library) over the CPU oracle kernels.  -> (rc, cost, cigar, stats, info);
    rc 0 = ran, 1 = outside the program, 2 = gave up (info[0]); info[1] = h calls, info[2] = prune_block calls, info[3] = 3-range
    splits, info[4] = plain initialisations, info[5] = h calls where the device form of GCSH (gcsh_dev.hpp) disagreed with gcsh.hpp,
    info[6] = builds of the flat arrays."""
    global _flib
    if _flib is None:
        build()
        L = C.CDLL(str(_DIR / "_build" / "libpa_apa2_full_emu.so"))
        L.pa_apa2_full_emu_align.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(AstarPa2ParamsC), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_void_p), C.POINTER(AstarPa2StatsC), C.c_void_p]
        L.pa_apa2_full_emu_align.restype = C.c_int
        _flib = L
    cost = C.c_int32(0)
    cig = C.c_void_p(None)
    stats = AstarPa2StatsC()
    info = np.zeros(8, np.int32)
    rc = _flib.pa_apa2_full_emu_align(_buf(a), len(a), _buf(b), len(b), C.byref(params), C.byref(cost), C.byref(cig), C.byref(stats), _p(info))
    s = None
    if cig.value:
        s = C.string_at(cig.value).decode()
        engine_lib().pa_cpu_free(cig)
    return rc, cost.value, s, stats.as_dict(), info.tolist()


def cpu_align_blocks(a: bytes, b: bytes, params: AstarPa2ParamsC):
    """The blocks of the engine's last completed pass (traceback mode, before the trace): (cost, f_max, [dict per block]).
    v of a block = list of (p, m) words covering rows [js, je)."""
    L = engine_lib()
    L.pa_cpu_align_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(AstarPa2ParamsC), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.pa_cpu_align_blocks.restype = C.c_int
    nb = len(a) // max(1, params.block_width) + 4
    rec = np.zeros((nb, 12), np.int32)
    v = np.zeros(2 * nb * (len(b) // 64 + 2), np.uint64)
    cost, f_max = C.c_int32(0), C.c_int32(0)
    n = L.pa_cpu_align_blocks(_buf(a), len(a), _buf(b), len(b), C.byref(params), C.byref(cost), C.byref(f_max), _p(rec), rec.size, _p(v), v.size)
    if n < 0:
        raise RuntimeError(f"pa_cpu_align_blocks rc={n}")
    keys = ["i0", "i1", "ojs", "oje", "js", "je", "fs", "fe", "top_val", "bot_val"]
    out = []
    for k in range(n):
        d = {key: int(rec[k, i]) for i, key in enumerate(keys)}
        off, w = int(rec[k, 10]), int(rec[k, 11])
        d["v"] = [(int(v[2 * (off + j)]), int(v[2 * (off + j) + 1])) for j in range(w)]
        out.append(d)
    return cost.value, f_max.value, out


_rlib = None


def rdv_emu_run(pairs, groups: int = 2, patience_us: float = 200.0):
    """oracle/rdv_emu.cpp: the `simple` band search of every pair on groups x 4 host threads with the product's rendezvous of half-wave
    blocks (csrc/rdv_logic.hpp) between the threads of a group; patience_us < 0: no rendezvous.  -> (rows, counters): per pair
    (status, cost, f_max_tries, num_blocks, computed_lanes, unique_lanes), counters = dict(fused, served, alone, withdrawn)."""
    global _rlib
    build()
    if _rlib is None:
        L = C.CDLL(str(_DIR / "_build" / "libpa_rdv_emu.so"))
        L.pa_rdv_emu_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.pa_rdv_emu_run.restype = C.c_int
        _rlib = L
    n = len(pairs)
    ab = [C.create_string_buffer(bytes(a), max(len(a), 1)) for a, _ in pairs]
    bb = [C.create_string_buffer(bytes(b), max(len(b), 1)) for _, b in pairs]
    ap = (C.c_void_p * n)(*[C.addressof(x) for x in ab])
    bp = (C.c_void_p * n)(*[C.addressof(x) for x in bb])
    al = (C.c_size_t * n)(*[len(a) for a, _ in pairs])
    bl = (C.c_size_t * n)(*[len(b) for _, b in pairs])
    out = (C.c_int64 * (8 * n))()
    cnt = (C.c_uint64 * 4)()
    rc = _rlib.pa_rdv_emu_run(ap, al, bp, bl, n, groups, float(patience_us), out, cnt)
    if rc != 0:
        raise RuntimeError(f"pa_rdv_emu_run rc={rc}")
    rows = [tuple(int(out[8 * i + k]) for k in range(6)) for i in range(n)]
    return rows, dict(zip(("fused", "served", "alone", "withdrawn"), (int(x) for x in cnt)))


_clib = None


def combine_emu_run(threads: int, calls: int, batch_us: int = 300, fail_every: int = 0) -> dict:
    """oracle/combine_emu.cpp: the call combiner's gathering protocol (csrc/combine_logic.hpp) on `threads` host threads with a stand-in
    batch -> wrong results, batches, the largest group, batches side by side, requests whose batch failed."""
    global _clib
    build()
    if _clib is None:
        L = C.CDLL(str(_DIR / "_build" / "libpa_combine_emu.so"))
        L.pa_combine_emu_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pa_combine_emu_run.restype = C.c_int
        _clib = L
    out = (C.c_int64 * 7)()
    _clib.pa_combine_emu_run(threads, calls, batch_us, fail_every, out)
    return dict(zip(("wrong", "batches", "largest_group", "side_by_side", "failed", "empty_groups", "late_returns"), (int(x) for x in out)))
