"""Host-side mirror of the reference's aligner surface for the hot path.

Reference (Rust):  AstarPa2Params::{nw,simple,full}().make_aligner(trace) -> Box<dyn AstarPa2StatsAligner>
                   Aligner::align(a, b) -> (Cost, Option<Cigar>)               astarpa2/src/lib.rs:200-214
                   astarpa2::astarpa2_{nw,simple,full}(a, b) -> (Cost, Cigar)   astarpa2/src/lib.rs:38-53
Every DP rectangle runs in the HIP library; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

from . import capi

DOMAIN = {"full": 0, "gap_start": 1, "gap_gap": 2, "astar": 3}
HEURISTIC = {"none": 0, "gap": 1, "sh": 2, "gcsh": 3}
DOUBLING = {"none": 0, "band": 1, "linear": 2}
START = {"zero": 0, "gap": 1, "h0": 2}


class _BlockParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sparse", "simd", "no_ilp", "incremental_doubling", "dt_trace", "max_g", "fr_drop")]


class _ParamsC(C.Structure):
    _fields_ = [("domain", C.c_int32), ("heuristic", C.c_int32), ("heuristic_k", C.c_int32), ("heuristic_p", C.c_int32), ("doubling", C.c_int32), ("doubling_start", C.c_int32),
                ("factor", C.c_float), ("delta", C.c_float), ("block_width", C.c_int32), ("front", _BlockParamsC),
                ("sparse_h", C.c_int32), ("prune", C.c_int32)]


class _StatsC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes",
                                           "dt_trace_tries", "dt_trace_success", "dt_trace_fallback", "fill_tries",
                                           "fill_success", "fill_fallback", "f_max_tries", "sanity_violations")] + \
               [(n, C.c_double) for n in ("t_compute", "t_dt", "t_fill", "t_precomp", "t_j_range", "t_fixed_j_range",
                                          "t_pruning", "t_contours_update")]


@dataclass
class BlockParams:  # blocks.rs:31-74
    sparse: bool = True
    simd: bool = True
    no_ilp: bool = False
    incremental_doubling: bool = True
    dt_trace: bool = False
    max_g: int = 40
    fr_drop: int = 20


@dataclass
class AstarPa2Params:  # params.rs:8-42
    name: str = ""
    domain: str = "astar"
    heuristic: str = "gap"
    k: int = 15  # HeuristicParams.k (seed length of SH / GCSH)
    p: int = 0  # HeuristicParams.p (local-pruning look-ahead of GCSH)
    doubling: str = "band"
    doubling_start: str = "h0"
    factor: float = 2.0
    delta: float = 1.0
    block_width: int = 256
    front: BlockParams = field(default_factory=BlockParams)
    sparse_h: bool = False
    prune: bool = False

    @staticmethod
    def nw() -> "AstarPa2Params":  # params.rs:46-68
        return AstarPa2Params(name="nw", domain="full", heuristic="none", doubling="none", block_width=256,
                              front=BlockParams(sparse=False, incremental_doubling=False, dt_trace=False))

    @staticmethod
    def simple() -> "AstarPa2Params":  # params.rs:70-96
        return AstarPa2Params(name="simple", domain="astar", heuristic="gap", doubling="band", doubling_start="h0",
                              factor=2.0, block_width=256,
                              front=BlockParams(sparse=True, incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10),
                              sparse_h=True, prune=False)

    @staticmethod
    def full() -> "AstarPa2Params":  # params.rs:98-128
        return AstarPa2Params(name="full", domain="astar", heuristic="gcsh", k=12, p=14, doubling="band", doubling_start="h0",
                              factor=2.0, block_width=256,
                              front=BlockParams(sparse=True, incremental_doubling=True, dt_trace=True, max_g=40, fr_drop=10),
                              sparse_h=True, prune=True)

    def _to_c(self) -> _ParamsC:
        f = self.front
        return _ParamsC(DOMAIN[self.domain], HEURISTIC[self.heuristic], self.k, self.p, DOUBLING[self.doubling], START[self.doubling_start],
                        self.factor, self.delta, self.block_width,
                        _BlockParamsC(int(f.sparse), int(f.simd), int(f.no_ilp), int(f.incremental_doubling),
                                      int(f.dt_trace), f.max_g, f.fr_drop), int(self.sparse_h), int(self.prune))

    def make_aligner(self, trace: bool) -> "AstarPa2":  # params.rs:132
        return AstarPa2(self, trace)


class AstarPa2:
    """Aligner + AstarPa2StatsAligner (lib.rs:200-214)."""

    def __init__(self, params: AstarPa2Params, trace: bool):
        self.params = params
        self.trace = trace

    def align_with_stats(self, a: bytes, b: bytes):
        L = capi.load()
        L.pa_align.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_ParamsC), C.c_int,
                               C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(_StatsC)]
        L.pa_align.restype = C.c_int
        cost, cig, stats = C.c_int32(0), C.c_void_p(None), _StatsC()
        pc = self.params._to_c()
        rc = L.pa_align(C.cast(C.c_char_p(a), C.c_void_p), len(a), C.cast(C.c_char_p(b), C.c_void_p), len(b), C.byref(pc),
                        int(self.trace), C.byref(cost), C.byref(cig), C.byref(stats))
        if rc == -1:
            raise ValueError("sequence contains a character outside ACGT")
        if rc != 0:
            raise capi.PaError(f"pa_align rc={rc}: {capi.last_error()}")
        cigar = None
        if cig.value:
            cigar = C.string_at(cig.value).decode()
            L.astarpa_free_cigar(cig)
        return cost.value, cigar, {n: getattr(stats, n) for n, _ in _StatsC._fields_}

    def align(self, a: bytes, b: bytes):
        cost, cigar, _ = self.align_with_stats(a, b)
        return cost, cigar

    def cost(self, a: bytes, b: bytes) -> int:  # lib.rs:177-179
        return AstarPa2(self.params, False).align(a, b)[0]


def astarpa2_nw(a: bytes, b: bytes):  # lib.rs:38-41
    return AstarPa2Params.nw().make_aligner(True).align(a, b)


def astarpa2_simple(a: bytes, b: bytes):  # lib.rs:43-47
    return AstarPa2Params.simple().make_aligner(True).align(a, b)


def astarpa2_full(a: bytes, b: bytes):  # lib.rs:49-53
    return AstarPa2Params.full().make_aligner(True).align(a, b)


def c_abi_align(symbol: str, a: bytes, b: bytes, *extra):
    """Call one of the reference's C symbols (astarpa2_simple / astarpa2_full / astarpa / astarpa_gcsh)."""
    L = capi.load()
    f = getattr(L, symbol)
    ptr, ln = C.c_void_p(None), C.c_size_t(0)
    cost = f(C.cast(C.c_char_p(a), C.c_void_p), len(a), C.cast(C.c_char_p(b), C.c_void_p), len(b), *extra, C.byref(ptr), C.byref(ln))
    s = C.string_at(ptr.value).decode()
    assert len(s) == ln.value
    L.astarpa_free_cigar(ptr)
    return int(cost), s
