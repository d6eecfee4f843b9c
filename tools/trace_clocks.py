"""Where the batched traceback's wavefronts spend their time (library built with PA_HIPCC_EXTRA=-DPA_TRACE_CLOCKS, which must also be
set when this runs so that the library is not rebuilt): python tools/trace_clocks.py"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd import capi  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

L = capi.load()
f = L.pa_debug_trace_clocks
f.argtypes = [C.POINTER(C.c_double)]
f.restype = C.c_int
divs = (0.01, 0.05, 0.10, 0.15)


def clocks():
    v = (C.c_double * 10)()
    assert f(v) == 0
    return list(v)


def run(pairs, label, prm):
    bt = pa.Batch(pairs, params=prm)
    bt.align()
    clocks()
    _, _, f_ms, t_ms = bt.align()
    c = clocks()
    us = lambda t: t * 1e-2  # noqa: E731  (100 MHz)
    n = len(pairs)
    print(f"{label}: {n} pairs trace {t_ms:.2f} ms; per pair: wavefront {us(c[4])/n:.0f} us = DT ok {us(c[0])/n:.0f} + DT failed {us(c[1])/n:.0f} + re-fill {us(c[2])/n:.0f} "
          f"+ parent steps {us(c[3])/n:.0f} (+ rest); blocks tried {c[7]/n:.1f}, DT levels {c[5]/n:.0f}, parent steps {c[6]/n:.0f} "
          f"({us(c[3])/max(c[6],1):.2f} us each)", flush=True)
    bt.close()


prm = pa.AstarPa2Params.simple()
for d in divs:
    for cnt in (64, 2500):
        run([generate_pair(10_000, d, seed=2_000_000 + 4 * i + divs.index(d)) for i in range(cnt)], f"{int(d * 100)} % only", prm)
run([generate_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(10_000)], "C4 mixed", prm)
run([generate_pair(100_000, 0.05, seed=3_000_000 + i) for i in range(4096)], "4096 x 100 kbp", pa.AstarPa2Params.full())
