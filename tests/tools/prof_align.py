import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd import capi
from astar_pairwise_aligner_amd.generate import generate_pair
pa.require_gpu()
divs = (0.01, 0.05, 0.10, 0.15)
base = [generate_pair(10000, divs[s % 4], seed=s + 1) for s in range(64)]
ps = [base[i % 64] for i in range(10000)]
b = pa.Batch(ps, trace=True)
b.align()
L = capi.load()
for _ in range(3):
    out = np.zeros(b.pairs, np.int32)
    cig = (C.c_void_p * b.pairs)()
    f, t = C.c_float(0), C.c_float(0)
    t0 = time.perf_counter()
    rc = L.pa_batch_align(b._h, capi._p(out), cig, C.byref(f), C.byref(t))
    t1 = time.perf_counter()
    strs = [C.string_at(cig[i]).decode() for i in range(b.pairs)]
    t2 = time.perf_counter()
    for i in range(b.pairs):
        L.astarpa_free_cigar(C.c_void_p(cig[i]))
    t3 = time.perf_counter()
    print(f"C call {1e3*(t1-t0):.1f} ms (kernels fwd {f.value:.1f} + trace {t.value:.1f}), python strings {1e3*(t2-t1):.1f} ms, frees {1e3*(t3-t2):.1f} ms")
