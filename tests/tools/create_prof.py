import sys, time, ctypes as C
sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd import capi
from astar_pairwise_aligner_amd.generate import generate_pair
pa.require_gpu()
pa.Batch([generate_pair(1000, 0.05, seed=1)]).run()
divs = (0.01, 0.05, 0.10, 0.15)
pairs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(10_000)]
L = capi.load()
for trial in range(3):
    t0 = time.perf_counter()
    n = len(pairs)
    ap = (C.c_void_p * n)(*[C.cast(C.c_char_p(a), C.c_void_p) for a, _ in pairs])
    bp = (C.c_void_p * n)(*[C.cast(C.c_char_p(b), C.c_void_p) for _, b in pairs])
    al = (C.c_size_t * n)(*[len(a) for a, _ in pairs])
    bl = (C.c_size_t * n)(*[len(b) for _, b in pairs])
    t1 = time.perf_counter()
    h = L.pa_batch_create(ap, al, bp, bl, n)
    t2 = time.perf_counter()
    L.pa_batch_destroy(h)
    t3 = time.perf_counter()
    h = L.pa_batch_create_trace(ap, al, bp, bl, n)
    t4 = time.perf_counter()
    L.pa_batch_destroy(h)
    print(f"python marshalling {1e3*(t1-t0):.1f} ms, pa_batch_create {1e3*(t2-t1):.1f} ms, destroy {1e3*(t3-t2):.1f} ms, pa_batch_create_trace {1e3*(t4-t3):.1f} ms")
