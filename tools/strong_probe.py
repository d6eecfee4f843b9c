"""Why do repetitions of one pa_batch_align over 100 000 C4 pairs differ (round 6: 425 ms .. 3.5 s)?  Times creation, alignment, closing
and the Python-side result handling separately, five times.  python tools/strong_probe.py [pairs]"""
import gc
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd import capi  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

pa.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
divs = (0.01, 0.05, 0.10, 0.15)
base = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(10_000)]
pairs = (base * ((n + 9999) // 10_000))[:n]
for rep in range(6):
    t0 = time.perf_counter()
    b = capi.Batch(list(pairs), trace=True)
    t1 = time.perf_counter()
    costs, cigars, _, _ = b.align()
    t2 = time.perf_counter()
    b.close()
    t3 = time.perf_counter()
    res = [(int(c), g) for c, g in zip(costs, cigars)]
    t4 = time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.1f} align {1e3*(t2-t1):.1f} close {1e3*(t3-t2):.1f} tuples {1e3*(t4-t3):.1f} ms  gc {gc.get_count()}", flush=True)
    del res, costs, cigars
