"""Where a small batch's time goes (what the call combiner behind pa_align creates): PA_ALIGN_PROFILE=1 python tools/combine_probe.py [pairs] [reps]"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
divs = (0.01, 0.05, 0.10, 0.15)
pairs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(n)]
prm = pa.AstarPa2Params.simple()
for r in range(reps):
    t0 = time.perf_counter()
    bt = pa.Batch(pairs, params=prm)
    t1 = time.perf_counter()
    _, _, f_ms, t_ms = bt.align_c_strings()
    t2 = time.perf_counter()
    bt.pair_stats()
    bt.close()
    t3 = time.perf_counter()
    print(f"rep {r}: {n} pairs  create {(t1-t0)*1e3:.2f} ms  align {(t2-t1)*1e3:.2f} ms (forward kernel {f_ms:.2f}, trace kernel {t_ms:.2f})  stats+destroy {(t3-t2)*1e3:.2f} ms  cache {pa.capi.alloc_cache_stats()}", flush=True)
