"""Batched alignment WITH traceback (pa_batch_align): forward / traceback kernel times and end-to-end pairs per second.
Usage: python tools/align_bench.py NxPAIRS[@div] ...   (div 'mix' = 1/5/10/15 % round robin, the C4 recipe)"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
for arg in sys.argv[1:]:
    shape, _, div = arg.partition("@")
    n, pairs = (int(x) for x in shape.split("x"))
    divs = (0.01, 0.05, 0.10, 0.15) if div in ("", "mix") else (float(div),)
    base = [generate_pair(n, divs[s % len(divs)], seed=s + 1) for s in range(min(pairs, 64))]
    ps = [base[i % len(base)] for i in range(pairs)]
    b = pa.Batch(ps, trace=True)
    st = b.stats()
    b.align()
    best = (1e9, 0, 0)
    for _ in range(3):
        t = time.perf_counter()
        costs, cigars, fwd, tr = b.align()
        dt = time.perf_counter() - t
        if dt < best[0]:
            best = (dt, fwd, tr)
    dt, fwd, tr = best
    print(f"n={n} pairs={pairs} div={div or 'mix'} shape={b.shape()['kernel']} forward_ms={fwd:.2f} trace_ms={tr:.2f} wall_ms={dt*1e3:.1f} "
          f"pairs/s={pairs/dt:.0f} GCUPS(wall)={st['cells']/dt/1e9:.0f} fallbacks={b.trace_fallbacks()} cost0={costs[0]} cigar0={cigars[0][:40]}", flush=True)
    b.close()
