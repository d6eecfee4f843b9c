"""Quick timing probe of the batched full-DP path (not the contract bench; see bench.py)."""
import os
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
cases = [(100000, 1), (100000, 32), (100000, 64), (100000, 128), (100000, 256), (10000, 1000), (10000, 10000)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (n, pairs) in cases:
    base = [generate_pair(n, 0.05, seed=s + 1) for s in range(min(pairs, 32))]
    ps = [base[i % len(base)] for i in range(pairs)]
    band = float(os.environ["PA_BAND"]) if "PA_BAND" in os.environ else None  # banded DP with this divergence hint
    t_first = time.time()
    b = pa.Batch(ps, band=band)
    st = b.stats()
    costs, ms = b.run()
    first_ms = (time.time() - t_first) * 1e3  # create + first pass (band re-runs included)
    best, bestk = 1e9, 1e9
    for _ in range(3):
        t = time.time()
        costs, ms = b.run()
        best = min(best, time.time() - t)
        bestk = min(bestk, ms)
    print(f"n={n} pairs={pairs} strips={int(st['strips'])} kernel_ms={bestk:.3f} wall_ms={best*1e3:.3f} "
          f"GCUPS(kernel)={st['cells']/bestk/1e6:.1f} GCUPS(wall)={st['cells']/best/1e9:.1f} cost0={costs[0]}"
          + f" kernel={b.shape()['kernel']}" + (f" band={band} shape={b.shape()['kernel']} create+first_pass_ms={first_ms:.1f} pairs/s={pairs/best:.0f}" if band is not None else ""), flush=True)
    b.close()
