"""Quick timing probe of the batched full-DP path (not the contract bench; see bench.py)."""
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
    b = pa.Batch(ps)
    st = b.stats()
    costs, ms = b.run()
    best, bestk = 1e9, 1e9
    for _ in range(3):
        t = time.time()
        costs, ms = b.run()
        best = min(best, time.time() - t)
        bestk = min(bestk, ms)
    print(f"n={n} pairs={pairs} strips={int(st['strips'])} kernel_ms={bestk:.3f} wall_ms={best*1e3:.3f} "
          f"GCUPS(kernel)={st['cells']/bestk/1e6:.1f} GCUPS(wall)={st['cells']/best/1e9:.1f} cost0={costs[0]}", flush=True)
    b.close()
