"""How much does the strip-to-strip hand-off cost?  Same wave count and columns, different chain lengths:
rows = 2048*S per pair (S chained strips), `pairs` chosen so that pairs*S ~ 7154 waves (7 per SIMD)."""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import random_sequence

pa.require_gpu()
n = 100_000
for S in (1, 2, 7, 49):
    pairs = 7154 // S
    base = [(random_sequence(n, seed=s + 1), random_sequence(2048 * S, seed=1000 + s)) for s in range(min(pairs, 16))]
    ps = [base[i % len(base)] for i in range(pairs)]
    b = pa.Batch(ps)
    st = b.stats()
    b.run()
    best = 1e9
    for _ in range(3):
        costs, ms = b.run()
        best = min(best, ms)
    steps = st["strips"] * (n + 64)
    print(f"chain S={S:3d} pairs={pairs:5d} strips={int(st['strips'])} kernel_ms={best:.3f} GCUPS={st['cells']/best/1e6:.0f} "
          f"ns per strip-step per SIMD={best*1e6/ (steps/1024):.1f}", flush=True)
    b.close()
