"""How much do SIMD sharing and the strip-to-strip hand-off cost?  Same columns, W wavefronts per SIMD, chains of S strips:
rows = 2048*K*S per pair, pairs = 1024*W/S.  Usage: PA_STRIP_K=4 python tools/chain_probe.py W:S [W:S ...]"""
import os
import sys

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import random_sequence

pa.require_gpu()
n = 100_000
K = int(os.environ.get("PA_STRIP_K", "1"))
for arg in sys.argv[1:]:
    W, S = (int(x) for x in arg.split(":"))
    pairs = 1024 * W // S
    base = [(random_sequence(n, seed=s + 1), random_sequence(2048 * K * S, seed=1000 + s)) for s in range(min(pairs, 8))]
    ps = [base[i % len(base)] for i in range(pairs)]
    b = pa.Batch(ps)
    st = b.stats()
    b.run()
    best = 1e9
    for _ in range(3):
        costs, ms = b.run()
        best = min(best, ms)
    print(f"K={K} W={W} chain S={S:3d} pairs={pairs:5d} strips={int(st['strips'])} kernel_ms={best:.3f} GCUPS={st['cells']/best/1e6:.0f} "
          f"ns per wave-step={best*1e6/(n+64):.1f}  per SIMD-share={best*1e6/(n+64)/max(1.0, st['strips']/1024):.1f}", flush=True)
    b.close()
