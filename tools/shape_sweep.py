"""Batch-shape sweep (not the contract bench): kernel time of the full-DP batch for every forced (mode, k) next to the
library's own choice, for mid-size batches of 100 kbp pairs.  Usage: python tools/shape_sweep.py [pairs ...]"""
import os
import sys

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
sizes = [int(x) for x in sys.argv[1:]] or [41, 146, 256, 384, 512, 768, 1024, 1536]
n = int(os.environ.get("PA_SWEEP_N", "100000"))
base = [generate_pair(n, 0.05, seed=s + 1) for s in range(16)]
for pairs in sizes:
    ps = [base[i % len(base)] for i in range(pairs)]
    row = []
    # (one wavefront per pair only makes sense with about a pair per SIMD: skip it for small batches, it takes minutes)
    for mode, k in [(None, None)] + [(m, k) for m in ("chain", "seq") for k in (2, 4, 8) if m == "chain" or pairs * (n / 100000) >= 200]:
        for v in ("PA_BATCH_MODE", "PA_STRIP_K"):
            os.environ.pop(v, None)
        if mode:
            os.environ["PA_BATCH_MODE"] = mode
            os.environ["PA_STRIP_K"] = str(k)
        b = pa.Batch(ps)
        st = b.stats()
        best = min(b.run()[1] for _ in range(3))
        sh = b.shape()
        row.append(f"{'auto' if not mode else mode}{'' if not mode else k}[{'seq' if sh['sequential'] else 'chain'}{sh['k']}]={st['cells'] / best / 1e9:.0f}")
        b.close()
    print(f"pairs={pairs}: " + " ".join(row), flush=True)
