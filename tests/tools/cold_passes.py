"""Cold (host buffers -> results) times of the batched paths: creation + one pass, nothing resident beforehand."""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
pa.Batch([generate_pair(1000, 0.05, seed=1)], trace=True).align()
divs = (0.01, 0.05, 0.10, 0.15)
c4 = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(10_000)]
c2 = [generate_pair(100_000, 0.05, seed=i + 1) for i in range(2048)]
for name, pairs, kw in (("C4 10000 x 10 kbp, cost + CIGAR", c4, dict(trace=True)), ("C4 10000 x 10 kbp, cost only", c4, {}),
                        ("2048 x 100 kbp, cost + CIGAR", c2, dict(trace=True)), ("2048 x 100 kbp, banded 6 %", c2, dict(band=0.06)),
                        ("2048 x 100 kbp, cost only", c2, {})):
    for trial in range(2):
        t0 = time.perf_counter()
        b = pa.Batch(pairs, **kw)
        t1 = time.perf_counter()
        out = b.align() if kw.get("trace") else b.run()
        t2 = time.perf_counter()
        b.close()
    print(f"{name:36s} create {1e3*(t1-t0):7.1f} ms + pass {1e3*(t2-t1):7.1f} ms = {1e3*(t2-t0):7.1f} ms -> {len(pairs)/(t2-t0):9.0f} pairs/s cold", flush=True)
