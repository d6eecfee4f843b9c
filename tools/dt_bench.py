"""Batched traceback with and without the device-side DT-trace (not the contract bench): C4 (10 000 x 10 kbp) and one 100 kbp pair."""
import sys

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
import oracle
from astar_pairwise_aligner_amd.generate import generate_pair
from tests.test_gpu_batch_align import dt_params
from tests.test_gpu_engine import gpu_params

pa.require_gpu()
prm = dt_params(oracle)
divs = (0.01, 0.05, 0.10, 0.15)
pairs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(10_000)]
b0 = pa.Batch(pairs, trace=True)
b1 = pa.Batch(pairs, trace=True, trace_params=gpu_params(pa, prm))
res = {}
for name, b in (("refill", b0), ("dt", b1)):
    best = 1e9
    for _ in range(3):
        c, g, f, t = b.align()
        best = min(best, t)
    res[name] = g
    print(name, "forward_ms", round(f, 2), "trace_ms", round(best, 2), "fallbacks", b.trace_fallbacks(), flush=True)
print("cigars differ in", sum(x != y for x, y in zip(res["refill"], res["dt"])), "of", len(pairs))
for i in range(0, 10_000, 250):
    w = oracle.cpu_align(*pairs[i], prm)
    assert (int(c[i]), res["dt"][i]) == (w[0], w[1]), i
print("sample equals the CPU-kernel engine")
one = [generate_pair(100_000, 0.05, seed=1)]
for tp in (None, gpu_params(pa, prm)):
    bb = pa.Batch(one, trace=True, trace_params=tp)
    print("100 kbp pair trace_ms", "dt" if tp else "refill", round(min(bb.align()[3] for _ in range(3)), 2))
