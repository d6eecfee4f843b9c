"""Round 6: the bit-sliced kernel (PA_SLICE=1, the library's rows per lane) against the strip kernels (PA_SLICE=0) and the default choice,
per batch shape: kernel ms of the best of three passes.  python tools/slice_sweep.py 100000x64 10000x4096 ..."""
import os
import sys

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for n, pairs in cases:
    base = [generate_pair(n, 0.05, seed=s + 1) for s in range(min(pairs, 64))]
    ps = [base[i % len(base)] for i in range(pairs)]
    row = []
    ref = None
    for mode in ("0", "1", None):
        if mode is None:
            os.environ.pop("PA_SLICE", None)
        else:
            os.environ["PA_SLICE"] = mode
        b = pa.Batch(ps)
        best = 1e9
        for _ in range(3):
            costs, ms = b.run()
            best = min(best, ms)
        if ref is None:
            ref = costs.copy()
        assert (costs == ref).all()
        sh = b.shape()
        row.append(f"{'default' if mode is None else 'PA_SLICE=' + mode}: {best:9.3f} ms {b.stats()['cells'] / best / 1e9:7.1f} TCUPS {sh['kernel']}")
        b.close()
    print(f"n={n} pairs={pairs} | " + " | ".join(row), flush=True)
