"""Where the creation of a batched A*PA2 batch goes (PA_ALIGN_PROFILE marks on stderr): python tools/create_marks.py simple|full [pairs] [n]"""
import os
import sys
import time

os.environ["PA_ALIGN_PROFILE"] = "1"
sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

pa.require_gpu()
preset = sys.argv[1] if len(sys.argv) > 1 else "simple"
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
divs = (0.01, 0.05, 0.10, 0.15) if n <= 10_000 else (0.05,)
pairs = [generate_pair(n, divs[i % len(divs)], seed=1_000_000 + i) for i in range(npairs)]
mk = pa.AstarPa2Params.simple if preset == "simple" else pa.AstarPa2Params.full
for rep in range(3):
    print(f"--- {preset} {npairs} x {n}: creation {rep}", file=sys.stderr, flush=True)
    t = time.perf_counter()
    b = pa.Batch(pairs, params=mk())
    print(f"    Batch(...) {1e3 * (time.perf_counter() - t):.1f} ms", file=sys.stderr, flush=True)
    b.align()
    b.close()
