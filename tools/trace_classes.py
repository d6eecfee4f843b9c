"""Traceback time of the batched A*PA2 by divergence class: C4's four classes alone (2500 pairs each: fewer wavefronts than the GPU holds,
so the kernel lasts as long as its slowest pair) and mixed.  python tools/trace_classes.py [simple|full]"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "simple"
prm = pa.AstarPa2Params.full() if preset == "full" else pa.AstarPa2Params.simple()
divs = (0.01, 0.05, 0.10, 0.15)


def run(pairs, label):
    bt = pa.Batch(pairs, params=prm)
    bt.align()
    best = None
    for _ in range(3):
        t = time.perf_counter()
        _, _, f_ms, t_ms = bt.align()
        dt = (time.perf_counter() - t) * 1e3
        if best is None or dt < best[0]:
            best = (dt, f_ms, t_ms, bt.last_c_abi_ms)
    print(f"{label}: {len(pairs)} pairs  align {best[0]:.2f} ms (C ABI {best[3]:.2f})  forward {best[1]:.2f}  trace {best[2]:.2f}", flush=True)
    bt.close()


for d in divs:
    for cnt in (64, 2500):
        run([generate_pair(10_000, d, seed=2_000_000 + 4 * i + divs.index(d)) for i in range(cnt)], f"{int(d * 100)} % only")
run([generate_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(10_000)], "C4 mixed")
