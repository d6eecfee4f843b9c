"""GPU: the device-side sweep (pa_align with the `simple` family) against the CPU engine; timings per pass with PA_SWEEP_TIMING=1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import astar_pairwise_aligner_amd as pa
import oracle
from tests.util_seq import gen_pair
from tests.test_sweep_emu import variants, KEYS
from tests.test_gpu_engine import gpu_params

pa.require_gpu()
vs = variants(oracle)
bad = 0
for name in ("simple", "dijkstra", "sh12", "gap_nosparseh", "linear"):
    oc = vs[name]
    for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (8191, 0.05, 8), (10000, 0.15, 4), (30000, 0.2, 6), (4097, 1.0, 3)]:
        a, b = gen_pair(n, e, seed)
        want = oracle.cpu_align(a, b, oc, trace=True)
        t = time.time()
        cost, cigar, stats = gpu_params(pa, oc).make_aligner(True).align_with_stats(a, b)
        dt = time.time() - t
        ok = cost == want[0] and cigar == want[1] and all(stats[k] == want[2][k] for k in KEYS)
        bad += not ok
        print(f"{name} n={n} e={e}: cost {cost}/{want[0]} cigar_eq={cigar == want[1]} ok={ok} {dt * 1e3:.2f} ms", flush=True)
        if not ok:
            print({k: (stats[k], want[2][k]) for k in KEYS if stats[k] != want[2][k]})
print("bad", bad)
# C3: 100 kbp, 5 %
a, b = gen_pair(100_000, 0.05, 1)
prm = pa.AstarPa2Params.simple() if hasattr(pa.AstarPa2Params, "simple") else gpu_params(pa, oracle.params_simple())
al = gpu_params(pa, oracle.params_simple()).make_aligner(True)
for rep in range(3):
    t = time.time()
    cost, cigar, stats = al.align_with_stats(a, b)
    print(f"C3 simple trace: cost {cost} {1e3 * (time.time() - t):.2f} ms  t_compute {stats['t_compute'] * 1e3:.2f} t_dt {stats['t_dt'] * 1e3:.2f} t_fill {stats['t_fill'] * 1e3:.2f}", flush=True)
al0 = gpu_params(pa, oracle.params_simple()).make_aligner(False)
for rep in range(3):
    t = time.time()
    cost, _ = al0.align(a, b)
    print(f"C3 simple cost-only: cost {cost} {1e3 * (time.time() - t):.2f} ms", flush=True)
want = oracle.cpu_align(a, b, oracle.params_simple(), trace=True)
print("C3 equal to CPU engine:", (cost, cigar) == (want[0], want[1]), all(stats[k] == want[2][k] for k in KEYS))
