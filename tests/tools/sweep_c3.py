"""GPU: C3 (100 kbp, 5 %) through the device-side sweep: per-pass timings (PA_SWEEP_TIMING=1), equality with the CPU engine."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import astar_pairwise_aligner_amd as pa
import oracle
from tests.util_seq import gen_pair
from tests.test_sweep_emu import KEYS
from tests.test_gpu_engine import gpu_params

pa.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
a, b = gen_pair(n, 0.05, 1)
al = gpu_params(pa, oracle.params_simple()).make_aligner(True)
for rep in range(3):
    t = time.time()
    cost, cigar, stats = al.align_with_stats(a, b)
    print(f"simple trace: cost {cost} {1e3 * (time.time() - t):.2f} ms  t_compute {stats['t_compute'] * 1e3:.2f} t_dt {stats['t_dt'] * 1e3:.2f} t_fill {stats['t_fill'] * 1e3:.2f}", flush=True)
al0 = gpu_params(pa, oracle.params_simple()).make_aligner(False)
for rep in range(2):
    t = time.time()
    c0, _ = al0.align(a, b)
    print(f"simple cost-only: cost {c0} {1e3 * (time.time() - t):.2f} ms", flush=True)
if n <= 200_000:
    want = oracle.cpu_align(a, b, oracle.params_simple(), trace=True)
    print("equal to CPU engine:", (cost, cigar) == (want[0], want[1]), all(stats[k] == want[2][k] for k in KEYS))
