"""GPU: the device-side sweep at scale: 1 Mbp and 10 Mbp pairs through pa_align (simple, trace off / on)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import astar_pairwise_aligner_amd as pa
import oracle
from tests.util_seq import gen_pair
from tests.test_gpu_engine import gpu_params

pa.require_gpu()
for n in [int(x) for x in sys.argv[1:]] or [1_000_000]:
    t = time.time()
    a, b = gen_pair(n, 0.05, 1)
    print(f"n={n} generated in {time.time() - t:.1f}s", flush=True)
    al0 = gpu_params(pa, oracle.params_simple()).make_aligner(False)
    for rep in range(2):
        t = time.time()
        cost, _, st = al0.align_with_stats(a, b)
        print(f"n={n} simple cost-only: cost {cost} in {time.time() - t:.3f} s  tries {st['f_max_tries']} blocks {st['num_blocks']} lanes {st['computed_lanes']} t_compute {st['t_compute']:.3f}", flush=True)
    if n <= 2_000_000:
        al1 = gpu_params(pa, oracle.params_simple()).make_aligner(True)
        t = time.time()
        c1, cigar, st = al1.align_with_stats(a, b)
        print(f"n={n} simple with trace: cost {c1} in {time.time() - t:.3f} s cigar {len(cigar)} chars, verify {oracle.cigar_verify(cigar, a, b) if n <= 1_000_000 else 'skipped'} t_dt {st['t_dt']:.3f} t_fill {st['t_fill']:.3f}", flush=True)
    # the same cost from the full-matrix batch kernel
    if n <= 1_000_000:
        costs, ms = pa.Batch([(a, b)]).run()
        print(f"n={n} full DP batch: cost {int(costs[0])} in {ms:.1f} ms", flush=True)
