"""GPU soak for the device-side A*PA2 sweep (pipelined passes) and the batched DT-trace: random parameter variants and pairs
against the host-driven engine over the CPU oracle kernels -- cost, CIGAR string and (traced) every band statistic.
Usage: python tests/tools/fuzz_sweep.py [seconds] [seed]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import astar_pairwise_aligner_amd as pa
import oracle
from tests.test_gpu_batch_align import dt_params
from tests.test_gpu_engine import gpu_params
from tests.test_sweep_emu import KEYS, variants
from tests.util_seq import gen_pair, rand_seq

pa.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
vs = variants(oracle)
aligners = {(name, tr): gpu_params(pa, vs[name]).make_aligner(tr) for name in vs for tr in (True, False)}
dtp = dt_params(oracle)
t0 = time.time()
n_sweep = n_dt = bad = 0
while time.time() - t0 < budget:
    name = rng.choice(list(vs))
    n = rng.choice([rng.randint(1, 600), rng.randint(600, 6000), rng.randint(6000, 60000)])
    e = rng.choice([0.0, 0.005, 0.02, 0.05, 0.1, 0.2, 0.4, 0.8])
    s = rng.randint(1, 10**9)
    a, b = gen_pair(n, e, s)
    mode = rng.random()
    if mode < 0.3 and n > 50:
        cut = rng.randint(0, len(b) - 1)
        ln = rng.randint(1, max(1, min(5000, len(b) // 2)))
        b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, s + 1) + b[cut:]
        b = b or b"A"
    elif mode < 0.35:
        b = rand_seq(rng.randint(1, n + 50), s + 2)
    tr = rng.random() < 0.8
    want = oracle.cpu_align(a, b, vs[name], trace=tr)
    cost, cigar, stats = aligners[(name, tr)].align_with_stats(a, b)
    if tr:
        ok = cost == want[0] and cigar == want[1] and all(stats[k] == want[2][k] for k in KEYS)
    else:  # cost only: the distance itself (the reference's single-block path may end on an upper bound, see tests/test_gpu_sweep.py)
        ok = cigar is None and cost == oracle.nw_cost(a, b, True) <= want[0]
    n_sweep += 1
    if not ok:
        bad += 1
        print("SWEEP MISMATCH", name, n, e, s, tr, cost, want[0], flush=True)
    if n_sweep % 8 == 0:  # a small batch through the DT-trace traceback
        pairs = [gen_pair(rng.choice([rng.randint(1, 700), rng.randint(700, 9000)]), rng.choice([0.0, 0.01, 0.05, 0.12, 0.2, 0.35]), rng.randint(1, 10**9))
                 for _ in range(12)]
        bt = pa.Batch(pairs, trace=True, trace_params=gpu_params(pa, dtp))
        costs, cigars, _, _ = bt.align()
        bt.close()
        for (x, y), c, g in zip(pairs, costs, cigars):
            w = oracle.cpu_align(x, y, dtp)
            n_dt += 1
            if (int(c), g) != (w[0], w[1]):
                bad += 1
                print("DT MISMATCH", len(x), len(y), int(c), w[0], flush=True)
print(f"{n_sweep} pairs through the sweep, {n_dt} pairs through the DT-trace batch, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
