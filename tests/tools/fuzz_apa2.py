"""GPU soak for the batched A*PA2 (pa_batch_create_params): random parameter variants and batches of random pairs -- lengths 1 to
60 000, divergence 0 to 80 %, long indels, unrelated pairs, empty sequences -- against the host engine over the CPU oracle kernels:
cost, CIGAR string and all twelve statistics of every pair.  The CPU side runs on a thread pool (ctypes releases the GIL).
Parameter sets: the `simple` family of tests/test_sweep_emu.py and (round 4) `full` and its relatives.
Usage: python tests/tools/fuzz_apa2.py [seconds] [seed]"""
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import astar_pairwise_aligner_amd as pa
import oracle
from tests.test_gpu_engine import gpu_params
from tests.test_sweep_emu import KEYS, variants
from tests.util_seq import gen_pair, rand_seq

pa.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
vs = variants(oracle)
# ... and the `full` family (round 4: GCSH with local pruning, pruning of matches, incremental doubling in the batch kernels)
from tests.test_restated_engine import variants as all_variants  # noqa: E402

for _name in ("full", "gcsh_noprune", "gcsh_k8_p0_prune", "gcsh_k6_p3_prune_incr", "gcsh_k10_p5_nosparseh", "gap_incr", "sh12_incr",
              "dijkstra_incr_nodt", "gap_incr_f15"):
    vs[_name] = all_variants(oracle)[_name][0]
t0 = time.time()
n_pairs = n_batches = bad = fallbacks = retries = 0
pool = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1))
while time.time() - t0 < budget:
    name = rng.choice(list(vs))
    pairs = []
    for _ in range(rng.choice([1, 3, 17, 64, 200])):
        n = rng.choice([rng.randint(1, 600), rng.randint(600, 6000), rng.randint(6000, 60000)]) if rng.random() < 0.7 else rng.randint(1, 3000)
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
        elif mode < 0.37:
            a, b = rng.choice([(b"", b), (a, b""), (b"", b"")])
        elif mode < 0.45:  # low complexity: runs of one letter, b = a behind a foreign head
            runs = []
            while sum(len(r) for r in runs) < n:
                runs.append(bytes([rng.choice(b"ACGT")]) * rng.randint(1, 700))
            a = b"".join(runs)[:n]
            b = bytes([rng.choice(b"ACGT")]) * rng.randint(0, 900) + a[rng.randint(0, min(n, 300)):]
            b = b or b"A"
        pairs.append((a, b))
    bt = pa.Batch(pairs, params=gpu_params(pa, vs[name]))
    costs, cigars, _, _ = bt.align()
    stats = bt.pair_stats()
    fallbacks += bt.trace_fallbacks()
    retries += bt.window_retries()
    bt.close()
    want = list(pool.map(lambda p: oracle.cpu_align(p[0], p[1], vs[name]), pairs))
    n_batches += 1
    for (a, b), c, g, st, w in zip(pairs, costs, cigars, stats, want):
        n_pairs += 1
        if not ((int(c), g) == (w[0], w[1]) and all(st[k] == w[2][k] for k in KEYS)):
            bad += 1
            print("MISMATCH", name, len(a), len(b), int(c), w[0], g == w[1], {k: (st[k], w[2][k]) for k in KEYS if st[k] != w[2][k]}, flush=True)
print(f"{n_pairs} pairs in {n_batches} batches through the batched A*PA2 ({len(vs)} parameter sets), {fallbacks} handed to the host engine, "
      f"{retries} aligned again with full block columns (their band left the window), {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
