"""Randomised parity soak: random batch shapes (strip height, chained / sequential), ragged sizes and divergences;
costs against the oracle, traced batches against the CPU-kernel engine's cost AND CIGAR string.
Usage: python tests/tools/fuzz_gpu.py SECONDS [SEED]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
import oracle
from tests.util_seq import gen_pair, rand_seq

pa.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
prm = oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True,
                         incremental_doubling=False, dt_trace=False)
t0 = time.time()
rounds = checked = 0
while time.time() - t0 < budget:
    k = int(rng.choice([1, 2, 4, 8]))
    mode = str(rng.choice(["chain", "seq", "auto"]))
    if mode == "auto":
        os.environ.pop("PA_BATCH_MODE", None)
        os.environ.pop("PA_STRIP_K", None)
    else:
        os.environ["PA_BATCH_MODE"] = mode
        os.environ["PA_STRIP_K"] = str(k)
    npairs = int(rng.integers(1, 40))
    pairs = []
    for _ in range(npairs):
        kind = rng.integers(0, 6)
        n = int(rng.choice([rng.integers(0, 70), rng.integers(0, 3000), rng.integers(2000, 2048 * k + 3000)]))
        if kind <= 2:
            pairs.append(gen_pair(n, float(rng.choice([0.0, 0.02, 0.1, 0.3])), seed=int(rng.integers(1 << 30))))
        elif kind == 3:
            pairs.append((rand_seq(n, seed=int(rng.integers(1 << 30))), rand_seq(int(rng.integers(0, 4000)), seed=int(rng.integers(1 << 30)))))
        elif kind == 5:
            # low complexity: runs of one letter (the first rows of b may not hold a[0] at all: the first-column case that random
            # sequences hide), b = a with a different head and a few edits
            runs, letters = [], b"ACGT"
            while sum(len(r) for r in runs) < max(n, 1):
                runs.append(bytes([letters[int(rng.integers(0, 4))]]) * int(rng.integers(1, 700)))
            a = b"".join(runs)[:max(n, 1)]
            head = bytes([letters[int(rng.integers(0, 4))]]) * int(rng.integers(0, 900))
            bb = bytearray(head + a[int(rng.integers(0, min(len(a), 300) + 1)):])
            for _ in range(int(rng.integers(0, 20))):
                if bb:
                    bb[int(rng.integers(0, len(bb)))] = letters[int(rng.integers(0, 4))]
            pairs.append((a, bytes(bb)))
        else:
            a = rand_seq(n, seed=int(rng.integers(1 << 30)))
            cut = int(rng.integers(0, n + 1))
            pairs.append((a, a[:cut] + rand_seq(int(rng.integers(0, 3000)), seed=7) + a[cut:]))
    traced = bool(rng.integers(0, 2))
    band = None if traced or rng.integers(0, 2) else float(rng.choice([0.0, 0.01, 0.05, 0.5]))
    if band is not None:
        os.environ.pop("PA_BATCH_MODE", None)
    b = pa.Batch(pairs, trace=traced, band=band)
    if traced:
        costs, cigars, _, _ = b.align()
    else:
        costs, _ = b.run()
        cigars = [None] * npairs
    for (x, y), c, cg in zip(pairs, costs, cigars):
        want = oracle.levenshtein(x, y) if len(x) * len(y) < 3_000_000 else oracle.nw_cost(x, y, True)
        assert c == want, ("cost", k, mode, traced, band, len(x), len(y), int(c), want)
        if traced and len(x) * len(y) < 40_000_000:
            wc, wcg, _ = oracle.cpu_align(x, y, prm)
            assert (c, cg) == (wc, wcg), ("cigar", k, mode, len(x), len(y))
        checked += 1
    b.close()
    rounds += 1
print(f"fuzz ok: {rounds} batches, {checked} pairs in {time.time() - t0:.0f} s")
