"""Randomised parity soak of the BIT-SLICED full-DP kernel (csrc/slice_kernel.hpp): random batches -- 40 to 2500 pairs, ragged lengths inside
the groups of 32 (capture events), unrelated / swapped / low-complexity / empty sequences, every instantiated number of rows per lane,
chains of strips -- with every distance compared with the oracle (pairs up to 3000 bp) and, for the whole batch, with the strip kernels
(PA_SLICE=0), which the oracle pins elsewhere.
Usage: python tests/tools/fuzz_slice.py SECONDS [SEED]"""
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
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ROWS = [1, 28, 32, 36, 40, 42, 44, 46, 48, 50, 52]
letters = b"ACGT"
t0 = time.time()
rounds = pairs_total = oracle_checked = 0
while time.time() - t0 < budget:
    npairs = int(rng.choice([rng.integers(40, 130), rng.integers(130, 700), rng.integers(700, 2500)]))
    top = int(rng.choice([300, 3000, 12_000, 40_000])) if npairs < 700 else int(rng.choice([300, 3000]))
    pairs = []
    for _ in range(npairs):
        kind = int(rng.integers(0, 8))
        n = int(rng.choice([rng.integers(1, 70), rng.integers(1, max(2, top // 4)), rng.integers(max(1, top // 2), top + 1)]))
        s = int(rng.integers(1 << 30))
        if kind <= 3:
            a, b = gen_pair(n, float(rng.choice([0.0, 0.01, 0.05, 0.15, 0.4])), seed=s)
        elif kind == 4:
            a, b = rand_seq(n, seed=s), rand_seq(int(rng.integers(1, top + 1)), seed=s + 1)  # unrelated, any lengths
        elif kind == 5:
            a, b = gen_pair(n, 0.1, seed=s)
            a, b = b, a
        elif kind == 6:  # low complexity: runs of one letter
            runs = []
            while sum(len(r) for r in runs) < n:
                runs.append(bytes([letters[int(rng.integers(0, 4))]]) * int(rng.integers(1, 400)))
            a = b"".join(runs)[:n]
            bb = bytearray(a[int(rng.integers(0, min(len(a), 200) + 1)):] or b"A")
            for _ in range(int(rng.integers(0, 12))):
                bb[int(rng.integers(0, len(bb)))] = letters[int(rng.integers(0, 4))]
            b = bytes(bb)
        else:  # an empty side now and then (such pairs are in no group)
            a, b = gen_pair(n, 0.05, seed=s)
            if rng.integers(0, 3) == 0:
                a = b""
            elif rng.integers(0, 2) == 0:
                b = b""
        pairs.append((a, b))
    rows = int(rng.choice(ROWS))
    os.environ["PA_SLICE"] = str(rows)
    bt = pa.Batch(pairs)
    sh = bt.shape()
    assert sh.get("sliced_rows_per_lane", 0) > 0 and (rows == 1 or sh["sliced_rows_per_lane"] == rows), sh
    got, _ = bt.run()
    again, _ = bt.run()
    bt.close()
    assert np.array_equal(got, again), f"round {rounds}: second pass over the resident batch differs (seed {seed})"
    os.environ["PA_SLICE"] = "0"
    ref = pa.Batch(pairs)
    want, _ = ref.run()
    ref.close()
    bad = np.nonzero(np.asarray(got) != np.asarray(want))[0]
    assert bad.size == 0, f"round {rounds} (seed {seed}, rows {rows}): pairs {bad[:8].tolist()} differ from the strip kernels: {[(int(got[i]), int(want[i]), len(pairs[i][0]), len(pairs[i][1])) for i in bad[:8]]}"
    for i, (a, b) in enumerate(pairs):
        if max(len(a), len(b)) <= 3000 and oracle_checked < 40 * (rounds + 1):
            assert int(got[i]) == oracle.levenshtein(a, b), f"round {rounds} pair {i}: {int(got[i])} != oracle"
            oracle_checked += 1
    rounds += 1
    pairs_total += npairs
print(f"fuzz_slice: seed {seed}: {rounds} batches, {pairs_total} pairs through slice_kernel (every R), all equal to the strip kernels; "
      f"{oracle_checked} of them also against the oracle; {time.time() - t0:.0f} s")
