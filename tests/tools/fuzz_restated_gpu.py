"""GPU soak against the SECOND restatement (oracle/astarpa2_restated.py: pure Python on big integers, written separately from the
Rust text, no line shared with csrc/engine.hpp): random parameter sets of tests/test_restated_engine.py (the three presets, GCSH with
pruning, incremental doubling, the other domains, linear search, dense blocks ...) and random pairs -- lengths 1 to 9 000, divergence
0 to 80 %, long indels, unrelated pairs.  pa_align (the sweep kernel, the host-driven HIP engine) and, where the parameters allow it,
the batched A*PA2 must return the restatement's cost, CIGAR string and eleven statistics.  The expected values are computed by a pool
of worker processes while the GPU works.
Usage: python tests/tools/fuzz_restated_gpu.py [seconds] [seed] [processes]"""
import os
import random
import sys
import time
from multiprocessing import get_context

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def expected(job):
    from oracle import astarpa2_restated as restated

    name, a, b, kw = job
    try:
        return restated.align(a, b, **kw)
    except Exception as e:  # noqa: BLE001
        return ("EXC", repr(e), {})


def main():
    import astar_pairwise_aligner_amd as pa
    import oracle
    from tests.test_gpu_engine import gpu_params
    from tests.test_restated_engine import KEYS, variants
    from tests.util_seq import gen_pair, rand_seq

    pa.require_gpu()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else min(64, os.cpu_count() or 1)
    vs = variants(oracle)
    gp = {name: gpu_params(pa, prm) for name, (prm, _) in vs.items()}
    batchable = {name for name in vs if pa.capi.batch_params_supported(gp[name])}
    t0 = time.time()
    n_pairs = bad = n_batched = 0
    per_variant = {}
    with get_context("spawn").Pool(nproc) as pool:
        while time.time() - t0 < budget:
            jobs = []
            for _ in range(4 * nproc):
                name = rng.choice(list(vs))
                n = rng.choice([rng.randint(1, 300), rng.randint(300, 2500), rng.randint(2500, 9000)])
                if name in ("nw",):
                    n = min(n, 2500)  # (dense blocks: one Block per column in the restatement)
                e = rng.choice([0.0, 0.01, 0.03, 0.08, 0.15, 0.3, 0.8])
                a, b = gen_pair(n, e, rng.randint(1, 10**9))
                mode = rng.random()
                if mode < 0.25 and n > 50:
                    cut = rng.randint(0, len(b) - 1)
                    ln = rng.randint(1, max(1, min(1500, len(b) // 2)))
                    b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, rng.randint(1, 10**9)) + b[cut:]
                    b = b or b"A"
                elif mode < 0.3:
                    b = rand_seq(rng.randint(1, n + 50), rng.randint(1, 10**9))
                jobs.append((name, a, b, vs[name][1]))
            want = pool.map_async(expected, jobs, chunksize=2)
            got = []
            for name, a, b, _ in jobs:
                got.append(pa.AstarPa2(gp[name], True).align_with_stats(a, b))
            # the batched A*PA2 on the jobs whose parameters it takes, one batch per variant
            batched = {}
            for name in batchable:
                idx = [i for i, j in enumerate(jobs) if j[0] == name]
                if not idx:
                    continue
                bt = pa.Batch([(jobs[i][1], jobs[i][2]) for i in idx], params=gp[name])
                cs, gs, _, _ = bt.align()
                st = bt.pair_stats()
                bt.close()
                for k, i in enumerate(idx):
                    batched[i] = (int(cs[k]), gs[k], st[k])
            want = want.get()
            for i, (job, g, w) in enumerate(zip(jobs, got, want)):
                n_pairs += 1
                per_variant[job[0]] = per_variant.get(job[0], 0) + 1
                routes = [("pa_align", g)] + ([("batch", batched[i])] if i in batched else [])
                n_batched += i in batched
                for route, r in routes:
                    ok = w[0] != "EXC" and (r[0], r[1]) == (w[0], w[1]) and all(int(r[2][k]) == w[2][k] for k in KEYS)
                    if not ok:
                        bad += 1
                        print("MISMATCH", route, job[0], len(job[1]), len(job[2]), r[0], w[0], r[1] == w[1],
                              {k: (int(r[2][k]), w[2].get(k)) for k in KEYS if int(r[2][k]) != w[2].get(k)} if w[0] != "EXC" else w[1], flush=True)
    print(f"fuzz_restated_gpu: {n_pairs} pairs through pa_align ({n_batched} of them also through the batched A*PA2), {len(per_variant)} parameter sets, "
          f"{bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print("per parameter set:", dict(sorted(per_variant.items())))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
