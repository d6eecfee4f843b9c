"""CPU soak: the product's host engine over the CPU oracle kernels (oracle.cpu_align = csrc/engine.hpp) against the second restatement
(oracle/astarpa2_restated.py), field by field, on worker processes: random parameter sets of tests/test_restated_engine.py, lengths 1
to 60 000, divergence 0 to 80 %, long indels, unrelated pairs, low-complexity stretches.
Usage: python tests/tools/fuzz_restated_cpu.py [seconds] [seed] [processes]"""
import os
import random
import sys
import time
from multiprocessing import get_context

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "dt_trace_tries", "dt_trace_success",
        "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback"]


def work(job):
    import oracle
    from oracle import astarpa2_restated as restated
    from tests.test_restated_engine import variants

    name, a, b = job
    prm, kw = variants(oracle)[name]
    try:
        want = oracle.cpu_align(a, b, prm)
    except Exception as e:  # noqa: BLE001
        want = ("ENGINE-EXC", repr(e), {})
    try:
        got = restated.align(a, b, **kw)
    except Exception as e:  # noqa: BLE001
        got = ("RESTATED-EXC", repr(e), {})
    if isinstance(want[0], str) or isinstance(got[0], str):
        return (name, len(a), len(b), f"{want[:2]} / {got[:2]}") if (isinstance(want[0], str) != isinstance(got[0], str)) else None
    if (got[0], got[1]) == (want[0], want[1]) and all(got[2][k] == want[2][k] for k in KEYS):
        return None
    return (name, len(a), len(b), got[0], want[0], got[1] == want[1], {k: (got[2][k], want[2][k]) for k in KEYS if got[2][k] != want[2][k]})


def main():
    import oracle
    from tests.test_restated_engine import variants
    from tests.util_seq import gen_pair, rand_seq

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, (os.cpu_count() or 2) - 1)
    oracle.build()
    names = list(variants(oracle))
    t0 = time.time()
    n_pairs = bad = 0
    with get_context("spawn").Pool(nproc) as pool:
        while time.time() - t0 < budget:
            jobs = []
            for _ in range(8 * nproc):
                name = rng.choice(names)
                n = rng.choice([rng.randint(1, 400), rng.randint(400, 4000), rng.randint(4000, 20000), rng.randint(20000, 60000)])
                if name == "nw":
                    n = min(n, 3000)
                if name in ("gap_nosparseh", "gcsh_k10_p5_nosparseh"):
                    n = min(n, 8000)  # (one heuristic call per row and column)
                e = rng.choice([0.0, 0.005, 0.02, 0.05, 0.1, 0.2, 0.4, 0.8])
                a, b = gen_pair(n, e, rng.randint(1, 10**9))
                mode = rng.random()
                if mode < 0.25 and n > 50:
                    cut = rng.randint(0, len(b) - 1)
                    ln = rng.randint(1, max(1, min(5000, len(b) // 2)))
                    b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, rng.randint(1, 10**9)) + b[cut:]
                    b = b or b"A"
                elif mode < 0.3:
                    b = rand_seq(rng.randint(1, n + 50), rng.randint(1, 10**9))
                elif mode < 0.36 and n > 100:  # a low-complexity stretch: seeds with many matches, runs the greedy steps love
                    cut = rng.randint(0, len(a) - 1)
                    run = bytes([rng.choice(b"ACGT")]) * rng.randint(20, 400)
                    a = a[:cut] + run + a[cut:]
                    b = b[:min(cut, len(b))] + run[:rng.randint(1, len(run))] + b[min(cut, len(b)):]
                jobs.append((name, a, b))
            for r in pool.imap_unordered(work, jobs, chunksize=1):
                n_pairs += 1
                if r is not None:
                    bad += 1
                    print("MISMATCH", r, flush=True)
            print(f"  ... {n_pairs} pairs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_restated_cpu: {n_pairs} pairs over {len(names)} parameter sets, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
