"""Step-by-step GPU check of the batched A*PA2 path, every step in its own process with a time limit (a kernel that hangs must
not take the whole gpurun call with it).  python tests/tools/apa2_debug.py [step]"""
import subprocess
import sys
import time

STEPS = {
    "tiny": [(100, 0.05, 1)],
    "one_block": [(250, 0.1, 2)],
    "blocks": [(3000, 0.1, 3)],
    "strips": [(10000, 0.15, 4)],
    "strips2": [(30000, 0.2, 6), (20000, 0.3, 5)],
    "many": [(n, e, 100 + n) for n in (1, 2, 31, 64, 65, 255, 256, 257, 511, 513, 1025, 2049, 4097, 8191) for e in (0.0, 0.05, 0.4, 1.0)],
}


def run_step(name):
    sys.path.insert(0, ".")
    import astar_pairwise_aligner_amd as pa
    import oracle
    from tests.test_sweep_emu import KEYS
    from tests.util_seq import gen_pair

    pairs = [gen_pair(n, e, s) for n, e, s in STEPS[name]]
    t0 = time.time()
    batch = pa.Batch(pairs, params=pa.AstarPa2Params.simple())
    print(name, "created", round(time.time() - t0, 3), flush=True)
    costs, cigars, f_ms, t_ms = batch.align()
    print(name, "aligned", round(time.time() - t0, 3), "fwd ms", f_ms, "trace ms", t_ms, "fallbacks", batch.trace_fallbacks(), flush=True)
    stats = batch.pair_stats()
    bad = 0
    for (a, b), c, cg, st in zip(pairs, costs, cigars, stats):
        want = oracle.cpu_align(a, b, oracle.params_simple())
        ok = c == want[0] and cg == want[1] and all(st[k] == want[2][k] for k in KEYS)
        if not ok:
            bad += 1
            print("  MISMATCH", len(a), len(b), "cost", c, want[0], "cigar_eq", cg == want[1], {k: (st[k], want[2][k]) for k in KEYS if st[k] != want[2][k]}, flush=True)
    print(name, "bad", bad, "of", len(pairs), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_step(sys.argv[1])
    else:
        for name in STEPS:
            try:
                r = subprocess.run([sys.executable, __file__, name], timeout=100)
                print("step", name, "rc", r.returncode, flush=True)
            except subprocess.TimeoutExpired:
                print("step", name, "TIMEOUT", flush=True)
                break
