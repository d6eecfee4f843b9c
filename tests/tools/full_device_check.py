"""Drives tools/apa2_full_device_check (the full-family per-pair program + flat GCSH running on the device, one thread per pair, one
launch per pass, contours re-derived on the host in between): random pairs over the presets it covers, compared with the host engine
over the CPU kernels -- cost, f_max_tries and the four block counters.
Usage: python tests/tools/full_device_check.py [pairs] [seed] [--host]"""
import os
import random
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from tests.util_seq import gen_pair, rand_seq  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
host = "--host" in sys.argv
base = dict(domain="astar", doubling="band", start="h0", factor=2.0, block_width=256, sparse=True, dt_trace=True, max_g=40, fr_drop=10)
VARIANTS = {  # name -> (k, p, prune, incremental, sparse_h, heuristic code, engine parameters)
    "full": (12, 14, 1, 1, 1, 3, oracle.params_full()),
    "gcsh_k6_p3": (6, 3, 1, 1, 1, 3, oracle.make_params(**base, heuristic="gcsh", k=6, p=3, prune=True, incremental_doubling=True, sparse_h=True)),
    "gcsh_k8_noincr": (8, 0, 1, 0, 1, 3, oracle.make_params(**base, heuristic="gcsh", k=8, p=0, prune=True, incremental_doubling=False, sparse_h=True)),
    "gcsh_k10_dense_h": (10, 5, 1, 0, 0, 3, oracle.make_params(**base, heuristic="gcsh", k=10, p=5, prune=True, incremental_doubling=False, sparse_h=False)),
    "simple": (0, 0, 0, 0, 1, 1, oracle.params_simple()),
    "gap_incr": (0, 0, 0, 1, 1, 1, oracle.make_params(**base, heuristic="gap", incremental_doubling=True, sparse_h=True)),
    "dijkstra_incr": (0, 0, 0, 1, 1, 0, oracle.make_params(**base, heuristic="none", incremental_doubling=True, sparse_h=True)),
}
jobs = []
for it in range(npairs):
    name = rng.choice(list(VARIANTS))
    n = rng.choice([rng.randint(1, 300), rng.randint(300, 2000), rng.randint(2000, 6000)])
    if name == "gcsh_k10_dense_h":
        n = min(n, 2500)
    a, b = gen_pair(n, rng.choice([0.0, 0.01, 0.05, 0.1, 0.2, 0.4]), rng.randint(1, 10**9))
    if rng.random() < 0.25 and n > 50:
        cut = rng.randint(0, len(b) - 1)
        ln = rng.randint(1, max(1, min(800, len(b) // 2)))
        b = (b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, it + 3) + b[cut:]) or b"A"
    jobs.append((name, a, b))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "pairs.txt")
    with open(path, "w") as f:
        for name, a, b in jobs:
            k, p, prune, incr, sh, heur, _ = VARIANTS[name]
            f.write(f"{k} {p} {prune} {incr} {sh} {heur} {a.decode()} {b.decode()}\n")
    r = subprocess.run([os.path.join(ROOT, "tools", "apa2_full_device_check"), path] + (["--host"] if host else []), capture_output=True, text=True, timeout=900)
print(r.stderr.strip())
lines = [ln.split() for ln in r.stdout.strip().split("\n") if ln.strip()]
assert len(lines) == len(jobs), (len(lines), len(jobs), r.stdout[-500:], r.stderr[-500:])
bad = multi = 0
for (name, a, b), ln in zip(jobs, lines):
    status, cost, tries, nb, ninc, comp, uniq, passes = map(int, ln)
    want = oracle.cpu_align(a, b, VARIANTS[name][6])
    ws = want[2]
    ok = status == 0 and (cost, tries, nb, ninc, comp, uniq) == (want[0], ws["f_max_tries"], ws["num_blocks"], ws["num_incremental_blocks"], ws["computed_lanes"], ws["unique_lanes"])
    multi += passes > 1
    if not ok:
        bad += 1
        print("MISMATCH", name, len(a), len(b), ln, want[0], {k: ws[k] for k in ("f_max_tries", "num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes")})
print(f"full_device_check: {len(jobs)} pairs ({multi} with more than one pass) over {len(VARIANTS)} parameter sets, {'host' if host else 'device'} run, {bad} mismatches")
sys.exit(1 if bad else 0)
