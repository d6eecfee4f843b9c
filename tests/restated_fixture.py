"""The seeded pairs and the row format of tests/golden/restated_<variant>.json.

Those files hold what the SECOND restatement of the reference's A*PA2 host logic (oracle/astarpa2_restated.py: pure Python on big integers,
written from the Rust text alone, no line shared with csrc/engine.hpp or any kernel) returns for N_PAIRS seeded pairs under each of the 26
parameter sets of tests/test_restated_engine.py `variants`: cost, SHA-256 of the CIGAR string (first 16 hex digits) and the eleven
statistics.  tests/golden/make_restated.py (run in the build container) writes them; tests/test_gpu_restated_fixtures.py pushes the same
pairs through pa_align and the batch kernels on the GPU and compares -- no engine.hpp in between; tests/test_restated_fixtures.py checks
a sample on the CPU.  The pairs are not stored: pair_for(i) regenerates them from the index (Python's Mersenne Twister through
random.Random(seed): randrange / choice / random are stable across 3.x)."""
import hashlib
import json
import random
from pathlib import Path

from tests.util_seq import gen_pair, mutate, rand_seq

GOLDEN = Path(__file__).resolve().parent / "golden"
N_PAIRS = 2048
KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "dt_trace_tries", "dt_trace_success",
        "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback"]
DENSE_MAX_N = 2500  # `nw` (dense blocks: one Block per column in the restatement) takes the pairs up to this length only


def pair_for(i: int):
    """Pair i: lengths 1 .. 9 000 (a third each below 300, below 2 500, above), divergence 0 .. 80 %, a quarter with one long indel, one in
    twenty unrelated, one in ten low-complexity (tandem copies of a short unit with a few edits)."""
    rng = random.Random(0x5EED0000 + i)
    n = rng.choice([rng.randint(1, 300), rng.randint(300, 2500), rng.randint(2500, 9000)])
    e = rng.choice([0.0, 0.01, 0.03, 0.08, 0.15, 0.3, 0.8])
    mode = rng.random()
    if mode < 0.10:
        unit = rand_seq(rng.randint(2, 60), rng.randint(1, 10**9))
        a = (unit * (n // len(unit) + 1))[:n]
        b = bytearray(a)
        for _ in range(rng.randint(0, max(1, int(e * len(b))))):
            q = rng.randrange(len(b))
            b[q] = rng.choice(b"ACGT")
        cut = rng.randint(0, len(b))
        b = bytes(b[:cut] + b[cut + rng.randint(0, min(200, len(b) // 3)):]) or b"A"
        return a, b
    a, b = gen_pair(n, e, rng.randint(1, 10**9))
    if mode < 0.35 and n > 50:
        cut = rng.randint(0, len(b) - 1)
        ln = rng.randint(1, max(1, min(1500, len(b) // 2)))
        b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, rng.randint(1, 10**9)) + b[cut:]
        b = b or b"A"
    elif mode < 0.40:
        b = rand_seq(rng.randint(1, n + 50), rng.randint(1, 10**9))
    return a, b


def takes(name: str, i: int, a: bytes) -> bool:
    return name != "nw" or len(a) <= DENSE_MAX_N


def row_of(cost: int, cigar: str, stats) -> list:
    return [int(cost), hashlib.sha256(cigar.encode()).hexdigest()[:16]] + [int(stats[k]) for k in KEYS]


def load(name: str) -> dict:
    return json.loads((GOLDEN / f"restated_{name}.json").read_text())


def variant_names() -> list:
    return sorted(p.stem[len("restated_"):] for p in GOLDEN.glob("restated_*.json") if not p.stem.startswith(("restated_long_", "restated_kcoll_")))


def params_from_kwargs(pa, kw: dict):
    """The package's AstarPa2Params for the keyword arguments of restated.align (its defaults: oracle/astarpa2_restated.py `Restated`) --
    straight from the fixture file, no oracle library involved."""
    d = dict(heuristic="gap", k=12, sparse_h=True, block_width=256, dt_trace=True, max_g=40, fr_drop=10, domain="astar", sparse=True,
             doubling="band", start="h0", factor=2.0, delta=1.0, incremental_doubling=False, p=0, prune=False)
    d.update(kw)
    heur = d["heuristic"] if d["domain"] == "astar" else "none"
    return pa.AstarPa2Params(domain=d["domain"], heuristic=heur, k=d["k"], p=d["p"], doubling=d["doubling"], doubling_start=d["start"],
                             factor=d["factor"], delta=d["delta"], block_width=d["block_width"],
                             front=pa.BlockParams(sparse=d["sparse"], simd=True, no_ilp=False, incremental_doubling=d["incremental_doubling"],
                                                  dt_trace=d["dt_trace"], max_g=d["max_g"], fr_drop=d["fr_drop"]),
                             sparse_h=d["sparse_h"], prune=d["prune"])


# ---- a second, smaller set of LONG pairs: bands of several strips (the K = 2 / 3 / 4 strips with their `eq` words in LDS), window retries,
#      re-fills taller than a strip -- tests/golden/restated_long_<variant>.json, same row format ----
N_LONG = 96
LONG_VARIANTS = ["simple", "full", "sh12", "gap_incr", "dijkstra", "gcsh_noprune", "linear300", "gap_nodt"]


def long_pair_for(i: int):
    """Long pair i: 8 000 .. 60 000 bases, divergence 1 .. 20 %, a third with one long indel (up to 4 000 bases)."""
    rng = random.Random(0x10DE0000 + i)
    n = rng.randint(8_000, 60_000)
    e = rng.choice([0.01, 0.03, 0.05, 0.08, 0.12, 0.2])
    a, b = gen_pair(n, e, rng.randint(1, 10**9))
    if rng.random() < 0.34:
        cut = rng.randint(0, len(b) - 1)
        ln = rng.randint(200, 4000)
        b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, rng.randint(1, 10**9)) + b[cut:]
        b = b or b"A"
    return a, b


def load_long(name: str) -> dict:
    return json.loads((GOLDEN / f"restated_long_{name}.json").read_text())


# ---- a third set: k-mers longer than 16 whose u32 keys COLLIDE -- the reference keys its match table on `q as u32`
#      (pa-heuristic/src/matches/exact.rs:47,53,56), i.e. on the LAST 16 characters of a k-mer, so seeds / k-mers of b that agree there match
#      each other whatever their first k - 16 characters.  tests/golden/restated_kcoll_<set>.json, same row format ----
N_COLL = 768
COLL_VARIANTS = {
    "gcsh_k20_p0_prune": dict(heuristic="gcsh", k=20, p=0, prune=True),
    "gcsh_k20_p5_prune_incr": dict(heuristic="gcsh", k=20, p=5, prune=True, incremental_doubling=True),
    "gcsh_k24_p14_incr": dict(heuristic="gcsh", k=24, p=14, incremental_doubling=True),
    "gcsh_k17_p3_prune_nodt": dict(heuristic="gcsh", k=17, p=3, prune=True, dt_trace=False),
    "sh20": dict(heuristic="sh", k=20),
    "sh31_incr": dict(heuristic="sh", k=31, incremental_doubling=True),
}


def collision_pair_for(i: int, k: int):
    """Collision pair i for seed length k > 16: a = 5 .. 400 seeds, of which a third to all take their last 16 characters from a small pool
    of tails (groups of seeds with ONE u32 key); b = a with 0 .. 15 % edits (an edit in the first k - 16 characters of a pooled seed leaves a
    k-mer that equals no seed and still matches the whole group), a quarter with a block of a's seeds shuffled in, a fifth with an indel."""
    rng = random.Random(0xC0110000 + 1000 * k + i)
    nseeds = rng.choice([rng.randint(5, 40), rng.randint(40, 150), rng.randint(150, 400)])
    pool = [rand_seq(16, rng.randint(1, 10**9)) for _ in range(max(1, nseeds // rng.choice([3, 6, 12, 30])))]
    share = rng.choice([0.34, 0.6, 1.0])
    seeds = []
    for _ in range(nseeds):
        head = rand_seq(k - 16, rng.randint(1, 10**9))
        seeds.append(head + (rng.choice(pool) if rng.random() < share else rand_seq(16, rng.randint(1, 10**9))))
    a = b"".join(seeds) + rand_seq(rng.randint(0, k - 1), rng.randint(1, 10**9))
    e = rng.choice([0.0, 0.01, 0.03, 0.08, 0.15])
    b = mutate(a, e, rng.randint(1, 10**9)) if e > 0 else a
    mode = rng.random()
    if mode < 0.25:
        cut = rng.randint(0, len(b))
        some = seeds[:]
        rng.shuffle(some)
        b = b[:cut] + b"".join(some[: rng.randint(1, min(20, nseeds))]) + b[cut:]
    elif mode < 0.45 and len(b) > 60:
        cut = rng.randint(0, len(b) - 1)
        b = b[:cut] + b[cut + rng.randint(1, min(600, len(b) // 2)):]
    return a, (b or b"A")


def load_coll(name: str) -> dict:
    return json.loads((GOLDEN / f"restated_kcoll_{name}.json").read_text())


def count_collision_matches(a: bytes, b: bytes, k: int) -> int:
    """(seed start, j) pairs whose k-mers DIFFER and whose last 16 characters agree: the matches only the u32 key makes."""
    tails = {}
    for s in range(0, len(a) - k + 1, k):
        tails.setdefault(a[s + k - 16:s + k], []).append(s)
    return sum(a[s:s + k] != b[j:j + k] for j in range(len(b) - k + 1) for s in tails.get(b[j + k - 16:j + k], ()))
