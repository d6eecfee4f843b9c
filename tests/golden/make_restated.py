"""Writes tests/golden/restated_<variant>.json: oracle/astarpa2_restated.py over the seeded pairs of tests/restated_fixture.py under every
parameter set of tests/test_restated_engine.py `variants` (run in the build container: python tests/golden/make_restated.py [processes]).
Nothing of csrc/ is imported: the expected values come from the second restatement alone.

These are CROSS-RESTATEMENT fixtures, not reference-generated ones: both restatements were written by this repository's authors from the Rust
text (the reference itself needs a Rust nightly toolchain and un-vendored crates: it cannot be built or run here).  They catch a divergence
between the two in-repo readings -- not a mis-reading both share.  The reference-derived vectors are tests/golden/pa_test_pairs.json and the
known answers of tests/test_oracle_kat.py; DEVIATIONS.md lists what each behaviour is pinned by."""
import json
import os
import sys
import time
from multiprocessing import get_context

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class _NoOracle:
    """`variants` wants the oracle module for the engine-side parameter structs; only the restatement's keyword arguments are used here."""

    def __getattr__(self, _):
        return lambda *a, **k: None


def work(job):
    from oracle import astarpa2_restated as restated
    from tests.restated_fixture import pair_for, row_of, takes

    name, kw, i = job
    a, b = pair_for(i)
    if not takes(name, i, a):
        return None
    cost, cigar, stats = restated.align(a, b, **kw)
    return row_of(cost, cigar, stats)


def work_long(job):
    from oracle import astarpa2_restated as restated
    from tests.restated_fixture import long_pair_for, row_of

    name, kw, i = job
    a, b = long_pair_for(i)
    cost, cigar, stats = restated.align(a, b, **kw)
    return row_of(cost, cigar, stats)


def work_coll(job):
    from oracle import astarpa2_restated as restated
    from tests.restated_fixture import collision_pair_for, row_of

    name, kw, i = job
    a, b = collision_pair_for(i, kw["k"])
    cost, cigar, stats = restated.align(a, b, **kw)
    return row_of(cost, cigar, stats)


def main():
    from tests.restated_fixture import GOLDEN, KEYS, N_PAIRS
    from tests.test_restated_engine import variants

    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    only = set(sys.argv[2:])
    vs = {name: kw for name, (_, kw) in variants(_NoOracle()).items()}
    from tests.restated_fixture import LONG_VARIANTS, N_LONG

    from tests.restated_fixture import COLL_VARIANTS, N_COLL

    with get_context("spawn").Pool(nproc) as pool:
        for name, kw in COLL_VARIANTS.items():  # k-mers beyond 16 with colliding u32 keys (python tests/golden/make_restated.py 8 kcoll)
            if only and "kcoll" not in only and name not in only:
                continue
            t0 = time.time()
            rows = pool.map(work_coll, [(name, kw, i) for i in range(N_COLL)], chunksize=4)
            doc = {"variant": name, "restated_kwargs": kw, "pairs": "tests/restated_fixture.py collision_pair_for(i, k), i = 0 .. n_pairs - 1",
                   "row": ["cost", "sha256(cigar)[:16]"] + KEYS, "n_pairs": N_COLL,
                   "source": "oracle/astarpa2_restated.py (second restatement; no csrc/ code involved); match keys: the reference's `q as u32` "
                             "(pa-heuristic/src/matches/exact.rs:47,53,56)", "rows": rows}
            (GOLDEN / f"restated_kcoll_{name}.json").write_text(json.dumps(doc, separators=(",", ":")) + "\n")
            print(f"kcoll {name}: {len(rows)} rows in {time.time() - t0:.1f} s", flush=True)
        if only == {"kcoll"}:
            return
        for name in LONG_VARIANTS:  # the long pairs (python tests/golden/make_restated.py 8 long: these alone)
            if only and "long" not in only and name not in only:
                continue
            t0 = time.time()
            rows = pool.map(work_long, [(name, vs[name], i) for i in range(N_LONG)], chunksize=1)
            doc = {"variant": name, "restated_kwargs": vs[name], "pairs": "tests/restated_fixture.py long_pair_for(i), i = 0 .. n_pairs - 1",
                   "row": ["cost", "sha256(cigar)[:16]"] + KEYS, "n_pairs": N_LONG,
                   "source": "oracle/astarpa2_restated.py (second restatement; no csrc/ code involved)", "rows": rows}
            (GOLDEN / f"restated_long_{name}.json").write_text(json.dumps(doc, separators=(",", ":")) + "\n")
            print(f"long {name}: {len(rows)} rows in {time.time() - t0:.1f} s", flush=True)
        if only == {"long"}:
            return
        for name, kw in vs.items():
            if only and name not in only:
                continue
            t0 = time.time()
            rows = pool.map(work, [(name, kw, i) for i in range(N_PAIRS)], chunksize=8)
            doc = {"variant": name, "restated_kwargs": kw, "pairs": "tests/restated_fixture.py pair_for(i), i = 0 .. n_pairs - 1",
                   "row": ["cost", "sha256(cigar)[:16]"] + KEYS, "n_pairs": N_PAIRS,
                   "source": "oracle/astarpa2_restated.py (second restatement; no csrc/ code involved)", "rows": rows}
            (GOLDEN / f"restated_{name}.json").write_text(json.dumps(doc, separators=(",", ":")) + "\n")
            print(f"{name}: {sum(r is not None for r in rows)} rows in {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
