"""Generates the regression fixtures of tests/golden/ (run from the repository root: python tests/golden/make_golden.py).

NOT reference-derived: the reference (Rust, nightly, un-vendored dependencies) cannot be built or run in this image, and its
own tests never compare CIGAR strings or band statistics.  These files freeze what THIS build's CPU oracle / CPU-kernel engine
produce today, so that later rounds cannot change a CIGAR tie-break, a band decision or an operator output unnoticed:
  rectangles.json           (a, b, h_in, v_in) -> (ret, h_out, v_out) of the scalar oracle (scalar::row, pa-bitpacking/src/scalar.rs:37-46)
  presets_pa_test_pairs.json (cost, cigar, band statistics) of the nw / simple / full presets on the 8 pairs of pa-test/src/lib.rs:7-20
                             and a few generated pairs
The costs in it ARE reference-pinned (== Levenshtein, the reference's acceptance rule); CIGAR strings and statistics are regression only."""
import json
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq  # noqa: E402

STAT_KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "dt_trace_tries", "dt_trace_success", "dt_trace_fallback",
             "fill_tries", "fill_success", "fill_fallback", "f_max_tries", "sanity_violations"]


def rectangles():
    rng = random.Random(20260927)
    out = []
    for n, w in [(1, 1), (7, 1), (16, 2), (33, 3), (64, 1), (100, 4), (255, 5), (256, 8), (256, 9), (300, 2), (17, 7), (129, 6)]:
        m = 64 * w - rng.randint(0, 63)
        a, b = rand_seq(n, rng.randint(1, 10**6)), rand_seq(m, rng.randint(1, 10**6))
        pa, pb = oracle.bitprofile_build(a, b)
        h = np.zeros(n, oracle.H_DTYPE)
        v = np.zeros(w, oracle.V_DTYPE)
        for i in range(n):  # any legal deltas: (p, m) in {(1,0), (0,1), (0,0)}
            r = rng.random()
            h[i] = (1, 0) if r < 0.6 else ((0, 1) if r < 0.8 else (0, 0))
        for j in range(w):
            p = rng.getrandbits(64)
            mm = rng.getrandbits(64) & ~p
            v[j] = (p, mm)
        h_in = [[int(x["p"]), int(x["m"])] for x in h]
        v_in = [[int(x["p"]), int(x["m"])] for x in v]
        ret = oracle.scalar_row(pa, pb, h, v)
        out.append({"a": a.decode(), "b": b.decode(), "h_in": h_in, "v_in": [[str(p), str(m)] for p, m in v_in], "ret": int(ret),
                    "h_out": [[int(x["p"]), int(x["m"])] for x in h], "v_out": [[str(int(x["p"])), str(int(x["m"]))] for x in v]})
    return out


def presets():
    pairs = [(a.decode(), b.decode()) for a, b in PA_TEST_PAIRS]
    for n, e, s in [(300, 0.1, 1), (1000, 0.05, 2), (2500, 0.2, 3), (513, 0.6, 4)]:
        a, b = gen_pair(n, e, s)
        pairs.append((a.decode(), b.decode()))
    out = []
    for a, b in pairs:
        rec = {"a": a, "b": b, "levenshtein": oracle.levenshtein(a.encode(), b.encode())}
        for name, prm in (("nw", oracle.params_nw()), ("simple", oracle.params_simple()), ("full", oracle.params_full())):
            cost, cigar, stats = oracle.cpu_align(a.encode(), b.encode(), prm, trace=True)
            rec[name] = {"cost": cost, "cigar": cigar, "stats": {k: int(stats[k]) for k in STAT_KEYS}}
        out.append(rec)
    return out


if __name__ == "__main__":
    here = Path(__file__).resolve().parent
    (here / "rectangles.json").write_text(json.dumps({"note": "regression fixture from this build's scalar oracle; see make_golden.py", "rectangles": rectangles()}, indent=0))
    (here / "presets_pa_test_pairs.json").write_text(json.dumps({"note": "regression fixture (costs == Levenshtein are reference-pinned; CIGARs and statistics are NOT reference-derived); see make_golden.py",
                                                                 "pairs": presets()}, indent=0))
    print("written")
