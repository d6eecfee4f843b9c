"""The rendezvous of half-wave blocks (csrc/rdv_logic.hpp: two wavefronts of a workgroup run their two blocks as one strip) on host threads:
oracle/rdv_emu.cpp instantiates the protocol the kernels use over std::atomic, four threads per workgroup, every thread running the
product's band-search program (csrc/apa2_logic.hpp) over the CPU kernels.  Whatever the timing -- patient, impatient, one or several
workgroups -- cost and statistics of every pair equal the program run alone (and the host engine), the counters balance, and nothing
hangs.  `make -C oracle tsan_rdv` runs the same under ThreadSanitizer (test_thread_sanitizer below, when the toolchain has it)."""
import random
import shutil
import subprocess
from pathlib import Path

import pytest

from tests.util_seq import gen_pair

ROOT = Path(__file__).resolve().parent.parent


def some_pairs(n, seed):
    rng = random.Random(seed)
    ps = [gen_pair(rng.choice([300, 900, 2000, 3500]), rng.choice([0.0, 0.02, 0.08, 0.15, 0.3]), seed=seed * 1000 + i) for i in range(n)]
    ps[3] = (b"", b"ACGT")  # degenerate: handed back, takes no part
    ps[7] = gen_pair(9000, 0.25, seed=seed)  # bands taller than half a wave run alone
    return ps


def test_results_do_not_depend_on_the_rendezvous(oracle):
    pairs = some_pairs(90, 4)
    alone, c0 = oracle.rdv_emu_run(pairs, groups=2, patience_us=-1)
    assert c0["fused"] == c0["served"] == 0
    for groups, patience in ((1, 300.0), (2, 200.0), (3, 5000.0), (2, 0.0)):
        got, c = oracle.rdv_emu_run(pairs, groups=groups, patience_us=patience)
        assert got == alone, (groups, patience)
        assert c["fused"] == c["served"], c  # every strip a partner ran was run by exactly one partner
        if patience >= 200.0:
            assert c["fused"] > 20, c  # blocks really met
    prm = oracle.params_simple()
    for i in range(0, len(pairs), 9):
        a, b = pairs[i]
        if not a or not b:
            assert alone[i][0] != 0
            continue
        cost, _, st = oracle.cpu_align(a, b, prm)
        assert alone[i][:6] == (0, cost, st["f_max_tries"], st["num_blocks"], st["computed_lanes"], st["unique_lanes"]), i


def test_the_last_wavefront_of_a_workgroup_does_not_wait(oracle):
    """One pair for four threads: three leave at once.  The one that works must not sit out its patience (ten seconds here) block after
    block: it sees `live <= 1` -- before posting, or while it waits (then it withdraws at once)."""
    import time

    t = time.time()
    rows, c = oracle.rdv_emu_run([gen_pair(3000, 0.05, seed=1)], groups=1, patience_us=1e7)
    assert rows[0][0] == 0 and c["fused"] == c["served"] == 0 and c["withdrawn"] <= 2 and c["alone"] > 5, c
    assert time.time() - t < 5.0


def test_thread_sanitizer():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "-s", "tsan_rdv"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("ThreadSanitizer build not available: " + r.stderr[-300:])
    out = subprocess.run([str(ROOT / "oracle" / "_build" / "rdv_emu_tsan"), "60", "2", "300"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    assert "differences 0" in out.stdout
