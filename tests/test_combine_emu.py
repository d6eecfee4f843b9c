"""The call combiner's gathering protocol (csrc/combine_logic.hpp, what engine_hip.hip runs behind pa_align and the astarpa-c symbols)
on host threads with a stand-in batch (oracle/combine_emu.cpp): every caller gets its own result, nobody is left waiting -- also when
batches throw --, requests travel in groups and several batches run side by side; `make -C oracle tsan_combine`: the same under
ThreadSanitizer.  The GPU side: tests/test_gpu_engine.py::test_concurrent_callers_are_combined_and_get_the_single_call_results."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_every_caller_gets_its_result_and_requests_travel_in_groups(oracle):
    r = oracle.combine_emu_run(threads=32, calls=60, batch_us=400)
    assert r["wrong"] == 0 and r["failed"] == 0
    assert r["batches"] < 32 * 60 // 3 and r["largest_group"] >= 8, r  # groups, not single requests
    assert 2 <= r["side_by_side"] <= 4, r  # several batches at a time, never more than the bound
    # a caller whose request already travels in a batch does not gather (round 5's advisor finding: it ran EMPTY groups and returned late)
    assert r["empty_groups"] == 0 and r["late_returns"] == 0, r
    slow = oracle.combine_emu_run(threads=32, calls=12, batch_us=6000)  # the advisor's repro: 32 threads, a 6 ms stand-in batch
    assert slow["wrong"] == 0 and slow["empty_groups"] == 0 and slow["late_returns"] == 0, slow
    one = oracle.combine_emu_run(threads=1, calls=20, batch_us=100)
    assert one["wrong"] == 0 and one["batches"] == 20 and one["largest_group"] == 1


def test_a_batch_that_throws_leaves_nobody_waiting(oracle):
    r = oracle.combine_emu_run(threads=16, calls=40, batch_us=200, fail_every=3)
    assert r["wrong"] == 0 and 0 < r["failed"] < 16 * 40, r  # the failed groups' callers were told so (they take the single-pair path)


def test_thread_sanitizer():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "-s", "tsan_combine"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("ThreadSanitizer build not available: " + r.stderr[-300:])
    out = subprocess.run([str(ROOT / "oracle" / "_build" / "combine_emu_tsan"), "16", "30"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
