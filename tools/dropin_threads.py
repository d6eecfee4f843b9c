"""Drop-in symbol under concurrent callers: T host threads, each calling astarpa2_simple() (C ABI, GIL released by ctypes) on its
share of the same 10 kbp pairs.  Prints pairs/s per thread count; every result is compared with the single-thread run.
    python tools/dropin_threads.py [--pairs 400] [--threads 1,2,4,8,16]"""
import argparse
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=400)
    ap.add_argument("--len", type=int, default=10_000)
    ap.add_argument("--threads", default="1,2,4,8,16")
    ap.add_argument("--symbol", default="astarpa2_simple")
    args = ap.parse_args()
    divs = (0.01, 0.05, 0.10, 0.15)
    pa.require_gpu()
    pairs = [generate_pair(args.len, divs[i % 4], seed=1_000_000 + i) for i in range(args.pairs)]
    pa.c_abi_align(args.symbol, *pairs[0])
    ref = None
    for T in [int(x) for x in args.threads.split(",")]:
        def work(t):
            return [(i, pa.c_abi_align(args.symbol, *pairs[i])) for i in range(t, len(pairs), T)]
        with ThreadPoolExecutor(T) as ex:
            gate = threading.Barrier(T)

            def warm(t):  # every worker thread exactly once: the library's device buffers are pooled per host thread
                gate.wait()
                for i in range(3):
                    pa.c_abi_align(args.symbol, *pairs[(t + i) % len(pairs)])
            list(ex.map(warm, range(T)))
            t0 = time.perf_counter()
            res = [r for part in ex.map(work, range(T)) for r in part]
            dt = time.perf_counter() - t0
        got = [r for _, r in sorted(res)]
        if ref is None:
            ref = got
        assert got == ref, f"{T} threads: results differ from the single-thread run"
        print(f"threads {T:3d}: {len(pairs) / dt:9.1f} pairs/s  ({dt * 1e3 / len(pairs) * T:.3f} ms per call per thread)", flush=True)


if __name__ == "__main__":
    main()
