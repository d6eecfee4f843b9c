"""pa-bin's whole loop on the GPU, from a file to a file (pa_align_file / pa_align_file_params): C4-shaped input (10 kbp pairs at
1/5/10/15 %) written as .seq, read + aligned (cost + CIGAR) + written as "{cost},{cigar}" lines.
python tools/align_file_bench.py [pairs]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

pa.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
divs = (0.01, 0.05, 0.10, 0.15)
with tempfile.TemporaryDirectory() as d:
    src, dst = os.path.join(d, "c4.seq"), os.path.join(d, "out.csv")
    with open(src, "wb") as f:
        for i in range(n):
            a, b = generate_pair(10_000, divs[i % 4], seed=1_000_000 + i)
            f.write(b">" + a + b"\n<" + b + b"\n")
    mb = os.path.getsize(src) / 1e6
    for label, params in (("A*PA2 simple (pa_align_file_params)", pa.AstarPa2Params.simple()), ("full DP with traceback (pa_align_file)", None)):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            got = pa.capi.align_file(src, dst, params=params)
            best = min(best, time.perf_counter() - t)
        assert got == n
        first = open(dst).readline().strip().split(",")[0]
        print(f"{label}: {n} pairs, {mb:.0f} MB in, {os.path.getsize(dst)/1e6:.0f} MB out: {best*1e3:.1f} ms = {n/best:.0f} pairs/s from file to file (first cost {first})", flush=True)
    t = time.perf_counter()
    pairs = pa.read_pairs(src)
    print(f"reading alone (with the copies into Python bytes): {(time.perf_counter()-t)*1e3:.1f} ms")
