"""C3 through pa_align, both presets, best of 5 (no oracle): python tools/c3_quick.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair
pa.require_gpu()
a, b = generate_pair(100_000, 0.05, seed=1)
for name, mk in (("simple", pa.AstarPa2Params.simple), ("full", pa.AstarPa2Params.full)):
    al = mk().make_aligner(True)
    al.align(a, b)
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); c, g = al.align(a, b); best = min(best, time.perf_counter() - t)
    print(name, c, round(best * 1e3, 2), "ms")
