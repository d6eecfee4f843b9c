"""Phase clocks of the GPU's GCSH match builder (csrc/gcsh_build_kernel.hpp) for one lone wavefront: PA_BUILD_CLOCKS=1 python tools/gcsh_build_clocks.py"""
import os
import sys

sys.path.insert(0, ".")
os.environ["PA_BUILD_CLOCKS"] = "1"
from astar_pairwise_aligner_amd import capi  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

for n, e in [(100_000, 0.05), (10_000, 0.01), (10_000, 0.05), (10_000, 0.15)]:
    a, b = generate_pair(n, e, seed=7)
    for _ in range(2):
        capi.gcsh_matches(a, b, 12, 14)
