"""Does the bit-sliced kernel park its wavefronts because strips wait for the strip above (polled boundary rows), or for something else?
One-strip groups (b of 3000 rows: no boundary rows at all) against the bench's 32-strip groups, same number of (group, strip) jobs.
Run under rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY (tools/_waitprobe.sh).  python tools/slice_wait_probe.py one|chain"""
import os
import sys

sys.path.insert(0, ".")
os.environ["PA_SLICE"] = "50"
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair, random_sequence  # noqa: E402

pa.require_gpu()
mode = sys.argv[1] if len(sys.argv) > 1 else "one"
if mode == "one":
    base = [(random_sequence(100_000, seed=s + 1), random_sequence(3_000, seed=1000 + s)) for s in range(32)]
    pairs = [base[i % 32] for i in range(65536)]  # 2048 groups x 1 strip
else:
    base = [generate_pair(100_000, 0.05, seed=s + 1) for s in range(32)]
    pairs = [base[i % 32] for i in range(2048)]  # 64 groups x 32 strips
b = pa.Batch(pairs)
print(mode, b.shape())
for _ in range(3):
    costs, ms = b.run()
    print(mode, "kernel ms", round(ms, 3), "cost0", int(costs[0]))
b.close()
