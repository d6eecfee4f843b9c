"""C4 through the per-pair A*PA2-simple engine (what a loop over `astarpa2_simple` costs), on a sample of the C4 pairs:
HIP engine vs the same engine over the CPU oracle kernels vs the batched GPU paths."""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
import oracle
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
divs = (0.01, 0.05, 0.10, 0.15)
pairs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(200)]
al = pa.AstarPa2Params.simple().make_aligner(True)
al.align(*pairs[0])
t = time.perf_counter()
got = [al.align(a, b) for a, b in pairs]
tg = time.perf_counter() - t
t = time.perf_counter()
want = [oracle.cpu_align(a, b, oracle.params_simple())[:2] for a, b in pairs]
tc = time.perf_counter() - t
assert got == want
t = time.perf_counter()
batch = pa.align_batch(pairs)
tb = time.perf_counter() - t
assert [c for c, _ in batch] == [c for c, _ in got]
print(f"C4 sample, {len(pairs)} pairs of 10 kbp, A*PA2-simple with traceback, one pair at a time: HIP engine {len(pairs)/tg:.0f} pairs/s, "
      f"engine over CPU kernels (1 core) {len(pairs)/tc:.0f} pairs/s; the same pairs through pa_batch_align (create + align): {len(pairs)/tb:.0f} pairs/s")
