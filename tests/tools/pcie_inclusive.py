import sys, time
sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair
pa.require_gpu()
pairs = [generate_pair(100000, 0.05, seed=i + 1) for i in range(4096)]
pa.Batch(pairs[:8]).run()
for trial in range(2):
    t0 = time.perf_counter()
    b = pa.Batch(pairs)
    t1 = time.perf_counter()
    costs, ms = b.run()
    t2 = time.perf_counter()
    b.close()
    cells = 4096 * 1e10
    print(f"4096 x 100 kbp from host buffers: create (gather + H2D + plan) {1e3*(t1-t0):.0f} ms, first pass {1e3*(t2-t1):.0f} ms (kernel {ms:.0f}) "
          f"-> PCIe-inclusive {cells/(t2-t0)/1e12:.1f} TCUPS vs resident {cells/(ms*1e-3)/1e12:.1f}")
