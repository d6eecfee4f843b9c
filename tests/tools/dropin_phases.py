"""Where the time of ONE short pair through pa_align(simple) goes (the drop-in loop): wall time per call over 64 distinct 10 kbp
pairs at 5 %, then three calls with PA_SWEEP_TIMING=1 (set in the environment by the caller) printing every pass.
python tests/tools/dropin_phases.py [n] [e]"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

pa.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
e = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
pairs = [generate_pair(n, e, seed=900 + i) for i in range(64)]
al = pa.AstarPa2Params.simple().make_aligner(True)
for a, b in pairs[:8]:
    al.align(a, b)
ts = []
for a, b in pairs:
    t = time.perf_counter()
    al.align(a, b)
    ts.append(time.perf_counter() - t)
ts.sort()
print(f"n={n} e={e}: median {ts[32]*1e3:.3f} ms  min {ts[0]*1e3:.3f}  p90 {ts[57]*1e3:.3f}  => {1/ (sum(ts)/64):.0f} pairs/s", flush=True)
