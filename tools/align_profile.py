"""Where the time of one pa_batch_align call goes (PA_ALIGN_PROFILE marks on stderr) for the C4 batch; PA_ALIGN_CHUNKS to vary the chunks.
python tools/align_profile.py [simple|full] [pairs]"""
import os
import sys
import time

sys.path.insert(0, ".")
os.environ["PA_ALIGN_PROFILE"] = "1"
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "simple"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
divs = (0.01, 0.05, 0.10, 0.15)
pairs = [generate_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(n)]
bt = pa.Batch(pairs, params=pa.AstarPa2Params.full() if preset == "full" else pa.AstarPa2Params.simple())
for rep in range(4):
    t = time.perf_counter()
    costs, cigars, f_ms, t_ms = bt.align()
    print(f"rep {rep}: align {1e3 * (time.perf_counter() - t):.2f} ms  c abi {bt.last_c_abi_ms:.2f} ms  forward {f_ms:.2f}  trace {t_ms:.2f}", file=sys.stderr, flush=True)
bt.close()
