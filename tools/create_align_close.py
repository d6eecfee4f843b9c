"""create -> align -> close of the C4 batch, three times, each phase timed (what the sharded legs pay per chunk): python tools/create_align_close.py [simple|full|trace]"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "simple"
divs = (0.01, 0.05, 0.10, 0.15)
pairs = [generate_pair(10_000, divs[i % 4], seed=1_000_000 + i) for i in range(10_000)]
for rep in range(4):
    t0 = time.perf_counter()
    if mode == "trace":
        b = pa.Batch(pairs, trace=True)
    else:
        b = pa.Batch(pairs, params=pa.AstarPa2Params.full() if mode == "full" else pa.AstarPa2Params.simple())
    t1 = time.perf_counter()
    b.align()
    t2 = time.perf_counter()
    b.close()
    t3 = time.perf_counter()
    print(f"{mode} rep {rep}: create {1e3*(t1-t0):.1f}  align {1e3*(t2-t1):.1f} (c abi {b.last_c_abi_ms:.1f})  close {1e3*(t3-t2):.1f}  total {1e3*(t3-t0):.1f} ms", flush=True)
