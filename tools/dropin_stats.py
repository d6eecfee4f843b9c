"""Per-call statistics of the drop-in loop's pairs (10 kbp, given divergence) through astarpa2_simple with traceback: wall time of the call next to
the engine's own clocks (band search, DT-trace, block re-fills).  python tools/dropin_stats.py [e]"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402

pa.require_gpu()
e = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
al = pa.AstarPa2Params.simple().make_aligner(True)
pairs = [generate_pair(10_000, e, seed=900 + i) for i in range(24)]
for a, b in pairs[:4]:
    al.align(a, b)
rows = []
for a, b in pairs:
    t = time.perf_counter()
    cost, cigar, st = al.align_with_stats(a, b)
    rows.append((time.perf_counter() - t, st))
keys = [k for k in rows[0][1] if k.startswith("t_") or k in ("f_max_tries", "dt_trace_tries", "dt_trace_success", "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback")]
print("e", e, "call ms median", round(sorted(r[0] for r in rows)[len(rows) // 2] * 1e3, 3))
for k in keys:
    vals = sorted(float(r[1][k]) for r in rows)
    print(f"  {k:22s} median {vals[len(vals) // 2] * (1e3 if k.startswith('t_') else 1):.3f}")
