"""How the HIP engine's per-block launch cost amortises with wider blocks (block_width is an A*PA2 parameter; the
reference's presets use 256).  Costs must not change; CIGARs are compared with the 256-wide run."""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
for n, e in [(100_000, 0.05), (100_000, 0.01)]:
    a, b = generate_pair(n, e, seed=1)
    for preset in ("simple", "full"):
        ref = None
        for bw in (256, 512, 1024, 2048, 4096):
            p = getattr(pa.AstarPa2Params, preset)()
            p.block_width = bw
            al = p.make_aligner(True)
            al.align(a[:3000], b[:3000])
            best = 1e9
            for _ in range(2):
                t = time.perf_counter()
                cost, cigar, st = al.align_with_stats(a, b)
                best = min(best, time.perf_counter() - t)
            ref = ref or (cost, cigar)
            print(f"{preset:6s} n={n} e={e} block_width={bw:5d} {best*1e3:7.1f} ms blocks={st['num_blocks']} lanes={st['computed_lanes']} "
                  f"cost={cost} same_cost={cost == ref[0]} same_cigar={cigar == ref[1]}", flush=True)
