"""C3: single pair, A*PA2-simple and A*PA2-full (band doubling + sparse blocks + DT trace), HIP engine vs the engine
over the CPU oracle kernels (same band decisions => same computed_lanes)."""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa
import oracle
from astar_pairwise_aligner_amd.generate import generate_pair

pa.require_gpu()
for n, e in [(100_000, 0.05), (100_000, 0.01), (1_000_000, 0.05)]:
    a, b = generate_pair(n, e, seed=1)
    for preset, gp, op in (("simple", pa.AstarPa2Params.simple(), oracle.params_simple()),
                           ("full", pa.AstarPa2Params.full(), oracle.params_full())):
        al = gp.make_aligner(True)
        al.align(a[:2000], b[:2000])
        t = time.perf_counter(); cost, cigar, st = al.align_with_stats(a, b); tg = time.perf_counter() - t
        t = time.perf_counter(); c2, cg2, st2 = oracle.cpu_align(a, b, op); tc = time.perf_counter() - t
        assert (cost, cigar) == (c2, cg2)
        cells_eq = len(a) * len(b)
        print(f"{preset:6s} n={n} e={e} cost={cost} tries={st['f_max_tries']} blocks={st['num_blocks']} lanes={st['computed_lanes']} "
              f"GPU engine {tg*1e3:.1f} ms (t_compute {st['t_compute']*1e3:.1f}, precomp {st['t_precomp']*1e3:.1f}, dt {st['t_dt']*1e3:.1f}, fill {st['t_fill']*1e3:.1f}) | "
              f"CPU engine {tc*1e3:.1f} ms (t_compute {st2['t_compute']*1e3:.1f}) | equivalent GCUPS gpu={cells_eq/tg/1e9:.0f} cpu={cells_eq/tc/1e9:.0f}", flush=True)
