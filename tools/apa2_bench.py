"""Timing of the batched A*PA2 path (pa_batch_create_params): C4 (10 000 x 10 kbp, 1/5/10/15 %) and batches of 100 kbp pairs.
python tools/apa2_bench.py [simple|full] [c4_pairs] [c3_pairs ...]"""
import sys
import time

sys.path.insert(0, ".")
import astar_pairwise_aligner_amd as pa  # noqa: E402
from astar_pairwise_aligner_amd.generate import generate_pair  # noqa: E402


PRESET = "simple"


def params():
    return pa.AstarPa2Params.full() if PRESET == "full" else pa.AstarPa2Params.simple()


def bench(pairs, label, reps=3):
    t = time.perf_counter()
    bt = pa.Batch(pairs, params=params())
    t_create = time.perf_counter() - t
    bt.align()
    best = (1e9, 0, 0, 0)
    for _ in range(reps):
        t = time.perf_counter()
        costs, cigars, f_ms, t_ms = bt.align()
        dt = time.perf_counter() - t
        if dt < best[0]:
            best = (dt, f_ms, t_ms, bt.last_c_abi_ms)
    st = bt.pair_stats()
    strip_instr = bt.shape()["valu_instructions"]
    lanes = sum(s["computed_lanes"] for s in st)
    print(f"{label}: {len(pairs)} pairs  create {t_create*1e3:.1f} ms  align {best[0]*1e3:.2f} ms (c abi {best[3]:.2f})  forward {best[1]:.2f} ms  trace {best[2]:.2f} ms  "
          f"=> {len(pairs)/best[0]:.0f} pairs/s ({len(pairs)/(best[3]*1e-3):.0f} at the C ABI)  computed lanes {lanes:.3e} = {lanes*256*64/(best[1]*1e-3)/1e9:.0f} band-GCUPS  "
          f"strip VALU instructions (model) {strip_instr:.3e} = {strip_instr/(best[1]*1e-3)/1e9:.0f} G/s  fallbacks {bt.trace_fallbacks()}  tries {sum(s['f_max_tries'] for s in st)/len(st):.2f}", flush=True)
    print(f"   half-wave blocks of the last forward pass: {bt.rdv_stats()}", flush=True)
    if PRESET == "full":
        fi = bt.full_info()
        print(f"   full: match building {abs(fi['build_ms']):.1f} ms ({'GPU' if fi['build_ms'] < 0 else 'host threads'}) for {fi['matches']:.0f} matches; h probes {fi['probes']:.3e}, load rounds {fi['rounds']:.3e}; wavefront-ms by phase {({k: round(v, 1) for k, v in fi['phase_wave_ms'].items()})}", flush=True)
    bt.close()
    t = time.perf_counter()
    pa.Batch(pairs, params=params()).close()
    print(f"   the same batch created again (large device buffers come from the library's cache): {(time.perf_counter() - t)*1e3:.1f} ms  {pa.capi.alloc_cache_stats()}", flush=True)
    return costs


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("simple", "full"):
        PRESET = sys.argv.pop(1)
    c4n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    divs = (0.01, 0.05, 0.10, 0.15)
    c4 = [generate_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(c4n)]
    bench(c4, f"C4 A*PA2-{PRESET}")
    for n3 in [int(x) for x in sys.argv[2:]] or [512]:
        c3 = [generate_pair(100_000, 0.05, seed=3_000_000 + i) for i in range(n3)]
        bench(c3, f"100 kbp @ 5 % A*PA2-{PRESET}")
