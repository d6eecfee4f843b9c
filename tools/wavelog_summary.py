"""Summarise a PA_STRIP_WAVELOG dump: wavefronts per (xcc, cu, simd), span per strip, polled chunks."""
import collections
import sys

rows = [l.split() for l in open(sys.argv[1]).read().splitlines()[1:]]
recs = [dict(job=int(r[0]), k=int(r[1]), word0=int(r[2]), xcc=int(r[3]), se=int(r[4]), cu=int(r[5]), simd=int(r[6]), wave=int(r[7]),
             t0=int(r[8]), t1=int(r[9]), polled=int(r[10])) for r in rows]
T0 = min(r["t0"] for r in recs)
T1 = max(r["t1"] for r in recs)
print(f"jobs={len(recs)} span={(T1 - T0) / 100:.1f} us")
per_simd = collections.Counter((r["xcc"], r["se"], r["cu"], r["simd"]) for r in recs)
per_cu = collections.Counter((r["xcc"], r["se"], r["cu"]) for r in recs)
print("waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()), "SIMDs used:", len(per_simd))
print("waves per CU histogram:", sorted(collections.Counter(per_cu.values()).items()), "CUs used:", len(per_cu))
print("waves per XCC:", sorted(collections.Counter(r["xcc"] for r in recs).items()))
dur = sorted((r["t1"] - r["t0"]) / 100 for r in recs)
print(f"strip duration us: min={dur[0]:.0f} median={dur[len(dur) // 2]:.0f} max={dur[-1]:.0f}")
st = sorted((r["t0"] - T0) / 100 for r in recs)
print(f"start offset us: median={st[len(st) // 2]:.0f} p90={st[int(len(st) * .9)]:.0f} max={st[-1]:.0f}")
tall = [r for r in recs if r["k"] > 1]
by_load = collections.defaultdict(list)
for r in tall:
    by_load[per_simd[(r["xcc"], r["se"], r["cu"], r["simd"])]].append((r["t1"] - r["t0"]) / 100)
for k, v in sorted(by_load.items()):
    print(f"tall strips on a SIMD with {k} wave(s): n={len(v)} mean duration={sum(v) / len(v):.0f} us")
pol = [r["polled"] for r in recs if r["word0"] > 0]
if pol:
    print(f"polled chunks per consumer strip: mean={sum(pol) / len(pol):.0f} max={max(pol)}")
