"""Summarise tools/pmc_apa2.sh: per kernel (apa2_kernel, trace_kernel) and per workload (the launches are ordered: first the C4 batch,
then the 100 kbp batch; every align() is one launch of each) the counters of the LARGEST launches, per launch."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_apa2"
out = {"kernels": {}}
for f in glob.glob(f"{root}/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if ("apa2_" in r["Name"] and "kernel" in r["Name"]) or "trace_kernel" in r["Name"]:
            out["kernels"].setdefault(r["Name"], {})["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                                                        "max_ns": float(r["MaxNs"]), "percent_of_gpu_time": float(r["Percentage"])}
# per-launch durations from the trace (to tell the two workloads apart)
for f in glob.glob(f"{root}/trace/*kernel_trace.csv"):
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("apa2_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]) or "trace_kernel" in r["Kernel_Name"]:
            durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in durs.items():
        out["kernels"].setdefault(k, {})["launch_ms"] = [round(x, 3) for x in v]
for f in sorted(glob.glob(f"{root}/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if ("apa2_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]) or "trace_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            # the last launches belong to the 100 kbp batch (4 align() calls per batch: warm-up + 3)
            out["kernels"].setdefault(k, {}).setdefault("counters_100kbp_batch", {})[c] = sum(v[-3:]) / len(v[-3:])
            out["kernels"][k].setdefault("counters_c4_batch", {})[c] = sum(v[1:4]) / max(1, len(v[1:4]))
for f in glob.glob(f"{root}/trace.log"):
    out["bench_lines"] = [ln.strip() for ln in open(f, errors="ignore") if "A*PA2-" in ln]
json.dump(out, open(f"{root}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
