"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max duration and resources.
Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(sgpr_count), "
    "max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs", "SGPRs", "LDS", "Scratch", "GridX", "GridY", "WorkgroupX"]
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
out.writerow(hdr)
for r in rows:
    out.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / total, 2)] + list(r[6:]))
