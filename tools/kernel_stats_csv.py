#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output directory -> one small CSV (kernel, calls, avg_us, total_ms, percent).
usage: tools/kernel_stats_csv.py <rocprof output dir> <out.csv> [header comment]"""
import csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(src, "*", "*kernel_stats.csv")) + glob.glob(os.path.join(src, "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        rows.append((r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
rows.sort(key=lambda r: -r[3])
with open(dst, "w") as fh:
    if len(sys.argv) > 3:
        fh.write("# " + sys.argv[3] + "\n")
    fh.write("kernel,calls,avg_us,total_ms,percent\n")
    for name, calls, avg, tot, pct in rows:
        fh.write(f"\"{name[:120]}\",{calls},{avg:.2f},{tot:.3f},{pct:.2f}\n")
print(open(dst).read()[:2500])
