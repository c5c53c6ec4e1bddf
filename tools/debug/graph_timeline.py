"""Kernel timeline of one HIP-graph replay of the step from a rocprofv3 kernel trace (GPU box):
   rocprofv3 --kernel-trace --output-format csv -d OUT -o p -- python3 tools/try_graph.py B 1 50
   python3 tools/debug/graph_timeline.py OUT/p_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay: walk back from the end until the first lift kernel of that step
names = [r["Kernel_Name"] for r in rows]
end = len(rows)
# (the step's first kernel: the lift's first launch, or -- channel-last features, round 6 -- the lift forward itself)
firsts = [i for i, n in enumerate(names) if any(k in n for k in ("feat_to_channel_last", "lift_prologue_kernel", "lift_operands_kernel"))]
if not firsts:
    firsts = [i for i, n in enumerate(names) if "lift_fwd_kernel" in n or "lift_fwd_coop_kernel" in n]
start = max(firsts)
seg = rows[start:end]
t0 = int(seg[0]["Start_Timestamp"])
busy_end = t0
print(f"{'start_us':>9s} {'dur_us':>8s} {'queue':>6s}  kernel")
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {r['Queue_Id']:>6s}  {r['Kernel_Name'][:70]}")
print("span %.1f us, sum of durations %.1f us" % ((max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e3,
                                                     sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3))
