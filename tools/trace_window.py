"""Sum kernel durations of a rocprofv3 kernel trace between the two marker kernels (the stamp kernel of rpe_clock_stamp) of tools/prof_forward.py.
Usage: python tools/trace_window.py <kernel_trace.csv> <steps>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "stamp" in r["Kernel_Name"]]
lo, hi = marks[-2], marks[-1]
win = rows[lo + 1:hi]
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[r["Kernel_Name"]]
    a[0] += d
    a[1] += 1
tot = sum(a[0] for a in agg.values())
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["End_Timestamp"])
print(f"window: {len(win)} kernels, kernel time {tot / 1e6 / steps:.3f} ms/step, wall {span / 1e6 / steps:.3f} ms/step, {len(win) / steps:.0f} launches/step")
for name, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(__import__("os").environ.get("TRACE_TOP", 45))]:
    print(f"{d / 1e6 / steps:8.3f} ms/step {n / steps:7.1f}/step {d / n / 1e3:9.1f} us  {name[:120]}")

if len(sys.argv) > 3:  # per-launch dump of the last step in the window: start offset, duration, queue, grid, name
    per = len(win) // steps
    last = win[-per:]
    t0 = int(last[0]["Start_Timestamp"])
    with open(sys.argv[3], "w") as f:
        f.write("start_us,dur_us,queue,grid,wg,name\n")
        for r in last:
            f.write("%.1f,%.1f,%s,%s,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                              r.get("Queue_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                                              r["Kernel_Name"][:100].replace(",", ";")))
