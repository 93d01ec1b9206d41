"""Per-launch means of rocprofv3 --pmc passes for one kernel -> JSON (the format of profiles/*_pmc.json).
Usage: python tools/pmc_summary.py <kernel-name-substring> <out.json> <workload note> <dir-with-counter_collection-csvs>..."""
import csv
import glob
import json
import os
import sys

needle, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
counters, durations, name = {}, [], None
for d in sys.argv[4:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(path)):
            if needle not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"]
            per_dispatch.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                per_dispatch[r["Dispatch_Id"]]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows = list(per_dispatch.values())[1:]  # the first launch carries cold caches / lazy module load
        for row in rows:
            for k, v in row.items():
                if k == "_dur":
                    durations.append(v)
                else:
                    counters.setdefault(k, []).append(v)
mean = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
derived = {}
if "FETCH_SIZE" in mean:
    derived["fetch_bytes_corrected"] = mean["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in mean:
    derived["write_bytes"] = mean["WRITE_SIZE"] * 1024
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    derived["hbm_traffic_bytes"] = derived["fetch_bytes_corrected"] + derived["write_bytes"]
if "TCC_HIT_sum" in mean:
    derived["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
derived["note"] = ("FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950 for 16 B/lane streams (MI355X_MICROARCH.md, HBM): "
                   "doubled; for narrower gathers the factor is uncalibrated (upper bound).  WRITE_SIZE exact.")
json.dump({"kernel": name, "workload": note, "counters_mean_per_launch": mean, "durations_us_under_pmc": sorted(durations),
           "derived": derived}, open(out, "w"), indent=1)
print(out, name, {k: round(v, 1) for k, v in mean.items()})
