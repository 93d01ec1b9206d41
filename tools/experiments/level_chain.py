#!/usr/bin/env python3
"""What one pyramid level's chain is made of: the kernels between the stage stamps of an eager SINGLE-stream forward.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python3 tools/experiments/level_chain.py run /tmp/lc_names.txt
    python3 tools/experiments/level_chain.py report /tmp/lc_names.txt $(find /tmp/lc -name '*kernel_trace.csv') [level]
"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "run":
    from rpeflow_amd import runtime
    runtime.configure()
    import torch
    import bench
    import rpeflow_amd.model as M
    from rpeflow_amd.synthetic import load_seeded_parameters
    dev = torch.device("cuda", 0)
    model = load_seeded_parameters(M.RPEFlow()).to(dev).eval()
    model.overlap_streams = False
    batch = bench.make_batch(4, dev)
    with torch.no_grad():
        for _ in range(3):
            model(batch)
        torch.cuda.synchronize()
        M.TRACE = M.StampTrace(dev)
        model(batch)
        torch.cuda.synchronize()
    open(sys.argv[2], "w").write("\n".join(M.TRACE.names))
else:
    names = open(sys.argv[2]).read().split("\n")
    rows = list(csv.DictReader(open(sys.argv[3])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "stamp_kernel" in r["Kernel_Name"]]
    marks = marks[-len(names):]
    want = sys.argv[4] if len(sys.argv) > 4 else "L4"
    for j in range(1, len(names)):
        if want not in names[j]:
            continue
        seg = rows[marks[j - 1] + 1:marks[j]]
        tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
        print("== %s -> %s: %d kernels, %.1f us of kernel time" % (names[j - 1], names[j], len(seg), tot))
        for r in seg:
            print("   %7.1f us  grid %-8s %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r["Kernel_Name"][:110]))
