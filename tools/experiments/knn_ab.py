#!/usr/bin/env python3
"""A/B of two builds of the library on the neighbour searches of the bench (one process per build, same box):
    python tools/experiments/knn_ab.py tools/experiments/librpe_probe_old.so rpeflow_amd/csrc/librpeflow_hip.so"""
import os, subprocess, sys
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--child"):
    for rep in range(2):
        for lib in sys.argv[1:]:
            out = subprocess.run([sys.executable, __file__, "--child"], env={**os.environ, "RPE_HIP_LIB": os.path.abspath(lib)}, capture_output=True, text=True)
            print("%-40s %s" % (os.path.basename(lib), out.stdout.strip().replace("\n", " | ")), flush=True)
            if out.returncode:
                print(out.stderr[-2000:])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpeflow_amd.csrc import wrapper as W
dev = torch.device("cuda", 0)
def timed(f, iters=40):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
g = torch.Generator().manual_seed(0)
for B, M, Q, k in [(8, 8192, 4096, 16), (4, 4096, 4096, 16), (8, 4096, 8192, 3)]:
    cloud = torch.cat([torch.rand(B, 1, M, generator=g) * 29 - 14.5, torch.rand(B, 1, M, generator=g) * 17 - 8.5, torch.rand(B, 1, M, generator=g) * 91 + 22], 1).to(dev)
    query = cloud[:, :, :Q] if Q <= M else torch.cat([torch.rand(B, 1, Q, generator=g) * 29 - 14.5, torch.rand(B, 1, Q, generator=g) * 17 - 8.5, torch.rand(B, 1, Q, generator=g) * 91 + 22], 1).to(dev)
    t = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, algo="sweep"))
    ti = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, algo="sweep", ties="index"))
    print("%dx%d->%d k%d: %.1f (index ties %.1f)" % (B, M, Q, k, t, ti))
