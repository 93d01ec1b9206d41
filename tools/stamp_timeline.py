"""Real multi-stream timeline of the graph-replayed forward (rocprofv3 serialises graph branches): GPU wall-clock
stamps dropped by one-thread kernels at stage boundaries of every stream.  Usage: python tools/stamp_timeline.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import rpeflow_amd.model as M

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = M.RPEFlow().to(dev).eval()
batch = bench.make_batch(4, dev)
for _ in range(3):
    model(batch)
torch.cuda.synchronize()
M.TRACE = M.StampTrace(dev)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    model(batch)
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
for name, us in sorted(M.TRACE.read(), key=lambda kv: kv[1]):
    print("%9.1f us  %s" % (us, name))
