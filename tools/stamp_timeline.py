"""Real multi-stream timeline of the graph-replayed forward (rocprofv3 serialises graph branches): GPU wall-clock
stamps dropped by one-thread kernels at stage boundaries of every stream.
Usage: python tools/stamp_timeline.py [--no-ahead]   (default: the harness's schedule, sampling one batch ahead)"""
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
order = model.sample_order(batch)
with torch.cuda.graph(graph):
    if "--no-ahead" in sys.argv:
        model(batch)
    else:
        model.forward_ahead(batch, order, batch)
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
stamps = M.TRACE.read()
t0 = min(us for _, us in stamps)  # read() counts from the first stamp ISSUED, which need not be the first to run
for name, us in sorted(stamps, key=lambda kv: kv[1]):
    print("%9.1f us  %s" % (us - t0, name))
