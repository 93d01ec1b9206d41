"""Full forward only (or, with "hotpath", the hot-path operator sequence of rpeflow_amd/hotpath.py), for rocprofv3 --kernel-trace:
warm-up (MIOpen find), a marker kernel, N steady-state steps, a marker.  tools/trace_window.py sums kernels between the markers:
what it lists is per STEP -- model construction, parameter fills and the warm-up are outside the window.
Usage: python tools/prof_forward.py [steps] [graph] [hotpath]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
from rpeflow_amd.model import RPEFlow

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
from rpeflow_amd.synthetic import load_seeded_parameters
if "hotpath" in sys.argv:
    from rpeflow_amd.hotpath import HotPathWorkload
    model = HotPathWorkload(batch=4, height=544, width=960, n_points=8192, device=dev, seed=1000)
    batch = None
    call = lambda: model()
else:
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batch = bench.make_batch(4, dev)
    call = lambda: model(batch)
for _ in range(3):
    call()
torch.cuda.synchronize()
from rpeflow_amd import _lib
probe = torch.zeros(2, dtype=torch.int64, device=dev)
mark = lambda: _lib.lib().rpe_clock_stamp(probe.data_ptr(), torch.cuda.current_stream().cuda_stream)  # marker kernel
step = call
if "graph" in sys.argv:  # replay the forward as one HIP graph, as bench.py does
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    graph.replay()
    torch.cuda.synchronize()
    step = graph.replay
mark()
for _ in range(steps):
    step()
mark()
torch.cuda.synchronize()
print("done", steps)
