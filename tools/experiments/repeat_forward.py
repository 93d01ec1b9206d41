#!/usr/bin/env python3
"""Is the forward repeatable inside ONE process?  N eager forwards (multi-stream, then single-stream) and N graph replays on the
benched batch: EPE deltas against the reference golden and the largest difference from the first output of each mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rpeflow_amd import runtime
runtime.configure()
import numpy as np, torch
import bench
from rpeflow_amd.model import RPEFlow
from rpeflow_amd.synthetic import load_seeded_parameters
from rpeflow_amd.evaluate import GraphedForward
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if os.environ.get("DET"):
    torch.backends.cudnn.deterministic = True
    print("torch.backends.cudnn.deterministic = True")
dev = torch.device("cuda", 0)
g = np.load(os.path.join(os.path.dirname(bench.__file__), "tests", "golden", "model_bench_b4_544x960.npz"))
model = load_seeded_parameters(RPEFlow()).to(dev).eval()
batch = bench.make_batch(4, dev, first_seed=1000)
def summarize(name, outs):
    f2 = [o[0] for o in outs]; f3 = [o[1] for o in outs]
    d2 = [float((a - f2[0]).abs().max()) for a in f2]; d3 = [float((a - f3[0]).abs().max()) for a in f3]
    epes = [bench.golden_epe_delta({"flow_2d": a, "flow_3d": b}, batch, g) for a, b in outs]
    e2 = sorted(set(round(e["epe2d"], 9) for e in epes))
    print("%-28s distinct outputs: %d | max |d flow_2d| vs first %.3g, flow_3d %.3g | dEPE2D values %s" % (
        name, len(set((round(a, 9), round(b, 9)) for a, b in zip(d2, d3))), max(d2), max(d3), ["%.3g" % x for x in e2]), flush=True)
    return d2
with torch.no_grad():
    for overlap in (True, False):
        model.overlap_streams = overlap
        outs = []
        for i in range(N):
            o = model(batch); torch.cuda.synchronize()
            outs.append((o["flow_2d"].float().clone(), o["flow_3d"].float().clone()))
        d2 = summarize("eager, overlap_streams=%s" % overlap, outs)
        print("   per-iteration max|d flow_2d|:", ["%.2g" % x for x in d2])
    model.overlap_streams = True
    fwd = GraphedForward(model, warmup=0)
    outs = []
    for i in range(N):
        o = fwd(batch, batch); torch.cuda.synchronize()
        outs.append((o["flow_2d"].float().clone(), o["flow_3d"].float().clone()))
    d2 = summarize("graph replay", outs)
    rep = fwd.entries[fwd._key(batch)]["graph"].replay
    for _ in range(10): rep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rep()
    torch.cuda.synchronize(); print("ms per replay: %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
    print("   per-iteration max|d flow_2d|:", ["%.2g" % x for x in d2])
