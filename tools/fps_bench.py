#!/usr/bin/env python3
"""furthest_point_sampling 8 x (8192 -> 4096): the pruned kernel against its paired form (two samples per round where provable)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpeflow_amd import _lib
from rpeflow_amd.synthetic import frame_pair
dev = torch.device("cuda", 0)

def run(t, S, algo):
    idx = torch.empty((t.shape[0], S), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().rpe_fps_algo(t.data_ptr(), *t.stride(), t.shape[0], t.shape[1], S, idx.data_ptr(), algo, torch.cuda.current_stream().cuda_stream), "fps")
    return idx

def timed(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3

g = torch.Generator().manual_seed(0)
uniform = (torch.rand(8, 8192, 3, generator=g) * 30).to(dev)
pcs = np.stack([frame_pair(1000 + i)["pcs"] for i in range(4)])  # [4,6,8192]
clouds = np.concatenate([pcs[:, :3], pcs[:, 3:]], 0).transpose(0, 2, 1)  # perspective clouds (before the IDS transform)
f, cx, cy = 1050.0, 479.5, 271.5
z = clouds[..., 2]
ids = np.stack([(cx + f / z * clouds[..., 0]) * (29 / 959) - 14.5, (cy + f / z * clouds[..., 1]) * (17 / 543) - 8.5, (f * np.log(z) + 1) * (29 / 959)], -1).astype(np.float32)
benched = torch.from_numpy(np.ascontiguousarray(ids)).to(dev)
for name, t in (("uniform cube", uniform), ("IDS-like benched clouds", benched)):
    a, b = run(t, 4096, 1), run(t, 4096, 2)
    print(name, "| indices equal (plain / pruned):", bool(torch.equal(a, b)),
          "| plain %.1f us (%.3f us/sample), pruned %.1f us (%.3f us/sample)" % (
              timed(lambda: run(t, 4096, 1)), timed(lambda: run(t, 4096, 1)) / 4095, timed(lambda: run(t, 4096, 2)), timed(lambda: run(t, 4096, 2)) / 4095), flush=True)
