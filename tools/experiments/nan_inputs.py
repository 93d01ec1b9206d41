#!/usr/bin/env python3
"""What the operators do with NaN coordinates (a corrupt sample inside a batch): every index they return has to stay inside its cloud
(the gathers behind them do not check), the other samples of the batch must not change, and the corrupt sample's flow should come out
NaN (the reference's evaluation masks NaN predictions, eval_withocc.py:86-87).

    python tools/experiments/nan_inputs.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
import rpeflow_amd.csrc as ops  # noqa: E402
from rpeflow_amd.model import RPEFlow  # noqa: E402
from rpeflow_amd.synthetic import load_seeded_parameters  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
bad = 0


def check(name, idx, M):
    global bad
    lo, hi = int(idx.min()), int(idx.max())
    ok = 0 <= lo and hi < M
    bad += not ok
    print("%-58s indices in [%d, %d] of [0, %d): %s" % (name, lo, hi, M, "ok" if ok else "OUT OF RANGE"))


for B, M, Q, k, D in ((4, 8192, 4096, 16, 3), (4, 2048, 2048, 16, 3), (4, 512, 256, 16, 3), (4, 4096, 8192, 3, 3), (4, 300, 200, 1, 3), (8, 4096, 34560, 1, 2), (2, 100, 64, 63, 3)):
    for where in ("cloud", "query", "both", "all of one sample"):
        x, q = torch.randn(B, M, D, device=dev), torch.randn(B, Q, D, device=dev)
        if where in ("cloud", "both"):
            x[:, ::7] = float("nan")
        if where in ("query", "both"):
            q[:, ::5, 0] = float("nan")
        if where == "all of one sample":
            x[1] = float("nan")
            q[1] = float("nan")
        idx = ops.k_nearest_neighbor(x, q, k)
        torch.cuda.synchronize()
        check("knn B=%d M=%d Q=%d k=%d D=%d, NaN in %s" % (B, M, Q, k, D, where), idx, M)

for N, S in ((8192, 4096), (2048, 512)):
    for where in ("some points", "all of one sample", "first point"):
        x = torch.randn(4, N, 3, device=dev)
        if where == "some points":
            x[:, ::9] = float("nan")
        elif where == "first point":
            x[:, 0] = float("nan")
        else:
            x[2] = float("nan")
        idx = ops.furthest_point_sampling(x, S)
        torch.cuda.synchronize()
        check("fps N=%d S=%d, NaN in %s" % (N, S, where), idx, N)

# the whole forward: sample 1 of 4 gets a NaN point cloud
torch.set_grad_enabled(False)
model = load_seeded_parameters(RPEFlow()).to(dev).eval()
batch = bench.make_batch(4, dev)
clean = {k: v.clone() for k, v in model(batch).items() if k in ("flow_2d", "flow_3d")}
for what in ("one NaN point", "whole cloud NaN"):
    b2 = {k: v.clone() for k, v in batch.items()}
    if what == "one NaN point":
        b2["pcs"][1, 0, 17] = float("nan")
    else:
        b2["pcs"][1, :3] = float("nan")
    out = model(b2)
    torch.cuda.synchronize()
    for key in ("flow_2d", "flow_3d"):
        others = [i for i in range(4) if i != 1]
        same = all(torch.equal(out[key][i], clean[key][i]) for i in others)
        frac = float(torch.isnan(out[key][1]).float().mean())
        print("forward, %s in sample 1: %s of the other samples %s; sample 1: %.1f %% NaN" % (what, key, "unchanged" if same else "CHANGED", 100 * frac))
        bad += not same
print("problems:", bad)
sys.exit(1 if bad else 0)
