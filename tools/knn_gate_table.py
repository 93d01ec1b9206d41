#!/usr/bin/env python3
"""Break-even table of the two k >= 2 neighbour-search kernels at the shapes of one forward (B = 4 frame pairs):
knn_select_kernel (insertion, lists across lanes) against knn_mfma_kernel (+ its tied rows' replay launch), forced through
RPE_KNN_ALGO_INSERT / RPE_KNN_ALGO_MATRIX; "auto" is what the library's gate picks.  Clouds: IDS-range coordinates
(x +-14.5, y +-8.5, z 22...113), prefixes of one random cloud as the pyramid levels are.  Times: HIP events over 50 launches.

    python tools/knn_gate_table.py > profiles/r05_knn_gate_table.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rpeflow_amd.csrc.wrapper import k_nearest_neighbor_ties  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def cloud(B, N):
    xyz = torch.rand(B, 3, N, generator=g) * torch.tensor([29.0, 17.0, 91.0])[None, :, None] + torch.tensor([-14.5, -8.5, 22.0])[None, :, None]
    return xyz.to(dev)


def time_us(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


rows = []
# (what, B, M, Q, k): the decoder's searches per level (RPEFlow_core.py:331, pwc3d_core.py:81, utils.py:148,167) and the pyramid's
for level, n in zip((1, 2, 3, 4, 5), (4096, 2048, 1024, 512, 256)):
    rows.append(("L%d 1in1 / 1in2 k=16" % level, 4, n, n, 16))
    rows.append(("L%d backwarp_3d k=3" % level, 4, n, n, 3))
    if level < 5:
        rows.append(("L%d knn_interpolation k=3" % level, 4, n // 2, n, 3))
for m, q in ((8192, 4096), (4096, 2048), (2048, 1024), (1024, 512), (512, 256)):
    rows.append(("pyramid %d -> %d k=16 (both frames)" % (m, q), 8, m, q, 16))
rows.append(("final up-sampling 4096 -> 8192 k=3", 4, 4096, 8192, 3))
print("%-42s %3s %5s %5s %3s | %9s %9s %9s | %s" % ("search", "B", "M", "Q", "k", "insert us", "matrix us", "auto us", "identical"))
for what, B, M, Q, k in rows:
    pts = cloud(B, max(M, Q))
    inp, qry = pts[:, :, :M].contiguous(), pts[:, :, :Q].contiguous()
    t = {}
    out = {}
    for algo in ("insert", "matrix", "auto"):
        try:
            out[algo] = k_nearest_neighbor_ties(inp, qry, k, algo=algo)
            t[algo] = time_us(lambda: k_nearest_neighbor_ties(inp, qry, k, algo=algo))
        except RuntimeError as err:
            t[algo], out[algo] = float("nan"), None
            print("   (%s: %s)" % (algo, err))
    same = all(o is None or torch.equal(o, out["insert"]) for o in out.values())
    print("%-42s %3d %5d %5d %3d | %9.1f %9.1f %9.1f | %s" % (what, B, M, Q, k, t["insert"], t["matrix"], t["auto"], same))
