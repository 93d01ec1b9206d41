#!/usr/bin/env python3
"""k_nearest_neighbor: the sweeping kernels against the grid kernel (csrc/knn_grid.h) at the forward's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpeflow_amd.csrc import wrapper as W

dev = torch.device("cuda", 0)


def timed(f, iters=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


g = torch.Generator().manual_seed(0)
for B, M, Q, D, k, self_q in [(8, 8192, 4096, 3, 16, False), (8, 8192, 8192, 3, 16, True), (4, 4096, 4096, 3, 16, True), (8, 4096, 8192, 3, 3, False),
                              (8, 2048, 4096, 3, 3, False), (4, 2048, 2048, 3, 16, True), (4, 1024, 1024, 3, 16, True), (8, 4096, 2048, 3, 16, False)]:
    x = torch.rand(B, 1, M, generator=g) * 29 - 14.5
    y = torch.rand(B, 1, M, generator=g) * 17 - 8.5
    z = torch.rand(B, 1, M, generator=g) * 91 + 22
    cloud = torch.cat([x, y, z], 1).to(dev)
    if self_q:
        query = cloud
    elif Q <= M:
        query = cloud[:, :, :Q]
    else:
        query = (torch.cat([torch.rand(B, 1, Q, generator=g) * 29 - 14.5, torch.rand(B, 1, Q, generator=g) * 17 - 8.5, torch.rand(B, 1, Q, generator=g) * 91 + 22], 1)).to(dev)
    a = W.k_nearest_neighbor_ties(cloud, query, k, algo="sweep")
    b = W.k_nearest_neighbor_ties(cloud, query, k, algo="grid")
    same = torch.equal(a, b)
    t_sweep = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, algo="sweep"))
    t_grid = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, algo="grid"))
    gi = W.GridSet(cloud.transpose(1, 2))
    gq = gi if self_q else W.GridSet(query.transpose(1, 2))
    t_build = timed(lambda: W.GridSet(cloud.transpose(1, 2)))
    t_search = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, input_grid=gi, query_grid=gq))
    t_index = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, input_grid=gi, query_grid=gq, ties="index"))
    print("B=%d %5d -> %5d k=%2d: sweep %6.1f us | grid %6.1f us (build %5.1f, search %6.1f, no tie handling %6.1f) | same %s" % (
        B, M, Q, k, t_sweep, t_grid, t_build, t_search, t_index, same), flush=True)

# where the grid kernel's time goes (rpe_knn_grid_set_stats)
from rpeflow_amd import _lib
names = ["waves", "steps A0", "steps A'", "steps B", "candidates", "serial fallbacks", "tied queries", "cyc bounds", "cyc A0", "cyc A'", "cyc B", "cyc rank", "cyc finish"]
for B, M, Q, k in [(8, 8192, 4096, 16), (4, 4096, 4096, 16), (8, 4096, 8192, 3)]:
    x = torch.rand(B, 1, M, generator=g) * 29 - 14.5; y = torch.rand(B, 1, M, generator=g) * 17 - 8.5; z = torch.rand(B, 1, M, generator=g) * 91 + 22
    cloud = torch.cat([x, y, z], 1).to(dev)
    query = cloud[:, :, :Q] if Q <= M else torch.cat([torch.rand(B, 1, Q, generator=g) * 29 - 14.5, torch.rand(B, 1, Q, generator=g) * 17 - 8.5, torch.rand(B, 1, Q, generator=g) * 91 + 22], 1).to(dev)
    gi = W.GridSet(cloud.transpose(1, 2)); gq = W.GridSet(query.transpose(1, 2))
    st = torch.zeros(16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().rpe_knn_grid_set_stats(st.data_ptr()), "stats")
    W.k_nearest_neighbor_ties(cloud, query, k, input_grid=gi, query_grid=gq)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().rpe_knn_grid_set_stats(None), "stats")
    v = st.cpu().tolist(); w = max(1, v[0])
    print("B=%d %d -> %d k=%d, %d steps in the cloud; per wave:" % (B, M, Q, k, (M + 63) // 64), ", ".join("%s %.1f" % (n, v[i] / w) for i, n in enumerate(names) if i > 0), flush=True)
