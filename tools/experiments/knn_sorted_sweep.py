"""Does the sweeping matrix kernel get faster when cloud and queries are in Morton order (pass B's tile test then skips most steps)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rpeflow_amd.csrc import wrapper as W
dev = torch.device("cuda", 0)
def timed(f, iters=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
g = torch.Generator().manual_seed(0)
for B, M, Q, k in [(8, 8192, 4096, 16), (4, 4096, 4096, 16), (8, 4096, 8192, 3)]:
    cloud = torch.cat([torch.rand(B, 1, M, generator=g) * 29 - 14.5, torch.rand(B, 1, M, generator=g) * 17 - 8.5, torch.rand(B, 1, M, generator=g) * 91 + 22], 1).to(dev)
    query = cloud[:, :, :Q].contiguous() if Q <= M else torch.cat([torch.rand(B, 1, Q, generator=g) * 29 - 14.5, torch.rand(B, 1, Q, generator=g) * 17 - 8.5, torch.rand(B, 1, Q, generator=g) * 91 + 22], 1).to(dev)
    gi, gq = W.GridSet(cloud.transpose(1, 2)), W.GridSet(query.transpose(1, 2))
    pc = gi.perm[:, :M].long(); pq = gq.perm[:, :Q].long()
    cs = torch.gather(cloud, 2, pc[:, None, :].expand(-1, 3, -1)).contiguous()
    qs = torch.gather(query, 2, pq[:, None, :].expand(-1, 3, -1)).contiguous()
    for ties in ("torch", "index"):
        t0 = timed(lambda: W.k_nearest_neighbor_ties(cloud, query, k, ties=ties, algo="sweep"))
        t1 = timed(lambda: W.k_nearest_neighbor_ties(cs, qs, k, ties=ties, algo="sweep"))
        t2 = timed(lambda: W.k_nearest_neighbor_ties(cs, query, k, ties=ties, algo="sweep"))
        print("B=%d %d -> %d k=%d ties=%s: sweep on the arrays as given %.1f us | cloud AND queries Morton-sorted %.1f us | cloud only %.1f us" % (B, M, Q, k, ties, t0, t1, t2), flush=True)
