import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
print("default preferred:", torch.backends.cuda.preferred_blas_library())
def timed(f, iters=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
shapes = [(4, 96, 510, 34560), (4, 255, 96, 34560), (8, 64, 340, 8640), (8, 510, 192, 135), (4, 215, 81, 8640), (4, 69, 113, 34560), (4, 256, 144, 34560)]
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    out = []
    for n, c, k, P in shapes:
        x = torch.randn(n, c, P, device=dev); w = torch.randn(k, c, device=dev)
        g = torch.cuda.CUDAGraph()
        wb = w.unsqueeze(0).expand(n, -1, -1)
        torch.bmm(wb, x); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10): y = torch.bmm(wb, x)
        out.append(round(timed(g.replay, 10) / 10, 1))
    print(lib, out)
