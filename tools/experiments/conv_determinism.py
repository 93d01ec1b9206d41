#!/usr/bin/env python3
"""Which convolution shapes of the forward are executed non-deterministically (atomic split-K solvers) by MIOpen's default
choice, and what does torch.backends.cudnn.deterministic cost for each?  Shapes from profiles/r03_solver_lottery.json."""
import json, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
dev = torch.device("cuda", 0)
cmds = json.load(open(os.path.join(os.path.dirname(__file__), "..", "..", "profiles", "r03_solver_lottery.json")))["choices_run0"]
def timed(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
bad = []
for cmd, alg in cmds.items():
    a = dict(re.findall(r"-(\w) (\d+)", cmd))
    n, c, H, W, k, y, x, p, q, u, v, l, j, g = (int(a[t]) for t in "ncHWkyxpquvljg")
    torch.manual_seed(0)
    inp = torch.randn(n, c, H, W, device=dev); w = torch.randn(k, c // g, y, x, device=dev) / (c * y * x) ** 0.5
    f = lambda: F.conv2d(inp, w, None, (u, v), (p, q), (l, j), g)
    outs = [f().clone() for _ in range(12)]
    nondet = any(not torch.equal(o, outs[0]) for o in outs[1:])
    if nondet:
        t_def = timed(f)
        with torch.backends.cudnn.flags(deterministic=True):
            again = [f().clone() for _ in range(6)]
            still = any(not torch.equal(o, again[0]) for o in again[1:])
            t_det = timed(f)
        md = max(float((o - outs[0]).abs().max()) for o in outs[1:])
        bad.append(cmd)
        print("NONDET %s (%s): max diff %.3g | default %.1f us, deterministic flag %.1f us (still nondet: %s)" % (cmd[5:90], alg[0][:30], md, t_def, t_det, still), flush=True)
print("%d of %d shapes non-deterministic" % (len(bad), len(cmds)))

# the same shapes as im2col (F.unfold) + one rocBLAS strided-batched GEMM: deterministic?  how fast?
torch.backends.cuda.preferred_blas_library("cublas")
for cmd in bad:
    a = dict(re.findall(r"-(\w) (\d+)", cmd))
    n, c, H, W, k, y, x, p, q, u, v, l, j, g = (int(a[t]) for t in "ncHWkyxpquvljg")
    torch.manual_seed(0)
    inp = torch.randn(n, c, H, W, device=dev); w = torch.randn(k, c // g, y, x, device=dev) / (c * y * x) ** 0.5
    Ho, Wo = (H + 2 * p - l * (y - 1) - 1) // u + 1, (W + 2 * q - j * (x - 1) - 1) // v + 1
    wb = w.reshape(1, k, -1).expand(n, -1, -1)
    def f():
        cols = F.unfold(inp, (y, x), dilation=(l, j), padding=(p, q), stride=(u, v))
        return torch.bmm(wb, cols).reshape(n, k, Ho, Wo)
    ref = F.conv2d(inp, w, None, (u, v), (p, q), (l, j), g)
    outs = [f().clone() for _ in range(8)]
    nondet = any(not torch.equal(o, outs[0]) for o in outs[1:])
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): o = f()
    t = timed(gr.replay, 5) / 10
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2):
        for _ in range(10): o = F.conv2d(inp, w, None, (u, v), (p, q), (l, j), g)
    t2 = timed(gr2.replay, 5) / 10
    print("unfold+bmm %s: %.1f us (MIOpen default in a graph: %.1f us) nondet %s, max |d| vs MIOpen %.3g" % (cmd[5:75], t, t2, nondet, float((outs[0] - ref).abs().max())), flush=True)
