#!/usr/bin/env python3
"""Every 1x1 convolution shape of the forward (from a solver_lottery.py log): MIOpen convolution vs torch.matmul forms."""
import json, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_solver_lottery.json"
cmds = json.load(open(src))["choices_run0"]

def timed(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

tot = {"conv": 0.0, "matmul": 0.0, "bmm_t": 0.0}
rows = []
for cmd, alg in cmds.items():
    a = dict(re.findall(r"-(\w) (\d+)", cmd))
    if a["y"] != "1" or a["x"] != "1": continue
    n, c, H, W, k, u = (int(a[x]) for x in ("n", "c", "H", "W", "k", "u"))
    x = torch.randn(n, c, H, W, device=dev); w = torch.randn(k, c, 1, 1, device=dev)
    t_conv = timed(lambda: F.conv2d(x, w, None, u))
    xs = x[:, :, ::u, ::u] if u > 1 else x
    def mm():
        xf = xs.reshape(n, c, -1)
        return torch.bmm(w.view(1, k, c).expand(n, -1, -1), xf)
    t_mm = timed(mm)
    def bt():  # [B,P,C] x [C,K] -> transpose back
        xf = xs.reshape(n, c, -1)
        return torch.matmul(xf.transpose(1, 2), w.view(k, c).t()).transpose(1, 2)
    t_bt = timed(bt)
    tot["conv"] += t_conv; tot["matmul"] += t_mm; tot["bmm_t"] += t_bt
    rows.append((t_mm - t_conv, "n=%d c=%d HxW=%dx%d k=%d s=%d: conv %6.1f (%s) matmul %6.1f bmm_t %6.1f" % (n, c, H, W, k, u, t_conv, alg[0][:18], t_mm, t_bt)))
for _, r in sorted(rows, reverse=True)[:40]: print(r)
print("totals us:", {k: round(v) for k, v in tot.items()}, "over", len(rows), "shapes")
