#!/usr/bin/env python3
"""On the GPU: every device kernel of one eager step that is NOT one of this library's (ATen element-wise / cat / copy / fill
kernels, runtime copies, library GEMMs), grouped by the ATen operator that launched it, its input shapes and the innermost
rpeflow_amd / bench.py source line -- torch.profiler with shapes and stacks.  The device-side companion of tools/aten_census.py.

    python tools/aten_gpu_census.py [hotpath|forward]
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from rpeflow_amd import runtime  # noqa: E402

runtime.configure()
what = sys.argv[1] if len(sys.argv) > 1 else "hotpath"
dev = torch.device("cuda", 0)
if what == "hotpath":
    from rpeflow_amd.hotpath import HotPathWorkload
    wl = HotPathWorkload(batch=4, height=544, width=960, n_points=8192, device=dev)
    step = lambda: wl()
else:
    import bench
    from rpeflow_amd.model import RPEFlow
    from rpeflow_amd.synthetic import load_seeded_parameters
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batch = bench.make_batch(4, dev)
    step = lambda: model(batch)
with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
launches = 0
for ev in prof.events():
    if not ev.kernels:
        continue
    # the innermost operator that owns the kernels: skip parents whose children own them
    if any(ch.kernels for ch in ev.cpu_children):
        continue
    names = [k.name for k in ev.kernels]
    where = "?"
    for fr in (ev.stack or []):
        if ("rpeflow_amd" in fr or "bench.py" in fr) and "aten_gpu_census" not in fr:
            where = fr.split("/")[-1][:60]
            break
    shapes = str([s for s in (ev.input_shapes or []) if s])[:90]
    for k in ev.kernels:
        launches += 1
        rows[(ev.name, where, shapes, k.name[:70])][0] += 1
        rows[(ev.name, where, shapes, k.name[:70])][1] += k.duration
print("device kernels in one eager %s step: %d" % (what, launches))
generic = {k: v for k, v in rows.items() if k[0].startswith("aten::") or "rocclr" in k[3] or "Memcpy" in k[3] or "Memset" in k[3]}
print("launched by ATen operators / runtime copies: %d kernels, %.1f us" % (sum(v[0] for v in generic.values()), sum(v[1] for v in generic.values())))
for (op, where, shapes, kern), (n, us) in sorted(generic.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d %7.1f us  %-22s %-46s %-60s %s" % (n, us, op, where, kern[:60], shapes))
