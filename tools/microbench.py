"""Per-operator timings on one MI355X (torch.cuda events on the launch stream).
Usage: python tools/microbench.py [corr] [knn] [fps]   -> prints one line per case."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rpeflow_amd.csrc as ops
from rpeflow_amd.csrc import wrapper as W


def timeit(fn, warmup=3, iters=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    which = sys.argv[1:] or ["corr", "knn", "fps"]
    dev = "cuda:0"
    torch.manual_seed(0)
    if "corr" in which:
        a = torch.randn(1, 256, 544, 960, device=dev)
        b = torch.randn(1, 256, 544, 960, device=dev)
        alg_bytes = 2 * a.numel() * 4 + 81 * 544 * 960 * 4
        for algo in (7, 2, 8, 7):
            us = timeit(lambda: W._correlation2d_algo(a, b, 4, algo))
            print(f"corr algo={algo} 1x256x544x960: {us:9.1f} us  {alg_bytes / us / 1e6:7.3f} TB/s algorithmic  ({alg_bytes / us / 1e6 / 8.0:.3f} of 8 TB/s)")
        for (B, C, H, Wd) in [(4, 32, 144, 240), (4, 64, 72, 120), (4, 96, 36, 60), (4, 128, 18, 30), (4, 192, 9, 15)]:
            x, y = torch.randn(B, C, H, Wd, device=dev), torch.randn(B, C, H, Wd, device=dev)
            us = timeit(lambda: ops.correlation2d(x, y, 4))
            us1 = timeit(lambda: W._correlation2d_algo(x, y, 4, 1))
            us2 = timeit(lambda: W._correlation2d_algo(x, y, 4, 2))
            dma = [timeit(lambda: W._correlation2d_algo(x, y, 4, al)) if Wd % 4 == 0 else float("nan") for al in (8, 7)]
            print(f"corr model {B}x{C}x{H}x{Wd}: picked {us:9.1f} us   direct {us1:9.1f} us  mfma {us2:9.1f} us  dma4/7 {dma[0]:7.1f} {dma[1]:7.1f}")
    if "knn" in which:
        for (B, M, Q, D, k) in [(4, 8192, 4096, 3, 16), (4, 4096, 4096, 3, 16), (4, 4096, 34560, 2, 1), (4, 2048, 8640, 2, 1),
                                (4, 4096, 4096, 3, 3), (4, 2048, 4096, 3, 3), (4, 256, 256, 3, 16), (4, 8192, 8192, 3, 16)]:
            p = torch.rand(B, D, M, device=dev) * 30
            q = torch.rand(B, D, Q, device=dev) * 30
            us = timeit(lambda: ops.k_nearest_neighbor(p, q, k))
            print(f"knn B={B} M={M} Q={Q} D={D} k={k}: {us:9.1f} us  {B * M * Q / us / 1e3:8.2f} Gpair/s")
    if "fps" in which:
        for (B, N, S) in [(8, 8192, 4096), (2, 8192, 4096), (8, 4096, 1024)]:
            p = (torch.rand(B, 3, N, device=dev) * 30).transpose(1, 2)
            from rpeflow_amd import _lib
            idx = torch.empty((B, S), dtype=torch.int64, device=dev)
            for algo, name in ((1, "plain"), (2, "pruned")):
                us = timeit(lambda: _lib.check(_lib.lib().rpe_fps_algo(p.data_ptr(), *p.stride(), B, N, S, idx.data_ptr(), algo, None), "fps"),
                            warmup=1, iters=3)
                print(f"fps {name} B={B} N={N} S={S}: {us:9.1f} us  {us / S:6.3f} us/sample")
    if "restormer" in which:
        from rpeflow_amd.restormer_ops import channel_layernorm, dwconv3
        for (B, C, H, Wd) in [(4, 96, 144, 240), (4, 81, 144, 240), (4, 32, 144, 240), (4, 96, 72, 120), (4, 64, 1, 4096)]:
            shape = (B, C, H, Wd) if H > 1 else (B, C, Wd)
            x, y = torch.randn(*shape, device=dev), torch.randn(*shape, device=dev)
            k = (3, 3) if H > 1 else (3,)
            wq = torch.randn(3 * C, 1, *k, device=dev)
            hid = int(C * 2.66)
            h = torch.randn(shape[0], 2 * hid, *shape[2:], device=dev)
            wg = torch.randn(2 * hid, 1, *k, device=dev)
            us = timeit(lambda: dwconv3([x, y, y], wq))
            byt = x.numel() * 4 * 6
            print(f"dwconv qkv  {shape}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s")
            us = timeit(lambda: dwconv3([h], wg, gate=True))
            byt = h.numel() * 4 * 1.5
            print(f"dwconv gate {tuple(h.shape)}: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s")
            g, bb = torch.rand(C, device=dev), torch.rand(C, device=dev)
            us = timeit(lambda: channel_layernorm(x, g, bb))
            print(f"layernorm   {shape}: {us:8.1f} us  {x.numel() * 8 / us / 1e6:6.2f} TB/s")
    if "attn" in which:
        from rpeflow_amd.model import _MutualAttention
        for (dims, C, heads, sp) in [(2, 96, 2, (144, 240)), (2, 81, 1, (144, 240)), (2, 32, 1, (144, 240)), (2, 81, 1, (72, 120)),
                                     (1, 64, 2, (4096,)), (1, 192, 4, (256,))]:
            m = _MutualAttention(C, heads, False, dims).to(dev).eval()
            x, y = torch.randn(4, C, *sp, device=dev), torch.randn(4, C, *sp, device=dev)
            with torch.no_grad():
                us = timeit(lambda: m(x, y, residual=x))
                us0 = timeit(lambda: x + m.project_out(m._forward_plain(x, y)) if False else x + m._forward_plain(x, y))
                from rpeflow_amd.restormer_ops import channel_attention_matrix, dwconv3
                qkv = dwconv3([x, y, y], m.qkv_dwconv.weight)
                usm = timeit(lambda: channel_attention_matrix(qkv, heads, m.temperature, m.project_out.weight))
            print(f"attention C={C} heads={heads} {sp}: fused {us:8.1f} us (matrix part {usm:7.1f})   plain torch {us0:8.1f} us")


if __name__ == "__main__":
    main()
