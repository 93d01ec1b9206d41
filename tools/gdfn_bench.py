#!/usr/bin/env python3
"""The gated feed-forward's tail at the forward's level-1 / level-2 shapes: dwconv3(gate) + 1x1 (two launches) against rpe_gdfn_tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpeflow_amd.model import _GatedFeedForward
from rpeflow_amd.restormer_ops import dwconv3, gdfn_tail
from rpeflow_amd.utils import conv_module

dev = "cuda:0"


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


with torch.no_grad():
    for dims, shape in [(2, (4, 96, 144, 240)), (2, (4, 81, 144, 240)), (2, (8, 32, 144, 240)), (2, (4, 81, 72, 120)), (2, (8, 64, 72, 120)),
                        (2, (4, 81, 36, 60)), (1, (8, 32, 4096)), (1, (4, 64, 4096)), (1, (8, 64, 2048))]:
        torch.manual_seed(0)
        ffn = _GatedFeedForward(shape[1], 2.66, False, dims).to(dev).eval()
        x, r = torch.randn(shape, device=dev), torch.randn(shape, device=dev)
        t = conv_module(ffn.project_in, x)
        acc = r.clone()
        two = lambda: conv_module(ffn.project_out, dwconv3([t], ffn.dwconv.weight, None, gate=True), residual=acc, inplace=True)
        one = lambda: gdfn_tail(t, ffn.dwconv.weight, None, ffn.project_out.weight, None, acc, inplace=True)
        print("%s C=%d: dwconv + 1x1 %.1f us | one launch %.1f us" % ("x".join(map(str, shape)), shape[1], timed(two), timed(one)), flush=True)
