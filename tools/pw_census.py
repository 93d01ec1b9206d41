#!/usr/bin/env python3
"""Every rpe_pointwise_conv launch of one forward (BASELINE config 3), by shape: how often, and what each shape costs alone.

The entry is recorded where the host calls it (the ctypes function of the loaded library is wrapped for one forward), then every
distinct shape is launched 40 times on fresh N(0,1) tensors between two events.  With RPE_HIP_LIB pointing at a library built
with other dispatch macros (tools/experiments/build_probe.sh) the same table is the A/B of a dispatch rule.

    python tools/pw_census.py [--shapes-only]
"""
import ctypes
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from rpeflow_amd import _lib  # noqa: E402
from rpeflow_amd.model import RPEFlow  # noqa: E402
from rpeflow_amd.synthetic import load_seeded_parameters  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batch = bench.make_batch(4, dev)
    model(batch)
    torch.cuda.synchronize()
    lib = _lib.lib()
    real = lib.rpe_pointwise_conv
    seen = Counter()

    def recorder(x, xbs, B, C, P, w, wbs, cout, scale, shift, act, slope, res, rbs, y, stream):
        val = lambda v: getattr(v, "value", v)
        seen[(int(val(B)), int(val(C)), int(val(P)), int(val(cout)), int(val(wbs)) != 0, bool(val(res)), int(val(act)))] += 1
        return real(x, xbs, B, C, P, w, wbs, cout, scale, shift, act, slope, res, rbs, y, stream)

    lib.rpe_pointwise_conv = recorder
    try:
        model(batch)
        torch.cuda.synchronize()
    finally:
        lib.rpe_pointwise_conv = real
    del model, batch
    print("%d launches, %d shapes" % (sum(seen.values()), len(seen)))
    if "--shapes-only" in sys.argv:
        for key, n in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3]):
            print(n, key)
        return
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    rows, total = [], 0.0
    for (B, C, P, cout, per_sample, has_res, act), n in seen.items():
        x = torch.randn(B, C, P, device=dev)
        kt, ot = (C + 3) // 4, (cout + 15) // 16
        w = torch.randn((B if per_sample else 1) * ot * kt * 64, device=dev) * 0.1
        res = torch.randn(B, cout, P, device=dev) if has_res else None
        y = torch.empty(B, cout, P, device=dev)
        shift = torch.randn(cout, device=dev)
        call = lambda: real(ptr(x), C * P, B, C, P, ptr(w), ot * kt * 64 if per_sample else 0, cout, None, ptr(shift), act, 0.1,
                            ptr(res) if has_res else None, cout * P if has_res else 0, ptr(y), stream)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(40):
            call()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 40 * 1e3
        flops = 2.0 * B * C * P * cout
        byts = 4.0 * B * P * (C + cout * (2 if has_res else 1))
        wgs_ks = ((P + 63) // 64) * ot * B
        wgs1 = ((P + 63) // 64) * ((ot + 3) // 4) * B
        kind = "ksplit" if wgs_ks <= 1024 and kt >= 16 else ("OT4" if ot >= 16 and wgs1 >= 4096 else ("OT2" if ot >= 8 and wgs1 >= 2048 else "OT1"))
        rows.append((n * us, n, B, C, cout, P, per_sample, has_res, kind, us, flops / us / 1e6, byts / us / 1e3))
        total += n * us
    rows.sort(reverse=True)
    print("  n   B  Cin Cout      P  w/b res  kernel     us  TFLOP/s   GB/s   n*us")
    for tot, n, B, C, cout, P, ps, hr, kind, us, tf, gbs in rows:
        print("%3d %3d %4d %4d %6d  %3s %3s  %-6s %6.1f  %7.1f %6.0f %6.0f" % (n, B, C, cout, P, "y" if ps else "-", "y" if hr else "-", kind, us, tf, gbs, tot))
    print("sum over the forward's launches, each shape alone: %.0f us" % total)


if __name__ == "__main__":
    main()
