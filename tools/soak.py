#!/usr/bin/env python3
"""Soak run of the replayed forward (BASELINE config 3): minutes of back-to-back batches through rpeflow_amd.evaluate.GraphedForward,
rotating over a few distinct batches so that every replay also samples the NEXT batch's clouds (the evaluation's steady state).

Checked: every replay's flow_2d / flow_3d equal, bit for bit, the first replay's for the same batch (compared on the device, counted
without a host sync); device memory and host RSS do not grow; the step time does not drift (windows of 256 replays).

    python tools/soak.py [--minutes 10] [--batches 4] [--out profiles/r05_soak.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rpeflow_amd import runtime  # noqa: E402
from rpeflow_amd.evaluate import GraphedForward  # noqa: E402
from rpeflow_amd.model import RPEFlow  # noqa: E402
from rpeflow_amd.synthetic import load_seeded_parameters  # noqa: E402

KEYS = ("flow_2d", "flow_3d")


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--minutes", type=float, default=10.0)
    p.add_argument("--batches", type=int, default=4)
    p.add_argument("--window", type=int, default=256)
    p.add_argument("--out", default=None)
    args = p.parse_args()
    torch.set_grad_enabled(False)
    runtime.configure()
    dev = torch.device("cuda", 0)
    proc = psutil.Process()
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batches = [bench.make_batch(4, dev, first_seed=1000 + 4 * i) for i in range(args.batches)]
    forward = GraphedForward(model, warmup=1, ahead=True)
    n = len(batches)
    expected = []
    for i in range(n):  # first round: what every later replay of batch i has to reproduce
        out = forward(batches[i], batches[(i + 1) % n])
        expected.append({k: out[k].clone() for k in KEYS})
    torch.cuda.synchronize()
    wrong = torch.zeros((), dtype=torch.int64, device=dev)
    mem0 = {"allocated_MB": torch.cuda.memory_allocated(dev) / 2**20, "reserved_MB": torch.cuda.memory_reserved(dev) / 2**20, "host_rss_MB": proc.memory_info().rss / 2**20}
    windows, steps, t_start = [], 0, time.perf_counter()
    hw, rss = [], []
    while time.perf_counter() - t_start < args.minutes * 60:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(args.window):
            i = (steps + j) % n
            out = forward(batches[i], batches[(i + 1) % n])
            for k in KEYS:
                wrong += (out[k] != expected[i][k]).any()
        torch.cuda.synchronize()
        windows.append((time.perf_counter() - t0) / args.window * 1e3)
        steps += args.window
        hw.append(runtime.hwmon_sclk_mhz(dev))
        rss.append(proc.memory_info().rss / 2**20)
    mem1 = {"allocated_MB": torch.cuda.memory_allocated(dev) / 2**20, "reserved_MB": torch.cuda.memory_reserved(dev) / 2**20, "host_rss_MB": proc.memory_info().rss / 2**20}
    windows_sorted = sorted(windows)
    third = max(1, len(windows) // 3)
    report = {
        "what": "GraphedForward(ahead=True) over %d rotating batches of 4 frame pairs (544x960 + 8192 points), every output compared with the first replay's" % n,
        "minutes": round((time.perf_counter() - t_start) / 60, 2), "replays": steps, "frame_pairs": 4 * steps,
        "replays_with_a_differing_output": int(wrong.item()),
        "ms_per_step_incl_compare": {"min_window": round(windows_sorted[0], 3), "median_window": round(windows_sorted[len(windows) // 2], 3), "max_window": round(windows_sorted[-1], 3),
                                     "first_third_mean": round(sum(windows[:third]) / third, 3), "last_third_mean": round(sum(windows[-third:]) / third, 3),
                                     "window_replays": args.window, "windows": len(windows)},
        "memory_start": {k: round(v, 1) for k, v in mem0.items()}, "memory_end": {k: round(v, 1) for k, v in mem1.items()},
        "host_rss_MB_at_tenths_of_the_run": [round(rss[min(len(rss) - 1, i * len(rss) // 10)], 1) for i in range(11)] if rss else None,
        "hwmon_sclk_MHz_between_windows": [min(x for x in hw if x) if any(hw) else None, max(x for x in hw if x) if any(hw) else None],
        "device": torch.cuda.get_device_name(dev),
    }
    print(json.dumps(report, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
    sys.exit(0 if report["replays_with_a_differing_output"] == 0 else 3)


if __name__ == "__main__":
    main()
