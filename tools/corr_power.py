#!/usr/bin/env python3
"""Power / clock trace of the correlation microbench (BASELINE config 2) for different operand contents.

The same launch (same instructions, same HBM traffic) takes 325 us on all-zero feature maps and ~405 us on N(0,1) ones
(DESIGN.md 4.1): the claim is that with random operands the matrix pipe and HBM together run into the package power limit and
the clock drops.  This tool keeps the evidence: while the kernel runs back to back for a few seconds per operand kind, a
thread samples the GPU's average socket power and shader clock (hwmon / amdgpu sysfs, else `rocm-smi --json`), and the
script prints per kind: us per launch, mean / max power, mean clock, and the launch time scaled to the nominal clock.

    python tools/corr_power.py [--seconds 3] [--out profiles/r03_corr_power.json]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rpeflow_amd.csrc as ops  # noqa: E402


def sysfs_sources():
    power, clock = None, None
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for p in glob.glob(os.path.join(card, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(card, "hwmon/hwmon*/power1_input")):
            if os.access(p, os.R_OK):
                power = power or p
        for c in glob.glob(os.path.join(card, "hwmon/hwmon*/freq1_input")):
            if os.access(c, os.R_OK):
                clock = clock or c
        if power or clock:
            break
    return power, clock


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        watts = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
        mhz = None
        for k, v in card.items():
            if k.startswith("sclk clock speed"):
                mhz = float(str(v).strip("()").lower().replace("mhz", ""))
        return watts, mhz
    except Exception:  # noqa: BLE001
        return None, None


class Sampler(threading.Thread):
    def __init__(self, period=0.02):
        super().__init__(daemon=True)
        self.power_path, self.clock_path = sysfs_sources()
        self.period = period if (self.power_path or self.clock_path) else 0.25
        self.samples, self.stop_flag = [], threading.Event()

    def run(self):
        while not self.stop_flag.is_set():
            watts = mhz = None
            try:
                if self.power_path:
                    watts = float(open(self.power_path).read()) / 1e6
                if self.clock_path:
                    mhz = float(open(self.clock_path).read()) / 1e6
            except (OSError, ValueError):
                pass
            if watts is None and mhz is None:
                watts, mhz = smi_sample()
            self.samples.append((time.perf_counter(), watts, mhz))
            time.sleep(self.period)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--seconds", type=float, default=3.0)
    p.add_argument("--out", default=None)
    p.add_argument("--source", choices=["sysfs", "smi"], default="sysfs")
    args = p.parse_args()
    dev = torch.device("cuda", 0)
    H, W, C = 544, 960, 256
    kinds = {
        "zeros": lambda: torch.zeros(1, C, H, W, device=dev),
        "normal * 1e-30": lambda: torch.randn(1, C, H, W, device=dev) * 1e-30,
        "constant 1.0": lambda: torch.ones(1, C, H, W, device=dev),
        "normal(0, 1)": lambda: torch.randn(1, C, H, W, device=dev),
    }
    result = {"workload": "correlation2d 1x256x544x960 md=4 fp32 (BASELINE config 2), back to back for %.1f s per operand kind" % args.seconds,
              "kinds": {}}
    for name, make in kinds.items():
        a, b = make(), make()
        for _ in range(150):
            ops.correlation2d(a, b, 4)
        torch.cuda.synchronize()
        sampler = Sampler()
        if args.source == "smi":
            sampler.power_path = sampler.clock_path = None
            sampler.period = 0.2
        sampler.start()
        t0 = time.perf_counter()
        n = 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.perf_counter() - t0 < args.seconds:
            for _ in range(50):
                ops.correlation2d(a, b, 4)
            n += 50
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        sampler.stop_flag.set()
        sampler.join()
        us = s.elapsed_time(e) / n * 1e3
        watts = [w for _, w, _ in sampler.samples if w is not None]
        mhz = [m for _, _, m in sampler.samples if m is not None]
        entry = {"us_per_launch": round(us, 1), "launches": n, "samples": len(sampler.samples),
                 "power_W_mean": round(sum(watts) / len(watts), 1) if watts else None, "power_W_max": round(max(watts), 1) if watts else None,
                 "sclk_MHz_mean": round(sum(mhz) / len(mhz), 1) if mhz else None, "sclk_MHz_min": round(min(mhz), 1) if mhz else None,
                 "source": "sysfs hwmon" if (sampler.power_path or sampler.clock_path) else "rocm-smi --json"}
        if mhz:
            entry["us_at_2400MHz"] = round(us * (sum(mhz) / len(mhz)) / 2400.0, 1)
        result["kinds"][name] = entry
        print(name, entry, flush=True)
        del a, b
        time.sleep(1.0)  # let the package cool between kinds
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(result, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
