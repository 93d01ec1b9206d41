#!/usr/bin/env python3
"""What clock does the correlation microbench (BASELINE config 2) -- or, with --op, one of the level-1 1x1 layers -- run at, and is
its cycle count constant?

Three instruments gave three answers in round 4 (hwmon 157 / 2350 MHz, the round-3 power trace 1964 MHz, GRBM_GUI_ACTIVE
1.5-2.0 GHz).  This tool uses the one that is measured on the device the kernel runs on, in the stream it runs in:
rpe_clock_stamp_all (s_memtime + s_memrealtime of every compute unit, each compared with itself only) in front of and behind
the launches (rpeflow_amd.runtime.ShaderClock).

1. Does s_memtime follow the engine clock on gfx950?  An idle stretch (two stamps with a host sleep between them) against a
   busy one: a counter at a constant rate gives the same MHz for both, the engine clock does not.
2. Per operand kind (zeros, constants, N(0,1)): us per launch, engine MHz over exactly those launches, hence CYCLES per launch.
   Constant cycles across kinds = the same schedule, the time difference is clock (power); different cycles = the schedule.
3. The hwmon reading of THIS device (by PCI address, label sclk) and the average socket power beside it, for comparison.

    python tools/corr_clock.py [--launches 400] [--out profiles/r05_corr_clock.json]
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rpeflow_amd.csrc as ops  # noqa: E402
from rpeflow_amd import runtime  # noqa: E402


def hwmon_dir(dev):
    prop = torch.cuda.get_device_properties(dev)
    bdf = "%04x:%02x:%02x.0" % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
    dirs = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
    return bdf, (dirs[0] if dirs else None)


class Hwmon(threading.Thread):
    """sclk and socket power of the device's own hwmon node, every 5 ms while a loop runs."""

    def __init__(self, d):
        super().__init__(daemon=True)
        self.d, self.clock, self.power, self.stop_flag = d, [], [], threading.Event()
        self.sclk = None
        for label in glob.glob(os.path.join(d or "/nonexistent", "freq*_label")):
            if open(label).read().strip() == "sclk":
                self.sclk = label.replace("_label", "_input")

    def run(self):
        while not self.stop_flag.is_set():
            try:
                if self.sclk:
                    self.clock.append(float(open(self.sclk).read()) / 1e6)
                for name in ("power1_average", "power1_input"):
                    p = os.path.join(self.d or "/nonexistent", name)
                    if os.path.exists(p):
                        self.power.append(float(open(p).read()) / 1e6)
                        break
            except (OSError, ValueError):
                pass
            time.sleep(0.005)

    def summary(self):
        m = lambda v: round(sum(v) / len(v), 1) if v else None
        return {"hwmon_sclk_MHz_mean": m(self.clock), "hwmon_sclk_MHz_range": [min(self.clock), max(self.clock)] if self.clock else None,
                "hwmon_power_W_mean": m(self.power), "hwmon_power_W_max": max(self.power) if self.power else None, "hwmon_readings": len(self.clock)}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--launches", type=int, default=400)
    p.add_argument("--out", default=None)
    p.add_argument("--op", default="corr", choices=["corr", "pw_in", "pw_mid", "pw_out"],
                   help="corr: the correlation microbench; pw_*: the level-1 cross block's 1x1 layers over 4 x 144 x 240 "
                        "(96 -> 510 project_in, 96 -> 96, 255 -> 96 + residual project_out)")
    args = p.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    bdf, hdir = hwmon_dir(dev)
    report = {"device": torch.cuda.get_device_name(dev), "pci": bdf, "hwmon_dir": hdir, "launches": args.launches}

    # 1. idle against busy
    if args.op == "corr":
        x = torch.randn(1, 256, 544, 960, device=dev)
        y = torch.randn(1, 256, 544, 960, device=dev)
        launch = lambda a, b: ops.correlation2d(a, b, 4)
    else:
        from rpeflow_amd import utils as U
        cin, cout, res = {"pw_in": (96, 510, False), "pw_mid": (96, 96, False), "pw_out": (255, 96, True)}[args.op]
        conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
        x = torch.randn(4, cin, 144, 240, device=dev)
        y = torch.randn(4, cout, 144, 240, device=dev)  # the residual of project_out
        launch = (lambda a, b: U.conv_module(conv, a, residual=b)) if res else (lambda a, b: U.conv_module(conv, a))
    report["op"] = args.op
    torch.cuda.synchronize()
    time.sleep(1.0)
    idle = runtime.ShaderClock(dev)
    with idle:
        torch.cuda.synchronize()
        time.sleep(0.5)
    torch.cuda.synchronize()
    cycles, ticks, khz = idle.raw()
    report["idle_MHz_per_xcd"] = idle.mhz_per_xcd()
    report["idle"] = {"shader_cycles": cycles, "wall_ticks": ticks, "wall_kHz": khz, "MHz": round(cycles / max(1, ticks) * khz / 1e3, 2),
                      "seconds_by_wall_counter": round(ticks / khz / 1e3, 4)}

    kinds = {"zeros": (torch.zeros_like(x), torch.zeros_like(y)), "ones": (torch.ones_like(x), torch.ones_like(y)), "normal": (x, y)}
    report["kinds"] = {}
    for name, (a, b) in kinds.items():
        for _ in range(150):  # settle
            launch(a, b)
        torch.cuda.synchronize()
        mon = Hwmon(hdir)
        mon.start()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        clock = runtime.ShaderClock(dev)
        s.record()
        with clock:
            for _ in range(args.launches):
                launch(a, b)
        e.record()
        torch.cuda.synchronize()
        mon.stop_flag.set()
        mon.join()
        us = s.elapsed_time(e) / args.launches * 1e3
        cycles, ticks, khz = clock.raw()  # (the median XCD's pair)
        mhz = cycles / max(1, ticks) * khz / 1e3
        report["kinds"][name] = {"us_per_launch": round(us, 2), "stamp_MHz": round(mhz, 1), "stamp_MHz_per_xcd": clock.mhz_per_xcd(), "units_read": clock.units(),
                                 "shader_cycles_per_launch": round(cycles / args.launches),
                                 "us_by_wall_counter": round(ticks / khz * 1e3 / args.launches, 2), **mon.summary()}
        print(name, report["kinds"][name], flush=True)
    k = report["kinds"]
    report["conclusion_inputs"] = {
        "s_memtime_follows_engine_clock": abs(report["idle"]["MHz"] - k["normal"]["stamp_MHz"]) > 0.05 * k["normal"]["stamp_MHz"],
        "cycles_normal_over_zeros": round(k["normal"]["shader_cycles_per_launch"] / max(1, k["zeros"]["shader_cycles_per_launch"]), 4),
        "us_normal_over_zeros": round(k["normal"]["us_per_launch"] / k["zeros"]["us_per_launch"], 4)}
    print(json.dumps(report, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
