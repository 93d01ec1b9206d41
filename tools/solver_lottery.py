#!/usr/bin/env python3
"""Which convolutions change MIOpen solver between fresh processes, and what that does to the parity margin.

    python tools/solver_lottery.py [--runs N] [--mode default|immediate|fast] [--out gpurun_out/lottery.json]

This driver never touches the GPU.  It starts N child processes one after the other, each with an EMPTY MIOpen user
database (MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR in a fresh temp directory: what the first process on a fresh
machine sees), MIOpen's command + info logging on, and the benched configuration of bench.py (batch 4 of 544x960 + 8192
points, seeded parameters): two eager forwards, |dEPE| against the reference's CPU golden, ms per graph replay.  The
child's stderr is parsed for (convolution command line -> chosen algorithm/solver); the summary lists, per run, the EPE
deltas and, across runs, every convolution whose choice differs.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
from rpeflow_amd import runtime
runtime.configure(%(configure_kw)s)
import numpy as np, torch
%(pre)s
import bench
from rpeflow_amd.model import RPEFlow
from rpeflow_amd.synthetic import load_seeded_parameters
dev = torch.device("cuda", 0)
g = np.load(os.path.join(%(root)r, "tests", "golden", "model_bench_b4_544x960.npz"))
model = load_seeded_parameters(RPEFlow()).to(dev).eval()
batch = bench.make_batch(4, dev, first_seed=1000)
with torch.no_grad():
    t0 = time.perf_counter()
    out = model(batch); torch.cuda.synchronize()
    first_s = time.perf_counter() - t0
    out = model(batch); torch.cuda.synchronize()
    d = bench.golden_epe_delta(out, batch, g)
    from rpeflow_amd.evaluate import GraphedForward
    fwd = GraphedForward(model, warmup=0)
    o = fwd(batch, batch); torch.cuda.synchronize()
    dg = bench.golden_epe_delta(o, batch, g)
    rep = fwd.entries[fwd._key(batch)]["graph"].replay
    for _ in range(10): rep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rep()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print("RESULT " + json.dumps({"eager": d, "graph": dg, "ms_per_step": round(ms, 3), "first_forward_s": round(first_s, 2)}))
'''

MODES = {
    "default": dict(env={}, pre="", configure_kw=""),
    "immediate": dict(env={}, pre="torch.backends.miopen.immediate = True", configure_kw=""),
    "fast": dict(env={"MIOPEN_FIND_MODE": "2"}, pre="", configure_kw=""),
}


def parse_log(text):
    """(conv command line -> list of chosen algorithms) from MIOpen's stderr log."""
    chosen = collections.OrderedDict()
    cmd = None
    for line in text.splitlines():
        m = re.search(r"MIOpenDriver (conv\w* .*)$", line)
        if m:
            cmd = re.sub(r"\s+", " ", m.group(1)).strip()
            continue
        m = re.search(r"Chosen Algorithm: *([\w<>:,]+)", line)
        if m and cmd is not None:
            chosen.setdefault(cmd, []).append(m.group(1))
            continue
        m = re.search(r"(?:FW|BW|WRW)? ?[Cc]hosen [Aa]lgo\w*\W+(\w+)", line)
        if m and cmd is not None:
            chosen.setdefault(cmd, []).append(m.group(1))
    return chosen


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--runs", type=int, default=3)
    p.add_argument("--mode", choices=list(MODES), default="default")
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lottery.json"))
    p.add_argument("--keep-log", action="store_true", help="keep the first run's raw MIOpen log lines that mention an algorithm next to --out")
    p.add_argument("--no-log", action="store_true", help="no MIOpen logging (timing and EPE only)")
    args = p.parse_args()
    mode = MODES[args.mode]
    runs = []
    for r in range(args.runs):
        with tempfile.TemporaryDirectory(prefix="miopen_fresh_") as tmp:
            env = dict(os.environ, MIOPEN_USER_DB_PATH=os.path.join(tmp, "db"), MIOPEN_CUSTOM_CACHE_DIR=os.path.join(tmp, "cache"), **mode["env"])
            if not args.no_log:
                env.update(MIOPEN_ENABLE_LOGGING_CMD="1", MIOPEN_LOG_LEVEL="5")
            os.makedirs(env["MIOPEN_USER_DB_PATH"]); os.makedirs(env["MIOPEN_CUSTOM_CACHE_DIR"])
            code = CHILD % {"root": ROOT, "pre": mode["pre"], "configure_kw": mode["configure_kw"]}
            proc = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
        res = next((json.loads(l[7:]) for l in proc.stdout.splitlines() if l.startswith("RESULT ")), None)
        if res is None:
            print("run %d failed:\n%s" % (r, proc.stderr[-3000:]), file=sys.stderr)
            raise SystemExit(1)
        res["chosen"] = parse_log(proc.stderr)
        if args.keep_log and r == 0:
            keep = [l for l in proc.stderr.splitlines() if re.search(r"lgorithm|MIOpenDriver|Solver|solver", l)]
            open(args.out + ".log", "w").write("\n".join(keep[:20000]))
        runs.append(res)
        print("run %d: eager dEPE2D %.3g dEPE3D %.3g | graph %.3g %.3g | %.3f ms | %d convolutions logged" % (
            r, res["eager"]["epe2d"], res["eager"]["epe3d"], res["graph"]["epe2d"], res["graph"]["epe3d"], res["ms_per_step"], len(res["chosen"])), flush=True)
    differing = {}
    for cmd in runs[0]["chosen"]:
        picks = [tuple(x["chosen"].get(cmd, [])) for x in runs]
        if len(set(picks)) > 1:
            differing[cmd] = [list(x) for x in picks]
    summary = {"mode": args.mode, "runs": [{k: v for k, v in x.items() if k != "chosen"} for x in runs], "differing_convolutions": differing,
               "n_convolutions": len(runs[0]["chosen"]), "choices_run0": {k: v for k, v in runs[0]["chosen"].items()}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(summary, open(args.out, "w"), indent=1)
    print("differing convolutions: %d of %d" % (len(differing), len(runs[0]["chosen"])))
    for cmd, picks in differing.items():
        print("  ", cmd, "->", picks)


if __name__ == "__main__":
    main()
