#!/usr/bin/env python3
"""Which thread of a rank burns a core while the forward is replayed, and what makes it?  (round 6)

A real rank of the evaluation used 3.4 cores (profiles/r05_host_rehearsal_real_*.json).  Per-thread CPU time
(rpeflow_amd.runtime.thread_cpu_seconds) says: loader threads 1.2-1.35 (the staging memcpy), the copy thread 0.94 (spinning
inside hipEventSynchronize: fixed by polling), the main thread 0.2 -- and ONE helper thread of a library at 0.99.  A thread
inherits the name of its creator, so the main thread is renamed phase by phase and the helper's name says when it was born;
the loops below say what keeps it busy: graph replays alone, eager launches alone, pinned H2D copies alone.

    python tools/experiments/thread_spin.py [replay|eager|copies|idle ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rpeflow_amd import runtime  # noqa: E402

runtime.configure()
runtime.name_thread("ph-import")
import torch  # noqa: E402

import bench  # noqa: E402
from rpeflow_amd.evaluate import GraphedForward  # noqa: E402
from rpeflow_amd.model import RPEFlow  # noqa: E402
from rpeflow_amd.synthetic import load_seeded_parameters  # noqa: E402


def account(label, seconds, body):
    before = runtime.thread_cpu_seconds()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        body()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = []
    for key, sec in runtime.thread_cpu_seconds().items():
        used = sec - before.get(key, 0.0)
        if used / dt >= 0.02:
            rows.append({"tid": key[0], "born": key[1], "cores": round(used / dt, 3)})
    rows.sort(key=lambda r: -r["cores"])
    print(json.dumps({"loop": label, "seconds": round(dt, 2), "iterations": n, "threads": rows,
                      "all_threads": sorted({k[1] for k in runtime.thread_cpu_seconds()})}), flush=True)


def main():
    which = sys.argv[1:] or ["idle", "replay", "replay_paced", "replay_ring", "eager", "copies", "replay+copies"]
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    runtime.name_thread("ph-hip-init")
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    runtime.name_thread("ph-model-cpu")
    model = load_seeded_parameters(RPEFlow())
    runtime.name_thread("ph-model-to")
    model = model.to(dev).eval()
    runtime.name_thread("ph-batch")
    batch = bench.make_batch(4, dev)
    runtime.name_thread("ph-first-fwd")
    model(batch)
    torch.cuda.synchronize()
    runtime.name_thread("ph-capture")
    forward = GraphedForward(model, warmup=1, ahead=True)
    forward(batch, batch)
    torch.cuda.synchronize()
    graph = forward.entries[forward._key(batch)]["graph"]
    runtime.name_thread("ph-pinned")
    src = torch.empty(52 << 20, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(52 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)
    runtime.name_thread("ph-loops")

    def replay():
        graph.replay()

    def replay_paced():  # one replay in flight at a time: the host never queues ahead
        graph.replay()
        torch.cuda.synchronize()

    def eager():
        model(batch)

    def copies():
        with torch.cuda.stream(side):
            dst.copy_(src, non_blocking=True)
        time.sleep(0.004)

    ring = []

    def replay_ring():  # what a rank of the evaluation does: at most three replays in flight, the host waits by POLLING an old event
        graph.replay()
        ev = torch.cuda.Event()
        ev.record()
        ring.append(ev)
        if len(ring) > 3:
            old = ring.pop(0)
            while not old.query():
                time.sleep(0.00025)

    def both():
        graph.replay()
        with torch.cuda.stream(side):
            dst.copy_(src, non_blocking=True)

    loops = {"idle": lambda: time.sleep(0.01), "replay": replay, "replay_paced": replay_paced, "replay_ring": replay_ring, "eager": eager, "copies": copies, "replay+copies": both}
    for name in which:
        account(name, 4.0, loops[name])


if __name__ == "__main__":
    main()
