#!/usr/bin/env python3
"""Rehearsal of the HOST side of an 8-rank evaluation on one node ("the >= 6x at 8 GPUs target is a host-side
problem") on a box that has ONE GPU and the node's CPU quota.

Every rank is a process running the real input pipeline of rpeflow_amd.evaluate (rpeflow_amd/loader.py: loader threads ->
pinned host ring -> copy thread -> device ring) over its rank-strided shard of a cached synthetic set, with
LOCAL_WORLD_SIZE = ranks so that default_workers() hands out the per-rank share of the cores.  What a real rank does on the
host besides feeding the pipeline -- hipGraphLaunch (2.8 ms of host time per replay, asynchronous) plus the loop around it --
is played by a busy loop of --host-ms per batch; the forward itself is NOT run (eight forwards on one GPU would measure
the GPU).  All ranks share this box's single PCIe link, which on a node is one link per GPU: the copy thread therefore moves
1 / ranks of every tensor (RPE_PIPE_COPY_FRACTION), so the link carries one GPU's worth of traffic in total while the host
staging -- the contended resource: cores and DRAM bandwidth -- does all of its work.

Prints one JSON line: per-rank batches/s (capacity of the host side; a rank needs 64 per second to keep an MI355X at
15.6 ms per batch fed), CPU seconds per batch, cores in use.

    python tools/host_rehearsal.py --ranks 8 --batches 96
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    from rpeflow_amd import runtime
    runtime.configure()
    import torch
    import torch.distributed as dist
    from rpeflow_amd import evaluate as E
    from rpeflow_amd.loader import InputPipeline
    from rpeflow_amd.synthetic import SyntheticPairs
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(max(1, runtime.usable_cores() // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0) if args.device == "cuda" else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    n = world * args.batches * args.batch
    data = SyntheticPairs(n, args.height, args.width, 8192, distinct=world * args.distinct, cache=True, pin=args.pinned)
    mine = E.shard_indices(n, rank, world)
    t_gen = data.prepare(indices=mine, threads=2)
    workers = E.default_workers() if args.workers is None else args.workers
    dist.barrier()

    def run(indices):
        pipe = InputPipeline(data, indices, args.batch, dev, workers=workers)
        t0, c0 = time.perf_counter(), time.process_time()
        spin = 0.0
        for batch, upcoming in pipe.pairs():
            t_end = time.perf_counter() + args.host_ms * 1e-3  # the host side of one replay: hipGraphLaunch + the loop
            while time.perf_counter() < t_end:
                pass
            spin += args.host_ms * 1e-3
        if dev.type == "cuda":
            torch.cuda.synchronize()
        return time.perf_counter() - t0, time.process_time() - c0, spin, pipe.stats

    run(mine[:3 * args.batch])  # threads, rings, pinned allocations: first use untimed
    dist.barrier()
    dt, cpu, spin, stats = run(mine)
    out = {"rank": rank, "batches_per_s": round(args.batches / dt, 2), "cpu_s_per_batch": round(cpu / args.batches, 5),
           "cores_in_use": round(cpu / dt, 2), "of_which_replay_stand_in": round(spin / dt, 2), "loader_threads": workers,
           "h2d_MB_per_batch": round(stats["bytes"] / max(1, stats["batches"]) / 1e6, 1), "generator_s_untimed": round(t_gen, 2)}
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        rates = [g["batches_per_s"] for g in gathered]
        print(json.dumps({
            "what": "host side of a %d-rank evaluation on one node, rehearsed on one box (tools/host_rehearsal.py)" % world,
            "ranks": world, "usable_cores": runtime.usable_cores(), "batch": args.batch, "frame": [args.height, args.width],
            "batches_per_rank": args.batches, "staging": "none (samples pinned)" if args.pinned else "pinned ring of host batches",
            "copy_fraction": float(os.environ.get("RPE_PIPE_COPY_FRACTION", "1")), "replay_stand_in_ms": args.host_ms,
            "batches_per_s_per_rank": rates, "min_batches_per_s": min(rates), "needed_batches_per_s": args.need,
            "feeds_the_gpus": bool(min(rates) >= args.need), "frame_pairs_per_s_capacity": round(sum(rates) * args.batch, 1),
            "cores_in_use_total": round(sum(g["cores_in_use"] for g in gathered), 2), "per_rank": gathered}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--ranks", type=int, default=8)
    p.add_argument("--batches", type=int, default=96, help="timed batches per rank")
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--height", type=int, default=544)
    p.add_argument("--width", type=int, default=960)
    p.add_argument("--distinct", type=int, default=8, help="distinct cached samples per rank (51.5 MB each)")
    p.add_argument("--host-ms", type=float, default=3.3, help="host time per batch a real rank spends launching the replay (DESIGN.md section 6)")
    p.add_argument("--need", type=float, default=64.0, help="batches/s a rank must sustain (1 / 15.6 ms)")
    p.add_argument("--workers", type=int, default=None)
    p.add_argument("--pinned", action="store_true", help="cached set in pinned memory: no staging pass")
    p.add_argument("--device", choices=["cuda", "cpu"], default="cuda")
    p.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    args = p.parse_args()
    if args.child:
        return child(args)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.ranks), LOCAL_WORLD_SIZE=str(args.ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("RPE_PIPE_COPY_FRACTION", str(1.0 / args.ranks))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=env, cwd=ROOT))
    rc = 0
    for proc in procs:
        rc = proc.wait() or rc
    raise SystemExit(rc)


if __name__ == "__main__":
    main()
