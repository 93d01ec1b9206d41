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

``--real-rank R`` (round 5): rank R is a REAL rank -- the true rpeflow_amd.evaluate.evaluate() on the box's GPU (model, HIP-graph
replay with the next batch's sampling inside, copy stream, device accumulators, a world-size-1 nccl group for its collective)
moving the whole of every batch over the PCIe link -- while the other ranks play the host side of their GPUs beside it under the
same core quota (their copy threads move 1 / 64 of every tensor, so the link is the real rank's).  Its batches/s against the
same run with ``--ranks 1`` is what seven neighbours cost a rank; hipGraphLaunch host time p50 / p99 is reported with it.
``--pace R``: the host-only ranks take at most R batches per second -- ranks whose GPU needs 15 ms a batch do not ask for more;
unpaced they run at the host's capacity (80-100 batches/s) and take more than a real neighbour's share of the cores.

    python tools/host_rehearsal.py --ranks 8 --batches 96 [--real-rank 0]
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
    # a thread inherits the name of the thread that creates it: the main thread carries the name of the phase it is in, so the
    # per-thread CPU account below says which phase a library's helper thread was born in ("at-import", "at-gloo", "at-hip", ...)
    runtime.name_thread("at-import")
    import torch
    import torch.distributed as dist
    from rpeflow_amd import evaluate as E
    from rpeflow_amd.loader import InputPipeline
    from rpeflow_amd.synthetic import SyntheticPairs
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(max(1, runtime.usable_cores() // world))
    runtime.name_thread("at-gloo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    runtime.name_thread("at-hip")
    real = args.real_rank is not None and rank == args.real_rank
    # host-only neighbours of a real rank may stay off the GPU altogether (--neighbours cpu): eight processes on ONE device share
    # its hardware queues, which a node with a GPU per rank does not
    on_gpu = args.device == "cuda" and (real or args.real_rank is None or args.neighbours == "cuda")
    dev = torch.device("cuda", 0) if on_gpu else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    if args.real_rank is not None:
        # the real rank's own world-size-1 nccl group (new_group is collective over the default group: every rank calls it)
        runtime.name_thread("at-nccl")
        solo = dist.new_group(ranks=[args.real_rank], backend="nccl") if args.device == "cuda" else None
    if real:
        return real_child(args, rank, world, dev, solo, dist)
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
        period = 1.0 / args.pace if args.pace > 0 else 0.0
        due = time.perf_counter()
        for batch, upcoming in pipe.pairs():
            t_end = time.perf_counter() + args.host_ms * 1e-3  # the host side of one replay: hipGraphLaunch + the loop
            while time.perf_counter() < t_end:
                pass
            spin += args.host_ms * 1e-3
            if period:  # a real rank is paced by its GPU: it asks for the next batch when the replay is done, not earlier
                due = max(due + period, time.perf_counter() - 4 * period)
                wait = due - time.perf_counter()
                if wait > 0:
                    time.sleep(wait)
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
        report(args, world, gathered, runtime)
    dist.barrier()
    dist.destroy_process_group()


def real_child(args, rank, world, dev, solo, dist):
    """The real rank: evaluate() over args.batches batches (its shard's worth), timed like the host-only ranks time theirs."""
    import torch
    from rpeflow_amd import evaluate as E
    from rpeflow_amd import runtime
    from rpeflow_amd.model import RPEFlow
    from rpeflow_amd.synthetic import SyntheticPairs, load_seeded_parameters
    runtime.name_thread("at-model")
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    forward = E.GraphedForward(model)
    n = args.batches * args.batch
    data = SyntheticPairs(n, args.height, args.width, 8192, distinct=args.distinct, cache=True, pin=args.pinned, first_seed=1000 + rank * args.distinct)
    t_gen = data.prepare(threads=2)
    workers = E.default_workers() if args.workers is None else args.workers
    warm = SyntheticPairs(3 * args.batch, args.height, args.width, 8192, distinct=args.distinct, first_seed=1000 + rank * args.distinct)
    warm.cache = data.cache
    runtime.name_thread("at-warm-up")
    E.evaluate(model, warm, args.batch, dev, 0, 1, group=solo, forward=forward, workers=workers)  # capture, MIOpen search, rings: untimed
    torch.cuda.synchronize()
    dist.barrier()  # everyone's set is generated, this rank's graph is captured: the host-only ranks warm their pipelines up now
    dist.barrier()  # the start line
    forward.host_times = []
    stats = {"timeline": True}  # device time per batch: forward, metric sums, and the gap before the next batch's first launch
    runtime.name_thread("rpe-main")
    threads0 = runtime.thread_cpu_seconds()
    t0, c0 = time.perf_counter(), time.process_time()
    metrics, _ = E.evaluate(model, data, args.batch, dev, 0, 1, group=solo, forward=forward, workers=workers, stats=stats)
    torch.cuda.synchronize()
    dt, cpu = time.perf_counter() - t0, time.process_time() - c0
    # where the cores go: CPU seconds per thread over the timed evaluation (threads that ended inside it -- the pipeline's own --
    # are read by the pipeline when it closes: stats["thread_cpu_s"])
    by_thread = {}
    for (tid, name), sec in runtime.thread_cpu_seconds().items():
        used = sec - threads0.get((tid, name), 0.0)
        if used > 0:
            by_thread[name] = by_thread.get(name, 0.0) + used
    for name, sec in stats.get("thread_cpu_s", {}).items():
        by_thread[name] = by_thread.get(name, 0.0) + sec
    cores_by_thread = {k: round(v / dt, 3) for k, v in sorted(by_thread.items(), key=lambda kv: -kv[1]) if v / dt >= 0.005}
    busiest = sorted(((sec - threads0.get(key, 0.0), key) for key, sec in runtime.thread_cpu_seconds().items()), reverse=True)[:6]
    busiest = [{"tid": key[0], "born": key[1], "cores": round(used / dt, 3)} for used, key in busiest if used / dt >= 0.005]
    launch = sorted(forward.host_times)
    pct = lambda q: round(launch[min(len(launch) - 1, int(q * len(launch)))] * 1e3, 3) if launch else None
    out = {"rank": rank, "real": True, "batches_per_s": round(args.batches / dt, 2), "ms_per_batch": round(dt / args.batches * 1e3, 3),
           "cpu_s_per_batch": round(cpu / args.batches, 5), "cores_in_use": round(cpu / dt, 2), "cores_by_thread": cores_by_thread, "busiest_surviving_threads": busiest,
           "loader_threads": stats.get("workers", workers),
           "h2d_MB_per_batch": round(stats["bytes"] / max(1, stats["batches"]) / 1e6, 1), "hipGraphLaunch_host_ms": {"p50": pct(0.5), "p99": pct(0.99), "max": pct(1.0)},
           "device_ms_per_batch": {k: stats.get("timeline_ms", {}).get(k) for k in ("forward", "accumulate", "gap", "host_loop")},
           "collective": "world-size-1 nccl group" if solo is not None else "none", "epe2d": metrics["EPE2D"], "generator_s_untimed": round(t_gen, 2)}
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        report(args, world, gathered, runtime)
    dist.barrier()
    dist.destroy_process_group()


def report(args, world, gathered, runtime):
    rates = [g["batches_per_s"] for g in gathered]
    real = [g for g in gathered if g.get("real")]
    print(json.dumps({
        "what": "host side of a %d-rank evaluation on one node, rehearsed on one box (tools/host_rehearsal.py)" % world,
        "ranks": world, "usable_cores": runtime.usable_cores(), "batch": args.batch, "frame": [args.height, args.width],
        "batches_per_rank": args.batches, "staging": "none (samples pinned)" if args.pinned else "pinned ring of host batches",
        "copy_fraction_host_only_ranks": float(os.environ.get("RPE_PIPE_COPY_FRACTION", "1")), "replay_stand_in_ms": args.host_ms,
        "host_only_ranks_paced_at_batches_per_s": args.pace or None, "host_only_ranks_on": args.neighbours if args.real_rank is not None else "cuda",
        "real_rank": real[0] if real else None,
        "batches_per_s_per_rank": rates, "min_batches_per_s": min(rates), "needed_batches_per_s": args.need,
        "feeds_the_gpus": bool(min(r for g, r in zip(gathered, rates) if not g.get("real")) >= args.need) if len(real) < world else None,
        "frame_pairs_per_s_capacity": round(sum(rates) * args.batch, 1),
        "cores_in_use_total": round(sum(g["cores_in_use"] for g in gathered), 2), "per_rank": gathered}), flush=True)


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
    p.add_argument("--pace", type=float, default=0.0, help="host-only ranks take at most this many batches per second, as ranks paced by a GPU do "
                                                            "(0: as fast as the host side goes -- the capacity measurement)")
    p.add_argument("--neighbours", choices=["cuda", "cpu"], default="cuda", help="with --real-rank: the host-only ranks keep their device ring and copy "
                                                                                 "stream on the (shared) GPU, or stay on the host: stage 1 only")
    p.add_argument("--real-rank", type=int, default=None, help="this rank runs the true evaluate() on the GPU; the others stay host-only")
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
        if args.real_rank is None:
            env.setdefault("RPE_PIPE_COPY_FRACTION", str(1.0 / args.ranks))
        elif r == args.real_rank:
            env["RPE_PIPE_COPY_FRACTION"] = "1"
            env["RPE_EVAL_TIMELINE"] = "1"  # (host seconds inside graph.replay() per batch)
        else:
            env.setdefault("RPE_PIPE_COPY_FRACTION", str(1.0 / 64))
        # own session per rank: a timeout (or Ctrl-C) of this launcher takes every rank's process group down with it
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=env, cwd=ROOT, start_new_session=True))
    rc = 0
    try:
        for proc in procs:
            rc = proc.wait() or rc
    finally:
        import signal
        for proc in procs:
            if proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    raise SystemExit(rc)


def _terminate(signum, frame):
    raise SystemExit(128 + signum)


if __name__ == "__main__":
    import signal
    signal.signal(signal.SIGTERM, _terminate)  # `timeout` sends SIGTERM: unwind through main()'s finally, which reaps the ranks
    main()
