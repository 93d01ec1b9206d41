#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of RPEFlow inference on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload ("forward", BASELINE config 3): a step = one full RPEFlow forward
(rpeflow_amd/model.py: every hot-path call on the HIP kernels, dense convs/attention on
PyTorch-ROCm) over a batch of B synthetic 544x960 frame pairs (RGB pair + 20-channel event
voxel + two 8192-point clouds), random-init weights, inputs resident in HBM.
--workload hotpath times the hot-path operator sequence alone (rpeflow_amd/hotpath.py).
Frame pairs are independent, so ranks shard them with no data-path collective (weak
scaling: B per GPU); the only collective is the MAX over ranks of the timed region.
Rank 0 prints ONE JSON line.

Extra objects on the line ("Measurement"):
  roofline       the dominant single-kernel category of the step: algorithmic bytes per
                 launch / average launch duration (HIP events inside the timed region)
  roofline_corr  BASELINE config 2, the correlation-only microbench 1x256x544x960 (N=1 only)
  cpu_baseline   the PyTorch-CPU port of the reference fallback (oracle/torch_ref.py) on the
                 host cores, bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import rpeflow_amd  # noqa: F401,E402 -- first: sets the HIP runtime's graph-queue count before the GPU is initialised

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
H, W, NPTS = 544, 960, 8192


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=4, help="frame pairs per GPU per step (conf/test/things.yaml batch_size)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--no-corr-microbench", action="store_true")
    p.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying HIP graphs")
    p.add_argument("--no-ahead", action="store_true", help="sample each batch's clouds inside its own forward instead of one batch ahead")
    p.add_argument("--workload", choices=["forward", "hotpath"], default="forward")
    return p.parse_args()


def algorithmic_bytes(workload, B):
    """Per-LAUNCH algorithmic bytes of each single-kernel category,
    averaged over the launches of that category in one step."""
    sizes, N = workload.sizes, [NPTS, 4096, 2048, 1024, 512, 256]
    C2 = [16, 32, 64, 96, 128, 192]
    out = {}
    # FPS(B,N,S): 12*B*N + 8*B*S ; one launch over 2B clouds
    out["fps+pyramid"] = 12 * 2 * B * NPTS + 8 * 2 * B * 4096
    # KNN(B,Q,M,D,k): 4*B*D*(Q+M) + 8*B*Q*k
    knn2d = [4 * B * 2 * (sizes[l][0] * sizes[l][1] + N[l]) + 8 * B * sizes[l][0] * sizes[l][1] for l in range(1, 6)]
    out["knn2d_k1"] = sum(2 * b for b in knn2d) / (2 * len(knn2d))
    knn3d = [4 * B * 3 * 2 * N[l] + 8 * B * N[l] * 16 for l in range(1, 6)]
    out["knn3d_k16"] = sum(knn3d) / len(knn3d)
    # correlation2d: 2*B*C*H*W*4 + B*81*H*W*4
    corr = [(2 * C2[l] + 81) * 4 * B * sizes[l][0] * sizes[l][1] for l in range(1, 6)]
    out["correlation2d"] = sum(corr) / len(corr)
    # backwarp_2d: 4*B*H*W*(2C+2), levels 4..1
    bw = [4 * B * sizes[l][0] * sizes[l][1] * (2 * C2[l] + 2) for l in range(1, 5)]
    out["backwarp_2d"] = sum(bw) / len(bw)
    return out


def corr_microbench(dev, iters=40):
    """BASELINE config 2: correlation2d 1x256x544x960, md=4, fp32, NCHW in/out."""
    import rpeflow_amd.csrc as ops
    a = torch.randn(1, 256, H, W, device=dev)
    b = torch.randn(1, 256, H, W, device=dev)
    for _ in range(120):  # the clocks need ~40 ms of sustained load to settle (first 20 launches: 500 us, then 405)
        ops.correlation2d(a, b, 4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.correlation2d(a, b, 4)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    alg = 2 * a.numel() * 4 + 81 * H * W * 4  # 1 238 753 280 B
    gbs = alg / us / 1e3
    traffic = None  # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this process)
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("corr_microbench_pmc.json"):
            traffic = int(json.load(open(os.path.join(ROOT, "profiles", name)))["derived"]["hbm_traffic_bytes"])
            break
    return {"kernel": "corr_mfma_dma_kernel<2,8,2,3,3,true>", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "us_per_launch": round(us, 1),
            "algorithmic_bytes": alg, "workload": "correlation2d 1x256x544x960 md=4 fp32 NCHW (BASELINE config 2)"}


def pmc_traffic(suffix):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this process)."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith(suffix):
            return int(json.load(open(os.path.join(ROOT, "profiles", name)))["derived"]["hbm_traffic_bytes"])
    return None


def usable_cores():
    """Cores this process may really use: affinity mask, capped by a cgroup CPU quota if there is one
    (os.cpu_count() reports the host's cores even inside a quota-limited container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def make_batch(B, device, first_seed=1000):
    from rpeflow_amd.synthetic import frame_pair
    samples = [frame_pair(first_seed + i, H, W, NPTS) for i in range(B)]
    return {k: torch.stack([torch.from_numpy(s[k]) for s in samples]).to(device) for k in samples[0]}


def cpu_baseline_worker(workload):
    """Child process (never touches the GPU): the reference's CPU/PyTorch fallback path, restated in
    oracle/torch_ref.py, on the host cores.  Sample: batch 1, full size, 1 untimed + 2 timed steps."""
    from types import SimpleNamespace
    from oracle import torch_ref
    from rpeflow_amd.hotpath import OP_NAMES, HotPathWorkload
    cores = usable_cores()
    torch.set_num_threads(cores)
    ops = SimpleNamespace(**{n: getattr(torch_ref, n) for n in OP_NAMES})
    if workload == "forward":
        from rpeflow_amd.model import RPEFlow
        torch.manual_seed(0)
        model = RPEFlow(ops=ops).eval()
        batch = make_batch(1, "cpu")
        step, what = (lambda: model(batch)), "full RPEFlow forward"
    else:
        wl = HotPathWorkload(batch=1, height=H, width=W, n_points=NPTS, device="cpu", ops=ops)
        step, what = wl, "hot-path operator sequence"
    step()
    steps = 2
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = (time.time() - t0) / steps
    print(json.dumps({"value": round(1.0 / dt, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                      "sample": f"{steps} timed steps (+1 warm-up) of the {what}, batch 1, 544x960 + 8192 pts, "
                                f"PyTorch-CPU port of the reference fallback path (matmul+topk KNN, Python-loop FPS, "
                                f"81-slice correlation, stock CPU convs/attention), {dt:.2f} s/step, "
                                f"host cpu_count={os.cpu_count()}"}))


def cpu_baseline(workload, timeout_s=420):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--workload", workload], capture_output=True,
                           text=True, timeout=timeout_s, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "frame-pairs/s", "cores": usable_cores(), "kind": "port",
                "sample": "worker failed: " + (r.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frame-pairs/s", "cores": usable_cores(), "kind": "port",
                "sample": f"worker exceeded {timeout_s} s"}


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.workload)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation in rpeflow_amd")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from rpeflow_amd import _lib
    from rpeflow_amd.hotpath import HotPathWorkload, SegmentGraphs, Timer
    _lib.lib()  # fail loudly now if librpeflow_hip.so is missing

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- hot-path sequence: per-category events (and the timed workload if --workload hotpath)
    wl = HotPathWorkload(batch=args.batch, height=H, width=W, n_points=NPTS, device=dev, seed=1000 + rank)
    for _ in range(max(args.warmup, 1)):
        wl()
    if args.eager:
        timer = Timer(True)
        hot_step = lambda: wl(timer)
    else:
        timer = SegmentGraphs()
        timer.capture(wl)          # one graph per span, shared pool
        timer.replay(timed=False)  # untimed replay: graph upload
        hot_step = timer.replay
    launch = "eager" if args.eager else "HIP graphs, one per operator span"
    dt_hot = timed(hot_step)
    dt = dt_hot

    # ---- full forward
    if args.workload == "forward":
        from rpeflow_amd.model import RPEFlow
        # (eval_withocc.py:159 sets cudnn.benchmark; measured here it buys <1 % and costs minutes of MIOpen search per process)
        torch.manual_seed(0)
        model = RPEFlow().to(dev).eval()
        batch = make_batch(args.batch, dev, first_seed=1000 + rank * args.batch)
        for _ in range(max(args.warmup, 1)):
            out = model(batch)
        torch.cuda.synchronize()
        assert torch.isfinite(out["flow_2d"]).all() and torch.isfinite(out["flow_3d"]).all(), "non-finite flow"
        fwd_step, launch = (lambda: model(batch)), "eager"
        if not args.eager:
            try:  # the whole forward as ONE HIP graph: ~3000 launches and the Python between them replayed in one call
                graph = torch.cuda.CUDAGraph()
                eager_out = {k: v.clone() for k, v in out.items()}
                # default: the evaluation harness's schedule (rpeflow_amd.evaluate.GraphedForward) -- the furthest-point
                # sampling of the FOLLOWING batch runs inside this batch's graph on its own stream, this batch starts from
                # the order the previous replay left; every replay still runs one full FPS over one batch of clouds
                order = None if args.no_ahead else model.sample_order(batch)
                # thread_local: a query from another thread (the RCCL watchdog of a multi-rank run) must not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = model(batch) if args.no_ahead else model.forward_ahead(batch, order, batch)
                for _ in range(max(args.warmup, 1)):  # untimed replays: graph upload, clocks back up after the capture
                    graph.replay()
                torch.cuda.synchronize()
                fwd_step, launch = graph.replay, "one HIP graph per forward" + (
                    "" if args.no_ahead else "; furthest-point sampling runs one batch ahead (the next batch's FPS inside this graph)")
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation, not a requirement
                torch.cuda.synchronize()
                launch = "eager (graph capture failed: %s)" % type(e).__name__
            if fwd_step == graph.replay:  # the replayed graph (two-stream branches included) must reproduce the eager forward
                for key in ("flow_2d", "flow_3d"):
                    err = (out[key] - eager_out[key]).abs().mean().item() / (eager_out[key].abs().mean().item() + 1e-6)
                    assert err < 1e-3, "graph replay differs from the eager forward: relative mean |d %s| = %g" % (key, err)
        dt = timed(fwd_step)

    totals = timer.totals_ms()
    if rank == 0:
        pairs = args.batch * args.steps * world
        breakdown = {k: round(v[0] / args.steps, 3) for k, v in sorted(totals.items(), key=lambda kv: -kv[1][0])}
        alg = algorithmic_bytes(wl, args.batch)
        single = {k: totals[k] for k in alg if k in totals}
        dom = max(single, key=lambda k: single[k][0])
        dom_us = single[dom][0] / single[dom][1] * 1e3
        gbs = alg[dom] / dom_us / 1e3
        line = {
            "metric": "frame-pairs/sec (544x960 + 8192 pts)",
            "value": round(pairs / dt, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("full RPEFlow forward (RGB pair + 20-ch event voxel + 2x8192 pts), random-init weights, "
                                    "FlyingThings3D val shapes (BASELINE config 3)") if args.workload == "forward" else
                                   ("RPEFlow hot path only (FPS, 43 KNN, correlation2d, warps, gathers, PointConv, Correlation3D) "
                                    "at FlyingThings3D shapes; dense 2D convs/attention excluded"),
                       "frame": [H, W], "points": NPTS, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "sharding": f"frame pairs over {world} rank(s), no data-path collective",
                       "launch": launch},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("fps_pmc.json") if dom == "fps+pyramid" else None,
                         "us_per_launch": round(dom_us, 1),
                         "launches_per_step": single[dom][1] // args.steps,
                         **({"note": "FPS is bound by the latency of 4095 DEPENDENT sampling iterations per cloud (SURVEY.md 8d), "
                                     "one workgroup per cloud on 2B of 256 CUs, not by HBM or MFMA: the HBM fraction is reported for "
                                     "form only; see us_per_iteration against the measured floor of the per-iteration "
                                     "reduce-barrier-broadcast chain (DESIGN.md section 4.3); roofline_corr is the bandwidth-bound kernel",
                             "us_per_iteration": round(dom_us / 4095, 4), "iteration_sync_floor_us": 0.41}
                            if dom == "fps+pyramid" else {})},
            "hotpath": {"frame_pairs_per_s": round(pairs / dt_hot, 3), "ms_per_step": round(dt_hot / args.steps * 1e3, 3),
                        "breakdown_ms_per_step": breakdown,
                        "note": "the hot-path operator sequence of one forward alone (rpeflow_amd/hotpath.py), timed in this run; "
                                "the roofline kernel is its dominant single-kernel category"},
        }
        if world == 1 and not args.no_corr_microbench:
            line["roofline_corr"] = corr_microbench(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
