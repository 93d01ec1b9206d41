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
Rank 0 prints ONE JSON line.  Started without a launcher (no WORLD_SIZE in the environment),
``--gpus N`` with N > 1 starts N fresh child processes itself, one per GPU, before this
process touches the GPU (cf. the reference's mp.spawn, train.py:289, 65).

Extra objects on the line ("Measurement"):
  roofline       the largest in-scope THROUGHPUT kernel of the step (pointconv_fused_kernel, level-1
                 estimator layer): SURVEY 8(d) flops per launch / average launch duration (HIP
                 events over back-to-back launches of that kernel) against the fp32 matrix peak
  roofline_fps   the step's longest kernel, furthest-point sampling: latency-bound, us per
                 dependent iteration against the synchronisation floor
  hotpath        the hot-path operator sequence alone, per category: ms, algorithmic bytes AND
                 flops, which roofline bounds it, floor = max(bytes / 8 TB/s, flops / 157.3 TF),
                 frac = floor / measured; roofline_frac = sum of floors / sum of durations
  epe_delta      |EPE2D|, |EPE3D| differences of THIS configuration's output (the replayed
                 graph, untimed) against the reference's CPU forward on the same batch
  roofline_corr  BASELINE config 2, the correlation-only microbench 1x256x544x960 (N=1 only)
  roofline_knn   the forward's largest 3-D neighbour search against the fp32 matrix peak, 2 D + 3
                 flops a pair (N=1 only)
  cpu_baseline   the PyTorch-CPU port of the reference fallback (oracle/torch_ref.py) on the
                 host cores, bounded sample (rank 0, N=1 only)
"""
import time

T_PROCESS = time.perf_counter()  # (before the heavy imports: a fresh box pages torch in for a minute or two, and that is start-up time)

import argparse  # noqa: E402
import gc  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from rpeflow_amd import runtime  # noqa: E402

RUNTIME = runtime.configure()  # before anything initialises the HIP runtime (graph-queue count; MIOpen solvers stay at defaults)

import torch  # noqa: E402

RUNTIME.update(torch=torch.__version__, hip=getattr(torch.version, "hip", None))  # what the queue count above was tuned on
try:
    RUNTIME["miopen"] = torch.backends.cudnn.version()
except Exception:  # noqa: BLE001 -- a build without MIOpen reports none
    RUNTIME["miopen"] = None

T_IMPORTED = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (SURVEY.md 8d prices KNN against it)
H, W, NPTS = 544, 960, 8192  # the correlation microbench's frame (BASELINE config 2) and the default workload's
# workload shapes: (frame H, W, batch per GPU, first sample seed, DSEC-style targets, golden of the reference's CPU forward)
CONFIGS = {
    "things": dict(H=544, W=960, batch=4, first_seed=1000, dsec=False, golden="model_bench_b4_544x960.npz",
                   name="FlyingThings3D val shapes (BASELINE config 3)"),
    "dsec": dict(H=480, W=640, batch=3, first_seed=2000, dsec=True, golden="model_bench_dsec_b3_480x640.npz",
                 name="DSEC eval shapes, conf/test/dsec.yaml (BASELINE config 5)"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", choices=list(CONFIGS), default="things", help="workload shapes: things = 544x960, batch 4 (the headline); dsec = 480x640, batch 3")
    p.add_argument("--batch", type=int, default=None, help="frame pairs per GPU per step (default: the configuration's: conf/test/*.yaml batch_size)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--no-corr-microbench", action="store_true")
    p.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying HIP graphs")
    p.add_argument("--no-ahead", action="store_true", help="sample each batch's clouds inside its own forward instead of one batch ahead")
    p.add_argument("--workload", choices=["forward", "eval", "hotpath", "selftest"], default="forward",
                   help="eval: the sharded evaluation end to end (loader threads, pinned ring, H2D on a copy stream, graph replay, metric "
                        "all-reduce), a step = one batch per rank; selftest: the rank launcher and timing protocol alone, on CPU tensors")
    p.add_argument("--eval-batches", type=int, default=64, help="batches per rank of the 'eval' leg that the forward workload reports as well (0: skip)")
    p.add_argument("--eval-workers", type=int, default=None, help="loader threads per rank (default: rpeflow_amd.evaluate.default_workers())")
    p.add_argument("--eval-pinned", action="store_true", help="(the default since round 6; kept for old command lines)")
    p.add_argument("--eval-pageable", action="store_true", help="hold the evaluation's cached synthetic set in pageable memory: loader threads then "
                   "stage every sample into the pinned ring (1.2 cores a rank at 60 batches/s; the default keeps the cached set pinned -- what "
                   "a registered / memory-mapped set or a dataset with load_into() gives -- and the H2D copies start from where the samples lie)")
    p.add_argument("--eval-raw-events", type=int, default=0, help="the evaluation's samples carry up to this many RAW events each ([n,4] float32, "
                   "as the reference's dataset loads them without a pre-processed file) instead of voxel grids; the input pipeline voxelises them on the device")
    p.add_argument("--eval-distinct", type=int, default=None, help="distinct samples of the evaluation's synthetic set (default: 16 per rank)")
    p.add_argument("--no-staggered-warmup", action="store_true", help="N > 1: every rank runs its first forward (MIOpen's solver search) at once "
                   "instead of rank 0 first and the others behind a barrier")
    p.add_argument("--share-gpu", action="store_true", help=argparse.SUPPRESS)  # tests: all ranks on the visible GPU(s), collective on gloo
    p.add_argument("--backend", choices=["nccl", "gloo", "none"], default="nccl",
                   help="process-group backend; nccl is RCCL.  A single rank joins a world-size-1 nccl group as well, so the one "
                        "collective of the evaluation and the timing protocol's barrier / MAX run on RCCL in every run; 'none' "
                        "(single rank only) skips the group")
    return p.parse_args()


NOMINAL_SCLK_MHZ = 2400.0   # MI355X_MICROARCH.md: peak engine clock
HBM_COPY_GBS = 6290.0       # MI355X_MICROARCH.md: what a device-to-device copy reaches of the 8 TB/s (the guide's figure; the run measures its own beside it)


def pmc_summary(suffix):
    """The newest PMC summary committed under profiles/ whose name ends in ``suffix`` (rocprofv3 cannot run inside this process;
    the passes are `tools/profile_round.sh`'s, collected as MI355X_MICROARCH.md prescribes), and its path; (None, None) without."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith(suffix):
            return json.load(open(os.path.join(ROOT, "profiles", name))), "profiles/" + name
    return None, None


def pmc_traffic(suffix):
    """(HBM bytes per launch, file): FETCH_SIZE with the guide's gfx950 correction + WRITE_SIZE."""
    summary, source = pmc_summary(suffix)
    return (int(summary["derived"]["hbm_traffic_bytes"]), source) if summary else (None, None)


def clocked(frac, clock, dev, launches=0):
    """Extra fields of a roofline object: the clock the shader engines ran at DURING the timed launches -- two clock stamps on the
    stream around them, every compute unit's cycle counter compared with itself only (rpeflow_amd.runtime.ShaderClock), else the
    hwmon reading of this very device, else null -- and the fraction scaled to the nominal clock, null unless the clock is
    plausible and the result <= 1."""
    sclk, source = clock.mhz(), "two per-compute-unit clock stamps around the timed launches (median over the units of d s_memtime / d s_memrealtime x the constant rate)"
    if sclk is None:
        sclk, source = runtime.hwmon_sclk_mhz(dev), "hwmon sclk of this device, one reading after the loop"
    at_nominal = round(frac * NOMINAL_SCLK_MHZ / sclk, 4) if sclk else None
    cycles, ticks, khz = clock.raw()
    return {"sclk_MHz": sclk, "sclk_source": source if sclk else None, "sclk_MHz_per_xcd": clock.mhz_per_xcd(), "sclk_units_read": clock.units(),
            "sclk_stamp_raw": {"shader_cycles": cycles, "wall_ticks": ticks, "wall_kHz": khz},
            "shader_cycles_per_launch": round(cycles / launches) if (sclk and launches) else None,  # (the median unit; compare: GRBM_GUI_ACTIVE / 8 of the PMC pass)
            "frac_at_nominal_clock": at_nominal if (at_nominal is not None and at_nominal <= 1.0) else None}


def corr_microbench(dev, iters=40):
    """BASELINE config 2: correlation2d 1x256x544x960, md=4, fp32, NCHW in/out."""
    import rpeflow_amd.csrc as ops
    a = torch.randn(1, 256, H, W, device=dev)
    b = torch.randn(1, 256, H, W, device=dev)
    for _ in range(120):  # the clocks need ~40 ms of sustained load to settle (first 20 launches: 500 us, then 405)
        ops.correlation2d(a, b, 4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    clock = runtime.ShaderClock(dev)
    s.record()
    with clock:  # the clock stamps sit inside the event pair, around the very launches that are timed
        for _ in range(iters):
            ops.correlation2d(a, b, 4)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3  # (the two one-wave stamp kernels inside the pair: ~4 us over >= 5 ms of launches)
    alg = 2 * a.numel() * 4 + 81 * H * W * 4  # 1 238 753 280 B
    gbs = alg / us / 1e3
    s.record()  # what a plain device-to-device copy of one operand reaches on THIS box right now (read + write bytes)
    for _ in range(10):
        b.copy_(a)
    e.record()
    torch.cuda.synchronize()
    copy_gbs = 2 * a.numel() * 4 / (s.elapsed_time(e) / 10) / 1e6
    traffic, source = pmc_traffic("corr_microbench_pmc.json")
    frac = gbs / HBM_PEAK_GBS
    return {"kernel": "corr_mfma_dma_kernel<2,8,2,3,3,true>", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(frac, 4), "traffic": traffic, "traffic_source": source, "us_per_launch": round(us, 1),
            **clocked(frac, clock, dev, iters), "copy_GBs_measured": round(copy_gbs, 1), "frac_of_measured_copy_bw": round(gbs / copy_gbs, 4),
            "frac_of_guide_copy_bw": round(gbs / HBM_COPY_GBS, 4),
            "algorithmic_bytes": alg, "workload": "correlation2d 1x256x544x960 md=4 fp32 NCHW (BASELINE config 2)"}


def knn_microbench(dev, iters=30):
    """SURVEY.md 8(d) prices k_nearest_neighbor against the fp32 matrix peak: a pair costs 2 D + 3 flops.  The forward's
    largest 3-D search, 8192 -> 4096 points, k = 16, both frames of the batch of 4 (B = 8), default tie mode."""
    import rpeflow_amd.csrc as ops
    g = torch.Generator(device="cpu").manual_seed(0)
    B, M, Q, D, k = 8, NPTS, NPTS // 2, 3, 16
    cloud = (torch.rand(B, D, M, generator=g) * 30).to(dev)
    query = cloud[:, :, :Q].contiguous()
    for _ in range(10):
        ops.k_nearest_neighbor(cloud, query, k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.k_nearest_neighbor(cloud, query, k)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    # (no engine-clock fields here: the stamps read XCD 0's cycle counter, which stands still while that XCD idles -- the tail of
    # the tied rows' replay launch keeps a few waves busy and the rest of the chip gated: 1430 "MHz" was read over these launches.
    # The correlation loop keeps every XCD busy from the first stamp to the second; there the reading is the clock.)
    from rpeflow_amd.csrc.wrapper import k_nearest_neighbor_ties
    for _ in range(5):
        k_nearest_neighbor_ties(cloud, query, k, ties="index")
    s.record()
    for _ in range(iters):
        k_nearest_neighbor_ties(cloud, query, k, ties="index")
    e.record()
    torch.cuda.synchronize()
    us_index = s.elapsed_time(e) / iters * 1e3
    pairs = B * M * Q
    tflops = pairs * (2 * D + 3) / us / 1e6
    traffic, source = pmc_traffic("knn16_pmc.json")  # the sweep kernel ...
    replay_traffic, replay_source = pmc_traffic("knn16_replay_pmc.json")  # ... and the tied rows' second launch
    if traffic is not None and replay_traffic is not None and replay_source[:12] == source[:12]:  # (same round's passes)
        traffic, source = traffic + replay_traffic, source + " + " + replay_source
    frac = tflops / MFMA_F32_PEAK_TFLOPS
    return {"kernel": "knn_mfma_kernel<3, true> + knn_tie_replay_kernel<3> (the sweep, then its tied rows redone the libstdc++ way by a second launch)",
            "bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(frac, 4),
            "traffic": traffic, "traffic_source": source,
            "us_per_launch": round(us, 1), "launches": 2, "pairs_per_s": round(pairs / us * 1e6), "algorithmic_bytes": 4 * B * D * (M + Q) + 8 * B * Q * k,
            "us_per_launch_lowest_index_ties": round(us_index, 1),  # the same search without the libstdc++ restatement of equal distances
            "frac_lowest_index_ties": round(pairs * (2 * D + 3) / us_index / 1e6 / MFMA_F32_PEAK_TFLOPS, 4),
            "workload": "k_nearest_neighbor 3-D, 8 x (8192 -> 4096), k = 16, fp32, indices as torch.topk returns them"}


def pointconv_microbench(dev, iters=40):
    """The largest in-scope THROUGHPUT kernel of the step: FlowEstimator3D's first PointConv at pyramid level 1
    (pwc3d_core.py:123,140 -> pointconv.py:100-119): B = 4, N = 4096, k = 16, 195 + 3 input channels -> 128, one launch of
    pointconv_fused_kernel<1,2,4,2,1> (gather, weight net, the 16 x 198 weighted sums, nn.Linear 3168 -> 128, bias, activation).
    Flops per launch = SURVEY 8(d)'s PointConv formula (rpeflow_amd.roofline.pointconv_flops); the rows and the neighbour table
    are prepared outside the timed launches, which are that one kernel and nothing else."""
    from rpeflow_amd import pointconv as PC
    from rpeflow_amd import roofline
    import rpeflow_amd.csrc as ops
    B, N, C, Cout, k = 4, 4096, 195, 128, 16
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz = (torch.rand(B, 3, N, generator=g) * 30).to(dev)
    feat = torch.randn(B, C, N, generator=g).to(dev)
    torch.manual_seed(0)
    layer = PC.PointConvNoSampling(C, Cout, norm=None, k=k).to(dev).eval()
    with torch.no_grad():
        knn = ops.k_nearest_neighbor(xyz, xyz, k)
        rows = PC.pack_rows(xyz, feat)
        for _ in range(10):
            layer(xyz, rows, knn)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            layer(xyz, rows, knn)
        e.record()
        torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    flops = roofline.pointconv_flops(B, N, C, Cout, k)
    alg = roofline.pointconv(B, N, N, C, Cout, False)
    tflops = flops / us / 1e6
    traffic, source = pmc_traffic("pointconv_pmc.json")
    summary, _ = pmc_summary("pointconv_pmc.json")
    # FETCH_SIZE's doubling on gfx950 is calibrated for 16-byte-per-lane streams; this kernel's reads are 64-byte gathered row
    # pieces and 1-KiB weight fragments from L2: the corrected figure is an upper bound, the counter as read a lower one
    as_counted = int(summary["counters_mean_per_launch"]["FETCH_SIZE"] * 1024 + summary["derived"]["write_bytes"]) if summary else None
    return {"kernel": "pointconv_fused_kernel<1,2,4,2,1> (PointConvNoSampling 195 -> 128 over 4 x 4096 points, k = 16: FlowEstimator3D.point_conv1 at pyramid level 1)",
            "bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_as_counted": as_counted, "traffic_source": source, "us_per_launch": round(us, 1), "launches_per_step": 1,
            "algorithmic_flops": flops, "algorithmic_bytes": alg, "hbm_GBs": round(alg / us / 1e3, 1),
            "note": "the largest in-scope throughput kernel of the step (rocprof: 180 us a launch, two launches a step with its 128 -> 128 twin), "
                    "timed here by HIP events over %d back-to-back launches of that kernel alone.  fp32 MFMA (v_mfma_f32_16x16x4_f32) against the "
                    "dense fp32 matrix peak; arithmetic intensity %.0f flop/B, far right of the ridge.  The step's longest single kernel, "
                    "furthest-point sampling, is latency-bound and reported as roofline_fps" % (iters, flops / alg)}


usable_cores = runtime.usable_cores  # affinity mask capped by the cgroup CPU quota (the GPU box grants 16 of 256 cores)


def make_batch(B, device, first_seed=1000, H=H, W=W, dsec=False):
    from rpeflow_amd.synthetic import frame_pair
    samples = [frame_pair(first_seed + i, H, W, NPTS, dsec=dsec) for i in range(B)]
    return {k: torch.stack([torch.from_numpy(s[k]) for s in samples]).to(device) for k in samples[0]}


def golden_epe_delta(out, batch, golden):
    """|EPE - reference EPE| of a forward over the benched batch: EPE2D / EPE3D against the synthetic targets, compared with
    the EPEs of the reference's CPU forward on the same batch and parameters (tests/golden/model_bench_b4_544x960.npz, made
    by tests/golden/make_golden.py model_bench), plus the mean absolute flow differences."""
    import numpy as np
    f2, f3 = out["flow_2d"].float().cpu().numpy(), out["flow_3d"].float().cpu().numpy()
    t2, t3 = batch["flow_2d"].cpu().numpy()[:, :2], batch["flow_3d"].cpu().numpy()[:, :3]
    epe2 = float(np.sqrt(((f2 - t2) ** 2).sum(1)).mean())
    epe3 = float(np.sqrt(((f3 - t3) ** 2).sum(1)).mean())
    return {"epe2d": abs(epe2 - float(golden["epe2d"])), "epe3d": abs(epe3 - float(golden["epe3d"])),
            "mean_abs_flow_2d": float(np.abs(f2[:, :, ::8, ::8] - golden["flow_2d_s8"]).mean()),
            "mean_abs_flow_3d": float(np.abs(f3 - golden["flow_3d"]).mean())}


def cpu_baseline_worker(workload, config="things"):
    """Child process (never touches the GPU): the reference's CPU/PyTorch fallback path, restated in
    oracle/torch_ref.py, on the host cores.  Sample: batch 4 (the benched batch), full size, 1 untimed + 3 timed steps."""
    from types import SimpleNamespace
    from oracle import torch_ref
    from rpeflow_amd.hotpath import OP_NAMES, HotPathWorkload
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = CONFIGS[config]
    H, W, CPU_BATCH = cfg["H"], cfg["W"], cfg["batch"]  # noqa: N806
    ops = SimpleNamespace(**{n: getattr(torch_ref, n) for n in OP_NAMES})
    if workload == "forward":
        from rpeflow_amd.model import RPEFlow
        from rpeflow_amd.synthetic import load_seeded_parameters
        model = load_seeded_parameters(RPEFlow(ops=ops)).eval()
        batch = make_batch(CPU_BATCH, "cpu", first_seed=cfg["first_seed"], H=H, W=W, dsec=cfg["dsec"])
        step, what = (lambda: model(batch)), "full RPEFlow forward"
    else:
        wl = HotPathWorkload(batch=CPU_BATCH, height=H, width=W, n_points=NPTS, device="cpu", ops=ops)
        step, what = wl, "hot-path operator sequence"
    step()
    steps = 3
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = (time.time() - t0) / steps
    print(json.dumps({"value": round(CPU_BATCH / dt, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                      "sample": f"{steps} timed steps (+1 warm-up) of the {what}, batch {CPU_BATCH}, {H}x{W} + 8192 pts, "
                                f"PyTorch-CPU port of the reference fallback path (matmul+topk KNN, Python-loop FPS, "
                                f"81-slice correlation, stock CPU convs/attention), {dt:.2f} s/step, "
                                f"host cpu_count={os.cpu_count()}"}))


def cpu_baseline(workload, config="things", timeout_s=420):
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--workload", workload, "--config", config], capture_output=True,
                           text=True, timeout=timeout_s, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "frame-pairs/s", "cores": usable_cores(), "kind": "port",
                "sample": "worker failed: " + (r.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frame-pairs/s", "cores": usable_cores(), "kind": "port",
                "sample": f"worker exceeded {timeout_s} s"}


def eval_leg(model, forward, dev, cfg, batch_size, n_batches, rank, world, dist, workers=None, pinned=False, backend="nccl", distinct=None, raw_events=0):
    """The sharded evaluation END TO END (rpeflow_amd.evaluate.evaluate, the counterpart of eval_withocc.py:43-135): every
    rank reads its shard of a cached synthetic set (frame pairs r, r + W, ...) through the input pipeline -- loader threads
    into pinned host batches, H2D on a copy stream, the forward replayed from the HIP graph with the next batch's sampling
    inside -- accumulates the 12 metric sums on the device and joins ONE float64[12] SUM all-reduce on the real backend.
    Timed from a barrier to the metrics being on the host; MAX over ranks.  The generator is not timed (0.3 s of numpy per
    sample: the set is generated once and read from memory, as a dataset on disk is read from the page cache)."""
    from rpeflow_amd import evaluate as E
    from rpeflow_amd.synthetic import SyntheticPairs
    per_rank = 4 * batch_size  # distinct samples a rank cycles through (0.21 GB a batch)
    n = world * n_batches * batch_size
    distinct = world * per_rank if distinct is None else distinct
    data = SyntheticPairs(n, cfg["H"], cfg["W"], NPTS, dsec=cfg["dsec"], distinct=distinct, cache=True, pin=pinned,
                          first_seed=cfg["first_seed"], events=raw_events)
    mine = E.shard_indices(n, rank, world)
    t_gen = data.prepare(indices=mine)
    warm = SyntheticPairs(world * 3 * batch_size, cfg["H"], cfg["W"], NPTS, dsec=cfg["dsec"], distinct=distinct, first_seed=cfg["first_seed"], events=raw_events)
    warm.cache, warm.pin = data.cache, pinned  # same samples: the loader threads, the rings and the graph get their first use untimed
    group = dist.group.WORLD if dist is not None else None  # explicit: evaluate() joins a collective only on its caller's say-so
    E.evaluate(model, warm, batch_size, dev, rank, world, group=group, forward=forward, workers=workers)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    stats = {"timeline": bool(os.environ.get("RPE_EVAL_TIMELINE"))}  # diagnostic: per-batch device times (adds 3 event records a batch)
    t0 = time.perf_counter()
    metrics, _ = E.evaluate(model, data, batch_size, dev, rank, world, group=group, forward=forward, workers=workers, stats=stats)  # ends on the host: finalize() reads the sums
    mine_dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    dts = [mine_dt]
    if dist is not None:
        t = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        t[rank] = mine_dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dts = [float(x) for x in t.tolist()]
    dt = max(dts)
    return dt, {
        "frame_pairs_per_s": round(n / dt, 3), "ms_per_batch": round(dt / n_batches * 1e3, 3), "batches_per_rank": n_batches,
        "world_size": dist.get_world_size() if dist is not None else 1,
        "per_rank_frame_pairs_per_s": [round(len(E.shard_indices(n, r, world)) / d, 3) for r, d in enumerate(dts)],
        "h2d_GBps_per_rank": round(stats["bytes"] / mine_dt / 1e9, 2), "h2d_MB_per_batch": round(stats["bytes"] / max(1, stats["batches"]) / 1e6, 1),
        "loader": {"threads": stats["workers"], "staging": "none: samples lie in pinned memory" if stats["direct"] else "pinned ring of host batches",
                   "copy": "dedicated HIP stream, one batch ahead of the replay", "copy_stream_probe_ms": stats.get("copy_stream_probe_ms"), "generator_s_untimed": round(t_gen, 2),
                   "distinct_samples": distinct,
                   **({"events": "raw: up to %d float32 events a sample, voxelised on the device by the copy stage (10 bins x 2 polarities)" % raw_events} if raw_events else {})},
        "collective": "one SUM all-reduce of float64[12] (%s)" % (backend if dist is not None else "single rank: none"),
        "metrics": {k: (float("%.12g" % v) if isinstance(v, float) else v) for k, v in metrics.items() if k != "counts"},
        "samples": metrics["counts"]["3d"] / NPTS,
        **({"timeline_ms": stats["timeline_ms"]} if "timeline_ms" in stats else {}), **({"trace": stats["trace"]} if "trace" in stats else {}),
    }


def launch_ranks(n_ranks, argv):
    """``--gpus N`` without a launcher: start N fresh processes of this script, one per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, the reference's mp.spawn pattern of train.py:289, 65), before this
    process has initialised the GPU (it never does).  Rank 0 prints the JSON line; the exit code is the worst child's."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        # LOCAL_WORLD_SIZE: all ranks are on this node and share its cores -- runtime.configure() in the child gives each its
        # share of OpenMP threads (this process's own OMP_NUM_THREADS, chosen for a single rank, carries the marker that lets
        # the child re-derive it) and rpeflow_amd.evaluate.default_workers() its share of loader threads
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT))
    worst = 0
    while procs:
        for proc in list(procs):
            rc = proc.poll()
            if rc is None:
                continue
            procs.remove(proc)
            if rc != 0:  # one rank failed: the others would wait in a collective for ever
                worst = worst or rc
                for other in procs:
                    other.terminate()
        time.sleep(0.05)
    return worst


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.workload, args.config)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    on_gpu = args.workload != "selftest"
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation in rpeflow_amd")
    if not on_gpu and args.backend != "gloo":
        raise SystemExit("--workload selftest runs on CPU tensors: use --backend gloo")
    # The line on stdout is the contract: ONE JSON line.  Libraries write there too (RCCL prints a five-line version banner to
    # stdout when its first communicator comes up), so file descriptor 1 is pointed at stderr for the run and the line goes to
    # the saved descriptor.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(line) + "\n").encode())

    dist = None
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if args.share_gpu else local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(dev)
    if args.backend == "none" and world > 1:
        raise SystemExit("--backend none is for a single rank")
    if world > 1 or (on_gpu and args.backend == "nccl"):
        # One rank joins a process group too (world size 1 is legal): RCCL is loaded, its watchdog thread runs beside the HIP-graph
        # capture, and the evaluation's float64[12] SUM all-reduce and the timing protocol's barrier / MAX execute on the real
        # backend in every run -- what a multi-GPU launch does, short of the xGMI transport (train.py:65 is the reference's set-up).
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        dist.init_process_group(args.backend, rank=rank, world_size=world, **({"device_id": dev} if (on_gpu and args.backend == "nccl") else {}))
        assert dist.get_world_size() == args.gpus

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    def timed(step):
        marks = [] if (on_gpu and os.environ.get("RPE_BENCH_STEP_TIMES")) else None  # diagnostic: per-step device times to stderr
        # The host runs one or two replays ahead of the device (a graph launch blocks beyond that), so a host pause longer than a
        # step shows in the rate, and a full garbage collection of this process (200 k tracked objects) is a pause of 38-67 ms,
        # measured here: one default run in ~20 read 15.9 instead of 15.0 ms with its evaluation leg right behind it at 15.5 as
        # always.  Collect now, not inside the K steps.
        t_gc = time.perf_counter()
        gc.collect()
        if marks is not None:
            print("full collection: %.1f ms, %d objects tracked" % ((time.perf_counter() - t_gc) * 1e3, len(gc.get_objects())), file=sys.stderr, flush=True)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            barrier()
            t0 = time.perf_counter()
            host = []  # when each step() returned, on the host's clock: free of charge, and a stalled launch shows as a gap
            for _ in range(args.steps):
                if marks is not None:
                    marks.append(torch.cuda.Event(enable_timing=True))
                    marks[-1].record()
                step()
                host.append(time.perf_counter())
            if marks is not None:
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
            barrier()
            dt = time.perf_counter() - t0
        finally:
            if gc_was_on:
                gc.enable()
        if marks is not None:
            print("step ms:", [round(a.elapsed_time(b), 2) for a, b in zip(marks[:-1], marks[1:])], file=sys.stderr, flush=True)
        if on_gpu and len(host) > 2:
            gaps = [round((b - a) * 1e3, 2) for a, b in zip([t0] + host[:-1], host)]
            if max(gaps) > 6 * dt / args.steps * 1e3 or os.environ.get("RPE_BENCH_HOST_GAPS"):  # (a launch that blocks on a full queue waits 2-3 steps)
                print("host ms between step() returns (a gap far above the step time is a stalled launch): %s; drain %.2f" %
                      (gaps, (t0 + dt - host[-1]) * 1e3), file=sys.stderr, flush=True)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def finish(line):
        if rank == 0:
            emit(line)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if not on_gpu:  # launcher + timing protocol only
        a = torch.randn(64, 64)
        for _ in range(args.warmup):
            a @ a
        dt = timed(lambda: a @ a)
        return finish({"metric": "selftest steps/s (launcher and timing protocol only, CPU tensors)", "value": round(args.steps * world / dt, 3),
                       "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                       "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "selftest", "backend": args.backend}})

    from rpeflow_amd import _lib, roofline
    from rpeflow_amd.hotpath import HotPathWorkload, SegmentGraphs, Timer
    _lib.lib()  # fail loudly now if librpeflow_hip.so is missing
    cfg = CONFIGS[args.config]
    H, W = cfg["H"], cfg["W"]  # noqa: N806 -- shadow the module's headline shape inside this run
    if args.batch is None:
        args.batch = cfg["batch"]

    # ---- full forward.  It is captured and timed BEFORE the hot-path sequence below: with the ~70 segment graphs of that
    # sequence already alive in the process, every other run's forward graph replayed 8 % slower (202-207 instead of
    # 218 frame-pairs/s, 4 of 8 runs; never in 30 captures without them) -- whatever the runtime derives its queue
    # mapping from depends on what was captured before.
    epe_delta, dt, launch, step_clock = None, None, None, None
    eval_info = None
    startup = {"imports_s": round(T_IMPORTED - T_PROCESS, 2)}  # this rank's way to its first step, seconds since its process started
    if args.workload in ("forward", "eval"):
        from rpeflow_amd.model import RPEFlow
        from rpeflow_amd.synthetic import load_seeded_parameters
        # (eval_withocc.py:159 sets cudnn.benchmark; measured here it buys <1 % and costs minutes of MIOpen search per process)
        model = load_seeded_parameters(RPEFlow()).to(dev).eval()  # the parameters of the committed model goldens
        batch = make_batch(args.batch, dev, first_seed=cfg["first_seed"] + rank * args.batch, H=H, W=W, dsec=cfg["dsec"])
        startup["model_and_batch_s"] = round(time.perf_counter() - T_PROCESS, 2)
        # The first forward of a process runs MIOpen's solver search for every convolution shape and writes what it TIMED to the
        # user's find database (~/.config/miopen), which every rank of the node shares.  Rank 0 goes first, alone; the others wait
        # at a barrier and then read its results instead of searching: one search per node instead of one per rank (22.5 s to the
        # first step with eight ranks searching at once on a fresh box), every rank on the same solvers, and no search timed while
        # other ranks load the machine -- with eight ranks sharing ONE GPU (--share-gpu) the contended timings picked solvers that
        # left the database 10 % slower for every later process on the box (16.6 instead of 15.0 ms per step, measured).
        staggered = dist is not None and world > 1 and not args.no_staggered_warmup
        if staggered and rank != 0:
            dist.barrier()
            startup["waited_for_rank0_s"] = round(time.perf_counter() - T_PROCESS, 2)
        for i in range(max(args.warmup, 1)):
            out = model(batch)
            if i == 0:  # HIP module loads + the solver search (rank 0) or the database reads (the others)
                sync()
                startup["first_step_s"] = round(time.perf_counter() - T_PROCESS, 2)
                if staggered and rank == 0:
                    dist.barrier()
        sync()
        assert torch.isfinite(out["flow_2d"]).all() and torch.isfinite(out["flow_3d"]).all(), "non-finite flow"
        fwd_step, launch, forward = (lambda: model(batch)), "eager", None
        if not args.eager:
            # the whole forward as ONE HIP graph: ~1100 launches and the Python between them replayed in one call.  A failed
            # capture is an error, not a reason to time something else.  The graph is the evaluation harness's own
            # (rpeflow_amd.evaluate.GraphedForward: static input buffers; the furthest-point sampling of the FOLLOWING batch
            # runs inside this batch's graph on its own stream, this batch starts from the order the previous replay left --
            # every replay still runs one full FPS over one batch of clouds); the 'eval' leg below feeds the same graph
            # from the input pipeline.
            from rpeflow_amd.evaluate import GraphedForward
            eager_out = {k: v.clone() for k, v in out.items()}
            forward = GraphedForward(model, warmup=0, ahead=not args.no_ahead)
            out = forward(batch, batch)  # capture + first replay (graph upload); announces the same batch again: steady state
            sync()
            startup["graph_captured_s"] = round(time.perf_counter() - T_PROCESS, 2)
            entry = forward.entries[forward._key(batch)]
            fwd_step, launch = entry["graph"].replay, "one HIP graph per forward" + (
                "" if args.no_ahead else "; furthest-point sampling runs one batch ahead (the next batch's FPS inside this graph)")
            for key in ("flow_2d", "flow_3d"):  # the replayed graph (multi-stream branches included) must reproduce the eager forward
                err = (out[key] - eager_out[key]).abs().mean().item() / (eager_out[key].abs().mean().item() + 1e-6)
                assert err < 1e-3, "graph replay differs from the eager forward: relative mean |d %s| = %g" % (key, err)
        golden = os.path.join(ROOT, "tests", "golden", cfg["golden"])
        if rank == 0 and args.batch == cfg["batch"] and os.path.exists(golden):  # untimed: this configuration's output vs the reference's
            import numpy as np
            d = golden_epe_delta(out, batch, np.load(golden))
            epe_delta = {"epe2d": float("%.3g" % d["epe2d"]), "epe3d": float("%.3g" % d["epe3d"]), "bound": 1e-4,
                         "within_bound": bool(d["epe2d"] < 1e-4 and d["epe3d"] < 1e-4),  # off the bound: the line is still printed, the exit code is 3
                         "against": "the reference's CPU forward on this batch and these parameters (tests/golden/%s)" % cfg["golden"]}
        # Untimed warm-up steps, immediately in front of the timed region: --warmup of them at least, and enough of them
        # for the clocks to be back up -- the chip drops them within tens of milliseconds of idling (the host-side parity
        # check above is such a pause) and needs ~50 ms of load to recover; without this the first timed steps of about
        # every other run went 5-8 % slow.  Up to 40 replays / 0.8 s: about one process in five, always the first on a
        # fresh machine, has ONE replay of ~48 ms instead of 18 among its first ~20 (per-step times, RPE_BENCH_STEP_TIMES=1;
        # same solvers, same numbers: a one-off stall of the runtime, not of a kernel), which read as 205 instead of 222.
        gc.collect()
        gc.freeze()  # (what exists now -- model, graphs, caches -- is permanent: later collections walk the new objects only)
        t_settle = time.perf_counter()
        n_replays = 0
        while n_replays < max(args.warmup, 1) or (n_replays < 40 and time.perf_counter() - t_settle < 0.8):
            fwd_step()
            n_replays += 1
            if n_replays % 4 == 0:
                sync()
        if args.workload == "forward":
            step_clock = runtime.ShaderClock(dev)
            with step_clock:  # one stamp launch in front of and one behind the barriers that bracket the K steps: outside the timed region
                dt = timed(fwd_step)
        n_eval = args.steps if args.workload == "eval" else args.eval_batches
        if forward is not None and n_eval > 0:
            dt_eval, eval_info = eval_leg(model, forward, dev, cfg, args.batch, n_eval, rank, world, dist, workers=args.eval_workers,
                                          pinned=not args.eval_pageable, backend=args.backend, distinct=args.eval_distinct, raw_events=args.eval_raw_events)
            if args.workload == "eval":
                dt = dt_eval
            else:
                eval_info["vs_forward"] = round(eval_info["frame_pairs_per_s"] / (args.batch * args.steps * world / dt), 4)
        elif args.workload == "eval":
            raise SystemExit("--workload eval replays HIP graphs: not available with --eager")

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
        for _ in range(4):  # untimed replays: graph upload, clocks back up after the capture
            timer.replay(timed=False)
        hot_step = timer.replay
    if args.workload == "hotpath":
        launch = "eager" if args.eager else "HIP graphs, one per operator span"
    dt_hot = timed(hot_step)
    if args.workload == "hotpath":
        dt = dt_hot

    totals = timer.totals_ms()
    line = None
    startups = [startup]
    if dist is not None and world > 1:  # every rank's start-up account, for rank 0's line
        startups = [None] * world
        dist.all_gather_object(startups, startup)
    if rank == 0:
        pairs = args.batch * args.steps * world
        alg = roofline.hotpath_bytes(args.batch, wl.sizes, NPTS)
        alg_flops = roofline.hotpath_flops(args.batch, wl.sizes, NPTS)
        table, sum_b, sum_f, sum_ms, sum_floor, sum_ms_tp, sum_floor_tp = {}, 0, 0, 0.0, 0.0, 0.0, 0.0
        for k, (ms, n) in sorted(totals.items(), key=lambda kv: -kv[1][0]):
            ms_step = ms / args.steps
            b, f = alg.get(k, 0), alg_flops.get(k, 0)
            floor_s, bound = roofline.floor_seconds(b, f)
            latency = k in roofline.LATENCY_BOUND
            table[k] = {"ms": round(ms_step, 3), "algorithmic_MB": round(b / 1e6, 2), "algorithmic_GFLOP": round(f / 1e9, 3),
                        "bound": "latency" if latency else bound, "floor_ms": round(floor_s * 1e3, 4),
                        "frac": round(floor_s / (ms_step * 1e-3), 4) if ms_step > 0 else None,
                        "hbm_frac": round(b / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_step > 0 else None}
            sum_b, sum_f, sum_ms, sum_floor = sum_b + b, sum_f + f, sum_ms + ms_step, sum_floor + floor_s
            if not latency:
                sum_ms_tp, sum_floor_tp = sum_ms_tp + ms_step, sum_floor_tp + floor_s
        fps_us = totals["fps+pyramid"][0] / totals["fps+pyramid"][1] * 1e3
        fps_bytes = roofline.fps(2 * args.batch, NPTS, 4096)
        line = {
            "metric": "frame-pairs/sec (%dx%d + 8192 pts)" % (H, W),
            "value": round(pairs / dt, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("full RPEFlow forward (RGB pair + 20-ch event voxel + 2x8192 pts), seeded random-init weights, "
                                    + cfg["name"]) if args.workload == "forward" else
                                   ("sharded evaluation end to end (rpeflow_amd.evaluate: loader threads -> pinned host batches -> H2D on a copy "
                                    "stream -> full RPEFlow forward replayed from its HIP graph -> metric sums on the device -> one float64[12] "
                                    "SUM all-reduce), a step = one batch per rank, host-to-device copies INSIDE the timed region; "
                                    + cfg["name"]) if args.workload == "eval" else
                                   ("RPEFlow hot path only (FPS, 43 KNN, correlation2d, warps, gathers, PointConv, Correlation3D) at "
                                    + cfg["name"] + "; dense 2D convs/attention excluded"),
                       "frame": [H, W], "points": NPTS, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "sharding": f"frame pairs over {world} rank(s), no data-path collective",
                       "launch": launch, "runtime": RUNTIME,
                       "process_group": ("%s, world size %d" % (args.backend, world)) if dist is not None else "none"},
            "roofline": pointconv_microbench(dev),
            "roofline_fps": {"kernel": "fps_pruned2_kernel (furthest_point_sampling, 2B clouds 8192 -> 4096)", "bound": "latency",
                             "achieved": round(fps_bytes / fps_us / 1e3, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(fps_bytes / fps_us / 1e3 / HBM_PEAK_GBS, 6), "algorithmic_bytes": fps_bytes,
                             "traffic": pmc_traffic("fps_pmc.json")[0], "traffic_source": pmc_traffic("fps_pmc.json")[1],
                             "us_per_launch": round(fps_us, 1), "launches_per_step": 1,
                             "us_per_iteration": round(fps_us / 4095, 4), "iteration_sync_floor_us": 0.41,
                             "frac_of_sync_floor": round(0.41 / (fps_us / 4095), 4),
                             "note": "the step's longest single kernel.  FPS is 4095 DEPENDENT sampling iterations per cloud (SURVEY.md 8d), one "
                                     "workgroup per cloud: bound by the per-iteration reduce-barrier-broadcast latency, not by HBM or MFMA -- the HBM "
                                     "fraction is reported for form; the figure of merit is us_per_iteration against iteration_sync_floor_us "
                                     "(DESIGN.md 4.3).  In the replayed forward it runs one batch ahead on its own stream, off the critical path"},
            "hotpath": {"frame_pairs_per_s": round(pairs / dt_hot, 3), "ms_per_step": round(dt_hot / args.steps * 1e3, 3),
                        "roofline_frac": round(sum_floor / (sum_ms * 1e-3), 5),
                        "roofline_frac_throughput_categories": round(sum_floor_tp / (sum_ms_tp * 1e-3), 5),
                        "hbm_frac": round(sum_b / (sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "algorithmic_bytes": sum_b, "algorithmic_flops": sum_f, "floor_ms": round(sum_floor * 1e3, 4),
                        "kernel_ms": round(sum_ms, 3), "categories": table,
                        "note": "the hot-path operator sequence of one forward alone (rpeflow_amd/hotpath.py), timed in this run with HIP events "
                                "per category.  Every category is priced at its roofline floor max(algorithmic bytes / 8 TB/s, algorithmic flops "
                                "/ 157.3 TFLOP/s fp32 matrix peak) (rpeflow_amd/roofline.py = SURVEY.md 8d; `bound` says which side sets it; "
                                "flops are the reference formulation's -- the binned nearest-pixel search evaluates far fewer pairs than the "
                                "exhaustive B*Q*M it is priced at); frac = floor / measured.  roofline_frac = sum of floors / sum of durations "
                                "over all categories; roofline_frac_throughput_categories leaves out the latency-bound furthest-point sampling "
                                "(4095 dependent iterations, roofline_fps); hbm_frac is the bytes-only figure earlier rounds printed"},
        }
        line["startup"] = {
            "time_to_first_step_s": max(s.get("first_step_s", 0.0) for s in startups) or None,
            "per_rank": startups,
            "note": "seconds from the start of each rank's process: imports done, model and batch on the device, first forward finished "
                    "(HIP module loads + MIOpen's solver search, cold user database on a fresh box), HIP graph captured"}
        if step_clock is not None:
            # what the device's own counters say about the K timed steps: a slow line with the usual cycles per step is a clock
            # (power / thermal / a neighbour on the node), one with more cycles is work or stalls
            torch.cuda.synchronize()
            cycles, ticks, khz = step_clock.raw()
            line["step_clock"] = {
                "shader_cycles_per_step": round(cycles / args.steps) if ticks > 0 else None,
                "ms_per_step_by_device_counter": round(ticks / khz / args.steps, 3) if (ticks > 0 and khz > 0) else None,
                "cycles_per_us": round(cycles / ticks * khz / 1e3, 1) if (ticks > 0 and khz > 0) else None,
                "cycles_per_us_per_xcd": step_clock.mhz_per_xcd(), "units_read": step_clock.units(),
                "note": "two rpe_clock_stamp_all launches around the timed steps (outside their barriers), median compute unit.  A compute "
                        "unit's cycle counter stands still while it is idle, and the forward does not keep every unit busy all the time: "
                        "cycles_per_us is busy-weighted, BELOW the engine clock, and comparable between runs of this same command only"}
        if epe_delta is not None:
            line["epe_delta"] = epe_delta
        if eval_info is not None:
            line["eval"] = eval_info
        if world == 1 and not args.no_corr_microbench:
            line["roofline_corr"] = corr_microbench(dev)
            line["roofline_knn"] = knn_microbench(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.config)
    if rank == 0:
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if epe_delta is not None and not epe_delta["within_bound"]:
        raise SystemExit(3)  # the line above carries the numbers; a configuration off the parity bound is not a valid result


if __name__ == "__main__":
    main()
