"""Sharded evaluation harness -- the counterpart of eval_withocc.py:43-135 / eval_noocc.py:46-116.

The reference evaluates on one GPU and pulls ~12 scalars per sample to the host with ``.item()``.
Here frame pairs are independent units (BatchNorm in eval mode, no cross-sample state), so:

* one process per GPU; rank r evaluates samples r, r+W, r+2W, ... -- no padding or duplication
  (``DistributedSampler``'s pad-to-even would bias the count-weighted means);
* the accumulators live in ONE float64[12] tensor on the device, updated without host syncs:
  {count, EPE, 1px, Fl} 2-D, {count, EPE, 5cm, 10cm} 3-D, the same for non-occluded points;
* after the loop a single SUM all-reduce of that tensor (RCCL over xGMI through
  torch.distributed's "nccl" backend; "gloo" on CPU in the tests) -- the template is
  ``dist_reduce_sum`` (utils.py:26-31), which the reference calls once per scalar.

Counts stay below 2^53, so float64 sums of counts are exact; the value sums differ from a
single-process run only by float64 re-association.
"""
import argparse
import gc
import json
import os
import time

import torch

FIELDS = ["count_2d", "epe_2d", "acc_1px", "fl", "count_3d", "epe_3d", "acc_5cm", "acc_10cm",
          "count_3d_noc", "epe_3d_noc", "acc_5cm_noc", "acc_10cm_noc"]


def new_accumulator(device):
    return torch.zeros(len(FIELDS), dtype=torch.float64, device=device)


_WORKSPACE = {}


def _accumulate_device(acc, outputs, inputs):
    """The same sums in one pass + a fixed-order reduction on the device (csrc/eval.hip, rpe_eval_accumulate): two
    launches instead of ~45 tensor ops over the full-resolution maps between two replays of the forward."""
    from . import _lib
    f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
    f2p, f3p, f2t, f3t = f32(outputs["flow_2d"]), f32(outputs["flow_3d"]), f32(inputs["flow_2d"]), f32(inputs["flow_3d"])
    occ = f32(inputs["occ_mask_3d"]) if "occ_mask_3d" in inputs else None
    B, HW, N = f2p.shape[0], f2p.shape[2] * f2p.shape[3], f3p.shape[2]
    assert f2p.shape[1] == 2 and f3p.shape[1] == 3 and f2t.shape[0] == B and f3t.shape[0] == B and f3t.shape[2] == N
    lib = _lib.lib()
    need = lib.rpe_eval_workspace_doubles(B * HW, B * N)
    key = (acc.device, torch.cuda.current_stream(acc.device).cuda_stream)
    work = _WORKSPACE.get(key)
    if work is None or work.numel() < need:
        work = _WORKSPACE[key] = torch.empty(need, dtype=torch.float64, device=acc.device)
    with torch.cuda.device(acc.device):
        rc = lib.rpe_eval_accumulate(f2p.data_ptr(), f2t.data_ptr(), f2t.shape[1], B, HW, f3p.data_ptr(), f3t.data_ptr(), f3t.shape[1], N,
                                     occ.data_ptr() if occ is not None else None, work.data_ptr(), acc.data_ptr(), _lib.stream_of(acc))
    _lib.check(rc, "eval_accumulate")
    return acc


@torch.no_grad()
def accumulate(acc, outputs, inputs):
    """Adds one batch to ``acc``.  Per-sample arithmetic of eval_withocc.py:65-108: EPE maps
    sqrt(sum diff^2) in fp32; masks from the extra target channel (if any) and not-NaN; Fl =
    epe > 3 and epe/|gt| > 0.05; the non-occluded group only where inputs carry occ_mask_3d.
    Device accumulators go through the HIP kernel (no tensor-op fallback on the GPU); the tensor-op form below is the
    host-side statement of the same sums for CPU tensors (the gloo tests, the accumulator goldens)."""
    if acc.is_cuda:
        return _accumulate_device(acc, outputs, inputs)
    f2p, f3p = outputs["flow_2d"].float(), outputs["flow_3d"].float()
    f2t, f3t = inputs["flow_2d"].float(), inputs["flow_3d"].float()
    m2 = f2t[:, 2] > 0 if f2t.shape[1] > 2 else torch.ones_like(f2t[:, 0], dtype=torch.bool)
    m3 = f3t[:, 3] > 0 if f3t.shape[1] > 3 else torch.ones_like(f3t[:, 0], dtype=torch.bool)
    f2t, f3t = f2t[:, :2], f3t[:, :3]
    epe2 = torch.sqrt(torch.sum((f2p - f2t) ** 2, dim=1))
    epe3 = torch.sqrt(torch.sum((f3p - f3t) ** 2, dim=1))
    m2 = m2 & ~torch.isnan(epe2)
    m3 = m3 & ~torch.isnan(epe3)
    fl = (epe2 > 3.0) & (epe2 / torch.linalg.norm(f2t, dim=1) > 0.05)

    def group(epe, mask, *thresholds_or_maps):
        d = lambda t: t.to(torch.float64).sum()
        out = [d(mask), d(torch.where(mask, epe, torch.zeros_like(epe)))]
        for t in thresholds_or_maps:
            hit = t if torch.is_tensor(t) else (epe < t)
            out.append(d(hit & mask))
        return out

    vals = group(epe2, m2, 1.0, fl) + group(epe3, m3, 0.05, 0.1)
    if "occ_mask_3d" in inputs:
        vals += group(epe3, m3 & (inputs["occ_mask_3d"] == 0), 0.05, 0.1)
    else:
        vals += [torch.zeros((), dtype=torch.float64, device=acc.device)] * 4
    acc += torch.stack(vals).to(acc.device)
    return acc


def finalize(acc):
    """eval_withocc.py:119-135: sums / counts (percentages for the accuracy entries)."""
    a = [float(x) for x in acc.tolist()]
    div = lambda s, c: s / c if c > 0 else float("nan")
    m = {"EPE2D": div(a[1], a[0]), "1px": 100 * div(a[2], a[0]), "Fl": 100 * div(a[3], a[0]),
         "EPE3D": div(a[5], a[4]), "5cm": 100 * div(a[6], a[4]), "10cm": 100 * div(a[7], a[4])}
    if a[8] > 0:
        m.update({"EPE3D_noc": div(a[9], a[8]), "5cm_noc": 100 * div(a[10], a[8]), "10cm_noc": 100 * div(a[11], a[8])})
    m["counts"] = {"2d": a[0], "3d": a[4], "3d_noc": a[8]}
    return m


def shard_indices(n_samples, rank, world_size):
    """Rank r takes r, r+W, ...; the union over ranks is exactly range(n_samples)."""
    return list(range(rank, n_samples, world_size))


def collate(samples):
    return {k: torch.stack([s[k] for s in samples]) for k in samples[0]}


def to_device(batch, device):
    """copy_to_device (utils.py:34-43) for the flat dict the datasets return."""
    return {k: v.to(device, non_blocking=True) for k, v in batch.items()}


class GraphedForward:
    """model(batch) replayed from one HIP graph per input signature: the first batch of a shape runs eagerly (MIOpen
    search, caches) and is then captured with static input buffers; later batches of that shape are copied into the
    buffers and replayed -- ~1400 kernel launches and the Python between them become one call.  The returned tensors
    are the graph's static outputs: consume them before the next call.

    ``ahead=True`` (default) captures RPEFlow.forward_ahead instead: a call that is also given the NEXT batch runs that
    batch's furthest-point sampling inside this replay, on its own stream, and the next call starts from the finished
    order.  A call whose batch is not the one announced by the previous call (first batch, shape change, skipped
    announcement) samples before replaying, so results never depend on the announcements being right."""
    INPUTS = ("images", "pcs", "intrinsics", "event_voxel")  # what RPEFlow.forward reads (models/RPEFlow.py:36-47)
    SAMPLING_INPUTS = ("pcs", "intrinsics")                  # what the sampling order depends on (plus the frame shape)

    def __init__(self, model, warmup=2, ahead=True):
        self.model, self.warmup, self.entries = model, warmup, {}
        self.host_times = [] if os.environ.get("RPE_EVAL_TIMELINE") else None  # diagnostic: host seconds inside graph.replay()
        self.host_copy_times = []
        self.ahead = ahead and hasattr(model, "forward_ahead")

    def _key(self, batch):
        return tuple((k, tuple(batch[k].shape), batch[k].dtype) for k in self.INPUTS)

    def _capture(self, batch):
        static = {k: batch[k].clone() for k in self.INPUTS}
        for _ in range(self.warmup):
            self.model(static)
        entry = {"static": static, "announced": None}
        if self.ahead:
            # the next batch's sampling inputs; ``images`` rides along for its shape only
            entry["next"] = {"images": static["images"], **{k: static[k].clone() for k in self.SAMPLING_INPUTS}}
            entry["order"] = self.model.sample_order(static)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):  # other threads (RCCL watchdog) may query events meanwhile
            if self.ahead:
                entry["out"] = self.model.forward_ahead(static, entry["order"], entry["next"])
            else:
                entry["out"] = self.model(static)
        entry["graph"] = graph
        return entry

    def replay_any(self):
        """One replay of a captured forward on whatever its input buffers hold (the input pipeline times its copy-stream
        candidates against it; the outputs are overwritten by the next real batch)."""
        next(iter(self.entries.values()))["graph"].replay()

    def _padded_entry(self, batch):
        """A captured graph of the same sample shapes and a LARGER batch: the short last batch of a shard rides in its first
        rows (samples are independent in eval mode; the other rows keep the previous batch), instead of costing a capture
        and a MIOpen search for a batch size seen once."""
        n = batch["pcs"].shape[0]
        for key, entry in self.entries.items():
            if all(tuple(batch[k].shape[1:]) == shape[1:] and batch[k].dtype == dtype and shape[0] > n for k, shape, dtype in key):
                return entry
        return None

    @torch.no_grad()
    def __call__(self, batch, next_batch=None):
        key = self._key(batch)
        entry = self.entries.get(key)
        n = None
        if entry is None:
            entry = self._padded_entry(batch)
            if entry is not None:
                n = batch["pcs"].shape[0]
            else:
                entry = self.entries[key] = self._capture(batch)
        static = entry["static"]
        t_a = time.perf_counter()
        for k in self.INPUTS:
            (static[k] if n is None else static[k][:n]).copy_(batch[k], non_blocking=True)
        if self.host_times is not None:
            self.host_copy_times.append(time.perf_counter() - t_a)
        if n is not None:  # rows n.. hold the previous batch: sample the mixture in place, announce nothing
            if self.ahead:
                entry["order"].copy_(self.model.sample_order(static))
                entry["announced"] = None
            entry["graph"].replay()
            return {k: v[:n] for k, v in entry["out"].items()}
        if self.ahead:
            if entry["announced"] is not batch["pcs"]:  # nobody sampled this batch ahead of time
                entry["order"].copy_(self.model.sample_order(static))
            entry["announced"] = None
            if next_batch is not None and self._key(next_batch) == key:
                for k in self.SAMPLING_INPUTS:
                    entry["next"][k].copy_(next_batch[k], non_blocking=True)
                entry["announced"] = next_batch["pcs"]
        if self.host_times is not None:
            t0 = time.perf_counter()
            entry["graph"].replay()
            self.host_times.append(time.perf_counter() - t0)
        else:
            entry["graph"].replay()
        return entry["out"]


def default_workers():
    """Loader threads per rank: the cores this process may use (affinity, cgroup quota), shared by the ranks of the node; at
    most 4, at least ``RPE_MIN_LOADER_THREADS`` (default 1).  A batch is 0.2 GB of memcpy: one thread stages ~70 batches/s
    when eight ranks share a 16-core quota (tools/host_rehearsal.py) -- the main and the copy thread of a rank are mostly
    asleep, so the loaders may use the rank's whole share."""
    from .runtime import cores_per_rank
    return max(int(os.environ.get("RPE_MIN_LOADER_THREADS", "1")), min(4, cores_per_rank()))


@torch.no_grad()
def evaluate(model, dataset, batch_size, device, rank=0, world_size=1, group=None, graph=None, workers=None,
             processes=False, forward=None, stats=None):
    """Evaluates this rank's shard and returns the GLOBAL metrics (identical on every rank).
    ``graph``: replay the forward from HIP graphs (default: on for GPU models that can be captured, i.e. IDS on device);
    ``forward``: a GraphedForward to reuse (its captures outlive this call).  Input pipeline (rpeflow_amd/loader.py, the
    counterpart of the reference's DataLoader(num_workers=8), eval_withocc.py:25-29): ``workers`` loader threads -- or
    DataLoader worker processes with ``processes=True`` -- fill pinned host batches, a copy stream moves them to the device
    one batch ahead of the forward.  ``stats``: a dict that receives the pipeline's counters."""
    from .loader import InputPipeline
    acc = new_accumulator(device)
    mine = shard_indices(len(dataset), rank, world_size)
    if graph is None:  # capture costs a few seconds once: worth it from a few dozen batches on
        graph = forward is not None or (torch.device(device).type == "cuda" and not getattr(model, "ids_on_host", False)
                                        and len(mine) >= 32 * batch_size)
    if graph and forward is None:
        forward = GraphedForward(model)
    busy = forward.replay_any if (forward is not None and forward.entries) else None
    pipe = InputPipeline(dataset, mine, batch_size, device, workers=default_workers() if workers is None else workers, processes=processes,
                         busy=busy)
    marks = [] if (stats is not None and stats.get("timeline") and torch.device(device).type == "cuda") else None
    host = []
    mark = (lambda: marks.append(torch.cuda.Event(enable_timing=True)) or marks[-1].record() or host.append(time.perf_counter())) if marks is not None else (lambda: None)
    for batch, upcoming in pipe.pairs():  # ``upcoming`` is resident already: its sampling runs inside this batch's replay
        mark()
        out = forward(batch, upcoming) if graph else model(batch)
        mark()
        accumulate(acc, out, batch)
        mark()
    import torch.distributed as dist
    if world_size > 1 or group is not None:
        # decided by the caller's arguments alone: an unsharded call (world_size=1, group=None) made inside a multi-rank job --
        # rank 0 validating on its own, or every rank evaluating the whole set -- must neither wait for the others nor multiply
        # its sums.  A launcher's single rank passes its world-size-1 group explicitly (bench.py, main() below): the sum is the
        # identity there, and the collective has then run on the real backend in every configuration
        if acc.is_cuda and dist.get_backend(group) == "gloo":  # (tests on a one-GPU box: the collective through host memory)
            acc_host = acc.cpu()
            dist.all_reduce(acc_host, op=dist.ReduceOp.SUM, group=group)
            acc.copy_(acc_host)
        else:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)  # the one collective of the evaluation
    if stats is not None:
        stats.update(pipe.stats, shard=len(mine), workers=pipe.workers)
        from . import loader
        probe = loader._COPY_STREAMS.get((torch.device(device).index or 0, "probe_ms"))
        if probe is not None:
            stats["copy_stream_probe_ms"] = probe
        if marks:  # device time per batch: the forward (input copies into the graph's buffers + replay), the metric sums, and
            torch.cuda.synchronize()  # the gap before the next batch's first launch (host-side stalls, waits for an H2D copy)
            ms = lambda a, b: a.elapsed_time(b)
            n = len(marks) // 3
            stats["timeline_ms"] = {"forward": round(sum(ms(marks[3 * i], marks[3 * i + 1]) for i in range(n)) / n, 3),
                                    "accumulate": round(sum(ms(marks[3 * i + 1], marks[3 * i + 2]) for i in range(n)) / n, 3),
                                    "gap": round(sum(ms(marks[3 * i + 2], marks[3 * i + 3]) for i in range(n - 1)) / max(1, n - 1), 3)}
            if forward is not None and forward.host_times:
                ht = forward.host_times[-n:]
                stats["timeline_ms"]["host_in_replay_call"] = round(sum(ht) / len(ht) * 1e3, 3)
                stats["timeline_ms"]["host_in_replay_call_samples"] = [round(x * 1e3, 2) for x in ht[8:20]]
                stats["timeline_ms"]["host_in_static_copies"] = [round(x * 1e3, 2) for x in forward.host_copy_times[-n:][8:20]]
                stats["timeline_ms"]["host_loop"] = round((host[-1] - host[0]) / max(1, n - 1) * 1e3 * (n - 1) / n, 3)
            if pipe.trace and len(marks) > 24:  # per batch, ms relative to the start of batch 8's forward: forward begin/end on the device, its H2D copy's
                base, hbase = marks[24], host[24]  # begin/end on the device, and when the host issued the replay and the copy
                rows = []
                for j, begin, done, t_issue in pipe.trace:
                    if 8 <= j < 16:
                        rows.append({"batch": j, "fwd": [round(ms(base, marks[3 * j]), 2), round(ms(base, marks[3 * j + 1]), 2)],
                                     "h2d": [round(ms(base, begin), 2), round(ms(base, done), 2)],
                                     "host_fwd_issue": round((host[3 * j] - hbase) * 1e3, 2), "host_h2d_issue": round((t_issue - hbase) * 1e3, 2)})
                stats["trace"] = rows
    return finalize(acc), acc


def first_forward_in_turn(step, rank, group=None):
    """Runs ``step()`` -- a process's FIRST forward -- on rank 0 alone, then on the other ranks of ``group`` together.
    The first forward runs MIOpen's solver search for every convolution shape and writes what it timed to the user's find
    database, which the ranks of a node share: in turn, the node searches once instead of once per rank, every rank ends up on
    the same solvers, and no search is timed while the other ranks load the machine (bench.py: 22.5 s to the first step with
    eight ranks searching at once; eight ranks timing their searches on one shared GPU left a database 10 % slower)."""
    import torch.distributed as dist
    together = group is not None and dist.get_world_size(group) > 1
    if together and rank != 0:
        dist.barrier(group)
    try:
        out = step()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    finally:
        if together and rank == 0:  # (also when the step raised: the others must not wait for ever)
            dist.barrier(group)
    return out


def main():
    p = argparse.ArgumentParser(description="Sharded synthetic evaluation (one process per GPU; launch with torchrun for N > 1)")
    p.add_argument("--samples", type=int, default=8)
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--height", type=int, default=544)
    p.add_argument("--width", type=int, default=960)
    p.add_argument("--points", type=int, default=8192)
    p.add_argument("--dsec", action="store_true")
    p.add_argument("--raw-events", type=int, default=0, help="samples carry up to this many raw events each instead of voxel grids (voxelised on the device)")
    p.add_argument("--weights", default=None, help="reference checkpoint ({'state_dict': ...}); default: seeded random init")
    p.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying HIP graphs")
    args = p.parse_args()
    from . import runtime
    runtime.configure()  # before the first GPU call: graph-queue count; MIOpen solvers stay at the library defaults
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    # a launcher sets all three (torchrun does, also for one rank); a stale MASTER_PORT in a plain shell is not a launcher
    launched = world > 1 or all(k in os.environ for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    group = None
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        group = dist.group.WORLD
    from .model import RPEFlow
    from .synthetic import load_seeded_parameters
    model = load_seeded_parameters(RPEFlow()).to(device).eval()
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location=device)["state_dict"], strict=True)
    from .synthetic import SyntheticPairs
    data = SyntheticPairs(args.samples, args.height, args.width, args.points, dsec=args.dsec, events=args.raw_events)
    # The model's ~200 k long-lived Python objects leave the collector's sight: a full collection of them is a 40-70 ms host
    # pause (measured in bench.py), and the loop below runs one or two replays ahead of the device at most.
    mine = shard_indices(len(data), rank, world)
    if not args.raw_events:  # the first forward (solver search) rank by rank, on this rank's first batch (an empty shard: just the barrier)
        first = to_device(collate([data[i] for i in mine[:args.batch]]), device) if mine else None
        first_forward_in_turn((lambda: model(first)) if mine else (lambda: None), rank, group)
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    metrics, _ = evaluate(model, data, args.batch, device, rank, world, group=group, graph=False if args.eager else None)
    torch.cuda.synchronize()
    metrics["seconds"] = round(time.perf_counter() - t0, 3)
    if rank == 0:
        print(json.dumps(metrics))
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
