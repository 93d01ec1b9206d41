"""Input pipeline of the sharded evaluation: the counterpart of ``DataLoader(dataset, batch_size, num_workers=8,
pin_memory=True)`` + ``copy_to_device`` (eval_withocc.py:25-29, 56; conf/test/things.yaml:5; utils.py:34-43).

The forward of a batch of four 544x960 frame pairs takes ~18 ms on an MI355X; its inputs are 208 MB (4 x [3 MB
uint8 frames + 42 MB event voxel + 6 MB target flow + clouds]).  A loop that builds the batch inline and copies it
from pageable memory on the compute stream feeds 4 frame pairs a second.  Here the three stages run concurrently:

  stage 1  ``workers`` threads (or DataLoader worker processes, ``processes=True``: datasets that hold the GIL while
           decoding) load samples STRAIGHT into a ring of pinned host batches -- the collate is the load, there is no
           per-sample tensor and no torch.stack pass;
  stage 2  one copy thread moves filled host batches into a ring of device batches on a dedicated HIP stream (SDMA
           engines: the copies overlap the previous batch's replay), in batch order;
  stage 3  the consumer (evaluate()) iterates ``pairs()``: (batch j, batch j+1) with j+1 already resident, so that the
           forward of j can run the furthest-point sampling of j+1 (model.forward_ahead).  The compute stream waits on
           the copy stream's event, never the host; a device batch is handed back with an event recorded on the
           compute stream, and the copy stream waits on that before overwriting it.

Samples that already lie in pinned memory (``SyntheticPairs(pin=True)``, a registered memory-mapped set) skip stage 1:
they are copied to the device from where they are.  A dataset that implements ``load_into(i, out)`` is handed the sample's
place in the pinned host batch and reads / decodes straight into it: stage 1 then copies nothing (the staging memcpy of
``__getitem__`` samples is 1.2-1.4 cores a rank at 60 batches/s of 206 MB).  The pipeline's threads are
named ``rpe-load`` / ``rpe-copy`` (top -H) and report their CPU seconds in ``stats["thread_cpu_s"]``.

Raw events.  The reference's dataset voxelises a sample's events on the CPU when no pre-processed file exists
(flyingthings3d.py:206-208: load_events_h5 -> eventsToVoxel, ~50 ms a sample in numpy/torch against a 15 ms batch).  A
dataset may return ``events`` [n,4] (x, y, t, polarity; float32 as load_events_h5 yields them, or float64) INSTEAD of
``event_voxel`` and describe them in ``dataset.event_format`` = dict(bins, polarity, height, width, max_events): the events
cross PCIe (16 B each instead of 42 MB of grid per sample) and stage 2 voxelises them on the device, on the copy stream,
bit for bit what eventsToVoxel returns (event_ops.events_to_voxel).  The consumer sees ``event_voxel`` either way.

On a CPU device (the gloo tests) the same code runs without the device ring: stage 1 prefetches, stage 2 passes on.
Batch order and contents never depend on the number of workers (tests/test_loader.py).
"""
import os
import queue
import threading
import time

import numpy as np
import torch


class _Stop(Exception):
    pass


_COPY_STREAMS = {}  # device index -> the stream pick_copy_stream chose


def pick_copy_stream(device, busy=None, candidates=8, probe_mb=32):
    """A stream whose copies really run beside the consumer's kernels.

    HIP streams share a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default; PyTorch's pool of 32 streams maps onto
    them round robin), a hardware queue executes in order, and a replayed multi-stream HIP graph occupies several of them for
    its whole duration: a copy stream that lands on one of those queues gets its H2D copies executed BETWEEN two replays
    instead of under one (measured: 206 MB per batch then cost 3.7 ms per batch, 18.4 -> 22.1 ms, on whichever stream the
    pool handed out).  Which queue a stream gets cannot be asked, so it is measured: ``busy()`` enqueues the consumer's work
    (one replay of the forward) on the current stream, a ``probe_mb`` copy is issued on each candidate right behind it, and
    the candidate whose copy finishes first -- under the replay, not after it -- is kept for this device."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key in _COPY_STREAMS:
        return _COPY_STREAMS[key]
    if busy is None:
        return torch.cuda.Stream(device)
    src = torch.empty(probe_mb << 18, dtype=torch.float32, pin_memory=True)
    dst = torch.empty(probe_mb << 18, dtype=torch.float32, device=device)
    best, results = None, []
    for _ in range(candidates):
        stream = torch.cuda.Stream(device)
        torch.cuda.synchronize(device)
        begin, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        busy()
        with torch.cuda.stream(stream):
            dst.copy_(src, non_blocking=True)
            done.record(stream)
        torch.cuda.synchronize(device)
        ms = begin.elapsed_time(done)
        results.append(round(ms, 2))
        if best is None or ms < best[0]:
            best = (ms, stream)
    _COPY_STREAMS[key] = best[1]
    _COPY_STREAMS[(key, "probe_ms")] = results
    return best[1]


def _put(q, item, stop):
    while not stop.is_set():
        try:
            return q.put(item, timeout=0.1)
        except queue.Full:
            pass
    raise _Stop


def _get(q, stop):
    while not stop.is_set():
        try:
            return q.get(timeout=0.1)
        except queue.Empty:
            pass
    raise _Stop


def _memcpy(dst, src):
    """One thread, one memcpy, GIL released.  (Tensor.copy_ would fan a 42 MB copy out over the calling thread's whole
    OpenMP team -- 128 threads on the GPU box, whose container grants 16 cores: with four loader threads doing that at
    once the CFS quota throttled every thread of the process, the main one included, and the staging ring filled at 5-10
    GB/s instead of 40.)"""
    if src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous() and src.device.type == "cpu":
        np.copyto(dst.numpy(), src.numpy())
    else:
        dst.copy_(src)


class InputPipeline:
    def __init__(self, dataset, indices, batch_size, device, workers=4, depth=3, processes=False, busy=None):
        self.dataset, self.device = dataset, torch.device(device)
        self.event_format = getattr(dataset, "event_format", None)  # raw events instead of voxel grids (module docstring)
        self.batches = [list(indices[s:s + batch_size]) for s in range(0, len(indices), batch_size)]  # the last one may be short
        self.batch_size, self.workers, self.depth, self.processes = batch_size, max(1, int(workers)), max(3, int(depth)), processes
        self.cuda = self.device.type == "cuda"
        self.busy = busy  # enqueues the consumer's per-batch device work once: lets pick_copy_stream find a stream that overlaps it
        self.sync_ready = os.environ.get("RPE_PIPE_SYNC_READY", "dev")      # how the consumer waits for a batch's H2D copy
        # how the copy stream waits for the consumer to be done with a device batch: on the HOST (the copy thread waits for the
        # event, then issues the copy).  With a device-side wait (copy stream waiting on an event of the compute stream while the
        # compute stream waits on the copy stream's: "dev"), hipGraphLaunch on the compute stream blocks its caller for 20-30 ms
        # two launches out of three (ROCm 7.2, measured).  "host" polls the event (Event.query + a short sleep); "spin" is the
        # plain Event.synchronize of rounds 3-5, which busy-waits a core away (tools/host_rehearsal.py: 0.94 against 0.02 cores).
        self.sync_release = os.environ.get("RPE_PIPE_SYNC_RELEASE", "host")
        self.poll_s = float(os.environ.get("RPE_PIPE_POLL_MS", "0.25")) * 1e-3
        self.depth = max(self.depth, int(os.environ.get("RPE_PIPE_DEPTH", self.depth)))
        # diagnostic (tools/host_rehearsal.py: N ranks sharing ONE PCIe link): move only this fraction of every tensor to the device
        self.copy_fraction = float(os.environ.get("RPE_PIPE_COPY_FRACTION", "1"))
        self.stats = {"batches": 0, "bytes": 0, "direct": 0}
        self._account_lock = threading.Lock()
        self.trace = [] if os.environ.get("RPE_EVAL_TIMELINE") else None  # (batch, copy begin / end events, host time of issue)
        self._threads, self._stop, self._error = [], threading.Event(), None

    def __len__(self):
        return len(self.batches)

    # ------------------------------------------------------------------ stage 1: samples -> host batches
    def _ring(self, like, count, **kw):
        q = queue.Queue()
        for _ in range(count):
            q.put(({k: torch.empty((self.batch_size,) + tuple(v.shape), dtype=v.dtype, **kw) for k, v in like.items()}, None))
        return q

    def _sample(self, i):
        first, self._first = self._first, None  # the sample _start() looked at is not loaded twice
        return first[1] if first is not None and first[0] == i else self.dataset[i]

    def _check_events(self, ev):
        """What eventsToVoxel's index_put_ would refuse, found on the host (the device stage then needs no round trip)."""
        f = self.event_format
        if ev.dim() != 2 or ev.shape[1] != 4 or ev.shape[0] > f["max_events"]:
            raise ValueError("events: expected [n <= %d, 4], got %s" % (f["max_events"], tuple(ev.shape)))
        if ev.shape[0]:
            a = ev.numpy()
            x, y = a[:, 0].astype(np.int32), a[:, 1].astype(np.int32)
            if int(x.min()) < 0 or int(x.max()) >= f["width"] or int(y.min()) < 0 or int(y.max()) >= f["height"]:
                raise IndexError("event coordinates outside the sensor")

    def _voxelise(self, dev, s, count, t_range):
        """Device batch slot s: dev["events"][s][:count] -> dev["event_voxel"][s], on the current (copy) stream."""
        from .event_ops import events_to_voxel
        f = self.event_format
        events_to_voxel(dev["events"][s][:count], num_bins=f["bins"], height=f["height"], width=f["width"], event_polarity=f["polarity"],
                        out=dev["event_voxel"][s], t_range=t_range, validate=False)

    def _worker(self):
        """One task = one SAMPLE of a batch (the first batch is resident after one sample's load time, not four).  Tasks and
        host slots are handed out in batch order under one lock: the copier's next batch can never starve for a slot."""
        from .runtime import name_thread
        name_thread("rpe-load")
        t_cpu = time.thread_time()
        try:
            while True:
                with self._next_lock:
                    if self._next >= len(self._tasks):
                        self._account("rpe-load", t_cpu)
                        return
                    j, n = self._tasks[self._next]
                    self._next += 1
                    if n == 0:
                        slot, busy = (None, None) if self._direct else _get(self._free_host, self._stop)
                        if busy is not None:
                            busy.synchronize()  # the H2D that last read this slot
                        self._open[j] = {"slot": slot, "left": len(self.batches[j]), "samples": [None] * len(self.batches[j])}
                    rec = self._open[j]
                index = self.batches[j][n]
                if rec["slot"] is not None and self._load_into is not None and not (self._first is not None and self._first[0] == index):
                    # the dataset decodes / reads STRAIGHT into the pinned host batch: no sample tensor, no staging copy
                    self._load_into(index, {k: v[n] for k, v in rec["slot"].items()})
                else:
                    sample = self._sample(index)
                    if "events" in sample:
                        self._check_events(sample["events"])
                    if rec["slot"] is None:  # pinned samples: copied from where they lie
                        rec["samples"][n] = sample
                    else:
                        for k, v in sample.items():
                            if k == "events":  # ragged: the first count rows of the slot's [max_events, 4]
                                _memcpy(rec["slot"][k][n][:v.shape[0]], v)
                                rec["slot"]["event_count"][n] = v.shape[0]
                            else:
                                _memcpy(rec["slot"][k][n], v)
                with self._filled_cv:
                    rec["left"] -= 1
                    if rec["left"] == 0:
                        del self._open[j]
                        self._filled[j] = ("direct", rec["samples"], len(rec["samples"])) if rec["slot"] is None else ("slot", rec["slot"], len(rec["samples"]))
                        self._filled_cv.notify_all()
        except _Stop:
            pass
        except BaseException as e:  # noqa: BLE001 -- surfaces in the consumer
            self._fail(e)

    def _process_source(self):
        """stage 1 with DataLoader worker processes: batches arrive collated in shared memory, one thread stages them."""
        from torch.utils.data import DataLoader, Subset
        try:
            flat = [i for b in self.batches for i in b]
            dl = DataLoader(Subset(self.dataset, flat), batch_size=self.batch_size, shuffle=False, num_workers=self.workers,
                            collate_fn=lambda s: {k: torch.stack([x[k] for x in s]) for k in s[0]}, prefetch_factor=2)
            for j, cpu in enumerate(dl):
                n = len(self.batches[j])
                if self.cuda:
                    slot, busy = _get(self._free_host, self._stop)
                    if busy is not None:
                        busy.synchronize()
                    for k, v in cpu.items():
                        _memcpy(slot[k][:n], v)
                else:
                    slot = cpu
                with self._filled_cv:
                    self._filled[j] = ("slot", slot, n)
                    self._filled_cv.notify_all()
        except _Stop:
            pass
        except BaseException as e:  # noqa: BLE001
            self._fail(e)

    # ------------------------------------------------------------------ stage 2: host batches -> device batches, in order
    def _copier(self):
        from .runtime import name_thread
        name_thread("rpe-copy")
        t_cpu = time.thread_time()
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            for j in range(len(self.batches) + 1):
                if j == len(self.batches):
                    self._account("rpe-copy", t_cpu)
                    break
                with self._filled_cv:
                    while j not in self._filled:
                        if self._stop.is_set():
                            raise _Stop
                        self._filled_cv.wait(0.1)
                    kind, src, n = self._filled.pop(j)
                if not self.cuda:  # CPU consumer: the host batch itself is what it gets; handed back on release
                    back = None if self.processes else (lambda src=src: self._free_host.put((src, None)))
                    _put(self._ready, ({k: v[:n] for k, v in src.items()}, None, back), self._stop)
                    continue
                dev, released = _get(self._free_dev, self._stop)
                if released is not None and self.sync_release == "host":
                    # the consumer's last kernel that read this device batch (host-side wait: see __init__).  Polled, not
                    # hipEventSynchronize: the host runs several replays ahead of the device, so this thread waits here most of
                    # every batch, and the runtime's wait SPINS -- a whole core per rank (0.94 measured, blocking-sync events
                    # included).  A batch is 15 ms; a 0.25 ms poll is nothing against the two batches of slack the ring holds.
                    while not released.query():
                        if self._stop.is_set():
                            raise _Stop
                        time.sleep(self.poll_s)
                elif released is not None and self.sync_release == "spin":
                    released.synchronize()
                with torch.cuda.stream(self._copy_stream):
                    if released is not None and self.sync_release != "host":
                        self._copy_stream.wait_event(released)
                    if self.trace is not None:  # diagnostic: when the copy of batch j really ran
                        begin = torch.cuda.Event(enable_timing=True)
                        begin.record(self._copy_stream)
                    if os.environ.get("RPE_PIPE_NOCOPY"):  # diagnostic: everything but the copies themselves
                        pass
                    elif kind == "slot" and self.copy_fraction < 1:
                        for k, v in src.items():
                            m = max(1, int(v[:n].numel() * self.copy_fraction))
                            dev[k].view(-1)[:m].copy_(v.view(-1)[:m], non_blocking=True)
                            self.stats["bytes"] += m * v.element_size()
                    elif kind == "slot":
                        for k, v in src.items():
                            if k == "events":
                                continue
                            dev[k][:n].copy_(v[:n], non_blocking=True)
                            self.stats["bytes"] += v[:n].numel() * v.element_size()
                        if "events" in src:
                            for s in range(n):
                                count = int(src["event_count"][s])
                                ev = src["events"][s][:count]
                                dev["events"][s][:count].copy_(ev, non_blocking=True)
                                self.stats["bytes"] += ev.numel() * ev.element_size()
                                self._voxelise(dev, s, count, (float(ev[0, 2]), float(ev[-1, 2])) if count else None)
                    else:
                        for s, sample in enumerate(src):
                            for k, v in sample.items():
                                if k == "events":
                                    dev[k][s][:v.shape[0]].copy_(v, non_blocking=True)
                                    self.stats["bytes"] += v.numel() * v.element_size()
                                    self._voxelise(dev, s, v.shape[0], (float(v[0, 2]), float(v[-1, 2])) if v.shape[0] else None)
                                    continue
                                if self.copy_fraction < 1:
                                    m = max(1, int(v.numel() * self.copy_fraction))
                                    dev[k][s].view(-1)[:m].copy_(v.view(-1)[:m], non_blocking=True)
                                    self.stats["bytes"] += m * v.element_size()
                                    continue
                                dev[k][s].copy_(v, non_blocking=True)
                                self.stats["bytes"] += v.numel() * v.element_size()
                        self.stats["direct"] += 1
                    done = torch.cuda.Event(enable_timing=self.trace is not None)
                    done.record(self._copy_stream)
                    if self.trace is not None:
                        self.trace.append((j, begin, done, time.perf_counter()))
                if kind == "slot":
                    self._free_host.put((src, done))
                _put(self._ready, ({k: v[:n] for k, v in dev.items() if k not in ("events", "event_count")}, done, dev), self._stop)  # dev: handed back with an event
        except _Stop:
            pass
        except BaseException as e:  # noqa: BLE001
            self._fail(e)

    def _account(self, name, since):
        """CPU seconds a pipeline thread used (it ends with the pipeline: nobody can read /proc for it afterwards)."""
        with self._account_lock:
            used = self.stats.setdefault("thread_cpu_s", {})
            used[name] = used.get(name, 0.0) + time.thread_time() - since

    def _fail(self, e):
        self._error = e
        self._stop.set()

    # ------------------------------------------------------------------ stage 3: the consumer
    def _start(self):
        self._stop.clear()
        self._error = None
        self._next, self._next_lock = 0, threading.Lock()
        self._tasks, self._open = [(j, n) for j, ids in enumerate(self.batches) for n in range(len(ids))], {}
        self._filled, self._filled_cv = {}, threading.Condition()
        first = self.dataset[self.batches[0][0]]  # defines keys, shapes and dtypes
        self._first = (self.batches[0][0], first)
        self._direct = self.cuda and not self.processes and all(v.is_pinned() for v in first.values())
        # ``dataset.load_into(i, out)`` (optional): fill ``out`` -- {key: the sample's place in a host batch, pinned on a GPU run}
        # -- with sample i.  A dataset that reads or decodes (np.load, h5py read_direct, an image decoder with an output
        # buffer) does so straight into the batch the H2D copy starts from; __getitem__ + memcpy costs a rank a core at 60
        # batches/s of 206 MB (tools/host_rehearsal.py --real-rank 0: 1.2 cores in the loader threads)
        self._load_into = None if (self.processes or self._direct or "events" in first) else getattr(self.dataset, "load_into", None)
        no_host_ring = self._direct or (self.processes and not self.cuda)
        host_like = dev_like = first
        if "events" in first:
            f = self.event_format
            if f is None or "event_voxel" in first:
                raise ValueError("a dataset that returns raw `events` describes them in `event_format` and returns no `event_voxel`")
            if not self.cuda or self.processes:
                raise RuntimeError("raw events are voxelised on the GPU by the pipeline's copy stage: a CUDA device, loader threads")
            host_like = dict(first, events=torch.empty((f["max_events"], 4), dtype=first["events"].dtype), event_count=torch.zeros((), dtype=torch.int64))
            dev_like = dict(host_like, event_voxel=torch.empty(((2 if f["polarity"] else 1) * f["bins"], f["height"], f["width"]), dtype=torch.float32))
        self._free_host = queue.Queue() if no_host_ring else self._ring(host_like, self.depth + self.workers, pin_memory=self.cuda)
        self._free_dev = self._ring(dev_like, self.depth, device=self.device) if self.cuda else queue.Queue()
        self._ready = queue.Queue(maxsize=self.depth)
        self._copy_stream = pick_copy_stream(self.device, self.busy) if self.cuda else None
        targets = [self._process_source] if self.processes else [self._worker] * self.workers
        self._threads = [threading.Thread(target=t, daemon=True) for t in targets + [self._copier]]
        for t in self._threads:
            t.start()

    def _take(self):
        """Next device batch; the compute stream is made to wait for its copy."""
        try:
            batch, done, dev = _get(self._ready, self._stop)
        except _Stop:
            raise RuntimeError("input pipeline failed") from self._error
        if done is not None:
            if self.sync_ready == "host":
                done.synchronize()  # (finished long ago in the steady state: the copy ran under the previous replay)
            else:
                torch.cuda.current_stream(self.device).wait_event(done)
        self.stats["batches"] += 1
        return batch, dev

    def _release(self, dev):
        if callable(dev):
            dev()
        elif dev is not None:
            # (a blocking-sync event changes nothing here: 63.3 batches/s and 3.3-3.5 cores either way -- the host-side wait of the
            # copy thread does not spin; tools/host_rehearsal.py --real-rank, profiles/r05_host_rehearsal_real_*.json)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))  # everything the consumer launched on this batch so far
            self._free_dev.put((dev, ev))

    def close(self):
        self._stop.set()
        for t in self._threads:
            t.join(timeout=10)
        self._threads = []

    def pairs(self):
        """Yields (batch j, batch j+1 or None) in shard order.  Batch j is handed back to the ring when the consumer asks for
        the next pair: consume it (launch everything that reads it) before that."""
        if not self.batches:
            return
        self._start()
        try:
            cur = self._take()
            for j in range(len(self.batches)):
                nxt = self._take() if j + 1 < len(self.batches) else None
                yield cur[0], (nxt[0] if nxt is not None else None)
                self._release(cur[1])
                cur = nxt
        finally:
            self.close()

    def __iter__(self):
        for batch, _ in self.pairs():
            yield batch
