"""Synthetic evaluation samples: no dataset ships with this repository, so the
harness and bench generate frame pairs of the FlyingThings3D / DSEC shapes from a seeded numpy stream."""
import numpy as np
import torch


def raw_events(seed, n, H, W):
    """[n', 4] float32 (x, y, t, polarity) in time order, n' a little below n and different per seed -- the array
    load_events_h5 builds from an event file (event_utils.py:11-20): microsecond timestamps, polarity 0 / 1."""
    r = np.random.default_rng(seed + 0x5EED)
    n = n - int(seed * 7919 % (n // 8 + 1))
    t = np.sort(r.integers(1_000_000, 1_050_000, n))
    return np.stack([r.integers(0, W, n), r.integers(0, H, n), t, r.integers(0, 2, n)], axis=1).astype(np.float32)


def frame_pair(seed, H=544, W=960, N=8192, f=1050.0, dsec=False, events=0):
    """One sample with the keys the reference datasets return (flyingthings3d.py:228-234): uint8 RGB
    pair, 20-channel event voxel, two back-projected clouds (pc2 = pc1 + N(0,0.05^2)), targets.
    ``dsec``: flow_3d carries a 4th mask channel and there is no occ_mask_3d (dsec.py:762,777-784).
    ``events`` > 0: up to that many raw events (``events`` [n,4], see raw_events) instead of the finished voxel grid --
    the dataset's path without a pre-processed file (flyingthings3d.py:206-208); the input pipeline voxelises them."""
    r = np.random.default_rng(seed)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    images = r.integers(0, 256, (6, H, W), dtype=np.uint8)
    event_voxel = r.standard_normal((20, H, W), dtype=np.float32) if not events else None
    z = r.uniform(2.0, 35.0, N)
    u = r.uniform(0.0, W - 1.0, N)
    v = r.uniform(0.0, H - 1.0, N)
    pc1 = np.stack([(u - cx) * z / f, (v - cy) * z / f, z]).astype(np.float32)
    pc2 = (pc1 + r.standard_normal((3, N)) * 0.05).astype(np.float32)
    flow_2d = np.concatenate([r.standard_normal((2, H, W)) * 5.0, np.ones((1, H, W))]).astype(np.float32)
    flow_3d = (pc2 - pc1).astype(np.float32)
    occ = (r.random(N) < 0.2).astype(np.float32)
    sample = {"images": images, "event_voxel": event_voxel, "pcs": np.concatenate([pc1, pc2]).astype(np.float32),
              "flow_2d": flow_2d, "flow_3d": flow_3d, "intrinsics": np.array([f, cx, cy], np.float32)}
    if dsec:
        sample["flow_3d"] = np.concatenate([flow_3d, (r.random((1, N)) < 0.9).astype(np.float32)])
    else:
        sample["occ_mask_3d"] = occ
    if events:
        del sample["event_voxel"]
        sample["events"] = raw_events(seed, events, H, W)
    return sample


class SyntheticPairs(torch.utils.data.Dataset):
    """Sample i is frame_pair(1000 + i % distinct) (``distinct`` defaults to the length: all samples different), so every
    rank can regenerate any sample.  The generator draws 12 M normal variates per sample (~0.3 s on one core): a set that
    is to be read faster than that is held in memory after its first use (``cache=True`` -- what the page cache does for a
    dataset on disk), optionally in pinned memory (``pin=True``), from where the input pipeline copies it to the device
    without a staging pass.  ``prepare()`` fills the cache with a thread pool (numpy's generators release the GIL)."""

    def __init__(self, n_samples, H=544, W=960, N=8192, dsec=False, distinct=None, cache=False, pin=False, first_seed=1000, events=0):
        self.n, self.kw = n_samples, dict(H=H, W=W, N=N, dsec=dsec, events=events)
        if events:  # raw events, voxelised by the input pipeline: 10 bins x 2 polarities = the model's 20 event channels
            self.event_format = dict(bins=10, polarity=True, height=H, width=W, max_events=events)
        self.distinct = n_samples if distinct is None else max(1, min(distinct, n_samples))
        self.first_seed, self.pin = first_seed, pin
        self.cache = {} if (cache or pin) else None

    def __len__(self):
        return self.n

    def _make(self, slot):
        sample = {k: torch.from_numpy(v) for k, v in frame_pair(self.first_seed + slot, **self.kw).items()}
        return {k: v.pin_memory() for k, v in sample.items()} if self.pin else sample

    def __getitem__(self, i):
        slot = i % self.distinct
        if self.cache is None:
            return self._make(slot)
        if slot not in self.cache:
            self.cache[slot] = self._make(slot)
        return self.cache[slot]

    def prepare(self, threads=None, indices=None):
        """Generates the distinct samples of a cached set now (``indices``: only those these samples map to -- a rank's
        shard); returns the seconds it took."""
        import os
        import time
        from concurrent.futures import ThreadPoolExecutor
        if self.cache is None:
            return 0.0
        t0 = time.perf_counter()
        wanted = range(self.distinct) if indices is None else sorted({i % self.distinct for i in indices})
        missing = [s for s in wanted if s not in self.cache]
        if threads is None:  # this rank's share of the cores the process may use (eight ranks x 16 generator threads on a 16-core quota otherwise)
            from .runtime import cores_per_rank
            threads = cores_per_rank()
        with ThreadPoolExecutor(max(1, min(threads, 16, len(missing) or 1))) as pool:
            for slot, sample in zip(missing, pool.map(self._make, missing)):
                self.cache[slot] = sample
        return time.perf_counter() - t0


# ------------------------------------------------------------------ seeded parameters (no checkpoint ships here)
def fill_params(shapes, seed):
    """Deterministic parameters for a module.  ``shapes``: list of (key, shape).  Every key draws from
    its own stream, default_rng(seed + crc32(key)), so the values do not depend on the order in which
    a module registers its parameters.  Weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1), running_var and
    norm scales in [0.5, 1.5), integer buffers (num_batches_tracked) zero."""
    import zlib
    out = {}
    for key, shape in shapes:
        shape = tuple(shape)
        r = np.random.default_rng(seed + zlib.crc32(key.encode()))
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros(shape, np.int64)
        elif key.endswith("running_var"):
            out[key] = r.uniform(0.5, 1.5, shape).astype(np.float32)
        elif key.endswith("running_mean") or key.endswith("bias"):
            out[key] = (r.standard_normal(shape) * 0.1).astype(np.float32)
        elif key.endswith("norm_fn.weight") or key.endswith("body.weight") or key.endswith("temperature"):
            out[key] = r.uniform(0.5, 1.5, shape).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            out[key] = (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    return out


MODEL_SEED = 4242
# Down-scaling of a few weight groups so that a random-init RPEFlow stays well conditioned: with plain
# N(0, 1/fan_in) weights the k-sums of PointConv / Correlation3D amplify ~10x per pyramid level and the
# 3-D decoder features reach 1e8, where fp32 rounding alone moves the 2-D flow by 1e-2.
MODEL_SCALES = [("conv_last_", 0.05), ("linear.", 0.25), ("weight_net1.convs.2", 0.15), ("weight_net2.convs.2", 0.15)]


def model_params(shapes, seed=MODEL_SEED):
    """Seeded parameters for the full model (reference and counterpart alike), keyed by state-dict name.  ``seed``: another
    fill of the same distribution (the stress goldens use a second one)."""
    params = fill_params(shapes, seed)
    for k in params:
        for pattern, scale in MODEL_SCALES:
            if pattern in k and k.endswith("weight") and params[k].ndim > 1:
                params[k] = (params[k] * np.float32(scale)).astype(np.float32)
    return params


def load_seeded_parameters(model):
    """Fill ``model`` (rpeflow_amd.model.RPEFlow or the reference's) with model_params: the parameters of every
    committed model golden, of the parity tests and of bench.py."""
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in model_params(shapes).items()}, strict=True)
    return model
