"""The call-trace fixtures (tests/golden/call_trace{,_stress}.{json,npz}): how a recorded argument becomes a tensor again.

tests/golden/make_golden.py call_trace wraps the hot-path names of the imported reference while its model runs one forward
and records every call the reference model itself makes: which function, from which reference line, which arguments went
positionally and which by keyword, and for every tensor argument its shape, dtype, STRIDES and storage offset (an expanded
mesh grid has batch stride 0, build_pc_pyramid hands over a transposed view, ...), its values and the call's output.
This module is shared by the generator and by the tests that replay the trace (tests/test_gpu_call_trace.py through
rpeflow_amd on the GPU, tests/test_call_trace_cpu.py through the oracle); it holds no reference code.

What is stored, per tensor (the "t" entries of call_trace.json -> arrays of call_trace.npz):
* geometry (coordinates, flows, grids), index tensors and every tensor of at most FULL_BYTES: the recorded values, in full
  (deduplicated by content: an output that is the next call's input is stored once);
* a larger feature tensor: NOT its values (the trace would be > 100 MB).  The generator replaces it by seeded values of the
  same shape, strides, mean and spread ("synthetic": seed, scale, shift -> ``synthetic_values``), re-runs the REFERENCE
  function on the arguments so changed and records that output; the stored output is the reference's for exactly the
  arguments a test rebuilds.  Which arguments are synthetic is written in each call's record;
* a float output above SAMPLE_ABOVE elements is stored at N_SAMPLES seeded flat positions (``sample_positions``).
"""
import json
import os

import numpy as np
import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_BYTES = 96 * 1024        # feature tensors up to this size keep their recorded values
SAMPLE_ABOVE = 16384          # float outputs with more elements are stored at seeded positions
N_SAMPLES = 4096

DTYPES = {"float32": torch.float32, "int64": torch.int64, "float64": torch.float64, "int32": torch.int32, "bool": torch.bool}


def synthetic_values(seed, shape, scale, shift):
    """Seeded stand-in for a large feature tensor: fl(fl(z * scale) + shift), z ~ N(0,1) float32 from numpy's PCG64 stream."""
    z = np.random.default_rng(seed).standard_normal(tuple(shape), dtype=np.float32)
    return (z * np.float32(scale) + np.float32(shift)).astype(np.float32)


def sample_positions(seed, numel, n=N_SAMPLES):
    return np.random.default_rng(seed).integers(0, numel, n)


def storage_extent(shape, strides, offset):
    """Elements of storage a (shape, strides, offset) view can touch."""
    if any(s == 0 for s in shape):
        return offset + 1
    return offset + 1 + sum((s - 1) * st for s, st in zip(shape, strides))


def strided_tensor(values, strides, offset, device):
    """A tensor on ``device`` with the given values, STRIDES and storage offset (``values``: contiguous numpy array of the
    logical shape).  Dimensions of stride 0 (expanded) are written once through their first index."""
    v = torch.from_numpy(np.ascontiguousarray(values))
    shape = list(v.shape)
    buf = torch.zeros(storage_extent(shape, strides, offset), dtype=v.dtype, device=device)
    # a writable window: every stride-0 dimension collapsed to its first index (all its copies are one memory location)
    w_shape = [1 if (st == 0 and s > 1) else s for s, st in zip(shape, strides)]
    window = buf.as_strided(w_shape, strides, offset)
    src = v
    for d, (s, st) in enumerate(zip(shape, strides)):
        if st == 0 and s > 1:
            src = src.narrow(d, 0, 1)
    window.copy_(src.to(device))
    out = buf.as_strided(shape, strides, offset)
    return out


TRACES = ("call_trace", "call_trace_stress")  # tests/golden/make_golden.py TRACE_CASES


class Trace:
    """<name>.json + <name>.npz."""

    def __init__(self, name="call_trace", path=HERE):
        with open(os.path.join(path, name + ".json")) as f:
            self.meta = json.load(f)
        self.arrays = np.load(os.path.join(path, name + ".npz"))
        self.calls = self.meta["calls"]
        self.name = name

    def parameters(self, path=HERE):
        """The seeded parameter fill the trace was recorded with, by state-dict key."""
        from tests import inputs as I
        keys = json.load(open(os.path.join(path, "state_dict_keys.json")))
        shapes = [(k, tuple(shape)) for k, shape, _ in keys]
        return I.model_params(shapes, **({"seed": I.STRESS_MODEL_SEED} if self.meta["case"].get("model_seed") == "stress" else {}))

    def values(self, t):
        """The logical values of tensor record ``t`` (numpy, recorded dtype)."""
        if "synthetic" in t:
            s = t["synthetic"]
            return synthetic_values(s["seed"], t["shape"], s["scale"], s["shift"])
        a = self.arrays[t["key"]]
        want = np.dtype(t["dtype"])
        return a.astype(want) if a.dtype != want else a

    def tensor(self, t, device):
        return strided_tensor(self.values(t).reshape(t["shape"]), t["strides"], t["offset"], device)

    def arguments(self, call, device):
        """(args, kwargs) as the reference passed them: positional / keyword split, strides and offsets as recorded,
        one tensor OBJECT for arguments that were one object in the reference's call (k_nearest_neighbor(xyz1, xyz1, ...))."""
        built, args, kwargs = {}, [], {}
        for a in call["args"]:
            if a["kind"] == "tensor":
                if "same_as" in a:
                    val = built[a["same_as"]]
                else:
                    val = self.tensor(a["t"], device)
                built[a["name"]] = val
            elif a["kind"] == "tensor_list":
                val = [self.tensor(t, device) for t in a["t"]]
            elif a["kind"] == "none":
                val = None
            else:
                val = a["value"]
            if a["passed"] == "kw":
                kwargs[a["name"]] = val
            else:
                args.append(val)
        return args, kwargs

    def output(self, o):
        """Expected output record -> (numpy values, flat positions or None)."""
        vals = self.values(o)
        pos = sample_positions(o["sampled"]["seed"], int(np.prod(o["shape"]))) if "sampled" in o else None
        return vals, pos


def compare_output(got, o, trace, exact, atol=0.0, rtol=0.0, what=""):
    """``got`` (torch tensor) against output record ``o``: shape, dtype, then values (at the stored positions when sampled)."""
    assert list(got.shape) == o["shape"], f"{what}: shape {list(got.shape)} vs {o['shape']}"
    assert str(got.dtype).replace("torch.", "") == o["dtype"], f"{what}: dtype {got.dtype} vs {o['dtype']}"
    want, pos = trace.output(o)
    g = got.detach().contiguous().cpu().numpy().reshape(-1)
    if pos is not None:
        g = g[pos]
    want = want.reshape(-1)
    if exact:
        if want.dtype.kind == "f":
            same = (g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))
        else:
            same = g == want
        assert same.all(), f"{what}: {int((~same).sum())} of {same.size} elements differ"
        return 0.0
    err = np.abs(g.astype(np.float64) - want.astype(np.float64))
    bound = atol + rtol * np.abs(want.astype(np.float64))
    assert (err <= bound).all(), f"{what}: max |diff| {err.max():.3e} (bound {atol:g} + {rtol:g}|ref|), {int((err > bound).sum())} of {err.size} outside"
    return float(err.max())
