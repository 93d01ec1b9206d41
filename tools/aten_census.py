#!/usr/bin/env python3
"""Which generic ATen operators does the hot-path operator sequence (or the full forward) launch, and from which source line?

Runs WITHOUT a GPU: the HIP entry points are replaced by no-ops (their outputs are allocated as usual and left
uninitialised), tensors live on the CPU, and a TorchDispatchMode records every ATen call that would be a kernel launch on
the device (views, allocations and metadata ops are skipped) with its shapes and the innermost rpeflow_amd / bench source
line.  This is a census of the glue between the HIP launches, not a numerical run.

    python tools/aten_census.py [hotpath|forward] [--batch 4]
"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from rpeflow_amd import _lib  # noqa: E402

VIEWS = {"view", "_unsafe_view", "reshape", "slice", "select", "expand", "transpose", "permute", "t", "unsqueeze", "squeeze", "as_strided",
         "alias", "detach", "empty", "empty_like", "empty_strided", "new_empty", "split", "split_with_sizes", "unbind", "narrow", "chunk",
         "lift_fresh", "_local_scalar_dense", "is_nonzero", "item", "size", "stride", "numel", "sym_size", "resolve_conj", "resolve_neg",
         "unfold", "view_as", "_reshape_alias", "unsafe_split", "unsafe_chunk", "movedim", "flatten", "is_pinned", "_has_compatible_shallow_copy_type",
         "equal", "real", "new_empty_strided", "set_", "is_same_size", "result_type", "can_cast", "promote_types"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.on = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if self.on and name not in VIEWS:
            shapes = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))[:3]
            if not shapes and isinstance(args[0] if args else None, (list, tuple)):
                shapes = tuple(tuple(a.shape) for a in args[0] if torch.is_tensor(a))[:4]
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if ("rpeflow_amd" in fr.filename or fr.filename.endswith("bench.py")) and "aten_census" not in fr.filename:
                    where = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
                    break
            self.rows[(name, where, str(shapes))] += 1
        return out


def patch_for_cpu():
    class Fake:
        def __getattr__(self, name):
            if name.endswith("_bytes") or name.endswith("_floats") or name.endswith("_doubles"):
                return lambda *a: 1 << 20
            return lambda *a: 0
    fake = Fake()
    _lib.lib = lambda: fake
    _lib.require_gpu = lambda *t, op=None: t[0].device
    _lib.stream_of = lambda t: None
    _lib.check = lambda rc, what: None
    import contextlib
    torch.Tensor.is_cuda = property(lambda self: True)  # the modules' "x.is_cuda" branches take their device side
    torch.cuda.device = lambda d: contextlib.nullcontext()
    torch.cuda.is_current_stream_capturing = lambda: False
    torch.cuda.current_stream = lambda d=None: type("S", (), {"cuda_stream": 0})()


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "hotpath"
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
    patch_for_cpu()
    census = Census()
    if what == "hotpath":
        from rpeflow_amd.hotpath import HotPathWorkload
        wl = HotPathWorkload(batch=batch, height=544, width=960, n_points=8192, device="cpu")
        step = lambda: wl()
    else:
        import bench
        from rpeflow_amd.model import RPEFlow
        model = RPEFlow().eval()
        model.overlap_streams = False  # (in stream order: the census counts launches, not their placement)
        b = bench.make_batch(batch, "cpu")
        step = lambda: model(b)
    with torch.no_grad():
        step()  # caches (packed weights, grids) fill outside the census
        with census:
            census.on = True
            step()
    total = sum(census.rows.values())
    print("ATen calls that launch on the device: %d per step" % total)
    by_op = collections.Counter()
    for (name, where, shapes), n in census.rows.items():
        by_op[name] += n
    print(dict(by_op.most_common()))
    for (name, where, shapes), n in sorted(census.rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("%3d  %-28s %-24s %s" % (n, name, where, shapes[:110]))


if __name__ == "__main__":
    main()
