"""The call-trace fixture without a GPU: what it holds, how its arguments are rebuilt, and the CPU oracle against it.

tests/golden/call_trace{,_stress}.{json,npz} record every hot-path call of one forward of the imported reference model, once per
parameter fill (tests/golden/make_golden.py call_trace, tests/trace_io.py).  Here: the fixture's inventory and the argument patterns of the
reference's call sites; ``strided_tensor`` (the rebuild with recorded strides / offsets); and every recorded FUNCTION call
replayed through oracle/oracle.py -- 43 KNN and the FPS call index for index, 156 gathers bit for bit, samplers / warps /
interpolation / correlation within the oracle's own golden bounds -- which pins the oracle to 2 x 301 more reference outputs,
on the reference model's own arguments.  The GPU replay of the same trace is tests/test_gpu_call_trace.py.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import trace_io as TIO

EXPECTED_COUNTS = {"k_nearest_neighbor": 43, "furthest_point_sampling": 1, "correlation2d": 5, "knn_interpolation": 13, "backwarp_2d": 4,
                   "backwarp_3d": 4, "grid_sample_wrapper": 45, "project_feat_with_nn_corr": 20, "PointConvDownSampling.forward": 10,
                   "PointConvNoSampling.forward": 10, "Correlation3D.forward": 5, "FlowEstimator3D.forward": 5, "FeaturePyramid3D.forward": 2,
                   "build_pc_pyramid": 1, "batch_indexing_channel_first": 136, "batch_indexing_channel_last": 20}


@pytest.fixture(scope="module", params=TIO.TRACES)
def trace(request):
    """Both recorded forwards: the model goldens' regime (B = 2) and the stress one (second parameter fill, large motion, points
    outside the frame, B = 1) -- tests/golden/make_golden.py TRACE_CASES."""
    return TIO.Trace(request.param)


def test_trace_holds_every_call_of_one_reference_forward(trace):
    counts = {}
    for call in trace.calls:
        counts[call["fn"]] = counts.get(call["fn"], 0) + 1
    assert counts == EXPECTED_COUNTS
    knn = [c for c in trace.calls if c["fn"] == "k_nearest_neighbor"]
    # RPEFlow_core.py:329-330: projected points [B,2,N] against the flattened mesh grid [B,2,HW], k by keyword
    grid_calls = [c for c in knn if c["site"] in ("models/RPEFlow_core.py:329", "models/RPEFlow_core.py:330")]
    assert len(grid_calls) == 10
    for c in grid_calls:
        pts, grid, k = c["args"]
        assert pts["t"]["shape"][1] == 2 and grid["t"]["shape"][1] == 2 and k == dict(name="k", passed="kw", kind="value", value=1)
    assert sum(all(a["passed"] == "kw" for a in c["args"]) for c in knn) == 5       # pwc3d_core.py:81: keyword-only
    assert sum("same_as" in c["args"][1] for c in knn) == 5                         # RPEFlow_core.py:331: (xyz1, xyz1, k=16)
    fps = [c for c in trace.calls if c["fn"] == "furthest_point_sampling"][0]
    assert fps["args"][0]["t"]["strides"][1:] == [1, 8192] and fps["site"] == "models/pwc3d_core.py:13"   # pc_both.transpose(1, 2)
    strided = {(c["fn"], c["site"]) for c in trace.calls for a in c["args"] if a["kind"] == "tensor" and "t" in a
               and tuple(a["t"]["strides"]) != torch.empty(a["t"]["shape"]).stride()}
    assert {("PointConvNoSampling.forward", "models/pwc3d_core.py:141"), ("backwarp_3d", "models/RPEFlow_core.py:358"),
            ("batch_indexing_channel_last", "models/pointconv.py:55"), ("batch_indexing_channel_first", "models/pwc3d_core.py:25")} <= strided
    # every call site is reference code, every nested call knows its parent
    assert all(c["site"].startswith("models/") for c in trace.calls)
    assert all(c["parent"] is None or c["parent"] < c["index"] for c in trace.calls)


def test_strided_tensor_rebuilds_views_as_recorded():
    base = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    views = [base, base.transpose(1, 2), base[:, 1:, :], base[:, :, 1:4], base[:, :, ::2], base[:1].expand(4, 3, 5),
             base[:, :, :1].expand(2, 3, 7), base.permute(2, 0, 1)[1:], torch.empty(2, 0, 3)]
    for v in views:
        got = TIO.strided_tensor(v.contiguous().numpy(), list(v.stride()), int(v.storage_offset()), "cpu")
        assert got.shape == v.shape and got.stride() == v.stride() and got.storage_offset() == v.storage_offset()
        assert torch.equal(got, v)


def to_numpy(args, kwargs):
    conv = lambda v: v.numpy() if torch.is_tensor(v) else v   # (numpy keeps the strides of the view)
    return [conv(a) for a in args], {k: conv(v) for k, v in kwargs.items()}


ORACLE = {
    "k_nearest_neighbor": (O.k_nearest_neighbor, None), "furthest_point_sampling": (O.furthest_point_sampling, None),
    "batch_indexing_channel_first": (O.batch_indexing_channel_first, None), "batch_indexing_channel_last": (O.batch_indexing_channel_last, None),
    "correlation2d": (O.correlation2d, 2e-6), "backwarp_2d": (O.backwarp_2d, 2e-6), "grid_sample_wrapper": (O.grid_sample_wrapper, 2e-6),
    "knn_interpolation": (O.knn_interpolation, 5e-6), "backwarp_3d": (O.backwarp_3d, 1e-5), "project_feat_with_nn_corr": (O.project_feat_with_nn_corr, 2e-6),
}
ARG_NAMES = {"batch_indexing_channel_first": ("data", "indices"), "batch_indexing_channel_last": ("data", "indices")}


@pytest.mark.parametrize("fn", sorted(ORACLE))
def test_oracle_reproduces_the_recorded_reference_calls(trace, fn):
    oracle_fn, bound = ORACLE[fn]
    calls = [c for c in trace.calls if c["fn"] == fn]
    assert len(calls) == EXPECTED_COUNTS[fn]
    worst = 0.0
    for call in calls:
        args, kwargs = to_numpy(*trace.arguments(call, "cpu"))
        got = torch.from_numpy(np.ascontiguousarray(oracle_fn(*args, **kwargs)))
        what = "call %d %s from %s" % (call["index"], fn, call["site"])
        if bound is None:
            TIO.compare_output(got, call["out"], trace, exact=True, what=what)
        else:
            want, _ = trace.output(call["out"])
            scale = max(1.0, float(np.abs(want).max()))
            worst = max(worst, TIO.compare_output(got, call["out"], trace, exact=False, atol=bound * scale, what=what) / scale)
    print("\n%s / %s: oracle = reference on %d recorded calls%s" % (trace.name, fn, len(calls), "" if bound is None else " (worst %.2e of the output scale)" % worst))
