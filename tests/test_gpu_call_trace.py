"""Replay of the calls the reference MODEL itself makes into the hot path, through rpeflow_amd on the GPU.

north_star: "exposed through the same Python operator signatures so models/RPEFlow.py ... consume them unchanged".
tests/golden/call_trace{,_stress}.{json,npz} (tests/golden/make_golden.py call_trace; format: tests/trace_io.py) hold all 324
calls one forward of the imported reference makes -- once in the model goldens' regime (B = 2), once off it (second parameter
fill, large motion, points outside the frame, B = 1) -- 43 k_nearest_neighbor (RPEFlow_core.py:329-331, pwc3d_core.py:81,
pointconv.py:46, utils.py:148), furthest_point_sampling (pwc3d_core.py:13), 5 correlation2d (RPEFlow_core.py:362), 156
gathers, 13 knn_interpolation, 4 + 4 warps, 45 grid_sample_wrapper, 20 project_feat_with_nn_corr, build_pc_pyramid, 20
PointConv, 5 Correlation3D, 5 FlowEstimator3D and 2 FeaturePyramid3D forwards -- with the argument patterns of the
reference's own call sites: positional or keyword (pwc3d_core.py:81 is keyword-only), the channel-first [B,2,HW] mesh grid
with keyword k=1 (contiguous, as it turns out: torch.stack materialises the expanded bases, utils.py:176-178),
build_pc_pyramid's transposed view and prefix slices of the sampling order, PointConv's transposed feature views, the channel
slice backwarp_3d receives as flow, one tensor object passed twice.  Every argument is rebuilt WITH THE
RECORDED STRIDES AND STORAGE OFFSET, the rpeflow_amd counterpart is called the way the reference called its own function, and
the result is held against the reference's output: indices and gathers bit for bit, floating point within the bound of the
operator's golden test.  (What the fixtures hold -- counts, call forms, strided arguments -- is asserted without a GPU in
tests/test_call_trace_cpu.py.)
"""
import numpy as np
import pytest
import torch

from tests import trace_io as TIO

pytestmark = pytest.mark.gpu

import rpeflow_amd.csrc as C  # noqa: E402
from rpeflow_amd import pointconv as PC  # noqa: E402
from rpeflow_amd import pwc3d_core as P3  # noqa: E402
from rpeflow_amd import utils as U  # noqa: E402

DEV = "cuda:0"
FUNCTIONS = {
    "k_nearest_neighbor": C.k_nearest_neighbor, "furthest_point_sampling": C.furthest_point_sampling, "correlation2d": C.correlation2d,
    "batch_indexing_channel_first": U.batch_indexing_channel_first, "batch_indexing_channel_last": U.batch_indexing_channel_last,
    "knn_interpolation": U.knn_interpolation, "backwarp_3d": U.backwarp_3d, "backwarp_2d": U.backwarp_2d,
    "grid_sample_wrapper": U.grid_sample_wrapper, "project_feat_with_nn_corr": U.project_feat_with_nn_corr,
    "build_pc_pyramid": P3.build_pc_pyramid,
}
CLASSES = {"PointConvDownSampling": PC.PointConvDownSampling, "PointConvNoSampling": PC.PointConvNoSampling,
           "Correlation3D": P3.Correlation3D, "FlowEstimator3D": P3.FlowEstimator3D, "FeaturePyramid3D": P3.FeaturePyramid3D}
# fn -> None (bit for bit) or (absolute bound relative to max(1, max |reference|), relative bound per element)
BOUNDS = {
    "k_nearest_neighbor": None, "furthest_point_sampling": None, "batch_indexing_channel_first": None, "batch_indexing_channel_last": None,
    "build_pc_pyramid": None,
    "correlation2d": (5e-6, 0.0), "knn_interpolation": (5e-6, 0.0), "backwarp_3d": (1e-5, 0.0), "backwarp_2d": (5e-6, 0.0),
    "grid_sample_wrapper": (5e-6, 0.0), "project_feat_with_nn_corr": (5e-6, 0.0),
    # fp32 sums of thousands of terms in another order (the module goldens' rule, tests/test_gpu_glue.py)
    "PointConvDownSampling.forward": (2e-6, 1e-4), "PointConvNoSampling.forward": (2e-6, 1e-4), "Correlation3D.forward": (2e-6, 1e-4),
    "FlowEstimator3D.forward": (2e-6, 1e-4), "FeaturePyramid3D.forward": (2e-6, 1e-4),
}
EXPECTED_COUNTS = {"k_nearest_neighbor": 43, "furthest_point_sampling": 1, "correlation2d": 5, "knn_interpolation": 13, "backwarp_2d": 4,
                   "backwarp_3d": 4, "grid_sample_wrapper": 45, "project_feat_with_nn_corr": 20, "PointConvDownSampling.forward": 10,
                   "PointConvNoSampling.forward": 10, "Correlation3D.forward": 5, "FlowEstimator3D.forward": 5, "FeaturePyramid3D.forward": 2,
                   "build_pc_pyramid": 1, "batch_indexing_channel_first": 136, "batch_indexing_channel_last": 20}


@pytest.fixture(scope="module", params=TIO.TRACES)
def trace(request):
    """Both recorded forwards: the model goldens' regime (B = 2) and the stress one (second parameter fill, large motion, ~5 % of
    the points projecting outside the frame, B = 1) -- tests/golden/make_golden.py TRACE_CASES."""
    return TIO.Trace(request.param)


@pytest.fixture(scope="module")
def parameters(trace):
    """The seeded parameter fill the trace was recorded with, by state-dict key."""
    return trace.parameters()


def check_output(got, o, trace, bound, what):
    if o["kind"] == "list":
        assert isinstance(got, (list, tuple)) and len(got) == len(o["items"]), what
        return max([check_output(g, item, trace, bound, "%s[%d]" % (what, i)) for i, (g, item) in enumerate(zip(got, o["items"]))] + [0.0])
    if bound is None or not got.dtype.is_floating_point:
        return TIO.compare_output(got, o, trace, exact=True, what=what)
    want, _ = trace.output(o)
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    return TIO.compare_output(got, o, trace, exact=False, atol=bound[0] * scale, rtol=bound[1], what=what) / scale


def build_module(call, parameters):
    m = call["module"]
    module = CLASSES[m["cls"]](**m["ctor"])
    prefix = m["name"] + "."
    state = {k[len(prefix):]: torch.from_numpy(v) for k, v in parameters.items() if k.startswith(prefix)}
    module.load_state_dict(state, strict=True)
    return module.to(DEV).eval()


@torch.no_grad()
def replay(trace, call, parameters):
    args, kwargs = trace.arguments(call, DEV)
    if "module" in call:
        got = build_module(call, parameters).forward(*args, **kwargs)
    else:
        got = FUNCTIONS[call["fn"]](*args, **kwargs)
    what = "call %d %s from %s" % (call["index"], call["fn"], call["site"])
    return check_output(got, call["out"], trace, BOUNDS[call["fn"]], what)


@pytest.mark.parametrize("fn", sorted(EXPECTED_COUNTS))
def test_replay_reference_calls(trace, parameters, fn):
    calls = [c for c in trace.calls if c["fn"] == fn]
    assert len(calls) == EXPECTED_COUNTS[fn]
    worst = 0.0
    for call in calls:
        worst = max(worst, replay(trace, call, parameters))
    sites = sorted({c["site"] for c in calls})
    print("\n%s / %s: %d calls from %s replayed; %s" % (trace.name, fn, len(calls), ", ".join(sites),
                                                   "bit for bit" if BOUNDS[fn] is None else "worst error %.2e of the output scale" % worst))
