"""The model counterpart (rpeflow_amd/model.py) against the reference model's golden output.

CPU: the module tree / state-dict must equal the reference's, and the wiring around the hot path
(pyramids, fusers, Restormer blocks, decode loop, IDS) is checked by running the model with the
PyTorch-CPU port of the hot-path ops (oracle/torch_ref.py) substituted for the HIP ones.
GPU: the real thing -- HIP kernels -- on the same sample."""
import json
import os

import numpy as np
import pytest
import torch

from tests import inputs as I



def reference_keys(golden_dir):
    return json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))


def seeded_state(model):
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    return {k: torch.from_numpy(v) for k, v in I.model_params(shapes).items()}


def sample_batch(device):
    s = I.frame_pair(1000, H=128, W=192, N=8192)
    return {k: torch.from_numpy(v)[None].to(device) for k, v in s.items()}


def epe(pred, target):
    return float(np.sqrt(((pred - target) ** 2).sum(1)).mean())


def test_state_dict_matches_reference(golden_dir):
    from rpeflow_amd.model import RPEFlow
    mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in RPEFlow().state_dict().items()]
    ref = reference_keys(golden_dir)
    assert len(mine) == len(ref) == 1123
    assert mine == ref  # same names, shapes, dtypes, in the same order


@torch.no_grad()
def test_model_wiring_on_cpu_with_ported_ops(golden_dir):
    from types import SimpleNamespace
    from oracle import torch_ref as R
    import rpeflow_amd.model as M
    from rpeflow_amd.hotpath import OP_NAMES
    model = M.RPEFlow(ops=SimpleNamespace(**{n: getattr(R, n) for n in OP_NAMES})).eval()
    model.load_state_dict(seeded_state(model), strict=True)
    out = model(sample_batch("cpu"))
    g = np.load(os.path.join(golden_dir, "model_128x192.npz"))
    f2, f3 = out["flow_2d"].numpy(), out["flow_3d"].numpy()
    assert f2.shape == g["flow_2d"].shape and f3.shape == g["flow_3d"].shape
    # same ops as the reference on the same host: agreement to fp32 rounding of re-associated sums
    assert np.abs(f2 - g["flow_2d"]).max() < 2e-3 and np.abs(f3 - g["flow_3d"]).max() < 2e-3
    s = I.frame_pair(1000, H=128, W=192, N=8192)
    assert abs(epe(f2, s["flow_2d"][None, :2]) - epe(g["flow_2d"], s["flow_2d"][None, :2])) < 1e-4
    assert abs(epe(f3, s["flow_3d"][None]) - epe(g["flow_3d"], s["flow_3d"][None])) < 1e-4


@pytest.mark.gpu
@torch.no_grad()
def test_model_on_gpu_matches_reference_golden(golden_dir):
    """north_star: EPE2D/EPE3D within 1e-4 of the reference CPU path on identical inputs."""
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow(ids_on_host=True).eval()
    model.load_state_dict(seeded_state(model), strict=True)
    model = model.to("cuda:0")
    out = model(sample_batch("cuda:0"))
    g = np.load(os.path.join(golden_dir, "model_128x192.npz"))
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    s = I.frame_pair(1000, H=128, W=192, N=8192)
    d2, d3 = np.abs(f2 - g["flow_2d"]), np.abs(f3 - g["flow_3d"])
    print("max |d flow_2d|", d2.max(), "max |d flow_3d|", d3.max(), "mean", d2.mean(), d3.mean())
    e2 = abs(epe(f2, s["flow_2d"][None, :2]) - epe(g["flow_2d"], s["flow_2d"][None, :2]))
    e3 = abs(epe(f3, s["flow_3d"][None]) - epe(g["flow_3d"], s["flow_3d"][None]))
    print("EPE2D diff", e2, "EPE3D diff", e3)
    assert e2 < 1e-4 and e3 < 1e-4
    # element-wise the two runs cannot be identical: KNNs on warped clouds depend on conv outputs, and a
    # 1e-6 difference there flips a near-tied neighbour now and then; bound the mean.
    assert d2.mean() < 1e-3 and d3.mean() < 1e-3


@pytest.mark.gpu
@torch.no_grad()
def test_model_on_gpu_dsec_shape(golden_dir):
    """BASELINE config 5 shapes: 480x640 frames (resized to 512x640 inside the model), 4-channel flow_3d target."""
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow(ids_on_host=True).eval()
    model.load_state_dict(seeded_state(model), strict=True)
    model = model.to("cuda:0")
    s = I.frame_pair(2000, H=480, W=640, N=8192, dsec=True)
    out = model({k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()})
    g = np.load(os.path.join(golden_dir, "model_dsec_480x640.npz"))
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert f2.shape == (1, 2, 480, 640)
    e2, e3 = epe(f2, s["flow_2d"][None, :2]), epe(f3, s["flow_3d"][None, :3])
    print("dsec EPE2D diff", abs(e2 - float(g["epe2d"])), "EPE3D diff", abs(e3 - float(g["epe3d"])))
    assert abs(e2 - float(g["epe2d"])) < 1e-4 and abs(e3 - float(g["epe3d"])) < 1e-4
    assert np.abs(f2[:, :, ::8, ::8] - g["flow_2d_s8"]).mean() < 1e-3 and np.abs(f3 - g["flow_3d"]).mean() < 1e-3


@pytest.mark.gpu
@torch.no_grad()
def test_model_graph_replay_and_single_stream_agree(golden_dir):
    """The forward as one HIP graph with its two-stream branches (encoder overlap, 2-D / 3-D decode chains) must give
    what the in-order single-stream forward gives, and replays must be repeatable."""
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow().eval()
    model.load_state_dict(seeded_state(model), strict=True)
    model = model.to("cuda:0")
    batch = sample_batch("cuda:0")
    model.overlap_streams = False
    ref = model(batch)
    model.overlap_streams = True
    for _ in range(2):
        eager = model(batch)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = model(batch)
    s = I.frame_pair(1000, H=128, W=192, N=8192)
    g = np.load(os.path.join(golden_dir, "model_128x192.npz"))
    runs = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        runs.append((out["flow_2d"].cpu().numpy().copy(), out["flow_3d"].cpu().numpy().copy()))
    for f2, f3 in runs[1:]:  # not bit-equal: rocBLAS / MIOpen pick split-K kernels that add with atomics
        print("replay-to-replay max |d|", np.abs(f2 - runs[0][0]).max(), np.abs(f3 - runs[0][1]).max())
        assert np.abs(f2 - runs[0][0]).mean() < 1e-5 and np.abs(f3 - runs[0][1]).mean() < 1e-5
    for got in (eager, {"flow_2d": torch.from_numpy(runs[0][0]), "flow_3d": torch.from_numpy(runs[0][1])}):
        f2, f3 = got["flow_2d"].cpu().numpy(), got["flow_3d"].cpu().numpy()
        assert abs(epe(f2, s["flow_2d"][None, :2]) - epe(ref["flow_2d"].cpu().numpy(), s["flow_2d"][None, :2])) < 1e-4
        assert abs(epe(f3, s["flow_3d"][None]) - epe(ref["flow_3d"].cpu().numpy(), s["flow_3d"][None])) < 1e-4
        # (the reference golden is compared in test_model_on_gpu_matches_reference_golden, which needs the IDS transform
        # on the host and therefore cannot be captured; here a loose bound guards against gross errors only)
        assert abs(epe(f2, s["flow_2d"][None, :2]) - epe(g["flow_2d"], s["flow_2d"][None, :2])) < 5e-3
        assert abs(epe(f3, s["flow_3d"][None]) - epe(g["flow_3d"], s["flow_3d"][None])) < 5e-3
