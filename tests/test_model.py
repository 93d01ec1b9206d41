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


# GPU forward against the reference's CPU output: north_star's bound, |EPE - reference EPE| < 1e-4, on all three golden
# shapes.  It holds because the neighbour SETS are the reference's: k_nearest_neighbor there is matmul + torch.topk
# (wrapper.py:115-117), whose pick among candidates at exactly the k-th distance follows libstdc++'s partial_sort /
# nth_element, and the KNN kernel restates that.  Measured: 1.9e-6 / 1.9e-6 / 4.0e-5 on EPE2D
# (128x192, DSEC 480x640, 544x960), <= 5e-6 on EPE3D; what is left is MIOpen vs oneDNN convolution rounding.
GOLDEN_EPE_TOL = 1e-4


def with_reference_ids(batch, golden, device):
    """Feed the clouds exactly as the reference's host-side IDS transform produced them when the golden was made
    (torch.log differs by an ulp between CPU models, which flips FPS / KNN decisions downstream)."""
    batch = dict(batch)
    batch["pcs_ids"] = torch.from_numpy(np.concatenate([golden["pc1_ids"], golden["pc2_ids"]], axis=1)).to(device)
    return batch


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
    g = np.load(os.path.join(golden_dir, "model_128x192.npz"))
    out = model(with_reference_ids(sample_batch("cuda:0"), g, "cuda:0"))
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all()
    s = I.frame_pair(1000, H=128, W=192, N=8192)
    d2, d3 = np.abs(f2 - g["flow_2d"]), np.abs(f3 - g["flow_3d"])
    print("max |d flow_2d|", d2.max(), "max |d flow_3d|", d3.max(), "mean", d2.mean(), d3.mean())
    e2 = abs(epe(f2, s["flow_2d"][None, :2]) - epe(g["flow_2d"], s["flow_2d"][None, :2]))
    e3 = abs(epe(f3, s["flow_3d"][None]) - epe(g["flow_3d"], s["flow_3d"][None]))
    print("EPE2D diff", e2, "EPE3D diff", e3)
    assert e2 < GOLDEN_EPE_TOL and e3 < GOLDEN_EPE_TOL
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
    g = np.load(os.path.join(golden_dir, "model_dsec_480x640.npz"))
    out = model(with_reference_ids({k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()}, g, "cuda:0"))
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert f2.shape == (1, 2, 480, 640)
    e2, e3 = epe(f2, s["flow_2d"][None, :2]), epe(f3, s["flow_3d"][None, :3])
    print("dsec EPE2D diff", abs(e2 - float(g["epe2d"])), "EPE3D diff", abs(e3 - float(g["epe3d"])))
    assert abs(e2 - float(g["epe2d"])) < GOLDEN_EPE_TOL and abs(e3 - float(g["epe3d"])) < GOLDEN_EPE_TOL
    assert np.abs(f2[:, :, ::8, ::8] - g["flow_2d_s8"]).mean() < 5e-3 and np.abs(f3 - g["flow_3d"]).mean() < 1e-3


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
    # the constants the forwards above shared (the coarsest level's zero flows, the level-0 index, the pyramid's zero point)
    # still hold their values after eager runs on one and two streams, a capture and three replays: nobody wrote through them
    core = model.pwc_fusion_core
    assert len(core._zeros) > 0 and core.constants_intact()
    first = dict(core._zeros)
    with torch.inference_mode():  # inference tensors are keyed apart: an inference-mode forward must not hand its constants to autograd later
        model(batch)
    assert len(core._zeros) > len(first) and all(core._zeros[k] is v for k, v in first.items())
    core.clear_constants()
    assert not core._zeros and core.constants_intact()
    again = model(batch)
    assert torch.equal(again["flow_3d"], eager["flow_3d"])


@pytest.mark.gpu
@torch.no_grad()
def test_callers_own_operators_never_see_a_shared_constant():
    """RPEFlow(ops=...) with a caller's operators: the coarsest level's zero flows are fresh tensors every forward (a port may
    write into what it is given), so nothing is cached for them."""
    from rpeflow_amd.hotpath import native_ops
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow(ops=native_ops()).eval()   # the same callables, but handed in: treated as a caller's own
    model.load_state_dict(seeded_state(model), strict=True)
    model = model.to("cuda:0")
    out = model(sample_batch("cuda:0"))
    assert torch.isfinite(out["flow_3d"]).all() and not model.pwc_fusion_core._zeros


@pytest.mark.gpu
@torch.no_grad()
def test_model_on_gpu_full_size_frame(golden_dir):
    """BASELINE config 3 shape (544x960 + 8192 points): EPE2D / EPE3D within 1e-4 of the reference's CPU forward."""
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow(ids_on_host=True).eval()
    model.load_state_dict(seeded_state(model), strict=True)
    model = model.to("cuda:0")
    s = I.frame_pair(3000, H=544, W=960, N=8192)
    g = np.load(os.path.join(golden_dir, "model_544x960.npz"))
    out = model(with_reference_ids({k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()}, g, "cuda:0"))
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert f2.shape == (1, 2, 544, 960)
    e2, e3 = epe(f2, s["flow_2d"][None, :2]), epe(f3, s["flow_3d"][None, :3])
    print("full-size EPE2D diff", abs(e2 - float(g["epe2d"])), "EPE3D diff", abs(e3 - float(g["epe3d"])))
    assert abs(e2 - float(g["epe2d"])) < GOLDEN_EPE_TOL and abs(e3 - float(g["epe3d"])) < GOLDEN_EPE_TOL
    assert np.abs(f2[:, :, ::8, ::8] - g["flow_2d_s8"]).mean() < 5e-3 and np.abs(f3 - g["flow_3d"]).mean() < 1e-3


@pytest.mark.gpu
@torch.no_grad()
@pytest.mark.parametrize("name,H,W,seed,dsec", [("model_128x192", 128, 192, 1000, False), ("model_dsec_480x640", 480, 640, 2000, True),
                                                 ("model_544x960", 544, 960, 3000, False)])
def test_model_matches_pytorch_port_with_same_selection(golden_dir, name, H, W, seed, dsec):
    """The HIP hot path against the PyTorch port of the reference's op sequence (oracle/torch_ref.py, itself pinned to the
    reference goldens on the CPU) on the same GPU, both using the same KNN / FPS selection: what is left is the arithmetic
    of every hot-path operator and fusion, and EPE2D / EPE3D must agree within the north-star 1e-4."""
    from types import SimpleNamespace
    import rpeflow_amd.csrc as ops
    import rpeflow_amd.model as M
    from oracle import torch_ref as R
    from rpeflow_amd.hotpath import OP_NAMES
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    s = I.frame_pair(seed, H=H, W=W, N=8192, dsec=dsec)
    batch = with_reference_ids({k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()}, g, "cuda:0")
    saved = R.k_nearest_neighbor, R.furthest_point_sampling
    R.k_nearest_neighbor = lambda input_xyz, query_xyz, k, cpp_impl=True: ops.k_nearest_neighbor(input_xyz, query_xyz, k)
    R.furthest_point_sampling = lambda xyz, n_samples, cpp_impl=True: ops.furthest_point_sampling(xyz, n_samples)
    try:
        flows = []
        for namespace in (None, SimpleNamespace(**{n: getattr(R, n) for n in OP_NAMES})):
            model = M.RPEFlow(ops=namespace).eval()
            model.load_state_dict(seeded_state(model), strict=True)
            out = model.to("cuda:0")(batch)
            flows.append((out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()))
    finally:
        R.k_nearest_neighbor, R.furthest_point_sampling = saved
    t2, t3 = s["flow_2d"][None, :2], s["flow_3d"][None, :3]
    d2, d3 = abs(epe(flows[0][0], t2) - epe(flows[1][0], t2)), abs(epe(flows[0][1], t3) - epe(flows[1][1], t3))
    print(name, "EPE2D diff", d2, "EPE3D diff", d3)
    assert d2 < 1e-4 and d3 < 1e-4


@pytest.mark.gpu
def test_sampling_one_batch_ahead_changes_nothing():
    """forward_ahead(A, order, next=B): the output is forward(A)'s, ``order`` leaves as sample_order(B) -- eagerly and
    replayed from one HIP graph over three different batches (the harness's schedule)."""
    from rpeflow_amd.model import RPEFlow
    torch.manual_seed(0)
    model = RPEFlow().to("cuda:0").eval()
    batches = [{k: torch.from_numpy(v)[None].to("cuda:0") for k, v in I.frame_pair(4000 + i, H=128, W=192, N=8192).items()}
               for i in range(3)]
    plain = [{k: v.clone() for k, v in model(b).items()} for b in batches]
    # two runs of the forward differ by convolution-solver rounding only; another sampling order would be O(1) off
    same = lambda a, ref: ((a - ref).abs().mean() / (ref.abs().mean() + 1e-6)).item() < 1e-3
    orders = [model.sample_order(b) for b in batches]
    assert not torch.equal(orders[0], orders[1]) and not same(plain[1]["flow_3d"], plain[0]["flow_3d"])
    order = orders[0].clone()
    out = model.forward_ahead(batches[0], order, batches[1])
    torch.cuda.synchronize()
    assert torch.equal(order, orders[1])
    for key in ("flow_2d", "flow_3d"):
        assert same(out[key], plain[0][key])

    from rpeflow_amd.evaluate import GraphedForward
    replayed = GraphedForward(model, warmup=1)
    for i, b in enumerate(batches):
        out = replayed(b, batches[i + 1] if i + 1 < len(batches) else None)
        for key in ("flow_2d", "flow_3d"):
            assert same(out[key], plain[i][key]), (i, key)
    # an unannounced batch (and a wrongly announced one) still gets its own sampling
    out = replayed(batches[0], batches[2])
    assert same(out["flow_3d"], plain[0]["flow_3d"])
    out = replayed(batches[1])
    assert same(out["flow_3d"], plain[1]["flow_3d"])


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["one NaN coordinate", "whole cloud NaN"])
def test_a_corrupt_sample_turns_nan_and_leaves_the_others_alone(what):
    """A NaN in one sample's point cloud: that sample's flows come out NaN (the reference's evaluation masks NaN predictions,
    eval_withocc.py:86-87), every other sample of the batch keeps its bits (samples are independent in eval mode,
    eval_withocc.py:46), and no neighbour / sampling index leaves its cloud on the way (the gathers behind them do not check)."""
    from rpeflow_amd.model import RPEFlow
    torch.manual_seed(0)
    model = RPEFlow().to("cuda:0").eval()
    samples = [I.frame_pair(4100 + i, H=128, W=192, N=8192) for i in range(3)]
    batch = {k: torch.stack([torch.from_numpy(s[k]) for s in samples]).to("cuda:0") for k in samples[0]}
    with torch.no_grad():
        clean = {k: v.clone() for k, v in model(batch).items() if k in ("flow_2d", "flow_3d")}
        if what == "one NaN coordinate":
            batch["pcs"][1, 0, 17] = float("nan")
        else:
            batch["pcs"][1, :3] = float("nan")
        out = model(batch)
    torch.cuda.synchronize()
    for key in ("flow_2d", "flow_3d"):
        assert torch.isfinite(clean[key]).all()
        assert torch.equal(out[key][0], clean[key][0]) and torch.equal(out[key][2], clean[key][2]), key
        assert torch.isnan(out[key][1]).all(), key


@pytest.mark.gpu
def test_every_pyramid_level_matches_the_reference(golden_dir):
    """Per-level intermediates: the up-sampled flows of all five decoder levels (what RPEFlow_core.decode returns,
    RPEFlow_core.py:426-432) against the reference's on the 128x192 golden sample."""
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow(ids_on_host=True).eval()
    model.load_state_dict(seeded_state(model), strict=True)
    model = model.to("cuda:0")
    model.keep_levels = True
    g = np.load(os.path.join(golden_dir, "model_128x192.npz"))
    out = model(with_reference_ids(sample_batch("cuda:0"), g, "cuda:0"))
    assert len(out["levels_2d"]) == 5 and len(out["levels_3d"]) == 5
    for i in range(5):
        want2, want3 = g["level%d_flow_2d" % i], g["level%d_flow_3d" % i]
        got2, got3 = out["levels_2d"][i].cpu().numpy(), out["levels_3d"][i].cpu().numpy()
        assert got2.shape == want2.shape and got3.shape == want3.shape
        d2, d3 = np.abs(got2 - want2).mean(), np.abs(got3 - want3).mean()
        print("level %d: mean |d flow_2d| %.2e (|flow| %.2e), mean |d flow_3d| %.2e (|flow| %.2e)" % (i, d2, np.abs(want2).mean(), d3, np.abs(want3).mean()))
        assert d2 <= 1e-4 * max(1.0, np.abs(want2).mean()) and d3 <= 1e-4 * max(1.0, np.abs(want3).mean())  # measured: 2e-7 relative


@pytest.mark.gpu
@torch.no_grad()
def test_benched_dsec_configuration_matches_reference_golden(golden_dir):
    """bench.py --config dsec (BASELINE config 5 shapes on one GPU): batch 3 of 480x640 DSEC-shaped frame pairs (4-channel
    flow_3d targets, no occlusion mask), built and replayed exactly as bench.py does, against the reference's CPU forward."""
    import bench
    from rpeflow_amd.model import RPEFlow
    from rpeflow_amd.synthetic import load_seeded_parameters
    cfg = bench.CONFIGS["dsec"]
    g = np.load(os.path.join(golden_dir, cfg["golden"]))
    dev = torch.device("cuda", 0)
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batch = bench.make_batch(cfg["batch"], dev, first_seed=cfg["first_seed"], H=cfg["H"], W=cfg["W"], dsec=True)
    model(batch)
    order = model.sample_order(batch)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out = model.forward_ahead(batch, order, batch)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    d = bench.golden_epe_delta(out, batch, g)
    print("benched DSEC configuration, graph replay:", d)
    assert out["flow_2d"].shape == (3, 2, 480, 640)
    assert d["epe2d"] < GOLDEN_EPE_TOL and d["epe3d"] < GOLDEN_EPE_TOL
    assert d["mean_abs_flow_2d"] < 1e-3 and d["mean_abs_flow_3d"] < 1e-4


@pytest.mark.gpu
@torch.no_grad()
def test_benched_configuration_matches_reference_golden(golden_dir):
    """The configuration bench.py times, built the way bench.py builds it -- batch 4 of 544x960 frame pairs + 8192 points
    (seeds 1000..1003), seeded parameters, IDS transform on the device, furthest-point sampling one batch ahead
    (forward_ahead) inside ONE HIP graph, MIOpen's default solvers (no environment overrides anywhere in tests/) --
    against the reference's CPU forward on the same batch: |EPE2D|, |EPE3D| differences < 1e-4 (north_star)."""
    import bench
    from rpeflow_amd.model import RPEFlow
    from rpeflow_amd.synthetic import load_seeded_parameters
    assert not [k for k in os.environ if k.startswith("MIOPEN_DEBUG_")], "the benched configuration runs MIOpen's defaults"
    g = np.load(os.path.join(golden_dir, "model_bench_b4_544x960.npz"))
    dev = torch.device("cuda", 0)
    model = load_seeded_parameters(RPEFlow()).to(dev).eval()
    batch = bench.make_batch(4, dev, first_seed=1000)
    out = model(batch)
    eager = bench.golden_epe_delta(out, batch, g)
    order = model.sample_order(batch)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out = model.forward_ahead(batch, order, batch)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    replayed = bench.golden_epe_delta(out, batch, g)
    print("benched configuration: eager", eager, "graph replay", replayed)
    for d in (eager, replayed):
        assert d["epe2d"] < GOLDEN_EPE_TOL and d["epe3d"] < GOLDEN_EPE_TOL
        assert d["mean_abs_flow_2d"] < 1e-3 and d["mean_abs_flow_3d"] < 1e-4


# ---- the composition off the easy regime (round-3 review, weak #1): a SECOND seeded parameter fill and large-motion samples --
# pc2 = rigid motion of pc1 + N(0, 0.5^2), 5 % / 10 % of the two clouds projecting outside the frame, zero-mask and NaN pixels in
# the 2-D targets (tests/inputs.py frame_pair_stress).  Goldens: the imported reference's CPU forward (make_golden.py model_stress).
STRESS = [("model_128x192_stress", 5000, 128, 192, False), ("model_544x960_stress", 5001, 544, 960, False),
          ("model_dsec_480x640_stress", 5002, 480, 640, True)]


def stress_state(model):
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    return {k: torch.from_numpy(v) for k, v in I.model_params(shapes, seed=I.STRESS_MODEL_SEED).items()}


def test_stress_samples_leave_the_frame_and_carry_masked_targets():
    for _, seed, H, W, dsec in STRESS:
        s = I.frame_pair_stress(seed, H=H, W=W, N=8192, dsec=dsec)
        f, cx, cy = s["intrinsics"]
        for pc, lo, hi in ((s["pcs"][:3], 0.03, 0.08), (s["pcs"][3:], 0.06, 0.2)):
            u, v = pc[0] / pc[2] * f + cx, pc[1] / pc[2] * f + cy
            out = np.mean((u < 0) | (u > W - 1) | (v < 0) | (v > H - 1))
            assert lo < out < hi and pc[2].min() >= 1.0, (H, W, out)
        assert np.abs(s["flow_3d"][:3]).mean() > 0.3  # large motion: the cross-cloud searches are not near-self searches
        assert 0.1 < (s["flow_2d"][2] == 0).mean() < 0.2 and 0.005 < np.isnan(s["flow_2d"][0]).mean() < 0.02


@torch.no_grad()
def test_model_wiring_on_cpu_with_ported_ops_stress(golden_dir):
    from types import SimpleNamespace
    from oracle import torch_ref as R
    import rpeflow_amd.model as M
    from rpeflow_amd.hotpath import OP_NAMES
    name, seed, H, W, dsec = STRESS[0]
    model = M.RPEFlow(ops=SimpleNamespace(**{n: getattr(R, n) for n in OP_NAMES})).eval()
    model.load_state_dict(stress_state(model), strict=True)
    s = I.frame_pair_stress(seed, H=H, W=W, N=8192, dsec=dsec)
    out = model({k: torch.from_numpy(v)[None] for k, v in s.items()})
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    e2, e3 = I.masked_epes(out["flow_2d"].numpy()[0], out["flow_3d"].numpy()[0], s)
    assert abs(e2 - float(g["epe2d"])) < 1e-4 and abs(e3 - float(g["epe3d"])) < 1e-4


@pytest.mark.gpu
@torch.no_grad()
@pytest.mark.parametrize("name,seed,H,W,dsec", STRESS)
def test_model_on_gpu_matches_reference_golden_under_stress(golden_dir, name, seed, H, W, dsec):
    """north_star's bound -- |EPE - reference EPE| < 1e-4, EPEs counted as the evaluators count them (masked, NaN-free) -- with
    the second parameter fill on the large-motion samples; at 128x192 also every decoder level's flows."""
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow(ids_on_host=True).eval()
    model.load_state_dict(stress_state(model), strict=True)
    model = model.to("cuda:0")
    model.keep_levels = H == 128
    s = I.frame_pair_stress(seed, H=H, W=W, N=8192, dsec=dsec)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    out = model(with_reference_ids({k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()}, g, "cuda:0"))
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all() and f2.shape == (1, 2, H, W)
    e2, e3 = I.masked_epes(f2[0], f3[0], s)
    d2, d3 = abs(e2 - float(g["epe2d"])), abs(e3 - float(g["epe3d"]))
    want2, got2 = (g["flow_2d"], f2) if "flow_2d" in g else (g["flow_2d_s8"], f2[:, :, ::8, ::8])
    print(name, "EPE2D diff", d2, "EPE3D diff", d3, "mean |d flow_2d|", np.abs(got2 - want2).mean(), "mean |d flow_3d|", np.abs(f3 - g["flow_3d"]).mean())
    assert d2 < GOLDEN_EPE_TOL and d3 < GOLDEN_EPE_TOL
    assert np.abs(got2 - want2).mean() < 5e-3 and np.abs(f3 - g["flow_3d"]).mean() < 1e-3
    if H == 128:
        for i in range(5):
            w2, w3 = g["level%d_flow_2d" % i], g["level%d_flow_3d" % i]
            a2, a3 = np.abs(out["levels_2d"][i].cpu().numpy() - w2).mean(), np.abs(out["levels_3d"][i].cpu().numpy() - w3).mean()
            print("level %d: mean |d flow_2d| %.2e (|flow| %.2e), mean |d flow_3d| %.2e (|flow| %.2e)" % (i, a2, np.abs(w2).mean(), a3, np.abs(w3).mean()))
            assert a2 <= 1e-4 * max(1.0, np.abs(w2).mean()) and a3 <= 1e-4 * max(1.0, np.abs(w3).mean())


# ---- the DEFAULT path (round-4 review, weak #1): RPEFlow() as bench.py, evaluate() and a user build it -- IDS transform on the
# device (rpe_ids_forward, correctly rounded logarithm), no ``pcs_ids`` from the golden, no ``ids_on_host`` -- against the
# reference's CPU forward on every model golden, the three large-motion ones with the second parameter fill included.
DEFAULT_PATH = [("model_128x192", 1000, 128, 192, False, False), ("model_dsec_480x640", 2000, 480, 640, True, False),
                ("model_544x960", 3000, 544, 960, False, False)] + [(n, s, h, w, d, True) for n, s, h, w, d in STRESS]


@pytest.mark.gpu
@torch.no_grad()
@pytest.mark.parametrize("name,seed,H,W,dsec,stress", DEFAULT_PATH)
def test_default_device_ids_path_matches_reference_golden(golden_dir, name, seed, H, W, dsec, stress):
    """|EPE2D - reference|, |EPE3D - reference| < 1e-4 (north_star) for the path users run: the clouds FPS / KNN see are
    computed on the device.  Also reported: how many of the 2 x 3 x 8192 transformed coordinates differ from the reference's
    (its CPU log is off the correctly rounded value on a few in ten thousand) and whether the sampling order is the one the
    reference's clouds give."""
    from rpeflow_amd.csrc import furthest_point_sampling
    from rpeflow_amd.model import RPEFlow
    model = RPEFlow().eval()
    assert not model.ids_on_host
    model.load_state_dict((stress_state if stress else seeded_state)(model), strict=True)
    model = model.to("cuda:0")
    s = (I.frame_pair_stress if stress else I.frame_pair)(seed, H=H, W=W, N=8192, dsec=dsec)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    batch = {k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()}
    assert "pcs_ids" not in batch
    out = model(batch)
    f2, f3 = out["flow_2d"].cpu().numpy(), out["flow_3d"].cpu().numpy()
    assert np.isfinite(f2).all() and np.isfinite(f3).all() and f2.shape == (1, 2, H, W)
    if stress:  # EPEs counted as the evaluators count them (masked, NaN-free): what the stress goldens store
        e2, e3 = I.masked_epes(f2[0], f3[0], s)
        r2, r3 = float(g["epe2d"]), float(g["epe3d"])
    else:       # plain means over every pixel / point (what make_golden.py stored for these; 128x192 stores the flows themselves)
        e2, e3 = epe(f2, s["flow_2d"][None, :2]), epe(f3, s["flow_3d"][None, :3])
        r2, r3 = (float(g["epe2d"]), float(g["epe3d"])) if "epe2d" in g else (epe(g["flow_2d"], s["flow_2d"][None, :2]), epe(g["flow_3d"], s["flow_3d"][None, :3]))
    ref_clouds = torch.from_numpy(np.concatenate([g["pc1_ids"], g["pc2_ids"]])).to("cuda:0")
    mine = torch.cat(model._clouds(batch, *model._cameras(batch)))
    off = int((mine != ref_clouds).sum())
    same_order = torch.equal(model.sample_order(batch), furthest_point_sampling(ref_clouds.transpose(1, 2), 4096))
    print(name, "default path: EPE2D diff", abs(e2 - r2), "EPE3D diff", abs(e3 - r3), "| transformed coordinates off the reference's:", off,
          "| sampling order equals the reference's:", same_order)
    assert abs(e2 - r2) < GOLDEN_EPE_TOL and abs(e3 - r3) < GOLDEN_EPE_TOL
    assert off <= 4 and same_order


@pytest.mark.gpu
@torch.no_grad()
def test_evaluation_metrics_under_stress_match_the_reference(golden_dir):
    """The masked / NaN-carrying targets through the device accumulators (rpe_eval_accumulate) on the model's own output:
    EPE2D / EPE3D of evaluate.finalize equal the evaluators' masked means computed on the host, and both are within the
    bound of the reference forward's."""
    from rpeflow_amd import evaluate as E
    from rpeflow_amd.model import RPEFlow
    name, seed, H, W, dsec = STRESS[2]
    model = RPEFlow(ids_on_host=True).eval()
    model.load_state_dict(stress_state(model), strict=True)
    model = model.to("cuda:0")
    s = I.frame_pair_stress(seed, H=H, W=W, N=8192, dsec=dsec)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    batch = with_reference_ids({k: torch.from_numpy(v)[None].to("cuda:0") for k, v in s.items()}, g, "cuda:0")
    out = model(batch)
    m = E.finalize(E.accumulate(E.new_accumulator("cuda:0"), out, batch))
    e2, e3 = I.masked_epes(out["flow_2d"].cpu().numpy()[0], out["flow_3d"].cpu().numpy()[0], s)
    assert abs(m["EPE2D"] - e2) < 1e-5 * e2 and abs(m["EPE3D"] - e3) < 1e-5 * max(1.0, e3)
    assert abs(m["EPE2D"] - float(g["epe2d"])) < 2e-4 and abs(m["EPE3D"] - float(g["epe3d"])) < 2e-4
    assert m["counts"]["2d"] == float(((s["flow_2d"][2] > 0) & ~np.isnan(s["flow_2d"][0])).sum())


def test_config_behaves_like_a_mapping_with_attributes():
    """hasattr / deepcopy / pickling of the configuration object (the reference's omegaconf.DictConfig allows all three)."""
    import copy
    import pickle
    from rpeflow_amd.model import things_config
    cfg = things_config()
    assert not hasattr(cfg, "fusion") and hasattr(cfg, "pwc2d") and cfg.pwc2d.max_displacement == 4
    assert copy.deepcopy(cfg) == cfg and pickle.loads(pickle.dumps(cfg)) == cfg
    from rpeflow_amd.model import RPEFlow
    m = RPEFlow()
    assert len(copy.deepcopy(m).state_dict()) == len(m.state_dict())
