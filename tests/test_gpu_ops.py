"""GPU parity tests for the four operators of models/csrc/__init__.py:1.
Every test calls the HIP kernels through the C ABI (rpeflow_amd.csrc -> ctypes ->
librpeflow_hip.so) and checks against the CPU oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import cases as K
from tests import inputs as I
from tests.check import assert_bits_equal, assert_knn_tie_aware

pytestmark = pytest.mark.gpu

import rpeflow_amd.csrc as ops  # noqa: E402
from rpeflow_amd import _lib  # noqa: E402
from rpeflow_amd.csrc import wrapper as W  # noqa: E402

DEV = "cuda:0"


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_library_is_loaded_from_tree():
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib().rpe_abi_version() == _lib.ABI_VERSION


def test_mfma_4x4x1_layout_probe():
    """The correlation kernel assumes: A lane 4g+i, B lane 4g+j -> D register i, lane 4g+j.  (A development probe: only a
    library built with -DRPE_EXPERIMENTAL exports it; the correlation goldens test the same assumption end to end.)"""
    if not hasattr(_lib.lib(), "rpe_probe_mfma4x4"):
        pytest.skip("default build: rpe_probe_mfma4x4 is an RPE_EXPERIMENTAL entry point")
    out = torch.zeros(256, device=DEV)
    _lib.check(_lib.lib().rpe_probe_mfma4x4(out.data_ptr(), None), "probe")
    d = out.cpu().numpy().reshape(64, 4)
    lane = np.arange(64)
    g, j = lane // 4, lane % 4
    expect = (4 * g[:, None] + np.arange(4)[None, :]) * (100.0 * (4 * g + j))[:, None]
    assert np.array_equal(d, expect.astype(np.float32)), d[:8]


# ---------------------------------------------------------------- squared_distance
@pytest.mark.parametrize("name", list(K.SQDIST_CASES))
def test_squared_distance(golden_dir, name):
    a, b = K.sqdist_inputs(name)
    out = ops.squared_distance(dev(a), dev(b)).cpu().numpy()
    assert_bits_equal(out, O.squared_distance(a, b), name + " vs oracle")
    assert_bits_equal(out, G(golden_dir, name)["dist"], name + " vs reference golden")


# ---------------------------------------------------------------- k_nearest_neighbor
@pytest.mark.parametrize("name", list(K.KNN_CASES))
def test_knn_golden_cases(golden_dir, name):
    inp, qry, k = K.knn_inputs(name)
    idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k)  # default mode: position for position
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    default_idx = ops.k_nearest_neighbor(dev(inp), dev(qry), k).cpu().numpy()
    assert np.array_equal(default_idx, idx)
    sets_only = W.k_nearest_neighbor_ties(dev(inp), dev(qry), k, ties="set").cpu().numpy()
    assert np.array_equal(np.sort(sets_only, -1), np.sort(idx, -1)), name + ": mode 1 returns another neighbour set"
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
    assert np.array_equal(idx, oi), f"{name}: {(idx != oi).sum()} indices differ from the oracle"
    assert_bits_equal(dist, od, name + " distances vs oracle")
    g = G(golden_dir, name)
    assert np.array_equal(idx, g["idx"]), name + ": indices differ from the reference's torch.topk output"
    assert_bits_equal(dist, g["dist"], name + " distances vs reference golden")
    # public entry point, both layouts (wrapper.py:119-122)
    assert np.array_equal(default_idx, ops.k_nearest_neighbor(dev(inp), dev(qry), k).cpu().numpy())
    if inp.shape[1] > 3:
        cf = ops.k_nearest_neighbor(input_xyz=dev(inp.transpose(0, 2, 1)), query_xyz=dev(qry.transpose(0, 2, 1)), k=k)
        assert np.array_equal(cf.cpu().numpy(), default_idx)


@pytest.mark.parametrize("B,M,Q,D,k", [
    (1, 5, 1, 3, 5), (3, 64, 65, 3, 16), (2, 63, 129, 2, 1), (1, 65, 7, 3, 64), (2, 129, 33, 1, 3),
    (1, 1000, 3, 3, 33), (5, 200, 1, 2, 2), (1, 4096, 256, 3, 16), (4, 256, 256, 3, 16), (2, 512, 2160, 2, 1),
])
def test_knn_ragged_shapes(B, M, Q, D, k):
    r = I.rng(7000 + M + Q + k)
    inp, qry = I.ids_cloud(r, B, M, D), I.ids_cloud(r, B, Q, D)
    idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k)
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert_bits_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("B,M,Q,D,k", [
    # rpe_knn_multi sends k >= 2, M >= 1024, 64 k <= M, B * Q >= 16384 to the matrix kernel (knn_mfma_kernel: distances on
    # v_mfma_f32_16x16x4_f32, threshold + collect): ragged M (not a multiple of 256 / 64) and Q (not a multiple of 64 / 16),
    # every D, k up to 63
    (2, 1500, 8200, 3, 16), (1, 4100, 16390, 3, 3), (3, 1024, 5500, 2, 5), (2, 1030, 8192, 1, 2), (1, 2000, 16384, 3, 31),
    (1, 4100, 16384, 3, 63), (8, 8192, 2048, 3, 16),
    # k = 1 with as many queries: knn_mfma_nearest_kernel (running minimum + index per lane)
    (2, 1100, 9000, 2, 1), (1, 4100, 16500, 3, 1), (3, 1024, 5500, 1, 1), (4, 4096, 8640, 2, 1),
])
def test_knn_matrix_kernel_shapes(B, M, Q, D, k):
    r = I.rng(7300 + M + Q + k)
    inp, qry = I.ids_cloud(r, B, M, D), I.ids_cloud(r, B, Q, D)
    idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k)
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert_bits_equal(dist.cpu().numpy(), od)
    # channel-first views, as the model passes them
    cf = ops.k_nearest_neighbor(dev(inp.transpose(0, 2, 1)), dev(qry.transpose(0, 2, 1)), k)
    assert np.array_equal(cf.cpu().numpy(), oi)


@pytest.mark.parametrize("B,M,Q,D,k", [
    # the decoder's mid-size searches, which the size gate keeps on the insertion kernel: forced through the matrix kernel
    # (RPE_KNN_ALGO_MATRIX) and through the insertion kernel (RPE_KNN_ALGO_INSERT), both = the oracle, ties included
    (4, 2048, 2048, 3, 16), (4, 1024, 1024, 3, 16), (4, 1024, 2048, 3, 3), (2, 256, 300, 3, 3), (3, 300, 70, 2, 4), (1, 1024, 17, 3, 16),
    (2, 4096, 1000, 3, 16), (2, 260, 33, 1, 2),
])
def test_knn_forced_kernels_agree_on_mid_size_searches(B, M, Q, D, k):
    r = I.rng(7400 + M + Q + k)
    inp, qry = I.ids_cloud(r, B, M, D), I.ids_cloud(r, B, Q, D)
    inp[:, M // 2:M // 2 + 20] = inp[:, :20]  # duplicated points: equal distances in some rows
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
    for algo in ("matrix", "insert", "auto"):
        idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k, algo=algo)
        assert np.array_equal(idx.cpu().numpy(), oi), algo
        assert_bits_equal(dist.cpu().numpy(), od, algo)


def test_knn_duplicates_and_strided_views():
    r = I.rng(7100)
    base = I.unit_cloud(r, 2, 150, 3)
    inp = np.concatenate([base, base, base[:, :40]], axis=1)  # every point 2-3 times: ties everywhere
    qry = base[:, ::3]
    idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), 8)
    oi, od = O.k_nearest_neighbor(inp, qry, 8, return_dists=True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    # non-contiguous prefix view of a channel-first tensor, as build_pc_pyramid makes (pwc3d_core.py:25)
    cf = dev(inp.transpose(0, 2, 1))
    sub = cf[:, :, :200]
    assert not sub.is_contiguous()
    got = ops.k_nearest_neighbor(sub, dev(qry.transpose(0, 2, 1)), 8).cpu().numpy()
    assert np.array_equal(got, O.k_nearest_neighbor(inp[:, :200], qry, 8))


def test_knn_self_query_full_size_property():
    """BASELINE size (8192 x 8192, k=16): checked through size-independent properties --
    row 0 is the query itself wherever its self-distance is the unique minimum, distances
    ascend, and a random sample of rows equals the oracle."""
    r = I.rng(7200)
    pts = I.ids_cloud(r, 2, 8192, 3)
    idx, dist = W.k_nearest_neighbor_with_distances(dev(pts), dev(pts), 16)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert (np.diff(dist, axis=-1) >= 0).all()
    assert idx.min() >= 0 and idx.max() < 8192
    assert all(len(set(row)) == 16 for row in idx[0, :512])
    rows = r.choice(8192, 300, replace=False)
    oi, od = O.k_nearest_neighbor(pts, pts[:, rows], 16, return_dists=True)
    assert np.array_equal(idx[:, rows], oi)
    assert_bits_equal(dist[:, rows], od)


def test_knn_nearest_pixel_search_full_size():
    """The forward's largest 2-D search (every pixel of a 144 x 240 map against the 8192 projected points, k = 1:
    project_feat_with_nn_corr's nn_proj, RPEFlow_core.py:325-327) through the matrix-pipe kernel: a random sample of rows
    equals the oracle; every index is in range."""
    r = I.rng(7250)
    pts = I.ids_cloud(r, 2, 8192, 2)
    gx, gy = np.meshgrid(np.arange(240, dtype=np.float32) * 0.125 - 14.5, np.arange(144, dtype=np.float32) * 0.125 - 8.5)
    grid = np.broadcast_to(np.stack([gx.ravel(), gy.ravel()], -1)[None], (2, 34560, 2)).copy()
    idx, dist = W.k_nearest_neighbor_with_distances(dev(pts), dev(grid), 1)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert idx.shape == (2, 34560, 1) and idx.min() >= 0 and idx.max() < 8192
    rows = r.choice(34560, 400, replace=False)
    oi, od = O.k_nearest_neighbor(pts, grid[:, rows], 1, return_dists=True)
    assert np.array_equal(idx[:, rows], oi)
    assert_bits_equal(dist[:, rows], od)


@pytest.mark.parametrize("B,M,Q,D,k,kind", [(8, 8192, 4096, 3, 16, "ids"), (2, 1500, 8200, 3, 17, "lattice"), (1, 2048, 16384, 3, 16, "lattice"),
                                              (4, 1100, 4100, 2, 3, "lattice"), (4, 4096, 8192, 3, 3, "ids"), (2, 1030, 8192, 1, 2, "lattice"),
                                              (3, 16384, 5500, 3, 5, "ids"), (1, 16448, 16384, 3, 16, "ids"), (70, 1024, 4100, 3, 16, "ids")])
def test_knn_tied_rows_replayed_by_the_second_launch(B, M, Q, D, k, kind):
    """With workspace the matrix kernel hands its tied rows to knn_tie_replay_kernel (a grid of its own: the row's distances
    put into LDS by four waves, libstdc++'s heap code replayed on them by one); without, it redoes them itself at the end of the
    launch.  Same rows, same arithmetic, same restatement: indices and distances bit for bit, in every tie mode -- on lattice
    clouds (every row tied), IDS-range clouds (~1 % tied), a cloud at the LDS limit (16384 points), one beyond it and a launch
    with more workgroups than the replay kernel scans (both: the in-kernel path even with workspace)."""
    r = I.rng(9700 + M + k)
    if kind == "lattice":
        inp, qry = r.integers(0, 6, (B, M, D)).astype(np.float32), r.integers(0, 6, (B, Q, D)).astype(np.float32)
    else:
        inp, qry = I.ids_cloud(r, B, M, D), I.ids_cloud(r, B, Q, D)
    ti, tq = dev(inp), dev(qry)
    lib = _lib.lib()
    for ties in ("torch", "set", "index"):
        mode = _lib.KNN_TIES[ties]
        got_i, got_d = W.k_nearest_neighbor_with_distances(ti, tq, k, ties=ties)  # (allocates the workspace)
        idx = torch.empty((B, Q, k), dtype=torch.int64, device=DEV)
        dist = torch.empty((B, Q, k), dtype=torch.float32, device=DEV)
        _lib.check(lib.rpe_knn(ti.data_ptr(), *ti.stride(), tq.data_ptr(), *tq.stride(), B, M, Q, D, k, mode, idx.data_ptr(), dist.data_ptr(),
                               None, 0, None), "rpe_knn without workspace")
        torch.cuda.synchronize()
        assert torch.equal(got_i, idx), (ties, int((got_i != idx).sum()))
        assert torch.equal(got_d.view(torch.int32), dist.view(torch.int32)), ties
    need = lib.rpe_knn_workspace_bytes(B, M, Q, D, k, 3)
    assert (need > 0) == (M <= 16384 and B * ((Q + 63) // 64) <= 4096) and lib.rpe_knn_workspace_bytes(B, M, Q, D, k, 0) == 0
    rows = r.choice(Q, 40, replace=False)
    oi, od = O.k_nearest_neighbor(inp[:1], qry[:1, rows], k, return_dists=True)
    got_i, got_d = W.k_nearest_neighbor_with_distances(ti, tq, k)  # ... and both equal the reference's order (oracle, sampled rows)
    assert np.array_equal(got_i[:1, rows].cpu().numpy(), oi)
    assert_bits_equal(got_d[:1, rows].cpu().numpy(), od)


def test_knn_errors():
    x = torch.rand(1, 10, 3, device=DEV)
    with pytest.raises(RuntimeError):  # fallback's topk: k > M
        ops.k_nearest_neighbor(x, x, 11)
    with pytest.raises(RuntimeError):  # k_nearest_neighbor.cpp:9
        ops.k_nearest_neighbor(x.double(), x.double(), 2)


# ---------------------------------------------------------------- furthest_point_sampling
@pytest.mark.parametrize("name", list(K.FPS_CASES))
def test_fps_golden_cases(golden_dir, name):
    xyz, S = K.fps_inputs(name)
    got = ops.furthest_point_sampling(dev(xyz), S).cpu().numpy()
    assert got.dtype == np.int64
    assert np.array_equal(got, O.furthest_point_sampling(xyz, S)), name + " vs oracle"
    assert np.array_equal(got, G(golden_dir, name)["idx"]), name + " vs reference golden"


def _fps_algo(xyz, S, algo):
    """rpe_fps_algo with the kernel chosen explicitly (1 plain, 2 pruned)."""
    t = dev(xyz)
    idx = torch.empty((t.shape[0], S), dtype=torch.int64, device=DEV)
    rc = _lib.lib().rpe_fps_algo(t.data_ptr(), *t.stride(), t.shape[0], t.shape[1], S, idx.data_ptr(), algo, None)
    if rc == -2:
        return None  # RPE_EUNSUPPORTED: the pruned kernel needs 1024 < N <= 16384
    _lib.check(rc, "rpe_fps_algo")
    torch.cuda.synchronize()
    return idx.cpu().numpy()


@pytest.mark.parametrize("algo", [1, 2])
def test_fps_both_kernels_agree(golden_dir, algo):
    """The plain kernel and the cluster-skipping kernel return identical samples, whichever one rpe_fps would pick."""
    for name in K.FPS_CASES:
        xyz, S = K.fps_inputs(name)
        got = _fps_algo(xyz, S, algo)
        if got is not None:
            assert np.array_equal(got, G(golden_dir, name)["idx"]), (name, algo)
    xyz = I.ids_cloud(I.rng(8300), 2, 5000)  # N not a multiple of 1024: padded lanes must never win
    assert np.array_equal(_fps_algo(xyz, 4999, algo), O.furthest_point_sampling(xyz, 4999))
    xyz = I.ids_cloud(I.rng(8301), 1, 700)   # N <= 1024: plain only
    got = _fps_algo(xyz, 300, algo)
    assert (got is None) if algo != 1 else np.array_equal(got, O.furthest_point_sampling(xyz, 300))


@pytest.mark.parametrize("kind", ["duplicates", "lattice", "clusters", "odd_and_even_lengths"])
def test_fps_pruned_kernel_on_tie_heavy_clouds(kind):
    """The cluster-skipping kernel where equal running distances abound: clouds with every point twice, integer lattices
    (hundreds of equal distances), two far-apart clusters, and sample counts of both parities."""
    r = I.rng(8400)
    if kind == "duplicates":
        base = I.ids_cloud(r, 2, 1500)
        clouds = [(np.concatenate([base, base], axis=1), 2999)]
    elif kind == "lattice":
        clouds = [(r.integers(0, 12, (2, 4000, 3)).astype(np.float32), 3000)]
    elif kind == "clusters":
        a = I.unit_cloud(r, 2, 2000)
        clouds = [(np.concatenate([a, a[:, ::-1] + np.float32(50.0)], axis=1), 3999)]
    else:
        xyz = I.ids_cloud(r, 1, 4100)
        clouds = [(xyz, S) for S in (1, 2, 3, 1000, 1001, 4099)]
    for xyz, S in clouds:
        want = O.furthest_point_sampling(xyz, S)
        for algo in (1, 2):
            assert np.array_equal(_fps_algo(xyz, S, algo), want), (kind, S, algo)


@pytest.mark.parametrize("B,N,S", [(1, 2, 1), (2, 65, 64), (3, 1023, 100), (1, 1025, 1024), (2, 3000, 700), (1, 9000, 50), (1, 20000, 40)])
def test_fps_ragged_shapes(B, N, S):
    xyz = I.ids_cloud(I.rng(8000 + N), B, N)
    got = ops.furthest_point_sampling(dev(xyz), S).cpu().numpy()
    assert np.array_equal(got, O.furthest_point_sampling(xyz, S))


def test_fps_transposed_view_as_the_model_passes_it():
    """build_pc_pyramid hands over pc.transpose(1, 2) of a channel-first cloud (pwc3d_core.py:13)."""
    xyz = I.ids_cloud(I.rng(8100), 4, 8192)
    cf = dev(xyz.transpose(0, 2, 1))
    got = ops.furthest_point_sampling(cf.transpose(1, 2), 4096).cpu().numpy()
    assert np.array_equal(got, O.furthest_point_sampling(xyz, 4096))
    # property at full size: samples are distinct, and every prefix is the FPS of that length
    assert all(len(set(row)) == 4096 for row in got)


# ---------------------------------------------------------------- correlation2d
DMA_ALGOS = (7, 8)  # LDS-DMA ring variants of the MFMA kernel (eight waves, two rows / one row a wave): need W % 4 == 0 and C % 2 / C % 4 == 0


def corr_algos(C, Wd, md):
    algos = [1]
    if md <= 4:
        algos.append(3)  # the small-map kernel (any shape, md <= 4)
    if md == 4:
        algos.append(2)
        if Wd % 4 == 0:
            algos += [a for a in DMA_ALGOS if C % (2 if a == 7 else 4) == 0]
    return algos


@pytest.mark.parametrize("name", list(K.CORR_CASES))
def test_correlation_golden_cases(golden_dir, name):
    a, b, md = K.corr_inputs(name)
    ref = G(golden_dir, name)["out"]
    for algo in corr_algos(a.shape[1], a.shape[3], md):
        out = W._correlation2d_algo(dev(a), dev(b), md, algo).cpu().numpy()
        assert out.shape == ref.shape
        # correlation_test.cpp:82-83: mean |diff| < 1e-6; plus a worst-element bound
        assert np.abs(out - ref).mean() < 1e-6, (name, algo)
        assert np.abs(out - ref).max() < 5e-6, (name, algo)
        assert np.abs(out - O.correlation2d(a, b, md)).max() < 5e-6, (name, algo)
    out = ops.correlation2d(dev(a), dev(b), md).cpu().numpy()
    assert np.abs(out - ref).max() < 5e-6


@pytest.mark.parametrize("B,C,H,Wd", [(1, 3, 5, 7), (2, 33, 17, 65), (1, 16, 9, 15), (4, 192, 9, 15), (1, 64, 70, 130), (2, 5, 8, 64),
                                      (1, 32, 10, 240), (2, 6, 3, 129), (1, 2, 2, 256), (2, 12, 40, 68), (1, 8, 33, 132),
                                      (3, 4, 16, 64), (1, 20, 7, 4)])
def test_correlation_ragged_shapes(B, C, H, Wd):
    r = I.rng(9000 + C + H + Wd)
    a, b = I.feature_map(r, B, C, H, Wd), I.feature_map(r, B, C, H, Wd)
    ref = O.correlation2d(a, b, 4)
    for algo in corr_algos(C, Wd, 4):
        got = W._correlation2d_algo(dev(a), dev(b), 4, algo).cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() < 5e-6, algo


@pytest.mark.parametrize("name", list(K.CORR_CASES))
def test_correlation_backward_matches_reference_autograd(golden_dir, name):
    """CorrelationFunction.backward (wrapper.py:27-37) against the gradients autograd derives from the reference's
    differentiable fallback."""
    a, b, md = K.corr_inputs(name)
    ta, tb = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    out = ops.correlation2d(ta, tb, md)
    assert out.requires_grad
    assert np.abs(out.detach().cpu().numpy() - G(golden_dir, name)["out"]).max() < 5e-6
    out.backward(dev(K.corr_grad_output(name)))
    ref = G(golden_dir, name + "_grad")
    assert np.abs(ta.grad.cpu().numpy() - ref["grad1"]).max() < 5e-6
    assert np.abs(tb.grad.cpu().numpy() - ref["grad2"]).max() < 5e-6


@pytest.mark.parametrize("B,C,H,Wd,md", [(1, 3, 5, 7, 4), (2, 33, 17, 65, 4), (1, 9, 40, 18, 2), (2, 8, 16, 16, 1), (1, 20, 33, 47, 3), (1, 1, 1, 1, 4)])
def test_correlation_backward_ragged_shapes(B, C, H, Wd, md):
    r = I.rng(9200 + C + H + Wd)
    a, b = I.feature_map(r, B, C, H, Wd), I.feature_map(r, B, C, H, Wd)
    n = 2 * md + 1
    go = r.standard_normal((B, n * n, H, Wd), dtype=np.float32)
    ta, tb = dev(a).requires_grad_(True), dev(b)  # only one input needs a gradient here
    ops.correlation2d(ta, tb, md).backward(dev(go))
    g1, g2 = O.correlation2d_backward(go, a, b, md)
    assert np.abs(ta.grad.cpu().numpy() - g1).max() < 5e-6 and tb.grad is None
    ta, tb = dev(a), dev(b).requires_grad_(True)
    ops.correlation2d(ta, tb, md).backward(dev(go))
    assert np.abs(tb.grad.cpu().numpy() - g2).max() < 5e-6


def test_correlation_fused_leaky_relu():
    r = I.rng(9100)
    a, b = I.feature_map(r, 2, 32, 18, 32), I.feature_map(r, 2, 32, 18, 32)
    ref = torch.nn.functional.leaky_relu(torch.from_numpy(O.correlation2d(a, b, 4)), 0.1).numpy()  # RPEFlow_core.py:362
    for algo in corr_algos(32, 32, 4):
        got = W._correlation2d_algo(dev(a), dev(b), 4, algo, leaky_slope=0.1).cpu().numpy()
        assert np.abs(got - ref).max() < 5e-6


def test_correlation_full_size_properties():
    """BASELINE microbench size 1x256x544x960: linearity in in1, zero borders, and the centre
    plane equals the per-pixel channel mean of in1*in2 computed by torch on the GPU."""
    torch.manual_seed(0)
    a = torch.randn(1, 256, 544, 960, device=DEV)
    b = torch.randn(1, 256, 544, 960, device=DEV)
    out = ops.correlation2d(a, b, 4)
    assert out.shape == (1, 81, 544, 960)
    centre = (a * b).mean(1)
    assert (out[:, 40] - centre).abs().max().item() < 2e-5
    # plane (dy=-4,dx=-4): rows 0..3 and cols 0..3 read outside the image -> exactly 0
    assert out[:, 0, :4].abs().max().item() == 0.0 and out[:, 0, :, :4].abs().max().item() == 0.0
    assert out[:, 80, -4:].abs().max().item() == 0.0 and out[:, 80, :, -4:].abs().max().item() == 0.0
    # shifted plane against torch: dy=+2, dx=-3 -> plane (2+4)*9 + (-3+4)
    sh = (a[:, :, :-2, 3:] * b[:, :, 2:, :-3]).mean(1)
    assert (out[:, 55, :-2, 3:] - sh).abs().max().item() < 2e-5
    # all 81 planes (BASELINE config 2 in full) against the reference's naive recipe restated in fp32 PyTorch on the GPU:
    # slices of the zero-padded second input, mean over channels (wrapper.py:56-65, correlation_test.cpp:27-42)
    padded = torch.nn.functional.pad(b, (4, 4, 4, 4))
    worst = 0.0
    for i in range(9):
        for j in range(9):
            plane = (a * padded[:, :, i:i + 544, j:j + 960]).mean(1)
            worst = max(worst, (out[:, i * 9 + j] - plane).abs().max().item())
    assert worst < 2e-5, worst
    del padded, plane
    out2 = ops.correlation2d(2.0 * a, b, 4)
    assert torch.equal(out2, 2.0 * out)
    # every kernel agrees on a crop, and the DMA-ring variants agree with the first MFMA kernel at full size
    ca, cb = a[:, :, :64, :128].contiguous(), b[:, :, :64, :128].contiguous()
    d = W._correlation2d_algo(ca, cb, 4, 1)
    for algo in (2,) + DMA_ALGOS:
        assert (d - W._correlation2d_algo(ca, cb, 4, algo)).abs().max().item() < 5e-6, algo
    full = W._correlation2d_algo(a, b, 4, 2)
    for algo in DMA_ALGOS:
        assert (full - W._correlation2d_algo(a, b, 4, algo)).abs().max().item() < 5e-6, algo


def test_knn_multi_equals_separate_calls():
    """rpe_knn_multi: the pyramid's five searches in one launch give exactly the five separate results."""
    r = I.rng(8800)
    cloud = I.ids_cloud(r, 3, 2048)
    levels = [dev(cloud[:, :n].copy()) for n in (2048, 1024, 512, 100, 37)]
    pairs = [(levels[i], levels[i + 1]) for i in range(4)]
    got = W.k_nearest_neighbor_multi(pairs, 16)
    for (inp, qry), g in zip(pairs, got):
        assert torch.equal(g, ops.k_nearest_neighbor(inp, qry, 16))
    # a launch whose jobs split between the matrix kernel (M >= 1024, B * Q >= 16384) and the insertion kernel
    big = I.ids_cloud(r, 8, 4096)
    lv = [dev(big[:, :n].copy()) for n in (4096, 2048, 1024, 300)]
    mixed = [(lv[0], lv[1]), (lv[1], lv[2]), (lv[2], lv[3]), (lv[0], lv[0])]
    for (inp, qry), g in zip(mixed, W.k_nearest_neighbor_multi(mixed, 16)):
        assert torch.equal(g, ops.k_nearest_neighbor(inp, qry, 16))
        assert np.array_equal(g[:2, :64].cpu().numpy(), O.k_nearest_neighbor(inp[:2].cpu().numpy(), qry[:2, :64].cpu().numpy(), 16))
    cf = [(a.transpose(1, 2).contiguous(), b.transpose(1, 2).contiguous()) for a, b in pairs]  # channel-first layout, k = 1 kernel
    for (inp, qry), g in zip(cf, W.k_nearest_neighbor_multi(cf, 1)):
        assert torch.equal(g, ops.k_nearest_neighbor(inp, qry, 1))


@pytest.mark.parametrize("B,M,Q,D,k", [(2, 700, 300, 3, 16), (1, 2048, 1024, 3, 16), (2, 130, 64, 3, 3), (1, 500, 777, 2, 3), (3, 40, 33, 3, 1),
                                       (1, 1500, 200, 3, 17), (2, 1023, 128, 3, 16), (2, 1024, 128, 3, 16), (1, 64, 64, 2, 1), (1, 63, 40, 2, 1),
                                       (2, 300, 300, 3, 2), (1, 5000, 96, 3, 5),
                                       # k - 1 > 16: topk's std::sort is an introsort there; 32 is the reference extension's cap
                                       # (k_nearest_neighbor_kernel.cu:24,68); both regimes (k * 64 <= M or not)
                                       (2, 700, 200, 3, 32), (1, 2048, 300, 3, 32), (1, 3000, 100, 3, 24), (1, 1500, 64, 2, 20),
                                       (1, 4000, 50, 3, 63), (1, 5000, 40, 3, 40), (1, 40, 40, 3, 40),
                                       # enough queries for the matrix kernel: on a lattice nearly every lane list overflows (serial
                                       # sweep per query) or ties (libstdc++ restatement); M ragged and a multiple of the 256-point chunk
                                       (2, 1500, 8200, 3, 17), (1, 2048, 16384, 3, 16), (4, 1100, 4100, 2, 3), (2, 1500, 8200, 2, 1),
                                       # nth_element regime, every query tied: the wave-wide emulation of libstdc++'s Hoare partition
                                       # (row lengths around the 64-lane rows and the three-element end game)
                                       (2, 511, 600, 3, 7), (1, 1000, 500, 3, 15), (3, 65, 200, 1, 2), (1, 777, 300, 2, 12), (2, 100, 400, 3, 16),
                                       (1, 129, 300, 3, 2), (2, 4, 50, 3, 3), (1, 67, 90, 2, 17)])
def test_knn_equal_distances_follow_torch_topk(B, M, Q, D, k):
    """Points on a coarse integer lattice: most distances tie.  Indices AND their order must be what the reference's
    matmul + torch.topk gives on the CPU, in both of topk's regimes (k * 64 <= M: partial_sort; else nth_element + sort)."""
    r = I.rng(9300 + M + k)
    inp = r.integers(0, 6, (B, M, D)).astype(np.float32)
    qry = r.integers(0, 6, (B, Q, D)).astype(np.float32)
    ti, td = torch.from_numpy(inp), torch.from_numpy(qry)
    dmat = -2 * torch.matmul(td, ti.permute(0, 2, 1))
    dmat += torch.sum(td ** 2, -1).view(B, Q, 1)
    dmat += torch.sum(ti ** 2, -1).view(B, 1, M)
    ref_idx = dmat.topk(k, dim=2, largest=False).indices.numpy()          # wrapper.py:115-117 on the CPU
    assert np.array_equal(O.k_nearest_neighbor(inp, qry, k), ref_idx)      # the oracle's restatement of it
    got = ops.k_nearest_neighbor(dev(inp), dev(qry), k).cpu().numpy()  # default: position for position, order of equal distances included
    assert np.array_equal(got, ref_idx), f"{(got != ref_idx).sum()} of {got.size} indices differ"
    got = W.k_nearest_neighbor_ties(dev(inp), dev(qry), k, ties="set").cpu().numpy()  # cheaper: the reference's neighbour SET, always
    assert np.array_equal(np.sort(got, -1), np.sort(ref_idx, -1)), "ties='set' returns another neighbour set"
    # the plain lowest-index rule is still available and still a valid neighbour set
    low, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k, ties="index")
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True, ties="index")
    assert np.array_equal(low.cpu().numpy(), oi)


def test_shader_clock_reads_every_compute_unit():
    """rpe_clock_stamp_all through rpeflow_amd.runtime.ShaderClock (bench.py's roofline_corr carries its reading): two stamps
    around a stretch of full-chip work reach (nearly) every compute unit, each unit's cycle counter is compared with itself
    only, and the median is a plausible engine clock -- or None, never an implausible number."""
    from rpeflow_amd import runtime
    a = torch.randn(1, 64, 144, 240, device="cuda:0")
    b = torch.randn(1, 64, 144, 240, device="cuda:0")
    for _ in range(20):
        ops.correlation2d(a, b, 4)
    clock = runtime.ShaderClock("cuda:0")
    with clock:
        for _ in range(200):
            ops.correlation2d(a, b, 4)
    torch.cuda.synchronize()
    assert clock.units() >= 128
    cycles, ticks, khz = clock.raw()
    assert cycles > 0 and ticks > 0 and khz > 0
    mhz = clock.mhz()
    assert mhz is None or runtime.SCLK_PLAUSIBLE_MHZ[0] <= mhz <= runtime.SCLK_PLAUSIBLE_MHZ[1]
    assert 1 <= len(clock.mhz_per_xcd()) <= 8
    print("engine clock over the loop: %s MHz, per XCD %s, %d compute units read" % (mhz, clock.mhz_per_xcd(), clock.units()))


@pytest.mark.parametrize("B,M,Q,k,D", [(4, 8192, 4096, 16, 3), (4, 2048, 2048, 16, 3), (4, 4096, 8192, 3, 3), (4, 300, 200, 1, 3), (8, 4096, 34560, 1, 2), (2, 100, 64, 63, 3)])
@pytest.mark.parametrize("where", ["cloud", "query", "one sample"])
def test_nan_coordinates_never_produce_an_index_outside_the_cloud(B, M, Q, k, D, where):
    """Every kernel family of k_nearest_neighbor (insertion, matrix + tie replay, nearest, binned 2-D) and furthest_point_sampling
    on clouds with NaN coordinates: whatever order NaN distances end up in, the indices stay inside [0, M) -- the gathers that consume
    them do not check."""
    g = torch.Generator().manual_seed(7)
    x, q = torch.randn(B, M, D, generator=g).to(DEV), torch.randn(B, Q, D, generator=g).to(DEV)
    if where == "cloud":
        x[:, ::7] = float("nan")
    elif where == "query":
        q[:, ::5, 0] = float("nan")
    else:
        x[1], q[1] = float("nan"), float("nan")
    idx = W.k_nearest_neighbor(x, q, k)
    assert idx.shape == (B, Q, k) and int(idx.min()) >= 0 and int(idx.max()) < M
    if D == 3 and M > 256:
        fps = W.furthest_point_sampling(x, M // 4)
        assert int(fps.min()) >= 0 and int(fps.max()) < M
