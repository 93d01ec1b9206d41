"""k_nearest_neighbor on spatially ordered sets (csrc/knn_grid.h: Morton-cell order, per-step boxes, steps that cannot hold a
neighbour skipped) must return EXACTLY what the sweeping kernels and the reference's matmul + torch.topk return
(models/csrc/wrapper.py:106-127): indices position for position, distance bit patterns, equal distances included."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import cases as K
from tests import inputs as I
from tests.test_gpu_ops import G, assert_bits_equal, dev

pytestmark = pytest.mark.gpu

from rpeflow_amd.csrc import wrapper as W  # noqa: E402
import rpeflow_amd.csrc as ops  # noqa: E402

ELIGIBLE = [n for n, c in K.KNN_CASES.items() if 2 <= c[5] <= 30 and c[2] >= 64 * c[5] and c[2] >= 256]


def both(inp, qry, k, **kw):
    idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k, algo="grid", **kw)
    return idx.cpu().numpy(), dist.cpu().numpy()


def test_there_are_golden_cases_for_this_kernel():
    assert len(ELIGIBLE) >= 5


@pytest.mark.parametrize("name", ELIGIBLE)
def test_grid_kernel_on_the_reference_goldens(golden_dir, name):
    inp, qry, k = K.knn_inputs(name)
    idx, dist = both(inp, qry, k)
    g = G(golden_dir, name)
    assert np.array_equal(idx, g["idx"]), "%s: %d indices differ from the reference's torch.topk output" % (name, (idx != g["idx"]).sum())
    assert_bits_equal(dist, g["dist"], name + " distances vs reference golden")
    sets_only = W.k_nearest_neighbor_ties(dev(inp), dev(qry), k, ties="set", algo="grid").cpu().numpy()
    assert np.array_equal(np.sort(sets_only, -1), np.sort(idx, -1))
    lowest = W.k_nearest_neighbor_ties(dev(inp), dev(qry), k, ties="index", algo="grid").cpu().numpy()
    assert np.array_equal(lowest, O.k_nearest_neighbor(inp, qry, k, ties="index"))


@pytest.mark.parametrize("B,M,Q,D,k", [
    (2, 1500, 8200, 3, 16), (1, 4100, 16390, 3, 3), (3, 1024, 5500, 2, 5), (2, 1030, 8192, 1, 2), (1, 2000, 1000, 3, 30),
    (8, 8192, 2048, 3, 16), (1, 256, 100, 3, 4), (2, 300, 17, 2, 2), (1, 16384, 700, 3, 16), (4, 4096, 4096, 3, 3),
    (1, 577, 1, 3, 9), (2, 8192, 63, 2, 8),
])
def test_grid_kernel_ragged_shapes_against_the_oracle(B, M, Q, D, k):
    """M not a multiple of 64, Q not a multiple of 16 / 64, every D, k from 2 to 30, clouds up to the kernel's 16384 points."""
    r = I.rng(7600 + M + Q + k)
    inp, qry = I.ids_cloud(r, B, M, D), I.ids_cloud(r, B, Q, D)
    idx, dist = both(inp, qry, k)
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
    assert np.array_equal(idx, oi), "%d of %d indices differ" % ((idx != oi).sum(), oi.size)
    assert_bits_equal(dist, od)
    cf = W.k_nearest_neighbor_ties(dev(inp.transpose(0, 2, 1)), dev(qry.transpose(0, 2, 1)), k, algo="grid")  # channel-first views
    assert np.array_equal(cf.cpu().numpy(), oi)


@pytest.mark.parametrize("kind", ["lattice", "duplicates", "clustered", "line", "far_queries", "one_point_many_times"])
def test_grid_kernel_degenerate_clouds(kind):
    """Equal distances everywhere (lane lists fill up: serial fallback, tie queue), cells with hundreds of points, points
    on a line (degenerate bounding box), queries far outside the cloud's box, a cloud that is one point."""
    r = I.rng(7700)
    B, M, Q, k = 2, 2500, 300, 8
    if kind == "lattice":
        inp, qry = r.integers(0, 7, (B, M, 3)).astype(np.float32), r.integers(0, 7, (B, Q, 3)).astype(np.float32)
    elif kind == "duplicates":
        base = I.unit_cloud(r, B, 500, 3)
        inp, qry = np.concatenate([base] * 5, 1), base[:, ::2].copy()
    elif kind == "clustered":
        inp = I.ids_cloud(r, B, M, 3)
        inp[:, : M - 40] = inp[:, :1] + r.standard_normal((B, M - 40, 3)).astype(np.float32) * np.float32(1e-3)  # all but 40 in one cell
        qry = np.concatenate([inp[:, :150] + np.float32(1e-4), I.ids_cloud(r, B, 150, 3)], 1)
    elif kind == "line":
        t = r.random((B, M, 1), dtype=np.float32)
        inp = np.concatenate([t * 30 - 15, np.full_like(t, 2.5), np.full_like(t, 40.0)], -1)
        qry = I.ids_cloud(r, B, Q, 3)
    elif kind == "far_queries":
        inp = I.unit_cloud(r, B, M, 3)
        qry = I.unit_cloud(r, B, Q, 3) * np.float32(5.0) + np.float32(100.0)
    else:
        inp = np.broadcast_to(np.float32([3.0, -2.0, 50.0]), (B, M, 3)).copy()
        qry = I.ids_cloud(r, B, Q, 3)
    idx, dist = both(inp, qry, k)
    oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
    assert np.array_equal(idx, oi), "%s: %d of %d indices differ" % (kind, (idx != oi).sum(), oi.size)
    assert_bits_equal(dist, od)


def test_grid_build_is_a_permutation_with_boxes():
    """rpe_knn_grid_build: perm is a permutation of the indices, sorted = points[perm] with |p|^2 as the oracle rounds it,
    padding marked +inf, every step's box holds its points, consecutive points are spatially close (cells in Morton order)."""
    r = I.rng(7800)
    B, N = 3, 5000
    pts = I.ids_cloud(r, B, N, 3)
    gs = W.GridSet(dev(pts))
    perm, srt, boxes = gs.perm.cpu().numpy(), gs.sorted.cpu().numpy(), gs.boxes.cpu().numpy()
    npad = (N + 63) // 64 * 64
    assert srt.shape == (B, npad // 64, 4, 64) and perm.shape == (B, npad) and boxes.shape == (B, npad // 64 + 1, 8)
    rot = lambda row, g: np.roll(row, -16 * g, axis=-1)  # coordinate row g is stored rotated by 16 g: undo
    flat = np.stack([rot(srt[:, :, 0], 0), rot(srt[:, :, 1], 1), rot(srt[:, :, 2], 2), srt[:, :, 3]], 1).reshape(B, 4, npad)
    for b in range(B):
        assert np.array_equal(np.sort(perm[b, :N]), np.arange(N))
        assert np.array_equal(flat[b, :3, :N], pts[b, perm[b, :N]].T)
        sq = pts[b, perm[b, :N]]
        pp = (sq[:, 0] * sq[:, 0] + sq[:, 1] * sq[:, 1]) + sq[:, 2] * sq[:, 2]
        assert np.array_equal(flat[b, 3, :N].view(np.uint32), pp.astype(np.float32).view(np.uint32))
        assert np.isinf(flat[b, 3, N:]).all() and (flat[b, :3, N:] == 0).all()
        for s in range(npad // 64):
            chunk = flat[b, :3, 64 * s:min(64 * s + 64, N)]
            assert (boxes[b, s, :3] <= chunk.min(1)).all() and (boxes[b, s, 3:6] >= chunk.max(1)).all()
            assert boxes[b, s, 6] >= flat[b, 3, 64 * s:min(64 * s + 64, N)].max()
        assert np.array_equal(boxes[b, -1, :3], pts[b].min(0)) and np.array_equal(boxes[b, -1, 3:6], pts[b].max(0))
        spread_sorted = np.prod(boxes[b, :-1, 3:6] - boxes[b, :-1, :3], axis=1).mean()
        spread_input = np.mean([np.prod(np.ptp(pts[b, 64 * s:64 * s + 64], axis=0)) for s in range(N // 64)])
        assert spread_sorted < 0.1 * spread_input  # a step's box is a small part of the cloud's
    again = W.GridSet(dev(pts))
    assert torch.equal(again.perm, gs.perm)  # cells are put back into index order: the layout does not depend on the scheduling


def test_built_sets_are_reusable_and_auto_dispatch_agrees():
    """One GridSet serves as the cloud of several searches and as the queries of a self search; prefix views of a channel-first
    tensor (build_pc_pyramid, pwc3d_core.py:25); the default dispatch returns the same indices as both explicit paths."""
    r = I.rng(7900)
    B, M = 4, 8192
    cloud = dev(np.ascontiguousarray(I.ids_cloud(r, B, M, 3).transpose(0, 2, 1)))  # [B,3,M] as the model holds it
    sub = cloud[:, :, :4096]
    assert not sub.is_contiguous()
    sweep = W.k_nearest_neighbor_ties(cloud, sub, 16, algo="sweep")
    grid = W.k_nearest_neighbor_ties(cloud, sub, 16, algo="grid")
    auto = ops.k_nearest_neighbor(cloud, sub, 16)
    assert torch.equal(sweep, grid) and torch.equal(sweep, auto)
    gs = W.GridSet(cloud.transpose(1, 2))
    gq = W.GridSet(sub.transpose(1, 2))
    assert torch.equal(W.k_nearest_neighbor_ties(cloud, sub, 16, input_grid=gs, query_grid=gq), sweep)
    self16 = W.k_nearest_neighbor_ties(cloud, cloud, 16, input_grid=gs, query_grid=gs)
    assert torch.equal(self16, W.k_nearest_neighbor_ties(cloud, cloud, 16, algo="sweep"))
    assert torch.equal(W.k_nearest_neighbor_ties(sub, cloud, 3, input_grid=gq, query_grid=gs), W.k_nearest_neighbor_ties(sub, cloud, 3, algo="sweep"))


def test_grid_kernel_full_size_against_the_oracle_rows():
    """BASELINE size: 8 x (8192 -> 4096), k = 16 (the forward's largest search, both frames of a batch of 4): a random sample
    of rows against the oracle, everything against the sweeping kernel."""
    r = I.rng(7950)
    pts = I.ids_cloud(r, 8, 8192, 3)
    qry = np.ascontiguousarray(pts[:, :4096])
    idx, dist = both(pts, qry, 16)
    assert (np.diff(dist, axis=-1) >= 0).all() and idx.min() >= 0 and idx.max() < 8192
    rows = r.choice(4096, 200, replace=False)
    oi, od = O.k_nearest_neighbor(pts, qry[:, rows], 16, return_dists=True)
    assert np.array_equal(idx[:, rows], oi)
    assert_bits_equal(dist[:, rows], od)
    sweep = W.k_nearest_neighbor_ties(dev(pts), dev(qry), 16, algo="sweep").cpu().numpy()
    assert np.array_equal(idx, sweep)


def test_grid_kernel_refuses_what_it_does_not_take():
    x = torch.rand(1, 100, 3, device="cuda:0")
    with pytest.raises(RuntimeError):
        W.k_nearest_neighbor_ties(x, x, 16, algo="grid")   # 64 k > M
    y = torch.rand(1, 4096, 3, device="cuda:0")
    with pytest.raises(RuntimeError):
        W.k_nearest_neighbor_ties(y, y, 1, algo="grid")    # k = 1: the nearest-point kernels
    with pytest.raises(RuntimeError):
        W.k_nearest_neighbor_ties(y, y, 40, algo="grid")   # k + 1 > 32
