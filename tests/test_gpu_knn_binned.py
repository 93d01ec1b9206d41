"""k_nearest_neighbor, k = 1, D = 2 on the binned cloud (csrc/knn_binned.hip, reached through rpe_knn's workspace) -- what the model's
nearest-projected-point searches run through (RPEFlow_core.py:327-330 -> wrapper.py:106-127).  The kernel prunes by cell
geometry, so every test asks for EXACT agreement -- indices and distance bit patterns -- with the sweeping kernels (every
pair evaluated), the oracle and the reference goldens, on raster and non-raster queries and on clouds built to break a pruning
rule: points far outside the query domain, non-finite coordinates, exact ties, empty regions, degenerate domains."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import cases as K
from tests import inputs as I
from tests.check import assert_bits_equal

pytestmark = pytest.mark.gpu

import rpeflow_amd.csrc as ops  # noqa: E402
from rpeflow_amd.csrc import wrapper as W  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def both(pts, qry):
    """(binned, sweep) results: indices and distances, as numpy."""
    out = []
    for algo in ("binned", "sweep"):
        i, d = W.k_nearest_neighbor_with_distances(pts, qry, 1, algo=algo)
        out.append((i.cpu().numpy(), d.cpu().numpy()))
    return out


def assert_same(pts, qry, what=""):
    (bi, bd), (si, sd) = both(pts, qry)
    assert np.array_equal(bi, si), f"{what}: {(bi != si).sum()} of {bi.size} indices differ from the sweep"
    assert np.array_equal(bd.view(np.uint32), sd.view(np.uint32)), f"{what}: distances differ from the sweep"
    return bi, bd


def raster(B, H, W, scale=1.0, x0=0.0, y0=0.0):
    g = I.pixel_grid(B, H, W)
    g[..., 0] = g[..., 0] * np.float32(scale) + np.float32(x0)
    g[..., 1] = g[..., 1] * np.float32(scale) + np.float32(y0)
    return g


@pytest.mark.parametrize("name", [n for n in K.KNN_CASES if "k1_pix" in n])
def test_reference_goldens(golden_dir, name):
    inp, qry, k = K.knn_inputs(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k, algo="binned")
    assert np.array_equal(idx.cpu().numpy(), g["idx"]), name
    assert_bits_equal(dist.cpu().numpy(), g["dist"], name)


@pytest.mark.parametrize("B,N,H,Wd", [(8, 4096, 144, 240), (8, 2048, 72, 120), (8, 1024, 36, 60), (8, 512, 18, 30), (8, 256, 9, 15),
                                      (6, 4096, 128, 160), (2, 8192, 144, 240), (3, 777, 37, 53), (1, 64, 5, 7)])
def test_pyramid_level_searches_equal_the_sweep_and_the_oracle(B, N, H, Wd):
    """The model's shapes (FlyingThings3D and DSEC pyramids, both frames batched), points a little beyond the image."""
    r = I.rng(4100 + N + H)
    pts, qry = I.pixel_cloud(r, B, N, H, Wd), raster(B, H, Wd)
    idx, dist = assert_same(dev(pts), dev(qry), "level")
    rows = r.choice(H * Wd, min(H * Wd, 200), replace=False)
    oi, od = O.k_nearest_neighbor(pts[:2], qry[:2][:, rows], 1, return_dists=True)
    assert np.array_equal(idx[:2][:, rows], oi)
    assert_bits_equal(dist[:2][:, rows], od)
    # channel-first views (what the model passes: [B, 2, N] and the mesh grid as [B, 2, H W]) read through strides
    cf = W.k_nearest_neighbor_ties(dev(pts).transpose(1, 2).contiguous(), dev(qry).transpose(1, 2).contiguous(), 1, algo="binned")
    assert np.array_equal(cf.cpu().numpy(), idx)


def test_dispatch_picks_the_binned_search_for_the_model_calls():
    """k_nearest_neighbor itself (algo="auto") equals both explicit kernels on a level-1 search."""
    r = I.rng(4200)
    pts, qry = dev(I.pixel_cloud(r, 4, 4096, 144, 240)), dev(raster(4, 144, 240))
    auto = ops.k_nearest_neighbor(pts, qry, 1)
    assert torch.equal(auto, W.k_nearest_neighbor_ties(pts, qry, 1, algo="binned"))
    assert torch.equal(auto, W.k_nearest_neighbor_ties(pts, qry, 1, algo="sweep"))
    with pytest.raises(RuntimeError):
        W.k_nearest_neighbor_ties(pts, qry, 2, algo="binned")


@pytest.mark.parametrize("order", ["random", "columns", "reversed", "two interleaved rasters"])
def test_queries_in_any_order(order):
    """Nothing assumes a raster: shuffled queries (every wave's box spans the image) give the same answers."""
    r = I.rng(4300)
    H, Wd, N, B = 72, 120, 2048, 3
    pts, qry = I.pixel_cloud(r, B, N, H, Wd), raster(B, H, Wd)
    if order == "random":
        qry = qry[:, r.permutation(H * Wd)]
    elif order == "columns":
        qry = qry.reshape(B, H, Wd, 2).transpose(0, 2, 1, 3).reshape(B, H * Wd, 2)
    elif order == "reversed":
        qry = qry[:, ::-1]
    else:
        qry = np.concatenate([qry[:, 0::2], qry[:, 1::2] + np.float32(0.37)], 1)
    assert_same(dev(pts), dev(np.ascontiguousarray(qry)), order)


def test_points_far_outside_and_queries_outside_the_sampled_domain():
    r = I.rng(4400)
    H, Wd, N, B = 60, 90, 1500, 2
    pts = I.pixel_cloud(r, B, N, H, Wd)
    pts[:, :40] *= np.float32(50.0)                       # far beyond the image on the positive side
    pts[:, 40:80] = -pts[:, 40:80] * np.float32(1e4)      # far on the negative side: |p|^2 ~ 1e12
    pts[0, 80:90] = np.float32(3e18)                      # |p|^2 overflows to +inf: never the nearest
    assert_same(dev(pts), dev(raster(B, H, Wd)), "outliers")
    # queries the build kernel's strided sample cannot all see: a raster plus stragglers far away (and one wave of them)
    qry = raster(B, H, Wd)
    qry[:, 1::97] += np.float32(4000.0)
    qry[:, 2000:2064] = np.float32(-1e5)
    assert_same(dev(pts), dev(qry), "straggler queries")
    # the cloud entirely outside the query domain, on one side / around it
    away = pts.copy()
    away[..., 0] += np.float32(500.0)
    assert_same(dev(away), dev(raster(B, H, Wd)), "cloud beside the domain")
    ring = pts.copy()
    ang = r.uniform(0, 2 * np.pi, (B, N)).astype(np.float32)
    ring[..., 0], ring[..., 1] = 45 + 300 * np.cos(ang), 30 + 300 * np.sin(ang)
    assert_same(dev(ring), dev(raster(B, H, Wd)), "cloud around the domain")


def test_non_finite_points_and_queries():
    """NaN / inf coordinates give NaN / +inf distances, which `d < best` never takes -- in either kernel; a query whose
    distances are all like that returns index 0."""
    r = I.rng(4500)
    H, Wd, N, B = 36, 60, 1024, 2
    pts = I.pixel_cloud(r, B, N, H, Wd)
    pts[0, 5], pts[0, 17, 0], pts[1, 0, 1], pts[1, 1000] = np.nan, np.inf, -np.inf, np.nan
    qry = raster(B, H, Wd)
    qry[0, 70], qry[1, 128:192, 0], qry[1, 300, 1] = np.nan, np.inf, -np.inf
    idx, dist = assert_same(dev(pts), dev(qry), "non-finite")
    assert idx[0, 70, 0] == 0 and (idx[1, 128:192, 0] == 0).all()
    finite_q = np.isfinite(qry).all(-1)
    assert not np.isin(idx[0][finite_q[0]], [5, 17]).any() and not np.isin(idx[1][finite_q[1]], [0, 1000]).any()
    allbad = np.full((1, 128, 2), np.nan, np.float32)
    i, _ = W.k_nearest_neighbor_with_distances(dev(allbad), dev(raster(1, 20, 30)), 1, algo="binned")
    assert (i == 0).all()


def test_equal_distances_take_the_lowest_index():
    """Duplicated points, lattice points (exact ties between different cells), a cloud that is one point."""
    r = I.rng(4600)
    H, Wd, B = 40, 56, 2
    base = I.pixel_cloud(r, B, 700, H, Wd)
    dup = np.concatenate([base, base[:, ::-1], base], 1)  # every point three times, the copies in other positions
    idx, _ = assert_same(dev(dup), dev(raster(B, H, Wd)), "duplicates")
    assert idx.max() < 1400  # never the third copy; which of the first two depends on the lower index
    lattice = np.stack(np.meshgrid(np.arange(0, 56, 2, dtype=np.float32), np.arange(0, 40, 2, dtype=np.float32)), -1).reshape(1, -1, 2)
    lattice = np.repeat(lattice[:, r.permutation(lattice.shape[1])], B, 0)
    idx, dist = assert_same(dev(lattice), dev(raster(B, H, Wd)), "lattice")  # odd pixels are equidistant to 2 or 4 lattice points
    oi, od = O.k_nearest_neighbor(lattice, raster(B, H, Wd), 1, return_dists=True)
    assert np.array_equal(idx, oi)
    assert_bits_equal(dist, od)
    one = np.broadcast_to(np.array([7.25, 3.5], np.float32), (B, 300, 2)).copy()
    idx, _ = assert_same(dev(one), dev(raster(B, H, Wd)), "one location")
    assert (idx == 0).all()


def test_empty_regions_and_degenerate_domains():
    r = I.rng(4700)
    H, Wd, B = 64, 96, 2
    corner = (r.random((B, 900, 2), dtype=np.float32) * np.float32(6.0)).astype(np.float32)  # all points in one corner: most cells empty
    assert_same(dev(corner), dev(raster(B, H, Wd)), "clustered")
    two = np.concatenate([corner, corner + np.array([88.0, 57.0], np.float32)], 1)
    assert_same(dev(two), dev(raster(B, H, Wd)), "two clusters")
    pts = I.pixel_cloud(r, B, 1200, H, Wd)
    same_q = np.broadcast_to(np.array([11.0, 13.0], np.float32), (B, 500, 2)).copy()  # zero-area query domain
    assert_same(dev(pts), dev(same_q), "identical queries")
    line_q = raster(B, 1, 700)  # zero-height domain
    assert_same(dev(pts), dev(line_q), "queries on a line")
    assert_same(dev(pts), dev(raster(B, 1, 1)), "one query")
    big = raster(B, 30, 50, scale=1e6, x0=-2e7, y0=3e7)  # huge coordinates: distances ~1e13, quantised to ~1e6
    pts_big = (r.random((B, 1000, 2), dtype=np.float32) * np.array([5e7, 3e7], np.float32) + np.array([-2e7, 3e7], np.float32)).astype(np.float32)
    assert_same(dev(pts_big), dev(big), "huge coordinates")
    tiny = raster(B, 30, 50, scale=1e-6)
    assert_same(dev((pts * np.float32(1e-6)).astype(np.float32)), dev(tiny), "tiny coordinates")


def test_ids_range_clouds_with_rounding_level_ties():
    """Coordinates in the range the IDS transform produces (|p|^2 ~ 1e2 ... 1e4, distances quantised to ~1e-3): the winner is
    often decided by rounding, so a geometric prune without the error margin would pick another point."""
    r = I.rng(4800)
    pts = I.ids_cloud(r, 2, 8192, 2)
    gx, gy = np.meshgrid(np.arange(240, dtype=np.float32) * 0.125 - 14.5, np.arange(144, dtype=np.float32) * 0.125 - 8.5)
    grid = np.broadcast_to(np.stack([gx.ravel(), gy.ravel()], -1)[None], (2, 34560, 2)).copy()
    idx, dist = assert_same(dev(pts), dev(grid), "ids range")
    rows = r.choice(34560, 300, replace=False)
    oi, od = O.k_nearest_neighbor(pts, grid[:, rows], 1, return_dists=True)
    assert np.array_equal(idx[:, rows], oi)
    assert_bits_equal(dist[:, rows], od)
    far = pts + np.float32(3000.0)  # |p|^2 ~ 2e7: one ulp of the distance is 2, hundreds of points tie for every query
    assert_same(dev(far), dev(grid + np.float32(3000.0)), "large offset")


def test_workspace_contract():
    """rpe_knn's optional workspace: results never depend on it; too little of it means the sweep ("auto") or an error (when
    the binned search was asked for explicitly); a misaligned pointer is refused."""
    from rpeflow_amd import _lib
    lib = _lib.lib()
    r = I.rng(1)
    pts, qry = dev(I.pixel_cloud(r, 8, 2048, 72, 120)), dev(raster(8, 72, 120))
    want = W.k_nearest_neighbor_ties(pts, qry, 1, algo="sweep")
    B, M, Q = 8, 2048, 72 * 120
    need = lib.rpe_knn_workspace_bytes(B, M, Q, 2, 1, 3)
    assert need >= 16 * M * B and lib.rpe_knn_workspace_bytes(B, M, Q, 2, 1, 3 | 0x100) == 0  # (forced sweep: none)
    assert lib.rpe_knn_workspace_bytes(B, 256, 135, 2, 1, 3) == 0 and lib.rpe_knn_workspace_bytes(B, 256, 135, 2, 1, 3 | 0x200) > 0
    work = torch.empty(need + 16, dtype=torch.uint8, device=DEV)

    def call(mode, ptr, nbytes):
        idx = torch.full((B, Q, 1), -7, dtype=torch.int64, device=DEV)
        rc = lib.rpe_knn(pts.data_ptr(), *pts.stride(), qry.data_ptr(), *qry.stride(), B, M, Q, 2, 1, mode, idx.data_ptr(), None, ptr, nbytes, None)
        torch.cuda.synchronize()
        return rc, idx

    for mode, ptr, nbytes in ((3, work.data_ptr(), need), (3, None, 0), (3, work.data_ptr(), need - 16), (3 | 0x100, work.data_ptr(), need),
                              (3 | 0x200, work.data_ptr(), need), (0, work.data_ptr(), need)):
        rc, idx = call(mode, ptr, nbytes)
        assert rc == 0 and torch.equal(idx, want), (mode, nbytes)
    assert call(3 | 0x200, work.data_ptr(), need - 16)[0] == -1   # binned demanded, workspace too small
    assert call(3 | 0x200, None, 0)[0] == -1
    assert call(3, work.data_ptr() + 4, need)[0] == -1            # misaligned
    assert call(3 | 0x300, work.data_ptr(), need)[0] == -1        # both algorithm flags
    assert call(2, None, 0)[0] == -1                              # not a tie mode
