"""Randomised parity of the hot-path operators against the oracle: shapes, layouts and cloud kinds drawn from a seeded stream,
the same acceptance rules as the hand-written cases (indices and distance bit patterns exact for KNN, indices exact for FPS,
<= 5e-6 for correlation and the samplers).  Inside the regular GPU suite every operator runs a FIXED number of cases
(RPE_FUZZ_CASES, default 200: the same seeds on every box, so what the record covers does not depend on the box's speed);
the long runs (profiles/r05_fuzz_parity.txt, profiles/r06_fuzz_parity.txt) set RPE_FUZZ_SECONDS instead -- a time budget PER
OPERATOR -- and RPE_FUZZ_SEED_OFFSET for fresh seeds.  Every case is reproducible from its printed seed."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I
from tests.check import assert_bits_equal

pytestmark = pytest.mark.gpu

import rpeflow_amd.csrc as ops  # noqa: E402
from rpeflow_amd import utils as U  # noqa: E402
from rpeflow_amd.csrc import wrapper as W  # noqa: E402

DEV = "cuda:0"
CASES = int(os.environ.get("RPE_FUZZ_CASES", "200"))          # the suite: this many seeds per operator, whatever the box
BUDGET = float(os.environ.get("RPE_FUZZ_SECONDS", "0"))       # a long run: seeds until this many seconds are used (overrides CASES)
OFFSET = int(os.environ.get("RPE_FUZZ_SEED_OFFSET", "0"))  # a long run on fresh seeds: RPE_FUZZ_SECONDS=300 RPE_FUZZ_SEED_OFFSET=10000000


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def cloud(r, B, N, D):
    """One of: IDS-range coordinates (ties among the best), unit cube, a lattice (every distance tied many times),
    a cloud with duplicated points, tight clusters."""
    kind = int(r.integers(0, 5))
    if kind == 0:
        return I.ids_cloud(r, B, N, D)
    if kind == 1:
        return I.unit_cloud(r, B, N, D)
    if kind == 2:
        return r.integers(0, 6, (B, N, D)).astype(np.float32)
    if kind == 3:
        base = I.ids_cloud(r, B, N, D)
        base[:, N // 2:] = base[:, :N - N // 2]
        return base
    centres = I.ids_cloud(r, B, 8, D)
    return (centres[:, r.integers(0, 8, N)] + r.normal(0, 0.05, (B, N, D))).astype(np.float32)


BASES = {"knn": 910000, "fps": 920000, "corr": 930000, "sample": 940000, "interp": 950000, "project": 960000}  # first seed per operator


def run(case, name):
    """Calls case(rng, seed) with seeds base, base + 1, ...: CASES of them, or -- with RPE_FUZZ_SECONDS -- until that budget is
    used; returns the number of cases."""
    t0, n = time.perf_counter(), 0
    base = BASES[name] + OFFSET
    while (time.perf_counter() - t0 < BUDGET or n < 2) if BUDGET > 0 else n < CASES:
        seed = base + n
        try:
            case(I.rng(seed), seed)
        except AssertionError as err:
            raise AssertionError("%s: case with seed %d failed: %s" % (name, seed, err)) from err
        n += 1
    print("fuzz %s: %d cases in %.1f s" % (name, n, time.perf_counter() - t0))
    return n


def test_fuzz_knn():
    def case(r, seed):
        D, k = int(r.integers(1, 4)), int(r.choice([1, 1, 2, 3, 3, 5, 16, 16, 16, 31, 63]))
        B = int(r.integers(1, 5))
        M = int(r.choice([k, k + 1, 63, 64, 65, 200, 512, 1000, 1024, 1025, 2048, 4100])) if r.random() < 0.8 else int(r.integers(k, 3000))
        M = max(M, k, 4)  # (fewer than four points / queries: the reference's own layout sniff, shape[1] <= 3, becomes ambiguous)
        Q = int(r.choice([4, 17, 64, 300, 1024, 2049, 4096])) if r.random() < 0.7 else int(r.integers(4, 5000))
        if B * Q * M > 3e7:  # keep the oracle's O(Q M) scan to a fraction of a second
            Q = max(4, int(3e7 / (B * M)))
        inp, qry = cloud(r, B, M, D), cloud(r, B, Q, D)
        if r.random() < 0.3:  # queries among the points: zero distances, self matches
            take = min(Q, M)
            qry[:, :take] = inp[:, :take]
        algo = str(r.choice(["auto", "auto", "matrix", "insert"]))
        if algo == "matrix" and not (k < 64 and M >= 64 * k and M >= 256):
            algo = "auto"
        oi, od = O.k_nearest_neighbor(inp, qry, k, return_dists=True)
        if r.random() < 0.5:  # channel-first views, as the model passes them
            idx, dist = W.k_nearest_neighbor_with_distances(dev(inp.transpose(0, 2, 1)), dev(qry.transpose(0, 2, 1)), k, algo=algo)
        else:
            idx, dist = W.k_nearest_neighbor_with_distances(dev(inp), dev(qry), k, algo=algo)
        assert np.array_equal(idx.cpu().numpy(), oi), (B, M, Q, D, k, algo)
        assert_bits_equal(dist.cpu().numpy(), od, str((B, M, Q, D, k, algo)))
    assert run(case, "knn") >= 2


def test_fuzz_fps():
    def case(r, seed):
        B = int(r.integers(1, 9))
        N = int(r.choice([65, 300, 1024, 2048, 4097, 8192]))
        S = int(r.choice([1, 2, 64, N // 4, N // 2, N - 1]))
        S = max(1, min(S, N - 1))
        pts = cloud(r, B, N, 3)
        got = ops.furthest_point_sampling(dev(pts), S).cpu().numpy()
        assert np.array_equal(got, O.furthest_point_sampling(pts, S)), (B, N, S)
    assert run(case, "fps") >= 2


def test_fuzz_correlation2d():
    def case(r, seed):
        B, C = int(r.integers(1, 4)), int(r.choice([1, 2, 3, 16, 32, 33, 64, 96, 128]))
        H, W_ = (int(r.integers(1, 80)), int(r.integers(1, 130))) if r.random() < 0.5 else [(9, 15), (18, 30), (36, 60), (72, 120), (73, 121), (72, 124)][int(r.integers(0, 6))]
        md = 4 if r.random() < 0.7 else int(r.integers(0, 7))
        a, b = I.feature_map(r, B, C, H, W_), I.feature_map(r, B, C, H, W_)
        got = ops.correlation2d(dev(a), dev(b), md).cpu().numpy()
        ref = O.correlation2d(a, b, md)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 5e-6, (B, C, H, W_, md, float(np.abs(got - ref).max()))
    assert run(case, "corr") >= 2


def test_fuzz_bilinear_sampling():
    def case(r, seed):
        B, C, H, W_ = int(r.integers(1, 4)), int(r.integers(1, 90)), int(r.integers(2, 70)), int(r.integers(2, 100))
        f = I.feature_map(r, B, C, H, W_)
        if r.random() < 0.5:
            flow = I.flow_field(r, B, H, W_, std=float(r.choice([0.5, 3.0, 30.0])))
            mode = str(r.choice(["border", "zeros"]))
            got, ref = U.backwarp_2d(dev(f), dev(flow), mode).cpu().numpy(), O.backwarp_2d(f, flow, mode)
        else:
            N = int(r.integers(1, 3000))
            xy = np.ascontiguousarray(I.pixel_cloud(r, B, N, H, W_).transpose(0, 2, 1))
            got, ref = U.grid_sample_wrapper(dev(f), dev(xy)).cpu().numpy(), O.grid_sample_wrapper(f, xy)
        assert np.abs(got - ref).max() <= 5e-6 * max(1.0, float(np.abs(ref).max())), (B, C, H, W_)
    assert run(case, "sample") >= 2


def test_fuzz_knn_interpolation_and_backwarp_3d():
    def case(r, seed):
        B, M, Q, C = int(r.integers(1, 4)), int(r.integers(4, 2000)), int(r.integers(4, 3000)), int(r.integers(1, 70))
        a, q = cloud(r, B, M, 3).transpose(0, 2, 1), cloud(r, B, Q, 3).transpose(0, 2, 1)
        a, q = np.ascontiguousarray(a), np.ascontiguousarray(q)
        feat = r.standard_normal((B, C, M), dtype=np.float32)
        got = U.knn_interpolation(dev(a), dev(feat), dev(q)).cpu().numpy()
        ref = O.knn_interpolation(a, feat, q)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), (B, M, Q, C)
        split = int(r.integers(0, C + 1))
        pair = U.knn_interpolation(dev(a), (dev(feat[:, :split]), dev(feat[:, split:])), dev(q)).cpu().numpy()
        assert np.array_equal(pair, got)
        if M == Q or r.random() < 0.3:
            n = min(M, Q)
            flow = (0.1 * r.standard_normal((B, 3, n))).astype(np.float32)
            w = U.backwarp_3d(dev(a[:, :, :n]), dev(q[:, :, :n]), dev(flow)).cpu().numpy()
            wr = O.backwarp_3d(np.ascontiguousarray(a[:, :, :n]), np.ascontiguousarray(q[:, :, :n]), flow)
            assert np.abs(w - wr).max() <= 2e-5 * max(1.0, float(np.abs(wr).max())), (B, n)
    assert run(case, "interp") >= 2


def test_fuzz_project_feat_with_nn_corr():
    def case(r, seed):
        B, C2, C3 = int(r.integers(1, 4)), int(r.integers(1, 100)), int(r.integers(0, 200))
        H, W_, N = int(r.integers(2, 40)), int(r.integers(2, 60)), int(r.integers(4, 1500))
        f2, f3 = I.feature_map(r, B, C2, H, W_), r.standard_normal((B, C3, N), dtype=np.float32)
        xy = np.ascontiguousarray(I.pixel_cloud(r, B, N, H, W_).transpose(0, 2, 1))
        got = U.project_feat_with_nn_corr(dev(xy), dev(f2), dev(f3)).cpu().numpy()  # (its own nearest-point search inside)
        ref = O.project_feat_with_nn_corr(xy, f2, f3)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 5e-6 * max(1.0, float(np.abs(ref).max())), (B, C2, C3, H, W_, N)
    assert run(case, "project") >= 2


def _close_sum(got, ref, what):
    """fp32 sums of thousands of terms: 2e-6 of the largest output + 1e-4 relative (the module tests' rule)."""
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6 * float(np.abs(ref).max()) + 1e-6, err_msg=str(what))


@torch.no_grad()
def test_fuzz_pointconv_modules():
    from rpeflow_amd import pointconv as PC
    from tests.test_oracle_golden import _shapes_pointconv

    def case(r, seed):
        C, Cout = int(r.integers(1, 200)), int(r.choice([8, 16, 32, 64, 96, 128, 192]))
        B, M = int(r.integers(1, 4)), int(r.integers(16, 600))
        down = r.random() < 0.5
        Q = int(r.integers(1, M + 1)) if down else M
        norm = "batch_norm" if (down and r.random() < 0.5) else None
        xyz = np.ascontiguousarray(I.ids_cloud(r, B, M).transpose(0, 2, 1))
        feat = r.standard_normal((B, C, M), dtype=np.float32)
        cls = PC.PointConvDownSampling if down else PC.PointConvNoSampling
        params = I.fill_params(_shapes_pointconv(C, Cout, norm), seed)
        m = cls(C, Cout, norm=norm)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m = m.to(DEV).eval()
        sampled = np.ascontiguousarray(xyz[:, :, :Q])
        out = m(dev(xyz), dev(feat), dev(sampled)) if down else m(dev(xyz), dev(feat))
        ref = O.pointconv(params, xyz, feat, sampled_xyz=sampled if down else None, k=16, norm=norm)
        _close_sum(out.cpu().numpy(), ref, (B, C, Cout, M, Q, down, norm))
    BASES["pointconv"] = 970000
    assert run(case, "pointconv") >= 2


@torch.no_grad()
def test_fuzz_correlation3d_module():
    from rpeflow_amd import pwc3d_core as P3
    from tests.test_oracle_golden import _shapes_corr3d

    def case(r, seed):
        C = int(r.choice([16, 32, 48, 64, 96, 128, 192]))
        B, N = int(r.integers(1, 4)), int(r.integers(16, 400))
        xyz1 = np.ascontiguousarray(I.ids_cloud(r, B, N).transpose(0, 2, 1))
        xyz2 = (xyz1 + r.standard_normal(xyz1.shape, dtype=np.float32) * np.float32(r.choice([0.02, 0.2, 2.0]))).astype(np.float32)
        feat1, feat2 = r.standard_normal((B, C, N), dtype=np.float32), r.standard_normal((B, C, N), dtype=np.float32)
        params = I.fill_params(_shapes_corr3d(C), seed)
        m = P3.Correlation3D(C, C, k=16)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m = m.to(DEV).eval()
        out = m(dev(xyz1), dev(feat1), dev(xyz2), dev(feat2))
        _close_sum(out.cpu().numpy(), O.correlation3d(params, xyz1, feat1, xyz2, feat2, k=16), (B, C, N))
    BASES["corr3d"] = 980000
    assert run(case, "corr3d") >= 2


def test_fuzz_gathers():
    def case(r, seed):
        B, C, N = int(r.integers(1, 5)), int(r.integers(1, 70)), int(r.integers(1, 3000))
        shape = tuple(int(v) for v in r.integers(1, 40, int(r.integers(1, 3))))
        data = r.standard_normal((B, C, N), dtype=np.float32)
        idx = r.integers(0, N, (B,) + shape).astype(np.int64)
        assert np.array_equal(U.batch_indexing_channel_first(dev(data), dev(idx)).cpu().numpy(), O.batch_indexing_channel_first(data, idx))
        last = np.ascontiguousarray(data.transpose(0, 2, 1))
        assert np.array_equal(U.batch_indexing_channel_last(dev(last), dev(idx)).cpu().numpy(), O.batch_indexing_channel_last(last, idx))
    BASES["gather"] = 990000
    assert run(case, "gather") >= 2
