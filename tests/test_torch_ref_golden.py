"""CPU: pin the PyTorch-CPU port (oracle/torch_ref.py, the cpu_baseline implementation)
against the reference outputs in tests/golden/."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests import cases as K
from tests import inputs as I
from tests.check import assert_knn_tie_aware
from tests.test_oracle_golden import _shapes_corr3d, _shapes_pointconv

T = torch.from_numpy


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ["knn3d_k16_ids_2x1024x777", "knn2d_k1_pix_2x1024x60x36", "knn3d_k32_unit_1x300x200"])
def test_knn(golden_dir, name):
    inp, qry, k = K.knn_inputs(name)
    g = G(golden_dir, name)
    d = R.squared_distance(T(qry), T(inp))
    vals, idx = d.topk(k, dim=2, largest=False)
    assert torch.equal(idx, R.k_nearest_neighbor(T(inp), T(qry), k))
    if not np.array_equal(vals.numpy().view(np.uint32), g["dist"].view(np.uint32)):
        pytest.skip("this host's BLAS rounds matmul differently from the golden host (SURVEY.md H1)")
    assert_knn_tie_aware(idx.numpy(), vals.numpy(), g["idx"], g["dist"], g["next_dist"], name)


@pytest.mark.parametrize("name", ["fps_ids_2x2048_512", "fps_dup_1x512_256"])
def test_fps(golden_dir, name):
    xyz, S = K.fps_inputs(name)
    assert np.array_equal(R.furthest_point_sampling(T(xyz), S).numpy(), G(golden_dir, name)["idx"])


def test_correlation(golden_dir):
    a, b, md = K.corr_inputs("corr_2x24x20x28_md4")
    assert np.abs(R.correlation2d(T(a), T(b), md).numpy() - G(golden_dir, "corr_2x24x20x28_md4")["out"]).max() < 1e-6


def test_glue(golden_dir):
    d, g = {k: T(v) for k, v in K.glue_inputs().items()}, G(golden_dir, "glue_ops")
    tol = dict(rtol=0, atol=1e-6)
    assert np.array_equal(R.batch_indexing_channel_first(d["feat_3d"], d["idx"]).numpy(), g["gather_cf"])
    assert np.array_equal(R.batch_indexing_channel_last(d["feat_3d"].transpose(1, 2), d["idx"]).numpy(), g["gather_cl"])
    np.testing.assert_allclose(R.backwarp_2d(d["feat_2d"], d["flow"], "border").numpy(), g["backwarp_2d"], **tol)
    np.testing.assert_allclose(R.grid_sample_wrapper(d["feat_2d"], d["xy"]).numpy(), g["grid_sample_wrapper"], **tol)
    np.testing.assert_allclose(R.knn_interpolation(d["xyz"], d["feat_3d"], d["xyz_q"], 3).numpy(), g["knn_interp"], **tol)
    np.testing.assert_allclose(R.backwarp_3d(d["xyz"], d["xyz"] + 0.1, d["flow3"], 3).numpy(), g["backwarp_3d"], **tol)
    np.testing.assert_allclose(R.project_feat_with_nn_corr(d["xy"], d["feat_2d"], d["feat_3d"]).numpy(), g["project_feat"], **tol)


def _load(m, shapes, seed):
    m.load_state_dict({k: T(v) for k, v in I.fill_params(shapes, seed).items()}, strict=True)
    return m.eval()


@torch.no_grad()
def test_blocks(golden_dir):
    c, x = K.BLOCK_CASES["pointconv_down"], K.block_inputs("pointconv_down")
    m = _load(R.PointConvDownSampling(c["C"], c["Cout"], norm=c["norm"], k=c["k"]), _shapes_pointconv(c["C"], c["Cout"], c["norm"]), c["seed"] + 1000)
    np.testing.assert_allclose(m(T(x["xyz"]), T(x["feat"]), T(x["sampled"])).numpy(), G(golden_dir, "pointconv_down")["out"], rtol=1e-5, atol=1e-5)
    c, x = K.BLOCK_CASES["pointconv_nosample"], K.block_inputs("pointconv_nosample")
    m = _load(R.PointConvNoSampling(c["C"], c["Cout"], norm=c["norm"], k=c["k"]), _shapes_pointconv(c["C"], c["Cout"], c["norm"]), c["seed"] + 1000)
    np.testing.assert_allclose(m(T(x["xyz"]), T(x["feat"])).numpy(), G(golden_dir, "pointconv_nosample")["out"], rtol=1e-5, atol=1e-5)
    c, x = K.BLOCK_CASES["correlation3d"], K.block_inputs("correlation3d")
    m = _load(R.Correlation3D(c["C"], c["C"], k=c["k"]), _shapes_corr3d(c["C"]), c["seed"] + 1000)
    out = m(T(x["xyz1"]), T(x["feat1"]), T(x["xyz2"]), T(x["feat2"]))
    np.testing.assert_allclose(out.numpy(), G(golden_dir, "correlation3d")["out"], rtol=1e-5, atol=1e-5)
