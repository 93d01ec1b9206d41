"""rpeflow_amd/roofline.py against the figures SURVEY.md section 8(d) writes out (bytes and flops of the hot-path operators)."""
from rpeflow_amd import roofline as R

SIZES = [(576 >> (l + 1), 960 >> (l + 1)) for l in range(6)]   # 544 x 960 -> resize_to_64x 576 x 960


def test_config2_correlation_bytes_and_flops():
    assert R.correlation2d(1, 256, 544, 960) == 1_238_753_280
    assert R.correlation2d_flops(1, 256, 544, 960) == 2 * 256 * 4876 * 8620  # 21.52 GFLOP
    # intensity 17.4 flop/B: left of the fp32 ridge (157.3 TF / 8 TB/s = 19.7) -> HBM-bound, floor 154.8 us
    floor, bound = R.floor_seconds(R.correlation2d(1, 256, 544, 960), R.correlation2d_flops(1, 256, 544, 960))
    assert bound == "hbm" and abs(floor * 1e6 - 154.8) < 0.1


def test_operator_formulas():
    assert R.knn_flops(8, 4096, 8192, 3) == 8 * 4096 * 8192 * 9               # B Q M (2 D + 3)
    assert R.knn(8, 4096, 8192, 3, 16) == 4 * 8 * 3 * (4096 + 8192) + 8 * 8 * 4096 * 16
    # PointConv: B Q [2 16 k (C + 3) + 2 16 (C + 3) Cout + 2 k (3 8 + 8 16)] -- the level-1 estimator layer is 15.03 GFLOP
    assert R.pointconv_flops(4, 4096, 195, 128) == 4 * 4096 * (2 * 16 * 16 * 198 + 2 * 16 * 198 * 128 + 2 * 16 * 152)
    assert abs(R.pointconv_flops(4, 4096, 195, 128) / 1e9 - 15.03) < 0.01
    # Correlation3D: 2 B N k ((2C + 3) C + C^2) + the weight nets and weighted sums
    assert R.correlation3d_flops(1, 1, 32, 1) == 2 * ((67 * 32) + 32 * 32) + 2 * 2 * (24 + 64 + 256) + 4 * 32
    assert R.conv1x1_flops(4, 96, 510, 144 * 240) == 2 * 4 * 144 * 240 * 96 * 510


def test_hotpath_tables_agree_and_price_the_matrix_categories_against_the_matrix_peak():
    b, f = R.hotpath_bytes(4, SIZES), R.hotpath_flops(4, SIZES)
    assert set(b) == set(f)
    bounds = {k: R.floor_seconds(b[k], f[k])[1] for k in b}
    for k in ("flow_estimator_3d", "feature_pyramid_3d", "correlation3d", "knn3d_k16", "knn2d_k1"):
        assert bounds[k] == "mfma", k
    for k in ("project_feat", "grid_sample", "backwarp_2d", "correlation2d", "project_pc2image", "flow_head_3d"):
        assert bounds[k] == "hbm", k
    # per sample the 3-D branch is ~19 GFLOP dense + ~0.54 G pair evaluations
    dense = (f["flow_estimator_3d"] + f["feature_pyramid_3d"] + f["correlation3d"]) / 4 / 1e9
    assert 18 < dense < 24
    pairs = (f["knn2d_k1"] / 7 + (f["knn3d_k16"]) / 9) / 4 / 1e9
    assert 0.3 < pairs < 0.6
