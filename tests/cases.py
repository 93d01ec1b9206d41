"""Golden-vector case table: one place that says which seeded inputs each
tests/golden/*.npz was generated from.  Used by tests/golden/make_golden.py
(with the reference) and by the parity tests (without it)."""
import numpy as np

from . import inputs as I

# name -> (coord generator, B, M, Q, D, k, seed)
KNN_CASES = {
    "knn3d_k16_ids_2x1024x777": ("ids", 2, 1024, 777, 3, 16, 101),
    "knn3d_k3_ids_2x1024x777": ("ids", 2, 1024, 777, 3, 3, 102),
    "knn3d_k16_unit_2x1024x777": ("unit", 2, 1024, 777, 3, 16, 103),
    "knn3d_k16_ids_1x4096x4096": ("ids", 1, 4096, 4096, 3, 16, 104),
    "knn3d_k3_unit_1x4096x4096": ("unit", 1, 4096, 4096, 3, 3, 105),
    "knn2d_k1_pix_2x1024x60x36": ("pix", 2, 1024, (36, 60), 2, 1, 106),
    "knn2d_k1_pix_1x4096x240x144": ("pix", 1, 4096, (144, 240), 2, 1, 107),
    "knn3d_k32_unit_1x300x200": ("unit", 1, 300, 200, 3, 32, 108),
    "knn3d_k1_ids_1x70x130": ("ids", 1, 70, 130, 3, 1, 109),
    # integer-lattice clouds: most distances tie; k = 32 is the reference extension's cap (k_nearest_neighbor_kernel.cu:24,68),
    # where torch.topk's std::sort of the first k - 1 is an introsort; both topk regimes (k * 64 <= M or not)
    "knn3d_k32_lattice_1x1500x160": ("lattice", 1, 1500, 160, 3, 32, 110),
    "knn3d_k32_lattice_2x2500x120": ("lattice", 2, 2500, 120, 3, 32, 111),
    "knn3d_k20_lattice_1x900x200": ("lattice", 1, 900, 200, 3, 20, 112),
}


def knn_inputs(name):
    kind, B, M, Q, D, k, seed = KNN_CASES[name]
    r = I.rng(seed)
    if kind == "pix":
        H, W = Q
        inp = I.pixel_cloud(r, B, M, H, W)
        qry = I.pixel_grid(B, H, W)
    elif kind == "ids":
        inp = I.ids_cloud(r, B, M, D)
        qry = I.ids_cloud(r, B, Q, D)
    elif kind == "lattice":
        inp = r.integers(0, 7, (B, M, D)).astype(np.float32)
        qry = r.integers(0, 7, (B, Q, D)).astype(np.float32)
    else:
        inp = I.unit_cloud(r, B, M, D)
        qry = I.unit_cloud(r, B, Q, D)
    return inp, qry, k


# name -> (kind, B, N, S, seed)
FPS_CASES = {
    "fps_ids_2x2048_512": ("ids", 2, 2048, 512, 201),
    "fps_ids_1x8192_4096": ("ids", 1, 8192, 4096, 202),
    "fps_unit_3x1000_333": ("unit", 3, 1000, 333, 203),
    "fps_dup_1x512_256": ("dup", 1, 512, 256, 204),
}


def fps_inputs(name):
    kind, B, N, S, seed = FPS_CASES[name]
    r = I.rng(seed)
    if kind == "ids":
        xyz = I.ids_cloud(r, B, N)
    elif kind == "unit":
        xyz = I.unit_cloud(r, B, N)
    else:  # every point appears twice: exercises the first-maximum tie rule
        half = I.unit_cloud(r, B, N // 2)
        xyz = np.concatenate([half, half], axis=1)
    return xyz, S


# name -> (B, N1, N2, D, kind, seed)
SQDIST_CASES = {
    "sqdist_ids_2x200x300x3": (2, 200, 300, 3, "ids", 301),
    "sqdist_unit_2x200x300x2": (2, 200, 300, 2, "unit", 302),
}


def sqdist_inputs(name):
    B, N1, N2, D, kind, seed = SQDIST_CASES[name]
    r = I.rng(seed)
    gen = I.ids_cloud if kind == "ids" else I.unit_cloud
    return gen(r, B, N1, D), gen(r, B, N2, D)


# name -> (B, C, H, W, md, seed)
CORR_CASES = {
    "corr_2x24x20x28_md4": (2, 24, 20, 28, 4, 401),
    "corr_1x32x36x60_md4": (1, 32, 36, 60, 4, 402),
    "corr_1x7x9x15_md4": (1, 7, 9, 15, 4, 403),
    "corr_1x5x11x13_md2": (1, 5, 11, 13, 2, 404),
}


def corr_inputs(name):
    B, C, H, W, md, seed = CORR_CASES[name]
    r = I.rng(seed)
    return I.feature_map(r, B, C, H, W), I.feature_map(r, B, C, H, W), md


def corr_grad_output(name):
    """Seeded upstream gradient for the correlation backward cases."""
    B, C, H, W, md, seed = CORR_CASES[name]
    n = 2 * md + 1
    return I.rng(seed + 5000).standard_normal((B, n * n, H, W), dtype=np.float32)


# glue ops: one small case each (B, C2, C3, H, W, N, seed)
GLUE = dict(B=2, C2=12, C3=10, H=18, W=30, N=200, M=97, seed=501)


def glue_inputs():
    g = GLUE
    r = I.rng(g["seed"])
    d = {}
    d["feat_2d"] = I.feature_map(r, g["B"], g["C2"], g["H"], g["W"])
    d["flow"] = I.flow_field(r, g["B"], g["H"], g["W"], std=4.0)
    d["feat_3d"] = r.standard_normal((g["B"], g["C3"], g["N"]), dtype=np.float32)
    d["xy"] = np.ascontiguousarray(I.pixel_cloud(r, g["B"], g["N"], g["H"], g["W"]).transpose(0, 2, 1))
    d["xyz"] = np.ascontiguousarray(I.ids_cloud(r, g["B"], g["N"]).transpose(0, 2, 1))
    d["xyz_q"] = np.ascontiguousarray(I.ids_cloud(r, g["B"], g["M"]).transpose(0, 2, 1))
    d["flow3"] = (r.standard_normal((g["B"], 3, g["N"]), dtype=np.float32) * np.float32(0.3))
    d["idx"] = r.integers(0, g["N"], (g["B"], g["M"], 5)).astype(np.int64)
    return d


# 3-D blocks: name -> dict
BLOCK_CASES = {
    "pointconv_down": dict(B=2, M=300, Q=120, C=13, Cout=20, k=16, norm="batch_norm", seed=601),
    "pointconv_nosample": dict(B=2, M=150, Q=150, C=29, Cout=17, k=16, norm=None, seed=602),
    "correlation3d": dict(B=2, N=140, C=12, k=16, seed=603),
    "flow_estimator3d": dict(B=2, N=200, channels=[20, 24, 24, 16], k=16, seed=604),
    "feature_pyramid3d": dict(B=2, N=512, samples=[256, 128, 64], channels=[8, 12, 16, 20], norm="batch_norm", k=16, seed=605),
}


# the same modules OUTSIDE the configuration the fused kernels are built for (any k, instance_norm, relu / no activation,
# training-mode BatchNorm, wide Correlation3D): rpeflow_amd runs the reference's op sequence on the GPU there
GENERAL_CASES = {
    "pointconv_down_k9_instnorm_relu": dict(kind="down", B=2, M=300, Q=120, C=13, Cout=20, k=9, norm="instance_norm", activation="relu", train=False, seed=611),
    "pointconv_nosample_k20_none": dict(kind="nosample", B=2, M=150, Q=150, C=29, Cout=17, k=20, norm=None, activation=None, train=False, seed=612),
    "pointconv_nosample_k16_bn_train": dict(kind="nosample", B=3, M=200, Q=200, C=10, Cout=12, k=16, norm="batch_norm", activation="leaky_relu", train=True, seed=613),
    "correlation3d_k5": dict(kind="corr", B=2, N=140, C=12, Cout=12, k=5, seed=614),
    "correlation3d_wide200": dict(kind="corr", B=1, N=90, C=6, Cout=200, k=16, seed=615),
}
for _name, _c in GENERAL_CASES.items():
    BLOCK_CASES[_name] = dict(_c, **({"N": _c["M"]} if "M" in _c else {}))


def block_inputs(name):
    c = BLOCK_CASES[name]
    r = I.rng(c["seed"])
    cl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1))
    if name.startswith("pointconv"):
        xyz = cl(I.ids_cloud(r, c["B"], c["M"]))
        feat = r.standard_normal((c["B"], c["C"], c["M"]), dtype=np.float32)
        sampled = xyz[:, :, : c["Q"]].copy() if name.startswith("pointconv_down") else None
        return dict(xyz=xyz, feat=feat, sampled=sampled)
    if name == "flow_estimator3d":
        return dict(xyz=cl(I.ids_cloud(r, c["B"], c["N"])), feat=r.standard_normal((c["B"], c["channels"][0], c["N"]), dtype=np.float32))
    if name == "feature_pyramid3d":
        pc1 = cl(I.ids_cloud(r, c["B"], c["N"]))
        return dict(pc1=pc1, pc2=(pc1 + r.standard_normal(pc1.shape, dtype=np.float32) * np.float32(0.05)).astype(np.float32))
    xyz1 = cl(I.ids_cloud(r, c["B"], c["N"]))
    xyz2 = (xyz1 + r.standard_normal(xyz1.shape, dtype=np.float32) * np.float32(0.2)).astype(np.float32)
    feat1 = r.standard_normal((c["B"], c["C"], c["N"]), dtype=np.float32)
    feat2 = r.standard_normal((c["B"], c["C"], c["N"]), dtype=np.float32)
    return dict(xyz1=xyz1, xyz2=xyz2, feat1=feat1, feat2=feat2)


# event voxelisation (event_utils.eventsToVoxel): (N events, H, W, bins, polarity split, seed)
EVENT_CASES = {
    "events_5000_24x40_b5": (5000, 24, 40, 5, False, 601),
    "events_20000_36x60_b10_pol": (20000, 36, 60, 10, True, 602),
    "events_300_9x15_b1_pol": (300, 9, 15, 1, True, 603),
    # float32 [N,4] arrays, as load_events_h5 hands them to the datasets (event_utils.py:11-20): float32 arithmetic throughout
    "events_f32_20000_36x60_b10_pol": (20000, 36, 60, 10, True, 604),
    "events_f32_5000_24x40_b5": (5000, 24, 40, 5, False, 605),
    # every event at ONE timestamp: dt = 0, t_norm = 0/0 -- the reference's weights are NaN in every bin of every touched pixel
    "events_same_t_40_6x8_b3_pol": (40, 6, 8, 3, True, 606),
    "events_f32_same_t_40_6x8_b3": (40, 6, 8, 3, False, 607),
}


def event_inputs(name):
    """[N,4] float64 (x, y, t, polarity) in time order: microsecond-like timestamps with repeats, polarity in {0, 1}."""
    n, H, W, bins, pol, seed = EVENT_CASES[name]
    r = I.rng(seed)
    t = np.sort(r.integers(1_000_000, 1_050_000, n)).astype(np.float64)
    if "same_t" in name:
        t[:] = t[0]
    ev = np.stack([r.integers(0, W, n).astype(np.float64), r.integers(0, H, n).astype(np.float64), t,
                   r.integers(0, 2, n).astype(np.float64)], axis=1)
    if name.startswith("events_f32"):
        ev = ev.astype(np.float32)  # (timestamps of 1.0e6 .. 1.05e6 us: exact in float32)
    return ev, H, W, bins, pol


# Section 8(f) blocks pinned at block level: Restormer cross blocks (restormer_arch.py:207-222, 287-302),
# convex_upsample (utils.py:201-214), resize_flow2d (utils.py:217-224)
FBLOCK_CASES = {
    "cross_block2d": dict(B=2, C=24, heads=2, H=20, W=28, seed=701),
    "cross_block2d_3heads": dict(B=1, C=48, heads=3, H=18, W=30, seed=702),
    "cross_block3d": dict(B=2, C=32, heads=4, N=300, seed=703),
    # the level-1 instantiations of the forward (144 x 240 maps, 4096 points): stored on a stride-8 grid / every 8th point
    "cross_block2d_level1_c96": dict(B=1, C=96, heads=2, H=144, W=240, seed=708, stride=8),
    "cross_block2d_level1_c81": dict(B=1, C=81, heads=1, H=144, W=240, seed=709, stride=8),
    "cross_block3d_level1_c32": dict(B=2, C=32, heads=1, N=4096, seed=710, stride=8),
    "convex_upsample4": dict(B=2, H=9, W=15, scale=4, seed=704),
    "convex_upsample8": dict(B=1, H=6, W=10, scale=8, seed=705),
    "resize_flow2d": dict(B=2, H=64, W=128, th=60, tw=120, seed=706),
    "resize_flow2d_same": dict(B=1, H=64, W=64, th=64, tw=64, seed=707),
}


def fblock_inputs(name):
    c = FBLOCK_CASES[name]
    r = I.rng(c["seed"])
    if name.startswith("cross_block2d"):
        return dict(x=I.feature_map(r, c["B"], c["C"], c["H"], c["W"]), y=I.feature_map(r, c["B"], c["C"], c["H"], c["W"]))
    if name.startswith("cross_block3d"):
        return dict(x=r.standard_normal((c["B"], c["C"], c["N"]), dtype=np.float32), y=r.standard_normal((c["B"], c["C"], c["N"]), dtype=np.float32))
    if name.startswith("convex_upsample"):
        s = c["scale"]
        return dict(flow=I.flow_field(r, c["B"], c["H"], c["W"], std=2.0), mask=r.standard_normal((c["B"], 9 * s * s, c["H"], c["W"]), dtype=np.float32))
    return dict(flow=I.flow_field(r, c["B"], c["H"], c["W"], std=3.0))


# The benched configuration (bench.py, BASELINE config 3): batch 4 of 544x960 frame pairs + 8192 points,
# samples frame_pair(1000 + i), parameters tests.inputs.model_params
BENCH_CASE = dict(B=4, H=544, W=960, N=8192, first_seed=1000)
# ... and the DSEC evaluation shape (bench.py --config dsec, BASELINE config 5): batch 3 (conf/test/dsec.yaml:24) of 480x640
# frame pairs + 8192 points, 4-channel flow_3d targets, samples frame_pair(2000 + i, dsec=True)
BENCH_CASE_DSEC = dict(B=3, H=480, W=640, N=8192, first_seed=2000)
