"""Seeded synthetic inputs shared by tests/golden/make_golden.py, the parity
tests and bench.py.  Everything comes from numpy.random.default_rng(seed) so
the golden generator (which runs next to the reference) and the tests (which
run without it) rebuild identical arrays; goldens store outputs only."""
import numpy as np


def rng(seed):
    return np.random.default_rng(seed)


def unit_cloud(r, B, N, D=3):
    """Points uniform in [0,1)^D, channel-last [B,N,D] (k_nearest_neighbor_test.cpp:31-32 style)."""
    return r.random((B, N, D), dtype=np.float32)


def ids_cloud(r, B, N, D=3):
    """Points in the range the model's IDS transform produces:
    x in +-14.5, y in +-8.5, z in 22..113; |p|^2 ~ 1e4 so fp32 distances tie often."""
    x = r.uniform(-14.5, 14.5, (B, N)).astype(np.float32)
    y = r.uniform(-8.5, 8.5, (B, N)).astype(np.float32)
    z = r.uniform(22.0, 113.0, (B, N)).astype(np.float32)
    return np.stack([x, y, z][:D], axis=-1)


def pixel_cloud(r, B, N, H, W):
    """2-D projected point positions, a little beyond the image, channel-last [B,N,2]."""
    x = r.uniform(-2.0, W + 1.0, (B, N)).astype(np.float32)
    y = r.uniform(-2.0, H + 1.0, (B, N)).astype(np.float32)
    return np.stack([x, y], axis=-1)


def pixel_grid(B, H, W):
    """mesh_grid flattened, channel-last [B,H*W,2] (models/utils.py:172-183, 300-301)."""
    gx = np.broadcast_to(np.arange(W, dtype=np.float32)[None, :], (H, W)).reshape(-1)
    gy = np.broadcast_to(np.arange(H, dtype=np.float32)[:, None], (H, W)).reshape(-1)
    return np.broadcast_to(np.stack([gx, gy], -1)[None], (B, H * W, 2)).copy()


def feature_map(r, B, C, H, W):
    return r.standard_normal((B, C, H, W), dtype=np.float32)


def flow_field(r, B, H, W, std=3.0):
    return (r.standard_normal((B, 2, H, W), dtype=np.float32) * np.float32(std)).astype(np.float32)


from rpeflow_amd.synthetic import MODEL_SCALES, MODEL_SEED, fill_params, model_params  # noqa: E402,F401  (the generators live in the package: bench.py uses them too)


def frame_pair(seed, H=544, W=960, N=8192, f=1050.0, dsec=False):
    """One synthetic evaluation sample; the generator lives in the package
    because bench and the harness use it too."""
    from rpeflow_amd.synthetic import frame_pair as gen
    return gen(seed, H, W, N, f, dsec)


STRESS_MODEL_SEED = 31337  # the second parameter fill (model_*_stress goldens)


def frame_pair_stress(seed, H=544, W=960, N=8192, f=1050.0, dsec=False):
    """A LARGE-MOTION sample for the composition's parity (round-3 review, weak #1): the second cloud is a rigid motion of the
    first (a 2-degree rotation about the vertical axis + a translation) plus N(0, 0.5^2) noise (lateral parts scaled by W / 960), about 5 % of the points of
    either cloud project outside the frame (so grid_sample_wrapper's zero padding, the nearest-pixel search off the raster and
    backwarp's border clamp all take part), and the 2-D targets carry zero-mask and NaN pixels as DSEC's do
    (eval_noocc.py:57-99 masks both).  Same keys and dtypes as frame_pair."""
    r = np.random.default_rng(seed)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    images = r.integers(0, 256, (6, H, W), dtype=np.uint8)
    event_voxel = r.standard_normal((20, H, W), dtype=np.float32)
    z = r.uniform(4.0, 35.0, N)
    m = 0.013  # margin on each side: (1 + 2m)^2 - 1 = 5.3 % of the area lies outside the frame
    u = r.uniform(-m * W, (1 + m) * W - 1.0, N)
    v = r.uniform(-m * H, (1 + m) * H - 1.0, N)
    pc1 = np.stack([(u - cx) * z / f, (v - cy) * z / f, z])
    g = W / 960.0  # the image-plane part of the motion scales with the frame, so that small test frames keep most points inside
    a = np.deg2rad(2.0 * g)
    rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    pc2 = rot @ pc1 + np.array([[0.6 * g], [-0.25 * g], [0.8]]) + r.standard_normal((3, N)) * np.array([[0.5 * g], [0.5 * g], [0.5]])
    pc2[2] = np.maximum(pc2[2], 1.0)  # depth stays in front of the camera (the IDS transform takes its logarithm)
    pc1, pc2 = pc1.astype(np.float32), pc2.astype(np.float32)
    flow_2d = np.concatenate([r.standard_normal((2, H, W)) * 25.0, (r.random((1, H, W)) < 0.85)]).astype(np.float32)
    flow_2d[:2][:, r.random((H, W)) < 0.01] = np.nan
    flow_3d = (pc2 - pc1).astype(np.float32)
    sample = {"images": images, "event_voxel": event_voxel, "pcs": np.concatenate([pc1, pc2]).astype(np.float32),
              "flow_2d": flow_2d, "flow_3d": flow_3d, "intrinsics": np.array([f, cx, cy], np.float32)}
    if dsec:
        sample["flow_3d"] = np.concatenate([flow_3d, (r.random((1, N)) < 0.9).astype(np.float32)])
    else:
        sample["occ_mask_3d"] = (r.random(N) < 0.2).astype(np.float32)
    return sample


def masked_epes(flow_2d, flow_3d, sample):
    """EPE2D / EPE3D of one sample's prediction as the evaluators count them (eval_withocc.py:71-87, eval_noocc.py:57-75):
    pixels / points with a zero mask channel or a NaN end-point error are left out."""
    t2, t3 = sample["flow_2d"], sample["flow_3d"]
    e2 = np.sqrt(((flow_2d - t2[:2]) ** 2).sum(0))
    m2 = (t2[2] > 0 if t2.shape[0] > 2 else np.ones_like(e2, bool)) & ~np.isnan(e2)
    e3 = np.sqrt(((flow_3d - t3[:3]) ** 2).sum(0))
    m3 = (t3[3] > 0 if t3.shape[0] > 3 else np.ones_like(e3, bool)) & ~np.isnan(e3)
    return float(e2[m2].astype(np.float64).mean()), float(e3[m3].astype(np.float64).mean())
