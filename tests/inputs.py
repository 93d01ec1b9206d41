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
