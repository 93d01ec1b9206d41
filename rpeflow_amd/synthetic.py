"""Synthetic evaluation samples: no dataset ships with this repository, so the
harness and bench generate frame pairs of the FlyingThings3D / DSEC shapes from a seeded numpy stream."""
import numpy as np
import torch


def frame_pair(seed, H=544, W=960, N=8192, f=1050.0, dsec=False):
    """One sample with the keys the reference datasets return (flyingthings3d.py:228-234): uint8 RGB
    pair, 20-channel event voxel, two back-projected clouds (pc2 = pc1 + N(0,0.05^2)), targets.
    ``dsec``: flow_3d carries a 4th mask channel and there is no occ_mask_3d (dsec.py:762,777-784)."""
    r = np.random.default_rng(seed)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    images = r.integers(0, 256, (6, H, W), dtype=np.uint8)
    event_voxel = r.standard_normal((20, H, W), dtype=np.float32)
    z = r.uniform(2.0, 35.0, N)
    u = r.uniform(0.0, W - 1.0, N)
    v = r.uniform(0.0, H - 1.0, N)
    pc1 = np.stack([(u - cx) * z / f, (v - cy) * z / f, z]).astype(np.float32)
    pc2 = (pc1 + r.standard_normal((3, N)) * 0.05).astype(np.float32)
    flow_2d = np.concatenate([r.standard_normal((2, H, W)) * 5.0, np.ones((1, H, W))]).astype(np.float32)
    flow_3d = (pc2 - pc1).astype(np.float32)
    occ = (r.random(N) < 0.2).astype(np.float32)
    sample = {"images": images, "event_voxel": event_voxel, "pcs": np.concatenate([pc1, pc2]).astype(np.float32),
              "flow_2d": flow_2d, "flow_3d": flow_3d, "intrinsics": np.array([f, cx, cy], np.float32)}
    if dsec:
        sample["flow_3d"] = np.concatenate([flow_3d, (r.random((1, N)) < 0.9).astype(np.float32)])
    else:
        sample["occ_mask_3d"] = occ
    return sample


class SyntheticPairs(torch.utils.data.Dataset):
    """Sample i is frame_pair(1000 + i): every rank can regenerate any sample."""

    def __init__(self, n_samples, H=544, W=960, N=8192, dsec=False):
        self.n, self.kw = n_samples, dict(H=H, W=W, N=N, dsec=dsec)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {k: torch.from_numpy(v) for k, v in frame_pair(1000 + i, **self.kw).items()}
