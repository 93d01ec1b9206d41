"""Event voxelisation on the GPU -- mirror of the reference's event_utils.eventsToVoxel (event_utils.py:109-128), which
its datasets call per sample on the CPU (flyingthings3d.py:206-208).  temporal_bilinear=True only (what the datasets use)."""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def events_to_voxel(events, num_bins=5, height=None, width=None, event_polarity=False):
    """events [N,4] (x, y, t, polarity) in time order, on the GPU (float64 like the reference's HDF5 arrays; float32 is
    up-cast).  Returns [num_bins, H, W], or [2*num_bins, H, W] with event_polarity (positive grids first), float32."""
    _lib.require_gpu(events, op="events_to_voxel")
    assert events.dim() == 2 and events.shape[1] == 4
    ev = events.double()
    n = ev.shape[0]
    xs, ys, pol = ev[:, 0].to(torch.int32), ev[:, 1].to(torch.int32), ev[:, 3].to(torch.int32)  # astype(np.int32), :24-26
    if height is None or width is None:
        width, height = int(xs.max()) + 1, int(ys.max()) + 1
    channels = 2 * num_bins if event_polarity else num_bins
    out = torch.zeros((channels, height, width), dtype=torch.float32, device=events.device)
    if n == 0:
        return out
    pixel = ys * width + xs
    if int(pixel.min()) < 0 or int(pixel.max()) >= height * width or int(xs.max()) >= width or int(xs.min()) < 0:
        raise IndexError("event coordinates outside the sensor")  # index_put_ raises in the reference too
    order = torch.sort(pixel, stable=True).indices
    pixel_s, t_s, pol_s = pixel[order].contiguous(), ev[:, 2][order].contiguous(), pol[order].contiguous()
    _, counts = torch.unique_consecutive(pixel_s, return_counts=True)
    run_start = (torch.cumsum(counts, 0) - counts).to(torch.int32).contiguous()
    t_first, t_last = float(ev[0, 2]), float(ev[-1, 2])
    with torch.cuda.device(events.device):
        rc = _lib.lib().rpe_events_to_voxel(_ptr(pixel_s), _ptr(t_s), _ptr(pol_s), _ptr(run_start), run_start.numel(), n,
                                            t_first, t_last, int(num_bins), int(bool(event_polarity)), height * width,
                                            _ptr(out), _lib.stream_of(out))
    _lib.check(rc, "events_to_voxel")
    return out
