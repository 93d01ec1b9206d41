"""Event voxelisation on the GPU -- mirror of the reference's event_utils.eventsToVoxel (event_utils.py:109-128), which
its datasets call per sample on the CPU (flyingthings3d.py:206-208).  temporal_bilinear=True only (what the datasets use)."""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def events_to_voxel(events, num_bins=5, height=None, width=None, event_polarity=False, out=None, t_range=None, validate=True):
    """events [N,4] (x, y, t, polarity) in time order, on the GPU.  float64 arrays are voxelised in float64 arithmetic, float32
    arrays -- what the reference's load_events_h5 hands its datasets (event_utils.py:11-20) -- in float32 arithmetic: each as
    numpy and torch treat such an array in eventsToVoxel.  Returns [num_bins, H, W], or [2*num_bins, H, W] with
    event_polarity (positive grids first), float32.
    ``out``: the grid to fill (zeroed here); ``t_range`` = (t of the first event, t of the last) if the caller knows them on
    the host and ``validate=False`` (coordinates already checked) spare the device round trips -- the input pipeline's
    voxelisation stage runs beside the forward of the previous batch and passes all three."""
    _lib.require_gpu(events, op="events_to_voxel")
    assert events.dim() == 2 and events.shape[1] == 4
    ev = events if events.dtype == torch.float32 else events.double()
    f32 = ev.dtype == torch.float32
    n = ev.shape[0]
    xs, ys, pol = ev[:, 0].to(torch.int32), ev[:, 1].to(torch.int32), ev[:, 3].to(torch.int32)  # astype(np.int32), :24-26
    if height is None or width is None:
        width, height = int(xs.max()) + 1, int(ys.max()) + 1
    channels = 2 * num_bins if event_polarity else num_bins
    if out is None:
        out = torch.zeros((channels, height, width), dtype=torch.float32, device=events.device)
    else:
        assert out.shape == (channels, height, width) and out.dtype == torch.float32 and out.is_contiguous() and out.device == events.device
        out.zero_()
    if n == 0:
        return out
    pixel = ys * width + xs
    if validate and (int(pixel.min()) < 0 or int(pixel.max()) >= height * width or int(xs.max()) >= width or int(xs.min()) < 0):
        raise IndexError("event coordinates outside the sensor")  # index_put_ raises in the reference too
    order = torch.sort(pixel, stable=True).indices
    pixel_s, t_s, pol_s = pixel[order].contiguous(), ev[:, 2][order].contiguous(), pol[order].contiguous()
    _, counts = torch.unique_consecutive(pixel_s, return_counts=True)
    run_start = (torch.cumsum(counts, 0) - counts).to(torch.int32).contiguous()
    t_first, t_last = (float(ev[0, 2]), float(ev[-1, 2])) if t_range is None else (float(t_range[0]), float(t_range[1]))
    with torch.cuda.device(events.device):
        rc = _lib.lib().rpe_events_to_voxel(_ptr(pixel_s), _ptr(t_s), int(f32), _ptr(pol_s), _ptr(run_start), run_start.numel(), n,
                                            t_first, t_last, int(num_bins), int(bool(event_polarity)), height * width,
                                            _ptr(out), _lib.stream_of(out))
    _lib.check(rc, "events_to_voxel")
    return out
