"""Host side of the four operators -- mirrors models/csrc/wrapper.py:40-127.

Same names, positional/keyword arguments, layout conventions and assertion
behaviour as the reference wrapper.  Differences, all deliberate:

* GPU tensors go to librpeflow_hip.so (C ABI, include/rpeflow_hip.h) on
  PyTorch's current stream; CPU tensors and ``cpp_impl=False`` raise, because
  this package ships no PyTorch/CPU path (the reference falls back to one,
  wrapper.py:67,100,124).
* Results equal the reference's *CPU fallback* (the parity oracle), not its
  CUDA kernels, where those differ: matmul-form distances, "first maximum"
  in FPS, index order inside KNN ties.
* No layout copies: KNN reads either point layout through strides (the
  reference transposes + copies, :119-122); correlation reads NCHW directly
  (the reference permutes both inputs to NHWC, :68-69).
"""
import ctypes

import torch

from .. import _lib

_NULL = ctypes.c_void_p(0)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _no_torch_path(op):
    raise NotImplementedError(
        f"{op}(cpp_impl=False): the pure-PyTorch path is not part of rpeflow_amd; "
        "use the reference implementation for it")


def _as_points(t, op, name):
    """[B,N,D] view + element strides of a [B,N,D] or [B,D,N] tensor (wrapper.py:119)."""
    if t.dim() != 3:
        raise RuntimeError(f"{op}: {name} must be 3-dimensional")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{op}: {name} must be a float tensor")  # k_nearest_neighbor.cpp:9
    return t


def squared_distance(xyz1: torch.Tensor, xyz2: torch.Tensor):
    """models/csrc/wrapper.py:40-52.  xyz1 [B,N1,D], xyz2 [B,N2,D], D<=3 -> [B,N1,N2]."""
    assert xyz1.shape[-1] == xyz2.shape[-1] and xyz1.shape[-1] <= 3  # assert channel_last
    _lib.require_gpu(xyz1, xyz2, op="squared_distance")
    xyz1, xyz2 = xyz1.float(), xyz2.float()
    B, N1, D = xyz1.shape
    N2 = xyz2.shape[1]
    out = torch.empty((B, N1, N2), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        rc = _lib.lib().rpe_squared_distance(_ptr(xyz1), *xyz1.stride(), _ptr(xyz2), *xyz2.stride(),
                                             B, N1, N2, D, _ptr(out), _lib.stream_of(xyz1))
    _lib.check(rc, "squared_distance")
    return out


class CorrelationFunction(torch.autograd.Function):
    """models/csrc/wrapper.py:18-37: the operator with its gradients (NCHW on both sides here)."""

    @staticmethod
    def forward(ctx, input1, input2, max_displacement):
        ctx.save_for_backward(input1, input2)
        ctx.max_displacement = int(max_displacement)
        return _correlation2d_forward(input1, input2, max_displacement)

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        B, C, H, W = input1.shape
        grad_output = grad_output.contiguous().float()
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g1 = torch.empty_like(input1) if need1 else None
        g2 = torch.empty_like(input2) if need2 else None
        with torch.cuda.device(input1.device):
            rc = _lib.lib().rpe_correlation2d_backward(_ptr(grad_output), _ptr(input1), _ptr(input2), B, C, H, W, ctx.max_displacement,
                                                       _ptr(g1) if need1 else None, _ptr(g2) if need2 else None,
                                                       _lib.stream_of(input1))
        _lib.check(rc, "correlation2d backward")
        return g1, g2, None


def correlation2d(input1: torch.Tensor, input2: torch.Tensor, max_displacement: int, cpp_impl=True):
    """models/csrc/wrapper.py:55-72.  NCHW in, [B,(2md+1)^2,H,W] out; differentiable in both inputs."""
    if not cpp_impl:
        _no_torch_path("correlation2d")
    _lib.require_gpu(input1, input2, op="correlation2d")
    assert input1.shape == input2.shape and input1.dim() == 4
    input1 = input1.contiguous().float()
    input2 = input2.contiguous().float()
    if torch.is_grad_enabled() and (input1.requires_grad or input2.requires_grad):
        return CorrelationFunction.apply(input1, input2, max_displacement)
    return _correlation2d_forward(input1, input2, max_displacement)


def _correlation2d_forward(input1, input2, max_displacement):
    B, C, H, W = input1.shape
    n = 2 * int(max_displacement) + 1
    out = torch.empty((B, n * n, H, W), dtype=torch.float32, device=input1.device)
    with torch.cuda.device(input1.device):
        rc = _lib.lib().rpe_correlation2d_forward(_ptr(input1), _ptr(input2), B, C, H, W, int(max_displacement),
                                                  0.0, 0, _ptr(out), _lib.stream_of(input1))
    _lib.check(rc, "correlation2d")
    return out


def furthest_point_sampling(xyz: torch.Tensor, n_samples: int, cpp_impl=True):
    """models/csrc/wrapper.py:75-103.  xyz [B,N,3] -> int64 [B,n_samples]."""
    assert xyz.shape[2] == 3 and xyz.shape[1] > n_samples
    if not cpp_impl:
        _no_torch_path("furthest_point_sampling")
    _lib.require_gpu(xyz, op="furthest_point_sampling")
    if xyz.dtype != torch.float32:
        raise RuntimeError("points_xyz must be a float tensor")  # furthest_point_sampling.cpp:8
    B, N, _ = xyz.shape
    idx = torch.empty((B, n_samples), dtype=torch.int64, device=xyz.device)
    with torch.cuda.device(xyz.device):
        rc = _lib.lib().rpe_fps(_ptr(xyz), *xyz.stride(), B, N, int(n_samples), _ptr(idx), _lib.stream_of(xyz))
    _lib.check(rc, "furthest_point_sampling")
    return idx


def k_nearest_neighbor(input_xyz: torch.Tensor, query_xyz: torch.Tensor, k: int, cpp_impl=True):
    """models/csrc/wrapper.py:106-127.  Points as [B,N,D] or [B,D,N] (D<=3) -> int64 [B,Q,k]; equal distances selected
    and ordered exactly as the reference's matmul + torch.topk does on the CPU."""
    return k_nearest_neighbor_ties(input_xyz, query_xyz, k, cpp_impl=cpp_impl, ties="torch")


def _knn_workspace(lib, sizes, D, k, mode, device):
    """Scratch for rpe_knn / rpe_knn_multi (include/rpeflow_hip.h): the binned cloud of a k = 1, D = 2 search, the tied-row
    queues of a large k >= 2 search; a fresh uninitialised tensor per call (inside a graph capture it comes from the graph's
    pool), None when the search needs none."""
    need = sum(lib.rpe_knn_workspace_bytes(B, M, Q, D, int(k), mode) for B, M, Q in sizes)
    return torch.empty(need, dtype=torch.uint8, device=device) if need > 0 else None


def k_nearest_neighbor_ties(input_xyz: torch.Tensor, query_xyz: torch.Tensor, k: int, cpp_impl=True, ties="torch", algo="auto",
                            return_distances=False):
    """k_nearest_neighbor with the treatment of EQUAL distances chosen per call (no global state): "torch" -- as above;
    "set" -- the reference's neighbour SET, equal distances inside it in index order (cheaper); "index" -- lowest index
    first (RPE_KNN_TIES_* of include/rpeflow_hip.h).
    ``algo``: "auto" | "sweep" (every query against every point) | "binned" (k = 1, D = 2: the cloud in a uniform cell grid,
    csrc/knn_binned.hip; what "auto" picks for the model's nearest-projected-point searches; raises where it does not apply);
    identical results whichever runs."""
    _as_points(input_xyz, "k_nearest_neighbor", "input_xyz")
    _as_points(query_xyz, "k_nearest_neighbor", "query_xyz")
    if input_xyz.shape[1] <= 3:  # channel_first to channel_last (a view; the kernel takes strides)
        assert query_xyz.shape[1] == input_xyz.shape[1]
        input_xyz = input_xyz.transpose(1, 2)
        query_xyz = query_xyz.transpose(1, 2)
    if not cpp_impl:
        _no_torch_path("k_nearest_neighbor")
    _lib.require_gpu(input_xyz, query_xyz, op="k_nearest_neighbor")
    B, M, D = input_xyz.shape
    Q = query_xyz.shape[1]
    if query_xyz.shape[0] != B or query_xyz.shape[2] != D:
        raise RuntimeError("k_nearest_neighbor: input_xyz and query_xyz disagree in batch or dimension")
    if k > M:
        raise RuntimeError("selected index k out of range")  # what the fallback's topk raises
    idx = torch.empty((B, Q, k), dtype=torch.int64, device=input_xyz.device)
    dist = torch.empty((B, Q, k), dtype=torch.float32, device=input_xyz.device) if return_distances else None
    lib, mode = _lib.lib(), _lib.KNN_TIES[ties] | _lib.KNN_ALGO[algo]
    if algo == "binned" and not (D == 2 and k == 1 and M >= 64):
        raise RuntimeError("k_nearest_neighbor: the binned search takes k = 1, D = 2, M >= 64")
    with torch.cuda.device(input_xyz.device):
        work = _knn_workspace(lib, [(B, M, Q)], D, k, mode, input_xyz.device)
        rc = lib.rpe_knn(_ptr(input_xyz), *input_xyz.stride(), _ptr(query_xyz), *query_xyz.stride(), B, M, Q, D, int(k), mode, _ptr(idx),
                         _ptr(dist) if dist is not None else _NULL, _ptr(work) if work is not None else _NULL,
                         work.numel() if work is not None else 0, _lib.stream_of(input_xyz))
    _lib.check(rc, "k_nearest_neighbor")
    return (idx, dist) if return_distances else idx


def k_nearest_neighbor_multi(pairs, k: int, ties="torch"):
    """[(input_xyz, query_xyz), ...] with one batch size, dimension and k -> [idx, ...], each exactly
    k_nearest_neighbor(input_xyz, query_xyz, k), from ONE call (at most 8 pairs; searches of one kind share a launch)."""
    jobs, outs, keep, sizes = (_lib.KnnJob * len(pairs))(), [], [], []
    B = D = None
    for i, (inp, qry) in enumerate(pairs):
        _as_points(inp, "k_nearest_neighbor", "input_xyz")
        _as_points(qry, "k_nearest_neighbor", "query_xyz")
        if inp.shape[1] <= 3:
            inp, qry = inp.transpose(1, 2), qry.transpose(1, 2)
        _lib.require_gpu(inp, qry, op="k_nearest_neighbor")
        if B is None:
            B, D = inp.shape[0], inp.shape[2]
        if inp.shape[0] != B or qry.shape[0] != B or inp.shape[2] != D or qry.shape[2] != D:
            raise RuntimeError("k_nearest_neighbor_multi: all pairs must share batch size and dimension")
        if k > inp.shape[1]:
            raise RuntimeError("selected index k out of range")
        idx = torch.empty((B, qry.shape[1], k), dtype=torch.int64, device=inp.device)
        jobs[i] = _lib.KnnJob(inp.data_ptr(), *inp.stride(), qry.data_ptr(), *qry.stride(), inp.shape[1], qry.shape[1],
                              idx.data_ptr(), None)
        outs.append(idx)
        keep += [inp, qry]
        sizes.append((B, inp.shape[1], qry.shape[1]))
    lib, mode = _lib.lib(), _lib.KNN_TIES[ties]
    with torch.cuda.device(outs[0].device):
        work = _knn_workspace(lib, sizes, D, k, mode, outs[0].device)
        rc = lib.rpe_knn_multi(ctypes.byref(jobs), len(pairs), B, D, int(k), mode, _ptr(work) if work is not None else _NULL,
                               work.numel() if work is not None else 0, _lib.stream_of(outs[0]))
    _lib.check(rc, "k_nearest_neighbor_multi")
    return outs


def k_nearest_neighbor_with_distances(input_xyz: torch.Tensor, query_xyz: torch.Tensor, k: int, ties="torch", algo="auto"):
    """k_nearest_neighbor plus the sorted squared distances the kernel selected on
    (what ``squared_distance(query, input).topk(k, largest=False).values`` holds)."""
    return k_nearest_neighbor_ties(input_xyz, query_xyz, k, ties=ties, algo=algo, return_distances=True)


def _correlation2d_algo(input1, input2, max_displacement, algo, leaky_slope=0.0):
    """correlation2d with an explicit kernel choice (1 direct, 2 MFMA) and the optional
    fused leaky_relu of RPEFlow_core.py:362; used by tests and bench."""
    _lib.require_gpu(input1, input2, op="correlation2d")
    input1, input2 = input1.contiguous().float(), input2.contiguous().float()
    B, C, H, W = input1.shape
    n = 2 * int(max_displacement) + 1
    out = torch.empty((B, n * n, H, W), dtype=torch.float32, device=input1.device)
    with torch.cuda.device(input1.device):
        rc = _lib.lib().rpe_correlation2d_forward(_ptr(input1), _ptr(input2), B, C, H, W, int(max_displacement),
                                                  float(leaky_slope), int(algo), _ptr(out), _lib.stream_of(input1))
    _lib.check(rc, "correlation2d")
    return out
