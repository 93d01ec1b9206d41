"""ctypes binding of librpeflow_hip.so (include/rpeflow_hip.h).

The library is loaded on first use.  If it is missing the operators raise:
there is no PyTorch or CPU fallback behind this boundary.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RPE_HIP_LIB") or os.path.join(_HERE, "csrc", "librpeflow_hip.so")  # (RPE_HIP_LIB: diagnostic builds, e.g. tools/corr_energy_probes.sh)

_c_f32p = ctypes.c_void_p
_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_ptr = ctypes.c_void_p

# name -> argtypes, exactly the prototypes of include/rpeflow_hip.h
_PROTOTYPES = {
    "rpe_abi_version": [],
    "rpe_knn_workspace_bytes": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int],
    "rpe_knn": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64,
                _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr],
    "rpe_squared_distance": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64,
                             _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_fps": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_fps_algo": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_ptr, _c_int, _c_ptr],
    "rpe_correlation2d_forward": [_c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  _c_float, _c_int, _c_ptr, _c_ptr],
    "rpe_dwconv3": [_c_ptr, _c_int, _c_ptr, _c_int, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                    _c_ptr, _c_ptr],
    "rpe_channel_layernorm": [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_i64, _c_float,
                              _c_ptr],
    "rpe_channel_attention_matrix": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_i64, _c_float,
                                     _c_ptr, _c_ptr, _c_int, _c_ptr],
    "rpe_channel_attention_workspace_floats": [_c_int, _c_int, _c_int, _c_i64],
    "rpe_gdfn_tail": [_c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    "rpe_convex_upsample": [_c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_events_to_voxel": [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_int, _c_int, ctypes.c_double, ctypes.c_double, _c_int, _c_int,
                            _c_i64, _c_ptr, _c_ptr],
    "rpe_channel_affine_act": [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_i64, _c_int, _c_float, _c_ptr],
    "rpe_channel_affine_add_act": [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_i64, _c_int, _c_float, _c_ptr],
    "rpe_residual_tail": [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_ptr],
    "rpe_corr3d_cost": [_c_ptr] * 11 + [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64,
                                        _c_int, _c_int, _c_int, _c_int, _c_float, _c_ptr, _c_ptr],
    "rpe_corr3d_n2n": [_c_ptr] * 7 + [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_correlation2d_backward": [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr],
    "rpe_knn_multi": [_c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_i64, _c_ptr],
    "rpe_clock_stamp": [_c_ptr, _c_ptr],
    "rpe_clock_stamp_all": [_c_ptr, _c_ptr, _c_ptr],
    "rpe_gather_channel_first": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_gather_channel_last": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_pointwise_conv": [_c_ptr, _c_i64, _c_int, _c_int, _c_i64, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_int, _c_float,
                           _c_ptr, _c_i64, _c_ptr, _c_ptr],
    "rpe_im2col": [_c_ptr] + [_c_int] * 12 + [_c_ptr, _c_ptr, _c_int, _c_float, _c_ptr, _c_ptr],
    "rpe_knn_interpolate": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int,
                            _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_float,
                            _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr],
    "rpe_resize_frames": [_c_ptr, _c_int, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_resize_flow2d": [_c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_ptr, _c_ptr],
    "rpe_upsample2x_pair": [_c_ptr, _c_int, _c_float, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr],
    "rpe_bilinear_sample": [_c_ptr, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int,
                            _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_project_points": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_ptr, _c_i64,
                           _c_float, _c_float, _c_float, _c_float, _c_ptr, _c_ptr],
    "rpe_project_feat_nn_corr": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_int, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64,
                                 _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_float, _c_float,
                                 _c_float, _c_float, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr],
    "rpe_pointconv_pack_rows": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_ptr, _c_ptr],
    "rpe_pointconv_fused": [_c_ptr, _c_int, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                            _c_float, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_int, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int,
                            _c_ptr, _c_ptr],
    "rpe_mlp1d_fused": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_int, _c_int,
                        _c_float, _c_int, _c_int, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr],
    "rpe_ids_forward": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_int, _c_int, _c_int,
                        _c_float, _c_float, _c_float, _c_float, _c_float, _c_ptr, _c_ptr],
    "rpe_eval_workspace_doubles": [_c_i64, _c_i64],
    "rpe_eval_accumulate": [_c_ptr, _c_ptr, _c_int, _c_int, _c_i64, _c_ptr, _c_ptr, _c_int, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    "rpe_ids_flow_inverse": [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_int, _c_int,
                             _c_float, _c_float, _c_float, _c_float, _c_float, _c_ptr, _c_ptr],
}

_lib = None
ABI_VERSION = 9  # RPE_ABI_VERSION of include/rpeflow_hip.h
KNN_TIES = {"torch": 3, "set": 1, "index": 0}  # RPE_KNN_TIES_* (how equal distances are resolved)
KNN_ALGO = {"auto": 0, "sweep": 0x100, "binned": 0x200, "matrix": 0x400, "insert": 0x800}  # RPE_KNN_ALGO_* (OR-ed into the mode)
# entry points only a library built with -DRPE_EXPERIMENTAL has (python -m rpeflow_amd.build --experimental)
_EXPERIMENTAL = {"rpe_probe_mfma4x4": [_c_ptr, _c_ptr]}


class SampleSource(ctypes.Structure):
    """rpe_sample_source of include/rpeflow_hip.h."""
    _fields_ = [("data", ctypes.c_void_p), ("sb", ctypes.c_int64), ("sc", ctypes.c_int64), ("channels", ctypes.c_int),
                ("scale_even", ctypes.c_float), ("scale_odd", ctypes.c_float), ("div_even", ctypes.c_float), ("div_odd", ctypes.c_float),
                ("subtract", ctypes.c_void_p),
                ("sub_sb", ctypes.c_int64), ("sub_sc", ctypes.c_int64), ("sub_sp", ctypes.c_int64)]


class KnnJob(ctypes.Structure):
    """rpe_knn_job of include/rpeflow_hip.h."""
    _fields_ = [("input", ctypes.c_void_p), ("in_sb", ctypes.c_int64), ("in_sn", ctypes.c_int64), ("in_sd", ctypes.c_int64),
                ("query", ctypes.c_void_p), ("q_sb", ctypes.c_int64), ("q_sn", ctypes.c_int64), ("q_sd", ctypes.c_int64),
                ("M", ctypes.c_int), ("Q", ctypes.c_int), ("idx", ctypes.c_void_p), ("dist", ctypes.c_void_p)]


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m rpeflow_amd.build` "
                "(hipcc --offload-arch=gfx950). rpeflow_amd has no CPU/PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = _c_int
        handle.rpe_channel_attention_workspace_floats.restype = _c_i64
        handle.rpe_knn_workspace_bytes.restype = _c_i64
        for name, argtypes in _EXPERIMENTAL.items():
            if hasattr(handle, name):
                getattr(handle, name).argtypes, getattr(handle, name).restype = argtypes, _c_int
        handle.rpe_error_string.argtypes = [_c_int]
        handle.rpe_error_string.restype = ctypes.c_char_p
        version = handle.rpe_abi_version()
        if version >> 16 and os.environ.get("RPE_ALLOW_DIAGNOSTIC_LIB") != "1":
            raise RuntimeError(f"{LIB_PATH} is a DIAGNOSTIC build (variant {version >> 16}): its kernels compute wrong results on purpose "
                               "(tools/corr_energy_probes.sh).  Unset RPE_HIP_LIB, or set RPE_ALLOW_DIAGNOSTIC_LIB=1 for a timing run")
        if version & 0xffff != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: ABI version {version & 0xffff}, this package binds version {ABI_VERSION}")
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().rpe_error_string(code).decode()
        raise RuntimeError(f"{what}: {msg} (code {code})")


def stream_of(t: torch.Tensor):
    """The HIP stream PyTorch is currently issuing on for t's device."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors, op):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                f"{op}: rpeflow_amd operators run on the GPU only (got a {t.device} tensor); "
                "there is no CPU fallback in this package")
    dev = tensors[0].device
    for t in tensors[1:]:
        if t.device != dev:
            raise RuntimeError(f"{op}: tensors on different devices ({dev} vs {t.device})")
    return dev
