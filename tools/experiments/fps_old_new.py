import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
t = (torch.rand(8, 8192, 3, generator=g) * 30).to(dev)
idx = torch.empty((8, 4096), dtype=torch.int64, device=dev)
here = os.path.dirname(os.path.abspath(__file__))
libs = {"round-2 fps.hip": ctypes.CDLL(os.path.join(here, "librpe_probefpsold.so")), "current": ctypes.CDLL(os.path.join(here, "..", "..", "rpeflow_amd", "csrc", "librpeflow_hip.so"))}
for name, lib in libs.items():
    lib.rpe_fps_algo.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
def run(lib): lib.rpe_fps_algo(t.data_ptr(), *t.stride(), 8, 8192, 4096, idx.data_ptr(), 2, None)
for rep in range(3):
    for name, lib in libs.items():
        for _ in range(3): run(lib)
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run(lib)
        e.record(); torch.cuda.synchronize()
        print("%-18s pruned kernel: %.1f us" % (name, s.elapsed_time(e) / 10 * 1e3), flush=True)
