#!/usr/bin/env python3
"""Host memcpy scaling over threads: numpy copyto (pageable -> pageable / pinned), ctypes.memmove, torch copy_ with one intra-op thread."""
import ctypes, sys, threading, time
import numpy as np, torch
pin = torch.cuda.is_available()
n = 42 << 18  # 42 MB of float32
src = [torch.randn(n) for _ in range(4)]
dst = [torch.empty(n, pin_memory=pin) for _ in range(4)]
def run(fn, threads, reps=6):
    def work(i):
        for _ in range(reps): fn(dst[i], src[i])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]
    return n * 4 * reps * threads / (time.perf_counter() - t0) / 1e9
def f_np(d, s): np.copyto(d.numpy(), s.numpy())
def f_mm(d, s): ctypes.memmove(d.data_ptr(), s.data_ptr(), n * 4)
def f_t1(d, s):
    torch.set_num_threads(1); d.copy_(s)
for name, fn in (("np.copyto", f_np), ("ctypes.memmove", f_mm), ("torch copy_ (1 intra-op thread)", f_t1)):
    fn(dst[0], src[0])
    print("%-34s" % name, "  ".join("%d thr: %5.1f GB/s" % (t, run(fn, t)) for t in (1, 2, 4)), "pinned" if pin else "pageable", flush=True)
