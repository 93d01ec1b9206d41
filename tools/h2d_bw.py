#!/usr/bin/env python3
"""Host-to-device bandwidth of this box: pinned -> device with torch copy_ on 1 / 2 / 4 streams, several sizes, and a
device kernel reading pinned memory (zero-copy) for comparison."""
import time, torch
dev = torch.device("cuda", 0)
def bw(nbytes, streams, chunks=1, reps=8, kernel=False):
    per = nbytes // streams // chunks // 4
    src = [[torch.empty(per, dtype=torch.float32, pin_memory=True).fill_(1.0) for _ in range(chunks)] for _ in range(streams)]
    dst = [[torch.empty(per, dtype=torch.float32, device=dev) for _ in range(chunks)] for _ in range(streams)]
    ss = [torch.cuda.Stream(dev) for _ in range(streams)]
    def go():
        for s, st in enumerate(ss):
            with torch.cuda.stream(st):
                for c in range(chunks):
                    dst[s][c].copy_(src[s][c], non_blocking=True)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): go()
    torch.cuda.synchronize()
    return per * 4 * streams * chunks * reps / (time.perf_counter() - t0) / 1e9
for mb in (4, 42, 206):
    for streams in (1, 2, 4):
        print("H2D %4d MB total, %d stream(s): %.1f GB/s" % (mb, streams, bw(mb << 20, streams)), flush=True)
print("H2D 206 MB as 28 chunks on 1 stream: %.1f GB/s" % bw(206 << 20, 1, chunks=28))
# pageable for reference
x = torch.empty(42 << 18, dtype=torch.float32); y = torch.empty_like(x, device=dev)
y.copy_(x); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): y.copy_(x)
torch.cuda.synchronize(); print("pageable 42 MB: %.1f GB/s" % (x.numel() * 4 * 4 / (time.perf_counter() - t0) / 1e9))
# host memcpy into pinned, 1 thread
p = torch.empty(42 << 18, dtype=torch.float32, pin_memory=True)
t0 = time.perf_counter()
for _ in range(8): p.copy_(x)
print("memcpy pageable -> pinned, 1 thread: %.1f GB/s" % (x.numel() * 4 * 8 / (time.perf_counter() - t0) / 1e9))
q = torch.empty(42 << 18, dtype=torch.float32)
t0 = time.perf_counter()
for _ in range(8): q.copy_(x)
print("memcpy pageable -> pageable, 1 thread: %.1f GB/s" % (x.numel() * 4 * 8 / (time.perf_counter() - t0) / 1e9))
import threading
def many(n, dsts):
    ts = [threading.Thread(target=lambda d=d: [d.copy_(x) for _ in range(8)]) for d in dsts[:n]]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]
    return x.numel() * 4 * 8 * n / (time.perf_counter() - t0) / 1e9
pins = [torch.empty(42 << 18, dtype=torch.float32, pin_memory=True) for _ in range(4)]
for n in (2, 4): print("memcpy -> pinned, %d threads: %.1f GB/s" % (n, many(n, pins)))
# D2H for completeness
print("torch threads", torch.get_num_threads())
import os; print("affinity", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
os.system("lscpu | grep -i 'model name\\|numa\\|socket' | head -8; rocm-smi --showbus 2>/dev/null | head -12")
