"""bench.py --gpus N without an outer launcher must start N ranks itself (the reference's pattern is
mp.spawn, train.py:289, 65).  Driven here on CPU tensors with gloo: same launcher, same rendezvous, same barrier + MAX
timing protocol, same single JSON line from rank 0 -- only the workload is a stand-in."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = run("--gpus", "2", "--workload", "selftest", "--backend", "gloo", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_launcher_world_size_must_match_gpus():
    """Under an outer launcher (WORLD_SIZE set) --gpus has to agree with it: a 1-rank run must not print n_gpus: 8."""
    r = run("--gpus", "2", "--workload", "selftest", "--backend", "gloo", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


def test_failed_rank_fails_the_launch():
    r = run("--gpus", "2", "--workload", "selftest", "--backend", "nccl")  # selftest refuses nccl: both ranks exit non-zero
    assert r.returncode != 0


def test_hotpath_sequence_runs_with_the_cpu_port():
    """bench.py's cpu_baseline leg (--workload hotpath) drives rpeflow_amd.hotpath.HotPathWorkload with the PyTorch-CPU port of
    the reference path as ``ops``: the harness must not depend on the native operators' extra arguments."""
    from types import SimpleNamespace

    import torch

    from oracle import torch_ref
    from rpeflow_amd.hotpath import OP_NAMES, HotPathWorkload

    ops = SimpleNamespace(**{n: getattr(torch_ref, n) for n in OP_NAMES})
    wl = HotPathWorkload(batch=1, height=64, width=128, n_points=8192, device="cpu", ops=ops)
    flow_3d, corr_2d = wl()
    assert flow_3d.shape == (1, 3, 8192) and corr_2d.shape[:2] == (1, 81)
    assert torch.isfinite(flow_3d).all() and torch.isfinite(corr_2d).all()
