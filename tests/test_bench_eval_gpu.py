"""The sharded evaluation of bench.py (--workload eval) on TWO ranks gives the metrics of ONE rank over the same synthetic set:
rank-strided shards without padding, the input pipeline per rank, the forward replayed from each rank's own HIP graph, one
float64[12] SUM all-reduce (eval_withocc.py:43-135 + the template dist_reduce_sum, utils.py:26-31).  The GPU box has one GPU,
so both ranks share it (--share-gpu) and the collective runs on gloo; on a multi-GPU node the same command runs one rank per
GPU on RCCL.  The forward is bit-reproducible, so the sums agree to float64 re-association."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "eval", "--no-cpu-baseline", "--no-corr-microbench",
                        "--eval-distinct", "12", *argv], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_reach_the_metrics_of_one():
    one = run("--gpus", "1", "--steps", "4")                                  # 16 frame pairs on one rank, inside a world-size-1 RCCL group
    bare = run("--gpus", "1", "--steps", "4", "--backend", "none")            # the same without any process group
    # RCCL executed: process group created as a multi-GPU launch creates it (init_process_group("nccl", device_id=dev),
    # train.py:65), the HIP-graph capture beside its watchdog thread, the float64[12] SUM on the device (utils.py:26-31)
    assert "(nccl)" in one["eval"]["collective"] and one["eval"]["world_size"] == 1 and "none" in bare["eval"]["collective"]
    assert one["eval"]["metrics"] == bare["eval"]["metrics"], (one["eval"]["metrics"], bare["eval"]["metrics"])  # SUM over one rank: identity
    two = run("--gpus", "2", "--steps", "2", "--share-gpu", "--backend", "gloo")  # the same 16 over two ranks (8 each)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["eval"]["world_size"] == 2
    assert one["eval"]["samples"] == two["eval"]["samples"] == 16.0
    assert len(two["eval"]["per_rank_frame_pairs_per_s"]) == 2 and all(v > 0 for v in two["eval"]["per_rank_frame_pairs_per_s"])
    a, b = one["eval"]["metrics"], two["eval"]["metrics"]
    assert set(a) == set(b) and {"EPE2D", "EPE3D", "1px", "Fl", "5cm", "10cm", "EPE3D_noc"} <= set(a)
    for k in a:
        assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(a[k])), (k, a[k], b[k])
    assert one["epe_delta"]["within_bound"] and two["epe_delta"]["within_bound"]
