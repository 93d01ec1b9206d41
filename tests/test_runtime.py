"""rpeflow_amd.runtime.configure(): the process settings of the entry points, per rank."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import json, os; from rpeflow_amd import runtime; out = runtime.configure(); "
        "print(json.dumps(dict(omp=os.environ['OMP_NUM_THREADS'], cores=runtime.usable_cores(), share=runtime.cores_per_rank(), "
        "local=out['local_world_size'], mark=os.environ.get(runtime.OMP_MARK))))")


def configure(**env):
    base = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "RPE_OMP_NUM_THREADS_BY_CONFIGURE")}
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(base, **env), capture_output=True, text=True, cwd=ROOT, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def test_openmp_threads_are_shared_out_between_the_ranks_of_a_node():
    alone = configure()
    assert int(alone["omp"]) == alone["cores"] and alone["local"] == 1 and alone["mark"] == "1"
    eight = configure(LOCAL_WORLD_SIZE="8", WORLD_SIZE="8")
    assert eight["local"] == 8 and int(eight["omp"]) == max(1, eight["cores"] // 8) == eight["share"]
    two = configure(WORLD_SIZE="2")  # a launcher that exports no LOCAL_WORLD_SIZE: one node assumed
    assert int(two["omp"]) == max(1, two["cores"] // 2)


def test_an_explicit_setting_wins_but_an_inherited_default_is_rederived():
    user = configure(OMP_NUM_THREADS="3", LOCAL_WORLD_SIZE="8")   # the user's (or torchrun's) own value: kept
    assert user["omp"] == "3" and user["mark"] is None
    # bench.py's launcher: the parent configured itself as a single rank (all cores, marked); its children re-derive their share
    child = configure(OMP_NUM_THREADS="16", RPE_OMP_NUM_THREADS_BY_CONFIGURE="1", LOCAL_WORLD_SIZE="8", WORLD_SIZE="8")
    assert int(child["omp"]) == max(1, child["cores"] // 8)


def test_threads_can_be_named_and_accounted():
    import threading
    import time
    from rpeflow_amd import runtime
    seen = {}

    def work():
        runtime.name_thread("rpe-test-thread")
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            pass
        seen.update({name: sec for (tid, name), sec in runtime.thread_cpu_seconds().items()})

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert seen.get("rpe-test-thread", 0.0) >= 0.1
