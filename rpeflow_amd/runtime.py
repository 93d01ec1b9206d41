"""Process-level settings of the HIP runtime for this package's entry points (bench.py, rpeflow_amd.evaluate).

Nothing here runs at import: a library must not change process-wide runtime knobs behind its user's back.  An entry
point calls ``configure()`` before anything touches the GPU; embedding applications decide for themselves
(INTEGRATION.md lists the settings)."""
import os

# Replaying a captured multi-stream HIP graph, the runtime spreads the graph's branches over this many hardware queues
# (its default: 4).  The forward's graph has four branches -- 2-D chain, 3-D chain, hoisted per-level work, the next
# batch's sampling -- of ~900 small kernels.  Rounds 1-2 ran it on three queues (216 vs 210 frame-pairs/s then; two: 179):
# fewer queues, cheaper dispatch of the small kernels.  Since round 3 the hoisted work starts beside the event pyramid and the
# first decoder level is ready before the next batch's sampling (3.2 ms) ends; on three queues the main chain then sits
# behind that sampling kernel in a shared queue until it finishes, on four it does not: 16.34 vs 16.58 ms per batch (median
# of 40, two runs each, tools/ab_step.sh).  The variable is read once, when the HIP runtime initialises.
GRAPH_QUEUES = "4"


def usable_cores():
    """Cores this process may really use: affinity mask, capped by a cgroup CPU quota if there is one (os.cpu_count()
    reports the host's cores even inside a quota-limited container: 256 on the GPU box, which grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def configure():
    """Idempotent; an explicit setting in the environment wins.  Returns what is in effect (bench.py prints it in ``config``).
    The MIOpen convolution solvers are left at the library's defaults everywhere -- tests, bench and evaluation alike.  The
    convolutions MIOpen would run through atomically accumulating split-K kernels (the source of the run-to-run differences of
    rounds 1-2) go through the deterministic GEMM paths of rpeflow_amd/utils.py instead, so the forward is bit-reproducible
    whatever MIOpen's timing-based search picks for the rest."""
    os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", GRAPH_QUEUES)
    # PyTorch sizes its OpenMP teams by the HOST's core count; inside a CPU quota that oversubscribes every host-side
    # tensor op (collate, Tensor.copy_) and gets the whole process throttled.  Read when OpenMP initialises.
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # multi-process GPU work on this driver needs dmabuf IPC
    out = {"DEBUG_HIP_FORCE_GRAPH_QUEUES": os.environ["DEBUG_HIP_FORCE_GRAPH_QUEUES"], "OMP_NUM_THREADS": os.environ["OMP_NUM_THREADS"],
           "miopen_solvers": "library defaults"}
    import sys
    torch = sys.modules.get("torch")
    if torch is not None:  # imported before this call (python -m rpeflow_amd.evaluate, an embedder): libgomp has read the
        try:               # environment already, so cap the intra-op team through the API and report what is in effect
            torch.set_num_threads(min(torch.get_num_threads(), int(os.environ["OMP_NUM_THREADS"])))
        except (RuntimeError, ValueError):
            pass
        out["torch_num_threads"] = torch.get_num_threads()
        out["hip"] = getattr(torch.version, "hip", None)
        out["torch"] = torch.__version__
    return out
