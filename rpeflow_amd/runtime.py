"""Process-level settings of the HIP runtime for this package's entry points (bench.py, rpeflow_amd.evaluate).

Nothing here runs at import: a library must not change process-wide runtime knobs behind its user's back.  An entry
point calls ``configure()`` before anything touches the GPU; embedding applications decide for themselves
(INTEGRATION.md lists the settings)."""
import os

# Replaying a captured multi-stream HIP graph, the runtime spreads the graph's branches over this many hardware queues
# (its default: 4).  The forward's graph has four branches -- 2-D chain, 3-D chain, hoisted per-level work, the next
# batch's sampling -- of ~1400 small kernels; with three queues the replay is 3 % faster (216 vs 210 frame-pairs/s,
# five A/B runs; two queues: 179).  The variable is read once, when the HIP runtime initialises.
GRAPH_QUEUES = "3"


# MIOpen picks a convolution's solver by TIMING the candidates the first time a process meets a shape that its user
# find-db does not hold, and several candidates of this model's 3x3 convolutions time within a few percent of each other
# (two Winograd variants, an implicit GEMM): on a fresh machine the first process therefore landed on different solver
# sets from run to run -- 222 or 202-214 frame-pairs/s, and EPEs 8e-6 or 8e-5 away from the reference's (both inside the
# 1e-4 bound, but not the same numbers).  The find results of one run over both benched configurations are kept in
# miopen_db/ and seed the user find-db of every process that calls configure() -- tests, bench and evaluation alike: the
# library's own default selection, made once instead of once per fresh machine.  Shapes the file does not hold are
# searched and appended as usual (in the per-user copy, never in the package).
_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def _seed_miopen_find_db():
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    # the working copy lives beside the package (git-ignored), or in the temp directory if the tree is read-only
    import tempfile
    candidates = [os.path.join(os.path.dirname(os.path.dirname(_DB_DIR)), ".miopen_user_db"),
                  os.path.join(tempfile.gettempdir(), "rpeflow_amd_miopen_user_db_%d" % os.getuid())]
    target = None
    for c in candidates:
        try:
            os.makedirs(c, exist_ok=True)
            if os.access(c, os.W_OK):
                target = c
                break
        except OSError:
            continue
    if target is None:
        return None
    try:
        for name in os.listdir(_DB_DIR):
            dst = os.path.join(target, name)
            if name.endswith(".ufdb.txt") and not os.path.exists(dst):
                tmp = "%s.%d.tmp" % (dst, os.getpid())  # (several ranks may start at once: write aside, then rename)
                with open(os.path.join(_DB_DIR, name), "rb") as f, open(tmp, "wb") as g:
                    g.write(f.read())
                os.replace(tmp, dst)
    except OSError:
        return None  # MIOpen keeps its own default location and searches as it always did
    os.environ["MIOPEN_USER_DB_PATH"] = target
    return target


def configure():
    """Idempotent; an explicit setting in the environment wins.  Returns what is in effect (bench.py prints it in ``config``).
    The MIOpen convolution solvers are the library's default selection everywhere -- tests, bench and evaluation alike --
    from a find-db seeded with the recorded search results of the benched shapes (above); with them the benched
    configuration is within 1e-5 of the reference's EPEs (tests/test_model.py)."""
    os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", GRAPH_QUEUES)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # multi-process GPU work on this driver needs dmabuf IPC
    db = _seed_miopen_find_db()
    return {"DEBUG_HIP_FORCE_GRAPH_QUEUES": os.environ["DEBUG_HIP_FORCE_GRAPH_QUEUES"],
            "miopen_solvers": "library defaults" + (", user find-db seeded from rpeflow_amd/miopen_db" if db else "")}
