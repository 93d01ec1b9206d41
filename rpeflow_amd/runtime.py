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


def local_world_size():
    """Ranks that share this node's cores: LOCAL_WORLD_SIZE (torchrun, bench.py's own launcher), else WORLD_SIZE, else 1."""
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    except ValueError:
        return 1


def cores_per_rank():
    """This rank's share of the usable cores (>= 1): what its OpenMP team and its loader threads may count on."""
    return max(1, usable_cores() // local_world_size())


OMP_MARK = "RPE_OMP_NUM_THREADS_BY_CONFIGURE"  # set beside OMP_NUM_THREADS when configure() chose it: a launcher re-derives it per rank


def name_thread(name):
    """Name the calling thread for top -H / /proc/<pid>/task/*/comm (15 characters): the pipeline's threads can then be told
    from the runtime's in a per-thread CPU account (tools/host_rehearsal.py)."""
    try:
        import ctypes
        ctypes.CDLL(None).prctl(15, name.encode()[:15], 0, 0, 0)  # PR_SET_NAME
    except (OSError, AttributeError):
        pass


def thread_cpu_seconds():
    """{(tid, name): user + system CPU seconds} of every thread of this process (/proc/self/task)."""
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                stat = open("/proc/self/task/%s/stat" % tid).read()
                name = stat[stat.index("(") + 1:stat.rindex(")")]
                fields = stat[stat.rindex(")") + 2:].split()
                out[(int(tid), name)] = (int(fields[11]) + int(fields[12])) / tick  # utime, stime
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return out


def configure():
    """Idempotent; an explicit setting in the environment wins.  Returns what is in effect (bench.py prints it in ``config``).
    The MIOpen convolution solvers are left at the library's defaults everywhere -- tests, bench and evaluation alike.  The
    convolutions MIOpen would run through atomically accumulating split-K kernels (the source of the run-to-run differences of
    rounds 1-2) go through the deterministic GEMM paths of rpeflow_amd/utils.py instead, so the forward is bit-reproducible
    whatever MIOpen's timing-based search picks for the rest."""
    os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", GRAPH_QUEUES)
    # PyTorch sizes its OpenMP teams by the HOST's core count; inside a CPU quota that oversubscribes every host-side
    # tensor op (collate, Tensor.copy_) and gets the whole process throttled.  Read when OpenMP initialises.  The ranks of
    # a node share the quota: each gets usable cores / LOCAL_WORLD_SIZE (eight ranks x 16 threads on 16 cores otherwise).
    if "OMP_NUM_THREADS" not in os.environ or os.environ.get(OMP_MARK) == "1":
        os.environ["OMP_NUM_THREADS"] = str(cores_per_rank())
        os.environ[OMP_MARK] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # multi-process GPU work on this driver needs dmabuf IPC
    out = {"DEBUG_HIP_FORCE_GRAPH_QUEUES": os.environ["DEBUG_HIP_FORCE_GRAPH_QUEUES"], "OMP_NUM_THREADS": os.environ["OMP_NUM_THREADS"],
           "usable_cores": usable_cores(), "local_world_size": local_world_size(), "miopen_solvers": "library defaults"}
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


SCLK_PLAUSIBLE_MHZ = (500.0, 2500.0)  # an engine clock outside this range on an MI355X under load is a broken reading


class ShaderClock:
    """The clock the shader engines ran at over a stretch of the CURRENT stream of ``device``, measured on the device the work
    runs on, in the stream it runs in: ``with ShaderClock(dev) as c: <launches>`` puts one rpe_clock_stamp_all in front of and
    one behind the launches (every compute unit's engine cycle counter s_memtime and the constant-rate s_memrealtime, stored
    per compute unit); ``c.mhz()`` (after the stream has been synchronised) = the median over the compute units of
    d(cycles) / d(ticks) x the constant rate, ``c.mhz_per_xcd()`` the median of each XCD.  ``None`` -- never a number -- when
    that is not a plausible engine clock: callers fall back to ``hwmon_sclk_mhz`` or print null.
    A compute unit's counter is only ever compared with itself.  The cycle counters of different parts of the chip are not one
    clock: two stamps taken wherever a one-workgroup kernel landed agreed with GRBM_GUI_ACTIVE / 8 in seven runs and read 1455
    "MHz" in the eighth (after a stretch of 8-workgroup launches the counters had drifted apart), one slot per XCD gave negative
    differences; and probes that run BESIDE the measured launches take compute units and a hardware queue from them (the
    correlation loop slowed from 402 to 492 us, or stalled behind the probes' queue).  Under the package power limit the XCDs
    do not run at one clock either.  An idle compute unit's counter stands still: the reading is the clock only for stretches that
    keep the chip busy."""

    KEYS = 2048

    def __init__(self, device):
        import ctypes
        import torch
        self.device = torch.device(device)
        self.slots = torch.zeros((2, self.KEYS, 2), dtype=torch.int64, device=self.device)  # [stamp][compute unit][(cycles, ticks)]
        self.khz = ctypes.c_int(0)

    def _stamp(self, i):
        import ctypes
        import torch
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().rpe_clock_stamp_all(self.slots[i].data_ptr(), ctypes.byref(self.khz), _lib.stream_of(self.slots)), "clock_stamp_all")

    def __enter__(self):
        self.slots.zero_()  # a reused object must not keep a first stamp for a compute unit this use does not reach
        self._stamp(0)
        return self

    def __exit__(self, *exc):
        self._stamp(1)

    def per_unit(self):
        """{key: (d cycles, d ticks)} for every compute unit both stamps reached."""
        first, second = self.slots.tolist()
        return {k: (b[0] - a[0], b[1] - a[1]) for k, (a, b) in enumerate(zip(first, second)) if a[1] > 0 and b[1] > a[1] and b[0] > a[0]}

    def raw(self):
        """(d engine cycles, d constant-rate ticks, constant rate in kHz) of the median compute unit."""
        pairs = sorted(self.per_unit().values(), key=lambda p: p[0] / p[1])
        if not pairs:
            return 0, 0, self.khz.value
        cycles, ticks = pairs[len(pairs) // 2]
        return cycles, ticks, self.khz.value

    def mhz_per_xcd(self):
        """Median clock of the compute units of each XCD (XCDs no unit of which was reached by both stamps are left out)."""
        khz = self.khz.value
        by = {}
        for k, (c, t) in self.per_unit().items():
            by.setdefault(k >> 8, []).append(c / t * khz / 1e3)
        return [round(sorted(v)[len(v) // 2], 1) for _, v in sorted(by.items())] if khz > 0 else []

    def units(self):
        return len(self.per_unit())

    def mhz(self):
        cycles, ticks, khz = self.raw()
        if ticks <= 0 or khz <= 0:
            return None
        mhz = round(cycles / ticks * khz / 1e3, 1)
        return mhz if SCLK_PLAUSIBLE_MHZ[0] <= mhz <= SCLK_PLAUSIBLE_MHZ[1] else None


def hwmon_sclk_mhz(device):
    """Fallback reading: amdgpu's hwmon node OF THE DEVICE PyTorch runs on (found through its PCI address), only if the node
    is labelled "sclk" and the value is a plausible engine clock; else None.  One instant, not an average."""
    import glob
    import torch
    try:
        prop = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)  # (integers in torch's properties)
    except (AttributeError, RuntimeError, TypeError):
        return None
    for label in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/freq*_label" % bdf):
        try:
            if open(label).read().strip() != "sclk":
                continue
            mhz = float(open(label.replace("_label", "_input")).read()) / 1e6
        except (OSError, ValueError):
            continue
        if SCLK_PLAUSIBLE_MHZ[0] <= mhz <= SCLK_PLAUSIBLE_MHZ[1]:
            return round(mhz, 1)
    return None
