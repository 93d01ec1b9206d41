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


SCLK_PLAUSIBLE_MHZ = (500.0, 2500.0)  # an engine clock outside this range on an MI355X under load is a broken reading


class ShaderClock:
    """The clock the shader engines ran at over a stretch of the CURRENT stream of ``device``, measured on the device the
    work runs on and inside the stream it runs in: ``with ShaderClock(dev) as c: <launches>`` puts one clock stamp
    (rpe_clock_stamp: s_memtime and s_memrealtime read by one wave) in front of and one behind the launches;
    ``c.mhz()`` (after the stream has been synchronised) = d(shader cycles) / d(constant-rate ticks) x the constant rate.
    ``None`` -- never a number -- when the result is not a plausible engine clock (e.g. a part whose s_memtime does not
    follow the engine clock): callers fall back to ``hwmon_sclk_mhz`` or print null.
    Measured on gfx950 (profiles/r05_corr_clock.json): the counter follows the engine clock -- 1.6 MHz over an idle half
    second, 2395 MHz under a compute loop with constant operands, 2023 MHz when the same loop hits the package power limit --
    and the cycles per launch it gives (804 k) equal GRBM_GUI_ACTIVE / 8 of a PMC pass (801 k).  The one-wave stamp kernel runs
    on the XCD that takes the first workgroup of a dispatch, and an idle XCD's counter stands still: the reading is the clock
    only for stretches that keep the whole chip busy (a launch with a long thin tail reads low)."""

    def __init__(self, device):
        import ctypes
        import torch
        self.device = torch.device(device)
        self.slots = torch.zeros(4, dtype=torch.int64, device=self.device)
        self.khz = ctypes.c_int(0)

    def _stamp(self, i):
        import ctypes
        import torch
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().rpe_clock_stamp(self.slots.data_ptr() + 16 * i, ctypes.byref(self.khz), _lib.stream_of(self.slots)), "clock_stamp")

    def __enter__(self):
        self._stamp(0)
        return self

    def __exit__(self, *exc):
        self._stamp(1)

    def raw(self):
        """(d shader cycles, d constant-rate ticks, constant rate in kHz)."""
        s0, w0, s1, w1 = self.slots.tolist()
        return s1 - s0, w1 - w0, self.khz.value

    def mhz(self):
        cycles, ticks, khz = self.raw()
        if ticks <= 0 or khz <= 0:
            return None
        mhz = cycles / ticks * khz / 1e3
        return round(mhz, 1) if SCLK_PLAUSIBLE_MHZ[0] <= mhz <= SCLK_PLAUSIBLE_MHZ[1] else None


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
