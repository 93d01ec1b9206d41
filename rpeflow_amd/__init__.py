"""rpeflow_amd -- RPEFlow's data-parallel hot path on MI355X (gfx950).

``rpeflow_amd.csrc`` exports the four operators of the reference's
``models/csrc/__init__.py:1`` with the same signatures; they call hand-written
HIP kernels in ``librpeflow_hip.so`` through a C ABI (``include/rpeflow_hip.h``).
There is no CPU or PyTorch fallback: a missing library or a CPU tensor raises.
"""
__version__ = "0.1.0"

import os as _os

# Replaying a captured multi-stream HIP graph, the runtime spreads the graph's branches over this many hardware queues
# (its default: 4).  The forward's graph has four branches -- 2-D chain, 3-D chain, hoisted per-level work, the next
# batch's sampling -- of ~1400 small kernels; with three queues the replay is 3 % faster (216 vs 210 frame-pairs/s,
# five A/B runs; two queues: 179).  Read once when the HIP runtime initialises, so it is set here, before anything
# touches the GPU; an explicit setting in the environment wins.
_os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "3")
